"""attention kernels against fp64 torch: three-way bf16 split vs scaled split-fp16, model-like operand magnitudes (dO ~ 1e-4)"""
import sys, torch
import os; R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, R + '/tests')
from test_attn_gpu import ref_attention
from speech_enhancement_amd import attention as A
from speech_enhancement_amd.weights import WeightPlan
torch.manual_seed(0)
for (B, T, Fq, axis) in ((1, 321, 4, 'time'), (4, 6, 101, 'freq')):
    maxpos = 512
    g = torch.Generator().manual_seed(3)
    qkv = (torch.randn(B, T, Fq, 192, generator=g) * 1.5).cuda()
    E = (torch.randn(2 * maxpos + 1, 16, generator=g) * 0.7).cuda()
    dO = (torch.randn(B, T, Fq, 64, generator=g) * 1e-4).cuda()
    geom = A.seq_geometry(B, T, Fq, axis)
    plan = WeightPlan(torch.device('cuda')); Es = plan.linear('e', E, planes='f16'); plan.run()
    am = qkv.abs().max().reshape(1).clone(); dam = dO.abs().max().reshape(1).clone()
    q64 = qkv.double().requires_grad_(True); E64 = E.double().requires_grad_(True)
    ref = ref_attention(q64, E64, B, T, Fq, axis, maxpos, 0.25); ref.backward(dO.double())
    for name, kw_f, kw_b in (('bf16x6', {}, {}), ('f16x3', dict(Es=Es, qkv_amax=am), dict(qkv_amax=am, do_amax=dam))):
        O, lse = A.attn_fwd(qkv.view(-1, 192), E, geom, maxpos=maxpos, **kw_f)
        dE = torch.zeros_like(E)
        dq = A.attn_bwd(qkv.view(-1, 192), E, O, dO.view(-1, 64), lse, geom, dE, maxpos=maxpos, **kw_b).view(B, T, Fq, 192)
        rel = lambda a, b: float((a.double() - b).abs().max() / b.abs().max())
        rms = lambda a, b: float(((a.double() - b) ** 2).mean().sqrt() / (b ** 2).mean().sqrt())
        gr = q64.grad
        print(f'{axis} n={geom[1]} {name}: O max {rel(O.view(B,T,Fq,64), ref.detach()):.1e} | dq max {rel(dq[...,:64], gr[...,:64]):.1e} rms {rms(dq[...,:64], gr[...,:64]):.1e} | dk max {rel(dq[...,64:128], gr[...,64:128]):.1e} rms {rms(dq[...,64:128], gr[...,64:128]):.1e} | dv max {rel(dq[...,128:], gr[...,128:]):.1e} | dE max {rel(dE, E64.grad):.1e} rms {rms(dE, E64.grad):.1e}')
