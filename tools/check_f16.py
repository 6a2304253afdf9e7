"""A/B of the scaled split-fp16 kernels (precision 3) against the six-product bf16 kernels on the same operands (one GPU box)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speech_enhancement_amd import gemm as GM, _lib as L, layers as LY, ops as O
from speech_enhancement_amd.weights import WeightPlan

torch.manual_seed(0)
dev = torch.device('cuda')
M = 4096 + 37


def rel(a, b):
    return ((a.double() - b.double()).abs().max() / b.double().abs().max()).item()


def plans(build):
    out = []
    for kind in (True, 'f16'):
        p = WeightPlan(dev)
        build(p, kind)
        p.run()
        out.append(p)
    return out


x = torch.randn(M, 64, device=dev)
st = O.row_stats(x, M)
g, b = torch.rand(64, device=dev) + 0.5, torch.randn(64, device=dev) * 0.1
W1, b1 = torch.randn(256, 64, device=dev) * 0.1, torch.randn(256, device=dev) * 0.1
W2, b2 = torch.randn(64, 256, device=dev) * 0.1, torch.randn(64, device=dev) * 0.1
pb, pf = plans(lambda p, k: (p.linear('w1', W1, planes=k), p.linear('w2', W2, planes=k), p.linear_T('w2t', W2, planes=k, scale=0.5),
                             p.linear_T('w1t', W1, planes=k)))
ref64 = None
for name, p in (('bf16x6', pb), ('f16x3', pf)):
    y, h = GM.ff_fwd(x, st, g, b, p.out['w1'], b1, p.out['w2'], b2, 0.0, 1, 2, 0.5)
    xl = ((x.double() - st[:, :1].double()) * st[:, 1:].double()) * g.double() + b.double()
    h64 = xl @ W1.double().t() + b1.double()
    y64 = x.double() + 0.5 * ((h64 * torch.sigmoid(h64)) @ W2.double().t() + b2.double())
    print('ff_fwd', name, 'H', rel(h, h64), 'Y', rel(y, y64))
    dy = torch.randn(M, 64, device=dev) * 1e-4
    dy._se_amax = dy.abs().max().reshape(1).clone()
    dg, db = torch.zeros(64, device=dev), torch.zeros(64, device=dev)
    am = (torch.zeros(1, device=dev), torch.zeros(1, device=dev))
    dz, dx = GM.ff_bwd_dgrad(dy, h, p.out['w2t'], p.out['w1t'], 0.0, 1, 2, ln=(x, st, g, None, dg, db), amax_out=am)
    dh64 = (dy.double() @ (0.5 * W2.double())) * (torch.sigmoid(h64) * (1 + h64 * (1 - torch.sigmoid(h64))))
    print('ff_bwd', name, 'dZ', rel(dz, dh64), 'amax dz', float(am[1]), float(dz.abs().max()), 'amax dx', float(am[0]), float(dx.abs().max()))

Wq = torch.randn(192, 64, device=dev) * 0.1
Wp, bp = torch.randn(256, 64, device=dev) * 0.1, torch.randn(256, device=dev) * 0.1
pb, pf = plans(lambda p, k: (p.linear('q', Wq, planes=k), p.linear('p', Wp, planes=k)))
xl = (((x.double() - st[:, :1].double()) * st[:, 1:].double()) * g.double() + b.double())
for name, p in (('bf16x6', pb), ('f16x3', pf)):
    q = torch.empty(M, 192, device=dev)
    GM.gemm_tap(GM.linear_desc(M, 64, 192, prologue=L.PRO_LN, **LY._lin3(p.out['q'], a_sexp=GM.LN_SEXP)), x, p.out['q'], q, rowstats=st, ps=g, pb=b)
    print('panel LN', name, rel(q, xl @ Wq.double().t()))
    u, zc = torch.empty(M, 128, device=dev), torch.empty(M, 256, device=dev)
    GM.gemm_tap(GM.linear_desc(M, 64, 256, ldc=128, prologue=L.PRO_LN, epilogue=L.EPI_BIAS | L.EPI_GLU, ldx=256,
                               **LY._lin3(p.out['p'], a_sexp=GM.LN_SEXP)), x, p.out['p'], u, bias=bp, AUX=zc, rowstats=st, ps=g, pb=b)
    z64 = xl @ Wp.double().t() + bp.double()
    print('panel GLU', name, 'z', rel(zc, z64), 'u', rel(u, z64[:, :128] * torch.sigmoid(z64[:, 128:])))

# strided conv (generic kernel)
B, T, Fq = 2, 9, 201
a2 = torch.randn(B, T, Fq, 64, device=dev)
wc, bc = torch.randn(64, 64, 1, 3, device=dev) * 0.05, torch.randn(64, device=dev) * 0.1
pb, pf = plans(lambda p, k: p.conv_fwd('c', wc, planes=k))
ref = torch.nn.functional.conv2d(a2.double().permute(0, 3, 1, 2), wc.double(), bc.double(), stride=(1, 2), padding=(0, 1)).permute(0, 2, 3, 1)
for name, p in (('bf16x6', pb), ('f16x3', pf)):
    R, _ = LY.conv_fwd(a2, B, T, Fq, 64, 0, 64, p.out['c'], bc, LY.TAPS_1x3, 64, To=T, Fo=101, sf=2)
    print('strided conv', name, rel(R, ref))
