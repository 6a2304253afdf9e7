"""time-axis / frequency-axis attention backward (scaled split-fp16) at the bench shapes, HIP-event timing of the phase-1 kernel
family; environment switches are read by the library at load: run once per variant.  usage: python tools/attn_bwd_ab.py [time|freq]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speech_enhancement_amd import attention as A
from speech_enhancement_amd.weights import WeightPlan
B, T, Fq = 16, 321, 101
g = torch.Generator().manual_seed(0)
qkv = torch.randn(B * T * Fq, 192, generator=g).cuda()
E = (torch.randn(1025, 16, generator=g) * 0.5).cuda()
dO = (torch.randn(B * T * Fq, 64, generator=g) * 1e-3).cuda()
plan = WeightPlan(torch.device('cuda')); Es = plan.linear('e', E, planes='f16'); plan.run()
am = qkv.abs().max().reshape(1).clone()
dam = dO.abs().max().reshape(1).clone()
for axis in (sys.argv[1:] or ['time', 'freq']):
    geom = A.seq_geometry(B, T, Fq, axis)
    O, lse = A.attn_fwd(qkv, E, geom, Es=Es, qkv_amax=am)
    dE = torch.zeros_like(E)
    f = lambda: A.attn_bwd(qkv, E, O, dO, lse, geom, dE, qkv_amax=am, do_amax=dam)
    for _ in range(3): f()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(11)]
    ev[0].record()
    for i in range(10):
        f(); ev[i + 1].record()
    torch.cuda.synchronize()
    ts = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(10))
    n, nseq = geom[1], geom[0]
    fl = nseq * 4 * 7 * 2.0 * n * n * 16
    print(f'{axis}: bwd family median {ts[5]:.3f} ms min {ts[0]:.3f} ms  ({fl / ts[5] / 1e9:.1f} TFLOP/s algorithmic)  env BWD4={os.environ.get("SE_ATTN_BWD4")} NW={os.environ.get("SE_ATTN_BWD4_NW")} DBG={os.environ.get("SE_ATTN_DBG")}', flush=True)
