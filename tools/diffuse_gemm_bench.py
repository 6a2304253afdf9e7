"""the two GEMMs of a CDiffuSE residual layer at the benchmark size (batch 32 x 32 000 samples, 64 -> 128 channels), alone:
python tools/diffuse_gemm_bench.py   (SE_GEMM_NO_PANEL=1: the generic tap kernel)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speech_enhancement_amd import gemm as GM, _lib as L  # noqa: E402
from speech_enhancement_amd.weights import WeightPlan  # noqa: E402

B, Lp, C = 32, 32000, 64
torch.manual_seed(0)
y = torch.randn(B, Lp, C, device='cuda')
y._se_amax = y.abs().max().reshape(1).clone()
Wd, bd = torch.randn(2 * C, C, 1, 3, device='cuda') * 0.08, torch.randn(2 * C, device='cuda') * 0.1
W2 = torch.randn(2 * C, C, device='cuda') * 0.1
plan = WeightPlan(torch.device('cuda'))
pd, p2 = plan.conv_fwd('d', Wd, planes='f16'), plan.linear('2', W2, planes='f16')
plan.run()
R = torch.empty(B, Lp, 2 * C, device='cuda')
st = torch.zeros(B, 2 * C, 2, device='cuda', dtype=torch.float64)


def bench(name, f, n=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f'{name:46s} {dt * 1e6:8.1f} us   {(B * Lp * (C + 2 * C) * 4) / dt / 1e12:5.2f} TB/s algorithmic')


for dil in (1, 8, 512):
    taps = [(0, -dil), (0, 0), (0, dil)]
    for ep, tag in ((L.EPI_BIAS | L.EPI_STATS, 'bias+stats'), (L.EPI_BIAS, 'bias')):
        d = GM.make_desc(B, 1, Lp, 1, Lp, taps, C, C, 2 * C, 2 * C, epilogue=ep, precision=3, a_amax=y._se_amax)
        bench(f'conv k3 dilation {dil} {tag}', lambda: GM.gemm_tap(d, y, pd, R, bias=bd, stats=st if ep & L.EPI_STATS else None))
for ep, tag in ((L.EPI_BIAS | L.EPI_STATS, 'bias+stats'), (L.EPI_BIAS, 'bias')):
    d2 = GM.make_desc(B, 1, Lp, 1, Lp, [(0, 0)], C, C, 2 * C, 2 * C, epilogue=ep, precision=3, a_sexp=13)
    bench(f'projection 64 -> 128 {tag}', lambda: GM.gemm_tap(d2, y, p2, R, bias=bd, stats=st if ep & L.EPI_STATS else None))
d3 = GM.linear_desc(B * Lp, C, 2 * C, epilogue=L.EPI_BIAS, precision=3, a_sexp=13)
bench('projection as one row GEMM (B = 1)', lambda: GM.gemm_tap(d3, y.view(-1, C), p2, R.view(-1, 2 * C), bias=bd))

# the generator's dilated dense convolution (B = 16, 321 x 201, Cin = 64 / 256 -> 64) with and without the InstanceNorm sums
from speech_enhancement_amd import layers as LY  # noqa: E402
Bg, T, Fq = 16, 321, 201
for Cin in (64, 256):
    x = torch.randn(Bg, T, Fq, 256, device='cuda')
    Wc = torch.randn(64, Cin, 2, 3, device='cuda') * 0.05
    plan2 = WeightPlan(torch.device('cuda'))
    pc = plan2.conv_fwd('c', Wc, rev=True, planes='f16')
    plan2.run()
    Rg = torch.empty(Bg, T, Fq, 64, device='cuda')
    stg = torch.zeros(Bg, 64, 2, device='cuda', dtype=torch.float64)
    for ep, tag in ((L.EPI_BIAS | L.EPI_STATS, 'bias+stats'), (L.EPI_BIAS, 'bias')):
        dg = GM.make_desc(Bg, T, Fq, T, Fq, LY.dense_taps(0), Cin, 256, 64, 64, epilogue=ep, precision=3, a_sexp=4)
        bench(f'generator conv3 Cin {Cin} {tag}', lambda: GM.gemm_tap(dg, x, pc, Rg, bias=bd[:64].contiguous(), stats=stg if ep & L.EPI_STATS else None))
