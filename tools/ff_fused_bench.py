"""se_ff_bwd_fused at the bench shape (M = 16 x 321 x 101 rows) alone: time per launch with and without dropout (median of three
groups of 20 launches).  SE_HIP_LIB selects a variant library (tools/build_variant_lib.sh) for a same-box A/B.
usage: python tools/ff_fused_bench.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speech_enhancement_amd import gemm as GM, ops as O
from speech_enhancement_amd.weights import WeightPlan

dev = torch.device('cuda')
M = int(os.environ.get('M', 16 * 321 * 101))
torch.manual_seed(0)
x = torch.randn(M, 64, device=dev)
st = O.row_stats(x, M)
g, b = torch.rand(64, device=dev) + 0.5, torch.randn(64, device=dev) * 0.1
W1, b1 = torch.randn(256, 64, device=dev) * 0.1, torch.randn(256, device=dev) * 0.1
W2 = torch.randn(64, 256, device=dev) * 0.1
p = WeightPlan(dev)
p.linear('w1', W1, planes='f16'); p.linear_T('w2t', W2, planes='f16', scale=0.5); p.linear_T('w1t', W1, planes='f16')
p.run()
dy = torch.randn(M, 64, device=dev) * 1e-3
dy._se_amax = dy.abs().max().reshape(1).clone()
dR2 = torch.randn(M, 64, device=dev) * 1e-3
gr = [torch.zeros(s, device=dev) for s in ((256, 64), (256,), (64, 256), (64,), (64,), (64,))]
am = torch.zeros(1, device=dev)


def run(n=20, drop=0.2):
    ts = []
    for _ in range(3):
        for _ in range(2):
            GM.ff_bwd_fused(dy, x, st, g, b, p.out['w1'], b1, p.out['w2t'], *gr, drop, 11, 12, 0.5, dR2=dR2, out_amax=am)
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(n):
            GM.ff_bwd_fused(dy, x, st, g, b, p.out['w1'], b1, p.out['w2t'], *gr, drop, 11, 12, 0.5, dR2=dR2, out_amax=am)
        torch.cuda.synchronize()
        ts.append((time.time() - t0) / n * 1e6)
    return sorted(ts)[1]


print(f'se_ff_bwd_fused: {run():8.1f} us per launch (drop 0.2), {run(drop=0.0):8.1f} (no dropout)', flush=True)
