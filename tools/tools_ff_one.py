"""fused feed-forward forward / input-gradient kernels at the benchmark shape (M = 16 * 321 * 101 tokens, hidden 256, pre-split weights)"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speech_enhancement_amd import gemm as GM, ops as O
from speech_enhancement_amd.weights import WeightPlan
M = 16 * 321 * 101
torch.manual_seed(0)
x = torch.randn(M, 64, device='cuda'); st = O.row_stats(x, M)
g, b = torch.rand(64, device='cuda') + 0.5, torch.randn(64, device='cuda') * 0.1
W1, b1 = torch.randn(256, 64, device='cuda') * 0.1, torch.randn(256, device='cuda') * 0.1
W2, b2 = torch.randn(64, 256, device='cuda') * 0.05, torch.randn(64, device='cuda') * 0.1
plan = WeightPlan(torch.device('cuda'))
W1p, W2p = plan.linear('w1', W1, planes=True), plan.linear('w2', W2, planes=True)
W2T, W1T = plan.linear_T('w2t', W2, planes=True, scale=0.5), plan.linear_T('w1t', W1, planes=True)
plan.run()
dy = torch.randn(M, 64, device='cuda')
def bench(f, n=10):
    for _ in range(2): f()
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.time() - t0) / n * 1e6
t = bench(lambda: GM.ff_fwd(x, st, g, b, W1p, b1, W2p, b2, 0.2, 11, 12, 0.5, precision=2, hid=256, out_stats=True))
print(f'ff_fwd  {t:7.1f} us')
y, h, _ = GM.ff_fwd(x, st, g, b, W1p, b1, W2p, b2, 0.2, 11, 12, 0.5, precision=2, hid=256, out_stats=True)
dg, db = torch.zeros(64, device='cuda'), torch.zeros(64, device='cuda')
t = bench(lambda: GM.ff_bwd_dgrad(dy, h, W2T, W1T, 0.2, 11, 12, precision=2, ln=(x, st, g, None, dg, db)))
print(f'ff_bwd  {t:7.1f} us')
