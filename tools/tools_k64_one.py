"""row-panel kernel (K = 64 -> N = 192 qkv / 256 pointwise-GLU, LayerNorm prologue, pre-split weights) at the benchmark size"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speech_enhancement_amd import gemm as GM, ops as O, _lib as L
from speech_enhancement_amd.weights import WeightPlan
M = 16 * 321 * 101
torch.manual_seed(0)
x = torch.randn(M, 64, device='cuda'); st = O.row_stats(x, M)
g, b = torch.rand(64, device='cuda') + 0.5, torch.randn(64, device='cuda') * 0.1
plan = WeightPlan(torch.device('cuda'))
Wq = plan.linear('q', torch.randn(192, 64, device='cuda') * 0.1, planes=True)
Wp = plan.linear('p', torch.randn(256, 64, device='cuda') * 0.1, planes=True)
bp = torch.randn(256, device='cuda') * 0.1
plan.run()
def bench(f, n=10):
    for _ in range(2): f()
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.time() - t0) / n * 1e6
qkv = torch.empty(M, 192, device='cuda')
print(f'LN -> 192 (qkv)       {bench(lambda: GM.gemm_tap(GM.linear_desc(M, 64, 192, prologue=L.PRO_LN), x, Wq, qkv, rowstats=st, ps=g, pb=b)):7.1f} us')
u = torch.empty(M, 128, device='cuda'); zc = torch.empty(M, 256, device='cuda')
print(f'LN -> 256 GLU (pw1)   {bench(lambda: GM.gemm_tap(GM.linear_desc(M, 64, 256, ldc=128, prologue=L.PRO_LN, epilogue=L.EPI_BIAS | L.EPI_GLU, ldx=256), x, Wp, u, bias=bp, AUX=zc, rowstats=st, ps=g, pb=b)):7.1f} us')
