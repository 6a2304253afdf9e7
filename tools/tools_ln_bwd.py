"""se_gemm_ln_bwd (input-gradient GEMM + LayerNorm backward on the accumulators) vs the two-kernel form at bench size"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speech_enhancement_amd import gemm as GM, ops as O
from speech_enhancement_amd.weights import WeightPlan
M = 16 * 321 * 101
for K in (192, 256):
    x, dy, dR = torch.randn(M, 64, device='cuda'), torch.randn(M, K, device='cuda'), torch.randn(M, 64, device='cuda')
    W = torch.randn(K, 64, device='cuda') * 0.1
    gam = torch.rand(64, device='cuda') + 0.5
    st = O.row_stats(x, M)
    plan = WeightPlan(torch.device('cuda')); WT = plan.linear_T('wt', W, planes=True); plan.run()
    dg, db = torch.zeros(64, device='cuda'), torch.zeros(64, device='cuda')
    def fused(): return GM.gemm_ln_bwd(dy, WT, x, st, gam, dR, dg, db)
    def two():
        dl = torch.empty(M, 64, device='cuda')
        GM.gemm_tap(GM.linear_desc(M, K, 64, precision=2), dy, WT, dl)
        return O.layernorm_bwd(x, st, gam, dl, dg, db, dR=dR)
    for name, f in (('fused', fused), ('two kernels', two), ('fused', fused), ('two kernels', two)):
        for _ in range(2): f()
        torch.cuda.synchronize(); t0 = time.time()
        for _ in range(10): f()
        torch.cuda.synchronize(); print(f'K={K} {name:12s} {(time.time() - t0) / 10 * 1e6:7.1f} us', flush=True)
