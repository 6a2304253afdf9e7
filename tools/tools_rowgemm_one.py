"""fp32-MFMA row GEMMs with a residual epilogue at the benchmark size: attention out-projection (64 -> 64, bias + dropout +
residual + row statistics) and pointwise conv 2 (BatchNorm-affine + Swish prologue, 128 -> 64, bias + residual + row statistics)"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speech_enhancement_amd import gemm as GM, _lib as L
M = 16 * 321 * 101
torch.manual_seed(0)
def bench(f, n=10):
    for _ in range(2): f()
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.time() - t0) / n * 1e6
o, y1 = torch.randn(M, 64, device='cuda'), torch.randn(M, 64, device='cuda')
Wo, bo = torch.randn(64, 64, device='cuda') * 0.1, torch.randn(64, device='cuda') * 0.1
y2, st = torch.empty(M, 64, device='cuda'), torch.empty(M, 2, device='cuda')
f1 = lambda: GM.gemm_tap(GM.linear_desc(M, 64, 64, epilogue=L.EPI_BIAS | L.EPI_RESID | L.EPI_DROP | L.EPI_ROWSTATS, alpha=1.0, ldr=64,
                                        epi_seed=5, drop_p=0.1), o, Wo, y2, bias=bo, R=y1, AUX=st)
print(f'to_out 64 -> 64 (+resid, drop, rowstats)  {bench(f1):7.1f} us')
h = torch.randn(M, 128, device='cuda'); W2, b2 = torch.randn(64, 128, device='cuda') * 0.1, torch.randn(64, device='cuda') * 0.1
sc, sh = torch.rand(128, device='cuda') + 0.5, torch.randn(128, device='cuda') * 0.1
f2 = lambda: GM.gemm_tap(GM.linear_desc(M, 128, 64, prologue=L.PRO_AFFINE_SWISH, epilogue=L.EPI_BIAS | L.EPI_RESID | L.EPI_ROWSTATS,
                                        alpha=1.0, ldr=64), h, W2, y2, bias=b2, R=y1, ps=sc, pb=sh, AUX=st)
print(f'pw2 128 -> 64 (bn+swish, +resid, rowstats) {bench(f2):7.1f} us')
dy = torch.randn(M, 64, device='cuda'); do = torch.empty(M, 64, device='cuda')
f3 = lambda: GM.gemm_tap(GM.linear_desc(M, 64, 64, prologue=L.PRO_DROP, pro_seed=5, drop_p=0.1), dy, Wo, do)
print(f'to_out dgrad 64 -> 64 (drop prologue)      {bench(f3):7.1f} us')
