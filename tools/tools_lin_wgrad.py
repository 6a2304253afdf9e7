"""LN-prologue linear weight gradient (dW[256,64] = dZ^T LN(x)) chunk sweep"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speech_enhancement_amd import gemm as GM, _lib as L
M = 16 * 321 * 101
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
x = torch.randn(M, 64, device='cuda'); dz = torch.randn(M, N, device='cuda')
st = torch.stack([x.mean(-1), (x.var(-1, unbiased=False) + 1e-5).rsqrt()], -1).contiguous()
g = torch.ones(64, device='cuda'); b = torch.zeros(64, device='cuda')
d = GM.linear_desc(M, 64, N, prologue=L.PRO_LN)
for ch in (None, 128, 256, 512, 768, 1024, 2048):
    dw = torch.zeros(N, 64, device='cuda'); db = torch.zeros(N, device='cuda')
    f = lambda: GM.gemm_tap_wgrad(d, x, dz, dw, db, rowstats=st, ps=g, pb=b, chunks=ch)
    for _ in range(2): f()
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(5): f()
    torch.cuda.synchronize(); dt = (time.time() - t0) / 5
    print(f'N={N} chunks={ch}: {dt*1e6:.1f} us  {2.0*M*64*N/dt/1e12:.1f} TF')
