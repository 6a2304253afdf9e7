import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from speech_enhancement_amd import gemm as GM, _lib as L
M = 16 * 321 * 101
C, N = int(sys.argv[1]), int(sys.argv[2])
x = torch.randn(M, C, device='cuda'); w = torch.randn(N, C, device='cuda') * C ** -0.5; y = torch.empty(M, N, device='cuda')
d = GM.linear_desc(M, C, N)
for _ in range(3):
    GM.gemm_tap(d, x, w, y)
torch.cuda.synchronize()
