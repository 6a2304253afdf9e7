"""one Conformer-shaped linear GEMM, for rocprofv3 --pmc runs: args C N [pro] [epi] [prec]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speech_enhancement_amd import gemm as GM, _lib as L
M = 16 * 321 * 101
C, N = int(sys.argv[1]), int(sys.argv[2])
pro = int(sys.argv[3]) if len(sys.argv) > 3 else 0
epi = int(sys.argv[4]) if len(sys.argv) > 4 else 0
prec = int(sys.argv[5]) if len(sys.argv) > 5 else 0
x = torch.randn(M, C, device='cuda'); w = torch.randn(N, C, device='cuda') * C ** -0.5
No = N // 2 if epi & L.EPI_GLU else N
y = torch.empty(M, No, device='cuda')
kw = {}
if pro == L.PRO_LN:
    kw['rowstats'] = torch.stack([x.mean(-1), x.var(-1).rsqrt()], -1).contiguous()
    kw['ps'] = torch.ones(C, device='cuda'); kw['pb'] = torch.zeros(C, device='cuda')
if epi & L.EPI_BIAS:
    kw['bias'] = torch.randn(N, device='cuda')
if epi & (L.EPI_SWISH_GRAD | L.EPI_GLU):
    kw['AUX'] = torch.randn(M, N, device='cuda')
if epi & L.EPI_RESID:
    kw['R'] = torch.randn(M, N, device='cuda')
d = GM.linear_desc(M, C, N, ldc=No, prologue=pro, epilogue=epi, ldx=N if 'AUX' in kw else 0, ldr=N if 'R' in kw else 0,
                   drop_p=0.2 if (epi & L.EPI_DROP or pro in (4, 5)) else 0.0, pro_seed=1, epi_seed=2, precision=prec)
for _ in range(3):
    GM.gemm_tap(d, x, w, y, **kw)
torch.cuda.synchronize()
