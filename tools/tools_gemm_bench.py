"""micro-benchmark of the tap-GEMM at the Conformer linear-layer shapes (M = 16*321*101 tokens)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speech_enhancement_amd import gemm as GM, _lib as L
M = 16 * 321 * 101


PREC = int(os.environ.get('PREC', '0'))


def run(name, C, N, pro=0, epi=0, aux=False, bias=False, resid=False, ldc=None, reps=5):
    x = torch.randn(M, C, device='cuda')
    w = torch.randn(N, C, device='cuda') * C ** -0.5
    No = N // 2 if epi & L.EPI_GLU else N
    y = torch.empty(M, ldc or No, device='cuda')
    kw = {}
    if aux:
        kw['AUX'] = torch.randn(M, N, device='cuda'); ldx = N
    else:
        ldx = 0
    if bias:
        kw['bias'] = torch.randn(N, device='cuda')
    if resid:
        kw['R'] = torch.randn(M, N, device='cuda')
    if pro == L.PRO_LN:
        kw['rowstats'] = torch.stack([x.mean(-1), x.var(-1).rsqrt()], -1).contiguous()
        kw['ps'] = torch.ones(C, device='cuda'); kw['pb'] = torch.zeros(C, device='cuda')
    d = GM.linear_desc(M, C, N, ldc=ldc or No, prologue=pro, epilogue=epi, ldx=ldx, ldr=N if resid else 0, drop_p=0.2 if (epi & L.EPI_DROP or pro in (4, 5)) else 0.0,
                       pro_seed=123, epi_seed=456, precision=PREC)
    for _ in range(2):
        GM.gemm_tap(d, x, w, y, **kw)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(reps):
        GM.gemm_tap(d, x, w, y, **kw)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / reps
    gb = 4.0 * M * (C + No + (N if aux else 0) + (N if resid else 0)) / 1e9
    print(f'{name:44s} {dt*1e6:8.1f} us  {2.0*M*C*N/dt/1e12:6.1f} TF  {gb/dt:7.0f} GB/s (min traffic {gb:.2f} GB)', flush=True)


run('K64->N256 plain', 64, 256)
run('K64->N256 +bias', 64, 256, epi=L.EPI_BIAS, bias=True)
run('K64->N256 LN pro + bias', 64, 256, pro=L.PRO_LN, epi=L.EPI_BIAS, bias=True)
run('K64->N256 swishgrad (AUX)', 64, 256, epi=L.EPI_SWISH_GRAD, aux=True)
run('K64->N256 drop pro + swishgrad + drop', 64, 256, pro=L.PRO_DROP, epi=L.EPI_SWISH_GRAD | L.EPI_DROP, aux=True)
run('K64->N256 GLU + Z', 64, 256, pro=L.PRO_LN, epi=L.EPI_BIAS | L.EPI_GLU, aux=True, bias=True)
run('K64->N64 plain', 64, 64)
run('K64->N64 resid+bias', 64, 64, epi=L.EPI_BIAS | L.EPI_RESID, bias=True, resid=True)
run('K64->N192 LN', 64, 192, pro=L.PRO_LN)
run('K256->N64 plain', 256, 64)
run('K256->N64 swish pro + resid', 256, 64, pro=L.PRO_SWISH, epi=L.EPI_BIAS | L.EPI_RESID, bias=True, resid=True)
run('K128->N64 plain', 128, 64)
run('K192->N64 plain', 192, 64)
