"""token-wise (row-GEMM) weight gradients of a Conformer block: full-tile kernel (wgrad_lin_kernel) vs the per-block kernel"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speech_enhancement_amd import gemm as GM, _lib as L
M = 16 * 321 * 101
torch.manual_seed(0)
cases = [('LN x[M,64] -> dW[256,64] (ff W1, pw1)', 64, 256, L.PRO_LN, 0),
         ('LN x[M,64] -> dW[192,64] (qkv)', 64, 192, L.PRO_LN, 0),
         ('swish+drop H[M,256] -> dW[64,256] (ff W2)', 256, 64, L.PRO_SWISH_DROP, L.EPI_DROP),
         ('bn+swish h[M,128] -> dW[64,128] (pw2)', 128, 64, L.PRO_AFFINE_SWISH, 0)]
for name, Cin, N, pro, epi in cases:
    x = torch.randn(M, Cin, device='cuda'); dy = torch.randn(M, N, device='cuda')
    st = torch.stack([x.mean(-1), (x.var(-1, unbiased=False) + 1e-5).rsqrt()], -1).contiguous()
    g = torch.rand(Cin, device='cuda') + 0.5; b = torch.randn(Cin, device='cuda') * 0.1
    d = GM.linear_desc(M, Cin, N, prologue=pro, epilogue=epi, pro_seed=5, epi_seed=7, drop_p=0.2 if pro == L.PRO_SWISH_DROP else 0.0)
    res = {}
    for mode in ('blocks', 'full', 'full-x6'):
        if mode == 'blocks':
            os.environ['SE_WGRAD_NO_LIN'] = '1'
        else:
            os.environ.pop('SE_WGRAD_NO_LIN', None)
        d.precision = 2 if mode == 'full-x6' else 0
        dw = torch.zeros(N, Cin, device='cuda'); db = torch.zeros(N, device='cuda')
        f = lambda: GM.gemm_tap_wgrad(d, x, dy, dw, db, rowstats=st, ps=g, pb=b, explicit_precision=True)
        f(); torch.cuda.synchronize()
        res[mode] = (dw.clone(), db.clone())
        for _ in range(2): f()
        torch.cuda.synchronize(); t0 = time.time()
        for _ in range(10): f()
        torch.cuda.synchronize(); dt = (time.time() - t0) / 10
        print(f'{name:48s} {mode:7s} {dt*1e6:7.1f} us  {2.0*M*Cin*N/dt/1e12:6.1f} TF', flush=True)
    for m2 in ('full', 'full-x6'):
        e = float((res[m2][0] - res['blocks'][0]).abs().max() / res['blocks'][0].abs().max())
        eb = float((res[m2][1] - res['blocks'][1]).abs().max() / res['blocks'][1].abs().max())
        print(f'    max relative difference {m2} vs blocks: dW {e:.2e}, dbias {eb:.2e}')
