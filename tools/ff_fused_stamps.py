"""Where the cycles of se_ff_bwd_fused go (diagnostic build -DSE_FF_STAMPS: tools/micro/bin/libse_stamps.so, built by
tools/build_stamps_lib.sh): per wave, the cycles of work (barrier release -> arrival at the next barrier) and of waiting at each of the two
barriers of a 32-row tile, accumulated in scalar registers and written once at the end; and the whole-kernel cycles of every workgroup.
usage: SE_HIP_LIB=tools/micro/bin/libse_stamps.so python tools/ff_fused_stamps.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from speech_enhancement_amd import gemm as GM, ops as O, _lib as L
from speech_enhancement_amd.weights import WeightPlan

dev = torch.device('cuda')
M = 16 * 321 * 101
torch.manual_seed(0)
x = torch.randn(M, 64, device=dev); st = O.row_stats(x, M)
g, b = torch.rand(64, device=dev) + 0.5, torch.randn(64, device=dev) * 0.1
W1, b1 = torch.randn(256, 64, device=dev) * 0.1, torch.randn(256, device=dev) * 0.1
W2 = torch.randn(64, 256, device=dev) * 0.1
p = WeightPlan(dev); p.linear('w1', W1, planes='f16'); p.linear_T('w2t', W2, planes='f16', scale=0.5); p.linear_T('w1t', W1, planes='f16'); p.run()
dy = torch.randn(M, 64, device=dev) * 1e-3; dy._se_amax = dy.abs().max().reshape(1).clone()
dR2 = torch.randn(M, 64, device=dev) * 1e-3
gr = [torch.zeros(s, device=dev) for s in ((256, 64), (256,), (64, 256), (64,), (64,), (64,))]
stamps = torch.zeros(4096 + 1024, device=dev, dtype=torch.int32)
L.lib().se_ff_fused_debug_stamps(C.c_void_p(stamps.data_ptr()))
for _ in range(3):
    GM.ff_bwd_fused(dy, x, st, g, b, p.out['w1'], b1, p.out['w2t'], *gr, 0.2, 11, 12, 0.5, dR2=dR2, out_amax=torch.zeros(1, device=dev))
torch.cuda.synchronize()
r = stamps.cpu().numpy().astype(np.uint32)
w = r[:4 * 8 * 16].reshape(4, 8, 16).astype(np.float64)
names = ['rows requested, keep bits, H, dP, S, dZ, dW2, dW1 -> barrier Q', 'dLN -> patches, next rows -> images -> barrier R (+ the LayerNorm backward before it)']
for wg in range(2):
    nt = w[wg, 0, 8]
    print(f'workgroup {wg}: {int(nt)} tiles; cycles per tile, waves 0-7')
    for k in range(2):
        print(f'  work before barrier {"QR"[k]}: {np.round(w[wg, :, k] / nt).astype(int).tolist()}   wait: {np.round(w[wg, :, 4 + k] / nt).astype(int).tolist()}')
    print('  per tile:', int(round(w[wg, 0, :8].sum() / nt)))
tot, lp = r[4096:4096 + 254].astype(np.float64), r[4096 + 512:4096 + 512 + 254].astype(np.float64)
print('all workgroups, wave 0: kernel cycles min / median / max', int(tot.min()), int(np.median(tot)), int(tot.max()), '| up to the last barrier of the tile loop',
      int(lp.min()), int(np.median(lp)), int(lp.max()))
