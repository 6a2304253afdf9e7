"""Barrier-arrival timeline of se_ff_bwd_fused (diagnostic build -DSE_FF_STAMPS: tools/micro/bin/libse_stamps.so, built by
tools/build_stamps_lib.sh): for the first 4 workgroups and their first 8 tiles, when (shader clock) every wave ARRIVED at each of
the 10 barriers of a tile.  Prints, per barrier interval, the arrival of the D waves (0-3) and of the W waves (4-7) relative to the
previous barrier's release (= the last arrival): who works how long in which interval, who waits for whom.
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
s = stamps.cpu().numpy().astype(np.uint32)[:4 * 8 * 8 * 10].reshape(4, 8, 8, 10).astype(np.int64)
if os.environ.get('SE_FF_FUSED_V') == '2':
    names = ['p', 'a0', 'b0', 'a1', 'b1', 'a2', 'b2', 'a3', 'b3', 'c']
    for wg in range(2):
        print(f'workgroup {wg}: arrival at each barrier relative to the previous release (cycles); D = waves 0-3, W = waves 4-7')
        rel_prev = None
        for t in range(1, 6):
            for k in range(10):
                arr = s[wg, :, t, k]
                if rel_prev is not None:
                    d = (arr - rel_prev) & 0xffffffff
                    print(f'  tile {t} ({names[k]:>2}): D {d[:4].tolist()}  W {d[4:].tolist()}   interval {int(d.max())}')
                rel_prev = int(arr.max())
        tile_t = [int(s[wg, :, t + 1, 0].max() - s[wg, :, t, 0].max()) for t in range(1, 6)]
        print('  cycles per tile:', tile_t)
else:
    # ping-pong kernel: per wave, cycles of work (barrier release -> arrival at the next) and of waiting per slot, averaged per tile
    r = stamps.cpu().numpy().astype(np.uint32)[:4 * 8 * 16].reshape(4, 8, 16).astype(np.float64)
    for wg in range(2):
        nt = r[wg, 0, 8]
        print(f'workgroup {wg}: {int(nt)} tiles; cycles per tile and slot, waves 0-3 | waves 4-7')
        for k in range(4):
            wk, wt = r[wg, :, k] / nt, r[wg, :, 4 + k] / nt
            print(f'  slot {k + 1} work {np.round(wk[:4]).astype(int).tolist()} | {np.round(wk[4:]).astype(int).tolist()}   wait {np.round(wt[:4]).astype(int).tolist()} | {np.round(wt[4:]).astype(int).tolist()}')
        print('  per tile:', int(round((r[wg, 0, :8].sum()) / nt)))
    tot = stamps.cpu().numpy().astype(np.uint32)[4096:4096 + 254].astype(np.float64); lp = stamps.cpu().numpy().astype(np.uint32)[4096 + 512:4096 + 512 + 254].astype(np.float64)
    print('all workgroups, wave 0: kernel cycles min / median / max', int(tot.min()), int(np.median(tot)), int(tot.max()), '| up to the end of the tile loop', int(lp.min()), int(np.median(lp)), int(lp.max()))
    print('  by XCD (workgroup % 8): median kernel cycles', [int(np.median(tot[i::8])) for i in range(8)])
