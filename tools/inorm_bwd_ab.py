"""A/B of the InstanceNorm + PReLU backward: two-pass kernels vs the one-pass kernel (csrc/se_norms.hip), standalone on an idle GPU.
usage: python tools/inorm_bwd_ab.py [B P C]"""
import sys
import torch
sys.path.insert(0, '.')
from speech_enhancement_amd import ops as O, _lib as L  # noqa: E402

B, P, C = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (16, 32421, 64)
x = torch.randn(B, P, C, device='cuda')
dy = torch.randn(B, P, C, device='cuda')
g, be, a = torch.rand(C, device='cuda') + 0.5, torch.randn(C, device='cuda') * 0.1, torch.full((C,), 0.25, device='cuda')
stats = O.col_stats(x, C, 0, B, P, C)
mr, _ = O.norm_finalize(stats, g, be, B, C, float(P))
dx = torch.empty_like(x)
dg, db, da = (torch.zeros(C, device='cuda') for _ in range(3))
for fused, spin in ((False, 0), (True, 200), (True, 20), (True, 0)):
    O.NORM_BWD_FUSED[0], O.NORM_BWD_SPIN_US[0] = fused, spin
    ts = []
    for it in range(12):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        O.norm_prelu_bwd(x, C, 0, mr, g, be, a, dy, C, 0, dx, C, 0, dg, db, da, B, P, C, per_batch=True)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts = sorted(ts[2:])
    byt = 4.0 * B * P * C
    print(f'fused={fused} spin_us={spin}: median {ts[len(ts) // 2]:.1f} us  min {ts[0]:.1f} us   ({3 * byt / ts[len(ts) // 2] / 1e6:.2f} TB/s on 3 planes)')
