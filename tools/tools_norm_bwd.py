"""InstanceNorm+PReLU backward (se_norm_prelu_bwd) at the dense-block shapes: time per call and algorithmic GB/s."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import speech_enhancement_amd as S
from speech_enhancement_amd import ops as O

B, T = 6, 321
for Fq, ldy in ((101, 64), (101, 256), (201, 64)):
    P = T * Fq
    R = torch.randn(B, T, Fq, 64, device='cuda')
    dy = torch.randn(B, T, Fq, ldy, device='cuda')
    mr = torch.rand(B, 64, 2, device='cuda') + 0.5
    g, b, sl = torch.randn(64, device='cuda'), torch.randn(64, device='cuda'), torch.full((64,), 0.25, device='cuda')
    dg, db, ds = (torch.zeros(64, device='cuda') for _ in range(3))
    dR = torch.empty_like(R)
    f = lambda: O.norm_prelu_bwd(R, 64, 0, mr, g, b, sl, dy, ldy, ldy - 64, dR, 64, 0, dg, db, ds, B, P, 64, per_batch=True)
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(30): f()
    torch.cuda.synchronize(); dt = (time.time() - t0) / 30
    print(f'F={Fq} ldy={ldy}: {dt*1e6:.1f} us per call (both passes), {5 * R.numel() * 4 / dt / 1e9:.0f} GB/s algorithmic')
