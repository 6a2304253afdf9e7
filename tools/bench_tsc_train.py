import sys, time, types, numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import speech_enhancement_amd as S
from speech_enhancement_amd import optim
NS = np.linspace(1e-4, 0.035, 50).tolist()
m = S.TSCNetDiffusion(64, 201, NS); m.apply(S.kaiming_init); m.cuda().train()
args = types.SimpleNamespace(optimizer='adamw', lr=5e-4, weight_decay=0.01, momentum=0.9, max_norm=0.0)
opt = optim.build_optimizer(args, m)
B = 16
clean = 0.1 * torch.randn(B, 32000, device='cuda'); noisy = clean + 0.05 * torch.randn_like(clean)
for _ in range(2): l = S.tsc_diffusion_step(m, opt, clean, noisy, NS)
torch.cuda.synchronize(); t0 = time.time()
for _ in range(4): l = S.tsc_diffusion_step(m, opt, clean, noisy, NS)
torch.cuda.synchronize(); dt = (time.time() - t0) / 4
print(f'TSC-diffusion train step, batch {B} x 2 s: {dt*1e3:.1f} ms/step = {B/dt:.1f} utt/s, loss {float(l):.4f}')
