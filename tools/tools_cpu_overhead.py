"""host-side enqueue time of one train step vs its GPU time"""
import os, sys, time, types, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import speech_enhancement_amd as S
from speech_enhancement_amd import train as TR, optim as OP, _lib
torch.manual_seed(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
G, D = S.TSCNet(64, 201), S.Discriminator(16)
G.apply(S.kaiming_init); D.apply(S.kaiming_init)
G.cuda().train(); D.cuda().train()
oargs = types.SimpleNamespace(optimizer='adamw', lr=5e-4, weight_decay=0.01, momentum=0.9, max_norm=0.0)
og, od = OP.build_optimizer(oargs, G), OP.build_optimizer(oargs, D)
clean = torch.randn(B, 32000, device='cuda') * 0.1; noisy = clean + 0.05 * torch.randn_like(clean)
q = torch.rand(B, device='cuda')
labels = {'est': q, 'clean': torch.full_like(q, 0.97), 'noisy': q * 0.5}
w = (0.1, 0.9, 0.2, 0.05)
for _ in range(3): TR.gan_step(G, D, og, od, clean, noisy, 'cmgan', w, labels=labels)
torch.cuda.synchronize()
enq = []; tot = []
for _ in range(5):
    t0 = time.time()
    TR.gan_step(G, D, og, od, clean, noisy, 'cmgan', w, labels=labels)
    t1 = time.time()
    torch.cuda.synchronize()
    t2 = time.time()
    enq.append(t1 - t0); tot.append(t2 - t0)
print(f'B={B}: enqueue {1e3*sum(enq)/5:.1f} ms, total {1e3*sum(tot)/5:.1f} ms')
