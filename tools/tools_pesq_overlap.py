"""PESQ side channel: step time with a slow label provider (sleep) vs labels supplied"""
import os, sys, time, types, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import speech_enhancement_amd as S
from speech_enhancement_amd import train as TR, optim as OP
torch.manual_seed(0)
B = 16
G, D = S.TSCNet(64, 201), S.Discriminator(16)
G.apply(S.kaiming_init); D.apply(S.kaiming_init)
G.cuda().train(); D.cuda().train()
oa = types.SimpleNamespace(optimizer='adamw', lr=5e-4, weight_decay=0.01, momentum=0.9, max_norm=0.0)
og, od = OP.build_optimizer(oa, G), OP.build_optimizer(oa, D)
clean = 0.1 * torch.randn(B, 32000, device='cuda'); noisy = clean + 0.05 * torch.randn_like(clean)
q = torch.rand(B, device='cuda')
w = (0.1, 0.9, 0.2, 0.05)
def run(labels, n=5):
    for _ in range(2): TR.gan_step(G, D, og, od, clean, noisy, 'cmgan', w, labels=labels)
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(n): TR.gan_step(G, D, og, od, clean, noisy, 'cmgan', w, labels=labels)
    torch.cuda.synchronize(); return (time.time() - t0) / n * 1e3
print('labels supplied: %.1f ms/step' % run({'est': q}))
for ms in (40, 80, 120):
    TR.set_pesq_provider(lambda c, d, ms=ms: (time.sleep(ms / 1e3), torch.rand(len(c)))[1])
    print('provider taking %d ms on the host, side channel: %.1f ms/step' % (ms, run(None)))
