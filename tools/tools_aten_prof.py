"""count the aten ops (PyTorch glue) in one train step"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import speech_enhancement_amd as S
from speech_enhancement_amd import train as TR, optim as OP
from torch.profiler import profile, ProfilerActivity
import types
torch.manual_seed(0)
B = 16
G, D = S.TSCNet(64, 201), S.Discriminator(16)
G.apply(S.kaiming_init); D.apply(S.kaiming_init)
G.cuda().train(); D.cuda().train()
oargs = types.SimpleNamespace(optimizer='adamw', lr=5e-4, weight_decay=0.01, momentum=0.9, max_norm=0.0)
og, od = OP.build_optimizer(oargs, G), OP.build_optimizer(oargs, D)
clean = torch.randn(B, 32000, device='cuda') * 0.1; noisy = clean + 0.05 * torch.randn_like(clean)
q = torch.rand(B, device='cuda')
labels = {'est': q, 'clean': torch.full_like(q, 0.97), 'noisy': q * 0.5}
w = (0.1, 0.9, 0.2, 0.05)
for _ in range(2): TR.gan_step(G, D, og, od, clean, noisy, 'cmgan', w, labels=labels)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=False, record_shapes=True) as prof:
    TR.gan_step(G, D, og, od, clean, noisy, 'cmgan', w, labels=labels)
    torch.cuda.synchronize()
rows = [e for e in prof.key_averages(group_by_input_shape=True) if e.key.startswith('aten::')]
rows.sort(key=lambda e: -e.count)
rows = [e for e in rows if e.device_time_total > 0 and e.key not in ('aten::zeros_like', 'aten::zero_', 'aten::clone', 'aten::contiguous', 'aten::zeros')]
print('launching aten ops per step:', sum(e.count for e in rows))
for e in rows[:90]:
    print(f'{e.key:28s} n={e.count:4d} cuda={e.device_time_total/1e3:7.3f} ms  {str(e.input_shapes)[:110]}')
