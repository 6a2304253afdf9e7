"""How far ahead of the GPU is the host?  Per train step (bench.py's workload): wall time for `gan_step` to RETURN with an empty queue
in front of it (= the host's enqueue time: Python + ctypes + HIP launch calls) next to the GPU time of the step.
usage: python tools/host_enqueue.py [batch]"""
import os
import sys
import time
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import speech_enhancement_amd as S  # noqa: E402
from speech_enhancement_amd import optim, train as TR  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = torch.device('cuda', 0)
torch.manual_seed(0)
G, D = S.TSCNet(64, 201), S.Discriminator(16)
G.apply(S.kaiming_init), D.apply(S.kaiming_init)
G.to(dev).train(), D.to(dev).train()
oargs = types.SimpleNamespace(optimizer='adamw', lr=5e-4, weight_decay=0.01, momentum=0.9, max_norm=0.0)
og, od = optim.build_optimizer(oargs, G), optim.build_optimizer(oargs, D)
clean, noisy, q = bench.synth_batch(B, 32000, 1, dev)
labels = {'est': q, 'clean': torch.full_like(q, 0.97), 'noisy': q * 0.5}
step = lambda: TR.gan_step(G, D, og, od, clean, noisy, 'cmgan', (0.1, 0.9, 0.2, 0.05), labels=labels)
for _ in range(4):
    step()
torch.cuda.synchronize()
host, total = [], []
for _ in range(6):
    t0 = time.perf_counter()
    step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    host.append((t1 - t0) * 1e3), total.append((t2 - t0) * 1e3)
print('batch', B, 'host enqueue ms/step', [round(x, 1) for x in host], 'step ms (enqueue .. drained)', [round(x, 1) for x in total])
t0 = time.perf_counter()
for _ in range(10):
    step()
torch.cuda.synchronize()
print('back-to-back ms/step', round((time.perf_counter() - t0) * 100, 2))
