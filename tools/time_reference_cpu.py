"""Times the IMPORTED reference's own CMGAN train step (core/function.py: train_gan, one iteration per call) on the host cores of
the BUILD CONTAINER at the bench protocol of BASELINE.md section 2: batch 2, 2 s @ 16 kHz clips, AdamW lr 5e-4, PESQ labels supplied
(batch_pesq stubbed: excluded from the timing), 3 warm-up + 5 timed steps, torch.set_num_threads(all cores).  The reference cannot
travel to the GPU box, so this figure is recorded here (profiles/r05_reference_cpu_build_container.json) and quoted by bench.py as
cpu_baseline.reference_in_build_container.  Reuses the import stubs of tests/golden/make_golden.py.
usage: python tools/time_reference_cpu.py [warmup=3] [steps=5]"""
import json
import os
import sys
import time
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
sys.path.insert(0, ROOT)
import make_golden as MG        # noqa: E402  (installs the stubs, imports the reference)
import make_golden_v2 as M2     # noqa: E402
import formula                  # noqa: E402
from bench import synth_batch   # noqa: E402

RF = MG.RF
warm, steps = (int(sys.argv[1]) if len(sys.argv) > 1 else 3), (int(sys.argv[2]) if len(sys.argv) > 2 else 5)
ncpu = os.cpu_count() or 1
torch.set_num_threads(ncpu)
torch.cuda.synchronize = lambda *a, **k: None
torch.cuda.max_memory_allocated = lambda *a, **k: 0
tdt = M2._setup('f32')
g, d = M2._models(formula.formula_state('generator'), formula.formula_state('discriminator'), tdt)
args = types.SimpleNamespace(debug=False, gpu=None, arch='cmgan', epochs=100, gen_first=False, max_norm=0.0, print_freq=1000,
                             comp_type='pow', optimizer='adamw', lr=5e-4, weight_decay=0.01, momentum=0.9)
og, od = MG.build_optimizer(args, g), MG.build_optimizer(args, d)
clean, noisy, q = synth_batch(2, 32000, 1, 'cpu')
RF.batch_pesq = lambda c, n: q.clone()
loader = [{'audio': clean, 'noisy': noisy}]
times = []
for i in range(warm + steps):
    t0 = time.time()
    RF.train_gan(loader, g, d, M2.Crit([]), og, od, M2._Log(), 10, args, M2._cfg([0.1, 0.9, 0.2, 0.05], 5e-4))
    times.append(time.time() - t0)
    print(f'step {i}: {times[-1]:.2f} s', flush=True)
t = times[warm:]
res = {'what': "the imported reference's own train_gan (core/function.py:182-330), one iteration per call: cmgan, batch 2, 2 s clips, "
               'AdamW lr 5e-4, PESQ labels supplied (batch_pesq stubbed), fp32',
       'where': 'build container (no GPU)', 'cores': ncpu, 'threads': ncpu, 'warmup': warm, 'steps': steps,
       'step_seconds': [round(x, 2) for x in t], 'mean_step_seconds': round(sum(t) / len(t), 2),
       'value': round(2.0 * len(t) / sum(t), 4), 'unit': 'utterances/sec', 'torch': torch.__version__}
json.dump(res, open(os.path.join(ROOT, 'profiles', 'r05_reference_cpu_build_container.json'), 'w'), indent=1)
print(json.dumps(res))
