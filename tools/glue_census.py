"""Which Python lines of a train step launch PyTorch's own small kernels (fill / copy / add / cat ...)?  a TorchDispatchMode
over ONE step of bench.py's workload; prints aten ops grouped by the innermost repository frame."""
import collections
import os
import sys
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import speech_enhancement_amd as S  # noqa: E402
from speech_enhancement_amd import optim, train as TR  # noqa: E402

dev = torch.device('cuda', 0)
torch.manual_seed(0)
G, D = S.TSCNet(64, 201), S.Discriminator(16)
G.apply(S.kaiming_init), D.apply(S.kaiming_init)
G.to(dev).train(), D.to(dev).train()
oargs = types.SimpleNamespace(optimizer='adamw', lr=5e-4, weight_decay=0.01, momentum=0.9, max_norm=0.0)
og, od = optim.build_optimizer(oargs, G), optim.build_optimizer(oargs, D)
clean, noisy, q = bench.synth_batch(16, 32000, 1, dev)
labels = {'est': q, 'clean': torch.full_like(q, 0.97), 'noisy': q * 0.5}
step = lambda: TR.gan_step(G, D, og, od, clean, noisy, 'cmgan', (0.1, 0.9, 0.2, 0.05), labels=labels)
for _ in range(3):
    step()
torch.cuda.synchronize()
import traceback
from torch.utils._python_dispatch import TorchDispatchMode

cnt = collections.Counter()
SKIP = ('aten.view', 'aten.detach', 'aten.t.', 'aten.transpose', 'aten.slice', 'aten.select', 'aten.reshape', 'aten._unsafe_view', 'aten.alias',
        'aten.unsqueeze', 'aten.squeeze', 'aten.expand', 'aten.permute', 'aten.as_strided', 'aten.empty', 'aten.is_', 'aten.sym_', 'aten.lift',
        'aten._local_scalar_dense', 'aten.unbind', 'aten.split', 'aten.narrow', 'aten.item', 'aten.stride', 'aten.size', 'aten.numel')


class Census(TorchDispatchMode):
    def __torch_dispatch__(self, func, types_, args=(), kwargs=None):
        name = str(func)
        if not name.startswith(SKIP):
            fr = [f for f in traceback.extract_stack() if ROOT in f.filename and 'glue_census' not in f.filename]
            where = f'{fr[-1].filename.replace(ROOT + "/", "")}:{fr[-1].lineno}' if fr else '(autograd engine / no repository frame)'
            cnt[(name, where)] += 1
        return func(*args, **(kwargs or {}))


with Census():
    step()
    torch.cuda.synchronize()
for (name, where), n in cnt.most_common(60):
    print(f'{n:4d}  {name:28s} {where}')
print('total', sum(cnt.values()))
