"""Which Python call sites issue the small torch launches of a train step (copies, fills, adds)?  torch.profiler with stacks over
ONE step after warm-up; prints, per aten op and source line inside the package, the number of launches.
usage: python tools/glue_census.py [arch]"""
import collections, os, sys, types
import torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import bench as B            # noqa: E402  (synth_batch)
import speech_enhancement_amd as S    # noqa: E402
from speech_enhancement_amd import optim, train as TR    # noqa: E402

arch = sys.argv[1] if len(sys.argv) > 1 else 'cmgan'
dev = torch.device('cuda', 0)
torch.manual_seed(0)
G, D = S.TSCNet(64, 201), S.Discriminator(16)
G.apply(S.kaiming_init); D.apply(S.kaiming_init)
G.to(dev).train(); D.to(dev).train()
oargs = types.SimpleNamespace(optimizer='adamw', lr=5e-4, weight_decay=0.01, momentum=0.9, max_norm=0.0)
og, od = optim.build_optimizer(oargs, G), optim.build_optimizer(oargs, D)
clean, noisy, q = B.synth_batch(16, 32000, 1, dev)
labels = {'est': q, 'clean': torch.full_like(q, 0.97), 'noisy': q * 0.5}
step = lambda: TR.gan_step(G, D, og, od, clean, noisy, arch, (0.1, 0.9, 0.2, 0.05), labels=labels)
for _ in range(3):
    step()
torch.cuda.synchronize()
import traceback
cnt = collections.Counter()
def site():
    for fr in reversed(traceback.extract_stack()[:-2]):
        if 'speech' in fr.filename and 'amd' in fr.filename:
            return f'{os.path.basename(fr.filename)}:{fr.lineno} {fr.line[:80]}'
    return '?'
def wrap(obj, name, tag=None, pred=None):
    f = getattr(obj, name)
    def g(*a, **k):
        if pred is None or pred(a, k):
            cnt[(tag or name, site())] += 1
        return f(*a, **k)
    setattr(obj, name, g)
T = torch.Tensor
for n in ('to', 'cuda', 'fill_', 'zero_', 'add_', '__iadd__', '__add__', '__radd__', 'copy_', 'clone', '__mul__', '__rmul__', 'mul', 'mul_', '__imul__',
          '__truediv__', '__sub__', 'sum', 'mean', 'pow', '__pow__', 'float', 'double'):
    wrap(T, n)
wrap(T, 'contiguous', pred=lambda a, k: not a[0].is_contiguous())
for n in ('tensor', 'as_tensor', 'zeros', 'ones', 'full', 'cat', 'stack', 'zeros_like', 'full_like', 'ones_like', 'empty'):
    wrap(torch, n)
step()
torch.cuda.synchronize()
if os.environ.get('GLUE_PROFILER') == '1':
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        step()
        torch.cuda.synchronize()
    print(prof.key_averages(group_by_input_shape=True).table(sort_by='count', row_limit=60, max_name_column_width=50, max_shapes_column_width=60))
    sys.exit(0)
tot = sum(v for (n, _), v in cnt.items() if n != 'empty')
print('wrapped torch calls in one step (without empty):', tot)
for (name, st), n in cnt.most_common(90):
    if name != 'empty':
        print(f'{n:4d}  {name:12s} {st}')
