"""Where the time of a streamed train step goes: every C-ABI launch of ONE step timed with HIP events on its own stream
(start = the stream reached the launch, end = the kernel finished), once in the default three-stream order and once in serial
order.  Prints per stream: launches, summed launch time, first / last time stamp; and per kernel family on the main stream the
summed duration in both orders (the slowdown a family suffers from sharing the GPU with the other streams)."""
import collections, os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import speech_enhancement_amd as S
from speech_enhancement_amd import train as TR, optim as OP, _lib, gemm as GM

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
step = lambda: TR.gan_step(G, D, og, od, clean, noisy, 'cmgan', w, labels=labels)


def run(streams):
    saved = (GM._LeafStream.enabled, TR._D_OVERLAP, GM.branch_stream.enabled)
    GM._LeafStream.enabled, TR._D_OVERLAP, GM.branch_stream.enabled = streams, streams, streams
    try:
        for _ in range(3): step()
        torch.cuda.synchronize()
        base, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        _lib.TIMER.start(every_call=True)
        base.record()
        step()
        end.record()
        torch.cuda.synchronize()
        _lib.TIMER.stop()
        recs = [(k, base.elapsed_time(e0), base.elapsed_time(e1), st) for k, _, _, e0, e1, st in _lib.TIMER.records]
        return recs, base.elapsed_time(end)
    finally:
        GM._LeafStream.enabled, TR._D_OVERLAP, GM.branch_stream.enabled = saved


main = torch.cuda.current_stream().cuda_stream
res = {}
for mode in (False, True, False, True):
    recs, total = run(mode)
    res[mode] = (recs, total)
    by = collections.defaultdict(list)
    for k, t0, t1, st in recs: by[st].append((t0, t1, k))
    print(f'== streams {"on" if mode else "off"}: step {total:.2f} ms (with {len(recs)} timed launches)')
    for st, v in sorted(by.items(), key=lambda kv: -len(kv[1])):
        print(f'   stream {"main" if st == main else hex(st)}: {len(v)} launches, sum of launch times {sum(b - a for a, b, _ in v):.2f} ms, '
              f'first {min(a for a, _, _ in v):.2f} ms, last end {max(b for _, b, _ in v):.2f} ms')
fam = {}
for mode in (False, True):
    agg = collections.Counter()
    for k, t0, t1, st in res[mode][0]:
        if st == main: agg[k] += t1 - t0
    fam[mode] = agg
print('main-stream families: ms serial -> ms with streams (ratio)')
for k, v in fam[True].most_common(40):
    s0 = fam[False].get(k, 0.0)
    print(f'   {k[:60]:60s} {s0:7.2f} -> {v:7.2f}  ({v / s0 if s0 else float("nan"):.2f}x)')
print(f'   total on main: {sum(fam[False].values()):.2f} -> {sum(fam[True].values()):.2f}')
# the phases of the streamed step: forward ends at the first backward-only launch
recs = sorted(res[True][0], key=lambda r: r[1])
marks = {}
for k, t0, t1, st in recs:
    for name in ('se_mse', 'se_attn_bwd', 'se_adamw', 'se_flat'):
        if name in k and name not in marks: marks[name] = t0
print('first time stamps (ms):', {k: round(v, 2) for k, v in marks.items()})
by = collections.defaultdict(list)
for k, t0, t1, st in recs: by[st].append((t0, t1, k))
for st, v in sorted(by.items(), key=lambda kv: -len(kv[1]))[:2]:
    print(f'last launches on stream {"main" if st == main else hex(st)}:')
    for t0, t1, k in sorted(v)[-6:]: print(f'      {t0:7.2f} -> {t1:7.2f}  {k[:70]}')
# how far the weight-gradient stream lags: for each of its launches, the time between the main stream reaching the hand-over
# point (its start stamp can not be earlier) and its end
leaf = sorted(by.items(), key=lambda kv: -len(kv[1]))[1][1]
idle = 0.0; prev = None
for t0, t1, k in sorted(leaf):
    prev = t1
busy = sum(t1 - t0 for t0, t1, _ in leaf)
span = max(t1 for _, t1, _ in leaf) - min(t0 for t0, _, _ in leaf)
print(f'weight-gradient stream: span {span:.2f} ms, inside launches {busy:.2f} ms, waiting for the main stream {span - busy:.2f} ms')
