import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/t_kernel_trace.csv', recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:70], r.get('Stream_Id', r.get('Queue_Id', '?'))))
rows.sort()
ends = [i for i, r in enumerate(rows) if 'adamw' in r[2]]
steps = ends[3::4]
i0, i1 = steps[-2], steps[-1]
win = rows[i0 + 1:i1 + 1]
t0 = win[0][0]; t1 = max(r[1] for r in win)
print('step', (t1 - t0) / 1e6, 'ms, kernels', len(win))
# concurrency timeline in 1-ms bins: busy time per stream id
import collections
bins = collections.defaultdict(lambda: collections.defaultdict(float))
for s, e, n, q in win:
    b0 = int((s - t0) / 1e6)
    b1 = int((e - t0) / 1e6)
    for b in range(b0, b1 + 1):
        lo = max(s, t0 + b * 1e6); hi = min(e, t0 + (b + 1) * 1e6)
        if hi > lo: bins[b][q] += (hi - lo) / 1e6
qs = sorted({q for s, e, n, q in win})
print('queues', qs)
for b in sorted(bins):
    print(f'{b:3d} ms ' + ' '.join(f'{bins[b][q]:.2f}' for q in qs))
print('last 25 kernels:')
for s, e, n, q in win[-25:]:
    print(f'{(s - t0) / 1e6:8.3f} {(e - s) / 1e3:8.1f} us q{q} {n}')
