#!/bin/bash
# per-queue busy time, overlap and idle gaps of the last traced step of the default bench: tools/trace_timeline.sh <tag> [bench args]
# NOTE: under rocprofv3 --kernel-trace all dispatches of this workload come out on ONE queue and strictly serialised (no two kernels
# in flight), so this shows the SERIAL order only: kernel-boundary gaps (3.5 ms over 936 launches) and the main-queue top list.
# For the concurrency of the real step use tools/streams_timeline.py (HIP events on each stream, no profiler attached).
TAG=${1:-tl0}
shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $OUT/trace -o t --output-format csv -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline "$@" > $OUT/bench.json 2> $OUT/trace.log
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
f = glob.glob(out + '/trace/**/t_kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
cols = rows[0].keys()
qk = 'Queue_Id' if 'Queue_Id' in cols else [c for c in cols if 'ueue' in c][0]
ev = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r[qk], r['Kernel_Name'][:50]) for r in rows))
# step boundaries: weight_prep_kernel runs once per step, at the start of the generator forward
marks = [s for s, e, q, n in ev if 'weight_prep' in n]
o = open(out + '/timeline.txt', 'w')
o.write(f'columns: {list(cols)}\n')
if len(marks) < 3:
    o.write('no step markers\n'); sys.exit(0)
# two stft launches per step (noisy + clean) or more: take the last third of the trace instead
t_end = ev[-1][1]
starts = sorted(set(marks))
# cluster marks closer than 5 ms
cl = [starts[0]]
for m in starts[1:]:
    if m - cl[-1] > 20e6: cl.append(m)
o.write(f'step starts (ms from first): {[round((c - cl[0]) / 1e6, 1) for c in cl]}\n')
t0, t1 = cl[-2], cl[-1]
step = [(s, e, q, n) for s, e, q, n in ev if t0 <= s < t1]
o.write(f'step window {round((t1 - t0) / 1e6, 2)} ms, {len(step)} kernels\n')
byq = collections.defaultdict(list)
for s, e, q, n in step: byq[q].append((s, e, n))
for q, v in sorted(byq.items(), key=lambda kv: -sum(e - s for s, e, _ in kv[1])):
    busy = sum(e - s for s, e, _ in v) / 1e6
    o.write(f'queue {q}: {len(v)} kernels, busy {busy:.2f} ms, first at {(v[0][0] - t0) / 1e6:.2f} ms, last ends {(max(e for _, e, _ in v) - t0) / 1e6:.2f} ms\n')
# coverage: time with 0 / 1 / >= 2 kernels running
pts = []
for s, e, q, n in step: pts.append((s, 1)); pts.append((e, -1))
pts.sort()
cov = collections.Counter(); cur = 0; last = t0
for t, d in pts:
    cov[min(cur, 3)] += t - last; last = t; cur += d
cov[0] += max(0, t1 - last)
o.write('time with k kernels in flight (ms): ' + ', '.join(f'{k}: {v / 1e6:.2f}' for k, v in sorted(cov.items())) + '\n')
# main queue = the one with most kernels; its gaps > 20 us
mq = max(byq, key=lambda q: len(byq[q]))
v = sorted(byq[mq])
gaps = [(v[i + 1][0] - v[i][1], v[i][2], v[i + 1][2], (v[i][1] - t0) / 1e6) for i in range(len(v) - 1) if v[i + 1][0] - v[i][1] > 20000]
o.write(f'main queue {mq}: {len(gaps)} gaps > 20 us, total {sum(g[0] for g in gaps) / 1e6:.2f} ms; all gaps total {sum(max(0, v[i + 1][0] - v[i][1]) for i in range(len(v) - 1)) / 1e6:.2f} ms\n')
for g in sorted(gaps, reverse=True)[:25]:
    o.write(f'   gap {g[0] / 1e3:8.1f} us at {g[3]:7.2f} ms after [{g[1]}] before [{g[2]}]\n')
# per-kernel slowdown on the main queue is not computable here; list the main queue's top kernels by time
agg = collections.Counter()
for s, e, n in v: agg[n] += e - s
o.write('main queue top kernels (ms):\n')
for n, t in agg.most_common(25): o.write(f'   {t / 1e6:7.2f}  {n}\n')
PY
rm -rf $OUT/trace
cat $OUT/timeline.txt
