"""GPU idle time of the default (three-stream) train step from a rocprofv3 kernel trace: union of the kernel intervals of all
streams over the timed steps vs wall time, the largest gaps and the kernels around them.
usage (GPU box): rocprofv3 --kernel-trace -d DIR -o t --output-format csv -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline
                 python3 tools/trace_gaps.py DIR"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/t_kernel_trace.csv', recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:60]))
rows.sort()
# the optimizer kernel closes a step: take the window between the 3rd-last and the last adamw of the generator
ends = [i for i, r in enumerate(rows) if 'adamw' in r[2]]
per_step = 4                      # AdamW launches per step: two flat buffers (decay / no decay) for each of the two models
steps = ends[per_step - 1::per_step]
i0, i1 = steps[-3], steps[-1]
win = rows[i0 + 1:i1 + 1]
t0, t1 = win[0][0], max(r[1] for r in win)
busy, cur_s, cur_e, gaps = 0, win[0][0], win[0][1], []
prev = win[0][2]
for s, e, n in win[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append((s - cur_e, prev, n))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
    prev = n
busy += cur_e - cur_s
nsteps = 2
print(f'window {(t1 - t0) / 1e6 / nsteps:.2f} ms/step, GPU busy (union over streams) {busy / 1e6 / nsteps:.2f} ms/step, idle {(t1 - t0 - busy) / 1e6 / nsteps:.2f} ms/step in {len(gaps) // nsteps} gaps/step')
print('sum of kernel durations', sum(e - s for s, e, n in win) / 1e6 / nsteps, 'ms/step')
for g, a, b in sorted(gaps, reverse=True)[:25]:
    print(f'{g / 1e3:8.1f} us  after {a[:45]:45s} before {b[:45]}')
