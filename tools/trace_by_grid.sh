#!/bin/bash
# per-launch-shape durations of the kernels matching a regex over the default bench: tools/trace_by_grid.sh <tag> <regex> [bench args]
TAG=${1:-t0}; RX=${2:-.}
shift; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $OUT/trace -o t --output-format csv -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline "$@" > $OUT/bench.json 2> $OUT/trace.log
python3 - "$OUT" "$RX" <<'PY'
import csv, glob, re, sys, collections
out, rx = sys.argv[1], re.compile(sys.argv[2])
f = glob.glob(out + '/trace/**/t_kernel_trace.csv', recursive=True)[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if rx.search(r['Kernel_Name']):
        agg[(r['Kernel_Name'][:60], r['Grid_Size_X'], r['Grid_Size_Y'], r['Workgroup_Size_X'])].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
with open(out + '/by_grid.txt', 'w') as o:
    for k, v in sorted(agg.items()):
        v.sort()
        o.write(f'{k[0]:60s} grid=({k[1]},{k[2]}) wg={k[3]} n={len(v)} min={v[0]:.1f} med={v[len(v)//2]:.1f} max={v[-1]:.1f} us\n')
PY
rm -rf $OUT/trace
cat $OUT/by_grid.txt
