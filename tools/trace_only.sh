#!/bin/bash
# kernel-trace statistics of the default bench (no PMC passes): tools/trace_only.sh <tag> [bench args]
TAG=${1:-t0}
shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o t --output-format csv -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline "$@" > $OUT/bench.json 2> $OUT/trace.log
cp $OUT/trace/*/t_kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null || cp $OUT/trace/t_kernel_stats.csv $OUT/kernel_stats.csv
rm -rf $OUT/trace
cut -c1-200 $OUT/bench.json
