#!/bin/bash
# same-box A/B of the default bench step: tools/micro/step_ab.sh "ENV_A=.." "ENV_B=.." [steps]   (empty string = defaults)
steps=${3:-10}
for rep in 1 2; do
  for cfg in "$1" "$2"; do
    echo "== [$cfg]"
    env SE_BENCH_NO_SECONDARY=1 $cfg python bench.py --no-cpu-baseline --steps $steps --warmup 3 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('ms_per_step', d['ms_per_step'], 'value', d['value'])"
  done
done
