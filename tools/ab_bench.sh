#!/bin/bash
# same-box A/B of two trees: tools/ab_bench.sh <other tree> [rounds] -> gpurun_out/ab_<k>_{base,new}.json (headline loop only, no CPU leg)
other=$1; rounds=${2:-2}
for k in $(seq 1 $rounds); do
  (cd $other && python bench.py --steps 20 --warmup 5 --no-cpu-baseline) > gpurun_out/ab_${k}_base.json 2> gpurun_out/ab_${k}_base.err
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/ab_${k}_new.json 2> gpurun_out/ab_${k}_new.err
done
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/ab_*_*.json')):
    try:
        d=json.load(open(f)); fam=d['roofline']['families']
        print(f, d['value'], d['ms_per_step'], {k.split(' ')[0][:22]:v['ms_per_step'] for k,v in list(fam.items())[:9]})
    except Exception as e: print(f,'ERR',e)
P
