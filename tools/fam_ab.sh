#!/bin/bash
# same-box A/B of the product library against a variant library (tools/build_variant_lib.sh): tools/fam_ab.sh <variant .so> [rounds]
# -> per family ms per step of both, from bench.py's serial-order attribution pass (headline loop only, no CPU leg)
lib=$1; rounds=${2:-2}
for k in $(seq 1 $rounds); do
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/fam_${k}_prod.json 2> gpurun_out/fam_${k}_prod.err
  SE_HIP_LIB=$lib python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/fam_${k}_var.json 2> gpurun_out/fam_${k}_var.err
done
python - <<'P'
import json,glob
r={}
for f in sorted(glob.glob('gpurun_out/fam_*_*.json')):
    d=json.load(open(f)); w=f.split('_')[-1][:-5]
    r.setdefault(w,[]).append(d)
fams=list(r['prod'][0]['roofline']['families'].keys())
print('ms per step   product ' + ' '.join(f"{d['ms_per_step']:.2f}" for d in r['prod']) + '   variant ' + ' '.join(f"{d['ms_per_step']:.2f}" for d in r['var']))
for k in fams:
    p=[d['roofline']['families'][k]['ms_per_step'] for d in r['prod']]; v=[d['roofline']['families'].get(k,{'ms_per_step':0})['ms_per_step'] for d in r['var']]
    dlt=sum(v)/len(v)-sum(p)/len(p)
    if abs(dlt) > 0.02: print(f"{k[:60]:60s} product {' '.join(f'{x:.3f}' for x in p)}   variant {' '.join(f'{x:.3f}' for x in v)}   delta {dlt:+.3f}")
P
