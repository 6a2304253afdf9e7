"""Builds profiles/r01_pmc_traffic.json from two separate rocprofv3 PMC passes (never combined with other traces):

  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc_fetch2 -o f --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc_write2 -o w --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline

Per launch: bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (counter unit KB; x2: on gfx950 FETCH_SIZE tallies the 128-B requests
of 16-B/lane loads at 64 B -- MI355X_MICROARCH.md, HBM section; calibrated here on row_stats64_kernel, which reads exactly
132.8 MB and reports 66.4 MB).  Keys are the kernel-family names bench.py's KernelTimer uses."""
import collections
import csv
import json
import re
import sys

fetch_csv = sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/pmc_fetch2/f_counter_collection.csv'
write_csv = sys.argv[2] if len(sys.argv) > 2 else 'gpurun_out/pmc_write2/w_counter_collection.csv'
out_json = sys.argv[3] if len(sys.argv) > 3 else 'profiles/r01_pmc_traffic.json'


def agg(path, counter):
    tot, cnt = collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] == counter:
            tot[r['Kernel_Name']] += float(r['Counter_Value'])
            cnt[r['Kernel_Name']] += 1
    return tot, cnt


def fam(name):
    n = name.split('(')[0].replace('void ', '').strip()
    m = re.match(r'gemm_tap_bf16x3_kernel<(\d), (\d), (true|false)>', n)
    if m:
        return f'gemm_tap_bf16x{3 if m.group(2) == "2" else 6}_kernel<{m.group(1)}>'
    m = re.match(r'conv3_bf16_kernel<(\d)>', n)
    if m:
        return f'conv3_bf16x{3 if m.group(1) == "2" else 6}'
    m = re.match(r'gemm_k64_panel_kernel<(\d), (\d), (true|false)>', n)
    if m:
        return f'gemm_k64_panel_bf16x{3 if m.group(2) == "2" else 6}<{m.group(1)}>'
    m = re.match(r'gemm_tap_kernel<(\d+), (\d), (true|false)>', n)
    if m:
        return f'gemm_tap_kernel<{m.group(1)},{m.group(2)}>'
    m = re.match(r'wgrad_kernel<(\d)>', n)
    if m:
        return f'wgrad_kernel<{m.group(1)}>'
    if n.startswith('wgrad3_kernel') or n.startswith('wgrad3_bf16_kernel'):
        return 'wgrad_kernel<0>'
    m = re.match(r'ff_(fwd|bwd)_kernel<(\d)>', n)
    if m:
        return f'ff_{m.group(1)}{"_dgrad" if m.group(1) == "bwd" else ""}_bf16x{3 if m.group(2) == "2" else 6}'
    if n.startswith('attn_bwd2_kernel'):
        return n + ' (+delta)'
    return n


f, fc = agg(fetch_csv, 'FETCH_SIZE')
w, wc = agg(write_csv, 'WRITE_SIZE')
out = collections.defaultdict(lambda: {'fetch_raw_kb': 0.0, 'write_kb': 0.0, 'launches': 0})
for k in set(f) | set(w):
    o = out[fam(k)]
    o['fetch_raw_kb'] += f.get(k, 0.0)
    o['write_kb'] += w.get(k, 0.0)
    o['launches'] += max(fc.get(k, 0), wc.get(k, 0))
res = {'source': __doc__, 'kernels': {}}
for k, o in sorted(out.items(), key=lambda kv: -(2 * kv[1]['fetch_raw_kb'] + kv[1]['write_kb'])):
    n = max(o['launches'], 1)
    res['kernels'][k] = {'launches_in_2_steps': n, 'fetch_bytes_per_launch_x2': round(2 * o['fetch_raw_kb'] * 1024 / n),
                         'write_bytes_per_launch': round(o['write_kb'] * 1024 / n),
                         'traffic_bytes_per_launch': round((2 * o['fetch_raw_kb'] + o['write_kb']) * 1024 / n)}
# se_attn_bwd = attn_delta_kernel + attn_bwd2_kernel behind one KernelTimer key
dl = res['kernels'].get('attn_delta_kernel')
if dl:
    for k, v in res['kernels'].items():
        if k.endswith('(+delta)'):
            for fld in ('fetch_bytes_per_launch_x2', 'write_bytes_per_launch', 'traffic_bytes_per_launch'):
                v[fld] += dl[fld]
json.dump(res, open(out_json, 'w'), indent=1)
print('wrote', out_json, len(res['kernels']), 'kernel families')
