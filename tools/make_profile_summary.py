"""Builds the per-round profile summary that bench.py and DESIGN.md quote.

  python tools/make_profile_summary.py gpurun_out/r02a r02a

inputs (written on the GPU box by tools/profile_round.sh):
  kernel_stats.csv   rocprofv3 --kernel-trace --stats of `bench.py --steps 5 --warmup 2` (7 steps in the trace)
  pmc_summary.json   per-kernel averages per dispatch of the SEPARATE --pmc passes (SQ set 1, SQ set 2, FETCH_SIZE,
                     WRITE_SIZE) over `bench.py --steps 1 --warmup 1` (tools/pmc_summary.py)
outputs:
  profiles/<tag>_kernel_stats.csv      (copy)
  profiles/<tag>_kernel_summary.json / .md   per kernel: calls per step, average duration, share of GPU time, MFMA-busy %,
                                       VALU-busy %, wave wait %, LDS bank-conflict ratio, HBM-side bytes per launch, GB/s
  profiles/pmc_traffic.json            bytes per launch keyed by bench.py's KernelTimer family names (roofline.traffic)

Derived quantities (MI355X_MICROARCH.md, rocprofv3 section): SQ_BUSY_CYCLES is summed over the 32 shader engines, so
kernel cycles = SQ_BUSY_CYCLES / 32; SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over the 1024 SIMDs -> MFMA-busy =
MFMA_BUSY / (32 * BUSY); SQ_ACTIVE_INST_* / SQ_WAVE_CYCLES / SQ_WAIT_* count quad-cycles -> VALU-busy = 4 * ACTIVE_VALU /
(1024 * BUSY / 32); HBM-side bytes = (2 * FETCH_SIZE + WRITE_SIZE) KB (gfx950 tallies 128-B read requests at 64 B)."""
import csv, json, os, re, shutil, sys

src, tag = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
steps_in_trace = int(sys.argv[3]) if len(sys.argv) > 3 else 7


def short(n):
    n = n.replace('void ', '')
    return re.sub(r'\(.*', '', n).strip()[:80]


def fam(name):
    """KernelTimer family (bench.py roofline keys) of a kernel name, or None.  The LAST template argument of the scaled split-fp16
    instantiations (precision 3) is F16 = true (ff_fwd carries one more flag behind it)."""
    n = short(name)
    args = [a.strip() for a in re.sub(r'^[^<]*<?', '', n).rstrip('>').split(',')] if '<' in n else []
    base = n.split('<')[0]
    if base == 'gemm_tap_bf16x3_kernel':              # <PRO, NPL, planes, LIN, NOLOAD?, F16>
        if args[-1] == 'true':
            return f'gemm_tap_f16x3_kernel<{args[0]}>'
        return f'gemm_tap_bf16x{3 if args[1] == "2" else 6}_kernel<{args[0]}>'
    if base == 'conv3_bf16_kernel':                   # <NPL, planes, F16, ...>
        if args[0] == '2' and args[2] == 'true':
            return 'conv3_f16x3'
        return f'conv3_bf16x{3 if args[0] == "2" else 6}'
    if base == 'conv1d_k64_wstat_kernel':             # in the train step: the K = 64 row GEMMs without a prologue (64 -> 128 input gradient)
        return 'gemm_k64_panel_f16x3<0>'
    if base in ('gemm_k64_wstat_kernel',):            # the W-stationary row panel of the train step (scaled fp16 only)
        return f'gemm_k64_panel_f16x3<{args[0]}>'
    if base == 'gemm_k64_panel_kernel':
        if args[-1] == 'true' and args[1] == '2' and len(args) >= 5:
            return f'gemm_k64_panel_f16x3<{args[0]}>'
        return f'gemm_k64_panel_bf16x{3 if args[1] == "2" else 6}<{args[0]}>'
    if base == 'dwconv_kernel':
        return 'dwconv31 dgrad + glu_bwd' if len(args) > 1 and args[1] == 'true' else 'dwconv31 (fwd / dgrad)'
    if base in ('dwconv_wgrad_kernel', 'dwconv_wgrad_reduce_kernel'):
        return 'dwconv31_wgrad (+ reduce)'
    # round 5: the fused kernels
    if base == 'dwconv_bwd_fused_kernel':
        return 'dwconv31 dgrad + glu_bwd + wgrad'
    if base == 'ff_fwd_ws_kernel':
        return 'ff_fwd_f16x3'
    if base == 'ff_bwd_fused_kernel':
        return 'ff_bwd_fused_f16x3'
    if base == 'lnbwd_fused_kernel':
        return 'lnbwd_wgrad_fused_f16x3'
    if base in ('ff_fwd_kernel', 'ff_bwd_kernel'):    # <NPL, planes, NB, F16 [, STORE_H]>
        kind = 'fwd' if base == 'ff_fwd_kernel' else 'bwd_dgrad'
        if len(args) >= 4 and args[3] == 'true':
            return f'ff_{kind}_f16x3'
        return f'ff_{kind}_bf16x{3 if args[0] == "2" else 6}'
    # weight gradients: keyed like gemm._wgrad_key (class + the arithmetic the C side dispatched to)
    if base == 'wgrad_kernel':
        return f'wgrad_f32<{args[0]}>'
    if base == 'wgrad_bf16_kernel':                   # <PRO, NPL>
        return f'wgrad_bf16x{3 if args[1] == "2" else 6}<{args[0]}>'
    if base == 'wgrad_lin_kernel':
        return f'wgrad_lin_f32<{args[0]}>'
    if base == 'wgrad_lin_bf16_kernel':               # <PRO, SH [, F16]>
        return f'wgrad_lin_f16x3<{args[0]}>' if args[-1] == 'true' else f'wgrad_lin_bf16x6<{args[0]}>'
    if base == 'wgrad3_kernel':
        return 'wgrad3_f32<0>'
    if base == 'wgrad3w_f16_kernel':
        return 'wgrad3_f16x3<0>'
    if base == 'wgrad3_bf16_kernel':                  # <NPL [, F16 [, ORD]]>
        if len(args) > 1 and args[1] == 'true':
            return 'wgrad3_f16x3<0>'
        return f'wgrad3_bf16x{3 if args[0] == "2" else 6}<0>'
    if base == 'gemm_tap_kernel':
        return f'gemm_tap_kernel<{args[0]},{args[1]}>'
    if base == 'attn_bwd4_kernel':                    # <NW, KPW, NCW, NKTM, MINW>: scaled fp16 only
        return 'attn_bwd_f16x3 (+tables, dE reduce) ' + ('n>128' if args[1] != '2' else 'n<=128')
    if base == 'attn_fwd3_kernel':
        return 'attn_fwd3_f16x3' if len(args) > 1 and args[1] == 'true' else 'attn_fwd3_bf16x6'
    if base in ('stft_fused_kernel', 'istft_fused_kernel'):
        return base.replace('_kernel', '')
    return None


stats = list(csv.DictReader(open(os.path.join(src, 'kernel_stats.csv'))))
pmc = json.load(open(os.path.join(src, 'pmc_summary.json')))


def demangle(names):
    import subprocess
    for tool in ('c++filt', '/opt/rocm/lib/llvm/bin/llvm-cxxfilt'):
        try:
            out = subprocess.run([tool], input='\n'.join(names), capture_output=True, text=True, check=True).stdout.split('\n')
            return dict(zip(names, out))
        except Exception:
            continue
    return {n: n for n in names}


_dm = demangle([k.replace('.kd', '') for k in pmc])
pmc = {short(_dm[k.replace('.kd', '')]): v for k, v in pmc.items()}
total_ns = sum(float(r['TotalDurationNs']) for r in stats)
rows = []
for r in stats:
    k = short(r['Name'])
    p = pmc.get(k) or pmc.get(k.replace(' ', '')) or {}
    if not p:       # rocprof's csv carries demangled names, the db mangled ones: match on the leading identifier
        key = re.sub(r'[<(].*', '', k)
        cand = [v for kk, v in pmc.items() if key and key in kk]
        p = cand[0] if len(cand) == 1 else {}
    if not p:       # a mangled name the PMC summary cut short (c++filt refuses it): compare (identifier, integer template arguments)
        base = re.sub(r'^void ', '', re.sub(r'[<(].*', '', k)).strip()
        args = [a.strip() for a in re.sub(r'^[^<]*<', '', k).rsplit('>', 1)[0].split(',')] if '<' in k else []
        args = [{'true': '1', 'false': '0'}.get(a, a) for a in args]
        for kk, v in pmc.items():
            mm = re.match(r'^_Z\d+' + re.escape(base) + r'I((?:L[ib]n?\d+E)+)E', kk)
            if mm and [('-' if t[2] == 'n' else '') + re.sub(r'\D', '', t) for t in re.findall(r'L[ib]n?\d+', mm.group(1))] == args:
                p = v
                break
    row = {'kernel': k, 'calls_per_step': round(int(r['Calls']) / steps_in_trace, 2), 'avg_us': round(float(r['AverageNs']) / 1e3, 1),
           'pct_gpu_time': round(100 * float(r['TotalDurationNs']) / total_ns, 2)}
    busy = p.get('SQ_BUSY_CYCLES')
    if busy:
        row['mfma_busy_pct'] = round(100 * p.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (32 * busy), 1)
        row['valu_busy_pct'] = round(100 * 4 * p.get('SQ_ACTIVE_INST_VALU', 0) / (1024 * busy / 32), 1)
        if p.get('SQ_WAVE_CYCLES'):
            row['wave_wait_pct'] = round(100 * p.get('SQ_WAIT_ANY', 0) / p['SQ_WAVE_CYCLES'], 1)
        if p.get('SQ_LDS_IDX_ACTIVE'):
            row['lds_conflict_ratio'] = round(p.get('SQ_LDS_BANK_CONFLICT', 0) / p['SQ_LDS_IDX_ACTIVE'], 2)
    if 'FETCH_SIZE' in p or 'WRITE_SIZE' in p:
        row['fetch_bytes_x2'] = int(2 * p.get('FETCH_SIZE', 0) * 1024)
        row['write_bytes'] = int(p.get('WRITE_SIZE', 0) * 1024)
        row['hbm_side_gbs'] = round((row['fetch_bytes_x2'] + row['write_bytes']) / (float(r['AverageNs']) * 1e-9) / 1e9, 0)
    rows.append(row)
os.makedirs(os.path.join(ROOT, 'profiles'), exist_ok=True)
shutil.copy(os.path.join(src, 'kernel_stats.csv'), os.path.join(ROOT, 'profiles', f'{tag}_kernel_stats.csv'))
for f in ('bench_line.json', 'bench_under_rocprof.json'):
    if os.path.exists(os.path.join(src, f)):
        shutil.copy(os.path.join(src, f), os.path.join(ROOT, 'profiles', f'{tag}_{f}'))
json.dump({'source': __doc__, 'kernels': rows}, open(os.path.join(ROOT, 'profiles', f'{tag}_kernel_summary.json'), 'w'), indent=1)
with open(os.path.join(ROOT, 'profiles', f'{tag}_kernel_summary.md'), 'w') as f:
    f.write(f'# {tag}: per-kernel summary (rocprofv3 kernel trace of `bench.py --steps 5 --warmup 2` + separate PMC passes)\n\n')
    f.write('| kernel | calls/step | avg us | % GPU time | MFMA busy % | VALU busy % | wave wait % | LDS conflict / active | HBM-side bytes/launch | GB/s |\n')
    f.write('|---|---|---|---|---|---|---|---|---|---|\n')
    for r in rows[:40]:
        f.write('| `%s` | %s | %s | %s | %s | %s | %s | %s | %s | %s |\n' % (
            r['kernel'], r['calls_per_step'], r['avg_us'], r['pct_gpu_time'], r.get('mfma_busy_pct', ''), r.get('valu_busy_pct', ''),
            r.get('wave_wait_pct', ''), r.get('lds_conflict_ratio', ''),
            (r['fetch_bytes_x2'] + r['write_bytes']) if 'fetch_bytes_x2' in r else '', r.get('hbm_side_gbs', '')))
# traffic per KernelTimer family (launch-weighted average over the kernels of the family)
famacc = {}
for r, st in zip(rows, stats):
    k = fam(st['Name'])
    if k is None or 'fetch_bytes_x2' not in r:
        continue
    a = famacc.setdefault(k, {'bytes': 0.0, 'calls': 0, 'mfma_busy_w': 0.0, 'ns': 0.0})
    a['bytes'] += (r['fetch_bytes_x2'] + r['write_bytes']) * int(st['Calls'])
    a['calls'] += int(st['Calls'])
    a['mfma_busy_w'] += r.get('mfma_busy_pct', 0.0) * float(st['TotalDurationNs'])
    a['ns'] += float(st['TotalDurationNs'])
out = {'source': f'tools/make_profile_summary.py {tag}: separate rocprofv3 --pmc passes (FETCH_SIZE x2 + WRITE_SIZE, KB)', 'kernels': {}}
for k, a in famacc.items():
    out['kernels'][k] = {'traffic_bytes_per_launch': int(a['bytes'] / max(a['calls'], 1)),
                         'mfma_busy_pct': round(a['mfma_busy_w'] / max(a['ns'], 1.0), 1)}
# step-level figures (the committed profile the bench line quotes): time-weighted MFMA-busy over every kernel of the trace and the
# HBM-side bytes of one step (2 FETCH_SIZE + WRITE_SIZE over all launches of a step)
tw = sum(r.get('mfma_busy_pct', 0.0) * float(st['TotalDurationNs']) for r, st in zip(rows, stats))
out['step'] = {'mfma_busy_pct_time_weighted': round(tw / max(total_ns, 1.0), 1),
               'hbm_side_gb_per_step': round(sum((r.get('fetch_bytes_x2', 0) + r.get('write_bytes', 0)) * int(st['Calls'])
                                                 for r, st in zip(rows, stats)) / steps_in_trace / 1e9, 1),
               'kernel_ms_per_step_serial_order': round(total_ns / steps_in_trace / 1e6, 2),
               'launches_per_step': round(sum(int(st['Calls']) for st in stats) / steps_in_trace)}
json.dump(out, open(os.path.join(ROOT, 'profiles', 'pmc_traffic.json'), 'w'), indent=1)
bl = os.path.join(src, 'bench_line.json')
if os.path.exists(bl):          # every family of the bench line that carries work must have found its counters
    try:
        fams = json.loads(open(bl).read().strip().splitlines()[-1])['roofline']['families']
        missing = [k for k in fams if k not in out['kernels']]
        print('families of the bench line WITHOUT PMC counters:', missing)
    except Exception as e:
        print('bench line not parsed:', e)
print(open(os.path.join(ROOT, 'profiles', f'{tag}_kernel_summary.md')).read()[:6000])
