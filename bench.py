#!/usr/bin/env python3
"""Headline benchmark: utterances/sec of the CMGAN train step (generator fwd+bwd+AdamW, discriminator 3x fwd +
bwd + AdamW, 2 STFT + 1 iSTFT fwd + 1 iSTFT bwd) on synthetic 2 s / 16 kHz clips, batch 16 per MI355X
(BASELINE.json configs[1]).  One process per GPU; N > 1 is launched by torch.distributed.run (RCCL).

Prints ONE JSON line on rank 0 (see the driver contract in the task statement).  `roofline` is measured live
with HIP events around every launch of the dominant kernel family during the timed steps; `cpu_baseline` times
the CPU oracle (a torch-CPU port of the same step) on a bounded sample on the host cores (rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time
import types

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_* dense peak (== fp32 vector peak)
PEAK_BF16_MFMA_TFLOPS = 2500.0    # MI355X_MICROARCH.md: dense bf16 MFMA peak
PEAK_HBM_GBS = 8000.0
GFLOP_PER_UTT_STEP = 440.0        # 3 x 145.96 (G fwd+bwd) + ~1.9 (D)


def synth_batch(B, L, seed, device):
    """BASELINE.md section 2 inputs: clean = 0.1 N(0,1), noisy = clean + 0.05 N(0,1), Q ~ U(0.2, 0.9)."""
    g = torch.Generator().manual_seed(seed)
    clean = 0.1 * torch.randn(B, L, generator=g)
    noisy = clean + 0.05 * torch.randn(B, L, generator=g)
    q = 0.2 + 0.7 * torch.rand(B, generator=torch.Generator().manual_seed(seed + 1))
    return clean.to(device), noisy.to(device), q.to(device)


def _oracle_step(B, threads):
    from oracle import se_oracle as Or
    import formula
    torch.set_num_threads(threads)
    gsd, dsd = formula.formula_state('generator'), formula.formula_state('discriminator')
    clean, noisy, q = synth_batch(B, 32000, 1, 'cpu')
    t0 = time.time()
    Or.train_step(gsd, dsd, clean, noisy, q, 'cmgan', (0.1, 0.9, 0.2, 0.05), lr=5e-4)
    return time.time() - t0


try:
    _DW_TWIN = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles', 'r04_dwconv_twin.json')))['inside_step']
except Exception:
    _DW_TWIN = {}


def _pipe16_parts(key):
    """16-bit MFMAs a kernel family executes per fp32-equivalent product (None: fp32 MFMA / no MFMA)"""
    if 'f16x3' in key or 'bf16x3' in key:
        return 3
    if 'bf16x6' in key:
        return 6
    return None


def _family_entry(key, v, steps, pmc):
    """one line of roofline.families: time per step + the rate against the roof that bounds the family; every scaled-fp16 /
    split-bf16 family ALSO carries frac_of_16bit_pipe (executed 16-bit MFMA rate / 2.5 PFLOP/s: the same arithmetic, one peak)"""
    e = {'ms_per_step': round(v['ms'] / steps, 3), 'launches_per_step': round(v['launches'] / steps, 2)}
    sec = v['ms'] * 1e-3
    if key in ('stft_fused', 'istft_fused'):
        # 0.64 MB per utterance and transform: latency-sized launches (0.2 % of the step); both roofs stated, neither binds
        nbytes = v['flops'] / (2.0 * 400 * 402) * (400 + 402) * 4.0          # frames x (400 samples in + 402 plane floats out)
        e.update({'tflops': round(v['flops'] / sec / 1e12, 2), 'gbs': round(nbytes / sec / 1e9, 1),
                  'frac_of_hbm_peak': round(nbytes / sec / 1e9 / PEAK_HBM_GBS, 4), 'bound': 'latency (10 MB per launch)'})
    elif v['flops'] > 0:
        tf = v['flops'] / sec / 1e12
        e['tflops'] = round(tf, 2)
        if _pipe16_parts(key):
            e['frac_of_16bit_pipe'] = round(tf * _pipe16_parts(key) / PEAK_BF16_MFMA_TFLOPS, 4)
        else:
            e['frac_of_f32_mfma'] = round(tf / PEAK_F32_MFMA_TFLOPS, 4)
    else:                                                                     # HBM-bound families: algorithmic bytes
        e.update({'gbs': round(v['bytes'] / sec / 1e9, 1), 'frac_of_hbm_peak': round(v['bytes'] / sec / 1e9 / PEAK_HBM_GBS, 3)})
    if key in _DW_TWIN:          # committed measurement of the access-pattern twin (profiles/r04_dwconv_twin.json): the structure's ceiling
        e['access_pattern_twin_gbs'] = _DW_TWIN[key]['twin_gbs']
        e['access_pattern_twin_frac_of_hbm_peak'] = round(_DW_TWIN[key]['twin_gbs'] / PEAK_HBM_GBS, 3)
        e['twin_source'] = 'profiles/r04_dwconv_twin.json (same loads / LDS staging / stores without the 31-tap FIR, inside the step)'
    p = pmc.get(key, {})
    if p.get('traffic_bytes_per_launch') is not None:
        e['traffic_bytes_per_launch_pmc'] = p['traffic_bytes_per_launch']
    if p.get('mfma_busy_pct'):
        e['mfma_busy_pct_pmc'] = p['mfma_busy_pct']
    return e


def cpu_baseline(budget_s=40.0):
    """The CPU oracle's CMGAN train step (torch-CPU port of the reference step: AdamW, PESQ labels supplied) timed on
    the host cores: thread count chosen by a quick sweep of a generator-only forward, then 1 warm-up + up to 3 timed
    steps at batch 2 (bounded by `budget_s`), and one batch-16 step if the batch-2 step is fast enough to afford it."""
    from oracle import se_oracle as Or
    import formula
    ncpu = os.cpu_count() or 1
    gsd = formula.formula_state('generator')
    clean, noisy, _ = synth_batch(2, 32000, 1, 'cpu')
    cn, nn_, _c = Or.normalize_pair(clean, noisy)
    spec = Or.compressed_stft(nn_)
    sweep = {}
    t_start = time.time()
    for th in sorted({min(ncpu, v) for v in (16, 32)}):      # 8 and 64 never won on the pool's hosts; beyond 64 torch-CPU oversubscribes (256 threads: 60x slower)
        torch.set_num_threads(th)
        with torch.no_grad():
            Or.tscnet_forward(gsd, spec, False)              # warm (allocator, thread pool)
            t0 = time.time()
            Or.tscnet_forward(gsd, spec, False)
        sweep[th] = round(time.time() - t0, 3)
    threads = min(sweep, key=sweep.get)
    warm = _oracle_step(2, threads)
    times = []
    while not times or (len(times) < 3 and (time.time() - t_start) + times[-1] < budget_s):
        times.append(_oracle_step(2, threads))
    dt = sum(times) / len(times)
    res = {'value': round(2.0 / dt, 4), 'unit': 'utterances/sec', 'cores': threads, 'host_cores': ncpu, 'kind': 'port',
           'sample': f'CMGAN train step of the CPU oracle (torch-CPU port of the reference step), batch 2, 2 s clips, '
                     f'AdamW, PESQ labels supplied: 1 warm-up ({warm:.1f} s) + {len(times)} timed steps '
                     f'({", ".join("%.1f" % x for x in times)} s) on {threads} threads '
                     f'(forward-only thread sweep, s: {sweep})'}
    if os.environ.get('SE_CPU_BASELINE_B16') == '1':          # ~100 s on a 256-core host (0.17 utt/s): opt-in, DESIGN.md quotes it
        t16 = _oracle_step(16, threads)
        res['batch16'] = {'value': round(16.0 / t16, 4), 'seconds': round(t16, 1)}
    return res


def launch_ranks(n, argv):
    """`bench.py --gpus N` without a launcher: start N ranks with torch.distributed.run as a CHILD process -- decided
    before anything in this process touches the GPU -- and hand its exit code back.  Rank 0 of the children prints the
    JSON line on the inherited stdout."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}', '--master-addr',
           '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--batch', type=int, default=16, help='utterances per GPU per step')
    ap.add_argument('--arch', default='cmgan')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    a = ap.parse_args()

    if 'WORLD_SIZE' not in os.environ and a.gpus > 1:
        avail = torch.cuda.device_count()              # does not initialise the GPU
        if avail < a.gpus:
            sys.exit(f'bench.py: --gpus {a.gpus} requested but only {avail} GPU(s) are visible')
        sys.exit(launch_ranks(a.gpus, sys.argv[1:]))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != a.gpus:
        sys.exit(f'bench.py: --gpus {a.gpus} does not match WORLD_SIZE={world} set by the launcher')
    # The CPU-baseline leg needs no GPU: it runs FIRST (rank 0, N = 1 only), bounded to ~30-40 s, so that the GPU section of the
    # command is contiguous and not a sliver at the start of a CPU-dominated run.
    cpu_res = cpu_baseline() if (world == 1 and not a.no_cpu_baseline) else None
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    force_dp = os.environ.get('SE_FORCE_DP') == '1'       # exercise the RCCL hooks on a single rank (self-test)
    if world > 1 or force_dp:
        dist.init_process_group('nccl', device_id=dev)

    import __graft_entry__
    if rank == 0:
        __graft_entry__.build()
    if world > 1:
        dist.barrier()
    import speech_enhancement_amd as S
    from speech_enhancement_amd import _lib, optim, train as TR

    torch.manual_seed(0)
    G, D = S.TSCNet(64, 201), S.Discriminator(16)
    G.apply(S.kaiming_init)
    D.apply(S.kaiming_init)
    G.to(dev).train()
    D.to(dev).train()
    hooks = TR.attach_data_parallel(G, D) if (world > 1 or force_dp) else None
    if hooks is not None and force_dp:
        hooks.world = 2          # take the multi-rank code paths (SyncBN exchange, grad averaging); 1 rank => x0.5 grads
    oargs = types.SimpleNamespace(optimizer='adamw', lr=5e-4, weight_decay=0.01, momentum=0.9, max_norm=0.0)
    og, od = optim.build_optimizer(oargs, G), optim.build_optimizer(oargs, D)
    weights = (0.1, 0.9, 0.2, 0.05) if a.arch in ('cmgan', 'cp') else (0.3, 0.7, 0.2, 0.05)
    B, L = a.batch, 32000
    clean, noisy, q = synth_batch(B, L, 1 + rank, dev)
    labels = {'est': q, 'clean': torch.full_like(q, 0.97), 'noisy': q * 0.5}

    def step():
        return TR.gan_step(G, D, og, od, clean, noisy, a.arch, weights, labels=labels, hooks=hooks)

    for _ in range(a.warmup):
        step()

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    fence()
    t0 = time.time()                     # the HEADLINE loop: no per-launch instrumentation (TIMER off), exactly K steps
    for _ in range(a.steps):
        out = step()
    fence()
    dt = time.time() - t0
    tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax)
    # attribution pass 1 (all ranks: the step contains collectives): 2 more steps in the SAME concurrent stream order with two HIP
    # events around every keyed launch and, for N > 1, around the data-parallel exchanges (`comm`)
    conc_steps = 2
    if hooks is not None:
        hooks.comm_events = []
    _lib.TIMER.start()
    for _ in range(conc_steps):
        step()
    fence()
    _lib.TIMER.stop()
    comm = None
    if hooks is not None:
        evs, hooks.comm_events = hooks.comm_events, None
        per = {}
        for kind, e0, e1 in evs:
            d_ = per.setdefault(kind, [0, 0.0])
            d_[0] += 1
            d_[1] += e0.elapsed_time(e1)
        comm = {k: {'calls_per_step': v[0] // conc_steps, 'ms_per_step': round(v[1] / conc_steps, 3)} for k, v in per.items()}
        comm['note'] = ('HIP events on the stream each exchange is issued on, rank 0: generator / discriminator flat-gradient '
                        'all-reduce (issue .. completion seen by the consumer stream: overlapped with the discriminator step), '
                        'SyncBatchNorm statistic exchanges, scp gradient triple')
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    bad = [k for k, v in out.items() if hasattr(v, 'item') and not torch.isfinite(v).all()]
    if bad:
        sys.exit(f'bench.py: non-finite loss terms after the timed steps: {bad}')
    conc = _lib.TIMER.summary()          # concurrent-order pass: weight-gradient and discriminator streams on -> launches overlap
    # Per-kernel attribution: with the streams on a launch's elapsed time is not the kernel's own (kernels of three streams share
    # the machine).  The roofline object is therefore measured on 2 extra steps in SERIAL stream order right after the timed
    # region (same process, same buffers, HIP events per launch); `value` / `ms_per_step` above are the timed region's.
    from speech_enhancement_amd import gemm as _GM
    _saved = (_GM._LeafStream.enabled, TR._D_OVERLAP, _GM.branch_stream.enabled)
    a_steps = a.steps
    serial_steps = 2 if (world == 1 and not force_dp) else 0      # N > 1: the other ranks have left; attribution from the timed region
    if not any(_saved):
        serial_steps = 0                                           # SE_NO_*_STREAM / SE_NO_D_OVERLAP: the timed region is serial already
    a = argparse.Namespace(**{**vars(a), 'steps': conc_steps})          # per-step figures below: per attribution pass
    if serial_steps:
        try:
            _GM._LeafStream.enabled, TR._D_OVERLAP, _GM.branch_stream.enabled = False, False, False
            step()
            torch.cuda.synchronize()
            _lib.TIMER.start()
            for _ in range(serial_steps):
                step()
            torch.cuda.synchronize()
            _lib.TIMER.stop()
        finally:
            _GM._LeafStream.enabled, TR._D_OVERLAP, _GM.branch_stream.enabled = _saved
        summ = _lib.TIMER.summary()
        a = argparse.Namespace(**{**vars(a), 'steps': serial_steps})  # per-step figures of the roofline object: serial pass
    else:
        summ = conc
    dom = max(summ.items(), key=lambda kv: kv[1]['ms']) if summ else None
    roof = None
    if dom is not None:
        k, v = dom
        traffic = mfma_busy = traffic_source = None
        pmc = {}
        tpath = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles', 'pmc_traffic.json')
        if os.path.exists(tpath):      # PMC passes are separate rocprofv3 runs (tools/profile_round.sh); per launch
            pj = json.load(open(tpath))
            pmc = pj.get('kernels', {})
            ent = pmc.get(k, {})
            traffic, mfma_busy = ent.get('traffic_bytes_per_launch'), ent.get('mfma_busy_pct')
            # NOT measured in this run: read from the committed profile of the builder's profiling box (rocprofv3 --pmc cannot
            # run inside this process)
            traffic_source = 'profiles/pmc_traffic.json <- ' + str(pj.get('source', 'profiles/ (separate rocprofv3 --pmc passes)'))
        if k.startswith('gemm_k64_panel') or k.startswith('dwconv'):
            ach = v['bytes'] / (v['ms'] * 1e-3) / 1e9
            roof = {'bound': 'hbm', 'kernel': k, 'achieved': round(ach, 1), 'peak': 8000.0, 'unit': 'GB/s',
                    'frac': round(ach / 8000.0, 4), 'traffic': traffic, 'traffic_source': traffic_source,
                    'note': 'algorithmic bytes (operands read once + result written once) / launch time; peak = HBM3E spec'}
        else:
            ach = v['flops'] / (v['ms'] * 1e-3) / 1e12
            if k.startswith('attn_'):
                # attention: algorithmic fp32 FLOPs of QK^T, q.E, AV (+ backward) against the fp32-MFMA peak, the bar
                # BASELINE.json's north_star names for this path (the kernels execute each 16-deep product as three 16x16x32
                # bf16 MFMAs on an exact three-way operand split: 6x the algorithmic FLOPs on the bf16 pipe)
                peak = PEAK_F32_MFMA_TFLOPS
                xparts, xkind = (3, 'three 16x16x16 fp16 MFMAs per 16-deep product on scaled (hi, lo) fp16 operands') if 'f16x3' in k else \
                                (6, 'three 16x16x32 bf16 MFMAs per 16-deep product on an exact three-way operand split')
                note = ('algorithmic fp32 FLOPs of the attention contractions (forward: QK^T, q.E, AV; backward: their nine '
                        'products) / family launch time (incl. the delta / table / dE-reduce helpers); peak = dense fp32 MFMA '
                        f'(north_star bar); executed as {xkind}: {round(ach * xparts, 1)} TFLOP/s on the 2.5 PFLOP/s 16-bit pipe')
            elif 'bf16x6' in k or 'bf16x3' in k or 'f16x3' in k:
                parts = 6 if 'bf16x6' in k else 3
                kind = 'scaled fp16 hi/lo' if 'f16x3' in k and 'bf16' not in k else 'bf16 hi/mid/lo'
                peak = PEAK_BF16_MFMA_TFLOPS / parts
                note = (f'algorithmic (fp32-equivalent) FLOPs; the kernel evaluates every product as {parts} 16-bit MFMAs '
                        f'({kind} operand split, fp32 accumulate), so its peak is the dense 16-bit MFMA peak / '
                        f'{parts} = {round(peak, 1)} TFLOP/s; executed MFMA rate = {round(ach * parts, 1)} TFLOP/s.  '
                        f'The peak is quoted at the spec clock: under this load on random operands the chip holds a lower one '
                        f'(the same launch runs 25 % faster on all-zero activations, DESIGN.md appendix)')
            else:
                peak = PEAK_F32_MFMA_TFLOPS
                note = 'algorithmic fp32 FLOPs; peak = dense fp32 MFMA (v_mfma_f32_*)'
            roof = {'bound': 'mfma', 'kernel': k, 'achieved': round(ach, 2), 'peak': round(peak, 1), 'unit': 'TFLOP/s',
                    'frac': round(ach / peak, 4), 'traffic': traffic, 'mfma_busy_pct_pmc': mfma_busy,
                    'traffic_source': traffic_source, 'note': note}
            if _pipe16_parts(k):
                roof['frac_of_16bit_pipe'] = round(ach * _pipe16_parts(k) / PEAK_BF16_MFMA_TFLOPS, 4)
        cv = conc.get(k)
        # the two bars BASELINE.json's north_star names: attention vs the fp32-MFMA peak, depthwise conv vs the HBM peak
        secondary = []
        for kk, vv in sorted(summ.items(), key=lambda kv: -kv[1]['ms']):
            if kk.startswith('attn_') and vv['flops'] > 0:
                ach2 = vv['flops'] / (vv['ms'] * 1e-3) / 1e12
                secondary.append({'bound': 'mfma', 'kernel': kk, 'achieved': round(ach2, 2), 'peak': PEAK_F32_MFMA_TFLOPS,
                                  'unit': 'TFLOP/s', 'frac': round(ach2 / PEAK_F32_MFMA_TFLOPS, 4),
                                  'launches_per_step': vv['launches'] // a.steps,
                                  'avg_launch_ms': round(vv['ms'] / vv['launches'], 4),
                                  'traffic': pmc.get(kk, {}).get('traffic_bytes_per_launch'), 'traffic_source': traffic_source,
                                  'frac_of_16bit_pipe': round(ach2 * _pipe16_parts(kk) / PEAK_BF16_MFMA_TFLOPS, 4) if _pipe16_parts(kk) else None,
                                  'mfma_busy_pct_pmc': pmc.get(kk, {}).get('mfma_busy_pct'),
                                  'note': 'algorithmic fp32 FLOPs of QK^T, q.E, AV (+ their backward) / family launch time; '
                                          'peak = dense fp32 MFMA (the bar north_star names); frac_of_16bit_pipe = the same work priced '
                                          'like conv3: executed 16-bit MFMAs (3 per product) / 2.5 PFLOP/s'})
            elif kk.startswith('conv3_') and vv['flops'] > 0:
                parts2 = 6 if 'bf16x6' in kk else 3
                ach2 = vv['flops'] / (vv['ms'] * 1e-3) / 1e12
                secondary.append({'bound': 'mfma', 'kernel': kk, 'achieved': round(ach2, 2), 'peak': round(PEAK_BF16_MFMA_TFLOPS / parts2, 1),
                                  'unit': 'TFLOP/s', 'frac': round(ach2 * parts2 / PEAK_BF16_MFMA_TFLOPS, 4),
                                  'frac_of_16bit_pipe': round(ach2 * parts2 / PEAK_BF16_MFMA_TFLOPS, 4),
                                  'launches_per_step': vv['launches'] // a.steps, 'avg_launch_ms': round(vv['ms'] / vv['launches'], 4),
                                  'traffic': pmc.get(kk, {}).get('traffic_bytes_per_launch'), 'traffic_source': traffic_source,
                                  'mfma_busy_pct_pmc': pmc.get(kk, {}).get('mfma_busy_pct'),
                                  'note': f'triple-tap convolutions (dominant kernel of rounds 1 - 2): fp32-equivalent FLOPs, {parts2} 16-bit '
                                          f'MFMAs per product; peak = 2500 / {parts2} TFLOP/s'})
            elif kk.startswith('dwconv31') and vv['bytes'] > 0:
                ach2 = vv['bytes'] / (vv['ms'] * 1e-3) / 1e9
                secondary.append({'bound': 'hbm', 'kernel': kk, 'achieved': round(ach2, 1), 'peak': PEAK_HBM_GBS, 'unit': 'GB/s',
                                  'frac': round(ach2 / PEAK_HBM_GBS, 4), 'launches_per_step': vv['launches'] // a.steps,
                                  'avg_launch_ms': round(vv['ms'] / vv['launches'], 4),
                                  'traffic': pmc.get(kk, {}).get('traffic_bytes_per_launch'), 'traffic_source': traffic_source,
                                  'note': 'algorithmic bytes (input read once + output written once) / launch time; peak = HBM3E spec'})
        roof['secondary'] = secondary
        roof.update({'launches_per_step': v['launches'] // a.steps, 'avg_launch_ms': round(v['ms'] / v['launches'], 4),
                     'avg_launch_ms_concurrent_order': round(cv['ms'] / cv['launches'], 4) if cv else None,
                     'measured_on': (f'{serial_steps} extra steps in serial stream order after the timed region and after a {conc_steps}-step '
                                     f'instrumented pass in concurrent order (the headline loop itself runs un-instrumented; in '
                                     f'concurrent order the discriminator and weight-gradient streams overlap with this kernel: '
                                     f'elapsed time per launch is then not the kernel\'s own)') if serial_steps else
                                    (f'{conc_steps} instrumented steps after the timed region (serial stream order)' if not any(_saved) else
                                     f'{conc_steps} instrumented steps after the timed region (launches of the three streams overlap: '
                                     f'elapsed time per launch is not the kernel\'s own)'),
                     'share_of_step_time': round(v['ms'] / a.steps / (dt / a_steps * 1e3), 3),
                     'families': {kk: _family_entry(kk, vv, a.steps, pmc) for kk, vv in sorted(summ.items(), key=lambda kv: -kv[1]['ms'])}})
        # step-level figures of the committed profile (separate rocprofv3 passes in serial stream order; NOT measured in this run)
        if os.path.exists(tpath) and pj.get('step'):
            roof['step_profile'] = dict(pj['step'], source=traffic_source,
                                        note='time-weighted MFMA-busy over every kernel of a step and HBM-side bytes per step '
                                             '(2 FETCH_SIZE + WRITE_SIZE over all launches), from the committed PMC passes')
    a = argparse.Namespace(**{**vars(a), 'steps': a_steps})
    res = {
        'metric': 'utterances/sec (2 s @16 kHz) CMGAN train step', 'value': round(world * B * a.steps / dt, 3),
        'unit': 'utterances/sec', 'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup,
        'ms_per_step': round(dt / a.steps * 1e3, 2), 'higher_is_better': True, 'scaling': 'weak',
        'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': f'{a.arch} generator+discriminator train step (main_gan.py train_gan loop body), '
                               f'batch {B}/GPU, 2 s @ 16 kHz, n_fft=400 hop=100, AdamW lr 5e-4, PESQ labels supplied, '
                               f'kaiming-init weights; fp32 results throughout: conv and token GEMMs as scaled fp16 hi/lo splits '
                               f'(3 MFMAs per product, 2^-24 relative, fp32 accumulate), attention on the same scaled fp16 splits (scales from measured operand maxima), '
                               f'everything else fp32 MFMA / fp32 VALU',
                   'global_batch': world * B, 'parallelism': f'dp{world}',
                   'effective_tflops': round(world * B * a.steps * GFLOP_PER_UTT_STEP / dt / 1e3, 2),
                   'dropout': 'generator ff/attn dropout p=0.2 on (counter-based masks in the GEMM pro/epilogues); '
                              'discriminator Dropout(0.3) on'},
        'losses': {k: round(float(v), 5) for k, v in out.items() if hasattr(v, 'item') or isinstance(v, float)},
        'roofline': roof,
    }
    if comm is not None:
        res['comm'] = comm
    if cpu_res is not None:
        res['cpu_baseline'] = cpu_res
    print(json.dumps(res))
    if world > 1 or force_dp:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
