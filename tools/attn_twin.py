"""Ceiling of the scaled-fp16 attention kernels from their measurement twins.

Three builds of the same library run the same launches at the bench shapes (B = 16: time axis 1616 sequences x 321 positions,
frequency axis 5136 x 101), each in its own process, interleaved over ROUNDS rounds on ONE box:
  product : libse_hip.so
  mfma    : -DSE_ATTN_TWIN=1  MFMAs + global / LDS traffic + barriers, the vector chain between the products compiled out
  valu    : -DSE_ATTN_TWIN=2  the vector chain + traffic, every MFMA replaced by two value-preserving v_fma_f32
(csrc/se_attn.hip: what exactly each twin keeps).  The twins compute garbage; only their launch times are read.
`ceiling_ms` = max(mfma, valu): the time of a kernel whose two pipes overlap perfectly.  The traffic is in both twins, so
(mfma + valu - product) / min(mfma, valu) is a LOWER bound of the share of the shorter pipe's time that the product already hides.

usage:  tools/attn_twin.py build      (in the build container: tools/micro/bin/libse_attn_twin{1,2}.so)
        tools/attn_twin.py            (on the GPU box: writes profiles/r06_attn_twin.json)
        tools/attn_twin.py child      (one library, selected by SE_HIP_LIB: prints one JSON line)
        tools/attn_twin.py padding    (product library: what the cells of the padded last 16-tile cost -> profiles/r06_attn_padding.json)"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIBS = {'product': None, 'mfma': os.path.join(ROOT, 'tools', 'micro', 'bin', 'libse_attn_twin1.so'),
        'valu': os.path.join(ROOT, 'tools', 'micro', 'bin', 'libse_attn_twin2.so')}
ROUNDS = 3


def child():
    import torch
    from speech_enhancement_amd import attention as A
    from speech_enhancement_amd.weights import WeightPlan
    B, T, Fq = 16, 321, 101
    g = torch.Generator().manual_seed(0)
    qkv = (torch.randn(B * T * Fq, 192, generator=g) * 1.5).cuda()
    E = (torch.randn(1025, 16, generator=g) * 0.5).cuda()
    dO = (torch.randn(B * T * Fq, 64, generator=g) * 1e-3).cuda()
    plan = WeightPlan(torch.device('cuda'))
    Es = plan.linear('e', E, planes='f16')
    plan.run()
    am, dam = qkv.abs().max().reshape(1).clone(), dO.abs().max().reshape(1).clone()
    out = {}

    def timed(f, n=10):
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
        ev[0].record()
        for i in range(n):
            f()
            ev[i + 1].record()
        torch.cuda.synchronize()
        ts = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(n))
        return ts[n // 2]
    for axis in ('time', 'freq'):
        geom = A.seq_geometry(B, T, Fq, axis)
        O, lse = A.attn_fwd(qkv, E, geom, Es=Es, qkv_amax=am)
        delta = torch.zeros(qkv.shape[0], 4, device='cuda')      # (the table the to_out GEMM's epilogue writes in the step)
        dE = torch.zeros_like(E)
        out[f'fwd_{axis}'] = timed(lambda: A.attn_fwd(qkv, E, geom, Es=Es, qkv_amax=am))
        out[f'bwd_{axis}'] = timed(lambda: A.attn_bwd(qkv, E, O, dO, lse, geom, dE, qkv_amax=am, do_amax=dam, delta=delta))
    print(json.dumps(out), flush=True)


def padding():
    """the same kernels at sequence lengths around the model's (101 = 6 tiles + 5, 321 = 20 tiles + 1): a length costs what its
    ceil(n / 16) tiles cost -- the time of n = 101 against n = 96 (no padded tile) and n = 112 (the padded tile full)"""
    import torch
    from speech_enhancement_amd import attention as A
    from speech_enhancement_amd.weights import WeightPlan
    B, out = 16, {}
    g = torch.Generator().manual_seed(0)
    E = (torch.randn(1025, 16, generator=g) * 0.5).cuda()
    plan = WeightPlan(torch.device('cuda'))
    Es = plan.linear('e', E, planes='f16')
    plan.run()
    for axis, lens, other in (('freq', (96, 101, 112), 321), ('time', (320, 321, 336), 101)):
        for n in lens:
            T, Fq = (other, n) if axis == 'freq' else (n, other)
            qkv = (torch.randn(B * T * Fq, 192, generator=g) * 1.5).cuda()
            dO = (torch.randn(B * T * Fq, 64, generator=g) * 1e-3).cuda()
            am, dam = qkv.abs().max().reshape(1).clone(), dO.abs().max().reshape(1).clone()
            geom = A.seq_geometry(B, T, Fq, axis)
            O, lse = A.attn_fwd(qkv, E, geom, Es=Es, qkv_amax=am)
            delta, dE = torch.zeros(qkv.shape[0], 4, device='cuda'), torch.zeros_like(E)
            ts = {}
            for name, f in (('fwd', lambda: A.attn_fwd(qkv, E, geom, Es=Es, qkv_amax=am)),
                            ('bwd', lambda: A.attn_bwd(qkv, E, O, dO, lse, geom, dE, qkv_amax=am, do_amax=dam, delta=delta))):
                for _ in range(3):
                    f()
                torch.cuda.synchronize()
                ev = [torch.cuda.Event(enable_timing=True) for _ in range(11)]
                ev[0].record()
                for i in range(10):
                    f()
                    ev[i + 1].record()
                torch.cuda.synchronize()
                ts[name + '_ms'] = round(sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(10))[5], 4)
            ts.update({'tiles': (n + 15) // 16, 'cells_computed_over_cells_needed': round(((n + 15) // 16 * 16) ** 2 / n ** 2, 3)})
            out[f'{axis}_n{n}'] = ts
            del qkv, dO, O, lse
    res = {'what': 'scaled-fp16 attention at B = 16 with the sequence length varied around the model\'s: the kernels compute whole 16 x 16 tiles, '
                   'so n = 101 (7 tiles, 5 rows in the last) costs what n = 112 costs; rows of different sequences have different keys, so the '
                   'tail rows of several sequences cannot share one MFMA tile (the unit of work is a (query tile, key tile) pair of ONE sequence)',
           'shapes': out}
    json.dump(res, open(os.path.join(ROOT, 'profiles', 'r06_attn_padding.json'), 'w'), indent=1)
    print(json.dumps(res))


def build():
    sh = os.path.join(ROOT, 'tools', 'build_variant_lib.sh')
    for t in (1, 2):
        subprocess.check_call(['bash', sh, f'attn_twin{t}', f'-DSE_ATTN_TWIN={t}', 'se_attn.hip'])


def main():
    res = {k: [] for k in LIBS}
    for _ in range(ROUNDS):
        for k, lib in LIBS.items():
            env = dict(os.environ)
            if lib:
                if not os.path.exists(lib):
                    sys.exit(f'{lib} missing: run `python tools/attn_twin.py build` in the build container first')
                env['SE_HIP_LIB'] = lib
            else:
                env.pop('SE_HIP_LIB', None)
            line = subprocess.run([sys.executable, os.path.abspath(__file__), 'child'], env=env, capture_output=True, text=True, check=True).stdout
            res[k].append(json.loads(line.strip().split('\n')[-1]))
    med = lambda v: sorted(v)[len(v) // 2]
    table = {}
    for key, (nseq, n) in {'fwd_time': (1616, 321), 'bwd_time': (1616, 321), 'fwd_freq': (5136, 101), 'bwd_freq': (5136, 101)}.items():
        p, m, v = (med([r[key] for r in res[k]]) for k in ('product', 'mfma', 'valu'))
        fl = nseq * 4 * (3 if key.startswith('fwd') else 7) * 2.0 * n * n * 16
        table[key] = {'sequences': nseq, 'n': n, 'product_ms': round(p, 4), 'mfma_twin_ms': round(m, 4), 'valu_twin_ms': round(v, 4),
                      'ceiling_ms (max of the twins: perfect overlap of the two pipes)': round(max(m, v), 4),
                      'product_over_ceiling': round(p / max(m, v), 3),
                      'overlap_lower_bound ((mfma + valu - product) / min(mfma, valu))': round((m + v - p) / min(m, v), 3),
                      'product_algorithmic_tflops': round(fl / p / 1e9, 1), 'ceiling_algorithmic_tflops': round(fl / max(m, v) / 1e9, 1),
                      'rounds': {k: [round(r[key], 4) for r in res[k]] for k in res}}
    out = {'what': 'scaled-fp16 attention kernels (attn_fwd3_kernel<TQ, true>, attn_bwd4_kernel exact bodies + tables + dE fold) at the bench shapes, '
                   'B = 16; product vs MFMA-only twin vs VALU-only twin; one process per library, interleaved rounds, HIP-event median of 10 launches',
           'kernels': table}
    path = os.path.join(ROOT, 'profiles', 'r06_attn_twin.json')
    json.dump(out, open(path, 'w'), indent=1)
    print(json.dumps(out))


if __name__ == '__main__':
    if sys.argv[1:] == ['child']:
        child()
    elif sys.argv[1:] == ['build']:
        build()
    elif sys.argv[1:] == ['padding']:
        padding()
    else:
        main()
