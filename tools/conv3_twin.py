"""Limiter of the triple-tap convolution kernel from its measurement twins.

Three builds of the library run conv3_bf16_kernel<2, true, true, 4, 1, *> (the 13-launch instantiation of a train step: scaled fp16,
128-row tiles, 4 waves per SIMD) at the dilated dense layers' shapes of the encoder (B = 16, T = 321, F = 201, Cin = 64 / 128 and, with
the 256-row tile instantiation, 192 / 256), each in its own process, interleaved rounds, one box:
  product : libse_hip.so
  mfma    : -DSE_CONV3_TWIN=1  LDS fragment reads + MFMAs + barriers + epilogue; no global load, no split, no staging store
  feed    : -DSE_CONV3_TWIN=2  everything but the MFMAs (each replaced by two value-preserving v_fma_f32)
usage:  tools/conv3_twin.py build | (no argument: run, writes profiles/r06_conv3_twin.json) | child"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIBS = {'product': None, 'mfma': os.path.join(ROOT, 'tools', 'micro', 'bin', 'libse_conv3_twin1.so'),
        'feed': os.path.join(ROOT, 'tools', 'micro', 'bin', 'libse_conv3_twin2.so')}


def child():
    import torch
    from speech_enhancement_amd import gemm as GM, layers as LY
    from speech_enhancement_amd.weights import WeightPlan
    B, T, F = 16, 321, 201
    out = {}
    g = torch.Generator().manual_seed(0)
    skip = torch.randn(B, T, F, 256, generator=g).cuda()
    for layer in (0, 1, 2, 3):
        C = 64 * (layer + 1)
        w = (torch.randn(64, C, 2, 3, generator=g) * (6 * C) ** -0.5).cuda()
        plan = WeightPlan(torch.device('cuda'))
        wp = plan.conv_fwd('w', w, planes='f16') if hasattr(plan, 'conv_fwd') else None
        if wp is None:
            raise SystemExit('WeightPlan has no conv_fwd')
        plan.run()
        taps = LY.dense_taps(layer)
        amax = skip[..., :C].abs().max().reshape(1).clone()
        y = torch.empty(B, T, F, 64, device='cuda')
        d = GM.make_desc(B, T, F, T, F, taps, C, 256, 64, 64, precision=3, a_amax=amax)
        f = lambda: GM.gemm_tap(d, skip, wp, y)
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(11)]
        ev[0].record()
        for i in range(10):
            f()
            ev[i + 1].record()
        torch.cuda.synchronize()
        out[f'Cin{C}'] = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(10))[5]
    print(json.dumps(out), flush=True)


def build():
    sh = os.path.join(ROOT, 'tools', 'build_variant_lib.sh')
    for t in (1, 2):
        subprocess.check_call(['bash', sh, f'conv3_twin{t}', f'-DSE_CONV3_TWIN={t}', 'se_gemm.hip'])


def main():
    res = {k: [] for k in LIBS}
    for _ in range(3):
        for k, lib in LIBS.items():
            env = dict(os.environ)
            if lib:
                env['SE_HIP_LIB'] = lib
            else:
                env.pop('SE_HIP_LIB', None)
            r = subprocess.run([sys.executable, os.path.abspath(__file__), 'child'], env=env, capture_output=True, text=True)
            if r.returncode:
                sys.exit(r.stderr[-2000:])
            res[k].append(json.loads(r.stdout.strip().split('\n')[-1]))
    med = lambda v: sorted(v)[len(v) // 2]
    table = {}
    B, T, F = 16, 321, 201
    for key in res['product'][0]:
        C = int(key[3:])
        p, m, v = (med([r[key] for r in res[k]]) for k in ('product', 'mfma', 'feed'))
        fl = 2.0 * B * T * F * 64 * 6 * C
        table[key] = {'product_ms': round(p, 4), 'mfma_twin_ms (LDS fragment reads + MFMAs + barriers)': round(m, 4),
                      'feed_twin_ms (loads + splits + staging + fragment reads + barriers, no MFMA)': round(v, 4),
                      'product_over_max_twin': round(p / max(m, v), 3),
                      'product_tflops_fp32_equivalent': round(fl / p / 1e9, 1), 'mfma_twin_tflops': round(fl / m / 1e9, 1),
                      'mfma_issue_floor_ms (3 MFMAs per product at 2.5 PFLOP/s)': round(3 * fl / 2.5e15 * 1e3, 4),
                      'rounds': {k: [round(r[key], 4) for r in res[k]] for k in res}}
    out = {'what': 'conv3_bf16_kernel (scaled fp16, forward tap order) on the encoder\'s dilated dense layers, B = 16, T = 321, F = 201, N = 64; product vs '
                   'MFMA-only twin vs feed-only twin; one process per library, interleaved rounds, HIP-event median of 10 launches', 'layers': table}
    json.dump(out, open(os.path.join(ROOT, 'profiles', 'r06_conv3_twin.json'), 'w'), indent=1)
    print(json.dumps(out))


if __name__ == '__main__':
    if sys.argv[1:] == ['child']:
        child()
    elif sys.argv[1:] == ['build']:
        build()
    else:
        main()
