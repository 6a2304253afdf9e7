"""What a producer-side reduction could save in the InstanceNorm + PReLU backward: the reduce pass of
se_norm_prelu_bwd (reads x and dY, 2 planes) against its twin WITHOUT the dY stream (-DSE_NORM_REDUCE_TWIN: what is left of the pass when
the kernel that produced dY -- a conv3 input gradient -- had emitted sum(du), sum(du x-hat), sum(dy min(y, 0)) from its epilogue; that
epilogue would have to read x, the twin's remaining stream, itself) at the three plane shapes of a train step, standalone on one box.
usage: tools/inorm_bwd_ab.py build | (no argument: run -> profiles/r06_inorm_reduce_twin.json) | child"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, 'tools', 'micro', 'bin', 'libse_norm_twin.so')
SHAPES = {'dense block encoder [16, 321*201, 64] (5 sites)': (16, 321 * 201, 64, 5), 'dense block decoders [16, 321*101, 64] (9 sites)': (16, 321 * 101, 64, 9),
          'sub-pixel output [16, 321*202, 64] (1 site)': (16, 321 * 202, 64, 1)}


def child():
    import torch
    from speech_enhancement_amd import ops as O, _lib as L
    out = {}
    for name, (B, P, C, _) in SHAPES.items():
        x, dy = torch.randn(B, P, C, device='cuda'), torch.randn(B, P, C, device='cuda')
        g, be, a = torch.rand(C, device='cuda') + 0.5, torch.randn(C, device='cuda') * 0.1, torch.full((C,), 0.25, device='cuda')
        stats = O.col_stats(x, C, 0, B, P, C)
        mr, _ = O.norm_finalize(stats, g, be, B, C, float(P))
        dx = torch.empty_like(x)
        dg, db, da = (torch.zeros(C, device='cuda') for _ in range(3))
        red = torch.zeros(L.lib().se_norm_prelu_bwd_workspace_bytes(O._i(B), O._i(C), O._i(1)) // 8, device='cuda', dtype=torch.float64)
        res = {}
        for label, phase in (('reduce', 1 | 16), ('apply', 2 | 8)):
            def f():
                L.call('se_norm_prelu_bwd_amax', L.ptr(x), O._i(C), O._i(0), L.ptr(mr), L.ptr(g), L.ptr(be), L.ptr(a), L.ptr(dy), O._i(C), O._i(0),
                       L.ptr(red), L.ptr(dx), O._i(C), O._i(0), L.ptr(dg), L.ptr(db), L.ptr(da), O._i(B), O._l(P), O._i(C), O._i(1), O._i(0),
                       O._i(phase), O._d(float(P)), L.ptr(None), L.stream())
            for _ in range(3):
                f()
            torch.cuda.synchronize()
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(21)]
            ev[0].record()
            for i in range(20):
                f()
                ev[i + 1].record()
            torch.cuda.synchronize()
            res[label] = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(20))[10] * 1e3
        out[name] = res
        del x, dy, dx
    print(json.dumps(out), flush=True)


if sys.argv[1:] == ['build']:
    subprocess.check_call(['bash', os.path.join(ROOT, 'tools', 'build_variant_lib.sh'), 'norm_twin', '-DSE_NORM_REDUCE_TWIN', 'se_norms.hip'])
elif sys.argv[1:] == ['child']:
    child()
else:
    res = {'product': [], 'twin': []}
    for _ in range(3):
        for k in res:
            env = dict(os.environ)
            if k == 'twin':
                env['SE_HIP_LIB'] = LIB
            else:
                env.pop('SE_HIP_LIB', None)
            r = subprocess.run([sys.executable, os.path.abspath(__file__), 'child'], env=env, capture_output=True, text=True)
            if r.returncode:
                sys.exit(r.stderr[-2000:])
            res[k].append(json.loads(r.stdout.strip().split('\n')[-1]))
    med = lambda v: sorted(v)[len(v) // 2]
    table, total = {}, 0.0
    for name, (B, P, C, sites) in SHAPES.items():
        pr = med([r[name]['reduce'] for r in res['product']])
        tw = med([r[name]['reduce'] for r in res['twin']])
        ap = med([r[name]['apply'] for r in res['product']])
        # sites per step: every InstanceNorm of the generator runs its backward once; the ones whose dY is last written by a conv3 input
        # gradient are the dense-block layers 0 .. 2 of each block (3 of 4 per block) -- the upper bound below counts ALL sites of the shape
        table[name] = {'reduce_us': round(pr, 1), 'reduce_without_dY_us': round(tw, 1), 'apply_us': round(ap, 1), 'saving_upper_bound_us_per_site': round(pr - tw, 1),
                       'sites_per_step': sites, 'gbs_reduce': round(2 * 4.0 * B * P * C / pr / 1e3, 1)}
        total += (pr - tw) * sites
    out = {'what': 'se_norm_prelu_bwd reduce pass (x, dY -> three sums per (b, channel)) vs the same pass without its dY stream, B = 16, standalone, '
                   'median of 20 launches, one process per library, three interleaved rounds', 'shapes': table,
           'saving_upper_bound_ms_per_step (all 15 sites of these shapes; only 9 of them take their dY from a conv3 input gradient)': round(total / 1e3, 3)}
    json.dump(out, open(os.path.join(ROOT, 'profiles', 'r06_inorm_reduce_twin.json'), 'w'), indent=1)
    print(json.dumps(out))
