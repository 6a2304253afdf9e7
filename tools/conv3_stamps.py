"""Where a conv3_bf16_kernel workgroup spends its time: s_memtime stamps (diagnostic build -DSE_CONV3_STAMPS) of workgroups 4096 .. 4159
(steady state of the launch), every wave, the first five (channel chunk, dt) groups.  Per group: top -> staged (wait for the A tile's
loads + split + LDS stores), then per tap: barrier 1 | prefetch issue + fragment reads + MFMAs | barrier 2.
usage: tools/conv3_stamps.py build | run (GPU box)"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, 'tools', 'micro', 'bin', 'libse_conv3_stamps.so')

if sys.argv[1:] == ['build']:
    subprocess.check_call(['bash', os.path.join(ROOT, 'tools', 'build_variant_lib.sh'), 'conv3_stamps', '-DSE_CONV3_STAMPS', 'se_gemm.hip'])
    sys.exit(0)
os.environ['SE_HIP_LIB'] = LIB
import torch  # noqa: E402
from speech_enhancement_amd import gemm as GM, layers as LY, _lib as L  # noqa: E402
from speech_enhancement_amd.weights import WeightPlan  # noqa: E402

B, T, F = 16, 321, 201
g = torch.Generator().manual_seed(0)
skip = torch.randn(B, T, F, 256, generator=g).cuda()
stamps = torch.zeros(64 * 4 * 64, dtype=torch.int32, device='cuda')
L.lib().se_conv3_debug_stamps(C.c_void_p(stamps.data_ptr()))
for name, layer, dgrad in (('fwd Cin64', 0, False), ('fwd Cin128', 1, False), ('fwd Cin256', 3, False), ('dgrad K64 N256', 3, True)):
    C_in = 64 * (layer + 1)
    w = (torch.randn(64, C_in, 2, 3, generator=g) * (6 * C_in) ** -0.5).cuda()
    plan = WeightPlan(torch.device('cuda'))
    taps = LY.dense_taps(layer)
    if dgrad:
        wp = plan.conv_dgrad('w', w, planes='f16')
        plan.run()
        dy = torch.randn(B, T, F, 64, generator=g).cuda() * 1e-3
        dx = torch.zeros(B, T, F, 256, device='cuda')
        d = GM.make_desc(B, T, F, T, F, [(-a, -c) for a, c in taps], 64, 64, C_in, 256, precision=3, a_amax=dy.abs().max().reshape(1).clone(),
                         epilogue=L.EPI_ACCUM)
        f = lambda: GM.gemm_tap(d, dy, wp, dx)
    else:
        wp = plan.conv_fwd('w', w, planes='f16')
        plan.run()
        y = torch.empty(B, T, F, 64, device='cuda')
        d = GM.make_desc(B, T, F, T, F, taps, C_in, 256, 64, 64, precision=3, a_amax=skip[..., :C_in].abs().max().reshape(1).clone())
        f = lambda: GM.gemm_tap(d, skip, wp, y)
    for _ in range(3):
        f()
    stamps.zero_()
    torch.cuda.synchronize()
    f()
    torch.cuda.synchronize()
    st = stamps.view(64, 4, 64).cpu().numpy().astype('int64') & 0xffffffff
    ngrp = min(5, (C_in if not dgrad else 64) // 32 * 2)
    print(f'== {name}: groups stamped {ngrp}')
    import numpy as np
    ok = st[:, :, 0] != 0
    d_ = lambda a, b: float(np.mean(((st[:, :, b] - st[:, :, a]) & 0xffffffff)[ok]))
    for gq in range(ngrp):
        base = 1 + 11 * gq
        row = [f'stage(wait A + split + store) {d_(base, base + 1):7.0f}']
        for s3 in range(3):
            row.append(f'tap{s3}: bar1 {d_(base + 1 + 3 * s3 if s3 == 0 else base + 4 + 3 * (s3 - 1), base + 2 + 3 * s3):6.0f} mfma {d_(base + 2 + 3 * s3, base + 3 + 3 * s3):6.0f} bar2 {d_(base + 3 + 3 * s3, base + 4 + 3 * s3):6.0f}')
        tot = d_(base, base + 10)
        print(f'  group {gq}: ' + ' | '.join(row) + f' | group total {tot:7.0f} cycles')
    if ngrp >= 2:
        print(f'  prologue (entry stamp -> first group top): {d_(0, 1):7.0f}')
