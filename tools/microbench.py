"""Micro-benchmarks and one-kernel timers of the HIP path (run on the GPU box): `python tools/microbench.py <name> [args...]`.

Each sub-command is one of the small scripts the kernels were tuned with (round 1-3), kept verbatim as a function; `list` prints
the names.  They time with wall clock around a synchronised loop unless stated otherwise and print one line per case.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
CMDS = {}


def cmd(fn):
    CMDS[fn.__name__] = fn
    return fn


@cmd
def attn_bench(argv):
    __import__('sys').argv = ['attn_bench'] + list(argv)
    """micro-benchmark of the attention kernels at the bench shapes (B=16); SE_ATTN_BWD=2 selects the v2 backward."""
    import os, sys, time, torch
    from speech_enhancement_amd import attention as A
    B, T, Fq = 16, 321, 101
    g = torch.Generator().manual_seed(0)
    qkv = (torch.randn(B * T * Fq, 192, generator=g)).cuda()
    E = (torch.randn(1025, 16, generator=g) * 0.5).cuda()
    dO = torch.randn(B * T * Fq, 64, generator=g).cuda()
    for axis in ('time', 'freq'):
        geom = A.seq_geometry(B, T, Fq, axis)
        nseq, n = geom[0], geom[1]
        O, lse = A.attn_fwd(qkv, E, geom)
        for mode in sys.argv[1:] or ['3']:
            os.environ['SE_ATTN_BWD'] = mode.split(':')[0]
            os.environ['SE_ATTN_DBG'] = mode.split(':')[1] if ':' in mode else '0'
            dE = torch.zeros_like(E)
            for _ in range(2):
                A.attn_bwd(qkv, E, O, dO, lse, geom, dE)
            torch.cuda.synchronize()
            t0 = time.time()
            for _ in range(5):
                A.attn_bwd(qkv, E, O, dO, lse, geom, dE)
            torch.cuda.synchronize()
            tb = (time.time() - t0) / 5
            t0 = time.time()
            for _ in range(5):
                A.attn_fwd(qkv, E, geom)
            torch.cuda.synchronize()
            tf = (time.time() - t0) / 5
            fl = nseq * 4 * 2.0 * n * n * 16
            print(f'{axis} bwd-mode={mode}: bwd {tb*1e3:.3f} ms ({7 * fl / tb / 1e12:.1f} TFLOP/s algorithmic)  '
                  f'fwd {tf*1e3:.3f} ms ({3 * fl / tf / 1e12:.1f} TFLOP/s)', flush=True)
        # scaled split-fp16 kernels
        from speech_enhancement_amd.weights import WeightPlan
        plan = WeightPlan(torch.device('cuda')); Es = plan.linear('e', E, planes='f16'); plan.run()
        am = qkv.abs().max().reshape(1).clone()
        dO._se_amax = dO.abs().max().reshape(1).clone()
        for _ in range(2): A.attn_fwd(qkv, E, geom, Es=Es, qkv_amax=am)
        torch.cuda.synchronize(); t0 = time.time()
        for _ in range(5): A.attn_fwd(qkv, E, geom, Es=Es, qkv_amax=am)
        torch.cuda.synchronize(); tf = (time.time() - t0) / 5
        tb = float('nan')
        if True:
            dE = torch.zeros_like(E)
            f = lambda: A.attn_bwd(qkv, E, O, dO, lse, geom, dE, qkv_amax=am, do_amax=dO._se_amax)
            for _ in range(2): f()
            torch.cuda.synchronize(); t0 = time.time()
            for _ in range(5): f()
            torch.cuda.synchronize(); tb = (time.time() - t0) / 5
        print(f'{axis} f16x3: bwd {tb*1e3:.3f} ms  fwd {tf*1e3:.3f} ms ({3 * fl / tf / 1e12:.1f} TFLOP/s)', flush=True)

@cmd
def attn_fwd_es(argv):
    __import__('sys').argv = ['attn_fwd_es'] + list(argv)
    """attention forward with the relative-position table split on the fly vs pre-split (se_attn_fwd_es), B = 16 shapes"""
    import os, sys, time, torch
    from speech_enhancement_amd import attention as A
    from speech_enhancement_amd.weights import WeightPlan
    B, T, Fq = 16, 321, 101
    g = torch.Generator().manual_seed(0)
    qkv = torch.randn(B * T * Fq, 192, generator=g).cuda()
    E = (torch.randn(1025, 16, generator=g) * 0.5).cuda()
    plan = WeightPlan(torch.device('cuda'))
    Es = plan.linear('es', E, planes=True)
    plan.run()
    for axis in ('time', 'freq'):
        geom = A.seq_geometry(B, T, Fq, axis)
        outs = {}
        for name, es in (('on the fly', None), ('pre-split', Es), ('on the fly', None), ('pre-split', Es)):
            for _ in range(2): o, lse = A.attn_fwd(qkv, E, geom, Es=es)
            torch.cuda.synchronize(); t0 = time.time()
            for _ in range(10): o, lse = A.attn_fwd(qkv, E, geom, Es=es)
            torch.cuda.synchronize(); dt = (time.time() - t0) / 10
            outs[name] = (o, lse)
            print(f'{axis:5s} {name:11s} {dt*1e3:.3f} ms', flush=True)
        print('   bit-identical:', torch.equal(outs['on the fly'][0], outs['pre-split'][0]) and torch.equal(outs['on the fly'][1], outs['pre-split'][1]))

@cmd
def attn_prof(argv):
    __import__('sys').argv = ['attn_prof'] + list(argv)
    """one attention forward + backward per axis at the bench shapes (B=16), for rocprofv3 counter passes"""
    import os, sys, torch
    from speech_enhancement_amd import attention as A
    B, T, Fq = 16, 321, 101
    g = torch.Generator().manual_seed(0)
    qkv = (torch.randn(B * T * Fq, 192, generator=g)).cuda()
    E = (torch.randn(1025, 16, generator=g) * 0.5).cuda()
    dO = torch.randn(B * T * Fq, 64, generator=g).cuda()
    for axis in ('time', 'freq'):
        geom = A.seq_geometry(B, T, Fq, axis)
        for _ in range(2):
            O, lse = A.attn_fwd(qkv, E, geom)
            dE = torch.zeros_like(E)
            A.attn_bwd(qkv, E, O, dO, lse, geom, dE)
    torch.cuda.synchronize()

@cmd
def bw(argv):
    __import__('sys').argv = ['bw'] + list(argv)
    import torch, time
    M = 16 * 321 * 101
    for shape in ((M, 256), (M, 64)):
        y = torch.empty(*shape, device='cuda'); x = torch.randn(*shape, device='cuda')
        for name, f in (('fill', lambda: y.fill_(1.0)), ('copy', lambda: y.copy_(x)), ('add', lambda: torch.add(x, 1.0, out=y))):
            for _ in range(3): f()
            torch.cuda.synchronize(); t0 = time.time()
            for _ in range(20): f()
            torch.cuda.synchronize(); dt = (time.time() - t0) / 20
            nbytes = y.numel() * 4 * (1 if name == 'fill' else 2)
            print(shape, name, f'{dt*1e6:.1f} us', f'{nbytes/dt/1e9:.0f} GB/s')

@cmd
def conv_one(argv):
    __import__('sys').argv = ['conv_one'] + list(argv)
    import os, sys, torch
    from speech_enhancement_amd import gemm as GM, _lib as L, layers as LY
    B, T, Fq = 16, 321, 201
    Cin = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    skip = torch.randn(B, T, Fq, 256, device='cuda')
    if os.environ.get('ZERO') == '1':      # all-zero operands: the clock the chip holds under load depends on the data (DVFS)
        skip.zero_()
    w = torch.randn(64, Cin, 2, 3, device='cuda') * 0.02
    wp = GM.pack_conv_fwd(w)
    if len(sys.argv) > 3 and sys.argv[3] == 'planes':      # pre-split weights (weights.WeightPlan)
        from speech_enhancement_amd.weights import WeightPlan
        plan = WeightPlan(torch.device('cuda'))
        wp = plan.conv_fwd('w', w, planes='f16' if len(sys.argv) > 2 and sys.argv[2] == '3' else True)
        plan.run()
    y = torch.empty(B, T, Fq, 64, device='cuda')
    prec = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    d = GM.make_desc(B, T, Fq, T, Fq, LY.dense_taps(3), Cin, 256, 64, 64, precision=prec, a_sexp=4, w_sexp=8)
    import time
    for _ in range(3):
        GM.gemm_tap(d, skip, wp, y)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(5):
        GM.gemm_tap(d, skip, wp, y)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / 5
    if os.environ.get('CHECK') == '1':
        ref = torch.nn.functional.conv2d(torch.nn.functional.pad(skip[:1, :, :, :Cin].double().permute(0, 3, 1, 2), (1, 1, 8, 0)), w.double(), dilation=(8, 1)).permute(0, 2, 3, 1)
        err = (y[:1].double() - ref).abs().max().item()
        print('max err vs fp64', err, 'rel to max', err / ref.abs().max().item(), 'rms rel', ((y[:1].double() - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item())
    print(f'conv Cin={Cin} precision={prec}: {dt*1e6:.0f} us, {2.0*B*T*Fq*64*6*Cin/dt/1e12:.1f} TF')

@cmd
def cpu_overhead(argv):
    __import__('sys').argv = ['cpu_overhead'] + list(argv)
    """host-side enqueue time of one train step vs its GPU time"""
    import os, sys, time, types, torch
    import speech_enhancement_amd as S
    from speech_enhancement_amd import train as TR, optim as OP, _lib
    torch.manual_seed(0)
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    G, D = S.TSCNet(64, 201), S.Discriminator(16)
    G.apply(S.kaiming_init); D.apply(S.kaiming_init)
    G.cuda().train(); D.cuda().train()
    oargs = types.SimpleNamespace(optimizer='adamw', lr=5e-4, weight_decay=0.01, momentum=0.9, max_norm=0.0)
    og, od = OP.build_optimizer(oargs, G), OP.build_optimizer(oargs, D)
    clean = torch.randn(B, 32000, device='cuda') * 0.1; noisy = clean + 0.05 * torch.randn_like(clean)
    q = torch.rand(B, device='cuda')
    labels = {'est': q, 'clean': torch.full_like(q, 0.97), 'noisy': q * 0.5}
    w = (0.1, 0.9, 0.2, 0.05)
    for _ in range(3): TR.gan_step(G, D, og, od, clean, noisy, 'cmgan', w, labels=labels)
    torch.cuda.synchronize()
    enq = []; tot = []
    for _ in range(5):
        t0 = time.time()
        TR.gan_step(G, D, og, od, clean, noisy, 'cmgan', w, labels=labels)
        t1 = time.time()
        torch.cuda.synchronize()
        t2 = time.time()
        enq.append(t1 - t0); tot.append(t2 - t0)
    print(f'B={B}: enqueue {1e3*sum(enq)/5:.1f} ms, total {1e3*sum(tot)/5:.1f} ms')

@cmd
def dw_bench(argv):
    __import__('sys').argv = ['dw_bench'] + list(argv)
    import os, sys, time, torch
    from speech_enhancement_amd import ops as O, attention as A
    B, T, Fq = 16, 321, 101
    x = torch.randn(B * T * Fq, 128, device='cuda'); dy = torch.randn_like(x)
    w = torch.randn(128, 31, device='cuda') * 0.1; b = torch.randn(128, device='cuda')
    for axis in ('time', 'freq'):
        geom = A.seq_geometry(B, T, Fq, axis)
        st = torch.zeros(1, 128, 2, device='cuda', dtype=torch.float64)
        gate = torch.randn(B * T * Fq, 128, device='cuda')
        for name, fn in (('fwd+stats', lambda: O.dwconv31(x, w, b, geom, stats=st)), ('dgrad', lambda: O.dwconv31(dy, w, None, geom, flip=True)),
                         ('dgrad+glu', lambda: O.dwconv31_glu_bwd(dy, w, x, gate, geom)),
                         ('wgrad', lambda: O.dwconv31_wgrad(x, dy, torch.zeros(128, 31, device='cuda'), torch.zeros(128, device='cuda'), geom)),
                         ('bwd fused', lambda: O.dwconv31_bwd_fused(dy, w, x, gate, torch.zeros(128, 31, device='cuda'), torch.zeros(128, device='cuda'), geom))):
            for _ in range(2): fn()
            torch.cuda.synchronize(); t0 = time.time()
            for _ in range(5): fn()
            torch.cuda.synchronize(); dt = (time.time() - t0) / 5
            print(f'{axis:5s} {name:10s} {dt*1e6:8.1f} us  {(5 if name in ("dgrad+glu", "bwd fused") else 2)*x.numel()*4/dt/1e9:7.0f} GB/s', flush=True)

@cmd
def ff_f16(argv):
    """the default feed-forward forward (scaled fp16, H not stored) at the benchmark shape"""
    import time, torch
    from speech_enhancement_amd import gemm as GM, ops as O
    from speech_enhancement_amd.weights import WeightPlan
    M = 16 * 321 * 101
    dev = torch.device('cuda')
    torch.manual_seed(0)
    x = torch.randn(M, 64, device=dev); st = O.row_stats(x, M)
    g, b = torch.rand(64, device=dev) + 0.5, torch.randn(64, device=dev) * 0.1
    W1, b1 = torch.randn(256, 64, device=dev) * 0.1, torch.randn(256, device=dev) * 0.1
    W2, b2 = torch.randn(64, 256, device=dev) * 0.05, torch.randn(64, device=dev) * 0.1
    p = WeightPlan(dev); p.linear('w1', W1, planes='f16'); p.linear('w2', W2, planes='f16'); p.run()
    for sh in (False, True):
        f = lambda: GM.ff_fwd(x, st, g, b, p.out['w1'], b1, p.out['w2'], b2, 0.2, 11, 12, 0.5, hid=256, store_h=sh)
        for _ in range(3): f()
        torch.cuda.synchronize(); t0 = time.time()
        for _ in range(20): f()
        torch.cuda.synchronize(); dt = (time.time() - t0) / 20
        print(f'ff_fwd f16x3 store_h={sh}: {dt*1e6:7.1f} us', flush=True)


@cmd
def normbwd_one(argv):
    """InstanceNorm + PReLU backward (reduce + apply launches) at the dense-block shape of the step: B = 16, 321 x 101 pixels, C = 64, the
    operands living in 256-wide skip slabs like in the model"""
    import time, torch
    from speech_enhancement_amd import ops as O
    B, P, C = 16, 321 * 101, 64
    dev = torch.device('cuda')
    torch.manual_seed(0)
    x = torch.randn(B, P, C, device=dev); dy = torch.randn(B, P, 256, device=dev); dx = torch.empty(B, P, C, device=dev)
    mr = torch.stack([x.mean(1), 1.0 / torch.sqrt(x.var(1, unbiased=False) + 1e-5)], -1).contiguous()
    g, be, sl = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.1, torch.full((C,), 0.25, device=dev)
    dg, db, ds = torch.zeros(C, device=dev), torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    am = torch.zeros(1, device=dev)
    f = lambda: O.norm_prelu_bwd(x, C, 0, mr, g, be, sl, dy, 256, 64, dx, C, 0, dg, db, ds, B, P, C, amax=am)
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(20): f()
    torch.cuda.synchronize(); dt = (time.time() - t0) / 20
    print(f'norm_prelu_bwd reduce + apply: {dt*1e6:7.1f} us  ({5 * 4.0 * B * P * C / dt / 1e9:6.0f} GB/s over 5 plane passes)', flush=True)


@cmd
def lnbwd_one(argv):
    """se_gemm_ln_bwd_wgrad at the benchmark shape (M = 16 * 321 * 101 rows; K = 192: qkv, 256: pointwise-GLU)"""
    import time, torch
    from speech_enhancement_amd import gemm as GM, ops as O
    from speech_enhancement_amd.weights import WeightPlan
    M = 16 * 321 * 101
    dev = torch.device('cuda')
    torch.manual_seed(0)
    x = torch.randn(M, 64, device=dev); st = O.row_stats(x, M)
    g, b = torch.rand(64, device=dev) + 0.5, torch.randn(64, device=dev) * 0.1
    dR = torch.randn(M, 64, device=dev) * 1e-3
    for K in (192, 256):
        W = torch.randn(K, 64, device=dev) * 0.1
        A = torch.randn(M, K, device=dev) * 1e-3
        A._se_amax = A.abs().max().reshape(1).clone()
        plan = WeightPlan(dev); WT = plan.linear_T('wt', W, planes='f16'); plan.run()
        dg, db, dW, dbias = torch.zeros(64, device=dev), torch.zeros(64, device=dev), torch.zeros(K, 64, device=dev), torch.zeros(K, device=dev)
        am = torch.zeros(1, device=dev)
        f = lambda: GM.gemm_ln_bwd_wgrad(A, WT, x, st, g, b, dR, dg, db, dW, dbias, out_amax=am)
        for _ in range(3): f()
        torch.cuda.synchronize(); t0 = time.time()
        for _ in range(20): f()
        torch.cuda.synchronize(); dt = (time.time() - t0) / 20
        print(f'lnbwd_wgrad K={K}: {dt*1e6:7.1f} us  {4.0*M*(K+192)/dt/1e9:6.0f} GB/s algorithmic', flush=True)


@cmd
def ff_one(argv):
    __import__('sys').argv = ['ff_one'] + list(argv)
    """fused feed-forward forward / input-gradient kernels at the benchmark shape (M = 16 * 321 * 101 tokens, hidden 256, pre-split weights)"""
    import os, sys, time, torch
    from speech_enhancement_amd import gemm as GM, ops as O
    from speech_enhancement_amd.weights import WeightPlan
    M = 16 * 321 * 101
    torch.manual_seed(0)
    x = torch.randn(M, 64, device='cuda'); st = O.row_stats(x, M)
    g, b = torch.rand(64, device='cuda') + 0.5, torch.randn(64, device='cuda') * 0.1
    W1, b1 = torch.randn(256, 64, device='cuda') * 0.1, torch.randn(256, device='cuda') * 0.1
    W2, b2 = torch.randn(64, 256, device='cuda') * 0.05, torch.randn(64, device='cuda') * 0.1
    plan = WeightPlan(torch.device('cuda'))
    W1p, W2p = plan.linear('w1', W1, planes=True), plan.linear('w2', W2, planes=True)
    W2T, W1T = plan.linear_T('w2t', W2, planes=True, scale=0.5), plan.linear_T('w1t', W1, planes=True)
    plan.run()
    dy = torch.randn(M, 64, device='cuda')
    def bench(f, n=10):
        for _ in range(2): f()
        torch.cuda.synchronize(); t0 = time.time()
        for _ in range(n): f()
        torch.cuda.synchronize(); return (time.time() - t0) / n * 1e6
    t = bench(lambda: GM.ff_fwd(x, st, g, b, W1p, b1, W2p, b2, 0.2, 11, 12, 0.5, precision=2, hid=256, out_stats=True))
    print(f'ff_fwd  {t:7.1f} us')
    y, h, _ = GM.ff_fwd(x, st, g, b, W1p, b1, W2p, b2, 0.2, 11, 12, 0.5, precision=2, hid=256, out_stats=True)
    dg, db = torch.zeros(64, device='cuda'), torch.zeros(64, device='cuda')
    t = bench(lambda: GM.ff_bwd_dgrad(dy, h, W2T, W1T, 0.2, 11, 12, precision=2, ln=(x, st, g, None, dg, db)))
    print(f'ff_bwd  {t:7.1f} us')

@cmd
def gemm_bench(argv):
    __import__('sys').argv = ['gemm_bench'] + list(argv)
    """micro-benchmark of the tap-GEMM at the Conformer linear-layer shapes (M = 16*321*101 tokens)."""
    import os, sys, time, torch
    from speech_enhancement_amd import gemm as GM, _lib as L
    M = 16 * 321 * 101


    PREC = int(os.environ.get('PREC', '0'))


    def run(name, C, N, pro=0, epi=0, aux=False, bias=False, resid=False, ldc=None, reps=5):
        x = torch.randn(M, C, device='cuda')
        w = torch.randn(N, C, device='cuda') * C ** -0.5
        No = N // 2 if epi & L.EPI_GLU else N
        y = torch.empty(M, ldc or No, device='cuda')
        kw = {}
        if aux:
            kw['AUX'] = torch.randn(M, N, device='cuda'); ldx = N
        else:
            ldx = 0
        if bias:
            kw['bias'] = torch.randn(N, device='cuda')
        if resid:
            kw['R'] = torch.randn(M, N, device='cuda')
        if pro == L.PRO_LN:
            kw['rowstats'] = torch.stack([x.mean(-1), x.var(-1).rsqrt()], -1).contiguous()
            kw['ps'] = torch.ones(C, device='cuda'); kw['pb'] = torch.zeros(C, device='cuda')
        d = GM.linear_desc(M, C, N, ldc=ldc or No, prologue=pro, epilogue=epi, ldx=ldx, ldr=N if resid else 0, drop_p=0.2 if (epi & L.EPI_DROP or pro in (4, 5)) else 0.0,
                           pro_seed=123, epi_seed=456, precision=PREC)
        for _ in range(2):
            GM.gemm_tap(d, x, w, y, **kw)
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(reps):
            GM.gemm_tap(d, x, w, y, **kw)
        torch.cuda.synchronize()
        dt = (time.time() - t0) / reps
        gb = 4.0 * M * (C + No + (N if aux else 0) + (N if resid else 0)) / 1e9
        print(f'{name:44s} {dt*1e6:8.1f} us  {2.0*M*C*N/dt/1e12:6.1f} TF  {gb/dt:7.0f} GB/s (min traffic {gb:.2f} GB)', flush=True)


    run('K64->N256 plain', 64, 256)
    run('K64->N256 +bias', 64, 256, epi=L.EPI_BIAS, bias=True)
    run('K64->N256 LN pro + bias', 64, 256, pro=L.PRO_LN, epi=L.EPI_BIAS, bias=True)
    run('K64->N256 swishgrad (AUX)', 64, 256, epi=L.EPI_SWISH_GRAD, aux=True)
    run('K64->N256 drop pro + swishgrad + drop', 64, 256, pro=L.PRO_DROP, epi=L.EPI_SWISH_GRAD | L.EPI_DROP, aux=True)
    run('K64->N256 GLU + Z', 64, 256, pro=L.PRO_LN, epi=L.EPI_BIAS | L.EPI_GLU, aux=True, bias=True)
    run('K64->N64 plain', 64, 64)
    run('K64->N64 resid+bias', 64, 64, epi=L.EPI_BIAS | L.EPI_RESID, bias=True, resid=True)
    run('K64->N192 LN', 64, 192, pro=L.PRO_LN)
    run('K256->N64 plain', 256, 64)
    run('K256->N64 swish pro + resid', 256, 64, pro=L.PRO_SWISH, epi=L.EPI_BIAS | L.EPI_RESID, bias=True, resid=True)
    run('K128->N64 plain', 128, 64)
    run('K192->N64 plain', 192, 64)

@cmd
def gemm_one(argv):
    __import__('sys').argv = ['gemm_one'] + list(argv)
    """one Conformer-shaped linear GEMM, for rocprofv3 --pmc runs: args C N [pro] [epi] [prec]"""
    import os, sys, torch
    from speech_enhancement_amd import gemm as GM, _lib as L
    M = 16 * 321 * 101
    C, N = int(sys.argv[1]), int(sys.argv[2])
    pro = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    epi = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    prec = int(sys.argv[5]) if len(sys.argv) > 5 else 0
    x = torch.randn(M, C, device='cuda'); w = torch.randn(N, C, device='cuda') * C ** -0.5
    No = N // 2 if epi & L.EPI_GLU else N
    y = torch.empty(M, No, device='cuda')
    kw = {}
    if pro == L.PRO_LN:
        kw['rowstats'] = torch.stack([x.mean(-1), x.var(-1).rsqrt()], -1).contiguous()
        kw['ps'] = torch.ones(C, device='cuda'); kw['pb'] = torch.zeros(C, device='cuda')
    if epi & L.EPI_BIAS:
        kw['bias'] = torch.randn(N, device='cuda')
    if epi & (L.EPI_SWISH_GRAD | L.EPI_GLU):
        kw['AUX'] = torch.randn(M, N, device='cuda')
    if epi & L.EPI_RESID:
        kw['R'] = torch.randn(M, N, device='cuda')
    d = GM.linear_desc(M, C, N, ldc=No, prologue=pro, epilogue=epi, ldx=N if 'AUX' in kw else 0, ldr=N if 'R' in kw else 0,
                       drop_p=0.2 if (epi & L.EPI_DROP or pro in (4, 5)) else 0.0, pro_seed=1, epi_seed=2, precision=prec)
    for _ in range(3):
        GM.gemm_tap(d, x, w, y, **kw)
    torch.cuda.synchronize()

@cmd
def infer_graph(argv):
    __import__('sys').argv = ['infer_graph'] + list(argv)
    """eager vs HIP-graph batch-1 inference latency (2 s and 10 s clips)"""
    import os, sys, time, types, numpy as np, torch
    import speech_enhancement_amd as S
    from speech_enhancement_amd import inference as INF
    torch.manual_seed(0)
    G = S.TSCNet(64, 201); G.apply(S.kaiming_init); G.cuda().eval()
    cfg = types.SimpleNamespace(N_FFT=400, HOP_SAMPLES=100)
    for L in (32000, 160000):
        x = (0.1 * np.random.RandomState(0).randn(L)).astype(np.float32)
        for _ in range(2): INF.predict(G, cfg, x)
        torch.cuda.synchronize(); t0 = time.time()
        for _ in range(5): INF.predict(G, cfg, x)
        te = (time.time() - t0) / 5
        enh = INF.GraphedEnhancer(G, cfg, L)
        for _ in range(2): enh(x)
        t0 = time.time()
        for _ in range(5): y = enh(x)
        tg = (time.time() - t0) / 5
        print(f'L={L}: eager {te*1e3:.1f} ms, graph {tg*1e3:.1f} ms, max diff {np.abs(y - INF.predict(G, cfg, x)).max():.2e}')

@cmd
def k64_one(argv):
    __import__('sys').argv = ['k64_one'] + list(argv)
    """row-panel kernel (K = 64 -> N = 192 qkv / 256 pointwise-GLU, LayerNorm prologue, pre-split weights) at the benchmark size"""
    import os, sys, time, torch
    from speech_enhancement_amd import gemm as GM, ops as O, _lib as L
    from speech_enhancement_amd.weights import WeightPlan
    M = 16 * 321 * 101
    torch.manual_seed(0)
    x = torch.randn(M, 64, device='cuda'); st = O.row_stats(x, M)
    g, b = torch.rand(64, device='cuda') + 0.5, torch.randn(64, device='cuda') * 0.1
    plan = WeightPlan(torch.device('cuda'))
    f16 = len(sys.argv) > 1 and sys.argv[1] == 'f16'
    from speech_enhancement_amd import layers as LY
    Wq = plan.linear('q', torch.randn(192, 64, device='cuda') * 0.1, planes='f16' if f16 else True)
    Wp = plan.linear('p', torch.randn(256, 64, device='cuda') * 0.1, planes='f16' if f16 else True)
    bp = torch.randn(256, device='cuda') * 0.1
    plan.run()
    def bench(f, n=10):
        for _ in range(2): f()
        torch.cuda.synchronize(); t0 = time.time()
        for _ in range(n): f()
        torch.cuda.synchronize(); return (time.time() - t0) / n * 1e6
    qkv = torch.empty(M, 192, device='cuda')
    kq = LY._lin3(Wq, a_sexp=GM.LN_SEXP) if f16 else {}
    kp = LY._lin3(Wp, a_sexp=GM.LN_SEXP) if f16 else {}
    print(f'LN -> 192 (qkv)       {bench(lambda: GM.gemm_tap(GM.linear_desc(M, 64, 192, prologue=L.PRO_LN, **kq), x, Wq, qkv, rowstats=st, ps=g, pb=b)):7.1f} us')
    u = torch.empty(M, 128, device='cuda'); zc = torch.empty(M, 256, device='cuda')
    print(f'LN -> 256 GLU + Z     {bench(lambda: GM.gemm_tap(GM.linear_desc(M, 64, 256, ldc=128, prologue=L.PRO_LN, epilogue=L.EPI_BIAS | L.EPI_GLU, ldx=256, **kp), x, Wp, u, bias=bp, AUX=zc, rowstats=st, ps=g, pb=b)):7.1f} us')
    y256 = torch.empty(M, 256, device='cuda')
    print(f'LN -> 256 plain       {bench(lambda: GM.gemm_tap(GM.linear_desc(M, 64, 256, prologue=L.PRO_LN, epilogue=L.EPI_BIAS, **kp), x, Wp, y256, bias=bp, rowstats=st, ps=g, pb=b)):7.1f} us')
    print(f'LN -> 256 GLU no aux  {bench(lambda: GM.gemm_tap(GM.linear_desc(M, 64, 256, ldc=128, prologue=L.PRO_LN, epilogue=L.EPI_BIAS | L.EPI_GLU, **kp), x, Wp, u, bias=bp, rowstats=st, ps=g, pb=b)):7.1f} us')
    print(f'LN -> 256 GLU + gate  {bench(lambda: GM.gemm_tap(GM.linear_desc(M, 64, 256, ldc=128, prologue=L.PRO_LN, epilogue=L.EPI_BIAS | L.EPI_GLU | L.EPI_GLU_GATE, ldx=128, **kp), x, Wp, u, bias=bp, AUX=zc, rowstats=st, ps=g, pb=b)):7.1f} us')

@cmd
def lin_wgrad(argv):
    __import__('sys').argv = ['lin_wgrad'] + list(argv)
    """token-wise (row-GEMM) weight gradients of a Conformer block: full-tile kernel (wgrad_lin_kernel) vs the per-block kernel"""
    import os, sys, time, torch
    from speech_enhancement_amd import gemm as GM, _lib as L
    M = 16 * 321 * 101
    torch.manual_seed(0)
    cases = [('LN x[M,64] -> dW[256,64] (ff W1, pw1)', 64, 256, L.PRO_LN, 0),
             ('LN x[M,64] -> dW[192,64] (qkv)', 64, 192, L.PRO_LN, 0),
             ('swish+drop H[M,256] -> dW[64,256] (ff W2)', 256, 64, L.PRO_SWISH_DROP, L.EPI_DROP),
             ('bn+swish h[M,128] -> dW[64,128] (pw2)', 128, 64, L.PRO_AFFINE_SWISH, 0)]
    for name, Cin, N, pro, epi in cases:
        x = torch.randn(M, Cin, device='cuda'); dy = torch.randn(M, N, device='cuda')
        st = torch.stack([x.mean(-1), (x.var(-1, unbiased=False) + 1e-5).rsqrt()], -1).contiguous()
        g = torch.rand(Cin, device='cuda') + 0.5; b = torch.randn(Cin, device='cuda') * 0.1
        d = GM.linear_desc(M, Cin, N, prologue=pro, epilogue=epi, pro_seed=5, epi_seed=7, drop_p=0.2 if pro == L.PRO_SWISH_DROP else 0.0,
                           a_sexp=3, w_sexp=3)
        res = {}
        for mode in ('blocks', 'full', 'full-x6', 'full-f16'):
            if mode == 'blocks':
                os.environ['SE_WGRAD_NO_LIN'] = '1'
            else:
                os.environ.pop('SE_WGRAD_NO_LIN', None)
            d.precision = {'full-x6': 2, 'full-f16': 3}.get(mode, 0)
            dw = torch.zeros(N, Cin, device='cuda'); db = torch.zeros(N, device='cuda')
            f = lambda: GM.gemm_tap_wgrad(d, x, dy, dw, db, rowstats=st, ps=g, pb=b, explicit_precision=True)
            f(); torch.cuda.synchronize()
            res[mode] = (dw.clone(), db.clone())
            for _ in range(2): f()
            torch.cuda.synchronize(); t0 = time.time()
            for _ in range(10): f()
            torch.cuda.synchronize(); dt = (time.time() - t0) / 10
            print(f'{name:48s} {mode:7s} {dt*1e6:7.1f} us  {2.0*M*Cin*N/dt/1e12:6.1f} TF', flush=True)
        for m2 in ('full', 'full-x6', 'full-f16'):
            e = float((res[m2][0] - res['blocks'][0]).abs().max() / res['blocks'][0].abs().max())
            eb = float((res[m2][1] - res['blocks'][1]).abs().max() / res['blocks'][1].abs().max())
            print(f'    max relative difference {m2} vs blocks: dW {e:.2e}, dbias {eb:.2e}')

@cmd
def ln_bwd(argv):
    __import__('sys').argv = ['ln_bwd'] + list(argv)
    """se_gemm_ln_bwd (input-gradient GEMM + LayerNorm backward on the accumulators) vs the two-kernel form at bench size"""
    import os, sys, time, torch
    from speech_enhancement_amd import gemm as GM, ops as O
    from speech_enhancement_amd.weights import WeightPlan
    M = 16 * 321 * 101
    for K in (192, 256):
        x, dy, dR = torch.randn(M, 64, device='cuda'), torch.randn(M, K, device='cuda'), torch.randn(M, 64, device='cuda')
        W = torch.randn(K, 64, device='cuda') * 0.1
        gam = torch.rand(64, device='cuda') + 0.5
        st = O.row_stats(x, M)
        plan = WeightPlan(torch.device('cuda')); WT = plan.linear_T('wt', W, planes=True); plan.run()
        dg, db = torch.zeros(64, device='cuda'), torch.zeros(64, device='cuda')
        def fused(): return GM.gemm_ln_bwd(dy, WT, x, st, gam, dR, dg, db)
        def two():
            dl = torch.empty(M, 64, device='cuda')
            GM.gemm_tap(GM.linear_desc(M, K, 64, precision=2), dy, WT, dl)
            return O.layernorm_bwd(x, st, gam, dl, dg, db, dR=dR)
        for name, f in (('fused', fused), ('two kernels', two), ('fused', fused), ('two kernels', two)):
            for _ in range(2): f()
            torch.cuda.synchronize(); t0 = time.time()
            for _ in range(10): f()
            torch.cuda.synchronize(); print(f'K={K} {name:12s} {(time.time() - t0) / 10 * 1e6:7.1f} us', flush=True)

@cmd
def mem(argv):
    __import__('sys').argv = ['mem'] + list(argv)
    """peak / reserved device memory of the train step over a few steps (stream concurrency on unless SE_NO_* are set)"""
    import sys, os, types, torch
    import speech_enhancement_amd as S
    from speech_enhancement_amd import train as TR, optim
    arch = sys.argv[1] if len(sys.argv) > 1 else 'cmgan'
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    torch.manual_seed(0)
    G, D = S.TSCNet(64, 201), S.Discriminator(16); G.apply(S.kaiming_init); D.apply(S.kaiming_init); G.cuda().train(); D.cuda().train()
    a = types.SimpleNamespace(optimizer='adamw', lr=5e-4, weight_decay=0.01, momentum=0.9, max_norm=0.0)
    og, od = optim.build_optimizer(a, G), optim.build_optimizer(a, D)
    clean = 0.1 * torch.randn(B, 32000, device='cuda'); noisy = clean + 0.05 * torch.randn_like(clean); q = torch.rand(B, device='cuda')
    labels = {'est': q, 'clean': torch.full_like(q, 0.97), 'noisy': q * 0.5}
    w = (0.1, 0.9, 0.2, 0.05) if arch == 'cmgan' else (0.3, 0.7, 0.2, 0.05)
    for s in range(12):
        TR.gan_step(G, D, og, od, clean, noisy, arch, w, labels=labels)
        if s in (1, 3, 7, 11):
            torch.cuda.synchronize()
            print(f'{arch} B={B} step {s + 1}: max allocated {torch.cuda.max_memory_allocated() / 2**30:.1f} GB, reserved {torch.cuda.memory_reserved() / 2**30:.1f} GB', flush=True)

@cmd
def norm_bwd(argv):
    __import__('sys').argv = ['norm_bwd'] + list(argv)
    """InstanceNorm+PReLU backward (se_norm_prelu_bwd) at the dense-block shapes: time per call and algorithmic GB/s."""
    import os, sys, time
    import torch
    import speech_enhancement_amd as S
    from speech_enhancement_amd import ops as O

    B, T = int(os.environ.get('NB', '6')), 321
    NCH = int(os.environ.get('NCH', '1'))           # batch chunks: reduce + apply per chunk (second pass out of the MALL)
    for Fq, ldy in ((101, 64), (101, 256), (201, 64), (201, 256)):
        P = T * Fq
        R = torch.randn(B, T, Fq, 64, device='cuda')
        dy = torch.randn(B, T, Fq, ldy, device='cuda')
        mr = torch.rand(B, 64, 2, device='cuda') + 0.5
        g, b, sl = torch.randn(64, device='cuda'), torch.randn(64, device='cuda'), torch.full((64,), 0.25, device='cuda')
        dg, db, ds = (torch.zeros(64, device='cuda') for _ in range(3))
        dR = torch.empty_like(R)
        def f():
            cb = B // NCH
            for b0 in range(0, B, cb):
                O.norm_prelu_bwd(R[b0:b0 + cb], 64, 0, mr[b0:b0 + cb], g, b, sl, dy[b0:b0 + cb], ldy, ldy - 64, dR[b0:b0 + cb], 64, 0, dg, db, ds, cb, P, 64, per_batch=True)
        for _ in range(3): f()
        torch.cuda.synchronize(); t0 = time.time()
        for _ in range(30): f()
        torch.cuda.synchronize(); dt = (time.time() - t0) / 30
        print(f'F={Fq} ldy={ldy}: {dt*1e6:.1f} us per call (both passes), {5 * R.numel() * 4 / dt / 1e9:.0f} GB/s algorithmic')

@cmd
def pesq_overlap(argv):
    __import__('sys').argv = ['pesq_overlap'] + list(argv)
    """PESQ side channel: step time with a slow label provider (sleep) vs labels supplied"""
    import os, sys, time, types, torch
    import speech_enhancement_amd as S
    from speech_enhancement_amd import train as TR, optim as OP
    torch.manual_seed(0)
    B = 16
    G, D = S.TSCNet(64, 201), S.Discriminator(16)
    G.apply(S.kaiming_init); D.apply(S.kaiming_init)
    G.cuda().train(); D.cuda().train()
    oa = types.SimpleNamespace(optimizer='adamw', lr=5e-4, weight_decay=0.01, momentum=0.9, max_norm=0.0)
    og, od = OP.build_optimizer(oa, G), OP.build_optimizer(oa, D)
    clean = 0.1 * torch.randn(B, 32000, device='cuda'); noisy = clean + 0.05 * torch.randn_like(clean)
    q = torch.rand(B, device='cuda')
    w = (0.1, 0.9, 0.2, 0.05)
    def run(labels, n=5):
        for _ in range(2): TR.gan_step(G, D, og, od, clean, noisy, 'cmgan', w, labels=labels)
        torch.cuda.synchronize(); t0 = time.time()
        for _ in range(n): TR.gan_step(G, D, og, od, clean, noisy, 'cmgan', w, labels=labels)
        torch.cuda.synchronize(); return (time.time() - t0) / n * 1e3
    print('labels supplied: %.1f ms/step' % run({'est': q}))
    for ms in (40, 80, 120):
        TR.set_pesq_provider(lambda c, d, ms=ms: (time.sleep(ms / 1e3), torch.rand(len(c)))[1])
        print('provider taking %d ms on the host, side channel: %.1f ms/step' % (ms, run(None)))

@cmd
def rowgemm_one(argv):
    __import__('sys').argv = ['rowgemm_one'] + list(argv)
    """fp32-MFMA row GEMMs with a residual epilogue at the benchmark size: attention out-projection (64 -> 64, bias + dropout +
    residual + row statistics) and pointwise conv 2 (BatchNorm-affine + Swish prologue, 128 -> 64, bias + residual + row statistics)"""
    import os, sys, time, torch
    from speech_enhancement_amd import gemm as GM, _lib as L
    M = 16 * 321 * 101
    torch.manual_seed(0)
    def bench(f, n=10):
        for _ in range(2): f()
        torch.cuda.synchronize(); t0 = time.time()
        for _ in range(n): f()
        torch.cuda.synchronize(); return (time.time() - t0) / n * 1e6
    o, y1 = torch.randn(M, 64, device='cuda'), torch.randn(M, 64, device='cuda')
    Wo, bo = torch.randn(64, 64, device='cuda') * 0.1, torch.randn(64, device='cuda') * 0.1
    y2, st = torch.empty(M, 64, device='cuda'), torch.empty(M, 2, device='cuda')
    f1 = lambda: GM.gemm_tap(GM.linear_desc(M, 64, 64, epilogue=L.EPI_BIAS | L.EPI_RESID | L.EPI_DROP | L.EPI_ROWSTATS, alpha=1.0, ldr=64,
                                            epi_seed=5, drop_p=0.1), o, Wo, y2, bias=bo, R=y1, AUX=st)
    print(f'to_out 64 -> 64 (+resid, drop, rowstats)  {bench(f1):7.1f} us')
    h = torch.randn(M, 128, device='cuda'); W2, b2 = torch.randn(64, 128, device='cuda') * 0.1, torch.randn(64, device='cuda') * 0.1
    sc, sh = torch.rand(128, device='cuda') + 0.5, torch.randn(128, device='cuda') * 0.1
    f2 = lambda: GM.gemm_tap(GM.linear_desc(M, 128, 64, prologue=L.PRO_AFFINE_SWISH, epilogue=L.EPI_BIAS | L.EPI_RESID | L.EPI_ROWSTATS,
                                            alpha=1.0, ldr=64), h, W2, y2, bias=b2, R=y1, ps=sc, pb=sh, AUX=st)
    print(f'pw2 128 -> 64 (bn+swish, +resid, rowstats) {bench(f2):7.1f} us')
    dy = torch.randn(M, 64, device='cuda'); do = torch.empty(M, 64, device='cuda')
    f3 = lambda: GM.gemm_tap(GM.linear_desc(M, 64, 64, prologue=L.PRO_DROP, pro_seed=5, drop_p=0.1), dy, Wo, do)
    print(f'to_out dgrad 64 -> 64 (drop prologue)      {bench(f3):7.1f} us')

@cmd
def wgrad_one(argv):
    __import__('sys').argv = ['wgrad_one'] + list(argv)
    """micro-benchmark of the conv weight-gradient kernel (dense layer shape) and of the split-bf16 conv forward."""
    import os, sys, time, torch
    from speech_enhancement_amd import gemm as GM, _lib as L, layers as LY
    B, T, Fq = 16, 321, 201
    Cin = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    mode = sys.argv[2] if len(sys.argv) > 2 else 'wgrad'
    prec = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    skip = torch.randn(B, T, Fq, 256, device='cuda')
    dR = torch.randn(B, T, Fq, 64, device='cuda')
    taps = LY.dense_taps(3)
    if mode == 'wgrad':
        d = GM.make_desc(B, T, Fq, T, Fq, taps, Cin, 256, 64, 64, precision=prec)
        dwp = torch.zeros(64, len(taps) * Cin, device='cuda')
        ch = int(os.environ['CHUNKS']) if 'CHUNKS' in os.environ else None
        f = lambda: GM.gemm_tap_wgrad(d, skip, dR, dwp, None, chunks=ch)
    else:
        w = torch.randn(64, Cin, 2, 3, device='cuda') * 0.02
        wp = GM.pack_conv_fwd(w)
        d = GM.make_desc(B, T, Fq, T, Fq, taps, Cin, 256, 64, 64, precision=prec)
        f = lambda: GM.gemm_tap(d, skip, wp, dR)
    for _ in range(2):
        f()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(4):
        f()
    torch.cuda.synchronize()
    dt = (time.time() - t0) / 4
    print(f'{mode} Cin={Cin}: {dt*1e6:.0f} us, {2.0*B*T*Fq*64*6*Cin/dt/1e12:.1f} TF')

@cmd
def aten_prof(argv):
    __import__('sys').argv = ['aten_prof'] + list(argv)
    """count the aten ops (PyTorch glue) in one train step"""
    import os, sys, torch
    import speech_enhancement_amd as S
    from speech_enhancement_amd import train as TR, optim as OP
    from torch.profiler import profile, ProfilerActivity
    import types
    torch.manual_seed(0)
    B = 16
    G, D = S.TSCNet(64, 201), S.Discriminator(16)
    G.apply(S.kaiming_init); D.apply(S.kaiming_init)
    G.cuda().train(); D.cuda().train()
    oargs = types.SimpleNamespace(optimizer='adamw', lr=5e-4, weight_decay=0.01, momentum=0.9, max_norm=0.0)
    og, od = OP.build_optimizer(oargs, G), OP.build_optimizer(oargs, D)
    clean = torch.randn(B, 32000, device='cuda') * 0.1; noisy = clean + 0.05 * torch.randn_like(clean)
    q = torch.rand(B, device='cuda')
    labels = {'est': q, 'clean': torch.full_like(q, 0.97), 'noisy': q * 0.5}
    w = (0.1, 0.9, 0.2, 0.05)
    for _ in range(2): TR.gan_step(G, D, og, od, clean, noisy, 'cmgan', w, labels=labels)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=False, record_shapes=True) as prof:
        TR.gan_step(G, D, og, od, clean, noisy, 'cmgan', w, labels=labels)
        torch.cuda.synchronize()
    rows = [e for e in prof.key_averages(group_by_input_shape=True) if e.key.startswith('aten::')]
    rows.sort(key=lambda e: -e.count)
    rows = [e for e in rows if e.device_time_total > 0 and e.key not in ('aten::zeros_like', 'aten::zero_', 'aten::clone', 'aten::contiguous', 'aten::zeros')]
    print('launching aten ops per step:', sum(e.count for e in rows))
    for e in rows[:90]:
        print(f'{e.key:28s} n={e.count:4d} cuda={e.device_time_total/1e3:7.3f} ms  {str(e.input_shapes)[:110]}')


if __name__ == '__main__':
    if len(sys.argv) < 2 or sys.argv[1] in ('list', '-h', '--help') or sys.argv[1] not in CMDS:
        print(__doc__)
        print('sub-commands:', ' '.join(sorted(CMDS)))
        sys.exit(0 if len(sys.argv) > 1 and sys.argv[1] in ('list', '-h', '--help') else 2)
    CMDS[sys.argv[1]](sys.argv[2:])
