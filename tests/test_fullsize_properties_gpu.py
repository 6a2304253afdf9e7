"""Benchmark-size (BASELINE configs[1]: B = 16, T = 321, F = 201 / 101) property checks of the large-grid paths: XCD-aware work
decode over ~10^4 workgroups, 32-bit lane offsets on 1 GB operands, whole-round chunking, sequence-sized tiles.  The oracle cannot
run these sizes in seconds, so every check is a size-independent property: two independently written kernels for the same op
agree, a transform inverts, an op is linear / equivariant, or a random SAMPLE of the output matches fp64 torch."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import formula

pytestmark = pytest.mark.gpu
B, T, Fe, Fp = 16, 321, 201, 101


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator(device='cuda').manual_seed(seed)
    return torch.randn(*shape, generator=g, device='cuda') * scale


def relerr(a, b):
    return float((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-30))


class env:
    def __init__(self, **kv):
        self.kv = kv

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kv}
        os.environ.update(self.kv)

    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def test_stft_istft_round_trip_and_linearity():
    from speech_enhancement_amd import frontend as FE
    x = rnd(B, 32000, seed=1, scale=0.1)
    win = torch.hamming_window(400, device='cuda')
    for comp in ('pow', 'log', 'norm'):
        spec = FE.compressed_stft(x, 400, 100, win, comp_type=comp)
        assert spec.shape == (B, 201, 321)
        y = FE.uncompressed_istft(spec, 400, 100, win, comp_type=comp)
        assert y.shape == x.shape and relerr(y, x) < 2e-5, comp
    # the un-compressed transform is linear
    a, b_ = rnd(B, 32000, seed=2), rnd(B, 32000, seed=3)
    sa, sb = FE.compressed_stft(a, 400, 100, win, comp_type='none'), FE.compressed_stft(b_, 400, 100, win, comp_type='none')
    sab = FE.compressed_stft(a + 0.5 * b_, 400, 100, win, comp_type='none')
    assert relerr(torch.view_as_real(sab), torch.view_as_real(sa + 0.5 * sb)) < 2e-6


@pytest.mark.parametrize('layer', [3, 1])
def test_dense_conv_triple_tap_kernels_vs_generic_kernels(layer):
    """conv3 / wgrad3 (halo-tile kernels) against the generic tap kernels on the encoder's dilated dense layer at B = 16"""
    from speech_enhancement_amd import gemm as GM, layers as LY, _lib as L
    C = 64 * (layer + 1)
    skip = rnd(B, T, Fe, 256, seed=1)
    w = rnd(64, C, 2, 3, seed=2, scale=(6 * C) ** -0.5)
    wp = GM.pack_conv_fwd(w)
    bias = rnd(64, seed=3)
    taps = LY.dense_taps(layer)
    outs, grads = [], []
    dR = rnd(B, T, Fe, 64, seed=4)
    for no3 in (False, True):
        with env(**({'SE_GEMM_NO_CONV3': '1'} if no3 else {})):
            y = torch.empty(B, T, Fe, 64, device='cuda')
            st = torch.zeros(B, 64, 2, device='cuda', dtype=torch.float64)
            d = GM.make_desc(B, T, Fe, T, Fe, taps, C, 256, 64, 64, epilogue=L.EPI_BIAS | L.EPI_STATS, precision=2)
            GM.gemm_tap(d, skip, wp, y, bias=bias, stats=st)
            dwp = torch.zeros(64, 6 * C, device='cuda')
            GM.gemm_tap_wgrad(GM.make_desc(B, T, Fe, T, Fe, taps, C, 256, 64, 64, precision=2), skip, dR, dwp, None)
            outs.append((y, st))
            grads.append(dwp)
    assert relerr(outs[0][0], outs[1][0]) < 2e-6 and relerr(outs[0][1], outs[1][1]) < 1e-9
    assert relerr(grads[0], grads[1]) < 2e-5
    # a random sample of output pixels against fp64 conv2d (padding: dil rows on top only, 1 column each side)
    g = torch.Generator().manual_seed(5)
    dil = 2 ** layer
    xs = skip[..., :C].double()
    for _ in range(8):
        b, t, f = int(torch.randint(B, (1,), generator=g)), int(torch.randint(T, (1,), generator=g)), int(torch.randint(Fe, (1,), generator=g))
        acc = bias.double().clone()
        for i, dt in enumerate((-dil, 0)):
            for j, df in enumerate((-1, 0, 1)):
                tt, ff = t + dt, f + df
                if 0 <= tt < T and 0 <= ff < Fe:
                    acc += w[:, :, i, j].double() @ xs[b, tt, ff]
        assert relerr(outs[0][0][b, t, f], acc) < 5e-6


def test_linear_weight_gradient_kernels_agree_at_bench_size():
    from speech_enhancement_amd import gemm as GM, ops as O, _lib as L
    M = B * T * Fp
    x, dz = rnd(M, 64, seed=1), rnd(M, 256, seed=2)
    st = O.row_stats(x, M)
    g, b_ = rnd(64, seed=3) * 0.1 + 1, rnd(64, seed=4) * 0.1
    res = []
    for blocks in (True, False):
        with env(**({'SE_WGRAD_NO_LIN': '1'} if blocks else {})):
            dw, db = torch.zeros(256, 64, device='cuda'), torch.zeros(256, device='cuda')
            GM.gemm_tap_wgrad(GM.linear_desc(M, 64, 256, prologue=L.PRO_LN), x, dz, dw, db, rowstats=st, ps=g, pb=b_)
            res.append((dw, db))
    assert relerr(res[0][0], res[1][0]) < 5e-6 and relerr(res[0][1], res[1][1]) < 5e-6
    # linear in dY
    dw2 = torch.zeros(256, 64, device='cuda')
    GM.gemm_tap_wgrad(GM.linear_desc(M, 64, 256, prologue=L.PRO_LN), x, dz * 0.5, dw2, None, rowstats=st, ps=g, pb=b_)
    assert relerr(dw2 * 2, res[1][0]) < 5e-6


@pytest.mark.parametrize('axis', ['time', 'freq'])
def test_attention_kernels_agree_and_are_sequence_local(axis):
    """the workgroup-cooperative scaled-fp16 backward vs the fp32-MFMA backward (two independent kernel families: se_attn_bwd4.h vs the
    dK / dV + dQ / dE passes of round 1) at B = 16; changing ONE sequence changes only that sequence's outputs (no cross-sequence
    leakage through the strided token geometry)"""
    from speech_enhancement_amd import attention as A
    ntok = B * T * Fp
    qkv, dO = rnd(ntok, 192, seed=1), rnd(ntok, 64, seed=2)
    E = rnd(1025, 16, seed=3, scale=0.5)
    geom = A.seq_geometry(B, T, Fp, axis)
    o, lse = A.attn_fwd(qkv, E, geom)
    res = []
    for f16 in (True, False):
        dE = torch.zeros_like(E)
        kw = dict(qkv_amax=qkv.abs().max().reshape(1).clone(), do_amax=dO.abs().max().reshape(1).clone()) if f16 else {}
        res.append((A.attn_bwd(qkv, E, o, dO, lse, geom, dE, **kw), dE))
    scale = float(res[1][0].abs().max())
    assert float((res[0][0] - res[1][0]).abs().max()) < 2e-5 * scale
    assert float((res[0][1] - res[1][1]).abs().max()) < 5e-5 * float(res[1][1].abs().max())
    # sequence locality: perturb the tokens of one sequence
    tok = torch.arange(ntok, device='cuda').view(B, T, Fp)
    sel = tok[3, :, 7] if axis == 'time' else tok[5, 11, :]
    q2 = qkv.clone()
    q2[sel] += 1.0
    o2, _ = A.attn_fwd(q2, E, geom)
    changed = (o2 - o).abs().amax(1) > 0
    assert bool(changed[sel].all()) and int(changed.sum()) == sel.numel()


def test_depthwise_conv_sampled_sequences_and_linearity():
    from speech_enhancement_amd import ops as O, attention as A
    x = rnd(B * T * Fp, 128, seed=1)
    w, b_ = rnd(128, 31, seed=2, scale=0.2), rnd(128, seed=3)
    for axis in ('time', 'freq'):
        geom = A.seq_geometry(B, T, Fp, axis)
        st = torch.zeros(1, 128, 2, device='cuda', dtype=torch.float64)
        y = O.dwconv31(x, w, b_, geom, stats=st)
        assert relerr(st[0, :, 0], y.double().sum(0)) < 1e-6 and relerr(st[0, :, 1], (y.double() ** 2).sum(0)) < 1e-6
        xv, yv = x.view(B, T, Fp, 128), y.view(B, T, Fp, 128)
        for (bi, oi) in ((0, 0), (7, 50), (15, 100 if axis == 'time' else 320)):
            seq_x = xv[bi, :, oi] if axis == 'time' else xv[bi, oi]           # [n, 128]
            seq_y = yv[bi, :, oi] if axis == 'time' else yv[bi, oi]
            ref = F.conv1d(F.pad(seq_x.double().t().unsqueeze(0), (15, 15)), w.double().unsqueeze(1), b_.double(), groups=128)[0].t()
            assert relerr(seq_y, ref) < 1e-5
        y2 = O.dwconv31(x * 0.5, w, None, geom)
        assert relerr(y2 * 2 + b_, y) < 2e-6


def test_fused_feed_forward_vs_separate_gemms_at_bench_size():
    """the fused feed-forward module (forward; input-gradient chain + LayerNorm backward; in-place scaled weight gradients) against
    the same module evaluated with separate fp32-MFMA GEMMs + the standalone LayerNorm backward, M = 518 736 tokens, dropout 0.2
    (the counter-based masks are functions of (seed, element): identical in both evaluations)"""
    from speech_enhancement_amd import layers as LY, gemm as GM
    M = B * T * Fp
    p = 'ff'
    P = {f'{p}.fn.norm.weight': rnd(64, seed=1) * 0.1 + 1, f'{p}.fn.norm.bias': rnd(64, seed=2) * 0.1,
         f'{p}.fn.fn.net.0.weight': rnd(256, 64, seed=3, scale=0.125), f'{p}.fn.fn.net.0.bias': rnd(256, seed=4) * 0.1,
         f'{p}.fn.fn.net.3.weight': rnd(64, 256, seed=5, scale=0.0625), f'{p}.fn.fn.net.3.bias': rnd(64, seed=6) * 0.1}
    x, dy = rnd(M, 64, seed=7), rnd(M, 64, seed=8)
    res = []
    saved = GM.LINEAR_PRECISION
    try:
        for prec in (2, 0):                       # 2: fused split-bf16 kernels; 0: separate fp32-MFMA GEMMs
            GM.LINEAR_PRECISION = prec
            G = {k: torch.zeros_like(v) for k, v in P.items()}
            y, saved_ctx = LY._ff_fwd(P, p, x, M, 0.2, 11, 12)
            dx = LY._ff_bwd(P, G, p, saved_ctx, dy, M)
            torch.cuda.synchronize()
            res.append((y, dx, G))
    finally:
        GM.LINEAR_PRECISION = saved
    assert relerr(res[0][0], res[1][0]) < 3e-6 and relerr(res[0][1], res[1][1]) < 5e-6
    for k in P:
        assert relerr(res[0][2][k], res[1][2][k]) < 2e-5, k


def test_gemm_ln_bwd_vs_two_kernels_at_bench_size():
    from speech_enhancement_amd import gemm as GM, ops as O
    from speech_enhancement_amd.weights import WeightPlan
    M = B * T * Fp
    x, dqkv, dR = rnd(M, 64, seed=1), rnd(M, 192, seed=2), rnd(M, 64, seed=3)
    W = rnd(192, 64, seed=4, scale=0.1)
    gam = rnd(64, seed=5) * 0.1 + 1
    st = O.row_stats(x, M)
    plan = WeightPlan(torch.device('cuda'))
    WT = plan.linear_T('wt', W, planes=True)
    plan.run()
    dg, db = torch.zeros(64, device='cuda'), torch.zeros(64, device='cuda')
    dX = GM.gemm_ln_bwd(dqkv, WT, x, st, gam, dR, dg, db)
    dl = torch.empty(M, 64, device='cuda')
    GM.gemm_tap(GM.linear_desc(M, 192, 64, precision=2), dqkv, WT, dl)
    dg2, db2 = torch.zeros(64, device='cuda'), torch.zeros(64, device='cuda')
    dX2 = O.layernorm_bwd(x, st, gam, dl, dg2, db2, dR=dR)
    assert relerr(dX, dX2) < 2e-6 and relerr(dg, dg2) < 2e-5 and relerr(db, db2) < 2e-5


def _models():
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden'))
    import formula
    import speech_enhancement_amd as S
    g, d = S.TSCNet(64, 201), S.Discriminator(16)
    g.load_state_dict(formula.formula_state('generator'))
    d.load_state_dict(formula.formula_state('discriminator'))
    g.cuda().train().set_dropout(0.0, 0.0)
    d.cuda().train()
    for m in d.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    return g, d


def test_batch16_forward_equals_single_clips_in_eval_mode():
    """eval mode (BatchNorm on running statistics): the clips of a batch are independent, so the batch-16 forward -- the grids the
    bench launches -- must reproduce the batch-1 forward (pinned against the reference by test_full_size_enhanced_magnitude)"""
    from speech_enhancement_amd import frontend as FE
    g, _ = _models()
    g.eval()
    x = rnd(B, 32000, seed=1, scale=0.1)
    with torch.no_grad():
        pl, _ = FE.stft_planes(x, 400, 100, 'pow')
        est = g.forward_planes(pl)
        for i in (0, 9, 15):
            one = g.forward_planes(pl[i:i + 1].contiguous())
            assert float((est[i] - one[0]).abs().max()) < 2e-5 * float(one.abs().max()), i


def test_streams_do_not_change_the_full_size_step():
    """one cmgan train step at the bench configuration (batch 16, 2 s clips) with the three HIP streams on vs the serial order:
    every parameter of both models relative to the size of its update (see test_model_gpu.test_streams_do_not_change_the_step)"""
    import types
    from speech_enhancement_amd import train as TR, optim, gemm as GM
    clean = rnd(B, 32000, seed=1, scale=0.1)
    noisy = clean + rnd(B, 32000, seed=2, scale=0.05)
    labels = {'est': 0.2 + 0.7 * torch.rand(B, device='cuda', generator=torch.Generator(device='cuda').manual_seed(3))}
    saved = (GM._LeafStream.enabled, TR._D_OVERLAP, GM.branch_stream.enabled)
    res, init = [], None
    try:
        for on in (False, True):
            GM._LeafStream.enabled, TR._D_OVERLAP, GM.branch_stream.enabled = on, on, on
            g, d = _models()
            named = lambda: list(g.named_parameters()) + [('D.' + n, p) for n, p in d.named_parameters()]
            if init is None:
                init = {n: p.detach().clone() for n, p in named()}
            a = types.SimpleNamespace(optimizer='sgd', lr=0.01, weight_decay=0.01, momentum=0.9, max_norm=0.0)
            og, od = optim.build_optimizer(a, g), optim.build_optimizer(a, d, lr=0.02)
            out = TR.gan_step(g, d, og, od, clean, noisy, 'cmgan', (0.1, 0.9, 0.2, 0.05), labels=labels)
            torch.cuda.synchronize()
            res.append(({k: float(v) for k, v in out.items()}, {n: p.detach().clone() for n, p in named()}))
    finally:
        GM._LeafStream.enabled, TR._D_OVERLAP, GM.branch_stream.enabled = saved
    for k, v in res[0][0].items():
        assert abs(v - res[1][0][k]) <= 1e-4 * abs(v) + 1e-7, (k, v, res[1][0][k])
    bad = []
    for n, p0 in res[0][1].items():
        upd = float((p0 - init[n]).abs().max())
        diff = float((p0 - res[1][1][n]).abs().max())
        if diff > 1e-2 * upd + 1e-6 * max(1.0, float(p0.abs().max())):
            bad.append((n, diff, upd))
    assert not bad, bad[:10]


# ---- round 4: config 3's per-rank workload under a check ---------------------------------------------
@pytest.mark.parametrize('comp', ['pow', 'log'])
def test_full_size_stft_backward_vs_oracle_fp64(comp):
    """frontend._STFTFn.backward at the bench geometry (B = 2, L = 32 000: T = 321 frames; every scp / cp step runs it,
    core/function.py:231-249) against the oracle's fp64 autograd on the well-conditioned clip (formula.cond_signals); bar as in
    tests/test_consistency_gpu.py: 2e-5 of max |dx| + 1.5 x the torch-CPU fp32 floor of the same chain"""
    from speech_enhancement_amd import frontend as FE
    from oracle import se_oracle as Or
    _, noisy = formula.cond_signals(2, 32000, 31)
    x0 = noisy.double()
    x0 = (x0 * torch.sqrt(x0.shape[-1] / (x0 ** 2).sum(-1, keepdim=True)))          # the normalised clip the loop re-analyses
    Bq, T = 2, 321
    k = torch.arange(Bq * 201 * T, dtype=torch.float64).view(Bq, 201, T)
    wr, wi, wm = torch.cos(k * 0.013), torch.sin(k * 0.017), torch.cos(k * 0.0071 + 1.0)
    x64 = x0.clone().requires_grad_(True)
    s64 = Or.compressed_stft(x64, comp=comp)
    (s64.real * wr + s64.imag * wi + s64.abs() * wm).sum().backward()
    dref = x64.grad.numpy()
    x32 = x0.float().clone().requires_grad_(True)
    s32 = Or.compressed_stft(x32, comp=comp)
    (s32.real * wr.float() + s32.imag * wi.float() + s32.abs() * wm.float()).sum().backward()
    floor = float(np.abs(x32.grad.double().numpy() - dref).max())
    x = x0.float().cuda().requires_grad_(True)
    P = FE.stft_planes_grad(x, 400, 100, comp)                             # [B, T, F, 4] = (|z|, Re, Im, 0)
    f = lambda w: w.permute(0, 2, 1).float().contiguous().cuda()
    fe = float((P[..., 1].detach().permute(0, 2, 1).double().cpu() - s64.real.detach()).abs().max()) / float(s64.detach().abs().max())
    f32e = float((s32.real.detach().double() - s64.real.detach()).abs().max()) / float(s64.detach().abs().max())
    print(comp, 'forward max err / max |z|', fe, '(torch-CPU fp32:', f32e, ')')
    assert fe < 5e-6 + 2.0 * f32e
    (P[..., 1] * f(wr) + P[..., 2] * f(wi) + P[..., 0] * f(wm)).sum().backward()
    err = float(np.abs(x.grad.double().cpu().numpy() - dref).max())
    print(comp, 'full-size dx max err', err, 'of max', float(np.abs(dref).max()), '(torch-CPU fp32 floor:', floor, ')')
    assert err < 2e-5 * float(np.abs(dref).max()) + 1.5 * floor


def test_scp_full_size_step_streams_equal_serial():
    """BASELINE config 3's per-rank share: ONE scp train step at B = 8, T = 321 (self-correcting weights, the consistency-preserving
    iSTFT / re-STFT path, three discriminator forwards) with the three HIP streams on vs the serial order: every loss finite and
    equal, w_E / w_N equal, every parameter relative to the size of its update (one step: the two-step comparison of scp is chaotic
    on white noise, DESIGN_APPENDIX A; the clips here are the well-conditioned ones)"""
    import types
    from speech_enhancement_amd import train as TR, optim, gemm as GM
    B8 = 8
    clean, noisy = formula.cond_signals(B8, 32000, 41)
    clean, noisy = clean.cuda(), noisy.cuda()
    gen = torch.Generator().manual_seed(5)
    labels = {k: (lo + (hi - lo) * torch.rand(B8, generator=gen)).cuda() for k, (lo, hi) in
              (('est', (0.3, 0.7)), ('clean', (0.9, 1.0)), ('noisy', (0.1, 0.5)))}
    saved = (GM._LeafStream.enabled, TR._D_OVERLAP, GM.branch_stream.enabled)
    res, init = [], None
    try:
        for on in (False, True):
            GM._LeafStream.enabled, TR._D_OVERLAP, GM.branch_stream.enabled = on, on, on
            g, d = _models()
            named = lambda: list(g.named_parameters()) + [('D.' + n, p) for n, p in d.named_parameters()]
            if init is None:
                init = {n: p.detach().clone() for n, p in named()}
            a = types.SimpleNamespace(optimizer='sgd', lr=0.01, weight_decay=0.01, momentum=0.9, max_norm=0.0)
            og, od = optim.build_optimizer(a, g), optim.build_optimizer(a, d, lr=0.02)
            out = TR.gan_step(g, d, og, od, clean, noisy, 'scp', (0.3, 0.7, 0.2, 0.05), labels=labels)
            torch.cuda.synchronize()
            res.append(({k: float(v) for k, v in out.items()}, {n: p.detach().clone() for n, p in named()}))
    finally:
        GM._LeafStream.enabled, TR._D_OVERLAP, GM.branch_stream.enabled = saved
    print('scp B=8 T=321 step:', {k: float('%.5g' % v) for k, v in res[0][0].items()})
    for k, v in res[0][0].items():
        assert np.isfinite(v), (k, v)
        assert abs(v - res[1][0][k]) <= 1e-4 * abs(v) + 1e-7, (k, v, res[1][0][k])
    assert 'w_E' in res[0][0] and 'w_N' in res[0][0]
    bad = []
    for n, p0 in res[0][1].items():
        upd = float((p0 - init[n]).abs().max())
        diff = float((p0 - res[1][1][n]).abs().max())
        if diff > 1e-2 * upd + 1e-6 * max(1.0, float(p0.abs().max())):
            bad.append((n, diff, upd))
    assert not bad, bad[:10]


def test_cp_full_size_step_vs_reference_loop(golden4):
    """ONE FULL-SIZE train_gan step of the REFERENCE for the consistency-preserving recipe `cp` (core/function.py:231-254; fp64
    golden, its own fp32 run gives the rounding spread) on the conditioned pair, B = 2, T = 321, vs gan_step: loss terms, every
    post-step parameter norm, the gradients of six generator and two discriminator tensors (first nesterov step: u = -1.9 lr g)"""
    import types
    from speech_enhancement_amd import train as TR, optim
    from oracle import se_oracle as Or
    clean, noisy = formula.cond_signals(2, 32000, int(golden4['cp_full_seed'][0]))
    g, d = _models()
    args = types.SimpleNamespace(optimizer='sgd', lr=0.01, weight_decay=0.01, momentum=0.9, max_norm=0.0)
    og, od = optim.build_optimizer(args, g), optim.build_optimizer(args, d, lr=0.02)
    lr = Or.lr_at(10.0, 0.01, 100)
    for o in (og, od):
        for grp in o.param_groups:
            grp['lr'] = lr
    out = TR.gan_step(g, d, og, od, clean.cuda(), noisy.cuda(), 'cp', (0.1, 0.9, 0.2, 0.05),
                      labels={'est': torch.tensor([0.35, 0.62], device='cuda')})
    torch.cuda.synchronize()
    pre, pre32 = 'cp_full_f64_', 'cp_full_f32_'
    mse = golden4[pre + 'mse_calls']
    for k, b in (('loss_mag', mse[0]), ('loss_ri', mse[1] + mse[2]), ('gan', mse[3]), ('L_E', mse[4]), ('L_C', mse[5]),
                 ('loss_g', golden4[pre + 'losses'][0]), ('loss_d', golden4[pre + 'losses'][1])):
        e = abs(float(out[k]) - b) / abs(b)
        assert e < (2e-4 if k in ('loss_mag', 'loss_ri', 'loss_g') else 1e-3), (k, float(out[k]), b)
    gs, ds = g.state_dict(), d.state_dict()
    gnorm = np.array([float(v.double().norm()) for v in gs.values()])
    dnorm = np.array([float(v.double().norm()) for v in ds.values()])
    ref_g, ref_d = golden4[pre + 'g_norm'], golden4[pre + 'd_norm']
    tol_g = 3e-4 * ref_g + 1e-5 + 1.5 * np.abs(golden4[pre32 + 'g_norm'] - ref_g)
    tol_d = 3e-4 * ref_d + 1e-5 + 1.5 * np.abs(golden4[pre32 + 'd_norm'] - ref_d)
    names_g = list(gs.keys())
    bad = [(names_g[i], gnorm[i], ref_g[i]) for i in range(len(ref_g)) if abs(gnorm[i] - ref_g[i]) > tol_g[i]]
    assert not bad, bad[:8]
    assert np.all(np.abs(dnorm - ref_d) <= tol_d)
    gp, dp = dict(g.named_parameters()), dict(d.named_parameters())
    scale = -1.0 / (lr * 1.9)
    rmsf = lambda a, b: float(np.sqrt(np.mean((np.asarray(a.detach().cpu() if torch.is_tensor(a) else a, np.float64) - b) ** 2)))
    for k in golden4.files:
        if not (k.startswith(pre + 'gupd:') or k.startswith(pre + 'dupd:')):
            continue
        name = k.split(':', 1)[1]
        gref = golden4[k].astype(np.float64) * scale
        nrm = np.sqrt(np.mean(gref ** 2)) + 1e-300
        spread = rmsf(golden4[pre32 + k[len(pre):]].astype(np.float64) * scale, gref)
        got = (gp if k.startswith(pre + 'gupd:') else dp)[name].grad
        e = rmsf(got, gref)
        print('cp full size', name, 'grad rel err %.2e (reference fp32 spread %.2e)' % (e / nrm, spread / nrm))
        # (fp32 storage of the reference's updates: relative 6e-8 of the PARAMETER, i.e. up to a few 1e-3 of a tiny update)
        store = 1.2e-7 * float(np.abs((gp if 'gupd' in k else dp)[name].detach().cpu().numpy()).max()) * abs(scale)
        assert e < 2e-4 * nrm + 1.5 * spread + store, (name, e / nrm, spread / nrm)
