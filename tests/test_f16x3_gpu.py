"""Kernel-level parity of the scaled split-fp16 arithmetic (se_gemm_desc.precision 3, the default of the train step) against
fp64 torch: every kernel family that has an F16 form, on its own, at the bar of the fp32 forms (the model-level tests run the
same kernels against the reference's goldens).  Operands carry realistic dynamic ranges: gradients at 1e-4 .. 1e-3 with their
producer-measured maximum (`_se_amax`), activations under the static exponents of layers.py / gemm.py."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).cuda()


def relerr(a, b):
    a, b = a.detach().double(), b.detach().double()
    return float((a - b).abs().max() / max(float(b.abs().max()), 1e-30))


@pytest.fixture(scope='module')
def env():
    from speech_enhancement_amd import gemm as GM, _lib as L, layers as LY, ops as O
    from speech_enhancement_amd.weights import WeightPlan
    return GM, L, LY, O, WeightPlan


def _amax(t):
    t._se_amax = t.abs().max().reshape(1).clone()
    return t


@pytest.mark.parametrize('i', [0, 2, 3])
def test_dense_conv_forward_backward(env, i):
    """DilatedDenseNet conv i + 1 (generator.py:6-32) from the 256-wide skip stack: conv3 forward, triple-tap weight gradient,
    input gradient accumulated into the gradient stack -- all three on two scaled fp16 planes."""
    GM, L, LY, O, WeightPlan = env
    B, T, Fq, C_in, dil = 2, 11, 101, 64 * (i + 1), 2 ** i
    skip = rnd(B, T, Fq, 256, seed=1)
    w = rnd(64, C_in, 2, 3, seed=2, scale=(C_in * 6) ** -0.5)
    b = rnd(64, seed=3, scale=0.1)
    plan = WeightPlan(torch.device('cuda'))
    wp = plan.conv_fwd('f', w, rev=True, planes='f16')
    wd = plan.conv_dgrad('d', w, rev=True, planes='f16')
    plan.run()
    assert wp.dtype == torch.float16 and wd.dtype == torch.float16
    # the reference concatenates newest-first (torch.cat([out, skip], 1)); the stack stores oldest-first
    x64 = torch.cat([skip[..., 64 * j:64 * (j + 1)] for j in reversed(range(i + 1))], -1).double().requires_grad_(True)
    w64, b64 = w.double().requires_grad_(True), b.double().requires_grad_(True)
    ref = F.conv2d(F.pad(x64.permute(0, 3, 1, 2), (1, 1, dil, 0)), w64, b64, dilation=(dil, 1)).permute(0, 2, 3, 1)
    R, stats = LY.conv_fwd(skip, B, T, Fq, 256, 0, C_in, wp, b, LY.dense_taps(i), 64)
    assert relerr(R, ref) < 3e-6
    assert relerr(stats[..., 0], ref.sum((1, 2))) < 1e-5
    dR = _amax(rnd(B, T, Fq, 64, seed=4, scale=3e-4))
    ref.backward(dR.double())
    dw, db = torch.zeros_like(w), torch.zeros_like(b)
    dskip = rnd(B, T, Fq, 256, seed=5, scale=1e-4)
    dskip0 = dskip.clone()
    LY.conv_bwd(skip, B, T, Fq, 256, 0, C_in, w, LY.dense_taps(i), dR, T, Fq, dw, db, rev=True, dx=dskip, lddx=256, dx_off=0,
                accumulate=True, wd=wd)
    torch.cuda.synchronize()
    assert relerr(dw, w64.grad) < 5e-6 and relerr(db, b64.grad) < 5e-6
    dx_ref = torch.cat([x64.grad[..., 64 * (i - j):64 * (i - j + 1)] for j in range(i + 1)], -1)
    assert relerr(dskip[..., :C_in] - dskip0[..., :C_in], dx_ref) < 5e-6
    assert torch.equal(dskip[..., C_in:], dskip0[..., C_in:])


def test_strided_and_subpixel_conv(env):
    """encoder conv_2 (stride 2 along F) and the sub-pixel conv (shuffle epilogue) through the generic tap kernel on fp16 planes"""
    GM, L, LY, O, WeightPlan = env
    B, T, Fq = 2, 9, 201
    a = rnd(B, T, Fq, 64, seed=1)
    wc, bc = rnd(64, 64, 1, 3, seed=2, scale=0.07), rnd(64, seed=3, scale=0.1)
    ws, bs = rnd(128, 64, 1, 3, seed=4, scale=0.07), rnd(128, seed=5, scale=0.1)
    plan = WeightPlan(torch.device('cuda'))
    pc, ps = plan.conv_fwd('c', wc, planes='f16'), plan.conv_fwd('s', ws, planes='f16')
    plan.run()
    ref = F.conv2d(a.double().permute(0, 3, 1, 2), wc.double(), bc.double(), stride=(1, 2), padding=(0, 1)).permute(0, 2, 3, 1)
    R, _ = LY.conv_fwd(a, B, T, Fq, 64, 0, 64, pc, bc, LY.TAPS_1x3, 64, To=T, Fo=101, sf=2)
    assert relerr(R, ref) < 3e-6
    a2 = rnd(B, T, 101, 64, seed=6)
    conv = F.conv2d(a2.double().permute(0, 3, 1, 2), ws.double(), bs.double(), padding=(0, 1))
    ref = conv.view(B, 2, 64, T, 101).permute(0, 2, 3, 4, 1).reshape(B, 64, T, 202).permute(0, 2, 3, 1)
    R, _ = LY.conv_fwd(a2, B, T, 101, 64, 0, 64, ps, bs, LY.TAPS_1x3, 128, shuffle2=True, want_stats=False)
    assert relerr(R, ref) < 3e-6


def _ln64(x, g, b):
    return F.layer_norm(x.double(), (64,), g.double(), b.double(), 1e-5)


@pytest.mark.parametrize('M', [4096 + 37, 256 * 1024 + 293])
def test_row_panels_and_glu_gate(env, M):
    """K = 64 row panels with the LayerNorm prologue: qkv (N = 192) and the first pointwise conv with the GLU epilogue keeping
    only the gate half (conformer.py:103-108,160-166); from 256 K rows on: the W-stationary persistent form (ragged last tile)"""
    GM, L, LY, O, WeightPlan = env
    x = rnd(M, 64, seed=1) * 1.5 + 0.2
    st = O.row_stats(x, M)
    g, b = rnd(64, seed=2) * 0.2 + 1.0, rnd(64, seed=3) * 0.1
    Wq = rnd(192, 64, seed=4, scale=0.1)
    Wp, bp = rnd(256, 64, seed=5, scale=0.1), rnd(256, seed=6, scale=0.1)
    plan = WeightPlan(torch.device('cuda'))
    pq, pp = plan.linear('q', Wq, planes='f16'), plan.linear('p', Wp, planes='f16')
    plan.run()
    xl = _ln64(x, g, b)
    q = torch.empty(M, 192, device='cuda')
    GM.gemm_tap(GM.linear_desc(M, 64, 192, prologue=L.PRO_LN, **LY._lin3(pq, a_sexp=GM.LN_SEXP)), x, pq, q, rowstats=st, ps=g, pb=b)
    assert relerr(q, xl @ Wq.double().t()) < 3e-6
    u, gate = torch.empty(M, 128, device='cuda'), torch.empty(M, 128, device='cuda')
    GM.gemm_tap(GM.linear_desc(M, 64, 256, ldc=128, prologue=L.PRO_LN, epilogue=L.EPI_BIAS | L.EPI_GLU | L.EPI_GLU_GATE, ldx=128,
                               **LY._lin3(pp, a_sexp=GM.LN_SEXP)), x, pp, u, bias=bp, AUX=gate, rowstats=st, ps=g, pb=b)
    z = xl @ Wp.double().t() + bp.double()
    assert relerr(gate, z[:, 128:]) < 3e-6 and relerr(u, z[:, :128] * torch.sigmoid(z[:, 128:])) < 3e-6


def test_pointwise_conv_behind_batchnorm_swish(env):
    """second pointwise conv of the Conformer conv module, 128 -> 64, with the BatchNorm-apply + Swish prologue, bias, residual and
    the fused row statistics (conformer.py:167-170): scaled split-fp16 generic row GEMM under the static exponent of the hidden
    activations; ragged last tile; an outlier of 60 standard deviations stays exact (|x| < 8191)"""
    GM, L, LY, O, WeightPlan = env
    M = 128 * 9 + 77
    h = rnd(M, 128, seed=1) * 2.0 + 0.3
    h[5, 17] = 130.0
    sc, sh = rnd(128, seed=2) * 0.2 + 0.5, rnd(128, seed=3) * 0.2
    W, b = rnd(64, 128, seed=4, scale=0.1), rnd(64, seed=5, scale=0.1)
    R = rnd(M, 64, seed=6)
    plan = WeightPlan(torch.device('cuda'))
    pw = plan.linear('w', W, planes='f16')
    plan.run()
    y, st = torch.empty(M, 64, device='cuda'), torch.empty(M, 2, device='cuda')
    GM.gemm_tap(GM.linear_desc(M, 128, 64, prologue=L.PRO_AFFINE_SWISH, epilogue=L.EPI_BIAS | L.EPI_RESID | L.EPI_ROWSTATS, alpha=1.0,
                               ldr=64, **LY._lin3(pw, a_sexp=GM.HID_SEXP)), h, pw, y, bias=b, R=R, ps=sc, pb=sh, AUX=st)
    z = h.double() * sc.double() + sh.double()
    ref = (z * torch.sigmoid(z)) @ W.double().t() + b.double() + R.double()
    assert relerr(y, ref) < 3e-6
    assert relerr(st[:, 0], ref.mean(1)) < 1e-5 and relerr(st[:, 1], 1.0 / torch.sqrt(ref.var(1, unbiased=False) + 1e-5)) < 1e-5
    y32 = torch.empty(M, 64, device='cuda')
    GM.gemm_tap(GM.linear_desc(M, 128, 64, prologue=L.PRO_AFFINE_SWISH, epilogue=L.EPI_BIAS | L.EPI_RESID, alpha=1.0, ldr=64),
                h, W, y32, bias=b, R=R, ps=sc, pb=sh)
    assert relerr(y, ref) < 4 * relerr(y32, ref) + 1e-6          # at the level of the fp32-MFMA kernel it replaces


@pytest.mark.parametrize('dil,Lp,B', [(1, 900, 2), (4, 1000, 3), (512, 900, 2), (64, 128 * 5, 1),
                                      (8, 128 * 1100 + 77, 2), (512, 128 * 700, 3)])       # >= 2048 row tiles: the W-stationary kernel
def test_dilated_conv1d_and_projection_panels_with_groupnorm_sums(env, dil, Lp, B):
    """the two GEMMs of a DiffWave residual layer (models/DiffuSE.py:98-127) on the K = 64 row panel: the dilated Conv1d(64 -> 128,
    k = 3, padding = dilation) as three shifted row fragments held in registers, and the 1 x 1 projections 64 -> 128 with batch
    entries; both with bias and the per-(entry, channel) fp64 sums of GroupNorm (SE_EPI_STATS); ragged last tile, dilation beyond
    the tile and beyond half the map; small maps take the row panel, long ones the persistent W-stationary kernel"""
    GM, L, LY, O, WeightPlan = env
    C = 64
    y = _amax(rnd(B, Lp, C, seed=dil) * 1.7)
    Wd, bd = rnd(2 * C, C, 3, seed=2, scale=0.08), rnd(2 * C, seed=3, scale=0.1)
    W2, b2 = rnd(2 * C, C, seed=4, scale=0.1), rnd(2 * C, seed=5, scale=0.1)
    plan = WeightPlan(torch.device('cuda'))
    pd = plan.conv_fwd('d', Wd.unsqueeze(2).contiguous(), planes='f16')
    p2 = plan.linear('2', W2, planes='f16')
    plan.run()
    taps = [(0, -dil), (0, 0), (0, dil)]
    R, st = torch.empty(B, Lp, 2 * C, device='cuda'), torch.zeros(B, 2 * C, 2, device='cuda', dtype=torch.float64)
    d = GM.make_desc(B, 1, Lp, 1, Lp, taps, C, C, 2 * C, 2 * C, epilogue=L.EPI_BIAS | L.EPI_STATS, precision=3, a_amax=y._se_amax)
    GM.gemm_tap(d, y, pd, R, bias=bd, stats=st)
    ref = F.conv1d(y.double().transpose(1, 2), Wd.double(), bd.double(), padding=dil, dilation=dil).transpose(1, 2)
    assert relerr(R, ref) < 3e-6
    assert relerr(st[..., 0], ref.sum(1)) < 1e-5 and relerr(st[..., 1], (ref * ref).sum(1)) < 1e-5
    g = torch.tanh(rnd(B, Lp, C, seed=7))                      # the gate output lies in (-1, 1): static exponent 13
    R2, st2 = torch.empty(B, Lp, 2 * C, device='cuda'), torch.zeros(B, 2 * C, 2, device='cuda', dtype=torch.float64)
    d2 = GM.make_desc(B, 1, Lp, 1, Lp, [(0, 0)], C, C, 2 * C, 2 * C, epilogue=L.EPI_BIAS | L.EPI_STATS, precision=3, a_sexp=13)
    GM.gemm_tap(d2, g, p2, R2, bias=b2, stats=st2)
    ref2 = g.double() @ W2.double().t() + b2.double()
    assert relerr(R2, ref2) < 3e-6
    assert relerr(st2[..., 0], ref2.sum(1)) < 1e-5 and relerr(st2[..., 1], (ref2 * ref2).sum(1)) < 1e-5


def test_feed_forward_pair(env):
    """fused feed-forward forward / input-gradient chain (conformer.py:53-71,128-145) on fp16 planes vs fp64"""
    GM, L, LY, O, WeightPlan = env
    M = 4096 + 37
    x = rnd(M, 64, seed=1) * 1.5 + 0.2
    st = O.row_stats(x, M)
    g, b = rnd(64, seed=2) * 0.2 + 1.0, rnd(64, seed=3) * 0.1
    W1, b1 = rnd(256, 64, seed=4, scale=0.1), rnd(256, seed=5, scale=0.1)
    W2, b2 = rnd(64, 256, seed=6, scale=0.1), rnd(64, seed=7, scale=0.1)
    plan = WeightPlan(torch.device('cuda'))
    p1, p2 = plan.linear('w1', W1, planes='f16'), plan.linear('w2', W2, planes='f16')
    p2t, p1t = plan.linear_T('w2t', W2, planes='f16', scale=0.5), plan.linear_T('w1t', W1, planes='f16')
    plan.run()
    y, h = GM.ff_fwd(x, st, g, b, p1, b1, p2, b2, 0.0, 1, 2, 0.5)
    x64 = x.double().requires_grad_(True)
    h64 = _ln64(x64, g, b) @ W1.double().t() + b1.double()
    y64 = x64 + 0.5 * ((h64 * torch.sigmoid(h64)) @ W2.double().t() + b2.double())
    assert relerr(h, h64) < 3e-6 and relerr(y, y64) < 3e-6
    dy = _amax(rnd(M, 64, seed=8, scale=2e-4))
    y64.backward(dy.double())
    dg, db = torch.zeros(64, device='cuda'), torch.zeros(64, device='cuda')
    am = (torch.zeros(1, device='cuda'), torch.zeros(1, device='cuda'))
    dz, dx = GM.ff_bwd_dgrad(dy, h, p2t, p1t, 0.0, 1, 2, ln=(x, st, g, None, dg, db), amax_out=am)
    s = torch.sigmoid(h64.detach())
    dz64 = (dy.double() @ (0.5 * W2.double())) * (s * (1 + h64.detach() * (1 - s)))
    assert relerr(dz, dz64) < 5e-6 and relerr(dx, x64.grad) < 5e-6
    assert abs(float(am[1]) - float(dz.abs().max())) <= 2e-3 * float(am[1])          # (tracked on the fp16 hi plane)
    assert abs(float(am[0]) - float(dx.abs().max())) <= 1e-6 * float(am[0])


@pytest.mark.parametrize('Cin,N,pro', [(64, 256, 'ln'), (256, 64, 'swish_drop'), (64, 192, 'ln'), (128, 64, 'affine_swish'),
                                       (64, 64, 'none_drop'), (64, 200, 'none')])
def test_token_wise_weight_gradients(env, Cin, N, pro):
    """whole-gradient row kernels on two fp16 planes (wgrad_lin_bf16_kernel<.., F16>): the feed-forward / pointwise-conv shapes
    and the narrower ones that share the kernel with idle waves (second pointwise conv, to_out, qkv), ragged last chunk"""
    GM, L, LY, O, WeightPlan = env
    M = 5000 + 37
    x, dy = rnd(M, Cin, seed=1), _amax(rnd(M, N, seed=2, scale=3e-4))
    st = O.row_stats(x, M) if Cin == 64 else None
    g, b = rnd(Cin, seed=3) * 0.1 + 1, rnd(Cin, seed=4) * 0.1
    code = {'ln': L.PRO_LN, 'swish_drop': L.PRO_SWISH_DROP, 'affine_swish': L.PRO_AFFINE_SWISH, 'none': L.PRO_NONE,
            'none_drop': L.PRO_NONE}[pro]
    dp = 0.2 if pro.endswith('drop') else 0.0
    sexp = {'ln': GM.LN_SEXP, 'swish_drop': GM.HID_SEXP, 'affine_swish': GM.HID_SEXP}.get(pro, LY.ATTN_O_SEXP)
    out = {}
    for prec in (0, 3):
        d = GM.linear_desc(M, Cin, N, prologue=code, epilogue=L.EPI_DROP if dp else 0, pro_seed=3, epi_seed=9, drop_p=dp,
                           precision=prec, a_sexp=sexp, w_amax=dy._se_amax)
        dw, db = torch.zeros(N, Cin, device='cuda'), torch.zeros(N, device='cuda')
        GM.gemm_tap_wgrad(d, x, dy, dw, db, rowstats=st, ps=g, pb=b, explicit_precision=True)
        out[prec] = (dw, db)
    # the fp32-MFMA kernel restates the prologue / dropout hash: the fp16 planes must agree with it at fp32 accuracy
    assert relerr(out[3][0], out[0][0]) < 5e-6 and relerr(out[3][1], out[0][1]) < 5e-6
    if pro in ('ln', 'none'):
        a = _ln64(x, g, b) if pro == 'ln' else x.double()
        assert relerr(out[3][0], dy.double().T @ a) < 5e-6 and relerr(out[3][1], dy.double().sum(0)) < 5e-6


def test_input_gradient_gemm_with_layernorm_backward(env):
    """se_gemm_ln_bwd on fp16 planes with a measured operand maximum (pointwise-conv input gradient, K = 256)"""
    GM, L, LY, O, WeightPlan = env
    M, K = 128 * 9 + 45, 256
    x, dR = rnd(M, 64, seed=1) * 1.5 + 0.3, rnd(M, 64, seed=3, scale=2e-4)
    dy = _amax(rnd(M, K, seed=2, scale=2e-4))
    W = rnd(K, 64, seed=4, scale=0.1)
    gam, bet = rnd(64, seed=5) * 0.2 + 1.0, rnd(64, seed=6) * 0.1
    st = O.row_stats(x, M)
    plan = WeightPlan(torch.device('cuda'))
    WT = plan.linear_T('wt', W, planes='f16')
    plan.run()
    dg, db = torch.zeros(64, device='cuda'), torch.zeros(64, device='cuda')
    am = torch.zeros(1, device='cuda')
    dX = GM.gemm_ln_bwd(dy, WT, x, st, gam, dR, dg, db, out_amax=am)
    x64 = x.double().requires_grad_(True)
    g64, b64 = gam.double().requires_grad_(True), bet.double().requires_grad_(True)
    (F.layer_norm(x64, (64,), g64, b64, 1e-5) @ W.double().T).backward(dy.double())
    assert relerr(dX, x64.grad + dR.double()) < 5e-6 and relerr(dg, g64.grad) < 1e-5 and relerr(db, b64.grad) < 1e-5
    assert abs(float(am) - float(dX.abs().max())) <= 1e-6 * float(am)


def test_output_maximum_of_the_vector_epilogue(env):
    """se_gemm_desc.y_amax: the scalar a GEMM raises to max |Y| (operand scale of the scaled split-fp16 attention: max |qkv| from the
    row panel, max |dO| from the generic row GEMM with the dropout prologue), ragged last tile, a running maximum on entry"""
    GM, L, LY, O, WeightPlan = env
    M = 128 * 6 + 51
    x = rnd(M, 64, seed=1) * 1.5 + 0.2
    st = O.row_stats(x, M)
    g, b = rnd(64, seed=2) * 0.2 + 1.0, rnd(64, seed=3) * 0.1
    Wq, Wo = rnd(192, 64, seed=4, scale=0.3), rnd(64, 64, seed=5, scale=0.3)
    plan = WeightPlan(torch.device('cuda'))
    pq, po = plan.linear('q', Wq, planes='f16'), plan.linear_T('o', Wo, planes='f16')
    plan.run()
    am = torch.zeros(1, device='cuda')
    q = torch.empty(M, 192, device='cuda')
    GM.gemm_tap(GM.linear_desc(M, 64, 192, prologue=L.PRO_LN, y_amax=am, **LY._lin3(pq, a_sexp=GM.LN_SEXP)), x, pq, q, rowstats=st, ps=g, pb=b)
    assert float(am) == float(q.abs().max())
    dy = _amax(rnd(M, 64, seed=6, scale=2e-4))
    am2 = torch.full((1,), 1e-9, device='cuda')             # a smaller running maximum on entry is raised
    do = torch.empty(M, 64, device='cuda')
    GM.gemm_tap(GM.linear_desc(M, 64, 64, prologue=L.PRO_DROP, pro_seed=3, drop_p=0.2, y_amax=am2, **LY._lin3(po, a_amax=dy._se_amax)),
                dy, po, do)
    assert float(am2) == float(do.abs().max())
    am3 = torch.full((1,), 1e9, device='cuda')              # a larger one is kept
    GM.gemm_tap(GM.linear_desc(M, 64, 64, y_amax=am3, **LY._lin3(po, a_amax=dy._se_amax)), dy, po, do)
    assert float(am3) == 1e9
    with pytest.raises(L.SeHipError):                       # the GLU / shuffle epilogues do not track it
        GM.gemm_tap(GM.linear_desc(M, 64, 256, ldc=128, epilogue=L.EPI_GLU, y_amax=am), x, rnd(256, 64, seed=7), torch.empty(M, 128, device='cuda'))
