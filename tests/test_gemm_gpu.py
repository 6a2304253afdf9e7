"""GPU parity of the tap-GEMM family against plain PyTorch fp32/fp64 references of the same op."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def G():
    from speech_enhancement_amd import gemm, _lib
    return gemm, _lib


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).cuda()


def relerr(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


@pytest.mark.parametrize('M,Cin,N', [(1000, 64, 64), (257, 64, 192), (4096, 256, 64), (130, 128, 64),
                                      (100, 16, 32), (100, 4, 16), (333, 400, 402)])
def test_linear(G, M, Cin, N):
    gemm, L = G
    x, w, b = rnd(M, Cin, seed=1), rnd(N, Cin, seed=2, scale=Cin ** -0.5), rnd(N, seed=3)
    y = torch.empty(M, N, device='cuda')
    gemm.gemm_tap(gemm.linear_desc(M, Cin, N, epilogue=L.EPI_BIAS), x, w, y, bias=b)
    ref = x.double() @ w.double().T + b.double()
    assert relerr(y, ref) < 2e-6


@pytest.mark.parametrize('M,drop', [(1000, 0.0), (4096 + 37, 0.2), (130, 0.2), (300001, 0.0), (300001, 0.2)])      # (>= 2048 row tiles: the shapes the W-stationary kernels would take)
def test_delta_epilogue_of_the_to_out_input_gradient(G, M, drop):
    """SE_EPI_DELTA: the row GEMM that produces dO = (mask dY) Wo also writes delta[m][h] = sum over head h's 16 columns of dO * O
    (the softmax-backward row constants the attention backward reads) -- fp32 kernel, split-fp16 generic kernel (no dropout) and the
    K = 64 row panel with its early second-operand fetch (dropout prologue), against fp64"""
    gemm, L = G
    from speech_enhancement_amd.weights import WeightPlan
    dy, w, o = rnd(M, 64, seed=1, scale=1e-3), rnd(64, 64, seed=2, scale=0.125), rnd(M, 64, seed=3)
    ref_do = dy.double() @ w.double().T
    plan = WeightPlan(torch.device('cuda'))
    w16 = plan.linear('w', w, planes='f16')
    plan.run()
    for W, kw in ((w, {}), (w16, dict(precision=3, a_amax=dy.abs().max().reshape(1).clone()))):
        if drop > 0 and not kw:
            continue
        do, delta = torch.empty(M, 64, device='cuda'), torch.full((M, 4), float('nan'), device='cuda')
        amax = torch.zeros(1, device='cuda')
        gemm.gemm_tap(gemm.linear_desc(M, 64, 64, prologue=L.PRO_DROP if drop > 0 else L.PRO_NONE, pro_seed=77, drop_p=drop,
                                       epilogue=L.EPI_DELTA, ldr=64, y_amax=amax, **kw), dy, W, do, R=o, AUX=delta)
        if drop == 0:
            assert relerr(do, ref_do) < 2e-6
        want = (do.double() * o.double()).view(M, 4, 16).sum(-1)      # from the dO the kernel stored (mask included)
        assert relerr(delta, want) < 2e-6 and bool(torch.isfinite(delta).all())
        assert abs(float(amax) - float(do.abs().max())) <= 1e-6 * float(do.abs().max())
    with pytest.raises(Exception):          # R without its row stride is refused
        gemm.gemm_tap(gemm.linear_desc(M, 64, 64, epilogue=L.EPI_DELTA), dy, w, do, R=o, AUX=delta)


def test_prologues(G):
    gemm, L = G
    M, Cin, N = 777, 64, 256
    x, w = rnd(M, Cin, seed=1) * 2 + 0.5, rnd(N, Cin, seed=2, scale=0.125)
    g, b = rnd(Cin, seed=4) * 0.1 + 1, rnd(Cin, seed=5) * 0.1
    mean = x.mean(-1)
    rstd = (x.var(-1, unbiased=False) + 1e-5).rsqrt()
    stats = torch.stack([mean, rstd], -1).contiguous()
    y = torch.empty(M, N, device='cuda')
    gemm.gemm_tap(gemm.linear_desc(M, Cin, N, prologue=L.PRO_LN), x, w, y, rowstats=stats, ps=g, pb=b)
    ref = F.layer_norm(x.double(), (Cin,), g.double(), b.double(), 1e-5) @ w.double().T
    assert relerr(y, ref) < 5e-6
    gemm.gemm_tap(gemm.linear_desc(M, Cin, N, prologue=L.PRO_SWISH), x, w, y)
    assert relerr(y, F.silu(x.double()) @ w.double().T) < 5e-6
    gemm.gemm_tap(gemm.linear_desc(M, Cin, N, prologue=L.PRO_AFFINE_SWISH), x, w, y, ps=g, pb=b)
    assert relerr(y, F.silu(x.double() * g.double() + b.double()) @ w.double().T) < 5e-6


def test_epilogues(G):
    gemm, L = G
    M, Cin, N = 500, 64, 256
    x, w, b = rnd(M, Cin, seed=1), rnd(N, Cin, seed=2, scale=0.125), rnd(N, seed=3)
    lin = x.double() @ w.double().T + b.double()
    # GLU (+ pre-GLU Z)
    y = torch.empty(M, N // 2, device='cuda')
    z = torch.empty(M, N, device='cuda')
    d = gemm.linear_desc(M, Cin, N, ldc=N // 2, epilogue=L.EPI_BIAS | L.EPI_GLU, ldx=N)
    gemm.gemm_tap(d, x, w, y, bias=b, AUX=z)
    assert relerr(z, lin) < 2e-6
    assert relerr(y, lin[:, :N // 2] * torch.sigmoid(lin[:, N // 2:])) < 2e-6
    # GLU keeping only the gate half (what the GLU backward needs next to the result itself)
    for prec in (0, 2):                   # (scaled-fp16 planes take the same epilogue: covered by the conformer / model tests)
        y2, gt = torch.empty(M, N // 2, device='cuda'), torch.empty(M, N // 2, device='cuda')
        d = gemm.linear_desc(M, Cin, N, ldc=N // 2, epilogue=L.EPI_BIAS | L.EPI_GLU | L.EPI_GLU_GATE, ldx=N // 2,
                             precision=prec)
        gemm.gemm_tap(d, x, w, y2, bias=b, AUX=gt)
        assert relerr(gt, lin[:, N // 2:]) < 2e-6 and relerr(y2, lin[:, :N // 2] * torch.sigmoid(lin[:, N // 2:])) < 2e-6
    # residual with alpha
    r = rnd(M, N, seed=7)
    y = torch.empty(M, N, device='cuda')
    gemm.gemm_tap(gemm.linear_desc(M, Cin, N, epilogue=L.EPI_BIAS | L.EPI_RESID, alpha=0.5, ldr=N), x, w, y, bias=b, R=r)
    assert relerr(y, r.double() + 0.5 * lin) < 2e-6
    # accumulate into a slab of a wider buffer
    buf = rnd(M, 512, seed=8)
    ref = buf.double().clone()
    ref[:, 128:128 + N] += lin
    gemm.gemm_tap(gemm.linear_desc(M, Cin, N, ldc=512, c_off=128, epilogue=L.EPI_BIAS | L.EPI_ACCUM), x, w, buf, bias=b)
    assert relerr(buf, ref) < 2e-6
    # swish-grad
    zz = rnd(M, N, seed=9)
    gemm.gemm_tap(gemm.linear_desc(M, Cin, N, epilogue=L.EPI_BIAS | L.EPI_SWISH_GRAD, ldx=N), x, w, y, bias=b, AUX=zz)
    s = torch.sigmoid(zz.double())
    assert relerr(y, lin * (s * (1 + zz.double() * (1 - s)))) < 2e-6


def _conv_case(G, B, T, Fq, Cin, N, kh, kw, dil, pad, stride=(1, 1), lda=None, a_off=0, seed=0):
    gemm, L = G
    lda = lda or Cin
    xbuf = rnd(B, T, Fq, lda, seed=seed)
    x = xbuf[..., a_off:a_off + Cin]
    w = rnd(N, Cin, kh, kw, seed=seed + 1, scale=(Cin * kh * kw) ** -0.5)
    b = rnd(N, seed=seed + 2)
    ref = F.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), b.double(), stride=stride, padding=pad,
                   dilation=dil).permute(0, 2, 3, 1).contiguous()
    To, Fo = ref.shape[1], ref.shape[2]
    taps = gemm.conv_taps(kh, kw, dil, pad)
    wp = gemm.pack_conv_fwd(w)
    d = gemm.make_desc(B, To, Fo, T, Fq, taps, Cin, lda, N, N, a_off=a_off, st=stride[0], sf=stride[1],
                       epilogue=L.EPI_BIAS | L.EPI_STATS)
    y = torch.empty(B, To, Fo, N, device='cuda')
    stats = torch.zeros(B, N, 2, device='cuda', dtype=torch.float64)
    gemm.gemm_tap(d, xbuf, wp, y, bias=b, stats=stats)
    assert relerr(y, ref) < 3e-6
    assert relerr(stats[..., 0], ref.sum((1, 2))) < 1e-5
    assert relerr(stats[..., 1], (ref ** 2).sum((1, 2))) < 1e-5
    return xbuf, x, w, b, ref, taps


def test_dilated_conv_from_skip_stack(G):
    # DilatedDenseNet layer 3: dilation 4 on T, causal pad (top only), input = 3 slabs of a 256-wide stack
    gemm, L = G
    B, T, Fq = 2, 21, 37
    xbuf = rnd(B, T, Fq, 256, seed=3)
    w = rnd(64, 192, 2, 3, seed=4, scale=0.03)
    b = rnd(64, seed=5)
    x = xbuf[..., :192].permute(0, 3, 1, 2).double()
    ref = F.conv2d(F.pad(x, (1, 1, 4, 0)), w.double(), b.double(), dilation=(4, 1)).permute(0, 2, 3, 1)
    taps = gemm.conv_taps(2, 3, (4, 1), (4, 1))
    wp = gemm.pack_conv_fwd(w)
    d = gemm.make_desc(B, T, Fq, T, Fq, taps, 192, 256, 64, 64, epilogue=L.EPI_BIAS)
    y = torch.empty(B, T, Fq, 64, device='cuda')
    gemm.gemm_tap(d, xbuf, wp, y, bias=b)
    assert relerr(y, ref) < 3e-6
    # slab-reversed packing: reference channel order newest-first
    w_ref = torch.cat([w[:, 128:192], w[:, 64:128], w[:, 0:64]], 1).contiguous()
    wp2 = gemm.pack_conv_fwd(w_ref, rev_slabs=True)
    assert torch.equal(wp, wp2)


def test_conv_variants(G):
    _conv_case(G, 2, 9, 41, 64, 64, 1, 3, (1, 1), (0, 1), stride=(1, 2))          # encoder conv_2
    _conv_case(G, 2, 9, 21, 64, 128, 1, 3, (1, 1), (0, 1))                          # sub-pixel conv
    _conv_case(G, 2, 41, 33, 4, 16, 4, 4, (1, 1), (1, 1), stride=(2, 2), seed=5)    # D layer 1 (C padded to 4)
    _conv_case(G, 2, 20, 16, 16, 32, 4, 4, (1, 1), (1, 1), stride=(2, 2), seed=6)   # D layer 2
    _conv_case(G, 1, 12, 10, 64, 128, 4, 4, (1, 1), (1, 1), stride=(2, 2), seed=7)  # D layer 4


def test_shuffle2(G):
    gemm, L = G
    B, T, Fq, Cin = 2, 5, 11, 64
    x = rnd(B, T, Fq, Cin, seed=1)
    w = rnd(128, Cin, 1, 3, seed=2, scale=0.07)
    b = rnd(128, seed=3)
    conv = F.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), b.double(), padding=(0, 1))   # [B,128,T,F]
    ref = conv.view(B, 2, 64, T, Fq).permute(0, 2, 3, 4, 1).reshape(B, 64, T, 2 * Fq).permute(0, 2, 3, 1)
    d = gemm.make_desc(B, T, Fq, T, Fq, gemm.conv_taps(1, 3, (1, 1), (0, 1)), Cin, Cin, 128, 64,
                       epilogue=L.EPI_BIAS | L.EPI_SHUFFLE2 | L.EPI_STATS)
    y = torch.empty(B, T, 2 * Fq, 64, device='cuda')
    stats = torch.zeros(B, 64, 2, device='cuda', dtype=torch.float64)
    gemm.gemm_tap(d, x, gemm.pack_conv_fwd(w), y, bias=b, stats=stats)
    assert relerr(y, ref) < 3e-6
    assert relerr(stats[..., 0], ref.sum((1, 2))) < 1e-5


@pytest.mark.parametrize('case', ['dilated', 'strided', 'dconv'])
def test_conv_backward(G, case):
    gemm, L = G
    if case == 'dilated':
        B, T, Fq, Cin, N, kh, kw, dil, pad, stride = 2, 19, 23, 128, 64, 2, 3, (2, 1), (2, 1), (1, 1)
    elif case == 'strided':
        B, T, Fq, Cin, N, kh, kw, dil, pad, stride = 2, 7, 41, 64, 64, 1, 3, (1, 1), (0, 1), (1, 2)
    else:
        B, T, Fq, Cin, N, kh, kw, dil, pad, stride = 2, 20, 17, 16, 32, 4, 4, (1, 1), (1, 1), (2, 2)
    x = rnd(B, T, Fq, Cin, seed=1).double().requires_grad_(True)
    w = rnd(N, Cin, kh, kw, seed=2, scale=(Cin * kh * kw) ** -0.5).double().requires_grad_(True)
    b = rnd(N, seed=3).double().requires_grad_(True)
    xin = x.permute(0, 3, 1, 2)
    if case == 'dilated':
        out = F.conv2d(F.pad(xin, (1, 1, 2, 0)), w, b, dilation=dil)
    else:
        out = F.conv2d(xin, w, b, stride=stride, padding=pad, dilation=dil)
    out = out.permute(0, 2, 3, 1)
    To, Fo = out.shape[1], out.shape[2]
    dy = rnd(B, To, Fo, N, seed=9)
    out.backward(dy.double())
    taps = gemm.conv_taps(kh, kw, dil, pad)
    x32, w32 = x.detach().float().contiguous(), w.detach().float().contiguous()
    # input gradient: same kernel, transposed packing, negated taps, up-mode for strides
    wd = gemm.pack_conv_dgrad(w32)
    dd = gemm.make_desc(B, T, Fq, To, Fo, [(-a, -c) for a, c in taps], N, N, Cin, Cin,
                        st=stride[0], sf=stride[1], up=1 if stride != (1, 1) else 0)
    dx = torch.empty(B, T, Fq, Cin, device='cuda')
    gemm.gemm_tap(dd, dy, wd, dx)
    assert relerr(dx, x.grad) < 5e-6
    # weight / bias gradient
    dwp = torch.zeros(N, kh * kw * Cin, device='cuda')
    db = torch.zeros(N, device='cuda')
    fd = gemm.make_desc(B, To, Fo, T, Fq, taps, Cin, Cin, N, N, st=stride[0], sf=stride[1])
    gemm.gemm_tap_wgrad(fd, x32, dy, dwp, db)
    dw = torch.zeros(N, Cin, kh, kw, device='cuda')
    gemm.unpack_conv_wgrad(dwp, dw)
    assert relerr(dw, w.grad) < 2e-5
    assert relerr(db, b.grad) < 2e-5


def test_wgrad_prologues(G):
    gemm, L = G
    M, Cin, N = 3001, 64, 256
    x, dy = rnd(M, Cin, seed=1) + 0.3, rnd(M, N, seed=2)
    g, b = rnd(Cin, seed=4) * 0.1 + 1, rnd(Cin, seed=5) * 0.1
    stats = torch.stack([x.mean(-1), (x.var(-1, unbiased=False) + 1e-5).rsqrt()], -1).contiguous()
    for pro, fn in ((L.PRO_LN, lambda v: F.layer_norm(v, (Cin,), g.double(), b.double(), 1e-5)),
                    (L.PRO_SWISH, F.silu),
                    (L.PRO_AFFINE_SWISH, lambda v: F.silu(v * g.double() + b.double())),
                    (L.PRO_NONE, lambda v: v)):
        dw = torch.zeros(N, Cin, device='cuda')
        db = torch.zeros(N, device='cuda')
        gemm.gemm_tap_wgrad(gemm.linear_desc(M, Cin, N, prologue=pro), x, dy, dw, db, rowstats=stats, ps=g, pb=b)
        assert relerr(dw, dy.double().T @ fn(x.double())) < 2e-5, pro
        assert relerr(db, dy.double().sum(0)) < 2e-5


def test_missing_operand_is_loud(G):
    gemm, L = G
    x, w = rnd(64, 64), rnd(64, 64)
    y = torch.empty(64, 64, device='cuda')
    with pytest.raises(L.SeHipError):
        gemm.gemm_tap(gemm.linear_desc(64, 64, 64, epilogue=L.EPI_BIAS), x, w, y)
    with pytest.raises(L.SeHipError):
        gemm.gemm_tap(gemm.linear_desc(64, 64, 64), x.cpu(), w, y)


def _mask(seed, rows, cols, p):
    """torch restatement of the kernels' counter-based dropout mask: aligned groups of 4 elements, murmur3 finalizer of
    (seed, group) + one multiply-xorshift step = four 16-bit fields, keep iff field >= round(p * 65536)."""
    M32 = 0xFFFFFFFF
    grp = torch.arange(rows * cols // 4, device='cuda', dtype=torch.int64)
    x = ((grp * 0x9E3779B1) & M32) ^ seed
    x = x ^ (x >> 16)
    x = (x * 0x85EBCA6B) & M32
    x = x ^ (x >> 13)
    x = (x * 0xC2B2AE35) & M32
    x = x ^ (x >> 16)
    y = (x * 0x9E3779B1 + 0x7F4A7C15) & M32
    y = y ^ (y >> 15)
    f = torch.stack([x & 0xFFFF, x >> 16, y & 0xFFFF, y >> 16], -1).reshape(-1)
    thr = int(p * 65536.0 + 0.5)
    return ((f >= thr).double() * (65536.0 / (65536.0 - thr))).view(rows, cols)


def test_feed_forward_with_dropout_fwd_bwd(G):
    """x + 0.5 * Drop(W2 Drop(Swish(W1 LN x))) with counter-based masks: forward and every gradient against a
    fp64 torch restatement using the SAME masks (so the backward provably re-creates the forward's masks)."""
    from speech_enhancement_amd import layers as LY
    M, p = 1500, 0.2
    g = torch.Generator().manual_seed(3)
    P = {'ff.fn.norm.weight': 1 + 0.1 * torch.randn(64, generator=g), 'ff.fn.norm.bias': 0.1 * torch.randn(64, generator=g),
         'ff.fn.fn.net.0.weight': torch.randn(256, 64, generator=g) / 8, 'ff.fn.fn.net.0.bias': 0.1 * torch.randn(256, generator=g),
         'ff.fn.fn.net.3.weight': torch.randn(64, 256, generator=g) / 16, 'ff.fn.fn.net.3.bias': 0.1 * torch.randn(64, generator=g)}
    P = {k: v.cuda() for k, v in P.items()}
    x, dy = rnd(M, 64, seed=5) + 0.2, rnd(M, 64, seed=6)
    sh, so = 0x1234ABCD, 0x0BADF00D
    y, saved = LY._ff_fwd(P, 'ff', x, M, p, sh, so)
    Gd = {k: torch.zeros_like(v) for k, v in P.items()}
    dx = LY._ff_bwd(P, Gd, 'ff', saved, dy, M)
    mh, mo = _mask(sh, M, 256, p), _mask(so, M, 64, p)
    assert abs(float(mh.mean()) - 1.0) < 0.02 and abs(float((mo > 0).double().mean()) - 0.8) < 0.02
    P64 = {k: v.double().requires_grad_(True) for k, v in P.items()}
    x64 = x.double().requires_grad_(True)
    h = F.layer_norm(x64, (64,), P64['ff.fn.norm.weight'], P64['ff.fn.norm.bias'], 1e-5)
    h = F.silu(h @ P64['ff.fn.fn.net.0.weight'].T + P64['ff.fn.fn.net.0.bias']) * mh
    ref = x64 + 0.5 * mo * (h @ P64['ff.fn.fn.net.3.weight'].T + P64['ff.fn.fn.net.3.bias'])
    assert relerr(y, ref) < 5e-6
    ref.backward(dy.double())
    assert relerr(dx, x64.grad) < 3e-5
    for k in P:
        assert relerr(Gd[k], P64[k].grad) < 5e-5, k


def test_split_bf16_precision_mode(G):
    """precision=1 (hi/lo bf16 split, 3 bf16 MFMAs, fp32 accumulate) on a dilated conv from a skip stack and its input
    gradient: error vs fp64 must stay ~1e-5 of the output scale (plain bf16 would be ~4e-3)."""
    gemm, L = G
    B, T, Fq = 2, 33, 45
    xbuf = rnd(B, T, Fq, 256, seed=3)
    w = rnd(64, 256, 2, 3, seed=4, scale=0.03)
    b = rnd(64, seed=5)
    x = xbuf.permute(0, 3, 1, 2).double()
    ref = F.conv2d(F.pad(x, (1, 1, 8, 0)), w.double(), b.double(), dilation=(8, 1)).permute(0, 2, 3, 1)
    taps = gemm.conv_taps(2, 3, (8, 1), (8, 1))
    wp = gemm.pack_conv_fwd(w)
    errs = {}
    for prec in (0, 1, 2):
        d = gemm.make_desc(B, T, Fq, T, Fq, taps, 256, 256, 64, 64, epilogue=L.EPI_BIAS | L.EPI_STATS, precision=prec)
        y = torch.empty(B, T, Fq, 64, device='cuda')
        st = torch.zeros(B, 64, 2, device='cuda', dtype=torch.float64)
        gemm.gemm_tap(d, xbuf, wp, y, bias=b, stats=st)
        errs[prec] = relerr(y, ref)
        assert relerr(st[..., 1], (ref ** 2).sum((1, 2))) < 1e-4
    print('conv relerr fp32 %.2e  bf16x3 %.2e  bf16x6 %.2e' % (errs[0], errs[1], errs[2]))
    assert errs[0] < 3e-6 and errs[1] < 1e-4 and errs[2] < 3e-6
    # input gradient (dgrad) through the same path
    dy = rnd(B, T, Fq, 64, seed=9)
    x64 = xbuf.double().requires_grad_(True)
    F.conv2d(F.pad(x64.permute(0, 3, 1, 2), (1, 1, 8, 0)), w.double(), None, dilation=(8, 1)).permute(0, 2, 3, 1).backward(dy.double())
    wd = gemm.pack_conv_dgrad(w)
    dd = gemm.make_desc(B, T, Fq, T, Fq, [(-a, -c) for a, c in taps], 64, 64, 256, 256, precision=1)
    dx = torch.empty(B, T, Fq, 256, device='cuda')
    gemm.gemm_tap(dd, dy, wd, dx)
    assert relerr(dx, x64.grad) < 1e-4
    # weight / bias gradient through the transposed-staging split-bf16 kernel (all three modes), LN-prologue linear too
    w64 = w.double().requires_grad_(True)
    b64 = b.double().requires_grad_(True)
    F.conv2d(F.pad(x, (1, 1, 8, 0)), w64, b64, dilation=(8, 1)).permute(0, 2, 3, 1).backward(dy.double())
    # (precision 2 on generic shapes runs the fp32 kernel: the six-product generic weight-gradient kernel was slower and is gone)
    for prec, tol in ((0, 3e-6), (1, 1e-4), (2, 3e-6)):
        fd = gemm.make_desc(B, T, Fq, T, Fq, taps, 256, 256, 64, 64, precision=prec)
        dwp = torch.zeros(64, len(taps) * 256, device='cuda')
        db = torch.zeros(64, device='cuda')
        gemm.gemm_tap_wgrad(fd, xbuf, dy, dwp, db, chunks=5)
        dw = torch.zeros_like(w)
        gemm.unpack_conv_wgrad(dwp, dw)
        assert relerr(dw, w64.grad) < tol, (prec, relerr(dw, w64.grad))
        assert relerr(db, b64.grad) < 1e-5, prec
    M = 1000
    xl = rnd(M, 64, seed=11)
    dyl = rnd(M, 192, seed=12)
    gam, bet = rnd(64, seed=13), rnd(64, seed=14)
    xn = F.layer_norm(xl.double(), (64,), gam.double(), bet.double())
    refw = dyl.double().t() @ xn
    st = torch.stack([xl.mean(-1), (xl.var(-1, unbiased=False) + 1e-5).rsqrt()], -1).contiguous()
    for prec, tol in ((0, 3e-6), (1, 1e-4), (2, 3e-6)):
        dl = gemm.linear_desc(M, 64, 192, prologue=L.PRO_LN, precision=prec)
        dwl = torch.zeros(192, 64, device='cuda')
        gemm.gemm_tap_wgrad(dl, xl, dyl, dwl, None, rowstats=st, ps=gam, pb=bet, chunks=3, explicit_precision=True)
        assert relerr(dwl, refw) < tol, (prec, relerr(dwl, refw))


@pytest.mark.parametrize('B,T,Fq,dil,C', [(1, 1, 2, 1, 64), (2, 3, 3, 2, 32), (1, 5, 7, 1, 96), (2, 4, 130, 2, 64), (3, 9, 65, 4, 128), (2, 3, 67, 1, 64),
                                            (1, 6, 201, 2, 192)])
def test_triple_tap_kernels_edge_shapes(G, B, T, Fq, dil, C):
    """conv3_bf16_kernel / wgrad3_kernel (one halo tile shared by the df = -1, 0, +1 taps): frequency-edge masks,
    time padding, tiles that straddle time rows and batch entries, tiny grids -- against fp64 conv2d autograd."""
    gemm, L = G
    xbuf = rnd(B, T, Fq, C, seed=21)
    w = rnd(64, C, 2, 3, seed=22, scale=0.05)
    dy = rnd(B, T, Fq, 64, seed=23)
    x64 = xbuf.double().requires_grad_(True)
    w64 = w.double().requires_grad_(True)
    y64 = F.conv2d(F.pad(x64.permute(0, 3, 1, 2), (1, 1, dil, 0)), w64, None, dilation=(dil, 1)).permute(0, 2, 3, 1)
    y64.backward(dy.double())
    taps = gemm.conv_taps(2, 3, (dil, 1), (dil, 1))
    wp = gemm.pack_conv_fwd(w)
    for prec, tol in ((2, 3e-6), (1, 1e-4)):
        d = gemm.make_desc(B, T, Fq, T, Fq, taps, C, C, 64, 64, precision=prec)
        y = torch.empty(B, T, Fq, 64, device='cuda')
        gemm.gemm_tap(d, xbuf, wp, y)
        assert relerr(y, y64) < tol, (prec, relerr(y, y64))
        dd = gemm.make_desc(B, T, Fq, T, Fq, [(-a, -c) for a, c in taps], 64, 64, C, C, precision=prec)
        dx = torch.empty(B, T, Fq, C, device='cuda')
        gemm.gemm_tap(dd, dy, gemm.pack_conv_dgrad(w), dx)
        assert relerr(dx, x64.grad) < tol, (prec, relerr(dx, x64.grad))
    for prec, tol in ((0, 3e-6), (2, 3e-6), (1, 1e-4)):      # 1 / 2: wgrad3_bf16_kernel when Fq > 66, else the generic split kernel
        for chunks in (1, 3):
            fd = gemm.make_desc(B, T, Fq, T, Fq, taps, C, C, 64, 64, precision=prec)
            dwp = torch.zeros(64, len(taps) * C, device='cuda')
            db = torch.zeros(64, device='cuda')
            gemm.gemm_tap_wgrad(fd, xbuf, dy, dwp, db, chunks=chunks)
            dw = torch.zeros_like(w)
            gemm.unpack_conv_wgrad(dwp, dw)
            assert relerr(dw, w64.grad) < tol, (prec, chunks, relerr(dw, w64.grad))
            assert relerr(db, dy.double().sum((0, 1, 2))) < 1e-5


def test_weight_plan_matches_per_use_packing_and_planes_are_bit_identical(G):
    """se_weight_prep (weights.WeightPlan): one launch reproduces the per-use repack / transpose / cat / scale helpers, and the
    pre-split bf16 planes give BIT-IDENTICAL GEMM results to the fp32 weights (the 3-way split is exact and the kernels
    evaluate the same six products): conv3 (dilated, skip stack), generic split kernel (strided), row-panel, fused
    feed-forward forward and backward."""
    gemm, L = G
    from speech_enhancement_amd import layers as LY
    from speech_enhancement_amd.weights import WeightPlan
    dev = torch.device('cuda')
    w = rnd(64, 192, 2, 3, seed=1, scale=0.05)              # DilatedDenseNet conv3 (newest-first slabs)
    w2 = rnd(64, 64, 1, 3, seed=2, scale=0.1)               # strided (1, 3)
    w1x2 = rnd(2, 64, 1, 2, seed=3, scale=0.1)              # 64 -> 2 channels, rows padded to 4
    W1, W2 = rnd(256, 64, seed=4, scale=0.1), rnd(64, 256, seed=5, scale=0.05)
    Wq, Wkv = rnd(64, 64, seed=6, scale=0.1), rnd(128, 64, seed=7, scale=0.1)
    plan = WeightPlan(dev)
    for planes in (False, True):
        t = 'p' if planes else 'f'
        plan.conv_fwd(('w', 'fwd', t), w, rev=True, planes=planes)
        plan.conv_dgrad(('w', 'dgrad', t), w, rev=True, planes=planes)
        plan.conv_fwd(('w2', 'fwd', t), w2, planes=planes)
        plan.conv_fwd(('w1x2', 'fwd', t), w1x2, N_pad=4, planes=planes)
        plan.linear(('W1', t), W1, planes=planes)
        plan.linear(('W2', t), W2, planes=planes)
        plan.linear_T(('W2', 'T0.5', t), W2, planes=planes, scale=0.5)
        plan.linear_T(('W1', 'T', t), W1, planes=planes)
        plan.linear(('qkv', t), Wq, planes=planes, rows=192)
        plan.linear(('qkv', t), Wkv, planes=planes, o_off=64)
        plan.linear_T(('qkvT', t), Wq, planes=planes, ld=192)
        plan.linear_T(('qkvT', t), Wkv, planes=planes, c_off=64)
    plan.conv_dgrad(('w1x2', 'dgrad'), w1x2, N_pad=4)
    plan.run()
    o = plan.out
    # fp32 items == the per-use helpers
    assert torch.equal(o[('w', 'fwd', 'f')], LY.pack_w(w, rev=True))
    assert torch.equal(o[('w', 'dgrad', 'f')], gemm.pack_conv_dgrad(w, rev_slabs=True))
    assert torch.equal(o[('w1x2', 'fwd', 'f')], LY.pack_w(LY.pad_rows(w1x2, 4)))
    assert torch.equal(o[('w1x2', 'dgrad')], gemm.pack_conv_dgrad(LY.pad_rows(w1x2, 4)))
    assert torch.equal(o[('W2', 'T0.5', 'f')], W2.t().contiguous() * 0.5)
    assert torch.equal(o[('qkv', 'f')], torch.cat([Wq, Wkv], 0))
    assert torch.equal(o[('qkvT', 'f')], torch.cat([Wq, Wkv], 0).t().contiguous())
    # planes: hi + mid + lo == the fp32 value exactly
    for key in (('w', 'fwd'), ('W2', 'T0.5'), ('qkvT',)):
        pl = o[key + ('p',)].float()
        assert torch.equal(pl[0] + pl[1] + pl[2], o[key + ('f',)])
    # GEMMs: planes vs fp32 weights, bit-identical
    B, T, Fq = 2, 5, 67
    skip = rnd(B, T, Fq, 256, seed=10)
    def run(W, d, A, N, **kw):
        y = torch.empty(A.shape[:-1] + (N,), device='cuda')
        gemm.gemm_tap(d, A, W, y, **kw)
        return y
    d = lambda: gemm.make_desc(B, T, Fq, T, Fq, LY.dense_taps(2), 192, 256, 64, 64, precision=2)
    assert torch.equal(run(o[('w', 'fwd', 'p')], d(), skip, 64), run(o[('w', 'fwd', 'f')], d(), skip, 64))
    Fo = (Fq + 2 - 3) // 2 + 1
    a2 = rnd(B, T, Fq, 64, seed=11)
    ds = lambda: gemm.make_desc(B, T, Fo, T, Fq, LY.TAPS_1x3, 64, 64, 64, 64, sf=2, precision=2)
    ys = [torch.empty(B, T, Fo, 64, device='cuda') for _ in range(2)]
    gemm.gemm_tap(ds(), a2, o[('w2', 'fwd', 'p')], ys[0])
    gemm.gemm_tap(ds(), a2, o[('w2', 'fwd', 'f')], ys[1])
    assert torch.equal(ys[0], ys[1])
    M = 1000
    x = rnd(M, 64, seed=12)
    from speech_enhancement_amd import ops as O
    st = O.row_stats(x, M)
    gam, bet = rnd(64, seed=13) * 0.1 + 1, rnd(64, seed=14) * 0.1
    dq = lambda: gemm.linear_desc(M, 64, 192, prologue=L.PRO_LN, precision=2)
    assert torch.equal(run(o[('qkv', 'p')], dq(), x, 192, rowstats=st, ps=gam, pb=bet),
                       run(o[('qkv', 'f')], dq(), x, 192, rowstats=st, ps=gam, pb=bet))
    dqk = rnd(M, 192, seed=15)
    dt = lambda: gemm.linear_desc(M, 192, 64, precision=2)
    assert torch.equal(run(o[('qkvT', 'p')], dt(), dqk, 64), run(o[('qkvT', 'f')], dt(), dqk, 64))
    # pre-split weights are refused where no six-product kernel would read them
    with pytest.raises(L.SeHipError):
        gemm.gemm_tap(gemm.linear_desc(M, 64, 192, prologue=L.PRO_LN, precision=0), x, o[('qkv', 'p')],
                      torch.empty(M, 192, device='cuda'), rowstats=st, ps=gam, pb=bet)


def test_wgrad_scale_accumulates_in_place(G):
    """se_gemm_tap_wgrad adds alpha * gradient into the caller's buffer (Scale(0.5) feed-forward weight gradient)."""
    gemm, L = G
    M = 900
    a, dy = rnd(M, 256, seed=1), rnd(M, 64, seed=2)
    dw0, db0 = rnd(64, 256, seed=3), rnd(64, seed=4)
    dw, db = dw0.clone(), db0.clone()
    gemm.gemm_tap_wgrad(gemm.linear_desc(M, 256, 64), a, dy, dw, db, scale=0.5)
    assert relerr(dw, dw0.double() + 0.5 * dy.double().T @ a.double()) < 2e-6
    assert relerr(db, db0.double() + 0.5 * dy.double().sum(0)) < 2e-6


@pytest.mark.parametrize('Cin,N,pro,prec', [(64, 256, 'ln', 0), (256, 64, 'swish_drop', 0), (64, 192, 'ln', 0), (128, 64, 'affine_swish', 0),
                                            (64, 256, 'ln', 2), (256, 64, 'swish_drop', 2), (64, 224, 'ln', 2), (200, 64, 'swish_drop', 2),
                                            (256, 64, 'affine_swish', 2), (64, 256, 'none', 2)])
def test_full_tile_linear_wgrad_equals_block_kernel(G, Cin, N, pro, prec, monkeypatch):
    """wgrad_lin_kernel / wgrad_lin_bf16_kernel (the whole [N x C] gradient in one workgroup; prec 0: fp32 MFMA, prec 2: six
    split-bf16 products) against the per-block kernel and fp64 torch, incl. a ragged last chunk, padded columns, the dropout
    mask on dY and the bias gradient (shapes outside the whole-gradient kernels run the block kernel in both modes)."""
    gemm, L = G
    M = 5000 + 37
    x, dy = rnd(M, Cin, seed=1), rnd(M, N, seed=2)
    from speech_enhancement_amd import ops as O
    st = O.row_stats(x, M) if Cin == 64 else None
    g, b = rnd(Cin, seed=3) * 0.1 + 1, rnd(Cin, seed=4) * 0.1
    code = {'ln': L.PRO_LN, 'swish_drop': L.PRO_SWISH_DROP, 'affine_swish': L.PRO_AFFINE_SWISH, 'none': L.PRO_NONE}[pro]
    dp = 0.2 if pro == 'swish_drop' else 0.0
    mk = lambda: gemm.linear_desc(M, Cin, N, prologue=code, epilogue=L.EPI_DROP if dp else 0, pro_seed=3, epi_seed=9, drop_p=dp,
                                  precision=prec)
    out = {}
    for mode in ('blocks', 'full'):
        if mode == 'blocks':
            monkeypatch.setenv('SE_WGRAD_NO_LIN', '1')
        else:
            monkeypatch.delenv('SE_WGRAD_NO_LIN')
        dw, db = torch.zeros(N, Cin, device='cuda'), torch.zeros(N, device='cuda')
        gemm.gemm_tap_wgrad(mk(), x, dy, dw, db, rowstats=st, ps=g, pb=b, explicit_precision=True)
        out[mode] = (dw, db)
    assert relerr(out['full'][0], out['blocks'][0]) < 5e-6 and relerr(out['full'][1], out['blocks'][1]) < 5e-6
    if pro in ('ln', 'none'):
        a = torch.nn.functional.layer_norm(x.double(), (Cin,), g.double(), b.double(), 1e-5) if pro == 'ln' else x.double()
        assert relerr(out['full'][0], dy.double().T @ a) < 5e-6 and relerr(out['full'][1], dy.double().sum(0)) < 5e-6


@pytest.mark.parametrize('K,planes', [(192, True), (256, True), (192, False)])
def test_gemm_ln_bwd_equals_gemm_then_layernorm_bwd(G, K, planes):
    """se_gemm_ln_bwd (input-gradient GEMM + LayerNorm backward on the accumulators) against se_gemm_tap + se_layernorm_bwd and
    against fp64 autograd of  y = LayerNorm(x) @ W.T  (dX incl. the residual path, dgamma, dbeta); ragged M."""
    gemm, L = G
    from speech_enhancement_amd import ops as O
    from speech_enhancement_amd.weights import WeightPlan
    M = 128 * 7 + 45
    x, dy, dR = rnd(M, 64, seed=1) * 1.5 + 0.3, rnd(M, K, seed=2), rnd(M, 64, seed=3)
    W = rnd(K, 64, seed=4, scale=0.1)                       # the projection's weight [K out, 64 in]
    gam, bet = rnd(64, seed=5) * 0.2 + 1.0, rnd(64, seed=6) * 0.1
    st = O.row_stats(x, M)
    plan = WeightPlan(torch.device('cuda'))
    WT = plan.linear_T('wt', W, planes=planes)              # [64][K]
    plan.run()
    dg, db = torch.zeros(64, device='cuda'), torch.zeros(64, device='cuda')
    dX = gemm.gemm_ln_bwd(dy, WT, x, st, gam, dR, dg, db)
    # two-kernel form
    dl = torch.empty(M, 64, device='cuda')
    gemm.gemm_tap(gemm.linear_desc(M, K, 64, precision=2), dy, WT, dl)
    dg2, db2 = torch.zeros(64, device='cuda'), torch.zeros(64, device='cuda')
    dX2 = O.layernorm_bwd(x, st, gam, dl, dg2, db2, dR=dR)
    assert relerr(dX, dX2) < 2e-6 and relerr(dg, dg2) < 1e-5 and relerr(db, db2) < 1e-5
    # fp64 autograd
    x64 = x.double().requires_grad_(True)
    g64, b64 = gam.double().requires_grad_(True), bet.double().requires_grad_(True)
    y = torch.nn.functional.layer_norm(x64, (64,), g64, b64, 1e-5) @ W.double().T
    y.backward(dy.double())
    assert relerr(dX, x64.grad + dR.double()) < 5e-6 and relerr(dg, g64.grad) < 1e-5 and relerr(db, b64.grad) < 1e-5


def test_row_statistics_from_producer_epilogues(G):
    """SE_EPI_ROWSTATS (row GEMM, N == 64) and se_ff_fwd_stats: (mean, rstd) of the RESULT rows equal se_row_stats on the
    stored result (same two-pass arithmetic on the same fp32 values), incl. a ragged last tile, dropout and a prologue."""
    gemm, L = G
    from speech_enhancement_amd import ops as O
    M = 128 * 5 + 77
    for K, pro in ((64, L.PRO_NONE), (128, L.PRO_AFFINE_SWISH)):
        x, w, b, r = rnd(M, K, seed=1), rnd(64, K, seed=2, scale=K ** -0.5), rnd(64, seed=3), rnd(M, 64, seed=4)
        sc, sh = rnd(K, seed=5) * 0.1 + 1, rnd(K, seed=6) * 0.1
        y, st = torch.empty(M, 64, device='cuda'), torch.full((M, 2), float('nan'), device='cuda')
        d = gemm.linear_desc(M, K, 64, prologue=pro, epilogue=L.EPI_BIAS | L.EPI_RESID | L.EPI_DROP | L.EPI_ROWSTATS, alpha=1.0,
                             ldr=64, epi_seed=17, drop_p=0.2)
        gemm.gemm_tap(d, x, w, y, bias=b, R=r, AUX=st, ps=sc, pb=sh)
        ref = O.row_stats(y, M)
        assert torch.isfinite(st).all() and relerr(st, ref) < 1e-6
    x = rnd(M, 64, seed=7)
    with pytest.raises(L.SeHipError):          # only whole 64-channel rows have row statistics
        gemm.gemm_tap(gemm.linear_desc(M, 64, 128, epilogue=L.EPI_ROWSTATS), x, rnd(128, 64, seed=14), torch.empty(M, 128, device='cuda'),
                      AUX=torch.empty(M, 2, device='cuda'))


def test_ff_stored_h_kernels_and_fused_layernorm_backward(G):
    """the cross-check path of the fused feed-forward kernels (scaled fp16 planes, H stored) at a ragged M: ff_fwd with / without the H
    store agree (the W-stationary kernel contracts the hidden units of a 16-block in a permuted order: rounding-level differences),
    the output row statistics equal se_row_stats, and the LayerNorm backward fused into ff_bwd_dgrad (ln=...) equals ff_bwd_dgrad +
    se_layernorm_bwd, incl. the residual path and dgamma / dbeta; forward and dZ against fp64."""
    gemm, L = G
    from speech_enhancement_amd import ops as O
    from speech_enhancement_amd.weights import WeightPlan
    M, hid = 128 * 5 + 77, 256
    x, dy, dR2 = rnd(M, 64, seed=1) * 1.3 + 0.2, rnd(M, 64, seed=2) * 1e-3, rnd(M, 64, seed=3) * 1e-3
    W1, b1 = rnd(hid, 64, seed=4, scale=0.1), rnd(hid, seed=5) * 0.1
    W2, b2 = rnd(64, hid, seed=6, scale=0.05), rnd(64, seed=7) * 0.1
    gam, bet = rnd(64, seed=8) * 0.2 + 1.0, rnd(64, seed=9) * 0.1
    st = O.row_stats(x, M)
    plan = WeightPlan(torch.device('cuda'))
    w1, w2 = plan.linear('w1', W1, planes='f16'), plan.linear('w2', W2, planes='f16')
    w2t, w1t = plan.linear_T('w2t', W2, planes='f16', scale=0.5), plan.linear_T('w1t', W1, planes='f16')
    plan.run()
    y, h, ost = gemm.ff_fwd(x, st, gam, bet, w1, b1, w2, b2, 0.0, 11, 12, 0.5, out_stats=True)
    y2, h2 = gemm.ff_fwd(x, st, gam, bet, w1, b1, w2, b2, 0.0, 11, 12, 0.5, store_h=False)
    assert h2 is None and relerr(y2, y) < 1e-6
    assert relerr(ost, O.row_stats(y, M)) < 1e-5
    xl = F.layer_norm(x.double(), (64,), gam.double(), bet.double(), 1e-5)
    h64 = xl @ W1.double().T + b1.double()
    assert relerr(h, h64) < 2e-6 and relerr(y, x.double() + 0.5 * (F.silu(h64) @ W2.double().T + b2.double())) < 2e-6
    with pytest.raises(Exception):          # plain fp32 weights take the unfused GEMM path (layers._ff_fwd), not this kernel
        gemm.ff_fwd(x, st, gam, bet, W1, b1, W2, b2)
    dy._se_amax = dy.abs().max().reshape(1).clone()
    z = lambda: torch.zeros(1, device='cuda')
    y, h = gemm.ff_fwd(x, st, gam, bet, w1, b1, w2, b2, 0.2, 11, 12, 0.5)
    z0, dln = gemm.ff_bwd_dgrad(dy, h, w2t, w1t, 0.2, 11, 12, amax_out=(z(), z()))
    dg0, db0 = torch.zeros(64, device='cuda'), torch.zeros(64, device='cuda')
    dx0 = O.layernorm_bwd(x, st, gam, dln, dg0, db0, dR=dy, dR2=dR2)
    dg1, db1 = torch.zeros(64, device='cuda'), torch.zeros(64, device='cuda')
    z1, dx1 = gemm.ff_bwd_dgrad(dy, h, w2t, w1t, 0.2, 11, 12, ln=(x, st, gam, dR2, dg1, db1), amax_out=(z(), z()))
    assert torch.equal(z1, z0)
    assert relerr(dx1, dx0) < 2e-6 and relerr(dg1, dg0) < 2e-5 and relerr(db1, db0) < 2e-5
    assert abs(float(dx1._se_amax) - float(dx1.abs().max())) <= 1e-6 * float(dx1.abs().max())


@pytest.mark.parametrize('B,T,Fq', [(2, 321, 201), (3, 16, 18), (1, 2, 2), (2, 37, 131), (16, 64, 300)])
def test_discriminator_first_stage_direct_kernels(G, B, T, Fq):
    """csrc/se_thin.hip vs torch: Conv2d(2, 16, 4, 2, 1, bias=False) (models/discriminator.py:39) on the transposed image --
    forward (+ the fp64 InstanceNorm sums), input gradient (channels 2, 3 of the planes zero), weight gradient (accumulating)."""
    import ctypes as C
    gemm, L = G
    x = rnd(B, T, Fq, 4, seed=1)
    x[..., 2:] = 0.25                                      # never read
    w = rnd(16, 2, 4, 4, seed=2, scale=0.2)
    To, Fo = (T - 2) // 2 + 1, (Fq - 2) // 2 + 1
    R = torch.empty(B, To, Fo, 16, device='cuda')
    stats = torch.zeros(B, 16, 2, device='cuda', dtype=torch.float64)
    L.call('se_dconv1_fwd', L.ptr(x), L.ptr(w), L.ptr(R), L.ptr(stats), C.c_int(B), C.c_int(T), C.c_int(Fq), C.c_int(16), L.stream())
    # the reference image is [B, 2, F, T] with kernel index (kh over F, kw over T)
    xi = x[..., :2].permute(0, 3, 2, 1).double().requires_grad_(True)
    wd = w.double().requires_grad_(True)
    ref = F.conv2d(xi, wd, None, 2, 1)                     # [B, 16, Fo, To]
    assert ref.shape == (B, 16, Fo, To)
    assert relerr(R, ref.detach().permute(0, 3, 2, 1)) < 2e-6
    rr = R.double()
    # fp32 partial sums over a workgroup's <= 4 x 64 pixels, fp64 from there on
    assert relerr(stats[..., 0], rr.sum((1, 2))) < 1e-6 and relerr(stats[..., 1], (rr * rr).sum((1, 2))) < 2e-7
    dR = rnd(B, To, Fo, 16, seed=3)
    ref.backward(dR.permute(0, 3, 2, 1).double())
    dx = torch.full((B, T, Fq, 4), 7.0, device='cuda')
    L.call('se_dconv1_dgrad', L.ptr(dR), L.ptr(w), L.ptr(dx), C.c_int(B), C.c_int(T), C.c_int(Fq), C.c_int(16), L.stream())
    assert relerr(dx[..., :2], xi.grad.permute(0, 3, 2, 1)) < 2e-6
    assert float(dx[..., 2:].abs().max()) == 0.0
    dw = torch.zeros(16, 2, 4, 4, device='cuda')
    L.call('se_dconv1_wgrad', L.ptr(x), L.ptr(dR), L.ptr(dw), C.c_int(B), C.c_int(T), C.c_int(Fq), C.c_int(16), L.stream())
    assert relerr(dw, wd.grad) < 3e-6
    L.call('se_dconv1_wgrad', L.ptr(x), L.ptr(dR), L.ptr(dw), C.c_int(B), C.c_int(T), C.c_int(Fq), C.c_int(16), L.stream())
    assert relerr(dw, 2 * wd.grad) < 3e-6                  # accumulates
    with pytest.raises(L.SeHipError):
        L.call('se_dconv1_fwd', L.ptr(x), L.ptr(w), L.ptr(R), None, C.c_int(B), C.c_int(T), C.c_int(Fq), C.c_int(32), L.stream())


@pytest.mark.parametrize('B,T,F2,n', [(2, 7, 202, 1), (3, 5, 33, 2), (1, 1, 2, 2), (16, 9, 202, 2)])
def test_generator_thin_convolutions_1x2(G, B, T, F2, n):
    """csrc/se_thin.hip vs torch fp64: Conv2d(64, n, (1, 2)) (models/generator.py:114, :128) forward (+ InstanceNorm sums), input
    gradient, weight / bias gradient (accumulating into PyTorch-layout tensors)."""
    import ctypes as C
    gemm, L = G
    Fo = F2 - 1
    x = rnd(B, T, F2, 64, seed=1)
    w, bias = rnd(n, 64, 1, 2, seed=2, scale=0.1), rnd(n, seed=3)
    y = torch.full((B, T, Fo, 4), 9.0, device='cuda')
    st = torch.zeros(B, 4, 2, device='cuda', dtype=torch.float64)
    L.call('se_conv1x2_fwd', L.ptr(x), L.ptr(w), L.ptr(bias), L.ptr(y), L.ptr(st), C.c_int(B), C.c_int(T), C.c_int(F2), C.c_int(n), L.stream())
    xi = x.permute(0, 3, 1, 2).double().requires_grad_(True)          # [B, 64, T, F2]
    wd, bd = w.double().requires_grad_(True), bias.double().requires_grad_(True)
    ref = F.conv2d(xi, wd, bd)                                         # [B, n, T, Fo]
    assert relerr(y[..., :n], ref.detach().permute(0, 2, 3, 1)) < 2e-6
    assert float(y[..., n:].abs().max()) == 0.0
    yy = y.double()
    assert relerr(st[..., 0], yy.sum((1, 2))) < 1e-6 and relerr(st[..., 1], (yy * yy).sum((1, 2))) < 1e-6
    dy = rnd(B, T, Fo, 4, seed=4)
    ref.backward(dy[..., :n].permute(0, 3, 1, 2).double())
    dx = torch.full((B, T, F2, 64), 5.0, device='cuda')
    L.call('se_conv1x2_dgrad', L.ptr(dy), L.ptr(w), L.ptr(dx), C.c_long(B * T), C.c_int(F2), C.c_int(n), L.stream())
    assert relerr(dx, xi.grad.permute(0, 2, 3, 1)) < 2e-6
    dw, db = torch.full((n, 64, 1, 2), 0.25, device='cuda'), torch.full((n,), 0.5, device='cuda')
    L.call('se_conv1x2_wgrad', L.ptr(x), L.ptr(dy), L.ptr(dw), L.ptr(db), C.c_long(B * T), C.c_int(F2), C.c_int(n), L.stream())
    assert relerr(dw - 0.25, wd.grad) < 5e-6 and relerr(db - 0.5, bd.grad) < 5e-6
    with pytest.raises(L.SeHipError):
        L.call('se_conv1x2_fwd', L.ptr(x), L.ptr(w), L.ptr(bias), L.ptr(y), None, C.c_int(B), C.c_int(T), C.c_int(F2), C.c_int(3), L.stream())


@pytest.mark.parametrize('B,P', [(2, 1000), (3, 77), (16, 6421)])
def test_generator_thin_convolution_3_to_64(G, B, P):
    """csrc/se_thin.hip vs torch fp64: the encoder's Conv2d(3, 64, (1, 1)) (models/generator.py:39) on the planes [B, P, 4]"""
    import ctypes as C
    gemm, L = G
    x = rnd(B, P, 4, seed=1)
    w, bias = rnd(64, 3, 1, 1, seed=2, scale=0.5), rnd(64, seed=3)
    R = torch.empty(B, P, 64, device='cuda')
    st = torch.zeros(B, 64, 2, device='cuda', dtype=torch.float64)
    L.call('se_conv3to64_fwd', L.ptr(x), L.ptr(w), L.ptr(bias), L.ptr(R), L.ptr(st), C.c_int(B), C.c_long(P), L.stream())
    wd, bd = w.view(64, 3).double().requires_grad_(True), bias.double().requires_grad_(True)
    ref = x[..., :3].double() @ wd.t() + bd
    assert relerr(R, ref.detach()) < 2e-6
    rr = R.double()
    assert relerr(st[..., 0], rr.sum(1)) < 1e-6 and relerr(st[..., 1], (rr * rr).sum(1)) < 1e-6
    dR = rnd(B, P, 64, seed=4)
    ref.backward(dR.double())
    dw, db = torch.full((64, 3, 1, 1), 0.25, device='cuda'), torch.full((64,), 0.5, device='cuda')
    L.call('se_conv3to64_wgrad', L.ptr(x), L.ptr(dR), L.ptr(dw), L.ptr(db), C.c_long(B * P), L.stream())
    assert relerr(dw.view(64, 3) - 0.25, wd.grad) < 5e-6 and relerr(db - 0.5, bd.grad) < 5e-6


@pytest.mark.parametrize('B,Ti,Fi,Cin,N', [(2, 160, 100, 16, 32), (2, 80, 50, 32, 64), (1, 40, 25, 64, 128), (3, 37, 21, 16, 32),
                                            (1, 2, 2, 32, 32), (2, 9, 300, 4, 64)])
def test_discriminator_input_gradient_by_parity_class(G, B, Ti, Fi, Cin, N):
    """se_dconv_dgrad (csrc/se_thin.hip) vs torch autograd in fp64: the input gradient of Conv2d(Cin, N, 4, 2, 1)
    (models/discriminator.py:42-50) on the transposed [T, F] image, one workgroup per 128 pixels of a parity class"""
    import ctypes as C
    gemm, L = G
    To, Fo = (Ti - 2) // 2 + 1, (Fi - 2) // 2 + 1
    w = rnd(N, Cin, 4, 4, seed=1, scale=(16 * Cin) ** -0.5)
    dR = rnd(B, To, Fo, N, seed=2)
    wd = gemm.pack_conv_dgrad(w)                               # [Cin][16][N]
    dx = torch.full((B, Ti, Fi, Cin), 7.0, device='cuda')
    L.call('se_dconv_dgrad', L.ptr(dR), L.ptr(wd), L.ptr(dx), C.c_int(B), C.c_int(Ti), C.c_int(Fi), C.c_int(N), C.c_int(Cin), L.stream())
    xi = torch.zeros(B, Cin, Fi, Ti, device='cuda', dtype=torch.float64, requires_grad=True)
    ref = F.conv2d(xi, w.double(), None, 2, 1)                 # [B, N, Fo, To]
    ref.backward(dR.permute(0, 3, 2, 1).double())
    assert relerr(dx, xi.grad.permute(0, 3, 2, 1)) < 3e-6


@pytest.mark.parametrize('M', [37, 4096 + 37, 70000])
def test_feed_forward_fused_backward(M):
    """se_ff_bwd_fused (csrc/se_ff_fused.hip): ONE persistent launch for dX, dgamma / dbeta and dW1 / db1 / dW2 / db2 of the module, H,
    S and dZ recomputed on chip -- against the stored-H kernels (ff_bwd_dgrad + the fp32-MFMA whole-gradient kernels on the stored
    H / dZ) and, without dropout, against fp64; with and without dropout (the same counter-based masks), with and without the
    second residual, accumulating into non-zero gradient buffers; row counts: less than one tile, a ragged last workgroup, many
    workgroups."""
    from speech_enhancement_amd import gemm as GM, _lib as L, ops as O
    from speech_enhancement_amd.weights import WeightPlan
    dev = torch.device('cuda')
    torch.manual_seed(M)
    x = torch.randn(M, 64, device=dev)
    st = O.row_stats(x, M)
    g, b = torch.rand(64, device=dev) + 0.5, torch.randn(64, device=dev) * 0.1
    W1, b1 = torch.randn(256, 64, device=dev) * 0.1, torch.randn(256, device=dev) * 0.1
    W2, b2 = torch.randn(64, 256, device=dev) * 0.1, torch.randn(64, device=dev) * 0.1
    p = WeightPlan(dev)
    p.linear('w1', W1, planes='f16'); p.linear('w2', W2, planes='f16')
    p.linear_T('w2t', W2, planes='f16', scale=0.5); p.linear_T('w1t', W1, planes='f16')
    p.run()
    rel = lambda a, r: float((a.double() - r.double()).abs().max() / r.double().abs().max())
    for drop, with_r2 in ((0.0, False), (0.2, True)):
        dy = torch.randn(M, 64, device=dev) * 1e-3
        dy._se_amax = dy.abs().max().reshape(1).clone()
        dR2 = torch.randn(M, 64, device=dev) * 1e-3 if with_r2 else None
        y, h = GM.ff_fwd(x, st, g, b, p.out['w1'], b1, p.out['w2'], b2, drop, 11, 12, 0.5)
        dg0, db0 = torch.zeros(64, device=dev), torch.zeros(64, device=dev)
        dz, dx0 = GM.ff_bwd_dgrad(dy, h, p.out['w2t'], p.out['w1t'], drop, 11, 12, ln=(x, st, g, dR2, dg0, db0),
                                  amax_out=(torch.zeros(1, device=dev), torch.zeros(1, device=dev)))
        dr = drop > 0
        dW1a, db1a, dW2a, db2a = (torch.zeros(s, device=dev) for s in ((256, 64), (256,), (64, 256), (64,)))
        GM.gemm_tap_wgrad(GM.linear_desc(M, 256, 64, prologue=L.PRO_SWISH_DROP if dr else L.PRO_SWISH, epilogue=L.EPI_DROP if dr else 0,
                                         pro_seed=11, epi_seed=12, drop_p=drop, precision=0), h, dy, dW2a, db2a, scale=0.5,
                          explicit_precision=True)
        GM.gemm_tap_wgrad(GM.linear_desc(M, 64, 256, prologue=L.PRO_LN, precision=0), x, dz, dW1a, db1a, rowstats=st, ps=g, pb=b,
                          explicit_precision=True)
        # the fused launch accumulates into buffers that already hold something
        init = [torch.randn(s, device=dev) * 1e-2 for s in ((256, 64), (256,), (64, 256), (64,), (64,), (64,))]
        dW1b, db1b, dW2b, db2b, dg1, db1_ = [t.clone() for t in init]
        dx1 = GM.ff_bwd_fused(dy, x, st, g, b, p.out['w1'], b1, p.out['w2t'], dW1b, db1b, dW2b, db2b, dg1, db1_, drop, 11, 12, 0.5,
                              dR2=dR2, out_amax=torch.zeros(1, device=dev))
        torch.cuda.synchronize()
        assert torch.isfinite(dx1).all()
        assert rel(dx1, dx0) < 2e-6, ('dX', rel(dx1, dx0))
        assert abs(float(dx1._se_amax) - float(dx1.abs().max())) <= 1e-6 * float(dx1.abs().max())
        for name, got, ini, ref in (('dW1', dW1b, init[0], dW1a), ('db1', db1b, init[1], db1a), ('dW2', dW2b, init[2], dW2a),
                                    ('db2', db2b, init[3], db2a), ('dgamma', dg1, init[4], dg0), ('dbeta', db1_, init[5], db0)):
            e = rel(got - ini, ref)
            assert e < (4e-6 if name in ('dgamma', 'dbeta') else 3e-6) + 2e-7 * float(ini.abs().max() / ref.abs().max()), (name, e, drop, M)
        if not dr:
            xl = ((x.double() - st[:, :1].double()) * st[:, 1:].double()) * g.double() + b.double()
            h64 = xl @ W1.double().t() + b1.double()
            sg = torch.sigmoid(h64)
            dz64 = (dy.double() @ (0.5 * W2.double())) * (sg * (1 + h64 * (1 - sg)))
            assert rel(dW1b - init[0], dz64.t() @ xl) < 2e-6 and rel(dW2b - init[2], 0.5 * dy.double().t() @ (h64 * sg)) < 2e-6
            assert rel(db1b - init[1], dz64.sum(0)) < 2e-6 and rel(db2b - init[3], 0.5 * dy.double().sum(0)) < 2e-6


def test_feed_forward_fused_backward_optional_operands():
    """se_ff_bwd_fused without its optional operands (no second residual, no b2 gradient, no output maximum) gives the same dX / dW as
    with them, repeated launches with the same seeds are bit-identical in dX (the weight gradients are fp32 atomics: order-dependent in
    the last bits), and a row count that is not a multiple of the 32-row tile leaves the rows past M untouched."""
    from speech_enhancement_amd import gemm as GM, ops as O
    from speech_enhancement_amd.weights import WeightPlan
    dev = torch.device('cuda')
    torch.manual_seed(7)
    M = 3 * 2048 + 19
    xf = torch.randn(M + 5, 64, device=dev)
    x = xf[:M]
    st = O.row_stats(x, M)
    g, b = torch.rand(64, device=dev) + 0.5, torch.randn(64, device=dev) * 0.1
    W1, b1 = torch.randn(256, 64, device=dev) * 0.1, torch.randn(256, device=dev) * 0.1
    W2 = torch.randn(64, 256, device=dev) * 0.1
    p = WeightPlan(dev)
    p.linear('w1', W1, planes='f16'); p.linear_T('w2t', W2, planes='f16', scale=0.5)
    p.run()
    dy = torch.randn(M, 64, device=dev) * 1e-3
    dy._se_amax = dy.abs().max().reshape(1).clone()

    def run(db2, amax):
        gr = [torch.zeros(s, device=dev) for s in ((256, 64), (256,), (64, 256))]
        dg, dbt = torch.zeros(64, device=dev), torch.zeros(64, device=dev)
        dx = GM.ff_bwd_fused(dy, x, st, g, b, p.out['w1'], b1, p.out['w2t'], gr[0], gr[1], gr[2], db2, dg, dbt, 0.2, 11, 12, 0.5,
                             dR2=None, out_amax=amax)
        torch.cuda.synchronize()
        return dx, gr, dg, dbt
    db2, am = torch.zeros(64, device=dev), torch.zeros(1, device=dev)
    dx0, gr0, dg0, dbt0 = run(db2, am)
    dx1, gr1, dg1, dbt1 = run(None, None)
    dx2, _, _, _ = run(None, None)
    assert torch.equal(dx0, dx1) and torch.equal(dx1, dx2)
    assert float(am) == float(dx0.abs().max())
    rel = lambda a, r: float((a - r).abs().max() / r.abs().max())
    for a_, r_ in zip(gr1 + [dg1, dbt1], gr0 + [dg0, dbt0]):
        assert rel(a_, r_) < 2e-6
    assert float(db2.abs().max()) > 0


@pytest.mark.parametrize('K', [192, 256])
@pytest.mark.parametrize('M', [21, 4096 + 37, 60001])
def test_ln_bwd_gemm_with_weight_gradient_in_one_sweep(M, K):
    """se_gemm_ln_bwd_wgrad (csrc/se_lnbwd_fused.hip): dX = dR + LNbwd(A W), dgamma / dbeta AND dW [K, 64] += A^T LN(x), db += sum A from
    ONE pass over A and x -- against se_gemm_ln_bwd_f16 + the fp32-MFMA weight-gradient kernel and against fp64; K = 192 (qkv: no bias,
    no ... ) and 256 (pointwise-GLU: bias); row counts below one tile, ragged, many workgroups; accumulating into non-zero buffers."""
    from speech_enhancement_amd import gemm as GM, _lib as L, ops as O
    from speech_enhancement_amd.weights import WeightPlan
    dev = torch.device('cuda')
    torch.manual_seed(M + K)
    x = torch.randn(M, 64, device=dev) * 2 + 0.3
    st = O.row_stats(x, M)
    g, b = torch.rand(64, device=dev) + 0.5, torch.randn(64, device=dev) * 0.1
    W = torch.randn(K, 64, device=dev) * 0.1                     # y = LN(x) W^T
    A = torch.randn(M, K, device=dev) * 1e-3
    A._se_amax = A.abs().max().reshape(1).clone()
    dR = torch.randn(M, 64, device=dev) * 1e-3 if K == 256 else None
    plan = WeightPlan(dev)
    WT = plan.linear_T('wt', W, planes='f16')                    # [2][64][K]
    plan.run()
    rel = lambda a_, r_: float((a_.double() - r_.double()).abs().max() / r_.double().abs().max())
    dg0, db0 = torch.zeros(64, device=dev), torch.zeros(64, device=dev)
    dx0 = GM.gemm_ln_bwd(A, WT, x, st, g, dR, dg0, db0, out_amax=torch.zeros(1, device=dev))
    dW0, dbias0 = torch.zeros(K, 64, device=dev), torch.zeros(K, device=dev)
    GM.gemm_tap_wgrad(GM.linear_desc(M, 64, K, prologue=L.PRO_LN, precision=0), x, A, dW0, dbias0, rowstats=st, ps=g, pb=b,
                      explicit_precision=True)
    init = [torch.randn(s, device=dev) * 1e-2 for s in ((64,), (64,), (K, 64), (K,))]
    dg1, db1, dW1, dbias1 = [t.clone() for t in init]
    dx1 = GM.gemm_ln_bwd_wgrad(A, WT, x, st, g, b, dR, dg1, db1, dW1, dbias1 if K == 256 else None, out_amax=torch.zeros(1, device=dev))
    torch.cuda.synchronize()
    assert torch.isfinite(dx1).all()
    assert rel(dx1, dx0) < 2e-6, ('dX', rel(dx1, dx0))
    assert abs(float(dx1._se_amax) - float(dx1.abs().max())) <= 1e-6 * float(dx1.abs().max())
    checks = [('dgamma', dg1, init[0], dg0), ('dbeta', db1, init[1], db0), ('dW', dW1, init[2], dW0)]
    if K == 256:
        checks.append(('dbias', dbias1, init[3], dbias0))
    for name, got, ini, ref in checks:
        e = rel(got - ini, ref)
        assert e < 4e-6 + 2e-7 * float(ini.abs().max() / ref.abs().max()), (name, e, M, K)
    xl = ((x.double() - st[:, :1].double()) * st[:, 1:].double()) * g.double() + b.double()
    assert rel(dW1 - init[2], A.double().t() @ xl) < 2e-6
    if K == 256:
        assert rel(dbias1 - init[3], A.double().sum(0)) < 2e-6
