"""GPU parity of the normalisation / depthwise / elementwise kernels against plain PyTorch references."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).cuda()


def relerr(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / max(float(b.abs().max()), 1e-6))


def test_layernorm_fwd_bwd():
    from speech_enhancement_amd import ops as O
    M = 1237
    x, g, b, r, dy = rnd(M, 64, seed=1) * 2 + 0.3, rnd(64, seed=2) * 0.1 + 1, rnd(64, seed=3) * 0.1, rnd(M, 64, seed=4), rnd(M, 64, seed=5)
    y, st = O.layernorm_fwd(x, g, b, R=r)
    x64 = x.double().requires_grad_(True)
    g64, b64 = g.double().requires_grad_(True), b.double().requires_grad_(True)
    ref = F.layer_norm(x64, (64,), g64, b64, 1e-5) + r.double()
    assert relerr(y, ref) < 1e-5
    st2 = O.row_stats(x, M)
    assert relerr(st2, st) < 1e-5
    ref.backward(dy.double())
    dg, db = torch.zeros(64, device='cuda'), torch.zeros(64, device='cuda')
    dx = O.layernorm_bwd(x, st, g, dy, dg, db, dR=r, dR2=r)
    assert relerr(dx, x64.grad + 2 * r.double()) < 2e-5
    assert relerr(dg, g64.grad) < 2e-5 and relerr(db, b64.grad) < 2e-5


@pytest.mark.parametrize('C', [64, 4, 16, 128])
def test_instance_norm_prelu(C):
    from speech_enhancement_amd import ops as O
    B, T, Fq = 3, 13, 29
    P = T * Fq
    x = rnd(B, T, Fq, C, seed=1) * 1.7 + 0.4
    g, be, a = rnd(C, seed=2) * 0.2 + 1, rnd(C, seed=3) * 0.2, rnd(C, seed=4) * 0.1 + 0.25
    dy = rnd(B, T, Fq, 2 * C, seed=5)
    stats = O.col_stats(x, C, 0, B, P, C)
    mr, ss = O.norm_finalize(stats, g, be, B, C, float(P))
    y = torch.zeros(B, T, Fq, 2 * C, device='cuda')
    O.affine_prelu(x, C, 0, ss, a, y, 2 * C, C, B, P, C)
    x64 = x.double().requires_grad_(True)
    p64 = [t.double().requires_grad_(True) for t in (g, be, a)]
    xn = F.instance_norm(x64.permute(0, 3, 1, 2), weight=p64[0], bias=p64[1], eps=1e-5)
    ref = F.prelu(xn, p64[2]).permute(0, 2, 3, 1)
    assert relerr(y[..., C:], ref) < 1e-5 and float(y[..., :C].abs().max()) == 0.0
    # the one-launch form (se_inorm_prelu_fwd: finalize + apply) gives the same bits and the same (mean, rstd)
    y1 = torch.zeros(B, T, Fq, 2 * C, device='cuda')
    mr1 = O.inorm_prelu_fwd(x, C, 0, stats, g, be, a, y1, 2 * C, C, B, P, C)
    assert torch.equal(y1, y) and torch.equal(mr1, mr)
    ref.backward(dy[..., C:].double())
    dg, db, da = (torch.zeros(C, device='cuda') for _ in range(3))
    dx = torch.empty_like(x)
    O.norm_prelu_bwd(x, C, 0, mr, g, be, a, dy, 2 * C, C, dx, C, 0, dg, db, da, B, P, C, per_batch=True)
    assert relerr(dx, x64.grad) < 5e-5
    assert relerr(dg, p64[0].grad) < 5e-5 and relerr(db, p64[1].grad) < 5e-5 and relerr(da, p64[2].grad) < 5e-5
    # the exchange form (reduce + local parameter gradients, [all-reduce], apply) against the fused two-launch form
    dg2, db2, da2 = (torch.zeros(C, device='cuda') for _ in range(3))
    dx2 = torch.empty_like(x)
    seen = []
    O.norm_prelu_bwd(x, C, 0, mr, g, be, a, dy, 2 * C, C, dx2, C, 0, dg2, db2, da2, B, P, C, per_batch=True,
                     allreduce=lambda red: seen.append(tuple(red.shape)))
    assert seen and relerr(dx2, dx) < 1e-6
    assert relerr(dg2, dg) < 1e-6 and relerr(db2, db) < 1e-6 and relerr(da2, da) < 1e-6


@pytest.mark.parametrize('B,P,C,ld', [(3, 377, 64, 64), (16, 4000, 64, 256), (5, 12345, 32, 32), (2, 70001, 64, 64), (7, 250, 128, 128),
                                        (16, 32421, 64, 64), (33, 1000, 16, 16)])
def test_instance_norm_prelu_backward_strided_and_ragged(B, P, C, ld):
    """se_norm_prelu_bwd (InstanceNorm(affine) + PReLU backward: reduce + apply launches) on strided operands (a channel slab of a wider
    map), ragged pixel counts and every channel count of the model, accumulating parameter gradients, against torch autograd in fp64.
    (The one-pass form of round 4 -- tiles kept in registers across an in-kernel rendezvous -- was correct but slower: removed.)"""
    from speech_enhancement_amd import ops as O
    x = rnd(B, P, ld, seed=1) * 1.7 + 0.4
    off = ld - C
    g, be, a = rnd(C, seed=2) * 0.2 + 1, rnd(C, seed=3) * 0.2, rnd(C, seed=4) * 0.1 + 0.25
    dy = rnd(B, P, ld, seed=5)
    stats = O.col_stats(x, ld, off, B, P, C)
    mr, _ = O.norm_finalize(stats, g, be, B, C, float(P))
    dg1, db1, da1 = (torch.full((C,), 0.5, device='cuda') for _ in range(3))
    dx1 = torch.full((B, P, ld), 3.0, device='cuda')
    am1 = torch.zeros(1, device='cuda')
    O.norm_prelu_bwd(x, ld, off, mr, g, be, a, dy, ld, off, dx1, ld, off, dg1, db1, da1, B, P, C, per_batch=True, amax=am1)
    torch.cuda.synchronize()
    if off:
        assert float((dx1[..., :off] - 3.0).abs().max()) == 0.0          # columns outside the slab untouched
    assert abs(float(am1) - float(dx1[..., off:].abs().max())) <= 1e-6 * float(am1)
    # and against torch autograd in fp64
    x64 = x[..., off:].double().requires_grad_(True)
    p64 = [t.double().requires_grad_(True) for t in (g, be, a)]
    xn = F.instance_norm(x64.permute(0, 2, 1), weight=p64[0], bias=p64[1], eps=1e-5)
    F.prelu(xn, p64[2]).permute(0, 2, 1).backward(dy[..., off:].double())
    assert relerr(dx1[..., off:], x64.grad) < 5e-5
    assert relerr(dg1 - 0.5, p64[0].grad) < 5e-5 and relerr(db1 - 0.5, p64[1].grad) < 5e-5 and relerr(da1 - 0.5, p64[2].grad) < 5e-5


def test_batchnorm_swish_bwd_and_running_stats():
    from speech_enhancement_amd import ops as O
    M, C = 5000, 128
    x = rnd(M, C, seed=1) * 1.3 - 0.2
    g, be = rnd(C, seed=2) * 0.2 + 1, rnd(C, seed=3) * 0.2
    dy = rnd(M, C, seed=4)
    rm, rv = rnd(C, seed=6) * 0.1, rnd(C, seed=7).abs() + 0.5
    rm0, rv0 = rm.clone(), rv.clone()
    stats = O.col_stats(x, C, 0, 1, M, C)
    mr, ss = O.norm_finalize(stats, g, be, 1, C, float(M), running_mean=rm, running_var=rv, momentum=0.1)
    x64 = x.double().requires_grad_(True)
    g64, b64 = g.double().requires_grad_(True), be.double().requires_grad_(True)
    rm64, rv64 = rm0.double(), rv0.double()
    ref = F.silu(F.batch_norm(x64, rm64, rv64, g64, b64, True, 0.1, 1e-5))
    assert relerr(rm, rm64) < 1e-5 and relerr(rv, rv64) < 1e-5
    ref.backward(dy.double())
    dg, db = torch.zeros(C, device='cuda'), torch.zeros(C, device='cuda')
    dx = torch.empty_like(x)
    O.norm_prelu_bwd(x, C, 0, mr, g, be, None, dy, C, 0, dx, C, 0, dg, db, None, 1, M, C, per_batch=False, act=1)
    assert relerr(dx, x64.grad) < 5e-5 and relerr(dg, g64.grad) < 5e-5 and relerr(db, b64.grad) < 5e-5


@pytest.mark.parametrize('axis,B,T,Fq', [('time', 2, 70, 5), ('freq', 2, 3, 101), ('time', 1, 321, 2), ('freq', 1, 2, 17),
                                         # tile-size boundaries of the kernel: 64 | 65 (8 -> 13 positions per slot), 104 | 105
                                         # (13 -> 14), 112 | 113 (one -> two tiles), and a long sequence (15 tiles)
                                         ('freq', 1, 2, 64), ('freq', 1, 2, 65), ('freq', 2, 1, 104), ('freq', 1, 2, 105),
                                         ('time', 1, 112, 3), ('time', 1, 113, 2), ('time', 1, 1601, 1)])
def test_dwconv31(axis, B, T, Fq):
    from speech_enhancement_amd import ops as O, attention as A
    x = rnd(B, T, Fq, 128, seed=1)
    w, b = rnd(128, 1, 31, seed=2, scale=0.2), rnd(128, seed=3)
    dy = rnd(B, T, Fq, 128, seed=4)
    geom = A.seq_geometry(B, T, Fq, axis)
    stats = torch.zeros(1, 128, 2, device='cuda', dtype=torch.float64)
    y = O.dwconv31(x.view(-1, 128), w.view(128, 31), b, geom, stats=stats).view(B, T, Fq, 128)
    x64 = x.double().requires_grad_(True)
    w64, b64 = w.double().requires_grad_(True), b.double().requires_grad_(True)
    seqs = x64.permute(0, 2, 3, 1) if axis == 'time' else x64.permute(0, 1, 3, 2)          # [B, o, 128, n]
    sh = seqs.shape
    r = F.conv1d(F.pad(seqs.reshape(-1, 128, sh[-1]), (15, 15)), w64, b64, groups=128).reshape(sh)
    ref = r.permute(0, 3, 1, 2) if axis == 'time' else r.permute(0, 1, 3, 2)
    assert relerr(y, ref) < 1e-5
    assert relerr(stats[0, :, 0], ref.sum((0, 1, 2))) < 1e-5 and relerr(stats[0, :, 1], (ref ** 2).sum((0, 1, 2))) < 1e-5
    ref.backward(dy.double())
    dx = O.dwconv31(dy.view(-1, 128), w.view(128, 31), None, geom, flip=True).view(B, T, Fq, 128)
    assert relerr(dx, x64.grad) < 1e-5
    dw, db = torch.zeros(128, 31, device='cuda'), torch.zeros(128, device='cuda')
    O.dwconv31_wgrad(x.view(-1, 128), dy.view(-1, 128), dw, db, geom)
    assert relerr(dw, w64.grad.view(128, 31)) < 2e-5 and relerr(db, b64.grad) < 2e-5
    # the input gradient fused with the backward of the GLU in front of the conv (se_dwconv31_glu_bwd): fp64 autograd of
    # conv(GLU(z)) w.r.t. z, and the measured max |dZ|
    z = rnd(B, T, Fq, 256, seed=5)
    z64 = z.double().requires_grad_(True)
    u64 = z64[..., :128] * torch.sigmoid(z64[..., 128:])
    sq = u64.permute(0, 2, 3, 1) if axis == 'time' else u64.permute(0, 1, 3, 2)
    rr = F.conv1d(F.pad(sq.reshape(-1, 128, sq.shape[-1]), (15, 15)), w.double(), None, groups=128).reshape(sq.shape)
    (rr.permute(0, 3, 1, 2) if axis == 'time' else rr.permute(0, 1, 3, 2)).backward(dy.double())
    amax = torch.zeros(1, device='cuda')
    u = (z[..., :128] * torch.sigmoid(z[..., 128:])).contiguous()            # what a GEMM with EPI_GLU | EPI_GLU_GATE keeps:
    gate = z[..., 128:].contiguous()                                          # the GLU result and the gate half
    dz = O.dwconv31_glu_bwd(dy.view(-1, 128), w.view(128, 31), u.view(-1, 128), gate.view(-1, 128), geom, amax=amax)
    assert relerr(dz.view(B, T, Fq, 256), z64.grad) < 1e-5
    assert abs(float(amax) - float(dz.abs().max())) <= 1e-6 * float(amax)


@pytest.mark.parametrize('axis,B,T,Fq', [('time', 2, 37, 5), ('freq', 2, 7, 101), ('time', 1, 321, 9), ('freq', 3, 5, 113),
                                         ('time', 2, 112, 3), ('freq', 1, 300, 16), ('time', 1, 16, 2)])
def test_dwconv31_backward_in_one_sweep(axis, B, T, Fq):
    """se_dwconv31_bwd_fused: dZ, max |dZ| and the ACCUMULATED weight / bias gradient against fp64 autograd of conv(GLU(z)) and
    against the two-launch path; sequence lengths below / at / above the 112-position tile, both token-stride geometries, more
    (sequence, tile) items than workgroups and fewer"""
    from speech_enhancement_amd import ops as O, attention as A
    w = rnd(128, 1, 31, seed=2, scale=0.2)
    dy = rnd(B, T, Fq, 128, seed=4)
    z = rnd(B, T, Fq, 256, seed=5)
    geom = A.seq_geometry(B, T, Fq, axis)
    z64 = z.double().requires_grad_(True)
    w64, b64 = w.double().requires_grad_(True), torch.zeros(128, device='cuda', dtype=torch.float64, requires_grad=True)
    u64 = z64[..., :128] * torch.sigmoid(z64[..., 128:])
    sq = u64.permute(0, 2, 3, 1) if axis == 'time' else u64.permute(0, 1, 3, 2)
    rr = F.conv1d(F.pad(sq.reshape(-1, 128, sq.shape[-1]), (15, 15)), w64, b64, groups=128).reshape(sq.shape)
    (rr.permute(0, 3, 1, 2) if axis == 'time' else rr.permute(0, 1, 3, 2)).backward(dy.double())
    u = (z[..., :128] * torch.sigmoid(z[..., 128:])).contiguous()
    gate = z[..., 128:].contiguous()
    amax = torch.zeros(1, device='cuda')
    dw0, db0 = rnd(128, 31, seed=7), rnd(128, seed=8)                       # accumulated into
    dw, db = dw0.clone(), db0.clone()
    dz = O.dwconv31_bwd_fused(dy.view(-1, 128), w.view(128, 31), u.view(-1, 128), gate.view(-1, 128), dw, db, geom, amax=amax)
    assert relerr(dz.view(B, T, Fq, 256), z64.grad) < 1e-5
    assert abs(float(amax) - float(dz.abs().max())) <= 1e-6 * float(amax)
    assert relerr(dw - dw0, w64.grad.view(128, 31)) < 2e-5 and relerr(db - db0, b64.grad) < 2e-5
    dz2 = O.dwconv31_glu_bwd(dy.view(-1, 128), w.view(128, 31), u.view(-1, 128), gate.view(-1, 128), geom)
    dw2, db2 = torch.zeros(128, 31, device='cuda'), torch.zeros(128, device='cuda')
    O.dwconv31_wgrad(u.view(-1, 128), dy.view(-1, 128), dw2, db2, geom)
    assert relerr(dz, dz2) < 2e-6 and relerr(dw - dw0, dw2) < 2e-5 and relerr(db - db0, db2) < 2e-5
    dbn = None                                                              # bias gradient is optional
    dw3 = torch.zeros(128, 31, device='cuda')
    O.dwconv31_bwd_fused(dy.view(-1, 128), w.view(128, 31), u.view(-1, 128), gate.view(-1, 128), dw3, dbn, geom)
    assert relerr(dw3, dw2) < 2e-5


def test_glu_bwd_and_optimizers():
    from speech_enhancement_amd import ops as O
    M = 777
    z, du = rnd(M, 256, seed=1), rnd(M, 128, seed=2)
    z64 = z.double().requires_grad_(True)
    (z64[:, :128] * torch.sigmoid(z64[:, 128:])).backward(du.double())
    assert relerr(O.glu_bwd(z, du, M, 128), z64.grad) < 1e-5
    u = (z[:, :128] * torch.sigmoid(z[:, 128:])).contiguous()
    assert relerr(O.glu_bwd_gate(u, z[:, 128:].contiguous(), du, M, 128), z64.grad) < 1e-5
    # AdamW / nesterov SGD vs torch.optim on one flat tensor
    p0, g1, g2 = rnd(1000, seed=3), rnd(1000, seed=4), rnd(1000, seed=5)
    for kind in ('adamw', 'sgd'):
        p = torch.nn.Parameter(p0.clone().double())
        opt = torch.optim.AdamW([p], lr=1e-2, weight_decay=0.01) if kind == 'adamw' else \
            torch.optim.SGD([p], lr=1e-2, momentum=0.9, nesterov=True)
        q, m, v = p0.clone(), torch.zeros_like(p0), torch.zeros_like(p0)
        for step, g in enumerate((g1, g2), 1):
            p.grad = g.double()
            opt.step()
            if kind == 'adamw':
                O.adamw(q, g, m, v, 1e-2, 0.9, 0.999, 1e-8, 0.01, step)
            else:
                O.sgd_nesterov(q, g, m, 1e-2, 0.9, step == 1)
        assert relerr(q, p.data) < 1e-5, kind


def test_zero_arena_hands_out_zeroed_disjoint_buffers():
    """ops.ZeroArena: step-scoped zero-filled scratch (one fill per step); outside a step plain torch.zeros."""
    from speech_enhancement_amd import ops as O
    dev = torch.device('cuda')
    assert not O.ARENA.active
    plain = O.zeros(3, 5, device=dev)
    assert plain.eq(0).all()
    for _ in range(2):                       # second step re-clears what the first one dirtied
        O.ARENA.begin(dev, min_bytes=1 << 16)
        try:
            a = O.zeros(7, 3, device=dev)
            b = O.zeros(2, 5, 2, device=dev, dtype=torch.float64)
            big = O.zeros(1 << 20, device=dev)             # does not fit the 64 KB arena: falls back, grown next step
            assert a.eq(0).all() and b.eq(0).all() and big.eq(0).all() and b.data_ptr() % 256 == 0
            assert a.data_ptr() + a.numel() * 4 <= b.data_ptr() or b.data_ptr() + b.numel() * 8 <= a.data_ptr()
            a.fill_(3.0); b.fill_(5.0)
        finally:
            O.ARENA.end()
    O.ARENA.__init__()


def test_layernorm_fwd_emits_the_statistics_of_its_result():
    """se_layernorm_fwd_stats: (mean, rstd) of the rows of Y = LN(X) g + b + R come out of the same pass and equal se_row_stats(Y)"""
    from speech_enhancement_amd import ops as O
    M = 16 * 37 + 5
    x, r = rnd(M, 64, seed=1) * 1.5 + 0.3, rnd(M, 64, seed=2)
    g, b = rnd(64, seed=3) * 0.2 + 1, rnd(64, seed=4) * 0.1
    y, st, ost = O.layernorm_fwd(x, g, b, R=r, out_stats=True)
    y0, st0 = O.layernorm_fwd(x, g, b, R=r)
    assert torch.equal(y, y0) and torch.equal(st, st0)
    ref = O.row_stats(y, M)
    assert relerr(ost, ref) < 1e-6
