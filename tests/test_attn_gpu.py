"""GPU parity of the fused relative-position attention against a plain PyTorch fp64 restatement."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def ref_attention(qkv, E, B, T, Fq, axis, maxpos, scale):
    """qkv [B,T,Fq,192] fp64 -> O [B,T,Fq,64] (materialises n x n; test sizes only)."""
    x = qkv if axis == 'freq' else qkv.transpose(1, 2)          # [B, outer, n, 192]
    n = x.shape[2]
    q, k, v = x[..., :64], x[..., 64:128], x[..., 128:]
    sp = lambda t: t.reshape(*t.shape[:3], 4, 16).transpose(2, 3)   # [B, outer, 4, n, 16]
    q, k, v = sp(q), sp(k), sp(v)
    idx = torch.arange(n, device=qkv.device)
    rel = (idx[:, None] - idx[None, :]).clamp(-maxpos, maxpos) + maxpos
    Er = E[rel]                                                     # [n, n, 16]
    logits = (q @ k.transpose(-1, -2) + torch.einsum('bohid,ijd->bohij', q, Er)) * scale
    o = torch.softmax(logits, -1) @ v                               # [B, outer, 4, n, 16]
    o = o.transpose(2, 3).reshape(*x.shape[:3], 64)
    return o if axis == 'freq' else o.transpose(1, 2)


def relerr(a, b):
    a, b = a.double(), b.double()
    a, b = a.detach(), b.detach()
    return float((a - b).abs().max() / max(float(b.abs().max()), 1e-1))


CASES = [(2, 37, 19, 'time', 512), (2, 37, 19, 'freq', 512), (1, 321, 3, 'time', 512), (3, 5, 101, 'freq', 512),
         (1, 40, 2, 'time', 8), (2, 3, 50, 'freq', 5), (1, 600, 1, 'time', 512), (1, 16, 4, 'time', 512),
         (1, 1, 1, 'freq', 512),
         (1, 1601, 1, 'time', 512)]       # 10 s utterance (inference shape): K/V exceed LDS -> streaming kernels


@pytest.mark.parametrize('B,T,Fq,axis,maxpos', CASES)
def test_attention_fwd_bwd(B, T, Fq, axis, maxpos):
    from speech_enhancement_amd import attention as A
    g = torch.Generator().manual_seed(B * 1000 + T * 10 + Fq)
    qkv = (torch.randn(B, T, Fq, 192, generator=g) * 1.5).cuda()
    E = (torch.randn(2 * maxpos + 1, 16, generator=g) * 0.7).cuda()
    dO = torch.randn(B, T, Fq, 64, generator=g).cuda()
    geom = A.seq_geometry(B, T, Fq, axis)
    O, lse = A.attn_fwd(qkv.view(-1, 192), E, geom, maxpos=maxpos)
    q64 = qkv.double().requires_grad_(True)
    E64 = E.double().requires_grad_(True)
    ref = ref_attention(q64, E64, B, T, Fq, axis, maxpos, 0.25)
    assert relerr(O.view(B, T, Fq, 64), ref) < 5e-6
    ref.backward(dO.double())
    dE = torch.zeros_like(E)
    dqkv = A.attn_bwd(qkv.view(-1, 192), E, O, dO.view(-1, 64), lse, geom, dE, maxpos=maxpos)
    dq = dqkv.view(B, T, Fq, 192)
    assert relerr(dq[..., :64], q64.grad[..., :64]) < 2e-5, 'dq'
    assert relerr(dq[..., 64:128], q64.grad[..., 64:128]) < 2e-5, 'dk'
    assert relerr(dq[..., 128:], q64.grad[..., 128:]) < 2e-5, 'dv'
    assert relerr(dE, E64.grad) < 2e-5, 'dE'


F16_CASES = [(2, 37, 19, 'time'), (3, 5, 101, 'freq'), (1, 321, 3, 'time'), (2, 16, 4, 'time'), (1, 1, 1, 'freq'), (2, 50, 7, 'freq'),
             # round 4 (workgroup-cooperative backward, se_attn_bwd4.h): every key-tile split of its two instantiations -- 8 .. 21 key
             # tiles over four waves (113, 130, 200: the generic body; 336: a full tail tile), 7 full tiles over two waves (112).  (Lengths
             # beyond 21 key tiles have no scaled-fp16 backward since round 6: they take the fp32 kernels, test_attention_fwd_bwd.)
             (1, 113, 2, 'time'), (1, 130, 2, 'time'), (1, 200, 2, 'time'), (1, 336, 1, 'time'), (2, 3, 112, 'freq')]


@pytest.mark.parametrize('B,T,Fq,axis', F16_CASES)
@pytest.mark.parametrize('mag', [1.5, 40.0, 0.02])
def test_attention_scaled_fp16(B, T, Fq, axis, mag):
    """the scaled split-fp16 kernels (se_attn_fwd_f16 / se_attn_bwd_f16: two fp16 planes per operand, scales from measured maxima)
    against the same fp64 restatement, at the bars of the split-bf16 kernels; operand magnitudes over three decades"""
    from speech_enhancement_amd import attention as A
    from speech_enhancement_amd.weights import WeightPlan
    maxpos = 512
    if T * Fq == 1 and mag > 1.5:
        pytest.skip('one key: dS = P (dP - D) is pure cancellation noise (2^-22 |dO| |V|), not comparable with the zero reference')
    g = torch.Generator().manual_seed(B * 1000 + T * 10 + Fq)
    qkv = (torch.randn(B, T, Fq, 192, generator=g) * mag).cuda()
    if mag > 1.5:
        qkv[..., :128] *= 1.5 / mag                                  # large V, logits as in the base case (the softmax amplifies
                                                                     # logit rounding by |logit|: fp32 itself is no better there)
    E = (torch.randn(2 * maxpos + 1, 16, generator=g) * 0.7).cuda()
    dO = (torch.randn(B, T, Fq, 64, generator=g) * 1e-3).cuda()
    geom = A.seq_geometry(B, T, Fq, axis)
    plan = WeightPlan(torch.device('cuda'))
    Es = plan.linear('e', E, planes='f16')
    plan.run()
    amax = qkv.abs().max().reshape(1).clone()
    O, lse = A.attn_fwd(qkv.view(-1, 192), E, geom, maxpos=maxpos, Es=Es, qkv_amax=amax)
    O0, lse0 = A.attn_fwd(qkv.view(-1, 192), E, geom, maxpos=maxpos)
    q64 = qkv.double().requires_grad_(True)
    E64 = E.double().requires_grad_(True)
    ref = ref_attention(q64, E64, B, T, Fq, axis, maxpos, 0.25)
    assert relerr(O.view(B, T, Fq, 64), ref) < 5e-6
    assert float((lse - lse0).abs().max()) < 2e-5 * max(1.0, float(lse0.abs().max()))
    ref.backward(dO.double())
    dE = torch.zeros_like(E)
    do_amax = dO.abs().max().reshape(1).clone()
    dqkv = A.attn_bwd(qkv.view(-1, 192), E, O, dO.view(-1, 64), lse, geom, dE, maxpos=maxpos, qkv_amax=amax, do_amax=do_amax)
    dq = dqkv.view(B, T, Fq, 192)
    for name, sl in (('dq', slice(0, 64)), ('dk', slice(64, 128)), ('dv', slice(128, 192))):
        gref = q64.grad[..., sl]               # (n = 1: dq = dk = 0 exactly, what is computed is the rounding of dP - D: the floor is a tenth of the gradient's scale)
        err = float((dq[..., sl].double() - gref).abs().max() / torch.maximum(gref.abs().max(), 0.1 * q64.grad.abs().max()))
        assert err < 2e-5, (name, err)
    err = float((dE.double() - E64.grad).abs().max() / torch.maximum(E64.grad.abs().max(), 0.1 * q64.grad.abs().max()))
    assert err < 2e-5, ('dE', err)


@pytest.mark.parametrize('B,T,Fq,axis,qsplit', [(1, 1601, 2, 'time', None), (1, 600, 3, 'time', None), (1, 600, 3, 'time', '1'),
                                                (1, 1203, 1, 'time', '5'), (2, 333, 2, 'time', '3')])
def test_attention_scaled_fp16_forward_only_long_sequences(B, T, Fq, axis, qsplit, monkeypatch):
    """inference (no backward, need_lse=False): se_attn_fwd_f16 beyond the training shapes -- the 1601 frames of a 10 s utterance
    (BASELINE config 4), offsets clamped at +-512 (conformer.py:113-114), the query blocks of an item dealt to several workgroups
    (chosen by the launcher for few long sequences; forced through SE_ATTN_FWD_QSPLIT) -- against the fp64 restatement"""
    from speech_enhancement_amd import attention as A
    from speech_enhancement_amd.weights import WeightPlan
    if qsplit is not None:
        monkeypatch.setenv('SE_ATTN_FWD_QSPLIT', qsplit)
    maxpos = 512
    g = torch.Generator().manual_seed(T)
    qkv = (torch.randn(B, T, Fq, 192, generator=g) * 1.2).cuda()
    E = (torch.randn(2 * maxpos + 1, 16, generator=g) * 0.7).cuda()
    geom = A.seq_geometry(B, T, Fq, axis)
    assert A.f16_fwd_shape_ok(geom, maxpos)
    plan = WeightPlan(torch.device('cuda'))
    Es = plan.linear('e', E, planes='f16')
    plan.run()
    amax = qkv.abs().max().reshape(1).clone()
    O, lse = A.attn_fwd(qkv.view(-1, 192), E, geom, maxpos=maxpos, Es=Es, qkv_amax=amax, need_lse=False)
    assert lse is None
    ref = ref_attention(qkv.double(), E.double(), B, T, Fq, axis, maxpos, 0.25)
    assert relerr(O.view(B, T, Fq, 64), ref) < 5e-6
    if not A.f16_shape_ok(geom, maxpos):
        with pytest.raises(Exception):              # with a backward to come (need_lse) the training shapes only
            A.attn_fwd(qkv.view(-1, 192), E, geom, maxpos=maxpos, Es=Es, qkv_amax=amax)


@pytest.mark.timeout(300)
@pytest.mark.parametrize('B,T,Fq,axis', [(1, 321, 3, 'time'), (3, 5, 101, 'freq')])
def test_bwd4_generic_body_beside_exact_bodies(B, T, Fq, axis, monkeypatch):
    """the waves of one attn_bwd4 workgroup run different body instantiations that must execute the same number of barriers
    (se_attn_bwd4.h, BARRIER CONTRACT): the bench shapes take the exact bodies by default and the generic body with
    SE_ATTN_DBG=64 -- both must finish (a mismatch hangs: this test runs under a timeout) and agree to rounding"""
    from speech_enhancement_amd import attention as A
    from speech_enhancement_amd.weights import WeightPlan
    maxpos = 512
    g = torch.Generator().manual_seed(7 + T)
    qkv = (torch.randn(B, T, Fq, 192, generator=g) * 1.5).cuda()
    E = (torch.randn(2 * maxpos + 1, 16, generator=g) * 0.7).cuda()
    dO = (torch.randn(B, T, Fq, 64, generator=g) * 1e-3).cuda()
    geom = A.seq_geometry(B, T, Fq, axis)
    plan = WeightPlan(torch.device('cuda'))
    Es = plan.linear('e', E, planes='f16')
    plan.run()
    amax = qkv.abs().max().reshape(1).clone()
    do_amax = dO.abs().max().reshape(1).clone()
    O, lse = A.attn_fwd(qkv.view(-1, 192), E, geom, maxpos=maxpos, Es=Es, qkv_amax=amax)
    res = []
    for dbg in (None, '64'):
        if dbg is None:
            monkeypatch.delenv('SE_ATTN_DBG', raising=False)
        else:
            monkeypatch.setenv('SE_ATTN_DBG', dbg)
        dE = torch.zeros_like(E)
        dqkv = A.attn_bwd(qkv.view(-1, 192), E, O, dO.view(-1, 64), lse, geom, dE, maxpos=maxpos, qkv_amax=amax, do_amax=do_amax)
        torch.cuda.synchronize()
        res.append((dqkv.clone(), dE.clone()))
    (d0, e0), (d1, e1) = res
    assert float((d0 - d1).abs().max()) <= 2e-6 * float(d0.abs().max())
    assert float((e0 - e1).abs().max()) <= 2e-6 * float(e0.abs().max())
