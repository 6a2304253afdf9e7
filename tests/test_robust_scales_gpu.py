"""Round 4: the operand scales of the scaled split-fp16 kernels are measured or proven from the CURRENT
parameters -- no static promise about InstanceNorm / LayerNorm gammas or the feed-forward weights is left to break silently:
  * a checkpoint whose InstanceNorm gammas are x64 and whose feed-forward W1 are x32 (the static exponents of round 3 would have
    clamped: 254 |gamma| 2^4 > 65504) still reproduces the oracle;
  * the proven bounds (se_act_bounds) dominate the activations they stand for and the measured maxima equal them;
  * rows of very different magnitude inside ONE operand (per-tensor scale): the per-row error is characterised against fp64 and
    against the fp32-MFMA kernel;
  * the whole step runs with the bf16 linear kernels next to the fp16 attention backward.
Reference semantics: models/generator.py:21-22 (InstanceNorm + PReLU), models/conformer.py:136-142 (feed-forward)."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(__file__), 'golden'))
import formula  # noqa: E402


@pytest.fixture(scope='module')
def S():
    import speech_enhancement_amd as S
    return S


@pytest.fixture(scope='module')
def golden():
    return np.load(os.path.join(os.path.dirname(__file__), 'golden', 'golden_v1.npz'))


def _scaled_state(gamma_x=64.0, w1_x=32.0, ln_x=1.0):
    sd = {k: v.clone() for k, v in formula.formula_state('generator').items()}
    for k in sd:
        inorm = ('.norm' in k and k.endswith('.weight') and ('dense' in k or 'decoder' in k) and 'fn.norm' not in k) or \
            k.endswith(('conv_1.1.weight', 'conv_2.1.weight'))
        if inorm:
            sd[k] = sd[k] * gamma_x
        if k.endswith('fn.fn.net.0.weight'):
            sd[k] = sd[k] * w1_x
        # (the LayerNorms in front of the feed-forward and convolution modules; the attention's own LayerNorm stays: its gain enters
        # the logits squared, and a softmax over logits of 1e5 is ill-conditioned in fp32 whoever computes it)
        if ln_x != 1.0 and (k.endswith(('fn.norm.weight', 'conv.net.0.weight')) and 'attn' not in k):
            sd[k] = sd[k] * ln_x
    return sd


def _spec(golden):
    a = torch.from_numpy(np.asarray(golden['fe_spec_pow']))          # [B, F, T, 2]
    return torch.complex(a[..., 0], a[..., 1])


def _rel_rms(a, b):
    a = a.detach().double().cpu().numpy() if torch.is_tensor(a) else np.asarray(a, np.float64)
    b = b.detach().double().cpu().numpy() if torch.is_tensor(b) else np.asarray(b, np.float64)
    return float(np.sqrt(np.mean((a - b) ** 2)) / (np.sqrt(np.mean(b ** 2)) + 1e-300))


@pytest.mark.parametrize('gamma_x,w1_x,ln_x', [(64.0, 32.0, 1.0), (1.0, 1.0, 200.0), (300.0, 1.0, 1.0)])
def test_large_norm_gains_and_ff_weights_match_the_oracle(S, golden, gamma_x, w1_x, ln_x):
    """forward (train mode: batch statistics) of a checkpoint far outside the static exponents of round 3 vs the fp64 oracle"""
    from oracle import se_oracle as Or
    sd = _scaled_state(gamma_x, w1_x, ln_x)
    g = S.TSCNet(64, 201)
    g.load_state_dict(sd)
    g.cuda().train()
    g.set_dropout(0.0, 0.0)
    spec = _spec(golden)
    with torch.no_grad():
        er, ei = g(spec.cuda())
        sd64 = {k: v.double() for k, v in sd.items()}
        ref = Or.tscnet_forward(sd64, spec.to(torch.complex128), train=True)
        r32 = Or.tscnet_forward(sd, spec, train=True)                  # the same oracle in fp32: the conditioning of this checkpoint
    rr, ri = ref[0], ref[1]
    e = max(_rel_rms(er, rr), _rel_rms(ei, ri))
    floor = max(_rel_rms(r32[0], rr), _rel_rms(r32[1], ri))
    print(f'gamma x{gamma_x} W1 x{w1_x} LN x{ln_x}: relative RMS error of the estimate {e:.2e} (torch-CPU fp32 oracle: {floor:.2e})')
    assert np.isfinite(e) and e < 5e-5 + 2.0 * floor


def test_bounds_dominate_and_measured_maxima_are_exact(S, golden):
    """the proven bounds of one forward (LayerNorm outputs, FF hidden activations, train-mode BatchNorm) against the activations
    recomputed in torch; inorm_prelu_fwd / copy_cols_amax raise exactly the maximum they see"""
    from speech_enhancement_amd import ops as O, layers as LY
    torch.manual_seed(3)
    # measured maxima
    B, P_, C_ = 2, 333, 64
    x = torch.randn(B, P_, C_, device='cuda') * 3
    stats = torch.stack([x.double().sum(1), (x.double() ** 2).sum(1)], -1).contiguous()          # [B, C, 2]
    gam, bet, sl = torch.randn(C_, device='cuda') * 5, torch.randn(C_, device='cuda'), torch.rand(C_, device='cuda') * 1.5
    y = torch.empty_like(x)
    am = torch.zeros(1, device='cuda')
    O.inorm_prelu_fwd(x, C_, 0, stats, gam, bet, sl, y, C_, 0, B, P_, C_, amax=am)
    assert float(am) == float(y.abs().max())
    src = torch.randn(1000, 64, device='cuda') * 7
    dst = torch.zeros(1000, 256, device='cuda')
    am2 = torch.zeros(1, device='cuda')
    O.copy_cols_amax(src, 64, dst, 256, 1000, 64, amax=am2)
    assert torch.equal(dst[:, :64], src) and float(dst[:, 64:].abs().max()) == 0.0 and float(am2) == float(src.abs().max())
    # proven bounds vs the activations of a real forward
    sd = _scaled_state(8.0, 6.0, 3.0)
    g = S.TSCNet(64, 201)
    g.load_state_dict(sd)
    g.cuda().train()
    g.set_dropout(0.0, 0.0)
    spec = _spec(golden)
    with torch.no_grad():
        g(spec.cuda())
    plan = g.__dict__['_wplan']
    assert plan.bounds_ready and len(plan.bounds) == 8 * 7
    P = dict(g.named_parameters())
    for (kind, site), sc in plan.bounds.items():
        b = float(sc)
        assert np.isfinite(b) and b > 0
        if kind == 'ln':
            pre = site + ('.fn.norm' if site.endswith(('ff1', 'ff2')) else ('.norm' if site.endswith('attn') else '.net.0'))
            gmx, bmx = float(P[pre + '.weight'].abs().max()), float(P[pre + '.bias'].abs().max())
            assert abs(b - (63 ** 0.5 * gmx + bmx)) <= 1e-5 * b
        if kind == 'hid':
            W1 = P[site + '.fn.fn.net.0.weight']
            ln = float(plan.bounds[('ln', site)])
            assert abs(b - 2.0 * (ln * float(W1.abs().sum(1).max()) + float(P[site + '.fn.fn.net.0.bias'].abs().max()))) <= 1e-4 * b
    # a LayerNorm output can reach its bound only for a one-hot row; random rows stay below it
    xx = torch.randn(4096, 64, device='cuda') * 50
    xx[0] = 0
    xx[0, 5] = 1e4
    site = 'TSCB_1.time_conformer.ff1'
    ln = torch.nn.functional.layer_norm(xx, (64,), P[site + '.fn.norm.weight'], P[site + '.fn.norm.bias'])
    assert float(ln.abs().max()) <= float(plan.bounds[('ln', site)]) * (1 + 1e-6)
    h = ln @ P[site + '.fn.fn.net.0.weight'].t() + P[site + '.fn.fn.net.0.bias']
    assert float((h * torch.sigmoid(h)).abs().max()) * 1.25 <= float(plan.bounds[('hid', site)])


def _per_row_bits(out, ref, rows_scale):
    """per-row relative error (row maximum norm) -> lost bits against the fp32 unit 2^-24"""
    err = (out.double() - ref).abs().amax(1) / ref.abs().amax(1).clamp_min(1e-300)
    return torch.log2(err.clamp_min(2.0 ** -30) / 2.0 ** -24).cpu().numpy()


def test_rows_of_mixed_magnitude_inside_one_operand(S):
    """dX = dY W with the rows of dY scaled by 2^0 .. 2^-22 inside ONE tensor (one scale per tensor): per-ROW error of the scaled
    split-fp16 row GEMM vs fp64, next to the fp32-MFMA kernel.  hi + lo carry 22 bits while lo is a normal fp16 (rows within
    2^17 of the maximum); below that one bit per binade goes: characterised here, bounded so that a regression shows."""
    from speech_enhancement_amd import gemm as GM, _lib as L
    from speech_enhancement_amd.weights import WeightPlan
    torch.manual_seed(11)
    M, K, N = 23 * 256, 256, 64
    dY = torch.randn(M, K, device='cuda')
    exps = torch.arange(23, device='cuda').repeat_interleave(256)
    dY = dY * torch.exp2(-exps.float())[:, None]
    W = torch.randn(N, K, device='cuda') * 0.1
    ref = dY.double() @ W.double().t()
    plan = WeightPlan(torch.device('cuda'))
    Wp = plan.linear('w', W, planes='f16')
    plan.run()
    am = dY.abs().max().reshape(1).clone()
    out3 = torch.empty(M, N, device='cuda')
    GM.gemm_tap(GM.linear_desc(M, K, N, precision=3, a_amax=am), dY, Wp, out3)
    out0 = torch.empty(M, N, device='cuda')
    GM.gemm_tap(GM.linear_desc(M, K, N, precision=0), dY, W, out0)
    b3 = _per_row_bits(out3, ref, exps).reshape(23, 256).max(1)
    b0 = _per_row_bits(out0, ref, exps).reshape(23, 256).max(1)
    print('lost bits per row group (row scale 2^-k), f16x3:', np.round(b3, 1).tolist())
    print('lost bits per row group (row scale 2^-k), fp32 :', np.round(b0, 1).tolist())
    # rows within 2^16 of the maximum: at the fp32 kernel's level (+- 1.5 bits: 256-term sums); 2^-22: <= 9 bits (15 left)
    assert np.all(b3[:17] <= np.maximum(b0[:17], 0) + 1.5)
    assert np.all(b3[17:] <= 4.8 + (np.arange(17, 23) - 16) * 1.0)       # measured 5.6 .. 10.3: one bit per binade below 2^-16
    # norm-wise the result is fp32-equivalent
    e3, e0 = (float((o.double() - ref).abs().max() / ref.abs().max()) for o in (out3, out0))
    print(f'max error / max |dX|: f16x3 {e3:.2e}, fp32 MFMA {e0:.2e}')
    assert e3 < 1e-6 and e3 < 1.5 * e0 + 1e-7


def test_step_with_bf16_linear_kernels_and_fp16_attention(S, golden):
    """SE_LINEAR_PRECISION=bf16x6 with the default fp16 attention: the qkv input-gradient GEMM takes the six-product planes
    (the plan holds no fp16 planes of Wqkv^T then): one whole train step runs and matches the default mode's losses"""
    import types
    from speech_enhancement_amd import train as TR, optim, gemm as GM

    def step():
        g = S.TSCNet(64, 201)
        g.load_state_dict(formula.formula_state('generator'))
        d = S.Discriminator(16)
        d.load_state_dict(formula.formula_state('discriminator'))
        g.cuda().train()
        g.set_dropout(0.0, 0.0)
        d.cuda().train()
        for m in d.modules():
            if isinstance(m, torch.nn.Dropout):
                m.p = 0.0
        args = types.SimpleNamespace(optimizer='adamw', lr=5e-4, weight_decay=0.01, momentum=0.9, max_norm=0.0)
        og, od = optim.build_optimizer(args, g), optim.build_optimizer(args, d)
        labels = {k: torch.tensor(golden[f'q_{k}']).cuda() for k in ('est', 'clean', 'noisy')}
        out = TR.gan_step(g, d, og, od, torch.tensor(golden['fe_clean']).cuda(), torch.tensor(golden['fe_noisy']).cuda(), 'cmgan',
                          (0.1, 0.9, 0.2, 0.05), labels=labels)
        torch.cuda.synchronize()
        return {k: float(v) for k, v in out.items()}, float(sum(p.double().norm() ** 2 for p in g.parameters()) ** 0.5)

    base, nb = step()
    saved = GM.LINEAR_PRECISION
    GM.LINEAR_PRECISION = 2
    try:
        alt, na = step()
    finally:
        GM.LINEAR_PRECISION = saved
    for k in base:
        assert abs(alt[k] - base[k]) <= 2e-4 * abs(base[k]) + 1e-6, (k, alt[k], base[k])
    assert abs(na - nb) <= 1e-4 * nb
