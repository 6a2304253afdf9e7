"""CDiffuSE (BASELINE config 5): oracle vs the reference's goldens (CPU), product vs goldens / oracle (GPU)."""
import json
import os

import numpy as np
import pytest
import torch

import formula
from oracle import diffuse_oracle as DO

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NOISE_SCHEDULE = np.linspace(1e-4, 0.035, 50).tolist()
FAST = [0.0001, 0.001, 0.01, 0.05, 0.2, 0.35]


def rms(a, b):
    a = np.asarray(a.detach().cpu() if torch.is_tensor(a) else a, np.float64)
    b = np.asarray(b.detach().cpu() if torch.is_tensor(b) else b, np.float64)
    return float(np.sqrt(np.mean((a - b) ** 2)))


@pytest.fixture(scope='module')
def gd():
    return np.load(os.path.join(ROOT, 'tests', 'golden', 'golden_diffuse.npz'))


def diffuse_state():
    """formula weights of the DiffuSE state_dict (names / shapes: tests/golden/diffuse_state_spec.json)"""
    spec = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'diffuse_state_spec.json')))
    sd = {}
    for k, shape in spec:
        base = 1.0 if (len(shape) == 1 and not k.endswith('.bias')) else 0.0        # GroupNorm weights are ones-initialised
        t = formula.formula_tensor('diffuse.' + k, tuple(shape), 'float32', base)
        if k == 'output_projection.weight':
            t = formula.formula_tensor('diffuse.' + k, tuple(shape), 'float32', 0.0) * 0.5
        sd[k] = t
    return sd


def test_oracle_forward_and_upsampler(gd):
    sd = diffuse_state()
    spec, audio = torch.from_numpy(gd['spec_mag']), torch.from_numpy(gd['audio'])
    with torch.no_grad():
        assert rms(DO.stft_mag(torch.from_numpy(gd['noisy'])), gd['spec_mag']) < 1e-6
        assert rms(DO.upsample(sd, spec), gd['upsampled']) < 1e-6 * max(1.0, float(np.abs(gd['upsampled']).max()))
        assert rms(DO.diffusion_embedding(sd, torch.tensor([12.375])), gd['embed_float']) < 1e-6
        y = DO.forward(sd, audio, spec, torch.tensor([7, 31]))
        assert rms(y, gd['fwd_int']) < 2e-5 * float(np.abs(gd['fwd_int']).max())
        y = DO.forward(sd, audio, spec, torch.tensor([12.375]))
        assert rms(y, gd['fwd_float']) < 2e-5 * float(np.abs(gd['fwd_float']).max())


@pytest.mark.parametrize('tag', ['full', 'fast'])
def test_oracle_schedule_and_sampler(gd, tag):
    sched = DO.inference_schedule(NOISE_SCHEDULE, FAST if tag == 'fast' else None)
    for k, v in sched.items():
        np.testing.assert_allclose(v, gd[f'sched_{tag}_{k}'], rtol=1e-6 if k == 'T' else 1e-12, atol=1e-12, err_msg=k)
    if tag == 'full':
        return          # the 50-step trajectory is checked on the GPU only (CPU: 50 x 30-layer forwards)
    with torch.no_grad():
        y = DO.predict(diffuse_state(), gd['noisy'][0], sched, gd[f'predict_{tag}_noise'][:, 0])
    assert rms(y, gd[f'predict_{tag}']) < 1e-4 * max(1e-3, float(np.abs(gd[f'predict_{tag}']).max()))


# ------------------------------------------------------------------------------------------------------------------
# GPU: the HIP path against the reference's goldens
# ------------------------------------------------------------------------------------------------------------------
def _model(S):
    m = S.DiffuSE(10, 100, 201, NOISE_SCHEDULE, 64, 30)
    sd = diffuse_state()
    assert [k for k in m.state_dict()] == [k for k in sd]            # same names, same order as the reference module
    m.load_state_dict(sd)
    return m.cuda().eval()


@pytest.mark.gpu
def test_diffuse_forward_vs_reference(gd):
    import speech_enhancement_amd as S
    m = _model(S)
    spec, audio = torch.from_numpy(gd['spec_mag']).cuda(), torch.from_numpy(gd['audio']).cuda()
    cond = m.conditioner(spec)
    # conditioner = conditioner_projection(upsampler(spec)): check layer 3 against the oracle built on the golden upsampling
    sd = diffuse_state()
    up = torch.from_numpy(gd['upsampled'])
    ref3 = torch.nn.functional.conv1d(up, sd['residual_layers.3.conditioner_projection.weight'],
                                      sd['residual_layers.3.conditioner_projection.bias']).transpose(1, 2)
    assert rms(cond[3], ref3) < 2e-5 * float(ref3.abs().max())
    y = m(audio, spec, torch.tensor([7, 31]))
    assert y.shape == (2, 1, 900)
    assert rms(y, gd['fwd_int']) < 5e-5 * float(np.abs(gd['fwd_int']).max())
    y = m(audio, torch.complex(spec, torch.zeros_like(spec)), torch.tensor([12.375]))      # complex input: abs() (the fix)
    assert rms(y, gd['fwd_float']) < 5e-5 * float(np.abs(gd['fwd_float']).max())


@pytest.mark.gpu
@pytest.mark.parametrize('tag', ['fast', 'full'])
def test_diffuse_reverse_sampler_vs_reference(gd, tag):
    """the reference's own predict() (6-step fast and full 50-step schedule, seeded noise draws) vs predict_diffuse"""
    import types
    import speech_enhancement_amd as S
    m = _model(S)
    cfg = types.SimpleNamespace(NOISE_SCHEDULE=NOISE_SCHEDULE, INFERENCE_NOISE_SCHEDULE=FAST, N_FFT=400, HOP_SAMPLES=100)
    sched = S.inference_schedule(cfg, fast_sampling=(tag == 'fast'))
    for mine, name in zip(sched, ('alpha', 'beta', 'alpha_cum', 'sigmas', 'T', 'c1', 'c2', 'c3', 'delta', 'delta_bar')):
        np.testing.assert_allclose(np.asarray(mine, np.float64), gd[f'sched_{tag}_{name}'], rtol=1e-6, atol=1e-12, err_msg=name)
    y = S.predict_diffuse(m, cfg, gd['noisy'][0], *sched, noises=gd[f'predict_{tag}_noise'])
    ref = gd[f'predict_{tag}']
    assert y.shape == ref.shape
    assert rms(y, ref) < 2e-4 * max(1e-3, float(np.abs(ref).max())), (tag, rms(y, ref), float(np.abs(ref).max()))


@pytest.mark.gpu
def test_diffuse_batch_parts_on_streams_match_serial(gd):
    """predict_diffuse cuts a batch into independent parts on separate HIP streams; clips never interact, so the result must
    equal the serial order (up to the fp64 atomics order of the GroupNorm sums) and clip 0 must still match the reference"""
    import types
    import speech_enhancement_amd as S
    m = _model(S)
    cfg = types.SimpleNamespace(NOISE_SCHEDULE=NOISE_SCHEDULE, INFERENCE_NOISE_SCHEDULE=FAST, N_FFT=400, HOP_SAMPLES=100)
    sched = S.inference_schedule(cfg, fast_sampling=True)
    rng = np.random.default_rng(5)
    clip = gd['noisy'][0].reshape(-1)
    nz0 = gd['predict_fast_noise'].reshape(len(FAST) - 1, 1, -1)
    x = np.stack([clip] + [clip * s + 0.01 * rng.standard_normal(clip.shape).astype(np.float32) for s in np.linspace(0.3, 1.2, 8)])
    nz = np.concatenate([nz0, rng.standard_normal((nz0.shape[0], 8, nz0.shape[2])).astype(np.float32)], 1)
    y1 = S.predict_diffuse(m, cfg, x, *sched, noises=nz, streams=1)
    y2 = S.predict_diffuse(m, cfg, x, *sched, noises=nz, streams=2)
    assert y1.shape == y2.shape == (9, nz.shape[2])
    assert np.abs(y1 - y2).max() < 1e-5 * max(1e-3, np.abs(y1).max())
    ref = gd['predict_fast'].reshape(-1)
    assert rms(y2[0], ref) < 2e-4 * max(1e-3, float(np.abs(ref).max()))


@pytest.mark.gpu
def test_diffuse_bench_size_parts_on_streams_match_serial():
    """BASELINE config 5 at its benchmark size -- batch 32 x 2 s clips (15.8 GB conditioner cache), the 6-step fast schedule --:
    the batch cut into 2 and 3 independent parts on separate HIP streams returns what the serial order returns (clips never
    interact: GroupNorm is per sample), clip by clip, and permuting the clips permutes the outputs (no cross-clip leakage)."""
    import types
    import speech_enhancement_amd as S
    m = _model(S)
    cfg = types.SimpleNamespace(NOISE_SCHEDULE=NOISE_SCHEDULE, INFERENCE_NOISE_SCHEDULE=FAST, N_FFT=400, HOP_SAMPLES=100)
    sched = S.inference_schedule(cfg, fast_sampling=True)
    rng = np.random.default_rng(7)
    B, Ls = 32, 32000
    x = (0.1 * rng.standard_normal((B, Ls))).astype(np.float32) * np.linspace(0.3, 1.5, B, dtype=np.float32)[:, None]
    nz = rng.standard_normal((len(FAST) - 1, B, Ls + 100)).astype(np.float32)
    y1 = S.predict_diffuse(m, cfg, x, *sched, noises=nz, streams=1)           # noise draws: [steps - 1, B, 100 T], T = Ls / 100 + 1
    assert y1.shape[0] == B and np.isfinite(y1).all()
    scale = max(1e-3, float(np.abs(y1).max()))
    for ns in (2, 3):
        yn = S.predict_diffuse(m, cfg, x, *sched, noises=nz, streams=ns)
        assert np.abs(yn - y1).max() < 1e-5 * scale, (ns, float(np.abs(yn - y1).max()))
    perm = rng.permutation(B)
    yp = S.predict_diffuse(m, cfg, x[perm], *sched, noises=nz[:, perm], streams=2)
    assert np.abs(yp - y1[perm]).max() < 1e-5 * scale


@pytest.mark.gpu
@pytest.mark.parametrize('B,Lp', [(1, 128), (3, 1000), (2, 4133)])
def test_gate_prologue_projection_vs_fp64(B, Lp):
    """SE_PRO_GATE (csrc/se_gemm.hip, conv1d_k64_wstat_kernel<1, true>): the residual block's gate built in the prologue of its 1 x 1
    projection -- R2 = W2 (sigmoid(GN(R)[:C] + cond[:C]) * tanh(GN(R)[C:] + cond[C:])) + b2 and the GroupNorm sums of R2 -- against
    fp64 torch and against the stand-alone gate kernel + the same GEMM; one tile, ragged last tiles, several batch entries"""
    import ctypes as C
    from speech_enhancement_amd import _lib as L, gemm as GM
    from speech_enhancement_amd.weights import WeightPlan
    dev = torch.device('cuda')
    torch.manual_seed(B * 7 + Lp)
    Cc = 64
    R = torch.randn(B, Lp, 2 * Cc, device=dev) * 1.5
    cond = torch.randn(B, Lp, 2 * Cc, device=dev)
    ss = torch.stack([torch.rand(B, 2 * Cc, device=dev) + 0.5, torch.randn(B, 2 * Cc, device=dev) * 0.3], -1).contiguous()   # (scale, shift)
    W2, b2 = torch.randn(2 * Cc, Cc, device=dev) * 0.2, torch.randn(2 * Cc, device=dev) * 0.1
    plan = WeightPlan(dev)
    w16 = plan.linear('w2', W2, planes='f16')
    plan.run()
    z = R.double() * ss[..., 0].double()[:, None, :] + ss[..., 1].double()[:, None, :] + cond.double()
    y2 = torch.sigmoid(z[..., :Cc]) * torch.tanh(z[..., Cc:])
    ref = y2 @ W2.double().t() + b2.double()
    out, st = torch.empty(B, Lp, 2 * Cc, device=dev), torch.zeros(B, 2 * Cc, 2, device=dev, dtype=torch.float64)
    d = GM.make_desc(B, 1, Lp, 1, Lp, [(0, 0)], Cc, 2 * Cc, 2 * Cc, 2 * Cc, ldw=Cc, ldx=2 * Cc, prologue=L.PRO_GATE,
                     epilogue=L.EPI_BIAS | L.EPI_STATS, precision=3, a_sexp=13)
    GM.gemm_tap(d, R, w16, out, bias=b2, AUX=cond, ps=ss, stats=st)
    scale = float(ref.abs().max())
    assert float((out.double() - ref).abs().max()) < 2e-6 * scale
    assert float((st[..., 0] - ref.sum(1)).abs().max()) < 1e-6 * scale * Lp
    assert float((st[..., 1] - (ref ** 2).sum(1)).abs().max()) < 2e-6 * scale * scale * Lp
    # the two-launch path (gate kernel + the same projection kernel without prologue)
    y2k = torch.empty(B, Lp, Cc, device=dev)
    L.call('se_diff_gate', L.ptr(R), L.ptr(ss), L.ptr(cond), L.ptr(y2k), C.c_int(B), C.c_long(Lp), C.c_int(Cc), L.stream())
    out2, st2 = torch.empty_like(out), torch.zeros_like(st)
    d2 = GM.make_desc(B, 1, Lp, 1, Lp, [(0, 0)], Cc, Cc, 2 * Cc, 2 * Cc, epilogue=L.EPI_BIAS | L.EPI_STATS, precision=3, a_sexp=13)
    GM.gemm_tap(d2, y2k, w16, out2, bias=b2, stats=st2)
    assert float((out - out2).abs().max()) < 2e-6 * scale
