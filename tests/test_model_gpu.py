"""GPU parity of the whole hot path (front-end, generator fwd/bwd, discriminator, train step) against the
committed golden vectors produced by the reference, and against the oracle on the same seeded inputs."""
import numpy as np
import pytest
import torch

import formula

pytestmark = pytest.mark.gpu


def rms(a, b):
    a = np.asarray(a.detach().cpu() if torch.is_tensor(a) else a, np.float64)
    b = np.asarray(b.detach().cpu() if torch.is_tensor(b) else b, np.float64)
    return float(np.sqrt(np.mean((a - b) ** 2)))


def t(a):
    return torch.from_numpy(np.asarray(a)).cuda()


def cplx(a):
    a = t(a)
    return torch.complex(a[..., 0], a[..., 1])


@pytest.fixture(scope='module')
def S():
    import speech_enhancement_amd as S
    return S


def load_g(S, train=True):
    g = S.TSCNet(64, 201)
    g.load_state_dict(formula.formula_state('generator'))
    g.cuda()
    g.train(train)
    g.set_dropout(0.0, 0.0)          # the fixtures were generated with every nn.Dropout at p = 0
    return g


def load_d(S, train=True):
    d = S.Discriminator(16)
    d.load_state_dict(formula.formula_state('discriminator'))
    d.cuda()
    for m in d.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    d.train(train)
    return d


def test_frontend(S, golden):
    from speech_enhancement_amd import frontend as FE, ops as O
    noisy = t(golden['fe_noisy'])
    c = O.clip_scale(noisy)
    for comp in ('pow', 'log', 'norm', 'none'):
        planes, xp = FE.stft_planes(noisy, 400, 100, comp, scale=c)
        spec = torch.view_as_real(FE.planes_to_spec(planes))
        ref = golden[f'fe_spec_{comp}']
        assert rms(spec, ref) < 3e-5 * max(1.0, float(np.abs(ref).max())), comp
        y = S.uncompressed_istft(cplx(ref), 400, 100, None, comp_type=comp)
        assert rms(y, golden[f'fe_istft_{comp}']) < 3e-5, comp
    assert rms(xp[:, 200:-200], golden['fe_noisy_n']) < 1e-6
    s = S.compressed_stft(t(golden['fe_noisy_n']), 400, 100, None)
    assert rms(torch.view_as_real(s), golden['fe_spec_pow']) < 3e-5


@pytest.mark.parametrize('tag', ['t17', 'f101'])
def test_conformer_block(S, golden, tag):
    """one Conformer block vs the reference's (golden): outputs, input gradient, rel-pos / depthwise gradients,
    BatchNorm running statistics."""
    from speech_enhancement_amd import layers as LY
    g = load_g(S)
    p = 'TSCB_1.time_conformer'
    P = {k: v for k, v in g.named_parameters()}
    P.update(dict(g.named_buffers()))
    x = t(golden[f'cf_{tag}_x'])                       # [Bs, n, 64] -> B=Bs, T=n, F'=1, time axis
    Bs, n, _ = x.shape
    with torch.no_grad():
        tok = x.reshape(Bs * n, 64).contiguous()
        y, ctx = LY.conformer_fwd(P, p, tok, Bs, n, 1, 'time', True, LY.NO_DP, dict(g.named_buffers()))
        y = y - tok                                    # the kernel path folds the TSCB residual in
        assert rms(y.view(Bs, n, 64), golden[f'cf_{tag}_y_train']) < 3e-5
        assert rms(P[f'{p}.conv.net.5.running_mean'], golden[f'cf_{tag}_rm']) < 1e-6
        assert rms(P[f'{p}.conv.net.5.running_var'], golden[f'cf_{tag}_rv']) < 1e-6
        G = {k: torch.zeros_like(v) for k, v in g.named_parameters()}
        dy = torch.cos(torch.arange(y.numel(), device='cuda').view(Bs, n, 64) * 0.01).reshape(Bs * n, 64)
        dx = LY.conformer_bwd(P, G, p, ctx, dy.contiguous(), Bs, n, 1) - dy
        for got, key in ((dx.view(Bs, n, 64), 'dx'), (G[f'{p}.attn.fn.rel_pos_emb.weight'], 'dE'),
                         (G[f'{p}.conv.net.4.conv.weight'], 'dWdw')):
            ref = golden[f'cf_{tag}_{key}']
            assert rms(got, ref) < 2e-4 * float(np.abs(ref).max()), key
        g.load_state_dict(formula.formula_state('generator'))
        ye, _ = LY.conformer_fwd(P, p, tok, Bs, n, 1, 'time', False)
        assert rms((ye - tok).view(Bs, n, 64), golden[f'cf_{tag}_y_eval']) < 3e-5


def test_tscnet_forward_backward(S, golden):
    g = load_g(S)
    spec = cplx(golden['fe_spec_pow'])
    er, ei = g(spec)
    assert rms(er, golden['g_real']) < 2e-4 and rms(ei, golden['g_imag']) < 2e-4
    wr = torch.cos(torch.arange(er.numel(), device='cuda').view_as(er) * 0.013)
    wi = torch.sin(torch.arange(ei.numel(), device='cuda').view_as(ei) * 0.017)
    (er * wr + ei * wi).sum().backward()
    names = [k for k, _ in g.named_parameters()]
    gn = np.array([float(p.grad.norm()) for _, p in g.named_parameters()])
    # Tolerances are conditioning-aware: the reference's own fp32 run (g_*) differs from its fp64 run (g64_*) by up
    # to 8 % on 18 encoder / first-Conformer gradients (fp32 rounding, not an algorithmic difference: both are the reference); a correct fp32
    # implementation may land on either side, so that deviation (x1.5) is added to the base tolerance.
    ref, ref32 = golden['g64_gradnorm'], golden['g_gradnorm']
    tol = 5e-3 * ref + 5e-5 * ref.max() + 1.5 * np.abs(ref32 - ref)
    bad = [(names[i], gn[i], ref[i]) for i in range(len(names)) if abs(gn[i] - ref[i]) > tol[i]]
    assert not bad, bad[:10]
    grads = dict((k, p.grad) for k, p in g.named_parameters())
    for k in golden.files:
        if k.startswith('g64_grad:'):
            r = golden[k]
            k32 = 'g_grad:' + k[9:]
            slack = 1.5 * rms(golden[k32], r) if k32 in golden.files else 0.0
            assert rms(grads[k[9:]], r) < 2e-3 * float(np.abs(r).max()) + slack, k
    g.load_state_dict(formula.formula_state('generator'))      # the train-mode pass advanced the BN running stats
    g.eval()
    with torch.no_grad():
        er, ei = g(spec)
    assert rms(er, golden['g_real_eval']) < 2e-4 and rms(ei, golden['g_imag_eval']) < 2e-4


def test_discriminator(S, golden):
    d = load_d(S)
    cm = t(golden['d_in_clean_mag'])
    nm = t(golden['d_in_noisy_mag']).requires_grad_(True)
    y = d(cm, nm)
    assert rms(y, golden['d_out_train']) < 1e-5
    (y.flatten() * torch.tensor([1.0, -2.0], device='cuda')).sum().backward()
    assert rms(nm.grad, golden['d64_dnoisy']) < 1e-3 * float(np.abs(golden['d64_dnoisy']).max())
    gn = np.array([float(p.grad.norm()) for _, p in d.named_parameters()])
    np.testing.assert_allclose(gn, golden['d64_gradnorm'], rtol=5e-3, atol=2e-5 * float(gn.max()))
    sd = d.state_dict()
    for li in (0, 3, 6, 9, 14, 17):
        assert rms(sd[f'layers.{li}.weight_u'], golden[f'd_u{li}']) < 1e-5
        assert rms(sd[f'layers.{li}.weight_v'], golden[f'd_v{li}']) < 1e-5
    d2 = load_d(S, train=False)
    with torch.no_grad():
        assert rms(d2(cm, nm.detach()), golden['d_out_eval']) < 1e-5


@pytest.mark.parametrize('arch,weights,optname', [('cmgan', (0.1, 0.9, 0.2, 0.05), 'sgd'),
                                                  ('cmgan', (0.1, 0.9, 0.2, 0.05), 'adamw'),
                                                  ('scp', (0.3, 0.7, 0.2, 0.05), 'sgd')])
def test_train_step_vs_reference_loop(S, golden, arch, weights, optname):
    """one train_gan iteration of the reference (golden, fp32 and fp64 runs) vs gan_step on the GPU."""
    import types
    from speech_enhancement_amd import train as TR, optim
    from oracle import se_oracle as Or
    g, d = load_g(S), load_d(S)
    base_lr = 0.01 if optname == 'sgd' else 5e-4
    args = types.SimpleNamespace(optimizer=optname, lr=base_lr, weight_decay=0.01, momentum=0.9, max_norm=0.0)
    og, od = optim.build_optimizer(args, g), optim.build_optimizer(args, d, lr=base_lr * 2)
    lr = Or.lr_at(10.0, base_lr, 100)
    for o in (og, od):
        for grp in o.param_groups:
            grp['lr'] = lr
    labels = {k: t(golden[f'q_{k}']) for k in ('est', 'clean', 'noisy')}
    out = TR.gan_step(g, d, og, od, t(golden['fe_clean']), t(golden['fe_noisy']), arch, weights, labels=labels)
    pre = f'step_{arch}_{optname}_f64_'
    mse = golden[pre + 'mse_calls']
    assert abs(float(out['loss_mag']) - mse[0]) < 2e-4 * mse[0]
    assert abs(float(out['loss_ri']) - (mse[1] + mse[2])) < 2e-4 * (mse[1] + mse[2])
    assert abs(float(out['gan']) - mse[3]) < 2e-4 * mse[3] + 1e-6
    assert abs(float(out['L_E']) - mse[4]) < 1e-3 * mse[4] + 1e-6
    assert abs(float(out['L_C']) - mse[5]) < 1e-3 * mse[5] + 1e-6
    assert abs(float(out['loss_g']) - float(golden[pre + 'gen_loss'])) < 2e-4 * abs(float(out['loss_g']))
    if arch == 'scp':
        return      # the fp32 consistency-path gradient is ill-conditioned; losses only
    gs, ds = g.state_dict(), d.state_dict()
    gnorm = np.array([float(v.double().norm()) for v in gs.values()])
    dnorm = np.array([float(v.double().norm()) for v in ds.values()])
    ref_g, ref_d = golden[pre + 'g_norm'], golden[pre + 'd_norm']
    if optname == 'sgd':
        np.testing.assert_allclose(gnorm, ref_g, rtol=3e-4, atol=1e-5)
        np.testing.assert_allclose(dnorm, ref_d, rtol=3e-4, atol=1e-5)
        for k in golden.files:
            if k.startswith(pre + 'g:') or k.startswith(pre + 'd:'):
                name = k.split(':', 1)[1]
                new = gs[name] if k.startswith(pre + 'g:') else ds[name]
                src = formula.formula_state('generator' if k.startswith(pre + 'g:') else 'discriminator')[name]
                if name.endswith(('_u', '_v')):
                    assert rms(new, golden[k]) < 1e-5, k
                    continue
                upd_ref = golden[k].astype(np.float64) - src.double().numpy()
                upd = new.double().cpu().numpy() - src.double().numpy()
                assert rms(upd, upd_ref) < 1e-2 * np.sqrt(np.mean(upd_ref ** 2)) + 1e-9, k
    else:
        numel_g = np.array([v.numel() for v in gs.values()])
        assert np.all(np.abs(gnorm - ref_g) < 3e-4 * ref_g + 2 * lr * np.sqrt(0.05 * numel_g) + 2 * lr)


@pytest.mark.parametrize('conv_precision', ['f32', 'bf16x3', 'bf16x6', 'f16x3'])
def test_full_size_enhanced_magnitude(S, golden, conv_precision):
    """north_star parity bar: RMS(|est| - |est_ref|) <= 1e-3 on the compressed enhanced magnitude of a 2 s clip, in
    every arithmetic mode of the convolution GEMMs (f16x3 = the default; the token-wise GEMMs stay in their default mode)."""
    from speech_enhancement_amd import frontend as FE, ops as O, layers as LY
    saved = (LY.CONV_PRECISION, LY.WGRAD_PRECISION[0])
    names = {v: k for k, v in LY._PREC.items()}
    LY.set_conv_precision(conv_precision)
    try:
        g = load_g(S)
        noisy = t(golden['full_noisy'])
        planes, _ = FE.stft_planes(noisy, 400, 100, 'pow', scale=O.clip_scale(noisy))
        with torch.no_grad():
            est = g.forward_planes(planes)
            audio = FE.istft_planes(est, 400, 100, 'pow')
    finally:
        LY.set_conv_precision(names[saved[0]], names[saved[1]])       # the session default, not a hard-coded mode
    ref = golden['full_est_mag']
    e = rms(est[0, :, :, 0], ref)
    print(conv_precision, 'enhanced-magnitude RMS error', e, 'reference RMS', float(np.sqrt(np.mean(ref.astype(np.float64) ** 2))))
    assert e < 1e-3
    if conv_precision in ('bf16x6', 'f16x3', 'f32'):
        assert e < 5e-5          # the fp32-equivalent modes sit two orders of magnitude inside the bar (measured 1.4e-5)
    assert rms(audio, golden['full_est_audio']) < 1e-3


def test_predict_and_load_model_vs_oracle(S, tmp_path):
    """inference_gan.load_model / predict (wrap-pad to a hop multiple, eval-mode BatchNorm) vs the oracle."""
    import types
    from oracle import se_oracle as Or
    gsd = formula.formula_state('generator')
    ck = {'gen_state_dict': {'module.' + k: v for k, v in gsd.items()}}
    path = str(tmp_path / 'ck.pth.tar')
    torch.save(ck, path)
    cfg = types.SimpleNamespace(N_FFT=400, HOP_SAMPLES=100)
    model = S.load_model(path, cfg, torch.device('cuda'))
    assert not model.training
    rs = np.random.RandomState(3)
    x = (0.1 * rs.randn(1657)).astype(np.float32)          # not a multiple of the hop
    y = S.predict(model, cfg, x, torch.device('cuda'))
    assert y.shape == x.shape
    xt = torch.from_numpy(x)[None]
    c = torch.sqrt(xt.shape[-1] / torch.sum(xt ** 2, -1))
    xn = xt * c[:, None]
    xn = torch.cat([xn, xn[:, :1700 - 1657]], -1)
    with torch.no_grad():
        er, ei = Or.tscnet_forward(gsd, Or.compressed_stft(xn), False)
        ref = Or.uncompressed_istft(torch.complex(er, ei).squeeze(1).permute(0, 2, 1)) / c[:, None]
    assert rms(y, ref.flatten()[:1657].numpy()) < 2e-4 * float(ref.abs().max()) + 1e-5


def test_validate_gan_and_synthetic_training_entry(S, tmp_path):
    """validate_gan vs the oracle losses; main_gan worker runs one synthetic epoch and writes a reference-format
    checkpoint."""
    import types
    from oracle import se_oracle as Or
    from speech_enhancement_amd import main_gan
    g, d = load_g(S), load_d(S)
    gsd, dsd = formula.formula_state('generator'), formula.formula_state('discriminator')
    rs = np.random.RandomState(5)
    clean = torch.from_numpy((0.1 * rs.randn(2, 1600)).astype(np.float32))
    noisy = clean + torch.from_numpy((0.05 * rs.randn(2, 1600)).astype(np.float32))
    q = torch.tensor([0.4, 0.7])
    cfg = S.get_config(types.SimpleNamespace(cfg=None))
    args = types.SimpleNamespace(gpu=0)
    batch = {'audio': clean.cuda(), 'noisy': noisy.cuda(), 'labels': {'est': q.cuda()}}
    vg, vd = S.validate_gan([batch], g, d, None, None, 0, args, cfg)
    cn, nn_, _ = Or.normalize_pair(clean, noisy)
    with torch.no_grad():
        r = Or.generator_losses(gsd, dsd, cn, nn_, 'cmgan', train=False)
        loss = 0.1 * r['loss_ri'] + 0.9 * r['loss_mag'] + 0.2 * r['time_loss'] + 0.05 * r['gan']
        d_gx = Or.discriminator_forward(dsd, r['clean_mag'], r['est_mag'], False)
        d_yy = Or.discriminator_forward(dsd, r['clean_mag'], r['clean_mag'], False)
        ld = ((d_yy.flatten() - 1) ** 2).mean() + ((d_gx.flatten() - q) ** 2).mean()
    assert abs(vg - float(loss)) < 2e-4 * float(loss) and abs(vd - float(ld)) < 1e-3 * float(ld) + 1e-6
    out = str(tmp_path / 'out')
    main_gan.main(['--cfg', '/dev/null', '-a', 'cmgan', '-b', '2', '--epochs', '8', '--start-epoch', '7', '--optimizer', 'adamw', '--lr', '5e-4',
                   '--crop-len', '1', '--synthetic', '2', '--output', out, '--gpu', '0', '-p', '1000'])
    ck = torch.load(str(tmp_path / 'out' / 'cmgan' / 'default' / 'checkpoint_0007.pth.tar'), map_location='cpu')
    assert set(ck) == {'epoch', 'arch', 'gen_state_dict', 'disc_state_dict', 'optimizer', 'optimizer_disc', 'best_loss'}
    assert len(ck['gen_state_dict']) == 359 and len(ck['disc_state_dict']) == 34
    assert all(torch.isfinite(v).all() for v in ck['gen_state_dict'].values() if v.is_floating_point())


def test_scp_discriminator_step_matches_reference(S, golden):
    """scp: the self-correcting discriminator update (three gradients, dot products, piecewise weights, the
    reference's 2x) is well conditioned even though the generator's consistency gradient is not: compare every
    discriminator tensor after the SGD step with the reference loop (fp64 golden)."""
    import types
    from speech_enhancement_amd import train as TR, optim
    from oracle import se_oracle as Or
    g, d = load_g(S), load_d(S)
    args = types.SimpleNamespace(optimizer='sgd', lr=0.01, weight_decay=0.01, momentum=0.9, max_norm=0.0)
    og, od = optim.build_optimizer(args, g), optim.build_optimizer(args, d, lr=0.02)
    lr = Or.lr_at(10.0, 0.01, 100)
    for o in (og, od):
        for grp in o.param_groups:
            grp['lr'] = lr
    labels = {k: t(golden[f'q_{k}']) for k in ('est', 'clean', 'noisy')}
    out = TR.gan_step(g, d, og, od, t(golden['fe_clean']), t(golden['fe_noisy']), 'scp', (0.3, 0.7, 0.2, 0.05), labels=labels)
    mse = golden['step_scp_sgd_f64_mse_calls']
    assert abs(float(out['L_N']) - mse[6]) < 1e-3 * mse[6] + 1e-6
    ds = d.state_dict()
    dnorm = np.array([float(v.double().norm()) for v in ds.values()])
    np.testing.assert_allclose(dnorm, golden['step_scp_sgd_f64_d_norm'], rtol=1e-3, atol=1e-5)
    k = 'step_scp_sgd_f64_d:layers.17.weight_orig'
    src = formula.formula_state('discriminator')['layers.17.weight_orig']
    upd_ref = golden[k].astype(np.float64) - src.double().numpy()
    upd = ds['layers.17.weight_orig'].double().cpu().numpy() - src.double().numpy()
    assert rms(upd, upd_ref) < 2e-2 * np.sqrt(np.mean(upd_ref ** 2)) + 1e-9


def test_graphed_inference_matches_eager(S):
    """the HIP-graph replay of the batch-1 enhancement pipeline returns what the eager predict() returns; lengths that pad
    to the same number of frames share one graph (length bucket), others get their own"""
    import types
    from speech_enhancement_amd import inference as INF
    torch.manual_seed(0)
    g = S.TSCNet(64, 201)
    g.apply(S.kaiming_init)
    g.cuda().eval()
    cfg = types.SimpleNamespace(N_FFT=400, HOP_SAMPLES=100)
    enh = INF.GraphedEnhancer(g, cfg, 4850)
    for seed, length in ((1, 4850), (2, 4850), (3, 4801), (4, 4900), (5, 3333)):
        x = (0.1 * np.random.RandomState(seed).randn(length)).astype(np.float32)
        ref = INF.predict(g, cfg, x)
        out = enh(x)
        assert out.shape == ref.shape and np.isfinite(out).all()
        assert np.abs(out - ref).max() <= 2e-6 * max(1.0, np.abs(ref).max()), (length, np.abs(out - ref).max())
    assert sorted(enh.buckets) == [34, 49]          # 4801..4900 samples -> 49 frames, 3333 -> 34


def test_ten_second_utterance_graphed_vs_eager_vs_oracle_prefix(S):
    """BASELINE config 4: batch-1 enhancement of a 10 s utterance (160 000 samples, T = 1601: the n > 512 relative-position
    clamp and the streaming attention kernels) through the graph bucket and through eager predict(); and a short clip vs
    the CPU oracle as the absolute anchor of the same code path"""
    import time
    import types
    from oracle import se_oracle as Or
    from speech_enhancement_amd import inference as INF
    gsd = formula.formula_state('generator')
    g = S.TSCNet(64, 201)
    g.load_state_dict(gsd)
    g.cuda().eval()
    cfg = types.SimpleNamespace(N_FFT=400, HOP_SAMPLES=100)
    x = (0.1 * np.random.RandomState(9).randn(160000)).astype(np.float32)
    ref = INF.predict(g, cfg, x)
    enh = INF.GraphedEnhancer(g, cfg)
    out = enh(x)
    assert out.shape == (160000,) and np.isfinite(out).all()
    assert np.abs(out - ref).max() <= 2e-6 * max(1.0, np.abs(ref).max())
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(3):
        enh(x)
    tg = (time.time() - t0) / 3
    t0 = time.time()
    for _ in range(3):
        INF.predict(g, cfg, x)
    te = (time.time() - t0) / 3
    print(f'10 s utterance: graph replay {tg * 1e3:.1f} ms, eager {te * 1e3:.1f} ms per utterance (real-time factor {tg / 10:.5f})')
    # absolute anchor: a 0.6 s clip of the same signal against the oracle (whole-utterance statistics: no prefix property)
    xs = x[:9650]
    ys = enh(xs)
    xt = torch.from_numpy(xs)[None]
    c = torch.sqrt(xt.shape[-1] / torch.sum(xt ** 2, -1))
    xn = torch.cat([xt * c[:, None], (xt * c[:, None])[:, :9700 - 9650]], -1)
    with torch.no_grad():
        er, ei = Or.tscnet_forward(gsd, Or.compressed_stft(xn), False)
        yo = Or.uncompressed_istft(torch.complex(er, ei).squeeze(1).permute(0, 2, 1)) / c[:, None]
    assert rms(ys, yo.flatten()[:9650].numpy()) < 2e-4 * float(yo.abs().max()) + 1e-5


def test_stale_label_option_hides_the_provider(S):
    """opt-in one-step-stale PESQ labels: the first call makes no discriminator update, the second
    one updates the discriminator with the FIRST batch and its labels and does NOT wait for the provider call of its own
    batch (ordering check: that call is held open until the second step has returned -- a step that waited for it would
    hang on the gate, which then times out and fails the test).  Also the staging-buffer hazard of this schedule: the
    provider call of batch 1 reads its host arrays only AFTER step 2 has submitted batch 2; it must still see batch 1
    (double-buffered pinned buffers, train.PesqSideChannel._claim_set)."""
    import threading
    import time
    import types
    from speech_enhancement_amd import train as TR, optim as OP
    calls, released, seen_late = [], [], []
    gate = threading.Event()
    side = TR.pesq_side_channel()

    def provider(clean_list, deg_list):
        k = len(calls)
        calls.append(float(np.abs(np.asarray(clean_list[0])).sum()))          # fingerprint of the batch it was given
        if k == 0:
            t_end = time.time() + 20.0
            while side.calls < base_calls + 2 and time.time() < t_end:        # until step 2 has submitted batch 2 ...
                time.sleep(0.005)
            time.sleep(0.1)                                                   # ... and its D2H copies have landed
            seen_late.append(float(np.abs(np.asarray(clean_list[0])).sum()))  # read the staging buffer LATE
        else:
            released.append(gate.wait(20.0))                                  # held open until step 2 has returned
        return torch.tensor([0.5] * len(clean_list), dtype=torch.float32)
    TR.set_pesq_provider(provider)
    try:
        g, d = load_g(S), load_d(S)
        oa = types.SimpleNamespace(optimizer='sgd', lr=1e-3, weight_decay=0.0, momentum=0.9, max_norm=0.0)
        og, od = OP.build_optimizer(oa, g), OP.build_optimizer(oa, d)
        torch.manual_seed(0)
        b1 = 0.1 * torch.randn(2, 1600, device='cuda')
        b2 = 0.1 * torch.randn(2, 1600, device='cuda')
        base_calls = side.calls
        d0 = torch.cat([p.detach().flatten() for p in d.parameters()]).clone()
        out1 = TR.gan_step(g, d, og, od, b1, b1 + 0.01, 'cmgan', (0.1, 0.9, 0.2, 0.05), stale_labels=True)
        assert float(out1['loss_d']) == 0.0
        assert torch.equal(d0, torch.cat([p.detach().flatten() for p in d.parameters()]))
        out2 = TR.gan_step(g, d, og, od, b2, b2 + 0.01, 'cmgan', (0.1, 0.9, 0.2, 0.05), stale_labels=True)
        torch.cuda.synchronize()
        gate.set()                                       # step 2 is back: only now may batch 2's provider call finish
        assert float(out2['loss_d']) > 0.0 and not torch.equal(d0, torch.cat([p.detach().flatten() for p in d.parameters()]))
        for f in side.last_use:
            if f is not None:
                f.result(timeout=30)
        assert len(calls) == 2 and calls[0] != calls[1]
        assert released == [True]                        # the gate was opened by the main thread, not by its timeout
        assert seen_late == [calls[0]]                   # batch 1's staging buffer was not overwritten by batch 2's copy
    finally:
        gate.set()
        TR.set_pesq_provider(None)


@pytest.mark.parametrize('arch', ['cmgan', 'scp'])
def test_pesq_side_channel_matches_supplied_labels(S, arch):
    """labels=None routes the PESQ provider through the asynchronous side channel (pinned D2H on a side stream + worker
    thread): same losses as supplying the provider's labels directly, and the provider's latency is hidden behind the
    generator backward."""
    import time
    import types
    from speech_enhancement_amd import train as TR, optim as OP
    calls = []

    def provider(clean_list, deg_list):          # deterministic stand-in for pesq: a function of the audio only
        calls.append(len(clean_list))
        time.sleep(0.05)
        q = [float(np.tanh(np.mean(np.abs(np.asarray(c) - np.asarray(dg))) * 5.0)) for c, dg in zip(clean_list, deg_list)]
        return torch.tensor(q, dtype=torch.float32).cuda()
    TR.set_pesq_provider(provider)
    try:
        torch.manual_seed(0)
        B, Ls = 4, 3200
        clean = 0.1 * torch.randn(B, Ls, device='cuda')
        noisy = clean + 0.05 * torch.randn(B, Ls, device='cuda')
        w = (0.1, 0.9, 0.2, 0.05) if arch == 'cmgan' else (0.3, 0.7, 0.2, 0.05)
        outs = []
        for use_side in (True, False):
            torch.manual_seed(1)
            g, d = S.TSCNet(64, 201), S.Discriminator(16)
            g.apply(S.kaiming_init); d.apply(S.kaiming_init)
            g.cuda().train(); d.cuda().train()
            g.set_dropout(0.0, 0.0)
            for m in d.modules():
                if isinstance(m, torch.nn.Dropout):
                    m.p = 0.0
            oa = types.SimpleNamespace(optimizer='sgd', lr=1e-3, weight_decay=0.0, momentum=0.9, max_norm=0.0)
            og, od = OP.build_optimizer(oa, g), OP.build_optimizer(oa, d)
            labels = None
            if not use_side:                      # what the side channel must have produced, computed synchronously
                with torch.no_grad():
                    from speech_enhancement_amd import frontend as FE, ops as O
                    c = O.clip_scale(noisy.contiguous())
                    npl, npad = FE.stft_planes(noisy, 400, 100, 'pow', scale=c)
                    cpl, cpad = FE.stft_planes(clean, 400, 100, 'pow', scale=c)
                    est_audio = FE.istft_planes(g.forward_planes(npl), 400, 100, 'pow')
                    Lh = est_audio.size(-1)
                    cn = list(cpad[:, 200:200 + Ls][:, :Lh].cpu().numpy())
                    labels = {'est': provider(cn, list(est_audio.cpu().numpy())), 'clean': provider(cn, cn),
                              'noisy': provider(cn, list(npad[:, 200:200 + Ls][:, :Lh].cpu().numpy()))}
                g.load_state_dict(g.state_dict())
            out = TR.gan_step(g, d, og, od, clean, noisy, arch, w, labels=labels)
            outs.append({k: float(v) for k, v in out.items()})
        for k in outs[0]:
            assert abs(outs[0][k] - outs[1][k]) <= 2e-5 * max(1.0, abs(outs[1][k])), (k, outs[0][k], outs[1][k])
        assert calls[0] == B
    finally:
        TR.set_pesq_provider(None)


# ---------------------------------------------------------------------------------------------------------
# round 2: validate_gan (scp / --gen-first), --max-norm, the front-end API and ONE FULL-SIZE reference step
# ---------------------------------------------------------------------------------------------------------
def _args(**kw):
    import types
    base = dict(gpu=0, arch='cmgan', epochs=100, gen_first=False, comp_type='pow', max_norm=0.0, print_freq=1000,
                debug=False)
    base.update(kw)
    return types.SimpleNamespace(**base)


@pytest.mark.parametrize('arch,weights,gen_first', [('cmgan', (0.1, 0.9, 0.2, 0.05), False),
                                                    ('scp', (0.3, 0.7, 0.2, 0.05), False),
                                                    ('scp', (0.3, 0.7, 0.2, 0.05), True),
                                                    ('cmgan', (0.1, 0.9, 0.2, 0.05), True)])
def test_validate_gan_vs_reference(S, golden, golden2, arch, weights, gen_first):
    """the reference's own validate_gan (fp64 goldens): consistency-preserving generator loss for scp
    (core/function.py:373-397) and the --gen-first gate on the GAN term (:401-413)"""
    import types
    g, d = load_g(S, train=True), load_d(S, train=True)          # validate_gan must switch both to eval itself
    cfg = types.SimpleNamespace(N_FFT=400, HOP_SAMPLES=100, LOSS_WEIGHTS=list(weights))
    batch = {'audio': t(golden['fe_clean']), 'noisy': t(golden['fe_noisy']), 'labels': {'est': t(golden['q_est'])}}
    vg, vd = S.validate_gan([batch], g, d, None, None, 10, _args(arch=arch, gen_first=gen_first), cfg)
    assert not g.training and not d.training
    ref = golden2[f'val_{arch}_f64' + ('_genfirst' if gen_first else '')]
    assert abs(vg - ref[0]) < 3e-4 * abs(ref[0]), (vg, ref[0])
    assert abs(vd - ref[1]) < 1e-3 * abs(ref[1]) + 1e-6, (vd, ref[1])


def test_gen_first_and_clipped_steps(S, golden, golden2):
    """train_gan behind the --gen-first gate (no GAN term, discriminator untouched) and with --max-norm 0.5 (fused
    flat-buffer clip), both against the reference loop (fp64)"""
    import types
    from speech_enhancement_amd import train as TR, optim
    from oracle import se_oracle as Or
    lr = Or.lr_at(10.0, 0.01, 100)
    labels = {'est': t(golden['q_est'])}
    for tag, kw in (('genfirst', dict(gan_on=False)), ('clip', dict(max_norm=0.5))):
        g, d = load_g(S), load_d(S)
        args = types.SimpleNamespace(optimizer='sgd', lr=0.01, weight_decay=0.01, momentum=0.9, max_norm=0.0)
        og, od = optim.build_optimizer(args, g), optim.build_optimizer(args, d, lr=0.02)
        for o in (og, od):
            for grp in o.param_groups:
                grp['lr'] = lr
        out = TR.gan_step(g, d, og, od, t(golden['fe_clean']), t(golden['fe_noisy']), 'cmgan', (0.1, 0.9, 0.2, 0.05),
                          labels=labels, **kw)
        gnorm = np.array([float(v.double().norm()) for v in g.state_dict().values()])
        dnorm = np.array([float(v.double().norm()) for v in d.state_dict().values()])
        np.testing.assert_allclose(gnorm, golden2[f'step_{tag}_g_norm'], rtol=3e-4, atol=1e-5)
        np.testing.assert_allclose(dnorm, golden2[f'step_{tag}_d_norm'], rtol=3e-4, atol=1e-5)
        if tag == 'genfirst':
            assert abs(float(out['loss_g']) - golden2['step_genfirst_losses'][0]) < 2e-4 * abs(float(out['loss_g']))
            assert float(out['loss_d']) == 0.0 and float(out['gan']) == 0.0
        else:
            for k, m_, src in (('g:mask_decoder.final_conv.weight', g, 'generator'),
                               ('d:layers.17.weight_orig', d, 'discriminator')):
                name = k.split(':', 1)[1]
                old = formula.formula_state(src)[name].double().numpy()
                upd_ref = golden2['step_clip_' + k].astype(np.float64) - old
                upd = m_.state_dict()[name].double().cpu().numpy() - old
                assert rms(upd, upd_ref) < 1e-2 * np.sqrt(np.mean(upd_ref ** 2)) + 1e-9, k


def test_frontend_public_api(S, golden, golden2):
    """batch_stft / normalize_batch / power_compress / power_uncompress / disassemble_spectrogram vs the reference's
    own outputs; a non-Hamming window is refused instead of being silently ignored"""
    import types
    from speech_enhancement_amd import _lib
    z = torch.complex(t(golden2['pc_in'][..., 0]), t(golden2['pc_in'][..., 1]))
    for comp in ('pow', 'log', 'norm', 'none'):
        ref = golden2[f'pc_{comp}']
        assert rms(torch.view_as_real(S.power_compress(z, comp)), ref) < 3e-6 * max(1.0, float(np.abs(ref).max())), comp
        ref = golden2[f'pu_{comp}']
        assert rms(torch.view_as_real(S.power_uncompress(z, comp)), ref) < 3e-6 * max(1.0, float(np.abs(ref).max())), comp
    m, r, i = S.disassemble_spectrogram(z)
    assert rms(m, golden2['dis_mag']) < 1e-6 and rms(r, golden2['dis_real']) == 0 and rms(i, golden2['dis_imag']) == 0
    batch = {'audio': t(golden['fe_clean']).cpu(), 'noisy': t(golden['fe_noisy']).cpu()}
    cn, nn_ = S.normalize_batch({k: v.cuda() for k, v in batch.items()}, types.SimpleNamespace(gpu=0))
    assert rms(cn, golden2['bs_clean']) < 1e-6 and rms(nn_, golden2['bs_noisy']) < 1e-6
    cfg = types.SimpleNamespace(N_FFT=400, HOP_SAMPLES=100)
    bs = S.batch_stft(batch, types.SimpleNamespace(gpu=0), cfg)          # host tensors + args.gpu, like the loader's
    assert len(bs) == 8 and all(x.is_cuda for x in bs)
    assert rms(bs[0], golden2['bs_clean']) < 1e-6 and rms(bs[1], golden2['bs_noisy']) < 1e-6
    assert rms(torch.view_as_real(bs[2]), golden2['bs_clean_spec']) < 3e-5
    assert rms(torch.view_as_real(bs[3]), golden2['bs_noisy_spec']) < 3e-5
    assert bs[4].shape == golden2['bs_clean_real'].shape and rms(bs[4], golden2['bs_clean_real']) < 3e-5
    assert rms(bs[5], golden2['bs_clean_imag']) < 3e-5
    assert rms(bs[6], golden2['bs_one_labels']) == 0 and rms(bs[7], golden2['bs_window']) < 1e-7
    # the reference's window is accepted, anything else is loud
    s = S.compressed_stft(bs[1], 400, 100, bs[7])
    assert rms(torch.view_as_real(s), golden2['bs_noisy_spec']) < 3e-5
    with pytest.raises(_lib.SeHipError):
        S.compressed_stft(bs[1], 400, 100, torch.hann_window(400, device='cuda'))
    with pytest.raises(_lib.SeHipError):
        S.uncompressed_istft(s, 400, 100, torch.ones(400, device='cuda'))


def _discriminator_grads_fp64(Or, clean_mag, est_mag, q_est, flips=None, rec=None):
    """fp64 gradients of L_C + L_E (cmgan; core/function.py:286-310) on GIVEN magnitudes [B,1,F,T] (the oracle's discriminator,
    oracle/se_oracle.py:312-331, restated with its PReLU decisions exposed), spectral-norm vectors advanced like the step does:
    generator-pass forward, then D(clean, est) ('gx'), then D(clean, clean) ('yy').  `rec[(tag, stage)]` receives the PReLU
    pre-activations; `flips` = {(tag, stage): [flat indices]} takes the OTHER slope at those entries (a pre-activation within
    rounding distance of zero may legitimately land on either side in fp32)."""
    import torch.nn.functional as F
    sd = {k: (v.double().clone().requires_grad_(not k.endswith(('_u', '_v'))) if v.is_floating_point() else v)
          for k, v in formula.formula_state('discriminator').items()}
    sn = {}

    def fwd(x, y, tag):
        h = torch.cat([x, y], dim=1)
        for si, li in enumerate((0, 3, 6, 9)):
            W = Or.spectral_weight(sd, f'layers.{li}', True, sn)
            h = F.conv2d(h, W, None, stride=2, padding=1)
            m = h.mean(dim=(2, 3), keepdim=True)
            v = ((h - m) ** 2).mean(dim=(2, 3), keepdim=True)
            g, b, a = sd[f'layers.{li+1}.weight'], sd[f'layers.{li+1}.bias'], sd[f'layers.{li+2}.weight']
            yv = (h - m) / torch.sqrt(v + Or.EPS_NORM) * g[None, :, None, None] + b[None, :, None, None]
            pos = yv >= 0
            if rec is not None:
                rec[(tag, si)] = yv.detach()
            if flips and (tag, si) in flips:
                pos = pos.clone()
                pos.view(-1)[torch.tensor(flips[(tag, si)])] ^= True
            h = torch.where(pos, yv, yv * a[None, :, None, None])
        h = h.amax(dim=(2, 3))
        W = Or.spectral_weight(sd, 'layers.14', True, sn)
        h = h @ W.T + sd['layers.14.bias']
        a = sd['layers.16.weight']
        h = torch.where(h >= 0, h, h * a[None, :])
        W = Or.spectral_weight(sd, 'layers.17', True, sn)
        h = h @ W.T + sd['layers.17.bias']
        return torch.sigmoid(sd['layers.18.slope'] * h)

    with torch.no_grad():
        fwd(clean_mag, est_mag, 'gen')
    sd.update(sn)
    d_gx = fwd(clean_mag, est_mag, 'gx')
    sd.update(sn)
    d_yy = fwd(clean_mag, clean_mag, 'yy')
    loss = Or._mse(d_yy.flatten(), torch.ones_like(q_est)) + Or._mse(d_gx.flatten(), q_est)
    names = [k for k, v in sd.items() if torch.is_tensor(v) and v.requires_grad]
    gr = torch.autograd.grad(loss, [sd[k] for k in names], allow_unused=True)
    return {k: gi.detach() for k, gi in zip(names, gr) if gi is not None}


def _explain_by_prelu_kinks(Or, clean_mag, est_mag, q_est, ours, d64, rec, names, tau=2e-5, cap=32):
    """The discriminator gradient is piecewise smooth in its inputs: at a PReLU pre-activation of (numerically) zero either slope is
    a legitimate fp32 outcome.  Candidates = the pre-activations of the two differentiated forwards with |y| < tau (y is an
    InstanceNorm output: O(1)), at most `cap`; for each one the fp64 gradient with THAT decision flipped; the residual ours - fp64 is
    fitted with 0/1 coefficients on the shifts that carry weight (> 2e-5 of the gradient).  Returns (per-tensor relative residual after the fit, chosen candidates)."""
    cands = []
    for key, yv in rec.items():
        if key[0] == 'gen':
            continue
        fl = yv.flatten().abs()
        for idx in torch.nonzero(fl < tau).flatten().tolist():
            cands.append((float(fl[idx]), key, idx))
    cands = sorted(cands)[:cap]
    nrm = {k: float(d64[k].norm()) + 1e-30 for k in names}
    vec = lambda gdict: torch.cat([((gdict[k].double().cpu() - d64[k]) / nrm[k]).flatten() for k in names])
    r = vec(ours)
    if not cands:
        return {k: float((ours[k].double().cpu() - d64[k]).norm()) / nrm[k] for k in names}, []
    cols = [vec(_discriminator_grads_fp64(Or, clean_mag, est_mag, q_est, flips={key: [idx]})) for _, key, idx in cands]
    big = [i for i, col in enumerate(cols) if float(col.norm()) > 2e-5]      # most decisions carry no weight: not fitted
    pick = []
    if big:
        c = torch.linalg.lstsq(torch.stack([cols[i] for i in big], 1), r[:, None]).solution[:, 0]
        pick = [i for i, ci in zip(big, c.tolist()) if ci > 0.5]
    flips = {}
    for i in pick:
        flips.setdefault(cands[i][1], []).append(cands[i][2])
    g_fit = _discriminator_grads_fp64(Or, clean_mag, est_mag, q_est, flips=flips) if pick else d64
    res = {k: float((ours[k].double().cpu() - g_fit[k]).norm()) / nrm[k] for k in names}
    return res, [(cands[i][1], cands[i][2], cands[i][0]) for i in pick]


def test_full_size_train_step_vs_reference(S, golden2, golden4):
    """ONE FULL-SIZE train_gan step of the reference (cmgan, nesterov-SGD, B=2, L=32 000 -> T=321, fp64 golden) vs gan_step:
    the first end-to-end comparison at the benchmark's geometry -- attn_bwd2_kernel<336,8,false>, the triple-tap
    weight-gradient kernels, the 201-wide decoders and the XCD-aware decode all run here.  Every loss term, every
    post-step parameter norm, BatchNorm running statistics and 10 gradients (first nesterov step: update = -1.9 lr g)."""
    import types
    from conftest import full_size_signals
    from speech_enhancement_amd import train as TR, optim
    from oracle import se_oracle as Or
    clean, noisy = full_size_signals(int(golden2['full_step_seed'][0]))
    g, d = load_g(S), load_d(S)
    args = types.SimpleNamespace(optimizer='sgd', lr=0.01, weight_decay=0.01, momentum=0.9, max_norm=0.0)
    og, od = optim.build_optimizer(args, g), optim.build_optimizer(args, d, lr=0.02)
    lr = Or.lr_at(10.0, 0.01, 100)
    for o in (og, od):
        for grp in o.param_groups:
            grp['lr'] = lr
    seen, orig_update = {}, TR._discriminator_update

    def spy(discriminator, optimizer_disc, est_d, clean_pl, *a, **k):
        seen['est'], seen['clean'] = est_d[..., 0].detach().clone(), clean_pl[..., 0].detach().clone()   # [B, T, F] magnitudes
        return orig_update(discriminator, optimizer_disc, est_d, clean_pl, *a, **k)
    TR._discriminator_update = spy
    try:
        out = TR.gan_step(g, d, og, od, clean.cuda(), noisy.cuda(), 'cmgan', (0.1, 0.9, 0.2, 0.05),
                          labels={'est': torch.tensor([0.35, 0.62], device='cuda')})
    finally:
        TR._discriminator_update = orig_update
    torch.cuda.synchronize()
    mse = golden2['full_step_mse_calls']
    errs = {}
    for k, b in (('loss_mag', mse[0]), ('loss_ri', mse[1] + mse[2]), ('gan', mse[3]), ('L_E', mse[4]), ('L_C', mse[5]),
                 ('loss_g', golden2['full_step_losses'][0]), ('loss_d', golden2['full_step_losses'][1])):
        errs[k] = abs(float(out[k]) - b) / abs(b)
        assert errs[k] < (2e-4 if k in ('loss_mag', 'loss_ri', 'loss_g') else 1e-3), (k, float(out[k]), b)
    gs, ds = g.state_dict(), d.state_dict()
    gnorm = np.array([float(v.double().norm()) for v in gs.values()])
    dnorm = np.array([float(v.double().norm()) for v in ds.values()])
    np.testing.assert_allclose(gnorm, golden2['full_step_g_norm'], rtol=3e-4, atol=1e-5)
    np.testing.assert_allclose(dnorm, golden2['full_step_d_norm'], rtol=3e-4, atol=1e-5)
    assert rms(gs['TSCB_1.time_conformer.conv.net.5.running_mean'], golden2['full_step_bn_rm']) < 1e-5
    assert rms(gs['TSCB_1.time_conformer.conv.net.5.running_var'], golden2['full_step_bn_rv']) < 1e-5
    grads = {('g', k): p.grad for k, p in g.named_parameters()}
    grads.update({('d', k): p.grad for k, p in d.named_parameters()})
    # the fp64 discriminator gradients on exactly the enhanced magnitude the HIP generator produced (its distance from the reference's
    # forward is bounded by test_full_size_enhanced_magnitude: 1.4e-5 RMS)
    est_ours = seen['est'].double().cpu().transpose(1, 2)[:, None]
    clean_ours = seen['clean'].double().cpu().transpose(1, 2)[:, None]
    q64 = torch.tensor([0.35, 0.62], dtype=torch.float64)
    rec = {}
    d64 = _discriminator_grads_fp64(Or, clean_ours, est_ours, q64, rec=rec)
    dnames = [k.split(':', 1)[1] for k in golden2.files if k.startswith('full_step_dupd:')]
    kink_res, kinks = _explain_by_prelu_kinks(Or, clean_ours, est_ours, q64, {n: grads[('d', n)] for n in dnames}, d64, rec, dnames)
    print('discriminator gradients on identical inputs: fp64 with the slope flipped at', kinks, '-> residuals', kink_res)
    for k in golden2.files:
        if k.startswith('full_step_gupd:') or k.startswith('full_step_dupd:'):
            name = k.split(':', 1)[1]
            ref = golden2[k].astype(np.float64) / (-lr * 1.9)
            nrm = np.sqrt(np.mean(ref ** 2)) + 1e-30
            e = rms(grads[('g' if 'gupd' in k else 'd', name)], ref) / nrm
            # calibrated bar: 2e-4 + 1.5 x the spread of the REFERENCE's own fp32 run of this step against its fp64 run
            # (golden_v4: 3e-3 .. 1e-2 on the interior tensors: the fp32 floor through 16 InstanceNorms and 8 Conformers)
            spread = rms(golden4['full32_step_' + k[len('full_step_'):]].astype(np.float64) / (-lr * 1.9), ref) / nrm
            if 'dupd' in k:
                # Discriminator tensors, round 5 (root cause of the round-4 exception for its first convolution: tests/diag_d_first_conv.py,
                # profiles/r05_d_first_conv_rootcause.txt): their gradient is a DISCONTINUOUS function of the enhanced magnitude and of
                # the discriminator's own rounding -- on this seed ONE stage-1 PReLU pre-activation (of 2.6 M) lies within 1e-6 of zero
                # and carries a large back-propagated weight: either slope is a legitimate fp32 outcome and moves layers.0.weight_orig
                # by a fixed 1.18e-3 (exactly the reference's own fp32-vs-fp64 spread; a second crossing: 1.64e-3).  So the
                # discriminator's arithmetic is compared on IDENTICAL inputs and decisions: the fp64 gradient on the enhanced magnitude
                # THIS run produced, with the slope choice at the pre-activations of |y| < 2e-5 fitted (0/1; at most 4 accepted) -- bar 2e-4, tighter than
                # any spread.  What remains against the golden (fp64 on our magnitude vs fp64 on the reference's, `e_input`) is a
                # property of the reference's function at two inputs 1.4e-5 apart.
                e_same = kink_res[name]
                e_nofit = rms(grads[('d', name)], d64[name]) / nrm
                e_input = float(np.sqrt(np.mean((d64[name].double().numpy() - ref) ** 2))) / nrm
                errs[name] = (e, spread, e_same, e_nofit, e_input)
                assert e_same < 2e-4, (k, e_same, e_nofit, kinks)
                # the fit is a flexible instrument: it may only explain what the root cause allows -- a handful of decisions, each
                # at a pre-activation within a few fp32 roundings of zero (y is an InstanceNorm output of O(1))
                assert len(kinks) <= 4 and all(ay < 1e-5 for _, _, ay in kinks), kinks
                # against the golden: tensors that needed no flip keep the 2.0x bar; a flipped decision moves a tensor by the
                # reference's own fp32-vs-fp64 spread once more (sanity cap only: the parity statement is e_same)
                assert e < 2e-4 + (3.0 if e_nofit > 2e-4 else 2.0) * spread, (k, e, spread, e_nofit, e_input)
                continue
            errs[name] = (e, spread)
            assert e < 2e-4 + 1.5 * spread, (k, e, spread)
    print('full-size step relative errors (ours, reference fp32 spread):',
          {k: (tuple(float('%.2e' % x) for x in v) if isinstance(v, tuple) else float('%.2e' % v)) for k, v in errs.items()})


@pytest.mark.parametrize('kind', ['lars', 'lamb'])
def test_fused_lars_lamb(S, golden, kind):
    """flat-buffer LARS / Lamb kernels: (1) the reference's own two-step toy trajectory (golden), (2) two steps on the
    real discriminator against the per-tensor torch restatement that the CPU suite pins to the same golden"""
    import types
    from speech_enhancement_amd import optim
    w = torch.nn.Parameter(torch.from_numpy(np.sin(np.arange(24) * 0.7).reshape(4, 6).astype(np.float32)).cuda())
    bb = torch.nn.Parameter(torch.from_numpy(np.cos(np.arange(4) * 1.3).astype(np.float32)).cuda())
    groups = [{'params': [w]}, {'params': [bb], 'weight_decay': 0.}]
    if kind == 'lars':
        opt = optim.FlatOptimizer(groups, 'lars', 0.1, weight_decay=0.01, momentum=0.9)
    else:
        opt = optim.FlatOptimizer(groups, 'lamb', 0.01, weight_decay=0.01, max_grad_norm=1.0)
    for stp in range(2):
        opt.zero_grad()
        w.grad.copy_(torch.from_numpy(np.cos(np.arange(24) * 0.3 + stp).reshape(4, 6).astype(np.float32)))
        bb.grad.copy_(torch.from_numpy(np.sin(np.arange(4) * 0.9 + stp).astype(np.float32)))
        opt.step()
    assert rms(w, golden[f'opt_{kind}_w']) < 2e-6 and rms(bb, golden[f'opt_{kind}_b']) < 2e-6
    # real model: same random gradients into both implementations
    torch.manual_seed(0)
    d1, d2 = load_d(S), load_d(S)
    args = types.SimpleNamespace(optimizer=kind, lr=0.05 if kind == 'lars' else 0.01, weight_decay=0.01, momentum=0.9,
                                 max_norm=1.0)
    o1 = optim.build_optimizer(args, d1)
    assert isinstance(o1, optim.FlatOptimizer)
    g2 = optim.set_weight_decay(d2)
    o2 = optim.TorchLARS(g2, args.lr, weight_decay=0.01, momentum=0.9) if kind == 'lars' else \
        optim.TorchLamb(g2, lr=args.lr, weight_decay=0.01, max_grad_norm=1.0)
    for stp in range(2):
        o1.zero_grad()
        for (n1, p1), (n2, p2) in zip(d1.named_parameters(), d2.named_parameters()):
            gr = torch.randn_like(p2) * (0.1 + 0.05 * stp)
            p1.grad.copy_(gr)
            p2.grad = gr.clone()
        o1.step()
        o2.step()
    for (n1, p1), (n2, p2) in zip(d1.named_parameters(), d2.named_parameters()):
        assert rms(p1, p2) < 1e-6 * max(1.0, float(p2.abs().max())), n1


def test_flat_clip_and_state_dict_interop(S):
    """clip_grad_norm on the flat buffers == torch.nn.utils.clip_grad_norm_; optimizer state moves both ways between
    FlatOptimizer and torch.optim (the checkpoint's 'optimizer' entry, main_gan.py:117, 300-309)"""
    import types
    from speech_enhancement_amd import optim
    torch.manual_seed(1)
    d1, d2 = load_d(S), load_d(S)
    for kind, lr in (('adamw', 5e-4), ('sgd', 0.01)):
        args = types.SimpleNamespace(optimizer=kind, lr=lr, weight_decay=0.01, momentum=0.9, max_norm=0.0)
        o1 = optim.build_optimizer(args, d1)
        g2 = optim.set_weight_decay(d2)
        o2 = torch.optim.AdamW(g2, lr=lr, weight_decay=0.01) if kind == 'adamw' else \
            torch.optim.SGD(g2, lr=lr, momentum=0.9, nesterov=True)
        for stp in range(2):
            o1.zero_grad()
            for p1, p2 in zip(d1.parameters(), d2.parameters()):
                gr = torch.randn_like(p2)
                p1.grad.copy_(gr)
                p2.grad = gr.clone()
            n1 = o1.clip_grad_norm(0.7)
            n2 = torch.nn.utils.clip_grad_norm_(d2.parameters(), 0.7)
            assert abs(float(n1) - float(n2)) < 1e-5 * float(n2)
            for p1, p2 in zip(d1.parameters(), d2.parameters()):
                assert rms(p1.grad, p2.grad) < 1e-7
            o1.step()
            o2.step()
            if stp == 0:          # swap the states through the serialised form, then keep stepping
                sd1, sd2 = o1.state_dict(), o2.state_dict()
                assert set(sd1) == {'state', 'param_groups'} and len(sd1['state']) == len(sd2['state'])
                o2.load_state_dict(sd1)
                o1.load_state_dict(sd2)
        for (n, p1), p2 in zip(d1.named_parameters(), d2.parameters()):
            assert rms(p1, p2) < 2e-6 * max(1.0, float(p2.abs().max())), (kind, n)


@pytest.mark.parametrize('backend', ['gloo', 'nccl'])
def test_world1_hooks_equal_plain_step(S, backend):
    """the data-parallel code paths on ONE rank (process group of size 1): deferred generator step behind the
    discriminator step, SyncBatchNorm exchange with count = M * world, averaged scp gradients -- with world == 1 every
    collective is the identity, so the result must equal the hook-free step (to the fp32-atomics noise of the weight gradients).
    backend 'nccl' (= RCCL): the DEVICE-buffer branch of the hooks -- in-place all-reduce of the flat gradient buffer with
    async_op=True, statistics reduced on the device (train.DataParallelHooks, stage_host False) -- the branch an N > 1 run on
    real hardware takes; 'gloo': the host-staged test transport."""
    import os
    import socket
    import types
    import torch.distributed as dist
    from speech_enhancement_amd import train as TR, optim, layers as LY
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    if backend == 'nccl':
        dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', torch.cuda.current_device()))
    else:
        dist.init_process_group('gloo', rank=0, world_size=1)
    try:
        torch.manual_seed(3)
        clean = 0.1 * torch.randn(2, 1600, device='cuda')
        noisy = clean + 0.05 * torch.randn(2, 1600, device='cuda')
        labels = {'est': torch.tensor([0.4, 0.6], device='cuda'), 'clean': torch.tensor([0.95, 0.97], device='cuda'),
                  'noisy': torch.tensor([0.3, 0.2], device='cuda')}
        res = []
        for use_hooks in (False, True):
            g, d = load_g(S), load_d(S)
            args = types.SimpleNamespace(optimizer='sgd', lr=0.01, weight_decay=0.01, momentum=0.9, max_norm=0.0)
            og, od = optim.build_optimizer(args, g), optim.build_optimizer(args, d, lr=0.02)
            hooks = None
            if use_hooks:
                hooks = TR.attach_data_parallel(g, d)
                assert hooks.world == 1 and hooks.stage_host == (backend == 'gloo')
                hooks.force_sync = True              # take the two-phase SyncBatchNorm backward although world == 1
                calls = []
                orig = hooks.allreduce
                hooks.allreduce = lambda t_, **kw: (calls.append(t_.numel()), orig(t_, **kw))[1]
                g.dp = hooks
            out = TR.gan_step(g, d, og, od, clean, noisy, 'scp', (0.3, 0.7, 0.2, 0.05), labels=labels, hooks=hooks)
            res.append(({k: float(v) for k, v in out.items()},
                        torch.cat([p.detach().flatten() for p in list(g.parameters()) + list(d.parameters())]).clone()))
            if use_hooks:
                assert len(calls) >= 8 + 8 + 1, calls      # 8 BN forward + 8 BN backward exchanges, the scp gradient triple
        for k in res[0][0]:          # weight-gradient atomics make the gradient dot products differ in the last digits
            assert abs(res[0][0][k] - res[1][0][k]) <= 1e-6 * abs(res[0][0][k]), k
        # weight gradients use fp32 atomics (run-to-run order): compare to atomics noise, not bitwise
        assert float((res[0][1] - res[1][1]).abs().max()) < 1e-6
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('arch', ['cmgan', 'scp', 'scp_cond'])
def test_streams_do_not_change_the_step(S, arch):
    """HIP-stream concurrency inside a step (weight gradients on a leaf stream during the generator backward; + the two-stage
    backward with the discriminator step on a side stream) against the fully serial order: losses and EVERY parameter of both
    models after two SGD steps, per tensor, relative to the size of that tensor's update.  Weight gradients accumulate with fp32
    atomics over ~100 row chunks whose partial sums cancel, so two serial runs already differ by up to a few 1e-4 of the update
    (measured: tools/streams_diag.py; back-to-back runs much less, later ones more -- the order follows the clocks); the
    two-stage backward additionally re-orders the sum of the contributions to d loss / d est (1e-7, amplified by the
    ill-conditioned first layers).  A gradient consumed before its stream had written it would be off by its WHOLE contribution
    (the padded 1- / 2-channel convolutions were, before their accumulate moved to the leaf stream)."""
    import types
    from speech_enhancement_amd import train as TR, optim, gemm as GM
    torch.manual_seed(5)
    clean = 0.1 * torch.randn(4, 3200, device='cuda')
    noisy = clean + 0.05 * torch.randn(4, 3200, device='cuda')
    cond = arch == 'scp_cond'
    if cond:
        #  scp on a WELL-CONDITIONED clip pair (no near-zero bin in the re-analysed spectrum): the two-step parameter
        # comparison holds there too, so a stream-ordering race on the scratch recycled in step 2 of scp is detected
        arch = 'scp'
        clean, noisy = (x.cuda() for x in formula.cond_signals(4, 3200, 7))
    labels = {'est': torch.tensor([0.4, 0.6, 0.5, 0.7], device='cuda'), 'clean': torch.full((4,), 0.96, device='cuda'),
              'noisy': torch.tensor([0.3, 0.2, 0.25, 0.35], device='cuda')}
    w = (0.1, 0.9, 0.2, 0.05) if arch == 'cmgan' else (0.3, 0.7, 0.2, 0.05)
    saved = (GM._LeafStream.enabled, TR._D_OVERLAP, GM.branch_stream.enabled)
    res, init = [], None
    try:
        for leaf, dside in ((False, False), (True, False), (True, True)):
            GM._LeafStream.enabled, TR._D_OVERLAP, GM.branch_stream.enabled = leaf, dside, leaf      # decoder branch stream too
            g, d = load_g(S), load_d(S)
            named = lambda: list(g.named_parameters()) + [('D.' + n, p) for n, p in d.named_parameters()]
            if init is None:
                init = {n: p.detach().clone() for n, p in named()}
            args = types.SimpleNamespace(optimizer='sgd', lr=0.01, weight_decay=0.01, momentum=0.9, max_norm=0.0)
            og, od = optim.build_optimizer(args, g), optim.build_optimizer(args, d, lr=0.02)
            outs, first = [], None
            for stp in range(2):                                                                         # 2 steps: recycled scratch
                outs.append(TR.gan_step(g, d, og, od, clean, noisy, arch, w, labels=labels))
                if stp == 0:
                    torch.cuda.synchronize()
                    first = {n: p.detach().clone() for n, p in named()}
            torch.cuda.synchronize()
            # cmgan: the parameters after BOTH steps; scp: after the FIRST -- its consistency-path gradient on white noise is
            # chaotic (|z|^-0.7 at near-zero bins): one-ulp differences of the step-1 parameters (all a
            # re-ordering can cause: measured 3.7e-9 absolute between any two stream modes, tools/streams_diag.py with STEPS=1)
            # move single step-2 gradient tensors by percents in ANY two runs, serial or not
            res.append(([{k: float(v) for k, v in o.items()} for o in outs],
                        first if (arch == 'scp' and not cond) else {n: p.detach().clone() for n, p in named()}))
    finally:
        GM._LeafStream.enabled, TR._D_OVERLAP, GM.branch_stream.enabled = saved
    for other in (1, 2):
        for si, (o0, o1) in enumerate(zip(res[0][0], res[other][0])):
            for k in o0:
                tol = 1e-4 if (arch == 'cmgan' or si == 0 or cond) else 5e-3
                assert abs(o0[k] - o1[k]) <= tol * abs(o0[k]) + 1e-7, (other, si, k, o0[k], o1[k])
    bad = []
    for n, p0 in res[0][1].items():
        scale = max(1.0, float(p0.abs().max()))
        upd = float((p0 - init[n]).abs().max())
        d_leaf = float((p0 - res[1][1][n]).abs().max())
        d_all = float((p0 - res[2][1][n]).abs().max())
        if d_leaf > 5e-3 * upd + 1e-6 * scale:       # a missed contribution is O(upd); order noise reached 3e-3 (prelu_out, 201 weights)
            bad.append(('leaf stream', n, d_leaf, upd))
        if d_all > 1e-2 * upd + 1e-6 * scale:
            bad.append(('all streams', n, d_all, upd))
    assert not bad, bad[:10]
