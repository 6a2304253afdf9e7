"""Generate the golden vectors under tests/golden/ by IMPORTING THE REFERENCE (read-only,
/root/reference) in the build container.  Run:  python tests/golden/make_golden.py

The reference is used only as a black box: formula weights are loaded through
``load_state_dict`` and inputs/outputs are stored.  Nothing from /root/reference is copied.
Stubs are installed for the three absent third-party modules (pesq, timm, termcolor); PESQ
labels are supplied as inputs (PESQ parity unpinned).
"""
import json
import os
import sys
import types

os.environ['PYTHONDONTWRITEBYTECODE'] = '1'
sys.dont_write_bytecode = True
import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
REF = '/root/reference'

# ---- stubs for absent third-party modules ---------------------------------
_Q = []


def _fake_pesq(sr, ref, deg, mode):
    raise RuntimeError('pesq is not available; labels are injected')


sys.modules['pesq'] = types.SimpleNamespace(pesq=_fake_pesq)


class _AverageMeter:
    def __init__(self):
        self.val = self.avg = self.sum = self.count = 0

    def update(self, v, n=1):
        self.val = v
        self.sum += v * n
        self.count += n
        self.avg = self.sum / self.count


_timm = types.ModuleType('timm')
_timm_utils = types.ModuleType('timm.utils')
_timm_utils.AverageMeter = _AverageMeter
_timm.utils = _timm_utils
sys.modules['timm'] = _timm
sys.modules['timm.utils'] = _timm_utils
sys.modules['termcolor'] = types.SimpleNamespace(colored=lambda s, *a, **k: s)
sys.path.insert(0, REF)

from models.generator import TSCNet                     # noqa: E402
from models.discriminator import Discriminator          # noqa: E402
from models.conformer import ConformerBlock             # noqa: E402
import core.function as RF                              # noqa: E402
from core.optimizer import build_optimizer              # noqa: E402
from utils.utils import adjust_learning_rate            # noqa: E402
import formula                                          # noqa: E402

torch.set_num_threads(8)
torch.manual_seed(0)


def base_of(name, t):
    if t.dim() == 1 and not name.endswith('.bias') and t.is_floating_point():
        return float(np.round(float(t.float().mean()), 4))
    return 0.0


def write_spec():
    g = TSCNet(64, 201)
    d = Discriminator(16)
    spec = {}
    for which, m in (('generator', g), ('discriminator', d)):
        spec[which] = [[k, list(v.shape), str(v.dtype).replace('torch.', ''), base_of(k, v)]
                       for k, v in m.state_dict().items()]
    with open(os.path.join(HERE, 'state_spec.json'), 'w') as f:
        json.dump(spec, f)
    return spec


def no_dropout(m):
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    return m


def signals(B, L, seed):
    rs = np.random.RandomState(seed)
    clean = (0.1 * rs.randn(B, L)).astype(np.float32)
    noisy = (clean + 0.05 * rs.randn(B, L)).astype(np.float32)
    return torch.from_numpy(clean), torch.from_numpy(noisy)


def c2np(z):
    return np.stack([z.real.numpy(), z.imag.numpy()], -1).astype(np.float32)


def main():
    write_spec()
    gsd = formula.formula_state('generator')
    dsd = formula.formula_state('discriminator')
    out = {}
    win = torch.hamming_window(400)

    # ---- 1. front-end ------------------------------------------------------
    clean, noisy = signals(2, 1600, 1)
    out['fe_clean'] = clean.numpy()
    out['fe_noisy'] = noisy.numpy()
    cn, nn_ = RF.normalize_batch({'audio': clean, 'noisy': noisy}, types.SimpleNamespace(gpu=None))
    out['fe_clean_n'] = cn.numpy()
    out['fe_noisy_n'] = nn_.numpy()
    for comp in ('pow', 'log', 'norm', 'none'):
        s = RF.compressed_stft(nn_, 400, 100, win, comp_type=comp)
        out[f'fe_spec_{comp}'] = c2np(s)
        out[f'fe_istft_{comp}'] = RF.uncompressed_istft(s, 400, 100, win, comp_type=comp).numpy()
    noisy_spec = RF.compressed_stft(nn_, 400, 100, win)
    clean_spec = RF.compressed_stft(cn, 400, 100, win)

    # ---- 2. conformer blocks ----------------------------------------------
    g = TSCNet(64, 201)
    g.load_state_dict(gsd)
    no_dropout(g)
    blk = g.TSCB_1.time_conformer
    for tag, shape in (('t17', (3, 17, 64)), ('f101', (2, 101, 64))):
        x = torch.from_numpy(np.sin(np.arange(np.prod(shape)) * 0.123).reshape(shape).astype(np.float32))
        g.load_state_dict(gsd)
        blk.train()
        xr = x.clone().requires_grad_(True)
        y = blk(xr)
        (y * torch.cos(torch.arange(y.numel()).view_as(y) * 0.01)).sum().backward()
        out[f'cf_{tag}_x'] = x.numpy()
        out[f'cf_{tag}_y_train'] = y.detach().numpy()
        out[f'cf_{tag}_dx'] = xr.grad.numpy()
        out[f'cf_{tag}_rm'] = blk.conv.net[5].running_mean.numpy().copy()
        out[f'cf_{tag}_rv'] = blk.conv.net[5].running_var.numpy().copy()
        out[f'cf_{tag}_dE'] = blk.attn.fn.rel_pos_emb.weight.grad.numpy().copy()
        out[f'cf_{tag}_dWdw'] = blk.conv.net[4].conv.weight.grad.numpy().copy()
        g.zero_grad()
        g.load_state_dict(gsd)
        blk.eval()
        with torch.no_grad():
            out[f'cf_{tag}_y_eval'] = blk(x).numpy()
    # attention alone with the +-512 clamp active
    x = torch.from_numpy(np.sin(np.arange(600 * 64) * 0.0371).reshape(1, 600, 64).astype(np.float32))
    g.load_state_dict(gsd)
    blk.eval()
    with torch.no_grad():
        out['attn600_x'] = x.numpy()
        out['attn600_y'] = blk.attn(x).numpy()

    # ---- 3. TSCNet forward / backward at T=17 -----------------------------
    g.load_state_dict(gsd)
    g.train()
    er, ei = g(noisy_spec)
    out['g_real'] = er.detach().numpy()
    out['g_imag'] = ei.detach().numpy()
    wr = torch.cos(torch.arange(er.numel()).view_as(er) * 0.013)
    wi = torch.sin(torch.arange(ei.numel()).view_as(ei) * 0.017)
    (er * wr + ei * wi).sum().backward()
    out['g_gradnorm'] = np.array([float(p.grad.norm()) for _, p in g.named_parameters()], np.float32)
    out['g_gradsum'] = np.array([float(p.grad.sum()) for _, p in g.named_parameters()], np.float32)
    for k in ('dense_encoder.conv_1.0.weight', 'TSCB_2.freq_conformer.attn.fn.to_q.weight',
              'mask_decoder.prelu_out.weight', 'TSCB_4.time_conformer.conv.net.4.conv.weight',
              'complex_decoder.conv.weight', 'dense_encoder.dilated_dense.conv4.weight'):
        out['g_grad:' + k] = dict(g.named_parameters())[k].grad.numpy().copy()
    g.zero_grad()
    # same thing in float64: pins the oracle's algorithm far below fp32 noise
    g64 = no_dropout(TSCNet(64, 201))
    g64.load_state_dict(gsd)
    g64.double().train()
    spec64 = torch.complex(noisy_spec.real.double(), noisy_spec.imag.double())
    er, ei = g64(spec64)
    out['g64_real'] = er.detach().numpy()
    (er * wr.double() + ei * wi.double()).sum().backward()
    out['g64_gradnorm'] = np.array([float(p.grad.norm()) for _, p in g64.named_parameters()], np.float64)
    for k in ('dense_encoder.conv_1.0.weight', 'TSCB_2.freq_conformer.attn.fn.to_q.weight',
              'TSCB_3.time_conformer.attn.fn.rel_pos_emb.weight', 'mask_decoder.prelu_out.weight'):
        out['g64_grad:' + k] = dict(g64.named_parameters())[k].grad.numpy().copy()
    del g64
    g.load_state_dict(gsd)
    g.eval()
    with torch.no_grad():
        er, ei = g(noisy_spec)
    out['g_real_eval'] = er.numpy()
    out['g_imag_eval'] = ei.numpy()

    # ---- 4. discriminator --------------------------------------------------
    d = Discriminator(16)
    d.load_state_dict(dsd)
    no_dropout(d)
    d.train()
    cm = clean_spec.abs().unsqueeze(1)
    nm = noisy_spec.abs().unsqueeze(1).clone().requires_grad_(True)
    y = d(cm, nm)
    out['d_in_clean_mag'] = cm.numpy()
    out['d_in_noisy_mag'] = nm.detach().numpy()
    out['d_out_train'] = y.detach().numpy()
    (y.flatten() * torch.tensor([1.0, -2.0])).sum().backward()
    out['d_dnoisy'] = nm.grad.numpy()
    out['d_gradnorm'] = np.array([float(p.grad.norm()) for _, p in d.named_parameters()], np.float32)
    out['d_grad:layers.0.weight_orig'] = d.layers[0].weight_orig.grad.numpy().copy()
    out['d_grad:layers.17.weight_orig'] = d.layers[17].weight_orig.grad.numpy().copy()
    for li in (0, 3, 6, 9, 14, 17):
        out[f'd_u{li}'] = d.layers[li].weight_u.numpy().copy()
        out[f'd_v{li}'] = d.layers[li].weight_v.numpy().copy()
    d64 = no_dropout(Discriminator(16))
    d64.load_state_dict(dsd)
    d64.double().train()
    nm64 = nm.detach().double().requires_grad_(True)
    y64 = d64(cm.double(), nm64)
    (y64.flatten() * torch.tensor([1.0, -2.0], dtype=torch.float64)).sum().backward()
    out['d64_out'] = y64.detach().numpy()
    out['d64_dnoisy'] = nm64.grad.numpy()
    out['d64_gradnorm'] = np.array([float(p.grad.norm()) for _, p in d64.named_parameters()], np.float64)
    del d64
    d.load_state_dict(dsd)
    d.eval()
    with torch.no_grad():
        out['d_out_eval'] = d(cm, nm.detach()).numpy()

    # ---- 5. full train_gan steps (reference loop itself) -------------------
    torch.cuda.synchronize = lambda *a, **k: None
    torch.cuda.max_memory_allocated = lambda *a, **k: 0
    qs = {'est': torch.tensor([0.35, 0.62]), 'clean': torch.tensor([0.97, 0.93]),
          'noisy': torch.tensor([0.21, 0.44])}
    out['q_est'], out['q_clean'], out['q_noisy'] = (qs[k].numpy() for k in ('est', 'clean', 'noisy'))

    class _Log:
        def info(self, *a, **k):
            pass

    W_CM, W_SCP = [0.1, 0.9, 0.2, 0.05], [0.3, 0.7, 0.2, 0.05]
    _orig_float = torch.Tensor.float
    # float64 runs (the loop itself under set_default_dtype(float64)) pin the algorithm; the
    # float32 cmgan runs show the fp32 noise floor.  The scp/cp gradient is ill-conditioned in
    # fp32 (pow(mag,0.3) backward at near-zero bins of the re-STFT), so scp is pinned in fp64 only.
    for arch, weights, optname, dt in (('cmgan', W_CM, 'sgd', 'f32'), ('cmgan', W_CM, 'adamw', 'f32'),
                                       ('cmgan', W_CM, 'sgd', 'f64'), ('cmgan', W_CM, 'adamw', 'f64'),
                                       ('scp', W_SCP, 'sgd', 'f64'), ('scp', W_SCP, 'adamw', 'f64'),
                                       ('scp', W_SCP, 'sgd', 'f32')):
        base_lr = 0.01 if optname == 'sgd' else 5e-4
        tdt = torch.float64 if dt == 'f64' else torch.float32
        torch.set_default_dtype(tdt)
        # the loop calls one_labels.float() (core/function.py:262); under the fp64 pin that must
        # stay float64 or autograd rejects the mixed-dtype MSE
        torch.Tensor.float = (lambda self: self.to(torch.float64)) if dt == 'f64' else _orig_float
        g = no_dropout(TSCNet(64, 201))
        d = no_dropout(Discriminator(16))
        g.load_state_dict(gsd)
        d.load_state_dict(dsd)
        g.to(tdt)
        d.to(tdt)
        args = types.SimpleNamespace(debug=False, gpu=None, arch=arch, epochs=100, gen_first=False,
                                     max_norm=0.0, print_freq=1000, comp_type='pow',
                                     optimizer=optname, lr=base_lr, weight_decay=0.01, momentum=0.9)
        sched = types.SimpleNamespace(LR=base_lr, EPOCHS=100, CYCLE_LIMIT=4, WARMUP_EPOCHS=4, MIN_LR=1e-6)
        config = types.SimpleNamespace(N_FFT=400, HOP_SAMPLES=100, LOSS_WEIGHTS=weights,
                                       TRAIN=types.SimpleNamespace(SCHEDULER=sched))
        og = build_optimizer(args, g)
        od = build_optimizer(args, d, lr=args.lr * 2)
        order = ['est', 'clean', 'noisy'] if arch == 'scp' else ['est']
        calls = []

        def fake_batch_pesq(c, n, _order=order, _calls=calls):
            k = _order[len(_calls) % len(_order)]
            _calls.append(k)
            return qs[k].clone().to(torch.get_default_dtype())

        RF.batch_pesq = fake_batch_pesq
        # capture loss terms through the criterion (order of calls is fixed by the loop)
        seen = []

        class Crit(torch.nn.MSELoss):
            def forward(self, a, b):
                v = super().forward(a, b)
                seen.append(float(v.detach().double()))
                return v

        loader = [{'audio': clean.clone().to(tdt), 'noisy': noisy.clone().to(tdt)}]
        gl, dl = RF.train_gan(loader, g, d, Crit(), og, od, _Log(), 10, args, config)
        out[f'step_{arch}_{optname}_{dt}_gen_loss'] = np.float32(gl)
        out[f'step_{arch}_{optname}_{dt}_disc_loss'] = np.float32(dl)
        out[f'step_{arch}_{optname}_{dt}_mse_calls'] = np.array(seen, np.float64)
        out[f'step_{arch}_{optname}_{dt}_lr'] = np.float32(og.param_groups[0]['lr'])
        gs, ds = g.state_dict(), d.state_dict()
        out[f'step_{arch}_{optname}_{dt}_g_norm'] = np.array([float(v.double().norm()) for v in gs.values()], np.float64)
        out[f'step_{arch}_{optname}_{dt}_g_sum'] = np.array([float(v.double().sum()) for v in gs.values()], np.float64)
        out[f'step_{arch}_{optname}_{dt}_d_norm'] = np.array([float(v.double().norm()) for v in ds.values()], np.float64)
        out[f'step_{arch}_{optname}_{dt}_d_sum'] = np.array([float(v.double().sum()) for v in ds.values()], np.float64)
        out[f'step_{arch}_{optname}_{dt}_g:mask_decoder.final_conv.weight'] = gs['mask_decoder.final_conv.weight'].numpy()
        out[f'step_{arch}_{optname}_{dt}_g:TSCB_1.time_conformer.attn.fn.to_q.weight'] = \
            gs['TSCB_1.time_conformer.attn.fn.to_q.weight'].numpy()
        out[f'step_{arch}_{optname}_{dt}_d:layers.17.weight_orig'] = ds['layers.17.weight_orig'].numpy()
        out[f'step_{arch}_{optname}_{dt}_d:layers.0.weight_u'] = ds['layers.0.weight_u'].numpy()

    torch.set_default_dtype(torch.float32)
    torch.Tensor.float = _orig_float

    # ---- 6. LR schedule ----------------------------------------------------
    sched = types.SimpleNamespace(LR=0.01, EPOCHS=100, CYCLE_LIMIT=4, WARMUP_EPOCHS=4, MIN_LR=1e-6)
    config = types.SimpleNamespace(TRAIN=types.SimpleNamespace(SCHEDULER=sched))
    opt = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=1.0)
    eps = [0, 0.5, 3.99, 4, 10, 24.9, 25, 26.5, 29, 49.99, 50, 60.25, 75, 99.5]
    out['lr_epochs'] = np.array(eps, np.float64)
    lrs = []
    for e in eps:
        adjust_learning_rate([opt], e, config)
        lrs.append(opt.param_groups[0]['lr'])
    out['lr_values'] = np.array(lrs, np.float64)

    # ---- 6b. LARS / Lamb: two steps of the reference's optimizers on a toy parameter set ------------------
    from core.optimizer import LARS as RefLARS, Lamb as RefLamb
    for oname, ctor in (('lars', lambda ps: RefLARS(ps, lr=0.1, weight_decay=0.01, momentum=0.9)),
                        ('lamb', lambda ps: RefLamb(ps, lr=0.01, weight_decay=0.01, max_grad_norm=1.0))):
        w = torch.nn.Parameter(torch.from_numpy(np.sin(np.arange(24) * 0.7).reshape(4, 6).astype(np.float32)))
        bb = torch.nn.Parameter(torch.from_numpy(np.cos(np.arange(4) * 1.3).astype(np.float32)))
        opt = ctor([{'params': [w]}, {'params': [bb], 'weight_decay': 0.}])
        for stp in range(2):
            w.grad = torch.from_numpy(np.cos(np.arange(24) * 0.3 + stp).reshape(4, 6).astype(np.float32))
            bb.grad = torch.from_numpy(np.sin(np.arange(4) * 0.9 + stp).astype(np.float32))
            opt.step()
        out[f'opt_{oname}_w'] = w.detach().numpy().copy()
        out[f'opt_{oname}_b'] = bb.detach().numpy().copy()

    # ---- 7. full-size (2 s) generator forward: the headline parity quantity -
    clean2, noisy2 = signals(1, 32000, 7)
    cn2, nn2 = RF.normalize_batch({'audio': clean2, 'noisy': noisy2}, types.SimpleNamespace(gpu=None))
    sp = RF.compressed_stft(nn2, 400, 100, win)
    g = no_dropout(TSCNet(64, 201))
    g.load_state_dict(gsd)
    g.train()
    with torch.no_grad():
        er, ei = g(sp)
    out['full_noisy'] = noisy2.numpy()
    out['full_est_mag'] = torch.sqrt(er ** 2 + ei ** 2)[0, 0].numpy().astype(np.float32)   # [T,F]
    out['full_est_audio'] = RF.uncompressed_istft(
        torch.complex(er, ei).squeeze(1).permute(0, 2, 1), 400, 100, win).numpy()

    np.savez_compressed(os.path.join(HERE, 'golden_v1.npz'), **out)
    print('wrote', len(out), 'arrays;', os.path.getsize(os.path.join(HERE, 'golden_v1.npz')) / 1e6, 'MB')


if __name__ == '__main__':
    main()
