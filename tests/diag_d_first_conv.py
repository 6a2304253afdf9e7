"""Root cause of the discriminator's first-convolution gradient sitting above the reference's own fp32 spread in
tests/test_model_gpu.py::test_full_size_train_step_vs_reference (round-4 review, "a bar was widened to pass").  CPU only, imports the
oracle (test infrastructure).  Output of a run: profiles/r05_d_first_conv_rootcause.txt.

Finding: the gradient of `layers.0.weight_orig` is a DISCONTINUOUS function of the enhanced magnitude -- the discriminator's PReLUs
(models/discriminator.py:41-50) switch slope where a pre-activation crosses zero, and the rounding of the GENERATOR forward
(1.4e-5 relative RMS in fp32) moves a handful of 2.6 M pre-activations across: a crossing shifts this gradient by a fixed
quantum that depends on the pixel (one stage-1 crossing: 1.18e-3 -- exactly the reference's own fp32-vs-fp64 spread --, 1.64e-3
with the next two; four other crossings together: 1.7e-5).  The
discriminator's own arithmetic contributes 5e-5 (fp32 D on identical inputs).  A random perturbation of the same RMS without a
crossing moves the gradient by 7e-5."""
import sys, time, numpy as np, torch
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
import formula
from conftest import full_size_signals
from oracle import se_oracle as Or
import torch.nn.functional as F
torch.set_num_threads(8)
g2 = np.load(os.path.join(ROOT, 'tests/golden/golden_v2.npz')); g4 = np.load(os.path.join(ROOT, 'tests/golden/golden_v4.npz'))
lr = Or.lr_at(10.0, 0.01, 100)
NAME = 'layers.0.weight_orig'
ref = g2['full_step_dupd:' + NAME].astype(np.float64) / (-lr * 1.9)
r32 = g4['full32_step_dupd:' + NAME].astype(np.float64) / (-lr * 1.9)
nrm = np.sqrt(np.mean(ref ** 2))
rel = lambda a: float(np.sqrt(np.mean((np.asarray(a, np.float64) - ref) ** 2)) / nrm)
print('reference fp32 run vs its fp64 run: %.3e' % rel(r32))
clean, noisy = full_size_signals(int(g2['full_step_seed'][0]))
q = torch.tensor([0.35, 0.62])

def gen_est(dt):
    sdg = {k: (v.to(dt) if v.is_floating_point() else v) for k, v in formula.formula_state('generator').items()}
    c, n, _ = Or.normalize_pair(clean.to(dt), noisy.to(dt))
    with torch.no_grad():
        spec = Or.compressed_stft(n)
        er, ei = Or.tscnet_forward(sdg, spec, True, None)
        er, ei = er.permute(0, 1, 3, 2), ei.permute(0, 1, 3, 2)
        est = torch.sqrt(er ** 2 + ei ** 2)
        cm = Or.compressed_stft(c).abs()[:, None]
    return cm, est

ARG = {}; TOP = {}; SN = {}; SGN = {}; MINABS = {}
def d_forward(sd, x, y, rnd, tag=''):
    h = torch.cat([x, y], dim=1)
    for si, li in enumerate((0, 3, 6, 9)):
        W = Or.spectral_weight(sd, f'layers.{li}', True, SN)
        h = rnd(f'conv{si}', F.conv2d(h, W, None, stride=2, padding=1))
        m = rnd(f'mean{si}', h.mean(dim=(2, 3), keepdim=True))
        v = rnd(f'var{si}', ((h - m) ** 2).mean(dim=(2, 3), keepdim=True))
        g, b, a = sd[f'layers.{li+1}.weight'], sd[f'layers.{li+1}.bias'], sd[f'layers.{li+2}.weight']
        yv = (h - m) / torch.sqrt(v + Or.EPS_NORM) * g[None, :, None, None] + b[None, :, None, None]
        SGN[(tag, si)] = (yv >= 0).clone(); MINABS[(tag, si)] = float(yv.abs().min())
        h = rnd(f'act{si}', torch.where(yv >= 0, yv, yv * a[None, :, None, None]))
    ARG[tag] = h.flatten(2).argmax(2).clone(); TOP[tag] = h.flatten(2).topk(2, dim=2).values.detach().clone()
    h = h.amax(dim=(2, 3))
    W = Or.spectral_weight(sd, 'layers.14', True, SN)
    h = h @ W.T + sd['layers.14.bias']
    a = sd['layers.16.weight']
    h = torch.where(h >= 0, h, h * a[None, :])
    W = Or.spectral_weight(sd, 'layers.17', True, SN)
    h = rnd('logit', h @ W.T + sd['layers.17.bias'])
    return rnd('out', torch.sigmoid(sd['layers.18.slope'] * h))

def d_grad(dt, cm, est, rnd=lambda n, t: t, grnd=None):
    sd = {k: (v.to(dt).clone().requires_grad_(v.is_floating_point() and 'weight_u' not in k and 'weight_v' not in k) if v.is_floating_point() else v)
          for k, v in formula.formula_state('discriminator').items()}
    cm, est = cm.to(dt), est.to(dt)
    SN.clear()
    with torch.no_grad():
        d_forward(sd, cm, est, rnd, 'gen'); sd.update({k: v.clone() for k, v in SN.items()})
    dgx = d_forward(sd, cm, est, rnd, 'gx'); sd.update({k: v.clone() for k, v in SN.items()})
    dyy = d_forward(sd, cm, cm, rnd, 'yy')
    L_C = Or._mse(dyy.flatten(), torch.ones(2, dtype=dt)); L_E = Or._mse(dgx.flatten(), q.to(dt))
    (L_C + L_E).backward()
    return sd[NAME].grad.double().numpy(), dyy.detach().double().flatten().numpy(), dgx.detach().double().flatten().numpy(), float(L_C), float(L_E)

t0 = time.time(); cm64, est64 = gen_est(torch.float64); print('gen fp64 %.1fs' % (time.time() - t0))
t0 = time.time(); cm32, est32 = gen_est(torch.float32); print('gen fp32 %.1fs' % (time.time() - t0))
print('est_mag fp32 vs fp64 rms rel: %.3e' % float(((est32.double() - est64) ** 2).mean().sqrt() / (est64 ** 2).mean().sqrt()))
gA, dyy, dgx, LC, LE = d_grad(torch.float64, cm64, est64)
print('A fp64 restatement vs golden: %.3e  D(c,c)=%s D(c,est)=%s residuals yy %s gx %s L_C %.6f L_E %.6f' % (rel(gA), dyy, dgx, dyy - 1, dgx - q.numpy(), LC, LE))
argA = {k: v.clone() for k, v in ARG.items()}
gA, *_ = d_grad(torch.float64, cm64, est64); argA = {k: v.clone() for k, v in ARG.items()}; sgnA = dict(SGN); topA = {k: v.clone() for k, v in TOP.items()}
print('A again %.3e' % rel(gA))
gap = (topA['gx'][..., 0] - topA['gx'][..., 1]) / topA['gx'][..., 0].abs()
print('smallest relative gaps between the two largest pixels of a stage-4 channel (gx forward):', np.sort(gap.flatten().numpy())[:6])
d = est32.double() - est64
for sc in (1.0, 0.5, 0.25, 0.125, -1.0):
    gP, *_ = d_grad(torch.float64, cm64, est64 + sc * d)
    flips = int((ARG['gx'] != argA['gx']).sum())
    print('fp64 D on est64 + %.3f (est32 - est64): grad change vs A %.3e, argmax flips %d' % (sc, float(np.sqrt(np.mean((gP - gA) ** 2)) / nrm), flips))
    print('   PReLU sign flips per (forward, stage):', {k: int((SGN[k] != sgnA[k]).sum()) for k in SGN if int((SGN[k] != sgnA[k]).sum())})
torch.manual_seed(0)
for sc in (1.0, 0.1):
    pert = est64 * (1 + sc * 1.4e-5 * torch.randn_like(est64))
    gP, *_ = d_grad(torch.float64, cm64, pert)
    print('fp64 D on est64 * (1 + %.1e N(0,1)): grad change %.3e, flips %d' % (sc * 1.4e-5, float(np.sqrt(np.mean((gP - gA) ** 2)) / nrm), int((ARG['gx'] != argA['gx']).sum())))
