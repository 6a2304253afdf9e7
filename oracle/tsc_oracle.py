"""CPU restatement (torch, fp32 / fp64) of the TSC-diffusion hybrid -- TEST INFRASTRUCTURE ONLY: imported by tests/ alone, never by
the product.  Follows `/root/reference/models/tsc_diffusion.py` (MergeBlock :15-40, TSCNet :43-90) and
`inference_diffuse.py:231-269` (predict_tsc); the shared pieces come from se_oracle.py / diffuse_oracle.py.  Pinned against vectors
produced by importing the reference (tests/golden/make_golden_tsc.py -> golden_tsc.npz)."""
import numpy as np
import torch
import torch.nn.functional as F

from . import diffuse_oracle as DO
from . import se_oracle as SO


def merge_block(sd, x, cond, step, p='merge_block'):
    """tsc_diffusion.py:27-40.  x, cond: [B, C, T, F]; step: [N] int or float (N = 1 broadcasts over the batch)."""
    emb = DO.diffusion_embedding({k[len(p) + 1:]: v for k, v in sd.items() if k.startswith(p + '.diffusion_embedding')}, step)
    d = F.linear(emb, sd[f'{p}.diffusion_projection.weight'], sd[f'{p}.diffusion_projection.bias'])[:, :, None, None]
    c = F.conv2d(cond, sd[f'{p}.conditioner_projection.weight'], sd[f'{p}.conditioner_projection.bias'])
    y = F.conv2d(x + d, sd[f'{p}.merge_diffusion.weight'], sd[f'{p}.merge_diffusion.bias']) + c
    gate, filt = torch.chunk(y, 2, dim=1)
    y = torch.sigmoid(gate) * torch.tanh(filt)
    res = F.conv2d(y, sd[f'{p}.output_residual.weight'], sd[f'{p}.output_residual.bias'])
    return (x + res) / np.sqrt(2.0)


def _inputs(spec):
    re = spec.real.transpose(1, 2)[:, None]
    im = spec.imag.transpose(1, 2)[:, None]
    mag = torch.sqrt(re * re + im * im)
    return mag, torch.atan2(im, re), torch.cat([mag, re, im], dim=1)


def forward(sd, x_spec, noisy_spec, step, train=False):
    """TSCNet.forward of tsc_diffusion.py:58-90 (eval mode: BatchNorm running statistics)."""
    mag, ph, x_in = _inputs(x_spec)
    _, _, n_in = _inputs(noisy_spec)
    out = SO.dense_encoder(sd, 'dense_encoder', x_in)
    out_noisy = SO.dense_encoder(sd, 'dense_encoder_noisy', n_in)
    for i in range(1, 5):
        out = SO.tscb(sd, f'TSCB_{i}', merge_block(sd, out, out_noisy, step), train, None)
    mask = SO.mask_decoder(sd, 'mask_decoder', out)
    cplx = SO.complex_decoder(sd, 'complex_decoder', out)
    out_mag = mask * mag
    return out_mag * torch.cos(ph) + cplx[:, 0:1], out_mag * torch.sin(ph) + cplx[:, 1:2]


def predict_tsc(sd, noisy_signal, sched, noises, comp='pow'):
    """inference_diffuse.py:231-269; sched: dict from diffuse_oracle.inference_schedule; noises: [steps - 1, 1, L_padded] draws."""
    x = torch.as_tensor(np.asarray(noisy_signal), dtype=torch.float32)[None]
    c = torch.sqrt(x.shape[-1] / torch.sum(x ** 2, -1))
    x = x * c[:, None]
    length = x.shape[-1]
    pad = -length % 100
    x = torch.cat([x, x[:, :pad]], -1)
    audio = noisy_audio = x
    orig = SO.compressed_stft(x, comp=comp)
    alpha, T, c1, c2, c3, delta_bar = (sched[k] for k in ('alpha', 'T', 'c1', 'c2', 'c3', 'delta_bar'))
    k = 0
    for n in range(len(alpha) - 1, -1, -1):
        er, ei = forward(sd, SO.compressed_stft(audio, comp=comp), orig, torch.tensor([float(T[n])]))
        pred = SO.uncompressed_istft(torch.complex(er, ei).squeeze(1).permute(0, 2, 1), comp=comp)
        if n > 0:
            audio = float(c1[n]) * audio + float(c2[n]) * noisy_audio - float(c3[n]) * pred
            audio = audio + float(delta_bar[n]) ** 0.5 * torch.as_tensor(noises[k])
            k += 1
        else:
            audio = float(c1[n]) * audio - float(c3[n]) * pred
            audio = 0.8 * audio + 0.2 * noisy_audio
    return (audio / c[:, None]).flatten()[:length]


def add_noise(audio, noisy, noise_schedule, t, noise):
    """core/function.py:25-44 with the draws supplied (t [N] int64, noise like audio)"""
    beta = np.array(noise_schedule)
    noise_level = torch.tensor(np.cumprod(1 - beta).astype(np.float32))
    ns = noise_level[t].unsqueeze(1)
    m = (((1 - noise_level[t]) / noise_level[t] ** 0.5) ** 0.5).unsqueeze(1)
    tail = (1.0 - (1 + m ** 2) * ns) ** 0.5 * noise
    noisy_audio = (1 - m) * ns ** 0.5 * audio + m * ns ** 0.5 * noisy + tail
    combine_noise = (m * ns ** 0.5 * (noisy - audio) + tail) / (1 - ns) ** 0.5
    return noisy_audio, combine_noise


def train_loss(sd, clean, noisy, noise_schedule, t, noise, comp='pow', train=True):
    """the loss of one train_tsc_diffusion iteration (core/function.py:472-505) as a differentiable function of the state dict `sd`
    (train mode: BatchNorm batch statistics; dropout 0 like the goldens): normalize_batch, add_noise, two compressed STFTs, the
    hybrid generator, iSTFT, mean |predicted - combine_noise|"""
    c = torch.sqrt(noisy.shape[-1] / torch.sum(noisy ** 2, -1))
    clean, noisy = clean * c[:, None], noisy * c[:, None]
    noisy_audio, combine_noise = add_noise(clean, noisy, noise_schedule, t, noise)
    orig = SO.compressed_stft(noisy, comp=comp)
    nz = SO.compressed_stft(noisy_audio, comp=comp)
    er, ei = forward(sd, nz, orig, t, train=train)      # train=False: validate_tsc_diffusion (eval mode: running statistics)
    pred = SO.uncompressed_istft(torch.complex(er, ei).squeeze(1).permute(0, 2, 1), comp=comp)
    return torch.mean(torch.abs(pred - combine_noise))
