"""TSC-diffusion hybrid: the reference's `models/tsc_diffusion.py` TSCNet -- the CMGAN generator with a second
DenseEncoder for the noisy conditioner and a `MergeBlock` (diffusion-step embedding + gated 1x1 convolutions) in front of every TSCB
-- and `inference_diffuse.predict_tsc` (:231-269), on the HIP kernels of the generator.  Inference only (like the CDiffuSE path):
eval-mode BatchNorm, no dropout, no backward.

state_dict names are the reference's (`dense_encoder_noisy.*`, `merge_block.diffusion_embedding.projection1.weight`,
`merge_block.merge_diffusion.weight [128, 64, 1, 1]`, ...), so its checkpoints load with `load_state_dict`.
"""
import math

import numpy as np
import torch
import torch.nn as nn

from . import _lib as L
from . import frontend as FE
from . import gemm as GM
from . import layers as LY
from . import ops as O
from .diffuse import _DiffusionEmbedding
from .generator import TSCNet, _dilated_dense


class TSCNetDiffusion(TSCNet):
    """models/tsc_diffusion.py:43-90.  forward(x complex [B, F, T], noisy_spec complex [B, F, T], diffusion_step [1] or [B])
    -> (real, imag) [B, 1, T, F]."""

    def __init__(self, num_channel=64, num_features=201, noise_schedule=None):
        super().__init__(num_channel, num_features)
        ch = num_channel
        enc = nn.Module()
        enc.conv_1 = nn.Sequential(nn.Conv2d(3, ch, (1, 1)), nn.InstanceNorm2d(ch, affine=True), nn.PReLU(ch))
        enc.dilated_dense = _dilated_dense(ch)
        enc.conv_2 = nn.Sequential(nn.Conv2d(ch, ch, (1, 3), (1, 2), padding=(0, 1)), nn.InstanceNorm2d(ch, affine=True), nn.PReLU(ch))
        self.dense_encoder_noisy = enc
        mb = nn.Module()
        mb.diffusion_embedding = _DiffusionEmbedding(len(noise_schedule))
        mb.diffusion_projection = nn.Linear(512, ch)
        mb.merge_diffusion = nn.Conv2d(ch, ch * 2, 1)
        mb.conditioner_projection = nn.Conv2d(ch, ch * 2, 1)
        mb.output_residual = nn.Conv2d(ch, ch, 1)
        self.merge_block = mb
        self._pnames = [k for k, _ in self.named_parameters()]

    def _merge(self, P, x, cond, step, B):
        """MergeBlock.forward (tsc_diffusion.py:27-40) on tokens [M, 64]: y = W_m (x + d) + b_m + W_c cond + b_c;
        out = (x + W_r (sigmoid(gate) tanh(filter)) + b_r) / sqrt(2).  d = the projected step embedding: with ONE step for the
        whole batch (predict_tsc) it folds into the bias (W_m d + b_m, weight-sized); per-clip steps add d to x first."""
        M = x.shape[0]
        mb = self.merge_block
        d = mb.diffusion_projection(mb.diffusion_embedding(step))                      # [N, 64], host-sized plumbing
        Wm, Wc = P['merge_block.merge_diffusion.weight'].view(128, 64), P['merge_block.conditioner_projection.weight'].view(128, 64)
        Wr = P['merge_block.output_residual.weight'].view(64, 64)
        if d.shape[0] == 1:
            bias = (P['merge_block.merge_diffusion.bias'] + (Wm @ d[0])).contiguous()
            xin = x
        else:
            if d.shape[0] != B:
                raise L.SeHipError(f'TSCNetDiffusion: diffusion_step must have 1 or {B} entries (got {d.shape[0]})')
            bias = P['merge_block.merge_diffusion.bias']
            xin = (x.view(B, -1, 64) + d[:, None, :]).reshape(M, 64)
        y = torch.empty(M, 128, device=x.device, dtype=torch.float32)
        GM.gemm_tap(GM.linear_desc(M, 64, 128, epilogue=L.EPI_BIAS, precision=2), xin, Wm.contiguous(), y, bias=bias)
        GM.gemm_tap(GM.linear_desc(M, 64, 128, epilogue=L.EPI_BIAS | L.EPI_ACCUM, precision=0), cond, Wc.contiguous(), y,
                    bias=P['merge_block.conditioner_projection.bias'])
        g = O.gate_tanh(y, M, 64)
        res = torch.empty(M, 64, device=x.device, dtype=torch.float32)
        GM.gemm_tap(GM.linear_desc(M, 64, 64, epilogue=L.EPI_BIAS), g, Wr.contiguous(), res, bias=P['merge_block.output_residual.bias'])
        r2 = 1.0 / math.sqrt(2.0)
        return O.axpbypcz(x, res, res, r2, r2, 0.0)

    @torch.no_grad()
    def forward_planes(self, xin, nin, diffusion_step):
        """xin / nin: planes [B, T, F, 4] of the current and of the conditioning spectrum -> est planes [B, T, F, 4]"""
        if self.training:
            raise L.SeHipError('TSCNetDiffusion is an inference path (eval mode): call .eval() first')
        P = dict(self.named_parameters())
        P.update(dict(self.named_buffers()))
        P['__prep__'] = self._prepare_weights(P, xin.device)
        B, T, Fq, _ = xin.shape
        step = torch.as_tensor(diffusion_step, device=xin.device)
        x, _ = LY.encoder_fwd(P, xin.contiguous(), B, T, Fq)
        xn, _ = LY.encoder_fwd(P, nin.contiguous(), B, T, Fq, p='dense_encoder_noisy')
        Fp = x.shape[2]
        tok, cond = x.view(B * T * Fp, 64), xn.view(B * T * Fp, 64)
        for i in range(1, 5):
            tok = self._merge(P, tok, cond, step, B)
            tok, _ = LY.conformer_fwd(P, f'TSCB_{i}.time_conformer', tok, B, T, Fp, 'time', False)
            tok, _ = LY.conformer_fwd(P, f'TSCB_{i}.freq_conformer', tok, B, T, Fp, 'freq', False)
        cplx, _ = LY.complex_decoder_fwd(P, tok, B, T, Fp)
        mask, _ = LY.mask_decoder_fwd(P, tok, B, T, Fp)
        return O.assemble(mask, 1, xin, cplx)

    def forward(self, x, noisy_spec, diffusion_step=None):
        est = self.forward_planes(FE.spec_to_planes(x), FE.spec_to_planes(noisy_spec), diffusion_step)
        return est[..., 1].unsqueeze(1), est[..., 2].unsqueeze(1)


@torch.no_grad()
def predict_tsc(model, args, config, noisy_signal, alpha, beta, alpha_cum, sigmas, T, c1, c2, c3, delta, delta_bar,
                device=torch.device('cuda'), noises=None):
    """inference_diffuse.py:231-269: the supportive reverse process in the compressed-STFT domain -- every step re-analyses the
    current audio, runs the hybrid generator conditioned on the noisy spectrum and the step, re-synthesises the predicted noise.
    `noises` (optional, [steps - 1, 1, padded length]) replaces torch.randn_like for reproducible parity runs."""
    noisy = torch.as_tensor(np.asarray(noisy_signal), dtype=torch.float32, device=device).unsqueeze(0)
    hop, n_fft = config.HOP_SAMPLES, config.N_FFT
    comp = getattr(args, 'comp_type', 'pow')
    c = O.clip_scale(noisy.contiguous())
    length = noisy.size(-1)
    padding_len = int(np.ceil(length / hop)) * hop - length
    noisy = noisy * c[:, None]
    noisy = torch.cat([noisy, noisy[:, :padding_len]], dim=-1).contiguous()
    audio = noisy_audio = noisy
    orig_planes, _ = FE.stft_planes(noisy, n_fft, hop, comp, padded=False)
    if noises is not None:
        noises = torch.as_tensor(np.asarray(noises), dtype=torch.float32, device=device)
    gamma = [0.2]
    k = 0
    for n in range(len(alpha) - 1, -1, -1):
        planes, _ = FE.stft_planes(audio.contiguous(), n_fft, hop, comp, padded=False)
        est = model.forward_planes(planes, orig_planes, torch.tensor([float(T[n])], device=device))
        predicted_noise = FE.istft_planes(est, n_fft, hop, comp)
        if n > 0:
            audio = float(c1[n]) * audio + float(c2[n]) * noisy_audio - float(c3[n]) * predicted_noise
            noise = torch.randn_like(audio) if noises is None else noises[k]
            k += 1
            audio = audio + float(delta_bar[n]) ** 0.5 * noise
        else:
            audio = float(c1[n]) * audio - float(c3[n]) * predicted_noise
            audio = (1 - gamma[n]) * audio + gamma[n] * noisy_audio
    audio = audio / c[:, None]
    return torch.flatten(audio)[:length].cpu().numpy()
