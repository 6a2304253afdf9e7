"""TSC-diffusion hybrid: the reference's `models/tsc_diffusion.py` TSCNet -- the CMGAN generator with a second
DenseEncoder for the noisy conditioner and a `MergeBlock` (diffusion-step embedding + gated 1x1 convolutions) in front of every TSCB
-- and `inference_diffuse.predict_tsc` (:231-269), on the HIP kernels of the generator.  Round 3 adds the training step of the
hybrid (`core/function.py:25-44` add_noise, `:453-532` train_tsc_diffusion): `tsc_diffusion_step` = add_noise -> two compressed
STFTs -> the generator in train mode (hand-written backward: both encoders, the four MergeBlock applications, TSCBs, decoders) ->
iSTFT -> L1 against the combined noise -> optimizer step.  fp32 arithmetic (the reference wraps this loop in fp16 autocast +
GradScaler: a precision policy of its CUDA run, not part of the function being computed).

state_dict names are the reference's (`dense_encoder_noisy.*`, `merge_block.diffusion_embedding.projection1.weight`,
`merge_block.merge_diffusion.weight [128, 64, 1, 1]`, ...), so its checkpoints load with `load_state_dict`.
"""
import math

import numpy as np
import torch
import torch.nn as nn

from . import _lib as L
from . import frontend as FE
from . import losses as LS
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


class _TSCDiffFn(torch.autograd.Function):
    """the hybrid generator as one autograd node: forward / backward are the hand-written layer functions of layers.py plus the
    MergeBlock of this file; `d` [B, 64] is the projected diffusion-step embedding (plain torch autograd produces it and receives
    its gradient)."""

    @staticmethod
    def forward(ctx, model, xin, nin, d, *params):
        P = dict(zip(model._pnames, params))
        P.update(model._buffer_dict())
        with torch.no_grad():
            P['__prep__'] = model._prepare_weights(P, xin.device)
            model._drop_calls += 1
            seed = (torch.initial_seed() * 2654435761 + model._drop_calls * 40503) & 0xFFFFFFFF
            est, c = model._train_fwd(P, xin.contiguous(), nin.contiguous(), d.detach(), seed)
        ctx.c, ctx.P, ctx.model = c, P, model
        return est

    @staticmethod
    def backward(ctx, dest):
        model, P = ctx.model, ctx.P
        direct = all(P[k].grad is not None and P[k].grad.is_contiguous() for k in model._pnames)
        with torch.no_grad():
            G = {k: (P[k].grad if direct else torch.zeros_like(P[k])) for k in model._pnames}
            GM.leaf_begin()
            try:
                dd = model._train_bwd(P, G, ctx.c, dest.contiguous())
            finally:
                GM.leaf_join(dest.device)
        ctx.c = None
        if direct:
            return (None, None, None, dd) + (None,) * len(model._pnames)
        return (None, None, None, dd) + tuple(G[k] for k in model._pnames)


def _merge_fwd(P, x, cond, d, B):
    """MergeBlock forward on tokens, keeping what the backward needs (tsc_diffusion.py:27-40)."""
    M = x.shape[0]
    Wm, Wc = P['merge_block.merge_diffusion.weight'].view(128, 64), P['merge_block.conditioner_projection.weight'].view(128, 64)
    Wr = P['merge_block.output_residual.weight'].view(64, 64)
    xin = (x.view(B, -1, 64) + d[:, None, :]).reshape(M, 64)        # per-clip step embedding (plumbing-sized torch add)
    y = torch.empty(M, 128, device=x.device, dtype=torch.float32)
    GM.gemm_tap(GM.linear_desc(M, 64, 128, epilogue=L.EPI_BIAS, precision=2), xin, Wm.contiguous(), y,
                bias=P['merge_block.merge_diffusion.bias'])
    GM.gemm_tap(GM.linear_desc(M, 64, 128, epilogue=L.EPI_BIAS | L.EPI_ACCUM, precision=0), cond, Wc.contiguous(), y,
                bias=P['merge_block.conditioner_projection.bias'])
    g = O.gate_tanh(y, M, 64)
    res = torch.empty(M, 64, device=x.device, dtype=torch.float32)
    GM.gemm_tap(GM.linear_desc(M, 64, 64, epilogue=L.EPI_BIAS), g, Wr.contiguous(), res, bias=P['merge_block.output_residual.bias'])
    r2 = 1.0 / math.sqrt(2.0)
    return O.axpbypcz(x, res, res, r2, r2, 0.0), (xin, cond, y, g)


def _merge_bwd(P, G, ctx, dout, B, dcond):
    """gradients of one MergeBlock application: returns (dx, dd [B, 64]); accumulates into dcond [M, 64] and the shared weights."""
    xin, cond, y, g = ctx
    M = dout.shape[0]
    dev = dout.device
    Wm, Wc = P['merge_block.merge_diffusion.weight'].view(128, 64), P['merge_block.conditioner_projection.weight'].view(128, 64)
    Wr = P['merge_block.output_residual.weight'].view(64, 64)
    r2 = 1.0 / math.sqrt(2.0)
    dres = O.axpbypcz(dout, dout, dout, r2, 0.0, 0.0)                # d out / d (x + res) = 1 / sqrt(2): dres = dx_direct
    dg = torch.empty(M, 64, device=dev, dtype=torch.float32)
    GM.gemm_tap(GM.linear_desc(M, 64, 64), dres, Wr.t().contiguous(), dg)                 # res = g Wr^T  ->  dg = dres Wr
    dy = O.gate_tanh_bwd(y, dg, M, 64)
    with GM.leaf_stream(g, dres, xin, cond, dy):
        GM.gemm_tap_wgrad(GM.linear_desc(M, 64, 64), g, dres, G['merge_block.output_residual.weight'].view(64, 64),
                          G['merge_block.output_residual.bias'])
        GM.gemm_tap_wgrad(GM.linear_desc(M, 64, 128), xin, dy, G['merge_block.merge_diffusion.weight'].view(128, 64),
                          G['merge_block.merge_diffusion.bias'])
        GM.gemm_tap_wgrad(GM.linear_desc(M, 64, 128), cond, dy, G['merge_block.conditioner_projection.weight'].view(128, 64),
                          G['merge_block.conditioner_projection.bias'])
    dx = torch.empty(M, 64, device=dev, dtype=torch.float32)
    GM.gemm_tap(GM.linear_desc(M, 128, 64, epilogue=L.EPI_RESID, alpha=1.0, ldr=64), dy, Wm.t().contiguous(), dx, R=dres)
    GM.gemm_tap(GM.linear_desc(M, 128, 64, epilogue=L.EPI_ACCUM), dy, Wc.t().contiguous(), dcond)
    dd = dy.view(B, -1, 128).sum(1) @ Wm                            # d (x + d): the per-clip column sums of dxin = dy Wm
    return dx, dd


def _train_fwd(self, P, xin, nin, d, seed):
    B, T, Fq, _ = xin.shape
    train = self.training
    buffers = self._buffer_dict() if train else None
    drop = (self.ff_dropout, self.attn_dropout)
    ctx = {'xin': xin, 'dims': (B, T, Fq)}
    x, ctx['enc'] = LY.encoder_fwd(P, xin, B, T, Fq)
    xn, ctx['encn'] = LY.encoder_fwd(P, nin, B, T, Fq, p='dense_encoder_noisy')
    Fp = x.shape[2]
    ctx['Fp'] = Fp
    tok, cond = x.view(B * T * Fp, 64), xn.view(B * T * Fp, 64)
    if d.shape[0] == 1:
        d = d.expand(B, 64)
    ctx['tscb'], ctx['merge'] = [], []
    for i in range(1, 5):
        tok, cm = _merge_fwd(P, tok, cond, d, B)
        tok, c1 = LY.conformer_fwd(P, f'TSCB_{i}.time_conformer', tok, B, T, Fp, 'time', train, self.dp, buffers, drop,
                                   LY.site_seed(seed, 100 + 2 * i))
        c1.pop('out_stats', None)
        tok, c2 = LY.conformer_fwd(P, f'TSCB_{i}.freq_conformer', tok, B, T, Fp, 'freq', train, self.dp, buffers, drop,
                                   LY.site_seed(seed, 101 + 2 * i))
        c2.pop('out_stats', None)
        ctx['tscb'].append((c1, c2))
        ctx['merge'].append(cm)
    cplx, ctx['cplx'] = LY.complex_decoder_fwd(P, tok, B, T, Fp)
    mask, ctx['mask'] = LY.mask_decoder_fwd(P, tok, B, T, Fp)
    est = O.assemble(mask, 1, xin, cplx)
    ctx['est'] = est
    return est, ctx


def _train_bwd(self, P, G, ctx, dest):
    B, T, Fq = ctx['dims']
    Fp = ctx['Fp']
    dev = dest.device
    dmask = torch.empty(B * T * Fq, device=dev, dtype=torch.float32)
    dcplx = torch.empty(B, T, Fq, 4, device=dev, dtype=torch.float32)
    O.assemble_bwd(ctx['est'], dest, ctx['xin'], dmask, 1, dcplx)
    dsk_c = LY.complex_decoder_bwd(P, G, ctx['cplx'], dcplx, B, T, Fp)
    dsk_m = LY.mask_decoder_bwd(P, G, ctx['mask'], dmask, B, T, Fp)
    dtok = (dsk_c[..., :64] + dsk_m[..., :64]).reshape(B * T * Fp, 64)
    del dsk_c, dsk_m, dcplx
    dcond = O.zeros(B * T * Fp, 64, device=dev)
    dd = None
    for i in (4, 3, 2, 1):
        c1, c2 = ctx['tscb'][i - 1]
        dtok = LY.conformer_bwd(P, G, f'TSCB_{i}.freq_conformer', c2, dtok, B, T, Fp, self.dp)
        dtok = LY.conformer_bwd(P, G, f'TSCB_{i}.time_conformer', c1, dtok, B, T, Fp, self.dp)
        dtok, ddi = _merge_bwd(P, G, ctx['merge'][i - 1], dtok, B, dcond)
        dd = ddi if dd is None else dd + ddi
        ctx['tscb'][i - 1] = ctx['merge'][i - 1] = None
    LY.encoder_bwd(P, G, ctx['enc'], dtok.view(B, T, Fp, 64), B, T, Fq)
    LY.encoder_bwd(P, G, ctx['encn'], dcond.view(B, T, Fp, 64), B, T, Fq, p='dense_encoder_noisy')
    return dd


TSCNetDiffusion._train_fwd = _train_fwd
TSCNetDiffusion._train_bwd = _train_bwd


def step_embedding(model, t):
    """MergeBlock's projected diffusion-step embedding d [N, 64] (tsc_diffusion.py:28-29): plain torch (autograd) on [N, 512]"""
    mb = model.merge_block
    return mb.diffusion_projection(mb.diffusion_embedding(t))


def add_noise(audio, noisy, noise_schedule, t=None, noise=None):
    """core/function.py:25-44: (noisy_audio, combine_noise, t) of the supportive forward process; t / noise may be supplied
    (reproducible parity runs), otherwise drawn like the reference ([N] uniform steps, standard normal noise)."""
    N, _ = audio.shape
    beta = np.array(noise_schedule)
    noise_level = torch.tensor(np.cumprod(1 - beta).astype(np.float32), device=audio.device)
    if t is None:
        t = torch.randint(0, len(noise_schedule), [N], device=audio.device)
    noise_scale = noise_level[t].unsqueeze(1)
    noise_scale_sqrt = noise_scale ** 0.5
    m = (((1 - noise_level[t]) / noise_level[t] ** 0.5) ** 0.5).unsqueeze(1)
    if noise is None:
        noise = torch.randn_like(audio)
    noisy_audio = (1 - m) * noise_scale_sqrt * audio + m * noise_scale_sqrt * noisy + (1.0 - (1 + m ** 2) * noise_scale) ** 0.5 * noise
    combine_noise = (m * noise_scale_sqrt * (noisy - audio) + (1.0 - (1 + m ** 2) * noise_scale) ** 0.5 * noise) / (1 - noise_scale) ** 0.5
    return noisy_audio, combine_noise, t


def tsc_diffusion_step(model, optimizer, clean, noisy, noise_schedule, n_fft=400, hop=100, comp_type='pow', max_norm=0.0,
                       t=None, noise=None, step=True):
    """One iteration of train_tsc_diffusion (core/function.py:472-525): normalize_batch, add_noise, compressed STFT of the noisy
    clip (conditioner) and of the noised clip, the hybrid generator, iSTFT, loss = mean |predicted - combine_noise|, backward,
    optional global-norm clipping, optimizer step.  Returns the loss (0-d tensor)."""
    from . import train as TR
    c = O.clip_scale(noisy.contiguous())
    clean, noisy = clean * c[:, None], noisy * c[:, None]
    noisy_audio, combine_noise, t = add_noise(clean, noisy, noise_schedule, t, noise)
    orig_pl, _ = FE.stft_planes(noisy.contiguous(), n_fft, hop, comp_type, padded=False)
    nz_pl, _ = FE.stft_planes(noisy_audio.contiguous(), n_fft, hop, comp_type, padded=False)
    d = step_embedding(model, t)
    params = [p for _, p in model.named_parameters()]
    O.ARENA.begin(clean.device)
    try:
        est = _TSCDiffFn.apply(model, nz_pl, orig_pl, d, *params)
        predicted = FE.istft_planes(est, n_fft, hop, comp_type)
        loss = LS.l1_time_loss(predicted, combine_noise.contiguous())
        optimizer.zero_grad()
        loss.backward()
        if max_norm != 0.0:
            TR.clip_grad_norm(optimizer, params, max_norm)
        if step:
            optimizer.step()
    finally:
        O.ARENA.end()
    return loss.detach()


@torch.no_grad()
def tsc_diffusion_validation_loss(model, clean, noisy, noise_schedule, n_fft=400, hop=100, comp_type='pow', t=None, noise=None):
    """One iteration of validate_tsc_diffusion (core/function.py:566-597): the same pipeline as the training step in eval mode
    (BatchNorm running statistics, no dropout), no backward.  Returns the loss (0-d tensor)."""
    if model.training:
        raise L.SeHipError('tsc_diffusion_validation_loss: call model.eval() first (the reference validates in eval mode)')
    c = O.clip_scale(noisy.contiguous())
    clean, noisy = clean * c[:, None], noisy * c[:, None]
    noisy_audio, combine_noise, t = add_noise(clean, noisy, noise_schedule, t, noise)
    orig_pl, _ = FE.stft_planes(noisy.contiguous(), n_fft, hop, comp_type, padded=False)
    nz_pl, _ = FE.stft_planes(noisy_audio.contiguous(), n_fft, hop, comp_type, padded=False)
    est = model.forward_planes(nz_pl, orig_pl, t)
    predicted = FE.istft_planes(est, n_fft, hop, comp_type)
    return LS.l1_time_loss(predicted, combine_noise.contiguous()).detach()


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
