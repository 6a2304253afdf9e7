"""CDiffuSE on the HIP path: the DiffWave-style conditional denoiser of models/DiffuSE.py and the supportive reverse
diffusion of inference_diffuse.py:117-228 (BASELINE config 5: 50-step sampling, batch 32).  Inference only.

The reference's diffusion path does not run as written: `predict` / `train` hand the COMPLEX torch.stft output to
DiffuSE.forward, whose SpectrogramUpsampler is a real ConvTranspose2d.  The fix defined here
(and used to generate the parity goldens from the reference itself, tests/golden/make_golden_diffuse.py): the conditioner is
the MAGNITUDE |STFT| (a complex input is reduced with abs()), the audio has 100 * T samples (clip zero-padded by one hop).

MI355X-first layout: feature maps are channels-last [B, L, C] (L = 100 T samples); every contraction is a tap-GEMM (the
dilated k = 3 convs as 3 taps at (0, -d), (0, 0), (0, +d); the 1x1 convs as row GEMMs; residual and skip projections share one
GEMM), everything between them is five fused streaming kernels (csrc/se_diffuse.hip).  87 % of the reference's FLOPs per
step are the 30 conditioner projections (201 -> 128 channels at every sample) of a spectrogram that does NOT change during
the 50 reverse steps: they are computed ONCE per utterance batch and kept in HBM (30 x B x L x 128 fp32 = 15.8 GB at batch 32
of 2 s clips -- what 288 GB are for); a step then costs 7.6 instead of 56.7 GFLOP per clip."""
import ctypes as C
import math
import os

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib as L
from . import frontend as FE
from . import gemm as GM
from . import layers as LY
from .weights import WeightPlan

# SE_DIFF_GATE_FUSED=0: the stand-alone gate kernel in front of the projection GEMM
GATE_FUSED = os.environ.get('SE_DIFF_GATE_FUSED', '1') != '0'
# SE_DIFF_ONE_STREAM=0: x and y = x + d_step both stored between the layers (mix: 7 plane passes instead of 6)
ONE_STREAM = os.environ.get('SE_DIFF_ONE_STREAM', '1') != '0'

_i, _l, _f, _d = C.c_int, C.c_long, C.c_float, C.c_double


class _DiffusionEmbedding(nn.Module):
    def __init__(self, max_steps):
        super().__init__()
        steps = torch.arange(max_steps).unsqueeze(1)
        dims = torch.arange(64).unsqueeze(0)
        table = steps * 10.0 ** (dims * 4.0 / 63.0)
        self.register_buffer('embedding', torch.cat([torch.sin(table), torch.cos(table)], dim=1), persistent=False)
        self.projection1 = nn.Linear(128, 512)
        self.projection2 = nn.Linear(512, 512)

    def forward(self, step):
        """models/DiffuSE.py:45-62 ([N] int or float steps -> [N, 512]); [N, 128]-sized host-side plumbing"""
        if step.dtype in (torch.int32, torch.int64):
            x = self.embedding[step]
        else:
            lo, hi = torch.floor(step).long(), torch.ceil(step).long()
            x = self.embedding[lo] + (self.embedding[hi] - self.embedding[lo]) * (step - lo).unsqueeze(-1)
        return F.silu(self.projection2(F.silu(self.projection1(x))))


class _Upsampler(nn.Module):
    def __init__(self, hop_length):
        super().__init__()
        Ls = int(np.sqrt(hop_length))
        assert Ls ** 2 == 100, 'Hop length must be a squre number'
        self.conv1 = nn.ConvTranspose2d(1, 1, [3, 2 * Ls], stride=[1, Ls], padding=[1, Ls // 2])
        self.conv2 = nn.ConvTranspose2d(1, 1, [3, 2 * Ls], stride=[1, Ls], padding=[1, Ls // 2])


class _ResidualBlock(nn.Module):
    def __init__(self, n_specs, ch, dilation):
        super().__init__()
        self.dilation = dilation
        self.dilated_conv = nn.Sequential(nn.Conv1d(ch, 2 * ch, 3, padding=dilation, dilation=dilation),
                                          nn.GroupNorm(2 * ch // 16, 2 * ch))
        self.diffusion_projection = nn.Linear(512, ch)
        self.conditioner_projection = nn.Conv1d(n_specs, 2 * ch, 1)
        self.output_projection = nn.Sequential(nn.Conv1d(ch, ch, 1), nn.GroupNorm(ch // 16, ch))
        self.output_residual = nn.Conv1d(ch, ch, 1)


class DiffuSE(nn.Module):
    """DiffuSE(dilation_cycle_length, hop_length, n_specs, noise_schedule, residual_channels, residual_layers) with the
    reference's constructor, state_dict names and forward(audio, spectrogram, diffusion_step) -> [B, 1, L]."""

    def __init__(self, dilation_cycle_length, hop_length, n_specs, noise_schedule, residual_channels, residual_layers):
        super().__init__()
        if residual_channels % 64 != 0:
            raise ValueError('the HIP path is built for residual_channels that are multiples of 64')
        self.C, self.n_specs, self.hop = residual_channels, n_specs, hop_length
        self.input_projection = nn.Conv1d(1, residual_channels, 1)
        self.diffusion_embedding = _DiffusionEmbedding(len(noise_schedule))
        self.spectrogram_upsampler = _Upsampler(hop_length)
        self.residual_layers = nn.ModuleList([_ResidualBlock(n_specs, residual_channels, 2 ** (i % dilation_cycle_length))
                                              for i in range(residual_layers)])
        self.skip_projection = nn.Conv1d(residual_channels, residual_channels, 1)
        self.output_projection = nn.Conv1d(residual_channels, 1, 1)
        for m in self.modules():
            if isinstance(m, nn.Conv1d):
                nn.init.kaiming_normal_(m.weight)
        nn.init.zeros_(self.output_projection.weight)
        self._packed = None

    # ---- weight packing (once per weight version) ----------------------------------------------------------------
    def _pack(self):
        # data_ptr and device too: model.to(device) / p.data = ... swap the storage without bumping _version
        # ... and the arithmetic mode the planes are built for (tests flip it in-process: a stale pack would compare a mode with itself)
        mode = (LY.CONV_PRECISION, os.environ.get('SE_DIFFUSE_PRECISION', 'f16x3'))
        ver = (mode,) + tuple((p._version, p.data_ptr(), str(p.device)) for p in self.parameters())
        if self._packed is not None and self._packed['ver'] == ver:
            return self._packed
        Cc, nl = self.C, len(self.residual_layers)
        K = (self.n_specs + 3) // 4 * 4
        pk = {'ver': ver, 'K': K, 'layers': []}
        # scaled split-fp16 planes of the two GEMMs of every residual layer (precision 3, as in the generator): the weights are
        # static at inference, prepared once per weight version in one launch
        f16 = LY.CONV_PRECISION == 3 and os.environ.get('SE_DIFFUSE_PRECISION', 'f16x3') == 'f16x3'
        plan = WeightPlan(next(self.parameters()).device) if f16 else None
        pk['plan'] = plan
        for li, blk in enumerate(self.residual_layers):
            wc = blk.conditioner_projection.weight.detach().reshape(2 * Cc, self.n_specs)
            wcp = torch.zeros(2 * Cc, K, device=wc.device)
            wcp[:, :self.n_specs] = wc
            pk['layers'].append({
                'wd': GM.pack_conv_fwd(blk.dilated_conv[0].weight.detach().unsqueeze(2).contiguous()),      # [2C][3][C]
                'bd': blk.dilated_conv[0].bias.detach().contiguous(),
                'wc': wcp.contiguous(), 'bc': blk.conditioner_projection.bias.detach().contiguous(),
                'w2': torch.cat([blk.output_residual.weight.detach().reshape(Cc, Cc),
                                 blk.output_projection[0].weight.detach().reshape(Cc, Cc)], 0).contiguous(),   # res | skip
                'b2': torch.cat([blk.output_residual.bias.detach(), blk.output_projection[0].bias.detach()]).contiguous(),
                'taps': [(0, -blk.dilation), (0, 0), (0, blk.dilation)]})
            if f16:
                lay = pk['layers'][-1]
                lay['wd16'] = plan.conv_fwd((li, 'wd'), blk.dilated_conv[0].weight.detach().unsqueeze(2), planes='f16')
                plan.linear((li, 'w2'), blk.output_residual.weight.detach().reshape(Cc, Cc), planes='f16', rows=2 * Cc)
                lay['w216'] = plan.linear((li, 'w2'), blk.output_projection[0].weight.detach().reshape(Cc, Cc), planes='f16', o_off=Cc)
        if f16:
            plan.run()
        pk['wdp'] = torch.cat([b.diffusion_projection.weight.detach() for b in self.residual_layers], 0).contiguous()  # [nl C, 512]
        pk['bdp'] = torch.cat([b.diffusion_projection.bias.detach() for b in self.residual_layers]).contiguous()
        pk['ws'] = (self.skip_projection.weight.detach().reshape(Cc, Cc) / math.sqrt(nl)).contiguous()
        self._packed = pk
        return pk

    # ---- the conditioner: computed once per spectrogram, reused by every reverse step ----------------------------------
    @torch.no_grad()
    def conditioner(self, spectrogram):
        """spectrogram [B, n_specs, T] (magnitude; complex is reduced with abs(): the input fix) -> list of the residual_layers
        conditioner projections, each [B, 100 T, 2C] channels-last"""
        if spectrogram.is_complex():
            spectrogram = spectrogram.abs()
        spec = spectrogram.float().contiguous()
        L.check_cuda(spec)
        pk = self._pack()
        B, Fs, T = spec.shape
        up, K, Cc = self.spectrogram_upsampler, pk['K'], self.C
        s1 = torch.empty(B, Fs, 10 * T, device=spec.device)
        L.call('se_diff_upsample', L.ptr(spec), L.ptr(up.conv1.weight.detach().reshape(-1).contiguous()), L.ptr(up.conv1.bias),
               L.ptr(s1), _i(B), _i(Fs), _i(T), _i(0), _i(0), L.stream())
        Lp = 100 * T
        s2 = torch.zeros(B, Lp, K, device=spec.device)
        L.call('se_diff_upsample', L.ptr(s1), L.ptr(up.conv2.weight.detach().reshape(-1).contiguous()), L.ptr(up.conv2.bias),
               L.ptr(s2), _i(B), _i(Fs), _i(10 * T), _i(1), _i(K), L.stream())
        out = []
        for lay in pk['layers']:
            cnd = torch.empty(B, Lp, 2 * Cc, device=spec.device)
            GM.gemm_tap(GM.linear_desc(B * Lp, K, 2 * Cc, epilogue=L.EPI_BIAS), s2, lay['wc'], cnd, bias=lay['bc'])
            out.append(cnd)
        return out

    @torch.no_grad()
    def denoise(self, audio, cond, diffusion_step):
        """one evaluation of the network given the cached conditioner: audio [B, L] -> predicted noise [B, L]"""
        L.check_cuda(audio)
        pk = self._pack()
        B, Lp = audio.shape
        Cc, nl, dev = self.C, len(self.residual_layers), audio.device
        if cond[0].shape[1] != Lp:
            raise L.SeHipError(f'DiffuSE: audio has {Lp} samples, the conditioner {cond[0].shape[1]} (= 100 T)')
        step = diffusion_step.to(dev)
        emb = self.diffusion_embedding(step)                                          # [N, 512], N in {1, B}
        dproj = (emb @ pk['wdp'].t() + pk['bdp']).view(emb.shape[0], nl, Cc).transpose(0, 1).contiguous()   # [nl][N][C]
        dB = emb.shape[0]
        # SE_DIFF_ONE_STREAM (default on): only y = x + d_step is kept between the layers (the mix recovers x = y - d_cur)
        x = None if ONE_STREAM else torch.empty(B, Lp, Cc, device=dev)
        y = torch.empty(B, Lp, Cc, device=dev)
        y2 = torch.empty(B, Lp, Cc, device=dev)
        skip = torch.empty(B, Lp, Cc, device=dev)
        R = torch.empty(B, Lp, 2 * Cc, device=dev)
        R2 = torch.empty(B, Lp, 2 * Cc, device=dev)
        ss = torch.empty(B, 2 * Cc, 2, device=dev)
        ss2 = torch.empty(B, Cc, 2, device=dev)
        audio = audio.float().contiguous()
        # scaled split-fp16: the producers of y (se_diff_input / se_diff_mix) raise max |y| per layer -- the operand scale of the
        # dilated conv; the gate output sigmoid * tanh lies in (-1, 1): static exponent 13
        f16 = pk['plan'] is not None
        yam = torch.zeros(nl, device=dev) if f16 else None
        L.call('se_diff_input_amax', L.ptr(audio), L.ptr(self.input_projection.weight.detach().reshape(-1).contiguous()),
               L.ptr(self.input_projection.bias), L.ptr(dproj[0]), _i(dB), L.ptr(x), L.ptr(y), _i(B), _l(Lp), _i(Cc),
               L.ptr(yam[0:1] if f16 else None), L.stream())
        st_all = torch.zeros(nl, 2, B, 2 * Cc, 2, device=dev, dtype=torch.float64)      # GroupNorm sums of all layers: ONE fill
        for i, (lay, blk) in enumerate(zip(pk['layers'], self.residual_layers)):
            st = st_all[i, 0]
            if f16:
                d = GM.make_desc(B, 1, Lp, 1, Lp, lay['taps'], Cc, Cc, 2 * Cc, 2 * Cc, epilogue=L.EPI_BIAS | L.EPI_STATS,
                                 precision=3, a_amax=yam[i:i + 1])
                GM.gemm_tap(d, y, lay['wd16'], R, bias=lay['bd'], stats=st)
            else:
                d = GM.make_desc(B, 1, Lp, 1, Lp, lay['taps'], Cc, Cc, 2 * Cc, 2 * Cc, epilogue=L.EPI_BIAS | L.EPI_STATS,
                                 precision=min(LY.CONV_PRECISION, 2))
                GM.gemm_tap(d, y, lay['wd'], R, bias=lay['bd'], stats=st)
            gn = blk.dilated_conv[1]
            L.call('se_group_finalize', L.ptr(st), _i(B), _i(2 * Cc), _i(0), _i(2 * Cc), _i(16), _d(float(Lp)), L.ptr(gn.weight),
                   L.ptr(gn.bias), L.ptr(ss), _f(gn.eps), L.stream())
            st2 = st_all[i, 1]
            if f16 and GATE_FUSED:
                # the gate as the PROLOGUE of the projection (SE_PRO_GATE, round 5): y2 = sigmoid * tanh of GroupNorm(R) + conditioner is
                # built while the rows are staged and never goes to memory (17 -> 15 plane passes per layer)
                d2 = GM.make_desc(B, 1, Lp, 1, Lp, [(0, 0)], Cc, 2 * Cc, 2 * Cc, 2 * Cc, ldw=Cc, ldx=2 * Cc, prologue=L.PRO_GATE,
                                  epilogue=L.EPI_BIAS | L.EPI_STATS, precision=3, a_sexp=13)
                GM.gemm_tap(d2, R, lay['w216'], R2, bias=lay['b2'], AUX=cond[i], ps=ss, stats=st2)
            elif f16:
                L.call('se_diff_gate', L.ptr(R), L.ptr(ss), L.ptr(cond[i]), L.ptr(y2), _i(B), _l(Lp), _i(Cc), L.stream())
                d2 = GM.make_desc(B, 1, Lp, 1, Lp, [(0, 0)], Cc, Cc, 2 * Cc, 2 * Cc, epilogue=L.EPI_BIAS | L.EPI_STATS,
                                  precision=3, a_sexp=13)
                GM.gemm_tap(d2, y2, lay['w216'], R2, bias=lay['b2'], stats=st2)
            else:
                L.call('se_diff_gate', L.ptr(R), L.ptr(ss), L.ptr(cond[i]), L.ptr(y2), _i(B), _l(Lp), _i(Cc), L.stream())
                d2 = GM.make_desc(B, 1, Lp, 1, Lp, [(0, 0)], Cc, Cc, 2 * Cc, 2 * Cc, epilogue=L.EPI_BIAS | L.EPI_STATS,
                                  precision=min(LY.CONV_PRECISION, 2))
                GM.gemm_tap(d2, y2, lay['w2'], R2, bias=lay['b2'], stats=st2)
            gn2 = blk.output_projection[1]
            L.call('se_group_finalize', L.ptr(st2), _i(B), _i(2 * Cc), _i(Cc), _i(Cc), _i(16), _d(float(Lp)), L.ptr(gn2.weight),
                   L.ptr(gn2.bias), L.ptr(ss2), _f(gn2.eps), L.stream())
            nxt = dproj[i + 1] if i + 1 < nl else None
            if ONE_STREAM:
                L.call('se_diff_mix_y', L.ptr(y), L.ptr(R2), L.ptr(ss2), L.ptr(dproj[i]), L.ptr(nxt), _i(dB), L.ptr(skip), _i(int(i == 0)),
                       _i(B), _l(Lp), _i(Cc), L.ptr(yam[i + 1:i + 2] if (f16 and i + 1 < nl) else None), L.stream())
            else:
                L.call('se_diff_mix_amax', L.ptr(x), L.ptr(R2), L.ptr(ss2), L.ptr(nxt), _i(dB), L.ptr(y), L.ptr(skip), _i(int(i == 0)),
                       _i(B), _l(Lp), _i(Cc), L.ptr(yam[i + 1:i + 2] if (f16 and i + 1 < nl) else None), L.stream())
        GM.gemm_tap(GM.make_desc(1, 1, B * Lp, 1, B * Lp, [(0, 0)], Cc, Cc, Cc, Cc, epilogue=L.EPI_BIAS), skip, pk['ws'], y2,
                    bias=self.skip_projection.bias)
        out = torch.empty(B, Lp, device=dev)
        L.call('se_diff_out', L.ptr(y2), L.ptr(self.output_projection.weight.detach().reshape(-1).contiguous()),
               L.ptr(self.output_projection.bias), L.ptr(out), _l(B * Lp), _i(Cc), L.stream())
        return out

    def forward(self, audio, spectrogram, diffusion_step):
        """models/DiffuSE.py:147-162 (inference): audio [B, 100 T], spectrogram [B, n_specs, T] -> [B, 1, 100 T]"""
        with torch.no_grad():          # inference only (BASELINE config 5): no backward is built for this model family
            return self.denoise(audio, self.conditioner(spectrogram), diffusion_step).unsqueeze(1)


def inference_schedule(config, fast_sampling=False):
    """inference_diffuse.py:117-191 (host arithmetic, float64 like the reference's numpy)"""
    train = np.array(config.NOISE_SCHEDULE)
    beta = np.array(config.INFERENCE_NOISE_SCHEDULE) if fast_sampling else train
    talpha_cum = np.cumprod(1 - train)
    alpha = 1 - beta
    alpha_cum = np.cumprod(alpha)
    n_ = len(alpha)
    sigmas = [0 for _ in alpha]
    for n in range(n_ - 1, -1, -1):
        sigmas[n] = (1.0 - alpha_cum[n - 1]) / (1.0 - alpha_cum[n]) * beta[n]
    T = []
    for s in range(n_):
        for t in range(len(train) - 1):
            if talpha_cum[t + 1] <= alpha_cum[s] <= talpha_cum[t]:
                T.append(t + (talpha_cum[t] ** 0.5 - alpha_cum[s] ** 0.5) / (talpha_cum[t] ** 0.5 - talpha_cum[t + 1] ** 0.5))
                break
    T = np.array(T, dtype=np.float32)
    m = [min((1 - alpha_cum[n]) / (alpha_cum[n] ** 0.5), 1) ** 0.5 for n in range(n_)]
    m[-1] = 1
    delta = [max(1 - (1 + m[n] ** 2) * alpha_cum[n], 0) for n in range(n_)]
    delta_cond, delta_bar = [0] * n_, [0] * n_
    c1, c2, c3 = [0] * n_, [0] * n_, [0] * n_
    for n in range(1, n_):
        r = (1 - m[n]) / (1 - m[n - 1])
        delta_cond[n] = delta[n] - r ** 2 * alpha[n] * delta[n - 1]
        delta_bar[n] = delta_cond[n] * delta[n - 1] / delta[n]
        c1[n] = r * (delta[n - 1] / delta[n]) * alpha[n] ** 0.5 + (1 - m[n - 1]) * (delta_cond[n] / delta[n]) / alpha[n] ** 0.5
        c2[n] = (m[n - 1] * delta[n] - (m[n] * (1 - m[n])) / (1 - m[n - 1]) * alpha[n] * delta[n - 1]) * \
            (alpha_cum[n - 1] ** 0.5 / delta[n])
        c3[n] = (1 - m[n - 1]) * (delta_cond[n] / delta[n]) * (1 - alpha_cum[n]) ** 0.5 / alpha[n] ** 0.5
    c1[0] = 1 / alpha[0] ** 0.5
    c3[0] = c1[0] * beta[0] / (1 - alpha_cum[0]) ** 0.5
    return alpha, beta, alpha_cum, sigmas, T, c1, c2, c3, delta, delta_bar


@torch.no_grad()
def predict(model, config, noisy_signal, alpha, beta, alpha_cum, sigmas, T, c1, c2, c3, delta, delta_bar,
            device=torch.device('cuda'), noises=None, streams=None):
    """inference_diffuse.py:194-228: supportive reverse diffusion of one clip [L] or a batch [B, L] of equal-length clips.
    `noises` (optional, [steps - 1, B, 100 T]) replaces torch.randn_like for reproducible parity runs.
    The conditioner projections are computed once and reused by every step.  `streams`: number of independent batch parts run
    concurrently (default $SE_DIFFUSE_STREAMS or 2; parts hold at least 4 clips)."""
    noisy = torch.as_tensor(np.asarray(noisy_signal), dtype=torch.float32, device=device)
    single = noisy.dim() == 1
    if single:
        noisy = noisy.unsqueeze(0)
    planes, _ = FE.stft_planes(noisy, config.N_FFT, config.HOP_SAMPLES, 'none', padded=False)      # |STFT| in channel 0
    spec = planes[..., 0].transpose(1, 2).contiguous()                                              # [B, F, T]
    Lp = config.HOP_SAMPLES * spec.shape[-1]
    noisy_audio = torch.zeros(noisy.shape[0], Lp, device=device)
    noisy_audio[:, :noisy.shape[1]] = noisy
    cond = model.conditioner(spec)
    B = noisy.shape[0]
    # Clips never interact (GroupNorm is per sample), so the batch can be cut into `streams` independent parts, each stepping
    # through the whole reverse process on its own HIP stream.  Measured at batch 32 (tools/bench_diffuse.py) on two boxes:
    # 17.65 / 17.61 / 16.9 and 16.6 / 18.0 / 17.0 utt/s for 1 / 2 / 3 parts -- two parts are never slower: the default.
    nparts = max(1, min(int(streams if streams is not None else 2), B // 4 or 1))
    cuts = [B * i // nparts for i in range(nparts + 1)]
    parts = [slice(cuts[i], cuts[i + 1]) for i in range(nparts)]
    main = torch.cuda.current_stream(device)
    lanes = [main] if nparts == 1 else [torch.cuda.Stream(device) for _ in parts]
    Tdev = torch.as_tensor(np.asarray(T, dtype=np.float32), device=device)
    if noises is not None:
        noises = torch.as_tensor(np.asarray(noises), dtype=torch.float32, device=device).reshape(-1, B, Lp)
    state = []
    for h, s in zip(parts, lanes):
        s.wait_stream(main)
        state.append({'audio': noisy_audio[h], 'noisy': noisy_audio[h], 'cond': [c[h] for c in cond]})
    gamma, k = [0.2], 0
    for n in range(len(alpha) - 1, -1, -1):
        for h, s, stt in zip(parts, lanes, state):
            with torch.cuda.stream(s):
                audio = stt['audio']
                eps = model.denoise(audio, stt['cond'], Tdev[n:n + 1])
                if n > 0:
                    audio = c1[n] * audio + c2[n] * stt['noisy'] - c3[n] * eps
                    noise = torch.randn_like(audio) if noises is None else noises[k, h]
                    audio = audio + delta_bar[n] ** 0.5 * noise
                else:
                    audio = c1[n] * audio - c3[n] * eps
                    audio = (1 - gamma[n]) * audio + gamma[n] * stt['noisy']
                    audio = torch.clamp(audio, -1.0, 1.0)
                stt['audio'] = audio
        k += 1
    for s, stt in zip(lanes, state):
        if s is not main:
            main.wait_stream(s)
            stt['audio'].record_stream(main)
    audio = state[0]['audio'] if nparts == 1 else torch.cat([stt['audio'] for stt in state], 0)
    return (torch.flatten(audio) if single else audio).cpu().numpy()
