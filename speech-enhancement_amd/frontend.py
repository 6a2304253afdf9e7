"""STFT / iSTFT + power-compression front-end (core/function.py:625-703) on the HIP kernels.

The 400-point real DFT of every frame is one fp32-MFMA tap GEMM: the reflect-padded (and normalised) signal is
the A operand with a row stride of one hop (frames overlap in memory, nothing is unfolded), the windowed DFT
matrix is the weight.  iSTFT = un-compress -> GEMM with the windowed inverse-DFT matrix -> overlap-add.
Internal layout: "planes" [B, T, F, 4] = (|z|, Re z, Im z, 0), channels-last like every other feature map.
"""
import ctypes as _C
import math

import torch

from . import _lib as L
from . import gemm as GM
from . import ops as O

_CACHE = {}
_ci, _cf = _C.c_int, _C.c_float


def hamming(n_fft, device):
    k = torch.arange(n_fft, dtype=torch.float64, device=device)
    return 0.54 - 0.46 * torch.cos(2.0 * math.pi * k / n_fft)


def _ceil4(x):
    return (x + 3) // 4 * 4


def dft_matrices(n_fft, hop, device):
    """(Wf [2F_pad][n_fft], Wi [n_fft][K_pad], env [Lmax...]) in fp32, built once per device in fp64."""
    key = (n_fft, hop, str(device))
    if key in _CACHE:
        return _CACHE[key]
    Fq = n_fft // 2 + 1
    w = hamming(n_fft, device)
    k = torch.arange(n_fft, dtype=torch.float64, device=device)
    f = torch.arange(Fq, dtype=torch.float64, device=device)
    ang = 2.0 * math.pi * f[:, None] * k[None, :] / n_fft             # [F, n_fft]
    Wf = torch.cat([torch.cos(ang) * w, -torch.sin(ang) * w], 0)        # rows: re_f | im_f
    # inverse: frame[k] = w[k]/n * sum_f c_f (re_f cos - im_f sin), c_0 = c_{n/2} = 1 else 2
    c = torch.full((Fq,), 2.0, dtype=torch.float64, device=device)
    c[0] = 1.0
    if n_fft % 2 == 0:
        c[-1] = 1.0
    Kp = _ceil4(2 * Fq)
    Wi = torch.zeros(n_fft, Kp, dtype=torch.float64, device=device)
    Wi[:, :Fq] = (torch.cos(ang) * c[:, None]).T * w[:, None] / n_fft
    Wi[:, Fq:2 * Fq] = (-torch.sin(ang) * c[:, None]).T * w[:, None] / n_fft
    out = (Wf.float().contiguous(), Wi.float().contiguous(), w)
    _CACHE[key] = out
    return out


def envelope(n_fft, hop, T, device):
    key = ('env', n_fft, hop, T, str(device))
    if key not in _CACHE:
        w2 = (hamming(n_fft, device) ** 2).cpu()                 # summed on the host: T tiny device launches otherwise (1601 at 10 s)
        Lp = n_fft + hop * (T - 1)
        env = torch.zeros(Lp, dtype=torch.float64)
        for t in range(T):
            env[t * hop:t * hop + n_fft] += w2
        _CACHE[key] = env.float().contiguous().to(device)
    return _CACHE[key]


def fused_matrices(device):
    """interleaved DFT matrices of the fused kernels (csrc/se_front.hip): Wf [400][416] (column 2f = Re, 2f + 1 = Im of bin f),
    Wi [404][416] (row 2f / 2f + 1 = Re / Im coefficient of bin f)"""
    key = ('fused', str(device))
    if key not in _CACHE:
        Wf, Wi, _ = dft_matrices(400, 100, device)           # Wf [402][400] rows re|im ; Wi [400][404] cols re|im
        Fq = 201
        Wfi = torch.zeros(400, 416, device=device, dtype=torch.float32)
        Wfi[:, 0:2 * Fq:2] = Wf[:Fq].t()
        Wfi[:, 1:2 * Fq:2] = Wf[Fq:2 * Fq].t()
        Wii = torch.zeros(404, 416, device=device, dtype=torch.float32)
        Wii[0:2 * Fq:2, :400] = Wi[:, :Fq].t()
        Wii[1:2 * Fq:2, :400] = Wi[:, Fq:2 * Fq].t()
        _CACHE[key] = (Wfi.contiguous(), Wii.contiguous())
    return _CACHE[key]


def stft_planes(x, n_fft=400, hop=100, comp='pow', scale=None, padded=True):
    """x [B, L] (L a multiple of hop) -> planes [B, T, F, 4].  scale: optional per-clip factor c[b] applied
    while padding (fuses normalize_batch, core/function.py:647-659).  Returns (planes, xp): xp = the scaled reflect-padded
    signal [B, L + n_fft] (its middle is the normalised clip the losses use) or None with padded=False.
    n_fft = 400 / hop = 100 (the reference's only analysis) runs the ONE-launch fused kernel (normalise + reflect-pad + frame +
    windowed DFT + compression); other sizes the pad -> tap-GEMM -> compress sequence."""
    L.check_cuda(x)
    B, Ls = x.shape
    fused = n_fft == 400 and hop == 100 and Ls > 200
    if (Ls % hop != 0 and not (fused and not padded)) or n_fft % hop != 0 or hop % 4 != 0:
        raise L.SeHipError(f'stft: need L % hop == 0, n_fft % hop == 0, hop % 4 == 0 (L={Ls}, hop={hop})')
    T = Ls // hop + 1
    Fq = n_fft // 2 + 1
    x = x.contiguous()
    pre = n_fft ** -0.5 if comp == 'norm' else 1.0
    if fused:
        Wfi, _ = fused_matrices(x.device)
        P = torch.empty(B, T, Fq, 4, device=x.device, dtype=torch.float32)
        L.call('se_stft_fused', L.ptr(x), L.ptr(scale), L.ptr(Wfi), L.ptr(P), _ci(B), _ci(Ls), _ci(n_fft), _ci(hop),
               _ci(O.COMP[comp]), _cf(pre), L.stream(), _key='stft_fused', _flops=2.0 * B * T * 400 * 402,
               _bytes=4.0 * B * (Ls + T * Fq * 4))
        return P, (O.reflect_pad_scale(x, scale, n_fft // 2) if padded else None)
    xp = O.reflect_pad_scale(x, scale, n_fft // 2)          # [B, L + n_fft]
    Wf, _, _ = dft_matrices(n_fft, hop, x.device)
    ldr = _ceil4(2 * Fq)
    R = torch.empty(B * T, ldr, device=x.device, dtype=torch.float32)
    rows_per_b = (Ls + n_fft) // hop                                      # "pixels" of stride hop per clip
    d = GM.make_desc(B, 1, T, 1, rows_per_b, [(0, 0)], n_fft, hop, 2 * Fq, ldr, ldw=n_fft)
    GM.gemm_tap(d, xp, Wf, R)
    return O.compress_planes(R, ldr, B * T, Fq, comp, pre).view(B, T, Fq, 4), xp


class _STFTFn(torch.autograd.Function):
    """differentiable STFT (the consistency-preserving re-STFT of est_audio, core/function.py:234-236)."""

    @staticmethod
    def forward(ctx, x, n_fft, hop, comp):
        B, Ls = x.shape
        T = Ls // hop + 1
        Fq = n_fft // 2 + 1
        xp = O.reflect_pad_scale(x.contiguous(), None, n_fft // 2)
        Wf, _, _ = dft_matrices(n_fft, hop, x.device)
        ldr = _ceil4(2 * Fq)
        R = torch.zeros(B * T, ldr, device=x.device, dtype=torch.float32)
        d = GM.make_desc(B, 1, T, 1, (Ls + n_fft) // hop, [(0, 0)], n_fft, hop, 2 * Fq, ldr, ldw=n_fft)
        GM.gemm_tap(d, xp, Wf, R)
        pre = n_fft ** -0.5 if comp == 'norm' else 1.0
        ctx.save_for_backward(R)
        ctx.cfg = (B, Ls, T, Fq, n_fft, hop, comp, ldr, pre)
        return O.compress_planes(R, ldr, B * T, Fq, comp, pre).view(B, T, Fq, 4)

    @staticmethod
    def backward(ctx, dP):
        (R,) = ctx.saved_tensors
        B, Ls, T, Fq, n_fft, hop, comp, ldr, pre = ctx.cfg
        Wf, _, _ = dft_matrices(n_fft, hop, R.device)
        dR = O.compress_planes_bwd(R, ldr, dP.contiguous(), B * T, Fq, comp, pre)
        key = ('WfT', n_fft, hop, str(R.device))
        if key not in _CACHE:          # [n_fft][ldr]: transposed DFT matrix, zero-padded columns
            WfT = torch.zeros(n_fft, ldr, device=R.device, dtype=torch.float32)
            WfT[:, :2 * Fq] = Wf.t()
            _CACHE[key] = WfT.contiguous()
        dfr = torch.empty(B * T, n_fft, device=R.device, dtype=torch.float32)
        GM.gemm_tap(GM.linear_desc(B * T, ldr, n_fft), dR, _CACHE[key], dfr)
        dxp = O.ola(dfr, None, B, T, n_fft, hop, trim=0, L_out=Ls + n_fft)
        return O.reflect_pad_bwd(dxp, None, B, Ls, n_fft // 2), None, None, None


def stft_planes_grad(x, n_fft=400, hop=100, comp='pow'):
    return _STFTFn.apply(x, n_fft, hop, comp)


class _ISTFTFn(torch.autograd.Function):
    """planes [B,T,F,4] (compressed Re/Im in channels 1,2) -> audio [B, hop*(T-1)]."""

    @staticmethod
    def forward(ctx, planes, n_fft, hop, comp):
        B, T, Fq, _ = planes.shape
        _, Wi, _ = dft_matrices(n_fft, hop, planes.device)
        lda = Wi.shape[1]
        post = n_fft ** 0.5 if comp == 'norm' else 1.0
        env = envelope(n_fft, hop, T, planes.device)
        planes = planes.contiguous()
        if n_fft == 400 and hop == 100 and T > 1:       # one launch: un-compress + inverse DFT + overlap-add + envelope + trim
            _, Wii = fused_matrices(planes.device)
            y = torch.empty(B, hop * (T - 1), device=planes.device, dtype=torch.float32)
            L.call('se_istft_fused', L.ptr(planes), L.ptr(Wii), L.ptr(env), L.ptr(y), _ci(B), _ci(T), _ci(n_fft), _ci(hop),
                   _ci(O.COMP[comp]), _cf(post), L.stream(), _key='istft_fused', _flops=2.0 * B * T * 402 * 400,
                   _bytes=4.0 * B * (T * Fq * 4 + hop * (T - 1)))
        else:
            Au = O.uncompress_rows(planes, B * T, Fq, lda, comp, post)
            frames = torch.empty(B * T, n_fft, device=planes.device, dtype=torch.float32)
            GM.gemm_tap(GM.linear_desc(B * T, lda, n_fft), Au, Wi, frames)
            y = O.ola(frames, env, B, T, n_fft, hop)
        ctx.save_for_backward(planes)
        ctx.cfg = (n_fft, hop, comp, lda, post)
        return y

    @staticmethod
    def backward(ctx, dy):
        (planes,) = ctx.saved_tensors
        n_fft, hop, comp, lda, post = ctx.cfg
        B, T, Fq, _ = planes.shape
        _, Wi, _ = dft_matrices(n_fft, hop, planes.device)
        env = envelope(n_fft, hop, T, planes.device)
        dfr = O.ola_bwd(dy.contiguous(), env, B, T, n_fft, hop)
        dA = torch.empty(B * T, lda, device=planes.device, dtype=torch.float32)
        GM.gemm_tap(GM.linear_desc(B * T, n_fft, lda), dfr, Wi.t().contiguous(), dA)
        dP = torch.zeros_like(planes)
        O.uncompress_rows_bwd(planes, dA, lda, dP, B * T, Fq, comp, post)
        return dP, None, None, None


def istft_planes(planes, n_fft=400, hop=100, comp='pow'):
    return _ISTFTFn.apply(planes, n_fft, hop, comp)


# ---- reference-shaped public API (complex [B, F, T] tensors) ---------------------------------------
def spec_to_planes(spec):
    """complex [B, F, T] -> planes [B, T, F, 4] (layout plumbing for external callers)."""
    re = spec.real.transpose(1, 2)
    im = spec.imag.transpose(1, 2)
    return torch.stack([torch.sqrt(re * re + im * im), re, im, torch.zeros_like(re)], -1).contiguous()


def planes_to_spec(planes):
    return torch.complex(planes[..., 1], planes[..., 2]).transpose(1, 2)


def _check_window(window, n_fft):
    """The kernels fold the periodic Hamming window (torch.hamming_window(n_fft), the only window the reference ever
    passes: core/function.py:668) into the DFT matrices.  Any other window would be silently wrong -> refuse it."""
    if window is None:
        return
    key = ('wchk', n_fft, window.data_ptr(), window._version, str(window.device))
    if _CACHE.get('wchk_last') == key:
        return
    if window.numel() != n_fft or \
            not torch.allclose(window.detach().double().cpu(), hamming(n_fft, 'cpu'), rtol=0.0, atol=1e-6):
        raise L.SeHipError(f'the HIP STFT/iSTFT path is built for the periodic Hamming window hamming_window({n_fft}) '
                           f'(core/function.py:668); got a different window')
    _CACHE['wchk_last'] = key


def compressed_stft(signal, n_fft, hop_length, window=None, comp_type='pow'):
    """core/function.py:685-693.  `window` must be None or the periodic Hamming window (checked)."""
    _check_window(window, n_fft)
    planes, _ = stft_planes(signal, n_fft, hop_length, comp_type)
    return planes_to_spec(planes)


def uncompressed_istft(spec, n_fft, hop_length, window=None, comp_type='pow'):
    """core/function.py:695-703.  `window` must be None or the periodic Hamming window (checked)."""
    _check_window(window, n_fft)
    return istft_planes(spec_to_planes(spec), n_fft, hop_length, comp_type)


def normalize_batch(batch, args=None):
    """core/function.py:647-659: moves the pair to args.gpu when given, scales both signals by c = sqrt(L / sum noisy^2)."""
    clean, noisy = batch['audio'], batch['noisy']
    if getattr(args, 'gpu', None) is not None:
        clean, noisy = clean.cuda(args.gpu, non_blocking=True), noisy.cuda(args.gpu, non_blocking=True)
    c = O.clip_scale(noisy.contiguous())
    return clean * c[:, None], noisy * c[:, None]


def disassemble_spectrogram(spec):
    """core/function.py:661-662."""
    return spec.abs(), spec.real, spec.imag


def power_compress(spec, comp_type=None):
    """core/function.py:625-634 on a complex [B,F,T] tensor (reference-shaped API; the train step never leaves the
    planes layout)."""
    B, Fq, T = spec.shape
    R = torch.cat([spec.real.transpose(1, 2), spec.imag.transpose(1, 2)], -1).reshape(B * T, 2 * Fq).contiguous()
    return planes_to_spec(O.compress_planes(R, 2 * Fq, B * T, Fq, comp_type).view(B, T, Fq, 4))


def power_uncompress(spec, comp_type=None):
    """core/function.py:636-645."""
    B, Fq, T = spec.shape
    A = O.uncompress_rows(spec_to_planes(spec), B * T, Fq, 2 * Fq, comp_type).view(B, T, 2 * Fq)
    return torch.complex(A[..., :Fq], A[..., Fq:]).transpose(1, 2)
