"""Functional forward / backward of the TSCNet generator on the HIP kernels.

Everything is channels-last fp32: feature maps [B, T, F, C], Conformer tokens = the same buffer viewed as
[B*T*F, C] (the reference's two permute+contiguous copies per TSCB, models/generator.py:69,71, become token
strides).  Each ``*_fwd`` returns (output, ctx); each ``*_bwd`` consumes the ctx, returns the input gradient and
adds parameter gradients into ``G`` (dict name -> tensor, fp32, zero-initialised by the caller).
Parameter names are the reference's state_dict keys.
"""
import ctypes as C

import torch

from . import _lib as L
from . import attention as A
from . import gemm as GM
from . import ops as O

EPS = 1e-5

# Arithmetic of the MFMA-bound convolution GEMMs (dilated dense stacks, sub-pixel, strided convs): 'f32' = exact
# fp32 MFMA; 'bf16x6' (default) = every fp32 operand split EXACTLY into three bf16 parts, six bf16 MFMAs per product,
# fp32 accumulation: fp32-equivalent results (dropped terms <= 2^-24) at 16/6 of the fp32 MFMA rate; 'bf16x3' = two
# parts / three MFMAs, ~1.5e-5 relative per product (meets the 1e-3 bar with less margin; opt-in).
import os as _os
EPS = 1e-5

# Arithmetic of the MFMA-bound convolution GEMMs (dilated dense stacks, sub-pixel, strided convs): 'f32' = exact
# fp32 MFMA; 'bf16x6' (default) = every fp32 operand split EXACTLY into three bf16 parts, six bf16 MFMAs per product,
# fp32 accumulation: fp32-equivalent results (dropped terms <= 2^-24) at 16/6 of the fp32 MFMA rate; 'bf16x3' = two
# parts / three MFMAs, ~1.5e-5 relative per product (meets the 1e-3 bar with less margin; opt-in); 'f16x3' = SCALED split-fp16
# (se_gemm_desc precision 3): every operand times a power of two that brings its maximum to [2^13, 2^14), then hi + lo in fp16
# (2 x 11 significand bits, 2^-24 relative), three fp16 MFMAs per product -- measured MORE accurate than both fp32-MFMA and
# bf16x6 against fp64 (rms 4.4e-7 vs 6.9e-7 / 6.0e-7 at Cin = 256) at 1.44x - 1.74x the bf16x6 speed.  The triple-tap unit-stride
# convolutions (90 % of the conv FLOPs) run it; the other conv shapes keep the six-product kernels under this setting.
import os as _os
_PREC = {'f32': 0, 'bf16x3': 1, 'bf16x6': 2, 'f16x3': 3}
CONV_PRECISION = _PREC[_os.environ.get('SE_CONV_PRECISION', 'f16x3')]
# static operand exponent of NORMALISED activations under precision 3 (InstanceNorm / LayerNorm outputs after PReLU, the TSCB
# token stream): |x| * 2^4 must stay below 65504, i.e. |x| < 4094 -- an InstanceNorm output is bounded by sqrt(T F - 1) |gamma| +
# |beta| = 254 |gamma| + |beta|; out-of-range values would be clamped (FP16_OVFL), not turned into inf.  Gradients and weights
# are scaled by their MEASURED maxima instead (amax scalars).
ACT_SEXP = 4
PW2_F16 = True     # pointwise conv 128 -> 64 behind BatchNorm + Swish on the scaled-fp16 kernel (False: fp32 MFMA; module switch for A/B runs)
ATTN_O_SEXP = 4    # attention outputs (convex combinations of the value rows): |o| < 4094
# attention products: 'f16x3' = the scaled split-fp16 kernels (se_attn_fwd_f16 / se_attn_bwd_f16_phase; operand scales from the
# maxima the qkv / to_out input-gradient GEMMs raise: se_gemm_desc.y_amax) wherever the sequence fits them, 'bf16x6' = the exact
# three-way bf16 split kernels everywhere (SE_ATTN_PRECISION)
ATTN_PRECISION = [_os.environ.get('SE_ATTN_PRECISION', 'f16x3')]


# weight-gradient GEMMs of the same convolutions (contraction over pixels; same operand splits, transposed staging).
# 'bf16x6' routes the triple-tap layers with Fo > 66 to wgrad3_bf16_kernel (108 vs 94 TFLOP/s for the fp32 triple kernel);
# every other shape runs the fp32-MFMA kernels under it (the generic six-product kernel is slower than fp32 MFMA).
WGRAD_PRECISION = [_PREC[_os.environ.get('SE_WGRAD_PRECISION', 'f16x3')]]


# input-gradient GEMM of the qkv / pointwise-GLU projections fused with the backward of the LayerNorm in front of them
# (se_gemm_ln_bwd); module switches (no environment variables since round 6): tests flip them to cross-check the unfused forms
FUSE_LN_BWD = True
# LayerNorm row statistics emitted by the producer of the rows instead of a separate se_row_stats pass
FUSE_ROWSTATS = True
# GLU backward in the epilogue of the depthwise input-gradient kernel (se_dwconv31_glu_bwd)
FUSE_GLU_BWD = True


def set_conv_precision(name, wgrad=None):
    global CONV_PRECISION
    CONV_PRECISION = _PREC[name]
    WGRAD_PRECISION[0] = _PREC[wgrad or name]


def _conv3_shape(taps, sf, C_in, Fo=None):
    """the triple-tap unit-stride shapes conv3_bf16_kernel / wgrad3_bf16_kernel cover (the only ones with a precision-3 kernel)"""
    if sf != 1 or C_in < 32 or len(taps) % 3 or (Fo is not None and Fo < 2):
        return False
    for i in range(0, len(taps), 3):
        if {t[0] for t in taps[i:i + 3]} != {taps[i][0]} or sorted(abs(t[1]) for t in taps[i:i + 3]) != [0, 1, 1] or \
                {t[1] for t in taps[i:i + 3]} != {-1, 0, 1}:
            return False
    return True


def _prec(p, taps, sf, C_in, Fo=None, have_scale=True):
    """precision 3 only where a scaled split-fp16 kernel exists and the operand scales are known; six-product otherwise"""
    if p != 3:
        return p
    return 3 if (have_scale and _conv3_shape(taps, sf, C_in, Fo)) else 2


def _prec_w(p, W, taps, sf, C_in, Fo=None, have_scale=True):
    """precision of a forward / input-gradient conv GEMM from the form its weight arrives in: scaled fp16 planes (the step's
    WeightPlan chose precision 3 for this layer) -> 3, bf16 planes -> 2, plain fp32 -> `_prec`"""
    if W.dtype == torch.float16:
        return 3
    if W.dtype == torch.bfloat16:
        return 2
    return _prec(p, taps, sf, C_in, Fo, have_scale)


class DPHooks:
    """Data-parallel hooks: SyncBatchNorm statistic exchange (main_gan.py:154-155).  Single-GPU default: none."""
    world = 1

    def allreduce(self, t):
        return t


NO_DP = DPHooks()


def _T(w):
    """[N, K] -> contiguous [K, N] (weight-sized plumbing)."""
    return w.t().contiguous()


# ------------------------------------------------------------------------------------------------
# conv (+ InstanceNorm + PReLU) building block
# ------------------------------------------------------------------------------------------------
def conv_fwd(x, B, Ti, Fi, lda, a_off, C_in, wp, bias, taps, N, To=None, Fo=None, sf=1, shuffle2=False,
             want_stats=True, a_amax=None):
    """raw conv output R [B, To, Fo(*2 if shuffle2), N(/2)] (+ fp64 (sum, sumsq) statistics).
    a_amax: device scalar >= max |x| over the channels read (raised by the producers of x: inorm_prelu_fwd / copy_cols_amax) -- the
    operand scale of the scaled split-fp16 kernels; None: the static exponent ACT_SEXP (callers without a measured maximum)"""
    To = To or Ti
    Fo = Fo or Fi
    ep = L.EPI_BIAS | (L.EPI_STATS if want_stats else 0) | (L.EPI_SHUFFLE2 if shuffle2 else 0)
    No = N // 2 if shuffle2 else N
    d = GM.make_desc(B, To, Fo, Ti, Fi, taps, C_in, lda, N, No, a_off=a_off, sf=sf, epilogue=ep,
                     precision=_prec_w(CONV_PRECISION, wp, taps, sf, C_in, Fo), a_sexp=ACT_SEXP, a_amax=a_amax, w_sexp=8)
    R = torch.empty(B, To, Fo * (2 if shuffle2 else 1), No, device=x.device, dtype=torch.float32)
    stats = O.zeros(B, No, 2, device=x.device, dtype=torch.float64) if want_stats else None
    GM.gemm_tap(d, x, wp, R, bias=bias, stats=stats)
    return R, stats


# module switch (A/B runs): 0 sends the decoders' last convolutions and the encoder's first one through the tap GEMM again
THIN_CONV = 15     # bits: 1 encoder fwd, 2 encoder wgrad, 4 decoders fwd, 8 decoders bwd


def conv1x2_fwd(x, w, bias, B, T, F2, want_stats):
    """Conv2d(64, n, (1, 2)), n = 1 / 2, as a direct kernel (csrc/se_thin.hip): x [B,T,F2,64] -> [B,T,F2-1,4] (+ fp64 sums [B,4,2])"""
    n = w.shape[0]
    y = torch.empty(B, T, F2 - 1, 4, device=x.device, dtype=torch.float32)
    st = O.zeros(B, 4, 2, device=x.device, dtype=torch.float64) if want_stats else None
    L.call('se_conv1x2_fwd', L.ptr(x), L.ptr(w), L.ptr(bias), L.ptr(y), L.ptr(st), C.c_int(B), C.c_int(T), C.c_int(F2), C.c_int(n),
           L.stream(), _key='thin conv (generator)', _bytes=4.0 * B * T * (F2 * 64 + (F2 - 1) * 4))
    return y, st


def conv1x2_bwd(x, w, dy, dw, dbias, B, T, F2):
    """gradients of conv1x2_fwd: dw / dbias accumulated in place (PyTorch layout) on the weight-gradient stream, returns dx"""
    n = w.shape[0]
    with GM.leaf_stream(x, dy):
        L.call('se_conv1x2_wgrad', L.ptr(x), L.ptr(dy), L.ptr(dw), L.ptr(dbias), C.c_long(B * T), C.c_int(F2), C.c_int(n), L.stream(),
               _key='thin conv (generator)', _bytes=4.0 * B * T * (F2 * 64 + (F2 - 1) * 4))
    dx = torch.empty(B, T, F2, 64, device=x.device, dtype=torch.float32)
    L.call('se_conv1x2_dgrad', L.ptr(dy), L.ptr(w), L.ptr(dx), C.c_long(B * T), C.c_int(F2), C.c_int(n), L.stream(),
           _key='thin conv (generator)', _bytes=4.0 * B * T * (F2 * 64 + (F2 - 1) * 4))
    return dx


def inorm_prelu_fwd(R, stats, g, b, slope, out, ldy, y_off, amax=None):
    B = R.shape[0]
    C_ = R.shape[-1]
    P = R.numel() // (B * C_)
    return O.inorm_prelu_fwd(R, C_, 0, stats, g, b, slope, out, ldy, y_off, B, P, C_, amax=amax)


def inorm_prelu_bwd(R, mr, g, b, slope, dy, ldy, y_off, dg, db, dslope):
    B = R.shape[0]
    C_ = R.shape[-1]
    P = R.numel() // (B * C_)
    dR = torch.empty_like(R)
    # max |dR| for the scaled split-fp16 kernels that read dR next (conv input / weight gradient): one atomic per wave
    amax = O.zeros(1, device=R.device) if 3 in (CONV_PRECISION, WGRAD_PRECISION[0]) else None
    O.norm_prelu_bwd(R, C_, 0, mr, g, b, slope, dy, ldy, y_off, dR, C_, 0, dg, db, dslope, B, P, C_, per_batch=True, amax=amax)
    dR._se_amax = amax
    return dR


def conv_bwd(x, B, Ti, Fi, lda, a_off, C_in, w, taps, dR, To, Fo, dw, dbias, sf=1, rev=False, dx=None, lddx=None,
             dx_off=0, accumulate=False, need_dx=True, w_for_dgrad=None, wd=None, a_amax=None):
    """gradients of a (non-shuffled) conv: weight/bias into dw/dbias (PyTorch layout, accumulated), input
    gradient into dx[..., dx_off:dx_off+C_in] (pixel stride lddx)."""
    N = dR.shape[-1]
    ntap = len(taps)
    amax = getattr(dR, '_se_amax', None)            # producer-measured max |dR| (inorm_prelu_bwd); None: no scaled-fp16 kernels here
    fd = GM.make_desc(B, To, Fo, Ti, Fi, taps, C_in, lda, N, N, a_off=a_off, sf=sf,
                      precision=_prec(WGRAD_PRECISION[0], taps, sf, C_in, Fo, have_scale=amax is not None and Fo > 66),
                      a_sexp=ACT_SEXP, a_amax=a_amax, w_amax=amax)
    with GM.leaf_stream(x, dR, amax, a_amax):
        dwp = O.zeros(N, ntap * C_in, device=dR.device)
        GM.gemm_tap_wgrad(fd, x, dR, dwp, dbias)
        _unpack_w(dwp, dw, C_in, rev)
    if not need_dx:
        return None
    if wd is None:               # no prepared [Cin][tap][N] matrix (weights.WeightPlan): build it now
        wd = GM.pack_conv_dgrad(w_for_dgrad if w_for_dgrad is not None else w, rev_slabs=rev)
        if wd.shape[0] != C_in:      # channel-padded input (Cin 3 -> 4 etc.)
            wd = torch.cat([wd, wd.new_zeros(C_in - wd.shape[0], wd.shape[1])], 0)
    if dx is None:
        dx = torch.empty(B, Ti, Fi, C_in, device=dR.device, dtype=torch.float32)
        lddx = C_in
    dtaps = [(-a, -c) for a, c in taps]
    dd = GM.make_desc(B, Ti, Fi, To, Fo, dtaps, N, N, C_in, lddx, c_off=dx_off, sf=sf,
                      up=1 if sf != 1 else 0, epilogue=L.EPI_ACCUM if accumulate else 0,
                      precision=_prec_w(CONV_PRECISION, wd, dtaps, sf, N, Fi, have_scale=amax is not None), a_amax=amax, w_sexp=8)
    GM.gemm_tap(dd, dR, wd, dx)
    return dx


def _unpack_w(dwp, dw, C_in, rev):
    """packed [N][tap][C_in] -> accumulate into PyTorch-layout dw [N, Cin_true, kh, kw] (Cin_true <= C_in)."""
    N, Ct, kh, kw = dw.shape
    if Ct == C_in:
        GM.unpack_conv_wgrad(dwp, dw, rev_slabs=rev, accumulate=True)
    else:   # padded input channels: drop the pad
        tmp = torch.zeros(N, C_in, kh, kw, device=dw.device, dtype=torch.float32)
        GM.unpack_conv_wgrad(dwp, tmp, rev_slabs=False, accumulate=False)
        dw += tmp[:, :Ct]


def pack_w(w, C_pad=None, rev=False):
    """PyTorch conv weight -> [N][tap][C] (optionally zero-padding the input channels to C_pad)."""
    if C_pad is not None and w.shape[1] != C_pad:
        w = torch.cat([w, w.new_zeros(w.shape[0], C_pad - w.shape[1], w.shape[2], w.shape[3])], 1).contiguous()
    return GM.pack_conv_fwd(w, rev_slabs=rev)


def pad_rows(w, n):
    """zero-pad the leading (output-channel) dim to n."""
    if w.shape[0] == n:
        return w.contiguous()
    return torch.cat([w, w.new_zeros(n - w.shape[0], *w.shape[1:])], 0).contiguous()


# ------------------------------------------------------------------------------------------------
# prepared weights: P['__prep__'] (weights.WeightPlan, refreshed by ONE launch per forward) holds every re-packed / transposed /
# pre-split weight of the generator; without a plan (direct calls of these functions with a plain dict) the same matrices are
# built on the fly, in fp32, by the round-1 helpers
# ------------------------------------------------------------------------------------------------
def _w(P, key, fallback):
    plan = P.get('__prep__')
    t = plan.out.get(key) if plan is not None else None
    return t if t is not None else fallback()


def build_generator_plan(P, device):
    from .weights import WeightPlan
    plan = WeightPlan(device)
    cpl = CONV_PRECISION in (2, 3)       # six-product kernels read the exact hi/mid/lo planes; everything else fp32
    c3 = 'f16' if CONV_PRECISION == 3 else cpl      # triple-tap unit-stride convolutions under precision 3: scaled fp16 planes
    lpl = GM.LINEAR_PRECISION in (2, 3)
    l3 = 'f16' if GM.LINEAR_PRECISION == 3 else lpl      # token-wise GEMMs under precision 3: scaled fp16 planes

    def dense(p):
        for i in range(4):
            n = f'{p}.conv{i+1}.weight'
            plan.conv_fwd((n, 'fwd'), P[n], rev=True, planes=c3)
            plan.conv_dgrad((n, 'dgrad'), P[n], rev=True, planes=c3)

    for e in ('dense_encoder', 'dense_encoder_noisy'):       # (the second one: TSC-diffusion hybrid, models/tsc_diffusion.py:47)
        if f'{e}.conv_1.0.weight' not in P:
            continue
        if not ((THIN_CONV & 3) == 3 and P[f'{e}.conv_1.0.weight'].shape[:2] == (64, 3)):      # (direct kernel: reads the PyTorch layout)
            plan.conv_fwd((f'{e}.conv_1.0.weight', 'fwd'), P[f'{e}.conv_1.0.weight'], C_pad=4)
        dense(f'{e}.dilated_dense')
        plan.conv_fwd((f'{e}.conv_2.0.weight', 'fwd'), P[f'{e}.conv_2.0.weight'], planes=c3)          # strided: generic split kernel
        plan.conv_dgrad((f'{e}.conv_2.0.weight', 'dgrad'), P[f'{e}.conv_2.0.weight'], planes=c3)
    for dec, last in (('mask_decoder', 'conv_1'), ('complex_decoder', 'conv')):
        dense(f'{dec}.dense_block')
        n = f'{dec}.sub_pixel.conv.weight'
        plan.conv_fwd((n, 'fwd'), P[n], planes=c3)
        plan.conv_dgrad((n, 'dgrad'), P[n], planes=c3 if dec == 'complex_decoder' else cpl)      # mask decoder: dS has no measured amax
        n = f'{dec}.{last}.weight'                                   # 64 -> 1 / 2 channels, rows padded to 4
        if (THIN_CONV & 12) != 12:                                             # (direct kernels: read the PyTorch layout)
            plan.conv_fwd((n, 'fwd'), P[n], N_pad=4, planes=cpl)
            plan.conv_dgrad((n, 'dgrad'), P[n], N_pad=4)              # C = 4 < 32: fp32 kernel
    for i in range(1, 5):
        for ax in ('time', 'freq'):
            p = f'TSCB_{i}.{ax}_conformer'
            for ff in ('ff1', 'ff2'):
                n1, n2 = f'{p}.{ff}.fn.fn.net.0.weight', f'{p}.{ff}.fn.fn.net.3.weight'
                if lpl and P[n1].shape[1] == 64 and P[n2].shape[0] == 64 and P[n1].shape[0] % 64 == 0:
                    f3 = 'f16' if (l3 == 'f16' and P[n1].shape[0] == 256) else True       # the fp16 FF kernels are built for hid = 256
                    plan.linear((n1, 'lin'), P[n1], planes=f3)
                    plan.linear((n2, 'lin'), P[n2], planes=f3)
                    plan.linear_T((n2, 'T0.5'), P[n2], planes=f3, scale=0.5)
                    plan.linear_T((n1, 'T'), P[n1], planes=f3)
            # proven bounds of the normalised activations that feed scaled split-fp16 GEMMs (refreshed by run_bounds every forward)
            for ff in ('ff1', 'ff2'):
                g_, b_ = P[f'{p}.{ff}.fn.norm.weight'], P[f'{p}.{ff}.fn.norm.bias']
                plan.bound(('ln', f'{p}.{ff}'), g_, b_)
                # |Swish(H) mask / keep| <= (|LN(x)| max row-l1(W1) + max |b1|) / keep, keep >= 1/2 (checked where it is used)
                plan.bound(('hid', f'{p}.{ff}'), g_, b_, W=P[f'{p}.{ff}.fn.fn.net.0.weight'], wb=P[f'{p}.{ff}.fn.fn.net.0.bias'], post=2.0)
            plan.bound(('ln', f'{p}.attn'), P[f'{p}.attn.norm.weight'], P[f'{p}.attn.norm.bias'])
            plan.bound(('ln', f'{p}.conv'), P[f'{p}.conv.net.0.weight'], P[f'{p}.conv.net.0.bias'])
            # train-mode BatchNorm over M tokens: |x_hat| <= sqrt(M - 1) (k1 of run_bounds); Swish(y) <= |y|
            plan.bound(('bn', f'{p}.conv'), P[f'{p}.conv.net.5.weight'], P[f'{p}.conv.net.5.bias'], ksel=1)
            a = f'{p}.attn.fn'
            plan.linear((a, 'qkv'), P[f'{a}.to_q.weight'], planes=l3, rows=192)
            plan.linear((a, 'qkv'), P[f'{a}.to_kv.weight'], planes=l3, o_off=64)
            # the three-way bf16 attention backward does not measure max |dqkv|: the input-gradient GEMM of the qkv projection keeps
            # the six-product kernel there; the scaled split-fp16 backward does (dqkv._se_amax) -> fp16 planes for that case
            plan.linear_T((a, 'qkvT'), P[f'{a}.to_q.weight'], planes=lpl, ld=192)
            plan.linear_T((a, 'qkvT'), P[f'{a}.to_kv.weight'], planes=lpl, c_off=64)
            if l3 == 'f16' and ATTN_PRECISION[0] == 'f16x3':
                plan.linear_T((a, 'qkvT16'), P[f'{a}.to_q.weight'], planes='f16', ld=192)
                plan.linear_T((a, 'qkvT16'), P[f'{a}.to_kv.weight'], planes='f16', c_off=64)
            plan.linear_T((f'{a}.to_out.weight', 'T'), P[f'{a}.to_out.weight'], planes='f16' if l3 == 'f16' else False)
            if l3 == 'f16':
                plan.linear((f'{a}.to_out.weight', 'lin'), P[f'{a}.to_out.weight'], planes='f16')
            # [3][2 maxpos + 1][16] bf16 or [2][..][16] scaled fp16 (+ max |E|)
            plan.linear((f'{a}.rel_pos_emb.weight', 'es'), P[f'{a}.rel_pos_emb.weight'],
                        planes='f16' if ATTN_PRECISION[0] == 'f16x3' else True)
            n = f'{p}.conv.net.2.weight'
            plan.linear((n, 'lin'), P[n], planes=l3)
            plan.linear_T((n, 'T'), P[n], planes=l3)
            n = f'{p}.conv.net.7.weight'
            plan.linear_T((n, 'T'), P[n], planes=l3)
            if l3 == 'f16' and PW2_F16:
                plan.linear((n, 'lin'), P[n], planes='f16')
    return plan


# ------------------------------------------------------------------------------------------------
# DilatedDenseNet (models/generator.py:6-32)
# ------------------------------------------------------------------------------------------------
def dense_taps(i):
    d = 2 ** i
    return GM.conv_taps(2, 3, (d, 1), (d, 1))      # causal in time: dt in {-d, 0}, df in {-1, 0, 1}


def dense_block_fwd(P, p, skip, B, T, Fq, amax=None):
    """skip: [B,T,Fq,256] whose slab 0 already holds the block input.  Returns (out [B,T,Fq,64], ctx).
    amax: device scalar >= max |block input| (raised by its producer); every InstanceNorm + PReLU of the block raises it further,
    so that layer i reads the maximum over the slabs 0 .. i it convolves (and the consumer of `out` the maximum over everything)"""
    ctx = {'skip': skip, 'R': [], 'mr': [], 'amax': amax}
    out = None
    for i in range(4):
        C_in = 64 * (i + 1)
        wp = _w(P, (f'{p}.conv{i+1}.weight', 'fwd'), lambda: pack_w(P[f'{p}.conv{i+1}.weight'], rev=True))
        R, stats = conv_fwd(skip, B, T, Fq, 256, 0, C_in, wp, P[f'{p}.conv{i+1}.bias'], dense_taps(i), 64, a_amax=amax)
        if i < 3:
            mr = inorm_prelu_fwd(R, stats, P[f'{p}.norm{i+1}.weight'], P[f'{p}.norm{i+1}.bias'],
                                 P[f'{p}.prelu{i+1}.weight'], skip, 256, 64 * (i + 1), amax=amax)
        else:
            out = torch.empty(B, T, Fq, 64, device=skip.device, dtype=torch.float32)
            mr = inorm_prelu_fwd(R, stats, P[f'{p}.norm{i+1}.weight'], P[f'{p}.norm{i+1}.bias'],
                                 P[f'{p}.prelu{i+1}.weight'], out, 64, 0, amax=amax)
        ctx['R'].append(R)
        ctx['mr'].append(mr)
    return out, ctx


def dense_block_bwd(P, G, p, ctx, dout, B, T, Fq):
    """returns dskip [B,T,Fq,256]; its slab 0 is the gradient of the block input."""
    skip = ctx['skip']
    dskip = torch.empty_like(skip)
    for i in (3, 2, 1, 0):
        C_in = 64 * (i + 1)
        if i == 3:
            dy, ldy, yoff = dout, 64, 0
        else:
            dy, ldy, yoff = dskip, 256, 64 * (i + 1)
        dR = inorm_prelu_bwd(ctx['R'][i], ctx['mr'][i], P[f'{p}.norm{i+1}.weight'], P[f'{p}.norm{i+1}.bias'],
                             P[f'{p}.prelu{i+1}.weight'], dy, ldy, yoff, G[f'{p}.norm{i+1}.weight'],
                             G[f'{p}.norm{i+1}.bias'], G[f'{p}.prelu{i+1}.weight'])
        conv_bwd(skip, B, T, Fq, 256, 0, C_in, P[f'{p}.conv{i+1}.weight'], dense_taps(i), dR, T, Fq,
                 G[f'{p}.conv{i+1}.weight'], G[f'{p}.conv{i+1}.bias'], rev=True, dx=dskip, lddx=256, dx_off=0,
                 accumulate=(i != 3), wd=_w(P, (f'{p}.conv{i+1}.weight', 'dgrad'), lambda: None), a_amax=ctx.get('amax'))
        ctx['R'][i] = None
    return dskip


# ------------------------------------------------------------------------------------------------
# DenseEncoder (models/generator.py:35-54)
# ------------------------------------------------------------------------------------------------
TAPS_1x1 = [(0, 0)]
TAPS_1x3 = [(0, -1), (0, 0), (0, 1)]
TAPS_1x2 = [(0, 0), (0, 1)]


def encoder_fwd(P, xin, B, T, Fq, p='dense_encoder'):
    ctx = {'xin': xin}
    if (THIN_CONV & 1) and P[f'{p}.conv_1.0.weight'].shape[:2] == (64, 3):
        R0 = torch.empty(B, T, Fq, 64, device=xin.device, dtype=torch.float32)
        st0 = O.zeros(B, 64, 2, device=xin.device, dtype=torch.float64)
        L.call('se_conv3to64_fwd', L.ptr(xin), L.ptr(P[f'{p}.conv_1.0.weight']), L.ptr(P[f'{p}.conv_1.0.bias']), L.ptr(R0), L.ptr(st0),
               C.c_int(B), C.c_long(T * Fq), L.stream(), _key='thin conv (generator)', _bytes=4.0 * B * T * Fq * 68)
    else:
        R0, st0 = conv_fwd(xin, B, T, Fq, 4, 0, 4,
                           _w(P, (f'{p}.conv_1.0.weight', 'fwd'), lambda: pack_w(P[f'{p}.conv_1.0.weight'], C_pad=4)),
                           P[f'{p}.conv_1.0.bias'], TAPS_1x1, 64)
    skip = torch.empty(B, T, Fq, 256, device=xin.device, dtype=torch.float32)
    blk = _amax(xin.device)                 # max |activation| of the skip stack / the block output: raised by every norm below
    ctx['mr0'] = inorm_prelu_fwd(R0, st0, P[f'{p}.conv_1.1.weight'], P[f'{p}.conv_1.1.bias'],
                                 P[f'{p}.conv_1.2.weight'], skip, 256, 0, amax=blk)
    ctx['R0'] = R0
    a2, ctx['dense'] = dense_block_fwd(P, f'{p}.dilated_dense', skip, B, T, Fq, amax=blk)
    Fo = (Fq + 2 - 3) // 2 + 1
    R5, st5 = conv_fwd(a2, B, T, Fq, 64, 0, 64,
                       _w(P, (f'{p}.conv_2.0.weight', 'fwd'), lambda: pack_w(P[f'{p}.conv_2.0.weight'])),
                       P[f'{p}.conv_2.0.bias'], TAPS_1x3, 64, To=T, Fo=Fo, sf=2, a_amax=blk)
    out = torch.empty(B, T, Fo, 64, device=xin.device, dtype=torch.float32)
    ctx['mr5'] = inorm_prelu_fwd(R5, st5, P[f'{p}.conv_2.1.weight'], P[f'{p}.conv_2.1.bias'],
                                 P[f'{p}.conv_2.2.weight'], out, 64, 0)
    ctx.update(a2=a2, R5=R5, Fo=Fo)
    return out, ctx


def encoder_bwd(P, G, ctx, dout, B, T, Fq, p='dense_encoder'):
    Fo = ctx['Fo']
    dR5 = inorm_prelu_bwd(ctx['R5'], ctx['mr5'], P[f'{p}.conv_2.1.weight'], P[f'{p}.conv_2.1.bias'],
                          P[f'{p}.conv_2.2.weight'], dout, 64, 0, G[f'{p}.conv_2.1.weight'],
                          G[f'{p}.conv_2.1.bias'], G[f'{p}.conv_2.2.weight'])
    da2 = conv_bwd(ctx['a2'], B, T, Fq, 64, 0, 64, P[f'{p}.conv_2.0.weight'], TAPS_1x3, dR5, T, Fo,
                   G[f'{p}.conv_2.0.weight'], G[f'{p}.conv_2.0.bias'], sf=2,
                   wd=_w(P, (f'{p}.conv_2.0.weight', 'dgrad'), lambda: None), a_amax=ctx['dense'].get('amax'))
    dskip = dense_block_bwd(P, G, f'{p}.dilated_dense', ctx['dense'], da2, B, T, Fq)
    dR0 = inorm_prelu_bwd(ctx['R0'], ctx['mr0'], P[f'{p}.conv_1.1.weight'], P[f'{p}.conv_1.1.bias'],
                          P[f'{p}.conv_1.2.weight'], dskip, 256, 0, G[f'{p}.conv_1.1.weight'],
                          G[f'{p}.conv_1.1.bias'], G[f'{p}.conv_1.2.weight'])
    if (THIN_CONV & 2) and P[f'{p}.conv_1.0.weight'].shape[:2] == (64, 3):
        with GM.leaf_stream(ctx['xin'], dR0):
            L.call('se_conv3to64_wgrad', L.ptr(ctx['xin']), L.ptr(dR0), L.ptr(G[f'{p}.conv_1.0.weight']), L.ptr(G[f'{p}.conv_1.0.bias']),
                   C.c_long(B * T * Fq), L.stream(), _key='thin conv (generator)', _bytes=4.0 * B * T * Fq * 68)
    else:
        conv_bwd(ctx['xin'], B, T, Fq, 4, 0, 4, P[f'{p}.conv_1.0.weight'], TAPS_1x1, dR0, T, Fq,
                 G[f'{p}.conv_1.0.weight'], G[f'{p}.conv_1.0.bias'], need_dx=False)
    return None


# ------------------------------------------------------------------------------------------------
# Conformer block (models/conformer.py:180-212) on the token view of [B, T, F', 64]
# ------------------------------------------------------------------------------------------------
def site_seed(base, idx):
    """dropout stream of one (layer, site): a 32-bit mix of the per-step base seed and the site index."""
    return (base * 0x9E3779B1 + (idx + 1) * 0x85EBCA6B + 0x1234567) & 0xFFFFFFFF


def _lin3(W, **kw):
    """precision / scale keywords of a token-wise GEMM whose weight W may be scaled fp16 planes (precision 3) -- then the A operand
    needs its scale: a_sexp (static: LayerNorm prologue) or a_amax (measured maximum of a gradient)"""
    if W.dtype == torch.float16:
        return dict(precision=3, **kw)
    return {}


def _bnd(P, key, sexp, need_k1=None):
    """operand-scale keywords of a bounded activation: the proven bound of this forward (weights.WeightPlan.run_bounds: a device
    scalar computed from the current parameters) when there is one, the static exponent otherwise (direct calls without a plan).
    need_k1: the sqrt(tokens - 1) a train-mode BatchNorm bound must have been derived with -- a plan whose last run_bounds used a
    smaller token count (another entry point, another batch) does not bound this call's data: static exponent then."""
    plan = P.get('__prep__')
    if plan is not None and plan.bounds_ready and key in plan.bounds and (need_k1 is None or plan.k1 >= need_k1 * (1.0 - 1e-6)):
        return dict(a_amax=plan.bounds[key])
    return dict(a_sexp=sexp)


def _amax(dev):
    """a zero-filled device scalar for a producer kernel to raise to max |output| (None when no scaled-fp16 kernel runs)"""
    return O.zeros(1, device=dev) if 3 in (GM.LINEAR_PRECISION, GM.WGRAD_LINEAR_PRECISION, CONV_PRECISION, WGRAD_PRECISION[0]) else None


def _ff_fwd(P, p, x, M, drop=0.0, seed_h=0, seed_o=0, st=None, want_out_stats=False):
    """x + 0.5 * Drop(W2 Drop(Swish(W1 LN(x)))) (Scale(0.5, PreNorm(FeedForward)), conformer.py:53-71,128-145).
    The two dropout masks are counter-based (hash(seed, element)) and re-evaluated in the backward.
    st: (mean, rstd) of the rows of x when the producer of x already emitted them (SE_EPI_ROWSTATS); want_out_stats: also
    return the statistics of the output rows (for the LayerNorm that reads them next) -- None when not produced here."""
    if st is None:
        st = O.row_stats(x, M)
    W1, W2 = P[f'{p}.fn.fn.net.0.weight'], P[f'{p}.fn.fn.net.3.weight']
    W1p = _w(P, (f'{p}.fn.fn.net.0.weight', 'lin'), lambda: W1)
    if W1p.dtype == torch.float16 and tuple(W1.shape) == (256, 64) and tuple(W2.shape) == (64, 256):
        # scaled fp16 planes from the step's WeightPlan: one fused kernel.  Default: H is neither stored nor read again (the fused
        # backward recomputes it; inference never needed it); SE_FF_FUSED=0: H is written once for the stored-H backward kernels
        res = GM.ff_fwd(x, st, P[f'{p}.fn.norm.weight'], P[f'{p}.fn.norm.bias'], W1p, P[f'{p}.fn.fn.net.0.bias'],
                        _w(P, (f'{p}.fn.fn.net.3.weight', 'lin'), lambda: W2), P[f'{p}.fn.fn.net.3.bias'], drop, seed_h, seed_o, 0.5,
                        out_stats=want_out_stats, store_h=not GM.FF_FUSED,
                        in_bound=_bnd(P, ('ln', p), 0).get('a_amax'), mid_bound=_bnd(P, ('hid', p), 0).get('a_amax'))
        if want_out_stats:
            y, z, ost = res
            return y, (x, st, z, drop, seed_h, seed_o), ost
        y, z = res
        return y, (x, st, z, drop, seed_h, seed_o)
    # plain fp32 weights (direct layer calls without a plan, SE_LINEAR_PRECISION != f16x3): two GEMMs with fused pro- / epilogues
    z = torch.empty(M, 256, device=x.device, dtype=torch.float32)
    GM.gemm_tap(GM.linear_desc(M, 64, 256, prologue=L.PRO_LN, epilogue=L.EPI_BIAS), x, P[f'{p}.fn.fn.net.0.weight'], z,
                bias=P[f'{p}.fn.fn.net.0.bias'], rowstats=st, ps=P[f'{p}.fn.norm.weight'], pb=P[f'{p}.fn.norm.bias'])
    y = torch.empty(M, 64, device=x.device, dtype=torch.float32)
    dr = drop > 0.0
    GM.gemm_tap(GM.linear_desc(M, 256, 64, prologue=L.PRO_SWISH_DROP if dr else L.PRO_SWISH,
                               epilogue=L.EPI_BIAS | L.EPI_RESID | (L.EPI_DROP if dr else 0), alpha=0.5, ldr=64,
                               pro_seed=seed_h, epi_seed=seed_o, drop_p=drop),
                z, P[f'{p}.fn.fn.net.3.weight'], y, bias=P[f'{p}.fn.fn.net.3.bias'], R=x)
    if want_out_stats:
        return y, (x, st, z, drop, seed_h, seed_o), None
    return y, (x, st, z, drop, seed_h, seed_o)


def _ff_bwd(P, G, p, saved, dy, M, dR2=None):
    """dy = gradient of (x + 0.5 FF(LN x)); returns dx (= dy + LN-path gradient (+ dR2))."""
    x, st, z, drop, seed_h, seed_o = saved
    dr = drop > 0.0
    W1, W2 = P[f'{p}.fn.fn.net.0.weight'], P[f'{p}.fn.fn.net.3.weight']
    W2Tp = _w(P, (f'{p}.fn.fn.net.3.weight', 'T0.5'), lambda: None)
    fused = W2Tp is not None and W2Tp.dtype == torch.float16 and getattr(dy, '_se_amax', None) is not None      # (as the forward: planes of the plan)
    if fused and z is None:
        # ONE launch: input gradient, LayerNorm backward and all four weight gradients, H recomputed from x (the weight gradients
        # are no longer leaves on the side stream: they accumulate inside the sweep that produces dx)
        return GM.ff_bwd_fused(dy, x, st, P[f'{p}.fn.norm.weight'], P[f'{p}.fn.norm.bias'], _w(P, (f'{p}.fn.fn.net.0.weight', 'lin'), lambda: None),
                               P[f'{p}.fn.fn.net.0.bias'], W2Tp,
                               G[f'{p}.fn.fn.net.0.weight'], G[f'{p}.fn.fn.net.0.bias'], G[f'{p}.fn.fn.net.3.weight'],
                               G[f'{p}.fn.fn.net.3.bias'], G[f'{p}.fn.norm.weight'], G[f'{p}.fn.norm.bias'], drop, seed_h, seed_o,
                               0.5, dR2=dR2, out_amax=_amax(dy.device), in_bound=_bnd(P, ('ln', p), 0).get('a_amax'),
                               mid_bound=_bnd(P, ('hid', p), 0).get('a_amax'))
    if z is None:
        raise L.SeHipError('_ff_bwd: the forward stored no H (fused path) but dy carries no measured maximum (dy._se_amax)')
    # dz = 0.5 * ((mask_o * dy) @ W2) * mask_h * swish'(z);  dh = dz @ W1
    if fused:
        # ... and the LayerNorm backward on the rows still in registers: dx = dy (+ dR2) + LNbwd(dz @ W1)
        dz, dx = GM.ff_bwd_dgrad(dy, z, _w(P, (f'{p}.fn.fn.net.3.weight', 'T0.5'), lambda: _T(W2) * 0.5),
                                 _w(P, (f'{p}.fn.fn.net.0.weight', 'T'), lambda: _T(W1)), drop, seed_h, seed_o,
                                 ln=(x, st, P[f'{p}.fn.norm.weight'], dR2, G[f'{p}.fn.norm.weight'], G[f'{p}.fn.norm.bias']),
                                 amax_out=(_amax(dy.device), _amax(dy.device)))
    else:
        dz = torch.empty(M, 256, device=x.device, dtype=torch.float32)
        GM.gemm_tap(GM.linear_desc(M, 64, 256, prologue=L.PRO_DROP if dr else L.PRO_NONE,
                                   epilogue=L.EPI_SWISH_GRAD | (L.EPI_DROP if dr else 0), ldx=256, pro_seed=seed_o,
                                   epi_seed=seed_h, drop_p=drop), dy, _T(W2) * 0.5, dz, AUX=z)
    # dW2 = 0.5 * (mask_o * dy)^T (mask_h * swish(z));  db2 = 0.5 * sum mask_o * dy
    dy_amax, dz_amax = getattr(dy, '_se_amax', None), getattr(dz, '_se_amax', None)
    with GM.leaf_stream(z, dy, x, dz, st, dy_amax, dz_amax):
        GM.gemm_tap_wgrad(GM.linear_desc(M, 256, 64, prologue=L.PRO_SWISH_DROP if dr else L.PRO_SWISH,
                                         epilogue=L.EPI_DROP if dr else 0, pro_seed=seed_h, epi_seed=seed_o, drop_p=drop,
                                         w_amax=dy_amax, **_bnd(P, ('hid', p), GM.HID_SEXP)),
                          z, dy, G[f'{p}.fn.fn.net.3.weight'], G[f'{p}.fn.fn.net.3.bias'], scale=0.5)
        # dW1 = dz^T LN(x);  db1 = sum dz
        GM.gemm_tap_wgrad(GM.linear_desc(M, 64, 256, prologue=L.PRO_LN, w_amax=dz_amax, **_bnd(P, ('ln', p), GM.LN_SEXP)), x, dz,
                          G[f'{p}.fn.fn.net.0.weight'], G[f'{p}.fn.fn.net.0.bias'], rowstats=st, ps=P[f'{p}.fn.norm.weight'],
                          pb=P[f'{p}.fn.norm.bias'])
    if fused:
        return dx
    dh = torch.empty(M, 64, device=x.device, dtype=torch.float32)
    GM.gemm_tap(GM.linear_desc(M, 256, 64), dz, _T(W1), dh)
    return O.layernorm_bwd(x, st, P[f'{p}.fn.norm.weight'], dh, G[f'{p}.fn.norm.weight'], G[f'{p}.fn.norm.bias'],
                           dR=dy, dR2=dR2)


def conformer_fwd(P, p, x, B, T, Fq, axis, train=True, dp=NO_DP, buffers=None, drop=(0.0, 0.0), seed=0, st_in=None):
    """x: [B*T*Fq, 64] tokens; returns LN(y4) + x (the TSCB adds the block input, generator.py:70,72).
    drop = (ff_dropout, attn_dropout) (generator.py:60-65: 0.2 / 0.2, conv dropout 0); seed: base of this
    block's five dropout streams."""
    M = x.shape[0]
    geom = A.seq_geometry(B, T, Fq, axis)
    ctx = {'geom': geom}
    pf, pa = (drop if train else (0.0, 0.0))
    # LayerNorm statistics of y1 / y2 / y3 come out of the kernels that produce those rows (se_ff_fwd_stats, SE_EPI_ROWSTATS):
    # three of the four se_row_stats passes of a block (26 us each, on the serial forward path) disappear
    # (st_in: the statistics of the rows of x, emitted by the previous block's post_norm: the fourth pass disappears too)
    r1 = _ff_fwd(P, f'{p}.ff1', x, M, pf, site_seed(seed, 0), site_seed(seed, 1), st=st_in, want_out_stats=FUSE_ROWSTATS)
    y1, ctx['ff1'], st2 = r1[0], r1[1], (r1[2] if len(r1) > 2 else None)
    # attention
    if st2 is None:
        st2 = O.row_stats(y1, M)
    Wqkv = _w(P, (f'{p}.attn.fn', 'qkv'),
              lambda: torch.cat([P[f'{p}.attn.fn.to_q.weight'], P[f'{p}.attn.fn.to_kv.weight']], 0).contiguous())
    qkv = torch.empty(M, 192, device=x.device, dtype=torch.float32)
    E = P[f'{p}.attn.fn.rel_pos_emb.weight']
    maxpos = (E.shape[0] - 1) // 2
    Es = _w(P, (f'{p}.attn.fn.rel_pos_emb.weight', 'es'), lambda: None)
    # scaled split-fp16 attention: the qkv GEMM raises max |qkv| for it; sequences outside that kernel (10 s clips) take the
    # three-way bf16 split / streaming kernels with the table split on the fly
    # (eval mode has no backward: the forward kernel alone also takes the 1601 frames of a 10 s utterance)
    attn16 = Es is not None and Es.dtype == torch.float16 and (A.f16_shape_ok(geom, maxpos) or
                                                               (not train and A.f16_fwd_shape_ok(geom, maxpos)))
    qkv_amax = O.zeros(1, device=x.device) if attn16 else None
    GM.gemm_tap(GM.linear_desc(M, 64, 192, prologue=L.PRO_LN, y_amax=qkv_amax, **_lin3(Wqkv, **_bnd(P, ('ln', f'{p}.attn'), GM.LN_SEXP))), y1, Wqkv, qkv,
                rowstats=st2, ps=P[f'{p}.attn.norm.weight'], pb=P[f'{p}.attn.norm.bias'])
    if Es is not None and Es.dtype == torch.float16 and not attn16:
        Es = None
    o, lse = A.attn_fwd(qkv, E, geom, maxpos=maxpos, scale=0.25, Es=Es, qkv_amax=qkv_amax, need_lse=train)
    y2 = torch.empty(M, 64, device=x.device, dtype=torch.float32)
    sa = site_seed(seed, 2)
    st3 = torch.empty(M, 2, device=x.device, dtype=torch.float32) if FUSE_ROWSTATS else None
    # the attention output is a convex combination of value rows: |o| <= max |v| <= max |qkv| -- the measured scalar of the qkv GEMM
    # is its operand scale (ATTN_O_SEXP only without one: the bf16 attention kernels)
    Wo_ = _w(P, (f'{p}.attn.fn.to_out.weight', 'lin'), lambda: P[f'{p}.attn.fn.to_out.weight'])
    GM.gemm_tap(GM.linear_desc(M, 64, 64, epilogue=L.EPI_BIAS | L.EPI_RESID | (L.EPI_DROP if pa > 0 else 0) |
                               (L.EPI_ROWSTATS if FUSE_ROWSTATS else 0), alpha=1.0, ldr=64, epi_seed=sa, drop_p=pa,
                               **_lin3(Wo_, **(dict(a_amax=qkv_amax) if qkv_amax is not None else dict(a_sexp=ATTN_O_SEXP)))), o,
                Wo_, y2, bias=P[f'{p}.attn.fn.to_out.bias'], R=y1, AUX=st3)
    ctx['attn'] = (y1, st2, Wqkv, qkv, o, lse, maxpos, pa, sa, qkv_amax)
    # conv module
    if st3 is None:
        st3 = O.row_stats(y2, M)
    u = torch.empty(M, 128, device=x.device, dtype=torch.float32)
    zc = torch.empty(M, 128, device=x.device, dtype=torch.float32)      # the gate half only (GLU backward: (u, gate))
    Wpw1 = _w(P, (f'{p}.conv.net.2.weight', 'lin'), lambda: P[f'{p}.conv.net.2.weight'].view(256, 64))
    GM.gemm_tap(GM.linear_desc(M, 64, 256, ldc=128, prologue=L.PRO_LN, epilogue=L.EPI_BIAS | L.EPI_GLU | L.EPI_GLU_GATE, ldx=128,
                               **_lin3(Wpw1, **_bnd(P, ('ln', f'{p}.conv'), GM.LN_SEXP))), y2,
                Wpw1, u, bias=P[f'{p}.conv.net.2.bias'], AUX=zc, rowstats=st3, ps=P[f'{p}.conv.net.0.weight'],
                pb=P[f'{p}.conv.net.0.bias'])
    Wdw = P[f'{p}.conv.net.4.conv.weight'].view(128, 31)
    g_bn, b_bn = P[f'{p}.conv.net.5.weight'], P[f'{p}.conv.net.5.bias']
    count = float(M * dp.world)
    if train:
        bnstats = O.zeros(1, 128, 2, device=x.device, dtype=torch.float64)
        h = O.dwconv31(u, Wdw, P[f'{p}.conv.net.4.conv.bias'], geom, stats=bnstats)
        dp.allreduce(bnstats)
        rm = buffers[f'{p}.conv.net.5.running_mean'] if buffers is not None else None
        rv = buffers[f'{p}.conv.net.5.running_var'] if buffers is not None else None
        mr, ss = O.norm_finalize(bnstats, g_bn, b_bn, 1, 128, count, running_mean=rm, running_var=rv, momentum=0.1)
        if buffers is not None:
            nbt = buffers.get('_nbt_pending')            # tscnet_fwd: the eight counters are incremented with ONE launch
            if nbt is not None:
                nbt.append(buffers[f'{p}.conv.net.5.num_batches_tracked'])
            else:
                buffers[f'{p}.conv.net.5.num_batches_tracked'] += 1
    else:
        h = O.dwconv31(u, Wdw, P[f'{p}.conv.net.4.conv.bias'], geom)
        mr, ss = O.bn_eval_scale(P[f'{p}.conv.net.5.running_mean'], P[f'{p}.conv.net.5.running_var'], g_bn, b_bn)
    sst = ss[0].t().contiguous()            # [2][128]: scale row, shift row (one small launch, not two)
    sc, sh = sst[0], sst[1]
    y3 = torch.empty(M, 64, device=x.device, dtype=torch.float32)
    # Swish(BatchNorm(h)) is bounded like the FF hidden activations (the weight gradient of this layer uses the same exponent)
    # (eval mode: running statistics bound nothing -- |BN(h)| depends on the data -- so the projection then runs on the fp32 kernel)
    bn_b = _bnd(P, ('bn', f'{p}.conv'), GM.HID_SEXP, need_k1=max(count - 1.0, 1.0) ** 0.5) if train else {}
    Wpw2 = _w(P, (f'{p}.conv.net.7.weight', 'lin'), lambda: None) if (train or 'a_amax' in bn_b) else None
    if Wpw2 is None:
        Wpw2 = P[f'{p}.conv.net.7.weight'].view(64, 128)
    st4 = torch.empty(M, 2, device=x.device, dtype=torch.float32) if FUSE_ROWSTATS else None
    GM.gemm_tap(GM.linear_desc(M, 128, 64, prologue=L.PRO_AFFINE_SWISH, epilogue=L.EPI_BIAS | L.EPI_RESID |
                               (L.EPI_ROWSTATS if FUSE_ROWSTATS else 0), alpha=1.0, ldr=64,
                               **_lin3(Wpw2, **bn_b)), h, Wpw2, y3,
                bias=P[f'{p}.conv.net.7.bias'], R=y2, ps=sc, pb=sh, AUX=st4)
    ctx['conv'] = (y2, st3, zc, u, h, mr, sc, sh, count)
    y4, ctx['ff2'] = _ff_fwd(P, f'{p}.ff2', y3, M, pf, site_seed(seed, 3), site_seed(seed, 4), st=st4)
    if FUSE_ROWSTATS:
        out, st5, ctx['out_stats'] = O.layernorm_fwd(y4, P[f'{p}.post_norm.weight'], P[f'{p}.post_norm.bias'], R=x, out_stats=True)
    else:
        out, st5 = O.layernorm_fwd(y4, P[f'{p}.post_norm.weight'], P[f'{p}.post_norm.bias'], R=x)
    ctx['post'] = (y4, st5)
    return out, ctx


def conformer_bwd(P, G, p, ctx, dout, B, T, Fq, dp=NO_DP, train=True):
    if not train:
        raise L.SeHipError('conformer_bwd in eval mode is not supported (BatchNorm uses running statistics)')
    M = dout.shape[0]
    geom = ctx['geom']
    dev = dout.device
    y4, st5 = ctx['post']
    dy4 = O.layernorm_bwd(y4, st5, P[f'{p}.post_norm.weight'], dout, G[f'{p}.post_norm.weight'],
                          G[f'{p}.post_norm.bias'], amax=_amax(dev))
    dy3 = _ff_bwd(P, G, f'{p}.ff2', ctx['ff2'], dy4, M)
    ctx['ff2'] = ctx['post'] = None
    # conv module: y3 = y2 + swish(bn(h)) @ Wpw2^T + b
    y2, st3, zc, u, h, mr, sc, sh, count = ctx['conv']
    Wpw2 = P[f'{p}.conv.net.7.weight'].view(64, 128)
    dact = torch.empty(M, 128, device=dev, dtype=torch.float32)
    Wpw2T = _w(P, (f'{p}.conv.net.7.weight', 'T'), lambda: _T(Wpw2))
    GM.gemm_tap(GM.linear_desc(M, 64, 128, **_lin3(Wpw2T, a_amax=getattr(dy3, '_se_amax', None))), dy3, Wpw2T, dact)
    with GM.leaf_stream(h, dy3, sc, sh, getattr(dy3, '_se_amax', None)):
        GM.gemm_tap_wgrad(GM.linear_desc(M, 128, 64, prologue=L.PRO_AFFINE_SWISH,
                                         **_bnd(P, ('bn', f'{p}.conv'), GM.HID_SEXP, need_k1=max(count - 1.0, 1.0) ** 0.5),
                                         w_amax=getattr(dy3, '_se_amax', None)), h, dy3,
                          G[f'{p}.conv.net.7.weight'].view(64, 128), G[f'{p}.conv.net.7.bias'], ps=sc, pb=sh)
    dh = torch.empty(M, 128, device=dev, dtype=torch.float32)
    g_bn, b_bn = P[f'{p}.conv.net.5.weight'], P[f'{p}.conv.net.5.bias']
    if train:
        O.norm_prelu_bwd(h, 128, 0, mr, g_bn, b_bn, None, dact, 128, 0, dh, 128, 0, G[f'{p}.conv.net.5.weight'],
                         G[f'{p}.conv.net.5.bias'], None, 1, M, 128, per_batch=False, act=1,
                         allreduce=(dp.allreduce if dp.world > 1 or getattr(dp, 'force_sync', False) else None), count=count)
    else:
        raise L.SeHipError('conformer_bwd in eval mode is not supported (BatchNorm uses running statistics)')
    Wdw = P[f'{p}.conv.net.4.conv.weight'].view(128, 31)
    dw_fused = FUSE_GLU_BWD and O.DW_BWD_FUSED and M * 1024 < (1 << 32) - 4096
    if dw_fused:           # ... and the weight / bias gradient from the rows the same kernel already holds (round 5)
        dzc = O.dwconv31_bwd_fused(dh, Wdw, u, zc, G[f'{p}.conv.net.4.conv.weight'].view(128, 31), G[f'{p}.conv.net.4.conv.bias'], geom,
                                   amax=_amax(dev))
        du = None
    elif FUSE_GLU_BWD:     # depthwise input gradient + GLU backward in one kernel: dU (266 MB at batch 16) never goes to memory
        dzc = O.dwconv31_glu_bwd(dh, Wdw, u, zc, geom, amax=_amax(dev))
        du = None
    else:
        du = O.dwconv31(dh, Wdw, None, geom, flip=True)
    if not dw_fused:
        with GM.leaf_stream(u, dh):
            O.dwconv31_wgrad(u, dh, G[f'{p}.conv.net.4.conv.weight'].view(128, 31), G[f'{p}.conv.net.4.conv.bias'], geom)
    if not FUSE_GLU_BWD:
        dzc = O.glu_bwd_gate(u, zc, du, M, 128, amax=_amax(dev))
    Wpw1 = P[f'{p}.conv.net.2.weight'].view(256, 64)
    Wpw1T = _w(P, (f'{p}.conv.net.2.weight', 'T'), lambda: _T(Wpw1))
    fused_pw1 = GM.LNBWD_FUSED and FUSE_LN_BWD and Wpw1T.dtype == torch.float16 and getattr(dzc, '_se_amax', None) is not None
    if not fused_pw1:
        with GM.leaf_stream(y2, dzc, st3, dzc._se_amax):
            GM.gemm_tap_wgrad(GM.linear_desc(M, 64, 256, prologue=L.PRO_LN, w_amax=dzc._se_amax, **_bnd(P, ('ln', f'{p}.conv'), GM.LN_SEXP)), y2, dzc,
                              G[f'{p}.conv.net.2.weight'].view(256, 64),
                              G[f'{p}.conv.net.2.bias'], rowstats=st3, ps=P[f'{p}.conv.net.0.weight'],
                              pb=P[f'{p}.conv.net.0.bias'])
    if fused_pw1:
        # ONE sweep over the rows for the input gradient, the LayerNorm backward and the weight / bias gradient (se_lnbwd_fused.hip)
        dy2 = GM.gemm_ln_bwd_wgrad(dzc, Wpw1T, y2, st3, P[f'{p}.conv.net.0.weight'], P[f'{p}.conv.net.0.bias'], dy3,
                                   G[f'{p}.conv.net.0.weight'], G[f'{p}.conv.net.0.bias'], G[f'{p}.conv.net.2.weight'].view(256, 64),
                                   G[f'{p}.conv.net.2.bias'], out_amax=_amax(dev), in_bound=_bnd(P, ('ln', f'{p}.conv'), 0).get('a_amax'))
    elif FUSE_LN_BWD and GM.LINEAR_PRECISION in (2, 3):
        # input-gradient GEMM + the LayerNorm backward on its accumulators: the [M, 64] product never goes to memory
        dy2 = GM.gemm_ln_bwd(dzc, Wpw1T, y2, st3, P[f'{p}.conv.net.0.weight'], dy3, G[f'{p}.conv.net.0.weight'],
                             G[f'{p}.conv.net.0.bias'], out_amax=_amax(dev))
    else:
        dl3 = torch.empty(M, 64, device=dev, dtype=torch.float32)
        GM.gemm_tap(GM.linear_desc(M, 256, 64, **_lin3(Wpw1T, a_amax=dzc._se_amax)), dzc, Wpw1T, dl3)
        dy2 = O.layernorm_bwd(y2, st3, P[f'{p}.conv.net.0.weight'], dl3, G[f'{p}.conv.net.0.weight'],
                              G[f'{p}.conv.net.0.bias'], dR=dy3, amax=_amax(dev))
        del dl3
    ctx['conv'] = None
    del dact, dh, du, dzc, dy3, dy4
    # attention: y2 = y1 + o @ Wo^T + bo
    y1, st2, Wqkv, qkv, o, lse, maxpos, pa, sa, qkv_amax = ctx['attn']
    Wo = P[f'{p}.attn.fn.to_out.weight']
    do = torch.empty(M, 64, device=dev, dtype=torch.float32)
    WoT = _w(P, (f'{p}.attn.fn.to_out.weight', 'T'), lambda: _T(Wo))
    do_amax = O.zeros(1, device=dev) if qkv_amax is not None else None      # max |dO| for the scaled split-fp16 attention backward
    # the softmax-backward row constants delta = rowsum(dO . O) per head leave this GEMM's epilogue (it holds dO, reads O: EPI_DELTA)
    # whenever the cooperative backward will read them: no stand-alone pass over dO and O
    delta = torch.empty(M, 4, device=dev, dtype=torch.float32) if (qkv_amax is not None and A.f16_shape_ok(geom, maxpos)) else None
    GM.gemm_tap(GM.linear_desc(M, 64, 64, prologue=L.PRO_DROP if pa > 0 else L.PRO_NONE, pro_seed=sa, drop_p=pa, y_amax=do_amax,
                               epilogue=L.EPI_DELTA if delta is not None else 0, ldr=64 if delta is not None else 0,
                               **_lin3(WoT, a_amax=getattr(dy2, '_se_amax', None))), dy2, WoT, do, R=o if delta is not None else None, AUX=delta)
    with GM.leaf_stream(o, dy2, getattr(dy2, '_se_amax', None), qkv_amax):       # every scalar the side-stream kernel reads stays referenced until the join
        GM.gemm_tap_wgrad(GM.linear_desc(M, 64, 64, epilogue=L.EPI_DROP if pa > 0 else 0, epi_seed=sa, drop_p=pa,
                                         w_amax=getattr(dy2, '_se_amax', None),
                                         **(dict(a_amax=qkv_amax) if qkv_amax is not None else dict(a_sexp=ATTN_O_SEXP))), o, dy2,
                          G[f'{p}.attn.fn.to_out.weight'], G[f'{p}.attn.fn.to_out.bias'])
    dqkv = A.attn_bwd(qkv, P[f'{p}.attn.fn.rel_pos_emb.weight'], o, do, lse, geom,
                      G[f'{p}.attn.fn.rel_pos_emb.weight'], maxpos=maxpos, scale=0.25,
                      leaf=GM.leaf_stream, qkv_amax=qkv_amax, do_amax=do_amax, dqkv_amax=_amax(dev) if qkv_amax is not None else None, delta=delta)
    dq_amax = getattr(dqkv, '_se_amax', None)
    # plan keys in order of preference: the fp16 planes exist only under LINEAR_PRECISION 3 + fp16 attention ('qkvT16'); with the
    # bf16 linear kernels next to the fp16 attention backward the six-product planes ('qkvT') serve; no plan: fp32 from the parameters
    WqkvT = _w(P, (f'{p}.attn.fn', 'qkvT16'), lambda: None) if dq_amax is not None else None
    if WqkvT is None:
        WqkvT = _w(P, (f'{p}.attn.fn', 'qkvT'),
                   lambda: _T(torch.cat([P[f'{p}.attn.fn.to_q.weight'], P[f'{p}.attn.fn.to_kv.weight']], 0)))
    if WqkvT.dtype != torch.float16:
        dq_amax = None
    gq, gkv = G[f'{p}.attn.fn.to_q.weight'], G[f'{p}.attn.fn.to_kv.weight']
    # to_q / to_kv are neighbours in the flat gradient buffer of the optimizers: the [192, 64] gradient of the fused projection
    # then accumulates in place; otherwise through a scratch matrix
    adjacent = gq.is_contiguous() and gkv.is_contiguous() and gq.data_ptr() + gq.numel() * 4 == gkv.data_ptr()
    dWqkv = torch.as_strided(gq, (192, 64), (64, 1)) if adjacent and gq.untyped_storage().nbytes() - gq.storage_offset() * 4 >= 192 * 64 * 4 \
        else O.zeros(192, 64, device=dev)
    fused_qkv = GM.LNBWD_FUSED and FUSE_LN_BWD and WqkvT.dtype == torch.float16 and dq_amax is not None
    if not fused_qkv:
        with GM.leaf_stream(y1, dqkv, st2, dq_amax):
            GM.gemm_tap_wgrad(GM.linear_desc(M, 64, 192, prologue=L.PRO_LN, w_amax=dq_amax, **_bnd(P, ('ln', f'{p}.attn'), GM.LN_SEXP)), y1, dqkv, dWqkv, None,
                              rowstats=st2, ps=P[f'{p}.attn.norm.weight'], pb=P[f'{p}.attn.norm.bias'])
            if dWqkv.data_ptr() != gq.data_ptr():
                gq += dWqkv[:64]
                gkv += dWqkv[64:]
    if fused_qkv:
        # (its OUTPUT dy1 feeds the scaled-fp16 feed-forward backward -> max |dy1|)
        dy1 = GM.gemm_ln_bwd_wgrad(dqkv, WqkvT, y1, st2, P[f'{p}.attn.norm.weight'], P[f'{p}.attn.norm.bias'], dy2,
                                   G[f'{p}.attn.norm.weight'], G[f'{p}.attn.norm.bias'], dWqkv, None, out_amax=_amax(dev),
                                   in_bound=_bnd(P, ('ln', f'{p}.attn'), 0).get('a_amax'))
        if dWqkv.data_ptr() != gq.data_ptr():
            gq += dWqkv[:64]
            gkv += dWqkv[64:]
    elif FUSE_LN_BWD and GM.LINEAR_PRECISION in (2, 3):
        # (its OUTPUT dy1 feeds the scaled-fp16 feed-forward backward -> max |dy1|)
        dy1 = GM.gemm_ln_bwd(dqkv, WqkvT, y1, st2, P[f'{p}.attn.norm.weight'], dy2, G[f'{p}.attn.norm.weight'],
                             G[f'{p}.attn.norm.bias'], out_amax=_amax(dev))
    else:
        dl2 = torch.empty(M, 64, device=dev, dtype=torch.float32)
        GM.gemm_tap(GM.linear_desc(M, 192, 64, **_lin3(WqkvT, a_amax=dq_amax)), dqkv, WqkvT, dl2)
        dy1 = O.layernorm_bwd(y1, st2, P[f'{p}.attn.norm.weight'], dl2, G[f'{p}.attn.norm.weight'],
                              G[f'{p}.attn.norm.bias'], dR=dy2, amax=_amax(dev))
        del dl2
    ctx['attn'] = None
    del do, dqkv, dy2
    # ff1, plus the TSCB residual (out = LN(y4) + x)
    dx = _ff_bwd(P, G, f'{p}.ff1', ctx['ff1'], dy1, M, dR2=dout)
    ctx['ff1'] = None
    return dx


# ------------------------------------------------------------------------------------------------
# decoders (models/generator.py:77-129)
# ------------------------------------------------------------------------------------------------
def _decoder_head_fwd(P, p, x, B, T, Fq):
    skip = torch.empty(B, T, Fq, 256, device=x.device, dtype=torch.float32)
    # strided slab copy (the only one per decoder) + max |x|: the TSCB output (LayerNorm + residual chain) has no bound by
    # construction, and it is the first operand of the block's scaled split-fp16 convolutions
    blk = _amax(x.device)
    O.copy_cols_amax(x.view(B * T * Fq, 64), 64, skip, 256, B * T * Fq, 64, amax=blk)
    return dense_block_fwd(P, f'{p}.dense_block', skip, B, T, Fq, amax=blk)


def mask_decoder_fwd(P, x, B, T, Fq):
    p = 'mask_decoder'
    ctx = {}
    d4, ctx['dense'] = _decoder_head_fwd(P, p, x, B, T, Fq)
    S, _ = conv_fwd(d4, B, T, Fq, 64, 0, 64,
                    _w(P, (f'{p}.sub_pixel.conv.weight', 'fwd'), lambda: pack_w(P[f'{p}.sub_pixel.conv.weight'])),
                    P[f'{p}.sub_pixel.conv.bias'], TAPS_1x3, 128, shuffle2=True, want_stats=False,
                    a_amax=ctx['dense'].get('amax'))                                                      # [B,T,2Fq,64]
    F2 = 2 * Fq
    Fo = F2 - 1
    if THIN_CONV & 4:
        r, st = conv1x2_fwd(S, P[f'{p}.conv_1.weight'], P[f'{p}.conv_1.bias'], B, T, F2, True)       # [B,T,Fo,4], channel 0
    else:
        w1 = _w(P, (f'{p}.conv_1.weight', 'fwd'), lambda: pack_w(pad_rows(P[f'{p}.conv_1.weight'], 4)))
        b1 = pad_rows(P[f'{p}.conv_1.bias'], 4)
        r, st = conv_fwd(S, B, T, F2, 64, 0, 64, w1, b1, TAPS_1x2, 4, To=T, Fo=Fo)            # [B,T,Fo,4], channel 0
    g4, be4, a4 = pad_rows(P[f'{p}.norm.weight'], 4), pad_rows(P[f'{p}.norm.bias'], 4), pad_rows(P[f'{p}.prelu.weight'], 4)
    uact = torch.empty_like(r)
    mr = inorm_prelu_fwd(r, st, g4, be4, a4, uact, 4, 0)
    wb = torch.cat([P[f'{p}.final_conv.weight'].view(1), P[f'{p}.final_conv.bias'].view(1)])
    mask = O.mask_tail(uact, 4, wb, P[f'{p}.prelu_out.weight'], B * T * Fo, Fo)
    ctx.update(d4=d4, S=S, r=r, mr=mr, uact=uact, wb=wb, pads=(g4, be4, a4), Fo=Fo)
    return mask, ctx


def mask_decoder_bwd(P, G, ctx, dmask, B, T, Fq):
    p = 'mask_decoder'
    Fo, F2 = ctx['Fo'], 2 * Fq
    dev = dmask.device
    n = B * T * Fo
    duact = torch.empty(B, T, Fo, 4, device=dev, dtype=torch.float32)
    dwb = O.zeros(2, device=dev, dtype=torch.float64)
    O.mask_tail_bwd(ctx['uact'], 4, ctx['wb'], P[f'{p}.prelu_out.weight'], dmask, duact, dwb,
                    G[f'{p}.prelu_out.weight'], n, Fo)
    G[f'{p}.final_conv.weight'] += dwb[0].float().view(1, 1, 1, 1)
    G[f'{p}.final_conv.bias'] += dwb[1].float().view(1)
    g4, be4, a4 = ctx['pads']
    dg4, db4, da4 = (O.zeros(4, device=dev) for _ in range(3))
    dr = inorm_prelu_bwd(ctx['r'], ctx['mr'], g4, be4, a4, duact, 4, 0, dg4, db4, da4)
    G[f'{p}.norm.weight'] += dg4[:1]
    G[f'{p}.norm.bias'] += db4[:1]
    G[f'{p}.prelu.weight'] += da4[:1]
    if THIN_CONV & 8:
        dS = conv1x2_bwd(ctx['S'], P[f'{p}.conv_1.weight'], dr, G[f'{p}.conv_1.weight'], G[f'{p}.conv_1.bias'], B, T, F2)
    else:
        dw1 = O.zeros(4, 64, 1, 2, device=dev)
        dbias1 = O.zeros(4, device=dev)
        dS = conv_bwd(ctx['S'], B, T, F2, 64, 0, 64, pad_rows(P[f'{p}.conv_1.weight'], 4), TAPS_1x2, dr, T, Fo, dw1, dbias1,
                      wd=_w(P, (f'{p}.conv_1.weight', 'dgrad'), lambda: None))
        with GM.leaf_stream(dw1, dbias1):            # dw1 / dbias1 are written on the weight-gradient stream (conv_bwd)
            G[f'{p}.conv_1.weight'] += dw1[:1]
            G[f'{p}.conv_1.bias'] += dbias1[:1]
    dd4 = _subpixel_bwd(P, G, f'{p}.sub_pixel', ctx['d4'], dS, B, T, Fq, a_amax=ctx['dense'].get('amax'))
    dskip = dense_block_bwd(P, G, f'{p}.dense_block', ctx['dense'], dd4, B, T, Fq)
    return dskip        # slab 0 = input gradient


def _subpixel_bwd(P, G, p, x, dS, B, T, Fq, a_amax=None):
    """SPConvTranspose2d backward: dS [B,T,2Fq,64] is the un-shuffled gradient; view it as [B,T,Fq,128] with
    channel r*64+c <- pixel 2f+r (the pixel shuffle is a pure re-indexing of the same memory)."""
    dconv = dS.view(B, T, Fq, 128)                      # [.., f, (r, c)] : memory order is already (f, r, c)
    dconv._se_amax = getattr(dS, '_se_amax', None)      # the producer's max |dS| travels with the view
    w = P[f'{p}.conv.weight']
    dw = G[f'{p}.conv.weight']
    return conv_bwd(x, B, T, Fq, 64, 0, 64, w, TAPS_1x3, dconv, T, Fq, dw, G[f'{p}.conv.bias'],
                    wd=_w(P, (f'{p}.conv.weight', 'dgrad'), lambda: None), a_amax=a_amax)


def complex_decoder_fwd(P, x, B, T, Fq):
    p = 'complex_decoder'
    ctx = {}
    d4, ctx['dense'] = _decoder_head_fwd(P, p, x, B, T, Fq)
    S, st = conv_fwd(d4, B, T, Fq, 64, 0, 64,
                     _w(P, (f'{p}.sub_pixel.conv.weight', 'fwd'), lambda: pack_w(P[f'{p}.sub_pixel.conv.weight'])),
                     P[f'{p}.sub_pixel.conv.bias'], TAPS_1x3, 128, shuffle2=True, want_stats=True, a_amax=ctx['dense'].get('amax'))
    F2 = 2 * Fq
    Fo = F2 - 1
    a = torch.empty_like(S)
    mr = inorm_prelu_fwd(S, st, P[f'{p}.norm.weight'], P[f'{p}.norm.bias'], P[f'{p}.prelu.weight'], a, 64, 0)
    if THIN_CONV & 4:
        cplx, _ = conv1x2_fwd(a, P[f'{p}.conv.weight'], P[f'{p}.conv.bias'], B, T, F2, False)
    else:
        wc = _w(P, (f'{p}.conv.weight', 'fwd'), lambda: pack_w(pad_rows(P[f'{p}.conv.weight'], 4)))
        bc = pad_rows(P[f'{p}.conv.bias'], 4)
        cplx, _ = conv_fwd(a, B, T, F2, 64, 0, 64, wc, bc, TAPS_1x2, 4, To=T, Fo=Fo, want_stats=False)
    ctx.update(d4=d4, S=S, mr=mr, a=a, Fo=Fo)
    return cplx, ctx


def complex_decoder_bwd(P, G, ctx, dcplx, B, T, Fq):
    p = 'complex_decoder'
    Fo, F2 = ctx['Fo'], 2 * Fq
    dev = dcplx.device
    if THIN_CONV & 8:
        da = conv1x2_bwd(ctx['a'], P[f'{p}.conv.weight'], dcplx.contiguous(), G[f'{p}.conv.weight'], G[f'{p}.conv.bias'], B, T, F2)
    else:
        dwc = O.zeros(4, 64, 1, 2, device=dev)
        dbc = O.zeros(4, device=dev)
        da = conv_bwd(ctx['a'], B, T, F2, 64, 0, 64, pad_rows(P[f'{p}.conv.weight'], 4), TAPS_1x2, dcplx, T, Fo, dwc, dbc,
                      wd=_w(P, (f'{p}.conv.weight', 'dgrad'), lambda: None))
        with GM.leaf_stream(dwc, dbc):               # written on the weight-gradient stream (conv_bwd)
            G[f'{p}.conv.weight'] += dwc[:2]
            G[f'{p}.conv.bias'] += dbc[:2]
    dS = inorm_prelu_bwd(ctx['S'], ctx['mr'], P[f'{p}.norm.weight'], P[f'{p}.norm.bias'], P[f'{p}.prelu.weight'], da,
                         64, 0, G[f'{p}.norm.weight'], G[f'{p}.norm.bias'], G[f'{p}.prelu.weight'])
    dd4 = _subpixel_bwd(P, G, f'{p}.sub_pixel', ctx['d4'], dS, B, T, Fq, a_amax=ctx['dense'].get('amax'))
    return dense_block_bwd(P, G, f'{p}.dense_block', ctx['dense'], dd4, B, T, Fq)


# ------------------------------------------------------------------------------------------------
# TSCNet (models/generator.py:132-167)
# ------------------------------------------------------------------------------------------------
def tscnet_fwd(P, xin, train=True, dp=NO_DP, buffers=None, drop=(0.0, 0.0), seed=0):
    """xin: planes [B, T, F, 4] = (|x|, Re x, Im x, 0) of the compressed noisy spectrum.
    returns est planes [B, T, F, 4] = (|est|, Re est, Im est, 0) and the ctx for tscnet_bwd."""
    B, T, Fq, _ = xin.shape
    ctx = {'xin': xin, 'dims': (B, T, Fq)}
    plan = P.get('__prep__')
    if plan is not None:
        # proven bounds of the normalised activations of this forward (LayerNorm outputs, feed-forward hidden activations, train-mode
        # BatchNorm outputs: |x_hat| <= sqrt(tokens over all ranks - 1)) from the current parameters: one launch
        Fp_ = (Fq + 2 - 3) // 2 + 1
        plan.run_bounds(k1=float(max(B * T * Fp_ * dp.world - 1, 1)) ** 0.5)
    x, ctx['enc'] = encoder_fwd(P, xin, B, T, Fq)
    Fp = x.shape[2]
    ctx['Fp'] = Fp
    tok = x.view(B * T * Fp, 64)
    ctx['tscb'] = []
    st_tok = None                  # row statistics of `tok`, handed from one block's post_norm to the next block's first LayerNorm
    if buffers is not None and train:
        buffers = dict(buffers)
        buffers['_nbt_pending'] = []
    for i in range(1, 5):
        tok, c1 = conformer_fwd(P, f'TSCB_{i}.time_conformer', tok, B, T, Fp, 'time', train, dp, buffers, drop,
                                site_seed(seed, 100 + 2 * i), st_in=st_tok)
        st_tok = c1.pop('out_stats', None)
        tok, c2 = conformer_fwd(P, f'TSCB_{i}.freq_conformer', tok, B, T, Fp, 'freq', train, dp, buffers, drop,
                                site_seed(seed, 101 + 2 * i), st_in=st_tok)
        st_tok = c2.pop('out_stats', None)
        ctx['tscb'].append((c1, c2))
    if buffers is not None and buffers.get('_nbt_pending'):
        torch._foreach_add_(buffers['_nbt_pending'], 1)
    # the two decoders are independent branches (models/generator.py:154-156): the complex decoder runs on a second stream
    with GM.branch_stream(tok) as br:
        cplx, ctx['cplx'] = complex_decoder_fwd(P, tok, B, T, Fp)
    mask, ctx['mask'] = mask_decoder_fwd(P, tok, B, T, Fp)
    br.join()
    est = O.assemble(mask, 1, xin, cplx)
    ctx['est'] = est
    return est, ctx


def tscnet_bwd(P, G, ctx, dest, dp=NO_DP):
    """dest: gradient w.r.t. the est planes (dmag, dre, dim, -).  Parameter gradients are added into G."""
    B, T, Fq = ctx['dims']
    Fp = ctx['Fp']
    dev = dest.device
    dmask = torch.empty(B * T * Fq, device=dev, dtype=torch.float32)
    dcplx = torch.empty(B, T, Fq, 4, device=dev, dtype=torch.float32)
    O.assemble_bwd(ctx['est'], dest, ctx['xin'], dmask, 1, dcplx)
    with GM.branch_stream(dcplx) as br:       # the decoder branches are independent in the backward too
        dsk_c = complex_decoder_bwd(P, G, ctx['cplx'], dcplx, B, T, Fp)
    dsk_m = mask_decoder_bwd(P, G, ctx['mask'], dmask, B, T, Fp)
    br.join()
    dtok = (dsk_c[..., :64] + dsk_m[..., :64]).reshape(B * T * Fp, 64)      # plumbing: sum of the two decoder branches
    del dsk_c, dsk_m, dcplx
    ctx['cplx'] = ctx['mask'] = None
    for i in (4, 3, 2, 1):
        c1, c2 = ctx['tscb'][i - 1]
        dtok = conformer_bwd(P, G, f'TSCB_{i}.freq_conformer', c2, dtok, B, T, Fp, dp)
        dtok = conformer_bwd(P, G, f'TSCB_{i}.time_conformer', c1, dtok, B, T, Fp, dp)
        ctx['tscb'][i - 1] = None
    encoder_bwd(P, G, ctx['enc'], dtok.view(B, T, Fp, 64), B, T, Fq)
    return None
