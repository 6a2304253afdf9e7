"""Python side of the tap-GEMM family (csrc/se_gemm.hip): descriptors, weight packing, launches."""
import ctypes as C

import torch

from . import _lib as L
from ._lib import GemmDesc


def conv_taps(kh, kw, dil=(1, 1), pad=(0, 0)):
    """tap offsets (dt, df) of a cross-correlation kernel, tap index = kh_i*kw + kw_i."""
    return [(i * dil[0] - pad[0], j * dil[1] - pad[1]) for i in range(kh) for j in range(kw)]


def make_desc(B, To, Fo, Ti, Fi, taps, C_in, lda, N, ldc, a_off=0, c_off=0, ldw=None, st=1, sf=1, up=0,
              prologue=L.PRO_NONE, epilogue=0, alpha=1.0, ldr=0, r_off=0, ldx=0, x_off=0, pro_seed=0, epi_seed=0,
              drop_p=0.0, precision=0, a_sexp=0, w_sexp=0, a_amax=None, w_amax=None, y_amax=None):
    d = GemmDesc()
    d.B, d.To, d.Fo, d.Ti, d.Fi = B, To, Fo, Ti, Fi
    d.st, d.sf, d.up = st, sf, up
    d.ntap = len(taps)
    for i, (dt, df) in enumerate(taps):
        d.dt[i], d.df[i] = dt, df
    d.C, d.lda, d.a_off = C_in, lda, a_off
    d.N, d.ldc, d.c_off = N, ldc, c_off
    d.ldw = ldw if ldw is not None else len(taps) * C_in
    d.prologue, d.epilogue, d.alpha = prologue, epilogue, alpha
    d.ldr, d.r_off, d.ldx, d.x_off = ldr, r_off, ldx, x_off
    d.pro_seed, d.epi_seed, d.drop_p = pro_seed & 0xFFFFFFFF, epi_seed & 0xFFFFFFFF, drop_p
    d.precision = precision
    d.a_sexp, d.w_sexp = a_sexp, w_sexp
    d.a_amax = a_amax.data_ptr() if a_amax is not None else None
    d.w_amax = w_amax.data_ptr() if w_amax is not None else None
    d.y_amax = y_amax.data_ptr() if y_amax is not None else None      # raised to max |Y| by the vector epilogue
    d._keep = (a_amax, w_amax, y_amax)        # the descriptor holds raw pointers: keep the scalars alive with it
    return d


# Arithmetic of the token-wise K = 64 -> N >= 128 layers (FF in-projection, qkv, pointwise-GLU): 'bf16x6' (default) routes
# them to the row-panel kernel (exact three-way bf16 split, six MFMAs per product, fp32-equivalent); 'f32' keeps the
# per-column-block fp32-MFMA kernel; 'bf16x3' is the opt-in two-way split.  The prologue-free K >= 128 input gradients use
# the generic split kernel; everything else (prologue GEMMs with K > 64, N = 64 with K = 64, accumulating epilogues, weight
# gradients) measured faster on the fp32-MFMA kernels and stays there.
# 'f16x3' (default since round 3) = the SCALED split-fp16 arithmetic of se_gemm_desc precision 3 wherever the operand scales are
# known (pre-split fp16 weight planes from the step's WeightPlan; LayerNorm outputs with a static exponent; gradients with the
# maximum their producer measured): fp32-equivalent like bf16x6 at half the MFMAs.  Calls without those scales (plain fp32
# weights, e.g. direct layer calls in tests) run the six-product kernels.
LINEAR_PRECISION = {'f32': 0, 'bf16x3': 1, 'bf16x6': 2, 'f16x3': 3}[__import__('os').environ.get('SE_LINEAR_PRECISION', 'f16x3')]
WGRAD_LINEAR_PRECISION = {'f32': 0, 'bf16x3': 2, 'bf16x6': 2, 'f16x3': 3}[__import__('os').environ.get('SE_WGRAD_PRECISION', 'f16x3')]      # (token-wise weight gradients: the convolutions' switch)
LN_SEXP = 6        # LayerNorm(64) outputs: |x| <= 7.94 |gamma| + |beta|; 2^6 keeps |x| < 1023 below the fp16 maximum
HID_SEXP = 3       # Swish(H) * dropout mask (FF hidden activations): |x| < 8191


def linear_desc(M, C_in, N, lda=None, ldc=None, **kw):
    """plain row GEMM: M rows, one tap."""
    if 'precision' not in kw:
        second_operand = kw.get('epilogue', 0) & L.EPI_ACCUM
        wide_k = C_in >= 128 and kw.get('prologue', L.PRO_NONE) == L.PRO_NONE      # K >= 128 -> 64 input gradients: 146 vs 181 us
        kw['precision'] = min(LINEAR_PRECISION, 2) if ((C_in == 64 and N >= 128 and not second_operand) or wide_k) else 0
    return make_desc(1, 1, M, 1, M, [(0, 0)], C_in, lda or C_in, N, ldc or N, **kw)


def gemm_tap(d, A, W, Y, bias=None, R=None, AUX=None, rowstats=None, ps=None, pb=None, stats=None):
    L.check_cuda(A, W, Y, bias, R, AUX, rowstats, ps, pb, stats)
    M = d.B * d.To * d.Fo
    # the weight-form fields are written into a COPY: in gemm_tap_wgrad the descriptor's w_amax means the scale of dY (and switches
    # precision 3 on), so a caller's descriptor reused there must not carry this call's weight scalar
    keep = getattr(d, '_keep', None)
    d = type(d).from_buffer_copy(d)
    d._keep = keep
    if W.dtype == torch.bfloat16:        # weights pre-split by the step's WeightPlan: [3 planes][rows][ld] bf16
        if d.precision != 2:
            raise L.SeHipError('gemm_tap: bf16 weight planes need a precision-2 descriptor')
        d.w_planes, d.ldw = W.stride(0), W.shape[2]
    elif W.dtype == torch.float16:       # precision 3: [2 planes][rows][ld] scaled fp16 + the scalar they were scaled by
        if d.precision != 3:
            raise L.SeHipError('gemm_tap: scaled fp16 weight planes need a precision-3 descriptor (operand scales known)')
        if not d.a_amax and not d.a_sexp:
            raise L.SeHipError('gemm_tap: precision 3 needs the scale of the A operand (a_amax or a static a_sexp)')
        d.w_planes, d.ldw = W.stride(0), W.shape[2]
        d.w_amax = W._se_amax.data_ptr()
    else:
        d.w_planes = 0
    L.call('se_gemm_tap', C.byref(d), L.ptr(A), L.ptr(W), L.ptr(bias), L.ptr(Y), L.ptr(R), L.ptr(AUX),
           L.ptr(rowstats), L.ptr(ps), L.ptr(pb), L.ptr(stats), L.stream(),
           _key=(('conv3_f16x3' if d.precision == 3 else f'conv3_bf16x{3 if d.precision == 1 else 6}')
                 if d.precision in (1, 2, 3) and d.C >= 32 and d.prologue == 0 and d.ntap >= 3 and d.ntap % 3 == 0 and not d.up
                 and d.st == 1 and d.sf == 1 and d.Ti == d.To and d.Fi == d.Fo and not d.epilogue & (L.EPI_GLU | L.EPI_DROP) else
                 (f'gemm_k64_panel_f16x3<{d.prologue}>' if d.precision == 3 else f'gemm_k64_panel_bf16x{3 if d.precision == 1 else 6}<{d.prologue}>')
                 if d.precision in (1, 2, 3) and d.C == 64 and d.N >= 128 and d.To == 1 and d.Ti == 1 and not d.epilogue & (L.EPI_ACCUM | L.EPI_SHUFFLE2)
                 and (d.ntap == 1 or (d.ntap == 3 and d.precision == 3)) else
                 f'gemm_tap_f16x3_kernel<{d.prologue}>' if d.precision == 3 else
                 f'gemm_tap_bf16x{3 if d.precision == 1 else 6}_kernel<{d.prologue}>' if d.precision in (1, 2) and d.C >= 32 else
                 f'gemm_tap_kernel<{16 if d.C < 32 else 32},{d.prologue}>') +
                (f' C{d.C} N{d.N} t{d.ntap} M{M} e{d.epilogue}' if _KEY_SHAPES else ''), _flops=2.0 * M * d.N * d.ntap * d.C,
           _bytes=4.0 * M * (d.C + d.N))
    return Y


class _LeafStream:
    """Weight gradients are leaves of the backward graph: nothing downstream reads them before the optimizer step.  Inside the
    generator backward they are issued on a second HIP stream, behind an event recorded when their operands are ready, and run
    concurrently with the input-gradient chain on the main stream (joined before the optimizer step).  Operands the side stream
    reads are kept referenced until the join, so that the main stream cannot recycle their memory early.  (record_stream would
    do the same without extending lifetimes, but blocks freed with a pending cross-stream event are not reusable until the
    lagging leaf stream has passed them: the caching allocator then reserved 150 GB for a 32 GB working set.)"""
    enabled = __import__('os').environ.get('SE_NO_WGRAD_STREAM') != '1'
    active = False
    streams = {}
    keep = []

    @classmethod
    def side(cls, device):
        k = torch.device(device).index
        if k not in cls.streams:
            cls.streams[k] = torch.cuda.Stream(device=device)
        return cls.streams[k]


class leaf_stream:
    """`with leaf_stream(*operands):` -- the launches inside go to the leaf stream when the generator backward has enabled it."""

    def __init__(self, *tensors):
        self.tensors = [t for t in tensors if t is not None]
        self.ctx = None

    def __enter__(self):
        if not (_LeafStream.active and self.tensors and self.tensors[0].is_cuda):
            return self
        dev = self.tensors[0].device
        main, side = torch.cuda.current_stream(dev), _LeafStream.side(dev)
        ev = torch.cuda.Event()
        ev.record(main)
        side.wait_event(ev)
        _LeafStream.keep.extend(self.tensors)
        self.ctx = torch.cuda.stream(side)
        self.ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self.ctx is not None:
            self.ctx.__exit__(*exc)
        return False


class branch_stream:
    """`with branch_stream(*inputs) as br: ...; br.join()` -- an independent branch of the graph (the complex decoder beside the
    mask decoder) on a second stream: it waits for an event recorded on the current stream on entry (its inputs are ready),
    `join()` makes the current stream wait for it.  Tensors it allocates belong to its stream's pool; they are only read by the
    main stream after `join()`, and the branch stream re-uses a freed block only after its next entry event, i.e. after every
    main-stream reader queued so far.  The inputs must stay referenced until `join()` (they do: locals of the caller).
    SE_NO_BRANCH_STREAM=1: the branch runs inline."""
    enabled = __import__('os').environ.get('SE_NO_BRANCH_STREAM') != '1'
    streams = {}

    def __init__(self, *tensors):
        self.dev = tensors[0].device if tensors and tensors[0] is not None and tensors[0].is_cuda else None
        self.ctx = self.side = self.main = None

    def __enter__(self):
        if not (branch_stream.enabled and self.dev is not None):
            return self
        k = self.dev.index
        if k not in branch_stream.streams:
            branch_stream.streams[k] = torch.cuda.Stream(device=self.dev)
        self.main, self.side = torch.cuda.current_stream(self.dev), branch_stream.streams[k]
        ev = torch.cuda.Event()
        ev.record(self.main)
        self.side.wait_event(ev)
        self.ctx = torch.cuda.stream(self.side)
        self.ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self.ctx is not None:
            self.ctx.__exit__(*exc)
            self.ctx = None
        return False

    def join(self):
        if self.side is not None:
            self.main.wait_stream(self.side)


def leaf_begin():
    _LeafStream.active = _LeafStream.enabled


def leaf_join(device):
    """the main stream waits for every weight gradient issued so far (before the optimizer step / gradient all-reduce)"""
    if _LeafStream.active:
        torch.cuda.current_stream(device).wait_stream(_LeafStream.side(device))
    _LeafStream.keep.clear()        # after the join: their memory returns to the main stream's pool with no cross-stream event
    _LeafStream.active = False


W3_ROUNDS = 1      # rounds of 512 workgroups of the triple-tap weight-gradient kernels (0: the generic rule below; measured 0 / 1 / 2)


def gemm_tap_wgrad(d, A, dY, dW, dbias=None, rowstats=None, ps=None, pb=None, chunks=None, explicit_precision=False,
                   scale=1.0):
    """dW (+ dbias) += scale * gradient: accumulates straight into the caller's (gradient) buffers."""
    L.check_cuda(A, dY, dW, dbias, rowstats, ps, pb)
    d.alpha = scale
    if _LeafStream.active and torch.cuda.current_stream(A.device) == _LeafStream.side(A.device):
        _LeafStream.keep.extend(t for t in (A, dY, rowstats, ps, pb) if t is not None)      # safety net: kept until the join
    if chunks is None:
        M = d.B * d.To * d.Fo
        nblk = d.ntap * ((d.C + 63) // 64) * ((d.N + 63) // 64)
        # 4 workgroups are resident per CU (LDS-bound): aim at a whole number of 1024-workgroup rounds -- a partial
        # last round costs a full one (2112 workgroups ran as 3 rounds, not 2.06) -- and at >= 4 rounds when the
        # rows allow it, so that the unequal start times of the first round even out
        rounds = 4 if M * nblk >= 4 * 1024 * 1024 else 2
        chunks = max(1, min((M + 255) // 256, (rounds * 1024) // nblk))
        if W3_ROUNDS and d.ntap % 3 == 0 and d.precision == 3 and d.prologue == L.PRO_NONE and not (d.epilogue & L.EPI_DROP):
            # the triple-tap kernels (wgrad3w_f16_kernel at C >= 128: 128 channels per block; wgrad3_bf16_kernel below) keep TWO workgroups
            # resident per CU: whole rounds of 512 workgroups -- 688 / 928 / 704 workgroups (the rule above at C = 128 / 192 / 256) ran as
            # two rounds with the second a third full: wgrad3 family 4.08 -> 3.71 ms per step with exactly one round
            blocks = (d.ntap // 3) * ((d.C + 127) // 128 if d.C >= 128 else (d.C + 63) // 64) * ((d.N + 63) // 64)
            chunks = max(8, min((M + 255) // 256, (W3_ROUNDS * 512 // blocks) // 8 * 8))
    Mt = d.B * d.To * d.Fo
    lin = d.ntap == 1 and d.B == 1 and d.To == 1
    saved = d.precision
    if lin and not explicit_precision:
        # the forward descriptor's choice is about the forward kernel only; precision 3 needs the dY scale (w_amax) too
        d.precision = WGRAD_LINEAR_PRECISION if (WGRAD_LINEAR_PRECISION != 3 or d.w_amax) else 2
    try:
        _wgrad_call(d, A, dY, dW, dbias, rowstats, ps, pb, chunks, Mt)
    finally:
        d.precision = saved
    return dW


_KEY_SHAPES = __import__('os').environ.get('SE_KEY_SHAPES') == '1'      # diagnostic: one timer family per GEMM shape


_WGRAD_ARITH = {0: 'f32', 1: 'bf16x3', 2: 'bf16x6', 3: 'f16x3'}
_WGRAD_CLASS = {0: 'wgrad', 1: 'wgrad3', 2: 'wgrad_lin'}


def _wgrad_key(prologue):
    """timer family of the weight-gradient launch that has just been issued: named by what se_gemm_tap_wgrad DISPATCHED to
    (se_gemm_tap_wgrad_last_kind) -- the dispatch depends on shape, scales and switches, and a family keyed by the request was
    priced against the wrong peak in round 4 (a 16-bit-pipe kernel against the fp32-MFMA peak: a fraction above 1)"""
    k = L.lib().se_gemm_tap_wgrad_last_kind()
    return f'{_WGRAD_CLASS.get(k >> 4, "wgrad")}_{_WGRAD_ARITH[k & 15]}<{prologue}>'


def _wgrad_call(d, A, dY, dW, dbias, rowstats, ps, pb, chunks, Mt):
    L.call('se_gemm_tap_wgrad', C.byref(d), L.ptr(A), L.ptr(dY), L.ptr(dW), L.ptr(dbias), L.ptr(rowstats),
           L.ptr(ps), L.ptr(pb), C.c_int(chunks), L.stream(),
           _key=lambda: _wgrad_key(d.prologue) + (f' C{d.C} N{d.N} t{d.ntap} M{Mt} p{d.precision} sf{d.sf}' if _KEY_SHAPES else ''),
           _flops=2.0 * Mt * d.N * d.ntap * d.C, _bytes=4.0 * Mt * (d.C + d.N))
    return dW


def gemm_ln_bwd(A, WT, x, stats, gamma, dR, dgamma, dbeta, out_amax=None):
    """dX = dR + LayerNorm-backward(A @ WT.T) (csrc/se_gemm.hip: se_gemm_ln_bwd): A [M, K], WT [64, K] fp32 or pre-split planes
    [3, 64, K] bf16 / [2, 64, K] scaled fp16 (then A._se_amax = the measured max |A|); x [M, 64], stats [M, 2], dR [M, 64] or
    None; dgamma / dbeta [64] accumulated; out_amax: optional zero-filled scalar raised to max |dX| (returned as dX._se_amax)."""
    L.check_cuda(A, WT, x, stats, gamma, dR, dgamma, dbeta)
    M, K = A.shape
    planes = WT.dtype in (torch.bfloat16, torch.float16)
    f16 = WT.dtype == torch.float16
    a_amax = getattr(A, '_se_amax', None)
    if f16 and a_amax is None:
        raise L.SeHipError('gemm_ln_bwd: scaled fp16 weight planes need the measured maximum of A (A._se_amax)')
    dX = torch.empty(M, 64, device=A.device, dtype=torch.float32)
    dX._se_amax = out_amax
    L.call('se_gemm_ln_bwd_f16', L.ptr(A), L.ptr(WT), C.c_int(WT.stride(0) if planes else 0), C.c_long(M), C.c_int(K), L.ptr(x),
           L.ptr(stats), L.ptr(gamma), L.ptr(dR), L.ptr(dX), L.ptr(dgamma), L.ptr(dbeta), C.c_int(3 if f16 else 2),
           L.ptr(a_amax if f16 else None), L.ptr(WT._se_amax if f16 else None), L.ptr(out_amax), L.stream(),
           _key='gemm_tap_f16x3_kernel<0>' if f16 else 'gemm_tap_bf16x6_kernel<0>', _flops=2.0 * M * 64 * K,
           _bytes=4.0 * M * (K + 192))
    return dX


# the LayerNorm-backward GEMM and the weight gradient of the same layer in one sweep over the rows (csrc/se_lnbwd_fused.hip);
# SE_LNBWD_FUSED=0: two launches again (se_gemm_ln_bwd_f16 + se_gemm_tap_wgrad on the weight-gradient stream)
LNBWD_FUSED = __import__('os').environ.get('SE_LNBWD_FUSED', '1') != '0'


def gemm_ln_bwd_wgrad(A, WT, x, stats, gamma, beta, dR, dgamma, dbeta, dW, dbias=None, out_amax=None, in_bound=None):
    """dX = dR + LayerNorm-backward(A @ WT.T) and dW [K, 64] += A^T LN(x), dbias [K] += sum A in ONE launch (se_gemm_ln_bwd_wgrad):
    A [M, K] with A._se_amax, WT [2, 64, K] scaled fp16 planes, K in (192, 256)."""
    L.check_cuda(A, WT, x, stats, gamma, beta, dR, dgamma, dbeta, dW, dbias)
    M, K = A.shape
    a_amax = getattr(A, '_se_amax', None)
    if WT.dtype != torch.float16 or a_amax is None or K not in (192, 256):
        raise L.SeHipError('gemm_ln_bwd_wgrad: needs scaled fp16 weight planes, the measured maximum of A and K in (192, 256)')
    dX = torch.empty(M, 64, device=A.device, dtype=torch.float32)
    dX._se_amax = out_amax
    L.call('se_gemm_ln_bwd_wgrad', L.ptr(A), L.ptr(WT), C.c_long(M), C.c_int(K), L.ptr(x), L.ptr(stats), L.ptr(gamma), L.ptr(beta),
           L.ptr(dR), L.ptr(dX), L.ptr(dgamma), L.ptr(dbeta), L.ptr(dW), L.ptr(dbias), L.ptr(a_amax), L.ptr(WT._se_amax),
           L.ptr(in_bound), C.c_int(LN_SEXP), L.ptr(out_amax), L.stream(), _key='lnbwd_wgrad_fused_f16x3', _flops=4.0 * M * 64 * K,
           _bytes=4.0 * M * (K + 192))
    return dX


def repack(src, No, Nt, Ni, so, stt, si, rev=0, out=None, accumulate=False):
    """dst[o][t][i] = src[o*so + i*si + t*stt]."""
    if out is None:
        out = torch.empty(No, Nt * Ni, device=src.device, dtype=torch.float32)
    L.call('se_repack', L.ptr(src), L.ptr(out), C.c_int(No), C.c_int(Nt), C.c_int(Ni), C.c_long(so),
           C.c_long(stt), C.c_long(si), C.c_int(rev), C.c_int(int(accumulate)), L.stream())
    return out


def unpack(src, dst, No, Nt, Ni, so, stt, si, rev=0, accumulate=False):
    """dst[o*so + i*si + t*stt] (+)= src[o][t][i]."""
    L.call('se_unpack', L.ptr(src), L.ptr(dst), C.c_int(No), C.c_int(Nt), C.c_int(Ni), C.c_long(so),
           C.c_long(stt), C.c_long(si), C.c_int(rev), C.c_int(int(accumulate)), L.stream())
    return dst


def pack_conv_fwd(w, rev_slabs=False):
    """PyTorch conv weight [N, Cin, kh, kw] -> [N][tap][Cin] (optionally un-reversing the
    newest-first slab order of DilatedDenseNet)."""
    N, Cin, kh, kw = w.shape
    return repack(w, N, kh * kw, Cin, Cin * kh * kw, 1, kh * kw, rev=1 if rev_slabs else 0)


def pack_conv_dgrad(w, rev_slabs=False):
    """[N, Cin, kh, kw] -> [Cin][tap][N] for the input-gradient GEMM (taps negated by the caller)."""
    N, Cin, kh, kw = w.shape
    return repack(w, Cin, kh * kw, N, kh * kw, 1, Cin * kh * kw, rev=2 if rev_slabs else 0)


def unpack_conv_wgrad(dwp, dw, rev_slabs=False, accumulate=False):
    """packed gradient [N][tap][Cin] -> PyTorch layout [N, Cin, kh, kw]."""
    N, Cin, kh, kw = dw.shape
    return unpack(dwp, dw, N, kh * kw, Cin, Cin * kh * kw, 1, kh * kw, rev=1 if rev_slabs else 0,
                  accumulate=accumulate)


def ff_fwd(x, rowstats, gamma, beta, W1, b1, W2, b2, drop_p=0.0, seed_h=0, seed_o=0, alpha=0.5, out_stats=False, store_h=True,
           in_bound=None, mid_bound=None):
    """fused Scale(alpha, PreNorm(FeedForward)) forward on scaled fp16 weight planes (csrc/se_ff.hip): returns (Y, H); store_h=False
    (the default path of the train step): the W-stationary kernel, H = W1 LN(x) + b1 is not stored (H = None) -- ff_bwd_fused
    recomputes it; store_h=True: H [M, hid] is written for ff_bwd_dgrad (the cross-check path)."""
    L.check_cuda(x, rowstats, gamma, beta, W1, b1, W2, b2)
    if W1.dtype != torch.float16 or W2.dtype != torch.float16:
        raise L.SeHipError('ff_fwd: needs the scaled fp16 weight planes of a WeightPlan (fp32 weights: the unfused GEMM path)')
    M, hid = x.shape[0], W1.shape[-2]
    H = torch.empty(M, hid, device=x.device, dtype=torch.float32) if store_h else None
    Y = torch.empty(M, 64, device=x.device, dtype=torch.float32)
    ost = torch.empty(M, 2, device=x.device, dtype=torch.float32) if out_stats else None     # (mean, rstd) of the rows of Y
    # in_bound / mid_bound: device scalars >= max |LN(x)| / max |Swish(H) mask / keep| (proven from the current parameters:
    # weights.WeightPlan.run_bounds); without them the static exponents LN_SEXP / HID_SEXP
    if mid_bound is not None and drop_p > 0.5:
        raise L.SeHipError('ff_fwd: the hidden-activation bound assumes a dropout keep probability >= 1/2')
    sc = L.F16Scales(in_bound.data_ptr() if in_bound is not None else None, LN_SEXP, HID_SEXP, W1._se_amax.data_ptr(),
                     W2._se_amax.data_ptr(), None, mid_bound.data_ptr() if mid_bound is not None else None)
    L.call('se_ff_fwd_f16', L.ptr(x), L.ptr(rowstats), L.ptr(gamma), L.ptr(beta), L.ptr(W1), L.ptr(b1), L.ptr(W2), L.ptr(b2),
           L.ptr(H), L.ptr(Y), L.ptr(ost), C.c_long(M), C.c_int(hid), C.c_float(drop_p), C.c_uint(seed_h & 0xFFFFFFFF),
           C.c_uint(seed_o & 0xFFFFFFFF), C.c_float(alpha), C.c_int(3 | 16), C.byref(sc), L.stream(), _key='ff_fwd_f16x3',
           _flops=4.0 * M * 64 * hid, _bytes=4.0 * M * (128 + (hid if H is not None else 0)))
    if out_stats:
        return Y, H, ost
    return Y, H


# The fused backward (csrc/se_ff_fused.hip; default with scaled fp16 planes): ONE persistent launch for dX, dgamma / dbeta AND the four
# weight gradients of the module -- H, S and dZ exist on chip only.  SE_FF_FUSED=0: the stored-H kernels (the forward writes H,
# ff_bwd_dgrad + two whole-gradient launches) -- the cross-check path of the fused kernels.
FF_FUSED = __import__('os').environ.get('SE_FF_FUSED', '1') != '0'


def ff_bwd_fused(dy, x, st, gamma, beta, W1, b1, W2T_scaled, dW1, db1, dW2, db2, dgamma, dbeta, drop_p=0.0, seed_h=0, seed_o=0,
                 alpha=0.5, dR2=None, out_amax=None, in_bound=None, mid_bound=None):
    """dx = dy + dR2 + LNbwd(dZ W1) and dW1 / db1 / dW2 / db2 / dgamma / dbeta accumulated, from x and dy alone (se_ff_bwd_fused)."""
    L.check_cuda(dy, x, st, gamma, beta, W1, b1, W2T_scaled, dW1, db1, dW2, db2, dgamma, dbeta, dR2)
    dy_amax = getattr(dy, '_se_amax', None)
    if dy_amax is None or W1.dtype != torch.float16 or W2T_scaled.dtype != torch.float16:
        raise L.SeHipError('ff_bwd_fused: needs scaled fp16 weight planes and the measured maximum of dy (dy._se_amax)')
    M, hid = x.shape[0], W1.shape[-2]
    dx = torch.empty(M, 64, device=dy.device, dtype=torch.float32)
    dx._se_amax = out_amax
    L.call('se_ff_bwd_fused', L.ptr(dy), L.ptr(x), L.ptr(st), L.ptr(gamma), L.ptr(beta), L.ptr(W1), L.ptr(b1), L.ptr(W2T_scaled),
           L.ptr(dR2), L.ptr(dx), L.ptr(dgamma), L.ptr(dbeta), L.ptr(dW1), L.ptr(db1), L.ptr(dW2), L.ptr(db2), C.c_long(M), C.c_int(hid),
           C.c_float(drop_p), C.c_uint(seed_h & 0xFFFFFFFF), C.c_uint(seed_o & 0xFFFFFFFF), C.c_float(alpha), L.ptr(dy_amax),
           L.ptr(W1._se_amax), L.ptr(W2T_scaled._se_amax), L.ptr(in_bound), C.c_int(LN_SEXP), L.ptr(mid_bound), C.c_int(HID_SEXP),
           L.ptr(out_amax), L.stream(), _key='ff_bwd_fused_f16x3',
           _flops=8.0 * M * 64 * hid,      # algorithmic: dZ chain 2 GEMMs + 2 weight gradients (the recomputed H = W1 LN(x) is not counted)
           _bytes=4.0 * M * 64 * (4 if dR2 is not None else 3))
    return dx


def ff_bwd_dgrad(dy, H, W2T_scaled, W1T, drop_p=0.0, seed_h=0, seed_o=0, ln=None, amax_out=(None, None)):
    """dgrad chain of the feed-forward module from a stored H (ff_bwd_kernel; scaled fp16 planes; the cross-check path of
    ff_bwd_fused): returns (dZ [M, hid], dLN [M, 64]); with ln = (x, rowstats, gamma, dR2 or None, dgamma, dbeta) the LayerNorm
    backward is applied in the same kernel and the second result is dX = dy + dR2 + LNbwd(dLN).  amax_out = (zero-filled scalars
    for max |dX|, max |dZ|): they travel with the results as ._se_amax."""
    L.check_cuda(dy, H, W2T_scaled, W1T)
    M, hid = H.shape
    dZ = torch.empty(M, hid, device=dy.device, dtype=torch.float32)
    out = torch.empty(M, 64, device=dy.device, dtype=torch.float32)
    dy_amax = getattr(dy, '_se_amax', None)
    if W2T_scaled.dtype != torch.float16 or W1T.dtype != torch.float16 or dy_amax is None:
        raise L.SeHipError('ff_bwd_dgrad: needs scaled fp16 weight planes and the measured maximum of dy (dy._se_amax)')
    x, st, g, dR2, dg, db = ln if ln is not None else (None,) * 6
    out._se_amax, dZ._se_amax = amax_out
    sc = L.F16Scales(dy_amax.data_ptr(), 0, 0, W2T_scaled._se_amax.data_ptr(), W1T._se_amax.data_ptr(),
                     out._se_amax.data_ptr(), dZ._se_amax.data_ptr())
    L.call('se_ff_bwd_dgrad_f16', L.ptr(dy), L.ptr(H), L.ptr(W2T_scaled), L.ptr(W1T), L.ptr(dZ), L.ptr(None if ln else out),
           C.c_long(M), C.c_int(hid), C.c_float(drop_p), C.c_uint(seed_h & 0xFFFFFFFF), C.c_uint(seed_o & 0xFFFFFFFF),
           C.c_int(3 | 16), L.ptr(x), L.ptr(st), L.ptr(g), L.ptr(dR2), L.ptr(out if ln else None), L.ptr(dg), L.ptr(db),
           C.byref(sc), L.stream(), _key='ff_bwd_dgrad_f16x3', _flops=4.0 * M * 64 * hid, _bytes=4.0 * M * (128 + 2 * hid + (128 if ln else 0)))
    return dZ, out
