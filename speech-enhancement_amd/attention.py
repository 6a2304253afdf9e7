"""Python side of the fused relative-position attention (csrc/se_attn.hip)."""
import ctypes as C

import torch

from . import _lib as L


def seq_geometry(B, T, Fq, axis):
    """token = (b*T + t)*Fq + f.  axis 'time': sequences (b, f) over t (generator.py:69);
    axis 'freq': sequences (b, t) over f (generator.py:71).
    returns (nseq, n, inner, outer_stride, inner_stride, pos_stride)."""
    if axis == 'time':
        return (B * Fq, T, Fq, T * Fq, 1, Fq)
    return (B * T, Fq, 1, Fq, 0, 1)


def f16_shape_ok(geom, maxpos=512):
    """the scaled split-fp16 attention BACKWARD (se_attn_bwd_f16_phase: the workgroup-cooperative kernel) takes this sequence geometry:
    at most 21 key tiles, the padded length + 128 within the offset table, 32-bit lane offsets"""
    nseq, n, inner, os_, is_, ps = geom
    npad = (n + 15) // 16 * 16
    return maxpos % 16 == 0 and npad + 128 <= maxpos and n <= 336 and ps * 192 * npad < 2 ** 31 - 1


def f16_fwd_shape_ok(geom, maxpos=512):
    """se_attn_fwd_f16 alone (no backward: inference) takes longer sequences: the V image (two fp16 planes) and the offset strips of
    8 waves within the 160 KB of LDS -- n <= 4079, e.g. the 1601 frames of a 10 s utterance; offsets beyond +-maxpos are clamped"""
    nseq, n, inner, os_, is_, ps = geom
    npad = (n + 15) // 16 * 16
    return maxpos % 16 == 0 and 2 * npad * 32 + 512 * 2 * 8 * 4 <= 160 * 1024 and ps * 192 * npad < 2 ** 31 - 1


def attn_fwd(qkv, E, geom, maxpos=512, scale=0.25, need_lse=True, Es=None, qkv_amax=None):
    """Es: optional pre-split planes of E (weights.WeightPlan): [3, 2*maxpos+1, 16] bf16, or [2, 2*maxpos+1, 16] scaled fp16 with
    its maximum in Es._se_amax -- together with qkv_amax (device scalar >= max |qkv|, e.g. raised by the qkv GEMM's epilogue) that
    selects the scaled split-fp16 kernel"""
    L.check_cuda(qkv, E, Es, qkv_amax)
    ntok = qkv.shape[0]
    O = torch.empty(ntok, 64, device=qkv.device, dtype=torch.float32)
    lse = torch.empty(ntok, 4, device=qkv.device, dtype=torch.float32) if need_lse else None
    nseq, n, inner, os_, is_, ps = geom
    if Es is not None and Es.dtype == torch.float16:
        if qkv_amax is None or not (f16_shape_ok(geom, maxpos) or (not need_lse and f16_fwd_shape_ok(geom, maxpos))):
            raise L.SeHipError('attn_fwd: fp16 planes of E need qkv_amax and a sequence the split-fp16 kernel takes')
        L.call('se_attn_fwd_f16', L.ptr(qkv), L.ptr(Es), C.c_long(Es.stride(0)), L.ptr(qkv_amax), L.ptr(Es._se_amax), L.ptr(O),
               L.ptr(lse), C.c_int(nseq), C.c_int(n), C.c_int(inner), C.c_long(os_), C.c_long(is_), C.c_long(ps), C.c_int(maxpos),
               C.c_float(scale), L.stream(), _key='attn_fwd3_f16x3', _flops=nseq * 4 * 3 * 2.0 * n * n * 16, _bytes=4.0 * ntok * 256)
        return O, lse
    L.call('se_attn_fwd_es', L.ptr(qkv), L.ptr(E), L.ptr(Es), C.c_long(Es.stride(0) if Es is not None else 0), L.ptr(O),
           L.ptr(lse), C.c_int(nseq), C.c_int(n), C.c_int(inner),
           C.c_long(os_), C.c_long(is_), C.c_long(ps), C.c_int(maxpos), C.c_float(scale), L.stream(),
           _key=('attn_fwd3_bf16x6' if 3 * ((n + 15) // 16 * 16) * 32 + 512 * (2 if n > 128 else 1) * (8 if n > 128 else 4) * 4 <= 160 * 1024 else 'attn_fwd_kernel'), _flops=nseq * 4 * 3 * 2.0 * n * n * 16, _bytes=4.0 * ntok * 256)
    return O, lse


def attn_bwd(qkv, E, O, dO, lse, geom, dE, maxpos=512, scale=0.25, leaf=None, qkv_amax=None, do_amax=None, dqkv_amax=None, delta=None):
    """returns dQKV [ntok,192]; accumulates into dE [2*maxpos+1, 16].  leaf: optional context-manager factory (gemm.leaf_stream):
    the reduction of the per-item dE tiles -- a leaf of the backward graph -- is then issued inside `leaf(ws, ...)`.
    qkv_amax / do_amax: device scalars >= max |qkv| / max |dO| (producer epilogues): both given and a shape the cooperative kernel
    takes -> the scaled split-fp16 kernel, which raises the zero-filled scalar dqkv_amax (optional) to max |dQKV| (returned as
    dqkv._se_amax); otherwise the fp32-MFMA kernels (any length).  delta: optional [ntok, 4] table rowsum(dO . O) per head (the
    to_out input-gradient GEMM writes it: EPI_DELTA); None: computed here."""
    L.check_cuda(qkv, E, O, dO, lse, dE, qkv_amax, do_amax, delta)
    ntok = qkv.shape[0]
    dqkv = torch.empty(ntok, 192, device=qkv.device, dtype=torch.float32)
    nseq, n, inner, os_, is_, ps = geom
    nbytes = L.lib().se_attn_bwd_workspace_bytes(C.c_long(ntok), C.c_int(maxpos), C.c_int(nseq), C.c_int(n))
    ws = torch.empty((nbytes + 3) // 4, device=qkv.device, dtype=torch.float32)
    f16 = qkv_amax is not None and do_amax is not None and f16_shape_ok(geom, maxpos)
    tk = dict(_key=('attn_bwd_f16x3 (+tables, dE reduce)' if f16 else 'attn_bwd_dkv + attn_bwd_dq (+delta)') + (' n>128' if n > 128 else ' n<=128'),
              _flops=nseq * 4 * 7 * 2.0 * n * n * 16, _bytes=4.0 * ntok * 512)
    if not f16:
        if delta is not None:
            raise L.SeHipError('attn_bwd: the fp32 kernels compute delta themselves')
        L.call('se_attn_bwd', L.ptr(qkv), L.ptr(E), L.ptr(O), L.ptr(dO), L.ptr(lse), L.ptr(dqkv), L.ptr(dE),
               C.c_int(nseq), C.c_int(n), C.c_int(inner), C.c_long(os_), C.c_long(is_), C.c_long(ps), C.c_long(ntok),
               C.c_int(maxpos), C.c_float(scale), L.ptr(ws), C.c_size_t(nbytes), L.stream(), **tk)
        return dqkv

    def run(phase, **kw):
        L.call('se_attn_bwd_f16_phase', L.ptr(qkv), L.ptr(E), L.ptr(O), L.ptr(dO), L.ptr(lse), L.ptr(delta), L.ptr(qkv_amax), L.ptr(do_amax),
               L.ptr(dqkv_amax), L.ptr(dqkv), L.ptr(dE), C.c_int(nseq), C.c_int(n), C.c_int(inner), C.c_long(os_), C.c_long(is_), C.c_long(ps),
               C.c_long(ntok), C.c_int(maxpos), C.c_float(scale), L.ptr(ws), C.c_size_t(nbytes), C.c_int(phase), L.stream(), **kw)
    if dqkv_amax is not None:
        dqkv._se_amax = dqkv_amax
    if leaf is None:
        run(3, **tk)
    else:
        run(1, **tk)
        with leaf(ws, dE):
            run(2)
    return dqkv
