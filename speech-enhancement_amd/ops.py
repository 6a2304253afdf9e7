"""Thin Python launchers for the non-GEMM kernels of libse_hip.so (csrc/se_norms.hip, se_dwconv.hip,
se_elem.hip).  Every function takes/returns CUDA tensors and raises SeHipError on failure."""
import ctypes as C
import os as _os

import torch

from . import _lib as L

f32 = torch.float32
f64 = torch.float64
_i, _l, _f, _d = C.c_int, C.c_long, C.c_float, C.c_double


def _new(*shape, like, dtype=f32):
    return torch.empty(*shape, device=like.device, dtype=dtype)


class ZeroArena:
    """Step-scoped zero-initialised scratch.  A train step needs ~250 small zero-filled buffers (statistics accumulators,
    packed weight-gradient tiles, atomically accumulated sums): one fill launch each.  Inside `with ARENA.step(device)` they
    are carved out of one buffer that a single fill clears at the start of the step; outside a step (tests, inference) every
    request is a plain torch.zeros.  A buffer handed out here must not outlive the step."""

    def __init__(self):
        self.buf, self.off, self.high, self.active, self.missed = None, 0, 0, False, 0

    def begin(self, device, min_bytes=16 << 20):
        need = max(min_bytes, 2 * (self.high + self.missed))
        if self.buf is None or self.buf.device != device or self.buf.numel() < need:
            self.buf = torch.zeros(need, device=device, dtype=torch.uint8)
        else:
            self.buf[:max(self.high, 256)].zero_()          # ONE fill for everything handed out last step
        self.off, self.missed, self.active = 0, 0, True

    def end(self):
        self.high = max(self.high, self.off)
        self.active = False

    def take(self, shape, dtype, device):
        n = 1
        for v in shape:
            n *= int(v)
        nbytes = n * torch.empty((), dtype=dtype).element_size()
        if not self.active or self.buf.device != device or nbytes == 0:
            return torch.zeros(*shape, device=device, dtype=dtype)
        start = (self.off + 255) & ~255
        if start + nbytes > self.buf.numel():
            self.missed += nbytes + 256                      # grown at the next begin()
            return torch.zeros(*shape, device=device, dtype=dtype)
        self.off = start + nbytes
        return self.buf[start:start + nbytes].view(dtype).view(*shape)


ARENA = ZeroArena()


def zeros(*shape, device, dtype=f32):
    """zero-filled scratch that does not outlive the current train step (see ZeroArena)."""
    return ARENA.take(shape, dtype, torch.device(device) if not isinstance(device, torch.device) else device)


def _zeros(*shape, like, dtype=f32):
    return ARENA.take(shape, dtype, like.device)


# ---------------------------------------------------------------- LayerNorm(64)
def row_stats(x, M, C_=64, ld=None, eps=1e-5):
    st = _new(M, 2, like=x)
    L.call('se_row_stats', L.ptr(x), L.ptr(st), _l(M), _i(C_), _i(ld or C_), _f(eps), L.stream())
    return st


def layernorm_fwd(x, g, b, R=None, want_stats=True, eps=1e-5, out_stats=False):
    """out_stats: also return (mean, rstd) of the rows of the RESULT (the next LayerNorm's statistics)"""
    M = x.numel() // 64
    y = torch.empty_like(x)
    st = _new(M, 2, like=x) if want_stats else None
    ost = _new(M, 2, like=x) if out_stats else None
    L.call('se_layernorm_fwd_stats', L.ptr(x), L.ptr(g), L.ptr(b), L.ptr(R), L.ptr(y), L.ptr(st), L.ptr(ost), _l(M), _i(64),
           _f(eps), L.stream())
    return (y, st, ost) if out_stats else (y, st)


def layernorm_bwd(x, st, g, dy, dg, db, dR=None, dR2=None, amax=None):
    """amax: optional zero-filled device scalar raised to max |dx| (travels with the result as dx._se_amax)"""
    M = x.numel() // 64
    dx = torch.empty_like(x)
    L.call('se_layernorm_bwd_amax', L.ptr(x), L.ptr(st), L.ptr(g), L.ptr(dy), L.ptr(dR), L.ptr(dR2), L.ptr(dx), L.ptr(dg),
           L.ptr(db), _l(M), _i(64), L.ptr(amax), L.stream())
    dx._se_amax = amax
    return dx


# ---------------------------------------------------------------- instance / batch norm pieces
def col_stats(x, ld, x_off, B, P, C_, stats=None):
    if stats is None:
        stats = _zeros(B, C_, 2, like=x, dtype=f64)
    L.call('se_col_stats', L.ptr(x), _i(ld), _i(x_off), L.ptr(stats), _i(B), _l(P), _i(C_), L.stream())
    return stats


def norm_finalize(stats, g, beta, nb, C_, count, eps=1e-5, running_mean=None, running_var=None, momentum=0.1):
    mr = _new(nb, C_, 2, like=g)
    ss = _new(nb, C_, 2, like=g)
    L.call('se_norm_finalize', L.ptr(stats), L.ptr(g), L.ptr(beta), L.ptr(mr), L.ptr(ss), _i(nb), _i(C_), _d(count),
           _f(eps), L.ptr(running_mean), L.ptr(running_var), _f(momentum), L.stream())
    return mr, ss


def inorm_prelu_fwd(x, ldx, x_off, stats, g, beta, slope, y, ldy, y_off, B, P, C_, eps=1e-5, amax=None):
    """InstanceNorm(affine) + PReLU from the producer's (sum, sumsq): one launch; returns mr [B, C, 2] = (mean, rstd).
    amax: optional device scalar raised to max |y| (the operand scale of the scaled split-fp16 convolutions that read y)"""
    mr = _new(B, C_, 2, like=g)
    L.call('se_inorm_prelu_fwd_amax', L.ptr(x), _i(ldx), _i(x_off), L.ptr(stats), L.ptr(g), L.ptr(beta), L.ptr(slope), L.ptr(y),
           _i(ldy), _i(y_off), L.ptr(mr), _i(B), _l(P), _i(C_), _d(float(P)), _f(eps), L.ptr(amax), L.stream())
    return mr


def copy_cols_amax(src, lds, dst, ldd, rows, C_, amax=None):
    """dst[r, :C] = src[r, :C] (row strides lds / ldd) and amax (zero-filled device scalar) raised to max |src|"""
    L.check_cuda(src, dst, amax)
    L.call('se_copy_cols_amax', L.ptr(src), _i(lds), L.ptr(dst), _i(ldd), _l(rows), _i(C_), L.ptr(amax), L.stream())
    return dst


def bn_eval_scale(rm, rv, g, beta, eps=1e-5):
    C_ = g.numel()
    ss = _new(1, C_, 2, like=g)
    mr = _new(1, C_, 2, like=g)
    L.call('se_bn_eval_scale', L.ptr(rm), L.ptr(rv), L.ptr(g), L.ptr(beta), L.ptr(ss), L.ptr(mr), _i(C_), _f(eps),
           L.stream())
    return mr, ss


def affine_prelu(x, ldx, x_off, ss, slope, y, ldy, y_off, B, P, C_):
    L.call('se_affine_prelu', L.ptr(x), _i(ldx), _i(x_off), L.ptr(ss), L.ptr(slope), L.ptr(y), _i(ldy), _i(y_off),
           _i(B), _l(P), _i(C_), L.stream())
    return y


def norm_prelu_bwd(x, ldx, x_off, mr, g, beta, slope, dy, ldy, y_off, dx, lddx, dx_off, dg, dbeta, dslope, B, P, C_,
                   per_batch=True, act=0, allreduce=None, count=None, amax=None):
    """allreduce: optional callable applied to the fp64 reduction buffer between the two phases
    (SyncBatchNorm backward); count: elements per statistic (defaults to the local count); amax: optional zero-filled device
    scalar raised to max |dx| (operand scale of the scaled split-fp16 GEMMs that read dx)."""
    if count is None:
        count = float(P if per_batch else P * B)
    red = zeros(L.lib().se_norm_prelu_bwd_workspace_bytes(_i(B), _i(C_), _i(int(per_batch))) // 8, device=x.device, dtype=f64)
    args = lambda phase: (L.ptr(x), _i(ldx), _i(x_off), L.ptr(mr), L.ptr(g), L.ptr(beta), L.ptr(slope),
                          L.ptr(dy), _i(ldy), _i(y_off), L.ptr(red), L.ptr(dx), _i(lddx), _i(dx_off), L.ptr(dg),
                          L.ptr(dbeta), L.ptr(dslope), _i(B), _l(P), _i(C_), _i(int(per_batch)), _i(act),
                          _i(phase), _d(count), L.ptr(amax), L.stream())
    if allreduce is None:
        L.call('se_norm_prelu_bwd_amax', *args(1 | 2 | 8 | 16))      # two launches: reduce, apply (+ parameter gradients); red from the arena
    else:
        L.call('se_norm_prelu_bwd_amax', *args(1 | 4 | 16))
        allreduce(red)
        L.call('se_norm_prelu_bwd_amax', *args(2))
    return dx


# ---------------------------------------------------------------- depthwise conv
def dwconv31(x, w, bias, geom, stats=None, flip=False):
    y = torch.empty_like(x)
    nseq, n, inner, os_, is_, ps = geom
    L.call('se_dwconv31', L.ptr(x), L.ptr(w), L.ptr(bias), L.ptr(y), L.ptr(stats), _i(int(flip)), _i(nseq), _i(n),
           _i(inner), _l(os_), _l(is_), _l(ps), L.stream(), _key='dwconv31 (fwd / dgrad)', _bytes=8.0 * x.numel())
    return y


def dwconv31_glu_bwd(dh, w, u, gate, geom, amax=None):
    """dZ [M, 256] = GLU-backward((u, gate), depthwise-conv input gradient of dh): one kernel, dU never goes to memory;
    u = a sigmoid(gate) [M, 128] is the forward GLU result, gate [M, 128] the gate half of the pre-GLU activations"""
    M = dh.shape[0]
    dz = torch.empty(M, 256, device=dh.device, dtype=torch.float32)
    nseq, n, inner, os_, is_, ps = geom
    L.call('se_dwconv31_glu_bwd', L.ptr(dh), L.ptr(w), L.ptr(u), L.ptr(gate), L.ptr(dz), L.ptr(amax), _i(nseq), _i(n), _i(inner), _l(os_),
           _l(is_), _l(ps), L.stream(), _key='dwconv31 dgrad + glu_bwd', _bytes=4.0 * (3 * dh.numel() + dz.numel()))
    dz._se_amax = amax
    return dz


# SE_DW_BWD_FUSED=0: the two-launch depthwise backward (dwconv31_glu_bwd + dwconv31_wgrad)
DW_BWD_FUSED = _os.environ.get('SE_DW_BWD_FUSED', '1') != '0'


def dwconv31_bwd_fused(dh, w, u, gate, dw, dbias, geom, amax=None):
    """the whole depthwise-conv backward in one sweep (csrc/se_dwconv.hip, dwconv_bwd_fused_kernel): returns dZ [M, 256] like
    dwconv31_glu_bwd AND accumulates dw [128, 31] / dbias [128] like dwconv31_wgrad(u, dh, ...)"""
    M = dh.shape[0]
    dz = torch.empty(M, 256, device=dh.device, dtype=torch.float32)
    nseq, n, inner, os_, is_, ps = geom
    ws = _new(L.lib().se_dwconv31_wgrad_workspace_bytes() // 4, like=dh)
    L.call('se_dwconv31_bwd_fused', L.ptr(dh), L.ptr(w), L.ptr(u), L.ptr(gate), L.ptr(dz), L.ptr(amax), L.ptr(dw), L.ptr(dbias),
           L.ptr(ws), _l(M), _i(nseq), _i(n), _i(inner), _l(os_), _l(is_), _l(ps), L.stream(),
           _key='dwconv31 dgrad + glu_bwd + wgrad', _bytes=4.0 * (3 * dh.numel() + dz.numel()))
    dz._se_amax = amax
    return dz


def dwconv31_wgrad(x, dy, dw, dbias, geom):
    nseq, n, inner, os_, is_, ps = geom
    ws = _new(L.lib().se_dwconv31_wgrad_workspace_bytes() // 4, like=x)
    L.call('se_dwconv31_wgrad', L.ptr(x), L.ptr(dy), L.ptr(dw), L.ptr(dbias), _i(nseq), _i(n), _i(inner), _l(os_),
           _l(is_), _l(ps), L.ptr(ws), L.stream(), _key='dwconv31_wgrad (+ reduce)', _bytes=8.0 * x.numel())


# ---------------------------------------------------------------- front-end glue
COMP = {None: 0, 'none': 0, 'norm': 0, 'pow': 1, 'log': 2}


def clip_scale(x):
    B, L_ = x.shape
    c = _new(B, like=x)
    L.call('se_clip_scale', L.ptr(x), L.ptr(c), _i(B), _i(L_), L.stream())
    return c


def reflect_pad_scale(x, c, pad):
    B, L_ = x.shape
    xp = _new(B, L_ + 2 * pad, like=x)
    L.call('se_reflect_pad_scale', L.ptr(x), L.ptr(c), L.ptr(xp), _i(B), _i(L_), _i(pad), L.stream())
    return xp


def compress_planes(R, ldr, rows, F_, comp, pre_scale=1.0):
    P = _new(rows, F_, 4, like=R)
    L.call('se_compress_planes', L.ptr(R), _i(ldr), L.ptr(P), _l(rows), _i(F_), _i(COMP[comp]), _f(pre_scale),
           L.stream())
    return P


def uncompress_rows(P, rows, F_, lda, comp, post_scale=1.0):
    A = _zeros(rows, lda, like=P)
    L.call('se_uncompress_rows', L.ptr(P), L.ptr(A), _i(lda), _l(rows), _i(F_), _i(COMP[comp]), _f(post_scale),
           L.stream())
    return A


def uncompress_rows_bwd(P, dA, lda, dP, rows, F_, comp, post_scale=1.0):
    L.call('se_uncompress_rows_bwd', L.ptr(P), L.ptr(dA), _i(lda), L.ptr(dP), _l(rows), _i(F_), _i(COMP[comp]),
           _f(post_scale), L.stream())
    return dP


def ola(frames, env, B, T, n_fft, hop, trim=None, L_out=None):
    trim = n_fft // 2 if trim is None else trim
    L_out = hop * (T - 1) if L_out is None else L_out
    y = _new(B, L_out, like=frames)
    L.call('se_ola', L.ptr(frames), L.ptr(env), L.ptr(y), _i(B), _i(T), _i(n_fft), _i(hop), _i(trim), _i(L_out),
           L.stream())
    return y


def reflect_pad_bwd(dxp, c, B, L_, pad):
    dx = _new(B, L_, like=dxp)
    L.call('se_reflect_pad_bwd', L.ptr(dxp), L.ptr(c), L.ptr(dx), _i(B), _i(L_), _i(pad), L.stream())
    return dx


def compress_planes_bwd(R, ldr, dP, rows, F_, comp, pre_scale=1.0):
    dR = _zeros(rows, ldr, like=R)
    L.call('se_compress_planes_bwd', L.ptr(R), _i(ldr), L.ptr(dP), L.ptr(dR), _l(rows), _i(F_), _i(COMP[comp]),
           _f(pre_scale), L.stream())
    return dR


def ola_bwd(dy, env, B, T, n_fft, hop):
    dfr = _new(B * T, n_fft, like=dy)
    L.call('se_ola_bwd', L.ptr(dy), L.ptr(env), L.ptr(dfr), _i(B), _i(T), _i(n_fft), _i(hop), L.stream())
    return dfr


# ---------------------------------------------------------------- generator output / misc
def assemble(mask, ldm, nin, cplx):
    n = nin.numel() // 4
    est = torch.empty_like(nin)
    L.call('se_assemble', L.ptr(mask), _i(ldm), L.ptr(nin), L.ptr(cplx), L.ptr(est), _l(n), L.stream())
    return est


def assemble_bwd(est, dest, nin, dmask, ldm, dcplx):
    n = nin.numel() // 4
    L.call('se_assemble_bwd', L.ptr(est), L.ptr(dest), L.ptr(nin), L.ptr(dmask), _i(ldm), L.ptr(dcplx), _l(n),
           L.stream())


def mask_tail(U, ldu, wb, slope, n, F_):
    M = _new(n, like=U)
    L.call('se_mask_tail', L.ptr(U), _i(ldu), L.ptr(wb), L.ptr(slope), L.ptr(M), _l(n), _i(F_), L.stream())
    return M


def mask_tail_bwd(U, ldu, wb, slope, dM, dU, dwb, dslope, n, F_):
    L.call('se_mask_tail_bwd', L.ptr(U), _i(ldu), L.ptr(wb), L.ptr(slope), L.ptr(dM), L.ptr(dU), L.ptr(dwb),
           L.ptr(dslope), _l(n), _i(F_), L.stream())


def glu_bwd(Z, dU, M, H, amax=None):
    """amax: optional zero-filled device scalar raised to max |dZ| (travels with the result as dZ._se_amax)"""
    dZ = torch.empty_like(Z)
    L.call('se_glu_bwd_amax', L.ptr(Z), L.ptr(dU), L.ptr(dZ), _l(M), _i(H), L.ptr(amax), L.stream())
    dZ._se_amax = amax
    return dZ


def glu_bwd_gate(U, G, dU, M, H, amax=None):
    """glu_bwd from the GLU result U = a sigmoid(g) [M, H] and the gate half G [M, H] (the value half is never stored)"""
    dZ = torch.empty(M, 2 * H, device=U.device, dtype=torch.float32)
    L.call('se_glu_bwd_gate', L.ptr(U), L.ptr(G), L.ptr(dU), L.ptr(dZ), _l(M), _i(H), L.ptr(amax), L.stream())
    dZ._se_amax = amax
    return dZ


def gate_tanh(Y, M, C_):
    G = torch.empty(M, C_, device=Y.device, dtype=torch.float32)
    L.call('se_gate_tanh', L.ptr(Y), L.ptr(G), _l(M), _i(C_), L.stream())
    return G


def gate_tanh_bwd(Y, dG, M, C_):
    dY = torch.empty(M, 2 * C_, device=Y.device, dtype=torch.float32)
    L.call('se_gate_tanh_bwd', L.ptr(Y), L.ptr(dG), L.ptr(dY), _l(M), _i(C_), L.stream())
    return dY


def spec_loss(A, B_, sums):
    L.call('se_spec_loss', L.ptr(A), L.ptr(B_), L.ptr(sums), _l(A.numel() // 4), L.stream())
    return sums


def spec_loss_bwd(A, B_, dA, up, cmag, cri, accumulate=False):
    L.call('se_spec_loss_bwd', L.ptr(A), L.ptr(B_), L.ptr(dA), L.ptr(up), _f(cmag), _f(cri), _l(A.numel() // 4),
           _i(int(accumulate)), L.stream())
    return dA


def l1_loss(a, lda, b, ldb, sums, rows, L_):
    L.call('se_l1_loss', L.ptr(a), _l(lda), L.ptr(b), _l(ldb), L.ptr(sums), _l(rows), _i(L_), L.stream())
    return sums


def l1_loss_bwd(a, lda, b, ldb, up, ck, rows, L_):
    da = _new(rows, L_, like=a)
    L.call('se_l1_loss_bwd', L.ptr(a), _l(lda), L.ptr(b), _l(ldb), L.ptr(da), L.ptr(up), _f(ck), _l(rows), _i(L_),
           L.stream())
    return da


def adamw(p, g, m, v, lr, b1, b2, eps, wd, step):
    L.call('se_adamw', L.ptr(p), L.ptr(g), L.ptr(m), L.ptr(v), _l(p.numel()), _f(lr), _f(b1), _f(b2), _f(eps), _f(wd),
           _i(step), L.stream())


def sgd_nesterov(p, g, buf, lr, momentum, first):
    L.call('se_sgd_nesterov', L.ptr(p), L.ptr(g), L.ptr(buf), _l(p.numel()), _f(lr), _f(momentum), _i(int(first)),
           L.stream())


def dot(a, b, out):
    L.call('se_dot', L.ptr(a), L.ptr(b), L.ptr(out), _l(a.numel()), L.stream())
    return out


def axpbypcz(a, b, c, alpha, beta, gamma, out=None):
    if out is None:
        out = torch.empty_like(a)
    L.call('se_axpbypcz', L.ptr(a), L.ptr(b), L.ptr(c), L.ptr(out), _f(alpha), _f(beta), _f(gamma), _l(a.numel()),
           L.stream())
    return out
