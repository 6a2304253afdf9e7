"""Metric discriminator behind the reference's surface (models/discriminator.py:35-62).

The four spectral-norm conv4x4/s2 + InstanceNorm + PReLU stages (all of the discriminator's FLOPs and bytes) run
on the HIP tap-GEMM / norm kernels with a hand-written backward; the spectral-norm power iteration + weight scaling
(se_spectral_norm) and the tail -- global max-pool over the final 12x20 map, Linear(128,64), Dropout, PReLU, Linear(64,1),
learnable sigmoid (se_disc_tail_fwd / _bwd) -- are single-workgroup HIP kernels too: the discriminator step launches no
vendor BLAS.
The image is processed as [B, T, F, C] (channels-last, T x F transposed w.r.t. the reference's [B, C, F, T]); the
4x4 taps are transposed accordingly, so no data is ever permuted.
"""
import ctypes as C

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib as L
from . import layers as LY
from . import ops as O

# arithmetic of the 4x4 stride-2 convolutions with >= 32 input channels (forward, input gradients): 2 = exact three-way bf16 split, six
# products (fp32-equivalent, the generic split kernel: K = 16 taps x C >= 512); 0 = fp32 MFMA (module switches for A/B runs)
D_PRECISION = 2
CLASS_DGRAD = True      # A/B switch: the input gradients through the `up`-mode tap GEMM
THIN_CONV1 = True      # A/B switch: the first stage through the tap GEMM again

# weight index (kh over F, kw over T)  ->  tap offset on the [T, F] grid
D_TAPS = [(kw - 1, kh - 1) for kh in range(4) for kw in range(4)]


def batch_pesq(clean, noisy):
    """models/discriminator.py:25-32.  PESQ is third-party arithmetic (PyPI `pesq`, absent here): the labels
    must come from a provider; see train.set_pesq_provider."""
    from . import train
    return train.pesq_labels(clean, noisy)


class LearnableSigmoid(nn.Module):
    def __init__(self, in_features, beta=1):
        super().__init__()
        self.beta = beta
        self.slope = nn.Parameter(torch.ones(in_features))

    def forward(self, x):
        return self.beta * torch.sigmoid(self.slope * x)


class _SNHolder(nn.Module):
    """parameter container with the state_dict names of the old-style torch spectral_norm hook."""

    def __init__(self, shape, bias=False):
        super().__init__()
        if bias:
            self.bias = nn.Parameter(torch.zeros(shape[0]))
        w = torch.empty(shape)
        nn.init.kaiming_uniform_(w, a=5 ** 0.5)
        self.weight_orig = nn.Parameter(w)
        h, wd = shape[0], w.numel() // shape[0]
        self.register_buffer('weight_u', F.normalize(torch.randn(h), dim=0, eps=1e-12))
        self.register_buffer('weight_v', F.normalize(torch.randn(wd), dim=0, eps=1e-12))

    def weight(self, train):
        """W / sigma with one power iteration in training mode (torch.nn.utils.spectral_norm); u, v buffers updated in place"""
        return spectral_weights([self], train)[0]


def _ptr_array(ts):
    return (C.c_void_p * len(ts))(*[t.data_ptr() if t is not None else 0 for t in ts])


def _zeros_like_many(ts):
    """zero tensors shaped like every tensor of `ts` (None stays None), carved out of ONE zero-filled buffer: the backward functions
    below need a dozen small zero-initialised outputs each -- a dozen fill launches per call otherwise (16-B aligned views)"""
    live = [t for t in ts if t is not None]
    if not live:
        return [None for _ in ts]
    offs, o = [], 0
    for t in live:
        offs.append(o)
        o += (t.numel() + 3) // 4 * 4
    flat = torch.zeros(o, device=live[0].device, dtype=torch.float32)
    it = iter(zip(live, offs))
    out = []
    for t in ts:
        if t is None:
            out.append(None)
        else:
            t_, o_ = next(it)
            out.append(flat[o_:o_ + t_.numel()].view(t_.shape))
    return out


def spectral_weights(holders, train, detach=False):
    """the spectrally-normalised weights of several _SNHolder layers from ONE launch (se_spectral_norm)"""
    Ws = [h.weight_orig.detach() if detach else h.weight_orig for h in holders]
    uv = [t for h in holders for t in (h.weight_u, h.weight_v)]
    return list(_SpectralNormFn.apply(bool(train), len(holders), *Ws, *uv))


class _SpectralNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, train, n, *ts):
        Ws, uv = ts[:n], ts[n:]
        us, vs = uv[0::2], uv[1::2]
        hs = [W.shape[0] for W in Ws]
        ws_ = [W.numel() // W.shape[0] for W in Ws]
        Wc = [W.detach().contiguous() for W in Ws]
        Wn = [torch.empty_like(W) for W in Wc]
        sigma = torch.empty(n, device=Wc[0].device, dtype=torch.float32)
        ha, wa = (C.c_int * n)(*hs), (C.c_int * n)(*ws_)
        L.call('se_spectral_norm', C.c_int(n), _ptr_array(Wc), _ptr_array(us), _ptr_array(vs), _ptr_array(Wn), ha, wa,
               L.ptr(sigma), C.c_int(int(train)), C.c_float(1e-12), L.stream())
        # the hook treats the (updated) u, v as constants of the graph
        uvc = torch.cat([t.reshape(-1) for t in (*us, *vs)])       # ONE copy of the 2 n small vectors (twelve clones before)
        ctx.save_for_backward(sigma, uvc, *Wn)
        ctx.n, ctx.hw = n, (hs, ws_)
        ctx.uv_sizes = [t.numel() for t in (*us, *vs)]
        return tuple(Wn)

    @staticmethod
    def backward(ctx, *dWn):
        n = ctx.n
        sigma, uvc, *Wn = ctx.saved_tensors
        uvs = list(torch.split(uvc, ctx.uv_sizes))
        us, vs = uvs[:n], uvs[n:]
        hs, ws_ = ctx.hw
        need = ctx.needs_input_grad[2:2 + n]
        dW = _zeros_like_many([Wn[i] if need[i] and dWn[i] is not None else None for i in range(n)])
        if any(d is not None for d in dW):
            dc = [dWn[i].contiguous() if dW[i] is not None else None for i in range(n)]
            L.call('se_spectral_norm_bwd', C.c_int(n), _ptr_array(dc), _ptr_array(Wn), _ptr_array(us), _ptr_array(vs),
                   L.ptr(sigma), _ptr_array(dW), (C.c_int * n)(*hs), (C.c_int * n)(*ws_), L.stream())
        return (None, None) + tuple(dW) + (None,) * (2 * n)


class _DiscTailFn(torch.autograd.Function):
    """a4 [B, To, Fo, 128] -> [B, 1]: max-pool, Linear(128,64), dropout mask, PReLU, Linear(64,1), beta * sigmoid(slope z)"""

    @staticmethod
    def forward(ctx, a4, W1, b1, mask, slope1, W2, b2, sslope, beta):
        B = a4.shape[0]
        P = a4.shape[1] * a4.shape[2]
        a4 = a4.contiguous()
        out = torch.empty(B, device=a4.device, dtype=torch.float32)
        ws = torch.empty(L.lib().se_disc_tail_workspace_bytes(C.c_int(B)) // 4, device=a4.device, dtype=torch.float32)
        args = [t.detach().contiguous() if t is not None else None for t in (W1, b1, mask, slope1, W2, b2, sslope)]
        L.call('se_disc_tail_fwd', L.ptr(a4), C.c_int(B), C.c_int(P), L.ptr(args[0]), L.ptr(args[1]), L.ptr(args[2]),
               L.ptr(args[3]), L.ptr(args[4]), L.ptr(args[5]), L.ptr(args[6]), C.c_float(beta), L.ptr(out), L.ptr(ws), L.stream())
        ctx.save_for_backward(ws, *[a for a in args if a is not None])
        ctx.has_mask, ctx.beta, ctx.shape = args[2] is not None, beta, a4.shape
        return out.view(B, 1)

    @staticmethod
    def backward(ctx, dout):
        ws, *rest = ctx.saved_tensors
        if ctx.has_mask:
            W1, b1, mask, slope1, W2, b2, sslope = rest
        else:
            (W1, b1, slope1, W2, b2, sslope), mask = rest, None
        B, To, Fo, _ = ctx.shape
        need = ctx.needs_input_grad
        dA = torch.zeros(ctx.shape, device=dout.device, dtype=torch.float32) if need[0] else None
        dW1, db1, ds1, dW2, db2, dss = _zeros_like_many([t if need[i] else None for i, t in
                                                          ((1, W1), (2, b1), (4, slope1), (5, W2), (6, b2), (7, sslope))])
        L.call('se_disc_tail_bwd', L.ptr(dout.contiguous().view(-1)), L.ptr(ws), C.c_int(B), C.c_int(To * Fo), L.ptr(W1),
               L.ptr(mask), L.ptr(slope1), L.ptr(W2), L.ptr(sslope), C.c_float(ctx.beta), L.ptr(dA), L.ptr(dW1), L.ptr(db1),
               L.ptr(ds1), L.ptr(dW2), L.ptr(db2), L.ptr(dss), L.stream())
        return dA, dW1, db1, None, ds1, dW2, db2, dss, None


def _out(n):
    return (n + 2 - 4) // 2 + 1


class _DConvStackFn(torch.autograd.Function):
    """xy planes [B,T,F,4] -> [B,T/16,F/16,128]; 4 x (conv4x4 s2 p1 (no bias), InstanceNorm(affine), PReLU)."""

    @staticmethod
    def forward(ctx, xy, *wts):
        Ws, gs, bs, sl = wts[0:4], wts[4:8], wts[8:12], wts[12:16]
        B, T, Fq, _ = xy.shape
        x, Ti, Fi, Cin = xy.contiguous(), T, Fq, 4
        saved = []
        for i in range(4):
            N = Ws[i].shape[0]
            To, Fo = _out(Ti), _out(Fi)
            R = torch.empty(B, To, Fo, N, device=x.device, dtype=torch.float32)
            stats = LY.O.zeros(B, N, 2, device=x.device, dtype=torch.float64)
            if i == 0 and N == 16 and THIN_CONV1:
                # 2 input channels, 16 taps, stride 2: a direct kernel (csrc/se_thin.hip) -- as a tap GEMM its tiles are 94 % padding
                L.call('se_dconv1_fwd', L.ptr(x), L.ptr(Ws[0].detach().contiguous()), L.ptr(R), L.ptr(stats), C.c_int(B), C.c_int(Ti),
                       C.c_int(Fi), C.c_int(N), L.stream(), _key='dconv1 (thin)', _bytes=4.0 * B * (Ti * Fi * 4 + To * Fo * N))
            else:
                wp = LY.pack_w(Ws[i].contiguous(), C_pad=Cin)
                d = LY.GM.make_desc(B, To, Fo, Ti, Fi, D_TAPS, Cin, Cin, N, N, st=2, sf=2, epilogue=L.EPI_STATS,
                                    precision=D_PRECISION if Cin >= 32 else 0)
                LY.GM.gemm_tap(d, x, wp, R, stats=stats)
            a = torch.empty_like(R)
            mr = LY.inorm_prelu_fwd(R, stats, gs[i], bs[i], sl[i], a, N, 0)
            saved.append((x, R, mr, Ti, Fi, To, Fo, Cin, N))
            x, Ti, Fi, Cin = a, To, Fo, N
        ctx.saved, ctx.wts, ctx.B = saved, wts, B
        return x

    @staticmethod
    def backward(ctx, dout):
        Ws, gs, bs, sl = ctx.wts[0:4], ctx.wts[4:8], ctx.wts[8:12], ctx.wts[12:16]
        B = ctx.B
        zz = _zeros_like_many([*Ws, *gs, *bs, *sl])
        dW, dg, db, ds = zz[0:4], zz[4:8], zz[8:12], zz[12:16]
        dy = dout.contiguous()
        dxy = None
        for i in (3, 2, 1, 0):
            x, R, mr, Ti, Fi, To, Fo, Cin, N = ctx.saved[i]
            dR = LY.inorm_prelu_bwd(R, mr, gs[i], bs[i], sl[i], dy, N, 0, dg[i], db[i], ds[i])
            need_dx = i > 0 or ctx.needs_input_grad[0]
            thin = i == 0 and N == 16 and THIN_CONV1
            if ctx.needs_input_grad[1 + i] and thin:
                L.call('se_dconv1_wgrad', L.ptr(x), L.ptr(dR), L.ptr(dW[0]), C.c_int(B), C.c_int(Ti), C.c_int(Fi), C.c_int(N), L.stream(),
                       _key='dconv1 (thin)', _bytes=4.0 * B * (Ti * Fi * 4 + To * Fo * N))
            elif ctx.needs_input_grad[1 + i]:
                fd = LY.GM.make_desc(B, To, Fo, Ti, Fi, D_TAPS, Cin, Cin, N, N, st=2, sf=2)
                dwp = LY.O.zeros(N, 16 * Cin, device=dR.device)
                LY.GM.gemm_tap_wgrad(fd, x, dR, dwp, None)
                LY._unpack_w(dwp, dW[i], Cin, False)
            if need_dx and thin:
                dx = torch.empty(B, Ti, Fi, Cin, device=dR.device, dtype=torch.float32)
                L.call('se_dconv1_dgrad', L.ptr(dR), L.ptr(Ws[0].detach().contiguous()), L.ptr(dx), C.c_int(B), C.c_int(Ti), C.c_int(Fi),
                       C.c_int(N), L.stream(), _key='dconv1 (thin)', _bytes=4.0 * B * (Ti * Fi * 4 + To * Fo * N))
                dy = dx
                dxy = dx
            elif need_dx:
                wd = LY.GM.pack_conv_dgrad(Ws[i].contiguous())            # [Cin_true][16][N]
                if wd.shape[0] != Cin:
                    wd = torch.cat([wd, wd.new_zeros(Cin - wd.shape[0], wd.shape[1])], 0)
                dx = torch.empty(B, Ti, Fi, Cin, device=dR.device, dtype=torch.float32)
                if CLASS_DGRAD and N % 32 == 0 and Cin <= 64 and To == _out(Ti) and Fo == _out(Fi):
                    # 4 of the 16 taps reach a pixel: one workgroup per 128 pixels of a parity class (csrc/se_thin.hip)
                    L.call('se_dconv_dgrad', L.ptr(dR), L.ptr(wd), L.ptr(dx), C.c_int(B), C.c_int(Ti), C.c_int(Fi), C.c_int(N), C.c_int(Cin),
                           L.stream(), _key='dconv dgrad (parity classes)', _flops=2.0 * B * Ti * Fi * Cin * 4 * N)
                else:
                    dd = LY.GM.make_desc(B, Ti, Fi, To, Fo, [(-a, -c) for a, c in D_TAPS], N, N, Cin, Cin, st=2, sf=2, up=1,
                                         precision=D_PRECISION if N >= 32 else 0)
                    LY.GM.gemm_tap(dd, dR, wd, dx)
                dy = dx
                if i == 0:
                    dxy = dx
        ctx.saved = None
        return (dxy,) + tuple(dW) + tuple(dg) + tuple(db) + tuple(ds)


class Discriminator(nn.Module):
    """Discriminator(ndf, in_channel=2).forward(x, y): x, y [B,1,F,T] -> [B,1]."""

    def __init__(self, ndf, in_channel=2):
        super().__init__()
        if in_channel != 2:
            raise ValueError('the HIP conv stack is built for in_channel == 2')
        chans = [in_channel, ndf, ndf * 2, ndf * 4, ndf * 8]
        mods = []
        for i in range(4):
            mods += [_SNHolder((chans[i + 1], chans[i], 4, 4)), nn.InstanceNorm2d(chans[i + 1], affine=True),
                     nn.PReLU(chans[i + 1])]
        mods += [nn.AdaptiveMaxPool2d(1), nn.Flatten(), _SNHolder((ndf * 4, ndf * 8), bias=True), nn.Dropout(0.3),
                 nn.PReLU(ndf * 4), _SNHolder((1, ndf * 4), bias=True), LearnableSigmoid(1)]
        self.layers = nn.Sequential(*mods)

    def forward_planes(self, pa, pb, detach_params=False):
        """pa, pb: planes [B,T,F,4] whose channel 0 is the magnitude (clean first, like the reference's callers).
        detach_params: the generator step only needs the gradient w.r.t. pb (the reference computes and then
        discards the discriminator's weight gradients there, core/function.py:261-279)."""
        z = torch.zeros_like(pa[..., 0])
        xy = torch.stack([pa[..., 0], pb[..., 0], z, z], -1)
        ly = self.layers
        train = self.training
        dt = (lambda t: t.detach()) if detach_params else (lambda t: t)
        sn = spectral_weights([ly[i] for i in (0, 3, 6, 9, 14, 17)], train, detach=detach_params)   # all six layers, one launch
        Ws = sn[:4]
        args = Ws + [dt(ly[i].weight) for i in (1, 4, 7, 10)] + [dt(ly[i].bias) for i in (1, 4, 7, 10)] + \
            [dt(ly[i].weight) for i in (2, 5, 8, 11)]
        a4 = _DConvStackFn.apply(xy, *args)
        B = a4.shape[0]
        p = ly[15].p
        mask = F.dropout(torch.ones(B, 64, device=a4.device), p, True) if (train and p > 0) else None   # nn.Dropout(0.3)
        return _DiscTailFn.apply(a4, sn[4], dt(ly[14].bias), mask, dt(ly[16].weight),
                                 sn[5], dt(ly[17].bias), dt(ly[18].slope), float(ly[18].beta))

    def forward(self, x, y):
        def planes(m):          # [B,1,F,T] -> [B,T,F,4]
            mm = m[:, 0].transpose(1, 2)
            z = torch.zeros_like(mm)
            return torch.stack([mm, z, z, z], -1)
        return self.forward_planes(planes(x), planes(y))
