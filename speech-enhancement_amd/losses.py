"""Loss terms of train_gan (core/function.py:251-272) as autograd Functions over the HIP reduction kernels."""
import torch

from . import ops as O


class _SpecLossFn(torch.autograd.Function):
    """planes A, B [.., 4] -> tensor [2] = (MSE(|A|,|B|), MSE(Re) + MSE(Im)); gradient flows to A only."""

    @staticmethod
    def forward(ctx, A, Bp):
        n = A.numel() // 4
        sums = torch.zeros(2, device=A.device, dtype=torch.float64)
        O.spec_loss(A.contiguous(), Bp.contiguous(), sums)
        ctx.save_for_backward(A, Bp)
        return (sums / n).float()

    @staticmethod
    def backward(ctx, g):
        A, Bp = ctx.saved_tensors
        n = A.numel() // 4
        dA = torch.empty_like(A)
        O.spec_loss_bwd(A.contiguous(), Bp.contiguous(), dA, g.contiguous().float(), 2.0 / n, 2.0 / n)
        return dA, None


def spec_losses(est_planes, clean_planes):
    v = _SpecLossFn.apply(est_planes, clean_planes)
    return v[0], v[1]          # loss_mag, loss_ri


class _L1Fn(torch.autograd.Function):
    """mean |a - b| over [rows, L]; b may be a strided row view (row stride ldb); gradient to a only."""

    @staticmethod
    def forward(ctx, a, b):
        rows, L_ = a.shape
        sums = torch.zeros(1, device=a.device, dtype=torch.float64)
        O.l1_loss(a, a.stride(0), b, b.stride(0), sums, rows, L_)
        ctx.save_for_backward(a, b)
        return (sums[0] / (rows * L_)).float()

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        rows, L_ = a.shape
        return O.l1_loss_bwd(a, a.stride(0), b, b.stride(0), g.reshape(1).contiguous().float(), 1.0 / (rows * L_),
                             rows, L_), None


def l1_time_loss(est_audio, clean_audio):
    if est_audio.stride(1) != 1 or clean_audio.stride(1) != 1:
        raise ValueError('l1_time_loss needs unit stride along time')
    return _L1Fn.apply(est_audio, clean_audio)
