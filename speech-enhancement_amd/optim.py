"""Optimizers of core/optimizer.py behind ``build_optimizer(args, model, lr=None)``.

AdamW and nesterov-SGD (the two the GAN recipe uses) are flat-buffer HIP kernels: parameters and gradients of a
model are re-pointed into one contiguous fp32 buffer per weight-decay group (decay / no-decay, the reference's
``set_weight_decay`` rule), so a step is two kernel launches and the data-parallel gradient exchange is one
RCCL all-reduce of one buffer.  LARS / Lamb keep the reference's per-tensor trust-ratio semantics on torch ops.
"""
import torch

from . import ops as O


def set_weight_decay(model, skip_list=(), skip_keywords=()):
    """core/optimizer.py:47-60: 1-D tensors and biases get weight_decay 0."""
    has_decay, no_decay = [], []
    for name, p in model.named_parameters():
        if not p.requires_grad:
            continue
        if p.dim() == 1 or name.endswith('.bias') or name in skip_list or any(k in name for k in skip_keywords):
            no_decay.append(p)
        else:
            has_decay.append(p)
    return [{'params': has_decay}, {'params': no_decay, 'weight_decay': 0.}]


class FlatOptimizer:
    """torch.optim-like surface (param_groups with 'lr', zero_grad, step, state_dict) over flat buffers."""

    def __init__(self, groups, kind, lr, weight_decay=0.0, momentum=0.9, betas=(0.9, 0.999), eps=1e-8):
        self.kind = kind
        self.param_groups = []
        self.step_count = 0
        self.momentum, self.betas, self.eps = momentum, betas, eps
        for g in groups:
            params = [p for p in g['params']]
            if not params:
                continue
            n = sum(p.numel() for p in params)
            dev = params[0].device
            flat = torch.empty(n, device=dev, dtype=torch.float32)
            grad = torch.zeros(n, device=dev, dtype=torch.float32)
            o = 0
            for p in params:
                k = p.numel()
                flat[o:o + k].copy_(p.data.reshape(-1))
                p.data = flat[o:o + k].view_as(p.data)
                p.grad = grad[o:o + k].view_as(p.data)
                o += k
            wd = g.get('weight_decay', weight_decay)
            st = {'m': torch.zeros_like(flat)}
            if kind == 'adamw':
                st['v'] = torch.zeros_like(flat)
            self.param_groups.append({'params': params, 'lr': lr, 'weight_decay': wd, 'flat': flat, 'grad': grad,
                                      'state': st})

    def flat_grads(self):
        return [g['grad'] for g in self.param_groups]

    def zero_grad(self, set_to_none=False):
        for g in self.param_groups:
            g['grad'].zero_()
            o = 0
            for p in g['params']:          # keep .grad aliased to the flat buffer
                k = p.numel()
                if p.grad is None or p.grad.data_ptr() != g['grad'][o:o + k].data_ptr():
                    p.grad = g['grad'][o:o + k].view_as(p.data)
                o += k

    @torch.no_grad()
    def step(self):
        self.step_count += 1
        for g in self.param_groups:
            if self.kind == 'adamw':
                O.adamw(g['flat'], g['grad'], g['state']['m'], g['state']['v'], g['lr'], self.betas[0], self.betas[1],
                        self.eps, g['weight_decay'], self.step_count)
            else:
                O.sgd_nesterov(g['flat'], g['grad'], g['state']['m'], g['lr'], self.momentum, self.step_count == 1)

    def state_dict(self):
        return {'kind': self.kind, 'step': self.step_count,
                'groups': [{'lr': g['lr'], 'weight_decay': g['weight_decay'],
                            'state': {k: v.clone() for k, v in g['state'].items()}} for g in self.param_groups]}

    def load_state_dict(self, sd):
        self.step_count = sd['step']
        for g, s in zip(self.param_groups, sd['groups']):
            g['lr'] = s['lr']
            for k, v in s['state'].items():
                g['state'][k].copy_(v)


class LARS(torch.optim.Optimizer):
    """core/optimizer.py:71-113 semantics: trust-ratio scaling and weight decay only for tensors with ndim > 1."""

    def __init__(self, params, lr=0, weight_decay=0, momentum=0.9, trust_coefficient=0.001):
        super().__init__(params, dict(lr=lr, weight_decay=weight_decay, momentum=momentum,
                                      trust_coefficient=trust_coefficient))

    @torch.no_grad()
    def step(self):
        for g in self.param_groups:
            for p in g['params']:
                if p.grad is None:
                    continue
                dp = p.grad
                if p.ndim > 1:
                    dp = dp.add(p, alpha=g['weight_decay'])
                    pn, un = torch.norm(p), torch.norm(dp)
                    one = torch.ones_like(pn)
                    q = torch.where(pn > 0., torch.where(un > 0, g['trust_coefficient'] * pn / un, one), one)
                    dp = dp.mul(q)
                st = self.state[p]
                if 'mu' not in st:
                    st['mu'] = torch.zeros_like(p)
                mu = st['mu']
                mu.mul_(g['momentum']).add_(dp)
                p.add_(mu, alpha=-g['lr'])


class Lamb(torch.optim.Optimizer):
    """core/optimizer.py:116-238 semantics (global grad-norm clip, Adam moments, per-tensor trust ratio)."""

    def __init__(self, params, lr=1e-3, bias_correction=True, betas=(0.9, 0.999), eps=1e-6, weight_decay=0.01,
                 grad_averaging=True, max_grad_norm=1.0, trust_clip=False, always_adapt=False):
        super().__init__(params, dict(lr=lr, bias_correction=bias_correction, betas=betas, eps=eps,
                                      weight_decay=weight_decay, grad_averaging=grad_averaging,
                                      max_grad_norm=max_grad_norm, trust_clip=trust_clip, always_adapt=always_adapt))

    @torch.no_grad()
    def step(self):
        dev = self.param_groups[0]['params'][0].device
        one = torch.tensor(1.0, device=dev)
        gn = torch.zeros(1, device=dev)
        for g in self.param_groups:
            for p in g['params']:
                if p.grad is not None:
                    gn.add_(p.grad.pow(2).sum())
        gn = gn.sqrt()
        mgn = torch.tensor(self.defaults['max_grad_norm'], device=dev)
        clip = torch.where(gn > mgn, gn / mgn, one)
        for g in self.param_groups:
            b1, b2 = g['betas']
            g['step'] = g.get('step', 0) + 1
            beta3 = 1 - b1 if g['grad_averaging'] else 1.0
            bc1 = 1 - b1 ** g['step'] if g['bias_correction'] else 1.0
            bc2 = 1 - b2 ** g['step'] if g['bias_correction'] else 1.0
            for p in g['params']:
                if p.grad is None:
                    continue
                grad = p.grad.div(clip)
                st = self.state[p]
                if not st:
                    st['exp_avg'] = torch.zeros_like(p)
                    st['exp_avg_sq'] = torch.zeros_like(p)
                st['exp_avg'].mul_(b1).add_(grad, alpha=beta3)
                st['exp_avg_sq'].mul_(b2).addcmul_(grad, grad, value=1 - b2)
                denom = (st['exp_avg_sq'].sqrt() / (bc2 ** 0.5)).add_(g['eps'])
                upd = (st['exp_avg'] / bc1).div_(denom)
                wd = g['weight_decay']
                if wd != 0:
                    upd.add_(p, alpha=wd)
                if wd != 0 or g['always_adapt']:
                    wn, un = p.norm(2.0), upd.norm(2.0)
                    tr = torch.where(wn > 0, torch.where(un > 0, wn / un, one), one)
                    if g['trust_clip']:
                        tr = torch.minimum(tr, one)
                    upd.mul_(tr)
                p.add_(upd, alpha=-g['lr'])


def build_optimizer(args, model, lr=None):
    """core/optimizer.py:15-44."""
    skip = model.no_weight_decay() if hasattr(model, 'no_weight_decay') else {}
    skip_kw = model.no_weight_decay_keywords() if hasattr(model, 'no_weight_decay_keywords') else {}
    groups = set_weight_decay(model, skip, skip_kw)
    name = args.optimizer.lower()
    if not lr:
        lr = args.lr
    if name == 'sgd':        # the reference passes no weight decay to SGD (core/optimizer.py:33-35)
        for g in groups:
            g['weight_decay'] = 0.0
        return FlatOptimizer(groups, 'sgd', lr, momentum=args.momentum)
    if name == 'adamw':
        return FlatOptimizer(groups, 'adamw', lr, weight_decay=args.weight_decay)
    if name == 'lars':
        return LARS(groups, lr, weight_decay=args.weight_decay, momentum=args.momentum)
    if name == 'lamb':
        return Lamb(groups, lr=lr, weight_decay=args.weight_decay, max_grad_norm=args.max_norm)
    return None
