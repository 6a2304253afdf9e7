"""Optimizers of core/optimizer.py behind ``build_optimizer(args, model, lr=None)``.

All four (nesterov-SGD, AdamW, LARS, Lamb) are flat-buffer HIP kernels: parameters and gradients of a model are
re-pointed into one contiguous fp32 buffer per weight-decay group (decay / no-decay, the reference's
``set_weight_decay`` rule), so a step is a handful of kernel launches, global-norm clipping works on the same buffers
(``clip_grad_norm``) and the data-parallel gradient exchange is one RCCL all-reduce per buffer.  LARS / Lamb get their
per-tensor trust ratios from a segmented norm kernel over the tensor offsets inside the flat buffer.
``state_dict()`` / ``load_state_dict()`` speak the torch.optim layout ({'state': {index: ...}, 'param_groups': [...]})
so optimizer states move both ways between this package and the reference (main_gan.py:117, 300-309).
``TorchLARS`` / ``TorchLamb`` are per-tensor torch restatements kept as the checker of the fused kernels (tests only).
"""
import ctypes as C

import torch

from . import _lib as L
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


_STATE_KEYS = {'sgd': ('momentum_buffer',), 'adamw': ('exp_avg', 'exp_avg_sq'), 'lars': ('mu',),
               'lamb': ('exp_avg', 'exp_avg_sq')}


class FlatOptimizer:
    """torch.optim-like surface (param_groups with 'lr', zero_grad, step, state_dict) over flat buffers."""

    def __init__(self, groups, kind, lr, weight_decay=0.0, momentum=0.9, betas=(0.9, 0.999), eps=None,
                 trust_coefficient=0.001, max_grad_norm=1.0, bias_correction=True, grad_averaging=True,
                 trust_clip=False, always_adapt=False):
        if kind not in _STATE_KEYS:
            raise ValueError(f'unknown optimizer kind {kind}')
        self.kind = kind
        self.param_groups = []
        self.step_count = 0
        self.momentum, self.betas = momentum, betas
        self.eps = eps if eps is not None else (1e-6 if kind == 'lamb' else 1e-8)
        self.trust_coefficient, self.max_grad_norm = trust_coefficient, max_grad_norm
        self.bias_correction, self.grad_averaging = bias_correction, grad_averaging
        self.trust_clip, self.always_adapt = trust_clip, always_adapt
        subs = []
        for g in groups:
            params = [p for p in g['params']]
            if not params:
                continue
            if kind == 'lars' and len({p.ndim > 1 for p in params}) > 1:
                # LARS adapts by tensor rank (core/optimizer.py:85), not by group: keep every flat buffer homogeneous
                hi = dict(g, params=[p for p in params if p.ndim > 1])
                lo = dict(g, params=[p for p in params if p.ndim <= 1])
                sub = [hi, lo]
            else:
                sub = [dict(g, params=params)]
            subs.extend(sub)
        # ONE gradient allocation for the whole model (the groups are 16-byte-aligned slices of it): the data-parallel exchange is a
        # single all-reduce per model and step (main_gan.py:133-171's DDP buckets), zero_grad a single fill
        dev0 = subs[0]['params'][0].device
        pads = [(sum(p.numel() for p in sg['params']) + 3) // 4 * 4 for sg in subs]
        self._grad_all = torch.zeros(sum(pads), device=dev0, dtype=torch.float32)
        o = 0
        for sg, npad in zip(subs, pads):
            self._add_group(sg, lr, weight_decay, self._grad_all[o:o + npad])
            o += npad
        dev = self.param_groups[0]['flat'].device
        self._sums = torch.zeros(len(self.param_groups), device=dev, dtype=torch.float64)

    def _add_group(self, g, lr, weight_decay, grad_buf):
        params = g['params']
        n = sum(p.numel() for p in params)
        dev = params[0].device
        npad = (n + 3) // 4 * 4
        flat = torch.zeros(npad, device=dev, dtype=torch.float32)[:n]
        grad = grad_buf[:n]
        o, offs = 0, [0]
        for p in params:
            k = p.numel()
            flat[o:o + k].copy_(p.data.reshape(-1))
            p.data = flat[o:o + k].view_as(p.data)
            p.grad = grad[o:o + k].view_as(p.data)
            o += k
            offs.append(o)
        wd = g.get('weight_decay', weight_decay)
        st = {'m': torch.zeros_like(flat)}
        if self.kind in ('adamw', 'lamb'):
            st['v'] = torch.zeros_like(flat)
        grp = {'params': params, 'lr': lr, 'weight_decay': wd, 'flat': flat, 'grad': grad, 'state': st}
        if self.kind in ('lars', 'lamb'):
            grp['seg_off'] = torch.tensor(offs, device=dev, dtype=torch.int64)
            grp['max_seg'] = max(p.numel() for p in params)
            grp['norms'] = torch.zeros(L.lib().se_segnorm_workspace_bytes(C.c_int(len(params))) // 8, device=dev,
                                       dtype=torch.float64)
            grp['adapt'] = bool(params[0].ndim > 1) if self.kind == 'lars' else bool(wd != 0 or self.always_adapt)
        self.param_groups.append(grp)

    def flat_grads(self):
        """the buffers a data-parallel exchange reduces: ONE per optimizer (every group's gradient is a slice of it)"""
        return [self._grad_all]

    def zero_grad(self, set_to_none=False):
        self._grad_all.zero_()
        for g in self.param_groups:
            o = 0
            for p in g['params']:          # keep .grad aliased to the flat buffer
                k = p.numel()
                if p.grad is None or p.grad.data_ptr() != g['grad'][o:o + k].data_ptr():
                    p.grad = g['grad'][o:o + k].view_as(p.data)
                o += k

    def _grad_sums(self):
        self._sums.zero_()
        for i, g in enumerate(self.param_groups):
            O.dot(g['grad'], g['grad'], self._sums[i:i + 1])
        return self._sums

    @torch.no_grad()
    def clip_grad_norm(self, max_norm):
        """torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm) on the flat buffers; returns the total norm
        (device scalar: no host sync)."""
        sums = self._grad_sums()
        for g in self.param_groups:
            L.call('se_grad_clip', L.ptr(g['grad']), C.c_long(g['grad'].numel()), L.ptr(sums), C.c_int(sums.numel()),
                   C.c_float(max_norm), L.stream())
        return sums.sum().sqrt()

    @torch.no_grad()
    def step(self):
        self.step_count += 1
        k = self.kind
        if k == 'lamb':
            sums = self._grad_sums()
            b1, b2 = self.betas
            bc1 = 1 - b1 ** self.step_count if self.bias_correction else 1.0
            bc2 = 1 - b2 ** self.step_count if self.bias_correction else 1.0
            beta3 = 1 - b1 if self.grad_averaging else 1.0
        for g in self.param_groups:
            st = g['state']
            if k == 'adamw':
                O.adamw(g['flat'], g['grad'], st['m'], st['v'], g['lr'], self.betas[0], self.betas[1], self.eps,
                        g['weight_decay'], self.step_count)
            elif k == 'sgd':
                O.sgd_nesterov(g['flat'], g['grad'], st['m'], g['lr'], self.momentum, self.step_count == 1)
            elif k == 'lars':
                L.call('se_lars_step', L.ptr(g['flat']), L.ptr(g['grad']), L.ptr(st['m']), L.ptr(g['seg_off']),
                       C.c_int(len(g['params'])), C.c_long(g['max_seg']), L.ptr(g['norms']), C.c_int(int(g['adapt'])),
                       C.c_float(g['lr']), C.c_float(g['weight_decay']), C.c_float(self.momentum),
                       C.c_float(self.trust_coefficient), L.stream())
            else:
                L.call('se_lamb_step', L.ptr(g['flat']), L.ptr(g['grad']), L.ptr(st['m']), L.ptr(st['v']),
                       L.ptr(g['seg_off']), C.c_int(len(g['params'])), C.c_long(g['max_seg']), L.ptr(g['norms']),
                       L.ptr(sums), C.c_int(sums.numel()), C.c_float(self.max_grad_norm), C.c_int(int(g['adapt'])),
                       C.c_int(int(self.trust_clip)), C.c_float(g['lr']), C.c_float(g['weight_decay']), C.c_float(b1),
                       C.c_float(b2), C.c_float(beta3), C.c_float(self.eps), C.c_float(bc1), C.c_float(bc2), L.stream())

    # ---- torch.optim-compatible (de)serialisation ----------------------------------------------------------------
    def _group_defaults(self, g):
        """the hyper-parameter keys torch.optim / the reference's LARS / Lamb keep in a param group"""
        if self.kind == 'sgd':
            ref = torch.optim.SGD([torch.zeros(1)], lr=g['lr'], momentum=self.momentum, nesterov=True).param_groups[0]
        elif self.kind == 'adamw':
            ref = torch.optim.AdamW([torch.zeros(1)], lr=g['lr'], betas=self.betas, eps=self.eps,
                                    weight_decay=g['weight_decay']).param_groups[0]
        elif self.kind == 'lars':
            ref = dict(lr=g['lr'], weight_decay=g['weight_decay'], momentum=self.momentum,
                       trust_coefficient=self.trust_coefficient)
        else:
            ref = dict(lr=g['lr'], bias_correction=self.bias_correction, betas=self.betas, eps=self.eps,
                       weight_decay=g['weight_decay'], grad_averaging=self.grad_averaging,
                       max_grad_norm=self.max_grad_norm, trust_clip=self.trust_clip, always_adapt=self.always_adapt)
            if self.step_count:
                ref['step'] = self.step_count
        out = {k: v for k, v in ref.items() if k != 'params'}
        out['weight_decay'] = g['weight_decay']
        return out

    def state_dict(self):
        keys = _STATE_KEYS[self.kind]
        state, groups, idx = {}, [], 0
        for g in self.param_groups:
            ids, o = [], 0
            for p in g['params']:
                k = p.numel()
                if self.step_count > 0:
                    bufs = (g['state']['m'], g['state'].get('v'))
                    ent = {name: buf[o:o + k].view_as(p).clone() for name, buf in zip(keys, bufs)}
                    if self.kind == 'adamw':
                        ent['step'] = torch.tensor(float(self.step_count))
                    state[idx] = ent
                ids.append(idx)
                idx += 1
                o += k
            groups.append(dict(self._group_defaults(g), params=ids))
        return {'state': state, 'param_groups': groups}

    def load_state_dict(self, sd):
        """accepts the torch.optim layout written by this class or by the reference's optimizers"""
        if 'param_groups' not in sd or 'state' not in sd:
            raise ValueError('optimizer state_dict is not in the torch.optim layout (keys: %s)' % sorted(sd))
        if len(sd['param_groups']) != len(self.param_groups):
            raise ValueError(f"optimizer state_dict has {len(sd['param_groups'])} param groups, expected "
                             f"{len(self.param_groups)}")
        keys = _STATE_KEYS[self.kind]
        step = 0
        for g, sg in zip(self.param_groups, sd['param_groups']):
            if len(sg['params']) != len(g['params']):
                raise ValueError('optimizer state_dict: parameter count mismatch in a param group')
            g['lr'] = sg.get('lr', g['lr'])
            g['weight_decay'] = sg.get('weight_decay', g['weight_decay'])
            step = max(step, int(sg.get('step', 0)))
            o = 0
            for p, idx in zip(g['params'], sg['params']):
                k = p.numel()
                ent = sd['state'].get(idx)
                if ent is not None:
                    for name, buf in zip(keys, (g['state']['m'], g['state'].get('v'))):
                        t = ent.get(name)
                        if t is None:          # e.g. SGD's momentum_buffer before the first step
                            continue
                        if t.numel() != k:
                            raise ValueError(f'optimizer state_dict: state {name} of parameter {idx} has {t.numel()} '
                                             f'elements, expected {k}')
                        buf[o:o + k].copy_(t.reshape(-1))
                        step = max(step, 1)
                    if 'step' in ent:
                        step = max(step, int(float(ent['step'])))
                o += k
        self.step_count = step


class TorchLARS(torch.optim.Optimizer):
    """per-tensor torch restatement of core/optimizer.py:71-113 (checker of the fused kernel)."""

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


class TorchLamb(torch.optim.Optimizer):
    """per-tensor torch restatement of core/optimizer.py:116-238 (checker of the fused kernel)."""

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


LARS, Lamb = TorchLARS, TorchLamb          # round-1 names (tests/test_host.py pins them to the reference goldens)


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
        return FlatOptimizer(groups, 'lars', lr, weight_decay=args.weight_decay, momentum=args.momentum)
    if name == 'lamb':       # the reference passes max_grad_norm=args.max_norm (core/optimizer.py:41-42)
        return FlatOptimizer(groups, 'lamb', lr, weight_decay=args.weight_decay, max_grad_norm=args.max_norm)
    return None
