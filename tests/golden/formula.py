"""Closed-form ("formula") parameter values used by every parity fixture.

No RNG: each tensor is a deterministic function of its name, shape and a per-tensor
``base`` constant recorded in ``state_spec.json`` (so the same weights can be rebuilt in
the build container, on the GPU box, in the oracle and in the product).
"""
import json
import os
import zlib

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


def _wave(name: str, numel: int) -> np.ndarray:
    crc = zlib.crc32(name.encode())
    idx = np.arange(numel, dtype=np.float64)
    return np.sin(idx * (0.37 + 0.001 * (crc % 97)) + (crc % 628) / 100.0)


def formula_tensor(name: str, shape, dtype: str, base: float) -> torch.Tensor:
    numel = int(np.prod(shape)) if len(shape) else 1
    if dtype == 'int64':
        return torch.zeros(shape, dtype=torch.int64)
    w = _wave(name, numel)
    if name.endswith(('weight_u', 'weight_v')):
        w = w / np.linalg.norm(w)
    elif name.endswith('running_var'):
        w = 1.0 + 0.2 * w
    elif name.endswith('running_mean'):
        w = 0.05 * w
    elif 'rel_pos_emb' in name:
        w = 0.3 * w
    elif len(shape) >= 2:
        fan_in = int(np.prod(shape[1:]))
        w = np.sqrt(3.0 / fan_in) * w
    elif name.endswith('.bias'):
        w = 0.02 * w
    else:                       # 1-D scale-like tensors: norm weights, PReLU slopes, sigmoid slope
        w = base + 0.1 * w * (abs(base) if base != 0 else 1.0)
    return torch.from_numpy(w.reshape(shape)).to(torch.float32)


def load_spec(which: str):
    with open(os.path.join(HERE, 'state_spec.json')) as f:
        return json.load(f)[which]


def formula_state(which: str):
    """which in {'generator', 'discriminator'} -> OrderedDict name -> tensor."""
    from collections import OrderedDict
    sd = OrderedDict()
    for name, shape, dtype, base in load_spec(which):
        sd[name] = formula_tensor(name, tuple(shape), dtype, base)
    return sd
