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


def cond_signals(B: int, L: int, seed: int):
    """A WELL-CONDITIONED (clean, noisy) pair: each clean clip is 40 sinusoids (50 Hz .. 7.9 kHz at 16 kHz) plus broadband
    noise, so no STFT bin is near zero and the |z|^0.3 backward of the consistency path is well conditioned."""
    rs = np.random.RandomState(seed)
    n = np.arange(L, dtype=np.float64)
    clean = np.zeros((B, L))
    for b in range(B):
        f, ph, a = rs.uniform(50.0, 7900.0, 40), rs.uniform(0.0, 2 * np.pi, 40), rs.uniform(0.01, 0.04, 40)
        clean[b] = (a[:, None] * np.sin(2 * np.pi * f[:, None] * n[None, :] / 16000.0 + ph[:, None])).sum(0)
    clean = clean + 0.03 * rs.randn(B, L)
    noisy = clean + 0.05 * rs.randn(B, L)
    return torch.from_numpy(clean.astype(np.float32)), torch.from_numpy(noisy.astype(np.float32))


def long_clip(L: int, seed: int) -> np.ndarray:
    """one noisy utterance-like clip of L samples: slowly modulated harmonics + noise (BASELINE config 4 golden)."""
    rs = np.random.RandomState(seed)
    n = np.arange(L, dtype=np.float64) / 16000.0
    f0 = 140.0 + 40.0 * np.sin(2 * np.pi * 0.3 * n)
    ph = 2 * np.pi * np.cumsum(f0) / 16000.0
    env = 0.5 * (1.0 + np.sin(2 * np.pi * 1.7 * n))
    x = sum((0.08 / h) * np.sin(h * ph + 0.3 * h) for h in range(1, 13)) * env
    return (x + 0.03 * rs.randn(L)).astype(np.float32)


def tsc_state():
    """formula weights of the TSC-diffusion hybrid (models/tsc_diffusion.py TSCNet; names / shapes: tsc_state_spec.json)"""
    from collections import OrderedDict
    with open(os.path.join(HERE, 'tsc_state_spec.json')) as f:
        spec = json.load(f)
    sd = OrderedDict()
    for name, shape, dtype, base in spec:
        sd[name] = formula_tensor('tscd.' + name, tuple(shape), dtype, base)
    return sd
