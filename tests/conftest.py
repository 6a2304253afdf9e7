import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests', 'golden')):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden():
    import numpy as np
    return np.load(os.path.join(ROOT, 'tests', 'golden', 'golden_v1.npz'))


@pytest.fixture(scope='session')
def golden2():
    """round-2 vectors (tests/golden/make_golden_v2.py): validate_gan, --gen-first, --max-norm, front-end API,
    one FULL-SIZE (T=321) reference train_gan step"""
    import numpy as np
    return np.load(os.path.join(ROOT, 'tests', 'golden', 'golden_v2.npz'))


@pytest.fixture(scope='session')
def golden3():
    """round-3 vectors (tests/golden/make_golden_v3.py): gradient of the reference's compressed_stft on a well-conditioned
    signal, the reference's train_gan for cp / sc / scp on that signal, the reference's predict on one 10 s clip"""
    import numpy as np
    return np.load(os.path.join(ROOT, 'tests', 'golden', 'golden_v3.npz'))


@pytest.fixture(scope='session')
def golden4():
    """round-4 vectors (tests/golden/make_golden_v4.py): the reference's FULL-SIZE train_gan steps -- the fp32 twin of golden_v2's
    fp64 cmgan step (calibrates the full-size gradient bars) and one `cp` step (fp64 + fp32) on the conditioned clip pair"""
    import numpy as np
    return np.load(os.path.join(ROOT, 'tests', 'golden', 'golden_v4.npz'))


def full_size_signals(seed, B=2, L=32000):
    """the inputs of the full-size golden step, regenerated from the seed (numpy legacy RandomState is bit-stable)"""
    import numpy as np
    import torch
    rs = np.random.RandomState(seed)
    clean = (0.1 * rs.randn(B, L)).astype(np.float32)
    noisy = (clean + 0.05 * rs.randn(B, L)).astype(np.float32)
    return torch.from_numpy(clean), torch.from_numpy(noisy)
