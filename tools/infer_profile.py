"""BASELINE config 4 (batch-1 enhancement of a 10 s clip, eager launches) for a rocprofv3 kernel trace: python3 tools/infer_profile.py [L] [n]"""
import os
import sys
import types

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import speech_enhancement_amd as S  # noqa: E402
from speech_enhancement_amd import inference as INF  # noqa: E402

Lc = int(sys.argv[1]) if len(sys.argv) > 1 else 160000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 5
torch.manual_seed(0)
G = S.TSCNet(64, 201)
G.apply(S.kaiming_init)
G.cuda().eval()
cfg = types.SimpleNamespace(N_FFT=400, HOP_SAMPLES=100)
x = (0.1 * np.random.RandomState(0).randn(Lc)).astype(np.float32)
for _ in range(n):
    INF.predict(G, cfg, x)
torch.cuda.synchronize()
