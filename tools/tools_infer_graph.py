"""eager vs HIP-graph batch-1 inference latency (2 s and 10 s clips)"""
import os, sys, time, types, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import speech_enhancement_amd as S
from speech_enhancement_amd import inference as INF
torch.manual_seed(0)
G = S.TSCNet(64, 201); G.apply(S.kaiming_init); G.cuda().eval()
cfg = types.SimpleNamespace(N_FFT=400, HOP_SAMPLES=100)
for L in (32000, 160000):
    x = (0.1 * np.random.RandomState(0).randn(L)).astype(np.float32)
    for _ in range(2): INF.predict(G, cfg, x)
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(5): INF.predict(G, cfg, x)
    te = (time.time() - t0) / 5
    enh = INF.GraphedEnhancer(G, cfg, L)
    for _ in range(2): enh(x)
    t0 = time.time()
    for _ in range(5): y = enh(x)
    tg = (time.time() - t0) / 5
    print(f'L={L}: eager {te*1e3:.1f} ms, graph {tg*1e3:.1f} ms, max diff {np.abs(y - INF.predict(G, cfg, x)).max():.2e}')
