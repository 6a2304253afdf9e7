"""config 4: batch 1, 10 s @ 16 kHz (T = 1601) enhancement forward (inference_gan.predict path)"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import speech_enhancement_amd as S
from speech_enhancement_amd import inference as INF
torch.manual_seed(0)
G = S.TSCNet(64, 201); G.apply(S.kaiming_init); G.cuda().eval()
noisy = (0.1 * torch.randn(1, 160000)).cuda()
with torch.no_grad():
    for _ in range(2):
        y = INF.enhance(G, noisy) if hasattr(INF, 'enhance') else None
    if y is None:
        spec = S.compressed_stft(noisy, 400, 100, None)
        for _ in range(2):
            er, ei = G(spec)
        torch.cuda.synchronize(); t0 = time.time()
        for _ in range(3):
            er, ei = G(spec)
        torch.cuda.synchronize(); dt = (time.time() - t0) / 3
        print('TSCNet forward 10 s clip: %.1f ms, output finite: %s' % (dt * 1e3, bool(torch.isfinite(er).all())))
