"""BASELINE config 5: CDiffuSE 50-step supportive reverse diffusion, batch 32, 2 s clips (synthetic), one MI355X.
Prints utterances/s, the one-off conditioner cost and the per-step time."""
import json, os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import speech_enhancement_amd as S
from speech_enhancement_amd import _lib

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
Ls = 32000
sched_tr = np.linspace(1e-4, 0.035, 50).tolist()
cfg = types.SimpleNamespace(NOISE_SCHEDULE=sched_tr, INFERENCE_NOISE_SCHEDULE=[0.0001, 0.001, 0.01, 0.05, 0.2, 0.35], N_FFT=400, HOP_SAMPLES=100)
torch.manual_seed(0)
m = S.DiffuSE(10, 100, 201, sched_tr, 64, 30).cuda().eval()
torch.nn.init.normal_(m.output_projection.weight, std=0.05)
x = (0.1 * torch.randn(B, Ls)).numpy()
sched = S.inference_schedule(cfg, fast_sampling=False)
S.predict_diffuse(m, cfg, x[:2], *sched)            # warm-up (LDS attributes, caches)
torch.cuda.synchronize()
by_streams = {}
for ns in (1, 2, 3, 2, 1):
    t0 = time.time()
    S.predict_diffuse(m, cfg, x, *sched, streams=ns)
    torch.cuda.synchronize()
    by_streams.setdefault(ns, []).append(round(B / (time.time() - t0), 2))
dt = B / max(by_streams[2])
# phases
noisy = torch.from_numpy(x).cuda()
from speech_enhancement_amd import frontend as FE
planes, _ = FE.stft_planes(noisy, 400, 100, 'none', padded=False)
spec = planes[..., 0].transpose(1, 2).contiguous()
torch.cuda.synchronize(); t0 = time.time()
cond = m.conditioner(spec)
torch.cuda.synchronize(); tc = time.time() - t0
audio = torch.zeros(B, 100 * spec.shape[-1], device='cuda')
_lib.TIMER.start()
torch.cuda.synchronize(); t0 = time.time()
for _ in range(5):
    m.denoise(audio, cond, torch.tensor([3.5], device='cuda'))
torch.cuda.synchronize(); ts = (time.time() - t0) / 5
_lib.TIMER.stop()
fam = {k: round(v['ms'] / 5, 3) for k, v in sorted(_lib.TIMER.summary().items(), key=lambda kv: -kv[1]['ms'])}
print(json.dumps({'metric': 'utterances/sec CDiffuSE 50-step supportive reverse diffusion (2 s @16 kHz)', 'value': round(B / dt, 2),
                  'batch': B, 'utt_per_s_by_streams': by_streams, 'seconds_per_batch': round(dt, 3), 'conditioner_ms_once': round(tc * 1e3, 1),
                  'step_ms': round(ts * 1e3, 2), 'conditioner_cache_GB': round(sum(c.numel() for c in cond) * 4 / 1e9, 2),
                  'gemm_families_ms_per_step': fam}))
