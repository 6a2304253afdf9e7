"""inference_gan.py's model loading and per-utterance enhancement (inference_gan.py:60-100) on the HIP path.

Whole-utterance batch-1 forward exactly like the reference (InstanceNorm and attention span the utterance).  The
metric loop (compute_metrics / librosa / torchaudio) is out of scope."""
from collections import OrderedDict

import numpy as np
import torch

from . import frontend as FE
from . import ops as O
from .generator import TSCNet


def load_model(model_path, config, device=torch.device('cuda')):
    """inference_gan.py:60-72: TSCNet(64, N_FFT//2+1), checkpoint['gen_state_dict'], eval().  The reference strips
    7 characters from every key because its checkpoints always come from DDP ('module.' prefix, main_gan.py:142);
    here the prefix is stripped only where present, so single-GPU checkpoints of this package load as well."""
    model = TSCNet(num_channel=64, num_features=config.N_FFT // 2 + 1).to(device)
    checkpoint = torch.load(model_path, map_location=device)
    sd = OrderedDict((k[7:] if k.startswith('module.') else k, v) for k, v in checkpoint['gen_state_dict'].items())
    model.load_state_dict(sd)
    model.eval()
    return model


@torch.no_grad()
def predict(model, config, noisy_signal, device=torch.device('cuda')):
    """inference_gan.py:75-100: normalise by c, wrap-pad the head of the signal to a multiple of the hop, STFT ->
    generator -> iSTFT, de-normalise, truncate."""
    noisy = torch.as_tensor(np.asarray(noisy_signal), dtype=torch.float32, device=device).unsqueeze(0)
    hop, n_fft = config.HOP_SAMPLES, config.N_FFT
    c = O.clip_scale(noisy.contiguous())
    length = noisy.size(-1)
    frame_num = int(np.ceil(length / hop))
    padding_len = frame_num * hop - length
    noisy = torch.cat([noisy, noisy[:, :padding_len]], dim=-1)
    planes, _ = FE.stft_planes(noisy, n_fft, hop, 'pow', scale=c)
    est = model.forward_planes(planes)
    est_audio = FE.istft_planes(est, n_fft, hop, 'pow') / c[:, None]
    est_audio = torch.flatten(est_audio)[:length].cpu().numpy()
    assert len(est_audio) == length, "Estimated audio and the origin audio must have the same length"
    return est_audio


class GraphedEnhancer:
    """predict() for a stream of equal-length utterances with the whole device-side pipeline (clip scale, STFT,
    generator, iSTFT, de-normalisation) captured once into a HIP graph and replayed per utterance.

    Batch-1 inference is launch-bound: ~1500 kernel launches cost ~20 ms of host time for ~10 ms of GPU work, a replayed
    graph removes the host side.  All kernels are launched on torch's current stream, which is the capture stream
    inside `torch.cuda.graph`; the dynamic-LDS attributes are raised by the eager warm-up run before the capture."""

    def __init__(self, model, config, length, device=torch.device('cuda')):
        self.model, self.config, self.length, self.device = model, config, int(length), device
        hop = config.HOP_SAMPLES
        self.padding_len = int(np.ceil(self.length / hop)) * hop - self.length
        self.static_in = torch.zeros(1, self.length, device=device, dtype=torch.float32)
        self.static_out = None
        with torch.no_grad():
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(2):
                    self._device_pipeline()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self.static_out = self._device_pipeline()
        torch.cuda.synchronize()

    def _device_pipeline(self):
        cfg = self.config
        noisy = self.static_in
        c = O.clip_scale(noisy)
        noisy = torch.cat([noisy, noisy[:, :self.padding_len]], dim=-1)
        planes, _ = FE.stft_planes(noisy, cfg.N_FFT, cfg.HOP_SAMPLES, 'pow', scale=c)
        est = self.model.forward_planes(planes)
        return FE.istft_planes(est, cfg.N_FFT, cfg.HOP_SAMPLES, 'pow') / c[:, None]

    @torch.no_grad()
    def __call__(self, noisy_signal):
        x = torch.as_tensor(np.asarray(noisy_signal), dtype=torch.float32).reshape(1, -1)
        if x.shape[1] != self.length:
            raise ValueError(f'GraphedEnhancer was captured for {self.length} samples, got {x.shape[1]}')
        self.static_in.copy_(x, non_blocking=True)
        self.graph.replay()
        return torch.flatten(self.static_out)[:self.length].cpu().numpy()
