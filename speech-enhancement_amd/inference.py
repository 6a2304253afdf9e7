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
    """inference_gan.py:60-72: TSCNet(64, N_FFT//2+1), checkpoint['gen_state_dict'] with the 7-char 'module.' prefix
    stripped unconditionally (checkpoints are saved from DDP), eval()."""
    model = TSCNet(num_channel=64, num_features=config.N_FFT // 2 + 1).to(device)
    checkpoint = torch.load(model_path, map_location=device)
    sd = OrderedDict((k[7:], v) for k, v in checkpoint['gen_state_dict'].items())
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
