"""inference_gan.py command line (flags :29-52).  Enhances every wav under config.DATA.TEST_NOISY_DIR when a wav
reader is available (soundfile / scipy.io.wavfile); the objective-metric loop is out of scope."""
import argparse
import glob
import os

import numpy as np
import torch

from .config import get_config
from .inference import load_model, predict


def parse_option(argv=None):
    p = argparse.ArgumentParser(description='runs GAN speech-enhancement inference')
    p.add_argument('--output', '-o', type=str, required=True)
    p.add_argument('--model_path', '-m', type=str, required=True, metavar='FILE')
    p.add_argument('--cfg', type=str, required=True, metavar='FILE')
    p.add_argument('--save', action='store_true')
    p.add_argument('--validate-epochs', action='store_true')
    p.add_argument('--start', default=None, type=int)
    p.add_argument('--end', default=None, type=int)
    p.add_argument('--gpu', default=0, type=int)
    p.add_argument('--opts', default=None, nargs='+')
    args, _ = p.parse_known_args(argv)
    return args, get_config(args)


def main(argv=None):
    from scipy.io import wavfile
    args, config = parse_option(argv)
    device = torch.device('cuda', args.gpu)
    model = load_model(args.model_path, config, device)
    os.makedirs(args.output, exist_ok=True)
    for path in sorted(glob.glob(f'{config.DATA.TEST_NOISY_DIR}/*.wav')):
        sr, x = wavfile.read(path)
        if sr != config.SAMPLE_RATE:
            raise RuntimeError(f'{path}: sample rate {sr} != {config.SAMPLE_RATE} (resampling is outside the hot path)')
        x = x.astype(np.float32) / (32768.0 if x.dtype == np.int16 else 1.0)
        y = predict(model, config, x, device)
        if args.save:
            wavfile.write(os.path.join(args.output, os.path.basename(path)), sr, y.astype(np.float32))


if __name__ == '__main__':
    main()
