"""Host utilities of utils/utils.py: LR schedule, init, checkpoint I/O."""
import math
import os
import shutil

import torch
import torch.nn as nn


def adjust_learning_rate(optimizers, epoch, config):
    """utils/utils.py:78-90: linear warm-up then half-cosine per cycle, peak halved each cycle; written to every
    param group of every optimizer; returns lr + MIN_LR."""
    s = config.TRAIN.SCHEDULER
    cycle = s.EPOCHS // s.CYCLE_LIMIT
    q, r = divmod(epoch, cycle)
    if r < s.WARMUP_EPOCHS:
        lr = 0.5 ** q * s.LR * r / s.WARMUP_EPOCHS
    else:
        lr = s.LR * 0.5 ** (q + 1) * (1. + math.cos(math.pi * (r - s.WARMUP_EPOCHS) / (cycle - s.WARMUP_EPOCHS)))
    for opt in optimizers:
        for g in opt.param_groups:
            g['lr'] = lr
    return lr + s.MIN_LR


def kaiming_init(m):
    """utils/utils.py:92-104 (reaches spectral-norm weight_orig too)."""
    if isinstance(m, (nn.Linear, nn.Conv2d, nn.Conv1d)):
        nn.init.kaiming_normal_(m.weight)
        if m.bias is not None:
            m.bias.data.fill_(0.01)
    elif hasattr(m, 'weight_orig'):
        nn.init.kaiming_normal_(m.weight_orig)
        if getattr(m, 'bias', None) is not None:
            m.bias.data.fill_(0.01)


def save_checkpoint(state, path, is_best, filename='checkpoint.pth.tar'):
    """utils/utils.py:68-75."""
    os.makedirs(path, exist_ok=True)
    torch.save(state, os.path.join(path, filename))
    if is_best:
        shutil.copyfile(os.path.join(path, filename), os.path.join(path, 'model_best.pth.tar'))


class AverageMeter:
    def __init__(self):
        self.val = self.avg = self.sum = self.count = 0

    def update(self, v, n=1):
        self.val = v
        self.sum += v * n
        self.count += n
        self.avg = self.sum / self.count
