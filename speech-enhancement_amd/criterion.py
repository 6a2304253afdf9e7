"""core/criterion.py:11-21."""
import torch.nn as nn


def build_criterion(criterion_type):
    name = criterion_type.lower()
    if name in ('mae', 'l1'):
        print('Criterion: MAE Loss')
        return nn.L1Loss()
    if name in ('mse', 'l2'):
        print('Criterion: MSE Loss')
        return nn.MSELoss()
    print('Invalid criterion!')
    return None
