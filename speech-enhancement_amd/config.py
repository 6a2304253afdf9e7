"""Config surface of config/default.py (yacs is absent here: a small attribute-dict with the same keys,
``BASE`` yaml chaining, ``--opts KEY VAL`` merging and the args -> config overrides of config/default.py:81-135)."""
import copy
import os

import yaml


class CfgNode(dict):
    def __init__(self, d=None):
        super().__init__()
        for k, v in (d or {}).items():
            self[k] = CfgNode(v) if isinstance(v, dict) else v
        self.__dict__['_frozen'] = False

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        if self.__dict__.get('_frozen'):
            raise AttributeError(f'config is frozen; cannot set {k}')
        self[k] = v

    def defrost(self):
        self.__dict__['_frozen'] = False
        for v in self.values():
            if isinstance(v, CfgNode):
                v.defrost()

    def freeze(self):
        self.__dict__['_frozen'] = True
        for v in self.values():
            if isinstance(v, CfgNode):
                v.freeze()

    def clone(self):
        return CfgNode(copy.deepcopy(dict(self)))

    def merge(self, other):
        for k, v in other.items():
            if k not in self:
                raise KeyError(f'non-existent config key: {k}')
            if isinstance(v, dict) and isinstance(self[k], CfgNode):
                self[k].merge(v)
            else:
                self[k] = v

    def merge_from_list(self, lst):
        assert len(lst) % 2 == 0
        for k, v in zip(lst[0::2], lst[1::2]):
            node = self
            parts = k.split('.')
            for p in parts[:-1]:
                node = node[p]
            if parts[-1] not in node:
                raise KeyError(f'non-existent config key: {k}')
            old = node[parts[-1]]
            if isinstance(v, str) and not isinstance(old, str):
                v = yaml.safe_load(v)
                if isinstance(v, str) and isinstance(old, (int, float)):     # '1e-5' is a string in YAML 1.1
                    v = type(old)(float(v))
                elif isinstance(old, float) and isinstance(v, int):
                    v = float(v)
            node[parts[-1]] = v


def defaults():
    """config/default.py:17-63."""
    return CfgNode({
        'SAMPLE_RATE': 16000, 'N_SPECS': 201, 'N_FFT': 400, 'HOP_SAMPLES': 100, 'CROP_FRAMES': 160,
        'RESIDUAL_LAYERS': 30, 'RESIDUAL_CHANNELS': 64, 'DILATION_CYCLE_LENGTH': 10, 'NOISE_SCHEDULE': 50,
        'INFERENCE_NOISE_SCHEDULE': [0.0001, 0.001, 0.01, 0.05, 0.2, 0.35], 'CROP_LEN': 1,
        'LOSS_WEIGHTS': [0.1, 0.9, 0.2, 0.05],
        'DATA': {'TRAIN_CLEAN_DIR': 'D:/data/DS_10283_2791/clean_trainset_28spk_wav',
                 'TRAIN_NOISY_DIR': 'D:/data/DS_10283_2791/noisy_trainset_28spk_wav',
                 'TEST_CLEAN_DIR': 'D:/data/DS_10283_2791/clean_testset_wav',
                 'TEST_NOISY_DIR': 'D:/data/DS_10283_2791/noisy_testset_wav', 'BATCH_SIZE': 32},
        'TRAIN': {'OPTIMIZER': {'NAME': 'sgd'}, 'CRITERION': {'NAME': 'l1'},
                  'SCHEDULER': {'LR': 1e-2, 'EPOCHS': 100, 'CYCLE_LIMIT': 4, 'WARMUP_EPOCHS': 4, 'MIN_LR': 1e-6}},
        'MODEL': {'NAME': 'diffuse', 'RESUME': ''},
        'OUTPUT': '', 'TAG': 'default'})


def _merge_file(config, path):
    with open(path) as f:
        y = yaml.safe_load(f) or {}
    for base in y.pop('BASE', ['']) if 'BASE' in y else []:
        if base:
            _merge_file(config, os.path.join(os.path.dirname(path), base))
    config.merge(y)


def get_config(args):
    """config/default.py:128-135 (+ update_config :81-126)."""
    config = defaults()
    if getattr(args, 'cfg', None):
        _merge_file(config, args.cfg)
    if getattr(args, 'opts', None):
        config.merge_from_list(args.opts)
    if getattr(args, 'batch_size', None):
        config.DATA.BATCH_SIZE = args.batch_size
    if getattr(args, 'arch', None):
        config.MODEL.NAME = args.arch
    if getattr(args, 'output', None):
        config.OUTPUT = args.output
    if getattr(args, 'tag', None):
        config.TAG = args.tag
    if getattr(args, 'lr', None):
        config.TRAIN.SCHEDULER.LR = args.lr
    if getattr(args, 'epochs', None):
        config.TRAIN.SCHEDULER.EPOCHS = args.epochs
    if getattr(args, 'crop_len', None):
        config.CROP_LEN = args.crop_len
    config.OUTPUT = os.path.join(config.OUTPUT, config.MODEL.NAME, config.TAG)
    config.freeze()
    return config
