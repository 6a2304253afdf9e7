"""main_gan.py's command line and worker (flags main_gan.py:35-115, worker :118-315) on the HIP path.

    python -m speech_enhancement_amd.main_gan -a cmgan --cfg config/baseline.yaml -b 16 --optimizer adamw --lr 5e-4 \\
        --crop-len 2 --synthetic 8

One process per GPU: launch N processes with torch.distributed.run (env:// rendezvous); `--multiprocessing-distributed`
spawns them itself like the reference.  The VoiceBank dataset / collator (librosa + pesq) is out of scope; with
`--synthetic N` the loaders are N synthetic batches per epoch with PESQ labels supplied (the bench's data recipe),
otherwise `speech_enhancement_amd.main_gan.DATASET_FACTORY` must be set to a callable returning (train, valid) loaders.
"""
import argparse
import os
import random
import types
import warnings

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from .config import get_config
from .discriminator import Discriminator
from .generator import TSCNet
from .optim import build_optimizer
from .train import attach_data_parallel, train_gan, validate_gan
from .utils import kaiming_init, save_checkpoint

model_names = ['cmgan', 'scp', 'cp', 'sc']
DATASET_FACTORY = None


def parse_option(argv=None):
    p = argparse.ArgumentParser(description='Speech enhancement training script')
    p.add_argument('-a', '--arch', metavar='ARCH', default='cmgan', choices=model_names + ['diffuse'])
    p.add_argument('--output', default='output', type=str, metavar='PATH')
    p.add_argument('--tag')
    p.add_argument('--cfg', type=str, required=True, metavar='FILE')
    p.add_argument('--opts', default=None, nargs='+')
    p.add_argument('-j', '--workers', default=32, type=int)
    p.add_argument('--epochs', default=100, type=int)
    p.add_argument('--start-epoch', default=0, type=int)
    p.add_argument('-b', '--batch-size', default=64, type=int)
    p.add_argument('--lr', '--learning-rate', default=0.01, type=float, dest='lr')
    p.add_argument('--momentum', default=0.9, type=float)
    p.add_argument('--wd', '--weight-decay', default=0.01, type=float, dest='weight_decay')
    p.add_argument('--max-norm', default=0.0, type=float)
    p.add_argument('-p', '--print-freq', default=10, type=int)
    p.add_argument('--resume', default='', type=str)
    p.add_argument('--world-size', default=-1, type=int)
    p.add_argument('--rank', default=-1, type=int)
    p.add_argument('--dist_url', default='env://')
    p.add_argument('--dist-backend', default='nccl', type=str)
    p.add_argument('--seed', default=None, type=int)
    p.add_argument('--gpu', default=None, type=int)
    p.add_argument('--multiprocessing-distributed', action='store_true')
    p.add_argument('--debug', action='store_true')
    p.add_argument('--optimizer', default='sgd', type=str, choices=['sgd', 'adamw', 'lars', 'lamb'])
    p.add_argument('--criterion', default='l1', type=str, choices=['mae', 'l1', 'mse', 'l2'])
    p.add_argument('--crop-len', default=1, type=int)
    p.add_argument('--gen-first', action='store_true')
    p.add_argument('--comp-type', default='pow', type=str, choices=['norm', 'log', 'pow', 'none'])
    p.add_argument('--synthetic', default=0, type=int, metavar='N',
                   help='train on N synthetic batches per epoch (labels supplied) instead of VoiceBank-DEMAND')
    args, _ = p.parse_known_args(argv)
    if args.arch == 'diffuse':      # the reference's default is not in its own choices and selects the cmgan branch
        args.arch = 'cmgan'
    return args, get_config(args)


def synthetic_loader(n_batches, batch, samples, seed):
    out = []
    for i in range(n_batches):
        g = torch.Generator().manual_seed(seed + i)
        clean = 0.1 * torch.randn(batch, samples, generator=g)
        noisy = clean + 0.05 * torch.randn(batch, samples, generator=g)
        q = 0.2 + 0.7 * torch.rand(batch, generator=g)
        out.append({'audio': clean, 'noisy': noisy,
                    'labels': {'est': q, 'clean': torch.full_like(q, 0.97), 'noisy': 0.5 * q}})
    return out


def main_worker(gpu, ngpus_per_node, args, config):
    args.gpu = gpu
    if args.distributed:
        if args.dist_url == 'env://' and args.rank == -1:
            args.rank = int(os.environ['RANK'])
        if args.multiprocessing_distributed:
            args.rank = args.rank * ngpus_per_node + gpu
        if args.gpu is None:
            args.gpu = int(os.environ.get('LOCAL_RANK', 0))
        torch.cuda.set_device(args.gpu)
        dist.init_process_group(backend=args.dist_backend, init_method=args.dist_url, world_size=args.world_size,
                                rank=args.rank)
        dist.barrier()
    elif args.gpu is None:
        args.gpu = 0
    torch.cuda.set_device(args.gpu)
    model = TSCNet(num_channel=64, num_features=config.N_FFT // 2 + 1)
    discriminator = Discriminator(ndf=16)
    model.apply(kaiming_init)
    discriminator.apply(kaiming_init)
    model.cuda(args.gpu)
    discriminator.cuda(args.gpu)
    if args.distributed:
        args.batch_size = int(args.batch_size / args.world_size)
        attach_data_parallel(model, discriminator)
    criterion = torch.nn.MSELoss()
    optimizer = build_optimizer(args, model)
    optimizer_disc = build_optimizer(args, discriminator, lr=args.lr * 2)
    best_loss = 1e8
    if args.resume and os.path.isfile(args.resume):
        ck = torch.load(args.resume, map_location=f'cuda:{args.gpu}')
        args.start_epoch = ck['epoch']
        strip = lambda sd: {(k[7:] if k.startswith('module.') else k): v for k, v in sd.items()}
        model.load_state_dict(strip(ck['gen_state_dict']))
        discriminator.load_state_dict(strip(ck['disc_state_dict']))
        optimizer.load_state_dict(ck['optimizer'])
        optimizer_disc.load_state_dict(ck['optimizer_disc'])
        best_loss = ck['best_loss']
    samples = config.CROP_FRAMES * config.HOP_SAMPLES * config.CROP_LEN
    if args.synthetic:
        rank = args.rank if args.distributed else 0
        train_loader = [{k: (v.cuda(args.gpu) if torch.is_tensor(v) else {a: b.cuda(args.gpu) for a, b in v.items()})
                         for k, v in b.items()} for b in synthetic_loader(args.synthetic, args.batch_size, samples, 1 + 1000 * rank)]
        valid_loader = train_loader[:1]
    elif DATASET_FACTORY is not None:
        train_loader, valid_loader = DATASET_FACTORY(args, config)
    else:
        raise RuntimeError('the VoiceBank dataset / collator is outside this package: pass --synthetic N or set '
                           'speech_enhancement_amd.main_gan.DATASET_FACTORY')
    import logging
    logger = logging.getLogger(args.arch)
    if not args.distributed or args.rank == 0:
        logging.basicConfig(level=logging.INFO)
    for epoch in range(args.start_epoch, args.epochs):
        tg, td = train_gan(train_loader, model, discriminator, criterion, optimizer, optimizer_disc, logger, epoch, args,
                           config)
        vg, vd = validate_gan(valid_loader, model, discriminator, criterion, logger, epoch, args, config)
        if not args.distributed or args.rank == 0:
            is_best = vd <= best_loss
            best_loss = min(best_loss, vd)
            pre = 'module.' if args.distributed else ''
            save_checkpoint({'epoch': epoch + 1, 'arch': args.arch,
                             'gen_state_dict': {pre + k: v for k, v in model.state_dict().items()},
                             'disc_state_dict': {pre + k: v for k, v in discriminator.state_dict().items()},
                             'optimizer': optimizer.state_dict(), 'optimizer_disc': optimizer_disc.state_dict(),
                             'best_loss': best_loss}, config.OUTPUT, is_best=is_best,
                            filename='checkpoint_{:04d}.pth.tar'.format(epoch))
            logger.info(f'Train Generator Loss: {tg:.3f}\tTrain Discriminator Loss: {td:.3f}\t'
                        f'Validation Generator Loss: {vg:.3f}\tValidation Discriminator Loss: {vd:.3f}')
    if args.distributed:
        dist.destroy_process_group()


def main(argv=None):
    args, config = parse_option(argv)
    if args.seed is not None:
        random.seed(args.seed)
        torch.manual_seed(args.seed)
        warnings.warn('You have chosen to seed training.')
    if args.dist_url == 'env://' and args.world_size == -1:
        args.world_size = int(os.environ.get('WORLD_SIZE', 1))
    args.distributed = args.world_size > 1 or args.multiprocessing_distributed
    ngpus = torch.cuda.device_count()
    if args.multiprocessing_distributed:
        args.world_size = ngpus * args.world_size
        mp.spawn(main_worker, nprocs=ngpus, args=(ngpus, args, config))
    else:
        main_worker(args.gpu, ngpus, args, config)


if __name__ == '__main__':
    main()
