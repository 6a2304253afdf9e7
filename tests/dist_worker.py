"""One rank of the multi-rank acceptance test: runs ONE gan_step on this rank's shard of a
fixed global batch and, on rank 0, writes losses / post-step parameter norms / BatchNorm running statistics /
self-correcting weights to an .npz file.  Started as a child process by tests/test_00_dist_gpu.py (the parent never
touches the GPU).  world == 1 is the single-process run at the same global batch.

    python tests/dist_worker.py <rank> <world> <port> <arch> <backend> <out.npz> [global_batch] [samples]

backend "nccl" = RCCL, one GPU per rank; backend "gloo" = all ranks share GPU 0 (RCCL refuses that), the hooks stage the
collectives through the host -- same step logic either way."""
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    rank, world, port = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    arch, backend, out_path = sys.argv[4], sys.argv[5], sys.argv[6]
    GB = int(sys.argv[7]) if len(sys.argv) > 7 else 4
    Ls = int(sys.argv[8]) if len(sys.argv) > 8 else 3200
    dev_id = rank if backend == 'nccl' else 0
    torch.cuda.set_device(dev_id)
    import formula
    import speech_enhancement_amd as S
    from speech_enhancement_amd import train as TR, optim
    hooks = None
    g, d = S.TSCNet(64, 201), S.Discriminator(16)
    g.load_state_dict(formula.formula_state('generator'))
    d.load_state_dict(formula.formula_state('discriminator'))
    g.cuda().train()
    d.cuda().train()
    g.set_dropout(0.0, 0.0)
    for m in d.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    if world > 1:
        os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        kw = {'device_id': torch.device('cuda', dev_id)} if backend == 'nccl' else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
        hooks = TR.attach_data_parallel(g, d)
    args = types.SimpleNamespace(optimizer='sgd', lr=0.01, weight_decay=0.01, momentum=0.9, max_norm=0.0)
    og, od = optim.build_optimizer(args, g), optim.build_optimizer(args, d, lr=0.02)
    rs = np.random.RandomState(17)
    clean = (0.1 * rs.randn(GB, Ls)).astype(np.float32)
    noisy = (clean + 0.05 * rs.randn(GB, Ls)).astype(np.float32)
    q = {'est': np.linspace(0.3, 0.8, GB).astype(np.float32), 'clean': np.linspace(0.9, 0.99, GB).astype(np.float32),
         'noisy': np.linspace(0.2, 0.5, GB).astype(np.float32)}
    per = GB // world
    sl = slice(rank * per, (rank + 1) * per)
    labels = {k: torch.from_numpy(v[sl]).cuda() for k, v in q.items()}
    weights = (0.1, 0.9, 0.2, 0.05) if arch == 'cmgan' else (0.3, 0.7, 0.2, 0.05)
    out = TR.gan_step(g, d, og, od, torch.from_numpy(clean[sl]).cuda(), torch.from_numpy(noisy[sl]).cuda(), arch, weights,
                      labels=labels, hooks=hooks)
    torch.cuda.synchronize()
    names = ['loss_ri', 'loss_mag', 'time_loss', 'gan', 'L_C', 'L_E'] + (['L_N'] if arch == 'scp' else [])
    losses = torch.tensor([float(out[k]) for k in names], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(losses)          # every term is a mean over the local shard: rank mean == global-batch value
        losses /= world
    if rank == 0:
        gs, ds = g.state_dict(), d.state_dict()
        res = {'losses': losses.numpy(), 'loss_names': np.array(names),
               'g_norm': np.array([float(v.double().norm()) for v in gs.values()]),
               'd_norm': np.array([float(v.double().norm()) for v in ds.values()]),
               'bn_rm': gs['TSCB_2.freq_conformer.conv.net.5.running_mean'].cpu().numpy(),
               'bn_rv': gs['TSCB_2.freq_conformer.conv.net.5.running_var'].cpu().numpy(),
               'g_w': gs['TSCB_3.time_conformer.conv.net.5.weight'].cpu().numpy(),       # BatchNorm gamma after the step
               'g_q': gs['TSCB_1.time_conformer.attn.fn.to_q.weight'].cpu().numpy(),
               'd_w': ds['layers.17.weight_orig'].cpu().numpy(),
               'w': np.array([float(out.get('w_E', 0.0)), float(out.get('w_N', 0.0))])}
        np.savez(out_path, **res)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
