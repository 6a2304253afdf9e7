"""40 train steps (bench configuration, batch 4, dropout off) with the HIP-stream concurrency on and off from identical
initial state: loss trajectories and final parameter norms side by side.  A race would show as NaN / a jump; rounding-order
differences (fp32 atomics, re-ordered gradient sums) grow slowly with the step count."""
import os, sys, types, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import speech_enhancement_amd as S
from speech_enhancement_amd import train as TR, optim, gemm as GM
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
res = {}
for mode in ('serial', 'streams'):
    on = mode == 'streams'
    GM._LeafStream.enabled, TR._D_OVERLAP, GM.branch_stream.enabled = on, on, on
    torch.manual_seed(0)
    G, D = S.TSCNet(64, 201), S.Discriminator(16)
    G.apply(S.kaiming_init); D.apply(S.kaiming_init)
    G.cuda().train(); D.cuda().train(); G.set_dropout(0.0, 0.0)
    for m in D.modules():
        if isinstance(m, torch.nn.Dropout): m.p = 0.0
    a = types.SimpleNamespace(optimizer='adamw', lr=5e-4, weight_decay=0.01, momentum=0.9, max_norm=0.0)
    og, od = optim.build_optimizer(a, G), optim.build_optimizer(a, D)
    gen = torch.Generator().manual_seed(1)
    traj = []
    for s in range(steps):
        clean = (0.1 * torch.randn(4, 16000, generator=gen)).cuda(); noisy = clean + (0.05 * torch.randn(4, 16000, generator=gen)).cuda()
        q = (0.2 + 0.7 * torch.rand(4, generator=gen)).cuda()
        out = TR.gan_step(G, D, og, od, clean, noisy, 'cmgan', (0.1, 0.9, 0.2, 0.05), labels={'est': q})
        traj.append((float(out['loss_g']), float(out['loss_d'])))
    torch.cuda.synchronize()
    res[mode] = (traj, float(sum(p.double().pow(2).sum() for p in G.parameters()).sqrt()), float(sum(p.double().pow(2).sum() for p in D.parameters()).sqrt()))
for s in range(0, steps, max(1, steps // 10)):
    a, b = res['serial'][0][s], res['streams'][0][s]
    print(f'step {s:3d}  loss_g {a[0]:.6f} / {b[0]:.6f}   loss_d {a[1]:.6f} / {b[1]:.6f}')
a, b = res['serial'][0][-1], res['streams'][0][-1]
print(f'last      loss_g {a[0]:.6f} / {b[0]:.6f}   loss_d {a[1]:.6f} / {b[1]:.6f}')
print('|G| ', res['serial'][1], res['streams'][1], ' |D| ', res['serial'][2], res['streams'][2])
bad = [x for t in res['streams'][0] for x in t if x != x]
print('NaNs with streams:', len(bad))
