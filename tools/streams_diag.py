import sys, types, torch
import os; ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + '/tests')
import speech_enhancement_amd as S
from speech_enhancement_amd import train as TR, optim, gemm as GM
sys.path.insert(0, ROOT + '/tests/golden')
import formula
def load(kind):
    m = S.TSCNet(64, 201) if kind == 'generator' else S.Discriminator(16)
    m.load_state_dict(formula.formula_state(kind)); m.cuda().train()
    if kind == 'generator': m.set_dropout(0.0, 0.0)
    else:
        for x in m.modules():
            if isinstance(x, torch.nn.Dropout): x.p = 0.0
    return m
torch.manual_seed(5)
clean = 0.1 * torch.randn(4, 3200, device='cuda'); noisy = clean + 0.05 * torch.randn(4, 3200, device='cuda')
labels = {'est': torch.tensor([0.4, 0.6, 0.5, 0.7], device='cuda'), 'clean': torch.full((4,), 0.96, device='cuda'), 'noisy': torch.tensor([0.3, 0.2, 0.25, 0.35], device='cuda')}
ARCH = sys.argv[1] if len(sys.argv) > 1 else 'cmgan'
w = (0.1, 0.9, 0.2, 0.05) if ARCH == 'cmgan' else (0.3, 0.7, 0.2, 0.05)
TR._D_OVERLAP = False
runs = []
MODES = [(False, False), (False, False), (True, False), (False, True), (True, True), (False, False)]      # (leaf stream, decoder branch stream)
for leaf, branch in MODES:
    GM._LeafStream.enabled, GM.branch_stream.enabled = leaf, branch
    g, d = load('generator'), load('discriminator')
    init = {n: p.detach().clone() for n, p in g.named_parameters()}
    args = types.SimpleNamespace(optimizer='sgd', lr=0.01, weight_decay=0.01, momentum=0.9, max_norm=0.0)
    og, od = optim.build_optimizer(args, g), optim.build_optimizer(args, d, lr=0.02)
    for _ in range(int(os.environ.get('STEPS', '2'))): TR.gan_step(g, d, og, od, clean, noisy, ARCH, w, labels=labels)
    torch.cuda.synchronize()
    runs.append({n: p.detach().clone() for n, p in g.named_parameters()})
names = ['complex_decoder.dense_block.conv4.weight', 'dense_encoder.dilated_dense.conv3.weight', 'dense_encoder.conv_1.0.weight', 'mask_decoder.dense_block.conv2.weight', 'TSCB_1.time_conformer.ff1.fn.fn.net.0.weight']
for n in names:
    upd = float((runs[0][n] - init[n]).abs().max())
    print(n, 'update', f'{upd:.2e}', 'diffs vs run0:', ' '.join(f'{float((runs[0][n]-r[n]).abs().max()):.1e}' for r in runs[1:]))
worst = []
for n in runs[0]:
    upd = float((runs[0][n] - init[n]).abs().max()) + 1e-12
    worst.append((max(float((runs[0][n]-r[n]).abs().max()) for r in runs[2:5]) / upd, float((runs[0][n]-runs[1][n]).abs().max())/upd, float((runs[0][n]-runs[5][n]).abs().max())/upd, n))
worst.sort(reverse=True)
print('modes', MODES[1:]); print('worst relative-to-update (stream runs | serial1 | serial5):')
for x in worst[:12]: print(f'{x[0]:.2e} {x[1]:.2e} {x[2]:.2e} {x[3]}')
