import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speech_enhancement_amd import ops as O, attention as A
B, T, Fq = 16, 321, 101
x = torch.randn(B * T * Fq, 128, device='cuda'); dy = torch.randn_like(x)
w = torch.randn(128, 31, device='cuda') * 0.1; b = torch.randn(128, device='cuda')
for axis in ('time', 'freq'):
    geom = A.seq_geometry(B, T, Fq, axis)
    st = torch.zeros(1, 128, 2, device='cuda', dtype=torch.float64)
    z = torch.randn(B * T * Fq, 256, device='cuda')
    for name, fn in (('fwd+stats', lambda: O.dwconv31(x, w, b, geom, stats=st)), ('dgrad', lambda: O.dwconv31(dy, w, None, geom, flip=True)),
                     ('dgrad+glu', lambda: O.dwconv31_glu_bwd(dy, w, z, geom)),
                     ('wgrad', lambda: O.dwconv31_wgrad(x, dy, torch.zeros(128, 31, device='cuda'), torch.zeros(128, device='cuda'), geom))):
        for _ in range(2): fn()
        torch.cuda.synchronize(); t0 = time.time()
        for _ in range(5): fn()
        torch.cuda.synchronize(); dt = (time.time() - t0) / 5
        print(f'{axis:5s} {name:10s} {dt*1e6:8.1f} us  {(5 if name == "dgrad+glu" else 2)*x.numel()*4/dt/1e9:7.0f} GB/s', flush=True)
