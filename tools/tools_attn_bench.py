"""micro-benchmark of the attention kernels at the bench shapes (B=16); SE_ATTN_BWD=2 selects the v2 backward."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speech_enhancement_amd import attention as A
B, T, Fq = 16, 321, 101
g = torch.Generator().manual_seed(0)
qkv = (torch.randn(B * T * Fq, 192, generator=g)).cuda()
E = (torch.randn(1025, 16, generator=g) * 0.5).cuda()
dO = torch.randn(B * T * Fq, 64, generator=g).cuda()
for axis in ('time', 'freq'):
    geom = A.seq_geometry(B, T, Fq, axis)
    nseq, n = geom[0], geom[1]
    O, lse = A.attn_fwd(qkv, E, geom)
    for mode in sys.argv[1:] or ['3']:
        os.environ['SE_ATTN_BWD'] = mode.split(':')[0]
        os.environ['SE_ATTN_DBG'] = mode.split(':')[1] if ':' in mode else '0'
        dE = torch.zeros_like(E)
        for _ in range(2):
            A.attn_bwd(qkv, E, O, dO, lse, geom, dE)
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(5):
            A.attn_bwd(qkv, E, O, dO, lse, geom, dE)
        torch.cuda.synchronize()
        tb = (time.time() - t0) / 5
        t0 = time.time()
        for _ in range(5):
            A.attn_fwd(qkv, E, geom)
        torch.cuda.synchronize()
        tf = (time.time() - t0) / 5
        fl = nseq * 4 * 2.0 * n * n * 16
        print(f'{axis} bwd-mode={mode}: bwd {tb*1e3:.3f} ms ({7 * fl / tb / 1e12:.1f} TFLOP/s algorithmic)  '
              f'fwd {tf*1e3:.3f} ms ({3 * fl / tf / 1e12:.1f} TFLOP/s)', flush=True)
