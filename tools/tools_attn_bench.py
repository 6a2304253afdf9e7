"""micro-benchmark of the attention kernels at the bench shapes (B=16): ablation via SE_ATTN_DBG."""
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
    O, lse = A.attn_fwd(qkv, E, geom)
    for dbg in [int(x) for x in sys.argv[1:]] or [0]:
        os.environ['SE_ATTN_DBG'] = str(dbg)
        dE = torch.zeros_like(E)
        for _ in range(2):
            A.attn_bwd(qkv, E, O, dO, lse, geom, dE)
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(3):
            A.attn_bwd(qkv, E, O, dO, lse, geom, dE)
        torch.cuda.synchronize()
        tb = (time.time() - t0) / 3
        t0 = time.time()
        for _ in range(3):
            A.attn_fwd(qkv, E, geom)
        torch.cuda.synchronize()
        tf = (time.time() - t0) / 3
        print(f'{axis} dbg={dbg}: bwd {tb*1e3:.2f} ms  fwd {tf*1e3:.2f} ms', flush=True)
