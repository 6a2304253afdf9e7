"""one attention forward + backward per axis at the bench shapes (B=16), for rocprofv3 counter passes"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speech_enhancement_amd import attention as A
B, T, Fq = 16, 321, 101
g = torch.Generator().manual_seed(0)
qkv = (torch.randn(B * T * Fq, 192, generator=g)).cuda()
E = (torch.randn(1025, 16, generator=g) * 0.5).cuda()
dO = torch.randn(B * T * Fq, 64, generator=g).cuda()
for axis in ('time', 'freq'):
    geom = A.seq_geometry(B, T, Fq, axis)
    for _ in range(2):
        O, lse = A.attn_fwd(qkv, E, geom)
        dE = torch.zeros_like(E)
        A.attn_bwd(qkv, E, O, dO, lse, geom, dE)
torch.cuda.synchronize()
