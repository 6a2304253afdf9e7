"""attention forward with the relative-position table split on the fly vs pre-split (se_attn_fwd_es), B = 16 shapes"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speech_enhancement_amd import attention as A
from speech_enhancement_amd.weights import WeightPlan
B, T, Fq = 16, 321, 101
g = torch.Generator().manual_seed(0)
qkv = torch.randn(B * T * Fq, 192, generator=g).cuda()
E = (torch.randn(1025, 16, generator=g) * 0.5).cuda()
plan = WeightPlan(torch.device('cuda'))
Es = plan.linear('es', E, planes=True)
plan.run()
for axis in ('time', 'freq'):
    geom = A.seq_geometry(B, T, Fq, axis)
    outs = {}
    for name, es in (('on the fly', None), ('pre-split', Es), ('on the fly', None), ('pre-split', Es)):
        for _ in range(2): o, lse = A.attn_fwd(qkv, E, geom, Es=es)
        torch.cuda.synchronize(); t0 = time.time()
        for _ in range(10): o, lse = A.attn_fwd(qkv, E, geom, Es=es)
        torch.cuda.synchronize(); dt = (time.time() - t0) / 10
        outs[name] = (o, lse)
        print(f'{axis:5s} {name:11s} {dt*1e3:.3f} ms', flush=True)
    print('   bit-identical:', torch.equal(outs['on the fly'][0], outs['pre-split'][0]) and torch.equal(outs['on the fly'][1], outs['pre-split'][1]))
