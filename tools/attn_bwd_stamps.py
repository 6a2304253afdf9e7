"""phase stamps of the workgroup-cooperative attention backward (SE_ATTN_DBG=32 build switch: s_memtime at 8 points of every query
tile, first 64 workgroups): prints the median cycles between the points per wave role.  usage: SE_ATTN_DBG=32 python tools/attn_bwd_stamps.py [time|freq]"""
import os, sys, ctypes as C
os.environ.setdefault('SE_ATTN_DBG', '32')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from speech_enhancement_amd import attention as A, _lib as L
from speech_enhancement_amd.weights import WeightPlan
B, T, Fq = 16, 321, 101
g = torch.Generator().manual_seed(0)
qkv = torch.randn(B * T * Fq, 192, generator=g).cuda()
E = (torch.randn(1025, 16, generator=g) * 0.5).cuda()
dO = (torch.randn(B * T * Fq, 64, generator=g) * 1e-3).cuda()
plan = WeightPlan(torch.device('cuda')); Es = plan.linear('e', E, planes='f16'); plan.run()
am = qkv.abs().max().reshape(1).clone(); dam = dO.abs().max().reshape(1).clone()
for axis in (sys.argv[1:] or ['time', 'freq']):
    geom = A.seq_geometry(B, T, Fq, axis)
    nseq, n, inner, os_, is_, ps = geom
    O, lse = A.attn_fwd(qkv, E, geom, Es=Es, qkv_amax=am)
    ntok = qkv.shape[0]
    nbytes = L.lib().se_attn_bwd_workspace_bytes(C.c_long(ntok), C.c_int(512), C.c_int(nseq), C.c_int(n))
    ws = torch.zeros((nbytes + 3) // 4 + (1 << 18), device="cuda", dtype=torch.float32)      # (+ room for the stamps)
    dqkv = torch.empty(ntok, 192, device='cuda'); dE = torch.zeros_like(E)
    for _ in range(2):
        L.call('se_attn_bwd_f16_phase', L.ptr(qkv), L.ptr(E), L.ptr(O), L.ptr(dO), L.ptr(lse), L.ptr(am), L.ptr(dam), L.ptr(None),
               L.ptr(dqkv), L.ptr(dE), C.c_int(nseq), C.c_int(n), C.c_int(inner), C.c_long(os_), C.c_long(is_), C.c_long(ps),
               C.c_long(ntok), C.c_int(512), C.c_float(0.25), L.ptr(ws), C.c_size_t(nbytes), C.c_int(1), L.stream())
    torch.cuda.synchronize()
    nkt = (n + 15) // 16
    R = 1025; ET = (R + 16 + 15) // 16 * 16
    al = lambda x: (x + 255) & ~255
    des = al(ntok * 16) + al(3 * R * 32) + al(3 * 16 * ET * 2)
    off = des // 4 + nseq * 4 * 2 * nkt * 256
    st = ws[off: off + 64 * 4 * nkt * 8].view(torch.int32).cpu().numpy().astype(np.int64).reshape(64, 4, nkt, 8)
    d = np.diff(st, axis=3) % (1 << 32)                       # [wg, wave, qt, 7 intervals]
    nxt = (st[:, :, 1:, 0] - st[:, :, :-1, 7]) % (1 << 32)
    names = ['P1 key phase', 'stage_store', 'wait Ba', 'consume', 'U+dq+frags', 'wait Bb', 'reduce+flush']
    print(axis, 'median cycles per query tile (all waves / slowest wave of a workgroup):')
    for k, nm in enumerate(names):
        print(f'  {nm:14s} {np.median(d[:, :, 1:-1, k]):8.0f}   max-wave {np.median(d[:, :, 1:-1, k].max(axis=1)):8.0f}')
    print(f'  loop back      {np.median(nxt):8.0f}')
    tot = (st[:, :, 1:, 0] - st[:, :, :-1, 0]) % (1 << 32)
    print(f'  whole tile     {np.median(tot):8.0f}')
