"""micro-benchmark of the conv weight-gradient kernel (dense layer shape) and of the split-bf16 conv forward."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speech_enhancement_amd import gemm as GM, _lib as L, layers as LY
B, T, Fq = 16, 321, 201
Cin = int(sys.argv[1]) if len(sys.argv) > 1 else 256
mode = sys.argv[2] if len(sys.argv) > 2 else 'wgrad'
prec = int(sys.argv[3]) if len(sys.argv) > 3 else 2
skip = torch.randn(B, T, Fq, 256, device='cuda')
dR = torch.randn(B, T, Fq, 64, device='cuda')
taps = LY.dense_taps(3)
if mode == 'wgrad':
    d = GM.make_desc(B, T, Fq, T, Fq, taps, Cin, 256, 64, 64, precision=prec)
    dwp = torch.zeros(64, len(taps) * Cin, device='cuda')
    ch = int(os.environ['CHUNKS']) if 'CHUNKS' in os.environ else None
    f = lambda: GM.gemm_tap_wgrad(d, skip, dR, dwp, None, chunks=ch)
else:
    w = torch.randn(64, Cin, 2, 3, device='cuda') * 0.02
    wp = GM.pack_conv_fwd(w)
    d = GM.make_desc(B, T, Fq, T, Fq, taps, Cin, 256, 64, 64, precision=prec)
    f = lambda: GM.gemm_tap(d, skip, wp, dR)
for _ in range(2):
    f()
torch.cuda.synchronize()
t0 = time.time()
for _ in range(4):
    f()
torch.cuda.synchronize()
dt = (time.time() - t0) / 4
print(f'{mode} Cin={Cin}: {dt*1e6:.0f} us, {2.0*B*T*Fq*64*6*Cin/dt/1e12:.1f} TF')
