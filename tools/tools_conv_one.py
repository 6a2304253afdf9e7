import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speech_enhancement_amd import gemm as GM, _lib as L, layers as LY
B, T, Fq = 16, 321, 201
Cin = int(sys.argv[1]) if len(sys.argv) > 1 else 256
skip = torch.randn(B, T, Fq, 256, device='cuda')
if os.environ.get('ZERO') == '1':      # all-zero operands: the clock the chip holds under load depends on the data (DVFS)
    skip.zero_()
w = torch.randn(64, Cin, 2, 3, device='cuda') * 0.02
wp = GM.pack_conv_fwd(w)
if len(sys.argv) > 3 and sys.argv[3] == 'planes':      # pre-split weights (weights.WeightPlan)
    from speech_enhancement_amd.weights import WeightPlan
    plan = WeightPlan(torch.device('cuda'))
    wp = plan.conv_fwd('w', w, planes='f16' if len(sys.argv) > 2 and sys.argv[2] == '3' else True)
    plan.run()
y = torch.empty(B, T, Fq, 64, device='cuda')
prec = int(sys.argv[2]) if len(sys.argv) > 2 else 2
d = GM.make_desc(B, T, Fq, T, Fq, LY.dense_taps(3), Cin, 256, 64, 64, precision=prec, a_sexp=4, w_sexp=8)
import time
for _ in range(3):
    GM.gemm_tap(d, skip, wp, y)
torch.cuda.synchronize()
t0 = time.time()
for _ in range(5):
    GM.gemm_tap(d, skip, wp, y)
torch.cuda.synchronize()
dt = (time.time() - t0) / 5
if os.environ.get('CHECK') == '1':
    ref = torch.nn.functional.conv2d(torch.nn.functional.pad(skip[:1, :, :, :Cin].double().permute(0, 3, 1, 2), (1, 1, 8, 0)), w.double(), dilation=(8, 1)).permute(0, 2, 3, 1)
    err = (y[:1].double() - ref).abs().max().item()
    print('max err vs fp64', err, 'rel to max', err / ref.abs().max().item(), 'rms rel', ((y[:1].double() - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item())
print(f'conv Cin={Cin} precision={prec}: {dt*1e6:.0f} us, {2.0*B*T*Fq*64*6*Cin/dt/1e12:.1f} TF')
