"""Per-step weight preparation in ONE launch (csrc/se_gemm.hip: weight_prep_kernel).

Every GEMM of the generator reads its weight as an [N][ld] matrix in a layout PyTorch does not store: conv weights tap-major
with the DilatedDenseNet slab order undone (models/generator.py:31), transposes for the input gradients, cat(to_q, to_kv),
Scale(0.5)'s factor folded in, and -- for the six-product split-bf16 kernels -- the exact hi / mid / lo bf16 planes.  Round 1
did this with ~170 small launches per step (repack kernels, .t().contiguous(), cat, mul) and re-split every weight tile in
every workgroup of every GEMM.  A WeightPlan records each prepared matrix once (parameter and arena addresses are stable),
uploads the item table, and `run()` refreshes all of them with a single kernel launch per forward.
"""
import ctypes as C

import torch

from . import _lib as L


class WItem(C.Structure):
    _fields_ = [('src', C.c_void_p), ('dst', C.c_void_p), ('No', C.c_int), ('Nt', C.c_int), ('Ni', C.c_int),
                ('Ni_dst', C.c_int), ('so', C.c_long), ('stt', C.c_long), ('si', C.c_long), ('rev', C.c_int),
                ('dst_ld', C.c_int), ('o_off', C.c_int), ('c_off', C.c_int), ('scale', C.c_float),
                ('plane_stride', C.c_long), ('fmt', C.c_int), ('amax', C.c_void_p)]


class BItem(C.Structure):        # se_bound_item (include/se_hip.h)
    _fields_ = [('g', C.c_void_p), ('b', C.c_void_p), ('alpha', C.c_void_p), ('W', C.c_void_p), ('wb', C.c_void_p),
                ('n', C.c_int), ('na', C.c_int), ('rows', C.c_int), ('cols', C.c_int), ('ksel', C.c_int),
                ('kconst', C.c_float), ('post', C.c_float), ('out', C.c_void_p)]


class WeightPlan:
    def __init__(self, device):
        self.device = device
        self.bounds = {}       # key -> device scalar: proven bound of a normalised activation (se_act_bounds), valid after run_bounds()
        self._bitems = []
        self._btable = None
        self._bscal = None
        self.bounds_ready = False
        self.k1 = self.k2 = 0.0     # the run-time constants the last run_bounds derived the ksel 1 / 2 bounds with
        self.out = {}          # key -> prepared tensor: fp32 [rows, ld], bf16 planes [3, rows, ld] or scaled fp16 planes [2, rows, ld]
        self._items = []
        self._srcs = []        # keeps the source tensors alive / lets `stale()` detect re-allocated parameters
        self._table = None
        self._max = 1
        self._amax = None      # one fp32 scalar per fp16-plane destination (se_gemm_desc.w_amax); zeroed before every run
        self._amax_slot = {}

    def add(self, key, src, No, Nt, Ni, so, stt, si, rev=0, scale=1.0, planes=False, rows=None, ld=None, o_off=0, c_off=0,
            Ni_dst=None):
        """dst[o_off + o][c_off + t * Ni_dst + i] = scale * src[o' * so + t * stt + i' * si]; the first `add` of a key
        allocates its (zero-filled, so padding stays zero) destination of `rows` x `ld`."""
        if not src.is_cuda or src.dtype != torch.float32 or not src.is_contiguous():
            raise L.SeHipError(f'weight plan: {key}: parameters must be contiguous fp32 CUDA tensors')
        Ni_dst = Ni_dst or Ni
        if key not in self.out:
            rows = rows or (o_off + No)
            ld = ld or (c_off + Nt * Ni_dst)
            if planes:
                if ld % 8:
                    raise L.SeHipError(f'weight plan: {key}: plane rows must be multiples of 8 elements (ld={ld})')
                if planes == 'f16':        # precision 3: two scaled fp16 planes + the amax scalar they were scaled by
                    # one scalar per 128-B line: 32 scalars in a line made the atomic maxima of 32 matrices contend for it
                    # (weight_amax_kernel: 238 us per step for 7 MB of weights)
                    if self._amax is None:
                        self._amax = torch.zeros(1024 * 32, device=self.device, dtype=torch.float32)
                    if len(self._amax_slot) >= self._amax.numel() // 32:
                        raise L.SeHipError('weight plan: more than 1024 fp16-plane matrices')
                    self._amax_slot[key] = 32 * len(self._amax_slot)
                    self.out[key] = torch.zeros(2, rows, ld, device=self.device, dtype=torch.float16)
                    self.out[key]._se_amax = self._amax[self._amax_slot[key]:self._amax_slot[key] + 1]
                else:
                    self.out[key] = torch.zeros(3, rows, ld, device=self.device, dtype=torch.bfloat16)
            else:
                self.out[key] = torch.zeros(rows, ld, device=self.device, dtype=torch.float32)
        dst = self.out[key]
        is_pl = dst.dtype in (torch.bfloat16, torch.float16)
        f16 = dst.dtype == torch.float16
        drows, dld = dst.shape[-2], dst.shape[-1]
        if o_off + No > drows or c_off + (Nt - 1) * Ni_dst + Ni > dld:
            raise L.SeHipError(f'weight plan: {key}: item exceeds its destination')
        self._items.append(WItem(src.data_ptr(), dst.data_ptr(), No, Nt, Ni, Ni_dst, so, stt, si, rev, dld, o_off, c_off,
                                 float(scale), drows * dld if is_pl else 0, 1 if f16 else 0,
                                 dst._se_amax.data_ptr() if f16 else None))
        self._srcs.append((src, src.data_ptr()))
        self._max = max(self._max, No * Nt * Ni)
        self._table = None
        return dst

    def bound(self, key, g, b=None, alpha=None, W=None, wb=None, ksel=0, kconst=7.9372539, post=1.0):
        """register a proven activation bound (k max|g| + max|b|) max(1, max|alpha|) [* max row-l1(W) + max|wb|] * post, stored in
        the scalar of `key`; k = kconst (LayerNorm(64): sqrt(63)) or the
        run-time k1 / k2 of run_bounds (ksel 1 / 2).  Returns the device scalar (valid after run_bounds)."""
        for t in (g, b, alpha, W, wb):
            if t is not None and (not t.is_cuda or t.dtype != torch.float32 or not t.is_contiguous()):
                raise L.SeHipError(f'weight plan: bound {key}: parameters must be contiguous fp32 CUDA tensors')
        if self._bscal is None:
            self._bscal = torch.zeros(256 * 32, device=self.device, dtype=torch.float32)     # one scalar per 128-B line
        if key in self.bounds:
            raise L.SeHipError(f'weight plan: bound {key} registered twice (one item per scalar)')
        if key not in self.bounds:
            if len(self.bounds) >= self._bscal.numel() // 32:
                raise L.SeHipError('weight plan: more than 256 activation bounds')
            i = 32 * len(self.bounds)
            self.bounds[key] = self._bscal[i:i + 1]
        pt = lambda t: t.data_ptr() if t is not None else None
        rows = W.shape[0] if W is not None else 0
        self._bitems.append(BItem(pt(g), pt(b), pt(alpha), pt(W), pt(wb), g.numel(), alpha.numel() if alpha is not None else 0, rows,
                                  W.numel() // rows if W is not None else 0, ksel, float(kconst), float(post),
                                  self.bounds[key].data_ptr()))
        for t in (g, b, alpha, W, wb):
            if t is not None:
                self._srcs.append((t, t.data_ptr()))
        self._btable = None
        return self.bounds[key]

    def run_bounds(self, k1=0.0, k2=0.0):
        """refresh every registered bound from the current parameter values (one launch)"""
        if not self._bitems:
            return
        if self._btable is None:
            arr = (BItem * len(self._bitems))(*self._bitems)
            self._btable = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(self.device)
        L.call('se_act_bounds', L.ptr(self._btable), C.c_int(len(self._bitems)), C.c_float(k1), C.c_float(k2), L.stream())
        self.bounds_ready = True
        self.k1, self.k2 = float(k1), float(k2)

    def stale(self):
        """True when a source parameter has been re-allocated since the plan was built (.to(), .data = ...)."""
        return any(t.data_ptr() != p for t, p in self._srcs)

    def run(self):
        self.bounds_ready = False           # (the bounds of the previous parameter values; run_bounds follows where they are used)
        if not self._items:
            return
        if self._table is None:
            arr = (WItem * len(self._items))(*self._items)
            self._table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(self.device)
        if self._amax is not None:
            self._amax.zero_()
        L.call('se_weight_prep', L.ptr(self._table), C.c_int(len(self._items)), C.c_long(self._max), L.stream())

    # ---- recipes -------------------------------------------------------------------------------------------------
    def conv_fwd(self, key, w, rev=False, C_pad=None, N_pad=None, planes=False):
        """PyTorch conv weight [N, Cin, kh, kw] -> [N (N_pad)][tap][Cin (C_pad)]."""
        N, Cin, kh, kw = w.shape
        Cd = C_pad or Cin
        return self.add(key, w, N, kh * kw, Cin, Cin * kh * kw, 1, kh * kw, rev=1 if rev else 0, planes=planes,
                        rows=N_pad or N, ld=kh * kw * Cd, Ni_dst=Cd)

    def conv_dgrad(self, key, w, rev=False, C_pad=None, N_pad=None, planes=False):
        """[N, Cin, kh, kw] -> [Cin (C_pad)][tap][N (N_pad)] (the caller negates the taps)."""
        N, Cin, kh, kw = w.shape
        Nd = N_pad or N
        return self.add(key, w, Cin, kh * kw, N, kh * kw, 1, Cin * kh * kw, rev=2 if rev else 0, planes=planes,
                        rows=C_pad or Cin, ld=kh * kw * Nd, Ni_dst=Nd)

    def linear(self, key, w, planes=False, scale=1.0, o_off=0, rows=None):
        N, K = w.shape[0], w.numel() // w.shape[0]
        return self.add(key, w, N, 1, K, K, 1, 1, planes=planes, scale=scale, o_off=o_off, rows=rows, ld=K)

    def linear_T(self, key, w, planes=False, scale=1.0, c_off=0, ld=None):
        N, K = w.shape[0], w.numel() // w.shape[0]
        return self.add(key, w, K, 1, N, 1, 1, K, planes=planes, scale=scale, c_off=c_off, ld=ld or N, rows=K)
