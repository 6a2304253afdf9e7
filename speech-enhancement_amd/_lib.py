"""ctypes binding of libse_hip.so (include/se_hip.h).  There is NO fallback: if the library is
missing or a call fails, the product raises."""
import ctypes as C
import os

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
# SE_HIP_LIB: an alternative build of the SAME library for A/B timing on one box (tools/micro/run_ab.sh) -- the in-tree file is
# never overwritten by an experiment; unset = the product library
LIB_PATH = os.environ.get('SE_HIP_LIB') or os.path.join(HERE, 'libse_hip.so')
SE_MAX_TAPS = 16

PRO_NONE, PRO_LN, PRO_SWISH, PRO_AFFINE_SWISH, PRO_SWISH_DROP, PRO_DROP, PRO_GATE = 0, 1, 2, 3, 4, 5, 6
EPI_BIAS, EPI_ACCUM, EPI_RESID, EPI_GLU, EPI_STATS, EPI_SWISH_GRAD, EPI_SHUFFLE2, EPI_DROP = 1, 2, 4, 8, 16, 32, 64, 128
EPI_ROWSTATS = 512
EPI_GLU_GATE = 2048
EPI_DELTA = 4096


class GemmDesc(C.Structure):
    _fields_ = [('B', C.c_int), ('To', C.c_int), ('Fo', C.c_int), ('Ti', C.c_int), ('Fi', C.c_int),
                ('st', C.c_int), ('sf', C.c_int), ('up', C.c_int), ('ntap', C.c_int),
                ('dt', C.c_int * SE_MAX_TAPS), ('df', C.c_int * SE_MAX_TAPS),
                ('C', C.c_int), ('lda', C.c_int), ('a_off', C.c_int),
                ('N', C.c_int), ('ldc', C.c_int), ('c_off', C.c_int), ('ldw', C.c_int),
                ('prologue', C.c_int), ('epilogue', C.c_int), ('alpha', C.c_float),
                ('ldr', C.c_int), ('r_off', C.c_int), ('ldx', C.c_int), ('x_off', C.c_int),
                ('pro_seed', C.c_uint), ('epi_seed', C.c_uint), ('drop_p', C.c_float), ('precision', C.c_int),
                ('w_planes', C.c_int), ('a_sexp', C.c_int), ('w_sexp', C.c_int), ('a_amax', C.c_void_p),
                ('w_amax', C.c_void_p), ('y_amax', C.c_void_p)]


class F16Scales(C.Structure):
    """se_f16_scales (include/se_hip.h): operand scales of the token-wise scaled split-fp16 kernels (precision 3)"""
    _fields_ = [('in_amax', C.c_void_p), ('in_sexp', C.c_int), ('mid_sexp', C.c_int), ('wa_amax', C.c_void_p),
                ('wb_amax', C.c_void_p), ('out_amax', C.c_void_p), ('mid_amax', C.c_void_p)]


class SeHipError(RuntimeError):
    pass


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SeHipError(f'{LIB_PATH} is missing: run `python __graft_entry__.py` (build()) first; '
                             'there is no CPU / PyTorch fallback for the hot path')
        _lib = C.CDLL(LIB_PATH)
        _lib.se_last_error.restype = C.c_char_p
        for f in ('se_attn_bwd_workspace_bytes', 'se_norm_prelu_bwd_workspace_bytes', 'se_segnorm_workspace_bytes',
                  'se_dwconv31_wgrad_workspace_bytes', 'se_disc_tail_workspace_bytes'):
            getattr(_lib, f).restype = C.c_size_t
    return _lib


def ptr(t):
    """device pointer of a tensor (or NULL)."""
    if t is None:
        return C.c_void_p(0)
    if not t.is_cuda:
        raise SeHipError('hot-path tensors must live on the GPU: there is no CPU fallback')
    return C.c_void_p(t.data_ptr())


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class KernelTimer:
    """Optional per-launch timing with HIP events on the launch stream (used by bench.py for the roofline object:
    algorithmic FLOPs or bytes of each launch / its measured duration).  Off by default: zero overhead."""

    def __init__(self):
        self.active = False
        self.every = False
        self.records = []

    def start(self, every_call=False):
        """every_call: also time the launches without a roofline key, and keep the stream each one went to (tools/streams_timeline.py)"""
        self.records = []
        self.every = every_call
        self.active = True

    def stop(self):
        self.active = False

    def summary(self):
        """key -> dict(launches, ms, flops, bytes); call after torch.cuda.synchronize()."""
        out = {}
        for key, flops, nbytes, e0, e1, _ in self.records:
            d = out.setdefault(key, {'launches': 0, 'ms': 0.0, 'flops': 0.0, 'bytes': 0.0})
            d['launches'] += 1
            d['ms'] += e0.elapsed_time(e1)
            d['flops'] += flops
            d['bytes'] += nbytes
        return out


TIMER = KernelTimer()


def call(name, *args, _key=None, _flops=0.0, _bytes=0.0):
    fn = getattr(lib(), name)
    if TIMER.active and (_key is not None or TIMER.every):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = fn(*args)
        e1.record()
        if callable(_key):          # a key that depends on what the C side dispatched to (evaluated after the call)
            _key = _key()
        TIMER.records.append((_key or name, _flops, _bytes, e0, e1, torch.cuda.current_stream().cuda_stream))
    else:
        rc = fn(*args)
    if rc != 0:
        raise SeHipError(f'{name} failed ({rc}): {lib().se_last_error().decode()}')


def check_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise SeHipError('hot-path tensors must live on the GPU (no CPU fallback)')
