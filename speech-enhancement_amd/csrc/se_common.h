// Shared device/host helpers for libse_hip.so (gfx950 only: wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "../../include/se_hip.h"

// Packed fp32 VALU ops (v_pk_mul / add / fma_f32) for ONE kernel of a translation unit that is built without them (build.py: they
// run on the matrix pipe and stall the MFMAs of the SIMD): worth it where the vector unit, not the matrix pipe, is the busy one --
// measured per kernel (attention forward: 0.681 -> 0.651 ms).
// (tried on ff_fwd_kernel and ff_bwd_fused_kernel too: no gain, 57.16 vs 57.22 / 57.10 vs 57.03 ms per step same box.)
#ifdef __HIP_DEVICE_COMPILE__
#define SE_PACKED_FP32_KERNEL __attribute__((target("packed-fp32-ops")))
#else
#define SE_PACKED_FP32_KERNEL            /* the host pass does not know the feature name */
#endif
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

extern thread_local char g_se_err[512];
int se_fail(const char* fmt, ...);
int se_check_launch(const char* what);

#define SE_REQUIRE(cond, ...) do { if (!(cond)) return se_fail(__VA_ARGS__); } while (0)

// sigmoid on the hardware transcendental path: v_exp_f32 (exp2 of x*log2e) + v_rcp_f32, ~1 ulp each (relative error
// ~2e-7, far inside the parity budget); the libm expf + IEEE divide cost ~10x more VALU issue slots and made the
// Swish / GLU pro- and epilogues of the K=64 GEMMs VALU-bound.
static __device__ __forceinline__ float sigmoidf_(float x) {
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
static __device__ __forceinline__ float swishf_(float x) { return x * sigmoidf_(x); }
static __device__ __forceinline__ float swish_gradf_(float x) {
  float s = sigmoidf_(x);
  return s * (1.0f + x * (1.0f - s));
}

// xor-16 / xor-32 butterfly steps on the VALU: gfx950's v_permlane16_swap / v_permlane32_swap exchange the odd 16- (32-) lane rows of
// one register with the even rows of another -- applied to two copies of v, the two results hold (own row's value, partner row's
// value) in some order on every lane, so a COMMUTATIVE combine of them is the butterfly step.  __shfl_xor(v, 16 | 32) is a
// ds_bpermute_b32 through the LDS crossbar: ~100 cycles of latency each, and in the attention kernels they sit on the softmax chain
// (tile maximum -> rescale decision -> exp2).  Checked against __shfl_xor on the device: tools/micro/permlane_chk.hip.
typedef unsigned u32x2sw_ __attribute__((ext_vector_type(2)));
// (the maximum as inline assembly: fmaxf on bit-cast values makes the compiler canonicalise both inputs first -- two v_max x, x per step
// on the softmax chain of the attention forward)
static __device__ __forceinline__ float max_raw_(float a, float b) {
  float o;
  asm("v_max_f32 %0, %1, %2" : "=v"(o) : "v"(a), "v"(b));
  return o;
}
static __device__ __forceinline__ float xor32_max_(float v) {
  const u32x2sw_ r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return max_raw_(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
static __device__ __forceinline__ float xor16_max_(float v) {
  const u32x2sw_ r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return max_raw_(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
static __device__ __forceinline__ float xor32_sum_(float v) {
  const u32x2sw_ r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
static __device__ __forceinline__ float xor16_sum_(float v) {
  const u32x2sw_ r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// full-wave (64-lane) butterfly sum
static __device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
static __device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
static __device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// The dynamic-LDS limit is a per-DEVICE function attribute: remember it per (call site, device), not per process -- a process
// that drives a second GPU would otherwise launch there with the default 64 KB limit and fail.  `done` = one bit per device
// ordinal of the call site (devices >= 32 simply set the attribute every time; the call is idempotent and cheap).
static inline bool se_raise_lds(const void* fn, size_t bytes, unsigned* done) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return false;
  if (dev >= 0 && dev < 32 && ((*done >> dev) & 1u)) return true;
  if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) return false;
  if (dev >= 0 && dev < 32) *done |= 1u << dev;
  return true;
}

// Compute units of the CURRENT device, cached per device ordinal (the persistent kernels size their grid with it on every launch,
// also during graph capture: one runtime query per device and process, error-checked; unknown -> 256, never less than 8).
static inline int se_cu_count() {
  static int cache[32] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 256;
  if (dev >= 0 && dev < 32 && cache[dev] > 0) return cache[dev];
  int v = 0;
  if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
  if (v < 8) v = 8;
  if (dev >= 0 && dev < 32) cache[dev] = v;
  return v;
}

// Raise the device scalar *p (a running maximum of non-negative floats, ordered like their bit patterns) to m; called by ONE lane
// per wave with the wave's maximum.  The plain load first: tens of thousands of waves hammering one address with atomics are
// serialised in the L2 (259 K atomics of a glu_bwd launch cost milliseconds); after the first few waves almost every wave sees
// a stored maximum >= its own and issues nothing.  (A stale read only costs a redundant atomic.)
static __device__ __forceinline__ void amax_raise_(float* p, float m) {
  if (!(m > 0.f)) return;
  const unsigned mb = __float_as_uint(m);
  if (__hip_atomic_load(reinterpret_cast<unsigned*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= mb) return;
  atomicMax(reinterpret_cast<unsigned*>(p), mb);
}

// 16-byte store of a feature-map element that is written once and read by a LATER kernel.  Non-temporal stores (SE_NT_STORE
// builds) were measured on the normalisation / GLU kernels: no gain on the step (76.7 vs 76.3 - 76.7 ms, two A/B pairs), unlike the
// depthwise conv where they win 4 - 9 % -- plain stores stay the default here.
static __device__ __forceinline__ void st4_stream_(float* p, float4 v) {
#ifdef SE_NT_STORE
  __builtin_nontemporal_store(__builtin_bit_cast(f32x4, v), reinterpret_cast<f32x4*>(p));
#else
  *reinterpret_cast<float4*>(p) = v;
#endif
}

// Buffer (SRSRC) loads with the hardware range check: a lane whose byte offset is >= `bytes` gets zeros.  The tile loaders use
// them instead of `ok ? load : 0`: predicated flat loads compile to divergent branches, after which the compiler can no
// longer count the loads in flight and falls back to s_waitcnt vmcnt(0) at the first use of ANY of them (the prefetch of the
// next tile then stalls the current one).  The descriptor is built from wave-uniform values only (readfirstlane'd).
static __device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc_(const void* p, unsigned bytes) {
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(size_t)p);
  const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)((size_t)p >> 32));
  return __builtin_amdgcn_make_buffer_rsrc((void*)(((size_t)hi << 32) | lo), 0, __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}
static __device__ __forceinline__ float4 buf_load4_(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
  typedef unsigned u32x4b_ __attribute__((ext_vector_type(4)));
  return __builtin_bit_cast(float4, (u32x4b_)__builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 0));
}
static __device__ __forceinline__ float2 buf_load2_(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
  typedef unsigned u32x2b_ __attribute__((ext_vector_type(2)));
  return __builtin_bit_cast(float2, (u32x2b_)__builtin_amdgcn_raw_buffer_load_b64(r, byte_off, 0, 0));
}
static __device__ __forceinline__ void buf_store4_(__amdgpu_buffer_rsrc_t r, unsigned byte_off, float4 v) {
  typedef unsigned u32x4b_ __attribute__((ext_vector_type(4)));
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4b_, v), r, byte_off, 0, 0);
}
constexpr unsigned BUF_OOB_ = 0xfffffff0u;      // a byte offset no descriptor of ours covers
static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }
static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
