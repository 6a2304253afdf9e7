// The fused feed-forward backward kernel (se_ff_fused.hip): argument block; transposed-read fragments and the fp16 split helper it
// shares with se_lnbwd_fused.hip.
#pragma once
#include "se_gemm_dev.h"

typedef short s16x4f_ __attribute__((ext_vector_type(4)));
typedef _Float16 h2f_ __attribute__((ext_vector_type(2)));

struct FfFusedArgs {
  const float* dY; const float* X; const float* stats; const float* gamma; const float* beta;
  const float* W1; const float* b1; const float* W2T;      // scaled fp16 planes [2][256][64]: W1, (alpha W2)^T
  const float* dR2; float* dX; float* dgamma; float* dbeta;
  float* dW1; float* db1; float* dW2; float* db2;           // [256][64], [256], [64][256], [64] (db2 may be NULL): accumulated
  long M; long rows_per_wg; float drop_p; unsigned seed_h, seed_o; float alpha;
  const float* dy_amax; const float* w1_amax; const float* w2t_amax; const float* in_amax; const float* mid_amax; float* out_amax;
  int ln_sexp, hid_sexp;
  int dbg;                 // (unused)
  unsigned* stamps;        // -DSE_FF_STAMPS builds only (tools/ff_fused_stamps.py): per-wave cycle accumulators, written once at the end
};

namespace fff {
static __device__ __forceinline__ u32x2_ tr8_(const unsigned char* p) {          // ds_read_b64_tr_b16 (EXEC must be full)
  return __builtin_bit_cast(u32x2_, __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4f_ __attribute__((address_space(3)))*)(p)));
}
// 8-deep fragment of a [rows = contraction index][cols] fp16 image with row stride STR: the lane gets column (lane & 31) of its block,
// contraction slots 8 kg .. 8 kg + 7 of the 16-deep step whose first row `p` already points at (p = this lane's tr address)
template <int STR>
static __device__ __forceinline__ bf16x8 trfrag_(const unsigned char* p) {
  const u32x2_ t0 = tr8_(p), t1 = tr8_(p + 4 * STR);
  return __builtin_bit_cast(bf16x8, (u32x4_){t0[0], t0[1], t1[0], t1[1]});
}
// the same fragment AND acc += the sum of its 8 fp16 values (v_dot2c_f32_f16 against (1, 1)).  The four words are taken from the
// two transposed reads BEFORE they are assembled into the fragment: read back out of the assembled ext-vector, hipcc 7.2 fed all
// four dot products from the fragment's FIRST register (the miscompile split_planes8_h works around; tools/micro/f16chk.hip)
template <int STR>
static __device__ __forceinline__ bf16x8 trfrag_sum_(const unsigned char* p, float& acc) {
  const u32x2_ t0 = tr8_(p), t1 = tr8_(p + 4 * STR);
  const unsigned a0 = t0[0], a1 = t0[1], a2 = t1[0], a3 = t1[1];
  const h2f_ one = {(_Float16)1.0f, (_Float16)1.0f};
  acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(h2f_, a0), one, acc, false);
  acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(h2f_, a1), one, acc, false);
  acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(h2f_, a2), one, acc, false);
  acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(h2f_, a3), one, acc, false);
  return __builtin_bit_cast(bf16x8, (u32x4_){a0, a1, a2, a3});
}
// (already scaled) y0..y3 -> packed fp16 hi words h0 h1 and lo words l0 l1 (scalar words: see split_planes8_h); lo = y * 1 - hi with an
// OPAQUE 1.0 so that the residual is ONE v_fma_mix_f32 per value (the plain form compiles to v_cvt_f32_f16 + v_sub_f32)
static __device__ __forceinline__ void split4_(float y0, float y1, float y2, float y3, float one, unsigned& h0, unsigned& h1, unsigned& l0,
                                               unsigned& l1) {
  h0 = pk_f16_(y0, y1); h1 = pk_f16_(y2, y3);
  const f16x2_ a = __builtin_bit_cast(f16x2_, h0), b = __builtin_bit_cast(f16x2_, h1);
  y0 = __builtin_fmaf(y0, one, -(float)a[0]); y1 = __builtin_fmaf(y1, one, -(float)a[1]);
  y2 = __builtin_fmaf(y2, one, -(float)b[0]); y3 = __builtin_fmaf(y3, one, -(float)b[1]);
  l0 = pk_f16_(y0, y1); l1 = pk_f16_(y2, y3);
}
}  // namespace fff

