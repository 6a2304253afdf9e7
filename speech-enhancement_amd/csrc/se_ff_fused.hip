// Fused backward of the Conformer feed-forward module, weight gradients included (round 5).
//
//   Scale(0.5, PreNorm(FeedForward)) backward (conformer.py:53-71, 128-145) in ONE persistent launch that reads X, dY (, dR2) and
//   writes dX: the hidden pre-activations H are recomputed (the forward stores none), the hidden gradient dZ never leaves the CU,
//   and dW1 / db1 / dW2 / db2 accumulate in registers over the rows a workgroup sweeps and leave once through fp32 atomics.
//   Replaces se_ff_bwd_dgrad (1.68 GB per module at 518 736 rows) + two whole-gradient launches (0.67 + 0.70 GB) by 0.53 GB.
//
// One 8-wave workgroup per CU, rows in tiles of 64, hidden units in four blocks of 64 ("slots"), scaled split-fp16 arithmetic
// (se_gemm_dev.h, precision 3).  The waves are SPECIALISED; waves w and w + 4 share a SIMD (MI355X_MICROARCH.md):
//   D waves 0..3 = (row group g = w & 1, hidden half jh = w >> 1): per slot the 32 rows x 32 hidden units tile of
//       H^T = W1 LN(X)^T + b1 and dP^T = W2s (mask_o dY)^T           (A = weight rows from LDS, B = the rows' fragments in registers:
//       the C layout then has the ROW on the lane and four consecutive hidden units per register quad -- one dropout hash per quad)
//       S = Swish(H) mask_h,  dZ = dP mask_h Swish'(H)               (registers; split to fp16 (hi, lo); written ROW-major [r][j]
//       into the exchange images; dZ's registers ARE the A fragments of)
//       dLN += dZ W1                                                 (B = W1 rows through transposed reads of the same W1 image)
//     and per tile the LayerNorm prologue / dropout mask of dY (into row-major fp16 images) and the LayerNorm backward.
//   W waves 4..7 = (kind = dW1 | dW2, hidden half jh): per slot two 32 x 32 tiles of
//       dW1[j][c] += sum_r dZ[r][j] LN(X)[r][c]      or      dW2[c][j] += sum_r (mask_o dY)[r][c] S[r][j]
//     with BOTH operands taken from the row-major images by hardware-transposed reads (ds_read_b64_tr_b16: the contraction index is
//     the image row); 128 accumulator registers per wave for the four slots; db1 / db2 by packed dot products on the same fragments.
//   W lags D by one slot (the exchange images are double-buffered); it also streams the next slot's weight block L2 -> registers ->
//   LDS.  Two barriers per slot: | D: H, dP, S, dZ -> images; W1 fragments for dLN -> registers || W: first half of the previous
//   slot's tiles; next weight block -> registers | D: dLN MFMAs from registers || W: weight block -> LDS; second half |.
//   Every wave executes the same number of barriers (D and W run different code paths: s_barrier counts arrivals).
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
};

namespace fff {
constexpr int RS = 144;                 // image row stride in bytes: 64 fp16 + 16 B pad (ds_read_b128 rows conflict-free)
constexpr int PL = 64 * RS;             // plane stride (hi | lo)
constexpr int IMG = 2 * PL;             // one [64][64] fp16 (hi, lo) image
constexpr int O_LN = 0, O_DY = IMG, O_W1 = 2 * IMG, O_W2 = 3 * IMG, O_ZS = 4 * IMG;     // ZS: [buffer][Z | S] images
constexpr int O_B1 = O_ZS + 4 * IMG, O_GB = O_B1 + 1024, LDS_BYTES = O_GB + 512;
constexpr int PS = 68;                  // epilogue patch row stride (floats)
static_assert(4 * 32 * PS * 4 <= 2 * IMG, "the four dLN patches live in exchange buffer 0");

static __device__ __forceinline__ u32x2_ tr8_(const unsigned char* p) {          // ds_read_b64_tr_b16 (EXEC must be full)
  return __builtin_bit_cast(u32x2_, __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4f_ __attribute__((address_space(3)))*)(p)));
}
// 8-deep fragment of a [rows = contraction index][cols] fp16 image: the lane gets column (lane & 31) of its block, contraction slots
// 8 kg .. 8 kg + 7 of the 16-deep step whose first row `p` already points at (p = this lane's tr address, see tr_base)
// DR: image rows between the fragment's elements 0..3 and 4..7 (4: natural order; 8: the order of an accumulator's register quads)
template <int DR = 4>
static __device__ __forceinline__ bf16x8 trfrag_(const unsigned char* p) {
  const u32x2_ t0 = tr8_(p), t1 = tr8_(p + DR * RS);
  return __builtin_bit_cast(bf16x8, (u32x4_){t0[0], t0[1], t1[0], t1[1]});
}
// (already scaled) y0..y3 -> packed fp16 hi words h0 h1 and lo words l0 l1 (scalar words: see split_planes8_h)
static __device__ __forceinline__ void split4_(float y0, float y1, float y2, float y3, unsigned& h0, unsigned& h1, unsigned& l0, unsigned& l1) {
  h0 = pk_f16_(y0, y1); h1 = pk_f16_(y2, y3);
  const f16x2_ a = __builtin_bit_cast(f16x2_, h0), b = __builtin_bit_cast(f16x2_, h1);
  l0 = pk_f16_(y0 - (float)a[0], y1 - (float)a[1]);
  l1 = pk_f16_(y2 - (float)b[0], y3 - (float)b[1]);
}
// the same fragment AND acc += the sum of its 8 fp16 values (v_dot2c_f32_f16 against (1, 1)).  The four words are taken from the
// two transposed reads BEFORE they are assembled into the fragment: read back out of the assembled ext-vector, hipcc 7.2 fed all
// four dot products from the fragment's FIRST register (the miscompile split_planes8_h works around; tools/micro/f16chk.hip)
template <int DR = 4>
static __device__ __forceinline__ bf16x8 trfrag_sum_(const unsigned char* p, float& acc) {
  const u32x2_ t0 = tr8_(p), t1 = tr8_(p + DR * RS);
  const unsigned a0 = t0[0], a1 = t0[1], a2 = t1[0], a3 = t1[1];
  const h2f_ one = {(_Float16)1.0f, (_Float16)1.0f};
  acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(h2f_, a0), one, acc, false);
  acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(h2f_, a1), one, acc, false);
  acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(h2f_, a2), one, acc, false);
  acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(h2f_, a3), one, acc, false);
  return __builtin_bit_cast(bf16x8, (u32x4_){a0, a1, a2, a3});
}
struct Scales { float s_in, s_dy, s_s, s_z, uh, u1, u2, un1, un2, ub1, ub2; };
}  // namespace fff

// ------------------------------------------------------------------------------------------------ D waves
static __device__ __forceinline__ void ff_fused_D(const FfFusedArgs& a, unsigned char* sm, const int wave, const int lane,
                                                  const long mbeg, const long mend, const int ntile, const fff::Scales& sc) {
  using namespace fff;
  const int g = wave & 1, jh = wave >> 1;
  const int r = lane & 31, kg = lane >> 5;
  const int gi = lane >> 4, q4 = (lane & 15) >> 2, p4 = lane & 3;
  const unsigned thr = drop_thr(a.drop_p);
  const float inv_keep = drop_inv_keep(a.drop_p);
  const bool dr = a.drop_p > 0.f;
  const float* b1s = reinterpret_cast<const float*>(sm + O_B1);
  const float* gbs = reinterpret_cast<const float*>(sm + O_GB);
  unsigned char* const myimg = sm + (jh == 0 ? O_LN : O_DY);
  // prologue / epilogue roles
  const int prr = lane >> 3, pcq = lane & 7;
  float4 raw[4][2];
  float2 rst[4];
  const float* const psrc = jh == 0 ? a.X : a.dY;
  auto load_raw = [&](long m0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      long m = m0 + 32 * g + prr + 8 * i;
      if (m > a.M - 1) m = a.M - 1;                      // unconditional loads; rows >= mend are zeroed below
      raw[i][0] = *reinterpret_cast<const float4*>(psrc + m * 64 + 8 * pcq);
      raw[i][1] = *reinterpret_cast<const float4*>(psrc + m * 64 + 8 * pcq + 4);
      rst[i] = *reinterpret_cast<const float2*>(a.stats + 2 * m);
    }
  };
  auto prologue_store = [&](long m0) {
    const float4 gm0 = *reinterpret_cast<const float4*>(gbs + 8 * pcq), gm1 = *reinterpret_cast<const float4*>(gbs + 8 * pcq + 4);
    const float4 bt0 = *reinterpret_cast<const float4*>(gbs + 64 + 8 * pcq), bt1 = *reinterpret_cast<const float4*>(gbs + 64 + 8 * pcq + 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const long m = m0 + 32 * g + prr + 8 * i;
      const bool ok = m < mend;
      float x[8];
      const float4 w0 = raw[i][0], w1 = raw[i][1];
      if (jh == 0) {
        const float mean = rst[i].x, rstd = rst[i].y;
        x[0] = (w0.x - mean) * rstd * gm0.x + bt0.x; x[1] = (w0.y - mean) * rstd * gm0.y + bt0.y;
        x[2] = (w0.z - mean) * rstd * gm0.z + bt0.z; x[3] = (w0.w - mean) * rstd * gm0.w + bt0.w;
        x[4] = (w1.x - mean) * rstd * gm1.x + bt1.x; x[5] = (w1.y - mean) * rstd * gm1.y + bt1.y;
        x[6] = (w1.z - mean) * rstd * gm1.z + bt1.z; x[7] = (w1.w - mean) * rstd * gm1.w + bt1.w;
      } else {
        float4 d0 = make_float4(1.f, 1.f, 1.f, 1.f), d1 = d0;
        if (dr) {
          d0 = drop_scale4(a.seed_o, (unsigned)(m * 64 + 8 * pcq), thr, inv_keep);
          d1 = drop_scale4(a.seed_o, (unsigned)(m * 64 + 8 * pcq + 4), thr, inv_keep);
        }
        x[0] = w0.x * d0.x; x[1] = w0.y * d0.y; x[2] = w0.z * d0.z; x[3] = w0.w * d0.w;
        x[4] = w1.x * d1.x; x[5] = w1.y * d1.y; x[6] = w1.z * d1.z; x[7] = w1.w * d1.w;
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) x[e] = ok ? x[e] : 0.f;
      bf16x8 o[2];
      split_planes8_h(x, jh == 0 ? sc.s_in : sc.s_dy, o);
      unsigned char* p = myimg + (32 * g + prr + 8 * i) * RS + 16 * pcq;
      *reinterpret_cast<bf16x8*>(p) = o[0];
      *reinterpret_cast<bf16x8*>(p + PL) = o[1];
    }
  };
  // per-lane LDS bases
  const unsigned char* const fragLN = sm + O_LN + (32 * g + r) * RS + 16 * kg;          // + pl * PL + 32 * ks
  const unsigned char* const fragDY = sm + O_DY + (32 * g + r) * RS + 16 * kg;
  const unsigned char* const w1a = sm + O_W1 + (32 * jh + r) * RS + 16 * kg;            // A fragments: weight row 32 jh + (lane & 31)
  const unsigned char* const w2a = sm + O_W2 + (32 * jh + r) * RS + 16 * kg;
  // transposed reads of the W1 image for dLN's B operand: contraction slots 8 kg .. of step ks' = hidden units
  // 32 jh + 16 ks' + {4 kg + 0..3, 8 + 4 kg + 0..3} (the order dZ's accumulator registers come in), column = channel 32 nt + (lane & 31)
  const unsigned char* const w1t = sm + O_W1 + (32 * jh + 4 * (gi >> 1) + q4) * RS + (16 * (gi & 1) + 4 * p4) * 2;
  const int zcol = (32 * jh + 4 * kg) * 2;                                               // + 16 q bytes: this lane's quad q of its row
  float* const patch = reinterpret_cast<float*>(sm + O_ZS);
  // epilogue role: rows 16 jh + err + 8 i (i = 0, 1) of group g, channels 8 ecq .. + 7
  const int err = lane >> 3, ecq = lane & 7;
  // gamma / beta gradients of channel 8 ecq + err (two registers per lane: the tile's per-lane sums are folded over the 8 row lanes at
  // the end of every epilogue and lane (err, ecq) keeps the total of ITS channel -- 16 persistent accumulators per lane did not fit)
  float agk = 0.f, abk = 0.f;
  float xmax = 0.f;
  bf16x8 lnf[4][2], dyf[4][2];
  f32x16 g0, g1;

  load_raw(mbeg);
  __syncthreads();                                       // (0) b1 / gamma / beta staged, weight block 0 in LDS
  for (int t = 0; t < ntile; ++t) {
    const long m0 = mbeg + 64L * t;
    prologue_store(m0);
    __syncthreads();                                     // (p) the tile's LN(X) / mask_o dY images are complete
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) {
        lnf[ks][pl] = *reinterpret_cast<const bf16x8*>(fragLN + pl * PL + 32 * ks);
        dyf[ks][pl] = *reinterpret_cast<const bf16x8*>(fragDY + pl * PL + 32 * ks);
      }
#pragma unroll
    for (int e = 0; e < 16; ++e) { g0[e] = 0.f; g1[e] = 0.f; }
    const long m = m0 + 32 * g + r;                      // this lane's row in the C layout of H^T / dP^T
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      // ================= A(s) =================
      f32x16 ah, ad;
#pragma unroll
      for (int e = 0; e < 16; ++e) { ah[e] = 0.f; ad[e] = 0.f; }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const bf16x8 a1h = *reinterpret_cast<const bf16x8*>(w1a + 32 * ks), a1l = *reinterpret_cast<const bf16x8*>(w1a + PL + 32 * ks);
        const bf16x8 a2h = *reinterpret_cast<const bf16x8*>(w2a + 32 * ks), a2l = *reinterpret_cast<const bf16x8*>(w2a + PL + 32 * ks);
        ah = mfma32_<true>(a1h, lnf[ks][1], ah); ad = mfma32_<true>(a2h, dyf[ks][1], ad);
        ah = mfma32_<true>(a1l, lnf[ks][0], ah); ad = mfma32_<true>(a2l, dyf[ks][0], ad);
        ah = mfma32_<true>(a1h, lnf[ks][0], ah); ad = mfma32_<true>(a2h, dyf[ks][0], ad);
      }
      unsigned zh[8], zl[8];                              // dZ of this lane's row, hidden units (quad q): words 2q, 2q + 1
      unsigned char* const zrow = sm + O_ZS + (s & 1) * 2 * IMG + (32 * g + r) * RS + zcol;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int jq = 64 * s + 32 * jh + 8 * q + 4 * kg;       // first of the quad's four consecutive hidden units
        const float4 b4 = *reinterpret_cast<const float4*>(b1s + jq);
        float4 mk = make_float4(1.f, 1.f, 1.f, 1.f);
        if (dr) mk = drop_scale4(a.seed_h, (unsigned)(m * 256 + jq), thr, inv_keep);
        const float bb[4] = {b4.x, b4.y, b4.z, b4.w}, mm[4] = {mk.x, mk.y, mk.z, mk.w};
        float sv[4], zv[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int i = 4 * q + e;
          const float h = fmaf(ah[i], sc.uh, bb[e]);
          const float sg = sigmoidf_(h);
          const float s0 = h * sg;
          const float sw = fmaf(s0, 1.0f - sg, sg);             // Swish'(h) = sg (1 + h (1 - sg))
          sv[e] = s0 * (mm[e] * sc.s_s);
          zv[e] = ad[i] * sw * (mm[e] * sc.u1);                 // (u1 carries the un-scaling of the accumulator AND dZ's scale)
        }
        unsigned sh0, sh1, sl0, sl1;
        split4_(sv[0], sv[1], sv[2], sv[3], sh0, sh1, sl0, sl1);
        split4_(zv[0], zv[1], zv[2], zv[3], zh[2 * q], zh[2 * q + 1], zl[2 * q], zl[2 * q + 1]);
        *reinterpret_cast<u32x2_*>(zrow + 16 * q) = (u32x2_){zh[2 * q], zh[2 * q + 1]};
        *reinterpret_cast<u32x2_*>(zrow + PL + 16 * q) = (u32x2_){zl[2 * q], zl[2 * q + 1]};
        *reinterpret_cast<u32x2_*>(zrow + IMG + 16 * q) = (u32x2_){sh0, sh1};
        *reinterpret_cast<u32x2_*>(zrow + IMG + PL + 16 * q) = (u32x2_){sl0, sl1};
      }
      // W1 fragments of the dLN product out of the image BEFORE the barrier (the W waves overwrite the weight images after it)
      bf16x8 wb[2][2][2];
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
          for (int pl = 0; pl < 2; ++pl) wb[nt][k2][pl] = trfrag_<8>(w1t + pl * PL + 16 * k2 * RS + 64 * nt);
      __syncthreads();                                   // (a)
      // ================= B(s) =================
      if (s == 3 && t + 1 < ntile) load_raw(m0 + 64);     // the next tile's rows: in flight during dLN, the patch exchange, the epilogue
#pragma unroll
      for (int k2 = 0; k2 < 2; ++k2) {
        const bf16x8 zhf = __builtin_bit_cast(bf16x8, (u32x4_){zh[4 * k2], zh[4 * k2 + 1], zh[4 * k2 + 2], zh[4 * k2 + 3]});
        const bf16x8 zlf = __builtin_bit_cast(bf16x8, (u32x4_){zl[4 * k2], zl[4 * k2 + 1], zl[4 * k2 + 2], zl[4 * k2 + 3]});
        g0 = mfma32_<true>(zhf, wb[0][k2][1], g0); g1 = mfma32_<true>(zhf, wb[1][k2][1], g1);
        g0 = mfma32_<true>(zlf, wb[0][k2][0], g0); g1 = mfma32_<true>(zlf, wb[1][k2][0], g1);
        g0 = mfma32_<true>(zhf, wb[0][k2][0], g0); g1 = mfma32_<true>(zhf, wb[1][k2][0], g1);
      }
      __syncthreads();                                   // (b)
    }
    // ================= A(4): this wave's partial dLN (its 32 hidden units of every slot) -> patch; epilogue operands requested ====
    {
      float* P = patch + wave * (32 * PS);
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = (e & 3) + 8 * (e >> 2) + 4 * kg;
        P[row * PS + r] = g0[e] * sc.u2;
        P[row * PS + 32 + r] = g1[e] * sc.u2;
      }
    }
    float4 ex[2][2], ey[2][2], er[2][2];
    float2 est[2];
    const long rows_ok = mend - m0 < 64 ? mend - m0 : 64;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      long mr = m0 + 32 * g + 16 * jh + err + 8 * i;
      if (mr > a.M - 1) mr = a.M - 1;
      est[i] = *reinterpret_cast<const float2*>(a.stats + 2 * mr);
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const long off = mr * 64 + 8 * ecq + 4 * k;
        ex[i][k] = *reinterpret_cast<const float4*>(a.X + off);
        ey[i][k] = *reinterpret_cast<const float4*>(a.dY + off);
        er[i][k] = a.dR2 ? *reinterpret_cast<const float4*>(a.dR2 + off) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    __syncthreads();                                     // (c) both halves' patches written
    // ================= B(4): LayerNorm backward on rows 16 jh + err + 8 i of group g =================
    {
      const float* P0 = patch + g * (32 * PS), * P1 = patch + (g + 2) * (32 * PS);
      const float4 gmA = *reinterpret_cast<const float4*>(gbs + 8 * ecq), gmB = *reinterpret_cast<const float4*>(gbs + 8 * ecq + 4);
      const float gl[8] = {gmA.x, gmA.y, gmA.z, gmA.w, gmB.x, gmB.y, gmB.z, gmB.w};
      const __amdgpu_buffer_rsrc_t Xrs = make_rsrc_(a.dX + m0 * 64, (unsigned)(rows_ok * 64 * 4));
      float ag[8], ab[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) { ag[e] = 0.f; ab[e] = 0.f; }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int rl = 16 * jh + err + 8 * i;             // row inside the group
        const long mr = m0 + 32 * g + rl;
        const bool ok = mr < mend;
        const float4 pa0 = *reinterpret_cast<const float4*>(P0 + rl * PS + 8 * ecq), pb0 = *reinterpret_cast<const float4*>(P1 + rl * PS + 8 * ecq);
        const float4 pa1 = *reinterpret_cast<const float4*>(P0 + rl * PS + 8 * ecq + 4), pb1 = *reinterpret_cast<const float4*>(P1 + rl * PS + 8 * ecq + 4);
        const float dv[8] = {pa0.x + pb0.x, pa0.y + pb0.y, pa0.z + pb0.z, pa0.w + pb0.w, pa1.x + pb1.x, pa1.y + pb1.y, pa1.z + pb1.z, pa1.w + pb1.w};
        const float xs[8] = {ex[i][0].x, ex[i][0].y, ex[i][0].z, ex[i][0].w, ex[i][1].x, ex[i][1].y, ex[i][1].z, ex[i][1].w};
        const float mean = est[i].x, rstd = est[i].y;
        float xh[8], dxh[8], s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          xh[e] = (xs[e] - mean) * rstd;
          dxh[e] = dv[e] * gl[e];
          s1 += dxh[e]; s2 += dxh[e] * xh[e];
          if (ok) { ag[e] += dv[e] * xh[e]; ab[e] += dv[e]; }
        }
#pragma unroll
        for (int o = 1; o < 8; o <<= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
        s1 *= (1.f / 64.f); s2 *= (1.f / 64.f);
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const float4 r1 = ey[i][k], r2 = er[i][k];
          float o4[4] = {r1.x + r2.x, r1.y + r2.y, r1.z + r2.z, r1.w + r2.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) o4[e] += rstd * (dxh[4 * k + e] - s1 - xh[4 * k + e] * s2);
          buf_store4_(Xrs, (unsigned)(((32 * g + rl) * 64 + 8 * ecq + 4 * k) * 4), make_float4(o4[0], o4[1], o4[2], o4[3]));
          if (ok) xmax = fmaxf(fmaxf(xmax, fmaxf(fabsf(o4[0]), fabsf(o4[1]))), fmaxf(fabsf(o4[2]), fabsf(o4[3])));
        }
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float sg = ag[e], sb = ab[e];
#pragma unroll
        for (int o = 8; o < 64; o <<= 1) { sg += __shfl_xor(sg, o, 64); sb += __shfl_xor(sb, o, 64); }
        agk += err == e ? sg : 0.f;
        abk += err == e ? sb : 0.f;
      }
    }
    __syncthreads();                                     // (d) end of the tile: row images, exchange buffers and patches are free
  }
  if (a.out_amax) {
    xmax = wave_max(xmax);
    if (lane == 0) amax_raise_(a.out_amax, xmax);
  }
  // gamma / beta gradients: one atomic per channel and wave (lane (err, ecq) holds channel 8 ecq + err)
  atomicAdd(&a.dgamma[8 * ecq + err], agk);
  atomicAdd(&a.dbeta[8 * ecq + err], abk);
}

// ------------------------------------------------------------------------------------------------ W waves
static __device__ __forceinline__ void ff_fused_W(const FfFusedArgs& a, unsigned char* sm, const int wave, const int lane,
                                                  const int ntile, const fff::Scales& sc) {
  using namespace fff;
  const int w4 = wave - 4, kind = w4 >> 1, jh = w4 & 1;        // kind 0: dW1 (Z, LN), kind 1: dW2 (dY, S)
  const int gi = lane >> 4, q4 = (lane & 15) >> 2, p4 = lane & 3;
  const int tidw = w4 * 64 + lane;
  // transposed-read address of this lane inside a [64 rows = contraction][64 cols] image: + pl * PL + 16 ks * RS + 2 * col0
  const int trb = (8 * (gi >> 1) + q4) * RS + (16 * (gi & 1) + 4 * p4) * 2;
  const unsigned char* const shb = sm + O_ZS + (kind == 0 ? 0 : IMG) + trb + 64 * jh;      // shared operand: Z (dW1) or S (dW2), columns 32 jh ..
  const unsigned char* const vab = sm + (kind == 0 ? O_LN : O_DY) + trb;                  // varying operand: LN (dW1) or dY (dW2), + 64 nt
  f32x16 acc[4][2];
#pragma unroll
  for (int jb = 0; jb < 4; ++jb)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[jb][nt][e] = 0.f;
  float zs[4] = {0.f, 0.f, 0.f, 0.f}, ys[2] = {0.f, 0.f};
  float4 stg[8];
  // (buffer loads: one wave-uniform descriptor per matrix + ONE 32-bit lane offset + immediates -- with flat 64-bit addresses the
  // compiler kept 14 precomputed address pairs alive and spilled them)
  const __amdgpu_buffer_rsrc_t W1rs = make_rsrc_(a.W1, 2u * 256u * 64u * 2u), W2rs = make_rsrc_(a.W2T, 2u * 256u * 64u * 2u);
  const unsigned stoff = (unsigned)((tidw >> 3) * 128 + (tidw & 7) * 16);               // row (tidw >> 3) of the block, 16-byte chunk tidw & 7
  auto stage_load = [&](int jb) {
#pragma unroll
    for (int i = 0; i < 8; ++i)          // i = matrix * 4 + plane * 2 + row half
      stg[i] = buf_load4_((i >> 2) ? W2rs : W1rs, stoff + (unsigned)(((i >> 1) & 1) * 256 * 128 + (64 * jb + 32 * (i & 1)) * 128));
  };
  auto stage_store = [&]() {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int row = (tidw + 256 * (i & 1)) >> 3, ch = tidw & 7;
      *reinterpret_cast<float4*>(sm + ((i >> 2) ? O_W2 : O_W1) + ((i >> 1) & 1) * PL + row * RS + 16 * ch) = stg[i];
    }
  };
  // two 16-deep steps (ks0, ks0 + 1) of the tiles of slot jb out of exchange buffer (jb & 1)
  auto wgrad_half = [&](auto jbc, int ks0) {
    constexpr int jb = decltype(jbc)::value;
    const unsigned char* sh = shb + (jb & 1) * 2 * IMG;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int ks = ks0 + kk;
      float zsum = 0.f;                   // (kind 0: this lane's part of db1 = sum over the rows of dZ)
      const bf16x8 sh_h = trfrag_sum_(sh + 16 * ks * RS, zsum), sh_l = trfrag_sum_(sh + PL + 16 * ks * RS, zsum);
      if (kind == 0) zs[jb] += zsum;
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        float ysum = 0.f;                 // (kind 1, first slot: db2 = column sums of mask_o dY)
        const bf16x8 va_h = trfrag_sum_(vab + 16 * ks * RS + 64 * nt, ysum), va_l = trfrag_sum_(vab + PL + 16 * ks * RS + 64 * nt, ysum);
        if (kind == 1 && jh == 0 && jb == 0) ys[nt] += ysum;
        if (kind == 0) {          // dW1[j][c]: A = Z^T (row j on the lane), B = LN (column c on the lane)
          acc[jb][nt] = mfma32_<true>(sh_h, va_l, acc[jb][nt]);
          acc[jb][nt] = mfma32_<true>(sh_l, va_h, acc[jb][nt]);
          acc[jb][nt] = mfma32_<true>(sh_h, va_h, acc[jb][nt]);
        } else {                  // dW2[c][j]: A = dY^T (row c), B = S (column j)
          acc[jb][nt] = mfma32_<true>(va_h, sh_l, acc[jb][nt]);
          acc[jb][nt] = mfma32_<true>(va_l, sh_h, acc[jb][nt]);
          acc[jb][nt] = mfma32_<true>(va_h, sh_h, acc[jb][nt]);
        }
      }
    }
  };
  stage_load(0);
  stage_store();
  __syncthreads();                                       // (0)
  for (int t = 0; t < ntile; ++t) {
    __syncthreads();                                     // (p)
    // slot 0: nothing to contract yet
    stage_load(1);
    __syncthreads();                                     // (a0)
    stage_store();
    __syncthreads();                                     // (b0)
    stage_load(2);
    wgrad_half(std::integral_constant<int, 0>{}, 0);
    __syncthreads();                                     // (a1)
    stage_store();
    wgrad_half(std::integral_constant<int, 0>{}, 2);
    __syncthreads();                                     // (b1)
    stage_load(3);
    wgrad_half(std::integral_constant<int, 1>{}, 0);
    __syncthreads();                                     // (a2)
    stage_store();
    wgrad_half(std::integral_constant<int, 1>{}, 2);
    __syncthreads();                                     // (b2)
    stage_load(0);
    wgrad_half(std::integral_constant<int, 2>{}, 0);
    __syncthreads();                                     // (a3)
    stage_store();
    wgrad_half(std::integral_constant<int, 2>{}, 2);
    __syncthreads();                                     // (b3)
    wgrad_half(std::integral_constant<int, 3>{}, 0);
    __syncthreads();                                     // (c)
    wgrad_half(std::integral_constant<int, 3>{}, 2);
    __syncthreads();                                     // (d)
  }
  // ---- flush: C layout [row = (e & 3) + 8 (e >> 2) + 4 kg][col = lane & 31] ----
  const int col = lane & 31, kg = lane >> 5;
#pragma unroll
  for (int jb = 0; jb < 4; ++jb)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = (e & 3) + 8 * (e >> 2) + 4 * kg;
        if (kind == 0) atomicAdd(&a.dW1[(64 * jb + 32 * jh + row) * 64 + 32 * nt + col], acc[jb][nt][e] * sc.un1);
        else atomicAdd(&a.dW2[(32 * nt + row) * 256 + 64 * jb + 32 * jh + col], acc[jb][nt][e] * sc.un2);
      }
  if (kind == 0) {
#pragma unroll
    for (int jb = 0; jb < 4; ++jb) {
      const float v = zs[jb] + __shfl_xor(zs[jb], 32, 64);
      if (kg == 0) atomicAdd(&a.db1[64 * jb + 32 * jh + col], v * sc.ub1);
    }
  } else if (jh == 0 && a.db2) {
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const float v = ys[nt] + __shfl_xor(ys[nt], 32, 64);
      if (kg == 0) atomicAdd(&a.db2[32 * nt + col], v * sc.ub2);
    }
  }
}

__global__ __launch_bounds__(512, 2) void ff_bwd_fused_kernel(FfFusedArgs a) {
  using namespace fff;
  __shared__ __attribute__((aligned(16))) unsigned char sm[LDS_BYTES];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const long mbeg = (long)blockIdx.x * a.rows_per_wg;
  long mend = mbeg + a.rows_per_wg;
  if (mend > a.M) mend = a.M;
  if (mbeg >= mend) return;                              // (whole workgroup: block-uniform)
  const int ntile = (int)((mend - mbeg + 63) / 64);
  f16_clamp_mode_();
  Scales sc;
  {
    const float inv_keep = drop_inv_keep(a.drop_p);
    const float dy_amax = __builtin_nontemporal_load(a.dy_amax), w2_amax = __builtin_nontemporal_load(a.w2t_amax);
    const int e_dy = f16_sexp_(dy_amax), e_w2 = f16_sexp_(w2_amax), e_w1 = f16_sexp_(__builtin_nontemporal_load(a.w1_amax));
    const int e_in = operand_sexp_(a.in_amax, a.ln_sexp), e_mid = operand_sexp_(a.mid_amax, a.hid_sexp);
    // |dZ| <= amax(dY) inv_keep^2 64 amax(W2s) 1.1 (64 terms, |Swish'| < 1.1): a few binades loose, as in ff_bwd_kernel
    const int e_dz = f16_sexp_(dy_amax * inv_keep * inv_keep * 64.f * w2_amax * 1.1f);
    sc.s_in = exp2i_(e_in); sc.s_dy = exp2i_(e_dy); sc.s_s = exp2i_(e_mid); sc.s_z = exp2i_(e_dz);
    sc.uh = exp2i_(-e_in - e_w1);
    sc.u1 = exp2i_(-e_dy - e_w2 + e_dz);                 // accumulator of dP -> dZ at its fp16 scale
    sc.u2 = exp2i_(-e_dz - e_w1);
    sc.un1 = exp2i_(-e_dz - e_in); sc.un2 = a.alpha * exp2i_(-e_mid - e_dy);
    sc.ub1 = exp2i_(-e_dz); sc.ub2 = a.alpha * exp2i_(-e_dy);
  }
  if (tid < 256) reinterpret_cast<float*>(sm + O_B1)[tid] = a.b1[tid];
  if (tid < 128) reinterpret_cast<float*>(sm + O_GB)[tid] = tid < 64 ? a.gamma[tid] : a.beta[tid - 64];
  if (wave < 4) ff_fused_D(a, sm, wave, lane, mbeg, mend, ntile, sc);
  else ff_fused_W(a, sm, wave, lane, ntile, sc);
}

extern "C" int se_ff_bwd_fused(const float* dY, const float* X, const float* stats, const float* gamma, const float* beta,
                               const float* W1, const float* b1, const float* W2T, const float* dR2, float* dX, float* dgamma,
                               float* dbeta, float* dW1, float* db1, float* dW2, float* db2, long M, int hid, float drop_p,
                               unsigned seed_h, unsigned seed_o, float alpha, const float* dy_amax, const float* w1_amax,
                               const float* w2t_amax, const float* in_amax, int ln_sexp, const float* mid_amax, int hid_sexp,
                               float* out_amax, void* stream) {
  SE_REQUIRE(dY && X && stats && gamma && beta && W1 && b1 && W2T && dX && dgamma && dbeta && dW1 && db1 && dW2, "ff_bwd_fused: null operand");
  SE_REQUIRE(dy_amax && w1_amax && w2t_amax, "ff_bwd_fused: the operand amax scalars are required (scaled split-fp16)");
  SE_REQUIRE(M > 0 && hid == 256, "ff_bwd_fused: M=%ld hid=%d (built for hid == 256: four slots of 64 hidden units)", M, hid);
  SE_REQUIRE((((size_t)W1 | (size_t)W2T) & 15) == 0, "ff_bwd_fused: weight planes must be 16-byte aligned");
  SE_REQUIRE(drop_p >= 0.f && drop_p <= 0.5f && M * (long)hid < 4294967296L, "ff_bwd_fused: drop_p (keep >= 1/2) / dropout index out of range");
  // one persistent 8-wave workgroup per CU (146 KB of LDS): rows dealt in multiples of the 64-row tile; at least 4 tiles per
  // workgroup so that the 32 768 atomics a workgroup leaves with are amortised
  int dev = 0, ncu = 256;
  if (hipGetDevice(&dev) == hipSuccess) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ncu = v;
  }
  long rpw = (M + ncu - 1) / ncu;
  if (rpw < 256) rpw = 256;
  rpw = (rpw + 63) / 64 * 64;
  const int nwg = (int)((M + rpw - 1) / rpw);
  FfFusedArgs a{dY, X, stats, gamma, beta, W1, b1, W2T, dR2, dX, dgamma, dbeta, dW1, db1, dW2, db2, M, rpw, drop_p, seed_h, seed_o, alpha,
                dy_amax, w1_amax, w2t_amax, in_amax, mid_amax, out_amax, ln_sexp, hid_sexp};
  hipLaunchKernelGGL(ff_bwd_fused_kernel, dim3((unsigned)nwg), dim3(512), 0, as_stream(stream), a);
  return se_check_launch("se_ff_bwd_fused");
}
