// Fused backward of the Conformer feed-forward module, weight gradients included (round 5; this form: round 6).
//
//   Scale(0.5, PreNorm(FeedForward)) backward (conformer.py:53-71, 128-145) in ONE persistent launch that reads X, dY (, dR2) and
//   writes dX: the hidden pre-activations H are recomputed (the forward stores none), the hidden gradient dZ never leaves the CU,
//   and dW1 / db1 / dW2 / db2 accumulate on chip over the rows a workgroup sweeps and leave once through fp32 atomics.
//   Replaces se_ff_bwd_dgrad (1.68 GB per module at 518 736 rows) + two whole-gradient launches (0.67 + 0.70 GB) by 0.53 GB.
//
// One 8-wave workgroup per CU, rows in tiles of 32, scaled split-fp16 arithmetic (se_gemm_dev.h, precision 3).  Wave w owns hidden
// units 32 w .. 32 w + 31 for the whole launch: its rows of (alpha W2)^T are REGISTER-resident B fragments, its rows of W1 come out of
// the W1 image that is resident in LDS anyway (no weight block is staged per tile).  The products of a tile put the unit on the LANE
// and the tile's rows in the registers:
//   H[r][j] = LN(X) W1^T + b1,  dP[r][j] = (mask_o dY) W2s      A = the tile's rows (row images in LDS, 16-byte fragments), B = the weights;
//                                                              C: lane = unit j, register e = row (e & 3) + 8 (e >> 2) + 4 kg
//   S = Swish(H) mask_h,  dZ = dP mask_h Swish'(H)               registers -> fp16 (hi, lo) words that ARE the 16-deep fragments of
//   dW1[j][c] += sum_r dZ[r][j] LN[r][c]   (A = dZ, from registers)   the row contraction: no exchange image for the weight gradients; the
//   dW2[c][j] += sum_r dY[r][c] S[r][j]    (B = S, from registers)    other operand by hardware-transposed reads of the row images, its
//                                                                      rows taken in the C layout's order
//   dLN[r][c] = sum_j dZ[r][j] W1[j][c]    contraction over the LANE index: dZ goes through ONE transposed image ZT[unit][row] (S never
//                                          leaves the registers); wave = (channel half, hidden quarter), both operands by transposed
//                                          reads (ZT; the W1 image), four partial patches (ds_add_f32 into ONE patch: 9 - 23 K cycles
//                                          per tile for the 128 instructions)
// The dropout keep bits of (row, unit) are hashed once per group of four units by the four lanes that share it (each lane takes four
// of its 16 rows) and exchanged with DPP quad broadcasts; a dropped unit gets sigmoid := 0, which makes Swish and Swish' vanish.
// All images are unpadded and XOR-swizzled on the chunk the reads move (16 B for the row / W1 images, 8 B for ZT):
// tools/micro/lds_ff4_bench.hip measures every pattern at the conflict-free rate except the ZT writes (2-way).  Two barriers per tile:
//   | rows t + 1 requested; keep bits; H, dP; S, dZ, dZ -> ZT; dW2, dW1 | Q | dLN -> patches; LayerNorm / dropout / split of rows t + 1 ->
//   images | R | LayerNorm backward of tile t out of the patches -> dX | (next tile: no barrier)
// Measured at the bench shape (518 736 rows, same box, tools/ff_fused_bench.py): 425 - 450 us per launch against 505 - 525 of the form
// this replaces (round 5: four waves for the input gradient, four for the weight gradients, 64-row tiles, weight blocks staged per
// quarter of the hidden units, Z and S through row-major images, 18 barriers per 64 rows).  Built on the same segments and declined:
//   * a two-group schedule (waves 0-3 one segment ahead of waves 4-7: the two waves of a SIMD always in a matrix and a vector segment,
//     one barrier per segment): the same time within 10 us;
//   * all fragment reads of a matrix segment issued up front / ring-buffered two units ahead behind scheduling barriers: slower;
//   * db2 from the dY waves with 32 atomic instructions per workgroup on the TWO cache lines of db2: +115 us per launch -- atomics of
//     different workgroups to one cache line are serialised at the memory side (~14 ns per wave instruction); hence the fold below.
#include "se_ff_fused.h"
namespace ffb {
constexpr int RW = 128;                  // bytes of a row of the LN / dY / W1 images: 64 fp16, unpadded
constexpr int PL = 32 * RW, IMG = 2 * PL;            // one tile image (hi | lo)
constexpr int W1PL = 256 * RW;                        // plane of the W1 image [256 units][64 channels]
constexpr int ZTR = 64, ZTPL = 256 * ZTR;            // ZT[unit][32 rows] fp16
constexpr int O_W1 = 0, O_ZT = 2 * W1PL, O_ROWS = O_ZT + 2 * ZTPL;      // ROWS: [buffer][LN | dY] images
// dLN partial sums of the four hidden quarters, [32 rows][64 channels] fp32 each: quarters 0, 1 in their own 16 KB, quarters 2, 3 over
// the row images of the CURRENT tile (dead after barrier Q; rewritten by the prologue of tile t + 2 after barrier Q of tile t + 1)
constexpr int O_PATCH = O_ROWS + 2 * 2 * IMG, O_GB = O_PATCH + 2 * 32 * 64 * 4, O_DB2 = O_GB + 512, LDS_BYTES = O_DB2 + 256 * 32;
// DB2: lane-private column sums of mask_o dY (the 256 lanes of the dY waves x 8 channels): registers are the scarce resource
static_assert(LDS_BYTES <= 163840, "one workgroup per CU");
// swizzle of the 16-byte chunk index of row r: the four rows of one transposed read and the 16 rows of one ds_read_b128 lane group
// all land in different banks
static __device__ __forceinline__ int sw16(int r) { return (((r >> 1) & 1) << 2) | ((r >> 2) & 3); }

static __device__ __forceinline__ bf16x8 tr2_(const unsigned char* p0, const unsigned char* p1) {
  const u32x2_ t0 = fff::tr8_(p0), t1 = fff::tr8_(p1);
  return __builtin_bit_cast(bf16x8, (u32x4_){t0[0], t0[1], t1[0], t1[1]});
}
template <int CTRL>
static __device__ __forceinline__ float dpp_(float v) {
  return __uint_as_float((unsigned)__builtin_amdgcn_mov_dpp((int)__float_as_uint(v), CTRL, 0xf, 0xf, true));
}
// sum over the 16 lanes of a DPP row, on every lane (quad xor 1, quad xor 2, half mirror, mirror)
static __device__ __forceinline__ float row16_sum_(float v) {
  v += dpp_<0xB1>(v); v += dpp_<0x4E>(v); v += dpp_<0x141>(v); v += dpp_<0x140>(v);
  return v;
}
}  // namespace ffb


__global__ __launch_bounds__(512, 2) void ff_bwd_fused_kernel(FfFusedArgs a) {
  using namespace ffb;
  using fff::split4_;
  __shared__ __attribute__((aligned(128))) unsigned char sm[LDS_BYTES];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const long mbeg = (long)blockIdx.x * a.rows_per_wg;
  long mend = mbeg + a.rows_per_wg;
  if (mend > a.M) mend = a.M;
  if (mbeg >= mend) return;                              // (whole workgroup: block-uniform)
  const int ntile = (int)((mend - mbeg + 31) / 32);
  f16_clamp_mode_();
#ifdef SE_FF_STAMPS      // diagnostic build: scalar accumulators of the work (barrier release -> arrival) and the wait per barrier; no memory traffic in the loop
  unsigned long long tp_ = __builtin_amdgcn_s_memtime();
  const unsigned long long t00_ = tp_;
  unsigned wk_[2] = {0, 0}, wt_[2] = {0, 0};
#define FF_SYNC(k) do { const unsigned long long t1_ = __builtin_amdgcn_s_memtime(); wk_[k] += (unsigned)(t1_ - tp_); __syncthreads(); \
    tp_ = __builtin_amdgcn_s_memtime(); wt_[k] += (unsigned)(tp_ - t1_); } while (0)
#else
#define FF_SYNC(k) __syncthreads()
#endif
  const float inv_keep = drop_inv_keep(a.drop_p);
  const unsigned thr = drop_thr(a.drop_p);
  const bool dr = a.drop_p > 0.f;
  float s_in, s_dy, uh, u2, un1, un2, ub1, ub2, mkS, mkZ;
  {
    const float dy_amax = __builtin_nontemporal_load(a.dy_amax), w2_amax = __builtin_nontemporal_load(a.w2t_amax);
    const int e_dy = f16_sexp_(dy_amax), e_w2 = f16_sexp_(w2_amax), e_w1 = f16_sexp_(__builtin_nontemporal_load(a.w1_amax));
    const int e_in = operand_sexp_(a.in_amax, a.ln_sexp), e_mid = operand_sexp_(a.mid_amax, a.hid_sexp);
    const int e_dz = f16_sexp_(dy_amax * inv_keep * inv_keep * 64.f * w2_amax * 1.1f);
    s_in = exp2i_(e_in); s_dy = exp2i_(e_dy);
    uh = exp2i_(-e_in - e_w1);
    mkS = inv_keep * exp2i_(e_mid);
    mkZ = inv_keep * exp2i_(-e_dy - e_w2 + e_dz);
    u2 = exp2i_(-e_dz - e_w1);
    un1 = exp2i_(-e_dz - e_in); un2 = a.alpha * exp2i_(-e_mid - e_dy);
    ub1 = exp2i_(-e_dz); ub2 = a.alpha * exp2i_(-e_dy);
  }
  float one;
  asm volatile("s_mov_b32 %0, 1.0" : "=s"(one));
  const float* gbs = reinterpret_cast<const float*>(sm + O_GB);
  float* const patch = reinterpret_cast<float*>(sm + O_PATCH);
  const int j = lane & 31, kg = lane >> 5;
  const int gi = lane >> 4, q4 = (lane & 15) >> 2, p4 = lane & 3, kgt = gi >> 1;
  {
    const __bf16* W1p = reinterpret_cast<const __bf16*>(a.W1);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int id = tid + 512 * i, pl = id >> 11, row = (id >> 3) & 255, ch = id & 7;
      const float4 v = *reinterpret_cast<const float4*>(W1p + (size_t)pl * 256 * 64 + row * 64 + 8 * ch);
      *reinterpret_cast<float4*>(sm + O_W1 + pl * W1PL + row * RW + ((ch ^ sw16(row)) << 4)) = v;
    }
    if (tid < 128) reinterpret_cast<float*>(sm + O_GB)[tid] = tid < 64 ? a.gamma[tid] : a.beta[tid - 64];
    if (tid < 256) { float* z = reinterpret_cast<float*>(sm + O_DB2) + (tid << 3); *reinterpret_cast<float4*>(z) = make_float4(0.f, 0.f, 0.f, 0.f); *reinterpret_cast<float4*>(z + 4) = make_float4(0.f, 0.f, 0.f, 0.f); }
  }
  bf16x8 W2b[4][2];
  {
    const size_t wpl = (size_t)256 * 64;
    const __bf16* p2 = reinterpret_cast<const __bf16*>(a.W2T) + (size_t)(32 * wave + j) * 64 + 8 * kg;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) {
        W2b[ks][pl] = *reinterpret_cast<const bf16x8*>(p2 + pl * wpl + 16 * ks);
      }
  }
  const float bias = a.b1[32 * wave + j];
  f32x16 aw1[2], aw2[2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int e = 0; e < 16; ++e) { aw1[nt][e] = 0.f; aw2[nt][e] = 0.f; }
  float bs1 = 0.f;
  float agk = 0.f, abk = 0.f, xmax = 0.f;
  const int fragA = j * RW, x16 = ((kg ^ sw16(j)) << 4);
  const int trA0 = ((4 * kgt + q4) * RW + (((2 * (gi & 1) + (p4 >> 1)) ^ (((q4 >> 1) << 2) | kgt)) << 4) + 8 * (p4 & 1));
  const int trA1 = (trA0 ^ 32) + 8 * RW;
  const int ztw = O_ZT + (32 * wave + j) * ZTR + ((kg ^ ((j >> 1) & 7)) << 3);
  const int ch = wave & 1, kq = wave >> 1;
  const int jz = 64 * kq + 8 * kgt + q4;
  const int ztr0 = O_ZT + jz * ZTR + (((4 * (gi & 1) + p4) ^ ((4 * kgt) | (q4 >> 1))) << 3);
  const int ztr1 = O_ZT + (jz + 4) * ZTR + (((4 * (gi & 1) + p4) ^ ((4 * kgt) | (q4 >> 1) | 2)) << 3);
  const int c16w = 4 * ch + 2 * (gi & 1) + (p4 >> 1);
  const int w1r0 = O_W1 + jz * RW + ((c16w ^ (((q4 >> 1) << 2) | (2 * kgt))) << 4) + 8 * (p4 & 1);
  const int w1r1 = O_W1 + (jz + 4) * RW + ((c16w ^ (((q4 >> 1) << 2) | (2 * kgt) | 1)) << 4) + 8 * (p4 & 1);
  const int padd = (4 * kg * 64 + 32 * ch + j) * 4 + (kq & 1) * 8192;
  const int err = lane >> 4, ecq = lane & 15;
  const int pten = wave >> 2, prow = 8 * (wave & 3) + (lane >> 3), poct = lane & 7;
  const int pst = pten * IMG + prow * RW + ((poct ^ sw16(prow)) << 4);
  const unsigned hash_lane = (unsigned)((8 * (j & 3) + 4 * kg) * 64 + 8 * wave + (j >> 2)) * 0x9E3779B1u;

  float4 raw0, raw1; float2 rst;
  auto load_raw = [&](long m0) {
    const long avail = a.M - m0 < 32 ? a.M - m0 : 32;
    const __amdgpu_buffer_rsrc_t Rr = make_rsrc_((pten == 0 ? a.X : a.dY) + m0 * 64, avail > 0 ? (unsigned)(avail * 256) : 0u);
    const __amdgpu_buffer_rsrc_t Sr = make_rsrc_(a.stats + m0 * 2, avail > 0 ? (unsigned)(avail * 8) : 0u);
    raw0 = buf_load4_(Rr, (unsigned)(prow * 256 + poct * 32));
    raw1 = buf_load4_(Rr, (unsigned)(prow * 256 + poct * 32 + 16));
    rst = buf_load2_(Sr, (unsigned)(prow * 8));
  };
  auto prologue_store = [&](long m0, int buf) {
    const long m = m0 + prow;
    const bool ok = m < mend;
    float x[8];
    if (pten == 0) {
      const float4 g0 = *reinterpret_cast<const float4*>(gbs + 8 * poct), g1 = *reinterpret_cast<const float4*>(gbs + 8 * poct + 4);
      const float4 t0 = *reinterpret_cast<const float4*>(gbs + 64 + 8 * poct), t1 = *reinterpret_cast<const float4*>(gbs + 64 + 8 * poct + 4);
      const float mean = rst.x, rstd = rst.y;
      x[0] = (raw0.x - mean) * rstd * g0.x + t0.x; x[1] = (raw0.y - mean) * rstd * g0.y + t0.y;
      x[2] = (raw0.z - mean) * rstd * g0.z + t0.z; x[3] = (raw0.w - mean) * rstd * g0.w + t0.w;
      x[4] = (raw1.x - mean) * rstd * g1.x + t1.x; x[5] = (raw1.y - mean) * rstd * g1.y + t1.y;
      x[6] = (raw1.z - mean) * rstd * g1.z + t1.z; x[7] = (raw1.w - mean) * rstd * g1.w + t1.w;
    } else {
      float4 d0 = make_float4(1.f, 1.f, 1.f, 1.f), d1 = d0;
      if (dr) {
        d0 = drop_scale4(a.seed_o, (unsigned)(m * 64 + 8 * poct), thr, inv_keep);
        d1 = drop_scale4(a.seed_o, (unsigned)(m * 64 + 8 * poct + 4), thr, inv_keep);
      }
      x[0] = raw0.x * d0.x; x[1] = raw0.y * d0.y; x[2] = raw0.z * d0.z; x[3] = raw0.w * d0.w;
      x[4] = raw1.x * d1.x; x[5] = raw1.y * d1.y; x[6] = raw1.z * d1.z; x[7] = raw1.w * d1.w;
      float* dbc = reinterpret_cast<float*>(sm + O_DB2) + ((tid & 255) << 3);
      float4 c0 = *reinterpret_cast<const float4*>(dbc), c1 = *reinterpret_cast<const float4*>(dbc + 4);
      const float k = ok ? 1.f : 0.f;
      c0.x += k * x[0]; c0.y += k * x[1]; c0.z += k * x[2]; c0.w += k * x[3]; c1.x += k * x[4]; c1.y += k * x[5]; c1.z += k * x[6]; c1.w += k * x[7];
      *reinterpret_cast<float4*>(dbc) = c0; *reinterpret_cast<float4*>(dbc + 4) = c1;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = ok ? x[e] : 0.f;
    bf16x8 o[2];
    split_planes8_h(x, pten == 0 ? s_in : s_dy, o);
    unsigned char* p = sm + O_ROWS + buf * 2 * IMG + pst;
    *reinterpret_cast<bf16x8*>(p) = o[0];
    *reinterpret_cast<bf16x8*>(p + PL) = o[1];
  };

  load_raw(mbeg);
  __syncthreads();
  prologue_store(mbeg, 0);
  __syncthreads();
  for (int t = 0; t < ntile; ++t) {
    const long m0 = mbeg + 32L * t;
    const bool more = t + 1 < ntile;
    const unsigned char* const rows = sm + O_ROWS + (t & 1) * 2 * IMG;
    const long avail = a.M - m0 < 32 ? a.M - m0 : 32;
    const unsigned eo = (unsigned)((4 * wave + err) * 256 + ecq * 16);
    load_raw(m0 + 32);
    float4 er2 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a.dR2) er2 = buf_load4_(make_rsrc_(a.dR2 + m0 * 64, (unsigned)(avail * 256)), eo);
    unsigned kw[4] = {0x1111u, 0x1111u, 0x1111u, 0x1111u};
    if (dr) {
      const unsigned base = hash_lane + (unsigned)(m0 * 64) * 0x9E3779B1u;
      unsigned w = 0;
#pragma unroll
      for (int i = 0; i < 4; ++i) w |= drop_keep4_pre(a.seed_h, base + (unsigned)(i * 64) * 0x9E3779B1u, thr) << (4 * i);
      const int sh = j & 3;
      kw[0] = (unsigned)__builtin_amdgcn_mov_dpp((int)w, 0x00, 0xf, 0xf, true) >> sh;
      kw[1] = (unsigned)__builtin_amdgcn_mov_dpp((int)w, 0x55, 0xf, 0xf, true) >> sh;
      kw[2] = (unsigned)__builtin_amdgcn_mov_dpp((int)w, 0xAA, 0xf, 0xf, true) >> sh;
      kw[3] = (unsigned)__builtin_amdgcn_mov_dpp((int)w, 0xFF, 0xf, 0xf, true) >> sh;
    }
    f32x16 ah, ad;
#pragma unroll
    for (int e = 0; e < 16; ++e) { ah[e] = 0.f; ad[e] = 0.f; }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const unsigned char* p = rows + fragA + (x16 ^ (32 * ks));
      const bf16x8 lh = *reinterpret_cast<const bf16x8*>(p), ll = *reinterpret_cast<const bf16x8*>(p + PL);
      const unsigned char* pw = sm + O_W1 + (32 * wave + j) * RW + (x16 ^ (32 * ks));
      const bf16x8 w_h = *reinterpret_cast<const bf16x8*>(pw), w_l = *reinterpret_cast<const bf16x8*>(pw + W1PL);
      ah = mfma32_<true>(ll, w_h, ah);
      ah = mfma32_<true>(lh, w_l, ah);
      ah = mfma32_<true>(lh, w_h, ah);
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const unsigned char* p = rows + IMG + fragA + (x16 ^ (32 * ks));
      const bf16x8 yh = *reinterpret_cast<const bf16x8*>(p), yl = *reinterpret_cast<const bf16x8*>(p + PL);
      ad = mfma32_<true>(yl, W2b[ks][0], ad);
      ad = mfma32_<true>(yh, W2b[ks][1], ad);
      ad = mfma32_<true>(yh, W2b[ks][0], ad);
    }
    unsigned sh_[4][2], sl_[4][2], zh_[4][2], zl_[4][2];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float sv[4], zv[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int i = 4 * q + e;
        const float h = fmaf(ah[i], uh, bias);
        float sg = sigmoidf_(h);
        sg = __uint_as_float(__float_as_uint(sg) & (unsigned)__builtin_amdgcn_sbfe((int)kw[q], 4 * e, 1));
        const float s0 = h * sg;
        const float sw = fmaf(s0, 1.0f - sg, sg);
        sv[e] = s0 * mkS;
        zv[e] = ad[i] * sw * mkZ;
        bs1 += zv[e];
      }
      split4_(sv[0], sv[1], sv[2], sv[3], one, sh_[q][0], sh_[q][1], sl_[q][0], sl_[q][1]);
      split4_(zv[0], zv[1], zv[2], zv[3], one, zh_[q][0], zh_[q][1], zl_[q][0], zl_[q][1]);
      *reinterpret_cast<u32x2_*>(sm + (ztw ^ (16 * q))) = (u32x2_){zh_[q][0], zh_[q][1]};
      *reinterpret_cast<u32x2_*>(sm + (ztw ^ (16 * q)) + ZTPL) = (u32x2_){zl_[q][0], zl_[q][1]};
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const bf16x8 s_h = __builtin_bit_cast(bf16x8, (u32x4_){sh_[2 * ks][0], sh_[2 * ks][1], sh_[2 * ks + 1][0], sh_[2 * ks + 1][1]});
      const bf16x8 s_l = __builtin_bit_cast(bf16x8, (u32x4_){sl_[2 * ks][0], sl_[2 * ks][1], sl_[2 * ks + 1][0], sl_[2 * ks + 1][1]});
      const bf16x8 z_h = __builtin_bit_cast(bf16x8, (u32x4_){zh_[2 * ks][0], zh_[2 * ks][1], zh_[2 * ks + 1][0], zh_[2 * ks + 1][1]});
      const bf16x8 z_l = __builtin_bit_cast(bf16x8, (u32x4_){zl_[2 * ks][0], zl_[2 * ks][1], zl_[2 * ks + 1][0], zl_[2 * ks + 1][1]});
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const unsigned char* p0 = rows + ((trA0 ^ (64 * nt)) + 16 * RW * ks), *p1 = rows + ((trA1 ^ (64 * nt)) + 16 * RW * ks);
        bf16x8 y_h, y_l;
        y_h = tr2_(p0 + IMG, p1 + IMG); y_l = tr2_(p0 + IMG + PL, p1 + IMG + PL);
        aw2[nt] = mfma32_<true>(y_h, s_l, aw2[nt]);
        aw2[nt] = mfma32_<true>(y_l, s_h, aw2[nt]);
        aw2[nt] = mfma32_<true>(y_h, s_h, aw2[nt]);
        const bf16x8 l_h = tr2_(p0, p1), l_l = tr2_(p0 + PL, p1 + PL);
        aw1[nt] = mfma32_<true>(z_h, l_l, aw1[nt]);
        aw1[nt] = mfma32_<true>(z_l, l_h, aw1[nt]);
        aw1[nt] = mfma32_<true>(z_h, l_h, aw1[nt]);
      }
    }
    FF_SYNC(0);                                          // (Q) ZT complete; the row images of this tile are free
    const float4 ex = buf_load4_(make_rsrc_(a.X + m0 * 64, (unsigned)(avail * 256)), eo);
    const float4 ey = buf_load4_(make_rsrc_(a.dY + m0 * 64, (unsigned)(avail * 256)), eo);
    const float2 est = buf_load2_(make_rsrc_(a.stats + m0 * 2, (unsigned)(avail * 8)), (unsigned)((4 * wave + err) * 8));
    {
      unsigned char* const pbase = kq < 2 ? sm + O_PATCH : sm + O_ROWS + (t & 1) * 2 * IMG;
      f32x16 gl;
#pragma unroll
      for (int e = 0; e < 16; ++e) gl[e] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const bf16x8 zh = tr2_(sm + ztr0 + 16 * ks * ZTR, sm + ztr1 + 16 * ks * ZTR);
        const bf16x8 zl = tr2_(sm + ztr0 + ZTPL + 16 * ks * ZTR, sm + ztr1 + ZTPL + 16 * ks * ZTR);
        const bf16x8 wh = tr2_(sm + w1r0 + 16 * ks * RW, sm + w1r1 + 16 * ks * RW);
        const bf16x8 wl = tr2_(sm + w1r0 + W1PL + 16 * ks * RW, sm + w1r1 + W1PL + 16 * ks * RW);
        gl = mfma32_<true>(zh, wl, gl);
        gl = mfma32_<true>(zl, wh, gl);
        gl = mfma32_<true>(zh, wh, gl);
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) *reinterpret_cast<float*>(pbase + padd + ((e & 3) + 8 * (e >> 2)) * 256) = gl[e] * u2;
    }
    if (more) prologue_store(m0 + 32, (t + 1) & 1);
    FF_SYNC(1);                                          // (R) patches complete; ZT free; images of tile t + 1 complete
    {
      const int rl = 4 * wave + err;
      const bool ok = m0 + rl < mend;
      const float* pp = patch + rl * 64 + 4 * ecq;
      const float* pq = reinterpret_cast<const float*>(rows) + rl * 64 + 4 * ecq;
      const float4 p0 = *reinterpret_cast<const float4*>(pp), p1 = *reinterpret_cast<const float4*>(pp + 2048);
      const float4 p2 = *reinterpret_cast<const float4*>(pq), p3 = *reinterpret_cast<const float4*>(pq + 2048);
      const float dv[4] = {(p0.x + p1.x) + (p2.x + p3.x), (p0.y + p1.y) + (p2.y + p3.y), (p0.z + p1.z) + (p2.z + p3.z),
                           (p0.w + p1.w) + (p2.w + p3.w)};
      const float4 gm = *reinterpret_cast<const float4*>(gbs + 4 * ecq);
      const float gl4[4] = {gm.x, gm.y, gm.z, gm.w}, xs[4] = {ex.x, ex.y, ex.z, ex.w};
      const float mean = est.x, rstd = est.y;
      float xh[4], dxh[4], s1 = 0.f, s2 = 0.f, ag[4], ab[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        xh[e] = (xs[e] - mean) * rstd;
        dxh[e] = dv[e] * gl4[e];
        s1 += dxh[e]; s2 += dxh[e] * xh[e];
        ag[e] = ok ? dv[e] * xh[e] : 0.f; ab[e] = ok ? dv[e] : 0.f;
      }
      s1 = row16_sum_(s1) * (1.f / 64.f); s2 = row16_sum_(s2) * (1.f / 64.f);
      float o4[4] = {ey.x + er2.x, ey.y + er2.y, ey.z + er2.z, ey.w + er2.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) o4[e] += rstd * (dxh[e] - s1 - xh[e] * s2);
      const long rows_ok = mend - m0 < 32 ? mend - m0 : 32;
      buf_store4_(make_rsrc_(a.dX + m0 * 64, (unsigned)(rows_ok * 256)), (unsigned)(rl * 256 + ecq * 16), make_float4(o4[0], o4[1], o4[2], o4[3]));
      if (ok) xmax = fmaxf(fmaxf(xmax, fmaxf(fabsf(o4[0]), fabsf(o4[1]))), fmaxf(fabsf(o4[2]), fabsf(o4[3])));
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float sg = xor32_sum_(xor16_sum_(ag[e])), sb = xor32_sum_(xor16_sum_(ab[e]));
        agk += err == e ? sg : 0.f;
        abk += err == e ? sb : 0.f;
      }
    }
  }
  // ---- leave.  The SMALL vectors (gamma / beta / b2 gradients: 64 floats = two cache lines each) are folded across the workgroup in LDS
  // and leave through ONE atomic instruction per vector: atomics of different workgroups to the same cache line are serialised at the
  // memory side (~14 ns per wave instruction): 32 instructions per workgroup on the b2 lines cost 115 us of a 450 us launch ----
  xmax = wave_max(xmax);
  __syncthreads();                                       // (the row images are free: fold area)
  {
    float* fold = reinterpret_cast<float*>(sm + O_ROWS);      // [wave 8][gamma 64 | beta 64], [dY wave 4][64], [wave 8] dX maxima
    if (lane == 0) fold[1280 + wave] = xmax;
    fold[wave * 128 + 4 * ecq + err] = agk;
    fold[wave * 128 + 64 + 4 * ecq + err] = abk;
    if (pten == 1) {
      const float* dbc = reinterpret_cast<const float*>(sm + O_DB2) + ((tid & 255) << 3);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float v = dbc[e];
        v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
        if (lane < 8) fold[1024 + (wave - 4) * 64 + 8 * poct + e] = v;
      }
    }
  }
  __syncthreads();
  if (wave == 0) {
    const float* fold = reinterpret_cast<const float*>(sm + O_ROWS);
    float g = 0.f, bt = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) { g += fold[w * 128 + lane]; bt += fold[w * 128 + 64 + lane]; }
    atomicAdd(&a.dgamma[lane], g);
    atomicAdd(&a.dbeta[lane], bt);
    if (a.out_amax && lane == 0) {
      float m = fold[1280];
#pragma unroll
      for (int w = 1; w < 8; ++w) m = fmaxf(m, fold[1280 + w]);
      amax_raise_(a.out_amax, m);
    }
    if (a.db2) atomicAdd(&a.db2[lane], ((fold[1024 + lane] + fold[1088 + lane]) + (fold[1152 + lane] + fold[1216 + lane])) * a.alpha);
  }
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = (e & 3) + 8 * (e >> 2) + 4 * kg;
      atomicAdd(&a.dW1[(32 * wave + row) * 64 + 32 * nt + j], aw1[nt][e] * un1);
      atomicAdd(&a.dW2[(32 * nt + row) * 256 + 32 * wave + j], aw2[nt][e] * un2);
    }
  {
    const float v = bs1 + __shfl_xor(bs1, 32, 64);
    if (kg == 0) atomicAdd(&a.db1[32 * wave + j], v * ub1);
  }
#ifdef SE_FF_STAMPS
  if (a.stamps && tid == 0) { a.stamps[4096 + blockIdx.x] = (unsigned)(__builtin_amdgcn_s_memtime() - t00_); a.stamps[4096 + 512 + blockIdx.x] = (unsigned)(tp_ - t00_); }
  if (a.stamps && blockIdx.x < 4 && lane == 0) {
    for (int k = 0; k < 2; ++k) { a.stamps[((int)blockIdx.x * 8 + wave) * 16 + k] = wk_[k]; a.stamps[((int)blockIdx.x * 8 + wave) * 16 + 4 + k] = wt_[k]; }
    a.stamps[((int)blockIdx.x * 8 + wave) * 16 + 8] = (unsigned)ntile;
  }
#endif
}

#ifdef SE_FF_STAMPS
static unsigned* g_ff_stamps = nullptr;
extern "C" void se_ff_fused_debug_stamps(void* p) { g_ff_stamps = reinterpret_cast<unsigned*>(p); }
#endif

extern "C" int se_ff_bwd_fused(const float* dY, const float* X, const float* stats, const float* gamma, const float* beta,
                               const float* W1, const float* b1, const float* W2T, const float* dR2, float* dX, float* dgamma,
                               float* dbeta, float* dW1, float* db1, float* dW2, float* db2, long M, int hid, float drop_p,
                               unsigned seed_h, unsigned seed_o, float alpha, const float* dy_amax, const float* w1_amax,
                               const float* w2t_amax, const float* in_amax, int ln_sexp, const float* mid_amax, int hid_sexp,
                               float* out_amax, void* stream) {
  SE_REQUIRE(dY && X && stats && gamma && beta && W1 && b1 && W2T && dX && dgamma && dbeta && dW1 && db1 && dW2, "ff_bwd_fused: null operand");
  SE_REQUIRE(dy_amax && w1_amax && w2t_amax, "ff_bwd_fused: the operand amax scalars are required (scaled split-fp16)");
  SE_REQUIRE(M > 0 && hid == 256, "ff_bwd_fused: M=%ld hid=%d (built for hid == 256: eight waves of 32 hidden units)", M, hid);
  SE_REQUIRE((((size_t)W1 | (size_t)W2T) & 15) == 0, "ff_bwd_fused: weight planes must be 16-byte aligned");
  SE_REQUIRE(drop_p >= 0.f && drop_p <= 0.5f && M * (long)hid < 4294967296L, "ff_bwd_fused: drop_p (keep >= 1/2) / dropout index out of range");
  // one persistent 8-wave workgroup per CU (156 KB of LDS): rows dealt in multiples of the 32-row tile; at least 8 tiles per
  // workgroup so that the 32 768 atomics a workgroup leaves with are amortised
  const int ncu = se_cu_count();
  long rpw = (M + ncu - 1) / ncu;
  if (rpw < 256) rpw = 256;
  rpw = (rpw + 31) / 32 * 32;
  const int nwg = (int)((M + rpw - 1) / rpw);
  FfFusedArgs a{dY, X, stats, gamma, beta, W1, b1, W2T, dR2, dX, dgamma, dbeta, dW1, db1, dW2, db2, M, rpw, drop_p, seed_h, seed_o, alpha,
                dy_amax, w1_amax, w2t_amax, in_amax, mid_amax, out_amax, ln_sexp, hid_sexp, 0, nullptr};
#ifdef SE_FF_STAMPS      // diagnostic builds only (tools/ff_fused_stamps.py)
  a.stamps = g_ff_stamps;
#endif
  hipLaunchKernelGGL(ff_bwd_fused_kernel, dim3((unsigned)nwg), dim3(512), 0, as_stream(stream), a);
  return se_check_launch("se_ff_bwd_fused");
}
