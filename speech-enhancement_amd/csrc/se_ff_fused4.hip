// Fused backward of the Conformer feed-forward module, weight gradients included -- symmetric waves, hidden unit on the lane (round 6).
//
//   Scale(0.5, PreNorm(FeedForward)) backward (conformer.py:53-71, 128-145) in ONE persistent launch that reads X, dY (, dR2) and
//   writes dX; H is recomputed, dW1 / db1 / dW2 / db2 accumulate in registers over the rows a workgroup sweeps and leave once.
//
// One 8-wave workgroup per CU, rows in tiles of 32; wave w owns hidden units 32 w .. 32 w + 31 for the whole launch: its rows of W1
// and of (alpha W2)^T are REGISTER-resident B fragments (no weight block is ever staged per tile).  The products of a tile are laid
// out so that the unit is on the LANE and the tile's rows are in the registers:
//   H[r][j] = LN(X) W1^T + b1,  dP[r][j] = (mask_o dY) W2s      A = the tile's rows (row images in LDS, 16-byte fragments), B = the weights;
//                                                              C: lane = unit j, register e = row (e & 3) + 8 (e >> 2) + 4 kg
//   S = Swish(H) mask_h,  dZ = dP mask_h Swish'(H)               registers -> fp16 (hi, lo) words that ARE the 16-deep fragments of
//   dW1[j][c] += sum_r dZ[r][j] LN[r][c]   (A = dZ, from registers)   the row contraction: no exchange image for the weight gradients; the
//   dW2[c][j] += sum_r dY[r][c] S[r][j]    (B = S, from registers)    other operand by hardware-transposed reads of the row images, its
//                                                                      rows taken in the C layout's order
//   dLN[r][c] = sum_j dZ[r][j] W1[j][c]    contraction over the LANE index: dZ goes through ONE transposed image ZT[unit][row] (S never
//                                          leaves the registers); wave = (channel half, hidden quarter), both operands by transposed
//                                          reads (ZT; the W1 image, resident in LDS for the whole launch), four partial patches
//                                          (ds_add_f32 into one patch: 9 - 23 K cycles per tile for the 128 instructions)
// The dropout keep bits of (row, unit) are hashed once per group of four units by the four lanes that share it (each lane takes four
// of its 16 rows) and exchanged with DPP quad broadcasts.  All images are unpadded and XOR-swizzled on the chunk the reads move
// (16 B for the row / W1 images, 8 B for ZT): every ds_read_b128, ds_read_b64_tr_b16 and ds_write_b64 of the tile loop is conflict-free
// in the bank model of tools/micro/lds_bank_model.py.  Two barriers per 32 rows:
//   | rows t + 1 requested; H, dP, S, dZ, dW2, dW1; dZ -> ZT | Q | dLN -> patch; LayerNorm / dropout / split of rows t + 1 -> images | R |
//   LayerNorm backward of tile t out of the patch -> dX | (next tile: no barrier)
// (The specialised-wave form this replaces -- four waves for the input gradient, four for the weight gradients, 64-row tiles, weight
// blocks staged per quarter, Z and S both through row-major images, 18 barriers per 64 rows -- ran at 453 us per launch at the bench
// shape: profiles/r05_ff_fused_v2_stamps.txt.)
#include "se_ff_fused.h"

namespace ff4 {
constexpr int RW = 128;                  // bytes of a row of the LN / dY / W1 images: 64 fp16, unpadded
constexpr int PL = 32 * RW, IMG = 2 * PL;            // one tile image (hi | lo)
constexpr int W1PL = 256 * RW;                        // plane of the W1 image [256 units][64 channels]
constexpr int ZTR = 64, ZTPL = 256 * ZTR;            // ZT[unit][32 rows] fp16
constexpr int O_W1 = 0, O_ZT = 2 * W1PL, O_ROWS = O_ZT + 2 * ZTPL;      // ROWS: [buffer][LN | dY] images
// dLN partial sums, [32 rows][64 channels] fp32 each: patch q holds hidden quarter q, then (added a segment later) quarter q + 2
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
static __device__ __forceinline__ bf16x8 tr2_sum_(const unsigned char* p0, const unsigned char* p1, float& acc) {
  const u32x2_ t0 = fff::tr8_(p0), t1 = fff::tr8_(p1);
  const unsigned a0 = t0[0], a1 = t0[1], a2 = t1[0], a3 = t1[1];
  const h2f_ one = {(_Float16)1.0f, (_Float16)1.0f};
  acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(h2f_, a0), one, acc, false);
  acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(h2f_, a1), one, acc, false);
  acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(h2f_, a2), one, acc, false);
  acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(h2f_, a3), one, acc, false);
  return __builtin_bit_cast(bf16x8, (u32x4_){a0, a1, a2, a3});
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
}  // namespace ff4

template <bool DR>
__global__ __launch_bounds__(512, 2) void ff_bwd_fused4_kernel(FfFusedArgs a) {
  using namespace ff4;
  using fff::split4_;
  __shared__ __attribute__((aligned(128))) unsigned char sm[LDS_BYTES];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const long mbeg = (long)blockIdx.x * a.rows_per_wg;
  long mend = mbeg + a.rows_per_wg;
  if (mend > a.M) mend = a.M;
  if (mbeg >= mend) return;                              // (whole workgroup: block-uniform)
  const int ntile = (int)((mend - mbeg + 31) / 32);
  f16_clamp_mode_();
#ifdef SE_FF_STAMPS      // diagnostic build: scalar accumulators of every slot's work (release -> arrival) and wait (arrival -> release); no memory traffic in the loop
  unsigned long long tp_ = __builtin_amdgcn_s_memtime();
  const unsigned long long t00_ = tp_;
  unsigned wk_[4] = {0, 0, 0, 0}, wt_[4] = {0, 0, 0, 0};
#define FF_SYNC(k) do { const unsigned long long t1_ = __builtin_amdgcn_s_memtime(); wk_[k] += (unsigned)(t1_ - tp_); __syncthreads(); \
    tp_ = __builtin_amdgcn_s_memtime(); wt_[k] += (unsigned)(tp_ - t1_); } while (0)
#else
#define FF_SYNC(k) __syncthreads()
#endif
  // ---- scales ----
  const float inv_keep = drop_inv_keep(a.drop_p);
  const unsigned thr = drop_thr(a.drop_p);
  float s_in, s_dy, uh, u2, un1, un2, ub1, ub2, mkS, mkZ;
  {
    const float dy_amax = __builtin_nontemporal_load(a.dy_amax), w2_amax = __builtin_nontemporal_load(a.w2t_amax);
    const int e_dy = f16_sexp_(dy_amax), e_w2 = f16_sexp_(w2_amax), e_w1 = f16_sexp_(__builtin_nontemporal_load(a.w1_amax));
    const int e_in = operand_sexp_(a.in_amax, a.ln_sexp), e_mid = operand_sexp_(a.mid_amax, a.hid_sexp);
    // |dZ| <= amax(dY) inv_keep^2 64 amax(W2s) 1.1 (64 terms, |Swish'| < 1.1): a few binades loose, as in ff_bwd_kernel
    const int e_dz = f16_sexp_(dy_amax * inv_keep * inv_keep * 64.f * w2_amax * 1.1f);
    s_in = exp2i_(e_in); s_dy = exp2i_(e_dy);
    uh = exp2i_(-e_in - e_w1);
    mkS = inv_keep * exp2i_(e_mid);
    mkZ = inv_keep * exp2i_(-e_dy - e_w2 + e_dz);         // accumulator of dP -> dZ at its fp16 scale
    u2 = exp2i_(-e_dz - e_w1);
    un1 = exp2i_(-e_dz - e_in); un2 = a.alpha * exp2i_(-e_mid - e_dy);
    ub1 = exp2i_(-e_dz); ub2 = a.alpha * exp2i_(-e_dy);
  }
  float one;
  asm volatile("s_mov_b32 %0, 1.0" : "=s"(one));
  const float* gbs = reinterpret_cast<const float*>(sm + O_GB);
  const float* const patch = reinterpret_cast<const float*>(sm + O_PATCH);

  const int j = lane & 31, kg = lane >> 5;                // products: unit 32 w + j on the lane (A fragments: row j of the tile), k group
  const int gi = lane >> 4, q4 = (lane & 15) >> 2, p4 = lane & 3, kgt = gi >> 1;
  // ---- stage the W1 image (all waves; once), gamma / beta ----
  {
    const __bf16* W1p = reinterpret_cast<const __bf16*>(a.W1);
#pragma unroll
    for (int i = 0; i < 8; ++i) {                         // 2 planes x 256 rows x 8 chunks = 4096 chunks of 16 B
      const int id = tid + 512 * i, pl = id >> 11, row = (id >> 3) & 255, ch = id & 7;
      const float4 v = *reinterpret_cast<const float4*>(W1p + (size_t)pl * 256 * 64 + row * 64 + 8 * ch);
      *reinterpret_cast<float4*>(sm + O_W1 + pl * W1PL + row * RW + ((ch ^ sw16(row)) << 4)) = v;
    }
    if (tid < 128) reinterpret_cast<float*>(sm + O_GB)[tid] = tid < 64 ? a.gamma[tid] : a.beta[tid - 64];
  }
  // ---- this wave's rows of (alpha W2)^T as B fragments, for the whole launch: lane (unit 32 w + j, kg) holds channels 16 ks + 8 kg .. ----
  // (W1's fragments of the same shape come out of the LDS image every tile: 32 registers the tile loop needs more than 8 KB of reads)
  bf16x8 W2b[4][2];
  {
    const size_t wpl = (size_t)256 * 64;
    const __bf16* p2 = reinterpret_cast<const __bf16*>(a.W2T) + (size_t)(32 * wave + j) * 64 + 8 * kg;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) W2b[ks][pl] = *reinterpret_cast<const bf16x8*>(p2 + pl * wpl + 16 * ks);
  }
  const float bias = a.b1[32 * wave + j];
  // weight-gradient accumulators of the wave's 32 units: dW1[j][c] (two channel halves), dW2[c][j] (two channel halves)
  f32x16 aw1[2], aw2[2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int e = 0; e < 16; ++e) { aw1[nt][e] = 0.f; aw2[nt][e] = 0.f; }
  float bs1 = 0.f;                                        // db1 of unit j (this lane's rows)
  float agk = 0.f, abk = 0.f, xmax = 0.f;                 // gamma / beta gradients of channel 4 ecq + err (see the epilogue)
  float* const dbc = reinterpret_cast<float*>(sm + O_DB2) + ((tid & 255) << 3);      // prologue role, dY waves: db2 of channels 8 poct .. + 7 (this lane's rows)
  if (tid < 256) { *reinterpret_cast<float4*>(dbc) = make_float4(0.f, 0.f, 0.f, 0.f); *reinterpret_cast<float4*>(dbc + 4) = make_float4(0.f, 0.f, 0.f, 0.f); }

  // ---- per-lane LDS offsets (relative to the row-image buffer / the image bases) ----
  // A fragments of the row images: row j, chunk (2 ks + kg) ^ sw16(j)  ->  byte (x16 ^ 32 ks)
  const int fragA = j * RW, x16 = ((kg ^ sw16(j)) << 4);
  // transposed reads of the row images, rows in the C layout's order: t0 = rows 16 ksub + 4 kgt + q4, t1 = + 8; column 32 nt + 16 (gi & 1) + 4 p4
  const int trA0 = ((4 * kgt + q4) * RW + (((2 * (gi & 1) + (p4 >> 1)) ^ (((q4 >> 1) << 2) | kgt)) << 4) + 8 * (p4 & 1));
  const int trA1 = (trA0 ^ 32) + 8 * RW;                  // (nt: ^ 64; ksub: + 16 RW; plane: + PL; dY: + IMG)
  // ZT writes: row 32 w + j, 8-byte chunk (2 q + kg) ^ ((j >> 1) & 7)  ->  (ztw ^ 16 q)
  const int ztw = O_ZT + (32 * wave + j) * ZTR + ((kg ^ ((j >> 1) & 7)) << 3);
  // dLN role: channel half ch, hidden quarter kq; transposed reads of ZT (A: column = row of the tile) and of the W1 image (B: column = channel)
  const int ch = wave & 1, kq = wave >> 1;
  const int jz = 64 * kq + 8 * kgt + q4;                  // + 16 ks (+ 4 for the second read)
  const int ztr0 = O_ZT + jz * ZTR + (((4 * (gi & 1) + p4) ^ ((4 * kgt) | (q4 >> 1))) << 3);
  const int ztr1 = O_ZT + (jz + 4) * ZTR + (((4 * (gi & 1) + p4) ^ ((4 * kgt) | (q4 >> 1) | 2)) << 3);
  const int c16w = 4 * ch + 2 * (gi & 1) + (p4 >> 1);
  const int w1r0 = O_W1 + jz * RW + ((c16w ^ (((q4 >> 1) << 2) | (2 * kgt))) << 4) + 8 * (p4 & 1);
  const int w1r1 = O_W1 + (jz + 4) * RW + ((c16w ^ (((q4 >> 1) << 2) | (2 * kgt) | 1)) << 4) + 8 * (p4 & 1);
  const int padd = (4 * kg * 64 + 32 * ch + j) * 4 + (kq & 1) * 8192;       // + ((e & 3) + 8 (e >> 2)) * 256
  // epilogue role: row 4 w + err, channels 4 ecq .. + 3 (a row = the 16 lanes of a DPP row); prologue role: tensor, row, channel octet
  const int err = lane >> 4, ecq = lane & 15;
  const int pten = wave >> 2, prow = 8 * (wave & 3) + (lane >> 3), poct = lane & 7;
  const int pst = pten * IMG + prow * RW + ((poct ^ sw16(prow)) << 4);
  const unsigned eo = (unsigned)((4 * wave + err) * 256 + ecq * 16);
  // keep bits: this lane hashes rows 8 (j & 3) + 4 kg + i (i = 0..3) of the group of four units it belongs to
  const unsigned hash_lane = (unsigned)((8 * (j & 3) + 4 * kg) * 64 + 8 * wave + (j >> 2)) * 0x9E3779B1u;

  float4 raw0, raw1; float2 rst;                          // the next tile's rows (prologue role)
  auto load_raw = [&](long m0) {
    const long avail = a.M - m0 < 32 ? a.M - m0 : 32;
    const __amdgpu_buffer_rsrc_t Rr = make_rsrc_((pten == 0 ? a.X : a.dY) + m0 * 64, avail > 0 ? (unsigned)(avail * 256) : 0u);
    const __amdgpu_buffer_rsrc_t Sr = make_rsrc_(a.stats + m0 * 2, avail > 0 ? (unsigned)(avail * 8) : 0u);
    raw0 = buf_load4_(Rr, (unsigned)(prow * 256 + poct * 32));
    raw1 = buf_load4_(Rr, (unsigned)(prow * 256 + poct * 32 + 16));
    rst = buf_load2_(Sr, (unsigned)(prow * 8));
  };
  // rows -> LayerNorm (X waves) / dropout mask (dY waves) -> x[8] (zeros past the workgroup's rows)
  auto prologue_math = [&](long m0, float (&x)[8]) {
    asm volatile("" : "+v"(raw0.x), "+v"(raw0.y), "+v"(raw0.z), "+v"(raw0.w), "+v"(raw1.x), "+v"(raw1.y), "+v"(raw1.z), "+v"(raw1.w), "+v"(rst.x), "+v"(rst.y));
    const long m = m0 + prow;
    const bool ok = m < mend;
    if (pten == 0) {
      const float4 g0 = *reinterpret_cast<const float4*>(gbs + 8 * poct), g1 = *reinterpret_cast<const float4*>(gbs + 8 * poct + 4);
      const float4 t0 = *reinterpret_cast<const float4*>(gbs + 64 + 8 * poct), t1 = *reinterpret_cast<const float4*>(gbs + 64 + 8 * poct + 4);
      const float mean = rst.x, rstd = ok ? rst.y : 0.f;
      x[0] = (raw0.x - mean) * rstd * g0.x + t0.x; x[1] = (raw0.y - mean) * rstd * g0.y + t0.y;
      x[2] = (raw0.z - mean) * rstd * g0.z + t0.z; x[3] = (raw0.w - mean) * rstd * g0.w + t0.w;
      x[4] = (raw1.x - mean) * rstd * g1.x + t1.x; x[5] = (raw1.y - mean) * rstd * g1.y + t1.y;
      x[6] = (raw1.z - mean) * rstd * g1.z + t1.z; x[7] = (raw1.w - mean) * rstd * g1.w + t1.w;
#pragma unroll
      for (int e = 0; e < 8; ++e) x[e] = ok ? x[e] : 0.f;
    } else {
      float4 d0 = make_float4(1.f, 1.f, 1.f, 1.f), d1 = d0;
      if (DR) {
        d0 = drop_scale4(a.seed_o, (unsigned)(m * 64 + 8 * poct), thr, inv_keep);
        d1 = drop_scale4(a.seed_o, (unsigned)(m * 64 + 8 * poct + 4), thr, inv_keep);
      }
      const float sc = ok ? s_dy : 0.f;                   // (the fp16 scale rides on the mask; db2 un-scales once at the end)
      x[0] = raw0.x * d0.x * sc; x[1] = raw0.y * d0.y * sc; x[2] = raw0.z * d0.z * sc; x[3] = raw0.w * d0.w * sc;
      x[4] = raw1.x * d1.x * sc; x[5] = raw1.y * d1.y * sc; x[6] = raw1.z * d1.z * sc; x[7] = raw1.w * d1.w * sc;
      float4 c0 = *reinterpret_cast<const float4*>(dbc), c1 = *reinterpret_cast<const float4*>(dbc + 4);
      c0.x += x[0]; c0.y += x[1]; c0.z += x[2]; c0.w += x[3]; c1.x += x[4]; c1.y += x[5]; c1.z += x[6]; c1.w += x[7];
      *reinterpret_cast<float4*>(dbc) = c0; *reinterpret_cast<float4*>(dbc + 4) = c1;
    }
  };
  auto prologue_store = [&](float (&x)[8], int buf) {
    bf16x8 o[2];
    split_planes8_h(x, pten == 0 ? s_in : 1.0f, o);
    unsigned char* p = sm + O_ROWS + buf * 2 * IMG + pst;
    *reinterpret_cast<bf16x8*>(p) = o[0];
    *reinterpret_cast<bf16x8*>(p + PL) = o[1];
  };
  // ---- LayerNorm backward of row 4 w + err, channels 4 ecq .. + 3 of the tile at m0 (rows_ok = 0: nothing is stored or summed) ----
  float4 ex = make_float4(0.f, 0.f, 0.f, 0.f), ey = ex, er2 = ex;
  float2 est = make_float2(0.f, 0.f);
  auto epilogue = [&](long m0, long rows_ok) {
    // (the loaded operands are pinned HERE: the compiler otherwise hoists the arithmetic that needs only them above the barrier, right
    // behind the loads -- and waits for the loads there, inside the matrix segment that was meant to cover their latency)
    asm volatile("" : "+v"(ex.x), "+v"(ex.y), "+v"(ex.z), "+v"(ex.w), "+v"(est.x), "+v"(est.y));
    asm volatile("" : "+v"(ey.x), "+v"(ey.y), "+v"(ey.z), "+v"(ey.w), "+v"(er2.x), "+v"(er2.y), "+v"(er2.z), "+v"(er2.w));
    const int rl = 4 * wave + err;
    const bool ok = rl < rows_ok;
    const float* pp = patch + rl * 64 + 4 * ecq;
    const float4 p0 = *reinterpret_cast<const float4*>(pp), p1 = *reinterpret_cast<const float4*>(pp + 2048);
    const float dv[4] = {p0.x + p1.x, p0.y + p1.y, p0.z + p1.z, p0.w + p1.w};
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
    buf_store4_(make_rsrc_(a.dX + m0 * 64, (unsigned)(rows_ok * 256)), eo, make_float4(o4[0], o4[1], o4[2], o4[3]));
    const float om = fmaxf(fmaxf(fabsf(o4[0]), fabsf(o4[1])), fmaxf(fabsf(o4[2]), fabsf(o4[3])));
    xmax = fmaxf(xmax, ok ? om : 0.f);
    // fold the wave's four rows (lane bits 4, 5); lane (err, ecq) keeps the total of channel 4 ecq + err
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float sg = xor32_sum_(xor16_sum_(ag[e])), sb = xor32_sum_(xor16_sum_(ab[e]));
      agk += err == e ? sg : 0.f;
      abk += err == e ? sb : 0.f;
    }
  };

  // ============================ the four segments of a tile (see the schedule below) ============================
  f32x16 ah, ad;
  unsigned kw[4] = {0x1111u, 0x1111u, 0x1111u, 0x1111u};       // keep bits of (row e, unit j): word[q] bit 4 i = row 8 q + 4 kg + i
  unsigned sh_[4][2], sl_[4][2], zh_[4][2], zl_[4][2];         // S, dZ of the tile as fragment words
  // S1 (matrix): H = LN W1^T, dP = dY W2s.  ALL sixteen fragment reads are issued before the first product: the partner wave of the
  // SIMD is in a vector segment, nobody else covers the LDS latency of a read-then-use sequence (measured: 2.5 - 3.9 K cycles for
  // these 24 products with the reads next to their uses)
  auto seg_products = [&](const unsigned char* rows) {
    bf16x8 lf[4][2], yf[4][2], W1b[4][2];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const unsigned char* p = rows + fragA + (x16 ^ (32 * ks)), *pw = sm + O_W1 + (32 * wave + j) * RW + (x16 ^ (32 * ks));
      lf[ks][0] = *reinterpret_cast<const bf16x8*>(p); lf[ks][1] = *reinterpret_cast<const bf16x8*>(p + PL);
      W1b[ks][0] = *reinterpret_cast<const bf16x8*>(pw); W1b[ks][1] = *reinterpret_cast<const bf16x8*>(pw + W1PL);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int e = 0; e < 16; ++e) { ah[e] = 0.f; ad[e] = 0.f; }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {                      // (the dY fragments are requested under the H products, as the LN fragments retire)
      const unsigned char* p = rows + IMG + fragA + (x16 ^ (32 * ks));
      yf[ks][0] = *reinterpret_cast<const bf16x8*>(p); yf[ks][1] = *reinterpret_cast<const bf16x8*>(p + PL);
      ah = mfma32_<true>(lf[ks][1], W1b[ks][0], ah);
      ah = mfma32_<true>(lf[ks][0], W1b[ks][1], ah);
      ah = mfma32_<true>(lf[ks][0], W1b[ks][0], ah);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      ad = mfma32_<true>(yf[ks][1], W2b[ks][0], ad);
      ad = mfma32_<true>(yf[ks][0], W2b[ks][1], ad);
      ad = mfma32_<true>(yf[ks][0], W2b[ks][0], ad);
    }
  };
  // S2 (vector): S = Swish(H) mask, dZ = dP mask Swish'(H) -> fragment words; dZ -> the wave's slice of ZT
  auto seg_chain = [&]() {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float sv[4], zv[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int i = 4 * q + e;
        const float h = fmaf(ah[i], uh, bias);
        float sg = sigmoidf_(h);
        // a dropped unit: sigmoid := 0 makes Swish AND Swish' vanish (one select for both)
        if (DR) sg = __uint_as_float(__float_as_uint(sg) & (unsigned)__builtin_amdgcn_sbfe((int)kw[q], 4 * e, 1));
        const float s0 = h * sg;
        const float sw = fmaf(s0, 1.0f - sg, sg);             // Swish'(h) = sg (1 + h (1 - sg))
        sv[e] = s0 * mkS;
        zv[e] = ad[i] * sw * mkZ;                             // (mkZ carries the un-scaling of the accumulator AND dZ's scale)
        bs1 += zv[e];
      }
      split4_(sv[0], sv[1], sv[2], sv[3], one, sh_[q][0], sh_[q][1], sl_[q][0], sl_[q][1]);
      split4_(zv[0], zv[1], zv[2], zv[3], one, zh_[q][0], zh_[q][1], zl_[q][0], zl_[q][1]);
      *reinterpret_cast<u32x2_*>(sm + (ztw ^ (16 * q))) = (u32x2_){zh_[q][0], zh_[q][1]};
      *reinterpret_cast<u32x2_*>(sm + (ztw ^ (16 * q)) + ZTPL) = (u32x2_){zl_[q][0], zl_[q][1]};
    }
  };
  // S3 (matrix): dW2 += dY^T S, dW1 += dZ^T LN (contraction over the tile's 32 rows: two 16-deep steps, the transposed operand's rows in
  // the C layout's order); dLN[32 rows x channels 32 ch ..] over the hidden units 64 kq .. + 63 (the ZT slices of this wave and of its
  // neighbour in the same group) -> partial patch: the first group stores, the second adds to what the first left a segment ago
  auto seg_gradients = [&](const unsigned char* rows, auto second, auto&& mid) {
    // 16 units of two transposed fragments (hi, lo of one operand): 0 - 3 dY^T of (ks, nt) for dW2, 4 - 7 LN of (ks, nt) for dW1, 8 - 15
    // (ZT, W1) of the four 16-deep steps of dLN; unit u + 2 is requested before the products of unit u (a ring of four)
    bf16x8 ring[4][2];
    auto fetch = [&](int u) {
      bf16x8 (&f)[2] = ring[u & 3];
      if (u < 8) {
        const int ks = (u >> 1) & 1, nt = u & 1, o = u < 4 ? IMG : 0;
        const unsigned char* p0 = rows + o + ((trA0 ^ (64 * nt)) + 16 * RW * ks), *p1 = rows + o + ((trA1 ^ (64 * nt)) + 16 * RW * ks);
        f[0] = tr2_(p0, p1); f[1] = tr2_(p0 + PL, p1 + PL);
      } else {
        const int ks = (u - 8) >> 1;
        if (u & 1) { f[0] = tr2_(sm + w1r0 + 16 * ks * RW, sm + w1r1 + 16 * ks * RW); f[1] = tr2_(sm + w1r0 + W1PL + 16 * ks * RW, sm + w1r1 + W1PL + 16 * ks * RW); }
        else { f[0] = tr2_(sm + ztr0 + 16 * ks * ZTR, sm + ztr1 + 16 * ks * ZTR); f[1] = tr2_(sm + ztr0 + ZTPL + 16 * ks * ZTR, sm + ztr1 + ZTPL + 16 * ks * ZTR); }
      }
    };
    f32x16 gl;
    fetch(0); fetch(1);
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if (u + 2 < 16) fetch(u + 2);
      __builtin_amdgcn_sched_barrier(0);
      const bf16x8 (&f)[2] = ring[u & 3];
      if (u < 4) {
        const int ks = u >> 1, nt = u & 1;
        const bf16x8 s_h = __builtin_bit_cast(bf16x8, (u32x4_){sh_[2 * ks][0], sh_[2 * ks][1], sh_[2 * ks + 1][0], sh_[2 * ks + 1][1]});
        const bf16x8 s_l = __builtin_bit_cast(bf16x8, (u32x4_){sl_[2 * ks][0], sl_[2 * ks][1], sl_[2 * ks + 1][0], sl_[2 * ks + 1][1]});
        aw2[nt] = mfma32_<true>(f[0], s_l, aw2[nt]);      // dW2[c][j]: A = dY^T (channel on the lane), B = S (unit on the lane)
        aw2[nt] = mfma32_<true>(f[1], s_h, aw2[nt]);
        aw2[nt] = mfma32_<true>(f[0], s_h, aw2[nt]);
      } else if (u < 8) {
        const int ks = (u >> 1) & 1, nt = u & 1;
        const bf16x8 z_h = __builtin_bit_cast(bf16x8, (u32x4_){zh_[2 * ks][0], zh_[2 * ks][1], zh_[2 * ks + 1][0], zh_[2 * ks + 1][1]});
        const bf16x8 z_l = __builtin_bit_cast(bf16x8, (u32x4_){zl_[2 * ks][0], zl_[2 * ks][1], zl_[2 * ks + 1][0], zl_[2 * ks + 1][1]});
        aw1[nt] = mfma32_<true>(z_h, f[1], aw1[nt]);      // dW1[j][c]: A = dZ^T (unit on the lane), B = LN (channel on the lane)
        aw1[nt] = mfma32_<true>(z_l, f[0], aw1[nt]);
        aw1[nt] = mfma32_<true>(z_h, f[0], aw1[nt]);
      } else if (u & 1) {
        const bf16x8 (&z)[2] = ring[(u - 1) & 3];
        if (u == 9) {
          mid();
#pragma unroll
          for (int e = 0; e < 16; ++e) gl[e] = 0.f;
        }
        gl = mfma32_<true>(z[0], f[1], gl);
        gl = mfma32_<true>(z[1], f[0], gl);
        gl = mfma32_<true>(z[0], f[0], gl);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      float* pc = reinterpret_cast<float*>(sm + O_PATCH + padd + ((e & 3) + 8 * (e >> 2)) * 256);
      if (decltype(second)::value) *pc += gl[e] * u2; else *pc = gl[e] * u2;
    }
  };
  // S4 (vector): keep bits of the tile at m0
  auto seg_keep_bits = [&](long m0) {
    if (!DR) return;
    const unsigned base = hash_lane + (unsigned)(m0 * 64) * 0x9E3779B1u;
    unsigned w = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) w |= drop_keep4_pre(a.seed_h, base + (unsigned)(i * 64) * 0x9E3779B1u, thr) << (4 * i);
    const int sh = j & 3;
    kw[0] = (unsigned)__builtin_amdgcn_mov_dpp((int)w, 0x00, 0xf, 0xf, true) >> sh;
    kw[1] = (unsigned)__builtin_amdgcn_mov_dpp((int)w, 0x55, 0xf, 0xf, true) >> sh;
    kw[2] = (unsigned)__builtin_amdgcn_mov_dpp((int)w, 0xAA, 0xf, 0xf, true) >> sh;
    kw[3] = (unsigned)__builtin_amdgcn_mov_dpp((int)w, 0xFF, 0xf, 0xf, true) >> sh;
  };
  auto seg_prologue = [&](long m0, int buf) {
    float x[8];
    prologue_math(m0, x);
    prologue_store(x, buf);
  };
  auto load_epi = [&](long m0) {                         // the epilogue's operands of the tile at m0 (X, dY: L2 hits -- the prologue read them)
    const long avail = a.M - m0 < 32 ? a.M - m0 : 32;
    ex = buf_load4_(make_rsrc_(a.X + m0 * 64, (unsigned)(avail * 256)), eo);
    ey = buf_load4_(make_rsrc_(a.dY + m0 * 64, (unsigned)(avail * 256)), eo);
    est = buf_load2_(make_rsrc_(a.stats + m0 * 2, (unsigned)(avail * 8)), (unsigned)((4 * wave + err) * 8));
  };
  auto load_r2 = [&](long m0) {                          // (first touch: requested a segment earlier than the rest)
    const long avail = a.M - m0 < 32 ? a.M - m0 : 32;
    er2 = buf_load4_(make_rsrc_(a.dR2 ? a.dR2 + m0 * 64 : a.X, a.dR2 ? (unsigned)(avail * 256) : 0u), eo);
  };
  auto rows_of = [&](int t) { return sm + O_ROWS + (t & 1) * 2 * IMG; };
  auto rows_in = [&](int t) { const long m0 = mbeg + 32L * t; return mend - m0 < 32 ? mend - m0 : 32L; };

  // ============================ schedule: the two waves of a SIMD (w, w + 4) are always in segments of different kinds ============================
  //   slot      waves 0-3 (units 0..127, X rows)                          waves 4-7 (units 128..255, dY rows)
  //   1(t)      S1(t)  products                           [matrix]        epilogue(t-1), keep bits(t), dY rows(t+1) -> images   [vector]
  //   2(t)      S2(t)  chain; epilogue(t-1)               [vector]        S1(t) products                                        [matrix]
  //   3(t)      S3(t)  weight gradients, dLN -> patches   [matrix]        S2(t) chain                                           [vector]
  //   4(t)      keep bits(t+1), X rows(t+1) -> images     [vector]        S3(t) weight gradients, dLN += patches                [matrix]
  // One barrier per slot.  A tile's row images are read in slots 1 - 4 of the tile and written in slots 1 / 4 of the tile before (two
  // buffers); a group's ZT slices are written in its chain and read in its next segment; the patches of tile t are complete after slot
  // 4(t), read in slots 1 - 2 of t + 1 and rewritten from slot 3(t + 1).
  load_raw(mbeg);
  __syncthreads();                                       // gamma / beta staged (W1 image, db2 zeros)
  seg_prologue(mbeg, 0);
  if (wave >= 4) load_raw(mbeg + 32);
  seg_keep_bits(mbeg);
  __syncthreads();                                       // images of tile 0
  if (wave < 4) {
    for (int t = 0; t < ntile; ++t) {
      const long m0 = mbeg + 32L * t;
      seg_products(rows_of(t));
      if (t > 0) { load_epi(m0 - 32); }
      FF_SYNC(0);
      load_raw(m0 + 32);                                 // X rows of the next tile (used in slot 4; past the last tile: zeros from the range check)
      seg_chain();
      if (t > 0) epilogue(m0 - 32, rows_in(t - 1));
      FF_SYNC(1);
      seg_gradients(rows_of(t), std::false_type{}, [&]() { load_r2(m0); });
      FF_SYNC(2);
      seg_keep_bits(m0 + 32);
      seg_prologue(m0 + 32, (t + 1) & 1);
      FF_SYNC(3);
    }
    load_epi(mbeg + 32L * (ntile - 1));
    __syncthreads();                                     // (slot 1 of the tile after the last: the other group's last patches are complete)
    epilogue(mbeg + 32L * (ntile - 1), rows_in(ntile - 1));
  } else {
    for (int t = 0; t < ntile; ++t) {
      const long m0 = mbeg + 32L * t;
      if (t > 0) { epilogue(m0 - 32, rows_in(t - 1)); seg_keep_bits(m0); }
      seg_prologue(m0 + 32, (t + 1) & 1);                // dY rows of the next tile (requested two slots ago)
      FF_SYNC(0);
      seg_products(rows_of(t));
      load_raw(m0 + 64);                                 // dY rows of the tile after the next (used in slot 1 of the next tile)
      FF_SYNC(1);
      seg_chain();
      load_r2(m0);
      FF_SYNC(2);
      seg_gradients(rows_of(t), std::true_type{}, [&]() { load_epi(m0); });
      FF_SYNC(3);
    }
    __syncthreads();
    epilogue(mbeg + 32L * (ntile - 1), rows_in(ntile - 1));
  }
  // ---- leave: dX maximum, LayerNorm parameter gradients, weight gradients ----
  if (a.out_amax) {
    xmax = wave_max(xmax);
    if (lane == 0) amax_raise_(a.out_amax, xmax);
  }
  atomicAdd(&a.dgamma[4 * ecq + err], agk);
  atomicAdd(&a.dbeta[4 * ecq + err], abk);
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
  if (pten == 1 && a.db2) {                              // the dY waves' column sums: fold the 8 rows of a wave (lane bits 3 .. 5), one atomic per wave and channel
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float v = dbc[e];
      v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
      if (lane < 8) atomicAdd(&a.db2[8 * poct + e], v * ub2);
    }
  }
#ifdef SE_FF_STAMPS
  if (a.stamps && tid == 0) { a.stamps[4096 + blockIdx.x] = (unsigned)(__builtin_amdgcn_s_memtime() - t00_); a.stamps[4096 + 512 + blockIdx.x] = (unsigned)(tp_ - t00_); }
  if (a.stamps && blockIdx.x < 4 && lane == 0) {
    for (int k = 0; k < 4; ++k) { a.stamps[((int)blockIdx.x * 8 + wave) * 16 + k] = wk_[k]; a.stamps[((int)blockIdx.x * 8 + wave) * 16 + 4 + k] = wt_[k]; }
    a.stamps[((int)blockIdx.x * 8 + wave) * 16 + 8] = (unsigned)ntile;
  }
#endif
}

int se_ff_fused4_launch(const FfFusedArgs& a0, int ncu, void* stream) {
  FfFusedArgs a = a0;
  // one persistent 8-wave workgroup per CU; rows dealt in multiples of the 32-row tile; at least 8 tiles per workgroup so that the
  // 32 768 atomics a workgroup leaves with are amortised
  long rpw = (a.M + ncu - 1) / ncu;
  if (rpw < 256) rpw = 256;
  rpw = (rpw + 31) / 32 * 32;
  a.rows_per_wg = rpw;
  const int nwg = (int)((a.M + rpw - 1) / rpw);
  if (a.drop_p > 0.f) hipLaunchKernelGGL(ff_bwd_fused4_kernel<true>, dim3((unsigned)nwg), dim3(512), 0, as_stream(stream), a);
  else hipLaunchKernelGGL(ff_bwd_fused4_kernel<false>, dim3((unsigned)nwg), dim3(512), 0, as_stream(stream), a);
  return se_check_launch("se_ff_bwd_fused");
}
