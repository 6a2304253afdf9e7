// Fused backward of the Conformer feed-forward module, weight gradients included -- v3: SYMMETRIC waves.
//
// se_ff_fused.hip (v2) specialises the waves of a workgroup (four compute the input-gradient chain, four the weight gradients); its
// barrier-arrival timeline (tools/ff_fused_stamps.py, profiles/r05_ff_fused_v2_stamps.txt) shows what that costs: the elementwise
// chain of a slot runs on ONE wave per SIMD (vector issue at 4+ cycles per instruction while the partner wave has little vector
// work), ten barriers per 64 rows, and in most intervals one kind of wave waits for the other: 26 000 cycles per 64-row tile for
// 7 700 cycles of matrix work.  Here every wave does the same work on its own 32 HIDDEN UNITS:
//   one 8-wave workgroup per CU, rows in tiles of 32, all 256 hidden units at once (wave w: units 32 w .. 32 w + 31), scaled
//   split-fp16 arithmetic (se_gemm_dev.h, precision 3).  Per tile and wave:
//   phase 1  H^T = W1 LN(X)^T + b1, dP^T = W2s (mask_o dY)^T  [32 units x 32 rows]: A = the wave's weight rows, REGISTER-resident for
//            the whole launch (no weight block ever goes through LDS), B = the tile's rows out of the row images; H first, the dP
//            products run under the elementwise work on H; S = Swish(H) mask_h and dZ = dP mask_h Swish'(H) -> fp16 (hi, lo) ->
//            ROW-major exchange images [32 rows][256 units]                                                      | barrier Q
//   phase 2  dW1[j][c] += sum_r dZ[r][j] LN[r][c], dW2[c][j] += sum_r dY[r][c] S[r][j] for the wave's units: four 32 x 32 tiles, both
//            operands by hardware-transposed reads of the row-major images (64 accumulator registers per wave, kept for the whole
//            launch); db1 / db2 by packed dot products on the same fragments;
//            dLN[32 rows x 32 channels] over 64 hidden units per wave (wave = channel half x hidden quarter): A = dZ rows out of the
//            exchange image, B = W1^T straight from memory (L1 / L2 resident) -> four partial patches                  | barrier R
//   phase 3  LayerNorm backward of four rows per wave from the summed patches -> dX, gamma / beta gradients; the NEXT tile's rows
//            (requested during phase 2) -> LayerNorm / dropout mask -> fp16 row images; the next tile's dropout keep bits   | barrier P
// Three barriers per 32 rows, no role ever waits for another, the vector work of a tile is spread over all eight waves.
#include "se_ff_fused.h"

namespace ff3 {
constexpr int RS = 144;                 // row stride (bytes) of the LN / dY images: 64 fp16 + 16 B pad
constexpr int PL = 32 * RS, IMG = 2 * PL;
constexpr int ZRS = 544;                // row stride of the exchange images: 256 fp16 + 32 B pad (16-byte aligned rows: ds_read_b128)
constexpr int ZPL = 32 * ZRS, ZIMG = 2 * ZPL;
constexpr int O_LN = 0, O_DY = IMG, O_Z = 2 * IMG, O_S = O_Z + ZIMG, O_PATCH = O_S + ZIMG;      // patch: [4 quarters][32 rows][64 ch] fp32
constexpr int O_B1 = O_PATCH + 4 * 32 * 64 * 4, O_GB = O_B1 + 1024, O_BITS = O_GB + 512, LDS_BYTES = O_BITS + 32 * 32;
}  // namespace ff3

__global__ __launch_bounds__(512, 2) void ff_bwd_fused3_kernel(FfFusedArgs a) {
  using namespace ff3;
  using fff::tr8_; using fff::trfrag_; using fff::trfrag_sum_; using fff::split4_;
  __shared__ __attribute__((aligned(16))) unsigned char sm[LDS_BYTES];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const long mbeg = (long)blockIdx.x * a.rows_per_wg;
  long mend = mbeg + a.rows_per_wg;
  if (mend > a.M) mend = a.M;
  if (mbeg >= mend) return;                              // (whole workgroup: block-uniform)
  const int ntile = (int)((mend - mbeg + 31) / 32);
  f16_clamp_mode_();
  // ---- scales ----
  const float inv_keep = drop_inv_keep(a.drop_p);
  const unsigned thr = drop_thr(a.drop_p);
  const bool dr = a.drop_p > 0.f;
  float s_in, s_dy, uh, u2, un1, un2, ub1, ub2, mkS, mkZ;
  {
    const float dy_amax = __builtin_nontemporal_load(a.dy_amax), w2_amax = __builtin_nontemporal_load(a.w2t_amax);
    const int e_dy = f16_sexp_(dy_amax), e_w2 = f16_sexp_(w2_amax), e_w1 = f16_sexp_(__builtin_nontemporal_load(a.w1_amax));
    const int e_w1t = f16_sexp_(__builtin_nontemporal_load(a.w1t_amax));
    const int e_in = operand_sexp_(a.in_amax, a.ln_sexp), e_mid = operand_sexp_(a.mid_amax, a.hid_sexp);
    // |dZ| <= amax(dY) inv_keep^2 64 amax(W2s) 1.1 (64 terms, |Swish'| < 1.1): a few binades loose, as in ff_bwd_kernel
    const int e_dz = f16_sexp_(dy_amax * inv_keep * inv_keep * 64.f * w2_amax * 1.1f);
    s_in = exp2i_(e_in); s_dy = exp2i_(e_dy);
    uh = exp2i_(-e_in - e_w1);
    mkS = inv_keep * exp2i_(e_mid);
    mkZ = inv_keep * exp2i_(-e_dy - e_w2 + e_dz);         // accumulator of dP -> dZ at its fp16 scale
    u2 = exp2i_(-e_dz - e_w1t);
    un1 = exp2i_(-e_dz - e_in); un2 = a.alpha * exp2i_(-e_mid - e_dy);
    ub1 = exp2i_(-e_dz); ub2 = a.alpha * exp2i_(-e_dy);
  }
  float one;
  asm volatile("s_mov_b32 %0, 1.0" : "=s"(one));
  if (tid < 256) reinterpret_cast<float*>(sm + O_B1)[tid] = a.b1[tid];
  if (tid < 128) reinterpret_cast<float*>(sm + O_GB)[tid] = tid < 64 ? a.gamma[tid] : a.beta[tid - 64];
  const float* b1s = reinterpret_cast<const float*>(sm + O_B1);
  const float* gbs = reinterpret_cast<const float*>(sm + O_GB);
  float* const patch = reinterpret_cast<float*>(sm + O_PATCH);

  const int r = lane & 31, kg = lane >> 5;                // phase 1: row / k group; C layout column = r
  const int gi = lane >> 4, q4 = (lane & 15) >> 2, p4 = lane & 3;
  // ---- this wave's weight rows as A fragments, for the whole launch: lane (unit 32 w + (lane & 31), kg) holds channels 16 ks + 8 kg .. ----
  bf16x8 W1a[4][2], W2a[4][2];
  {
    const size_t wpl = (size_t)256 * 64;
    const __bf16* p1 = reinterpret_cast<const __bf16*>(a.W1) + (size_t)(32 * wave + r) * 64 + 8 * kg;
    const __bf16* p2 = reinterpret_cast<const __bf16*>(a.W2T) + (size_t)(32 * wave + r) * 64 + 8 * kg;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) {
        W1a[ks][pl] = *reinterpret_cast<const bf16x8*>(p1 + pl * wpl + 16 * ks);
        W2a[ks][pl] = *reinterpret_cast<const bf16x8*>(p2 + pl * wpl + 16 * ks);
      }
  }
  // weight-gradient accumulators of the wave's 32 hidden units: dW1[j][c] (two channel halves), dW2[c][j] (two channel halves)
  f32x16 aw1[2], aw2[2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int e = 0; e < 16; ++e) { aw1[nt][e] = 0.f; aw2[nt][e] = 0.f; }
  float bs1 = 0.f, bs2a = 0.f, bs2b = 0.f;               // db1 of unit (lane & 31) (this lane's rows); wave 0: db2 of channel (lane & 31) (+ 32)
  float agk = 0.f, abk = 0.f, xmax = 0.f;                 // gamma / beta gradients of channel 4 ecq + err (see phase 3)

  // ---- per-lane LDS addresses ----
  const unsigned char* const fragLN = sm + O_LN + r * RS + 16 * kg;                    // + pl * PL + 32 * ks  (B fragments: row r)
  const unsigned char* const fragDY = sm + O_DY + r * RS + 16 * kg;
  unsigned char* const zsw = sm + O_Z + r * ZRS + (32 * wave + 4 * kg) * 2;            // + 16 q: quad q of row r; S image at + ZIMG
  const unsigned char* const bitw = sm + O_BITS + r * 32 + 4 * wave;                   // keep bits of row r, the wave's 8 groups
  // transposed reads: contraction index = image row.  Z / S: column = unit 32 w + (lane & 31); LN / dY: column = channel 32 nt + (lane & 31)
  const unsigned char* const trZ = sm + O_Z + (8 * (gi >> 1) + q4) * ZRS + (32 * wave + 16 * (gi & 1) + 4 * p4) * 2;   // + pl * ZPL + 16 ks * ZRS
  const unsigned char* const trL = sm + O_LN + (8 * (gi >> 1) + q4) * RS + (16 * (gi & 1) + 4 * p4) * 2;              // + pl * PL + 16 ks * RS + 64 nt
  // dLN role: channel half ch, hidden quarter kq
  const int ch = wave & 1, kq = wave >> 1;
  const unsigned char* const zrd = sm + O_Z + r * ZRS + (64 * kq + 8 * kg) * 2;        // + pl * ZPL + 32 * ks: dZ rows as A fragments
  const __bf16* const w1tp = reinterpret_cast<const __bf16*>(a.W1T) + (size_t)(32 * ch + r) * 256 + 64 * kq + 8 * kg;   // + pl * 64 * 256 + 16 ks
  // phase 3 roles
  const int err = lane >> 4, ecq = lane & 15;             // epilogue: row 4 w + err, channels 4 ecq .. + 3
  const int pten = wave >> 2, prow = 8 * (wave & 3) + (lane >> 3), poct = lane & 7;      // prologue: tensor, row, channel octet
  const int brow = tid >> 4, bpart = tid & 15;            // keep bits: row, groups 4 bpart .. + 3

  float4 raw0, raw1; float2 rst;                          // the next tile's rows (prologue role)
  auto load_raw = [&](long m0) {
    const long avail = a.M - m0 < 32 ? a.M - m0 : 32;
    const __amdgpu_buffer_rsrc_t Rr = make_rsrc_((pten == 0 ? a.X : a.dY) + m0 * 64, avail > 0 ? (unsigned)(avail * 256) : 0u);
    const __amdgpu_buffer_rsrc_t Sr = make_rsrc_(a.stats + m0 * 2, avail > 0 ? (unsigned)(avail * 8) : 0u);
    raw0 = buf_load4_(Rr, (unsigned)(prow * 256 + poct * 32));
    raw1 = buf_load4_(Rr, (unsigned)(prow * 256 + poct * 32 + 16));
    rst = buf_load2_(Sr, (unsigned)(prow * 8));
  };
  auto prologue_store = [&](long m0) {
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
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = ok ? x[e] : 0.f;
    bf16x8 o[2];
    split_planes8_h(x, pten == 0 ? s_in : s_dy, o);
    unsigned char* p = sm + (pten == 0 ? O_LN : O_DY) + prow * RS + 16 * poct;
    *reinterpret_cast<bf16x8*>(p) = o[0];
    *reinterpret_cast<bf16x8*>(p + PL) = o[1];
  };
  auto mask_bits = [&](long m0) {
    if (!dr || (a.dbg & 2)) return;
    const unsigned grp0 = (unsigned)((m0 + brow) * 64 + 4 * bpart);          // (m * 256 + 16 bpart) >> 2
    unsigned bits = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      unsigned f[4];
      drop_fields(a.seed_h, grp0 + k, f);
#pragma unroll
      for (int e = 0; e < 4; ++e) bits |= (f[e] >= thr ? 1u : 0u) << (4 * k + e);
    }
    *reinterpret_cast<unsigned short*>(sm + O_BITS + brow * 32 + 2 * bpart) = (unsigned short)bits;
  };

  load_raw(mbeg);
  __syncthreads();                                       // b1 / gamma / beta staged
  prologue_store(mbeg);
  mask_bits(mbeg);
  __syncthreads();                                       // (P) images and keep bits of tile 0
  for (int t = 0; t < ntile; ++t) {
    const long m0 = mbeg + 32L * t;
    const bool more = t + 1 < ntile;
    // ===================================== phase 1 =====================================
    f32x16 ah, ad;
#pragma unroll
    for (int e = 0; e < 16; ++e) { ah[e] = 0.f; ad[e] = 0.f; }
    if (!(a.dbg & 4)) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const bf16x8 bh = *reinterpret_cast<const bf16x8*>(fragLN + 32 * ks), bl = *reinterpret_cast<const bf16x8*>(fragLN + PL + 32 * ks);
        ah = mfma32_<true>(W1a[ks][0], bl, ah);
        ah = mfma32_<true>(W1a[ks][1], bh, ah);
        ah = mfma32_<true>(W1a[ks][0], bh, ah);
      }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const bf16x8 bh = *reinterpret_cast<const bf16x8*>(fragDY + 32 * ks), bl = *reinterpret_cast<const bf16x8*>(fragDY + PL + 32 * ks);
        ad = mfma32_<true>(W2a[ks][0], bl, ad);
        ad = mfma32_<true>(W2a[ks][1], bh, ad);
        ad = mfma32_<true>(W2a[ks][0], bh, ad);
      }
    }
    unsigned mbits = 0xffffffffu;
    if (dr) mbits = *reinterpret_cast<const unsigned*>(bitw) >> (4 * kg);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 b4 = *reinterpret_cast<const float4*>(b1s + 32 * wave + 8 * q + 4 * kg);
      const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
      float sv[4], zv[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int i = 4 * q + e;
        const float h = fmaf(ah[i], uh, bb[e]);
        const float sg = (a.dbg & 1) ? 0.5f : sigmoidf_(h);
        const float s0 = h * sg;
        const float sw = fmaf(s0, 1.0f - sg, sg);             // Swish'(h) = sg (1 + h (1 - sg))
        const bool keep = (mbits >> (8 * q + e)) & 1u;
        sv[e] = keep ? s0 * mkS : 0.f;
        zv[e] = keep ? ad[i] * sw * mkZ : 0.f;
      }
      unsigned zh0, zh1, zl0, zl1, sh0, sh1, sl0, sl1;
      split4_(sv[0], sv[1], sv[2], sv[3], one, sh0, sh1, sl0, sl1);
      split4_(zv[0], zv[1], zv[2], zv[3], one, zh0, zh1, zl0, zl1);
      *reinterpret_cast<u32x2_*>(zsw + 16 * q) = (u32x2_){zh0, zh1};
      *reinterpret_cast<u32x2_*>(zsw + ZPL + 16 * q) = (u32x2_){zl0, zl1};
      *reinterpret_cast<u32x2_*>(zsw + ZIMG + 16 * q) = (u32x2_){sh0, sh1};
      *reinterpret_cast<u32x2_*>(zsw + ZIMG + ZPL + 16 * q) = (u32x2_){sl0, sl1};
    }
    __syncthreads();                                     // (Q) exchange images complete
    // ===================================== phase 2 =====================================
    if (!(a.dbg & 128)) {
      // ---- weight gradients of the wave's units: contraction over the tile's 32 rows (two 16-deep steps) ----
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const bf16x8 z_h = trfrag_sum_<ZRS>(trZ + 16 * ks * ZRS, bs1), z_l = trfrag_sum_<ZRS>(trZ + ZPL + 16 * ks * ZRS, bs1);
        const bf16x8 s_h = trfrag_<ZRS>(trZ + ZIMG + 16 * ks * ZRS), s_l = trfrag_<ZRS>(trZ + ZIMG + ZPL + 16 * ks * ZRS);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          const bf16x8 l_h = trfrag_<RS>(trL + 16 * ks * RS + 64 * nt), l_l = trfrag_<RS>(trL + PL + 16 * ks * RS + 64 * nt);
          aw1[nt] = mfma32_<true>(z_h, l_l, aw1[nt]);     // dW1[j][c]: A = dZ^T (unit on the lane), B = LN (channel on the lane)
          aw1[nt] = mfma32_<true>(z_l, l_h, aw1[nt]);
          aw1[nt] = mfma32_<true>(z_h, l_h, aw1[nt]);
          bf16x8 y_h, y_l;
          if (wave == 0) {                                 // db2 = column sums of mask_o dY (one wave is enough)
            float& acc = nt ? bs2b : bs2a;
            y_h = trfrag_sum_<RS>(trL + (O_DY - O_LN) + 16 * ks * RS + 64 * nt, acc);
            y_l = trfrag_sum_<RS>(trL + (O_DY - O_LN) + PL + 16 * ks * RS + 64 * nt, acc);
          } else {
            y_h = trfrag_<RS>(trL + (O_DY - O_LN) + 16 * ks * RS + 64 * nt);
            y_l = trfrag_<RS>(trL + (O_DY - O_LN) + PL + 16 * ks * RS + 64 * nt);
          }
          aw2[nt] = mfma32_<true>(y_h, s_l, aw2[nt]);     // dW2[c][j]: A = dY^T (channel on the lane), B = S (unit on the lane)
          aw2[nt] = mfma32_<true>(y_l, s_h, aw2[nt]);
          aw2[nt] = mfma32_<true>(y_h, s_h, aw2[nt]);
        }
      }
    }
    // requested now (behind the weight-gradient tiles: next to their fragments the loads' registers spilled), used in phase 3: the epilogue's operands (this tile) and the next tile's rows
    float4 ex, ey, er2 = make_float4(0.f, 0.f, 0.f, 0.f);
    float2 est;
    {
      const long avail = a.M - m0 < 32 ? a.M - m0 : 32;
      const unsigned eo = (unsigned)((4 * wave + err) * 256 + ecq * 16);
      const __amdgpu_buffer_rsrc_t Xr = make_rsrc_(a.X + m0 * 64, (unsigned)(avail * 256)), Yr = make_rsrc_(a.dY + m0 * 64, (unsigned)(avail * 256));
      const __amdgpu_buffer_rsrc_t Sr = make_rsrc_(a.stats + m0 * 2, (unsigned)(avail * 8));
      ex = buf_load4_(Xr, eo); ey = buf_load4_(Yr, eo);
      if (a.dR2) er2 = buf_load4_(make_rsrc_(a.dR2 + m0 * 64, (unsigned)(avail * 256)), eo);
      est = buf_load2_(Sr, (unsigned)((4 * wave + err) * 8));
    }
    if (more) load_raw(m0 + 32);
    // ---- dLN[32 rows x channels 32 ch ..] over the hidden units 64 kq .. 64 kq + 63 ----
    {
      f32x16 gl;
#pragma unroll
      for (int e = 0; e < 16; ++e) gl[e] = 0.f;
      if (!(a.dbg & 8)) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const bf16x8 zh = *reinterpret_cast<const bf16x8*>(zrd + 32 * ks), zl = *reinterpret_cast<const bf16x8*>(zrd + ZPL + 32 * ks);
          const bf16x8 wh = *reinterpret_cast<const bf16x8*>(w1tp + 16 * ks), wl = *reinterpret_cast<const bf16x8*>(w1tp + (size_t)64 * 256 + 16 * ks);
          gl = mfma32_<true>(zh, wl, gl);
          gl = mfma32_<true>(zl, wh, gl);
          gl = mfma32_<true>(zh, wh, gl);
        }
      }
      float* P = patch + kq * (32 * 64) + 32 * ch + r;    // C layout: row = (e & 3) + 8 (e >> 2) + 4 kg, column = lane & 31
#pragma unroll
      for (int e = 0; e < 16; ++e) P[((e & 3) + 8 * (e >> 2) + 4 * kg) * 64] = gl[e] * u2;
    }
    __syncthreads();                                     // (R) patches complete; row and exchange images free
    // ===================================== phase 3 =====================================
    {
      // LayerNorm backward of row 4 w + err, channels 4 ecq .. + 3 (a row = 16 lanes)
      const int rl = 4 * wave + err;
      const bool ok = m0 + rl < mend;
      float dv[4];
      {
        const float* pp = patch + rl * 64 + 4 * ecq;
        const float4 p0 = *reinterpret_cast<const float4*>(pp), p1 = *reinterpret_cast<const float4*>(pp + 2048);
        const float4 p2 = *reinterpret_cast<const float4*>(pp + 4096), p3 = *reinterpret_cast<const float4*>(pp + 6144);
        dv[0] = (p0.x + p1.x) + (p2.x + p3.x); dv[1] = (p0.y + p1.y) + (p2.y + p3.y);
        dv[2] = (p0.z + p1.z) + (p2.z + p3.z); dv[3] = (p0.w + p1.w) + (p2.w + p3.w);
      }
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
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
      s1 *= (1.f / 64.f); s2 *= (1.f / 64.f);
      float o4[4] = {ey.x + er2.x, ey.y + er2.y, ey.z + er2.z, ey.w + er2.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) o4[e] += rstd * (dxh[e] - s1 - xh[e] * s2);
      const long rows_ok = mend - m0 < 32 ? mend - m0 : 32;
      buf_store4_(make_rsrc_(a.dX + m0 * 64, (unsigned)(rows_ok * 256)), (unsigned)(rl * 256 + ecq * 16), make_float4(o4[0], o4[1], o4[2], o4[3]));
      if (ok) xmax = fmaxf(fmaxf(xmax, fmaxf(fabsf(o4[0]), fabsf(o4[1]))), fmaxf(fabsf(o4[2]), fabsf(o4[3])));
      // fold the wave's four rows (lane bits 4, 5); lane (err, ecq) keeps the total of channel 4 ecq + err
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float sg = ag[e], sb = ab[e];
        sg += __shfl_xor(sg, 16, 64); sb += __shfl_xor(sb, 16, 64);
        sg += __shfl_xor(sg, 32, 64); sb += __shfl_xor(sb, 32, 64);
        agk += err == e ? sg : 0.f;
        abk += err == e ? sb : 0.f;
      }
    }
    if (more) {
      prologue_store(m0 + 32);
      mask_bits(m0 + 32);
    }
    __syncthreads();                                     // (P)
  }
  // ---- leave: dX maximum, LayerNorm parameter gradients, weight gradients ----
  if (a.out_amax) {
    xmax = wave_max(xmax);
    if (lane == 0) amax_raise_(a.out_amax, xmax);
  }
  atomicAdd(&a.dgamma[4 * ecq + err], agk);
  atomicAdd(&a.dbeta[4 * ecq + err], abk);
  const int col = lane & 31;
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = (e & 3) + 8 * (e >> 2) + 4 * kg;
      atomicAdd(&a.dW1[(32 * wave + row) * 64 + 32 * nt + col], aw1[nt][e] * un1);
      atomicAdd(&a.dW2[(32 * nt + row) * 256 + 32 * wave + col], aw2[nt][e] * un2);
    }
  {
    const float v = bs1 + __shfl_xor(bs1, 32, 64);
    if (kg == 0) atomicAdd(&a.db1[32 * wave + col], v * ub1);
  }
  if (wave == 0 && a.db2) {
    const float va = bs2a + __shfl_xor(bs2a, 32, 64), vb = bs2b + __shfl_xor(bs2b, 32, 64);
    if (kg == 0) { atomicAdd(&a.db2[col], va * ub2); atomicAdd(&a.db2[32 + col], vb * ub2); }
  }
}

int se_ff_fused3_launch(const FfFusedArgs& a0, int ncu, void* stream) {
  FfFusedArgs a = a0;
  // one persistent 8-wave workgroup per CU; rows dealt in multiples of the 32-row tile; at least 8 tiles per workgroup so that the
  // 32 768 atomics a workgroup leaves with are amortised
  long rpw = (a.M + ncu - 1) / ncu;
  if (rpw < 256) rpw = 256;
  rpw = (rpw + 31) / 32 * 32;
  a.rows_per_wg = rpw;
  const int nwg = (int)((a.M + rpw - 1) / rpw);
  hipLaunchKernelGGL(ff_bwd_fused3_kernel, dim3((unsigned)nwg), dim3(512), 0, as_stream(stream), a);
  return se_check_launch("se_ff_bwd_fused (v3)");
}
