// LayerNorm-backward GEMM and the weight gradient of the same layer in ONE sweep over the rows (round 5).
//
//   y = LN(x) W^T + b  (the qkv projection: K = 192 outputs; the pointwise-GLU convolution of the conv module: K = 256), backward:
//     dX = dR + LNbwd(A W)                    A = dL/dy [M, K]                  (se_gemm_ln_bwd: read A, x, dR; write dX)
//     dW[k][c] += sum_r A[r][k] LN(x)[r][c],  db[k] += sum_r A[r][k]            (se_gemm_tap_wgrad: read A and x AGAIN)
//   (conformer.py:87-89, 160-164 backwards).  Both contractions read the same rows; here a tile of 32 rows of A and x goes to LDS
//   once (fp16 (hi, lo) images, scaled split-fp16 arithmetic of se_gemm_dev.h) and feeds both: A 1 KB + x, dR, dX 0.75 KB per row
//   instead of 2 x (A + x) + dR + dX.
//
// One persistent 8-wave workgroup per CU, every wave the same work, ONE barrier per 32-row tile (second form of the round: the
// first kept the [64 x K] weight planes in LDS -- 70 KB -- and needed two barriers per tile because the images were single; it ran at
// ~10 K cycles per tile against ~1.5 K cycles of matrix work and 4 K of vector work).  Now every wave holds ITS block of W^T (32
// channels x 64 outputs x 2 planes = 32 VGPRs) for the whole launch, and the freed LDS double-buffers the images and the patches:
//   stage      the NEXT tile's rows (requested two tiles ago) -> fp16 images [(t + 1) & 1]; rows of tile t + 3 requested;
//   products   dW: wave w owns the 32 outputs k = 32 w .. (two 32 x 32 tiles over the channel halves, accumulators kept for the whole
//              launch), both operands by hardware-transposed reads of the row-major images [t & 1] (the contraction index is the image
//              row); db by packed dot products on the same fragments;
//              dLN[32 rows x 32 channels] over 64 of the K outputs per wave (wave = channel half x K part): A rows out of the image
//              (ds_read_b128), W^T from registers -> partial patches [t & 1]                                                   | barrier
//   epilogue   LayerNorm backward of four rows per wave on the summed patches -> dX (x, dR kept from the tile's staging: the lane that
//              staged a row segment finishes it) -- no barrier behind it: a fast wave's next stage / products touch the OTHER buffers.
// The kernel is HBM-bound by construction (1.75 KB per row; 24 matrix instructions per wave and tile).
#include "se_ff_fused.h"

struct LnBwdFusedArgs {
  const float* A; const float* WT;              // A [M][K] fp32; WT = W^T as scaled fp16 planes [2][64][K]
  const float* X; const float* stats; const float* gamma; const float* beta; const float* dR;
  float* dX; float* dgamma; float* dbeta; float* dW; float* dbias;        // dW [K][64], dbias [K] (may be NULL): accumulated
  long M; long rows_per_wg;
  const float* a_amax; const float* w_amax; const float* in_amax; float* out_amax; int ln_sexp;
};

template <int K>
__global__ __launch_bounds__(512, 2) void lnbwd_fused_kernel(LnBwdFusedArgs a) {
  using fff::trfrag_; using fff::trfrag_sum_; using fff::split4_;
  constexpr int RS = 144, LPL = 32 * RS;                 // LN image [32 rows][64 ch]
  constexpr int ARS = 2 * K + 32, APL = 32 * ARS;        // A image [32 rows][K]: + 32 B pad, rows 16-byte aligned
  constexpr int NQ = K / 64;                             // K parts of the dLN product = partial patches
  constexpr int NKB = K / 32;                            // 32-wide output blocks of dW (<= 8: one per wave)
  constexpr int PSZ = NQ * 32 * 64 * 4;                  // one set of partial patches
  constexpr int O_A = 0, O_LN = 4 * APL, O_PATCH = O_LN + 4 * LPL, O_GB = O_PATCH + 2 * PSZ;      // images / patches: [2 buffers]
  constexpr int LDS_BYTES = O_GB + 512;
  static_assert(LDS_BYTES <= 163840 && NKB <= 8, "one workgroup per CU");
  __shared__ __attribute__((aligned(16))) unsigned char sm[LDS_BYTES];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const long mbeg = (long)blockIdx.x * a.rows_per_wg;
  long mend = mbeg + a.rows_per_wg;
  if (mend > a.M) mend = a.M;
  if (mbeg >= mend) return;
  const int ntile = (int)((mend - mbeg + 31) / 32);
  f16_clamp_mode_();
  const int e_a = f16_sexp_(__builtin_nontemporal_load(a.a_amax)), e_w = f16_sexp_(__builtin_nontemporal_load(a.w_amax));
  const int e_in = operand_sexp_(a.in_amax, a.ln_sexp);
  const float s_a = exp2i_(e_a), s_in = exp2i_(e_in), u_ln = exp2i_(-e_a - e_w), u_w = exp2i_(-e_a - e_in), u_b = exp2i_(-e_a);
  float one;
  asm volatile("s_mov_b32 %0, 1.0" : "=s"(one));
  // ---- resident: gamma | beta ----
  if (tid < 128) reinterpret_cast<float*>(sm + O_GB)[tid] = tid < 64 ? a.gamma[tid] : a.beta[tid - 64];
  const float* gbs = reinterpret_cast<const float*>(sm + O_GB);
  float* const patch = reinterpret_cast<float*>(sm + O_PATCH);
  const int r = lane & 31, kg = lane >> 5;
  const int gi = lane >> 4, q4 = (lane & 15) >> 2, p4 = lane & 3;
  // staging / epilogue role: row srow of the tile, channels 4 scq .. + 3 of x / dR / dX; A chunks (scq + 16 i) of that row
  const int srow = tid >> 4, scq = tid & 15;
  constexpr int NA = K / 64;                             // float4 of A per lane and tile (a row = K / 4 float4 over 16 lanes)
  // weight-gradient role: output block kb = wave (k = 32 kb ..), both channel halves
  const bool wg_on = wave < NKB;
  const unsigned char* const trA = sm + O_A + (8 * (gi >> 1) + q4) * ARS + (32 * wave + 16 * (gi & 1) + 4 * p4) * 2;   // + pl * APL + 16 ks * ARS
  const unsigned char* const trL = sm + O_LN + (8 * (gi >> 1) + q4) * RS + (16 * (gi & 1) + 4 * p4) * 2;               // + pl * LPL + 16 ks * RS + 64 nt
  // dLN role: channel half ch, K part kq (64 outputs)
  const int ch = wave & 1, kq = wave >> 1;
  const bool dl_on = kq < NQ;
  const unsigned char* const arow = sm + O_A + r * ARS + (64 * (dl_on ? kq : 0) + 8 * kg) * 2;             // + buffer * 2 APL + pl * APL + 32 ks
  // this wave's block of W^T (channel 32 ch + r on the lane, outputs 64 kq + 8 kg + 32 ks ..): registers for the whole launch
  bf16x8 wfr[4][2];
  {
    const __bf16* wp = reinterpret_cast<const __bf16*>(a.WT) + (size_t)(32 * ch + r) * K + 64 * (dl_on ? kq : 0) + 8 * kg;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) wfr[ks][pl] = *reinterpret_cast<const bf16x8*>(wp + (size_t)pl * 64 * K + 16 * ks);
  }

  f32x16 aw[2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int e = 0; e < 16; ++e) aw[nt][e] = 0.f;
  float bsum = 0.f, agk = 0.f, abk = 0.f, xmax = 0.f;

  // rows in flight: TWO tiles ahead (one tile of products + epilogue, ~4 000 cycles, is shorter than the memory latency under load: with
  // one tile in flight the launch ran at 3.1 TB/s of its 0.93 GB) -- buffer (tile & 1); 52 registers per lane, 116 KB per CU
  float4 pa2[2][NA], px2[2], pr2[2];
  float2 pst2[2];
  pr2[0] = pr2[1] = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 kx, kr, nx, nr; float2 kst, nst;                         // kept for the epilogue: (k*) the tile whose patches are summed, (n*) the tile just staged
  auto load_rows = [&](long m0, auto bufc) {
    constexpr int bf = decltype(bufc)::value;
    float4 (&pa)[NA] = pa2[bf]; float4& px = px2[bf]; float4& pr = pr2[bf];
    const long avail = mend - m0 < 32 ? mend - m0 : 32;      // (<= 0 past this workgroup's rows: empty descriptors, zeros, no traffic)
    const __amdgpu_buffer_rsrc_t Ar = make_rsrc_(a.A + m0 * K, avail > 0 ? (unsigned)(avail * K * 4) : 0u);
    const __amdgpu_buffer_rsrc_t Xr = make_rsrc_(a.X + m0 * 64, avail > 0 ? (unsigned)(avail * 256) : 0u);
    const __amdgpu_buffer_rsrc_t Sr = make_rsrc_(a.stats + m0 * 2, avail > 0 ? (unsigned)(avail * 8) : 0u);
#pragma unroll
    for (int i = 0; i < NA; ++i) pa[i] = buf_load4_(Ar, (unsigned)(srow * K * 4 + (scq + 16 * i) * 16));
    px = buf_load4_(Xr, (unsigned)(srow * 256 + scq * 16));
    // (no residual: an empty descriptor -> zeros.  Every load of this function is UNCONDITIONAL: a load under `if` ends in register
    // copies at the join and the compiler waits for ALL loads in flight there -- the prefetch was worth nothing)
    pr = buf_load4_(make_rsrc_(a.dR ? a.dR + m0 * 64 : a.X, (a.dR && avail > 0) ? (unsigned)(avail * 256) : 0u), (unsigned)(srow * 256 + scq * 16));
    pst2[bf] = buf_load2_(Sr, (unsigned)(srow * 8));
  };
  auto stage = [&](long m0, auto bufc, auto imgc) {        // prefetch buffer bf -> images [ib]
    constexpr int bf = decltype(bufc)::value, ib = decltype(imgc)::value;
    float4 (&pa)[NA] = pa2[bf]; float4& px = px2[bf]; float4& pr = pr2[bf];
    const float2 pst = pst2[bf];
    const bool ok = m0 + srow < mend;                    // (rows past M arrive as zeros from the range check; rows past mend are zeroed here)
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      unsigned h0, h1, l0, l1;
      const float4 v = pa[i];
#ifdef LNB_ABL_NO_SPLIT
      h0 = __float_as_uint(v.x); h1 = __float_as_uint(v.y); l0 = __float_as_uint(v.z); l1 = __float_as_uint(v.w);
#else
      split4_(ok ? v.x * s_a : 0.f, ok ? v.y * s_a : 0.f, ok ? v.z * s_a : 0.f, ok ? v.w * s_a : 0.f, one, h0, h1, l0, l1);
#endif
      unsigned char* p = sm + O_A + ib * 2 * APL + srow * ARS + (scq + 16 * i) * 8;
      *reinterpret_cast<u32x2_*>(p) = (u32x2_){h0, h1};
      *reinterpret_cast<u32x2_*>(p + APL) = (u32x2_){l0, l1};
    }
    const float4 gm = *reinterpret_cast<const float4*>(gbs + 4 * scq), bt = *reinterpret_cast<const float4*>(gbs + 64 + 4 * scq);
    const float mean = pst.x, rstd = pst.y;
    unsigned h0, h1, l0, l1;
    split4_(ok ? ((px.x - mean) * rstd * gm.x + bt.x) * s_in : 0.f, ok ? ((px.y - mean) * rstd * gm.y + bt.y) * s_in : 0.f,
            ok ? ((px.z - mean) * rstd * gm.z + bt.z) * s_in : 0.f, ok ? ((px.w - mean) * rstd * gm.w + bt.w) * s_in : 0.f, one, h0, h1, l0, l1);
    unsigned char* p = sm + O_LN + ib * 2 * LPL + srow * RS + scq * 8;
    *reinterpret_cast<u32x2_*>(p) = (u32x2_){h0, h1};
    *reinterpret_cast<u32x2_*>(p + LPL) = (u32x2_){l0, l1};
    nx = px; nr = pr; nst = pst;
  };

  using B0 = std::integral_constant<int, 0>; using B1 = std::integral_constant<int, 1>;
  load_rows(mbeg, B0{});
  load_rows(mbeg + 32, B1{});                            // (rows past M: empty descriptors, no traffic)
  __syncthreads();                                       // gamma / beta
  stage(mbeg, B0{}, B0{});
  load_rows(mbeg + 64, B0{});
  kx = nx; kr = nr; kst = nst;
  __syncthreads();
  auto tile = [&](int t, auto bufc) {                    // bufc = t & 1: image / patch buffer of this tile
    constexpr int bf = decltype(bufc)::value;
    using OB = std::integral_constant<int, 1 - bf>;
    const long m0 = mbeg + 32L * t;
    // ===================================== next tile's images, then this tile's products =====================================
    stage(m0 + 32, OB{}, OB{});                           // (tile t + 1 sits in prefetch buffer (t + 1) & 1; past the last tile: zeros
    load_rows(m0 + 96, OB{});                             //  into images nobody reads -- straight-line code, see load_rows)
#ifndef LNB_ABL_NO_DW
    if (wg_on) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const bf16x8 a_h = trfrag_sum_<ARS>(trA + bf * 2 * APL + 16 * ks * ARS, bsum),
                     a_l = trfrag_sum_<ARS>(trA + bf * 2 * APL + APL + 16 * ks * ARS, bsum);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          const bf16x8 l_h = trfrag_<RS>(trL + bf * 2 * LPL + 16 * ks * RS + 64 * nt), l_l = trfrag_<RS>(trL + bf * 2 * LPL + LPL + 16 * ks * RS + 64 * nt);
          aw[nt] = mfma32_<true>(a_h, l_l, aw[nt]);       // dW[k][c]: A = A^T (output k on the lane), B = LN (channel on the lane)
          aw[nt] = mfma32_<true>(a_l, l_h, aw[nt]);
          aw[nt] = mfma32_<true>(a_h, l_h, aw[nt]);
        }
      }
    }
#endif
#ifndef LNB_ABL_NO_DLN
    if (dl_on) {
      f32x16 gl;
#pragma unroll
      for (int e = 0; e < 16; ++e) gl[e] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const bf16x8 ah = *reinterpret_cast<const bf16x8*>(arow + bf * 2 * APL + 32 * ks), al = *reinterpret_cast<const bf16x8*>(arow + bf * 2 * APL + APL + 32 * ks);
        gl = mfma32_<true>(ah, wfr[ks][1], gl);
        gl = mfma32_<true>(al, wfr[ks][0], gl);
        gl = mfma32_<true>(ah, wfr[ks][0], gl);
      }
      float* P = patch + bf * (PSZ / 4) + kq * (32 * 64) + 32 * ch + r;    // C layout: row = (e & 3) + 8 (e >> 2) + 4 kg, column = lane & 31
#pragma unroll
      for (int e = 0; e < 16; ++e) P[((e & 3) + 8 * (e >> 2) + 4 * kg) * 64] = gl[e] * u_ln;
    }
#endif
    __syncthreads();                                     // patches [bf] complete, images [1 - bf] staged
    // ===================================== epilogue of tile t =====================================
#ifdef LNB_ABL_NO_EPI
    {
      const long rows_ok = mend - m0 < 32 ? mend - m0 : 32;
      buf_store4_(make_rsrc_(a.dX + m0 * 64, (unsigned)(rows_ok * 256)), (unsigned)(srow * 256 + scq * 16), make_float4(kx.x + kr.x, kx.y + kr.y, kst.x, kst.y));
    }
#else
    {
      const bool ok = m0 + srow < mend;
      float dv[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const float4 p4v = *reinterpret_cast<const float4*>(patch + bf * (PSZ / 4) + q * 2048 + srow * 64 + 4 * scq);
        dv[0] += p4v.x; dv[1] += p4v.y; dv[2] += p4v.z; dv[3] += p4v.w;
      }
      const float4 gm = *reinterpret_cast<const float4*>(gbs + 4 * scq);
      const float gl4[4] = {gm.x, gm.y, gm.z, gm.w}, xs[4] = {kx.x, kx.y, kx.z, kx.w};
      const float mean = kst.x, rstd = kst.y;
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
      float o4[4] = {kr.x, kr.y, kr.z, kr.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) o4[e] += rstd * (dxh[e] - s1 - xh[e] * s2);
      const long rows_ok = mend - m0 < 32 ? mend - m0 : 32;
      buf_store4_(make_rsrc_(a.dX + m0 * 64, (unsigned)(rows_ok * 256)), (unsigned)(srow * 256 + scq * 16), make_float4(o4[0], o4[1], o4[2], o4[3]));
      if (ok) xmax = fmaxf(fmaxf(xmax, fmaxf(fabsf(o4[0]), fabsf(o4[1]))), fmaxf(fabsf(o4[2]), fabsf(o4[3])));
      // fold the wave's four rows (lane bits 4, 5); lane (row srow & 3, scq) keeps the total of channel 4 scq + (srow & 3)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float sg = ag[e], sb = ab[e];
        sg = xor16_sum_(sg); sb = xor16_sum_(sb);
        sg = xor32_sum_(sg); sb = xor32_sum_(sb);
        agk += (srow & 3) == e ? sg : 0.f;
        abk += (srow & 3) == e ? sb : 0.f;
      }
    }
#endif
    kx = nx; kr = nr; kst = nst;
  };
  for (int t = 0; t < ntile; t += 2) {
    tile(t, B0{});
    if (t + 1 < ntile) tile(t + 1, B1{});
  }
  xmax = wave_max(xmax);
  // gamma / beta gradients (and the dX maximum): folded across the eight waves in LDS, ONE atomic instruction per vector and workgroup (atomics of different
  // workgroups to the same cache lines are serialised at the memory side, ~24 ns per wave instruction -- tools/micro/atomic_line_bench.hip:
  // sixteen instructions per workgroup on these two vectors were 2 x 49 us at the end of a 230 - 260 us launch)
  __syncthreads();                                       // (the images are free: fold area)
  {
    float* fold = reinterpret_cast<float*>(sm + O_A);
    fold[wave * 128 + 4 * scq + (srow & 3)] = agk;
    fold[wave * 128 + 64 + 4 * scq + (srow & 3)] = abk;
    if (lane == 0) fold[1024 + wave] = xmax;
  }
  __syncthreads();
  if (wave == 0) {
    const float* fold = reinterpret_cast<const float*>(sm + O_A);
    float g = 0.f, bt = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) { g += fold[w * 128 + lane]; bt += fold[w * 128 + 64 + lane]; }
    atomicAdd(&a.dgamma[lane], g);
    atomicAdd(&a.dbeta[lane], bt);
    if (a.out_amax && lane == 0) {
      float m = fold[1024];
#pragma unroll
      for (int w = 1; w < 8; ++w) m = fmaxf(m, fold[1024 + w]);
      amax_raise_(a.out_amax, m);
    }
  }
  if (wg_on) {
    const int col = lane & 31;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int e = 0; e < 16; ++e) atomicAdd(&a.dW[(32 * wave + (e & 3) + 8 * (e >> 2) + 4 * kg) * 64 + 32 * nt + col], aw[nt][e] * u_w);
    if (a.dbias) {
      const float v = bsum + __shfl_xor(bsum, 32, 64);
      if (kg == 0) atomicAdd(&a.dbias[32 * wave + col], v * u_b);
    }
  }
}

extern "C" int se_gemm_ln_bwd_wgrad(const float* A, const float* WT, long M, int K, const float* X, const float* stats,
                                    const float* gamma, const float* beta, const float* dR, float* dX, float* dgamma, float* dbeta,
                                    float* dW, float* dbias, const float* a_amax, const float* w_amax, const float* in_amax,
                                    int ln_sexp, float* out_amax, void* stream) {
  SE_REQUIRE(A && WT && X && stats && gamma && beta && dX && dgamma && dbeta && dW, "gemm_ln_bwd_wgrad: null operand");
  SE_REQUIRE(a_amax && w_amax, "gemm_ln_bwd_wgrad: the operand amax scalars are required (scaled split-fp16)");
  SE_REQUIRE(M > 0 && (K == 192 || K == 256), "gemm_ln_bwd_wgrad: M=%ld K=%d (built for K = 192 (qkv) and 256 (pointwise-GLU))", M, K);
  SE_REQUIRE(((size_t)WT & 15) == 0 && ((size_t)A & 15) == 0, "gemm_ln_bwd_wgrad: A and the weight planes must be 16-byte aligned");
  const int ncu = se_cu_count();
  long rpw = (M + ncu - 1) / ncu;
  if (rpw < 128) rpw = 128;
  rpw = (rpw + 31) / 32 * 32;
  const int nwg = (int)((M + rpw - 1) / rpw);
  LnBwdFusedArgs a{A, WT, X, stats, gamma, beta, dR, dX, dgamma, dbeta, dW, dbias, M, rpw, a_amax, w_amax, in_amax, out_amax, ln_sexp};
  if (K == 256) hipLaunchKernelGGL(lnbwd_fused_kernel<256>, dim3((unsigned)nwg), dim3(512), 0, as_stream(stream), a);
  else hipLaunchKernelGGL(lnbwd_fused_kernel<192>, dim3((unsigned)nwg), dim3(512), 0, as_stream(stream), a);
  return se_check_launch("se_gemm_ln_bwd_wgrad");
}
