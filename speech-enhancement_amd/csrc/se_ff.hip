// Fused feed-forward kernels of the Conformer blocks (forward; input-gradient chain + LayerNorm backward).
#define SE_FF_EARLY_RES        // (A/B: comment out -- 3.97 -> 3.90 ms per step same-box)
#include "se_gemm_dev.h"

// ---------------------------------------------------------------------------------------------
// Fused feed-forward forward:  Y = X + alpha * Drop_o( W2 Drop_h( Swish( W1 LN(X) + b1 ) ) + b2 ),  H = W1 LN(X) + b1 kept
// for the backward (conformer.py:53-71, Scale(0.5, PreNorm(FeedForward))).  Same skeleton as the K = 64 row-panel
// kernel: every wave keeps the split LN(X) fragments of its 32 rows in registers and sweeps the hidden units in blocks
// of 64; the block of H it has just produced is written out, activated, re-split and fed -- through a wave-private LDS
// transpose -- straight back as the A operand of the second GEMM, whose 32 x 64 result stays in registers across the
// sweep.  H is written once and never re-read in the forward (unfused: + one 4 M hid-byte read and a second kernel).
struct FfArgs {
  const float* X; const float* rowstats; const float* gamma; const float* beta;
  const float* W1; const float* b1; const float* W2; const float* b2;
  float* H; float* Y; long M; int hid; float drop_p; unsigned seed_h, seed_o; float alpha;
  float* out_stats;      // optional [M][2]: (mean, rstd) of the rows of Y (the next LayerNorm's statistics)
  se_f16_scales sc;      // F16 kernels only (precision 3)
};

// WPL: W1 / W2 arrive pre-split (se_weight_prep: three bf16 planes each, 64 * hid elements apart): the weight blocks are
// plain 16-B copies into LDS instead of 8 fp32 loads + 8 three-way splits (144 VALU instructions) per thread and block.
// NB: number of 64-wide hidden blocks when known at compile time (4 for the Conformer's hid = 256; 0 = run-time loop).  With NB
// the sweep is straight-line code: at the head of a LOOP the wait-count pass has to merge the entry state (weight loads only)
// with the back edge (weight loads, then the H stores) and falls back to vmcnt(0) -- every block then waits for the write
// acknowledgements of the previous block's H stores.  Unrolled, the wait for the weights is vmcnt(#stores issued after them).
// F16 (precision 3): scaled split-fp16, two planes, three MFMAs per product (se_gemm_dev.h); needs WPL (se_weight_prep fmt 1
// planes and their amax scalars).  LN(X) is scaled by 2^in_sexp (a LayerNorm output is bounded by 7.94 |gamma| + |beta|), the
// activated hidden block by 2^mid_sexp; H (when stored) and Y are un-scaled exactly.  H == nullptr: H is not stored (the
// recomputing backward does not read it).
// STH: H is stored (compile time: behind a run-time `if (a.H)` the wait-count pass cannot count the H stores and falls back to
// draining them before every weight block -- the very thing NB avoids)
template <int NPL, bool WPL = false, int NB = 0, bool F16 = false, bool STH = true>
__global__ __launch_bounds__(256, 2) void ff_fwd_kernel(FfArgs a) {    // 2 workgroups per CU: VGPR + AGPR <= 256
  static_assert(!F16 || (WPL && NPL == 2), "the scaled split-fp16 kernels read pre-split fp16 planes");
  constexpr int SB = 72, PB = 64 * SB, SP = 36;
  // CHAIN (round 5; the default form: scaled fp16, H not stored): the hidden block never leaves the registers between the two GEMMs.
  // GEMM 1 runs TRANSPOSED (A = the W1 rows, B = the LayerNorm rows), so a lane holds 16 hidden units of ITS token row per 32-unit
  // half: (r & 3) + 8 (r >> 2) + 4 kg -- after bias / Swish / dropout exactly the eight values per k-group that a row fragment of
  // GEMM 2 needs, if GEMM 2 contracts the hidden units of a 16-block in the order [0-3, 8-11 | 4-7, 12-15]: the W2 block is staged
  // into LDS in that order (two 8-byte writes per chunk instead of one 16-byte write).  Gone: the 32 x 32 transposes through the
  // wave's LDS patch (16 ds_write_b32 + 4 ds_read_b128 per half and block) and the LDS round trip between the matrix products.
  constexpr bool CHAIN = F16 && !STH;
  __shared__ __attribute__((aligned(16))) __bf16 W1p[NPL * PB];
  __shared__ __attribute__((aligned(16))) __bf16 W2p[NPL * PB];
  __shared__ __attribute__((aligned(16))) float patch[4 * 32 * SP];    // wave-private 32 x 32 transposes
  __shared__ __attribute__((aligned(16))) float b1s[64];
  __shared__ __attribute__((aligned(16))) float gbs[128];               // LayerNorm gamma | beta
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float* cs = patch + wave * 32 * SP;
  const long m0 = (long)blockIdx.x * 128;
  const long row = m0 + wave * 32 + (lane & 31);
  const int kg = lane >> 5;
  const bool rok = row < a.M;
  const unsigned thr = drop_thr(a.drop_p);
  const float inv_keep = drop_inv_keep(a.drop_p);
  const bool dr = a.drop_p > 0.f;
  float s_in = 1.f, s_mid = 1.f, u1 = 1.f, u2 = 1.f;
  if (F16) {
    f16_clamp_mode_();
    const int e1 = operand_sexp_(a.sc.wa_amax, 0), e2 = operand_sexp_(a.sc.wb_amax, 0);
    // (round 4) in_amax / mid_amax, when given, are device scalars >= max |LN(X)| / max |Swish(H) * mask| -- proven bounds from the
    // current parameters (se_act_bounds) -- and replace the static exponents: no promise about gamma or W1 is left to break
    const int ein = operand_sexp_(a.sc.in_amax, a.sc.in_sexp), emid = operand_sexp_(a.sc.mid_amax, a.sc.mid_sexp);
    s_in = exp2i_(ein); s_mid = exp2i_(emid);
    u1 = exp2i_(-ein - e1); u2 = exp2i_(-emid - e2);
  }

  bf16x8 af1[4][NPL];
  {
    // every load of the prologue is unconditional (rows past M read row M - 1 and are zeroed by selects below): predicated
    // loads compile to divergent branches with an s_waitcnt vmcnt(0) in each -- 8 dependent memory round trips per workgroup
    const long rowl = rok ? row : a.M - 1;
    const float* __restrict__ xp = a.X + rowl * 64 + 8 * kg;
    const float2 mr = *reinterpret_cast<const float2*>(a.rowstats + 2 * rowl);
    const float mean = mr.x, rstd = mr.y;
    float4 v[4][2];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      v[ks][0] = *reinterpret_cast<const float4*>(xp + 16 * ks);
      v[ks][1] = *reinterpret_cast<const float4*>(xp + 16 * ks + 4);
    }
    // gamma / beta through LDS: fetched piecewise from global memory under register pressure they were 12 dependent L2 round trips
    if (tid < 32) *reinterpret_cast<float4*>(&gbs[4 * tid]) = *reinterpret_cast<const float4*>((tid < 16 ? a.gamma : a.beta - 64) + 4 * tid);
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      float x[8];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int c = 16 * ks + 8 * kg + 4 * h;
        const float4 gm = *reinterpret_cast<const float4*>(&gbs[c]), bt = *reinterpret_cast<const float4*>(&gbs[64 + c]);
        const float4 w = v[ks][h];
        x[4 * h] = rok ? (w.x - mean) * rstd * gm.x + bt.x : 0.f;
        x[4 * h + 1] = rok ? (w.y - mean) * rstd * gm.y + bt.y : 0.f;
        x[4 * h + 2] = rok ? (w.z - mean) * rstd * gm.z + bt.z : 0.f;
        x[4 * h + 3] = rok ? (w.w - mean) * rstd * gm.w + bt.w : 0.f;
      }
      if constexpr (F16) split_planes8_h(x, s_in, af1[ks]); else split_planes8<NPL>(x, af1[ks]);
    }
  }
  const int kq = tid & 15, r0 = tid >> 4;
  const int nb = NB > 0 ? NB : a.hid / 64;
  float4 rw1[4], rw2[4];
  const int pr = tid >> 2, pc = tid & 3;       // pre-split weights: row, 16-B chunks pc and pc + 4 of a 64 x 64 bf16 block
  const size_t wpl = (size_t)64 * (size_t)a.hid;
  f32x4 rp1[NPL * 2], rp2[NPL * 2];           // (flat register arrays: a 2-D array behind the lambda went to scratch memory)
  float b1n = 0.f;                             // bias of the next hidden block, fetched with its weights (lanes 0..63)
  auto load_w = [&](int jb) {
    b1n = a.b1[jb * 64 + (tid & 63)];
    if (WPL) {
      const __bf16* w1 = reinterpret_cast<const __bf16*>(a.W1) + ((unsigned)(jb * 64 + pr) * 64u + 8 * pc);
      const __bf16* w2 = reinterpret_cast<const __bf16*>(a.W2) + ((unsigned)pr * (unsigned)a.hid + jb * 64 + 8 * pc);
#pragma unroll
      for (int q = 0; q < NPL; ++q)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          rp1[q * 2 + h] = *reinterpret_cast<const f32x4*>(w1 + q * wpl + 32 * h);
          rp2[q * 2 + h] = *reinterpret_cast<const f32x4*>(w2 + q * wpl + 32 * h);
        }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int j = r0 + 16 * i;
        rw1[i] = *reinterpret_cast<const float4*>(a.W1 + (unsigned)(jb * 64 + j) * 64u + 4 * kq);
        rw2[i] = *reinterpret_cast<const float4*>(a.W2 + (unsigned)j * (unsigned)a.hid + jb * 64 + 4 * kq);
      }
    }
  };
  load_w(0);
  f32x16 y0, y1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { y0[r] = 0.f; y1[r] = 0.f; }
  const int frag = (lane & 31) * SB + 8 * kg;
  const int col = lane & 31, half = lane >> 5, cq = lane & 7, rr = lane >> 3;
#ifdef SE_FF_EARLY_RES
  // the residual rows in the epilogue's layout, requested right behind the prologue's read of the same lines (L1 / L2 hits) and
  // kept through the sweep (32 registers): read at the END of the workgroup they were a second HBM-side fetch of X
  float4 kept[2][4];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const long rg_ = m0 + wave * 32 + rr + 8 * i;
      kept[nt][i] = *reinterpret_cast<const float4*>(a.X + (rg_ < a.M ? rg_ : a.M - 1) * 64 + nt * 32 + cq * 4);
    }
#endif
  // H and Y leave through buffer stores whose descriptor covers the VALID rows of this workgroup's 128: a row past M is dropped
  // by the range check instead of a branch.  With branches around the stores the compiler can not count them, and the wait
  // for the next weight block (issued BEFORE them, so vmcnt(#stores) would do) became vmcnt(0): every hidden block waited for
  // the write acknowledgements of the previous one.
  const long rows_ok = a.M - m0 < 128 ? a.M - m0 : 128;
  const __amdgpu_buffer_rsrc_t Hrs = make_rsrc_(a.H + m0 * a.hid, (unsigned)(rows_ok * a.hid * 4));
  const __amdgpu_buffer_rsrc_t Yrs = make_rsrc_(a.Y + m0 * 64, (unsigned)(rows_ok * 64 * 4));
#pragma unroll
  for (int jb = 0; jb < nb; ++jb) {
    if (WPL) {
#pragma unroll
      for (int q = 0; q < NPL; ++q)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          *reinterpret_cast<f32x4*>(&W1p[q * PB + pr * SB + 8 * pc + 32 * h]) = rp1[q * 2 + h];
          if constexpr (CHAIN) {      // chunk c = pc + 4 h (hidden 8 c .. + 7) -> 16-block c >> 1, groups of four at 4 (c & 1) and 8 + 4 (c & 1)
            typedef float f32x2c_ __attribute__((ext_vector_type(2)));
            const int c = pc + 4 * h, at = 16 * (c >> 1) + 4 * (c & 1);
            const f32x4 v = rp2[q * 2 + h];
            *reinterpret_cast<f32x2c_*>(&W2p[q * PB + pr * SB + at]) = (f32x2c_){v[0], v[1]};
            *reinterpret_cast<f32x2c_*>(&W2p[q * PB + pr * SB + at + 8]) = (f32x2c_){v[2], v[3]};
          } else {
            *reinterpret_cast<f32x4*>(&W2p[q * PB + pr * SB + 8 * pc + 32 * h]) = rp2[q * 2 + h];
          }
        }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        split_store<NPL>(rw1[i], &W1p[(r0 + 16 * i) * SB + kq * 4], PB);
        split_store<NPL>(rw2[i], &W2p[(r0 + 16 * i) * SB + kq * 4], PB);
      }
    }
    if (tid < 64) b1s[tid] = b1n;
    __syncthreads();
    if (jb + 1 < nb) load_w(jb + 1);
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      bf16x8 bf0[NPL], bf1[NPL];
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl) {
        bf0[pl] = *reinterpret_cast<const bf16x8*>(&W1p[pl * PB + frag + 16 * ks]);
        bf1[pl] = *reinterpret_cast<const bf16x8*>(&W1p[pl * PB + 32 * SB + frag + 16 * ks]);
      }
#pragma unroll
      for (int ord = NPL - 1; ord >= 0; --ord)
#pragma unroll
        for (int qa = 0; qa <= ord; ++qa) {
          if constexpr (CHAIN) {      // H^T: hidden units down the accumulator rows, token rows across the lanes
            acc0 = mfma32_<F16>(bf0[ord - qa], af1[ks][qa], acc0);
            acc1 = mfma32_<F16>(bf1[ord - qa], af1[ks][qa], acc1);
          } else {
            acc0 = mfma32_<F16>(af1[ks][qa], bf0[ord - qa], acc0);
            acc1 = mfma32_<F16>(af1[ks][qa], bf1[ord - qa], acc1);
          }
        }
    }
    if constexpr (CHAIN) {
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {            // k-step 2 nt + j of GEMM 2: accumulator registers 8 j .. 8 j + 7 of this half
          const int ks = 2 * nt + j;
          float x[8];
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int q = 2 * j + h;              // quad: hidden 32 nt + 8 q + 4 kg .. + 3 of the block
            const int hu = 32 * nt + 8 * q + 4 * kg;
            const float4 b4 = *reinterpret_cast<const float4*>(&b1s[hu]);
            float4 sc = make_float4(1.f, 1.f, 1.f, 1.f);
            if (dr) sc = drop_scale4(a.seed_h, (unsigned)(row * a.hid + jb * 64 + hu), thr, inv_keep);
            const f32x16& ac = nt ? acc1 : acc0;
            x[4 * h] = swishf_(fmaf(ac[4 * q], u1, b4.x)) * sc.x;
            x[4 * h + 1] = swishf_(fmaf(ac[4 * q + 1], u1, b4.y)) * sc.y;
            x[4 * h + 2] = swishf_(fmaf(ac[4 * q + 2], u1, b4.z)) * sc.z;
            x[4 * h + 3] = swishf_(fmaf(ac[4 * q + 3], u1, b4.w)) * sc.w;
          }
          bf16x8 af2[NPL], bf0[NPL], bf1[NPL];
          split_planes8_h(x, s_mid, af2);
#pragma unroll
          for (int pl = 0; pl < NPL; ++pl) {
            bf0[pl] = *reinterpret_cast<const bf16x8*>(&W2p[pl * PB + frag + 16 * ks]);
            bf1[pl] = *reinterpret_cast<const bf16x8*>(&W2p[pl * PB + 32 * SB + frag + 16 * ks]);
          }
#pragma unroll
          for (int ord = NPL - 1; ord >= 0; --ord)
#pragma unroll
            for (int qa = 0; qa <= ord; ++qa) {
              y0 = mfma32_<F16>(af2[qa], bf0[ord - qa], y0);
              y1 = mfma32_<F16>(af2[qa], bf1[ord - qa], y1);
            }
        }
      }
      __syncthreads();
      continue;
    }
    const float bb0 = b1s[col], bb1 = b1s[32 + col];
    // per 32-column half of the block: transpose through the wave's patch, write H, re-split, second GEMM
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rl = (r & 3) + 8 * (r >> 2) + 4 * half;
        cs[rl * SP + col] = F16 ? (nt ? acc1[r] * u1 + bb1 : acc0[r] * u1 + bb0) : (nt ? acc1[r] + bb1 : acc0[r] + bb0);
      }
      if (STH) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int rl = rr + 8 * i;
          buf_store4_(Hrs, (unsigned)(((wave * 32 + rl) * a.hid + jb * 64 + nt * 32 + cq * 4) * 4),
                      *reinterpret_cast<const float4*>(&cs[rl * SP + cq * 4]));
        }
      }
#pragma unroll
      for (int k2 = 0; k2 < 2; ++k2) {
        const int ks = 2 * nt + k2;
        float x[8];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int kl = 16 * k2 + 8 * kg + 4 * h;                 // column inside this half
          const float4 pv = *reinterpret_cast<const float4*>(&cs[(lane & 31) * SP + kl]);
          float4 sc = make_float4(1.f, 1.f, 1.f, 1.f);
          if (dr) sc = drop_scale4(a.seed_h, (unsigned)(row * a.hid + jb * 64 + nt * 32 + kl), thr, inv_keep);
          x[4 * h] = swishf_(pv.x) * sc.x; x[4 * h + 1] = swishf_(pv.y) * sc.y;
          x[4 * h + 2] = swishf_(pv.z) * sc.z; x[4 * h + 3] = swishf_(pv.w) * sc.w;
        }
        bf16x8 af2[NPL], bf0[NPL], bf1[NPL];
        if constexpr (F16) split_planes8_h(x, s_mid, af2); else split_planes8<NPL>(x, af2);
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) {
          bf0[pl] = *reinterpret_cast<const bf16x8*>(&W2p[pl * PB + frag + 16 * ks]);
          bf1[pl] = *reinterpret_cast<const bf16x8*>(&W2p[pl * PB + 32 * SB + frag + 16 * ks]);
        }
#pragma unroll
        for (int ord = NPL - 1; ord >= 0; --ord)
#pragma unroll
          for (int qa = 0; qa <= ord; ++qa) {
            y0 = mfma32_<F16>(af2[qa], bf0[ord - qa], y0);
            y1 = mfma32_<F16>(af2[qa], bf1[ord - qa], y1);
          }
      }
    }
    __syncthreads();
  }
  // Y = X + alpha * Drop_o(acc + b2)
#ifndef SE_FF_EARLY_RES
  float4 kept[2][4];          // first the residual rows (all 8 loads in flight at once: they were 8 dependent round trips), then Y
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const long rg_ = m0 + wave * 32 + rr + 8 * i;
      kept[nt][i] = *reinterpret_cast<const float4*>(a.X + (rg_ < a.M ? rg_ : a.M - 1) * 64 + nt * 32 + cq * 4);
    }
#endif
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int rl = (r & 3) + 8 * (r >> 2) + 4 * half;
      cs[rl * SP + col] = (nt ? y1[r] : y0[r]) * u2;
    }
    const int n = nt * 32 + cq * 4;
    const float4 b2v = *reinterpret_cast<const float4*>(a.b2 + n);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int rl = rr + 8 * i;
      const long rg_ = m0 + wave * 32 + rl;
      const long rg = rg_ < a.M ? rg_ : a.M - 1;        // clamped: the residual load below stays unconditional
      float4 v = *reinterpret_cast<const float4*>(&cs[rl * SP + cq * 4]);
      v.x += b2v.x; v.y += b2v.y; v.z += b2v.z; v.w += b2v.w;
      if (dr) {
        const float4 d4 = drop_scale4(a.seed_o, (unsigned)(rg * 64 + n), thr, inv_keep);
        v.x *= d4.x; v.y *= d4.y; v.z *= d4.z; v.w *= d4.w;
      }
      const float4 xr = kept[nt][i];
      const float4 yo = make_float4(xr.x + a.alpha * v.x, xr.y + a.alpha * v.y, xr.z + a.alpha * v.z, xr.w + a.alpha * v.w);
      buf_store4_(Yrs, (unsigned)(((wave * 32 + rl) * 64 + n) * 4), yo);
      kept[nt][i] = yo;
    }
  }
  if (a.out_stats) {      // (mean, rstd) of the rows of Y: a row's 64 channels sit in the 8 lanes cq = 0..7 of an rr group, two passes
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const long rg = m0 + wave * 32 + rr + 8 * i;
      const bool ok = rg < a.M;
      float sm = 0.f;
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) sm += ok ? (kept[nt][i].x + kept[nt][i].y) + (kept[nt][i].z + kept[nt][i].w) : 0.f;
      sm += __shfl_xor(sm, 1, 64); sm += __shfl_xor(sm, 2, 64); sm += __shfl_xor(sm, 4, 64);
      const float mean = sm * (1.f / 64.f);
      float sq = 0.f;
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const float a0 = kept[nt][i].x - mean, a1 = kept[nt][i].y - mean, a2 = kept[nt][i].z - mean, a3 = kept[nt][i].w - mean;
        sq += ok ? (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3) : 0.f;
      }
      sq += __shfl_xor(sq, 1, 64); sq += __shfl_xor(sq, 2, 64); sq += __shfl_xor(sq, 4, 64);
      if (cq == 0 && ok) *reinterpret_cast<float2*>(a.out_stats + 2 * rg) = make_float2(mean, rsqrtf(sq * (1.f / 64.f) + 1e-5f));
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Round 5: the default forward (scaled fp16, hid = 256, H not stored) as a W-STATIONARY, barrier-free persistent kernel.
// ff_fwd_kernel re-stages the four weight blocks of both matrices for every 128 rows (128 KB of L2 -> LDS traffic and eight barriers
// per workgroup) and runs at ~200 us against ~90 us of vector work and ~46 us of matrix work.  Here ALL of W1 and W2 (two fp16
// planes each: 141 KB) sit in LDS for the whole launch, one 8-wave workgroup per CU; after the one barrier behind the staging every
// wave runs on its own: 32-row tiles, GEMM 1 transposed and the hidden block chained through registers into GEMM 2 (CHAIN above),
// GEMM 2 transposed as well -- Y^T[out][token] -- so that a lane ends with 32 of the 64 channels of ITS token: bias, dropout,
// residual, the row statistics (one v_permlane32_swap per sum) and the 16-byte stores all happen in registers.  The input channels
// are contracted in the same permuted order as the hidden units ([0-3, 8-11 | 4-7, 12-15] per 16-block), which makes the residual
// registers and the LayerNorm prologue registers THE SAME eight float4 (X is read once).  No LDS traffic besides the weight fragments.
constexpr int FW_SB1 = 72, FW_P1 = 256 * FW_SB1;       // W1 image: [256 hidden][64 channel positions], halves
constexpr int FW_SB2 = 264, FW_P2 = 64 * FW_SB2;       // W2 image: [64 outputs][256 hidden positions]
constexpr int FW_O_W2 = 2 * FW_P1 * 2, FW_O_B1 = FW_O_W2 + 2 * FW_P2 * 2, FW_O_B2 = FW_O_B1 + 1024, FW_O_GB = FW_O_B2 + 256;
constexpr int FW_LDS_BYTES = FW_O_GB + 512;

template <bool DR>       // DR: dropout on (compile time: a run-time test around every hash splits the tile's straight-line code)
__global__ __launch_bounds__(512, 2) void ff_fwd_ws_kernel(FfArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char fw_sm[];
  __bf16* W1i = reinterpret_cast<__bf16*>(fw_sm);
  __bf16* W2i = reinterpret_cast<__bf16*>(fw_sm + FW_O_W2);
  float* b1s = reinterpret_cast<float*>(fw_sm + FW_O_B1);
  float* b2s = reinterpret_cast<float*>(fw_sm + FW_O_B2);
  float* gbs = reinterpret_cast<float*>(fw_sm + FW_O_GB);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  typedef float f32x2w_ __attribute__((ext_vector_type(2)));
  // ---- staging: chunk c of a row's 16-byte chunks (8 consecutive units) -> positions 16 (c >> 1) + 4 (c & 1) and + 8 ----
  for (int i = tid; i < 2 * 256 * 8; i += 512) {        // W1 planes [2][256][64]
    const int pl = i >> 11, row = (i >> 3) & 255, c = i & 7;
    const f32x4 v = *reinterpret_cast<const f32x4*>(reinterpret_cast<const __bf16*>(a.W1) + (size_t)pl * 256 * 64 + row * 64 + 8 * c);
    __bf16* d = W1i + pl * FW_P1 + row * FW_SB1 + 16 * (c >> 1) + 4 * (c & 1);
    *reinterpret_cast<f32x2w_*>(d) = (f32x2w_){v[0], v[1]};
    *reinterpret_cast<f32x2w_*>(d + 8) = (f32x2w_){v[2], v[3]};
  }
  for (int i = tid; i < 2 * 64 * 32; i += 512) {        // W2 planes [2][64][256]
    const int pl = i >> 11, row = (i >> 5) & 63, c = i & 31;
    const f32x4 v = *reinterpret_cast<const f32x4*>(reinterpret_cast<const __bf16*>(a.W2) + (size_t)pl * 64 * 256 + row * 256 + 8 * c);
    __bf16* d = W2i + pl * FW_P2 + row * FW_SB2 + 16 * (c >> 1) + 4 * (c & 1);
    *reinterpret_cast<f32x2w_*>(d) = (f32x2w_){v[0], v[1]};
    *reinterpret_cast<f32x2w_*>(d + 8) = (f32x2w_){v[2], v[3]};
  }
  if (tid < 256) b1s[tid] = a.b1[tid];
  else if (tid < 320) b2s[tid - 256] = a.b2[tid - 256];
  else if (tid < 448) gbs[tid - 320] = tid < 384 ? a.gamma[tid - 320] : a.beta[tid - 384];
  f16_clamp_mode_();
  const int e1 = operand_sexp_(a.sc.wa_amax, 0), e2 = operand_sexp_(a.sc.wb_amax, 0);
  const int ein = operand_sexp_(a.sc.in_amax, a.sc.in_sexp), emid = operand_sexp_(a.sc.mid_amax, a.sc.mid_sexp);
  const float s_in = exp2i_(ein), s_mid = exp2i_(emid), u1 = exp2i_(-ein - e1), u2 = exp2i_(-emid - e2);
  const unsigned thr = drop_thr(a.drop_p);
  const float inv_keep = drop_inv_keep(a.drop_p);
  const float s_mid_k = DR ? s_mid * inv_keep : s_mid;      // hidden-dropout survivors: x inv_keep, folded into the split scale
  const int t = lane & 31, kg = lane >> 5;
  const __amdgpu_buffer_rsrc_t Xr = make_rsrc_(a.X, (unsigned)(a.M * 256)), Sr = make_rsrc_(a.rowstats, (unsigned)(a.M * 8)),
                               Yr = make_rsrc_(a.Y, (unsigned)(a.M * 256));
  const __amdgpu_buffer_rsrc_t Or = make_rsrc_(a.out_stats ? a.out_stats : a.Y, a.out_stats ? (unsigned)(a.M * 8) : 0u);
  __syncthreads();                                       // the only barrier: the images are read-only from here on
  const long ntile = (a.M + 31) / 32, stride = (long)gridDim.x * 8;
  // rows of a tile in the quad layout: xq[nt * 4 + q] = X[row][32 nt + 8 q + 4 kg .. + 3]
  float4 xq[8], xn[8];
  float2 st, stn;
  auto request = [&](long tile, float4 (&x)[8], float2& s2) {
    const long row = tile * 32 + t;                      // (rows past M: out-of-range offsets -> zeros)
    const unsigned off = (unsigned)(row < a.M ? row * 256 : 0xfffffff0L) + (unsigned)kg * 16u;
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = buf_load4_(Xr, row < a.M ? off + (unsigned)((i >> 2) * 128 + (i & 3) * 32) : BUF_OOB_);
    s2 = buf_load2_(Sr, row < a.M ? (unsigned)(row * 8) : BUF_OOB_);
  };
  long tile = (long)blockIdx.x * 8 + wave;
  request(tile, xq, st);
  for (; tile < ntile; tile += stride) {
    const long row = tile * 32 + t;
    const unsigned rowC = ((unsigned)row * 64u + (unsigned)kg) * 0x9E3779B1u;      // (row * 256 + 4 kg) / 4 times the hash constant
    // ---- LayerNorm -> B fragments of GEMM 1 (k-step ks = 2 nt + j: quads q = 2 j, 2 j + 1 of half nt) ----
    bf16x8 af1[4][2];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      float x[8];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int qi = (ks >> 1) * 4 + 2 * (ks & 1) + h, c = 32 * (ks >> 1) + 8 * (2 * (ks & 1) + h) + 4 * kg;
        const float4 gm = *reinterpret_cast<const float4*>(&gbs[c]), bt = *reinterpret_cast<const float4*>(&gbs[64 + c]);
        const float4 w = xq[qi];
        x[4 * h] = (w.x - st.x) * st.y * gm.x + bt.x; x[4 * h + 1] = (w.y - st.x) * st.y * gm.y + bt.y;
        x[4 * h + 2] = (w.z - st.x) * st.y * gm.z + bt.z; x[4 * h + 3] = (w.w - st.x) * st.y * gm.w + bt.w;
      }
      split_planes8_h(x, s_in, af1[ks]);
    }
    f32x16 yT[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { yT[0][r] = 0.f; yT[1][r] = 0.f; }
    // one k-step of GEMM 1 of hidden block jb -> acc (6 matrix instructions)
    auto gemm1_step = [&](int jb, int ks, f32x16 (&acc)[2]) {
      bf16x8 w0[2], w1[2];
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) {
        w0[pl] = *reinterpret_cast<const bf16x8*>(&W1i[pl * FW_P1 + (64 * jb + t) * FW_SB1 + 16 * ks + 8 * kg]);
        w1[pl] = *reinterpret_cast<const bf16x8*>(&W1i[pl * FW_P1 + (64 * jb + 32 + t) * FW_SB1 + 16 * ks + 8 * kg]);
      }
      acc[0] = mfma32_<true>(w0[1], af1[ks][0], acc[0]); acc[1] = mfma32_<true>(w1[1], af1[ks][0], acc[1]);      // smallest terms first
      acc[0] = mfma32_<true>(w0[0], af1[ks][1], acc[0]); acc[1] = mfma32_<true>(w1[0], af1[ks][1], acc[1]);
      acc[0] = mfma32_<true>(w0[0], af1[ks][0], acc[0]); acc[1] = mfma32_<true>(w1[0], af1[ks][0], acc[1]);
    };
    // bias / Swish / dropout / split of 16 hidden units (k-step 2 hh + j) of block jb and their GEMM 2 step (6 matrix instructions)
    auto chain_step = [&](int jb, int hh, int j, const f32x16 (&acc)[2]) {
      float x[8];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int q = 2 * j + h, hu = 64 * jb + 32 * hh + 8 * q + 4 * kg;      // hidden units hu .. hu + 3 of this token
        const float4 b4 = *reinterpret_cast<const float4*>(&b1s[hu]);
        // dropout: the group's hash argument (row * 64 + hu / 4) * C as a sum of the tile's row term, the lane's term and a
        // compile-time constant (mod 2^32: one add instead of a quarter-rate multiply); 1 / keep rides on the split scale
        unsigned km = 15u;
        if constexpr (DR) km = drop_keep4_pre(a.seed_h, rowC + (unsigned)(16 * jb + 8 * hh + 2 * q) * 0x9E3779B1u, thr);
        const float s0 = swishf_(fmaf(acc[hh][4 * q], u1, b4.x)), s1 = swishf_(fmaf(acc[hh][4 * q + 1], u1, b4.y));
        const float s2 = swishf_(fmaf(acc[hh][4 * q + 2], u1, b4.z)), s3 = swishf_(fmaf(acc[hh][4 * q + 3], u1, b4.w));
        x[4 * h] = (km & 1u) ? s0 : 0.f; x[4 * h + 1] = (km & 2u) ? s1 : 0.f;
        x[4 * h + 2] = (km & 4u) ? s2 : 0.f; x[4 * h + 3] = (km & 8u) ? s3 : 0.f;
      }
      bf16x8 af2[2], v0[2], v1[2];
      split_planes8_h(x, s_mid_k, af2);
      const int pos = 64 * jb + 16 * (2 * hh + j) + 8 * kg;
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) {
        v0[pl] = *reinterpret_cast<const bf16x8*>(&W2i[pl * FW_P2 + t * FW_SB2 + pos]);
        v1[pl] = *reinterpret_cast<const bf16x8*>(&W2i[pl * FW_P2 + (32 + t) * FW_SB2 + pos]);
      }
      yT[0] = mfma32_<true>(v0[1], af2[0], yT[0]); yT[1] = mfma32_<true>(v1[1], af2[0], yT[1]);
      yT[0] = mfma32_<true>(v0[0], af2[1], yT[0]); yT[1] = mfma32_<true>(v1[0], af2[1], yT[1]);
      yT[0] = mfma32_<true>(v0[0], af2[0], yT[0]); yT[1] = mfma32_<true>(v1[0], af2[0], yT[1]);
    };
    // software pipeline: k-step s of GEMM 1 of block jb + 1 sits in the instruction stream next to chain step s of block jb -- the two
    // are independent, so the in-order wave feeds the matrix pipe while its own vector instructions issue.  The memory fences keep the
    // scheduler from hoisting the fragment reads of LATER steps (everything hoisted at once: 650 spilled registers).
    f32x16 accA[2], accB[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { accA[0][r] = 0.f; accA[1][r] = 0.f; }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) gemm1_step(0, ks, accA);
#pragma unroll
    for (int jb = 0; jb < 4; ++jb) {
      f32x16 (&cur)[2] = (jb & 1) ? accB : accA;
      f32x16 (&nxt)[2] = (jb & 1) ? accA : accB;
      if (jb < 3) {
#pragma unroll
        for (int r = 0; r < 16; ++r) { nxt[0][r] = 0.f; nxt[1][r] = 0.f; }
      }
#pragma unroll
      for (int sidx = 0; sidx < 4; ++sidx) {
        asm volatile("" ::: "memory");
        if (jb < 3) gemm1_step(jb + 1, sidx, nxt);
        chain_step(jb, sidx >> 1, sidx & 1, cur);
      }
    }
    // (measured and not kept: the GEMM 2 step issued one stage later, and __builtin_amdgcn_sched_group_barrier shaping every stage
    // as one matrix instruction per ten vector instructions -- 178.6 / 176.0 us against 174.9: the kernel is not bound by the order of
    // its instructions but by what two waves per SIMD can hide; the output accumulators pinned to AccVGPRs (asm "+a": the compiler then
    // keeps 123 AGPRs next to 128 VGPRs and all 192 matrix instructions on a[..]): 184.5 against 183.6 us same box)
    // next tile's rows: requested HERE (the 64 registers of the LayerNorm fragments and of the hidden block are free again; in flight
    // during the epilogue and, on the SIMD's other wave, its products; past the end nothing is fetched)
    request(tile + stride, xn, stn);
    // ---- Y = X + alpha * Drop_o(acc + b2), row statistics: this lane holds channels 32 oh + 8 q + 4 kg + e of token t ----
    float4 yo[8];
    float sm = 0.f;
#pragma unroll
    for (int oh = 0; oh < 2; ++oh)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int n = 32 * oh + 8 * q + 4 * kg;
        const float4 b2v = *reinterpret_cast<const float4*>(&b2s[n]);
        float4 v = make_float4(fmaf(yT[oh][4 * q], u2, b2v.x), fmaf(yT[oh][4 * q + 1], u2, b2v.y), fmaf(yT[oh][4 * q + 2], u2, b2v.z),
                               fmaf(yT[oh][4 * q + 3], u2, b2v.w));
        if constexpr (DR) {
          const float4 d4 = drop_scale4(a.seed_o, (unsigned)(row * 64 + n), thr, inv_keep);
          v.x *= d4.x; v.y *= d4.y; v.z *= d4.z; v.w *= d4.w;
        }
        const float4 xr = xq[oh * 4 + q];
        const float4 o4 = make_float4(xr.x + a.alpha * v.x, xr.y + a.alpha * v.y, xr.z + a.alpha * v.z, xr.w + a.alpha * v.w);
        buf_store4_(Yr, row < a.M ? (unsigned)(row * 256 + n * 4) : BUF_OOB_, o4);
        yo[oh * 4 + q] = o4;
        sm += (o4.x + o4.y) + (o4.z + o4.w);
      }
    if (a.out_stats) {                                   // (mean, rstd) of the row of Y: own 32 channels + the partner lane's
      const float mean = xor32_sum_(sm) * (1.f / 64.f);
      float sq = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float a0 = yo[i].x - mean, a1 = yo[i].y - mean, a2 = yo[i].z - mean, a3 = yo[i].w - mean;
        sq += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
      }
      sq = xor32_sum_(sq);
      typedef unsigned u32x2o_ __attribute__((ext_vector_type(2)));
      const float2 ms = make_float2(mean, rsqrtf(sq * (1.f / 64.f) + 1e-5f));
      __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2o_, ms), Or, (row < a.M && kg == 0) ? (unsigned)(row * 8) : BUF_OOB_, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) xq[i] = xn[i];
    st = stn;
  }
}

// Fused feed-forward input-gradient chain (the two dgrad GEMMs of the same module):
//   dZ  = ((Drop_o(dY) W2s) .* Drop_h-mask .* Swish'(H))      [M, hid]   (W2s = alpha * W2, passed transposed)
//   dLN = dZ W1                                               [M, 64]    (input of the LayerNorm backward)
// dZ is written once (the weight-gradient GEMM reads it) and fed from registers / the wave's LDS patch into the second
// GEMM; unfused it was written, re-read by a second kernel, and H and dY each cost one more pass.
struct FfBwdArgs {
  const float* dY; const float* H; const float* W2T; const float* W1T;
  float* dZ; float* dLN; long M; int hid; float drop_p; unsigned seed_h, seed_o;
  // optional fused LayerNorm backward (X != nullptr): dX = dY + dR2 + LNbwd(dLN) is written instead of dLN, and the
  // gamma / beta gradients are accumulated (one atomic per channel per workgroup)
  const float* X; const float* stats; const float* gamma; const float* dR2; float* dX; float* dgamma; float* dbeta;
  se_f16_scales sc;      // F16 kernels only (precision 3)
};

// NB as in ff_fwd_kernel (straight-line sweep: exact wait counts).  No predicated global access anywhere: rows past M are clamped
// (loads) or dropped by the range check of a buffer descriptor (H loads, dZ / dX / dLN stores).
// F16 (precision 3): dY is scaled by its measured maximum (sc.in_amax, raised by the kernel that produced dY); dZ, formed in
// registers, by a BOUND of its maximum: |dZ| <= amax(dY) * inv_keep^2 * 64 amax(W2s) * 1.1 (64 terms, |Swish'| < 1.1) -- a few
// binades loose, well inside the 17 binades of full precision below an operand's maximum.  The kernel raises sc.out_amax /
// sc.mid_amax to max |dX| / max |dZ| (one atomic per wave) for the scaled kernels that read those tensors next.
template <int NPL, bool WPL = false, int NB = 0, bool F16 = false>
__global__ __launch_bounds__(256, 2) void ff_bwd_kernel(FfBwdArgs a) {
  static_assert(!F16 || (WPL && NPL == 2), "the scaled split-fp16 kernels read pre-split fp16 planes");
  constexpr int SB = 72, PB = 64 * SB, SP = 36;
  __shared__ __attribute__((aligned(16))) __bf16 Wa[NPL * PB];         // W2T block: rows = hidden units, k = channel
  __shared__ __attribute__((aligned(16))) __bf16 Wb[NPL * PB];         // W1T block: rows = channel, k = hidden units
  __shared__ __attribute__((aligned(16))) float patch[4 * 32 * SP];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float* cs = patch + wave * 32 * SP;
  const long m0 = (long)blockIdx.x * 128;
  const long row = m0 + wave * 32 + (lane & 31);
  const int kg = lane >> 5;
  const bool rok = row < a.M;
  const unsigned thr = drop_thr(a.drop_p);
  const float inv_keep = drop_inv_keep(a.drop_p);
  const bool dr = a.drop_p > 0.f;
  float s_dy = 1.f, s_dz = 1.f, u1 = 1.f, u2 = 1.f, xmax = 0.f;
  unsigned zmh = 0u;                           // packed fp16 pair: running max of |hi(dZ * s_dz)|
  if (F16) {
    f16_clamp_mode_();
    const float dy_amax = a.sc.in_amax ? __builtin_nontemporal_load(a.sc.in_amax) : exp2i_(13 - a.sc.in_sexp);
    const float wa_amax = __builtin_nontemporal_load(a.sc.wa_amax);
    const int e_dy = f16_sexp_(dy_amax), e_wa = f16_sexp_(wa_amax), e_wb = operand_sexp_(a.sc.wb_amax, 0);
    const int e_dz = f16_sexp_(dy_amax * inv_keep * inv_keep * 64.f * wa_amax * 1.1f);
    s_dy = exp2i_(e_dy); s_dz = exp2i_(e_dz); u1 = exp2i_(-e_dy - e_wa); u2 = exp2i_(-e_dz - e_wb);
  }

  bf16x8 af1[4][NPL];
  {
    const float* __restrict__ yp = a.dY + (rok ? row : a.M - 1) * 64 + 8 * kg;
    float4 v[4][2];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      v[ks][0] = *reinterpret_cast<const float4*>(yp + 16 * ks);
      v[ks][1] = *reinterpret_cast<const float4*>(yp + 16 * ks + 4);
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      float x[8];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int c = 16 * ks + 8 * kg + 4 * h;
        float4 w = v[ks][h];
        if (dr) {
          const float4 d4 = drop_scale4(a.seed_o, (unsigned)(row * 64 + c), thr, inv_keep);
          w.x *= d4.x; w.y *= d4.y; w.z *= d4.z; w.w *= d4.w;
        }
        x[4 * h] = rok ? w.x : 0.f; x[4 * h + 1] = rok ? w.y : 0.f; x[4 * h + 2] = rok ? w.z : 0.f; x[4 * h + 3] = rok ? w.w : 0.f;
      }
      if constexpr (F16) split_planes8_h(x, s_dy, af1[ks]); else split_planes8<NPL>(x, af1[ks]);
    }
  }
  const int kq = tid & 15, r0 = tid >> 4;
  const int nb = NB > 0 ? NB : a.hid / 64;
  const long rows_ok = a.M - m0 < 128 ? a.M - m0 : 128;
  const __amdgpu_buffer_rsrc_t Hrs = make_rsrc_(a.H + m0 * a.hid, (unsigned)(rows_ok * a.hid * 4));
  const __amdgpu_buffer_rsrc_t Zrs = make_rsrc_(a.dZ + m0 * a.hid, (unsigned)(rows_ok * a.hid * 4));
  // pre-split weight blocks: fetched into registers during the second GEMM of the previous block (where the H tile and the
  // first GEMM's accumulators are dead), written to LDS at the top of their own block
  const int pr = tid >> 2, pc = tid & 3;
  const size_t wpl = (size_t)64 * (size_t)a.hid;
  f32x4 va[NPL * 2], vb[NPL * 2];             // (ext-vector registers: an array of uint4 structs behind the lambda went to scratch memory)
  auto load_w = [&](int jb) {
    const __bf16* wa = reinterpret_cast<const __bf16*>(a.W2T) + ((unsigned)(jb * 64 + pr) * 64u + 8 * pc);
    const __bf16* wb = reinterpret_cast<const __bf16*>(a.W1T) + ((unsigned)pr * (unsigned)a.hid + jb * 64 + 8 * pc);
#pragma unroll
    for (int q = 0; q < NPL; ++q)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        va[q * 2 + h] = *reinterpret_cast<const f32x4*>(wa + q * wpl + 32 * h);
        vb[q * 2 + h] = *reinterpret_cast<const f32x4*>(wb + q * wpl + 32 * h);
      }
  };
  if (WPL) load_w(0);
  f32x16 g0, g1;                              // dLN accumulators (32 rows x 64 channels)
#pragma unroll
  for (int r = 0; r < 16; ++r) { g0[r] = 0.f; g1[r] = 0.f; }
  const int frag = (lane & 31) * SB + 8 * kg;
  const int col = lane & 31, half = lane >> 5, cq = lane & 7, rr = lane >> 3;
#pragma unroll
  for (int jb = 0; jb < nb; ++jb) {
    if (WPL) {       // pre-split planes (64 * hid elements apart): 16-B copies, no split
#pragma unroll
      for (int q = 0; q < NPL; ++q)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          *reinterpret_cast<f32x4*>(&Wa[q * PB + pr * SB + 8 * pc + 32 * h]) = va[q * 2 + h];
          *reinterpret_cast<f32x4*>(&Wb[q * PB + pr * SB + 8 * pc + 32 * h]) = vb[q * 2 + h];
        }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int j = r0 + 16 * i;
        const float4 wa = *reinterpret_cast<const float4*>(a.W2T + (unsigned)(jb * 64 + j) * 64u + 4 * kq);
        const float4 wb = *reinterpret_cast<const float4*>(a.W1T + (unsigned)j * (unsigned)a.hid + jb * 64 + 4 * kq);
        split_store<NPL>(wa, &Wa[j * SB + kq * 4], PB);
        split_store<NPL>(wb, &Wb[j * SB + kq * 4], PB);
      }
    }
    // pre-activations of this block for the Swish gradient: issued before the MFMAs, consumed in the epilogue
    float4 hp[8];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        hp[nt * 4 + i] = buf_load4_(Hrs, (unsigned)(((wave * 32 + rr + 8 * i) * a.hid + jb * 64 + nt * 32 + cq * 4) * 4));
      }
    __syncthreads();
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      bf16x8 bf0[NPL], bf1[NPL];
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl) {
        bf0[pl] = *reinterpret_cast<const bf16x8*>(&Wa[pl * PB + frag + 16 * ks]);
        bf1[pl] = *reinterpret_cast<const bf16x8*>(&Wa[pl * PB + 32 * SB + frag + 16 * ks]);
      }
#pragma unroll
      for (int ord = NPL - 1; ord >= 0; --ord)
#pragma unroll
        for (int qa = 0; qa <= ord; ++qa) {
          acc0 = mfma32_<F16>(af1[ks][qa], bf0[ord - qa], acc0);
          acc1 = mfma32_<F16>(af1[ks][qa], bf1[ord - qa], acc1);
        }
    }
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rl = (r & 3) + 8 * (r >> 2) + 4 * half;
        cs[rl * SP + col] = (nt ? acc1[r] : acc0[r]) * u1;
      }
      // dZ in the row-major lane layout: coalesced H / dZ accesses; the result goes back into the patch in place
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int rl = rr + 8 * i;
        const long rg = m0 + wave * 32 + rl;
        float4 v = *reinterpret_cast<const float4*>(&cs[rl * SP + cq * 4]);
        const float4 hz = hp[nt * 4 + i];
        if (dr) {
          const float4 d4 = drop_scale4(a.seed_h, (unsigned)(rg * a.hid + jb * 64 + nt * 32 + cq * 4), thr, inv_keep);
          v.x *= d4.x; v.y *= d4.y; v.z *= d4.z; v.w *= d4.w;
        }
        v.x *= swish_gradf_(hz.x); v.y *= swish_gradf_(hz.y); v.z *= swish_gradf_(hz.z); v.w *= swish_gradf_(hz.w);
        if (rg >= a.M) v = make_float4(0.f, 0.f, 0.f, 0.f);
        buf_store4_(Zrs, (unsigned)(((wave * 32 + rl) * a.hid + jb * 64 + nt * 32 + cq * 4) * 4), v);
        *reinterpret_cast<float4*>(&cs[rl * SP + cq * 4]) = v;
      }
      if (WPL && nt == 1 && jb + 1 < nb) load_w(jb + 1);
#pragma unroll
      for (int k2 = 0; k2 < 2; ++k2) {
        const int ks = 2 * nt + k2;
        float x[8];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const float4 pv = *reinterpret_cast<const float4*>(&cs[(lane & 31) * SP + 16 * k2 + 8 * kg + 4 * h]);
          x[4 * h] = pv.x; x[4 * h + 1] = pv.y; x[4 * h + 2] = pv.z; x[4 * h + 3] = pv.w;
        }
        bf16x8 af2[NPL], bf0[NPL], bf1[NPL];
        if constexpr (F16) {
          split_planes8_h(x, s_dz, af2);
          zmh = pk_absmax_f16_(zmh, af2[0]);        // running max |hi plane| (packed fp16): max |dZ| to 2^-11 relative
        } else split_planes8<NPL>(x, af2);
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) {
          bf0[pl] = *reinterpret_cast<const bf16x8*>(&Wb[pl * PB + frag + 16 * ks]);
          bf1[pl] = *reinterpret_cast<const bf16x8*>(&Wb[pl * PB + 32 * SB + frag + 16 * ks]);
        }
#pragma unroll
        for (int ord = NPL - 1; ord >= 0; --ord)
#pragma unroll
          for (int qa = 0; qa <= ord; ++qa) {
            g0 = mfma32_<F16>(af2[qa], bf0[ord - qa], g0);
            g1 = mfma32_<F16>(af2[qa], bf1[ord - qa], g1);
          }
      }
    }
    __syncthreads();
  }
  float4 gv[2][4];                            // dLN of rows rr + 8 i, columns nt * 32 + 4 cq .. + 3
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int rl = (r & 3) + 8 * (r >> 2) + 4 * half;
      cs[rl * SP + col] = (nt ? g1[r] : g0[r]) * u2;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) gv[nt][i] = *reinterpret_cast<const float4*>(&cs[(rr + 8 * i) * SP + cq * 4]);
  }
  if (F16 && a.sc.mid_amax) {
    const f16x2_ zp = __builtin_bit_cast(f16x2_, zmh);
    // (1 + 2^-10): hi is dZ rounded to 11 bits -- the recorded maximum must not be below the true one
    const float zmax = wave_max(fmaxf((float)zp[0], (float)zp[1])) * (1.f + 0x1p-10f) / s_dz;
    if (lane == 0) amax_raise_(a.sc.mid_amax, zmax);
  }
  if (a.X == nullptr) {
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const __amdgpu_buffer_rsrc_t Lrs = make_rsrc_(a.dLN + m0 * 64, (unsigned)(rows_ok * 64 * 4));
        buf_store4_(Lrs, (unsigned)(((wave * 32 + rr + 8 * i) * 64 + nt * 32 + cq * 4) * 4), gv[nt][i]);
      }
    return;
  }
  // LayerNorm backward on the rows in registers: a row's 64 channels sit in the 8 lanes cq = 0..7 of one rr group
  float ag[2][4] = {}, ab[2][4] = {};
  float4 gm[2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) gm[nt] = *reinterpret_cast<const float4*>(a.gamma + nt * 32 + cq * 4);
  // every operand row of the LayerNorm backward is requested here, before any is used (behind `if (ok)` they were a dozen
  // dependent round trips per workgroup)
  float2 mrs[4];
  float4 xvs[4][2], r1s[4][2], r2s[4][2];
  const __amdgpu_buffer_rsrc_t Xrs = make_rsrc_(a.dX + m0 * 64, (unsigned)(rows_ok * 64 * 4));
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const long rg_ = m0 + wave * 32 + rr + 8 * i;
    const long rgc = rg_ < a.M ? rg_ : a.M - 1;
    mrs[i] = *reinterpret_cast<const float2*>(a.stats + 2 * rgc);
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const long off = rgc * 64 + nt * 32 + cq * 4;
      xvs[i][nt] = *reinterpret_cast<const float4*>(a.X + off);
      r1s[i][nt] = *reinterpret_cast<const float4*>(a.dY + off);
      r2s[i][nt] = a.dR2 ? *reinterpret_cast<const float4*>(a.dR2 + off) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const long rg = m0 + wave * 32 + rr + 8 * i;
    const bool ok = rg < a.M;
    const float mean = mrs[i].x, rstd = mrs[i].y;
    float xh[2][4], dxh[2][4], s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const float4 xv = xvs[i][nt];
      const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
      const float dv[4] = {gv[nt][i].x, gv[nt][i].y, gv[nt][i].z, gv[nt][i].w};
      const float gl[4] = {gm[nt].x, gm[nt].y, gm[nt].z, gm[nt].w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        xh[nt][j] = (xs[j] - mean) * rstd;
        dxh[nt][j] = dv[j] * gl[j];
        s1 += dxh[nt][j]; s2 += dxh[nt][j] * xh[nt][j];
        if (ok) { ag[nt][j] += dv[j] * xh[nt][j]; ab[nt][j] += dv[j]; }
      }
    }
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
    s1 *= (1.f / 64.f); s2 *= (1.f / 64.f);
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const float4 r1 = r1s[i][nt], r2 = r2s[i][nt];
      float o4[4] = {r1.x + r2.x, r1.y + r2.y, r1.z + r2.z, r1.w + r2.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) o4[j] += rstd * (dxh[nt][j] - s1 - xh[nt][j] * s2);
      buf_store4_(Xrs, (unsigned)(((wave * 32 + rr + 8 * i) * 64 + nt * 32 + cq * 4) * 4), make_float4(o4[0], o4[1], o4[2], o4[3]));
      if (F16 && ok) xmax = fmaxf(fmaxf(xmax, fmaxf(fabsf(o4[0]), fabsf(o4[1]))), fmaxf(fabsf(o4[2]), fabsf(o4[3])));
    }
  }
  if (F16 && a.sc.out_amax) {
    xmax = wave_max(xmax);
    if (lane == 0) amax_raise_(a.sc.out_amax, xmax);
  }
  // gamma / beta gradients: fold the 8 row groups of the wave (lane bits 3..5), then the 4 waves through LDS
  float* redg = reinterpret_cast<float*>(Wa);         // [4 waves][64 channels][2]; the weight planes are free now
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float sg = ag[nt][j], sb = ab[nt][j];
#pragma unroll
      for (int o = 8; o < 64; o <<= 1) { sg += __shfl_xor(sg, o, 64); sb += __shfl_xor(sb, o, 64); }
      if (rr == 0) { redg[(wave * 64 + nt * 32 + cq * 4 + j) * 2] = sg; redg[(wave * 64 + nt * 32 + cq * 4 + j) * 2 + 1] = sb; }
    }
  __syncthreads();
  if (tid < 64) {
    float sg = 0.f, sb = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) { sg += redg[(w * 64 + tid) * 2]; sb += redg[(w * 64 + tid) * 2 + 1]; }
    atomicAdd(&a.dgamma[tid], sg);
    atomicAdd(&a.dbeta[tid], sb);
  }
}


extern "C" int se_ff_bwd_dgrad_f16(const float* dY, const float* H, const float* W2T, const float* W1T, float* dZ, float* dLN,
                                   long M, int hid, float drop_p, unsigned seed_h, unsigned seed_o, int precision,
                                   const float* X, const float* stats, const float* gamma, const float* dR2, float* dX,
                                   float* dgamma, float* dbeta, const se_f16_scales* sc, void* stream) {
  SE_REQUIRE(dY && H && W2T && W1T && dZ, "ff_bwd_dgrad: null operand");
  SE_REQUIRE(X ? (stats && gamma && dX && dgamma && dbeta) : dLN != nullptr,
             "ff_bwd_dgrad: either dLN, or all of X / stats / gamma / dX / dgamma / dbeta (fused LayerNorm backward)");
  SE_REQUIRE(M > 0 && hid >= 64 && hid % 64 == 0, "ff_bwd_dgrad: M=%ld hid=%d (hid must be a multiple of 64)", M, hid);
  const bool wpl = (precision & 16) != 0;
  precision &= 15;
  SE_REQUIRE(precision == 3 && wpl && sc && sc->wa_amax && sc->wb_amax && hid == 256 && (((size_t)W2T | (size_t)W1T) & 15) == 0,
             "ff_bwd_dgrad: scaled split-fp16 only (precision 3 | 16): pre-split fp16 planes, 16-byte aligned, with their amax scalars (sc), hid == 256");
  SE_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "ff_bwd_dgrad: drop_p=%f out of range", drop_p);
  SE_REQUIRE(M * (long)hid < 4294967296L, "ff_bwd_dgrad: dropout index exceeds 32 bits");
  FfBwdArgs a{dY, H, W2T, W1T, dZ, dLN, M, hid, drop_p, seed_h, seed_o, X, stats, gamma, dR2, dX, dgamma, dbeta, sc ? *sc : se_f16_scales{}};
  dim3 grid((unsigned)((M + 127) / 128)), block(256);
  hipLaunchKernelGGL((ff_bwd_kernel<2, true, 4, true>), grid, block, 0, as_stream(stream), a);
  return se_check_launch("se_ff_bwd_dgrad");
}

extern "C" int se_ff_fwd_f16(const float* X, const float* rowstats, const float* gamma, const float* beta, const float* W1,
                             const float* b1, const float* W2, const float* b2, float* H, float* Y, float* out_stats, long M,
                             int hid, float drop_p, unsigned seed_h, unsigned seed_o, float alpha, int precision,
                             const se_f16_scales* sc, void* stream) {
  SE_REQUIRE(X && rowstats && gamma && beta && W1 && b1 && W2 && b2 && Y, "ff_fwd: null operand");
  SE_REQUIRE(M > 0 && hid == 256, "ff_fwd: M=%ld hid=%d (the Conformer's hid = 256 only)", M, hid);
  SE_REQUIRE(precision == (3 | 16) && sc && sc->wa_amax && sc->wb_amax && (((size_t)W1 | (size_t)W2) & 15) == 0,
             "ff_fwd: scaled split-fp16 only (precision 3 | 16): pre-split fp16 planes, 16-byte aligned, with their amax scalars (sc)");
  SE_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "ff_fwd: drop_p=%f out of range", drop_p);
  SE_REQUIRE(M * (long)hid < 4294963200L, "ff_fwd: dropout index exceeds 32 bits");
  FfArgs a{X, rowstats, gamma, beta, W1, b1, W2, b2, H, Y, M, hid, drop_p, seed_h, seed_o, alpha, out_stats, *sc};
  if (!a.H) {      // the default: W-stationary, H not stored (the fused backward recomputes it)
    const int ncu = se_cu_count();
    const long need = (M + 255) / 256;
    const bool dr = drop_p > 0.f;
    static unsigned raised = 0, raised0 = 0;
    SE_REQUIRE(se_raise_lds(dr ? (const void*)ff_fwd_ws_kernel<true> : (const void*)ff_fwd_ws_kernel<false>, FW_LDS_BYTES, dr ? &raised : &raised0),
               "ff_fwd: cannot raise the dynamic LDS limit");
    if (dr) hipLaunchKernelGGL(ff_fwd_ws_kernel<true>, dim3((unsigned)(need < ncu ? need : ncu)), dim3(512), FW_LDS_BYTES, as_stream(stream), a);
    else hipLaunchKernelGGL(ff_fwd_ws_kernel<false>, dim3((unsigned)(need < ncu ? need : ncu)), dim3(512), FW_LDS_BYTES, as_stream(stream), a);
  } else {         // H stored for se_ff_bwd_dgrad_f16 (the cross-check path: SE_FF_FUSED=0)
    hipLaunchKernelGGL((ff_fwd_kernel<2, true, 4, true, true>), dim3((unsigned)((M + 127) / 128)), dim3(256), 0, as_stream(stream), a);
  }
  return se_check_launch("se_ff_fwd");
}

