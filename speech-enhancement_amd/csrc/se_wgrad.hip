// Weight-gradient kernels of the tap-GEMM family (fp32 MFMA and split-bf16, generic and triple-tap).
#include "se_gemm_dev.h"

// ---------------------------------------------------------------------------------------------
// weight gradient:  dW[n][tap*C + c] += sum_m dY[m][n] * pro(A[src(m,tap)][c])
// grid: (row chunks, ntap * ceil(C/64), ceil(N/64)); 4 waves = 2x2 tiles of 32(n) x 32(c).
struct WgradArgs {
  se_gemm_desc d;
  const float* A; const float* dY; float* dW; float* dbias;
  const float* rowstats; const float* ps; const float* pb;
  long rows_per_chunk;
  int nchunks;
  int buf_ok;            // both operands span < 2^32 bytes: the triple-tap split kernels use range-checked buffer loads
};

template <int PRO>
__global__ __launch_bounds__(256) void wgrad_kernel(WgradArgs g) {
  constexpr int SY = 68;            // 64 + 4 pad (float4-aligned rows)
  constexpr int MR = 64;            // rows staged per step (32 MFMAs per wave between barriers)
  __shared__ __attribute__((aligned(16))) float Ys[MR * SY];
  __shared__ __attribute__((aligned(16))) float Xs[MR * SY];
  const se_gemm_desc& d = g.d;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ncb = (d.C + 63) / 64;
  // siblings = the (tap, channel block, n block) workgroups of one row chunk: same XCD, adjacent in dispatch order,
  // so the dY / A rows they share come out of that XCD's L2
  const int nnb = (d.N + 63) / 64;
  const WorkId wk_ = decode_work(d.ntap * ncb * nnb, g.nchunks, 0);
  if (wk_.outer >= g.nchunks) return;
  const int chunk = wk_.outer, tc = wk_.inner / nnb, nb = wk_.inner - tc * nnb;
  const int tap = tc / ncb, cb = tc - tap * ncb;
  const int Mb = d.To * d.Fo;
  const long Mtot = (long)d.B * Mb;
  const long mbeg = (long)chunk * g.rows_per_chunk;
  long mend = mbeg + g.rows_per_chunk;
  if (mend > Mtot) mend = Mtot;
  const int q = tid & 15, r0 = tid >> 4;    // float4 column / tile row (4 passes of 16 rows)
  const int wn = wave >> 1, wc = wave & 1;
  const bool do_bias = g.dbias != nullptr && tc == 0;
  const bool lin = d.ntap == 1 && !d.up && d.st == 1 && d.sf == 1 && d.dt[0] == 0 && d.df[0] == 0 &&
                   d.Ti == d.To && d.Fi == d.Fo;
  const int TiFi = d.Ti * d.Fi;

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float bsum = 0.f;

  const int n_ld = nb * 64 + q * 4;        // dY column of this thread's float4
  const int c_ld = cb * 64 + q * 4;        // A channel of this thread's float4
  const bool nok = n_ld < d.N;             // N multiple of 4 (checked on host)
  const bool cok = c_ld < d.C;
  const float* __restrict__ Yg = g.dY + d.c_off + n_ld;
  const float* __restrict__ Ag = g.A + d.a_off + c_ld;

  float4 ry[4], rx[4];
  float mean[4] = {}, rstd[4] = {};
  bool xok[4];
  unsigned xpix[4];
  const unsigned thr = drop_thr(d.drop_p);
  const float inv_keep = drop_inv_keep(d.drop_p);
  const bool dy_drop = (d.epilogue & SE_EPI_DROP) != 0;
  // (batch entry, t, f) of this thread's 4 rows, advanced by MR rows per step with adds and compares: the two
  // divisions per row and step they replace (one of them 64-bit) cost more VALU issue slots than the step's MFMAs
  int cb_[4], ct_[4], cf_[4];
  const int adv_t = MR / d.Fo, adv_f = MR - adv_t * d.Fo;
  if (!lin) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      long mg = mbeg + r0 + i * 16;
      cb_[i] = (int)(mg / Mb);
      int m = (int)(mg - (long)cb_[i] * Mb);
      ct_[i] = m / d.Fo;
      cf_[i] = m - ct_[i] * d.Fo;
    }
  }
  auto load_tiles = [&](long mbase) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      long mg = mbase + r0 + i * 16;
      bool ok = mg < mend;
      long p = -1;
      if (lin) { if (ok && cok) p = mg; }
      else {
        if (ok && cok) {
          int pin = src_pixel_in(d, ct_[i], cf_[i], tap);
          p = pin >= 0 ? (long)cb_[i] * TiFi + pin : -1;
        }
        cf_[i] += adv_f; ct_[i] += adv_t;
        if (cf_[i] >= d.Fo) { cf_[i] -= d.Fo; ct_[i] += 1; }
        while (ct_[i] >= d.To) { ct_[i] -= d.To; cb_[i] += 1; }
      }
      ry[i] = (ok && nok) ? *reinterpret_cast<const float4*>(Yg + mg * d.ldc) : make_float4(0.f, 0.f, 0.f, 0.f);
      if (dy_drop && ok && nok) {
        unsigned base = (unsigned)(mg * d.N + n_ld);
        const float4 d4 = drop_scale4(d.epi_seed, base, thr, inv_keep);
        ry[i].x *= d4.x; ry[i].y *= d4.y; ry[i].z *= d4.z; ry[i].w *= d4.w;
      }
      xok[i] = p >= 0;
      xpix[i] = (unsigned)p;
      rx[i] = xok[i] ? *reinterpret_cast<const float4*>(Ag + p * d.lda) : make_float4(0.f, 0.f, 0.f, 0.f);
      if (PRO == SE_PRO_LN) {
        float2 mr = xok[i] ? *reinterpret_cast<const float2*>(g.rowstats + 2 * p) : make_float2(0.f, 0.f);
        mean[i] = mr.x;
        rstd[i] = mr.y;
      }
    }
  };

  float4 ps4 = make_float4(0.f, 0.f, 0.f, 0.f), pb4 = ps4;
  load_pro_vec<PRO>(g.ps, g.pb, c_ld, cok, ps4, pb4);      // this thread's 4 channels never change
  if (mbeg < mend) load_tiles(mbeg);
  for (long mb = mbeg; mb < mend; mb += MR) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float4 v = rx[i];
      if (PRO != SE_PRO_NONE && xok[i])
        v = apply_pro<PRO>(v, c_ld, d.C, mean[i], rstd[i], ps4, pb4, xpix[i], d.pro_seed, thr, inv_keep);
      *reinterpret_cast<float4*>(&Xs[(r0 + i * 16) * SY + q * 4]) = v;
      *reinterpret_cast<float4*>(&Ys[(r0 + i * 16) * SY + q * 4]) = ry[i];
    }
    __syncthreads();
    if (mb + MR < mend) load_tiles(mb + MR);
    // MFMA step s pairs tile rows {s, s + 32}: lane half h supplies row s + 32 h
    const float* yp = &Ys[(lane >> 5) * 32 * SY + wn * 32 + (lane & 31)];
    const float* xp = &Xs[(lane >> 5) * 32 * SY + wc * 32 + (lane & 31)];
#pragma unroll
    for (int s = 0; s < 32; ++s)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(yp[s * SY], xp[s * SY], acc, 0, 0, 0);
    if (do_bias && tid < 64) {
#pragma unroll
      for (int r = 0; r < MR; ++r) bsum += Ys[r * SY + tid];
    }
    __syncthreads();
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0): see wgrad_lin_bf16_kernel (no wait in front of every atomic)
  const int col = lane & 31, half = lane >> 5;
  const int c = cb * 64 + wc * 32 + col;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    int n = nb * 64 + wn * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
    if (n < d.N && c < d.C) atomicAdd(&g.dW[(long)n * d.ldw + (long)tap * d.C + c], d.alpha * acc[r]);
  }
  if (do_bias && tid < 64 && nb * 64 + tid < d.N) atomicAdd(&g.dbias[nb * 64 + tid], d.alpha * bsum);
}

// ---------------------------------------------------------------------------------------------
// Weight gradient of the unit-stride convolutions whose taps come in frequency triples (see conv3_bf16_kernel): one
// workgroup owns (row chunk, dt, channel block, n block) and accumulates the THREE taps df = -1, 0, +1 at once from one
// dY tile and one 66-row halo tile of A (flattened pixels m - 1 .. m + 64 of the dt-shifted rows): a third of the
// loads and LDS stores per MFMA of wgrad_kernel.  Rows whose frequency neighbour is padding are masked through a
// per-row multiplier table (mask[df][row], built at staging time).  fp32 MFMA.
__global__ __launch_bounds__(256) void wgrad3_kernel(WgradArgs g) {
  constexpr int SY = 68, MR = 64, HR = MR + 2;
  __shared__ __attribute__((aligned(16))) float Ys[MR * SY];
  __shared__ __attribute__((aligned(16))) float Xs[HR * SY];
  __shared__ float msk[2][MR];               // [0]: df = -1 allowed, [1]: df = +1 allowed
  const se_gemm_desc& d = g.d;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ncb = (d.C + 63) / 64, nnb = (d.N + 63) / 64, ngrp = d.ntap / 3;
  const WorkId wk_ = decode_work(ngrp * ncb * nnb, g.nchunks, 0);
  if (wk_.outer >= g.nchunks) return;
  const int chunk = wk_.outer, tc = wk_.inner / nnb, nb = wk_.inner - tc * nnb;
  const int gi = tc / ncb, cb = tc - gi * ncb;
  const int Mb = d.To * d.Fo;
  const long Mtot = (long)d.B * Mb;
  const long mbeg = (long)chunk * g.rows_per_chunk;
  long mend = mbeg + g.rows_per_chunk;
  if (mend > Mtot) mend = Mtot;
  const int q = tid & 15, r0 = tid >> 4;
  const int wn = wave >> 1, wc = wave & 1;
  const bool do_bias = g.dbias != nullptr && tc == 0;
  // accumulator slot s <-> tap 3 gi + s; its frequency offset
  int dfs[3];
#pragma unroll
  for (int s3 = 0; s3 < 3; ++s3) dfs[s3] = d.df[3 * gi + s3];
  const long shift = (long)d.dt[3 * gi] * d.Fo;           // flattened pixel shift of this triple's rows

  f32x16 acc[3];
#pragma unroll
  for (int s3 = 0; s3 < 3; ++s3)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[s3][r] = 0.f;
  float bsum = 0.f;
  const int n_ld = nb * 64 + q * 4, c_ld = cb * 64 + q * 4;
  const bool nok = n_ld < d.N, cok = c_ld < d.C;
  const float* __restrict__ Yg = g.dY + d.c_off + n_ld;
  const float* __restrict__ Ag = g.A + d.a_off + c_ld;

  float4 ry[4], rx[4], rh;
  // (entry base row, in-entry pixel) of this thread's 4 tile rows and of its halo row, advanced by MR per step with
  // adds and compares (MR <= Mb is host-checked)
  long eb[5];
  int ip[5];
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const long mg = mbeg + (i < 4 ? r0 + i * 16 : ((tid >> 4) & 1) * (MR - 1));
    const long bq = mg / Mb;
    eb[i] = bq * Mb;
    ip[i] = (int)(mg - eb[i]);
  }
  const int ishift = (int)shift;
  auto load_tiles = [&](long mbase) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const long mg = mbase + r0 + i * 16;
      const bool ok = mg < mend;
      ry[i] = (ok && nok) ? *reinterpret_cast<const float4*>(Yg + mg * d.ldc) : make_float4(0.f, 0.f, 0.f, 0.f);
      const int inb = ip[i] + ishift;
      const bool v = ok && cok && inb >= 0 && inb < Mb;
      rx[i] = v ? *reinterpret_cast<const float4*>(Ag + (eb[i] + inb) * d.lda) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (tid < 32) {                            // halo rows: source pixel of (first row) - 1 and of (last row) + 1
      const int hsel = tid >> 4;
      const long mg = hsel ? mbase + MR - 1 : mbase;
      const int inb = ip[4] + ishift;
      // out-of-entry / out-of-range neighbours are only ever read under a frequency-edge mask or for rows >= mend
      const int nbp = hsel ? inb + 1 : inb - 1;
      const bool v = mg < mend && cok && inb >= 0 && inb < Mb && nbp >= 0 && nbp < Mb;
      rh = v ? *reinterpret_cast<const float4*>(Ag + (eb[4] + nbp) * d.lda) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      ip[i] += MR;
      if (ip[i] >= Mb) { ip[i] -= Mb; eb[i] += Mb; }
    }
  };
  int fm = (tid < MR) ? (int)(((mbeg + tid) % Mb) % d.Fo) : 0;      // frequency index of row mb + tid (mask builder)
  const int fadv = MR % d.Fo;

  if (mbeg < mend) load_tiles(mbeg);
  for (long mb = mbeg; mb < mend; mb += MR) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *reinterpret_cast<float4*>(&Xs[(1 + r0 + i * 16) * SY + q * 4]) = rx[i];
      *reinterpret_cast<float4*>(&Ys[(r0 + i * 16) * SY + q * 4]) = ry[i];
    }
    if (tid < 32) *reinterpret_cast<float4*>(&Xs[((tid >> 4) * (HR - 1)) * SY + q * 4]) = rh;
    if (tid < MR) {                            // frequency-edge masks of the 64 rows of this step
      msk[0][tid] = fm == 0 ? 0.f : 1.f;
      msk[1][tid] = fm == d.Fo - 1 ? 0.f : 1.f;
      fm += fadv;                              // Mb is a multiple of Fo, so entry boundaries do not disturb f
      if (fm >= d.Fo) fm -= d.Fo;
    }
    __syncthreads();
    if (mb + MR < mend) load_tiles(mb + MR);
    const int hrow = (lane >> 5) * 32;
    const float* yp = &Ys[hrow * SY + wn * 32 + (lane & 31)];
    const float* xp = &Xs[(1 + hrow) * SY + wc * 32 + (lane & 31)];
    // tap order inside the loop: the three accumulator chains are independent, so consecutive MFMAs never wait on
    // each other's result; row s of A serves df = -1 at output row s + 1, df = 0 at s, df = +1 at s - 1
#pragma unroll
    for (int s = 0; s < 32; ++s) {
      const float yv = yp[s * SY];
#pragma unroll
      for (int s3 = 0; s3 < 3; ++s3) {
        const int df = dfs[s3];
        float xv = xp[(s + df) * SY];
        if (df != 0) xv *= (df < 0 ? msk[0] : msk[1])[hrow + s];
        acc[s3] = __builtin_amdgcn_mfma_f32_32x32x2f32(yv, xv, acc[s3], 0, 0, 0);
      }
    }
    if (do_bias && tid < 64) {
#pragma unroll
      for (int r = 0; r < MR; ++r) bsum += Ys[r * SY + tid];
    }
    __syncthreads();
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0): see wgrad_lin_bf16_kernel (no wait in front of every atomic)
  const int col = lane & 31, half = lane >> 5;
  const int c = cb * 64 + wc * 32 + col;
#pragma unroll
  for (int s3 = 0; s3 < 3; ++s3) {
    const int tap = 3 * gi + s3;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      int n = nb * 64 + wn * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      if (n < d.N && c < d.C) atomicAdd(&g.dW[(long)n * d.ldw + (long)tap * d.C + c], d.alpha * acc[s3][r]);
    }
  }
  if (do_bias && tid < 64 && nb * 64 + tid < d.N) atomicAdd(&g.dbias[nb * 64 + tid], d.alpha * bsum);
}

// ---------------------------------------------------------------------------------------------
// Split-bf16 weight gradient (precision 1 / 2, same operand splits as gemm_tap_bf16x3_kernel).  The contraction index
// of dW = dY^T X is the ROW index m, and v_mfma_f32_32x32x16_bf16 wants 8 consecutive k per lane: both tiles are
// therefore staged TRANSPOSED in LDS ([column][m], m contiguous).  Every thread owns a 4 (rows) x 4 (columns) block:
// four 16-B global loads (one per row, 16 lanes = one 256-B row segment), a register transpose, and per column one
// 8-B store of 4 consecutive m per plane.  LDS rows are laid out in 16-B cells, cell(r, ch) = 9 r + (r >> 4) + ch (this kernel):
// the 16-lane groups of both the fragment reads (16 consecutive rows, same ch) and the transposed stores (rows 4 l + j)
// then touch 16 distinct bank groups -- conflict-free (searched exhaustively; plain padding gives 4-way write conflicts).
template <int PRO, int NPL>
__global__ __launch_bounds__(256) void wgrad_bf16_kernel(WgradArgs g) {
  constexpr int MR = 64;
  constexpr int PLN = (9 * 64 + 4) * 8;          // bf16 elements of one [64 columns][64 m] plane
  __shared__ __attribute__((aligned(16))) __bf16 Yt[NPL * PLN];
  __shared__ __attribute__((aligned(16))) __bf16 Xt[NPL * PLN];
  const se_gemm_desc& d = g.d;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ncb = (d.C + 63) / 64;
  const int nnb = (d.N + 63) / 64;
  const WorkId wk_ = decode_work(d.ntap * ncb * nnb, g.nchunks, 0);
  if (wk_.outer >= g.nchunks) return;
  const int chunk = wk_.outer, tc = wk_.inner / nnb, nb = wk_.inner - tc * nnb;
  const int tap = tc / ncb, cb = tc - tap * ncb;
  const int Mb = d.To * d.Fo;
  const long Mtot = (long)d.B * Mb;
  const long mbeg = (long)chunk * g.rows_per_chunk;
  long mend = mbeg + g.rows_per_chunk;
  if (mend > Mtot) mend = Mtot;
  const int q = tid & 15, rg = tid >> 4;        // float4 column / group of 4 consecutive tile rows
  const int wn = wave >> 1, wc = wave & 1;
  const bool do_bias = g.dbias != nullptr && tc == 0;
  const bool lin = d.ntap == 1 && !d.up && d.st == 1 && d.sf == 1 && d.dt[0] == 0 && d.df[0] == 0 &&
                   d.Ti == d.To && d.Fi == d.Fo;
  const int TiFi = d.Ti * d.Fi;

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);

  const int n_ld = nb * 64 + q * 4, c_ld = cb * 64 + q * 4;
  const bool nok = n_ld < d.N, cok = c_ld < d.C;
  const float* __restrict__ Yg = g.dY + d.c_off + n_ld;
  const float* __restrict__ Ag = g.A + d.a_off + c_ld;

  // two register sets: the loads of step s + 2 are in flight while steps s and s + 1 run (one step of MFMAs is shorter
  // than a memory round trip and only 2 workgroups fit per CU, so a single-step prefetch left the latency exposed)
  struct Regs { float4 ry[4], rx[4]; float mean[4], rstd[4]; bool xok[4]; unsigned xpix[4]; };
  Regs R0, R1;
  const unsigned thr = drop_thr(d.drop_p);
  const float inv_keep = drop_inv_keep(d.drop_p);
  const bool dy_drop = (d.epilogue & SE_EPI_DROP) != 0;
  int cb_[4], ct_[4], cf_[4];
  const int adv_t = MR / d.Fo, adv_f = MR - adv_t * d.Fo;
  if (!lin) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      long mg = mbeg + rg * 4 + i;
      cb_[i] = (int)(mg / Mb);
      int m = (int)(mg - (long)cb_[i] * Mb);
      ct_[i] = m / d.Fo;
      cf_[i] = m - ct_[i] * d.Fo;
    }
  }
  auto load_tiles = [&](long mbase, Regs& R) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      long mg = mbase + rg * 4 + i;
      bool ok = mg < mend;
      long p = -1;
      if (lin) { if (ok && cok) p = mg; }
      else {
        if (ok && cok) {
          int pin = src_pixel_in(d, ct_[i], cf_[i], tap);
          p = pin >= 0 ? (long)cb_[i] * TiFi + pin : -1;
        }
        cf_[i] += adv_f; ct_[i] += adv_t;
        if (cf_[i] >= d.Fo) { cf_[i] -= d.Fo; ct_[i] += 1; }
        while (ct_[i] >= d.To) { ct_[i] -= d.To; cb_[i] += 1; }
      }
      R.ry[i] = (ok && nok) ? *reinterpret_cast<const float4*>(Yg + mg * d.ldc) : make_float4(0.f, 0.f, 0.f, 0.f);
      if (dy_drop && ok && nok) {
        unsigned base = (unsigned)(mg * d.N + n_ld);
        const float4 d4 = drop_scale4(d.epi_seed, base, thr, inv_keep);
        R.ry[i].x *= d4.x; R.ry[i].y *= d4.y; R.ry[i].z *= d4.z; R.ry[i].w *= d4.w;
      }
      R.xok[i] = p >= 0;
      R.xpix[i] = (unsigned)p;
      R.rx[i] = R.xok[i] ? *reinterpret_cast<const float4*>(Ag + p * d.lda) : make_float4(0.f, 0.f, 0.f, 0.f);
      if (PRO == SE_PRO_LN) {
        float2 mr = R.xok[i] ? *reinterpret_cast<const float2*>(g.rowstats + 2 * p) : make_float2(0.f, 0.f);
        R.mean[i] = mr.x;
        R.rstd[i] = mr.y;
      }
    }
  };
  // register transpose + split + store of one thread block: v[i] = row 4 rg + i, columns 4 q .. 4 q + 3
  auto stage_t = [&](const float4 (&v)[4], __bf16* T) {
    const float x[4][4] = {{v[0].x, v[0].y, v[0].z, v[0].w}, {v[1].x, v[1].y, v[1].z, v[1].w},
                           {v[2].x, v[2].y, v[2].z, v[2].w}, {v[3].x, v[3].y, v[3].z, v[3].w}};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int r = 4 * q + j;
      __bf16* dst = T + (9 * r + (r >> 4)) * 8 + 4 * rg;
      float e[4] = {x[0][j], x[1][j], x[2][j], x[3][j]};
      split_store4<NPL>(e, dst, PLN);
    }
  };

  float4 ps4 = make_float4(0.f, 0.f, 0.f, 0.f), pb4 = ps4;
  load_pro_vec<PRO>(g.ps, g.pb, c_ld, cok, ps4, pb4);
  if (mbeg < mend) load_tiles(mbeg, R0);
  if (mbeg + MR < mend) load_tiles(mbeg + MR, R1); else R1 = R0;
  const int ra_ = wn * 32 + (lane & 31), rb_ = wc * 32 + (lane & 31);
  const __bf16* yfrag = Yt + (9 * ra_ + (ra_ >> 4)) * 8 + 8 * (lane >> 5);
  const __bf16* xfrag = Xt + (9 * rb_ + (rb_ >> 4)) * 8 + 8 * (lane >> 5);
  auto step = [&](long mb, Regs& R) {
    if (PRO != SE_PRO_NONE) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (R.xok[i]) R.rx[i] = apply_pro<PRO>(R.rx[i], c_ld, d.C, R.mean[i], R.rstd[i], ps4, pb4, R.xpix[i], d.pro_seed, thr, inv_keep);
    }
    stage_t(R.rx, Xt);
    stage_t(R.ry, Yt);
    if (do_bias) {
#pragma unroll
      for (int i = 0; i < 4; ++i) { bsum.x += R.ry[i].x; bsum.y += R.ry[i].y; bsum.z += R.ry[i].z; bsum.w += R.ry[i].w; }
    }
    __syncthreads();
    if (mb + 2 * MR < mend) load_tiles(mb + 2 * MR, R);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      bf16x8 af[NPL], bf[NPL];
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl) {
        af[pl] = *reinterpret_cast<const bf16x8*>(yfrag + pl * PLN + 16 * ks);
        bf[pl] = *reinterpret_cast<const bf16x8*>(xfrag + pl * PLN + 16 * ks);
      }
#pragma unroll
      for (int ord = NPL - 1; ord >= 0; --ord)
#pragma unroll
        for (int qa = 0; qa <= ord; ++qa)
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[qa], bf[ord - qa], acc, 0, 0, 0);
    }
    __syncthreads();
  };
  for (long mb = mbeg; mb < mend; mb += 2 * MR) {
    step(mb, R0);
    if (mb + MR < mend) step(mb + MR, R1);
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0): see wgrad_lin_bf16_kernel (no wait in front of every atomic)
  const int col = lane & 31, half = lane >> 5;
  const int c = cb * 64 + wc * 32 + col;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    int n = nb * 64 + wn * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
    if (n < d.N && c < d.C) atomicAdd(&g.dW[(long)n * d.ldw + (long)tap * d.C + c], d.alpha * acc[r]);
  }
  if (do_bias) {       // column sums of dY: fold the 16 row groups through LDS (the tiles are free now)
    float* red = reinterpret_cast<float*>(Yt);
    *reinterpret_cast<float4*>(&red[rg * 64 + 4 * q]) = bsum;
    __syncthreads();
    if (tid < 64) {
      float s_ = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) s_ += red[r * 64 + tid];
      if (nb * 64 + tid < d.N) atomicAdd(&g.dbias[nb * 64 + tid], d.alpha * s_);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Split-bf16 triple-tap weight gradient (precision 1 / 2): wgrad3_kernel's tile sharing (one dY tile + one halo A tile
// per step feed the taps df = -1, 0, +1) with wgrad_bf16_kernel's transposed bf16 staging.  The contraction index (row m)
// is the contiguous LDS axis, so the +-1 row shift of a tap is an unaligned 8-element window: the fragment is assembled
// from the aligned 16-B chunk plus one neighbouring dword with four v_alignbit.  Rows are laid out at position m + 8
// (chunks 1..8; the two halo rows sit at positions 7 and 72).  Frequency-edge rows are cleared in the shifted fragment
// by a dword mask; with Fo > 66 (host-checked) a 66-row tile holds at most one edge row of each kind, whose position
// is a per-step scalar.  One split + one LDS store per element serves 72 MFMAs per wave and step instead of 24.
// F16 (precision 3): two planes of scaled fp16 (se_gemm_dev.h), three fp16 MFMAs per product; the activations are scaled by
// 2^sexp(a_amax | a_sexp), dY by 2^sexp(w_amax | w_sexp) -- the descriptor's second-operand scale -- and the sums are un-scaled
// (exactly) where they are added to dW.
// ORD: every triple lists its taps as df = -1, 0, +1 (host-checked; what gemm.conv_taps produces): the tap's shift is then a
// compile-time constant of the unrolled tap loop -- no run-time selection between the three fragment forms.
template <int NPL, bool F16 = false, bool ORD = false>
__global__ __launch_bounds__(256) void wgrad3_bf16_kernel(WgradArgs g) {
  constexpr int MR = 64;
  // (round 6: the cell maps of wgrad3w_f16_kernel below -- conflict-free ds_read_b128 fragments for the hardware's lane groups)
  constexpr int PLY = 64 * 8 * 8;             // Yt plane: [64 n][8 cells of m], cell(r, ch) = 8 r + (ch ^ gy(r)), gy = (r4, r3, r1 ^ r2)
  constexpr int PLX = (95 * 7 + 12 * 7 + 10) * 8;      // Xt plane: [64 c][10 cells: positions 0 .. 79], cell(r, ch) = 95 (r >> 3) + 12 (r & 7) + ch
  __shared__ __attribute__((aligned(16))) __bf16 Yt[NPL * PLY];
  __shared__ __attribute__((aligned(16))) __bf16 Xt[NPL * PLX];
  const se_gemm_desc& d = g.d;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ncb = (d.C + 63) / 64, nnb = (d.N + 63) / 64, ngrp = d.ntap / 3;
  const WorkId wk_ = decode_work(ngrp * ncb * nnb, g.nchunks, 0);
  if (wk_.outer >= g.nchunks) return;
  const int chunk = wk_.outer, tc = wk_.inner / nnb, nb = wk_.inner - tc * nnb;
  const int gi = tc / ncb, cb = tc - gi * ncb;
  const int Mb = d.To * d.Fo;
  const long Mtot = (long)d.B * Mb;
  const long mbeg = (long)chunk * g.rows_per_chunk;
  long mend = mbeg + g.rows_per_chunk;
  if (mend > Mtot) mend = Mtot;
  const int q = tid & 15, rg = tid >> 4;
  const int wn = wave >> 1, wc = wave & 1;
  const bool do_bias = g.dbias != nullptr && tc == 0;
  int dfs[3];
#pragma unroll
  for (int s3 = 0; s3 < 3; ++s3) dfs[s3] = d.df[3 * gi + s3];
  const int ishift = d.dt[3 * gi] * d.Fo;
  float sx = 1.f, sy = 1.f, unscale = 1.f;
  if (F16) {
    f16_clamp_mode_();
    const int ex = operand_sexp_(d.a_amax, d.a_sexp), ey = operand_sexp_(d.w_amax, d.w_sexp);
    sx = exp2i_(ex); sy = exp2i_(ey); unscale = exp2i_(-ex - ey);
  }

  f32x16 acc[3];
#pragma unroll
  for (int s3 = 0; s3 < 3; ++s3)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[s3][r] = 0.f;
  float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);
  const int n_ld = nb * 64 + q * 4, c_ld = cb * 64 + q * 4;
  const bool nok = n_ld < d.N, cok = c_ld < d.C;
  const float* __restrict__ Yg = g.dY + d.c_off + n_ld;
  const float* __restrict__ Ag = g.A + d.a_off + c_ld;
  // BUF (host-checked: both operands span < 2^32 bytes): range-checked buffer loads with 32-bit offsets instead of predicated
  // 64-bit-address loads -- an invalid row / channel is sent out of range and reads zeros; no branch per load
  const bool BUF = g.buf_ok != 0;
  const __amdgpu_buffer_rsrc_t Yr = make_rsrc_(g.dY + d.c_off, BUF ? (unsigned)((((long)Mtot - 1) * d.ldc + d.N) * 4) : 0u);
  const __amdgpu_buffer_rsrc_t Ar = make_rsrc_(g.A + d.a_off, BUF ? (unsigned)((((long)Mtot - 1) * d.lda + d.C) * 4) : 0u);
  const unsigned yb = nok ? (unsigned)n_ld * 4u : BUF_OOB_, xb = cok ? (unsigned)c_ld * 4u : BUF_OOB_;

  float4 ry[4], rx[4], rh;
  long eb[5];
  int ip[5];
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const long mg = mbeg + (i < 4 ? rg * 4 + i : ((tid >> 4) & 1) * (MR - 1));
    const long bq = mg / Mb;
    eb[i] = bq * Mb;
    ip[i] = (int)(mg - eb[i]);
  }
  auto load_tiles = [&](long mbase) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const long mg = mbase + rg * 4 + i;
      const bool ok = mg < mend;
      const int inb = ip[i] + ishift;
      const bool v = ok && cok && inb >= 0 && inb < Mb;
      if (BUF) {      // (wave-uniform)
        ry[i] = buf_load4_(Yr, ok ? (unsigned)mg * (unsigned)d.ldc * 4u + yb : BUF_OOB_);
        rx[i] = buf_load4_(Ar, v ? (unsigned)(eb[i] + inb) * (unsigned)d.lda * 4u + xb : BUF_OOB_);
      } else {
        ry[i] = (ok && nok) ? *reinterpret_cast<const float4*>(Yg + mg * d.ldc) : make_float4(0.f, 0.f, 0.f, 0.f);
        rx[i] = v ? *reinterpret_cast<const float4*>(Ag + (eb[i] + inb) * d.lda) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    if (tid < 32) {
      const int hsel = tid >> 4;
      const long mg = hsel ? mbase + MR - 1 : mbase;
      const int inb = ip[4] + ishift;
      const int nbp = hsel ? inb + 1 : inb - 1;
      const bool v = mg < mend && cok && inb >= 0 && inb < Mb && nbp >= 0 && nbp < Mb;
      if (BUF) rh = buf_load4_(Ar, v ? (unsigned)(eb[4] + nbp) * (unsigned)d.lda * 4u + xb : BUF_OOB_);
      else rh = v ? *reinterpret_cast<const float4*>(Ag + (eb[4] + nbp) * d.lda) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      ip[i] += MR;
      if (ip[i] >= Mb) { ip[i] -= Mb; eb[i] += Mb; }
    }
  };
  // 4 x 4 register transpose + split + 8-B store per column; poff = position of tile row 0 inside the LDS row
  auto stage_t = [&](const float4 (&v)[4], __bf16* T, int pln, bool halo_layout, float sc) {
    const float x[4][4] = {{v[0].x, v[0].y, v[0].z, v[0].w}, {v[1].x, v[1].y, v[1].z, v[1].w},
                           {v[2].x, v[2].y, v[2].z, v[2].w}, {v[3].x, v[3].y, v[3].z, v[3].w}};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int r = 4 * q + j;
      const int gy = ((r >> 2) & 6) | (((r >> 1) ^ (r >> 2)) & 1);
      __bf16* dst = halo_layout ? T + (95 * (r >> 3) + 12 * (r & 7)) * 8 + 8 + 4 * rg : T + (8 * r + ((rg >> 1) ^ gy)) * 8 + 4 * (rg & 1);
      if constexpr (F16) {
        split_store_h(make_float4(x[0][j], x[1][j], x[2][j], x[3][j]), sc, dst, pln);
      } else {
        float e[4] = {x[0][j], x[1][j], x[2][j], x[3][j]};
        split_store4<NPL>(e, dst, pln);
      }
    }
  };
  int fbase = (int)((mbeg % Mb) % d.Fo);      // frequency index of the step's first row (wave-uniform)
  if (mbeg < mend) load_tiles(mbeg);
  const int ra_ = wn * 32 + (lane & 31), rb_ = wc * 32 + (lane & 31), kg = lane >> 5;
  const __bf16* yfrag = Yt + ra_ * 64;
  int yoff[4];
  {
    const int gy = ((ra_ >> 2) & 6) | (((ra_ >> 1) ^ (ra_ >> 2)) & 1);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) yoff[ks] = ((kg | (2 * ks)) ^ gy) * 8;
  }
  const __bf16* xrow = Xt + (95 * (rb_ >> 3) + 12 * (rb_ & 7)) * 8;
  const int outer_el = kg ? 3 * 8 : 6;     // (+ 16 ks) the neighbour dword a lane cannot take from its partner lane (wgrad3w_f16_kernel)
  for (long mb = mbeg; mb < mend; mb += MR) {
    stage_t(rx, Xt, PLX, true, sx);
    stage_t(ry, Yt, PLY, false, sy);
    if (tid < 32) {                            // halo rows: positions 7 and 72
      const int ph = (tid >> 4) ? 72 : 7;
      const float hv[4] = {rh.x, rh.y, rh.z, rh.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int r = 4 * q + j;
        float e = hv[j];
        if constexpr (F16) {
          e *= sx;
          const _Float16 h = (_Float16)e, l = (_Float16)(e - (float)h);
          unsigned short* xt = reinterpret_cast<unsigned short*>(Xt);
          xt[(95 * (r >> 3) + 12 * (r & 7)) * 8 + ph] = __builtin_bit_cast(unsigned short, h);
          xt[PLX + (95 * (r >> 3) + 12 * (r & 7)) * 8 + ph] = __builtin_bit_cast(unsigned short, l);
        } else {
#pragma unroll
          for (int pl = 0; pl < NPL; ++pl) { __bf16 h = (__bf16)e; e -= (float)h; Xt[pl * PLX + (95 * (r >> 3) + 12 * (r & 7)) * 8 + ph] = h; }
        }
      }
    }
    if (do_bias) {
#pragma unroll
      for (int i = 0; i < 4; ++i) { bsum.x += ry[i].x; bsum.y += ry[i].y; bsum.z += ry[i].z; bsum.w += ry[i].w; }
    }
    // positions (in the 80-slot LDS row) of the frequency-edge rows of this tile, -100 when there is none
    const int r0f = fbase == 0 ? 0 : d.Fo - fbase;                  // tile row with frequency 0
    const int pL = r0f <= 64 ? 8 + r0f : (r0f == d.Fo - 1 ? 7 : -100);
    const int r1f = d.Fo - 1 - fbase;                                // tile row with frequency Fo - 1
    const int pR = r1f <= 64 ? 8 + r1f : (r1f == d.Fo - 1 ? 7 : -100);
    fbase += MR % d.Fo;
    if (fbase >= d.Fo) fbase -= d.Fo;
    __syncthreads();
    if (mb + MR < mend) load_tiles(mb + MR);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      bf16x8 af[NPL];
      unsigned cen[NPL][4], prv[NPL], nxt[NPL];
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl) {
        af[pl] = *reinterpret_cast<const bf16x8*>(yfrag + pl * PLY + yoff[ks]);
        const __bf16* xp = xrow + pl * PLX + 16 * ks;
        const uint4 cv = *reinterpret_cast<const uint4*>(xp + 8 * (kg + 1));
        const unsigned ld = *reinterpret_cast<const unsigned*>(xp + outer_el);      // the outer neighbour dword; the inner one: the partner lane's
        cen[pl][0] = cv.x; cen[pl][1] = cv.y; cen[pl][2] = cv.z; cen[pl][3] = cv.w;
        const u32x2sw_ s1 = __builtin_amdgcn_permlane32_swap(cv.x, cv.w, false, false);
        prv[pl] = kg ? s1[0] : ld;
        nxt[pl] = kg ? ld : s1[1];
      }
#pragma unroll
      for (int s3 = 0; s3 < 3; ++s3) {
        const int df = ORD ? s3 - 1 : dfs[s3];
        bf16x8 bf[NPL];
        if (df == 0) {
#pragma unroll
          for (int pl = 0; pl < NPL; ++pl) { uint4 u = make_uint4(cen[pl][0], cen[pl][1], cen[pl][2], cen[pl][3]); bf[pl] = *reinterpret_cast<bf16x8*>(&u); }
        } else {
          // element e of the shifted window is an edge row -> cleared.  A 64-row tile holds at most one edge row of each kind
          // (Fo > 66), so at most one of the four k-steps sees it: the mask arithmetic (12 VALU per plane pair and tap) sits
          // behind a wave-uniform test of the edge position against this k-step's 16 window elements
          const int e0 = (df > 0 ? pL - 1 : pR + 1) - 8 * (2 * ks + 1);      // e of the lanes kg == 0; kg == 1: e0 - 8
          const bool has_edge = e0 >= 0 && e0 < 16;
#pragma unroll
          for (int pl = 0; pl < NPL; ++pl) {
            unsigned o[4];
            if (df > 0) {
              o[0] = __builtin_amdgcn_alignbit(cen[pl][1], cen[pl][0], 16); o[1] = __builtin_amdgcn_alignbit(cen[pl][2], cen[pl][1], 16);
              o[2] = __builtin_amdgcn_alignbit(cen[pl][3], cen[pl][2], 16); o[3] = __builtin_amdgcn_alignbit(nxt[pl], cen[pl][3], 16);
            } else {
              o[0] = __builtin_amdgcn_alignbit(cen[pl][0], prv[pl], 16); o[1] = __builtin_amdgcn_alignbit(cen[pl][1], cen[pl][0], 16);
              o[2] = __builtin_amdgcn_alignbit(cen[pl][2], cen[pl][1], 16); o[3] = __builtin_amdgcn_alignbit(cen[pl][3], cen[pl][2], 16);
            }
            uint4 u = make_uint4(o[0], o[1], o[2], o[3]);
            bf[pl] = *reinterpret_cast<bf16x8*>(&u);
          }
          if (has_edge) {
            const int e = e0 - 8 * kg;
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) {
              uint4 u = *reinterpret_cast<uint4*>(&bf[pl]);
              unsigned* w4 = reinterpret_cast<unsigned*>(&u);
#pragma unroll
              for (int dd = 0; dd < 4; ++dd) w4[dd] &= e == 2 * dd ? 0xFFFF0000u : (e == 2 * dd + 1 ? 0x0000FFFFu : 0xFFFFFFFFu);
              bf[pl] = *reinterpret_cast<bf16x8*>(&u);
            }
          }
        }
#pragma unroll
        for (int ord = NPL - 1; ord >= 0; --ord)
#pragma unroll
          for (int qa = 0; qa <= ord; ++qa)
            acc[s3] = mfma32_<F16>(af[qa], bf[ord - qa], acc[s3]);
      }
    }
    __syncthreads();
  }
  const int col = lane & 31, half = lane >> 5;
  const int c = cb * 64 + wc * 32 + col;
  const float oscale = d.alpha * unscale;
  __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0): see wgrad_lin_bf16_kernel
#pragma unroll
  for (int s3 = 0; s3 < 3; ++s3) {
    const int tap = 3 * gi + s3;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      int n = nb * 64 + wn * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      if (n < d.N && c < d.C) atomicAdd(&g.dW[(long)n * d.ldw + (long)tap * d.C + c], oscale * acc[s3][r]);
    }
  }
  if (do_bias) {
    float* red = reinterpret_cast<float*>(Yt);
    *reinterpret_cast<float4*>(&red[rg * 64 + 4 * q]) = bsum;
    __syncthreads();
    if (tid < 64) {
      float s_ = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) s_ += red[r * 64 + tid];
      if (nb * 64 + tid < d.N) atomicAdd(&g.dbias[nb * 64 + tid], d.alpha * s_);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// WIDE form of wgrad3_bf16_kernel<2, F16, ORD> for C >= 128 (round 3): a workgroup owns 64 n x 128 c x 3 taps, a wave ALL 64 n (two
// A fragments) x one 32-channel block.  The expensive part of a k-step is the B side -- the unaligned +-1 window (four v_alignbit
// per plane), its edge test, three LDS reads per plane -- and it now feeds six MFMAs per tap instead of three; the dY tile is
// split and staged once per 128 channels instead of once per 64.  VALU per MFMA 12 -> 7.  Same cell layouts (rows 64..127 of the
// activation image continue the 10 r + (r >> 3) pattern), same scaled split-fp16 arithmetic, taps in the order df = -1, 0, +1.
__global__ __launch_bounds__(256, 2) void wgrad3w_f16_kernel(WgradArgs g) {
  constexpr int MR = 64, NPL = 2;
  // LDS images in 16-byte cells (8 values), laid out for the lane groups the hardware actually services (MI355X_MICROARCH.md, LDS:
  // ds_read_b128 = four groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, + 32; bank = dword mod 64 -- the round-3 maps `9 r + (r >> 4)` /
  // `10 r + (r >> 3)` had been searched for contiguous 16-lane groups and ran 2 - 2.5x their conflict-free cycles: SQ_LDS_BANK_CONFLICT /
  // SQ_LDS_ACTIVE 0.63).  tools/micro/lds_wgrad3w_model.py restates the bank model and both maps:
  //   Yt plane: [64 n][8 cells of m]: cell(r, ch) = 8 r + (ch ^ gy(r)), gy = (r4, r3, r1 ^ r2): b128 fragment reads conflict-free, the
  //             transposing 8-byte stores 2-way (16 lanes share 8 cells: their floor with whole-cell swizzles)
  //   Xt plane: [128 c][10 cells: positions 0 .. 79, data at 8 .. 71, halo elements 7 and 72]: cell(r, ch) = 95 (r >> 3) + 12 (r & 7) + ch:
  //             b128 reads conflict-free, stores 2-way
  constexpr int PLY = 64 * 8 * 8;              // elements per Yt plane
  constexpr int PLX = (95 * 15 + 12 * 7 + 10) * 8;      // elements per Xt plane (1519 cells)
  __shared__ __attribute__((aligned(16))) __bf16 Yt[NPL * PLY];
  __shared__ __attribute__((aligned(16))) __bf16 Xt[NPL * PLX];
  const se_gemm_desc& d = g.d;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ncb = (d.C + 127) / 128, nnb = (d.N + 63) / 64, ngrp = d.ntap / 3;
  const WorkId wk_ = decode_work(ngrp * ncb * nnb, g.nchunks, 0);
  if (wk_.outer >= g.nchunks) return;
  const int chunk = wk_.outer, tc = wk_.inner / nnb, nb = wk_.inner - tc * nnb;
  const int gi = tc / ncb, cb = tc - gi * ncb;
  const int Mb = d.To * d.Fo;
  const long Mtot = (long)d.B * Mb;
  const long mbeg = (long)chunk * g.rows_per_chunk;
  long mend = mbeg + g.rows_per_chunk;
  if (mend > Mtot) mend = Mtot;
  const int q = tid & 15, rg = tid >> 4;
  const bool do_bias = g.dbias != nullptr && tc == 0;
  const int ishift = d.dt[3 * gi] * d.Fo;
  f16_clamp_mode_();
  const int ex = operand_sexp_(d.a_amax, d.a_sexp), ey = operand_sexp_(d.w_amax, d.w_sexp);
  const float sx = exp2i_(ex), sy = exp2i_(ey), unscale = exp2i_(-ex - ey);

  f32x16 acc[2][3];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int s3 = 0; s3 < 3; ++s3)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][s3][r] = 0.f;
  float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);
  const int n_ld = nb * 64 + q * 4, c_ld = cb * 128 + q * 4;
  const bool nok = n_ld < d.N;
  const bool cok[2] = {c_ld < d.C, c_ld + 64 < d.C};
  // range-checked buffer descriptors, 32-bit byte offsets (host-checked < 2^32): an invalid row / channel is sent out of range
  // and reads zeros -- no branch, no 64-bit address per load (the predicated form spilled a pointer and reloaded it, with a full
  // vmcnt(0), in front of every load: the step's twelve loads ran one after the other)
  const __amdgpu_buffer_rsrc_t Yr = make_rsrc_(g.dY + d.c_off, (unsigned)((((long)Mtot - 1) * d.ldc + d.N) * 4));
  const __amdgpu_buffer_rsrc_t Ar = make_rsrc_(g.A + d.a_off, (unsigned)((((long)Mtot - 1) * d.lda + d.C) * 4));
  const unsigned yb = nok ? (unsigned)n_ld * 4u : BUF_OOB_;
  const unsigned xb[2] = {cok[0] ? (unsigned)c_ld * 4u : BUF_OOB_, cok[1] ? (unsigned)(c_ld + 64) * 4u : BUF_OOB_};

  float4 ry[4], rx[2][4], rh[2];
  long eb[5];
  int ip[5];
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const long mg = mbeg + (i < 4 ? rg * 4 + i : ((tid >> 4) & 1) * (MR - 1));
    const long bq = mg / Mb;
    eb[i] = bq * Mb;
    ip[i] = (int)(mg - eb[i]);
  }
  auto load_tiles = [&](long mbase) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const long mg = mbase + rg * 4 + i;
      const bool ok = mg < mend;
      ry[i] = buf_load4_(Yr, ok ? (unsigned)mg * (unsigned)d.ldc * 4u + yb : BUF_OOB_);
      const int inb = ip[i] + ishift;
      const bool v = ok && inb >= 0 && inb < Mb;
      const unsigned ro = (unsigned)(eb[i] + inb) * (unsigned)d.lda * 4u;
#pragma unroll
      for (int h = 0; h < 2; ++h) rx[h][i] = buf_load4_(Ar, v ? ro + xb[h] : BUF_OOB_);
    }
    if (tid < 32) {
      const int hsel = tid >> 4;
      const long mg = hsel ? mbase + MR - 1 : mbase;
      const int inb = ip[4] + ishift;
      const int nbp = hsel ? inb + 1 : inb - 1;
      const bool v = mg < mend && inb >= 0 && inb < Mb && nbp >= 0 && nbp < Mb;
      const unsigned ro = (unsigned)(eb[4] + nbp) * (unsigned)d.lda * 4u;
#pragma unroll
      for (int h = 0; h < 2; ++h) rh[h] = buf_load4_(Ar, v ? ro + xb[h] : BUF_OOB_);
    }
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      ip[i] += MR;
      if (ip[i] >= Mb) { ip[i] -= Mb; eb[i] += Mb; }
    }
  };
  // 4 x 4 register transpose + split + 8-B store per column (row r0 + 4 q + j of the image)
  auto stage_t = [&](const float4 (&v)[4], __bf16* T, int pln, bool halo_layout, float sc, int r0) {
    const float x[4][4] = {{v[0].x, v[0].y, v[0].z, v[0].w}, {v[1].x, v[1].y, v[1].z, v[1].w},
                           {v[2].x, v[2].y, v[2].z, v[2].w}, {v[3].x, v[3].y, v[3].z, v[3].w}};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int r = r0 + 4 * q + j;
      const int gy = ((r >> 2) & 6) | (((r >> 1) ^ (r >> 2)) & 1);
      __bf16* dst = halo_layout ? T + (95 * (r >> 3) + 12 * (r & 7)) * 8 + 8 + 4 * rg : T + (8 * r + ((rg >> 1) ^ gy)) * 8 + 4 * (rg & 1);
      split_store_h(make_float4(x[0][j], x[1][j], x[2][j], x[3][j]), sc, dst, pln);
    }
  };
  int fbase = (int)((mbeg % Mb) % d.Fo);      // frequency index of the step's first row (wave-uniform)
  if (mbeg < mend) load_tiles(mbeg);
  const int l31 = lane & 31, kg = lane >> 5;
  const int rb_ = wave * 32 + l31;
  const __bf16* yfrag[2];      // row of the lane's n (a = 0: l31, a = 1: 32 + l31 -- gy does not depend on r5); yoff[ks]: its swizzled cell of k-step ks
#pragma unroll
  for (int a = 0; a < 2; ++a) yfrag[a] = Yt + (a * 32 + l31) * 64;
  int yoff[4];
  {
    const int gy = ((l31 >> 2) & 6) | (((l31 >> 1) ^ (l31 >> 2)) & 1);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) yoff[ks] = ((kg | (2 * ks)) ^ gy) * 8;
  }
  const __bf16* xrow = Xt + (95 * (rb_ >> 3) + 12 * (rb_ & 7)) * 8;
  const int outer_el = kg ? 3 * 8 : 6;     // (+ 16 ks) the neighbour dword a lane cannot take from its partner: kg = 0: the last dword of cell 2 ks,
                                           // kg = 1: the first dword of cell 2 ks + 3
  const bool active = cb * 128 + wave * 32 < d.C;            // (wave-uniform) a wave whose channel block is padding skips the products
  for (long mb = mbeg; mb < mend; mb += MR) {
    stage_t(rx[0], Xt, PLX, true, sx, 0);
    stage_t(rx[1], Xt, PLX, true, sx, 64);
    stage_t(ry, Yt, PLY, false, sy, 0);
    if (tid < 32) {                            // halo rows: positions 7 and 72
      const int ph = (tid >> 4) ? 72 : 7;
      unsigned short* xt = reinterpret_cast<unsigned short*>(Xt);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const float hv[4] = {rh[h].x, rh[h].y, rh[h].z, rh[h].w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int r = 64 * h + 4 * q + j;
          const float e = hv[j] * sx;
          const _Float16 hh = (_Float16)e, ll = (_Float16)(e - (float)hh);
          xt[(95 * (r >> 3) + 12 * (r & 7)) * 8 + ph] = __builtin_bit_cast(unsigned short, hh);
          xt[PLX + (95 * (r >> 3) + 12 * (r & 7)) * 8 + ph] = __builtin_bit_cast(unsigned short, ll);
        }
      }
    }
    if (do_bias) {
#pragma unroll
      for (int i = 0; i < 4; ++i) { bsum.x += ry[i].x; bsum.y += ry[i].y; bsum.z += ry[i].z; bsum.w += ry[i].w; }
    }
    // positions (in the 80-slot LDS row) of the frequency-edge rows of this tile, -100 when there is none
    const int r0f = fbase == 0 ? 0 : d.Fo - fbase;                  // tile row with frequency 0
    const int pL = r0f <= 64 ? 8 + r0f : (r0f == d.Fo - 1 ? 7 : -100);
    const int r1f = d.Fo - 1 - fbase;                                // tile row with frequency Fo - 1
    const int pR = r1f <= 64 ? 8 + r1f : (r1f == d.Fo - 1 ? 7 : -100);
    fbase += MR % d.Fo;
    if (fbase >= d.Fo) fbase -= d.Fo;
    __syncthreads();
    if (mb + MR < mend) load_tiles(mb + MR);
    if (active) {
      // The +-1 position windows need one dword of each neighbouring cell.  Cells 2 ks + 1 (lanes kg = 0) and 2 ks + 2 (kg = 1) are read
      // as 16-byte fragments anyway and lanes l, l + 32 hold neighbouring cells of the same channel: the INNER neighbour dword comes
      // from the partner lane with one v_permlane32_swap, only the outer one (kg = 0: the last dword of cell 2 ks, kg = 1: the first
      // dword of cell 2 ks + 3) from LDS -- ONE ds_read_b32 per k-step and plane instead of two (a 32-lane dword read of a cell image is
      // 4-way bank-conflicted by construction: 32 lanes, 8 dword positions mod 32).  (The form that takes the outer dwords from the
      // partner's fragments of the neighbouring k-steps too -- no dword read at all -- needs 16 more live registers and spilled.)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        bf16x8 af[2][NPL];
        unsigned cen[NPL][4], prv[NPL], nxt[NPL];
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) {
#pragma unroll
          for (int a = 0; a < 2; ++a) af[a][pl] = *reinterpret_cast<const bf16x8*>(yfrag[a] + pl * PLY + yoff[ks]);
          const __bf16* xp = xrow + pl * PLX + 16 * ks;
          const uint4 cv = *reinterpret_cast<const uint4*>(xp + 8 * (kg + 1));
          const unsigned ld = *reinterpret_cast<const unsigned*>(xp + outer_el);
          cen[pl][0] = cv.x; cen[pl][1] = cv.y; cen[pl][2] = cv.z; cen[pl][3] = cv.w;
          // swap(x, y): r[0] = (x of lanes 0-31 | y of lanes 0-31), r[1] = (x of lanes 32-63 | y of lanes 32-63)
          const u32x2sw_ s1 = __builtin_amdgcn_permlane32_swap(cv.x, cv.w, false, false);     // r[0] high lanes: partner's last dword; r[1] low lanes: partner's first dword
          prv[pl] = kg ? s1[0] : ld;
          nxt[pl] = kg ? ld : s1[1];
        }
#pragma unroll
        for (int s3 = 0; s3 < 3; ++s3) {
          const int df = s3 - 1;
          bf16x8 bf[NPL];
          if (df == 0) {
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) { uint4 u = make_uint4(cen[pl][0], cen[pl][1], cen[pl][2], cen[pl][3]); bf[pl] = *reinterpret_cast<bf16x8*>(&u); }
          } else {
            const int e0 = (df > 0 ? pL - 1 : pR + 1) - 8 * (2 * ks + 1);      // e of the lanes kg == 0; kg == 1: e0 - 8
            const bool has_edge = e0 >= 0 && e0 < 16;
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) {
              unsigned o[4];
              if (df > 0) {
                o[0] = __builtin_amdgcn_alignbit(cen[pl][1], cen[pl][0], 16); o[1] = __builtin_amdgcn_alignbit(cen[pl][2], cen[pl][1], 16);
                o[2] = __builtin_amdgcn_alignbit(cen[pl][3], cen[pl][2], 16); o[3] = __builtin_amdgcn_alignbit(nxt[pl], cen[pl][3], 16);
              } else {
                o[0] = __builtin_amdgcn_alignbit(cen[pl][0], prv[pl], 16); o[1] = __builtin_amdgcn_alignbit(cen[pl][1], cen[pl][0], 16);
                o[2] = __builtin_amdgcn_alignbit(cen[pl][2], cen[pl][1], 16); o[3] = __builtin_amdgcn_alignbit(cen[pl][3], cen[pl][2], 16);
              }
              uint4 u = make_uint4(o[0], o[1], o[2], o[3]);
              bf[pl] = *reinterpret_cast<bf16x8*>(&u);
            }
            if (has_edge) {
              const int e = e0 - 8 * kg;
#pragma unroll
              for (int pl = 0; pl < NPL; ++pl) {
                uint4 u = *reinterpret_cast<uint4*>(&bf[pl]);
                unsigned* w4 = reinterpret_cast<unsigned*>(&u);
#pragma unroll
                for (int dd = 0; dd < 4; ++dd) w4[dd] &= e == 2 * dd ? 0xFFFF0000u : (e == 2 * dd + 1 ? 0x0000FFFFu : 0xFFFFFFFFu);
                bf[pl] = *reinterpret_cast<bf16x8*>(&u);
              }
            }
          }
#pragma unroll
          for (int ord = NPL - 1; ord >= 0; --ord)
#pragma unroll
            for (int qa = 0; qa <= ord; ++qa)
#pragma unroll
              for (int a = 0; a < 2; ++a)
                acc[a][s3] = mfma32_<true>(af[a][qa], bf[ord - qa], acc[a][s3]);
        }
      }
    }
    __syncthreads();
  }
  const int col = lane & 31, half = lane >> 5;
  const int c = cb * 128 + wave * 32 + col;
  const float oscale = d.alpha * unscale;
  __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0): see wgrad_lin_bf16_kernel
  if (active) {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int s3 = 0; s3 < 3; ++s3) {
        const int tap = 3 * gi + s3;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int n = nb * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
          if (n < d.N && c < d.C) atomicAdd(&g.dW[(long)n * d.ldw + (long)tap * d.C + c], oscale * acc[a][s3][r]);
        }
      }
  }
  if (do_bias) {
    float* red = reinterpret_cast<float*>(Yt);
    *reinterpret_cast<float4*>(&red[rg * 64 + 4 * q]) = bsum;
    __syncthreads();
    if (tid < 64) {
      float s_ = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) s_ += red[r * 64 + tid];
      if (nb * 64 + tid < d.N) atomicAdd(&g.dbias[nb * 64 + tid], d.alpha * s_);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Row-GEMM (nn.Linear / 1x1 conv) weight gradient with the WHOLE [N x C] gradient in one workgroup (N, C <= 256, N * C <= 16 K:
// every token-wise layer of the Conformer).  wgrad_kernel gives each 64 x 64 block of dW its own workgroup, so the 4 sibling
// workgroups of a 256 x 64 gradient each re-load the shared operand's rows and re-apply its prologue (LayerNorm, BatchNorm-affine
// + Swish, Swish + dropout hash), and a wave issues two LDS reads per MFMA for its 32 x 32 tile.  Here a wave owns
// (TN * 32) x (TC * 32) outputs (four 32 x 32 accumulators for the 256 x 64 / 64 x 256 shapes: one LDS read per MFMA), the 4
// waves tile [N x C] as WN x (4 / WN), every row of both operands is loaded, prologued and staged exactly once per launch, and a
// step is 32 rows = 64 MFMAs per wave between barriers.  fp32 MFMA (32x32x2), fp32 atomics across row chunks.
template <int PRO, int WN, int TN, int TC>
__global__ __launch_bounds__(256, 3) void wgrad_lin_kernel(WgradArgs g) {
  constexpr int WC = 4 / WN, NT = WN * TN * 32, CT = WC * TC * 32, MR = 32;
  constexpr int SYL = NT + 4, SXL = CT + 4;                    // LDS row strides (floats): 16-B aligned rows
  constexpr int QY = NT / 4, RY = 256 / QY, NY = MR / RY;       // dY tile: float4 columns, rows per pass, passes
  constexpr int QX = CT / 4, RX = 256 / QX, NX = MR / RX;
  __shared__ __attribute__((aligned(16))) float Ys[MR * SYL];
  __shared__ __attribute__((aligned(16))) float Xs[MR * SXL];
  const se_gemm_desc& d = g.d;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave / WC, wc = wave - wn * WC;
  const long Mtot = (long)d.To * d.Fo;                           // row GEMM: B == 1 (host-checked)
  const long mbeg = (long)blockIdx.x * g.rows_per_chunk;
  long mend = mbeg + g.rows_per_chunk;
  if (mend > Mtot) mend = Mtot;
  const int qy = tid % QY, ry0 = tid / QY, qx = tid % QX, rx0 = tid / QX;
  const int n_ld = qy * 4, c_ld = qx * 4;
  const bool nok = n_ld < d.N, cok = c_ld < d.C;
  const float* __restrict__ Yg = g.dY + d.c_off + n_ld;
  const float* __restrict__ Ag = g.A + d.a_off + c_ld;
  const unsigned thr = drop_thr(d.drop_p);
  const float inv_keep = drop_inv_keep(d.drop_p);
  const bool dy_drop = (d.epilogue & SE_EPI_DROP) != 0;

  f32x16 acc[TN][TC];
#pragma unroll
  for (int a = 0; a < TN; ++a)
#pragma unroll
    for (int b = 0; b < TC; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  float bsum = 0.f;
  float4 ry[NY], rx[NX];
  float mean[NX] = {}, rstd[NX] = {};
  bool xok[NX];
  auto load_tiles = [&](long mbase) {
#pragma unroll
    for (int i = 0; i < NY; ++i) {
      const long mg = mbase + ry0 + i * RY;
      const bool ok = mg < mend && nok;
      ry[i] = ok ? *reinterpret_cast<const float4*>(Yg + mg * d.ldc) : make_float4(0.f, 0.f, 0.f, 0.f);
      if (dy_drop && ok) {
        const float4 d4 = drop_scale4(d.epi_seed, (unsigned)(mg * d.N + n_ld), thr, inv_keep);
        ry[i].x *= d4.x; ry[i].y *= d4.y; ry[i].z *= d4.z; ry[i].w *= d4.w;
      }
    }
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const long mg = mbase + rx0 + i * RX;
      xok[i] = mg < mend && cok;
      rx[i] = xok[i] ? *reinterpret_cast<const float4*>(Ag + mg * d.lda) : make_float4(0.f, 0.f, 0.f, 0.f);
      if (PRO == SE_PRO_LN) {
        const float2 mr = xok[i] ? *reinterpret_cast<const float2*>(g.rowstats + 2 * mg) : make_float2(0.f, 0.f);
        mean[i] = mr.x;
        rstd[i] = mr.y;
      }
    }
  };
  float4 ps4 = make_float4(0.f, 0.f, 0.f, 0.f), pb4 = ps4;
  load_pro_vec<PRO>(g.ps, g.pb, c_ld, cok, ps4, pb4);           // this thread's 4 channels never change
  if (mbeg < mend) load_tiles(mbeg);
  const int half = lane >> 5, l31 = lane & 31;
  const float* yp = &Ys[half * 16 * SYL + wn * TN * 32 + l31];   // MFMA step s pairs tile rows {s, s + 16}
  const float* xp = &Xs[half * 16 * SXL + wc * TC * 32 + l31];
  const bool do_bias = g.dbias != nullptr;
  for (long mb = mbeg; mb < mend; mb += MR) {
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      float4 v = rx[i];
      if (PRO != SE_PRO_NONE && xok[i])
        v = apply_pro<PRO>(v, c_ld, d.C, mean[i], rstd[i], ps4, pb4, (unsigned)(mb + rx0 + i * RX), d.pro_seed, thr, inv_keep);
      *reinterpret_cast<float4*>(&Xs[(rx0 + i * RX) * SXL + qx * 4]) = v;
    }
#pragma unroll
    for (int i = 0; i < NY; ++i) *reinterpret_cast<float4*>(&Ys[(ry0 + i * RY) * SYL + qy * 4]) = ry[i];
    __syncthreads();
    if (mb + MR < mend) load_tiles(mb + MR);
    if (wn * TN * 32 < d.N && wc * TC * 32 < d.C) {               // wave-uniform: a wave whose whole tile is padding idles
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        float av[TN], bv[TC];
#pragma unroll
        for (int a = 0; a < TN; ++a) av[a] = yp[s * SYL + a * 32];
#pragma unroll
        for (int b = 0; b < TC; ++b) bv[b] = xp[s * SXL + b * 32];
#pragma unroll
        for (int a = 0; a < TN; ++a)
#pragma unroll
          for (int b = 0; b < TC; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[a], bv[b], acc[a][b], 0, 0, 0);
      }
    }
    if (do_bias && tid < NT) {
#pragma unroll
      for (int r = 0; r < MR; ++r) bsum += Ys[r * SYL + tid];
    }
    __syncthreads();
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0): see wgrad_lin_bf16_kernel (no wait in front of every atomic)
#pragma unroll
  for (int a = 0; a < TN; ++a)
#pragma unroll
    for (int b = 0; b < TC; ++b) {
      const int c = (wc * TC + b) * 32 + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = (wn * TN + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (n < d.N && c < d.C) atomicAdd(&g.dW[(long)n * d.ldw + c], d.alpha * acc[a][b][r]);
      }
    }
  if (do_bias && tid < NT && tid < d.N) atomicAdd(&g.dbias[tid], d.alpha * bsum);
}

template <int WN, int TN, int TC>
static int launch_wgrad_lin(const se_gemm_desc* d, const WgradArgs& g, dim3 grid, hipStream_t s) {
  const dim3 block(256);
  switch (d->prologue) {
    case SE_PRO_NONE: hipLaunchKernelGGL((wgrad_lin_kernel<SE_PRO_NONE, WN, TN, TC>), grid, block, 0, s, g); break;
    case SE_PRO_LN:      // (LayerNorm(64): C == 64, i.e. only the shape whose C fits one 64-column block -- no instantiation for the others)
      if constexpr (WN == 4) hipLaunchKernelGGL((wgrad_lin_kernel<SE_PRO_LN, WN, TN, TC>), grid, block, 0, s, g);
      else return se_fail("wgrad: the LayerNorm prologue needs C == 64");
      break;
    case SE_PRO_SWISH: hipLaunchKernelGGL((wgrad_lin_kernel<SE_PRO_SWISH, WN, TN, TC>), grid, block, 0, s, g); break;
    case SE_PRO_AFFINE_SWISH: hipLaunchKernelGGL((wgrad_lin_kernel<SE_PRO_AFFINE_SWISH, WN, TN, TC>), grid, block, 0, s, g); break;
    case SE_PRO_SWISH_DROP: hipLaunchKernelGGL((wgrad_lin_kernel<SE_PRO_SWISH_DROP, WN, TN, TC>), grid, block, 0, s, g); break;
    case SE_PRO_DROP: hipLaunchKernelGGL((wgrad_lin_kernel<SE_PRO_DROP, WN, TN, TC>), grid, block, 0, s, g); break;
    default: return se_fail("wgrad: unknown prologue %d", d->prologue);
  }
  return se_check_launch("se_gemm_tap_wgrad(lin)");
}

// ---------------------------------------------------------------------------------------------
// The same whole-gradient row GEMM on the bf16 matrix cores (exact hi / mid / lo split, six products: fp32-equivalent).
// wgrad_lin_kernel's fp32 MFMA does 64 FLOP / clk / SIMD -- at [256 x 64] x 518 736 rows that alone is 125 us, as long as the
// HBM time of the operands; six bf16 products cost 2.7x fewer matrix-pipe cycles.  The split-bf16 block kernel lost that
// advantage to its own VALU work (every 64 x 64 block re-splits both operand tiles); here each operand element is split ONCE
// per launch.  The contraction index is the row m, so the MFMA fragments need 8 consecutive rows of one column per lane:
//   * wide operand (<= 256 columns): a lane loads the float4 of its 4 columns for the 8 rows of its wave's row octet (eight
//     1-KB wave loads), transposes in registers, splits 8 values at a time (split_planes8) and writes one 16-B fragment cell
//     per (plane, column);
//   * narrow operand (<= 64 columns): a lane loads 2 consecutive rows x float4 and writes (row pair) dwords into the cells.
// LDS image: [plane][row octet][column cell][8 rows] bf16 (wide operand: cells swizzled, see swz below -- fragment reads and
// cell writes both conflict free), 60 KB per workgroup, two workgroups per CU; global loads of the next 32 rows are in flight during the 48 MFMAs.
// SH = 1: dY wide (N <= 256), X narrow (C <= 64): FF W1, pointwise-GLU conv.  SH = 2: X wide (C <= 256), dY narrow: FF W2.
// F16 (precision 3): two planes of scaled fp16 instead of three of bf16 (se_gemm_dev.h), three MFMAs per product: X scaled by
// 2^sexp(a_amax | a_sexp) after its prologue, dY by 2^sexp(w_amax | w_sexp); the sums are un-scaled where they are added to dW.
template <int PRO, int SH, bool F16 = false>
__global__ __launch_bounds__(256, 2) void wgrad_lin_bf16_kernel(WgradArgs g) {
  constexpr int MR = 32, WIDE = 256, NARROW = 64, NP = F16 ? 2 : 3;
  constexpr int WPLN = 4 * WIDE * 8, NPLN = 4 * NARROW * 8;         // 16-bit elements per plane
  __shared__ __attribute__((aligned(16))) __bf16 Ws[NP * WPLN];
  __shared__ __attribute__((aligned(16))) __bf16 Ns[NP * NPLN];
  const se_gemm_desc& d = g.d;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const long Mtot = (long)d.To * d.Fo;                               // row GEMM: B == 1 (host-checked)
  const long mbeg = (long)blockIdx.x * g.rows_per_chunk;
  long mend = mbeg + g.rows_per_chunk;
  if (mend > Mtot) mend = Mtot;
  constexpr bool XW = SH == 2;                                       // X is the wide operand
  const int wcols = XW ? d.C : d.N, ncols = XW ? d.N : d.C;
  const long ldw_ = XW ? d.lda : d.ldc, ldn_ = XW ? d.ldc : d.lda;
  const int wq = lane * 4;                                           // wide: this lane's 4 columns, rows 8 * wave + i
  const int nq = (tid & 15) * 4, nrp = tid >> 4;                     // narrow: 4 columns, rows 2 * nrp, 2 * nrp + 1
  const bool wok = wq < wcols, nok = nq < ncols;
  // (lanes of padding columns read column 0 and rows past the chunk read its last row: every load is unconditional -- a
  // predicated load is its own exec-masked branch region with its own wait -- and `fix` zeroes what must not count)
  const float* __restrict__ Wg = (XW ? g.A + d.a_off : g.dY + d.c_off) + (wok ? wq : 0);
  const float* __restrict__ Ng = (XW ? g.dY + d.c_off : g.A + d.a_off) + (nok ? nq : 0);
  const unsigned thr = drop_thr(d.drop_p);
  const float inv_keep = drop_inv_keep(d.drop_p);
  const bool dy_drop = (d.epilogue & SE_EPI_DROP) != 0;
  const bool do_bias = g.dbias != nullptr;
  float s_wide = 1.f, s_narrow = 1.f, unscale = 1.f;
  if (F16) {
    f16_clamp_mode_();
    const int ex = operand_sexp_(d.a_amax, d.a_sexp), ey = operand_sexp_(d.w_amax, d.w_sexp);
    s_wide = exp2i_(SH == 2 ? ex : ey); s_narrow = exp2i_(SH == 2 ? ey : ex); unscale = exp2i_(-ex - ey);
  }

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 rw[8], rn[2];
  float2 sw[8], sn[2];                                               // LayerNorm row statistics (PRO == SE_PRO_LN only)
  float4 psw = make_float4(0.f, 0.f, 0.f, 0.f), pbw = psw, psn = psw, pbn = psw;
  if (XW) load_pro_vec<PRO>(g.ps, g.pb, wq, wok, psw, pbw); else load_pro_vec<PRO>(g.ps, g.pb, nq, nok, psn, pbn);
  auto load_tiles = [&](long mbase) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      long mg = mbase + 8 * wave + i;
      mg = mg < mend ? mg : mend - 1;
      rw[i] = *reinterpret_cast<const float4*>(Wg + mg * ldw_);
      if (XW && PRO == SE_PRO_LN) sw[i] = *reinterpret_cast<const float2*>(g.rowstats + 2 * mg);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      long mg = mbase + 2 * nrp + i;
      mg = mg < mend ? mg : mend - 1;
      rn[i] = *reinterpret_cast<const float4*>(Ng + mg * ldn_);
      if (!XW && PRO == SE_PRO_LN) sn[i] = *reinterpret_cast<const float2*>(g.rowstats + 2 * mg);
    }
  };
  // operand transforms at staging time: the X prologue (the forward's LayerNorm / Swish / dropout, recomputed) and the dY mask
  auto fix = [&](float4 v, bool is_x, int c0, bool ok, float2 st, float4 ps4, float4 pb4, long mg) -> float4 {
    const bool live = ok && mg < mend;            // (selected at the end: an early return is an exec-masked region per call)
    if (is_x) {
      if (PRO != SE_PRO_NONE) v = apply_pro<PRO>(v, c0, d.C, st.x, st.y, ps4, pb4, mg, d.pro_seed, thr, inv_keep);
    } else if (dy_drop) {
      const float4 d4 = drop_scale4(d.epi_seed, (unsigned)(mg * d.N + c0), thr, inv_keep);
      v.x *= d4.x; v.y *= d4.y; v.z *= d4.z; v.w *= d4.w;
    }
    return make_float4(live ? v.x : 0.f, live ? v.y : 0.f, live ? v.z : 0.f, live ? v.w : 0.f);
  };
  if (mbeg < mend) load_tiles(mbeg);
  // fragment bases: A = dY^T (rows n), B = X (columns c); this wave's 64 x 64 block of dW
  // wide image: column c sits in cell swz(c) = 64 (c & 3) + ((c >> 2) + 4 (c & 3)) mod 64 of its row octet -- the 64 lanes of a
  // staging write (columns 4 lane + j) then cover 64 consecutive cells, and the 16 lanes of a fragment-read pass 16 distinct
  // 16-B bank groups (column-major cells made every staging write a 4-way bank conflict: 0.68 conflict cycles per active one)
  auto swz = [](int c) { return 64 * (c & 3) + (((c >> 2) + 4 * (c & 3)) & 63); };
  const int wf0 = swz(wave * 64 + l31), wf1 = swz(wave * 64 + 32 + l31);
  // narrow image (round 4): the same swizzle on 64 columns -- column c in cell 16 (c & 3) + ((c >> 2) + 4 (c & 3)) mod 16.  Column-
  // major cells put the 16 lanes of a staging write (columns 4 t + j: 16 dwords apart) on 2 banks x the row-pair dword: an 8-way
  // conflict on each of the 8 ds_write_b32 per lane and step (SQ_LDS_BANK_CONFLICT / IDX_ACTIVE 0.44 - 0.54 in profiles/r04a)
  auto swn = [](int c) { return 16 * (c & 3) + (((c >> 2) + 4 * (c & 3)) & 15); };
  const int nf0 = swn(l31), nf1 = swn(32 + l31);
  const __bf16* abase = XW ? Ns + (half * NARROW) * 8 : Ws + (half * WIDE) * 8;
  const __bf16* bbase = XW ? Ws + (half * WIDE) * 8 : Ns + (half * NARROW) * 8;
  const int aoff[2] = {XW ? nf0 * 8 : wf0 * 8, XW ? nf1 * 8 : wf1 * 8}, boff[2] = {XW ? wf0 * 8 : nf0 * 8, XW ? wf1 * 8 : nf1 * 8};
  constexpr int APL = XW ? NPLN : WPLN, BPL = XW ? WPLN : NPLN, AOC = (XW ? NARROW : WIDE) * 8, BOC = (XW ? WIDE : NARROW) * 8;
  const bool active = wave * 64 < wcols;                              // a wave whose whole block is padding idles
  for (long mb = mbeg; mb < mend; mb += MR) {
    {   // wide operand: 8 rows x 4 columns per lane -> 4 columns x 3 planes of 8-row cells
      float4 v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        v[i] = fix(rw[i], XW, wq, wok, (XW && PRO == SE_PRO_LN) ? sw[i] : make_float2(0.f, 0.f), psw, pbw, mb + 8 * wave + i);
        if (!XW && do_bias) { bsum.x += v[i].x; bsum.y += v[i].y; bsum.z += v[i].z; bsum.w += v[i].w; }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float x[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = j == 0 ? v[i].x : (j == 1 ? v[i].y : (j == 2 ? v[i].z : v[i].w));
        bf16x8 pl[NP];
        if constexpr (F16) split_planes8_h(x, s_wide, pl); else split_planes8<3, bf16x8>(x, pl);
#pragma unroll
        for (int q = 0; q < NP; ++q) *reinterpret_cast<bf16x8*>(&Ws[q * WPLN + (wave * WIDE + 64 * j + ((lane + 4 * j) & 63)) * 8]) = pl[q];
      }
    }
    {   // narrow operand: 2 rows x 4 columns per lane -> (row pair) dwords
      float4 v0 = fix(rn[0], !XW, nq, nok, (!XW && PRO == SE_PRO_LN) ? sn[0] : make_float2(0.f, 0.f), psn, pbn, mb + 2 * nrp);
      float4 v1 = fix(rn[1], !XW, nq, nok, (!XW && PRO == SE_PRO_LN) ? sn[1] : make_float2(0.f, 0.f), psn, pbn, mb + 2 * nrp + 1);
      if (XW && do_bias) { bsum.x += v0.x + v1.x; bsum.y += v0.y + v1.y; bsum.z += v0.z + v1.z; bsum.w += v0.w + v1.w; }
      float a4[4] = {v0.x, v0.y, v0.z, v0.w}, b4[4] = {v1.x, v1.y, v1.z, v1.w};
      // (swizzled cells: column nq + j = 4 t + j -> cell 16 j + (t + 4 j) mod 16, t = tid & 15)
      unsigned* cell = reinterpret_cast<unsigned*>(&Ns[((nrp >> 2) * NARROW) * 8 + 2 * (nrp & 3)]);
      const int tq = tid & 15;
      if constexpr (F16) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float ya = a4[j] * s_narrow, yb = b4[j] * s_narrow;
          const unsigned w = pk_f16_(ya, yb);
          const f16x2_ hh = __builtin_bit_cast(f16x2_, w);
          const int pc = (16 * j + ((tq + 4 * j) & 15)) * 8;
          cell[pc / 2] = w;
          cell[(NPLN + pc) / 2] = pk_f16_(ya - (float)hh[0], yb - (float)hh[1]);
        }
      } else {
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const unsigned w = pk_bf16_(a4[j], b4[j]);
            cell[(q * NPLN + (16 * j + ((tq + 4 * j) & 15)) * 8) / 2] = w;
            if (q < 2) { a4[j] -= __builtin_bit_cast(float, w << 16); b4[j] -= __builtin_bit_cast(float, w & 0xffff0000u); }
          }
      }
    }
    __syncthreads();
    if (mb + MR < mend) load_tiles(mb + MR);
    if (active) {
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        bf16x8 af[2][NP], bf[2][NP];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int q = 0; q < NP; ++q) {
            af[t][q] = *reinterpret_cast<const bf16x8*>(abase + q * APL + 2 * s * AOC + aoff[t]);
            bf[t][q] = *reinterpret_cast<const bf16x8*>(bbase + q * BPL + 2 * s * BOC + boff[t]);
          }
#pragma unroll
        for (int ord = NP - 1; ord >= 0; --ord)
#pragma unroll
          for (int qa = 0; qa <= ord; ++qa)
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
              for (int b = 0; b < 2; ++b)
                acc[a][b] = mfma32_<F16>(af[a][qa], bf[b][ord - qa], acc[a][b]);
      }
    }
    __syncthreads();
  }
  // one full wait here: the wait-count pass still sees the (guarded, on the last step never issued) prefetch of the next tile as
  // pending and put an `s_waitcnt vmcnt(0)` in front of EVERY atomic below that reuses one of its destination registers -- which
  // also waits for the previous atomic: the 48 - 96 atomics of a wave left one L2 round trip apart
  __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0)
  const int wn = XW ? 0 : wave, wc = XW ? wave : 0;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int c = (wc * 2 + b) * 32 + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = (wn * 2 + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (n < d.N && c < d.C) atomicAdd(&g.dW[(long)n * d.ldw + c], d.alpha * unscale * acc[a][b][r]);
      }
    }
  if (do_bias) {          // column sums of dY, folded over the lanes that shared a column group
    float* red = reinterpret_cast<float*>(Ws);
    constexpr int NG = XW ? 16 : 4, NC = XW ? NARROW : WIDE;
    const int grp = XW ? nrp : wave, c0 = XW ? nq : wq;
    *reinterpret_cast<float4*>(&red[grp * NC + c0]) = bsum;
    __syncthreads();
    if (tid < NC && tid < d.N) {
      float s_ = 0.f;
#pragma unroll
      for (int r = 0; r < NG; ++r) s_ += red[r * NC + tid];
      atomicAdd(&g.dbias[tid], d.alpha * s_);
    }
  }
}

template <int SH>
static int launch_wgrad_lin_f16(const se_gemm_desc* d, const WgradArgs& g, dim3 grid, hipStream_t s) {
  const dim3 block(256);
  switch (d->prologue) {
    case SE_PRO_NONE: hipLaunchKernelGGL((wgrad_lin_bf16_kernel<SE_PRO_NONE, SH, true>), grid, block, 0, s, g); break;
    case SE_PRO_LN: hipLaunchKernelGGL((wgrad_lin_bf16_kernel<SE_PRO_LN, SH, true>), grid, block, 0, s, g); break;
    case SE_PRO_SWISH: hipLaunchKernelGGL((wgrad_lin_bf16_kernel<SE_PRO_SWISH, SH, true>), grid, block, 0, s, g); break;
    case SE_PRO_AFFINE_SWISH: hipLaunchKernelGGL((wgrad_lin_bf16_kernel<SE_PRO_AFFINE_SWISH, SH, true>), grid, block, 0, s, g); break;
    case SE_PRO_SWISH_DROP: hipLaunchKernelGGL((wgrad_lin_bf16_kernel<SE_PRO_SWISH_DROP, SH, true>), grid, block, 0, s, g); break;
    default: return se_fail("wgrad: no scaled split-fp16 kernel for prologue %d", d->prologue);
  }
  return se_check_launch("se_gemm_tap_wgrad(lin f16x3)");
}

template <int SH>
static int launch_wgrad_lin_bf16(const se_gemm_desc* d, const WgradArgs& g, dim3 grid, hipStream_t s) {
  const dim3 block(256);
  switch (d->prologue) {
    case SE_PRO_NONE: hipLaunchKernelGGL((wgrad_lin_bf16_kernel<SE_PRO_NONE, SH>), grid, block, 0, s, g); break;
    case SE_PRO_LN: hipLaunchKernelGGL((wgrad_lin_bf16_kernel<SE_PRO_LN, SH>), grid, block, 0, s, g); break;
    case SE_PRO_SWISH: hipLaunchKernelGGL((wgrad_lin_bf16_kernel<SE_PRO_SWISH, SH>), grid, block, 0, s, g); break;
    case SE_PRO_AFFINE_SWISH: hipLaunchKernelGGL((wgrad_lin_bf16_kernel<SE_PRO_AFFINE_SWISH, SH>), grid, block, 0, s, g); break;
    case SE_PRO_SWISH_DROP: hipLaunchKernelGGL((wgrad_lin_bf16_kernel<SE_PRO_SWISH_DROP, SH>), grid, block, 0, s, g); break;
    case SE_PRO_DROP: hipLaunchKernelGGL((wgrad_lin_bf16_kernel<SE_PRO_DROP, SH>), grid, block, 0, s, g); break;
    default: return se_fail("wgrad: unknown prologue %d", d->prologue);
  }
  return se_check_launch("se_gemm_tap_wgrad(lin bf16x6)");
}

// which arithmetic / kernel class the last se_gemm_tap_wgrad of this thread ran (the dispatch below depends on shape, precision, scales and
// environment switches; bench.py keys its weight-gradient families by THIS, so that no family is priced against the wrong peak):
// bits 0..3: 0 fp32 MFMA, 1 bf16x3, 2 bf16x6, 3 scaled f16x3; bits 4..7: 0 generic tile kernel, 1 triple-tap (conv3) kernel, 2 whole-gradient
// (token-wise) kernel
static thread_local int g_wgrad_kind = 0;
extern "C" int se_gemm_tap_wgrad_last_kind(void) { return g_wgrad_kind; }

extern "C" int se_gemm_tap_wgrad(const se_gemm_desc* d, const float* A, const float* dY, float* dW,
                                 float* dbias, const float* rowstats, const float* pro_scale,
                                 const float* pro_shift, int chunks, void* stream) {
  if (int e = check_desc(d)) return e;
  SE_REQUIRE(A && dY && dW, "wgrad: null operand");
  SE_REQUIRE((d->N % 4) == 0 && (d->ldc % 4) == 0 && (d->c_off % 4) == 0, "wgrad: N/ldc/c_off must be multiples of 4");
  if (d->prologue == SE_PRO_LN) SE_REQUIRE(rowstats && pro_scale && pro_shift, "wgrad: LN prologue operands");
  if (d->prologue == SE_PRO_AFFINE_SWISH) SE_REQUIRE(pro_scale && pro_shift, "wgrad: affine prologue operands");
  const long Mtot = (long)d->B * d->To * d->Fo;
  if (chunks < 1) chunks = 1;
  long rpc = (Mtot + chunks - 1) / chunks;
  rpc = ((rpc + 63) / 64) * 64;
  chunks = (int)((Mtot + rpc - 1) / rpc);
  WgradArgs g{*d, A, dY, dW, dbias, rowstats, pro_scale, pro_shift, rpc, chunks, 0};
  g.buf_ok = ((Mtot - 1) * d->lda + d->C) * 4 < 4294967280L && ((Mtot - 1) * d->ldc + d->N) * 4 < 4294967280L;
  dim3 grid((unsigned)((long)d->ntap * cdiv(d->C, 64) * cdiv(d->N, 64) * ((chunks + 7) / 8 * 8))), block(256);
  hipStream_t s = as_stream(stream);
  if (d->prologue == SE_PRO_NONE && !(d->epilogue & SE_EPI_DROP) && !d->up && d->st == 1 && d->sf == 1 &&
      d->Ti == d->To && d->Fi == d->Fo && d->ntap >= 3 && d->ntap % 3 == 0 && d->Fo >= 2 && d->To * d->Fo >= 64 &&
      getenv("SE_GEMM_NO_CONV3") == nullptr) {
    bool triples = true, ordered = true;
    for (int t3 = 0; t3 < d->ntap && triples; t3 += 3) {
      int seen = 0;
      for (int j = 0; j < 3; ++j) {
        if (d->dt[t3 + j] != d->dt[t3] || d->df[t3 + j] < -1 || d->df[t3 + j] > 1) triples = false;
        else seen |= 1 << (d->df[t3 + j] + 1);
        if (d->df[t3 + j] != j - 1) ordered = false;
      }
      if (seen != 7) triples = false;
    }
    if (triples && (d->precision == 0 || d->Fo > 66)) {
      dim3 g3((unsigned)((long)(d->ntap / 3) * cdiv(d->C, 64) * cdiv(d->N, 64) * ((chunks + 7) / 8 * 8)));
      if (d->precision == 3 && ordered && d->C >= 128 && g.buf_ok) {
        dim3 g3w((unsigned)((long)(d->ntap / 3) * cdiv(d->C, 128) * cdiv(d->N, 64) * ((chunks + 7) / 8 * 8)));
        hipLaunchKernelGGL(wgrad3w_f16_kernel, g3w, block, 0, s, g);
      }
      else if (d->precision == 3 && ordered) hipLaunchKernelGGL((wgrad3_bf16_kernel<2, true, true>), g3, block, 0, s, g);
      else if (d->precision == 3) hipLaunchKernelGGL((wgrad3_bf16_kernel<2, true>), g3, block, 0, s, g);
      else if (d->precision == 1) hipLaunchKernelGGL(wgrad3_bf16_kernel<2>, g3, block, 0, s, g);
      else if (d->precision == 2) hipLaunchKernelGGL(wgrad3_bf16_kernel<3>, g3, block, 0, s, g);
      else hipLaunchKernelGGL(wgrad3_kernel, g3, block, 0, s, g);
      g_wgrad_kind = 16 | (d->precision & 3);
      return se_check_launch("se_gemm_tap_wgrad(conv3)");
    }
  }
  const bool want_f16 = d->precision == 3;
  se_gemm_desc dfb;
  if (d->precision == 3) { dfb = *d; dfb.precision = 2; d = &dfb; }      // shapes without a scaled split-fp16 kernel: six-product / fp32 ones
  // token-wise layers (row GEMM, fp32 MFMA, the whole gradient fits one workgroup): every operand row staged once per launch
  const bool lin = d->ntap == 1 && d->B == 1 && !d->up && d->st == 1 && d->sf == 1 && d->dt[0] == 0 && d->df[0] == 0 &&
                   d->Ti == d->To && d->Fi == d->Fo;
  if (lin && (d->precision == 0 || d->precision == 2) && (d->C % 4) == 0 && (d->lda % 4) == 0 && (d->a_off % 4) == 0 &&
      getenv("SE_WGRAD_NO_LIN") == nullptr) {
    // measured at M = 518 736 (tools/tools_lin_wgrad.py): [64 x 256] 254 vs 293 us, [256 x 64] 250 vs 257 us; [192 x 64] is slower here
    // (a quarter of the workgroup idles: 234 vs 189 us) and [64 x 128] equal (118 vs 120 us) -> those stay on wgrad_kernel
    int shape = 0;                                               // 1: N <= 256, C <= 64; 2: N <= 64, C <= 256
    if (d->C <= 64 && d->N > 192 && d->N <= 256) shape = 1;
    else if (d->N <= 64 && d->C > 128 && d->C <= 256) shape = 2;
    const bool f16_pro = d->prologue == SE_PRO_NONE || d->prologue == SE_PRO_LN || d->prologue == SE_PRO_SWISH ||
                         d->prologue == SE_PRO_AFFINE_SWISH || d->prologue == SE_PRO_SWISH_DROP;
    if (want_f16 && f16_pro && !shape) {
      // the two-plane kernel is bound by the HBM stream of its operands, not by its MFMAs: the narrower gradients ([64 x 128] of the
      // second pointwise conv, [64 x 64] of to_out, [192 x 64] of qkv) run on it too, with the waves of the padding columns idle
      if (d->C <= 64 && d->N <= 256) shape = 1;
      else if (d->N <= 64 && d->C <= 256) shape = 2;
    }
    if (shape && d->precision == 2) {
      // split-bf16 form: one resident round of 2 workgroups per CU (60 KB of LDS each)
      long rl = (Mtot + 511) / 512;
      if (rl < 256) rl = 256;
      rl = ((rl + 31) / 32) * 32;
      const int nch = (int)((Mtot + rl - 1) / rl);
      WgradArgs gl{*d, A, dY, dW, dbias, rowstats, pro_scale, pro_shift, rl, nch};
      g_wgrad_kind = 32 | ((want_f16 && f16_pro) ? 3 : 2);
      if (want_f16 && f16_pro)
        return shape == 1 ? launch_wgrad_lin_f16<1>(d, gl, dim3((unsigned)nch), s) : launch_wgrad_lin_f16<2>(d, gl, dim3((unsigned)nch), s);
      return shape == 1 ? launch_wgrad_lin_bf16<1>(d, gl, dim3((unsigned)nch), s) : launch_wgrad_lin_bf16<2>(d, gl, dim3((unsigned)nch), s);
    }
    if (shape) {
      // one resident round of 3 workgroups per CU (42 KB of LDS each), steps of 32 rows
      // (at least 8 steps per workgroup: every workgroup ends with N * C atomics, which dominated at small M -- 81 us of
      // atomics for 3 steps of work at batch 2)
      long rl = (Mtot + 767) / 768;
      if (rl < 256) rl = 256;
      rl = ((rl + 31) / 32) * 32;
      const int nch = (int)((Mtot + rl - 1) / rl);
      WgradArgs gl{*d, A, dY, dW, dbias, rowstats, pro_scale, pro_shift, rl, nch};
      const dim3 gg((unsigned)nch);
      g_wgrad_kind = 32;
      if (shape == 1) return launch_wgrad_lin<4, 2, 2>(d, gl, gg, s);
      return launch_wgrad_lin<1, 2, 2>(d, gl, gg, s);
    }
  }
  // generic (non-triple) shapes: the six-product split kernel was VALU-bound by its own splits and measured slower than the fp32-MFMA
  // kernel it is numerically equivalent to (77 vs 83 TFLOP/s; removed in round 6) -> precision 2 runs the fp32 kernel there
  if (d->precision == 1) {
    switch (d->prologue) {
      case SE_PRO_NONE: hipLaunchKernelGGL((wgrad_bf16_kernel<SE_PRO_NONE, 2>), grid, block, 0, s, g); break;
      case SE_PRO_LN: hipLaunchKernelGGL((wgrad_bf16_kernel<SE_PRO_LN, 2>), grid, block, 0, s, g); break;
      case SE_PRO_SWISH: hipLaunchKernelGGL((wgrad_bf16_kernel<SE_PRO_SWISH, 2>), grid, block, 0, s, g); break;
      case SE_PRO_AFFINE_SWISH: hipLaunchKernelGGL((wgrad_bf16_kernel<SE_PRO_AFFINE_SWISH, 2>), grid, block, 0, s, g); break;
      case SE_PRO_SWISH_DROP: hipLaunchKernelGGL((wgrad_bf16_kernel<SE_PRO_SWISH_DROP, 2>), grid, block, 0, s, g); break;
      case SE_PRO_DROP: hipLaunchKernelGGL((wgrad_bf16_kernel<SE_PRO_DROP, 2>), grid, block, 0, s, g); break;
      default: return se_fail("wgrad: unknown prologue %d", d->prologue);
    }
    g_wgrad_kind = d->precision;
    return se_check_launch("se_gemm_tap_wgrad(bf16)");
  }
  switch (d->prologue) {
    case SE_PRO_NONE: hipLaunchKernelGGL((wgrad_kernel<SE_PRO_NONE>), grid, block, 0, s, g); break;
    case SE_PRO_LN: hipLaunchKernelGGL((wgrad_kernel<SE_PRO_LN>), grid, block, 0, s, g); break;
    case SE_PRO_SWISH: hipLaunchKernelGGL((wgrad_kernel<SE_PRO_SWISH>), grid, block, 0, s, g); break;
    case SE_PRO_AFFINE_SWISH: hipLaunchKernelGGL((wgrad_kernel<SE_PRO_AFFINE_SWISH>), grid, block, 0, s, g); break;
    case SE_PRO_SWISH_DROP: hipLaunchKernelGGL((wgrad_kernel<SE_PRO_SWISH_DROP>), grid, block, 0, s, g); break;
    case SE_PRO_DROP: hipLaunchKernelGGL((wgrad_kernel<SE_PRO_DROP>), grid, block, 0, s, g); break;
    default: return se_fail("wgrad: unknown prologue %d", d->prologue);
  }
  g_wgrad_kind = 0;
  return se_check_launch("se_gemm_tap_wgrad");
}

