// Fused Shaw-relative-position multi-head attention for the Conformer blocks (models/conformer.py:74-125),
// fp32-exact on v_mfma_f32_16x16x4_f32.  heads = 4, dim_head = 16: one MFMA K-chain (4 steps) covers a head.
//
//   logits[i][j] = scale * ( q_i . k_j  +  q_i . E[clamp(i-j, -P, P) + P] ),  softmax over j,  out = P v
//
// Nothing n x n is ever written: per (sequence, head, 32-query block) one wave sweeps the keys 16 at a
// time with an online softmax.  All three contractions are MFMAs:
//   S^T  = K  . Q^T      (16 keys x 16 queries; query on the lane, 4 keys in the registers)
//   U    = Ew . Q^T      (16 relative offsets x 16 queries) -> skewed through 2 KB of wave-private LDS
//                         so that lane (query c, key j) picks U[c_abs - j]; the offset window slides by 16
//                         per key step, so only ONE new U tile is computed per step (no 2x rel-pos work)
//   O^T += V^T . P^T     (P^T's accumulator registers ARE the B operand: k-index g <-> key 4g+r)
// K / V / E fragments are read straight from global memory (L2-resident, 64-B head rows); the fp32 MFMA
// (32 cycles per 16x16x4) leaves the vector-memory pipe idle enough that LDS staging buys nothing here.
//
// Token geometry: token(s, p) = (s / inner) * outer_stride + (s % inner) * inner_stride + p * pos_stride, so
// the time Conformer ([B*F', T, C] in the reference, generator.py:69) and the frequency Conformer
// ([B*T, F', C], generator.py:71) both run on the one channels-last [B, T, F', C] buffer with no transposes.
#include "se_common.h"
#include <stdlib.h>
#ifndef SE_ATTN_NO_XCD_MAP
#define SE_ATTN_NO_XCD_MAP 0
#endif

struct AttnGeom {
  int nseq, n;             // sequences, positions per sequence
  int inner;               // sequences per outer group
  long outer_stride, inner_stride, pos_stride;   // in tokens
};

struct AttnArgs {
  AttnGeom g;
  const float* QKV;  // [tokens][192]: q | k | v, head h at columns h*16 .. h*16+15 of each third
  const float* E;    // [2*maxpos+1][16]
  float* O;          // [tokens][64]
  float* LSE;        // [tokens][4]   log-sum-exp of the scaled logits (natural log)
  int maxpos;
  float scale;
  const void* Es;    // optional: E pre-split into three bf16 planes [3][2*maxpos+1][16] (se_weight_prep), es_plane elements apart
  long es_plane;
  // scaled split-fp16 form of the v3 kernel (se_attn_fwd_f16): Es = TWO fp16 planes of E * 2^sexp(*e_amax) (se_weight_prep fmt 1),
  // qkv_amax = device scalar >= max |QKV| (raised by the epilogue of the qkv GEMM: se_gemm_desc.y_amax)
  const float* qkv_amax; const float* e_amax;
  int eager;         // (A/B switch SE_ATTN_FWD_LAZY=0: the softmax reference follows the running maximum at every key tile)
};

static __device__ __forceinline__ long tok_of(const AttnGeom& g, int s, int p) {
  return (long)(s / g.inner) * g.outer_stride + (long)(s % g.inner) * g.inner_stride + (long)p * g.pos_stride;
}

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

// one wave = one (sequence, head, 32-query block)
__global__ __launch_bounds__(256) void attn_fwd_kernel(AttnArgs a) {
  __shared__ float Ul[4][2][3][16 * 16];     // [wave][query tile][ring slot][rel row][query]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = lane & 15, g = lane >> 4;
  const int n = a.g.n;
  const int qblocks = (n + 31) / 32;
  long item = (long)blockIdx.x * 4 + wave;
  const long nitems = (long)a.g.nseq * 4 * qblocks;
  if (item >= nitems) return;
  const int qb = (int)(item % qblocks);
  const int head = (int)((item / qblocks) % 4);
  const int seq = (int)(item / ((long)qblocks * 4));
  const int i0 = qb * 32;
  const float* qkv = a.QKV + head * 16;
  const float l2e = 1.4426950408889634f * a.scale;

  // Q^T fragments (B operand): lane (query c, kgrp g) supplies Q[i][4s+g]
  float qf[2][4];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    int qi = i0 + 16 * t + c;
    if (qi > n - 1) qi = n - 1;
    const float* qp = qkv + tok_of(a.g, seq, qi) * 192;
#pragma unroll
    for (int s = 0; s < 4; ++s) qf[t][s] = qp[4 * s + g];
  }
  f32x4 o[2][2];
  float m[2], l[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    m[t] = -1e30f; l[t] = 0.f;
#pragma unroll
    for (int e = 0; e < 2; ++e) o[t][e] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }

  // relative-offset tiles: U_t(D) holds offsets delta in [D, D+15] for query tile t.
  // For key tile j0 and query tile t: D0 = i0 + 16t - j0; needed tiles U_t(D0) (delta_local >= 0) and
  // U_t(D0-16).  U_t(D0) of this step == U_t(D0'-16) of the previous step, so one new tile per step.
  auto e_frag = [&](int D, float (&ef)[4]) {   // A operand: lane (row i=c, kgrp g) supplies E[clamp(D+c)][4s+g]
    int d = D + c;
    d = d < -a.maxpos ? -a.maxpos : (d > a.maxpos ? a.maxpos : d);
    const float* ep = a.E + (long)(d + a.maxpos) * 16;
#pragma unroll
    for (int s = 0; s < 4; ++s) ef[s] = ep[4 * s + g];
  };
  auto u_tile = [&](const float (&ef)[4], int t, int slot) {
    f32x4 u = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 4; ++s) u = MFMA16(ef[s], qf[t][s], u);
#pragma unroll
    for (int r = 0; r < 4; ++r) Ul[wave][t][slot][(4 * g + r) * 16 + c] = u[r];
  };
  // prime: U_t(D0) for the first key tile (j0 = 0): D0 = i0 + 16t
  {
    float ef[4];
    e_frag(i0, ef);        u_tile(ef, 0, 0);
    e_frag(i0 + 16, ef);   u_tile(ef, 1, 0);
  }
  int slot_hi = 0;   // ring slot holding U_t(D0)
  const int nkt = (n + 15) / 16;
  for (int kt = 0; kt < nkt; ++kt) {
    const int j0 = kt * 16;
    const int slot_lo = slot_hi == 2 ? 0 : slot_hi + 1;
    // new low tiles U_t(D0 - 16); the two query tiles need E rows 16 apart
    float ef0[4], ef1[4];
    e_frag(i0 - j0 - 16, ef0);
    e_frag(i0 - j0, ef1);
    u_tile(ef0, 0, slot_lo);
    u_tile(ef1, 1, slot_lo);
    // K fragment (A operand): lane (key c, kgrp g) supplies K[j0+c][4s+g]; V^T fragment: lane (d=c, g) supplies
    // V[j0 + 4g + r][d] for PV step r
    int kj = j0 + c;
    if (kj > n - 1) kj = n - 1;
    const float* kp = qkv + tok_of(a.g, seq, kj) * 192 + 64;
    float kf[4], vf[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) kf[s] = kp[4 * s + g];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      int vj = j0 + 4 * g + r;
      if (vj > n - 1) vj = n - 1;
      vf[r] = qkv[tok_of(a.g, seq, vj) * 192 + 128 + c];
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      f32x4 s4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 4; ++s) s4 = MFMA16(kf[s], qf[t][s], s4);
      // add the skewed relative term, mask keys beyond n, online softmax
      float sc[4], tmax = -1e30f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int dl = c - (4 * g + r);                 // delta - D0, in [-15, 15]
        float u = dl >= 0 ? Ul[wave][t][slot_hi][dl * 16 + c] : Ul[wave][t][slot_lo][(16 + dl) * 16 + c];
        sc[r] = (j0 + 4 * g + r < n) ? (s4[r] + u) * l2e : -1e30f;
        tmax = fmaxf(tmax, sc[r]);
      }
      tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
      tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
      float mn = fmaxf(m[t], tmax);
      float corr = __builtin_amdgcn_exp2f(m[t] - mn);
      m[t] = mn;
      float p[4], ps = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) { p[r] = __builtin_amdgcn_exp2f(sc[r] - mn); ps += p[r]; }
      l[t] = l[t] * corr + ps;
#pragma unroll
      for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int r = 0; r < 4; ++r) o[t][e][r] *= corr;
      // O^T[d][query] += V^T[d][key] * P^T[key][query]; two accumulators break the dependent chain
      o[t][0] = MFMA16(vf[0], p[0], o[t][0]);
      o[t][1] = MFMA16(vf[1], p[1], o[t][1]);
      o[t][0] = MFMA16(vf[2], p[2], o[t][0]);
      o[t][1] = MFMA16(vf[3], p[3], o[t][1]);
    }
    slot_hi = slot_lo;
  }
  // finish: the 4 lane groups of a query hold partial sums of l; O^T rows d = 4g+r live in the registers
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    float lt = l[t];
    lt += __shfl_xor(lt, 16, 64);
    lt += __shfl_xor(lt, 32, 64);
    int qi = i0 + 16 * t + c;
    if (qi < n) {
      long tok = tok_of(a.g, seq, qi);
      float inv = 1.0f / lt;
      float4 ov = make_float4((o[t][0][0] + o[t][1][0]) * inv, (o[t][0][1] + o[t][1][1]) * inv,
                              (o[t][0][2] + o[t][1][2]) * inv, (o[t][0][3] + o[t][1][3]) * inv);
      *reinterpret_cast<float4*>(a.O + tok * 64 + head * 16 + 4 * g) = ov;
      if (g == 0 && a.LSE) a.LSE[tok * 4 + head] = (m[t] + log2f(lt)) * 0.6931471805599453f;
    }
  }
}

// delta[token][head] = sum_d dO * O  (softmax backward row constant)
__global__ void attn_delta_kernel(const float* __restrict__ dO, const float* __restrict__ O, float* __restrict__ D,
                                  long ntok) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;     // one thread per (token, head, quarter) -> 4 floats
  if (idx >= ntok * 16) return;
  float4 a = *reinterpret_cast<const float4*>(dO + idx * 4);
  float4 b = *reinterpret_cast<const float4*>(O + idx * 4);
  float s = a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
  s += __shfl_xor(s, 1, 64);
  s += __shfl_xor(s, 2, 64);
  if ((idx & 3) == 0) D[idx >> 2] = s;
}

struct AttnBwdArgs {
  AttnGeom g;
  const float* QKV; const float* E; const float* dO; const float* LSE; const float* Dl;
  float* dQKV;      // [tokens][192]; dq written by the dq kernel, dk|dv by the dkv kernel
  float* dE;        // [2*maxpos+1][16], accumulated with atomics
  int maxpos;
  float scale;
  int dbg;          // ablation switches for profiling builds (0 in production)
};

// ---- backward kernel 1: dK, dV.  One wave = (sequence, head, 32-key block); sweeps the queries 16 at a time.
// Orientation: key on the lane:  S[q][k] tile C-layout row = query 4g+r, col = key c.
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(AttnBwdArgs a) {
  __shared__ float Ul[4][2][3][16 * 16];     // [wave][key tile][-, hi, lo][query row][delta col]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = lane & 15, g = lane >> 4;
  const int n = a.g.n;
  const int kblocks = (n + 31) / 32;
  long item = (long)blockIdx.x * 4 + wave;
  const long nitems = (long)a.g.nseq * 4 * kblocks;
  if (item >= nitems) return;
  const int kb = (int)(item % kblocks);
  const int head = (int)((item / kblocks) % 4);
  const int seq = (int)(item / ((long)kblocks * 4));
  const int j0 = kb * 32;
  const float* qkv = a.QKV + head * 16;
  const float l2e = 1.4426950408889634f;

  // K^T / V^T fragments as B operands (lane (key c, kgrp g) supplies K[j][4s+g]) for both key tiles
  float kf[2][4], vf[2][4];
  bool kvalid[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    int kj = j0 + 16 * t + c;
    kvalid[t] = kj < n;
    if (kj > n - 1) kj = n - 1;
    const float* kp = qkv + tok_of(a.g, seq, kj) * 192;
#pragma unroll
    for (int s = 0; s < 4; ++s) { kf[t][s] = kp[64 + 4 * s + g]; vf[t][s] = kp[128 + 4 * s + g]; }
  }
  f32x4 dk[2], dv[2];     // dK^T / dV^T tiles: row d = 4g+r, col key c
#pragma unroll
  for (int t = 0; t < 2; ++t) { dk[t] = (f32x4){0.f, 0.f, 0.f, 0.f}; dv[t] = (f32x4){0.f, 0.f, 0.f, 0.f}; }

  // U_t(q0; D) = Q_tile . Ew(D)^T : rows = queries, cols = offsets delta in [D, D+15].
  // For query tile q0 and key tile t: D0 = q0 - (j0+16t); lane (key c), row query 4g+r needs
  // delta_local = (4g+r) - c in [-15,15] -> tile D0 (>=0) or D0-16 (<0).  The rows are the CURRENT queries,
  // so (unlike the forward, where the rows are offsets) both tiles are recomputed per step.
  auto u_tile = [&](const float (&qa)[4], int D, int t, int slot) {
    // B operand: E^T: lane (delta col c, kgrp g) supplies E[clamp(D+c)][4s+g]
    int d = D + c;
    d = d < -a.maxpos ? -a.maxpos : (d > a.maxpos ? a.maxpos : d);
    const float* ep = a.E + (long)(d + a.maxpos) * 16;
    f32x4 u = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 4; ++s) u = MFMA16(qa[s], ep[4 * s + g], u);
#pragma unroll
    for (int r = 0; r < 4; ++r) Ul[wave][t][slot][(4 * g + r) * 16 + c] = u[r];
  };
  const int nqt = (n + 15) / 16;
  for (int qt = 0; qt < nqt; ++qt) {
    const int q0 = qt * 16;
    // A operands: Q (lane (query row c, kgrp g) supplies Q[q0+c][4s+g]), dO likewise
    int qi = q0 + c; if (qi > n - 1) qi = n - 1;
    const long qtok = tok_of(a.g, seq, qi);
    float qa[4], doa[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) { qa[s] = qkv[qtok * 192 + 4 * s + g]; doa[s] = a.dO[qtok * 64 + head * 16 + 4 * s + g]; }
    // per-row (query 4g+r) constants
    float lse[4], dl[4];
    bool qvalid[4];
    // A^T operands for the dK/dV products: lane (d = c, k-index g) supplies X[q0 + 4g + r][d]
    float qT[4], doT[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      int qr = q0 + 4 * g + r;
      qvalid[r] = qr < n;
      if (qr > n - 1) qr = n - 1;
      long tk = tok_of(a.g, seq, qr);
      lse[r] = a.LSE[tk * 4 + head];
      dl[r] = a.Dl[tk * 4 + head];
      qT[r] = qkv[tk * 192 + c];
      doT[r] = a.dO[tk * 64 + head * 16 + c];
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int D0 = q0 - (j0 + 16 * t);
      u_tile(qa, D0, t, 1);
      u_tile(qa, D0 - 16, t, 2);
      f32x4 s4 = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 4; ++s) { s4 = MFMA16(qa[s], kf[t][s], s4); dp = MFMA16(doa[s], vf[t][s], dp); }
      float p[4], ds[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int dlc = (4 * g + r) - c;
        float u = dlc >= 0 ? Ul[wave][t][1][(4 * g + r) * 16 + dlc] : Ul[wave][t][2][(4 * g + r) * 16 + 16 + dlc];
        float sv = (s4[r] + u) * a.scale;
        bool ok = qvalid[r] && kvalid[t];
        p[r] = ok ? __builtin_amdgcn_exp2f((sv - lse[r]) * l2e) : 0.f;
        ds[r] = p[r] * (dp[r] - dl[r]) * a.scale;
      }
      // dV^T[d][key] += dO^T[d][q] * P[q][key];  dK^T[d][key] += Q^T[d][q] * dS[q][key]   (k-index g <-> query 4g+r)
#pragma unroll
      for (int r = 0; r < 4; ++r) { dv[t] = MFMA16(doT[r], p[r], dv[t]); dk[t] = MFMA16(qT[r], ds[r], dk[t]); }
    }
  }
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    int kj = j0 + 16 * t + c;
    if (kj < n) {
      long tok = tok_of(a.g, seq, kj);
      *reinterpret_cast<float4*>(a.dQKV + tok * 192 + 64 + head * 16 + 4 * g) = make_float4(dk[t][0], dk[t][1], dk[t][2], dk[t][3]);
      *reinterpret_cast<float4*>(a.dQKV + tok * 192 + 128 + head * 16 + 4 * g) = make_float4(dv[t][0], dv[t][1], dv[t][2], dv[t][3]);
    }
  }
}

// ---- backward kernel 2: dQ and dE.  One wave = (sequence, head, 16-query tile), query on the lane (as forward).
// dE is accumulated per workgroup in LDS over all the items the (persistent) workgroup processes and flushed
// once with global atomics.
template <int NPAD>
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(AttnBwdArgs a, int items_per_block) {
  extern __shared__ float smem[];
  // layout: dEacc [(2*NPAD) rows][16] | per-wave scratch: dU[2 slots... ] see below
  float* dEacc = smem;                                 // rows: delta + NPAD, delta in [-NPAD, NPAD)
  float* scratch = smem + 2 * NPAD * 16;               // [wave][4 tiles][256]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = lane & 15, g = lane >> 4;
  const int n = a.g.n;
  float* Uf = scratch + wave * 4 * 256;                // forward skew tiles: [2][256]
  float* dU = Uf + 2 * 256;                            // backward skew tiles: [2][256] (hi, lo)
  for (int i = threadIdx.x; i < 2 * NPAD * 16; i += 256) dEacc[i] = 0.f;
  __syncthreads();
  const int qtiles = (n + 15) / 16;
  const long nitems = (long)a.g.nseq * 4 * qtiles;
  const long ibeg = (long)blockIdx.x * items_per_block;
  const float l2e = 1.4426950408889634f;
  const int nkt = (n + 15) / 16;
  for (long it = ibeg + wave; it < ibeg + items_per_block && it < nitems; it += 4) {
    const int qt = (int)(it % qtiles);
    const int head = (int)((it / qtiles) % 4);
    const int seq = (int)(it / ((long)qtiles * 4));
    const int i0 = qt * 16;
    const float* qkv = a.QKV + head * 16;
    int qi = i0 + c;
    const bool qok = qi < n;
    if (qi > n - 1) qi = n - 1;
    const long qtok = tok_of(a.g, seq, qi);
    // B operands: Q^T, dO^T: lane (query c, kgrp g)
    float qf[4], dof[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) { qf[s] = qkv[qtok * 192 + 4 * s + g]; dof[s] = a.dO[qtok * 64 + head * 16 + 4 * s + g]; }
    // A operand for dE: Q^T with lane (d = c, k-index g) supplying Q[i0 + 4g + r][d]
    float qT[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      int qr = i0 + 4 * g + r; if (qr > n - 1) qr = n - 1;
      qT[r] = qkv[tok_of(a.g, seq, qr) * 192 + c];
    }
    const float lse = a.LSE[qtok * 4 + head], dlt = a.Dl[qtok * 4 + head];
    f32x4 dq = {0.f, 0.f, 0.f, 0.f};      // dQ^T tile: row d = 4g+r, col query c
    // zero the backward skew tiles
#pragma unroll
    for (int r = 0; r < 8; ++r) dU[r * 64 + lane] = 0.f;
    int hi = 0;    // dU slot holding offsets [D0, D0+15]
    for (int kt = 0; kt < nkt; ++kt) {
      const int j0 = kt * 16;
      const int D0 = i0 - j0;
      const int lo = hi ^ 1;
      // forward skew tiles U(D0), U(D0-16) for the logits
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        int d = D0 - 16 * h + c;
        d = d < -a.maxpos ? -a.maxpos : (d > a.maxpos ? a.maxpos : d);
        const float* ep = a.E + (long)(d + a.maxpos) * 16;
        f32x4 u = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 4; ++s) u = MFMA16(ep[4 * s + g], qf[s], u);
#pragma unroll
        for (int r = 0; r < 4; ++r) Uf[h * 256 + (4 * g + r) * 16 + c] = u[r];
      }
      int kj = j0 + c; if (kj > n - 1) kj = n - 1;
      const float* kp = qkv + tok_of(a.g, seq, kj) * 192;
      f32x4 s4 = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 4; ++s) { s4 = MFMA16(kp[64 + 4 * s + g], qf[s], s4); dp = MFMA16(kp[128 + 4 * s + g], dof[s], dp); }
      float ds[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int dlc = c - (4 * g + r);
        float u = dlc >= 0 ? Uf[dlc * 16 + c] : Uf[256 + (16 + dlc) * 16 + c];
        float sv = (s4[r] + u) * a.scale;
        bool ok = qok && (j0 + 4 * g + r < n);
        float p = ok ? __builtin_amdgcn_exp2f((sv - lse) * l2e) : 0.f;
        ds[r] = p * (dp[r] - dlt) * a.scale;
        // scatter into the offset-major tiles (each (row, query) cell is hit by exactly one key)
        if (dlc >= 0) dU[hi * 256 + dlc * 16 + c] += ds[r]; else dU[lo * 256 + (16 + dlc) * 16 + c] += ds[r];
      }
      // dQ^T[d][q] += K^T[d][key] * dS^T[key][q]   (A: lane (d=c, g) supplies K[j0+4g+r][d])
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int kr = j0 + 4 * g + r; if (kr > n - 1) kr = n - 1;
        dq = MFMA16(qkv[tok_of(a.g, seq, kr) * 192 + 64 + c], ds[r], dq);
      }
      // tile `hi` (offsets [D0, D0+15]) is now complete: consume it
      {
        // dQ^T[d][q] += E^T[d][delta] * dU[delta][q]   (A: lane (d=c, g) supplies E[clamp(D0+4g+r)][d])
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          int d = D0 + 4 * g + r;
          d = d < -a.maxpos ? -a.maxpos : (d > a.maxpos ? a.maxpos : d);
          dq = MFMA16(a.E[(long)(d + a.maxpos) * 16 + c], dU[hi * 256 + (4 * g + r) * 16 + c], dq);
        }
        // dE^T[d][delta] += Q^T[d][q] * dU^T[q][delta]   (B: lane (delta c, k-index g) supplies dU[delta c][q 4g+r])
        f32x4 de = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 4; ++r) de = MFMA16(qT[r], dU[hi * 256 + c * 16 + 4 * g + r], de);
        int d = D0 + c;
        d = d < -a.maxpos ? -a.maxpos : (d > a.maxpos ? a.maxpos : d);
        if (d >= -NPAD && d < NPAD) {
#pragma unroll
          for (int r = 0; r < 4; ++r) atomicAdd(&dEacc[(d + NPAD) * 16 + 4 * g + r], de[r]);
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) atomicAdd(&a.dE[(long)(d + a.maxpos) * 16 + 4 * g + r], de[r]);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) dU[hi * 256 + r * 64 + lane] = 0.f;
      }
      hi = lo;
    }
    // the last low tile (offsets [D0_last - 16, D0_last - 1]) still holds contributions
    {
      const int D0 = i0 - (nkt - 1) * 16 - 16;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int d = D0 + 4 * g + r;
        d = d < -a.maxpos ? -a.maxpos : (d > a.maxpos ? a.maxpos : d);
        dq = MFMA16(a.E[(long)(d + a.maxpos) * 16 + c], dU[hi * 256 + (4 * g + r) * 16 + c], dq);
      }
      f32x4 de = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int r = 0; r < 4; ++r) de = MFMA16(qT[r], dU[hi * 256 + c * 16 + 4 * g + r], de);
      int d = D0 + c;
      d = d < -a.maxpos ? -a.maxpos : (d > a.maxpos ? a.maxpos : d);
      if (d >= -NPAD && d < NPAD) {
#pragma unroll
        for (int r = 0; r < 4; ++r) atomicAdd(&dEacc[(d + NPAD) * 16 + 4 * g + r], de[r]);
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) atomicAdd(&a.dE[(long)(d + a.maxpos) * 16 + 4 * g + r], de[r]);
      }
    }
    if (qok) *reinterpret_cast<float4*>(a.dQKV + qtok * 192 + head * 16 + 4 * g) = make_float4(dq[0], dq[1], dq[2], dq[3]);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * NPAD * 16; i += 256) {
    float v = dEacc[i];
    int d = i / 16 - NPAD;
    if (v != 0.f && d >= -a.maxpos && d <= a.maxpos) atomicAdd(&a.dE[(long)(d + a.maxpos) * 16 + (i & 15)], v);
  }
}


// =====================================================================================================================
// v2 kernels: one workgroup = one (sequence, head); K / V of the head staged once in LDS and shared by the 4 waves,
// fragment loads are single 16-byte LDS / global accesses (contraction index d = 4g + s, so a lane's 4 MFMA steps read
// one contiguous float4), token address arithmetic hoisted out of the loops.
// =====================================================================================================================
// (sequence, head) item of workgroup b when ONE workgroup handles one item: workgroups are dealt round-robin to the 8 XCDs, so with
// item = b the four heads of a sequence -- which share every 128-B line of its QKV / dO rows (a head is 64 B of them) -- land on
// four different L2s and every line crosses the fabric twice.  Here the heads of a sequence are the workgroups b, b + 8, b + 16,
// b + 24: same XCD, dispatched together.  Measured (FETCH_SIZE): forward 0.86 -> see DESIGN.md; the tail (grid % 32) keeps item = b.
static __device__ __forceinline__ int xcd_item(int b, int nb) {
  if (b >= (nb & ~31) || SE_ATTN_NO_XCD_MAP) return b;
  return (((b >> 5) * 8 + (b & 7)) << 2) | ((b >> 3) & 3);
}
static __device__ __forceinline__ long seq_base(const AttnGeom& g, int s) {
  return (long)(s / g.inner) * g.outer_stride + (long)(s % g.inner) * g.inner_stride;
}
static __device__ __forceinline__ float f4c(const float4& v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : (i == 2 ? v.z : v.w)); }

// LDS: Ks[NP][16] | Vt[16][NP+4] | per-wave U ring [4][2 tiles][2 slots][256]
template <int TQ>      // 16-query tiles per wave block: 2 amortises the K / V fragment reads, 1 balances long sequences over 8 waves
__global__ __launch_bounds__(512) void attn_fwd2_kernel(AttnArgs a, int NP) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Ks = smem;
  float* Vt = Ks + NP * 16;
  const int VS = NP + 4;
  float* Ubase = Vt + 16 * VS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int NT = blockDim.x, NW = NT >> 6;      // waves per workgroup: launch parameter (LDS sized to match)
  const int c = lane & 15, g = lane >> 4;
  const int n = a.g.n;
  const int head = blockIdx.x & 3, seq = blockIdx.x >> 2;
  const long base = seq_base(a.g, seq), ps = a.g.pos_stride;
  const float* qkv = a.QKV + head * 16;
  for (int i = tid; i < NP * 4; i += NT) {
    int j = i >> 2, q = i & 3;
    float4 k4 = make_float4(0.f, 0.f, 0.f, 0.f), v4 = k4;
    if (j < n) {
      const float* p = qkv + (base + (long)j * ps) * 192;
      k4 = *reinterpret_cast<const float4*>(p + 64 + 4 * q);
      v4 = *reinterpret_cast<const float4*>(p + 128 + 4 * q);
    }
    *reinterpret_cast<float4*>(&Ks[j * 16 + 4 * q]) = k4;
    Vt[(4 * q + 0) * VS + j] = v4.x; Vt[(4 * q + 1) * VS + j] = v4.y;
    Vt[(4 * q + 2) * VS + j] = v4.z; Vt[(4 * q + 3) * VS + j] = v4.w;
  }
  __syncthreads();
  float* Ul = Ubase + wave * (TQ * 512);      // [tile t][slot][256]
  const float l2e = 1.4426950408889634f * a.scale;
  const int qblocks = (n + 16 * TQ - 1) / (16 * TQ), nkt = (n + 15) / 16;
  for (int qb = wave; qb < qblocks; qb += NW) {
    const int i0 = qb * 16 * TQ;
    float4 qf[TQ];
#pragma unroll
    for (int t = 0; t < TQ; ++t) {
      int qi = i0 + 16 * t + c; if (qi > n - 1) qi = n - 1;
      qf[t] = *reinterpret_cast<const float4*>(qkv + (base + (long)qi * ps) * 192 + 4 * g);
    }
    f32x4 o[TQ][2];
    float m[TQ], l[TQ];
#pragma unroll
    for (int t = 0; t < TQ; ++t) {
      m[t] = -1e30f; l[t] = 0.f;
      o[t][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; o[t][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    auto u_tile = [&](int D, int t, int slot) {
      int d = D + c;
      d = d < -a.maxpos ? -a.maxpos : (d > a.maxpos ? a.maxpos : d);
      float4 e = *reinterpret_cast<const float4*>(a.E + (long)(d + a.maxpos) * 16 + 4 * g);
      f32x4 u = {0.f, 0.f, 0.f, 0.f};
      u = MFMA16(e.x, qf[t].x, u); u = MFMA16(e.y, qf[t].y, u); u = MFMA16(e.z, qf[t].z, u); u = MFMA16(e.w, qf[t].w, u);
#pragma unroll
      for (int r = 0; r < 4; ++r) Ul[(t * 2 + slot) * 256 + (4 * g + r) * 16 + c] = u[r];
    };
#pragma unroll
    for (int t = 0; t < TQ; ++t) u_tile(i0 + 16 * t, t, 0);
    int hi = 0;
    for (int kt = 0; kt < nkt; ++kt) {
      const int j0 = kt * 16, lo = hi ^ 1;
#pragma unroll
      for (int t = 0; t < TQ; ++t) u_tile(i0 + 16 * t - j0 - 16, t, lo);
      float4 kf = *reinterpret_cast<const float4*>(&Ks[(j0 + c) * 16 + 4 * g]);
      float4 vf = *reinterpret_cast<const float4*>(&Vt[c * VS + j0 + 4 * g]);
#pragma unroll
      for (int t = 0; t < TQ; ++t) {
        f32x4 s4 = {0.f, 0.f, 0.f, 0.f};
        s4 = MFMA16(kf.x, qf[t].x, s4); s4 = MFMA16(kf.y, qf[t].y, s4);
        s4 = MFMA16(kf.z, qf[t].z, s4); s4 = MFMA16(kf.w, qf[t].w, s4);
        float sc[4], tmax = -1e30f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          int dl = c - (4 * g + r);
          float u = dl >= 0 ? Ul[(t * 2 + hi) * 256 + dl * 16 + c] : Ul[(t * 2 + lo) * 256 + (16 + dl) * 16 + c];
          sc[r] = (j0 + 4 * g + r < n) ? (s4[r] + u) * l2e : -1e30f;
          tmax = fmaxf(tmax, sc[r]);
        }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        float mn = fmaxf(m[t], tmax);
        float corr = __builtin_amdgcn_exp2f(m[t] - mn);
        m[t] = mn;
        float p[4], psum = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) { p[r] = __builtin_amdgcn_exp2f(sc[r] - mn); psum += p[r]; }
        l[t] = l[t] * corr + psum;
#pragma unroll
        for (int r = 0; r < 4; ++r) { o[t][0][r] *= corr; o[t][1][r] *= corr; }
        o[t][0] = MFMA16(vf.x, p[0], o[t][0]);
        o[t][1] = MFMA16(vf.y, p[1], o[t][1]);
        o[t][0] = MFMA16(vf.z, p[2], o[t][0]);
        o[t][1] = MFMA16(vf.w, p[3], o[t][1]);
      }
      hi = lo;
    }
#pragma unroll
    for (int t = 0; t < TQ; ++t) {
      float lt = l[t];
      lt += __shfl_xor(lt, 16, 64);
      lt += __shfl_xor(lt, 32, 64);
      int qi = i0 + 16 * t + c;
      if (qi < n) {
        long tok = base + (long)qi * ps;
        float inv = 1.0f / lt;
        *reinterpret_cast<float4*>(a.O + tok * 64 + head * 16 + 4 * g) =
            make_float4((o[t][0][0] + o[t][1][0]) * inv, (o[t][0][1] + o[t][1][1]) * inv,
                        (o[t][0][2] + o[t][1][2]) * inv, (o[t][0][3] + o[t][1][3]) * inv);
        if (g == 0 && a.LSE) a.LSE[tok * 4 + head] = (m[t] + log2f(lt)) * 0.6931471805599453f;
      }
    }
  }
}

// Single-pass backward: one workgroup walks (sequence, head) items; per item K, V are staged in LDS, each wave owns
// 16-query tiles (dQ in registers, sliding rel-pos window as in the forward), dK / dV are accumulated in LDS with
// ds_add_f32 and written once per item, dE is accumulated in LDS over ALL items of the workgroup and flushed once with
// global atomics.  P^T / dS^T tiles are transposed through wave-private LDS to serve as the B operands of the
// dV / dK products.  Requires |i - j| <= maxpos for all pairs (no clamp aliasing): NP <= maxpos.
// LDS (floats): Ks[NP][16] Vs[NP][16] | dKa[16][NP+4] dVa[16][NP+4] dEa[16][2NP+4] (transposed: the 16 lanes of a
// lane group add to 16 consecutive words, row strides = 4 mod 8 -> conflict-free ds_add) | scratch[4][6 tiles][16][20]
// NW = waves per workgroup (8 for the long time-axis sequences: two waves per SIMD hide each other's LDS / MFMA
// latency chains; the LDS budget then only allows K to be staged, V fragments come from global/L2), SV = V staged.
template <int NP, int NW, bool SV>
__global__ __launch_bounds__(NW * 64) void attn_bwd2_kernel(AttnBwdArgs a, const float* __restrict__ Et, int ET, int items_per_block) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int LK = NP + 4, LE = 2 * NP + 4, TS = 20, TILE = 16 * TS, NT = NW * 64;
  float* Ks = smem;
  float* Vs = Ks + NP * 16;
  float* dKa = Vs + (SV ? NP * 16 : 0);
  float* dVa = dKa + 16 * LK;
  float* dEa = dVa + 16 * LK;
  float* scratch = dEa + 16 * LE;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 15, g = lane >> 4;
  const int n = a.g.n;
  float* Uf = scratch + wave * 5 * TILE;   // [2 slots]; the consumed `hi` slot doubles as the P^T tile
  float* dU = Uf + 2 * TILE;               // [2]
  float* dSl = dU + 2 * TILE;              // [key][query]
  for (int i = tid; i < 16 * LE; i += NT) dEa[i] = 0.f;
  const long nitems = (long)a.g.nseq * 4;
  const long ibeg = (long)blockIdx.x * items_per_block;
  const int qtiles = (n + 15) / 16, nkt = (n + 15) / 16;
  const float l2e = 1.4426950408889634f;
  const long ps = a.g.pos_stride;
  for (long it = ibeg; it < ibeg + items_per_block && it < nitems; ++it) {
    const int head = (int)(it & 3), seq = (int)(it >> 2);
    const long base = seq_base(a.g, seq);
    const float* qkv = a.QKV + head * 16;
    __syncthreads();
    for (int i = tid; i < NP * 4; i += NT) {
      int j = i >> 2, q = i & 3;
      float4 k4 = make_float4(0.f, 0.f, 0.f, 0.f), v4 = k4;
      if (j < n) {
        const float* p = qkv + (base + (long)j * ps) * 192;
        k4 = *reinterpret_cast<const float4*>(p + 64 + 4 * q);
        if (SV) v4 = *reinterpret_cast<const float4*>(p + 128 + 4 * q);
      }
      *reinterpret_cast<float4*>(&Ks[i * 4]) = k4;
      if (SV) *reinterpret_cast<float4*>(&Vs[i * 4]) = v4;
    }
    for (int i = tid; i < 16 * LK; i += NT) { dKa[i] = 0.f; dVa[i] = 0.f; }
    __syncthreads();
    // Lock-step schedule instead of LDS atomics (ds_add_f32 costs ~500 cycles per wave-instruction here): in every
    // step the 4 waves work on 4 DIFFERENT key tiles (wave w starts its key sweep at tile 2w and wraps), so their
    // read-modify-writes of dKa / dVa (indexed by key tile) and dEa (indexed by the offset tile q - k, distinct
    // because (s_w' - s_w) mod nkt != w' - w) never overlap; one barrier per phase orders successive steps.
    // start offsets 2w need 2 (NW-1) < nkt; tiny sequences: a single wave (no conflicts possible)
    const int nwact = nkt >= 2 * NW - 1 ? NW : (nkt >= 7 ? 4 : 1);
    const int rounds = (qtiles + nwact - 1) / nwact;
    const int sw = 2 * wave;
    for (int round = 0; round < rounds; ++round) {
      const int qt = wave + nwact * round;
      const bool active = wave < nwact && qt < qtiles;
      const int i0 = qt * 16;
      int qi = i0 + c;
      const bool qok = active && qi < n;
      if (qi > n - 1) qi = n - 1;
      if (qi < 0) qi = 0;
      const long qtok = base + (long)qi * ps;
      float4 qf = make_float4(0.f, 0.f, 0.f, 0.f), dof = qf;
      float qT[4] = {0.f, 0.f, 0.f, 0.f}, doT[4] = {0.f, 0.f, 0.f, 0.f};
      float lse = 0.f, dlt = 0.f;
      if (active) {
        qf = *reinterpret_cast<const float4*>(qkv + qtok * 192 + 4 * g);
        dof = *reinterpret_cast<const float4*>(a.dO + qtok * 64 + head * 16 + 4 * g);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          int qr = i0 + 4 * g + r; if (qr > n - 1) qr = n - 1;
          long tk = base + (long)qr * ps;
          qT[r] = qkv[tk * 192 + c];
          doT[r] = a.dO[tk * 64 + head * 16 + c];
        }
        lse = a.LSE[qtok * 4 + head];
        dlt = a.Dl[qtok * 4 + head];
      }
      f32x4 dq = {0.f, 0.f, 0.f, 0.f};
      auto u_tile = [&](int D, int slot) {
        float4 e = *reinterpret_cast<const float4*>(a.E + (long)(D + c + a.maxpos) * 16 + 4 * g);
        f32x4 u = {0.f, 0.f, 0.f, 0.f};
        u = MFMA16(e.x, qf.x, u); u = MFMA16(e.y, qf.y, u); u = MFMA16(e.z, qf.z, u); u = MFMA16(e.w, qf.w, u);
#pragma unroll
        for (int r = 0; r < 4; ++r) Uf[slot * TILE + (4 * g + r) * TS + c] = u[r];
      };
      // consume a finished offset tile (offsets [D, D+15]): dQ += E^T dU ; dE^T += Q^T dU^T (plain RMW, see above)
      auto consume = [&](int D, int slot) {
        float4 et = *reinterpret_cast<const float4*>(Et + (long)c * ET + (D + 4 * g + a.maxpos));   // E[D+4g+r][d=c]
        dq = MFMA16(et.x, dU[slot * TILE + (4 * g + 0) * TS + c], dq);
        dq = MFMA16(et.y, dU[slot * TILE + (4 * g + 1) * TS + c], dq);
        dq = MFMA16(et.z, dU[slot * TILE + (4 * g + 2) * TS + c], dq);
        dq = MFMA16(et.w, dU[slot * TILE + (4 * g + 3) * TS + c], dq);
        float4 du = *reinterpret_cast<const float4*>(&dU[slot * TILE + c * TS + 4 * g]);       // dU[delta c][q 4g+r]
        f32x4 de = {0.f, 0.f, 0.f, 0.f};
        de = MFMA16(qT[0], du.x, de); de = MFMA16(qT[1], du.y, de); de = MFMA16(qT[2], du.z, de); de = MFMA16(qT[3], du.w, de);
        int row = D + c + NP;                 // de[r] = dE[delta = D + c][d = 4g + r]
        if (row >= 0 && row < 2 * NP) {
#pragma unroll
          for (int r = 0; r < 4; ++r) dEa[(4 * g + r) * LE + row] += de[r];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) { dU[slot * TILE + (4 * g + r) * TS + c] = 0.f; }
      };
      int hi = 0;
      // ONE barrier per step: between two barriers all waves are in the same step, where they own distinct key tiles
      // and distinct offset tiles (see above).  The last low offset tile of a run (16 (qt - kt - 1)) is consumed after
      // the NEXT barrier: there it can only meet offset tiles 16 (qt' - kt') of step t + 1, and
      // qt - (t + 2w) - 1 == qt' - (t + 1 + 2w') forces w == w' (same wave, program order).
      bool pend = false;
      int pendD = 0;
      for (int t = 0; t < nkt; ++t) {
        int kt = t + sw;
        if (kt >= nkt) kt -= nkt;
        if (nwact == 1) kt = t;
        const int j0 = kt * 16, D0 = i0 - j0;
        __syncthreads();
        if (pend) { consume(pendD, hi); pend = false; }
        if (active) {
          if (t == 0 || kt == 0) {           // start of a monotone run of key tiles: prime the window
#pragma unroll
            for (int r = 0; r < 10; ++r) dU[r * 64 + lane] = 0.f;
            hi = 0;
            u_tile(D0, 0);
          }
          const int lo = hi ^ 1;
          u_tile(D0 - 16, lo);
          const float4 kf = *reinterpret_cast<const float4*>(&Ks[(j0 + c) * 16 + 4 * g]);
          float4 vf;
          if (SV) vf = *reinterpret_cast<const float4*>(&Vs[(j0 + c) * 16 + 4 * g]);
          else { int kj = j0 + c; if (kj > n - 1) kj = n - 1;
                 vf = *reinterpret_cast<const float4*>(qkv + (base + (long)kj * ps) * 192 + 128 + 4 * g); }
          f32x4 s4 = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
          s4 = MFMA16(kf.x, qf.x, s4); dp = MFMA16(vf.x, dof.x, dp);
          s4 = MFMA16(kf.y, qf.y, s4); dp = MFMA16(vf.y, dof.y, dp);
          s4 = MFMA16(kf.z, qf.z, s4); dp = MFMA16(vf.z, dof.z, dp);
          s4 = MFMA16(kf.w, qf.w, s4); dp = MFMA16(vf.w, dof.w, dp);
          float ds[4], uu[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {            // all skew reads first: the `hi` slot is recycled as Pl below
            int dlc = c - (4 * g + r);
            uu[r] = dlc >= 0 ? Uf[hi * TILE + dlc * TS + c] : Uf[lo * TILE + (16 + dlc) * TS + c];
          }
          float* Pl = Uf + hi * TILE;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            int dlc = c - (4 * g + r);
            float u = uu[r];
            float sv = (s4[r] + u) * a.scale;
            bool ok = qok && (j0 + 4 * g + r < n);
            float p = ok ? __builtin_amdgcn_exp2f((sv - lse) * l2e) : 0.f;
            ds[r] = p * (dp[r] - dlt) * a.scale;
            if (dlc >= 0) dU[hi * TILE + dlc * TS + c] += ds[r]; else dU[lo * TILE + (16 + dlc) * TS + c] += ds[r];
            Pl[(4 * g + r) * TS + c] = p;
            dSl[(4 * g + r) * TS + c] = ds[r];
          }
          // dQ^T[d][q] += K^T[d][key] dS^T[key][q]   (A: lane (d=c, g) supplies K[j0+4g+r][c])
#pragma unroll
          for (int r = 0; r < 4; ++r) dq = MFMA16(Ks[(j0 + 4 * g + r) * 16 + c], ds[r], dq);
          // dV^T[d][key] += dO^T[d][q] P[q][key];  dK^T[d][key] += Q^T[d][q] dS[q][key]
          // B: lane (key c, k-index g) supplies X[q 4g+r][key c] = Xl[key c][4g + r]
          const float4 pb = *reinterpret_cast<const float4*>(&Pl[c * TS + 4 * g]);
          const float4 sb = *reinterpret_cast<const float4*>(&dSl[c * TS + 4 * g]);
          f32x4 dv = {0.f, 0.f, 0.f, 0.f}, dk = {0.f, 0.f, 0.f, 0.f};
          dv = MFMA16(doT[0], pb.x, dv); dk = MFMA16(qT[0], sb.x, dk);
          dv = MFMA16(doT[1], pb.y, dv); dk = MFMA16(qT[1], sb.y, dk);
          dv = MFMA16(doT[2], pb.z, dv); dk = MFMA16(qT[2], sb.z, dk);
          dv = MFMA16(doT[3], pb.w, dv); dk = MFMA16(qT[3], sb.w, dk);
#pragma unroll
          for (int r = 0; r < 4; ++r) {      // dv[r] = dV[key j0+c][d = 4g+r]; this wave owns key tile kt in this step
            dVa[(4 * g + r) * LK + j0 + c] += dv[r];
            dKa[(4 * g + r) * LK + j0 + c] += dk[r];
          }
          consume(D0, hi);
          hi = lo;
          if (kt == nkt - 1 || t == nkt - 1) { pend = true; pendD = D0 - 16; }   // end of a run: the last low tile
        }
      }
      __syncthreads();
      if (pend) consume(pendD, hi);
      if (qok) *reinterpret_cast<float4*>(a.dQKV + qtok * 192 + head * 16 + 4 * g) = make_float4(dq[0], dq[1], dq[2], dq[3]);
    }
    __syncthreads();
    for (int i = tid; i < NP * 4; i += NT) {
      int j = i >> 2, q = i & 3;
      if (j < n) {
        float* p = a.dQKV + (base + (long)j * ps) * 192 + head * 16 + 4 * q;
        *reinterpret_cast<float4*>(p + 64) = make_float4(dKa[(4 * q) * LK + j], dKa[(4 * q + 1) * LK + j],
                                                         dKa[(4 * q + 2) * LK + j], dKa[(4 * q + 3) * LK + j]);
        *reinterpret_cast<float4*>(p + 128) = make_float4(dVa[(4 * q) * LK + j], dVa[(4 * q + 1) * LK + j],
                                                          dVa[(4 * q + 2) * LK + j], dVa[(4 * q + 3) * LK + j]);
      }
    }
  }
  __syncthreads();
  for (int i = tid; i < 16 * 2 * NP; i += NT) {
    int dch = i / (2 * NP), row = i - dch * (2 * NP);
    float v = dEa[dch * LE + row];
    int d = row - NP;
    if (v != 0.f && d >= -a.maxpos && d <= a.maxpos) atomicAdd(&a.dE[(long)(d + a.maxpos) * 16 + dch], v);
  }
}


// =====================================================================================================================
// v3 backward: decoupled waves, key-stationary, split-bf16 MFMA.
//
// Why: the v2 kernel keeps dK / dV / dE in LDS accumulators shared by the waves of a workgroup; LDS float atomics run at one
// lane per ~3 cycles on gfx950 (ds_add_f32: 192 cycles per wave-instruction, tools/micro/lds_atomic_bench.hip), so v2 uses
// plain read-modify-writes under a lock-step schedule with one barrier per 16x16 tile pair -- all waves of a CU sit in the
// same phase (MFMA, then LDS, then VALU) and nothing overlaps: 0.24 of the fp32-MFMA peak.
//
// Here ONE WAVE owns a work item (sequence, head, group of KT3 = 7 key tiles = 112 keys) and there is no barrier and no shared
// accumulator in the main loop:
//   * dK^T / dV^T of its 7 key tiles stay in registers (key loop unrolled: static indices) and are stored once;
//   * the wave sweeps the query tiles; dQ^T of a query tile is summed over the wave's 7 key tiles in registers and then
//     either stored (one key group covers the sequence: n <= 112) or added to global memory with fp32 atomics (3 adds per
//     element for n = 321: 0.4 GB of atomic bytes per launch, overlapped with compute, well under the chip's 1.3 TB/s);
//   * the relative-position terms use a per-query-tile OFFSET STRIP in wave-private LDS: U[q][delta] = q . E[delta] for the
//     8 offset tiles the 7 key tiles touch is written once per query tile, every step reads its skewed 16x16 window and
//     overwrites it IN PLACE with dS (each (q, delta) cell belongs to exactly one key), and the finished strip W = skew(dS)
//     feeds  dQ^T += E^T W^T  and  dE^T += Q^T W  tile by tile;
//   * dE accumulators live in a REGISTER WINDOW of 8 offset tiles that slides with the query tile (v_mov rotation); the
//     tile leaving the window is flushed with global atomics into one of NREP replicas of the table (summed at the end).
// Every product is an exact 3-way bf16 split evaluated with THREE v_mfma_f32_16x16x32_bf16: the contraction length is 16, so
// the two halves of K = 32 carry two different split pairs (k slot (g, j): j < 4 -> first pair, j >= 4 -> second pair, index
// 4g + (j & 3)):  [a_hi|a_mid].[b_lo|b_mid] + [a_hi|a_lo].[b_mid|b_hi] + [a_hi|a_mid].[b_hi|b_hi]  = the six products of the
// fp32-equivalent split (dropped terms <= 2^-24 relative) at 48 instead of 128 matrix-pipe cycles per 16x16x16 product.
// K is staged once per item, pre-split, in wave-private LDS: its row fragments are ds_read_b64, its column fragments (for
// dQ^T += K^T dS^T) come from the same image through ds_read_b64_tr_b16.
// =====================================================================================================================
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x6 __attribute__((ext_vector_type(6)));
// mid / hi / lo bf16 planes of 4 fp32 values, packed two per dword (element j in half j & 1 of a plane's word j >> 1), kept
// as ONE 192-bit register run {m0 m1 h0 h1 l0 l1}: the two 4-dword windows [m|h] (words 0..3) and [h|l] (words 2..5) are the
// K = 32 operands of the three MFMAs below for an A operand AND for a B operand, so no operand is ever re-assembled with moves
struct S3 { u32x6 v; };
static __device__ __forceinline__ bf16x8 win_mh(const S3& s) { return __builtin_bit_cast(bf16x8, __builtin_shufflevector(s.v, s.v, 0, 1, 2, 3)); }
static __device__ __forceinline__ bf16x8 win_hl(const S3& s) { return __builtin_bit_cast(bf16x8, __builtin_shufflevector(s.v, s.v, 2, 3, 4, 5)); }
static __device__ __forceinline__ void set_m(S3& s, u32x2 p) { s.v[0] = p[0]; s.v[1] = p[1]; }
static __device__ __forceinline__ void set_h(S3& s, u32x2 p) { s.v[2] = p[0]; s.v[3] = p[1]; }
static __device__ __forceinline__ void set_l(S3& s, u32x2 p) { s.v[4] = p[0]; s.v[5] = p[1]; }
static __device__ __forceinline__ u32x2 get_m(const S3& s) { return (u32x2){s.v[0], s.v[1]}; }
static __device__ __forceinline__ u32x2 get_h(const S3& s) { return (u32x2){s.v[2], s.v[3]}; }
static __device__ __forceinline__ u32x2 get_l(const S3& s) { return (u32x2){s.v[4], s.v[5]}; }

static __device__ __forceinline__ unsigned pk_bf16(float a, float b) {        // one v_cvt_pk_bf16_f32 (RNE)
  f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
static __device__ __forceinline__ float bf_lo(unsigned w) { return __builtin_bit_cast(float, w << 16); }
static __device__ __forceinline__ float bf_hi(unsigned w) { return __builtin_bit_cast(float, w & 0xffff0000u); }
// exact 3-way split x = hi + mid + lo (8 + 8 + 8 mantissa bits): 18 VALU instructions for 4 values
static __device__ __forceinline__ S3 split3(float x0, float x1, float x2, float x3) {
  S3 s;
  s.v[2] = pk_bf16(x0, x1); s.v[3] = pk_bf16(x2, x3);
  x0 -= bf_lo(s.v[2]); x1 -= bf_hi(s.v[2]); x2 -= bf_lo(s.v[3]); x3 -= bf_hi(s.v[3]);
  s.v[0] = pk_bf16(x0, x1); s.v[1] = pk_bf16(x2, x3);
  x0 -= bf_lo(s.v[0]); x1 -= bf_hi(s.v[0]); x2 -= bf_lo(s.v[1]); x3 -= bf_hi(s.v[1]);
  s.v[4] = pk_bf16(x0, x1); s.v[5] = pk_bf16(x2, x3);
  return s;
}
static __device__ __forceinline__ S3 split3(const float4& v) { return split3(v.x, v.y, v.z, v.w); }
static __device__ __forceinline__ S3 split3(const f32x4& v) { return split3(v[0], v[1], v[2], v[3]); }
#define MFMA_BF(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)
// acc += A . B over the 16-long contraction whose index 4*kgroup + j (j = 0..3) the lane holds in a / b:
// [m|h].[h|l] + [h|l].[m|h] + [m|h].[m|h] = mh + hl + hm + lh + mm + hh, the six products of the fp32-equivalent split
static __device__ __forceinline__ f32x4 prod3(const S3& a, const S3& b, f32x4 acc) {
  acc = MFMA_BF(win_mh(a), win_hl(b), acc);
  acc = MFMA_BF(win_hl(a), win_mh(b), acc);
  acc = MFMA_BF(win_mh(a), win_mh(b), acc);
  return acc;
}
// two independent products interleaved (their accumulator chains hide each other's MFMA latency)
static __device__ __forceinline__ void prod3x2(const S3& a1, const S3& b1, f32x4& c1, const S3& a2, const S3& b2, f32x4& c2) {
  c1 = MFMA_BF(win_mh(a1), win_hl(b1), c1);
  c2 = MFMA_BF(win_mh(a2), win_hl(b2), c2);
  c1 = MFMA_BF(win_hl(a1), win_mh(b1), c1);
  c2 = MFMA_BF(win_hl(a2), win_mh(b2), c2);
  c1 = MFMA_BF(win_mh(a1), win_mh(b1), c1);
  c2 = MFMA_BF(win_mh(a2), win_mh(b2), c2);
}
// ---- scaled split-fp16 form of the same products (round 3) -----------------------------------------------------------------
// An operand is x * 2^sexp (exact; the tensor's largest magnitude in [2^13, 2^14)) = hi + lo with two fp16 words per value pair:
// the S3 run keeps hi in words 2..3 and lo in words 4..5 (word 0..1 unused).  A product is THREE v_mfma_f32_16x16x16_f16 (lo.hi,
// hi.lo, hi.hi: the K = 16 form has the lane layout of one half of the K = 32 form and costs the same 16 matrix-pipe cycles --
// tools/micro/mfma_rate.hip -- so the matrix pipe does what it did) but the split of 4 values is 8 VALU instructions instead of
// 18, and every LDS image / table has two planes instead of three.  The kernels are VALU-issue-bound (49 - 97 % busy), 60 % of it
// splits.  FP16_OVFL clamps instead of producing inf (unreachable: every scale comes from a measured maximum or a proven bound).
typedef _Float16 f16x4a __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2a __attribute__((ext_vector_type(2)));
static __device__ __forceinline__ unsigned pk_f16a(float a, float b) {            // one v_cvt_pk_f16_f32 (round to nearest even)
  f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2a));
}
static __device__ __forceinline__ void f16_clamp_mode_a() { asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1"); }
static __host__ __device__ __forceinline__ int f16_sexp_a(float amax) {            // amax * 2^sexp in [2^13, 2^14) (as se_gemm_dev.h)
  unsigned u;
  __builtin_memcpy(&u, &amax, 4);
  const int e = (int)((u >> 23) & 0xffu) - 127;
  const int sx = 13 - e;
  return !(amax > 0.f) ? 0 : (sx < -60 ? -60 : (sx > 60 ? 60 : sx));
}
static __device__ __forceinline__ float exp2ia(int e) { return __builtin_bit_cast(float, (unsigned)(127 + e) << 23); }
// (the packed words stay scalar values until the end: reading a word back out of a partially built ext-vector made hipcc 7.2 feed
// v_fma_mix_f32 from the wrong register -- se_gemm_dev.h split_planes8_h)
static __device__ __forceinline__ S3 split2h(float x0, float x1, float x2, float x3) {      // inputs already scaled
  const unsigned h0 = pk_f16a(x0, x1), h1 = pk_f16a(x2, x3);
  const f16x2a a = __builtin_bit_cast(f16x2a, h0), b = __builtin_bit_cast(f16x2a, h1);
  x0 -= (float)a[0]; x1 -= (float)a[1]; x2 -= (float)b[0]; x3 -= (float)b[1];
  const unsigned l0 = pk_f16a(x0, x1), l1 = pk_f16a(x2, x3);
  S3 s;
  s.v = (u32x6){0u, 0u, h0, h1, l0, l1};
  return s;
}
// F16: x * sc split into (hi, lo) fp16; otherwise the exact three-way bf16 split (sc is 1 there and ignored)
template <bool F16>
static __device__ __forceinline__ S3 splitx(float x0, float x1, float x2, float x3, float sc) {
  if constexpr (F16) return split2h(x0 * sc, x1 * sc, x2 * sc, x3 * sc);
  else return split3(x0, x1, x2, x3);
}
template <bool F16> static __device__ __forceinline__ S3 splitx(const float4& v, float sc) { return splitx<F16>(v.x, v.y, v.z, v.w, sc); }
// operands that are already at their scale (P with its 2^13 in the exponent argument, dS, W): no multiply (x * 1.0f is not dropped)
template <bool F16> static __device__ __forceinline__ S3 splitn(const f32x4& v) {
  if constexpr (F16) return split2h(v[0], v[1], v[2], v[3]);
  else return split3(v[0], v[1], v[2], v[3]);
}
template <bool F16> static __device__ __forceinline__ S3 splitn(const float4& v) {
  if constexpr (F16) return split2h(v.x, v.y, v.z, v.w);
  else return split3(v.x, v.y, v.z, v.w);
}
template <bool F16> static __device__ __forceinline__ S3 splitx(const f32x4& v, float sc) { return splitx<F16>(v[0], v[1], v[2], v[3], sc); }
#define MFMA_HF(a, b, c) __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(f16x4a, (a)), __builtin_bit_cast(f16x4a, (b)), (c), 0, 0, 0)
template <bool F16>
static __device__ __forceinline__ f32x4 prodx(const S3& a, const S3& b, f32x4 acc) {
  if constexpr (F16) {
    acc = MFMA_HF(get_l(a), get_h(b), acc);
    acc = MFMA_HF(get_h(a), get_l(b), acc);
    acc = MFMA_HF(get_h(a), get_h(b), acc);
    return acc;
  } else {
    return prod3(a, b, acc);
  }
}
template <bool F16>
static __device__ __forceinline__ void prodx2(const S3& a1, const S3& b1, f32x4& c1, const S3& a2, const S3& b2, f32x4& c2) {
  if constexpr (F16) {
    c1 = MFMA_HF(get_l(a1), get_h(b1), c1);
    c2 = MFMA_HF(get_l(a2), get_h(b2), c2);
    c1 = MFMA_HF(get_h(a1), get_l(b1), c1);
    c2 = MFMA_HF(get_h(a2), get_l(b2), c2);
    c1 = MFMA_HF(get_h(a1), get_h(b1), c1);
    c2 = MFMA_HF(get_h(a2), get_h(b2), c2);
  } else {
    prod3x2(a1, b1, c1, a2, b2, c2);
  }
}
static __device__ __forceinline__ u32x2 ld8(const void* p) { return *reinterpret_cast<const u32x2*>(p); }
static __device__ __forceinline__ void st8(void* p, u32x2 v) { *reinterpret_cast<u32x2*>(p) = v; }
// hardware-transposed LDS read (ds_read_b64_tr_b16): within each 16-lane group, lane 4q + p supplies the address of row q,
// columns 4p..4p+3 of a 4 x 16 block of 16-bit elements; lane i receives column i of the 4 rows.  EXEC must be full.
static __device__ __forceinline__ u32x2 tr8(const unsigned char* lds_ptr) {
  return __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (s16x4 __attribute__((address_space(3)))*)(lds_ptr)));
}

// the planes of one split operand in an image whose planes are `pb` bytes apart: (hi, mid, lo) bf16 or (hi, lo) fp16
template <bool F16>
static __device__ __forceinline__ void st_planes(unsigned char* p, int pb, const S3& s) {
  st8(p, get_h(s));
  if constexpr (F16) st8(p + pb, get_l(s));
  else { st8(p + pb, get_m(s)); st8(p + 2 * pb, get_l(s)); }
}
template <bool F16>
static __device__ __forceinline__ S3 ld_planes(const unsigned char* p, long pb) {
  S3 s;
  s.v = (u32x6){0u, 0u, 0u, 0u, 0u, 0u};
  set_h(s, ld8(p));
  if constexpr (F16) set_l(s, ld8(p + pb));
  else { set_m(s, ld8(p + pb)); set_l(s, ld8(p + 2 * pb)); }
  return s;
}
template <bool F16>
static __device__ __forceinline__ S3 tr_planes(const unsigned char* p, int pb) {
  S3 s;
  s.v = (u32x6){0u, 0u, 0u, 0u, 0u, 0u};
  set_h(s, tr8(p));
  if constexpr (F16) set_l(s, tr8(p + pb));
  else { set_m(s, tr8(p + pb)); set_l(s, tr8(p + 2 * pb)); }
  return s;
}

constexpr int NREP3 = 1;   // (unused by the scratch-based dE reduction; kept for the workspace layout)

// split tables of the relative-position embedding: Es[3][R][16] (row fragments), Ets[3][ET/16 tiles][16 d][16 offsets]
// (column fragments; tile-major so that one offset tile is 512 contiguous bytes per plane: 4 cache lines instead of 16)
__global__ void attn_split_tables_kernel(const float* __restrict__ E, __bf16* __restrict__ Es, __bf16* __restrict__ Ets,
                                         int R, int ET) {
  int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= R * 16) return;
  int row = idx >> 4, d = idx & 15;
  float x = E[idx];
#pragma unroll
  for (int pl = 0; pl < 3; ++pl) {
    __bf16 h = (__bf16)x;
    x -= (float)h;
    Es[((long)pl * R + row) * 16 + d] = h;
    Ets[(long)pl * ET * 16 + ((long)(row >> 4) * 16 + d) * 16 + (row & 15)] = h;      // [plane][tile][d][16 offsets]
  }
}
// the same two tables as two fp16 planes of E * 2^sexp(max |E|) (scaled split-fp16 backward): ONE workgroup measures the maximum,
// publishes it for the main kernel (e_amax) and splits; 16 K elements
__global__ __launch_bounds__(1024) void attn_split_tables_f16_kernel(const float* __restrict__ E, unsigned short* __restrict__ Es,
                                                                     unsigned short* __restrict__ Ets, float* __restrict__ e_amax,
                                                                     int R, int ET) {
  // gridDim.x workgroups: EVERY one measures max |E| over the whole (64 KB, L2-resident) table -- 16-B loads, four per thread --
  // and then splits its own share of the elements.  (One workgroup doing both passes with 4-B accesses: 26 us, eight times per
  // step on the main stream.)
  __shared__ float red[16];
  const int tid = threadIdx.x;
  f16_clamp_mode_a();
  float m = 0.f;
  if ((((size_t)E) & 15) == 0) {           // (a parameter packed behind an odd-sized one in a flat optimizer buffer is only 4-B aligned)
    const float4* E4 = reinterpret_cast<const float4*>(E);
    for (int i4 = tid; i4 < R * 4; i4 += 1024) {
      const float4 v = E4[i4];
      m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
  } else {
    for (int idx = tid; idx < R * 16; idx += 1024) m = fmaxf(m, fabsf(E[idx]));
  }
  m = wave_max(m);
  if ((tid & 63) == 0) red[tid >> 6] = m;
  __syncthreads();
  m = red[0];
#pragma unroll
  for (int w = 1; w < 16; ++w) m = fmaxf(m, red[w]);
  if (tid == 0 && blockIdx.x == 0) *e_amax = m;
  const float sc = exp2ia(f16_sexp_a(m));
  for (int idx = blockIdx.x * 1024 + tid; idx < R * 16; idx += gridDim.x * 1024) {
    const int row = idx >> 4, d = idx & 15;
    const float x = E[idx] * sc;
    const _Float16 h = (_Float16)x, l = (_Float16)(x - (float)h);
    const unsigned short hb = __builtin_bit_cast(unsigned short, h), lb = __builtin_bit_cast(unsigned short, l);
    Es[((long)0 * R + row) * 16 + d] = hb;
    Es[((long)1 * R + row) * 16 + d] = lb;
    const long to = ((long)(row >> 4) * 16 + d) * 16 + (row & 15);      // [plane][tile][d][16 offsets]
    Ets[(long)0 * ET * 16 + to] = hb;
    Ets[(long)1 * ET * 16 + to] = lb;
  }
}
// fp32 transposed table Et[16][ld] of the v2 kernel
__global__ void attn_transpose_table_kernel(const float* __restrict__ E, float* __restrict__ Et, int R, int ld) {
  int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= R * 16) return;
  Et[(long)(idx & 15) * ld + (idx >> 4)] = E[idx];
}

struct AttnBwd3Args {
  AttnGeom g;
  const float* QKV; const float* dO; const float* LSE; const float* Dl;
  float* dQKV;
  const __bf16* Es; const __bf16* Ets;
  float* dEs;                        // per-wave dE tiles: [wave item][nqt + KT][16 offsets][16 d]
  int R, ET, maxpos;
  float scale;
  int dbg;                           // timing ablations: SE_ATTN_DBG bits 1 no dE flush, 4 V := K, 8 E := Q (0 in production)
  // scaled split-fp16 form (se_attn_bwd_f16): device scalars >= max |QKV|, max |dO|, max |E| (the last one written by the table
  // kernel of the same call); Es / Ets then hold TWO fp16 planes of E * 2^sexp(*e_amax)
  const float* qkv_amax; const float* do_amax; const float* e_amax;
  float* dqkv_amax;                  // optional (F16): raised to max |dQKV| (operand scale of the consumers of the gradient)
  const float* O;                    // the forward output (attn_bwd4: delta computed in the kernel)
};

// key tiles of wave w when the 4 waves of a workgroup share one (sequence, head)
static __device__ __host__ __forceinline__ void group_split(int nkt, int w, int& kt0, int& cnt) {
  const int b = nkt >> 2, r = nkt & 3;
  cnt = b + (w < r ? 1 : 0);
  kt0 = w * b + (w < r ? w : r);
}

// key tiles of the register window a grouped wave with cnt tiles runs (the body is instantiated for KT and KT - 1 exactly)
static __device__ __host__ __forceinline__ int group_window(int KT, int cnt) { return (cnt == KT || cnt == KT - 1) ? cnt : KT; }
template <int KT, bool F16 = false>
struct Lds3 {
  static constexpr int NU = KT + 1;
  static constexpr int NPLA = F16 ? 2 : 3;                    // planes of a split operand: (hi, mid, lo) bf16 / (hi, lo) fp16
  // strip row stride (floats): + 4 makes the skewed cell accesses (row stride SW + 1) conflict free -- the three-plane images of
  // the bf16 form at KT = 7 left no room for it within 80 KB per workgroup (two per CU); the two-plane fp16 form has it
  static constexpr int SW = NU * 16 + ((KT == 7 && !F16) ? 0 : 4);
  static constexpr int KPB = KT * 16 * 32;                    // bytes of one plane of the K image
  static constexpr int KIMG = NPLA * KPB;                     // bytes: [planes][KT*16 keys][16 d] 16-bit
  static constexpr int DIMG = NPLA * 16 * 32;                 // bytes: [planes][16][16] 16-bit (one tile, split)
  static constexpr int STRIP = 16 * SW * 4;
  static constexpr int WAVE = KIMG + DIMG + STRIP;
};

// raw query-side operands of one query tile (prefetched one tile ahead)
struct QSide { float4 q4, do4; float qcf[4], docf[4], lse[4], dl[4]; };

// One wave: key tiles kt0 .. kt0 + nk_w - 1 of one (sequence, head), all query tiles.
//   NK    : key tiles the register arrays and the unrolled loops are built for;
//   EXACT : nk_w == NK, the unrolled key loop has no branches (one basic block per query tile: the compiler overlaps the LDS /
//           global latencies of step s + 1 with the arithmetic of step s);  otherwise NK = KT and steps s >= nk_w are skipped;
//   GROUP : the 4 waves of the workgroup share the (sequence, head): dQ tiles are summed through LDS (one barrier per query
//           tile) and stored once -- deterministic, no atomics.
// F16: scaled split-fp16 operands (two planes, 8-instruction splits, three K = 16 MFMAs per product).  Scales: Q, K, V by 2^sq,
// dO by 2^sdo, E by 2^se (measured maxima), P by 2^13 (folded into the exponent argument), dS by 2^sds with sds from the bound
// |dS| <= P scale (|dP| + |D|) <= 32 scale max|dO| max|V| (both are 16-term dot products against V and O, |O| <= max |V|); every
// accumulator is brought back by ONE power of two where it is stored.  dQ collects K^T dS^T (2^(sq + sds)) and E^T W^T
// (2^(se + sds)) in two accumulators.
// RING (grouped fp16 form, nkt <= 21): the dE tiles of the four waves of a workgroup are summed in LDS before they leave the CU.
// Wave w flushes offset tile D at query tile D + kt0_w + cnt_w (strictly increasing in w), so the contributions to one tile arrive
// in wave order, 5 - 6 query tiles apart, always behind the per-query-tile barrier of the dQ reduction: the first contributor
// STORES its tile into a 16-slot ring (slot = D mod 16: at most 16 tiles are open at any time), the next ones add to it, the last
// adds and writes the finished tile to the (sequence, head) item's table [2 nkt][256] -- 41 tile writes per item instead of 108,
// and the reduction kernel reads as much less.  The tiles still in the register windows after the last query tile go out in four
// rounds (wave 0 .. 3) with a barrier in between.  dQ's LDS exchange drops to one parity (+ one barrier) to make room for the ring.
template <int KT, int NK, bool EXACT, bool GROUP, bool F16 = false, bool RING = false>
static __device__ __forceinline__ void attn_bwd3_body(const AttnBwd3Args& a, unsigned char* smem3, const int wave, const int lane,
                                                      const long item, const int kt0, const int nk_w) {
  using L3 = Lds3<KT, F16>;
  constexpr int NU = NK + 1, SW = L3::SW;
  const int c = lane & 15, g = lane >> 4;
  const int n = a.g.n, nkt = (n + 15) >> 4, nqt = nkt;
  unsigned char* wl = smem3 + (size_t)wave * L3::WAVE;
  unsigned char* Kimg = wl;
  unsigned char* Dimg = wl + L3::KIMG;
  float* strip = reinterpret_cast<float*>(wl + L3::KIMG + L3::DIMG);
  float* dqs = reinterpret_cast<float*>(smem3 + (size_t)4 * L3::WAVE);      // GROUP: [2 parities (RING: 1)][4 waves][256]
  float* ring = dqs + 4 * 256;                                              // RING: [16 slots][256]
  const int head = (int)(item & 3), seq = (int)(item >> 2);
  const long base = seq_base(a.g, seq);
  const int ps = (int)a.g.pos_stride;
  const float* qb = a.QKV + base * 192 + head * 16;             // + pos * ps * 192 (+64: K, +128: V)
  const float* dob = a.dO + base * 64 + head * 16;
  const float* lseb = a.LSE + base * 4 + head;
  const float* dlb = a.Dl + base * 4 + head;
  float* dqb = a.dQKV + base * 192 + head * 16;
  const float l2e = 1.4426950408889634f;
  const long witem = GROUP ? item * 4 + wave : item;
  float* dEs = RING ? a.dEs + item * (long)(2 * nkt) * 256 : a.dEs + witem * (long)(nqt + KT) * 256;
  // RING: the offset tiles wave w flushes are [-(kt0_w + cnt_w), nqt - kt0_w - 1]
  int rlo[4] = {0, 0, 0, 0}, rhi[4] = {0, 0, 0, 0};
  if (RING) {
#pragma unroll
    for (int w = 0; w < 4; ++w) {      // (a wave whose tile count is neither KT nor KT - 1 runs the KT-wide window: group_window)
      int k0, cn;
      group_split(nkt, w, k0, cn);
      rlo[w] = -(k0 + group_window(KT, cn));
      rhi[w] = nqt - k0 - 1;
    }
  }
  const int trrow = c >> 2, trcol = c & 3;                      // transposed-read address roles of this lane
  const unsigned char* Esb = reinterpret_cast<const unsigned char*>(a.Es);      // planes (long)R * 32 bytes apart
  const unsigned char* Etb = reinterpret_cast<const unsigned char*>(a.Ets);     // planes (long)ET * 32 bytes apart
  const long esp = (long)a.R * 32, etp = (long)a.ET * 32;
  float sqf = 1.f, sdof = 1.f, kU = 1.f, kD = 1.f, kdl = a.scale, cq1 = 1.f, cq2 = 1.f, cdv = 1.f, lse13 = 0.f;
  float sc2 = a.scale * 1.4426950408889634f;
  if (F16) {
    f16_clamp_mode_a();
    const float aq = *a.qkv_amax, ado = *a.do_amax;
    const int sq = f16_sexp_a(aq), sdo = f16_sexp_a(ado), se = f16_sexp_a(*a.e_amax);
    const int sds = f16_sexp_a(32.f * a.scale * ado * aq);
    sqf = exp2ia(sq); sdof = exp2ia(sdo);
    kU = exp2ia(sq - se);                          // strip cells U = q.E at the scale of S = q.k
    sc2 *= exp2ia(-2 * sq);
    kD = a.scale * exp2ia(sds - 13 - sdo - sq);    // (dP accumulator) -> scale (dP) 2^(sds - 13): times P 2^13 = dS 2^sds
    kdl = a.scale * exp2ia(sds - 13);
    cq1 = exp2ia(-sq - sds); cq2 = exp2ia(-se - sds); cdv = exp2ia(-sdo - 13);
    lse13 = 13.f;
  }

  // ---- stage this wave's keys: K pre-split into the LDS image (row fragments b64, column fragments tr_b16) ----
  // all NK rows are requested before the first one is split (clamped key index, zeroed by a select): with the load behind
  // `if (key < n)` every key tile was its own dependent HBM round trip (rows of a time-axis sequence are 77 KB apart)
  float4 k4s[NK];
#pragma unroll
  for (int s = 0; s < NK; ++s) {
    int key = (kt0 + s) * 16 + c;
    if (key > n - 1) key = n - 1;
    k4s[s] = *reinterpret_cast<const float4*>(qb + (unsigned)(key * ps * 192 + 64 + 4 * g));
  }
#pragma unroll
  for (int s = 0; s < NK; ++s) {
    const int key = (kt0 + s) * 16 + c;
    const bool kok = (EXACT || s < nk_w) && key < n;
    const float4 k4 = make_float4(kok ? k4s[s].x : 0.f, kok ? k4s[s].y : 0.f, kok ? k4s[s].z : 0.f, kok ? k4s[s].w : 0.f);
    st_planes<F16>(Kimg + ((s * 16 + c) * 16 + 4 * g) * 2, L3::KPB, splitx<F16>(k4, sqf));
  }
  f32x4 dk[NK], dv[NK], de[NU];
  float omax = 0.f;                                                  // F16: max |dQKV| this wave stores
#pragma unroll
  for (int s = 0; s < NK; ++s) { dk[s] = (f32x4){0.f, 0.f, 0.f, 0.f}; dv[s] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
  for (int u = 0; u < NU; ++u) de[u] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // one finished offset tile of the dE window goes to this wave's slot (Dtile + kt0 + KT) of the scratch table:
  // v[r] = dE[delta = 16 Dtile + c][d = 4g + r]; tiles no (query, key) pair can reach are neither stored nor reduced
  auto flush = [&](const f32x4& v, int Dtile) {
    if (Dtile < -nkt || Dtile > nkt || (a.dbg & 1)) return;
    float4 val = make_float4(v[0] * cq1, v[1] * cq1, v[2] * cq1, v[3] * cq1);
    if (RING) {
      int first = 3, last = 0;                                    // (wave-uniform; this wave is one of the contributors)
#pragma unroll
      for (int w = 3; w >= 0; --w) if (Dtile >= rlo[w] && Dtile <= rhi[w]) first = w;
#pragma unroll
      for (int w = 0; w < 4; ++w) if (Dtile >= rlo[w] && Dtile <= rhi[w]) last = w;
      float4* rs = reinterpret_cast<float4*>(ring + ((Dtile + 64) & 15) * 256 + c * 16 + 4 * g);
      if (wave != first) { const float4 o = *rs; val.x += o.x; val.y += o.y; val.z += o.z; val.w += o.w; }
      if (wave == last) *reinterpret_cast<float4*>(dEs + (Dtile + nkt) * 256 + c * 16 + 4 * g) = val;
      else *rs = val;
      return;
    }
    *reinterpret_cast<float4*>(dEs + (Dtile + kt0 + KT) * 256 + c * 16 + 4 * g) = val;
  };
  auto load_qside = [&](int qt, QSide& o) {
    const int q0 = qt * 16;
    int qc = q0 + c; if (qc > n - 1) qc = n - 1;
    o.q4 = *reinterpret_cast<const float4*>(qb + (unsigned)(qc * ps * 192 + 4 * g));        // Q[q = c][4g..]
    o.do4 = *reinterpret_cast<const float4*>(dob + (unsigned)(qc * ps * 64 + 4 * g));
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int qr = q0 + 4 * g + j;
      if (qr > n - 1) qr = n - 1;
      o.qcf[j] = qb[(unsigned)(qr * ps * 192 + c)];          // Q[q = 4g + j][d = c]   (column fragment: contraction over queries)
      o.docf[j] = dob[(unsigned)(qr * ps * 64 + c)];
      o.lse[j] = lseb[(unsigned)(qr * ps * 4)];
      o.dl[j] = dlb[(unsigned)(qr * ps * 4)];
    }
  };
  QSide nxt;
  load_qside(0, nxt);

  for (int qt = 0; qt < nqt; ++qt) {
    // ---- query-side operands of this tile (prefetched during the previous tile; split once, used by all key tiles) ----
    const int q0 = qt * 16;
    const QSide cur = nxt;
    if (qt + 1 < nqt) load_qside(qt + 1, nxt);
    float nlse2[4], dl4s[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      // C-layout rows are queries 4g + r: p = exp2((s + u) * scale * log2e - lse * log2e); rows beyond n get -inf -> p = 0
      nlse2[j] = (q0 + 4 * g + j < n) ? lse13 - cur.lse[j] * l2e : -__builtin_inff();      // (F16: p carries P's factor 2^13)
      dl4s[j] = cur.dl[j] * kdl;
    }
    const S3 qrow = splitx<F16>(cur.q4, sqf);
    // dP pre-scaled by `scale` (F16: the factor is part of kD instead)
    const float dps = F16 ? sdof : a.scale;
    const S3 dorow = F16 ? splitx<true>(cur.do4, dps) : split3(cur.do4.x * dps, cur.do4.y * dps, cur.do4.z * dps, cur.do4.w * dps);
    const S3 qcol = splitx<F16>(cur.qcf[0], cur.qcf[1], cur.qcf[2], cur.qcf[3], sqf);
    const S3 docol = splitx<F16>(cur.docf[0], cur.docf[1], cur.docf[2], cur.docf[3], sdof);

    // ---- offset strip: U[q][delta] for the NU tiles Dtile = qt - kt0 - u, strip columns 16 (NK - u) + (delta & 15) ----
    // the E row fragments are requested in batches of EB tiles before their products: one L2 round trip per batch, not per tile
    constexpr int EB = 4;
#pragma unroll
    for (int u0 = 0; u0 < NU; u0 += EB) {
      S3 es[EB];
#pragma unroll
      for (int e = 0; e < EB; ++e) {
        const int u = u0 + e;
        if (u < NU) {
          const int row = 16 * (qt - kt0 - u) + c + a.maxpos;      // in range by the launch conditions
          const unsigned eo = (unsigned)(row * 16 + 4 * g);
          es[e] = ld_planes<F16>(Esb + 2 * eo, esp);
        }
      }
#pragma unroll
      for (int e = 0; e < EB; ++e) {
        const int u = u0 + e;
        if (u < NU) {
          const f32x4 uu = prodx<F16>(qrow, es[e], (f32x4){0.f, 0.f, 0.f, 0.f});   // C[q = 4g + r][delta_local = c]
#pragma unroll
          for (int r = 0; r < 4; ++r) strip[(4 * g + r) * SW + 16 * (NK - u) + c] = F16 ? uu[r] * kU : uu[r];
        }
      }
    }
    f32x4 dq = {0.f, 0.f, 0.f, 0.f};                                 // dQ^T[d = 4g + r][q = c]
    f32x4 dq2 = {0.f, 0.f, 0.f, 0.f};                                // F16: the E^T W^T part (its own scale)
    auto load_v = [&](int s_) {
      int kj = (kt0 + s_) * 16 + c; if (kj > n - 1) kj = n - 1;
      return *reinterpret_cast<const float4*>(qb + (unsigned)(kj * ps * 192 + 128 + 4 * g));
    };
    float4 vnext = load_v(0);

    // ---- the wave's key tiles ----
#pragma unroll
    for (int s = 0; s < NK; ++s) {
      if (EXACT || s < nk_w) {                                       // wave-uniform
        const int j0 = (kt0 + s) * 16;
        const bool kv = j0 + c < n;
        const S3 krow = ld_planes<F16>(Kimg + ((s * 16 + c) * 16 + 4 * g) * 2, L3::KPB);
        const float4 vcur = vnext;
        if (s + 1 < NK) vnext = load_v(s + 1);                       // one step ahead
        // (the run-time ablation switches double as scheduling fences: without these branch points the scheduler hoists
        // whole steps' loads, spills, and the kernel runs 8 % slower)
        const S3 vrow = (a.dbg & 4) ? krow : splitx<F16>(vcur, sqf);
        f32x4 s4 = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
        prodx2<F16>(qrow, krow, s4, dorow, vrow, dp);                // S[q = 4g + r][key = c], scale * dP[q][key]
        f32x4 pp, ds;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float* cell = &strip[(4 * g + r) * SW + 16 * (NK - s) + (4 * g + r) - c];    // the cell of offset q - key
          float p = __builtin_amdgcn_exp2f(fmaf(s4[r] + *cell, sc2, nlse2[r]));
          p = kv ? p : 0.f;
          pp[r] = p;
          ds[r] = F16 ? p * fmaf(dp[r], kD, -dl4s[r]) : p * (dp[r] - dl4s[r]);
          *cell = ds[r];                                             // W = skew(dS) replaces U in place
        }
        // contraction over the query rows 4g + r: the accumulator registers ARE the B operands
        const S3 pps = splitn<F16>(pp), dss = splitn<F16>(ds);
        // dV^T[d][key] += dO^T[d][q] P[q][key];  dK^T[d][key] += Q^T[d][q] dS[q][key]
        prodx2<F16>(docol, pps, dv[s], qcol, dss, dk[s]);
        // dQ^T[d][q] += K^T[d][key] dS^T[key][q]: both operands through hardware-transposed reads -- A from the K row
        // image, B from the split dS tile stored [key][q] (each lane writes 4 consecutive queries of its key: 8 bytes)
        st_planes<F16>(Dimg + c * 32 + g * 8, 512, dss);
        const S3 kcol = tr_planes<F16>(Kimg + ((s * 16 + 4 * g + trrow) * 16 + 4 * trcol) * 2, L3::KPB);
        const S3 dst = tr_planes<F16>(Dimg + (4 * g + trrow) * 32 + trcol * 8, 512);
        dq = prodx<F16>(kcol, dst, dq);
      }
    }

    // ---- consume the strip: dQ^T += E^T W^T, dE^T += Q^T W, one offset tile at a time ----
    const int lim = 16 * nk_w;
#pragma unroll
    for (int u0 = 0; u0 < NU; u0 += EB) {
      S3 ecs[EB];                                                    // E[Dt + 4g + j][d = c], a batch of tiles requested up front
#pragma unroll
      for (int e = 0; e < EB; ++e) {
        const int u = u0 + e;
        if (u < NU) {
          const unsigned eo = (unsigned)(((qt - kt0 - u) * 16 + a.maxpos + c) * 16 + 4 * g);   // tile (qt-kt0-u) + maxpos/16, row d = c
          ecs[e] = ld_planes<F16>(Etb + 2 * eo, etp);
        }
      }
#pragma unroll
      for (int e = 0; e < EB; ++e) {
        const int u = u0 + e;
        if (u < NU) {
          // a cell (q_local a, delta_local dl) of tile u belongs to key_rel = 16 u + a - dl (relative to the wave's first
          // key); cells whose key the wave does not own still hold U values: mask them
          float4 w4 = *reinterpret_cast<const float4*>(&strip[c * SW + 16 * (NK - u) + 4 * g]);     // W[a = c][dl = 4g + j]
          if (!EXACT || u == 0 || u == NU - 1) {
            const int kr = 16 * u + c - 4 * g;
            if ((unsigned)kr >= (unsigned)lim) w4.x = 0.f;
            if ((unsigned)(kr - 1) >= (unsigned)lim) w4.y = 0.f;
            if ((unsigned)(kr - 2) >= (unsigned)lim) w4.z = 0.f;
            if ((unsigned)(kr - 3) >= (unsigned)lim) w4.w = 0.f;
          }
          const S3 ws = splitn<F16>(w4);                           // (W = skew(dS): already at dS's scale)
          st_planes<F16>(Dimg + c * 32 + g * 8, 512, ws);                // image [a][dl]
          const S3 wt = tr_planes<F16>(Dimg + (4 * g + trrow) * 32 + trcol * 8, 512);      // W[a = 4g + j][dl = c]
          // dQ^T[d][q] += E^T[d][dl] W^T[dl][q];  dE^T[d][dl] += Q^T[d][q] W[q][dl]
          prodx2<F16>(ecs[e], ws, F16 ? dq2 : dq, qcol, wt, de[u]);
          if (a.dbg & 16) __builtin_amdgcn_s_sleep(1);               // (branch point: see the note at the V operand)
        }
      }
    }

    // The next tile's query-side operands were requested a whole tile ago: make them arrive HERE, before this tile's dQ / dE
    // stores are issued.  Left to the first use at the top of the next tile, the wait comes right after those stores, and behind
    // their (wave-uniform, but still control-flow) conditions the wait-count pass can only emit vmcnt(0): one exposed write
    // acknowledgement per query tile.
    asm volatile("" : "+v"(nxt.q4.x), "+v"(nxt.q4.y), "+v"(nxt.q4.z), "+v"(nxt.q4.w), "+v"(nxt.do4.x), "+v"(nxt.do4.y), "+v"(nxt.do4.z), "+v"(nxt.do4.w));
    asm volatile("" : "+v"(nxt.qcf[0]), "+v"(nxt.qcf[1]), "+v"(nxt.qcf[2]), "+v"(nxt.qcf[3]), "+v"(nxt.docf[0]), "+v"(nxt.docf[1]), "+v"(nxt.docf[2]), "+v"(nxt.docf[3]));
    asm volatile("" : "+v"(nxt.lse[0]), "+v"(nxt.lse[1]), "+v"(nxt.lse[2]), "+v"(nxt.lse[3]), "+v"(nxt.dl[0]), "+v"(nxt.dl[1]), "+v"(nxt.dl[2]), "+v"(nxt.dl[3]));
    // ---- dQ of this query tile ----
    if (F16) {
#pragma unroll
      for (int r = 0; r < 4; ++r) dq[r] = fmaf(dq[r], cq1, dq2[r] * cq2);
    }
    if (GROUP) {
      float* slot = dqs + (RING ? 0 : (qt & 1) * 4) * 256;
      *reinterpret_cast<float4*>(slot + wave * 256 + c * 16 + 4 * g) = make_float4(dq[0], dq[1], dq[2], dq[3]);
      __syncthreads();
      if (wave == (qt & 3) && q0 + c < n) {
        float4 t0 = *reinterpret_cast<const float4*>(slot + 0 * 256 + c * 16 + 4 * g);
        const float4 t1 = *reinterpret_cast<const float4*>(slot + 1 * 256 + c * 16 + 4 * g);
        const float4 t2 = *reinterpret_cast<const float4*>(slot + 2 * 256 + c * 16 + 4 * g);
        const float4 t3 = *reinterpret_cast<const float4*>(slot + 3 * 256 + c * 16 + 4 * g);
        t0.x += t1.x + (t2.x + t3.x); t0.y += t1.y + (t2.y + t3.y); t0.z += t1.z + (t2.z + t3.z); t0.w += t1.w + (t2.w + t3.w);
        *reinterpret_cast<float4*>(dqb + (unsigned)((q0 + c) * ps * 192 + 4 * g)) = t0;
        if (F16) omax = fmaxf(fmaxf(omax, fmaxf(fabsf(t0.x), fabsf(t0.y))), fmaxf(fabsf(t0.z), fabsf(t0.w)));
      }
      if (RING) __syncthreads();            // one parity: the slots are rewritten in the next query tile
    } else if (q0 + c < n) {
      *reinterpret_cast<float4*>(dqb + (unsigned)((q0 + c) * ps * 192 + 4 * g)) = make_float4(dq[0], dq[1], dq[2], dq[3]);
      if (F16) omax = fmaxf(fmaxf(omax, fmaxf(fabsf(dq[0]), fabsf(dq[1]))), fmaxf(fabsf(dq[2]), fabsf(dq[3])));
    }
    // ---- slide the dE window: tile qt - kt0 - NK is complete ----
    flush(de[NU - 1], qt - kt0 - NK);
#pragma unroll
    for (int u = NU - 1; u > 0; --u) de[u] = de[u - 1];
    de[0] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  // after the last rotation slot u holds tile nqt - kt0 - u
  if (RING) {                                // the remaining window tiles leave in wave order (contribution order), a barrier apart
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      if (wave == w) {
#pragma unroll
        for (int u = 1; u < NU; ++u) flush(de[u], nqt - kt0 - u);
      }
      __syncthreads();
    }
  } else {
#pragma unroll
    for (int u = 1; u < NU; ++u) flush(de[u], nqt - kt0 - u);
  }
  // ---- dK, dV of the wave's keys: C layout [d = 4g + r][key = c] ----
#pragma unroll
  for (int s = 0; s < NK; ++s) {
    const int key = (kt0 + s) * 16 + c;
    if ((EXACT || s < nk_w) && key < n) {
      float* p = dqb + (unsigned)(key * ps * 192 + 4 * g);
      *reinterpret_cast<float4*>(p + 64) = make_float4(dk[s][0] * cq1, dk[s][1] * cq1, dk[s][2] * cq1, dk[s][3] * cq1);
      *reinterpret_cast<float4*>(p + 128) = make_float4(dv[s][0] * cdv, dv[s][1] * cdv, dv[s][2] * cdv, dv[s][3] * cdv);
      if (F16) {
#pragma unroll
        for (int r = 0; r < 4; ++r) omax = fmaxf(omax, fmaxf(fabsf(dk[s][r] * cq1), fabsf(dv[s][r] * cdv)));
      }
    }
  }
  if (F16 && a.dqkv_amax) {
    omax = wave_max(omax);
    if (lane == 0) amax_raise_(a.dqkv_amax, omax);
  }
}

template <int KT, bool GROUP, bool F16 = false, bool RING = false>
__global__ __launch_bounds__(256, 2) void attn_bwd3_kernel(AttnBwd3Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem3[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nkt = (a.g.n + 15) >> 4;
  if (GROUP) {
    int kt0, cnt;
    group_split(nkt, wave, kt0, cnt);
    // every wave runs nqt barriers whichever branch it takes (s_barrier counts arrivals, not program counters)
    const long gitem = xcd_item((int)blockIdx.x, (int)gridDim.x);
    if (cnt == KT) attn_bwd3_body<KT, KT, true, true, F16, RING>(a, smem3, wave, lane, gitem, kt0, cnt);
    else if (cnt == KT - 1) attn_bwd3_body<KT, KT - 1, true, true, F16, RING>(a, smem3, wave, lane, gitem, kt0, cnt);
    else attn_bwd3_body<KT, KT, false, true, F16, RING>(a, smem3, wave, lane, gitem, kt0, cnt);
  } else {
    const long item = (long)blockIdx.x * 4 + wave;
    if (item >= (long)a.g.nseq * 4) return;                     // whole wave leaves: EXEC stays full for the others
    if (nkt == KT) attn_bwd3_body<KT, KT, true, false, F16>(a, smem3, wave, lane, item, 0, nkt);
    else attn_bwd3_body<KT, KT, false, false, F16>(a, smem3, wave, lane, item, 0, nkt);
  }
}

// dE[delta][d] += sum over the wave items of their finished tiles.  grid = (offset tile, chunk of 256 wave items); a
// workgroup streams 256 x 1 KB with 16-byte loads (4 items in flight per 64-lane group), sums through LDS, 1 KB of atomics.
template <int KT, bool GROUP>
__global__ __launch_bounds__(256) void attn_de_reduce3_kernel(const float* __restrict__ dEs, float* __restrict__ dE, long nwitems,
                                                              int nkt, int maxpos, int R) {
  __shared__ float4 part[4][64];
  const int Dtile = (int)blockIdx.x - nkt;                       // -nkt .. nkt
  const long w0 = (long)blockIdx.y * 256;
  const int nslot = nkt + KT, t = threadIdx.x & 63, sub = threadIdx.x >> 6;
  // a wave with key tiles [kt0, kt0 + cnt) stores the tiles -kt0 - cnt .. nqt - kt0 - 1 in slot (tile + kt0 + KT)
  int kt0s[4] = {0, 0, 0, 0}, cnts[4] = {nkt, nkt, nkt, nkt};
  if (GROUP) { for (int w = 0; w < 4; ++w) group_split(nkt, w, kt0s[w], cnts[w]); }
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8
  for (int i = sub; i < 256; i += 4) {                            // w0 and the stride are multiples of 4: wave role = sub
    const long w = w0 + i;
    const int slot = Dtile + kt0s[sub] + KT;
    if (w < nwitems && slot >= KT - cnts[sub] && slot < nslot) {
      const float4 v = *reinterpret_cast<const float4*>(dEs + (w * nslot + slot) * 256 + t * 4);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
  }
  part[sub][t] = s;
  __syncthreads();
  if (sub == 0) {
    const float4 b = part[1][t], c4 = part[2][t], d = part[3][t];
    s.x += b.x + (c4.x + d.x); s.y += b.y + (c4.y + d.y); s.z += b.z + (c4.z + d.z); s.w += b.w + (c4.w + d.w);
    const int row = 16 * Dtile + (t >> 2) + maxpos;              // element index in the tile = 4 t .. 4 t + 3 = [delta_l][d]
    if (row >= 0 && row < R) {
      float* p = &dE[(long)row * 16 + (t & 3) * 4];
      atomicAdd(p, s.x); atomicAdd(p + 1, s.y); atomicAdd(p + 2, s.z); atomicAdd(p + 3, s.w);
    }
  }
}

// the same reduction over the per-ITEM tables of the ring form: dEs [items][2 nkt][256], tile slot = D + nkt, D = -nkt .. nkt - 1
__global__ __launch_bounds__(256) void attn_de_reduce_items_kernel(const float* __restrict__ dEs, float* __restrict__ dE, long nitems,
                                                                   int nkt, int maxpos, int R) {
  __shared__ float4 part[4][64];
  const int Dtile = (int)blockIdx.x - nkt;                       // -nkt .. nkt - 1
  const long w0 = (long)blockIdx.y * 256;
  const int nslot = 2 * nkt, t = threadIdx.x & 63, sub = threadIdx.x >> 6;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);                    // (the table is shared by the heads: every item adds to it)
#pragma unroll 8
  for (int i = sub; i < 256; i += 4) {
    const long it = w0 + i;
    const long itc = it < nitems ? it : nitems - 1;              // (unconditional load, zeroed by the select)
    const float4 v = *reinterpret_cast<const float4*>(dEs + (itc * nslot + blockIdx.x) * 256 + t * 4);
    if (it < nitems) { s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
  }
  part[sub][t] = s;
  __syncthreads();
  if (sub == 0) {
    const float4 b = part[1][t], c4 = part[2][t], d = part[3][t];
    s.x += b.x + (c4.x + d.x); s.y += b.y + (c4.y + d.y); s.z += b.z + (c4.z + d.z); s.w += b.w + (c4.w + d.w);
    const int row = 16 * Dtile + (t >> 2) + maxpos;
    if (row >= 0 && row < R) {
      float* p = &dE[(long)row * 16 + (t & 3) * 4];
      atomicAdd(p, s.x); atomicAdd(p + 1, s.y); atomicAdd(p + 2, s.z); atomicAdd(p + 3, s.w);
    }
  }
}

#include "se_attn_bwd4.h"

// =====================================================================================================================
// v3 forward: the fwd2 structure (one workgroup per (sequence, head), 32- or 16-query blocks dealt to the waves, online
// softmax, sliding rel-pos window skewed through wave-private LDS) on the packed split-bf16 products of the v3 backward:
//   S^T  = K  . Q^T : K row fragments read from global memory one key tile ahead and split on the fly (shared by the TQ query
//                     tiles of the block); Q^T split once per block
//   U    = Ew . Q^T : E row fragments likewise (one new offset tile per step serves both query tiles, rotated in registers)
//   O^T += V^T . P^T: V is staged ONCE per workgroup, pre-split, in LDS and its column fragments (contraction over keys) come
//                     through ds_read_b64_tr_b16; P^T's accumulator registers are the B operand after one split
// 48 instead of 128 matrix-pipe cycles per 16x16x16 product; only V lives in LDS (32 KB at n = 321): two 8-wave workgroups per CU.
// =====================================================================================================================
// F16: the scaled split-fp16 form (two planes, three K = 16 MFMAs per product, 8-instruction splits): Q, K, V are scaled by
// 2^sq (sq from *qkv_amax), the E table arrives as two fp16 planes scaled by 2^se (se from *e_amax), P by 2^13 (folded into the
// exponent argument); S and U are brought to the logit scale by the two factors of one multiply + one FMA per element, the
// output by one multiply per element at the end.
template <int V> struct IC_ { static constexpr int value = V; };
template <int TQ, bool F16 = false>
// packed fp32 VALU ops allowed in THIS kernel (se_common.h): it is VALU-bound (83 % VALU-busy, 36 % MFMA-busy); halving the issue slots of
// part of its softmax chain pays for the matrix-pipe stalls: 0.681 -> 0.651 ms (n = 321), 0.270 -> 0.263 (n = 101); SE_ATTN_FWD_NO_PK builds: off
#ifdef SE_ATTN_FWD_NO_PK
#define ATTN_FWD_PK_ATTR
#else
#define ATTN_FWD_PK_ATTR SE_PACKED_FP32_KERNEL
#endif
__global__ __launch_bounds__(512) ATTN_FWD_PK_ATTR void attn_fwd3_kernel(AttnArgs a, int NP, int qsplit) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_f3[];
  static_assert(TQ == 1 || TQ == 2, "the offset-fragment ring of key_step is indexed by the step parity");
  constexpr int NPLA = F16 ? 2 : 3;
  unsigned char* Vimg = smem_f3;                               // [NPLA planes][NP keys][16 d] 16-bit
  float* Ubase = reinterpret_cast<float*>(smem_f3 + (size_t)NPLA * NP * 32);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int NT = blockDim.x, NW = NT >> 6;
  const int c = lane & 15, g = lane >> 4;
  const int n = a.g.n;
  // qsplit > 1 (few long sequences: a 10 s utterance is 101 x 4 items of 1601 positions): the query blocks of an item are dealt to
  // qsplit workgroups, each with its own V image
  const int part = qsplit > 1 ? (int)blockIdx.x % qsplit : 0;
  const int item_ = qsplit > 1 ? (int)blockIdx.x / qsplit : xcd_item((int)blockIdx.x, (int)gridDim.x);
  const int head = item_ & 3, seq = item_ >> 2;
  const long base = seq_base(a.g, seq);
  const int ps = (int)a.g.pos_stride;
  const float* qb = a.QKV + base * 192 + head * 16;
  const int vpb = NP * 32;                                     // bytes between the planes of the V image
  float sqf = 1.f, cS = 1.4426950408889634f * a.scale, cU = cS, osc = 1.f;
  if (F16) {
    f16_clamp_mode_a();
    const int sq = f16_sexp_a(*a.qkv_amax), se = f16_sexp_a(*a.e_amax);
    sqf = exp2ia(sq);
    cU = cS * exp2ia(-sq - se);
    cS = cS * exp2ia(-2 * sq);
    osc = exp2ia(-sq);
  }
  // V rows in batches of 4 per thread: requested together (clamped row, zeroed by a select), then split -- one load behind
  // `if (j < n)` per loop iteration was one dependent HBM round trip per iteration
  for (int i0 = tid; i0 < NP * 4; i0 += 4 * NT) {
    float4 vv[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      int j = (i0 + e * NT) >> 2;
      if (j > n - 1) j = n - 1;
      vv[e] = *reinterpret_cast<const float4*>(qb + (unsigned)(j * ps * 192 + 128 + 4 * (tid & 3)));
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int i = i0 + e * NT, j = i >> 2, q = i & 3;          // NT is a multiple of 4: q == tid & 3
      if (i < NP * 4) {
        const bool ok = j < n;
        const S3 vs = splitx<F16>(make_float4(ok ? vv[e].x : 0.f, ok ? vv[e].y : 0.f, ok ? vv[e].z : 0.f, ok ? vv[e].w : 0.f), sqf);
        st_planes<F16>(Vimg + (j * 16 + 4 * q) * 2, vpb, vs);
      }
    }
  }
  __syncthreads();
  float* Ul = Ubase + wave * (TQ * 512);      // [tile t][slot][256]
  const float l2e = 1.4426950408889634f * a.scale;
  const int qblocks = (n + 16 * TQ - 1) / (16 * TQ), nkt = (n + 15) / 16;
  const int trrow = c >> 2, trcol = c & 3;
  // lane-constant offsets (floats, without the tile's t * 512) of the four window cells, per parity of the hi slot: cell (key 4g + r,
  // query c) holds offset dl = c - (4g + r): in the hi strip for dl >= 0, else in the lo strip at 16 + dl
  int coff[2][4];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int dl = c - (4 * g + r);
      coff[h][r] = (dl >= 0 ? h * 256 : (h ^ 1) * 256 + 256) + dl * 16 + c;
    }
  const unsigned char* Esp = reinterpret_cast<const unsigned char*>(a.Es);
  auto e_row = [&](int D) -> S3 {             // E[clamp(D + c)][4g..4g+3], split: the A operand rows are offsets
    int d = D + c;
    d = d < -a.maxpos ? -a.maxpos : (d > a.maxpos ? a.maxpos : d);
    const unsigned eo = (unsigned)((d + a.maxpos) * 16 + 4 * g);
    if (F16 || Esp)                            // pre-split table: 8-byte loads per plane instead of a 16-byte load + a split
      return ld_planes<F16>(Esp + 2 * eo, 2 * a.es_plane);
    return split3(*reinterpret_cast<const float4*>(a.E + eo));
  };
  auto k_row = [&](int j0) {                  // K[j0 + c][4g..4g+3]
    int kj = j0 + c; if (kj > n - 1) kj = n - 1;
    return *reinterpret_cast<const float4*>(qb + (unsigned)(kj * ps * 192 + 64 + 4 * g));
  };
  for (int qbk = part * NW + wave; qbk < qblocks; qbk += NW * qsplit) {
    const int i0 = qbk * 16 * TQ;
    S3 qf[TQ];
#pragma unroll
    for (int t = 0; t < TQ; ++t) {
      int qi = i0 + 16 * t + c; if (qi > n - 1) qi = n - 1;
      qf[t] = splitx<F16>(*reinterpret_cast<const float4*>(qb + (unsigned)(qi * ps * 192 + 4 * g)), sqf);
    }
    f32x4 o[TQ];
    float m[TQ], l[TQ];
#pragma unroll
    for (int t = 0; t < TQ; ++t) { m[t] = -1e30f; l[t] = 0.f; o[t] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    auto u_tile = [&](const S3& es, int t, int slot) {
      const f32x4 u = prodx<F16>(es, qf[t], (f32x4){0.f, 0.f, 0.f, 0.f});
#pragma unroll
      for (int r = 0; r < 4; ++r) Ul[(t * 2 + slot) * 256 + (4 * g + r) * 16 + c] = u[r];
    };
    // offset tiles: query tile t at key tile j0 needs offsets [D, D + 15] (hi) and [D - 16, D - 1] (lo), D = i0 + 16 t - j0.
    // hi of step kt is lo of step kt - 1, and tile t's lo is tile t - 1's ... tile (t, D - 16) == tile (t - 1, D) shifted:
    // E fragments depend only on the offset base, so one new fragment per step serves all TQ tiles (rotated in registers)
    S3 ef[TQ + 1];                             // ef[t + 1] = E rows at base i0 + 16 t - j0 (hi of tile t), ef[0]: lo of tile 0
#pragma unroll
    for (int t = 0; t < TQ; ++t) { ef[t + 1] = e_row(i0 + 16 * t); u_tile(ef[t + 1], t, 0); }
    S3 enext = e_row(i0 - 16);
    float4 knext = k_row(0);
    // The strip slot that holds the hi offsets alternates with the key step: the loop is unrolled by two so that the slot is a
    // compile-time constant (cell addresses = lane-constant offset + immediate), and only the LAST key tile can hold keys >= n:
    // the validity selects live in its own instantiation of the step.
    auto key_step = [&](auto HC, auto MC, const int kt) {
      constexpr int hi = decltype(HC)::value, lo = hi ^ 1;
      constexpr bool MASK = decltype(MC)::value != 0;
      const int j0 = kt * 16;
      // the window slides by one fragment per step and a step reads the TQ newest ones: they live in a ring of TQ registers sets,
      // LOGICAL fragment t of step kt in slot (t - kt) mod TQ = (t - hi) mod TQ (TQ <= 2) -- nothing is moved (the shifting form
      // cost 4 TQ register moves per plane and step); ef[TQ] is only the hi fragment of the last tile at the start of a block
      constexpr int RR = hi % TQ;
      ef[(TQ - RR) % TQ] = enext;
      const S3 kf = splitx<F16>(knext, sqf);
      if (kt + 1 < nkt) { enext = e_row(i0 - j0 - 32); knext = k_row(j0 + 16); }      // one step ahead
#pragma unroll
      for (int t = 0; t < TQ; ++t) u_tile(ef[(t + TQ - RR) % TQ], t, lo);          // lo tile of query tile t: offsets base i0 + 16 (t - 1) - j0
      // V[j0 + 4g + j][d = c]: transposed read of the row image
      const S3 vcol = tr_planes<F16>(Vimg + ((j0 + 4 * g + trrow) * 16 + 4 * trcol) * 2, vpb);
#pragma unroll
      for (int t = 0; t < TQ; ++t) {
        const f32x4 s4 = prodx<F16>(kf, qf[t], (f32x4){0.f, 0.f, 0.f, 0.f});      // S^T[key 4g + r][query c]
        float sc[4], tmax = -1e30f, uu[4];
        // the four window cells first, through SELECTED ADDRESSES (a ternary over the two loads compiles to one exec-masked
        // branch per cell: eight branch regions per key step)
#pragma unroll
        for (int r = 0; r < 4; ++r) uu[r] = Ul[t * 512 + coff[hi][r]];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float lg = F16 ? fmaf(uu[r], cU, s4[r] * cS) : (s4[r] + uu[r]) * l2e;
          sc[r] = (!MASK || j0 + 4 * g + r < n) ? lg : -1e30f;
          tmax = fmaxf(tmax, sc[r]);
        }
        tmax = xor16_max_(tmax);                             // (VALU row swaps, not ds_bpermute: se_common.h)
        tmax = xor32_max_(tmax);
        // F16: the running reference m moves only when some query of the tile outgrows it by more than 2^LAZY (wave-uniform test):
        // after the first key tiles that is rare, and the rescale of o / l (an exp2, 5 multiplies) and its dependency on this
        // tile's maximum leave the chain.  p <= 2^LAZY then, so P's fp16 scale is 2^(13 - LAZY) (hi plane <= 2^13).
        constexpr float LAZY = 4.f, PSC = F16 ? 13.f - LAZY : 0.f;
        if (!F16 || a.eager || __builtin_amdgcn_ballot_w64(tmax > m[t] + LAZY) != 0) {
          const float mn = fmaxf(m[t], tmax);
          const float corr = __builtin_amdgcn_exp2f(m[t] - mn);
          m[t] = mn;
          l[t] *= corr;
#pragma unroll
          for (int r = 0; r < 4; ++r) o[t][r] *= corr;
        }
        const float mnp = m[t] - PSC;                               // F16: p and the running sum carry the factor 2^PSC of P's scale
        f32x4 p;
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) { p[r] = __builtin_amdgcn_exp2f(sc[r] - mnp); psum += p[r]; }
        l[t] += psum;
        o[t] = prodx<F16>(vcol, splitn<F16>(p), o[t]);         // O^T[d 4g + r][query c] += V^T[d][key] P^T[key][query]
      }
    };
    {
      const int nlast = nkt - 1;
      int kt = 0;
      for (; kt + 2 <= nlast; kt += 2) { key_step(IC_<0>{}, IC_<0>{}, kt); key_step(IC_<1>{}, IC_<0>{}, kt + 1); }
      if (kt < nlast) { key_step(IC_<0>{}, IC_<0>{}, kt); ++kt; }
      if (kt & 1) key_step(IC_<1>{}, IC_<1>{}, kt); else key_step(IC_<0>{}, IC_<1>{}, kt);
    }
#pragma unroll
    for (int t = 0; t < TQ; ++t) {
      float lt = l[t];
      lt = xor16_sum_(lt);
      lt = xor32_sum_(lt);
      const int qi = i0 + 16 * t + c;
      if (qi < n) {
        const long tok = base + (long)qi * ps;
        const float inv = osc / lt;                                 // (F16: the 2^13 of P cancels, V's 2^sq is taken out here)
        *reinterpret_cast<float4*>(a.O + tok * 64 + head * 16 + 4 * g) = make_float4(o[t][0] * inv, o[t][1] * inv, o[t][2] * inv, o[t][3] * inv);
        if (g == 0 && a.LSE) a.LSE[tok * 4 + head] = (m[t] + log2f(lt) - (F16 ? 9.f : 0.f)) * 0.6931471805599453f;      // (9 = 13 - LAZY)
      }
    }
  }
}

static int check_geom(const AttnGeom& g) {
  SE_REQUIRE(g.nseq > 0 && g.n > 0 && g.inner > 0, "attention: bad geometry");
  return 0;
}

static int attn_fwd_impl(const float* QKV, const float* E, const void* Es, long es_plane, float* O, float* LSE, int nseq, int n,
                         int inner, long outer_stride, long inner_stride, long pos_stride, int maxpos, float scale, void* stream,
                         const float* qkv_amax = nullptr, const float* e_amax = nullptr);
// shapes the split-fp16 forward takes (32-bit lane offsets, V image + strips within the LDS of a CU)
static bool attn_f16_fwd_ok(int n, long pos_stride) {
  const long NP = ((n + 15) / 16) * 16;
  return pos_stride * 192 * NP < 2147483647L && (size_t)2 * NP * 32 + (size_t)512 * 2 * 8 * sizeof(float) <= 160 * 1024;
}

extern "C" int se_attn_fwd(const float* QKV, const float* E, float* O, float* LSE, int nseq, int n, int inner,
                           long outer_stride, long inner_stride, long pos_stride, int maxpos, float scale,
                           void* stream) {
  return attn_fwd_impl(QKV, E, nullptr, 0, O, LSE, nseq, n, inner, outer_stride, inner_stride, pos_stride, maxpos, scale, stream);
}

extern "C" int se_attn_fwd_es(const float* QKV, const float* E, const void* Es, long es_plane, float* O, float* LSE, int nseq,
                              int n, int inner, long outer_stride, long inner_stride, long pos_stride, int maxpos, float scale,
                              void* stream) {
  SE_REQUIRE(Es == nullptr || (es_plane >= (long)(2 * maxpos + 1) * 16 && (es_plane % 4) == 0 && ((size_t)Es & 7) == 0),
             "attn_fwd_es: the pre-split table needs planes of >= (2 maxpos + 1) * 16 elements, 8-byte aligned");
  return attn_fwd_impl(QKV, E, Es, es_plane, O, LSE, nseq, n, inner, outer_stride, inner_stride, pos_stride, maxpos, scale, stream);
}
extern "C" int se_attn_fwd_f16(const float* QKV, const void* Es, long es_plane, const float* qkv_amax, const float* e_amax, float* O,
                               float* LSE, int nseq, int n, int inner, long outer_stride, long inner_stride, long pos_stride,
                               int maxpos, float scale, void* stream) {
  SE_REQUIRE(Es && qkv_amax && e_amax, "attn_fwd_f16: the fp16 planes of E and both maxima are required");
  SE_REQUIRE(es_plane >= (long)(2 * maxpos + 1) * 16 && (es_plane % 4) == 0 && ((size_t)Es & 7) == 0,
             "attn_fwd_f16: the pre-split table needs planes of >= (2 maxpos + 1) * 16 elements, 8-byte aligned");
  SE_REQUIRE(attn_f16_fwd_ok(n, pos_stride), "attn_fwd_f16: shape outside the split-fp16 kernel (n = %d): use se_attn_fwd_es", n);
  return attn_fwd_impl(QKV, reinterpret_cast<const float*>(Es), Es, es_plane, O, LSE, nseq, n, inner, outer_stride, inner_stride,
                       pos_stride, maxpos, scale, stream, qkv_amax, e_amax);
}

static int attn_fwd_impl(const float* QKV, const float* E, const void* Es, long es_plane, float* O, float* LSE, int nseq, int n,
                         int inner, long outer_stride, long inner_stride, long pos_stride, int maxpos, float scale, void* stream,
                         const float* qkv_amax, const float* e_amax) {
  static const int fwd_eager = getenv("SE_ATTN_FWD_LAZY") != nullptr && atoi(getenv("SE_ATTN_FWD_LAZY")) == 0;
  AttnArgs a{{nseq, n, inner, outer_stride, inner_stride, pos_stride}, QKV, E, O, LSE, maxpos, scale, Es, es_plane, qkv_amax, e_amax, fwd_eager};
  const bool f16 = qkv_amax != nullptr;
  if (int e = check_geom(a.g)) return e;
  SE_REQUIRE(QKV && E && O, "attn_fwd: null operand");
  const int NP = ((n + 15) / 16) * 16;
  // waves per workgroup: 32-query blocks are dealt round-robin to the waves; long sequences get 6 or 8 waves so that
  // 3-4 waves share a SIMD (the per-step chain S -> softmax -> PV is latency-bound with 2), short ones 4
  // waves per workgroup / 16-query tiles per wave block (measured at B = 16): short sequences (n <= 128) 4 waves x 1 tile
  // (n = 101: 0.37 ms vs 0.41 with 2 tiles), long ones 8 waves x 2 tiles (n = 321: 1.09 ms vs 1.39 with 4 waves -- the
  // per-step chain S -> softmax -> PV is latency-bound with 2 waves per SIMD, 8-wave workgroups put 4 there)
  const int qb32 = (n + 31) / 32;
  int nw = qb32 <= 4 ? 4 : 8, tq = qb32 <= 4 ? 1 : 2;
  if (const char* e = getenv("SE_ATTN_FWD_WAVES")) { int v = atoi(e); if (v == 4 || v == 6 || v == 8) nw = v; }
  if (const char* e = getenv("SE_ATTN_FWD_TQ")) { int v = atoi(e); if (v == 1 || v == 2) tq = v; }
  int use3 = pos_stride * 192 * (long)NP < 2147483647L;
  if (const char* e = getenv("SE_ATTN_FWD")) { if (atoi(e) == 2) use3 = 0; }
  if (use3) {
    // split-bf16 kernel; measured at B = 16 (ms): n = 321: 8 waves x 2 tiles 0.92, 7 x 2 0.95, 8 x 1 0.94, 4 x 2 0.97, 6 x 2 1.10
    // (fwd2: 1.07); n = 101: 4 x 1 0.342, 8 x 2 0.352, 4 x 2 0.367 (fwd2: 0.355)
    int nw3 = qb32 <= 4 ? 4 : 8, tq3 = qb32 <= 4 ? 1 : 2;
    if (const char* e = getenv("SE_ATTN_FWD_WAVES")) { int v = atoi(e); if (v >= 1 && v <= 8) nw3 = v; }
    if (const char* e = getenv("SE_ATTN_FWD_TQ")) { int v = atoi(e); if (v == 1 || v == 2) tq3 = v; }
    if (qb32 <= 4) { if (const char* e = getenv("SE_ATTN_FWD_WAVES_SMALL")) { int v = atoi(e); if (v >= 1 && v <= 8) nw3 = v; } }
    // fewer items than two rounds of workgroups (batch-1 inference): split the query blocks of an item over several workgroups
    int qsplit = 1;
    {
      const long items = (long)nseq * 4, qblk = (n + 16 * tq3 - 1) / (16 * tq3);
      while (items * qsplit < 1024 && (long)nw3 * qsplit * 2 <= qblk && qsplit < 8) ++qsplit;
      if (const char* e = getenv("SE_ATTN_FWD_QSPLIT")) { int v = atoi(e); if (v >= 1 && v <= 16) qsplit = v; }
    }
    const size_t sh3 = (size_t)(f16 ? 2 : 3) * NP * 32 + (size_t)512 * tq3 * nw3 * sizeof(float);
    if (sh3 <= 160 * 1024) {
      static size_t raised3[2][3] = {{0, 0, 0}, {0, 0, 0}};
      if (sh3 > 64 * 1024 && sh3 > raised3[f16][tq3]) {
        const void* fn = f16 ? (tq3 == 1 ? (const void*)attn_fwd3_kernel<1, true> : (const void*)attn_fwd3_kernel<2, true>)
                             : (tq3 == 1 ? (const void*)attn_fwd3_kernel<1> : (const void*)attn_fwd3_kernel<2>);
        SE_REQUIRE(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh3) == hipSuccess,
                   "attn_fwd: cannot raise the dynamic LDS limit to %zu", sh3);
        raised3[f16][tq3] = sh3;
      }
      const dim3 grid3((unsigned)(nseq * 4 * qsplit));
      if (f16 && tq3 == 1) hipLaunchKernelGGL((attn_fwd3_kernel<1, true>), grid3, dim3(64 * nw3), sh3, as_stream(stream), a, NP, qsplit);
      else if (f16) hipLaunchKernelGGL((attn_fwd3_kernel<2, true>), grid3, dim3(64 * nw3), sh3, as_stream(stream), a, NP, qsplit);
      else if (tq3 == 1) hipLaunchKernelGGL(attn_fwd3_kernel<1>, grid3, dim3(64 * nw3), sh3, as_stream(stream), a, NP, qsplit);
      else hipLaunchKernelGGL(attn_fwd3_kernel<2>, grid3, dim3(64 * nw3), sh3, as_stream(stream), a, NP, qsplit);
      return se_check_launch("se_attn_fwd");
    }
  }
  SE_REQUIRE(!f16, "attn_fwd_f16: the sequence does not fit the split-fp16 kernel (n = %d)", n);
  const size_t sh = ((size_t)NP * 16 + 16 * (size_t)(NP + 4) + 512 * (size_t)tq * nw) * sizeof(float);
  if (sh <= 160 * 1024) {       // K / V of one (sequence, head) fit in LDS: staged kernel
    static size_t raised[3] = {0, 0, 0};
    if (sh > 64 * 1024 && sh > raised[tq]) {
      const void* fn = tq == 1 ? (const void*)attn_fwd2_kernel<1> : (const void*)attn_fwd2_kernel<2>;
      SE_REQUIRE(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh) ==
                     hipSuccess, "attn_fwd: cannot raise the dynamic LDS limit to %zu", sh);
      raised[tq] = sh;
    }
    if (tq == 1) hipLaunchKernelGGL(attn_fwd2_kernel<1>, dim3(nseq * 4), dim3(64 * nw), sh, as_stream(stream), a, NP);
    else hipLaunchKernelGGL(attn_fwd2_kernel<2>, dim3(nseq * 4), dim3(64 * nw), sh, as_stream(stream), a, NP);
  } else {
    long items = (long)nseq * 4 * ((n + 31) / 32);
    hipLaunchKernelGGL(attn_fwd_kernel, dim3(cdiv(items, 4)), dim3(256), 0, as_stream(stream), a);
  }
  return se_check_launch("se_attn_fwd");
}

// workspace of se_attn_bwd: Dl [ntok][4] | Es [3][R][16] bf16 | Ets [3][16][ET] bf16 | per-wave dE tiles
static inline size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }
struct AttnWs { size_t dl, es, ets, des, total; int R, ET; };
// the v3 kernel runs when the padded sequence fits the offset table and one of its two shapes: returns KT (7: one wave per
// (sequence, head), n <= 112; 6: four waves per (sequence, head), n <= 384) or 0
static int attn_v3_shape(int n, int maxpos) {
  const int nkt = (n + 15) / 16;
  if ((maxpos % 16) != 0 || 16 * nkt + 16 * 8 > maxpos) return 0;
  if (nkt <= 7) return 7;
  if ((nkt + 3) / 4 <= 6) return 6;
  return 0;
}
static AttnWs attn_ws(long ntok, int maxpos, int nseq, int n) {
  AttnWs w;
  w.R = 2 * maxpos + 1;
  w.ET = (w.R + 16 + 15) / 16 * 16;
  w.dl = 0;
  w.es = al256((size_t)ntok * 4 * sizeof(float));
  w.ets = w.es + al256((size_t)3 * w.R * 16 * 2);
  w.des = w.ets + al256((size_t)3 * 16 * w.ET * 2);
  const int kt = attn_v3_shape(n, maxpos), nkt = (n + 15) / 16;
  const size_t witems = (size_t)nseq * 4 * (kt == 6 ? 4 : 1);
  w.total = w.des + (kt ? al256(witems * (size_t)(nkt + kt) * 256 * sizeof(float)) : 0);
  return w;
}
extern "C" size_t se_attn_bwd_workspace_bytes(long ntok, int maxpos, int nseq, int n) {
  return (ntok > 0 && maxpos >= 0 && nseq > 0 && n > 0) ? attn_ws(ntok, maxpos, nseq, n).total : 0;
}

template <int KT, bool GROUP, bool F16 = false>
static int launch_bwd3(const AttnBwd3Args& b, long nwitems, hipStream_t s, float* dE, int phase) {
  using L3 = Lds3<KT, F16>;
  const long items = (long)b.g.nseq * 4;
  const int nkt = (b.g.n + 15) / 16;
  if constexpr (GROUP && F16) {
    // ring form: the four waves' dE tiles are summed in LDS (16 open tiles at most: nkt - cnt_0 + 1 <= 16)
    int span_lo = 1 << 30, span_hi = 0;
    bool ordered = true;
    for (int w = 0, prev = -1; w < 4; ++w) {
      int k0, cn;
      group_split(nkt, w, k0, cn);
      const int tq = k0 + group_window(KT, cn);          // wave w flushes tile D at query tile D + tq
      span_lo = tq < span_lo ? tq : span_lo;
      span_hi = tq > span_hi ? tq : span_hi;
      ordered = ordered && tq > prev;
      prev = tq;
    }
    if (ordered && span_hi - span_lo + 1 <= 16 && getenv("SE_ATTN_NO_RING") == nullptr) {
      const size_t shr = (size_t)4 * L3::WAVE + (size_t)(4 * 256 + 16 * 256) * sizeof(float);
      static unsigned raised_r = 0;
      SE_REQUIRE(se_raise_lds((const void*)attn_bwd3_kernel<KT, true, true, true>, shr, &raised_r), "attn_bwd: cannot raise dynamic LDS limit to %zu", shr);
      if (phase & 1) hipLaunchKernelGGL((attn_bwd3_kernel<KT, true, true, true>), dim3(items), dim3(256), shr, s, b);
      if (phase & 2) hipLaunchKernelGGL(attn_de_reduce_items_kernel, dim3(2 * nkt, cdiv(items, 256)), dim3(256), 0, s, b.dEs, dE, items, nkt,
                                        b.maxpos, b.R);
      return 0;
    }
  }
  const size_t sh = (size_t)4 * L3::WAVE + (GROUP ? 2 * 4 * 256 * sizeof(float) : 0);
  static unsigned raised = 0;
  SE_REQUIRE(se_raise_lds((const void*)attn_bwd3_kernel<KT, GROUP, F16>, sh, &raised), "attn_bwd: cannot raise dynamic LDS limit to %zu", sh);
  if (phase & 1) hipLaunchKernelGGL((attn_bwd3_kernel<KT, GROUP, F16>), dim3(GROUP ? items : cdiv(items, 4)), dim3(256), sh, s, b);
  if (phase & 2) hipLaunchKernelGGL((attn_de_reduce3_kernel<KT, GROUP>), dim3(2 * nkt + 1, cdiv(nwitems, 256)), dim3(256), 0, s, b.dEs,
                                    dE, nwitems, nkt, b.maxpos, b.R);
  return 0;
}

static int attn_bwd_impl(const float* QKV, const float* E, const float* O, const float* dO, const float* LSE, float* dQKV,
                         float* dE, int nseq, int n, int inner, long outer_stride, long inner_stride, long pos_stride, long ntok,
                         int maxpos, float scale, void* ws, size_t ws_bytes, int phase, void* stream,
                         const float* qkv_amax = nullptr, const float* do_amax = nullptr, float* dqkv_amax = nullptr);

extern "C" int se_attn_bwd(const float* QKV, const float* E, const float* O, const float* dO, const float* LSE,
                           float* dQKV, float* dE, int nseq, int n, int inner, long outer_stride,
                           long inner_stride, long pos_stride, long ntok, int maxpos, float scale, void* ws,
                           size_t ws_bytes, void* stream) {
  return attn_bwd_impl(QKV, E, O, dO, LSE, dQKV, dE, nseq, n, inner, outer_stride, inner_stride, pos_stride, ntok, maxpos, scale,
                       ws, ws_bytes, 3, stream);
}

extern "C" int se_attn_bwd_phase(const float* QKV, const float* E, const float* O, const float* dO, const float* LSE,
                                 float* dQKV, float* dE, int nseq, int n, int inner, long outer_stride,
                                 long inner_stride, long pos_stride, long ntok, int maxpos, float scale, void* ws,
                                 size_t ws_bytes, int phase, void* stream) {
  SE_REQUIRE(phase == 1 || phase == 2 || phase == 3, "attn_bwd_phase: phase must be 1, 2 or 3");
  return attn_bwd_impl(QKV, E, O, dO, LSE, dQKV, dE, nseq, n, inner, outer_stride, inner_stride, pos_stride, ntok, maxpos, scale,
                       ws, ws_bytes, phase, stream);
}

extern "C" int se_attn_bwd_f16_phase(const float* QKV, const float* E, const float* O, const float* dO, const float* LSE,
                                     const float* qkv_amax, const float* do_amax, float* dqkv_amax, float* dQKV, float* dE, int nseq,
                                     int n, int inner, long outer_stride, long inner_stride, long pos_stride, long ntok, int maxpos,
                                     float scale, void* ws, size_t ws_bytes, int phase, void* stream) {
  SE_REQUIRE(phase == 1 || phase == 2 || phase == 3, "attn_bwd_f16_phase: phase must be 1, 2 or 3");
  SE_REQUIRE(qkv_amax && do_amax, "attn_bwd_f16: the maxima of QKV and dO are required");
  SE_REQUIRE(attn_v3_shape(n, maxpos) && pos_stride * 192 * (long)(16 * ((n + 15) / 16)) < 2147483647L,
             "attn_bwd_f16: shape outside the split-fp16 kernel (n = %d, maxpos = %d): use se_attn_bwd", n, maxpos);
  return attn_bwd_impl(QKV, E, O, dO, LSE, dQKV, dE, nseq, n, inner, outer_stride, inner_stride, pos_stride, ntok, maxpos, scale,
                       ws, ws_bytes, phase, stream, qkv_amax, do_amax, dqkv_amax);
}

// launch of the workgroup-cooperative backward (se_attn_bwd4.h): the exact-body kernel when every wave of the plan matches one of its
// bodies, the generic-body kernel otherwise
template <int NW, int KPW, int NCW, int NKTM, int MINW, int CA, int NA, int CB, int NB, int CC = -1, int NC_ = 0>
static int launch_bwd4(const AttnBwd3Args& b, const AttnBwd4Plan& pl, int nkt, long items, size_t shr, hipStream_t s) {
  if (attn_bwd4_exact<NW, CA, NA, CB, NB, CC, NC_>(pl, nkt, b.dbg)) {
    static unsigned raised = 0;
    auto kfn = attn_bwd4_kernel<NW, KPW, NCW, NKTM, MINW, CA, NA, CB, NB, CC, NC_, false>;
    SE_REQUIRE(se_raise_lds((const void*)kfn, shr, &raised), "attn_bwd: cannot raise dynamic LDS limit to %zu", shr);
    hipLaunchKernelGGL(kfn, dim3(items), dim3(NW * 64), shr, s, b, pl);
  } else {
    static unsigned raised_g = 0;
    auto kfn = attn_bwd4_kernel<NW, KPW, NCW, NKTM, MINW, CA, NA, CB, NB, CC, NC_, true>;
    SE_REQUIRE(se_raise_lds((const void*)kfn, shr, &raised_g), "attn_bwd: cannot raise dynamic LDS limit to %zu", shr);
    hipLaunchKernelGGL(kfn, dim3(items), dim3(NW * 64), shr, s, b, pl);
  }
  return 0;
}

static int attn_bwd_impl(const float* QKV, const float* E, const float* O, const float* dO, const float* LSE, float* dQKV,
                         float* dE, int nseq, int n, int inner, long outer_stride, long inner_stride, long pos_stride, long ntok,
                         int maxpos, float scale, void* ws, size_t ws_bytes, int phase, void* stream,
                         const float* qkv_amax, const float* do_amax, float* dqkv_amax) {
  SE_REQUIRE(QKV && E && O && dO && LSE && dQKV && dE && ws, "attn_bwd: null operand");
  SE_REQUIRE(ntok > 0 && maxpos >= 0 && nseq > 0 && n > 0, "attn_bwd: bad sizes");
  const AttnWs w = attn_ws(ntok, maxpos, nseq, n);
  SE_REQUIRE(ws_bytes >= w.total, "attn_bwd: workspace of %zu bytes, need %zu (se_attn_bwd_workspace_bytes)", ws_bytes, w.total);
  SE_REQUIRE(((uintptr_t)ws & 15) == 0, "attn_bwd: workspace must be 16-byte aligned");
  float* Dl = reinterpret_cast<float*>((char*)ws + w.dl);
  AttnBwdArgs a{{nseq, n, inner, outer_stride, inner_stride, pos_stride}, QKV, E, dO, LSE, Dl, dQKV, dE, maxpos, scale, 0};
  if (int e = check_geom(a.g)) return e;
  hipStream_t s = as_stream(stream);
  int kt3 = attn_v3_shape(n, maxpos);
  if (const char* e = getenv("SE_ATTN_BWD")) { if (atoi(e) == 2) kt3 = 0; }
  const bool v3 = kt3 && pos_stride * 192 * (long)(16 * ((n + 15) / 16)) < 2147483647L;
  // phase 2 = the reduction of the per-wave dE tiles alone (a leaf of the backward graph: the caller may issue it on another
  // stream once phase 1 has been queued); only the decoupled kernel has one -- the other kernels do everything in phase 1
  if (!(phase & 1) && !v3) return 0;
  // (the cooperative kernels compute delta from the O rows themselves: no table, no launch)
  static const int bwd4_mode0 = getenv("SE_ATTN_BWD4") ? atoi(getenv("SE_ATTN_BWD4")) : 3;
  static const int small_nw0 = getenv("SE_ATTN_BWD4_SMALL_NW") ? atoi(getenv("SE_ATTN_BWD4_SMALL_NW")) : 2;
  // SE_ATTN_DELTA_IN_KERNEL=1 (opt-in): the cooperative kernels compute delta from the O rows (no attn_delta_kernel launch).  Same-box
  // A/B: SLOWER -- n = 321 family 6.31 -> 6.51 ms per step, n = 101 2.77 -> 2.81 (the loader's extra row load and two cross-group adds
  // sit on the per-query-tile chain; the stand-alone pass is 41 us of pure streaming)
  static const bool delta_kernel_forced = !(getenv("SE_ATTN_DELTA_IN_KERNEL") != nullptr && atoi(getenv("SE_ATTN_DELTA_IN_KERNEL")) == 1);
  bool delta_in_kernel = false;
  if (v3 && qkv_amax != nullptr && (n + 15) / 16 <= 21 && (bwd4_mode0 & ((n + 15) / 16 <= 7 ? 1 : 2))) {
    const int nkt0 = (n + 15) / 16;
    const bool sm0 = nkt0 <= 7;
    const AttnBwd4Plan p0 = sm0 ? (small_nw0 == 4 ? attn_bwd4_plan(nkt0, 4, 1) : attn_bwd4_plan(nkt0, 2, 2)) : attn_bwd4_plan(nkt0, 4, 3);
    int kmax0 = 0;
    for (int w4 = 0; w4 < 8; ++w4) kmax0 = p0.cnt[w4] > kmax0 ? p0.cnt[w4] : kmax0;
    delta_in_kernel = !delta_kernel_forced && p0.M > 0 && (sm0 ? kmax0 <= (small_nw0 == 4 ? 2 : 4) : kmax0 <= 6);
  }
  if ((phase & 1) && !delta_in_kernel) hipLaunchKernelGGL(attn_delta_kernel, dim3(cdiv(ntok * 16, 256)), dim3(256), 0, s, dO, O, Dl, ntok);
  if (v3) {
    // decoupled split-bf16 kernel: one wave per (sequence, head[, key group]); no offset clamp can be active
    __bf16* Es = reinterpret_cast<__bf16*>((char*)ws + w.es);
    __bf16* Ets = reinterpret_cast<__bf16*>((char*)ws + w.ets);
    const bool f16 = qkv_amax != nullptr;
    // (F16: two planes in the room of three; the third plane's room of the Es region holds the measured max |E|)
    float* e_amax = reinterpret_cast<float*>((char*)ws + w.es + al256((size_t)2 * w.R * 16 * 2));
    if (phase & 1) {
      SE_REQUIRE(hipMemsetAsync((char*)ws + w.ets, 0, w.des - w.ets, s) == hipSuccess, "attn_bwd: workspace memset failed");
      if (f16) hipLaunchKernelGGL(attn_split_tables_f16_kernel, dim3(16), dim3(1024), 0, s, E, reinterpret_cast<unsigned short*>(Es),
                                  reinterpret_cast<unsigned short*>(Ets), e_amax, w.R, w.ET);
      else hipLaunchKernelGGL(attn_split_tables_kernel, dim3(cdiv((long)w.R * 16, 256)), dim3(256), 0, s, E, Es, Ets, w.R, w.ET);
    }
    AttnBwd3Args b{{nseq, n, inner, outer_stride, inner_stride, pos_stride}, QKV, dO, LSE, Dl, dQKV, Es, Ets,
                   reinterpret_cast<float*>((char*)ws + w.des), w.R, w.ET, maxpos, scale, 0, qkv_amax, do_amax, e_amax, dqkv_amax, O};
    if (const char* e = getenv("SE_ATTN_DBG")) b.dbg = atoi(e);
    if (delta_kernel_forced) b.dbg |= 128;
    const long items = (long)nseq * 4;
    int e;
    // round 4: the workgroup-cooperative kernel (se_attn_bwd4.h) for every shape its two instantiations cover; SE_ATTN_BWD4=0: v3
    static const int bwd4_mode = getenv("SE_ATTN_BWD4") ? atoi(getenv("SE_ATTN_BWD4")) : 3;
    if (f16) {
      const int nkt = (n + 15) / 16;
      // 8 <= nkt <= 21 (n = 321): four waves (two workgroups per CU), up to 6 key tiles and 3 classes per wave; nkt <= 7 (n = 101): two
      // waves with up to 4 key tiles and 2 classes (SE_ATTN_BWD4_SMALL_NW=4: four waves, 2 tiles / 1 class: 0.79 vs 0.77 ms)
      static const int small_nw = getenv("SE_ATTN_BWD4_SMALL_NW") ? atoi(getenv("SE_ATTN_BWD4_SMALL_NW")) : 2;
      const bool small = nkt <= 7;
      const AttnBwd4Plan pl = small ? (small_nw == 4 ? attn_bwd4_plan(nkt, 4, 1) : attn_bwd4_plan(nkt, 2, 2)) : attn_bwd4_plan(nkt, 4, 3);
      int kmax = 0;
      for (int w4 = 0; w4 < 8; ++w4) kmax = pl.cnt[w4] > kmax ? pl.cnt[w4] : kmax;
      const bool fits = pl.M > 0 && (small ? kmax <= (small_nw == 4 ? 2 : 4) : (kmax <= 6 && nkt <= 21));
      const size_t shr = small ? (small_nw == 4 ? attn_bwd4_lds<4, 7>(nkt) : attn_bwd4_lds<2, 7>(nkt)) : attn_bwd4_lds<4, 21>(nkt);
      if (fits && (bwd4_mode & (small ? 1 : 2))) {
        if (phase & 1) {
          if (small && small_nw == 4) e = launch_bwd4<4, 2, 1, 7, 3, 2, 1, 2, 1, 1, 1>(b, pl, nkt, items, shr, s);
          else if (small) e = launch_bwd4<2, 4, 2, 7, 2, 3, 2, 4, 2>(b, pl, nkt, items, shr, s);     // two waves, 256 VGPRs: four workgroups per CU (168 VGPRs: spills in the key phase, 1.00 ms)
          else e = launch_bwd4<4, 6, 3, 21, 2, 5, 3, 6, 2>(b, pl, nkt, items, shr, s);
          if (e) return e;
        }
        if (phase & 2) hipLaunchKernelGGL(attn_de_reduce_items_kernel, dim3(2 * nkt, cdiv(items, 256)), dim3(256), 0, s, b.dEs, dE, items, nkt,
                                          b.maxpos, b.R);
        return se_check_launch("se_attn_bwd");
      }
    }
    if (f16) e = kt3 == 7 ? launch_bwd3<7, false, true>(b, items, s, dE, phase) : launch_bwd3<6, true, true>(b, items * 4, s, dE, phase);
    else e = kt3 == 7 ? launch_bwd3<7, false>(b, items, s, dE, phase) : launch_bwd3<6, true>(b, items * 4, s, dE, phase);
    if (e) return e;
    return se_check_launch("se_attn_bwd");
  }
  // transposed table for the v2 kernel (fp32): built in the Ets region of the workspace
  float* Et = nullptr;
  const int et_ld = (2 * maxpos + 1 + 3) / 4 * 4;
  if (n <= 336 && maxpos >= 352 && (size_t)16 * et_ld * sizeof(float) <= w.total - w.es) {
    Et = reinterpret_cast<float*>((char*)ws + w.es);
    SE_REQUIRE(hipMemsetAsync(Et, 0, (size_t)16 * et_ld * sizeof(float), s) == hipSuccess, "attn_bwd: memset failed");
    hipLaunchKernelGGL(attn_transpose_table_kernel, dim3(cdiv((long)(2 * maxpos + 1) * 16, 256)), dim3(256), 0, s, E, Et,
                       2 * maxpos + 1, et_ld);
  }
  if (Et) {
    // single-pass staged kernel (no clamp aliasing possible: |i-j| < 352 <= maxpos)
    const long items2 = (long)nseq * 4;
    if (n <= 112) {
      const size_t sh = (2 * 112 * 16 + 2 * 16 * 116 + 16 * 228 + 4 * 5 * 320) * sizeof(float);
      static unsigned raised = 0;
  SE_REQUIRE(se_raise_lds((const void*)attn_bwd2_kernel<112, 4, true>, sh, &raised), "attn_bwd: cannot raise dynamic LDS limit");
      int nb = items2 < 512 ? (int)items2 : 512;
      int ipb = (int)((items2 + nb - 1) / nb);
      nb = (int)((items2 + ipb - 1) / ipb);
      hipLaunchKernelGGL((attn_bwd2_kernel<112, 4, true>), dim3(nb), dim3(256), sh, s, a, Et, et_ld, ipb);
    } else {
      const size_t sh = (336 * 16 + 2 * 16 * 340 + 16 * 676 + 8 * 5 * 320) * sizeof(float);
      static unsigned raised = 0;
  SE_REQUIRE(se_raise_lds((const void*)attn_bwd2_kernel<336, 8, false>, sh, &raised), "attn_bwd: cannot raise dynamic LDS limit");
      int nb = items2 < 256 ? (int)items2 : 256;
      int ipb = (int)((items2 + nb - 1) / nb);
      nb = (int)((items2 + ipb - 1) / ipb);
      hipLaunchKernelGGL((attn_bwd2_kernel<336, 8, false>), dim3(nb), dim3(512), sh, s, a, Et, et_ld, ipb);
    }
    return se_check_launch("se_attn_bwd");
  }
  long items = (long)nseq * 4 * ((n + 31) / 32);
  hipLaunchKernelGGL(attn_bwd_dkv_kernel, dim3(cdiv(items, 4)), dim3(256), 0, s, a);
  long qitems = (long)nseq * 4 * ((n + 15) / 16);
  int nblk = 1024;
  int ipb = (int)((qitems + nblk - 1) / nblk);
  ipb = ((ipb + 3) / 4) * 4;
  if (ipb < 4) ipb = 4;
  nblk = (int)((qitems + ipb - 1) / ipb);
  if (n <= 128) {
    size_t sh = (2 * 128 * 16 + 4 * 4 * 256) * sizeof(float);
    hipLaunchKernelGGL((attn_bwd_dq_kernel<128>), dim3(nblk), dim3(256), sh, s, a, ipb);
  } else if (n <= 352) {
    size_t sh = (2 * 352 * 16 + 4 * 4 * 256) * sizeof(float);
    hipLaunchKernelGGL((attn_bwd_dq_kernel<352>), dim3(nblk), dim3(256), sh, s, a, ipb);
  } else {
    size_t sh = (2 * 1024 * 16 + 4 * 4 * 256) * sizeof(float);
    SE_REQUIRE(hipFuncSetAttribute((const void*)attn_bwd_dq_kernel<1024>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)sh) == hipSuccess, "attn_bwd: cannot raise dynamic LDS limit");
    hipLaunchKernelGGL((attn_bwd_dq_kernel<1024>), dim3(nblk), dim3(256), sh, s, a, ipb);
  }
  return se_check_launch("se_attn_bwd");
}
