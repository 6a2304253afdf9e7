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


// (sequence, head) item of workgroup b when ONE workgroup handles one item: workgroups are dealt round-robin to the 8 XCDs, so with
// item = b the four heads of a sequence -- which share every 128-B line of its QKV / dO rows (a head is 64 B of them) -- land on
// four different L2s and every line crosses the fabric twice.  Here the heads of a sequence are the workgroups b, b + 8, b + 16,
// b + 24: same XCD, dispatched together.  Measured (FETCH_SIZE): forward 0.86 -> see DESIGN.md; the tail (grid % 32) keeps item = b.
static __device__ __forceinline__ int xcd_item(int b, int nb) {
  if (b >= (nb & ~31)) return b;
  return (((b >> 5) * 8 + (b & 7)) << 2) | ((b >> 3) & 3);
}
static __device__ __forceinline__ long seq_base(const AttnGeom& g, int s) {
  return (long)(s / g.inner) * g.outer_stride + (long)(s % g.inner) * g.inner_stride;
}
static __device__ __forceinline__ float f4c(const float4& v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : (i == 2 ? v.z : v.w)); }

// =====================================================================================================================
// Split-operand MFMA arithmetic shared by the v3 forward and the v4 backward (se_attn_bwd4.h).
// bf16x6: every product is an exact 3-way bf16 split evaluated with THREE v_mfma_f32_16x16x32_bf16: the contraction length is 16, so
// the two halves of K = 32 carry two different split pairs (k slot (g, j): j < 4 -> first pair, j >= 4 -> second pair, index
// 4g + (j & 3)):  [a_hi|a_mid].[b_lo|b_mid] + [a_hi|a_lo].[b_mid|b_hi] + [a_hi|a_mid].[b_hi|b_hi]  = the six products of the
// fp32-equivalent split (dropped terms <= 2^-24 relative) at 48 instead of 128 matrix-pipe cycles per 16x16x16 product.
// f16x3 (default): scaled (hi, lo) fp16 planes, three v_mfma_f32_16x16x16_f16 per product.
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
// Measurement twins of the scaled-fp16 attention kernels (tools/attn_twin.py builds them as variant libraries; WRONG results, never
// part of the product build): SE_ATTN_TWIN == 1 keeps the MFMAs, the global / LDS traffic and the barriers and compiles the vector
// chain between the products out (softmax, rescale, operand splits of P / dS / W: the raw accumulator registers are handed on as the
// next product's operand words, so every data dependence stays); SE_ATTN_TWIN == 2 keeps the vector chain and the traffic and
// replaces every MFMA by two value-preserving v_fma_f32 (acc + word * 0) that read its operands (accumulators stay 0: logits 0, no
// data-dependent branch is taken more often than in the real kernel's steady state).
#ifndef SE_ATTN_TWIN
#define SE_ATTN_TWIN 0
#endif
#if SE_ATTN_TWIN == 2
static __device__ __forceinline__ f32x4 twin_nomfma_(u32x2 a, u32x2 b, f32x4 c) {
  float c0 = c[0];
  asm volatile("v_fma_f32 %0, %1, 0, %0\n\tv_fma_f32 %0, %2, 0, %0" : "+v"(c0) : "v"(a[0]), "v"(b[1]));
  c[0] = c0;
  return c;
}
#define MFMA_HF(a, b, c) twin_nomfma_((a), (b), (c))
#else
#define MFMA_HF(a, b, c) __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(f16x4a, (a)), __builtin_bit_cast(f16x4a, (b)), (c), 0, 0, 0)
#endif
// (twin 1) four fp32 registers handed on as the (hi, lo) words of a split operand without any arithmetic
static __device__ __forceinline__ S3 twin_raw_(float a, float b, float c, float d) {
  S3 s;
  s.v = (u32x6){0u, 0u, __float_as_uint(a), __float_as_uint(b), __float_as_uint(c), __float_as_uint(d)};
  return s;
}
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


// split tables of the relative-position embedding for the backward: Es[2][R][16] (row fragments), Ets[2][ET/16 tiles][16 d][16 offsets]
// (column fragments; tile-major so that one offset tile is 512 contiguous bytes per plane: 4 cache lines instead of 16) as two fp16
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

struct AttnBwd4Args {
  AttnGeom g;
  const float* QKV; const float* dO; const float* LSE; const float* Dl;      // Dl [tokens][4]: delta = rowsum(dO . O) per head
  float* dQKV;
  const __bf16* Es; const __bf16* Ets;   // two fp16 planes of E * 2^sexp(*e_amax): row fragments [R][16] / tile-major column fragments
  float* dEs;                        // per-item dE tiles: [item][2 nkt][16 offsets][16 d]
  int R, ET, maxpos;
  float scale;
  int dbg;                           // SE_ATTN_DBG: 1 no U of the next tile, 2 no key steps, 4 no strip consumption (timing ablations), 32 stamps
                                     // (diagnostic builds), 64 the generic body for every wave (tests); 0 in production
  // device scalars >= max |QKV|, max |dO|, max |E| (the last one written by the table kernel of the same call)
  const float* qkv_amax; const float* do_amax; const float* e_amax;
  float* dqkv_amax;                  // optional: raised to max |dQKV| (operand scale of the consumers of the gradient)
};

// dE[delta][d] += sum over the (sequence, head) items of their finished offset tiles: dEs [items][2 nkt][256], tile slot = D + nkt,
// D = -nkt .. nkt - 1.  grid = (offset tile, chunk of 256 items); a workgroup streams 256 x 1 KB with 16-byte loads, sums through LDS, 1 KB of atomics.
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
// part of its softmax chain pays for the matrix-pipe stalls: 0.681 -> 0.651 ms (n = 321), 0.270 -> 0.263 (n = 101)
__global__ __launch_bounds__(512) SE_PACKED_FP32_KERNEL void attn_fwd3_kernel(AttnArgs a, int NP, int qsplit) {
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
#if SE_ATTN_TWIN == 1
        {
          float uu[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) uu[r] = Ul[t * 512 + coff[hi][r]];
          o[t] = prodx<F16>(vcol, twin_raw_(s4[0] , s4[1], uu[0], uu[1]), o[t]);
          asm volatile("" :: "v"(s4[2]), "v"(s4[3]), "v"(uu[2]), "v"(uu[3]));
          continue;
        }
#endif
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
        if (!F16 || __builtin_amdgcn_ballot_w64(tmax > m[t] + LAZY) != 0) {
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

// Forward: the v3 kernel (split-bf16 or scaled split-fp16 products, V of one (sequence, head) staged in LDS) whenever the sequence
// fits it, the streaming fp32-MFMA kernel of round 1 (any length, K / V fragments straight from memory) otherwise.
static int attn_fwd_impl(const float* QKV, const float* E, const void* Es, long es_plane, float* O, float* LSE, int nseq, int n,
                         int inner, long outer_stride, long inner_stride, long pos_stride, int maxpos, float scale, void* stream,
                         const float* qkv_amax, const float* e_amax) {
  AttnArgs a{{nseq, n, inner, outer_stride, inner_stride, pos_stride}, QKV, E, O, LSE, maxpos, scale, Es, es_plane, qkv_amax, e_amax};
  const bool f16 = qkv_amax != nullptr;
  if (int e = check_geom(a.g)) return e;
  SE_REQUIRE(QKV && E && O, "attn_fwd: null operand");
  const int NP = ((n + 15) / 16) * 16;
  if (pos_stride * 192 * (long)NP < 2147483647L) {
    // waves per workgroup / 16-query tiles per wave block, measured at B = 16 (ms): n = 321: 8 waves x 2 tiles 0.92, 7 x 2 0.95,
    // 8 x 1 0.94, 4 x 2 0.97, 6 x 2 1.10; n = 101: 4 x 1 0.342, 8 x 2 0.352, 4 x 2 0.367 -- the per-step chain S -> softmax -> PV
    // is latency-bound with 2 waves per SIMD, 8-wave workgroups put 4 there
    const int qb32 = (n + 31) / 32;
    const int nw3 = qb32 <= 4 ? 4 : 8, tq3 = qb32 <= 4 ? 1 : 2;
    // fewer items than two rounds of workgroups (batch-1 inference): split the query blocks of an item over several workgroups
    int qsplit = 1;
    {
      const long items = (long)nseq * 4, qblk = (n + 16 * tq3 - 1) / (16 * tq3);
      while (items * qsplit < 1024 && (long)nw3 * qsplit * 2 <= qblk && qsplit < 8) ++qsplit;
      if (const char* e = getenv("SE_ATTN_FWD_QSPLIT")) { int v = atoi(e); if (v >= 1 && v <= 16) qsplit = v; }      // (tests force it)
    }
    const size_t sh3 = (size_t)(f16 ? 2 : 3) * NP * 32 + (size_t)512 * tq3 * nw3 * sizeof(float);
    if (sh3 <= 160 * 1024) {
      static unsigned raised3[2][3] = {{0, 0, 0}, {0, 0, 0}};
      const void* fn = f16 ? (tq3 == 1 ? (const void*)attn_fwd3_kernel<1, true> : (const void*)attn_fwd3_kernel<2, true>)
                           : (tq3 == 1 ? (const void*)attn_fwd3_kernel<1> : (const void*)attn_fwd3_kernel<2>);
      SE_REQUIRE(se_raise_lds(fn, 160 * 1024, &raised3[f16][tq3]), "attn_fwd: cannot raise the dynamic LDS limit");
      const dim3 grid3((unsigned)(nseq * 4 * qsplit));
      if (f16 && tq3 == 1) hipLaunchKernelGGL((attn_fwd3_kernel<1, true>), grid3, dim3(64 * nw3), sh3, as_stream(stream), a, NP, qsplit);
      else if (f16) hipLaunchKernelGGL((attn_fwd3_kernel<2, true>), grid3, dim3(64 * nw3), sh3, as_stream(stream), a, NP, qsplit);
      else if (tq3 == 1) hipLaunchKernelGGL(attn_fwd3_kernel<1>, grid3, dim3(64 * nw3), sh3, as_stream(stream), a, NP, qsplit);
      else hipLaunchKernelGGL(attn_fwd3_kernel<2>, grid3, dim3(64 * nw3), sh3, as_stream(stream), a, NP, qsplit);
      return se_check_launch("se_attn_fwd");
    }
  }
  SE_REQUIRE(!f16, "attn_fwd_f16: the sequence does not fit the split-fp16 kernel (n = %d)", n);
  const long items = (long)nseq * 4 * ((n + 31) / 32);
  hipLaunchKernelGGL(attn_fwd_kernel, dim3(cdiv(items, 4)), dim3(256), 0, as_stream(stream), a);
  return se_check_launch("se_attn_fwd");
}

// workspace of se_attn_bwd: Dl [ntok][4] | Es [2][R][16] fp16 + max |E| | Ets [2][16][ET] fp16 | per-item dE tiles [nseq * 4][2 nkt][256] fp32
static inline size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }
struct AttnWs { size_t dl, es, ets, des, total; int R, ET; };
// the cooperative backward (se_attn_bwd4.h) takes a sequence when its padded length fits the offset table without a clamp and its
// key tiles fit one of the two instantiations (<= 7: two waves, <= 21: four waves)
static bool attn_bwd4_shape(int n, int maxpos, long pos_stride) {
  const int nkt = (n + 15) / 16;
  return (maxpos % 16) == 0 && 16 * nkt + 16 * 8 <= maxpos && nkt <= 21 && pos_stride * 192 * (long)(16 * nkt) < 2147483647L;
}
static AttnWs attn_ws(long ntok, int maxpos, int nseq, int n) {
  AttnWs w;
  w.R = 2 * maxpos + 1;
  w.ET = (w.R + 16 + 15) / 16 * 16;
  w.dl = 0;
  w.es = al256((size_t)ntok * 4 * sizeof(float));
  w.ets = w.es + al256((size_t)2 * w.R * 16 * 2) + 256;          // (+ 256: the measured max |E| behind the two planes)
  w.des = w.ets + al256((size_t)2 * 16 * w.ET * 2);
  const int nkt = (n + 15) / 16;
  w.total = w.des + (attn_bwd4_shape(n, maxpos, 1) ? al256((size_t)nseq * 4 * (size_t)(2 * nkt) * 256 * sizeof(float)) : 0);
  return w;
}
extern "C" size_t se_attn_bwd_workspace_bytes(long ntok, int maxpos, int nseq, int n) {
  return (ntok > 0 && maxpos >= 0 && nseq > 0 && n > 0) ? attn_ws(ntok, maxpos, nseq, n).total : 0;
}

static int attn_bwd_impl(const float* QKV, const float* E, const float* O, const float* dO, const float* LSE, const float* delta,
                         float* dQKV, float* dE, int nseq, int n, int inner, long outer_stride, long inner_stride, long pos_stride,
                         long ntok, int maxpos, float scale, void* ws, size_t ws_bytes, int phase, void* stream,
                         const float* qkv_amax = nullptr, const float* do_amax = nullptr, float* dqkv_amax = nullptr);

extern "C" int se_attn_bwd(const float* QKV, const float* E, const float* O, const float* dO, const float* LSE,
                           float* dQKV, float* dE, int nseq, int n, int inner, long outer_stride,
                           long inner_stride, long pos_stride, long ntok, int maxpos, float scale, void* ws,
                           size_t ws_bytes, void* stream) {
  return attn_bwd_impl(QKV, E, O, dO, LSE, nullptr, dQKV, dE, nseq, n, inner, outer_stride, inner_stride, pos_stride, ntok, maxpos, scale,
                       ws, ws_bytes, 3, stream);
}

extern "C" int se_attn_bwd_f16_phase(const float* QKV, const float* E, const float* O, const float* dO, const float* LSE,
                                     const float* delta, const float* qkv_amax, const float* do_amax, float* dqkv_amax, float* dQKV,
                                     float* dE, int nseq, int n, int inner, long outer_stride, long inner_stride, long pos_stride,
                                     long ntok, int maxpos, float scale, void* ws, size_t ws_bytes, int phase, void* stream) {
  SE_REQUIRE(phase == 1 || phase == 2 || phase == 3, "attn_bwd_f16_phase: phase must be 1, 2 or 3");
  SE_REQUIRE(qkv_amax && do_amax, "attn_bwd_f16: the maxima of QKV and dO are required");
  SE_REQUIRE(O || delta, "attn_bwd_f16: the forward output O or the table delta = rowsum(dO . O) [tokens][4] is required");
  SE_REQUIRE(attn_bwd4_shape(n, maxpos, pos_stride),
             "attn_bwd_f16: shape outside the split-fp16 kernel (n = %d, maxpos = %d): use se_attn_bwd", n, maxpos);
  return attn_bwd_impl(QKV, E, O, dO, LSE, delta, dQKV, dE, nseq, n, inner, outer_stride, inner_stride, pos_stride, ntok, maxpos, scale,
                       ws, ws_bytes, phase, stream, qkv_amax, do_amax, dqkv_amax);
}

// launch of the workgroup-cooperative backward (se_attn_bwd4.h): the exact-body kernel when every wave of the plan matches one of its
// bodies, the generic-body kernel otherwise
template <int NW, int KPW, int NCW, int NKTM, int MINW, int CA, int NA, int CB, int NB>
static int launch_bwd4(const AttnBwd4Args& b, const AttnBwd4Plan& pl, int nkt, long items, size_t shr, hipStream_t s) {
  if (attn_bwd4_exact<NW, CA, NA, CB, NB, -1, 0>(pl, nkt, b.dbg)) {
    static unsigned raised = 0;
    auto kfn = attn_bwd4_kernel<NW, KPW, NCW, NKTM, MINW, CA, NA, CB, NB, -1, 0, false>;
    SE_REQUIRE(se_raise_lds((const void*)kfn, shr, &raised), "attn_bwd: cannot raise dynamic LDS limit to %zu", shr);
    hipLaunchKernelGGL(kfn, dim3(items), dim3(NW * 64), shr, s, b, pl);
  } else {
    static unsigned raised_g = 0;
    auto kfn = attn_bwd4_kernel<NW, KPW, NCW, NKTM, MINW, CA, NA, CB, NB, -1, 0, true>;
    SE_REQUIRE(se_raise_lds((const void*)kfn, shr, &raised_g), "attn_bwd: cannot raise dynamic LDS limit to %zu", shr);
    hipLaunchKernelGGL(kfn, dim3(items), dim3(NW * 64), shr, s, b, pl);
  }
  return 0;
}

// Backward.  With the operand maxima (se_attn_bwd_f16_phase): the workgroup-cooperative scaled split-fp16 kernel; phase 1 = delta (unless
// the caller supplies the table), the split tables of E, the kernel; phase 2 = the reduction of the per-item dE tiles (a leaf of the backward
// graph: the caller may issue it on another stream once phase 1 has been queued).  Without them (se_attn_bwd): the fp32-MFMA kernels
// of round 1 for any sequence length (dK / dV pass + dQ / dE pass), everything in phase 1.
static int attn_bwd_impl(const float* QKV, const float* E, const float* O, const float* dO, const float* LSE, const float* delta,
                         float* dQKV, float* dE, int nseq, int n, int inner, long outer_stride, long inner_stride, long pos_stride,
                         long ntok, int maxpos, float scale, void* ws, size_t ws_bytes, int phase, void* stream,
                         const float* qkv_amax, const float* do_amax, float* dqkv_amax) {
  SE_REQUIRE(QKV && E && dO && LSE && dQKV && dE && ws, "attn_bwd: null operand");
  SE_REQUIRE(ntok > 0 && maxpos >= 0 && nseq > 0 && n > 0, "attn_bwd: bad sizes");
  const AttnWs w = attn_ws(ntok, maxpos, nseq, n);
  SE_REQUIRE(ws_bytes >= w.total, "attn_bwd: workspace of %zu bytes, need %zu (se_attn_bwd_workspace_bytes)", ws_bytes, w.total);
  SE_REQUIRE(((uintptr_t)ws & 15) == 0, "attn_bwd: workspace must be 16-byte aligned");
  float* Dl = reinterpret_cast<float*>((char*)ws + w.dl);
  const AttnGeom geom{nseq, n, inner, outer_stride, inner_stride, pos_stride};
  if (int e = check_geom(geom)) return e;
  hipStream_t s = as_stream(stream);
  const bool f16 = qkv_amax != nullptr;
  if (!f16 && !(phase & 1)) return 0;
  if ((phase & 1) && !delta) {
    SE_REQUIRE(O, "attn_bwd: the forward output O is required");
    hipLaunchKernelGGL(attn_delta_kernel, dim3(cdiv(ntok * 16, 256)), dim3(256), 0, s, dO, O, Dl, ntok);
  }
  if (f16) {
    __bf16* Es = reinterpret_cast<__bf16*>((char*)ws + w.es);
    __bf16* Ets = reinterpret_cast<__bf16*>((char*)ws + w.ets);
    float* e_amax = reinterpret_cast<float*>((char*)ws + w.es + al256((size_t)2 * w.R * 16 * 2));
    float* rep = reinterpret_cast<float*>((char*)ws + w.des);
    const int nkt = (n + 15) / 16;
    if (phase & 1) {
      // (the column-fragment table's padding starts at zero)
      SE_REQUIRE(hipMemsetAsync((char*)ws + w.ets, 0, w.des - w.ets, s) == hipSuccess, "attn_bwd: workspace memset failed");
      hipLaunchKernelGGL(attn_split_tables_f16_kernel, dim3(16), dim3(1024), 0, s, E, reinterpret_cast<unsigned short*>(Es),
                         reinterpret_cast<unsigned short*>(Ets), e_amax, w.R, w.ET);
      AttnBwd4Args b{geom, QKV, dO, LSE, delta ? delta : Dl, dQKV, Es, Ets, rep, w.R, w.ET, maxpos, scale, 0, qkv_amax, do_amax, e_amax, dqkv_amax};
      if (const char* e = getenv("SE_ATTN_DBG")) b.dbg = atoi(e);
      const long items = (long)nseq * 4;
      // 8 <= nkt <= 21 (n = 321): four waves (two workgroups per CU), up to 6 key tiles and 3 classes per wave; nkt <= 7 (n = 101): two
      // waves with up to 4 key tiles and 2 classes, 256 VGPRs: four workgroups per CU
      const bool small = nkt <= 7;
      const AttnBwd4Plan pl = small ? attn_bwd4_plan(nkt, 2, 2) : attn_bwd4_plan(nkt, 4, 3);
      int kmax = 0;
      for (int w4 = 0; w4 < 8; ++w4) kmax = pl.cnt[w4] > kmax ? pl.cnt[w4] : kmax;
      SE_REQUIRE(pl.M > 0 && kmax <= (small ? 4 : 6), "attn_bwd_f16: no plan for %d key tiles", nkt);
      const size_t shr = small ? attn_bwd4_lds<2, 7>(nkt) : attn_bwd4_lds<4, 21>(nkt);
      int e = small ? launch_bwd4<2, 4, 2, 7, 2, 3, 2, 4, 2>(b, pl, nkt, items, shr, s) : launch_bwd4<4, 6, 3, 21, 2, 5, 3, 6, 2>(b, pl, nkt, items, shr, s);
      if (e) return e;
    }
    if (phase & 2) {
      const long items = (long)nseq * 4;
      hipLaunchKernelGGL(attn_de_reduce_items_kernel, dim3(2 * nkt, cdiv(items, 256)), dim3(256), 0, s, rep, dE, items, nkt, maxpos, w.R);
    }
    return se_check_launch("se_attn_bwd");
  }
  AttnBwdArgs a{geom, QKV, E, dO, LSE, delta ? delta : Dl, dQKV, dE, maxpos, scale, 0};
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
    static unsigned raised = 0;
    SE_REQUIRE(se_raise_lds((const void*)attn_bwd_dq_kernel<1024>, sh, &raised), "attn_bwd: cannot raise dynamic LDS limit");
    hipLaunchKernelGGL((attn_bwd_dq_kernel<1024>), dim3(nblk), dim3(256), sh, s, a, ipb);
  }
  return se_check_launch("se_attn_bwd");
}

