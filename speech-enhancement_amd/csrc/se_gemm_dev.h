// Device-side building blocks shared by the tap-GEMM translation units (se_gemm.hip, se_ff.hip, se_wgrad.hip):
// descriptor decoding, XCD-aware work placement, prologues, the counter-based dropout hash, the vectorised epilogues,
// the bf16 operand split.  Everything is static / inline: each TU gets its own copy.
#pragma once
#include "se_common.h"
#include <stdlib.h>

struct GemmArgs {
  se_gemm_desc d;
  const float* A; const float* W; const float* bias; float* Y; const float* R; float* AUX;
  const float* rowstats; const float* ps; const float* pb; double* stats;
  float* dgamma; float* dbeta;     // SE_EPI_LN_BWD_ (se_gemm_ln_bwd): LayerNorm parameter gradients
  int ncb;      // column blocks per row tile
  int tiles;    // row tiles per batch entry
  int nouter;   // B * tiles
  int contig;   // 1: every XCD sweeps a contiguous range of row tiles (tap convolutions: the dt-shifted rows of a
                //    tile are the dt = 0 rows of a tile the same L2 has just seen); 0: round-robin
  float* amax_out = nullptr;   // SE_EPI_LN_BWD_: raised to max |dX| (operand scale of the scaled split-fp16 kernels reading dX)
};

// Workgroups are dispatched round-robin over the 8 XCDs, each with a private L2.  The `ninner` siblings of one
// `outer` work item (column blocks sharing an A tile; (tap, channel, n) blocks sharing a row chunk) are decoded so
// that they sit on the same XCD and next to each other in dispatch order: the shared operand is fetched into that
// L2 once instead of once per sibling from the fabric.  The launch pads `nouter` to a multiple of 8.
struct WorkId { int inner, outer; };
static __device__ __forceinline__ WorkId decode_work(int ninner, int nouter, int contig) {
  const unsigned w = blockIdx.x, xcd = w & 7u, slot = w >> 3;
  const unsigned inner = slot % (unsigned)ninner, ol = slot / (unsigned)ninner;
  const unsigned per = ((unsigned)nouter + 7u) >> 3;
  return {(int)inner, (int)(contig ? xcd * per + ol : ol * 8u + xcd)};
}

// source pixel (index inside batch entry b's grid) of output pixel (t, f) for one tap; -1 when outside
static __device__ __forceinline__ int src_pixel_in(const se_gemm_desc& d, int t, int f, int tap) {
  int ti, fi;
  if (!d.up) {
    ti = t * d.st + d.dt[tap];
    fi = f * d.sf + d.df[tap];
    if (ti < 0 || ti >= d.Ti || fi < 0 || fi >= d.Fi) return -1;
  } else {
    int tt = t + d.dt[tap], ff = f + d.df[tap];
    if (tt < 0 || ff < 0 || (tt % d.st) != 0 || (ff % d.sf) != 0) return -1;
    ti = tt / d.st; fi = tt >= 0 ? ff / d.sf : 0;
    if (ti >= d.Ti || fi >= d.Fi) return -1;
  }
  return ti * d.Fi + fi;
}
static __device__ __forceinline__ long src_pixel(const se_gemm_desc& d, int b, int t, int f, int tap) {
  int ti, fi;
  if (!d.up) {
    ti = t * d.st + d.dt[tap];
    fi = f * d.sf + d.df[tap];
    if (ti < 0 || ti >= d.Ti || fi < 0 || fi >= d.Fi) return -1;
  } else {
    int tt = t + d.dt[tap], ff = f + d.df[tap];
    if (tt < 0 || ff < 0 || (tt % d.st) != 0 || (ff % d.sf) != 0) return -1;
    ti = tt / d.st; fi = ff / d.sf;
    if (ti >= d.Ti || fi >= d.Fi) return -1;
  }
  return ((long)b * d.Ti + ti) * d.Fi + fi;
}

// Counter-based dropout mask.  Elements are hashed in aligned groups of 4 (every user processes float4s): one murmur3
// finalizer of (seed, idx >> 2) plus one multiply-xorshift step give 64 bits = four 16-bit fields, element j of the
// group is kept iff field_j >= thr16 = round(p * 65536); survivors are scaled by 65536 / (65536 - thr16), the exact
// inverse of the realised keep probability.  (The per-element 32-bit hash this replaces cost 3 quarter-rate integer
// multiplies per element -- more issue slots than the Swish it was fused with.)  The same (seed, index) pair is
// re-evaluated in the backward kernels, so no mask is ever stored.
static __device__ __forceinline__ void drop_fields(unsigned seed, unsigned grp, unsigned (&f)[4]) {
  unsigned x = grp * 0x9E3779B1u ^ seed;
  x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
  unsigned y = x * 0x9E3779B1u + 0x7F4A7C15u;
  y ^= y >> 15;
  f[0] = x & 0xFFFFu; f[1] = x >> 16; f[2] = y & 0xFFFFu; f[3] = y >> 16;
}
// the same fields from grp * 0x9E3779B1 supplied by the caller (a kernel that walks groups at fixed distances adds constants instead of
// paying the quarter-rate multiply per group); KEEP MASK of the 4 elements: bit j set iff element j survives
static __device__ __forceinline__ unsigned drop_keep4_pre(unsigned seed, unsigned grp_times_c, unsigned thr) {
  unsigned x = grp_times_c ^ seed;
  x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
  unsigned y = x * 0x9E3779B1u + 0x7F4A7C15u;
  y ^= y >> 15;
  return ((x & 0xFFFFu) >= thr ? 1u : 0u) | ((x >> 16) >= thr ? 2u : 0u) | ((y & 0xFFFFu) >= thr ? 4u : 0u) | ((y >> 16) >= thr ? 8u : 0u);
}
// scales of the 4 elements idx .. idx + 3 (idx a multiple of 4)
static __device__ __forceinline__ float4 drop_scale4(unsigned seed, unsigned idx, unsigned thr, float inv_keep) {
  unsigned f[4];
  drop_fields(seed, idx >> 2, f);
  return make_float4(f[0] >= thr ? inv_keep : 0.f, f[1] >= thr ? inv_keep : 0.f, f[2] >= thr ? inv_keep : 0.f,
                     f[3] >= thr ? inv_keep : 0.f);
}
static __device__ __forceinline__ float drop_scale(unsigned seed, unsigned idx, unsigned thr, float inv_keep) {
  unsigned f[4];
  drop_fields(seed, idx >> 2, f);
  const unsigned j = idx & 3u;
  const unsigned fj = j == 0 ? f[0] : (j == 1 ? f[1] : (j == 2 ? f[2] : f[3]));
  return fj >= thr ? inv_keep : 0.f;
}
static __device__ __forceinline__ unsigned drop_thr(float p) { return (unsigned)(p * 65536.0f + 0.5f); }
static __device__ __forceinline__ float drop_inv_keep(float p) { return 65536.0f / (65536.0f - (float)drop_thr(p)); }

// ps4 / pb4: the per-channel scale / shift of this float4's 4 channels (LN gamma / beta, BN-affine), fetched with the
// tile as two 16-B loads -- per-element scalar loads here cost 8 VMEM instructions per float4 of A.
template <int PRO>
static __device__ __forceinline__ float4 apply_pro(float4 v, int c, int C, float mean, float rstd,
                                                   float4 ps4, float4 pb4, long pix, unsigned seed,
                                                   unsigned thr, float inv_keep) {
  if (PRO == SE_PRO_NONE) return v;
  float x[4] = {v.x, v.y, v.z, v.w};
  const float ps[4] = {ps4.x, ps4.y, ps4.z, ps4.w}, pb[4] = {pb4.x, pb4.y, pb4.z, pb4.w};
  float dsc[4] = {1.f, 1.f, 1.f, 1.f};
  if (PRO == SE_PRO_SWISH_DROP || PRO == SE_PRO_DROP) {      // c and C are multiples of 4: one aligned group
    const float4 d4 = drop_scale4(seed, (unsigned)(pix * C + c), thr, inv_keep);
    dsc[0] = d4.x; dsc[1] = d4.y; dsc[2] = d4.z; dsc[3] = d4.w;
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    int cc = c + j;
    if (cc < C) {
      if (PRO == SE_PRO_LN) x[j] = (x[j] - mean) * rstd * ps[j] + pb[j];
      else if (PRO == SE_PRO_SWISH) x[j] = swishf_(x[j]);
      else if (PRO == SE_PRO_AFFINE_SWISH) x[j] = swishf_(x[j] * ps[j] + pb[j]);
      else if (PRO == SE_PRO_SWISH_DROP) x[j] = swishf_(x[j]) * dsc[j];
      else if (PRO == SE_PRO_DROP) x[j] = x[j] * dsc[j];
    } else {
      x[j] = 0.f;
    }
  }
  return make_float4(x[0], x[1], x[2], x[3]);
}
template <int PRO>
static __device__ __forceinline__ void load_pro_vec(const float* ps, const float* pb, int c, bool ok, float4& ps4, float4& pb4) {
  if (PRO == SE_PRO_LN || PRO == SE_PRO_AFFINE_SWISH) {
    ps4 = ok ? *reinterpret_cast<const float4*>(ps + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    pb4 = ok ? *reinterpret_cast<const float4*>(pb + c) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
}

// Vectorised epilogue (bias / dropout / swish-gradient / residual / accumulate / plain store): each 32x32 accumulator
// is transposed through a wave-private LDS patch so that every lane then owns 4 consecutive output columns: the
// AUX / R reads and the Y writes are 16-byte accesses (8 lanes = one 128-B row segment) and there are 4 of them per
// lane and tile instead of 16 four-byte ones.  cs: the wave's [32][cs_ld] patch (reuses the A staging tile).
// NOLOAD: the caller guarantees (host-checked) that none of the flags with a global LOAD in this epilogue is set (accumulate,
// residual, swish gradient): their branches vanish at compile time.  With run-time flags every row group's store sat behind an
// `s_waitcnt vmcnt(0)` for a load that was never issued -- which also drains the PREVIOUS group's store and any prefetch in flight:
// the stores of a wave were serialised at one memory round trip each (24 per 256-row tile of the qkv projection = 17 us per tile).
// DLT: the caller may be sent SE_EPI_DELTA (the row GEMM kernels only: conv3 / W-stationary forms compile the branch out)
template <bool HASPRE = false, bool HOISTR = true, bool NOLOAD = false, bool DLT = false>      // HOISTR: request an in-place residual for all row groups up front (16 VGPRs)
static __device__ __forceinline__ void gemm_epilogue_vec(const GemmArgs& g, const f32x16& acc0, const f32x16& acc1,
                                                         int m0, int by, int b, float* cs, int cs_ld, unsigned thr,
                                                         float inv_keep, float* red, const float* bias_s,
                                                         const float4 (&pre)[8] = {},     // pre[nt*4+i]: AUX / R values fetched early
                                                         float* vmax_defer = nullptr) {   // persistent callers: see below
  const se_gemm_desc& d = g.d;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int Mb = d.To * d.Fo, ep = (NOLOAD ? (d.epilogue & ~(SE_EPI_ACCUM | SE_EPI_RESID | SE_EPI_SWISH_GRAD)) : d.epilogue) & (DLT ? ~0 : ~SE_EPI_DELTA);
  const long ptile = (long)b * Mb + m0;
  float* __restrict__ Yb = g.Y + ptile * d.ldc + d.c_off;
  const float* __restrict__ Xb = g.AUX ? g.AUX + ptile * d.ldx + d.x_off : nullptr;
  const float* __restrict__ Rb = g.R ? g.R + ptile * d.ldr + d.r_off : nullptr;
  const unsigned pdrop = (unsigned)ptile;
  const int col = lane & 31, half = lane >> 5;
  const int cq = lane & 7, rr = lane >> 3;          // read-back role: float4 column, row within an 8-row pass
  const bool rowstats = (ep & SE_EPI_ROWSTATS) != 0;  // N == 64 (host-checked): the 8 lanes cq = 0..7 of an rr group hold a whole row
  float4 kept[2][4];
  float vmax = 0.f;                                  // max |stored value| (se_gemm_desc.y_amax)
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const f32x16& acc = nt ? acc1 : acc0;
#pragma unroll
    for (int r = 0; r < 16; ++r) cs[((r & 3) + 8 * (r >> 2) + 4 * half) * cs_ld + col] = acc[r];
    const int n = by * 64 + nt * 32 + cq * 4;        // first of this lane's 4 output columns
    float4 ssum = make_float4(0.f, 0.f, 0.f, 0.f), qsum = ssum;
    if (n < d.N) {                                   // N % 4 == 0 (host-checked)
      const float4 bias4 = *reinterpret_cast<const float4*>(bias_s + nt * 32 + cq * 4);   // staged before the K loop
      // the accumulate operand of the four row groups is requested up front (four named registers quads, not an array: an array
      // indexed under these conditions goes to scratch): loaded where it is used, every `load -> s_waitcnt vmcnt(0) -> store`
      // exposed one full memory latency per group -- the store of group i may alias the load of group i + 1
      float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0, a3 = a0;
      if (ep & SE_EPI_ACCUM) {
        const unsigned rb = (unsigned)(wave * 32 + rr), co = (unsigned)n;
        if (m0 + (int)rb < Mb) a0 = *reinterpret_cast<const float4*>(Yb + (rb * (unsigned)d.ldc + co));
        if (m0 + (int)rb + 8 < Mb) a1 = *reinterpret_cast<const float4*>(Yb + ((rb + 8) * (unsigned)d.ldc + co));
        if (m0 + (int)rb + 16 < Mb) a2 = *reinterpret_cast<const float4*>(Yb + ((rb + 16) * (unsigned)d.ldc + co));
        if (m0 + (int)rb + 24 < Mb) a3 = *reinterpret_cast<const float4*>(Yb + ((rb + 24) * (unsigned)d.ldc + co));
      }
      // ... and likewise the residual when it was not fetched before the K loop
      float4 r0 = make_float4(0.f, 0.f, 0.f, 0.f), r1 = r0, r2 = r0, r3 = r0;
      const bool res_here = HOISTR && !HASPRE && (ep & (SE_EPI_RESID | SE_EPI_DELTA)) != 0;      // (DELTA: R = the attention output O)
      if (res_here) {
        const unsigned rb = (unsigned)(wave * 32 + rr), co = (unsigned)n;
        if (m0 + (int)rb < Mb) r0 = *reinterpret_cast<const float4*>(Rb + (rb * (unsigned)d.ldr + co));
        if (m0 + (int)rb + 8 < Mb) r1 = *reinterpret_cast<const float4*>(Rb + ((rb + 8) * (unsigned)d.ldr + co));
        if (m0 + (int)rb + 16 < Mb) r2 = *reinterpret_cast<const float4*>(Rb + ((rb + 16) * (unsigned)d.ldr + co));
        if (m0 + (int)rb + 24 < Mb) r3 = *reinterpret_cast<const float4*>(Rb + ((rb + 24) * (unsigned)d.ldr + co));
      }
      // two phases: every row group's value first (whatever loads the flags ask for are waited for HERE), then the four stores
      // back to back.  With the store inside the first loop each group's store sat behind the `s_waitcnt vmcnt(0)` of the next
      // group's (possibly never issued) load -- which drains the store before it too: one memory round trip per group.
      float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0, v2 = v0, v3 = v0;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = wave * 32 + rr + 8 * i;
        if (m0 + row >= Mb) continue;
        float4 v = *reinterpret_cast<const float4*>(&cs[(rr + 8 * i) * cs_ld + cq * 4]);
        v.x += bias4.x; v.y += bias4.y; v.z += bias4.z; v.w += bias4.w;
        if (ep & SE_EPI_STATS) {
          ssum.x += v.x; ssum.y += v.y; ssum.z += v.z; ssum.w += v.w;
          qsum.x += v.x * v.x; qsum.y += v.y * v.y; qsum.z += v.z * v.z; qsum.w += v.w * v.w;
        }
        if (ep & SE_EPI_DROP) {
          const unsigned pe = (pdrop + (unsigned)row) * (unsigned)d.N + (unsigned)n;
          const float4 d4 = drop_scale4(d.epi_seed, pe, thr, inv_keep);       // pe is a multiple of 4 (N % 4 == 0, n % 4 == 0)
          v.x *= d4.x; v.y *= d4.y; v.z *= d4.z; v.w *= d4.w;
        }
        if (ep & SE_EPI_SWISH_GRAD) {
          float4 z = HASPRE ? pre[nt * 4 + i] : *reinterpret_cast<const float4*>(Xb + ((unsigned)row * (unsigned)d.ldx + (unsigned)n));
          v.x *= swish_gradf_(z.x); v.y *= swish_gradf_(z.y); v.z *= swish_gradf_(z.z); v.w *= swish_gradf_(z.w);
        }
        if (ep & SE_EPI_RESID) {
          float4 rv = HASPRE ? pre[nt * 4 + i] : (HOISTR ? (i == 0 ? r0 : (i == 1 ? r1 : (i == 2 ? r2 : r3)))
                                                         : *reinterpret_cast<const float4*>(Rb + ((unsigned)row * (unsigned)d.ldr + (unsigned)n)));
          v.x = rv.x + d.alpha * v.x; v.y = rv.y + d.alpha * v.y; v.z = rv.z + d.alpha * v.z; v.w = rv.w + d.alpha * v.w;
        }
        if (ep & SE_EPI_ACCUM) { const float4 o = i == 0 ? a0 : (i == 1 ? a1 : (i == 2 ? a2 : a3)); v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
#ifdef SE_EPI_ONE_PHASE      // (A/B builds: the store inside the first loop, as before)
        *reinterpret_cast<float4*>(Yb + ((unsigned)row * (unsigned)d.ldc + (unsigned)n)) = v;
#endif
        if (ep & SE_EPI_DELTA) {
          // softmax-backward row constant of the attention backward, delta[row][head] = sum over the head's 16 columns of dO * O, where
          // dO is THIS result (to_out input gradient): the 4 lanes cq = 4 h' .. 4 h' + 3 hold a head's columns (N == 64, host-checked)
          const float4 ov = HASPRE ? pre[nt * 4 + i] : (i == 0 ? r0 : (i == 1 ? r1 : (i == 2 ? r2 : r3)));
          float dl = v.x * ov.x + v.y * ov.y + v.z * ov.z + v.w * ov.w;
          dl += __shfl_xor(dl, 1, 64);
          dl += __shfl_xor(dl, 2, 64);
          if ((cq & 3) == 0) g.AUX[(ptile + row) * 4 + nt * 2 + (cq >> 2)] = dl;
        }
        if (i == 0) v0 = v; else if (i == 1) v1 = v; else if (i == 2) v2 = v; else v3 = v;
        if (g.amax_out) vmax = fmaxf(fmaxf(vmax, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
        if (rowstats) kept[nt][i] = v;
      }
#ifndef SE_EPI_ONE_PHASE
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = wave * 32 + rr + 8 * i;
        if (m0 + row >= Mb) continue;
        *reinterpret_cast<float4*>(Yb + ((unsigned)row * (unsigned)d.ldc + (unsigned)n)) = i == 0 ? v0 : (i == 1 ? v1 : (i == 2 ? v2 : v3));
      }
#endif
    }
    if (ep & SE_EPI_STATS) {      // fold the 8 row-lanes that share this column group, park per-wave partials in LDS
      float sv[8] = {ssum.x, ssum.y, ssum.z, ssum.w, qsum.x, qsum.y, qsum.z, qsum.w};
#pragma unroll
      for (int k = 0; k < 8; ++k) { sv[k] += __shfl_xor(sv[k], 8, 64); sv[k] = xor16_sum_(sv[k]); sv[k] = xor32_sum_(sv[k]); }
      if (rr == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { red[(wave * 64 + nt * 32 + cq * 4 + j) * 2] = sv[j]; red[(wave * 64 + nt * 32 + cq * 4 + j) * 2 + 1] = sv[4 + j]; }
      }
    }
  }
  // A PERSISTENT kernel passes vmax_defer and raises the scalar once, after its tile loop: amax_raise_ starts with a plain load whose
  // result is needed at once -- s_waitcnt vmcnt(0) in the middle of the loop, which also drains the next tile's prefetch and this
  // tile's stores (found in the ISA of gemm_k64_wstat_kernel, round 5)
  if (vmax_defer) *vmax_defer = fmaxf(*vmax_defer, vmax);
  else if (g.amax_out) {                             // (wave-uniform pointer test; one guarded atomic per wave)
    vmax = wave_max(vmax);
    if (lane == 0) amax_raise_(g.amax_out, vmax);
  }
  if (rowstats) {
    // (mean, rstd) over the 64 channels of every RESULT row -- the statistics the next LayerNorm(64) needs (se_row_stats on Y):
    // two passes over the values still in registers, exactly like row_stats64_kernel
    float* __restrict__ So = g.AUX + 2 * ptile;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = wave * 32 + rr + 8 * i;
      float sm = 0.f;
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) sm += (kept[nt][i].x + kept[nt][i].y) + (kept[nt][i].z + kept[nt][i].w);
      sm += __shfl_xor(sm, 1, 64); sm += __shfl_xor(sm, 2, 64); sm += __shfl_xor(sm, 4, 64);
      const float mean = sm * (1.f / 64.f);
      float sq = 0.f;
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const float a0 = kept[nt][i].x - mean, a1 = kept[nt][i].y - mean, a2 = kept[nt][i].z - mean, a3 = kept[nt][i].w - mean;
        sq += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
      }
      sq += __shfl_xor(sq, 1, 64); sq += __shfl_xor(sq, 2, 64); sq += __shfl_xor(sq, 4, 64);
      if (cq == 0 && m0 + row < Mb) *reinterpret_cast<float2*>(So + 2 * row) = make_float2(mean, rsqrtf(sq * (1.f / 64.f) + 1e-5f));
    }
  }
  if (ep & SE_EPI_STATS) {
    __syncthreads();
    const int tid = threadIdx.x;
    if (tid < 64) {
      float s_ = 0.f, q_ = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) { s_ += red[(w * 64 + tid) * 2]; q_ += red[(w * 64 + tid) * 2 + 1]; }
      int nn = by * 64 + tid;
      if (nn < d.N) {
        atomicAdd(&g.stats[((long)b * d.N + nn) * 2], (double)s_);
        atomicAdd(&g.stats[((long)b * d.N + nn) * 2 + 1], (double)q_);
      }
    }
  }
}
// the vector epilogue's bias operands go to LDS before the K loop (its barriers order the hand-off): a global load
// at the tail of the workgroup would expose one full memory latency per tile
static __device__ __forceinline__ void stage_bias(const GemmArgs& g, int by, float* bias_s) {
  if (threadIdx.x < 64) {
    const int n = by * 64 + threadIdx.x;
    bias_s[threadIdx.x] = ((g.d.epilogue & SE_EPI_BIAS) && n < g.d.N) ? g.bias[n] : 0.f;
  }
}
// LayerNorm-backward epilogue of a row GEMM with N == 64 (se_gemm_ln_bwd): the tile holds whole rows, so the product
// dL = A W^T never goes to memory -- dX = dR + rstd * (dxh - mean(dxh) - xh * mean(dxh * xh)), dxh = dL * gamma, is computed
// on the accumulators (a row's 64 channels sit in the 8 lanes cq = 0..7 of one rr group after the wave-private transpose) and
// the gamma / beta gradients are folded over the rows of the workgroup (one atomic per channel and workgroup).
// Operands: X = g.AUX [M][64], (mean, rstd) = g.rowstats, gamma = g.ps, dR = g.R, dX = g.Y; red: [4 waves][64][2] floats.
constexpr int SE_EPI_LN_BWD_ = 1024;        // internal epilogue flag (not part of the public mask)
// operands of the LayerNorm-backward epilogue, requested BEFORE the K loop (a global load at the tail of a workgroup exposes one
// full memory latency per tile: the first version of this epilogue waited 58 % of its wave cycles)
struct LnBwdPre { float4 x[2][4]; float4 r[2][4]; float2 mr[4]; };
static __device__ __forceinline__ void ln_bwd_prefetch(const GemmArgs& g, int m0, LnBwdPre& p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int Mb = g.d.To * g.d.Fo, cq = lane & 7, rr = lane >> 3;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const long rg = (long)m0 + wave * 32 + rr + 8 * i;
    const bool ok = rg < Mb;
    p.mr[i] = ok ? *reinterpret_cast<const float2*>(g.rowstats + 2 * rg) : make_float2(0.f, 0.f);
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const long off = rg * 64 + nt * 32 + cq * 4;
      p.x[nt][i] = ok ? *reinterpret_cast<const float4*>(g.AUX + off) : make_float4(0.f, 0.f, 0.f, 0.f);
      p.r[nt][i] = (ok && g.R) ? *reinterpret_cast<const float4*>(g.R + off) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
}
static __device__ __forceinline__ void gemm_epilogue_ln_bwd(const GemmArgs& g, const f32x16& acc0, const f32x16& acc1, int m0,
                                                            float* cs, int cs_ld, float* red, const LnBwdPre& pre) {
  const se_gemm_desc& d = g.d;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int Mb = d.To * d.Fo;
  const int col = lane & 31, half = lane >> 5, cq = lane & 7, rr = lane >> 3;
  float4 gv[2][4];                            // dL of rows rr + 8 i, columns nt * 32 + 4 cq .. + 3
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const f32x16& acc = nt ? acc1 : acc0;
#pragma unroll
    for (int r = 0; r < 16; ++r) cs[((r & 3) + 8 * (r >> 2) + 4 * half) * cs_ld + col] = acc[r];
#pragma unroll
    for (int i = 0; i < 4; ++i) gv[nt][i] = *reinterpret_cast<const float4*>(&cs[(rr + 8 * i) * cs_ld + cq * 4]);
  }
  float ag[2][4] = {}, ab[2][4] = {}, xmax = 0.f;
  float4 gm[2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) gm[nt] = *reinterpret_cast<const float4*>(g.ps + nt * 32 + cq * 4);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const long rg = (long)m0 + wave * 32 + rr + 8 * i;
    const bool ok = rg < Mb;
    const float mean = pre.mr[i].x, rstd = pre.mr[i].y;
    float xh[2][4], dxh[2][4], s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const float4 xv = pre.x[nt][i];
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
    if (ok) {
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const long off = rg * 64 + nt * 32 + cq * 4;
        const float4 r1 = pre.r[nt][i];
        float o4[4] = {r1.x, r1.y, r1.z, r1.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) o4[j] += rstd * (dxh[nt][j] - s1 - xh[nt][j] * s2);
        *reinterpret_cast<float4*>(g.Y + off) = make_float4(o4[0], o4[1], o4[2], o4[3]);
        xmax = fmaxf(fmaxf(xmax, fmaxf(fabsf(o4[0]), fabsf(o4[1]))), fmaxf(fabsf(o4[2]), fabsf(o4[3])));
      }
    }
  }
  if (g.amax_out) {
    xmax = wave_max(xmax);
    if (lane == 0) amax_raise_(g.amax_out, xmax);
  }
  // gamma / beta gradients: fold the 8 row groups of the wave (lane bits 3..5), then the 4 waves through LDS
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float sg = ag[nt][j], sb = ab[nt][j];
#pragma unroll
      for (int o = 8; o < 64; o <<= 1) { sg += __shfl_xor(sg, o, 64); sb += __shfl_xor(sb, o, 64); }
      if (rr == 0) { red[(wave * 64 + nt * 32 + cq * 4 + j) * 2] = sg; red[(wave * 64 + nt * 32 + cq * 4 + j) * 2 + 1] = sb; }
    }
  __syncthreads();
  if (tid < 64) {
    float sg = 0.f, sb = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) { sg += red[(w * 64 + tid) * 2]; sb += red[(w * 64 + tid) * 2 + 1]; }
    atomicAdd(&g.dgamma[tid], sg);
    atomicAdd(&g.dbeta[tid], sb);
  }
}

// GLU flavour of the vectorised epilogue: accumulator 0 = value columns, accumulator 1 = gate columns of the same 32
// outputs.  The gate tile is transposed first and parked in registers, then the value tile; Y = a * sigmoid(g) and
// the pre-GLU Z (both halves) leave as float4 stores.
// bias_lds: optional [value 32 | gate 32] bias of this column block staged in LDS (no global load at the tail of the block)
static __device__ __forceinline__ void gemm_epilogue_glu_vec(const GemmArgs& g, const f32x16& acc0, const f32x16& acc1,
                                                             int m0, int by, int b, float* cs, int cs_ld, const float* bias_lds = nullptr) {
  const se_gemm_desc& d = g.d;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int Mb = d.To * d.Fo, No = d.N / 2;
  const long ptile = (long)b * Mb + m0;
  float* __restrict__ Yb = g.Y + ptile * d.ldc + d.c_off;
  float* __restrict__ Zb = g.AUX ? g.AUX + ptile * d.ldx + d.x_off : nullptr;
  const int col = lane & 31, half = lane >> 5, cq = lane & 7, rr = lane >> 3;
  const int n = by * 32 + cq * 4;                  // value column; gate column = No + n
  float4 gate[4];
#pragma unroll
  for (int r = 0; r < 16; ++r) cs[((r & 3) + 8 * (r >> 2) + 4 * half) * cs_ld + col] = acc1[r];
#pragma unroll
  for (int i = 0; i < 4; ++i) gate[i] = *reinterpret_cast<const float4*>(&cs[(rr + 8 * i) * cs_ld + cq * 4]);
#pragma unroll
  for (int r = 0; r < 16; ++r) cs[((r & 3) + 8 * (r >> 2) + 4 * half) * cs_ld + col] = acc0[r];
  if (n >= No) return;
  float4 ba = make_float4(0.f, 0.f, 0.f, 0.f), bg = ba;
  if (bias_lds) { ba = *reinterpret_cast<const float4*>(bias_lds + cq * 4); bg = *reinterpret_cast<const float4*>(bias_lds + 32 + cq * 4); }
  else if (d.epilogue & SE_EPI_BIAS) { ba = *reinterpret_cast<const float4*>(g.bias + n); bg = *reinterpret_cast<const float4*>(g.bias + No + n); }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = wave * 32 + rr + 8 * i;
    if (m0 + row >= Mb) continue;
    float4 a = *reinterpret_cast<const float4*>(&cs[(rr + 8 * i) * cs_ld + cq * 4]);
    float4 gt = gate[i];
    a.x += ba.x; a.y += ba.y; a.z += ba.z; a.w += ba.w;
    gt.x += bg.x; gt.y += bg.y; gt.z += bg.z; gt.w += bg.w;
    if (Zb) {
      if (d.epilogue & SE_EPI_GLU_GATE) {      // the backward needs only (a sigmoid(g), g): the gate half alone, [M][No]
        *reinterpret_cast<float4*>(Zb + ((unsigned)row * (unsigned)d.ldx + (unsigned)n)) = gt;
      } else {
        *reinterpret_cast<float4*>(Zb + ((unsigned)row * (unsigned)d.ldx + (unsigned)n)) = a;
        *reinterpret_cast<float4*>(Zb + ((unsigned)row * (unsigned)d.ldx + (unsigned)(No + n))) = gt;
      }
    }
    *reinterpret_cast<float4*>(Yb + ((unsigned)row * (unsigned)d.ldc + (unsigned)n)) =
        make_float4(a.x * sigmoidf_(gt.x), a.y * sigmoidf_(gt.y), a.z * sigmoidf_(gt.z), a.w * sigmoidf_(gt.w));
  }
}
static __device__ __forceinline__ bool epilogue_glu_vec_ok(const se_gemm_desc& d) {
  return (d.epilogue & SE_EPI_GLU) && !(d.epilogue & (SE_EPI_STATS | SE_EPI_SHUFFLE2 | SE_EPI_DROP | SE_EPI_RESID | SE_EPI_ACCUM |
                                                      SE_EPI_SWISH_GRAD | 256)) &&
         (d.N & 7) == 0 && (d.ldc & 3) == 0 && (d.c_off & 3) == 0 && (d.ldx & 3) == 0 && (d.x_off & 3) == 0;
}
static __device__ __forceinline__ bool epilogue_vec_ok(const se_gemm_desc& d) {
  return !(d.epilogue & (SE_EPI_GLU | SE_EPI_SHUFFLE2 | 256)) && (d.N & 3) == 0 && (d.ldc & 3) == 0 &&
         (d.c_off & 3) == 0 && (d.ldx & 3) == 0 && (d.x_off & 3) == 0 && (d.ldr & 3) == 0 && (d.r_off & 3) == 0;
}

// epilogue shared by the fp32 and the split-bf16 kernels: acc0 / acc1 = the wave's two 32x32 accumulators
static __device__ __forceinline__ void gemm_epilogue(const GemmArgs& g, const f32x16& acc0, const f32x16& acc1, int m0,
                                                     int by, int b, float* red, unsigned thr, float inv_keep) {
  const se_gemm_desc& d = g.d;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int Mb = d.To * d.Fo;
  const bool glu = (d.epilogue & SE_EPI_GLU) != 0;
  // ------------------------------ epilogue ------------------------------
  const int ep = d.epilogue;
  const int col = lane & 31, half = lane >> 5;
  int n0, n1;           // original output-channel index of the two accumulators' column
  bool nok0, nok1;
  if (glu) {
    n0 = by * 32 + col; n1 = d.N / 2 + n0;
    nok0 = nok1 = n0 < d.N / 2;
  } else {
    n0 = by * 64 + col; n1 = n0 + 32;
    nok0 = n0 < d.N; nok1 = n1 < d.N;
  }
  float bias0 = 0.f, bias1 = 0.f;
  if (ep & SE_EPI_BIAS) {
    if (nok0) bias0 = g.bias[n0];
    if (nok1) bias1 = g.bias[n1];
  }
  float s0 = 0.f, q0 = 0.f, s1 = 0.f, q1 = 0.f;
  const int No = d.N / 2;   // SHUFFLE2 / GLU output channels
  // wave-uniform tile bases; lane offsets are 32-bit (a 128-row tile spans < 2^31 elements)
  const long ptile = (long)b * Mb + m0;
  float* __restrict__ Yb = g.Y + ptile * d.ldc + d.c_off;
  float* __restrict__ Xb = g.AUX ? g.AUX + ptile * d.ldx + d.x_off : nullptr;
  const float* __restrict__ Rb = g.R ? g.R + ptile * d.ldr + d.r_off : nullptr;
  const unsigned pdrop = (unsigned)ptile;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
    const int m = m0 + row;
    if (m >= Mb) continue;
    float v0 = acc0[r] + bias0, v1 = acc1[r] + bias1;
    if (ep & SE_EPI_STATS) { if (nok0) { s0 += v0; q0 += v0 * v0; } if (nok1) { s1 += v1; q1 += v1 * v1; } }
    const unsigned yo = (unsigned)row * (unsigned)d.ldc;
    if (glu) {
      if (nok0) {
        if (Xb) {
          const unsigned xo = (unsigned)row * (unsigned)d.ldx;
          if (ep & SE_EPI_GLU_GATE) Xb[xo + n0] = v1;
          else { Xb[xo + n0] = v0; Xb[xo + n1] = v1; }
        }
        Yb[yo + n0] = v0 * sigmoidf_(v1);
      }
      continue;
    }
    if (ep & SE_EPI_DROP) {       // dropout of the (bias-added) result, or of the hidden activation whose
                                   // gradient this is (with SWISH_GRAD): mask index = output element
      const unsigned pe = (pdrop + (unsigned)row) * (unsigned)d.N;
      v0 *= drop_scale(d.epi_seed, pe + n0, thr, inv_keep);
      v1 *= drop_scale(d.epi_seed, pe + n1, thr, inv_keep);
    }
    if (ep & SE_EPI_SWISH_GRAD) {
      const unsigned xo = (unsigned)row * (unsigned)d.ldx;
      if (nok0) v0 *= swish_gradf_(Xb[xo + n0]);
      if (nok1) v1 *= swish_gradf_(Xb[xo + n1]);
    }
    if (ep & SE_EPI_RESID) {
      const unsigned ro = (unsigned)row * (unsigned)d.ldr;
      if (nok0) v0 = Rb[ro + n0] + d.alpha * v0;
      if (nok1) v1 = Rb[ro + n1] + d.alpha * v1;
    }
    if (ep & SE_EPI_SHUFFLE2) {
      int t = m / d.Fo, f = m - t * d.Fo;
      float* __restrict__ Ys = g.Y + ((long)b * d.To * 2 * d.Fo) * d.ldc + d.c_off;
      if (nok0) { unsigned po = (unsigned)(t * 2 * d.Fo + 2 * f + (n0 >= No));
                  Ys[po * (unsigned)d.ldc + (n0 >= No ? n0 - No : n0)] = v0; }
      if (nok1) { unsigned po = (unsigned)(t * 2 * d.Fo + 2 * f + (n1 >= No));
                  Ys[po * (unsigned)d.ldc + (n1 >= No ? n1 - No : n1)] = v1; }
      continue;
    }
    if (ep & 256) { if (v0 == 12345.678f && v1 == 0.1234f) Yb[yo + n0] = v0; continue; }      // ablation: no stores
    if (ep & SE_EPI_ACCUM) { if (nok0) Yb[yo + n0] += v0; if (nok1) Yb[yo + n1] += v1; }
    else { if (nok0) Yb[yo + n0] = v0; if (nok1) Yb[yo + n1] = v1; }
  }
  if (ep & SE_EPI_STATS) {
    // rows live in registers (16 per lane) and in the two lane halves: fold halves, then waves via LDS
    s0 += __shfl_xor(s0, 32, 64); q0 += __shfl_xor(q0, 32, 64);
    s1 += __shfl_xor(s1, 32, 64); q1 += __shfl_xor(q1, 32, 64);
    if (half == 0) {
      red[(wave * 64 + col) * 2] = s0; red[(wave * 64 + col) * 2 + 1] = q0;
      red[(wave * 64 + 32 + col) * 2] = s1; red[(wave * 64 + 32 + col) * 2 + 1] = q1;
    }
    __syncthreads();
    if (tid < 64) {
      float s = 0.f, q = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) { s += red[(w * 64 + tid) * 2]; q += red[(w * 64 + tid) * 2 + 1]; }
      int n = by * 64 + tid;
      if (n < d.N) {
        int ns = (ep & SE_EPI_SHUFFLE2) ? (n >= No ? n - No : n) : n;
        int Ns = (ep & SE_EPI_SHUFFLE2) ? No : d.N;
        atomicAdd(&g.stats[((long)b * Ns + ns) * 2], (double)s);
        atomicAdd(&g.stats[((long)b * Ns + ns) * 2 + 1], (double)q);
      }
    }
  }
}

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

// NPL = 2: x = hi + lo (16 mantissa bits), products hh + hl + lh;  NPL = 3: x = hi + mid + lo (all 24 bits of an fp32:
// the split is exact), products hh + hm + mh + hl + lh + mm, dropped terms <= 2^-24 relative -> fp32-equivalent.
typedef float f32x2_ __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_ __attribute__((ext_vector_type(2)));
typedef unsigned u32x2_ __attribute__((ext_vector_type(2)));
static __device__ __forceinline__ unsigned pk_bf16_(float a, float b) {          // one v_cvt_pk_bf16_f32 (round to nearest even)
  f32x2_ v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_));
}
// Register-resident variants of the same packed split: x[] (8 or 4 floats) -> NPL planes of bf16x8 / bf16x4; x[] is consumed.
typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
template <int NPL, typename V8>
static __device__ __forceinline__ void split_planes8(float (&x)[8], V8 (&out)[NPL]) {
#pragma unroll
  for (int q = 0; q < NPL; ++q) {
    u32x4_ w;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      w[i] = pk_bf16_(x[2 * i], x[2 * i + 1]);
      if (q + 1 < NPL) {
        x[2 * i] -= __builtin_bit_cast(float, w[i] << 16);
        x[2 * i + 1] -= __builtin_bit_cast(float, w[i] & 0xffff0000u);
      }
    }
    out[q] = __builtin_bit_cast(V8, w);
  }
}
template <int NPL>
static __device__ __forceinline__ void split_store4(float (&e)[4], __bf16* p, int plane_stride) {
#pragma unroll
  for (int q = 0; q < NPL; ++q) {
    const unsigned a = pk_bf16_(e[0], e[1]), b = pk_bf16_(e[2], e[3]);
    *reinterpret_cast<u32x2_*>(p + q * plane_stride) = (u32x2_){a, b};
    if (q + 1 < NPL) {
      e[0] -= __builtin_bit_cast(float, a << 16); e[1] -= __builtin_bit_cast(float, a & 0xffff0000u);
      e[2] -= __builtin_bit_cast(float, b << 16); e[3] -= __builtin_bit_cast(float, b & 0xffff0000u);
    }
  }
}
// Pairs are converted with one v_cvt_pk_bf16_f32, the residuals taken against the shifted / masked halves and subtracted with
// v_pk_add_f32: 18 VALU instructions for the 3-way split of 4 values (the per-element form compiled to 31).  Same roundings.
template <int NPL>
static __device__ __forceinline__ void split_store(float4 v, __bf16* p, int plane_stride) {
  float x0 = v.x, x1 = v.y, x2 = v.z, x3 = v.w;
#pragma unroll
  for (int q = 0; q < NPL; ++q) {
    const unsigned a = pk_bf16_(x0, x1), b = pk_bf16_(x2, x3);
    *reinterpret_cast<u32x2_*>(p + q * plane_stride) = (u32x2_){a, b};
    if (q + 1 < NPL) {
      x0 -= __builtin_bit_cast(float, a << 16); x1 -= __builtin_bit_cast(float, a & 0xffff0000u);
      x2 -= __builtin_bit_cast(float, b << 16); x3 -= __builtin_bit_cast(float, b & 0xffff0000u);
    }
  }
}


// ---- precision 3: scaled split-fp16 (hi + lo, three products) --------------------------------------------------------------
// y = x * 2^sexp (exact) with sexp chosen so that the operand's largest |y| lies in [2^13, 2^14); hi = fp16(y) (round to nearest
// even, 11-bit significand), lo = fp16(y - hi): |lo| <= 2^-12 |y| and its own rounding error <= 2^-12 |lo|, so hi + lo
// represents y to 2^-24 |y| (the fp32 rounding unit) wherever lo is a NORMAL fp16 (|y| >= 2^-3, i.e. within 2^17 of the
// maximum); below that lo is subnormal with quantum 2^-24: absolute error <= 2^-25 = 2^-39 of the maximum.  hi*hi, hi*lo, lo*hi
// are exact in fp32 (22-bit products); lo*lo (<= 2^-24 relative) is dropped.  FP16_OVFL clamps an out-of-range conversion to
// +-65504 instead of producing inf (amax pointers make that unreachable; a static exponent is a promise by the caller).
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2_ __attribute__((ext_vector_type(2)));
static __device__ __forceinline__ unsigned pk_f16_(float a, float b) {            // one v_cvt_pk_f16_f32 (round to nearest even)
  f32x2_ v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2_));
}
static __device__ __forceinline__ void f16_clamp_mode_() { asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1"); }
static __host__ __device__ __forceinline__ int f16_sexp_(float amax) {             // amax * 2^sexp in [2^13, 2^14)
  unsigned u;
  __builtin_memcpy(&u, &amax, 4);
  const int e = (int)((u >> 23) & 0xffu) - 127;
  const int sx = 13 - e;
  return !(amax > 0.f) ? 0 : (sx < -60 ? -60 : (sx > 60 ? 60 : sx));
}
static __device__ __forceinline__ float exp2i_(int e) { return __builtin_bit_cast(float, (unsigned)(127 + e) << 23); }
static __device__ __forceinline__ int operand_sexp_(const float* amax, int sexp) {
  return amax ? f16_sexp_(__builtin_nontemporal_load(amax)) : sexp;
}
static __device__ __forceinline__ void split_store_h(float4 v, float s, __bf16* p, int plane_stride) {
#ifdef SE_ABL_NOSPLIT      // timing ablation (fp16-accurate results only): HALF of the split's instructions (no lo plane: 6 of 12 per 4 values)
  *reinterpret_cast<u32x2_*>(p) = (u32x2_){pk_f16_(v.x * s, v.y * s), pk_f16_(v.z * s, v.w * s)};
  *reinterpret_cast<u32x2_*>(p + plane_stride) = (u32x2_){0u, 0u};
  return;
#endif
  float x0 = v.x * s, x1 = v.y * s, x2 = v.z * s, x3 = v.w * s;
  const unsigned a = pk_f16_(x0, x1), b = pk_f16_(x2, x3);
  *reinterpret_cast<u32x2_*>(p) = (u32x2_){a, b};
  const f16x2_ ah = __builtin_bit_cast(f16x2_, a), bh = __builtin_bit_cast(f16x2_, b);
  x0 -= (float)ah[0]; x1 -= (float)ah[1]; x2 -= (float)bh[0]; x3 -= (float)bh[1];
  *reinterpret_cast<u32x2_*>(p + plane_stride) = (u32x2_){pk_f16_(x0, x1), pk_f16_(x2, x3)};
}
template <typename V8>
static __device__ __forceinline__ void split_planes8_h(float (&x)[8], float s, V8 (&out)[2]) {
  // (scalar words, assembled into the vectors at the end: with `u32x4_ h; h[i] = ...; bit_cast(h[i])` in the loop, hipcc 7.2
  // selected v_fma_mix_f32 with the FIRST word as the fp16 source of all eight residuals -- tools/micro/f16chk.hip)
  unsigned hw[4], lw[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float y0 = x[2 * i] * s, y1 = x[2 * i + 1] * s;
    const unsigned w = pk_f16_(y0, y1);
    const f16x2_ hh = __builtin_bit_cast(f16x2_, w);
    hw[i] = w;
    lw[i] = pk_f16_(y0 - (float)hh[0], y1 - (float)hh[1]);
  }
  out[0] = __builtin_bit_cast(V8, (u32x4_){hw[0], hw[1], hw[2], hw[3]});
  out[1] = __builtin_bit_cast(V8, (u32x4_){lw[0], lw[1], lw[2], lw[3]});
}
// running element-wise maximum of |x| over packed fp16 pairs: m (one pair) vs the 8 values of a fragment
static __device__ __forceinline__ unsigned pk_absmax_f16_(unsigned m, const bf16x8& frag) {
  const u32x4_ w = __builtin_bit_cast(u32x4_, frag);
  f16x2_ r = __builtin_bit_cast(f16x2_, m);
#pragma unroll
  for (int i = 0; i < 4; ++i) r = __builtin_elementwise_max(r, __builtin_bit_cast(f16x2_, w[i] & 0x7fff7fffu));
  return __builtin_bit_cast(unsigned, r);
}
// one 32x32x16 / 16x16x32 MFMA on 16-bit fragments held as bf16x8 bit patterns: bf16 or (F16) fp16 arithmetic
template <bool F16>
static __device__ __forceinline__ f32x16 mfma32_(const bf16x8& a, const bf16x8& b, const f32x16& c) {
  if constexpr (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
template <bool F16>
static __device__ __forceinline__ f32x4 mfma16_(const bf16x8& a, const bf16x8& b, const f32x4& c) {
  if constexpr (F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
// NPL-plane split of a float4 into LDS in either arithmetic (s: the fp16 operand scale, ignored for bf16)
template <int NPL, bool F16>
static __device__ __forceinline__ void split_store_x(float4 v, float s, __bf16* p, int plane_stride) {
  if constexpr (F16) { static_assert(NPL == 2, "split-fp16 has two planes"); split_store_h(v, s, p, plane_stride); }
  else split_store<NPL>(v, p, plane_stride);
}

// ---------------------------------------------------------------------------------------------
static int check_desc(const se_gemm_desc* d) {
  SE_REQUIRE(d->ntap >= 1 && d->ntap <= SE_MAX_TAPS, "gemm: ntap %d out of range", d->ntap);
  SE_REQUIRE(d->C > 0 && (d->C % 4) == 0, "gemm: C=%d must be a positive multiple of 4", d->C);
  SE_REQUIRE((d->lda % 4) == 0 && (d->a_off % 4) == 0, "gemm: lda/a_off must be multiples of 4");
  SE_REQUIRE((d->ldw % 4) == 0 && d->ldw >= d->ntap * d->C, "gemm: ldw=%d too small / unaligned", d->ldw);
  SE_REQUIRE(d->B > 0 && d->To > 0 && d->Fo > 0 && d->Ti > 0 && d->Fi > 0, "gemm: empty grid");
  SE_REQUIRE(d->st >= 1 && d->sf >= 1, "gemm: bad strides");
  SE_REQUIRE(d->N > 0, "gemm: N=%d", d->N);
  if (d->prologue == SE_PRO_LN) SE_REQUIRE(d->ntap == 1, "gemm: LN prologue needs ntap==1");
  SE_REQUIRE(d->drop_p >= 0.f && d->drop_p < 1.f, "gemm: drop_p=%f out of range", d->drop_p);
  // lane offsets are 32-bit element indices relative to a per-batch-entry base (64-bit): one batch entry of every operand
  // must stay below 2^32 elements (B = 64 x [321, 201, 256] is 16.5 M elements per entry: fine; guarded, not assumed)
  {
    const long in_e = (long)d->Ti * d->Fi * d->lda, out_px = (long)d->To * d->Fo * ((d->epilogue & SE_EPI_SHUFFLE2) ? 2 : 1);
    const long widest = d->ldc > d->ldr ? (d->ldc > d->ldx ? d->ldc : d->ldx) : (d->ldr > d->ldx ? d->ldr : d->ldx);
    SE_REQUIRE(in_e < 4294967296L && out_px * widest < 4294967296L,
               "gemm: a batch entry of an operand exceeds 2^32 elements (32-bit lane offsets): split the launch");
  }
  if (d->prologue == SE_PRO_SWISH_DROP || d->prologue == SE_PRO_DROP || (d->epilogue & SE_EPI_DROP))
    SE_REQUIRE((long)d->B * d->Ti * d->Fi * (d->C > d->N ? d->C : d->N) < 4294967296L, "gemm: dropout index exceeds 32 bits");
  return 0;
}

