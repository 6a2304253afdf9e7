// fp32-exact MFMA "tap GEMM" family for gfx950 (v_mfma_f32_32x32x2_f32).
//
//   Y[m][n] = epi( sum_tap sum_c pro(A[src(m,tap)][c]) * W[n][tap*C + c] )
//
// One kernel body serves nn.Linear, 1x1 / (2,3)-dilated / (1,3) / 4x4-strided convolutions (implicit
// GEMM: the A tile is gathered per tap straight from the channels-last feature map, nothing is
// im2col'ed), their input gradients (same kernel, transposed packed weights, negated taps) and the
// DFT / inverse DFT of the STFT front-end (frames are rows with lda = hop).
// A second kernel computes weight gradients (reduction over rows, fp32 atomics across row chunks).
//
// Tile: 128 rows x 64 cols per 256-thread workgroup (4 waves, each 32 rows x 64 cols = two 32x32
// accumulators), K staged through LDS in BK-float slabs; both LDS tiles are row-major with a +4
// pad (conflict-free ds_read_b128: 16-lane groups hit 16 distinct 4-bank slots, stride 36 or 20).
// K order inside a slab is permuted so that each lane reads ONE contiguous BK/2 run
// (lane half h supplies k in [h*BK/2, (h+1)*BK/2)) -> b128 LDS reads feed 4 MFMAs each.
#include "se_common.h"
#include <stdlib.h>

struct GemmArgs {
  se_gemm_desc d;
  const float* A; const float* W; const float* bias; float* Y; const float* R; float* AUX;
  const float* rowstats; const float* ps; const float* pb; double* stats;
  int ncb;      // column blocks per row tile
  int tiles;    // row tiles per batch entry
  int nouter;   // B * tiles
  int contig;   // 1: every XCD sweeps a contiguous range of row tiles (tap convolutions: the dt-shifted rows of a
                //    tile are the dt = 0 rows of a tile the same L2 has just seen); 0: round-robin
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
template <bool HASPRE = false>
static __device__ __forceinline__ void gemm_epilogue_vec(const GemmArgs& g, const f32x16& acc0, const f32x16& acc1,
                                                         int m0, int by, int b, float* cs, int cs_ld, unsigned thr,
                                                         float inv_keep, float* red, const float* bias_s,
                                                         const float4 (&pre)[8] = {}) {   // pre[nt*4+i]: AUX / R values fetched early
  const se_gemm_desc& d = g.d;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int Mb = d.To * d.Fo, ep = d.epilogue;
  const long ptile = (long)b * Mb + m0;
  float* __restrict__ Yb = g.Y + ptile * d.ldc + d.c_off;
  const float* __restrict__ Xb = g.AUX ? g.AUX + ptile * d.ldx + d.x_off : nullptr;
  const float* __restrict__ Rb = g.R ? g.R + ptile * d.ldr + d.r_off : nullptr;
  const unsigned pdrop = (unsigned)ptile;
  const int col = lane & 31, half = lane >> 5;
  const int cq = lane & 7, rr = lane >> 3;          // read-back role: float4 column, row within an 8-row pass
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const f32x16& acc = nt ? acc1 : acc0;
#pragma unroll
    for (int r = 0; r < 16; ++r) cs[((r & 3) + 8 * (r >> 2) + 4 * half) * cs_ld + col] = acc[r];
    const int n = by * 64 + nt * 32 + cq * 4;        // first of this lane's 4 output columns
    float4 ssum = make_float4(0.f, 0.f, 0.f, 0.f), qsum = ssum;
    if (n < d.N) {                                   // N % 4 == 0 (host-checked)
      const float4 bias4 = *reinterpret_cast<const float4*>(bias_s + nt * 32 + cq * 4);   // staged before the K loop
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
          float4 rv = HASPRE ? pre[nt * 4 + i] : *reinterpret_cast<const float4*>(Rb + ((unsigned)row * (unsigned)d.ldr + (unsigned)n));
          v.x = rv.x + d.alpha * v.x; v.y = rv.y + d.alpha * v.y; v.z = rv.z + d.alpha * v.z; v.w = rv.w + d.alpha * v.w;
        }
        float4* yp = reinterpret_cast<float4*>(Yb + ((unsigned)row * (unsigned)d.ldc + (unsigned)n));
        if (ep & SE_EPI_ACCUM) { float4 o = *yp; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
        *yp = v;
      }
    }
    if (ep & SE_EPI_STATS) {      // fold the 8 row-lanes that share this column group, park per-wave partials in LDS
      float sv[8] = {ssum.x, ssum.y, ssum.z, ssum.w, qsum.x, qsum.y, qsum.z, qsum.w};
#pragma unroll
      for (int k = 0; k < 8; ++k) { sv[k] += __shfl_xor(sv[k], 8, 64); sv[k] += __shfl_xor(sv[k], 16, 64); sv[k] += __shfl_xor(sv[k], 32, 64); }
      if (rr == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { red[(wave * 64 + nt * 32 + cq * 4 + j) * 2] = sv[j]; red[(wave * 64 + nt * 32 + cq * 4 + j) * 2 + 1] = sv[4 + j]; }
      }
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
// GLU flavour of the vectorised epilogue: accumulator 0 = value columns, accumulator 1 = gate columns of the same 32
// outputs.  The gate tile is transposed first and parked in registers, then the value tile; Y = a * sigmoid(g) and
// the pre-GLU Z (both halves) leave as float4 stores.
static __device__ __forceinline__ void gemm_epilogue_glu_vec(const GemmArgs& g, const f32x16& acc0, const f32x16& acc1,
                                                             int m0, int by, int b, float* cs, int cs_ld) {
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
  if (d.epilogue & SE_EPI_BIAS) { ba = *reinterpret_cast<const float4*>(g.bias + n); bg = *reinterpret_cast<const float4*>(g.bias + No + n); }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = wave * 32 + rr + 8 * i;
    if (m0 + row >= Mb) continue;
    float4 a = *reinterpret_cast<const float4*>(&cs[(rr + 8 * i) * cs_ld + cq * 4]);
    float4 gt = gate[i];
    a.x += ba.x; a.y += ba.y; a.z += ba.z; a.w += ba.w;
    gt.x += bg.x; gt.y += bg.y; gt.z += bg.z; gt.w += bg.w;
    if (Zb) {
      *reinterpret_cast<float4*>(Zb + ((unsigned)row * (unsigned)d.ldx + (unsigned)n)) = a;
      *reinterpret_cast<float4*>(Zb + ((unsigned)row * (unsigned)d.ldx + (unsigned)(No + n))) = gt;
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
        if (Xb) { const unsigned xo = (unsigned)row * (unsigned)d.ldx; Xb[xo + n0] = v0; Xb[xo + n1] = v1; }
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

template <int BK, int PRO, bool LIN>
__global__ __launch_bounds__(256) void gemm_tap_kernel(GemmArgs g) {
  constexpr int BM = 128, BN = 64, SA = BK + 4;
  constexpr int KQ = BK / 4;        // float4 per tile row
  constexpr int RPP = 256 / KQ;     // tile rows covered per staging pass
  constexpr int NA = BM / RPP, NB = BN / RPP;
  __shared__ __attribute__((aligned(16))) float As[BM * SA];
  __shared__ __attribute__((aligned(16))) float Bs[BN * SA];
  __shared__ float red[4 * 64 * 2];
  __shared__ __attribute__((aligned(16))) float bias_s[64];

  const se_gemm_desc& d = g.d;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const WorkId wk_ = decode_work(g.ncb, g.nouter, g.contig);
  if (wk_.outer >= g.nouter) return;
  const int b = wk_.outer / g.tiles, by = wk_.inner;
  const int Mb = d.To * d.Fo;
  const int m0 = (wk_.outer - b * g.tiles) * BM;
  const bool glu = (d.epilogue & SE_EPI_GLU) != 0;
  const int kq = tid % KQ, r0 = tid / KQ;
  // Addressing: per-batch-entry bases are wave-uniform (SGPR pairs) and every lane offset is a 32-bit element
  // index, so global accesses use the saddr + 32-bit voffset form and no 64-bit vector multiplies are issued.
  // `lin`: one tap, unit strides, identical in/out grids (nn.Linear, 1x1 conv): source pixel == row index.
  constexpr bool lin = LIN;        // host-checked; compile-time so that the tap / grid logic vanishes from the row GEMMs
  const int TiFi = d.Ti * d.Fi;
  const float* __restrict__ Ab = g.A + (long)b * TiFi * d.lda + d.a_off;
  const float* __restrict__ Wb = g.W;

  int rt[NA], rf[NA];
  bool rok[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    int m = m0 + r0 + i * RPP;
    rok[i] = m < Mb;
    if (lin) { rt[i] = m; rf[i] = 0; }
    else { rt[i] = m / d.Fo; rf[i] = m - rt[i] * d.Fo; }
  }
  // W rows of this column block (GLU pairs value column j with gate column N/2 + j)
  unsigned wrow[NB];
  bool wok[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    int j = r0 + i * RPP;
    int n;
    if (glu) { n = (j >> 5) * (d.N / 2) + by * 32 + (j & 31); wok[i] = (by * 32 + (j & 31)) < d.N / 2; }
    else { n = by * 64 + j; wok[i] = n < d.N; }
    wrow[i] = (unsigned)n * (unsigned)d.ldw;
  }
  float ln_mean[NA] = {}, ln_rstd[NA] = {};
  if (PRO == SE_PRO_LN) {
    const float* __restrict__ rs = g.rowstats + (long)b * TiFi * 2;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      int p = rok[i] ? (lin ? rt[i] : src_pixel_in(d, rt[i], rf[i], 0)) : -1;
      float2 mr = p >= 0 ? *reinterpret_cast<const float2*>(rs + 2 * p) : make_float2(0.f, 0.f);
      ln_mean[i] = mr.x;
      ln_rstd[i] = mr.y;
    }
  }

  const int nchunk = (d.C + BK - 1) / BK;
  const int NI = d.ntap * nchunk;
  float4 ra[NA], rb[NB];
  bool aok[NA];
  unsigned apix[NA];
  int cur_c = 0;
  float4 ps4 = make_float4(0.f, 0.f, 0.f, 0.f), pb4 = ps4;
  const unsigned thr = drop_thr(d.drop_p);
  const float inv_keep = drop_inv_keep(d.drop_p);

  auto load_tiles = [&](int it) {
    int chunk = lin ? it : it / d.ntap;  // channel chunk outer, tap inner: the taps of one chunk re-touch the same
    int tap = lin ? 0 : it - chunk * d.ntap;   // 128-B lines (df = -1, 0, 1) while they are still in L1 / L2
    int c0 = chunk * BK;
    int c = c0 + kq * 4;
    cur_c = c;
    bool cok = c < d.C;   // C is a multiple of 4
    const unsigned wk = (unsigned)(tap * d.C + c);
    load_pro_vec<PRO>(g.ps, g.pb, c, cok, ps4, pb4);
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      int p = (rok[i] && cok) ? (lin ? rt[i] : src_pixel_in(d, rt[i], rf[i], tap)) : -1;
      aok[i] = p >= 0;
      apix[i] = (unsigned)(b * TiFi + p);
      ra[i] = aok[i] ? *reinterpret_cast<const float4*>(Ab + ((unsigned)p * (unsigned)d.lda + (unsigned)c))
                     : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      rb[i] = (wok[i] && cok) ? *reinterpret_cast<const float4*>(Wb + (wrow[i] + wk))
                              : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };

  f32x16 acc0, acc1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
  const bool vec_ep = epilogue_vec_ok(d);
  if (vec_ep) stage_bias(g, by, bias_s);

  load_tiles(0);
  const float* Ap = &As[(wave * 32 + (lane & 31)) * SA + (lane >> 5) * (BK / 2)];
  const float* Bp0 = &Bs[(lane & 31) * SA + (lane >> 5) * (BK / 2)];
  const float* Bp1 = Bp0 + 32 * SA;

  for (int it = 0; it < NI; ++it) {
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      float4 v = ra[i];
      if (PRO != SE_PRO_NONE && aok[i])
        v = apply_pro<PRO>(v, cur_c, d.C, ln_mean[i], ln_rstd[i], ps4, pb4, apix[i], d.pro_seed, thr, inv_keep);
      *reinterpret_cast<float4*>(&As[(r0 + i * RPP) * SA + kq * 4]) = v;
    }
#pragma unroll
    for (int i = 0; i < NB; ++i)
      *reinterpret_cast<float4*>(&Bs[(r0 + i * RPP) * SA + kq * 4]) = rb[i];
    __syncthreads();
    if (it + 1 < NI) load_tiles(it + 1);
#pragma unroll
    for (int s4 = 0; s4 < BK / 2; s4 += 4) {
      float4 a = *reinterpret_cast<const float4*>(Ap + s4);
      float4 b0 = *reinterpret_cast<const float4*>(Bp0 + s4);
      float4 b1 = *reinterpret_cast<const float4*>(Bp1 + s4);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b0.x, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b1.x, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b0.y, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b1.y, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b0.z, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b1.z, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b0.w, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b1.w, acc1, 0, 0, 0);
    }
    __syncthreads();
  }

  if (SA >= 36 && vec_ep) gemm_epilogue_vec(g, acc0, acc1, m0, by, b, &As[wave * 32 * SA], SA, thr, inv_keep, red, bias_s);
  else if (SA >= 36 && epilogue_glu_vec_ok(d)) gemm_epilogue_glu_vec(g, acc0, acc1, m0, by, b, &As[wave * 32 * SA], SA);
  else gemm_epilogue(g, acc0, acc1, m0, by, b, red, thr, inv_keep);
}

// ---------------------------------------------------------------------------------------------
// Split-bf16 ("bf16x3") variant for the MFMA-bound layers (the dilated dense convolutions, K = 384..1536):
// every fp32 operand is split on the fly into hi = bf16(x), lo = bf16(x - hi) and the product is evaluated as
// a_hi b_hi + a_hi b_lo + a_lo b_hi on v_mfma_f32_32x32x16_bf16 with fp32 accumulation.  Each operand is then
// represented to 2^-16 relative, i.e. products carry ~1.5e-5 relative error (vs 6e-8 in fp32, 4e-3 in plain bf16)
// -- inside the 1e-3 parity budget with two orders of magnitude to spare (measured in tests/test_gemm_gpu.py and
// on the full model) -- while the three bf16 MFMAs cost 3/16 of the one fp32 MFMA they replace.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

// NPL = 2: x = hi + lo (16 mantissa bits), products hh + hl + lh;  NPL = 3: x = hi + mid + lo (all 24 bits of an fp32:
// the split is exact), products hh + hm + mh + hl + lh + mm, dropped terms <= 2^-24 relative -> fp32-equivalent.
template <int NPL>
static __device__ __forceinline__ void split_store(float4 v, __bf16* p, int plane_stride) {
  float x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int q = 0; q < NPL; ++q) {
    bf16x4 h;
#pragma unroll
    for (int j = 0; j < 4; ++j) { h[j] = (__bf16)x[j]; x[j] -= (float)h[j]; }
    *reinterpret_cast<bf16x4*>(p + q * plane_stride) = h;
  }
}

template <int PRO, int NPL, bool LIN>
__global__ __launch_bounds__(256) void gemm_tap_bf16x3_kernel(GemmArgs g) {
  constexpr int BM = 128, BN = 64, BK = 32, SA = 40;     // rows of 32 bf16 + 8 pad = 80 B: conflict-free b128 reads
  constexpr int KQ = BK / 4, RPP = 256 / KQ, NA = BM / RPP, NB = BN / RPP;
  constexpr int PA = BM * SA, PB = BN * SA;               // plane strides
  __shared__ __attribute__((aligned(16))) __bf16 Ap[NPL * PA];
  __shared__ __attribute__((aligned(16))) __bf16 Bp[NPL * PB];
  __shared__ float red[4 * 64 * 2];
  __shared__ __attribute__((aligned(16))) float bias_s[64];

  const se_gemm_desc& d = g.d;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const WorkId wk_ = decode_work(g.ncb, g.nouter, g.contig);
  if (wk_.outer >= g.nouter) return;
  const int b = wk_.outer / g.tiles, by = wk_.inner;
  const int Mb = d.To * d.Fo;
  const int m0 = (wk_.outer - b * g.tiles) * BM;
  const bool glu = (d.epilogue & SE_EPI_GLU) != 0;
  const int kq = tid % KQ, r0 = tid / KQ;
  constexpr bool lin = LIN;        // host-checked; compile-time so that the tap / grid logic vanishes from the row GEMMs
  const int TiFi = d.Ti * d.Fi;
  const float* __restrict__ Ab = g.A + (long)b * TiFi * d.lda + d.a_off;
  const float* __restrict__ Wb = g.W;

  int rt[NA], rf[NA];
  bool rok[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    int m = m0 + r0 + i * RPP;
    rok[i] = m < Mb;
    if (lin) { rt[i] = m; rf[i] = 0; }
    else { rt[i] = m / d.Fo; rf[i] = m - rt[i] * d.Fo; }
  }
  unsigned wrow[NB];
  bool wok[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    int j = r0 + i * RPP;
    int n;
    if (glu) { n = (j >> 5) * (d.N / 2) + by * 32 + (j & 31); wok[i] = (by * 32 + (j & 31)) < d.N / 2; }
    else { n = by * 64 + j; wok[i] = n < d.N; }
    wrow[i] = (unsigned)n * (unsigned)d.ldw;
  }
  float ln_mean[NA] = {}, ln_rstd[NA] = {};
  if (PRO == SE_PRO_LN) {
    const float* __restrict__ rs = g.rowstats + (long)b * TiFi * 2;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      int p = rok[i] ? (lin ? rt[i] : src_pixel_in(d, rt[i], rf[i], 0)) : -1;
      float2 mr = p >= 0 ? *reinterpret_cast<const float2*>(rs + 2 * p) : make_float2(0.f, 0.f);
      ln_mean[i] = mr.x;
      ln_rstd[i] = mr.y;
    }
  }
  const int nchunk = (d.C + BK - 1) / BK;
  const int NI = d.ntap * nchunk;
  float4 ra[NA], rb[NB];
  bool aok[NA];
  unsigned apix[NA];
  int cur_c = 0;
  float4 ps4 = make_float4(0.f, 0.f, 0.f, 0.f), pb4 = ps4;
  const unsigned thr = drop_thr(d.drop_p);
  const float inv_keep = drop_inv_keep(d.drop_p);

  auto load_tiles = [&](int it) {
    int chunk = lin ? it : it / d.ntap;  // channel chunk outer, tap inner: the taps of one chunk re-touch the same
    int tap = lin ? 0 : it - chunk * d.ntap;   // 128-B lines (df = -1, 0, 1) while they are still in L1 / L2
    int c0 = chunk * BK;
    int c = c0 + kq * 4;
    cur_c = c;
    bool cok = c < d.C;
    const unsigned wk = (unsigned)(tap * d.C + c);
    load_pro_vec<PRO>(g.ps, g.pb, c, cok, ps4, pb4);
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      int p = (rok[i] && cok) ? (lin ? rt[i] : src_pixel_in(d, rt[i], rf[i], tap)) : -1;
      aok[i] = p >= 0;
      apix[i] = (unsigned)(b * TiFi + p);
      ra[i] = aok[i] ? *reinterpret_cast<const float4*>(Ab + ((unsigned)p * (unsigned)d.lda + (unsigned)c))
                     : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < NB; ++i)
      rb[i] = (wok[i] && cok) ? *reinterpret_cast<const float4*>(Wb + (wrow[i] + wk)) : make_float4(0.f, 0.f, 0.f, 0.f);
  };

  f32x16 acc0, acc1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
  const bool vec_ep = epilogue_vec_ok(d);
  if (vec_ep) stage_bias(g, by, bias_s);
  load_tiles(0);
  // operand fragments: lane (r = lane & 31, h = lane >> 5) holds k = 16 ks + 8 h .. + 7 of row r (16 contiguous bytes)
  const int frag = (lane & 31) * SA + 8 * (lane >> 5);
  for (int it = 0; it < NI; ++it) {
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      float4 v = ra[i];
      if (PRO != SE_PRO_NONE && aok[i])
        v = apply_pro<PRO>(v, cur_c, d.C, ln_mean[i], ln_rstd[i], ps4, pb4, apix[i], d.pro_seed, thr, inv_keep);
      split_store<NPL>(v, &Ap[(r0 + i * RPP) * SA + kq * 4], PA);
    }
#pragma unroll
    for (int i = 0; i < NB; ++i)
      split_store<NPL>(rb[i], &Bp[(r0 + i * RPP) * SA + kq * 4], PB);
    __syncthreads();
    if (it + 1 < NI) load_tiles(it + 1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int ao = wave * 32 * SA + frag + 16 * ks, bo = frag + 16 * ks;
      bf16x8 af[NPL], bf0[NPL], bf1[NPL];
#pragma unroll
      for (int q = 0; q < NPL; ++q) {
        af[q] = *reinterpret_cast<const bf16x8*>(&Ap[q * PA + ao]);
        bf0[q] = *reinterpret_cast<const bf16x8*>(&Bp[q * PB + bo]);
        bf1[q] = *reinterpret_cast<const bf16x8*>(&Bp[q * PB + 32 * SA + bo]);
      }
      // all part products with order (qa + qb) < NPL, smallest terms first
#pragma unroll
      for (int ord = NPL - 1; ord >= 0; --ord)
#pragma unroll
        for (int qa = 0; qa <= ord; ++qa) {
          const int qb = ord - qa;
          acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[qa], bf0[qb], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[qa], bf1[qb], acc1, 0, 0, 0);
        }
    }
    __syncthreads();
  }
  float* cs = reinterpret_cast<float*>(Ap) + wave * 32 * 36;        // the staging planes are free now
  if (vec_ep) gemm_epilogue_vec(g, acc0, acc1, m0, by, b, cs, 36, thr, inv_keep, red, bias_s);
  else gemm_epilogue(g, acc0, acc1, m0, by, b, red, thr, inv_keep);
}

// ---------------------------------------------------------------------------------------------
// Split-bf16 kernel for the unit-stride convolutions whose taps come in frequency triples (dt, {-1, 0, +1}): the
// dilated dense layers (2 x 3 taps), the sub-pixel convolutions (1 x 3) and their input gradients -- 90 % of the
// convolution FLOPs.  The three taps of a triple read the SAME input rows shifted by one pixel, so the A tile is staged
// once per (channel chunk, dt) as a 130-row halo tile (flattened pixels m0 - 1 .. m0 + 128 of the dt-shifted row
// range) and the three taps read their MFMA fragments from it at row offsets 0 / 1 / 2: a third of the global loads,
// bf16 splits and LDS writes of the generic kernel (which is VALU-bound by exactly those).  The frequency padding
// (f = 0 with df = -1, f = F - 1 with df = +1) wraps to the neighbouring time row in flattened order and is masked per
// lane on the fragment instead.
template <int NPL>
__global__ __launch_bounds__(256) void conv3_bf16_kernel(GemmArgs g) {
  constexpr int BM = 128, BN = 64, BK = 32, SA = 40, HR = BM + 2;
  constexpr int PA = HR * SA, PB = BN * SA;
  __shared__ __attribute__((aligned(16))) __bf16 Ap[NPL * PA];
  __shared__ __attribute__((aligned(16))) __bf16 Bp[NPL * PB];
  __shared__ float red[4 * 64 * 2];
  __shared__ __attribute__((aligned(16))) float bias_s[64];
  const se_gemm_desc& d = g.d;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const WorkId wk_ = decode_work(g.ncb, g.nouter, g.contig);
  if (wk_.outer >= g.nouter) return;
  const int b = wk_.outer / g.tiles, by = wk_.inner;
  const int Mb = d.To * d.Fo;                 // == Ti * Fi (host-checked)
  const int m0 = (wk_.outer - b * g.tiles) * BM;
  const int kq = tid & 7, r0 = tid >> 3;      // float4 column of the 32-channel chunk, row within a 32-row pass
  const float* __restrict__ Ab = g.A + (long)b * Mb * d.lda + d.a_off;
  const float* __restrict__ Wb = g.W;
  unsigned wrow[2];
  bool wok[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) { int n = by * 64 + r0 + 32 * i; wok[i] = n < d.N; wrow[i] = (unsigned)n * (unsigned)d.ldw; }
  // frequency-edge masks of this lane's output pixel (fragment row lane & 31 of the wave's 32 rows)
  const int fpix = (m0 + wave * 32 + (lane & 31)) % d.Fo;
  const bool edgeL = fpix == 0, edgeR = fpix == d.Fo - 1;
  const int nchunk = (d.C + BK - 1) / BK, ngrp = d.ntap / 3;
  const int NI = nchunk * d.ntap;
  const unsigned thr = 0u;
  const float inv_keep = 1.f;

  float4 ra[4], rh, rb[2];
  auto load_a = [&](int grp_it) {             // halo tile of (chunk, triple) number grp_it
    const int chunk = grp_it / ngrp, gi = grp_it - chunk * ngrp;
    const int c = chunk * BK + kq * 4;
    const bool cok = c < d.C;
    const int q0 = m0 - 1 + d.dt[3 * gi] * d.Fo;          // flattened source pixel of halo row 0
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int q = q0 + 1 + r0 + 32 * i;
      ra[i] = (cok && q >= 0 && q < Mb) ? *reinterpret_cast<const float4*>(Ab + ((unsigned)q * (unsigned)d.lda + (unsigned)c))
                                        : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (tid < 16) {                            // halo rows 0 and 129
      const int q = q0 + (tid >> 3) * (HR - 1);
      rh = (cok && q >= 0 && q < Mb) ? *reinterpret_cast<const float4*>(Ab + ((unsigned)q * (unsigned)d.lda + (unsigned)c))
                                     : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto load_b = [&](int it) {
    const int chunk = it / d.ntap, tap = it - chunk * d.ntap;
    const int c = chunk * BK + kq * 4;
    const bool cok = c < d.C;
    const unsigned wk = (unsigned)(tap * d.C + c);
#pragma unroll
    for (int i = 0; i < 2; ++i)
      rb[i] = (wok[i] && cok) ? *reinterpret_cast<const float4*>(Wb + (wrow[i] + wk)) : make_float4(0.f, 0.f, 0.f, 0.f);
  };

  f32x16 acc0, acc1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
  const bool vec_ep = epilogue_vec_ok(d);
  if (vec_ep) stage_bias(g, by, bias_s);
  load_a(0);
  load_b(0);
  const int frag = (lane & 31) * SA + 8 * (lane >> 5);
  int it = 0, gi = 0;
  for (int gq = 0; gq < nchunk * ngrp; ++gq, gi = (gi + 1 == ngrp ? 0 : gi + 1)) {
    // stage the halo tile of this (chunk, triple); the previous iteration's trailing barrier freed Ap
#pragma unroll
    for (int i = 0; i < 4; ++i) split_store<NPL>(ra[i], &Ap[(1 + r0 + 32 * i) * SA + kq * 4], PA);
    if (tid < 16) split_store<NPL>(rh, &Ap[((tid >> 3) * (HR - 1)) * SA + kq * 4], PA);
#pragma unroll 1
    for (int s3 = 0; s3 < 3; ++s3, ++it) {
#pragma unroll
      for (int i = 0; i < 2; ++i) split_store<NPL>(rb[i], &Bp[(r0 + 32 * i) * SA + kq * 4], PB);
      __syncthreads();
      if (it + 1 < NI) load_b(it + 1);
      if (s3 == 0 && gq + 1 < nchunk * ngrp) load_a(gq + 1);
      const int df = d.df[3 * gi + s3];
      const bool kill = (df < 0 && edgeL) || (df > 0 && edgeR);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int ao = (wave * 32 + 1 + df) * SA + frag + 16 * ks, bo = frag + 16 * ks;
        bf16x8 af[NPL], bf0[NPL], bf1[NPL];
#pragma unroll
        for (int q = 0; q < NPL; ++q) {
          af[q] = *reinterpret_cast<const bf16x8*>(&Ap[q * PA + ao]);
          bf0[q] = *reinterpret_cast<const bf16x8*>(&Bp[q * PB + bo]);
          bf1[q] = *reinterpret_cast<const bf16x8*>(&Bp[q * PB + 32 * SA + bo]);
        }
        if (df != 0) {
#pragma unroll
          for (int q = 0; q < NPL; ++q) {
            f32x4 z = kill ? (f32x4){0.f, 0.f, 0.f, 0.f} : *reinterpret_cast<f32x4*>(&af[q]);
            af[q] = *reinterpret_cast<bf16x8*>(&z);
          }
        }
#pragma unroll
        for (int ord = NPL - 1; ord >= 0; --ord)
#pragma unroll
          for (int qa = 0; qa <= ord; ++qa) {
            const int qb = ord - qa;
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[qa], bf0[qb], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[qa], bf1[qb], acc1, 0, 0, 0);
          }
      }
      __syncthreads();
    }
  }
  float* cs = reinterpret_cast<float*>(Ap) + wave * 32 * 36;
  if (vec_ep) gemm_epilogue_vec(g, acc0, acc1, m0, by, b, cs, 36, thr, inv_keep, red, bias_s);
  else gemm_epilogue(g, acc0, acc1, m0, by, b, red, thr, inv_keep);
}

// ---------------------------------------------------------------------------------------------
// Row-panel kernel for the token-wise layers with K = 64 and N >= 128 (LN -> 256 / 192 / GLU-256, dY(64) -> 256), split
// bf16.  The per-column-block kernel above is issue-bound on these shapes: every one of the N/64 sibling workgroups
// re-loads, re-normalises and re-splits the same A rows and pays the same ~700 VALU instructions of set-up per wave for
// 64 MFMAs.  Here one workgroup owns 128 rows and sweeps ALL column blocks: each wave loads its 32 rows straight into
// the MFMA A-fragment layout (lane = row, 8 consecutive k), applies the prologue and the bf16 split ONCE and keeps the
// fragments in 16 * NPL VGPRs; only the 64 x 64 weight blocks stream through LDS (next block prefetched in registers).
template <int PRO, int NPL, bool PRE2>
__global__ __launch_bounds__(256) void gemm_k64_panel_kernel(GemmArgs g) {
  constexpr int SB = 72, PB = 64 * SB;         // 64 + 8 bf16 per W row: 144-B stride, conflict-free b128 fragment reads
  __shared__ __attribute__((aligned(16))) __bf16 Bp[NPL * PB];
  __shared__ __attribute__((aligned(16))) float patch[4 * 32 * 36];
  __shared__ float red[4 * 64 * 2];
  __shared__ __attribute__((aligned(16))) float bias_s[64];
  const se_gemm_desc& d = g.d;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int Mb = d.To * d.Fo;                  // row GEMM: B == 1
  const int m0 = blockIdx.x * 128;
  const bool glu = (d.epilogue & SE_EPI_GLU) != 0;
  const unsigned thr = drop_thr(d.drop_p);
  const float inv_keep = drop_inv_keep(d.drop_p);

  // ---- A fragments: row = lane & 31 of this wave's 32 rows, k = 16 ks + 8 (lane >> 5) .. + 7
  const int row = m0 + wave * 32 + (lane & 31), kg = lane >> 5;
  const bool rok = row < Mb;
  bf16x8 af[4][NPL];
  {
    const float* __restrict__ ap = g.A + (long)row * d.lda + d.a_off + 8 * kg;
    float mean = 0.f, rstd = 0.f;
    if (PRO == SE_PRO_LN && rok) { float2 mr = *reinterpret_cast<const float2*>(g.rowstats + 2 * (long)row); mean = mr.x; rstd = mr.y; }
    float4 v[4][2];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      v[ks][0] = rok ? *reinterpret_cast<const float4*>(ap + 16 * ks) : make_float4(0.f, 0.f, 0.f, 0.f);
      v[ks][1] = rok ? *reinterpret_cast<const float4*>(ap + 16 * ks + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      float x[8];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int c = 16 * ks + 8 * kg + 4 * h;
        float4 w = v[ks][h];
        if (PRO != SE_PRO_NONE && rok) {
          float4 ps4 = make_float4(0.f, 0.f, 0.f, 0.f), pb4 = ps4;
          load_pro_vec<PRO>(g.ps, g.pb, c, true, ps4, pb4);
          w = apply_pro<PRO>(w, c, 64, mean, rstd, ps4, pb4, (unsigned)row, d.pro_seed, thr, inv_keep);
        }
        x[4 * h] = w.x; x[4 * h + 1] = w.y; x[4 * h + 2] = w.z; x[4 * h + 3] = w.w;
      }
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl) {
        bf16x8 hh;
#pragma unroll
        for (int e = 0; e < 8; ++e) { hh[e] = (__bf16)x[e]; x[e] -= (float)hh[e]; }
        af[ks][pl] = hh;
      }
    }
  }
  // ---- W blocks: 64 rows x 64 k, 4 float4 per thread
  const int kq = tid & 15, r0 = tid >> 4;
  const int ncb = g.ncb;
  float4 rb[4];
  auto load_w = [&](int by) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int j = r0 + 16 * i;
      int n; bool ok;
      if (glu) { n = (j >> 5) * (d.N / 2) + by * 32 + (j & 31); ok = (by * 32 + (j & 31)) < d.N / 2; }
      else { n = by * 64 + j; ok = n < d.N; }
      rb[i] = ok ? *reinterpret_cast<const float4*>(g.W + (unsigned)n * (unsigned)d.ldw + 4 * kq) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  load_w(0);
  const bool vec_ep = epilogue_vec_ok(d);
  const int frag = (lane & 31) * SB + 8 * (lane >> 5);
  float* cs = patch + wave * 32 * 36;
  for (int by = 0; by < ncb; ++by) {
#pragma unroll
    for (int i = 0; i < 4; ++i) split_store<NPL>(rb[i], &Bp[(r0 + 16 * i) * SB + kq * 4], PB);
    if (vec_ep) stage_bias(g, by, bias_s);
    __syncthreads();
    if (by + 1 < ncb) load_w(by + 1);
    // second epilogue operand (pre-activation for the swish gradient, or the residual) of this column block: issued
    // before the MFMAs -- with 2 waves per SIMD nothing else would cover its latency at the tail
    float4 pre[8];
    if (PRE2) {
      const float* __restrict__ src = (d.epilogue & SE_EPI_SWISH_GRAD) ? g.AUX + (long)m0 * d.ldx + d.x_off : g.R + (long)m0 * d.ldr + d.r_off;
      const unsigned ld2 = (d.epilogue & SE_EPI_SWISH_GRAD) ? d.ldx : d.ldr;
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int prow = wave * 32 + (lane >> 3) + 8 * i, pn = by * 64 + nt * 32 + (lane & 7) * 4;
          pre[nt * 4 + i] = (m0 + prow < Mb && pn < d.N) ? *reinterpret_cast<const float4*>(src + ((unsigned)prow * ld2 + (unsigned)pn))
                                                         : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      bf16x8 bf0[NPL], bf1[NPL];
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl) {
        bf0[pl] = *reinterpret_cast<const bf16x8*>(&Bp[pl * PB + frag + 16 * ks]);
        bf1[pl] = *reinterpret_cast<const bf16x8*>(&Bp[pl * PB + 32 * SB + frag + 16 * ks]);
      }
#pragma unroll
      for (int ord = NPL - 1; ord >= 0; --ord)
#pragma unroll
        for (int qa = 0; qa <= ord; ++qa) {
          acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks][qa], bf0[ord - qa], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks][qa], bf1[ord - qa], acc1, 0, 0, 0);
        }
    }
    if (vec_ep) gemm_epilogue_vec<PRE2>(g, acc0, acc1, m0, by, 0, cs, 36, thr, inv_keep, red, bias_s, pre);
    else gemm_epilogue_glu_vec(g, acc0, acc1, m0, by, 0, cs, 36);
    __syncthreads();
  }
}

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
};

template <int NPL>
__global__ __launch_bounds__(256, 2) void ff_fwd_kernel(FfArgs a) {    // 2 workgroups per CU: VGPR + AGPR <= 256
  constexpr int SB = 72, PB = 64 * SB, SP = 36;
  __shared__ __attribute__((aligned(16))) __bf16 W1p[NPL * PB];
  __shared__ __attribute__((aligned(16))) __bf16 W2p[NPL * PB];
  __shared__ __attribute__((aligned(16))) float patch[4 * 32 * SP];    // wave-private 32 x 32 transposes
  __shared__ __attribute__((aligned(16))) float b1s[64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float* cs = patch + wave * 32 * SP;
  const long m0 = (long)blockIdx.x * 128;
  const long row = m0 + wave * 32 + (lane & 31);
  const int kg = lane >> 5;
  const bool rok = row < a.M;
  const unsigned thr = drop_thr(a.drop_p);
  const float inv_keep = drop_inv_keep(a.drop_p);
  const bool dr = a.drop_p > 0.f;

  bf16x8 af1[4][NPL];
  {
    const float* __restrict__ xp = a.X + row * 64 + 8 * kg;
    float mean = 0.f, rstd = 0.f;
    if (rok) { float2 mr = *reinterpret_cast<const float2*>(a.rowstats + 2 * row); mean = mr.x; rstd = mr.y; }
    float4 v[4][2];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      v[ks][0] = rok ? *reinterpret_cast<const float4*>(xp + 16 * ks) : make_float4(0.f, 0.f, 0.f, 0.f);
      v[ks][1] = rok ? *reinterpret_cast<const float4*>(xp + 16 * ks + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      float x[8];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int c = 16 * ks + 8 * kg + 4 * h;
        const float4 gm = *reinterpret_cast<const float4*>(a.gamma + c), bt = *reinterpret_cast<const float4*>(a.beta + c);
        const float4 w = v[ks][h];
        x[4 * h] = rok ? (w.x - mean) * rstd * gm.x + bt.x : 0.f;
        x[4 * h + 1] = rok ? (w.y - mean) * rstd * gm.y + bt.y : 0.f;
        x[4 * h + 2] = rok ? (w.z - mean) * rstd * gm.z + bt.z : 0.f;
        x[4 * h + 3] = rok ? (w.w - mean) * rstd * gm.w + bt.w : 0.f;
      }
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl) {
        bf16x8 hh;
#pragma unroll
        for (int e = 0; e < 8; ++e) { hh[e] = (__bf16)x[e]; x[e] -= (float)hh[e]; }
        af1[ks][pl] = hh;
      }
    }
  }
  const int kq = tid & 15, r0 = tid >> 4;
  const int nb = a.hid / 64;
  float4 rw1[4], rw2[4];
  auto load_w = [&](int jb) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int j = r0 + 16 * i;
      rw1[i] = *reinterpret_cast<const float4*>(a.W1 + (unsigned)(jb * 64 + j) * 64u + 4 * kq);
      rw2[i] = *reinterpret_cast<const float4*>(a.W2 + (unsigned)j * (unsigned)a.hid + jb * 64 + 4 * kq);
    }
  };
  load_w(0);
  f32x16 y0, y1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { y0[r] = 0.f; y1[r] = 0.f; }
  const int frag = (lane & 31) * SB + 8 * kg;
  const int col = lane & 31, half = lane >> 5, cq = lane & 7, rr = lane >> 3;
  for (int jb = 0; jb < nb; ++jb) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      split_store<NPL>(rw1[i], &W1p[(r0 + 16 * i) * SB + kq * 4], PB);
      split_store<NPL>(rw2[i], &W2p[(r0 + 16 * i) * SB + kq * 4], PB);
    }
    if (tid < 64) b1s[tid] = a.b1[jb * 64 + tid];
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
          acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af1[ks][qa], bf0[ord - qa], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af1[ks][qa], bf1[ord - qa], acc1, 0, 0, 0);
        }
    }
    const float bb0 = b1s[col], bb1 = b1s[32 + col];
    // per 32-column half of the block: transpose through the wave's patch, write H, re-split, second GEMM
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rl = (r & 3) + 8 * (r >> 2) + 4 * half;
        cs[rl * SP + col] = nt ? acc1[r] + bb1 : acc0[r] + bb0;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int rl = rr + 8 * i;
        const long rg = m0 + wave * 32 + rl;
        if (rg < a.M)
          *reinterpret_cast<float4*>(a.H + rg * a.hid + jb * 64 + nt * 32 + cq * 4) =
              *reinterpret_cast<const float4*>(&cs[rl * SP + cq * 4]);
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
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) {
          bf16x8 hh;
#pragma unroll
          for (int e = 0; e < 8; ++e) { hh[e] = (__bf16)x[e]; x[e] -= (float)hh[e]; }
          af2[pl] = hh;
          bf0[pl] = *reinterpret_cast<const bf16x8*>(&W2p[pl * PB + frag + 16 * ks]);
          bf1[pl] = *reinterpret_cast<const bf16x8*>(&W2p[pl * PB + 32 * SB + frag + 16 * ks]);
        }
#pragma unroll
        for (int ord = NPL - 1; ord >= 0; --ord)
#pragma unroll
          for (int qa = 0; qa <= ord; ++qa) {
            y0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af2[qa], bf0[ord - qa], y0, 0, 0, 0);
            y1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af2[qa], bf1[ord - qa], y1, 0, 0, 0);
          }
      }
    }
    __syncthreads();
  }
  // Y = X + alpha * Drop_o(acc + b2)
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int rl = (r & 3) + 8 * (r >> 2) + 4 * half;
      cs[rl * SP + col] = nt ? y1[r] : y0[r];
    }
    const int n = nt * 32 + cq * 4;
    const float4 b2v = *reinterpret_cast<const float4*>(a.b2 + n);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int rl = rr + 8 * i;
      const long rg = m0 + wave * 32 + rl;
      if (rg >= a.M) continue;
      float4 v = *reinterpret_cast<const float4*>(&cs[rl * SP + cq * 4]);
      v.x += b2v.x; v.y += b2v.y; v.z += b2v.z; v.w += b2v.w;
      if (dr) {
        const float4 d4 = drop_scale4(a.seed_o, (unsigned)(rg * 64 + n), thr, inv_keep);
        v.x *= d4.x; v.y *= d4.y; v.z *= d4.z; v.w *= d4.w;
      }
      const float4 xr = *reinterpret_cast<const float4*>(a.X + rg * 64 + n);
      *reinterpret_cast<float4*>(a.Y + rg * 64 + n) =
          make_float4(xr.x + a.alpha * v.x, xr.y + a.alpha * v.y, xr.z + a.alpha * v.z, xr.w + a.alpha * v.w);
    }
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
};

template <int NPL>
__global__ __launch_bounds__(256, 2) void ff_bwd_kernel(FfBwdArgs a) {
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

  bf16x8 af1[4][NPL];
  {
    const float* __restrict__ yp = a.dY + row * 64 + 8 * kg;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      float x[8];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int c = 16 * ks + 8 * kg + 4 * h;
        float4 w = rok ? *reinterpret_cast<const float4*>(yp + 16 * ks + 4 * h) : make_float4(0.f, 0.f, 0.f, 0.f);
        if (dr && rok) {
          const float4 d4 = drop_scale4(a.seed_o, (unsigned)(row * 64 + c), thr, inv_keep);
          w.x *= d4.x; w.y *= d4.y; w.z *= d4.z; w.w *= d4.w;
        }
        x[4 * h] = w.x; x[4 * h + 1] = w.y; x[4 * h + 2] = w.z; x[4 * h + 3] = w.w;
      }
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl) {
        bf16x8 hh;
#pragma unroll
        for (int e = 0; e < 8; ++e) { hh[e] = (__bf16)x[e]; x[e] -= (float)hh[e]; }
        af1[ks][pl] = hh;
      }
    }
  }
  const int kq = tid & 15, r0 = tid >> 4;
  const int nb = a.hid / 64;
  f32x16 g0, g1;                              // dLN accumulators (32 rows x 64 channels)
#pragma unroll
  for (int r = 0; r < 16; ++r) { g0[r] = 0.f; g1[r] = 0.f; }
  const int frag = (lane & 31) * SB + 8 * kg;
  const int col = lane & 31, half = lane >> 5, cq = lane & 7, rr = lane >> 3;
  for (int jb = 0; jb < nb; ++jb) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int j = r0 + 16 * i;
      const float4 wa = *reinterpret_cast<const float4*>(a.W2T + (unsigned)(jb * 64 + j) * 64u + 4 * kq);
      const float4 wb = *reinterpret_cast<const float4*>(a.W1T + (unsigned)j * (unsigned)a.hid + jb * 64 + 4 * kq);
      split_store<NPL>(wa, &Wa[j * SB + kq * 4], PB);
      split_store<NPL>(wb, &Wb[j * SB + kq * 4], PB);
    }
    // pre-activations of this block for the Swish gradient: issued before the MFMAs, consumed in the epilogue
    float4 hp[8];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const long rg = m0 + wave * 32 + rr + 8 * i;
        hp[nt * 4 + i] = rg < a.M ? *reinterpret_cast<const float4*>(a.H + rg * a.hid + jb * 64 + nt * 32 + cq * 4)
                                  : make_float4(0.f, 0.f, 0.f, 0.f);
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
          acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af1[ks][qa], bf0[ord - qa], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af1[ks][qa], bf1[ord - qa], acc1, 0, 0, 0);
        }
    }
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rl = (r & 3) + 8 * (r >> 2) + 4 * half;
        cs[rl * SP + col] = nt ? acc1[r] : acc0[r];
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
        else *reinterpret_cast<float4*>(a.dZ + rg * a.hid + jb * 64 + nt * 32 + cq * 4) = v;
        *reinterpret_cast<float4*>(&cs[rl * SP + cq * 4]) = v;
      }
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
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) {
          bf16x8 hh;
#pragma unroll
          for (int e = 0; e < 8; ++e) { hh[e] = (__bf16)x[e]; x[e] -= (float)hh[e]; }
          af2[pl] = hh;
          bf0[pl] = *reinterpret_cast<const bf16x8*>(&Wb[pl * PB + frag + 16 * ks]);
          bf1[pl] = *reinterpret_cast<const bf16x8*>(&Wb[pl * PB + 32 * SB + frag + 16 * ks]);
        }
#pragma unroll
        for (int ord = NPL - 1; ord >= 0; --ord)
#pragma unroll
          for (int qa = 0; qa <= ord; ++qa) {
            g0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af2[qa], bf0[ord - qa], g0, 0, 0, 0);
            g1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af2[qa], bf1[ord - qa], g1, 0, 0, 0);
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
      cs[rl * SP + col] = nt ? g1[r] : g0[r];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) gv[nt][i] = *reinterpret_cast<const float4*>(&cs[(rr + 8 * i) * SP + cq * 4]);
  }
  if (a.X == nullptr) {
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const long rg = m0 + wave * 32 + rr + 8 * i;
        if (rg < a.M) *reinterpret_cast<float4*>(a.dLN + rg * 64 + nt * 32 + cq * 4) = gv[nt][i];
      }
    return;
  }
  // LayerNorm backward on the rows in registers: a row's 64 channels sit in the 8 lanes cq = 0..7 of one rr group
  float ag[2][4] = {}, ab[2][4] = {};
  float4 gm[2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) gm[nt] = *reinterpret_cast<const float4*>(a.gamma + nt * 32 + cq * 4);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const long rg = m0 + wave * 32 + rr + 8 * i;
    const bool ok = rg < a.M;
    float mean = 0.f, rstd = 0.f;
    if (ok) { const float2 mr = *reinterpret_cast<const float2*>(a.stats + 2 * rg); mean = mr.x; rstd = mr.y; }
    float xh[2][4], dxh[2][4], s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const float4 xv = ok ? *reinterpret_cast<const float4*>(a.X + rg * 64 + nt * 32 + cq * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
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
        const float4 r1 = *reinterpret_cast<const float4*>(a.dY + off);
        float o4[4] = {r1.x, r1.y, r1.z, r1.w};
        if (a.dR2) { const float4 r2 = *reinterpret_cast<const float4*>(a.dR2 + off); o4[0] += r2.x; o4[1] += r2.y; o4[2] += r2.z; o4[3] += r2.w; }
#pragma unroll
        for (int j = 0; j < 4; ++j) o4[j] += rstd * (dxh[nt][j] - s1 - xh[nt][j] * s2);
        *reinterpret_cast<float4*>(a.dX + off) = make_float4(o4[0], o4[1], o4[2], o4[3]);
      }
    }
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

extern "C" int se_ff_bwd_dgrad(const float* dY, const float* H, const float* W2T, const float* W1T, float* dZ, float* dLN,
                               long M, int hid, float drop_p, unsigned seed_h, unsigned seed_o, int precision,
                               const float* X, const float* stats, const float* gamma, const float* dR2, float* dX,
                               float* dgamma, float* dbeta, void* stream) {
  SE_REQUIRE(dY && H && W2T && W1T && dZ, "ff_bwd_dgrad: null operand");
  SE_REQUIRE(X ? (stats && gamma && dX && dgamma && dbeta) : dLN != nullptr,
             "ff_bwd_dgrad: either dLN, or all of X / stats / gamma / dX / dgamma / dbeta (fused LayerNorm backward)");
  SE_REQUIRE(M > 0 && hid >= 64 && hid % 64 == 0, "ff_bwd_dgrad: M=%ld hid=%d (hid must be a multiple of 64)", M, hid);
  SE_REQUIRE(precision == 1 || precision == 2, "ff_bwd_dgrad: precision must be 1 (bf16x3) or 2 (bf16x6)");
  SE_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "ff_bwd_dgrad: drop_p=%f out of range", drop_p);
  SE_REQUIRE(M * (long)hid < 4294967296L, "ff_bwd_dgrad: dropout index exceeds 32 bits");
  FfBwdArgs a{dY, H, W2T, W1T, dZ, dLN, M, hid, drop_p, seed_h, seed_o, X, stats, gamma, dR2, dX, dgamma, dbeta};
  dim3 grid((unsigned)((M + 127) / 128)), block(256);
  if (precision == 1) hipLaunchKernelGGL(ff_bwd_kernel<2>, grid, block, 0, as_stream(stream), a);
  else hipLaunchKernelGGL(ff_bwd_kernel<3>, grid, block, 0, as_stream(stream), a);
  return se_check_launch("se_ff_bwd_dgrad");
}

extern "C" int se_ff_fwd(const float* X, const float* rowstats, const float* gamma, const float* beta, const float* W1,
                         const float* b1, const float* W2, const float* b2, float* H, float* Y, long M, int hid,
                         float drop_p, unsigned seed_h, unsigned seed_o, float alpha, int precision, void* stream) {
  SE_REQUIRE(X && rowstats && gamma && beta && W1 && b1 && W2 && b2 && H && Y, "ff_fwd: null operand");
  SE_REQUIRE(M > 0 && hid >= 64 && hid % 64 == 0, "ff_fwd: M=%ld hid=%d (hid must be a multiple of 64)", M, hid);
  SE_REQUIRE(precision == 1 || precision == 2, "ff_fwd: precision must be 1 (bf16x3) or 2 (bf16x6)");
  SE_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "ff_fwd: drop_p=%f out of range", drop_p);
  SE_REQUIRE(M * (long)hid < 4294967296L, "ff_fwd: dropout index exceeds 32 bits");
  FfArgs a{X, rowstats, gamma, beta, W1, b1, W2, b2, H, Y, M, hid, drop_p, seed_h, seed_o, alpha};
  dim3 grid((unsigned)((M + 127) / 128)), block(256);
  if (precision == 1) hipLaunchKernelGGL(ff_fwd_kernel<2>, grid, block, 0, as_stream(stream), a);
  else hipLaunchKernelGGL(ff_fwd_kernel<3>, grid, block, 0, as_stream(stream), a);
  return se_check_launch("se_ff_fwd");
}

// ---------------------------------------------------------------------------------------------
// weight gradient:  dW[n][tap*C + c] += sum_m dY[m][n] * pro(A[src(m,tap)][c])
// grid: (row chunks, ntap * ceil(C/64), ceil(N/64)); 4 waves = 2x2 tiles of 32(n) x 32(c).
struct WgradArgs {
  se_gemm_desc d;
  const float* A; const float* dY; float* dW; float* dbias;
  const float* rowstats; const float* ps; const float* pb;
  long rows_per_chunk;
  int nchunks;
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
  const int col = lane & 31, half = lane >> 5;
  const int c = cb * 64 + wc * 32 + col;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    int n = nb * 64 + wn * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
    if (n < d.N && c < d.C) atomicAdd(&g.dW[(long)n * d.ldw + (long)tap * d.C + c], acc[r]);
  }
  if (do_bias && tid < 64 && nb * 64 + tid < d.N) atomicAdd(&g.dbias[nb * 64 + tid], bsum);
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
  const int col = lane & 31, half = lane >> 5;
  const int c = cb * 64 + wc * 32 + col;
#pragma unroll
  for (int s3 = 0; s3 < 3; ++s3) {
    const int tap = 3 * gi + s3;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      int n = nb * 64 + wn * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      if (n < d.N && c < d.C) atomicAdd(&g.dW[(long)n * d.ldw + (long)tap * d.C + c], acc[s3][r]);
    }
  }
  if (do_bias && tid < 64 && nb * 64 + tid < d.N) atomicAdd(&g.dbias[nb * 64 + tid], bsum);
}

// ---------------------------------------------------------------------------------------------
// Split-bf16 weight gradient (precision 1 / 2, same operand splits as gemm_tap_bf16x3_kernel).  The contraction index
// of dW = dY^T X is the ROW index m, and v_mfma_f32_32x32x16_bf16 wants 8 consecutive k per lane: both tiles are
// therefore staged TRANSPOSED in LDS ([column][m], m contiguous).  Every thread owns a 4 (rows) x 4 (columns) block:
// four 16-B global loads (one per row, 16 lanes = one 256-B row segment), a register transpose, and per column one
// 8-B store of 4 consecutive m per plane.  LDS rows are laid out in 16-B cells, cell(r, ch) = 9 r + (r >> 4) + ch:
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
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl) {
        bf16x4 h;
#pragma unroll
        for (int i = 0; i < 4; ++i) { h[i] = (__bf16)e[i]; e[i] -= (float)h[i]; }
        *reinterpret_cast<bf16x4*>(dst + pl * PLN) = h;
      }
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
  const int col = lane & 31, half = lane >> 5;
  const int c = cb * 64 + wc * 32 + col;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    int n = nb * 64 + wn * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
    if (n < d.N && c < d.C) atomicAdd(&g.dW[(long)n * d.ldw + (long)tap * d.C + c], acc[r]);
  }
  if (do_bias) {       // column sums of dY: fold the 16 row groups through LDS (the tiles are free now)
    float* red = reinterpret_cast<float*>(Yt);
    *reinterpret_cast<float4*>(&red[rg * 64 + 4 * q]) = bsum;
    __syncthreads();
    if (tid < 64) {
      float s_ = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) s_ += red[r * 64 + tid];
      if (nb * 64 + tid < d.N) atomicAdd(&g.dbias[nb * 64 + tid], s_);
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
template <int NPL>
__global__ __launch_bounds__(256) void wgrad3_bf16_kernel(WgradArgs g) {
  constexpr int MR = 64;
  constexpr int PLY = (9 * 64 + 4) * 8;       // Yt plane: [64 n][64 m],  cell(r, ch) = 9 r + (r >> 4) + ch
  constexpr int PLX = (10 * 64 + 8) * 8;      // Xt plane: [64 c][80 positions], cell(r, ch) = 10 r + (r >> 3) + ch
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
      ry[i] = (ok && nok) ? *reinterpret_cast<const float4*>(Yg + mg * d.ldc) : make_float4(0.f, 0.f, 0.f, 0.f);
      const int inb = ip[i] + ishift;
      const bool v = ok && cok && inb >= 0 && inb < Mb;
      rx[i] = v ? *reinterpret_cast<const float4*>(Ag + (eb[i] + inb) * d.lda) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (tid < 32) {
      const int hsel = tid >> 4;
      const long mg = hsel ? mbase + MR - 1 : mbase;
      const int inb = ip[4] + ishift;
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
  // 4 x 4 register transpose + split + 8-B store per column; poff = position of tile row 0 inside the LDS row
  auto stage_t = [&](const float4 (&v)[4], __bf16* T, int pln, bool halo_layout) {
    const float x[4][4] = {{v[0].x, v[0].y, v[0].z, v[0].w}, {v[1].x, v[1].y, v[1].z, v[1].w},
                           {v[2].x, v[2].y, v[2].z, v[2].w}, {v[3].x, v[3].y, v[3].z, v[3].w}};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int r = 4 * q + j;
      __bf16* dst = halo_layout ? T + (10 * r + (r >> 3)) * 8 + 8 + 4 * rg : T + (9 * r + (r >> 4)) * 8 + 4 * rg;
      float e[4] = {x[0][j], x[1][j], x[2][j], x[3][j]};
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl) {
        bf16x4 h;
#pragma unroll
        for (int i = 0; i < 4; ++i) { h[i] = (__bf16)e[i]; e[i] -= (float)h[i]; }
        *reinterpret_cast<bf16x4*>(dst + pl * pln) = h;
      }
    }
  };
  int fbase = (int)((mbeg % Mb) % d.Fo);      // frequency index of the step's first row (wave-uniform)
  if (mbeg < mend) load_tiles(mbeg);
  const int ra_ = wn * 32 + (lane & 31), rb_ = wc * 32 + (lane & 31), kg = lane >> 5;
  const __bf16* yfrag = Yt + (9 * ra_ + (ra_ >> 4)) * 8 + 8 * kg;
  const __bf16* xrow = Xt + (10 * rb_ + (rb_ >> 3)) * 8;
  for (long mb = mbeg; mb < mend; mb += MR) {
    stage_t(rx, Xt, PLX, true);
    stage_t(ry, Yt, PLY, false);
    if (tid < 32) {                            // halo rows: positions 7 and 72
      const int ph = (tid >> 4) ? 72 : 7;
      const float hv[4] = {rh.x, rh.y, rh.z, rh.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int r = 4 * q + j;
        float e = hv[j];
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) { __bf16 h = (__bf16)e; e -= (float)h; Xt[pl * PLX + (10 * r + (r >> 3)) * 8 + ph] = h; }
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
      const int jc = 2 * ks + kg + 1;
      bf16x8 af[NPL];
      unsigned cen[NPL][4], prv[NPL], nxt[NPL];
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl) {
        af[pl] = *reinterpret_cast<const bf16x8*>(yfrag + pl * PLY + 16 * ks);
        const __bf16* xp = xrow + pl * PLX + 8 * jc;
        const uint4 cv = *reinterpret_cast<const uint4*>(xp);
        cen[pl][0] = cv.x; cen[pl][1] = cv.y; cen[pl][2] = cv.z; cen[pl][3] = cv.w;
        prv[pl] = *reinterpret_cast<const unsigned*>(xp - 2);
        nxt[pl] = *reinterpret_cast<const unsigned*>(xp + 8);
      }
#pragma unroll
      for (int s3 = 0; s3 < 3; ++s3) {
        const int df = dfs[s3];
        bf16x8 bf[NPL];
        if (df == 0) {
#pragma unroll
          for (int pl = 0; pl < NPL; ++pl) { uint4 u = make_uint4(cen[pl][0], cen[pl][1], cen[pl][2], cen[pl][3]); bf[pl] = *reinterpret_cast<bf16x8*>(&u); }
        } else {
          // element e of the shifted window is an edge row -> cleared
          const int e = df > 0 ? pL - (8 * jc + 1) : pR - (8 * jc - 1);
          unsigned mk[4];
#pragma unroll
          for (int dd = 0; dd < 4; ++dd) mk[dd] = e == 2 * dd ? 0xFFFF0000u : (e == 2 * dd + 1 ? 0x0000FFFFu : 0xFFFFFFFFu);
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
            uint4 u = make_uint4(o[0] & mk[0], o[1] & mk[1], o[2] & mk[2], o[3] & mk[3]);
            bf[pl] = *reinterpret_cast<bf16x8*>(&u);
          }
        }
#pragma unroll
        for (int ord = NPL - 1; ord >= 0; --ord)
#pragma unroll
          for (int qa = 0; qa <= ord; ++qa)
            acc[s3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[qa], bf[ord - qa], acc[s3], 0, 0, 0);
      }
    }
    __syncthreads();
  }
  const int col = lane & 31, half = lane >> 5;
  const int c = cb * 64 + wc * 32 + col;
#pragma unroll
  for (int s3 = 0; s3 < 3; ++s3) {
    const int tap = 3 * gi + s3;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      int n = nb * 64 + wn * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      if (n < d.N && c < d.C) atomicAdd(&g.dW[(long)n * d.ldw + (long)tap * d.C + c], acc[s3][r]);
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
      if (nb * 64 + tid < d.N) atomicAdd(&g.dbias[nb * 64 + tid], s_);
    }
  }
}

// ---------------------------------------------------------------------------------------------
__global__ void repack_kernel(const float* src, float* dst, int No, int Nt, int Ni, long so, long stt,
                              long si, int rev, int accumulate) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long total = (long)No * Nt * Ni;
  if (idx >= total) return;
  int i = (int)(idx % Ni);
  int t = (int)((idx / Ni) % Nt);
  int o = (int)(idx / ((long)Ni * Nt));
  int is = i, os = o;
  if (rev == 1) { int ns = Ni / 64; is = (ns - 1 - i / 64) * 64 + (i & 63); }
  if (rev == 2) { int ns = No / 64; os = (ns - 1 - o / 64) * 64 + (o & 63); }
  float v = src[os * so + is * si + t * stt];
  if (accumulate) dst[idx] += v; else dst[idx] = v;
}

// scatter form used to fold a packed gradient back into the PyTorch layout:
// dst[o*so + i*si + t*stt] (+)= src[o][t][i]
__global__ void unpack_kernel(const float* src, float* dst, int No, int Nt, int Ni, long so, long stt,
                              long si, int rev, int accumulate) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long total = (long)No * Nt * Ni;
  if (idx >= total) return;
  int i = (int)(idx % Ni);
  int t = (int)((idx / Ni) % Nt);
  int o = (int)(idx / ((long)Ni * Nt));
  int is = i, os = o;
  if (rev == 1) { int ns = Ni / 64; is = (ns - 1 - i / 64) * 64 + (i & 63); }
  if (rev == 2) { int ns = No / 64; os = (ns - 1 - o / 64) * 64 + (o & 63); }
  float* p = &dst[os * so + is * si + t * stt];
  if (accumulate) *p += src[idx]; else *p = src[idx];
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
  if (d->prologue == SE_PRO_SWISH_DROP || d->prologue == SE_PRO_DROP || (d->epilogue & SE_EPI_DROP))
    SE_REQUIRE((long)d->B * d->Ti * d->Fi * (d->C > d->N ? d->C : d->N) < 4294967296L, "gemm: dropout index exceeds 32 bits");
  return 0;
}

extern "C" int se_gemm_tap(const se_gemm_desc* d, const float* A, const float* W, const float* bias,
                           float* Y, const float* R, float* AUX, const float* rowstats,
                           const float* pro_scale, const float* pro_shift, double* stats, void* stream) {
  if (int e = check_desc(d)) return e;
  const int ep = d->epilogue;
  SE_REQUIRE(A && W && Y, "gemm: null operand");
  SE_REQUIRE(!(ep & SE_EPI_BIAS) || bias, "gemm: bias flag without bias");
  SE_REQUIRE(!(ep & SE_EPI_RESID) || R, "gemm: resid flag without R");
  SE_REQUIRE(!(ep & SE_EPI_SWISH_GRAD) || AUX, "gemm: swish-grad flag without AUX");
  SE_REQUIRE(!(ep & SE_EPI_STATS) || stats, "gemm: stats flag without buffer");
  SE_REQUIRE(!(ep & (SE_EPI_GLU | SE_EPI_SHUFFLE2)) || (d->N % 2) == 0, "gemm: GLU/shuffle need even N");
  if (d->prologue == SE_PRO_LN) SE_REQUIRE(rowstats && pro_scale && pro_shift, "gemm: LN prologue operands");
  if (d->prologue == SE_PRO_AFFINE_SWISH) SE_REQUIRE(pro_scale && pro_shift, "gemm: affine prologue operands");
  GemmArgs g{*d, A, W, bias, Y, R, AUX, rowstats, pro_scale, pro_shift, stats, 0, 0, 0, 0};
  const int Mb = d->To * d->Fo;
  const int ncols = (ep & SE_EPI_GLU) ? cdiv(d->N / 2, 32) : cdiv(d->N, 64);
  g.ncb = ncols;
  g.tiles = cdiv(Mb, 128);
  g.nouter = d->B * g.tiles;
  g.contig = d->ntap > 1;
  dim3 grid((unsigned)(ncols * (((long)g.nouter + 7) / 8 * 8))), block(256);
  // row GEMM (nn.Linear, 1x1 conv): one tap, unit strides, identical in/out grids -> source pixel == row index
  const bool lin = d->ntap == 1 && !d->up && d->st == 1 && d->sf == 1 && d->dt[0] == 0 && d->df[0] == 0 &&
                   d->Ti == d->To && d->Fi == d->Fo;
  hipStream_t s = as_stream(stream);
  {
    const bool vec_ok = !(ep & (SE_EPI_GLU | SE_EPI_SHUFFLE2 | SE_EPI_STATS | 256)) && (d->N & 3) == 0 && (d->ldc & 3) == 0 &&
                        (d->c_off & 3) == 0 && (d->ldx & 3) == 0 && (d->x_off & 3) == 0 && (d->ldr & 3) == 0 && (d->r_off & 3) == 0;
    const bool glu_ok = (ep & SE_EPI_GLU) && !(ep & (SE_EPI_STATS | SE_EPI_SHUFFLE2 | SE_EPI_DROP | SE_EPI_RESID | SE_EPI_ACCUM |
                                                       SE_EPI_SWISH_GRAD | 256)) &&
                        (d->N & 7) == 0 && (d->ldc & 3) == 0 && (d->c_off & 3) == 0 && (d->ldx & 3) == 0 && (d->x_off & 3) == 0;
    static const bool no_panel = getenv("SE_GEMM_NO_PANEL") != nullptr;
    if (lin && d->B == 1 && d->C == 64 && ncols >= 2 && (d->precision == 1 || d->precision == 2) && (vec_ok || glu_ok) &&
        !(ep & SE_EPI_ACCUM) && !((ep & SE_EPI_SWISH_GRAD) && (ep & SE_EPI_RESID)) && !no_panel) {
      dim3 pgrid(g.tiles);
      const bool pre2 = (ep & (SE_EPI_SWISH_GRAD | SE_EPI_RESID)) != 0;
#define LAUNCHP2(PRO, P2) do { if (d->precision == 1) hipLaunchKernelGGL((gemm_k64_panel_kernel<PRO, 2, P2>), pgrid, block, 0, s, g); \
                          else hipLaunchKernelGGL((gemm_k64_panel_kernel<PRO, 3, P2>), pgrid, block, 0, s, g); } while (0)
#define LAUNCHP(PRO) do { if (pre2) LAUNCHP2(PRO, true); else LAUNCHP2(PRO, false); } while (0)
      switch (d->prologue) {
        case SE_PRO_NONE: LAUNCHP(SE_PRO_NONE); break;
        case SE_PRO_LN: LAUNCHP(SE_PRO_LN); break;
        case SE_PRO_SWISH: LAUNCHP(SE_PRO_SWISH); break;
        case SE_PRO_AFFINE_SWISH: LAUNCHP(SE_PRO_AFFINE_SWISH); break;
        case SE_PRO_SWISH_DROP: LAUNCHP(SE_PRO_SWISH_DROP); break;
        case SE_PRO_DROP: LAUNCHP(SE_PRO_DROP); break;
        default: return se_fail("gemm: unknown prologue %d", d->prologue);
      }
#undef LAUNCHP
#undef LAUNCHP2
      return se_check_launch("se_gemm_tap(k64 panel)");
    }
  }
  if ((d->precision == 1 || d->precision == 2) && d->C >= 32 && d->prologue == SE_PRO_NONE && !d->up && d->st == 1 && d->sf == 1 &&
      d->Ti == d->To && d->Fi == d->Fo && d->ntap >= 3 && d->ntap % 3 == 0 && !(ep & (SE_EPI_GLU | SE_EPI_DROP)) && d->Fo >= 2) {
    bool triples = getenv("SE_GEMM_NO_CONV3") == nullptr;
    for (int t3 = 0; t3 < d->ntap && triples; t3 += 3) {
      int seen = 0;
      for (int j = 0; j < 3; ++j) {
        if (d->dt[t3 + j] != d->dt[t3] || d->df[t3 + j] < -1 || d->df[t3 + j] > 1) triples = false;
        else seen |= 1 << (d->df[t3 + j] + 1);
      }
      if (seen != 7) triples = false;
    }
    if (triples) {
      if (d->precision == 1) hipLaunchKernelGGL((conv3_bf16_kernel<2>), grid, block, 0, s, g);
      else hipLaunchKernelGGL((conv3_bf16_kernel<3>), grid, block, 0, s, g);
      return se_check_launch("se_gemm_tap(conv3)");
    }
  }
  if ((d->precision == 1 || d->precision == 2) && d->C >= 32) {      // split-bf16 paths (BK = 32 only)
#define LAUNCHB2(PRO, LIN_) do { if (d->precision == 1) hipLaunchKernelGGL((gemm_tap_bf16x3_kernel<PRO, 2, LIN_>), grid, block, 0, s, g); \
                          else hipLaunchKernelGGL((gemm_tap_bf16x3_kernel<PRO, 3, LIN_>), grid, block, 0, s, g); } while (0)
#define LAUNCHB(PRO) do { if (lin) LAUNCHB2(PRO, true); else LAUNCHB2(PRO, false); } while (0)
    switch (d->prologue) {
      case SE_PRO_NONE: LAUNCHB(SE_PRO_NONE); break;
      case SE_PRO_LN: LAUNCHB(SE_PRO_LN); break;
      case SE_PRO_SWISH: LAUNCHB(SE_PRO_SWISH); break;
      case SE_PRO_AFFINE_SWISH: LAUNCHB(SE_PRO_AFFINE_SWISH); break;
      case SE_PRO_SWISH_DROP: LAUNCHB(SE_PRO_SWISH_DROP); break;
      case SE_PRO_DROP: LAUNCHB(SE_PRO_DROP); break;
      default: return se_fail("gemm: unknown prologue %d", d->prologue);
    }
#undef LAUNCHB
#undef LAUNCHB2
    return se_check_launch("se_gemm_tap(bf16x3)");
  }
  int bk = d->C < 32 ? 16 : 32;    // 64 measured slower (occupancy 3 -> latency-bound)

#define LAUNCH(BK, PRO) do { if (lin) hipLaunchKernelGGL((gemm_tap_kernel<BK, PRO, true>), grid, block, 0, s, g); \
                             else hipLaunchKernelGGL((gemm_tap_kernel<BK, PRO, false>), grid, block, 0, s, g); } while (0)
#define LAUNCH_BK(PRO) do { if (bk == 16) LAUNCH(16, PRO); else LAUNCH(32, PRO); } while (0)
  switch (d->prologue) {
    case SE_PRO_NONE: LAUNCH_BK(SE_PRO_NONE); break;
    case SE_PRO_LN: LAUNCH_BK(SE_PRO_LN); break;
    case SE_PRO_SWISH: LAUNCH_BK(SE_PRO_SWISH); break;
    case SE_PRO_AFFINE_SWISH: LAUNCH_BK(SE_PRO_AFFINE_SWISH); break;
    case SE_PRO_SWISH_DROP: LAUNCH_BK(SE_PRO_SWISH_DROP); break;
    case SE_PRO_DROP: LAUNCH_BK(SE_PRO_DROP); break;
    default: return se_fail("gemm: unknown prologue %d", d->prologue);
  }
#undef LAUNCH_BK
#undef LAUNCH
  return se_check_launch("se_gemm_tap");
}

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
  WgradArgs g{*d, A, dY, dW, dbias, rowstats, pro_scale, pro_shift, rpc, chunks};
  dim3 grid((unsigned)((long)d->ntap * cdiv(d->C, 64) * cdiv(d->N, 64) * ((chunks + 7) / 8 * 8))), block(256);
  hipStream_t s = as_stream(stream);
  if (d->prologue == SE_PRO_NONE && !(d->epilogue & SE_EPI_DROP) && !d->up && d->st == 1 && d->sf == 1 &&
      d->Ti == d->To && d->Fi == d->Fo && d->ntap >= 3 && d->ntap % 3 == 0 && d->Fo >= 2 && d->To * d->Fo >= 64 &&
      getenv("SE_GEMM_NO_CONV3") == nullptr) {
    bool triples = true;
    for (int t3 = 0; t3 < d->ntap && triples; t3 += 3) {
      int seen = 0;
      for (int j = 0; j < 3; ++j) {
        if (d->dt[t3 + j] != d->dt[t3] || d->df[t3 + j] < -1 || d->df[t3 + j] > 1) triples = false;
        else seen |= 1 << (d->df[t3 + j] + 1);
      }
      if (seen != 7) triples = false;
    }
    if (triples && (d->precision == 0 || d->Fo > 66)) {
      dim3 g3((unsigned)((long)(d->ntap / 3) * cdiv(d->C, 64) * cdiv(d->N, 64) * ((chunks + 7) / 8 * 8)));
      if (d->precision == 1) hipLaunchKernelGGL(wgrad3_bf16_kernel<2>, g3, block, 0, s, g);
      else if (d->precision == 2) hipLaunchKernelGGL(wgrad3_bf16_kernel<3>, g3, block, 0, s, g);
      else hipLaunchKernelGGL(wgrad3_kernel, g3, block, 0, s, g);
      return se_check_launch("se_gemm_tap_wgrad(conv3)");
    }
  }
  // generic (non-triple) shapes: the six-product split kernel is VALU-bound by its own splits and measured slower than
  // the fp32-MFMA kernel it is numerically equivalent to (77 vs 83 TFLOP/s) -> precision 2 runs the fp32 kernel there
  if (d->precision == 1 || (d->precision == 2 && getenv("SE_WGRAD_FORCE_X6") != nullptr)) {
#define LAUNCHWB(PRO) do { if (d->precision == 1) hipLaunchKernelGGL((wgrad_bf16_kernel<PRO, 2>), grid, block, 0, s, g); \
                           else hipLaunchKernelGGL((wgrad_bf16_kernel<PRO, 3>), grid, block, 0, s, g); } while (0)
    switch (d->prologue) {
      case SE_PRO_NONE: LAUNCHWB(SE_PRO_NONE); break;
      case SE_PRO_LN: LAUNCHWB(SE_PRO_LN); break;
      case SE_PRO_SWISH: LAUNCHWB(SE_PRO_SWISH); break;
      case SE_PRO_AFFINE_SWISH: LAUNCHWB(SE_PRO_AFFINE_SWISH); break;
      case SE_PRO_SWISH_DROP: LAUNCHWB(SE_PRO_SWISH_DROP); break;
      case SE_PRO_DROP: LAUNCHWB(SE_PRO_DROP); break;
      default: return se_fail("wgrad: unknown prologue %d", d->prologue);
    }
#undef LAUNCHWB
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
  return se_check_launch("se_gemm_tap_wgrad");
}

extern "C" int se_repack(const float* src, float* dst, int No, int Nt, int Ni, long so, long stt, long si,
                         int rev, int accumulate, void* stream) {
  SE_REQUIRE(src && dst && No > 0 && Nt > 0 && Ni > 0, "repack: bad arguments");
  if (rev == 1) SE_REQUIRE(Ni % 64 == 0, "repack: slab reversal needs Ni %% 64 == 0");
  if (rev == 2) SE_REQUIRE(No % 64 == 0, "repack: slab reversal needs No %% 64 == 0");
  long total = (long)No * Nt * Ni;
  hipLaunchKernelGGL(repack_kernel, dim3(cdiv(total, 256)), dim3(256), 0, as_stream(stream), src, dst, No, Nt,
                     Ni, so, stt, si, rev, accumulate);
  return se_check_launch("se_repack");
}

extern "C" int se_unpack(const float* src, float* dst, int No, int Nt, int Ni, long so, long stt, long si,
                         int rev, int accumulate, void* stream) {
  SE_REQUIRE(src && dst && No > 0 && Nt > 0 && Ni > 0, "unpack: bad arguments");
  if (rev == 1) SE_REQUIRE(Ni % 64 == 0, "unpack: slab reversal needs Ni %% 64 == 0");
  if (rev == 2) SE_REQUIRE(No % 64 == 0, "unpack: slab reversal needs No %% 64 == 0");
  long total = (long)No * Nt * Ni;
  hipLaunchKernelGGL(unpack_kernel, dim3(cdiv(total, 256)), dim3(256), 0, as_stream(stream), src, dst, No, Nt,
                     Ni, so, stt, si, rev, accumulate);
  return se_check_launch("se_unpack");
}
