// HBM-bound normalisation kernels (gfx950): LayerNorm(64), InstanceNorm2d+PReLU, BatchNorm1d pieces.
// All are channels-last: a pixel/token row is C contiguous floats, lanes own float4 channel groups so
// every global access is 16 B/lane and a 64-channel row is one 256-B line per 16 lanes.
// Reductions: in-register over the rows a thread owns -> shuffles across the lanes that share a
// channel -> LDS across waves -> one fp64 (statistics) or fp32 (parameter gradients) atomic per
// channel per workgroup.
#include "se_common.h"

// ------------------------------------------------------------------------------------------
// LayerNorm over C = 64: 16 lanes per row, 4 rows per wave-iteration.
static __device__ __forceinline__ float sum16(float v) {
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__global__ __launch_bounds__(256) void row_stats64_kernel(const float* __restrict__ X, float* __restrict__ stats,
                                                          long M, int ld, float eps) {
  const int q = threadIdx.x & 15;
  long row = (long)blockIdx.x * 16 + (threadIdx.x >> 4);
  const long stride = (long)gridDim.x * 16;
  for (; row < M; row += stride) {
    float4 v = *reinterpret_cast<const float4*>(X + row * ld + q * 4);
    float mean = sum16(v.x + v.y + v.z + v.w) * (1.f / 64.f);
    float a = v.x - mean, b = v.y - mean, c = v.z - mean, d = v.w - mean;
    float var = sum16(a * a + b * b + c * c + d * d) * (1.f / 64.f);
    if (q == 0) { stats[2 * row] = mean; stats[2 * row + 1] = rsqrtf(var + eps); }
  }
}

// Y = LN(X)*g + b (+ R); stats written for the backward pass
__global__ __launch_bounds__(256) void layernorm64_fwd_kernel(const float* __restrict__ X, const float* __restrict__ g,
                                                              const float* __restrict__ b, const float* __restrict__ R,
                                                              float* __restrict__ Y, float* __restrict__ stats,
                                                              long M, float eps, float* __restrict__ out_stats) {
  const int q = threadIdx.x & 15;
  const float4 gg = *reinterpret_cast<const float4*>(g + q * 4);
  const float4 bb = *reinterpret_cast<const float4*>(b + q * 4);
  long row = (long)blockIdx.x * 16 + (threadIdx.x >> 4);
  const long stride = (long)gridDim.x * 16;
  for (; row < M; row += stride) {
    float4 v = *reinterpret_cast<const float4*>(X + row * 64 + q * 4);
    float mean = sum16(v.x + v.y + v.z + v.w) * (1.f / 64.f);
    float a0 = v.x - mean, a1 = v.y - mean, a2 = v.z - mean, a3 = v.w - mean;
    float var = sum16(a0 * a0 + a1 * a1 + a2 * a2 + a3 * a3) * (1.f / 64.f);
    float rstd = 1.0f / sqrtf(var + eps);
    float4 o = make_float4(a0 * rstd * gg.x + bb.x, a1 * rstd * gg.y + bb.y, a2 * rstd * gg.z + bb.z,
                           a3 * rstd * gg.w + bb.w);
    if (R) {
      float4 r = *reinterpret_cast<const float4*>(R + row * 64 + q * 4);
      o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
    }
    st4_stream_(Y + row * 64 + q * 4, o);
    if (stats && q == 0) { stats[2 * row] = mean; stats[2 * row + 1] = rstd; }
    if (out_stats) {      // (mean, rstd) of the RESULT row, exactly as row_stats64_kernel would compute them from Y: the next LayerNorm's
      const float m2 = sum16(o.x + o.y + o.z + o.w) * (1.f / 64.f);
      const float c0 = o.x - m2, c1 = o.y - m2, c2 = o.z - m2, c3 = o.w - m2;
      const float v2 = sum16(c0 * c0 + c1 * c1 + c2 * c2 + c3 * c3) * (1.f / 64.f);
      if (q == 0) { out_stats[2 * row] = m2; out_stats[2 * row + 1] = rsqrtf(v2 + eps); }
    }
  }
}

// dX = (dR) + rstd * (dxh - mean(dxh) - xh * mean(dxh * xh)),  dxh = dY * g;
// dg += sum dY * xh, db += sum dY   (fp32 atomics, one per channel per workgroup)
__global__ __launch_bounds__(256) void layernorm64_bwd_kernel(const float* __restrict__ X, const float* __restrict__ stats,
                                                              const float* __restrict__ g, const float* __restrict__ dY,
                                                              const float* __restrict__ dR, const float* __restrict__ dR2,
                                                              float* __restrict__ dX,
                                                              float* __restrict__ dg, float* __restrict__ db, long M,
                                                              float* amax_out) {
  __shared__ float red[16 * 64 * 2];
  float amx = 0.f;
  const int q = threadIdx.x & 15, sub = threadIdx.x >> 4;
  const float4 gg = *reinterpret_cast<const float4*>(g + q * 4);
  float ag[4] = {0, 0, 0, 0}, ab[4] = {0, 0, 0, 0};
  long row = (long)blockIdx.x * 16 + sub;
  const long stride = (long)gridDim.x * 16;
  for (; row < M; row += stride) {
    float4 v = *reinterpret_cast<const float4*>(X + row * 64 + q * 4);
    float4 dy = *reinterpret_cast<const float4*>(dY + row * 64 + q * 4);
    float mean = stats[2 * row], rstd = stats[2 * row + 1];
    float xh[4] = {(v.x - mean) * rstd, (v.y - mean) * rstd, (v.z - mean) * rstd, (v.w - mean) * rstd};
    float d[4] = {dy.x, dy.y, dy.z, dy.w};
    float gl[4] = {gg.x, gg.y, gg.z, gg.w};
    float dxh[4], s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      dxh[j] = d[j] * gl[j];
      s1 += dxh[j]; s2 += dxh[j] * xh[j];
      ag[j] += d[j] * xh[j]; ab[j] += d[j];
    }
    s1 = sum16(s1) * (1.f / 64.f);
    s2 = sum16(s2) * (1.f / 64.f);
    float o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = rstd * (dxh[j] - s1 - xh[j] * s2);
    if (dR) {
      float4 r = *reinterpret_cast<const float4*>(dR + row * 64 + q * 4);
      o[0] += r.x; o[1] += r.y; o[2] += r.z; o[3] += r.w;
    }
    if (dR2) {
      float4 r = *reinterpret_cast<const float4*>(dR2 + row * 64 + q * 4);
      o[0] += r.x; o[1] += r.y; o[2] += r.z; o[3] += r.w;
    }
    st4_stream_(dX + row * 64 + q * 4, make_float4(o[0], o[1], o[2], o[3]));
    amx = fmaxf(fmaxf(amx, fmaxf(fabsf(o[0]), fabsf(o[1]))), fmaxf(fabsf(o[2]), fabsf(o[3])));
  }
  if (amax_out) {
    amx = wave_max(amx);
    if ((threadIdx.x & 63) == 0) amax_raise_(amax_out, amx);
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) { red[(sub * 64 + q * 4 + j) * 2] = ag[j]; red[(sub * 64 + q * 4 + j) * 2 + 1] = ab[j]; }
  __syncthreads();
  if (threadIdx.x < 64) {
    float sg = 0.f, sb = 0.f;
#pragma unroll
    for (int s = 0; s < 16; ++s) { sg += red[(s * 64 + threadIdx.x) * 2]; sb += red[(s * 64 + threadIdx.x) * 2 + 1]; }
    atomicAdd(&dg[threadIdx.x], sg);
    atomicAdd(&db[threadIdx.x], sb);
  }
}

// ------------------------------------------------------------------------------------------
// Generic channels-last helpers: C in {4..256} multiple of 4, (C/4) lanes per pixel, 256/(C/4) pixels
// per block-iteration.  grid = (pixel chunks, B).
struct ChanIter {
  int lanes, psub, q, sub;
  __device__ ChanIter(int C) { lanes = C >> 2; psub = 256 / lanes; q = threadIdx.x % lanes; sub = threadIdx.x / lanes; }
};

// reduce per-thread channel partials [4*NV] over the pixel-sub dimension, then atomics (double)
template <int NV>
static __device__ __forceinline__ void block_reduce_atomic_d(float (&acc)[NV][4], const ChanIter& it, int C,
                                                              double* dst /* [C][NV] */) {
  __shared__ float red[256 * 4 * NV];
#pragma unroll
  for (int v = 0; v < NV; ++v)
#pragma unroll
    for (int j = 0; j < 4; ++j) red[((it.sub * C) + it.q * 4 + j) * NV + v] = acc[v][j];
  __syncthreads();
  for (int idx = threadIdx.x; idx < C * NV; idx += 256) {
    float s = 0.f;
    for (int p = 0; p < it.psub; ++p) s += red[p * C * NV + idx];
    atomicAdd(&dst[idx], (double)s);
  }
}

// column statistics of X: stats[b][c][2] += (sum, sumsq) over pixels
__global__ __launch_bounds__(256) void col_stats_kernel(const float* __restrict__ X, int ld, int x_off,
                                                        double* __restrict__ stats, long P, int C) {
  ChanIter it(C);
  const int b = blockIdx.y;
  float acc[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
  for (long p = (long)blockIdx.x * it.psub + it.sub; p < P; p += (long)gridDim.x * it.psub) {
    float4 v = *reinterpret_cast<const float4*>(X + ((long)b * P + p) * ld + x_off + it.q * 4);
    acc[0][0] += v.x; acc[0][1] += v.y; acc[0][2] += v.z; acc[0][3] += v.w;
    acc[1][0] += v.x * v.x; acc[1][1] += v.y * v.y; acc[1][2] += v.z * v.z; acc[1][3] += v.w * v.w;
  }
  block_reduce_atomic_d<2>(acc, it, C, stats + (long)b * C * 2);
}

// (sum, sumsq) -> (mean, rstd, scale = rstd*g, shift = beta - mean*rstd*g) per (b, c);  nb = B (instance
// norm) or 1 (batch norm; then running statistics are updated too when rm != NULL)
__global__ void norm_finalize_kernel(const double* __restrict__ stats, const float* __restrict__ g,
                                     const float* __restrict__ beta, float* __restrict__ mr,
                                     float* __restrict__ ss, int nb, int C, double count, float eps,
                                     float* rm, float* rv, float momentum) {
  int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= nb * C) return;
  int c = idx % C;
  double mean = stats[2 * idx] / count;
  double var = stats[2 * idx + 1] / count - mean * mean;
  if (var < 0) var = 0;
  float rstd = (float)(1.0 / sqrt(var + (double)eps));
  mr[2 * idx] = (float)mean;
  mr[2 * idx + 1] = rstd;
  float sc = rstd * g[c];
  ss[2 * idx] = sc;
  ss[2 * idx + 1] = beta[c] - (float)mean * sc;
  if (rm) {
    rm[c] = (1.f - momentum) * rm[c] + momentum * (float)mean;
    double unb = count > 1 ? var * count / (count - 1.0) : var;
    rv[c] = (1.f - momentum) * rv[c] + momentum * (float)unb;
  }
}

// eval-mode BatchNorm: scale/shift from running statistics
__global__ void bn_eval_scale_kernel(const float* rm, const float* rv, const float* g, const float* beta,
                                     float* ss, float* mr, int C, float eps) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float rstd = 1.0f / sqrtf(rv[c] + eps);
  ss[2 * c] = rstd * g[c];
  ss[2 * c + 1] = beta[c] - rm[c] * rstd * g[c];
  mr[2 * c] = rm[c];
  mr[2 * c + 1] = rstd;
}

// Y[.., y_off + c] = prelu(X * scale + shift);  slope per channel (or per frequency when slope_f != 0:
// MaskDecoder.prelu_out, models/generator.py:104,111-112)
__global__ __launch_bounds__(256) void affine_prelu_kernel(const float* __restrict__ X, int ldx, int x_off,
                                                           const float* __restrict__ ss, const float* __restrict__ slope,
                                                           float* __restrict__ Y, int ldy, int y_off, long P, int C) {
  ChanIter it(C);
  const int b = blockIdx.y;
  const float* s = ss + ((long)b * C + it.q * 4) * 2;
  const float sc[4] = {s[0], s[2], s[4], s[6]}, sh[4] = {s[1], s[3], s[5], s[7]};
  float sl[4] = {1.f, 1.f, 1.f, 1.f};
  if (slope) { sl[0] = slope[it.q * 4]; sl[1] = slope[it.q * 4 + 1]; sl[2] = slope[it.q * 4 + 2]; sl[3] = slope[it.q * 4 + 3]; }
  for (long p = (long)blockIdx.x * it.psub + it.sub; p < P; p += (long)gridDim.x * it.psub) {
    long pix = (long)b * P + p;
    float4 v = *reinterpret_cast<const float4*>(X + pix * ldx + x_off + it.q * 4);
    float u[4] = {v.x * sc[0] + sh[0], v.y * sc[1] + sh[1], v.z * sc[2] + sh[2], v.w * sc[3] + sh[3]};
#pragma unroll
    for (int j = 0; j < 4; ++j) u[j] = u[j] >= 0.f ? u[j] : u[j] * sl[j];
    st4_stream_(Y + pix * ldy + y_off + it.q * 4, make_float4(u[0], u[1], u[2], u[3]));
  }
}

constexpr int NB_U = 4;      // lines (line pairs) in flight per lane in the norm forward / backward passes

// InstanceNorm(affine) + PReLU forward in ONE launch: every workgroup turns the (sum, sumsq) of its 4 channels per lane into
// (scale, shift) itself (same fp64 arithmetic as norm_finalize_kernel), the first workgroup of a batch entry also stores
// (mean, rstd) for the backward pass; ~2048 workgroups over contiguous pixel runs like the backward passes.
__global__ __launch_bounds__(256) void inorm_prelu_fwd_kernel(
    const float* __restrict__ X, int ldx, int x_off, const double* __restrict__ stats, const float* __restrict__ g,
    const float* __restrict__ beta, const float* __restrict__ slope, float* __restrict__ Y, int ldy, int y_off,
    float* __restrict__ mr, long P, int C, double count, float eps, float* __restrict__ amax_out) {
  ChanIter it(C);
  const int b = blockIdx.y;
  float ymax = 0.f;                  // amax_out: max |Y| -- the operand scale of the scaled split-fp16 convolutions that read Y
  // one thread per channel does the fp64 arithmetic (two divisions, a square root, a reciprocal), the others pick the result up
  // from LDS: done by every lane for its 4 channels it was 256 / (C / 4) times redundant and cost more than the finalize launch
  __shared__ float scs[256], shs[256];
  if ((int)threadIdx.x < C) {
    const int c = threadIdx.x;
    const long idx = (long)b * C + c;
    const double mean = stats[2 * idx] / count;
    double var = stats[2 * idx + 1] / count - mean * mean;
    if (var < 0) var = 0;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    const float s_ = rstd * g[c];
    scs[c] = s_;
    shs[c] = beta[c] - (float)mean * s_;
    if (blockIdx.x == 0) { mr[2 * idx] = (float)mean; mr[2 * idx + 1] = rstd; }
  }
  __syncthreads();
  float sc[4], sh[4], sl[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int c = it.q * 4 + j;
    sc[j] = scs[c];
    sh[j] = shs[c];
    sl[j] = slope ? slope[c] : 1.f;
  }
  const long step = (long)it.psub * NB_U;
  const long chunk = ((P + gridDim.x - 1) / gridDim.x + step - 1) / step * step;
  const long p_end = min(P, (long)(blockIdx.x + 1) * chunk);
  const float* Xb = X + (long)b * P * ldx + x_off + it.q * 4;
  float* Yb = Y + (long)b * P * ldy + y_off + it.q * 4;
  for (long p0 = (long)blockIdx.x * chunk + it.sub; p0 < p_end; p0 += step) {
    float4 v[NB_U];
#pragma unroll
    for (int k = 0; k < NB_U; ++k) {
      const long p = p0 + (long)k * it.psub;
      v[k] = *reinterpret_cast<const float4*>(Xb + (p < p_end ? p : p0) * ldx);
    }
#pragma unroll
    for (int k = 0; k < NB_U; ++k) {
      const long p = p0 + (long)k * it.psub;
      float u[4] = {v[k].x * sc[0] + sh[0], v[k].y * sc[1] + sh[1], v[k].z * sc[2] + sh[2], v[k].w * sc[3] + sh[3]};
#pragma unroll
      for (int j = 0; j < 4; ++j) u[j] = u[j] >= 0.f ? u[j] : u[j] * sl[j];
      if (p < p_end) {
        st4_stream_(Yb + p * ldy, make_float4(u[0], u[1], u[2], u[3]));
        ymax = fmaxf(fmaxf(ymax, fmaxf(fabsf(u[0]), fabsf(u[1]))), fmaxf(fabsf(u[2]), fabsf(u[3])));
      }
    }
  }
  if (amax_out) {
    ymax = wave_max(ymax);
    if ((threadIdx.x & 63) == 0) amax_raise_(amax_out, ymax);
  }
}

// backward pass 1 of  Y = prelu(xh*g + beta), xh = (X-mean)*rstd:
// red[b][c][3] += (sum du, sum du*xh, sum dY*u*[u<0]),  du = dY * prelu'(u)
__global__ __launch_bounds__(256) void norm_prelu_bwd_reduce_kernel(
    const float* __restrict__ X, int ldx, int x_off, const float* __restrict__ mr, const float* __restrict__ g,
    const float* __restrict__ beta, const float* __restrict__ slope, const float* __restrict__ dY, int ldy, int y_off,
    double* __restrict__ red, long P, int C, int per_batch, int act) {
  ChanIter it(C);
  const int b = blockIdx.y;
  const int sb = per_batch ? b : 0;
  float mean[4], rstd[4], gg[4], bt[4], sl[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    int c = it.q * 4 + j;
    mean[j] = mr[((long)sb * C + c) * 2]; rstd[j] = mr[((long)sb * C + c) * 2 + 1];
    gg[j] = g[c]; bt[j] = beta[c]; sl[j] = slope ? slope[c] : 1.f;
  }
  float acc[3][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  // each workgroup owns one contiguous run of pixels; NB_U independent (X, dY) line pairs are requested per lane before any
  // of them is consumed (one pair in flight per lane left the memory pipe half empty: 3.0 TB/s)
  const long step = (long)it.psub * NB_U;
  const long chunk = ((P + gridDim.x - 1) / gridDim.x + step - 1) / step * step;
  const long p_end = min(P, (long)(blockIdx.x + 1) * chunk);
  const float* Xb = X + (long)b * P * ldx + x_off + it.q * 4;
  const float* Db = dY + (long)b * P * ldy + y_off + it.q * 4;
  for (long p0 = (long)blockIdx.x * chunk + it.sub; p0 < p_end; p0 += step) {
    float4 v[NB_U], d[NB_U];
#pragma unroll
    for (int k = 0; k < NB_U; ++k) {
      long p = p0 + (long)k * it.psub;
      bool ok = p < p_end;
      long pc = ok ? p : p0;
      v[k] = *reinterpret_cast<const float4*>(Xb + pc * ldx);
#ifdef SE_NORM_REDUCE_TWIN      // measurement build (tools/inorm_bwd_ab.py): the reduction WITHOUT its dY stream -- what is left of this pass when
      d[k] = v[k];              // the producer of dY emits the sums from its epilogue (which must still read x): wrong results
      (void)Db;
#else
      d[k] = *reinterpret_cast<const float4*>(Db + pc * ldy);
#endif
      if (!ok) d[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int k = 0; k < NB_U; ++k) {
      float x[4] = {v[k].x, v[k].y, v[k].z, v[k].w}, dy[4] = {d[k].x, d[k].y, d[k].z, d[k].w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float xh = (x[j] - mean[j]) * rstd[j];
        float u = xh * gg[j] + bt[j];
        float du = act ? dy[j] * swish_gradf_(u) : (u >= 0.f ? dy[j] : dy[j] * sl[j]);
        acc[0][j] += du; acc[1][j] += du * xh;
        acc[2][j] += (act || u >= 0.f) ? 0.f : dy[j] * u;
      }
    }
  }
  block_reduce_atomic_d<3>(acc, it, C, red + (long)sb * C * 3);
}

// backward pass 2: dX = rstd*g*(du - S1/cnt - xh*S2/cnt)
__global__ __launch_bounds__(256) void norm_prelu_bwd_apply_kernel(
    const float* __restrict__ X, int ldx, int x_off, const float* __restrict__ mr, const float* __restrict__ g,
    const float* __restrict__ beta, const float* __restrict__ slope, const float* __restrict__ dY, int ldy, int y_off,
    const double* __restrict__ red, float* __restrict__ dX, int lddx, int dx_off, long P, int C, int per_batch,
    double count, int act, float* dg, float* dbeta, float* dslope, int nbs, float* amax_out) {
  ChanIter it(C);
  const int b = blockIdx.y;
  const int sb = per_batch ? b : 0;
  float amx = 0.f;                               // max |dX| written by this lane (amax_out: the consumer's fp16 operand scale)
  if (dg && blockIdx.x == 0 && b == 0 && threadIdx.x < C) {      // norm_param_grad_kernel's work, one launch less
    const int c = threadIdx.x;
    double s1 = 0, s2 = 0, s3 = 0;
    for (int k = 0; k < nbs; ++k) { s1 += red[((long)k * C + c) * 3]; s2 += red[((long)k * C + c) * 3 + 1]; s3 += red[((long)k * C + c) * 3 + 2]; }
    dg[c] += (float)s2;
    dbeta[c] += (float)s1;
    if (dslope) dslope[c] += (float)s3;
  }
  __shared__ float m1s[256], m2s[256];          // the two fp64 divisions once per channel, not once per lane and channel
  if ((int)threadIdx.x < C) {
    const int c = threadIdx.x;
    m1s[c] = (float)(red[((long)sb * C + c) * 3] / count);
    m2s[c] = (float)(red[((long)sb * C + c) * 3 + 1] / count);
  }
  __syncthreads();
  float mean[4], rstd[4], gg[4], bt[4], sl[4], m1[4], m2[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    int c = it.q * 4 + j;
    mean[j] = mr[((long)sb * C + c) * 2]; rstd[j] = mr[((long)sb * C + c) * 2 + 1];
    gg[j] = g[c]; bt[j] = beta[c]; sl[j] = slope ? slope[c] : 1.f;
    m1[j] = m1s[c];
    m2[j] = m2s[c];
  }
  const long step = (long)it.psub * NB_U;
  const long chunk = ((P + gridDim.x - 1) / gridDim.x + step - 1) / step * step;
  const long p_end = min(P, (long)(blockIdx.x + 1) * chunk);
  const float* Xb = X + (long)b * P * ldx + x_off + it.q * 4;
  const float* Db = dY + (long)b * P * ldy + y_off + it.q * 4;
  float* Ob = dX + (long)b * P * lddx + dx_off + it.q * 4;
  for (long p0 = (long)blockIdx.x * chunk + it.sub; p0 < p_end; p0 += step) {
    float4 v[NB_U], d[NB_U];
#pragma unroll
    for (int k = 0; k < NB_U; ++k) {
      long p = p0 + (long)k * it.psub;
      long pc = p < p_end ? p : p0;
      v[k] = *reinterpret_cast<const float4*>(Xb + pc * ldx);
      d[k] = *reinterpret_cast<const float4*>(Db + pc * ldy);
    }
#pragma unroll
    for (int k = 0; k < NB_U; ++k) {
      long p = p0 + (long)k * it.psub;
      float x[4] = {v[k].x, v[k].y, v[k].z, v[k].w}, dy[4] = {d[k].x, d[k].y, d[k].z, d[k].w}, o[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float xh = (x[j] - mean[j]) * rstd[j];
        float u = xh * gg[j] + bt[j];
        float du = act ? dy[j] * swish_gradf_(u) : (u >= 0.f ? dy[j] : dy[j] * sl[j]);
        o[j] = rstd[j] * gg[j] * (du - m1[j] - xh * m2[j]);
      }
      if (p < p_end) {
        st4_stream_(Ob + p * lddx, make_float4(o[0], o[1], o[2], o[3]));
        amx = fmaxf(fmaxf(amx, fmaxf(fabsf(o[0]), fabsf(o[1]))), fmaxf(fabsf(o[2]), fabsf(o[3])));
      }
    }
  }
  if (amax_out) {                                // non-negative floats order like their bit patterns: one atomic per wave
    amx = wave_max(amx);
    if ((threadIdx.x & 63) == 0) amax_raise_(amax_out, amx);
  }
}


// parameter gradients from the reduced sums: dg[c] += sum_b S2, dbeta[c] += sum_b S1, dslope[c] += sum_b S3
__global__ void norm_param_grad_kernel(const double* red, float* dg, float* dbeta, float* dslope, int nb, int C) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double s1 = 0, s2 = 0, s3 = 0;
  for (int b = 0; b < nb; ++b) { s1 += red[((long)b * C + c) * 3]; s2 += red[((long)b * C + c) * 3 + 1]; s3 += red[((long)b * C + c) * 3 + 2]; }
  dg[c] += (float)s2;
  dbeta[c] += (float)s1;
  if (dslope) dslope[c] += (float)s3;
}

// ------------------------------------------------------------------------------------------
static int grid_for(long rows, int rows_per_block) {
  long nb = (rows + rows_per_block - 1) / rows_per_block;
  if (nb > 4096) nb = 4096;
  if (nb < 1) nb = 1;
  return (int)nb;
}
static int chan_ok(int C) { return C >= 4 && C <= 256 && (C % 4) == 0 && (256 % (C / 4)) == 0; }

extern "C" int se_row_stats(const float* X, float* stats, long M, int C, int ld, float eps, void* stream) {
  SE_REQUIRE(X && stats && M > 0, "row_stats: bad arguments");
  SE_REQUIRE(C == 64 && (ld % 4) == 0, "row_stats: only C == 64 is built (LayerNorm(64))");
  hipLaunchKernelGGL(row_stats64_kernel, dim3(grid_for(M, 16)), dim3(256), 0, as_stream(stream), X, stats, M, ld, eps);
  return se_check_launch("se_row_stats");
}

extern "C" int se_layernorm_fwd(const float* X, const float* g, const float* b, const float* R, float* Y,
                                float* stats, long M, int C, float eps, void* stream) {
  SE_REQUIRE(X && g && b && Y && M > 0, "layernorm_fwd: bad arguments");
  SE_REQUIRE(C == 64, "layernorm_fwd: only C == 64 is built");
  hipLaunchKernelGGL(layernorm64_fwd_kernel, dim3(grid_for(M, 16)), dim3(256), 0, as_stream(stream), X, g, b, R, Y,
                     stats, M, eps, (float*)nullptr);
  return se_check_launch("se_layernorm_fwd");
}

extern "C" int se_layernorm_fwd_stats(const float* X, const float* g, const float* b, const float* R, float* Y,
                                      float* stats, float* out_stats, long M, int C, float eps, void* stream) {
  SE_REQUIRE(X && g && b && Y && M > 0, "layernorm_fwd_stats: bad arguments");
  SE_REQUIRE(C == 64, "layernorm_fwd_stats: only C == 64 is built");
  hipLaunchKernelGGL(layernorm64_fwd_kernel, dim3(grid_for(M, 16)), dim3(256), 0, as_stream(stream), X, g, b, R, Y,
                     stats, M, eps, out_stats);
  return se_check_launch("se_layernorm_fwd_stats");
}

extern "C" int se_layernorm_bwd(const float* X, const float* stats, const float* g, const float* dY,
                                const float* dR, const float* dR2, float* dX, float* dg, float* db, long M, int C,
                                void* stream) {
  return se_layernorm_bwd_amax(X, stats, g, dY, dR, dR2, dX, dg, db, M, C, nullptr, stream);
}

extern "C" int se_layernorm_bwd_amax(const float* X, const float* stats, const float* g, const float* dY,
                                     const float* dR, const float* dR2, float* dX, float* dg, float* db, long M, int C,
                                     float* amax_out, void* stream) {
  SE_REQUIRE(X && stats && g && dY && dX && dg && db && M > 0, "layernorm_bwd: bad arguments");
  SE_REQUIRE(C == 64, "layernorm_bwd: only C == 64 is built");
  long nb = (M + 15) / 16;
  if (nb > 1024) nb = 1024;
  hipLaunchKernelGGL(layernorm64_bwd_kernel, dim3((int)nb), dim3(256), 0, as_stream(stream), X, stats, g, dY, dR, dR2, dX,
                     dg, db, M, amax_out);
  return se_check_launch("se_layernorm_bwd");
}

extern "C" int se_col_stats(const float* X, int ld, int x_off, double* stats, int B, long P, int C, void* stream) {
  SE_REQUIRE(X && stats && B > 0 && P > 0 && chan_ok(C), "col_stats: bad arguments (C=%d)", C);
  int psub = 256 / (C / 4);
  long nb = (P + psub - 1) / psub;
  if (nb > 512) nb = 512;
  hipLaunchKernelGGL(col_stats_kernel, dim3((int)nb, B), dim3(256), 0, as_stream(stream), X, ld, x_off, stats, P, C);
  return se_check_launch("se_col_stats");
}

extern "C" int se_norm_finalize(const double* stats, const float* g, const float* beta, float* mr, float* ss,
                                int nb, int C, double count, float eps, float* running_mean, float* running_var,
                                float momentum, void* stream) {
  SE_REQUIRE(stats && g && beta && mr && ss && nb > 0 && C > 0 && count > 0, "norm_finalize: bad arguments");
  hipLaunchKernelGGL(norm_finalize_kernel, dim3(cdiv((long)nb * C, 128)), dim3(128), 0, as_stream(stream), stats, g,
                     beta, mr, ss, nb, C, count, eps, running_mean, running_var, momentum);
  return se_check_launch("se_norm_finalize");
}

extern "C" int se_bn_eval_scale(const float* rm, const float* rv, const float* g, const float* beta, float* ss,
                                float* mr, int C, float eps, void* stream) {
  SE_REQUIRE(rm && rv && g && beta && ss && mr && C > 0, "bn_eval_scale: bad arguments");
  hipLaunchKernelGGL(bn_eval_scale_kernel, dim3(cdiv(C, 128)), dim3(128), 0, as_stream(stream), rm, rv, g, beta, ss,
                     mr, C, eps);
  return se_check_launch("se_bn_eval_scale");
}

extern "C" int se_affine_prelu(const float* X, int ldx, int x_off, const float* ss, const float* slope, float* Y,
                               int ldy, int y_off, int B, long P, int C, void* stream) {
  SE_REQUIRE(X && ss && Y && B > 0 && P > 0 && chan_ok(C), "affine_prelu: bad arguments (C=%d)", C);
  SE_REQUIRE((ldx % 4) == 0 && (x_off % 4) == 0 && (ldy % 4) == 0 && (y_off % 4) == 0, "affine_prelu: alignment");
  int psub = 256 / (C / 4);
  long nb = (P + psub - 1) / psub;
  if (nb > 1024) nb = 1024;
  hipLaunchKernelGGL(affine_prelu_kernel, dim3((int)nb, B), dim3(256), 0, as_stream(stream), X, ldx, x_off, ss, slope,
                     Y, ldy, y_off, P, C);
  return se_check_launch("se_affine_prelu");
}

extern "C" int se_inorm_prelu_fwd(const float* X, int ldx, int x_off, const double* stats, const float* g, const float* beta,
                                  const float* slope, float* Y, int ldy, int y_off, float* mr, int B, long P, int C,
                                  double count, float eps, void* stream) {
  return se_inorm_prelu_fwd_amax(X, ldx, x_off, stats, g, beta, slope, Y, ldy, y_off, mr, B, P, C, count, eps, nullptr, stream);
}

extern "C" int se_inorm_prelu_fwd_amax(const float* X, int ldx, int x_off, const double* stats, const float* g, const float* beta,
                                       const float* slope, float* Y, int ldy, int y_off, float* mr, int B, long P, int C,
                                       double count, float eps, float* amax_out, void* stream) {
  SE_REQUIRE(X && stats && g && beta && Y && mr && B > 0 && P > 0 && count > 0 && chan_ok(C), "inorm_prelu_fwd: bad arguments (C=%d)", C);
  SE_REQUIRE((ldx % 4) == 0 && (x_off % 4) == 0 && (ldy % 4) == 0 && (y_off % 4) == 0, "inorm_prelu_fwd: alignment");
  const int psub = 256 / (C / 4);
  long nb = (P + (long)psub * NB_U - 1) / ((long)psub * NB_U);
  const long nb_cap = 2048 / B > 1 ? 2048 / B : 1;
  if (nb > nb_cap) nb = nb_cap;
  hipLaunchKernelGGL(inorm_prelu_fwd_kernel, dim3((int)nb, B), dim3(256), 0, as_stream(stream), X, ldx, x_off, stats, g, beta,
                     slope, Y, ldy, y_off, mr, P, C, count, eps, amax_out);
  return se_check_launch("se_inorm_prelu_fwd");
}

extern "C" int se_norm_prelu_bwd(const float* X, int ldx, int x_off, const float* mr, const float* g,
                                 const float* beta, const float* slope, const float* dY, int ldy, int y_off,
                                 double* red, float* dX, int lddx, int dx_off, float* dg, float* dbeta,
                                 float* dslope, int B, long P, int C, int per_batch, int act, int phase, double count,
                                 void* stream) {
  return se_norm_prelu_bwd_amax(X, ldx, x_off, mr, g, beta, slope, dY, ldy, y_off, red, dX, lddx, dx_off, dg, dbeta, dslope, B, P, C,
                                per_batch, act, phase, count, nullptr, stream);
}

extern "C" int se_norm_prelu_bwd_amax(const float* X, int ldx, int x_off, const float* mr, const float* g,
                                      const float* beta, const float* slope, const float* dY, int ldy, int y_off,
                                      double* red, float* dX, int lddx, int dx_off, float* dg, float* dbeta,
                                      float* dslope, int B, long P, int C, int per_batch, int act, int phase, double count,
                                      float* amax_out, void* stream) {
  SE_REQUIRE(X && mr && g && beta && dY && red && dX && dg && dbeta && B > 0 && P > 0 && chan_ok(C),
             "norm_prelu_bwd: bad arguments (C=%d)", C);
  SE_REQUIRE((ldx % 4) == 0 && (x_off % 4) == 0 && (ldy % 4) == 0 && (y_off % 4) == 0 && (lddx % 4) == 0 &&
             (dx_off % 4) == 0, "norm_prelu_bwd: alignment");
  hipStream_t s = as_stream(stream);
  int nbs = per_batch ? B : 1;
  int psub = 256 / (C / 4);
  long nb = (P + (long)psub * NB_U - 1) / ((long)psub * NB_U);
  // ~2048 workgroups in total: each one pays a per-channel set-up (fp64 divisions) and, in pass 1, a block reduction + 3C fp64
  // atomics -- with 8192 four-iteration workgroups that overhead held the passes at 3.8 / 4.35 TB/s
  const long nb_cap = 2048 / B > 1 ? 2048 / B : 1;
  if (nb > nb_cap) nb = nb_cap;
  if (phase & 1) {      // reduce (a data-parallel caller all-reduces `red` between the two phases: SyncBatchNorm)
    if (!(phase & 16)) (void)hipMemsetAsync(red, 0, sizeof(double) * 3 * nbs * C, s);
    hipLaunchKernelGGL(norm_prelu_bwd_reduce_kernel, dim3((int)(nb > (nb_cap + 3) / 4 ? (nb_cap + 3) / 4 : nb), B), dim3(256), 0, s, X, ldx,
                       x_off, mr, g, beta, slope, dY, ldy, y_off, red, P, C, per_batch, act);
  }
  SE_REQUIRE(!((phase & 4) && (phase & 8)), "norm_prelu_bwd: phase bits 4 and 8 are alternatives");
  SE_REQUIRE(!(phase & 8) || (phase & 2), "norm_prelu_bwd: phase bit 8 rides on the apply pass (bit 2)");
  if (phase & 4)        // parameter gradients from the LOCAL sums (before any cross-rank all-reduce)
    hipLaunchKernelGGL(norm_param_grad_kernel, dim3(cdiv(C, 64)), dim3(64), 0, s, red, dg, dbeta, dslope, nbs, C);
  if (phase & 2) {
    hipLaunchKernelGGL(norm_prelu_bwd_apply_kernel, dim3((int)nb, B), dim3(256), 0, s, X, ldx,
                       x_off, mr, g, beta, slope, dY, ldy, y_off, red, dX, lddx, dx_off, P, C, per_batch, count, act,
                       (phase & 8) ? dg : nullptr, dbeta, dslope, nbs, amax_out);
  }
  return se_check_launch("se_norm_prelu_bwd");
}
