// "Thin" convolutions of the discriminator's first stage (models/discriminator.py:39-41: spectral-norm Conv2d(2, ndf, 4, 2, 1)),
// round 4.  On the [T, F] grid of the magnitude planes this layer has 2 input channels (4 in the plane format, two of them zero),
// ndf = 16 output channels and 16 taps with stride 2: as a tap GEMM it is a 128 x 64 x (16 taps x 16-channel chunk) tile problem
// whose tiles are 94 % padding and whose transposed form (the input gradient) zero-fills 12 of its 16 taps per pixel -- measured
// inside the step (bench.py with SE_KEY_SHAPES=1): forward 110 us x 3, input gradient 404 us (on the critical path of the
// generator loss), weight gradient 397 us x 2 for 8 / 16 / 33 MB of operands.  Direct kernels, one pixel (quarter) per thread:
//   forward        R[b][to][fo][n] = sum_{c, kh, kw} W[n][c][kh][kw] x[b][2 to + kw - 1][2 fo + kh - 1][c]   (+ InstanceNorm sums)
//   input gradient dx[b][t][f][c]  = sum over the 2 x 2 output pixels and 4 taps of its parity class
//   weight gradient dW[n][c][kh][kw] = sum_pixels dR[p][n] x[src(p, kh, kw)][c]
// x: planes [B][T][F][4] (channels 0, 1 used); W: the PyTorch layout [16][2][4][4] (kh over F, kw over T: the image is processed
// transposed, see discriminator.py); R, dR: [B][To][Fo][16].
#include "se_common.h"

namespace {
constexpr int DN = 16;      // output channels (ndf); host-checked

// ---- forward: 4 threads per output pixel (4 channels each); one batch entry per blockIdx.y ----
__global__ __launch_bounds__(256) void dconv1_fwd_kernel(const float* __restrict__ X, const float* __restrict__ W, float* __restrict__ R,
                                                         double* __restrict__ stats, int T, int F, int To, int Fo) {
  __shared__ __attribute__((aligned(16))) float ws[DN * 32];      // [kh][kw][c][n]: a thread's 4 channels are one 16-byte read
  __shared__ double red[4][DN][2];
  const int tid = threadIdx.x, b = blockIdx.y;
  for (int i = tid; i < DN * 32; i += 256) {           // W[n][c][kh][kw]
    const int kw = i & 3, kh = (i >> 2) & 3, c = (i >> 4) & 1, n = i >> 5;
    ws[((kh * 4 + kw) * 2 + c) * DN + n] = W[i];
  }
  __syncthreads();
  const int cg = tid & 3;                              // channels 4 cg .. 4 cg + 3
  const long P = (long)To * Fo;
  const float* __restrict__ Xb = X + (long)b * T * F * 4;
  float* __restrict__ Rb = R + (long)b * P * DN;
  float s[4] = {0.f, 0.f, 0.f, 0.f}, q[4] = {0.f, 0.f, 0.f, 0.f};
  for (long p = (long)blockIdx.x * 64 + (tid >> 2); p < P; p += (long)gridDim.x * 64) {
    const int to = (int)(p / Fo), fo = (int)(p - (long)to * Fo);
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    float2 xv[16];
#pragma unroll
    for (int kw = 0; kw < 4; ++kw) {
      const int t = 2 * to + kw - 1;
#pragma unroll
      for (int kh = 0; kh < 4; ++kh) {
        const int f = 2 * fo + kh - 1;
        const bool ok = t >= 0 && t < T && f >= 0 && f < F;
        const int tc = t < 0 ? 0 : (t >= T ? T - 1 : t), fc = f < 0 ? 0 : (f >= F ? F - 1 : f);      // (unconditional loads)
        const float2 v = *reinterpret_cast<const float2*>(Xb + ((long)tc * F + fc) * 4);
        xv[kh * 4 + kw] = make_float2(ok ? v.x : 0.f, ok ? v.y : 0.f);
      }
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const float4 w0 = *reinterpret_cast<const float4*>(&ws[(k * 2 + 0) * DN + 4 * cg]);
      const float4 w1 = *reinterpret_cast<const float4*>(&ws[(k * 2 + 1) * DN + 4 * cg]);
      acc[0] = fmaf(w0.x, xv[k].x, fmaf(w1.x, xv[k].y, acc[0]));
      acc[1] = fmaf(w0.y, xv[k].x, fmaf(w1.y, xv[k].y, acc[1]));
      acc[2] = fmaf(w0.z, xv[k].x, fmaf(w1.z, xv[k].y, acc[2]));
      acc[3] = fmaf(w0.w, xv[k].x, fmaf(w1.w, xv[k].y, acc[3]));
    }
    *reinterpret_cast<float4*>(Rb + p * DN + 4 * cg) = make_float4(acc[0], acc[1], acc[2], acc[3]);
#pragma unroll
    for (int j = 0; j < 4; ++j) { s[j] += acc[j]; q[j] = fmaf(acc[j], acc[j], q[j]); }
  }
  if (stats) {            // (sum, sum of squares) per (b, channel): lanes with equal cg folded, then the four waves, fp64 atomics
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
      for (int o = 4; o < 64; o <<= 1) { s[j] += __shfl_xor(s[j], o, 64); q[j] += __shfl_xor(q[j], o, 64); }
    }
    const int lane = tid & 63, wave = tid >> 6;
    if (lane < 4) {
#pragma unroll
      for (int j = 0; j < 4; ++j) { red[wave][4 * lane + j][0] = s[j]; red[wave][4 * lane + j][1] = q[j]; }
    }
    __syncthreads();
    if (tid < 2 * DN) {
      const int n = tid >> 1, k = tid & 1;
      const double v = red[0][n][k] + red[1][n][k] + red[2][n][k] + red[3][n][k];
      atomicAdd(&stats[((long)b * DN + n) * 2 + k], v);
    }
  }
}

// ---- input gradient: one input pixel per thread; dx channels 2, 3 are the zero padding of the plane format ----
__global__ __launch_bounds__(256) void dconv1_dgrad_kernel(const float* __restrict__ dR, const float* __restrict__ W, float* __restrict__ dX,
                                                           int T, int F, int To, int Fo) {
  __shared__ float ws[4 * 4 * DN * 2];                // [kw][kh][n][c]
  const int tid = threadIdx.x, b = blockIdx.y;
  for (int i = tid; i < DN * 32; i += 256) {           // W[n][c][kh][kw]
    const int kw = i & 3, kh = (i >> 2) & 3, c = (i >> 4) & 1, n = i >> 5;
    ws[((kw * 4 + kh) * DN + n) * 2 + c] = W[i];
  }
  __syncthreads();
  const long P = (long)T * F;
  const float* __restrict__ Rb = dR + (long)b * To * Fo * DN;
  float* __restrict__ Xb = dX + (long)b * P * 4;
  for (long p = (long)blockIdx.x * 256 + tid; p < P; p += (long)gridDim.x * 256) {
    const int t = (int)(p / F), f = (int)(p - (long)t * F);
    float a0 = 0.f, a1 = 0.f;
    // t = 2 to + kw - 1: kw has the parity of t + 1; the same in f
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int kw = ((t + 1) & 1) + 2 * i, to = (t + 1 - kw) >> 1;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int kh = ((f + 1) & 1) + 2 * j, fo = (f + 1 - kh) >> 1;
        if (to >= 0 && to < To && fo >= 0 && fo < Fo) {
          const float4* __restrict__ r4 = reinterpret_cast<const float4*>(Rb + ((long)to * Fo + fo) * DN);
          const float* w = &ws[(kw * 4 + kh) * DN * 2];
#pragma unroll
          for (int n4 = 0; n4 < 4; ++n4) {
            const float4 v = r4[n4];
            const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              a0 = fmaf(vv[e], w[(4 * n4 + e) * 2], a0);
              a1 = fmaf(vv[e], w[(4 * n4 + e) * 2 + 1], a1);
            }
          }
        }
      }
    }
    *reinterpret_cast<float4*>(Xb + p * 4) = make_float4(a0, a1, 0.f, 0.f);
  }
}

// ---- weight gradient: a workgroup walks chunks of up to 64 output pixels of one row (b, to); the chunk's dR tile and its 4-row
// input window are staged in LDS by coalesced loads, thread (tap, n) sums its two input channels over the chunk ----
__global__ __launch_bounds__(256) void dconv1_wgrad_kernel(const float* __restrict__ X, const float* __restrict__ dR, float* __restrict__ dW,
                                                           int B, int T, int F, int To, int Fo, int fchunks) {
  constexpr int FC = 64, XW = 2 * FC + 2;
  __shared__ __attribute__((aligned(16))) float rs[FC * DN];
  __shared__ float2 xs[4][XW];
  const int tid = threadIdx.x, n = tid & 15, tap = tid >> 4, kw = tap & 3, kh = tap >> 2;
  const long nchunk = (long)B * To * fchunks;
  double d0 = 0.0, d1 = 0.0;                           // fp32 inside a chunk (64 terms), fp64 across a workgroup's chunks
  for (long ch = blockIdx.x; ch < nchunk; ch += gridDim.x) {
    float a0 = 0.f, a1 = 0.f;
    const int fcix = (int)(ch % fchunks);
    const long bt = ch / fchunks;
    const int to = (int)(bt % To), b = (int)(bt / To), fo0 = fcix * FC;
    const int nf = Fo - fo0 < FC ? Fo - fo0 : FC;
    __syncthreads();                                   // the previous chunk is consumed
    {
      const int p = tid >> 2, q4 = tid & 3;            // dR tile: 64 pixels x 16 channels = 256 float4
      const float4 v = p < nf ? *reinterpret_cast<const float4*>(dR + (((long)b * To + to) * Fo + fo0 + p) * DN + 4 * q4)
                              : make_float4(0.f, 0.f, 0.f, 0.f);
      *reinterpret_cast<float4*>(&rs[p * DN + 4 * q4]) = v;
    }
    for (int i = tid; i < 4 * XW; i += 256) {          // input window: rows 2 to - 1 .. 2 to + 2, columns 2 fo0 - 1 .. 2 fo0 + 2 FC
      const int r = i / XW, cc = i - r * XW;
      const int t = 2 * to + r - 1, f = 2 * fo0 + cc - 1;
      const bool ok = t >= 0 && t < T && f >= 0 && f < F;
      const int tc = t < 0 ? 0 : (t >= T ? T - 1 : t), fc = f < 0 ? 0 : (f >= F ? F - 1 : f);
      const float2 v = *reinterpret_cast<const float2*>(X + (((long)b * T + tc) * F + fc) * 4);
      xs[r][cc] = make_float2(ok ? v.x : 0.f, ok ? v.y : 0.f);
    }
    __syncthreads();
#pragma unroll 8
    for (int p = 0; p < FC; ++p) {                     // (pixels past nf carry dR = 0)
      const float r = rs[p * DN + n];
      const float2 x = xs[kw][2 * p + kh];
      a0 = fmaf(r, x.x, a0);
      a1 = fmaf(r, x.y, a1);
    }
    d0 += (double)a0;
    d1 += (double)a1;
  }
  // dW[n][c][kh][kw] (accumulated: the caller zero-fills)
  atomicAdd(&dW[(n * 2 + 0) * 16 + kh * 4 + kw], (float)d0);
  atomicAdd(&dW[(n * 2 + 1) * 16 + kh * 4 + kw], (float)d1);
}

// ================================================================================================================================
// Thin convolutions of the generator (round 4): the decoders' last convolutions (models/generator.py:114 `conv_1 = Conv2d(64, 1,
// (1, 2))` of the mask decoder, :128 `conv = Conv2d(64, 2, (1, 2))` of the complex decoder) and the encoder's first one (:39
// `Conv2d(3, 64, (1, 1))`).  As tap GEMMs their 64-column tiles are 94 - 98 % padding (N = 1 / 2 of 64, or K = 3 of 16): 118 - 249 us
// per launch for 0.27 GB of operands.  Direct kernels, 16 lanes per pixel (one float4 of the 64 channels each).
//   x: [rows = B T][F2][64] channels-last; W: PyTorch layouts [n][64][1][2] / [64][3][1][1]; y: [rows][Fo = F2 - 1][4] (channels >= n zero)
// ================================================================================================================================
constexpr int TU = 4;        // pixels in flight per lane group

template <int NO>
__global__ __launch_bounds__(256) void conv1x2_fwd_kernel(const float* __restrict__ X, const float* __restrict__ W, const float* __restrict__ bias,
                                                          float* __restrict__ Y, double* __restrict__ stats, int T, int F2) {
  const int tid = threadIdx.x, l = tid & 15, grp = tid >> 4, b = blockIdx.y;
  const int Fo = F2 - 1;
  const long P = (long)T * Fo;
  float w[NO][2][4], bs[NO];
#pragma unroll
  for (int n = 0; n < NO; ++n) {
    bs[n] = bias[n];
#pragma unroll
    for (int j = 0; j < 4; ++j) { w[n][0][j] = W[(n * 64 + 4 * l + j) * 2]; w[n][1][j] = W[(n * 64 + 4 * l + j) * 2 + 1]; }
  }
  const float* __restrict__ Xb = X + (long)b * T * F2 * 64 + 4 * l;
  float* __restrict__ Yb = Y + (long)b * P * 4;
  float s[NO], q[NO];
#pragma unroll
  for (int n = 0; n < NO; ++n) { s[n] = 0.f; q[n] = 0.f; }
  for (long p0 = ((long)blockIdx.x * 16 + grp) * TU; p0 < P; p0 += (long)gridDim.x * 16 * TU) {
    float4 a[TU], c[TU];
#pragma unroll
    for (int u = 0; u < TU; ++u) {
      const long p = p0 + u < P ? p0 + u : P - 1;
      const int t = (int)(p / Fo), f = (int)(p - (long)t * Fo);
      const float* xp = Xb + ((long)t * F2 + f) * 64;
      a[u] = *reinterpret_cast<const float4*>(xp);
      c[u] = *reinterpret_cast<const float4*>(xp + 64);
    }
#pragma unroll
    for (int u = 0; u < TU; ++u) {
      float y[NO];
#pragma unroll
      for (int n = 0; n < NO; ++n) {
        float v = a[u].x * w[n][0][0] + a[u].y * w[n][0][1] + a[u].z * w[n][0][2] + a[u].w * w[n][0][3];
        v += c[u].x * w[n][1][0] + c[u].y * w[n][1][1] + c[u].z * w[n][1][2] + c[u].w * w[n][1][3];
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) v += __shfl_xor(v, o, 64);
        y[n] = v + bs[n];
      }
      if (l == 0 && p0 + u < P) {
        *reinterpret_cast<float4*>(Yb + (p0 + u) * 4) = make_float4(y[0], NO > 1 ? y[NO > 1 ? 1 : 0] : 0.f, 0.f, 0.f);
#pragma unroll
        for (int n = 0; n < NO; ++n) { s[n] += y[n]; q[n] = fmaf(y[n], y[n], q[n]); }
      }
    }
  }
  if (stats) {               // InstanceNorm sums of the mask decoder: (sum, sum of squares) per (b, channel), fp64 atomics per workgroup
    __shared__ float red[16][NO][2];
    if (l == 0) {
#pragma unroll
      for (int n = 0; n < NO; ++n) { red[grp][n][0] = s[n]; red[grp][n][1] = q[n]; }
    }
    __syncthreads();
    if (tid < NO * 2) {
      const int n = tid >> 1, k = tid & 1;
      double v = 0.0;
      for (int g = 0; g < 16; ++g) v += (double)red[g][n][k];
      atomicAdd(&stats[((long)b * 4 + n) * 2 + k], v);
    }
  }
}

// input gradient: dX[t][f][c] = sum_n dY[t][f][n] W[n][c][0] + dY[t][f - 1][n] W[n][c][1]
template <int NO>
__global__ __launch_bounds__(256) void conv1x2_dgrad_kernel(const float* __restrict__ dY, const float* __restrict__ W, float* __restrict__ dX,
                                                            long rows, int F2) {
  const int tid = threadIdx.x, l = tid & 15, grp = tid >> 4;
  const int Fo = F2 - 1;
  float w[NO][2][4];
#pragma unroll
  for (int n = 0; n < NO; ++n)
#pragma unroll
    for (int j = 0; j < 4; ++j) { w[n][0][j] = W[(n * 64 + 4 * l + j) * 2]; w[n][1][j] = W[(n * 64 + 4 * l + j) * 2 + 1]; }
  const long P = rows * F2;
  for (long p0 = ((long)blockIdx.x * 16 + grp) * TU; p0 < P; p0 += (long)gridDim.x * 16 * TU) {
    float4 d0[TU], d1[TU];
#pragma unroll
    for (int u = 0; u < TU; ++u) {
      const long p = p0 + u < P ? p0 + u : P - 1;
      const long r = p / F2;
      const int f = (int)(p - r * F2);
      const float* dp = dY + (r * Fo + f) * 4;
      d0[u] = f < Fo ? *reinterpret_cast<const float4*>(dp) : make_float4(0.f, 0.f, 0.f, 0.f);
      d1[u] = f >= 1 ? *reinterpret_cast<const float4*>(dp - 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < TU; ++u) {
      const float e0[2] = {d0[u].x, d0[u].y}, e1[2] = {d1[u].x, d1[u].y};
      float o[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float v = 0.f;
#pragma unroll
        for (int n = 0; n < NO; ++n) v += e0[n] * w[n][0][j] + e1[n] * w[n][1][j];
        o[j] = v;
      }
      if (p0 + u < P) *reinterpret_cast<float4*>(dX + (p0 + u) * 64 + 4 * l) = make_float4(o[0], o[1], o[2], o[3]);
    }
  }
}

// weight / bias gradient: dW[n][c][k] += sum_pixels dY[t][f' - k][n] x[t][f'][c] over the INPUT pixels f' (each x line is read once)
template <int NO>
__global__ __launch_bounds__(256) void conv1x2_wgrad_kernel(const float* __restrict__ X, const float* __restrict__ dY, float* __restrict__ dW,
                                                            float* __restrict__ dbias, long rows, int F2) {
  const int tid = threadIdx.x, l = tid & 15, grp = tid >> 4;
  const int Fo = F2 - 1;
  float acc[NO][2][4], ab[NO];
#pragma unroll
  for (int n = 0; n < NO; ++n) {
    ab[n] = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) { acc[n][0][j] = 0.f; acc[n][1][j] = 0.f; }
  }
  const long P = rows * F2;
  for (long p0 = ((long)blockIdx.x * 16 + grp) * TU; p0 < P; p0 += (long)gridDim.x * 16 * TU) {
    float4 x[TU], d0[TU], d1[TU];
#pragma unroll
    for (int u = 0; u < TU; ++u) {
      const bool ok = p0 + u < P;
      const long p = ok ? p0 + u : P - 1;
      const long r = p / F2;
      const int f = (int)(p - r * F2);
      const float* dp = dY + (r * Fo + f) * 4;
      x[u] = *reinterpret_cast<const float4*>(X + p * 64 + 4 * l);
      d0[u] = ok && f < Fo ? *reinterpret_cast<const float4*>(dp) : make_float4(0.f, 0.f, 0.f, 0.f);
      d1[u] = ok && f >= 1 ? *reinterpret_cast<const float4*>(dp - 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < TU; ++u) {
      const float xv[4] = {x[u].x, x[u].y, x[u].z, x[u].w}, e0[2] = {d0[u].x, d0[u].y}, e1[2] = {d1[u].x, d1[u].y};
#pragma unroll
      for (int n = 0; n < NO; ++n) {
        ab[n] += e0[n];
#pragma unroll
        for (int j = 0; j < 4; ++j) { acc[n][0][j] = fmaf(e0[n], xv[j], acc[n][0][j]); acc[n][1][j] = fmaf(e1[n], xv[j], acc[n][1][j]); }
      }
    }
  }
  __shared__ float red[16][NO * 2 * 64 + NO];
#pragma unroll
  for (int n = 0; n < NO; ++n) {
#pragma unroll
    for (int j = 0; j < 4; ++j) { red[grp][(n * 64 + 4 * l + j) * 2] = acc[n][0][j]; red[grp][(n * 64 + 4 * l + j) * 2 + 1] = acc[n][1][j]; }
    if (l == 0) red[grp][NO * 128 + n] = ab[n];
  }
  __syncthreads();
  for (int i = tid; i < NO * 128 + NO; i += 256) {
    float v = 0.f;
    for (int g = 0; g < 16; ++g) v += red[g][i];
    if (i < NO * 128) atomicAdd(&dW[i], v);            // [n][c][1][2]
    else atomicAdd(&dbias[i - NO * 128], v);
  }
}

// encoder head: R[p][n] = b[n] + sum_{c < 3} W[n][c] x[p][c]  (+ InstanceNorm sums); 16 lanes per pixel, 4 output channels each
__global__ __launch_bounds__(256) void conv3to64_fwd_kernel(const float* __restrict__ X, const float* __restrict__ W, const float* __restrict__ bias,
                                                            float* __restrict__ R, double* __restrict__ stats, long P) {
  const int tid = threadIdx.x, l = tid & 15, grp = tid >> 4, b = blockIdx.y;
  float w[4][3], bs[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    bs[j] = bias[4 * l + j];
#pragma unroll
    for (int c = 0; c < 3; ++c) w[j][c] = W[(4 * l + j) * 3 + c];
  }
  const float* __restrict__ Xb = X + (long)b * P * 4;
  float* __restrict__ Rb = R + (long)b * P * 64 + 4 * l;
  float s[4] = {0.f, 0.f, 0.f, 0.f}, q[4] = {0.f, 0.f, 0.f, 0.f};
  for (long p0 = ((long)blockIdx.x * 16 + grp) * TU; p0 < P; p0 += (long)gridDim.x * 16 * TU) {
    float4 x[TU];
#pragma unroll
    for (int u = 0; u < TU; ++u) x[u] = *reinterpret_cast<const float4*>(Xb + (p0 + u < P ? p0 + u : P - 1) * 4);
#pragma unroll
    for (int u = 0; u < TU; ++u) {
      float o[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = fmaf(w[j][2], x[u].z, fmaf(w[j][1], x[u].y, fmaf(w[j][0], x[u].x, bs[j])));
      if (p0 + u < P) {
        *reinterpret_cast<float4*>(Rb + (p0 + u) * 64) = make_float4(o[0], o[1], o[2], o[3]);
#pragma unroll
        for (int j = 0; j < 4; ++j) { s[j] += o[j]; q[j] = fmaf(o[j], o[j], q[j]); }
      }
    }
  }
  if (stats) {
    __shared__ float red[16][64][2];
#pragma unroll
    for (int j = 0; j < 4; ++j) { red[grp][4 * l + j][0] = s[j]; red[grp][4 * l + j][1] = q[j]; }
    __syncthreads();
    if (tid < 128) {
      const int n = tid >> 1, k = tid & 1;
      double v = 0.0;
      for (int g = 0; g < 16; ++g) v += (double)red[g][n][k];
      atomicAdd(&stats[((long)b * 64 + n) * 2 + k], v);
    }
  }
}

// dW[n][c] += sum_p dR[p][n] x[p][c] (c < 3), dbias[n] += sum_p dR[p][n]
__global__ __launch_bounds__(256) void conv3to64_wgrad_kernel(const float* __restrict__ X, const float* __restrict__ dR, float* __restrict__ dW,
                                                              float* __restrict__ dbias, long P) {
  const int tid = threadIdx.x, l = tid & 15, grp = tid >> 4;
  float acc[4][4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[j][c] = 0.f;
  for (long p0 = ((long)blockIdx.x * 16 + grp) * TU; p0 < P; p0 += (long)gridDim.x * 16 * TU) {
    float4 x[TU], d[TU];
#pragma unroll
    for (int u = 0; u < TU; ++u) {
      const bool ok = p0 + u < P;
      const long p = ok ? p0 + u : P - 1;
      x[u] = *reinterpret_cast<const float4*>(X + p * 4);
      d[u] = ok ? *reinterpret_cast<const float4*>(dR + p * 64 + 4 * l) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < TU; ++u) {
      const float dv[4] = {d[u].x, d[u].y, d[u].z, d[u].w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc[j][0] = fmaf(dv[j], x[u].x, acc[j][0]);
        acc[j][1] = fmaf(dv[j], x[u].y, acc[j][1]);
        acc[j][2] = fmaf(dv[j], x[u].z, acc[j][2]);
        acc[j][3] += dv[j];
      }
    }
  }
  __shared__ float red[16][64][4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int c = 0; c < 4; ++c) red[grp][4 * l + j][c] = acc[j][c];
  __syncthreads();
  {
    const int n = tid >> 2, c = tid & 3;
    float v = 0.f;
    for (int g = 0; g < 16; ++g) v += red[g][n][c];
    if (c < 3) atomicAdd(&dW[n * 3 + c], v);
    else atomicAdd(&dbias[n], v);
  }
}

// ================================================================================================================================
// Input gradient of the discriminator's 4x4 stride-2 padding-1 convolutions (models/discriminator.py:42-50, stages 2 - 4) by PARITY
// CLASS.  dx[t][f] = sum over the taps (kw, kh) with t + 1 - kw and f + 1 - kh even: 4 of the 16 taps, the same 4 for every pixel of
// a class (t mod 2, f mod 2).  The tap GEMM in `up` mode walks all 16 taps for every 128-pixel tile and zero-fills 12 of them per row
// (148 / 92 / 148 us per launch for 1 GFLOP of real work).  Here a workgroup owns 128 pixels of ONE class: K = 4 taps x N, no zero
// rows, output scattered to the class's pixels.  fp32 MFMA (32x32x2), LDS tiles as in gemm_tap_kernel (row-major, +4 pad).
//   dR [B][To][Fo][N], Wd [Cin][16 taps = kh * 4 + kw][N] (gemm.pack_conv_dgrad), dX [B][Ti][Fi][Cin]
// ================================================================================================================================
typedef float f32x16t __attribute__((ext_vector_type(16)));

template <int BN>
__global__ __launch_bounds__(256) void dconv_dgrad_cls_kernel(const float* __restrict__ dR, const float* __restrict__ Wd, float* __restrict__ dX,
                                                              int To, int Fo, int Ti, int Fi, int N, int Cin) {
  constexpr int BM = 128, BK = 32, SA = BK + 4, KQ = BK / 4, RPP = 256 / KQ, NA = BM / RPP, NB = (BN + RPP - 1) / RPP;
  __shared__ __attribute__((aligned(16))) float As[BM * SA];
  __shared__ __attribute__((aligned(16))) float Bs[BN * SA];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cls = blockIdx.y, pt = cls >> 1, pf = cls & 1, b = blockIdx.z;
  const int Tc = (Ti - pt + 1) >> 1, Fc = (Fi - pf + 1) >> 1, Mc = Tc * Fc;
  const int m0 = blockIdx.x * BM, cb = 0;
  if (m0 >= Mc) return;
  const int kq = tid % KQ, r0 = tid / KQ;
  const float* __restrict__ Rb = dR + (long)b * To * Fo * N;
  const int ldw = 16 * N;
  // the class's taps and their source offsets: kw = kw0, kw0 + 2 with to = t' + dto; the same along f
  const int kw0 = (pt + 1) & 1, kh0 = (pf + 1) & 1;
  const int dto0 = (pt + 1 - kw0) >> 1, dfo0 = (pf + 1 - kh0) >> 1;          // (second tap of a pair: one less)
  int rt[NA], rf[NA];
  bool rok[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int m = m0 + r0 + i * RPP;
    rok[i] = m < Mc;
    rt[i] = m / Fc; rf[i] = m - rt[i] * Fc;
  }
  const int nchunk = N / BK, NI = 4 * nchunk;            // N % 32 == 0 (host-checked)
  float4 ra[NA], rb[NB];
  auto load_tiles = [&](int it) {
    const int chunk = it >> 2, j = it & 3, jw = j & 1, jh = j >> 1;    // channel chunk outer, the 4 taps inner
    const int tap = (kh0 + 2 * jh) * 4 + kw0 + 2 * jw;
    const int c = chunk * BK + kq * 4;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int to = rt[i] + dto0 - jw, fo = rf[i] + dfo0 - jh;
      const bool ok = rok[i] && to >= 0 && to < To && fo >= 0 && fo < Fo;
      ra[i] = ok ? *reinterpret_cast<const float4*>(Rb + ((unsigned)(to * Fo + fo) * (unsigned)N + (unsigned)c)) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int n = cb * BN + r0 + i * RPP;
      rb[i] = (n < Cin && r0 + i * RPP < BN) ? *reinterpret_cast<const float4*>(Wd + ((unsigned)n * (unsigned)ldw + (unsigned)(tap * N + c)))
                                             : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  f32x16t acc0, acc1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
  load_tiles(0);
  const float* Ap = &As[(wave * 32 + (lane & 31)) * SA + (lane >> 5) * (BK / 2)];
  const float* Bp0 = &Bs[(lane & 31) * SA + (lane >> 5) * (BK / 2)];
  const float* Bp1 = Bp0 + 32 * SA;
  for (int it = 0; it < NI; ++it) {
#pragma unroll
    for (int i = 0; i < NA; ++i) *reinterpret_cast<float4*>(&As[(r0 + i * RPP) * SA + kq * 4]) = ra[i];
#pragma unroll
    for (int i = 0; i < NB; ++i)
      if (r0 + i * RPP < BN) *reinterpret_cast<float4*>(&Bs[(r0 + i * RPP) * SA + kq * 4]) = rb[i];
    __syncthreads();
    if (it + 1 < NI) load_tiles(it + 1);
#pragma unroll
    for (int s4 = 0; s4 < BK / 2; s4 += 4) {
      const float4 a = *reinterpret_cast<const float4*>(Ap + s4);
      const float4 b0 = *reinterpret_cast<const float4*>(Bp0 + s4);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b0.x, acc0, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b0.y, acc0, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b0.z, acc0, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b0.w, acc0, 0, 0, 0);
      if (BN > 32) {
        const float4 b1 = *reinterpret_cast<const float4*>(Bp1 + s4);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b1.x, acc1, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b1.y, acc1, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b1.z, acc1, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b1.w, acc1, 0, 0, 0);
      }
    }
    __syncthreads();
  }
  // C layout: lane (col = lane & 31, half = lane >> 5) holds rows (r & 3) + 8 (r >> 2) + 4 half: 32 lanes store 128 contiguous bytes
  const int col = lane & 31, half = lane >> 5;
  float* __restrict__ Xb = dX + (long)b * Ti * Fi * Cin;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = m0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
    if (m < Mc) {
      const int tq = m / Fc, fq = m - tq * Fc;
      float* o = Xb + ((long)(2 * tq + pt) * Fi + 2 * fq + pf) * Cin;
      if (col < Cin) o[col] = acc0[r];
      if (BN > 32 && col + 32 < Cin) o[col + 32] = acc1[r];
    }
  }
}

static int thin_grid(long pixels, long cap) {
  long nb = (pixels + 16 * TU - 1) / (16 * TU);
  if (nb > cap) nb = cap;
  return (int)(nb < 1 ? 1 : nb);
}
}  // namespace

extern "C" int se_dconv1_fwd(const float* X, const float* W, float* R, double* stats, int B, int T, int F, int N, void* stream) {
  SE_REQUIRE(X && W && R && B > 0 && T > 1 && F > 1 && N == DN, "dconv1_fwd: bad arguments (N = %d: built for ndf = 16)", N);
  const int To = (T + 2 - 4) / 2 + 1, Fo = (F + 2 - 4) / 2 + 1;
  const long P = (long)To * Fo;
  long nb = (P + 64 * 4 - 1) / (64 * 4);               // ~4 pixel groups per workgroup: the 2 KB weight table is staged once per workgroup
  if (nb > 2048) nb = 2048;
  hipLaunchKernelGGL(dconv1_fwd_kernel, dim3((unsigned)nb, B), dim3(256), 0, as_stream(stream), X, W, R, stats, T, F, To, Fo);
  return se_check_launch("se_dconv1_fwd");
}

extern "C" int se_dconv1_dgrad(const float* dR, const float* W, float* dX, int B, int T, int F, int N, void* stream) {
  SE_REQUIRE(dR && W && dX && B > 0 && T > 1 && F > 1 && N == DN, "dconv1_dgrad: bad arguments (N = %d: built for ndf = 16)", N);
  const int To = (T + 2 - 4) / 2 + 1, Fo = (F + 2 - 4) / 2 + 1;
  long nb = ((long)T * F + 255) / 256;
  if (nb > 1024) nb = 1024;
  hipLaunchKernelGGL(dconv1_dgrad_kernel, dim3((unsigned)nb, B), dim3(256), 0, as_stream(stream), dR, W, dX, T, F, To, Fo);
  return se_check_launch("se_dconv1_dgrad");
}

extern "C" int se_dconv1_wgrad(const float* X, const float* dR, float* dW, int B, int T, int F, int N, void* stream) {
  SE_REQUIRE(X && dR && dW && B > 0 && T > 1 && F > 1 && N == DN, "dconv1_wgrad: bad arguments (N = %d: built for ndf = 16)", N);
  const int To = (T + 2 - 4) / 2 + 1, Fo = (F + 2 - 4) / 2 + 1;
  const int fchunks = (Fo + 63) / 64;
  long nwg = (long)B * To * fchunks;                   // chunks; at most 512 workgroups (512 x 512 atomics at the end)
  if (nwg > 512) nwg = 512;
  hipLaunchKernelGGL(dconv1_wgrad_kernel, dim3((unsigned)nwg), dim3(256), 0, as_stream(stream), X, dR, dW, B, T, F, To, Fo, fchunks);
  return se_check_launch("se_dconv1_wgrad");
}

extern "C" int se_conv1x2_fwd(const float* X, const float* W, const float* bias, float* Y, double* stats, int B, int T, int F2, int n,
                              void* stream) {
  SE_REQUIRE(X && W && bias && Y && B > 0 && T > 0 && F2 > 1 && (n == 1 || n == 2), "conv1x2_fwd: bad arguments (n = %d: 1 or 2)", n);
  const dim3 grid(thin_grid((long)T * (F2 - 1), 2048 / B > 1 ? 2048 / B : 1), B);
  if (n == 1) hipLaunchKernelGGL(conv1x2_fwd_kernel<1>, grid, dim3(256), 0, as_stream(stream), X, W, bias, Y, stats, T, F2);
  else hipLaunchKernelGGL(conv1x2_fwd_kernel<2>, grid, dim3(256), 0, as_stream(stream), X, W, bias, Y, stats, T, F2);
  return se_check_launch("se_conv1x2_fwd");
}

extern "C" int se_conv1x2_dgrad(const float* dY, const float* W, float* dX, long rows, int F2, int n, void* stream) {
  SE_REQUIRE(dY && W && dX && rows > 0 && F2 > 1 && (n == 1 || n == 2), "conv1x2_dgrad: bad arguments (n = %d: 1 or 2)", n);
  const dim3 grid(thin_grid(rows * F2, 4096));
  if (n == 1) hipLaunchKernelGGL(conv1x2_dgrad_kernel<1>, grid, dim3(256), 0, as_stream(stream), dY, W, dX, rows, F2);
  else hipLaunchKernelGGL(conv1x2_dgrad_kernel<2>, grid, dim3(256), 0, as_stream(stream), dY, W, dX, rows, F2);
  return se_check_launch("se_conv1x2_dgrad");
}

extern "C" int se_conv1x2_wgrad(const float* X, const float* dY, float* dW, float* dbias, long rows, int F2, int n, void* stream) {
  SE_REQUIRE(X && dY && dW && dbias && rows > 0 && F2 > 1 && (n == 1 || n == 2), "conv1x2_wgrad: bad arguments (n = %d: 1 or 2)", n);
  const dim3 grid(thin_grid(rows * F2, 1024));
  if (n == 1) hipLaunchKernelGGL(conv1x2_wgrad_kernel<1>, grid, dim3(256), 0, as_stream(stream), X, dY, dW, dbias, rows, F2);
  else hipLaunchKernelGGL(conv1x2_wgrad_kernel<2>, grid, dim3(256), 0, as_stream(stream), X, dY, dW, dbias, rows, F2);
  return se_check_launch("se_conv1x2_wgrad");
}

extern "C" int se_conv3to64_fwd(const float* X, const float* W, const float* bias, float* R, double* stats, int B, long P, void* stream) {
  SE_REQUIRE(X && W && bias && R && B > 0 && P > 0, "conv3to64_fwd: bad arguments");
  const dim3 grid(thin_grid(P, 2048 / B > 1 ? 2048 / B : 1), B);
  hipLaunchKernelGGL(conv3to64_fwd_kernel, grid, dim3(256), 0, as_stream(stream), X, W, bias, R, stats, P);
  return se_check_launch("se_conv3to64_fwd");
}

extern "C" int se_conv3to64_wgrad(const float* X, const float* dR, float* dW, float* dbias, long npix, void* stream) {
  SE_REQUIRE(X && dR && dW && dbias && npix > 0, "conv3to64_wgrad: bad arguments");
  hipLaunchKernelGGL(conv3to64_wgrad_kernel, dim3(thin_grid(npix, 1024)), dim3(256), 0, as_stream(stream), X, dR, dW, dbias, npix);
  return se_check_launch("se_conv3to64_wgrad");
}

extern "C" int se_dconv_dgrad(const float* dR, const float* Wd, float* dX, int B, int Ti, int Fi, int N, int Cin, void* stream) {
  SE_REQUIRE(dR && Wd && dX && B > 0 && Ti > 1 && Fi > 1 && N >= 32 && (N % 32) == 0 && Cin >= 4 && Cin <= 64 && (Cin % 4) == 0,
             "dconv_dgrad: bad arguments (N = %d: a multiple of 32; Cin = %d: <= 64)", N, Cin);
  const int To = (Ti + 2 - 4) / 2 + 1, Fo = (Fi + 2 - 4) / 2 + 1;
  SE_REQUIRE((long)To * Fo * N < (1L << 31) && (long)Ti * Fi * Cin < (1L << 31), "dconv_dgrad: a batch entry must stay below 2^31 elements");
  const int Tc = (Ti + 1) / 2, Fc = (Fi + 1) / 2;            // the largest class
  const dim3 grid((unsigned)((Tc * Fc + 127) / 128), 4, (unsigned)B);
  if (Cin <= 32) hipLaunchKernelGGL(dconv_dgrad_cls_kernel<32>, grid, dim3(256), 0, as_stream(stream), dR, Wd, dX, To, Fo, Ti, Fi, N, Cin);
  else hipLaunchKernelGGL(dconv_dgrad_cls_kernel<64>, grid, dim3(256), 0, as_stream(stream), dR, Wd, dX, To, Fo, Ti, Fi, N, Cin);
  return se_check_launch("se_dconv_dgrad");
}
