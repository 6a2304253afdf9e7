// Depthwise 31-tap convolution along the sequence axis of the Conformer conv module
// (DepthWiseConv1d, models/conformer.py:40-48,166: zero pad (15,15), cross-correlation, groups = 128) with the
// BatchNorm1d batch statistics (conformer.py:167) produced in the same pass.  HBM-bound: each input row is
// read once per 64-position tile (+30 halo rows) into LDS, each output written once; lanes own float4 channel
// groups (a 128-channel row = 512 contiguous bytes per 32 lanes) and 8 consecutive positions, the 31 taps of
// their 4 channels live in registers.
//
// The same kernel with flipped taps is the input gradient; the weight gradient is a persistent kernel that
// keeps 31 x float4 accumulators per lane and flushes once per workgroup.
#include "se_common.h"

struct SeqGeom {
  int nseq, n, inner;
  long outer_stride, inner_stride, pos_stride;   // tokens
};
static __device__ __forceinline__ long tok_of(const SeqGeom& g, int s, int p) {
  return (long)(s / g.inner) * g.outer_stride + (long)(s % g.inner) * g.inner_stride + (long)p * g.pos_stride;
}

constexpr int DW_C = 128, DW_K = 31, DW_TILE = 64, DW_ROWS = DW_TILE + DW_K - 1;   // 94

struct DwArgs {
  SeqGeom g;
  const float* X; const float* W; const float* bias; float* Y; double* stats; int flip;
};

__global__ __launch_bounds__(256) void dwconv_kernel(DwArgs a) {
  __shared__ __attribute__((aligned(16))) float xs[DW_ROWS * DW_C];
  const int tid = threadIdx.x, cl = tid & 31, ps = tid >> 5;
  const int n = a.g.n;
  // taps of this lane's 4 channels: the 128 x 31 table is read ONCE per (persistent) workgroup with coalesced
  // loads into LDS and picked up as float4 channel groups (per-thread strided global loads of the table -- 124
  // scalar loads touching 32 cache lines each -- made the first version of this kernel 5x slower than HBM).
  for (int i = tid; i < DW_C * DW_K; i += 256) {
    int ch = i / DW_K, k = i - ch * DW_K;
    xs[(a.flip ? DW_K - 1 - k : k) * DW_C + ch] = a.W[i];
  }
  __syncthreads();
  float4 w[DW_K];
#pragma unroll
  for (int k = 0; k < DW_K; ++k) w[k] = *reinterpret_cast<const float4*>(&xs[k * DW_C + cl * 4]);
  float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
  if (a.bias) bv = *reinterpret_cast<const float4*>(a.bias + cl * 4);
  float s[4] = {0, 0, 0, 0}, q2[4] = {0, 0, 0, 0};
  const int tiles = (n + DW_TILE - 1) / DW_TILE;
  const long nitems = (long)a.g.nseq * tiles;
  for (long it = blockIdx.x; it < nitems; it += gridDim.x) {
    const int seq = (int)(it / tiles), p0 = (int)(it - (long)seq * tiles) * DW_TILE;
    const long base = (long)(seq / a.g.inner) * a.g.outer_stride + (long)(seq % a.g.inner) * a.g.inner_stride;
    const float* __restrict__ Xb = a.X + base * DW_C;
    float* __restrict__ Yb = a.Y + base * DW_C;
    const long rs = a.g.pos_stride * DW_C;      // floats between consecutive positions
    __syncthreads();                           // previous tile (or the weight table) fully consumed
    for (int i = tid; i < DW_ROWS * 32; i += 256) {
      int row = i >> 5, q = i & 31;
      int p = p0 - 15 + row;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (p >= 0 && p < n) v = *reinterpret_cast<const float4*>(Xb + (long)p * rs + q * 4);
      *reinterpret_cast<float4*>(&xs[row * DW_C + q * 4]) = v;
    }
    __syncthreads();
    float4 acc[8];
#pragma unroll
    for (int o = 0; o < 8; ++o) acc[o] = bv;
#pragma unroll
    for (int i = 0; i < 8 + DW_K - 1; ++i) {
      float4 x = *reinterpret_cast<const float4*>(&xs[(ps * 8 + i) * DW_C + cl * 4]);
#pragma unroll
      for (int o = 0; o < 8; ++o) {
        const int k = i - o;
        if (k >= 0 && k < DW_K) {
          acc[o].x += x.x * w[k].x; acc[o].y += x.y * w[k].y; acc[o].z += x.z * w[k].z; acc[o].w += x.w * w[k].w;
        }
      }
    }
#pragma unroll
    for (int o = 0; o < 8; ++o) {
      int p = p0 + ps * 8 + o;
      if (p < n) {
        *reinterpret_cast<float4*>(Yb + (long)p * rs + cl * 4) = acc[o];
        s[0] += acc[o].x; s[1] += acc[o].y; s[2] += acc[o].z; s[3] += acc[o].w;
        q2[0] += acc[o].x * acc[o].x; q2[1] += acc[o].y * acc[o].y; q2[2] += acc[o].z * acc[o].z; q2[3] += acc[o].w * acc[o].w;
      }
    }
  }
  if (a.stats) {            // one fp64 atomic per (channel, moment) per workgroup
    __syncthreads();
    float* red = xs;            // [8 slots][128][2]
#pragma unroll
    for (int j = 0; j < 4; ++j) { red[(ps * DW_C + cl * 4 + j) * 2] = s[j]; red[(ps * DW_C + cl * 4 + j) * 2 + 1] = q2[j]; }
    __syncthreads();
    float t = 0.f;              // tid -> (channel tid>>1, which tid&1)
#pragma unroll
    for (int sl = 0; sl < 8; ++sl) t += red[(sl * DW_C) * 2 + tid];
    atomicAdd(&a.stats[tid], (double)t);
  }
}

struct DwWgradArgs {
  SeqGeom g;
  const float* X; const float* dY; float* dW; float* dbias;
};

__global__ __launch_bounds__(256) void dwconv_wgrad_kernel(DwWgradArgs a) {
  __shared__ __attribute__((aligned(16))) float xs[DW_ROWS * DW_C];
  __shared__ __attribute__((aligned(16))) float ys[DW_TILE * DW_C];
  const int tid = threadIdx.x, cl = tid & 31, ps = tid >> 5;
  const int n = a.g.n;
  const int tiles = (n + DW_TILE - 1) / DW_TILE;
  const long nitems = (long)a.g.nseq * tiles;
  float4 acc[DW_K];
#pragma unroll
  for (int k = 0; k < DW_K; ++k) acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 bacc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (long it = blockIdx.x; it < nitems; it += gridDim.x) {
    const int seq = (int)(it / tiles), p0 = (int)(it % tiles) * DW_TILE;
    const long base = (long)(seq / a.g.inner) * a.g.outer_stride + (long)(seq % a.g.inner) * a.g.inner_stride;
    const float* __restrict__ Xb = a.X + base * DW_C;
    const float* __restrict__ Gb = a.dY + base * DW_C;
    const long rs = a.g.pos_stride * DW_C;
    __syncthreads();
    for (int i = tid; i < DW_ROWS * 32; i += 256) {
      int row = i >> 5, q = i & 31;
      int p = p0 - 15 + row;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (p >= 0 && p < n) v = *reinterpret_cast<const float4*>(Xb + (long)p * rs + q * 4);
      *reinterpret_cast<float4*>(&xs[row * DW_C + q * 4]) = v;
    }
    for (int i = tid; i < DW_TILE * 32; i += 256) {
      int row = i >> 5, q = i & 31;
      int p = p0 + row;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (p < n) v = *reinterpret_cast<const float4*>(Gb + (long)p * rs + q * 4);
      *reinterpret_cast<float4*>(&ys[row * DW_C + q * 4]) = v;
    }
    __syncthreads();
    float4 dy[8];
#pragma unroll
    for (int o = 0; o < 8; ++o) {
      dy[o] = *reinterpret_cast<const float4*>(&ys[(ps * 8 + o) * DW_C + cl * 4]);
      bacc.x += dy[o].x; bacc.y += dy[o].y; bacc.z += dy[o].z; bacc.w += dy[o].w;
    }
#pragma unroll
    for (int i = 0; i < 8 + DW_K - 1; ++i) {
      float4 x = *reinterpret_cast<const float4*>(&xs[(ps * 8 + i) * DW_C + cl * 4]);
#pragma unroll
      for (int o = 0; o < 8; ++o) {
        const int k = i - o;
        if (k >= 0 && k < DW_K) {
          acc[k].x += dy[o].x * x.x; acc[k].y += dy[o].y * x.y; acc[k].z += dy[o].z * x.z; acc[k].w += dy[o].w * x.w;
        }
      }
    }
  }
  // reduce the 8 position slots through LDS, 8 taps at a time, then one atomic per (channel, tap) per workgroup
  float* red = xs;     // [8 slots][8 taps][128]
#pragma unroll
  for (int k0 = 0; k0 < 32; k0 += 8) {
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      const int k = k0 + kk;
      float4 v = k < DW_K ? acc[k < DW_K ? k : 0] : bacc;      // slot k == 31 carries the bias gradient
      *reinterpret_cast<float4*>(&red[((ps * 8 + kk) * DW_C) + cl * 4]) = v;
    }
    __syncthreads();
    for (int i = tid; i < 8 * DW_C; i += 256) {
      int kk = i >> 7, ch = i & 127;
      float t = 0.f;
#pragma unroll
      for (int sl = 0; sl < 8; ++sl) t += red[(sl * 8 + kk) * DW_C + ch];
      int k = k0 + kk;
      if (k < DW_K) atomicAdd(&a.dW[ch * DW_K + k], t);
      else if (a.dbias) atomicAdd(&a.dbias[ch], t);
    }
  }
}

extern "C" int se_dwconv31(const float* X, const float* W, const float* bias, float* Y, double* stats, int flip,
                           int nseq, int n, int inner, long outer_stride, long inner_stride, long pos_stride,
                           void* stream) {
  SE_REQUIRE(X && W && Y && nseq > 0 && n > 0 && inner > 0, "dwconv31: bad arguments");
  DwArgs a{{nseq, n, inner, outer_stride, inner_stride, pos_stride}, X, W, bias, Y, stats, flip};
  long nitems = (long)nseq * cdiv(n, DW_TILE);
  int nblk = nitems < 768 ? (int)nitems : 768;          // persistent: 3 workgroups (48 KB LDS each) per CU
  hipLaunchKernelGGL(dwconv_kernel, dim3(nblk), dim3(256), 0, as_stream(stream), a);
  return se_check_launch("se_dwconv31");
}

extern "C" int se_dwconv31_wgrad(const float* X, const float* dY, float* dW, float* dbias, int nseq, int n,
                                 int inner, long outer_stride, long inner_stride, long pos_stride, void* stream) {
  SE_REQUIRE(X && dY && dW && nseq > 0 && n > 0 && inner > 0, "dwconv31_wgrad: bad arguments");
  DwWgradArgs a{{nseq, n, inner, outer_stride, inner_stride, pos_stride}, X, dY, dW, dbias};
  long nitems = (long)nseq * cdiv(n, DW_TILE);
  int nblk = nitems < 512 ? (int)nitems : 512;
  hipLaunchKernelGGL(dwconv_wgrad_kernel, dim3(nblk), dim3(256), 0, as_stream(stream), a);
  return se_check_launch("se_dwconv31_wgrad");
}
