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

// 512 threads: lane pair-channel cl = tid & 63 (2 channels), position slot ps = tid >> 6 (8 slots x 8 positions).
// Two channels per lane (instead of four) keep the 31 taps + 8 accumulators + the next tile's prefetch registers
// under ~100 VGPRs, so 3 workgroups (24 waves) stay resident per CU and the next tile's global loads are in flight
// while the current tile is computed.
__global__ __launch_bounds__(512) void dwconv_kernel(DwArgs a) {
  __shared__ __attribute__((aligned(16))) float xs[DW_ROWS * DW_C];
  const int tid = threadIdx.x, cl = tid & 63, ps = tid >> 6;
  const int n = a.g.n;
  for (int i = tid; i < DW_C * DW_K; i += 512) {
    int ch = i / DW_K, k = i - ch * DW_K;
    xs[(a.flip ? DW_K - 1 - k : k) * DW_C + ch] = a.W[i];
  }
  __syncthreads();
  float2 w[DW_K];
#pragma unroll
  for (int k = 0; k < DW_K; ++k) w[k] = *reinterpret_cast<const float2*>(&xs[k * DW_C + cl * 2]);
  float2 bv = make_float2(0.f, 0.f);
  if (a.bias) bv = *reinterpret_cast<const float2*>(a.bias + cl * 2);
  float s[2] = {0, 0}, q2[2] = {0, 0};
  const int tiles = (n + DW_TILE - 1) / DW_TILE;
  const long nitems = (long)a.g.nseq * tiles;
  constexpr int NPRE = (DW_ROWS * 32 + 511) / 512;      // float4 per thread per tile (6)
  float4 pre[NPRE];
  auto fetch = [&](long it) {
    const int seq = (int)(it / tiles), p0 = (int)(it - (long)seq * tiles) * DW_TILE;
    const long base = (long)(seq / a.g.inner) * a.g.outer_stride + (long)(seq % a.g.inner) * a.g.inner_stride;
    const float* __restrict__ Xb = a.X + base * DW_C;
    const long rs = a.g.pos_stride * DW_C;
#pragma unroll
    for (int k = 0; k < NPRE; ++k) {
      int i = tid + k * 512;
      int row = i >> 5, q = i & 31;
      int p = p0 - 15 + row;
      pre[k] = (row < DW_ROWS && p >= 0 && p < n) ? *reinterpret_cast<const float4*>(Xb + (long)p * rs + q * 4)
                                                  : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  // XCD-aware item order: workgroups b and b + 8 share an XCD (round-robin dispatch), so XCD x walks its own contiguous
  // eighth of the (sequence, tile) items with its gridDim / 8 workgroups side by side -- consecutive tiles of a sequence are
  // in flight on the same XCD at the same time and their 30 shared halo rows are served by that XCD's L2 instead of HBM
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, S = gridDim.x >> 3;
  const long Q = (nitems + 7) >> 3, ibase = (long)xcd * Q, iend = ibase + Q < nitems ? ibase + Q : nitems;
  long it = ibase + slot;
  if (it < iend) fetch(it);
  for (; it < iend; it += S) {
    const int seq = (int)(it / tiles), p0 = (int)(it - (long)seq * tiles) * DW_TILE;
    const long base = (long)(seq / a.g.inner) * a.g.outer_stride + (long)(seq % a.g.inner) * a.g.inner_stride;
    float* __restrict__ Yb = a.Y + base * DW_C;
    const long rs = a.g.pos_stride * DW_C;
    __syncthreads();                           // previous tile (or the tap table) fully consumed
#pragma unroll
    for (int k = 0; k < NPRE; ++k) {
      int i = tid + k * 512;
      if (i < DW_ROWS * 32) *reinterpret_cast<float4*>(&xs[(i >> 5) * DW_C + (i & 31) * 4]) = pre[k];
    }
    __syncthreads();
    if (it + S < iend) fetch(it + S);
    float2 acc[8];
#pragma unroll
    for (int o = 0; o < 8; ++o) acc[o] = bv;
#pragma unroll
    for (int i = 0; i < 8 + DW_K - 1; ++i) {
      float2 x = *reinterpret_cast<const float2*>(&xs[(ps * 8 + i) * DW_C + cl * 2]);
#pragma unroll
      for (int o = 0; o < 8; ++o) {
        const int k = i - o;
        if (k >= 0 && k < DW_K) { acc[o].x += x.x * w[k].x; acc[o].y += x.y * w[k].y; }
      }
    }
#pragma unroll
    for (int o = 0; o < 8; ++o) {
      int p = p0 + ps * 8 + o;
      if (p < n) {
        *reinterpret_cast<float2*>(Yb + (long)p * rs + cl * 2) = acc[o];
        s[0] += acc[o].x; s[1] += acc[o].y;
        q2[0] += acc[o].x * acc[o].x; q2[1] += acc[o].y * acc[o].y;
      }
    }
  }
  if (a.stats) {            // one fp64 atomic per (channel, moment) per workgroup
    __syncthreads();
    float* red = xs;            // [8 slots][128][2]
#pragma unroll
    for (int j = 0; j < 2; ++j) { red[(ps * DW_C + cl * 2 + j) * 2] = s[j]; red[(ps * DW_C + cl * 2 + j) * 2 + 1] = q2[j]; }
    __syncthreads();
    if (tid < 256) {
      float t = 0.f;              // tid -> (channel tid>>1, which tid&1)
#pragma unroll
      for (int sl = 0; sl < 8; ++sl) t += red[(sl * DW_C) * 2 + tid];
      atomicAdd(&a.stats[tid], (double)t);
    }
  }
}

struct DwWgradArgs {
  SeqGeom g;
  const float* X; const float* dY; float* dW; float* dbias;
  float* part;      // workspace [workgroups][32 taps (31 + bias)][128 channels]: per-workgroup partial sums
};

// weight gradient, same thread layout (512 threads, 2 channels per lane, 8 slots x 8 positions): 31 float2
// accumulators per lane, persistent over tiles, one flush per workgroup.
__global__ __launch_bounds__(512, 2) void dwconv_wgrad_kernel(DwWgradArgs a) {
  __shared__ __attribute__((aligned(16))) float xs[DW_ROWS * DW_C];
  __shared__ __attribute__((aligned(16))) float ys[DW_TILE * DW_C];
  const int tid = threadIdx.x, cl = tid & 63, ps = tid >> 6;
  const int n = a.g.n;
  const int tiles = (n + DW_TILE - 1) / DW_TILE;
  const long nitems = (long)a.g.nseq * tiles;
  // explicit 2-wide vectors: v_pk_fma_f32 does both channels of a lane in one instruction (the scalar float2 form
  // compiled to separate multiplies, packed adds and ~2 register moves per pair: 4.7x the VALU instructions)
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  f32x2 acc[DW_K];
#pragma unroll
  for (int k = 0; k < DW_K; ++k) acc[k] = (f32x2){0.f, 0.f};
  f32x2 bacc = {0.f, 0.f};
  constexpr int NPX = (DW_ROWS * 32 + 511) / 512, NPY = DW_TILE * 32 / 512;     // 6, 4
  float4 prx[NPX], pry[NPY];
  auto fetch = [&](long it) {
    const int seq = (int)(it / tiles), p0 = (int)(it % tiles) * DW_TILE;
    const long base = (long)(seq / a.g.inner) * a.g.outer_stride + (long)(seq % a.g.inner) * a.g.inner_stride;
    const float* __restrict__ Xb = a.X + base * DW_C;
    const float* __restrict__ Gb = a.dY + base * DW_C;
    const long rs = a.g.pos_stride * DW_C;
#pragma unroll
    for (int k = 0; k < NPX; ++k) {
      int i = tid + k * 512, row = i >> 5, q = i & 31, p = p0 - 15 + row;
      prx[k] = (row < DW_ROWS && p >= 0 && p < n) ? *reinterpret_cast<const float4*>(Xb + (long)p * rs + q * 4)
                                                  : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int k = 0; k < NPY; ++k) {
      int i = tid + k * 512, row = i >> 5, q = i & 31, p = p0 + row;
      pry[k] = (p < n) ? *reinterpret_cast<const float4*>(Gb + (long)p * rs + q * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  // no cross-tile register prefetch: without its 40 VGPRs the kernel fits 4 waves per SIMD, i.e. TWO of these
  // 512-thread workgroups per CU (2 x 79 KB of LDS), and the other workgroup's compute covers this one's loads
  long it = blockIdx.x;
  for (; it < nitems; it += gridDim.x) {
    fetch(it);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NPX; ++k) {
      int i = tid + k * 512;
      if (i < DW_ROWS * 32) *reinterpret_cast<float4*>(&xs[(i >> 5) * DW_C + (i & 31) * 4]) = prx[k];
    }
#pragma unroll
    for (int k = 0; k < NPY; ++k) {
      int i = tid + k * 512;
      *reinterpret_cast<float4*>(&ys[(i >> 5) * DW_C + (i & 31) * 4]) = pry[k];
    }
    __syncthreads();
    f32x2 dy[8];
#pragma unroll
    for (int o = 0; o < 8; ++o) {
      dy[o] = *reinterpret_cast<const f32x2*>(&ys[(ps * 8 + o) * DW_C + cl * 2]);
      bacc += dy[o];
    }
#pragma unroll
    for (int i = 0; i < 8 + DW_K - 1; ++i) {
      const f32x2 x = *reinterpret_cast<const f32x2*>(&xs[(ps * 8 + i) * DW_C + cl * 2]);
#pragma unroll
      for (int o = 0; o < 8; ++o) {
        const int k = i - o;
        if (k >= 0 && k < DW_K) acc[k] = __builtin_elementwise_fma(dy[o], x, acc[k]);
      }
    }
  }
  // reduce the 8 position slots through LDS, 8 taps at a time; the workgroup's 32 x 128 partial sums go to its own row of
  // the workspace with plain stores (512 workgroups hammering the same 4 096 addresses with atomics ran at the contended
  // atomic rate: ~90 us of a 300 us launch) and a second tiny kernel adds the rows up (8 chunk sums per address)
  float* red = xs;     // [8 slots][8 taps][128]
  float* prow = a.part + (long)blockIdx.x * 32 * DW_C;
#pragma unroll
  for (int k0 = 0; k0 < 32; k0 += 8) {
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      const int k = k0 + kk;
      const f32x2 v = k < DW_K ? acc[k < DW_K ? k : 0] : bacc;      // slot k == 31 carries the bias gradient
      *reinterpret_cast<f32x2*>(&red[((ps * 8 + kk) * DW_C) + cl * 2]) = v;
    }
    __syncthreads();
    for (int i = tid; i < 8 * DW_C; i += 512) {
      int kk = i >> 7, ch = i & 127;
      float t = 0.f;
#pragma unroll
      for (int sl = 0; sl < 8; ++sl) t += red[(sl * 8 + kk) * DW_C + ch];
      prow[(k0 + kk) * DW_C + ch] = t;
    }
  }
}

// dW[ch][k] += sum over workgroups of part[wg][k][ch] (k == 31: dbias).  grid = (16 blocks of 256 (k, ch) pairs, 8 chunks of
// workgroups): every thread adds up 64 rows, then one atomic per chunk (8 adds per address)
__global__ void dwconv_wgrad_reduce_kernel(const float* __restrict__ part, int nwg, float* __restrict__ dW,
                                           float* __restrict__ dbias) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;       // k * 128 + ch
  const int per = (nwg + gridDim.y - 1) / gridDim.y, w0 = blockIdx.y * per, w1 = w0 + per < nwg ? w0 + per : nwg;
  if (i >= 32 * DW_C || w0 >= w1) return;
  float t = 0.f;
#pragma unroll 8
  for (int w = w0; w < w1; ++w) t += part[(long)w * 32 * DW_C + i];
  const int k = i >> 7, ch = i & 127;
  if (k < DW_K) atomicAdd(&dW[ch * DW_K + k], t);
  else if (dbias) atomicAdd(&dbias[ch], t);
}

extern "C" int se_dwconv31(const float* X, const float* W, const float* bias, float* Y, double* stats, int flip,
                           int nseq, int n, int inner, long outer_stride, long inner_stride, long pos_stride,
                           void* stream) {
  SE_REQUIRE(X && W && Y && nseq > 0 && n > 0 && inner > 0, "dwconv31: bad arguments");
  DwArgs a{{nseq, n, inner, outer_stride, inner_stride, pos_stride}, X, W, bias, Y, stats, flip};
  long nitems = (long)nseq * cdiv(n, DW_TILE);
  int nblk = nitems < 768 ? (int)((nitems + 7) / 8 * 8) : 768;          // persistent: 3 workgroups (48 KB LDS each) per CU; multiple of 8
  hipLaunchKernelGGL(dwconv_kernel, dim3(nblk), dim3(512), 0, as_stream(stream), a);
  return se_check_launch("se_dwconv31");
}

extern "C" size_t se_dwconv31_wgrad_workspace_bytes(void) { return (size_t)512 * 32 * DW_C * sizeof(float); }

extern "C" int se_dwconv31_wgrad(const float* X, const float* dY, float* dW, float* dbias, int nseq, int n,
                                 int inner, long outer_stride, long inner_stride, long pos_stride, float* ws,
                                 void* stream) {
  SE_REQUIRE(X && dY && dW && ws && nseq > 0 && n > 0 && inner > 0, "dwconv31_wgrad: bad arguments");
  DwWgradArgs a{{nseq, n, inner, outer_stride, inner_stride, pos_stride}, X, dY, dW, dbias, ws};
  long nitems = (long)nseq * cdiv(n, DW_TILE);
  int nblk = nitems < 512 ? (int)nitems : 512;
  hipLaunchKernelGGL(dwconv_wgrad_kernel, dim3(nblk), dim3(512), 0, as_stream(stream), a);
  hipLaunchKernelGGL(dwconv_wgrad_reduce_kernel, dim3(32 * DW_C / 256, 8), dim3(256), 0, as_stream(stream), (const float*)ws, nblk,
                     dW, dbias);
  return se_check_launch("se_dwconv31_wgrad");
}
