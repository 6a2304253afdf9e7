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

constexpr int DW_C = 128, DW_K = 31;
// Non-temporal stores of the output rows (SE_DW_NT builds): faster in an isolated loop (107 -> 103 us on the time axis, 121 -> 111
// us on the frequency axis, tools/microbench.py dw_bench) but SLOWER inside the step (rocprofv3, serial order: 126.5 -> 136.6 us,
// fused input gradient 254 -> 274 us; profiles/r03a vs r03b): the consumer kernel starts right behind and finds part of a
// normally-stored 266 MB result in the 256 MB MALL.  Plain stores stay.
#ifdef SE_DW_NT
#define DW_STORE(p, v) __builtin_nontemporal_store((v), (p))
#else
#define DW_STORE(p, v) (*(p) = (v))
#endif
// non-temporal LOADS of the tiles: slower in the convolution (103 -> 122 us: neighbouring tiles re-read 30 halo rows out of L2);
// in the weight gradient 130 -> 123 us in isolation, not re-measured inside the step: plain loads there too
#define DW_LOAD4(p) (*reinterpret_cast<const f32x4*>(p))

struct DwArgs {
  SeqGeom g;
  const float* X; const float* W; const float* bias; float* Y; double* stats; int flip;
  // GLU variant (input gradient of the conv module, conformer.py:164-166 backwards): U = the GLU result a sigmoid(g) (the conv's
  // forward input) and Z = the gate half g, both [tokens][128] in the conv operands' layout; Y = dZ [tokens][256]; the depthwise
  // input gradient dU never goes to memory
  const float* Z; float* amax_out; const float* U;
};

// 512 threads: lane pair-channel cl = tid & 63 (2 channels), position slot ps = tid >> 6 (8 slots x PPS positions).
// A tile is 8 * PPS positions + 30 halo rows in LDS.  PPS is chosen per sequence length so that (a) a whole n = 101 sequence
// is ONE tile (PPS = 13: no halo re-read at all, its 15 + 15 padding rows are zeros that are never fetched) and (b) n = 321 is
// three tiles of 112 (PPS = 14: 142 rows read per 112 written instead of 94 per 64, and 336 computed positions for 321 instead
// of 384).  More positions per slot also mean more FMAs per LDS read (PPS * 31 per PPS + 30 rows).  Two channels per lane keep
// 31 taps + PPS accumulators at ~100 VGPRs: two 512-thread workgroups (2 x 72 KB of LDS) per CU, no cross-tile register
// prefetch -- the other workgroup's compute covers this one's loads.  The FMAs are explicit 2-wide vectors (v_pk_fma_f32: both
// channels of a lane in one instruction; this translation unit is built with packed fp32 ops enabled -- no MFMA here to stall).
typedef float f32x2 __attribute__((ext_vector_type(2)));

// GLU: the epilogue applies the backward of GLU (a * sigmoid(gate)) to the result on its way out: dZ[:, c] = dU sigmoid(g),
// dZ[:, 128 + c] = dU a sigmoid(g) (1 - sigmoid(g)) = dU u (1 - sigmoid(g)) with u = a sigmoid(g) the forward GLU result (so the
// pre-GLU value half a is never stored: 266 MB less written by the pw1 GEMM and read here, per block) -- the stand-alone glu_bwd pass (read Z and dU, write dZ: 1.33 GB per block at
// batch 16, 214 us x 8 per step) and the write + re-read of dU disappear; max |dZ| is raised for the scaled-fp16 consumers.
// TWIN (timing only, SE_DW_TWIN=1: wrong results): the exact load / LDS staging / store pattern of the kernel with the 31-tap FIR
// replaced by one LDS read per output -- what the access pattern alone costs inside the step
template <int PPS, bool GLU = false, bool TWIN = false>
__global__ __launch_bounds__(512, 4) void dwconv_kernel(DwArgs a) {
  constexpr int TILE = 8 * PPS, ROWS = TILE + DW_K - 1;
  __shared__ __attribute__((aligned(16))) float xs[ROWS * DW_C];
  const int tid = threadIdx.x, cl = tid & 63, ps = tid >> 6;
  const int n = a.g.n;
  for (int i = tid; i < DW_C * DW_K; i += 512) {
    int ch = i / DW_K, k = i - ch * DW_K;
    xs[(a.flip ? DW_K - 1 - k : k) * DW_C + ch] = a.W[i];
  }
  __syncthreads();
  f32x2 w[DW_K];
#pragma unroll
  for (int k = 0; k < DW_K; ++k) w[k] = *reinterpret_cast<const f32x2*>(&xs[k * DW_C + cl * 2]);
  f32x2 bv = {0.f, 0.f};
  if (a.bias) bv = *reinterpret_cast<const f32x2*>(a.bias + cl * 2);
  f32x2 s = {0.f, 0.f}, q2 = {0.f, 0.f};
  float zmax = 0.f;
  const int tiles = (n + TILE - 1) / TILE;
  const long nitems = (long)a.g.nseq * tiles;
  constexpr int NLD = (ROWS * 32 + 511) / 512;      // float4 per thread per tile
  // XCD-aware item order: workgroups b and b + 8 share an XCD (round-robin dispatch), so XCD x walks its own contiguous
  // eighth of the (sequence, tile) items with its gridDim / 8 workgroups side by side -- consecutive tiles of a sequence are
  // in flight on the same XCD at the same time and their 30 shared halo rows are served by that XCD's L2 instead of HBM
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, S = gridDim.x >> 3;
  const long Q = (nitems + 7) >> 3, ibase = (long)xcd * Q, iend = ibase + Q < nitems ? ibase + Q : nitems;
  for (long it = ibase + slot; it < iend; it += S) {
    const int seq = (int)(it / tiles), p0 = (int)(it - (long)seq * tiles) * TILE;
    const long base = (long)(seq / a.g.inner) * a.g.outer_stride + (long)(seq % a.g.inner) * a.g.inner_stride;
    const float* __restrict__ Xb = a.X + base * DW_C;
    float* __restrict__ Yt = a.Y + base * DW_C + (long)p0 * (a.g.pos_stride * DW_C);
    const long rs = a.g.pos_stride * DW_C;
    const unsigned rs32 = (unsigned)rs;
    const float* __restrict__ Xt = Xb + (long)(p0 - 15) * rs;      // row 0 of the tile (never dereferenced outside [0, n))
    float4 ld[NLD];
    // wave-uniform base per 16-row step (SGPRs) + ONE 32-bit lane offset for all of them (rows of a tile span < 2^30 bytes:
    // host-checked); per-load 64-bit lane addresses cost 2 VGPRs each and pushed the kernel over 128
    const unsigned ld_off = (unsigned)(tid >> 5) * rs32 + (unsigned)(tid & 31) * 4u;
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
      const int row = (tid >> 5) + 16 * k, p = p0 - 15 + row;
      const float* __restrict__ Xk = Xt + (long)(16 * k) * rs;
      ld[k] = (row < ROWS && p >= 0 && p < n) ? __builtin_bit_cast(float4, DW_LOAD4(Xk + ld_off)) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();                           // previous tile (or the tap table) fully consumed
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
      const int i = tid + k * 512;
      if (i < ROWS * 32) *reinterpret_cast<float4*>(&xs[(i >> 5) * DW_C + (i & 31) * 4]) = ld[k];
    }
    __syncthreads();
    f32x2 acc[PPS];
#pragma unroll
    for (int o = 0; o < PPS; ++o) acc[o] = bv;
    // rows in groups of 4, the next group's LDS reads issued before this group's FMAs; the compiler barriers keep the
    // fully unrolled loop from hoisting all PPS + 30 reads to the top (which cost 60 more VGPRs and a wave per SIMD)
    constexpr int NR = PPS + DW_K - 1, NG = (NR + 3) / 4;
    const float* xrow = &xs[(ps * PPS) * DW_C + cl * 2];
    f32x2 xg[2][4];
    if constexpr (TWIN) {
#pragma unroll
      for (int o = 0; o < PPS; ++o) acc[o] = *reinterpret_cast<const f32x2*>(xrow + (o + 15) * DW_C) * w[15];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) xg[0][j] = *reinterpret_cast<const f32x2*>(xrow + j * DW_C);
#pragma unroll
    for (int g = 0; g < (TWIN ? 0 : NG); ++g) {
      asm volatile("" ::: "memory");
      if (g + 1 < NG) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if ((g + 1) * 4 + j < NR) xg[(g + 1) & 1][j] = *reinterpret_cast<const f32x2*>(xrow + ((g + 1) * 4 + j) * DW_C);
      }
      asm volatile("" ::: "memory");
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int i = g * 4 + j;
        if (i < NR) {
#pragma unroll
          for (int o = 0; o < PPS; ++o) {
            const int k = i - o;
            if (k >= 0 && k < DW_K) acc[o] = __builtin_elementwise_fma(xg[g & 1][j], w[k], acc[o]);
          }
        }
      }
    }
    // pin the accumulators here: otherwise the compiler sinks each output's 31 FMAs into its `p < n` branch below -- every
    // row of the tile read into registers first (76 VGPRs), then one DEPENDENT chain of 31 FMAs per output
#pragma unroll
    for (int o = 0; o < PPS; ++o) asm volatile("" : "+v"(acc[o]));
    const unsigned st_off = (unsigned)(ps * PPS) * rs32 + (unsigned)cl * 2u;
    if constexpr (GLU) {
      // rows of dZ are 256 floats: twice the row stride of the conv operands (U, gate); positions past n read row n - 1
      // (unconditional loads, in groups of GG positions: all loads of a group in flight before the first is used), store nothing
      const long toff = base * DW_C + (long)p0 * rs + (long)(ps * PPS) * rs + cl * 2;
      const float* __restrict__ Ut = a.U + toff;
      const float* __restrict__ Zt = a.Z + toff;
      float* __restrict__ Dt = a.Y + 2 * (base * DW_C + (long)p0 * rs) + 2 * (long)(ps * PPS) * rs + cl * 2;
      const int plast = n - 1 - (p0 + ps * PPS);            // last valid position offset of this slot (may be negative)
      constexpr int GG = 2;
#pragma unroll
      for (int o0 = 0; o0 < PPS; o0 += GG) {
        f32x2 za[GG], zg[GG];
#pragma unroll
        for (int j = 0; j < GG; ++j) {
          if (o0 + j < PPS) {
            int oc = o0 + j; oc = oc > plast ? plast : oc; oc = oc < 0 ? 0 : oc;
            const long ro = (long)oc * rs;                   // an all-padding slot reads a valid dummy row
            za[j] = *reinterpret_cast<const f32x2*>((plast >= 0 ? Ut : a.U) + ro);
            zg[j] = *reinterpret_cast<const f32x2*>((plast >= 0 ? Zt : a.Z) + ro);
          }
        }
#pragma unroll
        for (int j = 0; j < GG; ++j) {
          const int o = o0 + j;
          if (o < PPS && o <= plast) {
            f32x2 da, dg;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
              const float sg = sigmoidf_(zg[j][e]);
              da[e] = acc[o][e] * sg;
              dg[e] = acc[o][e] * za[j][e] * (1.f - sg);
              zmax = fmaxf(zmax, fmaxf(fabsf(da[e]), fabsf(dg[e])));
            }
            DW_STORE(reinterpret_cast<f32x2*>(Dt + 2 * (long)o * rs), da);
            DW_STORE(reinterpret_cast<f32x2*>(Dt + 2 * (long)o * rs + DW_C), dg);
          }
        }
      }
    } else {
#pragma unroll
      for (int o = 0; o < PPS; ++o) {
        const int p = p0 + ps * PPS + o;
        if (p < n) {
          DW_STORE(reinterpret_cast<f32x2*>(Yt + (long)o * rs + st_off), acc[o]);
          s += acc[o];
          q2 = __builtin_elementwise_fma(acc[o], acc[o], q2);
        }
      }
    }
  }
  if (GLU && a.amax_out) {
    zmax = wave_max(zmax);
    if ((tid & 63) == 0) amax_raise_(a.amax_out, zmax);
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

// weight gradient, same thread layout and tiling (512 threads, 2 channels per lane, 8 slots x PPS positions): 31 f32x2
// accumulators per lane, persistent over tiles, one flush per workgroup.  Only X goes through LDS (halo rows shared by the
// slots); every lane reads the dY values of its own PPS positions straight from global memory (8 B per lane, 512 B per
// wave-instruction), so a workgroup needs 72 KB of LDS and two stay resident per CU.
template <int PPS>
__global__ __launch_bounds__(512, 4) void dwconv_wgrad_kernel(DwWgradArgs a) {
  constexpr int TILE = 8 * PPS, ROWS = TILE + DW_K - 1;
  __shared__ __attribute__((aligned(16))) float xs[ROWS * DW_C];
  const int tid = threadIdx.x, cl = tid & 63, ps = tid >> 6;
  const int n = a.g.n;
  const int tiles = (n + TILE - 1) / TILE;
  const long nitems = (long)a.g.nseq * tiles;
  f32x2 acc[DW_K];
#pragma unroll
  for (int k = 0; k < DW_K; ++k) acc[k] = (f32x2){0.f, 0.f};
  f32x2 bacc = {0.f, 0.f};
  constexpr int NLD = (ROWS * 32 + 511) / 512;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, S = gridDim.x >> 3;
  const long Q = (nitems + 7) >> 3, ibase = (long)xcd * Q, iend = ibase + Q < nitems ? ibase + Q : nitems;
  for (long it = ibase + slot; it < iend; it += S) {
    const int seq = (int)(it / tiles), p0 = (int)(it % tiles) * TILE;
    const long base = (long)(seq / a.g.inner) * a.g.outer_stride + (long)(seq % a.g.inner) * a.g.inner_stride;
    const float* __restrict__ Xb = a.X + base * DW_C;
    const float* __restrict__ Gt = a.dY + base * DW_C + (long)p0 * (a.g.pos_stride * DW_C);
    const long rs = a.g.pos_stride * DW_C;
    const unsigned rs32 = (unsigned)rs;
    const float* __restrict__ Xt = Xb + (long)(p0 - 15) * rs;      // row 0 of the tile (never dereferenced outside [0, n))
    float4 ld[NLD];
    // wave-uniform base per 16-row step (SGPRs) + ONE 32-bit lane offset for all of them (rows of a tile span < 2^30 bytes:
    // host-checked); per-load 64-bit lane addresses cost 2 VGPRs each and pushed the kernel over 128
    const unsigned ld_off = (unsigned)(tid >> 5) * rs32 + (unsigned)(tid & 31) * 4u;
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
      const int row = (tid >> 5) + 16 * k, p = p0 - 15 + row;
      const float* __restrict__ Xk = Xt + (long)(16 * k) * rs;
      ld[k] = (row < ROWS && p >= 0 && p < n) ? __builtin_bit_cast(float4, DW_LOAD4(Xk + ld_off)) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
      const int i = tid + k * 512;
      if (i < ROWS * 32) *reinterpret_cast<float4*>(&xs[(i >> 5) * DW_C + (i & 31) * 4]) = ld[k];
    }
    // dY after the staging registers are dead (31 accumulators + PPS dY values + the tile's loads would not fit 128 VGPRs)
    asm volatile("" ::: "memory");
    f32x2 dy[PPS];
    const unsigned dy_off = (unsigned)(ps * PPS) * rs32 + (unsigned)cl * 2u;
#pragma unroll
    for (int o = 0; o < PPS; ++o) {
      const int p = p0 + ps * PPS + o;
      dy[o] = p < n ? *reinterpret_cast<const f32x2*>(Gt + (long)o * rs + dy_off) : (f32x2){0.f, 0.f};
    }
    __syncthreads();
#pragma unroll
    for (int o = 0; o < PPS; ++o) bacc += dy[o];
    constexpr int NR = PPS + DW_K - 1, NG = (NR + 3) / 4;
    const float* xrow = &xs[(ps * PPS) * DW_C + cl * 2];
    f32x2 xg[2][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) xg[0][j] = *reinterpret_cast<const f32x2*>(xrow + j * DW_C);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      asm volatile("" ::: "memory");
      if (g + 1 < NG) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if ((g + 1) * 4 + j < NR) xg[(g + 1) & 1][j] = *reinterpret_cast<const f32x2*>(xrow + ((g + 1) * 4 + j) * DW_C);
      }
      asm volatile("" ::: "memory");
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int i = g * 4 + j;
        if (i < NR) {
#pragma unroll
          for (int o = 0; o < PPS; ++o) {
            const int k = i - o;
            if (k >= 0 && k < DW_K) acc[k] = __builtin_elementwise_fma(dy[o], xg[g & 1][j], acc[k]);
          }
        }
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

// dW[ch][k] += sum over workgroups of part[wg][k][ch] (k == 31: dbias).  grid = (16 blocks of 256 (k, ch) pairs, 32 chunks of
// workgroups)
__global__ void dwconv_wgrad_reduce_kernel(const float* __restrict__ part, int nwg, float* __restrict__ dW,
                                           float* __restrict__ dbias) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;       // k * 128 + ch
  const int per = (nwg + gridDim.y - 1) / gridDim.y, w0 = blockIdx.y * per, w1 = w0 + per < nwg ? w0 + per : nwg;
  if (i >= 32 * DW_C || w0 >= w1) return;
  // <= 16 partials per thread with all loads in flight at once, then one atomic per chunk (32 adds per address)
  float v[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) v[j] = w0 + j < w1 ? part[(long)(w0 + j) * 32 * DW_C + i] : 0.f;
  float t = 0.f;
#pragma unroll
  for (int j = 0; j < 16; ++j) t += v[j];
  for (int w = w0 + 16; w < w1; ++w) t += part[(long)w * 32 * DW_C + i];
  const int k = i >> 7, ch = i & 127;
  if (k < DW_K) atomicAdd(&dW[ch * DW_K + k], t);
  else if (dbias) atomicAdd(&dbias[ch], t);
}

extern "C" int se_dwconv31(const float* X, const float* W, const float* bias, float* Y, double* stats, int flip,
                           int nseq, int n, int inner, long outer_stride, long inner_stride, long pos_stride,
                           void* stream) {
  SE_REQUIRE(X && W && Y && nseq > 0 && n > 0 && inner > 0, "dwconv31: bad arguments");
  SE_REQUIRE(pos_stride > 0 && pos_stride * DW_C * 160L < (1L << 30), "dwconv31: position stride too large for 32-bit tile offsets");
  DwArgs a{{nseq, n, inner, outer_stride, inner_stride, pos_stride}, X, W, bias, Y, stats, flip, nullptr, nullptr};
  // tile = 8 slots x PPS positions: 64 (short sequences), 104 (n <= 104: the frequency axis, n = 101, is one tile), 112
  const int pps = n <= 64 ? 8 : (n <= 104 ? 13 : 14);
  const long nitems = (long)nseq * cdiv(n, 8 * pps);
  const int nblk = nitems < 512 ? (int)((nitems + 7) / 8 * 8) : 512;    // persistent: 2 workgroups (<= 72 KB LDS each) per CU; multiple of 8
  static const bool twin = getenv("SE_DW_TWIN") != nullptr;
  if (twin && pps == 13) hipLaunchKernelGGL((dwconv_kernel<13, false, true>), dim3(nblk), dim3(512), 0, as_stream(stream), a);
  else if (twin && pps == 14) hipLaunchKernelGGL((dwconv_kernel<14, false, true>), dim3(nblk), dim3(512), 0, as_stream(stream), a);
  else if (pps == 8) hipLaunchKernelGGL(dwconv_kernel<8>, dim3(nblk), dim3(512), 0, as_stream(stream), a);
  else if (pps == 13) hipLaunchKernelGGL(dwconv_kernel<13>, dim3(nblk), dim3(512), 0, as_stream(stream), a);
  else hipLaunchKernelGGL(dwconv_kernel<14>, dim3(nblk), dim3(512), 0, as_stream(stream), a);
  return se_check_launch("se_dwconv31");
}

extern "C" int se_dwconv31_glu_bwd(const float* dH, const float* W, const float* U, const float* G, float* dZ, float* amax_out,
                                   int nseq, int n, int inner, long outer_stride, long inner_stride, long pos_stride,
                                   void* stream) {
  SE_REQUIRE(dH && W && U && G && dZ && nseq > 0 && n > 0 && inner > 0, "dwconv31_glu_bwd: bad arguments");
  SE_REQUIRE(pos_stride > 0 && pos_stride * DW_C * 320L < (1L << 30), "dwconv31_glu_bwd: position stride too large for 32-bit tile offsets");
  DwArgs a{{nseq, n, inner, outer_stride, inner_stride, pos_stride}, dH, W, nullptr, dZ, nullptr, 1, G, amax_out, U};
  const int pps = n <= 64 ? 8 : (n <= 104 ? 13 : 14);
  const long nitems = (long)nseq * cdiv(n, 8 * pps);
  const int nblk = nitems < 512 ? (int)((nitems + 7) / 8 * 8) : 512;
  static const bool twin = getenv("SE_DW_TWIN") != nullptr;
  if (twin && pps == 13) hipLaunchKernelGGL((dwconv_kernel<13, true, true>), dim3(nblk), dim3(512), 0, as_stream(stream), a);
  else if (twin && pps == 14) hipLaunchKernelGGL((dwconv_kernel<14, true, true>), dim3(nblk), dim3(512), 0, as_stream(stream), a);
  else if (pps == 8) hipLaunchKernelGGL((dwconv_kernel<8, true>), dim3(nblk), dim3(512), 0, as_stream(stream), a);
  else if (pps == 13) hipLaunchKernelGGL((dwconv_kernel<13, true>), dim3(nblk), dim3(512), 0, as_stream(stream), a);
  else hipLaunchKernelGGL((dwconv_kernel<14, true>), dim3(nblk), dim3(512), 0, as_stream(stream), a);
  return se_check_launch("se_dwconv31_glu_bwd");
}

extern "C" size_t se_dwconv31_wgrad_workspace_bytes(void) { return (size_t)512 * 32 * DW_C * sizeof(float); }

extern "C" int se_dwconv31_wgrad(const float* X, const float* dY, float* dW, float* dbias, int nseq, int n,
                                 int inner, long outer_stride, long inner_stride, long pos_stride, float* ws,
                                 void* stream) {
  SE_REQUIRE(X && dY && dW && ws && nseq > 0 && n > 0 && inner > 0, "dwconv31_wgrad: bad arguments");
  SE_REQUIRE(pos_stride > 0 && pos_stride * DW_C * 160L < (1L << 30), "dwconv31_wgrad: position stride too large for 32-bit tile offsets");
  DwWgradArgs a{{nseq, n, inner, outer_stride, inner_stride, pos_stride}, X, dY, dW, dbias, ws};
  const int pps = n <= 64 ? 8 : (n <= 104 ? 13 : 14);
  const long nitems = (long)nseq * cdiv(n, 8 * pps);
  const int nblk = nitems < 512 ? (int)((nitems + 7) / 8 * 8) : 512;      // <= 512 rows of the workspace; multiple of 8 (XCD split)
  if (pps == 8) hipLaunchKernelGGL(dwconv_wgrad_kernel<8>, dim3(nblk), dim3(512), 0, as_stream(stream), a);
  else if (pps == 13) hipLaunchKernelGGL(dwconv_wgrad_kernel<13>, dim3(nblk), dim3(512), 0, as_stream(stream), a);
  else hipLaunchKernelGGL(dwconv_wgrad_kernel<14>, dim3(nblk), dim3(512), 0, as_stream(stream), a);
  hipLaunchKernelGGL(dwconv_wgrad_reduce_kernel, dim3(32 * DW_C / 256, 32), dim3(256), 0, as_stream(stream), (const float*)ws, nblk,
                     dW, dbias);
  return se_check_launch("se_dwconv31_wgrad");
}
