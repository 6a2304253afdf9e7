// Depthwise 31-tap convolution along the sequence axis of the Conformer conv module
// (DepthWiseConv1d, models/conformer.py:40-48,166: zero pad (15,15), cross-correlation, groups = 128) with the
// BatchNorm1d batch statistics (conformer.py:167) produced in the same pass.  HBM-bound: each input row is
// read once per 64-position tile (+30 halo rows) into LDS, each output written once; lanes own float4 channel
// groups (a 128-channel row = 512 contiguous bytes per 32 lanes) and 8 consecutive positions, the 31 taps of
// their 4 channels live in registers.
//
// The same kernel with flipped taps is the input gradient; the weight gradient is a persistent kernel that
// keeps 31 x float4 accumulators per lane and flushes once per workgroup.
#include <type_traits>
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

// ------------------------------------------------------------------------------------------------------------------------
// Round 5: the WHOLE backward of the depthwise convolution in one sweep -- input gradient + GLU backward (as dwconv_kernel<.., GLU>)
// AND the weight / bias gradient, which needs exactly the operands this kernel already reads: dW[c][k] = sum_p U[c][p] dH[c][p - k + 15]
// multiplies the GLU result U at the tile's own positions (read for the GLU backward anyway) with the dH rows of the tile's window.
// The stand-alone weight-gradient launch re-read (U, dH) = 531 MB per block.
//
// Structure (not the two-workgroup tiling above: 62 weight-gradient accumulators next to 62 taps do not fit 128 VGPRs, and folding
// per-tile sums into an LDS table with ds_add_f32 ran at ~180 cycles per wave-instruction: 1.7 ms per launch):
// ONE persistent 1024-thread workgroup per CU, 112-position tiles for every sequence length (n = 321: three tiles, n = 101: one),
// two channels per lane; the dH rows (+ 30 halo rows) AND the U rows of a tile are staged in LDS, both requested one tile ahead.
//   barrier, staged rows -> LDS, barrier; gate values of the own positions requested;
//   phase A (wave = position slot: 16 slots x 7 positions): taps from an LDS table (62 VGPRs, dead after the FIR), flipped-tap FIR,
//            GLU backward with U from LDS and the gate, dZ stores;
//   rows of the NEXT tile requested (buffer loads: padding rows and "no next tile" are out-of-range offsets -> zeros, no branches);
//   phase B (wave = tap group x position half: 8 groups of 4 taps x 2 halves of 56 positions): per position ONE new dH row and one
//            U row from LDS, four FMAs into 4 x 2 accumulators that stay in registers for the whole launch (the 32nd "tap" of group 7
//            is the bias gradient: the dH row itself).
// The accumulators go to the workgroup's row of the workspace at the end; dwconv_wgrad_reduce_kernel adds the rows up.
struct DwBwdArgs {
  SeqGeom g;
  const float* dH; const float* W; const float* U; const float* Z; float* dZ; float* amax_out; float* part;
  unsigned bytes128, bytes256;      // extents of the [tokens][128] operands and of dZ [tokens][256]
};
constexpr int FB_PPS = 7, FB_TILE = 16 * FB_PPS, FB_ROWS = FB_TILE + DW_K - 1, FB_NLD = (FB_ROWS * 32 + 1023) / 1024,
              FB_NLU = (FB_TILE * 32 + 1023) / 1024;
typedef unsigned u32x2fb_ __attribute__((ext_vector_type(2)));
#ifndef FB_NT
#define FB_NT 8                         // taps per wave in the weight-gradient phase (4: twice the LDS reads; 8: +8 persistent VGPRs)
#endif
constexpr int FB_LDS_BYTES = (FB_ROWS * DW_C + FB_TILE * DW_C + DW_K * DW_C) * 4;

__global__ __launch_bounds__(1024) void dwconv_bwd_fused_kernel(DwBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float fb_smem[];
  float* xs = fb_smem;                          // [142][128]: dH rows p0 - 15 .. p0 + 126
  float* us = fb_smem + FB_ROWS * DW_C;         // [112][128]: U rows p0 .. p0 + 111
  float* wt = us + FB_TILE * DW_C;              // [31][128]: taps in input-gradient (flipped) order
  const int tid = threadIdx.x, cl = tid & 63, ps = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned cl8 = (unsigned)cl * 8u;
  const int n = a.g.n;
  for (int i = tid; i < DW_C * DW_K; i += 1024) {
    const int ch = i / DW_K, k = i - ch * DW_K;
    wt[(DW_K - 1 - k) * DW_C + ch] = a.W[i];
  }
  const __amdgpu_buffer_rsrc_t rH = make_rsrc_(a.dH, a.bytes128), rU = make_rsrc_(a.U, a.bytes128), rG = make_rsrc_(a.Z, a.bytes128),
                               rD = make_rsrc_(a.dZ, a.bytes256);
  const int tiles = (n + FB_TILE - 1) / FB_TILE;
  const long nitems = (long)a.g.nseq * tiles;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, S = gridDim.x >> 3;
  const long Q = (nitems + 7) >> 3, ibase = (long)xcd * Q, iend = ibase + Q < nitems ? ibase + Q : nitems;
  const unsigned rsb = (unsigned)(a.g.pos_stride * DW_C * 4);          // bytes between consecutive positions of a sequence
  float zmax = 0.f;
  float4 ld[FB_NLD], lu[FB_NLU];
  // rows of item `it` (all-zero rows for it >= iend): dH row r = (tid >> 5) + 32 k <-> position p0 - 15 + r, U row r <-> p0 + r.
  // Offsets = a wave-uniform part per load + ONE lane constant (mod 2^32: rows above the sequence wrap and are deselected)
  const int lrow = tid >> 5;
  const unsigned lane_off = (unsigned)lrow * rsb + (unsigned)(tid & 31) * 16u;
  auto request_rows = [&](long it) {
    const bool live = it < iend;
    const long itc = live ? it : 0;
    const int seq = (int)(itc / tiles), p0 = (int)(itc - (long)seq * tiles) * FB_TILE;
    const unsigned base = (unsigned)(((long)(seq / a.g.inner) * a.g.outer_stride + (long)(seq % a.g.inner) * a.g.inner_stride) * (DW_C * 4));
    const int plo = 15 - p0 > 0 ? 15 - p0 : 0;           // dH row r is a real position iff plo <= r < phi (ONE unsigned compare)
    const int phi = n - (p0 - 15) < FB_ROWS ? n - (p0 - 15) : FB_ROWS;
    const unsigned span = live && phi > plo ? (unsigned)(phi - plo) : 0u;
    const unsigned uspan = live ? (unsigned)(n - p0 < FB_TILE ? n - p0 : FB_TILE) : 0u;
#pragma unroll
    for (int k = 0; k < FB_NLD; ++k) {
      const int row = lrow + 32 * k;
      const unsigned off = base + (unsigned)(p0 - 15 + 32 * k) * rsb + lane_off;
      ld[k] = buf_load4_(rH, (unsigned)(row - plo) < span ? off : BUF_OOB_);
    }
#pragma unroll
    for (int k = 0; k < FB_NLU; ++k) {
      const int row = lrow + 32 * k;
      const unsigned off = base + (unsigned)(p0 + 32 * k) * rsb + lane_off;
      lu[k] = buf_load4_(rU, (unsigned)row < uspan ? off : BUF_OOB_);
    }
  };
  // phase-B role of this wave: taps 4 tg .. 4 tg + 3 (k = 31: bias gradient), positions 56 ph .. 56 ph + 55 of the tile
  constexpr int NT = FB_NT, NGRP = 32 / NT, NPART = 16 / NGRP, PLEN = FB_TILE / NPART;      // taps per wave, tap groups, position parts
  const int tg = ps % NGRP, ph = ps / NGRP;
  f32x2 wacc[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) wacc[j] = (f32x2){0.f, 0.f};
  request_rows(ibase + slot);
  for (long it = ibase + slot; it < iend; it += S) {
    const int seq = (int)(it / tiles), p0 = (int)(it - (long)seq * tiles) * FB_TILE;
    const unsigned base = (unsigned)(((long)(seq / a.g.inner) * a.g.outer_stride + (long)(seq % a.g.inner) * a.g.inner_stride) * (DW_C * 4));
    __syncthreads();                           // previous tile's LDS reads (phase B) done; first pass: tap table written
#pragma unroll
    for (int k = 0; k < FB_NLD; ++k) {
      const int row = lrow + 32 * k;
      if (row < FB_ROWS) *reinterpret_cast<float4*>(&xs[row * DW_C + (tid & 31) * 4]) = ld[k];
    }
#pragma unroll
    for (int k = 0; k < FB_NLU; ++k) {
      const int row = lrow + 32 * k;
      if (row < FB_TILE) *reinterpret_cast<float4*>(&us[row * DW_C + (tid & 31) * 4]) = lu[k];
    }
    __syncthreads();
    // gate values of the own positions: requested ahead of the FIR that hides their latency
    const int pbase = p0 + ps * FB_PPS;                       // wave-uniform (ps is)
    f32x2 gt[FB_PPS];
#pragma unroll
    for (int o = 0; o < FB_PPS; ++o)
      gt[o] = __builtin_bit_cast(f32x2, buf_load2_(rG, pbase + o < n ? base + (unsigned)(pbase + o) * rsb + cl8 : BUF_OOB_));
    // ---- phase A: flipped-tap FIR + GLU backward
    {
      constexpr int NR = FB_PPS + DW_K - 1;
      const float* xrow = &xs[(ps * FB_PPS) * DW_C + cl * 2];
      f32x2 acc[FB_PPS];
#pragma unroll
      for (int o = 0; o < FB_PPS; ++o) acc[o] = (f32x2){0.f, 0.f};
      // the 31 taps in two halves (16 + 15): 32 tap registers at a time instead of 62 -- the weight-gradient accumulators of phase
      // B stay in registers next to them; rows 16 .. 21 of the window are read twice
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        constexpr int KH = 16;
        const int K0 = h * KH, K1 = h == 0 ? KH : DW_K;
        f32x2 w[KH];
#pragma unroll
        for (int k = 0; k < KH; ++k)
          if (K0 + k < K1) w[k] = *reinterpret_cast<const f32x2*>(&wt[(K0 + k) * DW_C + cl * 2]);
        const int R0 = K0, R1 = K1 - 1 + FB_PPS;               // window rows [R0, R1) carry taps [K0, K1)
        constexpr int GA = 4;
        f32x2 xg[2][GA];
#pragma unroll
        for (int j = 0; j < GA; ++j) xg[0][j] = *reinterpret_cast<const f32x2*>(xrow + (R0 + j) * DW_C);
#pragma unroll
        for (int g = 0; g < (KH + FB_PPS - 1 + GA - 1) / GA; ++g) {
          asm volatile("" ::: "memory");
          if (R0 + (g + 1) * GA < R1) {
#pragma unroll
            for (int j = 0; j < GA; ++j)
              if (R0 + (g + 1) * GA + j < R1) xg[(g + 1) & 1][j] = *reinterpret_cast<const f32x2*>(xrow + (R0 + (g + 1) * GA + j) * DW_C);
          }
          asm volatile("" ::: "memory");
#pragma unroll
          for (int j = 0; j < GA; ++j) {
            const int i = R0 + g * GA + j;
            if (i < R1) {
#pragma unroll
              for (int o = 0; o < FB_PPS; ++o) {
                const int k = i - o;
                if (k >= K0 && k < K1) acc[o] = __builtin_elementwise_fma(xg[g & 1][j], w[k - K0], acc[o]);
              }
            }
          }
        }
      }
#pragma unroll
      for (int o = 0; o < FB_PPS; ++o) asm volatile("" : "+v"(acc[o]));
      // GLU backward on the way out: dZ[:, c] = dU sigmoid(g), dZ[:, 128 + c] = dU u (1 - sigmoid(g)); rows of dZ are 256 floats
      const float* urow = &us[(ps * FB_PPS) * DW_C + cl * 2];
#pragma unroll
      for (int o = 0; o < FB_PPS; ++o) {
        const f32x2 u = *reinterpret_cast<const f32x2*>(urow + o * DW_C);
        f32x2 da, dg;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
#ifdef FB_ABL_NO_GLU
          const float sg = gt[o][e];
#else
          const float sg = sigmoidf_(gt[o][e]);
#endif
          da[e] = acc[o][e] * sg;
          dg[e] = acc[o][e] * u[e] * (1.f - sg);
          zmax = fmaxf(zmax, pbase + o < n ? fmaxf(fabsf(da[e]), fabsf(dg[e])) : 0.f);
        }
        const unsigned off = pbase + o < n ? 2u * (base + (unsigned)(pbase + o) * rsb) + cl8 : BUF_OOB_;
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2fb_, da), rD, off, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2fb_, dg), rD, off + (off == BUF_OOB_ ? 0u : (unsigned)DW_C * 4u), 0, 0);
      }
    }
    asm volatile("" : "+v"(zmax));             // the maximum is folded HERE (left alone, the compiler parks all 28 |dZ| values across phase B)
    asm volatile("" ::: "memory");
    request_rows(it + S);                      // in flight during phase B
    asm volatile("" ::: "memory");
    // ---- phase B: dW[k] += U[p] dH[p - k + 15]: dH row index (in xs) of position offset q and tap k = 4 tg + j is q + 30 - k
#ifndef FB_ABL_NO_B
    {
      const int q0 = ph * PLEN;
      const float* up = &us[q0 * DW_C + cl * 2];
      const float* xp = &xs[(q0 + 30 - NT * tg) * DW_C + cl * 2];       // row of j = 0; j-th tap: j rows up
      // BIAS (last tap group): its last slot is the bias gradient -- the dH row of the position itself (row q + 15 = xp row
      // t + 15 - 30 + NT (NGRP - 1) = t + 17 - NT)
      auto phase_b = [&](auto bias_tag) {
        constexpr bool BIAS = decltype(bias_tag)::value;
        constexpr int UB = 2, BOFF = 17 - NT;
        static_assert(PLEN % UB == 0, "position part not a multiple of the read group");
        f32x2 r[NT];                            // r[j]: the row of tap slot j for the CURRENT position (r[0] is read per position)
#pragma unroll
        for (int jj = 1; jj < NT; ++jj)
          if (!(BIAS && jj == NT - 1)) r[jj] = *reinterpret_cast<const f32x2*>(xp - jj * DW_C);
#pragma unroll 1
        for (int qq = 0; qq < PLEN; qq += UB) {
          f32x2 uu[UB], r0[UB], rb[UB];
#pragma unroll
          for (int e = 0; e < UB; ++e) {
            uu[e] = *reinterpret_cast<const f32x2*>(up + (qq + e) * DW_C);
            r0[e] = *reinterpret_cast<const f32x2*>(xp + (qq + e) * DW_C);
            if constexpr (BIAS) rb[e] = *reinterpret_cast<const f32x2*>(xp + (qq + e + BOFF) * DW_C);
          }
#pragma unroll
          for (int e = 0; e < UB; ++e) {
            r[0] = r0[e];
#pragma unroll
            for (int jj = 0; jj < NT; ++jj) {
              if (BIAS && jj == NT - 1) wacc[jj] += rb[e];
              else wacc[jj] = __builtin_elementwise_fma(uu[e], r[jj], wacc[jj]);
            }
#pragma unroll
            for (int jj = NT - 1; jj >= 1; --jj) r[jj] = r[jj - 1];
          }
        }
      };
      if (tg == NGRP - 1) phase_b(std::true_type{}); else phase_b(std::false_type{});
    }
#endif
  }
  if (a.amax_out) {
    zmax = wave_max(zmax);
    if ((tid & 63) == 0) amax_raise_(a.amax_out, zmax);
  }
  // the two position halves of every tap group summed through LDS, then the workgroup's [32][128] row of the workspace
  __syncthreads();
  float* red = xs;                              // [position parts][32 taps][128]
  int t2 = threadIdx.x;                         // indices re-derived behind an opaque copy: computed up front they sat in scratch
  asm volatile("" : "+v"(t2));                  // for the whole tile loop
  const int w2 = t2 >> 6, c2 = t2 & 63;
#pragma unroll
  for (int j = 0; j < NT; ++j) *reinterpret_cast<f32x2*>(&red[((w2 / NGRP) * 32 + NT * (w2 % NGRP) + j) * DW_C + c2 * 2]) = wacc[j];
  __syncthreads();
  float* prow = a.part + (long)blockIdx.x * 32 * DW_C;
  float4 t = *reinterpret_cast<const float4*>(red + t2 * 4);
#pragma unroll
  for (int h = 1; h < NPART; ++h) {
    const float4 v = *reinterpret_cast<const float4*>(red + h * 32 * DW_C + t2 * 4);
    t = make_float4(t.x + v.x, t.y + v.y, t.z + v.z, t.w + v.w);
  }
  *reinterpret_cast<float4*>(prow + t2 * 4) = t;
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
#ifdef SE_DW_TWIN      // measurement build (tools/build_variant_lib.sh dwtwin -DSE_DW_TWIN se_dwconv.hip): the access-pattern twin, wrong results
  if (pps == 13) hipLaunchKernelGGL((dwconv_kernel<13, false, true>), dim3(nblk), dim3(512), 0, as_stream(stream), a);
  else if (pps == 14) hipLaunchKernelGGL((dwconv_kernel<14, false, true>), dim3(nblk), dim3(512), 0, as_stream(stream), a);
  else
#endif
  if (pps == 8) hipLaunchKernelGGL(dwconv_kernel<8>, dim3(nblk), dim3(512), 0, as_stream(stream), a);
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
#ifdef SE_DW_TWIN
  if (pps == 13) hipLaunchKernelGGL((dwconv_kernel<13, true, true>), dim3(nblk), dim3(512), 0, as_stream(stream), a);
  else if (pps == 14) hipLaunchKernelGGL((dwconv_kernel<14, true, true>), dim3(nblk), dim3(512), 0, as_stream(stream), a);
  else
#endif
  if (pps == 8) hipLaunchKernelGGL((dwconv_kernel<8, true>), dim3(nblk), dim3(512), 0, as_stream(stream), a);
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

// input gradient + GLU backward + weight / bias gradient of the depthwise convolution in one sweep (dwconv_bwd_fused_kernel).
// dW [128][31] and dbias [128] are ACCUMULATED (atomics of the reduce kernel); ws: se_dwconv31_wgrad_workspace_bytes().
extern "C" int se_dwconv31_bwd_fused(const float* dH, const float* W, const float* U, const float* G, float* dZ, float* amax_out,
                                     float* dW, float* dbias, float* ws, long ntok, int nseq, int n, int inner, long outer_stride,
                                     long inner_stride, long pos_stride, void* stream) {
  SE_REQUIRE(dH && W && U && G && dZ && dW && ws && nseq > 0 && n > 0 && inner > 0 && ntok > 0, "dwconv31_bwd_fused: bad arguments");
  SE_REQUIRE(ntok * (DW_C * 8L) < (1L << 32) - 4096, "dwconv31_bwd_fused: operands beyond 32-bit buffer offsets (%ld tokens)", ntok);
  SE_REQUIRE(pos_stride > 0 && (long)(nseq - 1) / inner * outer_stride + (long)(inner - 1) * inner_stride + (long)(n - 1) * pos_stride < ntok,
             "dwconv31_bwd_fused: sequence geometry reaches past %ld tokens", ntok);
  DwBwdArgs a{{nseq, n, inner, outer_stride, inner_stride, pos_stride}, dH, W, U, G, dZ, amax_out, ws,
              (unsigned)(ntok * DW_C * 4), (unsigned)(ntok * DW_C * 8)};
  const long nitems = (long)nseq * cdiv(n, FB_TILE);
  const int ncu = se_cu_count();
  int nblk = nitems < ncu ? (int)((nitems + 7) / 8 * 8) : ncu / 8 * 8;      // one workgroup per CU; multiple of 8 (XCD split)
  if (nblk > 512) nblk = 512;                                              // rows of the workspace
  static unsigned lds_done = 0;
  SE_REQUIRE(se_raise_lds((const void*)dwconv_bwd_fused_kernel, FB_LDS_BYTES, &lds_done), "dwconv31_bwd_fused: cannot raise the LDS limit");
  hipLaunchKernelGGL(dwconv_bwd_fused_kernel, dim3(nblk), dim3(1024), FB_LDS_BYTES, as_stream(stream), a);
  hipLaunchKernelGGL(dwconv_wgrad_reduce_kernel, dim3(32 * DW_C / 256, 32), dim3(256), 0, as_stream(stream), (const float*)ws, nblk,
                     dW, dbias);
  return se_check_launch("se_dwconv31_bwd_fused");
}
