// Micro-benchmark: LDS cycles per wave-instruction of the access patterns of ff_bwd_fused4_kernel (csrc/se_ff_fused4.hip), 8 waves
// per workgroup, one workgroup per CU, every wave issuing the same instruction kind (the LDS pipe is shared: 8 x N instructions per
// iteration).  Ideal: ds_read_b128 8 cycles (1 KB at 128 B / clk), 8-byte reads / writes 4 cycles.
// Build: hipcc --offload-arch=gfx950 -O3 tools/micro/lds_ff4_bench.hip -o tools/micro/bin/lds_ff4_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int RW = 128, PL = 32 * RW, IMG = 2 * PL, W1PL = 256 * RW, ZTR = 64, ZTPL = 256 * ZTR;
constexpr int O_W1 = 0, O_ZT = 2 * W1PL, O_ROWS = O_ZT + 2 * ZTPL, LDS = O_ROWS + 4 * IMG;
static __device__ __forceinline__ int sw16(int r) { return (((r >> 1) & 1) << 2) | ((r >> 2) & 3); }
static __device__ __forceinline__ u32x2 tr8(const unsigned char* p) {
  return __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p)));
}
// MODE 0: A fragments of the row images (b128, swizzled)   1: the same unswizzled (reference: conflicts)
//      2: transposed reads of the row images                3: transposed reads of ZT     4: transposed reads of the W1 image
//      5: ZT writes (b64, swizzled)                         6: plain ds_read_b64 of consecutive 8-byte words (reference: ideal)
//      7: transposed reads of the row images WITHOUT the swizzle (reference)
template <int MODE>
__global__ __launch_bounds__(512) void bench(float* out, long long* cyc, int iters) {
  extern __shared__ __attribute__((aligned(128))) unsigned char sm[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < LDS / 4; i += 512) ((float*)sm)[i] = (float)i;
  __syncthreads();
  const int j = lane & 31, kg = lane >> 5, gi = lane >> 4, q4 = (lane & 15) >> 2, p4 = lane & 3, kgt = gi >> 1;
  const int fragA = j * RW, x16 = ((kg ^ sw16(j)) << 4);
  const int trA0 = ((4 * kgt + q4) * RW + (((2 * (gi & 1) + (p4 >> 1)) ^ (((q4 >> 1) << 2) | kgt)) << 4) + 8 * (p4 & 1));
  const int trA1 = (trA0 ^ 32) + 8 * RW;
  const int trN0 = (4 * kgt + q4) * RW + ((2 * (gi & 1) + (p4 >> 1)) << 4) + 8 * (p4 & 1), trN1 = trN0 + 8 * RW;
  const int ztw = O_ZT + (32 * wave + j) * ZTR + ((kg ^ ((j >> 1) & 7)) << 3);
  const int ch = wave & 1, kq = wave >> 1, jz = 64 * kq + 8 * kgt + q4;
  const int ztr0 = O_ZT + jz * ZTR + (((4 * (gi & 1) + p4) ^ ((4 * kgt) | (q4 >> 1))) << 3);
  const int ztr1 = O_ZT + (jz + 4) * ZTR + (((4 * (gi & 1) + p4) ^ ((4 * kgt) | (q4 >> 1) | 2)) << 3);
  const int c16w = 4 * ch + 2 * (gi & 1) + (p4 >> 1);
  const int w1r0 = O_W1 + jz * RW + ((c16w ^ (((q4 >> 1) << 2) | (2 * kgt))) << 4) + 8 * (p4 & 1);
  const int w1r1 = O_W1 + (jz + 4) * RW + ((c16w ^ (((q4 >> 1) << 2) | (2 * kgt) | 1)) << 4) + 8 * (p4 & 1);
  u32x4 a4 = {0, 0, 0, 0}; u32x2 a2 = {0, 0};
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    int opq = 0; asm volatile("" : "+v"(opq));      // (opaque zero: nothing is loop-invariant)
    const unsigned char* rows = sm + O_ROWS + (it & 1) * 2 * IMG + opq; const unsigned char* smo = sm + opq;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      if (MODE == 0) a4 += *(const u32x4*)(rows + fragA + (x16 ^ (32 * ks)));
      if (MODE == 1) a4 += *(const u32x4*)(rows + fragA + 32 * ks + 16 * kg);
      if (MODE == 2) a2 += tr8(rows + ((ks & 1 ? trA1 : trA0) ^ (64 * (ks >> 1))));
      if (MODE == 7) a2 += tr8(rows + (ks & 1 ? trN1 : trN0) + 64 * (ks >> 1));
      if (MODE == 3) a2 += tr8(smo + (ks & 1 ? ztr1 : ztr0) + 16 * (ks >> 1) * ZTR);
      if (MODE == 4) a2 += tr8(smo + (ks & 1 ? w1r1 : w1r0) + 16 * (ks >> 1) * RW);
      if (MODE == 5) { *(u32x2*)((unsigned char*)smo + (ztw ^ (16 * ks))) = a2; a2[0] += ks; }
      if (MODE == 6) a2 += *(const u32x2*)(smo + O_ZT + wave * 4096 + ks * 512 + lane * 8);
    }
  }
  __syncthreads();
  long long t1 = __builtin_amdgcn_s_memtime();
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
  out[blockIdx.x * 512 + tid] = (float)(a4[0] + a4[3] + a2[0] + a2[1]);
}
template <int MODE> static void run(const char* name) {
  float* out; long long* cyc; const int nblk = 256, iters = 4000;
  (void)hipMalloc(&out, sizeof(float) * 512 * nblk); (void)hipMalloc(&cyc, sizeof(long long) * nblk);
  (void)hipFuncSetAttribute((const void*)bench<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((bench<MODE>), dim3(nblk), dim3(512), LDS, 0, out, cyc, iters);
  (void)hipDeviceSynchronize();
  long long h[256]; (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double s = 0; for (int i = 0; i < 256; ++i) s += h[i];
  printf("%-52s %6.2f cycles per wave-instruction\n", name, s / 256 / iters / 32.0);
  (void)hipFree(out); (void)hipFree(cyc);
}
int main() {
  run<0>("row-image A fragment (b128, swizzled)"); run<1>("row-image A fragment (b128, no swizzle)");
  run<2>("row-image transposed read (swizzled)"); run<7>("row-image transposed read (no swizzle)");
  run<3>("ZT transposed read"); run<4>("W1-image transposed read"); run<5>("ZT write (b64, swizzled)"); run<6>("plain b64 read, consecutive");
  return 0;
}
