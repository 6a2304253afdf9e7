// Micro-benchmark: v_mfma_f32_32x32x16_f16 on one SIMD -- cycles per instruction when consecutive MFMAs accumulate into the SAME
// registers (NACC = 1: a dependent chain) vs 2 / 4 independent accumulators; one wave per SIMD and two waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_dep.hip -o tools/micro/bin/mfma_dep
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(512) void bench(float* out, long long* cyc, int iters) {
  const int tid = threadIdx.x;
  f16x8 a8, b8;
  for (int e = 0; e < 8; ++e) { a8[e] = (_Float16)(float)((tid + e) & 7); b8[e] = (_Float16)(float)((tid * 3 + e) & 7); }
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  __syncthreads();
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 16; ++m) acc[m % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a8, b8, acc[m % NACC], 0, 0, 0);
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][15];
  out[blockIdx.x * 512 + tid] = s;
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int NACC> static void run(int threads) {
  float* out; long long* cyc; const int nblk = 256, iters = 2000;
  (void)hipMalloc(&out, sizeof(float) * 512 * nblk); (void)hipMalloc(&cyc, sizeof(long long) * nblk);
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((bench<NACC>), dim3(nblk), dim3(threads), 0, 0, out, cyc, iters);
  (void)hipDeviceSynchronize();
  long long h[256]; (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double s = 0; for (int i = 0; i < 256; ++i) s += h[i];
  printf("accumulators %d, %d waves per SIMD: %6.1f cycles per MFMA per wave, %6.1f per SIMD\n", NACC, threads / 256, s / 256 / iters / 16, s / 256 / iters / 16 / (threads / 256));
  (void)hipFree(out); (void)hipFree(cyc);
}
int main() {
  run<1>(256); run<2>(256); run<4>(256); run<1>(512); run<2>(512); run<4>(512);
  return 0;
}
