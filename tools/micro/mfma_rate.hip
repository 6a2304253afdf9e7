// Micro-benchmark: issue rate of the 16x16 MFMA forms on gfx950 (cycles per instruction on one SIMD, one wave per SIMD,
// four independent accumulators): v_mfma_f32_16x16x32_{bf16,f16} vs the K = 16 forms v_mfma_f32_16x16x16_{f16,bf16_1k}.
// Build: hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_rate.hip -o tools/micro/bin/mfma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int KIND>
__global__ __launch_bounds__(256) void bench(float* out, long long* cyc, int iters) {
  const int tid = threadIdx.x;
  f16x8 a8, b8; bf16x8 ab8, bb8; f16x4 a4, b4; s16x4 as4, bs4;
  for (int e = 0; e < 8; ++e) { a8[e] = (_Float16)(float)((tid + e) & 7); b8[e] = (_Float16)(float)((tid * 3 + e) & 7); ab8[e] = (__bf16)(float)((tid + e) & 7); bb8[e] = (__bf16)(float)((tid + 2 * e) & 7); }
  for (int e = 0; e < 4; ++e) { a4[e] = a8[e]; b4[e] = b8[e]; as4[e] = (short)(0x3f80 + tid + e); bs4[e] = (short)(0x3f80 + e); }
  f32x4 acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      if (KIND == 0) acc[m & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab8, bb8, acc[m & 3], 0, 0, 0);
      if (KIND == 1) acc[m & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, acc[m & 3], 0, 0, 0);
      if (KIND == 2) acc[m & 3] = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, acc[m & 3], 0, 0, 0);
      if (KIND == 3) acc[m & 3] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(as4, bs4, acc[m & 3], 0, 0, 0);
    }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 4; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + tid] = s;
}
template <int KIND> static void run(const char* name) {
  float* out; long long* cyc; const int nblk = 256, iters = 2000;
  hipMalloc(&out, nblk * 256 * 4); hipMalloc(&cyc, nblk * 8);
  hipLaunchKernelGGL(bench<KIND>, dim3(nblk), dim3(256), 0, 0, out, cyc, iters);
  hipLaunchKernelGGL(bench<KIND>, dim3(nblk), dim3(256), 0, 0, out, cyc, iters);
  hipDeviceSynchronize();
  long long h[256]; hipMemcpy(h, cyc, nblk * 8, hipMemcpyDeviceToHost);
  double s = 0; for (int i = 0; i < nblk; ++i) s += h[i];
  // s_memtime ticks at 100 MHz on gfx9: report ticks per instruction and let the ratios speak
  printf("%-34s %8.4f ticks / MFMA (one wave per SIMD)\n", name, s / nblk / iters / 16);
  hipFree(out); hipFree(cyc);
}
int main() {
  run<0>("v_mfma_f32_16x16x32_bf16");
  run<1>("v_mfma_f32_16x16x32_f16");
  run<2>("v_mfma_f32_16x16x16_f16");
  run<3>("v_mfma_f32_16x16x16_bf16 (1k)");
  return 0;
}
