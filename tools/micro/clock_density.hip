// Micro-benchmark: the shader clock the part sustains under instruction streams of different density.  One 8-wave workgroup per CU
// (two waves per SIMD), every CU busy; wall time from HIP events, cycles from s_memtime -> effective clock = cycles / time.
//   mode 0: every wave MFMA only (v_mfma_f32_32x32x16_f16, dependent chain)      1: every wave vector FMAs only
//   mode 3: every wave alternates 1 MFMA with 7 vector FMAs (both pipes of every SIMD busy)
//   mode 5: every wave 1 MFMA + 3 vector FMAs, then an idle gap (s_sleep) of about the same length: half-dense stream
// (homogeneous waves only: with different kinds of waves the workgroup's time is the slower kind's and one counter says nothing)
// Build: hipcc --offload-arch=gfx950 -O3 tools/micro/clock_density.hip -o tools/micro/bin/clock_density
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(512) void bench(float* out, long long* cyc, int iters) {
  __shared__ f32x4 lds[2048];
  const int tid = threadIdx.x, wave = tid >> 6;
  f16x8 a8, b8;
  for (int e = 0; e < 8; ++e) { a8[e] = (_Float16)(float)((tid + e) & 7); b8[e] = (_Float16)(float)((tid * 3 + e) & 7); }
  f32x16 acc; for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  float v[8]; for (int e = 0; e < 8; ++e) v[e] = (float)(tid + e) * 1e-3f;
  lds[tid] = (f32x4){1.f, 2.f, 3.f, 4.f}; lds[tid + 512] = lds[tid]; lds[tid + 1024] = lds[tid]; lds[tid + 1536] = lds[tid];
  __syncthreads();
  const bool mfma_wave = MODE == 0 || ((MODE == 2 || MODE == 4) && wave < 4);
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 5) {
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a8, b8, acc, 0, 0, 0);
#pragma unroll
        for (int e = 0; e < 3; ++e) v[e] = __builtin_fmaf(v[e], 1.0001f, 0.5f);
        __builtin_amdgcn_s_sleep(1);
      }
    } else if (MODE == 3) {
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a8, b8, acc, 0, 0, 0);
#pragma unroll
        for (int e = 0; e < 7; ++e) v[e] = __builtin_fmaf(v[e], 1.0001f, 0.5f);
      }
    } else if (mfma_wave) {
#pragma unroll
      for (int m = 0; m < 8; ++m) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a8, b8, acc, 0, 0, 0);
    } else {
#pragma unroll
      for (int m = 0; m < 8; ++m) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = __builtin_fmaf(v[e], 1.0001f, 0.5f);
        if (MODE == 4) { f32x4 r = lds[(tid + 64 * m + it) & 2047]; v[m] += r[0]; }
      }
    }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  float s = acc[0] + acc[15];
  for (int e = 0; e < 8; ++e) s += v[e];
  out[blockIdx.x * 512 + tid] = s;
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int MODE> static void run(const char* name) {
  float* out; long long* cyc; const int nblk = 256, iters = 200000;
  (void)hipMalloc(&out, sizeof(float) * 512 * nblk); (void)hipMalloc(&cyc, sizeof(long long) * nblk);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((bench<MODE>), dim3(nblk), dim3(512), 0, 0, out, cyc, iters / 10);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((bench<MODE>), dim3(nblk), dim3(512), 0, 0, out, cyc, iters);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  long long h[256]; (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double s = 0; for (int i = 0; i < 256; ++i) s += h[i];
  printf("%-62s %8.2f ms, %6.1f cycles per iteration, effective clock %5.2f GHz\n", name, ms, s / 256 / iters, s / 256 / (ms * 1e6));
  (void)hipFree(out); (void)hipFree(cyc);
}
int main() {
  run<0>("every wave: MFMA only"); run<1>("every wave: vector FMAs only");
  run<3>("every wave: 1 MFMA + 7 vector FMAs interleaved"); run<5>("every wave: 1 MFMA + 3 vector FMAs + s_sleep");
  run<0>("every wave: MFMA only (again)");
  return 0;
}
