// Micro-benchmark: what does a [M][N] fp32 output cost to WRITE in the access pattern of the GEMM epilogues, next to a linear fill?
//   mode 0: linear -- every wave instruction writes 1 KB contiguous (64 lanes x 16 B)
//   mode 1: the vector epilogue -- a wave instruction writes 8 rows x 128 B (row stride N * 4 B), a wave covers 32 rows x 64 columns
//           of a 128-row tile, column block by column block (N / 64 of them), one tile per workgroup
//   mode 2: as 1, but a wave instruction writes 4 rows x 256 B (both 32-column halves of the 64-column block)
//   mode 3: as 1 with a read of the [M][64] input rows first (the row-panel kernels: read 1, write N / 64)
// Build: hipcc --offload-arch=gfx950 -O3 tools/micro/store_pattern.hip -o tools/micro/bin/store_pattern
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int MODE>
__global__ __launch_bounds__(256) void wr(float* __restrict__ Y, const float* __restrict__ X, long M, int N) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float4 val = make_float4(1.f + tid, 2.f, 3.f, 4.f);
  if (MODE == 0) {
    const long total4 = M * N / 4;
    for (long i = (long)blockIdx.x * 256 + tid; i < total4; i += (long)gridDim.x * 256) reinterpret_cast<float4*>(Y)[i] = val;
    return;
  }
  const long m0 = (long)blockIdx.x * 128;
  float4 acc = val;
  if (MODE == 3) {
    const long row = m0 + wave * 32 + (lane & 31);
    const float* p = X + (row < M ? row : M - 1) * 64 + 8 * (lane >> 5);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) { const float4 a = *reinterpret_cast<const float4*>(p + 16 * ks), b = *reinterpret_cast<const float4*>(p + 16 * ks + 4);
      acc.x += a.x + b.x; acc.y += a.y + b.y; acc.z += a.z + b.z; acc.w += a.w + b.w; }
  }
  for (int by = 0; by < N / 64; ++by) {
    if (MODE == 2) {
      const int cq = lane & 15, rr = lane >> 4;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const long row = m0 + wave * 32 + rr + 4 * i;
        if (row < M) *reinterpret_cast<float4*>(Y + row * N + by * 64 + cq * 4) = acc;
      }
    } else {
      const int cq = lane & 7, rr = lane >> 3;
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const long row = m0 + wave * 32 + rr + 8 * i;
          if (row < M) *reinterpret_cast<float4*>(Y + row * N + by * 64 + nt * 32 + cq * 4) = acc;
        }
    }
  }
}
template <int MODE> static void run(const char* name, long M, int N, float* Y, float* X) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const unsigned grid = MODE == 0 ? 256 * 16 : (unsigned)((M + 127) / 128);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(wr<MODE>, dim3(grid), dim3(256), 0, 0, Y, X, M, N);
  hipEventRecord(e0);
  const int reps = 20;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(wr<MODE>, dim3(grid), dim3(256), 0, 0, Y, X, M, N);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / reps, bytes = (double)M * N * 4 + (MODE == 3 ? (double)M * 256 : 0.0);
  printf("N %3d  %-34s %8.1f us  %5.2f TB/s\n", N, name, us, bytes / us / 1e6);
}
int main() {
  const long M = 16L * 321 * 101;
  float *Y, *X; hipMalloc(&Y, M * 256 * 4); hipMalloc(&X, M * 64 * 4); hipMemset(X, 0, M * 64 * 4);
  for (int N : {64, 128, 192, 256}) {
    run<0>("linear fill", M, N, Y, X);
    run<1>("epilogue pattern 8 rows x 128 B", M, N, Y, X);
    run<2>("4 rows x 256 B", M, N, Y, X);
    run<3>("read [M][64] + epilogue pattern", M, N, Y, X);
  }
  return 0;
}
