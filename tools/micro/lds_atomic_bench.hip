// Micro-benchmark: cost of LDS float atomics (ds_add_f32, no return) vs plain LDS stores and read-modify-writes, under the
// access pattern of the attention-backward accumulators (16 lanes of a lane group add to 16 consecutive words of a row,
// the 4 lane groups to 4 different rows; row stride = 4 mod 8 words), 8 waves per workgroup, one workgroup per CU.
// Build:  hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/micro/lds_atomic_bench.hip -o gpurun_out/lds_atomic_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

template <int MODE, int SAMEROWS>
__global__ __launch_bounds__(512) void bench(float* out, long long* cyc, int iters) {
  extern __shared__ float acc[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c = lane & 15, g = lane >> 4;
  for (int i = tid; i < 16 * 340; i += 512) acc[i] = 0.f;
  __syncthreads();
  float v = (float)tid * 1e-3f;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    // wave w touches key tile (it + 2w) % 21 (lock-step pattern: distinct tiles) or, with SAMEROWS, every wave the same tile
    int kt = SAMEROWS ? (it % 21) : ((it + 2 * wave) % 21);
    int j0 = kt * 16;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float* p = &acc[(4 * g + r) * 340 + j0 + c];
      if (MODE == 0) atomicAdd(p, v);                 // ds_add_f32
      else if (MODE == 1) *p = v;                      // ds_write_b32
      else { float x = *p; *p = x + v; }               // ds_read_b32 + ds_write_b32
    }
    v += 1.0f;
  }
  __syncthreads();
  long long t1 = __builtin_amdgcn_s_memtime();
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
  out[blockIdx.x * 512 + tid] = acc[tid];
}

template <int MODE, int SAMEROWS>
static void run(const char* name, int nblk) {
  float* out; long long* cyc;
  hipMalloc(&out, sizeof(float) * 512 * nblk);
  hipMalloc(&cyc, sizeof(long long) * nblk);
  const int iters = 2000;
  for (int rep = 0; rep < 2; ++rep)
    hipLaunchKernelGGL((bench<MODE, SAMEROWS>), dim3(nblk), dim3(512), 16 * 340 * 4, 0, out, cyc, iters);
  hipDeviceSynchronize();
  std::vector<long long> h(nblk);
  hipMemcpy(h.data(), cyc, sizeof(long long) * nblk, hipMemcpyDeviceToHost);
  double s = 0; for (auto x : h) s += x;
  // per iteration: 8 waves x 4 wave-instructions
  printf("%-34s %8.1f cycles/iter/workgroup  = %6.1f cycles per wave-instruction (8 waves x 4 per iter, LDS pipe shared)\n", name,
         s / nblk / iters, s / nblk / iters / 32.0);
  hipFree(out); hipFree(cyc);
}

int main() {
  run<0, 0>("ds_add_f32 distinct tiles", 256);
  run<0, 1>("ds_add_f32 same tile (contended)", 256);
  run<1, 0>("ds_write_b32", 256);
  run<2, 0>("read+write RMW distinct tiles", 256);
  return 0;
}
