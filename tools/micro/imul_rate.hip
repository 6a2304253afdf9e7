// Micro-benchmark: issue cost of the integer multiplies (the dropout hash's 32-bit multiplies) against a plain vector op, one wave per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 tools/micro/imul_rate.hip -o tools/micro/bin/imul_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int KIND>
__global__ __launch_bounds__(256) void bench(unsigned* out, long long* cyc, int iters) {
  unsigned a = threadIdx.x * 2654435761u + 1u, b = threadIdx.x | 1u, c = a ^ 0x55555555u, d = b + 77u;
  __syncthreads();
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      if (KIND == 0) { asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a) : "v"(b)); asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(c) : "v"(d)); }
      if (KIND == 1) { asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a) : "v"(b)); asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(c) : "v"(d)); }
      if (KIND == 2) { asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(a) : "v"(b)); asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(c) : "v"(d)); }
      if (KIND == 3) { asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a) : "v"(b)); asm volatile("v_xor_b32 %0, %0, %1" : "+v"(c) : "v"(d)); }
      if (KIND == 4) { asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a) : "v"(b)); asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(c) : "v"(d)); }
      if (KIND == 5) { asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(*(unsigned long long*)&a) : "v"(b), "v"(d) : "vcc"); }
    }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * 256 + threadIdx.x] = a + c;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int KIND> static void run(const char* name, int per_iter) {
  unsigned* out; long long* cyc; const int nblk = 256, iters = 20000;
  (void)hipMalloc(&out, 4 * 256 * nblk); (void)hipMalloc(&cyc, 8 * nblk);
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((bench<KIND>), dim3(nblk), dim3(256), 0, 0, out, cyc, iters);
  (void)hipDeviceSynchronize();
  long long h[256]; (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double s = 0; for (int i = 0; i < 256; ++i) s += h[i];
  printf("%-16s %6.2f cycles per instruction (one wave per SIMD)\n", name, s / 256 / iters / per_iter);
}
int main() {
  run<3>("v_xor_b32", 32); run<0>("v_mul_lo_u32", 32); run<1>("v_mul_u32_u24", 32); run<2>("v_mad_u32_u24", 32); run<4>("v_mul_hi_u32", 32);
  run<5>("v_mad_u64_u32", 16);
  return 0;
}
