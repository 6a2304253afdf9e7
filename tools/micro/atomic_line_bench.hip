// Micro-benchmark: fp32 atomic adds (no return) of MANY workgroups to the SAME cache lines -- how long the memory side takes per
// wave-level atomic instruction when G workgroups each add a 64-float vector into one of R replicas of that vector.
// Build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/micro/atomic_line_bench.hip -o tools/micro/bin/atomic_line_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void add_vec(float* v, int R, int per_wg) {
  float* p = v + (size_t)(blockIdx.x % R) * 64 + threadIdx.x;
  for (int i = 0; i < per_wg; ++i) atomicAdd(p, 1.0f);
}
int main() {
  float* v; (void)hipMalloc(&v, 4096 * 64 * sizeof(float)); (void)hipMemset(v, 0, 4096 * 64 * sizeof(float));
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int Gs[] = {256, 2048, 8192}, Rs[] = {1, 4, 16, 64, 256, 4096}, Ps[] = {1, 2, 16};
  for (int G : Gs) for (int P : Ps) for (int R : Rs) {
    if (R > G) continue;
    hipLaunchKernelGGL(add_vec, dim3(G), dim3(64), 0, 0, v, R, P);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int rep = 0; rep < 5; ++rep) hipLaunchKernelGGL(add_vec, dim3(G), dim3(64), 0, 0, v, R, P);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1000.0 / 5;
    printf("workgroups %5d x %2d instr, replicas %4d: %8.1f us per launch, %7.2f ns per instruction on one replica\n", G, P, R, us, us * 1000.0 / ((double)G * P / R));
  }
  return 0;
}
