// Micro-benchmark: LDS cycles per wave-instruction of the MFMA-fragment reads / staging writes of the split-bf16 GEMM
// kernels (conv3_bf16_kernel & co): ds_read_b128 with lane -> (row = lane & 31, 16-byte column = lane >> 5) at a run-time
// row stride, and the ds_write_b64 staging pattern (row = tid >> 3, 8-byte column = tid & 7).  4 waves per workgroup,
// NWG workgroups per CU resident, every wave in the same tile like the kernels.
// Build: hipcc --offload-arch=gfx950 -O3 tools/micro/lds_frag_bench.hip -o tools/micro/bin/lds_frag_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void bench(unsigned* out, long long* cyc, int iters, int stride, int kgoff) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 40960 / 4; i += 256) ((unsigned*)smem)[i] = i;
  __syncthreads();
  u32x4 a = {0, 0, 0, 0};
  const unsigned char* rd = smem + (wave * 32 + (lane & 31)) * stride + (lane >> 5) * kgoff;
  unsigned char* wr = smem + (tid >> 3) * stride + (tid & 7) * 8;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    const int io = (it & 3) * 4 * stride;     // iteration-dependent so the accesses stay in the loop
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      if (MODE == 0) { u32x4 v = *(const u32x4*)(rd + io + (r & 1) * 32 + (r >> 1) * 2 * stride); a += v; }
      if (MODE == 1) { *(u32x2*)(wr + io + (r & 3) * 32 * stride) = (u32x2){a[0], a[1]}; a[0] += r; }
      if (MODE == 2) { u32x2 v = *(const u32x2*)(rd + io + (r & 1) * 32 + (r >> 1) * 2 * stride); a[0] += v[0]; a[1] += v[1]; }
    }
  }
  __syncthreads();
  long long t1 = __builtin_amdgcn_s_memtime();
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
  out[blockIdx.x * 256 + tid] = a[0] + a[1] + a[2] + a[3];
}
template <int MODE> static void run(const char* name, int stride, int kgoff, int nwg) {
  unsigned* out; long long* cyc; const int nblk = 256 * nwg, iters = 2000;
  (void)hipMalloc(&out, sizeof(unsigned) * 256 * nblk); (void)hipMalloc(&cyc, sizeof(long long) * nblk);
  const int lds = 160 * 1024 / nwg - 1024;
  (void)hipFuncSetAttribute((const void*)bench<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((bench<MODE>), dim3(nblk), dim3(256), lds, 0, out, cyc, iters, stride, kgoff);
  (void)hipDeviceSynchronize();
  std::vector<long long> h(nblk); (void)hipMemcpy(h.data(), cyc, sizeof(long long) * nblk, hipMemcpyDeviceToHost);
  double s = 0; for (auto x : h) s += x;
  // per CU: nwg * 4 waves * 8 instructions per iteration share the LDS pipe
  printf("%-10s stride %4d B kg %3d  %d wg/CU: %6.2f cycles per wave-instruction (CU-wide LDS pipe)\n", name, stride, kgoff, nwg,
         s / nblk / iters / (8.0 * 4 * nwg));
  (void)hipFree(out); (void)hipFree(cyc);
}
int main() {
  const int strides[] = {64, 80, 96, 144, 160, 272};
  for (int nwg : {1, 3})
    for (int st : strides) run<0>("read b128", st, 16, nwg);
  for (int st : strides) run<2>("read b64", st, 8, 3);
  for (int st : strides) run<1>("write b64", st, 0, 3);
  return 0;
}
