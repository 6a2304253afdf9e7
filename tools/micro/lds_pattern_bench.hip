// Micro-benchmark: LDS cycles per wave-instruction of the access patterns of attn_bwd3_kernel (8 waves per workgroup,
// one workgroup per CU, every wave in its own region like the kernel).  Ground truth for the bank-conflict model.
// Build: hipcc --offload-arch=gfx950 -O3 tools/micro/lds_pattern_bench.hip -o tools/micro/bin/lds_pattern_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
constexpr int KT = 6, NK = 6, SW = 116, WAVE = 3 * KT * 16 * 32 + 1536 + 16 * SW * 4;

template <int MODE>
__global__ __launch_bounds__(512) void bench(float* out, long long* cyc, int iters) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c = lane & 15, g = lane >> 4;
  unsigned char* wl = smem + wave * WAVE;
  unsigned char* Kimg = wl; unsigned char* Dimg = wl + 3 * KT * 16 * 32;
  float* strip = (float*)(wl + 3 * KT * 16 * 32 + 1536);
  for (int i = lane; i < WAVE / 4; i += 64) ((float*)wl)[i] = (float)i;
  __syncthreads();
  float acc = 0.f; u32x2 a2 = {0, 0};
  const int trrow = c >> 2, trcol = c & 3;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    const int s = it % NK;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (MODE == 0) acc += strip[(4 * g + r) * SW + 16 * (NK - s) + (4 * g + r) - c];                 // cell read
      if (MODE == 1) strip[(4 * g + r) * SW + 16 * (NK - s) + (4 * g + r) - c] = acc + r;              // cell write
      if (MODE == 2) strip[(4 * g + r) * SW + 16 * (NK - s) + c] = acc + r;                            // U write
      if (MODE == 3) { u32x2 v = *(const u32x2*)(Kimg + (((r % 3) * KT * 16 + s * 16 + c) * 16 + 4 * g) * 2); a2 += v; }   // K row frag
      if (MODE == 4) { *(u32x2*)(Dimg + ((r % 3) * 16 + c) * 32 + g * 8) = a2; a2[0] += r; }           // image store
      if (MODE == 5) { u32x2 v = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(Dimg + ((r % 3) * 16 + 4 * g + trrow) * 32 + trcol * 8))); a2 += v; }
      if (MODE == 6) { u32x2 v = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(Kimg + (((r % 3) * KT * 16 + s * 16 + 4 * g + trrow) * 16 + 4 * trcol) * 2))); a2 += v; }
      if (MODE == 7) { float4 v = *(const float4*)(&strip[c * SW + 16 * (NK - s) + 4 * g]); acc += v.x + v.w; }   // w4
    }
  }
  __syncthreads();
  long long t1 = __builtin_amdgcn_s_memtime();
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
  out[blockIdx.x * 512 + tid] = acc + a2[0] + a2[1];
}
template <int MODE> static void run(const char* name) {
  float* out; long long* cyc; const int nblk = 256, iters = 2000;
  (void)hipMalloc(&out, sizeof(float) * 512 * nblk); (void)hipMalloc(&cyc, sizeof(long long) * nblk);
  (void)hipFuncSetAttribute((const void*)bench<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * WAVE);
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((bench<MODE>), dim3(nblk), dim3(512), 8 * WAVE, 0, out, cyc, iters);
  (void)hipDeviceSynchronize();
  std::vector<long long> h(nblk); (void)hipMemcpy(h.data(), cyc, sizeof(long long) * nblk, hipMemcpyDeviceToHost);
  double s = 0; for (auto x : h) s += x;
  printf("%-28s %7.2f cycles per wave-instruction (8 waves x 4 per iteration share the LDS pipe)\n", name, s / nblk / iters / 32.0);
  (void)hipFree(out); (void)hipFree(cyc);
}
int main() {
  run<0>("cell read (skewed b32)"); run<1>("cell write (skewed b32)"); run<2>("U write (b32)"); run<3>("K row fragment (b64)");
  run<4>("image store (b64)"); run<5>("image tr read (tr_b16)"); run<6>("K tr read (tr_b16)"); run<7>("w4 (b128)");
  return 0;
}
