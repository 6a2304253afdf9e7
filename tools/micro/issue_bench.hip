// Micro-benchmark: how the matrix pipe and the vector ALU of one SIMD share issue cycles on gfx950.
// One workgroup per CU, WPS waves per SIMD.  Every wave runs `iters` rounds of NM MFMAs (32x32x16 bf16, four independent
// accumulators) and/or NV vector instructions of one kind, either mixed in the SAME wave or split ACROSS waves (waves
// 0..3 matrix only, the others vector only).  Prints cycles per round per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 tools/micro/issue_bench.hip -o tools/micro/bin/issue_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int KIND>
static __device__ __forceinline__ void valu(float& a, float& b, float& c, float& d) {
  if (KIND == 0) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(*(f32x2*)&a) : "v"(*(f32x2*)&c));          // needs pairs; see below
  if (KIND == 1) asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(a) : "v"(c));
  if (KIND == 2) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(a) : "v"(c), "v"(d));
  if (KIND == 3) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(a) : "v"(c), "v"(d) : "vcc");
  if (KIND == 4) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a) : "v"(c), "v"(d));
  if (KIND == 5) asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(a) : "v"(c));
  if (KIND == 6) asm volatile("v_cndmask_b32 %0, %1, %2, s[20:21]" : "=v"(a) : "v"(c), "v"(d) : "s20", "s21");
  if (KIND == 7) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(*(f32x2*)&a) : "v"(*(f32x2*)&c));
  if (KIND == 8) asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(*(f32x2*)&a) : "v"(*(f32x2*)&c));
  if (KIND == 9) asm volatile("v_sub_f32 %0, %1, %2" : "=v"(a) : "v"(c), "v"(d));
  if (KIND == 10) asm volatile("v_exp_f32 %0, %1" : "=v"(a) : "v"(c));
  if (KIND == 11) asm volatile("v_perm_b32 %0, %1, %2, %1" : "=v"(a) : "v"(c), "v"(d));
  if (KIND == 12) asm volatile("v_max_f32 %0, %1, %2" : "=v"(a) : "v"(c), "v"(d));
  if (KIND == 13) asm volatile("v_mov_b32 %0, %1" : "=v"(a) : "v"(c));
  if (KIND == 14) asm volatile("v_pk_add_f16 %0, %1, %2" : "=v"(a) : "v"(c), "v"(d));
  if (KIND == 15) asm volatile("v_cmp_gt_f32 vcc, %1, %2\n v_cndmask_b32 %0, %1, %2, vcc" : "=v"(a) : "v"(c), "v"(d) : "vcc");
  if (KIND == 16) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "=v"(a) : "v"(c));
  if (KIND == 17) asm volatile("v_bfe_u32 %0, %1, 16, 16" : "=v"(a) : "v"(c));
  if (KIND == 18) asm volatile("v_alignbit_b32 %0, %1, %2, 16" : "=v"(a) : "v"(c), "v"(d));
}

template <int NM, int NV, int KIND, int SPLIT>
__global__ __launch_bounds__(1024) void bench(float* out, long long* cyc, int iters) {
  const int tid = threadIdx.x, wave = tid >> 6;
  bf16x8 af, bf;
  for (int e = 0; e < 8; ++e) { af[e] = (__bf16)(float)(tid + e); bf[e] = (__bf16)(float)(tid * 3 + e); }
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float x[8] __attribute__((aligned(8)));
  for (int i = 0; i < 8; ++i) x[i] = tid + i;
  const bool do_m = SPLIT ? wave < 4 : true, do_v = SPLIT ? wave >= 4 : true;
  __syncthreads();
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if (do_m && do_v) {
      constexpr int R = NM ? (NV + NM - 1) / NM : 0;
#pragma unroll
      for (int m = 0; m < NM; ++m) {
        acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf, acc[m & 3], 0, 0, 0);
#pragma unroll
        for (int v = 0; v < R; ++v) if (m * R + v < NV) valu<KIND>(x[(2 * v) & 7], x[(2 * v + 1) & 7], x[(2 * v + 4) & 7], x[(2 * v + 5) & 7]);
      }
      if (NM == 0) {
#pragma unroll
        for (int v = 0; v < NV; ++v) valu<KIND>(x[(2 * v) & 7], x[(2 * v + 1) & 7], x[(2 * v + 4) & 7], x[(2 * v + 5) & 7]);
      }
    } else if (do_m) {
#pragma unroll
      for (int m = 0; m < NM; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf, acc[m & 3], 0, 0, 0);
    } else {
#pragma unroll
      for (int v = 0; v < NV; ++v) valu<KIND>(x[(2 * v) & 7], x[(2 * v + 1) & 7], x[(2 * v + 4) & 7], x[(2 * v + 5) & 7]);
    }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  __syncthreads();
  long long t2 = __builtin_amdgcn_s_memtime();
  if (tid == 0) cyc[blockIdx.x] = t2 - t0;
  (void)t1;
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  for (int i = 0; i < 8; ++i) s += x[i];
  out[blockIdx.x * 1024 + tid] = s;
}
template <int NM, int NV, int KIND, int SPLIT> static void run(const char* name, int wps) {
  float* out; long long* cyc; const int nblk = 256, iters = 500;
  (void)hipMalloc(&out, sizeof(float) * 1024 * nblk); (void)hipMalloc(&cyc, sizeof(long long) * nblk);
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((bench<NM, NV, KIND, SPLIT>), dim3(nblk), dim3(256 * wps), 0, 0, out, cyc, iters);
  (void)hipDeviceSynchronize();
  std::vector<long long> h(nblk); (void)hipMemcpy(h.data(), cyc, sizeof(long long) * nblk, hipMemcpyDeviceToHost);
  double s = 0; for (auto x : h) s += x;
  printf("%-44s NM=%2d NV=%3d waves/SIMD=%d: %8.1f ticks per round\n", name, NM, NV, wps, s / nblk / iters);
  (void)hipFree(out); (void)hipFree(cyc);
}
#define BOTH(K, NAME) run<0, 168, K, 0>(NAME " only", 1); run<0, 168, K, 0>(NAME " only", 3); run<24, 168, K, 0>("same wave: mfma + " NAME, 1); run<24, 168, K, 0>("same wave: mfma + " NAME, 3);
int main() {
  run<24, 0, 1, 0>("mfma only", 1); run<24, 0, 1, 0>("mfma only", 3);
  BOTH(1, "v_and_b32") BOTH(0, "v_pk_add_f32") BOTH(7, "v_pk_mul_f32") BOTH(8, "v_pk_fma_f32") BOTH(9, "v_sub_f32") BOTH(4, "v_fma_f32")
  BOTH(2, "v_cvt_pk_bf16_f32") BOTH(3, "v_cndmask vcc") BOTH(6, "v_cndmask sgpr") BOTH(15, "v_cmp+v_cndmask") BOTH(10, "v_exp_f32") BOTH(11, "v_perm_b32")
  BOTH(12, "v_max_f32") BOTH(13, "v_mov_b32") BOTH(14, "v_pk_add_f16") BOTH(16, "v_mov_dpp row_shr") BOTH(17, "v_bfe_u32") BOTH(18, "v_alignbit")
  return 0;
}
