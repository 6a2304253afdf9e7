#include "/root/repo/speech-enhancement_amd/csrc/se_gemm_dev.h"
#include <vector>
#include <cmath>
thread_local char g_se_err[512];
int se_fail(const char*, ...) { return -1; }
int se_check_launch(const char*) { return 0; }
// C[32][32] = A[32][16] * B[32][16]^T with A split in registers, B split in registers
__global__ void k(const float* A, const float* B, float* C, float sa, float sb, int mode) {
  const int lane = threadIdx.x, r = lane & 31, kg = lane >> 5;
  float xa[8], xb[8];
  for (int j = 0; j < 8; ++j) { xa[j] = A[r * 16 + 8 * kg + j]; xb[j] = B[r * 16 + 8 * kg + j]; }
  bf16x8 af[2], bf[2];
  if (mode == 0) { split_planes8_h(xa, sa, af); split_planes8_h(xb, sb, bf); }
  else { split_planes8<2>(xa, af); split_planes8<2>(xb, bf); }
  f32x16 acc;
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  if (mode == 0) { acc = mfma32_<true>(af[0], bf[1], acc); acc = mfma32_<true>(af[1], bf[0], acc); acc = mfma32_<true>(af[0], bf[0], acc); }
  else { acc = mfma32_<false>(af[0], bf[1], acc); acc = mfma32_<false>(af[1], bf[0], acc); acc = mfma32_<false>(af[0], bf[0], acc); }
  const float u = mode == 0 ? 1.f / (sa * sb) : 1.f;
  for (int i = 0; i < 16; ++i) C[((i & 3) + 8 * (i >> 2) + 4 * kg) * 32 + r] = acc[i] * u;
}
int main() {
  std::vector<float> A(512), B(512), C(1024);
  for (int i = 0; i < 512; ++i) { A[i] = sinf(i * 0.37f) * 3.f; B[i] = cosf(i * 0.91f) * 0.1f; }
  float *dA, *dB, *dC;
  hipMalloc(&dA, 2048); hipMalloc(&dB, 2048); hipMalloc(&dC, 4096);
  hipMemcpy(dA, A.data(), 2048, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 2048, hipMemcpyHostToDevice);
  for (int mode = 0; mode < 2; ++mode) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC, 64.f, 8192.f, mode);
    hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost);
    double emax = 0, rmax = 0;
    for (int m = 0; m < 32; ++m) for (int n = 0; n < 32; ++n) {
      double ref = 0; for (int kk = 0; kk < 16; ++kk) ref += (double)A[m * 16 + kk] * B[n * 16 + kk];
      emax = fmax(emax, fabs(C[m * 32 + n] - ref)); rmax = fmax(rmax, fabs(ref));
    }
    printf("mode %d: max err %.3e of max %.3e\n", mode, emax, rmax);
  }
  return 0;
}
