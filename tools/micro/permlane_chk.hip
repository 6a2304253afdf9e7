// xor16_ / xor32_ butterfly helpers of se_common.h against __shfl_xor on the device: hipcc --offload-arch=gfx950 -O3 -I ../../speech-enhancement_amd/csrc
#include "se_common.h"
thread_local char g_se_err[512];
int se_fail(const char*, ...) { return 1; }
int se_check_launch(const char*) { return 0; }
__global__ void chk(const float* in, float* out) {
  const int t = threadIdx.x;
  const float v = in[t];
  out[t] = xor32_max_(v) - fmaxf(v, __shfl_xor(v, 32, 64));
  out[64 + t] = xor16_max_(v) - fmaxf(v, __shfl_xor(v, 16, 64));
  out[128 + t] = xor32_sum_(v) - (v + __shfl_xor(v, 32, 64));
  out[192 + t] = xor16_sum_(v) - (v + __shfl_xor(v, 16, 64));
}
int main() {
  float h[64], r[256], *d, *o;
  for (int i = 0; i < 64; ++i) h[i] = (float)((i * 37 + 11) % 64) - 20.5f;
  hipMalloc(&d, 256); hipMalloc(&o, 1024);
  hipMemcpy(d, h, 256, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(chk, dim3(1), dim3(64), 0, 0, d, o);
  hipMemcpy(r, o, 1024, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 256; ++i) bad += r[i] != 0.f;
  printf("permlane swap butterflies vs __shfl_xor: %d mismatches of 256\n", bad);
  return bad != 0;
}
