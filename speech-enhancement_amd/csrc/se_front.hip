// Fused STFT / iSTFT front-end for the reference's fixed analysis (n_fft = 400, hop = 100, periodic Hamming window,
// one-sided spectrum: core/function.py:685-703), one launch each:
//
//   se_stft_fused : waveform [B, L] -> compressed planes [B, T, 201, 4] = (|z|', Re z', Im z', 0):
//                   normalise (x * c[b]) + reflect-pad + frame (32 frames unfolded in LDS) -> windowed real DFT as an
//                   fp32-MFMA product with the [400 x 402] (cos | -sin interleaved per bin) matrix streamed from L2 ->
//                   power / log compression in the accumulator registers (the two parts of a bin sit in neighbouring lanes)
//   se_istft_fused: planes -> waveform [B, 100 (T - 1)]: un-compress while staging 32 spectra in LDS -> windowed inverse DFT
//                   (fp32 MFMA, [402 x 400] matrix) -> overlap-add of the 4 frames that cover a hop, envelope division and the
//                   200-sample trim, each output sample written exactly once (frames t0-3 .. t0+28 -> hops t0 .. t0+28)
//
// Both are tiny against the generator (0.1 GFLOP per utterance and transform); what the fusion removes is two launches and
// the [B*T, 404] fp32 intermediate per transform.  32x32x2 fp32 MFMA: exact fp32, 13 waves = the 13 column blocks of 32.
#include "se_common.h"

constexpr int FR_NFFT = 400, FR_HOP = 100, FR_F = 201, FR_TILE = 32, FR_NB = 13, FR_LDA = 401, FR_LDA2 = 405, FR_LDW = 416;
#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

// Wf: [400][416] fp32, column 2f = w[k] cos(2 pi f k / 400), 2f + 1 = -w[k] sin(.), zero beyond 402
__global__ __launch_bounds__(FR_NB * 64) void stft_fused_kernel(const float* __restrict__ x, const float* __restrict__ cs,
                                                               const float* __restrict__ Wf, float* __restrict__ P, int L, int T,
                                                               int comp, float pre_scale) {
  extern __shared__ __attribute__((aligned(16))) float fr[];        // [32 frames][FR_LDA]
  const int b = blockIdx.y, t0 = blockIdx.x * FR_TILE;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float c = cs ? cs[b] : 1.0f;
  const float* xb = x + (long)b * L;
  for (int i = tid; i < FR_TILE * FR_NFFT; i += FR_NB * 64) {
    const int f = i / FR_NFFT, k = i - f * FR_NFFT;
    int p = (t0 + f) * FR_HOP + k - FR_NFFT / 2;                      // sample index before padding
    if (p < 0) p = -p;
    if (p >= L) p = 2 * (L - 1) - p;
    fr[f * FR_LDA + k] = (t0 + f < T) ? xb[p] * c : 0.f;
  }
  __syncthreads();
  f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const int ai = lane & 31, kk = lane >> 5;
  const float* wp = Wf + kk * FR_LDW + wave * 32 + (lane & 31);
  const float* ap = fr + ai * FR_LDA + kk;
#pragma unroll 8
  for (int k0 = 0; k0 < FR_NFFT; k0 += 2) acc = MFMA32(ap[k0], wp[(long)k0 * FR_LDW], acc);
  // C layout: column = lane & 31, rows (r & 3) + 8 (r >> 2) + 4 (lane >> 5); even columns hold Re, odd columns Im of bin col/2
  const int col = wave * 32 + (lane & 31), f = col >> 1, hh = lane >> 5;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const float v = acc[r] * pre_scale, o = __shfl_xor(v, 1, 64);
    const int row = (r & 3) + 8 * (r >> 2) + 4 * hh, t = t0 + row;
    if ((col & 1) == 0 && f < FR_F && t < T) {
      const float re = v, im = o;
      const float mag = sqrtf(re * re + im * im);
      float m2 = mag;
      if (comp == 1) m2 = powf(mag, 0.3f);
      else if (comp == 2) m2 = log1pf(mag);
      const float kq = mag > 0.f ? m2 / mag : 0.f;           // angle(0) = 0: a zero bin stays (0, 0)
      *reinterpret_cast<float4*>(P + (((long)b * T + t) * FR_F + f) * 4) = make_float4(m2, re * kq, im * kq, 0.f);
    }
  }
}

// Wi: [404][416] fp32, row 2f = c_f w[n] cos(2 pi f n / 400) / 400, row 2f + 1 = -c_f w[n] sin(.) / 400 (c_0 = c_200 = 1, else 2)
__global__ __launch_bounds__(FR_NB * 64) void istft_fused_kernel(const float* __restrict__ P, const float* __restrict__ Wi,
                                                                const float* __restrict__ env, float* __restrict__ y, int T, int comp,
                                                                float post_scale) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* As = sm;                                  // [32 spectra][FR_LDA2]: (Re_u, Im_u) interleaved per bin
  float* Cs = sm + FR_TILE * FR_LDA2;              // [32 frames][FR_LDA]
  constexpr int HOPS = FR_TILE - 3;                // hops finished per workgroup
  const int b = blockIdx.y, h0 = 2 + blockIdx.x * HOPS;      // first padded hop of this tile (hop h covers padded samples 100 h ..)
  const int f0 = h0 - 3;                           // first frame needed
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int Lout = FR_HOP * (T - 1);
  for (int i = tid; i < FR_TILE * 202; i += FR_NB * 64) {
    const int fi = i / 202, f = i - fi * 202, t = f0 + fi;
    float ru = 0.f, iu = 0.f;
    if (f < FR_F && t >= 0 && t < T) {
      const float4 p = *reinterpret_cast<const float4*>(P + (((long)b * T + t) * FR_F + f) * 4);
      const float mag = sqrtf(p.y * p.y + p.z * p.z);
      float m2 = mag;
      if (comp == 1) m2 = powf(mag, 1.0f / 0.3f);
      else if (comp == 2) m2 = expm1f(mag);
      const float kq = mag > 0.f ? m2 / mag * post_scale : 0.f;
      ru = p.y * kq; iu = p.z * kq;
    }
    As[fi * FR_LDA2 + 2 * f] = ru;
    As[fi * FR_LDA2 + 2 * f + 1] = iu;
  }
  __syncthreads();
  f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const int ai = lane & 31, kk = lane >> 5;
  const float* wp = Wi + kk * FR_LDW + wave * 32 + (lane & 31);
  const float* ap = As + ai * FR_LDA2 + kk;
#pragma unroll 8
  for (int k0 = 0; k0 < 404; k0 += 2) acc = MFMA32(ap[k0], wp[(long)k0 * FR_LDW], acc);
  const int col = wave * 32 + (lane & 31), hh = lane >> 5;
  if (col < FR_NFFT) {
#pragma unroll
    for (int r = 0; r < 16; ++r) Cs[((r & 3) + 8 * (r >> 2) + 4 * hh) * FR_LDA + col] = acc[r];
  }
  __syncthreads();
  // overlap-add: padded sample sp = 100 h + j gets frame h - d, column 100 d + j, d = 0..3; output sample = sp - 200
  for (int i = tid; i < HOPS * FR_HOP; i += FR_NB * 64) {
    const int hl = i / FR_HOP, j = i - hl * FR_HOP, h = h0 + hl;
    const int s = h * FR_HOP + j - FR_NFFT / 2;
    if (s < Lout) {
      float v = 0.f;
#pragma unroll
      for (int d = 0; d < 4; ++d) v += Cs[(hl + 3 - d) * FR_LDA + FR_HOP * d + j];      // frames outside [0, T) were staged as zeros
      y[(long)b * Lout + s] = v / env[h * FR_HOP + j];
    }
  }
}

extern "C" int se_stft_fused(const float* x, const float* c, const float* Wf, float* P, int B, int L, int n_fft, int hop,
                             int comp, float pre_scale, void* stream) {
  SE_REQUIRE(x && Wf && P && B > 0, "stft_fused: null operand");
  SE_REQUIRE(n_fft == FR_NFFT && hop == FR_HOP, "stft_fused: built for n_fft = 400, hop = 100 (got %d, %d)", n_fft, hop);
  SE_REQUIRE(L > n_fft / 2, "stft_fused: L = %d must exceed the reflect pad", L);       // T = L / hop + 1 frames like torch.stft
  const int T = L / hop + 1;
  const size_t sh = (size_t)FR_TILE * FR_LDA * sizeof(float);
  hipLaunchKernelGGL(stft_fused_kernel, dim3(cdiv(T, FR_TILE), B), dim3(FR_NB * 64), sh, as_stream(stream), x, c, Wf, P, L, T, comp,
                     pre_scale);
  return se_check_launch("se_stft_fused");
}

extern "C" int se_istft_fused(const float* P, const float* Wi, const float* env, float* y, int B, int T, int n_fft, int hop,
                              int comp, float post_scale, void* stream) {
  SE_REQUIRE(P && Wi && env && y && B > 0 && T > 1, "istft_fused: bad arguments");
  SE_REQUIRE(n_fft == FR_NFFT && hop == FR_HOP, "istft_fused: built for n_fft = 400, hop = 100 (got %d, %d)", n_fft, hop);
  const size_t sh = (size_t)FR_TILE * (FR_LDA2 + FR_LDA) * sizeof(float);
  static unsigned raised = 0;
  SE_REQUIRE(se_raise_lds((const void*)istft_fused_kernel, sh, &raised), "istft_fused: cannot raise the dynamic LDS limit");
  const int hops = T - 1;                       // output hops: padded hops 2 .. T
  hipLaunchKernelGGL(istft_fused_kernel, dim3(cdiv(hops, FR_TILE - 3), B), dim3(FR_NB * 64), sh, as_stream(stream), P, Wi, env, y, T,
                     comp, post_scale);
  return se_check_launch("se_istft_fused");
}
