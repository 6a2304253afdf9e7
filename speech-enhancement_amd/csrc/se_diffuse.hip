// Streaming kernels of the CDiffuSE denoiser (models/DiffuSE.py; BASELINE config 5), channels-last [B, L, C] maps, L = 100 T
// samples.  The contractions (dilated k=3 convs, 1x1 convs, the conditioner projections) run on the tap-GEMM kernels; these
// kernels are what sits between them, fused so that every map is read once and written once per layer:
//   diff_upsample : SpectrogramUpsampler stage (ConvTranspose2d(1,1,[3,20], stride [1,10], padding [1,5]) + leaky_relu 0.4)
//   diff_input    : x = relu(w a + b) (input_projection, Conv1d(1, C, 1)) and y = x + d_0 (first diffusion_projection)
//   diff_gate     : z = GroupNorm(conv) + conditioner;  y = sigmoid(z[:C]) * tanh(z[C:])          (ResidualBlock :117-122)
//   diff_mix      : x <- (x + residual) / sqrt 2;  y_next = x + d_next;  skip_sum += GroupNorm(skip)   (:124-127, 155-158)
//   diff_out      : relu + output_projection (Conv1d(C, 1, 1))                                          (:160-161)
#include "se_common.h"

// in [B][F][Tin] -> out: layout 0: [B][F][10 Tin]; layout 1: channels-last [B][10 Tin][ldo] (columns >= F left untouched)
__global__ void diff_upsample_kernel(const float* __restrict__ in, const float* __restrict__ w, const float* __restrict__ bias,
                                     float* __restrict__ out, int F, int Tin, int layout, int ldo, long total) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int Tout = 10 * Tin;
  int b, f, l;
  if (layout == 0) { l = (int)(idx % Tout); f = (int)((idx / Tout) % F); b = (int)(idx / ((long)Tout * F)); }
  else { f = (int)(idx % F); l = (int)((idx / F) % Tout); b = (int)(idx / ((long)Tout * F)); }
  float acc = bias[0];
  // out[f][l] += in[f'][t'] W[kf][kt] with f = f' - 1 + kf, l = 10 t' - 5 + kt: kt = l + 5 - 10 t' in [0, 20) -> two t'
  const int tb = (l + 5) / 10;
#pragma unroll
  for (int dtp = 0; dtp < 2; ++dtp) {
    const int tp = tb - dtp, kt = l + 5 - 10 * tp;
    if (tp < 0 || tp >= Tin || kt < 0 || kt >= 20) continue;
#pragma unroll
    for (int kf = 0; kf < 3; ++kf) {
      const int fp = f + 1 - kf;
      if (fp >= 0 && fp < F) acc += in[((long)b * F + fp) * Tin + tp] * w[kf * 20 + kt];
    }
  }
  acc = acc >= 0.f ? acc : 0.4f * acc;
  if (layout == 0) out[idx] = acc;
  else out[((long)b * Tout + l) * ldo + f] = acc;
}

// one thread = one position x 4 channels
__global__ void diff_input_kernel(const float* __restrict__ audio, const float* __restrict__ w, const float* __restrict__ bias,
                                  const float* __restrict__ d0, int dB, float* __restrict__ x, float* __restrict__ y, long L, int C,
                                  long total, float* __restrict__ y_amax) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = idx < total;                 // (no early exit: the wave maximum below needs every lane)
  if (!live) idx = total - 1;
  const int q = C / 4, c4 = (int)(idx % q) * 4;
  const long pos = idx / q;
  const int b = (int)(pos / L);
  const float a = audio[pos];
  const float4 w4 = *reinterpret_cast<const float4*>(w + c4), b4 = *reinterpret_cast<const float4*>(bias + c4);
  const float4 d4 = *reinterpret_cast<const float4*>(d0 + (long)(dB > 1 ? b : 0) * C + c4);
  float4 xv = make_float4(fmaxf(w4.x * a + b4.x, 0.f), fmaxf(w4.y * a + b4.y, 0.f), fmaxf(w4.z * a + b4.z, 0.f),
                          fmaxf(w4.w * a + b4.w, 0.f));
  const float4 yv = make_float4(xv.x + d4.x, xv.y + d4.y, xv.z + d4.z, xv.w + d4.w);
  if (live) {
    if (x) *reinterpret_cast<float4*>(x + idx * 4) = xv;      // (x == NULL: the one-stream form keeps only y)
    *reinterpret_cast<float4*>(y + idx * 4) = yv;
  }
  if (y_amax) {        // max |y|: the operand scale of the scaled split-fp16 dilated conv that reads y
    const float m = wave_max(live ? fmaxf(fmaxf(fabsf(yv.x), fabsf(yv.y)), fmaxf(fabsf(yv.z), fabsf(yv.w))) : 0.f);
    if ((threadIdx.x & 63) == 0) amax_raise_(y_amax, m);
  }
}

// R [B, L, 2C], ss [B][2C][2] (scale, shift of the GroupNorm), cond [B, L, 2C] -> y [B, L, C]
__global__ void diff_gate_kernel(const float* __restrict__ R, const float* __restrict__ ss, const float* __restrict__ cond,
                                 float* __restrict__ y, long L, int C, long total) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int q = C / 4, c4 = (int)(idx % q) * 4;
  const long pos = idx / q;
  const int b = (int)(pos / L);
  const float* r = R + pos * 2 * C;
  const float* cd = cond + pos * 2 * C;
  const float* s = ss + (long)b * 2 * C * 2;
  const float4 g4 = *reinterpret_cast<const float4*>(r + c4), f4 = *reinterpret_cast<const float4*>(r + C + c4);
  const float4 cg = *reinterpret_cast<const float4*>(cd + c4), cf = *reinterpret_cast<const float4*>(cd + C + c4);
  float o[4];
  const float gv[4] = {g4.x, g4.y, g4.z, g4.w}, fv[4] = {f4.x, f4.y, f4.z, f4.w};
  const float cgv[4] = {cg.x, cg.y, cg.z, cg.w}, cfv[4] = {cf.x, cf.y, cf.z, cf.w};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float zg = gv[j] * s[(c4 + j) * 2] + s[(c4 + j) * 2 + 1] + cgv[j];
    const float zf = fv[j] * s[(C + c4 + j) * 2] + s[(C + c4 + j) * 2 + 1] + cfv[j];
    // hardware exp2 / rcp (~1 ulp each, se_common.h): libm's expf + tanhf made this streaming kernel VALU-bound (3.6 TB/s of 1.31 GB)
    o[j] = sigmoidf_(zg) * (1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.8853900817779268f * zf)));
  }
  *reinterpret_cast<float4*>(y + idx * 4) = make_float4(o[0], o[1], o[2], o[3]);
}

// ONE-STREAM form of the mix (round 5): only y = x + d_step is kept between the layers -- x is y - d_cur (d is a per-(batch entry,
// channel) vector), so the residual stream is read once and written once per layer instead of x read + x written + y written:
// y <- ((y - d_cur) + residual) / sqrt 2 + d_next in place; the last layer (d_next == NULL) writes nothing but the skip sum.
__global__ void diff_mix_y_kernel(float* __restrict__ y, const float* __restrict__ R2, const float* __restrict__ ss,
                                  const float* __restrict__ dc, const float* __restrict__ dn, int dB, float* __restrict__ skip, int first,
                                  long L, int C, long total, float* __restrict__ y_amax) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = idx < total;
  if (!live) idx = total - 1;
  const int q = C / 4, c4 = (int)(idx % q) * 4;
  const long pos = idx / q;
  const int b = (int)(pos / L);
  const float* r = R2 + pos * 2 * C;
  const float* s = ss + (long)b * C * 2;
  const float4 sk = *reinterpret_cast<const float4*>(r + C + c4);
  float ym = 0.f;
  if (dn) {
    const float4 yr = *reinterpret_cast<const float4*>(y + idx * 4);
    const float4 rs = *reinterpret_cast<const float4*>(r + c4);
    const float4 d0 = *reinterpret_cast<const float4*>(dc + (long)(dB > 1 ? b : 0) * C + c4);
    const float4 d1 = *reinterpret_cast<const float4*>(dn + (long)(dB > 1 ? b : 0) * C + c4);
    const float k = 0.70710678118654752f;
    const float4 yv = make_float4(((yr.x - d0.x) + rs.x) * k + d1.x, ((yr.y - d0.y) + rs.y) * k + d1.y, ((yr.z - d0.z) + rs.z) * k + d1.z,
                                  ((yr.w - d0.w) + rs.w) * k + d1.w);
    if (live) *reinterpret_cast<float4*>(y + idx * 4) = yv;
    ym = fmaxf(fmaxf(fabsf(yv.x), fabsf(yv.y)), fmaxf(fabsf(yv.z), fabsf(yv.w)));
  }
  if (y_amax) {
    const float m = wave_max(live ? ym : 0.f);
    if ((threadIdx.x & 63) == 0) amax_raise_(y_amax, m);
  }
  if (!live) return;
  float4 acc = first ? make_float4(0.f, 0.f, 0.f, 0.f) : *reinterpret_cast<const float4*>(skip + idx * 4);
  acc.x += sk.x * s[(c4 + 0) * 2] + s[(c4 + 0) * 2 + 1];
  acc.y += sk.y * s[(c4 + 1) * 2] + s[(c4 + 1) * 2 + 1];
  acc.z += sk.z * s[(c4 + 2) * 2] + s[(c4 + 2) * 2 + 1];
  acc.w += sk.w * s[(c4 + 3) * 2] + s[(c4 + 3) * 2 + 1];
  *reinterpret_cast<float4*>(skip + idx * 4) = acc;
}

// R2 [B, L, 2C] = (residual | skip before its GroupNorm); x updated in place; y_next written when d_next != NULL;
// skip_sum (+)= GroupNorm(skip)
__global__ void diff_mix_kernel(float* __restrict__ x, const float* __restrict__ R2, const float* __restrict__ ss,
                                const float* __restrict__ dn, int dB, float* __restrict__ ynext, float* __restrict__ skip, int first,
                                long L, int C, long total, float* __restrict__ y_amax) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = idx < total;
  if (!live) idx = total - 1;
  const int q = C / 4, c4 = (int)(idx % q) * 4;
  const long pos = idx / q;
  const int b = (int)(pos / L);
  const float* r = R2 + pos * 2 * C;
  const float* s = ss + (long)b * C * 2;
  const float4 xr = *reinterpret_cast<const float4*>(x + idx * 4);
  const float4 rs = *reinterpret_cast<const float4*>(r + c4), sk = *reinterpret_cast<const float4*>(r + C + c4);
  const float k = 0.70710678118654752f;
  float4 xn = make_float4((xr.x + rs.x) * k, (xr.y + rs.y) * k, (xr.z + rs.z) * k, (xr.w + rs.w) * k);
  float ym = 0.f;
  if (dn) {
    const float4 d4 = *reinterpret_cast<const float4*>(dn + (long)(dB > 1 ? b : 0) * C + c4);
    const float4 yv = make_float4(xn.x + d4.x, xn.y + d4.y, xn.z + d4.z, xn.w + d4.w);
    if (live) *reinterpret_cast<float4*>(ynext + idx * 4) = yv;
    ym = fmaxf(fmaxf(fabsf(yv.x), fabsf(yv.y)), fmaxf(fabsf(yv.z), fabsf(yv.w)));
  }
  if (y_amax) {
    const float m = wave_max(live ? ym : 0.f);
    if ((threadIdx.x & 63) == 0) amax_raise_(y_amax, m);
  }
  if (!live) return;
  *reinterpret_cast<float4*>(x + idx * 4) = xn;
  float4 acc = first ? make_float4(0.f, 0.f, 0.f, 0.f) : *reinterpret_cast<const float4*>(skip + idx * 4);
  acc.x += sk.x * s[(c4 + 0) * 2] + s[(c4 + 0) * 2 + 1];
  acc.y += sk.y * s[(c4 + 1) * 2] + s[(c4 + 1) * 2 + 1];
  acc.z += sk.z * s[(c4 + 2) * 2] + s[(c4 + 2) * 2 + 1];
  acc.w += sk.w * s[(c4 + 3) * 2] + s[(c4 + 3) * 2 + 1];
  *reinterpret_cast<float4*>(skip + idx * 4) = acc;
}

// out[pos] = b + sum_c relu(h[pos][c]) w[c]: 16 lanes per position (C = 64: 4 channels per lane)
__global__ void diff_out_kernel(const float* __restrict__ h, const float* __restrict__ w, const float* __restrict__ bias,
                                float* __restrict__ out, int C, long npos) {
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long pos = gid >> 4;
  const int sub = (int)(gid & 15);
  float acc = 0.f;
  if (pos < npos) {
    for (int c = sub * 4; c < C; c += 64) {
      const float4 v = *reinterpret_cast<const float4*>(h + pos * C + c), w4 = *reinterpret_cast<const float4*>(w + c);
      acc += fmaxf(v.x, 0.f) * w4.x + fmaxf(v.y, 0.f) * w4.y + fmaxf(v.z, 0.f) * w4.z + fmaxf(v.w, 0.f) * w4.w;
    }
  }
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if (sub == 0 && pos < npos) out[pos] = acc + bias[0];
}

#define DF_LAUNCH(kernel, n, s, ...) hipLaunchKernelGGL(kernel, dim3(cdiv((n), 256)), dim3(256), 0, as_stream(s), __VA_ARGS__)

extern "C" int se_diff_upsample(const float* in, const float* w, const float* bias, float* out, int B, int F, int Tin, int layout,
                                int ldo, void* stream) {
  SE_REQUIRE(in && w && bias && out && B > 0 && F > 0 && Tin > 0 && (layout == 0 || ldo >= F), "diff_upsample: bad arguments");
  const long total = (long)B * F * Tin * 10;
  DF_LAUNCH(diff_upsample_kernel, total, stream, in, w, bias, out, F, Tin, layout, ldo, total);
  return se_check_launch("se_diff_upsample");
}
extern "C" int se_diff_input_amax(const float* audio, const float* w, const float* bias, const float* d0, int dB, float* x, float* y,
                                  int B, long L, int C, float* y_amax, void* stream) {
  SE_REQUIRE(audio && w && bias && d0 && y && B > 0 && L > 0 && C > 0 && (C % 4) == 0, "diff_input: bad arguments");
  const long total = (long)B * L * (C / 4);
  DF_LAUNCH(diff_input_kernel, total, stream, audio, w, bias, d0, dB, x, y, L, C, total, y_amax);
  return se_check_launch("se_diff_input");
}
extern "C" int se_diff_input(const float* audio, const float* w, const float* bias, const float* d0, int dB, float* x, float* y,
                             int B, long L, int C, void* stream) {
  return se_diff_input_amax(audio, w, bias, d0, dB, x, y, B, L, C, nullptr, stream);
}
extern "C" int se_diff_gate(const float* R, const float* ss, const float* cond, float* y, int B, long L, int C, void* stream) {
  SE_REQUIRE(R && ss && cond && y && B > 0 && L > 0 && C > 0 && (C % 4) == 0, "diff_gate: bad arguments");
  const long total = (long)B * L * (C / 4);
  DF_LAUNCH(diff_gate_kernel, total, stream, R, ss, cond, y, L, C, total);
  return se_check_launch("se_diff_gate");
}
extern "C" int se_diff_mix_amax(float* x, const float* R2, const float* ss, const float* d_next, int dB, float* ynext, float* skip,
                                int first, int B, long L, int C, float* y_amax, void* stream) {
  SE_REQUIRE(x && R2 && ss && skip && (!d_next || ynext) && B > 0 && L > 0 && C > 0 && (C % 4) == 0, "diff_mix: bad arguments");
  const long total = (long)B * L * (C / 4);
  DF_LAUNCH(diff_mix_kernel, total, stream, x, R2, ss, d_next, dB, ynext, skip, first, L, C, total, y_amax);
  return se_check_launch("se_diff_mix");
}
extern "C" int se_diff_mix_y(float* y, const float* R2, const float* ss, const float* d_cur, const float* d_next, int dB, float* skip,
                             int first, int B, long L, int C, float* y_amax, void* stream) {
  SE_REQUIRE(y && R2 && ss && skip && (d_cur || !d_next) && B > 0 && L > 0 && C > 0 && (C % 4) == 0, "diff_mix_y: bad arguments");
  const long total = (long)B * L * (C / 4);
  DF_LAUNCH(diff_mix_y_kernel, total, stream, y, R2, ss, d_cur, d_next, dB, skip, first, L, C, total, y_amax);
  return se_check_launch("se_diff_mix_y");
}
extern "C" int se_diff_mix(float* x, const float* R2, const float* ss, const float* d_next, int dB, float* ynext, float* skip,
                           int first, int B, long L, int C, void* stream) {
  return se_diff_mix_amax(x, R2, ss, d_next, dB, ynext, skip, first, B, L, C, nullptr, stream);
}
extern "C" int se_diff_out(const float* h, const float* w, const float* bias, float* out, long npos, int C, void* stream) {
  SE_REQUIRE(h && w && bias && out && npos > 0 && C > 0 && (C % 64) == 0, "diff_out: C must be a multiple of 64");
  DF_LAUNCH(diff_out_kernel, npos * 16, stream, h, w, bias, out, C, npos);
  return se_check_launch("se_diff_out");
}

// GroupNorm statistics -> per-(batch, channel) scale / shift: stats [B][Ntot][2] fp64 (sum, sum of squares per channel over
// `count_per_channel` positions, from the GEMM epilogue), channels [c_off, c_off + N) in groups of `gsize`
__global__ void group_finalize_kernel(const double* __restrict__ stats, int Ntot, int c_off, int N, int gsize,
                                      double count_per_channel, const float* __restrict__ gamma, const float* __restrict__ beta,
                                      float* __restrict__ ss, float eps, int total) {
  int idx = blockIdx.x * blockDim.x + threadIdx.x;          // (b, channel)
  if (idx >= total) return;
  const int b = idx / N, c = idx - b * N, g0 = (c / gsize) * gsize;
  double s = 0.0, q = 0.0;
  for (int j = 0; j < gsize; ++j) {
    s += stats[((long)b * Ntot + c_off + g0 + j) * 2];
    q += stats[((long)b * Ntot + c_off + g0 + j) * 2 + 1];
  }
  const double cnt = count_per_channel * gsize, mean = s / cnt;
  double var = q / cnt - mean * mean;
  if (var < 0.0) var = 0.0;
  const float rstd = (float)(1.0 / sqrt(var + (double)eps));
  const float sc = rstd * gamma[c];
  ss[(long)idx * 2] = sc;
  ss[(long)idx * 2 + 1] = beta[c] - (float)mean * sc;
}
extern "C" int se_group_finalize(const double* stats, int B, int Ntot, int c_off, int N, int gsize, double count_per_channel,
                                 const float* gamma, const float* beta, float* ss, float eps, void* stream) {
  SE_REQUIRE(stats && gamma && beta && ss && B > 0 && N > 0 && gsize > 0 && (N % gsize) == 0 && c_off >= 0 && c_off + N <= Ntot,
             "group_finalize: bad arguments");
  DF_LAUNCH(group_finalize_kernel, (long)B * N, stream, stats, Ntot, c_off, N, gsize, count_per_channel, gamma, beta, ss, eps, B * N);
  return se_check_launch("se_group_finalize");
}
