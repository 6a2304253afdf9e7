// Elementwise / small-reduction kernels of the hot path (gfx950): STFT front-end glue (normalise, reflect-pad,
// power (un)compression, overlap-add), generator output assembly, GLU backward, spectral / time losses with
// their gradient seeds, flat-buffer optimizers.  All are HBM-bound streaming kernels: 16 B per lane where the
// layout allows, grid-stride over at most 2048 workgroups, block reductions through wave shuffles + LDS and one
// fp64 atomic per workgroup.
#include "se_common.h"

static __device__ __forceinline__ void block_sum_atomic(double v, double* dst) {
  __shared__ double part[4];
  v = wave_sum_d(v);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(dst, part[0] + part[1] + part[2] + part[3]);
  __syncthreads();
}
static inline int gs_grid(long n, int per_block) {
  long nb = (n + per_block - 1) / per_block;
  return (int)(nb > 2048 ? 2048 : (nb < 1 ? 1 : nb));
}
// reductions that end in block_sum_atomic: every workgroup adds to the SAME one or two doubles -- 2048 workgroups queued 4096
// atomics on one cache line (spec_loss: 54 us for 33 MB); one workgroup per CU
static inline int red_grid(long n, int per_block) {
  long nb = (n + per_block - 1) / per_block;
  return (int)(nb > 256 ? 256 : (nb < 1 ? 1 : nb));
}

// ---------------------------------------------------------------------------------------------
// c[b] = sqrt(L / sum x^2)   (normalize_batch, core/function.py:647-659; inference_gan.py:79-81)
__global__ __launch_bounds__(256) void clip_scale_kernel(const float* __restrict__ x, float* __restrict__ c, int L) {
  __shared__ double part[4];
  const float* xb = x + (long)blockIdx.x * L;
  // 16-B loads, four of them requested per trip (one 4-B load per trip left 125 dependent round trips: 52 us at the head of
  // every step); the tail and a base that is not 16-B aligned (L % 4 != 0) take the scalar loop
  double s = 0.0;
  int i0 = 0;
  if ((L & 3) == 0 && (((size_t)xb) & 15) == 0) {
    const int L4 = L >> 2;
    const float4* x4 = reinterpret_cast<const float4*>(xb);
    int i = threadIdx.x;
    for (; i + 768 < L4; i += 1024) {
      const float4 a = x4[i], b = x4[i + 256], c4 = x4[i + 512], d = x4[i + 768];
      s += ((double)a.x * a.x + (double)a.y * a.y) + ((double)a.z * a.z + (double)a.w * a.w);
      s += ((double)b.x * b.x + (double)b.y * b.y) + ((double)b.z * b.z + (double)b.w * b.w);
      s += ((double)c4.x * c4.x + (double)c4.y * c4.y) + ((double)c4.z * c4.z + (double)c4.w * c4.w);
      s += ((double)d.x * d.x + (double)d.y * d.y) + ((double)d.z * d.z + (double)d.w * d.w);
    }
    for (; i < L4; i += 256) { const float4 a = x4[i]; s += ((double)a.x * a.x + (double)a.y * a.y) + ((double)a.z * a.z + (double)a.w * a.w); }
    i0 = L;
  }
  for (int i = i0 + threadIdx.x; i < L; i += 256) { float v = xb[i]; s += (double)v * v; }
  s = wave_sum_d(s);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) c[blockIdx.x] = (float)sqrt((double)L / (part[0] + part[1] + part[2] + part[3]));
}

// xp[b][i] = c[b] * x[b][reflect(i - pad)]   (center=True reflect padding of torch.stft)
__global__ void reflect_pad_scale_kernel(const float* __restrict__ x, const float* __restrict__ c, float* __restrict__ xp,
                                         int L, int pad, long total) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  int Lp = L + 2 * pad;
  int b = (int)(idx / Lp), i = (int)(idx % Lp) - pad;
  if (i < 0) i = -i;
  if (i >= L) i = 2 * (L - 1) - i;
  xp[idx] = x[(long)b * L + i] * (c ? c[b] : 1.0f);
}

// frames R[row][ldr]: re at column f, im at column F+f  ->  planes P[row][f][4] = (mag', re', im', 0) with the
// power / log compression of power_compress (core/function.py:625-634)
__global__ void compress_planes_kernel(const float* __restrict__ R, int ldr, float* __restrict__ P, long rows, int F,
                                       int comp, float pre_scale) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows * F) return;
  long row = idx / F;
  int f = (int)(idx - row * F);
  float re = R[row * ldr + f] * pre_scale, im = R[row * ldr + F + f] * pre_scale;
  float mag = sqrtf(re * re + im * im), m2 = mag;
  if (comp == 1) m2 = powf(mag, 0.3f);
  else if (comp == 2) m2 = log1pf(mag);
  float k = mag > 0.f ? m2 / mag : 0.f;
  // angle(0) = 0 -> (cos, sin) = (1, 0): a zero bin stays (0, 0) because m2 = 0 for every mode
  *reinterpret_cast<float4*>(P + idx * 4) = make_float4(m2, re * k, im * k, 0.f);
}

// planes P[row][f][4] (compressed re at [1], im at [2]) -> un-compressed GEMM operand A[row][lda]: re_u | im_u
// (power_uncompress, core/function.py:636-645)
__global__ void uncompress_rows_kernel(const float* __restrict__ P, float* __restrict__ A, int lda, long rows, int F,
                                       int comp, float post_scale) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows * F) return;
  long row = idx / F;
  int f = (int)(idx - row * F);
  float4 p = *reinterpret_cast<const float4*>(P + idx * 4);
  float mag = sqrtf(p.y * p.y + p.z * p.z), m2 = mag;
  if (comp == 1) m2 = powf(mag, 1.0f / 0.3f);
  else if (comp == 2) m2 = expm1f(mag);
  float k = mag > 0.f ? m2 / mag * post_scale : 0.f;
  A[row * lda + f] = p.y * k;
  A[row * lda + F + f] = p.z * k;
}

// backward of the 'pow' un-compression u = z |z|^a (a = 1/0.3 - 1): dP[.., 1:3] (+)= J^T dA
__global__ void uncompress_rows_bwd_kernel(const float* __restrict__ P, const float* __restrict__ dA, int lda,
                                           float* __restrict__ dP, long rows, int F, int comp, float post_scale) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows * F) return;
  long row = idx / F;
  int f = (int)(idx - row * F);
  float4 p = *reinterpret_cast<const float4*>(P + idx * 4);
  float x = p.y, y = p.z;
  float du = dA[row * lda + f] * post_scale, dv = dA[row * lda + F + f] * post_scale;
  float r2 = x * x + y * y, r = sqrtf(r2);
  float gx = 0.f, gy = 0.f;
  if (r > 0.f) {
    float h, hp;     // u = z * h(r);  hp = h'(r) / r
    if (comp == 1) { const float a = 1.0f / 0.3f - 1.0f; h = powf(r, a); hp = a * powf(r, a - 2.0f); }
    else if (comp == 2) { float e = expm1f(r); h = e / r; hp = ((e + 1.0f) * r - e) / (r2 * r); }
    else { h = 1.0f; hp = 0.f; }
    float t = (du * x + dv * y) * hp;
    gx = du * h + t * x;
    gy = dv * h + t * y;
  }
  float4 o = *reinterpret_cast<float4*>(dP + idx * 4);
  o.y += gx; o.z += gy;
  *reinterpret_cast<float4*>(dP + idx * 4) = o;
}

// overlap-add of windowed frames Fr[b][t][n_fft] at hop, divided by the window envelope, trimmed by n_fft/2
// (torch.istft, core/function.py:701-702).  env[Lp] is precomputed on the host.
__global__ void ola_kernel(const float* __restrict__ Fr, const float* __restrict__ env, float* __restrict__ y,
                           int T, int n_fft, int hop, int L, int trim, long total) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  int b = (int)(idx / L), s = (int)(idx % L) + trim;
  int t1 = s / hop;
  float acc = 0.f;
  for (int t = t1; t >= 0 && s - t * hop < n_fft; --t)
    if (t < T) acc += Fr[((long)b * T + t) * n_fft + (s - t * hop)];
  y[idx] = env ? acc / env[s] : acc;
}

// backward of reflect_pad_scale: dx[b][i] = c[b] * (dxp[pad+i] + mirrored contributions)
__global__ void reflect_pad_bwd_kernel(const float* __restrict__ dxp, const float* __restrict__ c, float* __restrict__ dx,
                                       int L, int pad, long total) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  int Lp = L + 2 * pad;
  int b = (int)(idx / L), i = (int)(idx % L);
  const float* d = dxp + (long)b * Lp;
  float acc = d[pad + i];
  if (i >= 1 && i <= pad) acc += d[pad - i];
  if (i <= L - 2 && i >= L - 1 - pad) acc += d[pad + 2 * (L - 1) - i];
  dx[idx] = acc * (c ? c[b] : 1.0f);
}

// backward of compress_planes: dP planes (d mag', d re', d im') -> dR[row][ldr] (d re | d im of the raw DFT)
__global__ void compress_planes_bwd_kernel(const float* __restrict__ R, int ldr, const float* __restrict__ dP,
                                           float* __restrict__ dR, long rows, int F, int comp, float pre_scale) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows * F) return;
  long row = idx / F;
  int f = (int)(idx - row * F);
  float x = R[row * ldr + f] * pre_scale, y = R[row * ldr + F + f] * pre_scale;
  float4 d = *reinterpret_cast<const float4*>(dP + idx * 4);
  float r2 = x * x + y * y, r = sqrtf(r2);
  float gx = 0.f, gy = 0.f;
  if (r > 0.f) {     // the reference's pow(0, 0.3) backward is NaN at an exactly-zero bin; here that bin gets 0
    float gp, h, hp;   // m2 = g(r), gp = g'(r);  z' = z * h(r), h = g/r, hp = h'(r)/r
    if (comp == 1) { gp = 0.3f * powf(r, -0.7f); h = powf(r, -0.7f); hp = -0.7f * powf(r, -2.7f); }
    else if (comp == 2) { float lg = log1pf(r); gp = 1.0f / (1.0f + r); h = lg / r; hp = (gp * r - lg) / (r2 * r); }
    else { gp = 1.0f; h = 1.0f; hp = 0.f; }
    float t = (d.y * x + d.z * y) * hp + d.x * gp / r;
    gx = d.y * h + t * x;
    gy = d.z * h + t * y;
  }
  dR[row * ldr + f] = gx * pre_scale;
  dR[row * ldr + F + f] = gy * pre_scale;
}
// transpose of ola: dFr[b][t][k] = dy[b][t*hop + k - n_fft/2] / env[t*hop + k]
__global__ void ola_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ env, float* __restrict__ dFr,
                               int T, int n_fft, int hop, int L, long total) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  int k = (int)(idx % n_fft);
  long bt = idx / n_fft;
  int t = (int)(bt % T), b = (int)(bt / T);
  int sp = t * hop + k, s = sp - n_fft / 2;
  dFr[idx] = (s >= 0 && s < L) ? dy[(long)b * L + s] / env[sp] : 0.f;
}

// ---------------------------------------------------------------------------------------------
// generator output (models/generator.py:158-167): est = mask * noisy + complex_out;  planes (|est|, re, im, 0)
__global__ void assemble_kernel(const float* __restrict__ mask, int ldm, const float* __restrict__ nin,
                                const float* __restrict__ cplx, float* __restrict__ est, long n) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n) return;
  float m = mask[idx * ldm];
  float4 x = *reinterpret_cast<const float4*>(nin + idx * 4);
  float4 c = *reinterpret_cast<const float4*>(cplx + idx * 4);
  float re = m * x.y + c.x, im = m * x.z + c.y;
  *reinterpret_cast<float4*>(est + idx * 4) = make_float4(sqrtf(re * re + im * im), re, im, 0.f);
}
// d_est planes (dmag, dre, dim, -) -> dmask (ld ldm, channel 0), dcplx planes (d0, d1, 0, 0)
__global__ void assemble_bwd_kernel(const float* __restrict__ est, const float* __restrict__ dest,
                                    const float* __restrict__ nin, float* __restrict__ dmask, int ldm,
                                    float* __restrict__ dcplx, long n) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n) return;
  float4 e = *reinterpret_cast<const float4*>(est + idx * 4);
  float4 d = *reinterpret_cast<const float4*>(dest + idx * 4);
  float4 x = *reinterpret_cast<const float4*>(nin + idx * 4);
  float k = e.x > 0.f ? d.x / e.x : 0.f;
  float dre = d.y + k * e.y, dim = d.z + k * e.z;
  dmask[idx * ldm] = dre * x.y + dim * x.z;
  *reinterpret_cast<float4*>(dcplx + idx * 4) = make_float4(dre, dim, 0.f, 0.f);
}

// MaskDecoder tail (generator.py:110-112): v = u*w + b (final_conv 1x1, 1->1), mask = PReLU_F(v) (slope per f)
__global__ void mask_tail_kernel(const float* __restrict__ U, int ldu, const float* __restrict__ wb,
                                 const float* __restrict__ slope, float* __restrict__ M, long n, int F) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n) return;
  float v = U[idx * ldu] * wb[0] + wb[1];
  M[idx] = v >= 0.f ? v : v * slope[idx % F];
}
// backward: dU (channel 0 of an ld-4 buffer, channels 1..3 zeroed), dwb[2] += , dslope[F] +=
// A thread owns ONE frequency column f of a run of rows: the PReLU(201) slope gradient is summed in a register and leaves with one
// atomic per (thread, run).  (One atomic per element with v < 0 -- half a million on 201 addresses in 7 cache lines -- made this
// 4 MB kernel take 152 us at the head of the generator backward.)
__global__ __launch_bounds__(256) void mask_tail_bwd_kernel(const float* __restrict__ U, int ldu, const float* __restrict__ wb,
                                                            const float* __restrict__ slope, const float* __restrict__ dM,
                                                            float* __restrict__ dU, double* __restrict__ dwb,
                                                            float* __restrict__ dslope, long n, int F, int rows_per_block) {
  double sw = 0.0, sb = 0.0;
  const long nrows = n / F;
  const long r0 = (long)blockIdx.y * rows_per_block;
  const long r1 = r0 + rows_per_block < nrows ? r0 + rows_per_block : nrows;
  const float w0 = wb[0], w1 = wb[1];
  for (int f = blockIdx.x * 256 + threadIdx.x; f < F; f += gridDim.x * 256) {
    const float sl = slope[f];
    float ds = 0.f;
    for (long r = r0; r < r1; ++r) {
      const long idx = r * F + f;
      const float u = U[idx * ldu];
      const float v = u * w0 + w1;
      const float dm = dM[idx];
      const float dv = v >= 0.f ? dm : dm * sl;
      ds += v < 0.f ? dm * v : 0.f;
      sw += (double)dv * u; sb += dv;
      *reinterpret_cast<float4*>(dU + idx * 4) = make_float4(dv * w0, 0.f, 0.f, 0.f);
    }
    if (ds != 0.f) atomicAdd(&dslope[f], ds);
  }
  block_sum_atomic(sw, dwb);
  block_sum_atomic(sb, dwb + 1);
}

// GLU backward (models/conformer.py:30-37): Z = [a | g] (2H), dU (H) -> dZ.  G != NULL: Z = the GLU result u = a sigmoid(g) [M][H]
// and G = the gate half [M][H] (d g = dU u (1 - sigmoid(g)): the value half a is not needed)
__global__ void glu_bwd_kernel(const float* __restrict__ Z, const float* __restrict__ dU, float* __restrict__ dZ,
                               long M, int H, float* amax_out, const float* __restrict__ G) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;      // one float4 of H per thread
  int hq = H >> 2;
  const bool live = idx < M * hq;
  if (!live) idx = M * hq - 1;           // (no early return: the wave-wide maximum below needs every lane)
  long row = idx / hq;
  int q = (int)(idx - row * hq);
  float4 a = *reinterpret_cast<const float4*>(Z + row * (G ? 1 : 2) * H + q * 4);
  float4 g = G ? *reinterpret_cast<const float4*>(G + row * H + q * 4) : *reinterpret_cast<const float4*>(Z + row * 2 * H + H + q * 4);
  float4 d = *reinterpret_cast<const float4*>(dU + row * H + q * 4);
  float av[4] = {a.x, a.y, a.z, a.w}, gv[4] = {g.x, g.y, g.z, g.w}, dv[4] = {d.x, d.y, d.z, d.w}, da[4], dg[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float s = sigmoidf_(gv[j]);
    da[j] = dv[j] * s;
    dg[j] = dv[j] * av[j] * (G ? 1.f : s) * (1.f - s);
  }
  if (live) {
    st4_stream_(dZ + row * 2 * H + q * 4, make_float4(da[0], da[1], da[2], da[3]));
    st4_stream_(dZ + row * 2 * H + H + q * 4, make_float4(dg[0], dg[1], dg[2], dg[3]));
  }
  if (amax_out) {
    float m = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) m = fmaxf(m, fmaxf(fabsf(da[j]), fabsf(dg[j])));
    m = wave_max(live ? m : 0.f);
    if ((threadIdx.x & 63) == 0) amax_raise_(amax_out, m);
  }
}

// ---------------------------------------------------------------------------------------------
// MergeBlock gate of the TSC-diffusion hybrid (models/tsc_diffusion.py:34-35): Y [M, C] (gate | filter halves, ld 2C) ->
// G [M, C] = sigmoid(gate) * tanh(filter)
__global__ void gate_tanh_kernel(const float* __restrict__ Y, float* __restrict__ G, long M, int C) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;      // one float4 of C per thread
  const int cq = C >> 2;
  if (idx >= M * cq) return;
  const long row = idx / cq;
  const int q = (int)(idx - row * cq);
  const float4 a = *reinterpret_cast<const float4*>(Y + row * 2 * C + q * 4);
  const float4 f = *reinterpret_cast<const float4*>(Y + row * 2 * C + C + q * 4);
  auto th = [](float x) { const float e = __builtin_amdgcn_exp2f(-2.885390081777927f * fabsf(x)); const float t = (1.f - e) * __builtin_amdgcn_rcpf(1.f + e); return x < 0.f ? -t : t; };
  *reinterpret_cast<float4*>(G + row * C + q * 4) = make_float4(sigmoidf_(a.x) * th(f.x), sigmoidf_(a.y) * th(f.y),
                                                                sigmoidf_(a.z) * th(f.z), sigmoidf_(a.w) * th(f.w));
}

// backward of the gate: dY [M, 2C] = (dG tanh(f) s (1 - s) | dG s (1 - tanh(f)^2)), s = sigmoid(gate)   (training of the hybrid,
// core/function.py:453-532)
__global__ void gate_tanh_bwd_kernel(const float* __restrict__ Y, const float* __restrict__ dG, float* __restrict__ dY, long M, int C) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;      // one float4 of C per thread
  const int cq = C >> 2;
  if (idx >= M * cq) return;
  const long row = idx / cq;
  const int q = (int)(idx - row * cq);
  const float4 a = *reinterpret_cast<const float4*>(Y + row * 2 * C + q * 4);
  const float4 f = *reinterpret_cast<const float4*>(Y + row * 2 * C + C + q * 4);
  const float4 d = *reinterpret_cast<const float4*>(dG + row * C + q * 4);
  auto th = [](float x) { const float e = __builtin_amdgcn_exp2f(-2.885390081777927f * fabsf(x)); const float t = (1.f - e) * __builtin_amdgcn_rcpf(1.f + e); return x < 0.f ? -t : t; };
  const float av[4] = {a.x, a.y, a.z, a.w}, fv[4] = {f.x, f.y, f.z, f.w}, dv[4] = {d.x, d.y, d.z, d.w};
  float da[4], df[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float s_ = sigmoidf_(av[j]), t = th(fv[j]);
    da[j] = dv[j] * t * s_ * (1.f - s_);
    df[j] = dv[j] * s_ * (1.f - t * t);
  }
  *reinterpret_cast<float4*>(dY + row * 2 * C + q * 4) = make_float4(da[0], da[1], da[2], da[3]);
  *reinterpret_cast<float4*>(dY + row * 2 * C + C + q * 4) = make_float4(df[0], df[1], df[2], df[3]);
}

// ---------------------------------------------------------------------------------------------
// spectral losses on planes (mag, re, im, -): sums[0] += sum (mag-mag')^2, sums[1] += sum (re-re')^2 + (im-im')^2
__global__ __launch_bounds__(256) void spec_loss_kernel(const float* __restrict__ A, const float* __restrict__ Bp,
                                                        double* __restrict__ sums, long n) {
  double sm = 0.0, sr = 0.0;
#pragma unroll 4
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < n; idx += (long)gridDim.x * 256) {
    float4 a = *reinterpret_cast<const float4*>(A + idx * 4);
    float4 b = *reinterpret_cast<const float4*>(Bp + idx * 4);
    float dm = a.x - b.x, dr = a.y - b.y, di = a.z - b.z;
    sm += (double)dm * dm;
    sr += (double)dr * dr + (double)di * di;
  }
  block_sum_atomic(sm, sums);
  block_sum_atomic(sr, sums + 1);
}
// dA planes = (kmag*(mag-mag'), kri*(re-re'), kri*(im-im'), 0)   [kmag = 2*w_mag/N * upstream, ...]
__global__ void spec_loss_bwd_kernel(const float* __restrict__ A, const float* __restrict__ Bp, float* __restrict__ dA,
                                     const float* __restrict__ up, float cmag, float cri, long n, int accumulate) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n) return;
  const float kmag = up[0] * cmag, kri = up[1] * cri;     // upstream gradients stay on the device
  float4 a = *reinterpret_cast<const float4*>(A + idx * 4);
  float4 b = *reinterpret_cast<const float4*>(Bp + idx * 4);
  float4 o = make_float4(kmag * (a.x - b.x), kri * (a.y - b.y), kri * (a.z - b.z), 0.f);
  if (accumulate) {
    float4 p = *reinterpret_cast<float4*>(dA + idx * 4);
    o.x += p.x; o.y += p.y; o.z += p.z;
  }
  *reinterpret_cast<float4*>(dA + idx * 4) = o;
}
// sums[0] += sum |a - b| over rows of length L (a: row stride lda, b: row stride ldb)
__global__ __launch_bounds__(256) void l1_loss_kernel(const float* __restrict__ a, long lda, const float* __restrict__ b,
                                                      long ldb, double* __restrict__ sums, int L, long n) {
  double s = 0.0;
#pragma unroll 4
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < n; idx += (long)gridDim.x * 256) {
    long r = idx / L; int i = (int)(idx - r * L);
    s += fabsf(a[r * lda + i] - b[r * ldb + i]);
  }
  block_sum_atomic(s, sums);
}
__global__ void l1_loss_bwd_kernel(const float* __restrict__ a, long lda, const float* __restrict__ b, long ldb,
                                   float* __restrict__ da, const float* __restrict__ up, float ck, int L, long n) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n) return;
  const float k = up[0] * ck;
  long r = idx / L; int i = (int)(idx - r * L);
  float d = a[r * lda + i] - b[r * ldb + i];
  da[idx] = d > 0.f ? k : (d < 0.f ? -k : 0.f);
}

// ---------------------------------------------------------------------------------------------
// flat-buffer optimizers (torch.optim.AdamW / SGD(nesterov) as built by core/optimizer.py:33-36)
__global__ void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                             float* __restrict__ v, long n, float lr, float b1, float b2, float eps, float wd,
                             float bc1, float bc2) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n) return;
  float gi = g[idx], pi = p[idx];
  pi *= 1.0f - lr * wd;
  float mi = b1 * m[idx] + (1.0f - b1) * gi;
  float vi = b2 * v[idx] + (1.0f - b2) * gi * gi;
  m[idx] = mi; v[idx] = vi;
  float denom = sqrtf(vi) / sqrtf(bc2) + eps;
  p[idx] = pi - (lr / bc1) * mi / denom;
}
__global__ void sgd_nesterov_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ buf, long n,
                                    float lr, float momentum, int first) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n) return;
  float gi = g[idx];
  float bi = first ? gi : momentum * buf[idx] + gi;
  buf[idx] = bi;
  p[idx] -= lr * (gi + momentum * bi);
}
// out[0] += dot(a, b)   (self-correcting discriminator weights, core/function.py:719-732)
__global__ __launch_bounds__(256) void dot_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                  double* __restrict__ out, long n) {
  double s = 0.0;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < n; idx += (long)gridDim.x * 256) s += (double)a[idx] * b[idx];
  block_sum_atomic(s, out);
}
// y = alpha*a + beta*b + gamma*c
__global__ void axpbypcz_kernel(const float* a, const float* b, const float* c, float* y, float alpha, float beta,
                                float gamma, long n) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n) return;
  y[idx] = alpha * a[idx] + beta * b[idx] + gamma * c[idx];
}


// ---------------------------------------------------------------------------------------------
// Global-norm gradient clipping on a flat buffer (torch.nn.utils.clip_grad_norm_, core/function.py:275-276, 311-312):
// g *= min(1, max_norm / (sqrt(sum of the nsum partial sums of squares) + 1e-6)).  The sums are device scalars produced
// by se_dot(g, g) on each flat buffer of the model -> no host round trip.
__global__ void grad_clip_kernel(float* __restrict__ g, long n, const double* __restrict__ sums, int nsum, float max_norm) {
  long idx = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (idx >= n) return;
  double tot = 0.0;
  for (int i = 0; i < nsum; ++i) tot += sums[i];
  float coef = fminf(1.0f, max_norm / ((float)sqrt(tot) + 1e-6f));
  if (idx + 4 <= n) {
    float4 v = *reinterpret_cast<float4*>(g + idx);
    v.x *= coef; v.y *= coef; v.z *= coef; v.w *= coef;
    *reinterpret_cast<float4*>(g + idx) = v;
  } else {
    for (long i = idx; i < n; ++i) g[i] *= coef;
  }
}

// ---------------------------------------------------------------------------------------------
// LARS / Lamb of core/optimizer.py:63-238 on the flat buffers.  Both need one pair of norms PER PARAMETER TENSOR
// (trust ratio): the flat buffer is described by seg_off[nseg + 1] (element offsets of the tensors, in order); the grid is
// (chunks of SEG_CHUNK elements, nseg) and a workgroup whose chunk lies beyond its tensor exits at once (1.8 M elements
// in 335 tensors: the largest has 73 728 elements = 72 chunks).  Phase 1 accumulates norms[seg] = (sum p^2, sum u^2) in
// fp64 (one atomic pair per workgroup), phase 2 applies the update.
#define SEG_CHUNK 1024
// mode 0 (LARS, :71-113): u = g + wd * p                       (only called for the ndim > 1 group)
// mode 1 (Lamb, :176-238): m, v updated in place; u = (m / bc1) / (sqrt(v) / sqrt(bc2) + eps) + wd * p, with
//         g divided by clip = max(1, ||g||_global / max_grad_norm) (gsum: nsum partial sums of squares; nsum == 0: no clip)
__global__ __launch_bounds__(256) void seg_norms_kernel(const float* __restrict__ p, const float* __restrict__ g,
                                                        float* __restrict__ m, float* __restrict__ v,
                                                        const long* __restrict__ seg_off, double* __restrict__ norms,
                                                        int mode, float wd, float b1, float b2, float beta3, float eps,
                                                        float bc1, float rbc2, const double* __restrict__ gsum, int nsum,
                                                        float max_gn) {
  const int seg = blockIdx.y;
  const long lo = seg_off[seg], hi = seg_off[seg + 1];
  const long base = lo + (long)blockIdx.x * SEG_CHUNK;
  if (base >= hi) return;
  float rclip = 1.0f;
  if (mode == 1 && nsum > 0) {      // max_grad_norm == 0 zeroes every gradient, exactly like the reference's g / (gn / 0)
    double tot = 0.0;
    for (int i = 0; i < nsum; ++i) tot += gsum[i];
    float gn = (float)sqrt(tot);
    rclip = gn > max_gn ? max_gn / gn : 1.0f;
  }
  double sp = 0.0, su = 0.0;
  for (long i = base + threadIdx.x; i < hi && i < base + SEG_CHUNK; i += 256) {
    float pi = p[i], gi = g[i], u;
    if (mode == 0) {
      u = gi + wd * pi;
    } else {
      gi *= rclip;
      float mi = b1 * m[i] + beta3 * gi;
      float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
      m[i] = mi; v[i] = vi;
      u = (mi / bc1) / (sqrtf(vi) * rbc2 + eps) + wd * pi;
    }
    sp += (double)pi * pi;
    su += (double)u * u;
  }
  __shared__ double part[8];
  sp = wave_sum_d(sp); su = wave_sum_d(su);
  if ((threadIdx.x & 63) == 0) { part[threadIdx.x >> 6] = sp; part[4 + (threadIdx.x >> 6)] = su; }
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicAdd(norms + 2 * seg, part[0] + part[1] + part[2] + part[3]);
    atomicAdd(norms + 2 * seg + 1, part[4] + part[5] + part[6] + part[7]);
  }
}
// mode 0 LARS: q = trust * ||p|| / ||u|| (1 where either norm is 0); mu = momentum * mu + u * q; p -= lr * mu   (m == mu)
//        `adapt` = 0: the 1-D group -- u = g, q = 1 (no weight decay, no trust ratio, :93-103)
// mode 1 Lamb: trust = ||p|| / ||u|| (1 where either norm is 0; min(.,1) if trust_clip) when adapt, else 1; p -= lr * trust * u
__global__ __launch_bounds__(256) void seg_apply_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                        float* __restrict__ m, const float* __restrict__ v,
                                                        const long* __restrict__ seg_off, const double* __restrict__ norms,
                                                        int mode, int adapt, float lr, float wd, float momentum,
                                                        float trust_coef, float eps, float bc1, float rbc2, int trust_clip) {
  const int seg = blockIdx.y;
  const long lo = seg_off[seg], hi = seg_off[seg + 1];
  const long base = lo + (long)blockIdx.x * SEG_CHUNK;
  if (base >= hi) return;
  float q = 1.0f;
  if (adapt) {
    float pn = (float)sqrt(norms[2 * seg]), un = (float)sqrt(norms[2 * seg + 1]);
    if (pn > 0.f && un > 0.f) q = (mode == 0 ? trust_coef : 1.0f) * pn / un;
    if (mode == 1 && trust_clip) q = fminf(q, 1.0f);
  }
  for (long i = base + threadIdx.x; i < hi && i < base + SEG_CHUNK; i += 256) {
    float pi = p[i];
    if (mode == 0) {
      float u = adapt ? (g[i] + wd * pi) * q : g[i];
      float mu = momentum * m[i] + u;
      m[i] = mu;
      p[i] = pi - lr * mu;
    } else {
      float u = (m[i] / bc1) / (sqrtf(v[i]) * rbc2 + eps) + wd * pi;
      p[i] = pi - lr * q * u;
    }
  }
}

// =============================================================================================
#define EW_LAUNCH(kernel, n, s, ...) hipLaunchKernelGGL(kernel, dim3(cdiv((n), 256)), dim3(256), 0, as_stream(s), __VA_ARGS__)

extern "C" int se_clip_scale(const float* x, float* c, int B, int L, void* stream) {
  SE_REQUIRE(x && c && B > 0 && L > 0, "clip_scale: bad arguments");
  hipLaunchKernelGGL(clip_scale_kernel, dim3(B), dim3(256), 0, as_stream(stream), x, c, L);
  return se_check_launch("se_clip_scale");
}
extern "C" int se_reflect_pad_scale(const float* x, const float* c, float* xp, int B, int L, int pad, void* stream) {
  SE_REQUIRE(x && xp && B > 0 && L > pad && pad >= 0, "reflect_pad_scale: bad arguments (L=%d pad=%d)", L, pad);
  long total = (long)B * (L + 2 * pad);
  EW_LAUNCH(reflect_pad_scale_kernel, total, stream, x, c, xp, L, pad, total);
  return se_check_launch("se_reflect_pad_scale");
}
extern "C" int se_compress_planes(const float* R, int ldr, float* P, long rows, int F, int comp, float pre_scale,
                                  void* stream) {
  SE_REQUIRE(R && P && rows > 0 && F > 0 && ldr >= 2 * F, "compress_planes: bad arguments");
  EW_LAUNCH(compress_planes_kernel, rows * F, stream, R, ldr, P, rows, F, comp, pre_scale);
  return se_check_launch("se_compress_planes");
}
extern "C" int se_uncompress_rows(const float* P, float* A, int lda, long rows, int F, int comp, float post_scale,
                                  void* stream) {
  SE_REQUIRE(P && A && rows > 0 && F > 0 && lda >= 2 * F, "uncompress_rows: bad arguments");
  EW_LAUNCH(uncompress_rows_kernel, rows * F, stream, P, A, lda, rows, F, comp, post_scale);
  return se_check_launch("se_uncompress_rows");
}
extern "C" int se_uncompress_rows_bwd(const float* P, const float* dA, int lda, float* dP, long rows, int F, int comp,
                                      float post_scale, void* stream) {
  SE_REQUIRE(P && dA && dP && rows > 0 && F > 0 && lda >= 2 * F, "uncompress_rows_bwd: bad arguments");
  EW_LAUNCH(uncompress_rows_bwd_kernel, rows * F, stream, P, dA, lda, dP, rows, F, comp, post_scale);
  return se_check_launch("se_uncompress_rows_bwd");
}
extern "C" int se_ola(const float* Fr, const float* env, float* y, int B, int T, int n_fft, int hop, int trim,
                      int L, void* stream) {
  SE_REQUIRE(Fr && y && B > 0 && T > 0 && hop > 0 && n_fft >= hop, "ola: bad arguments");
  SE_REQUIRE(L > 0 && trim >= 0 && trim + L <= n_fft + hop * (T - 1), "ola: output range [%d, %d) outside the frames", trim, trim + L);
  long total = (long)B * L;
  EW_LAUNCH(ola_kernel, total, stream, Fr, env, y, T, n_fft, hop, L, trim, total);
  return se_check_launch("se_ola");
}
extern "C" int se_reflect_pad_bwd(const float* dxp, const float* c, float* dx, int B, int L, int pad, void* stream) {
  SE_REQUIRE(dxp && dx && B > 0 && L > pad && pad >= 0, "reflect_pad_bwd: bad arguments");
  long total = (long)B * L;
  EW_LAUNCH(reflect_pad_bwd_kernel, total, stream, dxp, c, dx, L, pad, total);
  return se_check_launch("se_reflect_pad_bwd");
}
extern "C" int se_compress_planes_bwd(const float* R, int ldr, const float* dP, float* dR, long rows, int F, int comp,
                                      float pre_scale, void* stream) {
  SE_REQUIRE(R && dP && dR && rows > 0 && F > 0 && ldr >= 2 * F, "compress_planes_bwd: bad arguments");
  EW_LAUNCH(compress_planes_bwd_kernel, rows * F, stream, R, ldr, dP, dR, rows, F, comp, pre_scale);
  return se_check_launch("se_compress_planes_bwd");
}
extern "C" int se_ola_bwd(const float* dy, const float* env, float* dFr, int B, int T, int n_fft, int hop, void* stream) {
  SE_REQUIRE(dy && env && dFr && B > 0 && T > 1, "ola_bwd: bad arguments");
  int L = hop * (T - 1);
  long total = (long)B * T * n_fft;
  EW_LAUNCH(ola_bwd_kernel, total, stream, dy, env, dFr, T, n_fft, hop, L, total);
  return se_check_launch("se_ola_bwd");
}
extern "C" int se_assemble(const float* mask, int ldm, const float* nin, const float* cplx, float* est, long n, void* stream) {
  SE_REQUIRE(mask && nin && cplx && est && n > 0, "assemble: bad arguments");
  EW_LAUNCH(assemble_kernel, n, stream, mask, ldm, nin, cplx, est, n);
  return se_check_launch("se_assemble");
}
extern "C" int se_assemble_bwd(const float* est, const float* dest, const float* nin, float* dmask, int ldm,
                               float* dcplx, long n, void* stream) {
  SE_REQUIRE(est && dest && nin && dmask && dcplx && n > 0, "assemble_bwd: bad arguments");
  EW_LAUNCH(assemble_bwd_kernel, n, stream, est, dest, nin, dmask, ldm, dcplx, n);
  return se_check_launch("se_assemble_bwd");
}
extern "C" int se_mask_tail(const float* U, int ldu, const float* wb, const float* slope, float* M, long n, int F, void* stream) {
  SE_REQUIRE(U && wb && slope && M && n > 0 && F > 0, "mask_tail: bad arguments");
  EW_LAUNCH(mask_tail_kernel, n, stream, U, ldu, wb, slope, M, n, F);
  return se_check_launch("se_mask_tail");
}
extern "C" int se_mask_tail_bwd(const float* U, int ldu, const float* wb, const float* slope, const float* dM, float* dU,
                                double* dwb, float* dslope, long n, int F, void* stream) {
  SE_REQUIRE(U && wb && slope && dM && dU && dwb && dslope && n > 0, "mask_tail_bwd: bad arguments");
  SE_REQUIRE(F > 0 && n % F == 0, "mask_tail_bwd: n must be a whole number of rows of F");
  const long nrows = n / F;
  const int rpb = nrows >= 4096 ? 48 : 1;                 // rows per workgroup run (one slope atomic per thread and run)
  hipLaunchKernelGGL(mask_tail_bwd_kernel, dim3((unsigned)((F + 255) / 256), (unsigned)((nrows + rpb - 1) / rpb)), dim3(256), 0,
                     as_stream(stream), U, ldu, wb, slope, dM, dU, dwb, dslope, n, F, rpb);
  return se_check_launch("se_mask_tail_bwd");
}
extern "C" int se_glu_bwd(const float* Z, const float* dU, float* dZ, long M, int H, void* stream) {
  return se_glu_bwd_amax(Z, dU, dZ, M, H, nullptr, stream);
}
extern "C" int se_glu_bwd_amax(const float* Z, const float* dU, float* dZ, long M, int H, float* amax_out, void* stream) {
  SE_REQUIRE(Z && dU && dZ && M > 0 && H > 0 && (H % 4) == 0, "glu_bwd: bad arguments");
  EW_LAUNCH(glu_bwd_kernel, M * (H / 4), stream, Z, dU, dZ, M, H, amax_out, (const float*)nullptr);
  return se_check_launch("se_glu_bwd");
}
extern "C" int se_glu_bwd_gate(const float* U, const float* G, const float* dU, float* dZ, long M, int H, float* amax_out,
                               void* stream) {
  SE_REQUIRE(U && G && dU && dZ && M > 0 && H > 0 && (H % 4) == 0, "glu_bwd_gate: bad arguments");
  EW_LAUNCH(glu_bwd_kernel, M * (H / 4), stream, U, dU, dZ, M, H, amax_out, G);
  return se_check_launch("se_glu_bwd_gate");
}
extern "C" int se_gate_tanh(const float* Y, float* G, long M, int C, void* stream) {
  SE_REQUIRE(Y && G && M > 0 && C > 0 && (C % 4) == 0, "gate_tanh: bad arguments");
  EW_LAUNCH(gate_tanh_kernel, M * (C / 4), stream, Y, G, M, C);
  return se_check_launch("se_gate_tanh");
}
extern "C" int se_gate_tanh_bwd(const float* Y, const float* dG, float* dY, long M, int C, void* stream) {
  SE_REQUIRE(Y && dG && dY && M > 0 && C > 0 && (C % 4) == 0, "gate_tanh_bwd: bad arguments");
  EW_LAUNCH(gate_tanh_bwd_kernel, M * (C / 4), stream, Y, dG, dY, M, C);
  return se_check_launch("se_gate_tanh_bwd");
}
extern "C" int se_spec_loss(const float* A, const float* Bp, double* sums, long n, void* stream) {
  SE_REQUIRE(A && Bp && sums && n > 0, "spec_loss: bad arguments");
  hipLaunchKernelGGL(spec_loss_kernel, dim3(red_grid(n, 256)), dim3(256), 0, as_stream(stream), A, Bp, sums, n);
  return se_check_launch("se_spec_loss");
}
extern "C" int se_spec_loss_bwd(const float* A, const float* Bp, float* dA, const float* up, float cmag, float cri, long n,
                                int accumulate, void* stream) {
  SE_REQUIRE(A && Bp && dA && up && n > 0, "spec_loss_bwd: bad arguments");
  EW_LAUNCH(spec_loss_bwd_kernel, n, stream, A, Bp, dA, up, cmag, cri, n, accumulate);
  return se_check_launch("se_spec_loss_bwd");
}
extern "C" int se_l1_loss(const float* a, long lda, const float* b, long ldb, double* sums, long rows, int L, void* stream) {
  SE_REQUIRE(a && b && sums && rows > 0 && L > 0, "l1_loss: bad arguments");
  long n = rows * L;
  hipLaunchKernelGGL(l1_loss_kernel, dim3(red_grid(n, 256)), dim3(256), 0, as_stream(stream), a, lda, b, ldb, sums, L, n);
  return se_check_launch("se_l1_loss");
}
extern "C" int se_l1_loss_bwd(const float* a, long lda, const float* b, long ldb, float* da, const float* up, float ck,
                              long rows, int L, void* stream) {
  SE_REQUIRE(a && b && da && up && rows > 0 && L > 0, "l1_loss_bwd: bad arguments");
  long n = rows * L;
  EW_LAUNCH(l1_loss_bwd_kernel, n, stream, a, lda, b, ldb, da, up, ck, L, n);
  return se_check_launch("se_l1_loss_bwd");
}
extern "C" int se_adamw(float* p, const float* g, float* m, float* v, long n, float lr, float b1, float b2, float eps,
                        float wd, int step, void* stream) {
  SE_REQUIRE(p && g && m && v && n > 0 && step >= 1, "adamw: bad arguments");
  float bc1 = 1.0f - powf(b1, (float)step), bc2 = 1.0f - powf(b2, (float)step);
  EW_LAUNCH(adamw_kernel, n, stream, p, g, m, v, n, lr, b1, b2, eps, wd, bc1, bc2);
  return se_check_launch("se_adamw");
}
extern "C" int se_sgd_nesterov(float* p, const float* g, float* buf, long n, float lr, float momentum, int first,
                               void* stream) {
  SE_REQUIRE(p && g && buf && n > 0, "sgd_nesterov: bad arguments");
  EW_LAUNCH(sgd_nesterov_kernel, n, stream, p, g, buf, n, lr, momentum, first);
  return se_check_launch("se_sgd_nesterov");
}
extern "C" int se_dot(const float* a, const float* b, double* out, long n, void* stream) {
  SE_REQUIRE(a && b && out && n > 0, "dot: bad arguments");
  hipLaunchKernelGGL(dot_kernel, dim3(gs_grid(n, 256)), dim3(256), 0, as_stream(stream), a, b, out, n);
  return se_check_launch("se_dot");
}
extern "C" int se_axpbypcz(const float* a, const float* b, const float* c, float* y, float alpha, float beta, float gamma,
                           long n, void* stream) {
  SE_REQUIRE(a && b && c && y && n > 0, "axpbypcz: bad arguments");
  EW_LAUNCH(axpbypcz_kernel, n, stream, a, b, c, y, alpha, beta, gamma, n);
  return se_check_launch("se_axpbypcz");
}

extern "C" int se_grad_clip(float* g, long n, const double* sums, int nsum, float max_norm, void* stream) {
  SE_REQUIRE(g && sums && n > 0 && nsum > 0 && max_norm > 0.f, "grad_clip: bad arguments");
  SE_REQUIRE(((uintptr_t)g & 15) == 0, "grad_clip: the gradient buffer must be 16-byte aligned");
  hipLaunchKernelGGL(grad_clip_kernel, dim3(cdiv(n, 1024)), dim3(256), 0, as_stream(stream), g, n, sums, nsum, max_norm);
  return se_check_launch("se_grad_clip");
}
static int seg_grid(long max_seg, int nseg, dim3* grid) {
  if (nseg <= 0 || nseg > 65535 || max_seg <= 0) return -1;
  *grid = dim3(cdiv(max_seg, SEG_CHUNK), nseg);
  return 0;
}
extern "C" int se_lars_step(float* p, const float* g, float* mu, const long* seg_off, int nseg, long max_seg,
                            double* norms, int adapt, float lr, float wd, float momentum, float trust_coef, void* stream) {
  SE_REQUIRE(p && g && mu && seg_off && norms, "lars_step: bad arguments");
  dim3 grid;
  SE_REQUIRE(seg_grid(max_seg, nseg, &grid) == 0, "lars_step: bad segment table (nseg=%d, max_seg=%ld)", nseg, max_seg);
  if (adapt) {
    if (hipMemsetAsync(norms, 0, sizeof(double) * 2 * nseg, as_stream(stream)) != hipSuccess) return se_fail("segment-norm workspace memset failed");
    hipLaunchKernelGGL(seg_norms_kernel, grid, dim3(256), 0, as_stream(stream), (const float*)p, g, (float*)nullptr,
                       (float*)nullptr, seg_off, norms, 0, wd, 0.f, 0.f, 0.f, 0.f, 1.f, 1.f, (const double*)nullptr, 0, 0.f);
  }
  hipLaunchKernelGGL(seg_apply_kernel, grid, dim3(256), 0, as_stream(stream), p, g, mu, (const float*)nullptr, seg_off,
                     (const double*)norms, 0, adapt, lr, wd, momentum, trust_coef, 0.f, 1.f, 1.f, 0);
  return se_check_launch("se_lars_step");
}
extern "C" int se_lamb_step(float* p, const float* g, float* m, float* v, const long* seg_off, int nseg, long max_seg,
                            double* norms, const double* gsum, int nsum, float max_grad_norm, int adapt, int trust_clip,
                            float lr, float wd, float b1, float b2, float beta3, float eps, float bc1, float bc2,
                            void* stream) {
  SE_REQUIRE(p && g && m && v && seg_off && norms && bc1 > 0.f && bc2 > 0.f, "lamb_step: bad arguments");
  SE_REQUIRE(nsum == 0 || gsum, "lamb_step: nsum > 0 needs the gradient sums");
  dim3 grid;
  SE_REQUIRE(seg_grid(max_seg, nseg, &grid) == 0, "lamb_step: bad segment table (nseg=%d, max_seg=%ld)", nseg, max_seg);
  float rbc2 = 1.0f / sqrtf(bc2);
  if (hipMemsetAsync(norms, 0, sizeof(double) * 2 * nseg, as_stream(stream)) != hipSuccess) return se_fail("segment-norm workspace memset failed");
  hipLaunchKernelGGL(seg_norms_kernel, grid, dim3(256), 0, as_stream(stream), (const float*)p, g, m, v, seg_off, norms, 1, wd,
                     b1, b2, beta3, eps, bc1, rbc2, gsum, nsum, max_grad_norm);
  hipLaunchKernelGGL(seg_apply_kernel, grid, dim3(256), 0, as_stream(stream), p, g, m, (const float*)v, seg_off,
                     (const double*)norms, 1, adapt, lr, wd, 0.f, 0.f, eps, bc1, rbc2, trust_clip);
  return se_check_launch("se_lamb_step");
}

// =============================================================================================
// Metric-discriminator tail and spectral normalisation as kernels (models/discriminator.py:39-57): the D step then has no
// vendor-BLAS launch.  All operands are weight- or [B, 128]-sized: one workgroup each, latency-sized work.
//
// spectral_norm (torch.nn.utils.spectral_norm, one power iteration in training mode, eps 1e-12):
//   v = normalize(W^T u), u = normalize(W v)   (train only; u, v updated in place)
//   sigma = u . (W v);  Wn = W / sigma
#define SN_MAX 8
struct SnBatch { const float* W[SN_MAX]; float* u[SN_MAX]; float* v[SN_MAX]; float* Wn[SN_MAX]; int h[SN_MAX], w[SN_MAX]; float* sigma; };
struct SnBwdBatch { const float* dWn[SN_MAX]; const float* Wn[SN_MAX]; const float* u[SN_MAX]; const float* v[SN_MAX];
                    float* dW[SN_MAX]; int h[SN_MAX], w[SN_MAX]; const float* sigma; };
// one workgroup (1024 threads) per weight matrix of the batch: the six spectral-norm layers of the discriminator in ONE launch
__global__ __launch_bounds__(1024) void spectral_norm_kernel(SnBatch a, int train, float eps) {
  extern __shared__ float sn[];           // v [w] | s = W v [h] | red [16]
  const int li = blockIdx.x, h = a.h[li], w = a.w[li];
  const float* __restrict__ W = a.W[li];
  float* u = a.u[li]; float* v = a.v[li]; float* Wn = a.Wn[li];
  float* vs = sn; float* ss = sn + w; float* red = ss + h;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, NT = 1024, NWV = 16;
  auto block_sum = [&](float x) {
    x = wave_sum(x);
    __syncthreads();
    if (lane == 0) red[wv] = x;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) t += red[i];
    return t;
  };
  if (train) {
    float nn = 0.f;
    for (int j = tid; j < w; j += NT) {
      float t = 0.f;
      for (int i = 0; i < h; ++i) t += W[(long)i * w + j] * u[i];
      vs[j] = t; nn += t * t;
    }
    const float nv = fmaxf(sqrtf(block_sum(nn)), eps);
    for (int j = tid; j < w; j += NT) vs[j] /= nv;
  } else {
    for (int j = tid; j < w; j += NT) vs[j] = v[j];
  }
  __syncthreads();
  float n2 = 0.f;
  for (int i = wv; i < h; i += NWV) {               // one wave per row
    float t = 0.f;
    for (int j = lane; j < w; j += 64) t += W[(long)i * w + j] * vs[j];
    t = wave_sum(t);
    if (lane == 0) { ss[i] = t; n2 += t * t; }
  }
  const float nu2 = block_sum(n2);
  float sigma;
  if (train) {
    const float nu = fmaxf(sqrtf(nu2), eps);
    sigma = nu2 / nu;                                // u . (W v) with u = s / nu
    for (int i = tid; i < h; i += NT) u[i] = ss[i] / nu;
    for (int j = tid; j < w; j += NT) v[j] = vs[j];
  } else {
    float d = 0.f;
    for (int i = tid; i < h; i += NT) d += u[i] * ss[i];
    sigma = block_sum(d);
  }
  const float inv = 1.0f / sigma;
  for (long i = tid; i < (long)h * w; i += NT) Wn[i] = W[i] * inv;
  if (tid == 0) a.sigma[li] = sigma;
}
// dW = (dWn - <dWn, Wn> u v^T) / sigma   (u, v are constants of the graph, as in torch's hook)
__global__ __launch_bounds__(1024) void spectral_norm_bwd_kernel(SnBwdBatch a) {
  __shared__ float red[16];
  const int li = blockIdx.x, h = a.h[li], w = a.w[li], tid = threadIdx.x;
  const float* __restrict__ dWn = a.dWn[li]; const float* __restrict__ Wn = a.Wn[li];
  if (!dWn) return;                                  // this layer needs no gradient
  float d = 0.f;
  for (long i = tid; i < (long)h * w; i += 1024) d += dWn[i] * Wn[i];
  d = wave_sum(d);
  if ((tid & 63) == 0) red[tid >> 6] = d;
  __syncthreads();
  float dot = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) dot += red[i];
  const float inv = 1.0f / a.sigma[li];
  const float* u = a.u[li]; const float* v = a.v[li]; float* dW = a.dW[li];
  for (long i = tid; i < (long)h * w; i += 1024) {
    const int r = (int)(i / w), cc = (int)(i - (long)r * w);
    dW[i] += (dWn[i] - dot * u[r] * v[cc]) * inv;
  }
}

// tail: global max-pool over the P positions of a [B, P, 128] map -> Linear(128, 64) -> dropout mask -> PReLU(64) ->
// Linear(64, 1) -> beta * sigmoid(slope * z).  One workgroup per batch element.  ws: float [B][SE_DISC_TAIL_WS] saved for the
// backward: pooled [128] | argmax [128] (as float bits) | h1 (pre-mask, pre-PReLU) [64] | z [1]
#define DT_C 128
#define DT_H 64
#define SE_DISC_TAIL_WS_ (DT_C + DT_C + DT_H + 4)
__global__ __launch_bounds__(256) void disc_tail_fwd_kernel(const float* __restrict__ A, int P, const float* __restrict__ W1,
                                                            const float* __restrict__ b1, const float* __restrict__ mask,
                                                            const float* __restrict__ slope1, const float* __restrict__ W2,
                                                            const float* __restrict__ b2, const float* __restrict__ sslope, float beta,
                                                            float* __restrict__ out, float* __restrict__ ws) {
  __shared__ float pool[DT_C], h2[DT_H];
  const int b = blockIdx.x, tid = threadIdx.x;
  float* w_ = ws + (long)b * SE_DISC_TAIL_WS_;
  if (tid < DT_C) {
    const float* a = A + (long)b * P * DT_C + tid;
    float m = a[0]; int am = 0;
    for (int p = 1; p < P; ++p) { float x = a[(long)p * DT_C]; if (x > m) { m = x; am = p; } }
    pool[tid] = m; w_[tid] = m; w_[DT_C + tid] = __int_as_float(am);
  }
  __syncthreads();
  if (tid < DT_H) {
    float t = b1[tid];
    for (int cc = 0; cc < DT_C; ++cc) t += W1[tid * DT_C + cc] * pool[cc];
    w_[2 * DT_C + tid] = t;
    if (mask) t *= mask[(long)b * DT_H + tid];
    h2[tid] = t >= 0.f ? t : t * slope1[tid];
  }
  __syncthreads();
  if (tid < 64) {
    float t = W2[tid] * h2[tid];
    t = wave_sum(t);
    if (tid == 0) {
      const float z = t + b2[0];
      w_[2 * DT_C + DT_H] = z;
      out[b] = beta / (1.0f + expf(-sslope[0] * z));
    }
  }
}
__global__ __launch_bounds__(256) void disc_tail_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ ws, int P,
                                                            const float* __restrict__ W1, const float* __restrict__ mask,
                                                            const float* __restrict__ slope1, const float* __restrict__ W2,
                                                            const float* __restrict__ sslope, float beta, float* __restrict__ dA,
                                                            float* __restrict__ dW1, float* __restrict__ db1, float* __restrict__ dslope1,
                                                            float* __restrict__ dW2, float* __restrict__ db2, float* __restrict__ dsslope) {
  __shared__ float dh1[DT_H];
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* w_ = ws + (long)b * SE_DISC_TAIL_WS_;
  const float z = w_[2 * DT_C + DT_H], sl = sslope[0];
  const float sg = 1.0f / (1.0f + expf(-sl * z));
  const float go = dout[b] * beta * sg * (1.0f - sg);          // d out / d (slope * z)
  const float dz = go * sl;
  if (tid == 0) { if (dsslope) atomicAdd(dsslope, go * z); if (db2) atomicAdd(db2, dz); }
  if (tid < DT_H) {
    float t = w_[2 * DT_C + tid];
    const float mk = mask ? mask[(long)b * DT_H + tid] : 1.0f;
    const float tm = t * mk;
    const float h2v = tm >= 0.f ? tm : tm * slope1[tid];
    if (dW2) atomicAdd(&dW2[tid], dz * h2v);
    const float dh2 = dz * W2[tid];
    if (tm < 0.f && dslope1) atomicAdd(&dslope1[tid], dh2 * tm);
    const float d1 = dh2 * (tm >= 0.f ? 1.0f : slope1[tid]) * mk;
    dh1[tid] = d1;
    if (db1) atomicAdd(&db1[tid], d1);
  }
  __syncthreads();
  if (dW1) for (int i = tid; i < DT_H * DT_C; i += 256) atomicAdd(&dW1[i], dh1[i / DT_C] * w_[i % DT_C]);
  if (dA && tid < DT_C) {
    float t = 0.f;
    for (int j = 0; j < DT_H; ++j) t += dh1[j] * W1[j * DT_C + tid];
    const int am = __float_as_int(w_[DT_C + tid]);
    dA[((long)b * P + am) * DT_C + tid] = t;               // dA is zero-initialised by the caller
  }
}

extern "C" int se_spectral_norm(int n, const float* const* W, float* const* u, float* const* v, float* const* Wn,
                                const int* h, const int* w, float* sigma, int train, float eps, void* stream) {
  SE_REQUIRE(n > 0 && n <= SN_MAX && W && u && v && Wn && h && w && sigma, "spectral_norm: bad arguments (at most %d matrices)", SN_MAX);
  SnBatch a;
  int maxhw = 0;
  for (int i = 0; i < n; ++i) {
    SE_REQUIRE(W[i] && u[i] && v[i] && Wn[i] && h[i] > 0 && w[i] > 0 && h[i] + w[i] <= 8192, "spectral_norm: bad matrix %d", i);
    a.W[i] = W[i]; a.u[i] = u[i]; a.v[i] = v[i]; a.Wn[i] = Wn[i]; a.h[i] = h[i]; a.w[i] = w[i];
    if (h[i] + w[i] > maxhw) maxhw = h[i] + w[i];
  }
  a.sigma = sigma;
  hipLaunchKernelGGL(spectral_norm_kernel, dim3(n), dim3(1024), (size_t)(maxhw + 16) * sizeof(float), as_stream(stream), a, train, eps);
  return se_check_launch("se_spectral_norm");
}
extern "C" int se_spectral_norm_bwd(int n, const float* const* dWn, const float* const* Wn, const float* const* u,
                                    const float* const* v, const float* sigma, float* const* dW, const int* h, const int* w,
                                    void* stream) {
  SE_REQUIRE(n > 0 && n <= SN_MAX && dWn && Wn && u && v && sigma && dW && h && w, "spectral_norm_bwd: bad arguments");
  SnBwdBatch a;
  for (int i = 0; i < n; ++i) {
    SE_REQUIRE(!dWn[i] || (Wn[i] && u[i] && v[i] && dW[i] && h[i] > 0 && w[i] > 0), "spectral_norm_bwd: bad matrix %d", i);
    a.dWn[i] = dWn[i]; a.Wn[i] = Wn[i]; a.u[i] = u[i]; a.v[i] = v[i]; a.dW[i] = dW[i]; a.h[i] = h[i]; a.w[i] = w[i];
  }
  a.sigma = sigma;
  hipLaunchKernelGGL(spectral_norm_bwd_kernel, dim3(n), dim3(1024), 0, as_stream(stream), a);
  return se_check_launch("se_spectral_norm_bwd");
}
extern "C" size_t se_disc_tail_workspace_bytes(int B) { return B > 0 ? (size_t)B * SE_DISC_TAIL_WS_ * sizeof(float) : 0; }
extern "C" int se_disc_tail_fwd(const float* A, int B, int P, const float* W1, const float* b1, const float* mask,
                                const float* slope1, const float* W2, const float* b2, const float* sslope, float beta, float* out,
                                float* ws, void* stream) {
  SE_REQUIRE(A && W1 && b1 && slope1 && W2 && b2 && sslope && out && ws && B > 0 && P > 0, "disc_tail_fwd: bad arguments");
  hipLaunchKernelGGL(disc_tail_fwd_kernel, dim3(B), dim3(256), 0, as_stream(stream), A, P, W1, b1, mask, slope1, W2, b2, sslope, beta,
                     out, ws);
  return se_check_launch("se_disc_tail_fwd");
}
extern "C" int se_disc_tail_bwd(const float* dout, const float* ws, int B, int P, const float* W1, const float* mask,
                                const float* slope1, const float* W2, const float* sslope, float beta, float* dA, float* dW1,
                                float* db1, float* dslope1, float* dW2, float* db2, float* dsslope, void* stream) {
  SE_REQUIRE(dout && ws && W1 && slope1 && W2 && sslope && B > 0 && P > 0, "disc_tail_bwd: bad arguments");
  hipLaunchKernelGGL(disc_tail_bwd_kernel, dim3(B), dim3(256), 0, as_stream(stream), dout, ws, P, W1, mask, slope1, W2, sslope, beta,
                     dA, dW1, db1, dslope1, dW2, db2, dsslope);
  return se_check_launch("se_disc_tail_bwd");
}


// ---- operand scales of the scaled split-fp16 kernels that are NOT gradients (round 4) -------------------------------------------
// dst[r][0 .. C) = src[r][0 .. C) (row strides lds / ldd, C % 4 == 0), raising *amax_out to max |src|: the slab copy at a decoder's
// entry (models/generator.py:84,113: the dense block's input, the TSCB output, has no bound by construction) hands the skip
// stack its operand scale
__global__ __launch_bounds__(256) void copy_cols_amax_kernel(const float* __restrict__ src, int lds, float* __restrict__ dst, int ldd,
                                                             long rows, int C4, float* __restrict__ amax_out) {
  float m = 0.f;
  const long total = rows * C4;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long r = i / C4;
    const int q = (int)(i - r * C4);
    const float4 v = *reinterpret_cast<const float4*>(src + r * lds + 4 * q);
    *reinterpret_cast<float4*>(dst + r * ldd + 4 * q) = v;
    m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
  }
  if (amax_out) {
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) amax_raise_(amax_out, m);
  }
}
extern "C" int se_copy_cols_amax(const float* src, int lds, float* dst, int ldd, long rows, int C, float* amax_out, void* stream) {
  SE_REQUIRE(src && dst && rows > 0 && C > 0 && (C % 4) == 0 && (lds % 4) == 0 && (ldd % 4) == 0 && lds >= C && ldd >= C,
             "copy_cols_amax: bad arguments (C=%d lds=%d ldd=%d)", C, lds, ldd);
  const long total = rows * (C / 4);
  long nb = (total + 256 * 8 - 1) / (256 * 8);
  if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(copy_cols_amax_kernel, dim3((unsigned)nb), dim3(256), 0, as_stream(stream), src, lds, dst, ldd, rows, C / 4, amax_out);
  return se_check_launch("se_copy_cols_amax");
}

// PROVEN bounds of normalised activations from the CURRENT parameters, one workgroup per item, every step (weights.WeightPlan):
//   b = (k max |gamma| + max |beta|) * max(1, max |slope|)        a normalised row / plane: |x_hat| <= sqrt(count - 1) = k
//   W != NULL:  b = b * max_j sum_i |W[j][i]| + max |wb|           the linear layer behind it (feed-forward: H = W1 LN(x) + b1)
//   b *= post                                                     (Swish(h) <= |h|; dropout keep factor)
// stored to *out (one item per scalar).  k = kconst, or the kernel argument k1 / k2 (counts
// known only at run time: BatchNorm over the tokens of the batch).
__global__ __launch_bounds__(256) void act_bounds_kernel(const se_bound_item* __restrict__ items, float k1, float k2) {
  const se_bound_item it = items[blockIdx.x];
  __shared__ float red[3][4];
  const int tid = threadIdx.x;
  float mg = 0.f, mb = 0.f, ma = 1.f, ml = 0.f, mw = 0.f;
  for (int i = tid; i < it.n; i += 256) { mg = fmaxf(mg, fabsf(it.g[i])); if (it.b) mb = fmaxf(mb, fabsf(it.b[i])); }
  if (it.alpha) for (int i = tid; i < it.na; i += 256) ma = fmaxf(ma, fabsf(it.alpha[i]));
  if (it.W) {
    for (int j = tid; j < it.rows; j += 256) {
      float l1 = 0.f;
      for (int i = 0; i < it.cols; ++i) l1 += fabsf(it.W[(long)j * it.cols + i]);
      ml = fmaxf(ml, l1);
      if (it.wb) mw = fmaxf(mw, fabsf(it.wb[j]));
    }
  }
  float v[5] = {mg, mb, ma, ml, mw};
#pragma unroll
  for (int q = 0; q < 5; ++q) {
    v[q] = wave_max(v[q]);
    __syncthreads();
    if ((tid & 63) == 0) red[0][tid >> 6] = v[q];
    __syncthreads();
    v[q] = fmaxf(fmaxf(red[0][0], red[0][1]), fmaxf(red[0][2], red[0][3]));
  }
  if (tid == 0) {
    const float k = it.ksel == 1 ? k1 : (it.ksel == 2 ? k2 : it.kconst);
    float bnd = (k * v[0] + v[1]) * v[2];
    if (it.W) bnd = bnd * v[3] + v[4];
    bnd *= it.post;
    // (one rounding up: the scale only has to be >= the true maximum).  A plain store: one item per scalar, and no window in
    // which a kernel of another stream could read a zeroed scalar
    *it.out = bnd * 1.0000002f;
  }
}
extern "C" int se_act_bounds(const se_bound_item* items_dev, int nitems, float k1, float k2, void* stream) {
  SE_REQUIRE(items_dev && nitems > 0 && nitems <= 65535, "act_bounds: bad arguments");
  hipLaunchKernelGGL(act_bounds_kernel, dim3((unsigned)nitems), dim3(256), 0, as_stream(stream), items_dev, k1, k2);
  return se_check_launch("se_act_bounds");
}
