/* C-ABI of libse_hip.so -- the MI355X (gfx950) kernels behind the CMGAN / SCP-GAN hot path.
 *
 * The reference (minyoungpark1/Speech-Enhancement) is 100 % Python and has no FFI of its own: every
 * entry point below replaces an *implicit vendor kernel* that the reference reaches through ATen
 * (SURVEY.md section 2a); the reference call site each one stands in for is cited per function.
 *
 * Conventions (all functions):
 *   - plain pointers + sizes, no torch types; every pointer is a DEVICE pointer owned by the caller;
 *   - stream-ordered on `stream` (a hipStream_t passed as void*), no allocation, no host sync;
 *   - return 0 on success, negative on error (se_last_error() gives the text); never throws;
 *   - feature maps are channels-last fp32: X[b][t][f][c], pixel stride `ld` floats.
 */
#ifndef SE_HIP_H
#define SE_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SE_MAX_TAPS 16

/* prologue applied to the A operand while it is staged into LDS */
enum { SE_PRO_NONE = 0, SE_PRO_LN = 1, SE_PRO_SWISH = 2, SE_PRO_AFFINE_SWISH = 3 };
/* epilogue flags (bit mask) */
enum {
  SE_EPI_BIAS = 1,       /* + bias[n]                                                   */
  SE_EPI_ACCUM = 2,      /* Y += result (dgrad into a shared gradient buffer)           */
  SE_EPI_RESID = 4,      /* Y = R + alpha * result                                      */
  SE_EPI_GLU = 8,        /* N = 2*No: Y[.., j] = a * sigmoid(g); also stores pre-GLU Z   */
  SE_EPI_STATS = 16,     /* per-(b, n) sum / sum-of-squares of the result (fp64 atomics) */
  SE_EPI_SWISH_GRAD = 32,/* Y = result * swish'(AUX[m][n])                              */
  SE_EPI_SHUFFLE2 = 64   /* sub-pixel: channel r*No+c -> pixel 2f+r, channel c (N = 2*No) */
};

/* One "tap GEMM":  Y[m][n] = epi( sum_tap sum_c pro(A[src(m,tap)][a_off+c]) * W[n][tap*C+c] )
 * m = (b, t, f) over the OUTPUT grid; src(m,tap) = (b, t*st+dt[tap], f*sf+df[tap]) (down mode) or
 * (b, (t+dt)/st, (f+df)/sf) when divisible (up mode = gradient of a strided conv); rows that fall
 * outside the input grid contribute zero.  Replaces nn.Conv2d / nn.Conv1d(k=1) / nn.Linear forward
 * and input-gradient (models/generator.py:19-20,39,45,82,100,103,122; models/conformer.py:87-89,
 * 136-142,164,169; models/discriminator.py:39-56) and the DFT of torch.stft/istft
 * (core/function.py:690-703). */
typedef struct {
  int B, To, Fo, Ti, Fi;
  int st, sf, up;
  int ntap;
  int dt[SE_MAX_TAPS], df[SE_MAX_TAPS];
  int C, lda, a_off;
  int N, ldc, c_off;
  int ldw;               /* floats between consecutive n rows of W (>= ntap*C)               */
  int prologue, epilogue;
  float alpha;           /* SE_EPI_RESID scale                                              */
  int ldr, r_off;        /* residual pixel stride / channel offset                          */
  int ldx, x_off;        /* AUX (SWISH_GRAD) or Z (GLU) pixel stride / channel offset        */
} se_gemm_desc;

int se_version(void);
const char* se_last_error(void);

/* forward / input-gradient tap GEMM.  rowstats: [M][2] (mean, rstd) for SE_PRO_LN;
 * pro_scale/pro_shift: per-channel (LN gamma/beta or BN scale/shift); stats: double [B][N][2]. */
int se_gemm_tap(const se_gemm_desc* d, const float* A, const float* W, const float* bias,
                float* Y, const float* R, float* AUX, const float* rowstats,
                const float* pro_scale, const float* pro_shift, double* stats, void* stream);

/* weight gradient: dW[n][tap*C + c] += sum_m dY[m][n] * pro(A[src(m,tap)][c]);  dW must be zeroed by
 * the caller (fp32 atomics across row chunks).  If dbias != NULL also dbias[n] += sum_m dY[m][n].
 * Replaces the weight-gradient kernels of the same ATen ops as se_gemm_tap. */
int se_gemm_tap_wgrad(const se_gemm_desc* d, const float* A, const float* dY, float* dW, float* dbias,
                      const float* rowstats, const float* pro_scale, const float* pro_shift,
                      int chunks, void* stream);

/* generic strided repack: dst[o][t][i] = src[o*so + i*si + t*stt] with optional reversal of the
 * 64-channel slabs of the o or i index (DilatedDenseNet concatenates newest-first,
 * models/generator.py:31).  rev: 0 none, 1 reverse slabs of i, 2 reverse slabs of o. */
int se_repack(const float* src, float* dst, int No, int Nt, int Ni, long so, long stt, long si,
              int rev, int accumulate, void* stream);

/* ---- norms / elementwise (se_norms.hip, se_elem.hip) -------------------------------------- */
/* per-row mean / rstd over C channels (LayerNorm statistics, models/conformer.py:67,162,204) */
int se_row_stats(const float* X, float* stats, long M, int C, int ld, float eps, void* stream);
/* Y = LN(X)*g + b (+ R): the ConformerBlock post_norm + TSCB residual (conformer.py:211, generator.py:70,72) */
int se_layernorm_fwd(const float* X, const float* g, const float* b, const float* R, float* Y,
                     float* stats, long M, int C, float eps, void* stream);
/* dX (+)= LN backward; dg, db accumulated with atomics (must be zeroed) */
int se_layernorm_bwd(const float* X, const float* stats, const float* g, const float* dY, float* dX,
                     float* dg, float* db, long M, int C, int accumulate, int affine_in_dy, void* stream);
/* InstanceNorm2d(affine)+PReLU apply from fp64 (sum, sumsq) stats: Y[..c_off+c] (generator.py:21-22 etc.) */
int se_inorm_prelu_fwd(const float* X, int ldx, const double* stats, const float* g, const float* b,
                       const float* slope, float* Y, int ldy, int y_off, int B, long P, int C,
                       float eps, void* stream);
int se_inorm_prelu_bwd_reduce(const float* X, int ldx, const double* stats, const float* g, const float* b,
                              const float* slope, const float* dY, int ldy, int y_off, double* red,
                              int B, long P, int C, float eps, void* stream);
int se_inorm_prelu_bwd_apply(const float* X, int ldx, const double* stats, const float* g, const float* b,
                             const float* slope, const float* dY, int ldy, int y_off, const double* red,
                             float* dX, float* dg, float* db, float* dslope, int B, long P, int C,
                             float eps, void* stream);
/* column statistics (sum, sumsq) of X[M][C] per batch into fp64 stats[B][C][2] (atomics, zeroed by caller) */
int se_col_stats(const float* X, int ld, int x_off, double* stats, int B, long P, int C, void* stream);

#ifdef __cplusplus
}
#endif
#endif
