/* C-ABI of libse_hip.so -- the MI355X (gfx950) kernels behind the CMGAN / SCP-GAN hot path.
 *
 * The reference (minyoungpark1/Speech-Enhancement) is 100 % Python and has no FFI of its own: every
 * entry point below replaces an *implicit vendor kernel* that the reference reaches through ATen
 * (SURVEY.md section 2a); the reference call site each one stands in for is cited per function.
 *
 * Conventions (all functions):
 *   - plain pointers + sizes, no torch types; every pointer is a DEVICE pointer owned by the caller;
 *   - stream-ordered on `stream` (a hipStream_t passed as void*), no allocation, no host sync;
 *   - return 0 on success, negative on error (se_last_error() gives the text of the calling THREAD's last error:
 *     the buffer is thread-local, the entry points keep no other state and are re-entrant); never throws;
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
enum { SE_PRO_NONE = 0, SE_PRO_LN = 1, SE_PRO_SWISH = 2, SE_PRO_AFFINE_SWISH = 3,
       SE_PRO_SWISH_DROP = 4,   /* swish then dropout (nn.Dropout after Swish, models/conformer.py:139)  */
       SE_PRO_DROP = 5,         /* dropout only (backward of an output dropout: mask * dY as the A operand) */
       SE_PRO_GATE = 6 };       /* CDiffuSE residual block (models/DiffuSE.py:117-122) as the prologue of its 1 x 1 output projection: the A
                                   operand is y2[c] = sigmoid(zg[c]) tanh(zf[c]), z = GroupNorm(R) + conditioner, built from rows of
                                   A = R [.., 128] (gate half | filter half, lda = 128), AUX = conditioner rows (same layout) and
                                   pro_scale = the GroupNorm (scale, shift) pairs [B][128][2]; C = 64, 1-D maps, precision 3 with a static
                                   a_sexp (|y2| < 1), N = 128: y2 never goes to memory */
/* epilogue flags (bit mask) */
enum {
  SE_EPI_BIAS = 1,       /* + bias[n]                                                   */
  SE_EPI_ACCUM = 2,      /* Y += result (dgrad into a shared gradient buffer)           */
  SE_EPI_RESID = 4,      /* Y = R + alpha * result                                      */
  SE_EPI_GLU = 8,        /* N = 2*No: Y[.., j] = a * sigmoid(g); also stores pre-GLU Z   */
  SE_EPI_STATS = 16,     /* per-(b, n) sum / sum-of-squares of the result (fp64 atomics) */
  SE_EPI_SWISH_GRAD = 32,/* Y = result * swish'(AUX[m][n])                              */
  SE_EPI_SHUFFLE2 = 64,  /* sub-pixel: channel r*No+c -> pixel 2f+r, channel c (N = 2*No) */
  SE_EPI_DROP = 128,     /* dropout on the (bias-added) result, mask = hash(epi_seed, m*N + n) (nn.Dropout after a
                            Linear, conformer.py:94,141); in se_gemm_tap_wgrad: the same mask applied to dY      */
  SE_EPI_GLU_GATE = 2048,/* with SE_EPI_GLU: AUX receives only the gate half g [M][No] (ldx >= No) instead of the pre-GLU [a | g]:
                            the GLU backward needs (u = a sigmoid(g), g) only -- d a = dU sigmoid(g), d g = dU u (1 - sigmoid(g)) */
  SE_EPI_ROWSTATS = 512  /* N == 64, row GEMM: AUX [M][2] receives (mean, rstd) over the 64 channels of every RESULT row,
                            eps = 1e-5 -- the statistics of the nn.LayerNorm(64) that reads this output next
                            (conformer.py:67,162), i.e. se_row_stats without its pass over the rows                  */
  , SE_EPI_DELTA = 4096   /* N == 64, row GEMM: AUX [M][4] receives delta[m][h] = sum over the 16 columns of head h of
                            result * R -- the softmax-backward row constant rowsum(dO . O) of the attention backward
                            (conformer.py:119-121 backwards) when this GEMM is the to_out input gradient (result = dO) and
                            R = the attention output O [M][64]; R is only read (no residual is added); excludes
                            SE_EPI_RESID / SE_EPI_SWISH_GRAD / SE_EPI_ROWSTATS                                        */
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
  unsigned pro_seed, epi_seed;   /* dropout streams of the prologue / epilogue masks             */
  float drop_p;          /* dropout probability (0 = off), realised as round(p * 65536) / 65536; kept elements are scaled by the exact inverse keep probability */
  int precision;         /* 0: fp32 MFMA; 1: split-bf16 hi/lo (3 bf16 MFMAs per product, ~1.5e-5 relative);
                            2: split-bf16 hi/mid/lo (6 bf16 MFMAs, exact 24-bit split: fp32-equivalent, ~1e-7);
                            3: SCALED split-fp16 hi/lo (3 fp16 MFMAs per product): each operand is multiplied by a power of
                               two that brings its largest magnitude to [2^13, 2^14) (exact), then x = hi + lo with two 11-bit
                               fp16 significands: 2^-24 relative representation error (the fp32 rounding unit) for every
                               element within 2^17 of the operand's maximum and 2^-39 of that maximum below; products
                               hi*hi + hi*lo + lo*hi are exact in the fp32 accumulator, the dropped lo*lo term is <= 2^-24
                               relative: fp32-equivalent like 2, at half the MFMAs and two thirds of the LDS bytes          */
  int w_planes;          /* 0: W is fp32 [N][ldw].  > 0 (se_gemm_tap, precision 2, C >= 32 only): W points to the weights
                            PRE-SPLIT by se_weight_prep -- three bf16 planes (hi, mid, lo) of [N][ldw] elements each,
                            w_planes elements apart (ldw, w_planes multiples of 8, 16-byte aligned); same results as the
                            fp32 operand (the split is exact), without re-splitting the weights in every workgroup.
                            precision 3: TWO fp16 planes (hi, lo) of the weights scaled by 2^sexp(*w_amax) (se_weight_prep,
                            fmt 1) */
  int a_sexp, w_sexp;    /* precision 3: log2 of the power-of-two operand scales used when the amax pointer below is NULL
                            (e.g. a_sexp = 4 for InstanceNorm / LayerNorm outputs: |x| < 4096 representable)               */
  const float* a_amax;   /* precision 3: device scalar >= max |A| (e.g. se_absmax, or a producer kernel's atomic maximum); the
                            kernel derives the scale 2^(13 - floor(log2 amax)) from it.  NULL: a_sexp                     */
  const float* w_amax;   /* the same for W; with pre-split fp16 planes it must be the scalar se_weight_prep scaled them by.
                            se_gemm_tap_wgrad: a_* scale the activations A, w_* the second operand dY                     */
  float* y_amax;         /* se_gemm_tap, any precision: when not NULL, a (zero-initialised or running) device scalar that the
                            vector epilogue raises to max |Y| over everything this launch stores -- the operand scale of a
                            scaled split-fp16 consumer (qkv -> se_attn_fwd_f16, dO -> se_attn_bwd_f16_phase).  Ignored by the
                            GLU / shuffle / scalar epilogues (host error if set with those)                              */
} se_gemm_desc;

int se_version(void);
const char* se_last_error(void);

/* Workspace sizes (bytes) of the entry points that take a caller-owned workspace; pure host functions, no GPU call.
 *   se_attn_bwd               : ws     = row constants [ntok][4] + fp16-split copies of E + per-item dE tiles
 *   se_norm_prelu_bwd         : red    = double [per_batch ? B : 1][C][3]
 *   se_dwconv31_wgrad         : ws     = float  [512 workgroups][32][128]
 *   se_disc_tail_fwd / _bwd   : ws     = float  [B][324]
 *   se_lars_step / lamb_step  : norms  = double [nseg][2]
 */
size_t se_attn_bwd_workspace_bytes(long ntok, int maxpos, int nseq, int n);
size_t se_norm_prelu_bwd_workspace_bytes(int B, int C, int per_batch);
size_t se_segnorm_workspace_bytes(int nseg);


/* forward / input-gradient tap GEMM.  rowstats: [M][2] (mean, rstd) for SE_PRO_LN;
 * pro_scale/pro_shift: per-channel (LN gamma/beta or BN scale/shift); stats: double [B][N][2]. */
int se_gemm_tap(const se_gemm_desc* d, const float* A, const float* W, const float* bias,
                float* Y, const float* R, float* AUX, const float* rowstats,
                const float* pro_scale, const float* pro_shift, double* stats, void* stream);

/* operand scales of the token-wise scaled split-fp16 kernels (precision 3; see se_gemm_desc.precision).  amax pointers are device
 * scalars: an input amax must be >= max |operand| (raised by the kernel that produced the operand, or by se_weight_prep for
 * weight planes); output amax scalars must be zero (or a running maximum) on entry and are raised with one atomic per wave */
typedef struct {
  const float* in_amax;  /* first A operand: se_ff_bwd_dgrad_f16: max |dY|; se_ff_fwd_f16: a bound of |LN(X)| (se_act_bounds); NULL: in_sexp */
  int in_sexp;           /* static exponent of a bounded first A operand (LayerNorm output: 6 -> |x| < 1023)                    */
  int mid_sexp;          /* static exponent of the in-kernel second A operand of se_ff_fwd_f16 (Swish(H) * mask: 3 -> < 8191)  */
  const float* wa_amax;  /* scalar of the first weight matrix's fp16 planes (W1 / W2T)                                       */
  const float* wb_amax;  /* the same for the second one (W2 / W1T)                                                            */
  float* out_amax;       /* se_ff_bwd_dgrad_f16: raised to max |dX|; may be NULL                                              */
  float* mid_amax;       /* se_ff_bwd_dgrad_f16: raised to max |dZ|; se_ff_fwd_f16: READ as a bound of |Swish(H) * mask| (NULL: mid_sexp) */
} se_f16_scales;

/* Fused feed-forward forward of a Conformer block (models/conformer.py:53-71,128-145: Scale(0.5, PreNorm(FeedForward))), scaled
   split-fp16 arithmetic (se_gemm_desc.precision 3), hid == 256:
   Y = X + alpha * Drop_o(W2 Drop_h(Swish(W1 LN(X) + b1)) + b2).  X, Y: [M, 64]; rowstats: [M, 2] (mean, rstd) from se_row_stats;
   W1 [hid, 64], W2 [64, hid]: two scaled fp16 planes each (se_weight_prep fmt 1), 16-byte aligned; precision must be 3 | 16;
   sc = their amax scalars + the bounds / static exponents of the activations; dropout masks as in se_gemm_tap (prologue seed =
   seed_h on H, epilogue = seed_o); out_stats (may be NULL): (mean, rstd) (eps 1e-5) of every row of Y, the statistics of the
   LayerNorm(64) that reads Y next.  H == NULL (the default path): the W-stationary kernel, H = W1 LN(X) + b1 is not stored (se_ff_bwd_fused
   recomputes it); H != NULL: [M, hid] is written for se_ff_bwd_dgrad_f16 (the cross-check path). */
int se_ff_fwd_f16(const float* X, const float* rowstats, const float* gamma, const float* beta, const float* W1,
                  const float* b1, const float* W2, const float* b2, float* H, float* Y, float* out_stats, long M, int hid,
                  float drop_p, unsigned seed_h, unsigned seed_o, float alpha, int precision, const se_f16_scales* sc, void* stream);

/* Input-gradient chain of the same module from a STORED H (the backward of se_ff_fwd_f16 with H != NULL, without the weight gradients):
   dZ = ((Drop_o(dY) W2s) .* Drop_h-mask .* Swish'(H)) [M, hid],  dLN = dZ W1 [M, 64].
   W2T = (alpha * W2)^T [hid, 64], W1T = W1^T [64, hid]: scaled fp16 planes; sc->in_amax = max |dY| (device scalar); precision 3 | 16.
   With X != NULL the LayerNorm backward (se_layernorm_bwd with dR = dY, optional dR2) is applied to dLN in registers:
   dX [M, 64] is written instead of dLN and dgamma / dbeta [64] are accumulated. */
int se_ff_bwd_dgrad_f16(const float* dY, const float* H, const float* W2T, const float* W1T, float* dZ, float* dLN, long M,
                        int hid, float drop_p, unsigned seed_h, unsigned seed_o, int precision, const float* X,
                        const float* stats, const float* gamma, const float* dR2, float* dX, float* dgamma, float* dbeta,
                        const se_f16_scales* sc, void* stream);

/* FUSED backward of the same module, weight gradients included (scaled split-fp16 only; hid == 256; csrc/se_ff_fused.hip): one
   persistent launch computes dX = dY + dR2 + LNbwd(dZ W1) (dgamma / dbeta accumulated; out_amax, may be NULL, raised to max |dX|)
   AND dW1 [hid][64] += dZ^T LN(X), db1 += sum dZ, dW2 [64][hid] += alpha (Drop_o dY)^T S, db2 (may be NULL) += alpha sum Drop_o dY
   with H, S = Swish(H) Drop_h-mask and dZ recomputed on chip: nothing [M, hid]-sized is read or written (the backward of
   conformer.py:53-71,128-145 behind se_ff_fwd_f16 with H = NULL).  W1 [hid][64], W2T = (alpha W2)^T [hid][64]: scaled fp16 planes
   with their amax scalars; dy_amax = max |dY|; in_amax / mid_amax (may be NULL: the static exponents ln_sexp / hid_sexp) = the
   proven bounds of |LN(X)| / |Swish(H) mask| the forward scaled its operands with (se_act_bounds); drop_p <= 1/2. */
int se_ff_bwd_fused(const float* dY, const float* X, const float* stats, const float* gamma, const float* beta, const float* W1,
                    const float* b1, const float* W2T, const float* dR2, float* dX, float* dgamma, float* dbeta, float* dW1,
                    float* db1, float* dW2, float* db2, long M, int hid, float drop_p, unsigned seed_h, unsigned seed_o, float alpha,
                    const float* dy_amax, const float* w1_amax, const float* w2t_amax, const float* in_amax, int ln_sexp,
                    const float* mid_amax, int hid_sexp, float* out_amax, void* stream);

/* Input-gradient GEMM of a projection that follows a LayerNorm(64), fused with that LayerNorm's backward
 * (models/conformer.py:67,162: PreNorm -> to_q/to_kv, LayerNorm -> pointwise conv):
 *   dX = dR + LNbwd(A W^T),  dgamma += sum_rows (A W^T) * xhat,  dbeta += sum_rows (A W^T)
 * A [M, K] = gradient w.r.t. the projection's output, W [64, K] (or its pre-split planes, w_planes elements apart; 0 = fp32),
 * X [M, 64] the LayerNorm input, stats [M, 2] its (mean, rstd), dR [M, 64] (may be NULL) the residual-path gradient.  The
 * [M, 64] product never goes to memory (se_gemm_tap + se_layernorm_bwd: one write and one read of it, and one launch, more). */
int se_gemm_ln_bwd(const float* A, const float* W, int w_planes, long M, int K, const float* X, const float* stats,
                   const float* gamma, const float* dR, float* dX, float* dgamma, float* dbeta, void* stream);

/* the same with precision 3 (W = scaled fp16 planes, a_amax / w_amax = the operand scalars, see se_gemm_desc) and/or
 * out_amax (may be NULL): raised to max |dX| */
int se_gemm_ln_bwd_f16(const float* A, const float* W, int w_planes, long M, int K, const float* X, const float* stats,
                       const float* gamma, const float* dR, float* dX, float* dgamma, float* dbeta, int precision,
                       const float* a_amax, const float* w_amax, float* out_amax, void* stream);

/* se_gemm_ln_bwd_f16 AND the weight gradient of the same layer in one sweep (scaled split-fp16 only; K = 192: the qkv projection,
   K = 256: the pointwise-GLU convolution -- models/conformer.py:87-89, 160-164 backwards; csrc/se_lnbwd_fused.hip):
     dX = dR + LNbwd(A WT^T) (dgamma / dbeta accumulated; out_amax, may be NULL, raised to max |dX|),
     dW [K][64] += A^T LN(X), dbias [K] (may be NULL) += column sums of A.
   A [M][K] fp32 with a_amax = max |A| (device scalar); WT = W^T [64][K] as scaled fp16 planes with w_amax; in_amax (may be NULL:
   ln_sexp) = the proven bound of |LN(X)|; dR may be NULL. */
int se_gemm_ln_bwd_wgrad(const float* A, const float* WT, long M, int K, const float* X, const float* stats, const float* gamma,
                         const float* beta, const float* dR, float* dX, float* dgamma, float* dbeta, float* dW, float* dbias,
                         const float* a_amax, const float* w_amax, const float* in_amax, int ln_sexp, float* out_amax, void* stream);

/* weight gradient: dW[n][tap*C + c] += alpha * sum_m dY[m][n] * pro(A[src(m,tap)][c]) (alpha = d->alpha: the factor of a
 * Scale(0.5, .) wrapper goes straight into the gradient buffer);  dW must be initialised by the caller (fp32 atomics across
 * row chunks).  If dbias != NULL also dbias[n] += alpha * sum_m dY[m][n].
 * Replaces the weight-gradient kernels of the same ATen ops as se_gemm_tap. */
int se_gemm_tap_wgrad(const se_gemm_desc* d, const float* A, const float* dY, float* dW, float* dbias,
                      const float* rowstats, const float* pro_scale, const float* pro_shift,
                      int chunks, void* stream);
/* which kernel class the calling thread's last se_gemm_tap_wgrad dispatched to (measurement only: bench.py keys its
 * weight-gradient families by it): bits 0..3 arithmetic (0 fp32 MFMA, 1 bf16x3, 2 bf16x6, 3 scaled f16x3), bits 4..7 class
 * (0 generic tile kernel, 1 triple-tap kernel, 2 whole-gradient token-wise kernel) */
int se_gemm_tap_wgrad_last_kind(void);

/* Weight preparation of a whole model in ONE launch (the per-step re-packing of every parameter the GEMMs read: tap order
 * of nn.Conv2d weights, transposes for the input gradients, the newest-first slab order of DilatedDenseNet
 * (models/generator.py:31), cat(to_q, to_kv), the 0.5 of Scale(0.5, .)), written as fp32 or as the exact three-way bf16 split
 * (hi, mid, lo planes) the six-product kernels consume (se_gemm_desc.w_planes, se_ff_fwd precision | 16):
 *   dst[o_off + o][c_off + t * Ni_dst + i] = scale * src[o' * so + t * stt + i' * si],   o < No, t < Nt, i < Ni
 * (o', i' = o, i with the 64-slabs of i (rev 1) or o (rev 2) reversed).  `items_dev` is a DEVICE array built once by the caller
 * (parameter and arena addresses are stable); max_elems = the largest No * Nt * Ni. */
typedef struct {
  const float* src;
  void* dst;             /* fp32 matrix, or plane 0 of the three bf16 planes */
  int No, Nt, Ni, Ni_dst;
  long so, stt, si;
  int rev;
  int dst_ld, o_off, c_off;
  float scale;
  long plane_stride;     /* 0: fp32 destination; > 0: 16-bit planes, this many elements apart */
  int fmt;               /* planes: 0 = three bf16 planes (exact hi / mid / lo); 1 = two fp16 planes (hi, lo) of
                            scale * src * 2^sexp(*amax) (se_gemm_desc precision 3)                                  */
  float* amax;           /* fmt 1: device scalar of the DESTINATION matrix (all items of one destination share it): must be
                            zero on entry; se_weight_prep first reduces max |scale * src| into it, then splits         */
} se_wprep_item;
int se_weight_prep(const se_wprep_item* items_dev, int nitems, long max_elems, void* stream);

/* generic strided repack: dst[o][t][i] = src[o*so + i*si + t*stt] with optional reversal of the
 * 64-channel slabs of the o or i index (DilatedDenseNet concatenates newest-first,
 * models/generator.py:31).  rev: 0 none, 1 reverse slabs of i, 2 reverse slabs of o. */
int se_repack(const float* src, float* dst, int No, int Nt, int Ni, long so, long stt, long si,
              int rev, int accumulate, void* stream);

int se_unpack(const float* src, float* dst, int No, int Nt, int Ni, long so, long stt, long si,
              int rev, int accumulate, void* stream);

/* ---- operand scales of bounded / copied activations (csrc/se_elem.hip; round 4) ------------------------------------- */
/* dst[r][0..C) = src[r][0..C) for `rows` rows (strides lds / ldd floats, C % 4 == 0) and *amax_out (zero-filled device scalar, may be
 * NULL) raised to max |src|: the slab copy at a decoder's entry (models/generator.py:84,113) */
int se_copy_cols_amax(const float* src, int lds, float* dst, int ldd, long rows, int C, float* amax_out, void* stream);
/* proven bound of a normalised activation from the current parameters:
 *   b = (k max|g| + max|b|) max(1, max|alpha|);  W: b = b max_j sum_i |W[j][i]| + max|wb|;  b *= post;  *out = b
 * k = kconst (ksel 0: LayerNorm(64): sqrt(63)), or the run-time arguments k1 / k2 of se_act_bounds (ksel 1 / 2: sqrt(count - 1)
 * of a BatchNorm / InstanceNorm whose count is known per call).  One launch per forward over a device-resident item table. */
typedef struct {
  const float* g; const float* b; const float* alpha;   /* gamma [n], beta [n] or NULL, PReLU slopes [na] or NULL */
  const float* W; const float* wb;                      /* optional linear layer behind the norm: W [rows][cols], bias [rows] or NULL */
  int n, na, rows, cols, ksel;
  float kconst, post;
  float* out;                                           /* one item per scalar */
} se_bound_item;
int se_act_bounds(const se_bound_item* items_dev, int nitems, float k1, float k2, void* stream);

/* ---- normalisation (csrc/se_norms.hip) ------------------------------------------------------ */
/* per-row (mean, rstd) over C=64 channels: the statistics of every nn.LayerNorm(64) that feeds a GEMM
 * prologue (models/conformer.py:67,162) */
int se_row_stats(const float* X, float* stats, long M, int C, int ld, float eps, void* stream);
/* Y = LN(X)*g + b (+ R): ConformerBlock.post_norm + the TSCB residual (conformer.py:204,211;
 * generator.py:70,72).  stats (optional) receives (mean, rstd) per row. */
int se_layernorm_fwd(const float* X, const float* g, const float* b, const float* R, float* Y,
                     float* stats, long M, int C, float eps, void* stream);
/* the same, also emitting out_stats[M][2] = (mean, rstd) of the rows of Y (what se_row_stats(Y) would give): the statistics the
 * next block's first LayerNorm needs (conformer.py:204 -> :67 of the following block) without another pass over Y; out_stats may be NULL */
int se_layernorm_fwd_stats(const float* X, const float* g, const float* b, const float* R, float* Y, float* stats,
                           float* out_stats, long M, int C, float eps, void* stream);
/* dX = (dR) + (dR2) + LayerNorm backward of dY; dg += sum dY*xhat, db += sum dY (fp32 atomics; caller zeroes) */
int se_layernorm_bwd(const float* X, const float* stats, const float* g, const float* dY,
                     const float* dR, const float* dR2, float* dX, float* dg, float* db, long M, int C,
                     void* stream);
/* the same; amax_out (may be NULL): zero / running maximum on entry, raised to max |dX| (operand scale of the scaled split-fp16
 * kernels that read dX next) */
int se_layernorm_bwd_amax(const float* X, const float* stats, const float* g, const float* dY, const float* dR,
                          const float* dR2, float* dX, float* dg, float* db, long M, int C, float* amax_out, void* stream);
/* stats[b][c][2] += (sum, sumsq) over the P pixels of batch b (fp64 atomics; caller zeroes).  Feeds
 * nn.InstanceNorm2d (generator.py:21,40,46,101,120; discriminator.py:40-49) and nn.BatchNorm1d
 * (conformer.py:167) when the producer kernel did not already emit them. */
int se_col_stats(const float* X, int ld, int x_off, double* stats, int B, long P, int C, void* stream);
/* (sum, sumsq) -> mr[nb][C][2] = (mean, rstd) and ss[nb][C][2] = (rstd*g, beta - mean*rstd*g); when
 * running_mean != NULL also the BatchNorm running-statistics update (momentum, unbiased var). */
int se_norm_finalize(const double* stats, const float* g, const float* beta, float* mr, float* ss,
                     int nb, int C, double count, float eps, float* running_mean, float* running_var,
                     float momentum, void* stream);
int se_bn_eval_scale(const float* rm, const float* rv, const float* g, const float* beta, float* ss,
                     float* mr, int C, float eps, void* stream);
/* Y[.., y_off+c] = prelu(X*scale + shift): the apply pass of InstanceNorm2d(affine)+PReLU */
int se_affine_prelu(const float* X, int ldx, int x_off, const float* ss, const float* slope, float* Y,
                    int ldy, int y_off, int B, long P, int C, void* stream);
/* se_norm_finalize (instance statistics, no running buffers) + se_affine_prelu in one launch: stats[B][C][2] (sum, sumsq) ->
 * Y = prelu(xhat*g + beta) and mr[B][C][2] = (mean, rstd) for the backward pass (generator.py:21-22,40-47,101-104,120-121;
 * discriminator.py:40-50) */
int se_inorm_prelu_fwd(const float* X, int ldx, int x_off, const double* stats, const float* g, const float* beta,
                       const float* slope, float* Y, int ldy, int y_off, float* mr, int B, long P, int C, double count,
                       float eps, void* stream);
/* the same, raising the zero-filled device scalar *amax_out (may be NULL) to max |Y|: the operand scale (se_gemm_desc.a_amax) of the
 * scaled split-fp16 convolutions that read Y -- measured, so that no promise about gamma / beta is needed */
int se_inorm_prelu_fwd_amax(const float* X, int ldx, int x_off, const double* stats, const float* g, const float* beta,
                       const float* slope, float* Y, int ldy, int y_off, float* mr, int B, long P, int C, double count,
                       float eps, float* amax_out, void* stream);
/* backward of Y = act(xhat*g + beta), act = PReLU (act=0; slope NULL = identity) or Swish (act=1), for
 * instance (per_batch=1) or batch (per_batch=0; BatchNorm1d+Swish, conformer.py:167-168) statistics;
 * red: workspace double[nb][C][3]; dg/dbeta/dslope accumulate.  phase bits: 1 = reduction, 4 = parameter
 * gradients (from the local sums), 2 = apply (data-parallel SyncBatchNorm all-reduces `red` between 1|4 and 2), 8 = the parameter
 * gradients are taken inside the apply pass instead of their own launch (alternative to 4: only without an exchange between the
 * passes), 16 = the caller hands `red` over zero-filled (no memset launch); count = elements per statistic (global count under
 * SyncBatchNorm). */
int se_norm_prelu_bwd(const float* X, int ldx, int x_off, const float* mr, const float* g,
                      const float* beta, const float* slope, const float* dY, int ldy, int y_off,
                      double* red, float* dX, int lddx, int dx_off, float* dg, float* dbeta,
                      float* dslope, int B, long P, int C, int per_batch, int act, int phase, double count,
                      void* stream);
/* the same; amax_out (may be NULL): device scalar, zero or a running maximum on entry, raised to max |dX| by the apply pass (one
 * atomic per wave) -- the operand scale of the scaled split-fp16 kernels that read dX next (se_gemm_desc precision 3: a_amax of
 * the conv input gradient, w_amax of the conv weight gradient) */
int se_norm_prelu_bwd_amax(const float* X, int ldx, int x_off, const float* mr, const float* g,
                           const float* beta, const float* slope, const float* dY, int ldy, int y_off,
                           double* red, float* dX, int lddx, int dx_off, float* dg, float* dbeta,
                           float* dslope, int B, long P, int C, int per_batch, int act, int phase, double count,
                           float* amax_out, void* stream);

/* ---- fused relative-position attention (csrc/se_attn.hip) ----------------------------------- */
/* Attention.forward without the projections (models/conformer.py:103-122): QKV [tokens][192] (q|k|v, head h =
 * columns 16h..16h+15 of each third), E = rel_pos_emb.weight [2*maxpos+1][16] -> O [tokens][64], LSE
 * [tokens][4].  token(s,p) = (s/inner)*outer_stride + (s%inner)*inner_stride + p*pos_stride. */
int se_attn_fwd(const float* QKV, const float* E, float* O, float* LSE, int nseq, int n, int inner,
                long outer_stride, long inner_stride, long pos_stride, int maxpos, float scale, void* stream);
/* the same with the embedding table also handed over PRE-SPLIT (se_weight_prep: three bf16 planes [2*maxpos+1][16], es_plane
 * elements apart; Es may be NULL): the split-bf16 forward kernel then loads its E fragments instead of splitting them per key step */
int se_attn_fwd_es(const float* QKV, const float* E, const void* Es, long es_plane, float* O, float* LSE, int nseq, int n,
                   int inner, long outer_stride, long inner_stride, long pos_stride, int maxpos, float scale, void* stream);
/* backward: dQKV [tokens][192] written, dE accumulated (caller zeroes); ws = workspace of
 * se_attn_bwd_workspace_bytes(ntok, maxpos, nseq, n) bytes, 16-byte aligned (softmax row constants, fp16-split copies of E, the
 * per-item dE tiles of the scaled split-fp16 kernel).  This entry runs the fp32-MFMA kernels (dK / dV pass, dQ / dE pass) for any
 * sequence length -- the cross-check path of se_attn_bwd_f16_phase and the backward of shapes outside it. */
int se_attn_bwd(const float* QKV, const float* E, const float* O, const float* dO, const float* LSE,
                float* dQKV, float* dE, int nseq, int n, int inner, long outer_stride,
                long inner_stride, long pos_stride, long ntok, int maxpos, float scale, void* ws, size_t ws_bytes,
                void* stream);

/* Scaled split-fp16 forms of the same attention (round 3; models/conformer.py:103-122): every operand is x * 2^sexp = hi + lo in
 * two fp16 planes (the scale from a measured maximum, see se_gemm_desc.precision 3), every 16-deep product three
 * v_mfma_f32_16x16x16_f16: fp32-equivalent results (tests/test_attn_gpu.py: 5e-6 / 2e-5 vs fp64 like the bf16 kernels) with
 * 8-instruction operand splits instead of 18.  Sequence shapes of the backward: n <= 336 (21 key tiles), 16 ceil(n / 16) + 128 <=
 * maxpos, maxpos % 16 == 0, 32-bit lane offsets (pos_stride * 192 * padded n < 2^31); of the forward: the V image of one (sequence,
 * head) within the LDS (n <= 4079): anything else is a host error -- use se_attn_fwd_es / se_attn_bwd.
 *   Es        : TWO fp16 planes [2][>= 2 maxpos + 1][16] of E * 2^sexp(*e_amax) (se_weight_prep fmt 1), es_plane elements apart
 *   qkv_amax  : device scalar >= max |QKV| (e.g. raised by the qkv GEMM: se_gemm_desc.y_amax)
 *   do_amax   : device scalar >= max |dO|  (the to_out input-gradient GEMM likewise)
 *   dqkv_amax : optional zero-initialised device scalar raised to max |dQKV| (the scale of the gradient's consumers)
 *   delta     : optional [tokens][4] table of the softmax-backward row constants rowsum(dO . O) per head (e.g. written by the
 *               to_out input-gradient GEMM: SE_EPI_DELTA); NULL: computed here from O and dO (one more launch); O may be NULL when
 *               delta is given
 * The backward splits E itself (and measures its maximum) in its workspace.  phase 1 = everything but the reduction of the per-item dE tiles,
 * phase 2 = that reduction alone (reads ws, accumulates dE) -- a leaf of the backward graph that the caller may issue on another stream
 * behind phase 1; phase 3 = both. */
int se_attn_fwd_f16(const float* QKV, const void* Es, long es_plane, const float* qkv_amax, const float* e_amax, float* O,
                    float* LSE, int nseq, int n, int inner, long outer_stride, long inner_stride, long pos_stride, int maxpos,
                    float scale, void* stream);
int se_attn_bwd_f16_phase(const float* QKV, const float* E, const float* O, const float* dO, const float* LSE,
                          const float* delta, const float* qkv_amax, const float* do_amax, float* dqkv_amax, float* dQKV, float* dE,
                          int nseq, int n, int inner, long outer_stride, long inner_stride, long pos_stride, long ntok, int maxpos,
                          float scale, void* ws, size_t ws_bytes, int phase, void* stream);

/* ---- depthwise conv k=31 along the sequence axis (csrc/se_dwconv.hip) -------------------------- */
/* DepthWiseConv1d forward (models/conformer.py:40-48,166) on [tokens][128] (+ fp64 BatchNorm statistics
 * [128][2] when stats != NULL); flip=1 with bias=NULL is the input gradient. */
int se_dwconv31(const float* X, const float* W, const float* bias, float* Y, double* stats, int flip,
                int nseq, int n, int inner, long outer_stride, long inner_stride, long pos_stride, void* stream);
/* input gradient of DepthWiseConv1d fused with the backward of the GLU in front of it (conformer.py:164-166 backwards):
 * dU = flipped-tap depthwise conv of dH [tokens][128] (never written), dZ [tokens][256] = (dU sigmoid(g), dU u (1 - sigmoid(g)))
 * with U = a sigmoid(g) the forward GLU result and G = the gate half g of the pre-GLU activations, both [tokens][128] (a GEMM
 * with SE_EPI_GLU | SE_EPI_GLU_GATE writes exactly these two); amax_out (may be NULL): raised to max |dZ| */
int se_dwconv31_glu_bwd(const float* dH, const float* W, const float* U, const float* G, float* dZ, float* amax_out, int nseq,
                        int n, int inner, long outer_stride, long inner_stride, long pos_stride, void* stream);
/* weight / bias gradient (accumulated into dW [128][31], dbias [128]); ws = workspace of
 * se_dwconv31_wgrad_workspace_bytes() bytes (per-workgroup partial sums, reduced in a fixed order: deterministic) */
size_t se_dwconv31_wgrad_workspace_bytes(void);
/* Round 5: input gradient + GLU backward (as se_dwconv31_glu_bwd) AND the weight / bias gradient (as se_dwconv31_wgrad with X = U,
 * dY = dH) of DepthWiseConv1d (models/conformer.py:40-48,164-166 backwards) in ONE sweep over (dH, U, G): the weight gradient uses the
 * rows the input-gradient FIR already holds in LDS.  dW [128][31], dbias [128] (may be NULL) are accumulated; ws as above; ntok = rows of
 * the [tokens][128] operands (32-bit buffer offsets: ntok * 1024 B < 4 GiB, checked). */
int se_dwconv31_bwd_fused(const float* dH, const float* W, const float* U, const float* G, float* dZ, float* amax_out, float* dW,
                          float* dbias, float* ws, long ntok, int nseq, int n, int inner, long outer_stride, long inner_stride,
                          long pos_stride, void* stream);
int se_dwconv31_wgrad(const float* X, const float* dY, float* dW, float* dbias, int nseq, int n, int inner,
                      long outer_stride, long inner_stride, long pos_stride, float* ws, void* stream);

/* ---- front-end glue, output assembly, losses, optimizers (csrc/se_elem.hip) --------------------- */
/* normalize_batch (core/function.py:647-659): c[b] = sqrt(L / sum x^2) */
int se_clip_scale(const float* x, float* c, int B, int L, void* stream);
/* center=True reflect padding of torch.stft (core/function.py:690) fused with the clip scale */
int se_reflect_pad_scale(const float* x, const float* c, float* xp, int B, int L, int pad, void* stream);
int se_reflect_pad_bwd(const float* dxp, const float* c, float* dx, int B, int L, int pad, void* stream);
/* power_compress / power_uncompress (core/function.py:625-645); comp: 0 none, 1 pow 0.3, 2 log1p */
int se_compress_planes(const float* R, int ldr, float* P, long rows, int F, int comp, float pre_scale, void* stream);
int se_compress_planes_bwd(const float* R, int ldr, const float* dP, float* dR, long rows, int F, int comp,
                           float pre_scale, void* stream);
int se_uncompress_rows(const float* P, float* A, int lda, long rows, int F, int comp, float post_scale, void* stream);
int se_uncompress_rows_bwd(const float* P, const float* dA, int lda, float* dP, long rows, int F, int comp,
                           float post_scale, void* stream);
/* Fused front-end for n_fft = 400, hop = 100 (csrc/se_front.hip), one launch each:
 *   se_stft_fused : x [B, L] (* c[b] if c != NULL) -> planes [B, L/100 + 1, 201, 4] = compressed (|z|, Re, Im, 0); replaces
 *                   torch.stft (center / reflect, periodic Hamming) + power_compress (core/function.py:685-693, 625-634).
 *                   Wf = [400][416]: column 2f = w[k] cos(2 pi f k / 400), 2f + 1 = -w[k] sin(.)
 *   se_istft_fused: planes -> y [B, 100 (T - 1)]; replaces power_uncompress + torch.istft (core/function.py:695-703, 636-645).
 *                   Wi = [404][416]: row 2f = c_f w[n] cos(2 pi f n / 400) / 400, row 2f + 1 = -c_f w[n] sin(.) / 400;
 *                   env = window envelope [100 (T - 1) + 400] */
int se_stft_fused(const float* x, const float* c, const float* Wf, float* P, int B, int L, int n_fft, int hop, int comp,
                  float pre_scale, void* stream);
int se_istft_fused(const float* P, const float* Wi, const float* env, float* y, int B, int T, int n_fft, int hop, int comp,
                   float post_scale, void* stream);
/* overlap-add / envelope division / trim of torch.istft (core/function.py:701-702) and its transpose */
int se_ola(const float* Fr, const float* env, float* y, int B, int T, int n_fft, int hop, int trim, int L, void* stream);
int se_ola_bwd(const float* dy, const float* env, float* dFr, int B, int T, int n_fft, int hop, void* stream);
/* TSCNet output assembly (models/generator.py:158-167) and MaskDecoder tail (:110-112) */
int se_assemble(const float* mask, int ldm, const float* nin, const float* cplx, float* est, long n, void* stream);
int se_assemble_bwd(const float* est, const float* dest, const float* nin, float* dmask, int ldm, float* dcplx,
                    long n, void* stream);
int se_mask_tail(const float* U, int ldu, const float* wb, const float* slope, float* M, long n, int F, void* stream);
int se_mask_tail_bwd(const float* U, int ldu, const float* wb, const float* slope, const float* dM, float* dU,
                     double* dwb, float* dslope, long n, int F, void* stream);
/* GLU backward (models/conformer.py:30-37) */
int se_glu_bwd(const float* Z, const float* dU, float* dZ, long M, int H, void* stream);
int se_glu_bwd_amax(const float* Z, const float* dU, float* dZ, long M, int H, float* amax_out, void* stream);   /* + max |dZ| */
/* the same from the GLU result U = a sigmoid(g) [M][H] and the gate half G [M][H] (SE_EPI_GLU | SE_EPI_GLU_GATE keeps only these) */
int se_glu_bwd_gate(const float* U, const float* G, const float* dU, float* dZ, long M, int H, float* amax_out, void* stream);
/* MergeBlock gate of the TSC-diffusion hybrid (models/tsc_diffusion.py:34-35): Y [M][2C] (gate | filter) -> G [M][C] = sigmoid(gate) tanh(filter) */
int se_gate_tanh(const float* Y, float* G, long M, int C, void* stream);
/* its backward (training of the hybrid, core/function.py:453-532): dY [M][2C] from Y and dG [M][C] */
int se_gate_tanh_bwd(const float* Y, const float* dG, float* dY, long M, int C, void* stream);
/* loss reductions of train_gan (core/function.py:251-258) and their gradient seeds (`up` = device scalars) */
int se_spec_loss(const float* A, const float* Bp, double* sums, long n, void* stream);
int se_spec_loss_bwd(const float* A, const float* Bp, float* dA, const float* up, float cmag, float cri, long n,
                     int accumulate, void* stream);
int se_l1_loss(const float* a, long lda, const float* b, long ldb, double* sums, long rows, int L, void* stream);
int se_l1_loss_bwd(const float* a, long lda, const float* b, long ldb, float* da, const float* up, float ck,
                   long rows, int L, void* stream);
/* flat-buffer optimizers (core/optimizer.py:33-36) and the self-correcting-weight helpers (:719-752) */
int se_adamw(float* p, const float* g, float* m, float* v, long n, float lr, float b1, float b2, float eps,
             float wd, int step, void* stream);
int se_sgd_nesterov(float* p, const float* g, float* buf, long n, float lr, float momentum, int first, void* stream);
int se_dot(const float* a, const float* b, double* out, long n, void* stream);
/* torch.nn.utils.clip_grad_norm_ (core/function.py:275-276, 311-312) on a flat gradient buffer: g *= min(1, max_norm /
 * (sqrt(sums[0] + .. + sums[nsum-1]) + 1e-6)); sums = device partial sums of squares (se_dot(g, g) per buffer). */
int se_grad_clip(float* g, long n, const double* sums, int nsum, float max_norm, void* stream);
/* LARS.step (core/optimizer.py:71-113) on one flat parameter group.  seg_off[nseg+1]: element offsets of the
 * parameter tensors inside the flat buffer; max_seg: the largest tensor; norms: workspace double [nseg][2]
 * (se_segnorm_workspace_bytes(nseg)).  adapt = 1 for the ndim > 1 group (weight decay + trust ratio), 0 for the 1-D group. */
int se_lars_step(float* p, const float* g, float* mu, const long* seg_off, int nseg, long max_seg, double* norms,
                 int adapt, float lr, float wd, float momentum, float trust_coef, void* stream);
/* Lamb.step (core/optimizer.py:176-238) on one flat parameter group: gsum[nsum] = partial sums of squares of ALL the
 * model's gradients (global-norm clip, :186-199; nsum == 0 disables it); bc1 / bc2 = bias corrections
 * (1 when off); beta3 = 1 - b1 with grad_averaging else 1; adapt = (wd != 0 || always_adapt). */
int se_lamb_step(float* p, const float* g, float* m, float* v, const long* seg_off, int nseg, long max_seg,
                 double* norms, const double* gsum, int nsum, float max_grad_norm, int adapt, int trust_clip, float lr,
                 float wd, float b1, float b2, float beta3, float eps, float bc1, float bc2, void* stream);
int se_axpbypcz(const float* a, const float* b, const float* c, float* y, float alpha, float beta, float gamma,
                long n, void* stream);

/* ---- metric-discriminator tail + spectral normalisation (models/discriminator.py:39-57; csrc/se_elem.hip) ---- */
/* torch.nn.utils.spectral_norm on n <= 8 weight matrices [h_i, w_i] in ONE launch (host arrays of device pointers / sizes;
 * the six spectral-norm layers of the discriminator): one power iteration each (train != 0: u_i [h_i], v_i [w_i] updated in
 * place), sigma[i] = u_i . (W_i v_i), Wn_i = W_i / sigma[i].  Backward: dW_i += (dWn_i - <dWn_i, Wn_i> u_i v_i^T) / sigma[i];
 * a NULL dWn[i] skips matrix i. */
int se_spectral_norm(int n, const float* const* W, float* const* u, float* const* v, float* const* Wn, const int* h,
                     const int* w, float* sigma, int train, float eps, void* stream);
int se_spectral_norm_bwd(int n, const float* const* dWn, const float* const* Wn, const float* const* u,
                         const float* const* v, const float* sigma, float* const* dW, const int* h, const int* w, void* stream);
/* AdaptiveMaxPool2d(1) over the P positions of A [B, P, 128] -> Linear(128, 64) -> dropout mask [B, 64] (NULL: none; entries
 * 0 or 1 / (1 - p)) -> PReLU(64) -> Linear(64, 1) -> beta * sigmoid(slope * z): out [B].  ws = workspace of
 * se_disc_tail_workspace_bytes(B) bytes, kept for the backward (dA must be zero-initialised; parameter gradients accumulate). */
size_t se_disc_tail_workspace_bytes(int B);
int se_disc_tail_fwd(const float* A, int B, int P, const float* W1, const float* b1, const float* mask, const float* slope1,
                     const float* W2, const float* b2, const float* sslope, float beta, float* out, float* ws, void* stream);
int se_disc_tail_bwd(const float* dout, const float* ws, int B, int P, const float* W1, const float* mask, const float* slope1,
                     const float* W2, const float* sslope, float beta, float* dA, float* dW1, float* db1, float* dslope1,
                     float* dW2, float* db2, float* dsslope, void* stream);

/* ---- the discriminator's first stage as direct kernels (csrc/se_thin.hip; round 4) ------------------------------------- */
/* Conv2d(2, 16, 4, stride 2, padding 1, bias=False) of models/discriminator.py:39 on the transposed [T, F] image: X planes
 * [B][T][F][4] (channels 0, 1), W in the PyTorch layout [16][2][4][4] (already spectrally normalised), R / dR [B][To][Fo][16],
 * To = (T - 2) / 2 + 1.  fwd: stats (may be NULL) += fp64 (sum, sum of squares) per (b, channel) (the InstanceNorm that follows);
 * dgrad: dX [B][T][F][4] written (channels 2, 3 zero); wgrad: dW [16][2][4][4] accumulated (zero-fill first). */
int se_dconv1_fwd(const float* X, const float* W, float* R, double* stats, int B, int T, int F, int N, void* stream);
int se_dconv1_dgrad(const float* dR, const float* W, float* dX, int B, int T, int F, int N, void* stream);
int se_dconv1_wgrad(const float* X, const float* dR, float* dW, int B, int T, int F, int N, void* stream);

/* Input gradient of Conv2d(Cin, N, 4, 2, 1) (models/discriminator.py:42-50, stages 2 - 4) by parity class: dR [B][To][Fo][N]
 * (To = (Ti - 2) / 2 + 1), Wd [Cin][16 taps = kh * 4 + kw][N] (the packed input-gradient matrix), dX [B][Ti][Fi][Cin] written
 * whole.  N % 32 == 0, Cin <= 64, Cin % 4 == 0. */
int se_dconv_dgrad(const float* dR, const float* Wd, float* dX, int B, int Ti, int Fi, int N, int Cin, void* stream);

/* ---- the generator's thin convolutions as direct kernels (csrc/se_thin.hip; round 4) --------------------------------------- */
/* Conv2d(64, n, (1, 2)), n = 1 (models/generator.py:114, mask decoder) or 2 (:128, complex decoder): X [B T][F2][64] channels-last,
 * W [n][64][1][2] and bias [n] in the PyTorch layout, Y / dY [B T][F2 - 1][4] (channels >= n written as zero / ignored); stats (may
 * be NULL) += fp64 (sum, sum of squares) per (b, channel) as double [B][4][2]; dgrad writes dX [rows][F2][64]; wgrad accumulates
 * dW [n][64][1][2] and dbias [n]. */
int se_conv1x2_fwd(const float* X, const float* W, const float* bias, float* Y, double* stats, int B, int T, int F2, int n, void* stream);
int se_conv1x2_dgrad(const float* dY, const float* W, float* dX, long rows, int F2, int n, void* stream);
int se_conv1x2_wgrad(const float* X, const float* dY, float* dW, float* dbias, long rows, int F2, int n, void* stream);
/* Conv2d(3, 64, (1, 1)) of the encoder (models/generator.py:39): X planes [B][P][4] (channel 3 ignored), W [64][3][1][1], bias [64],
 * R / dR [B][P][64]; stats (may be NULL) += double [B][64][2]; wgrad accumulates dW [64][3] and dbias [64] over npix = B P pixels. */
int se_conv3to64_fwd(const float* X, const float* W, const float* bias, float* R, double* stats, int B, long P, void* stream);
int se_conv3to64_wgrad(const float* X, const float* dR, float* dW, float* dbias, long npix, void* stream);

/* ---- CDiffuSE denoiser glue (models/DiffuSE.py; csrc/se_diffuse.hip), channels-last [B, L, C] maps ---------------------- */
/* SpectrogramUpsampler stage: ConvTranspose2d(1,1,[3,20], stride [1,10], padding [1,5]) + leaky_relu(0.4) on in [B][F][Tin];
 * layout 0: out [B][F][10 Tin], layout 1: out [B][10 Tin][ldo] (channels-last, the conditioner GEMM operand) */
int se_diff_upsample(const float* in, const float* w, const float* bias, float* out, int B, int F, int Tin, int layout, int ldo,
                     void* stream);
/* x = relu(w a + b) (input_projection, models/DiffuSE.py:150-151), y = x + d0[b or 0] (first diffusion_projection, :113-116) */
int se_diff_input(const float* audio, const float* w, const float* bias, const float* d0, int dB, float* x, float* y, int B,
                  long L, int C, void* stream);
/* the same, and the device scalar y_amax (zero-filled by the caller, may be NULL) is raised to max |y|: the operand scale of the
 * scaled split-fp16 dilated conv (se_gemm_desc precision 3, a_amax) that reads y */
int se_diff_input_amax(const float* audio, const float* w, const float* bias, const float* d0, int dB, float* x, float* y, int B,
                       long L, int C, float* y_amax, void* stream);
/* y = sigmoid(z[:C]) * tanh(z[C:]), z = R * scale + shift (GroupNorm of the dilated conv) + cond   (:117-122) */
int se_diff_gate(const float* R, const float* ss, const float* cond, float* y, int B, long L, int C, void* stream);
/* x <- (x + R2[:C]) / sqrt(2); ynext = x + d_next; skip (+)= GroupNorm(R2[C:])   (:124-127, 155-158) */
int se_diff_mix(float* x, const float* R2, const float* ss, const float* d_next, int dB, float* ynext, float* skip, int first,
                int B, long L, int C, void* stream);
/* the same, and y_amax (zero-filled, may be NULL) is raised to max |ynext| */
int se_diff_mix_amax(float* x, const float* R2, const float* ss, const float* d_next, int dB, float* ynext, float* skip, int first,
                     int B, long L, int C, float* y_amax, void* stream);
/* one-stream form (round 5): only y = x + d_step lives between the layers of the CDiffuSE denoiser (models/DiffuSE.py:113,124-127):
 * y <- ((y - d_cur) + residual) / sqrt 2 + d_next in place (d_next == NULL: last layer, y is left alone), skip sum as in se_diff_mix;
 * se_diff_input(_amax) accepts x == NULL. */
int se_diff_mix_y(float* y, const float* R2, const float* ss, const float* d_cur, const float* d_next, int dB, float* skip, int first,
                  int B, long L, int C, float* y_amax, void* stream);
/* nn.GroupNorm statistics -> ss [B][N][2] = (scale, shift) per (batch, channel) for channels [c_off, c_off + N) of the
 * fp64 (sum, sumsq) table stats [B][Ntot][2] a GEMM epilogue produced (SE_EPI_STATS), groups of gsize channels */
int se_group_finalize(const double* stats, int B, int Ntot, int c_off, int N, int gsize, double count_per_channel,
                      const float* gamma, const float* beta, float* ss, float eps, void* stream);
/* out[pos] = b + sum_c relu(h[pos][c]) w[c]   (:159-161) */
int se_diff_out(const float* h, const float* w, const float* bias, float* out, long npos, int C, void* stream);

#ifdef __cplusplus
}
#endif
#endif
