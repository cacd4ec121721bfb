/*
 * agx.h - C-ABI of libagx.so: MI355X (gfx950) kernels for the WGAN-GP cGAN
 * train step and the gene x gene Pearson / gamma evaluation of
 * rvinas/adversarial-gene-expression.
 *
 * The reference has NO native/FFI layer: its boundary is the Python function
 * surface of tf_utils.py / utils.py.  Each entry point
 * below therefore cites the reference Python lines whose arithmetic it
 * replaces; the Python mirror (adversarial_gene_expression_amd/{tf_utils,utils}.py)
 * binds them with ctypes.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in _host;
 *   - matrices are fp32 row-major with an explicit leading dimension (floats);
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream);
 *   - every function returns 0 on success, <0 = AGX_E_*; agx_last_error()
 *     gives a thread-local message.  Nothing throws across the ABI;
 *   - no entry point allocates or frees device memory; scratch comes in via
 *     explicit workspace arguments whose size is given by a *_ws_bytes query.
 */
#ifndef AGX_H
#define AGX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AGX_OK 0
#define AGX_E_INVALID -1  /* bad argument / shape / alignment */
#define AGX_E_HIP -2      /* a HIP runtime call failed */
#define AGX_E_WORKSPACE -3 /* workspace too small */
#define AGX_E_STATE -4    /* handle not bound / wrong call order */

#define AGX_MAX_LAYERS 8
#define AGX_MAX_CAT 16

/* ---- library ---------------------------------------------------------- */
int agx_version(void);
const char *agx_last_error(void);
/* Select `device`, verify it is a gfx950 part, cache its CU count. */
int agx_init(int device);
/* sizeof() of the by-pointer structs below as this library was compiled (binding self-check):
 * 0 agx_gemm_t, 1 agx_embed_t, 2 agx_small_grads_t, 3 agx_rng_seg_t, 4 agx_mlp_desc_t,
 * 5 agx_cgan_config_t, 6 agx_step_args_t, 7 agx_iter_args_t; -1 for any other index. */
int64_t agx_sizeof(int32_t which);

/* ---- in-stream kernel timing (measurement only; no reference counterpart) ----
 * When enabled, every launch site tagged with a kernel-class id brackets its launch(es)
 * with a pair of HIP events on the SAME stream; agx_prof_read synchronises those events and
 * returns the summed duration together with the algorithmic FLOPs / bytes of the launches
 * (2*M*N*K and 4*(M*K + K*N + M*N) for a GEMM, 20 or 28 B/param for RMSprop / Adam).
 * filter: -2 off (default), -1 every class, >= 0 that class only.                         */
enum {
  AGX_K_GEN_FWD_HID = 0, AGX_K_GEN_FWD_OUT, AGX_K_DISC_FWD_IN, AGX_K_DISC_FWD_HID,
  AGX_K_DISC_CHAIN, AGX_K_DISC_DX, AGX_K_GP2_IN, AGX_K_GP2_HID, AGX_K_DISC_DW_IN,
  AGX_K_DISC_DW_HID, AGX_K_DISC_DCOV, AGX_K_GEN_DW_OUT, AGX_K_GEN_DW_HID, AGX_K_GEN_BWD_OUT,
  AGX_K_GEN_BWD_HID, AGX_K_OPT_RMSPROP, AGX_K_OPT_ADAM, AGX_K_CORR_XTX, AGX_K_GEMM_UNTAGGED,
  AGX_K_COUNT
};
int agx_prof_enable(int32_t filter);
int agx_prof_reset(void);
int agx_prof_read(int32_t id, double *total_ms, int64_t *count, double *flops, double *bytes);
const char *agx_prof_name(int32_t id);
/* Queue a bounded busy-wait (<= 100 ms) so that the host can enqueue work behind it: brackets
 * recorded after it time back-to-back GPU execution instead of host launch gaps.  Brackets are
 * NOT usable inside a captured graph (ROCm 7.2 rejects hipEventElapsedTime on graph-recorded
 * events): agx_cgan_iter_capture refuses to capture while profiling is enabled.            */
int agx_prof_spin(void *stream, int32_t microseconds);
/* mean elapsed time (us) of n empty brackets on the stream: the part of a bracketed duration that
 * the two event records add (bench.py reports its roofline duration with this subtracted) */
int agx_prof_empty_bracket_us(void *stream, int32_t n, double *mean_us);
int agx_prof_enabled(void);
/* Clock probe: out_dev[0] = shader MHz seen by a kernel of `blocks` x 4 waves issuing `mfmas`
 * dependent 32x32x2 fp32 MFMAs each, out_dev[1] = cycles per MFMA (measurement only). */
int agx_prof_clock(void *stream, int32_t blocks, int32_t mfmas, float *out_dev);
/* Resident workgroups per CU the HIP runtime computes for a main tile configuration (0 = 64x64 NT, 1 = 64x64 TN with
 * the fused optimizer, 2 = 128x128 NN 8-wave, 3 = 128x128 SYRK, 4 = 64x64 NN, 5 = 128x128 NN 4-wave) and its dynamic LDS
 * bytes. */
int agx_prof_occupancy(int32_t which, int32_t *blocks_per_cu, int32_t *lds_bytes);

/* ---- dense layers: tf_utils.py:85-88, 132-135 (Dense+ReLU) and their
 *      autodiff (MatMul grads / BiasAddGrad / ReluGrad, tf_utils.py:220,255) -- */
typedef struct agx_gemm {
  /* C[M,N] = epilogue( alpha * op_a(A) . op_b(B) ) on fp32 MFMA (exact fp32). */
  const float *A; /* ta=0: [M,K] lda ; ta=1: stored [K,M] lda  (dW = X^T . delta) */
  const float *B; /* tb=0: [K,N] ldb ; tb=1: stored [N,K] ldb  (dX = delta . W^T) */
  float *C;       /* [M,N] ldc */
  int32_t M, N, K;
  int32_t lda, ldb, ldc;
  int32_t ta, tb;
  float alpha;
  const float *bias;    /* [N] or NULL: + bias[n]                               */
  const float *addrow;  /* [addmod, N] ld=ldadd or NULL: + addrow[m % addmod][n] */
  int32_t addmod, ldadd;
  int32_t act;          /* 0 none; 1 leaky-ReLU with `slope` (0 => ReLU)         */
  float slope;
  const float *mask;    /* [M,N] ld=ldmask or NULL: v *= (mask>0 ? 1 : slope);   */
  int32_t ldmask;       /*   may alias C (in-place ReLU-grad)                    */
  int32_t accumulate;   /* 1: C += result                                        */
  const float *rowscale;/* [M] or NULL: v *= rowscale[m] (applied before bias)   */
  float *rowsq;         /* [M, rowsq_ld] or NULL: rowsq[m][n/32] = sum over the   */
  int32_t rowsq_ld;     /*   32-column slab of C[m,:]^2 (deterministic partials;  */
                        /*   disables split-K); rowsq_ld >= ceil(N/32)            */
  float *ws;            /* split-K scratch (floats) or NULL                      */
  int64_t ws_floats;
  int32_t force_cfg;    /* -1 auto; 0/1/2 = 128x128 / 64x64 / 32x128 LDS tiles, 3 = direct 32x32 */
  int32_t force_splitk; /* <=0 auto                                              */
  int32_t prof_id;      /* AGX_K_* class for agx_prof_*; 0 with untagged = AGX_K_GEMM_UNTAGGED when prof_tagged == 0 */
  int32_t prof_tagged;  /* 1 if prof_id is meaningful                            */
  int32_t interp_rows;  /* R > 0 (then M == 2R, C has 3R rows): A = [real; fake]; C rows  */
  const float *interp_alpha; /* [0,2R) as usual and C[2R+b] = epilogue(r_b + alpha[b]*(f_b - r_b)):  */
                        /* the Dense layer applied to real + alpha*(fake-real) (tf_utils.py:162-167) */
                        /* by linearity, without a third pass over B.  Needs ws >= M*N floats. */
  const float *col_scale; /* [N] or NULL (then col_shift NULL too): optional per-feature affine stage    */
  const float *col_shift; /* after the bias, before the activation: v = v*col_scale[n] + col_shift[n] -    */
                        /* batch normalisation with folded statistics (agx_bn_fold).  The reference has */
                        /* no batch-norm (tf_utils.py:85-88): NULL on the parity path.                  */
  uint32_t *sk_flags;   /* NULL, or >= agx_gemm_sk_flag_words() device words, ZEROED ONCE by the caller and  */
  int32_t sk_flag_words;/* then left to the library: hand-off words of the stream-K form of a plain TN       */
                        /* weight-gradient product (tile count that quantises badly over the CUs: the        */
                        /* (tile, K-tile) space is dealt evenly over 2 x CUs workgroups, partial tiles meet  */
                        /* in ws in a fixed order - bit-reproducible).  Word [2 x CUs] is a spin-timeout     */
                        /* mark (stays 0).  Launches that share the words must not overlap.  NULL: never.    */
} agx_gemm_t;
int agx_gemm_f32(void *stream, const agx_gemm_t *p);
/* Upper bound on the split-K / stream-K scratch (floats) the heuristic may ask for. */
int64_t agx_gemm_ws_floats(int32_t M, int32_t N, int32_t K);
int64_t agx_gemm_sk_flag_words(void);

/* Optional batch-norm stage of a Dense layer (BASELINE.json north_star; the reference itself has none,
 * tf_utils.py:85-88, so nothing on the parity path calls these).
 *   inference / folded:  scale = gamma / sqrt(var + eps), shift = beta - mean * scale  (agx_bn_fold), then
 *                        ONE launch: agx_gemm_f32 with bias, col_scale / col_shift and the activation;
 *   batch statistics:    agx_gemm_f32 (bias only) -> agx_colstats over the batch rows (mean, population
 *                        std) -> agx_bn_fold(var = std^2) -> agx_affine_act in place.
 * var_or_std: is_std = 0 -> variances, 1 -> standard deviations (what agx_colstats returns).        */
int agx_bn_fold(void *stream, const float *mean, const float *var_or_std, int32_t is_std, const float *gamma,
                const float *beta, float eps, float *scale, float *shift, int32_t N);
/* y[m,n] = act(y[m,n]*scale[n] + shift[n]) in place; act: leaky-ReLU with `slope` (0 = ReLU), act = 0: none */
int agx_affine_act(void *stream, float *y, int32_t ldy, int32_t M, int32_t N, const float *scale,
                   const float *shift, int32_t act, float slope);
/* Backward of y = gamma * (x - mean) / sqrt(std^2 + eps) + beta over the M batch rows (batch statistics:
 * mean / std are functions of x), in place: on entry dz = dL/dy [M, N] (ld lddz), on return dL/dx:
 *   xhat = (x - mean) / sqrt(std^2 + eps);  dbeta = sum_m dz;  dgamma = sum_m dz * xhat;
 *   dx = gamma / sqrt(std^2 + eps) * (dz - dbeta / M - xhat * dgamma / M)
 * x: the layer's pre-normalisation values [M, N] (ld ldx); mean / std: what agx_colstats returned for them.
 * Column sums in a fixed order (deterministic).  No reference counterpart (tf_utils.py:85-88 has no BN). */
int agx_bn_backward(void *stream, float *dz, int32_t lddz, const float *x, int32_t ldx, const float *mean,
                    const float *std, const float *gamma, float eps, int32_t M, int32_t N, float *dgamma,
                    float *dbeta);
/* Keras moving statistics: mm = momentum * mm + (1 - momentum) * mean, mv = momentum * mv + (1 - momentum) * std^2
 * (biased batch variance: the non-fused path Keras takes for 2-D inputs) */
int agx_bn_moving_update(void *stream, float *moving_mean, float *moving_var, const float *mean, const float *std,
                         float momentum, int32_t N);

/* ---- embeddings: tf_utils.py:64-76, 120-131 ---------------------------- */
typedef struct agx_embed {
  int32_t n_cat;
  int32_t vocab[AGX_MAX_CAT];
  int32_t edim[AGX_MAX_CAT];
  int64_t table_off[AGX_MAX_CAT]; /* float offset of table i inside `tables` */
} agx_embed_t;
/* out[b,:] = [ lead[s, 0:n_lead] | num[s, 0:n_num] | emb_0[cat[s,0]] | emb_1[cat[s,1]] ... ]
 * with source row s = b % row_mod (row_mod <= 0: s = b), so that one launch can fill the
 * covariate columns of the stacked [real; fake; interpolated] critic batch.  Indices outside
 * [0, vocab) give a zero row (TF GPU Embedding semantics).                                   */
int agx_embed_concat_fwd(void *stream, const agx_embed_t *e, const float *tables,
                         const float *lead, int32_t n_lead, int32_t ld_lead,
                         const float *num, int32_t n_num, int32_t ld_num,
                         const int32_t *cat, int32_t ld_cat, int32_t B, int32_t row_mod,
                         float *out, int32_t ldo);
/* grad_tables[i][v,:] (+)= sum_{b: cat[b % row_mod, i]==v} dout[b, col0 + off_i : +edim_i]
 * (duplicates summed in a fixed order; deterministic).  accumulate=0 overwrites. */
int agx_embed_bwd(void *stream, const agx_embed_t *e, float *grad_tables,
                  const float *dout, int32_t ldd, int32_t col0,
                  const int32_t *cat, int32_t ld_cat, int32_t B, int32_t row_mod,
                  int32_t accumulate);

/* Stacked critic input (tf_utils.py:131 concat + :209-210/:247 augmentation) in one launch:
 *   X[dst_row0 + b, 0:G]       = src[b, 0:G] (+ aug_mask[b] * aug_noise[b * aug_ldn + g])   b < src_rows, if src
 *   X[r, G : G+n_num+E]        = [ num[r % B] | emb_0[cat[r % B, 0]] | ... ]                r < rows_cov
 * (also builds the generator's layer-1 input [z | num | emb] for several stacked z batches: G = z_dim) */
int agx_stack_rows(void *stream, const agx_embed_t *e, const float *tables, const float *src,
                   int32_t lds, int32_t dst_row0, int32_t src_rows, const float *num, int32_t n_num, int32_t ld_num,
                   const int32_t *cat, int32_t ld_cat, int32_t B, int32_t rows_cov, int32_t G, float *X,
                   int32_t ldx, const float *aug_mask, const float *aug_noise, int32_t aug_ldn);

/* ---- elementwise / reductions of the WGAN-GP loss: tf_utils.py:140-181 -- */
/* inter = real + alpha[b] * (fake - real)            (tf_utils.py:162-164) */
int agx_gp_interp(void *stream, const float *real, int32_t ldr, const float *fake, int32_t ldf,
                  const float *alpha, float *inter, int32_t ldi, int32_t B, int32_t G);
/* x[b,:] += mask[b] * noise[(row_bcast ? 0 : b), :]  (tf_utils.py:209-210, 247) */
int agx_aug_add(void *stream, float *x, int32_t ldx, const float *mask, const float *noise,
                int32_t ldn, int32_t row_bcast, int32_t B, int32_t G);
/* out[r] = A[r,:] . w + b[0]  (the critic's Dense(1), tf_utils.py:135) */
int agx_rowdot(void *stream, const float *A, int32_t lda, const float *w, const float *b,
               float *out, int32_t R, int32_t H);
/* delta[r,j] = act'(A[r,j]) * c(r) * w[j];  c = c0 for r<n0, c1 for n0<=r<n0+n1, c2 after
 * (seed of the hand-derived backward; c = -1/B real, +1/B fake, 1 for the GP rows) */
int agx_delta_init(void *stream, const float *A, int32_t lda, const float *w, float *delta,
                   int32_t ldd, int32_t R, int32_t H, int32_t n0, int32_t n1,
                   float c0, float c1, float c2, float slope);
/* agx_rowdot + agx_delta_init in one launch (one wave per row). */
int agx_critic_head(void *stream, const float *A, int32_t lda, const float *w, const float *b,
                    float *out, float *delta, int32_t ldd, int32_t R, int32_t H, int32_t n0,
                    int32_t n1, float c0, float c1, float c2, float slope);
/* Gradient-penalty finalisation from the per-slab row sums of squares a GEMM left in `rowsq`
 * (agx_gemm_t.rowsq): slopes[b] = sqrt(sum) (no epsilon, tf_utils.py:169), fscale[b] =
 * coef*(s-1)/s, d1[b,:] *= fscale[b] (so that dW_1 = X^T.D_1 carries the penalty term with the
 * UNSCALED input gradient left in X), xcov[b, 0:ncov] = 0.  Fixed summation order.          */
int agx_gp_finalize(void *stream, const float *rowsq, int32_t rowsq_ld, int32_t nslab,
                    float *slopes, float *fscale, float *d1, int32_t ldd, int32_t H1, float *xcov,
                    int32_t ldx, int32_t ncov, int32_t B, float coef);
/* GP slopes: s[b] = sqrt(sum_g g[b,g]^2) (NO epsilon, tf_utils.py:169);
 * then g[b,:] *= coef * (s-1)/s  (d gp / d g with coef = weight*2/B_global).   */
int agx_gp_rownorm_scale(void *stream, float *g, int32_t ldg, float *slopes, int32_t B,
                         int32_t G, float coef);
/* losses[0] = -sum(out[0:n0])*inv_b + sum(out[n0:n0+n1])*inv_b   (tf_utils.py:174-181)
 * losses[1] = sum((slopes-1)^2)*inv_b                            (tf_utils.py:170)
 * losses[2] = losses[0] + gp_weight*losses[1]                    (tf_utils.py:217-218)
 * n0 = 0 gives the generator loss mean(-D(fake)) with sign=-1    (tf_utils.py:147-151) */
int agx_wgan_loss(void *stream, const float *out, int32_t n0, int32_t n1, const float *slopes,
                  int32_t n_slopes, float inv_b, float gp_weight, float fake_sign, float *losses);
/* out[j] = sum_{r0<=r<r1} A[r,j]  (BiasAddGrad; deterministic) ; accumulate=1 adds */
int agx_colsum(void *stream, const float *A, int32_t lda, int32_t r0, int32_t r1, int32_t H,
               float *out, int32_t accumulate);
/* out[j] = sum_r c(r) * A[r,j], c as in agx_delta_init; out[H] (if bias_out) = sum_r<n0+n1 c(r) */
int agx_colsum_weighted(void *stream, const float *A, int32_t lda, int32_t R, int32_t H,
                        int32_t n0, int32_t n1, float c0, float c1, float c2, float *out,
                        float *bias_out);

/* Every small gradient piece of one step in ONE launch (same arithmetic as agx_colsum /
 * agx_colsum_weighted / agx_wgan_loss, with its own fixed - deterministic - summation order:
 * four interleaved partial sums per column; the embedding job replaces the
 * (delta . W_cov^T) product + agx_embed_bwd pair):
 *   cs_*   : n_colsum BiasAddGrad jobs  out[j] = sum_{r0<=r<r1} A[r,j]
 *   w_*    : Dense(1) gradient          out[j] = sum_r c(r) A[r,j], bias_out = n0*c0 + n1*c1  (w_A NULL: skip)
 *   emb_*  : dE_i[v,d] = (sum_{r<emb_rows: cat[r % emb_B, i]==v} D[r,:]) . W[emb_col0+off_i+d, :]
 *            (W = first-layer kernel [in,out], so the row is contiguous)               (emb_grad NULL: skip)
 *   l_*    : loss scalars as agx_wgan_loss                                             (l_losses NULL: skip) */
typedef struct agx_small_grads {
  int32_t n_colsum;
  const float *cs_A[AGX_MAX_LAYERS];
  int32_t cs_lda[AGX_MAX_LAYERS], cs_r0[AGX_MAX_LAYERS], cs_r1[AGX_MAX_LAYERS], cs_H[AGX_MAX_LAYERS];
  float *cs_out[AGX_MAX_LAYERS];
  const float *w_A;
  int32_t w_lda, w_R, w_H, w_n0, w_n1;
  float w_c0, w_c1, w_c2;
  float *w_out, *w_bias_out;
  float *emb_grad;
  agx_embed_t emb;
  const float *emb_D, *emb_W;
  const int32_t *emb_cat;
  int32_t emb_ldd, emb_H, emb_ldw, emb_col0, emb_ld_cat, emb_B, emb_rows;
  float *l_losses;
  const float *l_out, *l_slopes;
  int32_t l_n0, l_n1, l_ns;
  float l_inv_b, l_gp_weight, l_fake_sign;
  /* the critic output of row i may arrive as l_out_parts (0 or 1: a plain vector) partial sums
   * l_out[s * l_out_ld + i], added in part order (column-split critic tail launches) */
  int32_t l_out_parts, l_out_ld;
} agx_small_grads_t;
int agx_small_grads(void *stream, const agx_small_grads_t *jobs);

/* ---- optimizers: tf_utils.py:221, 256, 285-288 (Keras-v2 semantics) ---- */
/* rms = rho*rms + (1-rho) g^2 ; w -= lr * g / (sqrt(rms) + eps) */
int agx_opt_rmsprop(void *stream, float *w, const float *g, float *rms, int64_t n, float lr,
                    float rho, float eps, float grad_scale);
/* m,v EMA ; w -= lr_t * m / (sqrt(v) + eps), lr_t computed by the caller */
int agx_opt_adam(void *stream, float *w, const float *g, float *m, float *v, int64_t n,
                 float lr_t, float beta1, float beta2, float eps, float grad_scale);

/* ---- RNG (replaces tf.random.normal/uniform, tf_utils.py:162,317,322,372) */
/* Philox4x32-10; the stream is a deterministic function of (seed, offset, i); callers
 * advance `offset` by the (4-rounded) number of values drawn. */
int agx_rng_normal(void *stream, float *out, int64_t n, uint64_t seed, uint64_t offset);
int agx_rng_uniform(void *stream, float *out, int64_t n, uint64_t seed, uint64_t offset);
/* Same streams with the position read from device memory: offset = *base_dev + sub_offset
 * (lets a captured graph advance its own RNG position between replays). */
int agx_rng_at(void *stream, float *out, int64_t n, uint64_t seed, const uint64_t *base_dev,
               uint64_t sub_offset, int32_t normal);
int agx_set_u64(void *stream, uint64_t *dst, uint64_t value);
/* Several segments of those streams in ONE launch (all the z / alpha draws of a loop iteration,
 * tf_utils.py:316-322): segment i fills out[0..n) from position *base_dev + sub_offset. */
#define AGX_MAX_RNG_SEGS 16
typedef struct agx_rng_seg {
  float *out;
  int64_t n;
  uint64_t sub_offset;  /* multiple of 4 */
  int32_t normal; /* 1: N(0,1), 0: U[0,1) */
  int64_t skip;   /* out[j] = element skip + j of the stream at that position (a rank's rows */
                  /* of a global-batch draw)                                                */
} agx_rng_seg_t;
int agx_rng_multi(void *stream, uint64_t seed, const uint64_t *base_dev, int32_t n_seg,
                  const agx_rng_seg_t *segs_host);

/* ---- correlation path: utils.py:260-268, 808-873 ------------------------ */
/* per-column mean and population std (ddof=0) of x[N,G]  (utils.py:816-818, 264-267).  Two
 * deterministic stages (per-row-chunk fp64 partial sums about the column's first value, then a
 * fixed-order add: no atomics, bit-reproducible, well conditioned for |mean| >> std).
 * ws: agx_colstats_ws_doubles(N, G) doubles of scratch; nothing in it has to be initialised. */
int64_t agx_colstats_ws_doubles(int32_t N, int32_t G);
int agx_colstats(void *stream, const float *x, int32_t ldx, int32_t N, int32_t G, float *mean,
                 float *std, double *ws, int64_t ws_doubles);
/* agx_colstats without a workspace argument and without any allocation.  NO COUNTERPART IN THE
 * REFERENCE: it exists for callers that cannot allocate device memory (code under HIP graph
 * capture, tight serving loops, harnesses that forbid hipMalloc).  The per-chunk partial sums live
 * in a static device array of AGX_COLSTATS_NOALLOC_DOUBLES doubles (2 MiB per GPU, held until the
 * process ends); G <= AGX_COLSTATS_NOALLOC_MAX_G, larger G is rejected with AGX_E_INVALID.  Same
 * kernels and chunking as agx_colstats => bit-identical mean / std.  Calls that overlap on one GPU
 * (different streams or threads) share that array and RACE: serialise them.  agx_init(device)     *
 * resolves the array's address, so a call on an initialised device makes no runtime query and may  *
 * be recorded into a graph (a device that was never initialised resolves it at its first call:     *
 * make that call outside a capture).                                                               */
#define AGX_COLSTATS_NOALLOC_MAX_G 65536
#define AGX_COLSTATS_NOALLOC_DOUBLES 262144
int agx_colstats_noalloc(void *stream, const float *x, int32_t ldx, int32_t N, int32_t G,
                         float *mean, float *std);
/* out = (x - mean) / std  (utils.py:268, 819) */
int agx_standardize(void *stream, const float *x, int32_t ldx, int32_t N, int32_t G,
                    const float *mean, const float *std, float *out, int32_t ldo);
/* C[G,G] = alpha * Xs^T Xs exploiting symmetry (utils.py:824 with x is y); tile rows
 * [tile_begin, tile_end) of the upper-triangular 128x128 tile list are computed, so
 * that the list can be dealt across GPUs; pass 0, -1 for all tiles.    */
int agx_corr_xtx(void *stream, const float *xs, int32_t ldx, int32_t N, int32_t G, float alpha,
                 float *C, int32_t ldc, int64_t tile_begin, int64_t tile_end);
int64_t agx_corr_num_tiles(int32_t G);
/* Condensed strict upper triangle, row-major order, NaNs dropped (utils.py:837-849).
 * row_counts/row_offsets: G+1 int64 scratch each; *n_out_dev receives the length.   */
int agx_triu_pack(void *stream, const float *C, int32_t ldc, int32_t G, float *out,
                  int64_t *row_counts, int64_t *row_offsets, int64_t *n_out_dev);
/* Same for a float64 matrix (pure data movement: bit exact for either dtype). */
int agx_triu_pack_f64(void *stream, const double *C, int32_t ldc, int32_t G, double *out,
                      int64_t *row_counts, int64_t *row_offsets, int64_t *n_out_dev);
/* sums[0..4] (fp64) += sum a, sum b, sum a^2, sum b^2, sum ab over two fp32 vectors
 * transformed as (one_minus ? 1-v : v); the 1-D Pearson of utils.py:872 follows on host.
 * ws (here and in the two calls below): agx_sums_ws_doubles() doubles of scratch for the
 * workgroups' partial sums, added in index order by a finishing launch (deterministic). */
int64_t agx_sums_ws_doubles(void);
int agx_pearson_sums(void *stream, const float *a, const float *b, int64_t n, int32_t one_minus,
                     double *sums, double *ws);
/* Same five sums taken directly over the strict upper triangles of two G x G matrices
 * (1 - c), plus sums[5] = number of NaN entries met (then the caller must use the
 * packed path to reproduce the reference's NaN-dropping).  utils.py:863-873 fused. */
int agx_gamma_sums(void *stream, const float *C1, int32_t ld1, const float *C2, int32_t ld2,
                   int32_t G, double *sums, double *ws);
/* The same six sums over tiles [tile_begin, tile_end) of agx_corr_xtx's upper-triangular 128x128
 * tile list only (the share of one rank when the matrix is dealt across GPUs; the ranks' sums
 * add up to agx_gamma_sums' and are combined with one all-reduce of 6 doubles). */
int agx_gamma_sums_tiles(void *stream, const float *C1, int32_t ld1, const float *C2, int32_t ld2,
                         int32_t G, int64_t tile_begin, int64_t tile_end, double *sums, double *ws);

/* float64 inputs: the reference computes in the dtype it is given (utils.py:816-824 on float64 arrays, the
 * E. coli notebook path), so float64 matrices take a float64 path end to end - same two-stage statistics,
 * X^T Y / N on the fp64 matrix cores (v_mfma_f64_16x16x4_f64), fp64 sums for gamma.  Sized for the shapes
 * float64 callers have (N <= ~1000 samples, G <= ~4300 genes); the fp32 path above is the throughput path. */
int agx_colstats_f64(void *stream, const double *x, int32_t ldx, int32_t N, int32_t G, double *mean,
                     double *std, double *ws, int64_t ws_doubles);
int agx_standardize_f64(void *stream, const double *x, int32_t ldx, int32_t N, int32_t G,
                        const double *mean, const double *std, double *out, int32_t ldo);
/* C[G1, G2] = alpha * xs^T ys  (utils.py:824; pass the same pointer twice for the self-correlation) */
int agx_corr_xty_f64(void *stream, const double *xs, int32_t ldx, const double *ys, int32_t ldy, int32_t N,
                     int32_t G1, int32_t G2, double alpha, double *C, int32_t ldc);
int agx_pearson_sums_f64(void *stream, const double *a, const double *b, int64_t n, int32_t one_minus,
                         double *sums, double *ws);
int agx_gamma_sums_f64(void *stream, const double *C1, int32_t ld1, const double *C2, int32_t ld2, int32_t G,
                       double *sums, double *ws);

/* ---- whole train step: tf_utils.py:186-258 ------------------------------ */
typedef struct agx_mlp_desc {
  int32_t n_layers;                 /* Dense layers incl. the output layer      */
  int32_t dims[AGX_MAX_LAYERS + 1]; /* dims[0] = layer-1 input width            */
  int32_t ldw[AGX_MAX_LAYERS];      /* leading dim of kernel l ([in,out] layout) */
  int64_t w_off[AGX_MAX_LAYERS];    /* float offsets into the flat buffer       */
  int64_t b_off[AGX_MAX_LAYERS];
  agx_embed_t emb;                  /* table_off are flat offsets too           */
  int32_t n_num;
  int64_t n_flat;                   /* flat length incl. padding                */
  /* optional BatchNormalization after every hidden Dense, before the activation (north_star; the reference has   */
  /* none, tf_utils.py:85-88: bn = 0 on every parity path).  GENERATOR only (the critic's gradient penalty is a     */
  /* per-sample quantity: batch statistics would couple the samples) and eager step functions only.  Per hidden   */
  /* layer l four vectors of dims[l+1] floats in the flat buffer: gamma, beta (trained) and the moving mean /      */
  /* variance (their gradient entries stay 0, so the optimizer leaves them alone; updated by the training-mode    */
  /* forward).  Keras semantics: train_gen runs the generator with training=True (batch statistics, :244),        */
  /* train_disc and predict with training=False (moving statistics, :206, :372-375).                              */
  int32_t bn;
  float bn_eps, bn_momentum;        /* Keras defaults 1e-3, 0.99 */
  int64_t bn_gamma_off[AGX_MAX_LAYERS], bn_beta_off[AGX_MAX_LAYERS], bn_mean_off[AGX_MAX_LAYERS], bn_var_off[AGX_MAX_LAYERS];
} agx_mlp_desc_t;

typedef struct agx_cgan_config {
  int32_t x_dim, z_dim;
  agx_mlp_desc_t gen, disc;
  float slope;                      /* 0 = ReLU (reference)                     */
  int32_t max_batch;
  int32_t max_steps;                /* nb_critic + 1 of the captured iteration (agx_cgan_iter_*): the generator */
                                    /* runs ONCE per iteration on max_steps stacked z batches; 0 = 6          */
} agx_cgan_config_t;

typedef struct agx_cgan agx_cgan_t; /* opaque; host memory only */
int agx_cgan_create(const agx_cgan_config_t *cfg, agx_cgan_t **out);
int agx_cgan_destroy(agx_cgan_t *h);
int64_t agx_cgan_workspace_bytes(const agx_cgan_t *h);
int agx_cgan_bind(agx_cgan_t *h, float *gen_params, float *gen_grads, float *disc_params,
                  float *disc_grads, void *workspace, int64_t workspace_bytes);

/* Generator forward (tf_utils.py:359-376 predict; :206): out[B, x_dim]. */
int agx_cgan_gen_forward(agx_cgan_t *h, void *stream, const float *z, const int32_t *cat,
                         const float *num, int32_t B, float *out, int32_t ldo);
/* Critic forward on given x: scores[B] (diagnostics / tests). */
int agx_cgan_disc_forward(agx_cgan_t *h, void *stream, const float *x, int32_t ldx,
                          const int32_t *cat, const float *num, int32_t B, float *scores);

typedef struct agx_step_args {
  const float *x;      /* real batch [B, x_dim] (critic step only)              */
  int32_t ldx;
  const float *z;      /* [B, z_dim]                                             */
  const int32_t *cat;  /* [B, n_cat]                                             */
  const float *num;    /* [B, n_num]                                             */
  const float *alpha;  /* [B] GP interpolation coefficients (critic step only)   */
  int32_t B;           /* local batch                                            */
  int32_t B_global;    /* divisor of the means (== B on one GPU)                 */
  float gp_weight;     /* 10 (tf_utils.py:187)                                   */
  const float *aug_mask;       /* [B] or NULL   (tf_utils.py:202, 240)           */
  const float *aug_noise_fake; /* critic: [x_dim] broadcast row (:209); gen: [B,x_dim] (:247) */
  const float *aug_noise_real; /* critic: [B, x_dim] (:210)                      */
  float *losses;       /* device [4]: wgan term, gp, total, unused               */
} agx_step_args_t;
/* Gradients of one critic step into disc_grads (tf_utils.py:204-220); no optimizer. */
int agx_cgan_disc_grads(agx_cgan_t *h, void *stream, const agx_step_args_t *a);
/* Gradients of one generator step into gen_grads (tf_utils.py:242-255). */
int agx_cgan_gen_grads(agx_cgan_t *h, void *stream, const agx_step_args_t *a);

/* ---- whole loop iteration as ONE hipGraph: tf_utils.py:316-323 --------------
 * nb_critic x (z ~ N(0,1), alpha ~ U[0,1), critic gradients, RMSprop) + (z, generator
 * gradients, RMSprop), captured once per batch size and replayed; removes the per-kernel
 * launch cost that dominates at batch 32 (the reference pays 6 tf.function calls + 6 eager RNG
 * calls per 32 samples).  The draws are the SAME Philox stream positions the eager entry
 * points would use (z then alpha per critic step, 4-aligned), so graph and eager runs with
 * equal seeds give bit-identical weights.  Single GPU only (no collective inside).        */
typedef struct agx_iter_args {
  int32_t B;                /* rows of this batch (<= max_batch)                         */
  int32_t nb_critic;        /* 5 (tf_utils.py:262)                                       */
  float gp_weight;          /* 10                                                        */
  float lr_disc, lr_gen, rho, eps; /* Keras RMSprop(5e-4, 0.9, 1e-7) for both nets       */
  float *disc_rms, *gen_rms;       /* optimizer slots (flat, same layout as the params)  */
  uint64_t seed;            /* Philox seed                                               */
  const float *aug_mask_disc, *aug_noise_fake, *aug_noise_real; /* NULL or trace constants */
  const float *aug_mask_gen, *aug_noise_gen;
  /* data parallel (all 0 / NULL on one GPU): this rank holds rows [row_lo, row_lo+B) of a global  */
  /* batch of B_global rows; draws are the global batch's, means divide by B_global; split = 1      */
  /* captures one graph per segment between gradient all-reduces (agx_cgan_iter_launch_seg), the    */
  /* optimizer launch at the head of the next segment; split = 2 captures nb_critic + 1 segments    */
  /* WITHOUT optimizer launches (the caller reduce-scatters the gradients, updates its 1/R shard of */
  /* the parameters and all-gathers them between segments; no optimizer slots needed here);         */
  /* disc_losses / gen_losses (device [4]) redirect the loss scalars, e.g. into slots that ride the */
  /* gradient all-reduce.  split = 3: as 2, and the LARGEST weight-gradient product of each step      */
  /* (critic W_1, generator W_out) is left out of its segment: agx_cgan_dp_iteration launches it in  */
  /* row chunks and exchanges chunk c on a second stream while chunk c+1 multiplies.                 */
  int32_t B_global, row_lo, split;
  float *disc_losses, *gen_losses;
  /* optimizer of the captured iteration: 0 = RMSprop (fields above), 1 = Keras-v2 Adam on both nets   */
  /* (tf_utils.py:261-262 takes optimizer objects).  Adam: slots m / v per net, eps above; the          */
  /* bias-corrected step sizes lr_t = lr*sqrt(1-b2^t)/(1-b1^t) change every step, so the HOST supplies  */
  /* the nb_critic + 1 values of the coming iteration (critic steps, then the generator step):          */
  /* agx_cgan_iter_set_lr before a replay, adam_lr_t_host for agx_cgan_iter_eager.                      */
  int32_t opt_kind;
  float beta1, beta2;
  float *disc_m, *disc_v, *gen_m, *gen_v;
  const float *adam_lr_t_host;
} agx_iter_args_t;
#define AGX_MAX_ITER_SEGS 66
typedef struct agx_iter_graph agx_iter_graph_t;
int agx_cgan_iter_capture(agx_cgan_t *h, void *stream, const agx_iter_args_t *a,
                          agx_iter_graph_t **out);
/* Stage the batch (device pointers, B rows) and replay.  rng_offset = stream position of the
 * first draw; the iteration consumes agx_cgan_iter_rng_draws() positions.
 * losses_out (device, [nb_critic+1][4]) receives [wgan, gp, total, 0] per step.            */
int agx_cgan_iter_launch(agx_iter_graph_t *g, void *stream, const float *x, int32_t ldx,
                         const int32_t *cat, const float *num, uint64_t rng_offset,
                         float *losses_out);
/* Split form for data-parallel runs: stage once, then launch segments 0 .. num_segs-1 with the
 * exchange of the critic (after segments 0..nb_critic-1) or generator (after segment nb_critic)
 * gradient buffer in between (RCCL, issued by the caller on the same stream): a sum all-reduce
 * (split = 1) or reduce-scatter + sharded update + parameter all-gather (split = 2). */
int agx_cgan_iter_stage(agx_iter_graph_t *g, void *stream, const float *x, int32_t ldx,
                        const int32_t *cat, const float *num, uint64_t rng_offset);
int32_t agx_cgan_iter_num_segs(const agx_iter_graph_t *g);
int agx_cgan_iter_launch_seg(agx_iter_graph_t *g, void *stream, int32_t seg);
int64_t agx_cgan_iter_rng_draws(const agx_iter_graph_t *g);
/* Adam only: the n = nb_critic + 1 step sizes the next agx_cgan_iter_launch / _stage will put in device memory. */
int agx_cgan_iter_set_lr(agx_iter_graph_t *g, const float *lr_t_host, int32_t n);
/* The same launch sequence as the captured one-GPU iteration (fused epilogues included), queued
 * directly on `stream` instead of replayed: what agx_prof_* brackets (events cannot be captured) and
 * debuggers see.  *draws_out (host,
 * optional) receives the number of RNG stream positions consumed. */
int agx_cgan_iter_eager(agx_cgan_t *h, void *stream, const agx_iter_args_t *a, const float *x,
                        int32_t ldx, const int32_t *cat, const float *num, uint64_t rng_offset,
                        float *losses_out, int64_t *draws_out);
int agx_cgan_iter_destroy(agx_iter_graph_t *g);

/* ---- RCCL over xGMI for hosts without PyTorch (no reference counterpart: the reference ----
 *      is one process on one GPU, tf_utils.py:17-38).  One process per GPU; librccl is bound with
 *      dlopen at the first call (the copy already in the process, e.g. PyTorch's, is reused).
 *      Rank 0 calls agx_comm_unique_id and hands the 128 bytes to the other ranks by any host
 *      channel; every rank then calls agx_comm_init with the current HIP device selected.  The
 *      collectives are queued on `stream` like every other entry point (no host synchronisation):
 *        data-parallel step:   agx_cgan_disc_grads -> agx_comm_reduce_scatter_sum(flat gradients)
 *                              -> agx_opt_rmsprop on the own 1/R shard -> agx_comm_all_gather(parameters)
 *        tiled Gamma:          agx_gamma_sums_tiles on the own tiles -> agx_comm_allreduce_sum(6 doubles) */
typedef struct agx_comm agx_comm_t;
int agx_comm_unique_id(void *id128_host);
int agx_comm_init(const void *id128_host, int32_t nranks, int32_t rank, agx_comm_t **out);
int agx_comm_destroy(agx_comm_t *c);
int32_t agx_comm_size(const agx_comm_t *c);
int32_t agx_comm_rank(const agx_comm_t *c);
/* in place sum over ranks of n fp32 (f64 = 0) or fp64 (f64 = 1) values */
int agx_comm_allreduce_sum(agx_comm_t *c, void *stream, void *buf, int64_t n, int32_t f64);
/* out[0..n_per_rank) = sum over ranks of in[rank*n_per_rank ..]; in holds nranks*n_per_rank floats */
int agx_comm_reduce_scatter_sum(agx_comm_t *c, void *stream, const float *in, float *out, int64_t n_per_rank);
/* out[r*n_per_rank ..] = rank r's in[0..n_per_rank) */
int agx_comm_all_gather(agx_comm_t *c, void *stream, const float *in, float *out, int64_t n_per_rank);


/* ---- data-parallel iteration with the exchange overlapped, driven from this library (no reference ----
 *      counterpart: tf_utils.py:316-323 is one process on one GPU).  `g` was captured with split = 3.  Per
 *      optimizer step: segment s on `stream` (everything but the largest weight-gradient product), then that
 *      product in n_pieces row chunks rows[c] .. rows[c+1] (multiples of 4 rows) on `stream`, an event after
 *      each; on `comm_stream`, per chunk as soon as its event has fired: in-place reduce-scatter of piece c of
 *      the flat gradient buffer ([start[c], start[c] + per[c] * nranks): rank r owns [start + r per, + per)),
 *      the optimizer on the own share (slots of other shares are never touched), the loss scalars behind the
 *      gradients copied across, in-place all-gather of the piece of the parameter buffer.  The next segment
 *      waits for the last all-gather.  Pieces must tile [0, total), total >= n_flat + 64 loss slots, every
 *      start and per a multiple of 4 floats, and piece c must be complete once chunk c is (it ends at or
 *      before flat offset w_off + rows[c+1] * ldw of the largest kernel; the LAST piece holds everything
 *      behind it).  RMSprop (opt_kind 0) or Adam (1; lr_t_host = nb_critic + 1 bias-corrected step sizes).
 *      Both buffers of a net must extend to `total` floats.  Nothing is allocated; events are created at the
 *      first call on a graph.  With one rank every collective is the identity (testing).                  */
#define AGX_MAX_DP_PIECES 16
typedef struct agx_dp_net {
  int32_t n_pieces;
  int32_t rows[AGX_MAX_DP_PIECES + 1];
  int64_t start[AGX_MAX_DP_PIECES];
  int64_t per[AGX_MAX_DP_PIECES];
  float *slot0, *slot1;     /* RMSprop: rms, NULL; Adam: m, v (flat, parameter layout) */
  float lr;                 /* RMSprop learning rate                                    */
} agx_dp_net_t;
typedef struct agx_dp_args {
  agx_dp_net_t disc, gen;
  int32_t opt_kind;
  float rho, beta1, beta2, eps;
  const float *lr_t_host;
} agx_dp_args_t;
int agx_cgan_dp_iteration(agx_iter_graph_t *g, void *stream, void *comm_stream, agx_comm_t *comm,
                          const agx_dp_args_t *a, const float *x, int32_t ldx, const int32_t *cat,
                          const float *num, uint64_t rng_offset);

#ifdef __cplusplus
}
#endif
#endif /* AGX_H */
