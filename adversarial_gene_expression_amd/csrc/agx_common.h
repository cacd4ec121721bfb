// Internal helpers shared by the libagx translation units (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstdint>
#include "../../include/agx.h"

namespace agx {

void set_error(const char *fmt, ...);
int num_cus();
int gemm_prepare_device();

// Optimizer update fused into a weight-gradient GEMM epilogue (internal: captured single-GPU
// iteration only).  w / s1 point at the element that corresponds to C[0,0] (same leading
// dimension); rows [0, rows) of the result are consumed by the RMSprop update instead of stored.
struct FusedOpt {
  float *w, *s1;
  int rows;
  float lr, rho, eps;
  // optional: the same launch also runs the flat RMSprop over the rest of the network's parameter
  // buffer - [0, rest_n) minus [rest_hole_lo, rest_hole_hi) - in workgroups behind the product's tiles.
  // The gradients of that range must be complete BEFORE the launch (earlier launches on the stream).
  float *rest_w = nullptr, *rest_s1 = nullptr;
  const float *rest_g = nullptr;
  int64_t rest_n = 0, rest_hole_lo = 0, rest_hole_hi = 0;
  // optional (needs rest_n > 0): the same launch also prepares the NEXT step's stacked critic input in
  // another buffer - copy workgroups behind the update workgroups move nx_rows generated rows (G columns)
  // from nx_src to nx_dst, and update workgroup 0, which owns the embedding tables (flat offsets < 1024),
  // writes the covariate columns [G, G+ncov) of rows [0, nx_cov_rows) of nx_cov from the tables it has just
  // updated: [num[r % nx_B] | emb_0[cat[r % nx_B, 0]] | ...] like agx_stack_rows
  const float *nx_src = nullptr;
  float *nx_dst = nullptr, *nx_cov = nullptr;
  int nx_lds = 0, nx_ldd = 0, nx_rows = 0, nx_G = 0, nx_cov_rows = 0, nx_B = 0, nx_nnum = 0, nx_ldcat = 0;
  const float *nx_num = nullptr;
  const int32_t *nx_cat = nullptr;
  agx_embed_t nx_emb = {};
  // augmentation of the copied rows (tf_utils.py:209-210, :247): dst = src + nx_mask[r] * nx_noise[r * nx_ldn + g]
  const float *nx_mask = nullptr, *nx_noise = nullptr;
  int nx_ldn = 0;
};
// raw_slabs != NULL: no reduce launch; the split-K partial slabs stay in g->ws as [M][splitk][N]
// (also when splitk == 1) and *raw_slabs receives splitk - the caller's kernel finishes the product
// dual: rows [0, dual->rows) of a plain Dense product (alpha, bias, activation; LDS-tiled kernel, no split-K) are ALSO
// stored at dual->C (the generator's first generated batch straight into the stacked critic input: no copy launch).
struct DualOut {
  float *C;
  int ldc, rows;
};
bool gemm_can_dual(int M, int N, int K, int64_t ws_floats, int layout);
int gemm_run(void *stream, const agx_gemm_t *g, const FusedOpt *fo, int *raw_slabs, const DualOut *dual = nullptr);

// Small-batch critic tail (two hidden layers of width <= 256): everything between the split-K slabs
// of the first-layer product and the next big product in ONE launch per phase instead of 3 + 3.
struct TailArgs {
  const float *ws; int splitk; int m_slab;      // slabs [m_slab][splitk][H1]
  int H1, H2;
  const float *W2; int ldw2;
  float slope;
  float *A1; int lda1; float *A2; int lda2;
  // mode 0 (forward): rows_out = interp ? 3*interp_rows : m_slab
  const float *b1, *b2, *w3, *b3, *alpha;
  int interp_rows;
  float *D2; int ldd2; float *out;
  int n0, n1; float c0, c1, c2;
  // mode 1 (gradient-penalty second order on rows [row0, row0 + m_slab) of A1 / A2 / D1)
  int row0;
  const float *rowsq; int rowsq_ld, nslab;
  float *slopes, *fscale, *D1; int ldd1;
  float *xcov; int ldx, ncov;
  float coef;
  int rpw;   // rows per workgroup (set by critic_tail)
  // column split of the hidden Dense over nsplit (1, 2, 4) workgroups per row group (tail.hip): out becomes
  // out[ns * out_ld + row] partial dot products; A1m = copy of the interpolated rows' A1 (written by mode 0 with
  // interp_rows > 0, read by mode 1 as the ReLU mask), [interp_rows, lda1]
  int nsplit, out_ld;
  float *A1m;
  // mode 2 (two hidden Dense layers of the generator from its input rows, no slabs): A1 = act(in . W1 + b1),
  // A2 = act(A1 . W2 + b2) over rows [0, m_slab); K_in <= 128
  const float *in; int ld_in, K_in;
  const float *W1; int ldw1;
  // mode 3 (generator backward behind the output layer's delta slabs): A1 = act'(M1) * sum_z slab_z  (delta of the
  // last hidden layer, width H1), A2 = act'(M2) * (A1 . W2^T)  (delta of the layer before, width H2; W2 is that
  // layer's kernel [H2, H1], ldw2 >= H1); M1 / M2 = the layers' activations (ReLU masks)
  const float *M1; int ldm1; const float *M2; int ldm2;
  // mode 1, Gram form (cgan.hip critic_core): the ONE "slab" is T = d_1 . (W_g^T W_g) of the rows; the squared slope
  // is the row's dot product of T with d_1 (rowsq unused); gU / gF (ld ldg) receive d_1 and f*d_1 for the
  // S = d_1^T diag(f) d_1 product and D1 is left alone
  int gram; float *gU, *gF; int ldg;
  // mode 0 with interp_rows > 0, factored fake rows: the fake rows' layer-1 values come finished from ws2 [interp_rows, H1]
  // (no slabs); ws then holds the real rows' slabs only (m_slab == interp_rows)
  const float *ws2;
};
int critic_tail(void *stream, int mode, const TailArgs &a);
// rows [0, H) of Xt <- W[0:G, 0:H]^T, their covariate columns [G, G + ncov) <- 0 (elementwise.hip; Gram form of the penalty)
int wt_rows(void *stream, const float *W, int ldw, int G, int H, float *Xt, int ldx, int ncov, const float *bvec = nullptr,
            float *bpart = nullptr);
// factored fake rows (cgan.hip `ff`): [W_out; b_out; unit rows of the covariate columns; 0] in front of the stacked input,
// and the per-step [h | 1 | covariates] rows (elementwise.hip)
int ff_stage(void *stream, const float *Wout, int ldw, const float *bout, int HL, int G, int ncov, int he, float *Xhd, int ldx,
             int rows);
int ff_hx(void *stream, const float *hrows, int ldh, int HL, float *X, int ldx, int G, int ncov, int he, float *Hx, int ldhx,
          int B, const float *bpart, int ntile, const float *W1, int ldw, int H1, float *P, int ldp, const agx_embed_t *e,
          const float *tables, const float *num, int n_num, const int32_t *cat);
// true if the heuristic launches the TN product dW[M,N] (K rows) on a kernel that has the fused
// epilogue, without split-K (then fused and unfused runs sum in the same order: bit-identical)
bool gemm_can_fuse_opt(int M, int N, int K, int64_t ws_floats);
int colstats_scratch_init();   // corr.hip: address of the static scratch of agx_colstats_noalloc on the current device
// Loss-log snapshot done by the last small-gradient launch of a captured iteration: n floats from
// src to the address stored at *dst_slot (left there by iter_stage; NULL address: no copy).
struct LossCopy {
  float *const *dst_slot;
  const float *src;
  int n;
};
// Tiny TN weight-gradient products C[M,N] = A[K,M]^T . B[K,N] carried by the small-gradient launch
// (one wave per 32x32 output tile over the whole K, fragments straight from global memory): at batch
// 32 a hidden-layer gradient is a 5 us launch around 2 us of work.
struct TnJobs {
  int n;
  struct Job {
    const float *A, *B;
    float *C;
    int lda, ldb, ldc, M, N, K;
  } job[AGX_MAX_LAYERS];
};
// true if a product of this shape may ride the small-gradient launch (few tiles, short K)
static inline bool tn_job_ok(int M, int N, int K) {
  return K <= 256 && (int64_t)((M + 31) / 32) * ((N + 31) / 32) <= 256;
}
int small_grads_run(void *stream, const agx_small_grads_t *j, const LossCopy &lc, const TnJobs *tn = nullptr);
// Everything a replayed iteration needs staged, in ONE launch: B real rows into the stacked critic
// input, covariates into their staging buffers, RNG position and loss-snapshot address.
// (lrs: per-step Adam step sizes of the iteration about to run, copied to lr_dev[0..n))
struct LrTable {
  int n;
  float v[64];
};
// (X2: optional second stacked-input buffer that receives the same real rows)
struct StageDraws;
int iter_stage(void *stream, const float *x, int ldxs, float *X, float *X2, int ldx, int G, int B, const int32_t *cat,
               int32_t *scat, int n_cat, const float *num, float *snum, int n_num, uint64_t *rngpos,
               uint64_t rng_offset, float **loss_slot, float *losses_out, const LrTable *lrs, float *lr_dev,
               const float *aug_mask = nullptr, const float *aug_noise = nullptr, int aug_ldn = 0,
               const StageDraws *sd = nullptr);
// (aug_mask: the staged real rows get + aug_mask[b] * aug_noise[b * aug_ldn + g], tf_utils.py:210)
// All the draws of a captured iteration AND the generator's layer-1 input in ONE launch: segment i fills
// n values from stream position *base_dev + sub_offset (element skip + j) into out, as rows of `cols` values
// at leading dimension `ld` when cols > 0 (z straight into the [z | num | emb] rows) and flat otherwise;
// one more block row writes the covariate columns [zcols, zcols + ncov) of rows [0, rows_cov) of gin0
// from `tables` ([num | emb_0 | ...] of source row r % B, like agx_stack_rows).
struct DrawSeg {
  float *out;
  int64_t n;
  uint64_t sub_offset;
  int64_t skip;
  int32_t normal, cols, ld;
};
struct IterDraws {
  int n;
  DrawSeg seg[AGX_MAX_RNG_SEGS];
};
// xc (optional): the same launch writes the covariate columns [col0, col0 + ncov) of rows [0, rows) of the stacked
// critic input from the CRITIC's tables (row r takes the covariates of sample r % B)
struct CriticCov {
  const agx_embed_t *e = nullptr;
  const float *tables = nullptr;
  float *X = nullptr;
  int ldx = 0, col0 = 0, rows = 0;
};
int iter_draws(void *stream, uint64_t seed, const uint64_t *base_dev, const IterDraws &d, const agx_embed_t *e,
               const float *tables, const float *num, int n_num, const int32_t *cat, int ld_cat, int B, int rows_cov,
               int zcols, float *gin0, int ldg0, const CriticCov *xc = nullptr);
// The same draws as block rows BEHIND the staging launch of a replayed one-GPU iteration (one launch less per
// iteration): iter_draws_desc validates and packs what iter_draws would launch, iter_stage(..., sd) launches both.
// The draws then take the stream position from the launch argument and the covariates from the caller's arrays
// (the staged copies are written by the same launch).
struct StageDraws {
  int on;                       // 0: nothing packed (iter_stage launches the staging rows only)
  IterDraws d;
  unsigned long long seed;
  agx_embed_t e, e2;
  const float *tables, *tables2;
  int n_num, ld_cat, B, rows_cov, zcols, ldg0, ncov, cov_row, ldx, xcol0, xrows, ncov2;
  float *gin0, *X;
  int gx, rows;                 // grid of the draws part: gx blocks along x, `rows` block rows
};
int iter_draws_desc(uint64_t seed, const IterDraws &d, const agx_embed_t *e, const float *tables, int n_num, int ld_cat,
                    int B, int rows_cov, int zcols, float *gin0, int ldg0, const CriticCov *xc, StageDraws *out);
// Adam with the bias-corrected step size read from device memory (captured iterations)
int opt_adam_dev_lr(void *stream, float *w, const float *g, float *m, float *v, int64_t n, const float *lr_dev,
                    float beta1, float beta2, float eps);
// RMSprop over [0, n) minus the hole [hole_lo, hole_hi) (multiples of 4) that fused epilogues updated
int opt_rmsprop_hole(void *stream, float *w, const float *g, float *rms, int64_t n, int64_t hole_lo,
                     int64_t hole_hi, float lr, float rho, float eps);

// In-stream event bracket around the launches of one kernel class (see agx_prof_* in agx.h).
// Inactive (two loads, no HIP call) unless profiling is enabled for the class.
struct ProfScope {
  ProfScope(hipStream_t st, int id, double flops, double bytes);
  ~ProfScope();
  hipStream_t st_;
  void *slot_;
  hipEvent_t stop_;
};

#define AGX_CHECK_ARG(cond, ...)        \
  do {                                  \
    if (!(cond)) {                      \
      agx::set_error(__VA_ARGS__);      \
      return AGX_E_INVALID;             \
    }                                   \
  } while (0)

#define AGX_HIP(call)                                                              \
  do {                                                                             \
    hipError_t e_ = (call);                                                        \
    if (e_ != hipSuccess) {                                                        \
      agx::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, \
                     __LINE__);                                                    \
      return AGX_E_HIP;                                                            \
    }                                                                              \
  } while (0)

#define AGX_LAUNCH_CHECK() AGX_HIP(hipGetLastError())

#define AGX_TRY(expr)          \
  do {                         \
    int rc_ = (expr);          \
    if (rc_ != AGX_OK) return rc_; \
  } while (0)

static inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }
static inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// decode linear upper-triangular tile index t -> (ti <= tj) over a T x T tile grid
__device__ __forceinline__ void tri_decode(long long t, int T, int &ti, int &tj) {
  // row ti starts at offset ti*T - ti*(ti-1)/2
  double b = 2.0 * T + 1.0;
  int i = (int)((b - sqrt(b * b - 8.0 * (double)t)) * 0.5);
  if (i < 0) i = 0;
  if (i > T - 1) i = T - 1;
  auto start = [T](long long r) { return r * T - r * (r - 1) / 2; };
  while (i > 0 && start(i) > t) --i;
  while (i + 1 < T && start(i + 1) <= t) ++i;
  ti = i;
  tj = i + (int)(t - start(i));
}

// XCD-aware workgroup order.  Workgroups are dealt round-robin over the 8 XCDs (each with its own L2),
// so ids b and b+8 share an L2.  This bijection gives the workgroups of one XCD a CONTIGUOUS range of
// logical ids: neighbours in the logical order (same operand rows / same split-K slice) then hit the
// same L2 instead of fetching the same lines into eight of them.  Speed only, never correctness.
__device__ __forceinline__ int xcd_logical_id(int linear, int total) {
  const int q = total >> 3, r = total & 7, x = linear & 7;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (linear >> 3);
}

// 64-lane wave reductions
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

}  // namespace agx
