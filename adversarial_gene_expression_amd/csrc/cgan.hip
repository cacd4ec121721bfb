// Host-side orchestration of the WGAN-GP conditional-GAN train step
// (reference tf_utils.py:186-258).  Pure launch sequencing on one stream: no
// allocation, no synchronisation, no host<->device copies, so a caller may
// capture a whole iteration into a hipGraph.
//
// The backward pass is hand-derived (no autodiff engine):
//
// * The critic is evaluated ONCE on the stacked batch S = [real; fake; inter]
//   (3B rows).  The first-order chain  D_{l-1} = relu'(A_{l-1}) * (D_l . W_l^T)
//   is seeded with c (x) w_out^T, c = [-1/B | +1/B | 1]: rows 0..2B carry the
//   Wasserstein deltas, rows 2B..3B carry d_l of the gradient penalty
//   (g = d_1 . W_1[:G]^T is the input gradient of tf_utils.py:165-168).
// * With dg = weight*(2/B)*(s-1)/s * g the penalty's second-order chain
//   e_l = relu'(A_l) * (e_{l-1} . W_l), e_0 = dg, is written IN PLACE over the
//   inter rows of the activations (and dg over the inter rows of the input),
//   so that every weight gradient is one stacked product
//       dW_l = [a_{l-1}(real); a_{l-1}(fake); e_{l-1}]^T . [D_l]      (K = 3B)
//   which equals  sum_{real,fake} a^T delta  +  the penalty term e^T d.
//   The penalty contributes nothing to biases, embeddings or the covariate
//   rows of W_1 (ReLU masks are piecewise constant): the covariate columns of
//   the inter rows are zeroed before the dW_1 product and bias gradients sum
//   rows 0..2B only.
// * The first Dense layer is linear, so its pre-activation on the interpolated rows
//   real + alpha*(fake-real) (tf_utils.py:162-167) is r + alpha*(f - r) of the pre-activations of
//   the real and fake rows: layer 1 runs on 2B rows and the split-K reduce emits the third block
//   (agx_gemm_t.interp_rows) - the interpolated INPUT is never materialised.
// * Inside a captured iteration the generator weights are constant over the nb_critic critic
//   steps and the generator step (they change only at the iteration's end), so the generator
//   runs ONCE on the nb_critic+1 stacked z batches; each step picks up its block of fake rows.
// * Batches of at least the hidden widths never form a B x G matrix other than the real rows (gram_ok,
//   gen_fact_ok, iter_ff below): the penalty's input-gradient terms go through the Gram matrix of W_1's gene rows, the
//   generator's linear output layer (tf_utils.py:88) lets W_out W_g stand in for the generated rows.  Same algebra,
//   other rounding order; DESIGN.md section 3 "Large batches".
#include "agx_common.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <new>

struct agx_cgan {
  agx_cgan_config_t cfg;
  bool bound = false;
  float *gp = nullptr, *gg = nullptr, *dp = nullptr, *dg = nullptr;  // flat params / grads
  // workspace views
  int Bm = 0, S = 6, G = 0, Dd = 0, Dg = 0, ldx = 0, ldg0 = 0, ncov = 0;
  float *X = nullptr;                       // [3Bm, ldx] stacked critic input (+cov cols): = Xbuf[0] outside a captured iteration
  float *Xbuf[2] = {nullptr, nullptr};      // two of them: step k's layer-1 product assembles step k+1's rows in the other (chain_input)
  float *A[AGX_MAX_LAYERS + 1] = {};        // critic activations  [3Bm, ldh]
  float *D[AGX_MAX_LAYERS + 1] = {};        // critic deltas       [3Bm, ldh]
  int ldh[AGX_MAX_LAYERS + 1] = {};
  float *out = nullptr, *slopes = nullptr;  // [4][3Bm], [Bm]
  float *A1m = nullptr;                     // [Bm, ldh[1]]
  float *fscale = nullptr, *rowsq = nullptr;  // [Bm], [Bm, nslab]
  int nslab = 0;
  // Gram form of the penalty (critic_core): W_g^T W_g [H1, ldh1]; d_1 . (W_g^T W_g) [Bm, H1] (compact: read as a slab);
  // d_1 and f*d_1 of the interpolated rows [Bm, ldh1] each
  float *gramM = nullptr, *gramT = nullptr, *gramU = nullptr, *gramF = nullptr;
  int dw1_rows = 0;                         // K of the critic's layer-1 weight-gradient product of the step being recorded
  // factored generator step (gen_core): (W_out W_g)^T [H1, ldpt]; d_1^T h_L [H1, ldpt]; column sums of d_1 [H1]
  float *genPT = nullptr, *genQ = nullptr, *genCs = nullptr;
  int ldpt = 0;
  // factored fake rows (record_iteration `ff`): HD = hxp + H1 rows in FRONT of X and of D[1]:
  //   X:    [b_out; unit rows of the covariate columns; 0 (he rows) | W_out (HL rows) | 0-pad to hxp] then W_g^T (H1 rows)
  //   D[1]: [Hx^T d_1(fake)] (hx = he + HL of hxp rows)                                              then S     (H1 rows)
  // ffHx = per-step [1 | covariates | 0 | h] rows [Bm, ldhx]; ffP [HD + Bm, ldp = H1] = the first-layer image of those
  // rows and of the real rows: [b_out W_g; W_c; 0 | W_out W_g | 0 | W_g^T W_g | real rows' pre-activations] - rows he..
  // are ONE product (for GTEx at 512 rows: 256 + 256 + 512 = 1024 rows), the first he rows need none (ff_hx_kernel);
  // ffAf = the fake rows' first-layer values [Bm, H1]; ffBp = per-gene-tile partials of b_out . W_g (wt_rows_kernel)
  int HD = 0, he = 0, hx = 0, hxp = 0, ldhx = 0, ldp = 0, ntile = 0;
  float *ffHx = nullptr, *ffP = nullptr, *ffAf = nullptr, *ffBp = nullptr;
  bool ff = false;                          // set while an iteration with factored fake rows is recorded
  int ff_row0 = 0;                          // row offset of the step's z batch in gh[]
  float *gin0 = nullptr;                    // [S*Bm, ldg0] generator layer-1 input [z | num | emb]
  float *gh[AGX_MAX_LAYERS + 1] = {};       // generator activations [S*Bm, ldgh]
  float *gd[AGX_MAX_LAYERS + 1] = {};       // generator deltas      [Bm, ldgh]
  // generator batch-norm (cfg.gen.bn; eager step functions only): pre-normalisation values of the training-mode
  // forward, batch statistics per hidden layer, folded scale / shift, column-statistics scratch
  float *gpre[AGX_MAX_LAYERS + 1] = {};     // [Bm, ldgh]
  float *bn_mean[AGX_MAX_LAYERS + 1] = {}, *bn_std[AGX_MAX_LAYERS + 1] = {};
  float *bn_scale = nullptr, *bn_shift = nullptr;
  double *bn_ws = nullptr;
  int64_t bn_ws_doubles = 0;
  float *F = nullptr;                       // [S*Bm, ldsx] fake rows of a whole captured iteration
  int ldgh[AGX_MAX_LAYERS + 1] = {};
  // staging for the captured iteration (agx_cgan_iter_*)
  float *snum = nullptr, *zbuf = nullptr, *abuf = nullptr, *losslog = nullptr;
  float *lrdev = nullptr;                   // [64] Adam step sizes of the iteration being replayed
  int32_t *scat = nullptr;
  uint64_t *rngpos = nullptr;
  float **loss_slot = nullptr;              // device word: where the last step snapshots the loss log
  int ldsx = 0;
  float *skws = nullptr;                    // split-K scratch
  int64_t skws_floats = 0;
  uint32_t *skflags = nullptr;              // stream-K hand-off words (zeroed at bind, self-cleaning)
  int skflag_words = 0;
  bool use_sk = false;                      // set while the segments of a data-parallel iteration are recorded
  bool skip_big = false;                    // split == 3: the largest weight-gradient product is launched by agx_cgan_dp_iteration
  int64_t ws_floats_total = 0;
};

namespace {

inline int r4(int x) { return (x + 3) & ~3; }
inline int64_t r64(int64_t x) { return (x + 63) & ~(int64_t)63; }

int validate_mlp(const agx_mlp_desc_t &m, const char *name, int want_in, int want_out) {
  AGX_CHECK_ARG(m.n_layers >= 2 && m.n_layers <= AGX_MAX_LAYERS,
                "%s: n_layers=%d (need >= 1 hidden layer, <= %d layers)", name, m.n_layers,
                AGX_MAX_LAYERS);
  AGX_CHECK_ARG(m.emb.n_cat >= 0 && m.emb.n_cat <= AGX_MAX_CAT, "%s: n_cat=%d", name, m.emb.n_cat);
  int e = 0;
  for (int i = 0; i < m.emb.n_cat; ++i) {
    AGX_CHECK_ARG(m.emb.vocab[i] > 0 && m.emb.edim[i] > 0, "%s: bad embedding %d", name, i);
    AGX_CHECK_ARG(m.emb.table_off[i] >= 0 && m.emb.table_off[i] + (int64_t)m.emb.vocab[i] * m.emb.edim[i] <= m.n_flat,
                  "%s: embedding %d outside the flat buffer", name, i);
    e += m.emb.edim[i];
  }
  AGX_CHECK_ARG(m.dims[0] == want_in + m.n_num + e, "%s: dims[0]=%d != %d+%d+%d", name, m.dims[0],
                want_in, m.n_num, e);
  AGX_CHECK_ARG(m.dims[m.n_layers] == want_out, "%s: output width %d != %d", name,
                m.dims[m.n_layers], want_out);
  if (m.bn) {
    AGX_CHECK_ARG(m.bn_eps > 0.f && m.bn_momentum >= 0.f && m.bn_momentum <= 1.f, "%s: batch-norm eps / momentum", name);
    for (int l = 0; l < m.n_layers - 1; ++l)
      for (const int64_t off : {m.bn_gamma_off[l], m.bn_beta_off[l], m.bn_mean_off[l], m.bn_var_off[l]})
        AGX_CHECK_ARG(off >= 0 && off % 4 == 0 && off + m.dims[l + 1] <= m.n_flat, "%s: batch-norm vector of layer %d outside the flat buffer", name, l);
  }
  for (int l = 0; l < m.n_layers; ++l) {
    AGX_CHECK_ARG(m.dims[l + 1] > 0 && m.ldw[l] >= m.dims[l + 1], "%s: layer %d ldw", name, l);
    AGX_CHECK_ARG(m.w_off[l] >= 0 && m.w_off[l] % 4 == 0 && m.w_off[l] + (int64_t)m.dims[l] * m.ldw[l] <= m.n_flat,
                  "%s: kernel %d outside the flat buffer / unaligned", name, l);
    AGX_CHECK_ARG(m.b_off[l] >= 0 && m.b_off[l] + m.dims[l + 1] <= m.n_flat,
                  "%s: bias %d outside the flat buffer", name, l);
  }
  return AGX_OK;
}

struct Carver {
  int64_t off = 0;
  float *base;
  explicit Carver(float *b) : base(b) {}
  float *take(int64_t n) {
    float *p = base ? base + off : nullptr;
    off += r64(n);
    return p;
  }
};

void carve(agx_cgan *h, float *base) {
  const agx_cgan_config_t &c = h->cfg;
  const int Bm = c.max_batch, S = h->S;
  Carver cv(base);
  h->he = r4(1 + h->ncov);
  h->hx = h->he + c.gen.dims[c.gen.n_layers - 1];
  h->hxp = r4(h->hx);
  h->ldp = c.disc.dims[1];
  h->ntile = (c.x_dim + 63) / 64;
  h->ldhx = h->hxp;
  h->HD = h->hxp + c.disc.dims[1];
  for (int b = 0; b < 2; ++b) {
    h->Xbuf[b] = cv.take((int64_t)(3 * Bm + h->HD) * h->ldx);
    if (h->Xbuf[b]) h->Xbuf[b] += (size_t)h->HD * h->ldx;
  }
  h->X = h->Xbuf[0];
  const int L = c.disc.n_layers - 1;
  for (int l = 1; l <= L; ++l) {
    h->ldh[l] = r4(c.disc.dims[l]);
    h->A[l] = cv.take((int64_t)3 * Bm * h->ldh[l]);
    h->D[l] = cv.take((int64_t)(3 * Bm + (l == 1 ? h->HD : 0)) * h->ldh[l]);
    if (l == 1 && h->D[l]) h->D[l] += (size_t)h->HD * h->ldh[l];
  }
  h->out = cv.take(4 * 3 * Bm);                      // up to 4 partial dot products per row (column-split tails)
  h->A1m = cv.take((int64_t)Bm * h->ldh[1]);         // interpolated rows' layer-1 activations: ReLU mask of tail mode 1
  h->slopes = cv.take(Bm);
  h->fscale = cv.take(Bm);
  h->nslab = (c.x_dim + 31) / 32;
  h->rowsq = cv.take((int64_t)Bm * h->nslab);
  h->gramM = cv.take((int64_t)c.disc.dims[1] * h->ldh[1]);
  h->gramT = cv.take((int64_t)Bm * c.disc.dims[1]);
  h->gramU = cv.take((int64_t)Bm * h->ldh[1]);
  h->gramF = cv.take((int64_t)Bm * h->ldh[1]);
  h->ldpt = r4(c.gen.dims[c.gen.n_layers - 1]);
  h->genPT = cv.take((int64_t)c.disc.dims[1] * h->ldpt);
  h->genQ = cv.take((int64_t)c.disc.dims[1] * h->ldpt);
  h->genCs = cv.take(c.disc.dims[1]);
  h->ffHx = cv.take((int64_t)Bm * h->ldhx);
  h->ffP = cv.take((int64_t)(h->HD + Bm) * h->ldp);
  h->ffAf = cv.take((int64_t)Bm * c.disc.dims[1]);
  h->ffBp = cv.take((int64_t)h->ntile * c.disc.dims[1]);
  h->gin0 = cv.take((int64_t)S * Bm * h->ldg0);
  const int Lg = c.gen.n_layers - 1;
  for (int l = 1; l <= Lg; ++l) {
    h->ldgh[l] = r4(c.gen.dims[l]);
    h->gh[l] = cv.take((int64_t)S * Bm * h->ldgh[l]);
    h->gd[l] = cv.take((int64_t)Bm * h->ldgh[l]);
  }
  if (c.gen.bn) {
    int64_t wsd = 0;
    int maxh = 4;
    for (int l = 1; l <= Lg; ++l) {
      h->gpre[l] = cv.take((int64_t)Bm * h->ldgh[l]);
      h->bn_mean[l] = cv.take(h->ldgh[l]);
      h->bn_std[l] = cv.take(h->ldgh[l]);
      wsd = std::max<int64_t>(wsd, agx_colstats_ws_doubles(Bm, c.gen.dims[l]));
      maxh = std::max(maxh, h->ldgh[l]);
    }
    h->bn_scale = cv.take(maxh);
    h->bn_shift = cv.take(maxh);
    h->bn_ws_doubles = wsd;
    h->bn_ws = reinterpret_cast<double *>(cv.take(2 * wsd));
  }
  h->ldsx = r4(c.x_dim);
  h->F = cv.take((int64_t)S * Bm * h->ldsx);
  h->snum = cv.take((int64_t)Bm * std::max(c.gen.n_num, 1));
  h->scat = reinterpret_cast<int32_t *>(cv.take((int64_t)Bm * std::max(c.gen.emb.n_cat, 1)));
  h->zbuf = cv.take((int64_t)S * Bm * c.z_dim);
  h->abuf = cv.take((int64_t)S * Bm);
  h->losslog = cv.take(4 * 64);
  h->lrdev = cv.take(64);
  h->rngpos = reinterpret_cast<uint64_t *>(cv.take(16));
  h->loss_slot = h->rngpos ? reinterpret_cast<float **>(h->rngpos + 1) : nullptr;
  // split-K scratch: the largest request over the GEMM shapes of both steps
  int64_t need = 0;
  const int Bs[2] = {Bm, std::min(Bm, 32)};
  for (int bi = 0; bi < 2; ++bi) {
    const int B = Bs[bi];
    for (int l = 0; l < c.disc.n_layers - 1; ++l) {
      need = std::max(need, agx_gemm_ws_floats(3 * B, c.disc.dims[l + 1], c.disc.dims[l]));
      need = std::max(need, agx_gemm_ws_floats(2 * B, c.disc.dims[l + 1], c.disc.dims[l]));
      need = std::max(need, agx_gemm_ws_floats(B, c.disc.dims[l + 1], c.disc.dims[l]));
      need = std::max(need, agx_gemm_ws_floats(3 * B, c.disc.dims[l], c.disc.dims[l + 1]));
      need = std::max(need, agx_gemm_ws_floats(B, c.disc.dims[l], c.disc.dims[l + 1]));
      need = std::max(need, agx_gemm_ws_floats(c.disc.dims[l], c.disc.dims[l + 1], 3 * B));
    }
    need = std::max(need, (int64_t)2 * B * c.disc.dims[1]);  // layer 1 always goes through the slab reduce
    // Gram form of the penalty: W_g^T W_g, d_1 . M, d_1^T (f d_1), and the layer-1 product over 2B + H1 rows
    need = std::max(need, agx_gemm_ws_floats(c.disc.dims[1], c.disc.dims[1], c.x_dim));
    need = std::max(need, agx_gemm_ws_floats(B, c.disc.dims[1], c.disc.dims[1]));
    need = std::max(need, agx_gemm_ws_floats(c.disc.dims[1], c.disc.dims[1], B));
    need = std::max(need, agx_gemm_ws_floats(c.disc.dims[0], c.disc.dims[1], 2 * B + c.disc.dims[1]));
    // factored generator step: (W_out W_g)^T, d_1 P^T, d_1^T h_L, the output bias row, dW_out over H1 rows
    {
      const int HL = c.gen.dims[c.gen.n_layers - 1], H1 = c.disc.dims[1];
      need = std::max(need, agx_gemm_ws_floats(H1, HL, c.x_dim));
      need = std::max(need, agx_gemm_ws_floats(B, HL, H1));
      need = std::max(need, agx_gemm_ws_floats(H1, HL, B));
      need = std::max(need, agx_gemm_ws_floats(1, c.x_dim, H1));
      need = std::max(need, agx_gemm_ws_floats(HL, c.x_dim, H1));
      // factored fake rows: headroom rows through layer 1, Hx Px, Hx^T d_1, d_1 P^T, dW_1 over HD + B rows
      need = std::max(need, agx_gemm_ws_floats(h->HD + B, H1, c.disc.dims[0]));
      need = std::max(need, agx_gemm_ws_floats(B, H1, h->hx));
      need = std::max(need, agx_gemm_ws_floats(h->hx, H1, B));
      need = std::max(need, agx_gemm_ws_floats(c.disc.dims[0], H1, h->HD + B));
    }
    for (int l = 0; l < c.gen.n_layers; ++l) {
      need = std::max(need, agx_gemm_ws_floats(B, c.gen.dims[l + 1], c.gen.dims[l]));
      need = std::max(need, agx_gemm_ws_floats(S * B, c.gen.dims[l + 1], c.gen.dims[l]));
      need = std::max(need, agx_gemm_ws_floats(B, c.gen.dims[l], c.gen.dims[l + 1]));
      need = std::max(need, agx_gemm_ws_floats(c.gen.dims[l], c.gen.dims[l + 1], B));
    }
  }
  h->skws_floats = r64(need);
  h->skws = cv.take(h->skws_floats);
  h->skflag_words = (int)r64(agx_gemm_sk_flag_words());
  h->skflags = reinterpret_cast<uint32_t *>(cv.take(h->skflag_words));
  h->ws_floats_total = cv.off;
}

struct Gemm {  // GEMM call builder
  agx_gemm_t g;
  Gemm(agx_cgan *h, const float *A, int lda, int ta, const float *B, int ldb, int tb, float *C, int ldc,
    int M, int N, int K) {
    memset(&g, 0, sizeof(g));
    g.A = A; g.B = B; g.C = C;
    g.M = M; g.N = N; g.K = K;
    g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.ta = ta; g.tb = tb;
    g.alpha = 1.f;
    g.slope = h->cfg.slope;
    g.ws = h->skws;
    g.ws_floats = h->skws_floats;
    g.force_cfg = -1;
    g.force_splitk = 0;
    hh = h;
  }
  agx_cgan *hh;
  // allow the stream-K form (plain TN weight-gradient products of the data-parallel segments only: the one-GPU
  // iteration fuses RMSprop into the static 64x64 tiling, which measured faster than stream-K + fused update -
  // 169 against 191-200 us per 512-row dW_1 - and the eager step functions keep that tiling's summation order)
  Gemm &sk(bool on) {
    if (on) { g.sk_flags = hh->skflags; g.sk_flag_words = hh->skflag_words; }
    return *this;
  }
  Gemm &bias(const float *b) { g.bias = b; return *this; }
  Gemm &act() { g.act = 1; return *this; }
  Gemm &mask(const float *m, int ld) { g.mask = m; g.ldmask = ld; return *this; }
  Gemm &rowscale(const float *r) { g.rowscale = r; return *this; }
  Gemm &colaffine(const float *sc, const float *sh) { g.col_scale = sc; g.col_shift = sh; return *this; }
  Gemm &rowsq(float *q, int ld) { g.rowsq = q; g.rowsq_ld = ld; return *this; }
  Gemm &interp(int rows, const float *alpha) { g.interp_rows = rows; g.interp_alpha = alpha; return *this; }
  Gemm &id(int k) { g.prof_id = k; g.prof_tagged = 1; return *this; }
  Gemm &force(int cfg, int splitk) { g.force_cfg = cfg; g.force_splitk = splitk; return *this; }
  Gemm &opt(const agx::FusedOpt *f) { fo = f; return *this; }
  Gemm &dual(const agx::DualOut *d) { du = d; return *this; }
  const agx::FusedOpt *fo = nullptr;
  const agx::DualOut *du = nullptr;
  int run(hipStream_t st) { return agx::gemm_run(st, &g, fo, nullptr, du); }
  int run_slabs(hipStream_t st, int *splitk) { return agx::gemm_run(st, &g, nullptr, splitk); }
};

// tiny hidden-layer weight gradients inside the small-gradient launch (AGX_NO_TN_FUSE: own GEMM launches)
bool tn_fuse() {
  static const bool v = getenv("AGX_NO_TN_FUSE") == nullptr;
  return v;
}
// generator layer-1 input rows [0, R): [z | num | emb], covariates of source row r % B
int gen_input(agx_cgan *h, hipStream_t st, const float *z, const int32_t *cat, const float *num, int R,
              int B) {
  const agx_mlp_desc_t &m = h->cfg.gen;
  return agx_stack_rows(st, &m.emb, h->gp, z, h->cfg.z_dim, 0, R, num, m.n_num, m.n_num, cat, m.emb.n_cat,
                        B, R, h->cfg.z_dim, h->gin0, h->ldg0, nullptr, nullptr, 0);
}

// generator Dense stack over rows [0, R) of gin0; activations kept in h->gh[]; output [R, x_dim]
// batch: the rows of ONE training batch (R is a multiple of it in a captured iteration): the choice between the
// two-layer kernel and the GEMM launches must not depend on how many batches are stacked, or the captured iteration
// and the eager steps would round differently
// training (batch-norm generators only): batch statistics of the R rows (Keras training=True, tf_utils.py:244) and the
// moving statistics updated; otherwise the moving statistics, folded into the Dense epilogue (training=False, :206)
int gen_layers(agx_cgan *h, hipStream_t st, int R, float *out, int ldo, const agx::DualOut *dual = nullptr, int batch = 0,
               bool training = false, bool hidden_only = false) {
  const agx_mlp_desc_t &m = h->cfg.gen;
  const float *in = h->gin0;
  int ldin = h->ldg0;
  int l0 = 0;
  static const bool gen_tail = getenv("AGX_NO_GEN_TAIL") == nullptr;
  if (m.bn) {
    for (int l = 0; l < m.n_layers - 1; ++l) {
      float *o = h->gh[l + 1];
      const int ldo_l = h->ldgh[l + 1], H = m.dims[l + 1];
      const float *gamma = h->gp + m.bn_gamma_off[l], *beta = h->gp + m.bn_beta_off[l];
      float *mm = h->gp + m.bn_mean_off[l], *mv = h->gp + m.bn_var_off[l];
      Gemm g(h, in, ldin, 0, h->gp + m.w_off[l], m.ldw[l], 0, o, ldo_l, R, H, m.dims[l]);
      g.bias(h->gp + m.b_off[l]).id(AGX_K_GEN_FWD_HID);
      if (!training) {
        AGX_TRY(agx_bn_fold(st, mm, mv, 0, gamma, beta, m.bn_eps, h->bn_scale, h->bn_shift, H));
        AGX_TRY(g.colaffine(h->bn_scale, h->bn_shift).act().run(st));
      } else {
        AGX_CHECK_ARG(R <= h->Bm, "generator batch-norm: %d rows > max_batch %d", R, h->Bm);
        AGX_TRY(g.run(st));
        AGX_HIP(hipMemcpy2DAsync(h->gpre[l + 1], (size_t)ldo_l * 4, o, (size_t)ldo_l * 4, (size_t)H * 4, R,
                                 hipMemcpyDeviceToDevice, st));
        AGX_TRY(agx_colstats(st, o, ldo_l, R, H, h->bn_mean[l + 1], h->bn_std[l + 1], h->bn_ws, h->bn_ws_doubles));
        AGX_TRY(agx_bn_fold(st, h->bn_mean[l + 1], h->bn_std[l + 1], 1, gamma, beta, m.bn_eps, h->bn_scale, h->bn_shift, H));
        AGX_TRY(agx_affine_act(st, o, ldo_l, R, H, h->bn_scale, h->bn_shift, 1, h->cfg.slope));
        AGX_TRY(agx_bn_moving_update(st, mm, mv, h->bn_mean[l + 1], h->bn_std[l + 1], m.bn_momentum, H));
      }
      in = o;
      ldin = ldo_l;
    }
    l0 = m.n_layers - 1;
  } else
  if (gen_tail && m.n_layers == 3 && m.dims[0] <= 128 && m.dims[1] <= 256 && m.dims[2] <= 256 && (batch > 0 ? batch : R) <= 64) {
    // small batches: both hidden layers in ONE launch (tail.hip mode 2; always 4 column groups, so that a row's
    // summation order does not depend on the number of rows)
    agx::TailArgs t;
    memset(&t, 0, sizeof(t));
    t.m_slab = R; t.H1 = m.dims[1]; t.H2 = m.dims[2];
    t.in = h->gin0; t.ld_in = h->ldg0; t.K_in = m.dims[0];
    t.W1 = h->gp + m.w_off[0]; t.ldw1 = m.ldw[0]; t.b1 = h->gp + m.b_off[0];
    t.W2 = h->gp + m.w_off[1]; t.ldw2 = m.ldw[1]; t.b2 = h->gp + m.b_off[1];
    t.slope = h->cfg.slope;
    t.A1 = h->gh[1]; t.lda1 = h->ldgh[1]; t.A2 = h->gh[2]; t.lda2 = h->ldgh[2];
    t.nsplit = 4;
    AGX_TRY(agx::critic_tail(st, 2, t));
    in = h->gh[2];
    ldin = h->ldgh[2];
    l0 = 2;
  }
  for (int l = l0; l < m.n_layers - (hidden_only ? 1 : 0); ++l) {
    const bool last = (l == m.n_layers - 1);
    float *o = last ? out : h->gh[l + 1];
    const int ldo_l = last ? ldo : h->ldgh[l + 1];
    Gemm g(h, in, ldin, 0, h->gp + m.w_off[l], m.ldw[l], 0, o, ldo_l, R, m.dims[l + 1], m.dims[l]);
    g.bias(h->gp + m.b_off[l]).id(last ? AGX_K_GEN_FWD_OUT : AGX_K_GEN_FWD_HID);
    if (!last) g.act();
    if (last) g.dual(dual);
    AGX_TRY(g.run(st));
    in = o;
    ldin = ldo_l;
  }
  return AGX_OK;
}

// Small batches of the two-hidden-layer critic (the reference shape): the launches between the
// first-layer slabs and the next big product are one workgroup-per-row-group kernel (tail.hip)
bool tail_ok(const agx_cgan *h, int rows) {
  const agx_mlp_desc_t &m = h->cfg.disc;
  static const bool off = getenv("AGX_NO_TAIL") != nullptr;
  // (any batch: at 512 rows the tails still beat reduce + Dense + Dense(1) launches, 176.6 k vs 172.0 k samples/s)
  // (mode 1 zeroes the covariate columns of a row with one 256-thread group: ncov <= 256)
  return !off && m.n_layers == 3 && m.dims[1] <= 256 && m.dims[2] <= 256 && h->ncov <= 256;
}

// Gram form of the gradient penalty (critic_core), for batches of at least the first hidden width (gram_ratio2): the
// input gradient g = d_1 . W_g^T (W_g = gene rows of W_1) is never formed.  With M = W_g^T W_g [H1, H1]:
//   |g_i|^2 = d_1i . M . d_1i^T        (tf_utils.py:168-169)
//   (f g) . W_g = f * (d_1 . M)        (second-order chain's first product)
//   (f g)^T d_1 = W_g . S,  S = d_1^T diag(f) d_1   (the penalty's term of dW_1)
// so the two B x G x H1 products (and the B x G gradient, written and read back) become one H1 x H1 x G product and
// three tiny ones, and the stacked layer-1 weight-gradient product runs over 2B + H1 rows - [real; fake; W_g^T] against
// [D_1(real); D_1(fake); S] - instead of 3B.  Same algebra, other rounding: measured against the fp64 oracle like the
// direct form (tests/test_gpu_fullsize.py); AGX_NO_GRAM selects the direct form (cross-check, tests/test_gpu_knobs.py).
// twice the batch-to-width ratio from which the large-batch forms are taken (default 2: batch >= width - measured at GTEx size: 256 rows 157 k -> 194 k samples/s, 320 rows 161 k -> 218 k; AGX_GRAM_RATIO2: tuning)
int gram_ratio2() {
  static const int v = getenv("AGX_GRAM_RATIO2") ? std::max(2, atoi(getenv("AGX_GRAM_RATIO2"))) : 2;
  return v;
}
bool gram_ok(const agx_cgan *h, int B) {
  static const bool off = getenv("AGX_NO_GRAM") != nullptr;
  const agx_mlp_desc_t &m = h->cfg.disc;
  return !off && tail_ok(h, B) && 2 * B >= gram_ratio2() * m.dims[1] && B <= h->Bm;
}
// rows of the critic's stacked layer-1 weight-gradient product
int dw1_rows(const agx_cgan *h, int B) { return gram_ok(h, B) ? 2 * B + h->cfg.disc.dims[1] : 3 * B; }

// Factored generator step, same idea (gen_core): the generator's output layer is linear (tf_utils.py:88), so with d_1
// the critic's first-layer delta of the generated rows and h_L the generator's last hidden activations
//   dX = d_1 W_g^T  is never formed:   dW_out = h_L^T dX = (d_1^T h_L)^T-as-TN . W_g^T        (K = H1 instead of B)
//   db_out = colsum(dX) = colsum(d_1) . W_g^T;   dX W_out^T = d_1 . (W_out W_g)^T            (one H1 x HL x G product)
// for batches of at least the wider of the two hidden widths.  AGX_NO_GRAM selects the direct form here too.
bool gen_fact_ok(const agx_cgan *h, int B) {
  static const bool off = getenv("AGX_NO_GRAM") != nullptr;
  const agx_mlp_desc_t &m = h->cfg.disc, &gm = h->cfg.gen;
  const int HL = gm.dims[gm.n_layers - 1], H1 = m.dims[1];
  return !off && !gm.bn && 2 * B >= gram_ratio2() * std::max(HL, H1) && B <= h->Bm && h->ldx % 4 == 0;
}
// rows (K) of the generator's output-layer weight-gradient product
int gen_dw_rows(const agx_cgan *h, int B) { return gen_fact_ok(h, B) ? h->cfg.disc.dims[1] : B; }

agx::TailArgs tail_args(agx_cgan *h, int splitk, int m_slab) {
  const agx_mlp_desc_t &m = h->cfg.disc;
  agx::TailArgs t;
  memset(&t, 0, sizeof(t));
  t.ws = h->skws; t.splitk = splitk; t.m_slab = m_slab;
  t.H1 = m.dims[1]; t.H2 = m.dims[2];
  t.W2 = h->dp + m.w_off[1]; t.ldw2 = m.ldw[1];
  t.slope = h->cfg.slope;
  t.A1 = h->A[1]; t.lda1 = h->ldh[1]; t.A2 = h->A[2]; t.lda2 = h->ldh[2];
  return t;
}

// column split of a tail launch with `blocks` row groups: few groups leave most CUs idle while every workgroup pulls
// all of W_2 through one CU (tail.hip)
int tail_nsplit(int blocks) {
  static const int force = getenv("AGX_TAIL_NSPLIT") ? atoi(getenv("AGX_TAIL_NSPLIT")) : 0;
  if (force == 1 || force == 2 || force == 4) return force;
  return blocks <= 64 ? 4 : (blocks <= 128 ? 2 : 1);
}

// *out_parts (optional): in = the caller can read the critic output as partial sums, out = how many parts it got
// ws (optional): the slabs, if not the split-K scratch; ws2 (optional, interp_rows > 0): finished first-layer values of the
// fake rows (factored fake rows: the slabs then hold the real rows only)
int tail_forward(agx_cgan *h, hipStream_t st, int splitk, int m_slab, int interp_rows, const float *alpha, int n0,
                 int n1, float c0, float c1, float c2, int *out_parts = nullptr, const float *ws = nullptr,
                 const float *ws2 = nullptr) {
  const agx_mlp_desc_t &m = h->cfg.disc;
  agx::TailArgs t = tail_args(h, splitk, m_slab);
  if (ws) t.ws = ws;
  t.ws2 = ws2;
  t.A1m = h->A1m;
  t.out_ld = 3 * h->Bm;
  t.nsplit = 1;
  if (out_parts) {
    const int rpw = (interp_rows == 0 && m_slab <= 64) ? 2 : 4;   // as critic_tail deals the rows
    t.nsplit = tail_nsplit(interp_rows > 0 ? interp_rows : (m_slab + rpw - 1) / rpw);
    *out_parts = t.nsplit;
  }
  t.b1 = h->dp + m.b_off[0]; t.b2 = h->dp + m.b_off[1]; t.w3 = h->dp + m.w_off[2]; t.b3 = h->dp + m.b_off[2];
  t.alpha = alpha; t.interp_rows = interp_rows;
  t.D2 = h->D[2]; t.ldd2 = h->ldh[2]; t.out = h->out;
  t.n0 = n0; t.n1 = n1; t.c0 = c0; t.c1 = c1; t.c2 = c2;
  return agx::critic_tail(st, 0, t);
}

// critic Dense stack from layer index l0 over R rows (input of layer l0 already in place),
// then Dense(1) + seed of the backward chain D[L] = act'(A_L) * c(r) * w_out
int disc_layers_from(agx_cgan *h, hipStream_t st, int l0, int R, int n0, int n1, float c0, float c1,
                     float c2, int *out_parts = nullptr) {
  const agx_mlp_desc_t &m = h->cfg.disc;
  const int L = m.n_layers - 1;
  if (out_parts) *out_parts = 1;
  if (l0 == 0 && tail_ok(h, R)) {
    int sk = 0;
    Gemm g(h, h->X, h->ldx, 0, h->dp + m.w_off[0], m.ldw[0], 0, h->A[1], h->ldh[1], R, m.dims[1], m.dims[0]);
    AGX_TRY(g.id(AGX_K_DISC_FWD_IN).run_slabs(st, &sk));
    return tail_forward(h, st, sk, R, 0, nullptr, n0, n1, c0, c1, c2, out_parts);
  }
  for (int l = l0; l < L; ++l) {
    const float *in = l == 0 ? h->X : h->A[l];
    const int ldin = l == 0 ? h->ldx : h->ldh[l];
    Gemm g(h, in, ldin, 0, h->dp + m.w_off[l], m.ldw[l], 0, h->A[l + 1], h->ldh[l + 1], R,
        m.dims[l + 1], m.dims[l]);
    g.bias(h->dp + m.b_off[l]).act().id(l == 0 ? AGX_K_DISC_FWD_IN : AGX_K_DISC_FWD_HID);
    AGX_TRY(g.run(st));
  }
  return agx_critic_head(st, h->A[L], h->ldh[L], h->dp + m.w_off[L], h->dp + m.b_off[L], h->out,
                         h->D[L], h->ldh[L], R, m.dims[L], n0, n1, c0, c1, c2, h->cfg.slope);
}

int disc_cov_columns(agx_cgan *h, hipStream_t st, const int32_t *cat, const float *num, int R, int B) {
  const agx_mlp_desc_t &m = h->cfg.disc;
  return agx_stack_rows(st, &m.emb, h->dp, nullptr, 0, 0, 0, num, m.n_num, m.n_num, cat, m.emb.n_cat, B, R,
                        h->G, h->X, h->ldx, nullptr, nullptr, 0);
}

// first-order chain D[L] -> D[1] over R rows
int disc_backward_chain(agx_cgan *h, hipStream_t st, int R) {
  const agx_mlp_desc_t &m = h->cfg.disc;
  const int L = m.n_layers - 1;
  for (int l = L; l >= 2; --l) {
    // D[l-1] = act'(A[l-1]) * (D[l] . W_{l-1}^T);  W_{l-1}: [d_{l-1}, d_l]
    Gemm g(h, h->D[l], h->ldh[l], 0, h->dp + m.w_off[l - 1], m.ldw[l - 1], 1, h->D[l - 1],
        h->ldh[l - 1], R, m.dims[l - 1], m.dims[l]);
    g.mask(h->A[l - 1], h->ldh[l - 1]).id(AGX_K_DISC_CHAIN);
    AGX_TRY(g.run(st));
  }
  return AGX_OK;
}

// Critic gradients given X rows [0,2B) = [real; fake] (gene + covariate columns in place).
int critic_core(agx_cgan *h, hipStream_t st, const int32_t *cat, int B, int B_global, const float *alpha,
                float gp_weight, float *losses, const agx::FusedOpt *fo = nullptr) {
  const agx_mlp_desc_t &m = h->cfg.disc;
  const int L = m.n_layers - 1, R = 3 * B, G = h->G, ldx = h->ldx;
  const float invb = 1.0f / (float)B_global;
  float *Xi = h->X + (size_t)2 * B * ldx;
  const bool tail = tail_ok(h, B);
  int out_parts = 1;
  // factored fake rows (record_iteration `ff`; implies tail and gram): the fake rows x_f = h W_out + b_out (tf_utils.py:88,
  // :206) are never formed.  The HD rows in front of X - [W_out; b_out; unit rows of the covariate columns; W_g^T] - go
  // through the first layer ONCE per step: ffP = [W_out W_g; b_out W_g; W_c; 0; W_g^T W_g].  Its first hx rows turn the
  // step's [h | 1 | covariates] rows into the fake rows' first-layer values, its last H1 rows are the Gram matrix.
  const bool ff = h->ff;
  const int HD = h->HD, he = h->he, hx = h->hx, hxp = h->hxp, ldp = h->ldp;
  float *Xhd = h->X - (size_t)HD * ldx, *Dhd = h->D[1] - (size_t)HD * h->ldh[1];
  if (ff) {
    const int H1 = m.dims[1];
    const agx_mlp_desc_t &gm = h->cfg.gen;
    const int Lg = gm.n_layers - 1;
    const float *W1 = h->dp + m.w_off[0];
    // W_g^T behind the W_out rows (and the tiles' shares of b_out . W_g); then ONE product: rows he.. of the headroom
    // and the real rows through the first layer (GTEx at 512 rows: 1024 rows, the shape of the old [real; fake] forward)
    AGX_TRY(agx::wt_rows(st, W1, m.ldw[0], G, H1, Xhd + (size_t)hxp * ldx, ldx, h->ncov, h->gp + gm.b_off[Lg], h->ffBp));
    // (also the step's refresh of the real rows' covariate columns from the just-updated tables: before the product)
    AGX_TRY(agx::ff_hx(st, h->gh[Lg] + (size_t)h->ff_row0 * h->ldgh[Lg], h->ldgh[Lg], gm.dims[Lg], h->X, ldx, G, h->ncov, he,
                       h->ffHx, h->ldhx, B, h->ffBp, h->ntile, W1, m.ldw[0], H1, h->ffP, ldp, &m.emb, h->dp, h->snum, m.n_num, cat));
    AGX_TRY(Gemm(h, Xhd + (size_t)he * ldx, ldx, 0, W1, m.ldw[0], 0, h->ffP + (size_t)he * ldp, ldp, HD - he + B, H1, m.dims[0])
                .id(AGX_K_DISC_FWD_IN).run(st));
    AGX_TRY(Gemm(h, h->ffHx, h->ldhx, 0, h->ffP, ldp, 0, h->ffAf, H1, B, H1, hx).id(AGX_K_DISC_FWD_HID).run(st));
    // the tail takes the real rows' finished values as ONE slab (ffP is compact: ldp = H1) and the fake rows' from ffAf
    AGX_TRY(tail_forward(h, st, 1, B, B, alpha, B, B, -invb, invb, 1.f, &out_parts, h->ffP + (size_t)HD * ldp, h->ffAf));
  } else if (tail) {
    // layer-1 slabs of [real; fake]; one launch finishes them (interpolated rows by linearity), runs
    // the second hidden layer and Dense(1) on the 3B rows and seeds the backward chain
    int sk = 0;
    AGX_TRY(Gemm(h, h->X, ldx, 0, h->dp + m.w_off[0], m.ldw[0], 0, h->A[1], h->ldh[1], 2 * B, m.dims[1],
                 m.dims[0]).id(AGX_K_DISC_FWD_IN).run_slabs(st, &sk));
    AGX_TRY(tail_forward(h, st, sk, 2 * B, B, alpha, B, B, -invb, invb, 1.f, &out_parts));
  } else {
    // layer 1 on [real; fake]; the reduce emits the interpolated rows by linearity (:162-167, :213-214)
    AGX_TRY(Gemm(h, h->X, ldx, 0, h->dp + m.w_off[0], m.ldw[0], 0, h->A[1], h->ldh[1], 2 * B, m.dims[1],
                 m.dims[0]).bias(h->dp + m.b_off[0]).act().interp(B, alpha).id(AGX_K_DISC_FWD_IN).run(st));
    // deeper layers + Dense(1) on the 3B stacked rows, seed c = [-1/B | +1/B | 1]
    AGX_TRY(disc_layers_from(h, st, 1, R, B, B, -invb, invb, 1.f));
  }
  AGX_TRY(disc_backward_chain(h, st, R));
  float *D1i = h->D[1] + (size_t)2 * B * h->ldh[1];
  // (the layer-1 weight-gradient product comes LAST when it carries the flat update, see below)
  const bool dw1_last = fo && fo->rest_n > 0;
  // hidden-layer products of a few 32x32 tiles over K = 3B rows ride the small-gradient launch
  agx::TnJobs tn;
  tn.n = 0;
  const bool gram = tail && gram_ok(h, B);
  h->dw1_rows = ff ? HD + B : dw1_rows(h, B);
  AGX_CHECK_ARG(!ff || gram, "critic_core: factored fake rows without the Gram form");
  if (gram) {
    // (see gram_ok) W_g^T into the inter rows of X; M; T = d_1 . M; one launch: slopes, f, e_1, e_2, copies of d_1
    // and f*d_1; S into the first H1 inter rows of D_1
    const int H1 = m.dims[1], ldh1 = h->ldh[1];
    const float *Wg = h->dp + m.w_off[0];
    const float *M = ff ? h->ffP + (size_t)hxp * ldp : h->gramM;   // (ff: H1 rows of the one first-layer product)
    const int ldM = ff ? ldp : ldh1;
    if (!ff) {
    AGX_TRY(agx::wt_rows(st, Wg, m.ldw[0], G, H1, Xi, ldx, h->ncov));
    // (64x64 tiles over 32 K slices: 34.3 us against 40.8 for the plan's 128x128 tiles over 114 slices, whose partial
    //  sums are 30 MB; profiles/r03_gram_form.txt)
    Gemm gm(h, Wg, m.ldw[0], 1, Wg, m.ldw[0], 0, h->gramM, ldh1, H1, H1, G);
    if (H1 > 128 && G >= 4096) gm.force(1, 32);
    AGX_TRY(gm.id(AGX_K_DISC_DX).run(st));
    }
    AGX_TRY(Gemm(h, D1i, ldh1, 0, M, ldM, 0, h->gramT, H1, B, H1, H1).id(AGX_K_GP2_IN).run(st));
    agx::TailArgs t = tail_args(h, 1, B);
    t.ws = h->gramT;
    t.row0 = 2 * B;
    t.slopes = h->slopes; t.fscale = h->fscale; t.D1 = h->D[1]; t.ldd1 = ldh1;
    t.coef = gp_weight * 2.0f * invb;
    t.A1m = h->A1m;
    t.out_ld = 3 * h->Bm;
    t.nsplit = tail_nsplit((B + 3) / 4);
    t.gram = 1; t.gU = h->gramU; t.gF = h->gramF; t.ldg = ldh1;
    AGX_TRY(agx::critic_tail(st, 1, t));
    // S in front of the real rows' deltas (ff) / over the first H1 inter rows; ff: the fake rows' term of dW_1,
    // [W_out; b_out; unit rows]^T . (Hx^T d_1(fake)).  Where the layer-1 product is the step's last launch both ride the
    // small-gradient launch in front of it (one wave per 32x32 tile over the B rows: two launches of ~7 us less)
    float *Sout = ff ? Dhd + (size_t)hxp * ldh1 : D1i;
    if (ff && dw1_last && tn_fuse() && B <= 512 && tn.n + 2 <= AGX_MAX_LAYERS) {
      tn.job[tn.n++] = agx::TnJobs::Job{h->gramU, h->gramF, Sout, ldh1, ldh1, ldh1, H1, H1, B};
      tn.job[tn.n++] = agx::TnJobs::Job{h->ffHx, h->D[1] + (size_t)B * ldh1, Dhd, h->ldhx, ldh1, ldh1, hx, H1, B};
    } else {
    AGX_TRY(Gemm(h, h->gramU, ldh1, 1, h->gramF, ldh1, 0, Sout, ldh1, H1, H1, B).id(AGX_K_GP2_IN).run(st));
    if (ff)
      AGX_TRY(Gemm(h, h->ffHx, h->ldhx, 1, h->D[1] + (size_t)B * ldh1, ldh1, 0, Dhd, ldh1, hx, H1, B).id(AGX_K_GP2_IN).run(st));
    }
  } else {
  // g = d_1 . W_1[:G]^T -> inter rows of X, with the per-slab row sums of squares as a side
  // output of the epilogue                                                      :168-169
  AGX_TRY(Gemm(h, D1i, h->ldh[1], 0, h->dp + m.w_off[0], m.ldw[0], 1, Xi, ldx, B, G, m.dims[1])
              .rowsq(h->rowsq, h->nslab).id(AGX_K_DISC_DX).run(st));
  }
  // slopes; f = weight*(2/B)*(s-1)/s; D_1[inter] *= f (dg = f*g stays implicit); the penalty has
  // no gradient w.r.t. the covariate rows of W_1: zero those columns of the inter rows   :169-170
  if (gram) {
  } else if (tail) {
    // slabs of g . W_1[:G]; one launch: slopes, f, D_1[inter] *= f, e_1, e_2 (in place), covariate zeroing
    int sk = 0;
    float *A1i = h->A[1] + (size_t)2 * B * h->ldh[1];
    AGX_TRY(Gemm(h, Xi, ldx, 0, h->dp + m.w_off[0], m.ldw[0], 0, A1i, h->ldh[1], B, m.dims[1], G)
                .id(AGX_K_GP2_IN).run_slabs(st, &sk));
    agx::TailArgs t = tail_args(h, sk, B);
    t.row0 = 2 * B;
    t.rowsq = h->rowsq; t.rowsq_ld = h->nslab; t.nslab = h->nslab;
    t.slopes = h->slopes; t.fscale = h->fscale; t.D1 = h->D[1]; t.ldd1 = h->ldh[1];
    t.xcov = Xi + G; t.ldx = ldx; t.ncov = h->ncov;
    t.coef = gp_weight * 2.0f * invb;
    t.A1m = h->A1m;
    t.out_ld = 3 * h->Bm;
    t.nsplit = tail_nsplit((B + (B <= 64 ? 2 : 4) - 1) / (B <= 64 ? 2 : 4));
    AGX_TRY(agx::critic_tail(st, 1, t));
  } else {
  AGX_TRY(agx_gp_finalize(st, h->rowsq, h->nslab, h->nslab, h->slopes, h->fscale, D1i, h->ldh[1],
                          m.dims[1], Xi + G, ldx, h->ncov, B, gp_weight * 2.0f * invb));
  // second-order chain, in place over the inter rows of the activations
  {
    const float *in = Xi;
    int ldin = ldx, K = G;
    for (int l = 0; l < L; ++l) {
      float *Ai = h->A[l + 1] + (size_t)2 * B * h->ldh[l + 1];
      Gemm g(h, in, ldin, 0, h->dp + m.w_off[l], m.ldw[l], 0, Ai, h->ldh[l + 1], B, m.dims[l + 1], K);
      g.mask(Ai, h->ldh[l + 1]).id(l == 0 ? AGX_K_GP2_IN : AGX_K_GP2_HID);
      if (l == 0) g.rowscale(h->fscale);  // e_1 = relu'(A_1) * ((f*g) . W_1[:G])
      AGX_TRY(g.run(st));
      in = Ai;
      ldin = h->ldh[l + 1];
      K = m.dims[l + 1];
    }
  }
  }
  // kernel gradients: one stacked product per layer (K = 3B), everything on the one stream (side-stream
  // branches for the small products were measured and removed: every cross-queue fork/join costs 5-9 us of
  // idle GPU, 165.4 k against 170.6 k samples/s at 512 rows, 0.89 against 0.79 ms per iteration at 32)
  // With the flat update appended to the layer-1 product (fo->rest_n), that launch must come LAST:
  // it overwrites the covariate rows of W_1 the embedding gradient reads and needs every other
  // gradient of the step complete.
  auto weight_grad = [&](int l) -> int {
    if (l == 0 && h->skip_big) return AGX_OK;   // launched in row chunks by agx_cgan_dp_iteration (big_grad_chunk)
    const float *in = l == 0 ? (ff ? Xhd : h->X) : h->A[l];
    const int ldin = l == 0 ? ldx : h->ldh[l];
    // (where the layer-1 product is the step's last launch, also over up to 2048 rows: the launch splits a long K over
    //  a block's waves, 6 us of product + 5 us of slab reduce less per step at 512 rows)
    if (l > 0 && tn_fuse() && tn.n < AGX_MAX_LAYERS &&
        (agx::tn_job_ok(m.dims[l], m.dims[l + 1], R) ||
         (dw1_last && R <= 2048 && agx::cdiv(m.dims[l], 32) * agx::cdiv(m.dims[l + 1], 32) <= 256))) {
      tn.job[tn.n++] = agx::TnJobs::Job{in, h->D[l + 1], h->dg + m.w_off[l], ldin, h->ldh[l + 1], m.ldw[l],
                                        m.dims[l], m.dims[l + 1], R};
      return AGX_OK;
    }
    // (ff: [headroom; real]^T . [Hx^T d_1(fake); 0; S; D_1(real)] over HD + B rows)
    Gemm g(h, in, ldin, 1, (l == 0 && ff) ? Dhd : h->D[l + 1], h->ldh[l + 1], 0, h->dg + m.w_off[l], m.ldw[l], m.dims[l],
           m.dims[l + 1], l == 0 ? h->dw1_rows : R);
    g.id(l == 0 ? AGX_K_DISC_DW_IN : AGX_K_DISC_DW_HID).sk(l == 0 && !fo && h->use_sk);
    // layer 1 in a captured iteration: rows [0,G) of W_1 (all rows with dw1_last) are updated in the
    // epilogue.  Safe: every reader of those rows (forward, input gradient, second-order product)
    // precedes this launch in stream order
    if (l == 0) g.opt(fo);
    return g.run(st);
  };
  for (int l = L - 1; l >= (dw1_last ? 1 : 0); --l) AGX_TRY(weight_grad(l));
  // everything small in one launch: bias gradients (rows 0..2B), Dense(1) gradient, embedding
  // tables (from D_1 rows 0..2B and the covariate rows of W_1), loss scalars      :174-181, :217-218
  agx_small_grads_t j;
  memset(&j, 0, sizeof(j));
  j.n_colsum = L;
  for (int l = 0; l < L; ++l) {
    j.cs_A[l] = h->D[l + 1]; j.cs_lda[l] = h->ldh[l + 1]; j.cs_r0[l] = 0; j.cs_r1[l] = 2 * B;
    j.cs_H[l] = m.dims[l + 1]; j.cs_out[l] = h->dg + m.b_off[l];
  }
  j.w_A = h->A[L]; j.w_lda = h->ldh[L]; j.w_R = R; j.w_H = m.dims[L]; j.w_n0 = B; j.w_n1 = B;
  j.w_c0 = -invb; j.w_c1 = invb; j.w_c2 = 1.f;
  j.w_out = h->dg + m.w_off[L]; j.w_bias_out = h->dg + m.b_off[L];
  if (m.emb.n_cat > 0) {
    j.emb_grad = h->dg; j.emb = m.emb; j.emb_D = h->D[1]; j.emb_ldd = h->ldh[1]; j.emb_H = m.dims[1];
    j.emb_W = h->dp + m.w_off[0]; j.emb_ldw = m.ldw[0]; j.emb_col0 = G + m.n_num;
    j.emb_cat = cat; j.emb_ld_cat = m.emb.n_cat; j.emb_B = B; j.emb_rows = 2 * B;
  }
  j.l_losses = losses; j.l_out = h->out; j.l_slopes = h->slopes; j.l_n0 = B; j.l_n1 = B; j.l_ns = B;
  j.l_out_parts = out_parts; j.l_out_ld = 3 * h->Bm;
  j.l_inv_b = invb; j.l_gp_weight = gp_weight; j.l_fake_sign = 1.f;
  AGX_TRY(agx::small_grads_run(st, &j, agx::LossCopy{nullptr, nullptr, 0}, &tn));
  if (dw1_last) AGX_TRY(weight_grad(0));
  return AGX_OK;
}

// Generator gradients given X rows [0,B) = generated rows (+covariates) and the generator
// activations of those rows at row offset gen_row0 of gin0 / gh[].
int gen_core(agx_cgan *h, hipStream_t st, const int32_t *cat, int B, int B_global, int gen_row0,
             float *losses, const agx::FusedOpt *fo = nullptr, const agx::LossCopy *lc = nullptr) {
  const agx_mlp_desc_t &m = h->cfg.disc, &gm = h->cfg.gen;
  const int Lg = gm.n_layers - 1, G = h->G, ldx = h->ldx;
  const float invb = 1.0f / (float)B_global;
  const bool fact = gen_fact_ok(h, B), ff = h->ff;
  const int H1c = m.dims[1], HLg = gm.dims[Lg];
  AGX_CHECK_ARG(!ff || fact, "gen_core: factored fake rows without the factored generator step");
  // (ff: W_g^T sits in front of X, see critic_core)
  float *dX = ff ? h->X - (size_t)H1c * ldx : h->X + (size_t)B * ldx;
  // critic forward (:250) + seed -1/B of d mean(-D)/d out; critic weights untouched   :255
  int out_parts = 1;
  if (ff) {
    // the generated rows are never formed: their first-layer values are [h | 1 | covariates] . [W_out W_g; b_out W_g; W_c]
    float *Xhd = h->X - (size_t)h->HD * ldx;
    const float *W1 = h->dp + m.w_off[0];
    AGX_TRY(agx::wt_rows(st, W1, m.ldw[0], G, H1c, dX, ldx, h->ncov, h->gp + gm.b_off[Lg], h->ffBp));
    AGX_TRY(Gemm(h, Xhd + (size_t)h->he * ldx, ldx, 0, W1, m.ldw[0], 0, h->ffP + (size_t)h->he * h->ldp, h->ldp, HLg, H1c,
                 m.dims[0]).id(AGX_K_DISC_FWD_IN).run(st));
    AGX_TRY(agx::ff_hx(st, h->gh[Lg] + (size_t)gen_row0 * h->ldgh[Lg], h->ldgh[Lg], HLg, h->X, ldx, G, h->ncov, h->he, h->ffHx,
                       h->ldhx, B, h->ffBp, h->ntile, W1, m.ldw[0], H1c, h->ffP, h->ldp, &m.emb, h->dp, h->snum, m.n_num, cat));
    AGX_TRY(Gemm(h, h->ffHx, h->ldhx, 0, h->ffP, h->ldp, 0, h->ffAf, H1c, B, H1c, h->hx).id(AGX_K_DISC_FWD_HID).run(st));
    AGX_TRY(tail_forward(h, st, 1, B, 0, nullptr, B, 0, -invb, 0.f, 0.f, &out_parts, h->ffAf));
  } else {
  AGX_TRY(disc_layers_from(h, st, 0, B, B, 0, -invb, 0.f, 0.f, &out_parts));
  }
  AGX_TRY(disc_backward_chain(h, st, B));
  if (fact) {
    // (see gen_fact_ok) W_g^T into the rows the input gradient would take; P^T = W_g^T W_out^T; Q = d_1^T h_L
    if (!ff) {
    AGX_TRY(agx::wt_rows(st, h->dp + m.w_off[0], m.ldw[0], G, H1c, dX, ldx, 0));
    Gemm gp(h, dX, ldx, 0, h->gp + gm.w_off[Lg], gm.ldw[Lg], 1, h->genPT, h->ldpt, H1c, HLg, G);
    AGX_TRY(gp.id(AGX_K_GEN_BWD_OUT).run(st));
    }
    AGX_TRY(Gemm(h, h->D[1], h->ldh[1], 1, h->gh[Lg] + (size_t)gen_row0 * h->ldgh[Lg], h->ldgh[Lg], 0, h->genQ, h->ldpt,
                 H1c, HLg, B).id(AGX_K_GEN_DW_HID).run(st));
  } else {
  AGX_TRY(Gemm(h, h->D[1], h->ldh[1], 0, h->dp + m.w_off[0], m.ldw[0], 1, dX, ldx, B, G, m.dims[1])
              .id(AGX_K_DISC_DX).run(st));
  }
  // generator backward: delta chain and weight gradients on the one stream
  agx_small_grads_t j;
  memset(&j, 0, sizeof(j));
  agx::TnJobs tn;
  tn.n = 0;
  const float *dout = dX;
  int lddout = ldx;
  // small batches of the two-hidden-layer generator: split-K reduce + hidden delta product as one tail launch
  static const bool gen_tail_off = getenv("AGX_NO_GEN_TAIL") != nullptr;
  const bool tail3 = !gen_tail_off && !fact && Lg == 2 && !gm.bn && B <= 64 && gm.dims[1] <= 256 && gm.dims[2] <= 256 &&
                     gm.ldw[1] % 4 == 0;
  for (int l = Lg; l >= 0; --l) {
    const float *in = l == 0 ? h->gin0 + (size_t)gen_row0 * h->ldg0 : h->gh[l] + (size_t)gen_row0 * h->ldgh[l];
    const int ldin = l == 0 ? h->ldg0 : h->ldgh[l];
    // output layer in a captured iteration: W_out is updated in the epilogue of its gradient product,
    // which must then follow the delta product that still reads W_out (same stream, after it)
    const bool fused = fo && l == Lg;
    auto weight_grad = [&]() -> int {
      if (l == Lg && h->skip_big) return AGX_OK;   // (see critic_core)
      const bool f = fact && l == Lg;   // dW_out = Q-as-TN . W_g^T over H1 rows
      Gemm g(h, f ? h->genQ : in, f ? h->ldpt : ldin, 1, dout, lddout, 0, h->gg + gm.w_off[l], gm.ldw[l], gm.dims[l],
             gm.dims[l + 1], f ? H1c : B);
      g.id(l == Lg ? AGX_K_GEN_DW_OUT : AGX_K_GEN_DW_HID);
      if (fused) g.opt(fo);
      return g.run(st);
    };
    if (l == Lg && h->skip_big) {
      // launched in row chunks by agx_cgan_dp_iteration
    } else if (!fused && tn_fuse() && agx::tn_job_ok(gm.dims[l], gm.dims[l + 1], (fact && l == Lg) ? H1c : B)) {
      // a few 32x32 tiles over K = B rows: rides the small-gradient launch at the end of the step
      if (fact && l == Lg)   // (factored output layer: Q-as-TN . W_g^T over H1 rows)
        tn.job[tn.n++] = agx::TnJobs::Job{h->genQ, dout, h->gg + gm.w_off[l], h->ldpt, lddout, gm.ldw[l], gm.dims[l],
                                          gm.dims[l + 1], H1c};
      else
      tn.job[tn.n++] = agx::TnJobs::Job{in, dout, h->gg + gm.w_off[l], ldin, lddout, gm.ldw[l], gm.dims[l],
                                        gm.dims[l + 1], B};
    } else if (!fused) {
      AGX_TRY(weight_grad());
    }
    if (fact && l == Lg) {   // column sums of d_1 now, the bias row = those . W_g^T behind the small-gradient launch
      j.cs_A[j.n_colsum] = h->D[1]; j.cs_lda[j.n_colsum] = h->ldh[1]; j.cs_r0[j.n_colsum] = 0; j.cs_r1[j.n_colsum] = B;
      j.cs_H[j.n_colsum] = H1c; j.cs_out[j.n_colsum] = h->genCs;
    } else {
    j.cs_A[j.n_colsum] = dout; j.cs_lda[j.n_colsum] = lddout; j.cs_r0[j.n_colsum] = 0; j.cs_r1[j.n_colsum] = B;
    j.cs_H[j.n_colsum] = gm.dims[l + 1]; j.cs_out[j.n_colsum] = h->gg + gm.b_off[l];
    }
    ++j.n_colsum;
    if (l == Lg && tail3) {
      // the output layer's delta product leaves raw split-K slabs; one tail launch sums them, applies both ReLU masks
      // and runs the hidden delta product between them (tail.hip mode 3): one launch instead of reduce + product
      int sk = 0;
      Gemm g(h, dout, lddout, 0, h->gp + gm.w_off[l], gm.ldw[l], 1, h->gd[l], h->ldgh[l], B, gm.dims[l], gm.dims[l + 1]);
      AGX_TRY(g.id(AGX_K_GEN_BWD_OUT).run_slabs(st, &sk));
      agx::TailArgs t;
      memset(&t, 0, sizeof(t));
      t.ws = h->skws; t.splitk = sk; t.m_slab = B;
      t.H1 = gm.dims[l]; t.H2 = gm.dims[l - 1];
      t.W2 = h->gp + gm.w_off[l - 1]; t.ldw2 = gm.ldw[l - 1];
      t.slope = h->cfg.slope;
      t.A1 = h->gd[l]; t.lda1 = h->ldgh[l]; t.A2 = h->gd[l - 1]; t.lda2 = h->ldgh[l - 1];
      t.M1 = in; t.ldm1 = ldin;
      t.M2 = h->gh[l - 1] + (size_t)gen_row0 * h->ldgh[l - 1]; t.ldm2 = h->ldgh[l - 1];
      t.nsplit = tail_nsplit((B + 1) / 2);
      AGX_TRY(agx::critic_tail(st, 3, t));
    } else if (l == Lg - 1 && tail3) {
      // (gd[l] written by the tail launch above)
    } else if (l > 0) {
      // (factored output layer: dX W_out^T = d_1 . P^T)
      // (ff: P = W_out W_g is the first HL rows of the headroom product)
      Gemm g = (ff && l == Lg) ? Gemm(h, h->D[1], h->ldh[1], 0, h->ffP + (size_t)h->he * h->ldp, h->ldp, 1, h->gd[l], h->ldgh[l], B, HLg, H1c)
             : (fact && l == Lg) ? Gemm(h, h->D[1], h->ldh[1], 0, h->genPT, h->ldpt, 0, h->gd[l], h->ldgh[l], B, HLg, H1c)
                                 : Gemm(h, dout, lddout, 0, h->gp + gm.w_off[l], gm.ldw[l], 1, h->gd[l], h->ldgh[l], B, gm.dims[l],
                                        gm.dims[l + 1]);
      g.mask(in, ldin).id(l == Lg ? AGX_K_GEN_BWD_OUT : AGX_K_GEN_BWD_HID);
      AGX_TRY(g.run(st));
      if (gm.bn)   // gd[l] = dL/d(normalised value) -> dL/d(Dense output), gamma / beta gradients of hidden layer l - 1
        AGX_TRY(agx_bn_backward(st, h->gd[l], h->ldgh[l], h->gpre[l], h->ldgh[l], h->bn_mean[l], h->bn_std[l],
                                h->gp + gm.bn_gamma_off[l - 1], gm.bn_eps, B, gm.dims[l], h->gg + gm.bn_gamma_off[l - 1],
                                h->gg + gm.bn_beta_off[l - 1]));
    }
    if (fused && !(fo->rest_n > 0)) AGX_TRY(weight_grad());
    if (l > 0) {
      dout = h->gd[l];
      lddout = h->ldgh[l];
    }
  }
  if (gm.emb.n_cat > 0) {  // dout is now the first generator layer's delta gd[1]
    j.emb_grad = h->gg; j.emb = gm.emb; j.emb_D = dout; j.emb_ldd = lddout; j.emb_H = gm.dims[1];
    j.emb_W = h->gp + gm.w_off[0]; j.emb_ldw = gm.ldw[0]; j.emb_col0 = h->cfg.z_dim + gm.n_num;
    j.emb_cat = cat; j.emb_ld_cat = gm.emb.n_cat; j.emb_B = B; j.emb_rows = B;
  }
  j.l_losses = losses; j.l_out = h->out; j.l_n0 = B; j.l_n1 = 0; j.l_ns = 0;   // mean(-D(fake))  :253
  j.l_out_parts = out_parts; j.l_out_ld = 3 * h->Bm;
  j.l_inv_b = invb; j.l_gp_weight = 0.f; j.l_fake_sign = 1.f;
  AGX_TRY(agx::small_grads_run(st, &j, lc ? *lc : agx::LossCopy{nullptr, nullptr, 0}, &tn));
  if (fact)   // db_out = colsum(d_1) . W_g^T
    AGX_TRY(Gemm(h, h->genCs, H1c, 0, dX, ldx, 0, h->gg + gm.b_off[Lg], G, 1, G, H1c).id(AGX_K_GEN_DW_HID).run(st));
  if (fo && fo->rest_n > 0) {
    // output-layer product LAST: its launch also runs the flat update of every other generator
    // parameter, whose gradients the launches above have completed
    Gemm g(h, fact ? h->genQ : h->gh[Lg] + (size_t)gen_row0 * h->ldgh[Lg], fact ? h->ldpt : h->ldgh[Lg], 1, dX, ldx, 0,
           h->gg + gm.w_off[Lg], gm.ldw[Lg], gm.dims[Lg], gm.dims[Lg + 1], fact ? H1c : B);
    AGX_TRY(g.id(AGX_K_GEN_DW_OUT).opt(fo).run(st));
  }
  return AGX_OK;
}

int check_step(const agx_cgan *h, const agx_step_args_t *a, bool critic) {
  AGX_CHECK_ARG(h && a, "null handle/args");
  if (!h->bound) {
    agx::set_error("agx_cgan: workspace/params not bound");
    return AGX_E_STATE;
  }
  AGX_CHECK_ARG(a->B > 0 && a->B <= h->Bm, "batch %d outside (0, max_batch=%d]", a->B, h->Bm);
  AGX_CHECK_ARG(a->B_global >= a->B, "B_global %d < B %d", a->B_global, a->B);
  AGX_CHECK_ARG(a->z && a->losses, "z / losses missing");
  AGX_CHECK_ARG(h->cfg.gen.emb.n_cat == 0 || a->cat, "cat missing");
  AGX_CHECK_ARG(h->cfg.gen.n_num == 0 || a->num, "num missing");
  if (critic) {
    AGX_CHECK_ARG(a->x && a->ldx >= h->G && a->alpha, "x / alpha missing or ldx < x_dim");
    AGX_CHECK_ARG(!a->aug_mask || (a->aug_noise_fake && a->aug_noise_real), "aug noise missing");
  } else {
    AGX_CHECK_ARG(!a->aug_mask || a->aug_noise_fake, "aug noise missing");
  }
  return AGX_OK;
}

}  // namespace

extern "C" int agx_cgan_create(const agx_cgan_config_t *cfg, agx_cgan_t **out) {
  AGX_CHECK_ARG(cfg && out, "agx_cgan_create: null argument");
  AGX_CHECK_ARG(cfg->x_dim > 0 && cfg->z_dim > 0 && cfg->max_batch > 0, "agx_cgan_create: bad dims");
  AGX_CHECK_ARG(cfg->slope >= 0.f && cfg->slope < 1.f, "agx_cgan_create: slope must be in [0,1)");
  AGX_TRY(validate_mlp(cfg->gen, "generator", cfg->z_dim, cfg->x_dim));
  AGX_TRY(validate_mlp(cfg->disc, "critic", cfg->x_dim, 1));
  AGX_CHECK_ARG(cfg->disc.ldw[cfg->disc.n_layers - 1] == 1, "critic output kernel must have ldw == 1");
  AGX_CHECK_ARG(!cfg->disc.bn, "agx_cgan_create: batch-norm in the critic is not supported (the gradient penalty is a per-sample quantity)");
  AGX_CHECK_ARG(cfg->gen.n_num == cfg->disc.n_num && cfg->gen.emb.n_cat == cfg->disc.emb.n_cat,
                "generator and critic must share the covariate layout");
  agx_cgan *h = new (std::nothrow) agx_cgan();
  AGX_CHECK_ARG(h, "agx_cgan_create: out of host memory");
  h->cfg = *cfg;
  h->Bm = cfg->max_batch;
  h->G = cfg->x_dim;
  h->Dd = cfg->disc.dims[0];
  h->Dg = cfg->gen.dims[0];
  h->ldx = r4(h->Dd);
  h->ldg0 = r4(h->Dg);
  h->ncov = h->Dd - h->G;
  h->S = cfg->max_steps > 0 ? cfg->max_steps : 6;
  AGX_CHECK_ARG(h->S <= 63, "agx_cgan_create: max_steps %d > 63", h->S);
  carve(h, nullptr);
  *out = h;
  return AGX_OK;
}

extern "C" int agx_cgan_destroy(agx_cgan_t *h) {
  delete h;
  return AGX_OK;
}

extern "C" int64_t agx_cgan_workspace_bytes(const agx_cgan_t *h) {
  return h ? h->ws_floats_total * (int64_t)sizeof(float) : 0;
}

extern "C" int agx_cgan_bind(agx_cgan_t *h, float *gen_params, float *gen_grads, float *disc_params,
                             float *disc_grads, void *workspace, int64_t workspace_bytes) {
  AGX_CHECK_ARG(h && gen_params && gen_grads && disc_params && disc_grads && workspace,
                "agx_cgan_bind: null argument");
  AGX_CHECK_ARG(agx::aligned16(gen_params) && agx::aligned16(gen_grads) && agx::aligned16(disc_params) &&
                    agx::aligned16(disc_grads) && agx::aligned16(workspace),
                "agx_cgan_bind: buffers must be 16-byte aligned");
  if (workspace_bytes < agx_cgan_workspace_bytes(h)) {
    agx::set_error("agx_cgan_bind: workspace %lld B < required %lld B", (long long)workspace_bytes,
                   (long long)agx_cgan_workspace_bytes(h));
    return AGX_E_WORKSPACE;
  }
  h->gp = gen_params; h->gg = gen_grads; h->dp = disc_params; h->dg = disc_grads;
  carve(h, static_cast<float *>(workspace));
  // stream-K hand-off words start at zero; every launch leaves them at zero again (set-up call: not captured)
  AGX_HIP(hipMemset(h->skflags, 0, (size_t)h->skflag_words * sizeof(uint32_t)));
  h->bound = true;
  return AGX_OK;
}

extern "C" int agx_cgan_gen_forward(agx_cgan_t *h, void *stream, const float *z, const int32_t *cat,
                                    const float *num, int32_t B, float *out, int32_t ldo) {
  AGX_CHECK_ARG(h && z && out, "agx_cgan_gen_forward: null argument");
  if (!h->bound) { agx::set_error("agx_cgan_gen_forward: not bound"); return AGX_E_STATE; }
  AGX_CHECK_ARG(B > 0 && B <= h->Bm && ldo >= h->G, "agx_cgan_gen_forward: bad B=%d / ldo=%d", B, ldo);
  AGX_TRY(gen_input(h, agx::as_stream(stream), z, cat, num, B, B));
  return gen_layers(h, agx::as_stream(stream), B, out, ldo);
}

extern "C" int agx_cgan_disc_forward(agx_cgan_t *h, void *stream, const float *x, int32_t ldx,
                                     const int32_t *cat, const float *num, int32_t B, float *scores) {
  AGX_CHECK_ARG(h && x && scores, "agx_cgan_disc_forward: null argument");
  if (!h->bound) { agx::set_error("agx_cgan_disc_forward: not bound"); return AGX_E_STATE; }
  AGX_CHECK_ARG(B > 0 && B <= 3 * h->Bm && ldx >= h->G, "agx_cgan_disc_forward: bad B / ldx");
  hipStream_t st = agx::as_stream(stream);
  AGX_HIP(hipMemcpy2DAsync(h->X, (size_t)h->ldx * 4, x, (size_t)ldx * 4, (size_t)h->G * 4, B,
                           hipMemcpyDeviceToDevice, st));
  AGX_TRY(disc_cov_columns(h, st, cat, num, B, B));
  AGX_TRY(disc_layers_from(h, st, 0, B, B, 0, 0.f, 0.f, 0.f));
  AGX_HIP(hipMemcpyAsync(scores, h->out, (size_t)B * 4, hipMemcpyDeviceToDevice, st));
  return AGX_OK;
}

extern "C" int agx_cgan_disc_grads(agx_cgan_t *h, void *stream, const agx_step_args_t *a) {
  AGX_TRY(check_step(h, a, true));
  hipStream_t st = agx::as_stream(stream);
  const int B = a->B, G = h->G, ldx = h->ldx;
  float *Xr = h->X, *Xf = h->X + (size_t)B * ldx;
  // fake batch straight into rows B..2B (generator is not differentiated, tf_utils.py:206)
  AGX_TRY(gen_input(h, st, a->z, a->cat, a->num, B, B));
  AGX_TRY(gen_layers(h, st, B, Xf, ldx));
  AGX_HIP(hipMemcpy2DAsync(Xr, (size_t)ldx * 4, a->x, (size_t)a->ldx * 4, (size_t)G * 4, B,
                           hipMemcpyDeviceToDevice, st));
  if (a->aug_mask) {  // tf_utils.py:209-210
    AGX_TRY(agx_aug_add(st, Xf, ldx, a->aug_mask, a->aug_noise_fake, G, 1, B, G));
    AGX_TRY(agx_aug_add(st, Xr, ldx, a->aug_mask, a->aug_noise_real, G, 0, B, G));
  }
  AGX_TRY(disc_cov_columns(h, st, a->cat, a->num, 2 * B, B));
  return critic_core(h, st, a->cat, B, a->B_global, a->alpha, a->gp_weight, a->losses);
}

extern "C" int agx_cgan_gen_grads(agx_cgan_t *h, void *stream, const agx_step_args_t *a) {
  AGX_TRY(check_step(h, a, false));
  hipStream_t st = agx::as_stream(stream);
  const int B = a->B, G = h->G, ldx = h->ldx;
  AGX_TRY(gen_input(h, st, a->z, a->cat, a->num, B, B));
  AGX_TRY(gen_layers(h, st, B, h->X, ldx, nullptr, 0, true));                                  // :244 (training=True)
  if (a->aug_mask) AGX_TRY(agx_aug_add(st, h->X, ldx, a->aug_mask, a->aug_noise_fake, G, 0, B, G));  // :247
  AGX_TRY(disc_cov_columns(h, st, a->cat, a->num, B, B));
  return gen_core(h, st, a->cat, B, a->B_global, 0, a->losses);
}


// ---------------------------------------------------------------------------
// one loop iteration as a hipGraph
// ---------------------------------------------------------------------------
struct agx_iter_graph {
  agx_cgan *h = nullptr;
  int nseg = 0;
  hipGraph_t graph[AGX_MAX_ITER_SEGS] = {};
  hipGraphExec_t exec[AGX_MAX_ITER_SEGS] = {};
  int B = 0, nb_critic = 0;
  int64_t draws = 0;
  bool snap = false;   // the last step snapshots the loss log to the address staged per launch
  bool adam = false;
  bool chainx = false; // real rows are staged into both stacked-input buffers (chain_input)
  const float *aug_mask = nullptr, *aug_noise_real = nullptr;   // augmentation of the staged real rows (trace constants)
  agx::LrTable lrs;    // Adam: step sizes for the next replay (agx_cgan_iter_set_lr)
  agx::StageDraws sd = {};   // one-graph iterations: the draws ride the staging launch in front of the graph (sd.on)
  int split = 0, B_global = 0;
  bool ff = false;     // recorded with factored fake rows (iter_ff): agx_cgan_dp_iteration's chunk launches follow
  // agx_cgan_dp_iteration: one event per row chunk of the large product + "exchange of the step done"
  hipEvent_t ev_chunk[AGX_MAX_DP_PIECES] = {};
  hipEvent_t ev_done = nullptr;
  bool have_events = false;
};

namespace {
inline uint64_t r4u(uint64_t n) { return (n + 3) & ~(uint64_t)3; }

// One-GPU RMSprop iteration (with or without augmentation) whose critic products carry the flat update: the
// layer-1 product of step k also assembles step k+1's stacked input in the OTHER stacked-input buffer
// (generated rows + covariate columns from the freshly updated tables), so only step 0 keeps its
// stack_rows launch.  Real rows are staged into both buffers.  Must agree between the stage call and
// record_iteration: both ask here.
bool chain_input(const agx_cgan *h, const agx_iter_args_t *a) {
  static const bool off = getenv("AGX_NO_CHAIN_INPUT") != nullptr;
  const agx_mlp_desc_t &dm = h->cfg.disc;
  if (off || a->split || a->opt_kind != 0 || a->nb_critic < 1) return false;
  if (getenv("AGX_NO_FUSED_OPT") || getenv("AGX_NO_FUSED_REST")) return false;
  if (h->ldsx % 4 != 0 || h->ldx % 4 != 0) return false;
  int64_t emb_end = 0;
  for (int i = 0; i < dm.emb.n_cat; ++i)
    emb_end = std::max<int64_t>(emb_end, dm.emb.table_off[i] + (int64_t)dm.emb.vocab[i] * dm.emb.edim[i]);
  return emb_end <= 1024 && emb_end <= dm.w_off[0];
}

// Factored fake rows (critic_core / gen_core `ff`): iterations without augmentation at batches that take the Gram form
// and the factored generator step.  The generator's output layer is linear, so neither the critic steps nor the generator
// step need the generated rows themselves: the generator runs its hidden layers only, and [W_out; b_out] rides in front
// of the stacked critic input (agx_cgan::HD).  One-GPU and data-parallel segments alike (the layer-1 weight-gradient
// product is linear in each rank's own rows).  AGX_NO_FF (or AGX_NO_GRAM): off.
bool iter_ff(const agx_cgan *h, const agx_iter_args_t *a) {
  static const bool ff_off = getenv("AGX_NO_FF") != nullptr;
  return !ff_off && a->nb_critic > 0 && gram_ok(h, a->B) && gen_fact_ok(h, a->B) && !a->aug_mask_disc && !a->aug_mask_gen;
}

// Records segment `seg` of the iteration (seg < 0: the whole iteration).  Segments end where a
// data-parallel run all-reduces a gradient buffer:
//   0: draws, generator forward for all steps, critic gradients of step 0
//   k (1..nbc-1): critic RMSprop, critic gradients of step k
//   nbc: critic RMSprop, generator gradients          nbc+1: generator RMSprop
// (nbc == 0: segment 0 holds the generator gradients).
// split == 2: the segments hold NO optimizer launch (nbc+1 segments): the caller reduce-scatters the
// gradient buffer, updates its 1/R shard of the parameters and all-gathers them between segments.
// Where the iteration's draws go when they can ride the staging launch (one-GPU iterations, iter_stage(..., sd)):
// sd_out (capture): packed for the launch that precedes every replay, nothing recorded; launch (eager): the staging
// launch itself is issued here, with the draws when they fit one launch and without otherwise.
struct StageHook {
  agx::StageDraws *sd_out = nullptr;
  bool launch = false;
  const float *x = nullptr;
  int ldx = 0;
  const int32_t *cat = nullptr;
  const float *num = nullptr;
  uint64_t rng_offset = 0;
  float *losses_out = nullptr;
  const agx::LrTable *lrs = nullptr;
};

int record_iteration(agx_cgan *h, hipStream_t st, const agx_iter_args_t *a, int seg, int64_t *draws,
                     const StageHook *hook = nullptr) {
  const int B = a->B, Z = h->cfg.z_dim, nbc = a->nb_critic, S = nbc + 1, G = h->G;
  const int Bg = a->B_global > 0 ? a->B_global : B, lo = a->row_lo;
  const agx_mlp_desc_t &dm = h->cfg.disc;
  auto in_seg = [&](int s) { return seg < 0 || seg == s; };
  // One-GPU iteration (no all-reduce between gradient and update): the two big kernels' RMSprop
  // updates happen in the epilogue of their weight-gradient products - the 18.6 MB gradient is never
  // written or re-read - and the optimizer launch covers the rest of the flat buffer.
  const agx_mlp_desc_t &gm0 = h->cfg.gen;
  const int Lg0 = gm0.n_layers - 1;
  const int64_t dhole_lo = dm.w_off[0], dhole_hi = dm.w_off[0] + (int64_t)G * dm.ldw[0];
  const int64_t ghole_lo = gm0.w_off[Lg0], ghole_hi = gm0.w_off[Lg0] + (int64_t)gm0.dims[Lg0] * gm0.ldw[Lg0];
  const bool adam = a->opt_kind == 1;
  const bool fuse = seg < 0 && !a->split && B < 128 && !adam && !getenv("AGX_NO_FUSED_OPT");
  // from 128 rows on both big products fuse as well (+0.7 % at 512 rows)
  const bool fuse_big = seg < 0 && !a->split && B >= 128 && !adam && !getenv("AGX_NO_FUSED_OPT");
  const bool fuse_gbig = fuse_big;
  auto can_fuse_d = [&](int K) {
    return (fuse || fuse_big) && dhole_hi % 4 == 0 && dhole_hi <= (dm.n_flat & ~(int64_t)3) &&
           agx::gemm_can_fuse_opt(dm.dims[0], dm.dims[1], K, h->skws_floats);
  };
  bool fuse_d = can_fuse_d(dw1_rows(h, B));
  const bool fuse_g = (fuse || fuse_gbig) && ghole_hi % 4 == 0 && ghole_hi <= (gm0.n_flat & ~(int64_t)3) &&
                      agx::gemm_can_fuse_opt(gm0.dims[Lg0], gm0.dims[Lg0 + 1], gen_dw_rows(h, B), h->skws_floats);
  agx::FusedOpt dfo{h->dp + dhole_lo, a->disc_rms + dhole_lo, G, a->lr_disc, a->rho, a->eps};
  agx::FusedOpt gfo{h->gp + ghole_lo, a->gen_rms + ghole_lo, gm0.dims[Lg0], a->lr_gen, a->rho, a->eps};
  // ... and the same launch updates the rest of the flat buffer:
  // the critic's product then takes ALL rows of W_1, and no optimizer launch is left in the step
  const bool rest = !getenv("AGX_NO_FUSED_REST");
  const int64_t dfull_hi = dm.w_off[0] + (int64_t)dm.dims[0] * dm.ldw[0];
  const bool rest_d = rest && dfull_hi % 4 == 0 && dfull_hi <= (dm.n_flat & ~(int64_t)3);
  const bool ff = iter_ff(h, a);
  if (ff) fuse_d = can_fuse_d(h->HD + B);   // (the layer-1 weight-gradient product then runs over HD + B rows)
  if (rest_d) {
    dfo.rows = dm.dims[0];
    dfo.rest_w = h->dp; dfo.rest_g = h->dg; dfo.rest_s1 = a->disc_rms;
    dfo.rest_n = dm.n_flat; dfo.rest_hole_lo = dhole_lo; dfo.rest_hole_hi = dfull_hi;
  }
  if (rest) {
    gfo.rest_w = h->gp; gfo.rest_g = h->gg; gfo.rest_s1 = a->gen_rms;
    gfo.rest_n = gm0.n_flat; gfo.rest_hole_lo = ghole_lo; gfo.rest_hole_hi = ghole_hi;
  }
  const bool snap = seg < 0 && !a->split && !a->disc_losses && !a->gen_losses;
  const agx::LossCopy lcopy{h->loss_slot, h->losslog, (nbc + 1) * 4};
  uint64_t sub = 0;
  // Step 0's stacked input without a copy launch (chained input assembly only): the generator's output product stores
  // its first B rows ALSO into the fake rows of buffer 0, the draws launch writes the covariate columns.
  const bool chainx0 = seg < 0 && !ff && fuse_d && rest_d && chain_input(h, a);
  // (small batches only: at 512 rows the second copy of the rows costs the product more than the launch it saves,
  //  186.4 k against 189.6 k samples/s)
  const bool fold0 = chainx0 && nbc > 0 && !a->aug_mask_disc && B <= 64 && 2 * nbc + 1 <= AGX_MAX_RNG_SEGS && !getenv("AGX_NO_DRAW_FUSE") &&
                     agx::gemm_can_dual(S * B, gm0.dims[Lg0 + 1], gm0.dims[Lg0], h->skws_floats, 0);
  auto stage_now = [&](const agx::StageDraws *sd) -> int {
    const int ncat = gm0.emb.n_cat, nnum = gm0.n_num;
    return agx::iter_stage(st, hook->x, hook->ldx, h->Xbuf[0], (chain_input(h, a) && !ff) ? h->Xbuf[1] : nullptr, h->ldx, h->G, B,
                           hook->cat, h->scat, B * ncat, hook->num, h->snum, B * nnum, h->rngpos, hook->rng_offset,
                           h->loss_slot, hook->losses_out, hook->lrs, h->lrdev, a->aug_mask_disc, a->aug_noise_real, h->G, sd);
  };
  // draws + generator input in one launch (z straight into the [z | num | emb] rows) when the segment list fits
  static const bool draw_fuse_on = getenv("AGX_NO_DRAW_FUSE") == nullptr;
  const bool draw_fuse = draw_fuse_on && 2 * nbc + 1 <= AGX_MAX_RNG_SEGS;
  if (draw_fuse) {
    agx::IterDraws dr;
    dr.n = 0;
    for (int k = 0; k <= nbc; ++k) {
      dr.seg[dr.n++] = agx::DrawSeg{h->gin0 + (size_t)k * B * h->ldg0, (int64_t)B * Z, sub, (int64_t)lo * Z, 1, Z, h->ldg0};
      sub += r4u((uint64_t)Bg * Z);
      if (k < nbc) {
        dr.seg[dr.n++] = agx::DrawSeg{h->abuf + (size_t)k * B, (int64_t)B, sub, (int64_t)lo, 0, 0, 0};
        sub += r4u((uint64_t)Bg);
      }
    }
    // (with the chained input assembly the same launch writes the covariate columns of step 0's [real; fake] rows
    //  from the critic's tables, final since the last iteration; see fold0 below)
    agx::CriticCov xc;
    if (fold0) {
      xc.e = &dm.emb; xc.tables = h->dp; xc.X = h->Xbuf[0]; xc.ldx = h->ldx; xc.col0 = G; xc.rows = 2 * B;
    }
    static const bool merge_draws = getenv("AGX_NO_STAGE_DRAWS") == nullptr;   // (cross-check knob, tests/test_gpu_knobs.py)
    if (hook && seg < 0 && merge_draws) {
      agx::StageDraws local;
      agx::StageDraws *sd = hook->sd_out ? hook->sd_out : &local;
      AGX_TRY(agx::iter_draws_desc(a->seed, dr, &gm0.emb, h->gp, gm0.n_num, gm0.emb.n_cat, B, S * B, Z, h->gin0, h->ldg0,
                                   fold0 ? &xc : nullptr, sd));
      if (hook->launch) AGX_TRY(stage_now(sd));
    } else {
      if (hook && hook->launch) AGX_TRY(stage_now(nullptr));
      if (in_seg(0))
        AGX_TRY(agx::iter_draws(st, a->seed, h->rngpos, dr, &gm0.emb, h->gp, h->snum, gm0.n_num, h->scat, gm0.emb.n_cat, B,
                                S * B, Z, h->gin0, h->ldg0, fold0 ? &xc : nullptr));
    }
  } else {
    if (hook && hook->launch) AGX_TRY(stage_now(nullptr));
    // every draw of the iteration in one launch (per 16 segments), at the stream positions the eager
    // entry points use for the GLOBAL batch (z_k then alpha_k per critic step, then the generator
    // step's z); a rank keeps rows [lo, lo+B) of each draw
    agx_rng_seg_t segs[AGX_MAX_RNG_SEGS];
    int ns = 0;
    auto flush = [&]() -> int {
      int rc = (ns && in_seg(0)) ? agx_rng_multi(st, a->seed, h->rngpos, ns, segs) : AGX_OK;
      ns = 0;
      return rc;
    };
    for (int k = 0; k <= nbc; ++k) {
      if (ns + 2 > AGX_MAX_RNG_SEGS) AGX_TRY(flush());
      segs[ns++] = agx_rng_seg_t{h->zbuf + (size_t)k * B * Z, (int64_t)B * Z, sub, 1, (int64_t)lo * Z};
      sub += r4u((uint64_t)Bg * Z);
      if (k < nbc) {
        segs[ns++] = agx_rng_seg_t{h->abuf + (size_t)k * B, (int64_t)B, sub, 0, (int64_t)lo};
        sub += r4u((uint64_t)Bg);
      }
    }
    AGX_TRY(flush());
  }
  *draws = (int64_t)sub;
  // chained input assembly (see chain_input): needs the critic product to carry the flat update; steps then
  // alternate between the two stacked-input buffers
  const bool chainx = seg < 0 && !ff && fuse_d && rest_d && chain_input(h, a);
  float *Xb[2] = {h->Xbuf[0], chainx ? h->Xbuf[1] : h->Xbuf[0]};
  struct Restore { agx_cgan *h; ~Restore() { h->X = h->Xbuf[0]; h->use_sk = false; h->skip_big = false; h->ff = false; } } restore{h};
  h->ff = ff;
  h->use_sk = a->split >= 2;
  h->skip_big = a->split == 3;
  // fake rows of step k (k == nbc: the generator step's rows [0,B)) into buffer k&1; with_cov: also
  // the covariate columns of the step's rows from the CURRENT embedding tables (same launch)
  auto place_fake = [&](int k, hipStream_t s, bool with_cov = false) -> int {
    const bool gstep = (k == nbc);
    return agx_stack_rows(s, &dm.emb, h->dp, h->F + (size_t)k * B * h->ldsx, h->ldsx, gstep ? 0 : B, B, h->snum,
                          dm.n_num, dm.n_num, h->scat, dm.emb.n_cat, B, with_cov ? (gstep ? B : 2 * B) : 0, G,
                          Xb[k & 1], h->ldx,
                          gstep ? a->aug_mask_gen : a->aug_mask_disc, gstep ? a->aug_noise_gen : a->aug_noise_fake,
                          gstep ? G : 0);
  };
  if (in_seg(0)) {
    // ONE generator forward for all nb_critic+1 z batches (generator weights are constant until the
    // iteration's last update): fake rows of step k = F[k*B : (k+1)*B]
    if (!draw_fuse) AGX_TRY(gen_input(h, st, h->zbuf, h->scat, h->snum, S * B, B));
    const agx::DualOut du{h->Xbuf[0] + (size_t)B * h->ldx, h->ldx, B};
    if (ff) {
      // hidden layers only; [W_out; b_out; unit rows; 0] in front of the stacked input, zero padding rows in front of D_1
      AGX_TRY(gen_layers(h, st, S * B, h->F, h->ldsx, nullptr, B, false, true));
      AGX_TRY(agx::ff_stage(st, h->gp + gm0.w_off[Lg0], gm0.ldw[Lg0], h->gp + gm0.b_off[Lg0], gm0.dims[Lg0], G, h->ncov, h->he,
                            h->Xbuf[0] - (size_t)h->HD * h->ldx, h->ldx, h->hxp));
      if (h->hxp > h->hx)
        AGX_HIP(hipMemsetAsync(h->D[1] - (size_t)(h->HD - h->hx) * h->ldh[1], 0, (size_t)(h->hxp - h->hx) * h->ldh[1] * sizeof(float), st));
    } else
    AGX_TRY(gen_layers(h, st, S * B, h->F, h->ldsx, fold0 ? &du : nullptr, B));
    // real rows were staged into buffer 0 rows [0,B) (agx_cgan_iter_stage) and stay there for all
    // critic steps (augmentation noise is a trace constant, :210); buffer 1 gets a copy
    // (augmentation noise of the real rows, :210: added by the stage launch that copies them in)
  }
  for (int k = 0; k < nbc; ++k) {
    h->X = Xb[k & 1];
    if (in_seg(k)) {
      // covariate columns of [real; fake] from the CURRENT (just updated) embedding tables
      h->ff_row0 = k * B;
      if (ff) {}   // (critic_core refreshes the real rows' covariate columns itself: ff_hx)
      else
      if (!chainx || (k == 0 && !fold0)) AGX_TRY(place_fake(k, st, true));
      agx::FusedOpt dk = dfo;
      if (chainx) {   // step k+1's rows (the generator step's rows [0,B) when k+1 == nbc) into the other buffer
        const bool gnext = (k + 1 == nbc);
        float *Xn = Xb[(k + 1) & 1];
        dk.nx_src = h->F + (size_t)(k + 1) * B * h->ldsx; dk.nx_lds = h->ldsx;
        dk.nx_dst = Xn + (gnext ? 0 : (size_t)B * h->ldx); dk.nx_ldd = h->ldx;
        dk.nx_rows = B; dk.nx_G = G;
        dk.nx_cov = Xn; dk.nx_cov_rows = gnext ? B : 2 * B; dk.nx_B = B;
        dk.nx_num = h->snum; dk.nx_nnum = dm.n_num; dk.nx_cat = h->scat; dk.nx_ldcat = dm.emb.n_cat;
        dk.nx_emb = dm.emb;
        // augmentation of the copied rows, as place_fake() would apply it (:209 one noise row for all, :247 a row each)
        dk.nx_mask = gnext ? a->aug_mask_gen : a->aug_mask_disc;
        dk.nx_noise = gnext ? a->aug_noise_gen : a->aug_noise_fake;
        dk.nx_ldn = gnext ? G : 0;
      }
      AGX_TRY(critic_core(h, st, h->scat, B, Bg, h->abuf + (size_t)k * B, a->gp_weight,
                          a->disc_losses ? a->disc_losses : h->losslog + 4 * k, fuse_d ? &dk : nullptr));
    }
    if (in_seg(k + 1) && a->split < 2) {
      if (adam) AGX_TRY(agx::opt_adam_dev_lr(st, h->dp, h->dg, a->disc_m, a->disc_v, dm.n_flat, h->lrdev + k, a->beta1, a->beta2, a->eps));
      else if (fuse_d && rest_d) {}   // updated by the layer-1 product's launch
      else if (fuse_d) AGX_TRY(agx::opt_rmsprop_hole(st, h->dp, h->dg, a->disc_rms, dm.n_flat, dhole_lo, dhole_hi, a->lr_disc, a->rho, a->eps));
      else AGX_TRY(agx_opt_rmsprop(st, h->dp, h->dg, a->disc_rms, h->cfg.disc.n_flat, a->lr_disc, a->rho, a->eps, 1.f));
    }
  }
  h->X = Xb[nbc & 1];
  if (in_seg(nbc)) {
    if (ff) {}   // (gen_core: ff_hx)
    else
    if (!chainx) AGX_TRY(place_fake(nbc, st, true));   // else: assembled by the last critic step's layer-1 product
    AGX_TRY(gen_core(h, st, h->scat, B, Bg, nbc * B, a->gen_losses ? a->gen_losses : h->losslog + 4 * nbc,
                     fuse_g ? &gfo : nullptr, snap ? &lcopy : nullptr));
  }
  if (in_seg(nbc + 1) && a->split < 2) {
    if (adam) AGX_TRY(agx::opt_adam_dev_lr(st, h->gp, h->gg, a->gen_m, a->gen_v, gm0.n_flat, h->lrdev + nbc, a->beta1, a->beta2, a->eps));
    else if (fuse_g && rest) {}     // updated by the output-layer product's launch
    else if (fuse_g) AGX_TRY(agx::opt_rmsprop_hole(st, h->gp, h->gg, a->gen_rms, gm0.n_flat, ghole_lo, ghole_hi, a->lr_gen, a->rho, a->eps));
    else AGX_TRY(agx_opt_rmsprop(st, h->gp, h->gg, a->gen_rms, h->cfg.gen.n_flat, a->lr_gen, a->rho, a->eps, 1.f));
  }
  return AGX_OK;
}
}  // namespace

extern "C" int agx_cgan_iter_capture(agx_cgan_t *h, void *stream, const agx_iter_args_t *a,
                                     agx_iter_graph_t **out) {
  AGX_CHECK_ARG(h && a && out, "agx_cgan_iter_capture: null argument");
  if (!h->bound) { agx::set_error("agx_cgan_iter_capture: not bound"); return AGX_E_STATE; }
  AGX_CHECK_ARG(a->B > 0 && a->B <= h->Bm, "agx_cgan_iter_capture: batch %d outside (0, %d]", a->B, h->Bm);
  AGX_CHECK_ARG(a->nb_critic >= 0 && a->nb_critic < h->S, "agx_cgan_iter_capture: nb_critic %d needs max_steps > %d (have %d)", a->nb_critic, a->nb_critic, h->S);
  AGX_CHECK_ARG(a->nb_critic + 2 <= AGX_MAX_ITER_SEGS, "agx_cgan_iter_capture: nb_critic %d too large", a->nb_critic);
  AGX_CHECK_ARG(a->B_global == 0 || (a->B_global >= a->B && a->row_lo >= 0 && a->row_lo + a->B <= a->B_global),
                "agx_cgan_iter_capture: rows [%d, %d) outside the global batch %d", a->row_lo, a->row_lo + a->B, a->B_global);
  AGX_CHECK_ARG(a->B_global != 0 || a->row_lo == 0, "agx_cgan_iter_capture: row_lo without B_global");
  AGX_CHECK_ARG(a->opt_kind == 0 || a->opt_kind == 1, "agx_cgan_iter_capture: opt_kind %d", a->opt_kind);
  AGX_CHECK_ARG(a->split >= 0 && a->split <= 3, "agx_cgan_iter_capture: split %d", a->split);
  AGX_CHECK_ARG(a->split >= 2 || (a->opt_kind == 1 ? (a->disc_m && a->disc_v && a->gen_m && a->gen_v) : (a->disc_rms && a->gen_rms)),
                "agx_cgan_iter_capture: optimizer slots missing");
  AGX_CHECK_ARG(!a->aug_mask_disc || (a->aug_noise_fake && a->aug_noise_real), "agx_cgan_iter_capture: critic aug noise missing");
  AGX_CHECK_ARG(!a->aug_mask_gen || a->aug_noise_gen, "agx_cgan_iter_capture: generator aug noise missing");
  AGX_CHECK_ARG(stream != nullptr, "agx_cgan_iter_capture: capture needs a non-default stream");
  AGX_CHECK_ARG(!h->cfg.gen.bn, "agx_cgan_iter_capture: batch-norm generators run through the eager step functions only");
  if (agx_prof_enabled()) {
    agx::set_error("agx_cgan_iter_capture: event brackets cannot be captured; call agx_prof_enable(-2) first");
    return AGX_E_STATE;
  }
  hipStream_t st = agx::as_stream(stream);
  agx_iter_graph *g = new (std::nothrow) agx_iter_graph();
  AGX_CHECK_ARG(g, "agx_cgan_iter_capture: out of host memory");
  g->h = h; g->B = a->B; g->nb_critic = a->nb_critic;
  g->snap = !a->split && !a->disc_losses && !a->gen_losses;
  g->adam = a->opt_kind == 1 && a->split < 2;   // split >= 2: the caller's optimizer launches take their step size from the host
  g->split = a->split;
  g->ff = iter_ff(h, a);
  g->B_global = a->B_global > 0 ? a->B_global : a->B;
  g->chainx = chain_input(h, a) && !iter_ff(h, a);   // (factored fake rows: one stacked-input buffer)
  g->aug_mask = a->aug_mask_disc;
  g->aug_noise_real = a->aug_noise_real;
  g->lrs.n = 0;
  const int nseg = a->split >= 2 ? a->nb_critic + 1 : (a->split ? a->nb_critic + 2 : 1);
  for (int s = 0; s < nseg; ++s) {
    hipError_t e = hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed);
    if (e != hipSuccess) {
      agx_cgan_iter_destroy(g);
      agx::set_error("hipStreamBeginCapture failed: %s", hipGetErrorString(e));
      return AGX_E_HIP;
    }
    StageHook hook;
    hook.sd_out = &g->sd;
    int rc = record_iteration(h, st, a, a->split ? s : -1, &g->draws, !a->split ? &hook : nullptr);
    e = hipStreamEndCapture(st, &g->graph[s]);
    g->nseg = s + 1;
    if (rc != AGX_OK) {
      agx_cgan_iter_destroy(g);
      return rc;
    }
    if (e != hipSuccess || !g->graph[s]) {
      agx_cgan_iter_destroy(g);
      agx::set_error("hipStreamEndCapture failed: %s", hipGetErrorString(e));
      return AGX_E_HIP;
    }
    e = hipGraphInstantiate(&g->exec[s], g->graph[s], nullptr, nullptr, 0);
    if (e != hipSuccess) {
      agx_cgan_iter_destroy(g);
      agx::set_error("hipGraphInstantiate failed: %s", hipGetErrorString(e));
      return AGX_E_HIP;
    }
  }
  *out = g;
  return AGX_OK;
}

extern "C" int agx_cgan_iter_stage(agx_iter_graph_t *g, void *stream, const float *x, int32_t ldx,
                                   const int32_t *cat, const float *num, uint64_t rng_offset) {
  AGX_CHECK_ARG(g && g->nseg > 0 && x, "agx_cgan_iter_stage: null argument");
  agx_cgan *h = g->h;
  AGX_CHECK_ARG(ldx >= h->G, "agx_cgan_iter_stage: ldx < x_dim");
  hipStream_t st = agx::as_stream(stream);
  const int B = g->B, ncat = h->cfg.gen.emb.n_cat, nnum = h->cfg.gen.n_num;
  AGX_CHECK_ARG((ncat == 0 || cat) && (nnum == 0 || num), "agx_cgan_iter_stage: covariates missing");
  // one launch: real rows straight into rows [0,B) of stacked-input buffer 0, covariates, RNG position
  if (g->adam && g->lrs.n != g->nb_critic + 1) {
    agx::set_error("agx_cgan_iter_stage: Adam iteration needs agx_cgan_iter_set_lr(%d values) first", g->nb_critic + 1);
    return AGX_E_STATE;
  }
  return agx::iter_stage(st, x, ldx, h->Xbuf[0], g->chainx ? h->Xbuf[1] : nullptr, h->ldx, h->G, B, cat, h->scat, B * ncat, num, h->snum, B * nnum,
                         h->rngpos, rng_offset, h->loss_slot, nullptr, g->adam ? &g->lrs : nullptr, h->lrdev,
                         g->aug_mask, g->aug_noise_real, h->G, g->sd.on ? &g->sd : nullptr);
}

extern "C" int32_t agx_cgan_iter_num_segs(const agx_iter_graph_t *g) { return g ? g->nseg : 0; }

extern "C" int agx_cgan_iter_launch_seg(agx_iter_graph_t *g, void *stream, int32_t seg) {
  AGX_CHECK_ARG(g && seg >= 0 && seg < g->nseg && g->exec[seg], "agx_cgan_iter_launch_seg: bad segment");
  AGX_HIP(hipGraphLaunch(g->exec[seg], agx::as_stream(stream)));
  return AGX_OK;
}

extern "C" int agx_cgan_iter_launch(agx_iter_graph_t *g, void *stream, const float *x, int32_t ldx,
                                    const int32_t *cat, const float *num, uint64_t rng_offset,
                                    float *losses_out) {
  AGX_CHECK_ARG(g && g->nseg > 0 && x, "agx_cgan_iter_launch: null argument");
  hipStream_t st = agx::as_stream(stream);
  if (g->snap) {
    agx_cgan *h = g->h;
    AGX_CHECK_ARG(ldx >= h->G, "agx_cgan_iter_launch: ldx < x_dim");
    const int B = g->B, ncat = h->cfg.gen.emb.n_cat, nnum = h->cfg.gen.n_num;
    AGX_CHECK_ARG((ncat == 0 || cat) && (nnum == 0 || num), "agx_cgan_iter_launch: covariates missing");
    // the graph's last small-gradient launch copies the loss log to losses_out (address staged here)
    if (g->adam && g->lrs.n != g->nb_critic + 1) {
      agx::set_error("agx_cgan_iter_launch: Adam iteration needs agx_cgan_iter_set_lr(%d values) first", g->nb_critic + 1);
      return AGX_E_STATE;
    }
    AGX_TRY(agx::iter_stage(st, x, ldx, h->Xbuf[0], g->chainx ? h->Xbuf[1] : nullptr, h->ldx, h->G, B, cat, h->scat, B * ncat, num, h->snum,
                            B * nnum, h->rngpos, rng_offset, h->loss_slot, losses_out, g->adam ? &g->lrs : nullptr,
                            h->lrdev, g->aug_mask, g->aug_noise_real, h->G, g->sd.on ? &g->sd : nullptr));
    for (int s = 0; s < g->nseg; ++s) AGX_HIP(hipGraphLaunch(g->exec[s], st));
    return AGX_OK;
  }
  AGX_TRY(agx_cgan_iter_stage(g, stream, x, ldx, cat, num, rng_offset));
  for (int s = 0; s < g->nseg; ++s) AGX_HIP(hipGraphLaunch(g->exec[s], st));
  if (losses_out)
    AGX_HIP(hipMemcpyAsync(losses_out, g->h->losslog, (size_t)(g->nb_critic + 1) * 4 * sizeof(float),
                           hipMemcpyDeviceToDevice, st));
  return AGX_OK;
}

extern "C" int64_t agx_cgan_iter_rng_draws(const agx_iter_graph_t *g) { return g ? g->draws : 0; }

extern "C" int agx_cgan_iter_set_lr(agx_iter_graph_t *g, const float *lr_t_host, int32_t n) {
  AGX_CHECK_ARG(g && lr_t_host && n == g->nb_critic + 1 && n <= 64, "agx_cgan_iter_set_lr: need nb_critic + 1 values");
  g->lrs.n = n;
  for (int i = 0; i < n; ++i) g->lrs.v[i] = lr_t_host[i];
  return AGX_OK;
}

extern "C" int agx_cgan_iter_eager(agx_cgan_t *h, void *stream, const agx_iter_args_t *a, const float *x,
                                   int32_t ldx, const int32_t *cat, const float *num, uint64_t rng_offset,
                                   float *losses_out, int64_t *draws_out) {
  AGX_CHECK_ARG(h && a && x, "agx_cgan_iter_eager: null argument");
  if (!h->bound) { agx::set_error("agx_cgan_iter_eager: not bound"); return AGX_E_STATE; }
  AGX_CHECK_ARG(a->B > 0 && a->B <= h->Bm && ldx >= h->G, "agx_cgan_iter_eager: batch %d outside (0, %d] or ldx < x_dim", a->B, h->Bm);
  AGX_CHECK_ARG(a->nb_critic >= 0 && a->nb_critic < h->S, "agx_cgan_iter_eager: nb_critic %d needs max_steps > %d", a->nb_critic, a->nb_critic);
  AGX_CHECK_ARG(!a->split, "agx_cgan_iter_eager: one-GPU form only");
  AGX_CHECK_ARG(!h->cfg.gen.bn, "agx_cgan_iter_eager: batch-norm generators run through the eager step functions only");
  AGX_CHECK_ARG(a->opt_kind == 1 ? (a->disc_m && a->disc_v && a->gen_m && a->gen_v && a->adam_lr_t_host)
                                 : (a->opt_kind == 0 && a->disc_rms && a->gen_rms),
                "agx_cgan_iter_eager: optimizer slots (and adam_lr_t_host for Adam) required");
  agx::LrTable lrs;
  lrs.n = 0;
  if (a->opt_kind == 1) {
    lrs.n = a->nb_critic + 1;
    for (int i = 0; i < lrs.n; ++i) lrs.v[i] = a->adam_lr_t_host[i];
  }
  const int ncat = h->cfg.gen.emb.n_cat, nnum = h->cfg.gen.n_num;
  AGX_CHECK_ARG((ncat == 0 || cat) && (nnum == 0 || num), "agx_cgan_iter_eager: covariates missing");
  hipStream_t st = agx::as_stream(stream);
  const bool snap = !a->disc_losses && !a->gen_losses;
  // (the staging launch is issued inside record_iteration, where the iteration's draws join it)
  StageHook hook;
  hook.launch = true;
  hook.x = x; hook.ldx = ldx; hook.cat = cat; hook.num = num; hook.rng_offset = rng_offset;
  hook.losses_out = snap ? losses_out : nullptr;
  hook.lrs = lrs.n ? &lrs : nullptr;
  int64_t draws = 0;
  AGX_TRY(record_iteration(h, st, a, -1, &draws, &hook));
  if (draws_out) *draws_out = draws;
  return AGX_OK;
}

namespace {
// rows [r0, r1) of a step's largest weight-gradient product (critic: dW_1 = [a_0(real); a_0(fake); e_0]^T D_1 over the
// 3B stacked rows; generator: dW_out = h_L^T dX over its B rows), same operands as critic_core / gen_core use
int big_grad_chunk(agx_cgan *h, hipStream_t st, bool gen, int r0, int r1, int B, int gen_row0, bool ff) {
  const agx_mlp_desc_t &m = gen ? h->cfg.gen : h->cfg.disc;
  const int l = gen ? m.n_layers - 1 : 0;
  const bool gfact = gen && gen_fact_ok(h, B);   // (gen_core: dW_out = Q-as-TN . W_g^T, W_g^T in the rows the input gradient would take)
  // (ff, critic_core / gen_core: the critic's product starts HD rows in front of X and D_1; W_g^T sits right in front of X)
  const float *in = gen ? (gfact ? h->genQ : h->gh[l] + (size_t)gen_row0 * h->ldgh[l]) : h->Xbuf[0] - (ff ? (size_t)h->HD * h->ldx : 0);
  const int ldin = gen ? (gfact ? h->ldpt : h->ldgh[l]) : h->ldx;
  const float *dl = gen ? (ff ? h->Xbuf[0] - (size_t)h->cfg.disc.dims[1] * h->ldx : h->Xbuf[0] + (size_t)B * h->ldx)
                        : h->D[1] - (ff ? (size_t)h->HD * h->ldh[1] : 0);
  const int ldd = gen ? h->ldx : h->ldh[1];
  float *out = (gen ? h->gg : h->dg) + m.w_off[l];
  Gemm g(h, in + r0, ldin, 1, dl, ldd, 0, out + (size_t)r0 * m.ldw[l], m.ldw[l], r1 - r0, m.dims[l + 1], gen ? gen_dw_rows(h, B) : (ff ? h->HD + B : dw1_rows(h, B)));
  return g.id(gen ? AGX_K_GEN_DW_OUT : AGX_K_DISC_DW_IN).sk(true).run(st);
}

int check_dp_net(const agx_dp_net_t &n, const agx_mlp_desc_t &m, int l, int nranks, int opt_kind, const char *name) {
  AGX_CHECK_ARG(n.n_pieces >= 1 && n.n_pieces <= AGX_MAX_DP_PIECES, "%s: n_pieces %d", name, n.n_pieces);
  AGX_CHECK_ARG(n.slot0 && (opt_kind == 0 || n.slot1), "%s: optimizer slots missing", name);
  AGX_CHECK_ARG(n.rows[0] == 0 && n.rows[n.n_pieces] == m.dims[l], "%s: row chunks must cover [0, %d)", name, m.dims[l]);
  const int64_t big_lo = m.w_off[l];
  int64_t at = 0;
  for (int c = 0; c < n.n_pieces; ++c) {
    AGX_CHECK_ARG(n.rows[c + 1] > n.rows[c] && (n.rows[c] % 4) == 0, "%s: row chunk %d = [%d, %d)", name, c, n.rows[c], n.rows[c + 1]);
    AGX_CHECK_ARG(n.start[c] == at && n.per[c] > 0 && (n.start[c] % 4) == 0 && (n.per[c] % 4) == 0,
                  "%s: piece %d = (%lld, %lld) does not continue at %lld / is not 16-B aligned", name, c, (long long)n.start[c],
                  (long long)n.per[c], (long long)at);
    at += n.per[c] * nranks;
    // complete once chunk c is: nothing of it lies behind the rows computed so far (the last piece: everything)
    AGX_CHECK_ARG(c == n.n_pieces - 1 || at <= big_lo + (int64_t)n.rows[c + 1] * m.ldw[l],
                  "%s: piece %d ends at %lld, behind row %d of the large kernel", name, c, (long long)at, n.rows[c + 1]);
  }
  AGX_CHECK_ARG(at >= m.n_flat + 64, "%s: pieces cover %lld floats < n_flat + 64 loss slots", name, (long long)at);
  return AGX_OK;
}
}  // namespace

extern "C" int agx_cgan_dp_iteration(agx_iter_graph_t *g, void *stream, void *comm_stream, agx_comm_t *comm,
                                     const agx_dp_args_t *a, const float *x, int32_t ldx, const int32_t *cat,
                                     const float *num, uint64_t rng_offset) {
  AGX_CHECK_ARG(g && a && comm && x, "agx_cgan_dp_iteration: null argument");
  AGX_CHECK_ARG(g->split == 3 && g->nseg == g->nb_critic + 1, "agx_cgan_dp_iteration: the graph must be captured with split = 3");
  AGX_CHECK_ARG(comm_stream && stream != comm_stream, "agx_cgan_dp_iteration: the exchange needs a (non-default) stream of its own");
  AGX_CHECK_ARG(a->opt_kind == 0 || (a->opt_kind == 1 && a->lr_t_host), "agx_cgan_dp_iteration: opt_kind %d", a->opt_kind);
  agx_cgan *h = g->h;
  const int nranks = agx_comm_size(comm), rank = agx_comm_rank(comm);
  const int nbc = g->nb_critic, B = g->B;
  AGX_TRY(check_dp_net(a->disc, h->cfg.disc, 0, nranks, a->opt_kind, "agx_cgan_dp_iteration (critic)"));
  AGX_TRY(check_dp_net(a->gen, h->cfg.gen, h->cfg.gen.n_layers - 1, nranks, a->opt_kind, "agx_cgan_dp_iteration (generator)"));
  hipStream_t cs = agx::as_stream(stream), ms = agx::as_stream(comm_stream);
  if (!g->have_events) {
    for (int c = 0; c < AGX_MAX_DP_PIECES; ++c) AGX_HIP(hipEventCreateWithFlags(&g->ev_chunk[c], hipEventDisableTiming));
    AGX_HIP(hipEventCreateWithFlags(&g->ev_done, hipEventDisableTiming));
    g->have_events = true;
  }
  AGX_TRY(agx_cgan_iter_stage(g, stream, x, ldx, cat, num, rng_offset));
  for (int s = 0; s <= nbc; ++s) {
    const bool gen = s == nbc;
    const agx_dp_net_t &n = gen ? a->gen : a->disc;
    const agx_mlp_desc_t &m = gen ? h->cfg.gen : h->cfg.disc;
    float *params = gen ? h->gp : h->dp, *grads = gen ? h->gg : h->dg;
    if (s > 0) AGX_HIP(hipStreamWaitEvent(cs, g->ev_done, 0));   // the parameters of step s - 1 are in place
    AGX_HIP(hipGraphLaunch(g->exec[s], cs));
    for (int c = 0; c < n.n_pieces; ++c) {
      AGX_TRY(big_grad_chunk(h, cs, gen, n.rows[c], n.rows[c + 1], B, nbc * B, g->ff));
      AGX_HIP(hipEventRecord(g->ev_chunk[c], cs));
    }
    for (int c = 0; c < n.n_pieces; ++c) {
      const int64_t per = n.per[c], lo = n.start[c] + (int64_t)rank * per;
      AGX_HIP(hipStreamWaitEvent(ms, g->ev_chunk[c], 0));
      AGX_TRY(agx_comm_reduce_scatter_sum(comm, ms, grads + n.start[c], grads + lo, per));
      const int64_t n_upd = std::max<int64_t>(0, std::min<int64_t>(lo + per, m.n_flat) - lo);
      if (n_upd > 0) {
        if (a->opt_kind == 1)
          AGX_TRY(agx_opt_adam(ms, params + lo, grads + lo, n.slot0 + lo, n.slot1 + lo, n_upd, a->lr_t_host[s], a->beta1, a->beta2,
                               a->eps, 1.f));
        else
          AGX_TRY(agx_opt_rmsprop(ms, params + lo, grads + lo, n.slot0 + lo, n_upd, n.lr, a->rho, a->eps, 1.f));
      }
      if (lo + per > m.n_flat) {   // loss slots (+ padding): the reduced values themselves are what travels back
        const int64_t from = std::max<int64_t>(lo, m.n_flat);
        AGX_HIP(hipMemcpyAsync(params + from, grads + from, (size_t)(lo + per - from) * sizeof(float), hipMemcpyDeviceToDevice, ms));
      }
      AGX_TRY(agx_comm_all_gather(comm, ms, params + lo, params + n.start[c], per));
    }
    AGX_HIP(hipEventRecord(g->ev_done, ms));
  }
  AGX_HIP(hipStreamWaitEvent(cs, g->ev_done, 0));
  return AGX_OK;
}

extern "C" int agx_cgan_iter_destroy(agx_iter_graph_t *g) {
  if (g) {
    for (int s = 0; s < AGX_MAX_ITER_SEGS; ++s) {
      if (g->exec[s]) (void)hipGraphExecDestroy(g->exec[s]);
      if (g->graph[s]) (void)hipGraphDestroy(g->graph[s]);
    }
    for (int c = 0; c < AGX_MAX_DP_PIECES; ++c)
      if (g->ev_chunk[c]) (void)hipEventDestroy(g->ev_chunk[c]);
    if (g->ev_done) (void)hipEventDestroy(g->ev_done);
    delete g;
  }
  return AGX_OK;
}
