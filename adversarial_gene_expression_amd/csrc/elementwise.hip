// HBM-bound kernels of the cGAN train step: embedding gather/concat and its
// scatter-add gradient, WGAN-GP loss pieces, bias-gradient reductions, fused
// optimizers, Philox RNG.  Reference lines are cited per entry point in
// include/agx.h.
#include "agx_common.h"

#include <algorithm>

namespace {

using agx::wave_sum;

// ---------------------------------------------------------------------------
// embeddings
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) embed_concat_fwd_kernel(
    agx_embed_t e, const float *__restrict__ tables, const float *__restrict__ lead, int n_lead,
    int ld_lead, const float *__restrict__ num, int n_num, int ld_num,
    const int *__restrict__ cat, int ld_cat, int B, int row_mod, float *__restrict__ out, int ldo,
    int width) {
  const long long total = (long long)B * width;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int ob = (int)(idx / width);
    const int b = row_mod > 0 ? ob % row_mod : ob;
    int c = (int)(idx % width);
    float v = 0.f;
    if (c < n_lead) {
      v = lead[(size_t)b * ld_lead + c];
    } else if (c < n_lead + n_num) {
      v = num[(size_t)b * ld_num + (c - n_lead)];
    } else {
      c -= n_lead + n_num;
      for (int i = 0; i < e.n_cat; ++i) {
        if (c < e.edim[i]) {
          const int row = cat[(size_t)b * ld_cat + i];
          if (row >= 0 && row < e.vocab[i]) v = tables[e.table_off[i] + (size_t)row * e.edim[i] + c];
          break;
        }
        c -= e.edim[i];
      }
    }
    out[(size_t)ob * ldo + (int)(idx % width)] = v;
  }
}

// one block per table row (i, v); waves stride over the embedding columns,
// lanes over the batch; fixed summation order => deterministic
__global__ void __launch_bounds__(256) embed_bwd_kernel(agx_embed_t e, float *__restrict__ grad,
                                                        const float *__restrict__ dout, int ldd,
                                                        int col0, const int *__restrict__ cat,
                                                        int ld_cat, int B, int row_mod,
                                                        int accumulate) {
  int rowid = blockIdx.x, i = 0, coff = col0;
  while (i < e.n_cat && rowid >= e.vocab[i]) {
    rowid -= e.vocab[i];
    coff += e.edim[i];
    ++i;
  }
  if (i >= e.n_cat) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int d = wave; d < e.edim[i]; d += 4) {
    float s = 0.f;
    for (int b = lane; b < B; b += 64) {
      const int sb = row_mod > 0 ? b % row_mod : b;
      if (cat[(size_t)sb * ld_cat + i] == rowid) s += dout[(size_t)b * ldd + coff + d];
    }
    s = wave_sum(s);
    if (lane == 0) {
      float *g = grad + e.table_off[i] + (size_t)rowid * e.edim[i] + d;
      *g = accumulate ? *g + s : s;
    }
  }
}

// ---------------------------------------------------------------------------
// WGAN-GP elementwise pieces
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) gp_interp_kernel(const float *__restrict__ real, int ldr,
                                                        const float *__restrict__ fake, int ldf,
                                                        const float *__restrict__ alpha,
                                                        float *__restrict__ inter, int ldi, int B,
                                                        int G) {
  const int b = blockIdx.y;
  const float a = alpha[b];
  const float *r = real + (size_t)b * ldr;
  const float *f = fake + (size_t)b * ldf;
  float *o = inter + (size_t)b * ldi;
  for (int g = blockIdx.x * blockDim.x + threadIdx.x; g < G; g += gridDim.x * blockDim.x) {
    const float rv = r[g];
    const float diff = f[g] - rv;  // tf_utils.py:163
    o[g] = rv + a * diff;          // tf_utils.py:164 (mul then add, no fma contraction)
  }
}

__global__ void __launch_bounds__(256) aug_add_kernel(float *__restrict__ x, int ldx,
                                                      const float *__restrict__ mask,
                                                      const float *__restrict__ noise, int ldn,
                                                      int row_bcast, int B, int G) {
  const int b = blockIdx.y;
  const float m = mask[b];
  const float *n = noise + (row_bcast ? 0 : (size_t)b * ldn);
  float *o = x + (size_t)b * ldx;
  for (int g = blockIdx.x * blockDim.x + threadIdx.x; g < G; g += gridDim.x * blockDim.x)
    o[g] = o[g] + m * n[g];
}

// one wave per row
__global__ void __launch_bounds__(256) rowdot_kernel(const float *__restrict__ A, int lda,
                                                     const float *__restrict__ w,
                                                     const float *__restrict__ b,
                                                     float *__restrict__ out, int R, int H) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= R) return;
  const float *a = A + (size_t)row * lda;
  float s = 0.f;
  for (int j = lane; j < H; j += 64) s += a[j] * w[j];
  s = wave_sum(s);
  if (lane == 0) out[row] = s + (b ? b[0] : 0.f);
}

__global__ void __launch_bounds__(256) delta_init_kernel(const float *__restrict__ A, int lda,
                                                         const float *__restrict__ w,
                                                         float *__restrict__ delta, int ldd, int R,
                                                         int H, int n0, int n1, float c0, float c1,
                                                         float c2, float slope) {
  const long long total = (long long)R * H;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int r = (int)(idx / H), j = (int)(idx % H);
    const float c = r < n0 ? c0 : (r < n0 + n1 ? c1 : c2);
    const float a = A[(size_t)r * lda + j];
    const float v = c * w[j];
    delta[(size_t)r * ldd + j] = a > 0.f ? v : v * slope;
  }
}

// fused Dense(1) forward + backward seed: one wave per row
__global__ void __launch_bounds__(256) critic_head_kernel(const float *__restrict__ A, int lda,
                                                          const float *__restrict__ w,
                                                          const float *__restrict__ b,
                                                          float *__restrict__ out,
                                                          float *__restrict__ delta, int ldd, int R,
                                                          int H, int n0, int n1, float c0, float c1,
                                                          float c2, float slope) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= R) return;
  const float *a = A + (size_t)row * lda;
  float *d = delta + (size_t)row * ldd;
  const float c = row < n0 ? c0 : (row < n0 + n1 ? c1 : c2);
  float s = 0.f;
  for (int j = lane; j < H; j += 64) {
    const float av = a[j], wv = w[j];
    s += av * wv;
    const float v = c * wv;
    d[j] = av > 0.f ? v : v * slope;
  }
  s = wave_sum(s);
  if (lane == 0) out[row] = s + b[0];
}

// gradient-penalty finalisation, one block per interpolated row b:
//   s = sqrt(sum of the per-slab partials) (no epsilon, tf_utils.py:169); f = coef*(s-1)/s;
//   slopes[b] = s; fscale[b] = f; d1[b,:] *= f; xcov[b, 0:ncov] = 0
__global__ void __launch_bounds__(256) gp_finalize_kernel(const float *__restrict__ rowsq,
                                                          int rowsq_ld, int nslab,
                                                          float *__restrict__ slopes,
                                                          float *__restrict__ fscale,
                                                          float *__restrict__ d1, int ldd, int H1,
                                                          float *__restrict__ xcov, int ldx, int ncov,
                                                          float coef) {
  __shared__ float red[4];
  const int b = blockIdx.x;
  const float *q = rowsq + (size_t)b * rowsq_ld;
  float s = 0.f;
  for (int j = threadIdx.x; j < nslab; j += blockDim.x) s += q[j];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  const float sl = sqrtf(red[0] + red[1] + red[2] + red[3]);
  const float f = coef * (sl - 1.f) / sl;  // sl == 0 -> -inf, propagates NaN like the reference
  if (threadIdx.x == 0) {
    slopes[b] = sl;
    fscale[b] = f;
  }
  float *d = d1 + (size_t)b * ldd;
  for (int j = threadIdx.x; j < H1; j += blockDim.x) d[j] = d[j] * f;
  float *xc = xcov + (size_t)b * ldx;
  for (int j = threadIdx.x; j < ncov; j += blockDim.x) xc[j] = 0.f;
}

// Gram form of the gradient penalty (cgan.hip critic_core): rows [0, H) of Xt <- the transposed gene rows of the
// first kernel, Xt[h, g] = W[g, h]; covariate columns [G, G + ncov) of those rows <- 0 (the penalty has no gradient
// w.r.t. the covariate rows of W_1).  64 x 64 tiles through LDS: both sides move whole 256-B row segments.
// bvec (optional, [G]) / bpart [gridDim.x, H]: the tile's share of bvec . W[0:G, :] as a by-product of the pass
// (row sums over the tile's 64 rows in a fixed order; ff_hx_kernel adds the tiles up in order: deterministic)
__global__ void __launch_bounds__(256) wt_rows_kernel(const float *__restrict__ W, int ldw, int G, int H,
                                                      float *__restrict__ Xt, int ldx, int ncov,
                                                      const float *__restrict__ bvec, float *__restrict__ bpart) {
  __shared__ float t[64][65];
  __shared__ float bp[4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int g0 = blockIdx.x * 64, h0 = blockIdx.y * 64;
  float bs = 0.f;
#pragma unroll 4
  for (int r = ty; r < 64; r += 4) {
    const int g = g0 + r, hh = h0 + tx;
    const float w = (g < G && hh < H) ? W[(size_t)g * ldw + hh] : 0.f;
    t[r][tx] = w;
    if (bvec) bs += (g < G ? bvec[g] : 0.f) * w;
  }
  if (bvec) bp[ty][tx] = bs;
  __syncthreads();
  if (bvec && ty == 0 && h0 + tx < H) bpart[(size_t)blockIdx.x * H + h0 + tx] = (bp[0][tx] + bp[1][tx]) + (bp[2][tx] + bp[3][tx]);
#pragma unroll 4
  for (int r = ty; r < 64; r += 4) {
    const int hh = h0 + r, g = g0 + tx;
    if (hh < H && g < G) Xt[(size_t)hh * ldx + g] = t[tx][r];
  }
  if (blockIdx.x == 0)
    for (int q = threadIdx.x; q < 64 * ncov; q += 256) {
      const int hh = h0 + q / ncov;
      if (hh < H) Xt[(size_t)hh * ldx + G + q % ncov] = 0.f;
    }
}

// covariate column c of source row b: [num | emb_0 | emb_1 ...] (tf_utils.py:120-131); an index outside its table gives 0
__device__ __forceinline__ float cov_value(const agx_embed_t &e, const float *__restrict__ tables, const float *__restrict__ num,
                                           int n_num, int ld_num, const int *__restrict__ cat, int ld_cat, int b, int c) {
  if (c < n_num) return num[(size_t)b * ld_num + c];
  c -= n_num;
  for (int i = 0; i < e.n_cat; ++i) {
    if (c < e.edim[i]) {
      const int row = cat[(size_t)b * ld_cat + i];
      return (row >= 0 && row < e.vocab[i]) ? tables[e.table_off[i] + (size_t)row * e.edim[i] + c] : 0.f;
    }
    c -= e.edim[i];
  }
  return 0.f;
}

// Factored fake rows (cgan.hip record_iteration `ff`): the rows in front of the stacked critic input, one block per row.
//   r == 0: [b_out | 0]   1 <= r <= ncov: [0 | unit vector r - 1]   ncov < r < he: 0   he <= r < he + HL: [W_out[r - he] | 0]
//   above: 0   (he = 1 + ncov rounded up to a multiple of 4: the W_out rows start a 16-byte-aligned block of rows)
__global__ void __launch_bounds__(256) ff_stage_kernel(const float *__restrict__ Wout, int ldw, const float *__restrict__ bout,
                                                       int HL, int G, int ncov, int he, float *__restrict__ Xhd, int ldx) {
  const int r = blockIdx.y;
  float *row = Xhd + (size_t)r * ldx;
  const float *src = r == 0 ? bout : ((r >= he && r < he + HL) ? Wout + (size_t)(r - he) * ldw : nullptr);
  for (int c = blockIdx.x * 256 + threadIdx.x; c < G + ncov; c += gridDim.x * 256) {
    float v = 0.f;
    if (c < G) v = src ? src[c] : 0.f;
    else v = (r >= 1 && r <= ncov && c - G == r - 1) ? 1.f : 0.f;
    row[c] = v;
  }
}

// blocks [0, B): Hx[i] = [1 | covariates of row i | 0 .. he | h[i, 0:HL] | 0 ...] (width ldhx) - the constant of the
// output bias, the row's covariate columns FROM THE CURRENT TABLES (also written into the real row's covariate columns of
// X: the step's refresh, tf_utils.py:213-214 with the just-updated embeddings), the generator's last hidden activations of
// the step's z batch: the column order is the row order of ff_stage_kernel.  Blocks [B, B + he): the first he rows of the headroom's first-layer image P
// [.., ldp], which need no product: row 0 = b_out . W_g = the ntile partial rows of wt_rows_kernel added up in order,
// rows 1..ncov = the covariate rows of W_1 themselves, the rest 0.
__global__ void __launch_bounds__(256) ff_hx_kernel(const float *__restrict__ hrows, int ldh, int HL, float *__restrict__ X,
                                                    int ldx, int G, int ncov, int he, float *__restrict__ Hx, int ldhx, int B,
                                                    const float *__restrict__ bpart, int ntile, const float *__restrict__ W1,
                                                    int ldw, int H1, float *__restrict__ P, int ldp, const agx_embed_t e,
                                                    const float *__restrict__ tables, const float *__restrict__ num, int n_num,
                                                    const int *__restrict__ cat) {
  const int i = blockIdx.x;
  if (i >= B + he) {
    // row 0, 16 columns per block: 16 phases over the tiles (every load of a thread in flight together - one thread per
    // column walked the 284 tiles of the GTEx shape in 20 us), the phases added in order
    __shared__ float ph[16][16];
    const int cl = threadIdx.x & 15, p16 = threadIdx.x >> 4, c = (i - B - he) * 16 + cl;
    float q = 0.f;
    if (c < H1)
      for (int t = p16; t < ntile; t += 16) q += bpart[(size_t)t * H1 + c];
    ph[p16][cl] = q;
    __syncthreads();
    if (p16 == 0 && c < H1) {
      float v = 0.f;
#pragma unroll
      for (int k = 0; k < 16; ++k) v += ph[k][cl];
      P[c] = v;
    }
    return;
  }
  if (i >= B) {
    const int r = i - B;
    if (r == 0) return;   // (the blocks above)
    for (int c = threadIdx.x; c < H1; c += 256)
      P[(size_t)r * ldp + c] = r <= ncov ? W1[(size_t)(G + r - 1) * ldw + c] : 0.f;
    return;
  }
  for (int c = threadIdx.x; c < ldhx; c += 256) {
    float v = 0.f;
    if (c == 0) v = 1.f;
    else if (c <= ncov) {
      v = cov_value(e, tables, num, n_num, n_num, cat, e.n_cat, i, c - 1);
      X[(size_t)i * ldx + G + (c - 1)] = v;
    }
    else if (c >= he && c < he + HL) v = hrows[(size_t)i * ldh + (c - he)];
    Hx[(size_t)i * ldhx + c] = v;
  }
}

// one block per row: s = ||g||_2 ; g *= coef*(s-1)/s
__global__ void __launch_bounds__(256) gp_rownorm_scale_kernel(float *__restrict__ g, int ldg,
                                                               float *__restrict__ slopes, int G,
                                                               float coef) {
  __shared__ float red[4];
  float *row = g + (size_t)blockIdx.x * ldg;
  float s = 0.f;
  for (int j = threadIdx.x; j < G; j += blockDim.x) {
    const float v = row[j];
    s += v * v;
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  const float tot = red[0] + red[1] + red[2] + red[3];
  const float sl = sqrtf(tot);  // no epsilon (tf_utils.py:169)
  if (threadIdx.x == 0) slopes[blockIdx.x] = sl;
  const float f = coef * (sl - 1.f) / sl;  // sl == 0 -> -inf ; -inf * 0 = NaN as in the reference
  for (int j = threadIdx.x; j < G; j += blockDim.x) row[j] = row[j] * f;
}

__global__ void __launch_bounds__(256) wgan_loss_kernel(const float *__restrict__ out, int n0,
                                                        int n1, const float *__restrict__ slopes,
                                                        int ns, float inv_b, float gp_weight,
                                                        float fake_sign, float *__restrict__ losses) {
  __shared__ float red[3][4];
  float sr = 0.f, sf = 0.f, sg = 0.f;
  for (int i = threadIdx.x; i < n0; i += blockDim.x) sr += out[i];
  for (int i = threadIdx.x; i < n1; i += blockDim.x) sf += out[n0 + i];
  for (int i = threadIdx.x; i < ns; i += blockDim.x) {
    const float d = slopes[i] - 1.f;
    sg += d * d;
  }
  sr = wave_sum(sr);
  sf = wave_sum(sf);
  sg = wave_sum(sg);
  if ((threadIdx.x & 63) == 0) {
    red[0][threadIdx.x >> 6] = sr;
    red[1][threadIdx.x >> 6] = sf;
    red[2][threadIdx.x >> 6] = sg;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const float r = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    const float f = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    const float g = red[2][0] + red[2][1] + red[2][2] + red[2][3];
    const float w = -r * inv_b + fake_sign * f * inv_b;
    const float gp = g * inv_b;
    losses[0] = w;
    losses[1] = gp;
    losses[2] = w + gp_weight * gp;
    losses[3] = 0.f;
  }
}

// column sums over a row range: blockDim = (64 columns, 16 row phases); fixed order => deterministic
__global__ void __launch_bounds__(1024) colsum_kernel(const float *__restrict__ A, int lda, int r0,
                                                      int r1, int H, float *__restrict__ out,
                                                      int accumulate, int n0, int n1, float c0,
                                                      float c1, float c2, int weighted,
                                                      float *__restrict__ bias_out) {
  __shared__ float red[16][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int j = blockIdx.x * 64 + tx;
  float s = 0.f;
  if (j < H) {
    for (int r = r0 + ty; r < r1; r += 16) {
      float v = A[(size_t)r * lda + j];
      if (weighted) v *= (r < n0 ? c0 : (r < n0 + n1 ? c1 : c2));
      s += v;
    }
  }
  red[ty][tx] = s;
  __syncthreads();
  if (ty == 0 && j < H) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += red[k][tx];
    out[j] = accumulate ? out[j] + t : t;
  }
  if (weighted && bias_out && blockIdx.x == 0 && threadIdx.x == 0) {
    // BiasAddGrad of the Dense(1): the two means contribute n0*c0 and n1*c1 (sum of equal terms)
    bias_out[0] = (float)n0 * c0 + (float)n1 * c1;
  }
}

// ---------------------------------------------------------------------------
// optimizers (Keras-v2 formulas; dense apply == sparse apply on the scatter-added grad)
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) rmsprop_kernel(float *__restrict__ w,
                                                      const float *__restrict__ g,
                                                      float *__restrict__ rms, long long n,
                                                      float lr, float rho, float eps, float gs,
                                                      long long hole_lo4, long long hole_len4) {
  // float4 indices [hole_lo4, hole_lo4 + hole_len4) are skipped (updated by a fused GEMM epilogue)
  const long long n4 = n >> 2;
  const float omr = 1.f - rho;
  float4 *w4 = reinterpret_cast<float4 *>(w);
  const float4 *g4 = reinterpret_cast<const float4 *>(g);
  float4 *r4 = reinterpret_cast<float4 *>(rms);
  for (long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x; j < n4 - hole_len4;
       j += (long long)gridDim.x * blockDim.x) {
    const long long i = j < hole_lo4 ? j : j + hole_len4;
    float4 wv = w4[i], gv = g4[i], rv = r4[i];
#define AGX_RMS1(c)                                  \
  {                                                  \
    const float gg = gv.c * gs;                      \
    rv.c = rho * rv.c + omr * (gg * gg);             \
    wv.c = wv.c - lr * gg / (sqrtf(rv.c) + eps);     \
  }
    AGX_RMS1(x) AGX_RMS1(y) AGX_RMS1(z) AGX_RMS1(w)
#undef AGX_RMS1
    w4[i] = wv;
    r4[i] = rv;
  }
  const long long i = (n4 << 2) + (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    const float gg = g[i] * gs;
    const float r = rho * rms[i] + omr * (gg * gg);
    rms[i] = r;
    w[i] = w[i] - lr * gg / (sqrtf(r) + eps);
  }
}

__global__ void __launch_bounds__(256) adam_kernel(float *__restrict__ w,
                                                   const float *__restrict__ g,
                                                   float *__restrict__ m, float *__restrict__ v,
                                                   long long n, float lr_t, float b1, float b2,
                                                   float eps, float gs, const float *__restrict__ lr_dev) {
  // lr_dev: bias-corrected step size staged in device memory (a replayed graph cannot carry a
  // per-step scalar argument); the host computes it exactly as for the eager call
  if (lr_dev) lr_t = *lr_dev;
  const float omb1 = 1.f - b1, omb2 = 1.f - b2;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x) {
    const float gg = g[i] * gs;
    const float mi = b1 * m[i] + omb1 * gg;
    const float vi = b2 * v[i] + omb2 * (gg * gg);
    m[i] = mi;
    v[i] = vi;
    w[i] = w[i] - lr_t * mi / (sqrtf(vi) + eps);
  }
}

// ---------------------------------------------------------------------------
// Philox4x32-10
// ---------------------------------------------------------------------------
__device__ __forceinline__ void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3,
                                              unsigned k0, unsigned k1, unsigned (&out)[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0;
    const unsigned long long p1 = (unsigned long long)0xCD9E8D57u * c2;
    const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0;
    const unsigned n1 = (unsigned)p1;
    const unsigned n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1;
    const unsigned n3 = (unsigned)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__device__ __forceinline__ float u01(unsigned x) {  // [0,1) with 24 random bits
  return (float)(x >> 8) * (1.0f / 16777216.0f);
}

__global__ void set_u64_kernel(unsigned long long *dst, unsigned long long v) { *dst = v; }

// elements 4q..4q+3 of the (seed, offset)-keyed stream; NORMAL: Box-Muller, else U[0,1)
template <bool NORMAL>
__device__ __forceinline__ void rng_fill(float *__restrict__ out, long long n, unsigned long long seed,
                                         unsigned long long offset, long long q0, long long qstride,
                                         long long skip = 0, int cols = 0, int ld = 0) {
  // counter block q produces elements 4q .. 4q+3 of the (seed, offset)-keyed stream; out receives
  // elements [skip, skip + n)
  const long long qlo = skip >> 2, nq = (skip + n + 3) >> 2;
  for (long long q = qlo + q0; q < nq; q += qstride) {
    const unsigned long long ctr = (offset >> 2) + (unsigned long long)q;
    unsigned r[4];
    philox4x32_10((unsigned)ctr, (unsigned)(ctr >> 32), (unsigned)(offset & 3u), NORMAL ? 1u : 0u,
                  (unsigned)seed, (unsigned)(seed >> 32), r);
    float v[4];
    if (NORMAL) {
      // Box-Muller on two pairs; 1-u keeps the log argument in (0,1]
      const float u0 = 1.0f - u01(r[0]), u1 = u01(r[1]);
      const float u2 = 1.0f - u01(r[2]), u3 = u01(r[3]);
      const float ra = sqrtf(-2.0f * __logf(u0)), rb = sqrtf(-2.0f * __logf(u2));
      float s0, c0, s1, c1;
      __sincosf(6.28318530717958647692f * u1, &s0, &c0);
      __sincosf(6.28318530717958647692f * u3, &s1, &c1);
      v[0] = ra * c0; v[1] = ra * s0; v[2] = rb * c1; v[3] = rb * s1;
    } else {
      v[0] = u01(r[0]); v[1] = u01(r[1]); v[2] = u01(r[2]); v[3] = u01(r[3]);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const long long j = 4 * q + k - skip;
      if (j >= 0 && j < n) out[cols > 0 ? (j / cols) * ld + j % cols : j] = v[k];
    }
  }
}

template <bool NORMAL>
__global__ void __launch_bounds__(256) rng_kernel(float *__restrict__ out, long long n,
                                                  unsigned long long seed,
                                                  unsigned long long offset,
                                                  const unsigned long long *__restrict__ base_dev) {
  if (base_dev) offset += *base_dev;  // device-resident stream position (graph replay)
  rng_fill<NORMAL>(out, n, seed, offset, (long long)blockIdx.x * blockDim.x + threadIdx.x,
                   (long long)gridDim.x * blockDim.x);
}

// several stream segments in one launch: blockIdx.y = segment (out may be strided: rows of
// `row_len` values at leading dimension `ld`, e.g. z written straight into the layer-1 input)
struct RngSegs {
  int n;
  agx_rng_seg_t seg[AGX_MAX_RNG_SEGS];
};
__global__ void __launch_bounds__(256) rng_multi_kernel(const RngSegs segs, unsigned long long seed,
                                                        const unsigned long long *__restrict__ base_dev) {
  const agx_rng_seg_t sg = segs.seg[blockIdx.y];
  unsigned long long offset = sg.sub_offset + (base_dev ? *base_dev : 0ull);
  const long long q0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long qs = (long long)gridDim.x * blockDim.x;
  if (sg.normal) rng_fill<true>(sg.out, sg.n, seed, offset, q0, qs, sg.skip);
  else rng_fill<false>(sg.out, sg.n, seed, offset, q0, qs, sg.skip);
}

// draws of a whole iteration + covariate columns of the generator input (agx::iter_draws)
// (bx, by, gx: block coordinates and x extent of the draws grid - its own launch, or block rows behind the staging rows)
__device__ __forceinline__ void iter_draws_body(const agx::IterDraws &d, unsigned long long seed, unsigned long long base,
                                                const agx_embed_t &e, const float *__restrict__ tables,
                                                const float *__restrict__ num, int n_num,
                                                const int *__restrict__ cat, int ld_cat, int B, int rows_cov,
                                                int zcols, float *__restrict__ gin0, int ldg0, int ncov,
                                                int gen_cov_row, const agx_embed_t &e2, const float *__restrict__ tables2,
                                                float *__restrict__ X, int ldx, int xcol0, int xrows, int ncov2,
                                                int bx, int by, int gx) {
  if (by >= d.n + gen_cov_row) {
    // covariate columns of the stacked critic input, from the critic's tables (as stack_rows_kernel writes them)
    const long long total = (long long)xrows * ncov2;
    for (long long idx = (long long)bx * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gx * blockDim.x) {
      const int r = (int)(idx / ncov2), b = r % B;
      int c = (int)(idx % ncov2);
      float v = 0.f;
      if (c < n_num) {
        v = num[(size_t)b * n_num + c];
      } else {
        c -= n_num;
        for (int i = 0; i < e2.n_cat; ++i) {
          if (c < e2.edim[i]) {
            const int row = cat[(size_t)b * ld_cat + i];
            if (row >= 0 && row < e2.vocab[i]) v = tables2[e2.table_off[i] + (size_t)row * e2.edim[i] + c];
            break;
          }
          c -= e2.edim[i];
        }
      }
      X[(size_t)r * ldx + xcol0 + (int)(idx % ncov2)] = v;
    }
    return;
  }
  if (by < d.n) {
    const agx::DrawSeg sg = d.seg[by];
    const unsigned long long offset = sg.sub_offset + base;
    const long long q0 = (long long)bx * blockDim.x + threadIdx.x;
    const long long qs = (long long)gx * blockDim.x;
    if (sg.normal) rng_fill<true>(sg.out, sg.n, seed, offset, q0, qs, sg.skip, sg.cols, sg.ld);
    else rng_fill<false>(sg.out, sg.n, seed, offset, q0, qs, sg.skip, sg.cols, sg.ld);
    return;
  }
  const long long total = (long long)rows_cov * ncov;
  for (long long idx = (long long)bx * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gx * blockDim.x) {
    const int r = (int)(idx / ncov), b = r % B;
    int c = (int)(idx % ncov);
    float v = 0.f;
    if (c < n_num) {
      v = num[(size_t)b * n_num + c];
    } else {
      c -= n_num;
      for (int i = 0; i < e.n_cat; ++i) {
        if (c < e.edim[i]) {
          const int row = cat[(size_t)b * ld_cat + i];
          if (row >= 0 && row < e.vocab[i]) v = tables[e.table_off[i] + (size_t)row * e.edim[i] + c];
          break;
        }
        c -= e.edim[i];
      }
    }
    gin0[(size_t)r * ldg0 + zcols + (int)(idx % ncov)] = v;
  }
}

__global__ void __launch_bounds__(256) iter_draws_kernel(const agx::IterDraws d, unsigned long long seed,
                                                         const unsigned long long *__restrict__ base_dev,
                                                         agx_embed_t e, const float *__restrict__ tables,
                                                         const float *__restrict__ num, int n_num,
                                                         const int *__restrict__ cat, int ld_cat, int B, int rows_cov,
                                                         int zcols, float *__restrict__ gin0, int ldg0, int ncov,
                                                         int gen_cov_row, agx_embed_t e2, const float *__restrict__ tables2,
                                                         float *__restrict__ X, int ldx, int xcol0, int xrows, int ncov2) {
  iter_draws_body(d, seed, base_dev ? *base_dev : 0ull, e, tables, num, n_num, cat, ld_cat, B, rows_cov, zcols, gin0, ldg0,
                  ncov, gen_cov_row, e2, tables2, X, ldx, xcol0, xrows, ncov2, (int)blockIdx.x, (int)blockIdx.y, (int)gridDim.x);
}

// ---------------------------------------------------------------------------
// stacked critic input: gene columns of one row block + covariate columns of the leading rows
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) stack_rows_kernel(
    agx_embed_t e, const float *__restrict__ tables, const float *__restrict__ src, int lds,
    int dst_row0, int src_rows, const float *__restrict__ num, int n_num, int ld_num,
    const int *__restrict__ cat, int ld_cat, int B, int rows_cov, int G, float *__restrict__ X, int ldx,
    const float *__restrict__ aug_mask, const float *__restrict__ aug_noise, int aug_ldn, int ncov,
    int gene_blocks) {
  if ((int)blockIdx.x < gene_blocks) {
    if (!src) return;
    const int b = blockIdx.y;
    if (b >= src_rows) return;
    const float m = aug_mask ? aug_mask[b] : 0.f;
    const float *s = src + (size_t)b * lds;
    const float *nz = aug_mask ? aug_noise + (size_t)b * aug_ldn : nullptr;
    float *d = X + (size_t)(dst_row0 + b) * ldx;
    for (int g = blockIdx.x * blockDim.x + threadIdx.x; g < G; g += gene_blocks * blockDim.x) {
      float v = s[g];
      if (aug_mask) v = v + m * nz[g];  // tf_utils.py:209-210, :247
      d[g] = v;
    }
    return;
  }
  // covariate columns [G, G+ncov) of rows [0, rows_cov): [num | emb_0 | emb_1 ...] of source row r % B
  const int nb = gridDim.x - gene_blocks;
  const long long total = (long long)rows_cov * ncov;
  for (long long idx = ((long long)(blockIdx.y * nb + (blockIdx.x - gene_blocks))) * blockDim.x + threadIdx.x;
       idx < total; idx += (long long)nb * gridDim.y * blockDim.x) {
    const int r = (int)(idx / ncov);
    const int c = (int)(idx % ncov);
    X[(size_t)r * ldx + G + c] = cov_value(e, tables, num, n_num, ld_num, cat, ld_cat, r % B, c);
  }
}

// ---------------------------------------------------------------------------
// every small gradient piece of a step in ONE launch (1024-thread blocks):
//   blocks [0, cs_end)        column sums (BiasAddGrad) of up to AGX_MAX_LAYERS deltas
//   blocks [cs_end, w_end)    weighted column sums = Dense(1) kernel/bias gradient
//   blocks [w_end, emb_end)   embedding-table gradients straight from the first-layer deltas:
//                             dE_i[v,d] = (sum_{rows r: cat[r % B, i] == v} D[r,:]) . W[col0+off_i+d, :]
//   block  emb_end            the loss scalars (tf_utils.py:140-181)
// Fixed summation orders: deterministic.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void colsum_block(const float *__restrict__ A, int lda, int r0, int r1, int H,
                                             float *__restrict__ out, int blk, int weighted, int n0,
                                             int n1, float c0, float c1, float c2,
                                             float *__restrict__ bias_out, float (*red)[64], float *svec) {
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int j = blk * 64 + tx;
  if (r1 - r0 >= 128 && (lda & 3) == 0 && (reinterpret_cast<uintptr_t>(A) & 15) == 0) {
    // many rows (large batches): 16 float4 columns x 64 row phases, four 16-B loads in flight per
    // thread - 1024 rows are 16 row steps instead of 64 (the padding columns up to lda are readable)
    const int c4 = threadIdx.x & 15, ph = threadIdx.x >> 4;
    const int jc = blk * 64 + 4 * c4;
    float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;
    if (jc < H) {
      auto wt = [&](int r) { return weighted ? (r < n0 ? c0 : (r < n0 + n1 ? c1 : c2)) : 1.f; };
      int r = r0 + ph;
      for (; r + 192 < r1; r += 256) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4 *>(A + (size_t)(r + 64 * u) * lda + jc);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const float w = wt(r + 64 * u);
          float4 &a = (u & 1) ? a1 : a0;
          if (weighted) { a.x += v[u].x * w; a.y += v[u].y * w; a.z += v[u].z * w; a.w += v[u].w * w; }
          else { a.x += v[u].x; a.y += v[u].y; a.z += v[u].z; a.w += v[u].w; }
        }
      }
      for (; r < r1; r += 64) {
        const float4 v = *reinterpret_cast<const float4 *>(A + (size_t)r * lda + jc);
        const float w = wt(r);
        if (weighted) { a0.x += v.x * w; a0.y += v.y * w; a0.z += v.z * w; a0.w += v.w * w; }
        else { a0.x += v.x; a0.y += v.y; a0.z += v.z; a0.w += v.w; }
      }
    }
    float4 *s4 = reinterpret_cast<float4 *>(svec);   // [64 phases][16 float4 columns]
    s4[ph * 16 + c4] = make_float4(a0.x + a1.x, a0.y + a1.y, a0.z + a1.z, a0.w + a1.w);
    __syncthreads();
    if (threadIdx.x < 64 && j < H) {
      float t = 0.f;
      for (int k = 0; k < 64; ++k) t += svec[k * 64 + tx];
      out[j] = t;
    }
    if (weighted && bias_out && blk == 0 && threadIdx.x == 0) bias_out[0] = (float)n0 * c0 + (float)n1 * c1;
    return;
  }
  float s = 0.f;
  if (j < H) {
    // four independent partial sums, four loads in flight (a single dependent chain of loads made
    // this the longest kernel of a 512-row step: 126 us beside a 152 us product)
    float q[4] = {0.f, 0.f, 0.f, 0.f};
    int r = r0 + ty;
    for (; r + 48 < r1; r += 64) {
      float v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = A[(size_t)(r + 16 * u) * lda + j];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int ru = r + 16 * u;
        q[u] += weighted ? v[u] * (ru < n0 ? c0 : (ru < n0 + n1 ? c1 : c2)) : v[u];
      }
    }
    for (; r < r1; r += 16) {
      float v = A[(size_t)r * lda + j];
      if (weighted) v *= (r < n0 ? c0 : (r < n0 + n1 ? c1 : c2));
      q[0] += v;
    }
    s = (q[0] + q[1]) + (q[2] + q[3]);
  }
  red[ty][tx] = s;
  __syncthreads();
  if (ty == 0 && j < H) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += red[k][tx];
    out[j] = t;
  }
  if (weighted && bias_out && blk == 0 && threadIdx.x == 0) bias_out[0] = (float)n0 * c0 + (float)n1 * c1;
}

// One wave = one 32x32 tile of C = A^T . B over the whole K (v_mfma_f32_32x32x2_f32, k ascending in
// four accumulators that are added at the end: a fixed summation order).  Lane (i = lane & 31, h = lane >> 5) feeds k = 8g + 4h + e of
// row / column i, the same k for both operands; loads are clamped into the operands and masked after.
typedef float tn_f32x16 __attribute__((ext_vector_type(16)));
// (kb, ke: the k range of this wave - the whole K, or a quarter of it where a block's four waves share one tile of a
//  long-K job, tn_split_k; store: write C, else leave the sums in acc)
__device__ __forceinline__ bool tn_split_k(int K) { return K > 256; }
__device__ __forceinline__ void tn_wave_tile(const agx::TnJobs::Job &t, int tile, int kb, int ke, bool store, float *acc_out) {
  const int tiles_n = (t.N + 31) / 32;
  if (tile >= ((t.M + 31) / 32) * tiles_n) return;
  const int lane = threadIdx.x & 63, i = lane & 31, h = lane >> 5;
  const int m0 = (tile / tiles_n) * 32, n0 = (tile % tiles_n) * 32;
  const bool mv = m0 + i < t.M, nv = n0 + i < t.N;
  const float *__restrict__ a = t.A + min(m0 + i, t.M - 1);
  const float *__restrict__ b = t.B + min(n0 + i, t.N - 1);
  // four accumulators (one per e): four independent MFMA chains, each a quarter as long - a single fp32 chain over
  // K = 512 same-signed terms (S = d_1^T diag(f) d_1 at 512 rows) drifts by ~1e-5 relative
  tn_f32x16 acc4[4];
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc4[q][r] = 0.f;
  constexpr int U = 6;
  const int ngroups = (max(ke - kb, 0) + 7) / 8, kl = ke - 1;
  for (int g0 = 0; g0 < ngroups; g0 += U) {
    float av[U][4], bv[U][4];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int kc = max(min(kb + 8 * (g0 + u) + 4 * h + e, kl), 0);
        av[u][e] = a[(size_t)kc * t.lda];
        bv[u][e] = b[(size_t)kc * t.ldb];
      }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const bool kv = kb + 8 * (g0 + u) + 4 * h + e <= kl;
        acc4[e] = __builtin_amdgcn_mfma_f32_32x32x2f32((kv && mv) ? av[u][e] : 0.f, (kv && nv) ? bv[u][e] : 0.f, acc4[e], 0, 0, 0);
      }
  }
  tn_f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = (acc4[0][r] + acc4[1][r]) + (acc4[2][r] + acc4[3][r]);
  if (!store) {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc_out[r] = acc[r];
    return;
  }
  if (nv) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = m0 + (r & 3) + 8 * (r >> 2) + 4 * h;
      if (row < t.M) t.C[(size_t)row * t.ldc + n0 + i] = acc[r];
    }
  }
}

// the 32x32 tile `tile` of C from a wave's accumulator image (C/D map of the 32x32 MFMA, see tn_wave_tile)
__device__ __forceinline__ void tn_store_tile(const agx::TnJobs::Job &t, int tile, const float *acc) {
  const int tiles_n = (t.N + 31) / 32;
  const int lane = threadIdx.x & 63, i = lane & 31, h = lane >> 5;
  const int m0 = (tile / tiles_n) * 32, n0 = (tile % tiles_n) * 32;
  if (n0 + i < t.N) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = m0 + (r & 3) + 8 * (r >> 2) + 4 * h;
      if (row < t.M) t.C[(size_t)row * t.ldc + n0 + i] = acc[r];
    }
  }
}

// dE_i[rowid, d] = s . W[coff + d, :] for every d at once: one wave per embedding dimension (the d loop
// with a block-wide reduction per d was a chain of edim dependent round trips: the launch's long pole)
__device__ __forceinline__ void emb_dots(const agx_small_grads_t &j, int i, int rowid, int coff,
                                         const float *sfin, int H) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int d = wave; d < j.emb.edim[i]; d += 16) {
    const float *__restrict__ w = j.emb_W + (size_t)(coff + d) * j.emb_ldw;
    float v = 0.f;
    for (int k = lane; k < H; k += 64) v += sfin[k] * w[k];
    v = agx::wave_sum(v);
    if (lane == 0) j.emb_grad[j.emb.table_off[i] + (size_t)rowid * j.emb.edim[i] + d] = v;
  }
}

// Which job a workgroup runs is decided on the HOST: segment s covers blocks [start[s], start[s+1]) and is job
// (kind, idx).  The kernel used to walk the job lists itself - a chain of ~8 dependent scalar loads from the 1 KB
// argument block with a branch after each - and every workgroup spent 2.2 us of a 7.7 us launch there before its
// job began (phase stamps).  Now: the start table in one go, a branch-free count, one more load.
// Segments are ordered longest job first (embedding rows, products, loss, weighted and plain column sums) so that
// a launch of more workgroups than CUs (the generator step: 333) does not start its long pole in the second round.
struct SmallPlan {
  int n_seg;
  int start[37];   // start[n_seg] = number of blocks
  signed char kind[36], idx[36];   // kind: 0 product, 1 column sum, 2 weighted column sum, 3 embedding table, 4 loss
  short coff[36];  // embedding table idx: its first kernel row
};

__global__ void __launch_bounds__(1024) small_grads_kernel(const SmallPlan plan, const agx_small_grads_t j,
                                                           const agx::LossCopy lc, const agx::TnJobs tn) {
  __shared__ float red[16][64];
  __shared__ __attribute__((aligned(16))) float svec[8192];   // [0,6144): partial / final column sums, [6144,8192): row match flags
  __shared__ int elist[2048];   // matching rows of an embedding-table row (many-row path)
  __shared__ int wcnt[16];
  int seg = 0;
#pragma unroll
  for (int s = 1; s < 36; ++s) seg += (s < plan.n_seg && (int)blockIdx.x >= plan.start[s]) ? 1 : 0;
  const int kind = plan.kind[seg], jidx = plan.idx[seg];
  int blk = (int)blockIdx.x - plan.start[seg];
  if (kind == 0) {
    // the tiny weight-gradient products, 4 tiles per block - one wave per SIMD (16 tiles
    // on one CU queue 4 waves' MFMAs per SIMD: the launch grew from 8.5 to 13 us); waves 4..15 leave
    const agx::TnJobs::Job &t = tn.job[jidx];
    if (tn_split_k(t.K)) {
      // long K (S and Hx^T d_1 at 512 rows: 22 us as one wave per tile): the block's waves take a share of K each (four or
      // eight), leave their sums in LDS, and wave 0 adds them in wave order and stores
      const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
      const int nw = t.K > 1024 ? 8 : 4;   // (eight shares from 1024 rows on: the hidden-layer gradient over 3B = 1536 rows;
      //  sixteen in two stages measured no different: 16.3 against 16.4 us - the launch's long pole is elsewhere by then)
      const int kq = (((t.K + nw - 1) / nw) + 7) & ~7;
      // (every wave, wave 0 too, multiplies its share BEFORE the barrier: wave 0 used to start its own share behind it,
      //  so a job took two shares' time - 22.9 against 16.4 us per launch at 512 rows)
      float part[16];
      if (w < nw) tn_wave_tile(t, blk, w * kq, min(t.K, (w + 1) * kq), false, part);
      if (w >= 1 && w < nw) {
#pragma unroll
        for (int r = 0; r < 16; ++r) svec[((w - 1) * 16 + r) * 64 + lane] = part[r];
      }
      __syncthreads();
      if (w == 0) {
        for (int o = 0; o < nw - 1; ++o)
#pragma unroll
          for (int r = 0; r < 16; ++r) part[r] += svec[(o * 16 + r) * 64 + lane];
        tn_store_tile(t, blk, part);
      }
      return;
    }
    if (threadIdx.x < 256) tn_wave_tile(t, blk * 4 + (int)(threadIdx.x >> 6), 0, t.K, true, nullptr);
    return;
  }
  if (kind == 1) {
    const int c = jidx;
    colsum_block(j.cs_A[c], j.cs_lda[c], j.cs_r0[c], j.cs_r1[c], j.cs_H[c], j.cs_out[c], blk, 0, 0, 0,
                 0.f, 0.f, 0.f, nullptr, red, svec);
    return;
  }
  if (kind == 2) {
    colsum_block(j.w_A, j.w_lda, 0, j.w_R, j.w_H, j.w_out, blk, 1, j.w_n0, j.w_n1, j.w_c0, j.w_c1,
                 j.w_c2, j.w_bias_out, red, svec);
    return;
  }
  if (kind == 3) {
    {
      const int rowid = blk, i = jidx, coff = plan.coff[seg];
      const int H = j.emb_H;
      const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
      float *part = svec, *sfin = H <= 512 ? svec + 1024 : svec;
      // s[k] = sum over the matching rows of D[r,k]; narrow layers split the rows over 1024/Hp
      // thread groups whose partials meet in LDS in a fixed order
      // the 0/1 match of every row is staged in LDS first, so that the row loop below has no
      // dependent (index -> data) load chain: every D load is unconditional and independent
      float *match = svec + 6144;   // rows are processed in chunks of 2048
      // large batches, 16-B aligned deltas: a thread owns 4 consecutive columns, the rows are dealt
      // over 1024 / (H/4) thread groups (H = 256: 16 groups instead of 4), eight 16-B loads in flight
      const bool vec4 = j.emb_rows >= 128 && (H & 3) == 0 && H <= 1024 && (j.emb_ldd & 3) == 0 &&
                        (reinterpret_cast<uintptr_t>(j.emb_D) & 15) == 0;
      if (vec4) {
        const int cols4 = H >> 2, ng = 1024 / cols4;
        const int kk4 = threadIdx.x % cols4, rg = threadIdx.x / cols4;
        float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;
        for (int c0 = 0; c0 < j.emb_rows; c0 += 2048) {
          const int nrow = min(2048, j.emb_rows - c0);
          __syncthreads();
          for (int r = threadIdx.x; r < nrow; r += 1024)
            match[r] = j.emb_cat[(size_t)((c0 + r) % j.emb_B) * j.emb_ld_cat + i] == rowid ? 1.f : 0.f;
          __syncthreads();
          // compact list of the matching rows (ascending), then ONLY those rows are read: a table row matches
          // about rows / vocab of them, and reading all 2B rows (1 MB at 512 rows per GPU) through one CU per
          // table row was the launch's long pole (the list keeps the row loop free of dependent index loads)
          int cnt = 0;
          for (int base = 0; base < nrow; base += 1024) {
            const int r = base + (int)threadIdx.x;
            const bool m = r < nrow && match[r] != 0.f;
            const unsigned long long bal = __ballot(m);
            if (lane == 0) wcnt[wave] = __popcll(bal);
            __syncthreads();
            int off = cnt, tot = 0;
            for (int w = 0; w < 16; ++w) {
              const int cw = wcnt[w];
              off += w < wave ? cw : 0;
              tot += cw;
            }
            if (m) elist[off + __popcll(bal & ((1ull << lane) - 1ull))] = r;
            cnt += tot;
            __syncthreads();
          }
          if (rg < ng) {
            int q = rg;
            for (; q + 3 * ng < cnt; q += 4 * ng) {
              float4 d[4];
#pragma unroll
              for (int u = 0; u < 4; ++u)
                d[u] = *reinterpret_cast<const float4 *>(j.emb_D + (size_t)(c0 + elist[q + u * ng]) * j.emb_ldd + 4 * kk4);
#pragma unroll
              for (int u = 0; u < 4; ++u) {
                float4 &a = (u & 1) ? a1 : a0;
                a.x += d[u].x; a.y += d[u].y; a.z += d[u].z; a.w += d[u].w;
              }
            }
            for (; q < cnt; q += ng) {
              const float4 d = *reinterpret_cast<const float4 *>(j.emb_D + (size_t)(c0 + elist[q]) * j.emb_ldd + 4 * kk4);
              a0.x += d.x; a0.y += d.y; a0.z += d.z; a0.w += d.w;
            }
          }
        }
        float4 *p4 = reinterpret_cast<float4 *>(svec);           // [ng][cols4] float4 = <= 4096 floats
        if (rg < ng) p4[rg * cols4 + kk4] = make_float4(a0.x + a1.x, a0.y + a1.y, a0.z + a1.z, a0.w + a1.w);
        __syncthreads();
        float *sfin4 = svec + 4096;                              // [H] final sums (H <= 1024)
        if ((int)threadIdx.x < H) {
          float t = 0.f;
          for (int g = 0; g < ng; ++g) t += svec[(size_t)(g * cols4) * 4 + threadIdx.x];
          sfin4[threadIdx.x] = t;
        }
        __syncthreads();
        emb_dots(j, i, rowid, coff, sfin4, H);
        return;
      }
      for (int c0 = 0; c0 < j.emb_rows || c0 == 0; c0 += 2048) {
        const int nrow = min(2048, j.emb_rows - c0);
        __syncthreads();
        for (int r = threadIdx.x; r < nrow; r += 1024)
          match[r] = j.emb_cat[(size_t)((c0 + r) % j.emb_B) * j.emb_ld_cat + i] == rowid ? 1.f : 0.f;
        __syncthreads();
        if (H <= 512) {
          const int Hp = (H + 63) & ~63, ng = 1024 / Hp;
          const int kk = threadIdx.x % Hp, rg = threadIdx.x / Hp;
          float sk = 0.f;
          if (rg < ng && kk < H) {
            float q[4] = {0.f, 0.f, 0.f, 0.f};
            int r = rg;
            for (; r + 7 * ng < nrow; r += 8 * ng) {   // eight independent loads in flight
              float d[8];
#pragma unroll
              for (int u = 0; u < 8; ++u) d[u] = j.emb_D[(size_t)(c0 + r + u * ng) * j.emb_ldd + kk];
#pragma unroll
              for (int u = 0; u < 8; ++u) q[u & 3] += match[r + u * ng] != 0.f ? d[u] : 0.f;
            }
            for (; r < nrow; r += ng) {
              const float d = j.emb_D[(size_t)(c0 + r) * j.emb_ldd + kk];
              q[0] += match[r] != 0.f ? d : 0.f;   // select, not multiply: a NaN in another row must not leak
            }
            sk = (q[0] + q[1]) + (q[2] + q[3]);
          }
          if (rg < ng) part[rg * Hp + kk] = (c0 == 0 ? 0.f : part[rg * Hp + kk]) + sk;
        } else {
          for (int k = threadIdx.x; k < H; k += 1024) {
            float sk = 0.f;
#pragma unroll 4
            for (int r = 0; r < nrow; ++r) {
              const float d = j.emb_D[(size_t)(c0 + r) * j.emb_ldd + k];
              sk += match[r] != 0.f ? d : 0.f;
            }
            part[k] = (c0 == 0 ? 0.f : part[k]) + sk;   // H <= 2048 here (host check)
          }
        }
      }
      __syncthreads();
      if (H <= 512) {
        const int Hp = (H + 63) & ~63, ng = 1024 / Hp;
        if ((int)threadIdx.x < H) {
          float t = 0.f;
          for (int g = 0; g < ng; ++g) t += part[g * Hp + threadIdx.x];
          sfin[threadIdx.x] = t;
        }
      }
      __syncthreads();
      // dot with the edim_i kernel rows that the embedding columns feed
      emb_dots(j, i, rowid, coff, sfin, H);
      return;
    }
  }
  if (kind == 4) {
    float sr = 0.f, sf = 0.f, sg = 0.f;
    // critic output of row i: l_out_parts partial dot products (column-split tail launches), added in part order
    auto outv = [&](int i) {
      float v = j.l_out[i];
      for (int s = 1; s < j.l_out_parts; ++s) v += j.l_out[(size_t)s * j.l_out_ld + i];
      return v;
    };
    for (int i = threadIdx.x; i < j.l_n0; i += blockDim.x) sr += outv(i);
    for (int i = threadIdx.x; i < j.l_n1; i += blockDim.x) sf += outv(j.l_n0 + i);
    for (int i = threadIdx.x; i < j.l_ns; i += blockDim.x) {
      const float d = j.l_slopes[i] - 1.f;
      sg += d * d;
    }
    sr = wave_sum(sr); sf = wave_sum(sf); sg = wave_sum(sg);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { red[0][wave] = sr; red[1][wave] = sf; red[2][wave] = sg; }
    __syncthreads();
    if (threadIdx.x == 0) {
      float r = 0.f, f = 0.f, g = 0.f;
      for (int w = 0; w < 16; ++w) { r += red[0][w]; f += red[1][w]; g += red[2][w]; }
      const float wl = -r * j.l_inv_b + j.l_fake_sign * f * j.l_inv_b;
      const float gp = g * j.l_inv_b;
      j.l_losses[0] = wl;
      j.l_losses[1] = gp;
      j.l_losses[2] = wl + j.l_gp_weight * gp;
      j.l_losses[3] = 0.f;
      // last step of a captured iteration: snapshot the whole loss log (this step's row included)
      // into the buffer whose address the stage kernel left in device memory
      if (lc.dst_slot) {
        float *dst = *lc.dst_slot;
        if (dst)
          for (int i = 0; i < lc.n; ++i) dst[i] = lc.src[i];
      }
    }
  }
}

// One launch in front of a replayed iteration: real rows -> rows [0,B) of the stacked critic input,
// covariates -> staging, RNG stream position and the loss-snapshot address -> device memory.
template <bool AUG>
__device__ __forceinline__ void iter_stage_body(const float *__restrict__ x, int ldxs,
                                                float *__restrict__ X, float *__restrict__ X2, int ldx, int G, int vec,
                                                const int *__restrict__ cat, int *__restrict__ scat, int n_cat,
                                                const float *__restrict__ num, float *__restrict__ snum, int n_num,
                                                unsigned long long *rngpos, unsigned long long rng_offset,
                                                float **loss_slot, float *losses_out,
                                                const agx::LrTable &lrs, float *__restrict__ lr_dev,
                                                const float *__restrict__ aug_mask,
                                                const float *__restrict__ aug_noise, int aug_ldn, int bx, int b, int gx) {
  const float *s = x + (size_t)b * ldxs;
  float *d = X + (size_t)b * ldx;
  float *d2 = X2 ? X2 + (size_t)b * ldx : nullptr;
  const int stride = gx * blockDim.x, t0 = bx * blockDim.x + threadIdx.x;
  if (AUG) {   // augmented real rows (same expression as aug_add_kernel on the staged row)
    const float m = aug_mask[b];
    const float *nz = aug_noise + (size_t)b * aug_ldn;
    for (int g = t0; g < G; g += stride) {
      const float v = s[g] + m * nz[g];
      d[g] = v;
      if (d2) d2[g] = v;
    }
  } else if (vec) {
    const int g4 = G >> 2;
    for (int q = t0; q < g4; q += stride) {
      const float4 v = reinterpret_cast<const float4 *>(s)[q];
      reinterpret_cast<float4 *>(d)[q] = v;
      if (d2) reinterpret_cast<float4 *>(d2)[q] = v;
    }
    for (int g = (g4 << 2) + t0; g < G; g += stride) {
      d[g] = s[g];
      if (d2) d2[g] = s[g];
    }
  } else {
    for (int g = t0; g < G; g += stride) {
      d[g] = s[g];
      if (d2) d2[g] = s[g];
    }
  }
  if (bx == 0 && b == 0) {
    for (int i = threadIdx.x; i < n_cat; i += blockDim.x) scat[i] = cat[i];
    for (int i = threadIdx.x; i < n_num; i += blockDim.x) snum[i] = num[i];
    if (threadIdx.x == 0) {
      *rngpos = rng_offset;
      *loss_slot = losses_out;
    }
    if ((int)threadIdx.x < lrs.n) lr_dev[threadIdx.x] = lrs.v[threadIdx.x];
  }
}

template <bool AUG>
__global__ void __launch_bounds__(256) iter_stage_kernel(const float *__restrict__ x, int ldxs,
                                                         float *__restrict__ X, float *__restrict__ X2, int ldx, int G, int vec,
                                                         const int *__restrict__ cat, int *__restrict__ scat, int n_cat,
                                                         const float *__restrict__ num, float *__restrict__ snum, int n_num,
                                                         unsigned long long *rngpos, unsigned long long rng_offset,
                                                         float **loss_slot, float *losses_out,
                                                         const agx::LrTable lrs, float *__restrict__ lr_dev,
                                                         const float *__restrict__ aug_mask,
                                                         const float *__restrict__ aug_noise, int aug_ldn) {
  iter_stage_body<AUG>(x, ldxs, X, X2, ldx, G, vec, cat, scat, n_cat, num, snum, n_num, rngpos, rng_offset, loss_slot,
                       losses_out, lrs, lr_dev, aug_mask, aug_noise, aug_ldn, (int)blockIdx.x, (int)blockIdx.y, (int)gridDim.x);
}

// staging rows [0, B) of the grid's y axis, the iteration's draws (iter_draws_body) in the block rows behind them:
// the draws read the stream position and the covariates from the launch arguments, not from what the staging rows
// write, so the two parts share nothing inside the launch
template <bool AUG>
__global__ void __launch_bounds__(256) iter_stage_draws_kernel(const float *__restrict__ x, int ldxs,
                                                               float *__restrict__ X, float *__restrict__ X2, int ldx, int G, int vec,
                                                               const int *__restrict__ cat, int *__restrict__ scat, int n_cat,
                                                               const float *__restrict__ num, float *__restrict__ snum, int n_num,
                                                               unsigned long long *rngpos, unsigned long long rng_offset,
                                                               float **loss_slot, float *losses_out,
                                                               const agx::LrTable lrs, float *__restrict__ lr_dev,
                                                               const float *__restrict__ aug_mask,
                                                               const float *__restrict__ aug_noise, int aug_ldn, int B, int gxs,
                                                               const agx::StageDraws sd) {
  if ((int)blockIdx.y < B) {
    if ((int)blockIdx.x < gxs)
      iter_stage_body<AUG>(x, ldxs, X, X2, ldx, G, vec, cat, scat, n_cat, num, snum, n_num, rngpos, rng_offset, loss_slot,
                           losses_out, lrs, lr_dev, aug_mask, aug_noise, aug_ldn, (int)blockIdx.x, (int)blockIdx.y, gxs);
    return;
  }
  if ((int)blockIdx.x < sd.gx)
    iter_draws_body(sd.d, sd.seed, rng_offset, sd.e, sd.tables, num, sd.n_num, cat, sd.ld_cat, sd.B, sd.rows_cov, sd.zcols,
                    sd.gin0, sd.ldg0, sd.ncov, sd.cov_row, sd.e2, sd.tables2, sd.X, sd.ldx, sd.xcol0, sd.xrows, sd.ncov2,
                    (int)blockIdx.x, (int)blockIdx.y - B, sd.gx);
}

inline int grid_for(long long n, int per_block = 256, int cap = 4096) {
  long long b = (n + per_block - 1) / per_block;
  if (b < 1) b = 1;
  if (b > cap) b = cap;
  return (int)b;
}

}  // namespace

extern "C" int agx_embed_concat_fwd(void *stream, const agx_embed_t *e, const float *tables,
                                    const float *lead, int32_t n_lead, int32_t ld_lead,
                                    const float *num, int32_t n_num, int32_t ld_num,
                                    const int32_t *cat, int32_t ld_cat, int32_t B, int32_t row_mod,
                                    float *out, int32_t ldo) {
  AGX_CHECK_ARG(e && out, "agx_embed_concat_fwd: null argument");
  AGX_CHECK_ARG(e->n_cat >= 0 && e->n_cat <= AGX_MAX_CAT, "agx_embed_concat_fwd: n_cat %d", e->n_cat);
  AGX_CHECK_ARG(n_lead == 0 || lead, "agx_embed_concat_fwd: lead missing");
  AGX_CHECK_ARG(n_num == 0 || num, "agx_embed_concat_fwd: num missing");
  AGX_CHECK_ARG(e->n_cat == 0 || (cat && tables && ld_cat >= e->n_cat), "agx_embed_concat_fwd: cat/tables missing");
  int width = n_lead + n_num;
  for (int i = 0; i < e->n_cat; ++i) width += e->edim[i];
  AGX_CHECK_ARG(ldo >= width, "agx_embed_concat_fwd: ldo %d < width %d", ldo, width);
  if (B <= 0 || width == 0) return AGX_OK;
  hipLaunchKernelGGL(embed_concat_fwd_kernel, dim3(grid_for((long long)B * width)), dim3(256), 0,
                     agx::as_stream(stream), *e, tables, lead, n_lead, ld_lead, num, n_num, ld_num,
                     cat, ld_cat, B, row_mod, out, ldo, width);
  AGX_LAUNCH_CHECK();
  return AGX_OK;
}

extern "C" int agx_embed_bwd(void *stream, const agx_embed_t *e, float *grad_tables,
                             const float *dout, int32_t ldd, int32_t col0, const int32_t *cat,
                             int32_t ld_cat, int32_t B, int32_t row_mod, int32_t accumulate) {
  AGX_CHECK_ARG(e && e->n_cat >= 0 && e->n_cat <= AGX_MAX_CAT, "agx_embed_bwd: bad descriptor");
  if (e->n_cat == 0) return AGX_OK;
  AGX_CHECK_ARG(grad_tables && dout && cat && ld_cat >= e->n_cat, "agx_embed_bwd: null argument");
  int rows = 0;
  for (int i = 0; i < e->n_cat; ++i) rows += e->vocab[i];
  if (rows == 0) return AGX_OK;
  hipLaunchKernelGGL(embed_bwd_kernel, dim3(rows), dim3(256), 0, agx::as_stream(stream), *e,
                     grad_tables, dout, ldd, col0, cat, ld_cat, B < 0 ? 0 : B, row_mod, accumulate);
  AGX_LAUNCH_CHECK();
  return AGX_OK;
}

extern "C" int agx_gp_interp(void *stream, const float *real, int32_t ldr, const float *fake,
                             int32_t ldf, const float *alpha, float *inter, int32_t ldi, int32_t B,
                             int32_t G) {
  AGX_CHECK_ARG(real && fake && alpha && inter, "agx_gp_interp: null argument");
  AGX_CHECK_ARG(ldr >= G && ldf >= G && ldi >= G, "agx_gp_interp: leading dimension < G");
  if (B <= 0 || G <= 0) return AGX_OK;
  hipLaunchKernelGGL(gp_interp_kernel, dim3(std::min(agx::cdiv(G, 256), 64), B), dim3(256), 0,
                     agx::as_stream(stream), real, ldr, fake, ldf, alpha, inter, ldi, B, G);
  AGX_LAUNCH_CHECK();
  return AGX_OK;
}

extern "C" int agx_aug_add(void *stream, float *x, int32_t ldx, const float *mask,
                           const float *noise, int32_t ldn, int32_t row_bcast, int32_t B,
                           int32_t G) {
  AGX_CHECK_ARG(x && mask && noise, "agx_aug_add: null argument");
  AGX_CHECK_ARG(ldx >= G && (row_bcast || ldn >= G), "agx_aug_add: leading dimension < G");
  if (B <= 0 || G <= 0) return AGX_OK;
  hipLaunchKernelGGL(aug_add_kernel, dim3(std::min(agx::cdiv(G, 256), 64), B), dim3(256), 0,
                     agx::as_stream(stream), x, ldx, mask, noise, ldn, row_bcast, B, G);
  AGX_LAUNCH_CHECK();
  return AGX_OK;
}

extern "C" int agx_rowdot(void *stream, const float *A, int32_t lda, const float *w, const float *b,
                          float *out, int32_t R, int32_t H) {
  AGX_CHECK_ARG(A && w && out && lda >= H, "agx_rowdot: bad argument");
  if (R <= 0) return AGX_OK;
  hipLaunchKernelGGL(rowdot_kernel, dim3(agx::cdiv(R, 4)), dim3(256), 0, agx::as_stream(stream), A,
                     lda, w, b, out, R, H);
  AGX_LAUNCH_CHECK();
  return AGX_OK;
}

extern "C" int agx_delta_init(void *stream, const float *A, int32_t lda, const float *w,
                              float *delta, int32_t ldd, int32_t R, int32_t H, int32_t n0,
                              int32_t n1, float c0, float c1, float c2, float slope) {
  AGX_CHECK_ARG(A && w && delta && lda >= H && ldd >= H, "agx_delta_init: bad argument");
  if (R <= 0 || H <= 0) return AGX_OK;
  hipLaunchKernelGGL(delta_init_kernel, dim3(grid_for((long long)R * H)), dim3(256), 0,
                     agx::as_stream(stream), A, lda, w, delta, ldd, R, H, n0, n1, c0, c1, c2, slope);
  AGX_LAUNCH_CHECK();
  return AGX_OK;
}

extern "C" int agx_critic_head(void *stream, const float *A, int32_t lda, const float *w,
                               const float *b, float *out, float *delta, int32_t ldd, int32_t R,
                               int32_t H, int32_t n0, int32_t n1, float c0, float c1, float c2,
                               float slope) {
  AGX_CHECK_ARG(A && w && b && out && delta && lda >= H && ldd >= H, "agx_critic_head: bad argument");
  if (R <= 0 || H <= 0) return AGX_OK;
  hipLaunchKernelGGL(critic_head_kernel, dim3(agx::cdiv(R, 4)), dim3(256), 0, agx::as_stream(stream),
                     A, lda, w, b, out, delta, ldd, R, H, n0, n1, c0, c1, c2, slope);
  AGX_LAUNCH_CHECK();
  return AGX_OK;
}

// the same with 16-byte accesses on both sides (4 loads + 4 stores per thread instead of 16 + 16): needs ldw, ldx
// multiples of 4, 16-byte-aligned bases and ldx >= G rounded up to 4 (a tile's last float4 may reach up to 3 columns past
// G: zeros, into covariate / padding columns that hold zeros anyway)
__global__ void __launch_bounds__(256) wt_rows4_kernel(const float *__restrict__ W, int ldw, int G, int H,
                                                       float *__restrict__ Xt, int ldx, int ncov,
                                                       const float *__restrict__ bvec, float *__restrict__ bpart) {
  __shared__ float t[64][65];
  __shared__ float bp[16][64];
  const int c4 = threadIdx.x & 15, rq = threadIdx.x >> 4;
  const int g0 = blockIdx.x * 64, h0 = blockIdx.y * 64;
  float bs[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int r = rq + 16 * q, g = g0 + r, hh = h0 + 4 * c4;
    float4 w = make_float4(0.f, 0.f, 0.f, 0.f);
    if (g < G && hh < H) w = *reinterpret_cast<const float4 *>(W + (size_t)g * ldw + hh);   // (columns H.. of the last float4: padding of W, masked)
    if (hh + 1 >= H) w.y = 0.f;
    if (hh + 2 >= H) w.z = 0.f;
    if (hh + 3 >= H) w.w = 0.f;
    t[r][4 * c4] = w.x; t[r][4 * c4 + 1] = w.y; t[r][4 * c4 + 2] = w.z; t[r][4 * c4 + 3] = w.w;
    if (bvec) {
      const float b = g < G ? bvec[g] : 0.f;
      bs[0] += b * w.x; bs[1] += b * w.y; bs[2] += b * w.z; bs[3] += b * w.w;
    }
  }
  if (bvec) {
#pragma unroll
    for (int e = 0; e < 4; ++e) bp[rq][4 * c4 + e] = bs[e];
  }
  __syncthreads();
  if (bvec && threadIdx.x < 64 && h0 + (int)threadIdx.x < H) {
    float v = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) v += bp[k][threadIdx.x];
    bpart[(size_t)blockIdx.x * H + h0 + threadIdx.x] = v;
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int r = rq + 16 * q, hh = h0 + r, g = g0 + 4 * c4;
    if (hh < H && g < G)
      *reinterpret_cast<float4 *>(Xt + (size_t)hh * ldx + g) = make_float4(t[4 * c4][r], t[4 * c4 + 1][r], t[4 * c4 + 2][r], t[4 * c4 + 3][r]);
  }
  if (blockIdx.x == 0)
    for (int q = threadIdx.x; q < 64 * ncov; q += 256) {
      const int hh = h0 + q / ncov;
      if (hh < H) Xt[(size_t)hh * ldx + G + q % ncov] = 0.f;
    }
}

int agx::wt_rows(void *stream, const float *W, int ldw, int G, int H, float *Xt, int ldx, int ncov, const float *bvec,
                 float *bpart) {
  AGX_CHECK_ARG(W && Xt && G > 0 && H > 0 && ldw >= H && ldx >= G + ncov && ncov >= 0 && (!bvec || bpart), "wt_rows: bad argument");
  agx::ProfScope prof(agx::as_stream(stream), AGX_K_DISC_DX, 0.0, 8.0 * G * (double)H);
  // (the last float4 of a row may reach 3 columns past G: they must be columns this call zeroes anyway, or padding)
  const bool v4 = ldw % 4 == 0 && ldx % 4 == 0 && agx::aligned16(W) && agx::aligned16(Xt) && ldx >= ((G + 3) & ~3) &&
                  (ncov == 0 ? false : ((G + 3) & ~3) - G <= ncov);
  if (v4)
    hipLaunchKernelGGL(wt_rows4_kernel, dim3(agx::cdiv(G, 64), agx::cdiv(H, 64)), dim3(256), 0, agx::as_stream(stream), W, ldw, G, H,
                       Xt, ldx, ncov, bvec, bpart);
  else
  hipLaunchKernelGGL(wt_rows_kernel, dim3(agx::cdiv(G, 64), agx::cdiv(H, 64)), dim3(256), 0, agx::as_stream(stream), W, ldw, G, H,
                     Xt, ldx, ncov, bvec, bpart);
  AGX_LAUNCH_CHECK();
  return AGX_OK;
}

int agx::ff_stage(void *stream, const float *Wout, int ldw, const float *bout, int HL, int G, int ncov, int he, float *Xhd,
                  int ldx, int rows) {
  AGX_CHECK_ARG(Wout && bout && Xhd && HL > 0 && G > 0 && ncov >= 0 && ldw >= G && ldx >= G + ncov && he >= 1 + ncov &&
                    rows >= he + HL, "ff_stage: bad argument");
  hipLaunchKernelGGL(ff_stage_kernel, dim3(std::min(agx::cdiv(G + ncov, 256), 16), rows), dim3(256), 0, agx::as_stream(stream),
                     Wout, ldw, bout, HL, G, ncov, he, Xhd, ldx);
  AGX_LAUNCH_CHECK();
  return AGX_OK;
}

int agx::ff_hx(void *stream, const float *hrows, int ldh, int HL, float *X, int ldx, int G, int ncov, int he, float *Hx,
               int ldhx, int B, const float *bpart, int ntile, const float *W1, int ldw, int H1, float *P, int ldp,
               const agx_embed_t *e, const float *tables, const float *num, int n_num, const int32_t *cat) {
  AGX_CHECK_ARG(hrows && X && Hx && B > 0 && ldh >= HL && he >= 1 + ncov && ldhx >= he + HL && bpart && ntile > 0 && W1 && P &&
                    ldw >= H1 && ldp >= H1 && e && tables && (n_num == 0 || num) && (e->n_cat == 0 || cat), "ff_hx: bad argument");
  int ec = n_num;
  for (int i = 0; i < e->n_cat; ++i) ec += e->edim[i];
  AGX_CHECK_ARG(ec == ncov, "ff_hx: %d covariate columns, the tables make %d", ncov, ec);
  hipLaunchKernelGGL(ff_hx_kernel, dim3(B + he + agx::cdiv(H1, 16)), dim3(256), 0, agx::as_stream(stream), hrows, ldh, HL, X, ldx, G, ncov, he, Hx,
                     ldhx, B, bpart, ntile, W1, ldw, H1, P, ldp, *e, tables, num, n_num, cat);
  AGX_LAUNCH_CHECK();
  return AGX_OK;
}

extern "C" int agx_gp_finalize(void *stream, const float *rowsq, int32_t rowsq_ld, int32_t nslab,
                               float *slopes, float *fscale, float *d1, int32_t ldd, int32_t H1,
                               float *xcov, int32_t ldx, int32_t ncov, int32_t B, float coef) {
  AGX_CHECK_ARG(rowsq && slopes && fscale && d1 && rowsq_ld >= nslab && ldd >= H1,
                "agx_gp_finalize: bad argument");
  AGX_CHECK_ARG(ncov == 0 || xcov, "agx_gp_finalize: xcov missing");
  if (B <= 0) return AGX_OK;
  hipLaunchKernelGGL(gp_finalize_kernel, dim3(B), dim3(256), 0, agx::as_stream(stream), rowsq,
                     rowsq_ld, nslab, slopes, fscale, d1, ldd, H1, xcov, ldx, ncov, coef);
  AGX_LAUNCH_CHECK();
  return AGX_OK;
}

extern "C" int agx_gp_rownorm_scale(void *stream, float *g, int32_t ldg, float *slopes, int32_t B,
                                    int32_t G, float coef) {
  AGX_CHECK_ARG(g && slopes && ldg >= G, "agx_gp_rownorm_scale: bad argument");
  if (B <= 0) return AGX_OK;
  hipLaunchKernelGGL(gp_rownorm_scale_kernel, dim3(B), dim3(256), 0, agx::as_stream(stream), g, ldg,
                     slopes, G, coef);
  AGX_LAUNCH_CHECK();
  return AGX_OK;
}

extern "C" int agx_wgan_loss(void *stream, const float *out, int32_t n0, int32_t n1,
                             const float *slopes, int32_t n_slopes, float inv_b, float gp_weight,
                             float fake_sign, float *losses) {
  AGX_CHECK_ARG(out && losses && (n_slopes == 0 || slopes), "agx_wgan_loss: null argument");
  hipLaunchKernelGGL(wgan_loss_kernel, dim3(1), dim3(256), 0, agx::as_stream(stream), out, n0, n1,
                     slopes, n_slopes, inv_b, gp_weight, fake_sign, losses);
  AGX_LAUNCH_CHECK();
  return AGX_OK;
}

extern "C" int agx_colsum(void *stream, const float *A, int32_t lda, int32_t r0, int32_t r1,
                          int32_t H, float *out, int32_t accumulate) {
  AGX_CHECK_ARG(A && out && lda >= H && r0 >= 0 && r1 >= r0, "agx_colsum: bad argument");
  if (H <= 0) return AGX_OK;
  hipLaunchKernelGGL(colsum_kernel, dim3(agx::cdiv(H, 64)), dim3(1024), 0, agx::as_stream(stream), A,
                     lda, r0, r1, H, out, accumulate, 0, 0, 0.f, 0.f, 0.f, 0, (float *)nullptr);
  AGX_LAUNCH_CHECK();
  return AGX_OK;
}

extern "C" int agx_colsum_weighted(void *stream, const float *A, int32_t lda, int32_t R, int32_t H,
                                   int32_t n0, int32_t n1, float c0, float c1, float c2, float *out,
                                   float *bias_out) {
  AGX_CHECK_ARG(A && out && lda >= H && R >= 0, "agx_colsum_weighted: bad argument");
  if (H <= 0) return AGX_OK;
  hipLaunchKernelGGL(colsum_kernel, dim3(agx::cdiv(H, 64)), dim3(1024), 0, agx::as_stream(stream), A,
                     lda, 0, R, H, out, 0, n0, n1, c0, c1, c2, 1, bias_out);
  AGX_LAUNCH_CHECK();
  return AGX_OK;
}

extern "C" int agx_opt_rmsprop(void *stream, float *w, const float *g, float *rms, int64_t n,
                               float lr, float rho, float eps, float grad_scale) {
  AGX_CHECK_ARG(w && g && rms && n >= 0, "agx_opt_rmsprop: bad argument");
  AGX_CHECK_ARG(agx::aligned16(w) && agx::aligned16(g) && agx::aligned16(rms),
                "agx_opt_rmsprop: buffers must be 16-byte aligned");
  if (n == 0) return AGX_OK;
  agx::ProfScope prof(agx::as_stream(stream), AGX_K_OPT_RMSPROP, 6.0 * n, 20.0 * n);
  hipLaunchKernelGGL(rmsprop_kernel, dim3(grid_for(n / 4 + 256, 256, 2048)), dim3(256), 0,
                     agx::as_stream(stream), w, g, rms, (long long)n, lr, rho, eps, grad_scale, 0ll, 0ll);
  AGX_LAUNCH_CHECK();
  return AGX_OK;
}

int agx::opt_rmsprop_hole(void *stream, float *w, const float *g, float *rms, int64_t n, int64_t hole_lo,
                          int64_t hole_hi, float lr, float rho, float eps) {
  AGX_CHECK_ARG(w && g && rms && n >= 0, "opt_rmsprop_hole: bad argument");
  AGX_CHECK_ARG(hole_lo >= 0 && hole_lo <= hole_hi && hole_hi <= (n & ~(int64_t)3) && hole_lo % 4 == 0 && hole_hi % 4 == 0,
                "opt_rmsprop_hole: hole [%lld, %lld) must be 4-aligned inside [0, %lld)", (long long)hole_lo,
                (long long)hole_hi, (long long)n);
  const int64_t rest = n - (hole_hi - hole_lo);
  if (rest == 0) return AGX_OK;
  agx::ProfScope prof(agx::as_stream(stream), AGX_K_OPT_RMSPROP, 6.0 * rest, 20.0 * rest);
  hipLaunchKernelGGL(rmsprop_kernel, dim3(grid_for(rest / 4 + 256, 256, 2048)), dim3(256), 0,
                     agx::as_stream(stream), w, g, rms, (long long)n, lr, rho, eps, 1.f, (long long)(hole_lo / 4),
                     (long long)((hole_hi - hole_lo) / 4));
  AGX_LAUNCH_CHECK();
  return AGX_OK;
}

extern "C" int agx_opt_adam(void *stream, float *w, const float *g, float *m, float *v, int64_t n,
                            float lr_t, float beta1, float beta2, float eps, float grad_scale) {
  AGX_CHECK_ARG(w && g && m && v && n >= 0, "agx_opt_adam: bad argument");
  if (n == 0) return AGX_OK;
  agx::ProfScope prof(agx::as_stream(stream), AGX_K_OPT_ADAM, 10.0 * n, 28.0 * n);
  hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n, 256, 2048)), dim3(256), 0,
                     agx::as_stream(stream), w, g, m, v, (long long)n, lr_t, beta1, beta2, eps,
                     grad_scale, (const float *)nullptr);
  AGX_LAUNCH_CHECK();
  return AGX_OK;
}

int agx::opt_adam_dev_lr(void *stream, float *w, const float *g, float *m, float *v, int64_t n, const float *lr_dev,
                         float beta1, float beta2, float eps) {
  AGX_CHECK_ARG(w && g && m && v && lr_dev && n >= 0, "opt_adam_dev_lr: bad argument");
  if (n == 0) return AGX_OK;
  agx::ProfScope prof(agx::as_stream(stream), AGX_K_OPT_ADAM, 10.0 * n, 28.0 * n);
  hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n, 256, 2048)), dim3(256), 0, agx::as_stream(stream), w, g, m, v,
                     (long long)n, 0.f, beta1, beta2, eps, 1.f, lr_dev);
  AGX_LAUNCH_CHECK();
  return AGX_OK;
}

extern "C" int agx_rng_normal(void *stream, float *out, int64_t n, uint64_t seed, uint64_t offset) {
  AGX_CHECK_ARG(out && n >= 0, "agx_rng_normal: bad argument");
  if (n == 0) return AGX_OK;
  hipLaunchKernelGGL(rng_kernel<true>, dim3(grid_for((n + 3) / 4)), dim3(256), 0,
                     agx::as_stream(stream), out, (long long)n, (unsigned long long)seed,
                     (unsigned long long)offset, (const unsigned long long *)nullptr);
  AGX_LAUNCH_CHECK();
  return AGX_OK;
}

extern "C" int agx_rng_uniform(void *stream, float *out, int64_t n, uint64_t seed,
                               uint64_t offset) {
  AGX_CHECK_ARG(out && n >= 0, "agx_rng_uniform: bad argument");
  if (n == 0) return AGX_OK;
  hipLaunchKernelGGL(rng_kernel<false>, dim3(grid_for((n + 3) / 4)), dim3(256), 0,
                     agx::as_stream(stream), out, (long long)n, (unsigned long long)seed,
                     (unsigned long long)offset, (const unsigned long long *)nullptr);
  AGX_LAUNCH_CHECK();
  return AGX_OK;
}

extern "C" int agx_rng_at(void *stream, float *out, int64_t n, uint64_t seed, const uint64_t *base_dev,
                          uint64_t sub_offset, int32_t normal) {
  AGX_CHECK_ARG(out && base_dev && n >= 0, "agx_rng_at: bad argument");
  if (n == 0) return AGX_OK;
  if (normal)
    hipLaunchKernelGGL(rng_kernel<true>, dim3(grid_for((n + 3) / 4)), dim3(256), 0,
                       agx::as_stream(stream), out, (long long)n, (unsigned long long)seed,
                       (unsigned long long)sub_offset, (const unsigned long long *)base_dev);
  else
    hipLaunchKernelGGL(rng_kernel<false>, dim3(grid_for((n + 3) / 4)), dim3(256), 0,
                       agx::as_stream(stream), out, (long long)n, (unsigned long long)seed,
                       (unsigned long long)sub_offset, (const unsigned long long *)base_dev);
  AGX_LAUNCH_CHECK();
  return AGX_OK;
}

extern "C" int agx_set_u64(void *stream, uint64_t *dst, uint64_t value) {
  AGX_CHECK_ARG(dst, "agx_set_u64: null");
  hipLaunchKernelGGL(set_u64_kernel, dim3(1), dim3(1), 0, agx::as_stream(stream),
                     (unsigned long long *)dst, (unsigned long long)value);
  AGX_LAUNCH_CHECK();
  return AGX_OK;
}

extern "C" int agx_rng_multi(void *stream, uint64_t seed, const uint64_t *base_dev, int32_t n_seg,
                             const agx_rng_seg_t *segs) {
  AGX_CHECK_ARG(n_seg >= 0 && n_seg <= AGX_MAX_RNG_SEGS && (n_seg == 0 || segs), "agx_rng_multi: bad segment list");
  if (n_seg == 0) return AGX_OK;
  RngSegs rs;
  rs.n = n_seg;
  long long nmax = 0;
  for (int i = 0; i < n_seg; ++i) {
    AGX_CHECK_ARG(segs[i].out && segs[i].n >= 0 && segs[i].skip >= 0 && (segs[i].sub_offset & 3) == 0,
                  "agx_rng_multi: segment %d", i);
    rs.seg[i] = segs[i];
    nmax = std::max<long long>(nmax, segs[i].n + 3);
  }
  for (int i = n_seg; i < AGX_MAX_RNG_SEGS; ++i) rs.seg[i] = segs[0];
  hipLaunchKernelGGL(rng_multi_kernel, dim3(grid_for((nmax + 3) / 4, 256, 256), n_seg), dim3(256), 0,
                     agx::as_stream(stream), rs, (unsigned long long)seed,
                     (const unsigned long long *)base_dev);
  AGX_LAUNCH_CHECK();
  return AGX_OK;
}

int agx::iter_draws_desc(uint64_t seed, const agx::IterDraws &d, const agx_embed_t *e, const float *tables, int n_num,
                         int ld_cat, int B, int rows_cov, int zcols, float *gin0, int ldg0, const agx::CriticCov *xc,
                         agx::StageDraws *out) {
  AGX_CHECK_ARG(out && d.n > 0 && d.n <= AGX_MAX_RNG_SEGS && e && gin0 && B > 0 && rows_cov >= 0, "iter_draws_desc: bad argument");
  out->on = 0;
  long long nmax = 0;
  for (int i = 0; i < d.n; ++i) {
    AGX_CHECK_ARG(d.seg[i].out && d.seg[i].n >= 0 && d.seg[i].skip >= 0 && (d.seg[i].sub_offset & 3) == 0 &&
                      (d.seg[i].cols == 0 || d.seg[i].ld >= d.seg[i].cols), "iter_draws_desc: segment %d", i);
    nmax = std::max<long long>(nmax, d.seg[i].n + 3);
  }
  int ncov = n_num;
  for (int i = 0; i < e->n_cat; ++i) ncov += e->edim[i];
  AGX_CHECK_ARG(ldg0 >= zcols + ncov && (e->n_cat == 0 || (tables && ld_cat >= e->n_cat)), "iter_draws_desc: generator input description");
  const int cov_row = (rows_cov > 0 && ncov > 0) ? 1 : 0;
  agx_embed_t e2 = {};
  int ncov2 = 0, x_row = 0;
  if (xc && xc->rows > 0) {
    AGX_CHECK_ARG(xc->e && xc->X && xc->e->n_cat == e->n_cat && (xc->e->n_cat == 0 || xc->tables),
                  "iter_draws_desc: critic covariate description");
    e2 = *xc->e;
    ncov2 = n_num;
    for (int i = 0; i < e2.n_cat; ++i) ncov2 += e2.edim[i];
    AGX_CHECK_ARG(xc->ldx >= xc->col0 + ncov2, "iter_draws_desc: critic input too narrow");
    x_row = ncov2 > 0 ? 1 : 0;
  }
  const long long cov_work = std::max<long long>((long long)rows_cov * ncov, x_row ? (long long)xc->rows * ncov2 : 0);
  out->d = d; out->seed = seed; out->e = *e; out->e2 = e2;
  out->tables = tables; out->tables2 = x_row ? xc->tables : nullptr;
  out->n_num = n_num; out->ld_cat = ld_cat; out->B = B; out->rows_cov = rows_cov; out->zcols = zcols; out->ldg0 = ldg0;
  out->ncov = ncov; out->cov_row = cov_row;
  out->ldx = x_row ? xc->ldx : 0; out->xcol0 = x_row ? xc->col0 : 0; out->xrows = x_row ? xc->rows : 0; out->ncov2 = ncov2;
  out->gin0 = gin0; out->X = x_row ? xc->X : nullptr;
  out->gx = grid_for(std::max<long long>((nmax + 3) / 4, cov_work), 256, 256);   // (the grid iter_draws launches)
  out->rows = d.n + cov_row + x_row;
  out->on = 1;
  return AGX_OK;
}

int agx::iter_draws(void *stream, uint64_t seed, const uint64_t *base_dev, const agx::IterDraws &d, const agx_embed_t *e,
                    const float *tables, const float *num, int n_num, const int32_t *cat, int ld_cat, int B,
                    int rows_cov, int zcols, float *gin0, int ldg0, const agx::CriticCov *xc) {
  AGX_CHECK_ARG(d.n > 0 && d.n <= AGX_MAX_RNG_SEGS && e && gin0 && B > 0 && rows_cov >= 0, "iter_draws: bad argument");
  long long nmax = 0;
  for (int i = 0; i < d.n; ++i) {
    AGX_CHECK_ARG(d.seg[i].out && d.seg[i].n >= 0 && d.seg[i].skip >= 0 && (d.seg[i].sub_offset & 3) == 0 &&
                      (d.seg[i].cols == 0 || d.seg[i].ld >= d.seg[i].cols), "iter_draws: segment %d", i);
    nmax = std::max<long long>(nmax, d.seg[i].n + 3);
  }
  int ncov = n_num;
  for (int i = 0; i < e->n_cat; ++i) ncov += e->edim[i];
  AGX_CHECK_ARG(ldg0 >= zcols + ncov && (n_num == 0 || num) && (e->n_cat == 0 || (cat && tables && ld_cat >= e->n_cat)),
                "iter_draws: generator input description");
  const int cov_row = (rows_cov > 0 && ncov > 0) ? 1 : 0;
  agx_embed_t e2 = {};
  int ncov2 = 0, x_row = 0;
  if (xc && xc->rows > 0) {
    AGX_CHECK_ARG(xc->e && xc->X && xc->e->n_cat == e->n_cat && (xc->e->n_cat == 0 || xc->tables),
                  "iter_draws: critic covariate description");
    e2 = *xc->e;
    ncov2 = n_num;
    for (int i = 0; i < e2.n_cat; ++i) ncov2 += e2.edim[i];
    AGX_CHECK_ARG(xc->ldx >= xc->col0 + ncov2, "iter_draws: critic input too narrow");
    x_row = ncov2 > 0 ? 1 : 0;
  }
  // (blocks along x sized for the covariate rows as well: with one block the E. coli shape - 7 tables, 192 x 24 covariate
  //  values, a dependent cat -> table lookup each - took 20 us for this launch)
  const long long cov_work = std::max<long long>((long long)rows_cov * ncov, x_row ? (long long)xc->rows * ncov2 : 0);
  hipLaunchKernelGGL(iter_draws_kernel, dim3(grid_for(std::max<long long>((nmax + 3) / 4, cov_work), 256, 256), d.n + cov_row + x_row), dim3(256), 0,
                     agx::as_stream(stream), d, (unsigned long long)seed, (const unsigned long long *)base_dev, *e, tables,
                     num, n_num, cat, ld_cat, B, rows_cov, zcols, gin0, ldg0, ncov, cov_row, e2,
                     x_row ? xc->tables : nullptr, x_row ? xc->X : nullptr, x_row ? xc->ldx : 0, x_row ? xc->col0 : 0,
                     x_row ? xc->rows : 0, ncov2);
  AGX_LAUNCH_CHECK();
  return AGX_OK;
}

extern "C" int agx_stack_rows(void *stream, const agx_embed_t *e, const float *tables, const float *src,
                              int32_t lds, int32_t dst_row0, int32_t src_rows, const float *num, int32_t n_num,
                              int32_t ld_num, const int32_t *cat, int32_t ld_cat, int32_t B,
                              int32_t rows_cov, int32_t G, float *X, int32_t ldx, const float *aug_mask,
                              const float *aug_noise, int32_t aug_ldn) {
  AGX_CHECK_ARG(e && X && B > 0 && G > 0 && rows_cov >= 0 && dst_row0 >= 0, "agx_stack_rows: bad argument");
  AGX_CHECK_ARG(e->n_cat >= 0 && e->n_cat <= AGX_MAX_CAT, "agx_stack_rows: n_cat %d", e->n_cat);
  AGX_CHECK_ARG(!src || lds >= G, "agx_stack_rows: lds < G");
  AGX_CHECK_ARG(!aug_mask || aug_noise, "agx_stack_rows: aug noise missing");
  AGX_CHECK_ARG(!src || (src_rows > 0 && src_rows <= 65535), "agx_stack_rows: src_rows %d outside (0, 65535]", src_rows);
  AGX_CHECK_ARG(!aug_mask || src_rows <= B, "agx_stack_rows: augmentation needs src_rows <= B");
  int ncov = n_num;
  for (int i = 0; i < e->n_cat; ++i) ncov += e->edim[i];
  AGX_CHECK_ARG(ldx >= G + ncov, "agx_stack_rows: ldx %d < G + covariates %d", ldx, G + ncov);
  AGX_CHECK_ARG(n_num == 0 || num, "agx_stack_rows: num missing");
  AGX_CHECK_ARG(e->n_cat == 0 || (cat && tables && ld_cat >= e->n_cat), "agx_stack_rows: cat/tables missing");
  const int gene_blocks = src ? std::min(agx::cdiv(G, 256), 32) : 0;
  const int cov_blocks = (rows_cov > 0 && ncov > 0) ? 1 : 0;
  if (gene_blocks + cov_blocks == 0) return AGX_OK;
  hipLaunchKernelGGL(stack_rows_kernel, dim3(gene_blocks + cov_blocks, src ? src_rows : std::min(std::max(rows_cov, 1), 64)),
                     dim3(256), 0, agx::as_stream(stream), *e, tables, src, lds, dst_row0, src_rows, num, n_num, ld_num, cat,
                     ld_cat, B, cov_blocks ? rows_cov : 0, G, X, ldx, aug_mask, aug_noise, aug_ldn, ncov,
                     gene_blocks);
  AGX_LAUNCH_CHECK();
  return AGX_OK;
}

extern "C" int agx_small_grads(void *stream, const agx_small_grads_t *j) {
  return agx::small_grads_run(stream, j, agx::LossCopy{nullptr, nullptr, 0});
}

// blocks of a product job: four tiles per block, or one where the block's four waves split a long K (tn_split_k)
static int tn_blocks(const agx::TnJobs::Job &t) {
  const int tiles = agx::cdiv(t.M, 32) * agx::cdiv(t.N, 32);
  return t.K > 256 ? tiles : agx::cdiv(tiles, 4);
}

int agx::small_grads_run(void *stream, const agx_small_grads_t *j, const agx::LossCopy &lc, const agx::TnJobs *tn) {
  AGX_CHECK_ARG(j && j->n_colsum >= 0 && j->n_colsum <= AGX_MAX_LAYERS, "agx_small_grads: bad job list");
  int blocks = 0;
  agx::TnJobs tj;
  tj.n = 0;
  if (tn) {
    AGX_CHECK_ARG(tn->n >= 0 && tn->n <= AGX_MAX_LAYERS, "agx_small_grads: bad product list");
    tj = *tn;
    for (int q = 0; q < tj.n; ++q) {
      const agx::TnJobs::Job &t = tj.job[q];
      AGX_CHECK_ARG(t.A && t.B && t.C && t.M > 0 && t.N > 0 && t.K > 0 && t.lda >= t.M && t.ldb >= t.N && t.ldc >= t.N &&
                        t.K <= 2048 && (int64_t)agx::cdiv(t.M, 32) * agx::cdiv(t.N, 32) <= 256, "agx_small_grads: product job %d", q);
      blocks += tn_blocks(t);
    }
  }
  for (int c = 0; c < j->n_colsum; ++c) {
    AGX_CHECK_ARG(j->cs_A[c] && j->cs_out[c] && j->cs_lda[c] >= j->cs_H[c] && j->cs_r0[c] >= 0 &&
                      j->cs_r1[c] >= j->cs_r0[c] && j->cs_H[c] > 0, "agx_small_grads: colsum job %d", c);
    blocks += agx::cdiv(j->cs_H[c], 64);
  }
  if (j->w_A) {
    AGX_CHECK_ARG(j->w_out && j->w_lda >= j->w_H && j->w_H > 0 && j->w_R >= 0, "agx_small_grads: weighted colsum job");
    blocks += agx::cdiv(j->w_H, 64);
  }
  if (j->emb_grad) {
    AGX_CHECK_ARG(j->emb.n_cat > 0 && j->emb.n_cat <= AGX_MAX_CAT && j->emb_D && j->emb_W && j->emb_cat &&
                      j->emb_B > 0 && j->emb_rows >= 0 && j->emb_ldd >= j->emb_H && j->emb_ldw >= j->emb_H &&
                      j->emb_ld_cat >= j->emb.n_cat && j->emb_H <= 2048, "agx_small_grads: embedding job");
    for (int i = 0; i < j->emb.n_cat; ++i) {
      blocks += j->emb.vocab[i];
    }
  }
  if (j->l_losses) {
    AGX_CHECK_ARG(j->l_out && (j->l_ns == 0 || j->l_slopes), "agx_small_grads: loss job");
    blocks += 1;
  }
  if (blocks == 0) return AGX_OK;
  AGX_CHECK_ARG(!lc.dst_slot || (j->l_losses && lc.src && lc.n >= 0), "agx_small_grads: loss snapshot without a loss job");
  // block -> job plan, longest jobs first (see SmallPlan)
  SmallPlan plan = {};
  int nseg = 0, at = 0;
  auto seg = [&](int kind, int idx, int coff, int nb) {
    if (nb <= 0) return;
    plan.start[nseg] = at;
    plan.kind[nseg] = (signed char)kind;
    plan.idx[nseg] = (signed char)idx;
    plan.coff[nseg] = (short)coff;
    ++nseg;
    at += nb;
  };
  static_assert(2 * AGX_MAX_LAYERS + AGX_MAX_CAT + 2 <= 36, "SmallPlan segment table too small");
  if (j->emb_grad) {
    int coff = j->emb_col0;
    for (int i = 0; i < j->emb.n_cat; ++i) {
      AGX_CHECK_ARG(coff <= 32767, "agx_small_grads: embedding kernel row %d beyond the plan's range", coff);
      seg(3, i, coff, j->emb.vocab[i]);
      coff += j->emb.edim[i];
    }
  }
  for (int q = 0; q < tj.n; ++q) seg(0, q, 0, tn_blocks(tj.job[q]));
  if (j->l_losses) seg(4, 0, 0, 1);
  if (j->w_A) seg(2, 0, 0, agx::cdiv(j->w_H, 64));
  for (int c = 0; c < j->n_colsum; ++c) seg(1, c, 0, agx::cdiv(j->cs_H[c], 64));
  plan.n_seg = nseg;
  plan.start[nseg] = at;
  AGX_CHECK_ARG(at == blocks, "agx_small_grads: plan covers %d of %d blocks", at, blocks);
  hipLaunchKernelGGL(small_grads_kernel, dim3(blocks), dim3(1024), 0, agx::as_stream(stream), plan, *j, lc, tj);
  AGX_LAUNCH_CHECK();
  return AGX_OK;
}

int agx::iter_stage(void *stream, const float *x, int ldxs, float *X, float *X2, int ldx, int G, int B, const int32_t *cat,
                    int32_t *scat, int n_cat, const float *num, float *snum, int n_num, uint64_t *rngpos,
                    uint64_t rng_offset, float **loss_slot, float *losses_out, const agx::LrTable *lrs, float *lr_dev,
                    const float *aug_mask, const float *aug_noise, int aug_ldn, const agx::StageDraws *sd) {
  agx::LrTable none;
  none.n = 0;
  const int vec = agx::aligned16(x) && agx::aligned16(X) && (!X2 || agx::aligned16(X2)) && ldxs % 4 == 0 && ldx % 4 == 0;
  const int gx = std::max(1, std::min(8, agx::cdiv(G, 4 * 256)));
  AGX_CHECK_ARG(!aug_mask || aug_noise, "iter_stage: augmentation noise missing");
  if (sd && sd->on) {
    // staging rows + the iteration's draws in one launch (the draws read cat / num from the caller's arrays)
    AGX_CHECK_ARG(sd->B == B && (sd->n_num == 0 || num) && (sd->e.n_cat == 0 || cat), "iter_stage: draws description does not match");
    const dim3 grid(std::max(gx, sd->gx), B + sd->rows);
    if (aug_mask)
      hipLaunchKernelGGL(iter_stage_draws_kernel<true>, grid, dim3(256), 0, agx::as_stream(stream), x, ldxs, X, X2, ldx, G, vec,
                         cat, scat, n_cat, num, snum, n_num, reinterpret_cast<unsigned long long *>(rngpos),
                         (unsigned long long)rng_offset, loss_slot, losses_out, (lrs && lr_dev) ? *lrs : none, lr_dev,
                         aug_mask, aug_noise, aug_ldn, B, gx, *sd);
    else
      hipLaunchKernelGGL(iter_stage_draws_kernel<false>, grid, dim3(256), 0, agx::as_stream(stream), x, ldxs, X, X2, ldx, G, vec,
                         cat, scat, n_cat, num, snum, n_num, reinterpret_cast<unsigned long long *>(rngpos),
                         (unsigned long long)rng_offset, loss_slot, losses_out, (lrs && lr_dev) ? *lrs : none, lr_dev,
                         nullptr, nullptr, 0, B, gx, *sd);
    AGX_LAUNCH_CHECK();
    return AGX_OK;
  }
  if (aug_mask)
    hipLaunchKernelGGL(iter_stage_kernel<true>, dim3(gx, B), dim3(256), 0, agx::as_stream(stream), x, ldxs, X, X2, ldx, G, vec,
                       cat, scat, n_cat, num, snum, n_num, reinterpret_cast<unsigned long long *>(rngpos),
                       (unsigned long long)rng_offset, loss_slot, losses_out, (lrs && lr_dev) ? *lrs : none, lr_dev,
                       aug_mask, aug_noise, aug_ldn);
  else
    hipLaunchKernelGGL(iter_stage_kernel<false>, dim3(gx, B), dim3(256), 0, agx::as_stream(stream), x, ldxs, X, X2, ldx, G, vec,
                       cat, scat, n_cat, num, snum, n_num, reinterpret_cast<unsigned long long *>(rngpos),
                       (unsigned long long)rng_offset, loss_slot, losses_out, (lrs && lr_dev) ? *lrs : none, lr_dev,
                       nullptr, nullptr, 0);
  AGX_LAUNCH_CHECK();
  return AGX_OK;
}


// ---- optional batch-norm stage (no reference counterpart: tf_utils.py:85-88 has Dense + ReLU only) ----
namespace {
__global__ void __launch_bounds__(256) bn_fold_kernel(const float *__restrict__ mean, const float *__restrict__ vs,
                                                      int is_std, const float *__restrict__ gamma,
                                                      const float *__restrict__ beta, float eps,
                                                      float *__restrict__ scale, float *__restrict__ shift, int N) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  const float var = is_std ? vs[n] * vs[n] : vs[n];
  const float sc = (gamma ? gamma[n] : 1.f) / sqrtf(var + eps);
  scale[n] = sc;
  shift[n] = (beta ? beta[n] : 0.f) - mean[n] * sc;
}

__global__ void __launch_bounds__(256) affine_act_kernel(float *__restrict__ y, int ldy, int M, int N,
                                                         const float *__restrict__ scale,
                                                         const float *__restrict__ shift, int act, float slope) {
  const int r = blockIdx.y;
  float *row = y + (size_t)r * ldy;
  for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < N; c += gridDim.x * blockDim.x) {
    float v = row[c] * scale[c] + shift[c];
    if (act) v = v > 0.f ? v : v * slope;
    row[c] = v;
  }
}
}  // namespace

namespace {
// one workgroup per 32 columns: 32 column lanes x 8 row lanes; the 8 partial sums of a column meet in LDS in a fixed order
__global__ void __launch_bounds__(256) bn_backward_kernel(float *__restrict__ dz, int lddz, const float *__restrict__ x, int ldx,
                                                          const float *__restrict__ mean, const float *__restrict__ sd,
                                                          const float *__restrict__ gamma, float eps, int M, int N,
                                                          float *__restrict__ dgamma, float *__restrict__ dbeta) {
  __shared__ float red[2][8][32];
  const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  const bool ok = c < N;
  const float mu = ok ? mean[c] : 0.f, s0 = ok ? sd[c] : 1.f;
  const float inv = 1.f / sqrtf(s0 * s0 + eps);
  float s1 = 0.f, s2 = 0.f;
  if (ok)
    for (int r = rl; r < M; r += 8) {
      const float d = dz[(size_t)r * lddz + c];
      const float xh = (x[(size_t)r * ldx + c] - mu) * inv;
      s1 += d;
      s2 += d * xh;
    }
  red[0][rl][cl] = s1;
  red[1][rl][cl] = s2;
  __syncthreads();
  s1 = 0.f;
  s2 = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    s1 += red[0][k][cl];
    s2 += red[1][k][cl];
  }
  if (!ok) return;
  if (rl == 0) {
    dbeta[c] = s1;
    dgamma[c] = s2;
  }
  const float g = (gamma ? gamma[c] : 1.f) * inv, m1 = s1 / (float)M, m2 = s2 / (float)M;
  for (int r = rl; r < M; r += 8) {
    const float xh = (x[(size_t)r * ldx + c] - mu) * inv;
    float *p = dz + (size_t)r * lddz + c;
    *p = g * (*p - m1 - xh * m2);
  }
}

__global__ void __launch_bounds__(256) bn_moving_kernel(float *__restrict__ mm, float *__restrict__ mv,
                                                        const float *__restrict__ mean, const float *__restrict__ sd,
                                                        float momentum, int N) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  mm[n] = momentum * mm[n] + (1.f - momentum) * mean[n];
  mv[n] = momentum * mv[n] + (1.f - momentum) * (sd[n] * sd[n]);
}
}  // namespace

extern "C" int agx_bn_backward(void *stream, float *dz, int32_t lddz, const float *x, int32_t ldx, const float *mean,
                               const float *std, const float *gamma, float eps, int32_t M, int32_t N, float *dgamma,
                               float *dbeta) {
  AGX_CHECK_ARG(dz && x && mean && std && dgamma && dbeta && M > 0 && N > 0 && lddz >= N && ldx >= N && eps >= 0.f,
                "agx_bn_backward: bad argument");
  hipLaunchKernelGGL(bn_backward_kernel, dim3(agx::cdiv(N, 32)), dim3(256), 0, agx::as_stream(stream), dz, lddz, x, ldx, mean,
                     std, gamma, eps, M, N, dgamma, dbeta);
  AGX_LAUNCH_CHECK();
  return AGX_OK;
}

extern "C" int agx_bn_moving_update(void *stream, float *moving_mean, float *moving_var, const float *mean, const float *std,
                                    float momentum, int32_t N) {
  AGX_CHECK_ARG(moving_mean && moving_var && mean && std && N > 0 && momentum >= 0.f && momentum <= 1.f,
                "agx_bn_moving_update: bad argument");
  hipLaunchKernelGGL(bn_moving_kernel, dim3(agx::cdiv(N, 256)), dim3(256), 0, agx::as_stream(stream), moving_mean, moving_var,
                     mean, std, momentum, N);
  AGX_LAUNCH_CHECK();
  return AGX_OK;
}

extern "C" int agx_bn_fold(void *stream, const float *mean, const float *var_or_std, int32_t is_std,
                           const float *gamma, const float *beta, float eps, float *scale, float *shift, int32_t N) {
  AGX_CHECK_ARG(mean && var_or_std && scale && shift && N > 0 && eps >= 0.f, "agx_bn_fold: bad argument");
  hipLaunchKernelGGL(bn_fold_kernel, dim3(agx::cdiv(N, 256)), dim3(256), 0, agx::as_stream(stream), mean, var_or_std,
                     is_std, gamma, beta, eps, scale, shift, N);
  AGX_LAUNCH_CHECK();
  return AGX_OK;
}

extern "C" int agx_affine_act(void *stream, float *y, int32_t ldy, int32_t M, int32_t N, const float *scale,
                              const float *shift, int32_t act, float slope) {
  AGX_CHECK_ARG(y && scale && shift && M >= 0 && N > 0 && ldy >= N, "agx_affine_act: bad argument");
  hipStream_t st = agx::as_stream(stream);
  for (int r0 = 0; r0 < M; r0 += 65535) {
    const int rows = std::min(65535, M - r0);
    hipLaunchKernelGGL(affine_act_kernel, dim3(std::min(agx::cdiv(N, 256), 64), rows), dim3(256), 0, st,
                       y + (size_t)r0 * ldy, ldy, rows, N, scale, shift, act, slope);
    AGX_LAUNCH_CHECK();
  }
  return AGX_OK;
}
