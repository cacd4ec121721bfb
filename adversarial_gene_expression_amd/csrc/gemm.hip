// fp32-MFMA GEMM family for gfx950 (MI355X).
//
// One LDS-tiled kernel template serves every dense-layer product of the cGAN
// (reference tf_utils.py:85-88,132-135 forward; MatMul/ReLU grads of :220,:255)
// and the symmetric X^T X contraction of the Pearson matrix (utils.py:824):
//
//   NN  Y  = X . W          (ta=0, tb=0)   forward
//   NT  dX = d . W^T        (ta=0, tb=1)   input gradients / GP gradient
//   TN  dW = X^T . d        (ta=1, tb=0)   weight gradients, X^T X
//
// Arithmetic: v_mfma_f32_32x32x2_f32 (exact fp32 fma chain, fp32 accumulate).
// Operand tiles are staged global -> registers -> LDS (register prefetch of
// tile t+1 overlaps the MFMAs of tile t, two LDS buffers, one barrier per
// K-tile).  The LDS image of an operand depends on which way its source is
// contiguous so that both the staging stores and the per-lane fragment reads
// (one ds_read_b32 per lane per MFMA operand) are bank-conflict free:
//   k-contiguous source  -> image [rows][BK+1]   (odd stride)
//   row-contiguous source-> image [BK][rows]
#include "agx_common.h"

#include <algorithm>
#include <type_traits>
#include <cstdlib>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BK = 32;  // K granularity of split-K chunks; per-config K-tile depth is a template parameter

struct KParams {
  const float *A, *B;
  float *C;
  int M, N, K, lda, ldb, ldc;
  float alpha;
  const float *bias;
  const float *addrow;
  int addmod, ldadd;
  int act;
  float slope;
  const float *mask;
  int ldmask;
  int accumulate;
  const float *rowscale;  // v *= rowscale[row] (before bias)
  const float *col_scale, *col_shift;  // optional per-feature affine stage after the bias: v = v*scale[col] + shift[col]
                                       // (batch-norm with folded statistics; NULL on the reference path: no BN, tf_utils.py:85-88)
  float *rowsq;           // [M, rowsq_ld]: per 32-column slab sum of squares of the stored values
  int rowsq_ld;
  float *ws;
  float *c2;              // dual store: rows [0, c2_rows) of a plain product also go here (ld ldc2)
  int ldc2, c2_rows;
  int splitk, kchunk;
  int to_ws;              // 1: the main kernel stores raw partial slabs (also when splitk == 1)
  int ws_by_row;          // slab layout, see slab_off
  int interp_rows;        // R > 0: M == 2R rows are [real; fake]; the reduce emits rows [2R,3R) too
  const float *interp_alpha;
  // fused optimizer (captured single-GPU iteration, OPT instantiations only): rows [0, opt_rows) of
  // the result are a weight gradient that is consumed on the spot - RMSprop update of opt_w / opt_s1
  // at the same offsets as C, same expression order as rmsprop_kernel, so the weights are
  // bit-identical to gradient store + optimizer launch; the gradient of those rows is not stored
  float *opt_w, *opt_s1;
  int opt_rows;
  float opt_lr, opt_rho, opt_eps;
  // ... and the REST of the network's flat parameter buffer (everything outside [hole_lo, hole_hi),
  // whose gradients earlier launches left in rest_g) is updated by workgroups appended behind the
  // product's tiles, so a captured step needs no optimizer launch at all
  int opt_tiles;          // workgroups [0, opt_tiles) are GEMM tiles, the others run the flat update
  int opt_rest_blocks;
  float *rest_w, *rest_s1;
  const float *rest_g;
  long long rest_n, rest_hole_lo4, rest_hole_len4;
  // ... and the next step's stacked critic input (FusedOpt::nx_*): copy workgroups behind the update
  // workgroups, covariate columns by update workgroup 0
  int opt_copy_blocks;
  const float *nx_src;
  const float *nx_mask, *nx_noise;   // augmentation of the copied rows: + nx_mask[r] * nx_noise[r * nx_ldn + g]
  int nx_ldn;
  float *nx_dst, *nx_cov;
  int nx_lds, nx_ldd, nx_rows, nx_G, nx_cov_rows, nx_B, nx_nnum, nx_ldcat;
  const float *nx_num;
  const int *nx_cat;
  agx_embed_t nx_emb;
  // SYRK only
  long long tile_begin;
  int tiles_1d;
  // stream-K (SK instantiations): sk_part = one BM x BN partial tile per workgroup, sk_flags[w] = "workgroup w's
  // partial is published" (consumed and reset by the tile's finisher), sk_flags[sk_wgs] = spin-timeout word
  float *sk_part;
  unsigned *sk_flags;
  int sk_wgs, sk_nxcd;
};

// float offset of row `row` of split-K slab z: [z][M][N] for the library's own reduce kernels,
// [M][z][N] (all partials of a row within splitk*N contiguous floats) when a caller's kernel
// finishes the product
__device__ __forceinline__ size_t slab_off(const KParams &p, int z, int row) {
  return p.ws_by_row ? ((size_t)row * p.splitk + z) * p.N : ((size_t)z * p.M + row) * p.N;
}

// Sum over each half wave (lanes 0..31 / 32..63), valid in lanes 31 and 63 only: four row_shr DPP
// steps inside the 16-lane rows, then row_bcast:15 carries a row's total into the next row.  Five
// VALU steps per value; the __shfl_xor butterfly costs five LDS-crossbar permutes (ds_bpermute).
__device__ __forceinline__ float half_wave_sum_at31(float v) {
#define AGX_DPP_ADD(CTRL, ROWS) \
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROWS, 0xf, true))
  AGX_DPP_ADD(0x111, 0xf);  // row_shr:1
  AGX_DPP_ADD(0x112, 0xf);  // row_shr:2
  AGX_DPP_ADD(0x114, 0xf);  // row_shr:4
  AGX_DPP_ADD(0x118, 0xf);  // row_shr:8
  AGX_DPP_ADD(0x142, 0xa);  // row_bcast:15 into rows 1 and 3
#undef AGX_DPP_ADD
  return v;
}

// flat RMSprop over [0, rest_n) minus the hole, by workgroup `blk` of `nblk` (same expressions as
// rmsprop_kernel in elementwise.hip: bit-identical updates)
__device__ __forceinline__ void rest_rmsprop(const KParams &p, int blk, int nblk) {
  const long long n4 = p.rest_n >> 2;
  const float rho = p.opt_rho, omr = 1.f - p.opt_rho, lr = p.opt_lr, eps = p.opt_eps;
  float4 *w4 = reinterpret_cast<float4 *>(p.rest_w);
  const float4 *g4 = reinterpret_cast<const float4 *>(p.rest_g);
  float4 *r4 = reinterpret_cast<float4 *>(p.rest_s1);
  for (long long j = (long long)blk * blockDim.x + threadIdx.x; j < n4 - p.rest_hole_len4;
       j += (long long)nblk * blockDim.x) {
    const long long i = j < p.rest_hole_lo4 ? j : j + p.rest_hole_len4;
    float4 wv = w4[i], gv = g4[i], rv = r4[i];
#define AGX_RMS1(c)                                  \
  {                                                  \
    const float gg = gv.c * 1.f;                     \
    rv.c = rho * rv.c + omr * (gg * gg);             \
    wv.c = wv.c - lr * gg / (sqrtf(rv.c) + eps);     \
  }
    AGX_RMS1(x) AGX_RMS1(y) AGX_RMS1(z) AGX_RMS1(w)
#undef AGX_RMS1
    w4[i] = wv;
    r4[i] = rv;
  }
  const long long i = (n4 << 2) + (long long)blk * blockDim.x + threadIdx.x;
  if (i < p.rest_n) {
    const float gg = p.rest_g[i] * 1.f;
    const float r = rho * p.rest_s1[i] + omr * (gg * gg);
    p.rest_s1[i] = r;
    p.rest_w[i] = p.rest_w[i] - lr * gg / (sqrtf(r) + eps);
  }
}

// generated rows of the next step -> its stacked input (gene columns), by copy workgroup blk of nblk
__device__ __forceinline__ void next_copy(const KParams &p, int blk, int nblk) {
  const int g4 = p.nx_G >> 2, tail = p.nx_G & 3, per_row = g4 + (tail ? 1 : 0);
  const long long total = (long long)p.nx_rows * per_row;
  for (long long q = (long long)blk * blockDim.x + threadIdx.x; q < total; q += (long long)nblk * blockDim.x) {
    const int r = (int)(q / per_row), c = (int)(q % per_row);
    const float *s = p.nx_src + (size_t)r * p.nx_lds + 4 * c;
    float *d = p.nx_dst + (size_t)r * p.nx_ldd + 4 * c;
    if (p.nx_mask) {   // same expression as stack_rows_kernel (the noise rows are not 16-B aligned: scalar loads)
      const float m = p.nx_mask[r];
      const float *nz = p.nx_noise + (size_t)r * p.nx_ldn + 4 * c;
      if (c < g4) {
        float4 v = *reinterpret_cast<const float4 *>(s);
        v.x = v.x + m * nz[0]; v.y = v.y + m * nz[1]; v.z = v.z + m * nz[2]; v.w = v.w + m * nz[3];
        *reinterpret_cast<float4 *>(d) = v;
      } else {
        for (int e = 0; e < tail; ++e) d[e] = s[e] + m * nz[e];
      }
    } else if (c < g4) {
      *reinterpret_cast<float4 *>(d) = *reinterpret_cast<const float4 *>(s);
    } else {
      for (int e = 0; e < tail; ++e) d[e] = s[e];
    }
  }
}

// covariate columns of the next step's rows from the embedding tables this workgroup has just updated
// (same values as stack_rows_kernel writes: [num | emb_0 | emb_1 ...] of source row r % B)
// (eight elements per thread in branch-free sweeps - categories, then L1-bypassing table loads, all independent -
//  instead of one element at a time measured SLOWER: E. coli 0.3316 against 0.3208 ms, GTEx 0.4859 against 0.4840 ms)
__device__ __forceinline__ void next_cov(const KParams &p) {
  int ncov = p.nx_nnum;
  for (int i = 0; i < p.nx_emb.n_cat; ++i) ncov += p.nx_emb.edim[i];
  const int total = p.nx_cov_rows * ncov;
  const volatile float *tables = p.rest_w;
  for (int idx = threadIdx.x; idx < total; idx += blockDim.x) {
    const int r = idx / ncov, b = r % p.nx_B;
    int c = idx % ncov;
    float v = 0.f;
    if (c < p.nx_nnum) {
      v = p.nx_num[(size_t)b * p.nx_nnum + c];
    } else {
      c -= p.nx_nnum;
      for (int i = 0; i < p.nx_emb.n_cat; ++i) {
        if (c < p.nx_emb.edim[i]) {
          const int row = p.nx_cat[(size_t)b * p.nx_ldcat + i];
          if (row >= 0 && row < p.nx_emb.vocab[i]) v = tables[p.nx_emb.table_off[i] + (size_t)row * p.nx_emb.edim[i] + c];
          break;
        }
        c -= p.nx_emb.edim[i];
      }
    }
    p.nx_cov[(size_t)r * p.nx_ldd + p.nx_G + idx % ncov] = v;
  }
}

__device__ __forceinline__ void opt_rmsprop_apply(const KParams &p, size_t off, float g, float w, float r) {
  const float omr = 1.f - p.opt_rho;
  r = p.opt_rho * r + omr * (g * g);
  p.opt_s1[off] = r;
  p.opt_w[off] = w - p.opt_lr * g / (sqrtf(r) + p.opt_eps);
}

__device__ __forceinline__ float epilogue_store(const KParams &p, int row, int col, float v) {
  v *= p.alpha;
  if (p.rowscale) v *= p.rowscale[row];
  if (p.bias) v += p.bias[col];
  if (p.addrow) v += p.addrow[(size_t)(row % p.addmod) * p.ldadd + col];
  if (p.col_scale) v = v * p.col_scale[col] + p.col_shift[col];
  if (p.act) v = v > 0.f ? v : v * p.slope;
  if (p.mask) {
    float mk = p.mask[(size_t)row * p.ldmask + col];
    v = mk > 0.f ? v : v * p.slope;
  }
  float *c = p.C + (size_t)row * p.ldc + col;
  if (p.accumulate) v += *c;
  *c = v;
  return v;
}

// Same epilogue for NV values of ONE column: all side loads (bias, addrow, mask, old C) are issued
// up front with clamped (always valid) indices, so they overlap instead of one round trip each.
// Returns the stored values in v (0 for elements outside the matrix).
// pre: the column's {bias, col_scale, col_shift} already in registers (a caller that stores several sub-tiles
// must request them BEFORE its first store: vmcnt retires in issue order, so a load issued behind the stores of
// the previous sub-tile waits for their write acknowledgements - measured 25 us of epilogue per 128x128 tile).
template <int NV, bool OPT = false>
__device__ __forceinline__ void epilogue_column(const KParams &p, const int (&rows)[NV], int col,
                                                float (&v)[NV], const float *pre = nullptr) {
  const int cc = min(col, p.N - 1);
  const bool colok = col < p.N;
  float bias = 0.f, csc = 1.f, csh = 0.f, add[NV], mk[NV], old[NV], rs[NV];
  int cr[NV];
#pragma unroll
  for (int e = 0; e < NV; ++e) cr[e] = min(rows[e], p.M - 1);
  if (pre) {
    bias = pre[0];
    csc = pre[1];
    csh = pre[2];
  } else {
    if (p.bias) bias = p.bias[cc];
    if (p.col_scale) {
      csc = p.col_scale[cc];
      csh = p.col_shift[cc];
    }
  }
  if (p.rowscale) {
#pragma unroll
    for (int e = 0; e < NV; ++e) rs[e] = p.rowscale[cr[e]];
  }
  if (p.addrow) {
#pragma unroll
    for (int e = 0; e < NV; ++e) add[e] = p.addrow[(size_t)(cr[e] % p.addmod) * p.ldadd + cc];
  }
  if (p.mask) {
#pragma unroll
    for (int e = 0; e < NV; ++e) mk[e] = p.mask[(size_t)cr[e] * p.ldmask + cc];
  }
  if (p.accumulate) {
#pragma unroll
    for (int e = 0; e < NV; ++e) old[e] = p.C[(size_t)cr[e] * p.ldc + cc];
  }
  if (OPT) {
    // plain weight gradient (alpha only): update in groups of 4 values to bound the live registers
#pragma unroll
    for (int e0 = 0; e0 < NV; e0 += 4) {
      float ow[4], os[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        ow[e] = p.opt_w[(size_t)cr[e0 + e] * p.ldc + cc];
        os[e] = p.opt_s1[(size_t)cr[e0 + e] * p.ldc + cc];
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float x = v[e0 + e] * p.alpha;
        const bool ok = colok && rows[e0 + e] < p.M;
        if (ok) {
          const size_t off = (size_t)rows[e0 + e] * p.ldc + col;
          if (rows[e0 + e] < p.opt_rows) opt_rmsprop_apply(p, off, x, ow[e], os[e]);
          else p.C[off] = x;
        }
        v[e0 + e] = ok ? x : 0.f;
      }
    }
    return;
  }
#pragma unroll
  for (int e = 0; e < NV; ++e) {
    float x = v[e] * p.alpha;
    if (p.rowscale) x *= rs[e];
    if (p.bias) x += bias;
    if (p.addrow) x += add[e];
    if (p.col_scale) x = x * csc + csh;
    if (p.act) x = x > 0.f ? x : x * p.slope;
    if (p.mask) x = mk[e] > 0.f ? x : x * p.slope;
    if (p.accumulate) x += old[e];
    const bool ok = colok && rows[e] < p.M;
    if (ok) p.C[(size_t)rows[e] * p.ldc + col] = x;
    v[e] = ok ? x : 0.f;
  }
}

// ---- operand tile: ROWS x BK --------------------------------------------
// LDS image, the same for both source orientations: element (row r, k) lives at dword
//     ((k/4) * ROWS + swz(r, k/4)) * 4 + k%4 ,   swz(r, q) = r ^ (((r >> 3) & 3) ^ (q & 7))
// i.e. [k-quad][row][4 consecutive k].  An MFMA lane (row l31, half lh) fetches the four k of
// k-quad 2g+lh with ONE ds_read_b128 and feeds four 32x32x2 MFMAs from it (A and B use the same
// k order 8g + 4lh + e, so the product is unchanged): one LDS instruction per four MFMAs per
// fragment instead of one per MFMA - with one wave per SIMD every non-MFMA instruction costs
// issue time the matrix pipe cannot hide.  The XOR swizzle (low 3 row bits only, a bijection
// inside every 8-row group) makes all three access patterns hit distinct bank quads:
//   - fragment read: 16 consecutive rows of one k-quad;
//   - k-contiguous source: one ds_write_b128 per float4, 8 lanes = 8 k-quads of one row;
//   - row-contiguous source: a thread owns a 4(row) x 4(k) block, transposes it in registers and
//     writes 4 ds_write_b128 (lane j -> rows 4j+e).
template <int ROWS, bool KC, int NT, bool VEC, int BK>
struct Tile {
  static constexpr int F4 = ROWS * BK / 4;
  static constexpr int PT = F4 / NT;  // float4 per thread
  static_assert(F4 % NT == 0, "tile must divide over the block");
  static constexpr bool BLOCK4 = !KC && (PT % 4 == 0);  // row-contiguous: 4x4 register transpose
  static constexpr int LDS_FLOATS = BK * ROWS;

  __device__ static __forceinline__ int lds_quad(int r, int kq) {  // float4 index of (row, k-quad)
    return kq * ROWS + (r ^ (((r >> 3) & 3) ^ (kq & 7)));
  }

  // (r, k) of the first element of float4 number i of thread tid
  __device__ static __forceinline__ void coords(int tid, int i, int &r, int &k) {
    if (KC) {
      const int q = tid + i * NT;
      r = q / (BK / 4);
      k = (q % (BK / 4)) * 4;
    } else if (BLOCK4) {
      const int blk = tid + (i >> 2) * NT;  // 4x4 block: rows 4*rq.., k-quad kq; float4 i&3 = one k of it
      r = (blk % (ROWS / 4)) * 4;
      k = (blk / (ROWS / 4)) * 4 + (i & 3);
    } else {
      const int q = tid + i * NT;
      k = q / (ROWS / 4);
      r = (q % (ROWS / 4)) * 4;
    }
  }

  // Branch-free global loads: every address is clamped into the operand, so that hipcc issues
  // all loads of a tile back to back and waits only where the values are consumed (a load under
  // a per-element condition becomes a branch with its own s_waitcnt vmcnt(0) and serialises the
  // whole pipeline).  Out-of-range elements are zeroed later, in store().
  __device__ static __forceinline__ void load(const float *__restrict__ src, int ld, int r0, int R,
                                              int k0, int kend, int tid, float4 (&reg)[PT]) {
#pragma unroll
    for (int i = 0; i < PT; ++i) {
      int r, k;
      coords(tid, i, r, k);
      const int gr = r0 + r, gk = k0 + k;
      float4 v;
      if (KC) {
        const int cr = min(gr, R - 1);
        if (VEC) {
          const int ck = min(gk, (kend - 1) & ~3);
          v = *reinterpret_cast<const float4 *>(src + (size_t)cr * ld + ck);
        } else {
          const float *s = src + (size_t)cr * ld;
          const int kl = kend - 1;
          v.x = s[min(gk, kl)];
          v.y = s[min(gk + 1, kl)];
          v.z = s[min(gk + 2, kl)];
          v.w = s[min(gk + 3, kl)];
        }
      } else {
        const int ck = min(gk, kend - 1);
        if (VEC) {
          const int cr = min(gr, (R - 1) & ~3);
          v = *reinterpret_cast<const float4 *>(src + (size_t)ck * ld + cr);
        } else {
          const float *s = src + (size_t)ck * ld;
          const int rl = R - 1;
          v.x = s[min(gr, rl)];
          v.y = s[min(gr + 1, rl)];
          v.z = s[min(gr + 2, rl)];
          v.w = s[min(gr + 3, rl)];
        }
      }
      reg[i] = v;
    }
  }

  // registers -> LDS image (no masking)
  __device__ static __forceinline__ void store_fast(float *lds, int tid, const float4 (&reg)[PT]) {
    float4 *l4 = reinterpret_cast<float4 *>(lds);
    if (KC) {
#pragma unroll
      for (int i = 0; i < PT; ++i) {
        int r, k;
        coords(tid, i, r, k);
        l4[lds_quad(r, k >> 2)] = reg[i];
      }
    } else if (BLOCK4) {
#pragma unroll
      for (int b = 0; b < PT / 4; ++b) {
        int r, k;
        coords(tid, 4 * b, r, k);
        const float4 c0 = reg[4 * b], c1 = reg[4 * b + 1], c2 = reg[4 * b + 2], c3 = reg[4 * b + 3];
        // (16 ds_write_b32 straight from the load registers instead of this register transpose - an LDS write is free
        //  beside f32 MFMAs, a v_mov is not - measured no faster: 113 against 115 TF at 4096^3 TN, correlation 117 against 119)
        l4[lds_quad(r, k >> 2)] = make_float4(c0.x, c1.x, c2.x, c3.x);
        l4[lds_quad(r + 1, k >> 2)] = make_float4(c0.y, c1.y, c2.y, c3.y);
        l4[lds_quad(r + 2, k >> 2)] = make_float4(c0.z, c1.z, c2.z, c3.z);
        l4[lds_quad(r + 3, k >> 2)] = make_float4(c0.w, c1.w, c2.w, c3.w);
      }
    } else {
#pragma unroll
      for (int i = 0; i < PT; ++i) {
        int r, k;
        coords(tid, i, r, k);
        const int kq = k >> 2, e = k & 3;
        lds[lds_quad(r, kq) * 4 + e] = reg[i].x;
        lds[lds_quad(r + 1, kq) * 4 + e] = reg[i].y;
        lds[lds_quad(r + 2, kq) * 4 + e] = reg[i].z;
        lds[lds_quad(r + 3, kq) * 4 + e] = reg[i].w;
      }
    }
  }

  // masked form for edge tiles / the K tail: zero what lies outside the operand, then store
  __device__ static __forceinline__ void store(float *lds, int tid, const float4 (&reg)[PT], int r0,
                                               int R, int k0, int kend) {
    float4 m[PT];
#pragma unroll
    for (int i = 0; i < PT; ++i) {
      int r, k;
      coords(tid, i, r, k);
      const int gr = r0 + r, gk = k0 + k;
      float4 v = reg[i];
      if (KC) {
        const bool rv = gr < R;
        v.x = (rv && gk < kend) ? v.x : 0.f;
        v.y = (rv && gk + 1 < kend) ? v.y : 0.f;
        v.z = (rv && gk + 2 < kend) ? v.z : 0.f;
        v.w = (rv && gk + 3 < kend) ? v.w : 0.f;
      } else {
        const bool kv = gk < kend;
        v.x = (kv && gr < R) ? v.x : 0.f;
        v.y = (kv && gr + 1 < R) ? v.y : 0.f;
        v.z = (kv && gr + 2 < R) ? v.z : 0.f;
        v.w = (kv && gr + 3 < R) ? v.w : 0.f;
      }
      m[i] = v;
    }
    store_fast(lds, tid, m);
  }

  // ---- interior fast path (full tile inside the operand, 16-B aligned): the lane's element offsets
  // are computed once, the per-K-tile address is a wave-uniform base + that offset (no clamps, no
  // 64-bit vector multiply per load), and the LDS store needs no masking
  __device__ static __forceinline__ void lane_offsets(int ld, int tid, int (&off)[PT]) {
#pragma unroll
    for (int i = 0; i < PT; ++i) {
      int r, k;
      coords(tid, i, r, k);
      off[i] = KC ? r * ld + k : k * ld + r;
    }
  }
  __device__ static __forceinline__ const float *ubase(const float *src, int ld, int r0, int k0) {
    return KC ? src + (size_t)r0 * ld + k0 : src + (size_t)k0 * ld + r0;
  }
  __device__ static __forceinline__ void load_fast(const float *__restrict__ ub, const int (&off)[PT],
                                                   float4 (&reg)[PT]) {
#pragma unroll
    for (int i = 0; i < PT; ++i) reg[i] = *reinterpret_cast<const float4 *>(ub + off[i]);
  }

  // the four k (8g + 4*lh + 0..3) of row r for k-group g: one ds_read_b128
  __device__ static __forceinline__ float4 frag4(const float *lds, int r, int g, int lh) {
    return reinterpret_cast<const float4 *>(lds)[lds_quad(r, 2 * g + lh)];
  }
};

using agx::tri_decode;

#ifdef AGX_STAMPS
// diagnostic build only (tools/stamps.py): per-workgroup phase stamps of gemm_kernel, 8 x int64 per workgroup
// [start, first tile staged, main loop done, epilogue issued, stores drained, HW_ID, XCC_ID, tile]
__device__ long long *g_stamps = nullptr;
__device__ int g_stamp_m = -1, g_stamp_k = -1;   // only launches of this M / K are recorded (-1: any)
#define AGX_STAMP_ON (g_stamps && threadIdx.x == 0 && (g_stamp_m < 0 || p.M == g_stamp_m) && (g_stamp_k < 0 || p.K == g_stamp_k))
#define AGX_STAMP(i) \
  if (AGX_STAMP_ON) stamp_v[i] = (long long)__builtin_amdgcn_s_memrealtime()
#else
#define AGX_STAMP(i)
#endif

#ifndef AGX_DUAL_STORE
#define AGX_DUAL_STORE 1   // 0: build without the dual-store block of the plain epilogue (A/B of its cost in every instantiation)
#endif
#ifndef AGX_WPE_SMALL
#define AGX_WPE_SMALL 1
#endif
#ifndef AGX_WPE_W8
#define AGX_WPE_W8 1
#endif
#ifndef AGX_OPTI_WPE
#define AGX_OPTI_WPE 1
#endif
// resident waves per SIMD the register allocator must leave room for (512 registers per SIMD lane, VGPR+AGPR):
// two 4-wave 128x128 workgroups per CU (<= 256 registers) - one multiplies while the other loads or stores
constexpr int gemm_min_waves(int bm, int bn, int wm, int wn, bool opt, bool opti) {
  return (opt && opti) ? AGX_OPTI_WPE
                       : ((bm * bn <= 64 * 64 || bm * bn == 32 * 128) ? AGX_WPE_SMALL : (wm * wn == 8 ? AGX_WPE_W8 : 2));
}

// OPTI: the fused-optimizer epilogue and its prefetch take a clamp-free path on interior tiles (its own
// instantiation: compiled into the K = 1536 launches the same source measured 1-2 % SLOWER per iteration)
//
// SK (stream-K, TN weight-gradient products whose 128x128 tile count quantises badly over the CUs - the 512-row
// dW_1: 284 tiles x 48 K-tiles on 256 CUs): the launch has exactly sk_wgs = 2 x CUs workgroups (all co-resident);
// the (tile, K-tile) iteration space is dealt EVENLY and contiguously over them - independently per XCD: XCD x owns
// tiles [x T/8, (x+1) T/8) and its workgroups blockIdx % 8 == x, so every hand-off stays inside one XCD's L2 and a
// tile's A panel is fetched by one XCD only.  A workgroup whose share ends inside a tile publishes its accumulators as
// a partial tile (plain stores -> every wave drains -> barrier -> one lane: agent-scope release, relaxed flag store);
// the workgroup that reaches the tile's END is its finisher: it polls the flags of the (lower-numbered) workgroups
// that hold the tile's earlier K ranges (bounded spin, relaxed), acquires once, adds their partials in ascending K
// order and then its own accumulators, and runs the epilogue (plain store or the fused RMSprop update).  The
// partition depends only on (shape, sk_wgs): results are bit-reproducible run to run.  A waiter only ever waits for
// LOWER-numbered workgroups of its own XCD, which the dispatcher has started before it: no deadlock even if the
// grid were not fully resident; the spin is bounded anyway (timeout word sk_flags[sk_wgs], checked by the tests).
template <int BM, int BN, int WM, int WN, bool TA, bool TB, bool VEC, bool SYRK, int BK = 32, int DBG = 0,
          bool OPT = false, bool OPTI = false, bool SK = false>
__global__ void __launch_bounds__(WM *WN * 64) __attribute__((amdgpu_waves_per_eu(gemm_min_waves(BM, BN, WM, WN, OPT, OPTI))))
gemm_kernel(const KParams p) {
  constexpr int NT = WM * WN * 64;
  constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
  static_assert(TM >= 1 && TN >= 1, "wave tile");
  using TileA = Tile<BM, !TA, NT, VEC, BK>;
  using TileB = Tile<BN, TB, NT, VEC, BK>;
  constexpr int STAGE = TileA::LDS_FLOATS + TileB::LDS_FLOATS;

  extern __shared__ __attribute__((aligned(16))) float smem[];
#ifdef AGX_STAMPS
  long long stamp_v[5] = {0, 0, 0, 0, 0};
#endif
  AGX_STAMP(0);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int l31 = lane & 31, lh = lane >> 5;

  if (OPT && p.rest_n > 0 && (int)blockIdx.x >= p.opt_tiles) {
    const int rb = (int)blockIdx.x - p.opt_tiles;
    if (rb < p.opt_rest_blocks) {
      rest_rmsprop(p, rb, p.opt_rest_blocks);
      if (rb == 0 && p.nx_cov) {   // the tables (flat offsets < 1024) were updated by this workgroup's first pass
        __threadfence_block();
        __syncthreads();
        next_cov(p);
      }
    } else {
      next_copy(p, rb - p.opt_rest_blocks, p.opt_copy_blocks);
    }
    return;
  }
  // tiles of the product (an OPT launch without split-K may carry extra update workgroups behind them)
  const int gx = (OPT && p.rest_n > 0) ? p.opt_tiles : (int)gridDim.x;

  // stream-K: this workgroup's contiguous share [sk_it, sk_end) of its XCD's (tile, K-tile) iteration space,
  // walked from its END: the first thing a workgroup computes is the HEAD of its last tile (published at once as a
  // partial), the last thing is the TAIL of its first tile, whose earlier K ranges lower-numbered workgroups
  // published when they started - nobody waits for work that has not been done yet
  int sk_it = 0, sk_end = 1, sk_tile_lo = 0, sk_kt = 1, sk_local = 0, sk_wx = 1, sk_ix = 0;
  if (SK) {
    const int nx = p.sk_nxcd;
    const int T = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
    const int xcd = (int)blockIdx.x % nx;
    sk_local = (int)blockIdx.x / nx;
    sk_wx = gx / nx;
    sk_tile_lo = xcd * T / nx;
    sk_kt = (p.K + BK - 1) / BK;
    sk_ix = ((xcd + 1) * T / nx - sk_tile_lo) * sk_kt;   // (< 2^31 / sk_wx: checked on the host)
    sk_it = sk_local * sk_ix / sk_wx;
    sk_end = (sk_local + 1) * sk_ix / sk_wx;
  }
  do {   // (not stream-K: exactly one pass)
  int tm, tn, z;
  int kbeg, kend;
  bool sk_first = true, sk_last = true;
  int sk_tile_it0 = 0;
  if (SK) {
    const int tiles_n = (p.N + BN - 1) / BN;
    const int tl = (sk_end - 1) / sk_kt;              // the LAST tile of what is left of the share
    sk_tile_it0 = tl * sk_kt;
    const int kt0 = max(sk_it, sk_tile_it0) - sk_tile_it0, kt1 = sk_end - sk_tile_it0;
    sk_first = kt0 == 0;
    sk_last = kt1 == sk_kt;
    sk_end = sk_tile_it0 + kt0;
    tm = (sk_tile_lo + tl) / tiles_n;
    tn = (sk_tile_lo + tl) % tiles_n;
    z = 0;
    kbeg = kt0 * BK;
    kend = min(p.K, kt1 * BK);
  } else {
  if (SYRK) {
    tri_decode(p.tile_begin + agx::xcd_logical_id(blockIdx.x, gridDim.x), p.tiles_1d, tm, tn);
    z = 0;
  } else {
    // logical order: split-K slice major, then the tiles row-major (a column-major walk where its per-XCD range
    // touches fewer operand bytes measured no different: 64.5 against 65.3 us on the 512-row input gradient)
    const int lid = agx::xcd_logical_id(blockIdx.x + gx * blockIdx.y, gx * gridDim.y);
    const int tiles_n = (p.N + BN - 1) / BN;
    const int t = lid % gx;
    z = lid / gx;
    tm = t / tiles_n;
    tn = t % tiles_n;
  }
  kbeg = z * p.kchunk;
  kend = min(p.K, kbeg + p.kchunk);
  }
  const int m0 = tm * BM, n0 = tn * BN;
  const int nt = (kend - kbeg + BK - 1) / BK;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  // SYRK (the Pearson matrix, K = number of samples, up to 100000): two-level accumulation.  Every SYRK_FLUSH K-tiles
  // (2048 samples) the MFMA accumulators are added into a second register set and cleared, so no fp32 chain is longer
  // than 2048 products + K/2048 partial sums: at K = 100000 the error against float64 drops from 2.8e-5 (one chain; the
  // reference's np.dot blocks its K loop too, utils.py:824) to the 1e-6 range.  K <= 2048 never flushes: same bits as
  // the single chain (goldens unchanged).  One v_add per accumulator register per 64 K-tiles: no measurable cost.
  constexpr int SYRK_FLUSH = 64;
  f32x16 acc2[SYRK ? TM : 1][SYRK ? TN : 1];
  if (SYRK) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[SYRK ? i : 0][SYRK ? j : 0][r] = 0.f;
  }
  auto syrk_flush = [&](int tiles_done) {
    if (SYRK && (tiles_done & (SYRK_FLUSH - 1)) == 0) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          acc2[SYRK ? i : 0][SYRK ? j : 0] += acc[i][j];
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        }
    }
  };

  // fused optimizer with one 32x32 tile per wave: the weights and optimizer slots this lane will
  // update are requested NOW, so their round trip hides behind the whole K loop instead of
  // stalling the epilogue (clamped addresses; out-of-range elements are never stored)
  constexpr bool OPT_PRE = OPT && TM == 1 && TN == 1;
  // (starting each CU's k-th workgroup k * 1.7 us late, to spread the load / multiply / update phases
  // of this one-round grid over time, only added the delay: +4 us per launch per step of delay)
  float pw[OPT_PRE ? 16 : 1], ps[OPT_PRE ? 16 : 1];
  const bool opt_interior = OPT_PRE && OPTI && m0 + BM <= min(p.M, p.opt_rows) && n0 + BN <= p.N;
  auto opt_prefetch = [&]() {
  if (opt_interior) {
    // interior tile: one 64-bit base, 32-bit row offsets, no clamps
    const size_t base = (size_t)(m0 + wm * 32 + 4 * lh) * p.ldc + (n0 + wn * 32 + l31);
    const float *__restrict__ wb = p.opt_w + base;
    const float *__restrict__ sb = p.opt_s1 + base;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int o = ((r & 3) + 8 * (r >> 2)) * p.ldc;
      pw[r] = wb[o];
      ps[r] = sb[o];
    }
  } else if (OPT_PRE) {
    const int cc = min(n0 + wn * 32 + l31, p.N - 1);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int rr = min(m0 + wm * 32 + 4 * lh + (r & 3) + 8 * (r >> 2), p.M - 1);
      const size_t off = (size_t)rr * p.ldc + cc;
      pw[r] = p.opt_w[off];
      ps[r] = p.opt_s1[off];
    }
  }
  };
  // (requesting the first two K-tiles BEFORE these operands - vmcnt retires in issue order - measured slower:
  // 0.532 against 0.522 ms per 32-row iteration; so did moving them as dwordx4 through an LDS transposition)
  // (non-temporal loads and stores of the optimizer slots: 0.4947 against 0.4921 ms; of the weights too: 0.5034 -
  // both matrices are re-read from cache by the next step's launches)
  opt_prefetch();

  float4 ra[TileA::PT], rb[TileB::PT];
  int offa[TileA::PT], offb[TileB::PT];
  const bool fast_tile = VEC && DBG == 0 && m0 + BM <= p.M && n0 + BN <= p.N;
  const int nfull = (kend - kbeg) / BK;
  if (nt > 0) {
    TileA::load(p.A, p.lda, m0, p.M, kbeg, kend, tid, ra);
    TileB::load(p.B, p.ldb, n0, p.N, kbeg, kend, tid, rb);
    TileA::store(smem, tid, ra, m0, p.M, kbeg, kend);
    TileB::store(smem + TileA::LDS_FLOATS, tid, rb, n0, p.N, kbeg, kend);
  }
  // (the second K-tile requested before the barrier, not after it: 2.705 against 2.720 ms per 512-row iteration)
  if (fast_tile) {
    TileA::lane_offsets(p.lda, tid, offa);
    TileB::lane_offsets(p.ldb, tid, offb);
    if (nfull > 1) {
      TileA::load_fast(TileA::ubase(p.A, p.lda, m0, kbeg + BK), offa, ra);
      TileB::load_fast(TileB::ubase(p.B, p.ldb, n0, kbeg + BK), offb, rb);
    }
  }
  __syncthreads();
  AGX_STAMP(1);

  // MFMAs of one staged K-tile: per 8-k group one ds_read_b128 per fragment, then 4 x TM x TN MFMAs.
  // The reads of group g+1 are issued before the MFMAs of group g (register double buffer).
  auto mma_tile = [&](const float *cA, const float *cB, auto &&after_first_group) {
    constexpr int NG = BK / 8;
    float4 fa[2][TM], fb[2][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) fa[0][i] = TileA::frag4(cA, (wm * TM + i) * 32 + l31, 0, lh);
#pragma unroll
    for (int j = 0; j < TN; ++j) fb[0][j] = TileB::frag4(cB, (wn * TN + j) * 32 + l31, 0, lh);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const int cur = g & 1, nxt = cur ^ 1;
      if (g + 1 < NG) {
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[nxt][i] = TileA::frag4(cA, (wm * TM + i) * 32 + l31, g + 1, lh);
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[nxt][j] = TileB::frag4(cB, (wn * TN + j) * 32 + l31, g + 1, lh);
      }
      __builtin_amdgcn_sched_barrier(0);  // keep the reads ahead of the MFMAs (hipcc sinks them otherwise)
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            const float4 &a4 = fa[cur][i], &b4 = fb[cur][j];
            const float av = e == 0 ? a4.x : (e == 1 ? a4.y : (e == 2 ? a4.z : a4.w));
            const float bv = e == 0 ? b4.x : (e == 1 ? b4.y : (e == 2 ? b4.z : b4.w));
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i][j], 0, 0, 0);
          }
      __builtin_amdgcn_sched_barrier(0);
      if (g == 0) {
        after_first_group();
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };
  auto nothing = [] {};

  int t = 0;
  if (fast_tile) {
    // interior tile: K-tiles 1 .. nfull-1 are loaded without clamps and stored without masks
    // (the first K-tile through the same path: no gain at K = 96, 0.4 % slower at K = 1536)
    // Staging pipeline: the registers always hold the tile AFTER the one being multiplied.  Inside
    // iteration t, right after the first MFMA group, tile t+1 goes from the registers into the other
    // LDS buffer (free since the barrier that ended iteration t-1) and the loads of tile t+2 are
    // issued into the same registers: every global load has a whole iteration to land, and the
    // store / load instructions issue between MFMA groups instead of in front of the barrier.
    for (; t + 1 < nfull; ++t) {
      const float *cA = smem + (t & 1) * STAGE;
      float *nA = smem + ((t + 1) & 1) * STAGE;
      const bool ahead = t + 2 < nfull;
      const int k2 = kbeg + (t + 2) * BK;
      // (the next-but-one tile's loads issued a few at a time behind the MFMA groups instead of in this one burst:
      // 115 against 124 TF at 4096^3, +1-3 % at K = 256, no change of the 512-row iteration)
      mma_tile(cA, cA + TileA::LDS_FLOATS, [&] {
        TileA::store_fast(nA, tid, ra);
        TileB::store_fast(nA + TileA::LDS_FLOATS, tid, rb);
        if (ahead) {
          TileA::load_fast(TileA::ubase(p.A, p.lda, m0, k2), offa, ra);
          TileB::load_fast(TileB::ubase(p.B, p.ldb, n0, k2), offb, rb);
        }
      });
      __syncthreads();
      syrk_flush(t + 1);
    }
  }
  for (; t < nt; ++t) {
    const float *cA = smem + (t & 1) * STAGE;
    const float *cB = cA + TileA::LDS_FLOATS;
    const bool more = (t + 1 < nt);
    if (more && DBG != 1 && DBG != 3) {
      const int k0 = kbeg + (t + 1) * BK;
      TileA::load(p.A, p.lda, m0, p.M, k0, kend, tid, ra);
      TileB::load(p.B, p.ldb, n0, p.N, k0, kend, tid, rb);
    }
    mma_tile(cA, cB, nothing);
    if (DBG >= 2) {  // timing ablation only (wrong results): no LDS refill, no barrier
      if (DBG == 2) {
#pragma unroll
        for (int q = 0; q < TileA::PT; ++q) asm volatile("" ::"v"(ra[q].x), "v"(ra[q].w));
#pragma unroll
        for (int q = 0; q < TileB::PT; ++q) asm volatile("" ::"v"(rb[q].x), "v"(rb[q].w));
      }
      continue;
    }
    if (more) {
      float *nA = smem + ((t + 1) & 1) * STAGE;
      const int k0 = kbeg + (t + 1) * BK;
      TileA::store(nA, tid, ra, m0, p.M, k0, kend);
      TileB::store(nA + TileA::LDS_FLOATS, tid, rb, n0, p.N, k0, kend);
    }
    __syncthreads();
    if (more) syrk_flush(t + 1);
  }
  if (SYRK) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] += acc2[SYRK ? i : 0][SYRK ? j : 0];
  }

  AGX_STAMP(2);
  if (SK && !sk_last) {
    // this share ends inside the tile: publish the accumulators as this workgroup's partial tile
    // WRITE-THROUGH (sc1) 16-byte stores: the partial leaves this XCD's L2 as it is stored, so the hand-off needs no
    // release fence - an agent-scope release is a write-back of the XCD's whole L2 (buffer_wbl2; MI355X_MICROARCH.md,
    // visibility table): 162 -> 150.6 us on the 512-row dW_1, and 80 -> 45 us for a stream-K tail launch under a fused
    // epilogue, whose L2 is full of freshly updated weights (profiles/r03_hybrid_streamk.txt)
    typedef unsigned sk_u32x4 __attribute__((ext_vector_type(4)));
    float *pbase = p.sk_part + (size_t)blockIdx.x * (BM * BN);
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(pbase, 0, BM * BN * 4, 0x00020000);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          sk_u32x4 v;
          v.x = __float_as_uint(acc[i][j][4 * q]);
          v.y = __float_as_uint(acc[i][j][4 * q + 1]);
          v.z = __float_as_uint(acc[i][j][4 * q + 2]);
          v.w = __float_as_uint(acc[i][j][4 * q + 3]);
          __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, (((i * TN + j) * 4 + q) * NT + tid) * 16, 0, 16);   // aux 16 = sc1
        }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave drains its stores
    __syncthreads();
    if (tid == 0)   // (an agent-scope atomic store is an sc1 store)
      __hip_atomic_store(p.sk_flags + blockIdx.x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    continue;   // (only the first share a workgroup computes can stop short of its tile's end)
  }
  if (SK && !sk_first) {
    // finisher of a tile whose earlier K ranges other workgroups hold: local ids [c_lo, sk_local) of this XCD
    int c_lo = sk_local - 1;
    while (c_lo > 0 && c_lo * sk_ix / sk_wx > sk_tile_it0) --c_lo;
    if (tid == 0) {
      for (int c = c_lo; c < sk_local; ++c) {
        unsigned *f = p.sk_flags + ((size_t)c * p.sk_nxcd + blockIdx.x % p.sk_nxcd);
        int spins = 0;
        while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
          __builtin_amdgcn_s_sleep(4);
          if (++spins > (1 << 18)) {   // bounded: give up, leave a mark the host can read, never hang
            __hip_atomic_store(p.sk_flags + p.sk_wgs, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
          }
        }
        __hip_atomic_store(f, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // consumed: ready for the next launch
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    // own accumulators, then the partials in ascending K order (a fixed order: reproducible); per contributor all
    // 16-B loads of the partial are issued together
    for (int c = c_lo; c < sk_local; ++c) {
      const float4 *src = reinterpret_cast<const float4 *>(p.sk_part) +
                          ((size_t)c * p.sk_nxcd + blockIdx.x % p.sk_nxcd) * (BM * BN / 4) + tid;
      float4 v[TM * TN * 4];
#pragma unroll
      for (int e = 0; e < TM * TN * 4; ++e) v[e] = src[(size_t)e * NT];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float4 u = v[(i * TN + j) * 4 + q];
            acc[i][j][4 * q] += u.x;
            acc[i][j][4 * q + 1] += u.y;
            acc[i][j][4 * q + 2] += u.z;
            acc[i][j][4 * q + 3] += u.w;
          }
    }
  }
  // ---- epilogue: C/D map of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
  // per-column epilogue operands of ALL sub-tiles, requested before the first store (see epilogue_column)
  float epre[TN][3];
  bool stored = false;
  if (!SYRK && !OPT) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int cc = min(n0 + (wn * TN + j) * 32 + l31, p.N - 1);
      epre[j][0] = p.bias ? p.bias[cc] : 0.f;
      epre[j][1] = p.col_scale ? p.col_scale[cc] : 1.f;
      epre[j][2] = p.col_scale ? p.col_shift[cc] : 0.f;
    }
    // Raw split-K slabs and plain Dense products (alpha, bias, activation): ONE straight-line block of stores
    // over all sub-tiles.  Inside the general per-sub-tile code below hipcc ends every sub-tile with
    // s_waitcnt vmcnt(0) (its optional side loads meet there), i.e. a full drain of the sub-tile's stores before
    // the next one starts: four write round trips per wave of a 128x128 tile, 25 us of epilogue per tile measured.
    const bool plain = !p.rowscale && !p.addrow && !p.mask && !p.accumulate && !p.rowsq && !p.col_scale;
    if (p.to_ws || plain) {
      stored = true;
      float *dst = p.C;
      int ldd = p.ldc;
      if (p.to_ws) {
        dst = p.ws + (p.ws_by_row ? (size_t)z * p.N : (size_t)z * p.M * p.N);
        ldd = p.ws_by_row ? p.splitk * p.N : p.N;
      }
      const bool raw = p.to_ws != 0;
      auto store_all = [&](auto interior) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            const int col = n0 + (wn * TN + j) * 32 + l31;
            const int rbase = m0 + (wm * TM + i) * 32 + 4 * lh;
            float *__restrict__ cb = dst + (size_t)rbase * ldd + col;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int ro = (r & 3) + 8 * (r >> 2);
              float x = acc[i][j][r];
              if (!raw) {
                x *= p.alpha;
                if (p.bias) x += epre[j][0];
                if (p.act) x = x > 0.f ? x : x * p.slope;
              }
              if (decltype(interior)::value || (rbase + ro < p.M && col < p.N)) cb[ro * ldd] = x;
            }
          }
        }
      };
      if (m0 + BM <= p.M && n0 + BN <= p.N) store_all(std::true_type{});
      else store_all(std::false_type{});
      if (AGX_DUAL_STORE && !raw && p.c2 && m0 < p.c2_rows) {   // dual store of the leading rows, behind all the main stores
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            const int col = n0 + (wn * TN + j) * 32 + l31;
            const int rbase = m0 + (wm * TM + i) * 32 + 4 * lh;
            float *__restrict__ cb = p.c2 + (size_t)rbase * p.ldc2 + col;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int ro = (r & 3) + 8 * (r >> 2);
              float x = acc[i][j][r] * p.alpha;
              if (p.bias) x += epre[j][0];
              if (p.act) x = x > 0.f ? x : x * p.slope;
              if (rbase + ro < min(p.M, p.c2_rows) && col < p.N) cb[ro * p.ldc2] = x;
            }
          }
        }
      }
    }
  }
  if (!stored) {
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + (wn * TN + j) * 32 + l31;
      const int rbase = m0 + (wm * TM + i) * 32 + 4 * lh;
      if (SYRK) {
        const int G = p.N;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = rbase + (r & 3) + 8 * (r >> 2);
          if (row < G && col < G) p.C[(size_t)row * p.ldc + col] = acc[i][j][r] * p.alpha;
        }
        if (tm != tn && col < G) {  // mirror into the lower triangle
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int row = rbase + 8 * q;
            float *d = p.C + (size_t)col * p.ldc + row;
            if (row + 3 < G && (p.ldc & 3) == 0) {
              *reinterpret_cast<float4 *>(d) =
                  make_float4(acc[i][j][4 * q] * p.alpha, acc[i][j][4 * q + 1] * p.alpha,
                              acc[i][j][4 * q + 2] * p.alpha, acc[i][j][4 * q + 3] * p.alpha);
            } else {
#pragma unroll
              for (int s = 0; s < 4; ++s)
                if (row + s < G) d[s] = acc[i][j][4 * q + s] * p.alpha;
            }
          }
        }
      } else {
        int rows[16];
        float v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          rows[r] = rbase + (r & 3) + 8 * (r >> 2);
          v[r] = acc[i][j][r];
        }
        if (p.to_ws) {   // (OPT instantiations only: the others took the block above)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            if (rows[r] < p.M && col < p.N) p.ws[slab_off(p, z, rows[r]) + col] = v[r];
        } else if (opt_interior) {
          // interior tile: every element is updated - no per-element bounds tests, one 64-bit base and
          // 32-bit row offsets (the epilogue's instruction stream is on the launch's critical path: a single
          // extra branch per element measured +4 us per launch)
          const float omr = 1.f - p.opt_rho;
          float *__restrict__ wb = p.opt_w + (size_t)rbase * p.ldc + col;
          float *__restrict__ sb = p.opt_s1 + (size_t)rbase * p.ldc + col;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int o = ((r & 3) + 8 * (r >> 2)) * p.ldc;
            const float g = v[r] * p.alpha;
            const float rr = p.opt_rho * ps[r] + omr * (g * g);
            sb[o] = rr;
            wb[o] = pw[r] - p.opt_lr * g / (sqrtf(rr) + p.opt_eps);
          }
        } else if (OPT_PRE) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float x = v[r] * p.alpha;
            if (rows[r] < p.M && col < p.N) {
              const size_t off = (size_t)rows[r] * p.ldc + col;
              if (rows[r] < p.opt_rows) opt_rmsprop_apply(p, off, x, pw[r], ps[r]);
              else p.C[off] = x;
            }
          }
        } else if (OPT && m0 + BM <= min(p.M, p.opt_rows) && n0 + BN <= p.N) {
          // interior tile of a multi-block wave tile: the 16 weights and 16 slots of this 32x32 block are requested
          // together (one round trip per block; the general form below takes four groups of four)
          const float omr = 1.f - p.opt_rho;
          float *__restrict__ wb = p.opt_w + (size_t)rbase * p.ldc + col;
          float *__restrict__ sb = p.opt_s1 + (size_t)rbase * p.ldc + col;
          float ow[16], os[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int o = ((r & 3) + 8 * (r >> 2)) * p.ldc;
            ow[r] = wb[o];
            os[r] = sb[o];
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int o = ((r & 3) + 8 * (r >> 2)) * p.ldc;
            const float g = v[r] * p.alpha;
            const float rr = p.opt_rho * os[r] + omr * (g * g);
            sb[o] = rr;
            wb[o] = ow[r] - p.opt_lr * g / (sqrtf(rr) + p.opt_eps);
          }
        } else if (OPT) {
          epilogue_column<16, true>(p, rows, col, v);
        } else {
          epilogue_column<16>(p, rows, col, v, epre[j]);
          if (p.rowsq) {  // deterministic per-slab partial of sum_col v^2 (only without split-K)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const float q = half_wave_sum_at31(v[r] * v[r]);
              // v is 0 outside the matrix; lane 31's column names the slab.  A 32-column block that lies wholly
              // beyond N (N mod 64 in 1..32 under a 64-wide tile) has no slab: its index is the NEXT row's slab 0
              if (l31 == 31 && rows[r] < p.M && (col & ~31) < p.N)
                p.rowsq[(size_t)rows[r] * p.rowsq_ld + (col >> 5)] = q;
            }
          }
        }
      }
    }
  }
  }
#ifdef AGX_STAMPS
  AGX_STAMP(3);
  __builtin_amdgcn_s_waitcnt(0);
  AGX_STAMP(4);
  if (AGX_STAMP_ON) {
    long long *o = g_stamps + 8 * (size_t)(blockIdx.x + gridDim.x * blockIdx.y);
    for (int i = 0; i < 5; ++i) o[i] = stamp_v[i];
    o[5] = __builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_REG_HW_ID
    o[6] = __builtin_amdgcn_s_getreg((31 << 11) | 20);   // HW_REG_XCC_ID
    o[7] = tm * 100000 + tn * 100 + z;
  }
#endif
  } while (SK && sk_it < sk_end);   // stream-K shares (one pass otherwise)
}

// Split-K reduction + epilogue.  blockDim 256 = (256/PH lanes over 4-element chunks) x (PH split
// phases): each thread sums every PH-th slab for its 4 consecutive outputs, the phases meet in LDS
// in a fixed order (deterministic).  PH = 16 keeps the grid large when M*N is small (batch-32 regime).
// INTERP: the slabs hold the pre-activations of [real; fake] (M = 2R rows); every thread sums both
// rows b and R+b and also emits row 2R+b = r + alpha[b]*(f - r): the first Dense layer is linear,
// so this equals the layer applied to the interpolated input real + alpha*(fake - real)
// (tf_utils.py:162-167) without a third pass over the weights.
template <bool VEC4, int PH, bool INTERP>
__global__ void __launch_bounds__(256) splitk_reduce_kernel(const KParams p) {
  constexpr int LANES = 256 / PH;
  constexpr int NV = INTERP ? 2 : 1;
  __shared__ float4 red[NV][PH][LANES];
  const int rows_out = INTERP ? p.interp_rows : p.M;
  const size_t mn = (size_t)p.M * p.N;
  const size_t mn_out = (size_t)rows_out * p.N;
  const int tx = threadIdx.x % LANES, ty = threadIdx.x / LANES;
  const size_t idx = ((size_t)blockIdx.x * LANES + tx) * 4;
  float4 a[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) a[v] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (idx < mn_out) {
    for (int s = ty; s < p.splitk; s += PH) {
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const float *src = p.ws + (size_t)s * mn + (size_t)v * mn_out + idx;
        if (VEC4) {
          const float4 q = *reinterpret_cast<const float4 *>(src);
          a[v].x += q.x; a[v].y += q.y; a[v].z += q.z; a[v].w += q.w;
        } else {
          a[v].x += src[0];
          if (idx + 1 < mn_out) a[v].y += src[1];
          if (idx + 2 < mn_out) a[v].z += src[2];
          if (idx + 3 < mn_out) a[v].w += src[3];
        }
      }
    }
  }
#pragma unroll
  for (int v = 0; v < NV; ++v) red[v][ty][tx] = a[v];
  __syncthreads();
  if (ty == 0 && idx < mn_out) {
    float vals[NV][4];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      float4 t = red[v][0][tx];
#pragma unroll
      for (int k = 1; k < PH; ++k) {
        const float4 q = red[v][k][tx];
        t.x += q.x; t.y += q.y; t.z += q.z; t.w += q.w;
      }
      vals[v][0] = t.x; vals[v][1] = t.y; vals[v][2] = t.z; vals[v][3] = t.w;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (idx + e >= mn_out) break;
      const int row = (int)((idx + e) / p.N), col = (int)((idx + e) % p.N);
      if (INTERP) {
        const float r = vals[0][e], f = vals[NV - 1][e];
        const float diff = f - r;
        const float it = r + p.interp_alpha[row] * diff;
        epilogue_store(p, row, col, r);
        epilogue_store(p, rows_out + row, col, f);
        epilogue_store(p, 2 * rows_out + row, col, it);
      } else {
        epilogue_store(p, row, col, vals[0][e]);
      }
    }
  }
}

template <bool INTERP>
int launch_reduce(hipStream_t st, const KParams &p) {
  const int64_t mn_out = (int64_t)(INTERP ? p.interp_rows : p.M) * p.N;
  const bool vec = (mn_out % 4 == 0) && agx::aligned16(p.ws) && (((int64_t)p.M * p.N) % 4 == 0);
  // small outputs: 16 phases x 16 lanes (64 outputs per block) keeps >= a few hundred blocks in flight
  const bool many_phases = mn_out <= (int64_t)1 << 16 && p.splitk >= 8;
  const int per_block = many_phases ? 64 : 256;
  const int blocks = (int)agx::cdiv64(mn_out, per_block);
#define AGX_RED(V, P) hipLaunchKernelGGL((splitk_reduce_kernel<V, P, INTERP>), dim3(blocks), dim3(256), 0, st, p)
  if (many_phases) { if (vec) AGX_RED(true, 16); else AGX_RED(false, 16); }
  else             { if (vec) AGX_RED(true, 4);  else AGX_RED(false, 4); }
#undef AGX_RED
  AGX_LAUNCH_CHECK();
  return AGX_OK;
}

// ---------------------------------------------------------------------------
// Latency-oriented kernel for skinny / small products (batch 32 regime and the hidden layers):
// one 32x32 output tile per 4-wave block, the block's K range is split over the 4 waves in
// interleaved groups of 8, every lane loads its MFMA fragments STRAIGHT from global memory
// (k-contiguous operand: one 16-B load = 4 consecutive k of the lane's row; row-contiguous
// operand: 4 coalesced dword loads), U groups are issued before the first MFMA, so a wave pays
// one memory round trip per 8*U k instead of one per staged K-tile.  The k order inside a
// group is permuted identically for A and B (lane half h takes k = 4h..4h+3), which leaves the
// product unchanged.  The 4 partial tiles meet in LDS; split-K over blocks uses the same
// partial-slab + reduce path as the tiled kernel.
// ---------------------------------------------------------------------------
// branch-free: addresses clamped into the operand, masking by select after the load
template <bool KC, bool VEC>
__device__ __forceinline__ void frag_load4(const float *__restrict__ src, int ld, int r, int R, int k,
                                           int kend, float (&v)[4]) {
  const int kl = kend - 1;
  const int cr = min(r, R - 1);
  if (KC) {
    const float *s = src + (size_t)cr * ld;
    if (VEC) {
      const float4 q = *reinterpret_cast<const float4 *>(s + min(k, kl & ~3));
      v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
    } else {
      v[0] = s[min(k, kl)]; v[1] = s[min(k + 1, kl)]; v[2] = s[min(k + 2, kl)]; v[3] = s[min(k + 3, kl)];
    }
  } else {
    const float *s = src + cr;
    v[0] = s[(size_t)min(k, kl) * ld];
    v[1] = s[(size_t)min(k + 1, kl) * ld];
    v[2] = s[(size_t)min(k + 2, kl) * ld];
    v[3] = s[(size_t)min(k + 3, kl) * ld];
  }
  const bool rv = r < R;
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] = (rv && k + e < kend) ? v[e] : 0.f;
}

// Interior fast path: `ub` is a wave-uniform base (scalar registers: tile origin + k of the group),
// `lo` the lane's 32-bit element offset computed once per kernel, so a load costs no vector address
// arithmetic (the clamped form above spends ~8 VALU instructions per dword on min / 64-bit mul-add,
// more issue time than the MFMA it feeds).  No masking: callers use it only for full tiles / groups.
template <bool KC, bool VEC>
__device__ __forceinline__ void frag_fast4(const float *__restrict__ ub, int lo, int ld, float (&v)[4]) {
  if (KC) {
    if (VEC) {
      const float4 q = *reinterpret_cast<const float4 *>(ub + lo);
      v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
    } else {
      v[0] = ub[lo]; v[1] = ub[lo + 1]; v[2] = ub[lo + 2]; v[3] = ub[lo + 3];
    }
  } else {
    v[0] = ub[lo];
    v[1] = (ub + ld)[lo];
    v[2] = (ub + 2 * (size_t)ld)[lo];
    v[3] = (ub + 3 * (size_t)ld)[lo];
  }
}
// uniform base of the 8-k group starting at absolute k for the 32-row slab starting at row r0
template <bool KC>
__device__ __forceinline__ const float *frag_ubase(const float *src, int ld, int r0, int k) {
  return KC ? src + (size_t)r0 * ld + k : src + (size_t)k * ld + r0;
}
template <bool KC>
__device__ __forceinline__ int frag_lane_off(int ld, int i, int h) {
  return KC ? i * ld + 4 * h : 4 * h * ld + i;
}

// NW = 8 (long K ranges): two waves per SIMD, so one wave's loads are in flight while the other's MFMAs issue
// STAGED B operand (8-wave instantiations, 16-B aligned operands): the streamed weight matrix is not loaded in MFMA
// fragment order - k-contiguous (NT): lane (row, half) takes 16 B of its row per instruction, so one instruction
// touches 32 lines for 32 B each and every line is requested by four instructions; row-contiguous (NN): four dword
// loads per 8-k group - but in LINE order (float4, one instruction = eight whole 128-B lines: the wave's 8U k of 32
// rows in U instructions) and turned into fragments through a wave-private LDS block in the k-quad image of the tiled
// kernel (no workgroup barrier: a wave's LDS instructions execute in order).  Same values in the same lanes, so the
// product is bit-identical.  One pass over W_1 out of the Infinity Cache: 3.9 us in fragment order, 2.45 us in line
// order (tools/micro/stream_patterns.hip; L2 keeps nothing across launches: FETCH_SIZE = the whole matrix on every
// launch).  Measured (profiles/r02_direct_kernel_staging.txt): 64x256x18161 14.9 -> 14.0 us, 1024x256x18161 136 ->
// 119 us, 32x20000x1024 NT 30.4 -> 26.4 us, 32-row iteration 0.4956 -> 0.4887 ms.  Staging the A operand the same way
// (bit 2) loses: 16.2 us and 0.530 ms - its re-reads hit L2, the extra LDS hop only adds latency to every pass.
// What the input-gradient product (32 x 18154 x 256, one pass) spends beyond streaming W_1, part by part
// (tools/micro/dx_parts.hip, profiles/r02_direct_kernel_staging.txt): B loads 3.9 us (2.4 in line order), + the A
// fragments - the same 32 x 256 matrix for all 568 workgroups, L2 hits, but as many bytes into the CUs as B - 6.5,
// + MFMAs 6.8, + cross-wave reduction 6.9, + tile stored 7.4 us.  Tried against the A term and dropped: two column
// tiles per workgroup sharing the A fragments (284 workgroups: 18.0 against 9.2 us with the accumulators as an array,
// which hipcc parks in scratch; 14.6 us with two named accumulators, with or without a 128-register cap for two
// workgroups per CU - half as many waves to keep the stream in flight, two reductions in a row); the ragged last
// tile shifted into the interior path (9.08 against 9.12 us: the edge workgroup is not the long pole).
// AGX_DIRECT_STAGE bits: 1 = B of NT, 4 = B of NN, 2 = A (k-contiguous)
#ifndef AGX_DIRECT_STAGE
#define AGX_DIRECT_STAGE 5
#endif
template <bool TA, bool TB, bool VEC, int U, int NW = 4>
__global__ void __launch_bounds__(NW * 64) gemm_direct_kernel(const KParams p) {
  constexpr bool STAGE_B = NW == 8 && VEC && (TB ? (AGX_DIRECT_STAGE & 1) != 0 : (AGX_DIRECT_STAGE & 4) != 0);
  // both operands only while the two blocks of the 8 waves fit the 64 KB of static LDS (U = 4)
  constexpr bool STAGE_A = (AGX_DIRECT_STAGE & 2) && NW == 8 && VEC && !TA && (!STAGE_B || U <= 4);
  constexpr int NSTG = (STAGE_A ? 1 : 0) + (STAGE_B ? 1 : 0);
  using WT = Tile<32, true, 64, true, 8 * U>;   // one wave's block: 32 rows x 8U k, PT = U float4 per lane
  using WTB = Tile<32, TB, 64, true, 8 * U>;    // B: k-contiguous (NT) or row-contiguous (NN: 8 k-rows x 128 B per instruction)
  constexpr int SM_RED = NW * 32 * 33, SM_STG = NW * NSTG * WT::LDS_FLOATS;
  __shared__ __attribute__((aligned(16))) float smem_d[SM_RED > SM_STG ? SM_RED : SM_STG];
  float (*red)[32][33] = reinterpret_cast<float (*)[32][33]>(smem_d);   // (after the passes: the staging blocks are dead)
#ifdef AGX_STAMPS
  long long stamp_v[5] = {0, 0, 0, 0, 0};   // start | first pass multiplied | all passes | stores issued | drained
  bool first_pass = true;
#endif
  AGX_STAMP(0);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 31, h = lane >> 5;
  const int tiles_n = (p.N + 31) / 32;
  const int lid = agx::xcd_logical_id(blockIdx.x + gridDim.x * blockIdx.y, gridDim.x * gridDim.y);
  const int bt = lid % (int)gridDim.x, z = lid / (int)gridDim.x;  // all tiles of one K slice share an XCD's L2
  const int m0 = (bt / tiles_n) * 32, n0 = (bt % tiles_n) * 32;
  const int kbeg = z * p.kchunk;
  const int kend = min(p.K, kbeg + p.kchunk);
  const int ngroups = (kend - kbeg + 7) / 8;

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;

  // pass p covers groups [4U*p, 4U*(p+1)); wave w takes U consecutive ones, so that a wave's
  // loads of a k-contiguous row walk whole cache lines
  int g0 = wave * U;
  if (m0 + 32 <= p.M && n0 + 32 <= p.N) {  // interior tile: full k-groups without clamps or masks
    const int nfull = (kend - kbeg) / 8;
    const int la = frag_lane_off<!TA>(p.lda, i, h), lb = frag_lane_off<TB>(p.ldb, i, h);
    int offa[WT::PT], offb[WT::PT];
    if (STAGE_A) WT::lane_offsets(p.lda, lane, offa);
    if (STAGE_B) WTB::lane_offsets(p.ldb, lane, offb);
    float *stg_a = smem_d + wave * NSTG * WT::LDS_FLOATS;
    float *stg_b = stg_a + (STAGE_A ? WT::LDS_FLOATS : 0);
    for (; g0 + U <= nfull; g0 += NW * U) {
      float av[U][4], bv[U][4];
      float4 ra[WT::PT], rb[WT::PT];
      if (STAGE_A) WT::load_fast(WT::ubase(p.A, p.lda, m0, kbeg + 8 * g0), offa, ra);
      if (STAGE_B) WTB::load_fast(WTB::ubase(p.B, p.ldb, n0, kbeg + 8 * g0), offb, rb);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int k = kbeg + 8 * (g0 + u);
        if (!STAGE_A) frag_fast4<!TA, VEC>(frag_ubase<!TA>(p.A, p.lda, m0, k), la, p.lda, av[u]);
        if (!STAGE_B) frag_fast4<TB, VEC>(frag_ubase<TB>(p.B, p.ldb, n0, k), lb, p.ldb, bv[u]);
      }
      if (NSTG > 0) {
        // (the previous pass's fragment reads precede these writes in program order: in-order LDS, same wave)
        if (STAGE_A) WT::store_fast(stg_a, lane, ra);
        if (STAGE_B) WTB::store_fast(stg_b, lane, rb);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int u = 0; u < U; ++u) {
          if (STAGE_A) {
            const float4 q = WT::frag4(stg_a, i, u, h);
            av[u][0] = q.x; av[u][1] = q.y; av[u][2] = q.z; av[u][3] = q.w;
          }
          if (STAGE_B) {
            const float4 q = WTB::frag4(stg_b, i, u, h);
            bv[u][0] = q.x; bv[u][1] = q.y; bv[u][2] = q.z; bv[u][3] = q.w;
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u][e], bv[u][e], acc, 0, 0, 0);
#ifdef AGX_STAMPS
      if (first_pass) {
        AGX_STAMP(1);
        first_pass = false;
      }
#endif
    }
  }
  for (; g0 < ngroups; g0 += NW * U) {
    float av[U][4], bv[U][4];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int k = kbeg + 8 * (g0 + u) + 4 * h;
      frag_load4<!TA, VEC>(p.A, p.lda, m0 + i, p.M, k, kend, av[u]);
      frag_load4<TB, VEC>(p.B, p.ldb, n0 + i, p.N, k, kend, bv[u]);
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int e = 0; e < 4; ++e)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u][e], bv[u][e], acc, 0, 0, 0);
  }
  AGX_STAMP(2);
  if (NSTG > 0) __syncthreads();   // the partial sums below overwrite other waves' staging blocks
#pragma unroll
  for (int r = 0; r < 16; ++r) red[wave][(r & 3) + 8 * (r >> 2) + 4 * h][i] = acc[r];
  __syncthreads();
  {
    const int cc = tid & 31, col = n0 + cc;
    constexpr int NQ = 16 / NW;   // rows per thread: 32 rows over NW*2 thread rows
    int rows[NQ];
    float v[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int rr = (tid >> 5) + 2 * NW * q;
      rows[q] = m0 + rr;
      float t = red[0][rr][cc] + red[1][rr][cc] + red[2][rr][cc] + red[3][rr][cc];
      if constexpr (NW == 8) t += red[4][rr][cc] + red[5][rr][cc] + red[6][rr][cc] + red[7][rr][cc];
      v[q] = t;
    }
    if (p.to_ws) {
#pragma unroll
      for (int q = 0; q < NQ; ++q)
        if (rows[q] < p.M && col < p.N) p.ws[slab_off(p, z, rows[q]) + col] = v[q];
    } else {
      epilogue_column<NQ>(p, rows, col, v);
      if (p.rowsq) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          const float sq = half_wave_sum_at31(v[q] * v[q]);
          if (cc == 31 && rows[q] < p.M && (col & ~31) < p.N) p.rowsq[(size_t)rows[q] * p.rowsq_ld + (col >> 5)] = sq;
        }
      }
    }
  }
#ifdef AGX_STAMPS
  AGX_STAMP(3);
  __builtin_amdgcn_s_waitcnt(0);
  AGX_STAMP(4);
  if (AGX_STAMP_ON) {
    long long *o = g_stamps + 8 * (size_t)(blockIdx.x + gridDim.x * blockIdx.y);
    for (int q = 0; q < 5; ++q) o[q] = stamp_v[q];
    o[5] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
    o[6] = __builtin_amdgcn_s_getreg((31 << 11) | 20);
    o[7] = (m0 / 32) * 100000 + (n0 / 32) * 100 + z;
  }
#endif
}

// ---------------------------------------------------------------------------
// Wave-granular kernel for products whose output is large and whose K is short (weight
// gradients at batch 32: K = 3B = 96; wide outputs with K = H): ONE WAVE = one
// (32 TM) x (32 TN) output tile over the whole K slice, MFMA fragments straight from global
// memory / L2, no LDS, no barrier, register double buffering (the loads of the next U k-groups
// are in flight while the MFMAs of the current ones issue).  Work is scheduled per wave, so
// a grid of thousands of small independent tiles balances over the 1024 SIMDs where 128x128
// workgroup tiles quantise badly (284 tiles on 256 CUs).
// ---------------------------------------------------------------------------
template <bool TA, bool TB, bool VEC, int TM, int TN, int U>
__global__ void __launch_bounds__(64) gemm_wave_kernel(const KParams p) {
  const int lane = threadIdx.x, i = lane & 31, h = lane >> 5;
  const int tiles_n = (p.N + 32 * TN - 1) / (32 * TN);
  const int lid = agx::xcd_logical_id(blockIdx.x + gridDim.x * blockIdx.y, gridDim.x * gridDim.y);
  const int bt = lid % (int)gridDim.x, z = lid / (int)gridDim.x;
  const int m0 = (bt / tiles_n) * 32 * TM, n0 = (bt % tiles_n) * 32 * TN;
  const int kbeg = z * p.kchunk;
  const int kend = min(p.K, kbeg + p.kchunk);
  const int ngroups = (kend - kbeg + 7) / 8;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  float av0[U][TM][4], bv0[U][TN][4], av1[U][TM][4], bv1[U][TN][4];
#define AGX_WLOAD(AV, BV, G0)                                                              \
  _Pragma("unroll") for (int u = 0; u < U; ++u) {                                          \
    const int k = kbeg + 8 * ((G0) + u) + 4 * h;                                           \
    _Pragma("unroll") for (int a = 0; a < TM; ++a)                                         \
        frag_load4<!TA, VEC>(p.A, p.lda, m0 + 32 * a + i, p.M, k, kend, AV[u][a]);         \
    _Pragma("unroll") for (int b = 0; b < TN; ++b)                                         \
        frag_load4<TB, VEC>(p.B, p.ldb, n0 + 32 * b + i, p.N, k, kend, BV[u][b]);          \
  }
#define AGX_WFAST(AV, BV, G0)                                                              \
  _Pragma("unroll") for (int u = 0; u < U; ++u) {                                          \
    const int k = kbeg + 8 * ((G0) + u);                                                   \
    _Pragma("unroll") for (int a = 0; a < TM; ++a)                                         \
        frag_fast4<!TA, VEC>(frag_ubase<!TA>(p.A, p.lda, m0 + 32 * a, k), la, p.lda, AV[u][a]); \
    _Pragma("unroll") for (int b = 0; b < TN; ++b)                                         \
        frag_fast4<TB, VEC>(frag_ubase<TB>(p.B, p.ldb, n0 + 32 * b, k), lb, p.ldb, BV[u][b]);   \
  }
#define AGX_WMMA(AV, BV)                                                                   \
  _Pragma("unroll") for (int u = 0; u < U; ++u)                                            \
  _Pragma("unroll") for (int e = 0; e < 4; ++e)                                            \
  _Pragma("unroll") for (int a = 0; a < TM; ++a)                                           \
  _Pragma("unroll") for (int b = 0; b < TN; ++b)                                           \
      acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(AV[u][a][e], BV[u][b][e], acc[a][b], 0, 0, 0);
  int gdone = 0;
  if (m0 + 32 * TM <= p.M && n0 + 32 * TN <= p.N) {
    // interior tile: batches of U full k-groups through the clamp-free loads
    const int nfast = ((kend - kbeg) / 8 / U) * U;
    const int la = frag_lane_off<!TA>(p.lda, i, h), lb = frag_lane_off<TB>(p.ldb, i, h);
    if (nfast > 0) {
      AGX_WFAST(av0, bv0, 0)
      for (int g0 = 0; g0 < nfast; g0 += 2 * U) {
        if (g0 + U < nfast) { AGX_WFAST(av1, bv1, g0 + U) }
        AGX_WMMA(av0, bv0)
        if (g0 + U < nfast) {
          if (g0 + 2 * U < nfast) { AGX_WFAST(av0, bv0, g0 + 2 * U) }
          AGX_WMMA(av1, bv1)
        }
      }
    }
    gdone = nfast;
  }
  for (int g0 = gdone; g0 < ngroups; g0 += U) {  // edge tiles and the K tail: clamped + masked loads
    AGX_WLOAD(av0, bv0, g0)
    AGX_WMMA(av0, bv0)
  }
#undef AGX_WLOAD
#undef AGX_WFAST
#undef AGX_WMMA

#pragma unroll
  for (int a = 0; a < TM; ++a) {
#pragma unroll
    for (int b = 0; b < TN; ++b) {
      const int col = n0 + 32 * b + i;
      const int rbase = m0 + 32 * a + 4 * h;
      int rows[16];
      float v[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        rows[r] = rbase + (r & 3) + 8 * (r >> 2);
        v[r] = acc[a][b][r];
      }
      if (p.to_ws) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (rows[r] < p.M && col < p.N) p.ws[slab_off(p, z, rows[r]) + col] = v[r];
      } else {
        epilogue_column<16>(p, rows, col, v);
        if (p.rowsq) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float q = half_wave_sum_at31(v[r] * v[r]);
            if (i == 31 && rows[r] < p.M && (col & ~31) < p.N) p.rowsq[(size_t)rows[r] * p.rowsq_ld + (col >> 5)] = q;
          }
        }
      }
    }
  }
}

template <bool TA, bool TB, int TM, int TN>
int launch_wave_t(hipStream_t st, const KParams &p, bool vec) {
  const int tiles = agx::cdiv(p.M, 32 * TM) * agx::cdiv(p.N, 32 * TN);
  dim3 grid(tiles, p.splitk, 1), block(64, 1, 1);
  constexpr int U = (TM + TN <= 3) ? 4 : 2;
  if (vec) hipLaunchKernelGGL((gemm_wave_kernel<TA, TB, true, TM, TN, U>), grid, block, 0, st, p);
  else hipLaunchKernelGGL((gemm_wave_kernel<TA, TB, false, TM, TN, U>), grid, block, 0, st, p);
  AGX_LAUNCH_CHECK();
  return AGX_OK;
}

template <bool TA, bool TB>
int launch_wave(hipStream_t st, const KParams &p, bool vec, int tm, int tn) {
  if (tm == 1 && tn == 1) return launch_wave_t<TA, TB, 1, 1>(st, p, vec);
  if (tm == 1 && tn == 2) return launch_wave_t<TA, TB, 1, 2>(st, p, vec);
  if (tm == 2 && tn == 1) return launch_wave_t<TA, TB, 2, 1>(st, p, vec);
  if (tm == 2 && tn == 2) return launch_wave_t<TA, TB, 2, 2>(st, p, vec);
  if (tm == 3 && tn == 1) return launch_wave_t<TA, TB, 3, 1>(st, p, vec);
  if (tm == 1 && tn == 4) return launch_wave_t<TA, TB, 1, 4>(st, p, vec);
  if (tm == 4 && tn == 1) return launch_wave_t<TA, TB, 4, 1>(st, p, vec);
  return launch_wave_t<TA, TB, 3, 2>(st, p, vec);
}

template <bool TA, bool TB>
int launch_direct(hipStream_t st, const KParams &p, int tiles, bool vec, int groups_per_wave) {
  dim3 grid(tiles, p.splitk, 1), block(256, 1, 1);
  // long K range per workgroup (the 18k-gene first-layer products at batch 32): 8 waves share it
  constexpr int nw8_min = 200;
  if (std::min(p.kchunk, p.K) >= nw8_min) {
    // (8 k-groups per wave per pass whatever the range - 9, 12 or 18 in flight: +5..9 us per iteration.  Re-measured with
    // phase stamps in round 2: alone, 9 groups save a pass on the 18k-gene products (13.0 -> 11.2 us and 8.4 -> 6.7 us,
    // tools/stamps_direct.sh), inside the iteration they cost 4 us (0.5222 against 0.5184 ms), 10 groups cost 16 us; 9 only
    // where that saves a pass: 0.5193 against 0.5148 ms - in the graph a pass is bandwidth-, not latency-bound: the
    // two longer passes of the first layer take 4.5 + 4.7 us against 4.2 + 5.4 us for the three shorter ones; balancing the passes over the waves - 144 groups as
    // 3 x 6 per wave instead of 64 + 64 + 16 - measured 5 us per iteration SLOWER; 16 waves: no gain)
    dim3 block8(512, 1, 1);
    if (agx::cdiv(std::min(p.kchunk, p.K), 8) <= 32) {   // <= 4 groups per wave: all 8 waves get work
      if (vec) hipLaunchKernelGGL((gemm_direct_kernel<TA, TB, true, 4, 8>), grid, block8, 0, st, p);
      else hipLaunchKernelGGL((gemm_direct_kernel<TA, TB, false, 4, 8>), grid, block8, 0, st, p);
    } else if (vec) hipLaunchKernelGGL((gemm_direct_kernel<TA, TB, true, 8, 8>), grid, block8, 0, st, p);
    else hipLaunchKernelGGL((gemm_direct_kernel<TA, TB, false, 8, 8>), grid, block8, 0, st, p);
    AGX_LAUNCH_CHECK();
    return AGX_OK;
  }
#define AGX_DIRECT(V, UU) hipLaunchKernelGGL((gemm_direct_kernel<TA, TB, V, UU>), grid, block, 0, st, p)
#define AGX_DIRECT_U(V)                         \
  do {                                          \
    if (groups_per_wave <= 1) AGX_DIRECT(V, 1); \
    else if (groups_per_wave <= 2) AGX_DIRECT(V, 2); \
    else if (groups_per_wave <= 4) AGX_DIRECT(V, 4); \
    else if (groups_per_wave <= 6 || (groups_per_wave > 8 && groups_per_wave <= 12)) AGX_DIRECT(V, 6); \
    else AGX_DIRECT(V, 8);                      \
  } while (0)
  if (vec) AGX_DIRECT_U(true); else AGX_DIRECT_U(false);
#undef AGX_DIRECT_U
#undef AGX_DIRECT
  AGX_LAUNCH_CHECK();
  return AGX_OK;
}

template <int BM, int BN, int WM, int WN, bool TA, bool TB, bool VEC, bool SYRK, int BK = 32, bool OPT = false,
          bool OPTI = false, bool SK = false>
int launch(hipStream_t st, const KParams &p, int tiles) {
  using TileA = Tile<BM, !TA, WM * WN * 64, VEC, BK>;
  using TileB = Tile<BN, TB, WM * WN * 64, VEC, BK>;
  constexpr size_t lds = 2 * (size_t)(TileA::LDS_FLOATS + TileB::LDS_FLOATS) * sizeof(float);
  auto kern = gemm_kernel<BM, BN, WM, WN, TA, TB, VEC, SYRK, BK, 0, OPT, OPTI, SK>;  // LDS limit raised in agx_init
  dim3 grid(tiles + ((OPT && p.rest_n > 0) ? p.opt_rest_blocks + p.opt_copy_blocks : 0), p.splitk, 1), block(WM * WN * 64, 1, 1);
  hipLaunchKernelGGL(kern, grid, block, lds, st, p);
  AGX_LAUNCH_CHECK();
  return AGX_OK;
}

template <int BM, int BN, int WM, int WN, bool TA, bool TB, bool VEC, bool SYRK, int BK = 32, bool OPT = false,
          bool OPTI = false, bool SK = false>
int prepare_one() {
  using TileA = Tile<BM, !TA, WM * WN * 64, VEC, BK>;
  using TileB = Tile<BN, TB, WM * WN * 64, VEC, BK>;
  constexpr size_t lds = 2 * (size_t)(TileA::LDS_FLOATS + TileB::LDS_FLOATS) * sizeof(float);
  AGX_HIP(hipFuncSetAttribute(
      reinterpret_cast<const void *>(gemm_kernel<BM, BN, WM, WN, TA, TB, VEC, SYRK, BK, 0, OPT, OPTI, SK>),
      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  return AGX_OK;
}

template <int BM, int BN, int WM, int WN, int BK = 32>
int prepare_cfg() {
  AGX_TRY((prepare_one<BM, BN, WM, WN, false, false, true, false, BK>()));
  AGX_TRY((prepare_one<BM, BN, WM, WN, false, false, false, false, BK>()));
  AGX_TRY((prepare_one<BM, BN, WM, WN, false, true, true, false, BK>()));
  AGX_TRY((prepare_one<BM, BN, WM, WN, false, true, false, false, BK>()));
  AGX_TRY((prepare_one<BM, BN, WM, WN, true, false, true, false, BK>()));
  AGX_TRY((prepare_one<BM, BN, WM, WN, true, false, false, false, BK>()));
  return AGX_OK;
}

template <int BM, int BN, int WM, int WN, int BK = 32>
int launch_layout(hipStream_t st, const KParams &p, int tiles, int ta, int tb, bool vec) {
  if (!ta && !tb)
    return vec ? launch<BM, BN, WM, WN, false, false, true, false, BK>(st, p, tiles)
               : launch<BM, BN, WM, WN, false, false, false, false, BK>(st, p, tiles);
  if (!ta && tb)
    return vec ? launch<BM, BN, WM, WN, false, true, true, false, BK>(st, p, tiles)
               : launch<BM, BN, WM, WN, false, true, false, false, BK>(st, p, tiles);
  return vec ? launch<BM, BN, WM, WN, true, false, true, false, BK>(st, p, tiles)
             : launch<BM, BN, WM, WN, true, false, false, false, BK>(st, p, tiles);
}


struct Plan {
  int cfg, bm, bn, tiles, splitk, kchunk;
  int wtm, wtn;  // wave-kernel tile (cfg >= 8)
  int sk_wgs;    // > 0: stream-K over this many workgroups (128x128 tiles, TN)
};

// Stream-K applies (see gemm_kernel): aligned TN product, heuristic tile choice, no split-K, enough 128x128 tiles to
// give every XCD work but a tile count that leaves the best static tiling below 92 % of its rounds, and shares of
// >= 8 K-tiles that are no shorter than half a tile's K range (<= 3 partials per tile).
int streamk_wgs(int M, int N, int K, int layout, int force_cfg, int force_splitk) {
  static const bool off = getenv("AGX_NO_STREAMK") != nullptr;
  if (off || layout != 2 || force_cfg >= 0 || force_splitk > 0) return 0;
  const int cus = agx::num_cus();
  if (cus % 8 != 0) return 0;
  const long long t128 = (long long)agx::cdiv(M, 128) * agx::cdiv(N, 128);
  const long long t64 = (long long)agx::cdiv(M, 64) * agx::cdiv(N, 64);
  const int kt = agx::cdiv(K, BK);
  const int wgs = 2 * cus;
  if (t128 < cus / 2 || t128 > 16 * cus || kt < 32) return 0;   // (K = 512: 77 against 62 us for the static 64x64 tiling)
  const double e128 = (double)t128 / (double)(agx::cdiv64(t128, 2 * cus) * 2 * cus);
  const double e64 = (double)t64 / (double)(agx::cdiv64(t64, 4 * cus) * 4 * cus);
  if (std::max(e128, e64) >= 0.92) return 0;
  const long long per = t128 / 8 * kt / (wgs / 8);   // (lower bound of a workgroup's share)
  // (shares of at least half a tile's K range, i.e. <= 3 partials per tile: 9088 x 256 x 1536 with four per tile ran
  //  105 against 97 us for the static tiling, 18161 x 256 x 1536 with two to three 162 against 179 us)
  if (per < 8 || per * 2 < kt) return 0;
  if ((t128 / 8 + 1) * kt * (wgs / 8) >= (1ll << 31)) return 0;   // the kernel's share arithmetic is 32-bit
  return wgs;
}

const int kWaveTiles[8][2] = {{1, 1}, {1, 2}, {2, 1}, {2, 2}, {3, 1}, {1, 4}, {4, 1}, {3, 2}};

// layout: 0 NN, 1 NT, 2 TN, -1 unknown (workspace bound: the larger split of the candidates)
Plan make_plan(int M, int N, int K, int force_cfg, int force_splitk, int64_t ws_floats, int layout = -1,
               bool allow_sk = false) {
  Plan pl;
  pl.sk_wgs = 0;
  if (allow_sk) {
    const int wgs = streamk_wgs(M, N, K, layout, force_cfg, force_splitk);
    if (wgs > 0 && (int64_t)wgs * 128 * 128 <= ws_floats) {
      pl.cfg = 0; pl.bm = pl.bn = 128; pl.wtm = pl.wtn = 1;
      pl.tiles = agx::cdiv(M, 128) * agx::cdiv(N, 128);
      pl.splitk = 1;
      pl.kchunk = agx::cdiv(K > 0 ? K : 1, BK) * BK;
      pl.sk_wgs = wgs;
      return pl;
    }
  }
  const int cus = agx::num_cus();
  const int ktiles = agx::cdiv(K, BK);
  int cfg = force_cfg;
  if (cfg < 0) {
    const long long tilesL = (long long)agx::cdiv(M, 128) * agx::cdiv(N, 128);
    if (M <= 128 || ((long long)M * N <= (1 << 19) && K <= 2048))
      cfg = 3;  // skinny or small: latency-oriented direct kernel
    else if (tilesL < 2 * cus && K <= 512 && (long long)agx::cdiv(M, 64) * agx::cdiv(N, 64) >= cus)
      cfg = 1;  // short K, 128x128 tiles would leave a mostly empty second round: 4x as many 64x64 tiles
    else if (tilesL >= cus || (M > 64 && N > 64 && (long long)K * tilesL >= 40000)) {
      cfg = 0;
      // tile quantisation: 284 tiles on 256 CUs run as two rounds (55 % of the machine).  If the
      // 64x64 tiling fills its rounds clearly better, take it (18161x256: 1136 tiles, 89 %).
      const long long t64 = (long long)agx::cdiv(M, 64) * agx::cdiv(N, 64);
      const double e128 = (double)tilesL / (double)(agx::cdiv64(tilesL, cus) * cus);
      const double e64 = (double)t64 / (double)(agx::cdiv64(t64, cus) * cus);
      if (tilesL >= cus && e128 < 0.8 && e64 > e128 + 0.12) cfg = 1;
      // few output tiles over a long K (first critic layer at 512 rows: 1024 x 256 x 18161): the
      // 8-wave 128x128 workgroup, ONE per CU, measured 96 us against 105 for two 4-wave ones (and
      // 53 against 58 us at 512 x 256 x 18154); NN only (the layouts measured)
      if (layout == 0 && tilesL <= 16 && ktiles >= 128) cfg = 5;
    }
    else
      cfg = 1;
  }
  pl.cfg = cfg;
  pl.wtm = pl.wtn = 1;
  pl.bm = (cfg == 0 || cfg >= 4) ? 128 : (cfg == 1 ? 64 : 32);
  pl.bn = (cfg == 0 || cfg >= 4) ? 128 : (cfg == 1 ? 64 : (cfg == 2 ? 128 : 32));
  // (a 256x128 tile of 8 waves x 64x64 - a quarter fewer global loads per MFMA - measured 119-124 TF against 115-121 at
  //  4096^3 and 101 against 104 us on the 512-row first critic layer: not worth six more instantiations;
  //  profiles/r02_gemm_occupancy_and_ceiling.txt)
  // (160x128 / 128x160 tiles of 4 waves x five 32x32 blocks - 18161 = 113.5 x 160, so the 18k-gene dimension times 256
  //  or 512 makes 228 / 456 tiles = one round - measured 169.8 against 173.6 us on the 512-row dW_1 (one workgroup per
  //  CU has nothing to overlap with), 63.0 against 58.8 us on the input gradient, 347 against 327 us on the generator
  //  output layer; 230-256 registers with spills in four layouts: profiles/r02_tile160.txt)
  // (a 192x64 tile of 4 waves x three blocks - every row of the generator's stacked output layer 192 x 18154 x 256 in ONE
  //  tile, W_out streamed once, 284 workgroups - measured 35.9 against 29.8 us for the 64x64 tiling, which every other
  //  configuration here loses to as well (31.9 .. 60.9 us): with 8 K-tiles a workgroup is load-latency-bound and four
  //  resident 64x64 workgroups per CU overlap that where one large one cannot; profiles/r03_gen_out_tile_sweep.txt, cfg40 = that tile)
  if (cfg >= 20) {
    pl.bm = pl.bn = 128;
  } else if (cfg >= 8) {
    pl.wtm = kWaveTiles[cfg - 8][0];
    pl.wtn = kWaveTiles[cfg - 8][1];
    pl.bm = 32 * pl.wtm;
    pl.bn = 32 * pl.wtn;
  }
  pl.tiles = agx::cdiv(M, pl.bm) * agx::cdiv(N, pl.bn);
  int sk = force_splitk;
  if (sk <= 0) {
    sk = 1;
    if (cfg >= 8) {
      sk = 1;
    } else if (cfg == 3) {
      // keep the per-block K range around 128..512 (1-2 load rounds per wave) while the grid
      // stays within a few waves of workgroups
      // (one workgroup per CU measured best at the 18k-gene shapes: 64 x 256 x 18161 takes 13.8 us
      // with 16 slices vs 15.8 with 32; 32 x 256 x 18154 10.6 us with 32 slices vs 12.1 with 64)
      if (K > 512) {
        // (with the tail kernels reducing the slabs, 2 or 3 workgroups per CU still lose: the product
        // gains 4-7 us per iteration, the tails lose 10-35)
        sk = std::min(agx::cdiv(K, 128), agx::cdiv(cus, pl.tiles));
        if (sk > agx::cdiv(K, 128)) sk = agx::cdiv(K, 128);
      }
    } else {
      // workgroups that can be resident at once (LDS-limited): one round of blocks, no tail round
      const int slots = cus * ((cfg == 4 || (cfg == 5 && force_cfg < 0)) ? 1 : ((cfg == 0 || cfg == 5) ? 2 : 4));
      if (pl.tiles * 2 <= slots && ktiles >= 32) {
        sk = slots / pl.tiles;                 // fill the resident slots exactly once
        if (sk > ktiles / 4) sk = ktiles / 4;  // keep >= 4 K-tiles per slice
      }
    }
    if (sk > 128) sk = 128;
    if (sk < 1) sk = 1;
  }
  if (sk > ktiles) sk = ktiles > 0 ? ktiles : 1;
  const int64_t mn = (int64_t)M * N;
  if (sk > 1 && mn > 0 && (int64_t)sk * mn > ws_floats) sk = (int)(ws_floats / mn);
  if (sk < 1) sk = 1;
  int kt_per = agx::cdiv(ktiles > 0 ? ktiles : 1, sk);
  pl.kchunk = kt_per * BK;
  pl.splitk = K > 0 ? agx::cdiv(K, pl.kchunk) : 1;
  if (pl.splitk < 1) pl.splitk = 1;
  return pl;
}

}  // namespace

namespace agx {
// Raise the dynamic-LDS limit of every GEMM instantiation on the current device (called from
// agx_init, so that no attribute call happens inside a stream capture).
int gemm_prepare_device() {
  AGX_TRY((prepare_cfg<128, 128, 2, 2>()));
  AGX_TRY((prepare_cfg<64, 64, 2, 2, 16>()));
  AGX_TRY((prepare_cfg<32, 128, 1, 4>()));
  AGX_TRY((prepare_cfg<128, 128, 2, 4>()));
  AGX_TRY((prepare_cfg<128, 128, 2, 2, 16>()));
  AGX_TRY((prepare_cfg<128, 128, 2, 4, 16>()));
  AGX_TRY((prepare_cfg<128, 128, 2, 2, 64>()));
  AGX_TRY((prepare_one<128, 128, 2, 2, true, false, true, true>()));
  AGX_TRY((prepare_one<128, 128, 2, 2, true, false, false, true>()));
  // weight-gradient products with the optimizer in the epilogue (TN layout)
  AGX_TRY((prepare_one<128, 128, 2, 2, true, false, true, false, 32, true>()));
  AGX_TRY((prepare_one<128, 128, 2, 2, true, false, false, false, 32, true>()));
  AGX_TRY((prepare_one<64, 64, 2, 2, true, false, false, false, 32, true>()));
  AGX_TRY((prepare_one<64, 64, 2, 2, true, false, true, false, 16, true>()));
  AGX_TRY((prepare_one<64, 64, 2, 2, true, false, true, false, 32, true, true>()));
  AGX_TRY((prepare_one<64, 64, 2, 2, true, false, false, false, 32, true, true>()));
  // stream-K form of the plain 128x128 TN product
  AGX_TRY((prepare_one<128, 128, 2, 2, true, false, true, false, 32, false, false, true>()));
  return AGX_OK;
}
}  // namespace agx

extern "C" int64_t agx_gemm_ws_floats(int32_t M, int32_t N, int32_t K) {
  int sk = 1;
  for (int layout = -1; layout <= 0; ++layout) sk = std::max(sk, make_plan(M, N, K, -1, -1, INT64_MAX, layout).splitk);
  // (a TN product that may run stream-K keeps one 128x128 partial tile per workgroup there)
  const int64_t streamk = (int64_t)streamk_wgs(M, N, K, 2, -1, 0) * 128 * 128;
  return std::max<int64_t>(sk > 1 ? (int64_t)sk * M * N : 0, streamk);
}

extern "C" int64_t agx_gemm_sk_flag_words(void) { return 2 * (int64_t)agx::num_cus() + 16; }

extern "C" int agx_gemm_f32(void *stream, const agx_gemm_t *g) { return agx::gemm_run(stream, g, nullptr, nullptr); }

bool agx::gemm_can_dual(int M, int N, int K, int64_t ws_floats, int layout) {
  const Plan pl = make_plan(M, N, K, -1, 0, ws_floats, layout);
  return (pl.cfg == 0 || pl.cfg == 1 || pl.cfg == 2 || pl.cfg == 5) && pl.splitk == 1;
}

bool agx::gemm_can_fuse_opt(int M, int N, int K, int64_t ws_floats) {
  const Plan pl = make_plan(M, N, K, -1, 0, ws_floats, 2);
  return (pl.cfg == 0 || pl.cfg == 1) && pl.splitk == 1;
}

int agx::gemm_run(void *stream, const agx_gemm_t *g, const agx::FusedOpt *fo, int *raw_slabs, const agx::DualOut *dual) {
  AGX_CHECK_ARG(g != nullptr, "agx_gemm_f32: null descriptor");
  AGX_CHECK_ARG(g->M >= 0 && g->N >= 0 && g->K >= 0, "agx_gemm_f32: negative dimension");
  if (g->M == 0 || g->N == 0) return AGX_OK;
  AGX_CHECK_ARG(g->A && g->B && g->C, "agx_gemm_f32: null operand");
  AGX_CHECK_ARG(!(g->ta && g->tb), "agx_gemm_f32: ta=tb=1 is not a layout this path needs");
  AGX_CHECK_ARG(g->lda >= (g->ta ? g->M : g->K), "agx_gemm_f32: lda %d too small", g->lda);
  AGX_CHECK_ARG(g->ldb >= (g->tb ? g->K : g->N), "agx_gemm_f32: ldb %d too small", g->ldb);
  AGX_CHECK_ARG(g->ldc >= g->N, "agx_gemm_f32: ldc %d < N %d", g->ldc, g->N);
  AGX_CHECK_ARG(!g->addrow || (g->addmod > 0 && g->ldadd >= g->N), "agx_gemm_f32: bad addrow");
  AGX_CHECK_ARG(!g->mask || g->ldmask >= g->N, "agx_gemm_f32: bad mask ld");
  AGX_CHECK_ARG(g->force_cfg <= 23, "agx_gemm_f32: force_cfg out of range");

  KParams p{};
  p.A = g->A; p.B = g->B; p.C = g->C;
  p.M = g->M; p.N = g->N; p.K = g->K;
  p.lda = g->lda; p.ldb = g->ldb; p.ldc = g->ldc;
  p.alpha = g->alpha;
  p.bias = g->bias; p.addrow = g->addrow; p.addmod = g->addmod; p.ldadd = g->ldadd;
  p.act = g->act; p.slope = g->slope; p.mask = g->mask; p.ldmask = g->ldmask;
  p.accumulate = g->accumulate;
  p.rowscale = g->rowscale;
  AGX_CHECK_ARG((g->col_scale == nullptr) == (g->col_shift == nullptr), "agx_gemm_f32: col_scale and col_shift go together");
  p.col_scale = g->col_scale; p.col_shift = g->col_shift;
  p.rowsq = g->rowsq;
  p.rowsq_ld = g->rowsq_ld;
  p.ws = g->ws;
  AGX_CHECK_ARG(!g->rowsq || g->rowsq_ld >= agx::cdiv(g->N, 32), "agx_gemm_f32: rowsq_ld too small");
  // the row sum-of-squares side output is produced by the main kernel: no split-K with it
  const bool opt = fo && fo->rows > 0;
  if (opt) {
    AGX_CHECK_ARG(fo->w && fo->s1 && g->ta && !g->tb && !g->rowsq && g->interp_rows == 0 && !g->bias && !g->addrow &&
                      !g->act && !g->mask && !g->accumulate && !g->rowscale && !g->col_scale,
                  "agx_gemm_f32: the fused optimizer takes a plain TN weight-gradient product");
    p.opt_w = fo->w; p.opt_s1 = fo->s1;
    p.opt_rows = fo->rows < g->M ? fo->rows : g->M;
    p.opt_lr = fo->lr; p.opt_rho = fo->rho; p.opt_eps = fo->eps;
    if (fo->rest_n > 0) {
      AGX_CHECK_ARG(fo->rest_w && fo->rest_g && fo->rest_s1 && fo->rest_hole_lo % 4 == 0 && fo->rest_hole_hi % 4 == 0 &&
                        fo->rest_hole_lo >= 0 && fo->rest_hole_lo <= fo->rest_hole_hi &&
                        fo->rest_hole_hi <= (fo->rest_n & ~(int64_t)3),
                    "agx_gemm_f32: bad flat-update range of the fused optimizer");
      p.rest_w = fo->rest_w; p.rest_g = fo->rest_g; p.rest_s1 = fo->rest_s1;
      p.rest_n = fo->rest_n;
      p.rest_hole_lo4 = fo->rest_hole_lo / 4;
      p.rest_hole_len4 = (fo->rest_hole_hi - fo->rest_hole_lo) / 4;
      const int64_t rest4 = (fo->rest_n >> 2) - p.rest_hole_len4;
      p.opt_rest_blocks = (int)std::min<int64_t>(256, std::max<int64_t>(1, agx::cdiv64(rest4 + 256, 256)));
      if (fo->nx_cov || fo->nx_dst) {
        int64_t emb_end = 0;
        for (int i = 0; i < fo->nx_emb.n_cat; ++i)
          emb_end = std::max<int64_t>(emb_end, fo->nx_emb.table_off[i] + (int64_t)fo->nx_emb.vocab[i] * fo->nx_emb.edim[i]);
        AGX_CHECK_ARG(fo->nx_src && fo->nx_dst && fo->nx_cov && fo->nx_rows > 0 && fo->nx_G > 0 && fo->nx_lds >= fo->nx_G &&
                          fo->nx_ldd >= fo->nx_G && fo->nx_lds % 4 == 0 && fo->nx_ldd % 4 == 0 && agx::aligned16(fo->nx_src) &&
                          agx::aligned16(fo->nx_dst) && fo->nx_B > 0 && fo->nx_cov_rows >= 0 && fo->nx_emb.n_cat >= 0 &&
                          fo->nx_emb.n_cat <= AGX_MAX_CAT && (fo->nx_emb.n_cat == 0 || fo->nx_cat) &&
                          (fo->nx_nnum == 0 || fo->nx_num) && emb_end <= 1024 && emb_end <= fo->rest_hole_lo &&
                          true,
                      "agx_gemm_f32: bad next-input description of the fused optimizer");
        p.nx_src = fo->nx_src; p.nx_dst = fo->nx_dst; p.nx_cov = fo->nx_cov;
        p.nx_lds = fo->nx_lds; p.nx_ldd = fo->nx_ldd; p.nx_rows = fo->nx_rows; p.nx_G = fo->nx_G;
        p.nx_cov_rows = fo->nx_cov_rows; p.nx_B = fo->nx_B; p.nx_nnum = fo->nx_nnum; p.nx_ldcat = fo->nx_ldcat;
        p.nx_num = fo->nx_num; p.nx_cat = fo->nx_cat; p.nx_emb = fo->nx_emb;
        AGX_CHECK_ARG(!fo->nx_mask || (fo->nx_noise && fo->nx_ldn >= 0), "agx_gemm_f32: augmentation noise of the next input missing");
        p.nx_mask = fo->nx_mask; p.nx_noise = fo->nx_noise; p.nx_ldn = fo->nx_ldn;
        p.opt_copy_blocks = (int)std::min<int64_t>(128, std::max<int64_t>(1, agx::cdiv64((int64_t)fo->nx_rows * (fo->nx_G / 4 + 1), 4 * 256)));
      }
    }
  }
  const bool vec = agx::aligned16(g->A) && agx::aligned16(g->B) && (g->lda % 4 == 0) && (g->ldb % 4 == 0);
  // stream-K: plain or fused-optimizer TN products of callers that bring the hand-off words (zeroed once)
  // (not with the fused optimizer: the finisher's update round trips are exposed - 191-200 against 169 us per launch)
  const bool allow_sk = !opt && g->sk_flags && g->sk_flag_words >= agx_gemm_sk_flag_words() && vec && g->ws && agx::aligned16(g->ws) &&
                        !raw_slabs && !(dual && dual->rows > 0) && !g->rowsq && g->interp_rows == 0 && !g->bias && !g->addrow &&
                        !g->act && !g->mask && !g->accumulate && !g->rowscale && !g->col_scale;
  Plan pl = make_plan(g->M, g->N, g->K, g->force_cfg, g->rowsq ? 1 : g->force_splitk, g->ws ? g->ws_floats : 0,
                      g->ta ? 2 : (g->tb ? 1 : 0), allow_sk);
  // A fused-optimizer product over a short K (batch < 128) is a memory-bound launch whose epilogue is its long pole:
  // the 64x64 instantiation (one 32x32 block per wave, operands prefetched, clamp-free interior path) also serves the
  // shapes whose tile count would choose 128x128 (wide critic, 20007 x 1024 x 96: 123 -> 9x us per launch).  No
  // split-K either way, so every element keeps the k order of the unfused launch.
  // (128x128 tiles there: wide critic 17.0 k instead of 20.3 k samples/s, profiles/r02_ab_wide_fused_tiles.txt)
  if (opt && pl.cfg == 0 && !pl.sk_wgs && pl.splitk == 1 && g->K <= 512 && g->force_cfg < 0)
    pl = make_plan(g->M, g->N, g->K, 1, 1, g->ws ? g->ws_floats : 0, 2);
  // the fused epilogue exists for the LDS-tiled kernels without split-K; callers ask
  // gemm_can_fuse_opt first, so the product keeps the summation order of the unfused launch
  AGX_CHECK_ARG(!opt || ((pl.cfg == 0 || pl.cfg == 1) && pl.splitk == 1),
                "agx_gemm_f32: no fused-optimizer kernel for this shape (cfg %d, split-K %d)", pl.cfg, pl.splitk);
  p.splitk = pl.splitk;
  p.kchunk = pl.kchunk;
  p.to_ws = pl.splitk > 1;
  if (dual && dual->rows > 0) {
    // (callers ask gemm_can_dual first)
    AGX_CHECK_ARG(dual->C && dual->ldc >= g->N && !opt && !raw_slabs && g->interp_rows == 0 && !g->rowscale && !g->addrow &&
                      !g->mask && !g->accumulate && !g->rowsq && !g->col_scale && pl.splitk == 1 &&
                      (pl.cfg == 0 || pl.cfg == 1 || pl.cfg == 2 || pl.cfg == 5) && g->force_cfg < 0,
                  "agx_gemm_f32: dual output needs a plain Dense product on the tiled kernel (cfg %d, split-K %d)", pl.cfg, pl.splitk);
    p.c2 = dual->C; p.ldc2 = dual->ldc; p.c2_rows = dual->rows;
  }
  if (raw_slabs) {
    // the caller's own kernel finishes the product: leave the split-K slabs ws[z][M][N] unreduced
    AGX_CHECK_ARG(!opt && !g->rowsq && g->interp_rows == 0, "agx_gemm_f32: raw slabs combine with nothing");
    if (!g->ws || g->ws_floats < (int64_t)pl.splitk * g->M * g->N) {
      agx::set_error("agx_gemm_f32: raw slabs need a workspace of splitk*M*N floats");
      return AGX_E_WORKSPACE;
    }
    p.to_ws = 1;
    p.ws_by_row = 1;
    *raw_slabs = pl.splitk;
  }
  if (g->interp_rows > 0) {
    // [real; fake] -> [real; fake; interpolated]: always finished by the reduce kernel
    AGX_CHECK_ARG(g->M == 2 * g->interp_rows && g->interp_alpha, "agx_gemm_f32: interp needs M == 2*interp_rows and alpha");
    AGX_CHECK_ARG(!g->rowsq && !g->rowscale && !g->mask && !g->accumulate && !g->addrow,
                  "agx_gemm_f32: interp combines only with alpha/bias/act");
    if (!g->ws || g->ws_floats < (int64_t)g->M * g->N) {
      agx::set_error("agx_gemm_f32: interp needs a workspace of at least M*N floats");
      return AGX_E_WORKSPACE;
    }
    p.interp_rows = g->interp_rows;
    p.interp_alpha = g->interp_alpha;
    p.to_ws = 1;
  }
  hipStream_t st = agx::as_stream(stream);
  // algorithmic bytes: operands + result; a fused-optimizer product reads and writes weight and
  // slot (16 B) per updated element instead of storing the gradient (4 B): 4(MK + KN) + 16 MN
  agx::ProfScope prof(st, g->prof_tagged ? g->prof_id : AGX_K_GEMM_UNTAGGED,
                      2.0 * g->M * (double)g->N * g->K,
                      4.0 * ((double)g->M * g->K + (double)g->K * g->N + (double)g->M * g->N) +
                          (opt ? 12.0 * (double)p.opt_rows * g->N : 0.0));
  // (the next step's row copy and the flat update of the remaining parameters that such a launch may carry are
  //  NOT counted: SURVEY 8(d) prices the product and its optimizer traffic only; bench.py lists them apart)
  p.opt_tiles = pl.tiles;
  int rc;
  if (pl.sk_wgs) {
    p.sk_part = g->ws; p.sk_flags = g->sk_flags; p.sk_wgs = pl.sk_wgs; p.sk_nxcd = 8;
    p.opt_tiles = pl.sk_wgs;
    rc = launch<128, 128, 2, 2, true, false, true, false, 32, false, false, true>(st, p, pl.sk_wgs);
  } else if (opt && pl.cfg == 0)
    rc = vec ? launch<128, 128, 2, 2, true, false, true, false, 32, true>(st, p, pl.tiles)
             : launch<128, 128, 2, 2, true, false, false, false, 32, true>(st, p, pl.tiles);
  else if (opt && g->K <= 512)   // short K (batch < 128): the epilogue's instruction stream is on the critical path
    rc = vec ? launch<64, 64, 2, 2, true, false, true, false, 32, true, true>(st, p, pl.tiles)
             : launch<64, 64, 2, 2, true, false, false, false, 32, true, true>(st, p, pl.tiles);
  else if (opt && vec)
    // long K (the 512-row dW_1): K-tiles of 16; same k order, same bits: 2.679 against 2.703 ms per 512-row
    // iteration (112 registers: 4 workgroups per CU instead of 3; profiles/r03_fused_dw1_residency.txt; at K <= 512
    // it measured 0.5 % slower)
    rc = launch<64, 64, 2, 2, true, false, true, false, 16, true>(st, p, pl.tiles);
  else if (opt)
    rc = launch<64, 64, 2, 2, true, false, false, false, 32, true>(st, p, pl.tiles);
  else if (pl.cfg == 0)
    rc = launch_layout<128, 128, 2, 2>(st, p, pl.tiles, g->ta, g->tb, vec);
  else if (pl.cfg == 1)
    // K-tiles of 16 (16 KB of LDS per workgroup): same k order, same bits as K-tiles of 32, and 1-2.5 % faster on every
    // 64x64-tiled product of the iteration (profiles/r03_fused_dw1_residency.txt).  Not a residency effect: these
    // kernels hold 144 registers (128 + 16 accumulators), i.e. 3 workgroups per CU either way.
    rc = launch_layout<64, 64, 2, 2, 16>(st, p, pl.tiles, g->ta, g->tb, vec);
  else if (pl.cfg == 2)
    rc = launch_layout<32, 128, 1, 4>(st, p, pl.tiles, g->ta, g->tb, vec);
  else if (pl.cfg == 5)
    rc = launch_layout<128, 128, 2, 4>(st, p, pl.tiles, g->ta, g->tb, vec);
  else if (pl.cfg == 6)
    rc = launch_layout<128, 128, 2, 2, 16>(st, p, pl.tiles, g->ta, g->tb, vec);
  else if (pl.cfg == 4)
    rc = launch_layout<128, 128, 2, 2, 64>(st, p, pl.tiles, g->ta, g->tb, vec);
  else if (pl.cfg == 7)
    rc = launch_layout<128, 128, 2, 4, 16>(st, p, pl.tiles, g->ta, g->tb, vec);
  else if (pl.cfg >= 20) {  // timing ablations of the TN 128x128 kernel (tools/ only; results are wrong)
    AGX_CHECK_ARG(g->ta && !g->tb && vec, "agx_gemm_f32: ablation configs exist for aligned TN only");
    dim3 grid(pl.tiles, p.splitk, 1), block(256, 1, 1);
    constexpr size_t lds = 2 * (size_t)(128 * 32 + 32 * 128) * sizeof(float);
    if (pl.cfg == 21) hipLaunchKernelGGL((gemm_kernel<128, 128, 2, 2, true, false, true, false, 32, 1>), grid, block, lds, st, p);
    else if (pl.cfg == 22) hipLaunchKernelGGL((gemm_kernel<128, 128, 2, 2, true, false, true, false, 32, 2>), grid, block, lds, st, p);
    else if (pl.cfg == 23) hipLaunchKernelGGL((gemm_kernel<128, 128, 2, 2, true, false, true, false, 32, 3>), grid, block, lds, st, p);
    else hipLaunchKernelGGL((gemm_kernel<128, 128, 2, 2, true, false, true, false, 32, 0>), grid, block, lds, st, p);
    rc = AGX_OK;
  } else if (pl.cfg >= 8) {
    if (!g->ta && !g->tb) rc = launch_wave<false, false>(st, p, vec, pl.wtm, pl.wtn);
    else if (!g->ta && g->tb) rc = launch_wave<false, true>(st, p, vec, pl.wtm, pl.wtn);
    else rc = launch_wave<true, false>(st, p, vec, pl.wtm, pl.wtn);
  } else {
    const int gpw = agx::cdiv(agx::cdiv(std::min(p.kchunk, p.K), 8), 4);  // 8-wide k groups per wave
    if (!g->ta && !g->tb) rc = launch_direct<false, false>(st, p, pl.tiles, vec, gpw);
    else if (!g->ta && g->tb) rc = launch_direct<false, true>(st, p, pl.tiles, vec, gpw);
    else rc = launch_direct<true, false>(st, p, pl.tiles, vec, gpw);
  }
  if (rc != AGX_OK) return rc;
  if (raw_slabs) return AGX_OK;
  if (p.interp_rows > 0) return launch_reduce<true>(st, p);
  if (p.to_ws) return launch_reduce<false>(st, p);
  return AGX_OK;
}

// measurement only: resident workgroups per CU the runtime computes for the main tile configurations
// which: 0 = 64x64 NT, 1 = 64x64 TN fused optimizer (long K: K-tiles of 16), 2 = 128x128 NN 8-wave, 3 = 128x128 SYRK, 4 = 64x64 NN
extern "C" int agx_prof_occupancy(int32_t which, int32_t *blocks_per_cu, int32_t *lds_bytes) {
  AGX_CHECK_ARG(blocks_per_cu && lds_bytes, "agx_prof_occupancy: null argument");
  int nb = 0;
  size_t lds = 0;
  hipError_t e = hipErrorInvalidValue;
  constexpr size_t l64 = 2 * (size_t)(64 * 32 + 64 * 32) * sizeof(float), l128 = 2 * (size_t)(128 * 32 + 128 * 32) * sizeof(float);
  switch (which) {
    case 0: lds = l64 / 2; e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, gemm_kernel<64, 64, 2, 2, false, true, true, false, 16, 0, false, false>, 256, lds); break;
    case 1: lds = l64 / 2; e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, gemm_kernel<64, 64, 2, 2, true, false, true, false, 16, 0, true, false>, 256, lds); break;
    case 2: lds = l128; e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, gemm_kernel<128, 128, 2, 4, false, false, true, false, 32, 0, false, false>, 512, lds); break;
    case 3: lds = l128; e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, gemm_kernel<128, 128, 2, 2, true, false, true, true, 32, 0, false, false>, 256, lds); break;
    case 4: lds = l64 / 2; e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, gemm_kernel<64, 64, 2, 2, false, false, true, false, 16, 0, false, false>, 256, lds); break;
    case 5: lds = l128; e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, gemm_kernel<128, 128, 2, 2, false, false, true, false, 32, 0, false, false>, 256, lds); break;
    default: AGX_CHECK_ARG(false, "agx_prof_occupancy: which = %d", which);
  }
  AGX_HIP(e);
  *blocks_per_cu = nb;
  *lds_bytes = (int32_t)lds;
  return AGX_OK;
}

extern "C" int64_t agx_corr_num_tiles(int32_t G) {
  const int64_t T = agx::cdiv(G, 128);
  return T * (T + 1) / 2;
}

extern "C" int agx_corr_xtx(void *stream, const float *xs, int32_t ldx, int32_t N, int32_t G,
                            float alpha, float *C, int32_t ldc, int64_t tile_begin,
                            int64_t tile_end) {
  AGX_CHECK_ARG(xs && C, "agx_corr_xtx: null operand");
  AGX_CHECK_ARG(N >= 0 && G > 0 && ldx >= G && ldc >= G, "agx_corr_xtx: bad shape N=%d G=%d ldx=%d ldc=%d", N, G, ldx, ldc);
  const int64_t total = agx_corr_num_tiles(G);
  if (tile_end < 0 || tile_end > total) tile_end = total;
  AGX_CHECK_ARG(tile_begin >= 0 && tile_begin <= tile_end, "agx_corr_xtx: bad tile range");
  if (tile_begin == tile_end) return AGX_OK;
  KParams p{};
  p.A = xs; p.B = xs; p.C = C;
  p.M = G; p.N = G; p.K = N;
  p.lda = ldx; p.ldb = ldx; p.ldc = ldc;
  p.alpha = alpha;
  p.splitk = 1;
  p.kchunk = agx::cdiv(N > 0 ? N : 1, BK) * BK;
  p.tile_begin = tile_begin;
  p.tiles_1d = agx::cdiv(G, 128);
  const bool vec = agx::aligned16(xs) && (ldx % 4 == 0);
  hipStream_t st = agx::as_stream(stream);
  const int tiles = (int)(tile_end - tile_begin);
  // algorithmic work of the computed tiles: symmetric count N*G*(G+1) scaled by the tile share
  const double share = (double)tiles / (double)total;
  agx::ProfScope prof(st, AGX_K_CORR_XTX, share * (double)N * G * ((double)G + 1.0),
                      4.0 * share * ((double)N * G + (double)G * G));
  // (an 8-wave 2x4 workgroup and a BK = 64 variant measured 118.9 and 110.2 TFLOP/s against 119.9 at 18154 x 100000)
  return vec ? launch<128, 128, 2, 2, true, false, true, true>(st, p, tiles)
             : launch<128, 128, 2, 2, true, false, false, true>(st, p, tiles);
}

#ifdef AGX_STAMPS
extern "C" int agx_dbg_stamps(long long *buf, int m, int k) {
  AGX_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &buf, sizeof(buf)));
  AGX_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_m), &m, sizeof(m)));
  AGX_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_k), &k, sizeof(k)));
  return AGX_OK;
}
#endif
