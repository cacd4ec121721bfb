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

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BK = 32;

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
  float *ws;
  int splitk, kchunk;
  // SYRK only
  long long tile_begin;
  int tiles_1d;
};

__device__ __forceinline__ void epilogue_store(const KParams &p, int row, int col, float v) {
  v *= p.alpha;
  if (p.bias) v += p.bias[col];
  if (p.addrow) v += p.addrow[(size_t)(row % p.addmod) * p.ldadd + col];
  if (p.act) v = v > 0.f ? v : v * p.slope;
  if (p.mask) {
    float mk = p.mask[(size_t)row * p.ldmask + col];
    v = mk > 0.f ? v : v * p.slope;
  }
  float *c = p.C + (size_t)row * p.ldc + col;
  if (p.accumulate) v += *c;
  *c = v;
}

// ---- operand tile: ROWS x BK --------------------------------------------
template <int ROWS, bool KC, int NT, bool VEC>
struct Tile {
  static constexpr int F4 = ROWS * BK / 4;
  static constexpr int PT = F4 / NT;  // float4 per thread
  static_assert(F4 % NT == 0, "tile must divide over the block");
  static constexpr int LDS_FLOATS = KC ? ROWS * (BK + 1) : BK * ROWS;

  // (r, k) of the first element of float4 number q
  __device__ static __forceinline__ void coords(int q, int &r, int &k) {
    if (KC) {
      r = q / (BK / 4);
      k = (q % (BK / 4)) * 4;
    } else {
      k = q / (ROWS / 4);
      r = (q % (ROWS / 4)) * 4;
    }
  }

  __device__ static __forceinline__ void load(const float *__restrict__ src, int ld, int r0, int R,
                                              int k0, int kend, int tid, float4 (&reg)[PT]) {
#pragma unroll
    for (int i = 0; i < PT; ++i) {
      int r, k;
      coords(tid + i * NT, r, k);
      const int gr = r0 + r, gk = k0 + k;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (gr < R && gk < kend) {
        if (KC) {
          const float *s = src + (size_t)gr * ld + gk;
          if (VEC) {
            v = *reinterpret_cast<const float4 *>(s);
            if (gk + 1 >= kend) v.y = 0.f;
            if (gk + 2 >= kend) v.z = 0.f;
            if (gk + 3 >= kend) v.w = 0.f;
          } else {
            v.x = s[0];
            if (gk + 1 < kend) v.y = s[1];
            if (gk + 2 < kend) v.z = s[2];
            if (gk + 3 < kend) v.w = s[3];
          }
        } else {
          const float *s = src + (size_t)gk * ld + gr;
          if (VEC) {
            v = *reinterpret_cast<const float4 *>(s);
            if (gr + 1 >= R) v.y = 0.f;
            if (gr + 2 >= R) v.z = 0.f;
            if (gr + 3 >= R) v.w = 0.f;
          } else {
            v.x = s[0];
            if (gr + 1 < R) v.y = s[1];
            if (gr + 2 < R) v.z = s[2];
            if (gr + 3 < R) v.w = s[3];
          }
        }
      }
      reg[i] = v;
    }
  }

  __device__ static __forceinline__ void store(float *lds, int tid, const float4 (&reg)[PT]) {
#pragma unroll
    for (int i = 0; i < PT; ++i) {
      int r, k;
      coords(tid + i * NT, r, k);
      if (KC) {
        float *d = lds + r * (BK + 1) + k;  // odd stride: 4 scalar stores, conflict free
        d[0] = reg[i].x;
        d[1] = reg[i].y;
        d[2] = reg[i].z;
        d[3] = reg[i].w;
      } else {
        *reinterpret_cast<float4 *>(lds + k * ROWS + r) = reg[i];
      }
    }
  }

  __device__ static __forceinline__ float frag(const float *lds, int r, int k) {
    return KC ? lds[r * (BK + 1) + k] : lds[k * ROWS + r];
  }
};

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

template <int BM, int BN, int WM, int WN, bool TA, bool TB, bool VEC, bool SYRK>
__global__ void __launch_bounds__(WM *WN * 64) gemm_kernel(const KParams p) {
  constexpr int NT = WM * WN * 64;
  constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
  static_assert(TM >= 1 && TN >= 1, "wave tile");
  using TileA = Tile<BM, !TA, NT, VEC>;
  using TileB = Tile<BN, TB, NT, VEC>;
  constexpr int STAGE = TileA::LDS_FLOATS + TileB::LDS_FLOATS;

  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int l31 = lane & 31, lh = lane >> 5;

  int tm, tn;
  if (SYRK) {
    tri_decode(p.tile_begin + blockIdx.x, p.tiles_1d, tm, tn);
  } else {
    const int tiles_n = (p.N + BN - 1) / BN;
    tm = blockIdx.x / tiles_n;
    tn = blockIdx.x % tiles_n;
  }
  const int m0 = tm * BM, n0 = tn * BN;
  const int z = blockIdx.y;
  const int kbeg = z * p.kchunk;
  const int kend = min(p.K, kbeg + p.kchunk);
  const int nt = (kend - kbeg + BK - 1) / BK;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float4 ra[TileA::PT], rb[TileB::PT];
  if (nt > 0) {
    TileA::load(p.A, p.lda, m0, p.M, kbeg, kend, tid, ra);
    TileB::load(p.B, p.ldb, n0, p.N, kbeg, kend, tid, rb);
    TileA::store(smem, tid, ra);
    TileB::store(smem + TileA::LDS_FLOATS, tid, rb);
  }
  __syncthreads();

  for (int t = 0; t < nt; ++t) {
    const float *cA = smem + (t & 1) * STAGE;
    const float *cB = cA + TileA::LDS_FLOATS;
    const bool more = (t + 1 < nt);
    if (more) {
      const int k0 = kbeg + (t + 1) * BK;
      TileA::load(p.A, p.lda, m0, p.M, k0, kend, tid, ra);
      TileB::load(p.B, p.ldb, n0, p.N, k0, kend, tid, rb);
    }
#pragma unroll
    for (int kk = 0; kk < BK / 2; ++kk) {
      const int k = 2 * kk + lh;
      float a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = TileA::frag(cA, (wm * TM + i) * 32 + l31, k);
#pragma unroll
      for (int j = 0; j < TN; ++j) b[j] = TileB::frag(cB, (wn * TN + j) * 32 + l31, k);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (more) {
      float *nA = smem + ((t + 1) & 1) * STAGE;
      TileA::store(nA, tid, ra);
      TileB::store(nA + TileA::LDS_FLOATS, tid, rb);
    }
    __syncthreads();
  }

  // ---- epilogue: C/D map of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
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
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = rbase + (r & 3) + 8 * (r >> 2);
          if (row < p.M && col < p.N) {
            if (p.splitk > 1)
              p.ws[((size_t)z * p.M + row) * p.N + col] = acc[i][j][r];
            else
              epilogue_store(p, row, col, acc[i][j][r]);
          }
        }
      }
    }
  }
}

__global__ void __launch_bounds__(256) splitk_reduce_kernel(const KParams p) {
  const size_t mn = (size_t)p.M * p.N;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < mn;
       idx += (size_t)gridDim.x * blockDim.x) {
    float v = 0.f;
    for (int s = 0; s < p.splitk; ++s) v += p.ws[(size_t)s * mn + idx];
    epilogue_store(p, (int)(idx / p.N), (int)(idx % p.N), v);
  }
}

template <int BM, int BN, int WM, int WN, bool TA, bool TB, bool VEC, bool SYRK>
int launch(hipStream_t st, const KParams &p, int tiles) {
  using TileA = Tile<BM, !TA, WM * WN * 64, VEC>;
  using TileB = Tile<BN, TB, WM * WN * 64, VEC>;
  constexpr size_t lds = 2 * (size_t)(TileA::LDS_FLOATS + TileB::LDS_FLOATS) * sizeof(float);
  auto kern = gemm_kernel<BM, BN, WM, WN, TA, TB, VEC, SYRK>;
  static thread_local int attr_device = -2;
  int dev = -1;
  AGX_HIP(hipGetDevice(&dev));
  if (attr_device != dev) {
    AGX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_device = dev;
  }
  dim3 grid(tiles, p.splitk, 1), block(WM * WN * 64, 1, 1);
  hipLaunchKernelGGL(kern, grid, block, lds, st, p);
  AGX_LAUNCH_CHECK();
  return AGX_OK;
}

template <int BM, int BN, int WM, int WN>
int launch_layout(hipStream_t st, const KParams &p, int tiles, int ta, int tb, bool vec) {
  if (!ta && !tb)
    return vec ? launch<BM, BN, WM, WN, false, false, true, false>(st, p, tiles)
               : launch<BM, BN, WM, WN, false, false, false, false>(st, p, tiles);
  if (!ta && tb)
    return vec ? launch<BM, BN, WM, WN, false, true, true, false>(st, p, tiles)
               : launch<BM, BN, WM, WN, false, true, false, false>(st, p, tiles);
  return vec ? launch<BM, BN, WM, WN, true, false, true, false>(st, p, tiles)
             : launch<BM, BN, WM, WN, true, false, false, false>(st, p, tiles);
}

struct Plan {
  int cfg, bm, bn, tiles, splitk, kchunk;
};

Plan make_plan(int M, int N, int K, int force_cfg, int force_splitk, int64_t ws_floats) {
  Plan pl;
  const int cus = agx::num_cus();
  int cfg = force_cfg;
  if (cfg < 0) {
    const long long tilesL = (long long)agx::cdiv(M, 128) * agx::cdiv(N, 128);
    if (M <= 32)
      cfg = 2;
    else if (tilesL >= cus || (M > 64 && N > 64 && (long long)K * tilesL >= 40000))
      cfg = 0;
    else
      cfg = 1;
  }
  pl.cfg = cfg;
  pl.bm = cfg == 0 ? 128 : (cfg == 1 ? 64 : 32);
  pl.bn = cfg == 0 ? 128 : (cfg == 1 ? 64 : 128);
  pl.tiles = agx::cdiv(M, pl.bm) * agx::cdiv(N, pl.bn);
  const int ktiles = agx::cdiv(K, BK);
  int sk = force_splitk;
  if (sk <= 0) {
    sk = 1;
    if (pl.tiles < (cus * 3) / 4 && ktiles >= 8) {
      sk = agx::cdiv(2 * cus, pl.tiles);
      if (sk > ktiles / 4) sk = ktiles / 4;  // keep >= 4 K-tiles per slice
      if (sk > 128) sk = 128;
      if (sk < 1) sk = 1;
    }
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

extern "C" int64_t agx_gemm_ws_floats(int32_t M, int32_t N, int32_t K) {
  Plan pl = make_plan(M, N, K, -1, -1, INT64_MAX);
  return pl.splitk > 1 ? (int64_t)pl.splitk * M * N : 0;
}

extern "C" int agx_gemm_f32(void *stream, const agx_gemm_t *g) {
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
  AGX_CHECK_ARG(g->force_cfg <= 2, "agx_gemm_f32: force_cfg out of range");

  KParams p{};
  p.A = g->A; p.B = g->B; p.C = g->C;
  p.M = g->M; p.N = g->N; p.K = g->K;
  p.lda = g->lda; p.ldb = g->ldb; p.ldc = g->ldc;
  p.alpha = g->alpha;
  p.bias = g->bias; p.addrow = g->addrow; p.addmod = g->addmod; p.ldadd = g->ldadd;
  p.act = g->act; p.slope = g->slope; p.mask = g->mask; p.ldmask = g->ldmask;
  p.accumulate = g->accumulate;
  p.ws = g->ws;
  Plan pl = make_plan(g->M, g->N, g->K, g->force_cfg, g->force_splitk, g->ws ? g->ws_floats : 0);
  p.splitk = pl.splitk;
  p.kchunk = pl.kchunk;
  const bool vec = agx::aligned16(g->A) && agx::aligned16(g->B) && (g->lda % 4 == 0) && (g->ldb % 4 == 0);
  hipStream_t st = agx::as_stream(stream);
  agx::ProfScope prof(st, g->prof_tagged ? g->prof_id : AGX_K_GEMM_UNTAGGED,
                      2.0 * g->M * (double)g->N * g->K,
                      4.0 * ((double)g->M * g->K + (double)g->K * g->N + (double)g->M * g->N));
  int rc;
  if (pl.cfg == 0)
    rc = launch_layout<128, 128, 2, 2>(st, p, pl.tiles, g->ta, g->tb, vec);
  else if (pl.cfg == 1)
    rc = launch_layout<64, 64, 2, 2>(st, p, pl.tiles, g->ta, g->tb, vec);
  else
    rc = launch_layout<32, 128, 1, 4>(st, p, pl.tiles, g->ta, g->tb, vec);
  if (rc != AGX_OK) return rc;
  if (p.splitk > 1) {
    const int64_t mn = (int64_t)p.M * p.N;
    int blocks = (int)std::min<int64_t>(agx::cdiv64(mn, 256), 4096);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, st, p);
    AGX_LAUNCH_CHECK();
  }
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
  return vec ? launch<128, 128, 2, 2, true, false, true, true>(st, p, tiles)
             : launch<128, 128, 2, 2, true, false, false, true>(st, p, tiles);
}
