// HBM-bound kernels of the correlation / gamma evaluation path
// (reference utils.py:260-268, 808-873): column statistics, standardisation,
// condensed upper-triangle packing with the reference's NaN-dropping, and the
// five-sum reduction behind the final 1-D Pearson.  The G x G contraction
// itself is agx_corr_xtx in gemm.hip.
#include "agx_common.h"

#include <algorithm>
#include <mutex>

namespace {

using agx::wave_sum;

// Column statistics in two deterministic stages (no atomics: the result does not depend on the order
// in which workgroups finish).  Stage 1: blockDim 256 = 64 columns x 4 row phases; grid (ceil(G/64),
// row chunks); every chunk leaves its fp64 partial sums of (x - shift) and (x - shift)^2 per column in
// ws[chunk][2][G], shift = the column's first row (sums about a value inside the data keep the
// variance well conditioned when |mean| >> std: raw log-expression values).  Stage 2 adds the chunks
// in index order.
template <typename T>
__global__ void __launch_bounds__(256) colstats_partial_kernel(const T *__restrict__ x, int ldx,
                                                               int N, int G, int rows_per_chunk,
                                                               double *__restrict__ ws) {
  __shared__ double red[2][4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + tx;
  const int rbeg = blockIdx.y * rows_per_chunk;
  const int rend = min(N, rbeg + rows_per_chunk);
  double s = 0.0, q = 0.0;
  if (c < G) {
    const double shift = (double)x[c];
    for (int r = rbeg + ty; r < rend; r += 4) {
      const double v = (double)x[(size_t)r * ldx + c] - shift;
      s += v;
      q += v * v;
    }
  }
  red[0][ty][tx] = s;
  red[1][ty][tx] = q;
  __syncthreads();
  if (ty == 0 && c < G) {
    s = red[0][0][tx] + red[0][1][tx] + red[0][2][tx] + red[0][3][tx];
    q = red[1][0][tx] + red[1][1][tx] + red[1][2][tx] + red[1][3][tx];
    double *w = ws + (size_t)blockIdx.y * 2 * G;
    w[c] = s;
    w[G + c] = q;
  }
}

template <typename T>
__global__ void __launch_bounds__(256) colstats_finish_kernel(const double *__restrict__ ws, int chunks,
                                                              const T *__restrict__ x, int N, int G,
                                                              T *__restrict__ mean,
                                                              T *__restrict__ stdv) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= G) return;
  double s = 0.0, q = 0.0;
  for (int k = 0; k < chunks; ++k) {
    s += ws[(size_t)k * 2 * G + c];
    q += ws[(size_t)k * 2 * G + G + c];
  }
  const double d = s / (double)N;          // mean - shift
  double var = q / (double)N - d * d;
  if (var < 0.0) var = 0.0;
  mean[c] = (T)((double)x[c] + d);
  stdv[c] = (T)sqrt(var);
}

template <typename T>
__global__ void __launch_bounds__(256) standardize_kernel(const T *__restrict__ x, int ldx,
                                                          int G, const T *__restrict__ mean,
                                                          const T *__restrict__ stdv,
                                                          T *__restrict__ out, int ldo) {
  const int r = blockIdx.y;
  const T *xr = x + (size_t)r * ldx;
  T *orow = out + (size_t)r * ldo;
  for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < G; c += gridDim.x * blockDim.x)
    orow[c] = (xr[c] - mean[c]) / stdv[c];  // x/0 -> inf/NaN exactly as NumPy
}

// ---- condensed upper triangle with NaN dropping ---------------------------
template <typename T>
__global__ void __launch_bounds__(256) triu_count_kernel(const T *__restrict__ C, int ldc, int G,
                                                         long long *__restrict__ counts) {
  __shared__ int red[4];
  const int i = blockIdx.x;
  const T *row = C + (size_t)i * ldc;
  int n = 0;
  for (int j = i + 1 + threadIdx.x; j < G; j += blockDim.x) n += !isnan(row[j]);
  for (int o = 32; o > 0; o >>= 1) n += __shfl_xor(n, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = n;
  __syncthreads();
  if (threadIdx.x == 0) counts[i] = red[0] + red[1] + red[2] + red[3];
}

// single block exclusive scan of counts[0..G) -> offsets[0..G]; offsets[G] = total
__global__ void __launch_bounds__(1024) scan_kernel(const long long *__restrict__ counts, int G,
                                                    long long *__restrict__ offsets,
                                                    long long *__restrict__ total) {
  __shared__ long long part[1024];
  const int t = threadIdx.x;
  const int per = (G + 1023) / 1024;
  const int beg = t * per, end = min(G, beg + per);
  long long s = 0;
  for (int i = beg; i < end; ++i) s += counts[i];
  part[t] = s;
  __syncthreads();
  if (t == 0) {
    long long run = 0;
    for (int k = 0; k < 1024; ++k) {
      const long long v = part[k];
      part[k] = run;
      run += v;
    }
    offsets[G] = run;
    if (total) *total = run;
  }
  __syncthreads();
  long long run = part[t];
  for (int i = beg; i < end; ++i) {
    offsets[i] = run;
    run += counts[i];
  }
}

template <typename T>
__global__ void __launch_bounds__(256) triu_write_kernel(const T *__restrict__ C, int ldc, int G,
                                                         const long long *__restrict__ offsets,
                                                         T *__restrict__ out) {
  __shared__ int wtot[4];
  const int i = blockIdx.x;
  const T *row = C + (size_t)i * ldc;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  long long base = offsets[i];
  for (int j0 = i + 1; j0 < G; j0 += 256) {
    const int j = j0 + threadIdx.x;
    T v = 0;
    bool keep = false;
    if (j < G) {
      v = row[j];
      keep = !isnan(v);
    }
    const unsigned long long bal = __ballot(keep);
    const int before = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) wtot[wave] = __popcll(bal);
    __syncthreads();
    int wbase = 0;
    for (int w = 0; w < wave; ++w) wbase += wtot[w];
    const int tot = wtot[0] + wtot[1] + wtot[2] + wtot[3];
    if (keep) out[base + wbase + before] = v;
    base += tot;
    __syncthreads();
  }
}

// ---- five sums for the 1-D Pearson ----------------------------------------
// Deterministic: every block leaves its partial sums in part[block][6]; one finishing block adds the
// blocks' partials in index order (fixed tree) to sums[] - no atomics, so gamma is bit-reproducible.
__device__ __forceinline__ void block_add5(double (&s)[6], int nvals, double *__restrict__ part) {
  __shared__ double red[6][4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    if (k < nvals) {
      const double v = wave_sum(s[k]);
      if (lane == 0) red[k][wave] = v;
    }
  }
  __syncthreads();
  if (threadIdx.x < 6) {
    const int k = threadIdx.x;
    part[(size_t)blockIdx.x * 6 + k] = k < nvals ? red[k][0] + red[k][1] + red[k][2] + red[k][3] : 0.0;
  }
}

__global__ void __launch_bounds__(256) sums_finish_kernel(const double *__restrict__ part, int nblocks,
                                                          int nvals, double *__restrict__ sums) {
  __shared__ double red[6][4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double s[6] = {0, 0, 0, 0, 0, 0};
  for (int b = threadIdx.x; b < nblocks; b += 256)
#pragma unroll
    for (int k = 0; k < 6; ++k) s[k] += part[(size_t)b * 6 + k];
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    const double v = wave_sum(s[k]);
    if (lane == 0) red[k][wave] = v;
  }
  __syncthreads();
  if (threadIdx.x < nvals) {
    const int k = threadIdx.x;
    sums[k] += red[k][0] + red[k][1] + red[k][2] + red[k][3];
  }
}

template <typename T>
__global__ void __launch_bounds__(256) pearson_sums_kernel(const T *__restrict__ a,
                                                           const T *__restrict__ b, long long n,
                                                           int one_minus,
                                                           double *__restrict__ part) {
  double s[6] = {0, 0, 0, 0, 0, 0};
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x) {
    T fa = a[i], fb = b[i];
    if (one_minus) {
      fa = (T)1 - fa;  // utils.py:870-871 (subtraction in the input's own precision)
      fb = (T)1 - fb;
    }
    const double da = fa, db = fb;
    s[0] += da; s[1] += db; s[2] += da * da; s[3] += db * db; s[4] += da * db;
  }
  block_add5(s, 5, part);
}

template <typename T>
__global__ void __launch_bounds__(256) gamma_sums_kernel(const T *__restrict__ C1, int ld1,
                                                         const T *__restrict__ C2, int ld2,
                                                         int G, double *__restrict__ part) {
  double s[6] = {0, 0, 0, 0, 0, 0};
  for (int i = blockIdx.x; i < G; i += gridDim.x) {
    const T *r1 = C1 + (size_t)i * ld1;
    const T *r2 = C2 + (size_t)i * ld2;
    for (int j = i + 1 + threadIdx.x; j < G; j += blockDim.x) {
      const T fa = (T)1 - r1[j], fb = (T)1 - r2[j];
      if (isnan(fa) || isnan(fb)) {
        s[5] += 1.0;
      } else {
        const double da = fa, db = fb;
        s[0] += da; s[1] += db; s[2] += da * da; s[3] += db * db; s[4] += da * db;
      }
    }
  }
  block_add5(s, 6, part);
}

// same sums over a range of the upper-triangular 128x128 tile list of agx_corr_xtx (the tiles one
// rank owns when the G x G matrix is dealt across GPUs): one block per tile
__global__ void __launch_bounds__(256) gamma_sums_tiles_kernel(const float *__restrict__ C1, int ld1,
                                                               const float *__restrict__ C2, int ld2,
                                                               int G, long long tile_begin, long long tile_end,
                                                               int tiles_1d, double *__restrict__ part) {
  const int tx = threadIdx.x & 127, ty = threadIdx.x >> 7;
  double s[6] = {0, 0, 0, 0, 0, 0};
  for (long long t = tile_begin + blockIdx.x; t < tile_end; t += gridDim.x) {
    int ti, tj;
    agx::tri_decode(t, tiles_1d, ti, tj);
    const int i0 = ti * 128, j0 = tj * 128;
    const int j = j0 + tx;
    if (j >= G) continue;
    for (int r = ty; r < 128; r += 2) {
      const int i = i0 + r;
      if (i >= G || j <= i) continue;
      const float fa = 1.0f - C1[(size_t)i * ld1 + j], fb = 1.0f - C2[(size_t)i * ld2 + j];
      if (isnan(fa) || isnan(fb)) {
        s[5] += 1.0;
      } else {
        const double da = fa, db = fb;
        s[0] += da; s[1] += db; s[2] += da * da; s[3] += db * db; s[4] += da * db;
      }
    }
  }
  block_add5(s, 6, part);
}

// Scratch of the allocation-free entry point (agx_colstats_noalloc): one copy per GPU, held until the
// process ends.  In host code this name is only a stand-in for the symbol: the device address comes
// from hipGetSymbolAddress, once per device (scratch_for_current_device).
static __device__ __align__(256) double scratch[AGX_COLSTATS_NOALLOC_DOUBLES];

constexpr int kMaxDevices = 64;
std::mutex g_scratch_mu;
double *g_scratch_ptr[kMaxDevices] = {};

int scratch_for_current_device(double **out) {
  int dev = -1;
  AGX_HIP(hipGetDevice(&dev));
  AGX_CHECK_ARG(dev >= 0 && dev < kMaxDevices, "agx_colstats_noalloc: device index %d unsupported", dev);
  std::lock_guard<std::mutex> lock(g_scratch_mu);
  if (!g_scratch_ptr[dev]) {
    double *p = nullptr;
    AGX_HIP(hipGetSymbolAddress((void **)&p, HIP_SYMBOL(scratch)));
    AGX_CHECK_ARG(p != nullptr, "agx_colstats_noalloc: hipGetSymbolAddress returned a null address");
    g_scratch_ptr[dev] = p;
  }
  *out = g_scratch_ptr[dev];
  return AGX_OK;
}

// row chunks of the first stage: enough workgroups to fill the part, at least 64 rows each
int colstats_chunks(int N, int G) {
  const int gx = agx::cdiv(G, 64);
  int chunks = std::max(1, std::min(agx::cdiv(N, 64), agx::cdiv(4 * agx::num_cus(), gx)));
  const int rpc = agx::cdiv(N, chunks);
  return agx::cdiv(N, rpc);
}

template <typename T>
int colstats_run(hipStream_t st, const T *x, int ldx, int N, int G, T *mean, T *std, double *ws,
                 int chunks) {
  const int gx = agx::cdiv(G, 64);
  const int rpc = agx::cdiv(N, chunks);
  chunks = agx::cdiv(N, rpc);
  hipLaunchKernelGGL(colstats_partial_kernel<T>, dim3(gx, chunks), dim3(256), 0, st, x, ldx, N, G, rpc, ws);
  AGX_LAUNCH_CHECK();
  hipLaunchKernelGGL(colstats_finish_kernel<T>, dim3(agx::cdiv(G, 256)), dim3(256), 0, st, ws, chunks, x, N, G,
                     mean, std);
  AGX_LAUNCH_CHECK();
  return AGX_OK;
}

int sums_finish(hipStream_t st, const double *part, int nblocks, int nvals, double *sums) {
  hipLaunchKernelGGL(sums_finish_kernel, dim3(1), dim3(256), 0, st, part, nblocks, nvals, sums);
  AGX_LAUNCH_CHECK();
  return AGX_OK;
}

}  // namespace

// agx_init resolves the scratch address of the device it initialises, so that the first agx_colstats_noalloc call
// on that device makes no runtime query (a caller may already be capturing a graph)
int agx::colstats_scratch_init() {
  double *p = nullptr;
  return scratch_for_current_device(&p);
}

extern "C" int64_t agx_colstats_ws_doubles(int32_t N, int32_t G) {
  if (N <= 0 || G <= 0) return 0;
  return (int64_t)2 * G * colstats_chunks(N, G);
}

extern "C" int agx_colstats(void *stream, const float *x, int32_t ldx, int32_t N, int32_t G,
                            float *mean, float *std, double *ws, int64_t ws_doubles) {
  AGX_CHECK_ARG(x && mean && std && ws, "agx_colstats: null argument");
  AGX_CHECK_ARG(N > 0 && G > 0 && ldx >= G, "agx_colstats: bad shape N=%d G=%d ldx=%d", N, G, ldx);
  const int chunks = colstats_chunks(N, G);
  if (ws_doubles < (int64_t)2 * G * chunks) {
    agx::set_error("agx_colstats: workspace of %lld doubles < agx_colstats_ws_doubles(%d, %d) = %lld",
                   (long long)ws_doubles, N, G, (long long)2 * G * chunks);
    return AGX_E_WORKSPACE;
  }
  return colstats_run(agx::as_stream(stream), x, ldx, N, G, mean, std, ws, chunks);
}

extern "C" int agx_colstats_noalloc(void *stream, const float *x, int32_t ldx, int32_t N, int32_t G,
                                    float *mean, float *std) {
  AGX_CHECK_ARG(x && mean && std, "agx_colstats_noalloc: null argument");
  AGX_CHECK_ARG(N > 0 && G > 0 && ldx >= G, "agx_colstats_noalloc: bad shape N=%d G=%d ldx=%d", N, G, ldx);
  const int chunks = colstats_chunks(N, G);
  if (G > AGX_COLSTATS_NOALLOC_MAX_G || (int64_t)2 * G * chunks > AGX_COLSTATS_NOALLOC_DOUBLES) {
    agx::set_error("agx_colstats_noalloc: G=%d exceeds the static scratch (max G = %d, %d doubles per GPU); "
                   "use agx_colstats with a caller workspace", G, AGX_COLSTATS_NOALLOC_MAX_G,
                   AGX_COLSTATS_NOALLOC_DOUBLES);
    return AGX_E_INVALID;
  }
  double *ws = nullptr;
  AGX_TRY(scratch_for_current_device(&ws));
  // every partial slot the second stage reads is written by the first: nothing has to start at zero
  return colstats_run(agx::as_stream(stream), x, ldx, N, G, mean, std, ws, chunks);
}

extern "C" int agx_standardize(void *stream, const float *x, int32_t ldx, int32_t N, int32_t G,
                               const float *mean, const float *std, float *out, int32_t ldo) {
  AGX_CHECK_ARG(x && mean && std && out, "agx_standardize: null argument");
  AGX_CHECK_ARG(N >= 0 && G > 0 && ldx >= G && ldo >= G, "agx_standardize: bad shape");
  if (N == 0) return AGX_OK;
  AGX_CHECK_ARG(N <= 65535 * 16, "agx_standardize: N too large for one launch");
  hipStream_t st = agx::as_stream(stream);
  for (int r0 = 0; r0 < N; r0 += 65535) {  // gridDim.y limit
    const int rows = std::min(65535, N - r0);
    hipLaunchKernelGGL(standardize_kernel<float>, dim3(std::min(agx::cdiv(G, 256), 32), rows), dim3(256),
                       0, st, x + (size_t)r0 * ldx, ldx, G, mean, std, out + (size_t)r0 * ldo, ldo);
    AGX_LAUNCH_CHECK();
  }
  return AGX_OK;
}

namespace {
template <typename T>
int triu_pack_impl(void *stream, const T *C, int32_t ldc, int32_t G, T *out, int64_t *row_counts,
                   int64_t *row_offsets, int64_t *n_out_dev) {
  AGX_CHECK_ARG(C && out && row_counts && row_offsets, "agx_triu_pack: null argument");
  AGX_CHECK_ARG(G > 0 && ldc >= G, "agx_triu_pack: bad shape");
  hipStream_t st = agx::as_stream(stream);
  hipLaunchKernelGGL(triu_count_kernel<T>, dim3(G), dim3(256), 0, st, C, ldc, G,
                     reinterpret_cast<long long *>(row_counts));
  AGX_LAUNCH_CHECK();
  hipLaunchKernelGGL(scan_kernel, dim3(1), dim3(1024), 0, st,
                     reinterpret_cast<const long long *>(row_counts), G,
                     reinterpret_cast<long long *>(row_offsets),
                     reinterpret_cast<long long *>(n_out_dev));
  AGX_LAUNCH_CHECK();
  hipLaunchKernelGGL(triu_write_kernel<T>, dim3(G), dim3(256), 0, st, C, ldc, G,
                     reinterpret_cast<const long long *>(row_offsets), out);
  AGX_LAUNCH_CHECK();
  return AGX_OK;
}
}  // namespace

extern "C" int agx_triu_pack(void *stream, const float *C, int32_t ldc, int32_t G, float *out,
                             int64_t *row_counts, int64_t *row_offsets, int64_t *n_out_dev) {
  return triu_pack_impl<float>(stream, C, ldc, G, out, row_counts, row_offsets, n_out_dev);
}

extern "C" int agx_triu_pack_f64(void *stream, const double *C, int32_t ldc, int32_t G, double *out,
                                 int64_t *row_counts, int64_t *row_offsets, int64_t *n_out_dev) {
  return triu_pack_impl<double>(stream, C, ldc, G, out, row_counts, row_offsets, n_out_dev);
}

extern "C" int64_t agx_sums_ws_doubles(void) { return 6 * 2048; }

extern "C" int agx_pearson_sums(void *stream, const float *a, const float *b, int64_t n,
                                int32_t one_minus, double *sums, double *ws) {
  AGX_CHECK_ARG(a && b && sums && ws && n >= 0, "agx_pearson_sums: bad argument");
  if (n == 0) return AGX_OK;
  int blocks = (int)std::min<int64_t>(agx::cdiv64(n, 256 * 8), 2048);
  if (blocks < 1) blocks = 1;
  hipStream_t st = agx::as_stream(stream);
  hipLaunchKernelGGL(pearson_sums_kernel<float>, dim3(blocks), dim3(256), 0, st, a, b, (long long)n, one_minus, ws);
  AGX_LAUNCH_CHECK();
  return sums_finish(st, ws, blocks, 5, sums);
}

extern "C" int agx_gamma_sums(void *stream, const float *C1, int32_t ld1, const float *C2,
                              int32_t ld2, int32_t G, double *sums, double *ws) {
  AGX_CHECK_ARG(C1 && C2 && sums && ws && G > 0 && ld1 >= G && ld2 >= G, "agx_gamma_sums: bad argument");
  hipStream_t st = agx::as_stream(stream);
  const int blocks = std::min(G, 2048);
  hipLaunchKernelGGL(gamma_sums_kernel<float>, dim3(blocks), dim3(256), 0, st, C1, ld1, C2, ld2, G, ws);
  AGX_LAUNCH_CHECK();
  return sums_finish(st, ws, blocks, 6, sums);
}

extern "C" int agx_gamma_sums_tiles(void *stream, const float *C1, int32_t ld1, const float *C2,
                                    int32_t ld2, int32_t G, int64_t tile_begin, int64_t tile_end,
                                    double *sums, double *ws) {
  AGX_CHECK_ARG(C1 && C2 && sums && ws && G > 0 && ld1 >= G && ld2 >= G, "agx_gamma_sums_tiles: bad argument");
  const int64_t total = agx_corr_num_tiles(G);
  if (tile_end < 0 || tile_end > total) tile_end = total;
  AGX_CHECK_ARG(tile_begin >= 0 && tile_begin <= tile_end, "agx_gamma_sums_tiles: bad tile range");
  if (tile_begin == tile_end) return AGX_OK;
  hipStream_t st = agx::as_stream(stream);
  const int blocks = (int)std::min<int64_t>(tile_end - tile_begin, 2048);
  hipLaunchKernelGGL(gamma_sums_tiles_kernel, dim3(blocks), dim3(256), 0, st, C1, ld1, C2, ld2, G,
                     (long long)tile_begin, (long long)tile_end, agx::cdiv(G, 128), ws);
  AGX_LAUNCH_CHECK();
  return sums_finish(st, ws, blocks, 6, sums);
}


// ---------------------------------------------------------------------------
// float64 inputs: the reference computes in the input's dtype (utils.py:816-824: np.mean / np.std / np.dot of
// float64 arrays, the E. coli notebook path), so float64 matrices take a float64 path end to end: the same
// two-stage statistics, and X^T Y / N on the fp64 matrix cores (v_mfma_f64_16x16x4_f64).  One wave computes a
// 32 x 32 block of C (2 x 2 MFMA tiles) over all N samples, fragments straight from global memory: at the
// shapes float64 callers have (N <= 1000, G <= 4297: 25 GFLOP) the kernel is L2-bound at a few milliseconds.
// ---------------------------------------------------------------------------
namespace {
typedef double f64x4 __attribute__((ext_vector_type(4)));

// C[G1, G2] = alpha * X^T Y;  X [N, G1] ldx, Y [N, G2] ldy (row = sample)
__global__ void __launch_bounds__(256) xty_f64_kernel(const double *__restrict__ X, int ldx,
                                                      const double *__restrict__ Y, int ldy, int N, int G1,
                                                      int G2, double alpha, double *__restrict__ C, int ldc) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int tiles_n = (G2 + 31) / 32;
  const long long t = (long long)blockIdx.x * 4 + wave;
  const int i0 = (int)(t / tiles_n) * 32, j0 = (int)(t % tiles_n) * 32;
  if (i0 >= G1) return;
  const int c16 = lane & 15, k4 = lane >> 4;     // operand element: column c16 of the 16-wide tile, sample k4 of the 4
  f64x4 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = (f64x4){0.0, 0.0, 0.0, 0.0};
  int ci[2], cj[2];
  bool vi[2], vj[2];
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    ci[a] = i0 + 16 * a + c16; vi[a] = ci[a] < G1; ci[a] = min(ci[a], G1 - 1);
    cj[a] = j0 + 16 * a + c16; vj[a] = cj[a] < G2; cj[a] = min(cj[a], G2 - 1);
  }
  for (int n0 = 0; n0 < N; n0 += 4) {
    const int n = n0 + k4;
    const bool vn = n < N;
    const size_t r = (size_t)min(n, N - 1);
    double xa[2], yb[2];
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const double xv = X[r * ldx + ci[a]], yv = Y[r * ldy + cj[a]];
      xa[a] = (vn && vi[a]) ? xv : 0.0;
      yb[a] = (vn && vj[a]) ? yv : 0.0;
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[a], yb[b], acc[a][b], 0, 0, 0);
  }
  // C/D map of the fp64 16x16 tile: column = lane & 15, register e holds row 4*e + (lane >> 4)
  // (unlike the fp32 tiles, whose registers hold four CONSECUTIVE rows)
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int col = j0 + 16 * b + c16;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int row = i0 + 16 * a + 4 * e + k4;
        if (row < G1 && col < G2) C[(size_t)row * ldc + col] = acc[a][b][e] * alpha;
      }
    }
}
}  // namespace

extern "C" int agx_colstats_f64(void *stream, const double *x, int32_t ldx, int32_t N, int32_t G, double *mean,
                                double *std, double *ws, int64_t ws_doubles) {
  AGX_CHECK_ARG(x && mean && std && ws, "agx_colstats_f64: null argument");
  AGX_CHECK_ARG(N > 0 && G > 0 && ldx >= G, "agx_colstats_f64: bad shape N=%d G=%d ldx=%d", N, G, ldx);
  const int chunks = colstats_chunks(N, G);
  if (ws_doubles < (int64_t)2 * G * chunks) {
    agx::set_error("agx_colstats_f64: workspace of %lld doubles < agx_colstats_ws_doubles(%d, %d)", (long long)ws_doubles, N, G);
    return AGX_E_WORKSPACE;
  }
  return colstats_run<double>(agx::as_stream(stream), x, ldx, N, G, mean, std, ws, chunks);
}

extern "C" int agx_standardize_f64(void *stream, const double *x, int32_t ldx, int32_t N, int32_t G,
                                   const double *mean, const double *std, double *out, int32_t ldo) {
  AGX_CHECK_ARG(x && mean && std && out, "agx_standardize_f64: null argument");
  AGX_CHECK_ARG(N >= 0 && G > 0 && ldx >= G && ldo >= G, "agx_standardize_f64: bad shape");
  hipStream_t st = agx::as_stream(stream);
  for (int r0 = 0; r0 < N; r0 += 65535) {
    const int rows = std::min(65535, N - r0);
    hipLaunchKernelGGL(standardize_kernel<double>, dim3(std::min(agx::cdiv(G, 256), 32), rows), dim3(256), 0, st,
                       x + (size_t)r0 * ldx, ldx, G, mean, std, out + (size_t)r0 * ldo, ldo);
    AGX_LAUNCH_CHECK();
  }
  return AGX_OK;
}

extern "C" int agx_corr_xty_f64(void *stream, const double *xs, int32_t ldx, const double *ys, int32_t ldy, int32_t N,
                                int32_t G1, int32_t G2, double alpha, double *C, int32_t ldc) {
  AGX_CHECK_ARG(xs && ys && C && N > 0 && G1 > 0 && G2 > 0 && ldx >= G1 && ldy >= G2 && ldc >= G2,
                "agx_corr_xty_f64: bad argument");
  const long long tiles = (long long)agx::cdiv(G1, 32) * agx::cdiv(G2, 32);
  hipLaunchKernelGGL(xty_f64_kernel, dim3((unsigned)agx::cdiv64(tiles, 4)), dim3(256), 0, agx::as_stream(stream), xs, ldx,
                     ys, ldy, N, G1, G2, alpha, C, ldc);
  AGX_LAUNCH_CHECK();
  return AGX_OK;
}

extern "C" int agx_pearson_sums_f64(void *stream, const double *a, const double *b, int64_t n, int32_t one_minus,
                                    double *sums, double *ws) {
  AGX_CHECK_ARG(a && b && sums && ws && n >= 0, "agx_pearson_sums_f64: bad argument");
  if (n == 0) return AGX_OK;
  int blocks = (int)std::min<int64_t>(agx::cdiv64(n, 256 * 8), 2048);
  if (blocks < 1) blocks = 1;
  hipStream_t st = agx::as_stream(stream);
  hipLaunchKernelGGL(pearson_sums_kernel<double>, dim3(blocks), dim3(256), 0, st, a, b, (long long)n, one_minus, ws);
  AGX_LAUNCH_CHECK();
  return sums_finish(st, ws, blocks, 5, sums);
}

extern "C" int agx_gamma_sums_f64(void *stream, const double *C1, int32_t ld1, const double *C2, int32_t ld2, int32_t G,
                                  double *sums, double *ws) {
  AGX_CHECK_ARG(C1 && C2 && sums && ws && G > 0 && ld1 >= G && ld2 >= G, "agx_gamma_sums_f64: bad argument");
  hipStream_t st = agx::as_stream(stream);
  const int blocks = std::min(G, 2048);
  hipLaunchKernelGGL(gamma_sums_kernel<double>, dim3(blocks), dim3(256), 0, st, C1, ld1, C2, ld2, G, ws);
  AGX_LAUNCH_CHECK();
  return sums_finish(st, ws, blocks, 6, sums);
}
