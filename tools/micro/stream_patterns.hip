// Micro-benchmark: how fast does one pass over W_1 (18154 x 256 fp32 = 18.6 MB, row pitch 1 KB) arrive in registers
// for the access patterns of the skinny-GEMM "direct" kernels?  Same launch shape as the input-gradient product of a
// 32-row critic step: 568 workgroups x 512 threads, each workgroup owns 32 matrix rows (32 KB), each wave 32 rows x
// 32 floats.  Every wave issues all its loads, then reduces them (nothing else).
//   pattern 0: MFMA-fragment order of the NT product: lane (r = lane & 31, h = lane >> 5) loads float4 at row r,
//              floats 8g + 4h .. +3, g = 0..3: one instruction = 32 rows x 32 B
//   pattern 1: line order: one instruction = 8 rows x 128 B (lane -> row lane >> 3, float4 lane & 7), 4 instructions
//   pattern 2: the same bytes as a flat float4 stream (wave w of the grid reads 4 KB contiguous)
//   pattern 3: fragment order of the NN product over the transposed view: dword loads, lane & 31 -> column (128 B
//              contiguous per half wave), 8 rows per lane... (one instruction = 2 rows x 128 B)
//   hipcc --offload-arch=gfx950 -O3 stream_patterns.hip -o stream_patterns && ./stream_patterns
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int ROWS = 18176, LD = 256;   // 568 x 32 rows

template <int P>
__global__ void __launch_bounds__(512) k(const float *__restrict__ w, float *out) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const float *base = w + (size_t)blockIdx.x * 32 * LD + wave * 32;   // 32 rows x 32 floats of this wave
  float s = 0.f;
  if (P == 0) {
    const int r = lane & 31, h = lane >> 5;
    float4 v[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) v[g] = *reinterpret_cast<const float4 *>(base + (size_t)r * LD + 8 * g + 4 * h);
#pragma unroll
    for (int g = 0; g < 4; ++g) s += v[g].x + v[g].y + v[g].z + v[g].w;
  } else if (P == 1) {
    float4 v[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) v[g] = *reinterpret_cast<const float4 *>(base + (size_t)(8 * g + (lane >> 3)) * LD + 4 * (lane & 7));
#pragma unroll
    for (int g = 0; g < 4; ++g) s += v[g].x + v[g].y + v[g].z + v[g].w;
  } else if (P == 2) {
    const float *b = w + ((size_t)blockIdx.x * 8 + wave) * 1024;
    float4 v[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) v[g] = *reinterpret_cast<const float4 *>(b + g * 256 + 4 * lane);
#pragma unroll
    for (int g = 0; g < 4; ++g) s += v[g].x + v[g].y + v[g].z + v[g].w;
  } else {
    // this wave: rows 32 blockIdx .. +31 are the "k" of an NN product, its 32 columns the "n": lane -> column,
    // half -> k parity; 16 dword loads per lane
    const int c = lane & 31, h = lane >> 5;
    float v[16];
#pragma unroll
    for (int g = 0; g < 16; ++g) v[g] = base[(size_t)(2 * g + h) * LD + c];
#pragma unroll
    for (int g = 0; g < 16; ++g) s += v[g];
  }
  if (s == 12345.678f) out[threadIdx.x] = s;
}

template <int P>
float run(const float *w, float *out, int reps) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k<P>, dim3(ROWS / 32), dim3(512), 0, 0, w, out);
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k<P>, dim3(ROWS / 32), dim3(512), 0, 0, w, out);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / reps * 1e3f;
}

__global__ void empty_k(float *out) { if (out == nullptr) out[0] = 1.f; }

int main() {
  float *w, *out, *other;
  const size_t n = (size_t)ROWS * LD;
  CK(hipMalloc(&w, n * 4)); CK(hipMalloc(&out, 4096)); CK(hipMalloc(&other, (size_t)512 << 20));
  CK(hipMemset(w, 0, n * 4));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0));
  for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(empty_k, dim3(ROWS / 32), dim3(512), 0, 0, out);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const float t_empty = ms / 200 * 1e3f;
  printf("empty launch of the same shape: %.2f us\n", t_empty);
  const float t[4] = {run<0>(w, out, 200), run<1>(w, out, 200), run<2>(w, out, 200), run<3>(w, out, 200)};
  const char *name[4] = {"NT fragment order (32 rows x 32 B per instruction)", "line order (8 rows x 128 B)", "flat float4 stream",
                         "NN fragment order (dword, 2 rows x 128 B)"};
  for (int p = 0; p < 4; ++p)
    printf("pattern %d %-52s %6.2f us  (%.2f TB/s; minus empty launch: %.2f TB/s)\n", p, name[p], t[p], n * 4 / t[p] / 1e6,
           n * 4 / (t[p] - t_empty) / 1e6);
  // the same with the matrix evicted between launches (512 MB written in between), single launches timed
  for (int p = 0; p < 4; ++p) {
    float tot = 0.f;
    for (int i = 0; i < 10; ++i) {
      CK(hipMemsetAsync(other, i, (size_t)512 << 20));
      CK(hipEventRecord(e0));
      if (p == 0) hipLaunchKernelGGL(k<0>, dim3(ROWS / 32), dim3(512), 0, 0, w, out);
      if (p == 1) hipLaunchKernelGGL(k<1>, dim3(ROWS / 32), dim3(512), 0, 0, w, out);
      if (p == 2) hipLaunchKernelGGL(k<2>, dim3(ROWS / 32), dim3(512), 0, 0, w, out);
      if (p == 3) hipLaunchKernelGGL(k<3>, dim3(ROWS / 32), dim3(512), 0, 0, w, out);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms, e0, e1));
      tot += ms;
    }
    printf("pattern %d after 512 MB of other traffic: %6.2f us per launch (event bracket included)\n", p, tot / 10 * 1e3f);
  }
  return 0;
}
