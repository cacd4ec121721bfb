// Micro-benchmark: what the 32-row input-gradient product g = D_1 . W_1^T (32 x 18154 x 256, NT) costs beyond streaming
// W_1, built up part by part in the shape of gemm_direct_kernel<NT, U=4, 8 waves>: 568 workgroups x 512 threads, one
// 32-gene tile each, wave w takes k in [32w, 32w+32).
//   part 0: B (W_1 rows) loads only, fragment order         part 1: B in line order
//   part 2: part 0 + A loads (the same 32 x 256 matrix for every workgroup: L2 hits)
//   part 3: part 2 + the 16 MFMAs                            part 4: part 3 + cross-wave LDS reduction
//   part 5: part 4 + the 32 x 32 tile stored                 part 6: part 5 with B in line order through LDS
//   hipcc --offload-arch=gfx950 -O3 dx_parts.hip -o dx_parts && ./dx_parts
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int ROWS = 18176, LD = 256;

template <int P>
__global__ void __launch_bounds__(512) k(const float *__restrict__ w, const float *__restrict__ a, float *__restrict__ c,
                                         float *out) {
  __shared__ float red[8][32][33];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, i = lane & 31, h = lane >> 5;
  const float *bb = w + (size_t)blockIdx.x * 32 * LD + wave * 32;
  const float *ab = a + wave * 32;
  float4 bv[4], av[4];
  constexpr bool LINE = P == 1 || P == 6;
  if (!LINE) {
#pragma unroll
    for (int g = 0; g < 4; ++g) bv[g] = *reinterpret_cast<const float4 *>(bb + (size_t)i * LD + 8 * g + 4 * h);
  } else {
#pragma unroll
    for (int g = 0; g < 4; ++g) bv[g] = *reinterpret_cast<const float4 *>(bb + (size_t)(8 * g + (lane >> 3)) * LD + 4 * (lane & 7));
  }
  if (P >= 2) {
#pragma unroll
    for (int g = 0; g < 4; ++g) av[g] = *reinterpret_cast<const float4 *>(ab + (size_t)i * LD + 8 * g + 4 * h);
  }
  if (P == 6) {   // line order -> fragment order through this wave's LDS block ([k-quad][row] float4)
    float4 *st = reinterpret_cast<float4 *>(&red[0][0][0]) + wave * 256;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int r = 8 * g + (lane >> 3), q = lane & 7;
      st[q * 32 + (r ^ (q & 7))] = bv[g];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int q = 2 * g + h;
      bv[g] = st[q * 32 + (i ^ (q & 7))];
    }
    __syncthreads();
  }
  if (P < 3) {
    float s = 0.f;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      s += bv[g].x + bv[g].y + bv[g].z + bv[g].w;
      if (P >= 2) s += av[g].x + av[g].y + av[g].z + av[g].w;
    }
    if (s == 12345.678f) out[threadIdx.x] = s;
    return;
  }
  f32x16 acc = {};
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[g].x, bv[g].x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[g].y, bv[g].y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[g].z, bv[g].z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[g].w, bv[g].w, acc, 0, 0, 0);
  }
  if (P == 3) {
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[r];
    if (s == 12345.678f) out[threadIdx.x] = s;
    return;
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) red[wave][(r & 3) + 8 * (r >> 2) + 4 * h][i] = acc[r];
  __syncthreads();
  const int cc = threadIdx.x & 31;
  float v[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int rr = (threadIdx.x >> 5) + 16 * q;
    v[q] = red[0][rr][cc] + red[1][rr][cc] + red[2][rr][cc] + red[3][rr][cc] + red[4][rr][cc] + red[5][rr][cc] + red[6][rr][cc] + red[7][rr][cc];
  }
  if (P == 4) {
    if (v[0] + v[1] == 12345.678f) out[threadIdx.x] = v[0];
    return;
  }
#pragma unroll
  for (int q = 0; q < 2; ++q) c[(size_t)((threadIdx.x >> 5) + 16 * q) * ROWS + blockIdx.x * 32 + cc] = v[q];
}

template <int P>
float run(const float *w, const float *a, float *c, float *out, int reps) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k<P>, dim3(ROWS / 32), dim3(512), 0, 0, w, a, c, out);
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k<P>, dim3(ROWS / 32), dim3(512), 0, 0, w, a, c, out);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / reps * 1e3f;
}

int main() {
  float *w, *a, *c, *out;
  const size_t n = (size_t)ROWS * LD;
  CK(hipMalloc(&w, n * 4)); CK(hipMalloc(&a, 32 * LD * 4)); CK(hipMalloc(&c, (size_t)32 * ROWS * 4)); CK(hipMalloc(&out, 4096));
  CK(hipMemset(w, 0, n * 4)); CK(hipMemset(a, 0, 32 * LD * 4));
  const char *name[7] = {"B loads, fragment order", "B loads, line order", "+ A loads (shared 32 x 256)", "+ 16 MFMAs", "+ LDS reduction over the 8 waves",
                         "+ tile stored", "all parts, B in line order through LDS"};
  const float t[7] = {run<0>(w, a, c, out, 200), run<1>(w, a, c, out, 200), run<2>(w, a, c, out, 200), run<3>(w, a, c, out, 200),
                      run<4>(w, a, c, out, 200), run<5>(w, a, c, out, 200), run<6>(w, a, c, out, 200)};
  for (int p = 0; p < 7; ++p) printf("part %d %-44s %6.2f us per launch (back to back)\n", p, name[p], t[p]);
  return 0;
}
