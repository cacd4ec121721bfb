// Micro-benchmark: do f32 MFMAs (v_mfma_f32_32x32x2_f32) of one wave and plain f32 VALU work / global stores of
// ANOTHER wave on the same SIMD overlap on MI355X, or do they share the pipe?  (The f32 MFMA runs at the f32
// vector rate; if it occupies the vector ALU, every VALU instruction of a GEMM's prologue / epilogue costs
// matrix time - whichever wave issues it.)
//   hipcc --offload-arch=gfx950 -O3 mfma_valu.hip -o mfma_valu && ./mfma_valu
// 512-thread workgroups, one per CU: waves 0-3 (one per SIMD) issue NM MFMAs on 4 accumulators, waves 4-7
// issue NV independent v_fma_f32 (mode bit 1) or NS global_store_dword of 256 B (mode bit 2).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void __launch_bounds__(512) k(float *out, int nm, int nv, int ns, int mode, float seed) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (wave < 4) {
    if (!(mode & 1)) return;
    f32x16 a0 = {}, a1 = {}, a2 = {}, a3 = {};
    float x = seed + lane, y = seed * 0.5f + lane;
    for (int i = 0; i < nm; i += 4) {
      a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, x, a2, 0, 0, 0);
      a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, y, a3, 0, 0, 0);
    }
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += a0[r] + a1[r] + a2[r] + a3[r];
    if (s == 12345.678f) out[threadIdx.x] = s;
  } else if (mode & 2) {
    float v0 = seed + lane, v1 = v0 + 1.f, v2 = v0 + 2.f, v3 = v0 + 3.f, v4 = v0 + 4.f, v5 = v0 + 5.f, v6 = v0 + 6.f, v7 = v0 + 7.f;
    const float m = 1.0000001f, c = 1e-9f;
    for (int i = 0; i < nv; i += 8) {
      v0 = fmaf(v0, m, c); v1 = fmaf(v1, m, c); v2 = fmaf(v2, m, c); v3 = fmaf(v3, m, c);
      v4 = fmaf(v4, m, c); v5 = fmaf(v5, m, c); v6 = fmaf(v6, m, c); v7 = fmaf(v7, m, c);
    }
    const float s = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
    if (s == 12345.678f) out[threadIdx.x] = s;
  } else if (mode & 4) {
    float *o = out + 1024 + ((size_t)blockIdx.x * 4 + (wave - 4)) * 64 * 64 + lane;
    for (int i = 0; i < ns; ++i) o[(size_t)(i & 63) * 64] = seed + i;
  } else if (mode & 16) {
    // LDS traffic: ns ds_write_b32 (one dword per lane) or, with bit 32, ns ds_read_b128 (16 B per lane)
    __shared__ float4 lds[4][64 * 8];
    float4 *l = lds[wave - 4];
    float4 a = {seed, seed, seed, seed};
    if (mode & 32) {
      for (int i = 0; i < ns; i += 8) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = l[u * 64 + lane];
#pragma unroll
        for (int u = 0; u < 8; ++u) a.x += v[u].x;
        asm volatile("" ::: "memory");
      }
    } else {
      float *f = reinterpret_cast<float *>(l);
      for (int i = 0; i < ns; i += 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u) f[u * 64 + lane] = a.x + (float)u;
        asm volatile("" ::: "memory");
      }
      a.x = f[lane];
    }
    if (a.x == 12345.678f) out[threadIdx.x] = a.x;
  } else if (mode & 8) {
    // dwordx4 loads (1 KiB per wave-instruction) from a 64 KiB window per wave: L2 / L1 hits, 8 in flight
    const float4 *src = reinterpret_cast<const float4 *>(out + 1024 + ((size_t)blockIdx.x * 4 + (wave - 4)) * 64 * 64) + lane;
    float4 a = {0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < ns; i += 8) {
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = src[(size_t)((i + u) & 63) * 64];
#pragma unroll
      for (int u = 0; u < 8; ++u) { a.x += v[u].x; a.y += v[u].y; a.z += v[u].z; a.w += v[u].w; }
    }
    if (a.x + a.y + a.z + a.w == 12345.678f) out[threadIdx.x] = a.x;
  }
}

static float run(float *out, int nm, int nv, int ns, int mode) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, out, nm, nv, ns, mode, 1.5f);
  CK(hipEventRecord(e0));
  for (int w = 0; w < 10; ++w) hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, out, nm, nv, ns, mode, 1.5f);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 100.f;  // us per launch
}

int main() {
  float *out; CK(hipMalloc(&out, (1024 + (size_t)256 * 4 * 64 * 64) * sizeof(float)));
  const int nm = 8192;
  printf("MFMA alone (%d per wave, one wave per SIMD): %.1f us = %.1f cycles per MFMA at 2.4 GHz\n", nm, run(out, nm, 0, 0, 1), run(out, nm, 0, 0, 1) * 2400.0 / nm);
  for (int nv : {8192, 32768, 131072}) {
    const float tv = run(out, nm, nv, 0, 2), tb = run(out, nm, nv, 0, 3), tm = run(out, nm, 0, 0, 1);
    printf("VALU %6d fma/wave: alone %.1f us (%.2f cycles each)  MFMA alone %.1f  both %.1f us  (sum %.1f, max %.1f)\n", nv, tv, tv * 2400.0 / nv, tm, tb,
           tv + tm, tv > tm ? tv : tm);
  }
  for (int ns : {256, 1024, 4096}) {
    const float ts = run(out, nm, 0, ns, 4), tb = run(out, nm, 0, ns, 5), tm = run(out, nm, 0, 0, 1);
    printf("stores %5d/wave (256 B each): alone %.1f us (%.0f cycles each)  MFMA alone %.1f  both %.1f us  (sum %.1f)\n", ns, ts, ts * 2400.0 / ns, tm, tb, ts + tm);
  }
  for (int ns : {256, 1024, 4096}) {
    const float ts = run(out, nm, 0, ns, 8), tb = run(out, nm, 0, ns, 9), tm = run(out, nm, 0, 0, 1);
    printf("loads  %5d/wave (dwordx4, 1 KiB each, cache hits): alone %.1f us (%.0f cycles each)  MFMA alone %.1f  both %.1f us  (+%.0f cycles of MFMA time per load)\n", ns, ts,
           ts * 2400.0 / ns, tm, tb, (tb - tm) * 2400.0 / ns);
  }
  for (int ns : {4096, 16384}) {
    const float ts = run(out, nm, 0, ns, 16), tb = run(out, nm, 0, ns, 17), tm = run(out, nm, 0, 0, 1);
    printf("ds_write_b32 %5d/wave: alone %.1f us (%.1f cycles each)  MFMA alone %.1f  both %.1f us  (+%.1f cycles of MFMA time each)\n", ns, ts, ts * 2400.0 / ns, tm, tb,
           (tb - tm) * 2400.0 / ns);
    const float us = run(out, nm, 0, ns, 48), ub = run(out, nm, 0, ns, 49);
    printf("ds_read_b128 %5d/wave: alone %.1f us (%.1f cycles each)  MFMA alone %.1f  both %.1f us  (+%.1f cycles of MFMA time each)\n", ns, us, us * 2400.0 / ns, tm, ub,
           (ub - tm) * 2400.0 / ns);
  }
  return 0;
}
