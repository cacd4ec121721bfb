// Micro-benchmark: cost of a device-wide barrier (atomic ticket + bounded spin) on MI355X versus a
// dependent kernel boundary.  Decides whether a persistent whole-step kernel could beat 8 launches.
//   hipcc --offload-arch=gfx950 -O3 grid_barrier.hip -o grid_barrier && ./grid_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void __launch_bounds__(256) barrier_loop(unsigned *counter, int iters, unsigned *fail, float *sink, const float *src, int touch) {
  const unsigned nb = gridDim.x;
  float acc = 0.f;
  for (int it = 0; it < iters; ++it) {
    if (touch) {   // a little memory work between barriers, written before / read after the barrier (visibility check)
      sink[(size_t)blockIdx.x * 256 + threadIdx.x] = (float)it;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      __threadfence();
      atomicAdd(counter, 1u);
      const unsigned target = (unsigned)(it + 1) * nb;
      int spins = 0;
      while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        if (++spins > 2000000) { atomicAdd(fail, 1u); break; }   // bounded: never hangs
        __builtin_amdgcn_s_sleep(1);
      }
      __threadfence();
    }
    __syncthreads();
    if (touch) {
      const unsigned nbr = (blockIdx.x + 37) % nb;
      acc += sink[(size_t)nbr * 256 + threadIdx.x];   // must see the neighbour's value of this iteration
      if (sink[(size_t)nbr * 256 + threadIdx.x] != (float)it) atomicAdd(fail + 1, 1u);
    }
    __syncthreads();
    if (touch && threadIdx.x == 0) {   // second barrier so nobody overwrites before the neighbours have read
      __threadfence();
      atomicAdd(counter + 32, 1u);
      const unsigned target = (unsigned)(it + 1) * nb;
      int spins = 0;
      while (__hip_atomic_load(counter + 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        if (++spins > 2000000) { atomicAdd(fail, 1u); break; }
        __builtin_amdgcn_s_sleep(1);
      }
    }
    __syncthreads();
  }
  if (touch && acc == -1.f) sink[0] = acc;
}

__global__ void __launch_bounds__(256) tiny(float *sink) { sink[(size_t)blockIdx.x * 256 + threadIdx.x] += 1.f; }

int main() {
  unsigned *counter, *fail;
  float *sink;
  CK(hipMalloc(&counter, 4096)); CK(hipMalloc(&fail, 64)); CK(hipMalloc(&sink, 1024 * 256 * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipStream_t st; CK(hipStreamCreate(&st));
  for (int touch = 0; touch < 2; ++touch)
    for (int nb : {64, 256, 512}) {
      const int iters = 2000;
      CK(hipMemsetAsync(counter, 0, 4096, st)); CK(hipMemsetAsync(fail, 0, 64, st));
      CK(hipMemsetAsync(sink, 0, 1024 * 256 * 4, st));
      CK(hipEventRecord(e0, st));
      hipLaunchKernelGGL(barrier_loop, dim3(nb), dim3(256), 0, st, counter, iters, fail, sink, sink, touch);
      CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      unsigned f[2]; CK(hipMemcpy(f, fail, 8, hipMemcpyDeviceToHost));
      printf("grid barrier: %4d workgroups, touch=%d: %.3f us per %s (timeouts %u, stale reads %u)\n", nb, touch,
             ms * 1e3 / iters, touch ? "pair of barriers + store/load" : "barrier", f[0], f[1]);
    }
  // dependent launches of a tiny kernel, for comparison (graph-free, one stream)
  for (int nb : {64, 256}) {
    const int n = 2000;
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(tiny, dim3(nb), dim3(256), 0, st, sink);
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL(tiny, dim3(nb), dim3(256), 0, st, sink);
    CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("dependent tiny launches: %4d workgroups: %.3f us per launch\n", nb, ms * 1e3 / n);
  }
  // same inside a graph
  {
    const int n = 200;
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed));
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL(tiny, dim3(256), dim3(256), 0, st, sink);
    CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int r = 0; r < 10; ++r) CK(hipGraphLaunch(ge, st));
    CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("graph of %d dependent tiny kernels: %.3f us per kernel\n", n, ms * 1e3 / (10 * n));
  }
  return 0;
}
