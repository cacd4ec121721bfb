// In-stream HIP-event timing of tagged launch sites (measurement support for bench.py:
// the roofline figure must come from the kernel's own duration inside the timed region).
#include "agx_common.h"

#include <mutex>
#include <vector>

namespace {

struct Slot {
  hipEvent_t start, stop;
};
struct ClassStat {
  std::vector<Slot> used;
  std::vector<Slot> free_;
  double flops = 0, bytes = 0;
  int64_t dropped = 0;
};
const char *const kNames[AGX_K_COUNT] = {
    "gen_fwd_hidden", "gen_fwd_out", "disc_fwd_in", "disc_fwd_hidden", "disc_delta_chain",
    "disc_dx", "gp2_in", "gp2_hidden", "disc_dw_in", "disc_dw_hidden", "disc_dcov", "gen_dw_out",
    "gen_dw_hidden", "gen_bwd_out", "gen_bwd_hidden", "opt_rmsprop", "opt_adam", "corr_xtx",
    "gemm_untagged"};
ClassStat g_stat[AGX_K_COUNT];
int g_filter = -2;
std::mutex g_mu;
constexpr size_t kMaxSlots = 1 << 15;

}  // namespace

namespace agx {

ProfScope::ProfScope(hipStream_t st, int id, double flops, double bytes) : st_(st), slot_(nullptr) {
  if (g_filter == -2 || id < 0 || id >= AGX_K_COUNT || (g_filter >= 0 && g_filter != id)) return;
  std::lock_guard<std::mutex> lk(g_mu);
  ClassStat &c = g_stat[id];
  if (c.used.size() >= kMaxSlots) {
    ++c.dropped;
    return;
  }
  Slot s;
  if (!c.free_.empty()) {
    s = c.free_.back();
    c.free_.pop_back();
  } else {
    if (hipEventCreate(&s.start) != hipSuccess) return;
    if (hipEventCreate(&s.stop) != hipSuccess) return;
  }
  c.used.push_back(s);
  c.flops += flops;
  c.bytes += bytes;
  slot_ = &c.used.back();  // only dereferenced in the destructor of this same scope (no push between)
  (void)hipEventRecord(s.start, st_);
  stop_ = s.stop;
}

ProfScope::~ProfScope() {
  if (slot_) (void)hipEventRecord(stop_, st_);
}

}  // namespace agx

namespace {
// Bounded busy-wait (100 MHz wall clock); lets the host queue a whole iteration behind it so that
// the event brackets that follow time back-to-back GPU execution, not host launch gaps.
__global__ void spin_kernel(unsigned long long ticks) {
  const unsigned long long t0 = wall_clock64();
  for (int i = 0; i < (1 << 24); ++i) {
    if (wall_clock64() - t0 >= ticks) break;
    __builtin_amdgcn_s_sleep(32);
  }
}
}  // namespace

namespace {
typedef float f32x16p __attribute__((ext_vector_type(16)));
// measures the shader clock a short MFMA-dense kernel actually gets: s_memtime ticks per
// 100 MHz wall-clock tick over `n` dependent 32x32x2 MFMAs per wave
__global__ void __launch_bounds__(256) clock_probe_kernel(int n, float seed, float *out) {
  f32x16p acc;
  for (int r = 0; r < 16; ++r) acc[r] = seed;
  const unsigned long long c0 = clock64(), w0 = wall_clock64();
  float a = seed + threadIdx.x, b = seed - 1.f;
  for (int i = 0; i < n; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
  const unsigned long long c1 = clock64(), w1 = wall_clock64();
  float sum = 0.f;
  for (int r = 0; r < 16; ++r) sum += acc[r];
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    out[0] = (float)((double)(c1 - c0) / (double)(w1 - w0) * 100.0);  // MHz
    out[1] = (float)(c1 - c0) / (float)n;                              // cycles per MFMA
    out[2] = sum;
  }
}
}  // namespace

extern "C" int agx_prof_clock(void *stream, int32_t blocks, int32_t mfmas, float *out_dev) {
  AGX_CHECK_ARG(out_dev && blocks > 0 && blocks <= 4096 && mfmas > 0 && mfmas <= (1 << 20), "agx_prof_clock: bad argument");
  hipLaunchKernelGGL(clock_probe_kernel, dim3(blocks), dim3(256), 0, agx::as_stream(stream), mfmas, 0.f, out_dev);
  AGX_LAUNCH_CHECK();
  return AGX_OK;
}

extern "C" int agx_prof_spin(void *stream, int32_t microseconds) {
  AGX_CHECK_ARG(microseconds >= 0 && microseconds <= 100000, "agx_prof_spin: 0..100000 us");
  hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, agx::as_stream(stream),
                     (unsigned long long)microseconds * 100ull);
  AGX_LAUNCH_CHECK();
  return AGX_OK;
}

// Mean elapsed time of n EMPTY brackets (start and stop recorded back to back on the stream, queued
// behind a busy-wait like the measured ones): what a bracket adds to the launch it encloses.
extern "C" int agx_prof_empty_bracket_us(void *stream, int32_t n, double *mean_us) {
  AGX_CHECK_ARG(n > 0 && n <= 1024 && mean_us, "agx_prof_empty_bracket_us: bad argument");
  hipStream_t st = agx::as_stream(stream);
  std::vector<hipEvent_t> ev(2 * (size_t)n);
  for (auto &e : ev) AGX_HIP(hipEventCreate(&e));
  hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, st, 2000ull * 100ull);
  for (int i = 0; i < n; ++i) {
    (void)hipEventRecord(ev[2 * i], st);
    (void)hipEventRecord(ev[2 * i + 1], st);
  }
  AGX_HIP(hipStreamSynchronize(st));
  double tot = 0;
  for (int i = 0; i < n; ++i) {
    float ms = 0;
    AGX_HIP(hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1]));
    tot += ms;
  }
  for (auto &e : ev) (void)hipEventDestroy(e);
  *mean_us = tot / n * 1e3;
  return AGX_OK;
}

extern "C" int agx_prof_enable(int32_t filter) {
  AGX_CHECK_ARG(filter >= -2 && filter < AGX_K_COUNT, "agx_prof_enable: bad filter %d", filter);
  g_filter = filter;
  return AGX_OK;
}

extern "C" int agx_prof_enabled(void) { return g_filter != -2; }

extern "C" int agx_prof_reset(void) {
  std::lock_guard<std::mutex> lk(g_mu);
  for (auto &c : g_stat) {
    for (auto &s : c.used) {
      (void)hipEventSynchronize(s.stop);
      c.free_.push_back(s);
    }
    c.used.clear();
    c.flops = c.bytes = 0;
    c.dropped = 0;
  }
  return AGX_OK;
}

extern "C" int agx_prof_read(int32_t id, double *total_ms, int64_t *count, double *flops,
                             double *bytes) {
  AGX_CHECK_ARG(id >= 0 && id < AGX_K_COUNT && total_ms && count && flops && bytes,
                "agx_prof_read: bad argument");
  std::lock_guard<std::mutex> lk(g_mu);
  ClassStat &c = g_stat[id];
  double tot = 0;
  for (auto &s : c.used) {
    AGX_HIP(hipEventSynchronize(s.stop));
    float ms = 0;
    AGX_HIP(hipEventElapsedTime(&ms, s.start, s.stop));
    tot += ms;
  }
  *total_ms = tot;
  *count = (int64_t)c.used.size();
  *flops = c.flops;
  *bytes = c.bytes;
  return AGX_OK;
}

extern "C" const char *agx_prof_name(int32_t id) {
  return (id >= 0 && id < AGX_K_COUNT) ? kNames[id] : "";
}
