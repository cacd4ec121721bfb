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

extern "C" int agx_prof_enable(int32_t filter) {
  AGX_CHECK_ARG(filter >= -2 && filter < AGX_K_COUNT, "agx_prof_enable: bad filter %d", filter);
  g_filter = filter;
  return AGX_OK;
}

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
