// Library-level state: thread-local error text, device selection.
#include "agx_common.h"

#include <cstring>

namespace {
thread_local char g_err[512] = "";
int g_cus = 256;  // MI355X; refreshed by agx_init
}  // namespace

namespace agx {
void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
int num_cus() { return g_cus; }
}  // namespace agx

extern "C" int agx_version(void) { return 100; }

extern "C" const char *agx_last_error(void) { return g_err; }

extern "C" int agx_init(int device) {
  int n = 0;
  AGX_HIP(hipGetDeviceCount(&n));
  AGX_CHECK_ARG(device >= 0 && device < n, "agx_init: device %d out of range (have %d)", device, n);
  AGX_HIP(hipSetDevice(device));
  hipDeviceProp_t prop;
  AGX_HIP(hipGetDeviceProperties(&prop, device));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    agx::set_error("agx_init: device %d is %s; libagx is built for gfx950 (MI355X) only", device,
                   prop.gcnArchName);
    return AGX_E_INVALID;
  }
  g_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  AGX_TRY(agx::gemm_prepare_device());
  AGX_TRY(agx::colstats_scratch_init());
  return AGX_OK;
}

extern "C" int64_t agx_sizeof(int32_t which) {
  switch (which) {
    case 0: return sizeof(agx_gemm_t);
    case 1: return sizeof(agx_embed_t);
    case 2: return sizeof(agx_small_grads_t);
    case 3: return sizeof(agx_rng_seg_t);
    case 4: return sizeof(agx_mlp_desc_t);
    case 5: return sizeof(agx_cgan_config_t);
    case 6: return sizeof(agx_step_args_t);
    case 7: return sizeof(agx_iter_args_t);
    case 8: return sizeof(agx_dp_args_t);
    default: return -1;
  }
}
