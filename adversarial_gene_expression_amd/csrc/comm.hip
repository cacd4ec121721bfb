// RCCL over xGMI behind the C-ABI (agx_comm_*): lets a host without PyTorch run the
// data-parallel train step (reduce-scatter of the flat gradient buffer, all-gather of the parameters, or a
// plain sum all-reduce) and the 6-scalar all-reduce of the tiled Gamma evaluation on the stream the kernels
// run on.  No reference counterpart (the reference is single-process, tf_utils.py:17-38).
//
// librccl is bound at run time (dlopen, RTLD_LOCAL): libagx.so carries no link-time dependency on it, and inside
// a PyTorch process the copy PyTorch has already loaded is used instead of a second one.
#include "agx_common.h"

#include <dlfcn.h>

#include <cstring>

#include <mutex>
#include <new>

namespace {

typedef struct { char internal[128]; } rcclUniqueId;   // ncclUniqueId (NCCL_UNIQUE_ID_BYTES = 128)
typedef void *rcclComm_t;
enum { kSum = 0, kFloat32 = 7, kFloat64 = 8 };           // ncclSum, ncclFloat32, ncclFloat64 (rccl.h)

struct Rccl {
  void *lib = nullptr;
  int (*GetUniqueId)(rcclUniqueId *) = nullptr;
  int (*CommInitRank)(rcclComm_t *, int, rcclUniqueId, int) = nullptr;
  int (*CommDestroy)(rcclComm_t) = nullptr;
  int (*AllReduce)(const void *, void *, size_t, int, int, rcclComm_t, hipStream_t) = nullptr;
  int (*ReduceScatter)(const void *, void *, size_t, int, int, rcclComm_t, hipStream_t) = nullptr;
  int (*AllGather)(const void *, void *, size_t, int, rcclComm_t, hipStream_t) = nullptr;
  const char *(*GetErrorString)(int) = nullptr;
};
Rccl g_rccl;
std::mutex g_rccl_mu;

int load_rccl() {
  std::lock_guard<std::mutex> lock(g_rccl_mu);
  if (g_rccl.lib) return AGX_OK;
  const char *names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
  void *lib = nullptr;
  for (const char *n : names) {            // a copy that is already in the process (PyTorch's) first
    lib = dlopen(n, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
    if (lib) break;
  }
  for (int i = 0; !lib && i < 3; ++i) lib = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
  if (!lib) {
    agx::set_error("agx_comm: librccl.so not found (%s)", dlerror());
    return AGX_E_STATE;
  }
  Rccl r;
  r.lib = lib;
#define AGX_SYM(field, name)                                            \
  *(void **)(&r.field) = dlsym(lib, name);                              \
  if (!r.field) {                                                       \
    agx::set_error("agx_comm: librccl has no symbol %s", name);         \
    return AGX_E_STATE;                                                 \
  }
  AGX_SYM(GetUniqueId, "ncclGetUniqueId")
  AGX_SYM(CommInitRank, "ncclCommInitRank")
  AGX_SYM(CommDestroy, "ncclCommDestroy")
  AGX_SYM(AllReduce, "ncclAllReduce")
  AGX_SYM(ReduceScatter, "ncclReduceScatter")
  AGX_SYM(AllGather, "ncclAllGather")
  AGX_SYM(GetErrorString, "ncclGetErrorString")
#undef AGX_SYM
  g_rccl = r;
  return AGX_OK;
}

#define AGX_RCCL(call)                                                                     \
  do {                                                                                     \
    int rc_ = (call);                                                                      \
    if (rc_ != 0) {                                                                        \
      agx::set_error("%s failed: %s", #call, g_rccl.GetErrorString ? g_rccl.GetErrorString(rc_) : "?"); \
      return AGX_E_HIP;                                                                    \
    }                                                                                      \
  } while (0)

}  // namespace

struct agx_comm {
  rcclComm_t comm = nullptr;
  int nranks = 0, rank = 0;
};

extern "C" int agx_comm_unique_id(void *id128_host) {
  AGX_CHECK_ARG(id128_host, "agx_comm_unique_id: null argument");
  AGX_TRY(load_rccl());
  rcclUniqueId id;
  AGX_RCCL(g_rccl.GetUniqueId(&id));
  memcpy(id128_host, &id, sizeof(id));
  return AGX_OK;
}

extern "C" int agx_comm_init(const void *id128_host, int32_t nranks, int32_t rank, agx_comm_t **out) {
  AGX_CHECK_ARG(id128_host && out && nranks >= 1 && rank >= 0 && rank < nranks, "agx_comm_init: bad argument");
  AGX_TRY(load_rccl());
  agx_comm *c = new (std::nothrow) agx_comm();
  AGX_CHECK_ARG(c, "agx_comm_init: out of host memory");
  rcclUniqueId id;
  memcpy(&id, id128_host, sizeof(id));
  const int rc = g_rccl.CommInitRank(&c->comm, nranks, id, rank);   // one process per GPU: the current device
  if (rc != 0) {
    agx::set_error("ncclCommInitRank(%d ranks, rank %d) failed: %s", nranks, rank, g_rccl.GetErrorString(rc));
    delete c;
    return AGX_E_HIP;
  }
  c->nranks = nranks;
  c->rank = rank;
  *out = c;
  return AGX_OK;
}

extern "C" int agx_comm_destroy(agx_comm_t *c) {
  if (c) {
    if (c->comm) (void)g_rccl.CommDestroy(c->comm);
    delete c;
  }
  return AGX_OK;
}

extern "C" int32_t agx_comm_size(const agx_comm_t *c) { return c ? c->nranks : 0; }
extern "C" int32_t agx_comm_rank(const agx_comm_t *c) { return c ? c->rank : -1; }

extern "C" int agx_comm_allreduce_sum(agx_comm_t *c, void *stream, void *buf, int64_t n, int32_t f64) {
  AGX_CHECK_ARG(c && c->comm && buf && n >= 0, "agx_comm_allreduce_sum: bad argument");
  if (n == 0) return AGX_OK;
  AGX_RCCL(g_rccl.AllReduce(buf, buf, (size_t)n, f64 ? kFloat64 : kFloat32, kSum, c->comm, agx::as_stream(stream)));
  return AGX_OK;
}

extern "C" int agx_comm_reduce_scatter_sum(agx_comm_t *c, void *stream, const float *in, float *out,
                                           int64_t n_per_rank) {
  AGX_CHECK_ARG(c && c->comm && in && out && n_per_rank > 0, "agx_comm_reduce_scatter_sum: bad argument");
  AGX_RCCL(g_rccl.ReduceScatter(in, out, (size_t)n_per_rank, kFloat32, kSum, c->comm, agx::as_stream(stream)));
  return AGX_OK;
}

extern "C" int agx_comm_all_gather(agx_comm_t *c, void *stream, const float *in, float *out, int64_t n_per_rank) {
  AGX_CHECK_ARG(c && c->comm && in && out && n_per_rank > 0, "agx_comm_all_gather: bad argument");
  AGX_RCCL(g_rccl.AllGather(in, out, (size_t)n_per_rank, kFloat32, c->comm, agx::as_stream(stream)));
  return AGX_OK;
}
