// see comm.h
#include "comm.h"

#include <dlfcn.h>
#include <cstring>
#include <rccl/rccl.h>   // types and constants only: every entry point is resolved with dlsym

#include <mutex>

namespace mimrl {

namespace {
struct Rccl {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl g_rccl;
std::once_flag g_once;
bool g_ok = false;

int rccl_load() {
  std::call_once(g_once, [] {
    for (const char* nm : {"librccl.so.1", "librccl.so"}) {
      g_rccl.lib = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
      if (g_rccl.lib) break;
    }
    if (!g_rccl.lib) return;
    auto sym = [](const char* n) { return dlsym(g_rccl.lib, n); };
    g_rccl.GetUniqueId = reinterpret_cast<decltype(g_rccl.GetUniqueId)>(sym("ncclGetUniqueId"));
    g_rccl.CommInitRank = reinterpret_cast<decltype(g_rccl.CommInitRank)>(sym("ncclCommInitRank"));
    g_rccl.AllReduce = reinterpret_cast<decltype(g_rccl.AllReduce)>(sym("ncclAllReduce"));
    g_rccl.CommDestroy = reinterpret_cast<decltype(g_rccl.CommDestroy)>(sym("ncclCommDestroy"));
    g_rccl.GetErrorString = reinterpret_cast<decltype(g_rccl.GetErrorString)>(sym("ncclGetErrorString"));
    g_ok = g_rccl.GetUniqueId && g_rccl.CommInitRank && g_rccl.AllReduce && g_rccl.CommDestroy && g_rccl.GetErrorString;
  });
  if (!g_ok) return set_error(MIMRL_ERR_STATE, "RCCL is not available: librccl.so.1 could not be loaded (%s)", g_rccl.lib ? "missing symbols" : dlerror());
  return MIMRL_OK;
}
#define NCX(expr)                                                                                     \
  do {                                                                                                \
    const ncclResult_t r_ = (expr);                                                                   \
    if (r_ != ncclSuccess) return set_error(MIMRL_ERR_HIP, "RCCL: %s: %s", #expr, g_rccl.GetErrorString(r_)); \
  } while (0)
}  // namespace

int comm_unique_id(void* out128) {
  if (!out128) return set_error(MIMRL_ERR_ARG, "null argument");
  MX(rccl_load());
  static_assert(sizeof(ncclUniqueId) == 128, "mimrl_comm_unique_id hands out 128 bytes");
  NCX(g_rccl.GetUniqueId(static_cast<ncclUniqueId*>(out128)));
  return MIMRL_OK;
}

int comm_init(void** comm, const void* id128, int world, int rank) {
  if (!comm || !id128 || world < 1 || rank < 0 || rank >= world) return set_error(MIMRL_ERR_ARG, "comm_init: bad arguments (world %d, rank %d)", world, rank);
  MX(rccl_load());
  ncclUniqueId id;
  std::memcpy(&id, id128, sizeof id);
  ncclComm_t c = nullptr;
  NCX(g_rccl.CommInitRank(&c, world, id, rank));
  *comm = c;
  return MIMRL_OK;
}

int comm_allreduce_sum(void* comm, float* buf, size_t n, hipStream_t s) {
  if (!comm) return set_error(MIMRL_ERR_STATE, "no communicator");
  if (n == 0) return MIMRL_OK;
  NCX(g_rccl.AllReduce(buf, buf, n, ncclFloat32, ncclSum, static_cast<ncclComm_t>(comm), s));
  return MIMRL_OK;
}

int comm_allreduce_sum_bf16(void* comm, void* buf, size_t n, hipStream_t s) {
  if (!comm) return set_error(MIMRL_ERR_STATE, "no communicator");
  if (n == 0) return MIMRL_OK;
  NCX(g_rccl.AllReduce(buf, buf, n, ncclBfloat16, ncclSum, static_cast<ncclComm_t>(comm), s));
  return MIMRL_OK;
}

int comm_destroy(void* comm) {
  if (!comm) return MIMRL_OK;
  MX(rccl_load());
  NCX(g_rccl.CommDestroy(static_cast<ncclComm_t>(comm)));
  return MIMRL_OK;
}

}  // namespace mimrl
