#include "common.h"
#include <cstdarg>
#include <cstdio>
namespace mimrl {
std::string& last_error_slot() { static thread_local std::string s; return s; }
int set_error(int code, const char* fmt, ...) {
  char buf[1024];
  va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
  last_error_slot() = buf;
  return code;
}
// result-changing debug knobs: compiled out of the default build (common.h: dbg_env) and refused by mimrl_create
const char* const kDebugKnobs[] = {"MIMRL_DBG_SKIP_WGRAD", "MIMRL_DBG_SKIP_EST", "MIMRL_DBG_SKIP_DEFERRED", "MIMRL_DBG_SKIP_IMGT",
                                   "MIMRL_DBG_MI", "MIMRL_DBG_MLPB", "MIMRL_DBG_GEMM", "MIMRL_DBG_KMIX", "MIMRL_CUBE_PHASE",
                                   "MIMRL_DBG_JOIN_AT", "MIMRL_DBG_DEFER_MAIN", "MIMRL_KMIX_PG_INCHAIN", "MIMRL_KMIX_PG_PARKED",
                                   "MIMRL_EARLY_FLUSH", nullptr};
}  // namespace mimrl
