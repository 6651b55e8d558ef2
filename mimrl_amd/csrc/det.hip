// Host side and flush kernel of the deterministic build (det.h).  In the default build this file only answers mimrl_deterministic() = 0.
#include <atomic>
#include <cstring>
#include <mutex>
#include <vector>

#include "common.h"

namespace mimrl {

#ifdef MIMRL_DET
namespace {
constexpr unsigned kSlots = 1u << 23;      // 8 M addresses per launch (the grouped critic weight gradients touch ~2.5 M at cfg2)
DetCtx h_ctx = {};
std::atomic<bool> ready{false};
bool failed = false;
std::mutex mu;
std::vector<DetSetter>& setters() { static std::vector<DetSetter> v; return v; }

__global__ __launch_bounds__(256) void det_flush_kernel(DetCtx c) {
  // n changes only when the last PARTICIPATING workgroup of this launch resets it, i.e. after every participant has read it; a workgroup
  // that starts after that sees 0 and leaves (most launches of a step accumulate nothing: their flush is this one load)
  const unsigned n = __atomic_load_n(&c.ctl[0], __ATOMIC_RELAXED);
  const unsigned nb = min(gridDim.x, (n + 255u) / 256u);
  if (blockIdx.x >= nb) return;
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += nb * blockDim.x) {
    const unsigned h = c.list[i];
    float* p = reinterpret_cast<float*>(c.keys[h]);
    *p += (float)c.vals[h] * kDetInv;       // the ONE rounding of this address's sum
    c.keys[h] = 0ull;
    c.vals[h] = 0ll;
  }
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0 && atomicAdd(&c.ctl[1], 1u) == nb - 1) { c.ctl[1] = 0u; __threadfence(); c.ctl[0] = 0u; }
}
}  // namespace

void det_register_tu(DetSetter f) { setters().push_back(f); }

int det_init() {
  if (ready.load(std::memory_order_acquire)) return MIMRL_OK;
  std::lock_guard<std::mutex> g(mu);
  if (ready.load() || failed) return failed ? MIMRL_ERR_HIP : MIMRL_OK;
  failed = true;                            // (until the last line: a box without a GPU lands here once, not on every launch)
  HIPX(hipMalloc(&h_ctx.keys, sizeof(unsigned long long) * kSlots));
  HIPX(hipMalloc(&h_ctx.vals, sizeof(long long) * kSlots));
  HIPX(hipMalloc(&h_ctx.list, sizeof(unsigned) * kSlots));
  HIPX(hipMalloc(&h_ctx.ctl, sizeof(unsigned) * 4));
  HIPX(hipMemset(h_ctx.keys, 0, sizeof(unsigned long long) * kSlots));
  HIPX(hipMemset(h_ctx.vals, 0, sizeof(long long) * kSlots));
  HIPX(hipMemset(h_ctx.ctl, 0, sizeof(unsigned) * 4));
  h_ctx.mask = kSlots - 1;
  for (DetSetter f : setters()) f(h_ctx);
  HIPX(hipDeviceSynchronize());
  HIPX(hipGetLastError());
  failed = false;
  ready.store(true, std::memory_order_release);
  return MIMRL_OK;
}

int det_flush(hipStream_t s) {
  if (!ready.load(std::memory_order_acquire)) return MIMRL_OK;   // no table (init failed): acc_add fell back to float atomics
  det_flush_kernel<<<dim3(256), dim3(256), 0, s>>>(h_ctx);
  return MIMRL_OK;
}

namespace {
thread_local bool t_no_flush = false;
thread_local bool t_defer = false, t_pending = false, t_gemm_in_bucket = false;
// the bucket ranges of the handle whose DetDefer scope is open on this thread (round 6: one process-global table, overwritten at every
// handle's bind and never cleared, classified handle A's GEMM outputs against handle B's ranges -- ADVICE r05)
thread_local const DetRanges* t_ranges = nullptr;
}
bool det_target_in_bucket(const void* p) {
  const char* c = static_cast<const char*>(p);
  if (!t_ranges) return false;
  for (int i = 0; i < t_ranges->n; ++i)
    if (t_ranges->lo[i] && c >= static_cast<const char*>(t_ranges->lo[i]) && c < static_cast<const char*>(t_ranges->lo[i]) + t_ranges->bytes[i]) return true;
  return false;
}
DetGemmTarget::DetGemmTarget(const void* c) : prev_(t_gemm_in_bucket) { t_gemm_in_bucket = det_target_in_bucket(c); }
DetGemmTarget::~DetGemmTarget() { t_gemm_in_bucket = prev_; }
DetDefer::DetDefer(hipStream_t s, const DetRanges* r) : s_(s), prev_(t_defer), prev_r_(t_ranges) { t_defer = true; t_ranges = r; }
DetDefer::~DetDefer() {
  t_defer = prev_;
  t_ranges = prev_r_;
  if (!t_defer && t_pending) { t_pending = false; (void)det_flush(s_); }
}
DetNoFlush::DetNoFlush(bool on) : on_(on), prev_(t_no_flush) { if (on_) t_no_flush = true; }
DetNoFlush::~DetNoFlush() { if (on_) t_no_flush = prev_; }

bool det_launch_accumulates(const char* name) {
  if (t_no_flush) return false;
  if (t_defer) {
    // kernels whose every acc_add target is a gradient-bucket tensor (checked against the sources, round 5b): LayerNorm / bias / weight
    // gradients of the model backward and the estimator stacks, the BPTT's bias gradients, column / row sums into the buckets
    static const char* const kBucketOnly[] = {"ln_relu_drop_bwd", "head_bwd_kernel", "colln_bwd_kernel", "rowln_bwd_kernel", "kmix_bwd_kernel",
                                              "colsum_kernel", "rowsum_batched", "colln_param_grads", "daxis_param_grads", "rowln_param_grads",
                                              "laxis_bwd_kernel", "daxis_bwd_kernel", "gru_bwd_kernel", "top1_bwd_kernel", "concat_dw3_kernel",
                                              "mlp_frag_kernel", "gemm_group_kernel", "gemm_groupk_kernel"};
    for (const char* k : kBucketOnly)
      if (std::strstr(name, k)) { t_pending = true; return false; }
    if (t_gemm_in_bucket && std::strstr(name, "gemm_")) { t_pending = true; return false; }
  }
  // kernels that never call acc_add (tests/test_layout.py::test_det_flush_rule_matches_the_sources checks this list against the sources)
  static const char* const kSafe[] = {"_fwd", "adam_kernel", "images_kernel", "bf16_image", "knn_", "sample_anchors", "seq_lengths", "l0_pack", "l0_unpack",
                                      "text_post", "feat_mean", "tail_pre", "begin_stage", "mae_kernel", "finalize_stage", "stage_boundary",
                                      "cmi_assemble", "copy_rows", "gather_sum4", "wt_transpose", "pad_rows", "mi_bound", "cmi_loss", "pair_expand",
                                      "pair_reduce", "gauss_baseline", "dbg_spin", "det_flush", "dropout_inplace", "add_inplace",
                                      "mi_sep_fused", "gemm_tall_kernel", "concat_ws_reduce"};
  for (const char* k : kSafe)
    if (std::strstr(name, k)) return false;
  return true;
}

int det_overflowed() {
  if (!ready.load()) return 0;
  unsigned f[2] = {0, 0};
  if (hipMemcpy(f, h_ctx.ctl + 2, sizeof f, hipMemcpyDeviceToHost) != hipSuccess) return 1;
  return (f[0] != 0 ? 1 : 0) | (f[1] != 0 ? 2 : 0);
}
#endif

}  // namespace mimrl

extern "C" int mimrl_deterministic(void) {
#ifdef MIMRL_DET
  // 1 = deterministic build, every sum so far order-independent; | 2: the accumulation table ran full at some launch; | 4: some
  // contribution was NaN / Inf / >= 2^22 in magnitude -- in both cases that contribution went through a plain float atomic (a NaN
  // propagates exactly as in the default build) and the run was not bit-reproducible
  const int f = ::mimrl::det_overflowed();
  return 1 | ((f & 1) ? 2 : 0) | ((f & 2) ? 4 : 0);
#else
  return 0;
#endif
}
