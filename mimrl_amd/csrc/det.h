// Order-independent accumulation for the deterministic build (`make det` -> libmimrl_hip_det.so, selected by MIMRL_DETERMINISTIC=1;
// the reference's switch is torch.backends.cudnn.deterministic, Main.py:19-20).
//
// Every float atomicAdd in this library goes through acc_add() (global targets) or LdsAcc (LDS targets).  In the default build both ARE
// the float atomics they replace.  With -DMIMRL_DET:
//   * LdsAcc holds a 64-bit fixed-point sum (2^-40 units): integer addition is associative, so the order in which the waves of a
//     workgroup arrive no longer matters;
//   * acc_add(p, v) adds round(v * 2^40) into a 64-bit slot of an open-addressing table keyed by the target ADDRESS (first touch claims
//     the slot with a CAS and appends it to a dirty list).  det_flush() -- enqueued on the launch stream behind EVERY kernel launch of the
//     build (common.h redefines hipLaunchKernelGGL) -- walks the dirty list once: *p += float(sum * 2^-40), one rounding per address,
//     and hands the slots back.  The engine runs single-stream in this build, so a flush never sees a half-finished producer.
// Cost: a table probe + two 64-bit atomics per contribution, ~170 extra launches per step; bench.py reports the step time of this build
// beside the default one.  |sum| < 2^21 (gradient sums are orders of magnitude below it; values >= 2^-16 convert exactly); a NaN / Inf /
// |v| >= 2^22 contribution is NOT converted: it takes the float atomic (global) or poisons the LDS sum to NaN, and is flagged.
#pragma once
#include <hip/hip_runtime.h>

namespace mimrl {

// gradient-bucket address ranges of ONE engine handle (deterministic build: what a DetDefer scope may defer; unused by the default build)
struct DetRanges { const void* lo[2] = {nullptr, nullptr}; size_t bytes[2] = {0, 0}; int n = 0; };

#ifdef MIMRL_DET
struct DetCtx {
  unsigned long long* keys;   // target address, 0 = free
  long long* vals;            // fixed-point sum
  unsigned* list;             // claimed slots, in claim order (order is irrelevant: one add per address)
  unsigned* ctl;              // [0] = number of claimed slots, [1] = workgroups done (flush), [2] = overflow flag (sticky),
                              // [3] = a non-finite / out-of-range contribution fell back to the float atomic (sticky)
  unsigned mask;              // slots - 1
};
static __device__ DetCtx g_det;   // one copy per translation unit: det_register_tu() below
constexpr float kDetScale = 0x1p40f, kDetInv = 0x1p-40f;

__device__ __forceinline__ long long det_fix(float v) { return __float2ll_rn(v * kDetScale); }

// largest |v| a contribution may have: 2^40 * 2^22 = 2^62 keeps one contribution inside int64
constexpr float kDetMaxAbs = 0x1p22f;

__device__ __forceinline__ void acc_add(float* p, float v) {
  const DetCtx c = g_det;
  // NaN, Inf and |v| >= 2^22 do not convert (NaN -> 0 would silently DROP a divergence the default build propagates; large values
  // saturate): such a contribution goes through the float atomic, so the target becomes NaN / Inf exactly as in the default build, and
  // the sticky flag says this launch was not order-independent (mimrl_deterministic() bit 2)
  if (!(fabsf(v) < kDetMaxAbs)) { if (c.ctl) c.ctl[3] = 1u; atomicAdd(p, v); return; }
  const long long fx = det_fix(v);
  if (fx == 0) return;
  if (!c.keys) { atomicAdd(p, v); return; }   // no table (det_init failed)
  // slot = (hash of the 64-byte line, float within the line): the 16 floats of a line share one 128-byte run of `vals`, so a tile epilogue
  // touches as many lines of the table as of its target; a collision moves the whole line to the next group
  const unsigned long long key = reinterpret_cast<unsigned long long>(p);
  const unsigned low = (unsigned)(key >> 2) & 15u, gmask = c.mask >> 4;
  unsigned g = (unsigned)(((key >> 6) * 0x9E3779B97F4A7C15ull) >> 37) & gmask, h = 0;
  bool claimed = false;
  for (unsigned probe = 0;; ++probe) {
    h = (g << 4) | low;
    unsigned long long k = __atomic_load_n(&c.keys[h], __ATOMIC_RELAXED);
    if (k == 0) {
      k = atomicCAS(&c.keys[h], 0ull, key);
      if (k == 0) { claimed = true; break; }
    }
    if (k == key) break;
    if (probe > gmask) { c.ctl[2] = 1u; atomicAdd(p, v); return; }   // table full: flagged (mimrl_deterministic_overflowed)
    g = (g + 1) & gmask;
  }
  // first touches join the dirty list: ONE counter bump per wave (a bump per address serialised millions of atomics on one word:
  // 18 ms per cfg2 step)
  const unsigned long long m = __ballot(claimed);
  if (claimed) {
    const unsigned lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const int leader = __ffsll((long long)m) - 1;
    unsigned base = 0;
    if ((int)lane == leader) base = atomicAdd(&c.ctl[0], (unsigned)__popcll(m));
    base = __shfl(base, leader, 64);
    c.list[base + (unsigned)__popcll(m & ((1ull << lane) - 1ull))] = h;
  }
  atomicAdd(reinterpret_cast<unsigned long long*>(&c.vals[h]), (unsigned long long)fx);
}

struct LdsAcc {
  long long v;
  __device__ __forceinline__ void zero() { v = 0; }
  // a non-finite / out-of-range term poisons the sum (INT64_MIN marker, sticky under further adds of |x| < 2^22 for ~2^40 terms): get()
  // then answers NaN, which is what a float sum with a NaN / Inf term would (nearly always) give -- never a silently dropped term
  __device__ __forceinline__ void add(float x) {
    if (!(fabsf(x) < kDetMaxAbs)) { atomicExch(reinterpret_cast<unsigned long long*>(&v), 0x8000000000000000ull); return; }
    atomicAdd(reinterpret_cast<unsigned long long*>(&v), (unsigned long long)det_fix(x));
  }
  __device__ __forceinline__ float get() const {
    return (v < -(1ll << 61) || v > (1ll << 61)) ? __builtin_nanf("") : (float)v * kDetInv;
  }
};

// host side (det.cpp)
typedef void (*DetSetter)(const DetCtx&);
void det_register_tu(DetSetter f);
int det_init();                              // allocates the table, hands it to every translation unit (first launch; idempotent)
int det_flush(hipStream_t s);                // apply and clear what the launches so far accumulated
// may the kernel just launched have called acc_add?  `name` = the launch macro's kernel expression as text.  False for the kernels known
// never to (forward / bookkeeping / image / sampler kernels, by name) and inside a DetNoFlush scope; anything unknown answers true.
bool det_launch_accumulates(const char* name);
struct DetNoFlush {                          // host scope: "the launches in here do not accumulate" (when `on`)
  explicit DetNoFlush(bool on);
  ~DetNoFlush();
  bool on_, prev_;
};
// Round 5b: launches that accumulate ONLY into gradient buckets (weight / bias / LayerNorm gradients: nobody reads them before the pass that
// produced them is over) need no flush of their own.  Inside a DetDefer scope (mimrl_handle::model_backward / estimators_all) such launches --
// known by kernel name, or a GEMM whose output lies in a registered bucket range -- skip it; the scope's end flushes once.  Every other launch
// flushes as before (and takes the pending sums along: the flush walks the whole dirty list).
bool det_target_in_bucket(const void* p);     // inside the innermost DetDefer scope's ranges
struct DetGemmTarget {                        // gemm(): "the launch below accumulates into this output" (deferred inside a DetDefer scope if it is a bucket)
  explicit DetGemmTarget(const void* c);
  ~DetGemmTarget();
  bool prev_;
};
struct DetDefer {                             // `r`: the opening handle's bucket ranges (mimrl_handle::det_ranges, set at bind)
  DetDefer(hipStream_t s, const DetRanges* r);
  ~DetDefer();
  hipStream_t s_; bool prev_; const DetRanges* prev_r_;
};
int det_overflowed();                        // bit 0: the table ran full; bit 1: a non-finite / out-of-range contribution (both: that
                                             // contribution went through a plain float atomic)
namespace {
struct DetTuReg {
  DetTuReg() { det_register_tu([](const DetCtx& c) { (void)hipMemcpyToSymbol(HIP_SYMBOL(g_det), &c, sizeof(DetCtx)); }); }
};
static DetTuReg det_tu_reg_;
}  // namespace

#else   // ---------------------------------------------------------------- default build: plain float atomics
struct DetNoFlush { explicit DetNoFlush(bool) {} };
struct DetDefer { DetDefer(hipStream_t, const DetRanges*) {} };
struct DetGemmTarget { explicit DetGemmTarget(const void*) {} };
inline bool det_target_in_bucket(const void*) { return false; }

__device__ __forceinline__ void acc_add(float* p, float v) { atomicAdd(p, v); }
struct LdsAcc {
  float v;
  __device__ __forceinline__ void zero() { v = 0.f; }
  __device__ __forceinline__ void add(float x) { atomicAdd(&v, x); }
  __device__ __forceinline__ float get() const { return v; }
};
#endif

}  // namespace mimrl
