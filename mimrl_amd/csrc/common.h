// Shared device/host helpers for libmimrl_hip (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string>

#include "../../include/mimrl.h"

namespace mimrl {

// ----------------------------------------------------------------------------------------------
// error plumbing: nothing throws across the C ABI; every launcher returns 0 or a negative code and
// leaves a message in a thread-local string that mimrl_last_error() hands out.
// ----------------------------------------------------------------------------------------------
std::string& last_error_slot();
int set_error(int code, const char* fmt, ...);

#define HIPX(expr)                                                                                  \
  do {                                                                                              \
    hipError_t _e = (expr);                                                                         \
    if (_e != hipSuccess)                                                                           \
      return ::mimrl::set_error(MIMRL_ERR_HIP, "%s:%d %s -> %s", __FILE__, __LINE__, #expr, \
                                hipGetErrorString(_e));                                             \
  } while (0)

#define MX(expr)            \
  do {                      \
    int _r = (expr);        \
    if (_r != 0) return _r; \
  } while (0)

#define LAUNCH_CHECK() HIPX(hipGetLastError())
}  // namespace mimrl
#include "det.h"
#include "knobs.h"
#undef hipLaunchKernelGGL
#ifdef MIMRL_DET
// deterministic build: a launch that may have accumulated (acc_add) is followed, on its own stream, by the flush of the fixed-point
// accumulation table (det.h).  Round 5: not EVERY launch any more -- 92 flushes of ~7 us per cfg2 step were a quarter of the build's step
// time; det_launch_accumulates() knows the kernels that never call acc_add by name, and a host scope (DetNoFlush) says so for a template
// instantiation whose name does not (a GEMM without atomics / column sums).
#define hipLaunchKernelGGL(kernel, grid, block, lds, stream, ...)                 \
  do {                                                                            \
    (void)::mimrl::det_init();                                                    \
    hipLaunchKernelGGLInternal((kernel), (grid), (block), (lds), (stream), __VA_ARGS__); \
    if (::mimrl::det_launch_accumulates(#kernel)) (void)::mimrl::det_flush(stream); \
  } while (0)
#else
#define hipLaunchKernelGGL(kernel, grid, block, lds, stream, ...)                 \
  do {                                                                            \
    hipLaunchKernelGGLInternal((kernel), (grid), (block), (lds), (stream), __VA_ARGS__); \
  } while (0)
#endif
namespace mimrl {

// Environment knobs.  Tuning knobs (stream placement, kernel variants: results unchanged) are read with knob() (knobs.h: one table) where they
// are used.  DEBUG knobs change RESULTS (skip work, stop a kernel early, re-create a placement known to miscompute): they exist
// only in a -DMIMRL_DEBUG_KNOBS build (`make DEBUG_KNOBS=1`); in the default build dbg_env() is a constant nullptr -- the
// branches behind it fold away -- and mimrl_create() refuses to run while one of them is set in the environment.
#ifdef MIMRL_DEBUG_KNOBS
inline const char* dbg_env(const char* name) { return getenv(name); }
#else
inline const char* dbg_env(const char*) { return nullptr; }
#endif
// names of every result-changing knob (errors.cpp); mimrl_create checks them
extern const char* const kDebugKnobs[];

// ----------------------------------------------------------------------------------------------
// device math
// ----------------------------------------------------------------------------------------------
enum Act : int { ACT_NONE = 0, ACT_RELU = 1, ACT_GELU = 2, ACT_TANH = 3 };

// erf via Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, i.e. fp32 noise) on v_exp/v_rcp: ~15 instructions instead of
// libm erff's ~60 -- the exact-erf GELU (F.gelu default, MLPProcess.py:14 / Utils.py:86) is the dominant VALU cost
// of the CubeMLP kernels.
__device__ __forceinline__ float fast_erf(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * ax);
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float e = 1.0f - poly * __builtin_amdgcn_exp2f(-1.44269504088896340736f * ax * ax);
  return copysignf(e, x);
}
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + fast_erf(x * 0.70710678118654752440f)); }
// gelu'(x) = Phi(x) + x phi(x).  The exponential inside fast_erf(x / sqrt 2) IS exp(-x^2 / 2), the density's: one v_exp_f32 serves both
// (the backward epilogues of the CubeMLP kernels evaluate this 16-32 times per lane and are VALU-bound, tools/cube_phase.py)
__device__ __forceinline__ float gelu_grad_f(float x) {
  const float ax = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * ax);
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float ex = __builtin_amdgcn_exp2f(-1.44269504088896340736f * ax * ax);     // exp(-x^2 / 2)
  const float erfv = copysignf(1.0f - poly * ex, x);
  return 0.5f * (1.0f + erfv) + x * (0.39894228040143267794f * ex);
}
__device__ __forceinline__ float act_apply(int act, float x) {
  switch (act) {
    case ACT_RELU: return x > 0.f ? x : 0.f;
    case ACT_GELU: return gelu_f(x);
    case ACT_TANH: return tanhf(x);
    default: return x;
  }
}
// derivative w.r.t. the PRE-activation value u
__device__ __forceinline__ float act_grad(int act, float u) {
  switch (act) {
    case ACT_RELU: return u > 0.f ? 1.f : 0.f;
    case ACT_GELU: return gelu_grad_f(u);
    case ACT_TANH: { float t = tanhf(u); return 1.f - t * t; }
    default: return 1.f;
  }
}
// Activation known at compile time (ACT >= 0) or not (ACT < 0: the run-time switch).  Round 3b: an epilogue that calls act_apply(act, u) per
// element carries the whole switch -- three activations, ~13 scalar branches -- in every unrolled iteration; the fused CubeMLP forward spent
// ~200 cycles per element there (7.9 us of block 1's 50).  act_dispatch() branches ONCE around the loop: GELU (the model's activation) gets
// its own straight-line copy, everything else the generic one.
template <int ACT>
__device__ __forceinline__ float act_apply_c(int act, float x) {
  if constexpr (ACT == ACT_GELU) return gelu_f(x);
  else return act_apply(act, x);
}
template <int ACT>
__device__ __forceinline__ float act_grad_c(int act, float u) {
  if constexpr (ACT == ACT_GELU) return gelu_grad_f(u);
  else return act_grad(act, u);
}
template <int V> struct ActTag { static constexpr int value = V; };
template <class F>
__device__ __forceinline__ void act_dispatch(int act, F&& f) {
  if (act == ACT_GELU) f(ActTag<ACT_GELU>{}); else f(ActTag<-1>{});
}
__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + __expf(-x)); }
// hardware-rate versions for the recurrent gate math: v_exp_f32 + v_rcp_f32 (~1 ulp each), no IEEE division, no libm
__device__ __forceinline__ float fast_sigmoid(float x) {
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896340736f * x));
}
__device__ __forceinline__ float fast_tanh(float x) {   // 1 - 2/(exp(2x)+1); saturates cleanly to +-1
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.88539008177792681472f * x));
}

// Wave-wide reductions on the VALU (DPP lane permutes), result in every lane.  __shfl_xor is ds_bpermute on gfx9 -- an LDS round trip
// per step, six dependent ones per reduction; this is 6 DPP adds + one v_readlane.  All 64 lanes must be active (every caller reduces
// under wave-uniform control flow).  Lanes combine as: quads, 8s, 16-lane rows, then row_bcast:15 / row_bcast:31 carry the row sums
// into row 3, whose last lane holds the total.
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ float dpp_take(float old, float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float wave_sum(float v) {
#ifdef MIMRL_WAVE_SHFL
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
#endif
  v += dpp_take<0xB1>(0.f, v);          // quad_perm:[1,0,3,2]
  v += dpp_take<0x4E>(0.f, v);          // quad_perm:[2,3,0,1]
  v += dpp_take<0x141>(0.f, v);         // row_half_mirror
  v += dpp_take<0x140>(0.f, v);         // row_mirror
  v += dpp_take<0x142, 0xA>(0.f, v);    // row_bcast:15 into rows 1 and 3
  v += dpp_take<0x143, 0xC>(0.f, v);    // row_bcast:31 into rows 2 and 3
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
// sum over aligned groups of N = 2, 4, 8 or 16 lanes, result in every lane of the group
template <int N>
__device__ __forceinline__ float group_sum(float v) {
  static_assert(N == 2 || N == 4 || N == 8 || N == 16, "group_sum: 2, 4, 8 or 16 lanes");
  v += dpp_take<0xB1>(0.f, v);
  if (N >= 4) v += dpp_take<0x4E>(0.f, v);
  if (N >= 8) v += dpp_take<0x141>(0.f, v);
  if (N >= 16) v += dpp_take<0x140>(0.f, v);
  return v;
}
// sum over each half of the wave (the 32 lanes that share lane >> 5); the result is valid in the UPPER 16 lanes of each half only
// (lanes 16..31 and 48..63)
__device__ __forceinline__ float half_sum_hi(float v) {
  v = group_sum<16>(v);
  v += dpp_take<0x142, 0xA>(0.f, v);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#ifdef MIMRL_WAVE_SHFL
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
#endif
  v = fmaxf(v, dpp_take<0xB1>(v, v));
  v = fmaxf(v, dpp_take<0x4E>(v, v));
  v = fmaxf(v, dpp_take<0x141>(v, v));
  v = fmaxf(v, dpp_take<0x140>(v, v));
  v = fmaxf(v, dpp_take<0x142, 0xA>(v, v));
  v = fmaxf(v, dpp_take<0x143, 0xC>(v, v));
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// (butterfly version through ds_bpermute: kept as the cross-check of tools/dpp_check.hip; -DMIMRL_WAVE_SHFL builds everything with it)
__device__ __forceinline__ float wave_sum_shfl(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// block-wide sum for blockDim.x <= 1024 (multiple of 64); `red` = >=16 floats of LDS
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if (l == 0) red[w] = v;
  __syncthreads();
  float r = (l < nw) ? red[l] : 0.f;
  r = wave_sum(r);
  return r;
}
__device__ __forceinline__ float block_max(float v, float* red) {
  v = wave_max(v);
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if (l == 0) red[w] = v;
  __syncthreads();
  float r = (l < nw) ? red[l] : -INFINITY;
  r = wave_max(r);
  return r;
}

// dst[b, :] = sum_q src[q][b * ld[q] + off[q] + :]  over the sources whose row count covers b  (feature-gradient routing)
struct GatherSum { const float* src[12]; int ld[12]; int off[12]; int rows[12]; int n; };

// counter-based RNG for dropout: one 32-bit hash per element, keyed by (seed, stream id, step, index).
// (The reference's torch Philox stream cannot be reproduced; parity runs use p=0 or explicit masks.)
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
__device__ __forceinline__ float uniform01(uint32_t seed_lo, uint32_t seed_hi, uint32_t stream, uint32_t step,
                                           uint32_t idx) {
  uint32_t h = mix32(idx ^ mix32(step * 0x9E3779B9U + stream) ^ seed_lo);
  h = mix32(h + seed_hi * 0x85ebca6bU + 0x632be5abU);
  return (h >> 8) * (1.0f / 16777216.0f);
}

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;

__device__ __forceinline__ __bf16 to_bf16(float x) { return (__bf16)x; }  // v_cvt_pk_bf16_f32 (RNE, NaN-safe)
// fp16 MFMA operands (same rate as bf16, 11 instead of 8 significant bits) for FORWARD operands of bounded range (LayerNorm'd
// activations, weights): see cube_fused.hip.  Gradients stay bf16 (range).  v_cvt_f16_f32 rounds to nearest even, overflow -> inf.
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
__device__ __forceinline__ _Float16 to_f16(float x) { return (_Float16)x; }
__device__ __forceinline__ _Float16 to_f16_sat(float x) { return (_Float16)fminf(fmaxf(x, -65504.f), 65504.f); }

// compute units of the current device (256 on MI355X); cached
inline int device_cus() {
  static int cus = 0;
  if (!cus) {
    hipDeviceProp_t pr;
    int dev = 0;
    cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) ? pr.multiProcessorCount : 256;
  }
  return cus;
}

}  // namespace mimrl
