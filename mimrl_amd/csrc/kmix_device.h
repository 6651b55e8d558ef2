// Device helpers shared by model_ops.hip and cube_fused.hip: LayerNorm epsilon, counter-hash dropout, and the K-axis
// mix (tiny MLP + residual + LayerNorm over K) evaluated per (row, d) thread with compile-time array sizes.
#pragma once
#include "model_ops.h"

namespace mimrl {
namespace {

constexpr float LN_EPS = 1e-6f;   // every LayerNorm of the model uses eps=1e-6 (Model.py:260, MLPProcess.py:35-41)

__device__ __forceinline__ float drop_scale(float p, const RngKey& key, uint32_t stream, uint32_t idx) {
  if (p <= 0.f) return 1.f;
  const float u = uniform01(key.seed_lo, key.seed_hi, stream, (uint32_t)(*key.step + key.add), idx);
  return u >= p ? 1.f / (1.f - p) : 0.f;
}
// the same with the step counter already in a register: in a loop that also stores, the compiler must re-read *key.step for every
// call (the stores may alias it) -- one dependent memory round trip per row in tail_pre_kernel (13 us for 50 rows)
__device__ __forceinline__ float drop_scale_at(float p, const RngKey& key, uint32_t step, uint32_t stream, uint32_t idx) {
  if (p <= 0.f) return 1.f;
  const float u = uniform01(key.seed_lo, key.seed_hi, stream, step, idx);
  return u >= p ? 1.f / (1.f - p) : 0.f;
}


constexpr int KM = 8;   // max size of any K-axis dimension

template <int NK>
struct KMixVals {
  float x[NK], xn[NK], u[NK], h[NK], y[NK], xh[NK], sc[NK];   // sc: dropout scale of the MLP branch per output
  float mu, rs;
};

template <int NK>
__device__ __forceinline__ void ln_small(const float* v, int n, const float* g, const float* be, float* out, float* xh,
                                         float& mu, float& rs) {
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NK; ++i) s += i < n ? v[i] : 0.f;
  mu = s / n;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NK; ++i) { const float c = i < n ? v[i] - mu : 0.f; q += c * c; }
  rs = rsqrtf(q / n + LN_EPS);
#pragma unroll
  for (int i = 0; i < NK; ++i) { xh[i] = i < n ? (v[i] - mu) * rs : 0.f; out[i] = xh[i] * g[i] + be[i]; }
}

// sw: LDS copy of [w1 | b1 | w2 | b2 | wr | g | be] with row stride KM (missing pieces zero / identity)
template <int NK, int ACT = -1>   // ACT >= 0: activation known at compile time (common.h, act_dispatch)
__device__ __forceinline__ void kmix_forward_vals(const KMixW& w, const float* sw, KMixVals<NK>& v) {
  const float* w1 = sw; const float* b1 = w1 + KM * KM; const float* w2 = b1 + KM; const float* b2 = w2 + KM * KM;
  const float* wr = b2 + KM; const float* g = wr + KM * KM; const float* be = g + KM;
  if (w.ln_first) ln_small<NK>(v.x, w.ik, g, be, v.xn, v.xh, v.mu, v.rs);
#pragma unroll
  for (int j = 0; j < NK; ++j) {
    float s = b1[j];
#pragma unroll
    for (int k = 0; k < NK; ++k) s += w1[j * KM + k] * (w.ln_first ? v.xn[k] : v.x[k]);   // padded weights are zero
    v.u[j] = s; v.h[j] = j < w.hk ? act_apply_c<ACT>(w.act, s) : 0.f;
  }
#pragma unroll
  for (int o = 0; o < NK; ++o) {
    float s = b2[o], rr = 0.f;
#pragma unroll
    for (int j = 0; j < NK; ++j) s += w2[o * KM + j] * v.h[j];
#pragma unroll
    for (int k = 0; k < NK; ++k) rr += wr[o * KM + k] * v.x[k];
    v.y[o] = v.sc[o] * s + rr;
  }
}

// The same mix with the (padded) weights held in REGISTERS: a caller whose loop also stores to LDS (the fused CubeMLP forward writes its
// tile in place) would otherwise re-read all ~60 weight words from LDS for every (l, d) pair -- the stores may alias them
template <int NK>
struct KMixRegs {
  float w1[NK][NK], b1[NK], w2[NK][NK], b2[NK], wr[NK][NK], g[NK], be[NK];
  __device__ __forceinline__ void load(const float* sw) {
    const float* pw1 = sw; const float* pb1 = pw1 + KM * KM; const float* pw2 = pb1 + KM; const float* pb2 = pw2 + KM * KM;
    const float* pwr = pb2 + KM; const float* pg = pwr + KM * KM; const float* pbe = pg + KM;
#pragma unroll
    for (int i = 0; i < NK; ++i) {
      b1[i] = pb1[i]; b2[i] = pb2[i]; g[i] = pg[i]; be[i] = pbe[i];
#pragma unroll
      for (int j = 0; j < NK; ++j) { w1[i][j] = pw1[i * KM + j]; w2[i][j] = pw2[i * KM + j]; wr[i][j] = pwr[i * KM + j]; }
    }
  }
};
// (the ln_first = false, NK = ik = hk = ok form only: the fused forward kernel's case)
template <int NK, int ACT = -1>
__device__ __forceinline__ void kmix_forward_regs(const KMixW& w, const KMixRegs<NK>& q, KMixVals<NK>& v) {
#pragma unroll
  for (int j = 0; j < NK; ++j) {
    float s = q.b1[j];
#pragma unroll
    for (int k = 0; k < NK; ++k) s += q.w1[j][k] * v.x[k];
    v.u[j] = s; v.h[j] = act_apply_c<ACT>(w.act, s);
  }
#pragma unroll
  for (int o = 0; o < NK; ++o) {
    float s = q.b2[o], rr = 0.f;
#pragma unroll
    for (int j = 0; j < NK; ++j) s += q.w2[o][j] * v.h[j];
#pragma unroll
    for (int k = 0; k < NK; ++k) rr += q.wr[o][k] * v.x[k];
    v.y[o] = v.sc[o] * s + rr;
  }
}

__device__ __forceinline__ void kmix_stage_weights(const KMixW& w, float* sw) {
  for (int i = threadIdx.x; i < 3 * KM * KM + 4 * KM; i += blockDim.x) sw[i] = 0.f;
  __syncthreads();
  float* w1 = sw; float* b1 = w1 + KM * KM; float* w2 = b1 + KM; float* b2 = w2 + KM * KM;
  float* wr = b2 + KM; float* g = wr + KM * KM; float* be = g + KM;
  const int t = threadIdx.x;
  if (t < w.hk * w.ik) w1[(t / w.ik) * KM + t % w.ik] = w.w1[t];
  if (t < w.hk && w.b1) b1[t] = w.b1[t];
  if (t < w.ok * w.hk) w2[(t / w.hk) * KM + t % w.hk] = w.w2[t];
  if (t < w.ok && w.b2) b2[t] = w.b2[t];
  if (t < w.ok * w.ik) wr[(t / w.ik) * KM + t % w.ik] = w.wr ? w.wr[t] : ((t / w.ik) == (t % w.ik) ? 1.f : 0.f);
  const int nln = w.ln_first ? w.ik : w.ok;
  if (t < nln) { g[t] = w.g[t]; be[t] = w.be[t]; }
  __syncthreads();
}


}  // namespace
}  // namespace mimrl
