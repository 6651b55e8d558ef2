// Weight gradients of the concat critic's hidden layers, every operand row staged once (concat_dw.hip).
#pragma once
#include "common.h"

namespace mimrl {

struct ConcatDwArgs {
  const __bf16* dz[2];   // per layer: [E][rows, 256] gradient of the layer's pre-activation (bf16, concat_bwd_fused)
  const __bf16* act[2];  // per layer: [E][rows, 256] the layer's input activation (the forward kernel's bf16 copy)
  float* dw[2];          // per layer: [E] x dw_stride floats apart, [256, 256] row-major, accumulated
  long dw_stride;
  int nlayer;            // 1 or 2
  int E;                 // estimators
  long rows;             // B * B
  // optional third product in the same launch (extra workgroups): the score head's weight gradient dw3[e] [256] += ds[e]^T a2[e] with ds
  // [E][rows] fp32 and a2 [E][rows, 256] stored as fp16 (concat_fwd_a2_f16) -- as its own launch beside this one the two competed for the
  // CUs' intake and the slower of them closed stage 1's chain
  const float* ds = nullptr; const _Float16* a2 = nullptr; float* dw3 = nullptr;
  // optional: dz[0] is NOT read but regenerated (concat_dw.hip: GEN) from ds, the layer-2 sign words m2 [E][rows][8] (bit c of word g = sign of
  // column 32 g + c) and the score head's weight w3 (estimator e at + e * dw_stride); needs ds
  const uint32_t* m2 = nullptr; const float* w3 = nullptr;
  // optional: act[1] (a0) is NOT read but regenerated as bf16(relu(P[i] + Q[j])) for pair row i B + j from the separable first layer's two
  // projections P, Q [E][B][256] fp32 (ConcatFwdArgs); rows must be B * B, B a multiple of 32
  const float* P = nullptr; const float* Q = nullptr; int B = 0;
  int nsplit_l[2] = {0, 0}, kt_per_l[2] = {0, 0}, n3 = 0; long rows3 = 0;   // filled by concat_dw(): k-ranges per estimator of each layer, k-tiles each; dw3 workgroups per estimator, rows each
};
bool concat_dw_ok(int E, long rows, int hid);
int concat_dw(hipStream_t s, const ConcatDwArgs& a);

}  // namespace mimrl
