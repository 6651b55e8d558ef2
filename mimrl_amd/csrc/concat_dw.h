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
  int nsplit = 0, kt_per = 0, n3 = 0; long rows3 = 0;   // filled by concat_dw(): k-ranges per (layer, estimator); dw3 workgroups per estimator, rows each
};
bool concat_dw_ok(int E, long rows, int hid);
int concat_dw(hipStream_t s, const ConcatDwArgs& a);

}  // namespace mimrl
