// The stage boundary of a combined two-stage step (finalize_stage1, Model.py:341 + Customization.py:100-102; begin_stage(2); MAE,
// Solver.py:181-182) as a device function: its own one-workgroup launch (engine_kernels.hip: stage_boundary_kernel) or, round 5b, the
// first thing workgroup 0 of the critic clip + Adam launch does (estimator_ops.hip: adam8_kernel) -- one launch and one dependent-launch
// gap less between the stage-1 critic update and the stage-2 estimators.
#pragma once
#include "common.h"

namespace mimrl {

constexpr int SB_NE_MI = 5, SB_NE_CMI = 6;   // = NE_MI, NE_CMI of engine.h (asserted in engine_kernels.hip)

struct StageBoundaryArgs {
  float* scal; const float *mi, *cmi, *bce, *coef1;
  int *rng_step, *adam_step;
  const float *pred, *y; float* dpred; int B;
};

// all threads of ONE workgroup (blockDim.x a multiple of 64, <= 1024); `red`: 16 floats of LDS
__device__ __forceinline__ void stage_boundary_body(const StageBoundaryArgs& a, float* red) {
  if (threadIdx.x == 0) {
    // every value is READ before the first store (tools/isa_lint.py: interleaved with the stores into `scal`, which may alias, these were
    // 21 loads each followed by s_waitcnt vmcnt(0) -- 21 round trips in a one-thread kernel on the step's critical path, 7 us)
    float vm[SB_NE_MI], vl[SB_NE_MI], vc[SB_NE_CMI], vb[SB_NE_CMI], c1[SB_NE_MI + SB_NE_CMI];
#pragma unroll
    for (int e = 0; e < SB_NE_MI; ++e) { vm[e] = a.mi[e]; vl[e] = a.mi[SB_NE_MI + e]; c1[e] = a.coef1[e]; }
#pragma unroll
    for (int e = 0; e < SB_NE_CMI; ++e) { vc[e] = a.cmi[e]; vb[e] = a.bce[e]; c1[SB_NE_MI + e] = a.coef1[SB_NE_MI + e]; }
    const int rs = *a.rng_step, as = *a.adam_step;
    float loss = 0.f;
#pragma unroll
    for (int e = 0; e < SB_NE_MI; ++e) {
      a.scal[MIMRL_S1_MIS + e] = vm[e];
      a.scal[MIMRL_S1_LOSSES + e] = vl[e];
      loss += c1[e] * vl[e];
    }
#pragma unroll
    for (int e = 0; e < SB_NE_CMI; ++e) {
      a.scal[MIMRL_S1_MIS + SB_NE_MI + e] = vc[e];
      a.scal[MIMRL_S1_LOSSES + SB_NE_MI + e] = vb[e];
      loss += c1[SB_NE_MI + e] * vb[e];
    }
    a.scal[MIMRL_S1_LOSS] = loss;
    *a.rng_step = rs + 1;
    *a.adam_step = as + 1;
  }
  for (int i = threadIdx.x; i < 32; i += blockDim.x) a.scal[32 + i] = 0.f;
  float s = 0.f;
  for (int b = threadIdx.x; b < a.B; b += blockDim.x) {
    const float d = a.pred[b] - a.y[b];
    s += fabsf(d);
    a.dpred[b] = (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) / a.B;
  }
  s = block_sum(s, red);
  __syncthreads();                       // the zeroing of scal[32..63] above is complete before the task loss lands in it
  if (threadIdx.x == 0) a.scal[MIMRL_S2_TASK] = s / a.B;
}

}  // namespace mimrl
