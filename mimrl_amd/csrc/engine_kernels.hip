// The step's own small kernels (see engine.h): stage bookkeeping, MAE, the two loss finalisers, the fused stage boundary.
#include "engine.h"
#include "stage_boundary.h"

namespace mimrl {

namespace {

__global__ void begin_stage_kernel(int* rng_step, int* adam_step, float* scalars, int scal_off, int scal_n) {
  if (threadIdx.x == 0) {
    *rng_step += 1;
    if (adam_step) *adam_step += 1;
  }
  for (int i = threadIdx.x; i < scal_n; i += blockDim.x) scalars[scal_off + i] = 0.f;
}

// MAE (nn.L1Loss, Solver.py:181-182) + its gradient
__global__ void mae_kernel(const float* __restrict__ pred, const float* __restrict__ y, float* __restrict__ dpred,
                           float* __restrict__ task, int B) {
  __shared__ float red[16];
  float s = 0.f;
  for (int b = threadIdx.x; b < B; b += blockDim.x) {
    const float d = pred[b] - y[b];
    s += fabsf(d);
    if (dpred) dpred[b] = (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) / B;
  }
  s = block_sum(s, red);
  if (threadIdx.x == 0) *task = s / B;
}

// Model.py:341 + Customization.py:100-102
__global__ void finalize_stage1_kernel(float* scal, const float* mi, const float* cmi, const float* bce, const float* coef1) {
  if (threadIdx.x != 0) return;
  float loss = 0.f;
  for (int e = 0; e < NE_MI; ++e) {
    scal[MIMRL_S1_MIS + e] = mi[e];
    scal[MIMRL_S1_LOSSES + e] = mi[NE_MI + e];        // mi_loss (= -mi except for the `mine` bound)
    loss += coef1[e] * mi[NE_MI + e];
  }
  for (int e = 0; e < NE_CMI; ++e) {
    scal[MIMRL_S1_MIS + NE_MI + e] = cmi[e];
    scal[MIMRL_S1_LOSSES + NE_MI + e] = bce[e];
    loss += coef1[NE_MI + e] * bce[e];
  }
  scal[MIMRL_S1_LOSS] = loss;
}
// Model.py:357,381-386 + Customization.py:109-111
__global__ void finalize_stage2_kernel(float* scal, const float* mi, const float* cmi, const float* coef2, int have_mi) {
  if (threadIdx.x != 0) return;
  const float task = scal[MIMRL_S2_TASK];
  if (!have_mi) {
    for (int i = 0; i < 8; ++i) { scal[MIMRL_S2_MIS + i] = 0.f; scal[MIMRL_S2_LOSSES + i] = 0.f; }
    scal[MIMRL_S2_LOSS] = task;
    return;
  }
  const float ac_t = cmi[0], ta_c = cmi[1], vc_t = cmi[2], tv_c = cmi[3], tc_a = cmi[4], tc_v = cmi[5];
  float v[8];
  v[0] = mi[0]; v[1] = mi[1]; v[2] = mi[2];
  v[3] = mi[3] + mi[4];
  v[4] = tc_a + tc_v - ta_c - tv_c;
  v[5] = ac_t - ta_c;
  v[6] = vc_t - tv_c;
  v[7] = ta_c + tv_c;
  float loss = task;
  for (int i = 0; i < 8; ++i) {
    const float li = i < 3 ? mi[NE_MI + i] : -v[i];   // Model.py:386: f_t, f_a, f_v through their mi_loss, the rest through -value
    scal[MIMRL_S2_MIS + i] = v[i];
    scal[MIMRL_S2_LOSSES + i] = li;
    loss += coef2[i] * li;
  }
  scal[MIMRL_S2_LOSS] = loss;
}

// Stage boundary of a combined two-stage step as ONE launch: finalize_stage1 (Model.py:341) + begin_stage(2) + MAE (Solver.py:181-182).
// Behind the critic Adam; the stage-1 raw terms are still intact (stage 2's estimators overwrite them later).  (Body: stage_boundary.h --
// by default it rides on the critic Adam launch instead, estimator_ops.hip: adam8_kernel.)
static_assert(SB_NE_MI == NE_MI && SB_NE_CMI == NE_CMI, "stage_boundary.h restates the estimator counts");
__global__ void stage_boundary_kernel(StageBoundaryArgs a) {
  __shared__ float red[16];
  stage_boundary_body(a, red);
}

}  // namespace

namespace eng {

// fp32 <-> bf16 images of a gradient bucket around its bf16 all-reduce (mimrl_set_comm_critic_bf16); n a multiple of 4 (buckets are)
__global__ void bucket_to_bf16_kernel(const float* __restrict__ src, __bf16* __restrict__ dst, long n4) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const float4 v = reinterpret_cast<const float4*>(src)[i];
    bf16x4 p; p[0] = to_bf16(v.x); p[1] = to_bf16(v.y); p[2] = to_bf16(v.z); p[3] = to_bf16(v.w);
    reinterpret_cast<bf16x4*>(dst)[i] = p;
  }
}
__global__ void bucket_from_bf16_kernel(const __bf16* __restrict__ src, float* __restrict__ dst, long n4) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const bf16x4 p = reinterpret_cast<const bf16x4*>(src)[i];
    reinterpret_cast<float4*>(dst)[i] = make_float4((float)p[0], (float)p[1], (float)p[2], (float)p[3]);
  }
}
void launch_bucket_to_bf16(hipStream_t s, const float* src, void* dst, long n) {
  hipLaunchKernelGGL(bucket_to_bf16_kernel, dim3((unsigned)std::min<long>((n / 4 + 255) / 256, 2048)), dim3(256), 0, s, src, static_cast<__bf16*>(dst), n / 4);
}
void launch_bucket_from_bf16(hipStream_t s, const void* src, float* dst, long n) {
  hipLaunchKernelGGL(bucket_from_bf16_kernel, dim3((unsigned)std::min<long>((n / 4 + 255) / 256, 2048)), dim3(256), 0, s, static_cast<const __bf16*>(src), dst, n / 4);
}

void launch_begin_stage(hipStream_t s, int* rng_step, int* adam_step, float* scalars, int scal_off, int scal_n) {
  hipLaunchKernelGGL(begin_stage_kernel, dim3(1), dim3(64), 0, s, rng_step, adam_step, scalars, scal_off, scal_n);
}
void launch_mae(hipStream_t s, const float* pred, const float* y, float* dpred, float* task, int B) {
  hipLaunchKernelGGL(mae_kernel, dim3(1), dim3(256), 0, s, pred, y, dpred, task, B);
}
void launch_finalize_stage1(hipStream_t s, float* scal, const float* mi, const float* cmi, const float* bce, const float* coef1) {
  hipLaunchKernelGGL(finalize_stage1_kernel, dim3(1), dim3(64), 0, s, scal, mi, cmi, bce, coef1);
}
void launch_finalize_stage2(hipStream_t s, float* scal, const float* mi, const float* cmi, const float* coef2, int have_mi) {
  hipLaunchKernelGGL(finalize_stage2_kernel, dim3(1), dim3(64), 0, s, scal, mi, cmi, coef2, have_mi);
}
void launch_stage_boundary(hipStream_t s, float* scal, const float* mi, const float* cmi, const float* bce, const float* coef1, int* rng_step,
                           int* adam_step, const float* pred, const float* y, float* dpred, int B) {
  hipLaunchKernelGGL(stage_boundary_kernel, dim3(1), dim3(256), 0, s, StageBoundaryArgs{scal, mi, cmi, bce, coef1, rng_step, adam_step, pred, y, dpred, B});
}

}  // namespace eng

}  // namespace mimrl
