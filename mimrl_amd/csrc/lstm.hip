// See lstm.h.
#include "lstm.h"

namespace mimrl {

namespace {

constexpr int H = LSTM_H, G4 = 4 * LSTM_H;

__device__ __forceinline__ float sigm(float x) { return 1.f / (1.f + __expf(-x)); }

__global__ __launch_bounds__(512) void lstm_fwd_kernel(LstmFwdArgs a) {
  __shared__ __attribute__((aligned(16))) float sh[H];
  __shared__ float sg[G4];
  const int b = blockIdx.x, dir = blockIdx.y, mod = blockIdx.z, j = threadIdx.x;
  const LstmSeq& q = a.seq[mod][dir];
  const int T = a.T, len = a.lens[mod][b];
  float wrow[H];
#pragma unroll
  for (int k = 0; k < H; k += 4) {
    const float4 v = *reinterpret_cast<const float4*>(q.w_hh + (long)j * H + k);
    wrow[k] = v.x; wrow[k + 1] = v.y; wrow[k + 2] = v.z; wrow[k + 3] = v.w;
  }
  const float bh = q.b_hh[j];
  float c = 0.f;
  if (j < H) sh[j] = 0.f;
  __syncthreads();
  const float* gx_b = q.gx + (long)b * T * G4;
  float* out_b = q.out + (long)b * T * a.out_ld + dir * H;
  float* sv_b = q.saved ? q.saved + (long)b * T * 6 * H : nullptr;
  for (int step = 0; step < T; ++step) {
    const int t = dir ? T - 1 - step : step;
    if (t >= len) {                                   // uniform: one sample per workgroup
      if (j < H) out_b[(long)t * a.out_ld + j] = 0.f;
      continue;
    }
    float pre = gx_b[(long)t * G4 + j] + bh;
#pragma unroll
    for (int k = 0; k < H; k += 4) {
      const float4 hv = *reinterpret_cast<const float4*>(&sh[k]);
      pre += wrow[k] * hv.x + wrow[k + 1] * hv.y + wrow[k + 2] * hv.z + wrow[k + 3] * hv.w;
    }
    sg[j] = (j >= 2 * H && j < 3 * H) ? tanhf(pre) : sigm(pre);
    __syncthreads();
    if (j < H) {
      const float ig = sg[j], fg = sg[H + j], gg = sg[2 * H + j], og = sg[3 * H + j];
      const float cp = c;
      c = fg * c + ig * gg;
      const float tc = tanhf(c);
      const float hn = og * tc;
      sh[j] = hn;
      out_b[(long)t * a.out_ld + j] = hn;
      if (sv_b) {
        float* s = sv_b + (long)t * 6 * H;
        s[j] = ig; s[H + j] = fg; s[2 * H + j] = gg; s[3 * H + j] = og; s[4 * H + j] = cp; s[5 * H + j] = tc;
      }
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(512) void lstm_bwd_kernel(LstmBwdArgs a) {
  __shared__ float sdg[G4];
  __shared__ float red[4][H];
  const int b = blockIdx.x, dir = blockIdx.y, mod = blockIdx.z, tid = threadIdx.x;
  const int k = tid & (H - 1), part = tid >> 7;
  const LstmSeqBwd& q = a.seq[mod][dir];
  const int T = a.T, len = a.lens[mod][b];
  float wt[H];                                         // W_hh[part*128 + jj][k]
#pragma unroll
  for (int jj = 0; jj < H; ++jj) wt[jj] = q.w_hh[(long)(part * H + jj) * H + k];
  float dh_c = 0.f, dc_c = 0.f;
  const float* sv_b = q.saved + (long)b * T * 6 * H;
  const float* out_b = q.out + (long)b * T * a.out_ld + dir * H;
  const float* dout_b = q.dout + (long)b * T * a.dout_ld;
  float* dg_b = q.dg + (long)b * T * G4;
  float* hp_b = q.hprev + (long)b * T * H;
  for (int step = 0; step < T; ++step) {
    const int t = dir ? step : T - 1 - step;           // reverse of the forward visiting order
    if (t >= len) {
      dg_b[(long)t * G4 + tid] = 0.f;
      if (tid < H) hp_b[(long)t * H + tid] = 0.f;
      continue;
    }
    if (part == 0) {
      const float* s = sv_b + (long)t * 6 * H;
      const float ig = s[k], fg = s[H + k], gg = s[2 * H + k], og = s[3 * H + k], cp = s[4 * H + k], tc = s[5 * H + k];
      const float dh = dout_b[(long)t * a.dout_ld + k] + dh_c;
      const float dc = dc_c + dh * og * (1.f - tc * tc);
      const float dpi = dc * gg * ig * (1.f - ig);
      const float dpf = dc * cp * fg * (1.f - fg);
      const float dpg = dc * ig * (1.f - gg * gg);
      const float dpo = dh * tc * og * (1.f - og);
      dc_c = dc * fg;
      sdg[k] = dpi; sdg[H + k] = dpf; sdg[2 * H + k] = dpg; sdg[3 * H + k] = dpo;
      float* d = dg_b + (long)t * G4;
      d[k] = dpi; d[H + k] = dpf; d[2 * H + k] = dpg; d[3 * H + k] = dpo;
      const int tp = dir ? t + 1 : t - 1;
      hp_b[(long)t * H + k] = (tp >= 0 && tp < len) ? out_b[(long)tp * a.out_ld + k] : 0.f;
    }
    __syncthreads();
    float acc = 0.f;
#pragma unroll
    for (int jj = 0; jj < H; ++jj) acc += wt[jj] * sdg[part * H + jj];
    red[part][k] = acc;
    __syncthreads();
    if (part == 0) dh_c = red[0][k] + red[1][k] + red[2][k] + red[3][k];
    __syncthreads();
  }
}

}  // namespace

int lstm_forward(hipStream_t s, const LstmFwdArgs& a) {
  if (a.B <= 0 || a.T <= 0) return set_error(MIMRL_ERR_ARG, "lstm_forward: empty batch");
  hipLaunchKernelGGL(lstm_fwd_kernel, dim3(a.B, 2, a.nmod), dim3(512), 0, s, a);
  LAUNCH_CHECK();
  return MIMRL_OK;
}

int lstm_backward(hipStream_t s, const LstmBwdArgs& a) {
  if (a.B <= 0 || a.T <= 0) return set_error(MIMRL_ERR_ARG, "lstm_backward: empty batch");
  hipLaunchKernelGGL(lstm_bwd_kernel, dim3(a.B, 2, a.nmod), dim3(512), 0, s, a);
  LAUNCH_CHECK();
  return MIMRL_OK;
}

}  // namespace mimrl
