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


// =====================================================================================================================================
// Round 5: the LSTM recurrence on the matrix cores -- the bi-GRU kernels' design (gru.hip) with the LSTM cell (Model.py:250-252,441-447).
// One workgroup = up to 4 batch rows of one (modality, direction) for all T steps, 4 waves x 32 hidden units; every wave keeps its slice of
// W_hh -- all FOUR gates -- in registers as ready-made MFMA B fragments for the whole sequence (bf16: 128 VGPRs; fp32: 256, this kernel
// runs one wave per SIMD), the 4 x 128 state tile is double-buffered in LDS with ONE barrier per step, and the state is the A operand
// with every batch row replicated over 4 MFMA rows, so that accumulator register 0 of lane (n, kq) is the pre-activation of (batch row kq,
// unit 32 w + 2 n + s): i, f, g, o of one (unit, batch) pair in the same lane, the cell state c in a register.
//   forward:  32 products per step (bf16: v_mfma_f32_16x16x32_bf16, W_hh and h rounded to bf16 for the product; fp32: 16x16x4_f32, exact)
//   BPTT:     the dg tile [4, 4H] goes through LDS, carry = dg . W_hh: 32 products per step over k = 512
// Padding lanes mirror the last real row (no guarded access in the loop), positions t >= length emit 0 and do not advance the state.
// The saved record and the dg / h_prev outputs keep lstm.h's fp32 layouts (the weight-gradient GEMMs of the engine are unchanged).
// The scalar kernels above stay as the reference implementation the operator-level test compares this one with.
// =====================================================================================================================================
constexpr int G4M = 4 * LSTM_H;
#include "recur_device.h"

template <bool BF16>
struct LCfg;
template <> struct LCfg<true>  { static constexpr int KS_F = H / 32, KS_B = G4M / 32; using Frag = bf16x8; };
template <> struct LCfg<false> { static constexpr int KS_F = H / 4,  KS_B = G4M / 4;  using Frag = float; };

template <bool BF16, bool SAVE>
__global__ __launch_bounds__(256, 1) void lstm_fwd_mfma_kernel(LstmFwdArgs a, int btv) {
  using C = LCfg<BF16>;
  __shared__ __attribute__((aligned(16))) Tile<BF16, H> hs[2];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int tile = blockIdx.x, dir = blockIdx.y, mod = blockIdx.z;
  const LstmSeq& q = a.seq[mod][dir];
  const int B = a.B, T = a.T;
  const int n = lane & 15, kq = lane >> 4;
  const int u0 = 32 * w + 2 * n;
  const int b = min(tile * btv + min(kq, btv - 1), B - 1);      // padding lanes mirror the last real row of the tile
  const bool own = kq < btv && tile * btv + kq < B;             // ... and store nothing
  const int len = a.lens[mod][b];
  typename C::Frag wr[4][2][C::KS_F];
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      const float* row = q.w_hh + (long)(g * H + u0 + s2) * H;
#pragma unroll
      for (int ks = 0; ks < C::KS_F; ++ks) {
        if constexpr (BF16) {
          bf16x8 f;
#pragma unroll
          for (int j = 0; j < 8; ++j) f[j] = to_bf16(row[ks * 32 + 8 * kq + j]);
          wr[g][s2][ks] = f;
        } else {
          wr[g][s2][ks] = row[ks * 4 + kq];
        }
      }
    }
  float bh[4][2];
#pragma unroll
  for (int g = 0; g < 4; ++g) ldu<2>(q.b_hh + g * H + u0, bh[g]);
  float hreg[2] = {0.f, 0.f}, creg[2] = {0.f, 0.f};
  putu<2>(hs[0], kq, u0, hreg);
  __syncthreads();
  const float* gx_b = q.gx + (long)b * T * G4M + u0;
  float* out_b = q.out + (long)b * T * a.out_ld + dir * H + u0;
  float* sv_b = SAVE ? q.saved + (long)b * T * 6 * H + u0 : nullptr;
  float gxn[4][2];
  auto load_gx = [&](int step, float (&v)[4][2]) __attribute__((always_inline)) {
    const int sc = step < T ? step : T - 1;
    const int t = dir ? T - 1 - sc : sc;
#pragma unroll
    for (int g = 0; g < 4; ++g) ldu<2>(gx_b + (long)t * G4M + g * H, v[g]);
  };
  load_gx(0, gxn);
  for (int step = 0; step < T; ++step) {
    const int t = dir ? T - 1 - step : step, cur = step & 1;
    const bool valid = t < len;
    float gxc[4][2];
#pragma unroll
    for (int g = 0; g < 4; ++g) { gxc[g][0] = gxn[g][0]; gxc[g][1] = gxn[g][1]; }
    load_gx(step + 1, gxn);                                  // next step's inputs are in flight under this step's products
    f32x4 acc[4][2];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) acc[g][s2] = f32x4{gxc[g][s2] + bh[g][s2], 0.f, 0.f, 0.f};   // only register 0 is read
#pragma unroll
    for (int ks = 0; ks < C::KS_F; ++ks) {
      const auto sf = state_frag(hs[cur], ks, lane);
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) acc[g][s2] = mfma16(sf, wr[g][s2][ks], acc[g][s2]);
    }
    float hout[2];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      const float ig = sigm(acc[0][s2][0]), fg = sigm(acc[1][s2][0]), gg = tanhf(acc[2][s2][0]), og = sigm(acc[3][s2][0]);
      const float cp = creg[s2];
      const float cn = fg * cp + ig * gg;
      const float tc = tanhf(cn);
      const float hn = og * tc;
      if (valid) { creg[s2] = cn; hreg[s2] = hn; }
      hout[s2] = valid ? hn : 0.f;
      if constexpr (SAVE) {
        if (own && valid) {
          float* sp = sv_b + (long)t * 6 * H + s2;
          sp[0] = ig; sp[H] = fg; sp[2 * H] = gg; sp[3 * H] = og; sp[4 * H] = cp; sp[5 * H] = tc;
        }
      }
    }
    if (own) stu<2>(out_b + (long)t * a.out_ld, hout);
    putu<2>(hs[cur ^ 1], kq, u0, hreg);
    lds_barrier();
  }
}

template <bool BF16>
__global__ __launch_bounds__(256, 1) void lstm_bwd_mfma_kernel(LstmBwdArgs a, int btv) {
  using C = LCfg<BF16>;
  __shared__ __attribute__((aligned(16))) Tile<BF16, G4M> ds[2];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int tile = blockIdx.x, dir = blockIdx.y, mod = blockIdx.z;
  const LstmSeqBwd& q = a.seq[mod][dir];
  const int B = a.B, T = a.T;
  const int n = lane & 15, kq = lane >> 4;
  const int u0 = 32 * w + 2 * n;
  const int b = min(tile * btv + min(kq, btv - 1), B - 1);
  const bool own = kq < btv && tile * btv + kq < B;
  const int len = a.lens[mod][b];
  // B fragments of W_hh for carry[b][u] = sum_k dg[b][k] W_hh[k][u]  (k = gate row 0..511, column = unit u0 + s)
  typename C::Frag wr[2][C::KS_B];
#pragma unroll
  for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
    for (int ks = 0; ks < C::KS_B; ++ks) {
      if constexpr (BF16) {
        bf16x8 f;
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = to_bf16(q.w_hh[(long)(ks * 32 + 8 * kq + j) * H + u0 + s2]);
        wr[s2][ks] = f;
      } else {
        wr[s2][ks] = q.w_hh[(long)(ks * 4 + kq) * H + u0 + s2];
      }
    }
  float dh_c[2] = {0.f, 0.f}, dc_c[2] = {0.f, 0.f};
  const float* sv_b = q.saved + (long)b * T * 6 * H + u0;
  const float* out_b = q.out + (long)b * T * a.out_ld + dir * H + u0;
  const float* dout_b = q.dout + (long)b * T * a.dout_ld + u0;
  float* dg_b = q.dg + (long)b * T * G4M + u0;
  float* hp_b = q.hprev + (long)b * T * H + u0;
  for (int step = 0; step < T; ++step) {
    const int t = dir ? step : T - 1 - step, cur = step & 1;       // reverse of the forward visiting order
    const bool valid = t < len;
    const int tp = dir ? t + 1 : t - 1;
    const int tpc = tp < 0 ? 0 : (tp >= T ? T - 1 : tp);
    float sv[6][2], dO[2], hp[2];
#pragma unroll
    for (int g = 0; g < 6; ++g) ldu<2>(sv_b + (long)t * 6 * H + g * H, sv[g]);     // (padded steps read records nobody wrote: masked below)
    ldu<2>(dout_b + (long)t * a.dout_ld, dO);
    ldu<2>(out_b + (long)tpc * a.out_ld, hp);
    float dp[4][2];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      const float ig = sv[0][s2], fg = sv[1][s2], gg = sv[2][s2], og = sv[3][s2], cp = sv[4][s2], tc = sv[5][s2];
      const float dh = dO[s2] + dh_c[s2];
      const float dc = dc_c[s2] + dh * og * (1.f - tc * tc);
      dp[0][s2] = valid ? dc * gg * ig * (1.f - ig) : 0.f;
      dp[1][s2] = valid ? dc * cp * fg * (1.f - fg) : 0.f;
      dp[2][s2] = valid ? dc * ig * (1.f - gg * gg) : 0.f;
      dp[3][s2] = valid ? dh * tc * og * (1.f - og) : 0.f;
      if (valid) dc_c[s2] = dc * fg;
      hp[s2] = (valid && tp >= 0 && tp < len) ? hp[s2] : 0.f;
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) putu<2>(ds[cur], kq, g * H + u0, dp[g]);
    if (own) {
#pragma unroll
      for (int g = 0; g < 4; ++g) stu<2>(dg_b + (long)t * G4M + g * H, dp[g]);
      stu<2>(hp_b + (long)t * H, hp);
    }
    lds_barrier();
    f32x4 acc[2];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) acc[s2] = f32x4{valid ? 0.f : dh_c[s2], 0.f, 0.f, 0.f};   // a padded step passes the carry through (its dg tile is zero)
#pragma unroll
    for (int ks = 0; ks < C::KS_B; ++ks) {
      const auto sf = state_frag(ds[cur], ks, lane);
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) acc[s2] = mfma16(sf, wr[s2][ks], acc[s2]);
    }
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) dh_c[s2] = acc[s2][0];
    // ds[cur] is rewritten two steps from now; the barrier of the next step orders that write after these reads
  }
}

}  // namespace

static int lstm_btv(int B, int nmod) {
  int btv = (B * nmod * 2 + 127) / 128;      // ~128 workgroups, at most 4 rows each (the recurrence is latency-bound: spread it out)
  return btv < 1 ? 1 : (btv > BR ? BR : btv);
}

// mode: 0 = the scalar fp32 kernels (MIMRL_LSTM_SCALAR=1), 1 = fp32 precision mode, 2 = bf16 MFMA operands (the bf16 precision mode).
// Measured at cfg2's shape (tools/lstm_ab.py): bf16 MFMA 0.955 vs scalar 1.169 ms per two-stage step; the fp32 MFMA form (16x16x4_f32:
// 256 products per cell step at 1/16 of the bf16 rate) 2.40 vs 2.23 ms -- so the fp32 mode keeps the scalar kernels unless
// MIMRL_LSTM_MFMA_FP32=1 (both are exact fp32 fma chains; tests/test_gpu_step.py compares them).
static int lstm_mode(int mode) {
  if (knob_on("MIMRL_LSTM_SCALAR")) return 0;
  if (mode == 1 && !knob_on("MIMRL_LSTM_MFMA_FP32")) return 0;
  return mode;
}

int lstm_forward(hipStream_t s, const LstmFwdArgs& a, int mode) {
  if (a.B <= 0 || a.T <= 0) return set_error(MIMRL_ERR_ARG, "lstm_forward: empty batch");
  mode = lstm_mode(mode);
  const bool save = a.seq[0][0].saved != nullptr;
  if (mode == 0) {
    hipLaunchKernelGGL(lstm_fwd_kernel, dim3(a.B, 2, a.nmod), dim3(512), 0, s, a);
  } else {
    const int btv = lstm_btv(a.B, a.nmod);
    const dim3 grid((a.B + btv - 1) / btv, 2, a.nmod);
    if (mode == 2) {
      if (save) hipLaunchKernelGGL((lstm_fwd_mfma_kernel<true, true>), grid, dim3(256), 0, s, a, btv);
      else hipLaunchKernelGGL((lstm_fwd_mfma_kernel<true, false>), grid, dim3(256), 0, s, a, btv);
    } else {
      if (save) hipLaunchKernelGGL((lstm_fwd_mfma_kernel<false, true>), grid, dim3(256), 0, s, a, btv);
      else hipLaunchKernelGGL((lstm_fwd_mfma_kernel<false, false>), grid, dim3(256), 0, s, a, btv);
    }
  }
  LAUNCH_CHECK();
  return MIMRL_OK;
}

int lstm_backward(hipStream_t s, const LstmBwdArgs& a, int mode) {
  if (a.B <= 0 || a.T <= 0) return set_error(MIMRL_ERR_ARG, "lstm_backward: empty batch");
  mode = lstm_mode(mode);
  if (mode == 0) {
    hipLaunchKernelGGL(lstm_bwd_kernel, dim3(a.B, 2, a.nmod), dim3(512), 0, s, a);
  } else {
    const int btv = lstm_btv(a.B, a.nmod);
    const dim3 grid((a.B + btv - 1) / btv, 2, a.nmod);
    if (mode == 2) hipLaunchKernelGGL((lstm_bwd_mfma_kernel<true>), grid, dim3(256), 0, s, a, btv);
    else hipLaunchKernelGGL((lstm_bwd_mfma_kernel<false>), grid, dim3(256), 0, s, a, btv);
  }
  LAUNCH_CHECK();
  return MIMRL_OK;
}

}  // namespace mimrl
