// Fused data-gradient chains of one CubeMLP block (see cube_bwd_fused.h).
#include "cube_bwd_fused.h"

namespace mimrl {

namespace {

constexpr int CT = 128;       // L axis: columns per workgroup
constexpr int KP = 64 + 8;    // bf16 pitch of a [.][<=64] operand image (144 B: conflict-free 16-byte fragment reads)

__device__ __forceinline__ float half_sum32(float v) {   // sum over the 32 lanes that share lane>>5
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// ---------------------------------------------------------------------------------------------------------------
// L axis.  Workgroup = (sample b, 128 columns).  MFMA 32x32x16: M = rows of the transposed weight (h or i), N = columns,
// K = o or h.  dY / dU are kept TRANSPOSED in LDS ([column][k]) so that B-fragments are 16-byte reads; the weights
// are staged transposed and zero-padded to 64x64, which also takes care of ragged hl / ol / il.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void laxis_bwd_kernel(LAxisBwdArgs a) {
  __shared__ __attribute__((aligned(16))) __bf16 sdy[CT][KP];
  __shared__ __attribute__((aligned(16))) __bf16 sdu[CT][KP];
  __shared__ __attribute__((aligned(16))) __bf16 wts[3][64][KP];
  // wts[0] = W2^T [h][o], wts[1] = W1^T [i][h], wts[2] = Wr^T [i][o]
  float (*red)[2][CT] = reinterpret_cast<float (*)[2][CT]>(&sdu[0][0]);   // phase-1 scratch; sdu is first written in phase 2
  __shared__ float acc_l[3][64];   // per-l partial sums of dgamma, dbeta, db2
  __shared__ float acc_h[64];      // per-h partial sums of db1
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 31, lh = lane >> 5;
  const int b = blockIdx.y, c0 = blockIdx.x * CT;
  const int il = a.il, hl = a.hl, ol = a.ol, C = a.C;

  // weights, transposed and zero-padded to 64x64: zero fill with 16-byte stores, then scatter from coalesced reads
  {
    uint4* z = reinterpret_cast<uint4*>(&wts[0][0][0]);
    for (int i = tid; i < 3 * 64 * KP * 2 / 16; i += 256) z[i] = make_uint4(0u, 0u, 0u, 0u);
    if (tid < 64) { acc_l[0][tid] = 0.f; acc_l[1][tid] = 0.f; acc_l[2][tid] = 0.f; acc_h[tid] = 0.f; }
  }
  __syncthreads();
  for (int idx = tid; idx < ol * hl; idx += 256) { const int o = idx / hl, h = idx - o * hl; wts[0][h][o] = to_bf16(a.w2[idx]); }
  for (int idx = tid; idx < hl * il; idx += 256) { const int h = idx / il, i = idx - h * il; wts[1][i][h] = to_bf16(a.w1[idx]); }
  for (int idx = tid; idx < ol * il; idx += 256) { const int o = idx / il, i = idx - o * il; wts[2][i][o] = to_bf16(a.wr[idx]); }

  // ---- phase 1: LayerNorm over L, backward.  Thread = (column, half of the rows l = half, half+2, ...); two passes over
  // the (L2-resident) column instead of 64 live registers
  const int col = tid & (CT - 1), half = tid >> 7;
  const long cidx = (long)b * C + c0 + col;
  const float mu = a.mean[cidx], rs = a.rstd[cidx];
  const float* __restrict__ dzb = a.dz + (long)b * ol * C + c0 + col;
  const float* __restrict__ yb = a.y + (long)b * ol * C + c0 + col;
  float* __restrict__ dyb = a.dy + (long)b * ol * C + c0 + col;
  float s1 = 0.f, s2 = 0.f;
#pragma unroll 8
  for (int l = half; l < ol; l += 2) {
    const float gq = dzb[(long)l * C], xq = (yb[(long)l * C] - mu) * rs;
    const float dxh = gq * a.gamma[l];
    s1 += dxh; s2 += dxh * xq;
  }
  red[0][half][col] = s1;
  red[1][half][col] = s2;
  __syncthreads();
  s1 = (red[0][0][col] + red[0][1][col]) / ol;
  s2 = (red[1][0][col] + red[1][1][col]) / ol;
#pragma unroll 8
  for (int l = half; l < 64; l += 2) {
    float v = 0.f;
    if (l < ol) {                      // uniform per wave
      const float gq = dzb[(long)l * C], xq = (yb[(long)l * C] - mu) * rs;
      v = rs * (gq * a.gamma[l] - s1 - xq * s2);
      dyb[(long)l * C] = v;
    }
    sdy[col][l] = to_bf16(v);          // rows l >= ol: zero padding of the K axis
  }
  __syncthreads();

  if (a.db2 && tid >= 192 && tid - 192 < ol) {   // db2[o] = sum over this tile's columns of dY (bf16 image, as the weight-gradient GEMM sees it)
    float t = 0.f;
    for (int c = 0; c < CT; ++c) t += (float)sdy[(c + tid) & (CT - 1)][tid - 192];
    acc_l[2][tid - 192] = t;
  }
  const int nt = wave;                 // this wave's 32-column tile; it owns both 32-row M tiles of it
  const int cc = c0 + nt * 32 + lr;
  // ---- phase 2: dU = (W2^T dY) * act'(U)
  {
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    const int ksteps = (ol + 15) / 16;
    for (int ks = 0; ks < ksteps; ++ks) {
      const bf16x8 bf = *reinterpret_cast<const bf16x8*>(&sdy[nt * 32 + lr][ks * 16 + 8 * lh]);
      const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(&wts[0][lr][ks * 16 + 8 * lh]);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, bf, acc0, 0, 0, 0);
      if (hl > 32) {
        const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(&wts[0][32 + lr][ks * 16 + 8 * lh]);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, bf, acc1, 0, 0, 0);
      }
    }
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int h = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        float v = 0.f;
        if (h < hl) {
          const long ui = ((long)b * hl + h) * C + cc;
          v = (mt == 0 ? acc0[r] : acc1[r]) * act_grad(a.act, a.u[ui]);
          a.du[ui] = v;
        }
        sdu[nt * 32 + lr][h] = to_bf16(v);
        if (a.db1) {
          const float t = half_sum32(v);
          if (lr == 0 && h < hl) atomicAdd(&acc_h[h], t);
        }
      }
  }
  __syncthreads();
  // ---- phase 3: dX = W1^T dU + Wr^T dY
  {
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    const int k1 = (hl + 15) / 16, k2 = (ol + 15) / 16;
    for (int ks = 0; ks < k1 + k2; ++ks) {
      const bool first = ks < k1;
      const int kk = (first ? ks : ks - k1) * 16 + 8 * lh;
      const bf16x8 bf = *reinterpret_cast<const bf16x8*>(first ? &sdu[nt * 32 + lr][kk] : &sdy[nt * 32 + lr][kk]);
      const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(&wts[first ? 1 : 2][lr][kk]);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, bf, acc0, 0, 0, 0);
      if (il > 32) {
        const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(&wts[first ? 1 : 2][32 + lr][kk]);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, bf, acc1, 0, 0, 0);
      }
    }
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int i = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (i < il) a.dx[((long)b * il + i) * C + cc] = mt == 0 ? acc0[r] : acc1[r];
      }
  }
  __syncthreads();
  if (a.db2 && tid < ol) atomicAdd(&a.db2[tid], acc_l[2][tid]);
  if (a.db1 && tid >= 64 && tid - 64 < hl) atomicAdd(&a.db1[tid - 64], acc_h[tid - 64]);
}

}  // namespace

bool laxis_bwd_supported(int il, int hl, int ol, int C) {
  return il >= 1 && hl >= 1 && ol >= 1 && il <= 64 && hl <= 64 && ol <= 64 && C % CT == 0;
}

int laxis_bwd_fused(hipStream_t s, const LAxisBwdArgs& a) {
  if (!laxis_bwd_supported(a.il, a.hl, a.ol, a.C)) return set_error(MIMRL_ERR_ARG, "laxis_bwd_fused: unsupported shape");
  if (!a.wr) return set_error(MIMRL_ERR_ARG, "laxis_bwd_fused: needs the residual projection");
  hipLaunchKernelGGL(laxis_bwd_kernel, dim3(a.C / CT, a.B), dim3(256), 0, s, a);
  LAUNCH_CHECK();
  return MIMRL_OK;
}

bool daxis_bwd_supported(int, int, int) { return false; }
int daxis_bwd_fused(hipStream_t, const DAxisBwdArgs&) { return set_error(MIMRL_ERR_ARG, "daxis_bwd_fused: not built"); }

}  // namespace mimrl
