// Concat critic (VMI.py:58-65: scores[i, j] = f([x_i | y_j]), f = Linear/ReLU x3 + Linear, VMI.py:13-22), WEIGHTS-STATIONARY kernels (round 6).
//
// The dense contraction of this workload: E x B*B pair rows through two 256 x 256 hidden layers (cfg3: 5 x 65 536 rows, 86 GFLOP per pass).
// The round 2-5 kernels (concat_fused.hip) gave a workgroup 128 pair rows for the whole stack and STREAMED the weights: every tile re-staged
// both 256 x 256 matrices from L2 through LDS in 32-k chunks (0.65 GB of L2 -> LDS per pass at cfg3) and its four row-tile waves each re-read
// the same B fragments -- 640 KB of LDS reads per tile and layer against 4096 MFMA cycles per SIMD: LDS-bound at 12-20 % MFMA-busy.
//
// Here the weights never move after the first tile.  One PERSISTENT workgroup per CU, 8 waves = 2 per SIMD:
//   * waves 0-3 own layer 1, waves 4-7 layer 2; wave w of a layer holds the 64-feature slice [64 w, 64 w + 64) of its layer's matrix as
//     ready-made MFMA A fragments in REGISTERS for the whole launch (2 feature tiles x 16 k-steps x 4 VGPRs = 128 VGPRs; the way gru.hip keeps
//     W_hh), re-loaded only when the run crosses into the next estimator;
//   * the products are TRANSPOSED: C^T[feature][row] = W[feature][k] . act^T[k][row], i.e. the weights are the A operand and the activation
//     tile the B operand (one 16-byte LDS read per lane and k-step, shared by the wave's two feature tiles).  A lane then owns ONE pair row and
//     4 consecutive features per accumulator quad: the next layer's operand tile is written with 8-byte LDS stores (not 2-byte ones), the
//     ReLU sign word of (row, 32 features) is built in the lane, and the 256 -> 1 score head is a per-lane dot product;
//   * a two-stage pipeline over UNITS of 32 pair rows: in step s the layer-1 waves work on unit s while the layer-2 waves work on unit s - 1
//     and all waves generate layer 0 (relu(P_i + Q_j), the separable form of the first Linear) of unit s + 1 -- one barrier per step, every
//     SIMD always has one wave in its product and one in an epilogue.  LDS reads per unit and layer: 4 waves x 16 KB (the activation tile,
//     once per 64-feature slice) instead of 160 KB.
// The bias is the accumulator's initial value; results are those of concat_fused.hip up to fp32 summation order inside a product.
#include "concat_fused.h"

namespace mimrl {

namespace {

constexpr int CH = 256;            // hidden width (VMI.py:13-22 with hidden_dim 256)
constexpr int UR = 32;             // pair rows per pipeline unit
constexpr int AP = CH + 8;         // bf16 pitch of an activation tile row (528 B: conflict-free 16-byte fragment reads)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t sign_bits4(float a, float b, float c, float d) {
  return (a > 0.f ? 1u : 0u) | (b > 0.f ? 2u : 0u) | (c > 0.f ? 4u : 0u) | (d > 0.f ? 8u : 0u);
}

// SAVE: what the backward pass gets (ConcatFwdArgs::save): 0 nothing, 2 bf16 a0 / a1 + fp32 a2 + sign words (stage 1), 3 sign words only
template <int SAVE>
__global__ __launch_bounds__(512) void concat_fwd_ws_kernel(ConcatFwdArgs a, int units_e, int total, int per) {
  __shared__ __attribute__((aligned(16))) __bf16 act0[2][UR][AP];   // layer-0 outputs (operand of layer 1), double-buffered over units
  __shared__ __attribute__((aligned(16))) __bf16 act1[2][UR][AP];   // layer-1 outputs (operand of layer 2)
  __shared__ __attribute__((aligned(16))) float sbias[2][CH];       // [layer][feature]: each wave writes and reads only its own 64-feature slice
  __shared__ __attribute__((aligned(16))) float sw3[CH];            // score-head weight (layer-2 waves, own slice)
  __shared__ LdsAcc sc[2][UR];                                      // score sums of a unit over the four layer-2 waves
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 31, lh = lane >> 5;
  const int role = wave >> 2, ws = wave & 3;       // role 0: layer 1, role 1: layer 2; ws: 64-feature slice
  const int B = a.B;
  const int u0 = blockIdx.x * per, U = min(total, u0 + per) - u0;
  if (U <= 0) return;
  if (tid < 2 * UR) sc[tid >> 5][tid & 31].zero();
  bf16x8 wf[2][16];                                // this wave's weight slice: [feature tile][k-step], A fragments of v_mfma_f32_32x32x16_bf16
#pragma unroll
  for (int ct = 0; ct < 2; ++ct)
#pragma unroll
    for (int ks = 0; ks < 16; ++ks)
#pragma unroll
      for (int i = 0; i < 8; ++i) wf[ct][ks][i] = (__bf16)0.f;
  int my_e = -1;
  const int c4 = lane * 4;                         // layer-0 generation: lane = column quad, wave w = rows w, w + 8, w + 16, w + 24 of the unit
  float4 xq = make_float4(0.f, 0.f, 0.f, 0.f), yq[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) yq[q] = xq;
#pragma unroll 1
  for (int s = -1; s <= U + 1; ++s) {
    // ---- (1) layer 0 of unit s + 1: request P_i / Q_j now, use them behind the product
    const bool gen_on = s + 1 < U;
    long gbase = 0;                                // first pair row of unit s + 1 in the [E][B*B] row space
    if (gen_on) {
      const int lin = u0 + s + 1, ge = lin / units_e, grow0 = (lin - ge * units_e) * UR;
      gbase = (long)ge * B * B + grow0;
      const int gi = grow0 / B, gj0 = grow0 - gi * B;      // B % 32 == 0: a unit is one x row i and 32 consecutive y rows j
      const float* __restrict__ Pp = a.P + ((long)ge * B + gi) * CH + c4;
      const float* __restrict__ Qp = a.Q + ((long)ge * B + gj0 + wave) * CH + c4;
      xq = *reinterpret_cast<const float4*>(Pp);
#pragma unroll
      for (int q = 0; q < 4; ++q) yq[q] = *reinterpret_cast<const float4*>(Qp + (long)(8 * q) * CH);
    }
    // ---- (2) this wave's layer on its unit: layer 1 on unit s, layer 2 on unit s - 1
    const int pu = s - role;
    if (pu >= 0 && pu < U) {
      const int lin = u0 + pu, e = lin / units_e, row0 = (lin - e * units_e) * UR;
      const long prow = (long)e * B * B + row0 + lr;       // this lane's pair row
      if (e != my_e) {                                     // (wave-uniform) first unit / the run entered the next estimator: this wave's slice
        my_e = e;
        const __bf16* __restrict__ W = (role ? a.W2 : a.W1) + (long)e * a.pstride + (long)(ws * 64 + lr) * CH + 8 * lh;
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
          for (int ks = 0; ks < 16; ++ks) wf[ct][ks] = *reinterpret_cast<const bf16x8*>(W + (long)(ct * 32) * CH + ks * 16);
        sbias[role][ws * 64 + lane] = ((role ? a.b2 : a.b1) + (long)e * a.pstride)[ws * 64 + lane];
        if (role) sw3[ws * 64 + lane] = a.w3[(long)e * a.pstride + ws * 64 + lane];
        __builtin_amdgcn_wave_barrier();                   // (LDS operations of one wave complete in order: the reads below see these)
      }
      f32x16 acc[2];
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int q = 0; q < 4; ++q) {                      // accumulator quad q of tile ct = features 64 ws + 32 ct + 8 q + 4 lh + (0..3): starts at the bias
          const float4 bb = *reinterpret_cast<const float4*>(&sbias[role][ws * 64 + ct * 32 + 8 * q + 4 * lh]);
          acc[ct][4 * q] = bb.x; acc[ct][4 * q + 1] = bb.y; acc[ct][4 * q + 2] = bb.z; acc[ct][4 * q + 3] = bb.w;
        }
      const __bf16 (*src)[AP] = role ? act1[(s - 1) & 1] : act0[s & 1];
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) {
        const bf16x8 bfr = *reinterpret_cast<const bf16x8*>(&src[lr][ks * 16 + 8 * lh]);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[0][ks], bfr, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[1][ks], bfr, acc[1], 0, 0, 0);
      }
      if (role == 0) {
        // layer 1 epilogue: ReLU -> bf16 operand tile of layer 2 (8-byte LDS stores) + the sign word of (row, 32 features)
        __bf16 (*dst)[AP] = act1[s & 1];
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
          uint32_t bits = 0u;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float v0 = fmaxf(acc[ct][4 * q], 0.f), v1 = fmaxf(acc[ct][4 * q + 1], 0.f), v2 = fmaxf(acc[ct][4 * q + 2], 0.f), v3 = fmaxf(acc[ct][4 * q + 3], 0.f);
            bf16x4 b; b[0] = to_bf16(v0); b[1] = to_bf16(v1); b[2] = to_bf16(v2); b[3] = to_bf16(v3);
            *reinterpret_cast<bf16x4*>(&dst[lr][ws * 64 + ct * 32 + 8 * q + 4 * lh]) = b;
            if (SAVE >= 2) bits |= sign_bits4(v0, v1, v2, v3) << (8 * q);
          }
          if (SAVE >= 2) {
            bits <<= 4 * lh;
            bits |= (uint32_t)__shfl_xor((int)bits, 32);
            if (lh == 0) a.m1[prow * 8 + ws * 2 + ct] = bits;
          }
        }
      } else {
        // layer 2 epilogue: ReLU -> sign word, fp32 a2 (stage 1: the score head's weight gradient reads it), score head partial dot product
        float hp = 0.f;
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
          uint32_t bits = 0u;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int f0 = ws * 64 + ct * 32 + 8 * q + 4 * lh;
            float4 v;
            v.x = fmaxf(acc[ct][4 * q], 0.f); v.y = fmaxf(acc[ct][4 * q + 1], 0.f); v.z = fmaxf(acc[ct][4 * q + 2], 0.f); v.w = fmaxf(acc[ct][4 * q + 3], 0.f);
            const float4 w3v = *reinterpret_cast<const float4*>(&sw3[f0]);
            hp += v.x * w3v.x + v.y * w3v.y + v.z * w3v.z + v.w * w3v.w;
            if (SAVE == 2) *reinterpret_cast<float4*>(a.a2 + prow * CH + f0) = v;
            if (SAVE >= 2) bits |= sign_bits4(v.x, v.y, v.z, v.w) << (8 * q);
          }
          if (SAVE >= 2) {
            bits <<= 4 * lh;
            bits |= (uint32_t)__shfl_xor((int)bits, 32);
            if (lh == 0) a.m2[prow * 8 + ws * 2 + ct] = bits;
          }
        }
        hp += __shfl_xor(hp, 32);
        if (lh == 0) sc[(s - 1) & 1][lr].add(hp);
      }
    }
    // ---- (3) stage 1: the finished layer-1 tile of unit s - 1 leaves as bf16 in whole 512-byte rows (operand of the dW2 product)
    if (SAVE == 2 && s >= 1 && s <= U) {
      const int lin = u0 + s - 1, e = lin / units_e, row0 = (lin - e * units_e) * UR;
      __bf16* __restrict__ o = a.a1b + ((long)e * B * B + row0) * CH;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int idx = tid + 512 * q, row = idx >> 5, c8 = (idx & 31) * 8;
        *reinterpret_cast<u32x4*>(o + (long)row * CH + c8) = *reinterpret_cast<const u32x4*>(&act1[(s - 1) & 1][row][c8]);
      }
    }
    // ---- (4) layer 0 of unit s + 1 -> act0[(s + 1) & 1] (read by the layer-1 waves in the next step), its sign words, stage 1: its bf16 copy
    if (gen_on) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int row = wave + 8 * q;
        float4 v;
        v.x = fmaxf(xq.x + yq[q].x, 0.f); v.y = fmaxf(xq.y + yq[q].y, 0.f); v.z = fmaxf(xq.z + yq[q].z, 0.f); v.w = fmaxf(xq.w + yq[q].w, 0.f);
        bf16x4 b; b[0] = to_bf16(v.x); b[1] = to_bf16(v.y); b[2] = to_bf16(v.z); b[3] = to_bf16(v.w);
        if (SAVE == 2) *reinterpret_cast<bf16x4*>(a.a0b + (gbase + row) * CH + c4) = b;
        if (SAVE >= 2) {   // sign word of (row, 32 columns) = the nibbles of 8 neighbouring lanes (this wave holds one row: lane = column quad)
          uint32_t nib = sign_bits4(v.x, v.y, v.z, v.w);
          nib |= (uint32_t)__shfl_down((int)nib, 1) << 4;
          nib |= (uint32_t)__shfl_down((int)nib, 2) << 8;
          nib |= (uint32_t)__shfl_down((int)nib, 4) << 16;
          if ((lane & 7) == 0) a.m0[(gbase + row) * 8 + (lane >> 3)] = nib;
        }
        *reinterpret_cast<bf16x4*>(&act0[(s + 1) & 1][row][c4]) = b;
      }
    }
    // ---- (5) scores of unit s - 2 (complete since the last barrier)
    if (s >= 2 && tid < UR) {
      const int lin = u0 + s - 2, e = lin / units_e, row0 = (lin - e * units_e) * UR;
      a.scores[(long)e * B * B + row0 + tid] = sc[s & 1][tid].get() + a.b3[(long)e * a.pstride];
      sc[s & 1][tid].zero();
    }
    __syncthreads();
  }
}

int device_cus() {
  static int cus = 0;
  if (!cus) {
    hipDeviceProp_t pr;
    int dev = 0;
    cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) ? pr.multiProcessorCount : 256;
  }
  return cus;
}

}  // namespace

bool concat_fwd_ws_supported(int B, int hid, int save) { return hid == CH && B >= UR && B % UR == 0 && (save == 0 || save == 2 || save == 3); }

int concat_fwd_ws(hipStream_t s, const ConcatFwdArgs& a) {
  if (!concat_fwd_ws_supported(a.B, CH, a.save)) return set_error(MIMRL_ERR_ARG, "concat_fwd_ws: batch %d / save %d unsupported", a.B, a.save);
  const int units_e = (int)(((long)a.B * a.B) / UR), total = a.E * units_e;
  const int nwg0 = std::min(device_cus(), total), per = (total + nwg0 - 1) / nwg0, nwg = (total + per - 1) / per;
  const dim3 grid((unsigned)nwg);
  if (a.save == 0) hipLaunchKernelGGL(concat_fwd_ws_kernel<0>, grid, dim3(512), 0, s, a, units_e, total, per);
  else if (a.save == 2) hipLaunchKernelGGL(concat_fwd_ws_kernel<2>, grid, dim3(512), 0, s, a, units_e, total, per);
  else hipLaunchKernelGGL(concat_fwd_ws_kernel<3>, grid, dim3(512), 0, s, a, units_e, total, per);
  LAUNCH_CHECK();
  return MIMRL_OK;
}

}  // namespace mimrl
