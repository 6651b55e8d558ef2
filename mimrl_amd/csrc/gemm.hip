// Generic strided batched GEMM kernel (see gemm.h).  64x64 block tile, 4 waves (2x2), one 32x32 MFMA
// accumulator tile per wave, BK = 32, register-prefetched global->LDS staging.
#include "gemm.h"

namespace mimrl {

namespace {

constexpr int BM = 64, BN = 64, BK = 32;

template <bool BF16>
struct Smem;
template <>
struct Smem<false> {
  float a[BK][BM + 1];   // k-major: lanes 0..31 read 32 consecutive m -> conflict-free ds_read_b32
  float b[BK][BN + 1];
};
template <>
struct Smem<true> {
  __bf16 a[BM][BK + 8];  // m-major, 80-B rows: 16-B fragment reads, conflict-free for ds_read_b128
  __bf16 b[BN][BK + 8];
};

template <bool BF16>
__global__ __launch_bounds__(256) void gemm_kernel(GemmDesc d) {
  __shared__ __attribute__((aligned(16))) Smem<BF16> sm;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int bz = blockIdx.z;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const float* __restrict__ A = d.A + (long)bz * d.sa_b;
  const float* __restrict__ B = d.B + (long)bz * d.sb_b;
  const bool a_kfast = (d.sa_k == 1) || (d.sa_m != 1);   // which index runs fastest across threads
  const bool b_kfast = (d.sb_k == 1) && (d.sb_n != 1);

  float ra[8], rb[8];
  auto load_tiles = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int e = i * 256 + tid;
      int kk, mm;
      if (a_kfast) { kk = e & (BK - 1); mm = e >> 5; } else { mm = e & (BM - 1); kk = e >> 6; }
      const int gm = m0 + mm, gk = k0 + kk;
      ra[i] = (gm < d.M && gk < d.K) ? A[(long)gm * d.sa_m + (long)gk * d.sa_k] : 0.f;
      int kb, nn;
      if (b_kfast) { kb = e & (BK - 1); nn = e >> 5; } else { nn = e & (BN - 1); kb = e >> 6; }
      const int gn = n0 + nn, gkb = k0 + kb;
      rb[i] = (gn < d.N && gkb < d.K) ? B[(long)gkb * d.sb_k + (long)gn * d.sb_n] : 0.f;
    }
  };
  auto store_tiles = [&]() {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int e = i * 256 + tid;
      int kk, mm;
      if (a_kfast) { kk = e & (BK - 1); mm = e >> 5; } else { mm = e & (BM - 1); kk = e >> 6; }
      int kb, nn;
      if (b_kfast) { kb = e & (BK - 1); nn = e >> 5; } else { nn = e & (BN - 1); kb = e >> 6; }
      if constexpr (BF16) {
        sm.a[mm][kk] = to_bf16(ra[i]);
        sm.b[nn][kb] = to_bf16(rb[i]);
      } else {
        sm.a[kk][mm] = ra[i];
        sm.b[kb][nn] = rb[i];
      }
    }
  };

  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;

  const int ktiles = (d.K + BK - 1) / BK;
  load_tiles(0);
  for (int kt = 0; kt < ktiles; ++kt) {
    store_tiles();
    __syncthreads();
    if (kt + 1 < ktiles) load_tiles((kt + 1) * BK);
    if constexpr (BF16) {
#pragma unroll
      for (int ks = 0; ks < BK / 16; ++ks) {
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(&sm.a[wm * 32 + (lane & 31)][ks * 16 + 8 * (lane >> 5)]);
        const bf16x8 b = *reinterpret_cast<const bf16x8*>(&sm.b[wn * 32 + (lane & 31)][ks * 16 + 8 * (lane >> 5)]);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int ks = 0; ks < BK / 2; ++ks) {
        const float a = sm.a[ks * 2 + (lane >> 5)][wm * 32 + (lane & 31)];
        const float b = sm.b[ks * 2 + (lane >> 5)][wn * 32 + (lane & 31)];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
      }
    }
    __syncthreads();
  }

  // epilogue: lane holds column n, 16 rows
  const int n = n0 + wn * 32 + (lane & 31);
  if (n >= d.N) return;
  float* __restrict__ C = d.C + (long)bz * d.sc_b;
  const float bn = d.bias_n ? d.bias_n[(long)bz * d.bias_n_b + n] : 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    if (m >= d.M) continue;
    const long off = (long)m * d.sc_m + (long)n * d.sc_n;
    float v = d.alpha * acc[r] + bn;
    if (d.bias_m) v += d.bias_m[(long)bz * d.bias_m_b + m];
    if (d.beta != 0.f) v += d.beta * C[off];
    if (d.pre) d.pre[(long)bz * d.sc_b + off] = v;
    if (d.gradact_u) v *= act_grad(d.act, d.gradact_u[(long)bz * d.sc_b + off]);
    else v = act_apply(d.act, v);
    if (d.atomic) atomicAdd(&C[off], v);
    else C[off] = v;
  }
}

}  // namespace

int gemm(hipStream_t s, const GemmDesc& d, bool bf16) {
  if (d.M <= 0 || d.N <= 0 || d.batch <= 0) return MIMRL_OK;
  if (!d.A || !d.B || !d.C) return set_error(MIMRL_ERR_ARG, "gemm: null operand");
  dim3 grid((d.N + BN - 1) / BN, (d.M + BM - 1) / BM, d.batch);
  if (grid.y > 65535 || grid.z > 65535) return set_error(MIMRL_ERR_ARG, "gemm: grid too large (M=%d batch=%d)", d.M, d.batch);
  if (bf16) hipLaunchKernelGGL(gemm_kernel<true>, grid, dim3(256), 0, s, d);
  else hipLaunchKernelGGL(gemm_kernel<false>, grid, dim3(256), 0, s, d);
  LAUNCH_CHECK();
  return MIMRL_OK;
}

}  // namespace mimrl
