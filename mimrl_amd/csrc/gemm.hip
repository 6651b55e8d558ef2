// Generic strided batched GEMM kernel (see gemm.h).  64x64 block tile, 4 waves (2x2), one 32x32 MFMA accumulator
// tile per wave, BK = 32, register-prefetched global->LDS staging.
//   * interior tiles stage with 16-byte loads along whichever axis of the operand is contiguous; edge tiles and
//     misaligned operands fall back to guarded scalar loads (wave-uniform choice per tile);
//   * split-K: weight-gradient GEMMs (K = B*T ... B*L*K rows, tiny M x N) are spread over grid.z and accumulated with
//     float atomics into the zeroed gradient bucket, so they fill the chip instead of running on a handful of CUs.
#include "gemm.h"

namespace mimrl {

namespace {

constexpr int BM = 64, BN = 64, BK = 32;

template <bool BF16>
struct Smem;
template <>
struct Smem<false> {
  float a[BK][BM + 4];   // k-major: lanes 0..31 read 32 consecutive m; +4 keeps 16-B row alignment
  float b[BK][BN + 4];
};
template <>
struct Smem<true> {
  __bf16 a[BM][BK + 8];  // m-major, 80-B rows: 16-B fragment reads, conflict-free for ds_read_b128
  __bf16 b[BN][BK + 8];
};

struct KernelArgs {
  GemmDesc d;
  int ksplit;       // grid.z = batch * ksplit
  int kt_per;       // k-tiles per split
  int vec_a, vec_b; // operand may use the 16-byte path (alignment / stride conditions hold)
};

// stage one 64(rows) x 32(k) operand tile into registers.  `rfast`: rows (m or n) are the contiguous axis.
// layout of the 8 per-thread values:
//   vector path, k contiguous : v[0..3] = row r0, k k4..k4+3 ; v[4..7] = row r0+32
//   vector path, row contig.  : v[0..3] = k k0, rows r4..r4+3 ; v[4..7] = k k0+16
//   scalar path               : element e = i*256+tid ; (kfast) k = e&31,row = e>>5  | (rfast) row = e&63, k = e>>6
struct TileIdx {
  int tid;
  __device__ __forceinline__ void vec_k(int h, int& row, int& k) const { row = (tid >> 3) + 32 * h; k = (tid & 7) * 4; }
  __device__ __forceinline__ void vec_r(int h, int& row, int& k) const { k = (tid >> 4) + 16 * h; row = (tid & 15) * 4; }
  __device__ __forceinline__ void sc(int i, bool kfast, int& row, int& k) const {
    const int e = i * 256 + tid;
    if (kfast) { k = e & (BK - 1); row = e >> 5; } else { row = e & 63; k = e >> 6; }
  }
};

template <bool BF16>
__global__ __launch_bounds__(256) void gemm_kernel(KernelArgs ka) {
  const GemmDesc& d = ka.d;
  __shared__ __attribute__((aligned(16))) Smem<BF16> sm;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int bz = blockIdx.z / ka.ksplit, ks = blockIdx.z - bz * ka.ksplit;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const float* __restrict__ A = d.A + (long)bz * d.sa_b;
  const float* __restrict__ B = d.B + (long)bz * d.sb_b;
  const bool a_kfast = d.sa_k == 1;       // else rows (m) fastest
  const bool b_kfast = d.sb_k == 1 && d.sb_n != 1;
  const bool a_rfast = d.sa_m == 1 && !a_kfast;
  const bool b_rfast = d.sb_n == 1;
  const bool full_m = m0 + BM <= d.M, full_n = n0 + BN <= d.N;
  const bool va = ka.vec_a && full_m && (a_kfast || a_rfast);
  const bool vb = ka.vec_b && full_n && (b_kfast || b_rfast);
  const TileIdx ti{tid};

  const int ktiles = (d.K + BK - 1) / BK;
  const int kt0 = ks * ka.kt_per;
  const int kt1 = kt0 + ka.kt_per < ktiles ? kt0 + ka.kt_per : ktiles;

  float ra[8], rb[8];
  bool ra_vec = false, rb_vec = false;   // how the registers currently held were loaded
  auto load_operand = [&](const float* __restrict__ P, long s_r, long s_k, int r0, int R, bool kfast, bool vec_ok, int k0,
                          float* v, bool& used_vec) {
    const bool kfull = k0 + BK <= d.K;
    used_vec = vec_ok && kfull;
    if (used_vec) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        int row, k;
        if (kfast) ti.vec_k(h, row, k); else ti.vec_r(h, row, k);
        const float4 q = *reinterpret_cast<const float4*>(P + (long)(r0 + row) * s_r + (long)(k0 + k) * s_k);
        v[4 * h + 0] = q.x; v[4 * h + 1] = q.y; v[4 * h + 2] = q.z; v[4 * h + 3] = q.w;
      }
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        int row, k;
        ti.sc(i, kfast, row, k);
        const int gr = r0 + row, gk = k0 + k;
        v[i] = (gr < R && gk < d.K) ? P[(long)gr * s_r + (long)gk * s_k] : 0.f;
      }
    }
  };
  auto store_operand = [&](const float* v, bool kfast, bool used_vec, auto& tile_bf16, auto& tile_f32) {
    if (used_vec) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        int row, k;
        if (kfast) {
          ti.vec_k(h, row, k);
          if constexpr (BF16) {
            bf16x4 p; p[0] = to_bf16(v[4 * h]); p[1] = to_bf16(v[4 * h + 1]); p[2] = to_bf16(v[4 * h + 2]); p[3] = to_bf16(v[4 * h + 3]);
            *reinterpret_cast<bf16x4*>(&tile_bf16[row][k]) = p;
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) tile_f32[k + j][row] = v[4 * h + j];
          }
        } else {
          ti.vec_r(h, row, k);
          if constexpr (BF16) {
#pragma unroll
            for (int j = 0; j < 4; ++j) tile_bf16[row + j][k] = to_bf16(v[4 * h + j]);
          } else {
            *reinterpret_cast<float4*>(&tile_f32[k][row]) = make_float4(v[4 * h], v[4 * h + 1], v[4 * h + 2], v[4 * h + 3]);
          }
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        int row, k;
        ti.sc(i, kfast, row, k);
        if constexpr (BF16) tile_bf16[row][k] = to_bf16(v[i]);
        else tile_f32[k][row] = v[i];
      }
    }
  };

  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;

  if (kt0 < kt1) {
    load_operand(A, d.sa_m, d.sa_k, m0, d.M, a_kfast || !a_rfast, va, kt0 * BK, ra, ra_vec);
    load_operand(B, d.sb_n, d.sb_k, n0, d.N, b_kfast, vb, kt0 * BK, rb, rb_vec);
  }
  for (int kt = kt0; kt < kt1; ++kt) {
    store_operand(ra, a_kfast || !a_rfast, ra_vec, sm.a, sm.a);
    store_operand(rb, b_kfast, rb_vec, sm.b, sm.b);
    __syncthreads();
    if (kt + 1 < kt1) {
      load_operand(A, d.sa_m, d.sa_k, m0, d.M, a_kfast || !a_rfast, va, (kt + 1) * BK, ra, ra_vec);
      load_operand(B, d.sb_n, d.sb_k, n0, d.N, b_kfast, vb, (kt + 1) * BK, rb, rb_vec);
    }
    if constexpr (BF16) {
#pragma unroll
      for (int s = 0; s < BK / 16; ++s) {
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(&sm.a[wm * 32 + (lane & 31)][s * 16 + 8 * (lane >> 5)]);
        const bf16x8 b = *reinterpret_cast<const bf16x8*>(&sm.b[wn * 32 + (lane & 31)][s * 16 + 8 * (lane >> 5)]);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int s = 0; s < BK / 2; ++s) {
        const float a = sm.a[s * 2 + (lane >> 5)][wm * 32 + (lane & 31)];
        const float b = sm.b[s * 2 + (lane >> 5)][wn * 32 + (lane & 31)];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
      }
    }
    __syncthreads();
  }

  // epilogue: lane holds column n, 16 rows
  const int n = n0 + wn * 32 + (lane & 31);
  if (n >= d.N) return;
  float* __restrict__ C = d.C + (long)bz * d.sc_b;
  const bool atomic = d.atomic || ka.ksplit > 1;
  const float bn = d.bias_n ? d.bias_n[(long)bz * d.bias_n_b + n] : 0.f;
  float csum = 0.f;
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
    if (atomic) atomicAdd(&C[off], v);
    else C[off] = v;
    csum += v;
  }
  if (d.colsum) {   // fused bias gradient: lanes l and l^32 hold the same column
    csum += __shfl_xor(csum, 32, 64);
    if (lane < 32) atomicAdd(&d.colsum[(long)bz * d.colsum_b + n], csum);
  }
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

int gemm(hipStream_t s, const GemmDesc& d, bool bf16) {
  if (d.M <= 0 || d.N <= 0 || d.batch <= 0) return MIMRL_OK;
  if (!d.A || !d.B || !d.C) return set_error(MIMRL_ERR_ARG, "gemm: null operand");
  KernelArgs ka;
  ka.d = d;
  // 16-byte path: unit stride along one axis, the other stride and the batch stride multiples of 4 floats, base aligned
  ka.vec_a = aligned16(d.A) && d.sa_b % 4 == 0 &&
             ((d.sa_k == 1 && d.sa_m % 4 == 0) || (d.sa_m == 1 && d.sa_k != 1 && d.sa_k % 4 == 0));
  ka.vec_b = aligned16(d.B) && d.sb_b % 4 == 0 &&
             ((d.sb_k == 1 && d.sb_n != 1 && d.sb_n % 4 == 0) || (d.sb_n == 1 && d.sb_k % 4 == 0));
  const int tiles = ((d.N + BN - 1) / BN) * ((d.M + BM - 1) / BM) * d.batch;
  const int ktiles = (d.K + BK - 1) / BK;
  int ksplit = 1;
  if (d.atomic && d.beta == 0.f && !d.bias_n && !d.bias_m && !d.pre && !d.gradact_u && !d.colsum && d.act == ACT_NONE && ktiles >= 8) {
    // accumulate-into-zeroed-output GEMMs (weight gradients): split K until the grid has ~2 waves of workgroups
    ksplit = (512 + tiles - 1) / tiles;
    if (ksplit > ktiles / 4) ksplit = ktiles / 4;
    if (ksplit < 1) ksplit = 1;
  }
  ka.ksplit = ksplit;
  ka.kt_per = (ktiles + ksplit - 1) / ksplit;
  dim3 grid((d.N + BN - 1) / BN, (d.M + BM - 1) / BM, d.batch * ksplit);
  if (grid.y > 65535 || grid.z > 65535) return set_error(MIMRL_ERR_ARG, "gemm: grid too large (M=%d batch=%d)", d.M, d.batch);
  if (bf16) hipLaunchKernelGGL(gemm_kernel<true>, grid, dim3(256), 0, s, ka);
  else hipLaunchKernelGGL(gemm_kernel<false>, grid, dim3(256), 0, s, ka);
  LAUNCH_CHECK();
  return MIMRL_OK;
}

}  // namespace mimrl
