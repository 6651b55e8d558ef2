// Tall-and-skinny GEMM core for the cfg3-class products (round 5):   C[M, N] = sum_s A_s[M, K_s] . W_s[N, K_s]^T (+ bias_n)
//   M = B*T rows (128 000 at cfg3), N = 128 .. 768, K = 256 .. 768, both operands STORED in 16 bits and k-contiguous.
// These products (GRU layer-1 input projection, dh0 = sum_dir dgx . W_ih, Model.py:254-255 and its autograd) are bound by operand
// and output BYTES, not by the matrix pipe: dh0 moves 0.66 GB for 100 GFLOP.  What the 128x128 register-staged kernel of gemm.hip lost
// there: A re-read per column tile, a ds_write pass per 32-wide k-tile, one barrier-to-barrier memory round trip per k-tile.
//
// Structure (one workgroup = 4 waves, TWO workgroups per CU so that one's epilogue stores overlap the other's k-loop):
//   * tile 256 (M) x 128 (N), BK = 32, a 3-stage LDS ring (72 KiB) filled by LDS-DMA (global_load_lds_dwordx4: no staging registers,
//     no ds_write) with TWO k-tiles in flight across raw s_barriers under a counted s_waitcnt vmcnt(6) -- never 0 inside the loop;
//   * LDS image [row][32 k] = 64-byte rows; the DMA writes lane-linear, so the bank swizzle (16-byte chunk ^= (row >> 2) & 3) is applied
//     to the per-lane SOURCE address and again on the fragment read (ds_read_b128, conflict-free per 16-lane group);
//   * wave tile 128 (M) x 64 (N) = 8 accumulators of v_mfma_f32_32x32x16_{bf16,f16}, 6 fragment reads per 8 MFMAs;
//   * the product is computed TRANSPOSED (D[n][m]: W is the MFMA A operand): a lane then owns 4 consecutive n of one output row per
//     register quad -- 16-byte stores (8-byte for an fp16 output) instead of sixteen 4-byte ones;
//   * workgroup ids are dealt so that all column tiles (and all inner-batch entries that share A, e.g. the two GRU directions) of one
//     256-row block run back to back on ONE XCD: A leaves HBM once.
// Dispatched by gemm() for (k-contiguous, k-contiguous) 16-bit-stored operands with a plain epilogue when M >= 16384.
#include "gemm.h"

namespace mimrl {

namespace {

constexpr int TBM = 256, TBN = 128, TBK = 32, TSTAGES = 3;
constexpr int TA_BYTES = TBM * TBK * 2, TW_BYTES = TBN * TBK * 2, TSTAGE_BYTES = TA_BYTES + TW_BYTES;   // 16 KiB + 8 KiB
constexpr int TLDS_BYTES = TSTAGES * TSTAGE_BYTES;                                                      // 72 KiB
constexpr int TLOADS = TA_BYTES / 4096 + TW_BYTES / 4096;                                               // 6 DMA instructions per wave and k-tile

struct TallArgs {
  const char* A[2]; const char* W[2];   // segment bases (bytes)
  long lda, ldw;                        // row pitches in elements (same for both segments)
  int kt0, kt;                          // k-tiles of segment 0 / in total
  char* C; long ldc; int c_f16;
  const float* bias;
  int M, N, mt, nt, nbi, nbo;
  long sa_bi[2], sa_bo[2], sw_bi[2], sw_bo[2], sc_bi, sc_bo, sbias_bi, sbias_bo;   // batch strides in elements
};

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_void;

// acc[ni][mi] register quad g of lane (r31, hh) = output row mrow + 32 mi, columns ncol + 32 ni + 8 g .. + 3
template <bool CF16, bool FULLN>
__device__ __forceinline__ void tall_store(const TallArgs& a, const f32x16 (&acc)[2][4], const float* __restrict__ bias, long ocb, int mrow, int ncol) {
#pragma unroll
  for (int ni = 0; ni < 2; ++ni) {
    f32x4 bv[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) bv[g] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (bias) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = ncol + 32 * ni + 8 * g;
        if constexpr (FULLN) bv[g] = *reinterpret_cast<const f32x4*>(bias + n);
        else
#pragma unroll
          for (int e = 0; e < 4; ++e) bv[g][e] = bias[n + e < a.N ? n + e : a.N - 1];
      }
    }
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      const int m = mrow + 32 * mi;
      if (m < a.M) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int n = ncol + 32 * ni + 8 * g;
          f32x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = acc[ni][mi][4 * g + e] + bv[g][e];
          if constexpr (CF16) {
            _Float16* c = reinterpret_cast<_Float16*>(a.C) + ocb + (long)m * a.ldc + n;
            if constexpr (FULLN) {
              f16x4 h; h[0] = to_f16_sat(v[0]); h[1] = to_f16_sat(v[1]); h[2] = to_f16_sat(v[2]); h[3] = to_f16_sat(v[3]);
              *reinterpret_cast<f16x4*>(c) = h;
            } else {
#pragma unroll
              for (int e = 0; e < 4; ++e) if (n + e < a.N) c[e] = to_f16_sat(v[e]);
            }
          } else {
            float* c = reinterpret_cast<float*>(a.C) + ocb + (long)m * a.ldc + n;
            if constexpr (FULLN) *reinterpret_cast<f32x4*>(c) = v;
            else {
#pragma unroll
              for (int e = 0; e < 4; ++e) if (n + e < a.N) c[e] = v[e];
            }
          }
        }
      }
    }
  }
}

template <bool F16>
__global__ __launch_bounds__(256, 2) void gemm_tall_kernel(TallArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // ---- tile of this workgroup: ids equal mod 8 share an XCD; inside one XCD run [row block][inner batch][column tile]
  const unsigned id = blockIdx.x, xcd = id & 7u, j = id >> 3;
  const unsigned per = (unsigned)(a.nt * a.nbi);
  const unsigned rbl = j / per, rem = j - rbl * per, bi = rem / (unsigned)a.nt, ntile = rem - bi * (unsigned)a.nt;
  const unsigned rb = rbl * 8u + xcd;
  if (rb >= (unsigned)(a.mt * a.nbo)) return;
  const unsigned bo = rb / (unsigned)a.mt, mtile = rb - bo * (unsigned)a.mt;
  const int m0 = (int)mtile * TBM, n0 = (int)ntile * TBN;
  const char* Ab[2]; const char* Wb[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    Ab[s] = a.A[s] + 2 * ((long)bo * a.sa_bo[s] + (long)bi * a.sa_bi[s]);
    Wb[s] = a.W[s] + 2 * ((long)bo * a.sw_bo[s] + (long)bi * a.sw_bi[s]);
  }
  // ---- per-lane DMA sources: instruction i of this wave fills the 1 KiB LDS piece q = 4 i + wave = rows 16 q .. 16 q + 15 (4 lanes per
  // 64-byte row); lane -> (row = 16 q + lane / 4, LDS chunk lane % 4) reads SOURCE chunk (lane % 4) ^ ((row >> 2) & 3)
  const int lrow = lane >> 2, csrc = (lane & 3) ^ ((lane >> 4) & 3);
  unsigned voa[4], vow[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int gm = m0 + 16 * (4 * i + wave) + lrow;
    gm = gm < a.M ? gm : a.M - 1;                       // ragged last block: clamped rows land in outputs nobody stores
    voa[i] = (unsigned)gm * (unsigned)(a.lda * 2) + (unsigned)csrc * 16u;
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int gn = n0 + 16 * (4 * i + wave) + lrow;
    gn = gn < a.N ? gn : a.N - 1;
    vow[i] = (unsigned)gn * (unsigned)(a.ldw * 2) + (unsigned)csrc * 16u;
  }
  auto issue = [&](int t, int stage) __attribute__((always_inline)) {
    const int seg = t >= a.kt0 ? 1 : 0;
    const long kb = (long)(t - (seg ? a.kt0 : 0)) * (TBK * 2);
    const char* As = Ab[seg] + kb;
    const char* Ws = Wb[seg] + kb;
    char* sa = smem + stage * TSTAGE_BYTES + wave * 1024;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_global_load_lds((gbl_void*)(As + voa[i]), (lds_void*)(sa + i * 4096), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_global_load_lds((gbl_void*)(Ws + vow[i]), (lds_void*)(sa + TA_BYTES + i * 4096), 16, 0, 0);
  };

  // ---- fragment addresses: lane holds row (lane & 31), k = 8 (lane >> 5) .. + 7 of a 16-wide k-step: chunk 2 ks + (lane >> 5), swizzled
  const int wm = wave >> 1, wn = wave & 1;
  const int r31 = lane & 31, hh = lane >> 5, sw = (lane >> 2) & 3;
  const int fa = (wm * 128 + r31) * 64, fw = TA_BYTES + (wn * 64 + r31) * 64;
  const int c0 = (hh ^ sw) * 16, c1 = ((2 + hh) ^ sw) * 16;

  f32x16 acc[2][4];   // [n sub-tile][m sub-tile]
#pragma unroll
  for (int ni = 0; ni < 2; ++ni)
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ni][mi][r] = 0.f;

  const int KT = a.kt;
  issue(0, 0);
  if (KT > 1) issue(1, 1);
  int st = 0, st2 = 2;   // stage of tile t / of tile t + 2
  for (int t = 0; t < KT; ++t) {
    // tile t has landed (this wave's pieces) once at most the 6 requests of tile t + 1 are outstanding; the barrier then makes every
    // wave's pieces visible AND says every wave has finished reading stage (t - 1) % 3, which the requests for tile t + 2 overwrite
    if (t + 1 < KT) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(TLOADS) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (t + 2 < KT) issue(t + 2, st2);
    const char* sb = smem + st * TSTAGE_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int cc = ks ? c1 : c0;
      bf16x8 wf[2], af[4];
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) wf[ni] = *reinterpret_cast<const bf16x8*>(sb + fw + ni * 2048 + cc);
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) af[mi] = *reinterpret_cast<const bf16x8*>(sb + fa + mi * 2048 + cc);
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
          if constexpr (F16)
            acc[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, wf[ni]), __builtin_bit_cast(f16x8, af[mi]), acc[ni][mi], 0, 0, 0);
          else
            acc[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ni], af[mi], acc[ni][mi], 0, 0, 0);
        }
    }
    st = st == 2 ? 0 : st + 1;
    st2 = st2 == 2 ? 0 : st2 + 1;
  }

  // ---- epilogue: D[i][j] with j = lane & 31 -> m, i = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5) -> n: registers 4 g .. 4 g + 3 are 4 consecutive n
  const long ocb = (long)bo * a.sc_bo + (long)bi * a.sc_bi;
  const float* bias = a.bias ? a.bias + (long)bo * a.sbias_bo + (long)bi * a.sbias_bi : nullptr;
  const int mrow = m0 + wm * 128 + r31, ncol = n0 + wn * 64 + 4 * hh;
  const bool full_n = n0 + TBN <= a.N;   // (one branch per workgroup picks the straight-line store loop: nothing uniform inside it)
  if (a.c_f16) {
    if (full_n) tall_store<true, true>(a, acc, bias, ocb, mrow, ncol);
    else tall_store<true, false>(a, acc, bias, ocb, mrow, ncol);
  } else {
    if (full_n) tall_store<false, true>(a, acc, bias, ocb, mrow, ncol);
    else tall_store<false, false>(a, acc, bias, ocb, mrow, ncol);
  }
}

}  // namespace

bool gemm_tall_ok(const GemmDesc& d) {
  static const bool off = getenv("MIMRL_NO_GEMM_TALL") != nullptr;   // tuning knob: the 128x128 register-staged kernels as before
  static const long min_m = getenv("MIMRL_GEMM_TALL_MIN_M") ? atol(getenv("MIMRL_GEMM_TALL_MIN_M")) : 16384;
  if (off || !d.a_bf16 || !d.b_bf16 || d.M < min_m) return false;
  if (d.sa_k != 1 || d.sb_k != 1 || d.sc_n != 1) return false;
  if (d.K % TBK != 0 || d.K <= 0 || d.N % 4 != 0 || d.N < 32) return false;
  if (d.bias_m || d.beta != 0.f || d.pre || d.gradact_u || d.atomic || d.colsum || d.act != ACT_NONE || d.alpha != 1.f) return false;
  if (d.a_gap_rows || d.a_pad4) return false;
  if (d.sa_m % 8 != 0 || d.sb_n % 8 != 0 || (d.c_f16 ? d.sc_m % 4 != 0 : d.sc_m % 4 != 0)) return false;
  if ((reinterpret_cast<uintptr_t>(d.A) | reinterpret_cast<uintptr_t>(d.B) | reinterpret_cast<uintptr_t>(d.C)) & 15) return false;
  if (d.bias_n && (reinterpret_cast<uintptr_t>(d.bias_n) & 15)) return false;
  if (d.sa_b % 8 || d.sa_bo % 8 || d.sb_b % 8 || d.sb_bo % 8 || d.sc_b % 4 || d.sc_bo % 4 || d.bias_n_b % 4 || d.bias_n_bo % 4) return false;
  if (d.A2) {
    if (d.sa2_k != 1 || d.sb2_k != 1 || d.K2 % TBK != 0 || d.K2 <= 0 || d.sa2_m != d.sa_m || d.sb2_n != d.sb_n) return false;
    if ((reinterpret_cast<uintptr_t>(d.A2) | reinterpret_cast<uintptr_t>(d.B2)) & 15) return false;
    if (d.sa2_b % 8 || d.sb2_b % 8 || d.batch_in > 0) return false;
  }
  // 32-bit byte offsets inside one batch entry
  if ((double)d.M * d.sa_m * 2 >= 4.0e9 || (double)d.N * d.sb_n * 2 >= 4.0e9) return false;
  return true;
}

int gemm_tall(hipStream_t s, const GemmDesc& d) {
  TallArgs a;
  a.A[0] = reinterpret_cast<const char*>(d.A); a.W[0] = reinterpret_cast<const char*>(d.B);
  a.A[1] = reinterpret_cast<const char*>(d.A2 ? d.A2 : d.A); a.W[1] = reinterpret_cast<const char*>(d.B2 ? d.B2 : d.B);
  a.lda = d.sa_m; a.ldw = d.sb_n;
  a.kt0 = d.K / TBK; a.kt = a.kt0 + (d.A2 ? d.K2 / TBK : 0);
  a.C = reinterpret_cast<char*>(d.C); a.ldc = d.sc_m; a.c_f16 = d.c_f16;
  a.bias = d.bias_n;
  a.M = d.M; a.N = d.N;
  a.mt = (d.M + TBM - 1) / TBM; a.nt = (d.N + TBN - 1) / TBN;
  if (d.batch_in > 0) { a.nbi = d.batch_in; a.nbo = d.batch / d.batch_in; }
  else { a.nbi = 1; a.nbo = d.batch; }   // a flat batch: every entry has its own A as far as this kernel knows
  for (int q = 0; q < 2; ++q) { a.sa_bi[q] = a.sa_bo[q] = a.sw_bi[q] = a.sw_bo[q] = 0; }
  if (d.batch_in > 0) {
    a.sa_bi[0] = d.sa_b; a.sa_bo[0] = d.sa_bo; a.sw_bi[0] = d.sb_b; a.sw_bo[0] = d.sb_bo;
    a.sc_bi = d.sc_b; a.sc_bo = d.sc_bo; a.sbias_bi = d.bias_n_b; a.sbias_bo = d.bias_n_bo;
  } else {
    a.sa_bo[0] = d.sa_b; a.sw_bo[0] = d.sb_b; a.sa_bo[1] = d.sa2_b; a.sw_bo[1] = d.sb2_b;
    a.sc_bi = 0; a.sc_bo = d.sc_b; a.sbias_bi = 0; a.sbias_bo = d.bias_n_b;
  }
  const long rbs = (long)a.mt * a.nbo;
  const long grid = ((rbs + 7) / 8) * 8 * a.nt * a.nbi;
  if (grid <= 0 || grid > 0x7fffffffL) return set_error(MIMRL_ERR_ARG, "gemm_tall: grid out of range");
  static bool attr = false;
  if (!attr) {
    HIPX(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tall_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, TLDS_BYTES));
    HIPX(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tall_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, TLDS_BYTES));
    attr = true;
  }
  if (d.f16) hipLaunchKernelGGL(gemm_tall_kernel<true>, dim3((unsigned)grid), dim3(256), TLDS_BYTES, s, a);
  else hipLaunchKernelGGL(gemm_tall_kernel<false>, dim3((unsigned)grid), dim3(256), TLDS_BYTES, s, a);
  LAUNCH_CHECK();
  return MIMRL_OK;
}

}  // namespace mimrl
