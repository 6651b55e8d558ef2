// L-axis MLP of a CubeMLP block for LONG sequences (see cube_long.h).
#include "cube_long.h"
#include "kmix_device.h"
#include <type_traits>

namespace mimrl {

namespace {

#ifndef MIMRL_LAXIS_LK
#define MIMRL_LAXIS_LK 32
#endif
constexpr int LK = MIMRL_LAXIS_LK;   // k (= input rows l) per chunk (32 or 64; one chunk in flight per workgroup while the previous one is multiplied)
constexpr int LWQ = LK / 32;         // 16-byte weight pieces per thread, tile and chunk
constexpr int LCT = 128;          // columns per workgroup (4 waves x 32)
constexpr int LXP = LCT + 32;     // x image [k][column]: pitch = 64 B mod 256 B -> the 4 k-rows of a transposed read hit distinct bank quarters (gemm.hip: FT::RCP)
constexpr int LWP = LK + 8;       // weight image [row][k]: 80-byte rows, 16-byte fragment reads conflict-free (gemm.hip: FT::KCP)
constexpr int LMT = 4;            // M tiles of 32 rows: two for W1 (hl <= 64), two for Wr (ol <= 64)

template <bool F16> struct T16;
template <> struct T16<true> { typedef _Float16 t; static __device__ __forceinline__ t cvt(float x) { return to_f16_sat(x); } };
template <> struct T16<false> { typedef __bf16 t; static __device__ __forceinline__ t cvt(float x) { return to_bf16(x); } };

// explicit-wait 16-byte loads (gemm.hip: gld16 / fast_wait): invisible to the compiler's wait-count pass, which across a loop's back edge
// waits for EVERYTHING in flight -- with them two chunks stay in flight per workgroup
__device__ __forceinline__ void lgld16(f32x4& d, const void* p) { asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(d) : "v"(p) : "memory"); }

template <bool F16>
__device__ __forceinline__ f32x16 mma(const bf16x8& a, const bf16x8& b, const f32x16& c) {
  if constexpr (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// Workgroup = (sample b, 128 columns c of the [il, C] slab), 4 waves x 32 columns.  One pass over the slab:
//   [U; R] = [W1; Wr] . X_b   (M = 64 + 64 rows of 16-bit weights, K = il streamed in 32-row chunks through LDS, fp32 accumulate)
//   H = act(U + b1);  Y = W2 . H + b2 + R  (the accumulator layout of H IS a valid MFMA B operand under a permutation of k that the W2
//   fragments follow: no LDS round trip);  Z = LayerNorm over the ol rows of every column (fp32, two-pass like colln_fwd_kernel).
// The GEMM chain read X_b twice (W1 . X and Wr . X are separate launches: 2 x 197 MB per forward tail at cfg3) and wrote / re-read Y in between.
template <bool F16>
__global__ __launch_bounds__(256, 2) void laxis_fwd_long_kernel(LAxisLongArgs a) {
  typedef typename T16<F16>::t E;
  __shared__ __attribute__((aligned(16))) E sx[2][LK][LXP];
  __shared__ __attribute__((aligned(16))) E sw[2][LMT * 32][LWP];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 31, lh = lane >> 5;
  const int b = blockIdx.y, c0 = blockIdx.x * LCT;
  const int il = a.il, hl = a.hl, ol = a.ol, C = a.C;
  const int nch = (il + LK - 1) / LK;
  const float* __restrict__ xb = a.x + (long)b * il * C + c0;

  // staging registers of one chunk: this thread's (row, 4 columns) pieces of X and (row, 4 k) pieces of [W1; Wr]
  constexpr int NLD = LK / 8 + LMT * LWQ;   // loads per thread and chunk
  f32x4 rxs[2][LK / 8], rws[2][LMT][LWQ];   // two staging sets: chunks kc + 1 and kc + 2 are in flight while chunk kc is multiplied
  auto request = [&](auto S, int kc_) __attribute__((always_inline)) {
    f32x4 (&rx)[LK / 8] = rxs[decltype(S)::value];
    f32x4 (&rw)[LMT][LWQ] = rws[decltype(S)::value];
    const int kc = kc_ < nch ? kc_ : nch - 1;            // unconditional (past the end: the last chunk again, never published): the count stays exact
    const int k0 = kc * LK;
#pragma unroll
    for (int j = 0; j < LK / 8; ++j) {
      const int k = k0 + (tid >> 5) + 8 * j;
      lgld16(rx[j], xb + (long)min(k, il - 1) * C + (tid & 31) * 4);
    }
#pragma unroll
    for (int j = 0; j < LMT; ++j)     // tile j: rows 32 j .. 32 j + 31 of the stacked matrix [W1 (64 rows); Wr (64 rows)]
#pragma unroll
      for (int q = 0; q < LWQ; ++q) {
        const int idx = tid + 256 * q, row = idx / (LK / 4), kq = k0 + (idx % (LK / 4)) * 4;
        const float* src = j < 2 ? a.w1 + (long)min(32 * j + row, hl - 1) * il : a.wr + (long)min(32 * (j - 2) + row, ol - 1) * il;
        lgld16(rw[j][q], src + min(kq, il - 4));       // (il % 4 == 0: a k quad is inside or outside as a whole)
      }
  };
  // `newer`: loads issued after this set's (the other set's NLD, or 0)
  auto publish = [&](auto S, auto NEWER, int kc, int buf) __attribute__((always_inline)) {
    f32x4 (&rx)[LK / 8] = rxs[decltype(S)::value];
    f32x4 (&rw)[LMT][LWQ] = rws[decltype(S)::value];
    asm volatile("s_waitcnt vmcnt(%0)" : : "n"(decltype(NEWER)::value) : "memory");
#pragma unroll
    for (int j = 0; j < LK / 8; ++j) asm volatile("" : "+v"(rx[j]));
#pragma unroll
    for (int j = 0; j < LMT; ++j)
#pragma unroll
      for (int q = 0; q < LWQ; ++q) asm volatile("" : "+v"(rw[j][q]));
    const int k0 = kc * LK;
    typedef E E4 __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int j = 0; j < LK / 8; ++j) {
      const int kr = (tid >> 5) + 8 * j;
      const bool ok = k0 + kr < il;
      E4 p;
      p[0] = T16<F16>::cvt(ok ? rx[j][0] : 0.f); p[1] = T16<F16>::cvt(ok ? rx[j][1] : 0.f);
      p[2] = T16<F16>::cvt(ok ? rx[j][2] : 0.f); p[3] = T16<F16>::cvt(ok ? rx[j][3] : 0.f);
      *reinterpret_cast<E4*>(&sx[buf][kr][(tid & 31) * 4]) = p;
    }
#pragma unroll
    for (int j = 0; j < LMT; ++j)
#pragma unroll
      for (int q = 0; q < LWQ; ++q) {
        const int idx = tid + 256 * q, row = idx / (LK / 4), kq = (idx % (LK / 4)) * 4;
        const bool ok = (j < 2 ? 32 * j + row < hl : 32 * (j - 2) + row < ol) && k0 + kq < il;
        E4 p;
        p[0] = T16<F16>::cvt(ok ? rw[j][q][0] : 0.f); p[1] = T16<F16>::cvt(ok ? rw[j][q][1] : 0.f);
        p[2] = T16<F16>::cvt(ok ? rw[j][q][2] : 0.f); p[3] = T16<F16>::cvt(ok ? rw[j][q][3] : 0.f);
        *reinterpret_cast<E4*>(&sw[buf][32 * j + row][kq]) = p;
      }
  };

  f32x16 acc[LMT];
#pragma unroll
  for (int m = 0; m < LMT; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;

  using S0 = std::integral_constant<int, 0>; using S1 = std::integral_constant<int, 1>;
  using N0 = std::integral_constant<int, 0>; using NN = std::integral_constant<int, NLD>;
  auto multiply = [&](int cur) __attribute__((always_inline)) {
#pragma unroll
    for (int s = 0; s < LK / 16; ++s) {
      // B fragment: this wave's 32 columns, k = 16 s + 8 lh .. + 7, read transposed from the [k][column] image (gemm.hip: fast_frag, RC)
      typedef __attribute__((address_space(3))) bf16x4 lds4;
      const int j = lane & 15, q = j >> 2, p = j & 3;
      const E* xa = &sx[cur][s * 16 + 8 * (lane >> 5) + q][wave * 32 + 16 * ((lane >> 4) & 1) + 4 * p];
      const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4*)(xa));
      const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4*)(xa + 4 * LXP));
      bf16x8 bf;
      bf[0] = lo[0]; bf[1] = lo[1]; bf[2] = lo[2]; bf[3] = lo[3]; bf[4] = hi[0]; bf[5] = hi[1]; bf[6] = hi[2]; bf[7] = hi[3];
#pragma unroll
      for (int m = 0; m < LMT; ++m) {
        const bf16x8 af = *reinterpret_cast<const bf16x8*>(&sw[cur][32 * m + lr][s * 16 + 8 * lh]);
        acc[m] = mma<F16>(af, bf, acc[m]);
      }
    }
  };
  // chunk kc lives in staging set kc & 1 and LDS buffer kc & 1.  Every iteration issues exactly NLD loads and waits with NLD newer ones
  // behind the set it publishes; two iterations per trip so that the sets are compile-time registers.
  request(S0{}, 0);
  request(S1{}, 1);
  publish(S0{}, NN{}, 0, 0);
  __syncthreads();
  for (int kc = 0; kc < nch; kc += 2) {
    request(S0{}, kc + 2);
    multiply(0);
    publish(S1{}, NN{}, kc + 1 < nch ? kc + 1 : nch - 1, 1);       // chunk kc + 1 (past the end: a copy of the last chunk nobody multiplies)
    __syncthreads();
    if (kc + 1 >= nch) break;
    request(S1{}, kc + 3);
    multiply(1);
    publish(S0{}, NN{}, kc + 2 < nch ? kc + 2 : nch - 1, 0);
    __syncthreads();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the trailing (duplicate) requests: nothing may land in a dead register

  // W2 fragments for the second product (requested behind the main loop: held across it they cost 32 VGPRs and the second workgroup per CU), k ordered as the accumulator layout of H delivers it: k-step ks covers rows
  // base .. base + 15 with base = 16 ks; position j of lane half lh is row base + (j & 3) + 8 (j >> 2) + 4 lh
  bf16x8 w2f[2][4];   // [output tile][k-step]
#pragma unroll
  for (int ot = 0; ot < 2; ++ot)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      bf16x8 f;
      const int o = 32 * ot + lr;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int h = 16 * ks + (j & 3) + 8 * (j >> 2) + 4 * lh;
        const float v = a.w2[(long)min(o, ol - 1) * hl + min(h, hl - 1)];
        f[j] = __builtin_bit_cast(__bf16, T16<F16>::cvt((o < ol && h < hl) ? v : 0.f));
      }
      w2f[ot][ks] = f;
    }

  // ---- epilogue: accumulator element r of lane (lr, lh) = row (r & 3) + 8 (r >> 2) + 4 lh of its tile, column c0 + 32 wave + lr
  const int cc = c0 + wave * 32 + lr;
  bf16x8 hf[4];
  act_dispatch(a.act, [&](auto AT) __attribute__((always_inline)) {
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int h = 32 * m + (r & 3) + 8 * (r >> 2) + 4 * lh;
        const float u = acc[m][r] + (a.b1 ? a.b1[min(h, hl - 1)] : 0.f);
        const float hv = act_apply_c<decltype(AT)::value>(a.act, u);
        if (h < hl && a.u) {
          a.u[((long)b * hl + h) * C + cc] = u;
          a.h[((long)b * hl + h) * C + cc] = hv;
        }
        // k-step 2 m + (r >> 3), position r & 7 (see w2f)
        hf[2 * m + (r >> 3)][r & 7] = __builtin_bit_cast(__bf16, T16<F16>::cvt(h < hl ? hv : 0.f));
      }
  });
  f32x16 y[2] = {acc[2], acc[3]};
#pragma unroll
  for (int ot = 0; ot < 2; ++ot)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) y[ot] = mma<F16>(w2f[ot][ks], hf[ks], y[ot]);
  // rows o = 32 ot + (r & 3) + 8 (r >> 2) + 4 lh < ol of this column: + b2, LayerNorm over them (the other lane half holds the rest)
  float s1 = 0.f;
#pragma unroll
  for (int ot = 0; ot < 2; ++ot)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int o = 32 * ot + (r & 3) + 8 * (r >> 2) + 4 * lh;
      y[ot][r] += a.b2 ? a.b2[min(o, ol - 1)] : 0.f;
      s1 += o < ol ? y[ot][r] : 0.f;
    }
  s1 += __shfl_xor(s1, 32, 64);
  const float mu = s1 / ol;
  float s2 = 0.f;
#pragma unroll
  for (int ot = 0; ot < 2; ++ot)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int o = 32 * ot + (r & 3) + 8 * (r >> 2) + 4 * lh;
      const float d = y[ot][r] - mu;
      s2 += o < ol ? d * d : 0.f;
    }
  s2 += __shfl_xor(s2, 32, 64);
  const float rs = rsqrtf(s2 / ol + LN_EPS);
  if (a.mean && lh == 0) { a.mean[(long)b * C + cc] = mu; a.rstd[(long)b * C + cc] = rs; }
#pragma unroll
  for (int ot = 0; ot < 2; ++ot)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int o = 32 * ot + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (o < ol) {
        const long oi = ((long)b * ol + o) * C + cc;
        if (a.y) a.y[oi] = y[ot][r];
        a.z[oi] = (y[ot][r] - mu) * rs * a.g[o] + a.be[o];
      }
    }
}

}  // namespace

bool laxis_fwd_long_supported(int il, int hl, int ol, int C) {
  return il > 64 && il % 4 == 0 && hl >= 1 && hl <= 64 && ol >= 1 && ol <= 64 && C % LCT == 0;
}

int laxis_fwd_long(hipStream_t s, const LAxisLongArgs& a, bool f16) {
  if (!laxis_fwd_long_supported(a.il, a.hl, a.ol, a.C)) return set_error(MIMRL_ERR_ARG, "laxis_fwd_long: unsupported shape");
  if (!a.wr || !a.z || ((a.u != nullptr) != (a.h != nullptr))) return set_error(MIMRL_ERR_ARG, "laxis_fwd_long: needs the residual projection; u and h together");
  if (f16) hipLaunchKernelGGL(laxis_fwd_long_kernel<true>, dim3(a.C / LCT, a.B), dim3(256), 0, s, a);
  else hipLaunchKernelGGL(laxis_fwd_long_kernel<false>, dim3(a.C / LCT, a.B), dim3(256), 0, s, a);
  LAUNCH_CHECK();
  return MIMRL_OK;
}

}  // namespace mimrl
