// Fused ReLU-MLP stacks (see mlp_fused.h).  One workgroup = 32 rows of one group through every layer.
//   MFMA v_mfma_f32_32x32x16_bf16: M = the 32 rows (A-fragments from the LDS activation tile), N = 32 output columns
//   per tile (tiles dealt to the 4 waves), K = the layer's input width.  B-fragments come straight from global memory:
//   forward  W[n, k..k+8)   two 16-byte loads per lane and k-step (row n of the weight matrix is k-contiguous);
//   backward W[n..n+8, k]   eight 4-byte loads per lane (a half-wave reads 128 contiguous bytes of each row),
//   double-buffered in registers so that the next chunk of weights is in flight while the current one is multiplied.
#include "mlp_fused.h"

#include <cstdlib>

#include <type_traits>

namespace mimrl {

namespace {

typedef std::integral_constant<int, 0> I0;
typedef std::integral_constant<int, 1> I1;
constexpr int RT = 32;                        // rows per workgroup
constexpr int LDA = MLPF_MAX_WIDTH + 8;       // bf16 pitch: 784 B = odd multiple of 16 B -> conflict-free ds_read_b128

__device__ __forceinline__ bf16x8 pack8(const float4& x, const float4& y) {
  bf16x8 p;
  p[0] = to_bf16(x.x); p[1] = to_bf16(x.y); p[2] = to_bf16(x.z); p[3] = to_bf16(x.w);
  p[4] = to_bf16(y.x); p[5] = to_bf16(y.y); p[6] = to_bf16(y.z); p[7] = to_bf16(y.w);
  return p;
}

// rows [r0, r0+32) x [0, width) of a [.., width] fp32 matrix -> bf16 tile; rows >= rows_valid and columns up to the
// next multiple of 16 are zero
template <int NT = 256>
__device__ __forceinline__ void load_tile_bf16(const float* __restrict__ src, long rowbase, int r0, int rows_valid, int width,
                                               __bf16 (*t)[LDA], int tid) {
  const int w16 = (width + 15) & ~15;
  if ((width & 3) == 0) {
    const int q = w16 / 4;
    for (int i = tid; i < RT * q; i += NT) {
      const int r = i / q, c = (i - r * q) * 4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (r0 + r < rows_valid && c < width) v = *reinterpret_cast<const float4*>(src + (rowbase + r) * width + c);
      bf16x4 p; p[0] = to_bf16(v.x); p[1] = to_bf16(v.y); p[2] = to_bf16(v.z); p[3] = to_bf16(v.w);
      *reinterpret_cast<bf16x4*>(&t[r][c]) = p;
    }
  } else {
    for (int i = tid; i < RT * w16; i += NT) {
      const int r = i / w16, c = i - r * w16;
      t[r][c] = to_bf16((r0 + r < rows_valid && c < width) ? src[(rowbase + r) * width + c] : 0.f);
    }
  }
}

// Forward.  Weight rows are k-contiguous, the MFMA B-fragment wants (row n = lane&31, 8 k-values): loading fragments
// directly makes every wave-instruction touch 32 different cache lines.  Instead each wave pulls its OWN two 32-row
// tiles in fully coalesced 1 KiB instructions (16 lanes x 16 B per row, 4 rows per instruction), rounds them to bf16
// into a wave-private LDS slab and reads the fragments back from there; no other wave needs those rows (single 32-row
// M tile), so the hand-off needs no workgroup barrier.  Two chunks of 64 k-values are kept in flight in registers.
constexpr int WCH = 64;                       // k-values per staged chunk
constexpr int WLD = WCH + 8;                  // bf16 pitch of a staged weight tile (144 B)

__global__ __launch_bounds__(256) void mlp_fwd_kernel(MlpFusedArgs a) {
  __shared__ __attribute__((aligned(16))) __bf16 sa[2][RT][LDA];
  __shared__ __attribute__((aligned(16))) __bf16 wl[4][2][2][32][WLD];   // [wave][buffer][tile][row][k]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // workgroups are dealt round-robin to the 8 XCDs: give every group (one weight set) to ONE XCD so that its row tiles
  // share the weight lines in that XCD's L2 (grid.x = 8 * tiles * ceil(nb / 8), surplus ids exit)
  const int tiles = (a.rows + RT - 1) / RT;
  const int xslot = blockIdx.x >> 3;
  const int g = (blockIdx.x & 7) + 8 * (xslot / tiles), r0 = (xslot % tiles) * RT;
  if (g >= a.nb) return;
  const long rowbase = (long)g * a.brows + r0;
  const int lr = lane & 31, lh = lane >> 5;
  const int srow = lane >> 4, sk = (lane & 15) * 4;          // staging map: 4 rows x 64 k per instruction
  load_tile_bf16(a.in, rowbase, r0, a.rows, a.dims[0], sa[0], tid);
  __syncthreads();
  int cur = 0;
  for (int l = 0; l < a.nl; ++l) {
    const int K = a.dims[l], N = a.dims[l + 1];
    const float* __restrict__ W = a.W[l] + (long)g * a.pstride;
    const float* __restrict__ bias = a.b[l] + (long)g * a.pstride;
    const bool last = l == a.nl - 1;
    float* __restrict__ dst = last ? a.out : a.act[l];
    const int ntiles = (N + 31) / 32;
    const int nchunk = K / WCH;                  // K is a multiple of 64
    for (int nt0 = wave; nt0 < ntiles; nt0 += 8) {
      const int nt1 = nt0 + 4;
      const bool two = nt1 < ntiles;
      f32x16 acc0, acc1;
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
      float4 rg[2][2][8];                        // [register buffer][tile][instruction]
      auto fetch = [&](int c, auto S) {
        constexpr int s = decltype(S)::value;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int row = i * 4 + srow;
          rg[s][0][i] = *reinterpret_cast<const float4*>(W + (long)min(nt0 * 32 + row, N - 1) * K + c * WCH + sk);
          if (two) rg[s][1][i] = *reinterpret_cast<const float4*>(W + (long)min(nt1 * 32 + row, N - 1) * K + c * WCH + sk);
        }
      };
      auto stage = [&](int lb, auto S) {           // registers -> this wave's LDS slab `lb`
        constexpr int s = decltype(S)::value;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          if (t == 1 && !two) break;
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const float4 v = rg[s][t][i];
            bf16x4 p; p[0] = to_bf16(v.x); p[1] = to_bf16(v.y); p[2] = to_bf16(v.z); p[3] = to_bf16(v.w);
            *reinterpret_cast<bf16x4*>(&wl[wave][lb][t][i * 4 + srow][sk]) = p;
          }
        }
      };
      auto mult = [&](int c, int lb) {
#pragma unroll
        for (int u = 0; u < WCH / 16; ++u) {
          const bf16x8 af = *reinterpret_cast<const bf16x8*>(&sa[cur][lr][c * WCH + u * 16 + 8 * lh]);
          const bf16x8 b0 = *reinterpret_cast<const bf16x8*>(&wl[wave][lb][0][lr][u * 16 + 8 * lh]);
          acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, b0, acc0, 0, 0, 0);
          if (two) {
            const bf16x8 b1 = *reinterpret_cast<const bf16x8*>(&wl[wave][lb][1][lr][u * 16 + 8 * lh]);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, b1, acc1, 0, 0, 0);
          }
        }
      };
      // chunk c lives in register buffer c&1 until staged into LDS slab c&1; chunks c+1 (and c+2) are in flight meanwhile
      fetch(0, I0{});
      if (nchunk > 1) fetch(1, I1{});
      for (int c = 0; c < nchunk; c += 2) {
        stage(0, I0{});
        if (c + 2 < nchunk) fetch(c + 2, I0{});
        mult(c, 0);
        if (c + 1 < nchunk) {
          stage(1, I1{});
          if (c + 3 < nchunk) fetch(c + 3, I1{});
          mult(c + 1, 1);
        }
      }
      auto finish = [&](const f32x16& acc, int nt) {
        const int n = nt * 32 + lr;
        if (n >= N) return;
        const float bn = bias[n];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = (r & 3) + 8 * (r >> 2) + 4 * lh;
          float v = acc[r] + bn;
          if (!last) v = fmaxf(v, 0.f);
          if (r0 + m < a.rows) dst[(rowbase + m) * N + n] = v;
          if (!last) sa[cur ^ 1][m][n] = to_bf16(v);
        }
      };
      finish(acc0, nt0);
      if (two) finish(acc1, nt1);
    }
    __syncthreads();
    cur ^= 1;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Image variants: weights come from bf16 images (half the bytes, no conversion).  The data-gradient chain is the SAME
// loop as the forward pass run on the transposed images: out[r, k] = sum_n dz[r, n] WT[k, n], i.e. "rows" of the staged
// operand are output columns and are contiguous along the reduction axis -- coalesced 16-byte loads in both directions.
// ---------------------------------------------------------------------------------------------------------------
// one pair (or single) of 32-column output tiles of one layer: acc += A[32 x K] . W[tile rows, K]^T, W = bf16 image rows
__device__ __forceinline__ void img_tiles(f32x16& accA, f32x16& accB, int nt0, bool two, const __bf16* __restrict__ W, int N, int K,
                                          const __bf16 (*sa_cur)[LDA], __bf16 (*wlw)[2][32][WLD], int lr, int lh, int srow, int sk) {
        const int nt1 = nt0 + 4;
        const int nchunk = (K + WCH - 1) / WCH;
        uint4 rgA[2][4], rgB[2][4];
        auto fetch = [&](int c, auto S) {
          constexpr int s = decltype(S)::value;
          const int kk = c * WCH + sk;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int row = i * 8 + srow;
            rgA[s][i] = make_uint4(0u, 0u, 0u, 0u);
            rgB[s][i] = make_uint4(0u, 0u, 0u, 0u);
            if (kk < K) {      // (a ternary between the load and a local zero would go through scratch and a flat load)
              rgA[s][i] = *reinterpret_cast<const uint4*>(W + (long)min(nt0 * 32 + row, N - 1) * K + kk);
              if (two) rgB[s][i] = *reinterpret_cast<const uint4*>(W + (long)min(nt1 * 32 + row, N - 1) * K + kk);
            }
          }
        };
        auto stage = [&](int lb, auto S) {
          constexpr int s = decltype(S)::value;
#pragma unroll
          for (int i = 0; i < 4; ++i) *reinterpret_cast<uint4*>(&wlw[lb][0][i * 8 + srow][sk]) = rgA[s][i];
          if (two) {
#pragma unroll
            for (int i = 0; i < 4; ++i) *reinterpret_cast<uint4*>(&wlw[lb][1][i * 8 + srow][sk]) = rgB[s][i];
          }
        };
        auto mult = [&](int c, int lb) {
#pragma unroll
          for (int u = 0; u < WCH / 16; ++u) {
            if (c * WCH + u * 16 >= K) break;
            const bf16x8 af = *reinterpret_cast<const bf16x8*>(&sa_cur[lr][c * WCH + u * 16 + 8 * lh]);
            const bf16x8 b0 = *reinterpret_cast<const bf16x8*>(&wlw[lb][0][lr][u * 16 + 8 * lh]);
            accA = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, b0, accA, 0, 0, 0);
            if (two) {
              const bf16x8 b1 = *reinterpret_cast<const bf16x8*>(&wlw[lb][1][lr][u * 16 + 8 * lh]);
              accB = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, b1, accB, 0, 0, 0);
            }
          }
        };
        fetch(0, I0{});
        if (nchunk > 1) fetch(1, I1{});
        for (int c = 0; c < nchunk; c += 2) {
          stage(0, I0{});
          if (c + 2 < nchunk) fetch(c + 2, I0{});
          mult(c, 0);
          if (c + 1 < nchunk) {
            stage(1, I1{});
            if (c + 3 < nchunk) fetch(c + 3, I1{});
            mult(c + 1, 1);
          }
        }
}

// the same pair of output tiles with B-fragments loaded straight from the (L2-resident) image: for stacks with
// thousands of row tiles (the concat critic: B^2 rows per estimator) every workgroup re-reads the same small weights,
// so nothing is staged and nothing has to be hidden -- all loads of a layer are issued up front (K <= 256)
__device__ __forceinline__ void img_tiles_direct(f32x16& accA, f32x16& accB, int nt0, bool two, const __bf16* __restrict__ W,
                                                 int N, int K, const __bf16 (*sa_cur)[LDA], int lr, int lh) {
  const __bf16* __restrict__ w0 = W + (long)min(nt0 * 32 + lr, N - 1) * K + 8 * lh;
  const __bf16* __restrict__ w1 = W + (long)min((nt0 + 4) * 32 + lr, N - 1) * K + 8 * lh;
  bf16x8 b0[16], b1[16];
#pragma unroll
  for (int ks = 0; ks < 16; ++ks) {
    if (ks * 16 < K) {
      b0[ks] = *reinterpret_cast<const bf16x8*>(w0 + ks * 16);
      if (two) b1[ks] = *reinterpret_cast<const bf16x8*>(w1 + ks * 16);
    }
  }
#pragma unroll
  for (int ks = 0; ks < 16; ++ks) {
    if (ks * 16 < K) {
      const bf16x8 af = *reinterpret_cast<const bf16x8*>(&sa_cur[lr][ks * 16 + 8 * lh]);
      accA = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, b0[ks], accA, 0, 0, 0);
      if (two) accB = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, b1[ks], accB, 0, 0, 0);
    }
  }
}

template <bool BWD, bool DIRECT>
__global__ __launch_bounds__(256) void mlp_img_kernel(MlpFusedArgs a) {
  __shared__ __attribute__((aligned(16))) __bf16 sa[2][RT][LDA];
  __shared__ __attribute__((aligned(16))) __bf16 wl[DIRECT ? 1 : 4][2][2][32][WLD];   // [wave][buffer][tile][row][k]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tiles = (a.rows + RT - 1) / RT;
  const int xslot = blockIdx.x >> 3;
  // staged variant: a group (one weight set) is pinned to one XCD; direct variant (thousands of tiles per group, weights
  // hot in every L2): plain (tile, group) grid over the whole chip
  const int g = DIRECT ? blockIdx.y : (blockIdx.x & 7) + 8 * (xslot / tiles);
  const int r0 = (DIRECT ? blockIdx.x : xslot % tiles) * RT;
  if (g >= a.nb) return;
  const long rowbase = (long)g * a.brows + r0;
  const int lr = lane & 31, lh = lane >> 5;
  const int srow = lane >> 3, sk = (lane & 7) * 8;            // staging map: 8 rows x 64 k (bf16) per 16-byte instruction
  load_tile_bf16(BWD ? a.dout : a.in, rowbase, r0, a.rows, BWD ? a.dims[a.nl] : a.dims[0], sa[0], tid);
  if (BWD && a.db_top && tid < a.dims[a.nl]) {   // top-layer bias gradient: column sums of this tile's dout rows (fp32 source, L2-hot)
    const int w = a.dims[a.nl];
    const int nr = min(RT, a.rows - r0);
    float s = 0.f;
    for (int r = 0; r < nr; ++r) s += a.dout[(rowbase + r) * w + tid];
    acc_add(a.db_top + (long)g * a.pstride + tid, s);
  }
  __syncthreads();
  int cur = 0;
  for (int step = 0; step < a.nl; ++step) {
    const int l = BWD ? a.nl - 1 - step : step;
    const int K = BWD ? a.dims[l + 1] : a.dims[l];            // reduction width
    const int N = BWD ? a.dims[l] : a.dims[l + 1];            // output width
    float* __restrict__ dst = BWD ? (l > 0 ? a.dz[l] : a.din) : (l == a.nl - 1 ? a.out : a.act[l]);
    if (BWD && !dst) break;
    const __bf16* __restrict__ W = (BWD ? a.WbT[l] : a.Wb[l]) + (long)g * a.pstride;   // [N][K], K contiguous
    const bool last = BWD ? l == 0 : l == a.nl - 1;
    const int ntiles = (N + 31) / 32;
    f32x16 acc0, acc1, acc2;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; acc2[r] = 0.f; }
    // ReLU mask of the data-gradient chain = the forward activations of the layer below: requested BEFORE the products
    // (it does not depend on them); loaded in the epilogue it was a dependent memory round trip per tile and layer --
    // 25 of the kernel's 57 us.  Unconditional, clamped addresses (a guarded load is a branch + vmcnt(0)).
    const float* __restrict__ mask = (BWD && l > 0 && !(a.dbg & 1)) ? a.act[l - 1] : nullptr;
    float mk0[16], mk1[16], mk2[16];
    auto prefetch_mask = [&](float (&mk)[16], int nt) {
      const int n = min(nt * 32 + lr, N - 1);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = min((r & 3) + 8 * (r >> 2) + 4 * lh, a.rows - 1 - r0);
        mk[r] = mask[(rowbase + m) * N + n];
      }
    };
    if (BWD && mask) { prefetch_mask(mk0, wave); prefetch_mask(mk1, wave + 4); prefetch_mask(mk2, wave + 8); }
    // same for the bias of the forward pass: one value per lane and tile, but a dependent round trip each in the epilogue
    const float* __restrict__ bias = BWD ? nullptr : a.b[l] + (long)g * a.pstride;
    float bn3[3] = {0.f, 0.f, 0.f};
    if (bias) {
#pragma unroll
      for (int q = 0; q < 3; ++q) bn3[q] = bias[min((wave + 4 * q) * 32 + lr, N - 1)];
    }
    if (K < 16) {
      // degenerate reduction (the 2-logit top layer of the CMI classifier, backward): plain FMAs
      auto small = [&](f32x16& acc, int nt) {
        const int n = nt * 32 + lr;
        if (nt >= ntiles || n >= N) return;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = (r & 3) + 8 * (r >> 2) + 4 * lh;
          float v = 0.f;
          for (int kk = 0; kk < K; ++kk) v += (float)sa[cur][m][kk] * (float)W[(long)n * K + kk];
          acc[r] = v;
        }
      };
      small(acc0, wave); small(acc1, wave + 4); small(acc2, wave + 8);
    } else {
      // up to 3 output tiles per wave (N <= 384): a pair, then a single -- same staging as mlp_fwd_kernel
      f32x16 dummy = acc2;
      if constexpr (DIRECT) {
        if (wave < ntiles) img_tiles_direct(acc0, acc1, wave, wave + 4 < ntiles, W, N, K, sa[cur], lr, lh);
        if (wave + 8 < ntiles) img_tiles_direct(acc2, dummy, wave + 8, false, W, N, K, sa[cur], lr, lh);
      } else {
        if (wave < ntiles) img_tiles(acc0, acc1, wave, wave + 4 < ntiles, W, N, K, sa[cur], wl[wave], lr, lh, srow, sk);
        if (wave + 8 < ntiles) img_tiles(acc2, dummy, wave + 8, false, W, N, K, sa[cur], wl[wave], lr, lh, srow, sk);
      }
    }
    // epilogue
    float* __restrict__ db = (BWD && l > 0 && a.db[l - 1] && !(a.dbg & 2)) ? a.db[l - 1] + (long)g * a.pstride : nullptr;
    auto finish = [&](const f32x16& acc, int nt, const float (&mk)[16], const float bn) {
      const int n = nt * 32 + lr;
      if (nt >= ntiles || n >= N) return;
      float csum = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = (r & 3) + 8 * (r >> 2) + 4 * lh;
        const bool ok = r0 + m < a.rows;
        float v = acc[r] + bn;
        if (BWD) { if (mask) v = (ok && mk[r] > 0.f) ? v : 0.f; }
        else if (!last) v = fmaxf(v, 0.f);
        if (ok && !(a.dbg & 4)) dst[(rowbase + m) * N + n] = v;
        if (!last) sa[cur ^ 1][m][n] = to_bf16(v);
        csum += v;
      }
      if (db) {
        csum += __shfl_xor(csum, 32, 64);
        if (lh == 0) acc_add(&db[n], csum);
      }
    };
    finish(acc0, wave, mk0, bn3[0]); finish(acc1, wave + 4, mk1, bn3[1]); finish(acc2, wave + 8, mk2, bn3[2]);
    __syncthreads();
    cur ^= 1;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// 8-wave variant of the image kernel for SMALL stacks (tens of workgroups: the separable critic towers and the CMI classifiers
// at B <= 256), where the 4-wave kernel is a chain of memory round trips: per layer every wave streamed two 32-row weight tiles
// through its LDS slab, two 64-k chunks in flight, i.e. two dependent L2 round trips per layer and four layers in a row.
// Here a wave owns ONE 32-column output tile per layer (tile = wave; a second one, wave + 8, only where N = 384), holds the whole
// [32 x K] weight tile of the CURRENT layer in registers (K <= 384: 24 x 16 B per lane) and re-fills those registers with the
// NEXT layer's tile while the current one is staged and multiplied chunk by chunk -- weights do not depend on activations, so
// the only exposed round trip is the first layer's.
// ---------------------------------------------------------------------------------------------------------------
constexpr int W8_MAXCH = MLPF_MAX_WIDTH / WCH;   // 6 chunks of 64 k
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
// compile-time loop: the register tile below must only ever be indexed with constants (a runtime index sends it to scratch)
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}

// tools/hw/mlp_phase.hip compiles this file with MLP_PHASE_PROBE: workgroup 0 leaves wall_clock64 ticks (100 MHz) at the points below
#ifdef MLP_PHASE_PROBE
__device__ long long g_mlp_phase[64];
#define PHASE(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_mlp_phase[i] = (long long)wall_clock64(); } while (0)
#else
#define PHASE(i) do { } while (0)
#endif

template <bool BWD, int NCH>   // NCH: 64-k chunks held in registers (4: every reduction width <= 256; 6: <= 384)
__global__ __launch_bounds__(512) void mlp_img8_kernel(MlpFusedArgs a) {
  __shared__ __attribute__((aligned(16))) __bf16 sa[2][RT][LDA];
  __shared__ __attribute__((aligned(16))) __bf16 wl[8][2][32][WLD];      // [wave][buffer][row][k]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tiles = (a.rows + RT - 1) / RT;
  const int xslot = blockIdx.x >> 3;
  const int g = (blockIdx.x & 7) + 8 * (xslot / tiles);      // a group (one weight set) is pinned to one XCD
  const int r0 = (xslot % tiles) * RT;
  if (g >= a.nb) return;
  const long rowbase = (long)g * a.brows + r0;
  const int lr = lane & 31, lh = lane >> 5;
  const int srow = lane >> 3, sk = (lane & 7) * 8;            // staging map: 8 rows x 64 k (bf16) per 16-byte instruction
  // ragged last row tile: stores are guarded; the mask loads use a clamped wave-uniform row (see the mask loop)
  const int rows_here = min(RT, a.rows - r0);

  // geometry of the layer processed at `step`
  auto layer_of = [&](int step) { return BWD ? a.nl - 1 - step : step; };
  auto K_of = [&](int l) { return BWD ? a.dims[l + 1] : a.dims[l]; };
  auto N_of = [&](int l) { return BWD ? a.dims[l] : a.dims[l + 1]; };
  auto W_of = [&](int l) { return (BWD ? a.WbT[l] : a.Wb[l]) + (long)g * a.pstride; };
  auto has_target = [&](int l) { return !BWD || (l > 0 ? a.dz[l] != nullptr : a.din != nullptr); };

  // this wave's weight tile of one layer: chunk c = k in [64c, 64c+64), rows i*8 + srow.  Six separate arrays picked at compile
  // time, of a native vector type: HIP's uint4 is a struct whose assignment is a memcpy, and an array of them that lives across
  // the layer loop is not promoted to registers (384 B of scratch per lane)
  u32x4 rg0[4], rg1[4], rg2[4], rg3[4], rg4[4], rg5[4];
  static_assert(W8_MAXCH == 6, "six register chunks");
#define RG(c) (c == 0 ? rg0 : c == 1 ? rg1 : c == 2 ? rg2 : c == 3 ? rg3 : c == 4 ? rg4 : rg5)
  // request the whole tile `nt` of layer l (zeros beyond the layer's K / tiles and for the FMA-path layers; clamped rows are
  // never stored).  Every condition is wave-uniform.
  auto fetch_layer = [&](int l, int nt) __attribute__((always_inline)) {
    const int K = K_of(l), N = N_of(l);
    const __bf16* __restrict__ W = W_of(l);
    const bool any = K >= 16 && nt * 32 < N;
    static_for<0, NCH>([&](auto C) __attribute__((always_inline)) {
      constexpr int c = decltype(C)::value;
      if (any && c * WCH < K) {
#pragma unroll
        for (int i = 0; i < 4; ++i) RG(c)[i] = *reinterpret_cast<const u32x4*>(W + (long)min(nt * 32 + i * 8 + srow, N - 1) * K + c * WCH + sk);
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) RG(c)[i] = u32x4{0u, 0u, 0u, 0u};
      }
    });
  };
  // first layer's weights: in flight under the activation tile load
  PHASE(0);
  fetch_layer(layer_of(0), wave);

  load_tile_bf16<512>(BWD ? a.dout : a.in, rowbase, r0, a.rows, BWD ? a.dims[a.nl] : a.dims[0], sa[0], tid);
  if (BWD && a.db_top && tid < a.dims[a.nl]) {   // top-layer bias gradient: column sums of this tile's dout rows (fp32 source, L2-hot)
    const int w = a.dims[a.nl];
    const int nr = min(RT, a.rows - r0);
    float s = 0.f;
    for (int r = 0; r < nr; ++r) s += a.dout[(rowbase + r) * w + tid];
    acc_add(a.db_top + (long)g * a.pstride + tid, s);
  }
  if (BWD && a.dw_top && a.nl >= 2) {   // narrow top layer: dW[c][k] = sum over this tile's rows of dout[r][c] * act[r][k]  (fp32, L2-hot)
    const int w = a.dims[a.nl], K1 = a.dims[a.nl - 1];
    if (tid < w * K1) {
      const int c = tid / K1, kk = tid - c * K1;
      const int nr = min(RT, a.rows - r0);
      const float* __restrict__ ap = a.act[a.nl - 2] + rowbase * K1 + kk;
      const float* __restrict__ dp = a.dout + rowbase * w + c;
      float s = 0.f;
      for (int r = 0; r < nr; ++r) s += dp[(long)r * w] * ap[(long)r * K1];
      acc_add(a.dw_top + (long)g * a.pstride + tid, s);
    }
  }
  __syncthreads();
  PHASE(1);
  int cur = 0;
#pragma unroll 1
  for (int step = 0; step < a.nl; ++step) {
    const int l = layer_of(step);
    if (!has_target(l)) break;
    const int K = K_of(l), N = N_of(l);
    float* __restrict__ dst = BWD ? (l > 0 ? a.dz[l] : a.din) : (l == a.nl - 1 ? a.out : a.act[l]);
    const __bf16* __restrict__ W = W_of(l);
    const bool last = BWD ? l == 0 : l == a.nl - 1;
    const int ntiles = (N + 31) / 32;
    const bool more = step + 1 < a.nl && has_target(layer_of(step + 1));
    const int ln = more ? layer_of(step + 1) : l;
    const float* __restrict__ mask = (BWD && l > 0 && !(a.dbg & 1)) ? a.act[l - 1] : nullptr;
    const float* __restrict__ bias = BWD ? nullptr : a.b[l] + (long)g * a.pstride;
    float* __restrict__ db = (BWD && l > 0 && a.db[l - 1] && !(a.dbg & 2)) ? a.db[l - 1] + (long)g * a.pstride : nullptr;
    const int npass = ntiles > 8 ? 2 : 1;      // a second tile per wave only where N = 384 (input gradient of the CMI classifiers)
#pragma unroll 1
    for (int pass = 0; pass < npass; ++pass) {
      const int nt = wave + 8 * pass;
      const bool mine = nt < ntiles;
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      // ReLU mask of the data-gradient chain (= forward activations of the layer below) and the forward bias: requested before
      // the products, unconditional clamped addresses
      // (addresses = wave-uniform row pointer + ONE 32-bit lane offset shared by the 16 accesses: as 16 per-lane 64-bit addresses
      //  they cost 32 registers per access group and the backward instantiation spilled)
      float mk[16];
      if (BWD && mask) {
        // accumulator r of a lane is row mu + 4 * lh, mu = (r & 3) + 8 * (r >> 2).  Row pointer = wave-uniform min(mu, rows_here - 1), the
        // upper half-wave adds 4 rows through the ONE lane offset: in a ragged tile it then reads up to 4 rows past the group's last row
        // (the next group's rows, or -- last group -- the slack the caller guarantees behind the buffer: MlpFusedArgs::act_slack); those
        // values are discarded (`ok` below).  Per-access clamps of the upper half cost registers this 256-VGPR kernel does not have
        // (three variants tried: 3 / 15 / 49 spilled VGPRs).  Round 3 fix: the previous clamp min(mu, rows_here - 5) + 4 * lh was right for
        // the upper half only -- in a ragged tile the LOWER half read row rows_here - 5 for every mu beyond it (rows 8..11 of a 12-row
        // tile all saw row 7's mask), i.e. wrong hidden-layer gradients for every batch that is not a multiple of 32
        // (tests/test_gpu_fused_oracle.py::test_mi_estimators_vs_rounded_oracle[tiny_odd-s1] found it).
        const unsigned loff = (unsigned)(4 * lh * N + min(nt * 32 + lr, N - 1));
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float* __restrict__ mrow = mask + (rowbase + min((r & 3) + 8 * (r >> 2), rows_here - 1)) * (long)N;
          mk[r] = mrow[loff];
        }
      }
      const float bn = bias ? bias[min(nt * 32 + lr, N - 1)] : 0.f;
      if (K < 16) {
        // degenerate reduction (the 2-logit top layer of the CMI classifier, backward): plain FMAs
        const int n = nt * 32 + lr;
        if (mine && n < N) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int m = (r & 3) + 8 * (r >> 2) + 4 * lh;
            float v = 0.f;
            for (int kk = 0; kk < K; ++kk) v += (float)sa[cur][m][kk] * (float)W[(long)n * K + kk];
            acc[r] = v;
          }
        }
      } else {
        const int nchunk = (K + WCH - 1) / WCH;
        // the tile is in registers: stage chunk c into the wave's slab (two buffers alternate), multiply
        static_for<0, NCH>([&](auto C) __attribute__((always_inline)) {
          constexpr int c = decltype(C)::value;
          if (c < nchunk && mine) {
#pragma unroll
            for (int i = 0; i < 4; ++i) *reinterpret_cast<u32x4*>(&wl[wave][c & 1][i * 8 + srow][sk]) = RG(c)[i];
#pragma unroll
            for (int u = 0; u < WCH / 16; ++u) {
              if (c * WCH + u * 16 < K) {
                const bf16x8 af = *reinterpret_cast<const bf16x8*>(&sa[cur][lr][c * WCH + u * 16 + 8 * lh]);
                const bf16x8 b0 = *reinterpret_cast<const bf16x8*>(&wl[wave][c & 1][lr][u * 16 + 8 * lh]);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, b0, acc, 0, 0, 0);
              }
            }
          }
        });
      }
      // the registers are free again: the next tile (this layer's second one, else the next layer's) is requested now and arrives
      // under the epilogue and the barrier
      PHASE(2 + 4 * step + 2 * pass);
      if (pass + 1 < npass) fetch_layer(l, wave + 8);
      else if (more) fetch_layer(ln, wave);
      // epilogue
      const int n = nt * 32 + lr;
      if (mine && n < N) {
        float csum = 0.f;
        const unsigned soff = (unsigned)(4 * lh * N + n);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int mu = (r & 3) + 8 * (r >> 2), m = mu + 4 * lh;
          const bool ok = m < rows_here;
          float v = acc[r] + bn;
          if (BWD) { if (mask) v = (ok && mk[r] > 0.f) ? v : 0.f; }
          else if (!last) v = fmaxf(v, 0.f);
          float* __restrict__ drow = dst + (rowbase + mu) * (long)N;
          if (ok && !(a.dbg & 4)) drow[soff] = v;
          if (!last) sa[cur ^ 1][m][n] = to_bf16(v);
          csum += v;
        }
        if (db) {
          csum += __shfl_xor(csum, 32, 64);
          if (lh == 0) acc_add(&db[n], csum);
        }
      }
      PHASE(3 + 4 * step + 2 * pass);
    }
    __syncthreads();
    cur ^= 1;
  }
  PHASE(20);
}

#undef RG

// ---------------------------------------------------------------------------------------------------------------
// Round 3: the 8-wave kernel re-cut for the shapes the step actually runs (4 layers, hidden width 256; D0 = 128 / 384 inputs,
// D4 = 128 / 2 / 1 outputs).  tools/hw/mlp_phase.hip showed ONE workgroup of mlp_img8_kernel taking 18 us forward / 33 us backward
// (the whole grid is no slower): per layer 2.5 us between barrier and last MFMA, 1.6-2.2 us of epilogue.  Its ISA says why: with
// run-time layer shapes every register chunk sits behind branches, so the wait in front of the FIRST chunk is vmcnt(3..0) -- all of
// the layer's weights AND the previous epilogue's store acknowledgements -- and every MFMA waits for its own two ds_read_b128.
// Here (a) the shapes are template parameters, the four layers are straight-line code and the compiler counts vmcnt exactly;
// (b) the weights come from FRAGMENT-ORDER images (bf16_frag_images below): the 16 bytes lane (n = lane & 31, half = lane >> 5)
// feeds to MFMA k-step u of output tile nt are contiguous per wave instruction (1 KiB), so a weight goes global -> VGPR -> MFMA
// with no LDS staging (the old kernel wrote and re-read 16 KB per wave and layer); (c) the next layer's fragments are requested
// as soon as the current layer's last MFMA has issued.  What bounds it now is the per-CU rate out of L2 (MI355X_MICROARCH.md:
// 66-73 GB/s per CU): every workgroup streams the whole stack, 393-459 KB, i.e. >= 5.6-6.5 us per pass.
// ---------------------------------------------------------------------------------------------------------------
constexpr int FR_HID = 256;

template <int NF>
__device__ __forceinline__ void load_frags(bf16x8 (&w)[NF], const __bf16* __restrict__ img, int nt, int lane) {
  const bf16x8* __restrict__ p = reinterpret_cast<const bf16x8*>(img) + (long)nt * NF * 64 + lane;
#pragma unroll
  for (int u = 0; u < NF; ++u) w[u] = p[u * 64];
}
// narrow matrix [N < 32][K] read from the PLAIN image: lane n -> row min(n, N-1) (columns >= N are never stored)
template <int NF>
__device__ __forceinline__ void load_frags_plain(bf16x8 (&w)[NF], const __bf16* __restrict__ img, int N, int lane) {
  const __bf16* __restrict__ p = img + (long)min(lane & 31, N - 1) * (NF * 16) + 8 * (lane >> 5);
#pragma unroll
  for (int u = 0; u < NF; ++u) w[u] = *reinterpret_cast<const bf16x8*>(p + u * 16);
}
template <int NF>
__device__ __forceinline__ void mma_tile(f32x16& acc, const __bf16 (*t)[LDA], const bf16x8 (&w)[NF], int lr, int lh) {
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
  for (int u = 0; u < NF; ++u) {
    const bf16x8 af = *reinterpret_cast<const bf16x8*>(&t[lr][u * 16 + 8 * lh]);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, w[u], acc, 0, 0, 0);
  }
}

// FULL: rows is a multiple of 32 (a kernel-level choice, not a branch: where a guarded and an unguarded epilogue merge, the
// compiler's wait counting assumes the stores may be absent and makes the next layer's first MFMA wait for their acknowledgements)
// DIN (backward): the gradient w.r.t. the stack's input is wanted (also a kernel-level choice, for the same reason)
template <bool BWD, int D0, int D4, bool FULL, bool DIN = false>
__global__ __launch_bounds__(512) void mlp_frag_kernel(MlpFusedArgs a) {
  static_assert(D0 % 64 == 0 && D0 <= MLPF_MAX_WIDTH && (D4 % 32 == 0 || D4 <= 2), "stack shape");
  constexpr bool NARROW = D4 < 32;
  __shared__ __attribute__((aligned(16))) __bf16 sa[2][RT][LDA];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tiles = (a.rows + RT - 1) / RT;
  const int xslot = blockIdx.x >> 3;
  const int g = (blockIdx.x & 7) + 8 * (xslot / tiles);      // a group (one weight set) is pinned to one XCD
  const int r0 = (xslot % tiles) * RT;
  if (g >= a.nb) return;
  const long rowbase = (long)g * a.brows + r0;
  const int lr = lane & 31, lh = lane >> 5;
  const int rows_here = min(RT, a.rows - r0);
  const long pg = (long)g * a.pstride;
  PHASE(0);

  if constexpr (!BWD) {
    // ---------------- forward: in -> act[0] -> act[1] -> act[2] -> out
    // biases first in the memory queue (a load behind the next layer's weight requests would make the epilogue wait for all of them)
    float bn[4];
#pragma unroll
    for (int l = 0; l < 4; ++l) bn[l] = a.b[l][pg + min(wave * 32 + lr, (l == 3 ? D4 : FR_HID) - 1)];
    auto epilogue = [&](const f32x16& acc, int nt, int N, float bias, float* __restrict__ dst, auto NX, auto RELU, __bf16 (*nx)[LDA])
                        __attribute__((always_inline)) {
      const int n = nt * 32 + lr;
      if (NARROW && !decltype(NX)::value && n >= N) return;    // only the narrow top layer has columns without an owner
      float* __restrict__ d0 = dst + rowbase * (long)N + n;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = (r & 3) + 8 * (r >> 2) + 4 * lh;
        float v = acc[r] + bias;
        if constexpr (decltype(RELU)::value) v = fmaxf(v, 0.f);
        if (FULL || m < rows_here) d0[m * N] = v;
        if constexpr (decltype(NX)::value) nx[m][n] = to_bf16(v);
      }
    };
    f32x16 acc;
    bf16x8 w0[D0 / 16];
    load_frags<D0 / 16>(w0, a.Wf[0] + pg, wave, lane);
    load_tile_bf16<512>(a.in, rowbase, r0, a.rows, D0, sa[0], tid);
    __syncthreads();
    PHASE(1);
    mma_tile<D0 / 16>(acc, sa[0], w0, lr, lh);
    PHASE(2);
    bf16x8 w1[FR_HID / 16];
    load_frags<FR_HID / 16>(w1, a.Wf[1] + pg, wave, lane);
    epilogue(acc, wave, FR_HID, bn[0], a.act[0], std::true_type{}, std::true_type{}, sa[1]);
    PHASE(3);
    __syncthreads();
    mma_tile<FR_HID / 16>(acc, sa[1], w1, lr, lh);
    PHASE(6);
    bf16x8 w2[FR_HID / 16];
    load_frags<FR_HID / 16>(w2, a.Wf[2] + pg, wave, lane);
    epilogue(acc, wave, FR_HID, bn[1], a.act[1], std::true_type{}, std::true_type{}, sa[0]);
    PHASE(7);
    __syncthreads();
    mma_tile<FR_HID / 16>(acc, sa[0], w2, lr, lh);
    PHASE(10);
    bf16x8 w3[FR_HID / 16];
    constexpr int TOPT = NARROW ? 1 : D4 / 32;       // waves with a tile of the top layer
    if (wave < TOPT) {
      if constexpr (NARROW) load_frags_plain<FR_HID / 16>(w3, a.Wb[3] + pg, D4, lane);
      else load_frags<FR_HID / 16>(w3, a.Wf[3] + pg, wave, lane);
    }
    epilogue(acc, wave, FR_HID, bn[2], a.act[2], std::true_type{}, std::true_type{}, sa[1]);
    PHASE(11);
    __syncthreads();
    if (wave < TOPT) {
      mma_tile<FR_HID / 16>(acc, sa[1], w3, lr, lh);
      PHASE(14);
      epilogue(acc, wave, D4, bn[3], a.out, std::false_type{}, std::false_type{}, sa[0]);
      PHASE(15);
    }
  } else {
    // ---------------- backward (data-gradient chain): dout -> dz[3] -> dz[2] -> dz[1] (-> din)
    // dz_l = (dz_{l+1} W_l) masked by act[l-1] > 0; column sums -> db[l-1]; top layer's bias / narrow weight gradients from dout
    if (!NARROW && a.db_top && tid < D4) {      // column sums of this tile's dout rows (fp32 source, L2-hot): all 32 loads in flight at once
      const float* __restrict__ dp = a.dout + rowbase * D4 + tid;
      float s = 0.f;
      if (FULL) {
#pragma unroll
        for (int r = 0; r < RT; ++r) s += dp[r * D4];
      } else {
#pragma unroll 8
        for (int r = 0; r < rows_here; ++r) s += dp[r * D4];
      }
      acc_add(a.db_top + pg + tid, s);
    }
    // mask values of one output tile: row min(m, rows_here - 1) (clamped rows are never stored)
    auto load_mask = [&](float (&mk)[16], const float* __restrict__ mask, int n) __attribute__((always_inline)) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = min((r & 3) + 8 * (r >> 2) + 4 * lh, rows_here - 1);
        mk[r] = mask[(rowbase + m) * (long)FR_HID + n];
      }
    };
    auto epilogue = [&](const f32x16& acc, const float (&mk)[16], auto MASKED, int n, int N, float* __restrict__ dst, auto NX, __bf16 (*nx)[LDA],
                        float* __restrict__ db) __attribute__((always_inline)) {
      float csum = 0.f;
      float* __restrict__ d0 = dst + rowbase * (long)N + n;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = (r & 3) + 8 * (r >> 2) + 4 * lh;
        const bool ok = FULL || m < rows_here;
        float v = acc[r];
        if constexpr (decltype(MASKED)::value) v = (ok && mk[r] > 0.f) ? v : 0.f;
        if (ok) d0[m * N] = v;
        if constexpr (decltype(NX)::value) nx[m][n] = to_bf16(v);
        csum += v;
      }
      if (db) {
        csum += __shfl_xor(csum, 32, 64);
        if (lh == 0) acc_add(&db[n], csum);
      }
    };
    f32x16 acc;
    float mk[16];
    const int n = wave * 32 + lr;
    bf16x8 w2[FR_HID / 16];
    // top layer: reduction over the D4 outputs
    if constexpr (NARROW) {
      // 1- or 2-wide top layer: plain FMAs.  The fp32 dout tile goes through LDS once (rows beyond the tile's last one: zeros); the top
      // layer's bias gradient is its column sums and its WEIGHT gradient dW[c][k] = sum_r dout[r][c] * act2[r][k] is formed from the ReLU-mask
      // values this lane holds anyway (mask of dz[3] = act[2]: rows m of column n), one cross-half shuffle and one atomic per (c, n)
      __shared__ float sd[RT][2];
      load_frags<FR_HID / 16>(w2, a.WfT[2] + pg, wave, lane);
      load_mask(mk, a.act[2], n);
      float wv[D4];
#pragma unroll
      for (int c = 0; c < D4; ++c) wv[c] = (float)a.Wb[3][pg + (long)c * FR_HID + n];
      if (tid < RT * D4) {
        const int r = tid / D4, c = tid - r * D4;
        sd[r][c] = r < rows_here ? a.dout[(rowbase + r) * D4 + c] : 0.f;
      }
      __syncthreads();
      if (a.db_top && tid < D4) {
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < RT; ++r) s += sd[r][tid];
        acc_add(a.db_top + pg + tid, s);
      }
      float dw[D4];
#pragma unroll
      for (int c = 0; c < D4; ++c) dw[c] = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = (r & 3) + 8 * (r >> 2) + 4 * lh;
        float v = 0.f;
#pragma unroll
        for (int c = 0; c < D4; ++c) {
          const float d = sd[m][c];
          v += (float)to_bf16(d) * wv[c];
          dw[c] += d * mk[r];
        }
        acc[r] = v;
      }
      if (a.dw_top) {
#pragma unroll
        for (int c = 0; c < D4; ++c) {
          const float t = dw[c] + __shfl_xor(dw[c], 32, 64);
          if (lh == 0) acc_add(a.dw_top + pg + (long)c * FR_HID + n, t);
        }
      }
      PHASE(1);
      PHASE(2);
    } else {
      bf16x8 w3[D4 / 16];
      load_frags<D4 / 16>(w3, a.WfT[3] + pg, wave, lane);
      load_mask(mk, a.act[2], n);
      load_tile_bf16<512>(a.dout, rowbase, r0, a.rows, D4, sa[0], tid);
      __syncthreads();
      PHASE(1);
      mma_tile<D4 / 16>(acc, sa[0], w3, lr, lh);
      PHASE(2);
      load_frags<FR_HID / 16>(w2, a.WfT[2] + pg, wave, lane);
    }
    epilogue(acc, mk, std::true_type{}, n, FR_HID, a.dz[3], std::true_type{}, sa[1], a.db[2] ? a.db[2] + pg : nullptr);
    PHASE(3);
    __syncthreads();
    load_mask(mk, a.act[1], n);
    mma_tile<FR_HID / 16>(acc, sa[1], w2, lr, lh);
    PHASE(6);
    bf16x8 w1[FR_HID / 16];
    load_frags<FR_HID / 16>(w1, a.WfT[1] + pg, wave, lane);
    epilogue(acc, mk, std::true_type{}, n, FR_HID, a.dz[2], std::true_type{}, sa[0], a.db[1] ? a.db[1] + pg : nullptr);
    PHASE(7);
    __syncthreads();
    load_mask(mk, a.act[0], n);
    mma_tile<FR_HID / 16>(acc, sa[0], w1, lr, lh);
    PHASE(10);
    constexpr int T0 = D0 / 32;                       // input-gradient tiles: 4 (waves 0..3) or 12 (waves 0..3 take a second one)
    bf16x8 w0[FR_HID / 16];
    // (every wave loads, waves without a tile a copy of the last one: a conditional request would cost the exact wait counts)
    if constexpr (DIN) load_frags<FR_HID / 16>(w0, a.WfT[0] + pg, min(wave, T0 - 1), lane);
    epilogue(acc, mk, std::true_type{}, n, FR_HID, a.dz[1], std::true_type{}, sa[1], a.db[0] ? a.db[0] + pg : nullptr);
    PHASE(11);
    if constexpr (DIN) {
      __syncthreads();
      if (wave < T0) {
        mma_tile<FR_HID / 16>(acc, sa[1], w0, lr, lh);
        PHASE(14);
        if (T0 > 8 && wave + 8 < T0) load_frags<FR_HID / 16>(w0, a.WfT[0] + pg, wave + 8, lane);
        epilogue(acc, mk, std::false_type{}, n, D0, a.din, std::false_type{}, sa[0], nullptr);
        PHASE(15);
        if (T0 > 8 && wave + 8 < T0) {
          mma_tile<FR_HID / 16>(acc, sa[1], w0, lr, lh);
          epilogue(acc, mk, std::false_type{}, n + 256, D0, a.din, std::false_type{}, sa[0], nullptr);
        }
      }
    }
  }
  PHASE(20);
}

// Fragment-order image of a table of strided groups of matrices.  Entry e: a matrix with OUT output columns and RED reduction
// length whose element (c, r) is src[c * RED + r] (tr = 0: the forward weight [N = OUT, K = RED]) or src[r * OUT + c] (tr = 1: the same
// memory seen by the data-gradient product, OUT = K, RED = N).  dst holds, at the matrix's own offset and footprint,
//   dst[((c / 32 * (RED / 16) + r / 16) * 64 + (c % 32) + 32 * ((r % 16) / 8)) * 8 + r % 8] = bf16(element (c, r)).
// One workgroup = one 32-column tile x 64 reduction values: 4 KB written contiguously.
__global__ __launch_bounds__(256) void frag_images_kernel(const float* __restrict__ src, __bf16* __restrict__ dst, FragTable t) {
  int b = blockIdx.x, e = 0;
  while (e + 1 < t.n && b >= t.blk0[e + 1]) ++e;
  b -= t.blk0[e];
  const int OUT = t.OUT[e], RED = t.RED[e];
  const int q64 = RED / 64, per = (OUT / 32) * q64;
  const int g = b / per, rem = b - g * per;
  const int nt = rem / q64, u = (rem - nt * q64) * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63, lr = lane & 31, lh = lane >> 5;
  const long base = t.off[e] + (long)g * t.gstride[e];
  const int c = nt * 32 + lr, rr = u * 16 + 8 * lh;
  bf16x8 p;
  if (!t.tr[e]) {
    const float4 x = *reinterpret_cast<const float4*>(src + base + (long)c * RED + rr);
    const float4 y = *reinterpret_cast<const float4*>(src + base + (long)c * RED + rr + 4);
    p = pack8(x, y);
  } else {
#pragma unroll
    for (int i = 0; i < 8; ++i) p[i] = to_bf16(src[base + (long)(rr + i) * OUT + c]);
  }
  *reinterpret_cast<bf16x8*>(dst + t.dshift[e] + base + ((long)(nt * (RED / 16) + u) * 64 + lane) * 8) = p;
}

__global__ void bf16_image_kernel(const float* __restrict__ src, __bf16* __restrict__ dst, long n4) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const float4 v = reinterpret_cast<const float4*>(src)[i];
    bf16x4 p; p[0] = to_bf16(v.x); p[1] = to_bf16(v.y); p[2] = to_bf16(v.z); p[3] = to_bf16(v.w);
    reinterpret_cast<bf16x4*>(dst)[i] = p;
  }
}

__global__ void transpose_images_kernel(const float* __restrict__ src, __bf16* __restrict__ dstT, TransposeTable t) {
  __shared__ float tile[32][33];
  int z = blockIdx.z, e = 0;
  while (e < t.n - 1 && z >= t.nb[e]) { z -= t.nb[e]; ++e; }
  const int N = t.N[e], K = t.K[e];
  const int k0 = blockIdx.x * 32, n0 = blockIdx.y * 32;
  if (k0 >= K || n0 >= N) return;
  const long base = t.off[e] + (long)z * t.gstride[e];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int j = ty; j < 32; j += 8) tile[j][tx] = (n0 + j < N && k0 + tx < K) ? src[base + (long)(n0 + j) * K + k0 + tx] : 0.f;
  __syncthreads();
  for (int j = ty; j < 32; j += 8)
    if (k0 + j < K && n0 + tx < N) dstT[base + (long)(k0 + j) * N + n0 + tx] = to_bf16(tile[tx][j]);
}

__global__ __launch_bounds__(256) void mlp_bwd_kernel(MlpFusedArgs a) {
  __shared__ __attribute__((aligned(16))) __bf16 sa[2][RT][LDA];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // workgroups are dealt round-robin to the 8 XCDs: give every group (one weight set) to ONE XCD so that its row tiles
  // share the weight lines in that XCD's L2 (grid.x = 8 * tiles * ceil(nb / 8), surplus ids exit)
  const int tiles = (a.rows + RT - 1) / RT;
  const int xslot = blockIdx.x >> 3;
  const int g = (blockIdx.x & 7) + 8 * (xslot / tiles), r0 = (xslot % tiles) * RT;
  if (g >= a.nb) return;
  const long rowbase = (long)g * a.brows + r0;
  const int lr = lane & 31, lh = lane >> 5;
  load_tile_bf16(a.dout, rowbase, r0, a.rows, a.dims[a.nl], sa[0], tid);
  __syncthreads();
  int cur = 0;
  for (int l = a.nl - 1; l >= 0; --l) {
    float* __restrict__ target = l > 0 ? a.dz[l] : a.din;
    if (!target) break;
    const int NR = a.dims[l + 1], KO = a.dims[l];          // reduction width, output width
    const int nsteps = (NR + 15) / 16;
    const float* __restrict__ W = a.W[l] + (long)g * a.pstride;
    const float* __restrict__ mask = l > 0 ? a.act[l - 1] : nullptr;
    float* __restrict__ db = (l > 0 && a.db[l - 1]) ? a.db[l - 1] + (long)g * a.pstride : nullptr;
    const int ktiles = KO / 32;                             // KO is a multiple of 32
    constexpr int TW = 3;                                   // output tiles per wave (KO <= 384)
    f32x16 acc[TW];
#pragma unroll
    for (int t = 0; t < TW; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    constexpr int NS = 4;                                   // reduction steps (of 16) per register chunk
    const int nchunk = (nsteps + NS - 1) / NS;
    float wb[2][NS][TW][8];
    auto fetch = [&](int c, auto S) {
      constexpr int s = decltype(S)::value;
#pragma unroll
      for (int q = 0; q < NS; ++q) {
        const int ns = c * NS + q;
        if (ns >= nsteps) break;
#pragma unroll
        for (int t = 0; t < TW; ++t) {
          const int kt = wave + 4 * t;
          if (kt < ktiles) {
            const int kc = kt * 32 + lr;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              const int n = min(ns * 16 + 8 * lh + j, NR - 1);   // rows beyond NR meet zero A-columns
              wb[s][q][t][j] = W[(long)n * KO + kc];
            }
          }
        }
      }
    };
    auto mult = [&](int c, auto S) {
      constexpr int s = decltype(S)::value;
#pragma unroll
      for (int q = 0; q < NS; ++q) {
        const int ns = c * NS + q;
        if (ns >= nsteps) break;
        const bf16x8 af = *reinterpret_cast<const bf16x8*>(&sa[cur][lr][ns * 16 + 8 * lh]);
#pragma unroll
        for (int t = 0; t < TW; ++t) {
          if (wave + 4 * t < ktiles) {
            bf16x8 bfr;
#pragma unroll
            for (int j = 0; j < 8; ++j) bfr[j] = to_bf16(wb[s][q][t][j]);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bfr, acc[t], 0, 0, 0);
          }
        }
      }
    };
    fetch(0, I0{});
    for (int c = 0; c < nchunk; c += 2) {
      if (c + 1 < nchunk) fetch(c + 1, I1{});
      mult(c, I0{});
      if (c + 2 < nchunk) fetch(c + 2, I0{});
      if (c + 1 < nchunk) mult(c + 1, I1{});
    }
#pragma unroll
    for (int t = 0; t < TW; ++t) {
      const int kt = wave + 4 * t;
      if (kt >= ktiles) continue;
      const int kc = kt * 32 + lr;
      float csum = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = (r & 3) + 8 * (r >> 2) + 4 * lh;
        const bool ok = r0 + m < a.rows;
        float v = acc[t][r];
        if (mask) v = (ok && mask[(rowbase + m) * KO + kc] > 0.f) ? v : 0.f;   // post-activation > 0 <=> pre-activation > 0
        if (ok) target[(rowbase + m) * KO + kc] = v;
        if (l > 0) sa[cur ^ 1][m][kc] = to_bf16(v);
        csum += v;
      }
      if (db) {
        csum += __shfl_xor(csum, 32, 64);
        if (lh == 0) acc_add(&db[kc], csum);
      }
    }
    __syncthreads();
    cur ^= 1;
  }
}

}  // namespace

bool mlp_fused_supported(int nb, int rows, int nl, const int* dims) {
  if (nl < 1 || nl > MLPF_MAX_LAYERS || nb < 1 || rows < 1) return false;
  for (int l = 0; l < nl; ++l)
    if (dims[l] % 64 != 0 || dims[l] > MLPF_MAX_WIDTH) return false;   // every layer input: k-chunks of 64, <= 12 output tiles
  return dims[nl] >= 1 && dims[nl] <= 256;
}

// (round 1's direct-from-L2 instantiation for stacks with thousands of row tiles -- mlp_img_kernel<*, true> -- lost to the GEMM chain in round 2
// and is no longer instantiated: round 6)
// few workgroups (latency-bound stack): the 8-wave kernel with the register-resident, prefetched weight tile.  Every reduction
// width must be a whole number of 8-element pieces (16-byte loads) or < 16 (the FMA path).
static int mlp_small8(const MlpFusedArgs& a, bool bwd) {   // -> 0 (use the 4-wave kernel), or the register chunks needed: 4 / 6
  constexpr int waves = 0;   // (an environment knob until round 5: fixed at its measured optimum): 4 = never, 8 = both directions
  if (waves == 4) return 0;
  const long wgs = (long)((a.rows + RT - 1) / RT) * a.nb;
  if (wgs > 512) return 0;
  int kmax = 0;
  for (int l = 0; l < a.nl; ++l) {
    const int K = bwd ? a.dims[l + 1] : a.dims[l];
    if (K >= 16 && K % 8 != 0) return 0;
    kmax = K > kmax ? K : kmax;
  }
  // backward, ragged last row tile: the 8-wave kernel's mask loads touch up to 4 rows behind a group's last row (see the kernel);
  // without the caller's guarantee that this is mapped memory the 4-wave kernel (per-lane clamps) takes the stack
  if (bwd && a.rows % RT != 0 && !a.act_slack) return 0;
  if (bwd) return kmax <= 256 ? 4 : 0;      // (the 6-chunk backward instantiation spills)
  return kmax <= 256 ? 4 : 6;
}

// the fragment-image kernel takes: 4 layers, hidden 256, inputs 128 / 384, outputs 128 / 2 / 1, few workgroups; -> 0 or an id
static int mlp_frag_shape(const MlpFusedArgs& a, bool bwd) {
  const bool off = knob("MIMRL_MLP_NO_FRAG") != nullptr;    // tuning knob: the round-2 kernels (read per call: tests/test_gpu_fused_oracle.py toggles it)
  if (off || a.nl != 4 || !(bwd ? a.WfT[1] : a.Wf[0]) || !a.Wb[3]) return 0;
  if (a.dims[1] != FR_HID || a.dims[2] != FR_HID || a.dims[3] != FR_HID) return 0;
  const long wgs = (long)((a.rows + RT - 1) / RT) * a.nb;
  if (wgs > 512) return 0;
  const int d0 = a.dims[0], d4 = a.dims[4];
  if (d0 == 128 && d4 == 128) return 1;
  if (d0 == 384 && d4 == 2) return 2;
  if (d0 == 128 && d4 == 1) return 3;
  return 0;
}

static int check(const MlpFusedArgs& a) {
  if (!mlp_fused_supported(a.nb, a.rows, a.nl, a.dims)) return set_error(MIMRL_ERR_ARG, "mlp_fused: unsupported stack shape");
  if (a.pstride % 4 != 0) return set_error(MIMRL_ERR_ARG, "mlp_fused: group stride must be a multiple of 4 floats");
  return MIMRL_OK;
}

int mlp_stack_fwd_fused(hipStream_t s, const MlpFusedArgs& a) {
  MX(check(a));
  if (const int id = mlp_frag_shape(a, false)) {
    const dim3 grid(8 * ((a.rows + RT - 1) / RT) * ((a.nb + 7) / 8));
    const bool full = a.rows % RT == 0;
#define FRAGK(D0_, D4_) do { if (full) hipLaunchKernelGGL((mlp_frag_kernel<false, D0_, D4_, true>), grid, dim3(512), 0, s, a); \
                             else hipLaunchKernelGGL((mlp_frag_kernel<false, D0_, D4_, false>), grid, dim3(512), 0, s, a); } while (0)
    if (id == 1) FRAGK(128, 128); else if (id == 2) FRAGK(384, 2); else FRAGK(128, 1);
#undef FRAGK
    LAUNCH_CHECK();
    return MIMRL_OK;
  }
  if (a.Wb[0]) {
    if (mlp_small8(a, false) == 4) hipLaunchKernelGGL((mlp_img8_kernel<false, 4>), dim3(8 * ((a.rows + RT - 1) / RT) * ((a.nb + 7) / 8)), dim3(512), 0, s, a);
    else if (mlp_small8(a, false) == 6) hipLaunchKernelGGL((mlp_img8_kernel<false, 6>), dim3(8 * ((a.rows + RT - 1) / RT) * ((a.nb + 7) / 8)), dim3(512), 0, s, a);
    else hipLaunchKernelGGL((mlp_img_kernel<false, false>), dim3(8 * ((a.rows + RT - 1) / RT) * ((a.nb + 7) / 8)), dim3(256), 0, s, a);
    LAUNCH_CHECK();
    return MIMRL_OK;
  }
  hipLaunchKernelGGL(mlp_fwd_kernel, dim3(8 * ((a.rows + RT - 1) / RT) * ((a.nb + 7) / 8)), dim3(256), 0, s, a);
  LAUNCH_CHECK();
  return MIMRL_OK;
}

bool mlp_bwd_takes_top_wgrad(const MlpFusedArgs& a) {
  if (mlp_frag_shape(a, true)) return a.dims[4] <= 2;
  return a.WbT[0] && mlp_small8(a, true) == 4 && a.nl >= 2 && a.dims[a.nl] * a.dims[a.nl - 1] <= 512;
}

int mlp_stack_bwd_fused(hipStream_t s, const MlpFusedArgs& a_) {
  MlpFusedArgs a = a_;
  a.dbg = dbg_env("MIMRL_DBG_MLPB") ? atoi(dbg_env("MIMRL_DBG_MLPB")) : 0;
  MX(check(a));
  if (a.dw_top && !mlp_bwd_takes_top_wgrad(a)) return set_error(MIMRL_ERR_ARG, "mlp_stack_bwd_fused: dw_top not supported for this stack");
  if (const int id = mlp_frag_shape(a, true)) {
    const dim3 grid(8 * ((a.rows + RT - 1) / RT) * ((a.nb + 7) / 8));
    const bool full = a.rows % RT == 0;
#define FRAGK(D0_, D4_) do { if (full && a.din) hipLaunchKernelGGL((mlp_frag_kernel<true, D0_, D4_, true, true>), grid, dim3(512), 0, s, a); \
                             else if (full) hipLaunchKernelGGL((mlp_frag_kernel<true, D0_, D4_, true, false>), grid, dim3(512), 0, s, a); \
                             else if (a.din) hipLaunchKernelGGL((mlp_frag_kernel<true, D0_, D4_, false, true>), grid, dim3(512), 0, s, a); \
                             else hipLaunchKernelGGL((mlp_frag_kernel<true, D0_, D4_, false, false>), grid, dim3(512), 0, s, a); } while (0)
    if (id == 1) FRAGK(128, 128); else if (id == 2) FRAGK(384, 2); else FRAGK(128, 1);
#undef FRAGK
    LAUNCH_CHECK();
    return MIMRL_OK;
  }
  if (a.WbT[0]) {
    if (mlp_small8(a, true) == 4) hipLaunchKernelGGL((mlp_img8_kernel<true, 4>), dim3(8 * ((a.rows + RT - 1) / RT) * ((a.nb + 7) / 8)), dim3(512), 0, s, a);
    else hipLaunchKernelGGL((mlp_img_kernel<true, false>), dim3(8 * ((a.rows + RT - 1) / RT) * ((a.nb + 7) / 8)), dim3(256), 0, s, a);
    LAUNCH_CHECK();
    return MIMRL_OK;
  }
  if (a.db_top) return set_error(MIMRL_ERR_ARG, "mlp_stack_bwd_fused: db_top needs the transposed bf16 weight images");
  hipLaunchKernelGGL(mlp_bwd_kernel, dim3(8 * ((a.rows + RT - 1) / RT) * ((a.nb + 7) / 8)), dim3(256), 0, s, a);
  LAUNCH_CHECK();
  return MIMRL_OK;
}

int bf16_image(hipStream_t s, const float* src, __bf16* dst, long n) {
  if (n % 4 != 0) return set_error(MIMRL_ERR_ARG, "bf16_image: length must be a multiple of 4");
  hipLaunchKernelGGL(bf16_image_kernel, dim3(1024), dim3(256), 0, s, src, dst, n / 4);
  LAUNCH_CHECK();
  return MIMRL_OK;
}

int bf16_transposed_images(hipStream_t s, const float* src, __bf16* dstT, const TransposeTable& t) {
  if (t.n < 1 || t.n > 12) return set_error(MIMRL_ERR_ARG, "bf16_transposed_images: 1..12 table entries");
  int z = 0, kmax = 0, nmax = 0;
  for (int e = 0; e < t.n; ++e) { z += t.nb[e]; kmax = t.K[e] > kmax ? t.K[e] : kmax; nmax = t.N[e] > nmax ? t.N[e] : nmax; }
  hipLaunchKernelGGL(transpose_images_kernel, dim3((kmax + 31) / 32, (nmax + 31) / 32, z), dim3(256), 0, s, src, dstT, t);
  LAUNCH_CHECK();
  return MIMRL_OK;
}

int bf16_frag_images(hipStream_t s, const float* src, __bf16* dst, const FragTable& t_) {
  FragTable t = t_;
  if (t.n < 1 || t.n > 24) return set_error(MIMRL_ERR_ARG, "bf16_frag_images: 1..24 table entries");
  int blocks = 0;
  for (int e = 0; e < t.n; ++e) {
    if (t.OUT[e] % 32 != 0 || t.RED[e] % 64 != 0 || t.off[e] % 8 != 0 || t.gstride[e] % 8 != 0 || t.dshift[e] % 8 != 0)
      return set_error(MIMRL_ERR_ARG, "bf16_frag_images: entry %d is not [32k x 64k] at a 16-byte image offset", e);
    t.blk0[e] = blocks;
    blocks += t.nb[e] * (t.OUT[e] / 32) * (t.RED[e] / 64);
  }
  t.blk0[t.n] = blocks;
  hipLaunchKernelGGL(frag_images_kernel, dim3(blocks), dim3(256), 0, s, src, dst, t);
  LAUNCH_CHECK();
  return MIMRL_OK;
}

}  // namespace mimrl
