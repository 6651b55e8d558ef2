// Fused concat-critic forward (see concat_fused.h).
#include "concat_fused.h"

namespace mimrl {

namespace {

constexpr int CR = 128;            // pair rows per workgroup
constexpr int CH = 256;            // hidden width (VMI.py:13-22 with hidden_dim 256)
constexpr int AP = CH + 8;         // bf16 pitch of the activation tile (528 B rows)
constexpr int WKC = 32;            // k-values per staged weight chunk
constexpr int WP = WKC + 8;        // bf16 pitch of a staged weight row (80 B: conflict-free 16-byte fragment reads)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// 8 waves: wave = (wm, wn): 32-row tile wm of the 128 rows, 128-column half wn (four 32-column MFMA tiles)
template <int SAVE>   // what the backward pass gets: see ConcatFwdArgs::save (compile time: no branch inside the unrolled epilogues)
__global__ __launch_bounds__(512) void concat_fwd_kernel(ConcatFwdArgs a) {
  __shared__ __attribute__((aligned(16))) __bf16 act[CR][AP];
  __shared__ __attribute__((aligned(16))) __bf16 wb[2][CH][WP];
  __shared__ LdsAcc sc[CR];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 31, lh = lane >> 5;
  const int wm = wave & 3, wn = wave >> 2;
  const int e = blockIdx.y, B = a.B;
  const long row0 = (long)blockIdx.x * CR, ebase = (long)e * B * B;
  const float* __restrict__ P = a.P + (long)e * B * CH;
  const float* __restrict__ Q = a.Q + (long)e * B * CH;
  // ---- layer 0 in its separable form: act[p][u] = relu(P[i][u] + Q[j][u]),  p = i*B + j   (16 float4 pairs per thread, 4 in flight)
  {
    float* __restrict__ o0 = a.a0 + (ebase + row0) * CH;
    __bf16* __restrict__ o0b = a.a0b + (ebase + row0) * CH;
    // B a multiple of 128 (every benchmark shape): the tile is ONE x row, so this thread's P quad is the same for all 16 of its pair rows --
    // loaded once, and the Q quads in two batches of 8 (round 5: 4 batches of 4 + 4 = four dependent round trips in front of the products)
    auto gen = [&](auto ONE_I) __attribute__((always_inline)) {
      constexpr bool one_i = decltype(ONE_I)::value;
      constexpr int NQ = one_i ? 8 : 4;
      float4 xone = make_float4(0.f, 0.f, 0.f, 0.f);
      if (one_i) xone = *reinterpret_cast<const float4*>(P + (row0 / B) * CH + (tid & 63) * 4);
#pragma unroll 1
      for (int base = 0; base < CR * 64; base += 512 * NQ) {
        float4 x[NQ], y[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          const int idx = base + tid + 512 * q, row = idx >> 6, c4 = (idx & 63) * 4;
          const long p = row0 + row;
          const int i = (int)(p / B), j = (int)(p - (long)i * B);
          x[q] = one_i ? xone : *reinterpret_cast<const float4*>(P + (long)i * CH + c4);
          y[q] = *reinterpret_cast<const float4*>(Q + (long)j * CH + c4);
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          const int idx = base + tid + 512 * q, row = idx >> 6, c4 = (idx & 63) * 4;
          float4 v;
          v.x = fmaxf(x[q].x + y[q].x, 0.f); v.y = fmaxf(x[q].y + y[q].y, 0.f); v.z = fmaxf(x[q].z + y[q].z, 0.f); v.w = fmaxf(x[q].w + y[q].w, 0.f);
          bf16x4 b; b[0] = to_bf16(v.x); b[1] = to_bf16(v.y); b[2] = to_bf16(v.z); b[3] = to_bf16(v.w);
          if (SAVE == 1) *reinterpret_cast<float4*>(o0 + (long)row * CH + c4) = v;
          else if (SAVE == 2) *reinterpret_cast<bf16x4*>(o0b + (long)row * CH + c4) = b;
          if (SAVE >= 2) {   // sign word of (row, 32 columns) = the nibbles of 8 neighbouring lanes (a wave holds one row: lane = column quad)
            uint32_t nib = (v.x > 0.f ? 1u : 0u) | (v.y > 0.f ? 2u : 0u) | (v.z > 0.f ? 4u : 0u) | (v.w > 0.f ? 8u : 0u);
            nib |= (uint32_t)__shfl_down((int)nib, 1) << 4;
            nib |= (uint32_t)__shfl_down((int)nib, 2) << 8;
            nib |= (uint32_t)__shfl_down((int)nib, 4) << 16;
            if ((lane & 7) == 0) a.m0[(ebase + row0 + row) * 8 + (lane >> 3)] = nib;
          }
          *reinterpret_cast<bf16x4*>(&act[row][c4]) = b;
        }
      }
    };
    if (B % CR == 0) gen(std::true_type{}); else gen(std::false_type{});
  }
  if (tid < CR) sc[tid].zero();
  float hp[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) hp[r] = 0.f;
  // ---- the two hidden layers, activation tile resident in LDS
#pragma unroll 1
  for (int layer = 1; layer <= 2; ++layer) {
    const __bf16* __restrict__ W = (layer == 1 ? a.W1 : a.W2) + (long)e * a.pstride;
    const float* __restrict__ bias = (layer == 1 ? a.b1 : a.b2) + (long)e * a.pstride;
    float* __restrict__ dst = (layer == 1 ? a.a1 : a.a2) + (ebase + row0) * CH;
    __bf16* __restrict__ dstb = a.a1b + (ebase + row0) * CH;                       // (layer 1 only)
    uint32_t* __restrict__ dstm = (layer == 1 ? a.m1 : a.m2) + (ebase + row0) * 8;
    float bn[4], w3c[4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
      const int col = wn * 128 + ct * 32 + lr;
      bn[ct] = bias[col];
      w3c[ct] = layer == 2 ? a.w3[(long)e * a.pstride + col] : 0.f;
    }
    f32x16 acc[4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ct][r] = 0.f;
    // weight chunk kc = W[:, 32 kc .. 32 kc + 32): thread -> (row n = tid >> 1, 16-element half), two 16-byte pieces
    const int wn_ = tid >> 1, wh = (tid & 1) * 16;
    const __bf16* __restrict__ wsrc = W + (long)wn_ * CH + wh;
    u32x4 w0 = *reinterpret_cast<const u32x4*>(wsrc), w1 = *reinterpret_cast<const u32x4*>(wsrc + 8);
    *reinterpret_cast<u32x4*>(&wb[0][wn_][wh]) = w0;
    *reinterpret_cast<u32x4*>(&wb[0][wn_][wh + 8]) = w1;
    __syncthreads();          // (also: the activation tile of this layer is complete)
    int cur = 0;
#pragma unroll 1
    for (int kc = 0; kc < CH / WKC; ++kc) {
      const bool more = kc + 1 < CH / WKC;
      if (more) {
        w0 = *reinterpret_cast<const u32x4*>(wsrc + (kc + 1) * WKC);
        w1 = *reinterpret_cast<const u32x4*>(wsrc + (kc + 1) * WKC + 8);
      }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const bf16x8 af = *reinterpret_cast<const bf16x8*>(&act[wm * 32 + lr][kc * WKC + ks * 16 + 8 * lh]);
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
          const bf16x8 bfr = *reinterpret_cast<const bf16x8*>(&wb[cur][wn * 128 + ct * 32 + lr][ks * 16 + 8 * lh]);
          acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bfr, acc[ct], 0, 0, 0);
        }
      }
      if (more) {
        *reinterpret_cast<u32x4*>(&wb[cur ^ 1][wn_][wh]) = w0;
        *reinterpret_cast<u32x4*>(&wb[cur ^ 1][wn_][wh + 8]) = w1;
      }
      __syncthreads();
      cur ^= 1;
    }
    // every wave is done reading the activation tile: bias + ReLU, fp32 copy to HBM, bf16 back into the tile (in place)
    const unsigned soff = (unsigned)(4 * lh * CH + wn * 128 + lr);
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int mu = wm * 32 + (r & 3) + 8 * (r >> 2), m = mu + 4 * lh, col = wn * 128 + ct * 32 + lr;
        const float v = fmaxf(acc[ct][r] + bn[ct], 0.f);
        if (SAVE == 1 || (SAVE == 2 && layer == 2)) { float* __restrict__ drow = dst + (long)mu * CH + ct * 32; drow[soff] = v; }
        if (SAVE >= 2) {   // ReLU sign of (row, these 32 columns): the ballot's low word is row mu (lanes 0..31), its high word row mu + 4.
          // Every lane of a half stores the same word to the same address (no divergent branch in the unrolled loop).  Round 5 tried collecting
          // the words in the lane that owns their row (two v_cndmask per word) for ONE 16-byte store per row: 20 us SLOWER per launch
          // (300 / 210 vs 280 / 194 us, same box) -- the selects are a dependent chain on the epilogue, the 64 small stores are not
          const unsigned long long bal = __ballot(v > 0.f);
          dstm[(long)m * 8 + wn * 4 + ct] = lh ? (uint32_t)(bal >> 32) : (uint32_t)bal;
        }
        act[m][col] = to_bf16(v);
        hp[r] += v * w3c[ct];
      }
    __syncthreads();
    if (SAVE == 2 && layer == 1) {   // the finished bf16 tile goes out with coalesced 16-byte stores (8 per thread)
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int idx = tid + 512 * q, row = idx >> 5, c8 = (idx & 31) * 8;
        *reinterpret_cast<u32x4*>(dstb + (long)row * CH + c8) = *reinterpret_cast<const u32x4*>(&act[row][c8]);
      }
    }
  }
  // ---- score head (256 -> 1): the per-lane partial dot products are summed over this wave's 32-lane halves (= its 128 columns), then
  // over the two column halves through LDS
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const float t = half_sum_hi(hp[r]);                       // valid in lanes 16..31 / 48..63
    if (lr == 16) sc[wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh].add(t);
  }
  __syncthreads();
  if (tid < CR) a.scores[ebase + row0 + tid] = sc[tid].get() + a.b3[(long)e * a.pstride];
}

#ifdef MIMRL_PHASE_PROBE
// probe build: wall-clock ticks (100 MHz) per phase of concat_bwd_kernel, summed over the tiles of workgroup 0's run, [8 * WG + phase]:
// 0 dZ2 generation, 1 product W2, 2 epilogue 1 (+ dZ1 out), 3 product W1, 4 epilogue 0 (+ dP), 5 dQ flush, 6 tiles (tools/concat_phase.py)
__device__ long long g_cc_phase[16];
#define CPH_DECL long long cph_t = (long long)wall_clock64()
#define CPH(i) do { if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) { const long long n_ = (long long)wall_clock64(); g_cc_phase[(WG ? 8 : 0) + (i)] += n_ - cph_t; cph_t = n_; } } while (0)
#define CPH_COUNT() do { if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) g_cc_phase[(WG ? 8 : 0) + 6] += 1; } while (0)
#else
#define CPH_DECL do { } while (0)
#define CPH(i) do { } while (0)
#define CPH_COUNT() do { } while (0)
#endif

// ---------------------------------------------------------------------------------------------------------------
// backward data-gradient chain (see concat_fused.h); same wave mapping and weight streaming as the forward kernel
// ---------------------------------------------------------------------------------------------------------------
// one hidden layer of the chain: acc = G (LDS tile [128][256] bf16) . WT^T   with WT = [in v][out u] row-major (u contiguous)
__device__ __forceinline__ void bwd_product(f32x16 (&acc)[4], const __bf16 (*g)[AP], __bf16 (*wb)[CH][WP], const __bf16* __restrict__ WT,
                                            int tid, int lr, int lh, int wm, int wn) {
#pragma unroll
  for (int ct = 0; ct < 4; ++ct)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[ct][r] = 0.f;
  const int wn_ = tid >> 1, wh = (tid & 1) * 16;
  const __bf16* __restrict__ wsrc = WT + (long)wn_ * CH + wh;
  u32x4 w0 = *reinterpret_cast<const u32x4*>(wsrc), w1 = *reinterpret_cast<const u32x4*>(wsrc + 8);
  *reinterpret_cast<u32x4*>(&wb[0][wn_][wh]) = w0;
  *reinterpret_cast<u32x4*>(&wb[0][wn_][wh + 8]) = w1;
  __syncthreads();          // (also: the gradient tile of this layer is complete)
  int cur = 0;
#pragma unroll 1
  for (int kc = 0; kc < CH / WKC; ++kc) {
    const bool more = kc + 1 < CH / WKC;
    if (more) {
      w0 = *reinterpret_cast<const u32x4*>(wsrc + (kc + 1) * WKC);
      w1 = *reinterpret_cast<const u32x4*>(wsrc + (kc + 1) * WKC + 8);
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const bf16x8 af = *reinterpret_cast<const bf16x8*>(&g[wm * 32 + lr][kc * WKC + ks * 16 + 8 * lh]);
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) {
        const bf16x8 bfr = *reinterpret_cast<const bf16x8*>(&wb[cur][wn * 128 + ct * 32 + lr][ks * 16 + 8 * lh]);
        acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bfr, acc[ct], 0, 0, 0);
      }
    }
    if (more) {
      *reinterpret_cast<u32x4*>(&wb[cur ^ 1][wn_][wh]) = w0;
      *reinterpret_cast<u32x4*>(&wb[cur ^ 1][wn_][wh + 8]) = w1;
    }
    __syncthreads();
    cur ^= 1;
  }
}

// RUNS (round 5): a workgroup owns a contiguous RUN of tiles in (estimator, y block of 128 rows, x row i) order and keeps dQ = sum_i dZ0 of
// its (estimator, y block) in registers (the accumulator layout of the last product: 64 values per lane), added into a.dQ when the run
// leaves the block or ends: dz0 is not written at all (it was 335 MB out and, through pair_reduce_q, 335 MB back in per pass at cfg3).
template <bool COMPACT, bool WG, bool RUNS>   // COMPACT: bitmask / bf16 inputs; WG: stage 1 (weight-gradient operands and column-sum gradients)
__global__ __launch_bounds__(512) void concat_bwd_kernel(ConcatBwdArgs a) {
  __shared__ __attribute__((aligned(16))) __bf16 gt[CR][AP];
  __shared__ __attribute__((aligned(16))) __bf16 wb[2][CH][WP];
  __shared__ LdsAcc cs[2][CH];            // column sums of the tile (two quantities at a time)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 31, lh = lane >> 5;
  const int wm = wave & 3, wn = wave >> 2;
  const int B = a.B;
  constexpr bool wg = WG;
  const int tiles_e = (int)(((long)B * B) / CR);
  int lin, lin_end;
  if (RUNS) {
    const int total = a.E * tiles_e, per = (total + (int)gridDim.x - 1) / (int)gridDim.x;
    lin = blockIdx.x * per; lin_end = min(total, lin + per);
  } else { lin = blockIdx.y * tiles_e + blockIdx.x; lin_end = lin + 1; }
  float dq[RUNS ? 4 : 1][RUNS ? 16 : 1];
  if (RUNS) {
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
      for (int r = 0; r < 16; ++r) dq[RUNS ? ct : 0][RUNS ? r : 0] = 0.f;
  }
#pragma unroll 1
 for (; lin < lin_end; ++lin) {
  const int e = lin / tiles_e, rem = lin - e * tiles_e;
  // RUNS: tile order (y block, x row) inside an estimator; otherwise the natural row order (x row, y block)
  const long row0 = RUNS ? (long)(rem % B) * B + (long)(rem / B) * CR : (long)rem * CR;
  const long ebase = (long)e * B * B, tile = (ebase + row0) * CH;
  if (tid < CH) { cs[0][tid].zero(); cs[1][tid].zero(); }
  __syncthreads();
  CPH_DECL; CPH_COUNT();
  // compact: the sign words of this wave's 32 rows x 4 column groups are ONE 16-byte load per lane and layer (lane l and l + 32: row l of
  // the block), requested here, a whole phase ahead of their use; the word of (row, group) then comes out of lane `row` with v_readlane
  // (round 5: 64 broadcast loads and 64 registers per layer before)
  u32x4 m1row4 = {0u, 0u, 0u, 0u}, m0row4 = m1row4;
  if (COMPACT) {
    m1row4 = *reinterpret_cast<const u32x4*>(a.m1 + (ebase + row0 + wm * 32 + lr) * 8 + wn * 4);
    m0row4 = *reinterpret_cast<const u32x4*>(a.m0 + (ebase + row0 + wm * 32 + lr) * 8 + wn * 4);
  }
  // ---- dZ2 = ds w3^T (.) [a2 > 0]; this thread owns one column quad (c4) and 16 of the 128 rows
  {
    const int c4 = (tid & 63) * 4;
    const float4 w3v = *reinterpret_cast<const float4*>(a.w3 + (long)e * a.pstride + c4);
    float4 sdb = make_float4(0.f, 0.f, 0.f, 0.f), sdw = sdb;
    float sds = 0.f;
    // (compact: all 16 rows' sign words and ds values of this thread requested at once -- one round trip; the 4-at-a-time loop that the
    //  fp32 a2 path needs was four, and in stage 1 each waited for the previous batch's dZ2 stores as well: stores count on vmcnt)
    constexpr int NQ = COMPACT ? 16 : 4;
#pragma unroll 1
    for (int base = 0; base < CR * 64; base += 512 * NQ) {
      float4 av[NQ]; float dsv[NQ]; uint32_t mw[NQ];
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const int row = (base + tid + 512 * q) >> 6;
        if (COMPACT) {
          mw[q] = a.m2[(ebase + row0 + row) * 8 + (c4 >> 5)];
        } else {
          av[q] = *reinterpret_cast<const float4*>(a.a2 + tile + (long)row * CH + c4);
        }
        dsv[q] = a.ds[ebase + row0 + row];
      }
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const int row = (base + tid + 512 * q) >> 6;
        const float d = dsv[q];
        float4 v;
        if (COMPACT) {   // only the signs are needed (bitmask); round 5: in stage 1 too -- dw3 = sum ds a2 is concat_dw3_kernel's
          const uint32_t bits = mw[q] >> (c4 & 31);
          v.x = (bits & 1u) ? d * w3v.x : 0.f; v.y = (bits & 2u) ? d * w3v.y : 0.f;
          v.z = (bits & 4u) ? d * w3v.z : 0.f; v.w = (bits & 8u) ? d * w3v.w : 0.f;
        } else {
          v.x = av[q].x > 0.f ? d * w3v.x : 0.f; v.y = av[q].y > 0.f ? d * w3v.y : 0.f;
          v.z = av[q].z > 0.f ? d * w3v.z : 0.f; v.w = av[q].w > 0.f ? d * w3v.w : 0.f;
        }
        bf16x4 b; b[0] = to_bf16(v.x); b[1] = to_bf16(v.y); b[2] = to_bf16(v.z); b[3] = to_bf16(v.w);
        *reinterpret_cast<bf16x4*>(&gt[row][c4]) = b;
        if (wg) {
          if (!COMPACT) *reinterpret_cast<bf16x4*>(a.dz2 + tile + (long)row * CH + c4) = b;
          sdb.x += v.x; sdb.y += v.y; sdb.z += v.z; sdb.w += v.w;
          if (!COMPACT) { sdw.x += d * av[q].x; sdw.y += d * av[q].y; sdw.z += d * av[q].z; sdw.w += d * av[q].w; }
          if (c4 == 0) sds += d;
        }
      }
    }
    if (wg) {
      cs[0][c4].add(sdb.x); cs[0][c4 + 1].add(sdb.y); cs[0][c4 + 2].add(sdb.z); cs[0][c4 + 3].add(sdb.w);
      if (!COMPACT) { cs[1][c4].add(sdw.x); cs[1][c4 + 1].add(sdw.y); cs[1][c4 + 2].add(sdw.z); cs[1][c4 + 3].add(sdw.w); }
      sds = wave_sum(sds);                       // (only lane 0 of each wave, whose column quad is 0, carries a value)
    }
    __syncthreads();
    if (wg) {
      if (tid < CH) {
        acc_add(a.db2 + (long)e * a.pstride + tid, cs[0][tid].get());
        if (!COMPACT) acc_add(a.dw3 + (long)e * a.pstride + tid, cs[1][tid].get());
        cs[0][tid].zero(); cs[1][tid].zero();
      }
      if (lane == 0) acc_add(a.db3 + (long)e * a.pstride, sds);
    }
  }
  CPH(0);
  // ---- dZ1 = (dZ2 W2) (.) [a1 > 0]
  f32x16 acc[4];
  const unsigned soff = (unsigned)(4 * lh * CH + wn * 128 + lr);
  bwd_product(acc, gt, wb, a.W2T + (long)e * a.pstride, tid, lr, lh, wm, wn);
  if (wg && COMPACT) {
    // dZ2 leaves from the LDS tile in whole 512-byte rows, HERE: stores count on vmcnt like loads on this ISA, so the next wait for a load
    // also waits for every older store -- behind the product's last weight chunk the next load is an epilogue (no loads) away; in front
    // of the product its first chunk waited for them (stage-1 products 6.5 us per tile against 4.1 in stage 2)
#pragma unroll
    for (int p8 = 0; p8 < 8; ++p8) {
      const int row = p8 * 16 + (tid >> 5), c8 = (tid & 31) * 8;
      *reinterpret_cast<u32x4*>(a.dz2 + tile + (long)row * CH + c8) = *reinterpret_cast<const u32x4*>(&gt[row][c8]);
    }
    __syncthreads();     // (the epilogue below overwrites the tile in place)
  }
  CPH(1);
  {
    float csum[4] = {0.f, 0.f, 0.f, 0.f};
    const u32x4 mrow4 = m1row4;
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
      float mk[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int mu = wm * 32 + (r & 3) + 8 * (r >> 2);
        if (COMPACT) {
          const int rl = (r & 3) + 8 * (r >> 2);
          const uint32_t w_lo = __builtin_amdgcn_readlane(mrow4[ct], rl), w_hi = __builtin_amdgcn_readlane(mrow4[ct], rl + 4);
          const uint32_t wrd = lh ? w_hi : w_lo;
          mk[r] = ((wrd >> lr) & 1u) ? 1.f : 0.f;
        } else {
          const float* __restrict__ mrow = a.a1 + tile + (long)mu * CH + ct * 32;
          mk[r] = mrow[soff];
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int mu = wm * 32 + (r & 3) + 8 * (r >> 2), m = mu + 4 * lh, col = wn * 128 + ct * 32 + lr;
        const float v = mk[r] > 0.f ? acc[ct][r] : 0.f;
        const __bf16 vb = to_bf16(v);
        gt[m][col] = vb;                                   // (every wave left the product loop through its last barrier)
        if (wg) csum[ct] += v;
      }
    }
    if (wg) {
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) cs[0][wn * 128 + ct * 32 + lr].add(csum[ct]);
    }
  }
  __syncthreads();
  if (wg && tid < CH) { acc_add(a.db1 + (long)e * a.pstride + tid, cs[0][tid].get()); cs[0][tid].zero(); }
  CPH(2);
  // ---- dZ0 = (dZ1 W1) (.) [a0 > 0]  -> dz0 (fp32) and its column sums = dP[i]
  bwd_product(acc, gt, wb, a.W1T + (long)e * a.pstride, tid, lr, lh, wm, wn);
  if (wg) {
    // dZ1 leaves from the LDS tile in whole 512-byte rows (round 5; it left as 2-byte stores from the accumulator layout: 64 store
    // instructions per lane, two 64-byte row pieces each) -- behind the product for the reason given at dZ2; the tile is not written again
    // before the next tile's first barrier
#pragma unroll
    for (int p8 = 0; p8 < 8; ++p8) {
      const int row = p8 * 16 + (tid >> 5), c8 = (tid & 31) * 8;
      *reinterpret_cast<u32x4*>(a.dz1 + tile + (long)row * CH + c8) = *reinterpret_cast<const u32x4*>(&gt[row][c8]);
    }
  }
  CPH(3);
  {
    float csum[4] = {0.f, 0.f, 0.f, 0.f};
    const u32x4 mrow4 = m0row4;
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
      float mk[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int mu = wm * 32 + (r & 3) + 8 * (r >> 2);
        if (COMPACT) {   // (rounds 2-4 recomputed layer 0's sign from P_i + Q_j: 64 loads per lane)
          const int rl = (r & 3) + 8 * (r >> 2);
          const uint32_t w_lo = __builtin_amdgcn_readlane(mrow4[ct], rl), w_hi = __builtin_amdgcn_readlane(mrow4[ct], rl + 4);
          mk[r] = (((lh ? w_hi : w_lo) >> lr) & 1u) ? 1.f : 0.f;
        } else {
          const float* __restrict__ mrow = a.a0 + tile + (long)mu * CH + ct * 32;
          mk[r] = mrow[soff];
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int mu = wm * 32 + (r & 3) + 8 * (r >> 2);
        const float v = mk[r] > 0.f ? acc[ct][r] : 0.f;
        if (RUNS) dq[RUNS ? ct : 0][RUNS ? r : 0] += v;
        else { float* __restrict__ drow = a.dz0 + tile + (long)mu * CH + ct * 32; drow[soff] = v; }
        csum[ct] += v;
      }
    }
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) cs[0][wn * 128 + ct * 32 + lr].add(csum[ct]);
  }
  __syncthreads();
  if (tid < CH) {
    const long i = row0 / B;                               // B % 128 == 0: the whole tile belongs to one x row
    float* dp = a.dP + ((long)e * B + i) * CH + tid;
    if (B == CR) *dp = cs[0][tid].get(); else acc_add(dp, cs[0][tid].get());
  }
  CPH(4);
  if (RUNS && (lin + 1 == lin_end || (rem + 1) % B == 0)) {
    // the run leaves this (estimator, y block) = block `blk` of B tiles: its dQ partial goes to the block's slot for this workgroup
    // (plain stores; concat_dq_reduce_kernel adds a block's slots in workgroup order: run-to-run reproducible, unlike float atomics --
    //  whose 1-ulp order noise in dQ the bf16 roundings of the whole main-model backward amplified to 1.4e-4 of a GRU weight gradient)
    const int per = (a.E * tiles_e + (int)gridDim.x - 1) / (int)gridDim.x;
    const int blk = lin / B, wfirst = (blk * B) / per;
    float* __restrict__ slot = a.dq_part + ((long)blk * a.dq_slots + ((int)blockIdx.x - wfirst)) * (CR * CH);   // (wave-uniform base ...
    const unsigned lane_off = (unsigned)((wm * 32 + 4 * lh) * CH + wn * 128 + lr);                             //  ... + one 32-bit lane offset)
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        slot[lane_off + (unsigned)(((r & 3) + 8 * (r >> 2)) * CH + ct * 32)] = dq[RUNS ? ct : 0][RUNS ? r : 0];
        dq[RUNS ? ct : 0][RUNS ? r : 0] = 0.f;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  CPH(5);
 }
}

}  // namespace

// dQ[block][m][c] = sum over the block's slots, in workgroup order (see the flush in concat_bwd_kernel).  Grid (blocks, 16): 8 rows each.
namespace {
__global__ __launch_bounds__(512) void concat_dq_reduce_kernel(const float* __restrict__ part, float* __restrict__ dQ, int B, int per, int slots) {
  const int blk = blockIdx.x, wfirst = (blk * B) / per, wlast = ((blk + 1) * B - 1) / per, n = wlast - wfirst + 1;
  const int off = blockIdx.y * (8 * CH) + threadIdx.x * 4;            // 512 threads x float4 = 8 rows of 256
  const float* __restrict__ p = part + (long)blk * slots * (CR * CH) + off;
  float4 s = *reinterpret_cast<const float4*>(p);
  for (int k = 1; k < n; ++k) {
    const float4 v = *reinterpret_cast<const float4*>(p + (long)k * (CR * CH));
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  *reinterpret_cast<float4*>(dQ + (long)blk * (CR * CH) + off) = s;   // block = (estimator, y block of 128 rows): dQ is [E][B][256]
}
}  // namespace

// dw3[e][c] += sum_rows ds[e][row] * a2[e][row][c]: the score head's weight gradient (VMI.py:33), the one consumer of the fp32 a2 values.
// Its own streaming launch since round 5 (beside the weight-gradient GEMMs on the helper stream): inside concat_bwd_kernel the 128 KB a2
// tile was read in front of every tile's products (12 of 34 us per tile at cfg3, HBM-bound, nothing to overlap it with).
// Workgroup = 256 threads = 64 column quads x 4 row phases, 512 rows; 8 rows per thread in flight.
namespace {
__global__ __launch_bounds__(256) void concat_dw3_kernel(const float* __restrict__ ds, const float* __restrict__ a2, float* __restrict__ dw3,
                                                         long rows, long pstride) {
  __shared__ float part[4][CH];
  const int e = blockIdx.y, c4 = (threadIdx.x & 63) * 4, ph = threadIdx.x >> 6;
  const long r0 = (long)blockIdx.x * 512, r1 = min(rows, r0 + 512);
  const float* __restrict__ dse = ds + (long)e * rows;
  const float* __restrict__ ae = a2 + (long)e * rows * CH;
  float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
  for (long r = r0 + ph; r < r1; r += 32) {
    float4 v[8]; float d[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const long rr = min(r + 4 * q, r1 - 1);
      v[q] = *reinterpret_cast<const float4*>(ae + rr * CH + c4);
      d[q] = r + 4 * q < r1 ? dse[rr] : 0.f;
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) { sum.x += d[q] * v[q].x; sum.y += d[q] * v[q].y; sum.z += d[q] * v[q].z; sum.w += d[q] * v[q].w; }
  }
  *reinterpret_cast<float4*>(&part[ph][c4]) = sum;
  __syncthreads();
  const int c = threadIdx.x;
  acc_add(dw3 + (long)e * pstride + c, part[0][c] + part[1][c] + part[2][c] + part[3][c]);
}
}  // namespace
// a2 stored as fp16 (the weights-stationary forward kernel, concat_fwd_a2_f16): 8 columns = 16 bytes per lane, 32 lanes per row, 8 row phases;
// 8 loads per thread in flight, 1024 rows per workgroup
namespace {
__global__ __launch_bounds__(256) void concat_dw3_h_kernel(const float* __restrict__ ds, const _Float16* __restrict__ a2, float* __restrict__ dw3,
                                                           long rows, long pstride) {
  __shared__ float part[8][CH];
  const int e = blockIdx.y, c8 = (threadIdx.x & 31) * 8, ph = threadIdx.x >> 5;
  const long r0 = (long)blockIdx.x * 1024, r1 = min(rows, r0 + 1024);
  const float* __restrict__ dse = ds + (long)e * rows;
  const _Float16* __restrict__ ae = a2 + (long)e * rows * CH;
  float sum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (long r = r0 + ph; r < r1; r += 64) {
    f16x8 v[8]; float d[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const long rr = min(r + 8 * q, r1 - 1);
      v[q] = *reinterpret_cast<const f16x8*>(ae + rr * CH + c8);
      d[q] = r + 8 * q < r1 ? dse[rr] : 0.f;
    }
#pragma unroll
    for (int q = 0; q < 8; ++q)
#pragma unroll
      for (int j = 0; j < 8; ++j) sum[j] += d[q] * (float)v[q][j];
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) part[ph][c8 + j] = sum[j];
  __syncthreads();
  const int c = threadIdx.x;
  float t = 0.f;
#pragma unroll
  for (int p = 0; p < 8; ++p) t += part[p][c];
  acc_add(dw3 + (long)e * pstride + c, t);
}
}  // namespace
int concat_dw3(hipStream_t s, const float* ds, const float* a2, float* dw3, int E, int B, long pstride, bool a2_f16) {
  const long rows = (long)B * B;
  if (a2_f16) hipLaunchKernelGGL(concat_dw3_h_kernel, dim3((unsigned)((rows + 1023) / 1024), E), dim3(256), 0, s, ds, reinterpret_cast<const _Float16*>(a2), dw3, rows, pstride);
  else hipLaunchKernelGGL(concat_dw3_kernel, dim3((unsigned)((rows + 511) / 512), E), dim3(256), 0, s, ds, a2, dw3, rows, pstride);
  LAUNCH_CHECK();
  return MIMRL_OK;
}

#ifdef MIMRL_PHASE_PROBE
int concat_bwd_read_phases(long long* out) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_cc_phase), sizeof(long long) * 16) != hipSuccess) return 1;
  long long z[16] = {0};
  return hipMemcpyToSymbol(HIP_SYMBOL(g_cc_phase), z, sizeof z) == hipSuccess ? 0 : 1;
}
#endif

// Partition of the E * B * B / 128 tiles into one run per CU (per tiles each); a block of B tiles = one (estimator, y block) is covered by
// at most `slots` consecutive workgroups.  False: too few tiles for runs (per < 2) -- the caller keeps dz0 + pair_reduce_q.
bool concat_bwd_dq_plan(int E, int B, int* per, int* nwg, int* slots) {
  static int cus = 0;
  if (!cus) { hipDeviceProp_t pr; int dev = 0; if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&pr, dev) != hipSuccess) return false; cus = pr.multiProcessorCount > 0 ? pr.multiProcessorCount : 256; }
  if (B < CR || B % CR != 0) return false;
  const int total = E * (int)(((long)B * B) / CR);
  const int p = (total + cus - 1) / cus;
  if (p < 2) return false;
  *per = p; *nwg = (total + p - 1) / p; *slots = (B + p - 1) / p + 1;
  return true;
}
// floats of scratch the plan needs (<= E * B * B * 256, the dz0 buffer it replaces)
long concat_bwd_dq_scratch(int E, int B) {
  int per, nwg, slots;
  if (!concat_bwd_dq_plan(E, B, &per, &nwg, &slots)) return 0;
  return std::max((long)E * (B / CR) * slots * CR * CH, concat_bwd_ws_scratch(E, B));   // (either kernel may be the one that runs)
}

bool concat_bwd_fused_supported(int B, int hid) { return hid == CH && B >= CR && B % CR == 0; }

int concat_bwd_fused(hipStream_t s, const ConcatBwdArgs& a) {
  if (!concat_bwd_fused_supported(a.B, CH)) return set_error(MIMRL_ERR_ARG, "concat_bwd_fused: batch %d unsupported", a.B);
  if (!(a.dz0 || a.dQ) || !a.dP || !a.ds) return set_error(MIMRL_ERR_ARG, "concat_bwd_fused: null argument");
  if (a.compact ? !(a.m0 && a.m1 && a.m2) : !(a.a0 && a.a1 && a.a2))
    return set_error(MIMRL_ERR_ARG, "concat_bwd_fused: null activation input");
  if ((a.dz2 != nullptr) != (a.dz1 != nullptr) || (a.dz2 && !(a.db1 && a.db2 && a.dw3 && a.db3)))
    return set_error(MIMRL_ERR_ARG, "concat_bwd_fused: the weight-gradient outputs come together");
  const bool wg = a.dz2 != nullptr;
  // round 6: the weights-stationary kernel (concat_ws_bwd.hip) wherever the in-kernel dQ reduction is on; MIMRL_CONCAT_STREAMED=1: the kernel below
  if (a.dQ && a.compact && a.dq_part && concat_bwd_ws_supported(a.B, CH) && !knob_on("MIMRL_CONCAT_STREAMED")) return concat_bwd_ws(s, a);
  if (a.dQ) {   // runs of tiles with dQ in registers: one workgroup per CU, ceil(tiles / CUs) tiles each
    if (!a.compact || !a.dq_part) return set_error(MIMRL_ERR_ARG, "concat_bwd_fused: the in-kernel dQ reduction needs the compact saves and its scratch");
    int per = 0, nwg = 0, slots = 0;
    if (!concat_bwd_dq_plan(a.E, a.B, &per, &nwg, &slots)) return set_error(MIMRL_ERR_ARG, "concat_bwd_fused: no dQ plan for E = %d, B = %d", a.E, a.B);
    ConcatBwdArgs b = a;
    b.dq_slots = slots;
    const dim3 grid((unsigned)nwg);
    if (wg) hipLaunchKernelGGL((concat_bwd_kernel<true, true, true>), grid, dim3(512), 0, s, b);
    else hipLaunchKernelGGL((concat_bwd_kernel<true, false, true>), grid, dim3(512), 0, s, b);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(concat_dq_reduce_kernel, dim3((unsigned)(a.E * (a.B / CR)), 16), dim3(512), 0, s, a.dq_part, a.dQ, a.B, per, slots);
    LAUNCH_CHECK();
    return MIMRL_OK;
  }
  const dim3 grid((unsigned)(((long)a.B * a.B) / CR), a.E);
  if (a.compact) { if (wg) hipLaunchKernelGGL((concat_bwd_kernel<true, true, false>), grid, dim3(512), 0, s, a); else hipLaunchKernelGGL((concat_bwd_kernel<true, false, false>), grid, dim3(512), 0, s, a); }
  else { if (wg) hipLaunchKernelGGL((concat_bwd_kernel<false, true, false>), grid, dim3(512), 0, s, a); else hipLaunchKernelGGL((concat_bwd_kernel<false, false, false>), grid, dim3(512), 0, s, a); }
  LAUNCH_CHECK();
  return MIMRL_OK;
}

bool concat_fwd_fused_supported(int B, int hid) { return hid == CH && B >= 16 && ((long)B * B) % CR == 0; }

// does a saving forward pass of this shape leave a2 as fp16 (the weights-stationary kernel, round 6) rather than fp32?  The score head's
// weight gradient sum_rows ds * a2 cancels to ~1e-3 of its terms: bf16's 8 bits were not enough (round 4), fp16's 11 give ~2e-3 of the
// gradient's scale at B = 256 (sqrt(N) 2^-11 against N 1e-3) -- and 168 MB less to write and to read per critic pass at cfg3
bool concat_fwd_a2_f16(int B, int save) { return save == 2 && concat_fwd_ws_supported(B, CH, save) && !knob_on("MIMRL_CONCAT_STREAMED"); }

int concat_fwd_fused(hipStream_t s, const ConcatFwdArgs& a) {
  if (!concat_fwd_fused_supported(a.B, CH)) return set_error(MIMRL_ERR_ARG, "concat_fwd_fused: batch %d unsupported", a.B);
  // round 6: the weights-stationary persistent kernel (concat_ws.hip) wherever it applies (B a multiple of 32, compact or no saves);
  // MIMRL_CONCAT_STREAMED=1 keeps the weight-streaming kernel below (read per call: tests run both in one process)
  if (concat_fwd_ws_supported(a.B, CH, a.save) && !knob_on("MIMRL_CONCAT_STREAMED")) return concat_fwd_ws(s, a);
  if (!a.scores || a.save < 0 || a.save > 3) return set_error(MIMRL_ERR_ARG, "concat_fwd_fused: bad arguments");
  if ((a.save == 1 && !(a.a0 && a.a1 && a.a2)) || (a.save == 2 && !(a.a0b && a.a1b && a.a2)) || (a.save >= 2 && !(a.m0 && a.m1 && a.m2)))
    return set_error(MIMRL_ERR_ARG, "concat_fwd_fused: null save buffer");
  const dim3 grid((unsigned)(((long)a.B * a.B) / CR), a.E);
  if (a.save == 0) hipLaunchKernelGGL(concat_fwd_kernel<0>, grid, dim3(512), 0, s, a);
  else if (a.save == 1) hipLaunchKernelGGL(concat_fwd_kernel<1>, grid, dim3(512), 0, s, a);
  else if (a.save == 2) hipLaunchKernelGGL(concat_fwd_kernel<2>, grid, dim3(512), 0, s, a);
  else hipLaunchKernelGGL(concat_fwd_kernel<3>, grid, dim3(512), 0, s, a);
  LAUNCH_CHECK();
  return MIMRL_OK;
}

}  // namespace mimrl
