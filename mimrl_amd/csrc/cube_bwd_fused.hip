// Fused data-gradient chains of one CubeMLP block (see cube_bwd_fused.h).
#include "cube_bwd_fused.h"
#include "kmix_device.h"

namespace mimrl {

namespace {

// `make PHASE_PROBE=1` (tools/cube_phase.py): workgroup (0, 0) leaves 100 MHz ticks at phase boundaries of its last launch;
// slots: L axis 0..15 (ol <= 32) / 16..31 (ol > 32), D axis 32..47 (<= 200 workgroups) / 48..63
#ifdef MIMRL_PHASE_PROBE
__device__ long long g_cbwd_phase[64];
#define BPHASE(base, i) do { if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) g_cbwd_phase[(base) + (i)] = (long long)wall_clock64(); } while (0)
#else
#define BPHASE(base, i) do { } while (0)
#endif

constexpr int CT = 128;       // L axis: columns per workgroup
constexpr int KP = 64 + 8;    // bf16 pitch of a [.][<=64] operand image (144 B: conflict-free 16-byte fragment reads)

// ---------------------------------------------------------------------------------------------------------------
// L axis.  Workgroup = (sample b, 128 columns).  MFMA 32x32x16: M = rows of the transposed weight (h or i), N = columns,
// K = o or h.  dY / dU are kept TRANSPOSED in LDS ([column][k]) so that B-fragments are 16-byte reads; the weights
// are staged transposed and zero-padded to 64x64, which also takes care of ragged hl / ol / il.
// ---------------------------------------------------------------------------------------------------------------

// LONG (round 5b): il > 64 -- the input axis of a long-sequence block (cfg3 / cfg5: L = 500 / 1000 -> 50 -> 50).  Phases 1 and 2 are the short
// kernel's (LayerNorm backward over the <= 64 output rows, dU over the <= 64 hidden rows); phase 3, dX = W1^T dU + Wr^T dY, walks the il output
// rows in tiles of 64 and stages the two transposed weight tiles per step (requested one tile ahead).  One launch instead of colln_bwd + two
// GEMM launches with dY / dU round trips in between (192 us of the cfg3 chain).
template <bool LONG = false>
__global__ __launch_bounds__(256, LONG ? 2 : 1) void laxis_bwd_kernel(LAxisBwdArgs a) {   // (LONG: two workgroups per CU -- 768 workgroups of ~45 us each at cfg3)
  __shared__ __attribute__((aligned(16))) __bf16 sdyu[2][CT][KP];
  auto& sdy = sdyu[0];
  auto& sdu = sdyu[1];
  __shared__ __attribute__((aligned(16))) __bf16 wts[3][64][KP];
  // wts[0] = W2^T [h][o], wts[1] = W1^T [i][h], wts[2] = Wr^T [i][o]
  float (*red)[2][CT] = reinterpret_cast<float (*)[2][CT]>(&sdu[0][0]);   // phase-1 scratch; sdu is first written in phase 2
  __shared__ float acc_l[3][64];   // per-l partial sums of dgamma, dbeta, db2
  __shared__ LdsAcc acc_h[64];      // per-h partial sums of db1
  __shared__ float sgam[64];       // LayerNorm gain (tools/isa_lint.py: read from global inside `l < ol ? ... gamma[l]` it was 32 guarded loads, each behind vmcnt(0))
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 31, lh = lane >> 5;
  const int b = blockIdx.y, c0 = blockIdx.x * CT;
  const int il = a.il, hl = a.hl, ol = a.ol, C = a.C;
  const int pb = ol > 32 ? 16 : 0;
  BPHASE(pb, 0);

  // pre-activations for act'(U) of phase 2 (this wave's 32 columns x up to 64 rows h): requested before anything else
  // -- loaded in that phase's epilogue they were a dependent memory round trip in the middle of the kernel
  float uu[2][16];
  {
    const int ccu = c0 + wave * 32 + lr;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int h = min(mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh, hl - 1);
        uu[mt][r] = a.u[((long)b * hl + h) * C + ccu];
      }
  }
  // weights, transposed and zero-padded to 64x64: zero fill with 16-byte stores, then scatter from coalesced reads
  {
    uint4* z = reinterpret_cast<uint4*>(&wts[0][0][0]);
    for (int i = tid; i < 3 * 64 * KP * 2 / 16; i += 256) z[i] = make_uint4(0u, 0u, 0u, 0u);
    if (tid < 64) { acc_l[0][tid] = 0.f; acc_l[1][tid] = 0.f; acc_l[2][tid] = 0.f; acc_h[tid].zero(); sgam[tid] = a.gamma[tid < ol ? tid : ol - 1]; }
  }
  __syncthreads();
  BPHASE(pb, 1);
  {   // (tools/isa_lint.py: as three run-time-trip-count loops of load -> LDS store these were ~30 loads each behind an s_waitcnt vmcnt(0):
      //  3 us per workgroup.  Now <= 16 elements per thread and matrix, requested together from clamped indices, then scattered)
    const int n2 = ol * hl, n1 = hl * il, nr = ol * il;
    float wv[3][16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int idx = tid + 256 * j;
      wv[0][j] = a.w2[idx < n2 ? idx : n2 - 1];
      if constexpr (!LONG) {
        wv[1][j] = a.w1[idx < n1 ? idx : n1 - 1];
        wv[2][j] = a.wr[idx < nr ? idx : nr - 1];
      }
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int idx = tid + 256 * j;
      if (idx < n2) { const int o = idx / hl, h = idx - o * hl; wts[0][h][o] = to_bf16(wv[0][j]); }
      if constexpr (!LONG) {
        if (idx < n1) { const int h = idx / il, i = idx - h * il; wts[1][i][h] = to_bf16(wv[1][j]); }
        if (idx < nr) { const int o = idx / il, i = idx - o * il; wts[2][i][o] = to_bf16(wv[2][j]); }
      }
    }
  }

  BPHASE(pb, 2);
  // ---- phase 1: LayerNorm over L, backward.  Thread = (column, half of the rows l = half, half+2, ...); two passes over
  // the (L2-resident) column instead of 64 live registers
  const int col = tid & (CT - 1), half = tid >> 7;
  const long cidx = (long)b * C + c0 + col;
  const float mu = a.mean[cidx], rs = a.rstd[cidx];
  const float* __restrict__ dzb = a.dz + (long)b * ol * C + c0 + col;
  const float* __restrict__ yb = a.y + (long)b * ol * C + c0 + col;
  float* __restrict__ dyb = a.dy + (long)b * ol * C + c0 + col;
  // Round 3b (tools/cube_phase.py: 9.2 + 6.1 of this kernel's 29 us per workgroup at ol = 50): the thread's <= 32 rows of dz and y are
  // requested in ONE batch and stay in registers for the second pass -- they were four 8-row batches (four round trips), read twice
  float gq[32], xq[32];
#pragma unroll
  for (int j = 0; j < 32; ++j) {
    const int l = min(half + 2 * j, ol - 1);          // clamped, unconditional (rows beyond ol are masked below)
    gq[j] = dzb[(long)l * C]; xq[j] = yb[(long)l * C];
  }
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int j = 0; j < 32; ++j) {
    const int l = half + 2 * j;
    xq[j] = (xq[j] - mu) * rs;
    gq[j] = l < ol ? gq[j] * sgam[min(l, 63)] : 0.f;
    s1 += gq[j]; s2 += gq[j] * xq[j];
  }
  BPHASE(pb, 3);
  red[0][half][col] = s1;
  red[1][half][col] = s2;
  __syncthreads();
  s1 = (red[0][0][col] + red[0][1][col]) / ol;
  s2 = (red[1][0][col] + red[1][1][col]) / ol;
#pragma unroll
  for (int j = 0; j < 32; ++j) {
    const int l = half + 2 * j;
    if (l < ol) {
      const float v = rs * (gq[j] - s1 - xq[j] * s2);
      dyb[(long)l * C] = v;
      sdy[col][l] = to_bf16(v);
    }
  }
  for (int l = ol + ((half ^ ol) & 1); l < 64; l += 2) sdy[col][l] = to_bf16(0.f);   // rows l >= ol: zero padding of the K axis
  __syncthreads();

  if (a.db2 && tid >= 192 && tid - 192 < ol) {   // db2[o] = sum over this tile's columns of dY (bf16 image, as the weight-gradient GEMM sees it)
    float t = 0.f;
    for (int c = 0; c < CT; ++c) t += (float)sdy[(c + tid) & (CT - 1)][tid - 192];
    acc_l[2][tid - 192] = t;
  }
  BPHASE(pb, 4);
  const int nt = wave;                 // this wave's 32-column tile; it owns both 32-row M tiles of it
  const int cc = c0 + nt * 32 + lr;
  // ---- phase 2: dU = (W2^T dY) * act'(U)
  {
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    // (all four k-steps and both 32-row M tiles, always: every image is zero-padded to 64 x 64, and a run-time trip count / a run-time
    //  `hl > 32` inside the loop is a branch between every MFMA and its fragment reads)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const bf16x8 bf = *reinterpret_cast<const bf16x8*>(&sdy[nt * 32 + lr][ks * 16 + 8 * lh]);
      const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(&wts[0][lr][ks * 16 + 8 * lh]);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, bf, acc0, 0, 0, 0);
      const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(&wts[0][32 + lr][ks * 16 + 8 * lh]);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, bf, acc1, 0, 0, 0);
    }
    act_dispatch(a.act, [&](auto AT) __attribute__((always_inline)) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int h = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        float v = 0.f;
        if (h < hl) {
          const long ui = ((long)b * hl + h) * C + cc;
          v = (mt == 0 ? acc0[r] : acc1[r]) * act_grad_c<decltype(AT)::value>(a.act, uu[mt][r]);
          a.du[ui] = v;
        }
        sdu[nt * 32 + lr][h] = to_bf16(v);
        if (a.db1) {
          const float t = half_sum_hi(v);                        // (valid in lanes 16..31 / 48..63)
          if (lr == 16 && h < hl) acc_h[h].add(t);
        }
      }
    });
  }
  __syncthreads();
  BPHASE(pb, 5);
  // ---- phase 3: dX = W1^T dU + Wr^T dY
  if constexpr (LONG) {
    // tiles of 64 output rows i: W1^T [i][h] and Wr^T [i][o] of the tile go through wts[1] / wts[2]; thread = (i = tid & 63, rows h / o =
    // (tid >> 6) + 4 j): a wave reads 64 consecutive i of one weight row (256 B).  The next tile's 32 values are requested before this tile's products.
    float w1v[16], wrv[16];
    // (thread = (i = tid & 63, 16 CONSECUTIVE rows h / o = 16 (tid >> 6) + j): its 16 values of a tile are 32 contiguous bytes of the transposed
    //  image -- two 16-byte LDS stores per matrix instead of 32 two-byte ones with 8-way bank conflicts, 2-3 us per tile and workgroup)
    auto wreq = [&](int i0) __attribute__((always_inline)) {
      const int i = min(i0 + (tid & 63), il - 1);
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int r = 16 * (tid >> 6) + j;
        w1v[j] = a.w1[(long)min(r, hl - 1) * il + i];
        wrv[j] = a.wr[(long)min(r, ol - 1) * il + i];
      }
    };
    wreq(0);
    for (int i0 = 0; i0 < il; i0 += 64) {
      __syncthreads();                      // the previous tile's fragment reads of wts[1] / wts[2] (first trip: phase 2's of sdy / wts[0]) are done
      {
        const int ii = tid & 63, r0 = 16 * (tid >> 6);
        const bool iok = i0 + ii < il;
        bf16x8 p1[2], p2[2];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          p1[j >> 3][j & 7] = to_bf16(iok && r0 + j < hl ? w1v[j] : 0.f);
          p2[j >> 3][j & 7] = to_bf16(iok && r0 + j < ol ? wrv[j] : 0.f);
        }
        *reinterpret_cast<bf16x8*>(&wts[1][ii][r0]) = p1[0]; *reinterpret_cast<bf16x8*>(&wts[1][ii][r0 + 8]) = p1[1];
        *reinterpret_cast<bf16x8*>(&wts[2][ii][r0]) = p2[0]; *reinterpret_cast<bf16x8*>(&wts[2][ii][r0 + 8]) = p2[1];
      }
      __syncthreads();
      wreq(i0 + 64 < il ? i0 + 64 : i0);    // unconditional (past the end: this tile again, never stored)
      f32x16 acc0, acc1;
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        const bool first = ks < 4;
        const int kk = (first ? ks : ks - 4) * 16 + 8 * lh;
        const bf16x8 bf = *reinterpret_cast<const bf16x8*>(first ? &sdu[nt * 32 + lr][kk] : &sdy[nt * 32 + lr][kk]);
        const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(&wts[first ? 1 : 2][lr][kk]);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, bf, acc0, 0, 0, 0);
        const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(&wts[first ? 1 : 2][32 + lr][kk]);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, bf, acc1, 0, 0, 0);
      }
      {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int i = i0 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (i < il) a.dx[((long)b * il + i) * C + cc] = mt == 0 ? acc0[r] : acc1[r];
          }
      }
    }
  } else {
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const bool first = ks < 4;
      const int kk = (first ? ks : ks - 4) * 16 + 8 * lh;
      const bf16x8 bf = *reinterpret_cast<const bf16x8*>(first ? &sdu[nt * 32 + lr][kk] : &sdy[nt * 32 + lr][kk]);
      const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(&wts[first ? 1 : 2][lr][kk]);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, bf, acc0, 0, 0, 0);
      const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(&wts[first ? 1 : 2][32 + lr][kk]);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, bf, acc1, 0, 0, 0);
    }
    {
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int i = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (i < il) a.dx[((long)b * il + i) * C + cc] = mt == 0 ? acc0[r] : acc1[r];
        }
    }
  }
  __syncthreads();
  BPHASE(pb, 6);
  if (a.db2 && tid < ol) acc_add(&a.db2[tid], acc_l[2][tid]);
  if (a.db1 && tid >= 64 && tid - 64 < hl) acc_add(&a.db1[tid - 64], acc_h[tid - 64].get());
  BPHASE(pb, 7);
}

}  // namespace

#ifdef MIMRL_PHASE_PROBE
int cube_bwd_read_phases(long long* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_cbwd_phase), sizeof(long long) * 64) == hipSuccess ? 0 : 1; }
#endif

bool laxis_bwd_supported(int il, int hl, int ol, int C) {   // (il > 64: the LONG instantiation)
  return il >= 1 && hl >= 1 && ol >= 1 && hl <= 64 && ol <= 64 && C % CT == 0;
}

int laxis_bwd_fused(hipStream_t s, const LAxisBwdArgs& a) {
  if (!laxis_bwd_supported(a.il, a.hl, a.ol, a.C)) return set_error(MIMRL_ERR_ARG, "laxis_bwd_fused: unsupported shape");
  if (!a.wr) return set_error(MIMRL_ERR_ARG, "laxis_bwd_fused: needs the residual projection");
  if (a.il > 64) hipLaunchKernelGGL(laxis_bwd_kernel<true>, dim3(a.C / CT, a.B), dim3(256), 0, s, a);
  else hipLaunchKernelGGL(laxis_bwd_kernel<false>, dim3(a.C / CT, a.B), dim3(256), 0, s, a);
  LAUNCH_CHECK();
  return MIMRL_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// D axis.  Rows are independent: workgroup = 32 rows, 8 threads per row for the LayerNorm, then MFMA with M = the 32 rows
// (A-fragments from the LDS images of dY / dU), N = 32 output columns per wave, K = 128.  B-fragments are 16-byte loads
// from the pre-transposed bf16 weight images (24 per wave in total: no LDS staging of weights at all).
// ---------------------------------------------------------------------------------------------------------------
namespace {

constexpr int DR = 32;         // rows per workgroup
constexpr int DP = 128 + 8;    // bf16 pitch (272 B)

__global__ __launch_bounds__(256) void daxis_bwd_kernel(DAxisBwdArgs a) {
  __shared__ __attribute__((aligned(16))) __bf16 sdy[DR][DP];
  __shared__ __attribute__((aligned(16))) __bf16 sdu[DR][DP];
  __shared__ LdsAcc spg[3][128];     // this workgroup's column sums: dgamma, dbeta, db2
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 31, lh = lane >> 5;
  const long r0 = (long)blockIdx.x * DR;
  const bool pg = a.dgamma != nullptr;
  const int pb = gridDim.x <= 200 ? 32 : 48;
  BPHASE(pb, 0);
  if (pg) for (int i = tid; i < 3 * 128; i += 256) (&spg[0][0])[i].zero();   // (visible after the barrier behind phase 1... see below)
  // Round 3b: the LayerNorm's own operands go out FIRST (loads return in order: behind the 24 weight fragments and the 16 u values they
  // were the last to arrive although phase 1 is the first to need them)
  const int row1 = tid >> 3, part1 = tid & 7;
  const long r1 = r0 + row1;
  const bool ok1 = r1 < a.R;
  const long rc1 = ok1 ? r1 : a.R - 1;
  const float mu1 = a.mean[rc1], rs1 = a.rstd[rc1];
  float4 gz4[4], yv4[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    gz4[q] = *reinterpret_cast<const float4*>(a.dz + rc1 * 128 + part1 * 16 + q * 4);
    yv4[q] = *reinterpret_cast<const float4*>(a.y + rc1 * 128 + part1 * 16 + q * 4);
  }
  // B-fragments of this wave's 32 output columns, all three products
  const int n = wave * 32 + lr;
  bf16x8 bw2[8], bw1[8], bwr[8];
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) {
    bw2[ks] = *reinterpret_cast<const bf16x8*>(a.w2t + n * 128 + ks * 16 + 8 * lh);
    bw1[ks] = *reinterpret_cast<const bf16x8*>(a.w1t + n * 128 + ks * 16 + 8 * lh);
    bwr[ks] = *reinterpret_cast<const bf16x8*>(a.wrt + n * 128 + ks * 16 + 8 * lh);
  }
  // pre-activations for act'(U) of phase 2: requested now -- loaded in that phase's epilogue they were a dependent memory
  // round trip in the middle of the kernel (every load here is unconditional with a clamped row: a guarded load is a
  // branch with an immediate vmcnt(0))
  float uu[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const long row = min(r0 + (r & 3) + 8 * (r >> 2) + 4 * lh, a.R - 1);
    uu[r] = a.u[row * 128 + n];
  }
  BPHASE(pb, 1);
  // ---- phase 1: LayerNorm over D, backward: 8 threads per row, 16 consecutive columns each
  {
    const int row = row1, part = part1;
    const long r = r1;
    const bool ok = ok1;
    float g[16], xh[16], dyk[16];
    float s1 = 0.f, s2 = 0.f;
    const float mu = mu1, rs = rs1;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 gz = gz4[q];
      const float4 yv = yv4[q];
      const float gg[4] = {gz.x, gz.y, gz.z, gz.w}, yy[4] = {yv.x, yv.y, yv.z, yv.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        g[q * 4 + j] = gg[j];
        xh[q * 4 + j] = (yy[j] - mu) * rs;
        const float dxh = gg[j] * a.gamma[part * 16 + q * 4 + j];
        s1 += dxh; s2 += dxh * xh[q * 4 + j];
      }
    }
    s1 = group_sum<8>(s1); s2 = group_sum<8>(s2);
    s1 *= (1.f / 128.f); s2 *= (1.f / 128.f);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int c = part * 16 + q * 4 + j;
        v[j] = rs * (g[q * 4 + j] * a.gamma[c] - s1 - xh[q * 4 + j] * s2);
      }
      if (ok) *reinterpret_cast<float4*>(a.dy + r * 128 + part * 16 + q * 4) = make_float4(v[0], v[1], v[2], v[3]);
      bf16x4 p; p[0] = to_bf16(v[0]); p[1] = to_bf16(v[1]); p[2] = to_bf16(v[2]); p[3] = to_bf16(v[3]);
      *reinterpret_cast<bf16x4*>(&sdy[row][part * 16 + q * 4]) = p;
      if (pg) {   // keep dY in g[] for the column sums below (dz is no longer needed once dgamma / dbeta terms are formed)
#pragma unroll
        for (int j = 0; j < 4; ++j) { const float gz = ok ? g[q * 4 + j] : 0.f; xh[q * 4 + j] *= gz; g[q * 4 + j] = gz; v[j] = ok ? v[j] : 0.f; }
        dyk[q * 4 + 0] = v[0]; dyk[q * 4 + 1] = v[1]; dyk[q * 4 + 2] = v[2]; dyk[q * 4 + 3] = v[3];
      }
    }
    if (pg) {
      // column sums over this wave's 8 rows (lanes row*8 + part: xor 8, 16, 32), then one LDS atomic per column and wave
      __syncthreads();                 // spg zeroed
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        float sa = xh[j], sb = g[j], sc = dyk[j];
#pragma unroll
        for (int o = 8; o < 64; o <<= 1) { sa += __shfl_xor(sa, o, 64); sb += __shfl_xor(sb, o, 64); sc += __shfl_xor(sc, o, 64); }
        if (lane < 8) { spg[0][part * 16 + j].add(sa); spg[1][part * 16 + j].add(sb); spg[2][part * 16 + j].add(sc); }
      }
    }
  }
  __syncthreads();
  BPHASE(pb, 2);
  // ---- phase 2: dU = (dY W2) * act'(U)
  {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const bf16x8 af = *reinterpret_cast<const bf16x8*>(&sdy[lr][ks * 16 + 8 * lh]);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bw2[ks], acc, 0, 0, 0);
    }
    float csum = 0.f;
    act_dispatch(a.act, [&](auto AT) __attribute__((always_inline)) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = (r & 3) + 8 * (r >> 2) + 4 * lh;
      const long row = r0 + m;
      float v = 0.f;
      if (row < a.R) {
        v = acc[r] * act_grad_c<decltype(AT)::value>(a.act, uu[r]);
        a.du[row * 128 + n] = v;
      }
      sdu[m][n] = to_bf16(v);
      csum += v;
    }
    });
    if (pg) {   // db1[n] = sum over this tile's rows of dU (rows beyond R contributed zeros)
      csum += __shfl_xor(csum, 32, 64);
      if (lh == 0) acc_add(&a.db1[n], csum);
    }
  }
  __syncthreads();
  BPHASE(pb, 3);
  if (pg && tid < 128) {   // (spg complete: every wave's LDS atomics precede the barrier behind phase 1)
    acc_add(&a.dgamma[tid], spg[0][tid].get());
    acc_add(&a.dbeta[tid], spg[1][tid].get());
    acc_add(&a.db2[tid], spg[2][tid].get());
  }
  // ---- phase 3: dX = dU W1 + dY Wr
  {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(&sdu[lr][ks * 16 + 8 * lh]);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, bw1[ks], acc, 0, 0, 0);
      const bf16x8 a2 = *reinterpret_cast<const bf16x8*>(&sdy[lr][ks * 16 + 8 * lh]);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, bwr[ks], acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const long row = r0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (row < a.R) a.dx[row * 128 + n] = acc[r];
    }
  }
  BPHASE(pb, 4);
}

__global__ void wt_transpose_kernel(WtTransposeArgs a) {
  __shared__ float t[32][33];
  const float* __restrict__ src = a.src[blockIdx.z];
  __bf16* __restrict__ dst = a.dst[blockIdx.z];
  const int k0 = blockIdx.y * 32, n0 = blockIdx.x * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int j = ty; j < 32; j += 8) t[j][tx] = src[(k0 + j) * 128 + n0 + tx];
  __syncthreads();
  for (int j = ty; j < 32; j += 8) dst[(n0 + j) * 128 + k0 + tx] = to_bf16(t[tx][j]);
}

__global__ void rowln_param_grads_kernel(const float* __restrict__ y, const float* __restrict__ mean,
                                         const float* __restrict__ rstd, const float* __restrict__ dz,
                                         float* __restrict__ dgamma, float* __restrict__ dbeta, long R, int n) {
  // thread = column j (n <= 256), rows strided over the grid
  const int j = threadIdx.x;
  if (j >= n) return;
  float sg = 0.f, sb = 0.f;
  for (long r = blockIdx.x; r < R; r += gridDim.x) {
    const float g = dz[r * n + j];
    sg += g * (y[r * n + j] - mean[r]) * rstd[r];
    sb += g;
  }
  acc_add(&dgamma[j], sg);
  acc_add(&dbeta[j], sb);
}

// All four column-sum parameter gradients of the D axis in one streaming pass over dz, y, dY, dU ([R,128] each):
//   dgamma = sum dz * xhat, dbeta = sum dz, db2 = sum dY, db1 = sum dU.
// Workgroup = 64 rows: thread = (column quad, row lane), 8 rows each with 16-byte loads; LDS reduction over the 8 row lanes,
// then 512 atomics.  (Replaces rowln_param_grads + 2 x colsum: 3 launches of 30-40 us each at R = 19,200.)
__global__ __launch_bounds__(256) void daxis_param_grads_kernel(const float* __restrict__ y, const float* __restrict__ mean,
                                                                const float* __restrict__ rstd, const float* __restrict__ dz,
                                                                const float* __restrict__ dy, const float* __restrict__ du,
                                                                float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                                float* __restrict__ db2, float* __restrict__ db1, long R) {
  __shared__ float4 red[4][8][32];
  const int cq = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const long r0 = (long)blockIdx.x * 64;
  float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0, a3 = a0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const long r = r0 + rl + 8 * i;
    const long rc = r < R ? r : R - 1;             // unconditional loads from a clamped row, zeroed afterwards
    const float m = r < R ? 1.f : 0.f;
    const float mu = mean[rc], rs = rstd[rc] * m;
    const float4 g = *reinterpret_cast<const float4*>(dz + rc * 128 + cq * 4);
    const float4 yv = *reinterpret_cast<const float4*>(y + rc * 128 + cq * 4);
    const float4 v = *reinterpret_cast<const float4*>(dy + rc * 128 + cq * 4);
    const float4 u = *reinterpret_cast<const float4*>(du + rc * 128 + cq * 4);
    a0.x += g.x * (yv.x - mu) * rs; a0.y += g.y * (yv.y - mu) * rs; a0.z += g.z * (yv.z - mu) * rs; a0.w += g.w * (yv.w - mu) * rs;
    a1.x += g.x * m; a1.y += g.y * m; a1.z += g.z * m; a1.w += g.w * m;
    a2.x += v.x * m; a2.y += v.y * m; a2.z += v.z * m; a2.w += v.w * m;
    a3.x += u.x * m; a3.y += u.y * m; a3.z += u.z * m; a3.w += u.w * m;
  }
  red[0][rl][cq] = a0; red[1][rl][cq] = a1; red[2][rl][cq] = a2; red[3][rl][cq] = a3;
  __syncthreads();
  // 512 sums of 8: thread t -> quantity t >> 6 (and + 4 ... no: 256 threads, two sums each)
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int idx = threadIdx.x + 256 * h, q = idx >> 7, c = idx & 127;
    const float* base = reinterpret_cast<const float*>(&red[q][0][0]);
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) sum += base[j * 128 + c];
    float* dst = q == 0 ? dgamma : (q == 1 ? dbeta : (q == 2 ? db2 : db1));
    if (dst) acc_add(&dst[c], sum);
  }
}

}  // namespace

int daxis_param_grads(hipStream_t s, const float* y, const float* mean, const float* rstd, const float* dz, const float* dy,
                      const float* du, float* dgamma, float* dbeta, float* db2, float* db1, long R) {
  if (R <= 0) return MIMRL_OK;
  hipLaunchKernelGGL(daxis_param_grads_kernel, dim3((unsigned)((R + 63) / 64)), dim3(256), 0, s, y, mean, rstd, dz, dy, du, dgamma, dbeta,
                     db2, db1, R);
  LAUNCH_CHECK();
  return MIMRL_OK;
}

bool daxis_bwd_supported(int id, int hd, int od) { return id == 128 && hd == 128 && od == 128; }

int daxis_bwd_fused(hipStream_t s, const DAxisBwdArgs& a) {
  if (!a.w2t || !a.w1t || !a.wrt) return set_error(MIMRL_ERR_ARG, "daxis_bwd_fused: needs the transposed weight images");
  if ((a.dgamma || a.dbeta || a.db2 || a.db1) && !(a.dgamma && a.dbeta && a.db2 && a.db1))
    return set_error(MIMRL_ERR_ARG, "daxis_bwd_fused: the four parameter-gradient outputs come together");
  hipLaunchKernelGGL(daxis_bwd_kernel, dim3((unsigned)((a.R + DR - 1) / DR)), dim3(256), 0, s, a);
  LAUNCH_CHECK();
  return MIMRL_OK;
}

int wt_transpose_bf16(hipStream_t s, const WtTransposeArgs& a) {
  if (a.n < 1 || a.n > 12) return set_error(MIMRL_ERR_ARG, "wt_transpose_bf16: 1..12 matrices");
  hipLaunchKernelGGL(wt_transpose_kernel, dim3(4, 4, a.n), dim3(256), 0, s, a);
  LAUNCH_CHECK();
  return MIMRL_OK;
}

int rowln_param_grads(hipStream_t s, const float* y, const float* mean, const float* rstd, const float* dz, float* dgamma,
                      float* dbeta, long R, int n) {
  if (n > 256) return set_error(MIMRL_ERR_ARG, "rowln_param_grads: row length %d > 256", n);
  const int grid = (int)(R < 512 ? R : 512);
  hipLaunchKernelGGL(rowln_param_grads_kernel, dim3(grid), dim3(256), 0, s, y, mean, rstd, dz, dgamma, dbeta, R, n);
  LAUNCH_CHECK();
  return MIMRL_OK;
}

}  // namespace mimrl
