// Concat critic forward, weights-stationary, ONE WAVE PER SIMD (round 6, third cut).  Same arithmetic and outputs as concat_ws.hip.
//
// What tools/hw/concat_ws_bench's synthetic variants measured on gfx950: a wave's independent vector instructions issue UNDER its own
// MFMAs (64 MFMAs + 512 FMAs per iteration in one wave: 56.7 us against 53.8 us for the MFMAs alone), but the MFMAs of one wave and the
// vector instructions of ANOTHER wave of the same SIMD do not overlap at all (53.8 us + 46.0 us separately, 96.3 us together).  The
// 8-wave kernel of concat_ws.hip (a layer-1 and a layer-2 wave per SIMD) therefore pays products + epilogues in sequence however its roles
// are staggered (122 us at cfg3's shape where its products alone take 64).
//
// Here a workgroup is 4 waves, one per SIMD, 512 registers each: wave w holds the 64-feature slice [64 w, 64 w + 64) of BOTH hidden layers
// (2 x 128 registers of MFMA A fragments) and runs, per unit of 32 pair rows, two blocks of straight-line code in which the vector work
// rides under the wave's own matrix work:
//   block 1:  layer-1 product of unit t  (32 MFMAs)   ||  layer-2 epilogue of unit t - 2 (sign words, score head, stage 1: fp32 a2)
//   block 2:  layer-2 product of unit t - 1 (32 MFMAs) ||  layer-1 epilogue of unit t -> act1, layer 0 of unit t + 1 -> act0
// one barrier per unit (4 waves).  Out-of-range units at the ends of a run execute the same arithmetic on whatever the tiles hold (so that
// the blocks stay free of branches) and only their global stores are predicated.
#include "../../mimrl_amd/csrc/concat_ws_dev.h"

namespace mimrl {

namespace {

// where the stores of out-of-range units go (selected with a scalar pointer select: no branch inside the straight-line blocks); 128 KB covers the
// largest footprint of one unit's stores (fp32 a2: 32 rows x 1 KB) from any lane offset used below
__device__ uint32_t g_ws4_sink[32768];
template <class T> __device__ __forceinline__ T* sel(bool real, T* p) { return real ? p : reinterpret_cast<T*>(g_ws4_sink); }

// Program order of a block: behind every MFMA one LDS read and N vector instructions (sched_group_barrier: 0x008 MFMA, 0x100 DS read, 0x002
// VALU).  Left to itself the scheduler issues a block's 32 MFMAs first and the epilogue behind them -- and a wave issues in order, so its
// vector instructions then start when its last MFMA has been ISSUED, i.e. the overlap is lost.
#ifndef WS4_VALU_PER_MFMA
#define WS4_VALU_PER_MFMA 5
#endif
template <int N> __device__ __forceinline__ void interleave() {
#if WS4_VALU_PER_MFMA > 0
#pragma unroll
  for (int i = 0; i < 32; ++i) {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x002, N, 0);
  }
#endif
}

template <int SAVE>
__global__ __launch_bounds__(256) void concat_fwd_ws4_kernel(ConcatFwdArgs a, int units_e, int total, int per) {
  __shared__ __attribute__((aligned(16))) __bf16 act0[2][UR][AP];   // layer-0 outputs (operand of layer 1), double-buffered over units
  __shared__ __attribute__((aligned(16))) __bf16 act1[2][UR][AP];   // layer-1 outputs (operand of layer 2)
  __shared__ __attribute__((aligned(16))) float sbias[2][CH];       // [layer][feature]: each wave writes and reads only its own 64-feature slice
  __shared__ __attribute__((aligned(16))) float sw3[CH];            // score-head weight (own slice)
  __shared__ LdsAcc sc[2][UR];                                      // score sums of a unit over the four waves
  const int tid = threadIdx.x, lane = tid & 63, lr = lane & 31, lh = lane >> 5;
  const int ws = __builtin_amdgcn_readfirstlane(tid >> 6);          // wave = 64-feature slice (uniform)
  const int B = a.B, BB = B * B;
  const int u0 = blockIdx.x * per, U = min(total, u0 + per) - u0;
  if (U <= 0) return;
  if (tid < 2 * UR) sc[tid >> 5][tid & 31].zero();
  bf16x8 w1[2][16], w2[2][16];                     // this wave's slices: [feature tile][k-step], A fragments of v_mfma_f32_32x32x16_bf16
#pragma unroll
  for (int ct = 0; ct < 2; ++ct)
#pragma unroll
    for (int ks = 0; ks < 16; ++ks)
#pragma unroll
      for (int i = 0; i < 8; ++i) { w1[ct][ks][i] = (__bf16)0.f; w2[ct][ks][i] = (__bf16)0.f; }
  int e1 = -1, e2 = -1;
  float b3e = 0.f;
  const unsigned c4 = lane * 4;                    // layer-0 generation: lane = column quad, wave w = rows w + 4 q (q = 0 .. 7) of the unit
  const unsigned nib_sh = 4 * (lane & 7);
  const unsigned lane8 = lane >> 3, lr8 = lr * 8;
  const unsigned row_k = lr * CH + 8 * lh;
  const unsigned row_f = lr * CH + 4 * lh;
  const unsigned sh_lo = 4 * lh, sh_hi = 16 + 4 * lh;
  uint32_t ones = 0x00010001u;
  asm volatile("" : "+v"(ones));
  float4 xq = make_float4(0.f, 0.f, 0.f, 0.f), yq[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) yq[q] = xq;
  f32x16 acc1[2], acc2[2];
#pragma unroll
  for (int ct = 0; ct < 2; ++ct)
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc1[ct][r] = 0.f; acc2[ct][r] = 0.f; }
  // unit cursors (clamped to the run: an out-of-range unit repeats the last valid one, its stores are switched off)
  UnitPos G1;                                      // unit t + 1 (layer-0 generation)
  G1.e = u0 / units_e; G1.row0 = (u0 - G1.e * units_e) * UR; G1.base = G1.e * BB + G1.row0;
  G1.gi = G1.row0 / B; G1.gj0 = G1.row0 - G1.gi * B;
  UnitRef R0 = {G1.e, G1.base}, R1 = R0, R2 = R0, R3 = R0;          // units t, t - 1, t - 2, t - 3

  auto issue_gen_loads = [&](const UnitPos& u) __attribute__((always_inline)) {
    const GLOBAL_AS float* Pp = uptr(a.P + ((long)u.e * B + u.gi) * CH);
    const GLOBAL_AS float* Qp = uptr(a.Q + ((long)u.e * B + u.gj0 + ws) * CH);
    { const f32x4v t_ = *(const GLOBAL_AS f32x4v*)(Pp + c4); xq = make_float4(t_[0], t_[1], t_[2], t_[3]); }
#pragma unroll
    for (int q = 0; q < 8; ++q) { const f32x4v t_ = *(const GLOBAL_AS f32x4v*)(Qp + (long)(4 * q) * CH + c4); yq[q] = make_float4(t_[0], t_[1], t_[2], t_[3]); }
  };
  auto load_slice = [&](bf16x8 (&wf)[2][16], const __bf16* Wimg, int e) __attribute__((always_inline)) {
    const GLOBAL_AS __bf16* W = uptr(Wimg + (long)e * a.pstride + (long)(ws * 64) * CH);
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) wf[ct][ks] = *(const GLOBAL_AS bf16x8*)(W + (long)(ct * 32) * CH + ks * 16 + row_k);
  };
  auto bias_init = [&](f32x16 (&acc)[2], int layer) __attribute__((always_inline)) {
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 bb = *reinterpret_cast<const float4*>(&sbias[layer][ws * 64 + ct * 32 + 8 * q + 4 * lh]);
        acc[ct][4 * q] = bb.x; acc[ct][4 * q + 1] = bb.y; acc[ct][4 * q + 2] = bb.z; acc[ct][4 * q + 3] = bb.w;
      }
  };
  // product with the vector work of another unit riding under it IN PROGRAM ORDER: behind the two MFMAs of k-step ks the caller's `side(ks)`
  // is emitted (a wave issues in order -- the scheduler left to itself puts the 32 MFMAs first and the epilogue behind them, and
  // sched_group_barrier pipelines did not move it)
  auto product = [&](f32x16 (&acc)[2], const bf16x8 (&wf)[2][16], const __bf16 (*src)[AP], auto&& side) __attribute__((always_inline)) {
    bf16x8 fr[16];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) fr[ks] = *reinterpret_cast<const bf16x8*>(&src[lr][ks * 16 + 8 * lh]);
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      if (ks + 4 < 16) fr[ks + 4] = *reinterpret_cast<const bf16x8*>(&src[lr][(ks + 4) * 16 + 8 * lh]);
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[0][ks], fr[ks], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[1][ks], fr[ks], acc[1], 0, 0, 0);
      side(ks);
    }
  };
  typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
  // layer 1 epilogue of unit `u`, quad by quad: micro-step i (0 .. 7) = quad (i & 3) of tile (i >> 2): ReLU -> bf16 operand tile of layer 2
  // (8-byte LDS store) + its sign bits; the tile's sign word goes out behind its last quad
  uint32_t e1_lo = 0u, e1_hi = 0u;
  auto e1_step = [&](int i, const UnitRef& u, int buf, bool store) __attribute__((always_inline)) {
    const int ct = i >> 2, q = i & 3;
    const uint32_t p0 = relu_pack2(acc1[ct][4 * q], acc1[ct][4 * q + 1]), p1 = relu_pack2(acc1[ct][4 * q + 2], acc1[ct][4 * q + 3]);
    u32x2 b; b[0] = p0; b[1] = p1;
    *reinterpret_cast<u32x2*>(&act1[buf][lr][ws * 64 + ct * 32 + 8 * q + 4 * lh]) = b;
    if (SAVE >= 2) {
      if (q == 0) { e1_lo = sign_pair<2>(p1, sign_pair<0>(p0, 0u, ones), ones); }
      if (q == 1) { e1_lo = sign_pair<10>(p1, sign_pair<8>(p0, e1_lo, ones), ones); }
      if (q == 2) { e1_hi = sign_pair<2>(p1, sign_pair<0>(p0, 0u, ones), ones); }
      if (q == 3) {
        e1_hi = sign_pair<10>(p1, sign_pair<8>(p0, e1_hi, ones), ones);
        const uint32_t bits = or_halves((e1_lo << sh_lo) | (e1_hi << sh_hi));
        GLOBAL_AS uint32_t* o = uptr(sel(store, a.m1 + (long)u.base * 8 + ws * 2 + ct)); o[lr8] = bits;
      }
    }
  };
  // layer 2 epilogue of unit `u`, quad by quad (micro-step i = 0 .. 7), then the finish (i == 8): score head sum over the wave's features
  uint32_t e2_lo = 0u, e2_hi = 0u;
  float e2_hp = 0.f;
  auto e2_step = [&](int i, const UnitRef& u, int slot, bool store) __attribute__((always_inline)) {
    if (i == 8) {
      const float hp = add_halves(e2_hp) + (ws == 0 ? b3e : 0.f);
      sc[slot][lr].add(0.5f * hp);                          // (both halves hold the sum: each adds half of it -- no lane branch inside the block)
      e2_hp = 0.f;
      return;
    }
    const int ct = i >> 2, q = i & 3;
    const int f0 = ws * 64 + ct * 32 + 8 * q + 4 * lh;
    float4 v;
    v.x = relu1(acc2[ct][4 * q]); v.y = relu1(acc2[ct][4 * q + 1]); v.z = relu1(acc2[ct][4 * q + 2]); v.w = relu1(acc2[ct][4 * q + 3]);
    const float4 w3v = *reinterpret_cast<const float4*>(&sw3[f0]);
    e2_hp += v.x * w3v.x + v.y * w3v.y + v.z * w3v.z + v.w * w3v.w;
    if (SAVE == 2) { GLOBAL_AS float* o = uptr(sel(store, a.a2 + (long)u.base * CH + ws * 64 + ct * 32 + 8 * q)); f32x4v t_ = {v.x, v.y, v.z, v.w}; *(GLOBAL_AS f32x4v*)(o + row_f) = t_; }
    if (SAVE >= 2) {
      const uint32_t p0 = relu_pack2(v.x, v.y), p1 = relu_pack2(v.z, v.w);
      if (q == 0) { e2_lo = sign_pair<2>(p1, sign_pair<0>(p0, 0u, ones), ones); }
      if (q == 1) { e2_lo = sign_pair<10>(p1, sign_pair<8>(p0, e2_lo, ones), ones); }
      if (q == 2) { e2_hi = sign_pair<2>(p1, sign_pair<0>(p0, 0u, ones), ones); }
      if (q == 3) {
        e2_hi = sign_pair<10>(p1, sign_pair<8>(p0, e2_hi, ones), ones);
        const uint32_t bits = or_halves((e2_lo << sh_lo) | (e2_hi << sh_hi));
        GLOBAL_AS uint32_t* o = uptr(sel(store, a.m2 + (long)u.base * 8 + ws * 2 + ct)); o[lr8] = bits;
      }
    }
  };
  // layer 0 of unit `u`, row quad q (0 .. 7) of this thread
  auto gen_step = [&](int q, const UnitPos& u, int buf, bool store) __attribute__((always_inline)) {
    const int row = ws + 4 * q;
    u32x2 b;
    b[0] = relu_pack2(xq.x + yq[q].x, xq.y + yq[q].y);
    b[1] = relu_pack2(xq.z + yq[q].z, xq.w + yq[q].w);
    *reinterpret_cast<u32x2*>(&act0[buf][row][c4]) = b;
    if (SAVE == 2) { GLOBAL_AS __bf16* o = uptr(sel(store, a.a0b + ((long)u.base + row) * CH)); *(GLOBAL_AS u32x2*)(o + c4) = b; }
    if (SAVE >= 2) {
      uint32_t w = sign_pair<2>(b[1], sign_pair<0>(b[0], 0u, ones), ones) << nib_sh;
      w |= dpp<0xB1>(w);
      w |= dpp<0x4E>(w);
      w |= dpp<0x141>(w);
      GLOBAL_AS uint32_t* o = uptr(sel(store, a.m0 + ((long)u.base + row) * 8)); o[lane8] = w;
    }
  };

  // ---- prologue: layer 0 of unit 0
  issue_gen_loads(G1);
#pragma unroll
  for (int q = 0; q < 8; ++q) gen_step(q, G1, 0, true);
  const UnitRef first = {G1.e, G1.base};
  (void)first;
  if (U > 1) G1.advance(B, BB);
  __syncthreads();
#pragma unroll 1
  for (int t = 0; t <= U + 2; ++t) {
    const bool v_g = t + 1 < U, v_0 = t < U, v_1 = t >= 1 && t - 1 < U, v_2 = t >= 2 && t - 2 < U, v_3 = t >= 3 && t - 3 < U;
    // layer-1 weights / bias of the estimator of unit t: re-loaded only when the run crosses into the next estimator
    if (R0.e != e1) {
      e1 = R0.e;
      load_slice(w1, a.W1, e1);
      sbias[0][ws * 64 + lane] = (a.b1 + (long)e1 * a.pstride)[ws * 64 + lane];
      __builtin_amdgcn_wave_barrier();
    }
    // ---- block 1: layer-1 product of unit t  ||  layer-2 epilogue of unit t - 2 (accumulators of the last iteration's block 2)
    issue_gen_loads(G1);                                 // (unit t + 1; the cursor is clamped to the run: always a valid address)
    bias_init(acc1, 0);
    product(acc1, w1, act0[t & 1], [&](int ks) __attribute__((always_inline)) {
      if ((ks & 1) == 1) e2_step(ks >> 1, R2, t & 1, v_2);          // one quad of the layer-2 epilogue behind every second k-step
    });
    e2_step(8, R2, t & 1, v_2);
    // layer-2 weights / bias / score head of the estimator of unit t - 1 (behind the epilogue of unit t - 2, which may belong to the last one)
    if (R1.e != e2) {
      e2 = R1.e;
      load_slice(w2, a.W2, e2);
      sbias[1][ws * 64 + lane] = (a.b2 + (long)e2 * a.pstride)[ws * 64 + lane];
      sw3[ws * 64 + lane] = a.w3[(long)e2 * a.pstride + ws * 64 + lane];
      b3e = a.b3[(long)e2 * a.pstride];
      __builtin_amdgcn_wave_barrier();
    }
    // ---- block 2: layer-2 product of unit t - 1  ||  layer-1 epilogue of unit t -> act1[t & 1], layer 0 of unit t + 1 -> act0[(t + 1) & 1]
    bias_init(acc2, 1);
    product(acc2, w2, act1[(t - 1) & 1], [&](int ks) __attribute__((always_inline)) {
      if (ks < 8) e1_step(ks, R0, t & 1, v_0);                       // k-steps 0 .. 7: the eight quads of the layer-1 epilogue ...
      else gen_step(ks - 8, G1, (t + 1) & 1, v_g);                   // ... 8 .. 15: the thread's eight row quads of the next unit's layer 0
    });
    // ---- stage 1: the finished layer-1 tile of unit t - 1 leaves as bf16 in whole 512-byte rows
    if (SAVE == 2 && v_1) {
      GLOBAL_AS __bf16* o = uptr(a.a1b + (long)R1.base * CH);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const unsigned idx = tid + 256 * q, row = idx >> 5, c8 = (idx & 31) * 8;
        *(GLOBAL_AS u32x4*)(o + (row * CH + c8)) = *reinterpret_cast<const u32x4*>(&act1[(t - 1) & 1][row][c8]);
      }
    }
    // ---- scores of unit t - 3 (its four waves added their parts during iteration t - 1: a barrier has passed since)
    if (ws == 0) {
      const float v = sc[(t - 3) & 1][lr].get();
      sc[(t - 3) & 1][lr].zero();
      if (v_3) { GLOBAL_AS float* o = uptr(a.scores + R3.base); o[(unsigned)lr] = v; }
    }
    R3 = R2; R2 = R1; R1 = R0; R0 = UnitRef{G1.e, G1.base};
    if (t + 2 < U) G1.advance(B, BB);
    __syncthreads();
  }
}

}  // namespace

int concat_fwd_ws4(hipStream_t s, const ConcatFwdArgs& a) {
  if (!concat_fwd_ws_supported(a.B, CH, a.save)) return set_error(MIMRL_ERR_ARG, "concat_fwd_ws4: batch %d / save %d unsupported", a.B, a.save);
  const int units_e = (int)(((long)a.B * a.B) / UR), total = a.E * units_e;
  const int nwg0 = std::min(device_cus(), total), per = (total + nwg0 - 1) / nwg0, nwg = (total + per - 1) / per;
  const dim3 grid((unsigned)nwg);
  if (a.save == 0) hipLaunchKernelGGL(concat_fwd_ws4_kernel<0>, grid, dim3(256), 0, s, a, units_e, total, per);
  else if (a.save == 2) hipLaunchKernelGGL(concat_fwd_ws4_kernel<2>, grid, dim3(256), 0, s, a, units_e, total, per);
  else hipLaunchKernelGGL(concat_fwd_ws4_kernel<3>, grid, dim3(256), 0, s, a, units_e, total, per);
  LAUNCH_CHECK();
  return MIMRL_OK;
}

}  // namespace mimrl
