// Concat critic backward data-gradient chain, WEIGHTS-STATIONARY (round 6): the mirror of concat_ws.hip for
//   dZ2 = ds w3^T (.) [a2 > 0]  ->  dZ1 = (dZ2 W2) (.) [a1 > 0]  ->  dZ0 = (dZ1 W1) (.) [a0 > 0],   dP[i] = sum_j dZ0[i, j],  dQ[j] = sum_i dZ0[i, j]
// (autograd of VMI.py:58-65 through the Linear / ReLU stack of VMI.py:13-22).  One persistent workgroup per CU, 8 waves:
//   * waves 0-3 own the W2 product, waves 4-7 the W1 product; wave w of a product keeps its 64-column slice of the TRANSPOSED bf16 weight
//     image as MFMA fragments in registers for the whole launch (128 VGPRs);
//   * units of 32 pair rows in (estimator, y block of 32 rows, x row i) order -- i fastest, so that a wave of the W1 product keeps
//     dQ = sum_i dZ0 of its (estimator, y block) in registers (32 per lane) and writes it once per block to a slot of `dq_part`
//     (concat_ws_reduce_kernel adds a block's slots in workgroup order: run-to-run reproducible);
//   * the W2 product is TRANSPOSED (weights = A operand, gradient tile = B operand: a lane owns one pair row, its sign words are its own
//     two 32-bit loads, the dZ1 tile for the next product is written with 8-byte LDS stores); the W1 product is not (a lane owns one
//     feature: the column sums over the unit's rows -- dP -- and the dQ accumulators are plain per-lane adds; its sign words come from an
//     LDS copy of the unit's words transposed to [32-column group][row]);
//   * dP[i] is NOT accumulated with atomics: the unit's column sums go to dp_part[estimator][y block][i] with plain stores and the reduce
//     kernel adds the B / 32 y blocks;
//   * pipeline and barriers as in concat_ws.hip: iteration t = W2 product of unit t | W1 product of unit t - 2 | dZ2 generation of unit
//     t + 2, tiles in rings of four, one barrier per two iterations; no store sits behind a lane branch.
// Stage 1 (WG) additionally writes dZ2 / dZ1 as bf16 (operands of the weight-gradient GEMMs) and the bias gradients db1 = sum dZ1,
// db2 = sum dZ2, db3 = sum ds, accumulated in registers over the run and added to the bucket when the run leaves an estimator.
#include "concat_ws_dev.h"

namespace mimrl {

namespace {

// unit cursor in (estimator, y block, x row) order
struct BUnit {
  int e, yb, i;      // estimator, y block of 32 rows, x row
  int base;          // e * B*B + i * B + yb * 32: first pair row of the unit in the [E][B*B] row space
  __device__ __forceinline__ void advance(int B, int BB, int nyb) {
    ++i; base += B;
    if (i == B) { i = 0; ++yb; if (yb == nyb) { yb = 0; ++e; } base = e * BB + yb * UR; }
  }
};
struct BRef { int e, yb, i, base; };
__device__ __forceinline__ BRef ref(const BUnit& u) { return BRef{u.e, u.yb, u.i, u.base}; }

// v if bit k of w is set, else 0 (v_bfe_i32: the bit sign-extended to a 0 / -1 mask; v_and)
__device__ __forceinline__ float keep_if(float v, uint32_t w, int k) {
  const int m = __builtin_amdgcn_sbfe((int)w, k, 1);
  return __uint_as_float(__float_as_uint(v) & (uint32_t)m);
}
__device__ __forceinline__ uint32_t pack2(float x, float y) {
  typedef __bf16 bf16x2_ __attribute__((ext_vector_type(2)));
  typedef float f32x2_ __attribute__((ext_vector_type(2)));
  const f32x2_ f = {x, y};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, bf16x2_));
}

// DZ2 (stage 1): dZ2 is written out (false: its consumer regenerates it from ds, w3 and the layer-2 sign words -- concat_dw.hip)
template <bool WG, bool DZ2 = WG>
__global__ __launch_bounds__(512) void concat_bwd_ws_kernel(ConcatBwdArgs a, int total, int per, int slots, float* __restrict__ dp_part) {
  __shared__ __attribute__((aligned(16))) __bf16 g2[4][UR][AP];     // dZ2 tiles (operand of the W2 product), ring over units
  __shared__ __attribute__((aligned(16))) __bf16 g1[4][UR][AP];     // dZ1 tiles (operand of the W1 product)
  __shared__ __attribute__((aligned(16))) float sw3[8][CH];         // score-head weight of the generation phase's estimator, one private copy per wave
  __shared__ __attribute__((aligned(16))) uint32_t m0T[8][8][UR];   // layer-0 sign words of a unit, [32-column group][row]; ring of 8 (used 5 iterations on)
  const int tid = threadIdx.x, lane = tid & 63, lr = lane & 31, lh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int role = wave >> 2, ws = wave & 3;       // role 0: W2 product (transposed), role 1: W1 product; ws: 64-column slice
  const int B = a.B, BB = B * B, nyb = B / UR, units_e = nyb * B;
  const int u0 = blockIdx.x * per, U = min(total, u0 + per) - u0;
  if (U <= 0) return;
  bf16x8 wf[2][16];
#pragma unroll
  for (int ct = 0; ct < 2; ++ct)
#pragma unroll
    for (int ks = 0; ks < 16; ++ks)
#pragma unroll
      for (int i = 0; i < 8; ++i) wf[ct][ks][i] = (__bf16)0.f;
  int my_e = -1, gen_e = -1;
  const unsigned c4 = lane * 4, nib_sh = 4 * (lane & 7), lane8 = lane >> 3;
  const unsigned row_k = lr * CH + 8 * lh;
  f32x16 acc[2];
#pragma unroll
  for (int ct = 0; ct < 2; ++ct)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[ct][r] = 0.f;
  // role 1: dQ partial sums of the current (estimator, y block); role 0, stage 1: column sums of dZ1 (db1) over the run's units of an estimator
  float keep[2][16];
#pragma unroll
  for (int ct = 0; ct < 2; ++ct)
#pragma unroll
    for (int r = 0; r < 16; ++r) keep[ct][r] = 0.f;
  // dZ2 generation state: this estimator's score-head weight quad, stage 1: db2 / db3 partial sums
  float4 sdb = make_float4(0.f, 0.f, 0.f, 0.f);
  float sds = 0.f;
  float dsv[4] = {0.f, 0.f, 0.f, 0.f};
  uint32_t m2w[4] = {0u, 0u, 0u, 0u}, m0w = 0u;
  typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
  u32x2 m1w = {0u, 0u};

  BUnit G2, G3;                                     // units t + 2 / t + 3
  {
    const int e = u0 / units_e, rem = u0 - e * units_e, yb = rem / B, i = rem - yb * B;
    G2.e = e; G2.yb = yb; G2.i = i; G2.base = e * BB + i * B + yb * UR;
  }
  G3 = G2; G3.advance(B, BB, nyb);
  BRef R1m = ref(G2), R0 = R1m, R1 = R1m, R2 = R1m, R3 = R1m;

  auto issue_gen_loads = [&](const BUnit& u) __attribute__((always_inline)) {
    const GLOBAL_AS float* dsp = uptr(a.ds + (long)u.base + wave);
    const GLOBAL_AS uint32_t* mp = uptr(a.m2 + ((long)u.base + wave) * 8);
#pragma unroll
    for (int q = 0; q < 4; ++q) { dsv[q] = dsp[8 * q]; m2w[q] = mp[64 * q + lane8]; }   // (dsv: one value per wave and q -- moved to scalar registers where it is used)
    if (wave < 4) { const GLOBAL_AS uint32_t* m0p = uptr(a.m0 + (long)u.base * 8); m0w = m0p[(unsigned)tid]; }   // word (row tid >> 3, group tid & 7)
  };
  // stage 1: the bias-gradient sums of the generation phase (db2, db3) leave when the run leaves estimator `e`
  auto flush_gen_sums = [&](int e) __attribute__((always_inline)) {
    float* db2 = a.db2 + (long)e * a.pstride + c4;
    acc_add(db2, sdb.x); acc_add(db2 + 1, sdb.y); acc_add(db2 + 2, sdb.z); acc_add(db2 + 3, sdb.w);
    if (lane == 0) acc_add(a.db3 + (long)e * a.pstride, sds);
    sdb = make_float4(0.f, 0.f, 0.f, 0.f); sds = 0.f;
  };
  // dZ2 of unit u -> g2[buf] (+ stage 1: its bf16 copy and the db2 / db3 sums), the unit's layer-0 sign words -> m0T[slot8] transposed
  auto gen_finish = [&](const BUnit& u, int buf, int slot8) __attribute__((always_inline)) {
    if (u.e != gen_e) {                                  // (wave-uniform, once per estimator of the run)
      if (WG && gen_e >= 0) flush_gen_sums(gen_e);
      gen_e = u.e;
      const GLOBAL_AS float* wp = uptr(a.w3 + (long)u.e * a.pstride);
      const f32x4v t_ = *(const GLOBAL_AS f32x4v*)(wp + c4);
      *reinterpret_cast<float4*>(&sw3[wave][c4]) = make_float4(t_[0], t_[1], t_[2], t_[3]);
      __builtin_amdgcn_wave_barrier();
    }
    const float4 w3q = *reinterpret_cast<const float4*>(&sw3[wave][c4]);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int row = wave + 8 * q;
      const float d = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(dsv[q])));
      const uint32_t bits = m2w[q] >> nib_sh;
      const float vx = keep_if(d * w3q.x, bits, 0), vy = keep_if(d * w3q.y, bits, 1), vz = keep_if(d * w3q.z, bits, 2), vw = keep_if(d * w3q.w, bits, 3);
      u32x2 b; b[0] = pack2(vx, vy); b[1] = pack2(vz, vw);
      *reinterpret_cast<u32x2*>(&g2[buf][row][c4]) = b;
      if (WG) {
        if constexpr (DZ2) { GLOBAL_AS __bf16* o = uptr(a.dz2 + ((long)u.base + row) * CH); *(GLOBAL_AS u32x2*)(o + c4) = b; }
        sdb.x += vx; sdb.y += vy; sdb.z += vz; sdb.w += vw;
        sds += d;                                        // (every lane holds d: lane 0's sum is the one flushed)
      }
    }
    if (wave < 4) m0T[slot8][tid & 7][tid >> 3] = m0w;
  };
  auto ensure_weights = [&](int e) __attribute__((always_inline)) {
    if (e != my_e) {
      my_e = e;
      const GLOBAL_AS __bf16* W = uptr((role ? a.W1T : a.W2T) + (long)e * a.pstride + (long)(ws * 64) * CH);
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) wf[ct][ks] = *(const GLOBAL_AS bf16x8*)(W + (long)(ct * 32) * CH + ks * 16 + row_k);
    }
  };
  // acc = product of this wave's weight slice with the tile `src`: role 0 transposed (weights = A operand), role 1 not (tile = A operand)
  auto product = [&](const __bf16 (*src)[AP], auto role_c) __attribute__((always_inline)) {
    constexpr int ROLE = decltype(role_c)::value;
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ct][r] = 0.f;
    if (ROLE == 0) {
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) {
        const bf16x8 fr = *reinterpret_cast<const bf16x8*>(&src[lr][ks * 16 + 8 * lh]);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[0][ks], fr, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[1][ks], fr, acc[1], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) {
        const bf16x8 fr = *reinterpret_cast<const bf16x8*>(&src[lr][ks * 16 + 8 * lh]);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr, wf[0][ks], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr, wf[1][ks], acc[1], 0, 0, 0);
      }
    }
  };
  // role 0 (transposed: lane = pair row lr, accumulator quad q of tile ct = columns 64 ws + 32 ct + 8 q + 4 lh + (0..3)):
  // dZ1 = acc (.) [a1 > 0] -> bf16 tile g1[buf] (8-byte LDS stores); stage 1: the column sums (db1) accumulate per lane over the run
  auto epilogue_w2 = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      const uint32_t wsh = m1w[ct] >> (4 * lh);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float v0 = keep_if(acc[ct][4 * q], wsh, 8 * q), v1 = keep_if(acc[ct][4 * q + 1], wsh, 8 * q + 1),
                    v2 = keep_if(acc[ct][4 * q + 2], wsh, 8 * q + 2), v3 = keep_if(acc[ct][4 * q + 3], wsh, 8 * q + 3);
        u32x2 b; b[0] = pack2(v0, v1); b[1] = pack2(v2, v3);
        *reinterpret_cast<u32x2*>(&g1[buf][lr][ws * 64 + ct * 32 + 8 * q + 4 * lh]) = b;
        if (WG) { keep[ct][4 * q] += v0; keep[ct][4 * q + 1] += v1; keep[ct][4 * q + 2] += v2; keep[ct][4 * q + 3] += v3; }
      }
    }
  };
  // stage 1, role 0: db1 of estimator e += the per-lane column sums, reduced over the 32 rows (lanes) of each half -- once per estimator of the run
  auto flush_db1 = [&](int e) __attribute__((always_inline)) {
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float x = keep[ct][r];
        x += __shfl_xor(x, 1); x += __shfl_xor(x, 2); x += __shfl_xor(x, 4); x += __shfl_xor(x, 8); x += __shfl_xor(x, 16);
        if (lr == 0) acc_add(a.db1 + (long)e * a.pstride + ws * 64 + ct * 32 + 8 * (r >> 2) + 4 * lh + (r & 3), x);
        keep[ct][r] = 0.f;
      }
  };
  // role 1 (lane = column 64 ws + 32 ct + lr, accumulator r = row (r & 3) + 8 (r >> 2) + 4 lh): dZ0 = acc (.) [a0 > 0];
  // dQ partial sums per lane, the unit's column sums -> dp_part (plain stores)
  auto epilogue_w1 = [&](const BRef& u, int slot8) __attribute__((always_inline)) {
    float* dpo = dp_part + (((long)u.e * nyb + u.yb) * B + u.i) * CH + ws * 64;
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      float csum = 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const u32x4 w4 = *reinterpret_cast<const u32x4*>(&m0T[slot8][ws * 2 + ct][8 * q + 4 * lh]);   // the words of rows 8 q + 4 lh + (0..3)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int m = __builtin_amdgcn_sbfe((int)w4[i], lr, 1);
          const float v = __uint_as_float(__float_as_uint(acc[ct][4 * q + i]) & (uint32_t)m);
          keep[ct][4 * q + i] += v;
          csum += v;
        }
      }
      csum = add_halves(csum);
      { GLOBAL_AS float* o = uptr(dpo + ct * 32); o[(unsigned)lr] = csum; }   // (both halves hold the sum: same address, same value)
    }
  };
  // role 1: the dQ partial sums of block `blk` leave to this workgroup's slot (the run leaves the block, or ends)
  auto flush_dq = [&](int blk) __attribute__((always_inline)) {
    const int wfirst = (blk * B) / per;
    float* slot = a.dq_part + ((long)blk * slots + ((int)blockIdx.x - wfirst)) * (UR * CH) + ws * 64;
    GLOBAL_AS float* o = uptr(slot);
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        o[(unsigned)(((r & 3) + 8 * (r >> 2) + 4 * lh) * CH + ct * 32 + lr)] = keep[ct][r];
        keep[ct][r] = 0.f;
      }
  };

  // The two roles run SEPARATE copies of the iteration loop (the role is a compile-time constant inside: with one loop and run-time role
  // branches the register allocator saw the union of both roles' live values and spilled 16 weight registers into the product's MFMA chain).
  // Both copies execute the same number of barriers.
  auto run = [&](auto role_c) __attribute__((always_inline)) {
  constexpr int ROLE = decltype(role_c)::value;
  if (ROLE == 1) issue_gen_loads(G2);
#pragma unroll 1
  for (int t = -2; t <= U + 3; ++t) {
    const bool gen_on = t + 2 < U;
    // phase A
    if (ROLE == 0) {
      if (gen_on) issue_gen_loads(G2);
      if (t >= 0 && t < U) { const GLOBAL_AS uint32_t* mp = uptr(a.m1 + (long)R0.base * 8 + ws * 2); m1w = *(const GLOBAL_AS u32x2*)(mp + lr * 8); }
    } else {
      if (t >= 3 && t - 3 < U) {                         // epilogue of the product at the end of the last iteration (unit t - 3)
        epilogue_w1(R3, (t - 3) & 7);
        const int lin = u0 + t - 3;
        if (t - 3 == U - 1 || (lin + 1) % B == 0) flush_dq(R3.e * nyb + R3.yb);
      }
      if (gen_on) gen_finish(G2, (t + 2) & 3, (t + 2) & 7);
      if (t + 3 < U) issue_gen_loads(G3);
    }
    // phase B: W2 product on unit t / W1 product on unit t - 2
    {
      const int pu = t - 2 * ROLE;
      if (pu >= 0 && pu < U) {
        const int e = ROLE ? R2.e : R0.e;
        if (WG && ROLE == 0 && my_e >= 0 && e != my_e) flush_db1(my_e);
        ensure_weights(e);
        product(ROLE ? g1[(t - 2) & 3] : g2[t & 3], role_c);
      }
    }
    // phase C
    if (ROLE == 0) {
      if (t >= 0 && t < U) epilogue_w2(t & 3);
      if (gen_on) gen_finish(G2, (t + 2) & 3, (t + 2) & 7);
    }
    // stage 1: the finished dZ1 tile of unit t - 2 leaves as bf16 in whole 512-byte rows (operand of the dW1 product)
    if (WG && t >= 2 && t - 2 < U) {
      GLOBAL_AS __bf16* o = uptr(a.dz1 + (long)R2.base * CH);
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const unsigned idx = tid + 512 * q, row = idx >> 5, c8 = (idx & 31) * 8;
        *(GLOBAL_AS u32x4*)(o + (row * CH + c8)) = *reinterpret_cast<const u32x4*>(&g1[(t - 2) & 3][row][c8]);
      }
    }
    R3 = R2; R2 = R1; R1 = R0; R0 = R1m; R1m = ref(G2);
    G2 = G3;
    G3.advance(B, BB, nyb);
    if (t & 1) __syncthreads();
  }
  };
  if (role == 0) run(std::integral_constant<int, 0>{}); else run(std::integral_constant<int, 1>{});
  if (WG) {
    if (gen_e >= 0) flush_gen_sums(gen_e);
    if (role == 0 && my_e >= 0) flush_db1(my_e);
  }
}

// dQ[block][32][256] = sum of the block's slots in workgroup order; dP[e][i][256] = sum over the y blocks.  256 threads x float4 = 4 rows of 256.
// All the terms of an output are requested before the first add (groups of 8 clamped loads: the first version walked its 8-9 terms as a
// chain of dependent round trips -- 20 us per launch at cfg3 for 21 MB).
__global__ __launch_bounds__(256) void concat_ws_reduce_kernel(const float* __restrict__ dq_part, float* __restrict__ dQ, const float* __restrict__ dp_part,
                                                               float* __restrict__ dP, int E, int B, int per, int slots) {
  const int nyb = B / UR, nblk = E * nyb, nq = nblk * 8;
  const int off = threadIdx.x * 4;                       // 0 .. 1020: 4 rows x 256 columns
  const float* __restrict__ p;
  float* __restrict__ o;
  long stride;
  int n;
  if ((int)blockIdx.x < nq) {
    const int blk = blockIdx.x >> 3, part = blockIdx.x & 7;
    const int wfirst = (blk * B) / per, wlast = ((blk + 1) * B - 1) / per;
    n = wlast - wfirst + 1;
    p = dq_part + (long)blk * slots * (UR * CH) + part * (4 * CH) + off;
    stride = UR * CH;
    o = dQ + (long)blk * (UR * CH) + part * (4 * CH) + off;   // block = (estimator e, y block yb): rows yb * 32 .. of dQ[e]
  } else {
    const long r = (long)((int)blockIdx.x - nq) * 4 + (off >> 8);   // row (e, i) of dP
    if (r >= (long)E * B) return;
    const int e = (int)(r / B), i = (int)(r - (long)e * B), c = off & 255;
    n = nyb;
    p = dp_part + (((long)e * nyb) * B + i) * CH + c;
    stride = (long)B * CH;
    o = dP + r * CH + c;
  }
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int k0 = 0; k0 < n; k0 += 8) {
    float4 v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = *reinterpret_cast<const float4*>(p + (long)min(k0 + k, n - 1) * stride);
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (k0 + k < n) { s.x += v[k].x; s.y += v[k].y; s.z += v[k].z; s.w += v[k].w; }   // (in slot order: fixed summation order)
  }
  *reinterpret_cast<float4*>(o) = s;
}

void bwd_ws_plan(int E, int B, int* total, int* per, int* nwg, int* slots) {
  const int t = E * (B / UR) * B;
  int p = (t + device_cus() - 1) / device_cus();
  p = std::max(p, (B + 7) / 8);                      // (small problems: at most ~9 slots per (estimator, y block))
  *total = t; *per = p; *nwg = (t + p - 1) / p; *slots = (B + p - 1) / p + 1;
}

}  // namespace

bool concat_bwd_ws_supported(int B, int hid) { return hid == CH && B >= UR && B % UR == 0; }

// floats of scratch (dq_part followed by dp_part) the weights-stationary backward needs
long concat_bwd_ws_scratch(int E, int B) {
  if (!concat_bwd_ws_supported(B, CH)) return 0;
  int total, per, nwg, slots;
  bwd_ws_plan(E, B, &total, &per, &nwg, &slots);
  return (long)E * (B / UR) * slots * UR * CH + (long)E * (B / UR) * B * CH;
}

int concat_bwd_ws(hipStream_t s, const ConcatBwdArgs& a) {
  if (!concat_bwd_ws_supported(a.B, CH)) return set_error(MIMRL_ERR_ARG, "concat_bwd_ws: batch %d unsupported", a.B);
  if (!a.compact || !a.dQ || !a.dq_part || !a.dP || !a.ds || !(a.m0 && a.m1 && a.m2))
    return set_error(MIMRL_ERR_ARG, "concat_bwd_ws: needs the compact saves, dQ and its scratch");
  if ((a.dz2 != nullptr) != (a.dz1 != nullptr) || (a.dz2 && !(a.db1 && a.db2 && a.db3)))
    return set_error(MIMRL_ERR_ARG, "concat_bwd_ws: the weight-gradient outputs come together");
  int total, per, nwg, slots;
  bwd_ws_plan(a.E, a.B, &total, &per, &nwg, &slots);
  float* dp_part = a.dq_part + (long)a.E * (a.B / UR) * slots * UR * CH;
  if (a.dz2 && a.no_dz2) hipLaunchKernelGGL((concat_bwd_ws_kernel<true, false>), dim3((unsigned)nwg), dim3(512), 0, s, a, total, per, slots, dp_part);
  else if (a.dz2) hipLaunchKernelGGL(concat_bwd_ws_kernel<true>, dim3((unsigned)nwg), dim3(512), 0, s, a, total, per, slots, dp_part);
  else hipLaunchKernelGGL(concat_bwd_ws_kernel<false>, dim3((unsigned)nwg), dim3(512), 0, s, a, total, per, slots, dp_part);
  LAUNCH_CHECK();
  const int nq = a.E * (a.B / UR) * 8, np = (a.E * a.B + 3) / 4;
  hipLaunchKernelGGL(concat_ws_reduce_kernel, dim3((unsigned)(nq + np)), dim3(256), 0, s, a.dq_part, a.dQ, dp_part, a.dP, a.E, a.B, per, slots);
  LAUNCH_CHECK();
  return MIMRL_OK;
}

}  // namespace mimrl
