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
//   * a pipeline over UNITS of 32 pair rows: in iteration t the layer-1 waves work on unit t while the layer-2 waves work on unit t - 2 and
//     all waves generate layer 0 (relu(P_i + Q_j), the separable form of the first Linear) of unit t + 2; one barrier per TWO iterations.
//     LDS reads per unit and layer: 4 waves x 16 KB (the activation tile, once per 64-feature slice) instead of 160 KB.
// The bias is the accumulator's initial value; results are those of concat_fused.hip up to fp32 summation order inside a product.
#include "concat_ws_dev.h"

namespace mimrl {

namespace {




#ifdef WS_PHASE
// probe build (tools/hw/concat_ws_bench): 100 MHz ticks per phase, summed over the iterations of workgroup 0, for wave 0 (layer 1) and wave 4
// (layer 2): 0 phase A, 1 product, 2 phase C, 3 (layer 1: epilogue part of C), 4 copy-out, 5 scores, 6 barrier wait, 7 iterations
__device__ long long g_ws_phase[2][8];
__device__ long long g_ws_clk[2];   // shader-clock cycles / 100 MHz ticks of workgroup 0's whole run
#define WPH_DECL long long wph_t = (long long)wall_clock64()
#define WPH(i) do { if (blockIdx.x == 0 && (tid == 0 || tid == 256)) { const long long n_ = (long long)wall_clock64(); g_ws_phase[tid >> 8][i] += n_ - wph_t; wph_t = n_; } } while (0)
#else
#define WPH_DECL do { } while (0)
#define WPH(i) do { } while (0)
#endif

// SAVE: what the backward pass gets (ConcatFwdArgs::save): 0 nothing, 2 bf16 a0 / a1 + fp16 a2 + sign words (stage 1), 3 sign words only
//
// Iteration t of a run of U units (t = -2 .. U + 4); a barrier behind every ODD iteration (operand tiles live in rings of four units, every
// producer is two iterations ahead of its consumer: exactly one barrier in between -- half the barriers of a one-unit step):
//   layer-1 wave:  P / Q loads of unit t + 2 | product of unit t | epilogue -> act1[t & 3] | layer 0 of unit t + 2 -> act0[(t + 2) & 3]
//   layer-2 wave:  epilogue of unit t - 3 (the accumulators of its last product) | layer 0 of unit t + 2 from the loads it requested in
//                  iteration t - 1 | P / Q loads of unit t + 3 | product of unit t - 2
//   all waves:     stage 1: act1 of unit t - 2 leaves as bf16;  wave 0: scores of unit t - 5
// A0 (SAVE == 2): the bf16 copy of the generated layer 0 is written out (false: its consumer regenerates it from P and Q -- concat_dw.hip)
template <int SAVE, bool A0 = true>
__global__ __launch_bounds__(512) void concat_fwd_ws_kernel(ConcatFwdArgs a, int units_e, int total, int per) {
#ifdef WS_PHASE
  const long long clk0 = (long long)clock64(), wall0 = (long long)wall_clock64();
#endif
  __shared__ __attribute__((aligned(16))) __bf16 act0[4][UR][AP];   // layer-0 outputs (operand of layer 1), ring over units
  __shared__ __attribute__((aligned(16))) __bf16 act1[4][UR][AP];   // layer-1 outputs (operand of layer 2)
  __shared__ __attribute__((aligned(16))) float sbias[2][CH];       // [layer][feature]: each wave writes and reads only its own 64-feature slice
  __shared__ __attribute__((aligned(16))) float sw3[CH];            // score-head weight (layer-2 waves, own slice)
  __shared__ LdsAcc sc[4][UR];                                      // score sums of a unit over the four layer-2 waves
  const int tid = threadIdx.x, lane = tid & 63, lr = lane & 31, lh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);        // (uniform: the role branches and every unit index below stay on the scalar unit)
  const int role = wave >> 2, ws = wave & 3;       // role 0: layer 1, role 1: layer 2; ws: 64-feature slice
  const int B = a.B, BB = B * B;
  const int u0 = blockIdx.x * per, U = min(total, u0 + per) - u0;
  if (U <= 0) return;
  if (tid < 4 * UR) sc[tid >> 5][tid & 31].zero();
  bf16x8 wf[2][16];                                // this wave's weight slice: [feature tile][k-step], A fragments of v_mfma_f32_32x32x16_bf16
#pragma unroll
  for (int ct = 0; ct < 2; ++ct)
#pragma unroll
    for (int ks = 0; ks < 16; ++ks)
#pragma unroll
      for (int i = 0; i < 8; ++i) wf[ct][ks][i] = (__bf16)0.f;
  int my_e = -1;
  float b3e = 0.f;                                 // score-head bias of the current estimator (layer-2 waves)
  const unsigned c4 = lane * 4;                    // layer-0 generation: lane = column quad, wave w = rows w, w + 8, w + 16, w + 24 of the unit
  const unsigned nib_sh = 4 * (lane & 7);
  const unsigned lane8 = lane >> 3, lr8 = lr * 8;  // sign-word slots: (row of the wave, 32-column group of the lane) / (row of the lane)
  const unsigned row_k = lr * CH + 8 * lh;         // [row of the lane][k half] in a [.][256] array: weight fragments
  const unsigned row_f = lr * CH + 4 * lh;         // [row of the lane][feature quad half]: fp32 a2
  const unsigned sh_lo = 4 * lh, sh_hi = 16 + 4 * lh;
  uint32_t ones = 0x00010001u;
  asm volatile("" : "+v"(ones));                  // (a register operand of v_pk_min_u16: kept out of constant folding)
  float4 xq = make_float4(0.f, 0.f, 0.f, 0.f), yq[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) yq[q] = xq;
  f32x16 acc[2];
#pragma unroll
  for (int ct = 0; ct < 2; ++ct)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[ct][r] = 0.f;
  // unit cursors: G2 / G3 = units t + 2 / t + 3 (layer-0 generation), R[k] = unit t - k (k = -1 .. 5 kept as R1m, R0 .. R5)
  UnitPos G2, G3;
  G2.e = u0 / units_e; G2.row0 = (u0 - G2.e * units_e) * UR; G2.base = G2.e * BB + G2.row0;
  G2.gi = G2.row0 / B; G2.gj0 = G2.row0 - G2.gi * B;
  G3 = G2; G3.advance(B, BB);                      // (t = -2: unit t + 2 = unit 0)
  UnitRef R1m = {G2.e, G2.base}, R0 = R1m, R1 = R1m, R2 = R1m, R3 = R1m, R4 = R1m, R5 = R1m;

  auto issue_gen_loads = [&](const UnitPos& u) __attribute__((always_inline)) {
    // (every global address below = a wave-uniform 64-bit base + ONE of a handful of unsigned 32-bit lane offsets: saddr + voffset addressing.
    //  Per-lane 64-bit pointers for each of the ~12 access sites were hoisted out of the step loop and spilled: 26-43 scratch registers)
    const GLOBAL_AS float* Pp = uptr(a.P + ((long)u.e * B + u.gi) * CH);
    const GLOBAL_AS float* Qp = uptr(a.Q + ((long)u.e * B + u.gj0 + wave) * CH);
    { const f32x4v t_ = *(const GLOBAL_AS f32x4v*)(Pp + c4); xq = make_float4(t_[0], t_[1], t_[2], t_[3]); }
#pragma unroll
    for (int q = 0; q < 4; ++q) { const f32x4v t_ = *(const GLOBAL_AS f32x4v*)(Qp + (long)(8 * q) * CH + c4); yq[q] = make_float4(t_[0], t_[1], t_[2], t_[3]); }
  };
  // layer 0 of unit u -> act0[buf] (read by the layer-1 waves two iterations on), its sign words, stage 1: its bf16 copy.
  // ALL the arithmetic first, then the stores, and no store sits behind a lane branch: stores count on vmcnt like loads, and behind a branch
  // the compiler waits with vmcnt(0) -- the second quad's P / Q values then waited for the FIRST quad's sign-word store to complete (a memory
  // round trip per quad: 40 of the first version's 148 us).  Every lane of an 8-lane group stores the group's word (same address, same value).
  auto gen_finish = [&](const UnitPos& u, int buf) __attribute__((always_inline)) {
    uint32_t bq[4][2];
    uint32_t nq[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      bq[q][0] = relu_pack2(xq.x + yq[q].x, xq.y + yq[q].y);
      bq[q][1] = relu_pack2(xq.z + yq[q].z, xq.w + yq[q].w);
      if (SAVE >= 2) {   // sign word of (row, 32 columns) = the nibbles of 8 neighbouring lanes (this wave holds one row: lane = column quad)
        uint32_t w = sign_pair<2>(bq[q][1], sign_pair<0>(bq[q][0], 0u, ones), ones) << nib_sh;   // this lane's nibble at its place in its group's word ...
        w |= dpp<0xB1>(w);                                          // ... OR over the group, butterfly: lane ^ 1 (quad_perm [1,0,3,2]),
        w |= dpp<0x4E>(w);                                          //     lane ^ 2 (quad_perm [2,3,0,1]),
        w |= dpp<0x141>(w);                                         //     the other quad of the group (row_half_mirror)
        nq[q] = w;                                                  // every lane of the group holds the word
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int row = wave + 8 * q;
      typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
      u32x2 b; b[0] = bq[q][0]; b[1] = bq[q][1];
      *reinterpret_cast<u32x2*>(&act0[buf][row][c4]) = b;
      if (SAVE == 2 && A0) { GLOBAL_AS __bf16* o = uptr(a.a0b + ((long)u.base + row) * CH); *(GLOBAL_AS u32x2*)(o + c4) = b; }
      if (SAVE >= 2) { GLOBAL_AS uint32_t* o = uptr(a.m0 + ((long)u.base + row) * 8); o[lane8] = nq[q]; }
    }
  };
  // this wave's slice of its layer for estimator e (first unit / the run entered the next estimator)
  auto ensure_weights = [&](int e) __attribute__((always_inline)) {
    if (e != my_e) {
      my_e = e;
      const GLOBAL_AS __bf16* W = uptr((role ? a.W2 : a.W1) + (long)e * a.pstride + (long)(ws * 64) * CH);
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) wf[ct][ks] = *(const GLOBAL_AS bf16x8*)(W + (long)(ct * 32) * CH + ks * 16 + row_k);
      sbias[role][ws * 64 + lane] = ((role ? a.b2 : a.b1) + (long)e * a.pstride)[ws * 64 + lane];
      if (role) { sw3[ws * 64 + lane] = a.w3[(long)e * a.pstride + ws * 64 + lane]; b3e = a.b3[(long)e * a.pstride]; }
      __builtin_amdgcn_wave_barrier();                   // (LDS operations of one wave complete in order: the reads below see these)
    }
  };
  auto product = [&](const __bf16 (*src)[AP]) __attribute__((always_inline)) {
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int q = 0; q < 4; ++q) {                      // accumulator quad q of tile ct = features 64 ws + 32 ct + 8 q + 4 lh + (0..3): starts at the bias
#ifdef WS_X_NOBIAS
        const float4 bb = make_float4(0.f, 0.f, 0.f, 0.f);
#else
        const float4 bb = *reinterpret_cast<const float4*>(&sbias[role][ws * 64 + ct * 32 + 8 * q + 4 * lh]);
#endif
        acc[ct][4 * q] = bb.x; acc[ct][4 * q + 1] = bb.y; acc[ct][4 * q + 2] = bb.z; acc[ct][4 * q + 3] = bb.w;
      }
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
#ifdef WS_X_NOLDSREAD
      const bf16x8 bfr = wf[1][15 - ks];
      (void)src;
#else
      const bf16x8 bfr = *reinterpret_cast<const bf16x8*>(&src[lr][ks * 16 + 8 * lh]);
#endif
#ifdef WS_X_NOMFMA
      acc[0][ks] += (float)bfr[0] * (float)wf[0][ks][0]; acc[1][ks] += (float)bfr[1] * (float)wf[1][ks][0];
#else
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[0][ks], bfr, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[1][ks], bfr, acc[1], 0, 0, 0);
#endif
    }
  };
  // layer 1: ReLU -> bf16 operand tile of layer 2 (8-byte LDS stores) + the sign word of (row, 32 features)
  auto epilogue1 = [&](const UnitRef& u, int buf) __attribute__((always_inline)) {
#ifdef WS_X_NOEPI
    { float t = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) t += acc[0][r] + acc[1][r];
      if (t == 1.2345f) { GLOBAL_AS float* o = uptr(a.scores + u.base); o[(unsigned)lr] = t; } }
    return;
#endif
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      uint32_t pk[8];
      uint32_t bits = relu_tile(acc[ct], pk, sh_lo, sh_hi, ones);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
        u32x2 b; b[0] = pk[2 * q]; b[1] = pk[2 * q + 1];
        *reinterpret_cast<u32x2*>(&act1[buf][lr][ws * 64 + ct * 32 + 8 * q + 4 * lh]) = b;
      }
      if (SAVE >= 2) {
        bits = or_halves(bits);
        { GLOBAL_AS uint32_t* o = uptr(a.m1 + (long)u.base * 8 + ws * 2 + ct); o[lr8] = bits; }   // (both halves hold the word: no lane branch around a store)
      }
    }
  };
  // layer 2: ReLU -> sign word, fp16 a2 (stage 1: the score head's weight gradient reads it), score head partial dot product (on the fp32 values)
  auto epilogue2 = [&](const UnitRef& u, int slot) __attribute__((always_inline)) {
#ifdef WS_X_NOEPI
    { float t = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) t += acc[0][r] + acc[1][r];
      if (t == 1.2345f) { GLOBAL_AS float* o = uptr(a.scores + u.base); o[(unsigned)lr] = t; } }
    return;
#endif
    float hp = 0.f;
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      if (SAVE >= 2) {
        uint32_t pk[8];
        const uint32_t bits = or_halves(relu_tile(acc[ct], pk, sh_lo, sh_hi, ones));
        { GLOBAL_AS uint32_t* o = uptr(a.m2 + (long)u.base * 8 + ws * 2 + ct); o[lr8] = bits; }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int f0 = ws * 64 + ct * 32 + 8 * q + 4 * lh;
        float4 v;
        v.x = relu1(acc[ct][4 * q]); v.y = relu1(acc[ct][4 * q + 1]); v.z = relu1(acc[ct][4 * q + 2]); v.w = relu1(acc[ct][4 * q + 3]);
        const float4 w3v = *reinterpret_cast<const float4*>(&sw3[f0]);
        hp += v.x * w3v.x + v.y * w3v.y + v.z * w3v.z + v.w * w3v.w;
        if (SAVE == 2) {   // a2 leaves as fp16 (concat_fwd_a2_f16: same element indices behind the float-typed pointer)
          GLOBAL_AS _Float16* o = (GLOBAL_AS _Float16*)uptr(reinterpret_cast<_Float16*>(a.a2) + (long)u.base * CH + ws * 64 + ct * 32 + 8 * q);
          f16x4 t_; t_[0] = to_f16_sat(v.x); t_[1] = to_f16_sat(v.y); t_[2] = to_f16_sat(v.z); t_[3] = to_f16_sat(v.w);
          *(GLOBAL_AS f16x4*)(o + row_f) = t_;
        }
      }
    }
    hp = add_halves(hp);
    if (ws == 0) hp += b3e;                              // (the score head's bias, once per row)
    if (lh == 0) sc[slot][lr].add(hp);
  };

#ifdef WS_X_NOGEN
  const bool gen_any = false;
#else
  const bool gen_any = true;
#endif
  if (role == 1 && gen_any) issue_gen_loads(G2);         // (layer-2 waves use their loads at the START of an iteration: unit 0's go out here)
  WPH_DECL;
#pragma unroll 1
  for (int t = -2; t <= U + 4; ++t) {
    const bool gen_on = gen_any && t + 2 < U;            // unit t + 2 exists: its layer 0 is generated in this iteration
    // phase A (vector work of the layer-2 waves; the layer-1 waves only request their loads)
    if (role == 0) {
      if (gen_on) issue_gen_loads(G2);
    } else {
      if (t >= 3 && t - 3 < U) epilogue2(R3, (t - 3) & 3);   // (the accumulators of the product at the end of the last iteration)
      if (gen_on) gen_finish(G2, (t + 2) & 3);
      if (gen_any && t + 3 < U) issue_gen_loads(G3);     // (used at the start of the NEXT iteration: in flight over this one's product --
                                                         //  requested behind the product they arrived ~0.5 us late, every iteration)
    }
    WPH(0);
#ifdef WS_X_SYNTH
    // synthetic co-issue test: layer-1 waves = 64 MFMAs per iteration (SYNTH & 1), layer-2 waves = 512 independent VALU FMAs (SYNTH & 2)
    if (role == 0 && (WS_X_SYNTH & 1)) {
#pragma unroll
      for (int rep = 0; rep < 2; ++rep)
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
          acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[0][ks], wf[1][15 - ks], acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[1][ks], wf[0][15 - ks], acc[1], 0, 0, 0);
        }
    }
    if (role == 0 && (WS_X_SYNTH & 4)) {     // ONE wave: 8 independent vector FMAs behind every MFMA
      float f[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) f[r] = xq.x + r;
#pragma unroll
      for (int rep = 0; rep < 2; ++rep)
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
          acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[0][ks], wf[1][15 - ks], acc[0], 0, 0, 0);
#pragma unroll
          for (int r = 0; r < 8; ++r) f[r] = f[r] * 1.0001f + 0.5f;
          acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[1][ks], wf[0][15 - ks], acc[1], 0, 0, 0);
#pragma unroll
          for (int r = 8; r < 16; ++r) f[r] = f[r] * 0.9999f + 0.25f;
        }
      float x = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) x += f[r];
      xq.x = x * 1e-30f;
    }
    if (role == 1 && (WS_X_SYNTH & 2)) {
#pragma unroll 1
      for (int rep = 0; rep < 16; ++rep) {
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[0][r] = acc[0][r] * 1.0001f + 0.5f; acc[1][r] = acc[1][r] * 0.9999f + 0.25f; }
      }
    }
    if (t == U + 4) { float x = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) x += acc[0][r] + acc[1][r];
      if (x == 1.2345f) a.scores[lane] = x; }
    __syncthreads();
    continue;
#endif
    // phase B: this wave's product -- layer 1 on unit t, layer 2 on unit t - 2 (ONE call site: the 128 weight registers are not duplicated)
    {
      const int pu = t - 2 * role;
      if (pu >= 0 && pu < U) {
        ensure_weights(role ? R2.e : R0.e);
        product(role ? act1[(t - 2) & 3] : act0[t & 3]);
      }
    }
    WPH(1);
    // phase C (vector work of the layer-1 waves)
    if (role == 0) {
      if (t >= 0 && t < U) epilogue1(R0, t & 3);
      WPH(3);
      if (gen_on) gen_finish(G2, (t + 2) & 3);
    }
    WPH(2);
    // ---- stage 1: the finished layer-1 tile of unit t - 2 leaves as bf16 in whole 512-byte rows (operand of the dW2 product)
    if (SAVE == 2 && t >= 2 && t - 2 < U) {
      GLOBAL_AS __bf16* o = uptr(a.a1b + (long)R2.base * CH);
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const unsigned idx = tid + 512 * q, row = idx >> 5, c8 = (idx & 31) * 8;
        *(GLOBAL_AS u32x4*)(o + (row * CH + c8)) = *reinterpret_cast<const u32x4*>(&act1[(t - 2) & 3][row][c8]);
      }
    }
    WPH(4);
    // ---- scores of unit t - 5 (its four layer-2 waves added their parts during iteration t - 2: a barrier has passed since)
    if (t >= 5 && wave == 0) {                           // (lanes 32 .. 63 repeat lanes 0 .. 31: same address, same value)
      GLOBAL_AS float* o = uptr(a.scores + R5.base);
      o[(unsigned)lr] = sc[(t - 5) & 3][lr].get();
      sc[(t - 5) & 3][lr].zero();
    }
    WPH(5);
    // ---- next iteration's cursors
    R5 = R4; R4 = R3; R3 = R2; R2 = R1; R1 = R0; R0 = R1m; R1m = UnitRef{G2.e, G2.base};
    G2 = G3;
    G3.advance(B, BB);
    if (t & 1) __syncthreads();
    WPH(6);
#ifdef WS_PHASE
    if (blockIdx.x == 0 && (tid == 0 || tid == 256)) g_ws_phase[tid >> 8][7] += 1;
#endif
  }
#ifdef WS_PHASE
  if (blockIdx.x == 0 && tid == 0) { g_ws_clk[0] += (long long)clock64() - clk0; g_ws_clk[1] += (long long)wall_clock64() - wall0; }
#endif
}

}  // namespace

#ifdef WS_PHASE
int concat_ws_read_phases(long long* out) {   // 16 phase slots + 2 clock slots, read-and-clear
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ws_phase), sizeof(long long) * 16) != hipSuccess) return 1;
  if (hipMemcpyFromSymbol(out + 16, HIP_SYMBOL(g_ws_clk), sizeof(long long) * 2) != hipSuccess) return 1;
  long long z[18] = {0};
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_ws_clk), z, sizeof(long long) * 2) != hipSuccess) return 1;
  return hipMemcpyToSymbol(HIP_SYMBOL(g_ws_phase), z, sizeof(long long) * 16) == hipSuccess ? 0 : 1;
}
#endif

bool concat_fwd_ws_supported(int B, int hid, int save) { return hid == CH && B >= UR && B % UR == 0 && (save == 0 || save == 2 || save == 3); }

int concat_fwd_ws(hipStream_t s, const ConcatFwdArgs& a) {
  if (!concat_fwd_ws_supported(a.B, CH, a.save)) return set_error(MIMRL_ERR_ARG, "concat_fwd_ws: batch %d / save %d unsupported", a.B, a.save);
  const int units_e = (int)(((long)a.B * a.B) / UR), total = a.E * units_e;
  const int nwg0 = std::min(device_cus(), total), per = (total + nwg0 - 1) / nwg0, nwg = (total + per - 1) / per;
  const dim3 grid((unsigned)nwg);
  if (a.save == 0) hipLaunchKernelGGL(concat_fwd_ws_kernel<0>, grid, dim3(512), 0, s, a, units_e, total, per);
  else if (a.save == 2 && a.no_a0) hipLaunchKernelGGL((concat_fwd_ws_kernel<2, false>), grid, dim3(512), 0, s, a, units_e, total, per);
  else if (a.save == 2) hipLaunchKernelGGL(concat_fwd_ws_kernel<2>, grid, dim3(512), 0, s, a, units_e, total, per);
  else hipLaunchKernelGGL(concat_fwd_ws_kernel<3>, grid, dim3(512), 0, s, a, units_e, total, per);
  LAUNCH_CHECK();
  return MIMRL_OK;
}

}  // namespace mimrl
