// Device helpers shared by the weights-stationary concat-critic kernels (concat_ws.hip, concat_ws4.hip): packed bf16 ReLU / sign words,
// DPP / permlane moves, scalar-base pointers, unit cursors.  See concat_ws.hip for why each exists.
#pragma once
#include "concat_fused.h"
#include <type_traits>

namespace mimrl {
namespace {

constexpr int CH = 256;            // hidden width (VMI.py:13-22 with hidden_dim 256)
constexpr int UR = 32;             // pair rows per pipeline unit
constexpr int AP = CH + 8;         // bf16 pitch of an activation tile row (528 B: conflict-free 16-byte fragment reads)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4v __attribute__((ext_vector_type(4)));

// ReLU as ONE instruction (fmaxf lowers to a canonicalising v_max x, x followed by the max with 0): v_med3_f32(x, 0, +inf)
__device__ __forceinline__ float relu1(float x) {
#ifdef WS_RELU_FMAXF
  return fmaxf(x, 0.f);
#else
  return __builtin_amdgcn_fmed3f(x, 0.f, __builtin_inff());
#endif
}

typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
typedef float f32x2v __attribute__((ext_vector_type(2)));
// (x, y) -> packed bf16 pair of (relu x, relu y): v_cvt_pk_bf16_f32 + v_pk_max_i16 (a negative bf16 is a negative int16; rounding to bf16 and
// ReLU commute) -- one vector instruction per VALUE where fp32 v_max + conversion took 1.5.  The matrix pipe and the vector ALU of a SIMD do
// not overlap across its two waves (measured: tools/hw/concat_ws_bench _xSYNTH), so every vector instruction of an epilogue is step time.
// (Real instructions, no inline asm: the scheduler -- and sched_group_barrier -- must see them as vector-ALU work.)
__device__ __forceinline__ uint32_t relu_pack2(float x, float y) {
  const f32x2v f = {x, y};
  const s16x2 z = {0, 0};
  return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2, __builtin_convertvector(f, bf16x2v)), z));
}
// bits |= (lo > 0) << K | (hi > 0) << (K + 1) for a packed pair of ReLU outputs: v_pk_min_i16 with (1, 1), then v_dot2_u32_u16 with
// (2^K, 2^(K+1)); K <= 14.  The empty asm hides from the optimiser that w = max(., 0): it would fold min(max(x, 0), 1) into compares and
// selects per element (88 v_cmp + 97 v_cndmask + 72 v_perm per iteration in the first attempt).
template <int K> __device__ __forceinline__ uint32_t sign_pair(uint32_t w, uint32_t bits, uint32_t) {
  asm("" : "+v"(w));
  const s16x2 one = {1, 1};
  const s16x2 t = __builtin_elementwise_min(__builtin_bit_cast(s16x2, w), one);
  const u16x2 wt = {(unsigned short)(1u << K), (unsigned short)(2u << K)};
  return __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, t), wt, bits, false);
}
// the 16 accumulator values of a lane in one 32 x 32 tile (features 8 q + 4 lh + i of the tile, q = i-quad) -> their ReLU as four packed bf16x4
// and their sign bits at the places they have in the (row, 32 features) word
__device__ __forceinline__ uint32_t relu_tile(const f32x16& acc, uint32_t (&out)[8], unsigned sh_lo, unsigned sh_hi, uint32_t ones) {
  uint32_t lo = 0u, hi = 0u;                    // quads 0, 1 -> bits 0-3, 8-11 of `lo`; quads 2, 3 -> the same places of `hi` (= bits 16-19, 24-27)
#pragma unroll
  for (int q = 0; q < 4; ++q) { out[2 * q] = relu_pack2(acc[4 * q], acc[4 * q + 1]); out[2 * q + 1] = relu_pack2(acc[4 * q + 2], acc[4 * q + 3]); }
  lo = sign_pair<0>(out[0], lo, ones); lo = sign_pair<2>(out[1], lo, ones); lo = sign_pair<8>(out[2], lo, ones); lo = sign_pair<10>(out[3], lo, ones);
  hi = sign_pair<0>(out[4], hi, ones); hi = sign_pair<2>(out[5], hi, ones); hi = sign_pair<8>(out[6], hi, ones); hi = sign_pair<10>(out[7], hi, ones);
  return (lo << sh_lo) | (hi << sh_hi);         // sh_lo = 4 lh, sh_hi = 16 + 4 lh
}
// DPP move (a VALU instruction, where __shfl_* is an LDS crossbar round trip): 0xB1 / 0x4E = quad_perm [1,0,3,2] / [2,3,0,1], 0x141 = row_half_mirror
template <int CTRL> __device__ __forceinline__ uint32_t dpp(uint32_t x) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, 0xf, 0xf, true); }
// x of this lane OR x of the lane 32 away (v_permlane32_swap: the upper half of the first operand trades places with the lower half of the second)
__device__ __forceinline__ uint32_t or_halves(uint32_t x) {
  const auto r = __builtin_amdgcn_permlane32_swap(x, x, false, false);
  return r[0] | r[1];
}
__device__ __forceinline__ float add_halves(float x) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// a wave-uniform pointer the optimiser cannot fold a lane offset into (it re-associated base + uniform + lane into a per-lane 64-bit pointer per
// access site, hoisted all of them out of the step loop and spilled them): the halves pass through v_readfirstlane, the access then uses the
// scalar-base + 32-bit-lane-offset addressing mode
// The result is typed as a GLOBAL (address space 1) pointer: through the integer round trip the compiler loses track of the kernel argument
// the address came from and would emit flat_load / flat_store (which also count on lgkmcnt, i.e. tie the LDS waits to memory traffic).
#define GLOBAL_AS __attribute__((address_space(1)))
template <class T> __device__ __forceinline__ GLOBAL_AS T* uptr(T* p) {
  const unsigned long long v = reinterpret_cast<unsigned long long>(p);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return (GLOBAL_AS T*)(((unsigned long long)hi << 32) | lo);
}

// position of a pipeline unit in the [E][B*B] pair-row space, advanced one unit at a time (everything here is wave-uniform: SALU -- the first
// version divided by runtime values four times per step, half a microsecond of dependent scalar / vector arithmetic per step)
struct UnitPos {
  int e, row0;       // estimator, first pair row inside it
  int gi, gj0;       // x row i and first y row j of the unit (B % 32 == 0: one x row, 32 consecutive y rows)
  int base;          // e * B*B + row0 (pair rows: < 2^31)
  __device__ __forceinline__ void advance(int B, int BB) {
    row0 += UR; base += UR; gj0 += UR;
    if (gj0 == B) { gj0 = 0; ++gi; }
    if (row0 == BB) { row0 = 0; gi = 0; ++e; }
  }
};
struct UnitRef { int e, base; };


}  // namespace
}  // namespace mimrl
