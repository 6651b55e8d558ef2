// Device helpers shared by the persistent recurrence kernels (gru.hip, lstm.hip): the LDS state tile of a 4-batch-row workgroup, its
// MFMA A-fragments (every batch row replicated over 4 MFMA rows, so that accumulator register 0 of lane (n, kq) is the value of batch row
// kq and column n), the LDS-only workgroup barrier, and per-lane unit-pair loads / stores.  Included inside `namespace mimrl { namespace {`.
#pragma once

constexpr int BR = 4;        // batch rows per workgroup

typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;

// LDS state tile (4 batch rows).  bf16: [batch][k] rows of (K+32) bf16 -- the row pitch is 16 words mod 64, so the 16
//                  distinct 16-byte chunks a wave reads (4 rows x 4 k-quarters, each broadcast to 4 lanes) hit
//                  16 different bank groups;  fp32: [k][batch] (16 consecutive words per wave read).
template <bool BF16, int K>
struct Tile;
template <int K>
struct Tile<true, K> {
  __bf16 v[BR][K + 32];
};
template <int K>
struct Tile<false, K> {
  float v[K][BR];
};

// D[m, n] += sum_k A[m, k] B[k, n];  A = state rows, B = weight columns
__device__ __forceinline__ f32x4 mfma16(const bf16x8& a, const bf16x8& b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma16(const float& a, const float& b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// A-fragment of k-step ks: MFMA row m carries batch row m>>2 (every batch row replicated over 4 MFMA rows), so the
// accumulator register 0 of lane (n, kq) -- MFMA row 4kq -- is the result for batch row kq: the 16x16 result tile lands
// as ONE useful value per lane, 16 units x 4 batch rows, and the gate math runs on all 64 lanes without any shuffle.
template <int K>
__device__ __forceinline__ bf16x8 state_frag(const Tile<true, K>& t, int ks, int lane) {
  return *reinterpret_cast<const bf16x8*>(&t.v[(lane & 15) >> 2][ks * 32 + 8 * (lane >> 4)]);
}
template <int K>
__device__ __forceinline__ float state_frag(const Tile<false, K>& t, int ks, int lane) {
  return t.v[ks * 4 + (lane >> 4)][(lane & 15) >> 2];
}
// write "k" values k0, k0+1 of batch row b
template <int K>
__device__ __forceinline__ void put2(Tile<true, K>& t, int b, int k0, float x0, float x1) {
  bf16x2 p;
  p[0] = to_bf16(x0); p[1] = to_bf16(x1);
  *reinterpret_cast<bf16x2*>(&t.v[b][k0]) = p;
}
template <int K>
__device__ __forceinline__ void put2(Tile<false, K>& t, int b, int k0, float x0, float x1) {
  t.v[k0][b] = x0; t.v[k0 + 1][b] = x1;
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt(0), i.e. it would wait every step
// for the saved-gate / output stores and for the prefetched gx loads -- the whole point of the prefetch is to keep
// them in flight across the step boundary.  LDS operations of a wave complete in order, so lgkmcnt(0) before
// s_barrier makes this wave's state-tile writes visible to the other waves after the barrier.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// UPL consecutive values (units u0 .. u0 + UPL - 1) of one lane: 4- or 8-byte accesses
template <int UPL> struct Pack;
template <> struct Pack<2> { using F = float2; using G = bf16x8; };      // F: UPL floats; G: packed bf16 gate record {r z n hn} x UPL
template <> struct Pack<1> { using F = float;  using G = bf16x4; };
template <int UPL>
__device__ __forceinline__ void ldu(const float* p, float (&v)[UPL]) {
  if constexpr (UPL == 2) { const float2 x = *reinterpret_cast<const float2*>(p); v[0] = x.x; v[1] = x.y; } else v[0] = *p;
}
template <int UPL>
__device__ __forceinline__ void stu(float* p, const float (&v)[UPL]) {
  if constexpr (UPL == 2) *reinterpret_cast<float2*>(p) = make_float2(v[0], v[1]); else *p = v[0];
}
template <int UPL, int K>
__device__ __forceinline__ void putu(Tile<true, K>& t, int b, int k0, const float (&v)[UPL]) {
  if constexpr (UPL == 2) { bf16x2 q; q[0] = to_bf16(v[0]); q[1] = to_bf16(v[1]); *reinterpret_cast<bf16x2*>(&t.v[b][k0]) = q; }
  else t.v[b][k0] = to_bf16(v[0]);
}
template <int UPL, int K>
__device__ __forceinline__ void putu(Tile<false, K>& t, int b, int k0, const float (&v)[UPL]) {
#pragma unroll
  for (int e = 0; e < UPL; ++e) t.v[k0 + e][b] = v[e];
}
