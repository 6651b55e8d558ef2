// Persistent bidirectional-GRU recurrence kernels (forward and BPTT) for gfx950.
//
// Reference semantics: nn.GRU(d,128,num_layers=2,bidirectional) fed with packed sequences (Model.py:254-255,
// 441-447): gate order r,z,n; n = tanh(gx_n + r*(W_hn h + b_hn)); positions t >= length emit 0 and do not advance
// the state; the reverse direction therefore effectively starts at t = length-1.
//
// MI355X design.  The recurrence is the latency floor of the whole training step: 2 layers x T
// dependent steps per pass.  It is independent across batch rows, so one workgroup owns 4 batch rows of one
// (modality, direction) for all T steps -- no inter-workgroup traffic at all, and B=128 spreads over 128 CUs.
// Inside the workgroup the four waves (one per SIMD) split the 128 hidden units 32/32/32/32; each wave keeps its
// slice of W_hh (all three gates) in REGISTERS as ready-made MFMA B-fragments for the whole sequence (bf16: 96
// VGPRs, fp32: 192 VGPRs), so the only per-step shared-memory traffic is the 4x128 state tile (double-buffered in
// LDS, ONE barrier per step).  The per-step cost is the wave's own instruction stream (MFMA issue + gate math on
// the quarter-rate transcendental unit), so the mapping is chosen to leave NO idle lanes in the gate math:
// gates[batch, unit] = h[batch, :] . W_hh[unit, :]^T with the state as the A operand, each batch row replicated
// over 4 MFMA rows, so that accumulator register 0 of lane (n, kq) is exactly (batch kq, unit n) -- one useful
// value per lane per MFMA, the r/z/n values of one (unit, batch) pair in the same lane, two adjacent units per
// lane: all loads / stores are 8-byte, the gate math is register-only and 4x shorter than with the batch on the
// MFMA column axis (where 12 of 16 columns were padding).
// The input projections gx = x W_ih^T + b_ih are hoisted out of the loop into one GEMM per layer (gemm.hip).
//
// fp32 mode: v_mfma_f32_16x16x4_f32 (bit-exact fp32 fma chains) -- parity mode.
// bf16 mode: v_mfma_f32_16x16x32_bf16, W_hh and the state tile rounded to bf16 for the product, fp32 state,
//            fp32 accumulate, fp32 gate math.
#include "gru.h"

#include <cstdlib>
#include <type_traits>

namespace mimrl {

namespace {

constexpr int H = 128;       // hidden size (= d_common; the reference only runs with 128)
constexpr int G = 3 * H;
// 4 waves per workgroup; wave w owns units [32w, 32w+32); lane (n = lane&15, kq = lane>>4) owns the two adjacent
// units u0 = 32w + 2n, u0+1 of batch row kq

template <bool BF16>
struct Cfg;
template <>
struct Cfg<true> {
  static constexpr int KS_F = H / 32;   // k-steps over H (forward product)
  static constexpr int KS_B = G / 32;   // k-steps over 3H (backward product)
  using Frag = bf16x8;
};
template <>
struct Cfg<false> {
  static constexpr int KS_F = H / 4;
  static constexpr int KS_B = G / 4;
  using Frag = float;
};

#include "recur_device.h"

// BPTT outputs as packed bf16 (DGBF): `p` is the fp32-typed base of a bf16 array, `i` the element index
template <bool DGBF, int UPL>
__device__ __forceinline__ void stuo(float* p, long i, const float (&v)[UPL]) {
  if constexpr (DGBF) {
    if constexpr (UPL == 2) { bf16x2 q; q[0] = to_bf16(v[0]); q[1] = to_bf16(v[1]); *reinterpret_cast<bf16x2*>(reinterpret_cast<__bf16*>(p) + i) = q; }
    else reinterpret_cast<__bf16*>(p)[i] = to_bf16(v[0]);
  } else stu<UPL>(p + i, v);
}

// Operand loads of the software pipeline with EXPLICIT wait counts (bf16 kernels).  The compiler's waitcnt pass merges the counts of
// the loop's incoming edges conservatively: at the loop header it waited with vmcnt(1) / vmcnt(4) for loads that have EIGHT younger
// operations behind them (the previous step's five stores and three prefetches), i.e. for the youngest prefetches too -- the memory
// latency the pipeline exists to hide was back on the dependent chain of every second cell step (measured by removing the loads:
// 0.43 of the BPTT's 1.32 us per step).  Loads issued by inline asm are invisible to that pass; vm_wait<N>() is the one wait, with the
// exact count (every step issues the same sequence of loads and stores, unconditionally: see the padding-lane comment), and ties
// the destination registers so that no use can be scheduled in front of it.
__device__ __forceinline__ void gld(float& d, const float* p) { asm volatile("global_load_dword %0, %1, off" : "=v"(d) : "v"(p) : "memory"); }
__device__ __forceinline__ void gld(float2& d, const float* p) { asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(d) : "v"(p) : "memory"); }
__device__ __forceinline__ void gld(bf16x4& d, const float* p) { asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(d) : "v"(p) : "memory"); }
__device__ __forceinline__ void gld(bf16x8& d, const float* p) { asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(d) : "v"(p) : "memory"); }
__device__ __forceinline__ void gld(uint32_t& d, const void* p) { asm volatile("global_load_dword %0, %1, off" : "=v"(d) : "v"(p) : "memory"); }
template <int N, class A, class B, class C>
__device__ __forceinline__ void vm_wait(A& a, B& b, C& c) {
  asm volatile("s_waitcnt vmcnt(%3)" : "+v"(a), "+v"(b), "+v"(c) : "n"(N) : "memory");
}
__device__ __forceinline__ float comp(const float2& v, int e) { return e ? v.y : v.x; }
__device__ __forceinline__ float comp(const float& v, int) { return v; }
// 16-bit stored operand pairs of the BPTT kernel (IO16): two adjacent units in one dword
struct PairBf { uint32_t v; };   // bf16 x 2 (dout)
struct PairH { uint32_t v; };    // fp16 x 2 (the forward kernel's fp16 copy of its outputs: h_prev)
__device__ __forceinline__ void gld(PairBf& d, const void* p) { gld(d.v, p); }
__device__ __forceinline__ void gld(PairH& d, const void* p) { gld(d.v, p); }
__device__ __forceinline__ float comp(const PairBf& p, int e) { return __builtin_bit_cast(float, e ? (p.v & 0xffff0000u) : (p.v << 16)); }
__device__ __forceinline__ float comp(const PairH& p, int e) {
  typedef __attribute__((ext_vector_type(2))) _Float16 h2;
  const h2 h = __builtin_bit_cast(h2, p.v);
  return (float)(e ? h[1] : h[0]);
}

// saved-gate slab, "lane-native": one record {r z n hn} x UPL per (t, tile, wave, lane); bf16 mode packs it into ONE 8 x UPL-byte
// vector (a wave instruction stores 512 x UPL bytes contiguous), fp32 mode into UPL 16-byte ones.
template <bool BF16, int UPL>
struct SvRec { static constexpr int F = (BF16 ? 2 : 4) * UPL; };   // floats per record
template <bool BF16, int UPL>
__device__ __forceinline__ long sv_index(int t, int ntile, int tile, int w, int lane) {
  return ((((long)t * ntile + tile) * (8 / UPL) + w) * 64 + lane) * SvRec<BF16, UPL>::F;
}
template <int UPL>
struct Gates { float r[UPL], z[UPL], n[UPL], hn[UPL]; };

template <bool BF16, int UPL>
__device__ __forceinline__ void save_gates(float* p, const Gates<UPL>& g) {
  if constexpr (BF16) {
    typename Pack<UPL>::G a;
#pragma unroll
    for (int e = 0; e < UPL; ++e) { a[e] = to_bf16(g.r[e]); a[UPL + e] = to_bf16(g.z[e]); a[2 * UPL + e] = to_bf16(g.n[e]); a[3 * UPL + e] = to_bf16(g.hn[e]); }
    *reinterpret_cast<typename Pack<UPL>::G*>(p) = a;
  } else {
#pragma unroll
    for (int e = 0; e < UPL; ++e) *reinterpret_cast<float4*>(p + 4 * e) = make_float4(g.r[e], g.z[e], g.n[e], g.hn[e]);
  }
}
template <int UPL>
__device__ __forceinline__ void decode_gates(const typename Pack<UPL>::G& a, Gates<UPL>& g) {
#pragma unroll
  for (int e = 0; e < UPL; ++e) { g.r[e] = (float)a[e]; g.z[e] = (float)a[UPL + e]; g.n[e] = (float)a[2 * UPL + e]; g.hn[e] = (float)a[3 * UPL + e]; }
}
template <bool BF16, int UPL>
__device__ __forceinline__ void load_gates(const float* p, Gates<UPL>& g) {
  if constexpr (BF16) {
    decode_gates<UPL>(*reinterpret_cast<const typename Pack<UPL>::G*>(p), g);
  } else {
#pragma unroll
    for (int e = 0; e < UPL; ++e) { const float4 x = *reinterpret_cast<const float4*>(p + 4 * e); g.r[e] = x.x; g.z[e] = x.y; g.n[e] = x.z; g.hn[e] = x.w; }
  }
}

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void stamp_begin(const KernelStamp& k) {
  if (k.ring && threadIdx.x == 0) atomicMin(&k.ring[(((unsigned)*k.step & (unsigned)(k.slots - 1)) * 4u + k.id) * 2u], (unsigned long long)wall_clock64());
}
__device__ __forceinline__ void stamp_end(const KernelStamp& k) {
  if (k.ring && threadIdx.x == 0) atomicMin(&k.ring[(((unsigned)*k.step & (unsigned)(k.slots - 1)) * 4u + k.id) * 2u + 1u], ~(unsigned long long)wall_clock64());
}

// `make PHASE_PROBE=1` builds only (tools/gru_phase.sh): MIMRL_GRU_SKIP=<mask> removes one phase of the cell step from BOTH recurrence
// kernels -- results become wrong, the launch time shows what that phase costs on the dependent chain (the method that found the
// guarded loads in round 2 and the waitcnt merges in round 3).  1: products; 2: transcendental / gradient math; 4: global stores;
// 8: operand loads; 16: s_barrier; 32: LDS state tile (write + fragment reads).  Compiled out of the default build.
// (COMPILE-time: the kernels carry the mask as a template parameter and the probe build instantiates one copy per mask -- a run-time test
//  of the mask in the step loop is itself a branch whose join degrades the wait counts: that version of the probe ran 2x slower than
//  the kernel it was meant to explain.)
#define GSKIP(bit) ((SKIP & (bit)) != 0)

// `make PHASE_PROBE=1` builds only (tools/gru_bwd_phase.py, round 5): a cycle budget of the BPTT cell step from in-kernel stamps.  Phase
// elimination (above) does not work for this kernel -- removing a phase changes the register allocation around the explicit-wait loads and
// the table comes out non-monotone (profiles/r04_gru_phase.json).  Wave 0 of workgroup (0, 0, 0) reads the shader clock (s_memtime) at the
// phase boundaries of every cell step and sums the differences; the values are consumed at the END of the step only (an s_memtime result
// is waited for with lgkmcnt(0), which in the middle of the step would also wait for the LDS traffic it is meant to time).
// slots per layer (16): 0 operand wait, 1 gate / gradient math + LDS tile writes, 2 five global stores issued, 3 lgkmcnt(0) + s_barrier,
// 4 three prefetches issued, 5 twelve fragment reads returned, 6 24 MFMAs retired, 7 steps, 8 / 9 wall clock (100 MHz) at entry / exit
#ifdef MIMRL_PHASE_PROBE
__device__ long long g_gru_bwd_phase[32];
#define GPH_DECL unsigned long long gph_t[8]; long long gph_s[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define GPH(i) gph_t[i] = __builtin_amdgcn_s_memtime()
#define GPH_SUM() do { _Pragma("unroll") for (int i_ = 0; i_ < 7; ++i_) gph_s[i_] += (long long)(gph_t[i_ + 1] - gph_t[i_]); gph_s[7] += 1; } while (0)
#else
#define GPH_DECL
#define GPH(i) do { } while (0)
#define GPH_SUM() do { } while (0)
#endif

#ifndef GRU_BF16_MINB
#define GRU_BF16_MINB 2   // -DGRU_BF16_MINB=1: the AGPR-using build of the reproducibility hunt, debugging only
#endif
// UPL = hidden units per lane: 2 -> 4 waves x 32 units (256 threads), 1 -> 8 waves x 16 units (512 threads: two waves per SIMD).
// Round 4.  The per-step chain of a wave is  LDS read -> its MFMAs -> gate math (quarter-rate v_exp / v_rcp) -> LDS write -> barrier,
// and with ONE wave per SIMD nothing runs under any of it.  With two waves per SIMD each wave has half the products (12 instead of 24)
// and half the gate math (one unit per lane), and one wave's transcendentals run under the other's MFMAs; the matrix pipe of a SIMD
// still sees the same 24 products per step -- its floor, 384 cycles -- but no longer waits for 2 x the gate math in between.
template <bool BF16, bool SAVE, int UPL, int SKIP = 0, bool GXH = false, bool H16 = false, bool XIN = false, bool NO32 = false>   // GXH: gx stored as fp16; XIN: fused input projection; NO32: no fp32 outputs (H16 only)
__global__ __launch_bounds__(512 / UPL, BF16 ? (UPL == 2 ? GRU_BF16_MINB : 1) : 1) void gru_fwd_kernel(GruFwdArgs a) {
  using C = Cfg<BF16>;
  stamp_begin(a.stamp);
  __shared__ __attribute__((aligned(16))) Tile<BF16, H> hs[2];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int tile = blockIdx.x, dir = blockIdx.y, mod = blockIdx.z;
  const GruSeq& q = a.seq[mod][dir];
  const int B = a.B, T = a.T, ntile = gridDim.x;
  const int n = lane & 15, kq = lane >> 4;
  const int u0 = 16 * UPL * w + UPL * n;
  // padding lanes (kq >= btv, or rows past B) MIRROR the last real row of the tile instead of idling: they compute
  // and store exactly the same values, so no load or store in the loop is guarded and the waitcnt bookkeeping of the
  // software pipeline stays exact (a guarded access is a branch, and the wait after a branch merge is vmcnt(0))
  const int b = min(tile * a.btv + min(kq, a.btv - 1), B - 1);
  const int len = a.lens[mod][b];

  // ---- W_hh slice -> registers as MFMA B fragments: column n of N-tile (g, s) is gate row g*H + u0 + s
  typename C::Frag wr[3][UPL][C::KS_F];
#pragma unroll
  for (int g = 0; g < 3; ++g)
#pragma unroll
    for (int s = 0; s < UPL; ++s) {
      const float* row = q.w_hh + (long)(g * H + u0 + s) * H;
#pragma unroll
      for (int ks = 0; ks < C::KS_F; ++ks) {
        if constexpr (BF16) {
          const float4 lo = *reinterpret_cast<const float4*>(row + ks * 32 + 8 * kq);
          const float4 hi = *reinterpret_cast<const float4*>(row + ks * 32 + 8 * kq + 4);
          bf16x8 f;
          f[0] = to_bf16(lo.x); f[1] = to_bf16(lo.y); f[2] = to_bf16(lo.z); f[3] = to_bf16(lo.w);
          f[4] = to_bf16(hi.x); f[5] = to_bf16(hi.y); f[6] = to_bf16(hi.z); f[7] = to_bf16(hi.w);
          wr[g][s][ks] = f;
        } else {
          wr[g][s][ks] = row[ks * 4 + kq];
        }
      }
    }
  float bh[3][UPL];
#pragma unroll
  for (int g = 0; g < 3; ++g) ldu<UPL>(q.b_hh + g * H + u0, bh[g]);

  // ---- XIN: W_ih slice of this lane's unit as fp16 B fragments (k >= kp: zero), b_ih, and the batch row whose inputs feed this lane's
  //      A-fragment row (MFMA row m carries batch row m >> 2, like the state tile)
  [[maybe_unused]] f16x8 wx[3][3];
  [[maybe_unused]] float bi[3] = {0.f, 0.f, 0.f};
  [[maybe_unused]] const _Float16* x_b = nullptr;
  if constexpr (XIN) {
    static_assert(UPL == 1 && BF16 && !GXH, "fused input projection: 8-wave bf16 kernel, fp32 gate inputs");
    const _Float16* __restrict__ wih = a.wih[mod][dir];
    const int kp = a.kp;
#pragma unroll
    for (int g = 0; g < 3; ++g) {
      bi[g] = a.bih[mod][dir][g * H + u0];
#pragma unroll
      for (int ks = 0; ks < 3; ++ks) {
        const int k = ks * 32 + 8 * kq;
        f16x8 f = {0, 0, 0, 0, 0, 0, 0, 0};
        if (k < kp) f = *reinterpret_cast<const f16x8*>(wih + (long)(g * H + u0) * kp + k);
        wx[g][ks] = f;
      }
    }
    const int bx = min(tile * a.btv + min((lane & 15) >> 2, a.btv - 1), B - 1);
    x_b = a.xin[mod] + (long)bx * T * kp;
  }

  // ---- state
  float hreg[UPL];
#pragma unroll
  for (int s = 0; s < UPL; ++s) hreg[s] = 0.f;
  putu<UPL>(hs[0], kq, u0, hreg);
  __syncthreads();

  const float* gx_b = q.gx + (long)b * T * G + u0;                       // gx [B,T,3H]
  const _Float16* gxh_b = reinterpret_cast<const _Float16*>(q.gx) + (long)b * T * G + u0;   // (GXH: the same array as fp16 elements)
  float* out_b = q.out + (long)b * T * a.out_ld + dir * H + u0;          // out [B,T,out_ld]
  _Float16* out16_b = H16 ? q.out16 + (long)b * T * a.out_ld + dir * H + u0 : nullptr;   // (H16: the fp16 copy, same indexing)
  // (XIN: this 8-wave launch writes the slab in the FOUR-wave layout -- the record of lane pair (n, n ^ 1) -- so that the BPTT launch of the
  //  layer stays the 4-wave kernel: unit 16 w + n is element n & 1 of the record of 4-wave lane (kq, 8 (w & 1) + n / 2) of wave w / 2)
  float* sv_b = !SAVE ? nullptr : XIN ? q.saved + sv_index<BF16, 2>(0, ntile, tile, w >> 1, kq * 16 + 8 * (w & 1) + (n >> 1))
                                      : q.saved + sv_index<BF16, UPL>(0, ntile, tile, w, lane);
  const long sv_step = XIN ? (long)ntile * 4 * 64 * SvRec<BF16, 2>::F : (long)ntile * (8 / UPL) * 64 * SvRec<BF16, UPL>::F;

  // software pipeline, distance 2: gx of step+2 is requested at the end of step (two named buffers, loop unrolled by
  // two, so that no register copy has to wait for the youngest load); the tail re-reads the last step
  // (GXH: the buffer holds the RAW fp16 values; they are converted where the gate math reads them.  The first version converted inside
  //  load_gx -- a use right behind the load, i.e. a full memory latency on every cell step: 632 instead of 380 us at cfg3)
  typedef __attribute__((ext_vector_type(2))) _Float16 h2;
  struct GX {
    f16x8 xf[XIN ? 3 : 1];   // XIN: the raw input fragments of the step (no gx)
    typename std::conditional<GXH, typename std::conditional<UPL == 2, h2, _Float16>::type, float>::type v[3][GXH ? 1 : UPL];
    __device__ __forceinline__ float at(int g, int s) const {
      if constexpr (GXH && UPL == 2) return (float)v[g][0][s];
      else if constexpr (GXH) return (float)v[g][0];
      else return v[g][s];
    }
  };
  auto load_gx = [&](GX& dst, int step) {
    const int sc = step < T ? step : T - 1;
    const int t = dir ? T - 1 - sc : sc;
    // (compiler-tracked loads here: with the explicit-wait loads of the BPTT kernel this loop measured 31.0 instead of 29.3 us per
    //  launch -- its waitcnt counts are exact in every second step and two short in the others, and that beats one exact wait)
    if constexpr (XIN) {
      const _Float16* p = x_b + (long)t * a.kp;
#pragma unroll
      for (int ks = 0; ks < 3; ++ks) {
        int k = ks * 32 + 8 * (lane >> 4);
        k = k < a.kp ? k : a.kp - 8;                        // (past the packed width: any valid address, its weight fragment is zero)
        dst.xf[ks] = *reinterpret_cast<const f16x8*>(p + k);
      }
    } else if constexpr (GXH) {
      const _Float16* p = gxh_b + (long)t * G;
#pragma unroll
      for (int g = 0; g < 3; ++g) {
        if constexpr (UPL == 2) dst.v[g][0] = *reinterpret_cast<const h2*>(p + g * H);
        else dst.v[g][0] = p[g * H];
      }
    } else {
      const float* p = gx_b + (long)t * G;
#pragma unroll
      for (int g = 0; g < 3; ++g) ldu<UPL>(p + g * H, dst.v[g]);
    }
  };
  GX gxA, gxB;
  load_gx(gxA, 0);
  load_gx(gxB, 1);

  auto do_step = [&](const int step, const int cur, GX& gx) {
    const int t = dir ? T - 1 - step : step;

    f32x4 acc[3][UPL];
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
      for (int s = 0; s < UPL; ++s) acc[g][s] = f32x4{bh[g][s], 0.f, 0.f, 0.f};   // only register 0 is read
    [[maybe_unused]] f32x4 accx[3];
    if constexpr (XIN) {   // x_t W_ih^T + b_ih: independent of h, issued in front of the state fragments' LDS reads
#pragma unroll
      for (int g = 0; g < 3; ++g) accx[g] = f32x4{bi[g], 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 3; ++ks)
#pragma unroll
        for (int g = 0; g < 3; ++g) accx[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(gx.xf[ks], wx[g][ks], accx[g], 0, 0, 0);
    }
    if constexpr (BF16) {   // all A-fragments of the state tile requested up front (see gru_bwd_kernel)
      typename C::Frag sf[C::KS_F];
      if constexpr (!GSKIP(32)) {
#pragma unroll
      for (int ks = 0; ks < C::KS_F; ++ks) sf[ks] = state_frag(hs[cur], ks, lane);
      } else {
#pragma unroll
      for (int ks = 0; ks < C::KS_F; ++ks) sf[ks] = wr[0][0][ks];
      }
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (!GSKIP(1)) {
#pragma unroll
      for (int ks = 0; ks < C::KS_F; ++ks)
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
          for (int s = 0; s < UPL; ++s) acc[g][s] = mfma16(sf[ks], wr[g][s][ks], acc[g][s]);
      } else {
#pragma unroll
      for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int s = 0; s < UPL; ++s) acc[g][s][0] += (float)sf[g & 3][s];
      }
    } else {
#pragma unroll
      for (int ks = 0; ks < C::KS_F; ++ks) {
        const auto sf = state_frag(hs[cur], ks, lane);
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
          for (int s = 0; s < UPL; ++s) acc[g][s] = mfma16(sf, wr[g][s][ks], acc[g][s]);
      }
    }

    const bool valid = t < len;
    Gates<UPL> gt;
    float ho[UPL];
#pragma unroll
    for (int s = 0; s < UPL; ++s) {
      gt.hn[s] = acc[2][s][0];
      if constexpr (!GSKIP(2)) {
        gt.r[s] = fast_sigmoid((XIN ? accx[0][0] : gx.at(0, s)) + acc[0][s][0]);
        gt.z[s] = fast_sigmoid((XIN ? accx[1][0] : gx.at(1, s)) + acc[1][s][0]);
        gt.n[s] = fast_tanh((XIN ? accx[2][0] : gx.at(2, s)) + gt.r[s] * gt.hn[s]);
      } else {
        gt.r[s] = 0.25f * ((XIN ? accx[0][0] : gx.at(0, s)) + acc[0][s][0]);
        gt.z[s] = 0.25f * ((XIN ? accx[1][0] : gx.at(1, s)) + acc[1][s][0]);
        gt.n[s] = 0.5f * ((XIN ? accx[2][0] : gx.at(2, s)) + gt.r[s] * gt.hn[s]);
      }
      const float hnew = gt.n[s] + gt.z[s] * (hreg[s] - gt.n[s]);
      ho[s] = valid ? hnew : 0.f;
      hreg[s] = valid ? hnew : hreg[s];
    }
    if constexpr (!GSKIP(32)) putu<UPL>(hs[cur ^ 1], kq, u0, hreg);
    if constexpr (!GSKIP(4)) {
      if constexpr (!NO32) stu<UPL>(out_b + (long)t * a.out_ld, ho);
      if constexpr (H16) {
        if constexpr (UPL == 2) { typedef __attribute__((ext_vector_type(2))) _Float16 h2; h2 x; x[0] = to_f16_sat(ho[0]); x[1] = to_f16_sat(ho[1]); *reinterpret_cast<h2*>(out16_b + (long)t * a.out_ld) = x; }
        else out16_b[(long)t * a.out_ld] = to_f16_sat(ho[0]);
      }
      if constexpr (SAVE && XIN) {   // both lanes of a pair store the same 16 bytes (no guarded store in the loop)
        Gates<2> g2;
        const bool odd = (lane & 1) != 0;
        const float rn = dpp_take<0xB1>(0.f, gt.r[0]), zn = dpp_take<0xB1>(0.f, gt.z[0]), nn = dpp_take<0xB1>(0.f, gt.n[0]), hnn = dpp_take<0xB1>(0.f, gt.hn[0]);
        g2.r[0] = odd ? rn : gt.r[0];   g2.r[1] = odd ? gt.r[0] : rn;
        g2.z[0] = odd ? zn : gt.z[0];   g2.z[1] = odd ? gt.z[0] : zn;
        g2.n[0] = odd ? nn : gt.n[0];   g2.n[1] = odd ? gt.n[0] : nn;
        g2.hn[0] = odd ? hnn : gt.hn[0]; g2.hn[1] = odd ? gt.hn[0] : hnn;
        save_gates<BF16, 2>(sv_b + t * sv_step, g2);
      } else if constexpr (SAVE) save_gates<BF16, UPL>(sv_b + t * sv_step, gt);
    }
    if constexpr (!GSKIP(16)) lds_barrier(); else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // refill this buffer only now, behind the barrier (a compiler fence): its old contents are dead, so the loop-carried
    // value stays in the same registers.  Issued earlier, old and new values would be live together and the copy at the
    // loop back-edge would wait for the youngest load -- the whole latency back on the critical path.
    if constexpr (!GSKIP(8)) load_gx(gx, step + 2);
    asm volatile("" ::: "memory");   // ... and do not let the scheduler sink the loads towards their use either
  };
  // the first pair is peeled: the waitcnt bookkeeping at the loop header merges the counts of all incoming edges
  // conservatively, and the pre-loop path (two bare prefetches) would otherwise force near-zero counts -- i.e. waiting
  // for the youngest prefetch -- on every iteration
  int step = 0;
  if (T >= 2) {
    do_step(0, 0, gxA);
    do_step(1, 1, gxB);
    step = 2;
  }
  for (; step + 1 < T; step += 2) {
    do_step(step, 0, gxA);
    do_step(step + 1, 1, gxB);
  }
  if (step < T) do_step(step, 0, gxA);
  stamp_end(a.stamp);
}

// ------------------------------------------------------------------------------------------------
// backward through time.  Per step (processed in the reverse of the forward order):
//   dh   = dout[t] + carry                                  (valid rows; otherwise carry passes through)
//   dn = dh(1-z); dz = dh(h_prev - n); dn' = dn(1-n^2); dz' = dz z(1-z); dr' = dn' hn r(1-r)
//   dgx[t] = [dr', dz', dn']          (grad wrt x-side pre-activations, incl. b_ih)
//   dgh[t] = [dr', dz', dn' r]        (grad wrt h-side pre-activations, incl. b_hh)
//   carry  = dh z + dgh[t] . W_hh                            ([4,384].[384,128] on the matrix cores)
// dW_ih, dW_hh, biases and the gradient to the layer input are plain GEMMs over the stored dgx/dgh (engine).
// ------------------------------------------------------------------------------------------------
// __launch_bounds__(256, 2) for the bf16 instantiations: with (256, 1) the compiler parks 12 (backward) / 48 (forward) values in AGPRs,
// and next to THAT build a 223-VGPR kernel of another stream (kmix_bwd<MODE 2>) produced non-reproducible results (DESIGN.md section 5:
// 30 of 30 fresh engines exact with the AGPR-free build, ~70 % of them wrong with the other; mechanism not understood).  190 VGPRs, no
// AGPRs, no scratch, same speed.  The fp32 instantiations need more than 256 registers and keep (256, 1).  (UPL = 1: 512 threads, one
// workgroup per CU, at most 256 registers per lane by construction.)
// IO16 (round 5b; bf16 mode, UPL = 2 only): bit 0 -- dout is a bf16 array (GruBwdArgs::dout_bf16: its producer, the dh0 product or the
// LayerNorm backward, stored it that way); bit 1 -- h_prev comes from the forward kernel's fp16 copy of its outputs (GruSeqBwd::out16) and
// the fp32 outputs are not read (the forward launch may not even have written them).  Same NUMBER of loads per cell step, so the explicit
// wait counts do not change; half the bytes each.
template <bool BF16, bool DGBF, int UPL, int SKIP = 0, int IO16 = 0>
__global__ __launch_bounds__(512 / UPL, BF16 ? (UPL == 2 ? GRU_BF16_MINB : 1) : 1) void gru_bwd_kernel(GruBwdArgs a) {
  static_assert(IO16 == 0 || (BF16 && UPL == 2), "16-bit stored dout / h_prev: the 4-wave bf16 kernel only");
  using C = Cfg<BF16>;
  using F = typename Pack<UPL>::F;
  using GR = typename Pack<UPL>::G;
  using DOT = std::conditional_t<(IO16 & 1) != 0, PairBf, F>;
  using HPT = std::conditional_t<(IO16 & 2) != 0, PairH, F>;
  stamp_begin(a.stamp);
  GPH_DECL;
#ifdef MIMRL_PHASE_PROBE
  const long long gph_w0 = (long long)wall_clock64();
#endif
  __shared__ __attribute__((aligned(16))) Tile<BF16, G> ds[2];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int tile = blockIdx.x, dir = blockIdx.y, mod = blockIdx.z;
  const GruSeqBwd& q = a.seq[mod][dir];
  const int B = a.B, T = a.T, ntile = gridDim.x;
  const int n = lane & 15, kq = lane >> 4;
  const int u0 = 16 * UPL * w + UPL * n;
  // padding lanes mirror the last real row of the tile (see gru_fwd_kernel); only the bias sums must not count them
  const bool own = kq < a.btv && tile * a.btv + kq < B;
  const int b = min(tile * a.btv + min(kq, a.btv - 1), B - 1);
  const int len = a.lens[mod][b];

  // B fragments of W_hh (k = gate row 0..383, column = unit u0 + s)
  typename C::Frag wr[UPL][C::KS_B];
#pragma unroll
  for (int s = 0; s < UPL; ++s) {
#pragma unroll
    for (int ks = 0; ks < C::KS_B; ++ks) {
      if constexpr (BF16) {
        bf16x8 f;
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = to_bf16(q.w_hh[(long)(ks * 32 + 8 * kq + j) * H + u0 + s]);
        wr[s][ks] = f;
      } else {
        wr[s][ks] = q.w_hh[(long)(ks * 4 + kq) * H + u0 + s];
      }
    }
  }

  float carry[UPL];
  float sb[4][UPL];     // running bias-gradient sums of this lane's (unit, batch) slots: [dr', dz', dn', dn'*r]
#pragma unroll
  for (int e = 0; e < UPL; ++e) { carry[e] = 0.f; sb[0][e] = sb[1][e] = sb[2][e] = sb[3][e] = 0.f; }

  const long bb = b;
  // (byte pointers: the element size of the two operands depends on IO16)
  constexpr int HPB = (IO16 & 2) ? 2 : 4, DOB = (IO16 & 1) ? 2 : 4;
  const char* out_b = ((IO16 & 2) ? reinterpret_cast<const char*>(q.out16) : reinterpret_cast<const char*>(q.out)) +
                      (bb * T * a.out_ld + dir * H + u0) * HPB;             // forward outputs of THIS direction (h_prev source)
  const char* dout_b = reinterpret_cast<const char*>(q.dout) + (bb * T * a.dout_ld + a.dout_off * dir + u0) * DOB;
  const long dg_o = bb * (long)T * 4 * H + u0, hp_o = bb * (long)T * H + u0;   // element offsets (fp32 or bf16 elements: DGBF)
  const float* sv_b = q.saved + sv_index<BF16, UPL>(0, ntile, tile, w, lane);
  const long sv_step = (long)ntile * (8 / UPL) * 64 * SvRec<BF16, UPL>::F;
  // per-lane output bases in registers: with `q.dg` / `q.hprev` named inside the loop the compiler re-read them from the kernel arguments
  // on EVERY cell step (s_load_dwordx4 + s_waitcnt lgkmcnt(0) in front of the five stores: round-5 stamp budget)
  float* dg_lane = q.dg; float* hp_lane = q.hprev;
  asm volatile("" : "+v"(dg_lane), "+v"(hp_lane));

  // software pipeline, distance 2 (see gru_fwd_kernel): every load is unconditional and in bounds (padded steps read
  // stale-but-initialised records and are masked in the math).  h_prev is the previous VALID output of this direction;
  // with packed semantics that is simply out[tprev] when tprev is inside [0,len) and the zero initial state otherwise
  // (selected at use).
  struct Ops { Gates<UPL> g; GR graw; DOT DO; HPT HP; };
  auto fetch = [&](Ops& o, int step) {
    const int sc = step < T ? step : T - 1;
    const int t = dir ? sc : T - 1 - sc;
    const int tprev = dir ? t + 1 : t - 1;
    const int tc = tprev < 0 ? 0 : (tprev >= T ? T - 1 : tprev);
    if constexpr (BF16) {   // explicit-wait loads (see gld): the packed gate record is decoded behind the wait
      gld(o.graw, sv_b + t * sv_step);
      gld(o.DO, reinterpret_cast<const float*>(dout_b + (long)t * a.dout_ld * DOB));
      gld(o.HP, reinterpret_cast<const float*>(out_b + (long)tc * a.out_ld * HPB));
    } else {
      load_gates<BF16, UPL>(sv_b + t * sv_step, o.g);
      float x[UPL];
      ldu<UPL>(reinterpret_cast<const float*>(dout_b) + (long)t * a.dout_ld, x);
      if constexpr (UPL == 2) o.DO = make_float2(x[0], x[1]); else o.DO = x[0];
      ldu<UPL>(reinterpret_cast<const float*>(out_b) + (long)tc * a.out_ld, x);
      if constexpr (UPL == 2) o.HP = make_float2(x[0], x[1]); else o.HP = x[0];
    }
  };
  Ops opA, opB;

  // `wait` = integral_constant<int, N>: operations younger than this step's operands -- 8 in the steady state (the previous step's 5 stores
  // + 3 prefetches), 3 for the first step of the pipeline (the second buffer's prefetches), 0 for the un-pipelined single step of an odd
  // T; `pf` = false_type: no prefetch behind the barrier (single step)
  auto do_step = [&](const int step, const int cur, Ops& nx, auto wait, auto pf) {
    // forward visited t in order (dir ? T-1..0 : 0..T-1); backward walks it the other way round
    const int t = dir ? step : T - 1 - step;
    const bool valid = t < len;
    GPH(0);
    if constexpr (BF16) {
      vm_wait<decltype(wait)::value>(nx.graw, nx.DO, nx.HP);
      decode_gates<UPL>(nx.graw, nx.g);
    }
    GPH(1);
    const Ops& op = nx;
    float dhz[UPL], drp[UPL], dzp[UPL], dnp[UPL], dnr[UPL], hp[UPL];
    const int tprev = dir ? t + 1 : t - 1;
    const bool hp_ok = valid && tprev >= 0 && tprev < len;
#pragma unroll
    for (int e = 0; e < UPL; ++e) { drp[e] = dzp[e] = dnp[e] = dnr[e] = 0.f; hp[e] = hp_ok ? comp(op.HP, e) : 0.f; }
    if (valid && !GSKIP(2)) {
#pragma unroll
      for (int e = 0; e < UPL; ++e) {
        const float rr = op.g.r[e], zz = op.g.z[e], nn = op.g.n[e], hn = op.g.hn[e];
        const float dh = comp(op.DO, e) + carry[e];
        const float dn = dh * (1.f - zz);
        const float dz = dh * (hp[e] - nn);
        dnp[e] = dn * (1.f - nn * nn);
        dzp[e] = dz * zz * (1.f - zz);
        drp[e] = dnp[e] * hn * rr * (1.f - rr);
        dnr[e] = dnp[e] * rr;
        dhz[e] = dh * zz;
      }
    } else {
#pragma unroll
      for (int e = 0; e < UPL; ++e) dhz[e] = carry[e];
    }
    if constexpr (!GSKIP(32)) {
      putu<UPL>(ds[cur], kq, 0 * H + u0, drp);
      putu<UPL>(ds[cur], kq, 1 * H + u0, dzp);
      putu<UPL>(ds[cur], kq, 2 * H + u0, dnr);
    }
#pragma unroll
    for (int e = 0; e < UPL; ++e) { sb[0][e] += drp[e]; sb[1][e] += dzp[e]; sb[2][e] += dnp[e]; sb[3][e] += dnr[e]; }
#ifdef MIMRL_PHASE_PROBE
    asm volatile("" : "+v"(sb[0][0]), "+v"(sb[3][UPL - 1]), "+v"(dhz[0]));   // (the math has been ISSUED up to here)
#endif
    GPH(2);
    // (probe builds: the skipped stores / loads are replaced by the same NUMBER of cheap operations on one cache line, so that the
    //  explicit wait counts stay exact)
    constexpr bool sk4 = GSKIP(4), sk8 = GSKIP(8);
    stuo<DGBF, UPL>(hp_lane, sk4 ? hp_o : hp_o + (long)t * H, hp);
    const long dgt = sk4 ? dg_o : dg_o + (long)t * 4 * H;
    stuo<DGBF, UPL>(dg_lane, dgt + 0 * H, drp);
    stuo<DGBF, UPL>(dg_lane, dgt + (sk4 ? 0 : 1) * H, dzp);
    stuo<DGBF, UPL>(dg_lane, dgt + (sk4 ? 0 : 2) * H, dnp);
    stuo<DGBF, UPL>(dg_lane, dgt + (sk4 ? 0 : 3) * H, dnr);
    asm volatile("" ::: "memory");
    GPH(3);
    if constexpr (!GSKIP(16)) lds_barrier(); else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    GPH(4);
    if constexpr (decltype(pf)::value) fetch(nx, sk8 ? 0 : step + 2);   // behind the barrier: the old operands are dead (see gru_fwd_kernel)
    asm volatile("" ::: "memory");
    GPH(5);
    f32x4 acc[UPL];
#pragma unroll
    for (int s = 0; s < UPL; ++s) acc[s] = f32x4{dhz[s], 0.f, 0.f, 0.f};   // only register 0 is read
    if constexpr (BF16) {
      // ALL twelve A-fragments of the dgh tile are requested before the first product (12 x 16 B per lane, distinct registers).  Left to
      // itself the compiler reads every fragment into ONE register quad -- ds_read, s_waitcnt lgkmcnt(0), two MFMAs, twelve times: an LDS
      // round trip per k-step on the dependent chain of every cell step.  __builtin_amdgcn_sched_barrier keeps the scheduler from sinking
      // the reads back to their uses (an empty asm with a memory clobber does not).
      typename C::Frag sf[C::KS_B];
      if constexpr (!GSKIP(32)) {
#pragma unroll
      for (int ks = 0; ks < C::KS_B; ++ks) sf[ks] = state_frag(ds[cur], ks, lane);
      } else {
#pragma unroll
      for (int ks = 0; ks < C::KS_B; ++ks) sf[ks] = wr[0][ks];
      }
      __builtin_amdgcn_sched_barrier(0);
#ifdef MIMRL_PHASE_PROBE
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (the twelve fragments are in registers)
      __builtin_amdgcn_sched_barrier(0);
#endif
      GPH(6);
      if constexpr (!GSKIP(1)) {
#pragma unroll
      for (int ks = 0; ks < C::KS_B; ++ks)
#pragma unroll
        for (int s = 0; s < UPL; ++s) acc[s] = mfma16(sf[ks], wr[s][ks], acc[s]);
      } else {
#pragma unroll
      for (int s = 0; s < UPL; ++s) acc[s][0] += (float)sf[s][0];
      }
    } else {
#pragma unroll
      for (int ks = 0; ks < C::KS_B; ++ks) {
        const auto sf = state_frag(ds[cur], ks, lane);
#pragma unroll
        for (int s = 0; s < UPL; ++s) acc[s] = mfma16(sf, wr[s][ks], acc[s]);
      }
    }
#pragma unroll
    for (int s = 0; s < UPL; ++s) carry[s] = acc[s][0];
#ifdef MIMRL_PHASE_PROBE
    if constexpr (BF16) {
      asm volatile("v_mov_b32 %0, %0\n\ts_nop 0" : "+v"(carry[0]), "+v"(carry[UPL - 1]));   // (issues only once the last MFMA has retired)
      __builtin_amdgcn_sched_barrier(0);
      GPH(7);
      GPH_SUM();
    }
#endif
    // ds[cur] is rewritten two steps from now; the barrier of the next step orders that write after these reads
  };
  // Round 5, measured and NOT kept: the cell step re-cut around its dependent chain (carry -> dh -> five products -> tile -> barrier ->
  // fragments -> MFMAs), with everything else -- decode, the gate derivatives folded into per-unit coefficients one step ahead, the five
  // stores, bias sums, prefetch and operand wait -- moved behind the barrier / into the shadow of the 24 MFMAs.  The compiler interleaved
  // it as intended (stores and coefficient math between the products) and the bare chain alone runs at 0.52 us per step (26 us per launch
  // with the stores, prefetch and coefficients compiled out), but the full kernel took 46.2 us against 44.4 (three alternating runs):
  // one wave per SIMD issues every instruction of the step itself, the "shadow" work still takes its issue slots (an MFMA frees 8 of its
  // 16 cycles), and the coefficient form costs ~20 more VALU instructions than the direct one.  A second, mask-free copy of the loop for
  // full-length batches made the compiler re-home in-flight operand registers (tools/isa_inflight.py caught it before a GPU test did).
  // What the stamp budget (tools/gru_bwd_phase.py, profiles/r05_gru_bwd_phase.json) did give: the store phase contained a kernel-argument
  // reload -- s_load_dwordx4 + lgkmcnt(0) for q.hprev / q.dg on every step; the per-lane store bases now live in registers.
  // An ODD T runs one un-pipelined step FIRST (operands fetched, waited for with vmcnt(0), no prefetch), then the even remainder through
  // the pipeline -- never a single step BEHIND the loop.  Round 3 had that tail, and in its code the compiler re-homed the loop-carried
  // operand registers with v_mov copies placed IN FRONT of the explicit wait, i.e. it copied the destination of an asm load that was still
  // in flight (the loads are invisible to it by design): the last cell step of every odd-T sequence read stale gates, the BPTT gradients
  // were off by ~10 % and different from run to run (found in round 4 by the odd-T case of test_gradients_reproducible, which ADVICE r03
  // asked for; even T never ran that code).  Now the only place where in-flight asm destinations cross a block boundary is the loop's own
  // back edge, and tests/test_codeobj.py checks the ISA there (no instruction touches an asm-load destination between its load and its wait).
  using W0 = std::integral_constant<int, 0>; using W3 = std::integral_constant<int, 3>; using W8 = std::integral_constant<int, 8>;
  int step = 0;
  if (T & 1) {
    fetch(opA, 0);
    do_step(0, 0, opA, W0{}, std::false_type{});
    step = 1;
  }
  const int c0 = step;      // LDS tile of a step = step & 1 (a tile is rewritten two steps after it was read)
  if (step < T) {           // (T - step is even: whole pairs)
    fetch(opA, step);
    fetch(opB, step + 1);
    do_step(step, c0, opA, W3{}, std::true_type{});       // peeled first pair (see gru_fwd_kernel)
    do_step(step + 1, c0 ^ 1, opB, W8{}, std::true_type{});
    step += 2;
  }
  for (; step < T; step += 2) {
    do_step(step, c0, opA, W8{}, std::true_type{});
    do_step(step + 1, c0 ^ 1, opB, W8{}, std::true_type{});
  }
  // DRAIN the asm prefetches: the last two steps still issued fetch(nx, step + 2) -- six loads the compiler knows nothing about, into
  // registers it considers dead and is free to reuse (address arithmetic, the data of the bias-gradient atomics below) while they
  // are still in flight.  One vmcnt(0) behind the loop, in front of anything else; the ties keep the registers reserved.
  if constexpr (BF16) asm volatile("s_waitcnt vmcnt(0)" : "+v"(opA.graw), "+v"(opA.DO), "+v"(opA.HP), "+v"(opB.graw), "+v"(opB.DO), "+v"(opB.HP) :: "memory");
  // bias gradients: reduce over the 4 batch rows (lanes differing in bits 4..5), one atomic per unit
  if (q.db_ih || q.db_hh) {
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int e = 0; e < UPL; ++e) {
        float v = own ? sb[g][e] : 0.f;
        v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
        sb[g][e] = v;
      }
    if (kq == 0) {
#pragma unroll
      for (int e = 0; e < UPL; ++e) {
        const int unit = u0 + e;
        if (q.db_ih) {
          acc_add(&q.db_ih[0 * H + unit], sb[0][e]); acc_add(&q.db_ih[1 * H + unit], sb[1][e]);
          acc_add(&q.db_ih[2 * H + unit], sb[2][e]);
        }
        if (q.db_hh) {
          acc_add(&q.db_hh[0 * H + unit], sb[0][e]); acc_add(&q.db_hh[1 * H + unit], sb[1][e]);
          acc_add(&q.db_hh[2 * H + unit], sb[3][e]);
        }
      }
    }
  }
#ifdef MIMRL_PHASE_PROBE
  if (BF16 && DGBF && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) {
    const int o = a.dout_off == 0 ? 0 : 16;      // layer 1 (dout = [B, T, H] slices) / layer 0
#pragma unroll
    for (int i = 0; i < 8; ++i) g_gru_bwd_phase[o + i] = gph_s[i];
    g_gru_bwd_phase[o + 8] = gph_w0; g_gru_bwd_phase[o + 9] = (long long)wall_clock64();
  }
#endif
  stamp_end(a.stamp);
}

}  // namespace

#ifdef MIMRL_PHASE_PROBE
int gru_bwd_read_phases(long long* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_gru_bwd_phase), sizeof(long long) * 32) == hipSuccess ? 0 : 1; }
#endif

// waves per workgroup of the bf16 recurrence kernels: 4 (two units per lane); MIMRL_GRU_WAVES=8: one unit per lane, two waves per SIMD.
// Round 4 measured both at cfg2 (interleaved runs, in-graph launch stamps): forward 28.7 (8 waves) vs 30.0 us (4), BPTT 47.6 vs 44.0 us,
// step 0.855-0.858 vs 0.853-0.863 ms -- the 8-wave BPTT loses more than the forward gains, and the phase elimination of the forward
// kernel (tools/gru_phase.sh, profiles/r04_gru_phase.json) says why the halved per-wave work buys so little: of 0.62 us per cell step the
// 24 products per SIMD are 0.16 us (the matrix pipe's floor: 24 x 16 cycles, whatever the wave count), the LDS tile round trip 0.08, the
// barrier 0.06, the stores 0.06, the six transcendentals 0.04, the operand loads 0.02 -- and 0.2 us are neither (address / select / wait
// instructions of the step itself).  Default stays 4.
static int gru_upl() {
  static const int waves = knob("MIMRL_GRU_WAVES") ? atoi(knob("MIMRL_GRU_WAVES")) : 4;   // tuning knob
  return waves == 8 ? 1 : 2;
}

static int gru_skip() {
#ifdef MIMRL_PHASE_PROBE
  static const int v = knob("MIMRL_GRU_SKIP") ? atoi(knob("MIMRL_GRU_SKIP")) : 0;
  return v;
#else
  return 0;
#endif
}
void gru_probe_setup() {}

#ifdef MIMRL_PHASE_PROBE
// probe build: one instantiation per mask of tools/gru_phase.sh (bf16, saved gates, bf16 dg: the benchmarked kernels)
template <int UPL, int SKIP>
static void probe_fwd(dim3 grid, hipStream_t s, const GruFwdArgs& a) { hipLaunchKernelGGL((gru_fwd_kernel<true, true, UPL, SKIP>), grid, dim3(512 / UPL), 0, s, a); }
template <int UPL, int SKIP>
static void probe_bwd(dim3 grid, hipStream_t s, const GruBwdArgs& a, size_t pad) { hipLaunchKernelGGL((gru_bwd_kernel<true, true, UPL, SKIP>), grid, dim3(512 / UPL), pad, s, a); }
#define GRU_PROBE_CASES(X) X(1) X(2) X(4) X(8) X(16) X(32) X(3)
#endif

int gru_forward(hipStream_t s, const GruFwdArgs& a, bool bf16) {
  if (a.B <= 0 || a.T <= 0) return set_error(MIMRL_ERR_ARG, "gru_forward: empty batch");
  if (a.btv < 1 || a.btv > BR) return set_error(MIMRL_ERR_ARG, "gru_forward: btv must be in [1,4]");
  dim3 grid((a.B + a.btv - 1) / a.btv, 2, a.nmod);
  bool save = a.seq[0][0].saved != nullptr;
  for (int m = 0; m < a.nmod; ++m)
    for (int d = 0; d < 2; ++d)
      if ((a.seq[m][d].saved != nullptr) != save) return set_error(MIMRL_ERR_ARG, "gru_forward: saved slabs must be all set or all null");
#ifdef MIMRL_PHASE_PROBE
  if (bf16 && save && gru_skip()) {
#define X(M) if (gru_skip() == M) { if (gru_upl() == 1) probe_fwd<1, M>(grid, s, a); else probe_fwd<2, M>(grid, s, a); }
    GRU_PROBE_CASES(X)
#undef X
    LAUNCH_CHECK();
    return MIMRL_OK;
  }
#endif
  if (a.gx_f16 && !bf16) return set_error(MIMRL_ERR_ARG, "gru_forward: fp16-stored gx needs the bf16 recurrence mode");
  const bool h16 = a.seq[0][0].out16 != nullptr;
  for (int m = 0; m < a.nmod; ++m)
    for (int d = 0; d < 2; ++d)
      if ((a.seq[m][d].out16 != nullptr) != h16) return set_error(MIMRL_ERR_ARG, "gru_forward: fp16 output copies must be all set or all null");
  if (h16 && !bf16) return set_error(MIMRL_ERR_ARG, "gru_forward: the fp16 output copy exists in the bf16 recurrence mode only");
  if (a.no_out32 && !a.xin_on) return set_error(MIMRL_ERR_ARG, "gru_forward: no_out32 exists for the fused-projection kernel only");
  if (a.xin_on) {
    if (!bf16 || a.gx_f16 || a.kp % 8 != 0 || a.kp > 96 || a.kp < 8) return set_error(MIMRL_ERR_ARG, "gru_forward: the fused input projection needs the bf16 mode and kp in 8..96, a multiple of 8");
    if (a.no_out32 && !(h16 && save)) return set_error(MIMRL_ERR_ARG, "gru_forward: no_out32 needs the fp16 output copy and the saved-gate slab");
    if (h16 && a.no_out32) hipLaunchKernelGGL((gru_fwd_kernel<true, true, 1, 0, false, true, true, true>), grid, dim3(512), 0, s, a);
    else if (h16) {
      if (save) hipLaunchKernelGGL((gru_fwd_kernel<true, true, 1, 0, false, true, true>), grid, dim3(512), 0, s, a);
      else hipLaunchKernelGGL((gru_fwd_kernel<true, false, 1, 0, false, true, true>), grid, dim3(512), 0, s, a);
    } else {
      if (save) hipLaunchKernelGGL((gru_fwd_kernel<true, true, 1, 0, false, false, true>), grid, dim3(512), 0, s, a);
      else hipLaunchKernelGGL((gru_fwd_kernel<true, false, 1, 0, false, false, true>), grid, dim3(512), 0, s, a);
    }
  } else if (bf16 && a.gx_f16 && h16) {
    if (gru_upl() == 1) {
      if (save) hipLaunchKernelGGL((gru_fwd_kernel<true, true, 1, 0, true, true>), grid, dim3(512), 0, s, a);
      else hipLaunchKernelGGL((gru_fwd_kernel<true, false, 1, 0, true, true>), grid, dim3(512), 0, s, a);
    } else {
      if (save) hipLaunchKernelGGL((gru_fwd_kernel<true, true, 2, 0, true, true>), grid, dim3(256), 0, s, a);
      else hipLaunchKernelGGL((gru_fwd_kernel<true, false, 2, 0, true, true>), grid, dim3(256), 0, s, a);
    }
  } else if (bf16 && a.gx_f16) {
    if (gru_upl() == 1) {
      if (save) hipLaunchKernelGGL((gru_fwd_kernel<true, true, 1, 0, true>), grid, dim3(512), 0, s, a);
      else hipLaunchKernelGGL((gru_fwd_kernel<true, false, 1, 0, true>), grid, dim3(512), 0, s, a);
    } else {
      if (save) hipLaunchKernelGGL((gru_fwd_kernel<true, true, 2, 0, true>), grid, dim3(256), 0, s, a);
      else hipLaunchKernelGGL((gru_fwd_kernel<true, false, 2, 0, true>), grid, dim3(256), 0, s, a);
    }
  } else if (bf16 && h16) {
    if (gru_upl() == 1) {
      if (save) hipLaunchKernelGGL((gru_fwd_kernel<true, true, 1, 0, false, true>), grid, dim3(512), 0, s, a);
      else hipLaunchKernelGGL((gru_fwd_kernel<true, false, 1, 0, false, true>), grid, dim3(512), 0, s, a);
    } else {
      if (save) hipLaunchKernelGGL((gru_fwd_kernel<true, true, 2, 0, false, true>), grid, dim3(256), 0, s, a);
      else hipLaunchKernelGGL((gru_fwd_kernel<true, false, 2, 0, false, true>), grid, dim3(256), 0, s, a);
    }
  } else if (bf16) {
    if (gru_upl() == 1) {
      if (save) hipLaunchKernelGGL((gru_fwd_kernel<true, true, 1>), grid, dim3(512), 0, s, a);
      else hipLaunchKernelGGL((gru_fwd_kernel<true, false, 1>), grid, dim3(512), 0, s, a);
    } else {
      if (save) hipLaunchKernelGGL((gru_fwd_kernel<true, true, 2>), grid, dim3(256), 0, s, a);
      else hipLaunchKernelGGL((gru_fwd_kernel<true, false, 2>), grid, dim3(256), 0, s, a);
    }
  } else {
    if (save) hipLaunchKernelGGL((gru_fwd_kernel<false, true, 2>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((gru_fwd_kernel<false, false, 2>), grid, dim3(256), 0, s, a);
  }
  LAUNCH_CHECK();
  return MIMRL_OK;
}

int gru_backward(hipStream_t s, const GruBwdArgs& a, bool bf16) {
  if (a.B <= 0 || a.T <= 0) return set_error(MIMRL_ERR_ARG, "gru_backward: empty batch");
  if (a.btv < 1 || a.btv > BR) return set_error(MIMRL_ERR_ARG, "gru_backward: btv must be in [1,4]");
  dim3 grid((a.B + a.btv - 1) / a.btv, 2, a.nmod);
  if (a.dg_bf16 && !bf16) return set_error(MIMRL_ERR_ARG, "gru_backward: bf16 dg / h_prev storage needs the bf16 recurrence mode");
  // LDS padding (round 3b): a BPTT launch that covers at most half of the CUs (cfg2: 128 workgroups) asks for 144 KiB of dynamic LDS it never
  // touches, so that no LDS-using kernel parked beside the recurrence (the weight-gradient GEMMs) becomes resident on ITS CUs and shares
  // its SIMDs: alone the two launches take 45 + 48 us, with the parked kernels on the same CUs 51 + 66 us.  cfg2: 0.916-0.917 -> 0.907-0.911 ms
  // per step.  Larger launches keep their CUs shareable (at cfg3 every CU has a BPTT workgroup and the parked work needs a place).
  // MIMRL_GRU_LDS_PAD=<KiB> overrides (0 = off).
  static const int pad_env = knob("MIMRL_GRU_LDS_PAD") ? atoi(knob("MIMRL_GRU_LDS_PAD")) : -1;
  const int pad_kb = pad_env >= 0 ? pad_env : ((long)grid.x * grid.y * grid.z <= 128 ? 144 : 0);
  const size_t pad = bf16 && a.dg_bf16 && pad_kb > 0 ? (size_t)(pad_kb > 150 ? 150 : pad_kb) * 1024 : 0;
#ifdef MIMRL_PHASE_PROBE
  if (bf16 && a.dg_bf16 && gru_skip()) {
#define X(M) if (gru_skip() == M) { if (gru_upl() == 1) probe_bwd<1, M>(grid, s, a, 0); else probe_bwd<2, M>(grid, s, a, 0); }
    GRU_PROBE_CASES(X)
#undef X
    LAUNCH_CHECK();
    return MIMRL_OK;
  }
#endif
  const int upl = a.slab_upl ? a.slab_upl : gru_upl();
  const bool o16 = a.seq[0][0].out16 != nullptr;
  for (int m = 0; m < a.nmod; ++m)
    for (int d = 0; d < 2; ++d)
      if ((a.seq[m][d].out16 != nullptr) != o16) return set_error(MIMRL_ERR_ARG, "gru_backward: fp16 output copies must be all set or all null");
  const int io16 = (a.dout_bf16 ? 1 : 0) | (o16 ? 2 : 0);
  if (io16 && !(bf16 && a.dg_bf16 && upl == 2)) return set_error(MIMRL_ERR_ARG, "gru_backward: 16-bit stored dout / h_prev need the bf16 mode with bf16 dg and the 4-wave slab layout");
  if (bf16 && upl == 1) {
    auto k1 = gru_bwd_kernel<true, true, 1>;
    auto k0 = gru_bwd_kernel<true, false, 1>;
    if (pad) {
      static bool attr = false;
      if (!attr) { HIPX(hipFuncSetAttribute(reinterpret_cast<const void*>(k1), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024)); attr = true; }
    }
    if (a.dg_bf16) hipLaunchKernelGGL(k1, grid, dim3(512), pad, s, a);
    else hipLaunchKernelGGL(k0, grid, dim3(512), 0, s, a);
  } else if (bf16 && io16) {
    // 16-bit stored dout and / or h_prev from the forward's fp16 copy (IO16: see the kernel)
    typedef void (*K)(GruBwdArgs);
    const K k = io16 == 1 ? static_cast<K>(gru_bwd_kernel<true, true, 2, 0, 1>) : io16 == 2 ? static_cast<K>(gru_bwd_kernel<true, true, 2, 0, 2>) : static_cast<K>(gru_bwd_kernel<true, true, 2, 0, 3>);
    if (pad) {
      static bool attr[4] = {false, false, false, false};
      if (!attr[io16]) { HIPX(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024)); attr[io16] = true; }
    }
    hipLaunchKernelGGL(k, grid, dim3(256), pad, s, a);
  } else if (bf16) {
    auto k1 = gru_bwd_kernel<true, true, 2>;
    if (pad) {
      static bool attr = false;
      if (!attr) { HIPX(hipFuncSetAttribute(reinterpret_cast<const void*>(k1), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024)); attr = true; }
    }
    if (a.dg_bf16) hipLaunchKernelGGL(k1, grid, dim3(256), pad, s, a);
    else hipLaunchKernelGGL((gru_bwd_kernel<true, false, 2>), grid, dim3(256), 0, s, a);
  } else {
    hipLaunchKernelGGL((gru_bwd_kernel<false, false, 2>), grid, dim3(256), 0, s, a);
  }
  LAUNCH_CHECK();
  return MIMRL_OK;
}

// may a BPTT launch that reads a slab of layout `slab_upl` (GruBwdArgs::slab_upl) take 16-bit stored dout / h_prev?  (the engine decides
// what the producers write before the launch exists)
bool gru_bwd_io16_ok(int slab_upl) { return (slab_upl ? slab_upl : gru_upl()) == 2 && gru_skip() == 0; }

long gru_saved_floats(int B, int T) {
  return (long)T * B * 4 * 64 * 8;   // worst case: one batch row per workgroup (btv = 1)
}

int gru_pick_btv(int B, int nmod) {
  constexpr int force = 0;   // (an environment knob until round 5: fixed at its measured optimum)
  if (force >= 1 && force <= BR) return force;
  int btv = (B * nmod * 2 + 127) / 128;      // ~128 workgroups (measured best at B=128: 4 rows per workgroup)
  if (btv < 1) btv = 1;
  if (btv > BR) btv = BR;
  return btv;
}

}  // namespace mimrl
