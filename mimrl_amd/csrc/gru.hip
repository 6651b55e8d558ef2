// Persistent bidirectional-GRU recurrence kernels (forward and BPTT) for gfx950.
//
// Reference semantics: nn.GRU(d,128,num_layers=2,bidirectional) fed with packed sequences (Model.py:254-255,
// 441-447): gate order r,z,n; n = tanh(gx_n + r*(W_hn h + b_hn)); positions t >= length emit 0 and do not advance
// the state; the reverse direction therefore effectively starts at t = length-1.
//
// MI355X design.  The recurrence is the latency floor of the whole training step: 2 layers x T
// dependent steps per pass.  It is independent across batch rows, so one workgroup owns a 16-row batch tile of one
// (modality, direction) for all T steps -- no inter-workgroup traffic at all.  Inside the workgroup the four
// waves (one per SIMD) split the 128 hidden units 32/32/32/32; each wave keeps its slice of W_hh (all three
// gates) in REGISTERS as ready-made MFMA A-fragments for the whole sequence (bf16: 96 VGPRs, fp32: 192 VGPRs),
// so the only per-step shared-memory traffic is the 16x128 state tile (double-buffered in LDS, ONE barrier per
// step).  gates^T[unit, batch] = W_hh[unit, :] . h^T[:, batch] puts the batch on the MFMA lane axis and the
// r/z/n values of one (unit, batch) pair in the same lane/register slot, so the gate math is register-only.
// The input projections gx = x W_ih^T + b_ih are hoisted out of the loop into one GEMM per layer (gemm.hip).
//
// fp32 mode: v_mfma_f32_16x16x4_f32 (bit-exact fp32 fma chains) -- parity mode.
// bf16 mode: v_mfma_f32_16x16x32_bf16, W_hh and the state tile rounded to bf16 for the product, fp32 state,
//            fp32 accumulate, fp32 gate math.
#include "gru.h"

#include <cstdlib>

namespace mimrl {

namespace {

constexpr int H = 128;       // hidden size (= d_common; the reference only runs with 128)
constexpr int G = 3 * H;
constexpr int BT = 16;       // batch rows per workgroup (MFMA N)
// 4 waves per workgroup; wave w owns units [32w, 32w+32)

template <bool BF16>
struct Cfg;
template <>
struct Cfg<true> {
  static constexpr int KS_F = H / 32;   // k-steps over H (forward product)
  static constexpr int KS_B = G / 32;   // k-steps over 3H (backward product)
  using AFrag = bf16x8;
};
template <>
struct Cfg<false> {
  static constexpr int KS_F = H / 4;
  static constexpr int KS_B = G / 4;
  using AFrag = float;
};

// LDS state tile.  bf16: [batch][k] rows of (K+16) bf16 (288-B / 800-B rows -> conflict-free ds_read_b128);
//                  fp32: [k][batch] (k-major: a wave's 64 lanes read 64 consecutive words).
template <bool BF16, int K>
struct Tile;
template <int K>
struct Tile<true, K> {
  __bf16 v[BT][K + 16];
};
template <int K>
struct Tile<false, K> {
  float v[K][BT];
};

__device__ __forceinline__ f32x4 mfma16(const bf16x8& a, const bf16x8& b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma16(const float& a, const float& b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// B-fragment (state^T) of k-step ks for this lane
template <int K>
__device__ __forceinline__ bf16x8 bfrag(const Tile<true, K>& t, int ks, int lane) {
  return *reinterpret_cast<const bf16x8*>(&t.v[lane & 15][ks * 32 + 8 * (lane >> 4)]);
}
template <int K>
__device__ __forceinline__ float bfrag(const Tile<false, K>& t, int ks, int lane) {
  return t.v[ks * 4 + (lane >> 4)][lane & 15];
}
// write 4 consecutive "k" values (k0..k0+3) of batch column b
template <int K>
__device__ __forceinline__ void put4(Tile<true, K>& t, int b, int k0, float x0, float x1, float x2, float x3) {
  bf16x4 p;
  p[0] = to_bf16(x0); p[1] = to_bf16(x1); p[2] = to_bf16(x2); p[3] = to_bf16(x3);
  *reinterpret_cast<bf16x4*>(&t.v[b][k0]) = p;
}
template <int K>
__device__ __forceinline__ void put4(Tile<false, K>& t, int b, int k0, float x0, float x1, float x2, float x3) {
  t.v[k0][b] = x0; t.v[k0 + 1][b] = x1; t.v[k0 + 2][b] = x2; t.v[k0 + 3][b] = x3;
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt(0), i.e. it would wait every step
// for the saved-gate / output stores and for the prefetched gx loads -- the whole point of the prefetch is to keep
// them in flight across the step boundary.  LDS operations of a wave complete in order, so lgkmcnt(0) before
// s_barrier makes this wave's state-tile writes visible to the other waves after the barrier.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float a, float b, float c, float d) {
  *reinterpret_cast<float4*>(p) = make_float4(a, b, c, d);
}

// saved-gate slab: fp32 mode stores r,z,n,hn as four float4 per (lane, s); bf16 mode packs them into two 16-byte
// vectors [r|z] and [n|hn] of bf16 (half the store / load instructions of the two recurrence kernels).
template <bool BF16>
__device__ __forceinline__ void save_gates(float* base, int t, int ntile, int tile, int w, int s, int lane, const float* rr,
                                           const float* zz, const float* nn, const float* hn);
template <bool BF16>
__device__ __forceinline__ void load_gates(const float* base, int t, int ntile, int tile, int w, int s, int lane, float* rr,
                                           float* zz, float* nn, float* hn);

// saved-gate slab addressing ("lane-native": every wave-instruction stores 1 KiB contiguous)
//   index = ((((t*ntile + tile)*4 + q)*8 + (w*2+s))*64 + lane)*4      q: 0=r 1=z 2=n 3=hn
__device__ __forceinline__ long sv_index(int t, int ntile, int tile, int q, int w, int s, int lane) {
  return ((((long)t * ntile + tile) * 4 + q) * 8 + (w * 2 + s)) * 256 + lane * 4;   // 16 batch columns per tile slot
}

template <>
__device__ __forceinline__ void save_gates<false>(float* base, int t, int ntile, int tile, int w, int s, int lane,
                                                  const float* rr, const float* zz, const float* nn, const float* hn) {
  st4(base + sv_index(t, ntile, tile, 0, w, s, lane), rr[0], rr[1], rr[2], rr[3]);
  st4(base + sv_index(t, ntile, tile, 1, w, s, lane), zz[0], zz[1], zz[2], zz[3]);
  st4(base + sv_index(t, ntile, tile, 2, w, s, lane), nn[0], nn[1], nn[2], nn[3]);
  st4(base + sv_index(t, ntile, tile, 3, w, s, lane), hn[0], hn[1], hn[2], hn[3]);
}
template <>
__device__ __forceinline__ void save_gates<true>(float* base, int t, int ntile, int tile, int w, int s, int lane,
                                                 const float* rr, const float* zz, const float* nn, const float* hn) {
  bf16x8 a, b;
#pragma unroll
  for (int r = 0; r < 4; ++r) { a[r] = to_bf16(rr[r]); a[4 + r] = to_bf16(zz[r]); b[r] = to_bf16(nn[r]); b[4 + r] = to_bf16(hn[r]); }
  *reinterpret_cast<bf16x8*>(base + sv_index(t, ntile, tile, 0, w, s, lane)) = a;
  *reinterpret_cast<bf16x8*>(base + sv_index(t, ntile, tile, 1, w, s, lane)) = b;
}
template <>
__device__ __forceinline__ void load_gates<false>(const float* base, int t, int ntile, int tile, int w, int s, int lane,
                                                  float* rr, float* zz, float* nn, float* hn) {
  const float4 R = ld4(base + sv_index(t, ntile, tile, 0, w, s, lane)), Z = ld4(base + sv_index(t, ntile, tile, 1, w, s, lane));
  const float4 N = ld4(base + sv_index(t, ntile, tile, 2, w, s, lane)), Hn = ld4(base + sv_index(t, ntile, tile, 3, w, s, lane));
  rr[0] = R.x; rr[1] = R.y; rr[2] = R.z; rr[3] = R.w; zz[0] = Z.x; zz[1] = Z.y; zz[2] = Z.z; zz[3] = Z.w;
  nn[0] = N.x; nn[1] = N.y; nn[2] = N.z; nn[3] = N.w; hn[0] = Hn.x; hn[1] = Hn.y; hn[2] = Hn.z; hn[3] = Hn.w;
}
template <>
__device__ __forceinline__ void load_gates<true>(const float* base, int t, int ntile, int tile, int w, int s, int lane,
                                                 float* rr, float* zz, float* nn, float* hn) {
  const bf16x8 a = *reinterpret_cast<const bf16x8*>(base + sv_index(t, ntile, tile, 0, w, s, lane));
  const bf16x8 b = *reinterpret_cast<const bf16x8*>(base + sv_index(t, ntile, tile, 1, w, s, lane));
#pragma unroll
  for (int r = 0; r < 4; ++r) { rr[r] = (float)a[r]; zz[r] = (float)a[4 + r]; nn[r] = (float)b[r]; hn[r] = (float)b[4 + r]; }
}

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
template <bool BF16>
__global__ __launch_bounds__(256, 1) void gru_fwd_kernel(GruFwdArgs a) {
  using C = Cfg<BF16>;
  __shared__ __attribute__((aligned(16))) Tile<BF16, H> hs[2];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int tile = blockIdx.x, dir = blockIdx.y, mod = blockIdx.z;
  const GruSeq& q = a.seq[mod][dir];
  const int B = a.B, T = a.T, ntile = gridDim.x;
  const int bcol = lane & 15, kq = lane >> 4;
  const int b = tile * a.btv + bcol;
  const bool brow_ok = bcol < a.btv && b < B;
  const int len = brow_ok ? a.lens[mod][b] : 0;

  // ---- W_hh slice -> registers as MFMA A fragments: rows (gate g, unit 32w+16s+i), i = lane&15
  typename C::AFrag wr[3][2][C::KS_F];
#pragma unroll
  for (int g = 0; g < 3; ++g)
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const float* row = q.w_hh + (long)(g * H + 32 * w + 16 * s + (lane & 15)) * H;
#pragma unroll
      for (int ks = 0; ks < C::KS_F; ++ks) {
        if constexpr (BF16) {
          const float4 lo = ld4(row + ks * 32 + 8 * kq), hi = ld4(row + ks * 32 + 8 * kq + 4);
          bf16x8 f;
          f[0] = to_bf16(lo.x); f[1] = to_bf16(lo.y); f[2] = to_bf16(lo.z); f[3] = to_bf16(lo.w);
          f[4] = to_bf16(hi.x); f[5] = to_bf16(hi.y); f[6] = to_bf16(hi.z); f[7] = to_bf16(hi.w);
          wr[g][s][ks] = f;
        } else {
          wr[g][s][ks] = row[ks * 4 + kq];
        }
      }
    }
  // b_hh for this lane's output rows: unit = 32w + 16s + 4kq + r
  float bh[3][2][4];
#pragma unroll
  for (int g = 0; g < 3; ++g)
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const float4 v = ld4(q.b_hh + g * H + 32 * w + 16 * s + 4 * kq);
      bh[g][s][0] = v.x; bh[g][s][1] = v.y; bh[g][s][2] = v.z; bh[g][s][3] = v.w;
    }

  // ---- state
  float hreg[2][4];
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int r = 0; r < 4; ++r) hreg[s][r] = 0.f;
#pragma unroll
  for (int s = 0; s < 2; ++s) put4(hs[0], bcol, 32 * w + 16 * s + 4 * kq, 0.f, 0.f, 0.f, 0.f);
  __syncthreads();

  const long row_stride_gx = (long)T * G;          // gx [B,T,3H]
  const long row_stride_out = (long)T * a.out_ld;  // out [B,T,out_ld]
  const float* gx_b = q.gx + (long)(brow_ok ? b : 0) * row_stride_gx;
  float* out_b = q.out + (long)(brow_ok ? b : 0) * row_stride_out;

  float4 gxn[3][2];   // prefetched gx of the next step
  auto load_gx = [&](int t) {
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
      for (int s = 0; s < 2; ++s)
        gxn[g][s] = brow_ok ? ld4(gx_b + (long)t * G + g * H + 32 * w + 16 * s + 4 * kq) : make_float4(0.f, 0.f, 0.f, 0.f);
  };
  load_gx(dir ? T - 1 : 0);

  for (int step = 0; step < T; ++step) {
    const int t = dir ? T - 1 - step : step;
    const int cur = step & 1;
    float4 gxc[3][2];
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
      for (int s = 0; s < 2; ++s) gxc[g][s] = gxn[g][s];
    if (step + 1 < T && !(a.dbg & 4)) load_gx(dir ? T - 2 - step : step + 1);

    f32x4 acc[3][2];
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
      for (int s = 0; s < 2; ++s) acc[g][s] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (!(a.dbg & 2))
#pragma unroll
    for (int ks = 0; ks < C::KS_F; ++ks) {
      const auto bf = bfrag(hs[cur], ks, lane);
#pragma unroll
      for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int s = 0; s < 2; ++s) acc[g][s] = mfma16(wr[g][s][ks], bf, acc[g][s]);
    }

    const bool valid = t < len;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      float rr[4], zz[4], nn[4], hn[4], ho[4];
      const float gr[4] = {gxc[0][s].x, gxc[0][s].y, gxc[0][s].z, gxc[0][s].w};
      const float gz[4] = {gxc[1][s].x, gxc[1][s].y, gxc[1][s].z, gxc[1][s].w};
      const float gn[4] = {gxc[2][s].x, gxc[2][s].y, gxc[2][s].z, gxc[2][s].w};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        rr[r] = fast_sigmoid(gr[r] + acc[0][s][r] + bh[0][s][r]);
        zz[r] = fast_sigmoid(gz[r] + acc[1][s][r] + bh[1][s][r]);
        hn[r] = acc[2][s][r] + bh[2][s][r];
        nn[r] = fast_tanh(gn[r] + rr[r] * hn[r]);
        const float hnew = nn[r] + zz[r] * (hreg[s][r] - nn[r]);
        ho[r] = valid ? hnew : 0.f;
        hreg[s][r] = valid ? hnew : hreg[s][r];
      }
      const int unit = 32 * w + 16 * s + 4 * kq;
      put4(hs[cur ^ 1], bcol, unit, hreg[s][0], hreg[s][1], hreg[s][2], hreg[s][3]);
      if (brow_ok && !(a.dbg & 1)) st4(out_b + (long)t * a.out_ld + dir * H + unit, ho[0], ho[1], ho[2], ho[3]);
      if (q.saved && brow_ok && !(a.dbg & 1)) save_gates<BF16>(q.saved, t, ntile, tile, w, s, lane, rr, zz, nn, hn);
    }
    lds_barrier();
  }
}

// ------------------------------------------------------------------------------------------------
// backward through time.  Per step (processed in the reverse of the forward order):
//   dh   = dout[t] + carry                                  (valid rows; otherwise carry passes through)
//   dn = dh(1-z); dz = dh(h_prev - n); dn' = dn(1-n^2); dz' = dz z(1-z); dr' = dn' hn r(1-r)
//   dgx[t] = [dr', dz', dn']          (grad wrt x-side pre-activations, incl. b_ih)
//   dgh[t] = [dr', dz', dn' r]        (grad wrt h-side pre-activations, incl. b_hh)
//   carry  = dh z + dgh[t] . W_hh                            ([16,384].[384,128] on the matrix cores)
// dW_ih, dW_hh, biases and the gradient to the layer input are plain GEMMs over the stored dgx/dgh (engine).
// ------------------------------------------------------------------------------------------------
template <bool BF16>
__global__ __launch_bounds__(256, 1) void gru_bwd_kernel(GruBwdArgs a) {
  using C = Cfg<BF16>;
  __shared__ __attribute__((aligned(16))) Tile<BF16, G> ds[2];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int tile = blockIdx.x, dir = blockIdx.y, mod = blockIdx.z;
  const GruSeqBwd& q = a.seq[mod][dir];
  const int B = a.B, T = a.T, ntile = gridDim.x;
  const int bcol = lane & 15, kq = lane >> 4;
  const int b = tile * a.btv + bcol;
  const bool brow_ok = bcol < a.btv && b < B;
  const int len = brow_ok ? a.lens[mod][b] : 0;

  // A fragments of W_hh^T: rows = this wave's units (32w+16s+i), k = gate row index (0..383)
  typename C::AFrag wr[2][C::KS_B];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int unit = 32 * w + 16 * s + (lane & 15);
#pragma unroll
    for (int ks = 0; ks < C::KS_B; ++ks) {
      if constexpr (BF16) {
        bf16x8 f;
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = to_bf16(q.w_hh[(long)(ks * 32 + 8 * kq + j) * H + unit]);
        wr[s][ks] = f;
      } else {
        wr[s][ks] = q.w_hh[(long)(ks * 4 + kq) * H + unit];
      }
    }
  }

  float carry[2][4];
  float sb[4][2][4];     // running bias-gradient sums of this lane's (unit, batch) slots: [dr', dz', dn', dn'*r]
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      carry[s][r] = 0.f;
      sb[0][s][r] = sb[1][s][r] = sb[2][s][r] = sb[3][s][r] = 0.f;
    }

  const long rs_o = (long)T * a.out_ld;
  const long rs_d = (long)T * a.dout_ld;
  const long bb = brow_ok ? b : 0;
  const float* out_b = q.out + bb * rs_o + dir * H;        // forward outputs of THIS direction (h_prev source)
  const float* dout_b = q.dout + bb * rs_d + a.dout_off * dir;
  float* dg_b = q.dg + bb * (long)T * 4 * H;
  float* hp_b = q.hprev + bb * (long)T * H;

  // software pipeline: the six operand vectors of step+1 are requested before step's math (global latency ~1-2 us
  // would otherwise sit on the critical path of every step)
  struct Ops { float rr[4], zz[4], nn[4], hn[4]; float4 DO, HP; };
  Ops nx[2];
  auto fetch = [&](int step, Ops* o) {
    const int t = dir ? step : T - 1 - step;
    const int tprev = dir ? t + 1 : t - 1;
    const bool valid = t < len;
    const bool hp_ok = valid && tprev >= 0 && tprev < len;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int unit = 32 * w + 16 * s + 4 * kq;
      const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
      // h_prev is the previous VALID output of this direction; with packed semantics that is simply out[tprev]
      // when tprev is inside [0,len) and the zero initial state otherwise.
      if (valid) load_gates<BF16>(q.saved, t, ntile, tile, w, s, lane, o[s].rr, o[s].zz, o[s].nn, o[s].hn);
      o[s].DO = valid ? ld4(dout_b + (long)t * a.dout_ld + unit) : zero;
      o[s].HP = hp_ok ? ld4(out_b + (long)tprev * a.out_ld + unit) : zero;
    }
  };
  fetch(0, nx);

  for (int step = 0; step < T; ++step) {
    // forward visited t in order (dir ? T-1..0 : 0..T-1); backward walks it the other way round
    const int t = dir ? step : T - 1 - step;
    const int cur = step & 1;
    const bool valid = t < len;
    Ops op[2] = {nx[0], nx[1]};
    if (step + 1 < T && !(a.dbg & 4)) fetch(step + 1, nx);
    float dhz[2][4];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int unit = 32 * w + 16 * s + 4 * kq;
      float drp[4] = {0.f, 0.f, 0.f, 0.f}, dzp[4] = {0.f, 0.f, 0.f, 0.f}, dnp[4] = {0.f, 0.f, 0.f, 0.f},
            dnr[4] = {0.f, 0.f, 0.f, 0.f};
      const float4 HP = op[s].HP;
      if (valid) {
        const float4 DO = op[s].DO;
        const float* rr = op[s].rr; const float* zz = op[s].zz; const float* nn = op[s].nn; const float* hn = op[s].hn;
        const float dd[4] = {DO.x, DO.y, DO.z, DO.w}, hp[4] = {HP.x, HP.y, HP.z, HP.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float dh = dd[r] + carry[s][r];
          const float dn = dh * (1.f - zz[r]);
          const float dz = dh * (hp[r] - nn[r]);
          dnp[r] = dn * (1.f - nn[r] * nn[r]);
          dzp[r] = dz * zz[r] * (1.f - zz[r]);
          drp[r] = dnp[r] * hn[r] * rr[r] * (1.f - rr[r]);
          dnr[r] = dnp[r] * rr[r];
          dhz[s][r] = dh * zz[r];
        }
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) dhz[s][r] = carry[s][r];
      }
      put4(ds[cur], bcol, 0 * H + unit, drp[0], drp[1], drp[2], drp[3]);
      put4(ds[cur], bcol, 1 * H + unit, dzp[0], dzp[1], dzp[2], dzp[3]);
      put4(ds[cur], bcol, 2 * H + unit, dnr[0], dnr[1], dnr[2], dnr[3]);
#pragma unroll
      for (int r = 0; r < 4; ++r) { sb[0][s][r] += drp[r]; sb[1][s][r] += dzp[r]; sb[2][s][r] += dnp[r]; sb[3][s][r] += dnr[r]; }
      if (brow_ok && !(a.dbg & 1)) {
        st4(hp_b + (long)t * H + unit, HP.x, HP.y, HP.z, HP.w);
        st4(dg_b + (long)t * 4 * H + 0 * H + unit, drp[0], drp[1], drp[2], drp[3]);
        st4(dg_b + (long)t * 4 * H + 1 * H + unit, dzp[0], dzp[1], dzp[2], dzp[3]);
        st4(dg_b + (long)t * 4 * H + 2 * H + unit, dnp[0], dnp[1], dnp[2], dnp[3]);
        st4(dg_b + (long)t * 4 * H + 3 * H + unit, dnr[0], dnr[1], dnr[2], dnr[3]);
      }
    }
    lds_barrier();
    f32x4 acc[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) acc[s] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (!(a.dbg & 2))
#pragma unroll
    for (int ks = 0; ks < C::KS_B; ++ks) {
      const auto bf = bfrag(ds[cur], ks, lane);
#pragma unroll
      for (int s = 0; s < 2; ++s) acc[s] = mfma16(wr[s][ks], bf, acc[s]);
    }
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int r = 0; r < 4; ++r) carry[s][r] = dhz[s][r] + acc[s][r];
    // ds[cur] is rewritten two steps from now; the barrier of the next step orders that write after these reads
  }
  // bias gradients: reduce over the 16 batch lanes of each lane group (lanes differing in bits 0..3), one atomic per unit
  if (q.db_ih || q.db_hh) {
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = sb[g][s][r];
          v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
          sb[g][s][r] = v;
        }
    if (bcol == 0) {
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int unit = 32 * w + 16 * s + 4 * kq + r;
          if (q.db_ih) {
            atomicAdd(&q.db_ih[0 * H + unit], sb[0][s][r]); atomicAdd(&q.db_ih[1 * H + unit], sb[1][s][r]);
            atomicAdd(&q.db_ih[2 * H + unit], sb[2][s][r]);
          }
          if (q.db_hh) {
            atomicAdd(&q.db_hh[0 * H + unit], sb[0][s][r]); atomicAdd(&q.db_hh[1 * H + unit], sb[1][s][r]);
            atomicAdd(&q.db_hh[2 * H + unit], sb[3][s][r]);
          }
        }
    }
  }
}

}  // namespace

int gru_forward(hipStream_t s, const GruFwdArgs& a, bool bf16) {
  if (a.B <= 0 || a.T <= 0) return set_error(MIMRL_ERR_ARG, "gru_forward: empty batch");
  if (a.btv < 1 || a.btv > BT) return set_error(MIMRL_ERR_ARG, "gru_forward: btv must be in [1,16]");
  dim3 grid((a.B + a.btv - 1) / a.btv, 2, a.nmod);
  if (bf16) hipLaunchKernelGGL(gru_fwd_kernel<true>, grid, dim3(256), 0, s, a);
  else hipLaunchKernelGGL(gru_fwd_kernel<false>, grid, dim3(256), 0, s, a);
  LAUNCH_CHECK();
  return MIMRL_OK;
}

int gru_backward(hipStream_t s, const GruBwdArgs& a, bool bf16) {
  if (a.B <= 0 || a.T <= 0) return set_error(MIMRL_ERR_ARG, "gru_backward: empty batch");
  if (a.btv < 1 || a.btv > BT) return set_error(MIMRL_ERR_ARG, "gru_backward: btv must be in [1,16]");
  dim3 grid((a.B + a.btv - 1) / a.btv, 2, a.nmod);
  if (bf16) hipLaunchKernelGGL(gru_bwd_kernel<true>, grid, dim3(256), 0, s, a);
  else hipLaunchKernelGGL(gru_bwd_kernel<false>, grid, dim3(256), 0, s, a);
  LAUNCH_CHECK();
  return MIMRL_OK;
}

long gru_saved_floats(int B, int T) {
  return (long)T * B * 4 * 8 * 256;   // worst case: one batch row per workgroup (btv = 1)
}

int gru_pick_btv(int B, int nmod) {
  static const int force = getenv("MIMRL_GRU_BTV") ? atoi(getenv("MIMRL_GRU_BTV")) : 0;   // tuning knob
  if (force >= 1 && force <= BT) return force;
  int btv = (B * nmod * 2 + 127) / 128;      // ~128 workgroups (measured best at B=128: 4 rows per workgroup)
  if (btv < 1) btv = 1;
  if (btv > BT) btv = BT;
  return btv;
}

}  // namespace mimrl
