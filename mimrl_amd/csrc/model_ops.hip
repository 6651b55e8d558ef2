// Non-GEMM kernels of the model forward/backward (see model_ops.h for the reference citations).
#include "model_ops.h"
#include <algorithm>

#include <cstring>

#include "kmix_device.h"

namespace mimrl {

namespace {

// ------------------------------------------------------------------ lengths
__global__ void seq_lengths_kernel(const float* __restrict__ x, int T, int d, int* __restrict__ lens,
                                   const float* __restrict__ x2, int d2, int* __restrict__ lens2) {
  __shared__ int cnt;
  if (blockIdx.y == 1) { x = x2; d = d2; lens = lens2; }      // second modality of a paired launch
  const int b = blockIdx.x, lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  if (threadIdx.x == 0) cnt = 0;
  __syncthreads();
  // four rows per wave and pass, all loads issued before the first reduction: the scan is a chain of memory round trips
  for (int t0 = 4 * w; t0 < T; t0 += 4 * nw) {
    float s[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int t = t0 + q < T ? t0 + q : T - 1;
      const float* row = x + ((long)b * T + t) * d;
      s[q] = 0.f;
      for (int j = lane; j < d; j += 64) s[q] += fabsf(row[j]);
    }
    int c = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) c += (t0 + q < T && wave_sum(s[q]) != 0.f) ? 1 : 0;
    if (lane == 0 && c) atomicAdd(&cnt, c);
  }
  __syncthreads();
  if (threadIdx.x == 0) lens[b] = cnt > 0 ? cnt : 1;
}

// ------------------------------------------------------------------ text branch
// backward: `dmean` (optional) = gradient of this slot's temporal mean [B, D] (Model.py:466), added as dmean / T -- the
// feat_mean backward folded into its consumer
__global__ void text_post_kernel(const float* __restrict__ src, float* __restrict__ cube, long n, int T, int L, int K,
                                 int D, int slot, float p, RngKey key, uint32_t stream, int backward, const float* __restrict__ dmean) {
  const float invT = 1.f / T;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int dd = i % D;
    const long bt = i / D;
    const int t = bt % T;
    const long b = bt / T;
    const long ci = ((b * L + t) * K + slot) * D + dd;
    const float sc = drop_scale(p, key, stream, (uint32_t)i);
    if (backward) cube[i] = sc * (src[ci] + (dmean ? dmean[b * D + dd] * invT : 0.f));     // here: src = dcube, cube = dsrc (contiguous [B,T,D])
    else cube[ci] = sc * src[i];
  }
}

// ------------------------------------------------------------------ LN + ReLU + dropout on the bi-GRU output
struct LnSide { const float *h2, *gamma, *beta; float *mean, *rstd, *ds, *dgamma, *dbeta; int slot; float p; uint32_t stream; const float* dmean; int ds_bf16 = 0; };
template <int PER>   // D = 64*PER; blockIdx.y selects the modality (audio / video) of a paired launch
__global__ void ln_relu_drop_fwd_kernel(LnSide s0, LnSide s1, float* __restrict__ cube, long rows, int T, int L, int K,
                                        RngKey key) {
  const LnSide& sd = blockIdx.y ? s1 : s0;
  const float* __restrict__ h2 = sd.h2; const float* __restrict__ gamma = sd.gamma; const float* __restrict__ beta = sd.beta;
  float* __restrict__ mean = sd.mean; float* __restrict__ rstd = sd.rstd;
  const int slot = sd.slot; const float p = sd.p; const uint32_t stream = sd.stream;
  constexpr int D = 64 * PER;
  const int lane = threadIdx.x & 63;
  const long wid = (blockIdx.x * (long)blockDim.x + threadIdx.x) >> 6, nwv = ((long)gridDim.x * blockDim.x) >> 6;
  for (long r = wid; r < rows; r += nwv) {
    const float* hr = h2 + r * 2 * D;
    float v[PER], s = 0.f;
#pragma unroll
    for (int i = 0; i < PER; ++i) { v[i] = hr[lane + 64 * i] + hr[D + lane + 64 * i]; s += v[i]; }
    const float mu = wave_sum(s) * (1.f / D);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < PER; ++i) { const float c = v[i] - mu; q += c * c; }
    const float rs = rsqrtf(wave_sum(q) * (1.f / D) + LN_EPS);
    if (lane == 0) { mean[r] = mu; rstd[r] = rs; }
    const long b = r / T, t = r % T;
    float* out = cube + ((b * L + t) * K + slot) * D;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int j = lane + 64 * i;
      float y = (v[i] - mu) * rs * gamma[j] + beta[j];
      y = y > 0.f ? y : 0.f;
      out[j] = y * drop_scale(p, key, stream, (uint32_t)(r * D + j));
    }
  }
}


// Round 3: the same backward with 16 lanes per row (D = 128: two float4 per lane and operand), FOUR rows per wave and pass, grid sized
// by the rows.  The one-row-per-wave kernel above walks its rows as a dependent chain (4-byte loads -> two wave reductions -> stores,
// ~2.9 us per row) on at most 1024 waves: 28 us at cfg2 (6 rows per wave) and 360 us at cfg3 (125 rows per wave, 1.4 TB/s), on the
// critical chain in front of the BPTT.  Element indices of the dropout hash are unchanged.
#ifdef MIMRL_PHASE_PROBE
__device__ long long g_mo_phase[16];    // tail_pre_kernel: 0..5 (workgroup (0, slot 1)); ln_relu_drop_bwd16_kernel: 8..13 (workgroup (0, 0))
#define MPHASE(c, i) do { if ((c) && threadIdx.x == 0) g_mo_phase[i] = (long long)wall_clock64(); } while (0)
#else
#define MPHASE(c, i) do { } while (0)
#endif
__global__ __launch_bounds__(256) void ln_relu_drop_bwd16_kernel(LnSide s0, LnSide s1, const float* __restrict__ dcube, long rows, int T, int L,
                                                                  int K, RngKey key) {
  constexpr int D = 128;
  const LnSide& sd = blockIdx.y ? s1 : s0;
  const float* __restrict__ h2 = sd.h2; const float* __restrict__ gamma = sd.gamma; const float* __restrict__ beta = sd.beta;
  const float* __restrict__ mean = sd.mean; const float* __restrict__ rstd = sd.rstd;
  float* __restrict__ ds = sd.ds;
  const int slot = sd.slot; const float p = sd.p; const uint32_t stream = sd.stream;
  const float* __restrict__ dmean = sd.dmean;
  const float invT = 1.f / T;
  const uint32_t rstep = (uint32_t)(*key.step + key.add);   // once (see drop_scale_at)
  __shared__ LdsAcc sg[D], sb[D];
  const int tid = threadIdx.x, sub = tid & 15, rw = tid >> 4;          // 16 row slots per workgroup
  const bool pb = blockIdx.x == 0 && blockIdx.y == 0;
  MPHASE(pb, 8);
  if (tid < D) { sg[tid].zero(); sb[tid].zero(); }
  __syncthreads();
  MPHASE(pb, 9);
  const int c0 = 4 * sub, c1 = 64 + 4 * sub;                            // this lane's two column quads
  const float4 g0 = *reinterpret_cast<const float4*>(gamma + c0), g1 = *reinterpret_cast<const float4*>(gamma + c1);
  const float4 b0 = *reinterpret_cast<const float4*>(beta + c0), b1 = *reinterpret_cast<const float4*>(beta + c1);
  float ag[8], ab[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { ag[i] = 0.f; ab[i] = 0.f; }
  const long stride = (long)gridDim.x * 16;
  // round 5b: TWO rows per 16-lane slot and trip, every operand of both requested before the first reduction (one row per slot in flight
  // was 6 KB per wave: at 4 waves per CU the launch moved 3.2 TB/s at cfg3, 145 us on the chain in front of the layer-1 BPTT).  Loads are
  // unconditional from clamped rows; rows are still visited in increasing order per slot, so the parameter-gradient sums keep their order.
  const float invTm = dmean ? invT : 0.f;
  for (long r0 = (long)blockIdx.x * 16 + rw; r0 < rows; r0 += 2 * stride) {   // (the 16 lanes of a row slot share r: the group reductions are whole)
    float4 hf0[2], hf1[2], hb0[2], hb1[2], d0[2], d1[2], m0[2], m1[2];
    float mus[2], rss[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const long r = min(r0 + u * stride, rows - 1);
      const float* hr = h2 + r * 2 * D;
      const long b = r / T, t = r - b * T;
      const float* dc = dcube + ((b * L + t) * K + slot) * D;
      const float* dm = dmean ? dmean + b * D : gamma;      // (no gradient of the mean: any readable address, the values are multiplied by 0)
      hf0[u] = *reinterpret_cast<const float4*>(hr + c0); hf1[u] = *reinterpret_cast<const float4*>(hr + c1);
      hb0[u] = *reinterpret_cast<const float4*>(hr + D + c0); hb1[u] = *reinterpret_cast<const float4*>(hr + D + c1);
      d0[u] = *reinterpret_cast<const float4*>(dc + c0); d1[u] = *reinterpret_cast<const float4*>(dc + c1);
      m0[u] = *reinterpret_cast<const float4*>(dm + c0); m1[u] = *reinterpret_cast<const float4*>(dm + c1);
      mus[u] = mean[r]; rss[u] = rstd[r];
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const long r = r0 + u * stride;
      const bool ok = r < rows;
      const float mu = mus[u], rs = rss[u];
      const float hv[8] = {hf0[u].x + hb0[u].x, hf0[u].y + hb0[u].y, hf0[u].z + hb0[u].z, hf0[u].w + hb0[u].w, hf1[u].x + hb1[u].x, hf1[u].y + hb1[u].y, hf1[u].z + hb1[u].z, hf1[u].w + hb1[u].w};
      const float dv[8] = {d0[u].x + m0[u].x * invTm, d0[u].y + m0[u].y * invTm, d0[u].z + m0[u].z * invTm, d0[u].w + m0[u].w * invTm,
                           d1[u].x + m1[u].x * invTm, d1[u].y + m1[u].y * invTm, d1[u].z + m1[u].z * invTm, d1[u].w + m1[u].w * invTm};
      const float gv[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w}, bv[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
      float xh[8], dxh[8], s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int j = (i < 4 ? c0 : c1 - 4) + i;
        xh[i] = (hv[i] - mu) * rs;
        const float y = xh[i] * gv[i] + bv[i];
        float dy = dv[i] * drop_scale_at(p, key, rstep, stream, (uint32_t)(r * D + j));
        dy = (ok && y > 0.f) ? dy : 0.f;
        ag[i] += dy * xh[i];
        ab[i] += dy;
        dxh[i] = dy * gv[i];
        s1 += dxh[i];
        s2 += dxh[i] * xh[i];
      }
      s1 = group_sum<16>(s1) * (1.f / D);
      s2 = group_sum<16>(s2) * (1.f / D);
      if (ok) {
        float4 o0, o1;
        o0.x = rs * (dxh[0] - s1 - xh[0] * s2); o0.y = rs * (dxh[1] - s1 - xh[1] * s2); o0.z = rs * (dxh[2] - s1 - xh[2] * s2); o0.w = rs * (dxh[3] - s1 - xh[3] * s2);
        o1.x = rs * (dxh[4] - s1 - xh[4] * s2); o1.y = rs * (dxh[5] - s1 - xh[5] * s2); o1.z = rs * (dxh[6] - s1 - xh[6] * s2); o1.w = rs * (dxh[7] - s1 - xh[7] * s2);
        if (sd.ds_bf16) {   // (LnSide::ds_bf16: the layer-1 BPTT reads its dout as bf16 -- same element indices, half the bytes)
          __bf16* dsb = reinterpret_cast<__bf16*>(ds);
          bf16x4 q0, q1;
          q0[0] = to_bf16(o0.x); q0[1] = to_bf16(o0.y); q0[2] = to_bf16(o0.z); q0[3] = to_bf16(o0.w);
          q1[0] = to_bf16(o1.x); q1[1] = to_bf16(o1.y); q1[2] = to_bf16(o1.z); q1[3] = to_bf16(o1.w);
          *reinterpret_cast<bf16x4*>(dsb + r * D + c0) = q0;
          *reinterpret_cast<bf16x4*>(dsb + r * D + c1) = q1;
        } else {
          *reinterpret_cast<float4*>(ds + r * D + c0) = o0;
          *reinterpret_cast<float4*>(ds + r * D + c1) = o1;
        }
      }
    }
  }
  MPHASE(pb, 10);
  // parameter gradients: the four row slots of a wave share columns (lanes that differ in bits 4, 5), then LDS, then one atomic per column
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    float a = ag[i], bq = ab[i];
    a += __shfl_xor(a, 16, 64); a += __shfl_xor(a, 32, 64);
    bq += __shfl_xor(bq, 16, 64); bq += __shfl_xor(bq, 32, 64);
    if ((tid & 63) < 16) { const int j = (i < 4 ? c0 : c1 - 4) + i; sg[j].add(a); sb[j].add(bq); }
  }
  __syncthreads();
  MPHASE(pb, 11);
  if (tid < D) { acc_add(&sd.dgamma[tid], sg[tid].get()); acc_add(&sd.dbeta[tid], sb[tid].get()); }
  MPHASE(pb, 12);
}

// ------------------------------------------------------------------ the three pre-CubeMLP pieces of one forward tail in ONE launch
// text_post (slot 0) | LN + ReLU + dropout of the audio / video GRU outputs (slots 1, 2) -> cube, and the temporal means T_F / A_F / V_F
// of exactly those cube values (Model.py:452-466).  One workgroup per (sample, slot); a wave owns rows t = w, w+4, ...; two rows in
// flight per pass.  Element indices of the dropout hash are those of the separate kernels (the backward regenerates the masks).
struct TailPre { const float* tx_raw; LnSide a, v; float p_text; };
__global__ __launch_bounds__(256) void tail_pre_kernel(TailPre tp, float* __restrict__ cube, float* __restrict__ feats, int B, int T, int L, int K,
                                                       RngKey key) {
  constexpr int D = 128;
  __shared__ float part[4][D];
  const int b = blockIdx.x, slot = blockIdx.y, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  float acc0 = 0.f, acc1 = 0.f;
  const bool pb = blockIdx.x == 0 && blockIdx.y == 1;
  MPHASE(pb, 0);
  const uint32_t rstep = (uint32_t)(*key.step + key.add);   // once (see drop_scale_at)
  MPHASE(pb, 1);
  // RPP rows of a wave in flight per pass (round 3b: 2 -> 7; a pass is a dependent load -> LayerNorm -> store round trip and T = 50 was
  // seven of them per wave: 13 us on the chain of the stage-1 forward).  Rows are still visited in increasing t per wave, so the
  // temporal means keep their summation order.
  constexpr int RPP = 7;
  if (slot == 0) {
    for (int t0 = w; t0 < T; t0 += 4 * RPP) {
      float x[RPP][2];
#pragma unroll
      for (int q = 0; q < RPP; ++q) {
        const int t = t0 + 4 * q < T ? t0 + 4 * q : T - 1;
        const float* src = tp.tx_raw + ((long)b * T + t) * D;
        x[q][0] = src[lane]; x[q][1] = src[lane + 64];
      }
#pragma unroll
      for (int q = 0; q < RPP; ++q) {
        const int t = t0 + 4 * q;
        if (t < T) {
          const long r = (long)b * T + t;
          float* out = cube + (((long)b * L + t) * K + 0) * D;
          const float y0 = x[q][0] * drop_scale_at(tp.p_text, key, rstep, 0u, (uint32_t)(r * D + lane));
          const float y1 = x[q][1] * drop_scale_at(tp.p_text, key, rstep, 0u, (uint32_t)(r * D + lane + 64));
          out[lane] = y0; out[lane + 64] = y1;
          acc0 += y0; acc1 += y1;
        }
      }
    }
  } else {
    const LnSide& sd = slot == 1 ? tp.a : tp.v;
    const float g0 = sd.gamma[lane], g1 = sd.gamma[lane + 64], be0 = sd.beta[lane], be1 = sd.beta[lane + 64];
    for (int t0 = w; t0 < T; t0 += 4 * RPP) {
      float v[RPP][2];
#pragma unroll
      for (int q = 0; q < RPP; ++q) {
        const int t = t0 + 4 * q < T ? t0 + 4 * q : T - 1;
        const float* hr = sd.h2 + ((long)b * T + t) * 2 * D;
        v[q][0] = hr[lane] + hr[D + lane]; v[q][1] = hr[lane + 64] + hr[D + lane + 64];
      }
#pragma unroll
      for (int q = 0; q < RPP; ++q) {
        const int t = t0 + 4 * q;
        if (t < T) {                                            // (wave-uniform)
          const long r = (long)b * T + t;
          const float mu = wave_sum(v[q][0] + v[q][1]) * (1.f / D);
          const float c0 = v[q][0] - mu, c1 = v[q][1] - mu;
          const float rs = rsqrtf(wave_sum(c0 * c0 + c1 * c1) * (1.f / D) + LN_EPS);
          if (lane == 0) { sd.mean[r] = mu; sd.rstd[r] = rs; }
          float* out = cube + (((long)b * L + t) * K + sd.slot) * D;
          float y0 = c0 * rs * g0 + be0, y1 = c1 * rs * g1 + be1;
          y0 = (y0 > 0.f ? y0 : 0.f) * drop_scale_at(sd.p, key, rstep, sd.stream, (uint32_t)(r * D + lane));
          y1 = (y1 > 0.f ? y1 : 0.f) * drop_scale_at(sd.p, key, rstep, sd.stream, (uint32_t)(r * D + lane + 64));
          out[lane] = y0; out[lane + 64] = y1;
          acc0 += y0; acc1 += y1;
        }
      }
    }
  }
  MPHASE(pb, 2);
  part[w][lane] = acc0; part[w][lane + 64] = acc1;
  __syncthreads();
  if (threadIdx.x < D) {
    const int d = threadIdx.x;
    feats[((long)slot * B + b) * D + d] = (part[0][d] + part[1][d] + part[2][d] + part[3][d]) / T;
  }
  MPHASE(pb, 3);
}

// ------------------------------------------------------------------ the same three pieces for BOTH forward tails of a step in one launch
// (round 5).  The shared-prefix step runs two tails on the same encoder outputs -- stage 1's (critic update) and stage 2's (main update) --
// which differ only in their dropout keys.  As separate launches each tail read tx_raw and the bi-GRU outputs again, and read its cube
// slot a third time for the temporal means: at T = 500 (cfg3) 1.44 GB of HBM traffic on the chain of both tails (0.34 ms).  Here a row is
// read ONCE, its LayerNorm statistics are computed once, and the two masked copies go to the two cubes, the temporal means ride on the
// values in registers: 0.72 GB.  Workgroup = (sample, slot, chunk of `rpc` rows, rpc % 4 == 0); wave w owns rows t = w (mod 4) in
// increasing order, so with one chunk the means have the summation order of feat_mean_fwd_kernel / tail_pre_kernel.  Lane l owns
// columns 2l, 2l+1 (8-byte accesses).  With more than one chunk the partial sums go to `part` and tail_pre2_finish_kernel adds them in
// chunk order (fixed order: the means stay run-to-run reproducible).
struct TailPre2 {
  const float* tx_raw; LnSide a, v; float p_text;
  float* cube[2]; float* feats[2]; int add[2];
  float* part; int nchunk, rpc;
};
__global__ __launch_bounds__(256) void tail_pre2_kernel(TailPre2 tp, int B, int T, int L, int K, RngKey key) {
  constexpr int D = 128, RPP = 8;
  __shared__ float part[2][4][D];
  const int b = blockIdx.x, slot = blockIdx.y, ch = blockIdx.z, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int t_begin = ch * tp.rpc, t_end = min(T, t_begin + tp.rpc);
  const int c = 2 * lane;
  const uint32_t step = (uint32_t)*key.step;
  const uint32_t rs0 = step + (uint32_t)tp.add[0], rs1 = step + (uint32_t)tp.add[1];
  float* __restrict__ cube0 = tp.cube[0];
  float* __restrict__ cube1 = tp.cube[1];
  float2 acc0 = make_float2(0.f, 0.f), acc1 = acc0;
  if (slot == 0) {
    const float p = tp.p_text;
    for (int t0 = t_begin + w; t0 < t_end; t0 += 4 * RPP) {
      float2 x[RPP];
#pragma unroll
      for (int q = 0; q < RPP; ++q) {
        const int t = min(t0 + 4 * q, t_end - 1);
        x[q] = *reinterpret_cast<const float2*>(tp.tx_raw + ((long)b * T + t) * D + c);
      }
#pragma unroll
      for (int q = 0; q < RPP; ++q) {
        const int t = t0 + 4 * q;
        if (t < t_end) {
          const long r = (long)b * T + t;
          const long o = (((long)b * L + t) * K + 0) * D + c;
          const uint32_t e = (uint32_t)(r * D + c);
          float2 y0, y1;
          y0.x = x[q].x * drop_scale_at(p, key, rs0, 0u, e); y0.y = x[q].y * drop_scale_at(p, key, rs0, 0u, e + 1);
          y1.x = x[q].x * drop_scale_at(p, key, rs1, 0u, e); y1.y = x[q].y * drop_scale_at(p, key, rs1, 0u, e + 1);
          *reinterpret_cast<float2*>(cube0 + o) = y0;
          *reinterpret_cast<float2*>(cube1 + o) = y1;
          acc0.x += y0.x; acc0.y += y0.y; acc1.x += y1.x; acc1.y += y1.y;
        }
      }
    }
  } else {
    const LnSide& sd = slot == 1 ? tp.a : tp.v;
    const float* __restrict__ h2 = sd.h2;
    const float2 g = *reinterpret_cast<const float2*>(sd.gamma + c), be = *reinterpret_cast<const float2*>(sd.beta + c);
    const float p = sd.p; const uint32_t sid = sd.stream; const int oslot = sd.slot;
    for (int t0 = t_begin + w; t0 < t_end; t0 += 4 * RPP) {
      float2 v[RPP];
#pragma unroll
      for (int q = 0; q < RPP; ++q) {
        const int t = min(t0 + 4 * q, t_end - 1);
        const float* hr = h2 + ((long)b * T + t) * 2 * D + c;
        const float2 f = *reinterpret_cast<const float2*>(hr), r = *reinterpret_cast<const float2*>(hr + D);
        v[q].x = f.x + r.x; v[q].y = f.y + r.y;
      }
#pragma unroll
      for (int q = 0; q < RPP; ++q) {
        const int t = t0 + 4 * q;
        if (t < t_end) {                                        // (wave-uniform)
          const long r = (long)b * T + t;
          const float mu = wave_sum(v[q].x + v[q].y) * (1.f / D);
          const float c0 = v[q].x - mu, c1 = v[q].y - mu;
          const float rs = rsqrtf(wave_sum(c0 * c0 + c1 * c1) * (1.f / D) + LN_EPS);
          if (lane == 0) { sd.mean[r] = mu; sd.rstd[r] = rs; }
          float z0 = c0 * rs * g.x + be.x, z1 = c1 * rs * g.y + be.y;
          z0 = z0 > 0.f ? z0 : 0.f; z1 = z1 > 0.f ? z1 : 0.f;
          const long o = (((long)b * L + t) * K + oslot) * D + c;
          const uint32_t e = (uint32_t)(r * D + c);
          float2 y0, y1;
          y0.x = z0 * drop_scale_at(p, key, rs0, sid, e); y0.y = z1 * drop_scale_at(p, key, rs0, sid, e + 1);
          y1.x = z0 * drop_scale_at(p, key, rs1, sid, e); y1.y = z1 * drop_scale_at(p, key, rs1, sid, e + 1);
          *reinterpret_cast<float2*>(cube0 + o) = y0;
          *reinterpret_cast<float2*>(cube1 + o) = y1;
          acc0.x += y0.x; acc0.y += y0.y; acc1.x += y1.x; acc1.y += y1.y;
        }
      }
    }
  }
  part[0][w][c] = acc0.x; part[0][w][c + 1] = acc0.y;
  part[1][w][c] = acc1.x; part[1][w][c + 1] = acc1.y;
  __syncthreads();
  {
    const int o = threadIdx.x >> 7, d = threadIdx.x & 127;
    const float sum = part[o][0][d] + part[o][1][d] + part[o][2][d] + part[o][3][d];
    if (tp.nchunk == 1) tp.feats[o][((long)slot * B + b) * D + d] = sum / T;
    else tp.part[((((long)o * 3 + slot) * B + b) * tp.nchunk + ch) * D + d] = sum;
  }
}
__global__ __launch_bounds__(256) void tail_pre2_finish_kernel(TailPre2 tp, int B, int T) {
  constexpr int D = 128;
  const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;      // over [2][3][B][D]
  if (i >= 2L * 3 * B * D) return;
  const int d = i % D; const long sb = (i / D) % (3L * B); const int o = (int)(i / (3L * B * D));
  const float* p = tp.part + (((long)o * 3 * B + sb) * tp.nchunk) * D + d;
  float sum = 0.f;
  for (int ch = 0; ch < tp.nchunk; ++ch) sum += p[(long)ch * D];
  tp.feats[o][sb * D + d] = sum / T;
}

// ------------------------------------------------------------------ temporal means
__global__ void feat_mean_fwd_kernel(const float* __restrict__ cube, float* __restrict__ feats, int B, int T, int L,
                                     int K, int D) {
  // 512 threads = 128 feature columns x 4 time phases: four independent load streams per column instead of one T-long chain
  __shared__ float part[4][128];
  const int b = blockIdx.x, k = blockIdx.y, tq = threadIdx.x >> 7, d0 = threadIdx.x & 127;
  for (int dbase = 0; dbase < D; dbase += 128) {
    const int d = dbase + d0;
    float s = 0.f;
    if (d < D) {
#pragma unroll 4
      for (int t = tq; t < T; t += 4) s += cube[(((long)b * L + t) * K + k) * D + d];
    }
    part[tq][d0] = s;
    __syncthreads();
    if (tq == 0 && d < D) feats[((long)k * B + b) * D + d] = (part[0][d0] + part[1][d0] + part[2][d0] + part[3][d0]) / T;
    __syncthreads();
  }
}
__global__ void feat_mean_bwd_kernel(const float* __restrict__ dfeats, float* __restrict__ dcube, long n, int B, int T,
                                     int L, int K, int D) {
  const float inv = 1.f / T;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int d = i % D;
    long r = i / D;
    const int k = r % K; r /= K;
    const int t = r % T;
    const long b = r / T;
    dcube[((b * L + t) * K + k) * D + d] += dfeats[((long)k * B + b) * D + d] * inv;
  }
}

// ------------------------------------------------------------------ head
// (round 3b: 512 threads = D columns x 4 row phases for D = 128, each phase a stream of independent loads; one thread per column walking
//  all L*K rows was an 11 us latency chain on the step's critical path)
__global__ void head_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                float* __restrict__ ff, float* __restrict__ pred, int L, int K, int D, float scale) {
  __shared__ float red[16];
  __shared__ float ph[4][128];
  const int b = blockIdx.x;
  float part = 0.f;
  if (D == 128 && blockDim.x == 512) {
    const int d = threadIdx.x & 127, q = threadIdx.x >> 7, n = L * K;
    const float* xb = x + (long)b * n * D + d;
    float s = 0.f;
#pragma unroll 8
    for (int i = q; i < n; i += 4) s += xb[(long)i * D];
    ph[q][d] = s;
    __syncthreads();
    if (q == 0) {
      s = ((ph[0][d] + ph[1][d]) + (ph[2][d] + ph[3][d])) * scale;
      ff[(long)b * D + d] = s;
      part = s * w[d];
    }
  } else {
    for (int d = threadIdx.x; d < D; d += blockDim.x) {
      float s = 0.f;
      const float* xb = x + (long)b * L * K * D + d;
      for (int i = 0; i < L * K; ++i) s += xb[(long)i * D];
      s *= scale;
      ff[(long)b * D + d] = s;
      part += s * w[d];
    }
  }
  part = block_sum(part, red);
  if (threadIdx.x == 0) pred[b] = part + bias[0];
}
__global__ void head_bwd_kernel(const float* __restrict__ dff_ext, const float* __restrict__ dpred,
                                const float* __restrict__ w, const float* __restrict__ ff, float* __restrict__ dx,
                                float* __restrict__ dw, float* __restrict__ dbias, int L, int K, int D, float scale, GatherSum gs,
                                int use_gs) {
  const int b = blockIdx.x;
  const float dp = dpred[b];
  for (int d = threadIdx.x; d < D; d += blockDim.x) {
    float de = 0.f;
    if (use_gs) {
      for (int q = 0; q < gs.n; ++q)
        if (b < gs.rows[q]) de += gs.src[q][(long)b * gs.ld[q] + gs.off[q] + d];
    } else if (dff_ext) de = dff_ext[(long)b * D + d];
    const float g = scale * (de + dp * w[d]);
    float* xb = dx + (long)b * L * K * D + d;
    for (int i = 0; i < L * K; ++i) xb[(long)i * D] = g;
    acc_add(&dw[d], dp * ff[(long)b * D + d]);
  }
  if (threadIdx.x == 0) acc_add(dbias, dp);
}

// ------------------------------------------------------------------ row LayerNorm (one wave per row)
__global__ void rowln_fwd_kernel(const float* __restrict__ y, const float* __restrict__ gamma,
                                 const float* __restrict__ beta, float* __restrict__ z, float* __restrict__ mean,
                                 float* __restrict__ rstd, long R, int n) {
  const int lane = threadIdx.x & 63;
  const long wid = (blockIdx.x * (long)blockDim.x + threadIdx.x) >> 6, nwv = ((long)gridDim.x * blockDim.x) >> 6;
  for (long r = wid; r < R; r += nwv) {
    const float* yr = y + r * n;
    float s = 0.f;
    for (int j = lane; j < n; j += 64) s += yr[j];
    const float mu = wave_sum(s) / n;
    float q = 0.f;
    for (int j = lane; j < n; j += 64) { const float c = yr[j] - mu; q += c * c; }
    const float rs = rsqrtf(wave_sum(q) / n + LN_EPS);
    if (lane == 0) { mean[r] = mu; rstd[r] = rs; }
    for (int j = lane; j < n; j += 64) z[r * n + j] = (yr[j] - mu) * rs * gamma[j] + beta[j];
  }
}
template <int PER>   // n <= 64*PER; lane owns columns lane + 64*i for the whole kernel -> dgamma/dbeta accumulate in registers
__global__ void rowln_bwd_kernel(const float* __restrict__ y, const float* __restrict__ gamma,
                                 const float* __restrict__ mean, const float* __restrict__ rstd,
                                 const float* __restrict__ dz, float* __restrict__ dy, float* __restrict__ dgamma,
                                 float* __restrict__ dbeta, long R, int n) {
  __shared__ LdsAcc sh[2 * 64 * PER];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 2 * 64 * PER; i += blockDim.x) sh[i].zero();
  __syncthreads();
  float gam[PER], ag[PER], ab[PER];
#pragma unroll
  for (int i = 0; i < PER; ++i) { const int j = lane + 64 * i; gam[i] = j < n ? gamma[j] : 0.f; ag[i] = 0.f; ab[i] = 0.f; }
  const long wid = (blockIdx.x * (long)blockDim.x + threadIdx.x) >> 6, nwv = ((long)gridDim.x * blockDim.x) >> 6;
  for (long r = wid; r < R; r += nwv) {
    const float mu = mean[r], rs = rstd[r];
    float xh[PER], g[PER], s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int j = lane + 64 * i;
      const bool ok = j < n;
      xh[i] = ok ? (y[r * n + j] - mu) * rs : 0.f;
      g[i] = ok ? dz[r * n + j] : 0.f;
      const float dxh = g[i] * gam[i];
      s1 += dxh; s2 += dxh * xh[i];
      ag[i] += g[i] * xh[i]; ab[i] += g[i];
    }
    s1 = wave_sum(s1) / n; s2 = wave_sum(s2) / n;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int j = lane + 64 * i;
      if (j < n) dy[r * n + j] = rs * (g[i] * gam[i] - s1 - xh[i] * s2);
    }
  }
#pragma unroll
  for (int i = 0; i < PER; ++i) { sh[lane + 64 * i].add(ag[i]); sh[64 * PER + lane + 64 * i].add(ab[i]); }
  __syncthreads();
  for (int j = threadIdx.x; j < n; j += blockDim.x) { acc_add(&dgamma[j], sh[j].get()); acc_add(&dbeta[j], sh[64 * PER + j].get()); }
}

// ------------------------------------------------------------------ column LayerNorm (thread per (b,c))
__global__ void colln_fwd_kernel(const float* __restrict__ y, const float* __restrict__ gamma,
                                 const float* __restrict__ beta, float* __restrict__ z, float* __restrict__ mean,
                                 float* __restrict__ rstd, int B, int n, int C) {
  const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i >= (long)B * C) return;
  const long b = i / C;
  const int c = i % C;
  const float* yb = y + b * n * C + c;
  float s = 0.f;
  for (int l = 0; l < n; ++l) s += yb[(long)l * C];
  const float mu = s / n;
  float q = 0.f;
  for (int l = 0; l < n; ++l) { const float d = yb[(long)l * C] - mu; q += d * d; }
  const float rs = rsqrtf(q / n + LN_EPS);
  mean[i] = mu; rstd[i] = rs;
  float* zb = z + b * n * C + c;
  for (int l = 0; l < n; ++l) zb[(long)l * C] = (yb[(long)l * C] - mu) * rs * gamma[l] + beta[l];
}
// Workgroup = 64 consecutive columns (one per lane); its 4 waves split the L axis (l = wave, wave+4, ...), keep their
// (dz, xhat) pairs in registers -- one pass over memory -- and meet in LDS for the two column sums.
__global__ __launch_bounds__(256) void colln_bwd_kernel(const float* __restrict__ y, const float* __restrict__ gamma,
                                                        const float* __restrict__ mean, const float* __restrict__ rstd,
                                                        const float* __restrict__ dz, float* __restrict__ dy,
                                                        float* __restrict__ dgamma, float* __restrict__ dbeta, int B, int n, int C) {
  extern __shared__ float sh[];   // [2][n][64] per-column terms of dgamma / dbeta, then [2][4][64] partial column sums
  float* red = sh + 2 * n * 64;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long i = blockIdx.x * 64L + lane;          // column id = b * C + c   (C is a multiple of 64)
  const long b = i / C;
  const int c = i % C;
  const float* yb = y + b * n * C + c;
  const float* dzb = dz + b * n * C + c;
  float* dyb = dy + b * n * C + c;
  const float mu = mean[i], rs = rstd[i];
  constexpr int PER = 16;                          // n <= 64
  float g[PER], xh[PER];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int q = 0; q < PER; ++q) {
    const int l = wave + 4 * q;
    if (l < n) {
      g[q] = dzb[(long)l * C];
      xh[q] = (yb[(long)l * C] - mu) * rs;
      const float dxh = g[q] * gamma[l];
      s1 += dxh; s2 += dxh * xh[q];
      sh[l * 64 + lane] = g[q] * xh[q];
      sh[(n + l) * 64 + lane] = g[q];
    }
  }
  red[wave * 64 + lane] = s1;
  red[(4 + wave) * 64 + lane] = s2;
  __syncthreads();
  s1 = (red[lane] + red[64 + lane] + red[128 + lane] + red[192 + lane]) / n;
  s2 = (red[256 + lane] + red[320 + lane] + red[384 + lane] + red[448 + lane]) / n;
#pragma unroll
  for (int q = 0; q < PER; ++q) {
    const int l = wave + 4 * q;
    if (l < n) dyb[(long)l * C] = rs * (g[q] * gamma[l] - s1 - xh[q] * s2);
  }
  for (int q = threadIdx.x; q < 2 * n; q += blockDim.x) {
    float t = 0.f;
    for (int j = 0; j < 64; ++j) t += sh[q * 64 + ((j + q) & 63)];
    acc_add(q < n ? &dgamma[q] : &dbeta[q - n], t);
  }
}

// ------------------------------------------------------------------ K-axis mix, fused (thread per (row, d))
// Sizes are template parameters (NK = max(ik,hk,ok) rounded to {4,8}) so that every per-thread array lives in
// registers and every loop unrolls; runtime sizes only mask the tails.
template <int NK>
__global__ __launch_bounds__(256) void kmix_fwd_kernel(const float* __restrict__ x, float* __restrict__ z, KMixW w,
                                                       long R, int D) {
  __shared__ float sw[3 * KM * KM + 4 * KM];
  kmix_stage_weights(w, sw);
  const float* g = sw + 3 * KM * KM + 2 * KM; const float* be = g + KM;
  const bool pow2 = (D & (D - 1)) == 0; const int dsh = __ffs(D) - 1;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < R * D; i += (long)gridDim.x * blockDim.x) {
    const long r = pow2 ? (i >> dsh) : i / D; const int d = pow2 ? (int)(i & (D - 1)) : (int)(i % D);   // (64-bit division: ~80 instructions)
    KMixVals<NK> v;
#pragma unroll
    for (int k = 0; k < NK; ++k) v.x[k] = k < w.ik ? x[(r * w.ik + k) * D + d] : 0.f;
#pragma unroll
    for (int o = 0; o < NK; ++o)
      v.sc[o] = o < w.ok ? drop_scale(w.drop_p, w.key, w.stream_id, (uint32_t)((r * w.ok + o) * D + d)) : 0.f;
    kmix_forward_vals<NK>(w, sw, v);
    if (w.ln_first) {
#pragma unroll
      for (int o = 0; o < NK; ++o) if (o < w.ok) z[(r * w.ok + o) * D + d] = v.y[o];
    } else {
      float out[NK];
      ln_small<NK>(v.y, w.ok, g, be, out, v.xh, v.mu, v.rs);
#pragma unroll
      for (int o = 0; o < NK; ++o) if (o < w.ok) z[(r * w.ok + o) * D + d] = out[o];
    }
  }
}

// backward: per-thread gradients are reduced over the wave with DPP-free shuffles only ONCE per workgroup iteration:
// each thread accumulates its weight-gradient contributions in registers across its grid-stride iterations, and the
// wave/LDS/global reduction runs once at the end of the kernel.
// MODE 0: data gradient and parameter gradients together; MODE 1: data gradient only (the critical path of the CubeMLP
// backward: no accumulators, no reductions, no atomics); MODE 2: parameter gradients only (same arithmetic recomputed on
// a side stream)
#ifndef KMIX_MINB
#define KMIX_MINB 1
#endif
// `make PHASE_PROBE=1` (tools/cube_phase.py): workgroup 0 of the last kmix_bwd launch leaves 100 MHz ticks (slots 0..7: <= 1000 rows, 8..15)
#ifdef MIMRL_PHASE_PROBE
__device__ long long g_kmix_phase[16];
#define KPHASE(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_kmix_phase[(R <= 2000 ? 0 : 8) + (i)] = (long long)wall_clock64(); } while (0)
#else
#define KPHASE(i) do { } while (0)
#endif
// EXACT (round 3b): ik = hk = ok = NK, LayerNorm last, no dropout -- the model's configuration.  The kernel argument is copied with those
// fields set to constants, so every `k < w.ik` select, both `w.ln_first` branches and the dropout hash fold away (the loop is VALU-bound)
template <int NK, int MODE, bool EXACT = false>
__global__ __launch_bounds__(256, KMIX_MINB) void kmix_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dz,
                                                       float* __restrict__ dx, KMixW w_, long R, int D) {
  KMixW w = w_;
  if constexpr (EXACT) { w.ik = NK; w.hk = NK; w.ok = NK; w.ln_first = 0; w.drop_p = 0.f; }
  constexpr bool GRADS = MODE != 1, DX = MODE != 2;
  __shared__ float sw[3 * KM * KM + 4 * KM];
  __shared__ LdsAcc sg[3 * KM * KM + 4 * KM];   // gradient accumulators, same packing
  KPHASE(0);
  kmix_stage_weights(w, sw);
  for (int i = threadIdx.x; i < 3 * KM * KM + 4 * KM; i += blockDim.x) sg[i].zero();
  __syncthreads();
  const float* w1 = sw; const float* w2 = w1 + KM * KM + KM; const float* wr = w2 + KM * KM + KM;
  const float* g = wr + KM * KM;
  float aw1[NK][NK], aw2[NK][NK], awr[NK][NK], ab1[NK], ab2[NK], ag[NK], abe[NK];
#pragma unroll
  for (int i = 0; i < NK; ++i) {
    ab1[i] = ab2[i] = ag[i] = abe[i] = 0.f;
#pragma unroll
    for (int j = 0; j < NK; ++j) aw1[i][j] = aw2[i][j] = awr[i][j] = 0.f;
  }
  KPHASE(1);
  const long total = (GRADS && (w.dbg & 1)) ? 0 : R * D;
  const bool pow2 = (D & (D - 1)) == 0; const int dsh = __ffs(D) - 1;
  // (tools/cube_phase.py, cfg2 first block, per workgroup: weights 2.2, element loop 14.4, wave reductions + LDS atomics 3.4, global atomics
  //  0.7 us.  The loop is VALU-bound at three waves per SIMD: requesting four elements per thread in one batch costs 58 registers, one wave
  //  per SIMD, and the loop went to 16.8 us)
  act_dispatch(w.act, [&](auto AT) __attribute__((always_inline)) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long r = pow2 ? (i >> dsh) : i / D; const int d = pow2 ? (int)(i & (D - 1)) : (int)(i % D);   // (64-bit division: ~80 instructions)
    KMixVals<NK> v;
    float dzv[NK], dyv[NK], dym[NK], du[NK], dxn[NK], dxv[NK];
#pragma unroll
    for (int k = 0; k < NK; ++k) {   // unconditional, clamped: a guarded load is a branch with its own vmcnt(0) -- 2 NK dependent round trips
      const float* xp = &x[(r * w.ik + (k < w.ik ? k : w.ik - 1)) * D + d];
      const float xv = (MODE == 2 && (w.dbg & 8)) ? __hip_atomic_load(xp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *xp;
      v.x[k] = k < w.ik ? xv : 0.f;
    }
#pragma unroll
    for (int o = 0; o < NK; ++o) {
      const float* gp = &dz[(r * w.ok + (o < w.ok ? o : w.ok - 1)) * D + d];
      const float gv = (MODE == 2 && (w.dbg & 16)) ? __hip_atomic_load(gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *gp;
      dzv[o] = o < w.ok ? gv : 0.f;
      v.sc[o] = o < w.ok ? drop_scale(w.drop_p, w.key, w.stream_id, (uint32_t)((r * w.ok + o) * D + d)) : 0.f;
    }
    kmix_forward_vals<NK, decltype(AT)::value>(w, sw, v);
    if (w.ln_first) {
#pragma unroll
      for (int o = 0; o < NK; ++o) dyv[o] = dzv[o];
    } else {
      float out[NK];
      ln_small<NK>(v.y, w.ok, g, g + KM, out, v.xh, v.mu, v.rs);
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int o = 0; o < NK; ++o) { const float t = dzv[o] * g[o]; s1 += t; s2 += t * v.xh[o]; }
      s1 /= w.ok; s2 /= w.ok;
#pragma unroll
      for (int o = 0; o < NK; ++o) {
        dyv[o] = o < w.ok ? v.rs * (dzv[o] * g[o] - s1 - v.xh[o] * s2) : 0.f;
        if (GRADS) {
          float term = dzv[o] * v.xh[o];
          // debugging (parked MODE 2 only): accumulate an intermediate instead -- which quantity is the first one that differs run to run?
          if (MODE == 2 && (w.dbg & (64 | 128 | 256))) term = (w.dbg & 64) ? v.x[o] : (w.dbg & 128) ? v.y[o] : v.u[o];
          ag[o] += term; abe[o] += dzv[o];
        }
      }
    }
#pragma unroll
    for (int o = 0; o < NK; ++o) dym[o] = dyv[o] * v.sc[o];   // gradient entering the (dropped-out) MLP branch
#pragma unroll
    for (int j = 0; j < NK; ++j) {
      float s = 0.f;
#pragma unroll
      for (int o = 0; o < NK; ++o) s += w2[o * KM + j] * dym[o];
      du[j] = j < w.hk ? s * act_grad_c<decltype(AT)::value>(w.act, v.u[j]) : 0.f;
    }
#pragma unroll
    for (int k = 0; k < NK; ++k) {
      float s = 0.f, rr = 0.f;
#pragma unroll
      for (int j = 0; j < NK; ++j) s += w1[j * KM + k] * du[j];
#pragma unroll
      for (int o = 0; o < NK; ++o) rr += wr[o * KM + k] * dyv[o];
      dxn[k] = s; dxv[k] = rr;
    }
    if (w.ln_first) {
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int k = 0; k < NK; ++k) { const float t = dxn[k] * g[k]; s1 += t; s2 += t * v.xh[k]; }
      s1 /= w.ik; s2 /= w.ik;
#pragma unroll
      for (int k = 0; k < NK; ++k) {
        dxv[k] += k < w.ik ? v.rs * (dxn[k] * g[k] - s1 - v.xh[k] * s2) : 0.f;
        if (GRADS) { ag[k] += dxn[k] * v.xh[k]; abe[k] += dxn[k]; }
      }
    } else {
#pragma unroll
      for (int k = 0; k < NK; ++k) dxv[k] += dxn[k];
    }
#pragma unroll
    for (int k = 0; k < NK; ++k) if (DX && k < w.ik) dx[(r * w.ik + k) * D + d] = dxv[k];
    if (!GRADS) continue;
#pragma unroll
    for (int o = 0; o < NK; ++o) {
      ab2[o] += dym[o];
#pragma unroll
      for (int j = 0; j < NK; ++j) aw2[o][j] += dym[o] * v.h[j];
#pragma unroll
      for (int k = 0; k < NK; ++k) awr[o][k] += dyv[o] * v.x[k];
    }
#pragma unroll
    for (int j = 0; j < NK; ++j) {
      ab1[j] += du[j];
#pragma unroll
      for (int k = 0; k < NK; ++k) aw1[j][k] += du[j] * (w.ln_first ? v.xn[k] : v.x[k]);
    }
  }
  });
  KPHASE(2);
  if (!GRADS) return;
  if (MODE == 2 && (w.dbg & 32)) {   // debugging: is the LDS copy of the weights still what was staged?  (+1000 on dbe[0] per mismatch)
    __syncthreads();
    const int t = threadIdx.x;
    int bad = 0;
    if (t < w.hk * w.ik && sw[(t / w.ik) * KM + t % w.ik] != w.w1[t]) ++bad;
    if (t < w.ok * w.hk && sw[KM * KM + KM + (t / w.hk) * KM + t % w.hk] != w.w2[t]) ++bad;
    if (w.wr && t < w.ok * w.ik && sw[2 * (KM * KM + KM) + (t / w.ik) * KM + t % w.ik] != w.wr[t]) ++bad;
    if (t < w.ok && (sw[3 * KM * KM + 2 * KM + t] != w.g[t] || sw[3 * KM * KM + 3 * KM + t] != w.be[t])) ++bad;
    if (bad) acc_add(&w.dbe[0], 1000.f * bad);
  }
  // one wave reduction + LDS + global atomic per scalar, once per kernel
  LdsAcc* gw1 = sg; LdsAcc* gb1 = gw1 + KM * KM; LdsAcc* gw2 = gb1 + KM; LdsAcc* gb2 = gw2 + KM * KM;
  LdsAcc* gwr = gb2 + KM; LdsAcc* gg = gwr + KM * KM; LdsAcc* gbe = gg + KM;
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int i = 0; i < NK; ++i) {
#pragma unroll
    for (int j = 0; j < NK; ++j) {
      float a = wave_sum(aw1[i][j]); if (lane == 0) gw1[i * KM + j].add(a);
      a = wave_sum(aw2[i][j]); if (lane == 0) gw2[i * KM + j].add(a);
      a = wave_sum(awr[i][j]); if (lane == 0) gwr[i * KM + j].add(a);
    }
    float a = wave_sum(ab1[i]); if (lane == 0) gb1[i].add(a);
    a = wave_sum(ab2[i]); if (lane == 0) gb2[i].add(a);
    a = wave_sum(ag[i]); if (lane == 0) gg[i].add(a);
    a = wave_sum(abe[i]); if (lane == 0) gbe[i].add(a);
  }
  KPHASE(3);
  __syncthreads();
  KPHASE(4);
  const int t = threadIdx.x;
  if (w.dbg & 2) return;
  if (t < w.hk * w.ik) acc_add(&w.dw1[t], gw1[(t / w.ik) * KM + t % w.ik].get());
  if (t < w.hk && w.db1) acc_add(&w.db1[t], gb1[t].get());
  if (t < w.ok * w.hk) acc_add(&w.dw2[t], gw2[(t / w.hk) * KM + t % w.hk].get());
  if (t < w.ok && w.db2) acc_add(&w.db2[t], gb2[t].get());
  if (t < w.ok * w.ik && w.dwr) acc_add(&w.dwr[t], gwr[(t / w.ik) * KM + t % w.ik].get());
  const int nln = w.ln_first ? w.ik : w.ok;
  if (t < nln) { acc_add(&w.dg[t], gg[t].get()); acc_add(&w.dbe[t], gbe[t].get()); }
  KPHASE(5);
}

// ------------------------------------------------------------------ reductions / misc
__global__ void colsum_kernel(const float* __restrict__ X, long M, int N, long ld, float* __restrict__ out, long xs,
                              long os) {
  __shared__ float red[4][64];
  X += blockIdx.z * xs;
  out += blockIdx.z * os;
  const int c = threadIdx.x & 63, rr = threadIdx.x >> 6;
  const int n = blockIdx.x * 64 + c;
  const long m0 = (long)blockIdx.y * 256;
  const long m1 = m0 + 256 < M ? m0 + 256 : M;
  float s = 0.f;
  if (n < N) for (long m = m0 + rr; m < m1; m += 4) s += X[m * ld + n];
  red[rr][c] = s;
  __syncthreads();
  if (rr == 0 && n < N) acc_add(&out[n], red[0][c] + red[1][c] + red[2][c] + red[3][c]);
}
__global__ void rowsum_batched_kernel(const float* __restrict__ X, int B, int R, int C, float* __restrict__ out) {
  __shared__ float red[16];
  const int r = blockIdx.x;
  const int b0 = blockIdx.y * 8, b1 = b0 + 8 < B ? b0 + 8 : B;
  float s = 0.f;
  for (int b = b0; b < b1; ++b) {
    const float* p = X + ((long)b * R + r) * C;
    for (int c = threadIdx.x; c < C; c += blockDim.x) s += p[c];
  }
  s = block_sum(s, red);
  if (threadIdx.x == 0) acc_add(&out[r], s);
}
__global__ void add_inplace_kernel(float* __restrict__ y, const float* __restrict__ x, long n) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) y[i] += x[i];
}
__global__ void dropout_inplace_kernel(float* __restrict__ y, long n, float p, RngKey key, uint32_t stream) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    y[i] *= drop_scale(p, key, stream, (uint32_t)i);
}

inline int grid_for(long n, int block = 256, int cap = 2048) {
  long g = (n + block - 1) / block;
  return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

__global__ void l0_pack_kernel(L0Pack a, int pack_inputs, int pack_weights) {
  const int m = blockIdx.y;
  const long nx = pack_inputs ? a.rows * a.KP : 0, nw = pack_weights ? 2L * 384 * a.KP : 0, nb = pack_weights ? 2L * 384 : 0;
  const long n1 = (pack_weights && a.w1h) ? 2L * 384 * 256 / 4 : 0;   // float4 pieces of this modality's two W_ih_l1
  const int d = a.d[m];
  if (a.bs_rng && blockIdx.x == 0 && m == 0) {
    if (threadIdx.x == 0) { *a.bs_rng += 1; if (a.bs_adam) *a.bs_adam += 1; }
    for (int i = threadIdx.x; i < a.bs_n; i += blockDim.x) a.bs_scal[a.bs_off + i] = 0.f;
  }
  if (a.xh) {   // 16-bit packed operands: four columns per thread (KP % 4 == 0), 8-byte stores
    const long nx4 = nx / 4, nw4 = nw / 4;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < nx4 + nw4 + nb + n1; i += (long)gridDim.x * blockDim.x) {
      if (i < nx4) {
        const long e = i * 4; const unsigned r32 = (unsigned)e / (unsigned)a.KP; const long r = r32; const int c = (int)((unsigned)e - r32 * (unsigned)a.KP);   // (l0_pack(): rows * KP < 2^31)
        const float* src = a.x[m] + r * d;
        float v[4];
        // (unconditional loads from clamped columns, masked afterwards: `c + q < d ? src[q] : 0` is four branches, each with its own vmcnt(0))
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = src[min(c + q, d - 1)];
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = c + q < d ? v[q] : 0.f;
        f16x4 h; bf16x4 b;
#pragma unroll
        for (int q = 0; q < 4; ++q) { h[q] = to_f16_sat(v[q]); b[q] = to_bf16(v[q]); }
        *reinterpret_cast<f16x4*>(a.xh + (long)m * a.rows * a.KP + e) = h;
        *reinterpret_cast<bf16x4*>(a.xb + (long)m * a.rows * a.KP + e) = b;
      } else if (i < nx4 + nw4) {
        const long e = (i - nx4) * 4; const int dir = (int)(e / (384L * a.KP)); const long q0 = e - dir * 384L * a.KP;
        const int r = (int)(q0 / a.KP), c = (int)(q0 - (long)r * a.KP);
        const float* src = a.w_ih[m][dir] + (long)r * d + c;
        f16x4 h;
#pragma unroll
        for (int q = 0; q < 4; ++q) h[q] = to_f16_sat(c + q < d ? src[q] : 0.f);
        *reinterpret_cast<f16x4*>(a.wh + ((long)m * 2 + dir) * 384 * a.KP + q0) = h;
      } else if (i < nx4 + nw4 + nb) {
        const long j = i - nx4 - nw4; const int dir = (int)(j / 384), r = (int)(j - dir * 384L);
        a.bpack[((long)m * 2 + dir) * 384 + r] = a.b_ih[m][dir][r];
      } else {
        const long j = i - nx4 - nw4 - nb; const int dir = (int)(j / (384L * 256 / 4)); const long q = j - dir * (384L * 256 / 4);
        const float4 v = reinterpret_cast<const float4*>(a.w_ih1[m][dir])[q];
        const long o = ((long)m * 2 + dir) * 384 * 256 + q * 4;
        f16x4 h; h[0] = to_f16_sat(v.x); h[1] = to_f16_sat(v.y); h[2] = to_f16_sat(v.z); h[3] = to_f16_sat(v.w);
        bf16x4 b; b[0] = to_bf16(v.x); b[1] = to_bf16(v.y); b[2] = to_bf16(v.z); b[3] = to_bf16(v.w);
        *reinterpret_cast<f16x4*>(a.w1h + o) = h;
        *reinterpret_cast<bf16x4*>(a.w1b + o) = b;
        if (a.w1bt) {   // transposed bf16 image [modality][n = 256][k = direction * 384 + g]: the k-contiguous B operand of the tall dh0 product
          const int gg = (int)((q * 4) >> 8), n0 = (int)((q * 4) & 255);
          __bf16* t = a.w1bt + (long)m * 256 * 768 + (long)n0 * 768 + dir * 384 + gg;
#pragma unroll
          for (int e = 0; e < 4; ++e) t[e * 768] = b[e];
        }
      }
    }
    return;
  }
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < nx + nw + nb + n1; i += (long)gridDim.x * blockDim.x) {
    if (i >= nx + nw + nb) {
      const long j = i - nx - nw - nb; const int dir = (int)(j / (384L * 256 / 4)); const long q = j - dir * (384L * 256 / 4);
      const float4 v = reinterpret_cast<const float4*>(a.w_ih1[m][dir])[q];
      const long o = ((long)m * 2 + dir) * 384 * 256 + q * 4;
      f16x4 h; h[0] = to_f16_sat(v.x); h[1] = to_f16_sat(v.y); h[2] = to_f16_sat(v.z); h[3] = to_f16_sat(v.w);
      bf16x4 b; b[0] = to_bf16(v.x); b[1] = to_bf16(v.y); b[2] = to_bf16(v.z); b[3] = to_bf16(v.w);
      *reinterpret_cast<f16x4*>(a.w1h + o) = h;
      *reinterpret_cast<bf16x4*>(a.w1b + o) = b;
      if (a.w1bt) {
        const int gg = (int)((q * 4) >> 8), n0 = (int)((q * 4) & 255);
        __bf16* t = a.w1bt + (long)m * 256 * 768 + (long)n0 * 768 + dir * 384 + gg;
#pragma unroll
        for (int e = 0; e < 4; ++e) t[e * 768] = b[e];
      }
    } else if (i < nx) {
      const long r = i / a.KP; const int c = (int)(i - r * a.KP);
      a.xpack[(long)m * a.rows * a.KP + i] = c < d ? a.x[m][r * d + c] : 0.f;
    } else if (i < nx + nw) {
      const long j = i - nx; const int dir = (int)(j / (384L * a.KP)); const long q = j - dir * 384L * a.KP;
      const int r = (int)(q / a.KP), c = (int)(q - (long)r * a.KP);
      a.wpack[((long)m * 2 + dir) * 384 * a.KP + q] = c < d ? a.w_ih[m][dir][(long)r * d + c] : 0.f;
    } else {
      const long j = i - nx - nw; const int dir = (int)(j / 384), r = (int)(j - dir * 384L);
      a.bpack[((long)m * 2 + dir) * 384 + r] = a.b_ih[m][dir][r];
    }
  }
}

__global__ void pad_rows_kernel(const float* __restrict__ src, float* __restrict__ dst, int R, int C, int Cp) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < (long)R * Cp; i += (long)gridDim.x * blockDim.x) {
    const int r = (int)(i / Cp), c = (int)(i - (long)r * Cp);
    dst[i] = c < C ? src[(long)r * C + c] : 0.f;
  }
}

__global__ void l0_unpack_kernel(L0Unpack a) {
  const int md = blockIdx.y, m = md >> 1, dir = md & 1;
  const int d = a.d[m];
  const long nih = 384L * a.KP, nhh = 384L * 128;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < nih + nhh; i += (long)gridDim.x * blockDim.x) {
    if (i < nih) {
      const int r = (int)(i / a.KP), c = (int)(i - (long)r * a.KP);
      float* src = a.dwih_pack + (long)md * nih + i;
      if (c < d) a.g_ih[m][dir][(long)r * d + c] += *src;
      *src = 0.f;
    } else {
      const long j = i - nih;
      float* src = a.dwhh_pack + (long)md * nhh + j;
      a.g_hh[m][dir][j] += *src;
      *src = 0.f;
    }
  }
}

}  // namespace

int pad_rows(hipStream_t s, const float* src, float* dst, int R, int C, int Cp) {
  hipLaunchKernelGGL(pad_rows_kernel, dim3(grid_for((long)R * Cp, 256, 64)), dim3(256), 0, s, src, dst, R, C, Cp);
  LAUNCH_CHECK();
  return MIMRL_OK;
}
int l0_pack(hipStream_t s, const L0Pack& a, bool pack_inputs, bool pack_weights) {
  if (a.xh && (!a.xb || !a.wh || !a.w1h || !pack_inputs || !pack_weights || a.KP % 4 != 0))
    return set_error(MIMRL_ERR_ARG, "l0_pack: the 16-bit packed operands come as a set (inputs + weights + the layer-1 images)");
  const long n = (pack_inputs ? a.rows * a.KP : 0) + (pack_weights ? 2L * 384 * a.KP + 2L * 384 : 0) + ((pack_weights && a.w1h) ? 2L * 384 * 256 / 4 : 0);
  if (n <= 0) return MIMRL_OK;
  if (a.xh && a.rows * a.KP >= (1L << 31)) return set_error(MIMRL_ERR_ARG, "l0_pack: rows * KP must stay below 2^31 (32-bit piece indices)");
  hipLaunchKernelGGL(l0_pack_kernel, dim3(grid_for(n, 256, 1024), 2), dim3(256), 0, s, a, pack_inputs ? 1 : 0, pack_weights ? 1 : 0);
  LAUNCH_CHECK();
  return MIMRL_OK;
}
int l0_unpack_grads(hipStream_t s, const L0Unpack& a) {
  hipLaunchKernelGGL(l0_unpack_kernel, dim3(grid_for(384L * (a.KP + 128), 256, 256), 4), dim3(256), 0, s, a);
  LAUNCH_CHECK();
  return MIMRL_OK;
}

int seq_lengths(hipStream_t s, const float* x, int B, int T, int d, int* lens) {
  hipLaunchKernelGGL(seq_lengths_kernel, dim3(B), dim3(256), 0, s, x, T, d, lens, (const float*)nullptr, 0, (int*)nullptr);
  LAUNCH_CHECK();
  return MIMRL_OK;
}
int seq_lengths2(hipStream_t s, const float* xa, int da, int* lens_a, const float* xv, int dv, int* lens_v, int B, int T) {
  hipLaunchKernelGGL(seq_lengths_kernel, dim3(B, 2), dim3(1024), 0, s, xa, T, da, lens_a, xv, dv, lens_v);
  LAUNCH_CHECK();
  return MIMRL_OK;
}

int text_post_fwd(hipStream_t s, const float* src, float* cube, int B, int T, int L, int K, int D, int slot, float p,
                  RngKey key, uint32_t stream_id) {
  const long n = (long)B * T * D;
  hipLaunchKernelGGL(text_post_kernel, dim3(grid_for(n)), dim3(256), 0, s, src, cube, n, T, L, K, D, slot, p, key,
                     stream_id, 0, (const float*)nullptr);
  LAUNCH_CHECK();
  return MIMRL_OK;
}
int text_post_bwd(hipStream_t s, const float* dcube, float* dsrc, int B, int T, int L, int K, int D, int slot, float p,
                  RngKey key, uint32_t stream_id, const float* dmean) {
  const long n = (long)B * T * D;
  hipLaunchKernelGGL(text_post_kernel, dim3(grid_for(n)), dim3(256), 0, s, dcube, dsrc, n, T, L, K, D, slot, p, key,
                     stream_id, 1, dmean);
  LAUNCH_CHECK();
  return MIMRL_OK;
}

int ln_relu_drop_fwd(hipStream_t s, const float* h2, const float* gamma, const float* beta, float* cube, float* mean,
                     float* rstd, int B, int T, int L, int K, int D, int slot, float p, RngKey key, uint32_t stream_id) {
  if (D != 128) return set_error(MIMRL_ERR_ARG, "ln_relu_drop: d_common must be 128 (got %d)", D);
  const long rows = (long)B * T;
  const LnSide a{h2, gamma, beta, mean, rstd, nullptr, nullptr, nullptr, slot, p, stream_id, nullptr};
  hipLaunchKernelGGL(ln_relu_drop_fwd_kernel<2>, dim3(grid_for(rows * 64)), dim3(256), 0, s, a, a, cube, rows, T, L, K, key);
  LAUNCH_CHECK();
  return MIMRL_OK;
}
int ln_relu_drop_bwd(hipStream_t s, const float* h2, const float* gamma, const float* beta, const float* mean,
                     const float* rstd, const float* dcube, float* ds, float* dgamma, float* dbeta, int B, int T, int L,
                     int K, int D, int slot, float p, RngKey key, uint32_t stream_id) {
  if (D != 128) return set_error(MIMRL_ERR_ARG, "ln_relu_drop: d_common must be 128 (got %d)", D);
  const long rows = (long)B * T;
  const LnSide a{h2, gamma, beta, const_cast<float*>(mean), const_cast<float*>(rstd), ds, dgamma, dbeta, slot, p, stream_id, nullptr};
  hipLaunchKernelGGL(ln_relu_drop_bwd16_kernel, dim3((unsigned)std::min<long>((rows + 15) / 16, 1024)), dim3(256), 0, s, a, a, dcube, rows, T, L, K, key);
  LAUNCH_CHECK();
  return MIMRL_OK;
}
// audio and video in one launch (grid.y = modality)
int ln_relu_drop_fwd2(hipStream_t s, const LnSide2& a, const LnSide2& v, float* cube, int B, int T, int L, int K, int D,
                      RngKey key) {
  if (D != 128) return set_error(MIMRL_ERR_ARG, "ln_relu_drop: d_common must be 128 (got %d)", D);
  const long rows = (long)B * T;
  const LnSide sa{a.h2, a.gamma, a.beta, a.mean, a.rstd, a.ds, a.dgamma, a.dbeta, a.slot, a.p, a.stream, nullptr};
  const LnSide sv{v.h2, v.gamma, v.beta, v.mean, v.rstd, v.ds, v.dgamma, v.dbeta, v.slot, v.p, v.stream, nullptr};
  hipLaunchKernelGGL(ln_relu_drop_fwd_kernel<2>, dim3(grid_for(rows * 64), 2), dim3(256), 0, s, sa, sv, cube, rows, T, L, K, key);
  LAUNCH_CHECK();
  return MIMRL_OK;
}
int ln_relu_drop_bwd2(hipStream_t s, const LnSide2& a, const LnSide2& v, const float* dcube, int B, int T, int L, int K, int D,
                      RngKey key, const float* dmean_a, const float* dmean_v, int ds_bf16) {
  if (D != 128) return set_error(MIMRL_ERR_ARG, "ln_relu_drop: d_common must be 128 (got %d)", D);
  const long rows = (long)B * T;
  const LnSide sa{a.h2, a.gamma, a.beta, a.mean, a.rstd, a.ds, a.dgamma, a.dbeta, a.slot, a.p, a.stream, dmean_a, ds_bf16};
  const LnSide sv{v.h2, v.gamma, v.beta, v.mean, v.rstd, v.ds, v.dgamma, v.dbeta, v.slot, v.p, v.stream, dmean_v, ds_bf16};
  // workgroups per modality (cfg2: 400 -> 0.935, 200 -> 0.929, 100 -> 0.927, 50 -> 0.939 ms/step; round 5b, long inputs: 128 workgroups per
  // modality are one wave per SIMD -- 143 us at cfg3's 128 000 rows; 256 -> 139, 512 -> 121, 1024 -> 129 us: more workgroups hide more
  // latency, and every workgroup ends in 256 same-address float atomics)
  const int cap = rows > 16384 ? 512 : 128;
  hipLaunchKernelGGL(ln_relu_drop_bwd16_kernel, dim3((unsigned)std::min<long>((rows + 15) / 16, cap), 2), dim3(256), 0, s, sa, sv, dcube, rows, T, L, K, key);
  LAUNCH_CHECK();
  return MIMRL_OK;
}

int tail_pre_fwd(hipStream_t s, const float* tx_raw, float p_text, const LnSide2& a, const LnSide2& v, float* cube, float* feats, int B, int T,
                 int L, int K, int D, RngKey key) {
  if (D != 128 || K != 3 || a.slot != 1 || v.slot != 2) return set_error(MIMRL_ERR_ARG, "tail_pre_fwd: d_common 128, 3 modality slots");
  TailPre tp;
  tp.tx_raw = tx_raw; tp.p_text = p_text;
  tp.a = LnSide{a.h2, a.gamma, a.beta, a.mean, a.rstd, a.ds, a.dgamma, a.dbeta, a.slot, a.p, a.stream, nullptr};
  tp.v = LnSide{v.h2, v.gamma, v.beta, v.mean, v.rstd, v.ds, v.dgamma, v.dbeta, v.slot, v.p, v.stream, nullptr};
  hipLaunchKernelGGL(tail_pre_kernel, dim3(B, 3), dim3(256), 0, s, tp, cube, feats, B, T, L, K, key);
  LAUNCH_CHECK();
  return MIMRL_OK;
}

void tail_pre2_chunks(int B, int T, int* nchunk, int* rpc) {
  int n = (1024 + 3 * B - 1) / (3 * B);                     // ~1024 workgroups (4 per CU) ...
  n = std::max(1, std::min(n, std::min(16, (T + 31) / 32)));   // ... of at least 32 rows
  const int r = ((T + n - 1) / n + 3) & ~3;                  // rows per chunk, a multiple of 4 (the waves' time phases stay aligned)
  *nchunk = (T + r - 1) / r; *rpc = r;
}
int tail_pre2_fwd(hipStream_t s, const float* tx_raw, float p_text, const LnSide2& a, const LnSide2& v, float* const cube[2], float* const feats[2],
                  const int add[2], float* part, int B, int T, int L, int K, int D, RngKey key) {
  if (D != 128 || K != 3 || a.slot != 1 || v.slot != 2) return set_error(MIMRL_ERR_ARG, "tail_pre2_fwd: d_common 128, 3 modality slots");
  TailPre2 tp;
  tp.tx_raw = tx_raw; tp.p_text = p_text;
  tp.a = LnSide{a.h2, a.gamma, a.beta, a.mean, a.rstd, a.ds, a.dgamma, a.dbeta, a.slot, a.p, a.stream, nullptr};
  tp.v = LnSide{v.h2, v.gamma, v.beta, v.mean, v.rstd, v.ds, v.dgamma, v.dbeta, v.slot, v.p, v.stream, nullptr};
  for (int o = 0; o < 2; ++o) { tp.cube[o] = cube[o]; tp.feats[o] = feats[o]; tp.add[o] = add[o]; }
  tail_pre2_chunks(B, T, &tp.nchunk, &tp.rpc);
  tp.part = part;
  if (tp.nchunk > 1 && !part) return set_error(MIMRL_ERR_ARG, "tail_pre2_fwd: partial-sum scratch missing");
  hipLaunchKernelGGL(tail_pre2_kernel, dim3(B, 3, tp.nchunk), dim3(256), 0, s, tp, B, T, L, K, key);
  LAUNCH_CHECK();
  if (tp.nchunk > 1) {
    hipLaunchKernelGGL(tail_pre2_finish_kernel, dim3((unsigned)((2L * 3 * B * D + 255) / 256)), dim3(256), 0, s, tp, B, T);
    LAUNCH_CHECK();
  }
  return MIMRL_OK;
}

int feat_mean_fwd(hipStream_t s, const float* cube, float* feats, int B, int T, int L, int K, int D) {
  hipLaunchKernelGGL(feat_mean_fwd_kernel, dim3(B, K), dim3(512), 0, s, cube, feats, B, T, L, K, D);
  LAUNCH_CHECK();
  return MIMRL_OK;
}
int feat_mean_bwd(hipStream_t s, const float* dfeats, float* dcube, int B, int T, int L, int K, int D) {
  const long n = (long)B * T * K * D;
  hipLaunchKernelGGL(feat_mean_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, s, dfeats, dcube, n, B, T, L, K, D);
  LAUNCH_CHECK();
  return MIMRL_OK;
}

int head_fwd(hipStream_t s, const float* x, const float* w, const float* bias, float* ff, float* pred, int B, int L,
             int K, int D, int sum_l, int sum_k) {
  const float scale = (sum_l ? 1.f : 1.f / L) * (sum_k ? 1.f : 1.f / K);
  hipLaunchKernelGGL(head_fwd_kernel, dim3(B), dim3(D == 128 ? 512 : 128), 0, s, x, w, bias, ff, pred, L, K, D, scale);
  LAUNCH_CHECK();
  return MIMRL_OK;
}
int head_bwd(hipStream_t s, const float* dff_ext, const float* dpred, const float* w, const float* ff, float* dx,
             float* dw, float* dbias, int B, int L, int K, int D, int sum_l, int sum_k, const GatherSum* gather) {
  const float scale = (sum_l ? 1.f : 1.f / L) * (sum_k ? 1.f : 1.f / K);
  GatherSum gs;
  std::memset(&gs, 0, sizeof gs);
  if (gather) gs = *gather;
  hipLaunchKernelGGL(head_bwd_kernel, dim3(B), dim3(128), 0, s, dff_ext, dpred, w, ff, dx, dw, dbias, L, K, D, scale, gs, gather ? 1 : 0);
  LAUNCH_CHECK();
  return MIMRL_OK;
}

int rowln_fwd(hipStream_t s, const float* y, const float* gamma, const float* beta, float* z, float* mean, float* rstd,
              long R, int n) {
  hipLaunchKernelGGL(rowln_fwd_kernel, dim3(grid_for(R * 64)), dim3(256), 0, s, y, gamma, beta, z, mean, rstd, R, n);
  LAUNCH_CHECK();
  return MIMRL_OK;
}
int rowln_bwd(hipStream_t s, const float* y, const float* gamma, const float* mean, const float* rstd, const float* dz,
              float* dy, float* dgamma, float* dbeta, long R, int n) {
  if (n > 512) return set_error(MIMRL_ERR_ARG, "rowln_bwd: row length %d > 512", n);
  if (n <= 128)
    hipLaunchKernelGGL(rowln_bwd_kernel<2>, dim3(grid_for(R * 64, 256, 512)), dim3(256), 0, s, y, gamma, mean, rstd, dz, dy,
                       dgamma, dbeta, R, n);
  else
    hipLaunchKernelGGL(rowln_bwd_kernel<8>, dim3(grid_for(R * 64, 256, 512)), dim3(256), 0, s, y, gamma, mean, rstd, dz, dy,
                       dgamma, dbeta, R, n);
  LAUNCH_CHECK();
  return MIMRL_OK;
}
int colln_fwd(hipStream_t s, const float* y, const float* gamma, const float* beta, float* z, float* mean, float* rstd,
              int B, int n, int C) {
  const long tot = (long)B * C;
  hipLaunchKernelGGL(colln_fwd_kernel, dim3((tot + 255) / 256), dim3(256), 0, s, y, gamma, beta, z, mean, rstd, B, n, C);
  LAUNCH_CHECK();
  return MIMRL_OK;
}
// LayerNorm(L) parameter gradients as plain row sums over (sample, column): dgamma[l] = sum dz * xhat, dbeta[l] = sum dz.
__global__ void colln_param_grads_kernel(const float* __restrict__ y, const float* __restrict__ mean,
                                         const float* __restrict__ rstd, const float* __restrict__ dz,
                                         float* __restrict__ dgamma, float* __restrict__ dbeta, int B, int n, int C) {
  __shared__ float red[16];
  const int l = blockIdx.x;
  float sg = 0.f, sb = 0.f;
  for (int b = blockIdx.y; b < B; b += gridDim.y) {
    const float* yb = y + ((long)b * n + l) * C;
    const float* dzb = dz + ((long)b * n + l) * C;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
      const float g = dzb[c];
      sg += g * (yb[c] - mean[(long)b * C + c]) * rstd[(long)b * C + c];
      sb += g;
    }
  }
  sg = block_sum(sg, red);
  sb = block_sum(sb, red);
  if (threadIdx.x == 0) { acc_add(&dgamma[l], sg); acc_add(&dbeta[l], sb); }
}
int colln_param_grads(hipStream_t s, const float* y, const float* mean, const float* rstd, const float* dz, float* dgamma,
                      float* dbeta, int B, int n, int C) {
  hipLaunchKernelGGL(colln_param_grads_kernel, dim3(n, (B + 7) / 8), dim3(128), 0, s, y, mean, rstd, dz, dgamma, dbeta, B, n, C);
  LAUNCH_CHECK();
  return MIMRL_OK;
}

int colln_bwd(hipStream_t s, const float* y, const float* gamma, const float* mean, const float* rstd, const float* dz,
              float* dy, float* dgamma, float* dbeta, int B, int n, int C) {
  const long tot = (long)B * C;
  if (n > 64) return set_error(MIMRL_ERR_ARG, "colln_bwd: axis length %d > 64", n);
  if (C % 64 != 0) return set_error(MIMRL_ERR_ARG, "colln_bwd: %d columns per sample is not a multiple of 64", C);
  hipLaunchKernelGGL(colln_bwd_kernel, dim3(tot / 64), dim3(256), (2 * n + 8) * 64 * sizeof(float), s, y, gamma, mean,
                     rstd, dz, dy, dgamma, dbeta, B, n, C);
  LAUNCH_CHECK();
  return MIMRL_OK;
}

#ifdef MIMRL_PHASE_PROBE
int kmix_bwd_read_phases(long long* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_kmix_phase), sizeof(long long) * 16) == hipSuccess ? 0 : 1; }
int model_ops_read_phases(long long* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_mo_phase), sizeof(long long) * 16) == hipSuccess ? 0 : 1; }
#endif
int kmix_fwd(hipStream_t s, const float* x, float* z, KMixW w, long R, int D) {
  if (w.ik > KM || w.hk > KM || w.ok > KM) return set_error(MIMRL_ERR_ARG, "kmix: K-axis sizes must be <= %d", KM);
  if (!w.wr && w.ik != w.ok) return set_error(MIMRL_ERR_ARG, "kmix: identity residual needs ik == ok");
  const int mx = w.ik > w.hk ? (w.ik > w.ok ? w.ik : w.ok) : (w.hk > w.ok ? w.hk : w.ok);
  // (array sizes = loop trip counts of the per-element MLP: the model's K = 3 padded to 4 is 16 products where 9 are needed, and these
  //  kernels are VALU-bound)
  if (mx <= 3) hipLaunchKernelGGL(kmix_fwd_kernel<3>, dim3(grid_for(R * D)), dim3(256), 0, s, x, z, w, R, D);
  else if (mx <= 4) hipLaunchKernelGGL(kmix_fwd_kernel<4>, dim3(grid_for(R * D)), dim3(256), 0, s, x, z, w, R, D);
  else hipLaunchKernelGGL(kmix_fwd_kernel<8>, dim3(grid_for(R * D)), dim3(256), 0, s, x, z, w, R, D);
  LAUNCH_CHECK();
  return MIMRL_OK;
}
int kmix_bwd(hipStream_t s, const float* x, const float* dz, float* dx, KMixW w, long R, int D) {
  if (w.ik > KM || w.hk > KM || w.ok > KM) return set_error(MIMRL_ERR_ARG, "kmix: K-axis sizes must be <= %d", KM);
  const int mx = w.ik > w.hk ? (w.ik > w.ok ? w.ik : w.ok) : (w.hk > w.ok ? w.hk : w.ok);
  constexpr int wgs = 512;   // (an environment knob until round 5: fixed at its measured optimum)
  const bool exact3 = w.ik == 3 && w.hk == 3 && w.ok == 3 && !w.ln_first && w.drop_p <= 0.f;
  if (exact3) hipLaunchKernelGGL((kmix_bwd_kernel<3, 0, true>), dim3(grid_for(R * D, 256, wgs)), dim3(256), 0, s, x, dz, dx, w, R, D);
  else if (mx <= 3) hipLaunchKernelGGL((kmix_bwd_kernel<3, 0>), dim3(grid_for(R * D, 256, wgs)), dim3(256), 0, s, x, dz, dx, w, R, D);
  else if (mx <= 4) hipLaunchKernelGGL((kmix_bwd_kernel<4, 0>), dim3(grid_for(R * D, 256, wgs)), dim3(256), 0, s, x, dz, dx, w, R, D);
  else hipLaunchKernelGGL((kmix_bwd_kernel<8, 0>), dim3(grid_for(R * D, 256, wgs)), dim3(256), 0, s, x, dz, dx, w, R, D);
  LAUNCH_CHECK();
  return MIMRL_OK;
}
// the two halves of kmix_bwd as separate launches: `part` 1 = dx only (many workgroups, nothing shared), 2 = parameter
// gradients only (few workgroups: its cost is the 64 reductions + atomics per workgroup)
int kmix_bwd_part(hipStream_t s, const float* x, const float* dz, float* dx, KMixW w, long R, int D, int part) {
  if (w.ik > KM || w.hk > KM || w.ok > KM) return set_error(MIMRL_ERR_ARG, "kmix: K-axis sizes must be <= %d", KM);
  const int mx = w.ik > w.hk ? (w.ik > w.ok ? w.ik : w.ok) : (w.hk > w.ok ? w.hk : w.ok);
  if (part == 1) {
    constexpr int dx_wgs = 4096;   // (an environment knob until round 5: fixed at its measured optimum)
    const dim3 grid(grid_for(R * D, 256, dx_wgs));
    if (mx <= 3) hipLaunchKernelGGL((kmix_bwd_kernel<3, 1>), grid, dim3(256), 0, s, x, dz, dx, w, R, D);
    else if (mx <= 4) hipLaunchKernelGGL((kmix_bwd_kernel<4, 1>), grid, dim3(256), 0, s, x, dz, dx, w, R, D);
    else hipLaunchKernelGGL((kmix_bwd_kernel<8, 1>), grid, dim3(256), 0, s, x, dz, dx, w, R, D);
  } else {
    // workgroup count (tuning knob).  The kernel is ~33 us of fixed cost (weight staging, 64 wave reductions, LDS and global
    // atomics) + ~25 us of element work at cfg2; 256 workgroups is the measured optimum (128: 67 us, 256: 53 us, 512: 75 us).  A
    // two-level reduction through scratch slots with a last-arriver finisher was tried and lost (the __threadfence it needs is an
    // L2 write-back on this multi-XCD part: 157 us).
    constexpr int pg_wgs = 256;
    const int cap = pg_wgs > 0 ? pg_wgs : 256;
    const dim3 grid(grid_for(R * D, 256, cap));
    if (mx <= 3) hipLaunchKernelGGL((kmix_bwd_kernel<3, 2>), grid, dim3(256), 0, s, x, dz, dx, w, R, D);
    else if (mx <= 4) hipLaunchKernelGGL((kmix_bwd_kernel<4, 2>), grid, dim3(256), 0, s, x, dz, dx, w, R, D);
    else hipLaunchKernelGGL((kmix_bwd_kernel<8, 2>), grid, dim3(256), 0, s, x, dz, dx, w, R, D);
  }
  LAUNCH_CHECK();
  return MIMRL_OK;
}

int colsum(hipStream_t s, const float* X, long M, int N, long ld, float* out, int batch, long xs, long os) {
  dim3 grid((N + 63) / 64, (unsigned)((M + 255) / 256), batch);
  hipLaunchKernelGGL(colsum_kernel, grid, dim3(256), 0, s, X, M, N, ld, out, xs, os);
  LAUNCH_CHECK();
  return MIMRL_OK;
}
int rowsum_batched(hipStream_t s, const float* X, int B, int R, int C, float* out) {
  hipLaunchKernelGGL(rowsum_batched_kernel, dim3(R, (B + 7) / 8), dim3(256), 0, s, X, B, R, C, out);
  LAUNCH_CHECK();
  return MIMRL_OK;
}
int add_inplace(hipStream_t s, float* y, const float* x, long n) {
  hipLaunchKernelGGL(add_inplace_kernel, dim3(grid_for(n)), dim3(256), 0, s, y, x, n);
  LAUNCH_CHECK();
  return MIMRL_OK;
}
int dropout_inplace(hipStream_t s, float* y, long n, float p, RngKey key, uint32_t stream_id) {
  if (p <= 0.f) return MIMRL_OK;
  hipLaunchKernelGGL(dropout_inplace_kernel, dim3(grid_for(n)), dim3(256), 0, s, y, n, p, key, stream_id);
  LAUNCH_CHECK();
  return MIMRL_OK;
}

}  // namespace mimrl
