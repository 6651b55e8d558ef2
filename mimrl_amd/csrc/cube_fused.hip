// Fused CubeMLP block forward for gfx950 (see cube_fused.h).  16-bit MFMA operands, fp32 accumulation / LayerNorm.
//
// OPERAND TYPE: fp16, not bf16 (round 3).  The MFMA rate, the LDS footprint and every layout are the same, but fp16 keeps 11
// significant bits instead of 8, and this block is where the model is sensitive: the head sends a gradient that is a broadcast over
// (l, k), each LayerNorm of the backward pass cancels ~90 % of it, and what survives is decided by the forward values (act'(U), x-hat)
// -- a 2^-9 perturbation of them moved individual gradient entries by 10-70 % and the main-bucket gradient to cosine 0.964 against
// fp32 (float64 experiment per rounding point: weights 0.985, tile operands 0.992, tile write-backs 0.995; all in fp16: 0.9998).
// Range is not a concern on this path: every operand is a LayerNorm output, an activation of one, a weight, or the block input
// (text projection / relu(LN(gru))), which is converted with saturation.  Gradient operands (the backward kernels) stay bf16.
//
// One workgroup (4 waves) per sample.  The sample tile x[L, K, 128] (<= 64 x 512) is loaded once into LDS as bf16 and
// never leaves the CU until the block output is written:
//   phase L : per 64-column slab of the tile (columns are independent under the L-axis contraction)
//               U = W1.X + b1 ; H = act(U) ; Y = W2.H + Wr.X + b2 ; Z = LayerNorm over L  -> written back in place
//             W1/W2/Wr live in LDS as MFMA A-images for the whole phase; X^T and H^T slabs are the B-images (the
//             epilogue of the first product writes H directly in the layout the second product reads).
//   phase K : the 3x3-class K-axis MLP + LayerNorm over K, elementwise per (l, d), in place.
//   phase D : rows (l,k) x 128: H = act(Z W1^T + b1) ; Y = H W2^T + Z Wr^T + b2 ; LayerNorm over D.
//             All row tiles (<= 3 x 64 rows) stay resident as A-images, so each 64x64 weight image is staged from
//             L2 exactly once per workgroup (12 stagings per block).
// Everything the (unfused) backward pass needs is written out in fp32 with the unfused path's layouts.
#include "cube_fused.h"

#include "kmix_device.h"

#include <type_traits>

namespace mimrl {

namespace {

constexpr int D = 128;
constexpr int ILD = 72;                  // image row length (bf16): 64 + 8 -> 144-B rows, conflict-free ds_read_b128
constexpr int IMG = 64 * ILD;            // elements per 64x64 image
constexpr int CTL = 68, CTD = 132;       // fp32 staging tile row lengths

typedef _Float16 bf;   // the tile / operand element type of this kernel: fp16 (see the header comment)

// `make PHASE_PROBE=1` (tools/cube_phase.py): workgroup 0 leaves 100 MHz ticks at the phase boundaries of its last launch
#ifdef MIMRL_PHASE_PROBE
__device__ long long g_cube_phase[128];
#define CPHASE(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_cube_phase[(SAVE ? 64 : 0) + (NMT - 1) * 16 + (i)] = (long long)wall_clock64(); } while (0)
#else
#define CPHASE(i) do { } while (0)
#endif

__device__ __forceinline__ void mma64(f32x16& acc, const bf* A, const bf* B, int wm, int wn, int lane) {
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const f16x8 a = *reinterpret_cast<const f16x8*>(A + (wm * 32 + (lane & 31)) * ILD + s * 16 + 8 * (lane >> 5));
    const f16x8 b = *reinterpret_cast<const f16x8*>(B + (wn * 32 + (lane & 31)) * ILD + s * 16 + 8 * (lane >> 5));
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
  }
}
__device__ __forceinline__ int acc_row(int r, int wm, int lane) { return wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// stage a 64(n) x 64(k) weight image  Bw[n][k] = W[(n0+n)*ld + k0 + k]  (fp32 global -> bf16 LDS), zero beyond (N, K): request (registers)
// and commit (LDS) are separate so that the kernel's whole set-up -- sample tile, per-row parameters, L-axis weights -- is ONE
// memory round trip (as a sequence it was five: 7.9 of block 1's 54 us, tools/cube_phase.py)
struct WReq { float v[16]; };
__device__ __forceinline__ WReq stage_weight_request(const float* __restrict__ W, int ld, int n0, int k0, int N, int K, int tid) {
  const int n = tid >> 2, kc = (tid & 3) * 16;
  // 16 UNCONDITIONAL loads from clamped addresses, zeroed afterwards: a guarded load is a branch whose join waits for every
  // outstanding load, i.e. 16 dependent round trips per image (this staging was 10 of the block's 18 us of set-up)
  const int gn = n0 + n, gnc = gn < N ? gn : N - 1;
  WReq r;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int gk = k0 + kc + j;
    r.v[j] = W[(long)gnc * ld + (gk < K ? gk : K - 1)];
  }
  return r;
}
__device__ __forceinline__ void stage_weight_commit(bf* img, const WReq& r, int n0, int k0, int N, int K, int tid) {
  const int n = tid >> 2, kc = (tid & 3) * 16;
  const int gn = n0 + n;
  f16x8 lo, hi;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int gk = k0 + kc + j;
    const float x = (gn < N && gk < K) ? r.v[j] : 0.f;
    if (j < 8) lo[j] = to_f16(x); else hi[j - 8] = to_f16(x);
  }
  *reinterpret_cast<f16x8*>(img + n * ILD + kc) = lo;
  *reinterpret_cast<f16x8*>(img + n * ILD + kc + 8) = hi;
}
__device__ __forceinline__ void stage_weight(bf* img, const float* __restrict__ W, int ld, int n0, int k0, int N, int K, int tid) {
  stage_weight_commit(img, stage_weight_request(W, ld, n0, k0, N, K, tid), n0, k0, N, K, tid);
}
// 128-wide row-major weights (D axis): a 64x64 image = 4 float4 per thread.  Split into issue (global -> registers)
// and commit (registers -> bf16 LDS image) so that the next image's loads are in flight during the current MFMAs.
struct WImg { float4 q0, q1, q2, q3; };
__device__ __forceinline__ WImg load_weight128(const float* __restrict__ W, int n0, int k0, int tid) {
  const int n = tid >> 2, kc = (tid & 3) * 16;
  const float4* src = reinterpret_cast<const float4*>(W + (long)(n0 + n) * D + k0 + kc);
  return WImg{src[0], src[1], src[2], src[3]};
}
__device__ __forceinline__ void commit_weight128(bf* img, const WImg& w, int tid) {
  const int n = tid >> 2, kc = (tid & 3) * 16;
  f16x8 lo, hi;
  lo[0] = to_f16(w.q0.x); lo[1] = to_f16(w.q0.y); lo[2] = to_f16(w.q0.z); lo[3] = to_f16(w.q0.w);
  lo[4] = to_f16(w.q1.x); lo[5] = to_f16(w.q1.y); lo[6] = to_f16(w.q1.z); lo[7] = to_f16(w.q1.w);
  hi[0] = to_f16(w.q2.x); hi[1] = to_f16(w.q2.y); hi[2] = to_f16(w.q2.z); hi[3] = to_f16(w.q2.w);
  hi[4] = to_f16(w.q3.x); hi[5] = to_f16(w.q3.y); hi[6] = to_f16(w.q3.z); hi[7] = to_f16(w.q3.w);
  *reinterpret_cast<f16x8*>(img + n * ILD + kc) = lo;
  *reinterpret_cast<f16x8*>(img + n * ILD + kc + 8) = hi;
}

struct Carve {          // byte offsets into dynamic LDS (all multiples of 16)
  int xm, aw, bx, bh, ctl, red, ksw, prm; // phases L / K   (bx, bh, ctl, red: one set per wave group)
  int bw, ctd, dprm, az, ah;              // phase D (aliases the L-phase regions; Xm is dead once Az is built)
  int total;
};
__host__ __device__ inline Carve carve(int K, int nmt, int G) {
  Carve c;
  const int C = K * D, XP = C + 8;
  const int xm_bytes = 64 * XP * 2;
  c.xm = 0;
  c.aw = xm_bytes;
  c.bx = c.aw + 3 * IMG * 2;
  c.bh = c.bx + G * IMG * 2;
  c.ctl = c.bh + G * IMG * 2;
  c.red = c.ctl + G * 64 * CTL * 4;
  c.ksw = c.red + G * 2 * 4 * 64 * 4;
  c.prm = c.ksw + 256 * 4;
  const int l_end = c.prm + 4 * 64 * 4;
  c.bw = 0;                                   // 2 weight-image buffers per wave group
  c.dprm = c.bw + 2 * G * IMG * 2;
  int after = c.dprm + 2 * D * 4;
  if (after < xm_bytes) after = xm_bytes;     // Az must not overlap Xm (it is built from it)
  c.az = (after + 15) & ~15;
  c.ah = c.az + nmt * 2 * IMG * 2;
  // the fp32 LayerNorm tile aliases the H images (dead once the second product has been accumulated)
  const int ah_bytes = nmt * 2 * IMG * 2, ctd_bytes = 64 * CTD * 4;
  c.ctd = c.ah;
  const int d_end = c.ah + (ah_bytes > ctd_bytes ? ah_bytes : ctd_bytes);
  c.total = l_end > d_end ? l_end : d_end;
  return c;
}

// G wave groups of 4 waves each (256 * G threads).  With one workgroup per CU (the tile fills the LDS) G = 1 leaves ONE wave per SIMD:
// every LDS / global round trip and every VALU dependency chain of the epilogues is exposed.  G = 2: the groups take alternate column
// slabs in phase L (own B-images and staging tiles), interleave the (l, d) pairs of phase K, and each owns one 64-column half of the
// D-axis outputs (own weight-image buffers: 6 stagings per group instead of 12 in a row).
template <bool SAVE, int NMT, int G>
__global__ __launch_bounds__(256 * G) void cube_fwd_fused_kernel(CubeFusedArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NT = 256 * G;
  const int tid = threadIdx.x, grp = tid >> 8, t = tid & 255, lane = t & 63, wave = t >> 6, wm = wave >> 1, wn = wave & 1;
  const int b = blockIdx.x;
  const int K = a.K, C = K * D, XP = C + 8, il = a.il, hl = a.hl, ol = a.ol;
  const Carve cv = carve(K, NMT, G);
  bf* Xm = reinterpret_cast<bf*>(smem + cv.xm);
  bf* Aw = reinterpret_cast<bf*>(smem + cv.aw);
  bf* Bx = reinterpret_cast<bf*>(smem + cv.bx) + grp * IMG;
  bf* Bh = reinterpret_cast<bf*>(smem + cv.bh) + grp * IMG;
  float* CtL = reinterpret_cast<float*>(smem + cv.ctl) + grp * 64 * CTL;
  float* red = reinterpret_cast<float*>(smem + cv.red) + grp * 2 * 4 * 64;
  float* ksw = reinterpret_cast<float*>(smem + cv.ksw);
  float* prm = reinterpret_cast<float*>(smem + cv.prm);   // [4][64] L-axis b1, b2, gamma, beta

  CPHASE(0);
  // ------------------------------------------------------------------ load the sample tile (bf16), zero-pad rows >= il
  {
    // every request first (L-axis weights, per-row parameters, the tile), then the commits: one round trip
    // group 0 (or the only group): W1 and Wr; group 1: W2.  ONE unconditional request per register set with the operands selected
    // beforehand: assigned in the two arms of an if / else, the merge copies the registers and waits for the loads right there
    const bool g0 = G == 1 || grp == 0;
    const WReq w0 = stage_weight_request(g0 ? a.l_w1 : a.l_w2, g0 ? il : hl, 0, 0, g0 ? hl : ol, g0 ? il : hl, t);
    const WReq w1 = stage_weight_request(a.l_wr, il, 0, 0, ol, il, t);     // (group 1 of two requests it too and drops it)
    WReq w2;
    if (G == 1) w2 = stage_weight_request(a.l_w2, hl, 0, 0, ol, hl, t);
    float pr[4] = {0.f, 0.f, 0.f, 0.f};
    if (tid < 64) {
      pr[0] = (tid < hl && a.l_b1) ? a.l_b1[tid] : 0.f;
      pr[1] = (tid < ol && a.l_b2) ? a.l_b2[tid] : 0.f;
      pr[2] = tid < ol ? a.l_g[tid] : 0.f;
      pr[3] = tid < ol ? a.l_be[tid] : 0.f;
    }
    CPHASE(13);
    const float* xb = a.x + (long)b * il * C;
    const int nq = C / 4, total = 64 * nq;
    constexpr int NQ = SAVE ? 6 : 12;   // (SAVE build: two passes -- twelve pieces in flight beside both weight request sets spill five registers)                                            // 16-byte pieces per thread and pass: K = 3 (C = 384), two wave groups: ONE pass
    for (int i0 = tid; i0 < total; i0 += NT * NQ) {
      float4 q[NQ];
      // UNCONDITIONAL loads from clamped addresses, zeroed afterwards (the rule of DESIGN.md section 4, violated here until round 3b: the
      // guarded form `cond ? *p : 0` compiled to twelve branches with an s_waitcnt vmcnt(0) behind each load -- twelve serialized
      // round trips, the 7 us this set-up took)
#pragma unroll
      for (int j = 0; j < NQ; ++j) {
        const int i = i0 + NT * j;
        const int ic = i < total ? i : total - 1;
        const int l = ic / nq, c4 = (ic - l * nq) * 4;
        q[j] = *reinterpret_cast<const float4*>(xb + (long)(l < il ? l : il - 1) * C + c4);
      }
#pragma unroll
      for (int j = 0; j < NQ; ++j) {
        const int i = i0 + NT * j;
        if (!(i < total && i / nq < il)) q[j] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int j = 0; j < NQ; ++j) {
        const int i = i0 + NT * j;
        if (i < total) {
          const int l = i / nq, c4 = (i - l * nq) * 4;
          f16x4 p; p[0] = to_f16_sat(q[j].x); p[1] = to_f16_sat(q[j].y); p[2] = to_f16_sat(q[j].z); p[3] = to_f16_sat(q[j].w);   // (caller data: saturate)
          *reinterpret_cast<f16x4*>(Xm + l * XP + c4) = p;
        }
      }
    }
    CPHASE(14);
    // small per-row / per-column parameters -> LDS once (they sit on the critical path of every epilogue otherwise)
    if (tid < 64) { prm[0 * 64 + tid] = pr[0]; prm[1 * 64 + tid] = pr[1]; prm[2 * 64 + tid] = pr[2]; prm[3 * 64 + tid] = pr[3]; }
    // L-axis weights as A-images [m][k], zero padded to 64x64
    stage_weight_commit(g0 ? Aw + 0 * IMG : Aw + 1 * IMG, w0, 0, 0, g0 ? hl : ol, g0 ? il : hl, t);
    if (g0) stage_weight_commit(Aw + 2 * IMG, w1, 0, 0, ol, il, t);
    if (G == 1) stage_weight_commit(Aw + 1 * IMG, w2, 0, 0, ol, hl, t);
  }
  __syncthreads();
  CPHASE(1);
  if (a.dbg_phase == 1) return;

  // ------------------------------------------------------------------ phase L: 64-column slabs, one per wave group and round
  for (int nb = 0; nb < C; nb += 64 * G) {
    const int n0 = nb + 64 * grp;
    const bool on = n0 < C;                                   // (barriers are workgroup-wide: an idle group keeps step)
    if (on) {   // Bx[n][k] = X[k][n0+n]   (transpose within LDS; 16 consecutive k per thread -> two 16-byte stores)
      const int n = t & 63, kg = (t >> 6) * 16;
      f16x8 lo, hi;
#pragma unroll
      for (int j = 0; j < 8; ++j) { lo[j] = Xm[(kg + j) * XP + n0 + n]; hi[j] = Xm[(kg + 8 + j) * XP + n0 + n]; }
      *reinterpret_cast<f16x8*>(Bx + n * ILD + kg) = lo;
      *reinterpret_cast<f16x8*>(Bx + n * ILD + kg + 8) = hi;
    }
    __syncthreads();
    if (nb == 0) CPHASE(8);
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    if (on) {
      mma64(acc, Aw + 0 * IMG, Bx, wm, wn, lane);
      // U = acc + b1 ; H = act(U) -> Bh[n][m] (B-image of the second product), saved tensors to HBM
      const int n = wn * 32 + (lane & 31);
      act_dispatch(a.act, [&](auto AT) __attribute__((always_inline)) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          f16x4 p;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int m = acc_row(4 * g + q, wm, lane);
            float h = 0.f;
            if (m < hl) {
              const float u = acc[4 * g + q] + prm[m];
              h = act_apply_c<decltype(AT)::value>(a.act, u);
              if (SAVE) {
                a.l_u[((long)b * hl + m) * C + n0 + n] = u;
                a.l_h[((long)b * hl + m) * C + n0 + n] = h;
              }
            }
            p[q] = to_f16(h);
          }
          *reinterpret_cast<f16x4*>(Bh + n * ILD + wm * 32 + 8 * g + 4 * (lane >> 5)) = p;
        }
      });
    }
    __syncthreads();
    if (nb == 0) CPHASE(9);
    if (on) {
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = 0.f;
      mma64(acc, Aw + 1 * IMG, Bh, wm, wn, lane);     // W2 . H
      mma64(acc, Aw + 2 * IMG, Bx, wm, wn, lane);     // + Wr . X
      const int n = wn * 32 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = acc_row(r, wm, lane);
        CtL[m * CTL + n] = acc[r] + prm[64 + m];
      }
    }
    __syncthreads();
    if (nb == 0) CPHASE(10);
    {   // LayerNorm over the L axis (rows m < ol) for each of the 64 columns; 4 threads per column, rows in registers
      const int n = t & 63, q = t >> 6;
      float yv[16];
      float s = 0.f;
      if (on) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {                         // 16 independent LDS reads (one round trip, not 16)
          const int m = q + 4 * j;
          yv[j] = m < ol ? CtL[m * CTL + n] : 0.f;
          s += yv[j];
        }
        red[q * 64 + n] = s;
      }
      __syncthreads();
      float mu = 0.f;
      if (on) {
        mu = (red[n] + red[64 + n] + red[128 + n] + red[192 + n]) / ol;
        float v = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) { const float c = (q + 4 * j < ol) ? yv[j] - mu : 0.f; v += c * c; }
        red[256 + q * 64 + n] = v;
      }
      __syncthreads();
      if (on) {
        const float rs = rsqrtf((red[256 + n] + red[320 + n] + red[384 + n] + red[448 + n]) / ol + LN_EPS);
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          const int m = q + 4 * j;
          if (m < ol) {
            const float z = (yv[j] - mu) * rs * prm[128 + m] + prm[192 + m];
            const bf zb = to_f16(z);
            Xm[m * XP + n0 + n] = zb;                             // in place: this slab's X columns are dead
            if (SAVE) {
              a.l_y[((long)b * ol + m) * C + n0 + n] = yv[j];
              // the backward pass gets the value the K phase CONSUMED (the rounded one): kmix_bwd recomputes the K-axis forward from it,
              // and LayerNorm over K = 3 amplifies a 2^-9 difference of the evaluation point into percents of the gradient
              a.l_z[((long)b * ol + m) * C + n0 + n] = (float)zb;
            }
          }
        }
        if (SAVE && q == 0) { a.l_mean[(long)b * C + n0 + n] = mu; a.l_rstd[(long)b * C + n0 + n] = rs; }
      }
    }
    __syncthreads();
    if (nb == 0) CPHASE(11);
  }

  CPHASE(2);
  if (a.dbg_phase == 2) return;
  // ------------------------------------------------------------------ phase K: K-axis mix in place on rows l < ol
  // first D-axis weight image: requested here, a whole phase ahead of its use (it was a bare round trip in front of the D phase)
  WImg wnext;
  if constexpr (!SAVE) wnext = load_weight128(a.d_w1, (G == 1 ? 0 : grp) * 64, 0, t);   // (the SAVE build has no register to spare across phase K)
  kmix_stage_weights(a.kw, ksw);
  {
    // compile-time K (1..4): the per-pair MLP is fully unrolled over it and this phase is VALU-bound (8 waves on 4 SIMDs)
    auto phase_k = [&](auto NKc) __attribute__((always_inline)) {
      constexpr int NK = decltype(NKc)::value;
      act_dispatch(a.kw.act, [&](auto AT) __attribute__((always_inline)) {
      KMixRegs<NK> kq;          // weights in registers for the whole loop (see KMixRegs)
      kq.load(ksw);
#pragma unroll 2
      for (int i = tid; i < ol * D; i += NT) {   // (two (l, d) pairs interleaved: the VALU chains of one pair hide nothing)
        const int l = i >> 7, d = i & 127;
        KMixVals<NK> v;
#pragma unroll
        for (int k = 0; k < NK; ++k) { v.x[k] = (float)Xm[l * XP + k * D + d]; v.sc[k] = 1.f; }
        kmix_forward_regs<NK, decltype(AT)::value>(a.kw, kq, v);    // (cube_fused_supported: never ln_first, ik == hk == ok)
        float out[NK];
        ln_small<NK>(v.y, NK, kq.g, kq.be, out, v.xh, v.mu, v.rs);
#pragma unroll
        for (int o = 0; o < NK; ++o) {
          const bf ob = to_f16(out[o]);
          Xm[l * XP + o * D + d] = ob;
          if (SAVE) a.k_z[(((long)b * ol + l) * NK + o) * D + d] = (float)ob;   // (what the D phase consumed)
        }
      }
      });
    };
    if (K == 3) phase_k(std::integral_constant<int, 3>{});
    else if (K == 1) phase_k(std::integral_constant<int, 1>{});
    else if (K == 2) phase_k(std::integral_constant<int, 2>{});
    else phase_k(std::integral_constant<int, 4>{});
  }
  __syncthreads();

  CPHASE(3);
  if (a.dbg_phase == 3) return;
  // ------------------------------------------------------------------ phase D
  const int R = ol * K;
  float* CtD = reinterpret_cast<float*>(smem + cv.ctd);
  float* dgam = reinterpret_cast<float*>(smem + cv.dprm);   // D-axis LayerNorm gain / bias (staged after Xm died)
  float* dbet = dgam + D;
  bf* Az = reinterpret_cast<bf*>(smem + cv.az);     // [NMT][2][64][ILD]   A-images of Z_k (k halves of d)
  bf* Ah = reinterpret_cast<bf*>(smem + cv.ah);     // [NMT][2][64][ILD]   A-images of H
  for (int c = tid; c < NMT * 64 * 16; c += NT) {    // 16-byte chunks: row r, chunk ch (8 d-values)
    const int r = c >> 4, ch = c & 15;
    f16x8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = to_f16(0.f);
    if (r < R) {
      const int l = r / K, kk = r - l * K;
      v = *reinterpret_cast<const f16x8*>(Xm + l * XP + kk * D + ch * 8);
    }
    *reinterpret_cast<f16x8*>(Az + ((r >> 6) * 2 + (ch >> 3)) * IMG + (r & 63) * ILD + (ch & 7) * 8) = v;
  }
  __syncthreads();   // Xm is dead from here on (the weight-image buffers alias it)
  CPHASE(4);
  if (tid < D) { dgam[tid] = a.d_g[tid]; dbet[tid] = a.d_be[tid]; }   // visible after the barriers of the GEMM loops
  // a wave owns NTW of the two 64-column halves of the outputs (both for G = 1, its group's one for G = 2)
  constexpr int NTW = 2 / G, NI1 = 2 * NTW;        // weight images per product and group: (nt, kh)
  const int nt0 = G == 1 ? 0 : grp;
  // biases of the two D-axis products for this lane's columns: requested here, not in the epilogues behind the GEMM loops
  // (a dependent global round trip there, with one workgroup per CU and nothing to hide it)
  float b1v[NTW], b2v[NTW];
#pragma unroll
  for (int q = 0; q < NTW; ++q) {
    const int col = (nt0 + q) * 64 + wn * 32 + (lane & 31);
    b1v[q] = a.d_b1 ? a.d_b1[col] : 0.f;
    b2v[q] = a.d_b2 ? a.d_b2[col] : 0.f;
  }

  f32x16 acc[NMT][NTW];
#pragma unroll
  for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
    for (int q = 0; q < NTW; ++q)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[mt][q][i] = 0.f;
  // The 3 * NI1 weight images of this group, in consumption order: W1 (nt,kh), then W2 (nt,kh), then Wr (nt,kh).
  // Image i+1 is requested from L2 before the MFMAs of image i (register prefetch), two LDS image buffers alternate.
  auto wload = [&](int i) -> WImg {
    const float* W = i < NI1 ? a.d_w1 : (i < 2 * NI1 ? a.d_w2 : a.d_wr);
    const int j = i % NI1;
    return load_weight128(W, (nt0 + (j >> 1)) * 64, (j & 1) * 64, t);
  };
  bf* Bw2[2] = {reinterpret_cast<bf*>(smem + cv.bw) + (2 * grp) * IMG, reinterpret_cast<bf*>(smem + cv.bw) + (2 * grp + 1) * IMG};
  if constexpr (SAVE) wnext = wload(0);
  // H = act(Z W1^T + b1): NI1 weight images (nt, kh), each used by all row tiles
#pragma unroll
  for (int i = 0; i < NI1; ++i) {
    const int q = i >> 1, kh = i & 1;
    commit_weight128(Bw2[i & 1], wnext, t);
    wnext = wload(i + 1);
    __syncthreads();
#pragma unroll
    for (int mt = 0; mt < NMT; ++mt) mma64(acc[mt][q], Az + (mt * 2 + kh) * IMG, Bw2[i & 1], wm, wn, lane);
  }
  CPHASE(12);
  act_dispatch(a.act, [&](auto AT) __attribute__((always_inline)) {
#pragma unroll
  for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
    for (int q = 0; q < NTW; ++q) {
      const int nt = nt0 + q, n = wn * 32 + (lane & 31), col = nt * 64 + n;
      const float b1 = b1v[q];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = acc_row(r, wm, lane), row = mt * 64 + m;
        float h = 0.f;
        if (row < R) {
          const float u = acc[mt][q][r] + b1;
          h = act_apply_c<decltype(AT)::value>(a.act, u);
          if (SAVE) {
            a.d_u[((long)b * R + row) * D + col] = u;
            a.d_h[((long)b * R + row) * D + col] = h;
          }
        }
        Ah[(mt * 2 + nt) * IMG + m * ILD + n] = to_f16(h);     // A-image of the next product: k = this column
        acc[mt][q][r] = 0.f;
      }
    }
  });
  __syncthreads();
  CPHASE(5);
  if (a.dbg_phase == 4) return;
  // Y = H W2^T + Z Wr^T + b2: 2 * NI1 weight images
#pragma unroll
  for (int i = NI1; i < 3 * NI1; ++i) {
    const int which = i >= 2 * NI1, q = (i % NI1) >> 1, kh = i & 1;
    commit_weight128(Bw2[i & 1], wnext, t);
    if (i + 1 < 3 * NI1) wnext = wload(i + 1);
    __syncthreads();
    const bf* Asrc = which == 0 ? Ah : Az;
#pragma unroll
    for (int mt = 0; mt < NMT; ++mt) mma64(acc[mt][q], Asrc + (mt * 2 + kh) * IMG, Bw2[i & 1], wm, wn, lane);
  }
  __syncthreads();   // (the H images are dead: the LayerNorm tile below aliases them)
  CPHASE(6);
  if (a.dbg_phase == 5) return;
  // LayerNorm over D per row, through an fp32 LDS tile
#pragma unroll
  for (int mt = 0; mt < NMT; ++mt) {
#pragma unroll
    for (int q = 0; q < NTW; ++q) {
      const int n = wn * 32 + (lane & 31), col = (nt0 + q) * 64 + n;
      const float b2 = b2v[q];
#pragma unroll
      for (int r = 0; r < 16; ++r) CtD[acc_row(r, wm, lane) * CTD + col] = acc[mt][q][r] + b2;
    }
    __syncthreads();
    {   // 4 * G threads per row, 32 / G columns each (short shuffle reductions instead of two 6-step wave reductions per row)
      constexpr int TPR = 4 * G, CW = 32 / G;
      const int m = tid / TPR, part = tid % TPR, row = mt * 64 + m;
      float yv[CW];
      float s0 = 0.f;
#pragma unroll
      for (int j = 0; j < CW / 4; ++j) {
        const float4 q = *reinterpret_cast<const float4*>(&CtD[m * CTD + part * CW + 4 * j]);
        yv[4 * j] = q.x; yv[4 * j + 1] = q.y; yv[4 * j + 2] = q.z; yv[4 * j + 3] = q.w;
        s0 += (q.x + q.y) + (q.z + q.w);
      }
      s0 = group_sum<TPR>(s0);
      const float mu = s0 * (1.f / D);
      float v0 = 0.f;
#pragma unroll
      for (int j = 0; j < CW; ++j) { const float c = yv[j] - mu; v0 += c * c; }
      v0 = group_sum<TPR>(v0);
      const float rs = rsqrtf(v0 * (1.f / D) + LN_EPS);
      if (row < R) {
        const long o = ((long)b * R + row) * D + part * CW;
#pragma unroll
        for (int j = 0; j < CW / 4; ++j) {
          const int c = part * CW + 4 * j;
          float4 z;
          z.x = (yv[4 * j] - mu) * rs * dgam[c] + dbet[c];
          z.y = (yv[4 * j + 1] - mu) * rs * dgam[c + 1] + dbet[c + 1];
          z.z = (yv[4 * j + 2] - mu) * rs * dgam[c + 2] + dbet[c + 2];
          z.w = (yv[4 * j + 3] - mu) * rs * dgam[c + 3] + dbet[c + 3];
          *reinterpret_cast<float4*>(a.d_z + o + 4 * j) = z;
          if (SAVE) *reinterpret_cast<float4*>(a.d_y + o + 4 * j) = make_float4(yv[4 * j], yv[4 * j + 1], yv[4 * j + 2], yv[4 * j + 3]);
        }
        if (SAVE && part == 0) { a.d_mean[(long)b * R + row] = mu; a.d_rstd[(long)b * R + row] = rs; }
      }
    }
    __syncthreads();
  }
  CPHASE(7);
}

}  // namespace

#ifdef MIMRL_PHASE_PROBE
int cube_fwd_read_phases(long long* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_cube_phase), sizeof(long long) * 128) == hipSuccess ? 0 : 1; }
#endif

bool cube_fused_supported(int il, int hl, int ol, int ik, int hk, int ok, int id, int hd, int od, bool ln_first,
                          bool res_project, bool bias, const float* dropout_mlp) {
  (void)bias;
  if (ln_first || !res_project) return false;
  if (dropout_mlp[0] > 0.f || dropout_mlp[1] > 0.f || dropout_mlp[2] > 0.f) return false;
  if (id != D || hd != D || od != D) return false;
  if (il > 64 || hl > 64 || ol > 64 || il < 1 || hl < 1 || ol < 1) return false;
  if (ik != hk || ik != ok || ik < 1 || ik > 4) return false;
  if (ol * ok > 192) return false;
  return true;
}

int cube_block_fwd_fused(hipStream_t s, const CubeFusedArgs& a) {
  const int R = a.ol * a.K;
  const int nmt = (R + 63) / 64;
  if (nmt < 1 || nmt > 3) return set_error(MIMRL_ERR_ARG, "cube_fused: unsupported row count %d", R);
  constexpr int groups = 2;   // (an environment knob until round 5: fixed at its measured optimum): wave groups per workgroup
  const int G = groups == 1 ? 1 : 2;
  const Carve cv = carve(a.K, nmt, G);
  if (cv.total > 160 * 1024) return set_error(MIMRL_ERR_ARG, "cube_fused: LDS budget exceeded (%d B)", cv.total);
#define LAUNCH_FUSED(SAVE, NMT, GG)                                                                                \
  do {                                                                                                             \
    auto kern = cube_fwd_fused_kernel<SAVE, NMT, GG>;                                                              \
    HIPX(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, cv.total)); \
    hipLaunchKernelGGL(kern, dim3(a.B), dim3(256 * GG), cv.total, s, a);                                           \
  } while (0)
#define LAUNCH_NMT(SAVE, GG)                                                                                       \
  do {                                                                                                             \
    if (nmt == 1) LAUNCH_FUSED(SAVE, 1, GG); else if (nmt == 2) LAUNCH_FUSED(SAVE, 2, GG); else LAUNCH_FUSED(SAVE, 3, GG); \
  } while (0)
  if (a.save) { if (G == 1) LAUNCH_NMT(true, 1); else LAUNCH_NMT(true, 2); }
  else { if (G == 1) LAUNCH_NMT(false, 1); else LAUNCH_NMT(false, 2); }
#undef LAUNCH_NMT
#undef LAUNCH_FUSED
  LAUNCH_CHECK();
  return MIMRL_OK;
}

}  // namespace mimrl
