// Weight gradients of the concat critic's two hidden layers (stage 1), every operand row staged ONCE (round 6).
//
// Reference semantics: autograd of the three-layer critic of VMI.py:58-65 over the B x B pair rows -- per estimator e and hidden layer l
//   dW_l[e] [256, 256] += dZ_l[e]^T A_{l-1}[e],   dZ_l, A_{l-1}: [B*B, 256]
// with dZ_l as the fused backward kernel stored it (bf16) and A_{l-1} the bf16 activation copy of the forward kernel.
//
// Until now: one split-K GEMM launch per layer on 128 x 128 tiles -- 4 output tiles per estimator, so every dZ and A row went through two
// workgroups' LDS (0.67 GB staged per layer for 0.34 GB of operands at cfg3), 189 + 115 us in a row on stage 1's chain between the backward
// kernel and the critic update.  What bounds such a product is the rate at which a CU takes data in under load (~20-26 GB/s, gru_wgrad.hip),
// i.e. the bytes staged.  Here a workgroup holds the WHOLE 256 x 256 output of one (layer, estimator) for its k-range: 64 accumulator tiles
// over 8 waves (2 m-tiles x 4 n-tiles each = 128 registers), both operands through a [k][column] LDS image as loaded and
// ds_read_b64_tr_b16 (both are k-major in memory), register ring of FPF k-tiles with counted waits, each load issued behind an MFMA.  Both
// layers in ONE launch: workgroup = (layer, estimator, k-range), ~one per CU -- plus, optionally, the score head's weight gradient
// dw3[e] = ds[e]^T a2[e] (a streaming weighted column sum over the fp16 a2) on the remaining ninth of the CUs, sized to stage the same bytes
// per workgroup.  The result leaves through acc_add (float atomics; the deterministic build's table).  Algorithmic bytes per launch at cfg3 (5 estimators, B = 256): 4 x 168 MB = 0.67 GB.
#include "concat_dw.h"

#include <type_traits>

namespace mimrl {

namespace {

constexpr int CH = 256;                // hidden width
constexpr int KT = 32;                 // pair rows per k-tile
constexpr int PI = CH + 32;            // image pitch (elements): 576 B = 64 mod 256 -> the four k-rows of a transposed read hit distinct bank quarters
constexpr int NT = 512;                // 8 waves: 4 (pairs of m-tiles) x 2 (halves of the n-tiles)
constexpr int FPF = 3;                 // k-tiles in flight per workgroup (register ring)
constexpr int NL = 4;                  // 16-byte pieces per thread and k-tile: 2 of dZ, 2 of A (32 rows x 256 columns each)

typedef __attribute__((address_space(3))) bf16x4 lds4;

__device__ __forceinline__ void gld16(f32x4& d, const void* p) { asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(d) : "v"(p) : "memory"); }
__device__ __forceinline__ void gld4(uint32_t& d, const void* p) { asm volatile("global_load_dword %0, %1, off" : "=v"(d) : "v"(p) : "memory"); }
template <int N, int NEWER>
__device__ __forceinline__ void ring_wait(f32x4* v) {
  asm volatile("s_waitcnt vmcnt(%0)" : : "n"(NEWER) : "memory");
#pragma unroll
  for (int h = 0; h < N; ++h) asm volatile("" : "+v"(v[h]));
}
template <int NEWER>
__device__ __forceinline__ void gen_wait(uint32_t* g, f32x4* v) {   // the GEN set: four dwords + two pieces
  asm volatile("s_waitcnt vmcnt(%0)" : : "n"(NEWER) : "memory");
#pragma unroll
  for (int h = 0; h < 4; ++h) asm volatile("" : "+v"(g[h]));
  asm volatile("" : "+v"(v[0]));
  asm volatile("" : "+v"(v[1]));
}
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}
// MFMA operand fragment of the 32 columns starting at `col` of a [k][column] image, k-step s (gru_wgrad.hip: frag)
__device__ __forceinline__ bf16x8 frag(const __bf16* img, int fo, int col, int s) {
  const __bf16* a = img + s * 16 * PI + col + fo;
  const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4*)(a));
  const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4*)(a + 4 * PI));
  bf16x8 r;
  r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3]; r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
  return r;
}

// GEN (layer 2 only): the dZ2 tile is not loaded but regenerated -- dZ2[r][c] = bf16(ds[r] * w3[c]) where bit c of the layer-2 sign words is
// set, exactly what concat_bwd_ws_kernel computes for its own product (concat_ws_bwd.hip: gen_finish) and, until this round, also wrote out:
// 168 MB less to write there and to read here per critic pass at cfg3.  Per piece (row, 8 columns): one sign word + one ds value.
// MODE 2 (layer 1 only): the A_0 tile is regenerated instead -- a0[i B + j][c] = bf16(relu(P[i][c] + Q[j][c])), the separable first layer exactly
// as the forward kernel generates it (concat_ws.hip: gen_finish), which then does not save it (ConcatFwdArgs::save == 4).  As the rows are
// ordered a k-tile is one i and 32 consecutive j: the workgroup walks its k-tiles j-BLOCK-major (tile t -> j block t / B, i = t % B), keeps
// the block's 32 Q rows in LDS (reloaded when the block changes: at most once or twice per workgroup) and stages ONE P row per k-tile.
template <int MODE>
__device__ __forceinline__ void products(const ConcatDwArgs& a, int layer, int e, int kt0, int kt1, __bf16* sA, __bf16* sB, float* sQ) {
  constexpr bool GEN = MODE == 1, GA0 = MODE == 2;
  constexpr int NLD = GEN ? 6 : NL;    // load instructions per thread and k-tile
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, wm = w >> 1, wn = w & 1;
  const long rows = a.rows;
  const __bf16* __restrict__ dz = a.dz[layer] + (long)e * rows * CH;
  const __bf16* __restrict__ act = GA0 ? nullptr : a.act[layer] + (long)e * rows * CH;
  [[maybe_unused]] const float* __restrict__ Pe = GA0 ? a.P + (long)e * a.B * CH : nullptr;
  [[maybe_unused]] const float* __restrict__ Qe = GA0 ? a.Q + (long)e * a.B * CH : nullptr;
  [[maybe_unused]] const int Bn = a.B;
  [[maybe_unused]] int cur_jb = -1;
  // first pair row of k-tile t
  auto row0 = [&](int t) __attribute__((always_inline)) -> long {
    if constexpr (GA0) { const int jb = t / Bn, i = t - jb * Bn; return (long)i * Bn + (long)jb * KT; }
    else return (long)t * KT;
  };
  float* __restrict__ out = a.dw[layer] + (long)e * a.dw_stride;
  const int last = kt1 - 1;

  // per-thread piece coordinates (loop-invariant): piece p = tid + 512 i of a 32 x 256 tile -> row p >> 5, 16-byte chunk p & 31
  int gp[2], lp[2], rp[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int p = tid + NT * i, row = p >> 5, ch = p & 31;
    rp[i] = row; gp[i] = row * CH + 8 * ch; lp[i] = row * PI + 8 * ch;
  }
  const int chn = tid & 31;            // this thread's 8-column chunk (the same for both of its pieces)
  [[maybe_unused]] const uint32_t* __restrict__ m2e = nullptr;
  [[maybe_unused]] const float* __restrict__ dse = nullptr;
  [[maybe_unused]] float w3r[8];
  if constexpr (GEN) {
    m2e = a.m2 + (long)e * rows * 8 + (chn >> 2);
    dse = a.ds + (long)e * rows;
    const float* w3p = a.w3 + (long)e * a.dw_stride + 8 * chn;
#pragma unroll
    for (int c = 0; c < 8; ++c) w3r[c] = w3p[c];
  }
  const int j = lane & 15;
  const int fo = (8 * (lane >> 5) + (j >> 2)) * PI + 16 * ((lane >> 4) & 1) + 4 * (j & 3);

  f32x16 acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][n][r] = 0.f;

  f32x4 rg[FPF][NL];                    // [0, 1]: dZ pieces (not GEN), [2, 3]: A pieces
  [[maybe_unused]] uint32_t gw[FPF][4]; // GEN: {sign word, ds} of the two rows
  // load number `idx` of set jj: tile kt (clamped to the range's last tile; rows past the end re-read the last row)
  auto request_one = [&](auto J, auto IDX, int kt) __attribute__((always_inline)) {
    constexpr int jj = decltype(J)::value, i = decltype(IDX)::value;
    const int tc = kt < last ? kt : last;
    const long k0 = row0(tc);
    if constexpr (GA0) {
      constexpr int h = i & 1;
      if constexpr (i < 2) {
        const long r = k0 + rp[h] < rows ? k0 : rows - 1 - rp[h];
        gld16(rg[jj][i], dz + r * CH + gp[h]);
      } else gld16(rg[jj][i], Pe + (long)(tc % Bn) * CH + 8 * chn + 4 * h);   // the k-tile's P row: this thread's 8 columns, two quads
    } else if constexpr (GEN) {
      if constexpr (i < 4) {
        constexpr int h = i >> 1;
        const long r = (k0 + rp[h] < rows ? k0 : rows - 1 - rp[h]) + rp[h];
        if constexpr ((i & 1) == 0) gld4(gw[jj][i], m2e + r * 8); else gld4(gw[jj][i], dse + r);
      } else {
        constexpr int h = i - 4;
        const long r = k0 + rp[h] < rows ? k0 : rows - 1 - rp[h];
        gld16(rg[jj][2 + h], act + r * CH + gp[h]);
      }
    } else {
      constexpr int h = i & 1;
      const long r = k0 + rp[h] < rows ? k0 : rows - 1 - rp[h];
      gld16(rg[jj][i], (i < 2 ? dz : act) + r * CH + gp[h]);
    }
  };
  auto wait_set = [&](auto J, auto NEWER) __attribute__((always_inline)) {
    constexpr int jj = decltype(J)::value;
    if constexpr (GEN) gen_wait<decltype(NEWER)::value>(gw[jj], rg[jj] + 2);
    else ring_wait<NL, decltype(NEWER)::value>(rg[jj]);
  };
  // set jj holds tile kt: FPF - 1 newer sets may still be in flight behind it.  Tiles past the workgroup's range (the k-loop runs whole
  // groups of FPF tiles: one straight-line loop body, tools/isa_inflight.py can follow it) and rows past the end contribute zero dZ rows.
  auto publish = [&](auto J, int buf, int kt) __attribute__((always_inline)) {
    constexpr int jj = decltype(J)::value;
    const bool live = kt <= last;
    const int tc = live ? kt : last;
    const long k0 = row0(tc);
    if constexpr (GA0) {   // the j block's Q rows -> LDS when the block changes (block-uniform; the loads are the compiler's own, waited for here)
      const int jb = tc / Bn;
      if (jb != cur_jb) {
        cur_jb = jb;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int idx = tid + NT * k, row = idx >> 6, c4 = idx & 63;
          *reinterpret_cast<float4*>(sQ + row * CH + 4 * c4) = *reinterpret_cast<const float4*>(Qe + ((long)jb * KT + row) * CH + 4 * c4);
        }
        __syncthreads();
      }
    }
    wait_set(J, std::integral_constant<int, (FPF - 1) * NLD>{});
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    __bf16* A = sA + buf * (KT * PI);
    __bf16* Bm = sB + buf * (KT * PI);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      f32x4 piece;
      if constexpr (GEN) {
        const uint32_t bits = gw[jj][2 * i] >> (8 * (chn & 3));
        const float d = __uint_as_float(gw[jj][2 * i + 1]);
        bf16x8 v;
#pragma unroll
        for (int c = 0; c < 8; ++c) v[c] = to_bf16((bits >> c) & 1u ? d * w3r[c] : 0.f);
        piece = __builtin_bit_cast(f32x4, v);
      } else piece = rg[jj][i];
      *reinterpret_cast<f32x4*>(A + lp[i]) = live && k0 + rp[i] < rows ? piece : z;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      f32x4 piece;
      if constexpr (GA0) {
        const float4 q0 = *reinterpret_cast<const float4*>(sQ + rp[i] * CH + 8 * chn), q1 = *reinterpret_cast<const float4*>(sQ + rp[i] * CH + 8 * chn + 4);
        const f32x4 p0 = rg[jj][2], p1 = rg[jj][3];
        bf16x8 v;
        v[0] = to_bf16(fmaxf(p0[0] + q0.x, 0.f)); v[1] = to_bf16(fmaxf(p0[1] + q0.y, 0.f)); v[2] = to_bf16(fmaxf(p0[2] + q0.z, 0.f)); v[3] = to_bf16(fmaxf(p0[3] + q0.w, 0.f));
        v[4] = to_bf16(fmaxf(p1[0] + q1.x, 0.f)); v[5] = to_bf16(fmaxf(p1[1] + q1.y, 0.f)); v[6] = to_bf16(fmaxf(p1[2] + q1.z, 0.f)); v[7] = to_bf16(fmaxf(p1[3] + q1.w, 0.f));
        piece = __builtin_bit_cast(f32x4, v);
      } else piece = rg[jj][2 + i];
      *reinterpret_cast<f32x4*>(Bm + lp[i]) = piece;
    }
  };
  static_for<0, FPF>([&](auto J) __attribute__((always_inline)) {
    static_for<0, NLD>([&](auto IDX) __attribute__((always_inline)) { request_one(J, IDX, kt0 + decltype(J)::value); });
  });
  publish(std::integral_constant<int, 0>{}, 0, kt0);
  __syncthreads();
  int cur = 0;
  const int kt1r = kt0 + (kt1 - kt0 + FPF - 1) / FPF * FPF;
  for (int kb = kt0; kb < kt1r; kb += FPF) {
    static_for<0, FPF>([&](auto J) __attribute__((always_inline)) {
      constexpr int jj = decltype(J)::value;
      const int kt = kb + jj;
      const __bf16* A = sA + cur * (KT * PI);
      const __bf16* Bm = sB + cur * (KT * PI);
      static_for<0, KT / 16>([&](auto S) __attribute__((always_inline)) {
        constexpr int s = decltype(S)::value;
        bf16x8 af[2], bfr[4];
#pragma unroll
        for (int i = 0; i < 2; ++i) af[i] = frag(A, fo, 32 * (2 * wm + i), s);
#pragma unroll
        for (int n = 0; n < 4; ++n) bfr[n] = frag(Bm, fo, 32 * (4 * wn + n), s);
        static_for<0, 8>([&](auto E) __attribute__((always_inline)) {
          constexpr int q = decltype(E)::value, i = q >> 2, n = q & 3, c = s * 8 + q;
          acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[n], acc[i][n], 0, 0, 0);
          if constexpr (c < NLD) {   // the loads of tile kt + FPF, one behind each of the first NLD products (gru_wgrad.hip)
            __builtin_amdgcn_sched_barrier(0);
            request_one(J, std::integral_constant<int, c>{}, kt + FPF);
            __builtin_amdgcn_sched_barrier(0);
          }
        });
      });
      publish(std::integral_constant<int, (jj + 1) % FPF>{}, cur ^ 1, kt + 1);
      __syncthreads();
      cur ^= 1;
    });
  }
  // drain the ring's last (duplicate) requests; the ties keep their registers reserved up to the wait (gru_wgrad.hip)
  static_for<0, FPF>([&](auto J) __attribute__((always_inline)) { wait_set(J, std::integral_constant<int, 0>{}); });

  // epilogue: dW[m][n], m = dZ column (the layer's output unit), n = A column (its input unit)
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      float* o = out + (long)(32 * (2 * wm + i) + 4 * (lane >> 5)) * CH + 32 * (4 * wn + n) + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) acc_add(o + (long)((r & 3) + 8 * (r >> 2)) * CH, acc[i][n][r]);
    }
}

__global__ __launch_bounds__(NT) void concat_dw_kernel(ConcatDwArgs a) {
  __shared__ __attribute__((aligned(16))) __bf16 sA[2 * KT * PI];
  __shared__ __attribute__((aligned(16))) __bf16 sB[2 * KT * PI];
  __shared__ __attribute__((aligned(16))) float sQ[KT * CH];   // the j block's Q rows (regenerated a0 only)
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, wm = w >> 1, wn = w & 1;
  // workgroup -> (layer, estimator, k-range) | (score head: estimator, row range)
  const int n0 = a.E * a.nsplit_l[0], n1 = a.nlayer > 1 ? a.E * a.nsplit_l[1] : 0;
  if ((int)blockIdx.x >= n0 + n1) {   // the score head's rows: 8 columns per lane, 32 lanes per row, 16 row phases
    const int b3 = (int)blockIdx.x - n0 - n1, e = b3 / a.n3, part_i = b3 - e * a.n3;
    const long r0 = (long)part_i * a.rows3, r1 = min(a.rows, r0 + a.rows3);
    const float* __restrict__ dse = a.ds + (long)e * a.rows;
    const _Float16* __restrict__ ae = a.a2 + (long)e * a.rows * CH;
    const int c8 = (tid & 31) * 8, ph = tid >> 5;
    float sum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (long r = r0 + ph; r < r1; r += 128) {
      f16x8 v[8]; float d[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const long rr = min(r + 16 * q, r1 - 1);
        v[q] = *reinterpret_cast<const f16x8*>(ae + rr * CH + c8);
        d[q] = r + 16 * q < r1 ? dse[rr] : 0.f;
      }
#pragma unroll
      for (int q = 0; q < 8; ++q)
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) sum[jj] += d[q] * (float)v[q][jj];
    }
    float* part = reinterpret_cast<float*>(sA);          // [16][256] floats = 16 KB of the 36 KB image
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) part[ph * CH + c8 + jj] = sum[jj];
    __syncthreads();
    if (tid < CH) {
      float t = 0.f;
#pragma unroll
      for (int p = 0; p < 16; ++p) t += part[p * CH + tid];
      acc_add(a.dw3 + (long)e * a.dw_stride + tid, t);
    }
    return;
  }
  const int layer = (int)blockIdx.x >= n0 ? 1 : 0, rem = (int)blockIdx.x - (layer ? n0 : 0), nsp = a.nsplit_l[layer];
  const int e = rem / nsp, split = rem - e * nsp;
  const int ktiles = (int)((a.rows + KT - 1) / KT);
  const int kt0 = split * a.kt_per_l[layer];
  const int kt1 = kt0 + a.kt_per_l[layer] < ktiles ? kt0 + a.kt_per_l[layer] : ktiles;
  if (kt0 >= kt1) return;
  if (layer == 0 && a.m2) products<1>(a, layer, e, kt0, kt1, sA, sB, sQ);
  else if (layer == 1 && a.P) products<2>(a, layer, e, kt0, kt1, sA, sB, sQ);
  else products<0>(a, layer, e, kt0, kt1, sA, sB, sQ);
}

}  // namespace

bool concat_dw_ok(int E, long rows, int hid) { return hid == CH && E >= 1 && rows >= 1 && rows < (1L << 31) / PI; }

int concat_dw(hipStream_t s, const ConcatDwArgs& in) {
  if (in.nlayer < 1 || in.nlayer > 2) return set_error(MIMRL_ERR_ARG, "concat_dw: one or two layers");
  ConcatDwArgs a = in;
  const int ktiles = (int)((a.rows + KT - 1) / KT);
  // ~one workgroup per CU (one is resident: 512 threads, 72 KB of LDS), each busy for about the same time: per pair row a product stages
  // 2 x 512 bytes (1 x 512 + 36 when its dZ is regenerated), the score head 512 -- the CUs are dealt in shares that follow
  const bool with3 = a.ds && a.a2 && a.dw3;
  if ((a.a2 || a.dw3) && !with3) return set_error(MIMRL_ERR_ARG, "concat_dw: ds, a2 and dw3 come as a set");
  const bool gen = a.m2 != nullptr;
  if (gen && !(a.w3 && a.ds)) return set_error(MIMRL_ERR_ARG, "concat_dw: regenerating dZ2 needs m2, w3 and ds");
  if (!gen && !a.dz[0]) return set_error(MIMRL_ERR_ARG, "concat_dw: null dZ");
  const int cus = device_cus();
  // (shares in tenths of a full product's: a k-tile whose dZ is regenerated still costs its MFMAs, LDS traffic and barrier -- 0.6 of a staged
  //  one, measured, not the 0.5 its bytes say; the score head's rows 0.5)
  const bool gen0 = a.P != nullptr;
  if (gen0 && !(a.Q && a.nlayer == 2 && a.B >= KT && a.B % KT == 0 && (long)a.B * a.B == a.rows))
    return set_error(MIMRL_ERR_ARG, "concat_dw: regenerating a0 needs P, Q, two layers and rows = B * B with B a multiple of 32");
  if (!gen0 && a.nlayer > 1 && !a.act[1]) return set_error(MIMRL_ERR_ARG, "concat_dw: null a0");
  const int share[2] = {gen ? 6 : 10, a.nlayer > 1 ? (gen0 ? 6 : 10) : 0};
  const int total = share[0] + share[1] + (with3 ? 5 : 0);
  int used = 0;
  for (int l = 0; l < a.nlayer; ++l) {
    int nsplit = cus * share[l] / total / a.E;
    if (nsplit > ktiles) nsplit = ktiles;
    if (nsplit < 1) nsplit = 1;
    a.kt_per_l[l] = (ktiles + nsplit - 1) / nsplit;
    a.nsplit_l[l] = (ktiles + a.kt_per_l[l] - 1) / a.kt_per_l[l];
    used += a.E * a.nsplit_l[l];
  }
  if (a.nlayer < 2) { a.nsplit_l[1] = 0; a.kt_per_l[1] = 0; }
  a.n3 = 0;
  if (with3) {
    int n3 = (cus - used) / a.E;
    if (n3 < 1) n3 = 1;
    a.rows3 = ((a.rows + n3 - 1) / n3 + 127) / 128 * 128;
    a.n3 = (int)((a.rows + a.rows3 - 1) / a.rows3);
  }
  hipLaunchKernelGGL(concat_dw_kernel, dim3((unsigned)(used + a.E * a.n3)), dim3(NT), 0, s, a);
  LAUNCH_CHECK();
  return MIMRL_OK;
}

}  // namespace mimrl
