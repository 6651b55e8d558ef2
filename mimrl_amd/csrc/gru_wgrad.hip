// Layer-0 bi-GRU weight gradients in ONE pass over dg (round 6).
//
// Reference semantics: autograd of nn.GRU (Model.py:254-255) -- per (modality, direction) sequence s
//   dW_ih[s] [3H, d]  = sum over (b,t) of dgx[s][b,t,:]^T x[b,t,:]        dgx = dg columns [0, 3H)            = [dr' | dz' | dn']
//   dW_hh[s] [3H, H]  = sum over (b,t) of dgh[s][b,t,:]^T h_prev[s][b,t,:] dgh = dg columns [0, 2H) u [3H, 4H) = [dr' | dz' | dn' r]
// with dg [B*T, 4H] and h_prev [B*T, H] as the BPTT kernel stored them (bf16) and x the packed bf16 inputs [B*T, kp] (l0_pack).
//
// Until round 5 these were two split-K GEMM launches per layer, each reading three of dg's four column blocks (measured: 1.02 GB of HBM
// fetches for the pair at cfg3, FETCH_SIZE).  Here every dg element is read once (0.83 GB fetched, 0.70 algorithmic):
//   * a workgroup owns a k-range (rows of B*T) of ONE sequence and ONE of two ROLES -- P: [dr' | dz'] x [x | h_prev] (56 output tiles of
//     32 x 32 at kp <= 96: three column tiles of x, four of h_prev), Q: dn' x x and dn'r x h_prev (28 tiles).  Both roles stage the same
//     bytes per k-tile (256 dg columns + the x and h_prev rows: 29 KB), which is what bounds them; the two roles and the two directions of a
//     (modality, k-range) pair -- which share the x rows, the roles also the h_prev rows -- sit on the SAME XCD (workgroup ids 8 apart): the
//     re-reads are L2 hits or merge with the sibling's miss;
//   * 8 waves = 4 (m-tile of each 128-column dg block) x 2 (n-tiles wn, wn + 2, ..): 8 accumulator tiles per wave in role P, 4 in role Q;
//     operands row-contiguous in memory, i.e. k-major: staged through a [k][column] LDS image exactly as loaded (16-byte pieces, register
//     ring of FPF k-tiles in flight, explicit counted waits as in gemm.hip's fast path, each load issued behind an MFMA) and transposed for
//     free by ds_read_b64_tr_b16 on BOTH operands;
//   * the result leaves through acc_add (float atomics; the deterministic build's table) into the packed gradient scratch the Adam launch
//     folds into the bucket (engine: dwih_pack / dwhh_pack).
// What bounds it (round-6 probes): the rate at which ONE CU takes data in under load, ~20-26 GB/s whatever the wave count
// or ring depth -- so the lever is the bytes a CU stages, not the bytes HBM delivers.  Three roles of 28 tiles each (dg in 128-column blocks,
// [x | h_prev] staged three times: 1.28 GB through the CUs) took 223-252 us at cfg3's shape, these two roles (0.93 GB) 203 us; the GEMM pair
// (1.54 GB) 274 us back to back, 235 us as two overlapped launches in the step.  One role (0.74 GB) needs 84 accumulator tiles per workgroup:
// more than two waves per SIMD can hold.
#include "gru_wgrad.h"

#include <type_traits>

namespace mimrl {

namespace {

constexpr int H = 128;
constexpr int KT = 32;                 // rows of B*T per k-tile
constexpr int NXT = 3, NHT = 4;        // 32-column tiles of x (kp <= 96) and of h_prev
constexpr int NBT = NXT + NHT;
constexpr int PB = 32 * NBT;           // B image pitch (elements): 448 B = 192 mod 256 -> the four k-rows of a transposed read hit distinct bank quarters
constexpr int PA = 2 * H + 32;         // A image pitch (256 dg columns): 576 B = 64 mod 256
constexpr int NT = 512;                // threads per workgroup: 8 waves, two per SIMD
constexpr int FPF = 3;                 // k-tiles in flight per workgroup (register ring)
constexpr int NTW = 4;                 // n-tiles per wave, role P: tiles wn, wn + 2, wn + 4, wn + 6 (the last one of wn = 1 does not exist)

typedef __attribute__((address_space(3))) bf16x4 lds4;

__device__ __forceinline__ void gld16(f32x4& d, const void* p) { asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(d) : "v"(p) : "memory"); }
template <int N, int NEWER>
__device__ __forceinline__ void ring_wait(f32x4* v) {
  asm volatile("s_waitcnt vmcnt(%0)" : : "n"(NEWER) : "memory");
#pragma unroll
  for (int h = 0; h < N; ++h) asm volatile("" : "+v"(v[h]));
}
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}

// MFMA operand fragment of the 32 columns starting at `col` of a [k][column] image, k-step s (16 k): lane holds column col + (lane & 31),
// k = 8 (lane >> 5) .. +7 (gemm.hip: fast_frag, row-contiguous form)
__device__ __forceinline__ bf16x8 frag(const __bf16* img, int pitch, int fo, int col, int s) {
  const __bf16* a = img + s * 16 * pitch + col + fo;
  const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4*)(a));
  const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4*)(a + 4 * pitch));
  bf16x8 r;
  r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3]; r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
  return r;
}

// C = role Q (dn' x x and dn'r x h_prev: 28 output tiles); otherwise role P (dr' and dz' against all seven column tiles: 56)
template <bool C>
__device__ __forceinline__ void body(const GruWgradArgs& a, const GruWgradSeq& q, int kt0, int kt1, __bf16* sA, __bf16* sB) {
  constexpr int NA = 2;                // 16-byte pieces per thread and k-tile: dg (256 columns x 32 rows)
  constexpr int NM = C ? 1 : 2;        // accumulator rows of tiles per wave: P: m-tile wm of dr' and of dz'; Q: one row, its left half dn', its right half dn'r
  constexpr int NX = 1, NH = 1;        // x (kp / 8 <= 12 pieces per row), h_prev (16 per row)
  constexpr int NL = NA + NX + NH;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, wm = w >> 1, wn = w & 1;   // 8 waves: 4 (m-tile) x 2 (n-tiles wn, wn + 2, ..)
  const int kp = a.kp, nxp = kp >> 3;
  const long rows = a.rows;

  // ---- per-thread piece coordinates (loop-invariant): global element offset from the tile's first row, LDS element offset
  int ga[NA], la[NA], ra[NA], gx[NX], lx[NX], rx[NX], gh[NH], lh[NH], rh[NH];
  const int acol0 = C ? 2 * H : 0;
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int p = tid + NT * i, row = p >> 5, ch = p & 31;
    ra[i] = row; ga[i] = row * 4 * H + acol0 + 8 * ch; la[i] = row * PA + 8 * ch;
  }
#pragma unroll
  for (int i = 0; i < NX; ++i) {
    int p = tid + NT * i;
    p = p < KT * nxp ? p : KT * nxp - 1;             // (past the tile: the last piece again, stored twice)
    const int row = p / nxp, ch = p - row * nxp;
    rx[i] = row; gx[i] = row * kp + 8 * ch; lx[i] = row * PB + 8 * ch;
  }
#pragma unroll
  for (int i = 0; i < NH; ++i) {
    const int p = tid + NT * i, row = p >> 4, ch = p & 15;
    rh[i] = row; gh[i] = row * H + 8 * ch; lh[i] = row * PB + 32 * NXT + 8 * ch;
  }
  const int j = lane & 15;
  const int foA = (8 * (lane >> 5) + (j >> 2)) * PA + 16 * ((lane >> 4) & 1) + 4 * (j & 3);
  const int foB = (8 * (lane >> 5) + (j >> 2)) * PB + 16 * ((lane >> 4) & 1) + 4 * (j & 3);

  f32x16 acc[NM][NTW];
#pragma unroll
  for (int i = 0; i < NM; ++i)
#pragma unroll
    for (int n = 0; n < NTW; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][n][r] = 0.f;
  // column tile (of the B image) of accumulator n: P: wn + 2 n; Q: n < 2 -> x tile wn + 2 n, else h_prev tile wn + 2 (n - 2)
  int bt[NTW];
#pragma unroll
  for (int n = 0; n < NTW; ++n) {
    int t = C ? (n < 2 ? wn + 2 * n : NXT + wn + 2 * (n - 2)) : wn + 2 * n;
    const int lim = C && n < 2 ? NXT - 1 : NBT - 1;
    bt[n] = t < lim ? t : lim;                       // (a tile that does not exist: a duplicate nobody stores)
  }

  f32x4 rg[FPF][NL];
  if (kt0 >= kt1) return;
  const int last = kt1 - 1;
  // load number `idx` (compile-time) of set jj: tile kt (clamped to the range's last tile; rows past the arrays' end re-read the last row)
  auto request_one = [&](auto J, auto IDX, int kt) __attribute__((always_inline)) {
    constexpr int jj = decltype(J)::value, i = decltype(IDX)::value;
    const long k0 = (long)(kt < last ? kt : last) * KT;
    if constexpr (i < NA) { const long r = k0 + ra[i] < rows ? k0 : rows - 1 - ra[i]; gld16(rg[jj][i], q.dg + r * 4 * H + ga[i]); }
    else if constexpr (i < NA + NX) { constexpr int e = i - NA; const long r = k0 + rx[e] < rows ? k0 : rows - 1 - rx[e]; gld16(rg[jj][i], q.x + r * kp + gx[e]); }
    else { constexpr int e = i - NA - NX; const long r = k0 + rh[e] < rows ? k0 : rows - 1 - rh[e]; gld16(rg[jj][i], q.hp + r * H + gh[e]); }
  };
  auto request = [&](auto J, int kt) __attribute__((always_inline)) {
    static_for<0, NL>([&](auto IDX) __attribute__((always_inline)) { request_one(J, IDX, kt); });
  };
  // set jj holds tile kt (clamped like the request): FPF - 1 newer sets may still be in flight behind it.  Tiles past the workgroup's range
  // (the k-loop runs whole groups of FPF tiles: no early exit, one straight-line loop body -- tools/isa_inflight.py can follow it) and
  // rows past the end of the sequence arrays contribute zero dg rows.
  auto publish = [&](auto J, int buf, int kt) __attribute__((always_inline)) {
    constexpr int jj = decltype(J)::value;
    const bool live = kt <= last;
    const long k0 = (long)(live ? kt : last) * KT;
    ring_wait<NL, (FPF - 1) * NL>(rg[jj]);
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    __bf16* A = sA + buf * (KT * PA);
    __bf16* Bm = sB + buf * (KT * PB);
#pragma unroll
    for (int i = 0; i < NA; ++i) *reinterpret_cast<f32x4*>(A + la[i]) = live && k0 + ra[i] < rows ? rg[jj][i] : z;
#pragma unroll
    for (int i = 0; i < NX; ++i) *reinterpret_cast<f32x4*>(Bm + lx[i]) = rg[jj][NA + i];
#pragma unroll
    for (int i = 0; i < NH; ++i) *reinterpret_cast<f32x4*>(Bm + lh[i]) = rg[jj][NA + NX + i];
  };
  static_for<0, FPF>([&](auto J) __attribute__((always_inline)) { request(J, kt0 + decltype(J)::value); });
  publish(std::integral_constant<int, 0>{}, 0, kt0);
  __syncthreads();
  int cur = 0;
  const int kt1r = kt0 + (kt1 - kt0 + FPF - 1) / FPF * FPF;
  for (int kb = kt0; kb < kt1r; kb += FPF) {
    static_for<0, FPF>([&](auto J) __attribute__((always_inline)) {
      constexpr int jj = decltype(J)::value;
      const int kt = kb + jj;
      // the NL loads of tile kt + FPF (set jj was published one iteration ago: free) are issued ONE behind each of the first NL products:
      // with every CU streaming, a load instruction waits ~90 cycles for a slot in the CU's memory queue (phase stamps: 737 cycles for the
      // 6-8 loads of a set issued back to back, during which the wave issued nothing else); behind an MFMA that wait runs under the product
      const __bf16* A = sA + cur * (KT * PA);
      const __bf16* Bm = sB + cur * (KT * PB);
      static_for<0, KT / 16>([&](auto S) __attribute__((always_inline)) {
        constexpr int s = decltype(S)::value;
        bf16x8 af[2], bfr[NTW];
        af[0] = frag(A, PA, foA, 32 * wm, s);          // P: dr' tile wm / Q: dn' tile wm
        af[1] = frag(A, PA, foA, H + 32 * wm, s);      // P: dz' tile wm / Q: dn'r tile wm
#pragma unroll
        for (int n = 0; n < NTW; ++n) bfr[n] = frag(Bm, PB, foB, 32 * bt[n], s);
        static_for<0, NM * NTW>([&](auto E) __attribute__((always_inline)) {
          constexpr int e = decltype(E)::value, i = e / NTW, n = e % NTW, c = s * NM * NTW + e;
          acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[C ? (n >= 2 ? 1 : 0) : i], bfr[n], acc[i][n], 0, 0, 0);
          if constexpr (c < NL) {
            __builtin_amdgcn_sched_barrier(0);
            request_one(J, std::integral_constant<int, c>{}, kt + FPF);
            __builtin_amdgcn_sched_barrier(0);
          }
        });
      });
      publish(std::integral_constant<int, (jj + 1) % FPF>{}, cur ^ 1, kt + 1);   // tile kt + 1 (past the range: zero dg rows)
      __syncthreads();
      cur ^= 1;
    });
  }
  // DRAIN the ring's last (duplicate) requests: loads the compiler knows nothing about, into registers it considers dead and would reuse
  // while they are still in flight (tools/isa_inflight.py caught exactly that here).  The ties keep the registers reserved up to the wait.
  static_for<0, FPF>([&](auto J) __attribute__((always_inline)) { ring_wait<NL, 0>(rg[decltype(J)::value]); });

  // ---- epilogue: gate row = role block + 32 (2 wm + i) + the accumulator's row; column tile < NXT -> dW_ih, else dW_hh
  const int n_lane = lane & 31;
#pragma unroll
  for (int i = 0; i < NM; ++i)
#pragma unroll
    for (int n = 0; n < NTW; ++n) {
      const int t = C ? (n < 2 ? wn + 2 * n : NXT + wn + 2 * (n - 2)) : wn + 2 * n;
      if (t > (C && n < 2 ? NXT - 1 : NBT - 1)) continue;
      const int rbase = (C ? 2 * H : i * H) + 32 * wm + 4 * (lane >> 5);
      const bool to_ih = t < NXT;
      const int col = to_ih ? 32 * t + n_lane : 32 * (t - NXT) + n_lane;
      if (to_ih && col >= kp) continue;
      float* out = to_ih ? q.dw_ih + col : q.dw_hh + col;
      const int ld = to_ih ? kp : H;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc_add(out + (long)(rbase + (r & 3) + 8 * (r >> 2)) * ld, acc[i][n][r]);
    }
}

__global__ __launch_bounds__(NT) void gru_wgrad_kernel(GruWgradArgs a) {
  __shared__ __attribute__((aligned(16))) __bf16 sA[2 * KT * PA];
  __shared__ __attribute__((aligned(16))) __bf16 sB[2 * KT * PB];
  // workgroup id -> (modality, k-range, direction, role): ids 8 apart land on the same XCD, so the four workgroups of a (modality, k-range)
  // pair -- which read the same x rows, two of them the same h_prev rows -- are ids u, u + 8, u + 16, u + 24 of a group of 32
  const unsigned g = blockIdx.x / 32u, jn = blockIdx.x - g * 32u, slot = jn >> 3;
  const unsigned pu = g * 8u + (jn & 7u);
  if (pu >= 2u * (unsigned)a.nsplit) return;
  const int m = (int)(pu / (unsigned)a.nsplit), split = (int)(pu - (unsigned)m * (unsigned)a.nsplit);
  const int dir = (int)(slot >> 1), role = (int)(slot & 1u);
  // x columns kp .. 95 of the B image are never written: zero them once (their products land in columns nobody stores, but they must be finite)
  for (int i = threadIdx.x; i < 2 * KT * PB / 2; i += NT) reinterpret_cast<unsigned*>(sB)[i] = 0u;
  __syncthreads();
  const GruWgradSeq& q = a.seq[m * 2 + dir];
  const int kt0 = split * a.kt_per;
  const int ktiles = (int)((a.rows + KT - 1) / KT);
  const int kt1 = kt0 + a.kt_per < ktiles ? kt0 + a.kt_per : ktiles;
  if (role == 1) body<true>(a, q, kt0, kt1, sA, sB);
  else body<false>(a, q, kt0, kt1, sA, sB);
}

}  // namespace

bool gru_wgrad_ok(long rows, int kp) { return rows >= 1 && rows < (1L << 31) / (4 * H) && kp >= 8 && kp <= 32 * NXT && kp % 8 == 0; }

int gru_wgrad(hipStream_t s, const GruWgradArgs& in) {
  if (!gru_wgrad_ok(in.rows, in.kp)) return set_error(MIMRL_ERR_ARG, "gru_wgrad: rows in [1, 2^22), kp in 8..96 and a multiple of 8");
  GruWgradArgs a = in;
  const int ktiles = (int)((a.rows + KT - 1) / KT);
  // ONE workgroup is resident per CU (512 threads x 256 registers), and the four workgroups of a (modality, k-range) pair sit on one XCD: an
  // XCD of 32 CUs takes eight pairs, so 8 XCDs x 8 pairs / 2 modalities = 32 k-ranges, 256 workgroups, all resident from the start.  (The
  // three-role version first asked for 21 k-ranges of 12: XCDs 0 and 1 got 36 workgroups for 32 CUs, the last four ran behind the others
  // and the launch took twice as long.)
  int nsplit = (device_cus() / 8 / 4) * 8 / 2;
  if (nsplit > ktiles) nsplit = ktiles;
  if (nsplit < 1) nsplit = 1;
  a.kt_per = (ktiles + nsplit - 1) / nsplit;
  a.nsplit = (ktiles + a.kt_per - 1) / a.kt_per;
  const unsigned groups = (2u * (unsigned)a.nsplit + 7u) / 8u;
  hipLaunchKernelGGL(gru_wgrad_kernel, dim3(groups * 32u), dim3(NT), 0, s, a);
  LAUNCH_CHECK();
  return MIMRL_OK;
}

}  // namespace mimrl
