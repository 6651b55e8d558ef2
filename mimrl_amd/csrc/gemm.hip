// Generic strided batched GEMM kernel (see gemm.h).  64x64 block tile, 4 waves (2x2), one 32x32 MFMA accumulator
// tile per wave, register-prefetched global->LDS staging, small register/LDS footprint (4 workgroups per CU): at the
// sizes of this workload the k-loop is a chain of dependent memory round trips and inter-workgroup overlap is what
// hides it (a deeper register ring / double-buffered LDS measured slower in situ: it costs occupancy).
//   * two instantiations per precision: GENERIC (BK=32; guarded scalar loads on edge tiles / misaligned operands,
//     16-byte loads elsewhere) and LEAN (BK=64, 16-byte loads only; chosen by the host when M,N are multiples of 64
//     and both operands qualify) -- half the dependent round trips without the register cost of the scalar path;
//   * optional second product accumulated into the same tile (C = A.B + A2.B2): residual projections ride along;
//   * epilogue: bias (per row/column), beta*C, activation / activation-gradient, pre-activation copy, fused column
//     sums (bias gradients), plain or atomic store;
//   * split-K for accumulate-into-zeroed-output GEMMs (weight gradients with K = B*T ... B*L*K rows, tiny M x N).
#include <cstdio>
#include "gemm.h"

#include <type_traits>

#include <cstdlib>

namespace mimrl {

namespace {

constexpr int BM = 64, BN = 64;

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

template <bool BF16, int BK>
struct Smem;
template <int BK>
struct Smem<false, BK> {
  float a[BK][BM + 4];   // k-major: lanes 0..31 read 32 consecutive m; +4 keeps 16-B row alignment
  float b[BK][BN + 4];
};
template <int BK>
struct Smem<true, BK> {
  __bf16 a[BM][BK + 8];  // m-major, 80-B / 144-B rows: 16-B fragment reads, conflict-free for ds_read_b128
  __bf16 b[BN][BK + 8];
};

struct KernelArgs {
  GemmDesc d;
  int ksplit;       // grid.z = batch * ksplit
  int kt_per;       // k-tiles per split (single-segment GEMMs only)
  int vec_a, vec_b, vec_a2, vec_b2;   // operand may use the 16-byte path (alignment / stride conditions hold)
  int xcd_remap;
  int dbg;          // timing experiments only (MIMRL_DBG_GEMM): 1 = no output stores, 2 = no k-loop
};

// one operand of one product segment (workgroup-uniform)
struct Operand {
  const float* P;
  long s_r, s_k;     // strides of the row (m or n) axis and of k
  int r0, R;         // first row of this tile / number of rows
  int gap_at, gap;   // rows >= gap_at are stored `gap` rows further on (0 = none; multiples of 4)
  bool kfast;        // thread mapping: k runs fastest (else rows)
  bool vec;          // 16-byte path for this tile
};

template <int BK>
struct Map {
  static constexpr int NV = BK / 16;   // float4 per thread per operand (64 x BK tile, 256 threads)
  static constexpr int NE = 4 * NV;
  // k contiguous: 4 consecutive k of one row;  rows contiguous: 4 consecutive rows at NV consecutive k
  static __device__ __forceinline__ void vec(bool kfast, int tid, int h, int& row, int& k) {
    if (kfast) { row = tid / (BK / 4) + (1024 / BK) * h; k = (tid % (BK / 4)) * 4; }
    else { k = (tid >> 4) * NV + h; row = (tid & 15) * 4; }
  }
  static __device__ __forceinline__ void sc(bool kfast, int tid, int i, int& row, int& k) {
    const int e = i * 256 + tid;
    if (kfast) { k = e % BK; row = e / BK; } else { row = e & 63; k = e >> 6; }
  }
};

template <int BK, bool LEAN>
__device__ __forceinline__ void load_tile(const Operand& o, int K, int k0, int tid, float* v) {
  using M = Map<BK>;
  if (LEAN || o.vec) {
    // Round 3b (tools/isa_lint.py): the per-piece guards (`if (gk + 3 < K) q = *src`) compiled to a branch + s_waitcnt vmcnt(0) per piece --
    // every k-tile of the fp32 parity mode paid its loads as serialized round trips.  Now ONE tile-uniform branch: a k-tile that lies
    // wholly inside K (all but the last one) issues its NV 16-byte loads unconditionally, back to back; only a ragged tile takes the
    // guarded path.
    const bool ragged = k0 + BK > K || (o.kfast && (K & 3) != 0);
    float4 w[M::NV];
    if (!ragged) {
#pragma unroll
      for (int h = 0; h < M::NV; ++h) {
        int row, k;
        M::vec(o.kfast, tid, h, row, k);
        // k-contiguous operands may have a ragged last ROW tile: rows beyond R are clamped (their products land in output
        // rows/columns the epilogue never stores)
        int gr = (LEAN && o.kfast) ? (o.r0 + row < o.R ? o.r0 + row : o.R - 1) : o.r0 + row;
        if (gr >= o.gap_at) gr += o.gap;
        w[h] = *reinterpret_cast<const float4*>(o.P + (long)gr * o.s_r + (long)(k0 + k) * o.s_k);
      }
    } else {
#pragma unroll
      for (int h = 0; h < M::NV; ++h) {
        int row, k;
        M::vec(o.kfast, tid, h, row, k);
        const int gk = k0 + k;
        int gr = (LEAN && o.kfast) ? (o.r0 + row < o.R ? o.r0 + row : o.R - 1) : o.r0 + row;
        if (gr >= o.gap_at) gr += o.gap;
        const float* src = o.P + (long)gr * o.s_r + (long)gk * o.s_k;
        float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
        if (o.kfast) {
          if (gk + 3 < K) q = *reinterpret_cast<const float4*>(src);
          else {
            if (gk < K) q.x = src[0];
            if (gk + 1 < K) q.y = src[1];
            if (gk + 2 < K) q.z = src[2];
          }
        } else if (gk < K) {
          q = *reinterpret_cast<const float4*>(src);
        }
        w[h] = q;
      }
    }
#pragma unroll
    for (int h = 0; h < M::NV; ++h) { v[4 * h + 0] = w[h].x; v[4 * h + 1] = w[h].y; v[4 * h + 2] = w[h].z; v[4 * h + 3] = w[h].w; }
  } else {
#pragma unroll
    for (int i = 0; i < M::NE; ++i) {
      int row, k;
      M::sc(o.kfast, tid, i, row, k);
      const int gr = o.r0 + row, gk = k0 + k;
      const int grc = gr < o.R ? gr : o.R - 1, gkc = gk < K ? gk : K - 1;      // clamped, unconditional (see above)
      const float x = o.P[(long)(grc >= o.gap_at ? grc + o.gap : grc) * o.s_r + (long)gkc * o.s_k];
      v[i] = (gr < o.R && gk < K) ? x : 0.f;
    }
  }
}

// t: bf16 image [row][k]  /  fp32 image [k][row]
template <bool BF16, int BK, bool LEAN, typename T>
__device__ __forceinline__ void store_tile(const Operand& o, int tid, const float* v, T& t) {
  using M = Map<BK>;
  if (LEAN || o.vec) {
    if (o.kfast) {
#pragma unroll
      for (int h = 0; h < M::NV; ++h) {
        int row, k;
        M::vec(true, tid, h, row, k);
        if constexpr (BF16) {
          bf16x4 p; p[0] = to_bf16(v[4 * h]); p[1] = to_bf16(v[4 * h + 1]); p[2] = to_bf16(v[4 * h + 2]); p[3] = to_bf16(v[4 * h + 3]);
          *reinterpret_cast<bf16x4*>(&t[row][k]) = p;
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) t[k + j][row] = v[4 * h + j];
        }
      }
    } else {
      int row, k;
      M::vec(false, tid, 0, row, k);
      if constexpr (BF16) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {   // the NV consecutive k of a row leave as one (2*NV)-byte store
          typedef __attribute__((ext_vector_type(M::NV))) __bf16 bfv;
          bfv p;
#pragma unroll
          for (int h = 0; h < M::NV; ++h) p[h] = to_bf16(v[4 * h + j]);
          *reinterpret_cast<bfv*>(&t[row + j][k]) = p;
        }
      } else {
#pragma unroll
        for (int h = 0; h < M::NV; ++h)
          *reinterpret_cast<float4*>(&t[k + h][row]) = make_float4(v[4 * h], v[4 * h + 1], v[4 * h + 2], v[4 * h + 3]);
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < M::NE; ++i) {
      int row, k;
      M::sc(o.kfast, tid, i, row, k);
      if constexpr (BF16) t[row][k] = to_bf16(v[i]);
      else t[k][row] = v[i];
    }
  }
}

// epilogue of one 32x32 accumulator tile: lane holds column n, 16 rows (mbase + MFMA row pattern)
// GEN: the kernel instantiation that also serves bias_m / beta / pre / gradact_u outputs (it needs ~80 more VGPRs for the
// loads-first schedule below; plain GEMMs run the GEN = false instantiation at twice the occupancy)
// fp16-stored output of a wave that owns TWO adjacent 32-column blocks (TN = 2; N % 64 == 0, sc_n == 1): lane l holds column l & 31 of both
// blocks, rows m (lanes < 32) and m + 4 (lanes >= 32).  One v_permlane32_swap per row pair turns that into "all 64 lanes = one row, 64
// consecutive columns", so every store instruction writes a whole 128-byte line -- the per-block fp16 store wrote two 64-byte half lines
// and lost to the fp32 store it was meant to halve (round 4: cfg3 7.24 vs 6.83 ms).
__device__ __forceinline__ void epilogue_f16_pair(const GemmDesc& d, long oc, float bn0, float bn1, const f32x16& a0, const f32x16& a1, int mbase,
                                                  int ncol0, int lane) {
  _Float16* __restrict__ Ch = reinterpret_cast<_Float16*>(d.C) + oc + ncol0 + lane;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    float x = d.alpha * a0[r] + bn0, y = d.alpha * a1[r] + bn1;
    if (d.act != ACT_NONE) { x = act_apply(d.act, x); y = act_apply(d.act, y); }
    const auto sw = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, x), __builtin_bit_cast(unsigned, y), false, false);
    const int m = mbase + (r & 3) + 8 * (r >> 2);
    if (m < d.M) Ch[(long)m * d.sc_m] = to_f16_sat(__builtin_bit_cast(float, sw[0]));
    if (m + 4 < d.M) Ch[(long)(m + 4) * d.sc_m] = to_f16_sat(__builtin_bit_cast(float, sw[1]));
  }
}

template <bool GEN>
__device__ __forceinline__ void epilogue(const GemmDesc& d, bool atomic, int bz, long oc, const float bn, const f32x16& acc, int mbase,
                                         int n, int lane) {
  if (n >= d.N) return;   // lanes l and l^32 share n, so the pair exits together (shuffle below stays well-defined)
  float* __restrict__ C = d.C + oc;
  float csum = 0.f;
  // PLAIN outputs (bias_n / activation / plain or atomic store / column sums -- every projection and weight gradient): a
  // store loop with NO load in it.  On this ISA stores count on vmcnt like loads, and a conditional load in the loop body
  // (`beta ? C[off]`, `bias_m ?`, `gradact_u ?`: even when the condition is false) is a branch whose join waits for
  // vmcnt(0), i.e. for the PREVIOUS STORE to retire: the 16 stores of a tile then run at one memory latency each --
  // measured 1.4 TB/s for the store phase of a 128000x384 output, most of the time of every tall GEMM.
  if (!GEN || (!d.bias_m && d.beta == 0.f && !d.pre && !d.gradact_u)) {
    float v[16];
    if (d.act == ACT_NONE) {
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] = d.alpha * acc[r] + bn;
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] = act_apply(d.act, d.alpha * acc[r] + bn);
    }
    float* __restrict__ Cn = C + (long)n * d.sc_n;
    if (atomic) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = mbase + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (m < d.M) { acc_add(Cn + (long)m * d.sc_m, v[r]); csum += v[r]; }
      }
    } else if (!GEN && d.c_f16) {   // fp16-stored output (saturating; plain instantiations only: in the GEN ones the extra path cost scratch)
      _Float16* __restrict__ Ch = reinterpret_cast<_Float16*>(d.C) + oc + (long)n * d.sc_n;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = mbase + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (m < d.M) { Ch[(long)m * d.sc_m] = to_f16_sat(v[r]); csum += v[r]; }
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = mbase + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (m < d.M) { Cn[(long)m * d.sc_m] = v[r]; csum += v[r]; }
      }
    }
  } else if constexpr (GEN) {
    // general outputs: every optional operand is fetched FIRST -- 8 unconditional loads back to back from clamped rows (one
    // memory latency per operand kind and half tile, not per element) -- then a compute / store pass without loads
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      float v[8], t[8];
      auto row = [&](int q) { const int r = 8 * h + q; return mbase + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); };
      auto offs = [&](int q) { const int m = row(q); return (long)(m < d.M ? m : d.M - 1) * d.sc_m + (long)n * d.sc_n; };
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = d.alpha * acc[8 * h + q] + bn;
      if (d.bias_m) {
#pragma unroll
        for (int q = 0; q < 8; ++q) { const int m = row(q); t[q] = d.bias_m[(long)bz * d.bias_m_b + (m < d.M ? m : d.M - 1)]; }
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] += t[q];
      }
      if (d.beta != 0.f) {
#pragma unroll
        for (int q = 0; q < 8; ++q) t[q] = C[offs(q)];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] += d.beta * t[q];
      }
      if (d.gradact_u) {
#pragma unroll
        for (int q = 0; q < 8; ++q) t[q] = d.gradact_u[oc + offs(q)];
      }
      if (d.pre) {
#pragma unroll
        for (int q = 0; q < 8; ++q)
          if (row(q) < d.M) d.pre[oc + offs(q)] = v[q];
      }
      if (d.gradact_u) {
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] *= act_grad(d.act, t[q]);
      } else if (d.act != ACT_NONE) {
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = act_apply(d.act, v[q]);
      }
      if (atomic) {
#pragma unroll
        for (int q = 0; q < 8; ++q)
          if (row(q) < d.M) { acc_add(&C[offs(q)], v[q]); csum += v[q]; }
      } else {
#pragma unroll
        for (int q = 0; q < 8; ++q)
          if (row(q) < d.M) { C[offs(q)] = v[q]; csum += v[q]; }
      }
    }
  }
  if (d.colsum) {   // fused bias gradient: lanes l and l^32 hold the same column
    csum += __shfl_xor(csum, 32, 64);
    if (lane < 32) acc_add(&d.colsum[(long)bz * d.colsum_b + n], csum);
  }
}

// XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs (each with its own L2), so linear ids that
// are equal mod 8 share an L2.  Re-deal the ids so that one XCD owns whole slices of the slowest grid axis: all
// column tiles of a row block (they re-read the same activation rows), all tiles of one split-K chunk / batch entry.
__device__ __forceinline__ void tile_ids(int remap, unsigned& bx, unsigned& by, unsigned& bzr) {
  bx = blockIdx.x; by = blockIdx.y; bzr = blockIdx.z;
  if (!remap) return;
  const unsigned nx = gridDim.x, ny = gridDim.y, nz = gridDim.z;
  const unsigned lin = (bzr * ny + by) * nx + bx, xcd = lin & 7u, idx = lin >> 3;
  if (nz > 1) {
    const unsigned per = nx * ny;
    if (lin < per * (nz & ~7u)) { bzr = (idx / per) * 8 + xcd; const unsigned r = idx % per; by = r / nx; bx = r % nx; }
  } else if (lin < nx * (ny & ~7u)) {
    by = (idx / nx) * 8 + xcd; bx = idx % nx;
  }
}

template <bool BF16, int BK, bool LEAN, bool GEN>
__global__ __launch_bounds__(256) void gemm_kernel(KernelArgs ka) {
  const GemmDesc& d = ka.d;
  using M = Map<BK>;
  __shared__ __attribute__((aligned(16))) Smem<BF16, BK> sm;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  unsigned bx, by, bzr;
  tile_ids(ka.xcd_remap, bx, by, bzr);
  const int bz = bzr / ka.ksplit, ks = bzr - bz * ka.ksplit;
  long oa = (long)bz * d.sa_b, ob = (long)bz * d.sb_b, oc = (long)bz * d.sc_b, obn = (long)bz * d.bias_n_b;
  if (d.batch_in > 0) {
    const int bo = bz / d.batch_in, bi = bz - bo * d.batch_in;
    oa = (long)bo * d.sa_bo + (long)bi * d.sa_b; ob = (long)bo * d.sb_bo + (long)bi * d.sb_b;
    oc = (long)bo * d.sc_bo + (long)bi * d.sc_b; obn = (long)bo * d.bias_n_bo + (long)bi * d.bias_n_b;
  }
  const int m0 = by * BM, n0 = bx * BN;
  const bool full_m = m0 + BM <= d.M, full_n = n0 + BN <= d.N;

  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  // column bias of the epilogue: requested before the k-loop (behind it, it is a dependent round trip per workgroup)
  const int n_lane = n0 + wn * 32 + (lane & 31);
  const float bn_pre = d.bias_n ? d.bias_n[obn + (n_lane < d.N ? n_lane : d.N - 1)] : 0.f;

  // one product segment: acc += A[m0:m0+64, :K] . B[:K, n0:n0+64]   (called once, or twice for C = A.B + A2.B2)
  auto segment = [&](const float* __restrict__ Ap, long sa_m, long sa_k, const float* __restrict__ Bp, long sb_k, long sb_n,
                     int K, bool veca, bool vecb, int kt0, int kt1, int ga_at, int ga) {
    const bool a_k = sa_k == 1, a_r = sa_m == 1 && !a_k;
    const bool b_k = sb_k == 1 && sb_n != 1, b_r = sb_n == 1;
    const Operand oa{Ap, sa_m, sa_k, m0, d.M, ga_at, ga, a_k || !a_r, veca && full_m && (a_k || a_r)};
    const Operand ob{Bp, sb_n, sb_k, n0, d.N, 0, 0, b_k, vecb && full_n && (b_k || b_r)};
    float ra[M::NE], rb[M::NE];
    if (kt0 < kt1) {
      load_tile<BK, LEAN>(oa, K, kt0 * BK, tid, ra);
      load_tile<BK, LEAN>(ob, K, kt0 * BK, tid, rb);
    }
    for (int kt = kt0; kt < kt1; ++kt) {
      store_tile<BF16, BK, LEAN>(oa, tid, ra, sm.a);
      store_tile<BF16, BK, LEAN>(ob, tid, rb, sm.b);
      __syncthreads();
      if (kt + 1 < kt1) {
        load_tile<BK, LEAN>(oa, K, (kt + 1) * BK, tid, ra);
        load_tile<BK, LEAN>(ob, K, (kt + 1) * BK, tid, rb);
      }
      if constexpr (BF16) {
#pragma unroll
        for (int s = 0; s < BK / 16; ++s) {
          const bf16x8 a = *reinterpret_cast<const bf16x8*>(&sm.a[wm * 32 + (lane & 31)][s * 16 + 8 * (lane >> 5)]);
          const bf16x8 b = *reinterpret_cast<const bf16x8*>(&sm.b[wn * 32 + (lane & 31)][s * 16 + 8 * (lane >> 5)]);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
        }
      } else {
#pragma unroll
        for (int s = 0; s < BK / 2; ++s) {
          const float a = sm.a[s * 2 + (lane >> 5)][wm * 32 + (lane & 31)];
          const float b = sm.b[s * 2 + (lane >> 5)][wn * 32 + (lane & 31)];
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
      }
      __syncthreads();
    }
  };
  {
    const int ktiles = (d.K + BK - 1) / BK;
    const int kt0 = ks * ka.kt_per;                       // split-K only ever applies to single-product GEMMs
    const int kt1 = kt0 + ka.kt_per < ktiles ? kt0 + ka.kt_per : ktiles;
    segment(d.A + oa, d.sa_m, d.sa_k, d.B + ob, d.sb_k, d.sb_n, d.K, ka.vec_a, ka.vec_b, kt0, kt1, d.a_gap_rows ? d.a_gap_at : 0x7fffffff, d.a_gap_rows);
  }
  if (d.A2)
    segment(d.A2 + (long)bz * d.sa2_b, d.sa2_m, d.sa2_k, d.B2 + (long)bz * d.sb2_b, d.sb2_k, d.sb2_n, d.K2, ka.vec_a2, ka.vec_b2,
            0, (d.K2 + BK - 1) / BK, 0x7fffffff, 0);

  epilogue<GEN>(d, d.atomic || ka.ksplit > 1, bz, oc, bn_pre, acc, m0 + wm * 32, n_lane, lane);
}


// =================================================================================================
// FAST bf16 path (the layouts this workload actually uses, fixed at compile time).
//   block tile (64*TM) x (64*TN), 4 waves as 2x2, each wave TM x TN accumulator tiles of v_mfma_f32_32x32x16_bf16;
//   BK = 32; operands are fp32 in HBM and are rounded to bf16 on their way into LDS (16-byte global loads, 8-byte LDS
//   stores, nothing else: no runtime strides, no per-element address math);
//   LDS double-buffered: the global loads of k-tile t+1 are in flight under the MFMAs of tile t, their LDS stores go to
//   the other buffer behind the MFMAs, ONE barrier per k-tile.
// An operand is either
//   KC  k-contiguous   (row stride s, k stride 1): LDS image [row][k] (80-byte rows), fragments by ds_read_b128;
//   RC  row-contiguous (row stride 1, k stride s): LDS image [k][row], stored exactly as loaded (4 rows of one k per
//       lane, lanes along the rows: coalesced and conflict-free) and transposed for free by the fragment reads,
//       2 x ds_read_b64_tr_b16 (each 16-lane group fetches a 4(k) x 16(row) block column-major).
// The three combinations (A,B) = (KC,KC) forward products, (KC,RC) data gradients / left-multiplications,
// (RC,RC) weight gradients cover every large GEMM of the step; anything else falls back to gemm_kernel.
// =================================================================================================
constexpr int FBK = 32;
#ifndef MIMRL_GEMM_NPF
#define MIMRL_GEMM_NPF 2
#endif
constexpr int FPF = MIMRL_GEMM_NPF;   // k-tiles in flight per workgroup of the fast kernels (register ring, see fast_body)
template <int I, int N, class F>
__device__ __forceinline__ void gemm_static_for(F&& f) {
  if constexpr (I < N) { f(std::integral_constant<int, I>{}); gemm_static_for<I + 1, N>(f); }
}
constexpr int GEMM_GROUP_MAX = 6;
template <int T>
struct FT {
  static constexpr int R = 64 * T;
  static constexpr int KCP = FBK + 8;   // [row][k]: 80-byte rows, 16-byte fragment reads conflict-free
  static constexpr int RCP = R + 32;    // [k][row]: pitch = 64 B mod 256 B -> the 4 k-rows of a transposed read hit distinct bank quarters
  static constexpr int ELEMS = (R * KCP > FBK * RCP) ? R * KCP : FBK * RCP;
  static constexpr int NV = 2 * T;      // float4 per thread per k-tile
  static constexpr int KB = 2 * T;      // RC: consecutive k per thread (T=2: 32 row-quads x 8 k-quads; T=1: 16 x 16 k-pairs)
};

// F16: the 16-bit operand type is fp16 (saturating conversion), stored in the same LDS image as bit patterns (GemmDesc::f16)
template <bool F16 = false>
__device__ __forceinline__ bf16x4 cvt4(const f32x4& q) {
  if constexpr (F16) {
    f16x4 h; h[0] = to_f16_sat(q[0]); h[1] = to_f16_sat(q[1]); h[2] = to_f16_sat(q[2]); h[3] = to_f16_sat(q[3]);
    return __builtin_bit_cast(bf16x4, h);
  } else {
    bf16x4 p; p[0] = to_bf16(q[0]); p[1] = to_bf16(q[1]); p[2] = to_bf16(q[2]); p[3] = to_bf16(q[3]);
    return p;
  }
}

// every load is unconditional from a clamped (valid) address and zeroed afterwards (fast_fix): a guarded load is a branch whose
// join waits for ALL outstanding loads.  Round 3: the loads are inline asm with ONE explicit counted wait per register set
// (fast_wait): across the k-loop's back edge the compiler's own wait insertion flushes to vmcnt(0), i.e. it waited for the tiles
// requested in the same iteration and the register ring bought nothing.
// bf16-stored operand (BF): 16 bytes = 8 elements, half as many pieces per thread, no conversion
__device__ __forceinline__ void gld16(f32x4& d, const void* p) { asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(d) : "v"(p) : "memory"); }
template <int T, bool BF> struct FL { static constexpr int N = BF ? T : 2 * T; };   // load instructions (= registers sets of 16 B) per thread and k-tile
template <int T, bool KC, bool BF>
__device__ __forceinline__ void fast_load(const float* __restrict__ P, long s, int r0, int R, int gap_at, int gap, int K, int k0,
                                          int tid, f32x4* v) {
  if constexpr (BF) {
    const __bf16* __restrict__ Pb = reinterpret_cast<const __bf16*>(P);
    if constexpr (KC) {
#pragma unroll
      for (int h = 0; h < T; ++h) {
        const int idx = h * 256 + tid, row = idx >> 2, gk = k0 + (idx & 3) * 8;
        int gr = r0 + row < R ? r0 + row : R - 1;
        if (gr >= gap_at) gr += gap;
        const int gkc = gk < K ? gk : K - 8;               // K % 8 == 0
        gld16(v[h], Pb + (long)gr * s + gkc);
      }
    } else {
      const int rq = tid & (8 * T - 1), kq = tid / (8 * T);   // T=2: 16 row-octets x 16 k-pairs; T=1: 8 x 32 k
      int gr = r0 + rq * 8;
      if (gr >= R) gr = R - 8;                              // R % 8 == 0
      if (gr >= gap_at) gr += gap;
#pragma unroll
      for (int j = 0; j < T; ++j) {
        const int gk = k0 + kq * T + j;
        const int gkc = gk < K ? gk : K - 1;
        gld16(v[j], Pb + gr + (long)gkc * s);
      }
    }
  } else if constexpr (KC) {
#pragma unroll
    for (int h = 0; h < FT<T>::NV; ++h) {
      const int idx = h * 256 + tid, row = idx >> 3, gk = k0 + (idx & 7) * 4;
      int gr = r0 + row < R ? r0 + row : R - 1;          // ragged tiles: clamped rows land in outputs nobody stores
      if (gr >= gap_at) gr += gap;
      const int gkc = gk < K ? gk : K - 4;               // K % 4 == 0
      gld16(v[h], P + (long)gr * s + gkc);
    }
  } else {
    constexpr int KB = FT<T>::KB;
    const int rq = tid & (16 * T - 1), kq = tid / (16 * T);
    int gr = r0 + rq * 4;
    if (gr >= R) gr = R - 4;                              // R % 4 == 0: a row quad is inside or outside as a whole
    if (gr >= gap_at) gr += gap;
#pragma unroll
    for (int j = 0; j < KB; ++j) {
      const int gk = k0 + kq * KB + j;
      const int gkc = gk < K ? gk : K - 1;
      gld16(v[j], P + gr + (long)gkc * s);
    }
  }
}
// the registers of one fast_load: wait until at most NEWER later-issued loads are outstanding (loads return in order)
template <int N, int NEWER>
__device__ __forceinline__ void fast_wait(f32x4* v) {
  asm volatile("s_waitcnt vmcnt(%0)" : : "n"(NEWER) : "memory");
#pragma unroll
  for (int h = 0; h < N; ++h) asm volatile("" : "+v"(v[h]));
}
// ... and zero what fast_load fetched from a clamped address (k beyond K; rows beyond R in the row-contiguous layouts)
template <int T, bool KC, bool BF>
__device__ __forceinline__ void fast_fix(int r0, int R, int K, int k0, int tid, f32x4* v) {
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  if constexpr (BF) {
    if constexpr (KC) {
#pragma unroll
      for (int h = 0; h < T; ++h) if (k0 + ((h * 256 + tid) & 3) * 8 >= K) v[h] = z;
    } else {
      const int rq = tid & (8 * T - 1), kq = tid / (8 * T);
      const bool rin = r0 + rq * 8 < R;
#pragma unroll
      for (int j = 0; j < T; ++j) if (!rin || k0 + kq * T + j >= K) v[j] = z;
    }
  } else if constexpr (KC) {
#pragma unroll
    for (int h = 0; h < FT<T>::NV; ++h) if (k0 + ((h * 256 + tid) & 7) * 4 >= K) v[h] = z;
  } else {
    constexpr int KB = FT<T>::KB;
    const int rq = tid & (16 * T - 1), kq = tid / (16 * T);
    const bool rin = r0 + rq * 4 < R;
#pragma unroll
    for (int j = 0; j < KB; ++j) if (!rin || k0 + kq * KB + j >= K) v[j] = z;
  }
}

template <int T, bool KC, bool BF, bool F16 = false>
__device__ __forceinline__ void fast_store(const f32x4* v, __bf16* __restrict__ img, int tid) {
  if constexpr (BF) {
    if constexpr (KC) {
#pragma unroll
      for (int h = 0; h < T; ++h) {
        const int idx = h * 256 + tid;
        *reinterpret_cast<f32x4*>(img + (idx >> 2) * FT<T>::KCP + (idx & 3) * 8) = v[h];
      }
    } else {
      const int rq = tid & (8 * T - 1), kq = tid / (8 * T);
#pragma unroll
      for (int j = 0; j < T; ++j) *reinterpret_cast<f32x4*>(img + (kq * T + j) * FT<T>::RCP + rq * 8) = v[j];
    }
  } else if constexpr (KC) {
#pragma unroll
    for (int h = 0; h < FT<T>::NV; ++h) {
      const int idx = h * 256 + tid;
      *reinterpret_cast<bf16x4*>(img + (idx >> 3) * FT<T>::KCP + (idx & 7) * 4) = cvt4<F16>(v[h]);
    }
  } else {
    constexpr int KB = FT<T>::KB;
    const int rq = tid & (16 * T - 1), kq = tid / (16 * T);
#pragma unroll
    for (int j = 0; j < KB; ++j) *reinterpret_cast<bf16x4*>(img + (kq * KB + j) * FT<T>::RCP + rq * 4) = cvt4<F16>(v[j]);
  }
}

// MFMA operand fragment of the 32 rows starting at `rb`, k-step s (16 k): lane holds row rb + (lane & 31), k = 8*(lane>>5) .. +7
template <int T, bool KC>
__device__ __forceinline__ bf16x8 fast_frag(const __bf16* __restrict__ img, int rb, int s, int lane) {
  if constexpr (KC) {
    return *reinterpret_cast<const bf16x8*>(img + (rb + (lane & 31)) * FT<T>::KCP + s * 16 + 8 * (lane >> 5));
  } else {
    typedef __attribute__((address_space(3))) bf16x4 lds4;
    const int j = lane & 15, q = j >> 2, p = j & 3;      // lane 4q+p of a 16-lane group addresses block row q, columns 4p..4p+3
    const __bf16* a = img + (s * 16 + 8 * (lane >> 5) + q) * FT<T>::RCP + rb + 16 * ((lane >> 4) & 1) + 4 * p;
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4*)(a));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4*)(a + 4 * FT<T>::RCP));
    bf16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3]; r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
  }
}

// BH: the bf16-typed B pieces hold fp16 values (GemmDesc::b_f16cvt): 8 halves -> 8 bf16, in place, between the load's wait and the LDS store
__device__ __forceinline__ void h8_to_bf8(f32x4& v) {
  const f16x8 h = __builtin_bit_cast(f16x8, v);
  bf16x8 o;
#pragma unroll
  for (int i = 0; i < 8; ++i) o[i] = to_bf16((float)h[i]);
  v = __builtin_bit_cast(f32x4, o);
}

template <int TM, int TN, bool AKC, bool BKC, bool GEN, bool ABF = false, bool BBF = false, bool F16 = false, bool BH = false>
__device__ __forceinline__ void fast_body(const GemmDesc& d, int ksplit, int kt_per, int dbg, unsigned bx, unsigned by, unsigned bzr) {
  constexpr int BMf = 64 * TM, BNf = 64 * TN;
  __shared__ __attribute__((aligned(16))) __bf16 sA[2][FT<TM>::ELEMS];
  __shared__ __attribute__((aligned(16))) __bf16 sB[2][FT<TN>::ELEMS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int bz = bzr / ksplit, ks = bzr - bz * ksplit;
  long oa = (long)bz * d.sa_b, ob = (long)bz * d.sb_b, oc = (long)bz * d.sc_b, obn = (long)bz * d.bias_n_b;
  if (d.batch_in > 0) {
    const int bo = bz / d.batch_in, bi = bz - bo * d.batch_in;
    oa = (long)bo * d.sa_bo + (long)bi * d.sa_b; ob = (long)bo * d.sb_bo + (long)bi * d.sb_b;
    oc = (long)bo * d.sc_bo + (long)bi * d.sc_b; obn = (long)bo * d.bias_n_bo + (long)bi * d.bias_n_b;
  }
  const int m0 = by * BMf, n0 = bx * BNf;
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float bn_pre[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = n0 + wn * 32 * TN + j * 32 + (lane & 31);
    bn_pre[j] = d.bias_n ? d.bias_n[obn + (n < d.N ? n : d.N - 1)] : 0.f;
  }

  // Round 3: the k-loop keeps FPF k-tiles in flight in registers (a ring of FPF register sets), not one.  With one, every 32-wide
  // k-tile cost a full memory round trip (8 MFMAs hide ~0.1 us of it): the K = 768 text projection took 31 us for 20 MB, the K = 256
  // layer-1 input projection 28-32 us.  Every request is unconditional (a tile index past the last one re-reads the last tile into
  // registers that are never stored) so that the compiler's vmcnt counts stay exact: the store of tile k + 1 waits for that tile only.
  auto segment = [&](const float* __restrict__ Ap, long sa, const float* __restrict__ Bp, long sb, int K, int kt0, int kt1, int ga_at, int ga) {
    constexpr int LA = FL<TM, ABF>::N, LB = FL<TN, BBF>::N;
    f32x4 ra[FPF][LA], rb[FPF][LB];
    if (kt0 >= kt1) return;
    const int last = kt1 - 1;
    auto request = [&](auto J, int kt) __attribute__((always_inline)) {
      constexpr int j = decltype(J)::value;
      const int t = kt < last ? kt : last;
      fast_load<TM, AKC, ABF>(Ap, sa, m0, d.M, ga_at, ga, K, t * FBK, tid, ra[j]);
      fast_load<TN, BKC, BBF>(Bp, sb, n0, d.N, 0x7fffffff, 0, K, t * FBK, tid, rb[j]);
    };
    // set j holds tile kt (clamped like the request): FPF - 1 newer sets may still be in flight behind it
    auto publish = [&](auto J, int buf, int kt) __attribute__((always_inline)) {
      constexpr int j = decltype(J)::value;
      const int t = kt < last ? kt : last;
      fast_wait<LA, (FPF - 1) * (LA + LB) + LB>(ra[j]);
      fast_wait<LB, (FPF - 1) * (LA + LB)>(rb[j]);
      fast_fix<TM, AKC, ABF>(m0, d.M, K, t * FBK, tid, ra[j]);
      fast_fix<TN, BKC, BBF>(n0, d.N, K, t * FBK, tid, rb[j]);
      if constexpr (BH) {
#pragma unroll
        for (int h = 0; h < LB; ++h) h8_to_bf8(rb[j][h]);
      }
      fast_store<TM, AKC, ABF, F16>(ra[j], sA[buf], tid);
      fast_store<TN, BKC, BBF, F16>(rb[j], sB[buf], tid);
    };
    gemm_static_for<0, FPF>([&](auto J) __attribute__((always_inline)) { request(J, kt0 + decltype(J)::value); });
    publish(std::integral_constant<int, 0>{}, 0, kt0);
    __syncthreads();
    int cur = 0;
    for (int kb = kt0; kb < kt1; kb += FPF) {
      bool done = false;
      gemm_static_for<0, FPF>([&](auto J) __attribute__((always_inline)) {
        constexpr int j = decltype(J)::value;
        const int kt = kb + j;
        if (done || kt >= kt1) { done = true; return; }
        request(J, kt + FPF);                       // set j was published one iteration ago: free
#pragma unroll
        for (int s = 0; s < FBK / 16; ++s) {
          bf16x8 af[TM], bfr[TN];
#pragma unroll
          for (int i = 0; i < TM; ++i) af[i] = fast_frag<TM, AKC>(sA[cur], wm * 32 * TM + i * 32, s, lane);
#pragma unroll
          for (int jn = 0; jn < TN; ++jn) bfr[jn] = fast_frag<TN, BKC>(sB[cur], wn * 32 * TN + jn * 32, s, lane);
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int jn = 0; jn < TN; ++jn) {
              if constexpr (F16) acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, af[i]), __builtin_bit_cast(f16x8, bfr[jn]), acc[i][jn], 0, 0, 0);
              else acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[jn], acc[i][jn], 0, 0, 0);
            }
        }
        publish(std::integral_constant<int, (j + 1) % FPF>{}, cur ^ 1, kt + 1);   // tile kt + 1 (past the end: the last tile again, never read)
        __syncthreads();
        cur ^= 1;
      });
    }
    asm volatile("s_waitcnt vmcnt(0)" : : : "memory");   // the ring's last (duplicate) requests: nothing may land in a dead register
  };
  {
    const int ktiles = (d.K + FBK - 1) / FBK;
    const int kt0 = ks * kt_per;
    const int kt1 = dbg == 2 ? kt0 : (kt0 + kt_per < ktiles ? kt0 + kt_per : ktiles);
    // (element offsets: a bf16 operand's base is advanced in bf16 elements)
    const float* Ab = ABF ? reinterpret_cast<const float*>(reinterpret_cast<const __bf16*>(d.A) + oa) : d.A + oa;
    const float* Bb = BBF ? reinterpret_cast<const float*>(reinterpret_cast<const __bf16*>(d.B) + ob) : d.B + ob;
    segment(Ab, AKC ? d.sa_m : d.sa_k, Bb, BKC ? d.sb_n : d.sb_k, d.K, kt0, kt1, d.a_gap_rows ? d.a_gap_at : 0x7fffffff, d.a_gap_rows);
  }
  if (d.A2) {
    const long oa2 = (long)bz * d.sa2_b, ob2 = (long)bz * d.sb2_b;
    const float* Ab = ABF ? reinterpret_cast<const float*>(reinterpret_cast<const __bf16*>(d.A2) + oa2) : d.A2 + oa2;
    const float* Bb = BBF ? reinterpret_cast<const float*>(reinterpret_cast<const __bf16*>(d.B2) + ob2) : d.B2 + ob2;
    segment(Ab, AKC ? d.sa2_m : d.sa2_k, Bb, BKC ? d.sb2_n : d.sb2_k, d.K2, 0, (d.K2 + FBK - 1) / FBK, 0x7fffffff, 0);
  }
  if (dbg == 1) { if (acc[0][0][0] == 123.456f) d.C[0] = 1.f; return; }
  if constexpr (!GEN && TN == 2) {
    if (d.c_f16 && d.sc_n == 1 && d.N % 64 == 0) {   // (gemm() guarantees the plain store for c_f16)
#pragma unroll
      for (int i = 0; i < TM; ++i)
        epilogue_f16_pair(d, oc, bn_pre[0], bn_pre[1], acc[i][0], acc[i][1], m0 + wm * 32 * TM + i * 32, n0 + wn * 64, lane);
      return;
    }
  }
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
      epilogue<GEN>(d, d.atomic || ksplit > 1, bz, oc, bn_pre[j], acc[i][j], m0 + wm * 32 * TM + i * 32, n0 + wn * 32 * TN + j * 32 + (lane & 31), lane);
}

template <int TM, int TN, bool AKC, bool BKC, bool GEN>
__global__ __launch_bounds__(256) void gemm_fast_kernel(KernelArgs ka) {
  unsigned bx, by, bzr;
  tile_ids(ka.xcd_remap, bx, by, bzr);
  fast_body<TM, TN, AKC, BKC, GEN>(ka.d, ka.ksplit, ka.kt_per, ka.dbg, bx, by, bzr);
}
// fp16 operands (GemmDesc::f16): forward products (KC,KC) only
template <int TM, int TN, bool GEN>
__global__ __launch_bounds__(256) void gemm_fast_f16_kernel(KernelArgs ka) {
  unsigned bx, by, bzr;
  tile_ids(ka.xcd_remap, bx, by, bzr);
  fast_body<TM, TN, true, true, GEN, false, false, true>(ka.d, ka.ksplit, ka.kt_per, ka.dbg, bx, by, bzr);
}
// both operands STORED as fp16, k-contiguous (a forward product whose operands the producer already wrote in 16 bits: the layer-1 GRU
// input projection reads the layer-0 recurrence's fp16 copy of h and an fp16 weight image -- half the L2 -> LDS bytes of the kernel
// above, which is what bounds it: 235 MB per launch at cfg2, 3.1 GB at cfg3, ~7 TB/s either way).  Same values, same MFMA, same
// accumulation order as gemm_fast_f16_kernel: the conversion happened at the producer's store instead of at this kernel's load.
template <int TM, int TN>
__global__ __launch_bounds__(256) void gemm_fast_f16s_kernel(KernelArgs ka) {
  unsigned bx, by, bzr;
  tile_ids(ka.xcd_remap, bx, by, bzr);
  fast_body<TM, TN, true, true, false, true, true, true>(ka.d, ka.ksplit, ka.kt_per, ka.dbg, bx, by, bzr);
}
// bf16-stored operands (plain epilogue): A bf16 with B fp32 (data gradient dh0, dW_ih against fp32 inputs) or both bf16 (dW_hh)
template <int TM, int TN, bool AKC, bool BKC, bool BBF>
__global__ __launch_bounds__(256) void gemm_fast_bf_kernel(KernelArgs ka) {
  unsigned bx, by, bzr;
  tile_ids(ka.xcd_remap, bx, by, bzr);
  fast_body<TM, TN, AKC, BKC, false, true, BBF>(ka.d, ka.ksplit, ka.kt_per, ka.dbg, bx, by, bzr);
}

// A bf16, B stored as fp16 and converted (GemmDesc::b_f16cvt), both row-contiguous: the layer-1 dW_ih product
template <int TM, int TN>
__global__ __launch_bounds__(256) void gemm_fast_bfh_kernel(KernelArgs ka) {
  unsigned bx, by, bzr;
  tile_ids(ka.xcd_remap, bx, by, bzr);
  fast_body<TM, TN, false, false, false, true, true, false, true>(ka.d, ka.ksplit, ka.kt_per, ka.dbg, bx, by, bzr);
}

// GROUPED launch: up to GEMM_GROUP_MAX independent plain GEMMs of one layout class (64x64 tiles, no split-K) in ONE launch --
// the weight gradients of an estimator MLP stack (4 layers x 10 towers / 6 classifiers) are 4-5 launches of ~10 us of work
// each on the critical branch of stage 1; as one launch they fill the chip once.
struct GroupArgs {
  GemmDesc d[GEMM_GROUP_MAX];
  int start[GEMM_GROUP_MAX + 1];   // first linear workgroup id of each problem
  int n;
};
template <bool AKC, bool BKC>
__global__ __launch_bounds__(256) void gemm_group_kernel(GroupArgs ga) {
  int p = 0;
#pragma unroll
  for (int q = 1; q < GEMM_GROUP_MAX; ++q)
    if (q < ga.n && (int)blockIdx.x >= ga.start[q]) p = q;
  const GemmDesc& d = ga.d[p];
  const unsigned t = blockIdx.x - ga.start[p];
  const unsigned nx = (d.N + 63) / 64, ny = (d.M + 63) / 64;
  fast_body<1, 1, AKC, BKC, false>(d, 1, (d.K + FBK - 1) / FBK, 0, t % nx, (t / nx) % ny, t / (nx * ny));
}

// GROUPED SPLIT-K launch: up to GEMM_GROUPK_MAX plain accumulate-into-zeroed-output GEMMs (weight gradients: C += A.B with float
// atomics, optional batch reduced into the same C) of one layout class in ONE launch, each with its own split-K factor.  The parked
// CubeMLP weight gradients are 12 such products of 4..64 tiles each: alone every one is a ~20 us launch that cannot fill the chip.
constexpr int GEMM_GROUPK_MAX = 12;
struct GroupKProb {
  const float* A; const float* B; float* C;
  int M, N, K, batch;
  long sa, sa_b, sb, sb_b, sc_m, sc_b;   // sa / sb: the non-unit stride of the operand (row stride for KC, k stride for RC)
  int nsplit, kt_per;
};
struct GroupKArgs {
  GroupKProb p[GEMM_GROUPK_MAX];
  int start[GEMM_GROUPK_MAX];            // first linear workgroup id of each problem
  int n;
};
template <bool AKC, bool BKC>
__global__ __launch_bounds__(256) void gemm_groupk_kernel(GroupKArgs ga) {
  int p = 0;
#pragma unroll
  for (int q = 1; q < GEMM_GROUPK_MAX; ++q)
    if (q < ga.n && (int)blockIdx.x >= ga.start[q]) p = q;
  const GroupKProb& g = ga.p[p];
  GemmDesc d;
  d.A = g.A; d.B = g.B; d.C = g.C; d.M = g.M; d.N = g.N; d.K = g.K; d.batch = g.batch;
  d.sa_m = AKC ? g.sa : 1; d.sa_k = AKC ? 1 : g.sa; d.sa_b = g.sa_b;
  d.sb_n = BKC ? g.sb : 1; d.sb_k = BKC ? 1 : g.sb; d.sb_b = g.sb_b;
  d.sc_m = g.sc_m; d.sc_n = 1; d.sc_b = g.sc_b; d.atomic = 1;
  const unsigned t = blockIdx.x - ga.start[p];
  const unsigned nx = (g.N + 63) / 64, ny = (g.M + 63) / 64;
  fast_body<1, 1, AKC, BKC, false>(d, g.nsplit, g.kt_per, 0, t % nx, (t / nx) % ny, t / (nx * ny));
}

// layout class of one operand for the fast path: 1 = KC, 2 = RC, 0 = not eligible.  (row axis = m for A, n for B)
inline int fast_class(const float* P, long s_r, long s_k, long s_b, long s_bo, int R, int K, int bf = 0, int pad4 = 0) {
  const int q = bf ? 8 : 4;   // elements per 16-byte piece
  if (!aligned16(P) || s_b % q != 0 || s_bo % q != 0) return 0;
  const bool pad = pad4 && !bf;   // GemmDesc::a_pad4: the contiguous axis is readable (zero) up to the next multiple of 4
  if (s_k == 1 && s_r % q == 0 && (K % q == 0 || pad) && K >= q && s_r != 1) return 1;
  if (s_r == 1 && s_k % q == 0 && (R % q == 0 || pad) && R >= q) return 2;
  return 0;
}

// 16-byte path: unit stride along one axis, the other stride and the batch stride multiples of 4 floats, base aligned
inline bool vec_ok_a(const float* P, long s_m, long s_k, long s_b) {
  return aligned16(P) && s_b % 4 == 0 && ((s_k == 1 && s_m % 4 == 0) || (s_m == 1 && s_k != 1 && s_k % 4 == 0));
}
inline bool vec_ok_b(const float* P, long s_k, long s_n, long s_b) {
  return aligned16(P) && s_b % 4 == 0 && ((s_k == 1 && s_n != 1 && s_n % 4 == 0) || (s_n == 1 && s_k % 4 == 0));
}

}  // namespace

static bool plain_accumulate(const GemmDesc& d) {
  return !d.A2 && d.atomic && d.beta == 0.f && !d.bias_n && !d.bias_m && !d.pre && !d.gradact_u && !d.colsum && d.act == ACT_NONE;
}

// fast-path eligibility and tile shape; returns false when the generic kernel has to take the GEMM
static bool fast_plan(const GemmDesc& d, bool bf16, GemmPlan* p) {
  constexpr int no_fast = 0;   // (MIMRL_GEMM_NO_FAST went in round 6: the parametrised knob test found it BROKEN in bf16 mode -- bf16-stored operands exist on the fast path only)
  if (!bf16 || no_fast) return false;
  const int ca = fast_class(d.A, d.sa_m, d.sa_k, d.sa_b, d.sa_bo, d.M, d.K, d.a_bf16, d.a_pad4), cb = fast_class(d.B, d.sb_n, d.sb_k, d.sb_b, d.sb_bo, d.N, d.K, d.b_bf16);
  if (!ca || !cb || (ca == 2 && cb == 1)) return false;
  if (d.A2 && (fast_class(d.A2, d.sa2_m, d.sa2_k, d.sa2_b, 0, d.M, d.K2, d.a_bf16) != ca || fast_class(d.B2, d.sb2_n, d.sb2_k, d.sb2_b, 0, d.N, d.K2, d.b_bf16) != cb))
    return false;
  if (d.a_bf16 || d.b_bf16) {   // instantiated: A bf16 (+ B bf16), plain epilogue, layouts (KC,RC) and (RC,RC); gaps in 8-row units
    // (KC, KC) -- a forward product -- only as the fp16-operand kernel: both operands stored as fp16 (GemmDesc::f16)
    if (!d.a_bf16 || (ca == 1 && cb == 1 && !(d.f16 && d.b_bf16)) || d.bias_m || d.beta != 0.f || d.pre || d.gradact_u) return false;
    if (d.f16 && !(ca == 1 && cb == 1 && d.b_bf16)) return false;   // fp16 storage exists for that product only
    if (d.a_gap_rows && (d.a_gap_at % 8 != 0 || d.a_gap_rows % 8 != 0)) return false;
    if (d.b_f16cvt && !(ca == 2 && cb == 2 && d.b_bf16 && !d.A2)) return false;   // instantiated for the weight-gradient layout only
  }
  const int ktiles = (d.K + FBK - 1) / FBK;
  const bool acc = plain_accumulate(d) && ktiles >= 32;
  auto tiles = [&](int tm, int tn) { return (long)((d.M + 64 * tm - 1) / (64 * tm)) * ((d.N + 64 * tn - 1) / (64 * tn)) * d.batch; };
  auto split = [&](long t) {   // split-K factor for accumulate-into-zeroed-output GEMMs: ~2 workgroups per CU, >= 16 k-tiles each
    if (!acc) return 1;
    long ks = (512 + t - 1) / t;
    if (ks > ktiles / 16) ks = ktiles / 16;
    return (int)(ks < 1 ? 1 : ks);
  };
  const int cand[3][2] = {{2, 2}, {2, 1}, {1, 1}};
  int pick = 2;
  for (int c = 0; c < 3; ++c) {
    const int tm = cand[c][0], tn = cand[c][1];
    if (d.M < 64 * tm || d.N < 64 * tn) continue;
    const long t = tiles(tm, tn);
    constexpr long min_wgs = 700;   // (an environment knob until round 5: fixed at its measured optimum).  Round 3b: 224 -> 700 (cfg2, 4 interleaved runs each: 0.906-0.912 -> 0.886-0.901 ms; 1000 / 1250 the same): a workgroup is a serial prologue - loop - epilogue and a CU overlaps them only ACROSS workgroups, so ~3 per CU beat ~2 larger ones
    // (not for split-K accumulations over a long reduction: at cfg3's K = 128,000 rows smaller tiles mean more splits and more atomics,
    //  7.11 -> 7.52 ms with 700 for all)
    if (t * split(t) >= ((acc && ktiles > 1024) ? 224 : min_wgs) || c == 2) { pick = c; break; }
  }
  // round 5b: weight gradients over a LONG reduction with both operands 16-bit stored and N a multiple of 256 (dW_ih of GRU layer 1:
  // [384 x 256] per direction over B * T rows) take 128 x 256 tiles -- every row of A (dg: the larger operand) is then read by ONE workgroup
  // instead of two; the split-K chunks of a 128 x 128 grid drift apart in time and their re-reads miss the 4 MB L2 (1.39 GB fetched for 0.66)
  constexpr int wide_n = 0;   // (an environment knob until round 5: fixed at its measured optimum): 1 all, 2 M % 256 == 0 only, 3 the others only
  const bool wide = wide_n && (wide_n == 1 || (wide_n == 2) == (d.M % 256 == 0)) && acc && ktiles > 1024 && ca == 2 && cb == 2 && d.a_bf16 && d.b_bf16 && d.N % 256 == 0 && d.M >= 128;
  if (wide) {
    p->fast = 1; p->tm = 2; p->tn = 4; p->ca = ca; p->cb = cb;
    p->tiles = tiles(2, 4); p->nsplit = split(p->tiles); p->kt_per = (ktiles + p->nsplit - 1) / p->nsplit;
    p->variant = 24;
    return true;
  }
  p->fast = 1; p->tm = cand[pick][0]; p->tn = cand[pick][1]; p->ca = ca; p->cb = cb;
  p->tiles = tiles(p->tm, p->tn);
  p->nsplit = split(p->tiles);
  p->kt_per = (ktiles + p->nsplit - 1) / p->nsplit;
  p->variant = 10 + 4 * pick + (ca == 1 ? (cb == 1 ? 0 : 1) : 2);
  return true;
}

void gemm_plan(const GemmDesc& d, bool bf16, GemmPlan* p) {
  p->fast = 0; p->tm = p->tn = 1; p->ca = p->cb = 0;
  if (fast_plan(d, bf16, p)) return;
  constexpr int no_lean = 0;   // (an environment knob until round 5: fixed at its measured optimum)
  constexpr int no_big = 0;
  const bool va = vec_ok_a(d.A, d.sa_m, d.sa_k, d.sa_b) && d.sa_bo % 4 == 0, vb = vec_ok_b(d.B, d.sb_k, d.sb_n, d.sb_b) && d.sb_bo % 4 == 0;
  const bool va2 = d.A2 ? vec_ok_a(d.A2, d.sa2_m, d.sa2_k, d.sa2_b) : true, vb2 = d.B2 ? vec_ok_b(d.B2, d.sb2_k, d.sb2_n, d.sb2_b) : true;
  // lean = 16-byte loads only.  Ragged M / N are fine for k-contiguous operands (rows are clamped in the loader).
  const bool a_kfast = d.sa_k == 1 && (!d.A2 || d.sa2_k == 1), b_kfast = d.sb_k == 1 && d.sb_n != 1 && (!d.B2 || (d.sb2_k == 1 && d.sb2_n != 1));
  constexpr int no_ragged = 0;
  const bool lean = bf16 && !no_lean && (d.M % BM == 0 || (a_kfast && !no_ragged)) && (d.N % BN == 0 || (b_kfast && !no_ragged)) &&
                    va && vb && va2 && vb2 && d.K >= 64;
  const long tiles = (long)((d.N + BN - 1) / BN) * ((d.M + BM - 1) / BM) * d.batch;
  // small grids gain nothing from occupancy: stage 128 k-values per round trip (one-shot for K <= 128)
  const bool lean128 = lean && !no_big && tiles <= 256 && d.K >= 128 && (!d.A2 || d.K2 >= 64);
  p->variant = !bf16 ? 0 : lean128 ? 3 : lean ? 2 : 1;
  const int BKh = lean128 ? 128 : (lean ? 64 : 32);
  const int ktiles = (d.K + BKh - 1) / BKh;
  p->nsplit = 1; p->kt_per = ktiles; p->tiles = tiles;
  if (plain_accumulate(d) && ktiles >= 8) {
    // accumulate-into-zeroed-output GEMMs (weight gradients): split K until the grid has ~2 waves of workgroups
    int ksplit = (int)((512 + tiles - 1) / tiles);
    if (ksplit > ktiles / 4) ksplit = ktiles / 4;
    if (ksplit < 1) ksplit = 1;
    p->nsplit = ksplit;
    p->kt_per = (ktiles + ksplit - 1) / ksplit;
  }
}

int gemm_group(hipStream_t s, const GemmDesc* ds, int n, bool bf16) {
  constexpr int no_group = 0;   // (an environment knob until round 5: fixed at its measured optimum)
  bool ok = bf16 && !no_group && n >= 2 && n <= GEMM_GROUP_MAX;
  int ca = 0, cb = 0;
  long total = 0;
  for (int i = 0; ok && i < n; ++i) {
    const GemmDesc& d = ds[i];
    if (d.M <= 0 || d.N <= 0 || d.batch <= 0 || !d.A || !d.B || !d.C) { ok = false; break; }
    const int a = fast_class(d.A, d.sa_m, d.sa_k, d.sa_b, d.sa_bo, d.M, d.K), b = fast_class(d.B, d.sb_n, d.sb_k, d.sb_b, d.sb_bo, d.N, d.K);
    if (!a || !b || (a == 2 && b == 1) || (i > 0 && (a != ca || b != cb))) { ok = false; break; }
    ca = a; cb = b;
    if (d.A2 || d.bias_m || d.beta != 0.f || d.pre || d.gradact_u || d.a_gap_rows || d.a_bf16 || d.b_bf16) { ok = false; break; }   // plain epilogue, single product, fp32 storage
    total += (long)((d.M + 63) / 64) * ((d.N + 63) / 64) * d.batch;
  }
  if (!ok || total > 65535L * 16) {
    for (int i = 0; i < n; ++i) MX(gemm(s, ds[i], bf16));
    return MIMRL_OK;
  }
  GroupArgs ga;
  ga.n = n;
  int acc = 0;
  for (int i = 0; i < GEMM_GROUP_MAX; ++i) {
    if (i < n) { ga.d[i] = ds[i]; ga.start[i] = acc; acc += ((ds[i].M + 63) / 64) * ((ds[i].N + 63) / 64) * ds[i].batch; }
    else { ga.d[i] = GemmDesc(); ga.start[i] = 0x7fffffff; }
  }
  ga.start[GEMM_GROUP_MAX] = acc;
  if (ca == 1 && cb == 1) hipLaunchKernelGGL((gemm_group_kernel<true, true>), dim3(acc), dim3(256), 0, s, ga);
  else if (ca == 1) hipLaunchKernelGGL((gemm_group_kernel<true, false>), dim3(acc), dim3(256), 0, s, ga);
  else hipLaunchKernelGGL((gemm_group_kernel<false, false>), dim3(acc), dim3(256), 0, s, ga);
  LAUNCH_CHECK();
  return MIMRL_OK;
}

int gemm_group_splitk(hipStream_t s, const GemmDesc* ds, int n, bool bf16) {
  constexpr int no_group = 0;   // (an environment knob until round 5: fixed at its measured optimum)
  bool ok = bf16 && !no_group && n >= 2 && n <= GEMM_GROUPK_MAX;
  int ca = 0, cb = 0;
  long tiles = 0;
  double macs = 0;
  for (int i = 0; ok && i < n; ++i) {
    const GemmDesc& d = ds[i];
    if (d.M <= 0 || d.N <= 0 || d.batch <= 0 || !d.A || !d.B || !d.C) { ok = false; break; }
    const int a = fast_class(d.A, d.sa_m, d.sa_k, d.sa_b, d.sa_bo, d.M, d.K), b = fast_class(d.B, d.sb_n, d.sb_k, d.sb_b, d.sb_bo, d.N, d.K);
    if (!a || !b || a != b || (i > 0 && a != ca)) { ok = false; break; }
    ca = a; cb = b;
    if (!plain_accumulate(d) || d.alpha != 1.f || d.sc_n != 1 || d.batch_in > 0 || d.a_gap_rows || d.a_bf16 || d.b_bf16) { ok = false; break; }
    tiles += (long)((d.M + 63) / 64) * ((d.N + 63) / 64) * d.batch;
    macs += (double)d.M * d.N * d.K * d.batch;
  }
  // large products fill the chip on their own and want the 128-wide tiles of the single launches (measured: grouping costs
  // 90 us at cfg3 and 100 us at cfg5, saves 50 us at cfg2)
  if (macs > 3e9) ok = false;
  if (!ok) {
    for (int i = 0; i < n; ++i) MX(gemm(s, ds[i], bf16));
    return MIMRL_OK;
  }
  (void)cb;
  GroupKArgs ga;
  ga.n = n;
  // ~3 workgroups per CU over the whole group, at least 8 k-tiles (256 k) per workgroup
  const long want = (768 + tiles - 1) / tiles;
  long acc = 0;
  for (int i = 0; i < GEMM_GROUPK_MAX; ++i) {
    if (i >= n) { ga.p[i] = GroupKProb(); ga.start[i] = 0x7fffffff; continue; }
    const GemmDesc& d = ds[i];
    GroupKProb& g = ga.p[i];
    g.A = d.A; g.B = d.B; g.C = d.C; g.M = d.M; g.N = d.N; g.K = d.K; g.batch = d.batch;
    g.sa = ca == 1 ? d.sa_m : d.sa_k; g.sa_b = d.sa_b; g.sb = ca == 1 ? d.sb_n : d.sb_k; g.sb_b = d.sb_b;
    g.sc_m = d.sc_m; g.sc_b = d.sc_b;
    const int ktiles = (d.K + FBK - 1) / FBK;
    long ks = want < 1 ? 1 : want;
    if (ks > ktiles / 8) ks = ktiles / 8;
    if (ks < 1) ks = 1;
    g.kt_per = (int)((ktiles + ks - 1) / ks);
    g.nsplit = (ktiles + g.kt_per - 1) / g.kt_per;
    ga.start[i] = (int)acc;
    acc += (long)((d.M + 63) / 64) * ((d.N + 63) / 64) * d.batch * g.nsplit;
  }
  if (acc > 0x7fffffffL) return set_error(MIMRL_ERR_ARG, "gemm_group_splitk: grid too large");
  if (ca == 1) hipLaunchKernelGGL((gemm_groupk_kernel<true, true>), dim3((unsigned)acc), dim3(256), 0, s, ga);
  else hipLaunchKernelGGL((gemm_groupk_kernel<false, false>), dim3((unsigned)acc), dim3(256), 0, s, ga);
  LAUNCH_CHECK();
  return MIMRL_OK;
}

int gemm(hipStream_t s, const GemmDesc& d, bool bf16) {
  if (d.M <= 0 || d.N <= 0 || d.batch <= 0) return MIMRL_OK;
  static const int dbg_skip = dbg_env("MIMRL_DBG_SKIP_WGRAD") ? atoi(dbg_env("MIMRL_DBG_SKIP_WGRAD")) : 0;   // timing experiments only
  if (dbg_skip && d.atomic && (dbg_skip == 1 || (dbg_skip == 2 && d.batch > 1 && d.sc_b == 0))) return MIMRL_OK;
  if (!d.A || !d.B || !d.C) return set_error(MIMRL_ERR_ARG, "gemm: null operand");
  if ((d.A2 != nullptr) != (d.B2 != nullptr)) return set_error(MIMRL_ERR_ARG, "gemm: second product needs both operands");
  if (d.a_gap_rows && (d.a_gap_at % 4 != 0 || d.a_gap_rows % 4 != 0 || d.A2))
    return set_error(MIMRL_ERR_ARG, "gemm: the A-row gap must be a multiple of 4 rows (single product only)");
  if (d.batch_in > 0 && (d.A2 || d.bias_m || d.colsum || d.batch % d.batch_in != 0))
    return set_error(MIMRL_ERR_ARG, "gemm: two-level batch supports A, B, C, bias_n only");
  if (bf16 && gemm_tall_ok(d)) return gemm_tall(s, d);   // tall 16-bit-stored (KC, KC) products: the LDS-DMA kernel of gemm_tall.hip
  if (d.c_bf16) return set_error(MIMRL_ERR_ARG, "gemm: a bf16-stored output exists in the tall LDS-DMA kernel only (gemm_tall_ok)");
  GemmPlan pl;
  gemm_plan(d, bf16, &pl);
#ifdef MIMRL_GEMM_TRACE_BUILD     // diagnostic build (-DMIMRL_GEMM_TRACE_BUILD): which products miss the fast path, and why
  constexpr bool trace = true;
#else
  constexpr bool trace = false;
#endif
  if (trace && bf16 && !pl.fast)
    fprintf(stderr, "[gemm] generic: M %d N %d K %d batch %d | A %p sa %ld %ld %ld cls %d | B %p sb %ld %ld %ld cls %d | A2 %d K2 %d bias_m %d beta %g pre %d gu %d atomic %d\n",
            d.M, d.N, d.K, d.batch, (const void*)d.A, d.sa_m, d.sa_k, d.sa_b, fast_class(d.A, d.sa_m, d.sa_k, d.sa_b, d.sa_bo, d.M, d.K, d.a_bf16),
            (const void*)d.B, d.sb_k, d.sb_n, d.sb_b, fast_class(d.B, d.sb_n, d.sb_k, d.sb_b, d.sb_bo, d.N, d.K, d.b_bf16), d.A2 != nullptr, d.K2,
            d.bias_m != nullptr, (double)d.beta, d.pre != nullptr, d.gradact_u != nullptr, d.atomic);
  if (d.c_f16 && (d.atomic || pl.nsplit > 1 || d.bias_m || d.beta != 0.f || d.pre || d.gradact_u || d.colsum))
    return set_error(MIMRL_ERR_ARG, "gemm: an fp16-stored output takes the plain store only (no atomic / split-K / beta / pre / column sums)");
  // (deterministic build: only a product that accumulates -- atomics, split-K, fused column sums -- needs the table flushed behind it)
  DetNoFlush det_nf(!(d.atomic || pl.nsplit > 1 || d.colsum));
  DetGemmTarget det_gt(d.colsum ? nullptr : d.C);   // (a weight gradient into a bucket: flushed at the end of the pass, det.h DetDefer)
  KernelArgs ka;
  ka.d = d;
  ka.vec_a = vec_ok_a(d.A, d.sa_m, d.sa_k, d.sa_b) && d.sa_bo % 4 == 0;
  ka.vec_b = vec_ok_b(d.B, d.sb_k, d.sb_n, d.sb_b) && d.sb_bo % 4 == 0;
  ka.vec_a2 = d.A2 ? vec_ok_a(d.A2, d.sa2_m, d.sa2_k, d.sa2_b) : 1;
  ka.vec_b2 = d.B2 ? vec_ok_b(d.B2, d.sb2_k, d.sb2_n, d.sb2_b) : 1;
  constexpr int no_xcd = 0;   // (an environment knob until round 5: fixed at its measured optimum)
  ka.xcd_remap = !no_xcd;
  static const int dbg_gemm = dbg_env("MIMRL_DBG_GEMM") ? atoi(dbg_env("MIMRL_DBG_GEMM")) : 0;
  ka.dbg = dbg_gemm;
  ka.ksplit = pl.nsplit;
  ka.kt_per = pl.kt_per;
  const int bm = 64 * pl.tm, bn = 64 * pl.tn;   // (64 x 64 unless the fast path picked a larger tile)
  dim3 grid((d.N + bn - 1) / bn, (d.M + bm - 1) / bm, d.batch * pl.nsplit);
  if (grid.y > 65535 || grid.z > 65535) return set_error(MIMRL_ERR_ARG, "gemm: grid too large (M=%d batch=%d)", d.M, d.batch);
  const bool gen = d.bias_m || d.beta != 0.f || d.pre || d.gradact_u;
  if ((d.a_bf16 || d.b_bf16) && !pl.fast)
    return set_error(MIMRL_ERR_ARG, "gemm: bf16-stored operands need the fast path (bf16 mode, 8-element alignment, A bf16, layouts KC/RC or RC/RC, plain epilogue)");
  if (d.b_f16cvt && !(d.a_bf16 && d.b_bf16)) return set_error(MIMRL_ERR_ARG, "gemm: b_f16cvt needs a_bf16 = b_bf16 = 1");
#define FASTK(TM_, TN_, A_, B_)                                                                               \
  if (d.f16 && d.a_bf16) hipLaunchKernelGGL((gemm_fast_f16s_kernel<TM_, TN_>), grid, dim3(256), 0, s, ka);    \
  else if (d.f16 && gen) hipLaunchKernelGGL((gemm_fast_f16_kernel<TM_, TN_, true>), grid, dim3(256), 0, s, ka);    \
  else if (d.f16) hipLaunchKernelGGL((gemm_fast_f16_kernel<TM_, TN_, false>), grid, dim3(256), 0, s, ka);     \
  else if (gen) hipLaunchKernelGGL((gemm_fast_kernel<TM_, TN_, A_, B_, true>), grid, dim3(256), 0, s, ka);    \
  else hipLaunchKernelGGL((gemm_fast_kernel<TM_, TN_, A_, B_, false>), grid, dim3(256), 0, s, ka);            \
  break
#define FASTB(TM_, TN_, A_, B_)                                                                               \
  if (d.b_f16cvt) { if constexpr (!(A_) && !(B_)) hipLaunchKernelGGL((gemm_fast_bfh_kernel<TM_, TN_>), grid, dim3(256), 0, s, ka); } \
  else if (d.a_bf16 && d.b_bf16) hipLaunchKernelGGL((gemm_fast_bf_kernel<TM_, TN_, A_, B_, true>), grid, dim3(256), 0, s, ka);   \
  else if (d.a_bf16) hipLaunchKernelGGL((gemm_fast_bf_kernel<TM_, TN_, A_, B_, false>), grid, dim3(256), 0, s, ka);         \
  else if (gen) hipLaunchKernelGGL((gemm_fast_kernel<TM_, TN_, A_, B_, true>), grid, dim3(256), 0, s, ka);    \
  else hipLaunchKernelGGL((gemm_fast_kernel<TM_, TN_, A_, B_, false>), grid, dim3(256), 0, s, ka);            \
  break
#define GENK(BF_, BK_, LEAN_)                                                                                 \
  if (gen) hipLaunchKernelGGL((gemm_kernel<BF_, BK_, LEAN_, true>), grid, dim3(256), 0, s, ka);               \
  else hipLaunchKernelGGL((gemm_kernel<BF_, BK_, LEAN_, false>), grid, dim3(256), 0, s, ka);                  \
  break
  switch (pl.variant) {
    case 10: FASTK(2, 2, true, true);
    case 11: FASTB(2, 2, true, false);
    case 12: FASTB(2, 2, false, false);
    case 14: FASTK(2, 1, true, true);
    case 15: FASTB(2, 1, true, false);
    case 16: FASTB(2, 1, false, false);
    case 18: FASTK(1, 1, true, true);
    case 19: FASTB(1, 1, true, false);
    case 20: FASTB(1, 1, false, false);
    case 24:   // 128 x 256 tiles, (RC,RC), both operands 16-bit stored (fast_plan: wide)
      if (d.b_f16cvt) hipLaunchKernelGGL((gemm_fast_bfh_kernel<2, 4>), grid, dim3(256), 0, s, ka);
      else hipLaunchKernelGGL((gemm_fast_bf_kernel<2, 4, false, false, true>), grid, dim3(256), 0, s, ka);
      break;
    case 0: GENK(false, 32, false);
    case 3: GENK(true, 128, true);
    case 2: GENK(true, 64, true);
    default: GENK(true, 32, false);
  }
#undef FASTB
#undef GENK
#undef FASTK
  LAUNCH_CHECK();
  return MIMRL_OK;
}

}  // namespace mimrl
