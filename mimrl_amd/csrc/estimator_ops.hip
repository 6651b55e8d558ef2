// Estimator-side kernels (see estimator_ops.h for reference citations).
#include "estimator_ops.h"

namespace mimrl {

namespace {

inline int grid_for(long n, int block = 256, int cap = 4096) {
  long g = (n + block - 1) / block;
  return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

__global__ void copy_rows_kernel(CopyTable t, long n) {
  const int i = blockIdx.y;
  const float* __restrict__ s = t.src[i];
  float* __restrict__ d = t.dst[i];
  for (long j = blockIdx.x * (long)blockDim.x + threadIdx.x; j < n; j += (long)gridDim.x * blockDim.x) d[j] = s[j];
  if (i < 2 && t.z[i].p) {               // zero-fill job i rides on the row-i workgroups
    const ZeroJob z = t.z[i];
    for (long j = blockIdx.x * (long)blockDim.x + threadIdx.x; j < z.chunk * z.rep; j += (long)gridDim.x * blockDim.x)
      z.p[(j / z.chunk) * z.stride + j % z.chunk] = 0.f;
  }
}

__device__ __forceinline__ float softplus_f(float x) { return fmaxf(x, 0.f) + log1pf(__expf(-fabsf(x))); }

// ------------------------------------------------------------------------------------------------
// MI bounds on a [B,B] score matrix; one 1024-thread workgroup per estimator, deterministic reductions.
// ------------------------------------------------------------------------------------------------
// mi[e] = bound value; mil[e] (optional) = the estimator's loss term: -mi for every bound except `mine`, whose loss is
// mean(diag) - mean(exp_nodiag) / ma_et with ma_et = 0.99 + 0.01 mean(exp_nodiag) detached and NOT negated
// (Model.py:121-125).  Bit e of `lossform` says whether estimator e contributes through that loss (stage 1: all five,
// stage 2: f_t, f_a, f_v -- t_a and t_v enter through -mi, Model.py:386) -- it selects the gradient written to dscores.
// S / dS may alias (every gradient entry depends on its own score and on reductions finished before it is written) and
// may live in LDS (generic pointers): the fused separable-critic kernel below runs this body on its on-chip score tile.
// lb / dlb (optional, tuba and interpolate only): log-baseline log a(y_i) per row (VMI.py:72-110) and the gradient of the
// objective with respect to it.  With a baseline, S is modified in place for tuba (S_ij -= lb_i).
__device__ __forceinline__ void mi_bound_body(float* S, float* dS, float* __restrict__ mi, float* __restrict__ mil, float gs, int e, int B,
                              int bound, unsigned lossform, float* red, float* rowstat, const float* __restrict__ lb,
                              float* __restrict__ dlb) {
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, nw = blockDim.x >> 6;
  const float invB = 1.f / B;
  if (lb && bound == BOUND_TUBA) {
    for (long idx = tid; idx < (long)B * B; idx += blockDim.x) S[idx] -= lb[idx / B];
    __syncthreads();
  }

  if (bound == BOUND_INFONCE) {
    // mi = log B + mean_i( s_ii - logsumexp_j s_ij )                       (VMI.py:162-166)
    float part = 0.f;
    for (int i = w; i < B; i += nw) {
      float mx = -INFINITY;
      for (int j = lane; j < B; j += 64) mx = fmaxf(mx, S[(long)i * B + j]);
      mx = wave_max(mx);
      float se = 0.f;
      for (int j = lane; j < B; j += 64) se += __expf(S[(long)i * B + j] - mx);
      se = wave_sum(se);
      const float lse = mx + __logf(se);
      if (lane == 0) { rowstat[i] = lse; part += S[(long)i * B + i] - lse; }
    }
    const float tot = block_sum(part, red);
    if (tid == 0) { mi[e] = __logf((float)B) + tot * invB; if (mil) mil[e] = -mi[e]; }
    if (dS) {
      __syncthreads();
      for (long idx = tid; idx < (long)B * B; idx += blockDim.x) {
        const int i = idx / B, j = idx % B;
        const float pij = __expf(S[idx] - rowstat[i]);
        dS[idx] = gs * invB * ((i == j ? 1.f : 0.f) - pij);
      }
    }
    return;
  }

  if (bound == BOUND_INTERP) {
    // interpolated bound (VMI.py:201-250) with the constant baseline and alpha_logit = 0.01 (Model.py:118):
    //   I_ij = logaddexp(log a + L_ij - log(B-1), log(1-a)),  L_ij = log sum_{k != j} e^{s_ik}   (leave-one-out, per row)
    //   value = 1 + mean_{i != j}(s_jj - I_ij) - mean_{i != j} e^{s_ij - I_jj}
    float* lse = rowstat; float* vj = rowstat + B; float* Rr = rowstat + 2 * B; float* Cj = rowstat + 3 * B;
    float* Ur = rowstat + 4 * B; float* Cm = rowstat + 5 * B;   // baseline gradient: sum_{j != i}(1 - sigma_ij), C_i (1 - sigma_ii)
    const float la = -softplus_f(-0.01f), l1m = -softplus_f(0.01f), lbm1 = __logf((float)B - 1.f);
    const float M2 = (float)B * (B - 1.f);
    // p = e^{s - lse}; its complement 1 - p = -expm1(-d) is what the gradient divides by: computed the reference's way
    // (VMI.py:214-226), a plain 1 - p loses all digits when one score dominates its row
    auto omp_of = [&](float sv, float lsev) { const float d = lsev - sv; return d == 0.f ? 1.f - __expf(-1.f) : -expm1f(-d); };
    auto interp = [&](float sv, float lsev, float lbi, float& p, float& I, float& sig) {
      const float d = lsev - sv;
      const float omp = d == 0.f ? 1.f - __expf(-1.f) : -expm1f(-d);
      p = 1.f - omp;
      const float L = d == 0.f ? sv + 1.f + __logf(omp) : sv + d + __logf(omp);               // safe_d of compute_log_loomean
      const float u = la + L - lbm1, v = l1m + lbi;
      const float hi = fmaxf(u, v);
      I = hi + __logf(__expf(u - hi) + __expf(v - hi));
      sig = __expf(u - I);
    };
    float part_i = 0.f, part_d = 0.f;
    for (int i = w; i < B; i += nw) {
      float mx = -INFINITY;
      for (int j = lane; j < B; j += 64) mx = fmaxf(mx, S[(long)i * B + j]);
      mx = wave_max(mx);
      float se = 0.f;
      for (int j = lane; j < B; j += 64) se += __expf(S[(long)i * B + j] - mx);
      se = wave_sum(se);
      const float lsev = mx + __logf(se);
      float r = 0.f, isum = 0.f, ur = 0.f;
      const float lbi = lb ? lb[i] : 0.f;
      for (int j = lane; j < B; j += 64) {
        float p, I, sig;
        interp(S[(long)i * B + j], lsev, lbi, p, I, sig);
        if (j == i) { vj[i] = I; part_d += S[(long)i * B + j]; }
        else { r += sig / omp_of(S[(long)i * B + j], lsev); isum += I; ur += 1.f - sig; }
      }
      r = wave_sum(r); isum = wave_sum(isum); ur = wave_sum(ur);
      if (lane == 0) { lse[i] = lsev; Rr[i] = r; Ur[i] = ur; part_i += isum; }
    }
    const float sum_i = block_sum(part_i, red);
    const float dsum = block_sum(part_d, red);
    __syncthreads();
    float mxc = -INFINITY;
    for (long idx = tid; idx < (long)B * B; idx += blockDim.x) {
      const int i = idx / B, j = idx % B;
      if (i != j) mxc = fmaxf(mxc, S[idx] - vj[j]);
    }
    mxc = block_max(mxc, red);
    float csum = 0.f;
    for (int j = tid; j < B; j += blockDim.x) {
      float c = 0.f;
      for (int i = 0; i < B; ++i)
        if (i != j) c += __expf(S[(long)i * B + j] - vj[j] - mxc);
      csum += c;
      float p, I, sig;
      interp(S[(long)j * B + j], lse[j], lb ? lb[j] : 0.f, p, I, sig);
      Cj[j] = c * __expf(mxc) * sig / omp_of(S[(long)j * B + j], lse[j]);        // Q_j = C_j sigma_jj / (1 - p_jj)
      Cm[j] = c * __expf(mxc) * (1.f - sig);
    }
    csum = block_sum(csum, red);
    const float emx = __expf(mxc);
    const float marg = emx * csum / M2;
    const float val = 1.f + ((B - 1.f) * dsum - sum_i) / M2 - marg;
    if (tid == 0) { mi[e] = val; if (mil) mil[e] = -val; }
    if (!dS) return;
    __syncthreads();
    if (dlb)
      for (int i = tid; i < B; i += blockDim.x) dlb[i] = gs * (Cm[i] - Ur[i]) / M2;
    for (long idx = tid; idx < (long)B * B; idx += blockDim.x) {
      const int a = idx / B, b = idx % B;
      float p, I, sig;
      interp(S[idx], lse[a], lb ? lb[a] : 0.f, p, I, sig);
      float g;
      if (a == b) g = invB - p * Rr[a] / M2;
      else g = -p * (Rr[a] - sig / omp_of(S[idx], lse[a])) / M2 - (emx * __expf(S[idx] - vj[b] - mxc) - Cj[a] * p) / M2;
      dS[idx] = gs * g;
    }
    return;
  }

  // ---- bounds built from diag mean / off-diagonal log-mean-exp / softplus sums      (VMI.py:121-198)
  const float shift = (bound == BOUND_NWJ || bound == BOUND_JS) ? 1.f : 0.f;
  const bool clipped = bound == BOUND_SMILE;
  auto tr = [&](float v) { v -= shift; return clipped ? fminf(fmaxf(v, -1.f), 1.f) : v; };
  float mx = -INFINITY;
  for (long idx = tid; idx < (long)B * B; idx += blockDim.x) {
    const int i = idx / B, j = idx % B;
    if (i != j) mx = fmaxf(mx, tr(S[idx]));
  }
  mx = block_max(mx, red);
  float se = 0.f, dsum = 0.f, sp_all = 0.f, sp_diag = 0.f, nsp_diag = 0.f;
  for (long idx = tid; idx < (long)B * B; idx += blockDim.x) {
    const int i = idx / B, j = idx % B;
    const float v = S[idx];
    sp_all += softplus_f(v);
    if (i == j) { dsum += v; sp_diag += softplus_f(v); nsp_diag += -softplus_f(-v); }
    else se += __expf(tr(v) - mx);
  }
  se = block_sum(se, red);
  dsum = block_sum(dsum, red);
  sp_all = block_sum(sp_all, red);
  sp_diag = block_sum(sp_diag, red);
  nsp_diag = block_sum(nsp_diag, red);
  const float M = (float)B * (B - 1.f);
  const float lme = mx + __logf(se) - __logf(M);             // logmeanexp_nodiag
  const float dmean = dsum * invB;
  const float js = nsp_diag * invB - (sp_all - sp_diag) / M;  // js_fgan_lower_bound
  float val;
  switch (bound) {
    case BOUND_TUBA: val = 1.f + dmean - __expf(lme); break;
    case BOUND_NWJ: val = 1.f + (dmean - 1.f) - __expf(lme); break;
    case BOUND_DV: val = dmean - lme; break;
    case BOUND_JS_FGAN: val = js; break;
    case BOUND_JS: val = 1.f + (dmean - 1.f) - __expf(lme); break;        // value of nwj, gradient of js
    default: val = dmean - lme; break;                                    // SMILE: value dv(clamped lme), gradient js; MINE: dv value
  }
  const bool mine_loss = bound == BOUND_MINE && ((lossform >> e) & 1u);
  const float mean_et = __expf(mx) * se / ((float)B * B);                 // mean over all B^2 entries, diagonal = 0 (VMI.py:128-133)
  const float ma_et = 0.99f + 0.01f * mean_et;
  if (tid == 0) { mi[e] = val; if (mil) mil[e] = mine_loss ? dmean - mean_et / ma_et : -val; }
  if (!dS) return;
  if (mine_loss) {   // gs = -coefficient (d total / d mi for a "-mi" loss); here the loss enters with +coefficient
    const float c = -gs;
    for (long idx = tid; idx < (long)B * B; idx += blockDim.x) {
      const int i = idx / B, j = idx % B;
      dS[idx] = c * ((i == j) ? invB : -__expf(S[idx]) / (ma_et * (float)B * B));
    }
    return;
  }
  for (long idx = tid; idx < (long)B * B; idx += blockDim.x) {
    const int i = idx / B, j = idx % B;
    const float v = S[idx];
    float g;
    if (bound == BOUND_TUBA || bound == BOUND_NWJ) g = (i == j) ? invB : -__expf(v - shift) / M;
    else if (bound == BOUND_DV || bound == BOUND_MINE) g = (i == j) ? invB : -__expf(v - mx) / se;
    else g = (i == j) ? sigmoid_f(-v) * invB : -sigmoid_f(v) / M;          // js_fgan / js / smile
    dS[idx] = gs * g;
  }
  if (dlb && bound == BOUND_TUBA) {   // S_ij entered as s_ij - lb_i
    __syncthreads();
    for (int i = tid; i < B; i += blockDim.x) {
      float t = 0.f;
      for (int j = 0; j < B; ++j) t += dS[(long)i * B + j];
      dlb[i] = -t;
    }
  }
}

__global__ __launch_bounds__(1024) void mi_bound_kernel(const float* scores, float* dscores,
                                                        float* __restrict__ mi, float* __restrict__ mil,
                                                        const float* __restrict__ gscale, int B, int bound, unsigned lossform,
                                                        const float* __restrict__ lb, float* __restrict__ dlb, long lb_stride) {
  __shared__ float red[16];
  __shared__ float rowstat[6 * 1024];   // per-row / per-column statistics (InfoNCE: lse; interpolate: 6 vectors); B <= 1024
  const int e = blockIdx.x;
  mi_bound_body(const_cast<float*>(scores) + (long)e * B * B, dscores ? dscores + (long)e * B * B : nullptr, mi, mil,
                gscale ? gscale[e] : 0.f, e, B, bound, lossform, red, rowstat, lb ? lb + e * lb_stride : nullptr,
                dlb ? dlb + e * lb_stride : nullptr);
}

// InfoNCE on score matrices that live in memory (the concat critic; round 5): the bound is ROW-wise -- mi = log B + mean_i(s_ii - lse_i),
// dS_ij = gs / B ((i == j) - e^{s_ij - lse_i}) (VMI.py:162-166) -- so one wave per row, 16 rows per workgroup, B / 16 workgroups per
// estimator instead of ONE: mi_bound_kernel walked each 256 x 256 matrix of cfg3 with a single workgroup (5 CUs busy, 68 + 49 us on the
// chain of the two stages).  The row sums of a workgroup go to a slot, the last workgroup of an estimator (ticket) adds the slots in order:
// the value does not depend on arrival order.  B <= 1024 (a row in registers: 16 values per lane).
// Ticket counters and slots live in a CALLER-owned workspace (NceWs: zeroed once, self-resetting) -- per engine handle, so that two handles
// on different streams (or an op-level call beside a step) never share a ticket (round 5 kept them in process-global arrays).
constexpr int NCE_MAX_EST = 16;
struct NceWs { unsigned ticket[NCE_MAX_EST]; float part[NCE_MAX_EST][64]; };
static_assert(sizeof(NceWs) <= NCE_WS_FLOATS * sizeof(float), "NCE_WS_FLOATS (estimator_ops.h) too small");
__global__ __launch_bounds__(1024) void mi_infonce_rows_kernel(const float* __restrict__ scores, float* __restrict__ dscores, float* __restrict__ mi,
                                                               float* __restrict__ mil, const float* __restrict__ gscale, int B, NceWs* __restrict__ ws) {
  __shared__ float red[16];
  __shared__ int last;
  const int e = blockIdx.y, blk = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int i = blk * 16 + w;
  const float invB = 1.f / B, gs = gscale ? gscale[e] : 0.f;
  float part = 0.f;
  if (i < B) {                                                     // (wave-uniform)
    const float* __restrict__ row = scores + ((long)e * B + i) * B;
    float v[16], mx = -INFINITY;
#pragma unroll
    for (int q = 0; q < 16; ++q) { const int j = lane + 64 * q; v[q] = j < B ? row[j] : -INFINITY; mx = fmaxf(mx, v[q]); }
    const float sii = row[i];
    mx = wave_max(mx);
    float se = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) se += __expf(v[q] - mx);           // (padding: e^{-inf} = 0)
    se = wave_sum(se);
    const float lse = mx + __logf(se);
    if (lane == 0) part = sii - lse;
    if (dscores) {
      float* __restrict__ drow = dscores + ((long)e * B + i) * B;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int j = lane + 64 * q;
        if (j < B) drow[j] = gs * invB * ((i == j ? 1.f : 0.f) - __expf(v[q] - lse));
      }
    }
  }
  const float tot = block_sum(part, red);
  if (tid == 0) {
    ws->part[e][blk] = tot;
    __threadfence();
    last = atomicAdd(&ws->ticket[e], 1u) == gridDim.x - 1 ? 1 : 0;
  }
  __syncthreads();
  if (last && tid == 0) {
    __threadfence();
    float s = 0.f;
    for (unsigned b = 0; b < gridDim.x; ++b) s += __hip_atomic_load(&ws->part[e][b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    mi[e] = __logf((float)B) + s * invB;
    if (mil) mil[e] = -mi[e];
    ws->ticket[e] = 0u;
  }
}

// Separable critic, one estimator per workgroup, everything between the tower outputs and their gradients on chip:
//   scores = h(y) g(x)^T (VMI.py:55-57) -> bound (+ d/dscores, in place in LDS) -> d h = dS g,  d g = dS^T h.
// 16 waves: one 32x32 MFMA tile each (B <= 128, B % 32 == 0); bf16 operands, fp32 accumulate.
__global__ __launch_bounds__(1024) void mi_sep_fused_kernel(const float* __restrict__ tout, float* __restrict__ dtout,
                                                            float* __restrict__ mi, float* __restrict__ mil,
                                                            const float* __restrict__ gscale, int B, int bound,
                                                            unsigned lossform, int do_bwd, const float* __restrict__ lb,
                                                            float* __restrict__ dlb, long lb_stride, int dbg) {
  extern __shared__ float S[];      // [B][B] scores / their gradient, then the bf16 transpose of the gradient
  __shared__ float red[16];
  __shared__ float rowstat[6 * 128];
  const int e = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 31, lh = lane >> 5;
  const int nt = B / 32;
  const float* __restrict__ X = tout + (long)(2 * e) * B * 128;       // g(x)
  const float* __restrict__ Y = tout + (long)(2 * e + 1) * B * 128;   // h(y)
  auto frag_rows = [&](const float* __restrict__ M, int row, int ks) {   // 8 consecutive k of one row -> bf16x8
    const float4 a = *reinterpret_cast<const float4*>(M + (long)row * 128 + ks * 16 + 8 * lh);
    const float4 b = *reinterpret_cast<const float4*>(M + (long)row * 128 + ks * 16 + 8 * lh + 4);
    bf16x8 p;
    p[0] = to_bf16(a.x); p[1] = to_bf16(a.y); p[2] = to_bf16(a.z); p[3] = to_bf16(a.w);
    p[4] = to_bf16(b.x); p[5] = to_bf16(b.y); p[6] = to_bf16(b.z); p[7] = to_bf16(b.w);
    return p;
  };
  if (wave < nt * nt) {
    const int ti = wave / nt, tj = wave % nt;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks)
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows(Y, ti * 32 + lr, ks), frag_rows(X, tj * 32 + lr, ks), acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; ++r) S[(ti * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * B + tj * 32 + lr] = acc[r];
  }
  __syncthreads();
  if (dbg == 1) return;
  mi_bound_body(S, do_bwd ? S : nullptr, mi, mil, gscale ? gscale[e] : 0.f, e, B, bound, lossform, red, rowstat,
                lb ? lb + e * lb_stride : nullptr, (do_bwd && dlb) ? dlb + e * lb_stride : nullptr);
  if (!do_bwd || dbg == 2) return;
  __syncthreads();
  // The A fragment of d h = dS g wants, per lane, 8 consecutive j of ITS row i: with the rows 128 words apart all 32 lanes
  // hit one LDS bank (32-way conflict on every read, and 16 waves share the LDS).  A transposed bf16 copy dST[j][i]
  // (pitch B + 2: conflict-free to write and to read) turns them into the same lane-contiguous reads as d g = dS^T h.
  __bf16* dST = reinterpret_cast<__bf16*>(S + B * B);
  const int SP = B + 2;
  for (int idx = tid; idx < B * B; idx += 1024) {
    const int i = idx / B, j = idx - i * B;
    dST[j * SP + i] = to_bf16(S[idx]);
  }
  __syncthreads();
  // d h[i][n] = sum_j dS[i][j] g[j][n]   and   d g[j][n] = sum_i dS[i][j] h[i][n]:  nt x 4 tiles each, wave -> (row tile, n tile)
  if (wave < nt * 4) {
    const int tr = wave >> 2, tn = wave & 3;
    f32x16 ah, ag;
#pragma unroll
    for (int r = 0; r < 16; ++r) { ah[r] = 0.f; ag[r] = 0.f; }
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {         // B <= 128; fully unrolled so that all operand loads are in flight together
      if (ks * 16 >= B) break;
      const int k0 = ks * 16 + 8 * lh;
      bf16x8 a1, a2, b1, b2;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        a1[q] = dST[(k0 + q) * SP + tr * 32 + lr];                // dS[i][j..]
        a2[q] = to_bf16(S[(k0 + q) * B + tr * 32 + lr]);          // dS[i..][j]
        b1[q] = to_bf16(X[(long)(k0 + q) * 128 + tn * 32 + lr]);
        b2[q] = to_bf16(Y[(long)(k0 + q) * 128 + tn * 32 + lr]);
      }
      ah = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, ah, 0, 0, 0);
      ag = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b2, ag, 0, 0, 0);
    }
    float* __restrict__ dX = dtout + (long)(2 * e) * B * 128;
    float* __restrict__ dY = dtout + (long)(2 * e + 1) * B * 128;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = tr * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      dY[(long)row * 128 + tn * 32 + lr] = ah[r];
      dX[(long)row * 128 + tn * 32 + lr] = ag[r];
    }
  }
}

// Separable critic + InfoNCE, row-tiled: workgroup = (32 rows i of the score matrix, estimator).  InfoNCE is a sum of per-row terms
// (s_ii - logsumexp_j s_ij), so a row tile needs h(y) for its 32 rows and g(x) for all B columns and nothing from the other tiles:
//   scores tile = h_tile g^T -> row log-sum-exps -> value (atomic partial) -> dS tile -> d h_tile = dS g (owned rows: plain stores),
//   d g += dS^T h_tile (partial over this tile's rows: float atomics into the caller-zeroed gradient).
// 4 waves; g(x) and the h(y) tile are staged once into LDS as bf16 (coalesced 16-byte loads), every fragment comes from there.
// The one-workgroup-per-estimator kernel above is 5 workgroups of serial phases (37 us at B = 128); this one is 5 x B/32.
constexpr int NXP = 128 + 8;       // bf16 pitch of a staged [.][128] operand (272 B: conflict-free 16-byte fragment reads)
#ifdef MIMRL_PHASE_PROBE
__device__ long long g_nce_phase[8];
#define NPHASE(i) do { if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) g_nce_phase[i] = (long long)wall_clock64(); } while (0)
#else
#define NPHASE(i) do { } while (0)
#endif
__global__ __launch_bounds__(256) void mi_sep_nce_kernel(const float* __restrict__ tout, float* __restrict__ dtout,
                                                         float* __restrict__ mi, float* __restrict__ mil,
                                                         const float* __restrict__ gscale, int B, int do_bwd) {
  extern __shared__ __attribute__((aligned(16))) char nce_smem[];
  __shared__ float red[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 31, lh = lane >> 5;
  const int ti = blockIdx.x, e = blockIdx.y, i0 = ti * 32, nt = B / 32, SP = B + 1;
  __bf16* Xs = reinterpret_cast<__bf16*>(nce_smem);                       // [B][NXP]   g(x), all rows
  __bf16* Ys = Xs + (size_t)B * NXP;                                  // [32][NXP]  h(y), this tile's rows
  float* S = reinterpret_cast<float*>(Ys + 32 * NXP);                 // [32][SP]   scores, then their gradient
  const float* __restrict__ X = tout + (long)(2 * e) * B * 128;
  const float* __restrict__ Y = tout + (long)(2 * e + 1) * B * 128 + (long)i0 * 128;
  NPHASE(0);
  // ---- stage (all loads of a batch in flight together)
  {
    float4 q[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) q[j] = reinterpret_cast<const float4*>(Y)[tid + 256 * j];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int idx = tid + 256 * j, row = idx >> 5, c4 = (idx & 31) * 4;
      bf16x4 p; p[0] = to_bf16(q[j].x); p[1] = to_bf16(q[j].y); p[2] = to_bf16(q[j].z); p[3] = to_bf16(q[j].w);
      *reinterpret_cast<bf16x4*>(Ys + row * NXP + c4) = p;
    }
    for (int base = 0; base < B * 32; base += 256 * 8) {
      float4 x[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) { const int idx = base + tid + 256 * j; x[j] = reinterpret_cast<const float4*>(X)[idx < B * 32 ? idx : 0]; }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int idx = base + tid + 256 * j, row = idx >> 5, c4 = (idx & 31) * 4;
        if (idx < B * 32) {
          bf16x4 p; p[0] = to_bf16(x[j].x); p[1] = to_bf16(x[j].y); p[2] = to_bf16(x[j].z); p[3] = to_bf16(x[j].w);
          *reinterpret_cast<bf16x4*>(Xs + row * NXP + c4) = p;
        }
      }
    }
  }
  __syncthreads();
  NPHASE(1);
  // ---- scores tile [32 x B]: wave -> column tiles tj = wave, wave + 4, ...
  for (int tj = wave; tj < nt; tj += 4) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const bf16x8 a = *reinterpret_cast<const bf16x8*>(Ys + lr * NXP + ks * 16 + 8 * lh);
      const bf16x8 b = *reinterpret_cast<const bf16x8*>(Xs + (tj * 32 + lr) * NXP + ks * 16 + 8 * lh);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) S[((r & 3) + 8 * (r >> 2) + 4 * lh) * SP + tj * 32 + lr] = acc[r];
  }
  __syncthreads();
  NPHASE(2);
  // ---- row log-sum-exps, this tile's share of  mi = log B + mean_i(s_ii - lse_i)  (VMI.py:162-166), and dS = gs / B * (I - softmax rows)
  // in place.  Round 3b (tools/cube_phase.py: 4.1 + 2.0 of 12.1 us): a wave's 8 rows are worked on TOGETHER -- eight independent
  // max / sum reduction chains the scheduler can interleave instead of eight dependent ones in a row -- and the exponentials of the
  // sum are kept and scaled into the gradient (one exp per element and no separate pass behind a barrier)
  const float gsb = (gscale ? gscale[e] : 0.f) / B;
  float part = 0.f;
  {
    float sv[8][2], mx[8], se[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int i = wave + 4 * q;
#pragma unroll
      for (int h = 0; h < 2; ++h) { const int j = lane + 64 * h; sv[q][h] = j < B ? S[i * SP + j] : -INFINITY; }
      mx[q] = fmaxf(sv[q][0], sv[q][1]);
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) mx[q] = wave_max(mx[q]);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
#pragma unroll
      for (int h = 0; h < 2; ++h) sv[q][h] = lane + 64 * h < B ? __expf(sv[q][h] - mx[q]) : 0.f;
      se[q] = sv[q][0] + sv[q][1];
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) se[q] = wave_sum(se[q]);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int i = wave + 4 * q;
      const float lse = mx[q] + __logf(se[q]);
      if (lane == 0) part += S[i * SP + i0 + i] - lse;          // (read before this row is overwritten below: same lane order)
      if (do_bwd) {
        const float inv = 1.f / se[q];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int j = lane + 64 * h;
          if (j < B) S[i * SP + j] = gsb * ((j == i0 + i ? 1.f : 0.f) - sv[q][h] * inv);
        }
      }
    }
  }
  if (lane == 0) red[wave] = part;
  __syncthreads();
  NPHASE(3);
  if (tid == 0) {
    const float v = (red[0] + red[1] + red[2] + red[3]) / B + (ti == 0 ? __logf((float)B) : 0.f);
    acc_add(&mi[e], v);
    if (mil) acc_add(&mil[e], -v);
  }
  if (!do_bwd) return;
  NPHASE(4);
  float* __restrict__ dX = dtout + (long)(2 * e) * B * 128;
  float* __restrict__ dY = dtout + (long)(2 * e + 1) * B * 128 + (long)i0 * 128;
  const int tn = wave;                       // this wave's 32 feature columns in both products
  {   // d h_tile [32 x 128] = dS [32 x B] . g [B x 128]
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int ks = 0; ks * 16 < B; ++ks) {
      const int k0 = ks * 16 + 8 * lh;
      bf16x8 a, b;
#pragma unroll
      for (int q = 0; q < 8; ++q) { a[q] = to_bf16(S[lr * SP + k0 + q]); b[q] = Xs[(k0 + q) * NXP + tn * 32 + lr]; }
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) dY[(long)((r & 3) + 8 * (r >> 2) + 4 * lh) * 128 + tn * 32 + lr] = acc[r];
  }
  NPHASE(5);
  // d g [B x 128] += dS^T [B x 32] . h_tile [32 x 128]   (reduction over this tile's 32 rows: two k-steps)
  for (int tr = 0; tr < nt; ++tr) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int k0 = ks * 16 + 8 * lh;
      bf16x8 a, b;
#pragma unroll
      for (int q = 0; q < 8; ++q) { a[q] = to_bf16(S[(k0 + q) * SP + tr * 32 + lr]); b[q] = Ys[(k0 + q) * NXP + tn * 32 + lr]; }
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) acc_add(&dX[(long)(tr * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * 128 + tn * 32 + lr], acc[r]);
  }
  NPHASE(6);
}
#ifdef MIMRL_PHASE_PROBE
}  // namespace
int nce_read_phases(long long* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_nce_phase), sizeof(long long) * 8) == hipSuccess ? 0 : 1; }
namespace {
#endif

// ------------------------------------------------------------------------------------------------
// concat critic, first layer in its separable form
// ------------------------------------------------------------------------------------------------
__global__ void pair_expand_fwd_kernel(const float* __restrict__ P, const float* __restrict__ Q, float* __restrict__ a1,
                                       int B, int Hd) {
  const int i = blockIdx.x, e = blockIdx.y;
  const float* p = P + ((long)e * B + i) * Hd;
  const float* q = Q + (long)e * B * Hd;
  float* out = a1 + ((long)e * B + i) * B * Hd;
  for (long idx = threadIdx.x; idx < (long)B * Hd; idx += blockDim.x) {
    const float v = p[idx % Hd] + q[idx];
    out[idx] = v > 0.f ? v : 0.f;
  }
}
// du1 = da1 * (a1 > 0) in place;  dP[i,c] = sum_j du1[i,j,c]
__global__ void pair_reduce_p_kernel(const float* __restrict__ a1, float* __restrict__ da1, float* __restrict__ dP, int B,
                                     int Hd) {
  const int i = blockIdx.x, e = blockIdx.y;
  const float* a = a1 + ((long)e * B + i) * B * Hd;
  float* g = da1 + ((long)e * B + i) * B * Hd;
  for (int c = threadIdx.x; c < Hd; c += blockDim.x) {
    float s = 0.f;
    for (int j = 0; j < B; ++j) {
      const long o = (long)j * Hd + c;
      const float v = a[o] > 0.f ? g[o] : 0.f;
      g[o] = v;
      s += v;
    }
    dP[((long)e * B + i) * Hd + c] = s;
  }
}
// dQ[e][j][:] = sum_i du1[e][i][j][:]  (rows of one j are B * Hd floats apart).  Round 4: 16-byte loads, four row groups per workgroup
// with four independent accumulators each (16 rows in flight per workgroup instead of one 4-byte load per thread and iteration:
// 205 us for 335 MB at cfg3, 1.6 TB/s), the groups combined through LDS in a fixed order.
__global__ __launch_bounds__(256) void pair_reduce_q_kernel(const float* __restrict__ du1, float* __restrict__ dQ, int B, int Hd) {
  const int j = blockIdx.x, e = blockIdx.y;
  const float* g = du1 + (long)e * B * B * Hd + (long)j * Hd;
  const int lpr = Hd >> 2;                                     // lanes per row (float4 each)
  if ((Hd & 3) == 0 && lpr <= 256 && 256 % lpr == 0 && blockDim.x == 256) {
    __shared__ float4 red[256];
    const int groups = 256 / lpr, grp = threadIdx.x / lpr, c4 = threadIdx.x - grp * lpr;
    const long rs = (long)B * Hd;
    float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0, a3 = a0;
    const float4 z = a0;
    for (int i = grp; i < B; i += 4 * groups) {
      const int i1 = i + groups, i2 = i + 2 * groups, i3 = i + 3 * groups;
      const float4 v0 = *reinterpret_cast<const float4*>(g + (long)i * rs + 4 * c4);
      const float4 v1 = i1 < B ? *reinterpret_cast<const float4*>(g + (long)i1 * rs + 4 * c4) : z;
      const float4 v2 = i2 < B ? *reinterpret_cast<const float4*>(g + (long)i2 * rs + 4 * c4) : z;
      const float4 v3 = i3 < B ? *reinterpret_cast<const float4*>(g + (long)i3 * rs + 4 * c4) : z;
      a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;
      a1.x += v1.x; a1.y += v1.y; a1.z += v1.z; a1.w += v1.w;
      a2.x += v2.x; a2.y += v2.y; a2.z += v2.z; a2.w += v2.w;
      a3.x += v3.x; a3.y += v3.y; a3.z += v3.z; a3.w += v3.w;
    }
    a0.x += a1.x + (a2.x + a3.x); a0.y += a1.y + (a2.y + a3.y); a0.z += a1.z + (a2.z + a3.z); a0.w += a1.w + (a2.w + a3.w);
    red[threadIdx.x] = a0;
    __syncthreads();
    if (grp == 0) {
      for (int q = 1; q < groups; ++q) { const float4 r = red[q * lpr + c4]; a0.x += r.x; a0.y += r.y; a0.z += r.z; a0.w += r.w; }
      *reinterpret_cast<float4*>(dQ + ((long)e * B + j) * Hd + 4 * c4) = a0;
    }
    return;
  }
  for (int c = threadIdx.x; c < Hd; c += blockDim.x) {
    float s = 0.f;
    for (int i = 0; i < B; ++i) s += g[(long)i * B * Hd + c];
    dQ[((long)e * B + j) * Hd + c] = s;
  }
}
__global__ void relu_bwd_kernel(const float* __restrict__ a, float* __restrict__ g, long n) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    if (!(a[i] > 0.f)) g[i] = 0.f;
}

// ------------------------------------------------------------------------------------------------
__global__ void cmi_assemble_kernel(CmiAssembleArgs a) {
  const int row = blockIdx.x, c = blockIdx.y;
  float* out = a.out + ((long)c * 2 * a.n + row) * 384;
  for (int col = threadIdx.x; col < 384; col += blockDim.x) {
    const int part = col >> 7, j = col & 127;
    const CmiOperand& op = a.op[c][part];
    float v;
    if (row < a.n) {
      v = op.is_label ? op.cur[row] : op.cur[(long)row * 128 + j];
    } else {
      const int jj = row - a.n;
      const int src = part == 0 ? a.idx_x[(long)c * a.n + jj] : a.anchors[(long)c * a.m + jj / a.k];
      v = op.is_label ? op.bank[src] : op.bank[(long)src * 128 + j];
    }
    out[col] = v;
  }
}

// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cmi_loss_kernel(const float* __restrict__ logits, float* __restrict__ dlogits,
                                                       float* __restrict__ bce, float* __restrict__ cmi,
                                                       const float* __restrict__ g_bce, const float* __restrict__ g_cmi,
                                                       int n, int hardtanh) {
  __shared__ float red[16];
  const int e = blockIdx.x;
  const float* L = logits + (long)e * 2 * n * 2;
  float* dL = dlogits ? dlogits + (long)e * 2 * n * 2 : nullptr;
  const float gb = g_bce ? g_bce[e] : 0.f, gc = g_cmi ? g_cmi[e] : 0.f;
  float sb = 0.f, s1 = 0.f, s2 = 0.f;
  const float inv4n = 1.f / (4.f * n), inv2n = 1.f / (2.f * n);
  for (int r = threadIdx.x; r < 2 * n; r += blockDim.x) {
    const bool joint = r < n;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const float raw = L[r * 2 + c];
      const float o = fminf(fmaxf(raw, -10.f), 10.f);                  // Model.py:69
      const bool pass = raw >= -10.f && raw <= 10.f;
      float gam, dgo;
      if (hardtanh) { gam = fminf(fmaxf(o, 1e-4f), 1.f - 1e-4f); dgo = (o > 1e-4f && o < 1.f - 1e-4f) ? 1.f : 0.f; }
      else { gam = sigmoid_f(o); dgo = gam * (1.f - gam); }
      const float t = (joint ? (c == 0) : (c == 1)) ? 1.f : 0.f;       // Model.py:176-178
      const float lg = fmaxf(__logf(gam), -100.f), l1g = fmaxf(__logf(1.f - gam), -100.f);
      sb += -(t * lg + (1.f - t) * l1g);
      float dgam = gb * (-(t / gam - (1.f - t) / (1.f - gam)) * inv4n);
      if (c == 0) {
        const float lr = __logf(gam / (1.f - gam + 1e-6f));            // Model.py:215-216
        if (joint) s1 += lr; else s2 += lr;
        dgam += gc * (joint ? inv2n : -inv2n) * (1.f / gam + 1.f / (1.f - gam + 1e-6f));
      }
      if (dL) dL[r * 2 + c] = pass ? dgam * dgo : 0.f;
    }
  }
  sb = block_sum(sb, red);
  s1 = block_sum(s1, red);
  s2 = block_sum(s2, red);
  if (threadIdx.x == 0) {
    bce[e] = sb * inv4n;
    cmi[e] = 1.f + s1 * inv2n - s2 * inv2n;                            // Model.py:219 (divisor = stacked batch 2n)
  }
}

__global__ void gather_sum_kernel(float* __restrict__ dst, GatherSum g, int B, int D, int accumulate) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < (long)B * D; i += (long)gridDim.x * blockDim.x) {
    const int b = i / D, d = i % D;
    float s = accumulate ? dst[i] : 0.f;
    for (int q = 0; q < g.n; ++q)
      if (b < g.rows[q]) s += g.src[q][(long)b * g.ld[q] + g.off[q] + d];
    dst[i] = s;
  }
}

__global__ void gather_sum4_kernel(GatherSum4 a, int B, int D, int first) {
  const GatherSum& g = a.g[blockIdx.y + first];
  float* __restrict__ dst = a.dst[blockIdx.y + first];
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < (long)B * D; i += (long)gridDim.x * blockDim.x) {
    const int b = i / D, d = i % D;
    float s = 0.f;
    for (int q = 0; q < g.n; ++q)
      if (b < g.rows[q]) s += g.src[q][(long)b * g.ld[q] + g.off[q] + d];
    dst[i] = s;
  }
}

// one workgroup = 256 rows of one group; a thread owns 4 consecutive columns (float4) of every 4th..(256/ (din/4))-th row
__global__ __launch_bounds__(256) void top1_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ W, const float* __restrict__ act,
                                                       float* __restrict__ dz, float* __restrict__ dW, float* __restrict__ db_top,
                                                       float* __restrict__ db_below, int rows, int brows, int din, long pstride) {
  __shared__ float red[3][256 * 4];
  const int g = blockIdx.y, r0 = blockIdx.x * 256;
  const int tpr = din >> 2;                    // threads per row
  const int rpp = 256 / tpr;                   // rows per pass
  const int c4 = (threadIdx.x % tpr) * 4, rr = threadIdx.x / tpr;
  const float4 w = *reinterpret_cast<const float4*>(W + (long)g * pstride + c4);
  float4 sw = make_float4(0.f, 0.f, 0.f, 0.f), sb = make_float4(0.f, 0.f, 0.f, 0.f);
  float st = 0.f;
  const int rend = min(r0 + 256, rows);
  for (int r = r0 + rr; r < rend; r += rpp) {
    const long o = ((long)g * brows + r) * din + c4;
    const float d = dout[(long)g * brows + r];
    const float4 a = *reinterpret_cast<const float4*>(act + o);
    float4 z;
    z.x = a.x > 0.f ? d * w.x : 0.f; z.y = a.y > 0.f ? d * w.y : 0.f; z.z = a.z > 0.f ? d * w.z : 0.f; z.w = a.w > 0.f ? d * w.w : 0.f;
    *reinterpret_cast<float4*>(dz + o) = z;
    sw.x += d * a.x; sw.y += d * a.y; sw.z += d * a.z; sw.w += d * a.w;
    sb.x += z.x; sb.y += z.y; sb.z += z.z; sb.w += z.w;
    if (c4 == 0) st += d;
  }
  if (!dW && !db_below && !db_top) return;
  float* mine = &red[0][threadIdx.x * 4];
  mine[0] = sw.x; mine[1] = sw.y; mine[2] = sw.z; mine[3] = sw.w;
  mine = &red[1][threadIdx.x * 4];
  mine[0] = sb.x; mine[1] = sb.y; mine[2] = sb.z; mine[3] = sb.w;
  red[2][threadIdx.x] = st;
  __syncthreads();
  if (threadIdx.x < din) {                      // column c: sum over the rpp row slots
    const int c = threadIdx.x, slot = c >> 2, sub = c & 3;
    float a1 = 0.f, a2 = 0.f;
    for (int q = 0; q < rpp; ++q) { a1 += red[0][(q * tpr + slot) * 4 + sub]; a2 += red[1][(q * tpr + slot) * 4 + sub]; }
    if (dW) acc_add(dW + (long)g * pstride + c, a1);
    if (db_below) acc_add(db_below + (long)g * pstride + c, a2);
  }
  if (threadIdx.x == 0 && db_top) {
    float t = 0.f;
    for (int q = 0; q < rpp; ++q) t += red[2][q * tpr];
    acc_add(db_top + (long)g * pstride, t);
  }
}

__global__ void adam_kernel(AdamArgs a) {
  const int t = *a.step;
  const float lr = *a.lr;
  const float bc1 = 1.f - powf(a.beta1, (float)t), bc2 = 1.f - powf(a.beta2, (float)t);
  const float step_size = lr / bc1, inv_sqrt_bc2 = 1.f / sqrtf(bc2);
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < a.n; i += (long)gridDim.x * blockDim.x) {
    float g = a.g[i];
    if (i >= a.fold.lo_all && i < a.fold.hi_all) {
      for (int q = 0; q < a.fold.n; ++q)
        if (i >= a.fold.lo[q] && i < a.fold.hi[q]) {
          const long j = i - a.fold.lo[q];
          const int c = (int)(j % a.fold.d[q]);
          float* src = a.fold.src[q] + (j / a.fold.d[q]) * a.fold.ld[q] + c;
          g += *src;
          *src = 0.f;
          // the zero-padded columns [d, ld) of a packed row receive atomic adds too (products of zero operands: 0, or NaN once an
          // upstream gradient is non-finite) and belong to no parameter: the thread of the row's last real column resets them, so
          // the WHOLE packed buffer is zero again after the update, as l0_unpack_kernel left it
          if (c == a.fold.d[q] - 1)
            for (int e = a.fold.d[q]; e < a.fold.ld[q]; ++e) src[e - c] = 0.f;
        }
    }
    g *= a.gscale;
    if (a.clip > 0.f) g = fminf(fmaxf(g, -a.clip), a.clip);             // clip_grad_value_ (Solver.py:211-212)
    const float p = a.p[i];
    if (a.weight_decay != 0.f) g += a.weight_decay * p;
    const float m = a.beta1 * a.m[i] + (1.f - a.beta1) * g;
    const float v = a.beta2 * a.v[i] + (1.f - a.beta2) * g * g;
    a.m[i] = m; a.v[i] = v;
    const float pn = p - step_size * m / (sqrtf(v) * inv_sqrt_bc2 + a.eps);
    a.p[i] = pn;
    if (a.pimg) a.pimg[i] = to_bf16(pn);
    a.g[i] = 0.f;   // leave the bucket zeroed for the next accumulation pass (saves a memset per stage)
  }
}

// The same update, 8 consecutive elements per thread (two 16-byte pieces per array), for buckets without parked pieces (AdamArgs::Frag,
// ::sb): same per-element arithmetic as adam_kernel, so parameters and moments are bit-identical to it.
__global__ __launch_bounds__(256) void adam8_kernel(AdamArgs a) {
  __shared__ float red[16];
  if (a.sb_on && blockIdx.x == 0) stage_boundary_body(a.sb, red);
  const int t = *a.step;
  const float lr = *a.lr;
  const float bc1 = 1.f - powf(a.beta1, (float)t), bc2 = 1.f - powf(a.beta2, (float)t);
  const float step_size = lr / bc1, inv_sqrt_bc2 = 1.f / sqrtf(bc2);
  const long n8 = a.n >> 3;
  for (long i8 = blockIdx.x * (long)blockDim.x + threadIdx.x; i8 < n8; i8 += (long)gridDim.x * blockDim.x) {
    const long i = i8 << 3;
    const float4 g0 = *reinterpret_cast<const float4*>(a.g + i), g1 = *reinterpret_cast<const float4*>(a.g + i + 4);
    const float4 p0 = *reinterpret_cast<const float4*>(a.p + i), p1 = *reinterpret_cast<const float4*>(a.p + i + 4);
    const float4 m0 = *reinterpret_cast<const float4*>(a.m + i), m1 = *reinterpret_cast<const float4*>(a.m + i + 4);
    const float4 v0 = *reinterpret_cast<const float4*>(a.v + i), v1 = *reinterpret_cast<const float4*>(a.v + i + 4);
    const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w}, pp[8] = {p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, p1.z, p1.w};
    const float mm[8] = {m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, m1.z, m1.w}, vv[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
    float pn[8], mn[8], vn[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float g = gg[e] * a.gscale;
      if (a.clip > 0.f) g = fminf(fmaxf(g, -a.clip), a.clip);             // clip_grad_value_ (Solver.py:211-212)
      if (a.weight_decay != 0.f) g += a.weight_decay * pp[e];
      mn[e] = a.beta1 * mm[e] + (1.f - a.beta1) * g;
      vn[e] = a.beta2 * vv[e] + (1.f - a.beta2) * g * g;
      pn[e] = pp[e] - step_size * mn[e] / (sqrtf(vn[e]) * inv_sqrt_bc2 + a.eps);
    }
    *reinterpret_cast<float4*>(a.m + i) = make_float4(mn[0], mn[1], mn[2], mn[3]); *reinterpret_cast<float4*>(a.m + i + 4) = make_float4(mn[4], mn[5], mn[6], mn[7]);
    *reinterpret_cast<float4*>(a.v + i) = make_float4(vn[0], vn[1], vn[2], vn[3]); *reinterpret_cast<float4*>(a.v + i + 4) = make_float4(vn[4], vn[5], vn[6], vn[7]);
    *reinterpret_cast<float4*>(a.p + i) = make_float4(pn[0], pn[1], pn[2], pn[3]); *reinterpret_cast<float4*>(a.p + i + 4) = make_float4(pn[4], pn[5], pn[6], pn[7]);
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    *reinterpret_cast<float4*>(a.g + i) = z; *reinterpret_cast<float4*>(a.g + i + 4) = z;   // the bucket is left zeroed for the next accumulation pass
    if (a.pimg || a.frag.n > 0) {
      bf16x8 q;
#pragma unroll
      for (int e = 0; e < 8; ++e) q[e] = to_bf16(pn[e]);
      if (a.pimg) *reinterpret_cast<bf16x8*>(a.pimg + i) = q;
      // forward fragment image: element (c, r) of matrix g of entry e sits at lo + g * gstride + c * RED + r; its image position is
      // ((c / 32 * (RED / 16) + r / 16) * 64 + c % 32 + 32 * ((r % 16) / 8)) * 8 + r % 8 (mlp_fused.hip: frag_images_kernel).  r is a multiple
      // of 8 here (offsets, strides and RED are multiples of 8): the piece is one 16-byte store.
      // (32-bit arithmetic -- a bucket has < 2^31 elements --, quotients through one float reciprocal + a one-step correction: the 64-bit
      //  divisions of the first version cost this launch 5 us)
      const int ii = (int)i;
      for (int e = 0; e < a.frag.n; ++e) {
        const int j = ii - (int)a.frag.lo[e], gs = (int)a.frag.gstride[e];
        if (j < 0 || j >= a.frag.nb[e] * gs) continue;
        int g = (int)((float)j * a.frag.inv_gs[e]);
        g -= g * gs > j; g += (g + 1) * gs <= j;
        const int rem = j - g * gs;
        if (rem >= a.frag.mat[e]) continue;
        const int RED = a.frag.RED[e];
        int c = (int)((float)rem * a.frag.inv_red[e]);
        c -= c * RED > rem; c += (c + 1) * RED <= rem;
        const int r = rem - c * RED;
        const int pos = (int)a.frag.lo[e] + g * gs + (((c >> 5) * (RED >> 4) + (r >> 4)) * 64 + (c & 31) + 32 * ((r & 15) >> 3)) * 8;
        *reinterpret_cast<bf16x8*>(a.frag.dst + pos) = q;
      }
    }
  }
}

}  // namespace

int copy_rows(hipStream_t s, const CopyTable& t, long n) {
  if (t.n <= 0) return MIMRL_OK;
  hipLaunchKernelGGL(copy_rows_kernel, dim3(grid_for(n, 256, 64), t.n), dim3(256), 0, s, t, n);
  LAUNCH_CHECK();
  return MIMRL_OK;
}

bool mi_sep_fused_supported(int B) { return B >= 32 && B <= 128 && B % 32 == 0; }
int mi_sep_fused(hipStream_t s, const float* tout, float* dtout, float* mi, float* mil, const float* gscale, int E, int B,
                 int bound, unsigned lossform, int do_bwd, const float* lb, float* dlb, long lb_stride) {
  if (!mi_sep_fused_supported(B)) return set_error(MIMRL_ERR_ARG, "mi_sep_fused: batch %d unsupported", B);
  static bool attr = false;
  if (!attr) {
    HIPX(hipFuncSetAttribute(reinterpret_cast<const void*>(mi_sep_fused_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                             65536 + 128 * 130 * 2));
    attr = true;
  }
  hipLaunchKernelGGL(mi_sep_fused_kernel, dim3(E), dim3(1024), (size_t)B * B * sizeof(float) + (size_t)B * (B + 2) * 2, s, tout, dtout, mi, mil, gscale, B,
                     bound, lossform, do_bwd, lb, dlb, lb_stride, dbg_env("MIMRL_DBG_MI") ? atoi(dbg_env("MIMRL_DBG_MI")) : 0);
  LAUNCH_CHECK();
  return MIMRL_OK;
}

int mi_sep_nce_tiled(hipStream_t s, const float* tout, float* dtout, float* mi, float* mil, const float* gscale, int E, int B, int do_bwd) {
  if (!mi_sep_fused_supported(B)) return set_error(MIMRL_ERR_ARG, "mi_sep_nce_tiled: batch %d unsupported", B);
  const size_t lds = (size_t)(B + 32) * NXP * 2 + (size_t)32 * (B + 1) * sizeof(float);
  hipLaunchKernelGGL(mi_sep_nce_kernel, dim3(B / 32, E), dim3(256), lds, s, tout, dtout, mi, mil, gscale, B, do_bwd);
  LAUNCH_CHECK();
  return MIMRL_OK;
}

int mi_bound_fwd_bwd(hipStream_t s, const float* scores, float* dscores, float* mi, float* mil, const float* gscale, int E,
                     int B, int bound, unsigned lossform, const float* lb, float* dlb, long lb_stride, float* nce_ws) {
  if (B > 1024) return set_error(MIMRL_ERR_ARG, "mi_bound: batch %d > 1024 per rank", B);
  if (bound == BOUND_INFONCE && E <= NCE_MAX_EST && nce_ws) {   // row-wise bound: one wave per row (the general kernel -- no shared state -- is one workgroup per estimator)
    hipLaunchKernelGGL(mi_infonce_rows_kernel, dim3((B + 15) / 16, E), dim3(1024), 0, s, scores, dscores, mi, mil, gscale, B, reinterpret_cast<NceWs*>(nce_ws));
    LAUNCH_CHECK();
    return MIMRL_OK;
  }
  hipLaunchKernelGGL(mi_bound_kernel, dim3(E), dim3(1024), 0, s, scores, dscores, mi, mil, gscale, B, bound, lossform, lb, dlb,
                     lb_stride);
  LAUNCH_CHECK();
  return MIMRL_OK;
}

int pair_expand_fwd(hipStream_t s, const float* P, const float* Q, float* a1, int E, int B, int Hd) {
  hipLaunchKernelGGL(pair_expand_fwd_kernel, dim3(B, E), dim3(256), 0, s, P, Q, a1, B, Hd);
  LAUNCH_CHECK();
  return MIMRL_OK;
}
int pair_expand_bwd(hipStream_t s, const float* a1, float* da1, float* dP, float* dQ, int E, int B, int Hd) {
  hipLaunchKernelGGL(pair_reduce_p_kernel, dim3(B, E), dim3(256), 0, s, a1, da1, dP, B, Hd);
  LAUNCH_CHECK();
  hipLaunchKernelGGL(pair_reduce_q_kernel, dim3(B, E), dim3(256), 0, s, da1, dQ, B, Hd);
  LAUNCH_CHECK();
  return MIMRL_OK;
}
int pair_reduce_q(hipStream_t s, const float* du1, float* dQ, int E, int B, int Hd) {
  hipLaunchKernelGGL(pair_reduce_q_kernel, dim3(B, E), dim3(256), 0, s, du1, dQ, B, Hd);
  LAUNCH_CHECK();
  return MIMRL_OK;
}
int relu_bwd_inplace(hipStream_t s, const float* a, float* g, long n) {
  hipLaunchKernelGGL(relu_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, s, a, g, n);
  LAUNCH_CHECK();
  return MIMRL_OK;
}

int cmi_assemble(hipStream_t s, const CmiAssembleArgs& a) {
  hipLaunchKernelGGL(cmi_assemble_kernel, dim3(2 * a.n, a.ncall), dim3(128), 0, s, a);
  LAUNCH_CHECK();
  return MIMRL_OK;
}

int cmi_loss_fwd_bwd(hipStream_t s, const float* logits, float* dlogits, float* bce, float* cmi, const float* g_bce,
                     const float* g_cmi, int E, int n, int hardtanh) {
  hipLaunchKernelGGL(cmi_loss_kernel, dim3(E), dim3(256), 0, s, logits, dlogits, bce, cmi, g_bce, g_cmi, n, hardtanh);
  LAUNCH_CHECK();
  return MIMRL_OK;
}

int gather_sum(hipStream_t s, float* dst, const GatherSum& g, int B, int D, int accumulate) {
  hipLaunchKernelGGL(gather_sum_kernel, dim3(grid_for((long)B * D)), dim3(256), 0, s, dst, g, B, D, accumulate);
  LAUNCH_CHECK();
  return MIMRL_OK;
}

int gather_sum4(hipStream_t s, const GatherSum4& g, int B, int D, int first) {
  hipLaunchKernelGGL(gather_sum4_kernel, dim3(grid_for((long)B * D), 4 - first), dim3(256), 0, s, g, B, D, first);
  LAUNCH_CHECK();
  return MIMRL_OK;
}

int top1_bwd(hipStream_t s, const float* dout, const float* W, const float* act, float* dz, float* dW, float* db_top, float* db_below,
             int nb, int rows, int brows, int din, long pstride) {
  if (din % 4 != 0 || din > 1024 || 1024 % din != 0) return set_error(MIMRL_ERR_ARG, "top1_bwd: width %d unsupported", din);
  hipLaunchKernelGGL(top1_bwd_kernel, dim3((rows + 255) / 256, nb), dim3(256), 0, s, dout, W, act, dz, dW, db_top, db_below, rows, brows, din,
                     pstride);
  LAUNCH_CHECK();
  return MIMRL_OK;
}

int adam_step(hipStream_t s, const AdamArgs& a) {
  if (a.frag.n > 0 || a.sb_on) {
    if (a.n % 8 != 0 || a.fold.n != 0 || a.frag.n > 12 || (a.frag.n > 0 && a.n >= (1L << 24))) return set_error(MIMRL_ERR_ARG, "adam_step: the 8-wide kernel needs n %% 8 == 0, no parked pieces, <= 12 image entries");
    hipLaunchKernelGGL(adam8_kernel, dim3(grid_for(a.n / 8, 256, 2048)), dim3(256), 0, s, a);
    LAUNCH_CHECK();
    return MIMRL_OK;
  }
  hipLaunchKernelGGL(adam_kernel, dim3(grid_for(a.n, 256, 2048)), dim3(256), 0, s, a);
  LAUNCH_CHECK();
  return MIMRL_OK;
}

}  // namespace mimrl
