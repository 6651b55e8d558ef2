// Generic strided / batched GEMM on the gfx950 matrix cores.
//   C[b][m,n] = epilogue( alpha * sum_k A[b][m,k] * B[b][k,n] )
// fp32 mode: v_mfma_f32_32x32x2_f32 (exact fp32 fma chain); bf16 mode: operands rounded to bf16 while they are
// staged into LDS, v_mfma_f32_32x32x16_bf16, fp32 accumulate.  All matrices are fp32 in HBM with arbitrary
// element strides, so transposes, the L-axis mixes of CubeMLP (left-multiplication of a [L, K*D] sample tile)
// and the grouped critic towers are all expressed as strides -- no data is ever physically permuted.
#pragma once
#include "common.h"

namespace mimrl {

struct GemmDesc {
  const float* A = nullptr;
  const float* B = nullptr;
  float* C = nullptr;
  int M = 0, N = 0, K = 0, batch = 1;
  long sa_m = 0, sa_k = 0, sa_b = 0;
  long sb_k = 0, sb_n = 0, sb_b = 0;
  long sc_m = 0, sc_n = 0, sc_b = 0;
  // optional two-level batch: entry b = (outer, inner) = (b / batch_in, b % batch_in) with its own outer strides for
  // A, B, C (and pre / gradact_u, which share C's layout) and bias_n -- e.g. (modality, direction) of the GRU weights
  int batch_in = 0; long sa_bo = 0, sb_bo = 0, sc_bo = 0, bias_n_bo = 0;
  // optional gap in A's row (m) axis: rows m >= a_gap_at live a_gap_rows further on (dgh = columns [0,2H) u [3H,4H) of dg)
  int a_gap_at = 0, a_gap_rows = 0;
  const float* bias_n = nullptr; long bias_n_b = 0;   // + bias_n[b*bias_n_b + n]
  const float* bias_m = nullptr; long bias_m_b = 0;   // + bias_m[b*bias_m_b + m]
  float alpha = 1.f;
  float beta = 0.f;                  // + beta * C_old
  int act = ACT_NONE;                // applied after bias/beta
  float* pre = nullptr;              // optional: pre-activation copy (same strides as C)
  const float* gradact_u = nullptr;  // optional: multiply by act'(u[m,n]) (same strides as C) -- backward of act
  int atomic = 0;                    // C += result with float atomics (batch acts as an extra reduction when sc_b==0)
  float* colsum = nullptr; long colsum_b = 0;   // optional: colsum[b*colsum_b + n] += sum_m (stored value)  (bias gradients)
  // optional SECOND product accumulated into the same output tile:  C = epi( A.B + A2.B2 )   (e.g. W2.H + Wr.X)
  const float* A2 = nullptr; const float* B2 = nullptr; int K2 = 0;
  long sa2_m = 0, sa2_k = 0, sa2_b = 0, sb2_k = 0, sb2_n = 0, sb2_b = 0;
  // operand storage: A (and A2) / B (and B2) are bf16 arrays behind the float-typed pointers (strides in bf16 elements).  Fast
  // path only (16-byte pieces go straight to LDS, no conversion): everything must be 8-element aligned, else gemm() fails.
  // a_pad4: A is readable -- and ZERO -- up to the next multiple of 4 along its contiguous axis (a padded copy of a weight matrix whose
  // width is not a multiple of 4, e.g. the 50 x 50 L-axis fc2 of CubeMLP: with it the product takes the 16-byte-load kernels although
  // K % 4 != 0 (k-contiguous A) or M % 4 != 0 (row-contiguous A); the surplus columns multiply into outputs as zeros / are never stored)
  int a_pad4 = 0;
  // With f16 = 1 (below) and BOTH flags set the 16-bit storage type is fp16 and the layouts are (KC, KC): gemm_fast_f16s_kernel.
  int a_bf16 = 0, b_bf16 = 0;
  // with a_bf16 = b_bf16 = 1, both operands row-contiguous (a weight gradient): B is stored as FP16 (a forward pass wrote it for an fp16
  // product) and is converted to bf16 in registers on its way to LDS -- dW_ih of GRU layer 1 reads the layer-0 recurrence's fp16 copy of h
  // (256 MB at cfg3) instead of the fp32 outputs (512 MB, fetched ~2x by the 128-wide tiles)
  int b_f16cvt = 0;
  // bf16 mode only: round the operands to FP16 instead of bf16 (v_mfma_f32_32x32x16_f16: same rate, 11 instead of 8 significant bits;
  // saturating conversion).  For FORWARD products whose operands have a bounded range (inputs, weights, LayerNorm'd activations) and
  // feed the ill-conditioned backward of CubeMLP (see cube_fused.hip).  Honoured by the fast path with both operands k-contiguous
  // (every forward product of the step at aligned sizes); other kernels ignore it and round to bf16.
  int f16 = 0;
  // output storage: C is an FP16 array behind the float-typed pointer (strides in fp16 elements); plain store only (no atomic / split-K /
  // beta / pre).  For the hoisted GRU input projections gx[B,T,3H] of long sequences, which are written once and read once: at cfg3 the
  // projection is bound by its 786 MB of fp32 stores.
  int c_f16 = 0;
  // ... or a BF16 array (tall LDS-DMA kernel only, gemm_tall_ok): dh0, the gradient the layer-0 BPTT reads once
  int c_bf16 = 0;
};

// A is [M,K] row-major (lda), B given as W[N,K] row-major (ldw):  C = A * W^T
inline GemmDesc gemm_nt(const float* A, long lda, const float* W, long ldw, float* C, long ldc, int M, int N, int K) {
  GemmDesc d; d.A = A; d.B = W; d.C = C; d.M = M; d.N = N; d.K = K;
  d.sa_m = lda; d.sa_k = 1; d.sb_k = 1; d.sb_n = ldw; d.sc_m = ldc; d.sc_n = 1;
  return d;
}
// C = A * B, A [M,K] (lda), B [K,N] (ldb)
inline GemmDesc gemm_nn(const float* A, long lda, const float* Bm, long ldb, float* C, long ldc, int M, int N, int K) {
  GemmDesc d; d.A = A; d.B = Bm; d.C = C; d.M = M; d.N = N; d.K = K;
  d.sa_m = lda; d.sa_k = 1; d.sb_k = ldb; d.sb_n = 1; d.sc_m = ldc; d.sc_n = 1;
  return d;
}
// C = A^T * B, A stored [K,M] (lda), B [K,N] (ldb)   (weight gradients: dW = dY^T X)
inline GemmDesc gemm_tn(const float* A, long lda, const float* Bm, long ldb, float* C, long ldc, int M, int N, int K) {
  GemmDesc d; d.A = A; d.B = Bm; d.C = C; d.M = M; d.N = N; d.K = K;
  d.sa_m = 1; d.sa_k = lda; d.sb_k = ldb; d.sb_n = 1; d.sc_m = ldc; d.sc_n = 1;
  return d;
}

struct GemmPlan {
  int variant;        // 0 fp32 generic, 1 bf16 generic (BK 32), 2 bf16 lean BK 64, 3 bf16 lean BK 128, >= 10 fast path
  int fast, tm, tn;   // fast path: block tile (64*tm) x (64*tn)
  int ca, cb;         // fast path: operand layout classes (1 k-contiguous, 2 row-contiguous)
  int nsplit;         // split-K factor (partials are combined with float atomics)
  int kt_per;         // k-tiles per split
  long tiles;         // output tiles (64x64) incl. batch
};
void gemm_plan(const GemmDesc& d, bool bf16, GemmPlan* p);
int gemm(hipStream_t s, const GemmDesc& d, bool bf16);
// gemm_tall.hip: C[M, N] = A[M, K] . W[N, K]^T (+ A2 . W2^T) (+ bias_n) for M >= 4096 with BOTH operands stored in 16 bits and k-contiguous
// (a_bf16 = b_bf16 = 1, sa_k = sb_k = 1; bf16, or fp16 with f16 = 1), plain epilogue, fp32 or fp16 (c_f16) output: 256 x 128 tiles, 3-stage
// LDS ring filled by LDS-DMA.  gemm() routes there by itself; gemm_tall_ok() is the eligibility test.
bool gemm_tall_ok(const GemmDesc& d);
int gemm_tall(hipStream_t s, const GemmDesc& d);
// up to 6 independent GEMMs as ONE launch when they are plain (no second product / operand-reading epilogue), bf16 and of one
// operand-layout class of the fast path; otherwise n ordinary launches.  No split-K: meant for many small products.
int gemm_group(hipStream_t s, const GemmDesc* ds, int n, bool bf16);
// up to 12 plain accumulate-into-zeroed-output GEMMs (C += A.B with float atomics; weight gradients) of one layout class -- both
// operands k-contiguous or both row-contiguous -- as ONE launch, each problem with its own split-K factor; otherwise n launches.
int gemm_group_splitk(hipStream_t s, const GemmDesc* ds, int n, bool bf16);

}  // namespace mimrl
