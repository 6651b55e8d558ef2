// Fused data-gradient chains of one CubeMLP block (MLPProcess.py:94-122, ln_last form), bf16 MFMA operands.
//   L axis:  dZ -> LayerNorm(L) backward -> dY -> dU = (W2^T dY) * act'(U) -> dX = W1^T dU + Wr^T dY
//            columns (k,d) are independent: one workgroup = one sample x 128 columns, everything between dZ and dX in LDS
//   D axis:  dZ -> LayerNorm(D) backward -> dY -> dU = (dY W2) * act'(U) -> dX = dU W1 + dY Wr
//            rows are independent: one workgroup = 64 rows
// Both also write dY / dU (the weight-gradient GEMMs read them) and accumulate the bias gradients.
#pragma once
#include "common.h"
#include "model_ops.h"

namespace mimrl {

struct LAxisBwdArgs {
  const float *dz, *y, *mean, *rstd, *gamma;   // [B,ol,C] [B,ol,C] [B,C] [B,C] [ol]
  const float* u;                              // pre-activation [B,hl,C]
  const float *w2, *w1, *wr;                   // [ol,hl] [hl,il] [ol,il]
  float *dy, *du, *dx;                         // [B,ol,C] [B,hl,C] [B,il,C]
  float *db2, *db1;                            // [ol] or null, [hl] or null (accumulated); the LayerNorm parameter
                                               // gradients are row sums over (b, c): colln_param_grads, off the critical path
  int B, il, hl, ol, C, act;
};
bool laxis_bwd_supported(int il, int hl, int ol, int C);
int laxis_bwd_fused(hipStream_t s, const LAxisBwdArgs& a);

struct DAxisBwdArgs {
  const float *dz, *y, *mean, *rstd, *gamma;   // [R,128] [R,128] [R] [R] [128]
  const float* u;                              // pre-activation [R,128]
  const __bf16 *w2t, *w1t, *wrt;               // TRANSPOSED bf16 images [n][k] of W2[o,h], W1[h,i], Wr[o,i] (wt_transpose_bf16)
  float *dy, *du, *dx;                         // [R,128] x3
  long R;
  int act;                                     // id == hd == od == 128
  // optional (all four or none): parameter gradients that are column sums over the rows of what this kernel holds anyway --
  // dgamma = sum dz * xhat, dbeta = sum dz (LayerNorm), db2 = sum dY, db1 = sum dU; accumulated with atomics ([128] each).
  // Null: the caller runs rowln_param_grads / colsum side kernels instead.
  float *dgamma, *dbeta, *db2, *db1;
};
bool daxis_bwd_supported(int id, int hd, int od);
int daxis_bwd_fused(hipStream_t s, const DAxisBwdArgs& a);

// dst[m][n][k] = bf16(src[m][k][n]) for up to 12 square 128x128 matrices (one launch per stage, off the critical path)
struct WtTransposeArgs { const float* src[12]; __bf16* dst[12]; int n; };
int wt_transpose_bf16(hipStream_t s, const WtTransposeArgs& a);

// the four column-sum parameter gradients of the D axis (rows x 128) in one streaming pass: dgamma = sum dz * xhat, dbeta = sum dz,
// db2 = sum dY, db1 = sum dU (db2 / db1 may be null); accumulated with atomics
int daxis_param_grads(hipStream_t s, const float* y, const float* mean, const float* rstd, const float* dz, const float* dy,
                      const float* du, float* dgamma, float* dbeta, float* db2, float* db1, long R);

// LayerNorm(D) parameter gradients as column sums over rows: dgamma[j] = sum_r dz * xhat, dbeta[j] = sum_r dz
int rowln_param_grads(hipStream_t s, const float* y, const float* mean, const float* rstd, const float* dz, float* dgamma,
                      float* dbeta, long R, int n);

#ifdef MIMRL_PHASE_PROBE
int cube_bwd_read_phases(long long* out);   // 64 ticks, see cube_bwd_fused.hip
#endif

}  // namespace mimrl
