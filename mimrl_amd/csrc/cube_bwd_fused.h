// Fused data-gradient chains of one CubeMLP block (MLPProcess.py:94-122, ln_last form), bf16 MFMA operands.
//   L axis:  dZ -> LayerNorm(L) backward -> dY -> dU = (W2^T dY) * act'(U) -> dX = W1^T dU + Wr^T dY
//            columns (k,d) are independent: one workgroup = one sample x 128 columns, everything between dZ and dX in LDS
//   D axis:  dZ -> LayerNorm(D) backward -> dY -> dU = (dY W2) * act'(U) -> dX = dU W1 + dY Wr
//            rows are independent: one workgroup = 64 rows
// Both also write dY / dU (the weight-gradient GEMMs read them) and accumulate the bias gradients.
#pragma once
#include "common.h"

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
  const float *dz, *y, *mean, *rstd, *gamma;   // [R,od] [R,od] [R] [R] [od]
  const float* u;                              // pre-activation [R,hd]
  const float *w2, *w1, *wr;                   // [od,hd] [hd,id] [od,id]
  float *dy, *du, *dx;                         // [R,od] [R,hd] [R,id]
  float *dgamma, *dbeta, *db2, *db1;           // [od] [od] [od] or null, [hd] or null     (accumulated)
  long R;
  int act;                                     // id == hd == od == 128
};
bool daxis_bwd_supported(int id, int hd, int od);
int daxis_bwd_fused(hipStream_t s, const DAxisBwdArgs& a);

}  // namespace mimrl
