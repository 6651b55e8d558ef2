// L-axis MLP of a CubeMLP block (MLPProcess.py:95-104, ln_last form: Y = W2 act(W1 X + b1) + b2 + Wr X over the L axis of every
// (sample, k, d) column, then LayerNorm over the output rows) for LONG sequences -- L = 500 / 1000 (cfg3 / cfg5), where the sample tile of
// the fused block kernel (cube_fused.hip: [L, 3, 128] in LDS) does not exist and the GEMM chain read the [L, 384] slab of every sample twice.
// One launch, one pass over the slab; the saved activations (U, H, Y, mean, rstd) are those of the chain, so the backward is unchanged.
#pragma once
#include "common.h"

namespace mimrl {

struct LAxisLongArgs {
  const float* x;                                   // [B, il, C]
  const float *w1, *b1, *w2, *b2, *wr, *g, *be;     // [hl,il] [hl]|null [ol,hl] [ol]|null [ol,il] [ol] [ol]
  float *u, *h, *y, *z, *mean, *rstd;               // [B,hl,C] x 2 (both or null), [B,ol,C] (y: or null), [B,C] x 2 (both or null)
  int B, il, hl, ol, C, act;
};
bool laxis_fwd_long_supported(int il, int hl, int ol, int C);
// f16: fp16 MFMA operands (the forward products' 16-bit type, GemmDesc::f16), else bf16
int laxis_fwd_long(hipStream_t s, const LAxisLongArgs& a, bool f16);

}  // namespace mimrl
