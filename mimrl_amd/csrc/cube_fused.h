// Fused CubeMLP block forward (MLPProcess.py:94-122, ln_last form): one workgroup per sample keeps the [L, K, 128] tile
// in LDS (bf16) through L-axis MLP+residual -> LayerNorm(L) -> K-axis mix -> D-axis MLP+residual -> LayerNorm(D).
#pragma once
#include "common.h"
#include "model_ops.h"

namespace mimrl {

struct CubeFusedArgs {
  const float* x;                                              // block input  [B, il, K, 128]
  const float *l_w1, *l_b1, *l_w2, *l_b2, *l_wr, *l_g, *l_be;  // L axis: [hl,il] [hl] [ol,hl] [ol] [ol,il] [ol] [ol]
  KMixW kw;                                                    // K axis (ik == hk == ok == K)
  const float *d_w1, *d_b1, *d_w2, *d_b2, *d_wr, *d_g, *d_be;  // D axis: 128x128 each
  // activations kept for the backward pass (all null when save == 0); same buffers/layouts as the unfused path
  float *l_u, *l_h, *l_y, *l_z, *l_mean, *l_rstd;              // [B,hl,C] [B,hl,C] [B,ol,C] [B,ol,C] [B,C] [B,C]
  float* k_z;                                                  // [B, ol, K, 128]
  float *d_u, *d_h, *d_y, *d_mean, *d_rstd;                    // [B*ol*K, 128] x3, [B*ol*K] x2
  float* d_z;                                                  // block output [B, ol, K, 128]
  int B, il, hl, ol, K, act, save;
  int dbg_phase;                                               // tuning only: stop after phase n (0 = run everything)
};

// true if this block's configuration is covered by the fused kernel (else the engine uses the unfused path)
bool cube_fused_supported(int il, int hl, int ol, int ik, int hk, int ok, int id, int hd, int od, bool ln_first,
                          bool res_project, bool bias, const float* dropout_mlp);
int cube_block_fwd_fused(hipStream_t s, const CubeFusedArgs& a);

#ifdef MIMRL_PHASE_PROBE
int cube_fwd_read_phases(long long* out);   // 128 ticks, see cube_fused.hip
#endif

}  // namespace mimrl
