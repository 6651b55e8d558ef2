// Layer-0 bi-GRU weight gradients, one pass over dg (gru_wgrad.hip).
#pragma once
#include "common.h"

namespace mimrl {

struct GruWgradSeq {
  const __bf16* dg;     // [rows, 4H] = [dr' | dz' | dn' | dn' r]   (gru_bwd_kernel, bf16 dg)
  const __bf16* x;      // [rows, kp] packed bf16 inputs of the sequence's modality (l0_pack: xb)
  const __bf16* hp;     // [rows, H]  h_prev
  float* dw_ih;         // [3H, kp] += dgx^T x      (packed scratch, zero before the pass)
  float* dw_hh;         // [3H, H]  += dgh^T h_prev
};
struct GruWgradArgs {
  GruWgradSeq seq[4];   // [modality * 2 + direction]
  long rows;            // B * T
  int kp;               // packed input width (multiple of 8, <= 96)
  int nsplit = 0, kt_per = 0;   // filled by gru_wgrad()
};
bool gru_wgrad_ok(long rows, int kp);
int gru_wgrad(hipStream_t s, const GruWgradArgs& a);

}  // namespace mimrl
