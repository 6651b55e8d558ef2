// 1-layer bidirectional LSTM encoder of `--encoders lstm` (Model.py:250-252,441-447), packed-sequence semantics.
// Round 5: on the matrix cores like the bi-GRU (persistent workgroups of 4 batch rows, W_hh as register-resident MFMA B fragments, fp32
// MFMA = exact parity mode, bf16 operands in the bf16 mode); the round-1 scalar design -- one workgroup per (sample, direction,
// modality), 512 threads = the 512 gate rows -- stays as the reference the operator test compares with (MIMRL_LSTM_SCALAR=1).
#pragma once
#include "common.h"

namespace mimrl {

constexpr int LSTM_H = 128;

struct LstmSeq {
  const float* gx;     // [B,T,4H] x W_ih^T + b_ih, gate order i,f,g,o
  const float* w_hh;   // [4H,H]
  const float* b_hh;   // [4H]
  float* out;          // [B,T,out_ld]; this direction writes columns [dir*H, dir*H+H)
  float* saved;        // [B,T,6H]: i,f,g,o (activated), c_prev, tanh(c)   (null: inference)
};
struct LstmFwdArgs {
  LstmSeq seq[2][2];   // [modality][direction]
  const int* lens[2];
  int B, T, out_ld, nmod;
};
// mode: 0 scalar fp32 reference kernels, 1 fp32 MFMA (exact), 2 bf16 MFMA operands -- see lstm.hip
int lstm_forward(hipStream_t s, const LstmFwdArgs& a, int mode = 1);

struct LstmSeqBwd {
  const float* w_hh;   // [4H,H]
  const float* saved;  // from the forward pass
  const float* out;    // forward outputs (source of h_prev)
  const float* dout;   // [B,T,dout_ld] gradient w.r.t. this direction's output
  float* dg;           // [B,T,4H] gradient w.r.t. the gate pre-activations (x side == h side)
  float* hprev;        // [B,T,H]  h_{t-1} of this direction (operand of the dW_hh GEMM)
};
struct LstmBwdArgs {
  LstmSeqBwd seq[2][2];
  const int* lens[2];
  int B, T, out_ld, dout_ld, nmod;
};
int lstm_backward(hipStream_t s, const LstmBwdArgs& a, int mode = 1);

}  // namespace mimrl
