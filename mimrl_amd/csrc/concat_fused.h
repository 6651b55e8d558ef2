// Fused forward pass of the concat critic (VMI.py:59-65: scores[i,j] = f([x_i | y_j]) through Linear/ReLU x3 + Linear, VMI.py:13-22),
// bf16 MFMA operands, fp32 accumulate.  The B*B pair rows of an estimator are the real dense contraction of this workload
// (cfg3: 5 x 65,536 rows x 256 x 256 x 2 layers = 86 GFLOP per pass); as a chain of GEMMs every layer's [B*B, 256] activation
// goes out to HBM in fp32 and comes back (3 TB/s, HBM-bound at 107 TFLOP/s).  Here a workgroup owns 128 pair rows for the WHOLE
// stack: layer 0 in its separable form (relu(P_i + Q_j)) is generated straight into LDS, both 256x256 hidden layers run with the
// activation tile resident in LDS (bf16), weights streamed from the L2-resident bf16 image in double-buffered 32-k chunks, and the
// 256 -> 1 score head is a dot product on the accumulators.  The post-ReLU activations are still written out once in fp32 (the
// unfused backward pass reads them).
#pragma once
#include "common.h"

namespace mimrl {

struct ConcatFwdArgs {
  const float* P; const float* Q;            // [E][B][256]: W0x x_i and W0y y_j + b0 (the separable first layer)
  const __bf16* W1; const __bf16* W2;         // bf16 images of the two hidden layers, [256 out][256 in] row-major; estimator e at + e*pstride
  const float *b1, *b2, *w3, *b3;             // fp32 biases, score-head weight [256] and bias [1]; estimator e at + e*pstride
  long pstride;                               // parameter stride between estimators (elements, same for the image and the fp32 bucket)
  // what the backward pass gets (save): 0 nothing (evaluation), 1 fp32 activations a0, a1, a2 (the unfused GEMM-chain backward),
  // 2 compact: bf16 values a0b, a1b (operands of the weight-gradient products, which round to bf16 anyway), a2 as fp32 (weight-streaming
  //   kernel) or fp16 (weights-stationary kernel: concat_fwd_a2_f16) -- the score head's weight gradient sum ds * a2 cancels to ~1e-3 of
  //   its terms: bf16 values are not good enough; read by concat_dw3 only -- + ReLU
  //   bitmasks m0, m1, m2 (one 32-bit word per row and 32 columns; the fused backward, stage 1),
  // 3 bitmasks only (the fused backward of stage 2: no weight gradients, only the signs are needed)
  // (round 5: save >= 2 also writes m0, the sign of the pair-expanded layer 0 -- the backward recomputed it from P_i + Q_j with 64 loads
  //  per lane and tile)
  int save;
  float *a0, *a1, *a2;                        // [E][B*B][256] fp32 (save == 1)
  __bf16 *a0b, *a1b;                          // [E][B*B][256] bf16 (save == 2)
  uint32_t *m0, *m1, *m2;                     // [E][B*B][8]        (save == 2, 3): bit c of word [row][g] = sign of column 32 g + c
  float* scores;                              // [E][B*B]   row p = i*B + j
  int no_a0 = 0;                              // (weights-stationary kernel, save == 2) a0b is NOT written: concat_dw regenerates it from P and Q
  int E, B;
};
bool concat_fwd_fused_supported(int B, int hid);

// The data-gradient chain of the same stack per 128-row tile (B a multiple of 128: a tile = one x row i, 128 consecutive y rows j):
//   dZ2 = ds w3^T (.) [a2 > 0]  ->  dZ1 = (dZ2 W2) (.) [a1 > 0]  ->  dZ0 = (dZ1 W1) (.) [a0 > 0]
// with the gradient tile resident in LDS (bf16) and the TRANSPOSED bf16 weight images streamed as in the forward pass.  Written out:
// dz0 (fp32: pair_reduce_q sums it over i), dP[i] = sum_j dZ0 (this tile's column sums), and -- stage 1 only -- dZ2 / dZ1 as bf16 for
// the weight-gradient GEMMs plus the column-sum gradients db1, db2, dw3, db3 (float atomics into the zeroed bucket).
struct ConcatBwdArgs {
  const float* ds;                            // [E][B*B]   d loss / d score
  const float *a0, *a1, *a2;                  // saved post-ReLU activations, fp32 (the forward kernel's save == 1) -- or, compact:
  int compact;                                // 1: masks m0, m1, m2 instead (dw3 is then NOT computed here: concat_dw3)
  const uint32_t *m0, *m1, *m2; const float *P, *Q;   // (P, Q: unused since round 5)
  const float* w3;                            // score-head weight [256], estimator e at + e*pstride
  const __bf16 *W2T, *W1T;                    // transposed bf16 images [256 in][256 out], estimator e at + e*pstride
  long pstride;
  float* dz0;                                 // [E][B*B][256] fp32 (not written when dQ is set)
  float* dQ;                                  // optional (compact saves), [E][B][256]: dQ[j] = sum_i dZ0[i, j] from inside the launch (runs of tiles per
                                              // workgroup, partial sums in registers, then a fixed-order reduction of the partials) -- dz0 / pair_reduce_q
                                              // not needed.  Needs dq_part = scratch of concat_bwd_dq_scratch(E, B) floats (> 0: else not available)
  float* dq_part; int dq_slots;               // (dq_slots: filled in by concat_bwd_fused)
  float* dP;                                  // [E][B][256]; B == 128: plain stores, else accumulated (caller zeroes it)
  __bf16 *dz2, *dz1;                          // [E][B*B][256] bf16 or null (stage 2: no weight gradients)
  float *db1, *db2, *dw3, *db3;               // gradient slots (estimator e at + e*pstride) or null
  int E, B;
  int no_dz2 = 0;                             // (weights-stationary kernel, stage 1) dZ2 is NOT written: concat_dw regenerates it from ds, w3 and m2
};
bool concat_bwd_fused_supported(int B, int hid);
bool concat_bwd_dq_plan(int E, int B, int* per, int* nwg, int* slots);
long concat_bwd_dq_scratch(int E, int B);
int concat_bwd_fused(hipStream_t s, const ConcatBwdArgs& a);
int concat_fwd_fused(hipStream_t s, const ConcatFwdArgs& a);
// concat_ws.hip (round 6): the same forward pass with the hidden-layer weights resident in registers (persistent workgroups, two-stage wave
// pipeline over 32-row units); concat_fwd_fused routes there when concat_fwd_ws_supported
bool concat_fwd_ws_supported(int B, int hid, int save);
int concat_fwd_ws(hipStream_t s, const ConcatFwdArgs& a);
// concat_ws_bwd.hip (round 6): the backward chain with both transposed weight images resident in registers; concat_bwd_fused routes there
// when the in-kernel dQ reduction is requested (compact saves + dq_part of concat_bwd_dq_scratch floats)
bool concat_bwd_ws_supported(int B, int hid);
long concat_bwd_ws_scratch(int E, int B);
int concat_bwd_ws(hipStream_t s, const ConcatBwdArgs& a);
int concat_fwd_ws4(hipStream_t s, const ConcatFwdArgs& a);   // tools/hw/concat_ws4.hip (experiment, not in the library): one wave per SIMD
// dw3[e] += ds[e]^T a2[e]  (compact saves: concat_bwd_fused leaves the score head's weight gradient to this streaming launch)
int concat_dw3(hipStream_t s, const float* ds, const float* a2, float* dw3, int E, int B, long pstride, bool a2_f16 = false);
// does a saving (save == 2) forward pass of this batch size leave a2 as fp16 behind the float-typed pointer?
bool concat_fwd_a2_f16(int B, int save);
#ifdef MIMRL_PHASE_PROBE
int concat_bwd_read_phases(long long* out);   // 16 slots, read-and-clear (concat_fused.hip)
#endif

}  // namespace mimrl
