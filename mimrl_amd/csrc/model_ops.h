// Non-GEMM kernels of Model.forward / backward (Model.py:388-519, MLPProcess.py:64-122): length inference,
// LayerNorm(+ReLU+dropout) epilogues, temporal means, CubeMLP axis LayerNorms, the K-axis mix and the head.
#pragma once
#include "common.h"

namespace mimrl {

struct RngKey {            // dropout keying; `step` is read from device memory so that hipGraph replays advance it
  uint32_t seed_lo, seed_hi;
  const int* step;         // device counter (bumped by begin_stage)
  int add = 0;             // added to *step: a forward pass issued ahead of its stage's begin_stage uses +1
};

// lens[b] = max(1, #rows t with sum_d |x[b,t,d]| != 0)                    (Model.py:425-432)
int seq_lengths(hipStream_t s, const float* x, int B, int T, int d, int* lens);

// cube[b,t,slot,:] = dropout(src[b,t,:])  (text branch, Model.py:461,475)
int text_post_fwd(hipStream_t s, const float* src, float* cube, int B, int T, int L, int K, int D, int slot, float p,
                  RngKey key, uint32_t stream_id);
// dsrc[b,t,:] = dropout_mask * dcube[b,t,slot,:]
// dmean (optional): gradient of the slot's temporal mean [B, D], added as dmean / T (feat_mean backward folded in)
int text_post_bwd(hipStream_t s, const float* dcube, float* dsrc, int B, int T, int L, int K, int D, int slot, float p,
                  RngKey key, uint32_t stream_id, const float* dmean = nullptr);

// cube[b,t,slot,:] = dropout(relu(LN(h[b,t,:H] + h[b,t,H:])))            (Model.py:452-461)
struct LnSide2 { const float *h2, *gamma, *beta; float *mean, *rstd, *ds, *dgamma, *dbeta; int slot; float p; uint32_t stream; };
int ln_relu_drop_fwd2(hipStream_t s, const LnSide2& a, const LnSide2& v, float* cube, int B, int T, int L, int K, int D, RngKey key);
// dmean_a / dmean_v (optional): gradients of the audio / video temporal means [B, D] (feat_mean backward folded in)
// ds_bf16: ds is written as a bf16 array (same element indices) -- the layer-1 BPTT reads it as such (GruBwdArgs::dout_bf16)
int ln_relu_drop_bwd2(hipStream_t s, const LnSide2& a, const LnSide2& v, const float* dcube, int B, int T, int L, int K, int D,
                      RngKey key, const float* dmean_a = nullptr, const float* dmean_v = nullptr, int ds_bf16 = 0);
// Layer-0 GRU operands in a common, 16-byte-aligned shape.  audio [rows, d_a] and video [rows, d_v] (d = 74 / 35 for MOSI: rows
// of 296 / 140 bytes, no 16-byte loads possible, and different widths, so no batching) are copied into xpack[2][rows, KP]
// (zero padded), the four W_ih matrices [384, d] into wpack[2][2][384, KP] and the four b_ih into bpack[2][2][384]: the input
// projection of both modalities and directions is then ONE batched k-contiguous GEMM, and the W_ih / W_hh weight gradients
// two (accumulated in packed scratch and scattered back by l0_unpack_grads).
struct L0Pack {
  const float* x[2]; int d[2];               // audio, video
  const float* w_ih[2][2]; const float* b_ih[2][2];
  float* xpack; float* wpack; float* bpack;
  long rows; int KP;
  // optional: the stage's begin-of-stage bookkeeping rides on workgroup (0,0) of this launch (step counters += 1, scalar block zeroed):
  // the pack is the first launch of the captured step, and a separate single-thread kernel in front of it was a launch + a gap
  int* bs_rng = nullptr; int* bs_adam = nullptr; float* bs_scal = nullptr; int bs_off = 0, bs_n = 0;
  // optional (with pack_weights): 16-bit images of the four LAYER-1 input matrices W_ih_l1 [384, 256], [modality][direction] back to back --
  // fp16 for the forward projection (gemm_fast_f16s_kernel), bf16 for the data-gradient product dh0 (the kernels rounded them at every load)
  const float* w_ih1[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}}; _Float16* w1h = nullptr; __bf16* w1b = nullptr;
  // optional (with w1b): the same bf16 values TRANSPOSED and direction-concatenated, w1bt[modality][n = 256][k = direction * 384 + g] -- both
  // operands of dh0 = sum_dir dgx_dir . W_ih_dir are then k-contiguous (gemm_tall.hip)
  __bf16* w1bt = nullptr;
  // optional: the packed operands THEMSELVES as 16-bit arrays instead of fp32 (same shapes and indices): xh = fp16 inputs (forward projection),
  // xb = bf16 inputs (the W_ih weight-gradient product, whose other operand is bf16), wh = fp16 weights.  All three or none; xpack / wpack are
  // then not written.
  _Float16* xh = nullptr; __bf16* xb = nullptr; _Float16* wh = nullptr;
};
int l0_pack(hipStream_t s, const L0Pack& a, bool pack_inputs, bool pack_weights = true);
struct L0Unpack {
  float* g_ih[2][2]; float* g_hh[2][2]; int d[2];
  float* dwih_pack; float* dwhh_pack;        // [2][2][384, KP], [2][2][384, 128]: read, added to the gradients, re-zeroed
  int KP;
};
int l0_unpack_grads(hipStream_t s, const L0Unpack& a);
// dst[r][0..Cp) = src[r][0..C) followed by zeros (Cp = C rounded up to a multiple of 4): the padded copy GemmDesc::a_pad4 asks for
int pad_rows(hipStream_t s, const float* src, float* dst, int R, int C, int Cp);
int seq_lengths2(hipStream_t s, const float* xa, int da, int* lens_a, const float* xv, int dv, int* lens_v, int B, int T);
int ln_relu_drop_fwd(hipStream_t s, const float* h2, const float* gamma, const float* beta, float* cube, float* mean,
                     float* rstd, int B, int T, int L, int K, int D, int slot, float p, RngKey key, uint32_t stream_id);
// ds[b,t,:] (gradient of the fwd+bwd sum) from dcube; accumulates dgamma/dbeta
int ln_relu_drop_bwd(hipStream_t s, const float* h2, const float* gamma, const float* beta, const float* mean,
                     const float* rstd, const float* dcube, float* ds, float* dgamma, float* dbeta, int B, int T, int L,
                     int K, int D, int slot, float p, RngKey key, uint32_t stream_id);

// feats[k][b,:] = mean_{t<T} cube[b,t,k,:]   (Model.py:466)   feats laid out [K][B][D]
int feat_mean_fwd(hipStream_t s, const float* cube, float* feats, int B, int T, int L, int K, int D);
// text_post_fwd (slot 0) + ln_relu_drop_fwd2 (slots 1, 2) + feat_mean_fwd of one forward tail as ONE launch (same arithmetic and
// dropout keys; feats = [3][B, D] = T_F, A_F, V_F)
int tail_pre_fwd(hipStream_t s, const float* tx_raw, float p_text, const LnSide2& a, const LnSide2& v, float* cube, float* feats, int B, int T,
                 int L, int K, int D, RngKey key);
// The same for BOTH forward tails of a shared-prefix step in one launch (round 5): every source row is read once and written to
// cube[0] / cube[1] under the dropout keys key.step + add[0] / add[1]; feats[o] = the temporal means of cube[o].  `part`: scratch of
// 2 * 3 * B * nchunk * D floats when tail_pre2_chunks() says nchunk > 1 (then a second small launch adds the chunk sums in order).
void tail_pre2_chunks(int B, int T, int* nchunk, int* rpc);
int tail_pre2_fwd(hipStream_t s, const float* tx_raw, float p_text, const LnSide2& a, const LnSide2& v, float* const cube[2], float* const feats[2],
                  const int add[2], float* part, int B, int T, int L, int K, int D, RngKey key);
// dcube[b,t,k,:] += dfeats[k][b,:]/T  for t<T
int feat_mean_bwd(hipStream_t s, const float* dfeats, float* dcube, int B, int T, int L, int K, int D);

// head: F_F[b,:] = compose_{l,k} x[b,l,k,:]; pred[b] = F_F.w + bias        (Model.py:489-515)
int head_fwd(hipStream_t s, const float* x, const float* w, const float* bias, float* ff, float* pred, int B, int L,
             int K, int D, int sum_l, int sum_k);
// dx[b,l,k,:] = scale*(dff_ext[b,:] + dpred[b]*w);  dw += sum_b dpred[b]*ff[b,:];  dbias += sum_b dpred[b]
// gather (optional): dff_ext is not read; the gradient of F_F is summed from its sources on the fly (the estimators' input gradients:
// the feature-gradient routing of the F slot folded into this kernel, one launch less on the chain)
int head_bwd(hipStream_t s, const float* dff_ext, const float* dpred, const float* w, const float* ff, float* dx,
             float* dw, float* dbias, int B, int L, int K, int D, int sum_l, int sum_k, const GatherSum* gather = nullptr);

// ---- LayerNorm along the LAST axis of [R, n] rows (D-axis mix)
int rowln_fwd(hipStream_t s, const float* y, const float* gamma, const float* beta, float* z, float* mean, float* rstd,
              long R, int n);
int rowln_bwd(hipStream_t s, const float* y, const float* gamma, const float* mean, const float* rstd, const float* dz,
              float* dy, float* dgamma, float* dbeta, long R, int n);
// ---- LayerNorm along the FIRST axis of per-sample [n, C] tiles (L-axis mix): stats per (b, c)
int colln_fwd(hipStream_t s, const float* y, const float* gamma, const float* beta, float* z, float* mean, float* rstd,
              int B, int n, int C);
int colln_param_grads(hipStream_t s, const float* y, const float* mean, const float* rstd, const float* dz, float* dgamma,
                      float* dbeta, int B, int n, int C);   // dgamma/dbeta only (the fused L-axis backward does the rest)
int colln_bwd(hipStream_t s, const float* y, const float* gamma, const float* mean, const float* rstd, const float* dz,
              float* dy, float* dgamma, float* dbeta, int B, int n, int C);

// ---- K-axis mix (tiny 3x3-class MLP + residual + LN over K), fully fused; x:[R,ik,D] -> z:[R,ok,D]
struct KMixW {
  const float *w1, *b1, *w2, *b2, *wr, *g, *be;   // b1/b2/wr may be null (no bias / identity residual)
  float *dw1, *db1, *dw2, *db2, *dwr, *dg, *dbe;  // gradient slots (backward only)
  int ik, hk, ok, act, ln_first;
  float drop_p; RngKey key; uint32_t stream_id;   // dropout_k on the MLP branch (MLPProcess.py:81,110)
  int dbg;                                         // debugging only, MIMRL_DBG_KMIX (1: skip the element loop, 2: skip the final atomics; parked MODE 2: 8 / 16 device-scope loads of
                                                   // x / dz, 32 check the LDS weights, 64 / 128 / 256 accumulate x / y / u in place of the LayerNorm-gain term)
};
#ifdef MIMRL_PHASE_PROBE
int kmix_bwd_read_phases(long long* out);   // 16 ticks, see model_ops.hip
int model_ops_read_phases(long long* out);  // 16 ticks: tail_pre_kernel, ln_relu_drop_bwd16_kernel
#endif
int kmix_fwd(hipStream_t s, const float* x, float* z, KMixW w, long R, int D);
int kmix_bwd(hipStream_t s, const float* x, const float* dz, float* dx, KMixW w, long R, int D);
int kmix_bwd_part(hipStream_t s, const float* x, const float* dz, float* dx, KMixW w, long R, int D, int part);   // 1: dx only, 2: parameter gradients only

// ---- reductions / misc
// out[z*os + n] += sum_m X[z*xs + m*ld + n]   for z < batch
int colsum(hipStream_t s, const float* X, long M, int N, long ld, float* out, int batch = 1, long xs = 0, long os = 0);
// out[r] += sum_b sum_c X[(b*R + r)*C + c]
int rowsum_batched(hipStream_t s, const float* X, int B, int R, int C, float* out);
// y += x (n elements)
int add_inplace(hipStream_t s, float* y, const float* x, long n);
// y = dropout(y) in place / dy *= mask
int dropout_inplace(hipStream_t s, float* y, long n, float p, RngKey key, uint32_t stream_id);

}  // namespace mimrl
