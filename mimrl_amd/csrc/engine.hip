// Engine: the two-stage training step of MIMRL orchestrated on one HIP stream (optionally replayed as hipGraphs).
//
//   stage 1  (Solver.py:205-214): model forward -> 5 MI + 6 CMI estimators -> backward into the CRITIC weights
//            -> value-clip + Adam on the critic bucket.
//   stage 2  (Solver.py:221-236): model forward (activations kept) -> estimators (data gradients only) -> MAE ->
//            backward through head / CubeMLP / LN / bi-GRU BPTT / W_t -> value-clip + Adam on the main bucket.
// The reference back-propagates stage 1 through the main model and stage 2 into the critic weights as well, but
// those gradients are never applied: they are skipped here with no effect on any parameter.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <functional>
#include <vector>
#include <dlfcn.h>

#include "cube_fused.h"
#include "mlp_fused.h"
#include "concat_fused.h"
#include "cube_bwd_fused.h"
#include "estimator_ops.h"
#include "gemm.h"
#include "gru.h"
#include "lstm.h"
#include "layout.h"
#include "comm.h"
#include "model_ops.h"

namespace mimrl {

namespace {

constexpr int H = 128, G = 384, HID = 256, EMB = 128;
constexpr int NE_MI = 5, NE_CMI = 6;
constexpr int ACT_SLACK = 4 * MLPF_MAX_WIDTH;   // floats behind every saved-activation buffer of the fused MLP stacks (MlpFusedArgs::act_slack)

// feature slots: F,T,A,V ; 4 = labels (C)
enum { FT_F = 0, FT_T = 1, FT_A = 2, FT_V = 3, FT_C = 4 };
const int kMiWire[NE_MI][2] = {{FT_F, FT_T}, {FT_F, FT_A}, {FT_F, FT_V}, {FT_T, FT_A}, {FT_T, FT_V}};   // Model.py:313-319
const int kCmiWire[NE_CMI][3] = {{FT_A, FT_C, FT_T}, {FT_T, FT_A, FT_C}, {FT_V, FT_C, FT_T},           // Model.py:323-339
                                 {FT_T, FT_V, FT_C}, {FT_T, FT_C, FT_A}, {FT_T, FT_C, FT_V}};
const char* kVmi[NE_MI] = {"f_t", "f_a", "f_v", "t_a", "t_v"};
const char* kVcmi[NE_CMI] = {"ac_t", "ta_c", "vc_t", "tv_c", "tc_a", "tc_v"};

__global__ void begin_stage_kernel(int* rng_step, int* adam_step, float* scalars, int scal_off, int scal_n) {
  if (threadIdx.x == 0) {
    *rng_step += 1;
    if (adam_step) *adam_step += 1;
  }
  for (int i = threadIdx.x; i < scal_n; i += blockDim.x) scalars[scal_off + i] = 0.f;
}

// MAE (nn.L1Loss, Solver.py:181-182) + its gradient
__global__ void mae_kernel(const float* __restrict__ pred, const float* __restrict__ y, float* __restrict__ dpred,
                           float* __restrict__ task, int B) {
  __shared__ float red[16];
  float s = 0.f;
  for (int b = threadIdx.x; b < B; b += blockDim.x) {
    const float d = pred[b] - y[b];
    s += fabsf(d);
    if (dpred) dpred[b] = (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) / B;
  }
  s = block_sum(s, red);
  if (threadIdx.x == 0) *task = s / B;
}

// Model.py:341 + Customization.py:100-102
__global__ void finalize_stage1_kernel(float* scal, const float* mi, const float* cmi, const float* bce, const float* coef1) {
  if (threadIdx.x != 0) return;
  float loss = 0.f;
  for (int e = 0; e < NE_MI; ++e) {
    scal[MIMRL_S1_MIS + e] = mi[e];
    scal[MIMRL_S1_LOSSES + e] = mi[NE_MI + e];        // mi_loss (= -mi except for the `mine` bound)
    loss += coef1[e] * mi[NE_MI + e];
  }
  for (int e = 0; e < NE_CMI; ++e) {
    scal[MIMRL_S1_MIS + NE_MI + e] = cmi[e];
    scal[MIMRL_S1_LOSSES + NE_MI + e] = bce[e];
    loss += coef1[NE_MI + e] * bce[e];
  }
  scal[MIMRL_S1_LOSS] = loss;
}
// Model.py:357,381-386 + Customization.py:109-111
__global__ void finalize_stage2_kernel(float* scal, const float* mi, const float* cmi, const float* coef2, int have_mi) {
  if (threadIdx.x != 0) return;
  const float task = scal[MIMRL_S2_TASK];
  if (!have_mi) {
    for (int i = 0; i < 8; ++i) { scal[MIMRL_S2_MIS + i] = 0.f; scal[MIMRL_S2_LOSSES + i] = 0.f; }
    scal[MIMRL_S2_LOSS] = task;
    return;
  }
  const float ac_t = cmi[0], ta_c = cmi[1], vc_t = cmi[2], tv_c = cmi[3], tc_a = cmi[4], tc_v = cmi[5];
  float v[8];
  v[0] = mi[0]; v[1] = mi[1]; v[2] = mi[2];
  v[3] = mi[3] + mi[4];
  v[4] = tc_a + tc_v - ta_c - tv_c;
  v[5] = ac_t - ta_c;
  v[6] = vc_t - tv_c;
  v[7] = ta_c + tv_c;
  float loss = task;
  for (int i = 0; i < 8; ++i) {
    const float li = i < 3 ? mi[NE_MI + i] : -v[i];   // Model.py:386: f_t, f_a, f_v through their mi_loss, the rest through -value
    scal[MIMRL_S2_MIS + i] = v[i];
    scal[MIMRL_S2_LOSSES + i] = li;
    loss += coef2[i] * li;
  }
  scal[MIMRL_S2_LOSS] = loss;
}

// Stage boundary of a combined two-stage step as ONE launch: finalize_stage1 (Model.py:341) + begin_stage(2) + MAE (Solver.py:181-182).
// Behind the critic Adam; the stage-1 raw terms are still intact (stage 2's estimators overwrite them later).
__global__ void stage_boundary_kernel(float* scal, const float* mi, const float* cmi, const float* bce, const float* coef1, int* rng_step,
                                      int* adam_step, const float* __restrict__ pred, const float* __restrict__ y,
                                      float* __restrict__ dpred, int B) {
  __shared__ float red[16];
  if (threadIdx.x == 0) {
    // every value is READ before the first store (tools/isa_lint.py: interleaved with the stores into `scal`, which may alias, these were
    // 21 loads each followed by s_waitcnt vmcnt(0) -- 21 round trips in a one-thread kernel on the step's critical path, 7 us)
    float vm[NE_MI], vl[NE_MI], vc[NE_CMI], vb[NE_CMI], c1[NE_MI + NE_CMI];
#pragma unroll
    for (int e = 0; e < NE_MI; ++e) { vm[e] = mi[e]; vl[e] = mi[NE_MI + e]; c1[e] = coef1[e]; }
#pragma unroll
    for (int e = 0; e < NE_CMI; ++e) { vc[e] = cmi[e]; vb[e] = bce[e]; c1[NE_MI + e] = coef1[NE_MI + e]; }
    const int rs = *rng_step, as = *adam_step;
    float loss = 0.f;
#pragma unroll
    for (int e = 0; e < NE_MI; ++e) {
      scal[MIMRL_S1_MIS + e] = vm[e];
      scal[MIMRL_S1_LOSSES + e] = vl[e];
      loss += c1[e] * vl[e];
    }
#pragma unroll
    for (int e = 0; e < NE_CMI; ++e) {
      scal[MIMRL_S1_MIS + NE_MI + e] = vc[e];
      scal[MIMRL_S1_LOSSES + NE_MI + e] = vb[e];
      loss += c1[NE_MI + e] * vb[e];
    }
    scal[MIMRL_S1_LOSS] = loss;
    *rng_step = rs + 1;
    *adam_step = as + 1;
  }
  for (int i = threadIdx.x; i < 32; i += blockDim.x) scal[32 + i] = 0.f;
  float s = 0.f;
  for (int b = threadIdx.x; b < B; b += blockDim.x) {
    const float d = pred[b] - y[b];
    s += fabsf(d);
    dpred[b] = (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) / B;
  }
  s = block_sum(s, red);
  __syncthreads();                       // the zeroing of scal[32..63] above is complete before the task loss lands in it
  if (threadIdx.x == 0) scal[MIMRL_S2_TASK] = s / B;
}

// roctx ranges (tracing): host-side ranges around the stages and their phases for `rocprofv3 --marker-trace`.  The library
// is looked up at run time (libroctx64.so ships with ROCm): no link dependency, no cost when it is absent.  Inside a replayed hipGraph a
// range brackets the graph launch; the phase ranges show up in the capture step and in eager (profile) steps.
struct Roctx {
  int (*push)(const char*) = nullptr;
  int (*pop)() = nullptr;
  Roctx() {
    // rocprofv3 records the ranges of rocprofiler-sdk's roctx; roctracer's libroctx64 is what older tools see
    void* h = dlopen("librocprofiler-sdk-roctx.so", RTLD_LAZY | RTLD_LOCAL);
    if (!h) h = dlopen("librocprofiler-sdk-roctx.so.1", RTLD_LAZY | RTLD_LOCAL);
    if (!h) h = dlopen("libroctx64.so", RTLD_LAZY | RTLD_LOCAL);
    if (!h) h = dlopen("libroctx64.so.4", RTLD_LAZY | RTLD_LOCAL);
    if (h) {
      push = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA"));
      pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
      if (!push || !pop) push = nullptr;
    }
  }
};
inline Roctx& roctx() { static Roctx r; return r; }
struct Range {
  bool on;
  explicit Range(const char* name) : on(roctx().push != nullptr) { if (on) roctx().push(name); }
  ~Range() { if (on) roctx().pop(); }
};

struct Lin { long w = -1, b = -1; int out = 0, in = 0; };
struct GruDirW { long w_ih = 0, w_hh = 0, b_ih = 0, b_hh = 0; int din = 0; };
struct AxisW { Lin fc1, fc2; long res = -1, ln_g = -1, ln_b = -1; int in = 0, hid = 0, out = 0; };
struct BlockW { AxisW ax[3]; };
struct MixBuf { float *xn = nullptr, *xn_mean = nullptr, *xn_rstd = nullptr, *u = nullptr, *h = nullptr, *y = nullptr,
                      *z = nullptr, *mean = nullptr, *rstd = nullptr; };
struct BlockBuf { MixBuf l, k, d; };

}  // namespace

}  // namespace mimrl

using namespace mimrl;

struct mimrl_handle {
  mimrl_cfg cfg;
  hipStream_t stream = nullptr;        // stream every launch goes to (the caller's, or cap_stream while capturing)
  hipStream_t user_stream = nullptr;   // the caller's stream (graphs are launched here)
  hipStream_t cap_stream = nullptr;    // private non-default stream: capture is illegal on the legacy default stream
  Layout layout;
  mimrl_buffers bufs;
  bool bound = false;
  bool grads_clean[3] = {true, true, true};   // bucket known to be all-zero (fresh buffers / zeroed by the fused Adam)
  int bank_rows = 0;
  bool bf16 = false;                   // current GEMM operand mode (switched between forward / backward sections)
  int prec = 0;                        // MIMRL_PREC_* bit mask

  // parameter handles
  GruDirW gru[2][2][2];          // [mod a=0,v=1][layer][dir]
  long ln_g[2], ln_b[2], w_t;
  BlockW blk[MIMRL_MAX_BLOCKS];
  long cls_w, cls_b;
  // bf16 images of the critic bucket for the fused estimator stacks: straight (kept fresh by the critic Adam launch, rebuilt
  // after mimrl_bind / mimrl_params_changed) and per-matrix transposed (rebuilt beside every estimator forward pass)
  __bf16 *crit_img = nullptr, *crit_imgT = nullptr;
  // MFMA-fragment-order images of the stacks mlp_frag_kernel takes (mlp_fused.h): forward product and data-gradient product, ONE launch
  // for both; valid exactly when crit_img is (rebuilt by ensure_images and behind every critic Adam launch)
  __bf16 *crit_frag = nullptr, *crit_fragT = nullptr;
  FragTable ftab;
  bool frag_side_pending = false;      // the refresh behind the critic Adam runs on side 3 and has not been joined yet
  bool img_valid = false;
  unsigned knn_ovr_mask[2] = {0u, 0u};  // per stage: CMI calls whose neighbour rows come from bufs.knn_override
  bool knn_pre = true;                 // prefetch mode: stage 2's kNN sampling also runs inside stage 1, beside the encoder prefix (MIMRL_NO_KNN_PREFETCH=1: off)
  bool mi_fused_bwd_done = false;      // mi_forward already produced the tower-output gradients (mi_sep_fused)
  bool imgT_ready = false;             // a transposed-image refresh has been issued for the estimator pass being enqueued
  TransposeTable ttab;
  int ensure_images() {
    if (img_valid || !crit_img) return MIMRL_OK;
    MX(bf16_image(user_stream, bufs.crit_p, crit_img, layout.floats[MIMRL_GROUP_CRITIC]));
    if (crit_frag && ftab.n > 0) MX(bf16_frag_images(user_stream, bufs.crit_p, crit_frag, ftab));
    img_valid = true;
    return MIMRL_OK;
  }
  // log-baseline of tuba / interpolate (VMI.py:72-110): per-estimator vector over the y rows, pitch 2B (the y operand of
  // estimator e is slot 2e+1 of the tower-input buffer), its gradient, and for the trainable baseline an MLP 128-256-256-256-1
  long bl0 = 0, bl_stride = 0, bl_l[4][2];
  float *lbv = nullptr, *dlbv = nullptr, *bact[3] = {nullptr, nullptr, nullptr}, *bdz[3] = {nullptr, nullptr, nullptr}, *bdin = nullptr;
  bool has_baseline() const {
    return cfg.baseline_type != MIMRL_BASELINE_CONSTANT && (cfg.bound_type == MIMRL_BOUND_TUBA || cfg.bound_type == MIMRL_BOUND_INTERPOLATE);
  }
  int baseline_forward();
  int baseline_backward(int stage);
  long tower0 = 0, tower_stride = 0;   // critic bucket: first tower, distance between consecutive towers
  long tower_l[4][2];                  // per-layer (w,b) offsets relative to tower0
  long cmi0 = 0, cmi_stride = 0, cmi_l[4][2];

  // workspace
  char* ws = nullptr;
  size_t ws_bytes = 0, ws_used = 0;
  int* d_ints = nullptr;               // [0] rng step, [1] adam main step, [2] adam critic step (= bufs.counters when the caller owns them)
  int* d_ints_own = nullptr;           // private fallback storage
  float* d_consts = nullptr;           // coef1[11] coef2[8] gs_mi[2][5] g_bce[2][6] g_cmi[2][6]
  int *lens[2] = {nullptr, nullptr};
  bool kmix_pg_on_side3 = false;       // K-axis parameter-gradient kernels are in flight on side 3: the BPTT waits for them
  bool begin_in_pack = false;          // the stage-1 begin-of-stage bookkeeping is owed by the next layer-0 pack launch
  hipEvent_t ev_lens = nullptr;        // set while the length scan of this forward pass runs on side 0 (in front of the text projection)
  float *tx_raw = nullptr, *gx[2][2], *h0[2], *h1[2], *sv[2][2][2], *ln_mean[2], *ln_rstd[2];
  float* cube0 = nullptr;
  // layer-0 GRU operands in a common aligned shape (model_ops.h: L0Pack): one batched input projection, two batched weight gradients
  float *xpack = nullptr, *wpack = nullptr, *bpack = nullptr, *dwih_pack = nullptr, *dwhh_pack = nullptr;
  // 16-bit operands of the layer-1 input projection and of the dh0 product (round 4: those GEMMs are bound by L2 -> LDS operand bytes):
  // h0h = fp16 copy of the layer-0 outputs written by the recurrence kernel itself (per forward set), w1h / w1b = fp16 / bf16 images of
  // the four W_ih_l1 written by the layer-0 pack launch of the same forward pass
  _Float16* h0h[2] = {nullptr, nullptr}; _Float16* w1h = nullptr; __bf16* w1b = nullptr;
  __bf16* w1bt = nullptr;              // w1b transposed + direction-concatenated [modality][256][768]: B operand of the tall (k-contiguous) dh0 product
  bool h16_on = true;                  // MIMRL_NO_H16=1: fp32 operands as before (tuning knob; results are bit-identical either way)
  float* w2p[MIMRL_MAX_BLOCKS] = {};   // unfused L axis: fc2 [ol, hl] copied to row pitch roundup4(hl) when hl % 4 != 0 (GemmDesc::a_pad4)
  bool w2p_valid[MIMRL_MAX_BLOCKS] = {};   // ... holds the current parameters (set by the forward pass, cleared by the main update)
  bool xin_on = true;                  // MIMRL_NO_XIN=1: the layer-0 input projection as its own GEMM (tuning knob)
  bool part0_done = false;             // mimrl_stage_grads_part(h, 2, 0) ran on the bound batch and nothing since: part 1 may follow
  bool l0_xin = false;                 // this step's layer-0 forward ran the fused-projection (8-wave) kernel: its BPTT launch must match
  bool xpack16 = false;                // the packed layer-0 operands of this step are the 16-bit arrays (set by the forward pass)
  bool w1_img_valid = false;           // w1b holds the CURRENT main parameters (set by the forward pass, cleared by the main update)
  int KP() const { return ((cfg.d_a > cfg.d_v ? cfg.d_a : cfg.d_v) + 15) & ~15; }
  bool dg_bf16 = false;                // BPTT outputs dg / h_prev stored as bf16 (GRU encoders, bf16 recurrence + bf16 backward GEMMs; MIMRL_DG_FP32=1: off)
  bool concat_compact = false;         // the last concat forward saved bitmasks (+ bf16 values) for the fused backward, not fp32 activations
  bool fused_concat = true;            // concat critic forward as one launch (concat_fused.hip); MIMRL_NO_FUSED_CONCAT=1 at create time
  bool l0_packed = false;              // see mimrl_create
  bool l0_bwd_pack = false;            // small batches: only the INPUTS are packed (off the chain) and only the weight gradients use them
  BlockBuf bb[MIMRL_MAX_BLOCKS];
  __bf16* wtT[MIMRL_MAX_BLOCKS][3] = {};   // transposed bf16 images of the D-axis weights (fc2, fc1, res) for the fused backward
  // Second set of forward buffers.  In prefetch mode (mimrl_set_stage2_prefetch) stage 1 runs its forward pass and its
  // estimators on this set while the stage-2 forward pass of the SAME batch (same main parameters: stage 1 only
  // touches the critics) runs beside it on `pre_stream` into the primary set, which the stage-2 backward then reads.
  struct FwdSet {
    int* lens[2] = {nullptr, nullptr};
    float *tx_raw = nullptr, *gx[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}}, *h0[2] = {nullptr, nullptr}, *h1[2] = {nullptr, nullptr};
    float *ln_mean[2] = {nullptr, nullptr}, *ln_rstd[2] = {nullptr, nullptr}, *cube0 = nullptr, *feats = nullptr, *pred = nullptr;
    _Float16* h0h[2] = {nullptr, nullptr};
    BlockBuf bb[MIMRL_MAX_BLOCKS];
  } alt;
  void swap_fwd_set() {
    for (int m = 0; m < 2; ++m) {
      std::swap(lens[m], alt.lens[m]); std::swap(h0[m], alt.h0[m]); std::swap(h1[m], alt.h1[m]); std::swap(h0h[m], alt.h0h[m]);
      std::swap(ln_mean[m], alt.ln_mean[m]); std::swap(ln_rstd[m], alt.ln_rstd[m]);
      for (int d = 0; d < 2; ++d) std::swap(gx[m][d], alt.gx[m][d]);
    }
    std::swap(tx_raw, alt.tx_raw); std::swap(cube0, alt.cube0);
    std::swap(bufs.feats, alt.feats); std::swap(bufs.pred, alt.pred);
    for (int i = 0; i < MIMRL_MAX_BLOCKS; ++i) std::swap(bb[i], alt.bb[i]);
  }
  bool prefetch = false;               // mode switch (mimrl_set_stage2_prefetch)
  bool defer_tail = false;             // prefetch mode 2: the stage-2 forward tail is NOT issued beside stage 1 but by
                                       // mimrl_stage2_forward_tail (the caller runs it under the stage-1 gradient all-reduce)
  bool tail2_needed = false;           // deferred tail still to be issued before stage 2 may run
  bool fwd2_pending = false;           // a prefetched stage-2 forward is waiting to be consumed
  float grad_scale = 1.f;              // folded into the fused clip+Adam (mimrl_set_grad_scale)
  // data parallel (round 5): an RCCL communicator of this handle's own (mimrl_set_comm).  With it every update pass of the handle --
  // mimrl_stage{1,2}_step, mimrl_two_stage_step, captured or not -- all-reduces (SUM) the stage's gradient bucket between the gradient
  // pass and the fused clip + Adam, on the engine's own streams: the collectives are nodes of the captured step graph, and the main
  // bucket travels in two pieces -- [0, late_offset) on `comm_s` under the layer-0 BPTT, the layer-0 tail behind it.
  void* comm = nullptr; int comm_world = 1, comm_rank = 0;
  hipStream_t comm_s = nullptr;
  bool comm_split = true;              // MIMRL_DDP_SPLIT=0: the main bucket in one piece behind the whole backward pass
  KernelStamp kstamp;                  // launch stamps of the recurrence kernels (mimrl_set_kernel_stamps); id: 0/1 forward layer 0/1, 2/3 BPTT layer 1/0

  int run_fwd2_tail();
  hipStream_t pre_stream = nullptr;
  int carve_fwd(size_t* gmax_out);
  float *ff = nullptr, *dpred = nullptr;
  // estimators
  float *tin = nullptr, *ta[3], *tout = nullptr, *scores = nullptr, *dscores = nullptr;
  float *cP = nullptr, *cQ = nullptr, *ca[3];
  int split_part = 0;                  // 1: encoders_backward stops behind the layer-1 weight gradients (mimrl_stage_grads_part); 0: whole pass
  std::function<int()> pending_text;   // MIMRL_TEXT_LATE: the text branch captured behind the layer-0 input projection (1) / recurrence (2)
  int pending_text_at = 0;
  bool fold_unpack = false, unpack_pending = false;   // the packed layer-0 GRU weight gradients are scattered by the Adam launch (AdamArgs::fold)
  bool fold_unpack_on = true;          // MIMRL_NO_FOLD_UNPACK=1: keep the separate scatter kernel (tuning knob)
  bool gx_f16 = false;                 // the hoisted GRU input projections gx[B,T,3H] are stored as fp16 (long sequences, bf16 mode: create)
  int *knn_idx = nullptr, *knn_idx2 = nullptr;   // neighbour indices; stage 2 has its own set (prefetch mode samples it early)
  char* knn_scr[2] = {nullptr, nullptr};         // candidate lists of the MFMA kNN (knn_mfma.hip), one per stage
  size_t knn_scr_bytes = 0;
  float *cmi_in = nullptr, *cc[3], *logits = nullptr, *dlogits = nullptr;
  float *mi_raw = nullptr, *cmi_raw = nullptr, *bce_raw = nullptr;
  // backward temporaries
  float *dfeat = nullptr, *dtout = nullptr, *dta[3], *dtin = nullptr, *dca[3], *dP = nullptr, *dQ = nullptr;
  float *dcc[3], *dcin = nullptr;
  static constexpr int NGBUF = 24;   // cube backward: rotating (4 in use) or one-shot (deferred weight gradients)
  float* gbuf[NGBUF];
  size_t gbuf_floats = 0;
  float *dtx = nullptr, *ds[2], *dg[2][2][2], *hprev[2][2][2], *dh0[2];   // [layer][mod][dir]; dg = [dr'|dz'|dn'|dn'r] rows of 4H

  // side streams: independent branches of a stage run concurrently (and are captured as parallel graph branches)
  static constexpr int NSIDE = 6;      // 0: text branch, 1-3: per-(modality,direction) helpers / weight gradients, 4: kNN, 5: CMI
  hipStream_t side[NSIDE] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  std::vector<hipEvent_t> ev_pool;
  size_t ev_next = 0;
  bool multi_stream = true;
  bool fused_cube_bwd = true;          // bf16 mode: per-axis fused data-gradient chains of CubeMLP (MIMRL_NO_FUSED_CUBE_BWD=1 disables)
  bool fused_mlp = true;               // bf16 mode: estimator MLP stacks as one kernel per direction (MIMRL_NO_FUSED_MLP=1 disables)
  bool fused_cube = true;              // bf16 mode: CubeMLP blocks as one LDS-resident kernel (MIMRL_NO_FUSED_CUBE=1 disables)
  int fwd_f16 = 1;                     // bf16 mode: the forward products in front of / inside CubeMLP round their operands to FP16, not bf16
                                       // (GemmDesc::f16, cube_fused.hip: main-gradient cosine vs fp32 0.964 -> 0.998); MIMRL_FWD_BF16=1: off
  int next_event(hipEvent_t* e) {
    if (ev_next == ev_pool.size()) {
      hipEvent_t n;
      HIPX(hipEventCreateWithFlags(&n, hipEventDisableTiming));
      ev_pool.push_back(n);
    }
    *e = ev_pool[ev_next++];
    return MIMRL_OK;
  }
  int G_group(hipStream_t st, const GemmDesc* ds, int n) {
    if (!prof_on) return gemm_group(st, ds, n, bf16);
    for (int i = 0; i < n; ++i) MX(G_on(st, ds[i]));      // profiling: one event pair per product
    return MIMRL_OK;
  }
  unsigned side_mask = ~0u;            // sides that may be used right now; work for a masked-out side goes to `stream`
  bool side_on(int i) const { return multi_stream && ((side_mask >> i) & 1u); }
  hipStream_t S(int i) const { return side_on(i) ? side[i] : stream; }
  // side[lo..hi] wait for everything enqueued on `stream` so far
  int fork(int lo, int hi) {
    if (!multi_stream) return MIMRL_OK;
    hipEvent_t e = nullptr;
    for (int i = lo; i <= hi; ++i) {
      if (!side_on(i)) continue;
      if (!e) { MX(next_event(&e)); HIPX(hipEventRecord(e, stream)); }
      HIPX(hipStreamWaitEvent(side[i], e, 0));
    }
    return MIMRL_OK;
  }
  // `stream` waits for side[lo..hi]
  int join(int lo, int hi) {
    if (!multi_stream) return MIMRL_OK;
    for (int i = lo; i <= hi; ++i) {
      if (!side_on(i)) continue;
      hipEvent_t e;
      MX(next_event(&e));
      HIPX(hipEventRecord(e, side[i]));
      HIPX(hipStreamWaitEvent(stream, e, 0));
    }
    return MIMRL_OK;
  }
  // side[i] waits for side[j]
  int chain(int i, int j) {
    if (!multi_stream || S(i) == S(j)) return MIMRL_OK;
    hipEvent_t e;
    MX(next_event(&e));
    HIPX(hipEventRecord(e, S(j)));
    HIPX(hipStreamWaitEvent(S(i), e, 0));
    return MIMRL_OK;
  }
  // all-reduce of a stage's whole gradient bucket on `stream` (no communicator: nothing)
  int reduce_bucket(int stage) {
    if (!comm) return MIMRL_OK;
    Range rg(stage == 1 ? "mimrl.stage1.allreduce(crit_g) [RCCL]" : "mimrl.stage2.allreduce(main_g) [RCCL]");
    return comm_allreduce_sum(comm, stage == 1 ? bufs.crit_g : bufs.main_g, (size_t)layout.floats[stage == 1 ? MIMRL_GROUP_CRITIC : MIMRL_GROUP_MAIN], stream);
  }
  // stage 2 with a communicator: the gradient pass in two parts with the early range of the main bucket in flight under the second
  int enqueue_grads2_reduced(bool skip_zero);
  // GEMM family accounting of the phase profiler: HIP events on the launch stream around every gemm() of an eager step,
  // with the algorithmic FLOPs / bytes of the launch (operands and output counted once)
  struct GemmProf { hipEvent_t a, b; double flops, bytes; };
  std::vector<GemmProf> prof_gemm;
  int G_on(hipStream_t st, const GemmDesc& d) {
    if (!prof_on) return gemm(st, d, bf16);
    GemmProf g;
    if (!prof_pool.empty()) { g.a = prof_pool.back().first; g.b = prof_pool.back().second; prof_pool.pop_back(); }
    else { HIPX(hipEventCreate(&g.a)); HIPX(hipEventCreate(&g.b)); }
    auto distinct = [&](long s_b, long s_bo) -> double {
      if (d.batch_in > 0) return (double)(s_bo != 0 ? d.batch / d.batch_in : 1) * (s_b != 0 ? d.batch_in : 1);
      return s_b != 0 ? d.batch : 1;
    };
    g.flops = 2.0 * d.M * d.N * ((double)d.K + (d.A2 ? d.K2 : 0)) * d.batch;
    const double ea = d.a_bf16 ? 2.0 : 4.0, eb = d.b_bf16 ? 2.0 : 4.0;
    g.bytes = ea * d.M * d.K * distinct(d.sa_b, d.sa_bo) + eb * d.K * d.N * distinct(d.sb_b, d.sb_bo) +
              4.0 * d.M * d.N * distinct(d.sc_b, d.sc_bo) * ((d.beta != 0.f || d.atomic) ? 2 : 1);
    if (d.A2) g.bytes += ea * d.M * d.K2 * (d.sa2_b ? d.batch : 1) + eb * d.K2 * d.N * (d.sb2_b ? d.batch : 1);
    HIPX(hipEventRecord(g.a, st));
    const int r = gemm(st, d, bf16);
    HIPX(hipEventRecord(g.b, st));
    prof_gemm.push_back(g);
    return r;
  }

  // phase profiler
  bool prof_on = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_ev[MIMRL_NPHASES];
  std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_pool;
  struct Scope {
    mimrl_handle* h; int ph; hipEvent_t a = nullptr, b = nullptr;
    Scope(mimrl_handle* h_, int ph_) : h(h_), ph(ph_) {
      if (!h->prof_on) return;
      if (!h->prof_pool.empty()) { a = h->prof_pool.back().first; b = h->prof_pool.back().second; h->prof_pool.pop_back(); }
      else { (void)hipEventCreate(&a); (void)hipEventCreate(&b); }
      (void)hipEventRecord(a, h->stream);
    }
    ~Scope() {
      if (!a) return;
      (void)hipEventRecord(b, h->stream);
      h->prof_ev[ph].push_back({a, b});
    }
  };

  // graphs: [stage 1|2][kind: 0 = step (grads+apply), 1 = grads only]
  // The captured graphs bake the input addresses in, so they are cached PER INPUT SET: the caller may alternate between two
  // sets of (text, audio, video, labels) buffers (mimrl_set_inputs) -- the next batch is uploaded into the idle set while the
  // step runs on the active one, and switching costs no device work.
  struct GraphSet {
    hipGraphExec_t graph[3][4] = {{nullptr, nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr, nullptr}};   // [..][2], [..][3]: the two halves of a split stage-2 gradient pass
    int rows[3][4] = {{-1, -1, -1, -1}, {-1, -1, -1, -1}, {-1, -1, -1, -1}};
    hipGraphExec_t tail = nullptr; int tail_rows = -1;
    const void* in[4] = {nullptr, nullptr, nullptr, nullptr};
  } gsets[2];
  int cur_set = 0;
  GraphSet& GS() { return gsets[cur_set]; }
  // Invalidated graphs are RETIRED, not destroyed: hipGraphExecDestroy followed by instantiating and launching new graphs crashes
  // this HIP runtime in ~7-14 % of fresh processes (SIGSEGV in hip::Graph::UpdateStreams under hipGraphLaunch of the NEW graph:
  // rocgdb backtrace in DESIGN.md section 8; 3/40 and 7/50 runs of `bench.py --extras-only`, whose first action is a mode switch).
  // A retired exec is a few hundred bytes of host state per node; invalidation happens on mode switches and when the bank size
  // changes (once per training run), so the list stays short.  They are released with the handle.
  std::vector<hipGraphExec_t> retired;
  void retire(hipGraphExec_t& ex) { if (ex) { retired.push_back(ex); ex = nullptr; } }
  void drop_graphs(int set = -1) {
    for (int q = 0; q < 2; ++q) {
      if (set >= 0 && q != set) continue;
      for (int s = 0; s <= 2; ++s)
        for (int k = 0; k < 4; ++k) retire(gsets[q].graph[s][k]);
      retire(gsets[q].tail);
    }
  }
  float* P(long off) const { return bufs.main_p + off; }
  float* Gm(long off) const { return bufs.main_g + off; }
  float* CP(long off) const { return bufs.crit_p + off; }
  float* CG(long off) const { return bufs.crit_g + off; }
  long conv_w[2] = {0, 0}, conv_b[2] = {0, 0};   // --encoders conv: Conv1d weight [D, d, 3] / bias offsets (audio, video)
  int encoders_forward(bool save, int knn_stage);
  int conv_forward(int knn_stage);
  int conv_backward();
  int lstm_encoders_forward(bool save, int knn_stage);
  int lstm_encoders_backward();
  int rng_add = 0;
  RngKey key() const { return RngKey{(uint32_t)cfg.seed, (uint32_t)(cfg.seed >> 32), d_ints, rng_add}; }
  const float* coef1() const { return d_consts; }
  const float* coef2() const { return d_consts + 11; }
  const float* gs_mi(int stage) const { return d_consts + 19 + (stage - 1) * 5; }
  const float* g_bce(int stage) const { return d_consts + 29 + (stage - 1) * 6; }
  const float* g_cmi(int stage) const { return d_consts + 41 + (stage - 1) * 6; }
  int nprod() const { return (cfg.batch / cfg.k_neighbor) * cfg.k_neighbor; }   // rows of the product batch (Model.py:79)
  int m_anchor() const { return cfg.batch / cfg.k_neighbor; }

  int resolve();
  int alloc_workspace();
  template <typename T>
  int take(T** p, size_t count) {
    const size_t bytes = (count * sizeof(T) + 255) / 256 * 256;
    if (ws) {
      if (ws_used + bytes > ws_bytes) return set_error(MIMRL_ERR_STATE, "workspace overflow");
      *p = reinterpret_cast<T*>(ws + ws_used);
    }
    ws_used += bytes;
    return MIMRL_OK;
  }
  int carve();

  int G_(const GemmDesc& d) { return G_on(stream, d); }
  // diagnosis of bf16 fidelity (tools/bf16_diag.py): MIMRL_FWD_FP32_SITES=<mask> runs single forward sites with fp32 operands although the
  // precision mode says bf16 -- 1: W_t projection, 2: GRU layer-0 input projections, 4: layer-1 input projections, 8: estimator stacks.
  // Results stay valid (only more precise); tuning knob.
  static bool fp32_site(int bit) {
    static const int mask = knob("MIMRL_FWD_FP32_SITES") ? atoi(knob("MIMRL_FWD_FP32_SITES")) : 0;
    return (mask & bit) != 0;
  }
  struct PrecGuard {   // run a scope with fp32 GEMM operands
    mimrl_handle* h; bool saved;
    PrecGuard(mimrl_handle* h_, bool force_fp32) : h(h_), saved(h_->bf16) { if (force_fp32) h->bf16 = false; }
    ~PrecGuard() { h->bf16 = saved; }
  };
  // weight-gradient work parked by cube_backward and issued on the side streams once the data-gradient chain is through
  // (it then overlaps the latency-bound GRU BPTT instead of competing with the chain for CUs and L2)
  struct Deferred { int kind; int side; GemmDesc g; const float* src; long n0, n1, n2, n3; float* dst;
                    const float *p1 = nullptr, *p2 = nullptr, *p3 = nullptr; float* dst2 = nullptr; KMixW kw = KMixW();
                    const float *p4 = nullptr, *p5 = nullptr; float *dst3 = nullptr, *dst4 = nullptr; };
  std::vector<Deferred> deferred;
  int flush_deferred(int only_side = 0, hipEvent_t after = nullptr);
  int wg_helper = -1;                  // side stream that takes every second weight-gradient GEMM of an MLP stack (-1: none)
  int dbg_delay(hipStream_t st, int tag);   // critical-path probe (MIMRL_DBG_DELAY_TAG / _US): a spin kernel behind one phase
  int model_forward(bool train, bool save, int knn_stage = 0, int part = 0);   // part: 0 all, 1 prefix, 2 tail
  int cube_forward(bool train, bool save);
  int cube_backward(int cur_in, int* cur_out);
  int wt_images(hipStream_t st, bool bwd_bf16, bool launch, bool* d_fused);
  int model_backward();
  int encoders_backward(float* dcube);
  int gru_layer_backward(int l);
  struct StreamGuard {   // route every launch of a scope to another stream
    mimrl_handle* h; hipStream_t saved;
    StreamGuard(mimrl_handle* h_, hipStream_t st) : h(h_), saved(h_->stream) { h->stream = st; }
    ~StreamGuard() { h->stream = saved; }
  };
  int knn_launch(int stage, hipStream_t st);
  int mi_forward(int stage, bool want_grad);
  int cmi_forward(int stage, bool want_grad);
  int mi_backward(int stage);
  int cmi_backward(int stage);
  int route_feature_grads();
  GatherSum head_gather;               // sources of the F_F gradient (summed inside head_bwd) while head_gather_on
  bool head_gather_on = false;
  hipEvent_t ev_dmean = nullptr;       // T / A / V feature gradients ready (gathered on side 0)
  hipEvent_t ev_pre = nullptr;         // MIMRL_BPTT_FIRST: the point the parked kernels are flushed behind (encoders_backward -> gru_layer_backward)
  int estimators_all(int stage, bool want_grad, bool backward);
  // grouped MLP stacks living in the critic bucket (nb groups, uniform parameter stride `pstride`)
  int mlp_stack_forward(int nb, int rows, int brows, long p0, long pstride, int nl, const long (*l_off)[2], const int* dims,
                        const float* in, float* const* act, float* out);
  int mlp_stack_backward(int nb, int rows, int brows, long p0, long pstride, int nl, const long (*l_off)[2],
                         const int* dims, const float* in, float* const* act, float* dout, float* const* dtmp, float* din,
                         bool wgrad);
  int enqueue_grads(int stage, bool skip_zero = false);
  int enqueue_apply(int stage);
  int run(int stage, int kind);
  int run_step();                      // both stages as ONE captured graph where possible (mimrl_two_stage_step)
  bool keep_events = false;            // second stage of a combined capture: do not recycle the first stage's events
  // combined two-stage capture (run_step) only -- the state is provably periodic there:
  bool fuse_boundary = false;          // finalize_stage1 + begin_stage(2) + mae as ONE launch behind the critic Adam
  bool imgT_valid = false;             // transposed critic images match the critic parameters (refreshed once per step, behind Adam_vmi)
  bool skip_imgT_refresh = false;      // stage 1 of a combined step: the images built in the previous step's stage 2 are current
  bool wtT_prebuilt = false;           // D-axis weight images for the CubeMLP backward are built beside the encoders (off the chain) ...
  bool wtT_built = false;              // ... and that launch has been captured (only the shared-prefix path issues it)
};

// =================================================================================================
int mimrl_handle::resolve() {
  auto off = [&](const std::string& n, long* o) -> int {
    const LayoutEntry* e = layout.find(n);
    if (!e) return set_error(MIMRL_ERR_STATE, "layout: missing tensor %s", n.c_str());
    *o = e->offset;
    return MIMRL_OK;
  };
  auto opt = [&](const std::string& n) -> long {
    const LayoutEntry* e = layout.find(n);
    return e ? e->offset : -1;
  };
  const char* modn[2] = {"rnn_a", "rnn_v"};
  const int dmod[2] = {cfg.d_a, cfg.d_v};
  if (cfg.encoder == MIMRL_ENCODER_CONV) {
    MX(off("conv_a.weight", &conv_w[0])); MX(off("conv_a.bias", &conv_b[0]));
    MX(off("conv_v.weight", &conv_w[1])); MX(off("conv_v.bias", &conv_b[1]));
  }
  const int rnn_layers = cfg.encoder == MIMRL_ENCODER_GRU ? 2 : cfg.encoder == MIMRL_ENCODER_LSTM ? 1 : 0;
  for (int m = 0; m < 2; ++m)
    for (int l = 0; l < rnn_layers; ++l)
      for (int d = 0; d < 2; ++d) {
        const std::string sfx = "_l" + std::to_string(l) + (d ? "_reverse" : "");
        GruDirW& g = gru[m][l][d];
        MX(off(std::string(modn[m]) + ".weight_ih" + sfx, &g.w_ih));
        MX(off(std::string(modn[m]) + ".weight_hh" + sfx, &g.w_hh));
        MX(off(std::string(modn[m]) + ".bias_ih" + sfx, &g.b_ih));
        MX(off(std::string(modn[m]) + ".bias_hh" + sfx, &g.b_hh));
        g.din = l == 0 ? dmod[m] : 2 * H;
      }
  MX(off("ln_a.weight", &ln_g[0])); MX(off("ln_a.bias", &ln_b[0]));
  MX(off("ln_v.weight", &ln_g[1])); MX(off("ln_v.bias", &ln_b[1]));
  MX(off("W_t.weight", &w_t));
  int din[3] = {cfg.time_len, 3, cfg.d_common};
  const char axn[3] = {'l', 'k', 'd'};
  for (int i = 0; i < cfg.n_blocks; ++i) {
    const std::string pre = "mlp_encoder.layers_stack." + std::to_string(i);
    for (int ax = 0; ax < 3; ++ax) {
      AxisW& a = blk[i].ax[ax];
      a.in = din[ax]; a.hid = cfg.d_hiddens[i][ax]; a.out = cfg.d_outs[i][ax];
      const std::string m = pre + ".mlp_" + axn[ax];
      MX(off(m + ".fc1.weight", &a.fc1.w)); a.fc1.b = opt(m + ".fc1.bias"); a.fc1.out = a.hid; a.fc1.in = a.in;
      MX(off(m + ".fc2.weight", &a.fc2.w)); a.fc2.b = opt(m + ".fc2.bias"); a.fc2.out = a.out; a.fc2.in = a.hid;
      MX(off(pre + ".ln_" + axn[ax] + ".weight", &a.ln_g));
      MX(off(pre + ".ln_" + axn[ax] + ".bias", &a.ln_b));
      a.res = opt(pre + ".res_projection_" + axn[ax] + ".weight");
    }
    for (int ax = 0; ax < 3; ++ax) din[ax] = cfg.d_outs[i][ax];
  }
  MX(off("classifier.0.weight", &cls_w)); MX(off("classifier.0.bias", &cls_b));
  const int idx4[4] = {0, 2, 4, 6};
  // critic towers: uniform stride between consecutive towers (layout order is [estimator][tower][layer])
  {
    const bool sep = cfg.critic_type == MIMRL_CRITIC_SEPARATE;
    const std::string t0 = std::string("vmi_estimator_f_t.critic_model.") + (sep ? "MLP_g" : "MLP_f");
    const std::string t1 = sep ? "vmi_estimator_f_t.critic_model.MLP_h" : "vmi_estimator_f_a.critic_model.MLP_f";
    long a, b;
    MX(off(t0 + ".0.weight", &a)); MX(off(t1 + ".0.weight", &b));
    tower0 = a; tower_stride = b - a;
    for (int l = 0; l < 4; ++l) {
      long w, bb_;
      MX(off(t0 + "." + std::to_string(idx4[l]) + ".weight", &w));
      MX(off(t0 + "." + std::to_string(idx4[l]) + ".bias", &bb_));
      tower_l[l][0] = w - tower0; tower_l[l][1] = bb_ - tower0;
    }
    // verify uniformity
    const int ntw = sep ? 10 : 5;
    for (int t = 0; t < ntw; ++t) {
      const int e = sep ? t / 2 : t;
      const std::string nm = std::string("vmi_estimator_") + kVmi[e] + ".critic_model." +
                             (sep ? (t % 2 ? "MLP_h" : "MLP_g") : "MLP_f") + ".0.weight";
      long o; MX(off(nm, &o));
      if (o != tower0 + t * tower_stride) return set_error(MIMRL_ERR_STATE, "critic towers are not uniformly strided");
    }
  }
  {
    long a, b;
    MX(off("vcmi_estimator_ac_t.classifier.mlp.0.weight", &a));
    MX(off("vcmi_estimator_ta_c.classifier.mlp.0.weight", &b));
    cmi0 = a; cmi_stride = b - a;
    for (int l = 0; l < 4; ++l) {
      long w, bb_;
      MX(off("vcmi_estimator_ac_t.classifier.mlp." + std::to_string(idx4[l]) + ".weight", &w));
      MX(off("vcmi_estimator_ac_t.classifier.mlp." + std::to_string(idx4[l]) + ".bias", &bb_));
      cmi_l[l][0] = w - cmi0; cmi_l[l][1] = bb_ - cmi0;
    }
    for (int e = 0; e < NE_CMI; ++e) {
      long o; MX(off(std::string("vcmi_estimator_") + kVcmi[e] + ".classifier.mlp.0.weight", &o));
      if (o != cmi0 + e * cmi_stride) return set_error(MIMRL_ERR_STATE, "CMI classifiers are not uniformly strided");
    }
  }
  if (cfg.baseline_type == MIMRL_BASELINE_UNNORMALIZED) {
    long a, b;
    MX(off("vmi_estimator_f_t.baseline_model.MLP.0.weight", &a));
    MX(off("vmi_estimator_f_a.baseline_model.MLP.0.weight", &b));
    bl0 = a; bl_stride = b - a;
    for (int l = 0; l < 4; ++l) {
      long w, bb_;
      MX(off("vmi_estimator_f_t.baseline_model.MLP." + std::to_string(idx4[l]) + ".weight", &w));
      MX(off("vmi_estimator_f_t.baseline_model.MLP." + std::to_string(idx4[l]) + ".bias", &bb_));
      bl_l[l][0] = w - bl0; bl_l[l][1] = bb_ - bl0;
    }
  }
  {   // matrices whose transposed bf16 images the fused data-gradient chains read
    ttab.n = 0;
    auto add = [&](long o, int N, int K, int nb, long gs) {
      const int e = ttab.n++;
      ttab.off[e] = o; ttab.N[e] = N; ttab.K[e] = K; ttab.nb[e] = nb; ttab.gstride[e] = gs;
    };
    if (cfg.critic_type == MIMRL_CRITIC_SEPARATE) {
      const int d[5] = {EMB, HID, HID, HID, EMB};
      for (int l = 0; l < 4; ++l) add(tower0 + tower_l[l][0], d[l + 1], d[l], 10, tower_stride);
    } else {   // concat critic: the tail 256 -> 256 -> 256 -> 1 behind the pair-expanded first layer
      const int d[4] = {HID, HID, HID, 1};
      for (int l = 0; l < 3; ++l) add(tower0 + tower_l[l + 1][0], d[l + 1], d[l], NE_MI, tower_stride);
    }
    const int c[5] = {3 * EMB, HID, HID, HID, 2};
    for (int l = 0; l < 4; ++l) add(cmi0 + cmi_l[l][0], c[l + 1], c[l], NE_CMI, cmi_stride);
    // fragment-order images for the 4-layer stacks (separable towers, CMI classifiers, trainable baseline): a forward entry where
    // [N, K] is [32k x 64k], a data-gradient entry where [K, N] is
    std::memset(&ftab, 0, sizeof ftab);
    auto addf = [&](long o, int N, int K, int nb, long gs) {
      if (N % 32 == 0 && K % 64 == 0) { const int e = ftab.n++; ftab.off[e] = o; ftab.OUT[e] = N; ftab.RED[e] = K; ftab.nb[e] = nb; ftab.tr[e] = 0; ftab.gstride[e] = gs; }
      if (K % 32 == 0 && N % 64 == 0) {   // the data-gradient image lives one bucket length behind the forward one (crit_fragT)
        const int e = ftab.n++; ftab.off[e] = o; ftab.OUT[e] = K; ftab.RED[e] = N; ftab.nb[e] = nb; ftab.tr[e] = 1; ftab.gstride[e] = gs;
        ftab.dshift[e] = layout.floats[MIMRL_GROUP_CRITIC];
      }
    };
    if (cfg.critic_type == MIMRL_CRITIC_SEPARATE) {
      const int d[5] = {EMB, HID, HID, HID, EMB};
      for (int l = 0; l < 4; ++l) addf(tower0 + tower_l[l][0], d[l + 1], d[l], 10, tower_stride);
    }
    for (int l = 0; l < 3; ++l) addf(cmi0 + cmi_l[l][0], c[l + 1], c[l], NE_CMI, cmi_stride);
    if (cfg.baseline_type == MIMRL_BASELINE_UNNORMALIZED) {
      const int d[5] = {EMB, HID, HID, HID, 1};
      for (int l = 0; l < 4; ++l) add(bl0 + bl_l[l][0], d[l + 1], d[l], NE_MI, bl_stride);
      for (int l = 0; l < 3; ++l) addf(bl0 + bl_l[l][0], d[l + 1], d[l], NE_MI, bl_stride);
    }
  }
  return MIMRL_OK;
}

// forward-pass activations that exist twice (primary set / `alt` set, see FwdSet)
int mimrl_handle::carve_fwd(size_t* gmax_out) {
  const size_t B = cfg.batch, T = cfg.seq_len, L = cfg.time_len, D = cfg.d_common;
  const size_t BT_ = B * T;
  for (int m = 0; m < 2; ++m) MX(take(&lens[m], B));
  MX(take(&tx_raw, BT_ * D));
  for (int m = 0; m < 2; ++m) {
    for (int d = 0; d < 2; ++d) MX(take(&gx[m][d], BT_ * (cfg.encoder == MIMRL_ENCODER_LSTM ? 4 * H : G)));
    MX(take(&h0[m], BT_ * 2 * H));
    if (cfg.encoder == MIMRL_ENCODER_GRU) { float* t = nullptr; MX(take(&t, BT_ * H)); h0h[m] = reinterpret_cast<_Float16*>(t); }
    MX(take(&h1[m], BT_ * 2 * H));
    MX(take(&ln_mean[m], BT_));
    MX(take(&ln_rstd[m], BT_));
  }
  MX(take(&cube0, B * L * 3 * D));
  size_t gmax = B * L * 3 * D;
  int il = cfg.time_len, ik = 3, id = cfg.d_common;
  for (int i = 0; i < cfg.n_blocks; ++i) {
    const int hl = cfg.d_hiddens[i][0], hk = cfg.d_hiddens[i][1], hd = cfg.d_hiddens[i][2];
    const int ol = cfg.d_outs[i][0], ok = cfg.d_outs[i][1], od = cfg.d_outs[i][2];
    (void)hk;
    BlockBuf& b = bb[i];
    const size_t C = (size_t)ik * id;
    if (cfg.ln_first) { MX(take(&b.l.xn, B * il * C)); MX(take(&b.l.xn_mean, B * C)); MX(take(&b.l.xn_rstd, B * C)); }
    MX(take(&b.l.u, B * hl * C)); MX(take(&b.l.h, B * hl * C));
    MX(take(&b.l.y, B * ol * C));
    if (!cfg.ln_first) { MX(take(&b.l.z, B * ol * C)); MX(take(&b.l.mean, B * C)); MX(take(&b.l.rstd, B * C)); }
    else b.l.z = b.l.y;
    MX(take(&b.k.z, B * ol * ok * id));
    const size_t R2 = B * ol * ok;
    if (cfg.ln_first) { MX(take(&b.d.xn, R2 * id)); MX(take(&b.d.xn_mean, R2)); MX(take(&b.d.xn_rstd, R2)); }
    MX(take(&b.d.u, R2 * hd)); MX(take(&b.d.h, R2 * hd));
    MX(take(&b.d.y, R2 * od));
    if (!cfg.ln_first) { MX(take(&b.d.z, R2 * od)); MX(take(&b.d.mean, R2)); MX(take(&b.d.rstd, R2)); }
    else b.d.z = b.d.y;
    const size_t cand[] = {B * il * C, B * hl * C, B * ol * C, B * ol * ok * id, R2 * hd, R2 * od};
    for (size_t c : cand) gmax = c > gmax ? c : gmax;
    il = ol; ik = ok; id = od;
  }
  *gmax_out = gmax;
  return MIMRL_OK;
}

int mimrl_handle::carve() {
  const size_t B = cfg.batch, T = cfg.seq_len, D = cfg.d_common;
  const size_t BT_ = B * T;
  MX(take(&d_ints_own, 16));
  d_ints = d_ints_own;
  MX(take(&d_consts, 64));
  size_t gmax = 0;
  MX(carve_fwd(&gmax));
  for (int l = 0; l < 2; ++l)
    for (int m = 0; m < 2; ++m)
      for (int d = 0; d < 2; ++d) MX(take(&sv[l][m][d], (size_t)gru_saved_floats(cfg.batch, cfg.seq_len)));
  {   // the stage-1 side of prefetch mode (its features / prediction never reach the caller's buffers)
    float *f0 = bufs.feats, *p0 = bufs.pred;
    swap_fwd_set();
    size_t g2 = 0;
    int r = carve_fwd(&g2);
    if (r == 0) r = take(&bufs.feats, 4 * B * D);
    if (r == 0) r = take(&bufs.pred, B);
    swap_fwd_set();
    bufs.feats = f0; bufs.pred = p0;
    MX(r);
  }
  for (int i = 0; i < cfg.n_blocks; ++i)
    for (int q = 0; q < 3; ++q) { float* t = nullptr; MX(take(&t, 128 * 128 / 2)); wtT[i][q] = reinterpret_cast<__bf16*>(t); }
  {
    float *t1 = nullptr, *t2 = nullptr;
    MX(take(&t1, layout.floats[MIMRL_GROUP_CRITIC] / 2 + 64)); MX(take(&t2, layout.floats[MIMRL_GROUP_CRITIC] / 2 + 64));
    crit_img = reinterpret_cast<__bf16*>(t1); crit_imgT = reinterpret_cast<__bf16*>(t2);
    float* t3 = nullptr;   // both fragment-order images, back to back (one table, one launch: T entries carry the distance as dshift)
    MX(take(&t3, layout.floats[MIMRL_GROUP_CRITIC] + 64));
    crit_frag = reinterpret_cast<__bf16*>(t3); crit_fragT = crit_frag + layout.floats[MIMRL_GROUP_CRITIC];
  }
  MX(take(&ff, B * D));
  for (int i = 0; i < cfg.n_blocks; ++i) {
    const int hl = cfg.d_hiddens[i][0], ol = cfg.d_outs[i][0];
    if (hl % 4 != 0) MX(take(&w2p[i], (size_t)ol * ((hl + 3) & ~3) + 64));
  }
  MX(take(&dpred, B));
  if (cfg.encoder == MIMRL_ENCODER_GRU) {
    MX(take(&xpack, 2 * BT_ * KP())); MX(take(&wpack, (size_t)4 * G * KP())); MX(take(&bpack, (size_t)4 * G));
    MX(take(&dwih_pack, (size_t)4 * G * KP())); MX(take(&dwhh_pack, (size_t)4 * G * H));
    { float* t = nullptr; MX(take(&t, (size_t)4 * G * H)); w1h = reinterpret_cast<_Float16*>(t); MX(take(&t, (size_t)4 * G * H)); w1b = reinterpret_cast<__bf16*>(t);
      MX(take(&t, (size_t)4 * G * H)); w1bt = reinterpret_cast<__bf16*>(t); }
  }
  // estimators
  const bool sep = cfg.critic_type == MIMRL_CRITIC_SEPARATE;
  MX(take(&tin, 10 * B * EMB));
  MX(take(&scores, NE_MI * B * B));
  MX(take(&dscores, NE_MI * B * B));
  if (sep) {
    for (int l = 0; l < 3; ++l) MX(take(&ta[l], 10 * B * HID + ACT_SLACK));
    MX(take(&tout, 10 * B * EMB));
    MX(take(&dtout, 10 * B * EMB));
    for (int l = 0; l < 3; ++l) MX(take(&dta[l], 10 * B * HID));
  } else {
    MX(take(&cP, NE_MI * B * HID)); MX(take(&cQ, NE_MI * B * HID));
    MX(take(&dP, NE_MI * B * HID)); MX(take(&dQ, NE_MI * B * HID));
    for (int l = 0; l < 3; ++l) MX(take(&ca[l], NE_MI * B * B * HID + ACT_SLACK));   // (mlp_stack_backward's 8-wave kernel may read 4 rows past a ragged last tile: ADVICE r03)
    for (int l = 0; l < 3; ++l) MX(take(&dca[l], NE_MI * B * B * HID));
  }
  MX(take(&dtin, 10 * B * EMB));
  if (cfg.baseline_type != MIMRL_BASELINE_CONSTANT) {
    MX(take(&lbv, NE_MI * 2 * B)); MX(take(&dlbv, NE_MI * 2 * B)); MX(take(&bdin, NE_MI * 2 * B * EMB));
    if (cfg.baseline_type == MIMRL_BASELINE_UNNORMALIZED)
      for (int l = 0; l < 3; ++l) { MX(take(&bact[l], NE_MI * 2 * B * HID + ACT_SLACK)); MX(take(&bdz[l], NE_MI * 2 * B * HID)); }
  }
  const size_t n = nprod();
  MX(take(&knn_idx, NE_CMI * n)); MX(take(&knn_idx2, NE_CMI * n));
  knn_scr_bytes = knn_scratch_bytes(std::max(cfg.bank_capacity, 1), std::max(m_anchor(), 1), std::max(cfg.k_neighbor, 1));
  MX(take(&knn_scr[0], knn_scr_bytes)); MX(take(&knn_scr[1], knn_scr_bytes));
  MX(take(&cmi_in, NE_CMI * 2 * n * 384));
  for (int l = 0; l < 3; ++l) MX(take(&cc[l], NE_CMI * 2 * n * HID + ACT_SLACK));
  MX(take(&logits, NE_CMI * 2 * n * 2));
  MX(take(&dlogits, NE_CMI * 2 * n * 2));
  for (int l = 0; l < 3; ++l) MX(take(&dcc[l], NE_CMI * 2 * n * HID));
  MX(take(&dcin, NE_CMI * 2 * n * 384));
  MX(take(&mi_raw, 16)); MX(take(&cmi_raw, 8)); MX(take(&bce_raw, 8));   // mi_raw: 5 values + 5 loss terms
  MX(take(&dfeat, 4 * B * D));
  gbuf_floats = gmax;
  for (int i = 0; i < NGBUF; ++i) MX(take(&gbuf[i], gmax));
  MX(take(&dtx, BT_ * D));
  for (int m = 0; m < 2; ++m) {
    MX(take(&ds[m], BT_ * H));
    MX(take(&dh0[m], BT_ * 2 * H));
    for (int l = 0; l < 2; ++l)
      for (int d = 0; d < 2; ++d) {   // per layer: layer-1 weight-gradient GEMMs overlap the layer-0 BPTT
        MX(take(&dg[l][m][d], BT_ * 4 * H));
        MX(take(&hprev[l][m][d], BT_ * H));
      }
  }
  return MIMRL_OK;
}

int mimrl_handle::alloc_workspace() {
  ws = nullptr; ws_used = 0;
  MX(carve());                       // dry run: size
  ws_bytes = ws_used + 4096;
  HIPX(hipMalloc(reinterpret_cast<void**>(&ws), ws_bytes));
  HIPX(hipMemsetAsync(ws, 0, ws_bytes, stream));
  ws_used = 0;
  MX(carve());
  // constants
  float c[64];
  std::memset(c, 0, sizeof c);
  for (int i = 0; i < 11; ++i) c[i] = cfg.coef1[i];
  for (int i = 0; i < 8; ++i) c[11 + i] = cfg.coef2[i];
  const float* k1 = cfg.coef1; const float* k2 = cfg.coef2;
  for (int e = 0; e < NE_MI; ++e) c[19 + e] = -k1[e];                              // stage 1: d(loss)/d(mi_e)
  const float s2mi[NE_MI] = {-k2[0], -k2[1], -k2[2], -k2[3], -k2[3]};
  for (int e = 0; e < NE_MI; ++e) c[24 + e] = s2mi[e];
  for (int e = 0; e < NE_CMI; ++e) c[29 + e] = k1[NE_MI + e];                      // stage 1: coefficient of BCE_e
  // stage 2: d(loss)/d(cmi_e), Model.py:381-386 (order ac_t, ta_c, vc_t, tv_c, tc_a, tc_v)
  const float s2c[NE_CMI] = {-k2[5], k2[4] + k2[5] - k2[7], -k2[6], k2[4] + k2[6] - k2[7], -k2[4], -k2[4]};
  for (int e = 0; e < NE_CMI; ++e) c[47 + e] = s2c[e];                             // g_cmi(stage 2) = d_consts+41+6
  HIPX(hipMemcpyAsync(d_consts, c, sizeof c, hipMemcpyHostToDevice, stream));
  HIPX(hipStreamSynchronize(stream));
  return MIMRL_OK;
}

// =================================================================================================
// model forward
// =================================================================================================
// audio / video encoders: h1[m][B,T,2H] whose two halves the LN+ReLU+dropout epilogue adds (forward + reverse direction
// of the bi-GRU; the conv encoder writes the first half only, the second stays zero)
int mimrl_handle::encoders_forward(bool save, int knn_stage) {
  if (cfg.encoder == MIMRL_ENCODER_CONV) return conv_forward(knn_stage);
  if (cfg.encoder == MIMRL_ENCODER_LSTM) return lstm_encoders_forward(save, knn_stage);
  const int B = cfg.batch, T = cfg.seq_len;
  const long BT_ = (long)B * T;
  const float* xin[2] = {bufs.audio, bufs.video};
  const int dmod[2] = {cfg.d_a, cfg.d_v};
  // lengths (Model.py:425-432): only the recurrence needs them -> sides 4/5, next to the input projections
  if (!ev_lens) MX(seq_lengths2(S(4), xin[0], dmod[0], lens[0], xin[1], dmod[1], lens[1], B, T));
  // 16-bit operands for the layer-1 projection (see h0h): bf16 recurrence + fp16 forward operands + the packed layer-0 launch that also
  // writes the weight images; a site forced to fp32 (MIMRL_FWD_FP32_SITES) or fp16-stored gx keeps the fp32-operand kernel
  const bool use_h16 = h16_on && l0_packed && bf16 && fwd_f16 && (prec & MIMRL_PREC_BF16_GRU_FWD) && !fp32_site(4) && !fp32_site(2) && h0h[0] && w1h;
  // bi-GRU, 2 layers (Model.py:441-447); the four (modality,direction) input projections run on four streams
  for (int l = 0; l < 2; ++l) {
    GruFwdArgs a;
    a.B = B; a.T = T; a.out_ld = 2 * H; a.nmod = 2;
    a.btv = gru_pick_btv(B, 2);
    a.gx_f16 = gx_f16 ? 1 : 0;
    a.stamp = kstamp; a.stamp.id = l;
    if (l == 1) MX(fork(1, 3));
    if (l == 0 && !l0_packed && l0_bwd_pack && save) {   // packed copy of the inputs for the layer-0 weight gradients: side 0 has slack
      L0Pack pk;
      for (int m = 0; m < 2; ++m) { pk.x[m] = xin[m]; pk.d[m] = dmod[m]; for (int d = 0; d < 2; ++d) { pk.w_ih[m][d] = nullptr; pk.b_ih[m][d] = nullptr; } }
      pk.xpack = xpack; pk.wpack = wpack; pk.bpack = bpack; pk.rows = BT_; pk.KP = KP();
      MX(l0_pack(S(0), pk, true, false));
    }
    if (l == 0 && l0_packed) {
      // all four (modality, direction) projections of layer 0 as ONE batched launch on the packed operands
      L0Pack pk;
      for (int m = 0; m < 2; ++m) {
        pk.x[m] = xin[m]; pk.d[m] = dmod[m];
        for (int d = 0; d < 2; ++d) { pk.w_ih[m][d] = P(gru[m][0][d].w_ih); pk.b_ih[m][d] = P(gru[m][0][d].b_ih); }
      }
      pk.xpack = xpack; pk.wpack = wpack; pk.bpack = bpack; pk.rows = BT_; pk.KP = KP();
      // ... and of the layer-0 projection itself: the pack launch writes its operands as fp16 (+ a bf16 copy of the inputs for the W_ih
      // weight gradient, whose other operand -- dg -- is bf16) INSTEAD of fp32, in the same buffers
      const bool l0_16 = use_h16 && dg_bf16 && KP() % 8 == 0;
      xpack16 = l0_16;
      if (use_h16) {
        for (int m = 0; m < 2; ++m) for (int d = 0; d < 2; ++d) pk.w_ih1[m][d] = P(gru[m][1][d].w_ih);
        pk.w1h = w1h; pk.w1b = w1b; pk.w1bt = w1bt;
        w1_img_valid = true;
      }
      if (l0_16) {
        pk.xh = reinterpret_cast<_Float16*>(xpack); pk.xb = reinterpret_cast<__bf16*>(xpack + BT_ * KP());   // 2 * BT * KP halves each
        pk.wh = reinterpret_cast<_Float16*>(wpack);
      }
      if (begin_in_pack) {   // begin_stage(1) of the shared-prefix step rides on this launch (enqueue_grads)
        pk.bs_rng = d_ints; pk.bs_adam = d_ints + 2; pk.bs_scal = bufs.scalars; pk.bs_off = 0; pk.bs_n = 32;
        begin_in_pack = false;
      }
      MX(l0_pack(stream, pk, true));
      GemmDesc gd = gemm_nt(xpack, KP(), wpack, KP(), gx[0][0], G, (int)BT_, G, KP());
      gd.batch = 4; gd.batch_in = 2;
      gd.sa_b = 0; gd.sa_bo = BT_ * KP(); gd.sb_b = (long)G * KP(); gd.sb_bo = 2L * G * KP();
      gd.sc_b = gx[0][1] - gx[0][0]; gd.sc_bo = gx[1][0] - gx[0][0];
      gd.bias_n = bpack; gd.bias_n_b = G; gd.bias_n_bo = 2 * G;
      gd.f16 = fwd_f16;
      if (gx_f16) { gd.c_f16 = 1; gd.sc_b *= 2; gd.sc_bo *= 2; }   // buffer distances are fp32-element counts; fp16 elements: x2
      if (l0_16) { gd.a_bf16 = gd.b_bf16 = 1; }                    // (same element strides: the 16-bit arrays keep the fp32 ones' shapes)
      // Fused input projection (round 4): with the operands packed as fp16 the layer-0 recurrence kernel computes x W_ih^T + b_ih itself,
      // three k-steps per gate and cell step on a matrix pipe that is busy a quarter of the step: no GEMM launch, no gx round trip.
      l0_xin = l0_16 && xin_on && KP() <= 96 && !gx_f16;
      if (l0_xin) {
        a.xin_on = 1; a.kp = KP();
        for (int m = 0; m < 2; ++m) {
          a.xin[m] = pk.xh + (long)m * BT_ * KP();
          for (int d = 0; d < 2; ++d) { a.wih[m][d] = pk.wh + ((long)m * 2 + d) * G * KP(); a.bih[m][d] = bpack + ((long)m * 2 + d) * G; }
        }
      } else {
      PrecGuard pg(this, fp32_site(2));
      MX(G_on(stream, gd));
      }
      if (pending_text && pending_text_at == 1) { MX(pending_text()); pending_text = nullptr; }
    }
    for (int m = 0; m < 2; ++m) {
      a.lens[m] = lens[m];
      const float* in = l == 0 ? xin[m] : h0[m];
      // both directions read the same input: one GEMM, batch = direction (weights / outputs are a constant stride apart);
      // layer 1 has the same shape for audio and video: one launch, batch = (modality, direction)
      const GruDirW &gf = gru[m][l][0], &gr = gru[m][l][1];
      GemmDesc gd = gemm_nt(in, gf.din, P(gf.w_ih), gf.din, gx[m][0], G, (int)BT_, G, gf.din);
      gd.batch = 2; gd.sa_b = 0; gd.sb_b = gr.w_ih - gf.w_ih; gd.sc_b = gx[m][1] - gx[m][0];
      gd.bias_n = P(gf.b_ih); gd.bias_n_b = gr.b_ih - gf.b_ih;
      gd.f16 = fwd_f16;
      if (l == 1) {
        gd.batch = 4; gd.batch_in = 2;
        gd.sa_bo = h0[1] - h0[0]; gd.sb_bo = gru[1][l][0].w_ih - gf.w_ih; gd.sc_bo = gx[1][0] - gx[0][0];
        gd.bias_n_bo = gru[1][l][0].b_ih - gf.b_ih;
      }
      if (gx_f16) { gd.c_f16 = 1; gd.sc_b *= 2; gd.sc_bo *= 2; }
      if (l == 1 && use_h16) {   // both operands as stored fp16 (strides in fp16 elements; the images are [modality][direction][G, 2H])
        gd.A = reinterpret_cast<const float*>(h0h[0]); gd.a_bf16 = 1; gd.sa_bo = h0h[1] - h0h[0];
        gd.B = reinterpret_cast<const float*>(w1h); gd.b_bf16 = 1; gd.sb_b = (long)G * 2 * H; gd.sb_bo = 2L * G * 2 * H;
      }
      // layer 1: the m == 0 launch covers both modalities; layer 0: video beside audio (side 2, or behind the length scan on side 4
      // when the overlap mode has masked side 2 off -- both are joined in front of the recurrence)
      if ((l == 0 && !l0_packed) || (l == 1 && m == 0)) { PrecGuard pg(this, fp32_site(l == 0 ? 2 : 4)); MX(G_on(m == 0 ? stream : (side_on(2) ? S(2) : S(4)), gd)); }
      for (int d = 0; d < 2; ++d) {
        const GruDirW& g = gru[m][l][d];
        a.seq[m][d] = GruSeq{gx[m][d], P(g.w_hh), P(g.b_hh), l == 0 ? h0[m] : h1[m], save ? sv[l][m][d] : nullptr};
        if (l == 0 && use_h16) a.seq[m][d].out16 = h0h[m];
      }
    }
    MX(join(1, l == 0 ? 4 : 3));
    if (l == 0 && ev_lens) { HIPX(hipStreamWaitEvent(stream, ev_lens, 0)); ev_lens = nullptr; }
    if (l == 0 && knn_stage) {   // the kNN sampler needs only banks + anchors: overlap it with the recurrence (32 of 256 CUs busy)
      MX(fork(4, 4));
      MX(knn_launch(knn_stage, S(4)));
      MX(dbg_delay(S(4), 11));
    }
    { Scope sc(this, MIMRL_PH_GRU_FWD); MX(gru_forward(stream, a, (prec & MIMRL_PREC_BF16_GRU_FWD) != 0)); }
    if (l == 0 && pending_text && pending_text_at == 2) { MX(pending_text()); pending_text = nullptr; }
    MX(dbg_delay(stream, 1));
  }
  return MIMRL_OK;
}

// 1-layer bi-LSTM encoders (Model.py:250-252): hoisted input projection (one GEMM per modality, batch = direction), then
// the recurrence (lstm.hip).  Outputs land in h1[m][B,T,2H] like the GRU's, so everything downstream is shared.
int mimrl_handle::lstm_encoders_forward(bool save, int knn_stage) {
  const int B = cfg.batch, T = cfg.seq_len;
  const long BT_ = (long)B * T;
  const float* xin[2] = {bufs.audio, bufs.video};
  const int dmod[2] = {cfg.d_a, cfg.d_v};
  MX(seq_lengths2(S(4), xin[0], dmod[0], lens[0], xin[1], dmod[1], lens[1], B, T));
  MX(fork(2, 2));
  LstmFwdArgs a;
  a.B = B; a.T = T; a.out_ld = 2 * H; a.nmod = 2;
  for (int m = 0; m < 2; ++m) {
    a.lens[m] = lens[m];
    const GruDirW &gf = gru[m][0][0], &gr = gru[m][0][1];
    GemmDesc gd = gemm_nt(xin[m], gf.din, P(gf.w_ih), gf.din, gx[m][0], 4 * H, (int)BT_, 4 * H, gf.din);
    gd.batch = 2; gd.sa_b = 0; gd.sb_b = gr.w_ih - gf.w_ih; gd.sc_b = gx[m][1] - gx[m][0];
    gd.bias_n = P(gf.b_ih); gd.bias_n_b = gr.b_ih - gf.b_ih;
    MX(G_on(m == 0 ? stream : S(2), gd));
    for (int d = 0; d < 2; ++d) {
      const GruDirW& g = gru[m][0][d];
      a.seq[m][d] = LstmSeq{gx[m][d], P(g.w_hh), P(g.b_hh), h1[m], save ? sv[0][m][d] : nullptr};
    }
  }
  MX(join(2, 2));
  MX(join(4, 4));
  if (knn_stage) { MX(fork(4, 4)); MX(knn_launch(knn_stage, S(4))); }
  { Scope sc(this, MIMRL_PH_GRU_FWD); MX(lstm_forward(stream, a, (prec & MIMRL_PREC_BF16_GRU_FWD) ? 2 : 1)); }
  return MIMRL_OK;
}

int mimrl_handle::lstm_encoders_backward() {
  const int B = cfg.batch, T = cfg.seq_len;
  const long BT_ = (long)B * T;
  const float* xin[2] = {bufs.audio, bufs.video};
  LstmBwdArgs a;
  a.B = B; a.T = T; a.out_ld = 2 * H; a.dout_ld = H; a.nmod = 2;
  for (int m = 0; m < 2; ++m) {
    a.lens[m] = lens[m];
    for (int d = 0; d < 2; ++d)
      a.seq[m][d] = LstmSeqBwd{P(gru[m][0][d].w_hh), sv[0][m][d], h1[m], ds[m], dg[0][m][d], hprev[0][m][d]};
  }
  { Scope sc(this, MIMRL_PH_GRU_BWD); MX(lstm_backward(stream, a, (prec & MIMRL_PREC_BF16_GRU_BWD) ? 2 : 1)); }
  MX(fork(1, 3));
  int rr = 0;
  for (int m = 0; m < 2; ++m)
    for (int d = 0; d < 2; ++d) {
      const GruDirW& g = gru[m][0][d];
      const int sq = rr++ % 4;
      hipStream_t st = sq == 0 ? stream : S(sq);
      { GemmDesc q = gemm_tn(dg[0][m][d], 4 * H, xin[m], g.din, Gm(g.w_ih), g.din, 4 * H, g.din, (int)BT_); q.atomic = 1; MX(G_on(st, q)); }
      { GemmDesc q = gemm_tn(dg[0][m][d], 4 * H, hprev[0][m][d], H, Gm(g.w_hh), H, 4 * H, H, (int)BT_); q.atomic = 1; MX(G_on(st, q)); }
      MX(colsum(st, dg[0][m][d], BT_, 4 * H, 4 * H, Gm(g.b_ih)));
      MX(colsum(st, dg[0][m][d], BT_, 4 * H, 4 * H, Gm(g.b_hh)));
    }
  return MIMRL_OK;
}

// Conv1d(d, 128, kernel 3, padding 1) over time (Model.py:247-249,437-439) as three shifted GEMMs per modality:
// y[b,t] = b + W[:,:,0] x[b,t-1] + W[:,:,1] x[b,t] + W[:,:,2] x[b,t+1]; the zero padding is the row range of each tap.
int mimrl_handle::conv_forward(int knn_stage) {
  const int B = cfg.batch, T = cfg.seq_len;
  const float* xin[2] = {bufs.audio, bufs.video};
  const int dmod[2] = {cfg.d_a, cfg.d_v};
  if (knn_stage) { MX(fork(4, 4)); MX(knn_launch(knn_stage, S(4))); }
  MX(fork(2, 2));
  for (int m = 0; m < 2; ++m) {
    const int d = dmod[m];
    hipStream_t st = m == 0 ? stream : S(2);
    const int order[3] = {1, 0, 2};                     // the centre tap covers every row: it initialises the output
    for (int q = 0; q < 3; ++q) {
      const int tap = order[q];
      const int rows = tap == 1 ? T : T - 1;
      if (rows <= 0) continue;
      GemmDesc g;
      g.A = xin[m] + (tap == 2 ? d : 0); g.sa_m = d; g.sa_k = 1; g.sa_b = (long)T * d;
      g.B = P(conv_w[m]) + tap; g.sb_k = 3; g.sb_n = 3L * d; g.sb_b = 0;
      g.C = h1[m] + (tap == 0 ? 2 * H : 0); g.sc_m = 2 * H; g.sc_n = 1; g.sc_b = (long)T * 2 * H;
      g.M = rows; g.N = H; g.K = d; g.batch = B;
      if (tap == 1) g.bias_n = P(conv_b[m]); else g.beta = 1.f;
      MX(G_on(st, g));
    }
  }
  MX(join(2, 2));
  return MIMRL_OK;
}

// part 1 = the deterministic PREFIX (W_t projection, encoders: nothing random before their outputs tx_raw / h1), part 2 =
// the TAIL from the first dropout on; 0 = both.  In prefetch mode the two forward passes of one two-stage step see the
// same batch and the same main parameters, so their prefixes are the same function of the same inputs: it is evaluated
// once (into the primary set) and both tails read it.
int mimrl_handle::model_forward(bool train, bool save, int knn_stage, int part) {
  Range rg(part == 1 ? "mimrl.model_forward.prefix (Model.py:395-458)" : part == 2 ? "mimrl.model_forward.tail (Model.py:461-515)" : "mimrl.model_forward (Model.py:388-519)");
  const int B = cfg.batch, T = cfg.seq_len, L = cfg.time_len, D = cfg.d_common;
  const long BT_ = (long)B * T;
  const float pdrop[3] = {train ? cfg.dropout[0] : 0.f, train ? cfg.dropout[1] : 0.f, train ? cfg.dropout[2] : 0.f};
  if (part != 1 && T < L) HIPX(hipMemsetAsync(cube0, 0, sizeof(float) * (size_t)B * L * 3 * D, stream));
  // text_post + ln_relu_drop + feat_mean as one launch, one workgroup per (sample, slot): for short sequences, where the three
  // launches are latency (cfg2: -15 us per tail); a workgroup walking T = 1000 rows loses to the row-parallel kernels (cfg5: +60 us)
  static const bool fused_pre_on = knob("MIMRL_NO_FUSED_TAIL_PRE") == nullptr;   // tuning knob
  const bool fused_pre = fused_pre_on && T <= 128;
  if (part != 2) {
    MX(fork(0, 5));
    // text branch (side 0): W_t projection (Model.py:395) + dropout -> cube slot 0.  Captured BEFORE the encoders although it
    // has slack until the tail starts: graph nodes start in capture order, and a branch captured behind the two GRU layers is
    // dispatched behind them too and then delays the tail (measured: 1.46 vs 1.34 ms).
    // (default since round 4 -- with the kNN sampler's two launches on side 4 the scan there delayed the layer-0 input projection:
    //  cfg2 0.840 -> 0.829 ms, cfg3 6.97 -> 6.87; MIMRL_LENS_SIDE0=0 puts it back)
    static const bool prefix_split = !(knob("MIMRL_LENS_SIDE0") && atoi(knob("MIMRL_LENS_SIDE0")) == 0);   // tuning knob
    if (prefix_split && cfg.encoder == MIMRL_ENCODER_GRU && side_on(0)) {
      // lengths (Model.py:425-432): only the recurrence needs them.  Side 0 has slack (the text projection is needed at the tail);
      // on side 4 the scan sat in front of the video input projection, the longest chain ahead of the layer-0 recurrence
      MX(seq_lengths2(S(0), bufs.audio, cfg.d_a, lens[0], bufs.video, cfg.d_v, lens[1], B, T));
      MX(next_event(&ev_lens));
      HIPX(hipEventRecord(ev_lens, S(0)));
    }
    static const int text_late = knob("MIMRL_TEXT_LATE") ? atoi(knob("MIMRL_TEXT_LATE")) : 0;   // tuning knob (capture order)
    auto text_branch = [this, BT_, D, part, fused_pre, B, T, L, pdrop]() -> int {
      { PrecGuard pg(this, fp32_site(1)); GemmDesc g = gemm_nt(bufs.text, cfg.d_t, P(w_t), cfg.d_t, tx_raw, D, (int)BT_, D, cfg.d_t); g.f16 = fwd_f16; MX(G_on(S(0), g)); }
      MX(dbg_delay(S(0), 10));
      if (part == 0 && !fused_pre) MX(text_post_fwd(S(0), tx_raw, cube0, B, T, L, 3, D, 0, pdrop[0], key(), 0));
      return MIMRL_OK;
    };
    if (text_late && cfg.encoder == MIMRL_ENCODER_GRU && l0_packed) { pending_text = text_branch; pending_text_at = text_late; }
    else MX(text_branch());
    MX(encoders_forward(save, knn_stage));
    if (pending_text) { MX(pending_text()); pending_text = nullptr; }
    MX(join(0, 0));
    if (part == 1) return MIMRL_OK;
  } else if (!fused_pre) {
    MX(text_post_fwd(stream, tx_raw, cube0, B, T, L, 3, D, 0, pdrop[0], key(), 0));
  }
  // text dropout -> cube slot 0; fwd+bwd sum, LN, ReLU, dropout (Model.py:452-461) -> cube slots 1,2; T_F, A_F, V_F (Model.py:466)
  {
    LnSide2 sd[2];
    for (int m = 0; m < 2; ++m)
      sd[m] = LnSide2{h1[m], P(ln_g[m]), P(ln_b[m]), ln_mean[m], ln_rstd[m], nullptr, nullptr, nullptr, 1 + m, pdrop[1 + m],
                      (uint32_t)(1 + m)};
    if (fused_pre) {   // one launch instead of three on the chain of each tail
      MX(tail_pre_fwd(stream, tx_raw, pdrop[0], sd[0], sd[1], cube0, bufs.feats + (size_t)B * D, B, T, L, 3, D, key()));
    } else {
      MX(ln_relu_drop_fwd2(stream, sd[0], sd[1], cube0, B, T, L, 3, D, key()));   // audio and video in one launch
      MX(feat_mean_fwd(stream, cube0, bufs.feats + (size_t)B * D, B, T, L, 3, D));
    }
  }
  { Scope sc(this, MIMRL_PH_CUBE_FWD); MX(cube_forward(train, save)); }
  MX(dbg_delay(stream, save ? 2 : 13));
  // head (Model.py:489-515)
  const BlockBuf& last = bb[cfg.n_blocks - 1];
  const int ol = cfg.d_outs[cfg.n_blocks - 1][0], ok = cfg.d_outs[cfg.n_blocks - 1][1], od = cfg.d_outs[cfg.n_blocks - 1][2];
  if (od != D) return set_error(MIMRL_ERR_ARG, "last block d_out must equal d_common (features feed 128-wide estimators)");
  MX(head_fwd(stream, last.d.z, P(cls_w), P(cls_b), bufs.feats, bufs.pred, B, ol, ok, od, cfg.compose_t_sum,
              cfg.compose_k_sum));
  (void)ff;
  return MIMRL_OK;
}

int mimrl_handle::cube_forward(bool train, bool save) {
  const int B = cfg.batch;
  const float* x = cube0;
  int il = cfg.time_len, ik = 3, id = cfg.d_common;
  const bool no_fused = !fused_cube;
  for (int i = 0; i < cfg.n_blocks; ++i) {
    const BlockW& w = blk[i];
    BlockBuf& b = bb[i];
    const int hl = w.ax[0].hid, ol = w.ax[0].out, hk = w.ax[1].hid, ok = w.ax[1].out, hd = w.ax[2].hid, od = w.ax[2].out;
    const float pl = train ? cfg.dropout_mlp[0] : 0.f, pk = train ? cfg.dropout_mlp[1] : 0.f,
                pd = train ? cfg.dropout_mlp[2] : 0.f;
    const float pmlp[3] = {pl, pk, pd};
    // bf16 mode: the whole block as ONE kernel with the sample tile resident in LDS (cube_fused.hip)
    if (bf16 && !no_fused &&
        cube_fused_supported(il, hl, ol, ik, hk, ok, id, hd, od, cfg.ln_first != 0, cfg.res_project[i] != 0, cfg.bias != 0, pmlp)) {
      CubeFusedArgs fa;
      std::memset(&fa, 0, sizeof fa);
      auto PB = [&](long off) -> const float* { return off >= 0 ? P(off) : nullptr; };
      fa.x = x;
      fa.l_w1 = P(w.ax[0].fc1.w); fa.l_b1 = PB(w.ax[0].fc1.b); fa.l_w2 = P(w.ax[0].fc2.w); fa.l_b2 = PB(w.ax[0].fc2.b);
      fa.l_wr = P(w.ax[0].res); fa.l_g = P(w.ax[0].ln_g); fa.l_be = P(w.ax[0].ln_b);
      fa.kw.w1 = P(w.ax[1].fc1.w); fa.kw.b1 = PB(w.ax[1].fc1.b); fa.kw.w2 = P(w.ax[1].fc2.w); fa.kw.b2 = PB(w.ax[1].fc2.b);
      fa.kw.wr = P(w.ax[1].res); fa.kw.g = P(w.ax[1].ln_g); fa.kw.be = P(w.ax[1].ln_b);
      fa.kw.ik = ik; fa.kw.hk = hk; fa.kw.ok = ok; fa.kw.act = cfg.activation; fa.kw.ln_first = 0; fa.kw.drop_p = 0.f;
      fa.kw.key = key(); fa.kw.stream_id = 0;
      fa.d_w1 = P(w.ax[2].fc1.w); fa.d_b1 = PB(w.ax[2].fc1.b); fa.d_w2 = P(w.ax[2].fc2.w); fa.d_b2 = PB(w.ax[2].fc2.b);
      fa.d_wr = P(w.ax[2].res); fa.d_g = P(w.ax[2].ln_g); fa.d_be = P(w.ax[2].ln_b);
      if (save) {
        fa.l_u = b.l.u; fa.l_h = b.l.h; fa.l_y = b.l.y; fa.l_z = b.l.z; fa.l_mean = b.l.mean; fa.l_rstd = b.l.rstd;
        fa.k_z = b.k.z;
        fa.d_u = b.d.u; fa.d_h = b.d.h; fa.d_y = b.d.y; fa.d_mean = b.d.mean; fa.d_rstd = b.d.rstd;
      }
      fa.d_z = b.d.z;
      fa.dbg_phase = dbg_env("MIMRL_CUBE_PHASE") ? atoi(dbg_env("MIMRL_CUBE_PHASE")) : 0;
      fa.B = B; fa.il = il; fa.hl = hl; fa.ol = ol; fa.K = ik; fa.act = cfg.activation; fa.save = save ? 1 : 0;
      if (fa.dbg_phase == 100) { fa.save = 0; fa.dbg_phase = 0; }   // timing-only: skip the saved-activation stores
      MX(cube_block_fwd_fused(stream, fa));
      x = b.d.z;
      il = ol; ik = ok; id = od;
      continue;
    }
    // ------------------------------------------------ L axis (MLPProcess.py:95-104 / 65-74)
    {
      const AxisW& a = w.ax[0];
      const long C = (long)ik * id;
      const float* xi = x;
      if (cfg.ln_first) { MX(colln_fwd(stream, x, P(a.ln_g), P(a.ln_b), b.l.xn, b.l.xn_mean, b.l.xn_rstd, B, il, (int)C)); xi = b.l.xn; }
      GemmDesc g1;   // H = act(W1 . X_b + b1)
      g1.A = P(a.fc1.w); g1.sa_m = il; g1.sa_k = 1; g1.sa_b = 0;
      g1.B = xi; g1.sb_k = C; g1.sb_n = 1; g1.sb_b = (long)il * C;
      g1.C = b.l.h; g1.sc_m = C; g1.sc_n = 1; g1.sc_b = (long)hl * C;
      g1.M = hl; g1.N = (int)C; g1.K = il; g1.batch = B;
      g1.bias_m = a.fc1.b >= 0 ? P(a.fc1.b) : nullptr; g1.act = cfg.activation; g1.pre = b.l.u;
      g1.f16 = fwd_f16;
      MX(G_(g1));
      GemmDesc g2;   // Y = W2 . H_b + b2
      g2.A = P(a.fc2.w); g2.sa_m = hl; g2.sa_k = 1;
      if (bf16 && w2p[i] && hl % 4 != 0) {   // a 50-wide fc2 has 200-byte rows: without the padded copy this product (and dU in the
        const int hp = (hl + 3) & ~3;        // backward pass) falls to the scalar-load kernel -- 0.2-0.27 ms each at cfg3
        MX(pad_rows(stream, P(a.fc2.w), w2p[i], ol, hl, hp));
        w2p_valid[i] = true;
        g2.A = w2p[i]; g2.sa_m = hp; g2.a_pad4 = 1;
      }
      g2.B = b.l.h; g2.sb_k = C; g2.sb_n = 1; g2.sb_b = (long)hl * C;
      g2.C = b.l.y; g2.sc_m = C; g2.sc_n = 1; g2.sc_b = (long)ol * C;
      g2.M = ol; g2.N = (int)C; g2.K = hl; g2.batch = B;
      g2.bias_m = a.fc2.b >= 0 ? P(a.fc2.b) : nullptr;
      const bool fuse_res = a.res >= 0 && pl <= 0.f;     // Y = W2.H + b2 + Wr.X in ONE launch (no dropout in between)
      if (fuse_res) {
        g2.A2 = P(a.res); g2.sa2_m = il; g2.sa2_k = 1; g2.sa2_b = 0;
        g2.B2 = x; g2.sb2_k = C; g2.sb2_n = 1; g2.sb2_b = (long)il * C; g2.K2 = il;
      }
      MX(G_(g2));
      if (!fuse_res) {
        MX(dropout_inplace(stream, b.l.y, (long)B * ol * C, pl, key(), 10 + 3 * i));
        if (a.res >= 0) {   // Y += Wr . X_b
          GemmDesc g3 = g2;
          g3.A = P(a.res); g3.sa_m = il; g3.B = x; g3.sb_b = (long)il * C; g3.K = il; g3.bias_m = nullptr; g3.beta = 1.f;
          MX(G_(g3));
        } else {
          MX(add_inplace(stream, b.l.y, x, (long)B * ol * C));
        }
      }
      if (!cfg.ln_first) MX(colln_fwd(stream, b.l.y, P(a.ln_g), P(a.ln_b), b.l.z, b.l.mean, b.l.rstd, B, ol, (int)C));
    }
    // ------------------------------------------------ K axis (MLPProcess.py:106-112 / 76-82)
    {
      const AxisW& a = w.ax[1];
      KMixW kw;
      std::memset(&kw, 0, sizeof kw);
      kw.w1 = P(a.fc1.w); kw.b1 = a.fc1.b >= 0 ? P(a.fc1.b) : nullptr;
      kw.w2 = P(a.fc2.w); kw.b2 = a.fc2.b >= 0 ? P(a.fc2.b) : nullptr;
      kw.wr = a.res >= 0 ? P(a.res) : nullptr; kw.g = P(a.ln_g); kw.be = P(a.ln_b);
      kw.ik = ik; kw.hk = hk; kw.ok = ok; kw.act = cfg.activation; kw.ln_first = cfg.ln_first;
      kw.drop_p = pk; kw.key = key(); kw.stream_id = 11 + 3 * i;
      MX(kmix_fwd(stream, b.l.z, b.k.z, kw, (long)B * ol, id));
    }
    // ------------------------------------------------ D axis (MLPProcess.py:114-120 / 84-90)
    {
      const AxisW& a = w.ax[2];
      const long R2 = (long)B * ol * ok;
      const float* xi = b.k.z;
      if (cfg.ln_first) { MX(rowln_fwd(stream, b.k.z, P(a.ln_g), P(a.ln_b), b.d.xn, b.d.xn_mean, b.d.xn_rstd, R2, id)); xi = b.d.xn; }
      GemmDesc g1 = gemm_nt(xi, id, P(a.fc1.w), id, b.d.h, hd, (int)R2, hd, id);
      g1.bias_n = a.fc1.b >= 0 ? P(a.fc1.b) : nullptr; g1.act = cfg.activation; g1.pre = b.d.u;
      MX(G_(g1));
      GemmDesc g2 = gemm_nt(b.d.h, hd, P(a.fc2.w), hd, b.d.y, od, (int)R2, od, hd);
      g2.bias_n = a.fc2.b >= 0 ? P(a.fc2.b) : nullptr;
      const bool fuse_res = a.res >= 0 && pd <= 0.f;
      if (fuse_res) {
        g2.A2 = b.k.z; g2.sa2_m = id; g2.sa2_k = 1; g2.B2 = P(a.res); g2.sb2_k = 1; g2.sb2_n = id; g2.K2 = id;
      }
      MX(G_(g2));
      if (!fuse_res) {
        MX(dropout_inplace(stream, b.d.y, R2 * od, pd, key(), 12 + 3 * i));
        if (a.res >= 0) {
          GemmDesc g3 = gemm_nt(b.k.z, id, P(a.res), id, b.d.y, od, (int)R2, od, id);
          g3.beta = 1.f;
          MX(G_(g3));
        } else {
          MX(add_inplace(stream, b.d.y, b.k.z, R2 * od));
        }
      }
      if (!cfg.ln_first) MX(rowln_fwd(stream, b.d.y, P(a.ln_g), P(a.ln_b), b.d.z, b.d.mean, b.d.rstd, R2, od));
    }
    x = b.d.z;
    il = ol; ik = ok; id = od;
  }
  return MIMRL_OK;
}

// =================================================================================================
// CubeMLP backward.  The incoming gradient (w.r.t. the last block's output) lives in gbuf[cur_in]; on return
// gbuf[*cur_out] holds d(cube0).  Four rotating gradient buffers are enough: at any time at most
// {dY, dY through dropout, dU, dX} are live.
// =================================================================================================
// which blocks run the fused D-axis backward (needs bf16 operands in the backward section) + their transposed weight images
int mimrl_handle::wt_images(hipStream_t st, bool bwd_bf16, bool launch, bool* d_fused) {
  int din2 = cfg.d_common;
  WtTransposeArgs ta;
  ta.n = 0;
  for (int i = 0; i < cfg.n_blocks; ++i) {
    const AxisW& a = blk[i].ax[2];
    d_fused[i] = bwd_bf16 && fused_cube_bwd && !cfg.ln_first && cfg.dropout_mlp[2] <= 0.f && a.res >= 0 &&
                 daxis_bwd_supported(din2, a.hid, a.out) && ta.n + 3 <= 12;
    din2 = cfg.d_outs[i][2];
    if (!d_fused[i]) continue;
    const long srcs[3] = {a.fc2.w, a.fc1.w, a.res};
    for (int q = 0; q < 3; ++q) { ta.src[ta.n] = P(srcs[q]); ta.dst[ta.n] = wtT[i][q]; ++ta.n; }
  }
  if (launch && ta.n > 0) MX(wt_transpose_bf16(st, ta));
  return MIMRL_OK;
}

int mimrl_handle::cube_backward(int cur_in, int* cur_out) {
  const int B = cfg.batch;
  int dims[MIMRL_MAX_BLOCKS + 1][3];
  dims[0][0] = cfg.time_len; dims[0][1] = 3; dims[0][2] = cfg.d_common;
  for (int i = 0; i < cfg.n_blocks; ++i)
    for (int ax = 0; ax < 3; ++ax) dims[i + 1][ax] = cfg.d_outs[i][ax];
  // deferred mode: every gradient buffer is used once (weight-gradient GEMMs read dY / dU after the chain has moved on)
  static const bool no_defer = knob("MIMRL_NO_DEFER_WGRAD") != nullptr;   // tuning knob
  const int per_block = 7 + (cfg.dropout_mlp[0] > 0.f) + (cfg.dropout_mlp[2] > 0.f);   // buffers one block consumes
  const bool defer = multi_stream && !cfg.ln_first && !no_defer && per_block * cfg.n_blocks + 1 <= NGBUF;
  const int npool = defer ? NGBUF : 4;
  int used[NGBUF] = {0};
  used[cur_in] = 1;
  int cur = cur_in;
  auto grab = [&]() { for (int q = 0; q < npool; ++q) if (!used[q]) { used[q] = 1; return q; } return -1; };
  auto release = [&](int q) { if (!defer) used[q] = 0; };
  auto W_gemm = [&](int sd, const GemmDesc& g) -> int {
    if (defer) { deferred.push_back(Deferred{0, sd, g, nullptr, 0, 0, 0, 0, nullptr}); return MIMRL_OK; }
    return G_on(S(sd), g);
  };
  auto W_colsum = [&](int sd, const float* src, long rows, int cols, int ld, float* dst) -> int {
    if (defer) { deferred.push_back(Deferred{1, sd, GemmDesc(), src, rows, cols, ld, 0, dst}); return MIMRL_OK; }
    return colsum(S(sd), src, rows, cols, ld, dst);
  };
  auto W_rowsum = [&](int sd, const float* src, int nb, int rows, int cols, float* dst) -> int {
    if (defer) { deferred.push_back(Deferred{2, sd, GemmDesc(), src, nb, rows, cols, 0, dst}); return MIMRL_OK; }
    return rowsum_batched(S(sd), src, nb, rows, cols, dst);
  };
  auto W_lnpar = [&](int sd, const float* y, const float* mean, const float* rstd, const float* dz, float* dgam, float* dbet,
                     int nb, int n, int cols) -> int {
    if (defer) {
      Deferred d{3, sd, GemmDesc(), y, nb, n, cols, 0, dgam};
      d.p1 = mean; d.p2 = rstd; d.p3 = dz; d.dst2 = dbet;
      deferred.push_back(d);
      return MIMRL_OK;
    }
    return colln_param_grads(S(sd), y, mean, rstd, dz, dgam, dbet, nb, n, cols);
  };
  auto W_lnrow = [&](int sd, const float* y, const float* mean, const float* rstd, const float* dz, float* dgam, float* dbet,
                     long rows, int n) -> int {
    if (defer) {
      Deferred d{4, sd, GemmDesc(), y, rows, n, 0, 0, dgam};
      d.p1 = mean; d.p2 = rstd; d.p3 = dz; d.dst2 = dbet;
      deferred.push_back(d);
      return MIMRL_OK;
    }
    return rowln_param_grads(S(sd), y, mean, rstd, dz, dgam, dbet, rows, n);
  };
  auto W_dpg = [&](int sd, const float* y, const float* mean, const float* rstd, const float* dz, const float* dy, const float* du,
                   float* dgam, float* dbet, float* db2, float* db1, long rows) -> int {
    if (defer) {
      Deferred d{6, sd, GemmDesc(), y, rows, 0, 0, 0, dgam};
      d.p1 = mean; d.p2 = rstd; d.p3 = dz; d.p4 = dy; d.p5 = du; d.dst2 = dbet; d.dst3 = db2; d.dst4 = db1;
      deferred.push_back(d);
      return MIMRL_OK;
    }
    return daxis_param_grads(S(sd), y, mean, rstd, dz, dy, du, dgam, dbet, db2, db1, rows);
  };
  auto W_fork = [&](int lo, int hi) -> int { return defer ? MIMRL_OK : fork(lo, hi); };
  // transposed bf16 images of the D-axis weights for the fused data-gradient kernels (one small launch; in a combined step
  // it already ran at step start on side 0, off this chain)
  bool d_fused[MIMRL_MAX_BLOCKS] = {};
  MX(wt_images(stream, bf16, !(wtT_prebuilt && wtT_built), d_fused));
  auto W_join = [&](int lo, int hi) -> int { return defer ? MIMRL_OK : join(lo, hi); };
#define GRAB(var)                                                                                   \
  const int var = grab();                                                                           \
  if (var < 0) return set_error(MIMRL_ERR_STATE, "cube_backward: out of gradient buffers (line %d)", __LINE__)

  for (int i = cfg.n_blocks - 1; i >= 0; --i) {
    const BlockW& w = blk[i];
    BlockBuf& b = bb[i];
    const int il = dims[i][0], ik = dims[i][1], id = dims[i][2];
    const int hl = w.ax[0].hid, ol = w.ax[0].out, hk = w.ax[1].hid, ok = w.ax[1].out, hd = w.ax[2].hid, od = w.ax[2].out;
    const float* xblk = i == 0 ? cube0 : bb[i - 1].d.z;
    const float pl = cfg.dropout_mlp[0], pk = cfg.dropout_mlp[1], pd = cfg.dropout_mlp[2];
    // ------------------------------------------------ D axis backward
    if (d_fused[i]) {
      // one launch: LayerNorm(D) backward -> dY -> dU -> dX (row tiles); weight / bias / LayerNorm gradients stay side work
      const AxisW& a = w.ax[2];
      const long R2 = (long)B * ol * ok;
      GRAB(i_dy); GRAB(i_du); GRAB(i_dx);
      DAxisBwdArgs fa;
      fa.dz = gbuf[cur]; fa.y = b.d.y; fa.mean = b.d.mean; fa.rstd = b.d.rstd; fa.gamma = P(a.ln_g); fa.u = b.d.u;
      fa.w2t = wtT[i][0]; fa.w1t = wtT[i][1]; fa.wrt = wtT[i][2];
      fa.dy = gbuf[i_dy]; fa.du = gbuf[i_du]; fa.dx = gbuf[i_dx];
      fa.R = R2; fa.act = cfg.activation;
      // LayerNorm and bias gradients (column sums over the rows of dz, y, dY, dU).  Folded into the data-gradient kernel they cost
      // the chain 13 us per block (MIMRL_DAXIS_PG_FUSE=1: 19 -> 32 us); as ONE streaming side kernel instead of rowln_param_grads +
      // 2 x colsum (3 launches of 30-40 us each) they are ~10 us beside the BPTT (MIMRL_NO_DAXIS_PG_ONE=1: the three launches)
      static const bool pg_fuse = knob("MIMRL_DAXIS_PG_FUSE") != nullptr;        // tuning knobs
      static const bool no_pg_one = knob("MIMRL_NO_DAXIS_PG_ONE") != nullptr;
      const bool pg_fused = pg_fuse && a.fc2.b >= 0 && a.fc1.b >= 0;
      const bool pg_one = !pg_fused && !no_pg_one;
      fa.dgamma = fa.dbeta = fa.db2 = fa.db1 = nullptr;
      if (pg_fused) { fa.dgamma = Gm(a.ln_g); fa.dbeta = Gm(a.ln_b); fa.db2 = Gm(a.fc2.b); fa.db1 = Gm(a.fc1.b); }
      MX(daxis_bwd_fused(stream, fa));
      MX(W_fork(1, 3));
      if (pg_one) MX(W_dpg(2, b.d.y, b.d.mean, b.d.rstd, gbuf[cur], gbuf[i_dy], gbuf[i_du], Gm(a.ln_g), Gm(a.ln_b),
                           a.fc2.b >= 0 ? Gm(a.fc2.b) : nullptr, a.fc1.b >= 0 ? Gm(a.fc1.b) : nullptr, R2));
      if (!pg_fused && !pg_one) MX(W_lnrow(2, b.d.y, b.d.mean, b.d.rstd, gbuf[cur], Gm(a.ln_g), Gm(a.ln_b), R2, od));
      { GemmDesc g = gemm_tn(gbuf[i_dy], od, b.d.h, hd, Gm(a.fc2.w), hd, od, hd, (int)R2); g.atomic = 1; MX(W_gemm(1, g)); }
      if (!pg_fused && !pg_one && a.fc2.b >= 0) MX(W_colsum(1, gbuf[i_dy], R2, od, od, Gm(a.fc2.b)));
      { GemmDesc g = gemm_tn(gbuf[i_dy], od, b.k.z, id, Gm(a.res), id, od, id, (int)R2); g.atomic = 1; MX(W_gemm(2, g)); }
      { GemmDesc g = gemm_tn(gbuf[i_du], hd, b.k.z, id, Gm(a.fc1.w), id, hd, id, (int)R2); g.atomic = 1; MX(W_gemm(3, g)); }
      if (!pg_fused && !pg_one && a.fc1.b >= 0) MX(W_colsum(3, gbuf[i_du], R2, hd, hd, Gm(a.fc1.b)));
      MX(W_join(1, 3));
      release(cur); release(i_dy); release(i_du);
      cur = i_dx;
    } else {
      const AxisW& a = w.ax[2];
      const long R2 = (long)B * ol * ok;
      const float* xin = b.k.z;                              // residual / un-normalised input
      const float* xmlp = cfg.ln_first ? b.d.xn : b.k.z;     // what fc1 saw
      int i_dy = cur;
      if (!cfg.ln_first) {
        GRAB(q);
        MX(rowln_bwd(stream, b.d.y, P(a.ln_g), b.d.mean, b.d.rstd, gbuf[cur], gbuf[q], Gm(a.ln_g), Gm(a.ln_b), R2, od));
        release(cur);
        i_dy = q;
      }
      const float* dy = gbuf[i_dy];
      int i_dym = i_dy;
      if (pd > 0.f) {                                        // gradient entering the dropped-out MLP branch
        GRAB(q);
        HIPX(hipMemcpyAsync(gbuf[q], dy, sizeof(float) * R2 * od, hipMemcpyDeviceToDevice, stream));
        MX(dropout_inplace(stream, gbuf[q], R2 * od, pd, key(), 12 + 3 * i));
        i_dym = q;
      }
      const float* dym = gbuf[i_dym];
      MX(W_fork(1, 2));                                      // weight gradients leave the critical path
      { GemmDesc g = gemm_tn(dym, od, b.d.h, hd, Gm(a.fc2.w), hd, od, hd, (int)R2); g.atomic = 1; MX(W_gemm(1, g)); }
      if (a.fc2.b >= 0) MX(W_colsum(1, dym, R2, od, od, Gm(a.fc2.b)));
      if (a.res >= 0) { GemmDesc g = gemm_tn(dy, od, xin, id, Gm(a.res), id, od, id, (int)R2); g.atomic = 1; MX(W_gemm(2, g)); }
      GRAB(i_du);                                            // dU = (dYm . W2) * act'(U)
      { GemmDesc g = gemm_nn(dym, od, P(a.fc2.w), hd, gbuf[i_du], hd, (int)R2, hd, od); g.act = cfg.activation; g.gradact_u = b.d.u;
        if (a.fc1.b >= 0) g.colsum = Gm(a.fc1.b);     // db1 = column sums of dU, fused into the epilogue
        MX(G_(g)); }
      MX(W_fork(3, 3));
      { GemmDesc g = gemm_tn(gbuf[i_du], hd, xmlp, id, Gm(a.fc1.w), id, hd, id, (int)R2); g.atomic = 1; MX(W_gemm(3, g)); }
      GRAB(i_dx0);
      int i_dx = i_dx0;
      const bool fuse_dx = a.res >= 0 && !cfg.ln_first;        // dX = dU.W1 + dY.Wr in ONE launch
      { GemmDesc g = gemm_nn(gbuf[i_du], hd, P(a.fc1.w), id, gbuf[i_dx], id, (int)R2, id, hd);
        if (fuse_dx) { g.A2 = dy; g.sa2_m = od; g.sa2_k = 1; g.B2 = P(a.res); g.sb2_k = id; g.sb2_n = 1; g.K2 = od; }
        MX(G_(g)); }
      if (cfg.ln_first) {                                    // that was dXn: LayerNorm backward into the dU buffer,
        MX(join(3, 3));                                      // once the dW1 GEMM on side 3 has finished reading it
        MX(rowln_bwd(stream, b.k.z, P(a.ln_g), b.d.xn_mean, b.d.xn_rstd, gbuf[i_dx], gbuf[i_du], Gm(a.ln_g), Gm(a.ln_b), R2, id));
        release(i_dx);
        i_dx = i_du;
      } else {
        release(i_du);
      }
      if (!fuse_dx) {
        if (a.res >= 0) { GemmDesc g = gemm_nn(dy, od, P(a.res), id, gbuf[i_dx], id, (int)R2, id, od); g.beta = 1.f; MX(G_(g)); }
        else MX(add_inplace(stream, gbuf[i_dx], dy, R2 * id));
      }
      MX(W_join(1, 3));                                      // side streams are done with dy / dym / dU before they are recycled
      if (i_dym != i_dy) release(i_dym);
      release(i_dy);
      cur = i_dx;
    }
    // ------------------------------------------------ K axis backward
    {
      const AxisW& a = w.ax[1];
      KMixW kw;
      std::memset(&kw, 0, sizeof kw);
      kw.w1 = P(a.fc1.w); kw.b1 = a.fc1.b >= 0 ? P(a.fc1.b) : nullptr;
      kw.w2 = P(a.fc2.w); kw.b2 = a.fc2.b >= 0 ? P(a.fc2.b) : nullptr;
      kw.wr = a.res >= 0 ? P(a.res) : nullptr; kw.g = P(a.ln_g); kw.be = P(a.ln_b);
      kw.dw1 = Gm(a.fc1.w); kw.db1 = a.fc1.b >= 0 ? Gm(a.fc1.b) : nullptr;
      kw.dw2 = Gm(a.fc2.w); kw.db2 = a.fc2.b >= 0 ? Gm(a.fc2.b) : nullptr;
      kw.dwr = a.res >= 0 ? Gm(a.res) : nullptr; kw.dg = Gm(a.ln_g); kw.dbe = Gm(a.ln_b);
      kw.ik = ik; kw.hk = hk; kw.ok = ok; kw.act = cfg.activation; kw.ln_first = cfg.ln_first;
      kw.drop_p = pk; kw.key = key(); kw.stream_id = 11 + 3 * i;
      GRAB(q);
      // K-axis parameter gradients: IN the chain kernel (MODE 0: data + parameter gradients; 0.99 ms at cfg2).  The split of round
      // 2a -- data gradient on the chain, the parameter-gradient reductions as a parked kernel beside the BPTT, 0.97 ms -- was NOT
      // reproducible (see MIMRL_EARLY_FLUSH below), and starting that kernel early on side 3 with the BPTT waiting for it costs more
      // (1.02 ms: it fights the chain for CUs).  MIMRL_KMIX_PG_INCHAIN=0: the side-3 variant.
      static const int kmix_inchain = dbg_env("MIMRL_KMIX_PG_INCHAIN") ? atoi(dbg_env("MIMRL_KMIX_PG_INCHAIN")) : 1;   // tuning knob
      if (defer && kmix_inchain) {
        MX(kmix_bwd(stream, b.l.z, gbuf[cur], gbuf[q], kw, (long)B * ol, id));
      } else if (defer) {   // data gradient on the chain; gbuf[cur] stays alive in deferred mode
        MX(kmix_bwd_part(stream, b.l.z, gbuf[cur], gbuf[q], kw, (long)B * ol, id, 1));
        static const int kdbg = dbg_env("MIMRL_DBG_KMIX") ? atoi(dbg_env("MIMRL_DBG_KMIX")) : 0;
        static const bool kmix_park = dbg_env("MIMRL_KMIX_PG_PARKED") != nullptr;   // debugging: the round-2a placement (not reproducible!)
        if (kmix_park) {
          Deferred d{5, 2, GemmDesc(), b.l.z, (long)B * ol, id, 0, 0, nullptr};
          d.p3 = gbuf[cur]; d.kw = kw; d.kw.dbg = kdbg;
          deferred.push_back(d);
        } else {
          // The parameter gradients (same arithmetic recomputed) start RIGHT AWAY on side 3, beside the rest of the data-gradient chain,
          // and model_backward makes the BPTT wait for side 3: this kernel must never be resident next to gru_bwd_kernel -- beside it
          // the block-0 K-axis gradients came out 5-30 % off in most runs (see MIMRL_EARLY_FLUSH below and tools/kaxis_vals.py)
          KMixW kp = kw; kp.dbg = kdbg;
          MX(fork(3, 3));
          MX(kmix_bwd_part(S(3), b.l.z, gbuf[cur], nullptr, kp, (long)B * ol, id, 2));
          kmix_pg_on_side3 = true;
        }
      } else {
        MX(kmix_bwd(stream, b.l.z, gbuf[cur], gbuf[q], kw, (long)B * ol, id));
      }
      release(cur);
      cur = q;
    }
    // ------------------------------------------------ L axis backward (per-sample [.,C] tiles, C = ik*id)
    if (bf16 && fused_cube_bwd && !cfg.ln_first && pl <= 0.f && w.ax[0].res >= 0 && laxis_bwd_supported(il, hl, ol, ik * id)) {
      // one launch: LayerNorm(L) backward -> dY -> dU -> dX (+ LayerNorm and bias gradients); weight gradients stay GEMMs
      const AxisW& a = w.ax[0];
      const long C = (long)ik * id;
      GRAB(i_dy); GRAB(i_du); GRAB(i_dx);
      LAxisBwdArgs fa;
      fa.dz = gbuf[cur]; fa.y = b.l.y; fa.mean = b.l.mean; fa.rstd = b.l.rstd; fa.gamma = P(a.ln_g); fa.u = b.l.u;
      fa.w2 = P(a.fc2.w); fa.w1 = P(a.fc1.w); fa.wr = P(a.res);
      fa.dy = gbuf[i_dy]; fa.du = gbuf[i_du]; fa.dx = gbuf[i_dx];
      fa.db2 = a.fc2.b >= 0 ? Gm(a.fc2.b) : nullptr; fa.db1 = a.fc1.b >= 0 ? Gm(a.fc1.b) : nullptr;
      fa.B = B; fa.il = il; fa.hl = hl; fa.ol = ol; fa.C = (int)C; fa.act = cfg.activation;
      MX(laxis_bwd_fused(stream, fa));
      MX(W_fork(1, 3));
      MX(W_lnpar(1, b.l.y, b.l.mean, b.l.rstd, gbuf[cur], Gm(a.ln_g), Gm(a.ln_b), B, ol, (int)C));
      release(cur);
      { GemmDesc g; g.A = gbuf[i_dy]; g.sa_m = C; g.sa_k = 1; g.sa_b = (long)ol * C;        // dW2 += dY_b . H_b^T
        g.B = b.l.h; g.sb_k = 1; g.sb_n = C; g.sb_b = (long)hl * C;
        g.C = Gm(a.fc2.w); g.sc_m = hl; g.sc_n = 1; g.sc_b = 0; g.M = ol; g.N = hl; g.K = (int)C; g.batch = B; g.atomic = 1;
        MX(W_gemm(1, g)); }
      { GemmDesc g; g.A = gbuf[i_dy]; g.sa_m = C; g.sa_k = 1; g.sa_b = (long)ol * C;        // dWr += dY_b . X_b^T
        g.B = xblk; g.sb_k = 1; g.sb_n = C; g.sb_b = (long)il * C;
        g.C = Gm(a.res); g.sc_m = il; g.sc_n = 1; g.sc_b = 0; g.M = ol; g.N = il; g.K = (int)C; g.batch = B; g.atomic = 1;
        MX(W_gemm(2, g)); }
      { GemmDesc g; g.A = gbuf[i_du]; g.sa_m = C; g.sa_k = 1; g.sa_b = (long)hl * C;        // dW1 += dU_b . X_b^T
        g.B = xblk; g.sb_k = 1; g.sb_n = C; g.sb_b = (long)il * C;
        g.C = Gm(a.fc1.w); g.sc_m = il; g.sc_n = 1; g.sc_b = 0; g.M = hl; g.N = il; g.K = (int)C; g.batch = B; g.atomic = 1;
        MX(W_gemm(3, g)); }
      MX(W_join(1, 3));
      release(i_dy); release(i_du);
      cur = i_dx;
    } else {
      const AxisW& a = w.ax[0];
      const long C = (long)ik * id;
      const float* xmlp = cfg.ln_first ? b.l.xn : xblk;
      int i_dy = cur;
      if (!cfg.ln_first) {
        GRAB(q);
        MX(colln_bwd(stream, b.l.y, P(a.ln_g), b.l.mean, b.l.rstd, gbuf[cur], gbuf[q], Gm(a.ln_g), Gm(a.ln_b), B, ol, (int)C));
        release(cur);
        i_dy = q;
      }
      const float* dy = gbuf[i_dy];
      int i_dym = i_dy;
      if (pl > 0.f) {
        GRAB(q);
        HIPX(hipMemcpyAsync(gbuf[q], dy, sizeof(float) * B * ol * C, hipMemcpyDeviceToDevice, stream));
        MX(dropout_inplace(stream, gbuf[q], (long)B * ol * C, pl, key(), 10 + 3 * i));
        i_dym = q;
      }
      const float* dym = gbuf[i_dym];
      // dW2[ol,hl] += sum_b dYm_b[ol,C] . H_b[hl,C]^T
      MX(W_fork(1, 2));
      { GemmDesc g; g.A = dym; g.sa_m = C; g.sa_k = 1; g.sa_b = (long)ol * C;
        g.B = b.l.h; g.sb_k = 1; g.sb_n = C; g.sb_b = (long)hl * C;
        g.C = Gm(a.fc2.w); g.sc_m = hl; g.sc_n = 1; g.sc_b = 0; g.M = ol; g.N = hl; g.K = (int)C; g.batch = B; g.atomic = 1;
        MX(W_gemm(1, g)); }
      if (a.fc2.b >= 0) MX(W_rowsum(1, dym, B, ol, (int)C, Gm(a.fc2.b)));
      if (a.res >= 0) {
        GemmDesc g; g.A = dy; g.sa_m = C; g.sa_k = 1; g.sa_b = (long)ol * C;
        g.B = xblk; g.sb_k = 1; g.sb_n = C; g.sb_b = (long)il * C;
        g.C = Gm(a.res); g.sc_m = il; g.sc_n = 1; g.sc_b = 0; g.M = ol; g.N = il; g.K = (int)C; g.batch = B; g.atomic = 1;
        MX(W_gemm(2, g));
      }
      GRAB(i_du);                                            // dU_b[hl,C] = (W2^T . dYm_b) * act'(U)
      { GemmDesc g; g.A = P(a.fc2.w); g.sa_m = 1; g.sa_k = hl; g.sa_b = 0;
        if (bf16 && w2p[i] && w2p_valid[i] && hl % 4 != 0) { g.A = w2p[i]; g.sa_k = (hl + 3) & ~3; g.a_pad4 = 1; }
        g.B = dym; g.sb_k = C; g.sb_n = 1; g.sb_b = (long)ol * C;
        g.C = gbuf[i_du]; g.sc_m = C; g.sc_n = 1; g.sc_b = (long)hl * C; g.M = hl; g.N = (int)C; g.K = ol; g.batch = B;
        g.act = cfg.activation; g.gradact_u = b.l.u;
        MX(G_(g)); }
      MX(W_fork(3, 3));
      { GemmDesc g; g.A = gbuf[i_du]; g.sa_m = C; g.sa_k = 1; g.sa_b = (long)hl * C;
        g.B = xmlp; g.sb_k = 1; g.sb_n = C; g.sb_b = (long)il * C;
        g.C = Gm(a.fc1.w); g.sc_m = il; g.sc_n = 1; g.sc_b = 0; g.M = hl; g.N = il; g.K = (int)C; g.batch = B; g.atomic = 1;
        MX(W_gemm(3, g)); }
      if (a.fc1.b >= 0) MX(W_rowsum(3, gbuf[i_du], B, hl, (int)C, Gm(a.fc1.b)));
      GRAB(i_dx0);                                           // dX_b[il,C] = W1^T . dU_b (+LN-first bwd) + Wr^T . dY_b
      int i_dx = i_dx0;
      { GemmDesc g; g.A = P(a.fc1.w); g.sa_m = 1; g.sa_k = il; g.sa_b = 0;
        g.B = gbuf[i_du]; g.sb_k = C; g.sb_n = 1; g.sb_b = (long)hl * C;
        g.C = gbuf[i_dx]; g.sc_m = C; g.sc_n = 1; g.sc_b = (long)il * C; g.M = il; g.N = (int)C; g.K = hl; g.batch = B;
        if (a.res >= 0 && !cfg.ln_first) {                   // + Wr^T . dY_b in the same launch
          g.A2 = P(a.res); g.sa2_m = 1; g.sa2_k = il; g.sa2_b = 0;
          g.B2 = dy; g.sb2_k = C; g.sb2_n = 1; g.sb2_b = (long)ol * C; g.K2 = ol;
        }
        MX(G_(g)); }
      if (cfg.ln_first) {
        MX(join(3, 3));                                      // dW1 / db1 on side 3 still read the dU buffer
        MX(colln_bwd(stream, xblk, P(a.ln_g), b.l.xn_mean, b.l.xn_rstd, gbuf[i_dx], gbuf[i_du], Gm(a.ln_g), Gm(a.ln_b), B, il, (int)C));
        release(i_dx);
        i_dx = i_du;
      } else {
        release(i_du);
      }
      if (a.res >= 0 && !cfg.ln_first) {
        // fused above
      } else if (a.res >= 0) {
        GemmDesc g; g.A = P(a.res); g.sa_m = 1; g.sa_k = il; g.sa_b = 0;
        g.B = dy; g.sb_k = C; g.sb_n = 1; g.sb_b = (long)ol * C;
        g.C = gbuf[i_dx]; g.sc_m = C; g.sc_n = 1; g.sc_b = (long)il * C; g.M = il; g.N = (int)C; g.K = ol; g.batch = B; g.beta = 1.f;
        MX(G_(g));
      } else {
        MX(add_inplace(stream, gbuf[i_dx], dy, (long)B * il * C));
      }
      MX(W_join(1, 3));
      if (i_dym != i_dy) release(i_dym);
      release(i_dy);
      cur = i_dx;
    }
    // Hand this block's parked parameter-gradient work to ONE side stream right away: it then overlaps the rest of the data-gradient
    // chain instead of queueing up beside the BPTT (1 = last block only, 2 = every block, 0 = everything behind the chain).
    // DEFAULT 2 SINCE ROUND 2b, FOR CORRECTNESS: with 0 the K-axis parameter-gradient kernel (kmix_bwd<MODE 2>) ran beside the
    // layer-1 BPTT and its results were NOT reproducible -- block-0 K-axis gradients off by 5-30 % in most runs, every other tensor
    // exact (tools/determinism.py, tools/kaxis_vals.py).  Established by elimination: exact with the parked kernels on the main
    // stream, with a join in front of the BPTT, or flushed early; wrong only while gru_bwd_kernel is resident next to it; device-scope
    // loads of its inputs repair two of three components.  The mechanism is not understood (no out-of-bounds LDS / global write was
    // found in either kernel); until it is, nothing register-heavy runs beside the recurrence.  tests/test_gpu_step.py::
    // test_stage2_gradients_reproducible pins it.  Speed: neutral at cfg2 (0.984 vs 0.986 ms).
    static const int early = dbg_env("MIMRL_EARLY_FLUSH") ? atoi(dbg_env("MIMRL_EARLY_FLUSH")) : 2;
    if (defer && (early == 2 || (early == 1 && i == cfg.n_blocks - 1))) MX(flush_deferred(1));
  }
#undef GRAB
  *cur_out = cur;
  return MIMRL_OK;
}

// Critical-path probe: with MIMRL_DBG_DELAY_TAG=<n> a single-wave kernel that spins MIMRL_DBG_DELAY_US (default 50)
// microseconds is enqueued behind phase <n> on that phase's stream.  Step-time increase / injected time = how much of
// that phase sits on the critical path of the captured graph (tools/critical_path.sh); costs nothing when unset.
__global__ void dbg_spin_kernel(long ticks) {
  const long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(16);
}
int mimrl_handle::dbg_delay(hipStream_t st, int tag) {
  static const int want = knob("MIMRL_DBG_DELAY_TAG") ? atoi(knob("MIMRL_DBG_DELAY_TAG")) : -1;
  static const int us = knob("MIMRL_DBG_DELAY_US") ? atoi(knob("MIMRL_DBG_DELAY_US")) : 50;
  if (tag != want) return MIMRL_OK;
  hipLaunchKernelGGL(dbg_spin_kernel, dim3(1), dim3(64), 0, st, (long)us * 100);   // wall_clock64 ticks at 100 MHz
  LAUNCH_CHECK();
  return MIMRL_OK;
}

// `after`: an event recorded earlier on the main stream; the side streams then wait for THAT point instead of the main
// stream's current position (lets the caller enqueue -- and, in a captured graph, order -- main-stream work in front of the
// parked kernels without making them depend on it)
int mimrl_handle::flush_deferred(int only_side, hipEvent_t after) {
  if (deferred.empty()) return MIMRL_OK;
  if (after && multi_stream) {
    for (int i = 1; i <= 3; ++i) if (side_on(i)) HIPX(hipStreamWaitEvent(side[i], after, 0));
  } else if (only_side > 0) MX(fork(only_side, only_side)); else MX(fork(1, 3));
  for (int q = 1; q <= 3; ++q) MX(dbg_delay(S(q), 9));
  static const int wg_sides = knob("MIMRL_WG_SIDES") ? atoi(knob("MIMRL_WG_SIDES")) : 3;
  static const int dbg_skip_kinds = dbg_env("MIMRL_DBG_SKIP_DEFERRED") ? atoi(dbg_env("MIMRL_DBG_SKIP_DEFERRED")) : 0;   // timing experiments only (bit = kind)
  // the weight-gradient GEMMs as (at most) two grouped split-K launches, one per operand-layout class: D-axis products are
  // (RC,RC), the batch-reduced L-axis products (KC,KC).  Alone each is a ~20 us launch of 4..64 tiles.
  static const bool no_wg_groupk = knob("MIMRL_NO_WG_GROUPK") != nullptr;   // tuning knob
  // (short sequences only: at T = 1000 the recurrence beside them runs for a millisecond, launch latencies are hidden and one
  // chip-filling launch in front of the BPTT costs more than it saves -- cfg5: 3.85 vs 3.74 ms)
  const bool groupk = !no_wg_groupk && !prof_on && bf16 && !((dbg_skip_kinds >> 0) & 1) && cfg.seq_len <= 128;
  if (groupk) {
    std::vector<GemmDesc> cls[2];
    for (const Deferred& d : deferred) if (d.kind == 0) cls[d.g.sa_k == 1 ? 0 : 1].push_back(d.g);
    for (int c = 0; c < 2; ++c) {
      hipStream_t st = only_side > 0 ? S(only_side) : S(1 + c % wg_sides);
      for (size_t i = 0; i < cls[c].size(); i += 12) MX(gemm_group_splitk(st, cls[c].data() + i, (int)std::min<size_t>(12, cls[c].size() - i), bf16));
    }
  }
  int rr = 2;
  for (const Deferred& d : deferred) {
    if ((dbg_skip_kinds >> d.kind) & 1) continue;
    if (groupk && d.kind == 0) continue;
    static const bool dbg_defer_main = dbg_env("MIMRL_DBG_DEFER_MAIN") != nullptr;   // debugging: parked non-GEMM kernels on the main stream
    hipStream_t st = dbg_defer_main ? stream : only_side > 0 ? S(only_side) : S(1 + (groupk ? rr++ : d.side - 1) % wg_sides);
    if (d.kind == 0) MX(G_on(st, d.g));
    else if (d.kind == 1) MX(colsum(st, d.src, d.n0, (int)d.n1, (int)d.n2, d.dst));
    else if (d.kind == 2) MX(rowsum_batched(st, d.src, (int)d.n0, (int)d.n1, (int)d.n2, d.dst));
    else if (d.kind == 3) MX(colln_param_grads(st, d.src, d.p1, d.p2, d.p3, d.dst, d.dst2, (int)d.n0, (int)d.n1, (int)d.n2));
    else if (d.kind == 4) MX(rowln_param_grads(st, d.src, d.p1, d.p2, d.p3, d.dst, d.dst2, d.n0, (int)d.n1));
    else if (d.kind == 6) MX(daxis_param_grads(st, d.src, d.p1, d.p2, d.p3, d.p4, d.p5, d.dst, d.dst2, d.dst3, d.dst4, d.n0));
    else MX(kmix_bwd_part(st, d.src, d.p3, nullptr, d.kw, d.n0, (int)d.n1, 2));
  }
  deferred.clear();
  return MIMRL_OK;
}

// =================================================================================================
// model backward (stage 2): needs dfeat (F,T,A,V contributions of the estimators) and dpred
// =================================================================================================
int mimrl_handle::model_backward() {
  Range rg("mimrl.model_backward");
  const int B = cfg.batch, T = cfg.seq_len, L = cfg.time_len, D = cfg.d_common;
  const long BT_ = (long)B * T;
  const int nb = cfg.n_blocks;
  const int ol = cfg.d_outs[nb - 1][0], ok = cfg.d_outs[nb - 1][1], od = cfg.d_outs[nb - 1][2];
  // head backward -> gradient of the last cube output (in gbuf[0])
  MX(head_bwd(stream, dfeat, dpred, P(cls_w), bufs.feats, gbuf[0], Gm(cls_w), Gm(cls_b), B, ol, ok, od,
              cfg.compose_t_sum, cfg.compose_k_sum, head_gather_on ? &head_gather : nullptr));
  int ci = 0;
  deferred.clear();
  { Scope sc(this, MIMRL_PH_CUBE_BWD); MX(cube_backward(0, &ci)); }
  MX(dbg_delay(stream, 7));
  return encoders_backward(gbuf[ci]);
}

// Everything of the backward pass in front of the CubeMLP: text dropout + W_t gradient, LayerNorm / ReLU / dropout of both recurrent
// encoders, BPTT of both bi-GRU layers and their weight gradients (Model.py:395-466 under autograd).  dcube [B, L, 3, D] = gradient of
// the stacked cube input; the T_F / A_F / V_F mean gradients are read from dfeat.  (Also the body of mimrl_probe_encoders.)
int mimrl_handle::encoders_backward(float* dcube) {
  const int B = cfg.batch, T = cfg.seq_len, L = cfg.time_len, D = cfg.d_common;
  const long BT_ = (long)B * T;
  // T_F/A_F/V_F means (Model.py:466): dcube[b,t,k,:] += dfeat[1+k][b,:]/T -- folded into the two consumers of dcube below
  // (no feat_mean_bwd launch on the chain).  The critical consumer goes first in capture order; the text branch has slack.
  const float* dmean = dfeat + (size_t)B * D;      // [3][B, D]: gradients of T_F, A_F, V_F
  // capture order of the two consumers (graph nodes are dispatched in capture order): the LayerNorm backward, head of the critical
  // BPTT chain, first; the text branch (35 us of W_t weight gradient with slack until the end of the stage) behind it.  History: while
  // the side streams were congested by the parked CubeMLP weight gradients the opposite order was faster (1.229 vs 1.259 ms); with the
  // grouped / fused parameter-gradient kernels it is this one (0.980 vs 0.988 ms).  MIMRL_TEXT_BWD_FIRST=1: the other order.
  static const bool text_bwd_first = knob("MIMRL_TEXT_BWD_FIRST") != nullptr;
  auto text_bwd = [&]() -> int {   // text branch (side 0): dW_t = dtx^T . text
    MX(fork(0, 0));
    MX(text_post_bwd(S(0), dcube, dtx, B, T, L, 3, D, 0, cfg.dropout[0], key(), 0, dmean));
    { GemmDesc g = gemm_tn(dtx, D, bufs.text, cfg.d_t, Gm(w_t), cfg.d_t, D, cfg.d_t, (int)BT_); g.atomic = 1; MX(G_on(S(0), g)); }
    return MIMRL_OK;
  };
  if (text_bwd_first) MX(text_bwd());
  if (head_gather_on && ev_dmean) { HIPX(hipStreamWaitEvent(stream, ev_dmean, 0)); ev_dmean = nullptr; }   // dmean gathered on side 0
  // audio / video: LN+ReLU+dropout backward -> ds (shared by both directions of layer 1)
  {
    LnSide2 sd[2];
    for (int m = 0; m < 2; ++m)
      sd[m] = LnSide2{h1[m], P(ln_g[m]), P(ln_b[m]), ln_mean[m], ln_rstd[m], ds[m], Gm(ln_g[m]), Gm(ln_b[m]), 1 + m,
                      cfg.dropout[1 + m], (uint32_t)(1 + m)};
    MX(ln_relu_drop_bwd2(stream, sd[0], sd[1], dcube, B, T, L, 3, D, key(), dmean + (size_t)B * D, dmean + 2 * (size_t)B * D));
  }
  if (!text_bwd_first) MX(text_bwd());
  // CubeMLP weight gradients: side 1..3, beside the layer-1 BPTT.  Tuning knob MIMRL_BPTT_FIRST=1 captures the BPTT launch in
  // front of the parked kernels (graph nodes are dispatched in capture order).  Measured on cfg2: 1.58 vs 1.36 ms -- the
  // recurrence is latency-bound and loses more to the weight-gradient kernels sharing its CUs from the first cell step on
  // than the ~100 us it waits behind their first wave; default off.
  // debugging: make the main stream wait for sides 1..3 (the parked kernels) at point n: 1 before the BPTT, 2 behind the layer-1 BPTT,
  // 3 behind the dh0 product, 4 behind the layer-0 BPTT
  static const int dbg_join_at = dbg_env("MIMRL_DBG_JOIN_AT") ? atoi(dbg_env("MIMRL_DBG_JOIN_AT")) : 0;
  static const bool bptt_first = knob("MIMRL_BPTT_FIRST") != nullptr;
  ev_pre = nullptr;
  if (bptt_first && multi_stream && cfg.encoder == MIMRL_ENCODER_GRU && !deferred.empty()) {
    MX(next_event(&ev_pre));
    HIPX(hipEventRecord(ev_pre, stream));
  } else {
    MX(flush_deferred());
    if (dbg_join_at == 1) MX(join(1, 3));
  }
  if (cfg.encoder != MIMRL_ENCODER_GRU) kmix_pg_on_side3 = false;   // (joined with every other side at the end of those paths)
  if (cfg.encoder == MIMRL_ENCODER_CONV) {
    MX(conv_backward());
    MX(join(0, 5));
    return MIMRL_OK;
  }
  if (cfg.encoder == MIMRL_ENCODER_LSTM) {
    MX(lstm_encoders_backward());
    MX(join(0, 5));
    return MIMRL_OK;
  }
  if (kmix_pg_on_side3) { MX(join(3, 3)); kmix_pg_on_side3 = false; }   // (long finished by now: they started beside the CubeMLP chain)
  MX(gru_layer_backward(1));
  if (split_part == 1) return join(0, 5);   // data parallel, split reduce: everything but the layer-0 GRU gradients is final here
  MX(gru_layer_backward(0));
  return join(0, 5);
}

// BPTT of one bi-GRU layer (both modalities, both directions) + its weight gradients (+ the gradient to the layer below)
int mimrl_handle::gru_layer_backward(int l) {
  const int B = cfg.batch, T = cfg.seq_len;
  const long BT_ = (long)B * T;
  static const int dbg_join_at = dbg_env("MIMRL_DBG_JOIN_AT") ? atoi(dbg_env("MIMRL_DBG_JOIN_AT")) : 0;
  const float* xin[2] = {bufs.audio, bufs.video};
  {
    GruBwdArgs a;
    a.B = B; a.T = T; a.out_ld = 2 * H; a.nmod = 2;
    a.dout_ld = l == 1 ? H : 2 * H; a.dout_off = l == 1 ? 0 : H;
    a.btv = gru_pick_btv(B, 2);
    a.stamp = kstamp; a.stamp.id = l == 1 ? 2 : 3;
    // layer 0 without packed inputs keeps fp32 dg: its dW_ih product reads the caller's unaligned [rows, 74 / 35] inputs through
    // the generic kernel, which has no bf16-operand variant
    const bool lbf = dg_bf16 && (l == 1 || l0_packed || l0_bwd_pack);
    a.dg_bf16 = lbf ? 1 : 0;
    a.slab_upl = l == 0 && l0_xin ? 2 : 0;   // the fused-projection forward wrote the 4-wave slab layout whatever MIMRL_GRU_WAVES says
    for (int m = 0; m < 2; ++m) {
      a.lens[m] = lens[m];
      for (int d = 0; d < 2; ++d) {
        const GruDirW& g = gru[m][l][d];
        a.seq[m][d] = GruSeqBwd{P(g.w_hh), sv[l][m][d], l == 1 ? h1[m] : h0[m], l == 1 ? ds[m] : dh0[m], dg[l][m][d],
                                hprev[l][m][d], Gm(g.b_ih), Gm(g.b_hh)};
      }
    }
    { Scope sc(this, MIMRL_PH_GRU_BWD); MX(gru_backward(stream, a, (prec & MIMRL_PREC_BF16_GRU_BWD) != 0)); }
    if ((l == 1 && dbg_join_at == 2) || (l == 0 && dbg_join_at == 4)) MX(join(1, 3));
    if (l == 1 && ev_pre) MX(flush_deferred(0, ev_pre));
    MX(dbg_delay(stream, 8));
    // side streams of the GRU weight gradients (tuning knobs).  Sides 1..3 still carry the parked CubeMLP parameter-gradient
    // kernels at this point; sides 0 (text branch), 4 and 5 (kNN sampler, CMI branch) have been idle since the forward pass.
    static const int l0_side = knob("MIMRL_L0_WG_SIDE") ? atoi(knob("MIMRL_L0_WG_SIDE")) : 1;
    static const int l1_side0 = knob("MIMRL_L1_WG_SIDE") ? atoi(knob("MIMRL_L1_WG_SIDE")) : 1;
    MX(fork(1, (l == 0 || l1_side0 == 4) ? 5 : 3));   // the weight gradients below depend on the BPTT only
    if (l0_side == 0 && l == 0) MX(fork(0, 0));
    static const bool dh0_last = knob("MIMRL_DH0_LAST") != nullptr;   // tuning knob: capture order of dh0 vs the side-stream weight gradients
    auto dh0_gemm = [&]() -> int {   // gradient to the layer-0 outputs, dh0 = sum_dir dgx_dir . W_ih_l1_dir
      // one dual-product GEMM (both directions accumulate in the same output tile), batch = modality
      {
        GemmDesc q = gemm_nn(dg[l][0][0], 4 * H, P(gru[0][l][0].w_ih), 2 * H, dh0[0], 2 * H, (int)BT_, 2 * H, G);
        q.A2 = dg[l][0][1]; q.sa2_m = 4 * H; q.sa2_k = 1;
        q.B2 = P(gru[0][l][1].w_ih); q.sb2_k = 2 * H; q.sb2_n = 1; q.K2 = G;
        q.batch = 2;
        q.sa_b = dg[l][1][0] - dg[l][0][0]; q.sa2_b = dg[l][1][1] - dg[l][0][1];
        q.sb_b = gru[1][l][0].w_ih - gru[0][l][0].w_ih; q.sb2_b = gru[1][l][1].w_ih - gru[0][l][1].w_ih;
        q.sc_b = dh0[1] - dh0[0];
        if (lbf) { q.a_bf16 = 1; q.sa_b *= 2; q.sa2_b *= 2; }   // buffer distances are fp32-element counts; bf16 elements: x2
        if (lbf && w1_img_valid && h16_on && w1b) {   // the weights from the bf16 image of this step's forward pass: half the B bytes
          q.B = reinterpret_cast<const float*>(w1b); q.B2 = reinterpret_cast<const float*>(w1b + (long)G * 2 * H);
          q.b_bf16 = 1; q.sb_b = 2L * G * 2 * H; q.sb2_b = 2L * G * 2 * H;
          // B * T >= 4096 rows: both operands k-contiguous -- the transposed image of the same bf16 values, the two
          // directions as two k-segments of one [256, 768] matrix per modality -- so that the LDS-DMA kernel of gemm_tall.hip takes it
          if (w1bt) {
            GemmDesc t = q;
            t.B = reinterpret_cast<const float*>(w1bt); t.B2 = reinterpret_cast<const float*>(w1bt + G);
            t.sb_k = 1; t.sb_n = 2 * G; t.sb2_k = 1; t.sb2_n = 2 * G; t.sb_b = 2L * H * 2 * G; t.sb2_b = 2L * H * 2 * G;
            if (gemm_tall_ok(t)) q = t;
          }
        }
        MX(G_(q));
      }
      return MIMRL_OK;
    };
    if (l == 1 && !dh0_last) MX(dh0_gemm());
    if (l == 1 && dbg_join_at == 3) MX(join(1, 3));
    // weight gradients of this layer: off the critical path.  Layer 1: side 1..3 (they overlap the layer-0 BPTT);
    // layer 0 is the tail of the stage.
    if (l == 0 && (l0_packed || l0_bwd_pack)) {
      // layer 0: the W_ih (against the packed inputs) and W_hh gradients of all four (modality, direction) pairs as two batched
      // launches into packed scratch, scattered into the bucket by one small kernel: 3 launches on 2 streams close the stage
      // instead of four GEMMs in a row (the per-modality widths 74 / 35 ruled out both batching and 16-byte loads)
      const long s_dg = dg[0][0][1] - dg[0][0][0], o_dg = dg[0][1][0] - dg[0][0][0];
      const long s_hp = hprev[0][0][1] - hprev[0][0][0], o_hp = hprev[0][1][0] - hprev[0][0][0];
      { GemmDesc q = gemm_tn(dg[0][0][0], 4 * H, xpack, KP(), dwih_pack, KP(), G, KP(), (int)BT_);
        q.batch = 4; q.batch_in = 2; q.sa_b = s_dg; q.sa_bo = o_dg; q.sb_b = 0; q.sb_bo = BT_ * KP(); q.sc_b = (long)G * KP(); q.sc_bo = 2L * G * KP();
        if (lbf) { q.a_bf16 = 1; q.sa_b *= 2; q.sa_bo *= 2; }
        if (lbf && xpack16) { q.B = reinterpret_cast<const float*>(reinterpret_cast<const __bf16*>(xpack + BT_ * KP())); q.b_bf16 = 1; }   // the bf16 copy of the packed inputs
        q.atomic = 1; MX(G_on(stream, q)); }
      { GemmDesc q = gemm_tn(dg[0][0][0], 4 * H, hprev[0][0][0], H, dwhh_pack, H, G, H, (int)BT_);
        q.a_gap_at = 2 * H; q.a_gap_rows = H;
        q.batch = 4; q.batch_in = 2; q.sa_b = s_dg; q.sa_bo = o_dg; q.sb_b = s_hp; q.sb_bo = o_hp; q.sc_b = (long)G * H; q.sc_bo = 2L * G * H;
        if (lbf) { q.a_bf16 = q.b_bf16 = 1; q.sa_b *= 2; q.sa_bo *= 2; q.sb_b *= 2; q.sb_bo *= 2; }
        q.atomic = 1; MX(G_on(S(l0_side), q)); }
      MX(join(l0_side, l0_side));
      L0Unpack up;
      for (int m = 0; m < 2; ++m) {
        up.d[m] = gru[m][0][0].din;
        for (int d = 0; d < 2; ++d) { up.g_ih[m][d] = Gm(gru[m][0][d].w_ih); up.g_hh[m][d] = Gm(gru[m][0][d].w_hh); }
      }
      up.dwih_pack = dwih_pack; up.dwhh_pack = dwhh_pack; up.KP = KP();
      if (fold_unpack) unpack_pending = true;      // the Adam launch behind this pass takes the packed pieces itself (enqueue_apply)
      else MX(l0_unpack_grads(stream, up));
      return MIMRL_OK;
    }
    int rr = 0;
    for (int m = 0; m < 2; ++m) {
      // dg rows are [dr'|dz'|dn'|dn'r]: dgx = columns [0,3H); dgh = columns [0,2H) and [3H,4H).  Both directions in one
      // launch each (batch = direction): dW_ih and dW_hh (the latter reads dg through a row gap).
      const float* in = l == 0 ? xin[m] : h0[m];
      const GruDirW &gf = gru[m][l][0], &gr = gru[m][l][1];
      const long s_dg = dg[l][m][1] - dg[l][m][0], s_hp = hprev[l][m][1] - hprev[l][m][0];
      const bool both = l == 1;            // layer 1: same shapes for audio and video -> batch = (modality, direction)
      if (both && m == 1) break;
      auto two = [&](GemmDesc& q, long a_o, long b_o, long c_o) { if (both) { q.batch = 4; q.batch_in = 2; q.sa_bo = a_o; q.sb_bo = b_o; q.sc_bo = c_o; } };
      const long o_dg = dg[l][1][0] - dg[l][0][0], o_hp = hprev[l][1][0] - hprev[l][0][0], o_in = both ? h0[1] - h0[0] : 0;
      const long o_wih = gru[1][l][0].w_ih - gru[0][l][0].w_ih, o_whh = gru[1][l][0].w_hh - gru[0][l][0].w_hh;
      static const int tail_n = knob("MIMRL_TAIL_STREAMS") ? atoi(knob("MIMRL_TAIL_STREAMS")) : 1;   // tuning knobs
      static const int wg_sides = knob("MIMRL_WG_SIDES") ? atoi(knob("MIMRL_WG_SIDES")) : 3;
      auto pick = [&]() { if (l == 0) { const int q = rr++ % tail_n; return q == 0 ? stream : S(q); } return l1_side0 == 4 ? S(4 + rr++ % 2) : S(1 + rr++ % wg_sides); };
      { GemmDesc q = gemm_tn(dg[l][m][0], 4 * H, in, gf.din, Gm(gf.w_ih), gf.din, G, gf.din, (int)BT_);
        q.batch = 2; q.sa_b = s_dg; q.sb_b = 0; q.sc_b = gr.w_ih - gf.w_ih; q.atomic = 1; two(q, o_dg, o_in, o_wih);
        if (lbf) { q.a_bf16 = 1; q.sa_b *= 2; q.sa_bo *= 2; }
        MX(G_on(pick(), q)); }
      { GemmDesc q = gemm_tn(dg[l][m][0], 4 * H, hprev[l][m][0], H, Gm(gf.w_hh), H, G, H, (int)BT_);   // dgh = dg columns [0,2H) u [3H,4H)
        q.a_gap_at = 2 * H; q.a_gap_rows = H;
        q.batch = 2; q.sa_b = s_dg; q.sb_b = s_hp; q.sc_b = gr.w_hh - gf.w_hh; q.atomic = 1; two(q, o_dg, o_hp, o_whh);
        if (lbf) { q.a_bf16 = q.b_bf16 = 1; q.sa_b *= 2; q.sa_bo *= 2; q.sb_b *= 2; q.sb_bo *= 2; }
        MX(G_on(pick(), q)); }
    }
    if (l == 1 && dh0_last) MX(dh0_gemm());
  }
  return MIMRL_OK;
}

// Conv1d encoder backward: the inputs are data, so only dW[:, :, tap] = sum_b dy_b[rows]^T x_b[shifted rows] and the bias
// gradient are needed (three batch-reduced GEMMs per modality, off the critical path by construction: nothing follows)
int mimrl_handle::conv_backward() {
  const int B = cfg.batch, T = cfg.seq_len;
  const float* xin[2] = {bufs.audio, bufs.video};
  const int dmod[2] = {cfg.d_a, cfg.d_v};
  MX(fork(1, 3));
  int rr = 0;
  for (int m = 0; m < 2; ++m) {
    const int d = dmod[m];
    for (int tap = 0; tap < 3; ++tap) {
      const int rows = tap == 1 ? T : T - 1;
      if (rows <= 0) continue;
      GemmDesc g;
      g.A = ds[m] + (tap == 0 ? H : 0); g.sa_m = 1; g.sa_k = H; g.sa_b = (long)T * H;
      g.B = xin[m] + (tap == 2 ? d : 0); g.sb_k = d; g.sb_n = 1; g.sb_b = (long)T * d;
      g.C = Gm(conv_w[m]) + tap; g.sc_m = 3L * d; g.sc_n = 3; g.sc_b = 0;
      g.M = H; g.N = d; g.K = rows; g.batch = B; g.atomic = 1;
      const int q = rr++ % 4;
      MX(G_on(q == 0 ? stream : S(q), g));
    }
    MX(colsum(stream, ds[m], (long)B * T, H, H, Gm(conv_b[m])));
  }
  return MIMRL_OK;
}

// =================================================================================================
// grouped MLP stacks (critic towers, concat-critic tail, CMI classifiers).  Activations are ReLU (Model.py:285).
//   layer l:  A_{l+1} = relu?( A_l W_l^T + b_l ),  A_0 = in, last layer linear.
// `brows` = rows between consecutive groups in the activation buffers (>= rows).
// =================================================================================================
int mimrl_handle::mlp_stack_forward(int nb, int rows, int brows, long p0, long pstride, int nl, const long (*l_off)[2],
                                    const int* dims, const float* in, float* const* act, float* out) {
  // concat-critic tail (thousands of row tiles): the direct-from-L2 fused variant was the faster one in round 1; with the round-2
  // GEMM kernels the plain chain wins (cfg3 9.80 vs 10.15 ms), so it is opt-in now (MIMRL_FUSED_MLP_BIG=1)
  static const bool use_big = knob("MIMRL_FUSED_MLP_BIG") != nullptr;   // tuning knob
  const bool big_ok = use_big && img_valid && crit_img && rows >= 2048 && dims[0] <= 256;
  if (bf16 && fused_mlp && (rows <= 512 || big_ok) && mlp_fused_supported(nb, rows, nl, dims)) {   // one launch (mlp_fused.hip)
    MlpFusedArgs fa;
    std::memset(&fa, 0, sizeof fa);
    fa.nb = nb; fa.rows = rows; fa.brows = brows; fa.nl = nl; fa.pstride = pstride; fa.in = in; fa.out = out;
    for (int l = 0; l <= nl; ++l) fa.dims[l] = dims[l];
    for (int l = 0; l < nl; ++l) {
      fa.W[l] = CP(p0 + l_off[l][0]); fa.b[l] = CP(p0 + l_off[l][1]);
      if (l < nl - 1) fa.act[l] = act[l];
      if (img_valid && crit_img) fa.Wb[l] = crit_img + p0 + l_off[l][0];
      if (img_valid && crit_frag && ftab.n > 0) fa.Wf[l] = crit_frag + p0 + l_off[l][0];
    }
    return mlp_stack_fwd_fused(stream, fa);
  }
  for (int l = 0; l < nl; ++l) {
    const int din_ = dims[l], dout_ = dims[l + 1];
    GemmDesc g;
    g.A = l == 0 ? in : act[l - 1]; g.sa_m = din_; g.sa_k = 1; g.sa_b = (long)brows * din_;
    g.B = CP(p0 + l_off[l][0]); g.sb_k = 1; g.sb_n = din_; g.sb_b = pstride;
    g.C = l == nl - 1 ? out : act[l]; g.sc_m = dout_; g.sc_n = 1; g.sc_b = (long)brows * dout_;
    g.M = rows; g.N = dout_; g.K = din_; g.batch = nb;
    g.bias_n = CP(p0 + l_off[l][1]); g.bias_n_b = pstride;
    g.act = l == nl - 1 ? ACT_NONE : ACT_RELU;
    MX(G_(g));
  }
  return MIMRL_OK;
}

int mimrl_handle::mlp_stack_backward(int nb, int rows, int brows, long p0, long pstride, int nl, const long (*l_off)[2],
                                     const int* dims, const float* in, float* const* act, float* dout, float* const* dtmp,
                                     float* din, bool wgrad) {
  float* dz = dout;
  int pp = 0;
  // the fused data-gradient chain runs on the transposed bf16 images (the same coalesced loop as the forward pass)
  static const bool fused_bwd = knob("MIMRL_NO_FUSED_MLP_BWD") == nullptr;
  // (stacks with thousands of row tiles -- the concat critic -- keep the GEMM chain here: measured faster than the fused one)
  const bool use_fused = bf16 && fused_mlp && fused_bwd && imgT_ready && rows <= 512 && nl <= 4 && mlp_fused_supported(nb, rows, nl, dims);
  // single-output top layer (the concat critic's score head) over many rows: one streaming kernel instead of three GEMMs with
  // one real column in 64 (dz, dW, both bias gradients)
  static const bool no_top1 = knob("MIMRL_NO_TOP1") != nullptr;   // tuning knob
  const bool top1 = !use_fused && !no_top1 && dims[nl] == 1 && nl >= 2 && dims[nl - 1] % 4 == 0 && dims[nl - 1] <= 1024 && 1024 % dims[nl - 1] == 0 &&
                    dtmp[0] != nullptr;
  if (wgrad && !use_fused && !top1)   // bias gradient of the top layer; the lower ones come out of the dA GEMM epilogues below
    MX(colsum(stream, dout, rows, dims[nl], dims[nl], CG(p0 + l_off[nl - 1][1]), nb, (long)brows * dims[nl], pstride));
  if (use_fused) {
    // the whole data-gradient chain in one launch (dtmp must hold nl-1 buffers here); weight gradients follow as GEMMs
    MlpFusedArgs fa;
    std::memset(&fa, 0, sizeof fa);
    fa.nb = nb; fa.rows = rows; fa.brows = brows; fa.nl = nl; fa.pstride = pstride; fa.in = in; fa.dout = dout; fa.din = din;
    fa.act_slack = 1;   // ta / cc / bact are carved with ACT_SLACK floats behind them
    for (int l = 0; l <= nl; ++l) fa.dims[l] = dims[l];
    for (int l = 0; l < nl; ++l) {
      fa.W[l] = CP(p0 + l_off[l][0]);
      fa.WbT[l] = crit_imgT + p0 + l_off[l][0];
      if (img_valid && crit_fragT && ftab.n > 0) { fa.WfT[l] = crit_fragT + p0 + l_off[l][0]; fa.Wb[l] = crit_img + p0 + l_off[l][0]; }
      if (l < nl - 1) { fa.act[l] = act[l]; fa.dz[l + 1] = dtmp[l]; if (wgrad) fa.db[l] = CG(p0 + l_off[l][1]); }
    }
    if (wgrad) fa.db_top = CG(p0 + l_off[nl - 1][1]);   // the top layer's bias gradient rides along (was a separate column-sum launch)
    // ... and so does the weight gradient of a narrow top layer (the 2-logit CMI head: not eligible for the grouped launch, it was a
    // 17 us generic GEMM in front of it on the CMI branch of stage 1)
    static const bool no_top_wg = knob("MIMRL_NO_TOP_WGRAD_FUSE") != nullptr;   // tuning knob
    const bool top_wg = wgrad && !no_top_wg && dims[nl] % 4 != 0 && mlp_bwd_takes_top_wgrad(fa);
    if (top_wg) fa.dw_top = CG(p0 + l_off[nl - 1][0]);
    MX(mlp_stack_bwd_fused(stream, fa));
    if (!wgrad) return MIMRL_OK;
    // the nl weight-gradient GEMMs are independent of each other: on the critical branch (wg_helper >= 0) every second
    // one goes to a helper side stream
    static const bool no_split = knob("MIMRL_NO_WG_SPLIT") != nullptr;   // tuning knob
    const int hs = (multi_stream && !no_split) ? wg_helper : -1;
    if (hs >= 0) MX(fork(hs, hs));
    GemmDesc gs[MLPF_MAX_LAYERS];
    for (int l = nl - 1; l >= 0; --l) {   // dW_l = dZ_l^T A_l
      const int din_ = dims[l], dout_ = dims[l + 1];
      GemmDesc& g = gs[nl - 1 - l];
      g = GemmDesc();
      g.A = l == nl - 1 ? dout : fa.dz[l + 1]; g.sa_m = 1; g.sa_k = dout_; g.sa_b = (long)brows * dout_;
      g.B = l == 0 ? in : act[l - 1]; g.sb_k = din_; g.sb_n = 1; g.sb_b = (long)brows * din_;
      g.C = CG(p0 + l_off[l][0]); g.sc_m = din_; g.sc_n = 1; g.sc_b = pstride;
      g.M = dout_; g.N = din_; g.K = rows; g.batch = nb;
    }
    // the nl weight-gradient products are independent of each other: ONE grouped launch (gemm_group; it falls back to nl launches
    // when a product is not eligible, e.g. the 2-row top layer of the CMI classifiers, which then goes alone)
    static const bool no_group = knob("MIMRL_NO_WG_GROUP") != nullptr;   // tuning knob: the round-1 schedule (helper stream, alternating)
    if (!no_group) {
      int lo = 0;
      if (top_wg) lo = 1;                                                             // done inside the data-gradient kernel
      else if (dims[nl] % 4 != 0) { MX(G_on(hs >= 0 ? S(hs) : stream, gs[0])); lo = 1; }   // not row-contiguous-eligible: beside the group
      MX(G_group(stream, gs + lo, nl - lo));
    } else {
      static const int helper_par = knob("MIMRL_WG_SPLIT_PARITY") ? atoi(knob("MIMRL_WG_SPLIT_PARITY")) : 0;   // tuning knob
      for (int q = top_wg ? 1 : 0; q < nl; ++q) MX(G_on((hs >= 0 && (q & 1) == helper_par) ? S(hs) : stream, gs[q]));
    }
    if (hs >= 0) MX(join(hs, hs));
    return MIMRL_OK;
  }
  static const bool no_big_side = knob("MIMRL_NO_WG_BIG_SIDE") != nullptr;   // tuning knob
  const bool big_side = wgrad && multi_stream && !no_big_side && wg_helper >= 0 && rows >= 2048 && nl <= 3;
  for (int l = nl - 1; l >= 0; --l) {
    const int din_ = dims[l], dout_ = dims[l + 1];
    const float* a_in = l == 0 ? in : act[l - 1];
    if (top1 && l == nl - 1) {
      MX(top1_bwd(stream, dz, CP(p0 + l_off[l][0]), act[l - 1], dtmp[pp], wgrad ? CG(p0 + l_off[l][0]) : nullptr,
                  wgrad ? CG(p0 + l_off[l][1]) : nullptr, wgrad ? CG(p0 + l_off[l - 1][1]) : nullptr, nb, rows, brows, din_, pstride));
      dz = dtmp[pp]; pp ^= 1;
      continue;
    }
    if (wgrad) {   // dW_l = dZ^T A_l     (one writer per tensor: plain stores into the zeroed bucket)
      GemmDesc g;
      g.A = dz; g.sa_m = 1; g.sa_k = dout_; g.sa_b = (long)brows * dout_;
      g.B = a_in; g.sb_k = din_; g.sb_n = 1; g.sb_b = (long)brows * din_;
      g.C = CG(p0 + l_off[l][0]); g.sc_m = din_; g.sc_n = 1; g.sc_b = pstride;
      g.M = dout_; g.N = din_; g.K = rows; g.batch = nb;
      // thousands of rows (concat critic: B*B per estimator): accumulate into the zeroed bucket with atomics so that the GEMM may
      // split K -- as plain stores the 5 x 16 output tiles ran 2048 k-tiles each on 80 CUs (1.6 ms per layer at cfg3)
      if (rows >= 2048) g.atomic = 1;
      // ... and they are 200+ us kernels that only READ dz_l / act_{l-1}: beside the data-gradient chain on the helper stream (no
      // gradient buffer is reused within a stack of <= 3 layers, so nothing is overwritten under them)
      if (big_side) { MX(fork(wg_helper, wg_helper)); MX(G_on(S(wg_helper), g)); }
      else MX(G_(g));
    }
    float* target = l > 0 ? dtmp[pp] : din;
    if (!target) break;
    GemmDesc g;   // dZ_{l-1} = (dZ_l W_l) * relu'(A_l)  [+ column sums -> db_{l-1}]; for l == 0: plain input gradient
    g.A = dz; g.sa_m = dout_; g.sa_k = 1; g.sa_b = (long)brows * dout_;
    g.B = CP(p0 + l_off[l][0]); g.sb_k = din_; g.sb_n = 1; g.sb_b = pstride;
    g.C = target; g.sc_m = din_; g.sc_n = 1; g.sc_b = (long)brows * din_;
    g.M = rows; g.N = din_; g.K = dout_; g.batch = nb;
    if (l > 0) {
      g.act = ACT_RELU; g.gradact_u = act[l - 1];     // post-activation > 0  <=>  pre-activation > 0
      if (wgrad) { g.colsum = CG(p0 + l_off[l - 1][1]); g.colsum_b = pstride; }
    }
    MX(G_(g));
    if (l > 0) { dz = target; pp ^= 1; }
  }
  if (big_side) MX(join(wg_helper, wg_helper));
  return MIMRL_OK;
}

// =================================================================================================
// estimators.  Three independent branches: kNN sampling (needs only banks + anchors -> launched before the model
// forward on side 4), the CMI classifiers (side 5) and the MI critics (main stream).
// =================================================================================================
// stages: 1, 2, or 3 = BOTH stages' samplers as one set of launches (overlap mode: stage 2 draws with the RNG step begin_stage(2) will set)
int mimrl_handle::knn_launch(int stages, hipStream_t st) {
  const int m = m_anchor(), k = cfg.k_neighbor;
  const float* bank[5] = {bufs.bank_f, bufs.bank_t, bufs.bank_a, bufs.bank_v, bufs.bank_c};
  KnnArgs ka;
  AnchorDraws ad;
  ka.N = bank_rows; ka.m = m; ka.k = k; ka.ncall = 0; ad.n = 0;
  for (int stage = 1; stage <= 2; ++stage) {
    if (!((stages >> (stage - 1)) & 1)) continue;
    int32_t* anc = bufs.anchors + (size_t)(stage - 1) * NE_CMI * m;
    int* idx = stage == 2 ? knn_idx2 : knn_idx;
    const unsigned ovr = bufs.knn_override ? knn_ovr_mask[stage - 1] : 0u;
    const int add = stages == 3 && stage == 2 ? 1 : rng_add;
    for (int e = 0; e < NE_CMI; ++e) {
      const int z = kCmiWire[e][2];
      KnnCall& kc = ka.call[ka.ncall++];
      kc.Z = ((ovr >> e) & 1u) ? nullptr : bank[z];     // null: the kernel leaves this call's rows alone
      kc.dz = z == FT_C ? 1 : EMB;
      kc.anchors = anc + (size_t)e * m; kc.idx_x = idx + (size_t)e * nprod();
      ad.out[ad.n] = anc + (size_t)e * m; ad.call[ad.n] = e; ad.stream_id[ad.n] = 100u + stage; ad.step_add[ad.n] = add; ++ad.n;
    }
  }
  if (cfg.device_anchors) MX(sample_anchors(st, ad, m, bank_rows, (uint32_t)cfg.seed, (uint32_t)(cfg.seed >> 32), d_ints));
  MX(knn_sample(st, ka, knn_scr[stages == 2 ? 1 : 0], knn_scr_bytes));
  for (int stage = 1; stage <= 2; ++stage) {
    if (!((stages >> (stage - 1)) & 1)) continue;
    const unsigned ovr = bufs.knn_override ? knn_ovr_mask[stage - 1] : 0u;
    int* idx = stage == 2 ? knn_idx2 : knn_idx;
    for (int e = 0; e < NE_CMI; ++e)   // caller-supplied neighbour rows (mimrl_set_knn_override_mask): copied in at every step
      if ((ovr >> e) & 1u)
        HIPX(hipMemcpyAsync(idx + (size_t)e * nprod(), bufs.knn_override + ((size_t)(stage - 1) * NE_CMI + e) * nprod(),
                            sizeof(int32_t) * nprod(), hipMemcpyDeviceToDevice, st));
  }
  return MIMRL_OK;
}

int mimrl_handle::mi_forward(int stage, bool want_grad) {
  const int B = cfg.batch;
  const size_t BD = (size_t)B * EMB;
  const bool sep = cfg.critic_type == MIMRL_CRITIC_SEPARATE;
  static const bool no_fused_mi = knob("MIMRL_NO_FUSED_MI") != nullptr;        // tuning knobs
  static const bool no_nce_tiled = knob("MIMRL_NO_MI_NCE_TILED") != nullptr;
  const bool fused_mi = sep && !no_fused_mi && (prec & MIMRL_PREC_BF16_GEMM_FWD) && (prec & MIMRL_PREC_BF16_GEMM_BWD) && mi_sep_fused_supported(B);
  const bool nce_tiled = fused_mi && !no_nce_tiled && cfg.bound_type == MIMRL_BOUND_INFONCE && !has_baseline();
  {   // tower inputs: x operand -> slot 2e, y operand -> slot 2e+1
    CopyTable t;
    t.n = 10;
    for (int e = 0; e < NE_MI; ++e)
      for (int sd = 0; sd < 2; ++sd) {
        t.src[e * 2 + sd] = bufs.feats + kMiWire[e][sd] * BD;
        t.dst[e * 2 + sd] = tin + (e * 2 + sd) * BD;
      }
    if (nce_tiled) {   // the row-tiled InfoNCE kernel accumulates: its outputs are zeroed by this launch
      t.z[0].p = mi_raw; t.z[0].chunk = 2 * NE_MI; t.z[0].stride = 0; t.z[0].rep = 1;
      if (want_grad) { t.z[1].p = dtout; t.z[1].chunk = (long)BD; t.z[1].stride = 2 * (long)BD; t.z[1].rep = NE_MI; }
    }
    MX(copy_rows(stream, t, (long)BD));
  }
  if (sep) {
    const int dims[5] = {EMB, HID, HID, HID, EMB};
    MX(mlp_stack_forward(10, B, B, tower0, tower_stride, 4, tower_l, dims, tin, ta, tout));
    mi_fused_bwd_done = false;
    if (nce_tiled) {   // InfoNCE: one workgroup per (estimator, 32 score rows) instead of one per estimator
      mi_fused_bwd_done = want_grad;
      return mi_sep_nce_tiled(stream, tout, dtout, mi_raw, mi_raw + NE_MI, gs_mi(stage), NE_MI, B, want_grad ? 1 : 0);
    }
    if (fused_mi) {
      // scores, bound, d/dscores and the gradients of both tower outputs in one launch per stage (estimator_ops.hip)
      mi_fused_bwd_done = want_grad;
      if (has_baseline()) MX(baseline_forward());
      return mi_sep_fused(stream, tout, dtout, mi_raw, mi_raw + NE_MI, gs_mi(stage), NE_MI, B, cfg.bound_type,
                          stage == 1 ? 0x1fu : 0x07u, want_grad ? 1 : 0, has_baseline() ? lbv : nullptr,
                          has_baseline() ? dlbv : nullptr, 2L * B);
    }
    GemmDesc g;   // scores_e = h(y) g(x)^T   (VMI.py:55-57)
    g.A = tout + BD; g.sa_m = EMB; g.sa_k = 1; g.sa_b = 2 * (long)BD;
    g.B = tout; g.sb_k = 1; g.sb_n = EMB; g.sb_b = 2 * (long)BD;
    g.C = scores; g.sc_m = B; g.sc_n = 1; g.sc_b = (long)B * B;
    g.M = B; g.N = B; g.K = EMB; g.batch = NE_MI;
    MX(G_(g));
  } else {
    // layer 0 in separable form: W0 [x|y] = W0x x + W0y y   (VMI.py:59-65: scores[i,j] = f(x_i, y_j))
    GemmDesc gp;
    gp.A = tin; gp.sa_m = EMB; gp.sa_k = 1; gp.sa_b = 2 * (long)BD;
    gp.B = CP(tower0 + tower_l[0][0]); gp.sb_k = 1; gp.sb_n = 2 * EMB; gp.sb_b = tower_stride;
    gp.C = cP; gp.sc_m = HID; gp.sc_n = 1; gp.sc_b = (long)B * HID;
    gp.M = B; gp.N = HID; gp.K = EMB; gp.batch = NE_MI;
    MX(G_(gp));
    GemmDesc gq = gp;
    gq.A = tin + BD; gq.B = CP(tower0 + tower_l[0][0]) + EMB; gq.C = cQ;
    gq.bias_n = CP(tower0 + tower_l[0][1]); gq.bias_n_b = tower_stride;
    MX(G_(gq));
    // pair expansion + both hidden layers + score head in ONE launch with the activation tile in LDS (concat_fused.hip); the
    // unfused chain (fp32 mode, no bf16 image yet, MIMRL_NO_FUSED_CONCAT=1) is pair_expand + three GEMMs
    if (fused_concat && bf16 && img_valid && crit_img && concat_fwd_fused_supported(B, HID)) {
      ConcatFwdArgs fa;
      fa.P = cP; fa.Q = cQ;
      fa.W1 = crit_img + tower0 + tower_l[1][0]; fa.W2 = crit_img + tower0 + tower_l[2][0];
      fa.b1 = CP(tower0 + tower_l[1][1]); fa.b2 = CP(tower0 + tower_l[2][1]);
      fa.w3 = CP(tower0 + tower_l[3][0]); fa.b3 = CP(tower0 + tower_l[3][1]);
      fa.pstride = tower_stride; fa.scores = scores; fa.E = NE_MI; fa.B = B;
      // what the backward pass will read: the fused one takes ReLU bitmasks (+ bf16 values for stage 1's weight gradients), 32x / 2x
      // fewer bytes than the fp32 activations the GEMM-chain backward needs; an evaluation saves nothing
      concat_compact = want_grad && (prec & MIMRL_PREC_BF16_GEMM_BWD) && imgT_ready && concat_bwd_fused_supported(B, HID);
      const size_t half = (size_t)NE_MI * B * B * (HID / 2);      // floats: the bf16 copy fills the lower half of each fp32-sized buffer
      fa.save = !want_grad ? 0 : !concat_compact ? 1 : stage == 1 ? 2 : 3;
      fa.a0 = ca[0]; fa.a1 = ca[1]; fa.a2 = ca[2];
      fa.a0b = reinterpret_cast<__bf16*>(ca[0]); fa.a1b = reinterpret_cast<__bf16*>(ca[1]);
      fa.m1 = reinterpret_cast<uint32_t*>(ca[1] + half); fa.m2 = reinterpret_cast<uint32_t*>(ca[2] + half);   // (written in stage 2 only: in stage 1 a2 stays fp32 and fills its buffer)
      MX(concat_fwd_fused(stream, fa));
    } else {
      concat_compact = false;
      MX(pair_expand_fwd(stream, cP, cQ, ca[0], NE_MI, B, HID));
      const int dims[4] = {HID, HID, HID, 1};
      MX(mlp_stack_forward(NE_MI, B * B, B * B, tower0, tower_stride, 3, &tower_l[1], dims, ca[0], &ca[1], scores));
    }
  }
  if (has_baseline()) MX(baseline_forward());
  return mi_bound_fwd_bwd(stream, scores, want_grad ? dscores : nullptr, mi_raw, mi_raw + NE_MI, gs_mi(stage), NE_MI, B,
                          cfg.bound_type, stage == 1 ? 0x1fu : 0x07u, has_baseline() ? lbv : nullptr,
                          has_baseline() && want_grad ? dlbv : nullptr, 2L * B);
}

int mimrl_handle::cmi_forward(int stage, bool want_grad) {
  const int B = cfg.batch, n = nprod(), m = m_anchor(), k = cfg.k_neighbor;
  const size_t BD = (size_t)B * EMB;
  const float* cur[5] = {bufs.feats, bufs.feats + BD, bufs.feats + 2 * BD, bufs.feats + 3 * BD, bufs.labels};
  const float* bank[5] = {bufs.bank_f, bufs.bank_t, bufs.bank_a, bufs.bank_v, bufs.bank_c};
  CmiAssembleArgs ca_;
  ca_.anchors = bufs.anchors + (size_t)(stage - 1) * NE_CMI * m;
  ca_.idx_x = stage == 2 ? knn_idx2 : knn_idx; ca_.out = cmi_in; ca_.n = n; ca_.m = m; ca_.k = k; ca_.ncall = NE_CMI;
  for (int e = 0; e < NE_CMI; ++e)
    for (int o = 0; o < 3; ++o) {
      const int f = kCmiWire[e][o];
      ca_.op[e][o] = CmiOperand{cur[f], bank[f], f == FT_C ? 1 : 0};
    }
  MX(cmi_assemble(stream, ca_));
  const int cdims[5] = {3 * EMB, HID, HID, HID, 2};
  MX(mlp_stack_forward(NE_CMI, 2 * n, 2 * n, cmi0, cmi_stride, 4, cmi_l, cdims, cmi_in, cc, logits));
  return cmi_loss_fwd_bwd(stream, logits, want_grad ? dlogits : nullptr, bce_raw, cmi_raw, g_bce(stage), g_cmi(stage),
                          NE_CMI, n, cfg.cmi_hardtanh);
}

// log a(y) for every estimator's y operand (VMI.py:101-108)
__global__ void gauss_baseline_kernel(const float* __restrict__ y, float* __restrict__ lb, const float* __restrict__ dlb,
                                      float* __restrict__ dy, int B, int D) {
  // forward (dlb == null): lb[e][i] = sum_d log N(y_id; 0, 1);  backward: dy[e][i][:] = dlb[e][i] * (-y[e][i][:])
  const int e = blockIdx.y, i = blockIdx.x;
  const float* yr = y + ((long)(2 * e + 1) * B + i) * D;
  if (!dlb) {
    __shared__ float red[16];
    float s = 0.f;
    for (int d = threadIdx.x; d < D; d += blockDim.x) s += -0.5f * yr[d] * yr[d] - 0.91893853320467274178f;
    s = block_sum(s, red);
    if (threadIdx.x == 0) lb[(long)e * 2 * B + i] = s;
  } else {
    const float g = dlb[(long)e * 2 * B + i];
    for (int d = threadIdx.x; d < D; d += blockDim.x) dy[((long)e * 2 * B + i) * D + d] = -g * yr[d];
  }
}

int mimrl_handle::baseline_forward() {
  const int B = cfg.batch;
  if (cfg.baseline_type == MIMRL_BASELINE_GAUSSAIN) {
    hipLaunchKernelGGL(gauss_baseline_kernel, dim3(B, NE_MI), dim3(128), 0, stream, tin, lbv, (const float*)nullptr,
                       (float*)nullptr, B, EMB);
    LAUNCH_CHECK();
    return MIMRL_OK;
  }
  const int dims[5] = {EMB, HID, HID, HID, 1};
  return mlp_stack_forward(NE_MI, B, 2 * B, bl0, bl_stride, 4, bl_l, dims, tin + (size_t)B * EMB, bact, lbv);
}

int mimrl_handle::baseline_backward(int stage) {
  const int B = cfg.batch;
  if (cfg.baseline_type == MIMRL_BASELINE_GAUSSAIN) {
    if (stage != 2) return MIMRL_OK;     // no parameters; in stage 1 the features are constants
    hipLaunchKernelGGL(gauss_baseline_kernel, dim3(B, NE_MI), dim3(128), 0, stream, tin, (float*)nullptr, dlbv, bdin, B, EMB);
    LAUNCH_CHECK();
    return MIMRL_OK;
  }
  const int dims[5] = {EMB, HID, HID, HID, 1};
  return mlp_stack_backward(NE_MI, B, 2 * B, bl0, bl_stride, 4, bl_l, dims, tin + (size_t)B * EMB, bact, dlbv, bdz,
                            stage == 2 ? bdin : nullptr, stage == 1);
}

int mimrl_handle::mi_backward(int stage) {
  if (has_baseline()) MX(baseline_backward(stage));
  const int B = cfg.batch;
  const size_t BD = (size_t)B * EMB;
  const bool sep = cfg.critic_type == MIMRL_CRITIC_SEPARATE;
  const bool wgrad = stage == 1;
  float* din_mi = stage == 2 ? dtin : nullptr;
  if (sep && mi_fused_bwd_done) {   // dtout already written by mi_sep_fused
    const int dims[5] = {EMB, HID, HID, HID, EMB};
    return mlp_stack_backward(10, B, B, tower0, tower_stride, 4, tower_l, dims, tin, ta, dtout, dta, din_mi, wgrad);
  }
  if (sep) {
    GemmDesc gh;   // d h = dS g
    gh.A = dscores; gh.sa_m = B; gh.sa_k = 1; gh.sa_b = (long)B * B;
    gh.B = tout; gh.sb_k = EMB; gh.sb_n = 1; gh.sb_b = 2 * (long)BD;
    gh.C = dtout + BD; gh.sc_m = EMB; gh.sc_n = 1; gh.sc_b = 2 * (long)BD;
    gh.M = B; gh.N = EMB; gh.K = B; gh.batch = NE_MI;
    MX(G_(gh));
    GemmDesc gg = gh;   // d g = dS^T h
    gg.sa_m = 1; gg.sa_k = B; gg.B = tout + BD; gg.C = dtout;
    MX(G_(gg));
    const int dims[5] = {EMB, HID, HID, HID, EMB};
    return mlp_stack_backward(10, B, B, tower0, tower_stride, 4, tower_l, dims, tin, ta, dtout, dta, din_mi, wgrad);
  }
  const int dims[4] = {HID, HID, HID, 1};
  if (concat_compact) {
    // the data-gradient chain of the tail (score head -> both hidden layers -> masked gradient of the pair-expanded layer) as ONE launch
    // with the gradient tile in LDS (concat_fused.hip); dZ2 / dZ1 leave it as bf16 for the two weight-gradient GEMMs of stage 1
    ConcatBwdArgs fa;
    std::memset(&fa, 0, sizeof fa);
    const size_t half = (size_t)NE_MI * B * B * (HID / 2);
    fa.ds = dscores; fa.compact = 1;
    fa.m1 = reinterpret_cast<const uint32_t*>(ca[1] + half); fa.m2 = reinterpret_cast<const uint32_t*>(ca[2] + half);
    fa.a2 = ca[2]; fa.P = cP; fa.Q = cQ;
    fa.w3 = CP(tower0 + tower_l[3][0]);
    fa.W2T = crit_imgT + tower0 + tower_l[2][0]; fa.W1T = crit_imgT + tower0 + tower_l[1][0];
    fa.pstride = tower_stride; fa.dz0 = dca[2]; fa.dP = dP; fa.E = NE_MI; fa.B = B;
    __bf16* dz2 = reinterpret_cast<__bf16*>(dca[0]); __bf16* dz1 = reinterpret_cast<__bf16*>(dca[1]);
    if (wgrad) {
      fa.dz2 = dz2; fa.dz1 = dz1;
      fa.db1 = CG(tower0 + tower_l[1][1]); fa.db2 = CG(tower0 + tower_l[2][1]);
      fa.dw3 = CG(tower0 + tower_l[3][0]); fa.db3 = CG(tower0 + tower_l[3][1]);
    }
    if (B > 128) HIPX(hipMemsetAsync(dP, 0, sizeof(float) * NE_MI * B * HID, stream));   // two or more tiles add into each dP row
    MX(concat_bwd_fused(stream, fa));
    const bool side_wg = wgrad && multi_stream && wg_helper >= 0;
    if (wgrad) {   // dW2 = dZ2^T a1, dW1 = dZ1^T a0: K = B*B rows, split-K with atomics, beside pair_reduce_q on the helper stream
      if (side_wg) MX(fork(wg_helper, wg_helper));
      for (int l = 2; l >= 1; --l) {
        GemmDesc g;
        g.A = reinterpret_cast<const float*>(l == 2 ? dz2 : dz1); g.a_bf16 = 1; g.sa_m = 1; g.sa_k = HID; g.sa_b = (long)B * B * HID;
        g.B = l == 2 ? ca[1] : ca[0]; g.b_bf16 = 1; g.sb_k = HID; g.sb_n = 1; g.sb_b = (long)B * B * HID;   // (the bf16 copies)
        g.C = CG(tower0 + tower_l[l][0]); g.sc_m = HID; g.sc_n = 1; g.sc_b = tower_stride;
        g.M = HID; g.N = HID; g.K = B * B; g.batch = NE_MI; g.atomic = 1;
        MX(G_on(side_wg ? S(wg_helper) : stream, g));
      }
    }
    MX(pair_reduce_q(stream, dca[2], dQ, NE_MI, B, HID));
    if (side_wg) MX(join(wg_helper, wg_helper));
  } else {
    // (the gradient of the pair-expanded first layer gets its own buffer: the fused chain keeps every dZ alive for the
    // weight-gradient GEMMs)
    MX(mlp_stack_backward(NE_MI, B * B, B * B, tower0, tower_stride, 3, &tower_l[1], dims, ca[0], &ca[1], dscores, dca,
                          dca[2], wgrad));
    MX(pair_expand_bwd(stream, ca[0], dca[2], dP, dQ, NE_MI, B, HID));
  }
  if (wgrad) {
    GemmDesc g;   // dW0[:, :128] = dP^T x ; dW0[:, 128:] = dQ^T y ; db0 = colsum(dQ)
    g.A = dP; g.sa_m = 1; g.sa_k = HID; g.sa_b = (long)B * HID;
    g.B = tin; g.sb_k = EMB; g.sb_n = 1; g.sb_b = 2 * (long)BD;
    g.C = CG(tower0 + tower_l[0][0]); g.sc_m = 2 * EMB; g.sc_n = 1; g.sc_b = tower_stride;
    g.M = HID; g.N = EMB; g.K = B; g.batch = NE_MI;
    MX(G_(g));
    GemmDesc g2 = g;
    g2.A = dQ; g2.B = tin + BD; g2.C = CG(tower0 + tower_l[0][0]) + EMB;
    MX(G_(g2));
    return colsum(stream, dQ, B, HID, HID, CG(tower0 + tower_l[0][1]), NE_MI, (long)B * HID, tower_stride);
  }
  GemmDesc g;   // dx = dP W0x ; dy = dQ W0y
  g.A = dP; g.sa_m = HID; g.sa_k = 1; g.sa_b = (long)B * HID;
  g.B = CP(tower0 + tower_l[0][0]); g.sb_k = 2 * EMB; g.sb_n = 1; g.sb_b = tower_stride;
  g.C = dtin; g.sc_m = EMB; g.sc_n = 1; g.sc_b = 2 * (long)BD;
  g.M = B; g.N = EMB; g.K = HID; g.batch = NE_MI;
  MX(G_(g));
  GemmDesc g2 = g;
  g2.A = dQ; g2.B = CP(tower0 + tower_l[0][0]) + EMB; g2.C = dtin + BD;
  return G_(g2);
}

int mimrl_handle::cmi_backward(int stage) {
  const int n = nprod();
  const int cdims[5] = {3 * EMB, HID, HID, HID, 2};
  if (stage == 1)
    return mlp_stack_backward(NE_CMI, 2 * n, 2 * n, cmi0, cmi_stride, 4, cmi_l, cdims, cmi_in, cc, dlogits, dcc, nullptr, true);
  // stage 2: only the n joint rows carry gradient to the model (the product rows come from the detached banks)
  return mlp_stack_backward(NE_CMI, n, 2 * n, cmi0, cmi_stride, 4, cmi_l, cdims, cmi_in, cc, dlogits, dcc, dcin, false);
}

// stage 2: route input gradients back to F_F, T_F, A_F, V_F (deterministic gather-sum)
int mimrl_handle::route_feature_grads() {
  const int B = cfg.batch, n = nprod();
  const size_t BD = (size_t)B * EMB;
  GatherSum4 g4;
  for (int f = 0; f < 4; ++f) {
    GatherSum& gs = g4.g[f];
    gs.n = 0;
    for (int e = 0; e < NE_MI; ++e)
      for (int sd = 0; sd < 2; ++sd)
        if (kMiWire[e][sd] == f) {
          gs.src[gs.n] = dtin + (e * 2 + sd) * BD; gs.ld[gs.n] = EMB; gs.off[gs.n] = 0; gs.rows[gs.n] = B; ++gs.n;
        }
    if (has_baseline())   // the baseline is a function of the y operand (VMI.py:101-108)
      for (int e = 0; e < NE_MI; ++e)
        if (kMiWire[e][1] == f) {
          gs.src[gs.n] = bdin + (size_t)e * 2 * B * EMB; gs.ld[gs.n] = EMB; gs.off[gs.n] = 0; gs.rows[gs.n] = B; ++gs.n;
        }
    for (int e = 0; e < NE_CMI; ++e)
      for (int o = 0; o < 3; ++o)
        if (kCmiWire[e][o] == f) {
          gs.src[gs.n] = dcin + (size_t)e * 2 * n * 384; gs.ld[gs.n] = 384; gs.off[gs.n] = o * EMB; gs.rows[gs.n] = n; ++gs.n;
        }
    g4.dst[f] = dfeat + f * BD;
  }
  // The F slot's sum is folded into head_bwd (its only consumer); T / A / V are needed only behind the CubeMLP backward: side 0,
  // off the chain (was one launch + a queue hop between the stage-2 estimators and the head: ~20 us)
  static const bool no_head_gather = knob("MIMRL_NO_HEAD_GATHER") != nullptr;   // tuning knob
  head_gather_on = !no_head_gather && multi_stream && side_on(0);
  if (!head_gather_on) return gather_sum4(stream, g4, B, EMB);
  head_gather = g4.g[0];
  MX(fork(0, 0));
  MX(gather_sum4(S(0), g4, B, EMB, 1));
  MX(next_event(&ev_dmean));
  HIPX(hipEventRecord(ev_dmean, S(0)));
  return MIMRL_OK;
}

// all estimator work of one stage, given that knn_launch() already runs on side 4 and the features are ready on `stream`
int mimrl_handle::estimators_all(int stage, bool want_grad, bool backward) {
  Range rg(stage == 1 ? "mimrl.estimators.stage1 (Model.py:305-341)" : "mimrl.estimators.stage2 (Model.py:343-386)");
  const bool bf_fwd = (prec & MIMRL_PREC_BF16_GEMM_FWD) != 0 && !fp32_site(8), bf_bwd = (prec & MIMRL_PREC_BF16_GEMM_BWD) != 0;
  static const bool dbg_skip_imgt = dbg_env("MIMRL_DBG_SKIP_IMGT") != nullptr;   // timing experiments only (stale images: wrong gradients)
  static const bool imgt_first = knob("MIMRL_IMGT_FIRST") != nullptr;         // tuning knob
  bool imgT_pending = false;
  imgT_ready = false;
  if (frag_side_pending) { MX(join(3, 3)); frag_side_pending = false; }
  if (backward && bf_bwd && fused_mlp && crit_imgT && ttab.n > 0) {   // transposed weight images for the fused data-gradient chains,
    if (!(skip_imgT_refresh && stage == 1)) {                         // built beside the forward stacks (combined step: once per step,
      MX(fork(3, 3));                                                 // in stage 2 -- stage 1 of the NEXT step sees the same critics)
      // capture order: the launch itself goes BEHIND the CMI branch's forward kernels (see cmi_branch) -- graph nodes are dispatched in
      // capture order, and as the first child of the stage boundary it held up both forward branches by ~18 us (MIMRL_IMGT_FIRST=1)
      imgT_pending = !imgt_first && multi_stream && side_on(3) && side_on(5);
      if (!imgT_pending && !dbg_skip_imgt) MX(bf16_transposed_images(S(3), bufs.crit_p, crit_imgT, ttab));
    }
    imgT_ready = true;
  }
  static const int dbg_skip = dbg_env("MIMRL_DBG_SKIP_EST") ? atoi(dbg_env("MIMRL_DBG_SKIP_EST")) : 0;   // timing experiments only
  MX(fork(5, 5));
  MX(chain(5, 4));                       // the CMI branch needs the kNN indices
  auto cmi_branch = [&]() -> int {
    if (dbg_skip & 1) return MIMRL_OK;
    StreamGuard g(this, S(5));
    bf16 = bf_fwd;
    MX(cmi_forward(stage, want_grad));
    if (imgT_pending) {   // side 3 already waits for the stage boundary (fork above); only the launch was held back
      imgT_pending = false;
      if (!dbg_skip_imgt) MX(bf16_transposed_images(side[3], bufs.crit_p, crit_imgT, ttab));
    }
    MX(dbg_delay(stream, stage == 1 ? 4 : 14));
    // (no helper side stream for this branch's weight gradients: it runs on side 5, and a fork / join pair hanging off a captured stream
    //  other than the capture's origin sends this HIP runtime's EndCapture into an endless recursion -- tried, core dump)
    if (backward) { bf16 = bf_bwd; if (imgT_ready && !(skip_imgT_refresh && stage == 1)) MX(chain(5, 3)); MX(cmi_backward(stage)); MX(dbg_delay(stream, stage == 1 ? 6 : 16)); }
    return MIMRL_OK;
  };
  auto mi_branch = [&]() -> int {
    bf16 = bf_fwd;
    if (dbg_skip & 2) return MIMRL_OK;
    { Scope sc(this, MIMRL_PH_EST_FWD); MX(mi_forward(stage, want_grad)); }
    MX(dbg_delay(stream, stage == 1 ? 3 : 15));
    if (backward) {
      bf16 = bf_bwd;
      if (imgT_ready && !(skip_imgT_refresh && stage == 1)) MX(join(3, 3));
      Scope sc(this, MIMRL_PH_EST_BWD);
      wg_helper = stage == 1 ? 1 : -1;     // the MI branch is the critical one of stage 1 (tools/critical_path.sh)
      const int r = mi_backward(stage);
      wg_helper = -1;
      MX(r);
      MX(dbg_delay(stream, stage == 1 ? 5 : 17));
    }
    return MIMRL_OK;
  };
  // capture order of the two branches (graph nodes are dispatched in capture order; bit 0: stage 1, bit 1: stage 2 -> MI first)
  static const int mi_first = knob("MIMRL_EST_MI_FIRST") ? atoi(knob("MIMRL_EST_MI_FIRST")) : 0;
  if ((mi_first >> (stage - 1)) & 1) { MX(mi_branch()); MX(cmi_branch()); }
  else { MX(cmi_branch()); MX(mi_branch()); }
  bf16 = bf_fwd;
  if (!multi_stream) return MIMRL_OK;
  return join(5, 5);
}

// =================================================================================================
// stage drivers
// =================================================================================================
int mimrl_handle::enqueue_grads(int stage, bool skip_zero) {
  Range rg(stage == 1 ? "mimrl.stage1.grads (Solver.py:205-210)" : "mimrl.stage2.grads (Solver.py:221-232)");
  const int B = cfg.batch;
  const bool have_banks = bank_rows > 0;
  if (!keep_events) ev_next = 0;
  if (stage == 1) {
    static const bool no_share = knob("MIMRL_NO_SHARED_PREFIX") != nullptr;   // tuning knob: evaluate the prefix twice
    static const bool prefix_split = knob("MIMRL_BEGIN_ON_SIDE") != nullptr;   // tuning knob
    const bool share = prefetch && !no_share;
    // counters + scalar reset: the first consumers are the kNN sampler and the recurrence, both behind the join of side 4 in
    // encoders_forward -- in the shared-prefix step it runs on side 4 beside the input projections instead of in front of them
    const bool begin_on_side = share && have_banks && skip_zero && prefix_split && multi_stream && cfg.encoder == MIMRL_ENCODER_GRU;
    if (begin_on_side) MX(fork(4, 4));
    // shared-prefix step with packed layer-0 operands: the pack launch is the first kernel of the prefix on this stream and nothing in
    // front of the recurrence reads the counters or the scalars -- the bookkeeping rides on it (one launch + one gap less on the chain)
    // (measured neutral, 0.970 vs 0.966 ms: the single-thread kernel hides in the gap between two graph launches -- opt-in)
    // (round 4, with the length scan on side 0: -4 us on average over four alternating runs, cfg3 neutral -- on by default; =0: the separate kernel)
    static const bool want_begin_in_pack = !(knob("MIMRL_BEGIN_IN_PACK") && atoi(knob("MIMRL_BEGIN_IN_PACK")) == 0);   // tuning knob
    begin_in_pack = want_begin_in_pack && share && have_banks && skip_zero && !begin_on_side && l0_packed && cfg.encoder == MIMRL_ENCODER_GRU;
    if (!begin_in_pack) {
      hipLaunchKernelGGL(begin_stage_kernel, dim3(1), dim3(64), 0, begin_on_side ? side[4] : stream, d_ints, have_banks ? d_ints + 2 : nullptr,
                         bufs.scalars, 0, 32);
      LAUNCH_CHECK();
    }
    if (!have_banks) return MIMRL_OK;
    if (!skip_zero) HIPX(hipMemsetAsync(bufs.crit_g, 0, sizeof(float) * layout.floats[MIMRL_GROUP_CRITIC], stream));   // epoch-0 rule: zero loss, no update (Customization.py:97-98, Solver.py:201-203)
    bf16 = (prec & MIMRL_PREC_BF16_GEMM_FWD) != 0;
    static const bool pre_first = knob("MIMRL_PREFETCH_FIRST") != nullptr;   // tuning knob: capture order of the two chains
    auto issue_prefetch = [&](hipEvent_t e) -> int {
      // the stage-2 forward pass of this batch depends on nothing stage 1 changes: one sequential branch on its own
      // stream, into the primary buffers (stage 1 itself works on the alternate set)
      HIPX(hipStreamWaitEvent(pre_stream, e, 0));
      const bool ms = multi_stream;
      const unsigned sm = side_mask;
      int r;
      {
        StreamGuard g(this, pre_stream);
        multi_stream = false; rng_add = 1;     // begin_stage(2) has not run yet: use the dropout key it will produce
        r = model_forward(true, true, 0);
        multi_stream = ms; rng_add = 0; side_mask = sm;
      }
      return r;
    };
    hipEvent_t e_begin = nullptr;
    if (prefetch) {
      MX(next_event(&e_begin));
      HIPX(hipEventRecord(e_begin, stream));
      if (pre_first && !share) MX(issue_prefetch(e_begin));
      if (!share) swap_fwd_set();
    }
    int r1;
    if (share) {
      // (1) prefix, once, into the primary set, on the main stream (text projection on side 0, stage 1's kNN sampler on
      //     side 4).  It has to be the capture's origin stream that forks the sides: a fork / join pair hanging off
      //     another captured stream sends this HIP runtime's EndCapture into an endless recursion.
      side_mask = 0x11u;
      r1 = model_forward(true, true, knn_pre ? 3 : 1, 1);   // knn_pre: BOTH stages' samplers as one set of launches on side 4 (round 4)
      side_mask = ~0u;
      MX(r1);
      if (wtT_prebuilt) {   // combined step: the CubeMLP backward's weight images (main parameters only) on side 0 -- captured BEHIND the
        bool df[MIMRL_MAX_BLOCKS];   // encoders (nodes start in capture order: in front of them it delayed the input projections by 12 us);
        MX(fork(0, 0));              // side 0 is joined at the end of stage 2, which is part of the same capture
        MX(wt_images(S(0), (prec & MIMRL_PREC_BF16_GEMM_BWD) != 0, true, df));
        wtT_built = true;
      }
      // (stage 2's kNN sampler runs beside the prefix too -- the recurrence leaves half the CUs idle; its anchor key is the step counter
      //  begin_stage(2) will set: knn_launch(3))
      hipEvent_t e_prefix = nullptr;
      MX(next_event(&e_prefix));
      HIPX(hipEventRecord(e_prefix, stream));
      // (2) stage 2's tail: one sequential branch behind the prefix on pre_stream, into the primary set
      //     (deferred mode: issued later by mimrl_stage2_forward_tail, under the stage-1 gradient all-reduce)
      if (!defer_tail) HIPX(hipStreamWaitEvent(pre_stream, e_prefix, 0));
      if (!defer_tail) {
        StreamGuard g(this, pre_stream);
        const bool ms = multi_stream;
        multi_stream = false; rng_add = 1;     // begin_stage(2) has not run yet: use the dropout key it will produce
        r1 = model_forward(true, true, 0, 2);
        multi_stream = ms; rng_add = 0;
      }
      MX(r1);
      // (3) stage 1's tail on the main stream, into the alternate set, reading the primary set's prefix
      swap_fwd_set();
      float* own_tx = tx_raw; float* own_h1[2] = {h1[0], h1[1]};
      tx_raw = alt.tx_raw; h1[0] = alt.h1[0]; h1[1] = alt.h1[1];
      {
        const bool ms = multi_stream;
        multi_stream = false;
        r1 = model_forward(true, false, 0, 2);
        multi_stream = ms;
      }
      tx_raw = own_tx; h1[0] = own_h1[0]; h1[1] = own_h1[1];
    } else if (prefetch) {
      // two forward passes now run side by side; with only 4 hardware queues, more branches would just be serialised
      // behind one another (measured: with 5+ concurrent branches the step falls back to the sequential time, and the
      // prefetch chain as a separate graph on its own HIP stream is slower too), so stage 1's own forward pass keeps
      // two sides only
      side_mask = 0x11u;                 // text branch (side 0) + kNN sampler (side 4; knn_pre: stage 2's as well, one set of launches)
      r1 = model_forward(true, false, knn_pre ? 3 : 1);
      side_mask = ~0u;
    } else {
      r1 = model_forward(true, false, 1);
    }
    if (r1 == 0) r1 = estimators_all(1, true, true);
    if (prefetch) swap_fwd_set();
    MX(r1);
    if (prefetch && !share && !pre_first) MX(issue_prefetch(e_begin));
    if (!fuse_boundary) {
      hipLaunchKernelGGL(finalize_stage1_kernel, dim3(1), dim3(64), 0, stream, bufs.scalars, mi_raw, cmi_raw, bce_raw, coef1());
      LAUNCH_CHECK();
    }
    if (prefetch && !(share && defer_tail)) {   // rejoin before the stage ends (a captured graph must not leave a dangling branch)
      hipEvent_t e;
      MX(next_event(&e));
      HIPX(hipEventRecord(e, pre_stream));
      HIPX(hipStreamWaitEvent(stream, e, 0));
    }
    return MIMRL_OK;
  }
  if (fuse_boundary) {
    hipLaunchKernelGGL(stage_boundary_kernel, dim3(1), dim3(256), 0, stream, bufs.scalars, mi_raw, cmi_raw, bce_raw, coef1(), d_ints, d_ints + 1,
                       bufs.pred, bufs.labels, dpred, B);
  } else {
    hipLaunchKernelGGL(begin_stage_kernel, dim3(1), dim3(64), 0, stream, d_ints, d_ints + 1, bufs.scalars, 32, 32);
  }
  LAUNCH_CHECK();
  if (!skip_zero) HIPX(hipMemsetAsync(bufs.main_g, 0, sizeof(float) * layout.floats[MIMRL_GROUP_MAIN], stream));
  bf16 = (prec & MIMRL_PREC_BF16_GEMM_FWD) != 0;
  if (prefetch && have_banks) {   // forward pass (and kNN sampling) already done beside stage 1
    if (!knn_pre) { MX(fork(4, 4)); MX(knn_launch(2, S(4))); MX(dbg_delay(S(4), 18)); }
  } else {
    MX(model_forward(true, true, have_banks ? 2 : 0));
  }
  if (!fuse_boundary) {
    hipLaunchKernelGGL(mae_kernel, dim3(1), dim3(256), 0, stream, bufs.pred, bufs.labels, dpred, bufs.scalars + MIMRL_S2_TASK, B);
    LAUNCH_CHECK();
  }
  if (have_banks) {
    MX(estimators_all(2, true, true));
    MX(route_feature_grads());
  } else {
    head_gather_on = false;
    HIPX(hipMemsetAsync(dfeat, 0, sizeof(float) * 4 * B * EMB, stream));
  }
  // (writes scalars only: beside the backward chain on side 0; model_backward joins every side before the stage ends)
  MX(fork(0, 0));
  hipLaunchKernelGGL(finalize_stage2_kernel, dim3(1), dim3(64), 0, S(0), bufs.scalars, mi_raw, cmi_raw, coef2(),
                     have_banks ? 1 : 0);
  LAUNCH_CHECK();
  bf16 = (prec & MIMRL_PREC_BF16_GEMM_BWD) != 0;
  MX(model_backward());
  bf16 = (prec & MIMRL_PREC_BF16_GEMM_FWD) != 0;
  return MIMRL_OK;
}

int mimrl_handle::enqueue_apply(int stage) {
  Range rg(stage == 1 ? "mimrl.stage1.clip+adam (Solver.py:211-213)" : "mimrl.stage2.clip+adam (Solver.py:233-235)");
  if (stage == 1 && bank_rows <= 0) return MIMRL_OK;
  AdamArgs a;
  if (stage == 1) {
    a.p = bufs.crit_p; a.g = bufs.crit_g; a.m = bufs.crit_m; a.v = bufs.crit_v; a.n = layout.floats[MIMRL_GROUP_CRITIC];
    a.lr = bufs.lr_critic; a.step = d_ints + 2;
    if (img_valid) a.pimg = crit_img;   // keep the straight bf16 image in step with the parameters
  } else {
    a.p = bufs.main_p; a.g = bufs.main_g; a.m = bufs.main_m; a.v = bufs.main_v; a.n = layout.floats[MIMRL_GROUP_MAIN];
    a.lr = bufs.lr_main; a.step = d_ints + 1;
  }
  a.beta1 = cfg.beta1; a.beta2 = cfg.beta2; a.eps = cfg.adam_eps; a.weight_decay = cfg.weight_decay; a.clip = cfg.grad_clip;
  a.gscale = grad_scale;
  if (stage == 2) { w1_img_valid = false; for (bool& v : w2p_valid) v = false; }
  if (stage == 2 && unpack_pending) {
    unpack_pending = false;
    const int G = 3 * 128;
    for (int m = 0; m < 2; ++m)
      for (int d = 0; d < 2; ++d) {
        const int md = m * 2 + d, din = gru[m][0][0].din;
        int q = a.fold.n++;
        a.fold.lo[q] = Gm(gru[m][0][d].w_ih) - bufs.main_g; a.fold.hi[q] = a.fold.lo[q] + (long)G * din;
        a.fold.src[q] = dwih_pack + (long)md * G * KP(); a.fold.d[q] = din; a.fold.ld[q] = KP();
        q = a.fold.n++;
        a.fold.lo[q] = Gm(gru[m][0][d].w_hh) - bufs.main_g; a.fold.hi[q] = a.fold.lo[q] + (long)G * 128;
        a.fold.src[q] = dwhh_pack + (long)md * G * 128; a.fold.d[q] = 128; a.fold.ld[q] = 128;
      }
    a.fold.lo_all = a.fold.lo[0]; a.fold.hi_all = a.fold.hi[0];
    for (int q = 1; q < a.fold.n; ++q) { a.fold.lo_all = std::min(a.fold.lo_all, a.fold.lo[q]); a.fold.hi_all = std::max(a.fold.hi_all, a.fold.hi[q]); }
  }
  Scope sc(this, MIMRL_PH_OPT);
  MX(adam_step(stream, a));
  if (stage == 1 && img_valid && crit_frag && ftab.n > 0) {
    // combined step: beside the stage boundary on side 3 (the stage-2 estimators join it before their first stack)
    static const bool inline_frag = knob("MIMRL_FRAG_INLINE") != nullptr;   // tuning knob
    if (fuse_boundary && side_on(3) && !inline_frag) { MX(fork(3, 3)); MX(bf16_frag_images(side[3], bufs.crit_p, crit_frag, ftab)); frag_side_pending = true; }
    else MX(bf16_frag_images(stream, bufs.crit_p, crit_frag, ftab));
  }
  return dbg_delay(stream, 12);
}

// Stage-2 gradient pass under data parallelism (reference counterpart: nn.DataParallel reduces AFTER backward, Solver.py:33-35; north_star:
// "all-reduce ... overlapped with the other stage's backward"): everything but the layer-0 recurrence gradients is final behind part 0
// (layout.cpp puts those tensors at the tail of the bucket), so [0, late_offset) -- 0.80 M of 1.08 M floats -- is all-reduced on its own
// stream while the layer-0 BPTT and its weight gradients run; the tail follows on the main stream.  Captured like everything else.
int mimrl_handle::enqueue_grads2_reduced(bool skip_zero) {
  const bool ke = keep_events;
  split_part = 1; fold_unpack = false;
  int r = enqueue_grads(2, skip_zero);
  split_part = 0;
  MX(r);
  const long n_main = layout.floats[MIMRL_GROUP_MAIN], early = layout.late_offset;
  hipEvent_t e0, e1;
  MX(next_event(&e0)); MX(next_event(&e1));
  if (early > 0) {
    Range rg("mimrl.stage2.allreduce(main_g[early]) [RCCL, under the layer-0 BPTT]");
    HIPX(hipEventRecord(e0, stream));
    HIPX(hipStreamWaitEvent(comm_s, e0, 0));
    MX(comm_allreduce_sum(comm, bufs.main_g, (size_t)early, comm_s));
    HIPX(hipEventRecord(e1, comm_s));
  }
  keep_events = true;                       // (the events above stay reserved while part 1 draws its own)
  bf16 = (prec & MIMRL_PREC_BF16_GEMM_BWD) != 0;
  r = gru_layer_backward(0);
  if (r == 0) r = join(0, 5);
  bf16 = (prec & MIMRL_PREC_BF16_GEMM_FWD) != 0;
  keep_events = ke;
  MX(r);
  if (n_main > early) {
    Range rg("mimrl.stage2.allreduce(main_g[layer-0 tail]) [RCCL]");
    MX(comm_allreduce_sum(comm, bufs.main_g + early, (size_t)(n_main - early), stream));
  }
  if (early > 0) HIPX(hipStreamWaitEvent(stream, e1, 0));
  return MIMRL_OK;
}

// kind 0: grads + apply (single-GPU step); kind 1: grads only; kind 2: apply only (never captured: one kernel)
int mimrl_handle::run(int stage, int kind) {
  if (!bound) return set_error(MIMRL_ERR_STATE, "mimrl_bind must be called before any step");
  if (stage != 1 && stage != 2) return set_error(MIMRL_ERR_ARG, "stage must be 1 or 2");
  MX(ensure_images());
  if (stage == 1 && kind != 1) imgT_valid = false;     // a critic update outside the combined step: its periodic image state is gone
  if (kind == 2) { grads_clean[stage] = true; return enqueue_apply(stage); }
  // kind 0 (fused step): the previous apply left the bucket zeroed, so no memset node; kind 1 (grads only, e.g. before
  // an all-reduce): always zero first -- the caller may call it repeatedly
  if (kind == 4 && (stage != 2 || cfg.encoder != MIMRL_ENCODER_GRU))
    return set_error(MIMRL_ERR_ARG, "mimrl_stage_grads_part: only stage 2 of the GRU encoders splits");
  // part 1 (the layer-0 BPTT) consumes dh0 and the saved gates part 0 left behind: out of order it would add gradients of a stale
  // batch into main_g without a word.  Any other staged call in between invalidates the hand-over.
  if (kind == 4 && !part0_done)
    return set_error(MIMRL_ERR_STATE, "mimrl_stage_grads_part: part 1 must directly follow part 0 of the same stage-2 pass");
  part0_done = false;
  if (prefetch && bank_rows > 0 && kind != 4) {
    if (stage == 1) { fwd2_pending = true; tail2_needed = defer_tail; }
    else if (!fwd2_pending)
      return set_error(MIMRL_ERR_STATE, "stage-2 prefetch mode: stage 2 must follow a stage-1 call on the same batch");
    else if (tail2_needed)
      return set_error(MIMRL_ERR_STATE, "deferred-tail mode: call mimrl_stage2_forward_tail between stage 1 and stage 2");
    else fwd2_pending = false;
  }
  const bool skip_zero = kind == 0 && grads_clean[stage];
  auto body = [&]() -> int {
    if (kind == 4) {                 // second half of a split stage-2 gradient pass: the layer-0 GRU backward
      if (!keep_events) ev_next = 0;
      bf16 = (prec & MIMRL_PREC_BF16_GEMM_BWD) != 0;
      int r = gru_layer_backward(0);
      if (r == 0) r = join(0, 5);
      bf16 = (prec & MIMRL_PREC_BF16_GEMM_FWD) != 0;
      return r;
    }
    if (kind == 0 && comm && !(stage == 1 && bank_rows <= 0)) {   // data parallel: gradient pass, all-reduce, update -- one enqueue
      if (stage == 2 && comm_split && cfg.encoder == MIMRL_ENCODER_GRU) MX(enqueue_grads2_reduced(skip_zero));
      else { MX(enqueue_grads(stage, skip_zero)); MX(reduce_bucket(stage)); }
      return enqueue_apply(stage);
    }
    split_part = kind == 3 ? 1 : 0;
    fold_unpack = kind == 0 && stage == 2 && fold_unpack_on;   // the update follows in the same enqueue: it scatters the packed layer-0 pieces
    const int r = enqueue_grads(stage, skip_zero);
    split_part = 0; fold_unpack = false;
    if (r != 0) unpack_pending = false;
    MX(r);
    if (kind == 0) MX(enqueue_apply(stage));
    return MIMRL_OK;
  };
  if (kind == 0 && !skip_zero) {   // a graph captured now would bake the memset in; run this one eagerly instead
    grads_clean[stage] = true;
    return body();
  }
  if (kind == 1 || kind == 3) grads_clean[stage] = false;
  if (!cfg.use_graph || prof_on) { const int r = body(); part0_done = r == 0 && kind == 3; return r; }
  const int gk = kind >= 3 ? kind - 1 : kind;     // graph cache slot: 0 step, 1 grads, 2 / 3 the halves of a split stage-2 pass
  hipGraphExec_t& ex = GS().graph[stage][gk];
  if (ex && GS().rows[stage][gk] != bank_rows) retire(ex);   // bank size is baked into the kernel arguments
  if (!ex) {
    hipGraph_t g = nullptr;
    if (!cap_stream) HIPX(hipStreamCreateWithFlags(&cap_stream, hipStreamNonBlocking));
    HIPX(hipStreamBeginCapture(cap_stream, hipStreamCaptureModeThreadLocal));
    stream = cap_stream;
    const int r = body();
    stream = user_stream;
    const hipError_t ce = hipStreamEndCapture(cap_stream, &g);
    if (r != 0) { if (g) (void)hipGraphDestroy(g); return r; }
    if (ce != hipSuccess) return set_error(MIMRL_ERR_HIP, "hipStreamEndCapture: %s", hipGetErrorString(ce));
    const hipError_t ie = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    if (ie != hipSuccess) { ex = nullptr; return set_error(MIMRL_ERR_HIP, "hipGraphInstantiate: %s", hipGetErrorString(ie)); }
    GS().rows[stage][gk] = bank_rows;
  }
  HIPX(hipGraphLaunch(ex, stream));
  part0_done = kind == 3;
  return MIMRL_OK;
}

// Deferred-tail mode (data parallel): the stage-2 forward tail of the bound batch -- LN+ReLU+dropout, CubeMLP, head, with the
// activations saved for the backward pass -- as its own launch on the caller's stream.  The caller starts the all-reduce of the
// stage-1 (critic) gradients first; this work needs neither those gradients nor the critic update, so the collective hides under it.
int mimrl_handle::run_fwd2_tail() {
  if (!bound) return set_error(MIMRL_ERR_STATE, "mimrl_bind must be called before any step");
  if (!(prefetch && defer_tail && bank_rows > 0)) return MIMRL_OK;          // nothing deferred in the other modes
  if (!tail2_needed) return set_error(MIMRL_ERR_STATE, "mimrl_stage2_forward_tail: no stage-1 call is pending");
  tail2_needed = false;
  auto body = [&]() -> int {
    if (!keep_events) ev_next = 0;
    const bool ms = multi_stream;
    bf16 = (prec & MIMRL_PREC_BF16_GEMM_FWD) != 0;
    multi_stream = false; rng_add = 1;       // begin_stage(2) has not run yet: use the dropout key it will produce
    const int r = model_forward(true, true, 0, 2);
    multi_stream = ms; rng_add = 0;
    return r;
  };
  if (!cfg.use_graph || prof_on) return body();
  if (GS().tail && GS().tail_rows != bank_rows) retire(GS().tail);
  if (!GS().tail) {
    hipGraph_t g = nullptr;
    if (!cap_stream) HIPX(hipStreamCreateWithFlags(&cap_stream, hipStreamNonBlocking));
    HIPX(hipStreamBeginCapture(cap_stream, hipStreamCaptureModeThreadLocal));
    stream = cap_stream;
    const int r = body();
    stream = user_stream;
    const hipError_t ce = hipStreamEndCapture(cap_stream, &g);
    if (r != 0) { if (g) (void)hipGraphDestroy(g); return r; }
    if (ce != hipSuccess) return set_error(MIMRL_ERR_HIP, "hipStreamEndCapture: %s", hipGetErrorString(ce));
    const hipError_t ie = hipGraphInstantiate(&GS().tail, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    if (ie != hipSuccess) { GS().tail = nullptr; return set_error(MIMRL_ERR_HIP, "hipGraphInstantiate: %s", hipGetErrorString(ie)); }
    GS().tail_rows = bank_rows;
  }
  HIPX(hipGraphLaunch(GS().tail, stream));
  return MIMRL_OK;
}

// Captured-graph post-processing (diagnostics / tuning knobs, off by default):
//   MIMRL_GRAPH_DOT=<file>   dump the captured two-stage step (hipGraphDebugDotPrint: kernel names, edges)
//   MIMRL_GRAPH_REORDER=1    re-insert every node's outgoing edges so that the child with the LONGEST path to a sink comes first.  This
//                            HIP runtime maps graph nodes to hardware queues by a depth-first walk in which a node's first edge keeps
//                            the parent's queue and every further edge moves to the next one (tools/hw/graph_order.hip): with the chain's
//                            continuation first, the chain of dependent launches stays on one in-order queue.
__global__ void graph_pad_kernel() {}
static int graph_postprocess(hipGraph_t g) {
  static const char* dot = knob("MIMRL_GRAPH_DOT");
  static const bool reorder = knob("MIMRL_GRAPH_REORDER") != nullptr;
  if (reorder) {
    size_t nn = 0, ne = 0;
    HIPX(hipGraphGetNodes(g, nullptr, &nn));
    std::vector<hipGraphNode_t> nodes(nn);
    HIPX(hipGraphGetNodes(g, nodes.data(), &nn));
    HIPX(hipGraphGetEdges(g, nullptr, nullptr, &ne));
    std::vector<hipGraphNode_t> from(ne), to(ne);
    HIPX(hipGraphGetEdges(g, from.data(), to.data(), &ne));
    std::vector<std::vector<int>> out(nn);
    {
      std::vector<std::pair<hipGraphNode_t, int>> ix(nn);
      for (size_t i = 0; i < nn; ++i) ix[i] = {nodes[i], (int)i};
      std::sort(ix.begin(), ix.end());
      auto idx = [&](hipGraphNode_t n) { return std::lower_bound(ix.begin(), ix.end(), std::make_pair(n, -1))->second; };
      for (size_t e = 0; e < ne; ++e) out[idx(from[e])].push_back(idx(to[e]));
    }
    std::vector<int> h(nn, -1);
    std::function<int(int)> height = [&](int v) -> int { if (h[v] >= 0) return h[v]; int m = 0; for (int c : out[v]) m = std::max(m, 1 + height(c)); return h[v] = m; };
    for (size_t v = 0; v < nn; ++v) height((int)v);
    static const int mode = atoi(knob("MIMRL_GRAPH_REORDER"));   // 1: by height; 2: the child captured on the parent's stream first
    const auto& ns = capture_streams();
    auto stream_of = [&](int v) -> hipStream_t { auto it = ns.find(nodes[v]); return it == ns.end() ? (hipStream_t)-1 : it->second; };
    int changed = 0;
    for (size_t v = 0; v < nn; ++v) {
      if (out[v].size() < 2) continue;
      std::vector<int> o = out[v];
      if (mode == 4) {   // MIMRL_GRAPH_PERM: digit i = which child of the i-th fork node comes first (0 = as captured)
        static const char* perm = knob("MIMRL_GRAPH_PERM");
        static int fork_no = 0;
        int k = perm && fork_no < (int)strlen(perm) ? (perm[fork_no] >= 'a' ? perm[fork_no] - 'a' + 10 : perm[fork_no] - '0') : 0;
        ++fork_no;
        if (k > 0 && k < (int)o.size()) { const int c = o[k]; o.erase(o.begin() + k); o.insert(o.begin(), c); }   // child k first, the others keep their order
        if (knob("MIMRL_GRAPH_VERBOSE")) fprintf(stderr, "[graph] fork %d: node %zu, %zu children\n", fork_no - 1, v, o.size());
      } else if (mode >= 2) {
        const hipStream_t ps = stream_of((int)v);
        if (ps == (hipStream_t)-1) continue;
        std::stable_sort(o.begin(), o.end(), [&](int a, int b) { return (stream_of(a) == ps) > (stream_of(b) == ps); });
      } else {
      std::stable_sort(o.begin(), o.end(), [&](int a, int b) { return h[a] > h[b]; });
      }
      // mode 3: as 2, and the side children of successive forks are spread over the other queues: k empty nodes in front of them push
      // them from queue s + 1 to s + 1 + k (k cycles 0, 1, 2 over the forks; MIMRL_GRAPH_PAD=<list of k per fork> overrides)
      int pads = 0;
      if (mode == 3 && stream_of(o[0]) == stream_of((int)v)) {
        static const char* padlist = knob("MIMRL_GRAPH_PAD");
        static int fork_no = 0;
        pads = padlist && fork_no < (int)strlen(padlist) ? padlist[fork_no] - '0' : fork_no % 3;
        ++fork_no;
      }
      if (o == out[v] && pads == 0) continue;
      std::vector<hipGraphNode_t> f(o.size(), nodes[v]), t;
      for (int c : out[v]) t.push_back(nodes[c]);
      HIPX(hipGraphRemoveDependencies(g, f.data(), t.data(), t.size()));
      HIPX(hipGraphAddDependencies(g, &nodes[v], &nodes[o[0]], 1));
      for (int k = 0; k < pads; ++k) {   // (a one-thread kernel: an EMPTY node in that place cost 200-400 us per step)
        hipGraphNode_t pn;
        hipKernelNodeParams kp = {};
        kp.func = reinterpret_cast<void*>(graph_pad_kernel); kp.gridDim = dim3(1); kp.blockDim = dim3(1); kp.sharedMemBytes = 0;
        kp.kernelParams = nullptr; kp.extra = nullptr;
        HIPX(hipGraphAddKernelNode(&pn, g, &nodes[v], 1, &kp));
      }
      t.clear();
      for (size_t c = 1; c < o.size(); ++c) t.push_back(nodes[o[c]]);
      if (!t.empty()) HIPX(hipGraphAddDependencies(g, f.data(), t.data(), t.size()));
      ++changed;
    }
    if (knob("MIMRL_GRAPH_VERBOSE")) fprintf(stderr, "[graph] %zu nodes, %zu edges, %d fork nodes re-ordered\n", nn, ne, changed);
  }
  if (dot) HIPX(hipGraphDebugDotPrint(g, dot, hipGraphDebugDotFlagsVerbose));
  return MIMRL_OK;
}

// Solver.step(): stage 1 then stage 2 on the bound batch.  In overlap mode with graphs the two stages are ONE captured
// graph (one launch, no idle device between the stage-1 Adam and the stage-2 estimators); otherwise two run() calls.
int mimrl_handle::run_step() {
  Range rg("mimrl.two_stage_step (Solver.step)");
  if (!bound) return set_error(MIMRL_ERR_STATE, "mimrl_bind must be called before any step");
  static const bool no_step_graph = knob("MIMRL_NO_STEP_GRAPH") != nullptr;   // tuning knob
  const bool combined = cfg.use_graph && !prof_on && prefetch && !defer_tail && bank_rows > 0 && grads_clean[1] && grads_clean[2] && !no_step_graph;
  if (!combined) { MX(run(1, 0)); if (defer_tail) MX(run_fwd2_tail()); return run(2, 0); }
  MX(ensure_images());
  const bool bf_bwd = (prec & MIMRL_PREC_BF16_GEMM_BWD) != 0;
  const bool use_imgT = bf_bwd && fused_mlp && crit_imgT && ttab.n > 0;
  if (use_imgT && !imgT_valid) {   // the captured stage 1 relies on the images left by the previous step's stage 2
    MX(bf16_transposed_images(user_stream, bufs.crit_p, crit_imgT, ttab));
    imgT_valid = true;
  }
  hipGraphExec_t& ex = GS().graph[0][0];
  if (ex && GS().rows[0][0] != bank_rows) retire(ex);   // bank size is baked into the kernel arguments
  if (!ex) {
    static const bool no_boundary = knob("MIMRL_NO_FUSED_BOUNDARY") != nullptr;   // tuning knob: the round-1 stage boundary
    hipGraph_t g = nullptr;
    if (!cap_stream) HIPX(hipStreamCreateWithFlags(&cap_stream, hipStreamNonBlocking));
    HIPX(hipStreamBeginCapture(cap_stream, hipStreamCaptureModeThreadLocal));
    capture_track(knob("MIMRL_GRAPH_REORDER") != nullptr);
    stream = cap_stream;
    fuse_boundary = !no_boundary; skip_imgT_refresh = use_imgT && !no_boundary; wtT_prebuilt = bf_bwd && fused_cube_bwd && !no_boundary;
    wtT_built = false;
    int r = enqueue_grads(1, true);
    if (r == 0) r = reduce_bucket(1);
    if (r == 0) r = enqueue_apply(1);
    keep_events = true; fold_unpack = fold_unpack_on && !comm;   // (data parallel: the packed layer-0 pieces must be IN the bucket before it is reduced)
    if (r == 0) {
      if (comm && comm_split && cfg.encoder == MIMRL_ENCODER_GRU) r = enqueue_grads2_reduced(true);
      else { r = enqueue_grads(2, true); if (r == 0) r = reduce_bucket(2); }
    }
    keep_events = false; fold_unpack = false;
    if (r == 0) r = enqueue_apply(2);
    unpack_pending = false;
    fuse_boundary = false; skip_imgT_refresh = false; wtT_prebuilt = false;
    stream = user_stream;
    const hipError_t ce = hipStreamEndCapture(cap_stream, &g);
    if (r != 0) { if (g) (void)hipGraphDestroy(g); return r; }
    if (ce != hipSuccess) return set_error(MIMRL_ERR_HIP, "hipStreamEndCapture: %s", hipGetErrorString(ce));
    { const int pr = graph_postprocess(g); capture_track(false); if (pr != 0) { (void)hipGraphDestroy(g); return pr; } }
    const hipError_t ie = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    if (ie != hipSuccess) { ex = nullptr; return set_error(MIMRL_ERR_HIP, "hipGraphInstantiate: %s", hipGetErrorString(ie)); }
    GS().rows[0][0] = bank_rows;
  }
  HIPX(hipGraphLaunch(ex, stream));
  fwd2_pending = false;
  return MIMRL_OK;
}

// =================================================================================================
// C ABI
// =================================================================================================
extern "C" {

const char* mimrl_last_error(void) { return mimrl::last_error_slot().c_str(); }
#ifdef MIMRL_PHASE_PROBE
// `make PHASE_PROBE=1` only (not part of the ABI): in-kernel phase ticks of the fused CubeMLP forward (tools/cube_phase.py)
int mimrl_dbg_cube_phases(long long* out) { return mimrl::cube_fwd_read_phases(out); }
int mimrl_dbg_cube_bwd_phases(long long* out) { return mimrl::cube_bwd_read_phases(out); }
int mimrl_dbg_kmix_phases(long long* out) { return mimrl::kmix_bwd_read_phases(out); }
int mimrl_dbg_nce_phases(long long* out) { return mimrl::nce_read_phases(out); }
int mimrl_dbg_model_ops_phases(long long* out) { return mimrl::model_ops_read_phases(out); }
int mimrl_dbg_gru_bwd_phases(long long* out) { return mimrl::gru_bwd_read_phases(out); }
#endif
int mimrl_abi_version(void) { return MIMRL_ABI_VERSION; }

int mimrl_device_check(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return set_error(MIMRL_ERR_NODEVICE, "no HIP device visible");
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return set_error(MIMRL_ERR_NODEVICE, "hipGetDevice failed");
  hipDeviceProp_t p;
  if (hipGetDeviceProperties(&p, dev) != hipSuccess) return set_error(MIMRL_ERR_NODEVICE, "hipGetDeviceProperties failed");
  if (std::strncmp(p.gcnArchName, "gfx950", 6) != 0)
    return set_error(MIMRL_ERR_NODEVICE, "device %d is %s; this library is built for gfx950 (MI355X) only", dev, p.gcnArchName);
  return MIMRL_OK;
}

int mimrl_create(const mimrl_cfg* cfg, void* hip_stream, mimrl_handle** out) {
  if (!cfg || !out) return set_error(MIMRL_ERR_ARG, "null argument");
  *out = nullptr;
  MX(validate_cfg(*cfg));
  if (cfg->batch > 1024) return set_error(MIMRL_ERR_ARG, "batch per rank must be <= 1024");
#ifndef MIMRL_DEBUG_KNOBS
  // result-changing debug knobs do not exist in this build; a run that sets one expects something this library will not do
  for (int i = 0; kDebugKnobs[i]; ++i)
    if (getenv(kDebugKnobs[i]))
      return set_error(MIMRL_ERR_ARG, "%s is set, but result-changing debug knobs are compiled out of this build "
                       "(rebuild with `make DEBUG_KNOBS=1` for timing experiments; never for real runs)", kDebugKnobs[i]);
#endif
  MX(mimrl_device_check());
  if (knob_on("MIMRL_KNOBS")) knobs_print(stderr);
  mimrl_handle* h = new (std::nothrow) mimrl_handle();
  if (!h) return set_error(MIMRL_ERR_STATE, "out of host memory");
  h->cfg = *cfg;
  if (h->cfg.beta1 == 0.f) h->cfg.beta1 = 0.9f;
  if (h->cfg.beta2 == 0.f) h->cfg.beta2 = 0.999f;
  if (h->cfg.adam_eps == 0.f) h->cfg.adam_eps = 1e-8f;
  h->stream = h->user_stream = reinterpret_cast<hipStream_t>(hip_stream);
  h->multi_stream = knob("MIMRL_SINGLE_STREAM") == nullptr;
#ifdef MIMRL_DET
  h->multi_stream = false;   // deterministic build: one stream, so the flush behind a launch never meets a half-finished producer (det.h)
  MX(det_init());
#endif
  h->fused_cube = knob("MIMRL_NO_FUSED_CUBE") == nullptr;
  h->fused_mlp = knob("MIMRL_NO_FUSED_MLP") == nullptr;
  h->knn_pre = knob("MIMRL_NO_KNN_PREFETCH") == nullptr;
  h->fold_unpack_on = knob("MIMRL_NO_FOLD_UNPACK") == nullptr;
  h->h16_on = knob("MIMRL_NO_H16") == nullptr;
  h->xin_on = knob("MIMRL_NO_XIN") == nullptr;
  h->fused_cube_bwd = knob("MIMRL_NO_FUSED_CUBE_BWD") == nullptr;
  h->fused_concat = knob("MIMRL_NO_FUSED_CONCAT") == nullptr;
  h->fwd_f16 = knob("MIMRL_FWD_BF16") == nullptr;
  // packed layer-0 operands: one batched projection + two batched weight gradients instead of 2 + 4 launches: the four layer-0
  // weight-gradient GEMMs in a row are what closes the stage behind the BPTT.  (History: before the parked CubeMLP weight gradients
  // became two grouped launches the side streams were the bottleneck and packing lost at cfg2, 1.34 vs 1.32 ms; since then it
  // wins, 1.14 vs 1.18 ms.)  MIMRL_L0_PACK=0 / 1 forces it.
  h->l0_packed = knob("MIMRL_L0_PACK") ? atoi(knob("MIMRL_L0_PACK")) != 0 : true;
  // (inputs packed on side 0, only the layer-0 weight gradients batched: MIMRL_L0_BWD_PACK=1 with MIMRL_L0_PACK=0; 1.15 ms at cfg2)
  // dg[B,T,4H] / h_prev are consumed only by GEMMs that round their operands to bf16 anyway: storing them as bf16 changes no
  // number in this mode and halves what the BPTT writes and the weight-gradient / dh0 products read
  h->dg_bf16 = cfg->encoder == MIMRL_ENCODER_GRU && (cfg->precision & MIMRL_PREC_BF16_GRU_BWD) && (cfg->precision & MIMRL_PREC_BF16_GEMM_BWD) &&
               knob("MIMRL_DG_FP32") == nullptr;
  h->l0_bwd_pack = knob("MIMRL_L0_BWD_PACK") ? atoi(knob("MIMRL_L0_BWD_PACK")) != 0 : false;
  // gx[B,T,3H] -- written once by the input projection, read once by the recurrence -- as fp16 (MIMRL_GX_F16=1; OFF by default).  Round 4
  // built it for cfg3, whose two projections on the chain are bound by 786 MB of fp32 stores each (449 / 302 us), and measured a LOSS:
  // 7.24 against 6.83 ms per step.  The accumulator layout gives a lane one column of 16 rows, so an fp16 store instruction writes two
  // 64-byte half lines (fp32: two full 128-byte lines): half the bytes, but partial-line writes.  Winning needs the tile staged through LDS
  // and written back as whole rows -- a different epilogue.  The path stays (tests/test_gpu_fused_oracle.py holds it to the rounded oracle).
  {
    const bool can = cfg->encoder == MIMRL_ENCODER_GRU && (cfg->precision & MIMRL_PREC_BF16_GRU_FWD) && (cfg->precision & MIMRL_PREC_BF16_GEMM_FWD);
    h->gx_f16 = can && knob("MIMRL_GX_F16") && atoi(knob("MIMRL_GX_F16")) != 0;
  }
  if (cfg->encoder == MIMRL_ENCODER_GRU && ((cfg->precision & MIMRL_PREC_BF16_GRU_FWD) != 0) != ((cfg->precision & MIMRL_PREC_BF16_GRU_BWD) != 0)) {
    mimrl_destroy(h);   // the forward kernel writes the gate slab in the format (bf16 / fp32 records) the BPTT kernel of the SAME mode reads
    return set_error(MIMRL_ERR_ARG, "MIMRL_PREC_BF16_GRU_FWD and MIMRL_PREC_BF16_GRU_BWD must be set together");
  }
  for (int i = 0; i < mimrl_handle::NSIDE; ++i)
    if (hipStreamCreateWithFlags(&h->side[i], hipStreamNonBlocking) != hipSuccess) { mimrl_destroy(h); return set_error(MIMRL_ERR_HIP, "hipStreamCreate failed"); }
  gru_probe_setup();
  h->prec = cfg->precision;
  h->bf16 = (h->prec & MIMRL_PREC_BF16_GEMM_FWD) != 0;
  std::memset(&h->bufs, 0, sizeof h->bufs);
  int r = build_layout(h->cfg, &h->layout);
  if (r == 0) r = h->resolve();
  if (r == 0) r = h->alloc_workspace();
  if (r != 0) { mimrl_destroy(h); return r; }
  *out = h;
  return MIMRL_OK;
}

int mimrl_bind(mimrl_handle* h, const mimrl_buffers* b) {
  if (!h || !b) return set_error(MIMRL_ERR_ARG, "null argument");
  const void* need[] = {b->main_p, b->main_g, b->main_m, b->main_v, b->crit_p, b->crit_g, b->crit_m, b->crit_v,
                        b->text, b->audio, b->video, b->labels, b->lr_main, b->lr_critic, b->pred, b->feats, b->scalars};
  for (const void* p : need)
    if (!p) return set_error(MIMRL_ERR_ARG, "mimrl_bind: a required buffer is null");
  h->bufs = *b;
  h->bound = true;
  h->part0_done = false;
  h->cur_set = 0;
  for (int q = 0; q < 2; ++q) for (int i = 0; i < 4; ++i) h->gsets[q].in[i] = nullptr;
  h->gsets[0].in[0] = b->text; h->gsets[0].in[1] = b->audio; h->gsets[0].in[2] = b->video; h->gsets[0].in[3] = b->labels;
  h->img_valid = false;
  h->imgT_valid = false;
  h->d_ints = b->counters ? b->counters : h->d_ints_own;   // graphs are rebuilt below, so the new address is baked in
  h->drop_graphs();
  return MIMRL_OK;
}

int mimrl_set_inputs(mimrl_handle* h, int set, const float* text, const float* audio, const float* video, const float* labels) {
  if (!h || set < 0 || set > 1 || !text || !audio || !video || !labels) return set_error(MIMRL_ERR_ARG, "mimrl_set_inputs: bad argument");
  if (!h->bound) return set_error(MIMRL_ERR_STATE, "mimrl_bind must be called first");
  const void* p[4] = {text, audio, video, labels};
  mimrl_handle::GraphSet& g = h->gsets[set];
  if (g.in[0] && (g.in[0] != p[0] || g.in[1] != p[1] || g.in[2] != p[2] || g.in[3] != p[3])) {
    HIPX(hipStreamSynchronize(h->user_stream));
    h->drop_graphs(set);                      // this set's graphs were captured with other addresses
  }
  for (int i = 0; i < 4; ++i) g.in[i] = p[i];
  h->cur_set = set;
  h->part0_done = false;                      // another batch: a pending part-0 hand-over is void
  h->bufs.text = text; h->bufs.audio = audio; h->bufs.video = video; h->bufs.labels = labels;
  return MIMRL_OK;
}

int mimrl_set_bank_rows(mimrl_handle* h, int rows) {
  if (!h) return set_error(MIMRL_ERR_ARG, "null handle");
  if (rows < 0 || rows > h->cfg.bank_capacity) return set_error(MIMRL_ERR_ARG, "bank rows %d outside [0,%d]", rows, h->cfg.bank_capacity);
  if (rows > 0) {
    if (!h->bufs.bank_c || !h->bufs.bank_f || !h->bufs.bank_t || !h->bufs.bank_a || !h->bufs.bank_v || !h->bufs.anchors)
      return set_error(MIMRL_ERR_STATE, "banks/anchors must be bound before enabling them");
    if (rows - h->m_anchor() < h->cfg.k_neighbor)
      return set_error(MIMRL_ERR_ARG, "bank of %d rows is too small for %d anchors and k=%d", rows, h->m_anchor(), h->cfg.k_neighbor);
  }
  h->bank_rows = rows;
  return MIMRL_OK;
}

int mimrl_stage1_step(mimrl_handle* h) { return h ? h->run(1, 0) : set_error(MIMRL_ERR_ARG, "null handle"); }
int mimrl_stage2_step(mimrl_handle* h) { return h ? h->run(2, 0) : set_error(MIMRL_ERR_ARG, "null handle"); }
int mimrl_two_stage_step(mimrl_handle* h) { return h ? h->run_step() : set_error(MIMRL_ERR_ARG, "null handle"); }
int mimrl_stage_grads(mimrl_handle* h, int stage) { return h ? h->run(stage, 1) : set_error(MIMRL_ERR_ARG, "null handle"); }
int mimrl_stage_apply(mimrl_handle* h, int stage) { return h ? h->run(stage, 2) : set_error(MIMRL_ERR_ARG, "null handle"); }
int mimrl_stage_grads_part(mimrl_handle* h, int stage, int part) {
  if (!h) return set_error(MIMRL_ERR_ARG, "null handle");
  if (stage != 2 || (part != 0 && part != 1)) return set_error(MIMRL_ERR_ARG, "mimrl_stage_grads_part: stage 2, part 0 or 1");
  if (h->cfg.encoder != MIMRL_ENCODER_GRU) return set_error(MIMRL_ERR_ARG, "mimrl_stage_grads_part: GRU encoders only");
  return h->run(2, part == 0 ? 3 : 4);
}

int mimrl_forward(mimrl_handle* h, int train_mode, int with_losses) {
  if (!h) return set_error(MIMRL_ERR_ARG, "null handle");
  if (!h->bound) return set_error(MIMRL_ERR_STATE, "mimrl_bind must be called first");
  MX(h->ensure_images());
  hipLaunchKernelGGL(begin_stage_kernel, dim3(1), dim3(64), 0, h->stream, h->d_ints, (int*)nullptr, h->bufs.scalars, 32, 32);
  LAUNCH_CHECK();
  h->ev_next = 0;
  MX(h->model_forward(train_mode != 0, false));
  if (!with_losses) return MIMRL_OK;
  hipLaunchKernelGGL(mae_kernel, dim3(1), dim3(256), 0, h->stream, h->bufs.pred, h->bufs.labels, (float*)nullptr,
                     h->bufs.scalars + MIMRL_S2_TASK, h->cfg.batch);
  LAUNCH_CHECK();
  h->ev_next = 0;
  if (h->bank_rows > 0) { MX(h->fork(4, 4)); MX(h->knn_launch(2, h->S(4))); MX(h->estimators_all(2, false, false)); }
  hipLaunchKernelGGL(finalize_stage2_kernel, dim3(1), dim3(64), 0, h->stream, h->bufs.scalars, h->mi_raw, h->cmi_raw,
                     h->coef2(), h->bank_rows > 0 ? 1 : 0);
  LAUNCH_CHECK();
  return MIMRL_OK;
}

int mimrl_estimate(mimrl_handle* h, int stage) {
  if (!h) return set_error(MIMRL_ERR_ARG, "null handle");
  if (!h->bound) return set_error(MIMRL_ERR_STATE, "mimrl_bind must be called first");
  if (stage != 1 && stage != 2) return set_error(MIMRL_ERR_ARG, "stage must be 1 or 2");
  if (h->bank_rows <= 0) return set_error(MIMRL_ERR_STATE, "mimrl_estimate needs non-empty banks");
  MX(h->ensure_images());
  h->ev_next = 0;
  MX(h->fork(4, 4));
  MX(h->knn_launch(stage, h->S(4)));
  MX(h->estimators_all(stage, false, false));
  if (stage == 1) {
    hipLaunchKernelGGL(finalize_stage1_kernel, dim3(1), dim3(64), 0, h->stream, h->bufs.scalars, h->mi_raw, h->cmi_raw,
                       h->bce_raw, h->coef1());
  } else {
    hipLaunchKernelGGL(mae_kernel, dim3(1), dim3(256), 0, h->stream, h->bufs.pred, h->bufs.labels, (float*)nullptr,
                       h->bufs.scalars + MIMRL_S2_TASK, h->cfg.batch);
    hipLaunchKernelGGL(finalize_stage2_kernel, dim3(1), dim3(64), 0, h->stream, h->bufs.scalars, h->mi_raw, h->cmi_raw,
                       h->coef2(), 1);
  }
  LAUNCH_CHECK();
  return MIMRL_OK;
}

// ---- test probes: one sub-block of the step, run by the engine's OWN code path (same kernels, buffers and precision mode as a step of
//      this handle) on caller-supplied operands.  The fused bf16 kernels (cube_fwd_fused, daxis / kmix / laxis_bwd, concat_fwd / concat_bwd,
//      mlp_img8 + mi_sep_nce) have no stand-alone entry: these are how tests/test_gpu_fused_oracle.py compares them with the oracle.
int mimrl_probe_cube(mimrl_handle* h, const float* x, float* out, const float* dout, float* dx) {
  if (!h || !x || !out) return set_error(MIMRL_ERR_ARG, "null argument");
  if (!h->bound) return set_error(MIMRL_ERR_STATE, "mimrl_bind must be called first");
  if (dout && !dx) return set_error(MIMRL_ERR_ARG, "dout without dx");
  const mimrl_cfg& c = h->cfg;
  const int nb = c.n_blocks;
  const size_t nin = (size_t)c.batch * c.time_len * 3 * c.d_common;
  const size_t nout = (size_t)c.batch * c.d_outs[nb - 1][0] * c.d_outs[nb - 1][1] * c.d_outs[nb - 1][2];
  h->ev_next = 0;
  HIPX(hipMemcpyAsync(h->cube0, x, sizeof(float) * nin, hipMemcpyDeviceToDevice, h->stream));
  h->bf16 = (h->prec & MIMRL_PREC_BF16_GEMM_FWD) != 0;
  MX(h->cube_forward(false, dout != nullptr));
  HIPX(hipMemcpyAsync(out, h->bb[nb - 1].d.z, sizeof(float) * nout, hipMemcpyDeviceToDevice, h->stream));
  if (!dout) return MIMRL_OK;
  if (nout > h->gbuf_floats) return set_error(MIMRL_ERR_STATE, "probe: gradient buffer too small");
  HIPX(hipMemsetAsync(h->bufs.main_g, 0, sizeof(float) * h->layout.floats[MIMRL_GROUP_MAIN], h->stream));
  HIPX(hipMemcpyAsync(h->gbuf[0], dout, sizeof(float) * nout, hipMemcpyDeviceToDevice, h->stream));
  h->bf16 = (h->prec & MIMRL_PREC_BF16_GEMM_BWD) != 0;
  const bool pre = h->wtT_prebuilt;
  h->wtT_prebuilt = false;                 // (the combined step builds the D-axis weight images beside the encoders; here: in place)
  h->deferred.clear();
  int ci = 0;
  int r = h->cube_backward(0, &ci);
  if (r == 0) r = h->flush_deferred();
  h->wtT_prebuilt = pre;
  h->kmix_pg_on_side3 = false;
  h->bf16 = (h->prec & MIMRL_PREC_BF16_GEMM_FWD) != 0;
  MX(r);
  MX(h->join(0, 5));
  HIPX(hipMemcpyAsync(dx, h->gbuf[ci], sizeof(float) * nin, hipMemcpyDeviceToDevice, h->stream));
  h->grads_clean[2] = false;
  return MIMRL_OK;
}

int mimrl_probe_encoders(mimrl_handle* h, float* cube_x, const float* dcube, const float* dmean) {
  if (!h || !cube_x) return set_error(MIMRL_ERR_ARG, "null argument");
  if (!h->bound) return set_error(MIMRL_ERR_STATE, "mimrl_bind must be called first");
  const mimrl_cfg& c = h->cfg;
  const size_t nin = (size_t)c.batch * c.time_len * 3 * c.d_common, nf = (size_t)c.batch * c.d_common;
  MX(h->ensure_images());
  h->ev_next = 0;
  h->bf16 = (h->prec & MIMRL_PREC_BF16_GEMM_FWD) != 0;
  MX(h->model_forward(true, dcube != nullptr, 0, 0));
  HIPX(hipMemcpyAsync(cube_x, h->cube0, sizeof(float) * nin, hipMemcpyDeviceToDevice, h->stream));
  if (!dcube) return MIMRL_OK;
  if (nin > h->gbuf_floats) return set_error(MIMRL_ERR_STATE, "probe: gradient buffer too small");
  HIPX(hipMemsetAsync(h->bufs.main_g, 0, sizeof(float) * h->layout.floats[MIMRL_GROUP_MAIN], h->stream));
  HIPX(hipMemcpyAsync(h->gbuf[0], dcube, sizeof(float) * nin, hipMemcpyDeviceToDevice, h->stream));
  if (dmean) HIPX(hipMemcpyAsync(h->dfeat + nf, dmean, sizeof(float) * 3 * nf, hipMemcpyDeviceToDevice, h->stream));
  else HIPX(hipMemsetAsync(h->dfeat, 0, sizeof(float) * 4 * nf, h->stream));
  h->bf16 = (h->prec & MIMRL_PREC_BF16_GEMM_BWD) != 0;
  h->deferred.clear();
  h->head_gather_on = false; h->ev_dmean = nullptr; h->kmix_pg_on_side3 = false;
  const int r = h->encoders_backward(h->gbuf[0]);
  h->bf16 = (h->prec & MIMRL_PREC_BF16_GEMM_FWD) != 0;
  h->grads_clean[2] = false;
  return r;
}

int mimrl_probe_mi(mimrl_handle* h, int stage, float* mi, float* scores, float* dtin_out) {
  if (!h || !mi) return set_error(MIMRL_ERR_ARG, "null argument");
  if (!h->bound) return set_error(MIMRL_ERR_STATE, "mimrl_bind must be called first");
  if (stage != 1 && stage != 2) return set_error(MIMRL_ERR_ARG, "stage must be 1 or 2");
  const int B = h->cfg.batch;
  const bool bf_fwd = (h->prec & MIMRL_PREC_BF16_GEMM_FWD) != 0, bf_bwd = (h->prec & MIMRL_PREC_BF16_GEMM_BWD) != 0;
  h->ev_next = 0;
  MX(h->ensure_images());
  h->imgT_ready = false;
  if (bf_bwd && h->fused_mlp && h->crit_imgT && h->ttab.n > 0) {   // as estimators_all does, but in line
    MX(bf16_transposed_images(h->stream, h->bufs.crit_p, h->crit_imgT, h->ttab));
    h->imgT_ready = true;
  }
  if (stage == 1) HIPX(hipMemsetAsync(h->bufs.crit_g, 0, sizeof(float) * h->layout.floats[MIMRL_GROUP_CRITIC], h->stream));
  h->bf16 = bf_fwd;
  int r = h->mi_forward(stage, true);
  if (r == 0) { h->bf16 = bf_bwd; h->wg_helper = -1; r = h->mi_backward(stage); }
  h->bf16 = bf_fwd;
  MX(r);
  MX(h->join(0, 5));
  HIPX(hipMemcpyAsync(mi, h->mi_raw, sizeof(float) * 2 * NE_MI, hipMemcpyDeviceToDevice, h->stream));
  if (scores) {
    if (h->cfg.critic_type != MIMRL_CRITIC_CONCAT) return set_error(MIMRL_ERR_ARG, "probe: the separable fused path does not materialise scores");
    HIPX(hipMemcpyAsync(scores, h->scores, sizeof(float) * NE_MI * B * B, hipMemcpyDeviceToDevice, h->stream));
  }
  if (dtin_out) {
    if (stage != 2) return set_error(MIMRL_ERR_ARG, "probe: tower-input gradients exist in stage 2 only");
    HIPX(hipMemcpyAsync(dtin_out, h->dtin, sizeof(float) * 2 * NE_MI * B * EMB, hipMemcpyDeviceToDevice, h->stream));
  }
  if (stage == 1) h->grads_clean[1] = false;
  return MIMRL_OK;
}

int mimrl_probe_cmi(mimrl_handle* h, int stage, const float* cmi_in_, float* logits_out, float* vals, float* dcin_out) {
  if (!h || !cmi_in_ || !logits_out || !vals) return set_error(MIMRL_ERR_ARG, "null argument");
  if (!h->bound) return set_error(MIMRL_ERR_STATE, "mimrl_bind must be called first");
  if (stage != 1 && stage != 2) return set_error(MIMRL_ERR_ARG, "stage must be 1 or 2");
  const int n = h->nprod();
  const bool bf_fwd = (h->prec & MIMRL_PREC_BF16_GEMM_FWD) != 0, bf_bwd = (h->prec & MIMRL_PREC_BF16_GEMM_BWD) != 0;
  h->ev_next = 0;
  MX(h->ensure_images());
  h->imgT_ready = false;
  if (bf_bwd && h->fused_mlp && h->crit_imgT && h->ttab.n > 0) {
    MX(bf16_transposed_images(h->stream, h->bufs.crit_p, h->crit_imgT, h->ttab));
    h->imgT_ready = true;
  }
  if (stage == 1) HIPX(hipMemsetAsync(h->bufs.crit_g, 0, sizeof(float) * h->layout.floats[MIMRL_GROUP_CRITIC], h->stream));
  HIPX(hipMemcpyAsync(h->cmi_in, cmi_in_, sizeof(float) * NE_CMI * 2 * n * 384, hipMemcpyDeviceToDevice, h->stream));
  // cmi_forward minus the kNN / assemble part (Model.py:185-219 on a caller-assembled batch) ...
  h->bf16 = bf_fwd;
  const int cdims[5] = {3 * EMB, HID, HID, HID, 2};
  int r = h->mlp_stack_forward(NE_CMI, 2 * n, 2 * n, h->cmi0, h->cmi_stride, 4, h->cmi_l, cdims, h->cmi_in, h->cc, h->logits);
  if (r == 0) r = cmi_loss_fwd_bwd(h->stream, h->logits, h->dlogits, h->bce_raw, h->cmi_raw, h->g_bce(stage), h->g_cmi(stage), NE_CMI, n, h->cfg.cmi_hardtanh);
  // ... and cmi_backward as a step runs it
  if (r == 0) { h->bf16 = bf_bwd; h->wg_helper = -1; r = h->cmi_backward(stage); }
  h->bf16 = bf_fwd;
  MX(r);
  MX(h->join(0, 5));
  HIPX(hipMemcpyAsync(logits_out, h->logits, sizeof(float) * NE_CMI * 2 * n * 2, hipMemcpyDeviceToDevice, h->stream));
  HIPX(hipMemcpyAsync(vals, h->bce_raw, sizeof(float) * NE_CMI, hipMemcpyDeviceToDevice, h->stream));
  HIPX(hipMemcpyAsync(vals + NE_CMI, h->cmi_raw, sizeof(float) * NE_CMI, hipMemcpyDeviceToDevice, h->stream));
  if (dcin_out) {
    if (stage != 2) return set_error(MIMRL_ERR_ARG, "probe: classifier-input gradients exist in stage 2 only");
    HIPX(hipMemcpyAsync(dcin_out, h->dcin, sizeof(float) * NE_CMI * 2 * n * 384, hipMemcpyDeviceToDevice, h->stream));
  }
  if (stage == 1) h->grads_clean[1] = false;
  return MIMRL_OK;
}

int mimrl_set_kernel_stamps(mimrl_handle* h, unsigned long long* ring, int slots) {
  if (!h) return set_error(MIMRL_ERR_ARG, "null handle");
  if (ring && (slots < 1 || (slots & (slots - 1)))) return set_error(MIMRL_ERR_ARG, "stamp ring: slots must be a power of two");
  h->kstamp.ring = ring; h->kstamp.step = h->d_ints; h->kstamp.slots = ring ? slots : 0;
  h->drop_graphs();     // captured launches bake the kernel arguments in
  return MIMRL_OK;
}

int mimrl_profile_enable(mimrl_handle* h, int on) {
  if (!h) return set_error(MIMRL_ERR_ARG, "null handle");
  h->prof_on = on != 0;
  return MIMRL_OK;
}

int mimrl_profile_read(mimrl_handle* h, float* ms_sum, int32_t* launches) {
  if (!h || !ms_sum || !launches) return set_error(MIMRL_ERR_ARG, "null argument");
  HIPX(hipStreamSynchronize(h->stream));
  for (int p = 0; p < MIMRL_NPHASES; ++p) {
    double acc = 0.0;
    for (auto& ev : h->prof_ev[p]) {
      float ms = 0.f;
      HIPX(hipEventElapsedTime(&ms, ev.first, ev.second));
      acc += ms;
      h->prof_pool.push_back(ev);
    }
    ms_sum[p] = (float)acc;
    launches[p] = (int32_t)h->prof_ev[p].size();
    h->prof_ev[p].clear();
  }
  return MIMRL_OK;
}

int mimrl_profile_read_gemm(mimrl_handle* h, double out[4]) {
  if (!h || !out) return set_error(MIMRL_ERR_ARG, "null argument");
  HIPX(hipStreamSynchronize(h->stream));
  HIPX(hipDeviceSynchronize());
  out[0] = out[1] = out[2] = 0.0; out[3] = (double)h->prof_gemm.size();
  for (auto& g : h->prof_gemm) {
    float ms = 0.f;
    HIPX(hipEventElapsedTime(&ms, g.a, g.b));
    out[0] += g.flops; out[1] += g.bytes; out[2] += ms;
    h->prof_pool.push_back({g.a, g.b});
  }
  h->prof_gemm.clear();
  return MIMRL_OK;
}

int mimrl_params_changed(mimrl_handle* h) {
  if (!h) return set_error(MIMRL_ERR_ARG, "null handle");
  h->img_valid = false;
  h->imgT_valid = false;
  return MIMRL_OK;
}

int mimrl_set_knn_override_mask(mimrl_handle* h, int stage, unsigned call_mask) {
  if (!h || (stage != 1 && stage != 2)) return set_error(MIMRL_ERR_ARG, "bad handle / stage");
  if (call_mask && !h->bufs.knn_override) return set_error(MIMRL_ERR_STATE, "mimrl_buffers.knn_override is not bound");
  if (h->knn_ovr_mask[stage - 1] == (call_mask & 63u)) return MIMRL_OK;
  HIPX(hipStreamSynchronize(h->user_stream));
  h->drop_graphs();
  h->knn_ovr_mask[stage - 1] = call_mask & 63u;
  return MIMRL_OK;
}

int mimrl_stage2_forward_tail(mimrl_handle* h) { return h ? h->run_fwd2_tail() : set_error(MIMRL_ERR_ARG, "null handle"); }

int mimrl_set_grad_scale(mimrl_handle* h, float scale) {
  if (!h) return set_error(MIMRL_ERR_ARG, "null handle");
  if (scale == h->grad_scale) return MIMRL_OK;
  HIPX(hipStreamSynchronize(h->user_stream));
  h->drop_graphs();
  h->grad_scale = scale;
  return MIMRL_OK;
}

int mimrl_comm_unique_id(void* out128) { return mimrl::comm_unique_id(out128); }

int mimrl_set_comm(mimrl_handle* h, const void* unique_id128, int world, int rank) {
  if (!h) return set_error(MIMRL_ERR_ARG, "null handle");
  HIPX(hipStreamSynchronize(h->user_stream));
  h->drop_graphs();
  if (h->comm) { MX(mimrl::comm_destroy(h->comm)); h->comm = nullptr; }
  h->comm_world = 1; h->comm_rank = 0;
  if (!unique_id128) return MIMRL_OK;                       // NULL: back to a replica without collectives
  MX(mimrl::comm_init(&h->comm, unique_id128, world, rank));
  h->comm_world = world; h->comm_rank = rank;
  if (!h->comm_s) HIPX(hipStreamCreateWithFlags(&h->comm_s, hipStreamNonBlocking));
  const char* sp = knob("MIMRL_DDP_SPLIT");
  h->comm_split = !(sp && sp[0] == '0');
  // one eager collective now: RCCL's lazy set-up (buffers, proxy threads) must not happen inside a stream capture
  if (h->bound) {
    HIPX(hipMemsetAsync(h->bufs.scalars, 0, sizeof(float), h->user_stream));
    MX(mimrl::comm_allreduce_sum(h->comm, h->bufs.scalars, 1, h->user_stream));
    MX(mimrl::comm_allreduce_sum(h->comm, h->bufs.scalars, 1, h->comm_s));
    HIPX(hipStreamSynchronize(h->user_stream)); HIPX(hipStreamSynchronize(h->comm_s));
  }
  return MIMRL_OK;
}

int64_t mimrl_main_late_offset(const mimrl_handle* h) { return h ? (int64_t)h->layout.late_offset : 0; }

int mimrl_set_stage2_prefetch(mimrl_handle* h, int on) {
  if (!h) return set_error(MIMRL_ERR_ARG, "null handle");
#ifdef MIMRL_DET
  on = 0;                    // deterministic build: no second forward pass on its own stream (same results, sequential schedule)
#endif
  if ((on != 0) == h->prefetch && (on == 2) == h->defer_tail) return MIMRL_OK;
  if (on && !h->pre_stream) HIPX(hipStreamCreateWithFlags(&h->pre_stream, hipStreamNonBlocking));
  HIPX(hipStreamSynchronize(h->user_stream));
  h->drop_graphs();
  h->prefetch = on != 0;
  h->defer_tail = on == 2;
  h->fwd2_pending = false;
  h->tail2_needed = false;
  return MIMRL_OK;
}

int64_t mimrl_workspace_bytes(const mimrl_handle* h) { return h ? (int64_t)h->ws_bytes : 0; }

void mimrl_destroy(mimrl_handle* h) {
  if (!h) return;
  h->drop_graphs();
  for (auto g : h->retired) (void)hipGraphExecDestroy(g);
  h->retired.clear();
  for (int p = 0; p < MIMRL_NPHASES; ++p)
    for (auto& ev : h->prof_ev[p]) h->prof_pool.push_back(ev);
  for (auto& g : h->prof_gemm) h->prof_pool.push_back({g.a, g.b});
  for (auto& ev : h->prof_pool) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
  for (int i = 0; i < mimrl_handle::NSIDE; ++i)
    if (h->side[i]) (void)hipStreamDestroy(h->side[i]);
  for (auto e : h->ev_pool) (void)hipEventDestroy(e);
  if (h->cap_stream) (void)hipStreamDestroy(h->cap_stream);
  if (h->pre_stream) (void)hipStreamDestroy(h->pre_stream);
  if (h->comm) (void)mimrl::comm_destroy(h->comm);
  if (h->comm_s) (void)hipStreamDestroy(h->comm_s);
  if (h->ws) (void)hipFree(h->ws);
  delete h;
}

// ---- operator-level entry points ------------------------------------------------------------
int mimrl_op_gemm(void* stream, const float* A, const float* B, float* C, int M, int N, int K, int batch,
                  const int64_t st[9], const float* bias_n, const float* bias_m, float alpha, float beta, int act,
                  int precision) {
  if (!st) return set_error(MIMRL_ERR_ARG, "null strides");
  GemmDesc d;
  d.A = A; d.B = B; d.C = C; d.M = M; d.N = N; d.K = K; d.batch = batch;
  d.sa_m = st[0]; d.sa_k = st[1]; d.sa_b = st[2]; d.sb_k = st[3]; d.sb_n = st[4]; d.sb_b = st[5];
  d.sc_m = st[6]; d.sc_n = st[7]; d.sc_b = st[8];
  d.bias_n = bias_n; d.bias_m = bias_m; d.alpha = alpha; d.beta = beta; d.act = act & 0xff;
  d.atomic = (act >> 8) & 1;   // bit 8: accumulate with atomics (enables split-K / batch-group reduction)
  return gemm(reinterpret_cast<hipStream_t>(stream), d, (precision & 1) != 0);
}

int mimrl_op_gemm_ex(void* stream, const float* A, const float* B, float* C, int M, int N, int K, int batch, const int64_t st[9],
                     const float* A2, const float* B2, int K2, const int64_t st2[6], int a_gap_at, int a_gap_rows, const float* bias_n,
                     const float* gradact_u, float* colsum, int act, int precision) {
  if (!st || (A2 && !st2)) return set_error(MIMRL_ERR_ARG, "null strides");
  GemmDesc d;
  d.A = A; d.B = B; d.C = C; d.M = M; d.N = N; d.K = K; d.batch = batch;
  d.sa_m = st[0]; d.sa_k = st[1]; d.sa_b = st[2]; d.sb_k = st[3]; d.sb_n = st[4]; d.sb_b = st[5];
  d.sc_m = st[6]; d.sc_n = st[7]; d.sc_b = st[8];
  if (A2) {
    d.A2 = A2; d.B2 = B2; d.K2 = K2;
    d.sa2_m = st2[0]; d.sa2_k = st2[1]; d.sa2_b = st2[2]; d.sb2_k = st2[3]; d.sb2_n = st2[4]; d.sb2_b = st2[5];
  }
  d.a_gap_at = a_gap_at; d.a_gap_rows = a_gap_rows;
  d.bias_n = bias_n; d.gradact_u = gradact_u; d.colsum = colsum; d.act = act & 0xff; d.atomic = (act >> 8) & 1;
  return gemm(reinterpret_cast<hipStream_t>(stream), d, (precision & 1) != 0);
}

int mimrl_op_gemm16(void* stream, const void* A, const void* B, void* C, int M, int N, int K, int batch, const int64_t st[9],
                    const void* A2, const void* B2, int K2, const int64_t st2[6], int batch_in, const int64_t st_bo[5],
                    const float* bias_n, int flags) {
  if (!st || (A2 && !st2) || (batch_in > 0 && !st_bo)) return set_error(MIMRL_ERR_ARG, "null strides");
  if (batch_in < 0 || (batch_in > 0 && batch % batch_in != 0)) return set_error(MIMRL_ERR_ARG, "batch_in must divide batch");
  GemmDesc d;
  d.A = static_cast<const float*>(A); d.B = static_cast<const float*>(B); d.C = static_cast<float*>(C);
  d.M = M; d.N = N; d.K = K; d.batch = batch;
  d.sa_m = st[0]; d.sa_k = st[1]; d.sa_b = st[2]; d.sb_k = st[3]; d.sb_n = st[4]; d.sb_b = st[5];
  d.sc_m = st[6]; d.sc_n = st[7]; d.sc_b = st[8];
  if (A2) {
    d.A2 = static_cast<const float*>(A2); d.B2 = static_cast<const float*>(B2); d.K2 = K2;
    d.sa2_m = st2[0]; d.sa2_k = st2[1]; d.sa2_b = st2[2]; d.sb2_k = st2[3]; d.sb2_n = st2[4]; d.sb2_b = st2[5];
  }
  if (batch_in > 0) { d.batch_in = batch_in; d.sa_bo = st_bo[0]; d.sb_bo = st_bo[1]; d.sc_bo = st_bo[2]; d.bias_n_bo = st_bo[3]; }
  if (st_bo) d.bias_n_b = st_bo[4];
  d.bias_n = bias_n;
  d.a_bf16 = flags & 1; d.b_bf16 = (flags >> 1) & 1; d.f16 = (flags >> 2) & 1; d.c_f16 = (flags >> 3) & 1; d.atomic = (flags >> 4) & 1;
  if (flags >> 8) { d.a_gap_at = (flags >> 8) & 0xfff; d.a_gap_rows = (flags >> 20) & 0xfff; }
  return gemm(reinterpret_cast<hipStream_t>(stream), d, true);
}

int mimrl_op_gemm_wgrad_group(void* stream, int n, const float* const* A, const float* const* B, float* const* C, const int32_t* dims,
                              const int64_t* st, int precision) {
  if (n <= 0 || n > 12 || !A || !B || !C || !dims || !st) return set_error(MIMRL_ERR_ARG, "gemm_wgrad_group: 1..12 problems");
  GemmDesc d[12];
  for (int i = 0; i < n; ++i) {
    d[i].A = A[i]; d[i].B = B[i]; d[i].C = C[i];
    d[i].M = dims[4 * i]; d[i].N = dims[4 * i + 1]; d[i].K = dims[4 * i + 2]; d[i].batch = dims[4 * i + 3];
    const int64_t* q = st + 9 * i;
    d[i].sa_m = q[0]; d[i].sa_k = q[1]; d[i].sa_b = q[2]; d[i].sb_k = q[3]; d[i].sb_n = q[4]; d[i].sb_b = q[5];
    d[i].sc_m = q[6]; d[i].sc_n = q[7]; d[i].sc_b = q[8];
    d[i].atomic = 1;
  }
  return gemm_group_splitk(reinterpret_cast<hipStream_t>(stream), d, n, (precision & 1) != 0);
}

int64_t mimrl_op_gru_saved_floats(int B, int T) { return gru_saved_floats(B, T); }

int mimrl_op_gru_forward(void* stream, const float* gx_f, const float* gx_r, const float* whh_f, const float* whh_r,
                         const float* bhh_f, const float* bhh_r, const int32_t* lens, float* out, float* saved_f,
                         float* saved_r, int B, int T, int precision) {
  GruFwdArgs a;
  std::memset(&a, 0, sizeof a);
  a.B = B; a.T = T; a.out_ld = 2 * H; a.nmod = 1; a.btv = gru_pick_btv(B, 1);
  a.lens[0] = lens; a.lens[1] = lens;
  a.seq[0][0] = GruSeq{gx_f, whh_f, bhh_f, out, saved_f};
  a.seq[0][1] = GruSeq{gx_r, whh_r, bhh_r, out, saved_r};
  return gru_forward(reinterpret_cast<hipStream_t>(stream), a, (precision & 1) != 0);
}

int mimrl_op_gru_backward(void* stream, const float* whh_f, const float* whh_r, const float* saved_f,
                          const float* saved_r, const int32_t* lens, const float* out, const float* dout, float* dg_f,
                          float* dg_r, float* hprev_f, float* hprev_r, int B, int T, int precision) {
  GruBwdArgs a;
  std::memset(&a, 0, sizeof a);
  a.B = B; a.T = T; a.out_ld = 2 * H; a.dout_ld = 2 * H; a.dout_off = H; a.nmod = 1; a.btv = gru_pick_btv(B, 1);
  a.lens[0] = lens; a.lens[1] = lens;
  a.seq[0][0] = GruSeqBwd{whh_f, saved_f, out, dout, dg_f, hprev_f, nullptr, nullptr};
  a.seq[0][1] = GruSeqBwd{whh_r, saved_r, out, dout, dg_r, hprev_r, nullptr, nullptr};
  return gru_backward(reinterpret_cast<hipStream_t>(stream), a, (precision & 1) != 0);
}

int mimrl_op_mi_bound(void* stream, const float* scores, float* dscores, float* mi, const float* gscale, int E, int B,
                      int bound) {
  return mi_bound_fwd_bwd(reinterpret_cast<hipStream_t>(stream), scores, dscores, mi, nullptr, gscale, E, B, bound, 0u);
}
int mimrl_op_mi_bound_baseline(void* stream, float* scores, float* dscores, float* mi, const float* gscale, const float* lb,
                               float* dlb, int E, int B, int bound) {
  return mi_bound_fwd_bwd(reinterpret_cast<hipStream_t>(stream), scores, dscores, mi, nullptr, gscale, E, B, bound, 0u, lb, dlb, B);
}
int mimrl_op_mi_bound_ex(void* stream, const float* scores, float* dscores, float* mi, float* mi_loss, const float* gscale,
                         int E, int B, int bound, uint32_t lossform) {
  return mi_bound_fwd_bwd(reinterpret_cast<hipStream_t>(stream), scores, dscores, mi, mi_loss, gscale, E, B, bound, lossform);
}

int mimrl_op_mi_sep_infonce(void* stream, const float* tout, float* dtout, float* mi, float* mi_loss, const float* gscale, int E,
                            int B, int tiled) {
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (!tiled) return mi_sep_fused(st, tout, dtout, mi, mi_loss, gscale, E, B, BOUND_INFONCE, 0x1fu, dtout ? 1 : 0);
  return mi_sep_nce_tiled(st, tout, dtout, mi, mi_loss, gscale, E, B, dtout ? 1 : 0);   // accumulates: caller zeroes mi / mi_loss / dtout
}

int mimrl_op_knn(void* stream, const float* Z, int dz, int N, const int32_t* anchors, int m, int k, int32_t* idx_out) {
  KnnArgs a;
  std::memset(&a, 0, sizeof a);
  a.call[0] = KnnCall{Z, dz, anchors, idx_out};
  a.N = N; a.m = m; a.k = k; a.ncall = 1;
  return knn_sample(reinterpret_cast<hipStream_t>(stream), a);
}

int mimrl_probe_knn(mimrl_handle* h, int stage, int32_t* idx_out) {
  if (!h || !idx_out) return set_error(MIMRL_ERR_ARG, "mimrl_probe_knn: null argument");
  if (stage != 1 && stage != 2) return set_error(MIMRL_ERR_ARG, "stage must be 1 or 2");
  HIPX(hipMemcpyAsync(idx_out, stage == 2 ? h->knn_idx2 : h->knn_idx, sizeof(int32_t) * NE_CMI * h->nprod(), hipMemcpyDeviceToDevice, h->user_stream));
  return MIMRL_OK;
}

int mimrl_op_sample_anchors(void* stream, int32_t* anchors_out, int ncall, int m, int N, uint64_t seed, const int32_t* step,
                            uint32_t stream_id, int step_add) {
  if (!anchors_out || !step || ncall < 1 || ncall > KNN_MAX_CALLS) return set_error(MIMRL_ERR_ARG, "mimrl_op_sample_anchors: bad argument");
  AnchorDraws d;
  d.n = ncall;
  for (int c = 0; c < ncall; ++c) { d.out[c] = anchors_out + (size_t)c * m; d.call[c] = c; d.stream_id[c] = stream_id; d.step_add[c] = step_add; }
  return sample_anchors(reinterpret_cast<hipStream_t>(stream), d, m, N, (uint32_t)seed, (uint32_t)(seed >> 32), step);
}

int mimrl_op_cmi_loss(void* stream, const float* logits, float* dlogits, float* bce, float* cmi, const float* g_bce,
                      const float* g_cmi, int E, int n, int hardtanh) {
  return cmi_loss_fwd_bwd(reinterpret_cast<hipStream_t>(stream), logits, dlogits, bce, cmi, g_bce, g_cmi, E, n, hardtanh);
}

static int fill_mlp_args(MlpFusedArgs* fa, int nb, int rows, int brows, int nl, const int32_t* dims, const float* const* W,
                         int64_t pstride) {
  if (!dims || !W || nl < 1 || nl > MLPF_MAX_LAYERS) return set_error(MIMRL_ERR_ARG, "mlp_stack: bad arguments");
  std::memset(fa, 0, sizeof *fa);
  fa->nb = nb; fa->rows = rows; fa->brows = brows; fa->nl = nl; fa->pstride = pstride;
  for (int l = 0; l <= nl; ++l) fa->dims[l] = dims[l];
  for (int l = 0; l < nl; ++l) fa->W[l] = W[l];
  return MIMRL_OK;
}

int mimrl_op_mlp_stack_forward(void* stream, int nb, int rows, int brows, int nl, const int32_t* dims, const float* const* W,
                               const float* const* b, int64_t pstride, const float* in, float* const* act, float* out) {
  MlpFusedArgs fa;
  MX(fill_mlp_args(&fa, nb, rows, brows, nl, dims, W, pstride));
  if (!b || !in || !out || (nl > 1 && !act)) return set_error(MIMRL_ERR_ARG, "mlp_stack_forward: null argument");
  for (int l = 0; l < nl; ++l) { fa.b[l] = b[l]; if (l < nl - 1) fa.act[l] = act[l]; }
  fa.in = in; fa.out = out;
  return mlp_stack_fwd_fused(reinterpret_cast<hipStream_t>(stream), fa);
}

int mimrl_op_mlp_stack_backward(void* stream, int nb, int rows, int brows, int nl, const int32_t* dims, const float* const* W,
                                int64_t pstride, const float* const* act, const float* dout, float* const* dz, float* din,
                                float* const* db) {
  MlpFusedArgs fa;
  MX(fill_mlp_args(&fa, nb, rows, brows, nl, dims, W, pstride));
  if (!dout || (nl > 1 && (!act || !dz))) return set_error(MIMRL_ERR_ARG, "mlp_stack_backward: null argument");
  for (int l = 0; l < nl - 1; ++l) { fa.act[l] = const_cast<float*>(act[l]); fa.dz[l + 1] = dz[l + 1]; fa.db[l] = db ? db[l] : nullptr; }
  fa.dout = dout; fa.din = din;
  return mlp_stack_bwd_fused(reinterpret_cast<hipStream_t>(stream), fa);
}

int mimrl_op_adam(void* stream, float* p, float* g, float* m, float* v, int64_t n, const float* lr, const int32_t* step,
                  float beta1, float beta2, float eps, float weight_decay, float clip) {
  AdamArgs a;
  a.p = p; a.g = g; a.m = m; a.v = v; a.n = n; a.lr = lr; a.step = step;
  a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.weight_decay = weight_decay; a.clip = clip;
  return adam_step(reinterpret_cast<hipStream_t>(stream), a);
}

}  // extern "C"
