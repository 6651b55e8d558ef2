// Engine: the two-stage training step of MIMRL orchestrated on one HIP stream (optionally replayed as hipGraphs).
//
//   stage 1  (Solver.py:205-214): model forward -> 5 MI + 6 CMI estimators -> backward into the CRITIC weights
//            -> value-clip + Adam on the critic bucket.
//   stage 2  (Solver.py:221-236): model forward (activations kept) -> estimators (data gradients only) -> MAE ->
//            backward through head / CubeMLP / LN / bi-GRU BPTT / W_t -> value-clip + Adam on the main bucket.
// The reference back-propagates stage 1 through the main model and stage 2 into the critic weights as well, but
// those gradients are never applied: they are skipped here with no effect on any parameter.//
// Round 5: the engine is split over six translation units (it was one file of 3,600 lines):
//   engine.h              this header: constants, helper types, the handle (struct mimrl_handle: arena pointers, schedule state, stream helpers)
//   engine_kernels.hip    the step's own small kernels (begin_stage, MAE, finalize_stage1 / 2, stage_boundary) behind launch wrappers
//   engine_arena.hip      parameter resolution, workspace arena, weight-image tables
//   engine_forward.hip    model forward: encoders (GRU / LSTM / conv), forward tail, CubeMLP
//   engine_backward.hip   model backward: CubeMLP, LayerNorm, BPTT, weight gradients
//   engine_estimators.hip MLP stacks, kNN sampler, MI / CMI estimators forward + backward
//   engine_step.hip       enqueue_grads / enqueue_apply, the data-parallel reduce, graph capture and post-processing, run / run_step
//   engine_abi.hip        the C ABI (include/mimrl.h): create / bind / steps / probes / operator-level entry points
#pragma once
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
#include "cube_long.h"
#include "estimator_ops.h"
#include "gemm.h"
#include "gru.h"
#include "gru_wgrad.h"
#include "concat_dw.h"
#include "lstm.h"
#include "layout.h"
#include "comm.h"
#include "model_ops.h"

namespace mimrl {

// (a NAMED namespace: these types are members of mimrl_handle, which every engine_*.hip sees)
namespace eng {


constexpr int H = 128, G = 384, HID = 256, EMB = 128;
constexpr int NE_MI = 5, NE_CMI = 6;
constexpr int ACT_SLACK = 4 * MLPF_MAX_WIDTH;   // floats behind every saved-activation buffer of the fused MLP stacks (MlpFusedArgs::act_slack)

// feature slots: F,T,A,V ; 4 = labels (C)
enum { FT_F = 0, FT_T = 1, FT_A = 2, FT_V = 3, FT_C = 4 };
const int kMiWire[NE_MI][2] = {{FT_F, FT_T}, {FT_F, FT_A}, {FT_F, FT_V}, {FT_T, FT_A}, {FT_T, FT_V}};   // Model.py:313-319
const int kCmiWire[NE_CMI][3] = {{FT_A, FT_C, FT_T}, {FT_T, FT_A, FT_C}, {FT_V, FT_C, FT_T},           // Model.py:323-339
                                 {FT_T, FT_V, FT_C}, {FT_T, FT_C, FT_A}, {FT_T, FT_C, FT_V}};
static const char* const kVmi[NE_MI] = {"f_t", "f_a", "f_v", "t_a", "t_v"};
static const char* const kVcmi[NE_CMI] = {"ac_t", "ta_c", "vc_t", "tv_c", "tc_a", "tc_v"};

// engine_kernels.hip: the step's own kernels (one thread block each), enqueued on `s`
void launch_begin_stage(hipStream_t s, int* rng_step, int* adam_step, float* scalars, int scal_off, int scal_n);
void launch_bucket_to_bf16(hipStream_t s, const float* src, void* dst, long n);
void launch_bucket_from_bf16(hipStream_t s, const void* src, float* dst, long n);
void launch_mae(hipStream_t s, const float* pred, const float* y, float* dpred, float* task, int B);
void launch_finalize_stage1(hipStream_t s, float* scal, const float* mi, const float* cmi, const float* bce, const float* coef1);
void launch_finalize_stage2(hipStream_t s, float* scal, const float* mi, const float* cmi, const float* coef2, int have_mi);
void launch_stage_boundary(hipStream_t s, float* scal, const float* mi, const float* cmi, const float* bce, const float* coef1, int* rng_step,
                           int* adam_step, const float* pred, const float* y, float* dpred, int B);

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

}  // namespace eng

using namespace eng;

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
  DetRanges det_ranges;                // deterministic build: this handle's gradient buckets (det.h: DetDefer)
  float* nce_ws = nullptr;             // per-handle workspace of the row-tiled InfoNCE kernel (estimator_ops.h: NCE_WS_FLOATS)
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
  bool frag_tr_deferred = false;       // the data-gradient fragment images are still to be launched (enqueue_apply -> estimators_all: ftab_tr)
  FragTable ftab_tr;
  bool adam_frag_on = true;            // MIMRL_ADAM_FRAG=0 (mimrl_create)
  bool boundary_in_adam = false;       // the stage boundary rode on the critic Adam launch (enqueue_apply -> run)
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
  float* tailp_part = nullptr;   // chunk sums of dual_tail_pre()
  bool pre_done = false;         // the pre-CubeMLP pieces of the forward tails were written by dual_tail_pre(): model_forward(part 2) skips them
  int dual_tail_pre();           // engine_forward.hip
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
  bool laxis_bwd_long_on = true;       // MIMRL_LAXIS_BWD_LONG=0 (mimrl_create): colln_bwd + GEMM chain for the L-axis backward of long sequences
  bool laxis_long_on = true;           // MIMRL_LAXIS_LONG=0 (mimrl_create): the GEMM chain for the L axis of long sequences
  bool rec16_on = true, dwih_h16_on = true;   // MIMRL_REC16=0 / MIMRL_DWIH_H16=0 (mimrl_create)
  bool hp16_live = false;              // this pass's layer-0 BPTT reads h_prev from h0h (set by the forward pass; with the fused projection the fp32 outputs were not even written)
  bool dh0_bf16_live = false, ds_bf16_live = false;   // this pass's dh0 / ds were written as bf16 (gru_layer_backward / encoders_backward -> the BPTT launches)
  bool h16_live = false;               // h0h holds the fp16 copy of THIS pass's layer-0 outputs (set by the forward pass)
  bool w1_img_valid = false;           // w1b holds the CURRENT main parameters (set by the forward pass, cleared by the main update)
  int KP() const { return ((cfg.d_a > cfg.d_v ? cfg.d_a : cfg.d_v) + 15) & ~15; }
  bool dg_bf16 = false;                // BPTT outputs dg / h_prev stored as bf16 (GRU encoders, bf16 recurrence + bf16 backward GEMMs; MIMRL_DG_FP32=1: off)
  bool concat_compact = false;         // the last concat forward saved bitmasks (+ bf16 values) for the fused backward, not fp32 activations
  bool a0_regen_live = false;         // stage 1's concat forward did not save a0 (round 6b): concat_dw regenerates it from P and Q
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
  // mimrl_set_comm_critic_bf16: the critic bucket (13.4 MB at the benchmark sizes, the larger of the two collectives) crosses the links as
  // bf16 -- rounded once before the all-reduce, summed by RCCL in bf16, widened again in front of clip + Adam
  bool comm_crit_bf16 = false; void* comm_crit16 = nullptr;
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
  // which of the two anchor / neighbour-index sets a stage uses: its own -- except inside a pipelined critic pass (mimrl_stage1_pipe), where stage 1
  // alternates between both (the sampler of the NEXT call runs beside this call's estimators; stage 2 is not running then)
  int knn_flip = 0;
  int knn_slot(int stage) const { return (stage - 1) ^ knn_flip; }
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
    if (stage == 1 && comm_crit_bf16) {
      const long n = layout.floats[MIMRL_GROUP_CRITIC];
      launch_bucket_to_bf16(stream, bufs.crit_g, comm_crit16, n);
      LAUNCH_CHECK();
      MX(comm_allreduce_sum_bf16(comm, comm_crit16, (size_t)n, stream));
      launch_bucket_from_bf16(stream, comm_crit16, bufs.crit_g, n);
      LAUNCH_CHECK();
      return MIMRL_OK;
    }
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
    hipGraphExec_t pipe[2][4] = {{nullptr, nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr, nullptr}};   // mimrl_stage1_pipe [look-ahead off / on][forward-set parity * 2 + kNN-set parity]
    int pipe_rows[2][4] = {{-1, -1, -1, -1}, {-1, -1, -1, -1}};
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
      for (int a_ = 0; a_ < 2; ++a_) for (int b_ = 0; b_ < 4; ++b_) retire(gsets[q].pipe[a_][b_]);
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
  // (rounds 2-5 had an accuracy-bisection knob here -- MIMRL_FWD_FP32_SITES: single forward sites with fp32 operands in bf16 mode; the rounded-
  //  operand oracle tests of round 4 replaced that use, the knob went in round 6)
  static constexpr bool fp32_site(int) { return false; }
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
  // epoch-ordered critic pass with the next batch's forward pass beside the update (engine_step.hip)
  bool pipe_primed = false; int fwd_parity = 0, pipe_set = 0;   // pipe_set: the input set whose batch's features the primary forward set holds
  int pipe_forward_body(bool other_inputs);
  int run_stage1_pipe_prime();
  int run_stage1_pipe(bool next_valid);
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
