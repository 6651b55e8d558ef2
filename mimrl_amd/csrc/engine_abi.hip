// Engine, part of the split of round 5 (see engine.h): the C ABI of include/mimrl.h.
#include "engine.h"

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
int mimrl_dbg_concat_bwd_phases(long long* out) { return mimrl::concat_bwd_read_phases(out); }
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
  h->knn_pre = true;
  h->fold_unpack_on = knob("MIMRL_NO_FOLD_UNPACK") == nullptr;
  h->h16_on = knob("MIMRL_NO_H16") == nullptr;
  h->xin_on = knob("MIMRL_NO_XIN") == nullptr;
  h->fused_cube_bwd = knob("MIMRL_NO_FUSED_CUBE_BWD") == nullptr;
  h->laxis_bwd_long_on = !(knob("MIMRL_LAXIS_BWD_LONG") && atoi(knob("MIMRL_LAXIS_BWD_LONG")) == 0);
  h->laxis_long_on = !(knob("MIMRL_LAXIS_LONG") && atoi(knob("MIMRL_LAXIS_LONG")) == 0);
  h->rec16_on = !(knob("MIMRL_REC16") && atoi(knob("MIMRL_REC16")) == 0);
  h->dwih_h16_on = !(knob("MIMRL_DWIH_H16") && atoi(knob("MIMRL_DWIH_H16")) == 0);
  h->adam_frag_on = !(knob("MIMRL_ADAM_FRAG") && atoi(knob("MIMRL_ADAM_FRAG")) == 0);
  // opt-in (measured slower at cfg2, 0.814-0.819 vs 0.799-0.800 ms: the parked block-0 weight gradients then start together with the layer-1 BPTT
  // instead of 30 us ahead of it, and the BPTT beside them takes 74 instead of 53 us -- DESIGN section 7)
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
  h->l0_bwd_pack = false;
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
  {   // deterministic build (det.h: DetDefer): the tensors whose accumulated sums nobody reads before the end of a gradient pass -- the two
      // gradient buckets (NOT the packed layer-0 scratch: l0_unpack_grads reads it inside the pass)
    h->det_ranges.lo[0] = b->main_g; h->det_ranges.lo[1] = b->crit_g;
    h->det_ranges.bytes[0] = sizeof(float) * (size_t)h->layout.floats[MIMRL_GROUP_MAIN];
    h->det_ranges.bytes[1] = sizeof(float) * (size_t)h->layout.floats[MIMRL_GROUP_CRITIC];
    h->det_ranges.n = 2;       // (per handle, handed to every DetDefer scope this handle opens: no process-global table)
  }
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

int mimrl_stage1_pipe_prime(mimrl_handle* h) { return h ? h->run_stage1_pipe_prime() : set_error(MIMRL_ERR_ARG, "null handle"); }
int mimrl_stage1_pipe(mimrl_handle* h, int next_valid) { return h ? h->run_stage1_pipe(next_valid != 0) : set_error(MIMRL_ERR_ARG, "null handle"); }

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
  h->pipe_primed = false;
  launch_begin_stage(h->stream, h->d_ints, (int*)nullptr, h->bufs.scalars, 32, 32);
  LAUNCH_CHECK();
  h->ev_next = 0;
  MX(h->model_forward(train_mode != 0, false));
  if (!with_losses) return MIMRL_OK;
  launch_mae(h->stream, h->bufs.pred, h->bufs.labels, (float*)nullptr,
                     h->bufs.scalars + MIMRL_S2_TASK, h->cfg.batch);
  LAUNCH_CHECK();
  h->ev_next = 0;
  if (h->bank_rows > 0) { MX(h->fork(4, 4)); MX(h->knn_launch(2, h->S(4))); MX(h->estimators_all(2, false, false)); }
  launch_finalize_stage2(h->stream, h->bufs.scalars, h->mi_raw, h->cmi_raw,
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
    launch_finalize_stage1(h->stream, h->bufs.scalars, h->mi_raw, h->cmi_raw,
                       h->bce_raw, h->coef1());
  } else {
    launch_mae(h->stream, h->bufs.pred, h->bufs.labels, (float*)nullptr,
                       h->bufs.scalars + MIMRL_S2_TASK, h->cfg.batch);
    launch_finalize_stage2(h->stream, h->bufs.scalars, h->mi_raw, h->cmi_raw,
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

int mimrl_set_comm_critic_bf16(mimrl_handle* h, int on) {
  if (!h) return set_error(MIMRL_ERR_ARG, "null handle");
  if ((on != 0) == h->comm_crit_bf16) return MIMRL_OK;
  HIPX(hipStreamSynchronize(h->user_stream));
  h->drop_graphs();
  if (on && !h->comm_crit16) HIPX(hipMalloc(&h->comm_crit16, sizeof(uint16_t) * (size_t)h->layout.floats[MIMRL_GROUP_CRITIC]));
  h->comm_crit_bf16 = on != 0;
  if (on && h->comm && h->bound) {   // RCCL's set-up for the new datatype outside any capture
    HIPX(hipMemsetAsync(h->comm_crit16, 0, 16, h->user_stream));
    MX(mimrl::comm_allreduce_sum_bf16(h->comm, h->comm_crit16, 8, h->user_stream));
    HIPX(hipStreamSynchronize(h->user_stream));
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
  if (h->comm_crit16) (void)hipFree(h->comm_crit16);
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

int mimrl_op_concat_dw(void* stream, const void* dz2, const void* a1, float* dw2, const void* dz1, const void* a0, float* dw1, int E, int64_t rows,
                       int64_t dw_stride, const float* ds, const void* a2h, float* dw3, const uint32_t* m2, const float* w3, const float* P,
                       const float* Q, int B) {
  if ((!dz2 && !m2) || !a1 || !dw2) return set_error(MIMRL_ERR_ARG, "mimrl_op_concat_dw: null argument");
  if ((dz1 != nullptr) != (a0 != nullptr || P != nullptr) || (dz1 != nullptr) != (dw1 != nullptr)) return set_error(MIMRL_ERR_ARG, "mimrl_op_concat_dw: the second layer comes as a set");
  if (!concat_dw_ok(E, rows, 256)) return set_error(MIMRL_ERR_ARG, "mimrl_op_concat_dw: E >= 1, rows >= 1");
  ConcatDwArgs a;
  a.dz[0] = static_cast<const __bf16*>(dz2); a.act[0] = static_cast<const __bf16*>(a1); a.dw[0] = dw2;
  a.dz[1] = static_cast<const __bf16*>(dz1); a.act[1] = static_cast<const __bf16*>(a0); a.dw[1] = dw1;
  a.nlayer = dz1 ? 2 : 1; a.E = E; a.rows = rows; a.dw_stride = dw_stride;
  a.ds = ds; a.a2 = static_cast<const _Float16*>(a2h); a.dw3 = dw3; a.m2 = m2; a.w3 = w3; a.P = P; a.Q = Q; a.B = B;
  return concat_dw(reinterpret_cast<hipStream_t>(stream), a);
}

int mimrl_op_gru_wgrad(void* stream, const void* const* dg, const void* const* x, const void* const* hp, float* const* dw_ih,
                       float* const* dw_hh, int64_t rows, int kp) {
  if (!dg || !x || !hp || !dw_ih || !dw_hh) return set_error(MIMRL_ERR_ARG, "mimrl_op_gru_wgrad: null argument");
  GruWgradArgs a;
  for (int s = 0; s < 4; ++s) {
    if (!dg[s] || !x[s] || !hp[s] || !dw_ih[s] || !dw_hh[s]) return set_error(MIMRL_ERR_ARG, "mimrl_op_gru_wgrad: null array for sequence %d", s);
    a.seq[s] = GruWgradSeq{static_cast<const __bf16*>(dg[s]), static_cast<const __bf16*>(x[s]), static_cast<const __bf16*>(hp[s]), dw_ih[s], dw_hh[s]};
  }
  a.rows = rows; a.kp = kp;
  return gru_wgrad(reinterpret_cast<hipStream_t>(stream), a);
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
