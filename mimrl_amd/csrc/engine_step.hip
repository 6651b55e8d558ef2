// Engine, part of the split of round 5 (see engine.h): gradient passes, updates, the data-parallel reduce, graph capture, run / run_step.
#include "engine.h"

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
    constexpr bool prefix_split = false;   // (an environment knob until round 5: fixed at its measured optimum)
    const bool share = prefetch && !no_share;
    // counters + scalar reset: the first consumers are the kNN sampler and the recurrence, both behind the join of side 4 in
    // encoders_forward -- in the shared-prefix step it runs on side 4 beside the input projections instead of in front of them
    const bool begin_on_side = share && have_banks && skip_zero && prefix_split && multi_stream && cfg.encoder == MIMRL_ENCODER_GRU;
    if (begin_on_side) MX(fork(4, 4));
    // shared-prefix step with packed layer-0 operands: the pack launch is the first kernel of the prefix on this stream and nothing in
    // front of the recurrence reads the counters or the scalars -- the bookkeeping rides on it (one launch + one gap less on the chain)
    // (measured neutral, 0.970 vs 0.966 ms: the single-thread kernel hides in the gap between two graph launches -- opt-in)
    // (round 4, with the length scan on side 0: -4 us on average over four alternating runs, cfg3 neutral -- on by default; =0: the separate kernel)
    constexpr bool want_begin_in_pack = true;   // (an environment knob until round 5: fixed at its measured optimum)
    begin_in_pack = want_begin_in_pack && share && have_banks && skip_zero && !begin_on_side && l0_packed && cfg.encoder == MIMRL_ENCODER_GRU;
    if (!begin_in_pack) {
      launch_begin_stage(begin_on_side ? side[4] : stream, d_ints, have_banks ? d_ints + 2 : nullptr,
                         bufs.scalars, 0, 32);
      LAUNCH_CHECK();
    }
    if (!have_banks) return MIMRL_OK;
    if (!skip_zero) HIPX(hipMemsetAsync(bufs.crit_g, 0, sizeof(float) * layout.floats[MIMRL_GROUP_CRITIC], stream));   // epoch-0 rule: zero loss, no update (Customization.py:97-98, Solver.py:201-203)
    bf16 = (prec & MIMRL_PREC_BF16_GEMM_FWD) != 0;
    constexpr bool pre_first = false;   // (an environment knob until round 5: fixed at its measured optimum): capture order of the two chains
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
      // (1b) long sequences: the pre-CubeMLP pieces of both tails in one launch (each source row read once)
      static const bool dual_pre_on = knob("MIMRL_NO_DUAL_TAIL_PRE") == nullptr;   // tuning knob
      const bool dual_pre = dual_pre_on && !defer_tail && cfg.seq_len > 128 && cfg.d_common == 128 && rng_add == 0;
      if (dual_pre) { MX(dual_tail_pre()); pre_done = true; }
      struct PreDone { bool& f; ~PreDone() { f = false; } } pre_done_guard{pre_done};
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
      launch_finalize_stage1(stream, bufs.scalars, mi_raw, cmi_raw, bce_raw, coef1());
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
    if (!boundary_in_adam)   // (else: workgroup 0 of the critic Adam launch did it -- enqueue_apply)
      launch_stage_boundary(stream, bufs.scalars, mi_raw, cmi_raw, bce_raw, coef1(), d_ints, d_ints + 1,
                         bufs.pred, bufs.labels, dpred, B);
    boundary_in_adam = false;
  } else {
    launch_begin_stage(stream, d_ints, d_ints + 1, bufs.scalars, 32, 32);
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
    launch_mae(stream, bufs.pred, bufs.labels, dpred, bufs.scalars + MIMRL_S2_TASK, B);
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
  launch_finalize_stage2(S(0), bufs.scalars, mi_raw, cmi_raw, coef2(),
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
  // Combined step (round 5b): the critic update writes the FORWARD fragment images itself and does the stage boundary in its workgroup 0
  // (AdamArgs::frag / ::sb) -- what used to sit between the update and the stage-2 estimators (frag_images 10 us on side 3 beside a 5 us
  // boundary launch, two dependent-launch gaps) is gone from the chain; the data-gradient fragment images, first read by the stage-2
  // backward stacks, stay a launch on side 3 that those stacks join through the transposed-image refresh.  MIMRL_ADAM_FRAG=0: the round-5a sequence.
  boundary_in_adam = false;
  const bool have_frag = stage == 1 && img_valid && crit_frag && ftab.n > 0;
  bool fwd_in_adam = false;
  if (stage == 1 && fuse_boundary && adam_frag_on && a.n % 8 == 0) {
    if (have_frag && a.n < (1L << 24)) {   // (the kernel's image index arithmetic is exact for buckets below 2^24 elements; larger: the frag_images launch)
      for (int e = 0; e < ftab.n && a.frag.n < 12; ++e)
        if (!ftab.tr[e] && ftab.dshift[e] == 0) {
          const int q = a.frag.n++;
          a.frag.lo[q] = ftab.off[e]; a.frag.gstride[q] = ftab.gstride[e]; a.frag.nb[q] = ftab.nb[e];
          a.frag.mat[q] = ftab.OUT[e] * ftab.RED[e]; a.frag.RED[q] = ftab.RED[e];
          a.frag.inv_gs[q] = 1.f / (float)ftab.gstride[e]; a.frag.inv_red[q] = 1.f / (float)ftab.RED[e];
        }
      int nfwd = 0;
      for (int e = 0; e < ftab.n; ++e) nfwd += !ftab.tr[e];
      if (a.frag.n == nfwd) { a.frag.dst = crit_frag; fwd_in_adam = true; } else a.frag.n = 0;
    }
    a.sb_on = 1;
    a.sb = StageBoundaryArgs{bufs.scalars, mi_raw, cmi_raw, bce_raw, coef1(), d_ints, d_ints + 1, bufs.pred, bufs.labels, dpred, cfg.batch};
    boundary_in_adam = true;
  }
  MX(adam_step(stream, a));
  if (have_frag) {
    // combined step: beside the stage boundary on side 3 (the stage-2 estimators join it before their first stack)
    constexpr bool inline_frag = false;   // (an environment knob until round 5: fixed at its measured optimum)
    FragTable ft = ftab;
    if (fwd_in_adam) {   // only the data-gradient entries are left for the launch
      ft.n = 0;
      for (int e = 0; e < ftab.n; ++e)
        if (ftab.tr[e]) {
          const int q = ft.n++;
          ft.off[q] = ftab.off[e]; ft.OUT[q] = ftab.OUT[e]; ft.RED[q] = ftab.RED[e]; ft.nb[q] = ftab.nb[e]; ft.tr[q] = 1;
          ft.gstride[q] = ftab.gstride[e]; ft.dshift[q] = ftab.dshift[e];
        }
    }
    if (ft.n > 0) {
      // (fwd_in_adam: not even the launch happens here -- as the update's FIRST captured child it would keep the update's hardware queue and
      //  push the stage-2 estimators' first kernels onto other queues, 21 us behind the update; estimators_all issues it on side 3 in front
      //  of the transposed-image refresh, in capture order behind the forward stacks)
      if (fwd_in_adam && fuse_boundary && side_on(3) && !inline_frag) { ftab_tr = ft; frag_tr_deferred = true; }
      else if (fuse_boundary && side_on(3) && !inline_frag) { MX(fork(3, 3)); MX(bf16_frag_images(side[3], bufs.crit_p, crit_frag, ft)); frag_side_pending = true; }
      else MX(bf16_frag_images(stream, bufs.crit_p, crit_frag, ft));
    }
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
  pipe_primed = false; knn_flip = 0;                    // (this call's forward pass overwrites the primary forward set)
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

// Epoch-ordered critic pass (round 6; reference: Solver.py:200-216 runs `stage1_n` full passes of critic updates over the loader with the main
// model frozen, THEN one model pass).  In such a pass Model.forward of batch i + 1 depends on nothing the critic update on batch i changes,
// so one call = the estimators + clip + Adam of the bound batch, whose forward pass a previous call (or mimrl_stage1_pipe_prime) already left
// in the primary forward set, with the forward pass of the NEXT batch -- the other input set -- beside them on `pre_stream`, into the
// alternate forward set.  Afterwards the two forward sets trade places (host pointers; graphs are cached per (input set, forward-set parity)).
// Arithmetic, dropout keys and anchor draws are those of the sequential order: the look-ahead pass uses the key the next call's
// begin-of-stage will set (rng_add = 1), exactly like the stage-2 forward prefetch of a two-stage step.
int mimrl_handle::pipe_forward_body(bool other_inputs) {
  // forward pass (no saved activations: stage 1 trains the critics only) of the bound batch, or of the OTHER input set's batch into the
  // alternate forward set
  const void* keep_in[3] = {bufs.text, bufs.audio, bufs.video};
  if (other_inputs) {
    const GraphSet& o = gsets[1 - cur_set];
    bufs.text = static_cast<const float*>(o.in[0]); bufs.audio = static_cast<const float*>(o.in[1]); bufs.video = static_cast<const float*>(o.in[2]);
    swap_fwd_set();
  }
  const bool ms = multi_stream;
  if (other_inputs) multi_stream = false;      // one sequential branch on pre_stream (a fork hanging off a non-origin captured stream breaks EndCapture)
  rng_add = 1;                                 // the begin-of-stage that belongs to this batch has not run yet
  bf16 = (prec & MIMRL_PREC_BF16_GEMM_FWD) != 0;
  const int r = model_forward(true, false, 0);
  rng_add = 0; multi_stream = ms;
  if (other_inputs) {
    swap_fwd_set();
    bufs.text = static_cast<const float*>(keep_in[0]); bufs.audio = static_cast<const float*>(keep_in[1]); bufs.video = static_cast<const float*>(keep_in[2]);
  }
  return r;
}

int mimrl_handle::run_stage1_pipe_prime() {
  if (!bound) return set_error(MIMRL_ERR_STATE, "mimrl_bind must be called before any step");
  if (bank_rows <= 0) return set_error(MIMRL_ERR_STATE, "mimrl_stage1_pipe_prime: stage 1 needs the feature banks (epoch-0 rule)");
  if (prefetch || comm) return set_error(MIMRL_ERR_STATE, "mimrl_stage1_pipe_prime: not in stage-2 prefetch mode / with a communicator");
  MX(ensure_images());
  if (!keep_events) ev_next = 0;
  if (knn_ovr_mask[0] != 0u) return set_error(MIMRL_ERR_STATE, "mimrl_stage1_pipe_prime: caller-supplied neighbour rows (exact tie order) need the sequential pass");
  rng_add = 1;                                 // (anchor draws of the call that follows: the key its begin-of-stage will set)
  { const int rk = knn_launch(1, stream); rng_add = 0; MX(rk); }
  MX(pipe_forward_body(false));
  pipe_primed = true; pipe_set = cur_set;
  return MIMRL_OK;
}

int mimrl_handle::run_stage1_pipe(bool next_valid) {
  if (!bound) return set_error(MIMRL_ERR_STATE, "mimrl_bind must be called before any step");
  if (bank_rows <= 0) return set_error(MIMRL_ERR_STATE, "mimrl_stage1_pipe: stage 1 needs the feature banks (epoch-0 rule)");
  if (prefetch || comm) return set_error(MIMRL_ERR_STATE, "mimrl_stage1_pipe: not in stage-2 prefetch mode / with a communicator");
  if (!pipe_primed || pipe_set != cur_set)
    return set_error(MIMRL_ERR_STATE, "mimrl_stage1_pipe: the bound batch has no forward pass yet (mimrl_stage1_pipe_prime, or the look-ahead pass of the last call belongs to the other input set)");
  if (next_valid && !gsets[1 - cur_set].in[0]) return set_error(MIMRL_ERR_STATE, "mimrl_stage1_pipe: the other input set was never bound (mimrl_set_inputs)");
  if (!pre_stream) HIPX(hipStreamCreateWithFlags(&pre_stream, hipStreamNonBlocking));
  MX(ensure_images());
  imgT_valid = false;                          // a critic update outside the combined step (as run(1, 0))
  part0_done = false;
  if (!grads_clean[1]) HIPX(hipMemsetAsync(bufs.crit_g, 0, sizeof(float) * layout.floats[MIMRL_GROUP_CRITIC], stream));
  grads_clean[1] = true;
  auto body = [&]() -> int {
    if (!keep_events) ev_next = 0;
    launch_begin_stage(stream, d_ints, d_ints + 2, bufs.scalars, 0, 32);
    LAUNCH_CHECK();
    bf16 = (prec & MIMRL_PREC_BF16_GEMM_FWD) != 0;
    hipEvent_t e_begin = nullptr;
    MX(next_event(&e_begin));
    HIPX(hipEventRecord(e_begin, stream));
    // TWO branches behind the begin-of-stage node: (a) on pre_stream the NEXT call's kNN sampler (banks only; into the other anchor /
    // neighbour-index set, with the key the next begin-of-stage will set) followed by the next batch's forward pass -- one sequential chain,
    // captured first (graph nodes are dispatched in capture order); (b) this call's estimators, whose sampler the previous call (or the prime
    // call) already ran.  As a third branch -- or at the head of (b) -- the sampler's merge kernel ran 134-151 us beside the forward pass's
    // kernels instead of 18 and held the estimators back (gpurun_out/epoch_tl_pipe_stage1.txt of the first two cuts).
    if (next_valid) {
      HIPX(hipStreamWaitEvent(pre_stream, e_begin, 0));
      StreamGuard g(this, pre_stream);
      knn_flip ^= 1; rng_add = 1;
      const int rk = knn_launch(1, stream);
      knn_flip ^= 1; rng_add = 0;
      MX(rk);
      MX(pipe_forward_body(true));
    }
    const unsigned keep_mask = side_mask;
    side_mask &= ~(1u << 4);                   // (nothing on side 4 in this call: the CMI branch's chain(5, 4) then orders it behind the main stream)
    { const int re = estimators_all(1, true, true); side_mask = keep_mask; MX(re); }
    launch_finalize_stage1(stream, bufs.scalars, mi_raw, cmi_raw, bce_raw, coef1());
    LAUNCH_CHECK();
    if (next_valid) {                          // rejoin before the stage ends (a captured graph must not leave a dangling branch)
      hipEvent_t e;
      MX(next_event(&e));
      HIPX(hipEventRecord(e, pre_stream));
      HIPX(hipStreamWaitEvent(stream, e, 0));
    }
    return enqueue_apply(1);
  };
  int r;
  if (!cfg.use_graph || prof_on) r = body();
  else {
    // (the forward-set roles are baked into the kernel arguments: one graph per parity -- a pass with an even number of batches flips the
    //  relation between input set and forward set for the next pass; the bank size is baked in as well)
    hipGraphExec_t& ex = GS().pipe[next_valid ? 1 : 0][fwd_parity * 2 + knn_flip];
    int& tag = GS().pipe_rows[next_valid ? 1 : 0][fwd_parity * 2 + knn_flip];
    const int want = bank_rows;
    if (ex && tag != want) retire(ex);
    if (!ex) {
      hipGraph_t g = nullptr;
      if (!cap_stream) HIPX(hipStreamCreateWithFlags(&cap_stream, hipStreamNonBlocking));
      HIPX(hipStreamBeginCapture(cap_stream, hipStreamCaptureModeThreadLocal));
      stream = cap_stream;
      const int rb = body();
      stream = user_stream;
      const hipError_t ce = hipStreamEndCapture(cap_stream, &g);
      if (rb != 0) { if (g) (void)hipGraphDestroy(g); return rb; }
      if (ce != hipSuccess) return set_error(MIMRL_ERR_HIP, "hipStreamEndCapture: %s", hipGetErrorString(ce));
      const hipError_t ie = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
      (void)hipGraphDestroy(g);
      if (ie != hipSuccess) { ex = nullptr; return set_error(MIMRL_ERR_HIP, "hipGraphInstantiate: %s", hipGetErrorString(ie)); }
      tag = want;
    }
    HIPX(hipGraphLaunch(ex, stream));
    r = MIMRL_OK;
  }
  MX(r);
  if (next_valid) { swap_fwd_set(); fwd_parity ^= 1; knn_flip ^= 1; }   // the look-ahead pass's sets are the current ones of the next call
  else knn_flip = 0;                           // end of the pass: stage 1 / stage 2 back on their own sets
  pipe_primed = next_valid; pipe_set = 1 - cur_set;
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

// Captured-graph diagnostics: MIMRL_GRAPH_DOT=<file> dumps the captured two-stage step (hipGraphDebugDotPrint: kernel names, edges).
// (Rounds 4-5 also re-inserted fork edges here -- by height, captured-stream-first, pad nodes, a searched permutation: the captured order
// was a local optimum of all of them; the machinery went in round 6, git keeps it.)
static int graph_postprocess(hipGraph_t g) {
  static const char* dot = knob("MIMRL_GRAPH_DOT");
  if (dot) HIPX(hipGraphDebugDotPrint(g, dot, hipGraphDebugDotFlagsVerbose));
  return MIMRL_OK;
}

// Solver.step(): stage 1 then stage 2 on the bound batch.  In overlap mode with graphs the two stages are ONE captured
// graph (one launch, no idle device between the stage-1 Adam and the stage-2 estimators); otherwise two run() calls.
int mimrl_handle::run_step() {
  Range rg("mimrl.two_stage_step (Solver.step)");
  if (!bound) return set_error(MIMRL_ERR_STATE, "mimrl_bind must be called before any step");
  pipe_primed = false; knn_flip = 0;
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
    constexpr bool no_boundary = false;   // (an environment knob until round 5: fixed at its measured optimum): the round-1 stage boundary
    hipGraph_t g = nullptr;
    if (!cap_stream) HIPX(hipStreamCreateWithFlags(&cap_stream, hipStreamNonBlocking));
    HIPX(hipStreamBeginCapture(cap_stream, hipStreamCaptureModeThreadLocal));
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
    { const int pr = graph_postprocess(g); if (pr != 0) { (void)hipGraphDestroy(g); return pr; } }
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
