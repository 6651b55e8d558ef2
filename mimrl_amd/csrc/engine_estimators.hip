// Engine, part of the split of round 5 (see engine.h): grouped MLP stacks, kNN sampler, MI / CMI estimators forward + backward.
#include "engine.h"

// =================================================================================================
// grouped MLP stacks (critic towers, concat-critic tail, CMI classifiers).  Activations are ReLU (Model.py:285).
//   layer l:  A_{l+1} = relu?( A_l W_l^T + b_l ),  A_0 = in, last layer linear.
// `brows` = rows between consecutive groups in the activation buffers (>= rows).
// =================================================================================================
int mimrl_handle::mlp_stack_forward(int nb, int rows, int brows, long p0, long pstride, int nl, const long (*l_off)[2],
                                    const int* dims, const float* in, float* const* act, float* out) {
  // (stacks with thousands of row tiles -- the unfused concat-critic tail -- take the GEMM chain: cfg3 9.80 vs 10.15 ms in round 2)
  if (bf16 && fused_mlp && rows <= 512 && mlp_fused_supported(nb, rows, nl, dims)) {   // one launch (mlp_fused.hip)
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
  constexpr bool no_top1 = false;   // (an environment knob until round 5: fixed at its measured optimum)
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
    constexpr bool no_top_wg = false;   // (an environment knob until round 5: fixed at its measured optimum)
    const bool top_wg = wgrad && !no_top_wg && dims[nl] % 4 != 0 && mlp_bwd_takes_top_wgrad(fa);
    if (top_wg) fa.dw_top = CG(p0 + l_off[nl - 1][0]);
    MX(mlp_stack_bwd_fused(stream, fa));
    if (!wgrad) return MIMRL_OK;
    // the nl weight-gradient GEMMs are independent of each other: on the critical branch (wg_helper >= 0) every second
    // one goes to a helper side stream
    constexpr bool no_split = false;   // (an environment knob until round 5: fixed at its measured optimum)
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
    constexpr bool no_group = false;   // (an environment knob until round 5: fixed at its measured optimum): the round-1 schedule (helper stream, alternating)
    if (!no_group) {
      int lo = 0;
      if (top_wg) lo = 1;                                                             // done inside the data-gradient kernel
      else if (dims[nl] % 4 != 0) { MX(G_on(hs >= 0 ? S(hs) : stream, gs[0])); lo = 1; }   // not row-contiguous-eligible: beside the group
      MX(G_group(stream, gs + lo, nl - lo));
    } else {
      constexpr int helper_par = 0;   // (an environment knob until round 5: fixed at its measured optimum)
      for (int q = top_wg ? 1 : 0; q < nl; ++q) MX(G_on((hs >= 0 && (q & 1) == helper_par) ? S(hs) : stream, gs[q]));
    }
    if (hs >= 0) MX(join(hs, hs));
    return MIMRL_OK;
  }
  constexpr bool no_big_side = false;   // (an environment knob until round 5: fixed at its measured optimum)
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
    int32_t* anc = bufs.anchors + (size_t)knn_slot(stage) * NE_CMI * m;
    int* idx = knn_slot(stage) ? knn_idx2 : knn_idx;
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
    int* idx = knn_slot(stage) ? knn_idx2 : knn_idx;
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
      // sign bitmasks: m1 behind the bf16 a1, m0 and m2 behind the bf16 a0 (in stage 1 the fp32 a2 fills its own buffer)
      fa.m1 = reinterpret_cast<uint32_t*>(ca[1] + half);
      fa.m0 = reinterpret_cast<uint32_t*>(ca[0] + half); fa.m2 = fa.m0 + (size_t)NE_MI * B * B * 8;
      // round 6b: stage 1 does not save a0 when the one-launch weight-gradient kernel will regenerate it from P and Q (the same conditions
      // under which mi_backward routes there: concat_dw on, weights-stationary kernels with the in-kernel dQ reduction)
      {
        const long dq_need = concat_bwd_dq_scratch(NE_MI, B);
        a0_regen_live = fa.save == 2 && knob("MIMRL_NO_CONCAT_DW") == nullptr && knob("MIMRL_NO_CONCAT_DQ") == nullptr &&
                        concat_dw_ok(NE_MI, (long)B * B, HID) && concat_fwd_a2_f16(B, 2) && concat_bwd_ws_supported(B, HID) &&
                        dq_need > 0 && dq_need <= (long)NE_MI * B * B * HID;
        fa.no_a0 = a0_regen_live ? 1 : 0;
      }
      MX(concat_fwd_fused(stream, fa));
    } else {
      concat_compact = false; a0_regen_live = false;
      MX(pair_expand_fwd(stream, cP, cQ, ca[0], NE_MI, B, HID));
      const int dims[4] = {HID, HID, HID, 1};
      MX(mlp_stack_forward(NE_MI, B * B, B * B, tower0, tower_stride, 3, &tower_l[1], dims, ca[0], &ca[1], scores));
    }
  }
  if (has_baseline()) MX(baseline_forward());
  return mi_bound_fwd_bwd(stream, scores, want_grad ? dscores : nullptr, mi_raw, mi_raw + NE_MI, gs_mi(stage), NE_MI, B,
                          cfg.bound_type, stage == 1 ? 0x1fu : 0x07u, has_baseline() ? lbv : nullptr,
                          has_baseline() && want_grad ? dlbv : nullptr, 2L * B, nce_ws);
}

int mimrl_handle::cmi_forward(int stage, bool want_grad) {
  const int B = cfg.batch, n = nprod(), m = m_anchor(), k = cfg.k_neighbor;
  const size_t BD = (size_t)B * EMB;
  const float* cur[5] = {bufs.feats, bufs.feats + BD, bufs.feats + 2 * BD, bufs.feats + 3 * BD, bufs.labels};
  const float* bank[5] = {bufs.bank_f, bufs.bank_t, bufs.bank_a, bufs.bank_v, bufs.bank_c};
  CmiAssembleArgs ca_;
  ca_.anchors = bufs.anchors + (size_t)knn_slot(stage) * NE_CMI * m;
  ca_.idx_x = knn_slot(stage) ? knn_idx2 : knn_idx; ca_.out = cmi_in; ca_.n = n; ca_.m = m; ca_.k = k; ca_.ncall = NE_CMI;
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
    fa.m1 = reinterpret_cast<const uint32_t*>(ca[1] + half);
    fa.m0 = reinterpret_cast<const uint32_t*>(ca[0] + half); fa.m2 = fa.m0 + (size_t)NE_MI * B * B * 8;
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
    // dQ[j] = sum_i dZ0[i, j] inside the kernel (round 5: runs of tiles per workgroup, the partial sums in registers); before, dZ0 went
    // out in fp32 and pair_reduce_q read it back: 2 x 335 MB per pass at cfg3.  MIMRL_NO_CONCAT_DQ=1: that path
    static const bool dq_knob = knob("MIMRL_NO_CONCAT_DQ") == nullptr;   // tuning knob
    const long dq_need = concat_bwd_dq_scratch(NE_MI, B);
    const bool dq_in_kernel = dq_knob && dq_need > 0 && dq_need <= (long)NE_MI * B * B * HID;
    if (dq_in_kernel) { fa.dQ = dQ; fa.dq_part = dca[2]; }   // (the partials live in the dz0 buffer they replace)
    // round 6b: the weights-stationary backward does not write dZ2 when the one-launch weight-gradient kernel will regenerate it
    static const bool dw_on = knob("MIMRL_NO_CONCAT_DW") == nullptr;   // tuning knob
    const bool use_dw = wgrad && dw_on && concat_dw_ok(NE_MI, (long)B * B, HID);
    const bool dz2_regen = use_dw && dq_in_kernel && concat_bwd_ws_supported(B, HID) && !knob_on("MIMRL_CONCAT_STREAMED");
    fa.no_dz2 = dz2_regen ? 1 : 0;
    if (wgrad && a0_regen_live && !dz2_regen) return set_error(MIMRL_ERR_STATE, "mi_backward: the forward pass did not save a0 but the one-launch weight-gradient kernel is off");
    MX(concat_bwd_fused(stream, fa));
    const bool side_wg = wgrad && multi_stream && wg_helper >= 0;
    if (wgrad) {   // dW2 = dZ2^T a1, dW1 = dZ1^T a0: K = B*B rows, split-K with atomics, beside pair_reduce_q on the helper stream
      if (side_wg) MX(fork(wg_helper, wg_helper));
      bool dw3_done = false;
      // round 6: both layers as ONE launch whose workgroups hold a whole 256 x 256 output (concat_dw.hip: every dZ / A row staged once);
      // MIMRL_NO_CONCAT_DW=1: the two split-K GEMMs on 128 x 128 tiles
      if (use_dw) {
        ConcatDwArgs w;
        w.dz[0] = dz2; w.act[0] = reinterpret_cast<const __bf16*>(ca[1]); w.dw[0] = CG(tower0 + tower_l[2][0]);
        w.dz[1] = dz1; w.act[1] = reinterpret_cast<const __bf16*>(ca[0]); w.dw[1] = CG(tower0 + tower_l[1][0]);
        w.nlayer = 2; w.E = NE_MI; w.rows = (long)B * B; w.dw_stride = tower_stride;
        dw3_done = concat_fwd_a2_f16(B, 2);   // the score head's weight gradient rides on the same launch when a2 is the fp16 copy
        if (dw3_done) { w.ds = dscores; w.a2 = reinterpret_cast<const _Float16*>(ca[2]); w.dw3 = CG(tower0 + tower_l[3][0]); }
        if (dz2_regen) { w.ds = dscores; w.m2 = fa.m2; w.w3 = fa.w3; w.dz[0] = nullptr; }   // dZ2 was never written: regenerated from ds, w3, m2
        if (a0_regen_live) { w.P = cP; w.Q = cQ; w.B = B; w.act[1] = nullptr; }             // a0 was never written: regenerated from P, Q
        MX(concat_dw(side_wg ? S(wg_helper) : stream, w));   // (beside the layer-0 weight gradients and bias sums below, which need dP / dQ only)
      } else
      for (int l = 2; l >= 1; --l) {
        GemmDesc g;
        g.A = reinterpret_cast<const float*>(l == 2 ? dz2 : dz1); g.a_bf16 = 1; g.sa_m = 1; g.sa_k = HID; g.sa_b = (long)B * B * HID;
        g.B = l == 2 ? ca[1] : ca[0]; g.b_bf16 = 1; g.sb_k = HID; g.sb_n = 1; g.sb_b = (long)B * B * HID;   // (the bf16 copies)
        g.C = CG(tower0 + tower_l[l][0]); g.sc_m = HID; g.sc_n = 1; g.sc_b = tower_stride;
        g.M = HID; g.N = HID; g.K = B * B; g.batch = NE_MI; g.atomic = 1;
        MX(G_on(side_wg ? S(wg_helper) : stream, g));
      }
      // the score head's weight gradient streams a2 on the main stream, under the products on the helper (one of the two products on the
      // main stream as well: no change, 5.87-5.90 vs 5.87-5.92 ms at cfg3)
      if (!dw3_done) MX(concat_dw3(stream, dscores, ca[2], CG(tower0 + tower_l[3][0]), NE_MI, B, tower_stride, concat_fwd_a2_f16(B, 2)));
    }
    if (!dq_in_kernel) MX(pair_reduce_q(stream, dca[2], dQ, NE_MI, B, HID));
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
  constexpr bool no_head_gather = false;   // (an environment knob until round 5: fixed at its measured optimum)
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
  DetDefer det_defer(stream, &det_ranges);   // (deterministic build, det.h: the stacks' weight / bias gradients are flushed once, at the end of the stage's estimator work)
  const bool bf_fwd = (prec & MIMRL_PREC_BF16_GEMM_FWD) != 0 && !fp32_site(8), bf_bwd = (prec & MIMRL_PREC_BF16_GEMM_BWD) != 0;
  static const bool dbg_skip_imgt = dbg_env("MIMRL_DBG_SKIP_IMGT") != nullptr;   // timing experiments only (stale images: wrong gradients)
  constexpr bool imgt_first = false;         // (an environment knob until round 5: fixed at its measured optimum)
  bool imgT_pending = false;
  imgT_ready = false;
  if (frag_side_pending) { MX(join(3, 3)); frag_side_pending = false; }
  // data-gradient fragment images whose launch enqueue_apply left to this stage (the forward ones came with the critic update): on side 3
  // in front of the transposed-image refresh, which the backward stacks join -- or right here when that refresh does not happen
  auto frag_tr_launch = [&](hipStream_t st) -> int {
    if (!frag_tr_deferred) return MIMRL_OK;
    frag_tr_deferred = false;
    return bf16_frag_images(st, bufs.crit_p, crit_frag, ftab_tr);
  };
  if (backward && bf_bwd && fused_mlp && crit_imgT && ttab.n > 0) {   // transposed weight images for the fused data-gradient chains,
    if (!(skip_imgT_refresh && stage == 1)) {                         // built beside the forward stacks (combined step: once per step,
      MX(fork(3, 3));                                                 // in stage 2 -- stage 1 of the NEXT step sees the same critics)
      // capture order: the launch itself goes BEHIND the CMI branch's forward kernels (see cmi_branch) -- graph nodes are dispatched in
      // capture order, and as the first child of the stage boundary it held up both forward branches by ~18 us (MIMRL_IMGT_FIRST=1)
      imgT_pending = !imgt_first && multi_stream && side_on(3) && side_on(5);
      if (!imgT_pending) MX(frag_tr_launch(S(3)));
      if (!imgT_pending && !dbg_skip_imgt) MX(bf16_transposed_images(S(3), bufs.crit_p, crit_imgT, ttab));
    }
    imgT_ready = true;
  }
  if (!imgT_pending) MX(frag_tr_launch(stream));   // (no refresh on side 3 in this stage: on the chain, as before round 5b)
  static const int dbg_skip = dbg_env("MIMRL_DBG_SKIP_EST") ? atoi(dbg_env("MIMRL_DBG_SKIP_EST")) : 0;   // timing experiments only
  MX(fork(5, 5));
  MX(chain(5, 4));                       // the CMI branch needs the kNN indices
  constexpr int interleave = 0;
  const bool imgT_late = ((interleave >> (stage - 1)) & 1) && interleave >= 4;   // 4 + mask: the image launches on side 3 are captured behind BOTH forward halves
  auto imgT_launch = [&]() -> int {   // side 3 already waits for the stage boundary (fork above); only the launch was held back
    imgT_pending = false;
    MX(frag_tr_launch(side[3]));
    if (!dbg_skip_imgt) MX(bf16_transposed_images(side[3], bufs.crit_p, crit_imgT, ttab));
    return MIMRL_OK;
  };
  // part: 1 = forward, 2 = backward, 3 = both
  auto cmi_branch = [&](int part) -> int {
    if (dbg_skip & 1) return MIMRL_OK;
    StreamGuard g(this, S(5));
    if (part & 1) {
      bf16 = bf_fwd;
      MX(cmi_forward(stage, want_grad));
      if (imgT_pending && !imgT_late) MX(imgT_launch());
      MX(dbg_delay(stream, stage == 1 ? 4 : 14));
    }
    // (no helper side stream for this branch's weight gradients: it runs on side 5, and a fork / join pair hanging off a captured stream
    //  other than the capture's origin sends this HIP runtime's EndCapture into an endless recursion -- tried, core dump)
    if ((part & 2) && backward) { bf16 = bf_bwd; if (imgT_ready && !(skip_imgT_refresh && stage == 1)) MX(chain(5, 3)); MX(cmi_backward(stage)); MX(dbg_delay(stream, stage == 1 ? 6 : 16)); }
    return MIMRL_OK;
  };
  auto mi_branch = [&](int part) -> int {
    if (dbg_skip & 2) return MIMRL_OK;
    if (part & 1) {
      bf16 = bf_fwd;
      { Scope sc(this, MIMRL_PH_EST_FWD); MX(mi_forward(stage, want_grad)); }
      MX(dbg_delay(stream, stage == 1 ? 3 : 15));
    }
    if ((part & 2) && backward) {
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
  constexpr int mi_first = 0;
  // round 5b: both FORWARD halves are captured in front of either backward half (bit mask like mi_first; default: stage 2).  With the stage
  // boundary inside the critic update both branches hang off the update node directly, and with a whole branch captured first the other
  // branch's first kernel landed on a hardware queue BEHIND the first branch's weight-gradient launch (head-of-line: 49 us late).
  if ((interleave >> (stage - 1)) & 1) {
    if ((mi_first >> (stage - 1)) & 1) { MX(mi_branch(1)); MX(cmi_branch(1)); }
    else { MX(cmi_branch(1)); MX(mi_branch(1)); }
    if (imgT_pending) MX(imgT_launch());
    if ((mi_first >> (stage - 1)) & 1) { MX(mi_branch(2)); MX(cmi_branch(2)); }
    else { MX(cmi_branch(2)); MX(mi_branch(2)); }
  } else if ((mi_first >> (stage - 1)) & 1) { MX(mi_branch(3)); MX(cmi_branch(3)); }
  else { MX(cmi_branch(3)); MX(mi_branch(3)); }
  bf16 = bf_fwd;
  if (!multi_stream) return MIMRL_OK;
  return join(5, 5);
}
