// Engine, part of the split of round 5 (see engine.h): model forward: encoders (GRU / LSTM / conv), forward tail, CubeMLP.
#include "engine.h"

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
  // round 5b: with the fp16 copy feeding the layer-1 projection, dW_ih of layer 1 AND the layer-0 BPTT (h_prev), nobody reads the fp32 layer-0
  // outputs of a training pass any more -- the fused-projection forward then does not write them (262 MB per pass at cfg3).  The two flags
  // describe the pass whose activations the backward will read, i.e. they are recorded by SAVING passes only: in the non-shared prefetch mode
  // (MIMRL_NO_SHARED_PREFIX) stage 1's own, non-saving forward pass is captured behind the saving stage-2 pass and must not overwrite them.
  if (save) { h16_live = use_h16; hp16_live = false; }   // (hp16_live: set below, once the layer-0 kernel is chosen)
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
      const bool hp16_now = use_h16 && rec16_on && save && dg_bf16 && (prec & MIMRL_PREC_BF16_GRU_BWD) != 0 && gru_bwd_io16_ok(l0_xin ? 2 : 0);
      if (save) hp16_live = hp16_now;
      if (l0_xin) {
        a.xin_on = 1; a.kp = KP();
        a.no_out32 = hp16_now && dwih_h16_on ? 1 : 0;   // every consumer of this pass's layer-0 outputs reads the fp16 copy
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
  if (part != 1 && T < L && !pre_done) HIPX(hipMemsetAsync(cube0, 0, sizeof(float) * (size_t)B * L * 3 * D, stream));
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
    constexpr bool prefix_split = true;   // (an environment knob until round 5: fixed at its measured optimum)
    if (prefix_split && cfg.encoder == MIMRL_ENCODER_GRU && side_on(0)) {
      // lengths (Model.py:425-432): only the recurrence needs them.  Side 0 has slack (the text projection is needed at the tail);
      // on side 4 the scan sat in front of the video input projection, the longest chain ahead of the layer-0 recurrence
      MX(seq_lengths2(S(0), bufs.audio, cfg.d_a, lens[0], bufs.video, cfg.d_v, lens[1], B, T));
      MX(next_event(&ev_lens));
      HIPX(hipEventRecord(ev_lens, S(0)));
    }
    constexpr int text_late = 0;   // (an environment knob until round 5: fixed at its measured optimum)
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
  } else if (!fused_pre && !pre_done) {
    MX(text_post_fwd(stream, tx_raw, cube0, B, T, L, 3, D, 0, pdrop[0], key(), 0));
  }
  // text dropout -> cube slot 0; fwd+bwd sum, LN, ReLU, dropout (Model.py:452-461) -> cube slots 1,2; T_F, A_F, V_F (Model.py:466)
  if (!pre_done) {
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

// The pre-CubeMLP pieces (text dropout, LN + ReLU + dropout of the encoder outputs, temporal means: Model.py:452-466) of BOTH forward tails
// of a shared-prefix training step in one launch on `stream`: the primary set (stage 2's tail, dropout key of the step counter
// begin_stage(2) will set: +1) and the alternate set (stage 1's tail, the current counter).  Long sequences only: below 129 steps each
// tail has its own one-launch tail_pre_fwd, which is latency and runs beside the other tail's.
int mimrl_handle::dual_tail_pre() {
  const int B = cfg.batch, T = cfg.seq_len, L = cfg.time_len, D = cfg.d_common;
  float* const cubes[2] = {cube0, alt.cube0};
  float* const feats[2] = {bufs.feats + (size_t)B * D, alt.feats + (size_t)B * D};
  const int add[2] = {1, 0};
  if (T < L)
    for (int o = 0; o < 2; ++o) HIPX(hipMemsetAsync(cubes[o], 0, sizeof(float) * (size_t)B * L * 3 * D, stream));
  LnSide2 sd[2];
  for (int m = 0; m < 2; ++m)
    sd[m] = LnSide2{h1[m], P(ln_g[m]), P(ln_b[m]), ln_mean[m], ln_rstd[m], nullptr, nullptr, nullptr, 1 + m, cfg.dropout[1 + m], (uint32_t)(1 + m)};
  MX(tail_pre2_fwd(stream, tx_raw, cfg.dropout[0], sd[0], sd[1], cubes, feats, add, tailp_part, B, T, L, 3, D, key()));
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
      // round 5b, long sequences (the fused block kernel above needs L <= 64): one pass over the [L, C] slab of every sample instead of two
      // GEMM launches that each read it, a padded-copy launch and the LayerNorm launch (cfg3: 275 -> ~80 us per forward tail)
      const bool long_l = bf16 && laxis_long_on && !cfg.ln_first && pl <= 0.f && a.res >= 0 && laxis_fwd_long_supported(il, hl, ol, (int)C);
      if (long_l) {
        if (w2p[i] && hl % 4 != 0) {   // (the padded fc2 copy is still what the backward's dU product reads)
          MX(pad_rows(stream, P(a.fc2.w), w2p[i], ol, hl, (hl + 3) & ~3));
          w2p_valid[i] = true;
        }
        LAxisLongArgs la;
        la.x = x; la.w1 = P(a.fc1.w); la.b1 = a.fc1.b >= 0 ? P(a.fc1.b) : nullptr; la.w2 = P(a.fc2.w); la.b2 = a.fc2.b >= 0 ? P(a.fc2.b) : nullptr;
        la.wr = P(a.res); la.g = P(a.ln_g); la.be = P(a.ln_b);
        la.u = b.l.u; la.h = b.l.h; la.y = b.l.y; la.z = b.l.z; la.mean = b.l.mean; la.rstd = b.l.rstd;
        la.B = B; la.il = il; la.hl = hl; la.ol = ol; la.C = (int)C; la.act = cfg.activation;
        MX(laxis_fwd_long(stream, la, fwd_f16));
      } else {
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
