// Engine, part of the split of round 5 (see engine.h): parameter resolution, workspace arena, weight-image tables.
#include "engine.h"


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
  MX(take(&nce_ws, NCE_WS_FLOATS));   // ticket + slots of mi_infonce_rows_kernel (the arena is zeroed once; the kernel resets its ticket)
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
  {   // chunk sums of the two-tail pre-CubeMLP launch (model_ops.hip: tail_pre2_kernel); one chunk = none needed
    int nchunk = 1, rpc = 0;
    tail_pre2_chunks(cfg.batch, cfg.seq_len, &nchunk, &rpc);
    if (nchunk > 1) MX(take(&tailp_part, 2 * 3 * B * (size_t)nchunk * D));
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
