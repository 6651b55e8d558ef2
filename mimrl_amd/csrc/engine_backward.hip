// Engine, part of the split of round 5 (see engine.h): model backward: CubeMLP, LayerNorm, BPTT and its weight gradients.
#include "engine.h"

int mimrl_handle::cube_backward(int cur_in, int* cur_out) {
  const int B = cfg.batch;
  int dims[MIMRL_MAX_BLOCKS + 1][3];
  dims[0][0] = cfg.time_len; dims[0][1] = 3; dims[0][2] = cfg.d_common;
  for (int i = 0; i < cfg.n_blocks; ++i)
    for (int ax = 0; ax < 3; ++ax) dims[i + 1][ax] = cfg.d_outs[i][ax];
  // deferred mode: every gradient buffer is used once (weight-gradient GEMMs read dY / dU after the chain has moved on)
  constexpr bool no_defer = false;   // (an environment knob until round 5: fixed at its measured optimum)
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
      constexpr bool pg_fuse = false;        // (an environment knob until round 5: fixed at its measured optimum)
      constexpr bool no_pg_one = false;
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
    if (bf16 && fused_cube_bwd && !cfg.ln_first && pl <= 0.f && w.ax[0].res >= 0 && laxis_bwd_supported(il, hl, ol, ik * id) &&
        (il <= 64 || laxis_bwd_long_on)) {   // (il > 64: the LONG instantiation, round 5b; MIMRL_LAXIS_BWD_LONG=0: the GEMM chain below)
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
  // MIMRL_DBG_DELAY_TAG=<tag>[:<microseconds>] (default 50 us)
  static const char* spec = knob("MIMRL_DBG_DELAY_TAG");
  static const int want = spec ? atoi(spec) : -1;
  static const int us = spec && strchr(spec, ':') ? atoi(strchr(spec, ':') + 1) : 50;
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
  constexpr int wg_sides = 3;
  static const int dbg_skip_kinds = dbg_env("MIMRL_DBG_SKIP_DEFERRED") ? atoi(dbg_env("MIMRL_DBG_SKIP_DEFERRED")) : 0;   // timing experiments only (bit = kind)
  // the weight-gradient GEMMs as (at most) two grouped split-K launches, one per operand-layout class: D-axis products are
  // (RC,RC), the batch-reduced L-axis products (KC,KC).  Alone each is a ~20 us launch of 4..64 tiles.
  constexpr bool no_wg_groupk = false;   // (an environment knob until round 5: fixed at its measured optimum)
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
  DetDefer det_defer(stream, &det_ranges);   // deterministic build: one flush of the accumulation table at the end of the pass for everything that only feeds the gradient buckets
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
  constexpr bool text_bwd_first = false;
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
    // round 5b: ds as bf16 when the layer-1 BPTT launch can read it that way (GRU, bf16 BPTT with bf16 dg, 4-wave kernel)
    ds_bf16_live = rec16_on && cfg.encoder == MIMRL_ENCODER_GRU && dg_bf16 && (prec & MIMRL_PREC_BF16_GRU_BWD) != 0 && gru_bwd_io16_ok(0);
    MX(ln_relu_drop_bwd2(stream, sd[0], sd[1], dcube, B, T, L, 3, D, key(), dmean + (size_t)B * D, dmean + 2 * (size_t)B * D, ds_bf16_live ? 1 : 0));
  }
  if (!text_bwd_first) MX(text_bwd());
  // CubeMLP weight gradients: side 1..3, beside the layer-1 BPTT.  Tuning knob MIMRL_BPTT_FIRST=1 captures the BPTT launch in
  // front of the parked kernels (graph nodes are dispatched in capture order).  Measured on cfg2: 1.58 vs 1.36 ms -- the
  // recurrence is latency-bound and loses more to the weight-gradient kernels sharing its CUs from the first cell step on
  // than the ~100 us it waits behind their first wave; default off.
  // debugging: make the main stream wait for sides 1..3 (the parked kernels) at point n: 1 before the BPTT, 2 behind the layer-1 BPTT,
  // 3 behind the dh0 product, 4 behind the layer-0 BPTT
  static const int dbg_join_at = dbg_env("MIMRL_DBG_JOIN_AT") ? atoi(dbg_env("MIMRL_DBG_JOIN_AT")) : 0;
  constexpr bool bptt_first = false;
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
    const bool io16 = lbf && (prec & MIMRL_PREC_BF16_GRU_BWD) != 0 && gru_bwd_io16_ok(a.slab_upl);
    a.dout_bf16 = io16 && (l == 1 ? ds_bf16_live : dh0_bf16_live) ? 1 : 0;
    if ((l == 1 ? ds_bf16_live : dh0_bf16_live) && !a.dout_bf16) return set_error(MIMRL_ERR_STATE, "gru_layer_backward: dout was stored as bf16 but this BPTT launch cannot read it");
    const bool hp16 = l == 0 && io16 && hp16_live && h0h[0] && h0h[1];
    if (l == 0 && hp16_live && !hp16) return set_error(MIMRL_ERR_STATE, "gru_layer_backward: the forward pass counted on the fp16 h_prev path");
    for (int m = 0; m < 2; ++m) {
      a.lens[m] = lens[m];
      for (int d = 0; d < 2; ++d) {
        const GruDirW& g = gru[m][l][d];
        a.seq[m][d] = GruSeqBwd{P(g.w_hh), sv[l][m][d], l == 1 ? h1[m] : h0[m], l == 1 ? ds[m] : dh0[m], dg[l][m][d],
                                hprev[l][m][d], Gm(g.b_ih), Gm(g.b_hh)};
        if (hp16) a.seq[m][d].out16 = h0h[m];
      }
    }
    { Scope sc(this, MIMRL_PH_GRU_BWD); MX(gru_backward(stream, a, (prec & MIMRL_PREC_BF16_GRU_BWD) != 0)); }
    if ((l == 1 && dbg_join_at == 2) || (l == 0 && dbg_join_at == 4)) MX(join(1, 3));
    if (l == 1 && ev_pre) MX(flush_deferred(0, ev_pre));
    MX(dbg_delay(stream, 8));
    // side streams of the GRU weight gradients (tuning knobs).  Sides 1..3 still carry the parked CubeMLP parameter-gradient
    // kernels at this point; sides 0 (text branch), 4 and 5 (kNN sampler, CMI branch) have been idle since the forward pass.
    constexpr int l0_side = 1;
    constexpr int l1_side0 = 1;
    MX(fork(1, (l == 0 || l1_side0 == 4) ? 5 : 3));   // the weight gradients below depend on the BPTT only
    if (l0_side == 0 && l == 0) MX(fork(0, 0));
    constexpr bool dh0_last = false;   // (an environment knob until round 5: fixed at its measured optimum): capture order of dh0 vs the side-stream weight gradients
    auto dh0_gemm = [&]() -> int {   // gradient to the layer-0 outputs, dh0 = sum_dir dgx_dir . W_ih_l1_dir
      dh0_bf16_live = false;
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
            if (gemm_tall_ok(t)) {
              q = t;
              // round 5b: the tall kernel stores dh0 as bf16 when the layer-0 BPTT launch can read it that way (same element indices)
              if (rec16_on && (prec & MIMRL_PREC_BF16_GRU_BWD) != 0 && (l0_packed || l0_bwd_pack) && gru_bwd_io16_ok(l0_xin ? 2 : 0)) { q.c_bf16 = 1; q.sc_b *= 2; dh0_bf16_live = true; }
            }
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
      // round 6: ONE pass over dg for both products of all four sequences (gru_wgrad.hip) when every operand is stored as bf16;
      // MIMRL_NO_GRU_WGRAD=1: the two batched split-K GEMMs below (each reads three of dg's four column blocks)
      static const bool wgrad_on = knob("MIMRL_NO_GRU_WGRAD") == nullptr;   // tuning knob
      // (long sequences only: at cfg2's 6400 rows the launch is 40 us against 31 for the pair, whose two launches overlap -- 0.801 vs 0.799 ms)
      const bool one_pass = wgrad_on && lbf && xpack16 && BT_ >= 32768 && gru_wgrad_ok(BT_, KP());
      if (one_pass) {
        GruWgradArgs w;
        const __bf16* xb = reinterpret_cast<const __bf16*>(xpack + BT_ * KP());   // the bf16 copy of the packed inputs, [modality][B*T, KP]
        for (int m = 0; m < 2; ++m)
          for (int d = 0; d < 2; ++d)
            w.seq[m * 2 + d] = GruWgradSeq{reinterpret_cast<const __bf16*>(dg[0][m][d]), xb + (long)m * BT_ * KP(),
                                           reinterpret_cast<const __bf16*>(hprev[0][m][d]), dwih_pack + (long)(m * 2 + d) * G * KP(),
                                           dwhh_pack + (long)(m * 2 + d) * G * H};
        w.rows = BT_; w.kp = KP();
        MX(gru_wgrad(stream, w));
      } else {
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
      }
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
      constexpr int tail_n = 1;   // (an environment knob until round 5: fixed at its measured optimum)
      constexpr int wg_sides = 3;
      auto pick = [&]() { if (l == 0) { const int q = rr++ % tail_n; return q == 0 ? stream : S(q); } return l1_side0 == 4 ? S(4 + rr++ % 2) : S(1 + rr++ % wg_sides); };
      { GemmDesc q = gemm_tn(dg[l][m][0], 4 * H, in, gf.din, Gm(gf.w_ih), gf.din, G, gf.din, (int)BT_);
        q.batch = 2; q.sa_b = s_dg; q.sb_b = 0; q.sc_b = gr.w_ih - gf.w_ih; q.atomic = 1; two(q, o_dg, o_in, o_wih);
        if (lbf) { q.a_bf16 = 1; q.sa_b *= 2; q.sa_bo *= 2; }
        // layer 1: the layer-0 outputs from the recurrence kernel's fp16 copy (half the bytes of an operand the 128-wide tiles fetch ~2x;
        // fp16 -> bf16 in registers: the product rounds to bf16 anyway, the double rounding moves a value by <= 2^-11 of itself).
        // MIMRL_DWIH_H16=0: the fp32 outputs.
        if (lbf && both && dwih_h16_on && h16_live && h0h[0] && h0h[1]) {
          q.B = reinterpret_cast<const float*>(h0h[0]); q.b_bf16 = 1; q.b_f16cvt = 1; q.sb_bo = h0h[1] - h0h[0];
        }
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
