"""CPU oracle for the MIMRL two-stage training step (TEST INFRASTRUCTURE, see oracle/__init__.py).

Functional restatement -- every function takes an explicit ``params`` dict keyed by the
reference's ``state_dict`` names and plain tensors.  Autograd is
used for gradients (it is the ground truth the hand-written HIP backward is checked
against).  Citations are ``file:line`` into /root/reference.

Parity: pinned by tests/golden/*.npz (captured from the real reference by
tests/golden/make_golden.py).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor
Params = Dict[str, Tensor]

VMI_NAMES = ["f_t", "f_a", "f_v", "t_a", "t_v"]                      # Model.py:290-295
VCMI_NAMES = ["ac_t", "ta_c", "vc_t", "tv_c", "tc_a", "tc_v"]        # Model.py:298-303


# --------------------------------------------------------------------------------------
# small pieces
# --------------------------------------------------------------------------------------
def _act(name: str):
    """Utils.py:84-97 activation table (functional)."""
    return {"gelu": F.gelu, "relu": F.relu, "tanh": torch.tanh, "elu": F.elu,
            "leakyrelu": F.leaky_relu, "hardtanh": F.hardtanh}[name]


def _linear(p: Params, name: str, x: Tensor) -> Tensor:
    w = p[name + ".weight"]
    b = p.get(name + ".bias")
    y = x @ w.t()
    return y if b is None else y + b


def gru_direction(x: Tensor, lengths: Tensor, w_ih: Tensor, w_hh: Tensor, b_ih: Tensor,
                  b_hh: Tensor, reverse: bool) -> Tensor:
    """One direction of one nn.GRU layer with packed-sequence semantics.

    Model.py:441-447 (pack_padded_sequence -> nn.GRU -> pad_packed_sequence): positions
    t >= length emit 0 and do not advance the state; the reverse direction starts at
    t = length-1.  Gate order r,z,n;  n = tanh(W_in x + b_in + r*(W_hn h + b_hn)).
    x: [B,T,D] -> [B,T,H]
    """
    B, T, _ = x.shape
    H = w_hh.shape[1]
    gx = x @ w_ih.t() + b_ih                                   # [B,T,3H]
    h = x.new_zeros(B, H)
    outs: List[Optional[Tensor]] = [None] * T
    steps = range(T - 1, -1, -1) if reverse else range(T)
    for t in steps:
        gh = h @ w_hh.t() + b_hh
        r = torch.sigmoid(gx[:, t, :H] + gh[:, :H])
        z = torch.sigmoid(gx[:, t, H:2 * H] + gh[:, H:2 * H])
        n = torch.tanh(gx[:, t, 2 * H:] + r * gh[:, 2 * H:])
        h_new = (1.0 - z) * n + z * h
        valid = (lengths > t).to(x.dtype).unsqueeze(1)          # [B,1]
        h = valid * h_new + (1.0 - valid) * h
        outs[t] = valid * h_new
    return torch.stack(outs, dim=1)


def bigru2(p: Params, prefix: str, x: Tensor, lengths: Tensor) -> Tensor:
    """nn.GRU(d, H, num_layers=2, bidirectional=True, batch_first=True) (Model.py:254-255)
    followed by the fwd/bwd half *sum* of Model.py:452-453.  -> [B,T,H]"""
    inp = x
    for layer in range(2):
        outs = []
        for rev, sfx in ((False, ""), (True, "_reverse")):
            outs.append(gru_direction(
                inp, lengths,
                p[f"{prefix}.weight_ih_l{layer}{sfx}"], p[f"{prefix}.weight_hh_l{layer}{sfx}"],
                p[f"{prefix}.bias_ih_l{layer}{sfx}"], p[f"{prefix}.bias_hh_l{layer}{sfx}"], rev))
        inp = torch.cat(outs, dim=-1)                           # [B,T,2H]
    H = inp.shape[-1] // 2
    return inp[..., :H] + inp[..., H:]


def lstm_direction(x: Tensor, lengths: Tensor, w_ih: Tensor, w_hh: Tensor, b_ih: Tensor, b_hh: Tensor,
                   reverse: bool) -> Tensor:
    """One direction of nn.LSTM(d, H, 1, bidirectional=True) with packed-sequence semantics (Model.py:250-252,441-447).
    Gate order i,f,g,o;  c = f*c + i*g;  h = o*tanh(c).  x: [B,T,D] -> [B,T,H]"""
    B, T, _ = x.shape
    H = w_hh.shape[1]
    gx = x @ w_ih.t() + b_ih                                   # [B,T,4H]
    h = x.new_zeros(B, H)
    c = x.new_zeros(B, H)
    outs: List[Optional[Tensor]] = [None] * T
    steps = range(T - 1, -1, -1) if reverse else range(T)
    for t in steps:
        g = gx[:, t] + h @ w_hh.t() + b_hh
        i, f, gg, o = torch.sigmoid(g[:, :H]), torch.sigmoid(g[:, H:2 * H]), torch.tanh(g[:, 2 * H:3 * H]), torch.sigmoid(g[:, 3 * H:])
        c_new = f * c + i * gg
        h_new = o * torch.tanh(c_new)
        valid = (lengths > t).to(x.dtype).unsqueeze(1)
        h = valid * h_new + (1.0 - valid) * h
        c = valid * c_new + (1.0 - valid) * c
        outs[t] = valid * h_new
    return torch.stack(outs, dim=1)


def bilstm1(p: Params, prefix: str, x: Tensor, lengths: Tensor) -> Tensor:
    """nn.LSTM(d, H, 1, bidirectional=True, batch_first=True) (Model.py:250-252) + the half sum of Model.py:452-453."""
    outs = [lstm_direction(x, lengths, p[f"{prefix}.weight_ih_l0{sfx}"], p[f"{prefix}.weight_hh_l0{sfx}"],
                           p[f"{prefix}.bias_ih_l0{sfx}"], p[f"{prefix}.bias_hh_l0{sfx}"], rev)
            for rev, sfx in ((False, ""), (True, "_reverse"))]
    return outs[0] + outs[1]


def _dropout(x: Tensor, p_drop: float, mask: Optional[Tensor]) -> Tensor:
    """Inverted dropout with an *explicit* keep mask (0/1) so both sides can share it."""
    if mask is None or p_drop <= 0.0:
        return x
    return x * mask / (1.0 - p_drop)


# --------------------------------------------------------------------------------------
# CubeMLP  (MLPProcess.py)
# --------------------------------------------------------------------------------------
def _axis_mlp(p: Params, pre: str, x: Tensor, act) -> Tensor:
    """MLP.forward (MLPProcess.py:17-21) on the last dim."""
    return _linear(p, pre + ".fc2", act(_linear(p, pre + ".fc1", x)))


def cube_block(p: Params, pre: str, x: Tensor, act, ln_first: bool, res_project: bool,
               eps: float = 1e-6) -> Tensor:
    """MLPsBlock.forward_ln_last (MLPProcess.py:94-122) / forward_ln_first (:64-92), dropout_mlp = 0.
    x: [B,L,K,D]"""
    def ln(name, y):
        return F.layer_norm(y, (y.shape[-1],), p[f"{pre}.{name}.weight"], p[f"{pre}.{name}.bias"], eps)

    def res(name, y):
        return y @ p[f"{pre}.{name}.weight"].t() if res_project else y

    # --- L axis: operate on x permuted to [B,K,D,L]
    xl = x.permute(0, 2, 3, 1)
    if ln_first:
        yl = _axis_mlp(p, pre + ".mlp_l", ln("ln_l", xl), act) + res("res_projection_l", xl)
    else:
        yl = ln("ln_l", _axis_mlp(p, pre + ".mlp_l", xl, act) + res("res_projection_l", xl))
    x = yl.permute(0, 3, 1, 2)                                  # [B,L',K,D]
    # --- K axis: [B,L,D,K]
    xk = x.permute(0, 1, 3, 2)
    if ln_first:
        yk = _axis_mlp(p, pre + ".mlp_k", ln("ln_k", xk), act) + res("res_projection_k", xk)
    else:
        yk = ln("ln_k", _axis_mlp(p, pre + ".mlp_k", xk, act) + res("res_projection_k", xk))
    x = yk.permute(0, 1, 3, 2)
    # --- D axis
    if ln_first:
        x = _axis_mlp(p, pre + ".mlp_d", ln("ln_d", x), act) + res("res_projection_d", x)
    else:
        x = ln("ln_d", _axis_mlp(p, pre + ".mlp_d", x, act) + res("res_projection_d", x))
    return x


def cube_mlp(p: Params, opt, x: Tensor) -> Tensor:
    """MLPEncoder.forward (MLPProcess.py:134-137)."""
    act = _act(opt.activate)
    for i in range(len(opt.d_hiddens)):
        x = cube_block(p, f"mlp_encoder.layers_stack.{i}", x, act, opt.ln_first, opt.res_project[i])
    return x


# --------------------------------------------------------------------------------------
# Model.forward (Model.py:388-519), BERT replaced by precomputed text features
# --------------------------------------------------------------------------------------
def infer_lengths(x: Tensor) -> Tensor:
    """Model.py:425-432 + Utils.py:297-298: length = #rows with non-zero |.|-sum, clamped >= 1."""
    valid = (x.abs().sum(-1) != 0).to(torch.int64).sum(1)
    return torch.clamp(valid, min=1)


def model_forward(p: Params, opt, t_feat: Tensor, a: Tensor, v: Tensor,
                  masks: Optional[Dict[str, Tensor]] = None):
    """-> (pred[B,1], F_F, T_F, A_F, V_F [B,128]).  ``t_feat`` is the BERT last hidden state
    [B,T,768] (Model.py:391).  ``masks`` optionally carries explicit dropout keep-masks
    't','a','v' ([B,T,128]) so that a device run with the same masks is comparable."""
    masks = masks or {}
    t = t_feat @ p["W_t.weight"].t()                            # Model.py:395
    D = opt.d_common
    if getattr(opt, "encoders", "gru") == "conv":               # Model.py:247-249,437-438: Conv1d(k=3, padding=1) over time
        conv = lambda x, m: F.conv1d(x.transpose(1, 2), p[f"conv_{m}.weight"].reshape(D, x.shape[-1], 3), p[f"conv_{m}.bias"],
                                     padding=1).transpose(1, 2)
        ah, vh = conv(a, "a"), conv(v, "v")
    else:
        la, lv = infer_lengths(a), infer_lengths(v)             # Model.py:425-432
        rnn = bilstm1 if getattr(opt, "encoders", "gru") == "lstm" else bigru2
        ah = rnn(p, "rnn_a", a, la)                             # Model.py:441-453
        vh = rnn(p, "rnn_v", v, lv)
    ah = F.relu(F.layer_norm(ah, (D,), p["ln_a.weight"], p["ln_a.bias"], 1e-6))   # Model.py:457
    vh = F.relu(F.layer_norm(vh, (D,), p["ln_v.weight"], p["ln_v.bias"], 1e-6))
    t = _dropout(t, opt.dropout[0], masks.get("t"))             # Model.py:461
    ah = _dropout(ah, opt.dropout[1], masks.get("a"))
    vh = _dropout(vh, opt.dropout[2], masks.get("v"))
    T_F, A_F, V_F = t.mean(1), ah.mean(1), vh.mean(1)           # Model.py:466
    L = opt.time_len
    pad = lambda y: F.pad(y, (0, 0, 0, L - y.shape[1]))          # Model.py:468-470
    x = torch.stack([pad(t), pad(ah), pad(vh)], dim=2)          # Model.py:475  [B,L,3,D]
    x = cube_mlp(p, opt, x)                                     # Model.py:481
    fk = x.mean(2) if opt.features_compose_k == "mean" else x.sum(2)       # Model.py:489-492
    ff = fk.mean(1) if opt.features_compose_t == "mean" else fk.sum(1)     # Model.py:499-502
    F_F = ff                                                    # Model.py:507-511
    pred = ff @ p["classifier.0.weight"].t() + p["classifier.0.bias"]      # Model.py:271-274,515
    return pred, F_F, T_F, A_F, V_F


# --------------------------------------------------------------------------------------
# critics + InfoNCE (VMI.py)
# --------------------------------------------------------------------------------------
def _tower(p: Params, pre: str, x: Tensor, idxs=(0, 2, 4, 6)) -> Tensor:
    """VMI.py:13-22 ``mlps``: Linear/ReLU x3 + Linear."""
    h = x
    for j, i in enumerate(idxs):
        h = _linear(p, f"{pre}.{i}", h)
        if j < len(idxs) - 1:
            h = F.relu(h)
    return h


def critic_scores(p: Params, name: str, critic_type: str, x: Tensor, y: Tensor) -> Tensor:
    """CriticModel.forward (VMI.py:53-69).  -> scores [B,B]"""
    pre = f"vmi_estimator_{name}.critic_model"
    if critic_type == "separate":
        g = _tower(p, pre + ".MLP_g", x)
        h = _tower(p, pre + ".MLP_h", y)
        return h @ g.t()                                        # VMI.py:57
    if critic_type == "concat":
        B = x.shape[0]
        # VMI.py:61-65: raw[i,j] = f([x_j | y_i]); scores = raw^T
        xt = x.unsqueeze(0).expand(B, B, -1)
        yt = y.unsqueeze(1).expand(B, B, -1)
        raw = _tower(p, pre + ".MLP_f", torch.cat([xt, yt], dim=2).reshape(B * B, -1))
        return raw.reshape(B, B).t()
    raise NotImplementedError(critic_type)


def infonce_lower_bound(scores: Tensor) -> Tensor:
    """VMI.py:162-166."""
    nll = torch.mean(scores.diag() - torch.logsumexp(scores, dim=1))
    return math.log(scores.shape[0]) + nll


def _logmeanexp_nodiag(x: Tensor) -> Tensor:
    """VMI.py:121-126."""
    B = x.shape[0]
    x = x - torch.diag(torch.full((B,), float("inf"), dtype=x.dtype))
    return torch.logsumexp(x.reshape(-1), 0) - math.log(B * (B - 1.0))


def tuba_lower_bound(scores: Tensor, log_baseline: Optional[Tensor] = None) -> Tensor:
    """VMI.py:148-154; log_baseline [B,1] is subtracted row-wise (None = the constant zero baseline)."""
    if log_baseline is not None:
        scores = scores - log_baseline
    return 1.0 + scores.diag().mean() - torch.exp(_logmeanexp_nodiag(scores))


def log_baseline(p: Params, name: str, opt, y: Tensor) -> Optional[Tensor]:
    """BaselineModel.forward (VMI.py:72-110) -> [B,1] or None for the constant baseline.  `gaussain`: sum of
    Normal(mu=0, rho=1) log-densities (Model.py:290-295 pass mu=0, rho=1); `unnormalized`: a trainable mlps(128,256,1,2)."""
    kind = getattr(opt, "baseline_type", "constant")
    if kind == "constant":
        return None
    if kind == "gaussain":
        return (-0.5 * y * y - 0.5 * math.log(2 * math.pi)).sum(-1, keepdim=True)
    if kind == "unnormalized":
        h = y
        for j, i in enumerate((0, 2, 4, 6)):
            h = _linear(p, f"vmi_estimator_{name}.baseline_model.MLP.{i}", h)
            if j < 3:
                h = torch.relu(h)
        return h.reshape(-1, 1)
    raise NotImplementedError(kind)


def nwj_lower_bound(scores: Tensor) -> Tensor:
    """VMI.py:157-159."""
    return tuba_lower_bound(scores - 1.0)


def dv_lower_bound(scores: Tensor) -> Tensor:
    """VMI.py:136-139."""
    return scores.diag().mean() - _logmeanexp_nodiag(scores)


def js_fgan_lower_bound(scores: Tensor) -> Tensor:
    """VMI.py:169-174."""
    B = scores.shape[0]
    d = scores.diag()
    return torch.mean(-F.softplus(-d)) - (F.softplus(scores).sum() - F.softplus(d).sum()) / (B * (B - 1))


def js_lower_bound(scores: Tensor) -> Tensor:
    """VMI.py:177-182 (value = nwj, gradient = js)."""
    js = js_fgan_lower_bound(scores)
    return js + (nwj_lower_bound(scores) - js).detach()


def smile_lower_bound(scores: Tensor, clip: float = 1.0) -> Tensor:
    """VMI.py:185-198 (clip is forced to 1)."""
    dv = scores.diag().mean() - _logmeanexp_nodiag(torch.clamp(scores, -clip, clip))
    js = js_fgan_lower_bound(scores)
    return js + (dv - js).detach()


def interp_lower_bound(scores: Tensor, log_baseline: Optional[Tensor] = None, alpha_logit: float = 0.01) -> Tensor:
    """VMI.py:201-250 with the constant baseline (log a(y) = 0) and the alpha_logit VMIEstimator hard-codes (Model.py:118).
    nce baseline = leave-one-out log-mean-exp of each row; interpolated with the constant baseline in log space."""
    B = scores.shape[0]
    lse = torch.logsumexp(scores, dim=1, keepdim=True)
    d = lse - scores                                                   # >= 0; log(sum_k e^s_ik) - s_ij
    safe_d = torch.where(d == 0, torch.ones_like(d), d)
    loo_lme = scores + (safe_d + torch.log(-torch.expm1(-safe_d))) - math.log(B - 1.0)    # compute_log_loomean
    log_alpha = -F.softplus(torch.tensor(-float(alpha_logit), dtype=scores.dtype))
    log_1m = -F.softplus(torch.tensor(float(alpha_logit), dtype=scores.dtype))
    base = torch.zeros_like(loo_lme) if log_baseline is None else log_baseline.repeat(1, B)     # element (i,j) = log a(y_i)
    interp = torch.logsumexp(torch.stack((log_alpha + loo_lme, log_1m + base)), dim=0)
    critic_marg = scores - torch.diag(interp)                          # broadcasts over rows: s_ij - interp_jj
    marg = torch.exp(_logmeanexp_nodiag(critic_marg))
    critic_joint = torch.diag(scores) - interp                         # s_jj - interp_ij
    joint = (critic_joint.sum() - torch.diag(critic_joint).sum()) / (B * (B - 1.0))
    return 1 + joint - marg


BOUNDS = {"infonce": infonce_lower_bound, "nwj": nwj_lower_bound, "tuba": tuba_lower_bound,
          "dv": dv_lower_bound, "js_fgan": js_fgan_lower_bound, "js": js_lower_bound,
          "smile": smile_lower_bound, "interpolate": interp_lower_bound}


def vmi_estimate(p: Params, name: str, opt, x: Tensor, y: Tensor) -> Tuple[Tensor, Tensor]:
    """VMIEstimator.forward (Model.py:115-148), constant baseline.  -> (mi, mi_loss)"""
    s = critic_scores(p, name, opt.critic_type, x, y)
    if opt.bound_type == "mine":                                # Model.py:121-125 + VMI.py:128-133,142-145
        B = s.shape[0]
        t = s.diag()
        et = torch.exp(s) * (1.0 - torch.eye(B, dtype=s.dtype))    # exp_nodiag: exp(-inf) = 0 on the diagonal
        mi = t.mean() - _logmeanexp_nodiag(s)
        ma_et = (1 - 0.01) * 1 + 0.01 * et.mean()
        return mi, t.mean() - (1 / ma_et.mean()).detach() * et.mean()   # NB: not negated in the reference
    if opt.bound_type in ("tuba", "interpolate"):               # the only bounds that read the baseline (Model.py:127-142)
        mi = BOUNDS[opt.bound_type](s, log_baseline(p, name, opt, y))
    else:
        mi = BOUNDS[opt.bound_type](s)
    return mi, -mi


# --------------------------------------------------------------------------------------
# kNN product sampler + classifier CMI (Model.py:75-106, 150-225)
# --------------------------------------------------------------------------------------
def knn_indices(Z: np.ndarray, anchors: np.ndarray, k: int) -> np.ndarray:
    """Exact Euclidean kNN of Z[anchors] among the non-anchor rows of Z (Model.py:81-86).
    Returns indices into the ORIGINAL bank, [m,k], nearest first.

    Ties.  For the 128-column feature banks (continuous values) there are none and a brute-force argsort is the restatement.
    For a 1-column Z -- the label bank of the ta_c / tv_c estimators; real MOSI / MOSEI labels are discrete, so every anchor
    has many rows at distance 0 -- the result IS scikit-learn's tie order: the reference calls
    ``sklearn.neighbors.NearestNeighbors(n_neighbors=k, radius=radius, metric='euclidean')`` (Model.py:82), whose
    algorithm='auto' builds a KDTree for <= 15 features.  That third-party dependency is not part of /root/reference
    (pinned here: scikit-learn 1.7.2, the version of this image); the oracle calls it exactly as the reference does, the product
    restates it in csrc/knn_r1.cpp, and tests/test_knn_ties.py pins the one against the other."""
    Z = np.asarray(Z)
    N = Z.shape[0]
    keep = np.ones(N, dtype=bool)
    keep[anchors] = False
    cand = np.nonzero(keep)[0]
    if Z.ndim == 2 and Z.shape[1] == 1 and k < len(cand) // 2:
        from sklearn.neighbors import NearestNeighbors
        neigh = NearestNeighbors(n_neighbors=k, radius=1.0, metric="euclidean")
        neigh.fit(Z[cand])                                                     # Model.py:83-85 (rows that are not anchors)
        return cand[neigh.kneighbors(Z[np.asarray(anchors)], return_distance=False)].astype(np.int64)
    Zd = np.asarray(Z, dtype=np.float64)
    out = np.empty((len(anchors), k), dtype=np.int64)
    for i, a in enumerate(anchors):
        d = ((Zd[cand] - Zd[a]) ** 2).sum(1)
        order = np.argsort(d, kind="stable")[:k]
        out[i] = cand[order]
    return out


def prod_knn_sample(X: Tensor, Y: Tensor, Z: Tensor, anchors: np.ndarray, k: int):
    """Model.py:75-106 with the ``np.random.choice`` draw (``anchors``) made explicit.
    Rows are anchor-major: (X[nn_j(i)], Y[i], Z[i]); narrow operands tiled to the widest."""
    nn_idx = knn_indices(Z.detach().numpy(), anchors, k)                  # [m,k]
    ix = torch.as_tensor(nn_idx.reshape(-1))
    iyz = torch.as_tensor(np.repeat(np.asarray(anchors, dtype=np.int64), k))
    bx, by, bz = X.detach()[ix], Y.detach()[iyz], Z.detach()[iyz]
    w = max(bx.shape[1], by.shape[1], bz.shape[1])
    tile = lambda b: b if b.shape[1] == w else b.repeat(1, w // b.shape[1])   # Model.py:98-104
    return tile(bx), tile(by), tile(bz), nn_idx


def cmi_classifier(p: Params, name: str, feats: Tensor, last_act: str) -> Tensor:
    """MLP_For_CMI.forward (Model.py:65-72)."""
    pre = f"vcmi_estimator_{name}.classifier.mlp"
    h = _tower(p, pre, feats)
    h = torch.clamp(h, -10, 10)
    if last_act == "sigmoid":
        return torch.sigmoid(h)
    if last_act == "hardtanh":
        return F.hardtanh(h, 1e-4, 1 - 1e-4)
    raise NotImplementedError(last_act)


def vcmi_estimate(p: Params, name: str, opt, x, y, z, kx, ky, kz) -> Tuple[Tensor, Tensor]:
    """VCMIEstimator.forward + estimate_cmi (Model.py:157-225).  -> (cmi, bce_loss)"""
    E = 128                                                     # Model.py:285 embed_dim
    tile = lambda f: f if f.shape[1] == E else f.repeat(1, E // f.shape[1])    # Model.py:161-166
    joint = torch.cat([tile(x), tile(y), tile(z)], dim=1)
    prod = torch.cat([kx, ky, kz], dim=1)
    n = prod.shape[0]
    if joint.shape[0] != n:                                     # Model.py:180-182
        joint = joint[:n]
    batch = torch.cat([joint, prod], dim=0)                     # [2n,384]
    target = torch.zeros(2 * n, 2, dtype=batch.dtype)
    target[:n, 0] = 1.0
    target[n:, 1] = 1.0
    out = cmi_classifier(p, name, batch, opt.cmi_last_acticate)
    loss = F.binary_cross_entropy(out, target)                  # Model.py:198
    gamma = cmi_classifier(p, name, batch, opt.cmi_last_acticate)[:, 0]        # Model.py:206 (2nd pass)
    lr = torch.log(gamma / (1 - gamma + 1e-6))
    cmi = 1.0 + lr[:n].sum() / (2 * n) - lr[n:].sum() / (2 * n)                 # Model.py:215-219
    return cmi, loss


# operand wiring of the six CMI estimators: (sample X,Y,Z banks) ; (estimate x,y,z)  -- Model.py:323-339
_CMI_WIRING = {
    "ac_t": ("A", "C", "T"), "ta_c": ("T", "A", "C"), "vc_t": ("V", "C", "T"),
    "tv_c": ("T", "V", "C"), "tc_a": ("T", "C", "A"), "tc_v": ("T", "C", "V"),
}
_MI_WIRING = {"f_t": ("F", "T"), "f_a": ("F", "A"), "f_v": ("F", "V"), "t_a": ("T", "A"), "t_v": ("T", "V")}


def estimator_terms(p: Params, opt, labels: Tensor, feats: Dict[str, Tensor], banks: Dict[str, Tensor],
                    anchors: Sequence[np.ndarray]):
    """Shared body of compute_vmi_loss_stage1/2 (Model.py:305-339 / 343-379).
    feats: F,T,A,V [B,128]; banks: C [N,1], F,T,A,V [N,128]; anchors: six index arrays."""
    B = labels.shape[0]
    C = labels.reshape(-1, 1).repeat(1, opt.d_common)           # Model.py:307
    cur = dict(feats, C=C)
    mi, mi_loss = {}, {}
    for n in VMI_NAMES:
        a, b = _MI_WIRING[n]
        mi[n], mi_loss[n] = vmi_estimate(p, n, opt, cur[a], cur[b])
    cmi, cmi_loss = {}, {}
    for n, anc in zip(VCMI_NAMES, anchors):
        a, b, c = _CMI_WIRING[n]
        kx, ky, kz, _ = prod_knn_sample(banks[a], banks[b], banks[c], anc, opt.k_neighbor)
        cmi[n], cmi_loss[n] = vcmi_estimate(p, n, opt, cur[a], cur[b], cur[c], kx, ky, kz)
    return mi, mi_loss, cmi, cmi_loss


def stage1_terms(p, opt, labels, feats, banks, anchors):
    """Model.compute_vmi_loss_stage1 (Model.py:305-341) -> (mis[11], losses[11])."""
    mi, ml, cmi, cl = estimator_terms(p, opt, labels, feats, banks, anchors)
    return ([mi[n] for n in VMI_NAMES] + [cmi[n] for n in VCMI_NAMES],
            [ml[n] for n in VMI_NAMES] + [cl[n] for n in VCMI_NAMES])


def stage2_terms(p, opt, labels, feats, banks, anchors):
    """Model.compute_vmi_loss_stage2 (Model.py:343-386) -> (mis[8], losses[8])."""
    mi, ml, cmi, _ = estimator_terms(p, opt, labels, feats, banks, anchors)
    inv = mi["t_a"] + mi["t_v"]
    spec_t = cmi["tc_a"] + cmi["tc_v"] - cmi["ta_c"] - cmi["tv_c"]
    spec_a = cmi["ac_t"] - cmi["ta_c"]
    spec_v = cmi["vc_t"] - cmi["tv_c"]
    comp = cmi["ta_c"] + cmi["tv_c"]
    return ([mi["f_t"], mi["f_a"], mi["f_v"], inv, spec_t, spec_a, spec_v, comp],
            [ml["f_t"], ml["f_a"], ml["f_v"], -inv, -spec_t, -spec_a, -spec_v, -comp])


# --------------------------------------------------------------------------------------
# losses, optimiser, the two stage steps (Solver.py:204-216, 220-238; Customization.py:91-115)
# --------------------------------------------------------------------------------------
def is_critic_param(name: str) -> bool:
    """Solver.py:124-133: optimiser split by substring."""
    return ("vmi" in name) or ("vcmi" in name)


def task_loss_mae(pred: Tensor, labels: Tensor) -> Tensor:
    """Solver.py:181-182, 334-335."""
    return (pred.reshape(-1) - labels.reshape(-1)).abs().mean()


def stage_loss(p, opt, stage: int, batch, banks, anchors, masks=None):
    """One forward of Solver.train's loop body up to the scalar loss.
    batch = (t_feat, a, v, labels); banks None/empty => epoch-0 rule (Customization.py:97-98,105-106)."""
    t_feat, a, v, labels = batch
    pred, F_F, T_F, A_F, V_F = model_forward(p, opt, t_feat, a, v, masks)
    feats = {"F": F_F, "T": T_F, "A": A_F, "V": V_F}
    task = task_loss_mae(pred, labels)
    empty = banks is None or len(banks["C"]) == 0
    if stage == 1:
        if empty:
            return torch.zeros(()), [torch.zeros(()) for _ in range(8)], pred, feats, task
        mis, losses = stage1_terms(p, opt, labels, feats, banks, anchors)
        loss = sum(l * c for l, c in zip(losses, opt.loss_mi_coefficient1))
        return loss, mis, pred, feats, task
    if empty:
        return task, [torch.zeros(()) for _ in range(8)], pred, feats, task
    mis, losses = stage2_terms(p, opt, labels, feats, banks, anchors)
    loss = task + sum(l * c for l, c in zip(losses, opt.loss_mi_coefficient2))
    return loss, mis, pred, feats, task


class AdamState:
    """torch.optim.Adam (Solver.py:144-146): betas (0.9,0.999), eps 1e-8, L2 weight decay."""

    def __init__(self, params: Params, names: Sequence[str]):
        self.names = list(names)
        self.m = {n: torch.zeros_like(params[n]) for n in self.names}
        self.v = {n: torch.zeros_like(params[n]) for n in self.names}
        self.t = 0

    def step(self, params: Params, grads: Dict[str, Tensor], lr: float, wd: float = 0.0,
             b1: float = 0.9, b2: float = 0.999, eps: float = 1e-8):
        self.t += 1
        bc1 = 1.0 - b1 ** self.t
        bc2 = 1.0 - b2 ** self.t
        for n in self.names:
            g = grads.get(n)
            if g is None:
                continue
            if wd != 0.0:
                g = g + wd * params[n]
            self.m[n] = b1 * self.m[n] + (1 - b1) * g
            self.v[n] = b2 * self.v[n] + (1 - b2) * g * g
            denom = self.v[n].sqrt() / math.sqrt(bc2) + eps
            params[n] = (params[n] - (lr / bc1) * self.m[n] / denom).detach()


def stage_step(params: Params, opt, stage: int, adam: AdamState, batch, banks, anchors, masks=None, lr_scale: float = 1.0):
    """One optimiser update of stage 1 (critics; Solver.py:205-214) or stage 2 (main; :221-236).
    Mutates ``params`` in place (dict entries are replaced).  Returns dict of observables.
    ``lr_scale`` = the factor the epoch-level lr schedulers have applied so far (Solver.py:52-57,153-169)."""
    leaves = {k: v.detach().clone().requires_grad_(True) for k, v in params.items()}
    loss, mis, pred, feats, task = stage_loss(leaves, opt, stage, batch, banks, anchors, masks)
    grads: Dict[str, Tensor] = {}
    if loss.requires_grad:
        names = [n for n in adam.names]
        gs = torch.autograd.grad(loss, [leaves[n] for n in names], allow_unused=True)
        clip = float(opt.gradient_clip)
        for n, g in zip(names, gs):
            if g is None:
                continue
            grads[n] = g.clamp(-clip, clip) if clip > 0 else g   # Solver.py:211-212 clip_grad_value_
    lr = float(opt.learning_rate) * (float(opt.mi_lr_rate) if stage == 1 else 1.0) * lr_scale   # Solver.py:135-142
    if grads or stage == 2:
        adam.step(params, grads, lr, float(opt.weight_decay))
    return {"loss": loss.detach(), "mis": [m.detach() for m in mis], "pred": pred.detach(),
            "feats": {k: v.detach() for k, v in feats.items()}, "task": task.detach(), "grads": grads}


def two_stage_step(params: Params, opt, adam_vmi: AdamState, adam_main: AdamState, batch, banks,
                   anchors1, anchors2, masks1=None, masks2=None):
    """``Solver.step``: one stage-1 update then one stage-2 update on the same batch."""
    r1 = stage_step(params, opt, 1, adam_vmi, batch, banks, anchors1, masks1)
    r2 = stage_step(params, opt, 2, adam_main, batch, banks, anchors2, masks2)
    return r1, r2


# --------------------------------------------------------------------------------------
# epoch level: Solver.train / Solver.evaluate (Solver.py:194-248, 250-270)
# --------------------------------------------------------------------------------------
def _bank_dict(banks):
    if banks is None or len(banks["C"]) == 0:
        return None
    return banks


def train_epoch(params: Params, opt, epoch: int, adam_vmi: AdamState, adam_main: AdamState, batches, banks, draw,
                lr_scale: float = 1.0):
    """Solver.train (Solver.py:194-248).  ``batches``: list of (t_feat, a, v, labels) -- the last one may be shorter
    (Parameters.py:21 drop_last defaults to False); ``banks``: dict C,F,T,A,V of the previous epoch's stage-2 pass or
    None/empty; ``draw(N, m)`` -> six anchor arrays, called once per estimator pass in the reference's order (Model.py:81).
    Returns the reference's return tuple as a dict (losses averaged over len(batches), new banks = features of THIS pass)."""
    banks = _bank_dict(banks)
    nb = len(batches)
    run_mi = 0.0
    if epoch > 0:                                                          # Solver.py:200-203: epoch 0 skips stage 1
        for _ in range(int(opt.stage1_n)):
            for batch in batches:
                anc = draw(len(banks["C"]), batch[3].shape[0] // opt.k_neighbor) if banks is not None else None
                r = stage_step(params, opt, 1, adam_vmi, batch, banks, anc, lr_scale=lr_scale)
                run_mi += float(r["loss"])
    new = {k: [] for k in "CFTAV"}
    run, mis, preds, targs = 0.0, [0.0] * 8, [], []
    for batch in batches:
        anc = draw(len(banks["C"]), batch[3].shape[0] // opt.k_neighbor) if banks is not None else None
        r = stage_step(params, opt, 2, adam_main, batch, banks, anc, lr_scale=lr_scale)
        new["C"].append(batch[3].reshape(-1, 1))                           # Solver.py:223-227
        for k in "FTAV":
            new[k].append(r["feats"][k])
        run += float(r["loss"])
        mis = [a + float(b) for a, b in zip(mis, r["mis"])]
        preds.append(r["pred"].reshape(-1))
        targs.append(batch[3].reshape(-1))
    return {"loss": run / nb, "loss_mi": run_mi / nb, "mis": [m / nb for m in mis], "pred": torch.cat(preds), "target": torch.cat(targs),
            "banks": {k: torch.cat(v, 0) for k, v in new.items()}}


def evaluate_epoch(params: Params, opt, batches, banks, draw):
    """Solver.evaluate (Solver.py:250-270): the stage-2 loss of every batch under no_grad, eval mode (dropout off)."""
    banks = _bank_dict(banks)
    run, mis, preds = 0.0, [0.0] * 8, []
    with torch.no_grad():
        for batch in batches:
            anc = draw(len(banks["C"]), batch[3].shape[0] // opt.k_neighbor) if banks is not None else None
            loss, m, pred, feats, task = stage_loss(params, opt, 2, batch, banks, anc)
            run += float(loss)
            mis = [a + float(b) for a, b in zip(mis, m)]
            preds.append(pred.reshape(-1))
    nb = len(batches)
    return {"loss": run / nb, "mis": [m / nb for m in mis], "pred": torch.cat(preds)}
