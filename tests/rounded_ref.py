"""float64 restatements of the fused bf16 sub-blocks with the kernels' ROUNDING POINTS made explicit (test infrastructure).

Each function takes a ``Rounding`` (forward operand rounding + backward operand rounding).  With ``EXACT`` it is the oracle's function (tests/test_rounded_ref.py asserts
equality with oracle.mimrl_ref.cube_block / critic_scores in float64, which pins these restatements to the oracle and through it to
the reference: MLPProcess.py:94-122, VMI.py:53-69); with ``BF16`` / ``F16_FWD`` it is what the HIP kernels compute, up to fp32
accumulation order: round-to-nearest-even of exactly the operands the kernels round -- in the forward products AND in the two
backward products of every layer (data gradient: rounded incoming gradient x rounded weight; weight gradient: rounded incoming
gradient x rounded saved input) -- and float64 everywhere else (biases, LayerNorm, activations and their derivatives, bias and
LayerNorm parameter gradients, the K-axis mix: fp32 VALU code in the kernels).

Rounding points (documented per kernel):
  cube_fwd_fused_kernel (cube_fused.hip), per block -- its 16-bit type is FP16 (q = f16_ste), every other kernel's is bf16
    L phase : X tile, W1, W2, Wr -> 16 bit (MFMA operands); U, act(U) in fp32; H -> bf16 (operand of W2.H); Y, LayerNorm(L) in fp32;
              Z -> bf16 back into the LDS tile (the K phase reads the ROUNDED Z; the fp32 Z is only saved for the backward)
    K phase : fp32 arithmetic on the bf16 tile, fp32 K-axis weights; output -> bf16 back into the tile
    D phase : tile rows, W1, W2, Wr -> bf16 operands; U, act(U) fp32; H -> bf16; Y, LayerNorm(D) fp32; block output fp32
              (the next block rounds it when it loads its tile)
  unfused bf16 chain (engine_forward.hip cube_forward, MIMRL_NO_FUSED_CUBE=1 / ln_first): only GEMM operands are rounded
    (round_tile=False): LayerNorm outputs and the K-axis mix stay fp32
  concat critic (concat_fused.hip + the two layer-0 GEMMs): x, y, W0 -> bf16 (P = x W0x^T, Q = y W0y^T + b0, fp32 accumulate);
    a0 = relu(P_i + Q_j) -> bf16; W1, W2 bf16 images; a1 = relu(a0 W1^T + b1) -> bf16; a2 = relu(a1 W2^T + b2) stays fp32;
    score = a2 . w3 + b3 with fp32 w3
  separable critic (mlp_img8_kernel + mi_sep_nce_kernel): tower input, every W -> bf16; every hidden activation -> bf16 (it is the
    next layer's MFMA operand); tower outputs fp32, rounded to bf16 as operands of scores = h g^T
"""
import math

import torch
import torch.nn.functional as F

from oracle import mimrl_ref as R


def r_bf16(x):
    """bf16 round-to-nearest-even of a float64 tensor (through fp32, as the kernels convert)."""
    return x.to(torch.float32).to(torch.bfloat16).to(x.dtype)


def r_f16(x):
    """fp16 round-to-nearest-even (the fused CubeMLP forward's and the forward projections' operand type since round 3)."""
    return x.to(torch.float32).to(torch.float16).to(x.dtype)


def identity(x):
    return x


class Rounding:
    """fwd: rounding of both operands of a FORWARD product; bwd: rounding of the three operands of the two BACKWARD products of that
    layer (the incoming gradient, the weight for the data gradient, the saved input for the weight gradient).  bwd=None: the backward
    products reuse the forward's rounded operands (straight-through)."""
    def __init__(self, fwd=identity, bwd=None):
        self.fwd, self.bwd = fwd, bwd


EXACT = Rounding()
BF16 = Rounding(r_bf16, r_bf16)            # every bf16 GEMM / fused MLP kernel: operands rounded where they are staged, both passes
F16_FWD = Rounding(r_f16, r_bf16)          # fp16 forward operands, bf16 gradient products (cube_fwd_fused + *_bwd kernels)


class _MM(torch.autograd.Function):
    """y = x @ w^T with the kernels' operand rounding in BOTH passes (see Rounding)."""
    @staticmethod
    def forward(ctx, x, w, rnd):
        xf, wf = rnd.fwd(x), rnd.fwd(w)
        ctx.rnd = rnd
        ctx.save_for_backward(x, w, xf, wf)
        return xf @ wf.t()

    @staticmethod
    def backward(ctx, g):
        x, w, xf, wf = ctx.saved_tensors
        rnd = ctx.rnd
        if rnd.bwd is None:
            gq, xb, wb = g, xf, wf
        else:
            gq, xb, wb = rnd.bwd(g), rnd.bwd(x), rnd.bwd(w)
        dx = gq @ wb
        dw = gq.reshape(-1, gq.shape[-1]).t() @ xb.reshape(-1, xb.shape[-1])
        return dx, dw, None


def mm(x, w, rnd):
    return _MM.apply(x, w, rnd)


class _RoundSte(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, fn):
        return fn(x)

    @staticmethod
    def backward(ctx, g):
        return g, None


def ste(fn, x):
    """value rounded by ``fn``, gradient passed through (a tile written back rounded; its consumers' backward reads what they consumed)"""
    return x if fn is identity else _RoundSte.apply(x, fn)


def _ln(p, name, y, eps=1e-6):
    return F.layer_norm(y, (y.shape[-1],), p[name + ".weight"], p[name + ".bias"], eps)


def _mlp_res(p, pre, ax, x_mlp, x_res, act, rnd):
    """fc2(act(fc1 x_mlp)) + res x_res on the last dim; the three products round their operands as ``rnd`` says."""
    b1, b2 = p.get(f"{pre}.mlp_{ax}.fc1.bias"), p.get(f"{pre}.mlp_{ax}.fc2.bias")
    u = mm(x_mlp, p[f"{pre}.mlp_{ax}.fc1.weight"], rnd)
    if b1 is not None:
        u = u + b1
    y = mm(act(u), p[f"{pre}.mlp_{ax}.fc2.weight"], rnd) + mm(x_res, p[f"{pre}.res_projection_{ax}.weight"], rnd)
    return y if b2 is None else y + b2


def cube_block_q(p, pre, x, act, rnd, round_tile=True, ln_first=False):
    """MLPsBlock.forward_ln_last (MLPProcess.py:94-122) / forward_ln_first (:64-92) with res_projection, dropout 0.
    x [B,L,K,D] float64.  ln_first exists only as the unfused chain (round_tile=False)."""
    tile = rnd.fwd if round_tile else identity
    if ln_first:
        assert not round_tile
        xl = x.permute(0, 2, 3, 1)
        x = _mlp_res(p, pre, "l", _ln(p, f"{pre}.ln_l", xl), xl, act, rnd).permute(0, 3, 1, 2)
        xk = x.permute(0, 1, 3, 2)
        x = _mlp_res(p, pre, "k", _ln(p, f"{pre}.ln_k", xk), xk, act, EXACT).permute(0, 1, 3, 2)
        return _mlp_res(p, pre, "d", _ln(p, f"{pre}.ln_d", x), x, act, rnd)
    xl = x.permute(0, 2, 3, 1)
    z = _ln(p, f"{pre}.ln_l", _mlp_res(p, pre, "l", xl, xl, act, rnd))                      # [B,K,D,L']
    x = ste(tile, z).permute(0, 3, 1, 2)                                                     # tile <- rounded Z
    xk = x.permute(0, 1, 3, 2)
    zk = _ln(p, f"{pre}.ln_k", _mlp_res(p, pre, "k", xk, xk, act, EXACT))                    # fp32 K-axis mix on the tile values
    x = ste(tile, zk).permute(0, 1, 3, 2)
    return _ln(p, f"{pre}.ln_d", _mlp_res(p, pre, "d", x, x, act, rnd))


def cube_mlp_q(p, opt, x, rnd, round_tile=True):
    act = R._act(opt.activate)
    for i in range(len(opt.d_hiddens)):
        x = cube_block_q(p, f"mlp_encoder.layers_stack.{i}", x, act, rnd, round_tile, opt.ln_first)
    return x


def critic_scores_q(p, name, critic_type, x, y, rnd):
    """CriticModel.forward (VMI.py:53-69) with the kernels' rounding points.  -> scores [B,B] (row i: y_i, column j: x_j)."""
    pre = f"vmi_estimator_{name}.critic_model"

    def tower(tp, v):
        h = v
        for j, i in enumerate((0, 2, 4, 6)):
            h = mm(h, p[f"{tp}.{i}.weight"], rnd) + p[f"{tp}.{i}.bias"]
            if j < 3:
                h = F.relu(h)
        return h

    if critic_type == "separate":
        g, h = tower(pre + ".MLP_g", x), tower(pre + ".MLP_h", y)
        return mm(h, g, rnd)                                      # scores = h g^T: dh = dS g, dg = dS^T h with the same rounding
    B, E = x.shape
    w0, b0 = p[f"{pre}.MLP_f.0.weight"], p[f"{pre}.MLP_f.0.bias"]
    P = mm(x, w0[:, :E], rnd)                                     # [B,256] the x half of layer 0
    Q = mm(y, w0[:, E:], rnd) + b0
    # raw[i,j] = f([x_j | y_i]) (VMI.py:61-65); scores = raw^T, i.e. scores[a,b] = f([x_a | y_b])
    a0 = F.relu(P.unsqueeze(1) + Q.unsqueeze(0))                  # [a, b, 256]
    a1 = F.relu(mm(a0, p[f"{pre}.MLP_f.2.weight"], rnd) + p[f"{pre}.MLP_f.2.bias"])
    a2 = F.relu(mm(a1, p[f"{pre}.MLP_f.4.weight"], rnd) + p[f"{pre}.MLP_f.4.bias"])
    s = a2 @ p[f"{pre}.MLP_f.6.weight"].t() + p[f"{pre}.MLP_f.6.bias"]     # score head: fp32 VALU on the fp32 accumulators
    return s.squeeze(-1)


def cmi_logits_q(p, name, batch, rnd):
    """MLP_For_CMI's tower (Model.py:47-72: 384-256-256-256-2, ReLU) before the +-10 clamp, with the kernels' operand rounding."""
    pre = f"vcmi_estimator_{name}.classifier.mlp"
    h = batch
    for j, i in enumerate((0, 2, 4, 6)):
        h = mm(h, p[f"{pre}.{i}.weight"], rnd) + p[f"{pre}.{i}.bias"]
        if j < 3:
            h = F.relu(h)
    return h


def cmi_terms_q(p, name, batch, rnd, last_act="sigmoid"):
    """-> (logits, bce, cmi) of one VCMIEstimator on an assembled batch [2n, 384] (joint rows first; Model.py:185-219)."""
    n = batch.shape[0] // 2
    logits = cmi_logits_q(p, name, batch, rnd)
    h = torch.clamp(logits, -10, 10)
    out = torch.sigmoid(h) if last_act == "sigmoid" else F.hardtanh(h, 1e-4, 1 - 1e-4)
    target = torch.zeros(2 * n, 2, dtype=batch.dtype)
    target[:n, 0] = 1.0
    target[n:, 1] = 1.0
    bce = F.binary_cross_entropy(out, target)
    gamma = out[:, 0]
    lr = torch.log(gamma / (1 - gamma + 1e-6))
    cmi = 1.0 + lr[:n].sum() / (2 * n) - lr[n:].sum() / (2 * n)
    return logits, bce, cmi


MI_WIRE = {"f_t": (0, 1), "f_a": (0, 2), "f_v": (0, 3), "t_a": (1, 2), "t_v": (1, 3)}     # Model.py:313-319 (F,T,A,V slots)


def mi_terms_q(p, opt, feats, rnd):
    """The five InfoNCE values (Model.py:313-319, VMI.py:162-166) and score matrices on feats [4,B,128]."""
    out, sc = [], []
    for n in R.VMI_NAMES:
        ix, iy = MI_WIRE[n]
        s = critic_scores_q(p, n, opt.critic_type, feats[ix], feats[iy], rnd)
        sc.append(s)
        out.append(R.infonce_lower_bound(s))
    return out, sc


# --------------------------------------------------------------------------------------
# encoders: W_t projection + 2-layer bi-GRU (gemm_fast_f16 + gru.hip) with the kernels' rounding points
# --------------------------------------------------------------------------------------
class _GruDirQ(torch.autograd.Function):
    """One direction of one nn.GRU layer on the hoisted input projection gx = x W_ih^T + b_ih, forward AND backward written out as
    gru.hip computes them (Model.py:254-255,441-447 packed semantics), with ``rq`` at the kernels' rounding points:
      gru_fwd_kernel<bf16>   per cell step the fp32 state h and W_hh are rounded for the product h W_hh^T (fp32 accumulate, b_hh added
                             in fp32); gates in fp32; the state itself stays fp32; the gates saved for BPTT -- r, z, n and W_hn h + b_hn --
                             are stored rounded.
      gru_bwd_kernel<bf16>   dh = dout + carry; dn, dz, dn', dz', dr' from the ROUNDED saved gates and the fp32 h_prev; the carry product
                             [dr'|dz'|dn' r] W_hh with both operands rounded; dW_hh = sum_t rq(dgh_t)^T rq(h_prev_t) (bf16-stored BPTT
                             outputs feed the weight-gradient GEMM); db_hh = sum_t dgh_t from the un-rounded fp32 values (in-kernel
                             sums); the gradient w.r.t. gx (= [dr'|dz'|dn']) leaves un-rounded -- its consumers (dW_ih, the gradient to
                             the layer below, both products of `mm`) round it themselves, the same value the bf16-stored dg holds.
    With rq = identity this IS autograd of oracle.gru_direction (tests/test_rounded_ref.py)."""
    @staticmethod
    def forward(ctx, gx, w_hh, b_hh, lengths, reverse, rq, qd=identity, qh=identity):
        """qd / qh (round 5b, MIMRL_REC16): storage rounding of the BPTT's two streamed operands -- dout as its producer stored it (bf16: the dh0
        product's epilogue / the LayerNorm backward) and h_prev as the forward kernel's fp16 copy of its outputs (layer 0)."""
        B, T, G = gx.shape
        H = G // 3
        wq = rq(w_hh)
        h = gx.new_zeros(B, H)
        out = gx.new_zeros(B, T, H)
        gates = gx.new_zeros(B, T, 4, H)
        order = list(range(T - 1, -1, -1)) if reverse else list(range(T))
        for t in order:
            gh = rq(h) @ wq.t() + b_hh
            r = torch.sigmoid(gx[:, t, :H] + gh[:, :H])
            z = torch.sigmoid(gx[:, t, H:2 * H] + gh[:, H:2 * H])
            hn = gh[:, 2 * H:]
            n = torch.tanh(gx[:, t, 2 * H:] + r * hn)
            hnew = n + z * (h - n)
            valid = (lengths > t).unsqueeze(1)
            out[:, t] = torch.where(valid, hnew, torch.zeros_like(hnew))
            h = torch.where(valid, hnew, h)
            gates[:, t] = torch.stack([r, z, n, hn], 1)
        ctx.save_for_backward(rq(gates), out, wq, lengths)
        ctx.meta = (order, rq, qd, qh)
        return out

    @staticmethod
    def backward(ctx, dout):
        gates, out, wq, lengths = ctx.saved_tensors
        order, rq, qd, qh = ctx.meta
        B, T, H = out.shape
        carry = out.new_zeros(B, H)
        dgx = out.new_zeros(B, T, 3 * H)
        dw = torch.zeros_like(wq)
        db = out.new_zeros(3 * H)
        for k in range(len(order) - 1, -1, -1):
            t = order[k]
            valid = (lengths > t).unsqueeze(1)
            if k > 0:
                tp = order[k - 1]
                hp_ok = valid & (lengths > tp).unsqueeze(1)
                hp = torch.where(hp_ok, qh(out[:, tp]), torch.zeros_like(carry))
            else:
                hp = torch.zeros_like(carry)
            r, z, n, hn = gates[:, t, 0], gates[:, t, 1], gates[:, t, 2], gates[:, t, 3]
            dh = qd(dout[:, t]) + carry
            dn = dh * (1 - z)
            dz = dh * (hp - n)
            dnp = dn * (1 - n * n)
            dzp = dz * z * (1 - z)
            drp = dnp * hn * r * (1 - r)
            dnr = dnp * r
            zero = torch.zeros_like(dh)
            drp, dzp, dnp, dnr = (torch.where(valid, x, zero) for x in (drp, dzp, dnp, dnr))
            dhz = torch.where(valid, dh * z, carry)
            dgh = torch.cat([drp, dzp, dnr], 1)
            dgx[:, t] = torch.cat([drp, dzp, dnp], 1)
            carry = dhz + rq(dgh) @ wq
            dw = dw + rq(dgh).t() @ rq(hp)
            db = db + dgh.sum(0)
        return dgx, dw, db, None, None, None, None, None


def bigru2_q(p, prefix, x, lengths, rnd, rq, gxq=identity, rec16=False):
    """oracle.bigru2 with the kernels' rounding: hoisted projections through ``mm`` (``rnd``: fp16 forward operands, bf16 gradient
    operands), recurrences through _GruDirQ (``rq``: bf16); ``gxq``: storage rounding of the projection's OUTPUT gx (fp16 for long
    sequences, engine_abi.hip gx_f16; straight-through: the BPTT's gradient w.r.t. gx does not see it)."""
    inp = x
    for layer in range(2):
        outs = []
        for rev, sfx in ((False, ""), (True, "_reverse")):
            gx = ste(gxq, mm(inp, p[f"{prefix}.weight_ih_l{layer}{sfx}"], rnd) + p[f"{prefix}.bias_ih_l{layer}{sfx}"])
            # rec16 (engine default in the bf16 BPTT mode): dout of both layers is stored as bf16, h_prev of layer 0 comes from the fp16 copy
            qd, qh = (r_bf16, r_f16 if layer == 0 else identity) if rec16 else (identity, identity)
            outs.append(_GruDirQ.apply(gx, p[f"{prefix}.weight_hh_l{layer}{sfx}"], p[f"{prefix}.bias_hh_l{layer}{sfx}"], lengths, rev, rq, qd, qh))
        inp = torch.cat(outs, dim=-1)
    H = inp.shape[-1] // 2
    return inp[..., :H] + inp[..., H:]


def encoders_q(p, opt, t_feat, a, v, rnd, rq, gxq=identity, rec16=False):
    """Model.forward up to the stacked cube input (Model.py:395-475; oracle.model_forward lines 187-206), dropout 0.
    -> (x [B,L,3,D], T_F, A_F, V_F)."""
    D, L = opt.d_common, opt.time_len
    t = mm(t_feat, p["W_t.weight"], rnd)
    la, lv = R.infer_lengths(a), R.infer_lengths(v)
    ah = F.relu(F.layer_norm(bigru2_q(p, "rnn_a", a, la, rnd, rq, gxq, rec16), (D,), p["ln_a.weight"], p["ln_a.bias"], 1e-6))
    vh = F.relu(F.layer_norm(bigru2_q(p, "rnn_v", v, lv, rnd, rq, gxq, rec16), (D,), p["ln_v.weight"], p["ln_v.bias"], 1e-6))
    pad = lambda y: F.pad(y, (0, 0, 0, L - y.shape[1]))
    return torch.stack([pad(t), pad(ah), pad(vh)], dim=2), t.mean(1), ah.mean(1), vh.mean(1)
