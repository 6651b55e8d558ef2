"""-m gpu: the bf16-only FUSED kernel families of the benchmarked mode against the oracle.

cube_fwd_fused_kernel, daxis / kmix / laxis_bwd (+ their weight-gradient GEMMs), concat_fwd / concat_bwd_kernel and the
mlp_img8 + mi_sep_nce pair have no fp32 mode, so the fp32 parity suite never runs them.  Here they are driven through the C ABI
probes (include/mimrl.h: mimrl_probe_cube / mimrl_probe_mi -- the engine's own code path) and compared, output AND every parameter /
input gradient, with float64 autograd of the oracle evaluated on operands rounded to bf16 at exactly the points the kernels round
(tests/rounded_ref.py, which is the oracle itself when the rounding hook is the identity: tests/test_rounded_ref.py).
What is left between the two is fp32 accumulation order, bf16 rounding of the BACKWARD operands (dY, dU: 2^-9 relative per
operand) and rounding-boundary flips.  Bands are 3x the measured error (recorded in gpurun_out/fused_oracle_errors.json)."""
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from mimrl_amd.engine import HipEngine
from oracle import mimrl_ref as R
from tests import rounded_ref as Q
from tests.helpers import case, oracle_params

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_LOG = {}


def _record(key, val):
    _LOG[key] = val
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        json.dump(_LOG, open(os.path.join(ROOT, "gpurun_out", "fused_oracle_errors.json"), "w"), indent=1, sort_keys=True)
    except OSError:
        pass


def perturbed_params(opt, seed):
    """Portable initialisation + N(0, 0.02) on every bias (the reference zero-initialises the critic biases, VMI.py:47-51: with
    zeros a wrong bias path would go unnoticed)."""
    p = oracle_params(opt, seed, torch.float64)
    g = torch.Generator().manual_seed(1000 + seed)
    return {k: (v + 0.02 * torch.randn(v.shape, generator=g, dtype=torch.float64) if k.endswith(".bias") else v) for k, v in p.items()}


def errs(got, want):
    got = np.asarray(got, np.float64); want = np.asarray(want, np.float64)
    scale = np.abs(want).max() + 1e-30
    return {"scale": float(scale), "max_rel_scale": float(np.abs(got - want).max() / scale), "l2_rel": float(np.linalg.norm(got - want) / (np.linalg.norm(want) + 1e-30)),
            "cos": float((got * want).sum() / (np.linalg.norm(got) * np.linalg.norm(want) + 1e-30))}


def tensor_ok(e, tol, group_scale, chaos=0.0):
    """One gradient tensor against the rounded-operand float64 reference: max-norm error relative to the tensor's scale within ``tol``, or
    within 3 x the reference's OWN chaos (chaos_floor) where that is larger.  Round 4: the third way out of round 3 -- L2 within tol
    and no entry beyond 15 x tol -- is gone, and the separable-critic test point was moved off the kinks (critics after ten oracle updates
    on real features, compared WITHOUT any chaos band), so that a 10-30 % gradient bug cannot hide.
    A tensor whose own scale is below 2 % of the largest gradient of the probe (a sum that cancels, e.g. the last-layer bias under
    InfoNCE's shift invariance) is held to ``tol`` of THAT scale: it is noise in both implementations."""
    scale = max(e["scale"], 0.02 * group_scale)
    mx = e["max_rel_scale"] * e["scale"] / scale
    return mx <= max(tol, 3.0 * chaos * e["scale"] / scale)


def chaos_floor(g_a, g_b):
    """(iii) Rounding is discontinuous, so the rounded-operand reference is itself chaotic: evaluated with every bias moved by JITTER = 1e-6
    (a few fp32 ulps of an O(1) pre-activation -- what two correct fp32-accumulating implementations differ by; biases are added AFTER the rounded products, so the nudge reaches the next rounding
    instead of being rounded away with the inputs) its gradient tensors move by `chaos` of their scale -- wherever a sum cancels
    (bias / LayerNorm-parameter gradients, towers fed by non-negative features) that is percents.  No implementation can be closer to
    the reference than the reference is to itself: the band of a tensor is max(tol, 3 x chaos)."""
    return [float((a - b).abs().max() / (a.abs().max() + 1e-30)) for a, b in zip(g_a, g_b)]


def make(name, monkeypatch, env=(), ln_first=False):
    for k in env:
        monkeypatch.setenv(k, "1")
    c, opt, batch, banks = case(name)
    opt.ln_first = bool(ln_first)
    eng = HipEngine(opt, 768, 74, 35, seq_len=c["T"], bank_capacity=c["N"], precision="bf16")
    p = perturbed_params(opt, c["seed"])
    eng.load_params(p)
    return c, opt, p, eng


# (name, fused kernels?, env, ln_first).  tiny_odd: L = 8 (zero-padded tile), blocks 8-3-128=3-3-128, B = 12.  ln_first (MLPProcess.py:64-92)
# only exists as the unfused bf16 GEMM chain; it is tested at cfg1's shape -- at the tiny fixtures' L = 6 / 4 the bf16 gradient of this
# block is noise (float64 experiment: rounded-vs-exact dx differs by 35-97 % there).
CUBE = [("cfg1_sep", True, (), False), ("cfg2_sep", True, (), False), ("tiny_odd", True, (), False), ("cfg1_sep", False, (), True),
        ("cfg1_sep", False, ("MIMRL_NO_FUSED_CUBE", "MIMRL_NO_FUSED_CUBE_BWD"), False)]


@pytest.mark.parametrize("name", ["cfg3_small", "cfg5_small"])
def test_long_sequence_laxis_backward_kernel_matches_the_gemm_chain(name, monkeypatch):
    """Round 5b: the L-axis BACKWARD of block 0 at L = 500 / 1000 is the LONG instantiation of laxis_bwd_kernel (csrc/cube_bwd_fused.hip: phases 1 / 2
    of the short kernel, then dX = W1^T dU + Wr^T dY walked in tiles of 64 input rows) instead of colln_bwd + two GEMM launches
    (MLPProcess.py:95-104 under autograd).  Same operands and roundings (bf16) as the chain up to the accumulation order: dx and every parameter
    gradient within 2e-3 in L2 of MIMRL_LAXIS_BWD_LONG=0, outputs (the forward is untouched) bit-identical."""
    out = {}
    for tag, env in (("long", None), ("chain", "0")):
        if env:
            monkeypatch.setenv("MIMRL_LAXIS_BWD_LONG", env)
        else:
            monkeypatch.delenv("MIMRL_LAXIS_BWD_LONG", raising=False)
        c, opt, p, eng = make(name, monkeypatch)
        B, L = c["B"], opt.time_len
        g = torch.Generator().manual_seed(11)
        x = torch.randn(B, L, 3, 128, generator=g, dtype=torch.float64)
        x[:, :, 1:] = torch.relu(x[:, :, 1:])
        ol, ok = opt.d_outs[-1][0], opt.d_outs[-1][1]
        dout = torch.randn(B, ol, ok, 128, generator=g, dtype=torch.float64)
        o, dx = eng.probe_cube(x, dout)
        torch.cuda.synchronize()
        out[tag] = {"out": o.double().cpu().numpy().copy(), "dx": dx.double().cpu().numpy().copy(),
                    **{n: v.double().cpu().numpy().copy() for n, v in eng.grads.items() if n.startswith("mlp_encoder.")}}
        eng.close()
    assert L > 64 and np.array_equal(out["long"]["out"], out["chain"]["out"])
    assert not np.array_equal(out["long"]["dx"], out["chain"]["dx"]), "the LONG instantiation did not run"
    rec = {}
    for n, want in out["chain"].items():
        rec[n] = float(np.linalg.norm(out["long"][n] - want) / max(np.linalg.norm(want), 1e-30))
        assert np.isfinite(out["long"][n]).all() and rec[n] <= 2e-3, (n, rec[n])
    _record(f"laxis_bwd_long_vs_chain/{name}", dict(sorted(rec.items(), key=lambda kv: -kv[1])[:8]))


@pytest.mark.parametrize("name", ["cfg3_small", "cfg5_small"])
def test_long_sequence_laxis_kernel_matches_the_gemm_chain(name, monkeypatch):
    """Round 5b: the L-axis MLP of CubeMLP block 0 at L = 500 / 1000 (cfg3 / cfg5: too long for the LDS-resident block kernel) is ONE launch that
    reads the [L, 384] slab of every sample once (csrc/cube_long.hip: [W1; Wr] stacked as one MFMA left operand, the second product straight from
    the accumulators, LayerNorm over the output rows in registers) instead of the GEMM chain's two passes + padded-copy + LayerNorm launches
    (MLPProcess.py:95-104).  Same saved activations, unchanged backward.  The kernel rounds the operands of Wr . X to fp16 like every forward product
    (the chain's dual-product launch rounded them to bf16), so it is not the chain bit for bit; both are held to the UN-ROUNDED float64 oracle
    (outputs and autograd): the one-pass kernel may be at most 1.25x as far from it as the chain in any tensor -- measured: closer (1.1e-2 against
    2.7e-2 of the output scale at cfg5's shape) -- and the two paths differ from each other by no more than the sum of those distances."""
    out = {}
    for tag, env in (("long", None), ("chain", "0")):
        if env:
            monkeypatch.setenv("MIMRL_LAXIS_LONG", env)
        else:
            monkeypatch.delenv("MIMRL_LAXIS_LONG", raising=False)
        c, opt, p, eng = make(name, monkeypatch)
        B, L = c["B"], opt.time_len
        g = torch.Generator().manual_seed(11)
        x = torch.randn(B, L, 3, 128, generator=g, dtype=torch.float64)
        x[:, :, 1:] = torch.relu(x[:, :, 1:])
        ol, ok = opt.d_outs[-1][0], opt.d_outs[-1][1]
        dout = torch.randn(B, ol, ok, 128, generator=g, dtype=torch.float64)
        o, dx = eng.probe_cube(x, dout)
        torch.cuda.synchronize()
        out[tag] = {"out": o.double().cpu().numpy().copy(), "dx": dx.double().cpu().numpy().copy(),
                    **{n: v.double().cpu().numpy().copy() for n, v in eng.grads.items() if n.startswith("mlp_encoder.")}}
        if tag == "long":
            names = [n for n in p if n.startswith("mlp_encoder.")]
            leaves = {n: p[n].clone().requires_grad_(True) for n in names}
            xr = x.clone().requires_grad_(True)
            ex = R.cube_mlp({**p, **leaves}, opt, xr)
            gex = torch.autograd.grad((ex * dout).sum(), [xr] + [leaves[n] for n in names])
            exact = {"out": ex.detach().numpy(), "dx": gex[0].numpy(), **{n: t.numpy() for n, t in zip(names, gex[1:])}}
        eng.close()
    assert L > 64
    assert not np.array_equal(out["long"]["out"], out["chain"]["out"]), "the one-pass kernel did not run"
    rec = {}
    for n, want in exact.items():
        nrm = max(np.linalg.norm(want), 1e-30)
        el, ec = np.linalg.norm(out["long"][n] - want) / nrm, np.linalg.norm(out["chain"][n] - want) / nrm
        d = np.linalg.norm(out["long"][n] - out["chain"][n]) / nrm
        rec[n] = {"long_vs_exact": float(el), "chain_vs_exact": float(ec), "long_vs_chain": float(d)}
        # (the 3- and 10-element LayerNorm / K-axis tensors are sums of terms that cancel to a few per cent: noise in either path, 14-20 % off the
        #  oracle; everything with a real signal -- out, dx, the weight matrices -- is within a few per cent and the one-pass kernel is not worse)
        slack = 1e-3 if want.size >= 1024 else 0.1
        assert np.isfinite(out["long"][n]).all() and el <= 1.25 * ec + slack and d <= el + ec + 1e-6, (n, rec[n])
    _record(f"laxis_long_vs_chain/{name}", rec)


@pytest.mark.parametrize("name,fused,env,ln_first", CUBE, ids=[f"{n}-{'fused' if f else 'chain'}{'-ln_first' if l else ''}" for n, f, _, l in CUBE])
@pytest.mark.parametrize("upstream", ["random", "broadcast"])
def test_cube_stack_vs_rounded_oracle(name, fused, env, ln_first, upstream, monkeypatch):
    """MLPEncoder forward + autograd (MLPProcess.py:94-137).  `broadcast` = the gradient the head really sends: one [B,128] vector
    broadcast over (l, k) (Model.py:489-502 means), which LayerNorm cancels to ~10 % per axis -- the ill-conditioned case."""
    c, opt, p, eng = make(name, monkeypatch, env, ln_first)
    B, L = c["B"], opt.time_len
    g = torch.Generator().manual_seed(11)
    # the cube's real input: text projection / relu(LN(gru)) slots, O(1) entries with a non-negative half
    x = torch.randn(B, L, 3, 128, generator=g, dtype=torch.float64)
    x[:, :, 1:] = torch.relu(x[:, :, 1:])
    ol, ok = opt.d_outs[-1][0], opt.d_outs[-1][1]
    if upstream == "random":
        dout = torch.randn(B, ol, ok, 128, generator=g, dtype=torch.float64)
    else:
        dout = (torch.randn(B, 1, 1, 128, generator=g, dtype=torch.float64) / (ol * ok)).expand(B, ol, ok, 128).contiguous()
    out, dx = eng.probe_cube(x, dout)
    torch.cuda.synchronize()
    names = [n for n in p if n.startswith("mlp_encoder.")]
    rnd = Q.F16_FWD if fused else Q.BF16

    def reference(xin, jitter=0.0):
        leaves = {n: (p[n] + jitter * torch.randn(p[n].shape, generator=g, dtype=torch.float64) if n.endswith(".bias") else p[n].clone()).requires_grad_(True)
                  for n in names}
        xr = xin.clone().requires_grad_(True)
        out_ = Q.cube_mlp_q({**p, **leaves}, opt, xr, rnd, round_tile=fused)
        return out_.detach(), torch.autograd.grad((out_ * dout).sum(), [xr] + [leaves[n] for n in names])

    ref, gr = reference(x)
    _, gr2 = reference(x, JITTER)
    chaos = dict(zip(["dx"] + names, chaos_floor(gr, gr2)))
    exact = R.cube_mlp(p, opt, x)                                   # the un-rounded oracle: how far bf16 itself is from fp32/fp64
    key = f"cube/{name}/{'fused' if fused else 'chain'}{'-ln_first' if ln_first else ''}/{upstream}"
    rec = {"out": errs(out.cpu(), ref), "out_vs_unrounded_oracle": errs(out.cpu(), exact), "dx": errs(dx.cpu(), gr[0])}
    rec["dx"]["chaos"] = chaos["dx"]
    worst = ("dx", rec["dx"]["max_rel_scale"])
    for n, gw in zip(names, gr[1:]):
        e = errs(eng.grads[n].cpu(), gw)
        e["chaos"] = chaos[n]
        rec[n] = e
        if e["max_rel_scale"] > worst[1]:
            worst = (n, e["max_rel_scale"])
    rec["worst_grad"] = worst
    _record(key, rec)
    eng.close()
    assert rec["out"]["max_rel_scale"] <= (6e-3 if fused else 1e-2), (key, rec["out"])
    gs = max(e["scale"] for n, e in rec.items() if isinstance(e, dict) and n.startswith("mlp_encoder."))
    # (the UNFUSED bf16 chain under the broadcast gradient -- not a path bench.py times -- is the one case beyond 3 x chaos: ln_d.weight of
    #  block 0 at 5.0e-2 with chaos 0.9e-2, ten more tensors at 1-3 %; it gets an explicit 6e-2 instead of the old blanket 15 x tol)
    tol = 6e-2 if (not fused and upstream == "broadcast" and not ln_first) else TOL
    bad = {n: e for n, e in rec.items() if isinstance(e, dict) and (n == "dx" or n.startswith("mlp_encoder.")) and
           not tensor_ok(e, tol, gs if n != "dx" else e["scale"], e["chaos"])}
    assert not bad, (key, {n: (e["max_rel_scale"], e["l2_rel"], e["chaos"]) for n, e in bad.items()})


# Measured (gpurun_out/fused_oracle_errors.json, round 3; max-norm error / tensor scale, against the rounded-operand reference):
#   concat critic, fused or chain backward: <= 2.0e-3 (every tensor);  separable critic: 69 of 80 tensors <= 1e-2, the h towers fed by the
#   non-negative A_F / V_F up to 9e-2 (reference chaos there: same size);  CubeMLP fused at cfg2: 41 of 43 tensors <= 6e-3, ln_d.weight
#   1.3e-2, ln_k.weight (3 scalars, each the residual of a 2.5-M-term cancelling sum) 2.8e-2;  cfg1 (4x fewer terms): <= 1.5e-2 / 3.3e-2.
JITTER = 1e-6
TOL = 1e-2           # per-tensor gradient error <= 1e-2 of the tensor's scale (or 4 x the reference's own chaos)

MI = [("cfg1_cat", 1), ("cfg1_cat", 2), ("cfg2_cat", 1), ("cfg2_cat", 2), ("cfg3_small", 2), ("cfg2_sep", 1), ("cfg2_sep", 2), ("tiny_odd", 1)]


@pytest.mark.parametrize("name,stage", MI, ids=[f"{n}-s{s}" for n, s in MI])
def test_mi_estimators_vs_rounded_oracle(name, stage, monkeypatch):
    """Critic forward (VMI.py:53-69) + InfoNCE (VMI.py:162-166) + backward: stage 1 = every vmi_estimator_* parameter gradient,
    stage 2 = the gradient w.r.t. both operands of every estimator.  cfg2_cat = cfg2 with the concat critic (B = 128: the fused
    concat backward; B = 32 / 16: fused forward + GEMM-chain backward)."""
    base = name.replace("cfg2_cat", "cfg2_sep")
    for k in ():
        monkeypatch.setenv(k, "1")
    c, opt, batch, banks = case(base)
    if name == "cfg2_cat":
        opt.critic_type = "concat"
    eng = HipEngine(opt, 768, 74, 35, seq_len=c["T"], bank_capacity=c["N"], precision="bf16")
    p = perturbed_params(opt, c["seed"])
    eng.load_params(p)
    B = c["B"]
    g = torch.Generator().manual_seed(5)
    names = [n for n in p if n.startswith("vmi_estimator_")]
    if opt.critic_type == "separate" and B >= 32:
        # Test point (round 4).  At initialisation InfoNCE ~ 0, the softmax over the scores is uniform and the tower gradients are what is
        # left of a near-total cancellation: the rounded-operand reference itself moves by up to 20 % of a tensor's scale under a 1e-6 nudge
        # of the biases (72 of 80 tensors above 1 %), so round 3's band for the h towers of f_a / f_v was 3 x 12 %.  The REAL features of the
        # fixture batch (oracle forward pass) and critics after ten float64 oracle Adam steps on them (InfoNCE 0.01 ... 0.34, lr 4e-3) are a
        # well-conditioned point: the kernels agree with the reference to 1.6e-3 there and the test needs no chaos band (below).
        with torch.no_grad():
            _, F_F, T_F, A_F, V_F = R.model_forward(p, opt, *[b.double() for b in batch[:3]])
        feats = torch.stack([F_F, T_F, A_F, V_F])
        adam = R.AdamState(p, names)
        for _ in range(10):
            leaves = {n: p[n].clone().requires_grad_(True) for n in names}
            obj = sum(-R.infonce_lower_bound(R.critic_scores({**p, **leaves}, n, opt.critic_type, feats[Q.MI_WIRE[n][0]], feats[Q.MI_WIRE[n][1]]))
                      for n in R.VMI_NAMES)
            gs = torch.autograd.grad(obj, [leaves[n] for n in names], allow_unused=True)
            adam.step(p, {n: gw for n, gw in zip(names, gs) if gw is not None}, 4e-3)
        eng.load_params(p)
    else:
        feats = 0.25 * torch.randn(4, B, 128, generator=g, dtype=torch.float64)      # the scale of real features (means over T of O(1) rows)
        feats[2:] = feats[2:].abs()                                                    # A_F, V_F are means of ReLU outputs
    r = eng.probe_mi(stage, feats)
    torch.cuda.synchronize()
    k1, k2 = opt.loss_mi_coefficient1, opt.loss_mi_coefficient2
    gsc = [-k1[e] for e in range(5)] if stage == 1 else [-k2[0], -k2[1], -k2[2], -k2[3], -k2[3]]

    def reference(fin, jitter=0.0):
        leaves = {n: (p[n] + jitter * torch.randn(p[n].shape, generator=g, dtype=torch.float64) if n.endswith(".bias") else p[n].clone()).requires_grad_(True)
                  for n in names}
        ops, mis_, scores_ = [], [], []                                            # separate leaves per (estimator, operand)
        for e, n in enumerate(R.VMI_NAMES):
            ix, iy = Q.MI_WIRE[n]
            xo, yo = fin[ix].clone().requires_grad_(True), fin[iy].clone().requires_grad_(True)
            ops += [xo, yo]
            sc = Q.critic_scores_q({**p, **leaves}, n, opt.critic_type, xo, yo, Q.BF16)
            scores_.append(sc)
            mis_.append(R.infonce_lower_bound(sc))
        obj = sum(gsc[e] * mis_[e] for e in range(5))
        if stage == 1:
            gr_ = torch.autograd.grad(obj, [leaves[n] for n in names], allow_unused=True)
            gr_ = [torch.zeros_like(leaves[n]) if gw is None else gw for n, gw in zip(names, gr_)]
        else:
            gr_ = torch.autograd.grad(obj, ops)
        return [m.detach() for m in mis_], [sc.detach() for sc in scores_], gr_

    mis, scores, gr = reference(feats)
    _, _, gr2 = reference(feats, JITTER)
    chaos = chaos_floor(gr, gr2)
    key = f"mi/{name}/stage{stage}"
    rec = {"mi": errs(r["mi"].cpu(), torch.stack(mis)), "mi_values": [float(m) for m in mis]}
    if r["scores"] is not None:
        rec["scores"] = errs(r["scores"].cpu(), torch.stack(scores))
    worst = ("-", 0.0)
    if stage == 1:
        got = [eng.grads[n].cpu() for n in names]
        labels = names
    else:
        got = list(r["dtin"].cpu().reshape(10, B, 128))
        labels = [f"dtin[{R.VMI_NAMES[i // 2]}][{'xy'[i % 2]}]" for i in range(10)]
    for n, gg, gw, ch in zip(labels, got, gr, chaos):
        e = errs(gg, gw)
        e["chaos"] = ch
        rec[n] = e
        if e["max_rel_scale"] > worst[1]:
            worst = (n, e["max_rel_scale"])
    rec["worst_grad"] = worst
    _record(key, rec)
    eng.close()
    np.testing.assert_allclose(r["mi"].cpu().numpy(), [float(m) for m in mis], rtol=2e-3, atol=2e-5)
    if r["scores"] is not None:
        assert rec["scores"]["max_rel_scale"] <= 3e-3, (key, rec["scores"])
    gt = {n: e for n, e in rec.items() if isinstance(e, dict) and n not in ("mi", "scores")}
    gs = max(e["scale"] for e in gt.values())
    # trained test point: NO chaos band at all -- every tensor within 1e-2 of its scale (measured: <= 1.6e-3 for the 70 tensors above 2 % of
    # the probe's largest gradient, <= 7.7e-3 for the cancelling last-layer biases)
    trained = opt.critic_type == "separate" and B >= 32
    bad = {n: e for n, e in gt.items() if not tensor_ok(e, TOL, gs, 0.0 if trained else e["chaos"])}
    assert not bad, (key, {n: (e["max_rel_scale"], e["l2_rel"], e["chaos"]) for n, e in bad.items()})


@pytest.mark.parametrize("name,stage", [("cfg2_sep", 1), ("cfg2_sep", 2), ("cfg1_sep", 1), ("tiny_odd", 1), ("tiny_odd", 2)],
                         ids=lambda v: str(v))
def test_cmi_classifiers_vs_rounded_oracle(name, stage):
    """The six CMI classifiers (MLP_For_CMI, Model.py:47-72; BCE / CMI value, Model.py:185-219): mlp_img8_kernel<fwd, 6 chunks> with its
    384-wide first layer, cmi_loss_kernel, mlp_img8_kernel<bwd> with the narrow 2-logit top layer (its weight gradient is produced
    inside the data-gradient kernel) and the grouped weight-gradient GEMMs -- logits, both values, stage 1: every vcmi_estimator_*
    gradient, stage 2: the gradient w.r.t. the joint rows.  tiny_odd: 2n = 24 rows per classifier (ragged 32-row tiles), k = 3."""
    c, opt, batch, banks = case(name)
    eng = HipEngine(opt, 768, 74, 35, seq_len=c["T"], bank_capacity=c["N"], precision="bf16")
    p = perturbed_params(opt, c["seed"])
    eng.load_params(p)
    n = (c["B"] // opt.k_neighbor) * opt.k_neighbor
    g = torch.Generator().manual_seed(9)
    cin = 0.4 * torch.randn(6, 2 * n, 384, generator=g, dtype=torch.float64)
    cin[:, :, 128:256] = cin[:, :, 128:256].abs()          # feature operands of the real batches include non-negative ones (A_F, V_F)
    r = eng.probe_cmi(stage, cin)
    torch.cuda.synchronize()
    names = [k for k in p if k.startswith("vcmi_estimator_")]
    k1, k2 = opt.loss_mi_coefficient1, opt.loss_mi_coefficient2
    g2 = [-k2[5], k2[4] + k2[5] - k2[7], -k2[6], k2[4] + k2[6] - k2[7], -k2[4], -k2[4]]

    def reference(jitter=0.0):
        leaves = {k: (p[k] + jitter * torch.randn(p[k].shape, generator=g, dtype=torch.float64) if k.endswith(".bias") else p[k].clone()).requires_grad_(True)
                  for k in names}
        xs, lg, bces, cmis = [], [], [], []
        for e, nm in enumerate(R.VCMI_NAMES):
            xe = cin[e].clone().requires_grad_(True)
            xs.append(xe)
            l_, b_, c_ = Q.cmi_terms_q({**p, **leaves}, nm, xe, Q.BF16, opt.cmi_last_acticate)
            lg.append(l_.detach()); bces.append(b_); cmis.append(c_)
        if stage == 1:
            obj = sum(k1[5 + e] * bces[e] for e in range(6))
            gr_ = torch.autograd.grad(obj, [leaves[k] for k in names], allow_unused=True)
            gr_ = [torch.zeros_like(leaves[k]) if gw is None else gw for k, gw in zip(names, gr_)]
        else:
            obj = sum(g2[e] * cmis[e] for e in range(6))
            gr_ = [t[:n] for t in torch.autograd.grad(obj, xs)]
        return torch.stack(lg), [float(b) for b in bces], [float(c_) for c_ in cmis], gr_

    lg, bces, cmis, gr = reference()
    _, _, _, gr2 = reference(JITTER)
    chaos = chaos_floor(gr, gr2)
    key = f"cmi/{name}/stage{stage}"
    rec = {"logits": errs(r["logits"].cpu(), lg), "bce": [float(x) for x in r["bce"].cpu()], "cmi": [float(x) for x in r["cmi"].cpu()]}
    if stage == 1:
        got, labels = [eng.grads[k].cpu() for k in names], names
    else:
        got, labels = [r["dcin"][e, :n].cpu() for e in range(6)], [f"dcin[{nm}]" for nm in R.VCMI_NAMES]
    for lab, gg, gw, ch in zip(labels, got, gr, chaos):
        e_ = errs(gg, gw)
        e_["chaos"] = ch
        rec[lab] = e_
    _record(key, rec)
    eng.close()
    assert rec["logits"]["max_rel_scale"] <= 3e-3, (key, rec["logits"])
    np.testing.assert_allclose(rec["bce"], bces, rtol=2e-3, atol=1e-5)
    np.testing.assert_allclose(rec["cmi"], cmis, rtol=2e-3, atol=2e-4)
    gt = {k: v for k, v in rec.items() if isinstance(v, dict) and k != "logits"}
    gs = max(v["scale"] for v in gt.values())
    bad = {k: v for k, v in gt.items() if not tensor_ok(v, TOL, gs, v["chaos"])}
    assert not bad, (key, {k: (v["max_rel_scale"], v["l2_rel"], v["chaos"]) for k, v in bad.items()})


@pytest.mark.parametrize("name", ["cfg2_sep", "tiny_odd"])
def test_fragment_image_mlp_kernel_equals_the_staged_one(name, monkeypatch):
    """mlp_frag_kernel (round 3b: shapes as template parameters, weights from MFMA-fragment-order images) against mlp_img8_kernel / the
    4-wave kernel it replaced (MIMRL_MLP_NO_FRAG=1), through the engine's own estimator paths: CMI classifiers and MI towers, forward
    values and every gradient the probes return, stage 1 and stage 2, full and ragged (tiny_odd) row tiles.  Same products, same
    rounding points; what may differ is the order of the float atomics of the bias / weight-gradient accumulation."""
    c, opt, batch, banks = case(name)
    n = (c["B"] // opt.k_neighbor) * opt.k_neighbor
    g = torch.Generator().manual_seed(11)
    cin = 0.4 * torch.randn(6, 2 * n, 384, generator=g, dtype=torch.float64)
    out = {}
    for tag in ("frag", "img8"):
        if tag == "img8":
            monkeypatch.setenv("MIMRL_MLP_NO_FRAG", "1")
        else:
            monkeypatch.delenv("MIMRL_MLP_NO_FRAG", raising=False)
        eng = HipEngine(opt, 768, 74, 35, seq_len=c["T"], bank_capacity=c["N"], precision="bf16")
        p = perturbed_params(opt, c["seed"])
        eng.load_params(p)
        res = {}
        for stage in (1, 2):
            r = eng.probe_cmi(stage, cin)
            torch.cuda.synchronize()
            res[f"cmi{stage}/logits"] = r["logits"].double().cpu()
            if stage == 1:
                for k in p:
                    if k.startswith("vcmi_estimator_"):
                        res[f"cmi1/{k}"] = eng.grads[k].double().cpu().clone()
            else:
                res["cmi2/dcin"] = r["dcin"][:, :n].double().cpu()
        out[tag] = res
        eng.close()
    for k, a in out["frag"].items():
        b = out["img8"][k]
        scale = float(b.abs().max()) + 1e-30
        assert float((a - b).abs().max()) <= 2e-5 * scale + 1e-9, (k, float((a - b).abs().max()), scale)


# (name, T override, dcube kind).  cfg2_sep / cfg2_ragged = the bench shape (B = 128, T = 50; ragged: four batch rows of different lengths per
# recurrence workgroup); T = 49 / 1: the odd-T path of the BPTT kernel (an un-pipelined first step, round 4); cfg1: B = 32 (one batch row per
# recurrence workgroup)
# gxh: fp16-stored input projections (MIMRL_GX_F16=1: an opt-in path, slower at cfg3 with the present store pattern -- engine_abi.hip: mimrl_create)
ENC = [("cfg2_sep", None, True, False), ("cfg2_ragged", None, True, False), ("cfg2_ragged", 49, True, False), ("cfg1_ragged", None, True, False),
       ("cfg2_sep", 1, True, False), ("tiny_ragged", None, True, False), ("cfg2_sep", None, False, False), ("cfg2_ragged", None, False, False),
       ("cfg2_ragged", None, True, True), ("cfg2_ragged", 49, True, True)]


@pytest.mark.parametrize("name,T_", [("cfg2_sep", None), ("cfg2_sep", 49), ("cfg2_ragged", None)])
def test_fused_layer0_input_projection_matches_the_gemm(name, T_, monkeypatch):
    """Round 4: the layer-0 recurrence kernel computes x W_ih^T + b_ih itself (GruFwdArgs::xin_on: three fp16 k-steps per gate and cell step
    from the packed inputs, 8-wave kernel; the BPTT launch of that layer follows with the 8-wave layout of the saved-gate slab) instead of
    reading a gx the projection GEMM wrote.  Same operands, same roundings, another accumulation order: the encoder outputs and every
    gradient agree with the GEMM path (MIMRL_NO_XIN=1) -- outputs to 2e-3 of their scale, gradients to 1e-2 in L2 (single elements: ReLU-kink flips; both paths
    pass the rounded-operand oracle test above at 2e-3 with its margin filter) -- and are not bit-identical (so the test knows the fused path really ran)."""
    c, opt, batch, banks = case(name)
    T = c["T"] if T_ is None else T_
    batch = tuple(b[:, :T] if b.dim() == 3 else b for b in batch)
    g = torch.Generator().manual_seed(3)
    dcube = torch.randn(c["B"], opt.time_len, 3, 128, generator=g) / T
    out = {}
    for tag, env in (("xin", None), ("gemm", "1")):
        if env:
            monkeypatch.setenv("MIMRL_NO_XIN", env)
        else:
            monkeypatch.delenv("MIMRL_NO_XIN", raising=False)
        eng = HipEngine(opt, 768, 74, 35, seq_len=T, bank_capacity=c["N"], precision="bf16")
        eng.load_params(perturbed_params(opt, c["seed"]))
        eng.set_batch(*batch)
        x = eng.probe_encoders(dcube)
        torch.cuda.synchronize()
        out[tag] = (x.double().cpu().numpy().copy(), {n: v.double().cpu().numpy().copy() for n, v in eng.grads.items() if n.startswith(("W_t", "rnn_", "ln_a", "ln_v"))})
        eng.close()
    xa, xb = out["xin"][0], out["gemm"][0]
    assert not np.array_equal(xa, xb), "the fused projection did not run"
    assert np.abs(xa - xb).max() <= 2e-3 * np.abs(xb).max()
    top = max(np.abs(v).max() for v in out["gemm"][1].values())
    for n, want in out["gemm"][1].items():
        scale = max(np.abs(want).max(), 1e-2 * top)
        # (no margin filter here: a last-bit difference in a LayerNorm input flips a ReLU unit now and then, one unit's whole contribution --
        #  7.8e-3 of the scale measured in rnn_v.weight_ih_l0, the deepest tensor; the oracle test above filters those units out)
        #  and 3.4e-2 in rnn_v.weight_ih_l1 of the ragged fixture -- hence an L2 criterion with a loose cap on single elements)
        diff = out["xin"][1][n] - want
        assert np.linalg.norm(diff) <= 1e-2 * max(np.linalg.norm(want), 1e-2 * top * np.sqrt(want.size)), (n, np.linalg.norm(diff) / np.linalg.norm(want))
        assert np.abs(diff).max() <= 1e-1 * scale, (n, np.abs(diff).max() / scale)


@pytest.mark.parametrize("name,T_", [("cfg2_sep", None), ("cfg2_sep", 49), ("cfg2_ragged", None)])
def test_16bit_stored_projection_operands_change_nothing(name, T_, monkeypatch):
    """Round 4: the layer-1 GRU input projection reads its operands as STORED fp16 (the layer-0 recurrence writes an fp16 copy of h next to the
    fp32 one, the layer-0 pack launch an fp16 image of W_ih_l1: gemm_fast_f16s_kernel) and the dh0 product reads W_ih_l1 from a bf16 image
    -- half the L2 -> LDS bytes of two GEMMs on the step's chain.  The fp32-operand kernels rounded the same values with the same
    conversions at every load, so NOTHING may change: the encoder outputs (no atomics on that path) bit for bit, every gradient within the
    run-to-run band of the float atomics in the weight-gradient GEMMs.  (MIMRL_NO_H16=1: the fp32-operand kernels.)"""
    c, opt, batch, banks = case(name)
    T = c["T"] if T_ is None else T_
    batch = tuple(b[:, :T] if b.dim() == 3 else b for b in batch)
    g = torch.Generator().manual_seed(3)
    dcube = torch.randn(c["B"], opt.time_len, 3, 128, generator=g) / T
    monkeypatch.setenv("MIMRL_NO_XIN", "1")      # (the fused layer-0 projection accumulates in another order: its own test below)
    monkeypatch.setenv("MIMRL_REC16", "0")       # (round 5b: bf16 dh0 / fp16 h_prev exist with the 16-bit operands only and DO change roundings: own test below)
    out = {}
    for tag, env in (("h16", None), ("fp32", "1")):
        if env:
            monkeypatch.setenv("MIMRL_NO_H16", env)
        else:
            monkeypatch.delenv("MIMRL_NO_H16", raising=False)
        eng = HipEngine(opt, 768, 74, 35, seq_len=T, bank_capacity=c["N"], precision="bf16")
        eng.load_params(perturbed_params(opt, c["seed"]))
        eng.set_batch(*batch)
        x = eng.probe_encoders(dcube)
        torch.cuda.synchronize()
        out[tag] = (x.cpu().numpy().copy(), {n: v.double().cpu().numpy().copy() for n, v in eng.grads.items() if n.startswith(("W_t", "rnn_", "ln_a", "ln_v"))})
        eng.close()
    assert np.array_equal(out["h16"][0], out["fp32"][0]), "encoder outputs differ"
    top = max(np.abs(v).max() for v in out["fp32"][1].values())
    worst = {}
    for n, want in out["fp32"][1].items():
        scale = max(np.abs(want).max(), 1e-3 * top)
        # round 5b: dW_ih of layer 1 reads the fp16 copy too and converts it to bf16 in registers (GemmDesc::b_f16cvt): fp32 -> fp16 -> bf16 differs
        # from fp32 -> bf16 for the ~3 % of the values that sit within 2^-12 of a bf16 rounding midpoint, by one bf16 ulp each -- a perturbation of
        # that one product's operand, far below the bf16 rounding itself (measured <= the band below; every other tensor: atomics order only)
        band = 2e-3 if "weight_ih_l1" in n else 1e-4
        worst[n] = float(np.abs(out["h16"][1][n] - want).max() / scale)
        assert worst[n] <= band, (n, worst[n])
    print("h16-vs-fp32 worst relative differences:", sorted(worst.items(), key=lambda kv: -kv[1])[:6])


@pytest.mark.parametrize("name,T_", [("cfg2_sep", None), ("cfg2_ragged", 49)])
def test_16bit_stored_bptt_operands_stay_inside_the_bf16_band(name, T_, monkeypatch):
    """Round 5b (MIMRL_REC16, default on): the two operands the BPTT launches stream per cell step are stored in 16 bits -- dout as bf16 by its
    producer (the LayerNorm backward for layer 1, the tall dh0 product's epilogue for layer 0) and the layer-0 h_prev from the forward
    kernel's fp16 copy of its outputs, whose fp32 twin the fused-projection forward then does not write at all (0.85 GB less HBM traffic per
    cfg3 step).  Unlike the 16-bit stored GEMM operands these ARE new rounding points (a gradient that was added to the fp32 carry un-rounded
    is now rounded to bf16 first), so against MIMRL_REC16=0 the encoder outputs stay bit-identical and every gradient moves by a bf16-rounding-sized
    amount: <= 1e-2 of the tensor's scale per element, <= 5e-3 in L2 -- the rounded-operand oracle test below models the new points and holds
    the kernels to its usual 3x bands."""
    c, opt, batch, banks = case(name)
    T = c["T"] if T_ is None else T_
    batch = tuple(b[:, :T] if b.dim() == 3 else b for b in batch)
    g = torch.Generator().manual_seed(3)
    dcube = torch.randn(c["B"], opt.time_len, 3, 128, generator=g) / T
    out = {}
    for tag, env in (("rec16", None), ("fp32", "0")):
        if env:
            monkeypatch.setenv("MIMRL_REC16", env)
        else:
            monkeypatch.delenv("MIMRL_REC16", raising=False)
        eng = HipEngine(opt, 768, 74, 35, seq_len=T, bank_capacity=c["N"], precision="bf16")
        eng.load_params(perturbed_params(opt, c["seed"]))
        eng.set_batch(*batch)
        x = eng.probe_encoders(dcube)
        torch.cuda.synchronize()
        out[tag] = (x.cpu().numpy().copy(), {n: v.double().cpu().numpy().copy() for n, v in eng.grads.items() if n.startswith(("W_t", "rnn_", "ln_a", "ln_v"))})
        eng.close()
    assert np.array_equal(out["rec16"][0], out["fp32"][0]), "encoder outputs differ"
    top = max(np.abs(v).max() for v in out["fp32"][1].values())
    worst, moved = {}, 0
    for n, want in out["fp32"][1].items():
        scale = max(np.abs(want).max(), 1e-3 * top)
        diff = out["rec16"][1][n] - want
        worst[n] = float(np.abs(diff).max() / scale)
        moved += int(worst[n] > 1e-4)
        assert worst[n] <= 1e-2, (n, worst[n])
        assert np.linalg.norm(diff) <= 5e-3 * max(np.linalg.norm(want), 1e-3 * top * np.sqrt(want.size)), (n, np.linalg.norm(diff) / np.linalg.norm(want))
    assert moved >= 8, "the 16-bit stored operands were not in use"
    _record(f"rec16_vs_fp32_stored_bptt_operands/{name}{'' if T_ is None else '-T' + str(T_)}", dict(sorted(worst.items(), key=lambda kv: -kv[1])[:8]))


@pytest.mark.parametrize("name,T_,margin,gxh", ENC, ids=[f"{n}{'' if t is None else '-T' + str(t)}{'' if mg else '-full'}{'-gxf16' if gh else ''}" for n, t, mg, gh in ENC])
def test_encoders_vs_rounded_oracle(name, T_, margin, gxh, monkeypatch):
    _encoders_case(name, T_, margin, gxh, monkeypatch)


@pytest.mark.parametrize("name,T_,gxh", [("cfg2_sep", None, False), ("cfg2_ragged", 49, False), ("cfg2_ragged", None, True)])
def test_encoders_through_the_tall_gemm_kernel(name, T_, gxh, monkeypatch):
    """Round 5: the layer-1 input projection and the dh0 data gradient at B * T >= 4096 rows (cfg2 and up) run on the LDS-DMA
    kernel of csrc/gemm_tall.hip (256 x 128 tiles, 16-bit stored k-contiguous operands; dh0 reads the transposed, direction-concatenated bf16
    image of W_ih_l1).  MIMRL_GEMM_TALL_MIN_M lowers the row threshold so that the cfg2-shaped encoder parity cases -- every W_t / rnn_* / ln_*
    gradient against the rounded-operand float64 oracle, same bands -- exercise that path too (ragged rows: B * T = 6272 is not a multiple of
    the 256-row tile; fp16-stored gx)."""
    monkeypatch.setenv("MIMRL_GEMM_TALL_MIN_M", "1024")
    _encoders_case(name, T_, True, gxh, monkeypatch, tag="-tall")


def _encoders_case(name, T_, margin, gxh, monkeypatch, tag=""):
    """The recurrence kernels of the benchmarked mode -- gru_bwd_kernel<bf16, bf16 dg> is the largest kernel of the step, gru_fwd_kernel<bf16>
    the fourth -- with the fp16 input projections, LayerNorm / ReLU and every weight-gradient GEMM around them, driven through
    mimrl_probe_encoders (the step's own code path) on a fixture batch, against float64 autograd of the oracle (Model.py:395-466; nn.GRU on
    packed sequences: Model.py:254-255,441-447) with operands rounded exactly where gru.hip / gemm_fast_f16 / gemm_fast_bf round
    (tests/rounded_ref.py::_GruDirQ, bigru2_q, encoders_q: the state tile and W_hh -> bf16 per cell step, the saved gates -> bf16, the dgh
    tile -> bf16 for the carry product, projection operands -> fp16 forward / bf16 backward, dg and h_prev -> bf16).  Until round 4 these
    kernels were held to the oracle only at B = 16, T = 12, one layer, against the un-rounded fp32 cell at 2e-2 / 4e-2.
    Every gradient tensor (W_t, the 32 rnn_* tensors, ln_a / ln_v) <= 3x the measured error, no escape hatch; the upstream gradient is
    random in the cube input AND in the three temporal means."""
    c, opt, batch, banks = case(name)
    T = c["T"] if T_ is None else T_
    batch = tuple(b[:, :T] if b.dim() == 3 else b for b in batch)
    monkeypatch.setenv("MIMRL_GX_F16", "1" if gxh else "0")
    gxq = Q.r_f16 if gxh else Q.identity
    eng = HipEngine(opt, 768, 74, 35, seq_len=T, bank_capacity=c["N"], precision="bf16")
    p = perturbed_params(opt, c["seed"])
    eng.load_params(p)
    eng.set_batch(*batch)
    B, L = c["B"], opt.time_len
    g = torch.Generator().manual_seed(21)
    dcube = torch.randn(B, L, 3, 128, generator=g, dtype=torch.float64) / T
    dmean = torch.randn(3, B, 128, generator=g, dtype=torch.float64)
    tb = tuple(b.double() for b in batch)
    if margin:
        # ReLU kinks: kernel and reference agree to 2e-4 in the LayerNorm outputs (rounding-boundary flips of the fp16 /
        # bf16 operands), so ~1e-4 of the 1.6 M ReLU units have different masks in the two evaluations and each flip moves a gradient
        # by one unit's whole contribution: 1-3 % of the layer-1 weight gradients' scale, the size of the bug this test is for.  The
        # upstream gradient is therefore zero wherever a unit's pre-activation is within 2e-3 of the kink (10x the forward difference; 0.2 %
        # of the units) and the A_F / V_F mean gradients -- which reach every unit -- are off; `full` keeps both and a kink-aware band.
        with torch.no_grad():
            D = opt.d_common
            la, lv = R.infer_lengths(tb[1]), R.infer_lengths(tb[2])
            pre = [F.layer_norm(Q.bigru2_q(p, f"rnn_{m}", tb[1 + i], ln_, Q.F16_FWD, Q.r_bf16, gxq), (D,), p[f"ln_{m}.weight"], p[f"ln_{m}.bias"], 1e-6)
                   for i, (m, ln_) in enumerate((("a", la), ("v", lv)))]
        for i in range(2):
            dcube[:, :T, 1 + i][pre[i].abs() < 2e-3] = 0.0
        dmean[1:] = 0.0
    x = eng.probe_encoders(dcube, dmean)
    torch.cuda.synchronize()
    names = [n for n in p if n.startswith(("W_t", "rnn_", "ln_a", "ln_v"))]

    def reference(rnd, rq):
        leaves = {n: p[n].clone().requires_grad_(True) for n in names}
        # (rec16: the engine's default storage of the BPTT's streamed operands -- dout bf16, layer-0 h_prev from the fp16 copy; the layer-0 dout
        #  rounding applies where dh0 comes out of the tall GEMM kernel, B * T >= its row threshold: every case here but tiny_ragged, whose
        #  difference the bands absorb)
        xr, tf, af, vf = Q.encoders_q({**p, **leaves}, opt, tb[0], tb[1], tb[2], rnd, rq, gxq if rq is not Q.identity else Q.identity,
                                      rec16=rq is not Q.identity and os.environ.get("MIMRL_REC16", "1") != "0")
        obj = (xr * dcube).sum() + (tf * dmean[0]).sum() + (af * dmean[1]).sum() + (vf * dmean[2]).sum()
        return xr.detach(), torch.autograd.grad(obj, [leaves[n] for n in names])

    ref, gr = reference(Q.F16_FWD, Q.r_bf16)
    exact, gx = reference(Q.EXACT, Q.identity)              # the un-rounded oracle: how far the 16-bit mode is from fp32 / fp64
    key = f"encoders/{name}{'' if T_ is None else '-T' + str(T_)}{'' if margin else '-full'}{'-gxf16' if gxh else ''}{tag}"
    rec = {"cube_x": errs(x.cpu(), ref), "cube_x_vs_unrounded_oracle": errs(x.cpu(), exact)}
    worst = ("", 0.0)
    for n, gw, ge in zip(names, gr, gx):
        e = errs(eng.grads[n].cpu(), gw)
        e["vs_unrounded_oracle"] = errs(eng.grads[n].cpu(), ge)["max_rel_scale"]
        rec[n] = e
        if e["max_rel_scale"] > worst[1]:
            worst = (n, e["max_rel_scale"])
    rec["worst_grad"] = worst
    _record(key, rec)
    eng.close()
    assert len(names) == 37
    assert rec["cube_x"]["max_rel_scale"] <= 3e-3, (key, rec["cube_x"])
    tol = ENC_TOL if margin else ENC_TOL_FULL
    bad = {n: e["max_rel_scale"] for n, e in rec.items() if isinstance(e, dict) and n in names and e["max_rel_scale"] > tol}
    assert not bad, (key, bad)


ENC_TOL = 6e-3        # margin-filtered upstream gradient (no ReLU-mask flips between kernel and reference): 3x the measured 2.0e-3
ENC_TOL_FULL = 6e-2   # every unit carries gradient: 3x the measured 2e-2 (mask flips of ~1e-4 of the units, see above)


def test_encoders_with_8_wave_recurrence_knob():
    """MIMRL_GRU_WAVES=8 picks the one-unit-per-lane recurrence kernels, but the fused layer-0 projection forward always
    writes its saved-gate slab in the 4-wave record layout -- the layer-0 BPTT launch has to follow the SLAB (GruBwdArgs::slab_upl), not the
    knob.  The knob is read once per process, so the encoder parity cases run again in a child process with it set."""
    import subprocess
    import sys
    env = dict(os.environ, MIMRL_GRU_WAVES="8")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-m", "gpu", "-x", "-p", "no:cacheprovider",
                        "-k", "test_encoders_vs_rounded_oracle and (cfg2_sep or cfg1_ragged) and not gxf16"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout and "failed" not in r.stdout, r.stdout[-1000:]
