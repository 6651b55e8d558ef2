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
    """One gradient tensor against the rounded-operand float64 reference.  Max-norm error relative to the tensor's scale within
    ``tol`` -- or, for the two documented exceptions, L2 within ``tol`` and no entry off by more than 15 x tol:
    (i) ReLU kinks: a unit whose pre-activation is within fp32 noise of zero has a different mask in two equally valid evaluation
    orders, and every entry it feeds moves by one row's contribution (tests/gpu_helpers.py::grad_close);
    (ii) rounding is chaotic: once kernel and reference differ by 1e-3 somewhere upstream, a quarter of the bf16 roundings of the
    next gradient operand land on different neighbours -- single entries carry a full 2^-9 step of one operand.
    A tensor whose own scale is below 2 % of the largest gradient of the probe (a sum that cancels, e.g. the last-layer bias under
    InfoNCE's shift invariance) is held to ``tol`` of THAT scale: it is noise in both implementations."""
    scale = max(e["scale"], 0.02 * group_scale)
    mx = e["max_rel_scale"] * e["scale"] / scale
    tol = max(tol, 3.0 * chaos * e["scale"] / scale)      # (iii) see chaos_floor()
    return mx <= tol or (e["l2_rel"] * e["scale"] / scale <= tol and mx <= 15 * tol)


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
    bad = {n: e for n, e in rec.items() if isinstance(e, dict) and (n == "dx" or n.startswith("mlp_encoder.")) and
           not tensor_ok(e, TOL, gs if n != "dx" else e["scale"], e["chaos"])}
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
    feats = 0.25 * torch.randn(4, B, 128, generator=g, dtype=torch.float64)      # the scale of real features (means over T of O(1) rows)
    feats[2:] = feats[2:].abs()                                                    # A_F, V_F are means of ReLU outputs
    r = eng.probe_mi(stage, feats)
    torch.cuda.synchronize()
    names = [n for n in p if n.startswith("vmi_estimator_")]
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
    bad = {n: e for n, e in gt.items() if not tensor_ok(e, TOL, gs, e["chaos"])}
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
