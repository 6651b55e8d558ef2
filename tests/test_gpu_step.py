"""-m gpu: the full two-stage step through the C ABI vs (a) reference-generated goldens, (b) the CPU oracle."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from mimrl_amd import _lib, synth
from mimrl_amd.engine import HipEngine
from oracle import mimrl_ref as R
from tests.gpu_helpers import assert_close, grad_close, oracle_raw_grads
from tests.helpers import case, load_golden, oracle_params

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)

pytestmark = pytest.mark.gpu

TINY = ["tiny_sep", "tiny_cat", "tiny_ragged", "tiny_alt", "tiny_conv", "tiny_mine", "tiny_odd", "tiny_interp", "tiny_lstm", "tiny_tuba_un", "tiny_interp_ga",
        "tiny_sum", "tiny_disc"]
# cfg2_sep = BASELINE configs[1] at FULL size (the bench configuration); cfg3_small / cfg5_small = configs[2] / [4] with only the
# batch reduced (T = time_len = 500 / 1000, concat critic for cfg3)
ALL = TINY + ["cfg1_sep", "cfg1_cat", "cfg1_disc", "cfg1_ragged", "cfg1_lstm", "cfg2_sep", "cfg2_ragged", "cfg3_small", "cfg5_small"]


def make_engine(name, precision="fp32", use_graph=False):
    c, opt, batch, banks = case(name)
    eng = HipEngine(opt, 768, 74, 35, seq_len=c["T"], bank_capacity=c["N"], precision=precision, use_graph=use_graph)
    p = oracle_params(opt, c["seed"])
    eng.load_params(p)
    eng.set_batch(*batch)
    return c, opt, batch, banks, p, eng


def _record_errors(key, value):
    """Measured errors of the round's new parity tests -> gpurun_out/r05_step_errors.json (copied to profiles/ by hand)."""
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.path.join(root, "gpurun_out", "r06_step_errors.json")
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        try:
            data = json.load(open(path)) if os.path.exists(path) else {}
        except ValueError:
            data = {}
        data[key] = json.loads(json.dumps(value, default=float))
        json.dump(data, open(path, "w"), indent=1, sort_keys=True)
    except (OSError, ValueError):
        pass


@pytest.mark.parametrize("name", ALL)
def test_forward_matches_golden(name):
    c, opt, batch, banks, p, eng = make_engine(name)
    g = load_golden(name)
    eng.forward(train=False)
    torch.cuda.synchronize()
    feats = eng.feats.cpu().numpy()
    assert_close(eng.pred.cpu().numpy(), g["fwd_pred"].reshape(-1), 1e-3, 1e-5, "pred")
    for i, k in enumerate(["F_F", "T_F", "A_F", "V_F"]):
        assert_close(feats[i], g["fwd_" + k], 1e-3, 2e-5, k)
    eng.close()


@pytest.mark.parametrize("name", ["tiny_sep", "cfg1_sep"])
def test_epoch0_rule(name):
    c, opt, batch, banks, p, eng = make_engine(name)
    g = load_golden(name)
    before = eng.crit["p"].clone()
    eng.set_banks(None, None, None, None, None)
    eng.stage1_step()
    eng.forward(train=True, with_losses=True)
    s = eng.read_scalars()
    assert s[_lib.S1_LOSS] == 0.0 and torch.equal(before, eng.crit["p"])      # Solver.py:201-203
    assert_close(s[_lib.S2_LOSS], g["e0_stage2_loss"], 1e-3, 1e-6, "epoch-0 stage-2 loss = task loss")
    assert np.all(s[_lib.S2_MIS:_lib.S2_MIS + 8] == 0)
    eng.close()


@pytest.mark.parametrize("name", ALL + ["tiny_ragged+l0pack", "cfg1_sep+l0pack"])
def test_stage_losses_and_all_gradients_vs_oracle(name, monkeypatch):
    """Every MI/CMI value, both losses and EVERY parameter gradient of both stages against the oracle's autograd.
    (`+l0pack`: with the packed layer-0 GRU operands forced on -- the default only above 16384 rows, i.e. cfg3 / cfg5.)"""
    if name.endswith("+l0pack"):
        name = name[:-7]
        monkeypatch.setenv("MIMRL_L0_PACK", "1")
    c, opt, batch, banks, p, eng = make_engine(name)
    g = load_golden(name)
    anchors = g["anchors"][0]
    eng.set_banks(*(banks[k] for k in "CFTAV"))
    crit = [n for n in p if R.is_critic_param(n)]
    main = [n for n in p if not R.is_critic_param(n)]
    adam_v = R.AdamState(p, crit)
    for stage, names in ((1, crit), (2, main)):
        eng.set_anchors(stage, anchors[stage - 1], exact_ties=bool(c.get("discrete")))
        eng.stage_grads(stage)
        torch.cuda.synchronize()
        s = eng.read_scalars()
        loss, mis, pred, feats, task, grads = oracle_raw_grads(p, opt, stage, batch, banks, anchors[stage - 1], names)
        if stage == 1:
            assert_close(s[_lib.S1_LOSS], loss.item(), 1e-3, 1e-5, "stage-1 loss")
            assert_close(s[_lib.S1_MIS:_lib.S1_MIS + 11], [m.item() for m in mis], 1e-3, 2e-5, "stage-1 MI/CMI")
            assert_close(s[_lib.S1_LOSS], g["traj_s1_loss"][0], 1e-3, 1e-5, "stage-1 loss vs reference")
            assert_close(s[_lib.S1_MIS:_lib.S1_MIS + 11], g["traj_s1_mis"][0], 1e-3, 2e-5, "stage-1 MI/CMI vs reference")
        else:
            assert_close(s[_lib.S2_LOSS], loss.item(), 1e-3, 1e-5, "stage-2 loss")
            assert_close(s[_lib.S2_TASK], task.item(), 1e-3, 1e-6, "task loss")
            assert_close(s[_lib.S2_MIS:_lib.S2_MIS + 8], [m.item() for m in mis], 1e-3, 5e-5, "stage-2 MI terms")
            assert_close(s[_lib.S2_LOSS], g["traj_s2_loss"][0], 1e-3, 1e-5, "stage-2 loss vs reference")
            assert_close(s[_lib.S2_MIS:_lib.S2_MIS + 8], g["traj_s2_mis"][0], 1e-3, 5e-5, "stage-2 MI vs reference")
        bad = []
        for n in names:
            try:
                # interpolate + Gaussian baseline (kernel pinned to 1e-3 by test_bounds_with_log_baseline): with log a(y) ~ -120 the
                # reference's own fp32 value is 0.6 % off its float64 value, so equivalent fp32 evaluation orders; two fp32
                # implementations (oracle vs reference too) already differ by ~1 % in single gradient entries there
                grad_close(eng.grads[n].cpu().numpy(), grads[n].numpy(), 1.5e-1 if name == "tiny_interp_ga" else 3e-3, n)
            except AssertionError as e:
                bad.append(str(e))
        assert not bad, f"stage {stage}: {len(bad)}/{len(names)} gradient tensors off:\n" + "\n".join(bad[:12])
        # reference-captured small gradient tensors (pre-clip)
        for key in g.files:
            if key.startswith(f"s{stage}_grad:"):
                n = key.split(":", 1)[1]
                # stage 2 is evaluated after the critic update, which carries Adam sign-flip noise (see below)
                grad_close(eng.grads[n].cpu().numpy(), g[key], (3e-3 if stage == 1 else 3e-2) * (4 if name == "tiny_interp_ga" else 1),
                           "vs reference " + n, atol=3e-7 if stage == 1 else 3e-6)
        if stage == 1:
            # apply the critic update on both sides so that stage 2 is evaluated exactly where the reference
            # evaluates it (Solver.py:213 precedes :221); Adam t=1 ~ lr*sign(g), hence compare loosely here
            eng.stage_apply(1)
            clip = float(opt.gradient_clip)
            adam_v.step(p, {n: grads[n].clamp(-clip, clip) for n in crit}, float(opt.learning_rate) * float(opt.mi_lr_rate),
                        float(opt.weight_decay))
            dp = max((eng.params[n].cpu() - p[n]).abs().max().item() for n in crit)
            assert dp <= 2.5 * float(opt.learning_rate), f"critic update differs by {dp}"
            eng.load_params({n: p[n] for n in crit}, strict=False)   # remove sign-flip noise of ~0 gradients
    eng.close()


@pytest.mark.parametrize("name", ["tiny_sep", "tiny_cat", "tiny_ragged", "tiny_conv", "tiny_mine", "tiny_odd", "tiny_lstm", "tiny_sum", "tiny_disc", "cfg1_sep", "cfg1_disc",
                                  "cfg1_ragged", "cfg1_lstm", "cfg2_sep", "cfg2_ragged", "cfg3_small", "cfg5_small"])
@pytest.mark.parametrize("use_graph", [False, True])
def test_two_stage_trajectory(name, use_graph):
    """Alternating stage-1/stage-2 updates (Solver.step) vs the reference trajectory and the oracle."""
    c, opt, batch, banks, p, eng = make_engine(name, use_graph=use_graph)
    g = load_golden(name)
    anchors = g["anchors"]
    eng.set_banks(*(banks[k] for k in "CFTAV"))
    crit = [n for n in p if R.is_critic_param(n)]
    main = [n for n in p if not R.is_critic_param(n)]
    adam_v, adam_m = R.AdamState(p, crit), R.AdamState(p, main)
    steps = min(anchors.shape[0], 3)
    for it in range(steps):
        eng.set_anchors(1, anchors[it, 0], exact_ties=bool(c.get("discrete")))
        eng.set_anchors(2, anchors[it, 1], exact_ties=bool(c.get("discrete")))
        eng.stage1_step()
        eng.stage2_step()
        s = eng.read_scalars()
        r1, r2 = R.two_stage_step(p, opt, adam_v, adam_m, batch, banks, anchors[it, 0], anchors[it, 1])
        rt, at = (1e-3, 2e-5) if it == 0 else ((3e-2, 5e-3) if it < 3 else (0.25, 0.05))
        assert_close(s[_lib.S1_LOSS], g["traj_s1_loss"][it], rt, at, f"it{it} s1 loss vs reference")
        assert_close(s[_lib.S2_LOSS], g["traj_s2_loss"][it], rt, at, f"it{it} s2 loss vs reference")
        assert_close(s[_lib.S2_TASK], g["traj_s2_task"][it], rt, at, f"it{it} task vs reference")
        assert_close(s[_lib.S2_MIS:_lib.S2_MIS + 8], g["traj_s2_mis"][it], rt, 5e-5 if it == 0 else 5e-3, f"it{it} MI vs reference")
        assert_close(s[_lib.S1_LOSS], r1["loss"].item(), rt, at, f"it{it} s1 loss vs oracle")
        assert_close(s[_lib.S2_LOSS], r2["loss"].item(), rt, at, f"it{it} s2 loss vs oracle")
        if it == 0:   # parameters after one clip+Adam update of each bucket (tolerance = a few sign flips of ~0 grads)
            names1 = [str(x) for x in g["s1_gnorm_names"]]
            ps = np.array([eng.params[n].double().sum().item() for n in names1])
            np.testing.assert_allclose(ps, g["s1_psum_after"], rtol=1e-4, atol=0.05)
            names2 = [str(x) for x in g["s2_gnorm_names"]]
            ps = np.array([eng.params[n].double().sum().item() for n in names2])
            np.testing.assert_allclose(ps, g["s2_psum_after"], rtol=1e-4, atol=0.05)
    eng.close()


def test_bf16_mode_tracks_fp32():
    """bf16 MFMA operands (GEMMs + GRU recurrence), fp32 accumulation/state/statistics/optimizer.
    Every loss and MI/CMI term stays within 2e-2 relative (+ small absolute band: CMI is a difference of log-ratio
    sums) of the fp32 path and of the reference.  Gradients are compared by direction only: this model's backward
    (broadcast means through LayerNorms over K=3 / L) cancels most of the signal, so operand rounding in the
    FORWARD pass perturbs individual gradient entries -- measured and documented in DESIGN.md: O(10 %) with bf16 forward operands,
    which is why the forward products on the model path round to fp16 instead (same MFMA rate, 8x finer)."""
    res = {}
    for prec in ("fp32", "bf16"):
        c, opt, batch, banks, p, eng = make_engine("cfg1_sep", precision=prec)
        g = load_golden("cfg1_sep")
        anchors = g["anchors"][0]
        eng.set_banks(*(banks[k] for k in "CFTAV"))
        eng.set_anchors(1, anchors[0])
        eng.set_anchors(2, anchors[1])
        eng.stage_grads(1)
        torch.cuda.synchronize()
        g1 = eng.crit["g"].double().cpu().numpy().copy()
        eng.stage_grads(2)
        torch.cuda.synchronize()
        res[prec] = (eng.read_scalars().copy(), g1, eng.main["g"].double().cpu().numpy().copy())
        eng.close()
    a, b = res["fp32"][0], res["bf16"][0]
    assert_close(b[_lib.S1_LOSS], a[_lib.S1_LOSS], 2e-2, 1e-3, "bf16 stage-1 loss")
    assert_close(b[_lib.S1_LOSS], g["traj_s1_loss"][0], 2e-2, 1e-3, "bf16 stage-1 loss vs reference")
    assert_close(b[_lib.S1_MIS:_lib.S1_MIS + 11], a[_lib.S1_MIS:_lib.S1_MIS + 11], 2e-2, 2e-2, "bf16 MI/CMI")
    assert_close(b[_lib.S2_LOSS], a[_lib.S2_LOSS], 2e-2, 1e-3, "bf16 stage-2 loss")
    assert_close(b[_lib.S2_MIS:_lib.S2_MIS + 8], a[_lib.S2_MIS:_lib.S2_MIS + 8], 2e-2, 4e-2, "bf16 MI terms")
    # measured 0.9994 / 0.9990.  The main bucket was 0.964 until round 3: fp16 instead of bf16 operands in the
    # forward products of CubeMLP and of the projections in front of it (cube_fused.hip) -- the bound below is what pins that
    for i, nm, lim in ((1, "critic", 0.998), (2, "main", 0.995)):
        va, vb = res["fp32"][i], res["bf16"][i]
        cos = float(va @ vb / (np.linalg.norm(va) * np.linalg.norm(vb)))
        assert cos > lim, f"bf16 {nm} gradient direction: cosine {cos}"


@pytest.mark.parametrize("name", ["tiny_sep", "cfg1_sep"])
def test_fused_cube_forward_matches_unfused(name, monkeypatch):
    """bf16 mode: the LDS-resident fused CubeMLP block kernel (cube_fused.hip) against the unfused GEMM/LN/K-mix chain
    in the same precision mode: forward outputs, every loss / MI term and the whole main-bucket gradient (the fused
    kernel also produces the saved activations the backward consumes)."""
    res = {}
    for tag, env in (("fused", None), ("unfused", "1")):
        if env:
            monkeypatch.setenv("MIMRL_NO_FUSED_CUBE", env)
        else:
            monkeypatch.delenv("MIMRL_NO_FUSED_CUBE", raising=False)
        c, opt, batch, banks, p, eng = make_engine(name, precision="bf16")
        g = load_golden(name)
        eng.set_banks(*(banks[k] for k in "CFTAV"))
        eng.set_anchors(1, g["anchors"][0, 0])
        eng.set_anchors(2, g["anchors"][0, 1])
        eng.stage_grads(1)
        eng.stage_grads(2)
        torch.cuda.synchronize()
        res[tag] = (eng.read_scalars().copy(), eng.feats.cpu().numpy().copy(), eng.pred.cpu().numpy().copy(),
                    eng.main["g"].double().cpu().numpy().copy())
        eng.close()
    (sa, fa, pa, ga), (sb, fb, pb, gb) = res["fused"], res["unfused"]
    # the fused kernel keeps the inter-mix activations as bf16 in LDS: ~1 % of the activation scale
    assert_close(pa, pb, 3e-2, 1e-2, "pred")
    assert_close(fa[0], fb[0], 3e-2, 1e-2, "F_F")
    assert_close(fa[1:], fb[1:], 1e-6, 1e-7, "T_F/A_F/V_F do not pass through CubeMLP")
    assert_close(sa[_lib.S1_LOSS], sb[_lib.S1_LOSS], 1e-2, 1e-3, "stage-1 loss")
    assert_close(sa[_lib.S2_LOSS], sb[_lib.S2_LOSS], 1e-2, 1e-3, "stage-2 loss")
    assert_close(sa[_lib.S2_MIS:_lib.S2_MIS + 8], sb[_lib.S2_MIS:_lib.S2_MIS + 8], 2e-2, 2e-2, "MI terms")
    # (since round 3 the fused kernel's operands are fp16 and the unfused chain's bf16: the two gradients are no longer "the same
    #  numbers in a different order" -- both are held against the oracle in tests/test_gpu_fused_oracle.py; here only the direction,
    #  at the fixture where the bf16 gradient is not noise: L = 6 / 4 LayerNorm axes make tiny_sep's a coin toss, cos 0.96)
    cos = float(ga @ gb / (np.linalg.norm(ga) * np.linalg.norm(gb)))
    assert cos > (0.98 if name != "tiny_sep" else 0.9), f"main gradient direction fused vs unfused: cosine {cos}"


@pytest.mark.parametrize("precision,use_graph,split", [("fp32", False, False), ("fp32", True, False), ("bf16", True, False),
                                                       ("fp32", True, True)])
def test_stage2_prefetch_matches_sequential(precision, use_graph, split):
    _prefetch_matches_sequential(precision, use_graph, split)


def test_prefetch_without_the_shared_prefix_matches_sequential():
    """MIMRL_NO_SHARED_PREFIX=1 (a tuning knob: both forward passes evaluate the encoders): stage 1's own, NON-saving forward pass is captured
    behind the saving stage-2 pass, so per-pass facts the backward relies on -- which 16-bit copies the saving pass wrote, whether it wrote its
    fp32 layer-0 outputs at all (round 5b: MIMRL_REC16) -- must be those of the SAVING pass.  The knob is read once per process: the overlap
    parity cases run again in a child process with it set."""
    import subprocess
    import sys
    env = dict(os.environ, MIMRL_NO_SHARED_PREFIX="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-m", "gpu", "-x", "-p", "no:cacheprovider",
                        "-k", "test_stage2_prefetch_matches_sequential and bf16"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout and "failed" not in r.stdout, r.stdout[-1000:]


def _prefetch_matches_sequential(precision, use_graph, split):
    """Solver.step() overlap mode (mimrl_set_stage2_prefetch): the stage-2 forward pass runs beside stage 1.
    Same parameters, inputs and dropout keys => same losses / predictions / parameters as the sequential order
    (dropout is ON here so that a wrong mask key between forward and backward would show)."""
    import copy
    name = "cfg1_sep"
    out = []
    for pre in (False, True):
        c, opt, batch, banks = case(name)
        opt = copy.copy(opt)
        opt.dropout = [0.1, 0.1, 0.1, 0.1]
        opt.dropout_mlp = [0.1, 0.1, 0.1]         # also exercises the unfused CubeMLP path with its dropped-out gradient buffers
        eng = HipEngine(opt, 768, 74, 35, seq_len=c["T"], bank_capacity=c["N"], precision=precision, use_graph=use_graph)
        eng.load_params(oracle_params(opt, c["seed"]))
        eng.set_batch(*batch)
        eng.set_banks(*(banks[k] for k in "CFTAV"))
        g = load_golden(name)
        eng.set_anchors(1, g["anchors"][0, 0])
        eng.set_anchors(2, g["anchors"][0, 1])
        eng.set_stage2_prefetch(pre)
        rec = []
        for _ in range(3):
            if split and pre:    # the data-parallel call sequence (dist.ddp_stage_step): grads | all-reduce | apply, per stage
                for st in (1, 2):
                    eng.stage_grads(st)
                    eng.stage_apply(st)
            else:
                eng.step()
            rec.append((eng.read_scalars().copy(), eng.pred.cpu().numpy().copy(), eng.feats.cpu().numpy().copy()))
        rec.append(torch.cat([eng.params[n].flatten() for n in sorted(eng.params)]).cpu().numpy())
        if pre:   # stage 2 without its stage 1 must fail loudly in this mode
            eng.stage1_step(); eng.stage2_step()
            with pytest.raises(Exception):
                eng.stage2_step()
        eng.close()
        out.append(rec)
    seq, pre = out
    for it in range(3):
        # the forward is bit-stable; later steps inherit the atomics-order noise of the preceding updates
        # (bf16: a flipped operand rounding after the first update is a 2^-9 relative kick, amplified by the LayerNorms)
        if it > 0 and precision != "fp32":
            continue    # bf16 runs are compared at step 0 (exact) and through the final parameters only
        rt, at = (1e-6, 1e-7) if it == 0 else (1e-2, 1e-3)
        assert_close(pre[it][1], seq[it][1], rt, at, f"it{it} pred")
        assert_close(pre[it][2], seq[it][2], rt, at, f"it{it} feats")
        assert_close(pre[it][0][:64], seq[it][0][:64], max(rt, 1e-5), max(at, 1e-5), f"it{it} scalars")
    cos = float(np.dot(pre[3], seq[3]) / (np.linalg.norm(pre[3]) * np.linalg.norm(seq[3])))
    assert cos > 1 - (1e-6 if precision == "fp32" else 1e-3), cos


def test_fused_mlp_stacks_match_unfused(monkeypatch):
    """bf16 mode: estimator MLP stacks as fused kernels (mlp_fused.hip) vs the grouped-GEMM chain in the same precision:
    every MI/CMI term, both losses, and the critic / main gradient buckets."""
    res = {}
    for tag, env in (("fused", None), ("unfused", "1")):
        if env:
            monkeypatch.setenv("MIMRL_NO_FUSED_MLP", env)
        else:
            monkeypatch.delenv("MIMRL_NO_FUSED_MLP", raising=False)
        c, opt, batch, banks, p, eng = make_engine("cfg1_sep", precision="bf16")
        g = load_golden("cfg1_sep")
        eng.set_banks(*(banks[k] for k in "CFTAV"))
        eng.set_anchors(1, g["anchors"][0, 0])
        eng.set_anchors(2, g["anchors"][0, 1])
        eng.stage_grads(1)
        gc = eng.crit["g"].double().cpu().numpy().copy()
        eng.stage_grads(2)
        torch.cuda.synchronize()
        res[tag] = (eng.read_scalars().copy(), gc, eng.main["g"].double().cpu().numpy().copy())
        eng.close()
    (sa, ca, ma), (sb, cb, mb) = res["fused"], res["unfused"]
    assert_close(sa[:64], sb[:64], 2e-3, 2e-4, "scalars (losses, MI / CMI terms)")
    for name, x, y, lim in (("critic", ca, cb, 0.9995), ("main", ma, mb, 0.999)):
        cos = float(x @ y / (np.linalg.norm(x) * np.linalg.norm(y)))
        assert cos > lim, f"{name} gradient direction fused vs unfused: cosine {cos}"


@pytest.mark.parametrize("name", ["tiny_sep", "cfg1_sep"])
def test_fused_cube_backward_matches_unfused(name, monkeypatch):
    """bf16 mode: the per-axis fused data-gradient kernels of CubeMLP (cube_bwd_fused.hip) against the unfused
    LayerNorm-backward + GEMM chain in the same precision: the whole main-bucket gradient, tensor by tensor."""
    res = {}
    for tag, env in (("fused", None), ("unfused", "1")):
        if env:
            monkeypatch.setenv("MIMRL_NO_FUSED_CUBE_BWD", env)
        else:
            monkeypatch.delenv("MIMRL_NO_FUSED_CUBE_BWD", raising=False)
        c, opt, batch, banks, p, eng = make_engine(name, precision="bf16")
        g = load_golden(name)
        eng.set_banks(*(banks[k] for k in "CFTAV"))
        eng.set_anchors(1, g["anchors"][0, 0])
        eng.set_anchors(2, g["anchors"][0, 1])
        eng.stage_grads(2)
        torch.cuda.synchronize()
        res[tag] = {n: eng.grads[n].double().cpu().numpy().copy() for n in eng.grads if not R.is_critic_param(n)}
        eng.close()
    worst = 1.0
    for n, ga in res["fused"].items():
        gb = res["unfused"][n]
        na, nb = np.linalg.norm(ga), np.linalg.norm(gb)
        if nb < 1e-12:
            assert na < 1e-9, n
            continue
        cos = float((ga * gb).sum() / (na * nb))
        worst = min(worst, cos)
        assert cos > 0.995 and abs(na / nb - 1) < 0.05, f"{n}: cosine {cos}, norm ratio {na / nb}"
    assert worst > 0.995


@pytest.mark.parametrize("name", TINY + ["cfg1_cat"])
def test_bf16_fused_paths_on_every_fixture(name):
    """Every fixture (odd sizes, ragged, conv encoder, concat critic, ln_first, mine ...) through the bf16 product path with
    all fused kernels, hipGraph and Solver.step() overlap mode on: losses and MI terms of the first step stay within bf16
    distance of the fp32 run, and two more steps stay finite."""
    res = {}
    for precision in ("fp32", "bf16"):
        c, opt, batch, banks, p, eng = make_engine(name, precision=precision, use_graph=precision == "bf16")
        g = load_golden(name)
        eng.set_banks(*(banks[k] for k in "CFTAV"))
        eng.set_anchors(1, g["anchors"][0, 0], exact_ties=bool(c.get("discrete")))
        eng.set_anchors(2, g["anchors"][0, 1], exact_ties=bool(c.get("discrete")))
        eng.set_stage2_prefetch(precision == "bf16")
        eng.step()
        res[precision] = eng.read_scalars().copy()
        if precision == "bf16":
            eng.step(); eng.step()
            assert np.isfinite(eng.read_scalars()).all()
            assert torch.isfinite(eng.main["p"]).all() and torch.isfinite(eng.crit["p"]).all()
        eng.close()
    a, b = res["bf16"], res["fp32"]
    assert_close(a[_lib.S1_LOSS], b[_lib.S1_LOSS], 2e-2, 2e-2, "stage-1 loss")
    if opt.cmi_last_acticate != "hardtanh":
        assert_close(a[_lib.S2_LOSS], b[_lib.S2_LOSS], 2e-2, 2e-2, "stage-2 loss")
    # the hardtanh CMI head clamps probabilities to [1e-4, 1-1e-4]: with 8 samples one logit crossing the clamp moves a
    # CMI term by ~1 in any reduced precision (same numbers with every fused kernel switched off), so only MI terms there
    nmi = 4 if opt.cmi_last_acticate == "hardtanh" else 8
    assert_close(a[_lib.S2_MIS:_lib.S2_MIS + nmi], b[_lib.S2_MIS:_lib.S2_MIS + nmi], 5e-2, 5e-2, "stage-2 MI terms")   # (CMI terms of 8-sample fixtures: differences of log-ratio sums)


@pytest.mark.parametrize("name", ["tiny_cat", "cfg1_cat"])
def test_fused_concat_forward_matches_gemm_chain(name, monkeypatch):
    """The fused concat-critic forward (pair expansion + two hidden layers + score head per 128-row tile, activations in LDS) against
    the pair_expand + three-GEMM chain in the same bf16 mode: same bf16-rounded operands, fp32 accumulation, so the MI values agree to
    fp32 summation-order noise and every critic gradient (the backward pass reads the activations the forward kernel saved) to 2e-3
    of its scale.  (tiny_cat: B*B = 64 rows -- not a multiple of the 128-row tile, both runs take the chain: the dispatch rule.)"""
    res = {}
    for tag in ("fused", "chain"):
        if tag == "chain":
            monkeypatch.setenv("MIMRL_NO_FUSED_CONCAT", "1")
        else:
            monkeypatch.delenv("MIMRL_NO_FUSED_CONCAT", raising=False)
        c, opt, batch, banks, p, eng = make_engine(name, precision="bf16")
        g = load_golden(name)
        eng.set_banks(*(banks[k] for k in "CFTAV"))
        eng.set_anchors(1, g["anchors"][0, 0], exact_ties=bool(c.get("discrete")))
        eng.stage_grads(1)
        torch.cuda.synchronize()
        res[tag] = (eng.read_scalars().copy(), {n: v.double().cpu().numpy().copy() for n, v in eng.grads.items() if "MLP_f" in n})
        eng.close()
    (sa, ga), (sb, gb) = res["fused"], res["chain"]
    # (InfoNCE at init is log B minus a mean of O(1) numbers, ~1e-5: one bf16 rounding of one activation that falls the other way
    #  in the other summation order moves it by 1e-5)
    assert_close(sa[_lib.S1_MIS:_lib.S1_MIS + 5], sb[_lib.S1_MIS:_lib.S1_MIS + 5], 1e-4, 5e-5, "MI values fused vs chain")
    assert len(ga) == 40
    for n in ga:
        grad_close(ga[n], gb[n], 2e-3, n)


@pytest.mark.parametrize("name", ["cfg2_sep", "cfg2_ragged"])
def test_cfg2_full_size_in_bench_mode(name):
    """(cfg2_ragged: the same with ragged audio / video lengths -- four batch rows of different lengths per recurrence workgroup.)
    BASELINE configs[1] at full size (B=128, T=50, N=1284) in EXACTLY the mode bench.py times -- bf16 MFMA operands, every
    fused kernel, one hipGraph per two-stage step, Solver.step() overlap mode with the shared encoder prefix -- against the
    reference's golden values and the fp32 engine.  (Anchors are host-supplied here so that both modes and the reference
    see the same kNN samples; bench.py draws them on the device.)"""
    g = load_golden(name)
    res = {}
    for precision in ("fp32", "bf16"):
        c, opt, batch, banks, p, eng = make_engine(name, precision=precision, use_graph=precision == "bf16")
        eng.set_banks(*(banks[k] for k in "CFTAV"))
        eng.set_anchors(1, g["anchors"][0, 0])
        eng.set_anchors(2, g["anchors"][0, 1])
        eng.set_stage2_prefetch(precision == "bf16")
        eng.step()
        torch.cuda.synchronize()
        res[precision] = (eng.read_scalars().copy(), eng.pred.cpu().numpy().copy(), eng.feats.cpu().numpy().copy())
        if precision == "bf16":
            for _ in range(3):
                eng.step()
            assert np.isfinite(eng.read_scalars()).all() and torch.isfinite(eng.main["p"]).all() and torch.isfinite(eng.crit["p"]).all()
        eng.close()
    (sa, pa, fa), (sb, pb, fb) = res["fp32"], res["bf16"]
    # fp32 mode vs the real reference: the 1e-3 bar
    assert_close(sa[_lib.S1_LOSS], g["traj_s1_loss"][0], 1e-3, 1e-5, "fp32 stage-1 loss vs reference")
    assert_close(sa[_lib.S2_LOSS], g["traj_s2_loss"][0], 1e-3, 1e-5, "fp32 stage-2 loss vs reference")
    assert_close(sa[_lib.S2_MIS:_lib.S2_MIS + 8], g["traj_s2_mis"][0], 1e-3, 5e-5, "fp32 MI terms vs reference")
    # bench mode vs the REFERENCE's values: bands = 3x the error measured in round 3 (three runs: losses 5e-6 relative; stage-1 MI / CMI
    # values <= 1.8e-5 absolute on values of 1e-5 / 1.0; stage-2 MI terms <= 1.9e-3 absolute on values of 2e-4 .. 2.08 -- the CMI-derived
    # terms are differences of log-ratio sums).  Until round 3 these bands were 5e-3 / 2e-2: wide enough to hide a wrong kernel.
    assert_close(sb[_lib.S1_LOSS], g["traj_s1_loss"][0], 2e-5, 1e-6, "bench-mode stage-1 loss vs reference")
    assert_close(sb[_lib.S2_LOSS], g["traj_s2_loss"][0], 2e-5, 1e-6, "bench-mode stage-2 loss vs reference")
    assert_close(sb[_lib.S2_TASK], g["traj_s2_task"][0], 2e-5, 1e-6, "bench-mode task loss vs reference")
    assert_close(sb[_lib.S1_MIS:_lib.S1_MIS + 11], g["traj_s1_mis"][0], 0, 6e-5, "bench-mode stage-1 MI/CMI vs reference")
    assert_close(sb[_lib.S2_MIS:_lib.S2_MIS + 8], g["traj_s2_mis"][0], 0, 6e-3, "bench-mode MI terms vs reference")
    assert_close(pb, pa, 2e-2, 2e-2, "bench-mode predictions vs fp32")
    assert_close(fb, fa, 2e-2, 2e-2, "bench-mode features vs fp32")


def _bench_engine(workload, precision, use_graph, device_anchors=True, dropout=0.0, **env):
    import bench
    seq = None
    if "@" in workload:                                       # "cfg2@49": T = 49 steps inside time_len = 50 (odd T: the peeled tail of the
        workload, seq = workload.split("@")                   # unrolled-by-two recurrence loops)
    opt, N = bench.workload(workload)
    opt.dropout = [dropout] * 4                               # 0: deterministic comparisons
    B, T = opt.batch_size, int(seq) if seq else opt.time_len
    eng = HipEngine(opt, 768, 74, 35, seq_len=T, bank_capacity=N, precision=precision, use_graph=use_graph, seed=1,
                    device_anchors=device_anchors)
    eng.load_params(synth.default_state([(n, tuple(v.shape)) for n, v in eng.params.items()], 0))
    batch = synth.synthetic_batch(B, T, seed=0)
    eng.set_batch(*batch)
    banks = synth.synthetic_banks(N, seed=0)
    eng.set_banks(*(banks[k] for k in "CFTAV"))
    return opt, N, batch, banks, eng


def _anchor_draw_checks(anc, N, m, seed, rng_step, seen):
    """anc [2, 6, m] = the draws of one step read back from mimrl_buffers.anchors: per call m DISTINCT rows in [0, N), equal to the
    restated selection (the m smallest (hash, row) keys with stream id 100 + stage and the RNG step of that stage: counters[0] is
    incremented once per stage), and no draw repeats an earlier one."""
    from tests.test_gpu_ops import _anchor_keys
    assert anc.shape == (2, 6, m)
    for st in range(2):
        for c in range(6):
            a = anc[st, c]
            assert a.min() >= 0 and a.max() < N and len(set(a.tolist())) == m, (st, c)
            want = (np.sort(_anchor_keys(N, seed, rng_step + st, 101 + st, c))[:m] & np.uint64(0xFFFFFFFF)).astype(np.int64)
            assert np.array_equal(a, want), ("device draw != restated selection", st, c)
            assert tuple(a.tolist()) not in seen, ("repeated draw", st, c)
            seen.add(tuple(a.tolist()))


@pytest.mark.parametrize("workload", ["cfg2", "cfg3"])
def test_device_drawn_anchors_step_vs_oracle(workload):
    """THE mode bench.py times: bf16, every fused kernel, one hipGraph per two-stage step, overlap mode, kNN anchors
    drawn ON THE DEVICE inside the step (Model.py:81: np.random.choice(range(N), m, replace=False) per CMI estimator; dropout off so that
    the oracle can follow).  After each step the draws are read back and (a) checked -- m distinct rows in [0, N) per call, different for
    each of the 6 calls x 2 stages x 2 steps, equal to the restated hash selection -- and (b) FED TO THE ORACLE, whose own kNN (exact
    float64 brute force / scikit-learn, anchors excluded: Model.py:83-86) and estimators then have to reproduce both losses and all
    11 + 8 MI / CMI values of the step: a sampler that returned duplicate, constant or out-of-range rows, a kNN that did not exclude
    exactly those rows, or a stage that used another stage's draw would all fail here.
    Stage 1 and the task loss of every step are compared with the oracle on the parameters the step started from; the stage-2 values
    with the oracle on (those main parameters, the critics the step's stage-1 update produced) -- for step 0 at cfg2 the critic update
    itself is the oracle's too (R.two_stage_step from the initial state: the oracle's fp32 critic update and the engine's bf16 one differ
    by Adam sign flips at lr 4e-3, which moves the CMI-derived stage-2 terms by 6.5e-3; band 2e-2).  Bands: 3x the errors measured in
    round 4 (profiles/r04_step_errors.json, "device_anchors": cfg2 losses <= 6.2e-5 relative, stage-1 MI / CMI <= 2.0e-4, stage-2 terms
    <= 3.4e-4 absolute; cfg3 1.5e-5 / 2.8e-5 / 7.3e-5)."""
    opt, N, batch, banks, eng = _bench_engine(workload, "bf16", True, device_anchors=True)
    eng.set_stage2_prefetch(True)
    m = opt.batch_size // opt.k_neighbor
    tb = tuple(torch.from_numpy(x) for x in batch)
    bk = {k: torch.from_numpy(v) for k, v in banks.items()}
    crit = [n for n in eng.params if R.is_critic_param(n)]
    main = [n for n in eng.params if not R.is_critic_param(n)]
    big = workload == "cfg3"
    band = dict(loss=5e-5, mi1=1e-4, mi2=2.5e-4, step2=1e-3) if big else dict(loss=2e-4, mi1=6e-4, mi2=1e-3, step2=2e-2)
    seen = set()
    errs = {}
    wire_z = {"ac_t": "T", "ta_c": "C", "vc_t": "T", "tv_c": "C", "tc_a": "A", "tc_v": "V"}      # the bank each call searches (Model.py:323-339)
    checks = []
    for step in range(1 if big else 2):
        p0 = {n: v.detach().cpu().clone() for n, v in eng.params.items()}
        eng.step()
        torch.cuda.synchronize()
        anc = eng.anchors.cpu().numpy().astype(np.int64)
        _anchor_draw_checks(anc, N, m, 1, 2 * step + 1, seen)
        # the neighbour rows the step's product samples were gathered from = the oracle's kNN on those anchors (anchors excluded)
        for st in (1, 2):
            got = eng.probe_knn(st).cpu().numpy()
            for c, name in enumerate(R.VCMI_NAMES):
                Z = banks[wire_z[name]]
                want = R.knn_indices(Z, anc[st - 1, c], opt.k_neighbor)
                assert not np.isin(got[c], anc[st - 1, c]).any(), ("an anchor row among the neighbours", st, name)
                if not np.array_equal(got[c], want):         # only exact fp32 ties may differ
                    Z64 = Z.astype(np.float64)
                    for i, j in zip(*np.nonzero(got[c] != want)):
                        dg = ((Z64[got[c][i, j]] - Z64[anc[st - 1, c][i]]) ** 2).sum()
                        dr = ((Z64[want[i, j]] - Z64[anc[st - 1, c][i]]) ** 2).sum()
                        assert abs(dg - dr) <= 1e-6 * dr + 1e-30, (st, name, i, j)
        s = eng.read_scalars().copy()
        p_mix = dict(p0)
        for n in crit:
            p_mix[n] = eng.params[n].detach().cpu().clone()
        with torch.no_grad():
            l1, mis1, *_ = R.stage_loss(p0, opt, 1, tb, bk, anc[0])
            l2, mis2, pred, feats, task = R.stage_loss(p_mix, opt, 2, tb, bk, anc[1])
        want1 = np.array([x.item() for x in mis1]); want2 = np.array([x.item() for x in mis2])
        e = dict(s1_loss=abs(s[_lib.S1_LOSS] - l1.item()) / abs(l1.item()), s2_loss=abs(s[_lib.S2_LOSS] - l2.item()) / abs(l2.item()),
                 task=abs(s[_lib.S2_TASK] - task.item()) / abs(task.item()),
                 mi1=float(np.abs(s[_lib.S1_MIS:_lib.S1_MIS + 11] - want1).max()), mi2=float(np.abs(s[_lib.S2_MIS:_lib.S2_MIS + 8] - want2).max()))
        checks += [(s[_lib.S1_LOSS], l1.item(), band["loss"], 1e-6, f"step {step}: stage-1 loss"),
                   (s[_lib.S1_MIS:_lib.S1_MIS + 11], want1, 0, band["mi1"], f"step {step}: stage-1 MI/CMI"),
                   (s[_lib.S2_TASK], task.item(), band["loss"], 1e-6, f"step {step}: task loss"),
                   (s[_lib.S2_LOSS], l2.item(), band["loss"], 1e-6, f"step {step}: stage-2 loss"),
                   (s[_lib.S2_MIS:_lib.S2_MIS + 8], want2, 0, band["mi2"], f"step {step}: stage-2 MI terms")]
        if step == 0 and not big:      # the whole step from the oracle alone (its own critic update between the stages)
            pq = {n: v.clone() for n, v in p0.items()}
            r1, r2 = R.two_stage_step(pq, opt, R.AdamState(pq, crit), R.AdamState(pq, main), tb, bk, anc[0], anc[1])
            w2 = np.array([x.item() for x in r2["mis"]])
            e.update(oracle_step_s2_loss=abs(s[_lib.S2_LOSS] - r2["loss"].item()) / abs(r2["loss"].item()),
                     oracle_step_mi2=float(np.abs(s[_lib.S2_MIS:_lib.S2_MIS + 8] - w2).max()))
            checks += [(s[_lib.S1_LOSS], r1["loss"].item(), band["loss"], 1e-6, "stage-1 loss vs oracle step"),
                       (s[_lib.S2_LOSS], r2["loss"].item(), band["loss"], 1e-6, "stage-2 loss vs oracle step"),
                       (s[_lib.S2_MIS:_lib.S2_MIS + 8], w2, 0, band["step2"], "stage-2 MI terms vs oracle step")]
        errs[str(step)] = e
        print(f"device-anchor step {workload}/{step}: {e}")
    eng.close()
    _record_errors("device_anchors/" + workload, errs)
    for got, want, rtol, atol, msg in checks:
        assert_close(got, want, rtol, atol, msg)


@pytest.mark.parametrize("stage", [1, 2])
def test_fused_concat_backward_matches_gemm_chain(stage, monkeypatch):
    """cfg2 with the concat critic (B = 128: 5 x 16,384 pair rows), bf16: the fused forward + fused data-gradient chain of the critic
    tail against pair_expand + GEMM chain + top1_bwd + pair_reduce (MIMRL_NO_FUSED_CONCAT=1).  Stage 1: every critic gradient
    (weights through the bf16-stored dZ operands, biases / score head through the in-kernel column sums); stage 2: every main-model
    gradient (through dP / dQ -> the feature gradients).  Same bf16-rounded MFMA operands on both sides except dZ, which the fused
    kernel rounds once when it stores it for the weight-gradient GEMMs (the chain keeps fp32 and the GEMM rounds it on load: the
    same value).  What differs is the forward summation order, and with 4.2e7 hidden activations per pass ~1e3 of them sit within
    fp32 noise of the ReLU kink: their masks differ between the two runs, a relative perturbation of ~4e-3 of the
    gradient signal.  Critic gradients (stage 1) agree to 3e-3 of their scale; the main-model gradients (stage 2), which the bf16
    model backward amplifies ~10x (LayerNorm cancellation), to 5e-2 -- a wrong dP / dQ would be O(1).
    (Chain vs chain on the same inputs: 1e-6, tools/concat_ab.py.)"""
    rng = np.random.default_rng(5)
    res = {}
    for tag in ("fused", "chain"):
        if tag == "chain":
            monkeypatch.setenv("MIMRL_NO_FUSED_CONCAT", "1")
        else:
            monkeypatch.delenv("MIMRL_NO_FUSED_CONCAT", raising=False)
        opt, N, batch, banks, eng = _bench_engine("cfg2-concat", "bf16", False, device_anchors=False)
        if tag == "fused":
            m = opt.batch_size // opt.k_neighbor
            anchors = np.stack([rng.choice(N, size=m, replace=False) for _ in range(6)])
        eng.set_anchors(stage, anchors)
        eng.stage_grads(stage)
        torch.cuda.synchronize()
        pick = (lambda n: "MLP_f" in n) if stage == 1 else (lambda n: not n.startswith("vmi") and not n.startswith("vcmi"))
        res[tag] = (eng.read_scalars().copy(), {n: v.double().cpu().numpy().copy() for n, v in eng.grads.items() if pick(n)})
        eng.close()
    (sa, ga), (sb, gb) = res["fused"], res["chain"]
    off = _lib.S1_MIS if stage == 1 else _lib.S2_MIS
    assert_close(sa[off:off + 4], sb[off:off + 4], 1e-4, 1e-5, "MI values fused vs chain")
    assert len(ga) >= 40
    for n in ga:
        grad_close(ga[n], gb[n], 3e-3 if stage == 1 else 5e-2, n)


REPRO = [("cfg2", 2, False), ("cfg1", 2, False), ("cfg2", 2, True), ("cfg2", 1, False), ("cfg2", 1, True), ("cfg2-concat", 1, False),
         ("cfg2-concat", 2, True), ("cfg3", 1, False), ("cfg3", 2, True), ("cfg5", 2, False), ("cfg2@49", 2, True), ("cfg2@1", 2, False)]


@pytest.mark.parametrize("workload,stage,graph", REPRO, ids=[f"{w}-s{s}-{'graph' if g else 'eager'}" for w, s, g in REPRO])
def test_gradients_reproducible(workload, stage, graph):
    """Fresh engine, same inputs, the mode bench.py runs: EVERY gradient tensor of the stage must come out the same three times
    (float atomics reorder additions: observed <= 3e-6 of the tensor scale with the separable critic, <= 2.7e-4 in the concat critic's
    split-K weight gradients over B*B rows; bands 1e-4 (3e-4 at cfg3 / cfg5) / 1e-3) -- both stages, eager and captured, odd T (49, 1),
    separable and concat critics, cfg3 (T = 500) and cfg5 (T = 1000) shapes.  Round 2b found the block-0 K-axis parameter gradients
    off by 5-30 % from run to run while their kernel ran beside the layer-1 BPTT (the structural fix keeps
    register-heavy kernels away from the recurrence, tests/test_codeobj.py pins the register facts it relies on) -- every parity test
    passed at the time, because bf16-vs-fp32 bands are wider than that.  (tools/determinism.py is the same check as a CLI.)"""
    runs, anchors = [], None
    for r in range(3):
        opt, N, batch, banks, eng = _bench_engine(workload, "bf16", graph, device_anchors=False)
        if anchors is None:
            rng = np.random.default_rng(5)
            anchors = np.stack([rng.choice(N, size=opt.batch_size // opt.k_neighbor, replace=False) for _ in range(6)])
        eng.set_anchors(stage, anchors)
        eng.stage_grads(stage)
        torch.cuda.synchronize()
        runs.append({n: v.double().cpu().numpy().copy() for n, v in eng.grads.items() if n.startswith("v") == (stage == 1)})
        eng.close()
    top = max(np.abs(v).max() for v in runs[0].values())
    concat = "concat" in workload or workload.startswith("cfg3")
    for r in (1, 2):
        for n in runs[0]:
            # Per-tensor band 1e-4 of the tensor's own scale (observed <= 3e-6: float atomics reorder additions); only the concat critic's
            # tensors -- split-K weight gradients over B * B pair rows, observed 2.7e-4 -- get 1e-3 (the round-3 version applied
            # 1e-3 and a 1 %-of-the-largest-tensor floor to everything, under which a K-axis-class bug in a small tensor could pass).
            # (A tensor whose gradient cancels to nothing is fp32 noise: the score head's bias of a concat critic has gradient
            #  sum(dS) = 0 under InfoNCE's shift invariance -- 5 x 65536 terms at cfg3.)
            wide = concat and ".MLP_f." in n
            if wide and n.endswith("MLP_f.6.bias"):
                continue
            # cfg3 / cfg5 sum each recurrence weight gradient over T * B = 128 000+ rows with float atomics: 1.0006e-4 of the tensor's scale
            # was seen once in eight runs (rnn_v.weight_ih_l0_reverse), so the long shapes get 3e-4 -- the bug class this guards is 5-30 %.
            band = 1e-3 if wide else 3e-4 if workload[:4] in ("cfg3", "cfg5") else 1e-4
            scale = max(np.abs(runs[0][n]).max(), (1e-2 if wide else 1e-3) * top)
            assert np.abs(runs[r][n] - runs[0][n]).max() <= band * scale, (r, n, np.abs(runs[r][n] - runs[0][n]).max() / scale)


DET = [("cfg2", 2, True), ("cfg2", 1, False), ("cfg2-concat", 2, True), ("cfg2-concat", 1, False), ("cfg2@49", 2, False), ("cfg3", 2, True)]


@pytest.mark.parametrize("workload,stage,graph", DET, ids=[f"{w}-s{s}-{'graph' if g else 'eager'}" for w, s, g in DET])
def test_deterministic_build_is_bit_exact(workload, stage, graph, tmp_path):
    """MIMRL_DETERMINISTIC=1 (libmimrl_hip_det.so: csrc/det.h -- every float atomic of the default build is order-independent 64-bit
    fixed-point accumulation there, one stream; the reference's switch is torch.backends.cudnn.deterministic, Main.py:19-20):
    tests/det_worker.py, in its own process, finds (a) every gradient tensor of the stage bit-identical over three fresh engines and (b)
    every parameter, Adam moment and scalar bit-identical after three full steps (device-drawn anchors, dropout 0.1) of two fresh engines.
    Here: (c) those gradients agree with the DEFAULT build's to 2e-3 of each tensor's scale (the sums are the exact ones rounded once; the
    default build rounds per float atomic)."""
    import subprocess
    out = str(tmp_path / "det_grads.npz")
    env = dict(os.environ, MIMRL_DETERMINISTIC="1", PYTHONPATH=ROOT)
    env.pop("MIMRL_LIB_PATH", None)
    r = subprocess.run([sys.executable, os.path.join(HERE, "det_worker.py"), workload, str(stage), "1" if graph else "0", out], env=env,
                       cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "grads bit-identical" in r.stdout and "3 steps bit-identical" in r.stdout, r.stdout
    det = dict(np.load(out))
    assert not _lib.DETERMINISTIC
    opt, N, batch, banks, eng = _bench_engine(workload, "bf16", graph, device_anchors=False)
    rng = np.random.default_rng(5)
    eng.set_anchors(stage, np.stack([rng.choice(N, size=opt.batch_size // opt.k_neighbor, replace=False) for _ in range(6)]))
    eng.stage_grads(stage)
    torch.cuda.synchronize()
    mine = {n: v.double().cpu().numpy().copy() for n, v in eng.grads.items() if n.startswith("v") == (stage == 1)}
    eng.close()
    assert set(mine) == set(det)
    top = max(np.abs(v).max() for v in mine.values())
    concat = "concat" in workload or workload.startswith("cfg3")
    for n in mine:
        wide = concat and ".MLP_f." in n
        if wide and n.endswith("MLP_f.6.bias"):
            continue
        # (not the run-to-run band: an intermediate that differs in its last bit -- one rounding of the exact sum instead of one per float
        #  atomic -- flips a bf16 rounding somewhere downstream now and then; measured 4.4e-4 of the tensor scale in the layer-0 GRU weights,
        #  the deepest tensors, against 2e-3 for the bench mode vs the rounded-operand oracle)
        # T = 500 (cfg3): 2.7e-3 measured through ten times as many recurrence steps; band 1e-2, against 6e-2 for that shape vs the oracle
        band = 1e-2 if workload[:4] in ("cfg3", "cfg5") else 5e-3 if wide else 2e-3
        scale = max(np.abs(mine[n]).max(), (1e-2 if wide else 1e-3) * top)
        assert np.abs(det[n] - mine[n]).max() <= band * scale, (n, np.abs(det[n] - mine[n]).max() / scale)


def test_cfg3_full_size_properties(monkeypatch):
    """BASELINE configs[2] at FULL size (MOSEI-shaped B=256, T=500, concat critic, k=2, N=16326 banks): no reference run is
    affordable at this size, so size-independent properties: (a) the epoch-0 rule, (b) the task loss and the features against
    the oracle's forward pass on the same inputs (fp32: 1e-3), (c) bf16 + fused + graph + overlap == bf16 unfused eager
    sequential within the bf16 band, (d) three updates stay finite, (e) workspace fits and anchors are under the 16384 cap."""
    opt, N, batch, banks, eng = _bench_engine("cfg3", "fp32", False, device_anchors=False)
    assert N < 16384 and eng.workspace_bytes() < 64 * 2 ** 30
    # (a) epoch-0 rule
    eng.set_banks(None, None, None, None, None)
    before = eng.crit["p"].clone()
    eng.stage1_step()
    eng.forward(train=True, with_losses=True)
    s0 = eng.read_scalars()
    assert s0[_lib.S1_LOSS] == 0.0 and torch.equal(before, eng.crit["p"]) and np.all(s0[_lib.S2_MIS:_lib.S2_MIS + 8] == 0)
    # (b) forward vs the oracle (one CPU forward pass of B=256, T=500: seconds)
    p = {n: v.detach().cpu() for n, v in eng.params.items()}
    tb = tuple(torch.from_numpy(x) for x in batch)
    with torch.no_grad():
        pred, F_F, T_F, A_F, V_F = R.model_forward(p, opt, *tb[:3])
        task = R.task_loss_mae(pred, tb[3])
    assert_close(s0[_lib.S2_TASK], task.item(), 1e-3, 1e-6, "task loss vs oracle")
    assert_close(s0[_lib.S2_LOSS], task.item(), 1e-3, 1e-6, "epoch-0 stage-2 loss = task loss")
    feats = eng.feats.cpu().numpy()
    for i, (k, want) in enumerate((("F_F", F_F), ("T_F", T_F), ("A_F", A_F), ("V_F", V_F))):
        assert_close(feats[i], want.numpy(), 1e-3, 5e-5, k)
    eng.close()
    # (c) + (d): the benchmarked mode vs the plain mode, same anchors
    rng = np.random.default_rng(3)
    m = opt.batch_size // opt.k_neighbor
    anchors = [np.stack([rng.choice(N, size=m, replace=False) for _ in range(6)]) for _ in range(2)]
    res = {}
    for tag, graph, pre, envs in (("plain", False, False, ("MIMRL_NO_FUSED_CUBE", "MIMRL_NO_FUSED_CUBE_BWD", "MIMRL_NO_FUSED_MLP", "MIMRL_NO_FUSED_MI",
                                                           "MIMRL_NO_FUSED_CONCAT")),
                                  ("bench", True, True, ())):
        for e in ("MIMRL_NO_FUSED_CUBE", "MIMRL_NO_FUSED_CUBE_BWD", "MIMRL_NO_FUSED_MLP", "MIMRL_NO_FUSED_MI", "MIMRL_NO_FUSED_CONCAT"):
            monkeypatch.delenv(e, raising=False)
        for e in envs:
            monkeypatch.setenv(e, "1")
        opt, N, batch, banks, eng = _bench_engine("cfg3", "bf16", graph, device_anchors=False)
        eng.set_anchors(1, anchors[0]); eng.set_anchors(2, anchors[1])
        eng.set_stage2_prefetch(pre)
        eng.step()
        torch.cuda.synchronize()
        res[tag] = (eng.read_scalars().copy(), eng.pred.cpu().numpy().copy())
        if tag == "bench":
            eng.step(); eng.step()
            assert np.isfinite(eng.read_scalars()).all() and torch.isfinite(eng.main["p"]).all() and torch.isfinite(eng.crit["p"]).all()
        eng.close()
    (sa, pa), (sb, pb) = res["plain"], res["bench"]
    assert_close(sb[_lib.S1_LOSS], sa[_lib.S1_LOSS], 1e-2, 1e-3, "stage-1 loss")
    assert_close(sb[_lib.S2_LOSS], sa[_lib.S2_LOSS], 1e-2, 1e-3, "stage-2 loss")
    assert_close(sb[_lib.S2_MIS:_lib.S2_MIS + 8], sa[_lib.S2_MIS:_lib.S2_MIS + 8], 3e-2, 3e-2, "MI terms")
    assert_close(pb, pa, 3e-2, 2e-2, "predictions")


def test_cfg3_full_size_mi_values_vs_oracle():
    """BASELINE configs[2] at FULL size (B=256, T=500, concat critic: 5 x 65,536 pair rows through 256-256-256-1; N=16326 banks, k=2),
    fp32: both stage losses and all 11 + 8 MI / CMI values against the oracle under no_grad (VMI.py:58-65, Model.py:305-386) -- the
    reference's B*B pair expansion, its host kNN and the CMI classifiers at the size the benchmark's MFMA fraction is quoted on.
    (Gradients of this configuration are pinned with only the batch reduced: cfg3_small.)"""
    opt, N, batch, banks, eng = _bench_engine("cfg3", "fp32", False, device_anchors=False)
    rng = np.random.default_rng(6)
    m = opt.batch_size // opt.k_neighbor
    anchors = [np.stack([rng.choice(N, size=m, replace=False) for _ in range(6)]) for _ in range(2)]
    eng.set_anchors(1, anchors[0]); eng.set_anchors(2, anchors[1])
    p = {n: v.detach().cpu().clone() for n, v in eng.params.items()}
    eng.stage_grads(1)
    eng.stage_grads(2)
    torch.cuda.synchronize()
    s = eng.read_scalars()
    tb = tuple(torch.from_numpy(x) for x in batch)
    bk = {k: torch.from_numpy(v) for k, v in banks.items()}
    with torch.no_grad():
        l1, mis1, *_ = R.stage_loss(p, opt, 1, tb, bk, anchors[0])
        l2, mis2, pred, feats, task = R.stage_loss(p, opt, 2, tb, bk, anchors[1])
    assert_close(s[_lib.S1_LOSS], l1.item(), 1e-3, 1e-5, "stage-1 loss")
    assert_close(s[_lib.S1_MIS:_lib.S1_MIS + 11], [x.item() for x in mis1], 1e-3, 2e-5, "stage-1 MI/CMI")
    assert_close(s[_lib.S2_TASK], task.item(), 1e-3, 1e-6, "task loss")
    assert_close(s[_lib.S2_LOSS], l2.item(), 1e-3, 1e-5, "stage-2 loss")
    assert_close(s[_lib.S2_MIS:_lib.S2_MIS + 8], [x.item() for x in mis2], 1e-3, 5e-5, "stage-2 MI terms")
    assert_close(eng.pred.cpu().numpy(), pred.numpy().reshape(-1), 1e-3, 1e-5, "predictions")
    eng.close()


def test_cfg5_full_size_vs_oracle():
    """BASELINE configs[4], reference-supported subset at FULL size (B=32, T=1000, gru, d_common=128, fp32: 2000 serial cell
    steps per pass): both stage losses, all 11 + 8 MI/CMI terms and the predictions against the oracle's forward passes
    (no_grad: the oracle's autograd through 4000 Python-level cell steps takes minutes; gradients at T=1000 are pinned by
    the cfg5_small fixture), then three updates stay finite."""
    opt, N, batch, banks, eng = _bench_engine("cfg5", "fp32", False, device_anchors=False)
    rng = np.random.default_rng(4)
    m = opt.batch_size // opt.k_neighbor
    anchors = [np.stack([rng.choice(N, size=m, replace=False) for _ in range(6)]) for _ in range(2)]
    eng.set_anchors(1, anchors[0]); eng.set_anchors(2, anchors[1])
    p = {n: v.detach().cpu().clone() for n, v in eng.params.items()}
    eng.stage_grads(1)
    eng.stage_grads(2)
    torch.cuda.synchronize()
    s = eng.read_scalars()
    tb = tuple(torch.from_numpy(x) for x in batch)
    bk = {k: torch.from_numpy(v) for k, v in banks.items()}
    with torch.no_grad():
        l1, mis1, *_ = R.stage_loss(p, opt, 1, tb, bk, anchors[0])
        l2, mis2, pred, feats, task = R.stage_loss(p, opt, 2, tb, bk, anchors[1])
    assert_close(s[_lib.S1_LOSS], l1.item(), 1e-3, 1e-5, "stage-1 loss")
    assert_close(s[_lib.S1_MIS:_lib.S1_MIS + 11], [x.item() for x in mis1], 1e-3, 2e-5, "stage-1 MI/CMI")
    assert_close(s[_lib.S2_TASK], task.item(), 1e-3, 1e-6, "task loss")
    assert_close(s[_lib.S2_LOSS], l2.item(), 1e-3, 1e-5, "stage-2 loss")
    assert_close(s[_lib.S2_MIS:_lib.S2_MIS + 8], [x.item() for x in mis2], 1e-3, 5e-5, "stage-2 MI terms")
    assert_close(eng.pred.cpu().numpy(), pred.numpy().reshape(-1), 1e-3, 1e-5, "predictions")
    for _ in range(3):
        eng.step()
    assert np.isfinite(eng.read_scalars()).all() and torch.isfinite(eng.main["p"]).all() and torch.isfinite(eng.crit["p"]).all()
    eng.close()


def test_bench_mode_trains_like_fp32():
    """Does the benchmarked mode (bf16 MFMA operands, every fused kernel, hipGraph, Solver.step overlap) TRAIN like fp32?
    100 two-stage iterations over 4 cycling cfg1-shaped batches (lr 4e-3, dropout off, shared host anchors) from one
    initialisation in three runs: fp32 eager sequential, fp32 graph+overlap (same arithmetic, different summation order: the
    chaos floor of this adversarial objective -- Adam's ~lr*sign(g) steps amplify last-ulp differences), and the bench mode.
    The bench mode must (a) train: task MAE falls below half its initial window, and (b) stay as close to fp32 as fp32 stays
    to itself: window-mean gaps (task MAE, stage-1 loss, 8 MI/CMI series) within 3x the floor + a small band.
    Measured (tools/bf16_convergence.py): final-window task MAE 0.30 / 0.26 / 0.36; see DESIGN.md section 2."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("bf16_convergence", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                                                   "tools", "bf16_convergence.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    win = lambda r: r.reshape(5, 20, -1).mean(1)
    # (one fp32 pair is a noisy estimate of the floor: the bands are the fp32-vs-fp32 gaps seen over several boxes)
    band = np.array([0.25, 0.6, 0.3, 0.15, 0.15, 0.6, 2.5, 2.5, 2.5, 2.5])    # task, s1 loss, f_t f_a f_v inv | spec_t spec_a spec_v comp
    # ONE attempt (round 3).  Through round 2 the gap criterion was accepted in one of three attempts: a single (fp32, fp32, bf16)
    # triple landed outside 3x its own floor + band about once in ten.  With fp16 instead of bf16 forward operands on the model path
    # (cube_fused.hip) the worst gap / (3 floor + band) over twelve independent attempts is 0.78 (tools/_t.py-style sweep, median
    # 0.42), and the bench mode's final-window task MAE (0.14-0.29) is at or below fp32's (0.21-0.34).
    a, b, c = mod.run("fp32", False, False), mod.run("fp32", True, True), mod.run("bf16", True, True)
    assert np.isfinite(c).all()
    wa, wb, wc = win(a), win(b), win(c)
    for tag, w in (("fp32 eager", wa), ("fp32 graph+overlap", wb), ("bench mode", wc)):
        assert w[-1, 0] < 0.5 * w[0, 0], f"{tag}: task MAE {w[0, 0]:.3f} -> {w[-1, 0]:.3f} did not halve in 100 steps"
    floor = np.abs(wa - wb).max(0)
    gap = np.minimum(np.abs(wc - wa), np.abs(wc - wb)).max(0)
    assert np.all(gap <= 3 * floor + band), ("bench-vs-fp32 gaps outside 3x the fp32-vs-fp32 floor + band (gap, floor)",
                                             np.round(gap, 3).tolist(), np.round(floor, 3).tolist())


def test_bf16_stored_bptt_outputs_change_nothing(monkeypatch):
    """bf16 mode stores the BPTT outputs dg[B,T,4H] / h_prev of GRU layer 1 as bf16: their only consumers (dh0, dW_ih, dW_hh
    products) round their operands to bf16 anyway, so the gradients must equal the fp32-stored run up to the summation-order
    noise of the split-K atomics (MIMRL_DG_FP32=1 = the fp32-stored run)."""
    res = {}
    monkeypatch.setenv("MIMRL_NO_XIN", "1")   # (the fused layer-0 projection exists with bf16 dg only: keep the forward pass the same in both runs)
    monkeypatch.setenv("MIMRL_REC16", "0")    # (round 5b: 16-bit stored dout / h_prev exist with bf16 dg only and are new rounding points: tests/test_gpu_fused_oracle.py)
    monkeypatch.setenv("MIMRL_DWIH_H16", "0") # (... and so does the layer-1 dW_ih product's fp16 -> bf16 operand: test_16bit_stored_projection_operands_change_nothing)
    for tag, env in (("bf16", None), ("fp32", "1")):
        if env:
            monkeypatch.setenv("MIMRL_DG_FP32", env)
        else:
            monkeypatch.delenv("MIMRL_DG_FP32", raising=False)
        c, opt, batch, banks, p, eng = make_engine("cfg1_sep", precision="bf16")
        g = load_golden("cfg1_sep")
        eng.set_banks(*(banks[k] for k in "CFTAV"))
        eng.set_anchors(2, g["anchors"][0, 1])
        eng.stage_grads(2)
        torch.cuda.synchronize()
        res[tag] = {n: eng.grads[n].double().cpu().numpy().copy() for n in eng.grads if n.startswith("rnn_")}
        eng.close()
    assert len(res["bf16"]) == 32
    for n, ga in res["bf16"].items():
        gb = res["fp32"][n]
        scale = np.abs(gb).max() + 1e-30
        assert np.abs(ga - gb).max() <= 2e-5 * scale, f"{n}: rel diff {np.abs(ga - gb).max() / scale:.2e}"


@pytest.mark.parametrize("workload", ["cfg2", "cfg1"])
def test_critic_update_with_fragment_images_and_stage_boundary_changes_nothing(workload, monkeypatch):
    """Round 5b: in the combined (captured) step the critic clip + Adam launch writes the forward fragment-order images of the estimator stacks
    itself and its workgroup 0 does the stage boundary (finalize_stage1 + begin_stage(2) + MAE: Model.py:341, Solver.py:181-182) -- estimator_ops.hip:
    adam8_kernel, AdamArgs::frag / ::sb.  Against the round-5a sequence (MIMRL_ADAM_FRAG=0: scalar Adam kernel, frag_images on side 3, stage_boundary
    as a launch) a step must leave the same 64 scalars and predictions up to the run-to-run noise of the float atomics in the gradients (device-drawn
    anchors: the RNG step counters must have advanced identically, or the draws -- and everything else -- differ); a second step, which runs on
    the parameters and images the first one left, stays within the band that noise grows to (measured 1.9e-4 on values of 0.1 - 4: Adam's first
    steps are sign-like, a last-bit difference in a near-zero gradient moves a parameter by 2 lr -- ~2 % of them after two steps)."""
    res = {}
    for tag, env in (("in_adam", None), ("launches", "0")):
        if env:
            monkeypatch.setenv("MIMRL_ADAM_FRAG", env)
        else:
            monkeypatch.delenv("MIMRL_ADAM_FRAG", raising=False)
        opt, N, batch, banks, eng = _bench_engine(workload, "bf16", True)
        sc = []
        for _ in range(2):
            eng.step()
            torch.cuda.synchronize()
            sc.append(eng.read_scalars().copy())
        res[tag] = (np.stack(sc), {n: v.double().cpu().numpy().copy() for n, v in eng.params.items()})
        eng.close()
    a, b = res["in_adam"], res["launches"]
    assert np.isfinite(a[0]).all() and np.abs(a[0][0]).max() > 1 and a[0][0][33] > 0      # (33 = MIMRL_S2_TASK: the boundary's MAE)
    assert np.allclose(a[0][0], b[0][0], rtol=1e-5, atol=2e-6), np.abs(a[0][0] - b[0][0]).max()
    assert np.allclose(a[0][1], b[0][1], rtol=2e-3, atol=2e-3), np.abs(a[0][1] - b[0][1]).max()
    for n, pa in a[1].items():
        assert np.isfinite(pa).all() and np.abs(pa - b[1][n]).max() <= 2 * 2 * 4e-3 + 1e-6, n      # (at most two sign-like Adam steps apart)


def _registered_knobs():
    """(name, help) of every row of csrc/knobs.cpp's table -- the one list of environment knobs the native library reads."""
    import re
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mimrl_amd", "csrc", "knobs.cpp")).read()
    return re.findall(r'^\s*\{"(MIMRL_[A-Z0-9_]+)",\s*"[^"]*",\s*"[^"]*",\s*"((?:[^"\\]|\\.)*)"\}', src, flags=re.M)


# the value each knob is exercised with (default "1"; knobs whose help starts with "0:" are switched with "0")
_KNOB_VALUES = {"MIMRL_GRU_WAVES": "8", "MIMRL_GRU_LDS_PAD": "64", "MIMRL_GEMM_TALL_MIN_M": "64", "MIMRL_DBG_DELAY_TAG": "7:20",
                "MIMRL_DDP_SPLIT": "0"}
_KNOB_RUNS = {}


def _knob_value(knob, help_):
    return _KNOB_VALUES.get(knob, "0" if help_.startswith("0:") else "1")


def _knob_run(tmp_path_factory, knob, value):
    """Result of the worker under (knob, value).  The FIRST call runs the whole table (and the default run) four processes at a time -- one
    process per knob because the library reads most of them once per process -- and caches it: ~40 x 10 s of mostly host time."""
    if not _KNOB_RUNS:
        from concurrent.futures import ThreadPoolExecutor
        root = tmp_path_factory.mktemp("knobs")

        def one(kv):
            k, v = kv
            d = root / (k or "default")
            d.mkdir()
            env = dict(os.environ, PYTHONPATH=ROOT, OMP_NUM_THREADS="8")
            for name in [n for n in env if n.startswith("MIMRL_") and n != "MIMRL_LIB_PATH"]:
                env.pop(name)
            if k:
                env[k] = str(d / "graph.dot") if k == "MIMRL_GRAPH_DOT" else v
            r = subprocess.run([sys.executable, os.path.join(HERE, "knob_worker.py"), str(d / "g.npz")], env=env, capture_output=True, text=True, timeout=420)
            return (k, v), (r, d)

        jobs = [(None, "")] + [(n, _knob_value(n, h)) for n, h in _registered_knobs()]
        with ThreadPoolExecutor(4) as ex:
            _KNOB_RUNS.update(dict(ex.map(one, jobs)))
    return _KNOB_RUNS[(knob, value)]


def test_knob_table_is_small_and_complete():
    """Round 6: <= 40 registered knobs (99 in round 5: the measured-and-lost kernels and the capture-order switches whose
    optimum is known are gone), and every knob() site in the sources names a registered row (an unregistered read is only reported at run time)."""
    import glob
    import re
    names = [n for n, _ in _registered_knobs()]
    assert 20 <= len(names) <= 40 and len(set(names)) == len(names), len(names)
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mimrl_amd", "csrc")
    read = set()
    for f in glob.glob(os.path.join(root, "*.hip")) + glob.glob(os.path.join(root, "*.h")) + glob.glob(os.path.join(root, "*.cpp")):
        if not f.endswith("knobs.cpp"):
            read |= set(re.findall(r'knob(?:_on|_int)?\("(MIMRL_[A-Z0-9_]+)"', open(f).read()))
    assert read == set(names), (sorted(read - set(names)), sorted(set(names) - read))


@pytest.mark.parametrize("knob,help_", _registered_knobs(), ids=[n for n, _ in _registered_knobs()])
def test_every_registered_knob_vs_oracle(knob, help_, tmp_path_factory):
    """Every environment knob the library still reads (csrc/knobs.cpp) is a pinned path: with the knob set, the bf16 bench mode on `tiny_sep`
    and `cfg1_cat` gives stage losses / MI / CMI values within the bf16 bands of the ORACLE (autograd of oracle/mimrl_ref.py, itself held to the
    reference's fixtures), gradient buckets at cosine >= 0.995 / 0.98 to the oracle's, and every gradient tensor no
    further from the oracle's than 5x the default run's distance (floor 8e-2 of the tensor's norm, cap 0.5: knobs that move the model's
    rounding points move the critics' near-cancelling bias gradients by 3-5e-2 at the tiny fixture); two captured-graph steps stay finite.
    (ADVICE r04 / r05 both found wrong gradients behind exactly such switches.)"""
    value = _knob_value(knob, help_)
    r0, d0 = _knob_run(tmp_path_factory, None, "")
    assert r0.returncode == 0 and "KNOB_OK" in r0.stdout, "default run: " + r0.stdout[-1500:] + r0.stderr[-3000:]
    r, d = _knob_run(tmp_path_factory, knob, value)
    assert r.returncode == 0 and "KNOB_OK" in r.stdout, f"{knob}={value}: " + r.stdout[-1500:] + r.stderr[-3000:]
    assert "is read but not registered" not in r.stderr
    if knob == "MIMRL_KNOBS":
        assert "environment knobs" in r.stderr and all(n in r.stderr for n, _ in _registered_knobs())
    if knob == "MIMRL_GRAPH_DOT":
        assert os.path.getsize(d / "graph.dot") > 1000
    a, b = np.load(d / "g.npz"), np.load(d0 / "g.npz")
    assert set(a.files) == set(b.files) and len(a.files) > 100
    ra, rb = json.loads(r.stdout.split("KNOB_OK ", 1)[1].splitlines()[0]), json.loads(r0.stdout.split("KNOB_OK ", 1)[1].splitlines()[0])
    for fx in ra:                                        # bucket direction vs the ORACLE: not worse than 3x the default run's distance (floors 5e-3 / 2e-2)
        for st, floor in (("cos_s1", 5e-3), ("cos_s2", 2e-2)):
            assert 1.0 - ra[fx][st] <= max(3.0 * (1.0 - rb[fx][st]), floor), f"{knob}={value}: {fx} {st} {ra[fx][st]} (default run {rb[fx][st]})"
    worst = ("", 0.0)
    for k in a.files:
        if not k.startswith("err|"):
            continue
        # per tensor: distance to the ORACLE's gradient (relative L2, floored at 5 % of the bucket's largest RMS -- knob_worker.py) against the
        # default run's distance for the same tensor: a knob may move rounding points (tensors that nearly cancel at the initial point then move
        # by tens of per cent in EVERY bf16 run, the default one included) but must not be further from the oracle than 5x the default run is
        ea, eb = float(a[k]), float(b[k])
        fac = 10.0 if knob == "MIMRL_FWD_BF16" else 5.0   # (bf16 instead of fp16 forward operands: 3 mantissa bits fewer BY DESIGN)
        kaxis = any(t in k for t in ("mlp_k.", "ln_k.", "res_projection_k"))   # 3 x 3 / 3-element tensors behind LayerNorms over K = 3: DESIGN section 2
        band = 0.5 if kaxis else min(max(fac * eb, 8e-2), 0.5)   # (the default run's own distance moves by 2x run to run on the ill-conditioned K-axis / bias tensors;
                                                         #  a WRONG tensor is ~1 from the oracle: never inside 0.5)
        ratio = ea / band
        if ratio > worst[1]:
            worst = (k, ratio)
        assert np.isfinite(ea) and ea <= band, f"{knob}={value}: {k[4:]} is {ea:.3e} from the oracle (default run {eb:.3e})"
    _record_errors(f"knob/{knob}={value}", {"worst_tensor": worst[0], "worst_error_over_band": worst[1], "vs_oracle": ra})


@pytest.mark.parametrize("graph", [False, True])
def test_split_stage2_gradients_equal_the_whole_pass(graph):
    """mimrl_stage_grads_part (the data-parallel split reduce, dist.ddp_stage2_split): part 0 + part 1 leave the SAME main gradient bucket as
    one mimrl_stage_grads(2) -- and after part 0 alone every range outside HipEngine.late_grad_ranges() (everything but the layer-0 GRU
    tensors) already holds its final value, which is what allows its all-reduce to start under part 1."""
    res = {}
    for tag in ("whole", "split"):
        opt, N, batch, banks, eng = _bench_engine("cfg2", "bf16", graph, device_anchors=False)
        rng = np.random.default_rng(5)
        anchors = np.stack([rng.choice(N, size=opt.batch_size // opt.k_neighbor, replace=False) for _ in range(6)])
        eng.set_anchors(2, anchors)
        if tag == "whole":
            eng.stage_grads(2)
            torch.cuda.synchronize()
        else:
            eng.stage_grads_part(2, 0)
            torch.cuda.synchronize()
            early = eng.main["g"].clone()
            eng.stage_grads_part(2, 1)
            torch.cuda.synchronize()
            late = eng.late_grad_ranges()
            # round 5: the layout puts rnn_v.*_l0* and rnn_a.*_l0* at the TAIL of the main bucket -- one late range, one early range
            assert len(late) == 1 and late[0][1] == early.numel() and sum(b - a for a, b in late) > 250000
            keep = torch.ones(early.numel(), dtype=torch.bool, device=early.device)
            for a, b in late:
                keep[a:b] = False
            assert torch.equal(early[keep], eng.main["g"][keep]), "a gradient outside the late ranges changed in part 1"
            assert (early[~keep] != eng.main["g"][~keep]).any(), "part 1 wrote nothing"
        res[tag] = {n: v.double().cpu().numpy().copy() for n, v in eng.grads.items() if not n.startswith("v")}
        eng.close()
    top = max(np.abs(v).max() for v in res["whole"].values())
    for n, want in res["whole"].items():
        scale = max(np.abs(want).max(), 1e-3 * top)
        assert np.abs(res["split"][n] - want).max() <= 1e-4 * scale, n      # (float atomics reorder additions: the reproducibility band)


# fp32 engine: per-tensor band on the stored slices (fraction of the tensor's own scale; 3x the measured 4.9e-3 / 1.7e-3) -- bench mode:
# (critic bucket, main bucket) cosine floors.  Measured: fp32 cosines 1 - 3e-8; bench 0.99981 / 0.99772 (cfg3_full), 0.99948 / 0.99751
# (cfg5_full): 3x the measured distance from 1.  (Round 5a measured 0.98935 / 0.97537 for the main bucket: the GEMM chain of the long-sequence
# L axis rounded the residual product's operands to bf16; the one-pass kernel of round 5b -- csrc/cube_long.hip -- rounds them to fp16 like every
# other forward product, and the timed mode's main-model gradient is within 3.9 / 4.0 degrees of the reference's instead of 8.4 / 12.8.)  Per TENSOR the bench mode is off by up to the tensor's whole scale in the small
# ill-conditioned ones at this depth (mlp_k.fc1.bias, ln_a.bias, single rows of rnn_a.weight_ih_l1: the backward of the broadcast means
# through LayerNorms over K = 3 / L cancels most of the signal) -- recorded in profiles/r05_step_errors.json, not asserted.
FULL_BANDS = {"cfg3_full": (1.5e-2, (0.9994, 0.9931)), "cfg5_full": (6e-3, (0.9984, 0.9925))}


@pytest.mark.parametrize("name", ["cfg3_full", "cfg5_full"])
@pytest.mark.parametrize("mode", ["fp32", "bench"])
def test_full_size_gradients_vs_reference(name, mode):
    """the gradients of BASELINE configs[2] at FULL size (cfg3_full: B = 256, T = 500, concat critic, N = 16326) and of the
    configs[4] subset at the size the step tests run (cfg5_full: B = 32, T = 1000) against the REAL reference -- tests/golden/make_golden.py
    ran the reference's own Solver-body once in the build container (minutes of CPU autograd) and stored, per tensor, the gradient norm,
    the gradient sum and a 512-entry strided slice, plus the 11 + 8 values and both losses (Model.py:305-386, Solver.py:205-236).
    fp32 engine: the losses and the 11 + 8 values at 1e-3, every slice within the band of FULL_BANDS, norms within 3e-3; bench mode (bf16 /
    fp16 MFMA operands, fused kernels, captured): losses 2e-3, values 2e-2, and the DIRECTION of each gradient bucket (cosine over the
    re-weighted slices) -- see FULL_BANDS for what is and is not asserted per tensor."""
    from tests.golden.configs import grad_slice_index
    c, opt, batch, banks, p, eng = make_engine(name, precision="fp32" if mode == "fp32" else "bf16", use_graph=mode == "bench")
    g = load_golden(name)
    anchors = g["anchors"][0]
    eng.set_banks(*(banks[k] for k in "CFTAV"))
    eng.set_anchors(1, anchors[0]); eng.set_anchors(2, anchors[1])
    band = FULL_BANDS[name][0 if mode == "fp32" else 1]
    rec, cosines = {}, {}
    for stage, key, bucket in ((1, "s1", eng.crit), (2, "s2", eng.main)):
        eng.stage_grads(stage)             # (the reference's stage-2 pass sees the critics AFTER their update: Solver.py:211-214, 221)
        torch.cuda.synchronize()
        s = eng.read_scalars()
        if stage == 1:
            assert_close(s[_lib.S1_LOSS], g["traj_s1_loss"][0], 1e-3 if mode == "fp32" else 2e-3, 2e-5, "stage-1 loss vs reference")
            assert_close(s[_lib.S1_MIS:_lib.S1_MIS + 11], g["traj_s1_mis"][0], 1e-3 if mode == "fp32" else 2e-2, 5e-5 if mode == "fp32" else 2e-3, "stage-1 MI / CMI vs reference")
        else:
            assert_close(s[_lib.S2_LOSS], g["traj_s2_loss"][0], 1e-3 if mode == "fp32" else 2e-3, 2e-5, "stage-2 loss vs reference")
            assert_close(s[_lib.S2_TASK], g["traj_s2_task"][0], 1e-3 if mode == "fp32" else 2e-3, 2e-5, "task loss vs reference")
            assert_close(s[_lib.S2_MIS:_lib.S2_MIS + 8], g["traj_s2_mis"][0], 1e-3 if mode == "fp32" else 2e-2, 5e-5 if mode == "fp32" else 6e-3, "stage-2 MI terms vs reference")
        names = [str(x) for x in g[key + "_gnorm_names"]]
        grads = {n: eng.grads[n].double().cpu().numpy() for n in names}
        top = max(float(x) for x in g[key + "_gnorm"] / np.sqrt([max(grads[n].size, 1) for n in names])) + 1e-30   # largest RMS gradient
        worst = ("", 0.0)
        cat_got, cat_want = [], []
        for i, n in enumerate(names):
            got = grads[n].reshape(-1)
            want = g[key + "_grad:" + n].reshape(-1) if key + "_grad:" + n in g.files else g[key + "_gslice:" + n]
            idx = np.arange(got.size) if want.size == got.size else grad_slice_index(got.size)
            scale = max(float(np.abs(want).max()), float(g[key + "_gnorm"][i]) / np.sqrt(got.size), 1e-3 * top)
            err = float(np.abs(got[idx] - want).max()) / scale
            nerr = abs(float(np.linalg.norm(got)) - float(g[key + "_gnorm"][i])) / max(float(g[key + "_gnorm"][i]), 1e-3 * top * np.sqrt(got.size))
            rec[n] = {"slice_max_rel_scale": err, "norm_rel": nerr}
            # (slices weighted back to their tensors: sqrt(numel / slice) -- the cosine below estimates the whole bucket's direction)
            wgt = np.sqrt(got.size / max(len(idx), 1))
            cat_got.append(got[idx] * wgt); cat_want.append(want * wgt)
            if err > worst[1]:
                worst = (n, err)
            if stage == 1 and i == len(names) - 1:
                pass
            # (the concat critic's score head sums 65 536 pair rows of an InfoNCE whose gradient sums to zero: its bias is pure cancellation)
            if n.endswith("MLP_f.6.bias"):
                continue
        cg, cw = np.concatenate(cat_got), np.concatenate(cat_want)
        cosines[key] = float(cg @ cw / (np.linalg.norm(cg) * np.linalg.norm(cw) + 1e-300))
        if stage == 1:
            eng.stage_apply(1)
    top5 = sorted(((v["slice_max_rel_scale"], n) for n, v in rec.items() if not n.endswith("MLP_f.6.bias")), reverse=True)[:5]
    topn = sorted(((v["norm_rel"], n) for n, v in rec.items() if not n.endswith("MLP_f.6.bias")), reverse=True)[:3]
    _record_errors(f"full_size_gradients/{name}/{mode}", {"worst_slices": top5, "worst_norms": topn, "band": band, "tensors": len(rec), "bucket_cosine": cosines})
    eng.close()
    if mode == "fp32":
        assert min(cosines.values()) >= 1 - 1e-6, cosines
        for n, v in rec.items():
            if n.endswith("MLP_f.6.bias"):
                continue
            assert v["slice_max_rel_scale"] <= band, (name, mode, n, v)
            assert v["norm_rel"] <= 3e-3, (name, mode, n, "norm", v)
    else:
        assert cosines["s1"] >= band[0] and cosines["s2"] >= band[1], (name, cosines, band)


@pytest.mark.parametrize("name", ["cfg3_full", "cfg5_full"])
def test_full_size_timed_mode_every_tensor(name, monkeypatch):
    """at FULL size no gradient tensor of the timed (bf16, fused, captured) mode is left unasserted.  At the
    initial point InfoNCE is ~0 and most tensors are differences of nearly cancelling terms (test_full_size_gradients_vs_reference keeps the
    bucket cosines there), so the per-tensor comparison runs at a WELL-CONDITIONED point: the critics after 40 Adam steps of the fp32 engine
    (Solver.py:205-214 forty times on the fixture batch), where the MI / CMI terms are no longer at their trivial value.  There, for BOTH
    stages and EVERY tensor of both buckets,

        || g_bench - g_fp32 ||  <=  max(3e-2, 8 x jitter) x max(|| g_fp32 ||, 5e-2 x the bucket's largest RMS x sqrt(numel))

    against the fp32 engine -- which IS pinned to the reference per tensor at this size (the `fp32` leg of the test above: <= 4.9e-3 of every
    tensor's scale) -- where `jitter` is the same distance between two equally valid roundings of the bench mode itself (forward operands
    rounded to bf16 instead of fp16, fp32-stored BPTT outputs, the critics as unfused GEMM chains: MIMRL_FWD_BF16 + MIMRL_DG_FP32 +
    MIMRL_NO_FUSED_CONCAT / _MLP / _MI; tensors of near-trivial estimators, whose gradient is cancellation noise in any precision, are judged
    at 5 % of the bucket's largest RMS): a tensor may be as far from fp32 as
    re-rounding moves it (x 8: the two rounded runs share most of their rounding points, the fp32 run has none -- measured ratio 5-6 on the
    near-trivial critics' biases), a wrong kernel confined to one tensor is not.  Worst tensors -> profiles/r06_step_errors.json."""
    g = load_golden(name)
    anchors = g["anchors"][0]

    def grads_at(params, precision, graph, env=()):
        for k, v in env:
            monkeypatch.setenv(k, v)
        c, opt, batch, banks, p, eng = make_engine(name, precision=precision, use_graph=graph)
        for k, _ in env:
            monkeypatch.delenv(k, raising=False)
        eng.set_banks(*(banks[k] for k in "CFTAV"))
        eng.set_anchors(1, anchors[0]); eng.set_anchors(2, anchors[1])
        if params is None:               # the well-conditioned point: ten critic updates on the fixture batch (fp32 engine)
            for _ in range(40):
                eng.stage1_step()
            torch.cuda.synchronize()
            params = {n: v.detach().clone() for n, v in eng.params.items()}
        else:
            eng.load_params(params)
        out, scal = {}, {}
        for stage in (1, 2):
            eng.stage_grads(stage)
            torch.cuda.synchronize()
            names = [n for n in eng.grads if R.is_critic_param(n) == (stage == 1)]
            out[stage] = {n: eng.grads[n].double().cpu().numpy().copy() for n in names}
            sc = eng.read_scalars()
            scal[stage] = float(sc[_lib.S1_LOSS if stage == 1 else _lib.S2_LOSS])
        mis = eng.read_scalars()[_lib.S1_MIS:_lib.S1_MIS + 5].copy()
        eng.close()
        return params, out, scal, mis

    params, g32, l32, mis32 = grads_at(None, "fp32", False)
    assert np.abs(mis32).max() > 1e-3, f"the critics did not leave the trivial point: MI values {mis32}"
    _, gb, lb, _ = grads_at(params, "bf16", True)
    _, gj, lj, _ = grads_at(params, "bf16", True, env=(("MIMRL_FWD_BF16", "1"), ("MIMRL_DG_FP32", "1"), ("MIMRL_NO_FUSED_CONCAT", "1"),
                                                      ("MIMRL_NO_FUSED_MLP", "1"), ("MIMRL_NO_FUSED_MI", "1")))
    assert_close(lb[1], l32[1], 2e-2, 2e-3, "stage-1 loss, bench vs fp32 engine")
    assert_close(lb[2], l32[2], 2e-2, 2e-3, "stage-2 loss, bench vs fp32 engine")
    rec, bad = [], []
    for stage in (1, 2):
        top = max(np.linalg.norm(v) / np.sqrt(v.size) for v in g32[stage].values()) + 1e-30
        for n, want in g32[stage].items():
            floor = 5e-2 * top * np.sqrt(want.size)
            den = max(np.linalg.norm(want), floor)
            err = float(np.linalg.norm(gb[stage][n] - want) / den)
            jit = float(np.linalg.norm(gj[stage][n] - gb[stage][n]) / den)
            band = max(3e-2, 8.0 * jit)
            rec.append((err / band, err, jit, band, f"s{stage}:{n}"))
            if not (np.isfinite(gb[stage][n]).all() and err <= band):
                bad.append(f"s{stage}:{n}: rel L2 {err:.3e} > band {band:.3e} (jitter {jit:.3e})")
    rec.sort(reverse=True)
    _record_errors(f"full_size_timed_mode_every_tensor/{name}", {
        "tensors": len(rec), "over_3e-2": sum(r[1] > 3e-2 for r in rec), "worst_by_band_fraction": [dict(tensor=r[4], rel_l2=r[1], jitter=r[2], band=r[3]) for r in rec[:8]],
        "worst_by_error": [dict(tensor=r[4], rel_l2=r[1], jitter=r[2], band=r[3]) for r in sorted(rec, key=lambda r: -r[1])[:8]],
        "mi_values_at_point": [float(x) for x in mis32], "losses": {"fp32": l32, "bench": lb, "bench_rerounded": lj}})
    assert not bad, f"{len(bad)}/{len(rec)} tensors of the timed mode outside their band:\n" + "\n".join(bad[:12])


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_mfma_lstm_matches_the_scalar_kernels(precision, monkeypatch):
    """--encoders lstm (Model.py:250-252,441-447) runs on the matrix cores since round 5 (lstm.hip: persistent workgroups of
    4 batch rows, W_hh of all four gates as register-resident MFMA B fragments; fp32 mode = v_mfma_f32_16x16x4_f32, exact fp32 fma chains).
    The round-1 scalar kernels stay reachable (MIMRL_LSTM_SCALAR=1): at cfg1's shape with ragged lengths both must give the same losses,
    MI / CMI values, predictions and main-model gradients -- fp32: summation order only (the products are exact fp32 fma chains in both;
    5e-4 of each tensor's scale through 50 cell steps and the ill-conditioned CubeMLP backward); bf16 mode: the scalar kernels are fp32, so
    this is the bf16-operand error of the recurrence."""
    res = {}
    monkeypatch.setenv("MIMRL_LSTM_MFMA_FP32", "1")      # (the fp32 mode defaults to the scalar kernels: the fp32 MFMA form measured slower)
    for tag, env in (("mfma", None), ("scalar", "1")):
        if env:
            monkeypatch.setenv("MIMRL_LSTM_SCALAR", env)
        else:
            monkeypatch.delenv("MIMRL_LSTM_SCALAR", raising=False)
        c, opt, batch, banks, p, eng = make_engine("cfg1_lstm", precision=precision)
        g = load_golden("cfg1_lstm")
        eng.set_banks(*(banks[k] for k in "CFTAV"))
        eng.set_anchors(1, g["anchors"][0][0]); eng.set_anchors(2, g["anchors"][0][1])
        eng.stage_grads(1)
        eng.stage_grads(2)
        torch.cuda.synchronize()
        res[tag] = (eng.read_scalars().copy(), eng.pred.cpu().numpy().copy(),
                    {n: v.double().cpu().numpy().copy() for n, v in eng.grads.items() if not R.is_critic_param(n)})
        eng.close()
    tol = 1e-5 if precision == "fp32" else 2e-2
    (sa, pa, ga), (sb, pb, gb) = res["mfma"], res["scalar"]
    assert_close(sa, sb, tol * 10, tol * 10, "scalars")
    assert_close(pa, pb, tol * 10, tol * 10, "predictions")
    top = max(np.abs(v).max() for v in gb.values())
    worst = ("", 0.0)
    for n in gb:
        scale = max(np.abs(gb[n]).max(), 1e-3 * top)
        e = np.abs(ga[n] - gb[n]).max() / scale
        if e > worst[1]:
            worst = (n, float(e))
    va = np.concatenate([ga[n].reshape(-1) for n in gb]); vb = np.concatenate([gb[n].reshape(-1) for n in gb])
    cos = float(va @ vb / (np.linalg.norm(va) * np.linalg.norm(vb)))
    _record_errors(f"mfma_lstm_vs_scalar/{precision}", {"worst_gradient": worst, "main_bucket_cosine": cos})
    if precision == "fp32":
        assert worst[1] <= 5e-4, worst
    else:   # bf16 operands in the recurrence against fp32 ones: the direction of the main gradient (per tensor the ill-conditioned CubeMLP
        assert cos >= 0.995, (cos, worst)   # LayerNorm parameters move by up to 0.18 of their scale, as under the bf16 GRU: DESIGN.md section 2)
