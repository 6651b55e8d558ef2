"""Pins oracle/ (our CPU restatement) to outputs captured from the REAL reference (tests/golden/*.npz,
made by tests/golden/make_golden.py).  CPU-only."""
import numpy as np
import pytest
import torch

from oracle import mimrl_ref as R
from tests.helpers import case, load_golden, oracle_params, rel_close

TINY = ["tiny_sep", "tiny_cat", "tiny_ragged", "tiny_alt", "tiny_conv", "tiny_mine", "tiny_odd", "tiny_interp", "tiny_lstm", "tiny_tuba_un", "tiny_interp_ga",
        "tiny_sum", "tiny_disc"]
ALL = TINY + ["cfg1_sep", "cfg1_cat", "cfg1_disc", "cfg1_ragged", "cfg1_lstm", "cfg2_sep", "cfg2_ragged", "cfg3_small", "cfg5_small"]


@pytest.mark.parametrize("name", ALL)
def test_forward_matches_reference(name):
    c, opt, batch, banks = case(name)
    g = load_golden(name)
    p = oracle_params(opt, c["seed"])
    with torch.no_grad():
        pred, F_F, T_F, A_F, V_F = R.model_forward(p, opt, *batch[:3])
    for k, val in zip(["pred", "F_F", "T_F", "A_F", "V_F"], [pred, F_F, T_F, A_F, V_F]):
        np.testing.assert_allclose(val.numpy(), g["fwd_" + k], rtol=2e-4, atol=2e-5, err_msg=k)


@pytest.mark.parametrize("name", ALL)
def test_epoch0_rule(name):
    c, opt, batch, banks = case(name)
    g = load_golden(name)
    p = oracle_params(opt, c["seed"])
    with torch.no_grad():
        loss, mis, *_ = R.stage_loss(p, opt, 2, batch, None, None)
        l1, *_ = R.stage_loss(p, opt, 1, batch, None, None)
    assert rel_close(loss, g["e0_stage2_loss"], 1e-4)
    assert float(l1) == float(g["e0_stage1_loss"]) == 0.0
    assert np.all(g["e0_stage2_mis"] == 0) and all(float(m) == 0 for m in mis)


@pytest.mark.parametrize("name", ALL)
def test_two_stage_trajectory_matches_reference(name):
    c, opt, batch, banks = case(name)
    g = load_golden(name)
    p = oracle_params(opt, c["seed"])
    crit = [n for n in p if R.is_critic_param(n)]
    main = [n for n in p if not R.is_critic_param(n)]
    adam_v, adam_m = R.AdamState(p, crit), R.AdamState(p, main)
    anchors = g["anchors"]
    # beyond 3 alternating updates on one batch the InfoNCE critics are chaotic (values swing by O(1)
    # under fp32 summation-order noise), so only the first three iterations are asserted.
    steps = min(anchors.shape[0], 3)
    for it in range(steps):
        r1, r2 = R.two_stage_step(p, opt, adam_v, adam_m, batch, banks, anchors[it, 0], anchors[it, 1])
        # it == 0: tight (1e-3 rel, atol ~1e-5 because InfoNCE ~ 0 at init).
        # it >= 1: Adam's early steps are ~lr*sign(g); elements whose grad is ~0 flip sign on fp32 summation
        # noise (each flip moves a weight by 2*lr = 8e-3), so later points carry a looser, documented band.
        rt, at = (1e-3, 2e-5) if it == 0 else ((3e-2, 2e-3) if it < 3 else (0.25, 0.05))   # chaotic growth
        assert rel_close(r1["loss"], g["traj_s1_loss"][it], rt, at), (it, r1["loss"], g["traj_s1_loss"][it])
        assert rel_close([float(m) for m in r1["mis"]], g["traj_s1_mis"][it], rt, at), it
        assert rel_close(r2["loss"], g["traj_s2_loss"][it], rt, at), (it, r2["loss"], g["traj_s2_loss"][it])
        assert rel_close(r2["task"], g["traj_s2_task"][it], rt, at)
        assert rel_close([float(m) for m in r2["mis"]], g["traj_s2_mis"][it], rt, 5e-5 if it == 0 else (5e-3 if it < 3 else 0.05)), \
            (it, [float(m) for m in r2["mis"]], g["traj_s2_mis"][it])
        if it == 0:
            names1 = [str(s) for s in g["s1_gnorm_names"]]
            gn = np.array([r1["grads"][n].norm().item() if n in r1["grads"] else 0.0 for n in names1])
            # golden norms are pre-clip; recompute pre-clip norms is not possible post-clip -> compare the
            # small full tensors (clip 1.5 never binds on them) and the post-step parameter checksums.
            for key in g.files:
                if key.startswith("s1_grad:"):
                    n = key.split(":", 1)[1]
                    np.testing.assert_allclose(r1["grads"][n].numpy(), np.clip(g[key], -1.5, 1.5),
                                               rtol=1e-1 if name == "tiny_interp_ga" else 1e-2, atol=1e-3 if name == "tiny_interp_ga" else 1e-5, err_msg=n)
            assert gn.shape == g["s1_gnorm"].shape
            ps = np.array([p[n].double().sum().item() for n in names1])
            # Adam t=1 moves every element by ~lr*sign(g): allow a handful of sign flips of ~0 grads (2*lr each)
            np.testing.assert_allclose(ps, g["s1_psum_after"], rtol=1e-4, atol=0.05)
            names2 = [str(s) for s in g["s2_gnorm_names"]]
            for key in g.files:
                if key.startswith("s2_grad:"):
                    n = key.split(":", 1)[1]
                    np.testing.assert_allclose(r2["grads"][n].numpy(), np.clip(g[key], -1.5, 1.5),
                                               rtol=1e-1 if name == "tiny_interp_ga" else 1e-2, atol=1e-3 if name == "tiny_interp_ga" else 1e-5, err_msg=n)
            if name == "tiny_interp_ga":
                # with log a(y) ~ -120 the reference's fp32 bound is 0.6 % off its own float64 value: stage-2 gradients carry
                # percent-level fp32 noise, Adam's first step (lr * sign g) flips on it, and nothing after it is comparable
                break
            ps = np.array([p[n].double().sum().item() for n in names2])
            np.testing.assert_allclose(ps, g["s2_psum_after"], rtol=1e-4, atol=0.05)
            psq = np.array([(p[n].double() ** 2).sum().item() for n in names2])
            np.testing.assert_allclose(psq, g["s2_psq_after"], rtol=1e-4, atol=5e-3)


def test_unit_estimators_match_reference():
    g = load_golden("units")
    from types import SimpleNamespace
    from mimrl_amd import synth
    x, y = torch.from_numpy(g["x"]), torch.from_numpy(g["y"])
    for critic in ("separate", "concat"):
        opt = SimpleNamespace(critic_type=critic, d_common=128, d_hiddens=[], d_outs=[], time_len=1, bias=True,
                              ln_first=False, res_project=[])
        from mimrl_amd import layout
        shapes = [(n, s) for n, s in layout.named_shapes(opt, 768, 74, 35) if n.startswith("vmi_estimator_f_t.")]
        p = {n: torch.from_numpy(synth.portable_tensor(n, s, 3)) for n, s in shapes}
        s = R.critic_scores(p, "f_t", critic, x, y)
        np.testing.assert_allclose(s.numpy(), g[f"{critic}_scores"], rtol=1e-4, atol=1e-5)
        for bound in ("infonce", "nwj", "tuba", "dv", "js", "js_fgan", "smile", "interpolate"):
            xt, yt = x.clone().requires_grad_(True), y.clone().requires_grad_(True)
            mi = R.BOUNDS[bound](R.critic_scores(p, "f_t", critic, xt, yt))
            (-mi).backward()
            assert rel_close(mi.item(), g[f"{critic}_{bound}_mi"], 1e-4, 1e-5), (critic, bound)
            np.testing.assert_allclose(xt.grad.numpy(), g[f"{critic}_{bound}_dx"], rtol=2e-3, atol=1e-6)
            np.testing.assert_allclose(yt.grad.numpy(), g[f"{critic}_{bound}_dy"], rtol=2e-3, atol=1e-6)


def test_unit_cmi_matches_reference():
    g = load_golden("units")
    from types import SimpleNamespace
    from mimrl_amd import layout, synth
    x, y, c = (torch.from_numpy(g[k]) for k in ("x", "y", "c"))
    banks = {k: torch.from_numpy(v) for k, v in synth.synthetic_banks(60, seed=5).items()}
    base = SimpleNamespace(critic_type="separate", d_common=128, d_hiddens=[], d_outs=[], time_len=1, bias=True,
                           ln_first=False, res_project=[], k_neighbor=2)
    shapes = [(n, s) for n, s in layout.named_shapes(base, 768, 74, 35) if n.startswith("vcmi_estimator_ta_c.")]
    p = {n: torch.from_numpy(synth.portable_tensor(n, s, 3)) for n, s in shapes}
    for last in ("sigmoid", "hardtanh"):
        base.cmi_last_acticate = last
        anchors = g[f"cmi_{last}_anchors"]
        kx, ky, kz, _ = R.prod_knn_sample(banks["T"], banks["A"], banks["C"], anchors, 2)
        np.testing.assert_array_equal(kx.numpy(), g[f"cmi_{last}_kx"])
        np.testing.assert_array_equal(kz.numpy()[:, 0], g[f"cmi_{last}_kz_col0"])
        xt, yt = x.clone().requires_grad_(True), y.clone().requires_grad_(True)
        cmi, bce = R.vcmi_estimate(p, "ta_c", base, xt, yt, c, kx, ky, kz)
        (cmi + bce).backward()
        assert rel_close(cmi.item(), g[f"cmi_{last}_cmi"], 1e-4, 1e-6)
        assert rel_close(bce.item(), g[f"cmi_{last}_bce"], 1e-4, 1e-6)
        np.testing.assert_allclose(xt.grad.numpy(), g[f"cmi_{last}_dx"], rtol=2e-3, atol=1e-7)
        np.testing.assert_allclose(yt.grad.numpy(), g[f"cmi_{last}_dy"], rtol=2e-3, atol=1e-7)


@pytest.mark.parametrize("name", ["cfg3_full", "cfg5_full"])
def test_full_size_fixtures_are_reference_outputs(name):
    """The full-size fixtures (round 5: one reference step at B = 256 / T = 500 / concat / N = 16326 and at B = 32 / T = 1000, generated by
    tests/golden/make_golden.py from the REAL reference) hold what the GPU tests read -- 11 + 8 values, both losses, per-tensor gradient
    norms / sums and a 512-entry slice of every tensor.  The oracle's autograd at this size takes minutes, so the CPU tier pins the oracle to
    them through the part that takes seconds: the cfg5_full forward pass (2000 serial cell steps, T = 1000); cfg3_full's forward pass is
    pinned on the GPU box (tests/test_gpu_step.py::test_cfg3_full_size_properties uses the same inputs)."""
    from tests.golden.configs import grad_slice_index
    g = load_golden(name)
    c, opt, batch, banks = case(name)
    p = oracle_params(opt, c["seed"])
    for key, crit in (("s1", True), ("s2", False)):
        names = [str(x) for x in g[key + "_gnorm_names"]]
        assert all(R.is_critic_param(n) == crit for n in names) and len(names) > 60
        assert np.all(np.isfinite(g[key + "_gnorm"])) and np.all(g[key + "_gnorm"] >= 0)
        for n in names:
            numel = int(p[n].numel())
            k = key + ("_grad:" if numel <= 512 else "_gslice:") + n
            assert k in g.files and g[k].size == min(numel, 512), k
            assert grad_slice_index(numel).size == min(numel, 512)
    assert g["traj_s1_mis"].shape == (1, 11) and g["traj_s2_mis"].shape == (1, 8) and g["anchors"].shape[:3] == (1, 2, 6)
    if name == "cfg5_full":
        with torch.no_grad():
            pred, F_F, T_F, A_F, V_F = R.model_forward(p, opt, *batch[:3])
        for k, val in zip(["pred", "F_F", "T_F", "A_F", "V_F"], [pred, F_F, T_F, A_F, V_F]):
            np.testing.assert_allclose(val.numpy(), g["fwd_" + k], rtol=2e-4, atol=2e-5, err_msg=k)
