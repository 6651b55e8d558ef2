"""Pins the oracle's epoch loop (oracle.train_epoch / evaluate_epoch) to fixtures produced by the REAL reference's
Solver.train / Solver.evaluate (Solver.py:194-270; tests/golden/make_golden.py::gen_epoch).  CPU-only."""
import numpy as np
import pytest
import torch

from oracle import mimrl_ref as R
from tests.epoch_helpers import DrawReplay, bands, epoch_case, lr_scale
from tests.helpers import oracle_params, rel_close


@pytest.mark.parametrize("name", ["epoch_tiny", "epoch_tail"])
def test_epoch_loop_matches_reference_solver(name):
    c, opt, sets, g = epoch_case(name)
    p = oracle_params(opt, c["seed"])
    crit = [n for n in p if R.is_critic_param(n)]
    main = [n for n in p if not R.is_critic_param(n)]
    adam_v, adam_m = R.AdamState(p, crit), R.AdamState(p, main)
    draw = DrawReplay(g)
    banks = None
    for ep in range(c["epochs"]):
        rt, at, aat = bands(ep)
        r = R.train_epoch(p, opt, ep, adam_v, adam_m, sets["train"], banks, draw, lr_scale=lr_scale(c, ep))
        banks = r["banks"]
        assert rel_close(r["loss"], g[f"ep{ep}_train_loss"], rt, at), (ep, r["loss"], g[f"ep{ep}_train_loss"])
        assert rel_close(r["loss_mi"], g[f"ep{ep}_train_loss_mi"], rt, at), (ep, r["loss_mi"], g[f"ep{ep}_train_loss_mi"])
        assert rel_close(r["mis"], g[f"ep{ep}_train_mis"], rt, at), (ep, r["mis"], g[f"ep{ep}_train_mis"])
        mae = (r["pred"] - r["target"]).abs().mean().item()
        assert rel_close(mae, g[f"ep{ep}_train_mae"], rt, at)
        for k in "CFTAV":                                   # bank hand-over (Solver.py:223-227,244): features of THIS pass
            assert banks[k].shape == g[f"ep{ep}_bank_{k}"].shape
            np.testing.assert_allclose(banks[k].numpy(), g[f"ep{ep}_bank_{k}"], rtol=rt, atol=aat, err_msg=f"ep{ep} bank {k}")
        for tag in ("valid", "test"):
            e = R.evaluate_epoch(p, opt, sets[tag], banks, draw)
            assert rel_close(e["loss"], g[f"ep{ep}_{tag}_loss"], rt, at), (ep, tag, e["loss"], g[f"ep{ep}_{tag}_loss"])
            assert rel_close(e["mis"], g[f"ep{ep}_{tag}_mis"], rt, at), (ep, tag)
            np.testing.assert_allclose(e["pred"].numpy(), g[f"ep{ep}_{tag}_pred"], rtol=rt, atol=aat)
        np.testing.assert_allclose(g[f"ep{ep}_lr_next"], [c["lr"] * lr_scale(c, ep + 1)] * 2, rtol=1e-12)
    assert draw.done(), "the reference drew more anchors than the oracle's loop consumed"
    names = [str(n) for n in g["final_names"]]
    ps = np.array([p[n].double().sum().item() for n in names])
    np.testing.assert_allclose(ps, g["final_psum"], rtol=1e-3, atol=0.05)
