"""-m gpu: the reference-shaped Python surface (Solver / Model / Customization / Main) on the device -- the boundary a
user of the reference actually calls -- against fixtures produced by the REAL reference's own
Solver.train / Solver.evaluate (tests/golden/epoch_*.npz) and against the oracle."""
import copy
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from mimrl_amd import _lib, synth
from mimrl_amd.Customization import compute_custumized_loss, compute_outputs_from_model, other_model_operations
from mimrl_amd.Model import Model
from mimrl_amd.Solver import Solver
from oracle import mimrl_ref as R
from tests.epoch_helpers import DrawReplay, bands, epoch_case
from tests.gpu_helpers import assert_close
from tests.helpers import case, load_golden, oracle_params

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def as_datas(batch):
    """(t, a, v, y) -> the reference's 'Dec' 11-tuple (Customization.py:46) with BERT features in slot 6."""
    t, a, v, y = batch
    ones = torch.ones(a.shape[0], a.shape[1], dtype=torch.long)
    return (None, a, v, None, None, y.reshape(-1, 1), t, torch.zeros_like(ones), ones, None, None)


def solver_opt(opt, **kw):
    o = copy.copy(opt)
    o.task_name, o.seed, o.epochs_num, o.save_best_features = kw.pop("task_name", "pytest"), 0, 1, False
    o.precision, o.no_graph, o.host_anchors = kw.pop("precision", "fp32"), kw.pop("no_graph", True), kw.pop("host_anchors", True)
    for k, v in kw.items():
        setattr(o, k, v)
    return o


@pytest.mark.parametrize("name", ["epoch_tiny", "epoch_tail"])
@pytest.mark.parametrize("use_graph", [False, True])
def test_solver_train_evaluate_match_reference_solver(name, use_graph, monkeypatch):
    """mimrl_amd.Solver.train / evaluate over several epochs == the reference's Solver.train / evaluate
    (Solver.py:194-270): stage1_n critic passes, epoch-0 skip, bank hand-over, eval-mode stage-2 loss with banks,
    multi-step lr schedule, and (epoch_tail) a partial last batch under drop_last=False (Parameters.py:21)."""
    c, opt, sets, g = epoch_case(name)
    loaders = tuple([as_datas(b) for b in sets[k]] for k in ("train", "valid", "test")) + (768, 74, 35)
    sol = Solver(solver_opt(opt, no_graph=not use_graph), loaders)
    sol.model.load_state_dict(oracle_params(opt, c["seed"]))
    draw = DrawReplay(g)
    monkeypatch.setattr(synth, "draw_anchors", lambda N, m, calls=6: draw(N, m))   # the reference's recorded np.random.choice draws
    banks = ([], [], [], [], [])
    for ep in range(c["epochs"]):
        rt, at, aat = bands(ep)
        r = sol.train(ep, sol.train_loader, *banks)
        banks = r[4:]
        assert_close(r[0], g[f"ep{ep}_train_loss"], rt, at, f"ep{ep} train loss")
        assert_close(r[1], g[f"ep{ep}_train_loss_mi"], rt, at, f"ep{ep} stage-1 loss")
        assert_close(r[2], g[f"ep{ep}_train_mis"], rt, at, f"ep{ep} MI means")
        assert_close(r[3]["mae"], g[f"ep{ep}_train_mae"], rt, at, f"ep{ep} train mae")
        for k, bk in zip("CFTAV", banks):
            assert tuple(bk.shape) == g[f"ep{ep}_bank_{k}"].shape, (k, bk.shape)
            assert_close(bk.cpu().numpy(), g[f"ep{ep}_bank_{k}"], rt, aat, f"ep{ep} bank {k}")
        for tag, ld in (("valid", sol.valid_loader), ("test", sol.test_loader)):
            e = sol.evaluate(ld, *banks)
            assert_close(e[0], g[f"ep{ep}_{tag}_loss"], rt, at, f"ep{ep} {tag} loss")
            assert_close(e[1], g[f"ep{ep}_{tag}_mis"], rt, at, f"ep{ep} {tag} MI means")
            assert_close(e[3].reshape(-1), g[f"ep{ep}_{tag}_pred"], rt, aat, f"ep{ep} {tag} predictions")
            assert_close(e[2]["mae"], g[f"ep{ep}_{tag}_mae"], rt, at, f"ep{ep} {tag} mae")
    assert draw.done(), "the reference drew more anchors than Solver.train/evaluate consumed"
    names = [str(n) for n in g["final_names"]]
    ps = np.array([sol.model.state_dict()[n].double().sum().item() for n in names])
    np.testing.assert_allclose(ps, g["final_psum"], rtol=1e-3, atol=0.05)
    if name == "epoch_tail":
        assert list(sol._tails) == [4], "the partial last batch (36 = 4*8 + 4) runs on a second handle of batch 4"


@pytest.mark.parametrize("name,precision", [("tiny_sep", "fp32"), ("cfg1_sep", "bf16"), ("cfg1_cat", "bf16")])
def test_pipelined_critic_pass_equals_the_sequential_one(name, precision, monkeypatch):
    """Round 6: in the reference's epoch schedule (Solver.py:200-216) the critic passes run over the whole loader with the
    main model frozen, so `Solver.train` CAN issue Model.forward of batch i + 1 beside the critic update on batch i (`mimrl_stage1_pipe`,
    HipEngine.stage1_pass) when the batches are device-resident.  Same batches, same order, same dropout keys and device-drawn anchors: the
    pass must give the stage-1 losses, the critic parameters and -- through the model pass that follows -- every returned value of the
    sequential pass (the default: the look-ahead pass is opt-in, MIMRL_EPOCH_PIPE=1 -- it measured slower), up to the order of float
    atomics."""
    c, opt, batch, banks = case(name)
    B, T = c["B"], c["T"]
    o = solver_opt(opt, host_anchors=False, no_graph=False, precision=precision, stage1_n=2)
    o.dropout = [0.1, 0.1, 0.1, 0.1]
    train = [as_datas(tuple(torch.as_tensor(x).cuda() for x in synth.synthetic_batch(B, T, seed=40 + i))) for i in range(5)]
    train = [tuple(x.cuda() if torch.is_tensor(x) else x for x in d) for d in train]
    res = {}
    for tag in ("seq", "seq2", "pipe"):
        if tag != "pipe":
            monkeypatch.delenv("MIMRL_EPOCH_PIPE", raising=False)
        else:
            monkeypatch.setenv("MIMRL_EPOCH_PIPE", "1")
        sol = Solver(o, (train, train[:1], train[:1], 768, 74, 35))
        sol.model.load_state_dict(oracle_params(opt, c["seed"]))
        called = {"n": 0}
        orig = sol.engine.stage1_pass
        def spy(*a, **k):
            called["n"] += 1
            return orig(*a, **k)
        sol.engine.stage1_pass = spy
        nb = min(c["N"], 5 * B)                       # (the banks of a training run hold one row per training sample)
        r = sol.train(1, sol.train_loader, *(torch.as_tensor(np.asarray(banks[k]))[:nb] for k in "CFTAV"))
        torch.cuda.synchronize()
        assert called["n"] == (2 if tag == "pipe" else 0), (tag, called)
        res[tag] = (r[0], r[1], np.asarray(r[2]), {n: v.double().cpu().numpy() for n, v in sol.model.state_dict().items()})
        sol.engine.close()
    a, b, b2 = res["pipe"], res["seq"], res["seq2"]
    # the yardstick is the sequential pass AGAINST ITSELF (a second run): 10 critic + 5 model Adam steps at lr 4e-3 amplify the order of the
    # float atomics (every entry whose gradient is ~0 moves by ~lr per step in a direction the noise picks) -- in bf16 mode two sequential
    # runs already drift apart by ~1 lr per entry on average
    def dist(x, y):
        return (abs(x[1] - y[1]), abs(x[0] - y[0]), float(np.abs(x[2] - y[2]).max()),
                max(float(np.abs(pa - y[3][n]).mean()) for n, pa in x[3].items()))
    noise, got = dist(b, b2), dist(a, b)
    lr = float(o.learning_rate)
    floor = (3e-4, 3e-4, 2e-3, 0.05 * lr) if precision == "fp32" else (5e-3, 1e-2, 2e-2, 0.5 * lr)   # (one pair of runs is a noisy estimate of the noise)
    for what, g_, n_, f_ in zip(("mean stage-1 loss", "mean stage-2 loss", "MI means", "mean parameter drift"), got, noise, floor):
        assert g_ <= max(3.0 * n_, f_), f"{what}: pipelined vs sequential {g_:.3e}, sequential vs sequential {n_:.3e}"
    for n, pa in a[3].items():
        assert np.isfinite(pa).all() and np.abs(pa - b[3][n]).max() <= 15 * 2 * lr, n


@pytest.mark.parametrize("precision,use_graph", [("fp32", False), ("bf16", True)])
def test_solver_step_equals_engine_step_and_oracle(precision, use_graph, monkeypatch):
    """Solver(opt, loaders).step(datas) x3 == HipEngine.step() x3 (bitwise: same library calls) == the oracle."""
    from mimrl_amd.engine import HipEngine
    c, opt, batch, banks = case("tiny_sep")
    g = load_golden("tiny_sep")
    anchors = g["anchors"]
    datas = as_datas(batch)
    sol = Solver(solver_opt(opt, precision=precision, no_graph=not use_graph), ([datas] * 5, [datas], [datas], 768, 74, 35))
    p = oracle_params(opt, c["seed"])
    sol.model.load_state_dict(p)
    sol.engine.set_banks(*(banks[k] for k in "CFTAV"))
    eng = HipEngine(opt, 768, 74, 35, seq_len=c["T"], bank_capacity=c["N"], precision=precision, use_graph=use_graph)
    eng.load_params(p)
    eng.set_batch(*batch)
    eng.set_banks(*(banks[k] for k in "CFTAV"))
    eng.set_stage2_prefetch(True)
    import itertools
    seq = itertools.chain([anchors[it, st] for it in range(3) for st in (0, 1)], itertools.repeat(anchors[0, 0]))
    monkeypatch.setattr(synth, "draw_anchors", lambda N, m, calls=6: next(seq))
    crit = [n for n in p if R.is_critic_param(n)]
    main = [n for n in p if not R.is_critic_param(n)]
    adam_v, adam_m = R.AdamState(p, crit), R.AdamState(p, main)
    for it in range(3):
        l1, l2, mis, pred = sol.step(datas)
        eng.set_anchors(1, anchors[it, 0]); eng.set_anchors(2, anchors[it, 1])
        eng.step()
        s = eng.read_scalars()
        if precision == "fp32" and it == 0:   # same library calls; reductions / weight gradients use float atomics (last-ulp order noise,
            assert_close(float(l1), s[_lib.S1_LOSS], 2e-6, 1e-7, "Solver.step vs HipEngine.step, stage-1 loss")     # amplified by later updates)
            assert_close(float(l2), s[_lib.S2_LOSS], 2e-6, 1e-7, "Solver.step vs HipEngine.step, stage-2 loss")
            assert_close(pred.reshape(-1).cpu().numpy(), eng.pred.cpu().numpy(), 2e-6, 1e-7, "predictions")
        r1, r2 = R.two_stage_step(p, opt, adam_v, adam_m, batch, banks, anchors[it, 0], anchors[it, 1])
        rt, at = ((1e-3, 2e-5) if it == 0 else (3e-2, 5e-3)) if precision == "fp32" else (3e-2, 2e-2)
        assert_close(float(l1), r1["loss"].item(), rt, at, f"it{it} stage-1 loss vs oracle")
        assert_close(float(l2), r2["loss"].item(), rt, at, f"it{it} stage-2 loss vs oracle")
        # (bf16 after an update: a CMI term is a difference of two log-ratio sums over 8 samples, one logit moving by 2^-8 shows)
        mi_at = 5e-5 if (it == 0 and precision == "fp32") else (3e-2 if precision == "fp32" or it == 0 else 0.15)
        assert_close(mis.cpu().numpy(), [m.item() for m in r2["mis"]], rt, mi_at, f"it{it} MI terms")
    # usability: a lone stage call after step() must not trip over the overlap mode (it switches back to sequential)
    sol.stage1_step(datas)
    sol.stage2_step(datas)
    sol.stage2_step(datas)
    assert np.isfinite(sol.engine.read_scalars()).all()
    eng.close()


@pytest.mark.parametrize("name", ["tiny_sep", "tiny_cat", "tiny_sum"])
def test_model_and_customization_surface(name):
    """Model.forward(return_features=True), compute_vmi_loss_stage1/2, compute_outputs_from_model and
    compute_custumized_loss (incl. the empty-bank rule, Customization.py:97-98,105-106) against the reference goldens."""
    c, opt, batch, banks = case(name)
    g = load_golden(name)
    o = solver_opt(opt)
    np.random.seed(0)
    model = Model(o, 768, 74, 35, bank_capacity=c["N"], init="portable")
    assert other_model_operations(model, o) is None
    model.load_state_dict(oracle_params(opt, c["seed"]))
    assert set(n for n, _ in model.named_parameters()) == set(oracle_params(opt, c["seed"]))
    model.eval()
    out = model(batch[0], None, None, batch[1], batch[2], return_features=True)
    assert len(out) == 5 and tuple(out[0].shape) == (c["B"], 1)
    for val, k in zip(out, ["pred", "F_F", "T_F", "A_F", "V_F"]):
        assert_close(val.cpu().numpy().reshape(g["fwd_" + k].shape), g["fwd_" + k], 1e-3, 2e-5, k)
    assert len(model(batch[0], None, None, batch[1], batch[2])) == 1
    # the glue functions, reference signatures
    model.train()
    datas = as_datas(batch)
    outputs = compute_outputs_from_model(model, datas, o)
    labels = datas[5]
    task = (outputs[0].reshape(-1) - labels.reshape(-1).cuda()).abs().mean()
    empty = ([], [], [], [], [])
    l1, mis1 = compute_custumized_loss(model, task, outputs, labels, None, o, 1, *empty)
    l2, mis2 = compute_custumized_loss(model, task, outputs, labels, None, o, 2, *empty)
    assert float(l1) == 0.0 and float(l2) == float(task) and all(float(m) == 0 for m in mis2) and len(mis2) == 8
    assert_close(float(l2), g["e0_stage2_loss"], 1e-3, 1e-6, "epoch-0 stage-2 loss")
    bank_t = tuple(banks[k] for k in "CFTAV")
    anchors = g["anchors"][0]
    seq = iter([anchors[0], anchors[1]])
    real = synth.draw_anchors
    synth.draw_anchors = lambda N, m, calls=6: next(seq)
    try:
        l1, mis1 = compute_custumized_loss(model, task, outputs, labels, None, o, 1, *bank_t)
        s1 = model.engine.read_scalars().copy()
        mis_a, losses_a = model.compute_vmi_loss_stage2(outputs[0], labels, *outputs[1:], *bank_t)
    finally:
        synth.draw_anchors = real
    assert_close(float(l1), g["traj_s1_loss"][0], 1e-3, 1e-5, "stage-1 loss through compute_custumized_loss")
    assert_close(s1[_lib.S1_MIS:_lib.S1_MIS + 11], g["traj_s1_mis"][0], 1e-3, 2e-5, "11 stage-1 MI / CMI values")
    assert len(mis1) == 11 and len(mis_a) == 8 and len(losses_a) == 8
    # stage 2 here is evaluated BEFORE any critic update (the golden trajectory applies stage 1 first): compare with the oracle
    p = oracle_params(opt, c["seed"])
    with torch.no_grad():
        _, mis_o, *_ = R.stage_loss(p, opt, 2, batch, banks, anchors[1])
    assert_close([float(m) for m in mis_a], [float(m) for m in mis_o], 1e-3, 5e-5, "8 stage-2 MI terms")
    # the feature ARGUMENTS mean what they mean in the reference (Model.py:343): tensors that are not views of the engine's buffers
    # (here: shuffled copies on the host) are what the estimators run on, not the last forward's features
    perm = torch.randperm(c["B"], generator=torch.Generator().manual_seed(0))
    feats_o = [f.detach().cpu()[perm].clone() for f in outputs[1:]]
    pred_o, lab_o = outputs[0].detach().cpu()[perm].clone(), torch.as_tensor(labels).reshape(-1)[perm].clone()
    seq = iter([anchors[1]])
    synth.draw_anchors = lambda N, m, calls=6: next(seq)
    try:
        mis_b, _ = model.compute_vmi_loss_stage2(pred_o, lab_o, *feats_o, *bank_t)
    finally:
        synth.draw_anchors = real
    with torch.no_grad():
        mi_t, _ = R.stage2_terms(p, opt, lab_o, dict(zip("FTAV", feats_o)), banks, anchors[1])
    assert_close([float(m) for m in mis_b], [float(m) for m in mi_t], 1e-3, 5e-5, "8 stage-2 MI terms on caller-owned features")
    # nn.Module habits of the reference's callers: .to(device), torch.save(model) / torch.load
    import io
    assert model.to("cuda") is model and model.to(torch.device("cuda", 0)) is model
    with pytest.raises(_lib.MimrlError):
        model.to("cpu")
    buf = io.BytesIO()
    torch.save(model, buf)
    buf.seek(0)
    m2 = torch.load(buf, weights_only=False)
    for k, v in model.state_dict().items():
        assert torch.equal(v.cpu(), m2.state_dict()[k].cpu()), k
    m2.eval()
    out2 = m2(batch[0], None, None, batch[1], batch[2], return_features=True)
    model.eval()
    out1 = model(batch[0], None, None, batch[1], batch[2], return_features=True)
    assert_close(out2[0].cpu().numpy(), out1[0].cpu().numpy(), 1e-6, 1e-7, "reloaded model's prediction")
    m2.engine.close()
    model.engine.close()


def test_checkpoint_round_trip(tmp_path, monkeypatch):
    """Solver.checkpoint / load_checkpoint: parameters, both Adam moment buckets and the device step counters
    (Solver.py:57-62: 'model', 'optim_main', 'optim_vmi') -- save, continue; reload, continue: identical results."""
    c, opt, batch, banks = case("tiny_sep")
    datas = as_datas(batch)
    loaders = ([datas] * 5, [datas], [datas], 768, 74, 35)
    o = solver_opt(opt, host_anchors=False)
    sol = Solver(o, loaders)
    sol.model.load_state_dict(oracle_params(opt, c["seed"]))
    sol.engine.set_banks(*(banks[k] for k in "CFTAV"))
    for _ in range(3):
        sol.step(datas)
    ck = sol.checkpoint(0)
    path = tmp_path / "ck.pth.tar"
    torch.save(ck, path)
    assert set(ck) >= {"epoch", "model", "optim_main", "optim_vmi"} and int(ck["optim_main"]["step"]) == 3 and int(ck["optim_vmi"]["step"]) == 3
    ref = [tuple(float(x) for x in sol.step(datas)[:2]) for _ in range(2)]
    sol2 = Solver(o, loaders)
    assert sol2.load_checkpoint(str(path)) == 0
    # the banks travel with the checkpoint (the reference would resume with empty banks = the epoch-0 rule for a whole epoch)
    assert sol2.engine.bank_rows == c["N"] and len(sol2.resume_banks) == 5 and sol2.resume_banks[1].shape == (c["N"], 128)
    np.testing.assert_array_equal(sol2.engine.bank["F"][:c["N"]].cpu().numpy(), np.asarray(banks["F"], np.float32))
    got = [tuple(float(x) for x in sol2.step(datas)[:2]) for _ in range(2)]
    np.testing.assert_allclose(got, ref, rtol=1e-5, atol=1e-6)
    # the moments are stored per parameter NAME (raw flat buckets loaded silently onto the wrong parameters once the main bucket
    # was re-ordered); a checkpoint with flat buckets loads only under the layout fingerprint that wrote it
    assert isinstance(ck["optim_main"]["m"], dict) and set(ck["optim_main"]["m"]) | set(ck["optim_vmi"]["m"]) == set(ck["model"])
    assert ck["optim_main"]["m"]["W_t.weight"].shape == ck["model"]["W_t.weight"].shape
    st = sol.engine.optimizer_state()
    flat = dict(ck, optim_main=dict(ck["optim_main"], m=st["main_m"].cpu(), v=st["main_v"].cpu()),
                optim_vmi=dict(ck["optim_vmi"], m=st["crit_m"].cpu(), v=st["crit_v"].cpu()))
    sol3 = Solver(o, loaders)
    assert sol3.load_checkpoint(dict(flat)) == 0                                  # tagged with this build's layout: accepted
    torch.testing.assert_close(sol3.engine.main["m"], sol.engine.main["m"])
    for bad in (dict(flat, layout="0" * 32), {k: v for k, v in flat.items() if k != "layout"}):   # another layout / an untagged (pre-round-6) file
        with pytest.raises(_lib.MimrlError, match="flat bucket"):
            sol3.load_checkpoint(bad)
    # per-name loading really scatters by name: a permuted dict order gives the same buckets
    rev = dict(ck, optim_main=dict(ck["optim_main"], m=dict(reversed(list(ck["optim_main"]["m"].items())))))
    sol3.engine.main["m"].zero_()
    sol3.load_checkpoint(rev)
    for n_, t_ in ck["optim_main"]["m"].items():
        e_ = [x for x in sol3.engine.entries if x[0] == n_][0]
        torch.testing.assert_close(sol3.engine.main["m"][e_[2]:e_[2] + t_.numel()].cpu(), t_.reshape(-1))


def test_main_cli_subprocess_smoke(tmp_path):
    """`python Main.py --dataset synthetic ... --epochs_num 2`: the runner end to end (two epochs incl. stage 1, evaluation,
    checkpoints and prediction files written like Solver.py:513-531), partial last batches included (n=100, B=16)."""
    cmd = [sys.executable, os.path.join(ROOT, "Main.py"), "--dataset", "synthetic", "--synthetic_n", "100", "--batch_size", "16",
           "--time_len", "12", "--d_hiddens", "12-3-128=4-3-128", "--d_outs", "12-3-128=4-3-128", "--bias", "--res_project", "1-1",
           "--dropout", "0.1-0.1-0.1-0.1", "--dropout_mlp", "0.0-0.0-0.0", "--epochs_num", "2", "--stage1_n", "1", "--precision", "bf16",
           "--loss_mi_coefficient1", "-".join(["1.0"] * 11), "--loss_mi_coefficient2", "-".join(["0.01"] * 8), "--task_name", "cli_smoke",
           "--gradient_clip", "1.5", "--learning_rate", "1e-3"]
    r = subprocess.run(cmd, cwd=tmp_path, capture_output=True, text=True, timeout=600, env=dict(os.environ, PYTHONPATH=ROOT))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    task = tmp_path / "TaskRuning" / "cli_smoke"
    for f in ("Running.log", "best_valid_model.pth.tar", "best_test_model.pth.tar", "predictions_val.npy", "predictions_test.npy",
              "targets_val.npy"):
        assert (task / f).exists(), f
    log = (task / "Running.log").read_text()
    assert "Epoch:[  2]" in log and "Training complete." in log and "nan" not in log.lower()
    ck = torch.load(task / "best_valid_model.pth.tar", map_location="cpu")
    assert "W_t.weight" in ck["model"] and "vmi_estimator_f_t.critic_model.MLP_g.0.weight" in ck["model"]


def test_resident_dataset_equals_host_dataset():
    """data.get_data_loader: by default the (small) dataset lives in HBM and a batch is a device-to-device slice copy on the upload
    stream; --host_data keeps it in pinned host memory (H2D per batch).  Same seeds -> the same epoch either way, including the
    one-batch upload lookahead of Solver._iter_loaded and a partial last batch (n=100, B=16)."""
    from mimrl_amd import Parameters
    from mimrl_amd.data import get_data_loader
    argv = ["--dataset", "synthetic", "--synthetic_n", "100", "--batch_size", "16", "--time_len", "12", "--d_hiddens", "12-3-128=4-3-128",
            "--d_outs", "12-3-128=4-3-128", "--bias", "--res_project", "1-1", "--dropout", "0.0-0.0-0.0-0.0", "--dropout_mlp", "0.0-0.0-0.0",
            "--loss_mi_coefficient1", "-".join(["1.0"] * 11), "--loss_mi_coefficient2", "-".join(["0.01"] * 8), "--learning_rate", "1e-4",
            "--task_name", "pytest_resident"]
    out = {}
    for host in (False, True):
        opt = Parameters.parse_args(argv + (["--host_data"] if host else []))
        loaders = get_data_loader(opt)
        assert loaders[0].t.is_cuda == (not host) and (host is False or loaders[0].t.is_pinned())
        sol = Solver(solver_opt(opt, host_anchors=False, no_graph=False), loaders)
        sol.model.load_state_dict(oracle_params(opt, 3))
        banks = ([], [], [], [], [])
        res = []
        for ep in range(2):
            r = sol.train(ep, sol.train_loader, *banks)
            banks = r[4:]
            e = sol.evaluate(sol.valid_loader, *banks)
            res.append((float(r[0]), float(r[1]), np.asarray(r[2], dtype=np.float64), float(e[0]), banks[1].cpu().numpy().copy()))
        out[host] = res
    for (a, b) in zip(out[False], out[True]):
        np.testing.assert_allclose(a[0], b[0], rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(a[1], b[1], rtol=1e-5, atol=1e-7)
        # (two runs of the same arithmetic: float atomics reorder additions, Adam's lr * sign(g) first steps turn that into 2e-4 parameter
        #  differences, and the CMI-derived terms are differences of log-ratio sums -- observed up to 5.6e-5 on a term of -3.4e-3)
        np.testing.assert_allclose(a[2], b[2], rtol=1e-4, atol=2e-4)
        np.testing.assert_allclose(a[3], b[3], rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(a[4], b[4], rtol=1e-4, atol=1e-5)


def test_sibling_handles_stay_coherent_without_help():
    """Two HipEngine handles on ONE set of buckets (``share``: how Solver runs a partial last batch) cache their own bf16 weight
    images; a caller that alternates them -- without Solver._engine_for's manual params_changed -- must still train on current
    weights: the shared parameter-version counter makes every compute call refresh a stale handle.  Same batch through
    A, B, A == three steps of a lone engine (bf16 mode: float atomics reorder additions, Adam's lr * sign(g) steps flip ~0 entries)."""
    from mimrl_amd.engine import HipEngine
    c, opt, batch, banks = case("cfg1_sep")
    opt.learning_rate = 1e-4
    g = load_golden("cfg1_sep")
    p = oracle_params(opt, c["seed"])

    def mk(share=None):
        e = HipEngine(opt, 768, 74, 35, seq_len=c["T"], bank_capacity=c["N"], precision="bf16", use_graph=True, share=share)
        if share is None:
            e.load_params(p)
            e.set_banks(*(banks[k] for k in "CFTAV"))
        else:
            e.set_bank_rows(share.bank_rows)
        e.set_batch(*batch)
        e.set_anchors(1, g["anchors"][0, 0]); e.set_anchors(2, g["anchors"][0, 1])
        e.set_stage2_prefetch(True)
        return e

    A = mk()
    B = mk(share=A)
    for e in (A, B, A):
        e.step()
    C_ = mk()
    for _ in range(3):
        C_.step()
    torch.cuda.synchronize()
    d = (torch.cat([A.main["p"], A.crit["p"]]) - torch.cat([C_.main["p"], C_.crit["p"]])).abs()
    assert d.max().item() <= 6.5e-4 and d.mean().item() <= 2e-5, (d.max().item(), d.mean().item())
    # and the stale-image failure this guards against is real: without the refresh the second handle steps on old critic weights
    for e in (B, A, C_):
        e.close()
