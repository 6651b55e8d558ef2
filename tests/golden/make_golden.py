#!/usr/bin/env python3
"""Generate golden fixtures by IMPORTING THE REAL REFERENCE (dev container only).

    python tests/golden/make_golden.py          # writes tests/golden/*.npz

The reference (/root/reference, kiva12138/MIMRL) is imported read-only with the stubs of SURVEY.md
section 8(c) / Appendix A (``.cuda()`` -> identity, BertModel -> precomputed-feature stub).  Parameters come
from ``mimrl_amd.synth.portable_tensor`` and inputs from ``mimrl_amd.synth.synthetic_*`` -- both are
pure functions of a seed, so a fixture stores only (a) the kNN anchor draws and (b) reference OUTPUTS.
Nothing of the reference's source is copied; the fixtures are data.
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")

import torch  # noqa: E402

torch.Tensor.cuda = lambda self, *a, **k: self
torch.nn.Module.cuda = lambda self, *a, **k: self

import Model as RM  # noqa: E402  (reference)
import Customization as RC  # noqa: E402  (reference)

from mimrl_amd import synth  # noqa: E402
from tests.golden.configs import (CONFIGS, EPOCH_CONFIGS, epoch_data, grad_slice_index, make_opt, quantize_labels,  # noqa: E402
                                  split_batches)  # noqa: E402


class StubBert(torch.nn.Module):
    """Stands in for bert-base-uncased: ids[:,0] carries the sample index into a feature table."""

    def __init__(self):
        super().__init__()
        self.feat = None

    def forward(self, input_ids=None, attention_mask=None, token_type_ids=None):
        return (self.feat[input_ids[:, 0]],)


RM.BertConfig.from_pretrained = staticmethod(lambda *a, **k: None)
RM.BertModel.from_pretrained = staticmethod(lambda *a, **k: StubBert())


def build_reference_model(opt, seed):
    torch.manual_seed(0)
    model = RM.Model(opt, 768, 74, 35)
    sd = model.state_dict()
    new = {k: torch.from_numpy(synth.portable_tensor(k, tuple(v.shape), seed)) for k, v in sd.items()}
    model.load_state_dict(new)
    model.train()
    return model


def split_params(model):
    """Solver.py:119-133 split."""
    vmi, main = [], []
    for n, p in model.named_parameters():
        if "bert" in n:
            continue
        (vmi if ("vmi" in n or "vcmi" in n) else main).append((n, p))
    return vmi, main


def run_forward(model, t, a, v):
    B, T = a.shape[:2]
    model.bertmodel.feat = torch.from_numpy(t)
    ids = torch.arange(B).reshape(B, 1).repeat(1, T)
    ones = torch.ones(B, T, dtype=torch.long)
    return model(ids, torch.zeros_like(ids), ones, torch.from_numpy(a), torch.from_numpy(v), return_features=True)


def stage_pass(model, opt, stage, batch, banks_t, seed_np):
    """Solver.train loop body up to backward (Solver.py:205-212 / 221-234) through the reference's own
    Customization.compute_custumized_loss; returns loss, mis, anchors, outputs."""
    t, a, v, y = batch
    labels = torch.from_numpy(y)
    outputs = run_forward(model, t, a, v)
    task = torch.nn.L1Loss()(outputs[0].reshape(-1), labels.reshape(-1))       # Solver.py:181-182,334-335
    N = banks_t[0].shape[0] if len(banks_t[0]) else 0
    anchors = np.zeros((6, 0), np.int64)
    if N:
        m = y.shape[0] // opt.k_neighbor
        np.random.seed(seed_np)
        anchors = synth.draw_anchors(N, m, 6)
        probe = np.random.random()
        np.random.seed(seed_np)                                                # replay for the reference
    task_val = task.item()            # NB: Customization.py:109-111 does `loss = task_loss; loss += ...` IN PLACE
    wrapper = types.SimpleNamespace(module=model)                              # DataParallel.module
    loss, mis = RC.compute_custumized_loss(wrapper, task, outputs, labels, None, opt, stage, *banks_t)
    if N:
        assert np.random.random() == probe, "anchor replay diverged from the reference's RNG use"
    return loss, mis, anchors, outputs, task_val


def gen(name, c):
    opt = make_opt(c)
    seed = c["seed"]
    B, T, N = c["B"], c["T"], c["N"]
    out = {}
    batch = synth.synthetic_batch(B, T, seed=seed, ragged=c.get("ragged", False))
    banks = synth.synthetic_banks(N, seed=seed)
    batch, banks = quantize_labels(c, batch, banks)
    banks_t = [torch.from_numpy(banks[k]) for k in "CFTAV"]
    model = build_reference_model(opt, seed)
    vmi, main = split_params(model)
    opt_main = torch.optim.Adam([{"params": [p for _, p in main], "lr": opt.learning_rate}],
                                lr=opt.learning_rate, weight_decay=opt.weight_decay)
    opt_vmi = torch.optim.Adam([{"params": [p for _, p in vmi], "lr": opt.learning_rate * opt.mi_lr_rate}],
                               lr=opt.learning_rate, weight_decay=opt.weight_decay)
    all_params = [p for p in model.parameters() if p.requires_grad]

    # ---- F5: forward
    with torch.no_grad():
        o = run_forward(model, *batch[:3])
    for k, val in zip(["pred", "F_F", "T_F", "A_F", "V_F"], o):
        out["fwd_" + k] = val.numpy().copy()
    slices = c.get("slices", False)

    # ---- F9: epoch-0 rule (empty banks)
    loss0, mis0, _, _, task0 = stage_pass(model, opt, 2, batch, [[]] * 5, 0)
    out["e0_stage2_loss"] = np.float32(loss0.item())
    out["e0_stage2_mis"] = np.array([float(m) for m in mis0], np.float32)
    l01, _, _, _, _ = stage_pass(model, opt, 1, batch, [[]] * 5, 0)
    out["e0_stage1_loss"] = np.float32(float(l01))

    n_steps = c.get("traj", 1)
    traj = {k: [] for k in ["s1_loss", "s1_mis", "s2_loss", "s2_mis", "s2_task"]}
    anchors_all = []
    for it in range(n_steps):
        # ---- stage 1 (Solver.py:205-214)
        loss, mis, anc1, _, _ = stage_pass(model, opt, 1, batch, banks_t, 1000 + 2 * it)
        opt_vmi.zero_grad()
        for p in all_params:
            p.grad = None
        loss.backward()
        if it == 0:
            out["s1_gnorm_names"] = np.array([n for n, _ in vmi])
            out["s1_gnorm"] = np.array([p.grad.norm().item() for _, p in vmi], np.float64)
            out["s1_gsum"] = np.array([p.grad.sum().item() for _, p in vmi], np.float64)
            for n, p in vmi:
                if p.numel() <= 512:
                    out["s1_grad:" + n] = p.grad.numpy().copy()
                elif slices:
                    out["s1_gslice:" + n] = p.grad.numpy().reshape(-1)[grad_slice_index(p.numel())].copy()
        torch.nn.utils.clip_grad_value_(all_params, opt.gradient_clip)
        opt_vmi.step()
        traj["s1_loss"].append(loss.item())
        traj["s1_mis"].append([float(m) for m in mis])
        if it == 0:
            out["s1_psum_after"] = np.array([p.detach().double().sum().item() for _, p in vmi])
            out["s1_psq_after"] = np.array([(p.detach().double() ** 2).sum().item() for _, p in vmi])
        # ---- stage 2 (Solver.py:221-236)
        loss, mis, anc2, outputs, task = stage_pass(model, opt, 2, batch, banks_t, 1001 + 2 * it)
        opt_main.zero_grad()
        for p in all_params:
            p.grad = None
        loss.backward()
        if it == 0:
            out["s2_gnorm_names"] = np.array([n for n, _ in main])
            out["s2_gnorm"] = np.array([p.grad.norm().item() for _, p in main], np.float64)
            out["s2_gsum"] = np.array([p.grad.sum().item() for _, p in main], np.float64)
            for n, p in main:
                if p.numel() <= 512:
                    out["s2_grad:" + n] = p.grad.numpy().copy()
                elif slices:
                    out["s2_gslice:" + n] = p.grad.numpy().reshape(-1)[grad_slice_index(p.numel())].copy()
            out["s2_pred"] = outputs[0].detach().numpy().copy()
        torch.nn.utils.clip_grad_value_(all_params, opt.gradient_clip)
        opt_main.step()
        traj["s2_loss"].append(loss.item())
        traj["s2_mis"].append([float(m) for m in mis])
        traj["s2_task"].append(task)
        if it == 0:
            out["s2_psum_after"] = np.array([p.detach().double().sum().item() for _, p in main])
            out["s2_psq_after"] = np.array([(p.detach().double() ** 2).sum().item() for _, p in main])
        anchors_all.append(np.stack([anc1, anc2]))
    out["anchors"] = np.stack(anchors_all)                     # [steps, 2, 6, m]
    for k, val in traj.items():
        out["traj_" + k] = np.array(val, np.float64)
    path = os.path.join(HERE, f"{name}.npz")
    np.savez_compressed(path, **out)
    print(f"{name}: wrote {path} ({os.path.getsize(path)/1024:.1f} KiB); s1_loss0={traj['s1_loss'][0]:.6f} "
          f"s2_loss0={traj['s2_loss'][0]:.6f} s2_mis0={np.round(traj['s2_mis'][0], 5).tolist()}")


def gen_epoch(name, c):
    """Epoch fixture through the reference's OWN Solver (Solver.py:18-36 ctor, :194-248 train, :250-270 evaluate), stubs of
    SURVEY.md Appendix A.  Records what Solver.solve would log per epoch, the feature banks handed from epoch to epoch,
    and every anchor draw (np.random.choice, Model.py:81) in call order."""
    import random
    sys.modules.setdefault("torch.utils.tensorboard", types.SimpleNamespace(
        SummaryWriter=lambda *a, **k: types.SimpleNamespace(add_scalar=lambda *a, **k: None, close=lambda: None)))
    sys.modules.setdefault("DataLoaderLocal", types.SimpleNamespace(mosi_r2c_7=None, pom_r2c_7=None, r2c_2=None, r2c_7=None,
                                                                    LocalDataset=None))
    import transformers
    transformers.BertTokenizer.from_pretrained = classmethod(lambda cls, *a, **k: None)
    import Parameters as RP  # reference
    import Solver as RS      # reference

    B, T = c["B"], c["T"]
    data = epoch_data(c)
    text = np.concatenate([data[k][0] for k in ("train", "valid", "test")])
    base = {"train": 0, "valid": c["n_train"], "test": c["n_train"] + c["n_valid"]}

    def loader(k):
        out = []
        for j, (t, a, v, y) in enumerate(split_batches(data[k], B)):
            ids = (base[k] + j * B + np.arange(len(y))).reshape(-1, 1).repeat(T, 1)
            out.append((None, torch.from_numpy(a), torch.from_numpy(v), None, None, torch.from_numpy(y).reshape(-1, 1), ids,
                        np.zeros_like(ids), np.ones_like(ids), None, None))
        return out

    loaders = (loader("train"), loader("valid"), loader("test"), 768, 74, 35)
    RS.get_data_loader = lambda opt: loaders
    one = lambda n, v: "-".join([str(v)] * n)
    sys.argv = ["Main.py", "--dataset", "mosi_Dec", "--parallel", "--text", "none", "--batch_size", str(B), "--time_len", str(T),
                "--d_hiddens", c["cube"], "--d_outs", c["cube"], "--bias", "--res_project", "1-1", "--dropout", "0.0-0.0-0.0-0.0",
                "--dropout_mlp", "0.0-0.0-0.0", "--loss_mi_coefficient1", one(11, 1.0), "--loss_mi_coefficient2", one(8, 0.01),
                "--stage1_n", str(c["stage1_n"]), "--lr_decrease", "multi_step", "--lr_decrease_iter", c["lr_iter"],
                "--lr_decrease_rate", str(c["lr_rate"]), "--learning_rate", str(c["lr"]), "--critic_type", c["critic"],
                "--bound_type", "infonce", "--k_neighbor", "2", "--gradient_clip", "1.5", "--loss", "MAE", "--optm", "Adam",
                "--encoders", "gru", "--d_common", "128", "--activate", "gelu", "--task_name", "golden_" + name,
                "--epochs_num", str(c["epochs"])]
    opt = RP.parse_args()
    random.seed(0); np.random.seed(0); torch.manual_seed(0)
    sol = RS.Solver(opt)                       # nn.DataParallel without GPUs calls the module directly (Solver.py:33-35)
    model = sol.model.module
    sd = model.state_dict()
    model.load_state_dict({k: torch.from_numpy(synth.portable_tensor(k, tuple(v.shape), c["seed"])) for k, v in sd.items()})
    model.bertmodel.feat = torch.from_numpy(text)

    draws = []
    real_choice = np.random.choice

    def rec_choice(a, size=None, replace=True, p=None):
        r = real_choice(a, size=size, replace=replace, p=p)
        draws.append(np.asarray(r, np.int64).copy())
        return r

    np.random.choice = rec_choice
    np.random.seed(c["seed"])
    out = {}
    try:
        banks = ([], [], [], [], [])
        for ep in range(c["epochs"]):
            r = sol.train(ep, sol.train_loader, *banks)
            banks = r[4:]
            out[f"ep{ep}_train_loss"], out[f"ep{ep}_train_loss_mi"] = np.float64(r[0]), np.float64(r[1])
            out[f"ep{ep}_train_mis"] = np.array(r[2], np.float64)
            out[f"ep{ep}_train_mae"], out[f"ep{ep}_train_corr"] = np.float64(r[3]["mae"]), np.float64(r[3]["corr"])
            for k, bk in zip("CFTAV", banks):
                out[f"ep{ep}_bank_{k}"] = bk.detach().numpy().astype(np.float32).copy()
            for tag, ld in (("valid", sol.valid_loader), ("test", sol.test_loader)):
                e = sol.evaluate(ld, *banks)
                out[f"ep{ep}_{tag}_loss"], out[f"ep{ep}_{tag}_mis"] = np.float64(e[0]), np.array(e[1], np.float64)
                out[f"ep{ep}_{tag}_mae"] = np.float64(e[2]["mae"])
                out[f"ep{ep}_{tag}_pred"] = np.asarray(e[3], np.float32).reshape(-1)
            sol.lr_schedule_main.step(); sol.lr_schedule_vmi.step()          # Solver.py:52-57
            out[f"ep{ep}_lr_next"] = np.array([sol.optimizer_main.param_groups[1]["lr"], sol.optimizer_vmi.param_groups[0]["lr"]])
    finally:
        np.random.choice = real_choice
    names = [n for n, _ in model.named_parameters() if "bert" not in n]
    out["final_names"] = np.array(names)
    out["final_psum"] = np.array([p.detach().double().sum().item() for n, p in model.named_parameters() if "bert" not in n])
    out["final_psq"] = np.array([(p.detach().double() ** 2).sum().item() for n, p in model.named_parameters() if "bert" not in n])
    out["draw_len"] = np.array([len(d) for d in draws], np.int64)
    out["draws"] = np.concatenate(draws) if draws else np.zeros(0, np.int64)
    path = os.path.join(HERE, f"{name}.npz")
    np.savez_compressed(path, **out)
    print(f"{name}: wrote {path} ({os.path.getsize(path)/1024:.1f} KiB); {len(draws)} anchor draws; " +
          " | ".join(f"ep{e}: loss {out[f'ep{e}_train_loss']:.5f} mi {out[f'ep{e}_train_loss_mi']:.5f} val {out[f'ep{e}_valid_loss']:.5f}"
                     for e in range(c["epochs"])))


def gen_units():
    """F1/F2: estimator-level fixtures straight from the reference classes (all bounds, both critics)."""
    out = {}
    B, N, k = 16, 60, 2
    g = np.random.Generator(np.random.PCG64(7))
    x = g.standard_normal((B, 128)).astype(np.float32)
    y = (0.5 * x + g.standard_normal((B, 128))).astype(np.float32)
    out["x"], out["y"] = x, y
    for critic in ("separate", "concat"):
        for bound in ("infonce", "nwj", "tuba", "dv", "js", "js_fgan", "smile", "interpolate"):
            est = RM.VMIEstimator(critic, "constant", bound, 128, 256, 128, 2, "relu", 0, 1)
            sd = est.state_dict()
            pre = "vmi_estimator_f_t."
            est.load_state_dict({kk: torch.from_numpy(synth.portable_tensor(pre + kk, tuple(vv.shape), 3))
                                 for kk, vv in sd.items()})
            xt = torch.from_numpy(x).requires_grad_(True)
            yt = torch.from_numpy(y).requires_grad_(True)
            mi, loss = est(xt, yt)
            loss.backward()
            out[f"{critic}_{bound}_mi"] = np.float64(mi.item())
            out[f"{critic}_{bound}_dx"] = xt.grad.numpy().copy()
            out[f"{critic}_{bound}_dy"] = yt.grad.numpy().copy()
            if bound == "infonce":
                out[f"{critic}_scores"] = est.critic_model(xt, yt).detach().numpy().copy()
    # CMI
    banks = synth.synthetic_banks(N, seed=5)
    z = g.standard_normal((B, 128)).astype(np.float32)
    c = g.uniform(-3, 3, size=(B, 1)).astype(np.float32)
    out["z"], out["c"] = z, c
    for last in ("sigmoid", "hardtanh"):
        est = RM.VCMIEstimator(128, 256, 2, "relu", k, 1.0, last)
        pre = "vcmi_estimator_ta_c."
        est.load_state_dict({kk: torch.from_numpy(synth.portable_tensor(pre + kk, tuple(vv.shape), 3))
                             for kk, vv in est.state_dict().items()})
        np.random.seed(11)
        anchors = synth.draw_anchors(N, B // k, 1)[0]
        np.random.seed(11)
        Xb, Yb, Zb = (torch.from_numpy(banks[n]) for n in ("T", "A", "C"))
        kx, ky, kz = RM.prod_knn_sample(Xb, Yb, Zb, B, k, 1.0)
        xt = torch.from_numpy(x).requires_grad_(True)
        yt = torch.from_numpy(y).requires_grad_(True)
        ct = torch.from_numpy(c)
        cmi, loss = est(xt, yt, ct, kx, ky, kz)
        (cmi + loss).backward()
        out[f"cmi_{last}_anchors"] = anchors
        out[f"cmi_{last}_kx"] = kx.detach().numpy().copy()
        out[f"cmi_{last}_kz_col0"] = kz.detach().numpy()[:, 0].copy()
        out[f"cmi_{last}_cmi"] = np.float64(cmi.item())
        out[f"cmi_{last}_bce"] = np.float64(loss.item())
        out[f"cmi_{last}_dx"] = xt.grad.numpy().copy()
        out[f"cmi_{last}_dy"] = yt.grad.numpy().copy()
    path = os.path.join(HERE, "units.npz")
    np.savez_compressed(path, **out)
    print(f"units: wrote {path} ({os.path.getsize(path)/1024:.1f} KiB)")


def gen_flags():
    """Defaults and a README-style invocation parsed by the REFERENCE's Parameters.parse_args (Parameters.py:4-74)."""
    import json
    import Parameters as RP  # reference
    out = {}
    for tag, argv in (("defaults", []), ("readme", ["--d_hiddens", "50-3-128=10-3-128", "--d_outs", "50-3-128=10-3-128",
                                                   "--dropout_mlp", "0.0-0.0-0.0", "--dropout", "0.1-0.1-0.1-0.1", "--bias",
                                                   "--res_project", "1-1", "--critic_type", "separate", "--bound_type",
                                                   "infonce", "--k_neighbor", "2", "--stage1_n", "2", "--gradient_clip", "1.5",
                                                   "--learning_rate", "4e-3", "--loss", "MAE", "--optm", "Adam"])):
        sys.argv = ["Main.py"] + argv
        out[tag] = {"argv": argv, "parsed": vars(RP.parse_args())}
    path = os.path.join(HERE, "ref_flags.json")
    json.dump(out, open(path, "w"), indent=1, sort_keys=True)
    print("flags: wrote", path)


if __name__ == "__main__":
    os.chdir("/tmp")
    which = sys.argv[1:] or list(CONFIGS) + ["units", "flags"] + list(EPOCH_CONFIGS)
    for name in which:
        if name in EPOCH_CONFIGS:
            gen_epoch(name, EPOCH_CONFIGS[name])
        elif name == "units":
            gen_units()
        elif name == "flags":
            gen_flags()
        else:
            gen(name, CONFIGS[name])
