"""CPU tier: CLI parity with the reference's Parameters.py (fixture captured from the reference), host-side logic
(lr schedules, anchors draw), and the gloo world_size-2 data-parallel path driven by an oracle-backed engine."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from mimrl_amd import Parameters, synth

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def test_flags_match_reference_defaults_and_readme():
    ref = json.load(open(os.path.join(HERE, "golden", "ref_flags.json")))
    for tag in ("defaults", "readme"):
        ours = vars(Parameters.parse_args(ref[tag]["argv"]))
        for k, v in ref[tag]["parsed"].items():
            assert k in ours, f"missing reference flag --{k}"
            assert ours[k] == v, f"--{k}: {ours[k]!r} != reference {v!r} ({tag})"
    with pytest.raises(SystemExit):
        Parameters.parse_args(["--cmi_last_acticate", "tanh"])       # reference: choices=['hardtanh','sigmoid']


def test_anchor_draw_consumes_numpy_rng_like_reference():
    """Model.py:81 draws np.random.choice(range(N), size=m, replace=False) once per prod_knn_sample call."""
    np.random.seed(123)
    ours = synth.draw_anchors(1000, 16, 6)
    np.random.seed(123)
    ref = np.stack([np.random.choice(range(1000), size=16, replace=False) for _ in range(6)])
    np.testing.assert_array_equal(ours, ref)
    assert all(len(set(r)) == 16 for r in ours)


def test_lr_schedules():
    from mimrl_amd.Solver import Solver
    s = Solver.__new__(Solver)
    s.opt = Parameters.parse_args(["--lr_decrease", "multi_step", "--lr_decrease_iter", "2-4", "--lr_decrease_rate", "0.1"])
    assert [round(s.lr_factor(e), 6) for e in range(6)] == [1, 1, 0.1, 0.1, 0.01, 0.01]      # MultiStepLR
    s.opt = Parameters.parse_args(["--lr_decrease", "step", "--lr_decrease_iter", "3", "--lr_decrease_rate", "0.5"])
    assert [s.lr_factor(e) for e in range(7)] == [1, 1, 1, 0.5, 0.5, 0.5, 0.25]              # StepLR
    s.opt = Parameters.parse_args(["--lr_decrease", "exp", "--lr_decrease_rate", "0.5"])
    assert [s.lr_factor(e) for e in range(3)] == [1, 0.5, 0.25]                              # ExponentialLR


@pytest.mark.parametrize("mode,task", [("min", "regression"), ("max", "classification")])
def test_lr_plateau_matches_torch_scheduler(mode, task):
    """--lr_decrease plateau (reference Solver.py:163-166 builds ReduceLROnPlateau(mode, patience=lr_decrease_iter, factor=lr_decrease_rate)
    for BOTH optimizers and steps them with val_loss, :50-52): the host-side schedule equals torch's scheduler on synthetic loss series,
    for the main and the critic rate (mi_lr_rate scaled), including the eps rule once the rates are tiny."""
    from mimrl_amd.Solver import Solver, _Plateau
    rng = np.random.default_rng(3)
    series = [np.concatenate([np.linspace(2.0, 1.0, 6), 1.0 + 0.3 * rng.random(40)]),         # improves, then a noisy plateau
              1.0 + 1e-5 * np.arange(60),                                                       # inside the relative threshold
              rng.random(120) * 3.0, np.full(80, 0.7)]
    for ser, patience, factor, lr0, rate in ((series[0], 2, 0.5, 4e-3, 1.0), (series[1], 0, 0.1, 1e-3, 0.5), (series[2], 3, 0.3, 4e-3, 2.0),
                                            (series[3], 1, 0.01, 1e-3, 1.0)):
        if mode == "max":
            ser = -ser + 4.0
        ps = [torch.nn.Parameter(torch.zeros(1)) for _ in range(2)]
        opts = [torch.optim.Adam([ps[0]], lr=lr0), torch.optim.Adam([ps[1]], lr=lr0 * rate)]
        refs = [torch.optim.lr_scheduler.ReduceLROnPlateau(o, mode=mode, patience=patience, factor=factor) for o in opts]
        s = Solver.__new__(Solver)
        s.opt = Parameters.parse_args(["--lr_decrease", "plateau", "--lr_decrease_iter", str(patience), "--lr_decrease_rate", str(factor),
                                       "--task", task, "--learning_rate", str(lr0), "--mi_lr_rate", str(rate)])
        s.base_lr = lr0
        s._plateau = [_Plateau(lr0, mode, patience, factor), _Plateau(lr0 * rate, mode, patience, factor)]
        for v in ser:
            for r in refs:
                r.step(float(v))
            s.lr_schedule_step(float(v))
            assert s._plateau[0].lr == opts[0].param_groups[0]["lr"] and s._plateau[1].lr == opts[1].param_groups[0]["lr"]
            assert s.lr_factor(0) == pytest.approx(opts[0].param_groups[0]["lr"] / lr0)
        assert s._plateau[0].lr < lr0 or patience >= len(ser)          # (every series does reduce)


def test_ddp_two_ranks_gloo():
    """world_size 2 on CPU (gloo): grads -> all-reduce(mean) -> clip+Adam equals single-process Adam on the mean."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", PYTHONPATH=ROOT)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29631", os.path.join(HERE, "ddp_gloo_worker.py")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "DDP_OK" in r.stdout


def test_bench_launcher_ladder_never_comes_back_empty():
    """`python bench.py --gpus 2` WITHOUT torchrun's environment is the launcher: it starts the ranks as children,
    and a rung that hangs (test hook) is killed by the watchdog -- ranks included, although torchrun puts them in their own sessions -- and
    replaced by fresh children on the next rung.  Without a GPU every rung fails, and the launcher STILL prints one JSON line with the
    reason of every rung, exit code 1; no rank process survives it."""
    env = dict(os.environ, PYTHONPATH=ROOT, MIMRL_DIST_BACKEND="gloo", MIMRL_BENCH_TEST_FAIL="rccl-in-graph:hang")
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "7331", "--warmup", "1", "--ddp-timeout", "8"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 1, r.stdout[-1000:] + r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    at = d["launcher"]["attempts"]
    assert d["value"] is None and d["n_gpus"] == 2 and [a["rung"] for a in at] == ["rccl-in-graph", "torch-between-graphs", "torch-between-graphs-sequential"]
    assert at[0]["reason"].startswith("watchdog") and at[0]["seconds"] < 40 and not any(a["ok"] for a in at)
    assert "needs an MI355X" in at[1]["stderr_tail"]                       # the ranks' own message, not torchrun's exit-code table
    left = []
    for pid in filter(str.isdigit, os.listdir("/proc")):
        try:
            cl = open(f"/proc/{pid}/cmdline", "rb").read()
        except OSError:
            continue
        if b"bench.py" in cl and b"7331" in cl:
            left.append(pid)
    assert not left, f"rank processes survived the watchdog: {left}"
    # the torchrun form with a mismatching --gpus still fails loudly
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--no-cpu-baseline"], env=dict(env, WORLD_SIZE="1"),
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr


def test_upload_lookahead_order_and_host_data_flag():
    """Solver._iter_loaded: batch i+1 is staged AFTER the caller's step on batch i has been enqueued (the generator resumes behind the
    yield), committed in front of the next step, and a partial last batch (another engine handle) is bound with set_batch instead.
    data.get_data_loader: --host_data parses; without a GPU the loaders hold plain host tensors either way."""
    from types import SimpleNamespace
    from mimrl_amd.Solver import Solver
    from mimrl_amd.data import get_data_loader

    log = []

    class FakeEngine:
        def __init__(self, b):
            self.cfg = SimpleNamespace(batch=b)
        def set_batch(self, t, a, v, y): log.append(("set", self.cfg.batch, int(y[0])))
        def stage_batch(self, t, a, v, y): log.append(("stage", self.cfg.batch, int(y[0])))
        def commit_batch(self): log.append(("commit", self.cfg.batch))

    engines = {}
    sol = Solver.__new__(Solver)
    sol._engine_for = lambda b, training=True: engines.setdefault(b, FakeEngine(b))
    mk = lambda i, b: (None, torch.zeros(b, 2, 3), torch.zeros(b, 2, 3), None, None, torch.full((b, 1), float(i)), torch.zeros(b, 2, 4),
                       None, None, None, None)
    loader = [mk(0, 4), mk(1, 4), mk(2, 4), mk(3, 2)]
    for e, datas in sol._iter_loaded(loader):
        log.append(("step", e.cfg.batch, int(datas[5][0])))
    assert log == [("set", 4, 0), ("step", 4, 0), ("stage", 4, 1), ("commit", 4), ("step", 4, 1), ("stage", 4, 2), ("commit", 4),
                   ("step", 4, 2), ("set", 2, 3), ("step", 2, 3)], log

    argv = ["--dataset", "synthetic", "--synthetic_n", "20", "--batch_size", "8", "--time_len", "4"]
    for extra in ([], ["--host_data"]):
        opt = Parameters.parse_args(argv + extra)
        assert bool(opt.host_data) == bool(extra)
        if not torch.cuda.is_available():
            tr = get_data_loader(opt)[0]
            assert not tr.t.is_cuda and len(tr) == 3
