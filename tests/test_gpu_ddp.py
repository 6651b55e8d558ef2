"""-m gpu: data parallelism with the REAL engine: two processes share the test box's one GPU (gloo backend; launched under
torch.distributed.run before anything touches the GPU)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def test_ddp_two_processes_one_gpu(tmp_path):
    """HipEngine through dist.ddp_two_stage_step (deferred-tail mode, async critic all-reduce, 1/world folded into Adam) on
    different local batches: replicas bit-identical after 3 steps and equal to single-process Adam on the mean gradient;
    broadcast -> params_changed; then Solver under world 2: rank-sharded loader (no duplicate bank rows after the
    all-gather), two epochs of train(), evaluate(), Solver.step(), replicas still identical."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29641", os.path.join(HERE, "ddp_gpu_worker.py")]
    r = subprocess.run(cmd, env=env, cwd=tmp_path, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-4000:]
    assert "DDP_GPU_OK" in r.stdout


@pytest.mark.parametrize("extra", [[], ["--no-prefetch"]])
def test_bench_two_ranks_prints_its_line(tmp_path, extra):
    """`bench.py --gpus 2` exactly as the driver launches it (torch.distributed.run, one rank per process), here with both ranks on
    the one GPU over gloo (MIMRL_DIST_BACKEND; RCCL refuses two ranks per device): the whole N > 1 flow -- deferred-tail DDP step,
    barriers, MAX over ranks, the eager profile steps on rank 0 -- ends with ONE JSON line whose value is the whole-job rate."""
    import json
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0", MIMRL_DIST_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29643", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2"] + extra
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 6 and d["scaling"] == "weak" and d["losses_finite"]
    assert abs(d["value"] - 2 * 1e3 / d["ms_per_step"]) < 1e-6 * d["value"]
    assert d["config"]["global_batch"] == 256 and d["config"]["parallelism"] == "dp2"
