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
