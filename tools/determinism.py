"""Diagnostic: are the stage-2 main-model gradients of a bench workload reproducible run to run (fresh engine, same inputs)?
usage: python tools/determinism.py [workload] [reps] [stage] [graph] [exact]   -- prints the worst tensors (relative to the tensor's scale)
and how many tensors are bit-identical; with `exact` the exit code is 1 unless ALL are (what MIMRL_DETERMINISTIC=1 promises:
tests/test_gpu_step.py::test_deterministic_build_is_bit_exact runs this file under that switch)."""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from tests.test_gpu_step import _bench_engine   # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
stage = int(sys.argv[3]) if len(sys.argv) > 3 else 2
graph = len(sys.argv) > 4 and sys.argv[4] == "graph"
exact = "exact" in sys.argv[5:]
from mimrl_amd import _lib   # noqa: E402
print("deterministic build:", _lib.load().mimrl_deterministic())
runs = []
anchors = None
for r in range(reps):
    opt, N, batch, banks, eng = _bench_engine(wl, "bf16", graph, device_anchors=False)
    if anchors is None:
        rng = np.random.default_rng(5)
        m = opt.batch_size // opt.k_neighbor
        anchors = np.stack([rng.choice(N, size=m, replace=False) for _ in range(6)])
    eng.set_anchors(stage, anchors)
    eng.stage_grads(stage)
    torch.cuda.synchronize()
    runs.append({n: v.double().cpu().numpy().copy() for n, v in eng.grads.items() if n.startswith("v") == (stage == 1)})
    eng.close()
for r in range(1, reps):
    # (tensors whose gradient is identically zero -- e.g. the last bias of an InfoNCE g-tower: rows of dS sum to zero -- are fp32 noise)
    rows = sorted(((np.abs(runs[r][n] - runs[0][n]).max() / (np.abs(runs[0][n]).max() + 1e-12), n) for n in runs[0]
                   if np.abs(runs[0][n]).max() > 1e-7), reverse=True)
    same = sum(np.array_equal(runs[r][n], runs[0][n]) for n in runs[0])
    print("run %d vs 0:" % r, [(float("%.2e" % a), n) for a, n in rows[:3]], "bit-identical tensors: %d of %d" % (same, len(runs[0])))
    if exact and same != len(runs[0]):
        sys.exit(1)
