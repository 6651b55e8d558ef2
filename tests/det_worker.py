"""Run under MIMRL_DETERMINISTIC=1 by tests/test_gpu_step.py::test_deterministic_build_is_bit_exact (the library is chosen at import, so
the deterministic build gets its own process).  usage: det_worker.py <workload> <stage> <graph 0|1> <out.npz>
  (a) three fresh engines, the same inputs: every gradient tensor of the stage bit-identical;
  (b) two fresh engines, three full two-stage steps each (device-drawn anchors, Adam, dropout on): every parameter, both Adam moments and
      all scalars bit-identical;
  (c) the gradients of (a) go to <out.npz>: the parent compares them with the default build's."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from mimrl_amd import _lib                               # noqa: E402
from tests.test_gpu_step import _bench_engine            # noqa: E402


def main():
    workload, stage, graph, out = sys.argv[1], int(sys.argv[2]), sys.argv[3] == "1", sys.argv[4]
    assert _lib.DETERMINISTIC and _lib.load().mimrl_deterministic() == 1, "not the deterministic build"
    runs, anchors = [], None
    for r in range(3):
        opt, N, batch, banks, eng = _bench_engine(workload, "bf16", graph, device_anchors=False)
        if anchors is None:
            rng = np.random.default_rng(5)
            anchors = np.stack([rng.choice(N, size=opt.batch_size // opt.k_neighbor, replace=False) for _ in range(6)])
        eng.set_anchors(stage, anchors)
        eng.stage_grads(stage)
        torch.cuda.synchronize()
        runs.append({n: v.cpu().numpy().copy() for n, v in eng.grads.items() if n.startswith("v") == (stage == 1)})
        eng.close()
    for r in (1, 2):
        diff = [n for n in runs[0] if not np.array_equal(runs[r][n], runs[0][n])]
        assert not diff, ("gradients differ between runs of the deterministic build", r, diff[:5])
    np.savez(out, **runs[0])
    print("grads bit-identical:", len(runs[0]), "tensors x 3 runs")

    states = []
    for r in range(2):
        opt, N, batch, banks, eng = _bench_engine(workload, "bf16", graph, device_anchors=True, dropout=0.1)
        scal = []
        for _ in range(3):
            eng.step()
            torch.cuda.synchronize()
            scal.append(eng.scalars.cpu().numpy().copy())
        st = {n: v.cpu().numpy().copy() for n, v in eng.params.items()}
        for grp, bucket in (("main", eng.main), ("crit", eng.crit)):
            st[grp + ".m"] = bucket["m"].cpu().numpy().copy()
            st[grp + ".v"] = bucket["v"].cpu().numpy().copy()
        st["scalars"] = np.stack(scal)
        states.append(st)
        eng.close()
    diff = [n for n in states[0] if not np.array_equal(states[0][n], states[1][n], equal_nan=True)]
    assert not diff, ("three full steps differ between runs of the deterministic build", diff[:5])
    assert np.isfinite(states[0]["scalars"]).all()
    assert _lib.load().mimrl_deterministic() == 1, "the accumulation table overflowed"
    print("3 steps bit-identical:", len(states[0]), "arrays")


if __name__ == "__main__":
    main()
