"""Diagnostic: dump the gradients of one stage of a bench workload to an .npz (compare builds / knobs: tools/grad_dump.py cfg2 2 graph out.npz;
with two more arguments `cmp a.npz b.npz` prints the worst tensors of a vs b relative to the tensor's scale)."""
import sys

import numpy as np

if sys.argv[1] == "cmp":
    a, b = dict(np.load(sys.argv[2])), dict(np.load(sys.argv[3]))
    rows = sorted(((np.abs(a[n].astype(np.float64) - b[n]).max() / (np.abs(a[n]).max() + 1e-12), n) for n in a if np.abs(a[n]).max() > 1e-7),
                  reverse=True)
    print([(float("%.2e" % x), n) for x, n in rows[:6]])
    sys.exit(0)
import torch

sys.path.insert(0, ".")
from tests.test_gpu_step import _bench_engine   # noqa: E402

wl, stage, graph, out = sys.argv[1], int(sys.argv[2]), sys.argv[3] == "graph", sys.argv[4]
opt, N, batch, banks, eng = _bench_engine(wl, "bf16", graph, device_anchors=False)
rng = np.random.default_rng(5)
eng.set_anchors(stage, np.stack([rng.choice(N, size=opt.batch_size // opt.k_neighbor, replace=False) for _ in range(6)]))
eng.stage_grads(stage)
torch.cuda.synchronize()
np.savez(out, **{n: v.cpu().numpy().copy() for n, v in eng.grads.items() if n.startswith("v") == (stage == 1)})
