"""Diagnostic: stage-2 main-model gradients of cfg2-concat (bf16), fused vs chain concat critic, and chain vs chain (noise floor)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from tests.test_gpu_step import _bench_engine   # noqa: E402


def run(no_fused, anchors=None):
    if no_fused:
        os.environ["MIMRL_NO_FUSED_CONCAT"] = "1"
    else:
        os.environ.pop("MIMRL_NO_FUSED_CONCAT", None)
    opt, N, batch, banks, eng = _bench_engine("cfg2-concat", "bf16", False, device_anchors=False)
    if anchors is None:
        rng = np.random.default_rng(5)
        m = opt.batch_size // opt.k_neighbor
        anchors = np.stack([rng.choice(N, size=m, replace=False) for _ in range(6)])
    eng.set_anchors(2, anchors)
    eng.stage_grads(2)
    torch.cuda.synchronize()
    g = {n: v.double().cpu().numpy().copy() for n, v in eng.grads.items() if not n.startswith("v")}
    s = eng.read_scalars().copy()
    eng.close()
    return anchors, s, g


anc, s1, c1 = run(True)
_, s2, c2 = run(True, anc)
_, s3, f1 = run(False, anc)


def worst(a, b, k=5):
    rows = sorted(((np.abs(a[n] - b[n]).max() / (np.abs(b[n]).max() + 1e-12), n) for n in a), reverse=True)
    return rows[:k]


_, s4, f2 = run(False, anc)
print("chain vs chain:", worst(c1, c2, 3))
print("fused vs chain:", worst(f1, c1, 3))
print("fused vs fused:", worst(f1, f2, 3))
