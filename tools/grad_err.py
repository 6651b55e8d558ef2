"""Diagnostic: relative gradient errors (vs the oracle's autograd, per tensor) of one golden case in fp32 -- top entries.
usage (GPU box, repo root): python tools/grad_err.py cfg1_cat [stage]"""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from oracle import mimrl_ref as R                      # noqa: E402  (diagnostic tool: the oracle is the checker here)
from tests.gpu_helpers import oracle_raw_grads          # noqa: E402
from tests.helpers import load_golden                   # noqa: E402
from tests.test_gpu_step import make_engine             # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "cfg1_cat"
c, opt, batch, banks, p, eng = make_engine(name)
g = load_golden(name)
anchors = g["anchors"][0]
eng.set_banks(*(banks[k] for k in "CFTAV"))
crit = [n for n in p if R.is_critic_param(n)]
main = [n for n in p if not R.is_critic_param(n)]
for stage, names in ((1, crit), (2, main)):
    eng.set_anchors(stage, anchors[stage - 1], exact_ties=bool(c.get("discrete")))
    eng.stage_grads(stage)
    torch.cuda.synchronize()
    loss, mis, pred, feats, task, grads = oracle_raw_grads(p, opt, stage, batch, banks, anchors[stage - 1], names)
    rows = []
    for n in names:
        got, want = eng.grads[n].cpu().numpy().astype(np.float64), grads[n].numpy().astype(np.float64)
        scale = np.abs(want).max() + 1e-12
        rows.append((np.abs(got - want).max() / scale, scale, n))
    rows.sort(reverse=True)
    print(f"stage {stage}: worst relative-to-scale errors")
    for r, s, n in rows[:6]:
        print(f"   {r:.3e}  scale {s:.3e}  {n}")
eng.close()
