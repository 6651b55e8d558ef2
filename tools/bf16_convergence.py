#!/usr/bin/env python3
"""Does the benchmarked mode (bf16 MFMA operands, fused kernels, hipGraph, Solver.step overlap) TRAIN like fp32?
100 two-stage iterations over 4 cycling cfg1-shaped batches from the same initialisation, dropout off, host-drawn anchors
shared by all runs; prints task-MAE and the 8 MI/CMI series (means over windows of 20 steps) for
  fp32 eager sequential | fp32 graph+overlap (same arithmetic, different summation order = the chaos floor) | bench mode."""
import copy, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mimrl_amd import _lib, synth
from mimrl_amd.engine import HipEngine
from tests.golden.configs import CONFIGS, make_opt


def run(precision, graph, prefetch, steps=100, lr=4e-3, name="cfg1_sep"):
    c = dict(CONFIGS[name], lr=lr)
    opt = make_opt(c)
    eng = HipEngine(opt, 768, 74, 35, seq_len=c["T"], bank_capacity=c["N"], precision=precision, use_graph=graph, seed=0)
    eng.load_params(synth.default_state([(n, tuple(v.shape)) for n, v in eng.params.items()], 0))
    banks = synth.synthetic_banks(c["N"], seed=0)
    eng.set_banks(*(banks[k] for k in "CFTAV"))
    eng.set_stage2_prefetch(prefetch)
    batches = [synth.synthetic_batch(c["B"], c["T"], seed=10 + i) for i in range(4)]
    rng = np.random.default_rng(0)
    m = c["B"] // 2
    rec = []
    for it in range(steps):
        eng.set_batch(*batches[it % 4])
        for st in (1, 2):
            eng.set_anchors(st, np.stack([rng.choice(c["N"], size=m, replace=False) for _ in range(6)]))
        eng.step()
        s = eng.read_scalars()
        rec.append(np.concatenate([[s[_lib.S2_TASK], s[_lib.S1_LOSS]], s[_lib.S2_MIS:_lib.S2_MIS + 8]]))
    eng.close()
    return np.array(rec)


if __name__ == "__main__":
    lr = float(sys.argv[1]) if len(sys.argv) > 1 else 4e-3
    runs = {"fp32 eager": run("fp32", False, False, lr=lr), "fp32 graph+overlap": run("fp32", True, True, lr=lr),
            "bench mode": run("bf16", True, True, lr=lr)}
    names = ["task", "s1loss", "f_t", "f_a", "f_v", "inv", "spec_t", "spec_a", "spec_v", "comp"]
    for w in range(0, 100, 20):
        print(f"--- steps {w}..{w + 19} (window means)")
        for k, r in runs.items():
            print(f"  {k:20s} " + " ".join(f"{n}={v:8.4f}" for n, v in zip(names, r[w:w + 20].mean(0))))
    a, b, c = runs["fp32 eager"], runs["fp32 graph+overlap"], runs["bench mode"]
    print("max |window-mean gap| fp32-vs-fp32 :", np.round(np.abs(a.reshape(5, 20, -1).mean(1) - b.reshape(5, 20, -1).mean(1)).max(0), 4).tolist())
    print("max |window-mean gap| bench-vs-fp32:", np.round(np.abs(a.reshape(5, 20, -1).mean(1) - c.reshape(5, 20, -1).mean(1)).max(0), 4).tolist())
