#!/usr/bin/env python3
"""Bring-up probe: create an engine for a bench workload, run a few two-stage steps, report finiteness / time / memory.
    python tools/bringup.py cfg3 bf16 [graph] [prefetch]"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from mimrl_amd import _lib, synth
from mimrl_amd.engine import HipEngine

name, prec = sys.argv[1], sys.argv[2]
graph, pre = "graph" in sys.argv[3:], "prefetch" in sys.argv[3:]
opt, N = bench.workload(name)
B, T = opt.batch_size, opt.time_len
t0 = time.time()
eng = HipEngine(opt, 768, 74, 35, seq_len=T, bank_capacity=N, precision=prec, use_graph=graph, seed=1, device_anchors=True)
print(f"{name} {prec} graph={graph} prefetch={pre}: workspace {eng.workspace_bytes() / 2**30:.2f} GiB, create {time.time() - t0:.1f}s", flush=True)
eng.load_params(synth.default_state([(n, tuple(v.shape)) for n, v in eng.params.items()], 0))
eng.set_batch(*synth.synthetic_batch(B, T, seed=0))
banks = synth.synthetic_banks(N, seed=0)
eng.set_banks(*(banks[k] for k in "CFTAV"))
eng.set_stage2_prefetch(pre)
for i in range(3):
    eng.step()
    torch.cuda.synchronize()
    s = eng.read_scalars()
    print(f" step {i}: s1 {s[_lib.S1_LOSS]:.5f} s2 {s[_lib.S2_LOSS]:.5f} task {s[_lib.S2_TASK]:.5f} finite={np.isfinite(s).all()} "
          f"mis2 {np.round(s[_lib.S2_MIS:_lib.S2_MIS + 8], 4).tolist()}", flush=True)
n = 5
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(n):
    eng.step()
torch.cuda.synchronize()
print(f" {1e3 * (time.perf_counter() - t0) / n:.2f} ms/step; params finite: "
      f"{bool(torch.isfinite(eng.main['p']).all() and torch.isfinite(eng.crit['p']).all())}", flush=True)
eng.close()
