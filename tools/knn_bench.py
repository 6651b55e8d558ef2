"""Stand-alone timing of the kNN product sampler (mimrl_op_knn + mimrl_op_sample_anchors) at a bench shape.
usage: python tools/knn_bench.py [N] [m] [k] [reps]     (run under tools/kstat-style rocprofv3 for per-kernel times)"""
import ctypes as C
import sys

import numpy as np
import torch

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mimrl_amd import _lib  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 16326
m = int(sys.argv[2]) if len(sys.argv) > 2 else 128
k = int(sys.argv[3]) if len(sys.argv) > 3 else 2
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 50
lib = _lib.load()
g = np.random.default_rng(0)
Z = torch.from_numpy(g.standard_normal((N, 128)).astype(np.float32)).cuda()
Z1 = torch.from_numpy(g.uniform(-3, 3, (N, 1)).astype(np.float32)).cuda()
anc = torch.zeros(1, m, dtype=torch.int32, device="cuda")
step = torch.zeros(1, dtype=torch.int32, device="cuda")
out = torch.zeros(m, k, dtype=torch.int32, device="cuda")
P = lambda t: C.c_void_p(t.data_ptr())
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for name, bank, dz in (("wide", Z, 128), ("label", Z1, 1)):
    for r in range(3):
        _lib.check(lib.mimrl_op_sample_anchors(st, P(anc), 1, m, N, 5, P(step), 101, r))
        _lib.check(lib.mimrl_op_knn(st, P(bank), dz, N, P(anc), m, k, P(out)))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for r in range(reps):
        _lib.check(lib.mimrl_op_knn(st, P(bank), dz, N, P(anc), m, k, P(out)))
    e1.record()
    torch.cuda.synchronize()
    print(f"{name}: N={N} m={m} k={k}: {e0.elapsed_time(e1) / reps * 1e3:.1f} us per call (one call of the engine's 4 + 2 per stage)")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
anc6 = torch.zeros(12, m, dtype=torch.int32, device="cuda")
e0.record()
for r in range(reps):
    _lib.check(lib.mimrl_op_sample_anchors(st, P(anc6), 12, m, N, 5, P(step), 101, r))
e1.record()
torch.cuda.synchronize()
print(f"sample_anchors x12: {e0.elapsed_time(e1) / reps * 1e3:.1f} us")
