"""Is the step time of the captured two-stage graph a property of the PROCESS or of the capture / instantiation?
tools/graph_lottery.py [engines] [steps]: builds the cfg2 bench engine several times in ONE process and times each (ms per step)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.test_gpu_step import _bench_engine

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
out = []
for k in range(n):
    opt, N, batch, banks, eng = _bench_engine("cfg2", "bf16", True, dropout=0.1)
    eng.set_stage2_prefetch(1)                       # Solver.step overlap mode: both stages ONE captured graph (what bench.py times)
    for _ in range(400):
        eng.step()
    torch.cuda.synchronize()
    ts = []
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(steps):
            eng.step()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / steps * 1e3)
    out.append(ts)
    print("engine %d: %s ms/step" % (k, " ".join("%.4f" % t for t in ts)), flush=True)
    eng.close()
