"""Same ENGINE (same arena, same addresses), the step graph captured again and again: does the step time change with the capture?
(tools/graph_lottery.py showed 0.792-0.811 ms between engines of one process.)  A mode switch retires the captured graphs."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.test_gpu_step import _bench_engine

opt, N, batch, banks, eng = _bench_engine("cfg2", "bf16", True, dropout=0.1)
for k in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    eng.set_stage2_prefetch(0)
    eng.step()
    eng.set_stage2_prefetch(1)
    for _ in range(300):
        eng.step()
    torch.cuda.synchronize()
    ts = []
    for rep in range(2):
        t0 = time.perf_counter()
        for _ in range(300):
            eng.step()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / 300 * 1e3)
    print("capture %d: %s ms/step" % (k, " ".join("%.4f" % t for t in ts)), flush=True)
eng.close()
