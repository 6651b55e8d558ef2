"""Where the four recurrence launches sit inside the replayed step graph, WITHOUT a profiler attached: the in-kernel launch stamps
(mimrl_set_kernel_stamps: first workgroup start / last workgroup end, 100 MHz wall clock) of N steady-state steps, printed relative to the
layer-0 forward launch of each step.  The gaps between them are the non-recurrent segments of the critical path (projection GEMM between
the forward layers; tail + estimators + CubeMLP backward between forward l1 and BPTT l1; dh0 between the BPTT launches; BPTT l0 end ->
next step's forward l0 start = weight-gradient tail + Adam + the prefix's front end).  GPU box: python tools/stamp_gaps.py [workload] [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from mimrl_amd import synth
from mimrl_amd.engine import HipEngine

wl = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
opt, N = bench.workload(wl)
B, T = opt.batch_size, opt.time_len
eng = HipEngine(opt, 768, 74, 35, seq_len=T, bank_capacity=N, precision="bf16", use_graph=True, seed=1234, device_anchors=True)
eng.load_params(synth.default_state([(n, tuple(v.shape)) for n, v in eng.params.items()], 0))
eng.set_batch(*synth.synthetic_batch(B, T, seed=0))
banks = synth.synthetic_banks(N, seed=0)
eng.set_banks(*(banks[k] for k in "CFTAV"))
eng.set_stage2_prefetch(1)
eng.kernel_stamps(1 << 10)          # before the capture: the ring pointer is a kernel argument of the captured launches
for _ in range(8):
    eng.step()
torch.cuda.synchronize()
eng.kernel_stamps(1 << 10)
for _ in range(steps):
    eng.step()
torch.cuda.synchronize()
r = eng._stamps.cpu().numpy().view(np.uint64)
full = np.uint64(0xFFFFFFFFFFFFFFFF)
per_id = []
for i in range(4):       # (the forward launches are stamped under stage 1's step counter, the BPTT launches under stage 2's: different ring slots)
    ok = (r[:, i, 0] != full) & (r[:, i, 1] != full)
    v = sorted(zip((r[ok, i, 0].astype(np.float64) / 100.0).tolist(), ((~r[ok, i, 1]).astype(np.float64) / 100.0).tolist()))
    per_id.append(v)
n = min(len(v) for v in per_id)
# align: the k-th launch of each kind; drop leading entries of kinds that start before the first layer-0 forward launch
t_first = per_id[0][0][0]
per_id = [[x for x in v if x[0] >= t_first] for v in per_id]
n = min(len(v) for v in per_id)
rows = [[per_id[i][k] for i in range(4)] for k in range(n)]
names = eng.STAMP_IDS
print("per step, us relative to the start of %s:" % names[0])
seg = []
for k in range(1, len(rows) - 1):
    t0 = rows[k][0][0]
    nxt = rows[k + 1][0][0]
    line = "  ".join("%s %7.1f..%7.1f (%5.1f)" % (names[i][4:], rows[k][i][0] - t0, rows[k][i][1] - t0, rows[k][i][1] - rows[k][i][0]) for i in range(4))
    print(line + "   period %.1f" % (nxt - t0))
    seg.append([rows[k][0][1] - rows[k][0][0], rows[k][1][0] - rows[k][0][1], rows[k][1][1] - rows[k][1][0], rows[k][2][0] - rows[k][1][1],
                rows[k][2][1] - rows[k][2][0], rows[k][3][0] - rows[k][2][1], rows[k][3][1] - rows[k][3][0], nxt - rows[k][3][1], nxt - t0])
seg = np.median(np.array(seg), axis=0)
lab = ["fwd l0", "gap (l1 projection)", "fwd l1", "gap (tails, estimators, CubeMLP backward, LN backward)", "BPTT l1", "gap (dh0)", "BPTT l0",
       "gap (weight-gradient tail, Adam, next prefix front end)", "period"]
print("median segments (us):")
for l, v in zip(lab, seg):
    print("  %-60s %8.1f" % (l, v))
