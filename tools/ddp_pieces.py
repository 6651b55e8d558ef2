"""Diagnostic (one GPU): HIP-event durations of the pieces of the data-parallel step (dist.ddp_two_stage_step without collectives)."""
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
import bench
from mimrl_amd import synth
from mimrl_amd.engine import HipEngine

opt, N = bench.workload("cfg2"); B, T = opt.batch_size, opt.time_len
eng = HipEngine(opt, 768, 74, 35, seq_len=T, bank_capacity=N, precision="bf16", use_graph=True, seed=1, device_anchors=True)
eng.load_params(synth.default_state([(n, tuple(v.shape)) for n, v in eng.params.items()], 0))
eng.set_batch(*synth.synthetic_batch(B, T, seed=0)); banks = synth.synthetic_banks(N, seed=0); eng.set_banks(*(banks[k] for k in "CFTAV"))
eng.set_stage2_prefetch(2)
E = lambda: torch.cuda.Event(enable_timing=True)
names = ["stage_grads(1)", "stage2_forward_tail", "stage_apply(1)", "stage_grads(2)", "stage_apply(2)"]
fns = [lambda: eng.stage_grads(1), eng.stage2_forward_tail, lambda: eng.stage_apply(1), lambda: eng.stage_grads(2), lambda: eng.stage_apply(2)]
acc = [0.0] * 5; n = 0
for it in range(60):
    ev = [E() for _ in range(6)]
    ev[0].record()
    for i, f in enumerate(fns):
        f(); ev[i + 1].record()
    torch.cuda.synchronize()
    if it >= 10:
        n += 1
        for i in range(5): acc[i] += ev[i].elapsed_time(ev[i + 1])
for nm, a in zip(names, acc): print("%-22s %.3f ms" % (nm, a / n))
print("%-22s %.3f ms (host-synchronised per step: pieces do not overlap launch latency)" % ("sum", sum(acc) / n))
eng.set_stage2_prefetch(True)
for _ in range(10): eng.step()
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(100): eng.step()
torch.cuda.synchronize(); print("combined step          %.3f ms" % (1e3 * (time.perf_counter() - t) / 100))
