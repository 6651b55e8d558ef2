"""Diagnostic: where the upload of the next batch sits relative to the steps (HIP-event timeline of the fresh-batch schedule)."""
import os, sys, torch
sys.path.insert(0, os.getcwd())
import bench
from mimrl_amd import synth
from mimrl_amd.engine import HipEngine

opt, N = bench.workload("cfg2"); B, T = opt.batch_size, opt.time_len
eng = HipEngine(opt, 768, 74, 35, seq_len=T, bank_capacity=N, precision="bf16", use_graph=True, seed=1, device_anchors=True)
eng.load_params(synth.default_state([(n, tuple(v.shape)) for n, v in eng.params.items()], 0))
eng.set_batch(*synth.synthetic_batch(B, T, seed=0)); banks = synth.synthetic_banks(N, seed=0); eng.set_banks(*(banks[k] for k in "CFTAV"))
eng.set_stage2_prefetch(True)
host = [tuple(torch.from_numpy(x).pin_memory() for x in synth.synthetic_batch(B, T, seed=100 + i)) for i in range(4)]
eng.stage_batch(*host[0])
main, cp = eng.stream, eng._copy_stream
E = lambda: torch.cuda.Event(enable_timing=True)
rows = []
base = E()
for it in range(14):
    if it == 6:
        base.record(main)
    eng.commit_batch()
    c0, c1, s0, s1 = E(), E(), E(), E()
    idle = 1 - eng._active
    with torch.cuda.stream(cp):
        cp.wait_event(eng._free_ev[idle])
        c0.record(cp)
    eng.stage_batch(*host[it % 4])
    c1.record(cp)
    s0.record(main)
    eng.step()
    s1.record(main)
    if it >= 6:
        rows.append((c0, c1, s0, s1))
torch.cuda.synchronize()
for i, (c0, c1, s0, s1) in enumerate(rows):
    print("it %d  H2D %7.3f .. %7.3f   step %7.3f .. %7.3f" % (i, base.elapsed_time(c0), base.elapsed_time(c1), base.elapsed_time(s0), base.elapsed_time(s1)))
