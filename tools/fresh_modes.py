import os, sys, time, torch
sys.path.insert(0, os.getcwd())
import bench
from mimrl_amd import synth, dist as mdist
from mimrl_amd.engine import HipEngine
opt, N = bench.workload("cfg2"); B, T = opt.batch_size, opt.time_len
eng = HipEngine(opt, 768, 74, 35, seq_len=T, bank_capacity=N, precision="bf16", use_graph=True, seed=1, device_anchors=True)
eng.load_params(synth.default_state([(n, tuple(v.shape)) for n, v in eng.params.items()], 0))
eng.set_batch(*synth.synthetic_batch(B, T, seed=0)); banks = synth.synthetic_banks(N, seed=0); eng.set_banks(*(banks[k] for k in "CFTAV"))
host = [tuple(torch.from_numpy(x).pin_memory() for x in synth.synthetic_batch(B, T, seed=100 + i)) for i in range(4)]
def timed(fn, n=150):
    for _ in range(6): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t) / n
i = [0]
for mode, stepfn in ((True, eng.step), (2, lambda: mdist.ddp_two_stage_step(eng, 1)), (False, eng.step)):
    eng.set_stage2_prefetch(mode)
    a = timed(stepfn)
    eng.stage_batch(*host[0])
    def fresh():
        eng.commit_batch(); stepfn(); i[0] += 1; eng.stage_batch(*host[i[0] % 4])
    b = timed(fresh)
    eng.commit_batch()
    print("prefetch mode %s: resident %.3f  fresh batch %.3f" % (mode, a, b))
