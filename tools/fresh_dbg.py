import sys, os, time, torch
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
def timed(fn, n=200):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t) / n
print("step only          %.3f" % timed(eng.step))
eng.stage_batch(*host[0])
i = [0]
def fresh():
    eng.commit_batch(); i[0] += 1; eng.stage_batch(*host[i[0] % 4]); eng.step()
print("fresh (full)       %.3f" % timed(fresh))
def switch_only():
    eng.commit_batch(); eng._staged_ev.record(eng._copy_stream); eng.step()
print("commit+step, no H2D %.3f" % timed(switch_only))
small = tuple(x[:1] for x in host[0])
def h2d_only():
    with torch.cuda.stream(eng._copy_stream):
        for dst, src in zip(eng._sets[1 - eng._active], host[0]): dst.copy_(src, non_blocking=True)
    eng.step()
print("H2D (no switch)+step %.3f" % timed(h2d_only))
def host_cost():
    t = time.perf_counter()
    for _ in range(200):
        eng.commit_batch(); eng.stage_batch(*host[0])
    return 1e3 * (time.perf_counter() - t) / 200
torch.cuda.synchronize(); print("host time of commit+stage per call %.3f ms" % host_cost()); torch.cuda.synchronize()
