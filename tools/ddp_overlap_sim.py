"""Diagnostic (one GPU): does a kernel on another stream -- standing in for the RCCL all-reduce of the critic bucket -- overlap the
deferred stage-2 forward tail of dist.ddp_two_stage_step, or does the tail graph queue up behind it?  Normal- against high-priority
side stream."""
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
import bench
from mimrl_amd import synth, dist as mdist
from mimrl_amd.engine import HipEngine

opt, N = bench.workload("cfg2"); B, T = opt.batch_size, opt.time_len
eng = HipEngine(opt, 768, 74, 35, seq_len=T, bank_capacity=N, precision="bf16", use_graph=True, seed=1, device_anchors=True)
eng.load_params(synth.default_state([(n, tuple(v.shape)) for n, v in eng.params.items()], 0))
eng.set_batch(*synth.synthetic_batch(B, T, seed=0)); banks = synth.synthetic_banks(N, seed=0); eng.set_banks(*(banks[k] for k in "CFTAV"))
eng.set_stage2_prefetch(2)
main = eng.stream


def timed(fn, n=150):
    for _ in range(6): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t) / n


# calibrate the spin kernel
cyc = 100000
torch.cuda.synchronize(); t = time.perf_counter(); torch.cuda._sleep(cyc * 20); torch.cuda.synchronize()
us_per_cyc = 1e6 * (time.perf_counter() - t) / (cyc * 20)
print("spin kernel: %.4f us per 1000 cycles" % (1e3 * us_per_cyc))


def sim(X, us):
    n = int(us / us_per_cyc)
    def f():
        eng.stage_grads(1)
        if X is not None:
            X.wait_stream(main)
            with torch.cuda.stream(X): torch.cuda._sleep(n)
        eng.stage2_forward_tail()
        if X is not None: main.wait_stream(X)
        eng.stage_apply(1); eng.stage_grads(2); eng.stage_apply(2)
    return timed(f)


with torch.cuda.stream(main):
    print("ddp schedule, no side kernel              %.3f" % sim(None, 0))
    for us in (100, 200, 400):
        for name, X in (("normal", torch.cuda.Stream()), ("high  ", torch.cuda.Stream(priority=-1))):
            print("side kernel %3d us on a %s-priority stream  %.3f" % (us, name, sim(X, us)))
