"""Diagnostic: which dependency of the fresh-batch schedule costs the step time (variants of commit_batch / stage_batch)."""
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
import bench
from mimrl_amd import synth, _lib
from mimrl_amd.engine import HipEngine, check, _ptr

opt, N = bench.workload("cfg2"); B, T = opt.batch_size, opt.time_len
eng = HipEngine(opt, 768, 74, 35, seq_len=T, bank_capacity=N, precision="bf16", use_graph=True, seed=1, device_anchors=True)
eng.load_params(synth.default_state([(n, tuple(v.shape)) for n, v in eng.params.items()], 0))
eng.set_batch(*synth.synthetic_batch(B, T, seed=0)); banks = synth.synthetic_banks(N, seed=0); eng.set_banks(*(banks[k] for k in "CFTAV"))
eng.set_stage2_prefetch(True)
host = [tuple(torch.from_numpy(x).pin_memory() for x in synth.synthetic_batch(B, T, seed=100 + i)) for i in range(4)]
eng.stage_batch(*host[0])
if os.environ.get("COPY_PRIO"):
    eng._copy_stream = torch.cuda.Stream(eng.device, priority=int(os.environ["COPY_PRIO"]))
main, cp = eng.stream, eng._copy_stream


def timed(fn, n=150):
    for _ in range(6): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t) / n


def commit(wait_staged=True, record_free=True, switch=True):
    old = eng._active; new = 1 - old
    if record_free: eng._free_ev[old].record(main)
    if wait_staged: main.wait_event(eng._staged_ev)
    if switch:
        t, a, v, y = eng._sets[new]
        check(eng.lib.mimrl_set_inputs(eng.handle, new, _ptr(t), _ptr(a), _ptr(v), _ptr(y)))
        eng.text, eng.audio, eng.video, eng.labels = t, a, v, y
        eng._active = new


def stage(src, wait_free=True, record=True, parts=(0, 1, 2, 3)):
    idle = 1 - eng._active
    with torch.cuda.stream(cp):
        if wait_free: cp.wait_event(eng._free_ev[idle])
        for q in parts: eng._sets[idle][q].copy_(src[q], non_blocking=True)
        if record: eng._staged_ev.record(cp)


i = [0]
def run(**kw):
    ck = {k: kw[k] for k in ("wait_staged", "record_free", "switch") if k in kw}
    sk = {k: kw[k] for k in ("wait_free", "record", "parts") if k in kw}
    def f():
        commit(**ck); i[0] += 1; stage(host[i[0] % 4], **sk); eng.step()
    return timed(f)

print("step only                      %.3f" % timed(eng.step))
print("full                           %.3f" % run())
print("no wait_staged on main         %.3f" % run(wait_staged=False))
print("no wait_free on copy           %.3f" % run(wait_free=False))
print("neither wait                   %.3f" % run(wait_staged=False, wait_free=False))
print("neither wait, no switch        %.3f" % run(wait_staged=False, wait_free=False, switch=False))
print("full, text only                %.3f" % run(parts=(0,)))
print("full, audio+video+labels only  %.3f" % run(parts=(1, 2, 3)))
print("full, no copies at all         %.3f" % run(parts=()))

def lockstep(**kw):
    ck = {k: kw[k] for k in ("wait_staged", "record_free", "switch") if k in kw}
    sk = {k: kw[k] for k in ("wait_free", "record", "parts") if k in kw}
    def f():
        torch.cuda.synchronize()
        commit(**ck); i[0] += 1; stage(host[i[0] % 4], **sk); eng.step()
    return timed(f)

def step_sync():
    torch.cuda.synchronize(); eng.step()
print("host lock-step: step only      %.3f" % timed(step_sync))
print("host lock-step: full           %.3f" % lockstep())
print("host lock-step: no wait_free   %.3f" % lockstep(wait_free=False))
print("host lock-step: neither wait   %.3f" % lockstep(wait_free=False, wait_staged=False))
def late():
    torch.cuda.synchronize()
    commit(); i[0] += 1; eng.step(); time.sleep(0.0005); stage(host[i[0] % 4], wait_free=False)
print("host lock-step: H2D enqueued 0.5 ms into the step  %.3f" % timed(late))

def hostwait():
    commit(); i[0] += 1
    eng._free_ev[1 - eng._active].synchronize()
    stage(host[i[0] % 4], wait_free=False); eng.step()
print("host waits for the idle set, then copies  %.3f" % timed(hostwait))
