import faulthandler, sys, os, time
faulthandler.enable()
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import bench
from mimrl_amd import synth
from mimrl_amd.engine import HipEngine
opt, N = bench.workload("cfg2")
B, T = opt.batch_size, opt.time_len
for rep in range(int(sys.argv[1])):
    eng = HipEngine(opt, 768, 74, 35, seq_len=T, bank_capacity=N, precision="bf16", use_graph=True, seed=1, device_anchors=True)
    eng.load_params(synth.default_state([(n, tuple(v.shape)) for n, v in eng.params.items()], 0))
    eng.set_batch(*synth.synthetic_batch(B, T, seed=0))
    banks = synth.synthetic_banks(N, seed=0); eng.set_banks(*(banks[k] for k in "CFTAV"))
    eng.set_stage2_prefetch(1)
    for _ in range(20): eng.step()
    torch.cuda.synchronize()
    host = [tuple(torch.from_numpy(x).pin_memory() for x in synth.synthetic_batch(B, T, seed=100 + i)) for i in range(2)]
    for cyc in range(6):
        eng.set_stage2_prefetch(False)
        for _ in range(8): eng.step()
        torch.cuda.synchronize()
        eng.set_stage2_prefetch(True)
        eng.stage_batch(*host[0])
        for i in range(8):
            eng.commit_batch(); eng.stage_batch(*host[i % 2]); eng.step()
        torch.cuda.synchronize()
        eng.profile(True)
        for _ in range(2): eng.stage1_step(); eng.stage2_step()
        eng.profile_read(); eng.profile_read_gemm(); eng.profile(False)
    eng.close()
    print("rep", rep, "ok", flush=True)
