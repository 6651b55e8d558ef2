"""--encoders lstm at cfg2's shape (B = 128, T = 50): two-stage step time with the MFMA recurrence kernels (round 5) against the scalar
fp32 ones of round 1 (MIMRL_LSTM_SCALAR=1).  GPU box: python tools/lstm_ab.py [precision]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from mimrl_amd import synth
from mimrl_amd.engine import HipEngine

prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
opt, N = bench.workload("cfg2")
opt.encoders = "lstm"
B, T = opt.batch_size, opt.time_len
for scalar in ("0", "1", "0", "1"):
    os.environ["MIMRL_LSTM_SCALAR"] = scalar
    eng = HipEngine(opt, 768, 74, 35, seq_len=T, bank_capacity=N, precision=prec, use_graph=True, seed=1234, device_anchors=True)
    eng.load_params(synth.default_state([(n, tuple(v.shape)) for n, v in eng.params.items()], 0))
    eng.set_batch(*synth.synthetic_batch(B, T, seed=0))
    banks = synth.synthetic_banks(N, seed=0)
    eng.set_banks(*(banks[k] for k in "CFTAV"))
    eng.set_stage2_prefetch(1)
    for _ in range(10):
        eng.step()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(100):
        eng.step()
    torch.cuda.synchronize()
    print(f"{prec} lstm {'scalar fp32 kernels' if scalar == '1' else 'MFMA kernels      '}: {(time.perf_counter() - t) * 10:.3f} ms per two-stage step")
    eng.close()
