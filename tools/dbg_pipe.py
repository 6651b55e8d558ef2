import os, sys, copy
import numpy as np, torch
sys.path.insert(0, "/root/repo")
from mimrl_amd import _lib, synth
from mimrl_amd.engine import HipEngine
from tests.helpers import case, oracle_params
name = sys.argv[1] if len(sys.argv) > 1 else "tiny_sep"
c, opt, batch, banks = case(name)
B, T = c["B"], c["T"]
opt.dropout = [0.1, 0.1, 0.1, 0.1]
bs = [tuple(torch.as_tensor(x).cuda() for x in synth.synthetic_batch(B, T, seed=40 + i)) for i in range(4)]
out = {}
for mode in ("seq", "pipe"):
    eng = HipEngine(opt, 768, 74, 35, seq_len=T, bank_capacity=c["N"], precision="bf16", use_graph=True, seed=5, device_anchors=True)
    eng.load_params(oracle_params(opt, c["seed"]))
    eng.set_banks(*(banks[k] for k in "CFTAV"))
    rec = []
    if mode == "seq":
        for b in bs:
            eng.set_batch(*b); eng.stage1_step(); torch.cuda.synchronize()
            rec.append((float(eng.scalars[_lib.S1_LOSS]), eng.feats.double().abs().sum().item(), eng.anchors[0].sum().item()))
    else:
        def on(e):
            torch.cuda.synchronize()
            rec.append((float(e.scalars[_lib.S1_LOSS]), 0.0, 0))
        eng.stage1_pass(bs, on)
    out[mode] = rec
    eng.close()
for a, b in zip(out["seq"], out["pipe"]):
    print("seq loss %.6f  pipe loss %.6f   diff %.2e" % (a[0], b[0], abs(a[0] - b[0])))
