"""Diagnostic: forward features / prediction of one golden case (fp32) against the reference-generated golden."""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from tests.helpers import load_golden                   # noqa: E402
from tests.test_gpu_step import make_engine             # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "cfg1_cat"
c, opt, batch, banks, p, eng = make_engine(name)
g = load_golden(name)
eng.forward(train=False)
torch.cuda.synchronize()
feats = eng.feats.cpu().numpy().astype(np.float64)
print("pred max|err|", np.abs(eng.pred.cpu().numpy() - g["fwd_pred"].reshape(-1)).max())
for i, k in enumerate(["F_F", "T_F", "A_F", "V_F"]):
    w = g["fwd_" + k].astype(np.float64)
    print(k, "max|err| %.3e  max|want| %.3e" % (np.abs(feats[i] - w).max(), np.abs(w).max()))
eng.close()
