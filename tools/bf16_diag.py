"""Diagnostic (GPU box): per-tensor gradient error of the bf16 path vs the fp32 path."""
import numpy as np, torch
from tests.test_gpu_step import make_engine
from tests.helpers import load_golden
res = {}
import sys
MODES = ["fp32"] + sys.argv[1:]
for prec in MODES:
    c, opt, batch, banks, p, eng = make_engine("cfg1_sep", precision=prec)
    g = load_golden("cfg1_sep")
    eng.set_banks(*(banks[k] for k in "CFTAV"))
    eng.set_anchors(1, g["anchors"][0, 0]); eng.set_anchors(2, g["anchors"][0, 1])
    eng.stage_grads(1); torch.cuda.synchronize()
    g1 = {n: v.cpu().numpy().copy() for n, v in eng.grads.items() if "vmi" in n or "vcmi" in n}
    eng.stage_grads(2); torch.cuda.synchronize()
    g2 = {n: v.cpu().numpy().copy() for n, v in eng.grads.items() if not ("vmi" in n or "vcmi" in n)}
    res[prec] = (g1, g2, eng.read_scalars().copy(), eng.pred.cpu().numpy().copy(), eng.labels.cpu().numpy().copy())
    eng.close()
for mode in MODES[1:]:
  print("=========== mode", mode)
  for st in (0, 1):
    rows = []
    for n in res["fp32"][st]:
        a, b = res["fp32"][st][n], res[mode][st][n]
        rows.append((np.abs(a - b).max() / (np.abs(a).max() + 1e-12), np.abs(a).max(), n))
    va = np.concatenate([res["fp32"][st][n].ravel() for n in res["fp32"][st]]).astype(np.float64)
    vb = np.concatenate([res[mode][st][n].ravel() for n in res["fp32"][st]]).astype(np.float64)
    print(f"   whole-bucket cosine {va @ vb / np.linalg.norm(va) / np.linalg.norm(vb):.5f}  norm ratio {np.linalg.norm(vb) / np.linalg.norm(va):.4f}")
    rows.sort(reverse=True)
    print(f"--- stage {st+1}: worst 6 of {len(rows)}; median rel err {np.median([r[0] for r in rows if r[1] > 1e-6]):.3e}")
    for r in rows[:6]:
        print(f"{r[0]:9.3e} scale {r[1]:9.3e} {r[2]}")
