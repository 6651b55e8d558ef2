"""Worker of tests/test_gpu_step.py::test_every_registered_knob_vs_oracle: ONE process with ONE environment knob set (the library reads most
knobs once per process), the benchmarked bf16 mode on two fixtures (`tiny_sep`: separable critics, T = 6; `cfg1_cat`: BASELINE configs[0]'s
shape with the concat critic).  For both stages: stage loss and MI / CMI values against the oracle's autograd (oracle/mimrl_ref.py, pinned to
the reference by tests/golden), gradient buckets by cosine against the oracle, every gradient tensor dumped for the parent (which compares
it with the default-knob run), then two captured-graph steps in Solver.step() overlap mode that must stay finite."""
import json
import os
import sys

os.environ.setdefault("OMP_NUM_THREADS", "8")      # (the oracle is ~55 k small CPU ops: on a 256-core host all-core threading makes each op a barrier over idle threads)
import numpy as np
import torch

torch.set_num_threads(8)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mimrl_amd import _lib  # noqa: E402
from mimrl_amd.engine import HipEngine  # noqa: E402
from oracle import mimrl_ref as R  # noqa: E402
from tests.gpu_helpers import assert_close, oracle_raw_grads  # noqa: E402
from tests.helpers import case, load_golden, oracle_params  # noqa: E402


def run(name, out):
    c, opt, batch, banks = case(name)
    g = load_golden(name)
    anchors = g["anchors"][0]
    p = oracle_params(opt, c["seed"])
    crit = [n for n in p if R.is_critic_param(n)]
    main = [n for n in p if not R.is_critic_param(n)]
    eng = HipEngine(opt, 768, 74, 35, seq_len=c["T"], bank_capacity=c["N"], precision="bf16", use_graph=True)
    eng.load_params(p)
    eng.set_batch(*batch)
    eng.set_banks(*(banks[k] for k in "CFTAV"))
    res = {}
    for stage, names in ((1, crit), (2, main)):
        eng.set_anchors(stage, anchors[stage - 1])
        eng.stage_grads(stage)
        torch.cuda.synchronize()
        s = eng.read_scalars()
        loss, mis, pred, feats, task, grads = oracle_raw_grads(p, opt, stage, batch, banks, anchors[stage - 1], names)
        if stage == 1:
            assert_close(s[_lib.S1_LOSS], loss.item(), 2e-2, 2e-2, f"{name}: stage-1 loss")
            assert_close(s[_lib.S1_MIS:_lib.S1_MIS + 11], [m.item() for m in mis], 5e-2, 5e-2, f"{name}: stage-1 MI/CMI")
        else:
            assert_close(s[_lib.S2_LOSS], loss.item(), 2e-2, 2e-2, f"{name}: stage-2 loss")
            assert_close(s[_lib.S2_TASK], task.item(), 2e-2, 2e-3, f"{name}: task loss")
            assert_close(s[_lib.S2_MIS:_lib.S2_MIS + 8], [m.item() for m in mis], 5e-2, 5e-2, f"{name}: stage-2 MI terms")
        got = torch.cat([eng.grads[n].double().reshape(-1).cpu() for n in names])
        want = torch.cat([grads[n].double().reshape(-1) for n in names])
        cos = float((got @ want) / (got.norm() * want.norm() + 1e-300))
        res[f"cos_s{stage}"] = cos
        assert torch.isfinite(got).all() and cos >= 0.9, f"{name}: stage-{stage} bucket cosine vs oracle {cos}"     # (the parent compares with the default run's)
        top = max(float(grads[n].double().norm()) / np.sqrt(max(grads[n].numel(), 1)) for n in names) + 1e-30
        for n in names:
            g_, w_ = eng.grads[n].double().cpu(), grads[n].double()
            den = max(float(w_.norm()), 5e-2 * top * np.sqrt(w_.numel()))    # (tensors below 5 % of the bucket's largest RMS -- the near-trivial critics' biases at the initial point -- are judged at that scale)
            out[f"err|{name}|s{stage}|{n}"] = np.float64(float((g_ - w_).norm()) / den)
            out[f"{name}|s{stage}|{n}"] = g_.float().numpy().copy()
        res[f"loss_s{stage}"] = float(s[_lib.S1_LOSS if stage == 1 else _lib.S2_LOSS])
    # (no stage_apply in between: stage 2 is evaluated at the SAME critics in every knob run -- no Adam sign-flip noise between runs; the
    #  buckets are dirty now, so the first step() below takes the per-stage path and the second one the combined captured graph)
    eng.set_stage2_prefetch(1)
    eng.set_anchors(1, anchors[0]); eng.set_anchors(2, anchors[1])
    for _ in range(2):
        eng.step()
    torch.cuda.synchronize()
    assert np.isfinite(eng.read_scalars()).all() and torch.isfinite(eng.main["p"]).all() and torch.isfinite(eng.crit["p"]).all(), f"{name}: graph steps"
    eng.close()
    return res


def main():
    out = {}
    res = {name: run(name, out) for name in ("tiny_sep", "cfg1_cat")}
    np.savez(sys.argv[1], **out)
    print("KNOB_OK " + json.dumps(res))


if __name__ == "__main__":
    main()
