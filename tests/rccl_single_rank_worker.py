"""Worker of tests/test_gpu_ddp.py::test_rccl_call_path_single_rank: one process, backend "nccl" (RCCL), world size 1, collectives forced."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mimrl_amd import dist as mdist, synth  # noqa: E402
from mimrl_amd.engine import HipEngine  # noqa: E402
from tests.golden.configs import CONFIGS, make_opt  # noqa: E402
from tests.helpers import oracle_params  # noqa: E402


def run(precision, name, deferred, use_dist, split=False):
    if deferred:
        os.environ["MIMRL_DDP_DEFERRED_TAIL"] = "1"
    else:
        os.environ.pop("MIMRL_DDP_DEFERRED_TAIL", None)
    os.environ["MIMRL_DDP_SPLIT"] = "1" if split else "0"      # the split main-bucket reduce: async RCCL all-reduces of bucket views
    c = dict(CONFIGS[name], lr=1e-4)
    opt = make_opt(c)
    banks = synth.synthetic_banks(c["N"], seed=c["seed"])
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):                     # the engine's stream is NOT the stream the collectives are called from
        eng = HipEngine(opt, 768, 74, 35, seq_len=c["T"], bank_capacity=c["N"], precision=precision, use_graph=True, seed=3)
        eng.load_params(oracle_params(opt, c["seed"]))
        eng.set_batch(*synth.synthetic_batch(c["B"], c["T"], seed=100))
        eng.set_banks(*(banks[k] for k in "CFTAV"))
    side.synchronize()
    assert torch.cuda.current_stream() != eng.stream
    eng.set_stage2_prefetch(2 if deferred else 1)
    g = np.random.default_rng(7)
    for it in range(3):
        a = [np.stack([g.choice(c["N"], size=c["B"] // 2, replace=False) for _ in range(6)]) for _ in range(2)]
        eng.set_anchors(1, a[0]); eng.set_anchors(2, a[1])
        if use_dist:
            mdist.ddp_two_stage_step(eng, 1)
        else:
            with torch.cuda.stream(eng.stream):
                eng.stage_grads(1); eng.stage2_forward_tail(); eng.stage_apply(1); eng.stage_grads(2); eng.stage_apply(2)
    torch.cuda.synchronize()
    flat = torch.cat([eng.main["p"], eng.crit["p"]]).clone()
    eng.close()
    return flat


def main():
    dist.init_process_group(backend="nccl", rank=0, world_size=1)
    assert mdist._collectives_on(1)
    for precision, name in (("fp32", "tiny_sep"), ("bf16", "cfg2_sep")):
        for deferred in (False, True):
            a = run(precision, name, deferred, True)
            b = run(precision, name, deferred, False)
            d = (a - b).abs()
            # same arithmetic; float atomics reorder additions (Adam's lr * sign(g) steps flip entries with g ~ 0: 2 * lr each)
            assert torch.isfinite(a).all() and d.max().item() <= 6.5e-4 and d.mean().item() <= 2e-5, (precision, deferred, d.max().item(), d.mean().item())
        a = run(precision, name, False, True, split=True)      # early piece in flight (async work handles on views) under stage_grads_part(2, 1)
        d = (a - b).abs()
        assert torch.isfinite(a).all() and d.max().item() <= 6.5e-4 and d.mean().item() <= 2e-5, (precision, "split", d.max().item(), d.mean().item())
    t = torch.ones(1 << 20, device="cuda")
    w = dist.all_reduce(t, async_op=True)
    w.wait()
    torch.cuda.synchronize()
    assert float(t.sum()) == float(1 << 20)
    dist.destroy_process_group()
    print("RCCL_SINGLE_RANK_OK")


if __name__ == "__main__":
    main()
