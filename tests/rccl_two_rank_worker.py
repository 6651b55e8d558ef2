"""Worker of tests/test_gpu_ddp.py::test_two_real_rccl_ranks_in_graph (launched by torch.distributed.run, TWO processes on TWO GPUs):
the engine's own RCCL communicator with two ranks -- the collectives are nodes of the captured two-stage graph (dist.attach_comm), the
default transport of `Solver` and `bench.py --gpus N` -- for 3 steps on rank-local batches: replicas bit-identical, and equal to
single-process Adam on the mean of the two local gradients.  Reference counterpart: nn.DataParallel's gradient reduce, Solver.py:33-35."""
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


def local_batch(c, r):
    return synth.synthetic_batch(c["B"], c["T"], seed=100 + r)


def local_anchors(c, r, it):
    g = np.random.default_rng(1000 * it + r)
    return [np.stack([g.choice(c["N"], size=c["B"] // 2, replace=False) for _ in range(6)]) for _ in range(2)]


def part(world, rank, name, precision, graph=True, critic=None):
    c = dict(CONFIGS[name], lr=1e-4)
    if critic:
        c["critic"] = critic
    opt = make_opt(c)
    banks = synth.synthetic_banks(c["N"], seed=c["seed"])
    mk = lambda g: HipEngine(opt, 768, 74, 35, seq_len=c["T"], bank_capacity=c["N"], precision=precision, use_graph=g, seed=rank)
    eng = mk(graph)
    p = oracle_params(opt, c["seed"])
    eng.load_params({k: v + (0.01 * rank) for k, v in p.items()})          # rank 1 starts elsewhere ...
    mdist.broadcast_(eng.main["p"]); mdist.broadcast_(eng.crit["p"])        # ... and is overwritten by rank 0's replica
    eng.params_changed()
    eng.set_batch(*local_batch(c, rank))
    eng.set_banks(*(banks[k] for k in "CFTAV"))
    assert mdist.attach_comm(eng, world, rank), getattr(eng, "ddp_transport_reason", None)
    assert mdist.has_comm(eng, world) and eng.comm_world == world
    eng.set_stage2_prefetch(1)
    for it in range(3):
        a = local_anchors(c, rank, it)
        eng.set_anchors(1, a[0]); eng.set_anchors(2, a[1])
        mdist.ddp_two_stage_step(eng, world)                                 # = eng.step(): ONE captured graph, three collectives inside it
    torch.cuda.synchronize()
    flat = torch.cat([eng.main["p"], eng.crit["p"]]).reshape(1, -1)
    both = mdist.allgather_rows(flat, world)
    assert torch.isfinite(flat).all() and torch.equal(both[0], both[1]), f"replicas diverged ({name}, {precision}, graph={graph})"
    # single-process reference on this rank's GPU: Adam on the mean of the two local gradients (two plain engines, sequential stages)
    A, Bq = mk(False), mk(False)
    A.load_params(p); Bq.load_params(p)
    for e, r in ((A, 0), (Bq, 1)):
        e.set_batch(*local_batch(c, r))
        e.set_banks(*(banks[k] for k in "CFTAV"))
    for it in range(3):
        for e, r in ((A, 0), (Bq, 1)):
            a = local_anchors(c, r, it)
            e.set_anchors(1, a[0]); e.set_anchors(2, a[1])
        for stage in (1, 2):
            A.stage_grads(stage); Bq.stage_grads(stage)
            A.bucket_grad(stage).add_(Bq.bucket_grad(stage)).mul_(0.5)
            A.stage_apply(stage)
            Bq.main["p"].copy_(A.main["p"]); Bq.crit["p"].copy_(A.crit["p"]); Bq.params_changed()
            Bq.bucket_grad(stage).zero_()
    ref = torch.cat([A.main["p"], A.crit["p"]])
    d = (flat[0] - ref).abs()
    lim = (6.5e-4, 2e-6) if precision == "fp32" else (6.5e-4, 2e-5)           # (Adam's lr * sign(g) steps flip where the mean gradient is ~0)
    assert d.max().item() <= lim[0] and d.mean().item() <= lim[1], (name, precision, d.max().item(), d.mean().item())
    for e in (eng, A, Bq):
        e.close()
    return d.max().item()


def main():
    world, rank, local = mdist.init_from_env("nccl")
    assert world == 2 and dist.get_backend() == "nccl"
    torch.cuda.set_device(local)
    worst = part(world, rank, "tiny_sep", "fp32")
    part(world, rank, "tiny_sep", "fp32", graph=False)                       # the collectives on the eager path
    part(world, rank, "cfg2_sep", "bf16")                                    # the bench mode at the bench shape (B = 128 per rank)
    part(world, rank, "cfg2_sep", "bf16", critic="concat")
    dist.barrier()
    if rank == 0:
        print("RCCL_TWO_RANK_OK", worst)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
