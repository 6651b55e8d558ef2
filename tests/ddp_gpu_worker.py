"""Worker of tests/test_gpu_ddp.py (launched by torch.distributed.run, TWO processes on the ONE GPU of the test box, gloo
backend -- RCCL refuses two ranks on one device).  Drives the real HipEngine through mimrl_amd.dist, then the real Solver."""
import copy
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


def engine_part(world, rank, name="tiny_sep", precision="fp32", critic=None, deferred=False, split=False):
    """3 data-parallel two-stage steps of the real engine on rank-local batches: replicas bit-identical, and equal to single-process
    Adam on the mean of the two local gradients.  ``precision="bf16"`` + ``name="cfg2_sep"`` is the bench mode at the bench shape."""
    if deferred:
        os.environ["MIMRL_DDP_DEFERRED_TAIL"] = "1"
    else:
        os.environ.pop("MIMRL_DDP_DEFERRED_TAIL", None)
    if split:          # round 4: main bucket reduced in two pieces, the early one under the layer-0 BPTT (dist.ddp_stage2_split)
        os.environ["MIMRL_DDP_SPLIT"] = "1"
    else:
        os.environ.pop("MIMRL_DDP_SPLIT", None)
    c = dict(CONFIGS[name], lr=1e-4)
    if critic:
        c["critic"] = critic
    opt = make_opt(c)
    banks = synth.synthetic_banks(c["N"], seed=c["seed"])
    mk = lambda graph: HipEngine(opt, 768, 74, 35, seq_len=c["T"], bank_capacity=c["N"], precision=precision, use_graph=graph, seed=rank)
    eng = mk(True)
    p = oracle_params(opt, c["seed"])
    eng.load_params({k: v + (0.01 * rank) for k, v in p.items()})         # rank 1 starts elsewhere ...
    mdist.broadcast_(eng.main["p"]); mdist.broadcast_(eng.crit["p"])       # ... and is overwritten by rank 0's replica
    eng.params_changed()
    eng.set_batch(*local_batch(c, rank))
    eng.set_banks(*(banks[k] for k in "CFTAV"))
    eng.set_stage2_prefetch(mdist.ddp_prefetch_mode(world))               # 1: stage-2 tail beside stage 1; 2: deferred under the collective
    assert mdist.ddp_prefetch_mode(world) == (2 if deferred else 1)
    for it in range(3):
        a = local_anchors(c, rank, it)
        eng.set_anchors(1, a[0]); eng.set_anchors(2, a[1])
        mdist.ddp_two_stage_step(eng, world)
    torch.cuda.synchronize()
    flat = torch.cat([eng.main["p"], eng.crit["p"]]).reshape(1, -1)
    both = mdist.allgather_rows(flat, world)
    assert torch.equal(both[0], both[1]), "replicas diverged"
    # single-process reference: Adam on the mean of the two local gradients (two plain engines, sequential stages)
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
    # Adam's first steps are ~lr*sign(g): entries whose mean gradient is ~0 flip on summation order (2*lr each).  bf16 mode adds the
    # float atomics of the split-K weight gradients (run-to-run differences of ~1e-6 of a tensor's scale): more entries near a flip
    lim = (6.5e-4, 2e-6) if precision == "fp32" else (6.5e-4, 2e-5)
    assert d.max().item() <= lim[0] and d.mean().item() <= lim[1], (name, precision, critic, d.max().item(), d.mean().item())
    for e in (eng, A, Bq):
        e.close()
    return d.max().item()


def solver_part(world, rank):
    from mimrl_amd import Parameters
    from mimrl_amd.Solver import Solver
    opt = Parameters.parse_args(["--dataset", "synthetic", "--synthetic_n", "66", "--batch_size", "8", "--time_len", "6", "--d_hiddens",
                                 "6-3-128=4-3-128", "--d_outs", "6-3-128=4-3-128", "--bias", "--res_project", "1-1", "--dropout",
                                 "0.1-0.1-0.1-0.1", "--dropout_mlp", "0.0-0.0-0.0", "--epochs_num", "2", "--stage1_n", "1",
                                 "--loss_mi_coefficient1", "-".join(["1.0"] * 11), "--loss_mi_coefficient2", "-".join(["0.01"] * 8),
                                 "--gradient_clip", "1.5", "--learning_rate", "1e-3", "--task_name", f"ddp_rank{rank}"])
    sol = Solver(opt)
    assert sol.world == world and len(sol.train_loader) == 4, (sol.world, len(sol.train_loader))      # 66 // 2 = 33 per rank -> 4 full batches
    banks = ([], [], [], [], [])
    for ep in range(2):
        r = sol.train(ep, sol.train_loader, *banks)
        banks = r[4:]
        assert banks[1].shape[0] == world * 4 * 8, banks[1].shape                                   # all-gathered: every rank's rows
        F = banks[1].cpu().numpy()
        assert len(np.unique(np.round(F, 6), axis=0)) == F.shape[0], "duplicate bank rows: ranks are training on the same samples"
        assert np.isfinite(r[0]) and np.isfinite(r[1])
    e = sol.evaluate(sol.valid_loader, *banks)
    assert np.isfinite(e[0])
    l1, l2, mis, pred = sol.step(next(iter(sol.train_loader)))                                       # Solver.step under DDP (deferred-tail mode)
    torch.cuda.synchronize()
    flat = torch.cat([sol.engine.main["p"], sol.engine.crit["p"]]).reshape(1, -1)
    both = mdist.allgather_rows(flat, world)
    assert torch.equal(both[0], both[1]), "replicas diverged during Solver.train"
    assert torch.isfinite(flat).all() and np.isfinite(float(l1)) and np.isfinite(float(l2))


def main():
    world, rank, local = mdist.init_from_env("gloo")
    assert world == 2
    torch.cuda.set_device(0)
    worst = engine_part(world, rank)
    engine_part(world, rank, deferred=True)                                    # the round-2 schedule stays correct
    engine_part(world, rank, name="cfg2_sep", precision="bf16")                 # the bench mode at the bench shape (B = 128 per rank)
    engine_part(world, rank, name="cfg2_sep", precision="bf16", critic="concat")  # + the fused concat critic
    engine_part(world, rank, split=True)                                        # split reduce: fp32 tiny ...
    engine_part(world, rank, name="cfg2_sep", precision="bf16", split=True)     # ... and the bench mode at the bench shape
    os.environ.pop("MIMRL_DDP_DEFERRED_TAIL", None)
    os.environ.pop("MIMRL_DDP_SPLIT", None)
    solver_part(world, rank)
    dist.barrier()
    if rank == 0:
        print("DDP_GPU_OK", worst)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
