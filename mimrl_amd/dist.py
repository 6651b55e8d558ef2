"""Data-parallel replicas: one process per GPU, RCCL (torch.distributed backend "nccl") over xGMI.

The reference only has single-process nn.DataParallel (Solver.py:33-35).  Here every rank owns a full replica and a
local batch; the estimators are local to the rank (B_local x B_local InfoNCE), so the only exchange
per stage is ONE all-reduce(mean) of that stage's flat gradient bucket (critics: 3.36 M floats, main: 1.08 M floats),
followed by the fused clip+Adam -- value-clipping the averaged gradient, as a single process would.
The helpers are backend-agnostic so the world_size-2 gloo tests can drive them on CPU.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def env_world():
    return int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))


def init_from_env(backend: str = None):
    """Initialise torch.distributed from torchrun's environment; returns (world, rank, local_rank)."""
    world, rank, local = env_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # this pool's host driver only has dmabuf IPC (RCCL's xGMI transport needs it)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:   # MIMRL_DIST_BACKEND=gloo: testing only (two ranks on ONE device, which RCCL refuses)
            backend = os.environ.get("MIMRL_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
        elif torch.cuda.is_available():
            local = local % torch.cuda.device_count()
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return world, rank, local


def allreduce_sum_(flat: torch.Tensor, world: int, async_op: bool = False):
    """In-place SUM over ranks of one flat bucket (a single collective; RCCL picks ring/tree/direct over xGMI).  The 1/world
    factor of the mean is folded into the engine's fused clip+Adam (``set_grad_scale``): no scaling pass over the bucket."""
    if world <= 1:
        return None
    return dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=async_op)


def broadcast_(flat: torch.Tensor, src: int = 0):
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(flat, src=src)
    return flat


def prepare_engine(engine, world: int):
    """Once per engine: fold the mean's 1/world into Adam."""
    if getattr(engine, "_ddp_world", None) != world:
        engine.set_grad_scale(1.0 / world)
        engine._ddp_world = world


def ddp_stage_step(engine, stage: int, world: int):
    """grads -> all-reduce(sum) of the stage's bucket -> clip+Adam on bucket/world.  ``engine`` needs stage_grads /
    stage_apply / bucket_grad(stage) / set_grad_scale.  The collective is enqueued behind the kernels in stream order: no host
    synchronisation (RCCL); value-clipping happens after averaging, as in a single process."""
    prepare_engine(engine, world)
    engine.stage_grads(stage)
    if world > 1 and engine.has_update(stage):
        allreduce_sum_(engine.bucket_grad(stage), world)
    engine.stage_apply(stage)


def ddp_two_stage_step(engine, world: int):
    """Solver.step() under data parallelism, with the critic-bucket collective hidden:

        stage_grads(1)            shared encoder prefix, stage-1 tail, 11 estimators fwd+bwd  -> crit_g complete
        all-reduce(crit_g) ASYNC  on RCCL's communication stream (13.4 MB, the larger bucket)
        stage2_forward_tail()     LN+ReLU+dropout, CubeMLP, head of the stage-2 pass: needs neither crit_g nor the critic
                                  update, so it runs UNDER the collective (engine in deferred-tail mode)
        wait; stage_apply(1)      clip + Adam on crit_g / world  ->  stage 2's estimators see the updated critics
        stage_grads(2); all-reduce(main_g); stage_apply(2)

    The main bucket (4.3 MB) is reduced in one piece behind the backward pass: its last producers (GRU layer-0 and W_t
    weight gradients) finish with the stage, and CubeMLP's weight gradients are deliberately issued late (beside the BPTT),
    so there is no early-ready prefix of that bucket to reduce ahead of time.
    Requires ``engine.set_stage2_prefetch(2)`` (Solver.step does it)."""
    prepare_engine(engine, world)
    engine.stage_grads(1)
    work = allreduce_sum_(engine.bucket_grad(1), world, async_op=True) if (world > 1 and engine.has_update(1)) else None
    engine.stage2_forward_tail()
    if work is not None:
        work.wait()
    engine.stage_apply(1)
    engine.stage_grads(2)
    if world > 1:
        allreduce_sum_(engine.bucket_grad(2), world)
    engine.stage_apply(2)


def allgather_rows(x: torch.Tensor, world: int) -> torch.Tensor:
    """Concatenate per-rank bank rows (once per epoch, off the hot path)."""
    if world <= 1:
        return x
    outs = [torch.empty_like(x) for _ in range(world)]
    dist.all_gather(outs, x.contiguous())
    return torch.cat(outs, 0)
