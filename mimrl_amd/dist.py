"""Data-parallel replicas: one process per GPU, RCCL over xGMI.

The reference only has single-process nn.DataParallel (Solver.py:33-35).  Here every rank owns a full replica and a
local batch; the estimators are local to the rank (B_local x B_local InfoNCE), so the only exchange
per stage is the all-reduce(mean) of that stage's flat gradient bucket (critics: 3.36 M floats, main: 1.08 M floats),
followed by the fused clip+Adam -- value-clipping the averaged gradient, as a single process would.

Two transports (round 5):
  * IN THE LIBRARY (default on GPUs): ``attach_comm`` gives the engine its own RCCL communicator (mimrl_set_comm); the collectives are
    then nodes of the engine's captured step graph -- world > 1 runs the SAME single graph per step as world = 1 (cross-stage overlap and
    in-graph Adam included), with three collectives in it: critic bucket, main bucket [0, late) under the layer-0 BPTT, its layer-0 tail.
    torch.distributed only carries the 128-byte unique id to the ranks.
  * torch.distributed collectives between per-stage graph launches (rounds 1-4; MIMRL_DDP_TORCH=1, and whenever the process group is not
    RCCL: the world_size-2 gloo tests on CPU / on one GPU, which RCCL refuses).
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


import contextlib


def _engine_stream(engine):
    """Context that makes the ENGINE's stream torch's current stream.  torch.distributed orders a collective against the current
    stream at call time (the RCCL communication stream waits for it, and wait() makes it wait for the collective), while the engine
    launches on the stream it captured at construction: if the caller's current stream were another one, the all-reduce could read
    a gradient bucket before stage_grads has produced it and stage_apply could run before the reduce."""
    st = getattr(engine, "stream", None)
    if st is None or not torch.cuda.is_available():
        return contextlib.nullcontext()
    return torch.cuda.stream(st)


def _collectives_on(world: int) -> bool:
    # MIMRL_DDP_FORCE_COLLECTIVES=1: issue the collectives at world == 1 too (a one-rank communicator) -- how the RCCL call path,
    # its stream ordering against the engine's stream and the async work handle are exercised on a ONE-GPU box (tests/test_gpu_ddp.py)
    return world > 1 or (dist.is_initialized() and os.environ.get("MIMRL_DDP_FORCE_COLLECTIVES") is not None)


def allreduce_sum_(flat: torch.Tensor, world: int, async_op: bool = False):
    """In-place SUM over ranks of one flat bucket (a single collective; RCCL picks ring/tree/direct over xGMI).  The 1/world
    factor of the mean is folded into the engine's fused clip+Adam (``set_grad_scale``): no scaling pass over the bucket."""
    if not _collectives_on(world):
        return None
    return dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=async_op)


def broadcast_(flat: torch.Tensor, src: int = 0):
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(flat, src=src)
    return flat


def torch_transport_forced() -> bool:
    """MIMRL_DDP_TORCH=1 (any value but '' / '0'): keep the torch.distributed transport even where the in-library one is available."""
    return os.environ.get("MIMRL_DDP_TORCH", "0") not in ("", "0")


def attach_comm(engine, world: int, rank: int) -> bool:
    """Create the engine's own RCCL communicator (collective).  Rank 0 draws the unique id, torch.distributed broadcasts its 128 bytes.
    Returns False (and leaves the torch.distributed transport in charge) when MIMRL_DDP_TORCH=1, on CPU, when the process group is
    not RCCL, or when ANY rank failed to create its communicator: every rank reports success / failure through a MIN all-reduce and all of
    them take the same transport (a rank that raised alone would leave the others inside ncclCommInitRank or, later, in a collective
    its peers never issue -- ADVICE r05).  world == 1: a one-rank communicator without any process group (MIMRL_DDP_FORCE_COLLECTIVES=1:
    the one-GPU tests / bench); there a failure raises, nobody else is waiting.
    ``engine.ddp_transport_reason`` says why the answer was False."""
    engine.ddp_transport_reason = None
    if torch_transport_forced() or not torch.cuda.is_available() or not hasattr(engine, "set_comm"):
        engine.ddp_transport_reason = "MIMRL_DDP_TORCH set" if torch_transport_forced() else "no GPU / engine without set_comm"
        return False
    if world > 1:
        if not dist.is_initialized() or dist.get_backend() != "nccl":
            engine.ddp_transport_reason = "process group is not RCCL"
            return False
        err = None
        payload = torch.zeros(128, dtype=torch.uint8, device=engine.device)
        try:
            if rank == 0:
                payload.copy_(torch.frombuffer(bytearray(type(engine).comm_unique_id()), dtype=torch.uint8))
        except Exception as e:      # noqa: BLE001 -- librccl missing on rank 0: the others learn it from the all-zero id + the MIN reduce below
            err = e
        dist.broadcast(payload, src=0)
        uid = bytes(payload.cpu().numpy().tobytes())
        if err is None and any(uid):
            try:
                engine.set_comm(uid, world, rank)
            except Exception as e:  # noqa: BLE001
                err = e
        elif err is None:
            err = RuntimeError("rank 0 could not draw an RCCL unique id")
        flag = torch.tensor([0 if err is not None else 1], device=engine.device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            if err is None:         # this rank has a communicator its peers do not: drop it
                engine.set_comm(None, 1, 0)
            engine.ddp_transport_reason = f"in-library communicator failed on {'this' if err is not None else 'another'} rank: {err!r}"[:300]
            flush_c_stdio()
            return False
    else:
        engine.set_comm(type(engine).comm_unique_id(), world, rank)
    if os.environ.get("MIMRL_DDP_BF16_CRITIC", "0") not in ("", "0"):      # opt-in: the 13.4 MB critic bucket crosses the links as bf16
        engine.set_comm_critic_bf16(True)
    engine._ddp_world = world
    flush_c_stdio()
    return True


def flush_c_stdio():
    """RCCL prints a version banner through C stdio when a communicator is created; on a pipe that buffer is written at exit, i.e. BEHIND
    whatever Python printed meanwhile (bench.py's one JSON line must be the last thing on stdout)."""
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:   # noqa: BLE001
        pass


def has_comm(engine, world: int) -> bool:
    return getattr(engine, "comm_world", 0) == world and world >= 1


def prepare_engine(engine, world: int):
    """Once per engine: fold the mean's 1/world into Adam."""
    if getattr(engine, "_ddp_world", None) != world:
        engine.set_grad_scale(1.0 / world)
        engine._ddp_world = world


def ddp_stage_step(engine, stage: int, world: int):
    """grads -> all-reduce(sum) of the stage's bucket -> clip+Adam on bucket/world.  ``engine`` needs stage_grads /
    stage_apply / bucket_grad(stage) / set_grad_scale.  The collective is enqueued behind the kernels in stream order: no host
    synchronisation (RCCL); value-clipping happens after averaging, as in a single process."""
    if has_comm(engine, world):          # the library reduces inside the stage's own enqueue
        (engine.stage1_step if stage == 1 else engine.stage2_step)()
        return
    prepare_engine(engine, world)
    with _engine_stream(engine):
        engine.stage_grads(stage)
        if _collectives_on(world) and engine.has_update(stage):
            allreduce_sum_(engine.bucket_grad(stage), world)
        engine.stage_apply(stage)


def deferred_tail() -> bool:
    """MIMRL_DDP_DEFERRED_TAIL=1: the round-2 schedule (stage-2 forward tail issued UNDER the critic-bucket collective, engine
    prefetch mode 2) instead of the default (tail beside stage 1, mode 1)."""
    return os.environ.get("MIMRL_DDP_DEFERRED_TAIL") is not None


def ddp_prefetch_mode(world: int) -> int:
    """Engine overlap mode Solver.step / bench.py select for a data-parallel two-stage step."""
    return 2 if (world > 1 and deferred_tail()) else 1


def ddp_two_stage_step(engine, world: int):
    """Solver.step() under data parallelism (reference counterpart: nn.DataParallel's gradient reduce, Solver.py:33-35):

        stage_grads(1)            shared encoder prefix, BOTH forward tails (stage 1's, and stage 2's beside it: prefetch mode 1),
                                  11 estimators fwd+bwd                                             -> crit_g complete
        all-reduce(crit_g)        RCCL, 13.4 MB
        stage_apply(1)            clip + Adam on crit_g / world  ->  stage 2's estimators see the updated critics
        stage_grads(2); all-reduce(main_g) (4.3 MB); stage_apply(2)

    Round 3: the stage-2 forward tail runs BESIDE stage 1's estimators (where the single-GPU graph hides it too), not under the first
    collective as in round 2 ("deferred tail", still available: MIMRL_DDP_DEFERRED_TAIL=1 + prefetch mode 2).  Arithmetic: the tail
    (0.15 ms) is hidden either way, but beside stage 1 it is hidden even when the collective is shorter than the tail, and the
    per-rank launch structure measured WITHOUT communication on one MI355X is 1.19 ms against 1.26 ms (single-GPU graph: 1.01 ms;
    bench.py: ms_per_step_ddp_schedule_no_comm).  The dependency all-reduce -> Adam_vmi -> stage-2 critic forward is a true one
    (SURVEY.md section 5); nothing of stage 2 that is independent of the updated critics is left to put under it.  The main bucket is
    reduced in one piece: 90 % of it are GRU / W_t gradients that become final with the last kernels of the stage.
    NOT MEASURED ON MORE THAN ONE GPU (no multi-GPU box available to the builder): see DESIGN.md section 6.

    Round 5: with the engine's own communicator (``attach_comm``) none of the above is driven from here -- ``engine.step()`` is the one
    captured graph of the single-GPU step with the three collectives inside it."""
    if has_comm(engine, world):
        engine.step()
        return
    prepare_engine(engine, world)
    on = _collectives_on(world)
    with _engine_stream(engine):
        engine.stage_grads(1)
        if deferred_tail():
            work = allreduce_sum_(engine.bucket_grad(1), world, async_op=True) if (on and engine.has_update(1)) else None
            engine.stage2_forward_tail()
            if work is not None:
                work.wait()
        elif on and engine.has_update(1):
            allreduce_sum_(engine.bucket_grad(1), world)
        engine.stage_apply(1)
        if split_reduce(world) and hasattr(engine, "stage_grads_part") and getattr(getattr(engine, "cfg", None), "encoder", 0) == 0:
            ddp_stage2_split(engine, world)
            return
        engine.stage_grads(2)
        if on:
            allreduce_sum_(engine.bucket_grad(2), world)
        engine.stage_apply(2)


def split_reduce(world: int = 2) -> bool:
    """Reduce the main bucket in two pieces, the first one UNDER the rest of the stage-2 backward pass (ddp_stage2_split): the default at
    world > 1 since round 4 (the split costs a rank 12 us per step without communication -- bench.py:
    ms_per_step_ddp_split_schedule_no_comm 1.026 vs 1.013 ms -- and takes 3.2 of the bucket's 4.3 MB off the exposed path).
    MIMRL_DDP_SPLIT=0 / 1 forces it off / on (1 also at world == 1, for the one-GPU tests and the bench extra)."""
    v = os.environ.get("MIMRL_DDP_SPLIT")
    if v is not None:
        return v != "0"
    return world > 1


def _complement(ranges, n):
    out, at = [], 0
    for a, b in sorted(ranges):
        if a > at:
            out.append((at, a))
        at = max(at, b)
    if at < n:
        out.append((at, n))
    return out


def ddp_stage2_split(engine, world: int):
    """Stage 2 with the gradient all-reduce overlapped with the backward pass (north_star; reference counterpart: nn.DataParallel's
    reduce AFTER backward, Solver.py:33-35):

        stage_grads_part(2, 0)   forward tail / estimators / CubeMLP + head backward, layer-1 BPTT and its weight gradients, W_t gradient
        all-reduce(EARLY) async  every main-bucket range except the layer-0 GRU tensors: 0.80 M of 1.08 M floats (3.2 of 4.3 MB), on the
                                 communication stream, ordered behind part 0
        stage_grads_part(2, 1)   layer-0 BPTT + its weight gradients (~0.1 ms at cfg2) -- the early piece travels under it
        all-reduce(LATE)         the two layer-0 ranges (rnn_v.*_l0*, rnn_a.*_l0*: 1.1 MB)
        wait(EARLY); stage_apply(2)

    What it costs a rank: one more graph launch, and the layer-1 weight-gradient GEMMs no longer run beside the layer-0 BPTT (they must be
    final before the early piece leaves) -- bench.py reports the no-communication step time of this schedule next to the unsplit one
    (``ms_per_step_ddp_split_schedule_no_comm``: +12 us at cfg2).  The communication side is NOT measured (one-GPU boxes only); RCCL's
    call path for it -- asynchronous all-reduces of bucket views under a running graph -- is exercised with a one-rank communicator
    (tests/rccl_single_rank_worker.py), the arithmetic with two gloo ranks on one GPU: replicas bit-identical, equal to mean-gradient
    Adam (tests/ddp_gpu_worker.py), and on CPU with a NaN-poisoned late range (tests/ddp_gloo_worker.py)."""
    on = _collectives_on(world)
    engine.stage_grads_part(2, 0)
    g = engine.bucket_grad(2)
    late = engine.late_grad_ranges()
    early = _complement(late, g.numel())
    works = [allreduce_sum_(g[a:b], world, async_op=True) for a, b in early] if on else []
    engine.stage_grads_part(2, 1)
    if on:
        for a, b in late:
            allreduce_sum_(g[a:b], world)
        for w in works:
            if w is not None:
                w.wait()
    engine.stage_apply(2)


def allgather_rows(x: torch.Tensor, world: int) -> torch.Tensor:
    """Concatenate per-rank bank rows (once per epoch, off the hot path)."""
    if world <= 1:
        return x
    outs = [torch.empty_like(x) for _ in range(world)]
    dist.all_gather(outs, x.contiguous())
    return torch.cat(outs, 0)
