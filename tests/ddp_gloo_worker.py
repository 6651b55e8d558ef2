"""Worker of tests/test_cli_and_host.py::test_ddp_two_ranks_gloo (launched by torch.distributed.run, gloo, CPU).
An oracle-backed stand-in with the HipEngine's data-parallel surface drives mimrl_amd.dist.ddp_stage_step."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mimrl_amd import dist as mdist, synth  # noqa: E402
from oracle import mimrl_ref as R  # noqa: E402
from tests.golden.configs import CONFIGS, make_opt  # noqa: E402
from tests.helpers import oracle_params  # noqa: E402


class OracleEngine:
    """stage_grads / bucket_grad / stage_apply on flat buckets, arithmetic by the CPU oracle."""

    def __init__(self, opt, params, batch, banks, anchors):
        self.opt, self.p, self.batch, self.banks, self.anchors = opt, params, batch, banks, anchors
        self.names = {1: [n for n in params if R.is_critic_param(n)], 2: [n for n in params if not R.is_critic_param(n)]}
        self.adam = {s: R.AdamState(params, self.names[s]) for s in (1, 2)}
        self.flat = {s: torch.zeros(sum(params[n].numel() for n in self.names[s])) for s in (1, 2)}
        self.bank_rows = len(banks["C"])

    def has_update(self, stage):
        return stage == 2 or self.bank_rows > 0

    def set_grad_scale(self, scale):
        self.gscale = float(scale)

    def bucket_grad(self, stage):
        return self.flat[stage]

    def stage_grads(self, stage):
        leaves = {k: v.detach().clone().requires_grad_(True) for k, v in self.p.items()}
        loss, *_ = R.stage_loss(leaves, self.opt, stage, self.batch, self.banks, self.anchors[stage - 1])
        gs = torch.autograd.grad(loss, [leaves[n] for n in self.names[stage]], allow_unused=True)
        self.flat[stage] = torch.cat([(g if g is not None else torch.zeros_like(leaves[n])).reshape(-1)
                                      for n, g in zip(self.names[stage], gs)])
        self.loss = loss.detach()

    # the split stage-2 pass of the data-parallel step (HipEngine.stage_grads_part / late_grad_ranges): part 0 leaves every range but the
    # layer-0 GRU tensors final, part 1 fills those in
    def late_grad_ranges(self):
        out, o = [], 0
        for n in self.names[2]:
            k = self.p[n].numel()
            if n.startswith("rnn_") and "_l0" in n:
                if out and out[-1][1] == o:
                    out[-1] = (out[-1][0], o + k)
                else:
                    out.append((o, o + k))
            o += k
        return out

    def stage_grads_part(self, stage, part):
        assert stage == 2
        if part == 0:
            self.stage_grads(2)
            self._full = self.flat[2].clone()
            for a, b in self.late_grad_ranges():
                self.flat[2][a:b] = float("nan")              # not final yet: a reduce of these before part 1 would poison the result
        else:
            for a, b in self.late_grad_ranges():
                self.flat[2][a:b] = self._full[a:b]

    def stage_apply(self, stage):
        clip = float(self.opt.gradient_clip)
        grads, o = {}, 0
        for n in self.names[stage]:
            k = self.p[n].numel()
            grads[n] = (self.flat[stage][o:o + k] * getattr(self, "gscale", 1.0)).reshape(self.p[n].shape).clamp(-clip, clip)
            o += k
        lr = float(self.opt.learning_rate) * (float(self.opt.mi_lr_rate) if stage == 1 else 1.0)
        self.adam[stage].step(self.p, grads, lr, float(self.opt.weight_decay))


def main():
    world, rank, _ = mdist.init_from_env("gloo")
    assert world == 2
    torch.set_num_threads(2)
    c = CONFIGS["tiny_sep"]
    opt = make_opt(c)
    banks = {k: torch.from_numpy(v) for k, v in synth.synthetic_banks(c["N"], seed=c["seed"]).items()}
    rng = np.random.default_rng(5)
    anchors = [np.stack([rng.choice(c["N"], size=c["B"] // 2, replace=False) for _ in range(6)]) for _ in range(2)]

    def local_batch(r):
        return tuple(torch.from_numpy(x) for x in synth.synthetic_batch(c["B"], c["T"], seed=100 + r))

    eng = OracleEngine(opt, oracle_params(opt, c["seed"]), local_batch(rank), banks, anchors)
    for stage in (1, 2):
        mdist.ddp_stage_step(eng, stage, world)
    # single-process reference: Adam on the rank-mean of the two local gradients
    ref = OracleEngine(opt, oracle_params(opt, c["seed"]), local_batch(0), banks, anchors)
    other = OracleEngine(opt, ref.p, local_batch(1), banks, anchors)
    for stage in (1, 2):
        ref.stage_grads(stage)
        other.p = ref.p
        other.stage_grads(stage)
        ref.flat[stage] = 0.5 * (ref.flat[stage] + other.flat[stage])
        ref.stage_apply(stage)
    worst = max((eng.p[n] - ref.p[n]).abs().max().item() for n in eng.p)
    assert worst < 1e-5, f"rank {rank}: params differ from single-process mean-gradient Adam by {worst}"
    # replicas stay identical
    flat = torch.cat([eng.p[n].reshape(-1) for n in sorted(eng.p)])
    both = mdist.allgather_rows(flat.reshape(1, -1), world)
    assert torch.equal(both[0], both[1]), "replicas diverged"
    # round 4: Solver.step's data-parallel schedule with the SPLIT main-bucket reduce (dist.ddp_stage2_split): same result
    os.environ["MIMRL_DDP_SPLIT"] = "1"
    eng2 = OracleEngine(opt, oracle_params(opt, c["seed"]), local_batch(rank), banks, anchors)
    assert len(eng2.late_grad_ranges()) == 2
    mdist.ddp_two_stage_step(eng2, world)
    os.environ.pop("MIMRL_DDP_SPLIT")
    worst2 = max((eng2.p[n] - ref.p[n]).abs().max().item() for n in eng2.p)
    assert worst2 < 1e-5 and all(torch.isfinite(v).all() for v in eng2.p.values()), f"rank {rank}: split reduce differs by {worst2}"
    if rank == 0:
        print("DDP_OK", worst, worst2)


if __name__ == "__main__":
    main()
