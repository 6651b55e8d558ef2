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
    if rank == 0:
        print("DDP_OK", worst)


if __name__ == "__main__":
    main()
