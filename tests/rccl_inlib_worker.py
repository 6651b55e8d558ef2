"""Worker of tests/test_gpu_ddp.py::test_rccl_inside_the_library_single_rank: ONE process, the engine's own RCCL communicator
(mimrl_set_comm) with world size 1 -- no torch.distributed process group at all."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mimrl_amd import dist as mdist, synth  # noqa: E402
from mimrl_amd.engine import HipEngine  # noqa: E402
from tests.golden.configs import CONFIGS, make_opt  # noqa: E402
from tests.helpers import oracle_params  # noqa: E402

SHAPES = {"cfg3": dict(B=256, T=500, N=16326, seed=0, critic="concat", cube="50-3-128=10-3-128")}   # cfg4's per-rank shape (BASELINE configs[3])


def run(precision, name, comm, graph=True, split=True, overlap=True, steps=3, crit_bf16=False, want_m=False):
    os.environ["MIMRL_DDP_SPLIT"] = "1" if split else "0"
    c = dict(SHAPES.get(name) or CONFIGS[name], lr=1e-4)
    opt = make_opt(c)
    banks = synth.synthetic_banks(c["N"], seed=c["seed"])
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        eng = HipEngine(opt, 768, 74, 35, seq_len=c["T"], bank_capacity=c["N"], precision=precision, use_graph=graph, seed=3)
        eng.load_params(oracle_params(opt, c["seed"]))
        eng.set_batch(*synth.synthetic_batch(c["B"], c["T"], seed=100))
        eng.set_banks(*(banks[k] for k in "CFTAV"))
    side.synchronize()
    eng.set_stage2_prefetch(1 if overlap else 0)
    if comm:
        assert mdist.attach_comm(eng, 1, 0) and mdist.has_comm(eng, 1)
        if crit_bf16:
            eng.set_comm_critic_bf16(True)
        off = eng.late_grad_ranges()
        assert len(off) == 1 and 0 < off[0][0] < off[0][1] == eng.main["g"].numel()
    g = np.random.default_rng(7)
    for it in range(steps):
        a = [np.stack([g.choice(c["N"], size=c["B"] // 2, replace=False) for _ in range(6)]) for _ in range(2)]
        eng.set_anchors(1, a[0]); eng.set_anchors(2, a[1])
        if comm:
            mdist.ddp_two_stage_step(eng, 1)          # = eng.step(): one captured graph, three collectives inside it
        else:
            eng.step()
    torch.cuda.synchronize()
    flat = torch.cat([eng.main["p"], eng.crit["p"]]).clone()
    scal = eng.read_scalars().copy()
    m = eng.crit["m"].clone() if want_m else None
    eng.close()
    return (flat, scal, m) if want_m else (flat, scal)


def bf16_distance(m):
    """After ONE Adam step from zero moments m = (1 - beta1) g: relative distance of g to the nearest bf16 value, per element."""
    g = (m.double() / (1.0 - 0.9)).float()
    g = g[g.abs() > 1e-20]
    return ((g - g.bfloat16().float()).abs() / g.abs()).double()


def main():
    cases = [("fp32", "tiny_sep", True, True, True), ("bf16", "cfg2_sep", True, True, True), ("bf16", "cfg2_sep", True, False, True),
             ("bf16", "cfg2_sep", False, True, True), ("bf16", "cfg2_sep", True, True, False), ("bf16", "cfg1_cat", True, True, True)]
    if os.environ.get("MIMRL_TEST_CFG4_SHAPE"):
        cases.append(("bf16", "cfg3", True, True, True))
    for precision, name, graph, split, overlap in cases:
        a, sa = run(precision, name, True, graph, split, overlap)
        b, sb = run(precision, name, False, graph, split, overlap)
        d = (a - b).abs()
        # a one-rank SUM is the identity: same arithmetic up to the order of float atomics (and, with the communicator, the layer-0 weight
        # gradients reach the bucket through the unpack kernel instead of Adam's fold); Adam's lr * sign(g) steps flip entries with g ~ 0
        assert torch.isfinite(a).all() and d.max().item() <= 6.5e-4 and d.mean().item() <= 2e-5, (precision, name, graph, split, overlap, d.max().item(), d.mean().item())
        assert np.allclose(sa, sb, rtol=2e-3, atol=2e-4), (precision, name, np.abs(sa - sb).max())
    if not os.environ.get("MIMRL_TEST_CFG4_SHAPE"):
        # the critic bucket as bf16 on the wire (mimrl_set_comm_critic_bf16): with one rank the "sum" is the rank's own rounded gradient, so
        # after ONE step from zero Adam moments m / (1 - beta1) must BE bf16 values (fp32 path, the control: 2^-10 away on average), and
        # three steps stay within Adam-sign-flip distance of the plain engine
        _, _, m16 = run("bf16", "cfg2_sep", True, steps=1, crit_bf16=True, want_m=True)
        _, _, m32 = run("bf16", "cfg2_sep", True, steps=1, crit_bf16=False, want_m=True)
        d16, d32 = bf16_distance(m16), bf16_distance(m32)
        assert d16.numel() > 1_000_000 and torch.quantile(d16[:4_000_000], 0.999).item() < 2e-6, torch.quantile(d16[:4_000_000], 0.999).item()
        assert d32.median().item() > 2e-4, d32.median().item()
        a, sa = run("bf16", "cfg2_sep", True, crit_bf16=True)
        b, sb = run("bf16", "cfg2_sep", False)
        d = (a - b).abs()
        assert torch.isfinite(a).all() and d.max().item() <= 6.5e-4 and d.mean().item() <= 4e-5, ("critic bf16", d.max().item(), d.mean().item())
        assert np.allclose(sa, sb, rtol=5e-3, atol=5e-4), ("critic bf16", np.abs(sa - sb).max())
    print("RCCL_INLIB_OK")


if __name__ == "__main__":
    main()
