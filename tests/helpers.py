"""Shared helpers for the parity tests (oracle side)."""
import os

import numpy as np
import torch

from mimrl_amd import layout, synth
from tests.golden.configs import CONFIGS, make_opt, quantize_labels

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    return np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)


def oracle_params(opt, seed, dtype=torch.float32):
    shapes = layout.named_shapes(opt, 768, 74, 35)
    return {n: torch.from_numpy(synth.portable_tensor(n, s, seed)).to(dtype) for n, s in shapes}


def case(name, dtype=torch.float32):
    c = CONFIGS[name]
    opt = make_opt(c)
    t, a, v, y = synth.synthetic_batch(c["B"], c["T"], seed=c["seed"], ragged=c.get("ragged", False))
    banks = synth.synthetic_banks(c["N"], seed=c["seed"])
    (t, a, v, y), banks = quantize_labels(c, (t, a, v, y), banks)
    batch = tuple(torch.from_numpy(x).to(dtype) for x in (t, a, v, y))
    banks_t = {k: torch.from_numpy(val).to(dtype) for k, val in banks.items()}
    return c, opt, batch, banks_t


def rel_close(got, want, rtol=1e-3, atol=1e-5):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    return np.all(np.abs(got - want) <= atol + rtol * np.abs(want))
