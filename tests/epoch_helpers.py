"""Shared by the epoch-level tests: replay of the reference's anchor draws and the comparison bands."""
import numpy as np
import torch

from tests.golden.configs import EPOCH_CONFIGS, epoch_data, make_opt, split_batches
from tests.helpers import load_golden


class DrawReplay:
    """Hands out the reference's recorded ``np.random.choice`` draws (Model.py:81) six at a time, checking (N, m)."""

    def __init__(self, g):
        self.flat, self.lens, self.i, self.pos = g["draws"], g["draw_len"], 0, 0

    def __call__(self, N, m):
        out = []
        for _ in range(6):
            n = int(self.lens[self.i])
            assert n == m, f"draw {self.i}: reference drew {n} anchors, caller expects {m}"
            a = self.flat[self.pos:self.pos + n]
            assert a.max() < N
            out.append(a)
            self.i += 1
            self.pos += n
        return np.stack(out)

    def done(self):
        return self.i == len(self.lens)


def epoch_case(name):
    c = EPOCH_CONFIGS[name]
    opt = make_opt(dict(c, N=0))
    data = epoch_data(c)
    sets = {k: [tuple(torch.from_numpy(x) for x in b) for b in split_batches(v, c["B"])] for k, v in data.items()}
    return c, opt, sets, load_golden(name)


def lr_scale(c, epoch):
    """MultiStepLR (Solver.py:160-163) stepped once per epoch (Solver.py:52-57)."""
    return c["lr_rate"] ** sum(epoch >= int(m) for m in c["lr_iter"].split("-"))


# Comparison bands per epoch (scalar rtol, scalar atol, array atol).  Both fixtures train at lr = 1e-4: at the README's
# 4e-3 (and already at 1e-3) Adam's early ~lr*sign(g) updates amplify fp32 summation-order noise chaotically -- the
# reference, the fp32 oracle and the fp64 oracle then disagree by O(0.1) in the features after ONE epoch (measured), so
# nothing could be pinned.  At 1e-4 the oracle tracks the reference's own Solver to <= 3e-3 over three epochs.
# Bank entries / predictions in epoch 0 (third value): one Adam sign flip on a gradient entry that is zero to fp32 noise moves that
# parameter by 2 lr after its first update and a post-ReLU bank entry of the last batches by ~2e-3 (seen on hardware when only the
# summation order of the wave reductions changed), hence 3e-3 there rather than 1e-3.
def bands(epoch):
    return (1e-3, 2e-5, 3e-3) if epoch == 0 else (3e-3, 3e-4, 1e-2)
