"""Entry point with the reference's shape (Main.py:13-33): seed, parse flags, ``Solver(opt).solve()``."""
import faulthandler
import random

import numpy as np
import torch

from .Parameters import parse_args
from .Solver import Solver


def set_random_seed(opt):
    random.seed(opt.seed)
    np.random.seed(opt.seed)
    torch.manual_seed(opt.seed)


def main(argv=None):
    faulthandler.enable()
    opt = parse_args(argv)
    set_random_seed(opt)
    solver = Solver(opt)
    return solver.solve()


if __name__ == "__main__":
    main()
