"""Diagnostic: wall time of Solver.train epochs (the reference's epoch-ordered schedule: a stage-1 pass over the loader, then a stage-2
pass -- Solver.py:194-247) at cfg2 on synthetic MOSI-sized data, dataset resident in HBM (default) against --host_data (pinned host
memory, one H2D per batch).  usage: python tools/epoch_time.py [n_samples] [epochs]"""
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from mimrl_amd import Parameters
from mimrl_amd.Solver import Solver
from mimrl_amd.data import get_data_loader

n = sys.argv[1] if len(sys.argv) > 1 else "1284"
epochs = int(sys.argv[2]) if len(sys.argv) > 2 else 6
argv = ["--dataset", "synthetic", "--synthetic_n", n, "--batch_size", "128", "--time_len", "50", "--d_hiddens", "50-3-128=10-3-128",
        "--d_outs", "50-3-128=10-3-128", "--bias", "--res_project", "1-1", "--dropout", "0.1-0.1-0.1-0.1", "--dropout_mlp", "0.0-0.0-0.0",
        "--loss_mi_coefficient1", "-".join(["1.0"] * 11), "--loss_mi_coefficient2", "-".join(["0.01"] * 8), "--learning_rate", "1e-4",
        "--precision", "bf16", "--task_name", "epoch_time", "--stage1_n", "1"]
for host in (False, True):
    opt = Parameters.parse_args(argv + (["--host_data"] if host else []))
    opt.seed, opt.save_best_features = 0, False
    sol = Solver(opt, get_data_loader(opt))
    banks = ([], [], [], [], [])
    ts = []
    for ep in range(epochs):
        torch.cuda.synchronize(); t = time.perf_counter()
        r = sol.train(ep, sol.train_loader, *banks)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
        banks = r[4:]
    nb = len(sol.train_loader)
    best = min(ts[2:])
    print("%s data: %d batches/epoch, epoch %.2f ms (best of %d after 2 warm-up epochs) = %.3f ms per batch (stage-1 + stage-2 pass) = %.0f two-stage iterations/s"
          % ("host    " if host else "resident", nb, 1e3 * best, epochs - 2, 1e3 * best / nb, nb / best))
