"""Diagnostic: where the HOST time of one Solver.train stage-2 pass goes (cfg2, resident data) -- per-iteration wall time of each piece."""
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from collections import defaultdict
from mimrl_amd import Parameters, _lib
from mimrl_amd.Solver import Solver
from mimrl_amd.data import get_data_loader

argv = ["--dataset", "synthetic", "--synthetic_n", "12800", "--batch_size", "128", "--time_len", "50", "--d_hiddens", "50-3-128=10-3-128",
        "--d_outs", "50-3-128=10-3-128", "--bias", "--res_project", "1-1", "--dropout", "0.1-0.1-0.1-0.1", "--dropout_mlp", "0.0-0.0-0.0",
        "--loss_mi_coefficient1", "-".join(["1.0"] * 11), "--loss_mi_coefficient2", "-".join(["0.01"] * 8), "--learning_rate", "1e-4",
        "--precision", "bf16", "--task_name", "epoch_time", "--stage1_n", "1"] + sys.argv[1:]
opt = Parameters.parse_args(argv); opt.seed, opt.save_best_features = 0, False
sol = Solver(opt, get_data_loader(opt))
banks = ([], [], [], [], [])
for ep in range(2):
    banks = sol.train(ep, sol.train_loader, *banks)[4:]
sol._set_banks(*banks)
T = defaultdict(float)
acc = torch.zeros(_lib.NSCALARS, device=sol.engine.device)
for stage in (1, 2):
    torch.cuda.synchronize(); t_all = time.perf_counter()
    t0 = time.perf_counter()
    n = 0
    for e, datas in sol._iter_loaded(sol.train_loader):
        t1 = time.perf_counter(); T[stage, "iter (commit + previous stage_batch)"] += t1 - t0
        sol._anchors(e, stage); t2 = time.perf_counter(); T[stage, "anchors"] += t2 - t1
        sol._stage(e, stage); t3 = time.perf_counter(); T[stage, "stage launch"] += t3 - t2
        if stage == 2:
            x = [e.labels.reshape(-1, 1).clone()] + [e.feats[i].clone() for i in range(4)] + [e.pred.clone(), e.labels.clone()]
            acc[32:] += e.scalars[32:]
        else:
            acc[_lib.S1_LOSS] += e.scalars[_lib.S1_LOSS]
        t0 = time.perf_counter(); T[stage, "bookkeeping (clones, accumulate)"] += t0 - t3
        n += 1
    t_host = time.perf_counter() - t_all
    torch.cuda.synchronize(); t_dev = time.perf_counter() - t_all
    print("stage %d pass: %d batches, host loop %.3f ms/batch, until device idle %.3f ms/batch" % (stage, n, 1e3 * t_host / n, 1e3 * t_dev / n))
    for (s, k), v in T.items():
        if s == stage: print("    %-42s %.3f ms/batch" % (k, 1e3 * v / n))

# variants of the stage-2 pass (timing only)
from mimrl_amd.engine import HipEngine
def run_pass(label):
    torch.cuda.synchronize(); t = time.perf_counter(); n = 0
    for e, datas in sol._iter_loaded(sol.train_loader):
        sol._stage(e, 2); n += 1
    torch.cuda.synchronize(); print("%-60s %.3f ms/batch" % (label, 1e3 * (time.perf_counter() - t) / n))
run_pass("stage-2 pass, no bookkeeping")
orig_stage, orig_commit = HipEngine.stage_batch, HipEngine.commit_batch
HipEngine.stage_batch = lambda self, *a: None
run_pass("... stage_batch = no-op (commit only)")
HipEngine.commit_batch = lambda self: None
run_pass("... commit_batch = no-op too (same batch every step)")
HipEngine.stage_batch, HipEngine.commit_batch = orig_stage, orig_commit
def sb(self, *a):
    sol.engine.stream.synchronize(); orig_stage(self, *a)
e0 = sol.engine
e0.stage2_step(); torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(100): e0.stage2_step()
torch.cuda.synchronize(); print("%-60s %.3f ms/batch" % ("engine.stage2_step() back to back", 1e3 * (time.perf_counter() - t) / 100))
