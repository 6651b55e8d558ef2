"""GPU box: two epochs of Solver.train on 16 device-resident cfg2 batches (the epoch schedule of bench.py's extras) -- the command
tools/epoch_timeline.sh profiles.  MIMRL_NO_EPOCH_PIPE=1: without the look-ahead forward pass of the critic passes."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from types import SimpleNamespace
args = SimpleNamespace(workload="cfg2", precision="bf16", no_graph=False)
t0 = time.time()
print(bench.epoch_schedule(args, 128, 50, nbatch=16, epochs=2))
