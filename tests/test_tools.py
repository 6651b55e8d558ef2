"""The profile post-processing tools (tools/*.py) on tiny synthetic rocprofv3 CSVs.  CPU-only."""
import csv
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _write(path, header, rows):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(header)
        w.writerows(rows)


def test_pmc_summary(tmp_path):
    hdr = ["Dispatch_Id", "Kernel_Name", "Counter_Name", "Counter_Value"]
    name = "void mimrl::(anonymous namespace)::gru_fwd_kernel<true, true>(mimrl::GruFwdArgs)"
    _write(str(tmp_path / "f" / "x_counter_collection.csv"), hdr, [[i, name, "FETCH_SIZE", 1000.0] for i in range(4)])
    _write(str(tmp_path / "w" / "x_counter_collection.csv"), hdr, [[i, name, "WRITE_SIZE", 500.0] for i in range(4)])
    out = str(tmp_path / "pmc.json")
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "pmc_summary.py"), str(tmp_path / "f"), str(tmp_path / "w"), out, "2"])
    k = json.load(open(out))["kernels"]["gru_fwd_kernel<true, true>"]
    assert k["calls_per_step"] == 2.0
    assert k["fetch_bytes_corrected_per_launch"] == 1000.0 * 1024 * 2      # KB -> bytes, doubled on gfx950
    assert k["write_bytes_per_launch"] == 500.0 * 1024
    assert k["traffic_bytes_per_launch"] == 1000.0 * 1024 * 2 + 500.0 * 1024


def test_timeline(tmp_path):
    hdr = ["Kernel_Name", "Start_Timestamp", "End_Timestamp", "Queue_Id"]
    rows, t = [], 0
    for step in range(6):                       # per step: two anchor draws (stage 1, stage 2) and one other kernel each
        for k in ("mimrl::sample_anchors_kernel(int*)", "void mimrl::gemm_kernel<true, 64, true>(x)",
                  "mimrl::sample_anchors_kernel(int*)", "mimrl::adam_kernel(x)"):
            rows.append([k, t, t + 10000, 1])
            t += 12000
    _write(str(tmp_path / "p" / "x_kernel_trace.csv"), hdr, rows)
    out = str(tmp_path / "tl.txt")
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "timeline.py"), str(tmp_path / "p"), out, "2"])
    lines = open(out).read().strip().split("\n")
    assert lines[-1].startswith("step span") and "kernels 4" in lines[-1]
    assert "sample_anchors_kernel" in lines[0]
