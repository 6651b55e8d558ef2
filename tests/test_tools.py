"""The profile post-processing tools (tools/*.py) on tiny synthetic rocprofv3 CSVs.  CPU-only."""
import csv
import json
import os
import subprocess
import sys

import pytest

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


def test_host_code_is_clean_under_asan_and_ubsan():
    """`make asan` + tools/asan_host.sh: the host-side translation units (knn_r1.cpp -- the scikit-learn KDTree restatement, layout.cpp,
    errors.cpp) built with -fsanitize=address,undefined and driven by the CPU tests of tests/test_knn_ties.py / test_layout.py with the
    sanitizer runtime preloaded into python (the reference has no sanitizer story; GPU sanitizers are not available on
    this pool, so this is the host half only)."""
    import shutil
    import subprocess
    rt = subprocess.run(["/opt/rocm/lib/llvm/bin/clang", "-print-file-name=libclang_rt.asan-x86_64.so"], capture_output=True, text=True).stdout.strip() \
        if os.path.exists("/opt/rocm/lib/llvm/bin/clang") else ""
    if not (rt and os.path.exists(rt) and shutil.which("make")):
        pytest.skip("no clang AddressSanitizer runtime in this image")
    r = subprocess.run(["bash", os.path.join(ROOT, "tools", "asan_host.sh")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    assert " passed" in r.stdout and "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr
