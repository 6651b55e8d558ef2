#!/bin/bash
# kernel stats of the deterministic build's cfg2 step (tools/det_prof.sh on the GPU box; summary -> gpurun_out/det_stats.txt)
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp MIMRL_DETERMINISTIC=1
rm -rf /tmp/detprof
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/detprof -- python3 $PWD/bench.py --steps 20 --warmup 3 --prewarm-ms 0 --profile-steps 0 --no-cpu-baseline --no-extra > /tmp/det_bench.log 2>&1
tail -5 /tmp/det_bench.log >&2; f=$(find /tmp/detprof -name "*kernel_stats.csv" | head -1)
mkdir -p gpurun_out
python3 - "$f" <<'PY' > gpurun_out/det_stats.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:25]:
    print("%-90s calls %7s avg %9.1f us  %5.1f %%" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
PY
tail -1 /tmp/det_bench.log | head -c 300 >> gpurun_out/det_stats.txt
