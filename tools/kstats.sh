#!/bin/bash
# top kernels of a bench run under rocprofv3 (GPU box): tools/kstats.sh <out.txt> [ENV=V ...] -- <bench args>
out=$1; shift; envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
root=$PWD; export TMPDIR=/tmp; mkdir -p gpurun_out; for e in "${envs[@]}"; do export "$e"; done
(cd /tmp && rm -rf /tmp/ks && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- python3 $root/bench.py --no-cpu-baseline --prewarm-ms 0 --profile-steps 0 --no-extra "$@" > /tmp/ks.log 2>&1)
f=$(find /tmp/ks -name "*kernel_stats.csv" | head -1)
python3 - "$f" > gpurun_out/$out <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:16]:
    print("%-84s calls %6s avg %9.1f us  %5.1f %%" % (r["Name"].replace("void mimrl::(anonymous namespace)::", "")[:84], r["Calls"], float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
PY
cat gpurun_out/$out
