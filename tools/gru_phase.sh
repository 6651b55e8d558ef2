#!/bin/bash
# Phase elimination of the recurrence kernels (GPU box; needs mimrl_amd/libmimrl_hip_probe.so = a `make PHASE_PROBE=1` build of csrc/):
# launch durations of gru_fwd / gru_bwd at cfg2 with one phase of the cell step removed (MIMRL_GRU_SKIP, see gru.hip).
# usage: tools/gru_phase.sh [waves] -> prints a table, writes gpurun_out/gru_phase_w<waves>.json
root=$(cd "$(dirname "$0")/.." && pwd)
waves=${1:-8}
export MIMRL_LIB_PATH=$root/mimrl_amd/libmimrl_hip_probe.so TMPDIR=/tmp MIMRL_GRU_WAVES=$waves
mkdir -p $root/gpurun_out
echo "{" > $root/gpurun_out/gru_phase_w$waves.json
first=1
for sk in 0 1 2 4 8 16 32 3; do   # (63 = every phase off hung the box in round 4: not run)
  rm -rf /tmp/gp_prof
  (cd /tmp && MIMRL_GRU_SKIP=$sk timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/gp_prof -- python3 $root/bench.py --steps 30 --warmup 5 --profile-steps 0 --no-cpu-baseline --prewarm-ms 0 --no-extra > /tmp/gp.log 2>&1)
  f=$(find /tmp/gp_prof -name "*kernel_stats.csv" | head -1)
  python3 - "$f" "$sk" "$first" >> $root/gpurun_out/gru_phase_w$waves.json <<'PY'
import csv, sys
rows = {}
for r in csv.DictReader(open(sys.argv[1])):
    if "gru_fwd_kernel" in r["Name"] or "gru_bwd_kernel" in r["Name"]:
        nm = r["Name"].split("(anonymous namespace)::")[-1].split("(")[0]
        rows[nm] = float(r["AverageNs"]) / 1e3
print(("" if sys.argv[3] == "1" else ",") + '"skip_%s": %s' % (sys.argv[2], str(rows).replace("'", '"')))
sys.stderr.write("skip %3s  " % sys.argv[2] + "  ".join("%s %.1f us" % (k, v) for k, v in sorted(rows.items())) + "\n")
PY
  first=0
done
echo "}" >> $root/gpurun_out/gru_phase_w$waves.json
