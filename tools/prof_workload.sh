#!/bin/bash
# rocprofv3 kernel stats of one bench workload: tools/prof_workload.sh <workload> <steps> [extra bench args]
# -> gpurun_out/prof_<workload>/{kernel_stats.csv,bench.json}
set -u
wl=$1; steps=$2; shift 2
root=$PWD; out=gpurun_out/prof_$wl; mkdir -p $out
export TMPDIR=/tmp
(cd /tmp && rm -rf /tmp/p_$wl && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$wl -- python3 $root/bench.py --workload $wl --steps $steps --warmup 3 --no-cpu-baseline --prewarm-ms 0 --profile-steps 0 --no-extra "$@" > $root/$out/bench.json 2> $root/$out/bench.err)
cp $(find /tmp/p_$wl -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$out/kernel_stats.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
n=$steps+3
print("total kernel ms/step %.3f"%(tot/1e6/n))
for r in rows[:28]:
    print("%6.1f%% %8.1f us/step %6.1f calls/step avg %8.1f us  %s"%(100*float(r["TotalDurationNs"])/tot, float(r["TotalDurationNs"])/1e3/n, int(r["Calls"])/n, float(r["AverageNs"])/1e3, r["Name"][:90]))
PY
tail -c 300 $out/bench.json | head -c 300; echo
