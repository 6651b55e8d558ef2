#!/bin/bash
# HBM fetch bytes per launch of the one-pass GRU weight-gradient kernel and of the GEMM pair (rocprofv3 --pmc FETCH_SIZE): tools/wg_fetch.sh [rows] [kp]
root=$PWD; export TMPDIR=/tmp
(cd /tmp && rm -rf /tmp/p_wg && timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/p_wg -- python3 $root/tools/gru_wgrad_bench.py ${1:-128000} ${2:-80} > /tmp/p_wg.log 2>&1)
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("/tmp/p_wg/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: [0.0, 0])
for fn in f:
    for r in csv.DictReader(open(fn)):
        if r.get("Counter_Name") != "FETCH_SIZE":
            continue
        k = r["Kernel_Name"].split("(")[0][-60:]
        acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
for k, (v, n) in sorted(acc.items(), key=lambda x: -x[1][0])[:6]:
    print("%-62s launches %4d  fetch per launch %8.1f MB (KB x 1024 x 2: the guide's gfx950 correction)" % (k, n, v / n * 1024 * 2 / 1e6))
PY
