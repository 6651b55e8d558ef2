#!/bin/bash
# queue ids of the chain / side kernels of tools/hw/graph_order under rocprofv3 (GPU box)
export TMPDIR=/tmp; root=$PWD
(cd /tmp && rm -rf /tmp/go && timeout 120 rocprofv3 --kernel-trace --output-format csv -d /tmp/go -- $root/tools/hw/graph_order 3 > /tmp/go.log 2>&1)
f=$(find /tmp/go -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
print(rows[0].keys())
# last replay of variant B region is hard to find: print a window of 60 kernels from the middle of the trace
n = len(rows)
for seg in (n // 14, 3 * n // 14 + 10):
    w = rows[seg:seg + 36]
    t0 = int(w[0]["Start_Timestamp"])
    print("----")
    for r in w:
        g = r.get("Grid_Size_X") or r.get("Grid_Size")
        print("%8.1f %6.1f q%s grid %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Queue_Id"], g))
PY
