#!/bin/bash
# PMC HBM traffic of the graph schedule (two separate passes, the guide's recipe): tools/pmc_hbm.sh <workload> <steps> <out.json>
set -u
wl=$1; steps=$2; out=$3; root=$PWD; export TMPDIR=/tmp
(cd /tmp && rm -rf /tmp/p_f && timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/p_f -- python3 $root/bench.py --workload $wl --steps $steps --warmup 2 --no-cpu-baseline --prewarm-ms 0 --profile-steps 0 --no-extra > /tmp/p_f.log 2>&1)
(cd /tmp && rm -rf /tmp/p_w && timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/p_w -- python3 $root/bench.py --workload $wl --steps $steps --warmup 2 --no-cpu-baseline --prewarm-ms 0 --profile-steps 0 --no-extra > /tmp/p_w.log 2>&1)
python3 tools/pmc_summary.py /tmp/p_f /tmp/p_w $out $((steps + 2))
