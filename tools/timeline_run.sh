#!/bin/bash
# kernel timeline of one steady-state bench step: tools/timeline_run.sh <workload> <steps> <out.txt> [env...]
wl=$1; steps=$2; out=$3; shift 3
root=$PWD; export TMPDIR=/tmp
(cd /tmp && rm -rf /tmp/p_tl && env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_tl -- python3 $root/bench.py --workload $wl --steps $steps --warmup 5 --no-cpu-baseline --prewarm-ms 0 --profile-steps 0 --no-extra > /tmp/p_tl.log 2>&1)
python tools/timeline.py /tmp/p_tl $out 3
