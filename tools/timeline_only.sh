#!/bin/bash
# timeline of the captured cfg2 step (GPU box): -> gpurun_out/timeline_now.txt
root=$PWD; export TMPDIR=/tmp; mkdir -p gpurun_out
(cd /tmp && rm -rf /tmp/p_tl && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_tl -- python3 $root/bench.py --steps 50 --warmup 10 --no-cpu-baseline --prewarm-ms 0 --profile-steps 0 --no-extra "$@" > /tmp/p_tl.log 2>&1)
python3 tools/timeline.py /tmp/p_tl gpurun_out/timeline_now.txt 3
