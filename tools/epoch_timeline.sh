#!/bin/bash
# timeline of one critic-pass step of the epoch schedule, with and without the look-ahead forward pass (GPU box) -> gpurun_out/epoch_tl_*.txt
export TMPDIR=/tmp; root=$PWD; mkdir -p gpurun_out
(cd /tmp && rm -rf /tmp/p_ep && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/p_ep -- python3 $root/tools/epoch_probe.py > /tmp/p_ep.log 2>&1)
tail -2 /tmp/p_ep.log | cut -c1-300
# the trace holds both modes (pipelined first); steps from the end: 16 model-pass steps, then the critic pass
python3 tools/timeline.py /tmp/p_ep gpurun_out/epoch_tl_nolook_stage1.txt 24
python3 tools/timeline.py /tmp/p_ep gpurun_out/epoch_tl_nolook_stage2.txt 6
python3 tools/timeline.py /tmp/p_ep gpurun_out/epoch_tl_pipe_stage1.txt 120
