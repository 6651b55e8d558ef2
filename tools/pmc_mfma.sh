#!/bin/bash
# MFMA-busy evidence (north_star: "rocprof HBM GB/s and MFMA-busy against gfx950 peak"): one PMC pass per workload, its own run
# (no FETCH/WRITE counters, no --stats), graph schedule = the timed one.  usage: tools/pmc_mfma.sh <workload> <steps> <out.json>
set -u
wl=$1; steps=$2; out=$3; shift 3
root=$PWD; export TMPDIR=/tmp
(cd /tmp && rm -rf /tmp/p_mf_$wl && rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_F32 --kernel-trace --output-format csv -d /tmp/p_mf_$wl -- python3 $root/bench.py --workload $wl --steps $steps --warmup 3 --no-cpu-baseline --prewarm-ms 0 --profile-steps 0 --no-extra "$@" > /tmp/p_mf_$wl.log 2>&1)
tail -2 /tmp/p_mf_$wl.log
python3 tools/pmc_mfma_summary.py /tmp/p_mf_$wl $out $((steps+3)) "workload $wl, graph schedule, $steps timed + 3 warm-up steps"
