#!/bin/bash
# Round-end evidence run (GPU box, from the repo root): bench lines, rocprofv3 kernel stats, PMC HBM traffic, critical-path
# probe.  Everything lands in gpurun_out/final/; the summaries are then copied into profiles/.
set -u
out=gpurun_out/final; mkdir -p $out
export TMPDIR=/tmp
timeout 300 python bench.py > $out/bench_default.json 2> $out/bench_default.err
timeout 200 python bench.py --no-cpu-baseline --workload cfg1 > $out/bench_cfg1.json 2>/dev/null
timeout 200 python bench.py --no-cpu-baseline --workload cfg2-concat > $out/bench_cfg2_concat.json 2>/dev/null
timeout 200 python bench.py --no-cpu-baseline --profile-steps 0 --no-prefetch > $out/bench_no_prefetch.json 2>/dev/null
MIMRL_NO_SHARED_PREFIX=1 timeout 200 python bench.py --no-cpu-baseline --profile-steps 0 > $out/bench_no_shared_prefix.json 2>/dev/null
root=$PWD
(cd /tmp && rm -rf /tmp/p_stats && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_stats -- python3 $root/bench.py --steps 50 --warmup 10 --no-cpu-baseline > /tmp/p_stats.log 2>&1)
python tools/profile_summary.py /tmp/p_stats $out/kernel_stats.json 83
cp $(find /tmp/p_stats -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
python tools/timeline.py /tmp/p_stats $out/timeline.txt 30
(cd /tmp && rm -rf /tmp/p_f && rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/p_f -- python3 $root/bench.py --steps 10 --warmup 2 --no-cpu-baseline --profile-steps 0 --no-graph > /tmp/p_f.log 2>&1)
(cd /tmp && rm -rf /tmp/p_w && rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/p_w -- python3 $root/bench.py --steps 10 --warmup 2 --no-cpu-baseline --profile-steps 0 --no-graph > /tmp/p_w.log 2>&1)
python tools/pmc_summary.py /tmp/p_f /tmp/p_w $out/pmc_hbm_traffic.json 12
tools/critical_path.sh 100 > $out/critical_path.txt 2>&1
tail -3 $out/critical_path.txt; tail -c 600 $out/bench_default.err
