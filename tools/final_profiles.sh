#!/bin/bash
# Round-end evidence run (GPU box, from the repo root): bench lines, rocprofv3 kernel stats of the TIMED schedule, PMC HBM traffic, MFMA-busy
# counters, timelines, critical-path probe.  Everything lands in gpurun_out/final/; the summaries are then copied into profiles/ as r<NN>_*.
# usage: tools/final_profiles.sh [quick]      (quick: skip the critical-path sweep and the secondary workloads)
set -u
out=gpurun_out/final; mkdir -p $out
export TMPDIR=/tmp
root=$PWD
# kernel stats + timeline of the default bench: the SAME command the driver runs, minus the CPU legs (graph schedule = the timed one)
(cd /tmp && rm -rf /tmp/p_stats && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_stats -- python3 $root/bench.py --no-cpu-baseline --prewarm-ms 0 --no-extra > /tmp/p_stats.log 2>&1)
python3 tools/profile_summary.py /tmp/p_stats $out/kernel_stats.json 243      # 20 warm-up + 200 timed + 23 eager profile steps
cp $(find /tmp/p_stats -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
(cd /tmp && rm -rf /tmp/p_tl && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_tl -- python3 $root/bench.py --steps 50 --warmup 10 --no-cpu-baseline --prewarm-ms 0 --profile-steps 0 --no-extra > /tmp/p_tl.log 2>&1)
python3 tools/timeline.py /tmp/p_tl $out/timeline.txt 3
python3 tools/profile_summary.py /tmp/p_tl $out/kernel_stats_graph_only.json 60
cp $(find /tmp/p_tl -name "*kernel_stats.csv" | head -1) $out/kernel_stats_graph_only.csv
# PMC HBM traffic: two separate passes (the guide's recipe), graph schedule
(cd /tmp && rm -rf /tmp/p_f && timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/p_f -- python3 $root/bench.py --steps 10 --warmup 2 --no-cpu-baseline --prewarm-ms 0 --profile-steps 0 --no-extra > /tmp/p_f.log 2>&1)
(cd /tmp && rm -rf /tmp/p_w && timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/p_w -- python3 $root/bench.py --steps 10 --warmup 2 --no-cpu-baseline --prewarm-ms 0 --profile-steps 0 --no-extra > /tmp/p_w.log 2>&1)
python3 tools/pmc_summary.py /tmp/p_f /tmp/p_w $out/pmc_hbm_traffic.json 12
timeout 300 tools/pmc_mfma.sh cfg2 20 $out/pmc_mfma_busy_cfg2.json
for i in 1 2 3; do timeout 600 python bench.py $( [ $i -gt 1 ] && echo --no-cpu-baseline ) > $out/bench_default_$i.json 2> $out/bench_default_$i.err; done
if [ "${1:-}" != "quick" ]; then
  timeout 300 tools/pmc_mfma.sh cfg3 6 $out/pmc_mfma_busy_cfg3.json
  (cd /tmp && rm -rf /tmp/p_c3 && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_c3 -- python3 $root/bench.py --workload cfg3 --steps 10 --warmup 3 --no-cpu-baseline --prewarm-ms 0 --profile-steps 0 --no-extra > /tmp/p_c3.log 2>&1)
  python3 tools/profile_summary.py /tmp/p_c3 $out/kernel_stats_cfg3.json 13
  cp $(find /tmp/p_c3 -name "*kernel_stats.csv" | head -1) $out/kernel_stats_cfg3.csv
  python3 tools/timeline.py /tmp/p_c3 $out/timeline_cfg3.txt 3
  (cd /tmp && rm -rf /tmp/p_f3 && timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/p_f3 -- python3 $root/bench.py --workload cfg3 --steps 4 --warmup 2 --no-cpu-baseline --prewarm-ms 0 --profile-steps 0 --no-extra > /tmp/p_f3.log 2>&1)
  (cd /tmp && rm -rf /tmp/p_w3 && timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/p_w3 -- python3 $root/bench.py --workload cfg3 --steps 4 --warmup 2 --no-cpu-baseline --prewarm-ms 0 --profile-steps 0 --no-extra > /tmp/p_w3.log 2>&1)
  python3 tools/pmc_summary.py /tmp/p_f3 /tmp/p_w3 $out/pmc_hbm_traffic_cfg3.json 6
  timeout 200 python bench.py --no-cpu-baseline --workload cfg1 > $out/bench_cfg1.json 2>/dev/null
  timeout 200 python bench.py --no-cpu-baseline --workload cfg2-concat > $out/bench_cfg2_concat.json 2>/dev/null
  timeout 300 python bench.py --no-cpu-baseline --workload cfg3 --steps 30 --warmup 5 --profile-steps 5 > $out/bench_cfg3.json 2>/dev/null
  timeout 300 python bench.py --no-cpu-baseline --workload cfg5 --steps 50 --warmup 5 --profile-steps 5 > $out/bench_cfg5_bf16.json 2>/dev/null
  timeout 900 tools/critical_path.sh 100 > $out/critical_path.txt 2>&1; tail -3 $out/critical_path.txt
fi
tail -c 400 $out/bench_default_1.err
