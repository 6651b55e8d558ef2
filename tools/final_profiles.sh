#!/bin/bash
# Round-end evidence run (GPU box, from the repo root): bench lines, rocprofv3 kernel stats, PMC HBM traffic, timeline,
# critical-path probe.  Everything lands in gpurun_out/final/; the summaries are then copied into profiles/ as r<NN>_*.
# usage: tools/final_profiles.sh [quick]      (quick: skip the critical-path sweep)
set -u
out=gpurun_out/final; mkdir -p $out
export TMPDIR=/tmp
root=$PWD
for i in 1 2 3; do timeout 400 python bench.py $( [ $i -gt 1 ] && echo --no-cpu-baseline ) > $out/bench_default_$i.json 2> $out/bench_default_$i.err; done
timeout 200 python bench.py --no-cpu-baseline --workload cfg1 > $out/bench_cfg1.json 2>/dev/null
timeout 200 python bench.py --no-cpu-baseline --workload cfg2-concat > $out/bench_cfg2_concat.json 2>/dev/null
timeout 300 python bench.py --no-cpu-baseline --workload cfg3 --steps 30 --warmup 5 --profile-steps 5 > $out/bench_cfg3.json 2>/dev/null
timeout 300 python bench.py --no-cpu-baseline --workload cfg5 --steps 50 --warmup 5 --profile-steps 5 > $out/bench_cfg5_bf16.json 2>/dev/null
timeout 300 python bench.py --no-cpu-baseline --workload cfg5 --precision fp32 --steps 20 --warmup 3 --profile-steps 3 > $out/bench_cfg5_fp32.json 2>/dev/null
# kernel stats + timeline of the default bench
(cd /tmp && rm -rf /tmp/p_stats && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_stats -- python3 $root/bench.py --steps 50 --warmup 10 --no-cpu-baseline --profile-steps 0 --no-extra > /tmp/p_stats.log 2>&1)
python tools/profile_summary.py /tmp/p_stats $out/kernel_stats.json 60
cp $(find /tmp/p_stats -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
python tools/timeline.py /tmp/p_stats $out/timeline.txt 3
# cfg3 kernel stats
(cd /tmp && rm -rf /tmp/p_c3 && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_c3 -- python3 $root/bench.py --workload cfg3 --steps 10 --warmup 3 --no-cpu-baseline --profile-steps 0 --no-extra > /tmp/p_c3.log 2>&1)
python tools/profile_summary.py /tmp/p_c3 $out/kernel_stats_cfg3.json 13
cp $(find /tmp/p_c3 -name "*kernel_stats.csv" | head -1) $out/kernel_stats_cfg3.csv
# PMC HBM traffic: two separate passes (the guide's recipe), eager launches so that every kernel is attributed
(cd /tmp && rm -rf /tmp/p_f && rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/p_f -- python3 $root/bench.py --steps 10 --warmup 2 --no-cpu-baseline --profile-steps 0 --no-graph --no-extra > /tmp/p_f.log 2>&1)
(cd /tmp && rm -rf /tmp/p_w && rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/p_w -- python3 $root/bench.py --steps 10 --warmup 2 --no-cpu-baseline --profile-steps 0 --no-graph --no-extra > /tmp/p_w.log 2>&1)
python tools/pmc_summary.py /tmp/p_f /tmp/p_w $out/pmc_hbm_traffic.json 12
if [ "${1:-}" != "quick" ]; then tools/critical_path.sh 100 > $out/critical_path.txt 2>&1; tail -3 $out/critical_path.txt; fi
tail -c 400 $out/bench_default_1.err
