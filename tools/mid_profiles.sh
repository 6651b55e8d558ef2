set -u
out=gpurun_out/mid; mkdir -p $out; export TMPDIR=/tmp; root=$PWD
(cd /tmp && rm -rf /tmp/p_tl && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_tl -- python3 $root/bench.py --steps 50 --warmup 10 --no-cpu-baseline --prewarm-ms 0 --profile-steps 0 --no-extra > /tmp/p_tl.log 2>&1)
python3 tools/timeline.py /tmp/p_tl $out/timeline.txt 3
timeout 300 tools/pmc_mfma.sh cfg3 6 $out/pmc_mfma_busy_cfg3.json
(cd /tmp && rm -rf /tmp/p_c3 && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_c3 -- python3 $root/bench.py --workload cfg3 --steps 10 --warmup 3 --no-cpu-baseline --prewarm-ms 0 --profile-steps 0 --no-extra > /tmp/p_c3.log 2>&1)
python3 tools/timeline.py /tmp/p_c3 $out/timeline_cfg3.txt 3
