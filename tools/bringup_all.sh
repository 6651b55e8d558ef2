mkdir -p gpurun_out/bringup
for args in "cfg3 fp32" "cfg3 bf16" "cfg3 bf16 graph prefetch" "cfg5 fp32" "cfg5 bf16 graph prefetch"; do
  echo "=== $args"; timeout 240 python tools/bringup.py $args 2>&1 | tail -12
done > gpurun_out/bringup/log.txt 2>&1
cat gpurun_out/bringup/log.txt
