#!/bin/bash
# same-box A/B on cfg2 and cfg3 (GPU box): tools/ab2.sh "<ENV=V ...>" ["<ENV2=V>" ...]  -> gpurun_out/ab2.txt
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; : > gpurun_out/ab2.txt
r2() { env "$@" timeout 300 python bench.py --no-cpu-baseline --no-extra --profile-steps 0 2>/dev/null | tail -1 | grep -o "ms_per_step\": [0-9.]*" | head -1; }
r3() { env "$@" timeout 300 python bench.py --workload cfg3 --steps 30 --warmup 5 --prewarm-ms 0 --profile-steps 0 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | grep -o "ms_per_step\": [0-9.]*" | head -1; }
for rep in 1 2; do
  echo "default cfg2 $(r2 A=1) cfg3 $(r3 A=1)" >> gpurun_out/ab2.txt
  for k in "$@"; do echo "$k cfg2 $(r2 $k) cfg3 $(r3 $k)" >> gpurun_out/ab2.txt; done
done
cat gpurun_out/ab2.txt
