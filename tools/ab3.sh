#!/bin/bash
# same-box A/B on one workload (GPU box): tools/ab3.sh <workload> <steps> "<ENV=V ...>" ...  -> gpurun_out/ab3.txt
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; wl=$1; steps=$2; shift 2; : > gpurun_out/ab3.txt
r() { env "$@" timeout 300 python bench.py --workload $wl --steps $steps --warmup 5 --profile-steps 0 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | grep -o "ms_per_step\": [0-9.]*" | head -1; }
for rep in 1 2 3; do
  echo "default $wl $(r A=1)" >> gpurun_out/ab3.txt
  for k in "$@"; do echo "$k $wl $(r $k)" >> gpurun_out/ab3.txt; done
done
cat gpurun_out/ab3.txt
