#!/bin/bash
# cfg2: BPTT launch durations (in-graph stamps) and step time under environment knobs: tools/ab_stamps.sh "<ENV=V>" ...
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; : > gpurun_out/ab_stamps.txt
for k in A=1 "$@" A=1 "$@"; do
  echo "== $k" >> gpurun_out/ab_stamps.txt
  env $k python tools/stamp_gaps.py cfg2 200 2>/dev/null | tail -9 | grep -E "BPTT|period|dh0" >> gpurun_out/ab_stamps.txt
  env $k timeout 300 python bench.py --no-cpu-baseline --no-extra --profile-steps 0 2>/dev/null | tail -1 | grep -o "ms_per_step\": [0-9.]*" | head -1 >> gpurun_out/ab_stamps.txt
done
cat gpurun_out/ab_stamps.txt
