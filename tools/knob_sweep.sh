#!/bin/bash
# A/B of scheduling knobs on the cfg2 bench line (GPU box): tools/knob_sweep.sh "KNOB=V" "KNOB2=V" ...  -> gpurun_out/knob_sweep.txt
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; : > gpurun_out/knob_sweep.txt
run() { env "$@" timeout 300 python bench.py --no-cpu-baseline --no-extra --profile-steps 0 2>/dev/null | tail -1 | grep -o "ms_per_step\": [0-9.]*" | head -1; }
for rep in 1 2; do
  echo "default $(run A=1)" >> gpurun_out/knob_sweep.txt
  for k in "$@"; do echo "$k $(run $k)" >> gpurun_out/knob_sweep.txt; done
done
cat gpurun_out/knob_sweep.txt
