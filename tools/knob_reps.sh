#!/bin/bash
# tools/knob_reps.sh <reps> "KNOB=V" ...: ms/step of the default cfg2 bench line per knob set, <reps> alternating fresh processes each (sorted) --
# schedule changes are judged on the distribution: a captured graph's queue mapping varies from process to process
reps=$1; shift
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; out=gpurun_out/knob_reps.txt; : > $out
run() { env "$@" timeout 300 python bench.py --no-cpu-baseline --no-extra --profile-steps 0 2>/dev/null | tail -1 | grep -o "ms_per_step\": [0-9.]*" | head -1 | cut -d' ' -f2 | cut -c1-6; }
declare -A acc
for rep in $(seq $reps); do
  acc[default]="${acc[default]} $(run A=1)"
  for k in "$@"; do acc[$k]="${acc[$k]} $(run $k)"; done
done
for k in default "$@"; do echo "$k: $(echo ${acc[$k]} | tr ' ' '\n' | sort -n | tr '\n' ' ')" >> $out; done
cat $out
