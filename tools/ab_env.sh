#!/bin/bash
# same-box A/B of an environment knob: tools/ab_env.sh "<ENV=1 ...>" <workload> <steps> [reps]
# prints ms/step of alternating runs (baseline first)
knob=$1; wl=$2; steps=$3; reps=${4:-2}
for r in $(seq $reps); do
  a=$(python bench.py --workload $wl --steps $steps --warmup 10 --no-cpu-baseline --no-extra --profile-steps 0 2>/dev/null | python -c "import json,sys; print('%.3f' % json.loads(sys.stdin.read())['ms_per_step'])")
  b=$(env $knob python bench.py --workload $wl --steps $steps --warmup 10 --no-cpu-baseline --no-extra --profile-steps 0 2>/dev/null | python -c "import json,sys; print('%.3f' % json.loads(sys.stdin.read())['ms_per_step'])")
  echo "$wl default $a ms | $knob $b ms"
done
