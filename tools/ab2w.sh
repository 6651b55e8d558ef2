#!/bin/bash
# A/B two prebuilt libraries on the same box for a workload: tools/ab2w.sh <workload> <steps> ab/lib_old.so ab/lib_new.so  (alternating, 3 rounds; restores the last one)
wl=$1; steps=$2; shift 2
for rep in 1 2 3; do
  for so in "$@"; do
    cp "$so" mimrl_amd/libmimrl_hip.so
    ms=$(timeout 300 python bench.py --workload $wl --steps $steps --warmup 10 --profile-steps 0 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "import sys,json; print('%.4f' % json.loads(sys.stdin.read())['ms_per_step'])")
    echo "$wl rep$rep [$so] $ms"
  done
done
