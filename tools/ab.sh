#!/bin/bash
# A/B two prebuilt libraries on the same box: tools/ab.sh ab/lib_old.so ab/lib_new.so  (alternating, 3 rounds)
for rep in 1 2 3; do
  for so in "$@"; do
    cp "$so" mimrl_amd/libmimrl_hip.so
    ms=$(timeout 200 python bench.py --profile-steps 0 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; print('%.4f' % json.loads(sys.stdin.read())['ms_per_step'])")
    echo "rep$rep [$so] $ms"
  done
done
