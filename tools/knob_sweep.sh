#!/bin/bash
# usage: tools/knob_sweep.sh "ENV1=a ENV2=b" "ENV1=c" ...   -- one bench line (ms/step) per knob set, two repeats each (same box)
for rep in 1 2; do
  for ks in "$@"; do
    ms=$(env $ks timeout 200 python bench.py --profile-steps 0 --no-cpu-baseline --no-extra --steps 400 2>/dev/null | tail -1 | python -c "import sys,json; print('%.4f' % json.loads(sys.stdin.read())['ms_per_step'])")
    echo "rep$rep [$ks] $ms"
  done
done
