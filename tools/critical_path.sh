#!/bin/bash
# Critical-path sensitivity of the captured two-stage step: inject a spin kernel of US microseconds behind one phase
# (MIMRL_DBG_DELAY_TAG, engine_backward.hip: dbg_delay) and report the step-time increase per injected microsecond.
# usage: tools/critical_path.sh [US]      (run on the GPU box from the repo root)
US=${1:-100}
run() { env "$@" timeout 200 python bench.py --profile-steps 0 --no-cpu-baseline --no-extra --steps 150 2>/dev/null | tail -1 | python -c "import sys,json; print('%.4f' % json.loads(sys.stdin.read())['ms_per_step'])"; }
b1=$(run X=0); 
names=(x "gru_fwd(x2)" "cube_fwd tail s2" "MI fwd s1" "CMI fwd s1" "MI bwd s1" "CMI bwd s1" "cube_bwd" "gru_bwd(x2)" "wgrad sides(3 streams)" "text GEMM" "kNN s1" "adam(x2)" "cube_fwd tail s1" "CMI fwd s2" "MI fwd s2" "CMI bwd s2" "MI bwd s2" "kNN s2")
mult=(0 2 1 1 1 1 1 1 2 1 1 1 2 1 1 1 1 1 1)
for tag in 1 13 2 10 11 3 4 5 6 12 18 15 14 17 16 7 8 9; do
  ms=$(run MIMRL_DBG_DELAY_TAG=$tag:$US)
  echo "$tag|${names[$tag]}|${mult[$tag]}|$ms"
done > /tmp/cp.txt
b2=$(run X=0)
python - "$b1" "$b2" "$US" <<'PY'
import sys
b1, b2, us = float(sys.argv[1]), float(sys.argv[2]), float(sys.argv[3])
base = 0.5 * (b1 + b2)
print("baseline %.4f / %.4f ms, injected %d us per occurrence" % (b1, b2, us))
for line in open("/tmp/cp.txt"):
    tag, name, mult, ms = line.strip().split("|")
    d = (float(ms) - base) * 1e3
    print("tag %2s %-26s step %.4f ms  +%6.1f us  sensitivity %.2f" % (tag, name, float(ms), d, d / (us * int(mult))))
PY
