#!/bin/bash
# one-off knob sweep (same box): ms/step per knob, default first and last
run() { env $1 python bench.py --workload $2 --steps $3 --warmup 10 --no-cpu-baseline --no-extra --profile-steps 0 2>/dev/null | python -c "import json,sys; print('%.4f' % json.loads(sys.stdin.read())['ms_per_step'])"; }
for wl in cfg3:40 cfg2:300; do
  w=${wl%%:*}; s=${wl##*:}
  for k in A=1 MIMRL_GRU_WAVES=8 MIMRL_NO_XIN=1 MIMRL_GRU_LDS_PAD=0 MIMRL_REC16=0 MIMRL_NO_H16=1 MIMRL_NO_DUAL_TAIL_PRE=1 MIMRL_NO_FUSED_TAIL_PRE=1 MIMRL_NO_GEMM_TALL=1 MIMRL_ADAM_FRAG=0 A=1; do
    echo "$w $k $(run $k $w $s)"
  done
done
