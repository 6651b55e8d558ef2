run() { env "$@" timeout 300 python bench.py --workload cfg3 --steps 30 --warmup 5 --prewarm-ms 0 --profile-steps 0 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%s cfg3 ms/step %.4f' % (' '.join(sys.argv[1:]), d['ms_per_step']))" "$@"; }
run A=0
run MIMRL_GRU_LDS_PAD=144
run MIMRL_BPTT_FIRST=1
run MIMRL_GRU_WAVES=8
run A=0
