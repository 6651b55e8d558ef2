python -m pytest tests/test_gpu_ops.py -q -m gpu -k "gru" 2>&1 | tail -3
python -m pytest tests/test_gpu_fused_oracle.py -q -m gpu -k "encoders" 2>&1 | tail -3
MIMRL_GRU_WAVES=4 python -m pytest tests/test_gpu_fused_oracle.py -q -m gpu -k "encoders" 2>&1 | tail -3
for w in 8 4 8 4; do echo "WAVES=$w"; MIMRL_GRU_WAVES=$w python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extra --profile-steps 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('ms/step %.4f'%d['ms_per_step'], [(k['kernel'][:30], round(k['avg_launch_us'],1)) for k in d['kernels'][:2]])"; done
