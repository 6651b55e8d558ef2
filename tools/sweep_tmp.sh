for v in "MIMRL_GRU_LDS_PAD=140" "X=1" "MIMRL_GRU_LDS_PAD=148" "X=2" "MIMRL_GRU_LDS_PAD=140" "MIMRL_GRU_LDS_PAD=148"; do
  echo -n "$v  "; env $v python bench.py --steps 400 --warmup 30 --no-extra --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"
done
