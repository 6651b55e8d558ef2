#!/bin/bash
# LDS bank-conflict share of the round-6b streaming kernels (rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE, stand-alone benches)
root=$PWD; export TMPDIR=/tmp
for b in "gru_wgrad_bench.py 128000 80" "concat_dw_bench.py 256 5"; do
  (cd /tmp && rm -rf /tmp/p_lds && timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d /tmp/p_lds -- python3 $root/tools/$b > /tmp/p_lds.log 2>&1)
  python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for fn in glob.glob("/tmp/p_lds/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        n = r["Kernel_Name"]
        k = "gru_wgrad" if "gru_wgrad" in n else "concat_dw_kernel" if "concat_dw_kernel" in n else "gemm_fast_bf" if "gemm_fast_bf" in n else None
        if k: acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); acc[k]["n_" + r["Counter_Name"]] += 1
for k, v in acc.items():
    c, a = v.get("SQ_LDS_BANK_CONFLICT", 0), v.get("SQ_LDS_IDX_ACTIVE", 0)
    print("%-18s bank-conflict cycles / LDS-active cycles = %.3f   (%.3g / %.3g)" % (k, c / a if a else float("nan"), c, a))
PY
done
