#!/bin/bash
# average duration of kernels matching $1 under the given env knobs: tools/kstat.sh <pattern> "ENV=.." ...   (from the repo root)
pat=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
for ks in "$@"; do
  rm -rf /tmp/ks_prof
  (cd /tmp && TMPDIR=/tmp env $ks rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_prof -- python3 $root/bench.py --steps 30 --warmup 5 --profile-steps 0 --no-cpu-baseline --prewarm-ms 0 --no-extra > /tmp/ks.log 2>&1)
  f=$(find /tmp/ks_prof -name "*kernel_stats.csv" | head -1)
  echo "[$ks]"
  python3 - "$f" "$pat" <<'PY'
import csv, re, sys
for r in csv.DictReader(open(sys.argv[1])):
    if re.search(sys.argv[2], r["Name"]):
        nm = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").replace("mimrl::", "")
        print("   %-50s calls %5s avg_us %7.1f" % (nm.split("(")[0][:50], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
