#!/bin/bash
# per-kernel times of the stand-alone sampler harness: tools/knn_bench.sh [N m k]
root=$(cd "$(dirname "$0")/.." && pwd)
N=${1:-16326}; m=${2:-128}; k=${3:-2}
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kb_prof
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kb_prof -- python3 $root/tools/knn_bench.py $N $m $k 50 > /tmp/kb.log 2>&1
tail -3 /tmp/kb.log
python3 - <<'PY'
import csv,glob
f=glob.glob('/tmp/kb_prof/**/*kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'knn' in r['Name'] or 'sample' in r['Name']: print("%-60s calls %4s avg %7.1f us min %7.1f" % (r['Name'].replace('mimrl::(anonymous namespace)::','')[:60], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3))
PY
