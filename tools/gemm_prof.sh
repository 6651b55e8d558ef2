#!/bin/bash
# per-kernel durations of tools/gemm_bench.py from a rocprofv3 kernel trace (the wall-clock loop of the bench is launch-bound
# for kernels under ~20 us): tools/gemm_prof.sh [ENV=...]
export TMPDIR=/tmp; root=$PWD
(cd /tmp && rm -rf /tmp/p_gb && env PYTHONPATH=$root "$@" rocprofv3 --kernel-trace --output-format csv -d /tmp/p_gb -- python3 $root/tools/gemm_bench.py > /tmp/p_gb.log 2>&1)
python3 - <<'PY'
import csv, glob, re
rows=list(csv.DictReader(open(glob.glob("/tmp/p_gb/**/*kernel_trace.csv", recursive=True)[0])))
names=[l.split("  ")[0].strip() for l in open("/tmp/p_gb.log") if "us" in l and "TF/s" in l]
ks=[(int(r["Start_Timestamp"]), int(r["End_Timestamp"])-int(r["Start_Timestamp"]), r["Kernel_Name"]) for r in rows if "gemm" in r["Kernel_Name"]]
ks.sort()
# group consecutive launches of the bench: 5 warm-up + iters per case
i=0
import itertools
durs=[d for _,d,_ in ks]
kn=[re.sub(r"\(anonymous namespace\)::","",n).split("(")[0].replace("void mimrl::","") for _,_,n in ks]
pos=0
for line in open("/tmp/p_gb.log"):
    if "TF/s" not in line: continue
    name=line[:34].strip()
    iters=10 if name.startswith(("cfg3","concat")) else 50
    n=5+iters
    seg=durs[pos+5:pos+n]; kname=kn[pos+5] if pos+5 < len(kn) else "?"
    pos+=n
    if seg: print("%-44s %9.1f us (min %8.1f)  %s" % (name, sum(seg)/len(seg)/1e3, min(seg)/1e3, kname))
PY
