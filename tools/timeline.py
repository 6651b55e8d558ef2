"""Print the kernel timeline of ONE steady-state step from a rocprofv3 --kernel-trace CSV (start offset, duration, queue,
name) plus the idle gaps of the whole device -- the tool used to find what is on the critical path of the graph step.
usage: timeline.py <rocprof output dir> <out.txt> [step index from the end, default 3]"""
import csv, glob, re, sys

src, dst = sys.argv[1], sys.argv[2]
back = int(sys.argv[3]) if len(sys.argv) > 3 else 3
trace = glob.glob(src + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(trace)))
ks = []
for r in rows:
    name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0].replace("void mimrl::", "").replace("mimrl::", "")
    ks.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, r.get("Queue_Id", "?")))
ks.sort()
# a step starts at its anchor draw (round 4: ONE sample_anchors launch per step for both stages; rounds 1-3: two -- every second one)
marks = [i for i, k in enumerate(ks) if k[2].startswith("sample_anchors")]
if not any(k[2].startswith("knn_tile_kernel") for k in ks):
    marks = marks[::2]
lo, hi = marks[-back - 1], marks[-back]
step = ks[lo:hi]
t0 = step[0][0]
out = []
busy_end = t0
idle = 0
for s, e, n, q in step:
    gap = s - busy_end
    if gap > 0:
        idle += gap
    out.append("%9.1f %8.1f  q%-3s %s%s" % ((s - t0) / 1e3, (e - s) / 1e3, q, n, ("   <-- device idle %.1f us before" % (gap / 1e3)) if gap > 1500 else ""))
    busy_end = max(busy_end, e)
out.append("step span %.1f us, kernels %d, sum of durations %.1f us, device-idle %.1f us" % (
    (busy_end - t0) / 1e3, len(step), sum(e - s for s, e, _, _ in step) / 1e3, idle / 1e3))
open(dst, "w").write("\n".join(out) + "\n")
print(out[-1])
