"""Condense a rocprofv3 --kernel-trace --stats output directory into a small committed summary (profiles/)."""
import collections, csv, glob, json, re, sys
src, dst, steps = sys.argv[1], sys.argv[2], int(sys.argv[3])
stats = glob.glob(src + "/**/*kernel_stats.csv", recursive=True)[0]
trace = glob.glob(src + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(stats)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
out = {"source": "rocprofv3 --kernel-trace --stats --output-format csv", "steps_profiled": steps,
       "total_kernel_ms_per_step": tot / 1e6 / steps, "kernels": []}
for r in rows:
    name = re.sub(r"\(anonymous namespace\)::", "", r["Name"]).split("(")[0]
    out["kernels"].append({"name": name, "calls_per_step": int(r["Calls"]) / steps, "avg_us": float(r["AverageNs"]) / 1e3,
                           "min_us": float(r["MinNs"]) / 1e3, "max_us": float(r["MaxNs"]) / 1e3,
                           "share": float(r["TotalDurationNs"]) / tot})
json.dump(out, open(dst, "w"), indent=1)
print("wrote", dst, "kernels:", len(rows), "ms/step", out["total_kernel_ms_per_step"])
