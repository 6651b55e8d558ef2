"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE), corrected as
/opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950.
usage: pmc_summary.py <fetch pass dir> <write pass dir> <out.json> <steps in trace>"""
import collections, csv, glob, json, re, sys

fdir, wdir, dst, steps = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])

def collect(d, counter):
    """-> (sum per kernel, launches per kernel, steps seen): only dispatches from the first step on (the first sample_anchors launch: one
    per step since round 4) are counted -- the engine's one-off arena memset (tens of GB at cfg3) used to be averaged into the runtime's
    fill kernel and, through it, into the per-step total (round 5)"""
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    first = next((i for i, r in enumerate(rows) if "sample_anchors" in r["Kernel_Name"]), 0)
    rows = rows[first:]
    tot, cnt = collections.Counter(), collections.Counter()
    for r in rows:
        name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0].replace("void mimrl::", "").replace("mimrl::", "")
        tot[name] += float(r["Counter_Value"]); cnt[name] += 1
    nsteps = sum(v for k, v in cnt.items() if k.startswith("sample_anchors"))
    return tot, cnt, nsteps

ft, fc, fsteps = collect(fdir, "FETCH_SIZE")
wt, wc, wsteps = collect(wdir, "WRITE_SIZE")
if fsteps > 0:
    steps = fsteps
out = {"source": "rocprofv3 --pmc FETCH_SIZE --kernel-trace / rocprofv3 --pmc WRITE_SIZE --kernel-trace (two separate passes) -- "
                 "python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --prewarm-ms 0 --profile-steps 0 --no-extra (the replayed-graph schedule `value` is measured on; tools/final_profiles.sh)",
       "units": "rocprofv3 reports FETCH_SIZE/WRITE_SIZE in KB; bytes = KB*1024",
       "correction": "MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE tallies 128-B read requests at 64 B -> doubled here; WRITE_SIZE taken as is",
       "steps_in_trace": steps, "kernels": {}}
for k in sorted(ft):
    n = max(fc[k], 1)
    fb = ft[k] / n * 1024 * 2
    wb = wt.get(k, 0.0) / max(wc.get(k, 1), 1) * 1024
    out["kernels"][k] = {"calls_per_step": fc[k] / steps, "fetch_kb_raw_per_launch": ft[k] / n,
                         "fetch_bytes_corrected_per_launch": fb, "write_bytes_per_launch": wb, "traffic_bytes_per_launch": fb + wb}
out["traffic_bytes_per_step"] = sum(v["traffic_bytes_per_launch"] * v["calls_per_step"] for v in out["kernels"].values())
json.dump(out, open(dst, "w"), indent=1)
g = {k: v for k, v in out["kernels"].items() if "gru_fwd" in k}
print("wrote", dst, "steps", steps, "traffic per step %.2f GB" % (out["traffic_bytes_per_step"] / 1e9), "gru_fwd:", json.dumps(g))
