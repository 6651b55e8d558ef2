"""Per-kernel MFMA utilisation from one rocprofv3 --pmc pass (SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE ...).
MI355X_MICROARCH.md: SQ_VALU_MFMA_BUSY_CYCLES counts cycles (32 per v_mfma_f32_32x32x16_bf16, summed over all SIMDs);
GRBM_GUI_ACTIVE is summed over the 8 XCDs, so the kernel's busy clock cycles = GRBM_GUI_ACTIVE / 8.
mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 256 CUs * 4 SIMDs): the share of all MFMA pipes' cycles that
were busy while the kernel was resident (gfx94x MfmaUtil formula; ROCm 7.2 ships no gfx950 derived-counter section).
usage: pmc_mfma_summary.py <pmc pass dir> <out.json> <steps in trace> [note]"""
import collections, csv, glob, json, re, sys

src, dst, steps = sys.argv[1], sys.argv[2], int(sys.argv[3])
note = sys.argv[4] if len(sys.argv) > 4 else ""
f = glob.glob(src + "/**/*counter_collection.csv", recursive=True)[0]
tot = collections.defaultdict(collections.Counter)
cnt = collections.Counter()
seen = set()
for r in csv.DictReader(open(f)):
    name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0].replace("void mimrl::", "").replace("mimrl::", "")
    tot[name][r["Counter_Name"]] += float(r["Counter_Value"])
    key = (r.get("Dispatch_Id"), name)
    if key not in seen:
        seen.add(key); cnt[name] += 1
NCU, NSIMD, NXCD = 256, 4, 8
out = {"source": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE [SQ_INSTS_VALU_MFMA_MOPS_BF16 ...] --kernel-trace "
                 "--output-format csv -- python3 bench.py ... (its own pass: no FETCH/WRITE counters, no stats)", "note": note,
       "formula": "mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 * 256 CU * 4 SIMD)", "steps_in_trace": steps, "kernels": {}}
agg_m = agg_g = 0.0
for k in sorted(tot, key=lambda k: -tot[k].get("GRBM_GUI_ACTIVE", 0.0)):
    c = tot[k]; n = max(cnt[k], 1)
    gui = c.get("GRBM_GUI_ACTIVE", 0.0) / NXCD
    mf = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    agg_m += mf; agg_g += gui
    row = {"calls_per_step": cnt[k] / steps, "gui_cycles_per_launch": gui / n, "mfma_busy_cycles_per_launch": mf / n,
           "mfma_busy_frac": mf / (gui * NCU * NSIMD) if gui else 0.0}
    for extra in c:
        if extra not in ("GRBM_GUI_ACTIVE", "SQ_VALU_MFMA_BUSY_CYCLES"):
            row[extra.lower() + "_per_launch"] = c[extra] / n
    out["kernels"][k] = row
out["all_kernels"] = {"mfma_busy_cycles_per_step": agg_m / steps, "gui_cycles_per_step_summed_over_kernels": agg_g / steps,
                      "mfma_busy_frac_time_weighted": agg_m / (agg_g * NCU * NSIMD) if agg_g else 0.0}
json.dump(out, open(dst, "w"), indent=1)
print("wrote", dst, json.dumps(out["all_kernels"]))
for k, v in list(out["kernels"].items())[:12]:
    print("%6.2f%% mfma-busy  %9.0f gui cyc/launch  %5.1f calls/step  %s" % (100 * v["mfma_busy_frac"], v["gui_cycles_per_launch"], v["calls_per_step"], k[:80]))
