"""Summarise rocprofv3 --pmc CSVs for the big gemm dispatches: python tools/pmc_summary.py <tag>"""
import collections, csv, glob, json, sys
tag = sys.argv[1]
out = {}
for d in (f"pmc_{tag}_a", f"pmc_{tag}_b", f"pmc_{tag}_c"):
    fs = glob.glob(f"gpurun_out/{d}/*/*counter_collection.csv")
    if not fs:
        continue
    rows = list(csv.DictReader(open(fs[0])))
    for kn in ("gemm_f64_kernel<10, true, 0, 0>", "gemm_f64_kernel<10, false, 0, 0>"):
        byd = collections.defaultdict(dict)
        for r in rows:
            if kn in r["Kernel_Name"]:
                byd[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
                byd[r["Dispatch_Id"]]["duration_ms"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
        if byd:
            out.setdefault(kn, {}).update(max(byd.values(), key=lambda x: x["duration_ms"]))
for kn, v in out.items():
    if "GRBM_GUI_ACTIVE" in v:
        cyc = v["GRBM_GUI_ACTIVE"] / 8
        v["derived_clock_GHz"] = cyc / (v["duration_ms"] * 1e6)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in v:
            v["derived_mfma_util"] = v["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * cyc)
    if "SQ_WAVE_CYCLES" in v:
        for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
            v["frac_" + k] = v[k] / v["SQ_WAVE_CYCLES"]
    print(kn, json.dumps({k: (round(x, 4) if x < 1e4 else float(f"{x:.4g}")) for k, x in v.items()}))
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], "w"), indent=1)
