"""Per-kernel averages of tools/pmc_kernels.sh passes: python tools/pmc_kernel_summary.py <gpurun_out> <tag> <substr> [...]"""
import collections, csv, glob, json, os, sys
root, tag, subs = sys.argv[1], sys.argv[2], sys.argv[3:]
out = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sorted(glob.glob(f"{root}/pmc_k_{tag}_*/")):
    fs = glob.glob(d + "*/*counter_collection.csv")
    if not fs:
        continue
    seen = set()
    for r in csv.DictReader(open(max(fs, key=os.path.getmtime))):
        k = next((s for s in subs if s in r["Kernel_Name"]), None)
        if k is None:
            continue
        out[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if (d, r["Dispatch_Id"]) not in seen:
            seen.add((d, r["Dispatch_Id"]))
            out[k]["duration_us"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
res = {}
for k, v in out.items():
    a = {c: sum(x) / len(x) for c, x in v.items()}
    if "GRBM_GUI_ACTIVE" in a:
        cyc = a["GRBM_GUI_ACTIVE"] / 8
        a["clock_GHz"] = cyc / (a["duration_us"] * 1e3)
    if "SQ_WAVE_CYCLES" in a:
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
            a["frac_" + c] = a[c] / a["SQ_WAVE_CYCLES"]
    if "SQ_VALU_MFMA_BUSY_CYCLES" in a and "SQ_BUSY_CYCLES" in a:
        a["mfma_busy_over_sq_busy"] = a["SQ_VALU_MFMA_BUSY_CYCLES"] / a["SQ_BUSY_CYCLES"]
    if "SQ_LDS_IDX_ACTIVE" in a:
        a["lds_conflict_frac"] = a["SQ_LDS_BANK_CONFLICT"] / max(a["SQ_LDS_IDX_ACTIVE"], 1)
    res[k] = {c: (round(x, 4) if abs(x) < 1e4 else float(f"{x:.5g}")) for c, x in a.items()}
    print(k, json.dumps(res[k]))
json.dump(res, open(f"{root}/pmc_k_{tag}.json", "w"), indent=1)
