"""Per-kernel averages of the tools/pmc_fft.sh passes: python tools/pmc_fft_summary.py <gpurun_out> <tag>"""
import collections, csv, glob, json, os, re, sys
root, tag = sys.argv[1], sys.argv[2]
out = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sorted(glob.glob(f"{root}/pmc_fft_{tag}_*/")):
    fs = glob.glob(d + "*/*counter_collection.csv")
    if not fs:
        continue
    seen = set()
    for r in csv.DictReader(open(max(fs, key=os.path.getmtime))):
        m = re.search(r"fft_pass_kernel<(\d+)[,>]", r["Kernel_Name"])
        if not m:
            continue
        k = "pass<%s>" % m.group(1)
        out[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if (d, r["Dispatch_Id"]) not in seen:
            seen.add((d, r["Dispatch_Id"]))
            out[k]["duration_us"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
res = {}
for k, v in out.items():
    a = {c: sum(x) / len(x) for c, x in v.items()}
    if "FETCH_SIZE" in a:
        a["fetch_MB_x2"] = 2 * a.pop("FETCH_SIZE") / 1024      # KB units; gfx950 tallies 128-B requests at 64 B
    if "WRITE_SIZE" in a:
        a["write_MB"] = a.pop("WRITE_SIZE") / 1024
    if "SQ_WAVE_CYCLES" in a:
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
            a["frac_" + c] = a[c] / a["SQ_WAVE_CYCLES"]
    if "SQ_LDS_IDX_ACTIVE" in a:
        a["lds_conflict_frac"] = a["SQ_LDS_BANK_CONFLICT"] / max(a["SQ_LDS_IDX_ACTIVE"], 1)
    res[k] = {c: (round(x, 4) if abs(x) < 1e4 else float(f"{x:.5g}")) for c, x in a.items()}
    print(k, json.dumps(res[k]))
json.dump(res, open(f"{root}/pmc_fft_{tag}.json", "w"), indent=1)
