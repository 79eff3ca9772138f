#!/usr/bin/env python3
"""Idle time BETWEEN kernels, per phase of a step, from a rocprofv3 --kernel-trace CSV (how much of the
panel phases is launch latency?).  A phase is a maximal run of consecutive dispatches whose kernels belong to one family
(lu, qr, svd); for every family: dispatches, busy time (sum of kernel durations), span (first start to last end of each run),
idle = span - busy, and the distribution of the gaps.  usage: python3 tools/trace_gaps.py <dir or kernel_trace.csv> [out.json]"""
import csv
import glob
import json
import os
import statistics
import sys

FAMILIES = [
    ("lu", ("lu_leaf_kernel", "lu_rankk_kernel", "lu_u12_kernel", "lu2_extract_L_kernel", "lu3_")),
    ("svd", ("jacobi_",)),
    ("qr_small", ("cq_", "sy_reduce_kernel", "splitk_reduce_kernel")),
    ("qr_tall", ("sy_kernel", "tr_kernel")),
]


def family(name, grid):
    for fam, keys in FAMILIES:
        if any(k in name for k in keys):
            return fam
    if "gemm_f64_kernel" in name:
        return "gemm_small" if grid < 256 * 512 else "gemm"      # the l x l products of the Cholesky rounds vs operator passes
    return "other"


def main():
    src = sys.argv[1]
    f = src if src.endswith(".csv") else glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = []
    for r in csv.DictReader(open(f)):
        g = int(r.get("Grid_Size_X", r.get("Grid_Size", "0")) or 0)
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], g))
    rows.sort()
    fams = {}
    # small gemms inside a qr_small run count as qr_small (the Cholesky's trailing updates / the inverse's products)
    labels = [family(n, g) for _, _, n, g in rows]
    for i, lab in enumerate(labels):
        if lab == "gemm_small":
            prev = next((labels[j] for j in range(i - 1, -1, -1) if labels[j] != "gemm_small"), "other")
            labels[i] = "qr_small" if prev in ("qr_small", "qr_tall") else ("svd" if prev == "svd" else "other_small")
    i = 0
    while i < len(rows):
        lab = labels[i]
        j = i
        while j + 1 < len(rows) and labels[j + 1] == lab:
            j += 1
        d = fams.setdefault(lab, {"runs": 0, "dispatches": 0, "busy_us": 0.0, "span_us": 0.0, "gaps_us": []})
        d["runs"] += 1
        d["dispatches"] += j - i + 1
        d["busy_us"] += sum(rows[k][1] - rows[k][0] for k in range(i, j + 1)) / 1e3
        d["span_us"] += (rows[j][1] - rows[i][0]) / 1e3
        d["gaps_us"] += [(rows[k + 1][0] - rows[k][1]) / 1e3 for k in range(i, j)]
        i = j + 1
    out = {}
    for lab, d in fams.items():
        g = d.pop("gaps_us")
        d["idle_us"] = d["span_us"] - d["busy_us"]
        if g:
            d["gap_median_us"] = statistics.median(g)
            d["gap_mean_us"] = sum(g) / len(g)
            d["gap_p90_us"] = sorted(g)[int(0.9 * (len(g) - 1))]
        out[lab] = d
    for lab, d in sorted(out.items(), key=lambda kv: -kv[1]["span_us"]):
        print(f"{lab:12s} runs {d['runs']:5d} dispatches {d['dispatches']:6d} busy {d['busy_us'] / 1e3:9.2f} ms span {d['span_us'] / 1e3:9.2f} ms "
              f"idle {d['idle_us'] / 1e3:8.2f} ms  gap median {d.get('gap_median_us', 0):6.2f} us mean {d.get('gap_mean_us', 0):6.2f} p90 {d.get('gap_p90_us', 0):6.2f}")
    if len(sys.argv) > 2:
        json.dump(out, open(sys.argv[2], "w"), indent=1)


if __name__ == "__main__":
    main()
