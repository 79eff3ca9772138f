#!/usr/bin/env python3
"""Summarise a rocprofv3 kernel trace per (kernel, grid size): the big operator products (grid = 512 workgroups x
512 threads at C2) separately from the small panel products that share the kernel name.
    python tools/trace_summary.py <kernel_trace.csv> [out.json]"""
import csv, json, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(list)
for r in rows:
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")
    grid = int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1) or 1)
    us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    acc[(name, grid, us >= 2000.0)].append(us)     # dispatches >= 2 ms apart: the passes over the operator
out = []
for (name, grid, long_), v in acc.items():
    out.append({"kernel": name, "grid_threads": grid, "operator_pass": bool(long_), "calls": len(v), "avg_us": sum(v) / len(v), "min_us": min(v),
                "max_us": max(v), "total_ms": sum(v) / 1e3})
out.sort(key=lambda d: -d["total_ms"])
for d in out[:12]:
    print(f'{d["kernel"][:60]:60s} grid={d["grid_threads"]:>9d} {"pass" if d["operator_pass"] else "    "} calls={d["calls"]:5d} avg={d["avg_us"]:10.1f} us total={d["total_ms"]:8.2f} ms')
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], "w"), indent=1)
