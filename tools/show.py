#!/usr/bin/env python3
"""Print the few numbers of a bench.py JSON line that an A/B run looks at:  python tools/show.py file.json"""
import json, sys
d = json.load(open(sys.argv[1]))
p = d["phases_ms_per_step"]
print(f'{d["ms_per_step"]:.2f} ms/step  gemm {p["gemm_n"] + p["gemm_t"]:.1f}  lu {p["lu"]:.2f}  qr {p["qr"]:.2f}  svd {p["svd"]:.2f}',
      d.get("path_counters"))
