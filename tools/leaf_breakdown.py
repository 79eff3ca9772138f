#!/usr/bin/env python3
"""Per-position timing of the LU kernels from a rocprofv3 kernel trace: python tools/leaf_breakdown.py <kernel_trace.csv>"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
dur = lambda r: (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
leaf = [dur(r) for r in rows if 'lu_leaf' in r['Kernel_Name']]
per = collections.defaultdict(list)
for i, d in enumerate(leaf):
    per[i % 8].append(d)
print("leaf avg by kp:", {k * 8: round(sum(v) / len(v), 1) for k, v in sorted(per.items())}, "n =", len(leaf))
rk = [dur(r) for r in rows if 'lu_rankk' in r['Kernel_Name']]
per = collections.defaultdict(list)
for i, d in enumerate(rk):
    per[i % 4].append(d)
print("rankk avg by block:", {k: round(sum(v) / len(v), 1) for k, v in per.items()})
u = [dur(r) for r in rows if 'lu_u12' in r['Kernel_Name']]
if u:
    print('u12', len(u), round(sum(u) / len(u), 1))
