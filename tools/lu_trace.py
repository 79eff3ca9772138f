#!/usr/bin/env python3
"""Phase stamps of the LU leaf kernel's pivot steps (debug build tools/libgsi_hip_trace.so, -DGSI_LU_TRACE).
   GSI_HIP_LIB=tools/libgsi_hip_trace.so GSI_LU_TRACE=gpurun_out/lu_trace.txt python tools/lu_trace.py [m] [l]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gsi_amd as gsi
m = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
l = int(sys.argv[2]) if len(sys.argv) > 2 else 64
ctx = gsi.Context(0)
Y = gsi.DeviceMatrix(ctx, m, l).randn(3)
rng = np.random.default_rng(0)
# LU through the rangefinder-free primitive needs host data; use a small host panel upload instead
Yh = rng.standard_normal((m, l))
L = gsi.lu_L(Yh, ctx=ctx)
L = gsi.lu_L(Yh, ctx=ctx)
path = os.environ.get("GSI_LU_TRACE")
rows = [ln.split() for ln in open(path)]
names = ["argmax+3sync", "publish->poll ok", "leader publish / result seen", "sync", "update"]
for r in rows:
    t = [int(x) for x in r[2:]]
    if t[0] == 0:
        continue
    base = t[0]
    d = [(t[i + 1] - t[i]) * 10 if t[i + 1] and t[i] else None for i in range(5)]
    print(r[0], r[1], "ns:", d, "step total", (t[5] - t[0]) * 10)
