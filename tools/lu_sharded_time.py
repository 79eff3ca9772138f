#!/usr/bin/env python3
"""Device time of the row-sharded LU's local work (one rank, no collectives) against the register-resident LU:
   python tools/lu_sharded_time.py [rows] [l]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gsi_amd as gsi
m = int(sys.argv[1]) if len(sys.argv) > 1 else 125000
l = int(sys.argv[2]) if len(sys.argv) > 2 else 320
ctx = gsi.Context(0)
Y = np.random.default_rng(0).standard_normal((m, l))
for name, fn in (("register-resident lu_L", gsi.lu_L), ("row-sharded lu_L_sharded (1 rank)", gsi.lu_L_sharded)):
    fn(Y, ctx=ctx)
    ctx.profile(True); ctx.phase_reset()
    fn(Y, ctx=ctx)
    ph = ctx.phase_times(); ctx.profile(False)
    print(f"{name}: m={m} l={l}: LU phase {ph['lu'][0]:.2f} ms", flush=True)
