#!/usr/bin/env python3
"""lu(Y).L of panels taller than the register-resident path holds: the resident kernel with overflow rows (default up to
5 x 2^20 rows) against the streamed lazily evaluated leaves, device-resident panels, phase timer.  (The per-column sweeps of
round 1 this was first measured against are retired: tools/rejected_kernels/panel_lu_round1_sweeps.hip.txt,
profiles/r03_lu_tall_time.log.)   python tools/lu_tall_time.py"""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import os, sys, json, numpy as np
sys.path.insert(0, sys.argv[1])
import gsi_amd as gsi
ctx = gsi.Context(0); lib = ctx.lib
out = {}
for m, l in [(1100000, 320), (1210000, 320), (1500000, 320), (2000000, 320), (3000000, 320), (4000000, 320), (16777216, 64), (134217728, 48)]:
    Y = gsi.DeviceMatrix(ctx, m, l)
    ts = []
    for rep in range(3):
        Y.randn(5)
        ctx.sync(); ctx.profile(True); ctx.phase_reset()
        gsi.lu_L_dev(Y)
        ph = ctx.phase_times(); ctx.profile(False)
        ts.append(ph["lu"][0])
    out["%dx%d" % (m, l)] = min(ts)
    Y.close(); ctx.release_cache()
print(json.dumps(out))
'''
for tag, extra in (("default (resident + overflow rows up to 5 x 2^20 rows, streamed beyond)", {}), ("streamed", {"GSI_LU_OV": "0"})):
    env = dict(os.environ); env.update(extra)
    r = subprocess.run([sys.executable, "-c", CODE, ROOT], capture_output=True, text=True, env=env)
    print(tag, r.stdout.strip().splitlines()[-1] if r.returncode == 0 and r.stdout.strip() else r.stderr[-2000:], flush=True)
