#!/usr/bin/env python3
"""The small l x l Jacobi SVD inside svd(B) (RandMatFact.jl:86) alone: milliseconds of the `svd` phase and sweeps, for the
persistent kernel (default) against one launch per round (GSI_SVD_PERSIST=0), each in a process of its own.
usage: python3 tools/svd_small_time.py"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import sys, json, numpy as np
sys.path.insert(0, sys.argv[1])
import gsi_amd as gsi
ctx = gsi.Context(0)
out = {}
for l, decay in [(320, 0.75), (320, 3.0), (256, 0.75), (160, 1.0), (48, 1.0), (600, 0.75)]:
    rng = np.random.default_rng(l)
    n = 4 * l + 37
    W = (rng.standard_normal((n, l)) * (np.arange(1, l + 1.0) ** -decay)) @ rng.standard_normal((l, l))
    ts, sw = [], None
    for rep in range(6):
        ctx.profile(True); ctx.phase_reset()
        S, V = gsi.svd_tall(W, ctx=ctx)
        ph = ctx.phase_times(); ctx.profile(False)
        ts.append(ph["svd"][0]); sw = ctx.counters()["jacobi_sweeps"]
    ref = np.linalg.svd(W, compute_uv=False)
    out["l=%d decay %.2f" % (l, decay)] = {"svd_ms_min": min(ts), "svd_ms_median": sorted(ts)[len(ts) // 2], "sweeps": sw,
                                           "sv_rel_err": float(np.max(np.abs(S - ref) / ref[0]))}
print(json.dumps(out))
'''
for tag, extra in (("persistent (default)", {}), ("one launch per round (GSI_SVD_PERSIST=0)", {"GSI_SVD_PERSIST": "0"})):
    env = dict(os.environ)
    env.update(extra)
    r = subprocess.run([sys.executable, "-c", CODE, ROOT], capture_output=True, text=True, env=env)
    print("#", tag)
    if r.returncode == 0 and r.stdout.strip():
        for k, v in json.loads(r.stdout.strip().splitlines()[-1]).items():
            print("  %-20s svd %.3f ms (median %.3f), %s sweeps, sv err %.1e" % (k, v["svd_ms_min"], v["svd_ms_median"], v["sweeps"], v["sv_rel_err"]))
    else:
        print(r.stderr[-3000:])
    sys.stdout.flush()
