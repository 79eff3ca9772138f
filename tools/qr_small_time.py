#!/usr/bin/env python3
"""Thin QR (CholeskyQR2; RandMatFact.jl:75-76) of a tall panel: milliseconds of the `qr` phase with the l x l Cholesky and
inverse in ONE launch (default) against the blocked multi-launch form (GSI_CQ_FUSED=0), each in a process of its own; and
the distance of the two Q's ranges.   usage: python3 tools/qr_small_time.py"""
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
for m, l in [(1000000, 320), (1000000, 256), (125000, 320), (65536, 160), (20000, 48), (3000, 33)]:
    rng = np.random.default_rng(l)
    Y = gsi.DeviceMatrix(ctx, m, l).randn(3)
    Yh = Y.to_host() if m <= 125000 else None
    Y.close()
    if Yh is None:
        Yh = np.asfortranarray(rng.standard_normal((m, l)))
    ts = []
    for rep in range(4):
        ctx.profile(True); ctx.phase_reset()
        Q, R = gsi.qr_thinQ(Yh, return_R=True, ctx=ctx)
        ph = ctx.phase_times(); ctx.profile(False)
        ts.append(ph["qr"][0])
    orth = float(np.abs(Q[:20000].T @ Q[:20000] * (m / min(m, 20000)) - np.eye(l)).max()) if m > 20000 else float(np.abs(Q.T @ Q - np.eye(l)).max())
    rec = float(np.abs(Q[:5000] @ R - Yh[:5000]).max() / np.abs(Yh[:5000]).max())
    out["%dx%d" % (m, l)] = {"qr_ms_min": min(ts), "qr_ms_median": sorted(ts)[len(ts) // 2], "QR_minus_Y": rec, "counters": ctx.counters()}
print(json.dumps(out))
'''
for tag, extra in (("fused chol + inverse (default)", {}), ("blocked multi-launch (GSI_CQ_FUSED=0)", {"GSI_CQ_FUSED": "0"})):
    env = dict(os.environ)
    env.update(extra)
    r = subprocess.run([sys.executable, "-c", CODE, ROOT], capture_output=True, text=True, env=env)
    print("#", tag)
    if r.returncode == 0 and r.stdout.strip():
        for k, v in json.loads(r.stdout.strip().splitlines()[-1]).items():
            print("  %-14s qr %.3f ms (median %.3f)  |QR - Y| %.1e  %s" % (k, v["qr_ms_min"], v["qr_ms_median"], v["QR_minus_Y"], v["counters"]))
    else:
        print(r.stderr[-3000:])
    sys.stdout.flush()
