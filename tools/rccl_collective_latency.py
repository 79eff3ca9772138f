#!/usr/bin/env python3
"""What ONE small RCCL collective costs on the library's stream, measured ("measure, don't assume"):
the row-sharded LU issues one all-gather of a (4 + 2 l)-double record per pivot step plus an all-reduce per leaf / block
(pipeline.cpp:lu_panel_sharded: ~364 collectives per LU at l = 320).  With a 1-rank communicator (GSI_FORCE_COMM=1) every
one of them still goes through librccl's launch path on the stream -- a LOWER bound of the multi-GPU cost (no wire, no
peer synchronisation).  Prints the LU phase time with and without the communicator and the difference per collective.
    python tools/rccl_collective_latency.py [rows] [l]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import os, sys, numpy as np
sys.path.insert(0, %r)
import gsi_amd as gsi
m, l = int(sys.argv[1]), int(sys.argv[2])
ctx = gsi.Context(0)
if os.environ.get("GSI_FORCE_COMM"):
    ctx.comm_init(1, 0, ctx.unique_id())
Y = np.random.default_rng(0).standard_normal((m, l))
gsi.lu_L_sharded(Y, ctx=ctx)
best = 1e30
for _ in range(3):
    ctx.profile(True); ctx.phase_reset()
    gsi.lu_L_sharded(Y, ctx=ctx)
    ph = ctx.phase_times(); ctx.profile(False)
    best = min(best, ph["lu"][0])
print(best)
''' % ROOT
m = sys.argv[1] if len(sys.argv) > 1 else "125000"
l = sys.argv[2] if len(sys.argv) > 2 else "320"
out = {}
for tag, extra in (("no communicator", {}), ("1-rank RCCL communicator", {"GSI_FORCE_COMM": "1"})):
    env = dict(os.environ); env.update(extra); env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, "-c", code, m, l], capture_output=True, text=True, env=env, timeout=600)
    if r.returncode != 0:
        print(tag, "FAILED", r.stderr[-1500:]); sys.exit(1)
    out[tag] = float(r.stdout.strip().splitlines()[-1])
    print(f"{tag}: sharded LU of {m} x {l}: {out[tag]:.2f} ms", flush=True)
L = int(l)
ncoll = L + (L // 8 - (L + 63) // 64) + max((L + 63) // 64 - 1, 0)     # one all-gather per pivot + leaf and block U12 all-reduces
d = out["1-rank RCCL communicator"] - out["no communicator"]
print(f"collectives per LU: {ncoll}; through librccl they add {d:.2f} ms = {1e3 * d / ncoll:.1f} us per collective (1 rank: lower bound)")
