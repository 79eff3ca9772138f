#!/usr/bin/env python3
"""Row-sharded LU with `world` ranks as threads on ONE GPU (GSI_LOCAL_COMM=1): the in-kernel pivot exchange between the ranks'
persistent leaf kernels against the per-step form (GSI_LU_NO_MR=1: three launches + one collective per pivot step).  Both
ranks share the GPU and the in-process collectives are host barriers, so the absolute times say little about 8 GPUs; the
launch and collective COUNTS are what carries over.   python tools/lu_mr_time.py [world] [rows] [l]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import os, sys, threading, time, numpy as np
sys.path.insert(0, %r)
import gsi_amd as gsi
world, m, l = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
ctx0 = gsi.Context(0); uid = ctx0.unique_id()
Y = np.random.default_rng(0).standard_normal((m, l))
bar = threading.Barrier(world); out = {}
def run(rank):
    ctx = ctx0 if rank == 0 else gsi.Context(0)
    ctx.comm_init(world, rank, uid)
    gsi.lu_L_sharded(Y, ctx=ctx)
    best = 1e30
    for _ in range(3):
        bar.wait(); ctx.profile(True); ctx.phase_reset()
        gsi.lu_L_sharded(Y, ctx=ctx)
        ph = ctx.phase_times(); ctx.profile(False)
        best = min(best, ph["lu"][0])
    out[rank] = best
ts = [threading.Thread(target=run, args=(r,)) for r in range(world)]
[t.start() for t in ts]; [t.join() for t in ts]
print(max(out.values()))
''' % ROOT
world = sys.argv[1] if len(sys.argv) > 1 else "2"
m = sys.argv[2] if len(sys.argv) > 2 else "250000"
l = sys.argv[3] if len(sys.argv) > 3 else "320"
for tag, extra in (("in-kernel exchange (persistent leaves)", {}), ("per-step launches + collectives", {"GSI_LU_NO_MR": "1"})):
    env = dict(os.environ); env.update(extra); env["GSI_LOCAL_COMM"] = "1"
    r = subprocess.run([sys.executable, "-c", code, world, m, l], capture_output=True, text=True, env=env, timeout=900)
    if r.returncode != 0:
        print(tag, "FAILED", r.stderr[-1500:]); sys.exit(1)
    print(f"{tag}: {world} ranks on one GPU, {m} x {l}: LU phase {float(r.stdout.strip().splitlines()[-1]):.2f} ms (slowest rank)", flush=True)
