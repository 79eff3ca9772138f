#!/usr/bin/env python3
"""One process, `world` rank threads on one GPU (run with GSI_LOCAL_COMM=1): a few row-sharded LUs, for rocprofv3 --kernel-trace.
    GSI_LOCAL_COMM=1 python tools/lu_mr_probe.py [world] [rows] [l]"""
import os, sys, threading, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gsi_amd as gsi
world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
m = int(sys.argv[2]) if len(sys.argv) > 2 else 250000
l = int(sys.argv[3]) if len(sys.argv) > 3 else 320
ctx0 = gsi.Context(0); uid = ctx0.unique_id()
Y = np.random.default_rng(0).standard_normal((m, l))
bar = threading.Barrier(world); out = {}
def run(rank):
    ctx = ctx0 if rank == 0 else gsi.Context(0)
    ctx.comm_init(world, rank, uid)
    for it in range(3):
        bar.wait(); t0 = time.perf_counter()
        gsi.lu_L_sharded(Y, ctx=ctx)
        out[(rank, it)] = time.perf_counter() - t0
ts = [threading.Thread(target=run, args=(r,)) for r in range(world)]
[t.start() for t in ts]; [t.join() for t in ts]
print({k: round(v * 1e3, 1) for k, v in out.items()})
