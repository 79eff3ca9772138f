#!/usr/bin/env python3
"""Many row-sharded LUs in a row through the persistent leaves with the in-kernel pivot exchange (one hop and two hops), ranks
as THREADS (GSI_LOCAL_COMM) and as PROCESSES (GSI_SHM_COMM) on one GPU: a protocol race shows as a poll time-out
(GSI_ERR_INTERNAL) or as pivots that differ from dgetrf's.   python tools/lu_mr_stress.py [iterations]"""
import os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ITER = sys.argv[1] if len(sys.argv) > 1 else "150"
BODY = r'''
import os, sys, time, numpy as np
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gsi_amd as gsi
from oracle import oracle as orc
shapes = [(30000, 72), (5000, 24), (399, 24), (120000, 136), (3001, 40), (64000, 64), (100, 25), (640, 72)]   # incl. one-workgroup shards
if os.environ.get("GSI_STRESS_SHAPES"):
    shapes = [tuple(int(v) for v in t.split("x")) for t in os.environ["GSI_STRESS_SHAPES"].split(",")]
panels = []
for k, (m, l) in enumerate(shapes):
    rng = np.random.default_rng(100 + k)
    Y = rng.standard_normal((m, l))
    Y[m // 2:m // 2 + 50] = Y[10:60]                       # ties across the shards
    panels.append((Y, orc.lu_pivots(Y)))
def run(ctx, iters):
    bad = 0
    for it in range(iters):
        Y, ref = panels[it % len(panels)]
        L, piv = gsi.lu_L_sharded(Y, return_pivots=True, ctx=ctx)
        if not np.array_equal(piv, ref):
            bad += 1
    return bad
'''
THREADS = BODY + r'''
import threading
world, iters = int(sys.argv[1]), int(sys.argv[2])
ctx0 = gsi.Context(0); uid = ctx0.unique_id(); res = {}
def th(rank):
    ctx = ctx0 if rank == 0 else gsi.Context(0)
    ctx.comm_init(world, rank, uid)
    res[rank] = run(ctx, iters)
ts = [threading.Thread(target=th, args=(r,)) for r in range(world)]
t0 = time.time(); [t.start() for t in ts]; [t.join() for t in ts]
assert len(res) == world, "a rank thread died"
print("bad", sum(res.values()), "seconds", round(time.time() - t0, 1))
'''
PROCS = BODY + r'''
world, rank, iters, tmp = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
ctx = gsi.Context(0)
idf = os.path.join(tmp, "uid")
if rank == 0:
    open(idf + ".tmp", "wb").write(bytes(ctx.unique_id())); os.rename(idf + ".tmp", idf)
while not os.path.exists(idf): time.sleep(0.01)
ctx.comm_init(world, rank, open(idf, "rb").read())
t0 = time.time()
print("bad", run(ctx, iters), "seconds", round(time.time() - t0, 1))
'''
def go(tag, env_extra, world, procs):
    env = dict(os.environ); env.update(env_extra); env["GSI_LU_MR_REQUIRE"] = "1"
    pre = "ROOT = %r\n" % ROOT
    if not procs:
        r = subprocess.run([sys.executable, "-c", pre + THREADS, str(world), ITER], capture_output=True, text=True, env=env, timeout=900)
        outs = [(r.returncode, r.stdout.strip().splitlines()[-1:] , r.stderr[-600:])]
    else:
        with tempfile.TemporaryDirectory() as tmp:
            ps = [subprocess.Popen([sys.executable, "-c", pre + PROCS, str(world), str(k), ITER, tmp], stdout=subprocess.PIPE,
                                   stderr=subprocess.PIPE, text=True, env=env) for k in range(world)]
            outs = []
            for p in ps:
                so, se = p.communicate(timeout=900)
                outs.append((p.returncode, so.strip().splitlines()[-1:], se[-600:]))
    ok = all(rc == 0 and o and o[0].startswith("bad 0 ") for rc, o, _ in outs)
    print(f"{tag}: world {world}, {ITER} LUs per rank:", "OK" if ok else "FAILED", [o for _, o, _ in outs], flush=True)
    if not ok:
        print(outs); sys.exit(1)
if os.environ.get("GSI_STRESS_ONLY"):          # e.g. GSI_STRESS_ONLY=procs:3:GSI_LU_MR_OV_GRID=40
    kind, world, *kv = os.environ["GSI_STRESS_ONLY"].split(":")
    extra = dict(t.split("=") for t in kv)
    extra.update({"GSI_SHM_COMM": "1", "GSI_SHM_TIMEOUT_S": "120"} if kind == "procs" else {"GSI_LOCAL_COMM": "1"})
    go(os.environ["GSI_STRESS_ONLY"], extra, int(world), kind == "procs")
    sys.exit(0)
go("threads, one hop", {"GSI_LOCAL_COMM": "1"}, 3, False)
go("threads, two hops", {"GSI_LOCAL_COMM": "1", "GSI_LU_MR_HIER": "1"}, 2, False)
go("processes, one hop", {"GSI_SHM_COMM": "1", "GSI_SHM_TIMEOUT_S": "120"}, 3, True)
go("processes, two hops", {"GSI_SHM_COMM": "1", "GSI_SHM_TIMEOUT_S": "120", "GSI_LU_MR_HIER": "1"}, 4, True)
go("threads, overflow rows (2 workgroups per rank)", {"GSI_LOCAL_COMM": "1", "GSI_LU_MR_OV_GRID": "2"}, 2, False)
go("processes, overflow rows (2 workgroups per rank)", {"GSI_SHM_COMM": "1", "GSI_SHM_TIMEOUT_S": "120", "GSI_LU_MR_OV_GRID": "2"}, 3, True)
