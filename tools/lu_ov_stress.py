#!/usr/bin/env python3
"""The resident LU kernel with lazily evaluated overflow rows against the streamed leaves on many device-generated panels
just above the resident window: pivots and a checksum of L must agree bit for bit.   python tools/lu_ov_stress.py [n]"""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N = sys.argv[1] if len(sys.argv) > 1 else "60"
CODE = r'''
import os, sys, json, hashlib, numpy as np
sys.path.insert(0, sys.argv[1])
import gsi_amd as gsi
ctx = gsi.Context(0); out = []
rng = np.random.default_rng(0)
for k in range(int(sys.argv[2])):
    m = int(rng.integers(1048577, 1400000)); l = int(rng.choice([8, 17, 24, 40, 64, 72]))
    Y = gsi.DeviceMatrix(ctx, m, l).randn(1000 + k)
    _, piv = gsi.lu_L_dev(Y, return_pivots=True)
    L = Y.to_host()
    out.append([m, l, hashlib.sha256(piv.tobytes()).hexdigest()[:16], hashlib.sha256(np.ascontiguousarray(L).tobytes()).hexdigest()[:16]])
    Y.close()
print(json.dumps(out))
'''
res = {}
for tag, extra in (("overflow rows", {}), ("streamed", {"GSI_LU_OV": "0"})):
    env = dict(os.environ); env.update(extra)
    r = subprocess.run([sys.executable, "-c", CODE, ROOT, N], capture_output=True, text=True, env=env)
    if r.returncode != 0:
        print(tag, "FAILED", r.stderr[-1500:]); sys.exit(1)
    res[tag] = json.loads(r.stdout.strip().splitlines()[-1])
bad = [a for a, b in zip(res["overflow rows"], res["streamed"]) if a != b]
print(f"{N} panels of 1 048 577 .. 1 400 000 rows: {len(bad)} differ", bad[:3])
sys.exit(1 if bad else 0)
