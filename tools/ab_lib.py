#!/usr/bin/env python3
"""A/B two builds of the library in ONE process on ONE GPU (box-to-box variance is +-2 %):
    python tools/ab_lib.py tools/ab/libgsi_old.so [--l 160] [--rounds 4]
alternates the operator products A*X / A'*X of the shipped libgsi_hip.so and of the given library."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gsi_amd as gsi
ap = argparse.ArgumentParser()
ap.add_argument("others", nargs="+")
ap.add_argument("--grid", type=int, default=256)
ap.add_argument("--l", type=int, default=160)
ap.add_argument("--rounds", type=int, default=4)
ap.add_argument("--reps", type=int, default=5)
a = ap.parse_args()
n = a.grid * a.grid
libs = {"shipped": gsi._lib.load()}
for o in a.others:
    libs[os.path.basename(o)] = gsi._lib.load(os.path.abspath(o))
st = {}
for name, lib in libs.items():
    ctx = gsi.Context(0, lib=lib)
    st[name] = (ctx, gsi.gridcov_operator(ctx, a.grid, a.grid, 16.0, 0), gsi.DeviceMatrix(ctx, n, a.l).randn(1),
                gsi.DeviceMatrix(ctx, n, a.l))
tot = {k: [0.0, 0.0] for k in libs}
for r in range(a.rounds):
    order = list(libs)
    for name in (order if r % 2 == 0 else order[::-1]):
        ctx, op, X, Y = st[name]
        lib = ctx.lib
        for trans in (0, 1):
            gsi._lib.check(lib.gsi_op_mul_dev(ctx.h, op.h, trans, X.h, Y.h), lib)
            ctx.sync(); ctx.profile(True); ctx.phase_reset()
            for _ in range(a.reps):
                gsi._lib.check(lib.gsi_op_mul_dev(ctx.h, op.h, trans, X.h, Y.h), lib)
            ph = ctx.phase_times(); ctx.profile(False)
            key = "gemm_t" if trans else "gemm_n"
            tot[name][trans] += ph[key][0] / ph[key][1]
for name in libs:
    print(f"{name:18s}", f"l={a.l}: NN {tot[name][0]/a.rounds:.3f} ms  TN {tot[name][1]/a.rounds:.3f} ms", flush=True)
