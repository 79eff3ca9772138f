#!/usr/bin/env python3
"""A/B two builds of the library on the FFT covariance product in ONE process (box-to-box variance exceeds the differences
looked for):  python tools/ab_fft.py tools/ab/libgsi_prev.so [--Ns 1000 1000] [--l 256] [--rounds 4]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gsi_amd as gsi
ap = argparse.ArgumentParser()
ap.add_argument("others", nargs="+")
ap.add_argument("--Ns", type=int, nargs="+", default=[1000, 1000])
ap.add_argument("--l", type=int, default=256)
ap.add_argument("--rounds", type=int, default=4)
ap.add_argument("--reps", type=int, default=4)
ap.add_argument("--fftrf", action="store_true")
a = ap.parse_args()
import numpy as np
n = int(np.prod(a.Ns))
libs = {"shipped": gsi._lib.load()}
for o in a.others:
    libs[os.path.basename(o)] = gsi._lib.load(os.path.abspath(o))
st = {}
for name, lib in libs.items():
    ctx = gsi.Context(0, lib=lib)
    st[name] = (ctx, gsi.fft_powerlaw_operator(ctx, a.Ns, -3.5, fftrf=a.fftrf), gsi.DeviceMatrix(ctx, n, a.l).randn(1), gsi.DeviceMatrix(ctx, n, a.l))
tot = {k: 0.0 for k in libs}
for r in range(a.rounds):
    order = list(libs)
    for name in (order if r % 2 == 0 else order[::-1]):
        ctx, op, X, Y = st[name]
        lib = ctx.lib
        gsi._lib.check(lib.gsi_op_mul_dev(ctx.h, op.h, 0, X.h, Y.h), lib); ctx.sync()
        t0 = time.perf_counter()
        for _ in range(a.reps):
            gsi._lib.check(lib.gsi_op_mul_dev(ctx.h, op.h, 0, X.h, Y.h), lib)
        ctx.sync()
        tot[name] += (time.perf_counter() - t0) / a.reps
ya = {name: st[name][3].to_host()[:, 0] for name in libs}
ref = ya["shipped"]
for name in libs:
    print(f"{name:18s} grid {a.Ns} l={a.l}: {1e3 * tot[name] / a.rounds:.3f} ms per product; max |diff vs shipped| {np.abs(ya[name] - ref).max():.2e}", flush=True)
