#!/usr/bin/env python3
"""A/B two builds of the library on the thin QR (CholeskyQR2; RandMatFact.jl:57-58, 75-76) in ONE process on ONE GPU:
    python3 tools/ab_qr.py tools/ab/libgsi_prev.so [--rounds 4]
alternates gsi_qr_thinQ of a device-generated 10^6 x 320 panel (and 10^6 x 256, 125000 x 320, 65536 x 160) between the shipped
libgsi_hip.so and the given one: milliseconds of the `qr` phase per factorization, and whether Q and R are the same bits."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gsi_amd as gsi  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("other")
ap.add_argument("--rounds", type=int, default=4)
a = ap.parse_args()
libs = {"shipped": gsi._lib.load(), os.path.basename(a.other): gsi._lib.load(os.path.abspath(a.other))}
ctxs = {k: gsi.Context(0, lib=v) for k, v in libs.items()}
for m, l in [(1000000, 320), (1000000, 256), (125000, 320), (65536, 160)]:
    Yh = np.asfortranarray(np.random.default_rng(l).standard_normal((m, l)))
    tot = {k: 0.0 for k in libs}
    res = {}
    for r in range(a.rounds + 1):
        order = list(libs)
        for name in (order if r % 2 == 0 else order[::-1]):
            ctx = ctxs[name]
            ctx.profile(True); ctx.phase_reset()
            Q, R = gsi.qr_thinQ(Yh, return_R=True, ctx=ctx)
            ph = ctx.phase_times(); ctx.profile(False)
            if r > 0:
                tot[name] += ph["qr"][0]
            res[name] = (Q, R)
    names = list(libs)
    same = bool(np.array_equal(res[names[0]][0], res[names[1]][0]) and np.array_equal(res[names[0]][1], res[names[1]][1]))
    print(f"{m} x {l}: " + ", ".join(f"{k} {tot[k] / a.rounds:.3f} ms" for k in names) + f"; Q and R bit-identical: {same}", flush=True)
    del res
