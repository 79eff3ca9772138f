#!/usr/bin/env python3
"""The l x l Jacobi SVD (activity-driven sweeps) through gsi.svd_tall on many matrices: graded, clustered, rank-deficient,
nearly orthogonal columns, widths on both sides of the 16-column blocking -- singular values against numpy (dgesdd).
    python tools/svd_small_stress.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gsi_amd as gsi
ctx = gsi.Context(0)
rng = np.random.default_rng(0)
worst = 0.0; count = 0
for l in (33, 48, 96, 160, 200, 256, 320, 336, 400, 512, 600):
    n = 3 * l + 7
    for kind in ("graded", "cluster", "rankdef", "nearorth", "flat", "power"):
        Q1 = np.linalg.qr(rng.standard_normal((n, l)))[0]; Q2 = np.linalg.qr(rng.standard_normal((l, l)))[0]
        if kind == "graded": s = np.logspace(0, -7, l)
        elif kind == "cluster": s = np.concatenate([np.full(l // 2, 1.0), np.full(l - l // 2, 1e-3)]) * (1 + 1e-9 * rng.standard_normal(l))
        elif kind == "rankdef": s = np.concatenate([np.logspace(0, -3, l - 20), np.zeros(20)])
        elif kind == "nearorth": s = 1.0 + 1e-7 * rng.standard_normal(l)
        elif kind == "flat": s = np.ones(l)
        else: s = (np.arange(l) + 1.0) ** -1.5
        W = (Q1 * s) @ Q2.T
        S, V = gsi.svd_tall(W)
        Sref = np.linalg.svd(W, compute_uv=False)
        err = float(np.abs(S - Sref).max() / Sref[0])
        orth = float(np.abs(V.T @ V - np.eye(l)).max()) if kind != "rankdef" else 0.0
        worst = max(worst, err); count += 1
        if err > 1e-12 or orth > 1e-11 or np.any(np.diff(S) > 0):
            print("FAIL", l, kind, err, orth); sys.exit(1)
print(f"{count} matrices, largest |S - S_dgesdd| / S_1 = {worst:.2e}; sweeps of the last one: {ctx.counters()}")
