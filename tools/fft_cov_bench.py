#!/usr/bin/env python3
"""Matrix-free FFT covariance operator (SURVEY 8 f2 / BASELINE configs[2]): product and randsvd timings.
    python tools/fft_cov_bench.py [--Ns 1000 1000] [--l 256] [--beta -3.5]
Algorithmic HBM bytes per column pair (16 B per complex double; the zero padding is neither stored nor read): forward
pass along axis a < d-1 reads prod_{b<a} M_b * N_a * prod_{b>a} N_b and writes the same with M_a, the inverse passes
mirror that; the last axis is one fused pass (forward, spectrum, inverse on the lines in LDS) that reads and writes
prod_{b<d-1} M_b * N_{d-1}; plus 8 B per embedded point for the spectrum."""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gsi_amd as gsi
ap = argparse.ArgumentParser()
ap.add_argument("--Ns", type=int, nargs="+", default=[1000, 1000])
ap.add_argument("--l", type=int, default=256)
ap.add_argument("--beta", type=float, default=-3.5)
ap.add_argument("--q", type=int, default=2)
ap.add_argument("--no-svd", action="store_true")
ap.add_argument("--fftrf", action="store_true", help="FFTRF.jl convention: 2N embedding, integer wavenumbers (power-of-two grids)")
a = ap.parse_args()
ctx = gsi.Context(0)
lib = ctx.lib
n = int(np.prod(a.Ns))
Ms = [1 if N == 1 else (2 * N if a.fftrf else 1 << int(np.ceil(np.log2(2 * N)))) for N in a.Ns]
M = int(np.prod(Ms)); d = sum(1 for m in Ms if m > 1)
op = gsi.fft_powerlaw_operator(ctx, a.Ns, a.beta, fftrf=a.fftrf)
X = gsi.DeviceMatrix(ctx, n, a.l).randn(1)
Y = gsi.DeviceMatrix(ctx, n, a.l)
gsi._lib.check(lib.gsi_op_mul_dev(ctx.h, op.h, 0, X.h, Y.h), lib); ctx.sync()
t0 = time.perf_counter(); reps = 3
for _ in range(reps):
    gsi._lib.check(lib.gsi_op_mul_dev(ctx.h, op.h, 0, X.h, Y.h), lib)
ctx.sync(); dt = (time.perf_counter() - t0) / reps
pairs = (a.l + 1) // 2
Nq = [N for N in a.Ns if N > 1]; Mq = [m for m in Ms if m > 1]
units = 0
for ax in range(len(Nq) - 1):
    lo = int(np.prod(Mq[:ax])); hi = int(np.prod(Nq[ax + 1:]))
    units += 2 * lo * (Nq[ax] + Mq[ax]) * hi          # forward + mirrored inverse
units += 2 * int(np.prod(Mq[:-1])) * Nq[-1]           # the last axis: forward, spectrum, inverse in one pass (read N, write N)
bytes_pair = 16 * units + 8 * M
print(f"grid {a.Ns} -> embedding {Ms}, n = {n}, l = {a.l}: A*X {dt*1e3:.2f} ms, "
      f"{pairs * bytes_pair / dt / 1e9:.0f} GB/s algorithmic ({bytes_pair/1e6:.0f} MB per column pair), device bytes {ctx.device_bytes()/1e9:.2f} GB", flush=True)
if not a.no_svd:
    K, p = a.l - a.l // 5, a.l // 5
    Y.close()                                              # 512^3: every n x l panel counts
    Z = gsi.DeviceMatrix(ctx, n, a.l); S = gsi.DeviceMatrix(ctx, a.l, 1)
    if n * a.l * 8 < 20e9:                                 # warm-up (skipped when the cached workspaces of a first call would not leave room for a second)
        gsi._lib.check(lib.gsi_randsvd_dev(ctx.h, op.h, X.h, K, p, a.q, Z.h, S.h), lib); ctx.sync()
    ctx.profile(True); ctx.phase_reset()
    t0 = time.perf_counter()
    gsi._lib.check(lib.gsi_randsvd_dev(ctx.h, op.h, X.h, K, p, a.q, Z.h, S.h), lib)
    ctx.sync(); dt = time.perf_counter() - t0
    ph = ctx.phase_times(); ctx.profile(False)
    print(f"randsvd K={K} p={p} q={a.q}: {dt*1e3:.1f} ms", {k: round(v[0], 1) for k, v in ph.items() if v[0] > 0}, ctx.counters(), flush=True)
    Sh = S.to_host()[:, 0]
    print("S[:5] =", Sh[:5], " S[K-1]/S[0] =", Sh[K - 1] / Sh[0])
