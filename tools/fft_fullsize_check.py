#!/usr/bin/env python3
"""The FFT covariance operator at FULL size against a host FFT (scipy.fft, all cores) of the oracle's own spectrum: the
small-grid parity tests cannot see an index that overflows at 10^9 embedding points.  Two columns per grid.
    python tools/fft_fullsize_check.py            (512^3 needs ~60 GB of host memory and about a minute)"""
import os, sys, time
import numpy as np
import scipy.fft as sfft
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gsi_amd as gsi
from oracle import oracle as orc
ctx = gsi.Context(0)
cases = [((3000, 3000), -3.5, True), ((300, 300, 300), -3.5, True), ((512, 512, 512), -3.5, True), ((700, 900), -2.5, False)]
if len(sys.argv) > 1:
    cases = [c for c in cases if "x".join(map(str, c[0])) in sys.argv[1:]]
for Ns, beta, fftrf in cases:
    n = int(np.prod(Ns))
    rng = np.random.default_rng(n)
    X = np.asfortranarray(rng.standard_normal((n, 2)))
    t0 = time.perf_counter()
    op = gsi.fft_powerlaw_operator(ctx, list(Ns), beta, fftrf=fftrf)
    Y = op.matmul(X)
    op.close(); ctx.release_cache()
    t_gpu = time.perf_counter() - t0
    t0 = time.perf_counter()
    lam, Ms = orc.fft_powerlaw_spectrum(list(Ns), beta, fftrf)
    box = tuple(slice(0, N) for N in Ns)
    err = 0.0; ref_max = 0.0
    for c in range(2):
        w = np.zeros(Ms)
        w[box] = X[:, c].reshape(Ns, order="F")
        f = sfft.fftn(w, workers=-1)
        del w
        f *= lam
        y = sfft.ifftn(f, workers=-1, overwrite_x=True).real[box].reshape(-1, order="F")
        del f
        err = max(err, float(np.abs(Y[:, c] - y).max())); ref_max = max(ref_max, float(np.abs(y).max()))
    print(f"grid {Ns} (n = {n}, embedding {tuple(Ms)}, {'FFTRF' if fftrf else 'isotropic'} convention, beta = {beta}): "
          f"max |HIP - host FFT| = {err:.3e} = {err / ref_max:.2e} of max |A x|   (HIP incl. plan {t_gpu:.1f} s, host {time.perf_counter() - t0:.1f} s)",
          flush=True)
