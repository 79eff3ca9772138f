"""Synthetic inputs shared by the tests, bench.py and the golden-vector generator.

Test data only.  The reference's inputs for this path are random matrices
(test/testrmf.jl:5-9), FFTRF power-law fields (test/testrpcga.jl:87) and Gram
matrices (test/testrpcga.jl:104-107); FFTRF itself is out of scope
(BASELINE.json north_star: stays on the Julia/CPU side), so ``powerlaw_fields``
is only a spectral-synthesis stand-in that produces fields of the same kind.
"""
import numpy as np


def exact_rank_matrix(rng, n, m):
    """``makeA(n, m) = randn(n, m) * randn(m, n)``  (test/testrmf.jl:5-9)."""
    return rng.standard_normal((n, m)) @ rng.standard_normal((m, n))


def grid_points(nx, ny):
    x, y = np.meshgrid(np.arange(nx, dtype=np.float64), np.arange(ny, dtype=np.float64), indexing="ij")
    return np.stack([x.ravel(), y.ravel()], axis=1)


def gaussian_cov(nx, ny, ell):
    """A_ij = exp(-|x_i-x_j|^2 / (2 ell^2)) on an nx x ny unit grid."""
    P = grid_points(nx, ny)
    d2 = ((P[:, None, :] - P[None, :, :]) ** 2).sum(-1)
    return np.exp(-d2 / (2.0 * ell * ell))


def exponential_cov(nx, ny, ell):
    P = grid_points(nx, ny)
    d = np.sqrt(((P[:, None, :] - P[None, :, :]) ** 2).sum(-1))
    return np.exp(-d / ell)


def powerlaw_fields(rng, shape, nfields, k0=2.0, dk=3.14, beta=-3.5):
    """Power-law random fields by spectral synthesis on a 2x padded periodic grid,
    cropped and renormalised to mean k0 / std dk (the kind of field
    test/testrpcga.jl:87 samples)."""
    ny, nx = shape
    fy = np.fft.fftfreq(2 * ny) * 2 * ny
    fx = np.fft.fftfreq(2 * nx) * 2 * nx
    f2 = fy[:, None] ** 2 + fx[None, :] ** 2
    with np.errstate(divide="ignore"):
        amp = np.where(f2 > 0, f2 ** (0.25 * beta), 0.0)
    out = []
    for _ in range(nfields):
        phi = rng.standard_normal(amp.shape)
        k = np.fft.ifft2(amp * np.exp(2j * np.pi * phi)).real[:ny, :nx]
        k = dk * (k - k.mean()) / k.std(ddof=1) + k0
        out.append(k.ravel().copy())
    return out


def rel_sv_err(S, Sref, K, floor=1e-10):
    """max_{i<K, sigma_i > floor*sigma_1} |S_i - Sref_i| / Sref_i  (SURVEY.md 8d)."""
    S = np.asarray(S)[:K]
    Sref = np.asarray(Sref)[:K]
    mask = Sref > floor * Sref[0]
    return float(np.max(np.abs(S[mask] - Sref[mask]) / Sref[mask]))


def random_shape_cases(seed, count, nmax):
    """(m, n, K, p, q, spectrum_decay) tuples for the shape sweeps: tiny, ragged, rectangular, l == min(m, n) ..."""
    rng = np.random.default_rng(seed)
    cases = []
    for _ in range(count):
        n = int(rng.integers(1, nmax + 1))
        m = n if rng.random() < 0.5 else int(rng.integers(max(1, n // 2), 2 * n + 1))
        l = int(rng.integers(1, min(m, n) + 1))
        p = int(rng.integers(0, l))          # K = l - p >= 1
        q = int(rng.integers(0, 4))
        cases.append((m, n, l - p, p, q, float(rng.uniform(0.3, 2.0))))
    return cases


def decaying_matrix(rng, m, n, decay):
    """m x n matrix with singular values (i+1)^-decay and Haar-ish factors: distinct, well separated spectrum."""
    r = min(m, n)
    U, _ = np.linalg.qr(rng.standard_normal((m, r)))
    V, _ = np.linalg.qr(rng.standard_normal((n, r)))
    return (U * (np.arange(1, r + 1.0) ** -decay)[None, :]) @ V.T
