"""Which covariance do FFTRF.powerlaw_structuredgrid fields have?  (style of test/testrpcga.jl:60-81:
a sample covariance against the operator.)  oracle.fftrf_powerlaw_structuredgrid restates FFTRF.jl:8-100; the
circulant-embedding operator in FFTRF's own convention (`fftrf=True`: 2N embedding, integer wavenumbers) must be the
covariance of its raw fields, on square, non-square and non-power-of-two grids; the isotropic convention
(`fftrf=False`) coincides with it only on equal power-of-two axes.  CPU only."""
import numpy as np
import pytest

from oracle import oracle as orc
import cpuref


def _sample_cov(Ns, beta, nsamples, seed):
    rng = np.random.default_rng(seed)
    n = int(np.prod(Ns))
    acc = np.zeros((n, n))
    for _ in range(nsamples):
        f = orc.fftrf_powerlaw_structuredgrid(Ns, 0.0, 1.0, beta, rng, raw=True).reshape(-1, order="F")
        acc += np.outer(f, f)                                 # the raw field has mean zero by construction
    return acc / nsamples


@pytest.mark.parametrize("Ns,beta", [((8, 8), -3.5), ((8, 4), -3.5), ((6, 5), -2.5), ((4, 2, 4), -3.0)])
def test_fftrf_fields_have_the_fftrf_convention_covariance(Ns, beta):
    n = int(np.prod(Ns))
    nsamples = 6000
    C = _sample_cov(list(Ns), beta, nsamples, 1234)
    A = orc.fft_powerlaw_apply(np.eye(n), list(Ns), beta, fftrf=True)      # unit diagonal
    scale = np.trace(C) / n
    err = np.linalg.norm(C / scale - A) / np.linalg.norm(A)
    assert err < 0.09                                                       # Monte-Carlo error ~ 1/sqrt(samples)
    assert np.abs(np.diag(C) / scale - 1.0).max() < 0.1                     # stationary: constant variance
    # the isotropic power-of-two-embedding convention is a DIFFERENT operator unless all axes are equal powers of two
    B = orc.fft_powerlaw_apply(np.eye(n), list(Ns), beta, fftrf=False)
    same = len(set(Ns)) == 1 and all(N & (N - 1) == 0 for N in Ns)
    diff = np.linalg.norm(A - B) / np.linalg.norm(A)
    assert (diff < 1e-12) if same else (diff > 1e-3)


def test_fftrf_normalised_fields():
    """FFTRF.jl:94-98: every returned field has mean k0 and (corrected) standard deviation dk."""
    rng = np.random.default_rng(3)
    f = orc.fftrf_powerlaw_structuredgrid([25, 25], 2.0, 3.14, -3.5, rng)      # test/testrpcga.jl:87
    assert f.shape == (25, 25)
    assert abs(f.mean() - 2.0) < 1e-12 and abs(f.std(ddof=1) - 3.14) < 1e-12
    g = orc.fftrf_powerlaw_structuredgrid([6, 4, 5], 0.0, 1.0, -3.0, rng)
    assert g.shape == (6, 4, 5)


@pytest.mark.parametrize("Ns,beta", [((8, 4), -3.5), ((6, 5), -2.5), ((4, 2, 4), -3.0), ((16,), -2.0)])
def test_fftrf_convention_operator_cpuref(gsi, Ns, beta):
    """gsi_op_fft_powerlaw_fftrf through the shipped api.cpp / pipeline.cpp (CPU reference backend, naive DFT of any
    length) against the oracle's numpy-FFT definition."""
    lib = cpuref.load_cpuref()
    cx = gsi.Context(0, lib=lib)
    n = int(np.prod(Ns))
    op = gsi.fft_powerlaw_operator(cx, Ns, beta, fftrf=True)
    A = op.matmul(np.eye(n))
    Aref = orc.fft_powerlaw_apply(np.eye(n), list(Ns), beta, fftrf=True)
    assert np.abs(A - Aref).max() < 1e-12
    assert np.abs(A - A.T).max() < 1e-12 and np.abs(np.diag(A) - 1.0).max() < 1e-12
    op.close()
    cx.close()
