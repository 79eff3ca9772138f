"""Generate the committed golden vectors under tests/golden/ from the scipy oracle.

The reference is Julia and cannot run in this image, and it ships no fixtures of its own, so these
vectors come from oracle/oracle.py (same LAPACK routines, explicit Omega).  They freeze the oracle:
tests compare oracle-now, the C restatement and the HIP path against them.
Run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import oracle as orc                      # noqa: E402
from helpers import gaussian_cov, exponential_cov, powerlaw_fields   # noqa: E402


def main():
    rng = np.random.default_rng(20261004)
    # case 1: dense Gaussian covariance, q = 2
    A = gaussian_cov(16, 12, 3.0)                       # n = 192
    K, p, q = 12, 6, 2
    Om = rng.standard_normal((192, K + p))
    Z, S, Q = orc.randsvd_full(A, K, p, q, Om)
    Y = A @ Om
    np.savez_compressed(os.path.join(HERE, "dense_gauss_n192.npz"), grid=np.array([16, 12]), ell=3.0, K=K, p=p, q=q,
                        Omega=Om, S=S, Z=Z, Q=Q, lu_pivots=orc.lu_pivots(Y), lu_L=orc.lu_L(Y))
    # case 2: exponential covariance, q = 0 and q = 3
    A = exponential_cov(12, 12, 5.0)                    # n = 144
    Om = rng.standard_normal((144, 10))
    out = {"grid": np.array([12, 12]), "ell": 5.0, "K": 7, "p": 3, "Omega": Om}
    for qq in (0, 3):
        Z, S, Q = orc.randsvd_full(A, 7, 3, qq, Om)
        out[f"S_q{qq}"] = S
        out[f"Z_q{qq}"] = Z
    np.savez_compressed(os.path.join(HERE, "dense_exp_n144.npz"), **out)
    # case 3: LowRankCovMatrix (the shape of test/testrpcga.jl:83-102, reduced)
    fields = np.array(powerlaw_fields(rng, (10, 10), 24))   # N = 24, n = 100
    Om = rng.standard_normal((100, 9))
    xis, _ = orc.getxis_fields(list(fields), 6, 3, 3, Om)
    np.savez_compressed(os.path.join(HERE, "lowrank_n100_N24.npz"), fields=fields, Omega=Om, K=6, p=3, q=3,
                        xis=np.array(xis), dense=orc.LowRankCovMatrix(fields).todense())
    # case 4: Nystrom KAT of test/testrmf.jl:21-29 (closed form)
    np.savez_compressed(os.path.join(HERE, "nystrom_kat.npz"), A=np.array([[2.0, -1, 0], [-1, 2, -1], [0, -1, 2]]),
                        eigenvalues=np.array([2 + np.sqrt(2), 2.0, 2 - np.sqrt(2)]))
    print("wrote", sorted(f for f in os.listdir(HERE) if f.endswith(".npz")))


if __name__ == "__main__":
    main()
