"""Element-wise pinning of the oracle (and, on a GPU box, of the HIP path) to the REFERENCE's own output.

The fixtures come from `julia/make_reference_fixtures.jl`, which runs GeostatInversion.jl's RandMatFact on fixed seeds
where Julia and the package exist and writes Omega (the `randn` stream the reference consumed), its Z and S.  The build
image has no Julia, so these files are normally ABSENT and the tests skip -- DESIGN.md section 3 then says "parity pinned
by the reference's KATs/properties only".  When someone drops the fixtures into tests/golden/ref_*/, the oracle is pinned."""
import os

import numpy as np
import pytest

from oracle import oracle as orc
from helpers import rel_sv_err

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load(case):
    d = os.path.join(GOLD, case)
    if not os.path.isdir(d):
        pytest.skip(f"tests/golden/{case}/ not present (run julia/make_reference_fixtures.jl where Julia + GeostatInversion.jl exist)")
    out = {k[:-4]: np.load(os.path.join(d, k)) for k in os.listdir(d) if k.endswith(".npy")}
    out["params"] = dict(kv.split("=") for kv in open(os.path.join(d, "params.txt")).read().split())
    return out


def _check(Z, S, ref, K):
    assert rel_sv_err(S, ref["S"], K) < 1e-9                              # north_star bar: 1e-5
    assert orc.xis_error_up_to_sign(Z, ref["Z"], K) < 1e-6                # test/testrpcga.jl:100
    assert np.all(Z[:, K:] == 0) and np.all(ref["Z"][:, K:] == 0)         # RandMatFact.jl:87-88


def _operator(ref, case, mod):
    if case == "ref_c1":
        return ref["A"] if mod is orc else ref["A"]
    fields = ref["fields"].T                                              # one field per row
    return mod.LowRankCovMatrix(fields)


@pytest.mark.parametrize("case", ["ref_c1", "ref_lrcm625"])
def test_oracle_equals_reference(case):
    ref = _load(case)
    K, p, q = (int(ref["params"][k]) for k in ("K", "p", "q"))
    Z, S, _ = orc.randsvd_full(_operator(ref, case, orc), K, p, q, ref["Omega"])
    _check(Z, S, ref, K)


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["ref_c1", "ref_lrcm625"])
def test_hip_equals_reference(gsi, case):
    ref = _load(case)
    K, p, q = (int(ref["params"][k]) for k in ("K", "p", "q"))
    Z, S = gsi.randsvd(_operator(ref, case, gsi), K, p, q, Omega=ref["Omega"], return_S=True)
    _check(Z, S, ref, K)
