"""The driver's multi-GPU command on the one-GPU box: `GSI_BENCH_ONE_GPU=1 python3 bench.py --gpus 2
--steps 3` -- bench.py starts its own two rank processes (both on GPU 0, joined by the library's shared-memory communicator;
RCCL refuses two ranks on one device), and the line must say what ran: n_gpus, the LU form and the self-test mask of the
in-kernel pivot exchange, collectives per step, the ranks the communicator joined, the N-rank numbers against the one-GPU
path, and ONE row-sharded step of the implicit 10^6 x 10^6 covariance (the operator north_star's >= 6x names)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.skipif(bool(os.environ.get("GSI_SKIP_BENCH_MULTIRANK")), reason="GSI_SKIP_BENCH_MULTIRANK set")
def test_bench_gpus_2_starts_by_itself_on_one_gpu(gsi):
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "GSI_BENCH_RDV"):
        env.pop(k, None)
    env["GSI_BENCH_ONE_GPU"] = "1"
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "3"], capture_output=True, text=True,
                       timeout=1500, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-6000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[-4000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["config"]["n"] == 1000000
    pc = out["path_counters"]
    assert pc["n_ranks_seen"] == 2
    assert pc["lu_form"].startswith("persistent"), pc          # the in-kernel exchange, not the per-step collectives
    assert pc["lu_selftest_mask"] > 0 and pc["lu_form_same_on_all_ranks"] is True
    assert pc["lu_timeouts"] == 0
    assert 0 < pc["collectives_per_step"] < 200, pc           # 4 LUs x 6 + products, TSQR, svd: tens, not the ~1500 of the per-step form
    mr = out["multi_rank_vs_one_gpu"]
    assert mr["sv_rel_err_vs_one_gpu"] < 1e-9 and mr["xis_err_up_to_sign_vs_one_gpu_rank0_rows"] < 1e-6, mr
    imp = out["secondary"]["implicit_dense_1e6"]
    assert "error" not in imp, imp
    assert imp["n_gpus"] == 2 and imp["lu_form"] == "replicated"
    assert imp["ZtZ_diag_vs_S_max_rel"] < 1e-9 and imp["trailing_p_columns_max_abs"] == 0.0 and imp["sv_descending_positive"]
    assert imp["ms_per_step"] > 0
