"""`python3 bench.py --gpus N` must start by itself: the process the driver starts becomes a launcher that
never touches the GPU, spawns N rank processes with RANK / LOCAL_RANK / WORLD_SIZE and a rendezvous directory, relays rank 0's
one line and fails when any rank does.  CPU only: the ranks stop before loading the product library (GSI_BENCH_LAUNCH_TEST)."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(mode, n, extra_env=None):
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "GSI_BENCH_RDV"):
        env.pop(k, None)
    env["GSI_BENCH_LAUNCH_TEST"] = mode
    env.update(extra_env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "3"],
                          capture_output=True, text=True, timeout=120, env=env, cwd=ROOT)


def test_launcher_spawns_ranks_and_relays_rank0_line():
    r = _run("ok", 3)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["launch_test"] == 3 and out["token"] == "launch-test"
    assert out["gpu_modules_loaded"] is False


def test_launcher_fails_when_a_rank_fails_and_ends_the_others():
    t0 = time.time()
    r = _run("fail", 3, {"GSI_BENCH_FAIL_GRACE_S": "2"})
    assert r.returncode == 3, (r.returncode, r.stderr[-2000:])
    assert r.stdout.strip() == ""
    assert time.time() - t0 < 60          # rank 0 was stuck for 600 s: the launcher ended it


def test_no_torch_in_bench():
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "import torch" not in src and "torch.distributed" not in src.replace("torch.distributed.run", "")
