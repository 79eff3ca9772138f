#!/usr/bin/env python3
"""bench.py -- randSVD throughput of the HIP path on MI355X.

One "step" = one `randsvd(A, K, p, q)` (RandMatFact.jl:83-90) with the operator A and the Gaussian
test matrix Omega already resident in HBM.  Workload at N = 1: BASELINE.json configs[1] -- dense
fp64 65536 x 65536 Gaussian covariance (256 x 256 unit grid, ell = 16), K = 128, p = 32 (l = 160),
q = 2.  For N > 1 the matrix is row-sharded over the ranks, RCCL all-reduce / all-gather between the
passes.  Default `--scaling weak`: every GPU keeps the SAME 34 GB row shard as at
N = 1 (n grows as sqrt(N): 256 x round(256 sqrt(N)) grid, per-GPU contraction work fixed) -- multi-GPU exists
here to factor covariances that do not fit one GPU.  `--scaling strong` shards the N = 1 matrix instead.

metric value   = algorithmic GB/s of the whole job: (2q+2) * (8 n^2 + 16 n l) bytes / step time
roofline       = the dominant kernel (the fp64 MFMA contraction A*X / A'*X): 2 n^2 l flop per
                 launch / its average launch duration, HIP events on the library's stream inside
                 the timed region; peak = 78.6 TFLOP/s dense fp64 MFMA
cpu_baseline   = the numpy/scipy oracle (same LAPACK/BLAS call sequence as the Julia reference) on
                 a bounded sample of the same workload (n = 50176), all host cores; the same sample
                 gives `sv_rel_err` (top-K singular values, GPU vs oracle, same Omega)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

PEAK_FP64_MFMA_TFLOPS = 78.6
PEAK_HBM_GBS = 8000.0


def algorithmic_bytes(n, l, q):
    P = 2 * q + 2
    return P * (8.0 * n * n + 16.0 * n * l)


def cpu_baseline_and_parity(gsi, ctx, K, p, q, grid=224, ell=14.0):
    """Oracle timed on the host on a bounded sample; GPU result on the same inputs for parity."""
    import numpy as np
    from oracle import oracle as orc
    from helpers import rel_sv_err
    n, l = grid * grid, K + p
    gx = np.repeat(np.arange(grid, dtype=np.float64), grid)          # point i = (i // grid, i % grid)
    gy = 1.3 * np.tile(np.arange(grid, dtype=np.float64), grid)      # anisotropic spacing: no x<->y degenerate pairs
    A = (gx[:, None] - gx[None, :]) ** 2
    A += (gy[:, None] - gy[None, :]) ** 2
    A *= -1.0 / (2.0 * ell * ell)
    np.exp(A, out=A)
    rng = np.random.default_rng(0)
    Omega = rng.standard_normal((n, l))
    t0 = time.perf_counter()
    Zref, Sref, _ = orc.randsvd_full(A, K, p, q, Omega)
    t_cpu = time.perf_counter() - t0
    Z, S = gsi.randsvd(A, K, p, q, Omega=Omega, return_S=True, ctx=ctx)
    err = rel_sv_err(S, Sref, K)
    xerr = orc.xis_error_up_to_sign(Z, Zref, K)
    del A
    try:
        from threadpoolctl import threadpool_info
        cores = max([i.get("num_threads", 1) for i in threadpool_info()] + [1])
    except Exception:
        cores = os.cpu_count() or 1
    return {
        "value": algorithmic_bytes(n, l, q) / t_cpu / 1e9, "unit": "GB/s", "cores": int(cores), "kind": "port",
        "sample": f"same workload at n={n} ({grid}x{grid} grid, y spacing 1.3, ell={ell}), K={K}, p={p}, q={q}: "
                  f"numpy/scipy oracle (dgemm/dgetrf/dgeqp3/dgesdd, OpenBLAS) {t_cpu:.2f} s",
        "seconds": t_cpu,
    }, err, xerr


def main():
    # stdout carries exactly ONE line (the JSON result of rank 0): libraries that chat on fd 1 (gloo's
    # "[Gloo] Rank 0 is connected ...", RCCL's version banner) are sent to stderr for the whole run.
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--grid", type=int, default=256, help="nx = ny of the unit grid (n = grid^2)")
    ap.add_argument("--ell", type=float, default=16.0)
    ap.add_argument("--K", type=int, default=128)
    ap.add_argument("--p", type=int, default=32)
    ap.add_argument("--q", type=int, default=2)
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
    import gsi_amd as gsi

    dist = None
    use_dist = world > 1 or bool(os.environ.get("GSI_BENCH_FORCE_DIST"))   # the latter: rehearse the N > 1 code on one GPU
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")     # one node: no hostname resolution for the rendezvous group
        import torch
        import torch.distributed as dist
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)   # rendezvous / barrier only

    ctx = gsi.Context(local_rank)
    if use_dist:
        import torch
        ids = [ctx.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(ids, src=0)
        ctx.comm_init(world, rank, ids[0])                  # RCCL communicator over xGMI

    nx = args.grid
    ny = args.grid if args.scaling == "strong" else int(round(args.grid * world ** 0.5))   # n^2 / N fixed
    n = nx * ny
    K, p, q = args.K, args.p, args.q
    l = K + p
    op = gsi.gridcov_operator(ctx, nx, ny, args.ell, 0)                     # A resident in HBM
    Omega = gsi.DeviceMatrix(ctx, n, l).randn(1234)                          # Omega resident in HBM
    Z = gsi.DeviceMatrix(ctx, n, l)
    S = gsi.DeviceMatrix(ctx, l, 1)
    lib = ctx.lib

    def step():
        gsi._lib.check(lib.gsi_randsvd_dev(ctx.h, op.h, Omega.h, K, p, q, Z.h, S.h), lib)

    def barrier():
        ctx.sync()
        if dist is not None:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    ctx.profile(True)
    ctx.phase_reset()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    phases = ctx.phase_times()
    ctx.profile(False)
    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    ms_per_step = 1e3 * elapsed / args.steps
    value = algorithmic_bytes(n, l, q) * args.steps / elapsed / 1e9

    # dominant kernel: the fp64 MFMA contraction over the operator (this rank's row shard)
    row0, mloc = ctx.shard(n)
    g_ms = phases["gemm_n"][0] + phases["gemm_t"][0]
    g_cnt = phases["gemm_n"][1] + phases["gemm_t"][1]
    avg_ms = g_ms / max(g_cnt, 1)
    flops_per_launch = 2.0 * mloc * n * l
    achieved = flops_per_launch / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "gemm_traffic.json")
    if os.path.exists(tpath) and world == 1 and n == 65536 and l == 160:   # counters were collected on this shape
        try:
            traffic = json.load(open(tpath)).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    roofline = {"bound": "mfma", "achieved": achieved, "peak": PEAK_FP64_MFMA_TFLOPS, "unit": "TFLOP/s",
                "frac": achieved / PEAK_FP64_MFMA_TFLOPS, "traffic": traffic,
                "kernel": "gemm_f64_kernel<NT,TRANS_A,GEN> (v_mfma_f64_16x16x4_f64)",
                "avg_launch_ms": avg_ms, "launches": int(g_cnt),
                "hbm_frac_of_A_stream": (8.0 * mloc * n / (avg_ms * 1e-3) / 1e9 / PEAK_HBM_GBS) if avg_ms > 0 else 0.0}

    if rank == 0:
        out = {
            "metric": "randSVD GB/s + top-k singular-value rel-err", "value": value, "unit": "GB/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"dense fp64 {n}x{n} Gaussian covariance ({nx}x{ny} grid, "
                                   f"ell={args.ell}), K={K}, p={p}, q={q} (BASELINE.json configs[1]"
                                   + ("" if world == 1 else
                                      (", n scaled by sqrt(N): same 34 GB row shard per GPU" if args.scaling == "weak"
                                       else ", same matrix row-sharded")) + ")",
                       "n": n, "K": K, "p": p, "q": q, "parallelism": f"row-shard x{world}",
                       "operator_bytes_per_gpu": 8.0 * mloc * n},
            "roofline": roofline,
            "phases_ms_per_step": {k: v[0] / args.steps for k, v in phases.items()},
            "phase_launch_groups_per_step": {k: v[1] / args.steps for k, v in phases.items()},
            "path_counters": ctx.counters(),
        }
        if world == 1 and not args.no_cpu_baseline:
            cb, err, xerr = cpu_baseline_and_parity(gsi, ctx, K, p, q)
            out["cpu_baseline"] = cb
            out["sv_rel_err"] = err
            out["xis_err_up_to_sign"] = xerr
        else:
            out["cpu_baseline"] = None
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
