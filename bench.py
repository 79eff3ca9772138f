#!/usr/bin/env python3
"""bench.py -- randSVD throughput of the HIP path on MI355X, on the configuration BASELINE.json's metric names.

One "step" = one `randsvd(A, K, p, q)` (RandMatFact.jl:83-90) with the operator A and the Gaussian test matrix
Omega already resident in HBM.

Headline workload (N = 1, BASELINE.json metric "n=1e6 rank=256"): n = 10^6 grid points,
A = LowRankCovMatrix over N_s = 1024 synthetic sample fields (8.2 GB of mean-removed samples in HBM, generated on
the device), K = 256, p = 64 (l = 320), q = 2.  Every product A*X is two MFMA contractions S (S'X)/(N_s-1).
For --gpus N > 1 the SAME problem is row-sharded over the ranks (`--scaling strong`, the default: north_star's
">= 6x at 8 GPUs" is a strong-scaling statement); `--scaling weak` gives every rank 10^6 rows instead.
`python3 bench.py --gpus N` starts its own N rank processes (fresh children, spawned before this process touches the
GPU; one per device) and relays rank 0's line; under torch.distributed.run (RANK / WORLD_SIZE in the environment) it
is one of the ranks.  Either way the ranks find each other through a directory (the communicator id as a file) and
every barrier / reduction of the harness goes through the library's own communicator (gsi_ctx_host_allgather):
no torch, no second communication layer.  The N > 1 line also carries which LU form ran (path_counters.lu_form, the
self-test mask of the in-kernel pivot exchange, collectives per step, ranks seen) and ONE row-sharded step of the
n = 10^6 implicit dense covariance -- the operator north_star's ">= 6x at 8 GPUs" is about.

metric value = algorithmic GB/s of the whole job: [(2q+2) products x (16 n N_s + 16 n l) + (2q + 2) panel
               factorizations x 16 n l] bytes / step time        
roofline     = the dominant kernel, the fp64 MFMA contraction gemm_f64_kernel: 2 n N_s l flop per launch over its
               average launch duration (HIP events on the library's stream around every operator contraction inside
               the timed region), against the 78.6 TFLOP/s dense fp64 MFMA peak; the same launches against the HBM
               peak ("hbm"); `traffic` = HBM bytes per launch from the rocprofv3 PMC passes of this very command
               (tools/pmc_bench_traffic.sh -> profiles/r05_bench_traffic.json, older rounds' files behind it; keyed on a
               hash over all of csrc/ + the public header: null when collected on a different build)
phases_hbm   = the HBM-bound panel phases (LU, QR, Z = Q_W U): one read + one write of the n x l panel per
               factorization over the measured time per factorization, against 8 TB/s
secondary    = BASELINE.json configs[1] (dense fp64 65536^2, K = 128, q = 2: the MFMA-bound stored operator), the
               matrix-free FFT covariance of a 1000^2 grid (configs[2] at n = 1e6: the HBM-bound operator) and ONE
               step of the n = 10^6 implicit dense exponential covariance (entries generated in the contraction kernel)
cpu_baseline = the numpy/scipy oracle (the reference's algorithm: N_s rank-1 ger!/gemv sweeps per product,
               dgetrf/dgeqp3/dgesdd panels) on a bounded sample of the same operator class, all host cores; the
               same sample gives `sv_rel_err` (top-K singular values, GPU vs oracle, same Omega)
"""
import argparse
import hashlib
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

PEAK_FP64_MFMA_TFLOPS = 78.6
PEAK_HBM_GBS = 8000.0
TRAFFIC_FILES = ["r05_bench_traffic.json", "r04_bench_traffic.json", "r03_bench_traffic.json", "r02_bench_traffic.json"]   # newest first; replayed only on a hash match


def kernel_source_hash():
    """sha256 over EVERY file of csrc/ (kernels, backend, pipeline, ABI) + the public header: any change to what is
    launched, or how, invalidates the replayed PMC traffic figures."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "geostatinversion.jl_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if os.path.isfile(os.path.join(d, f)):
            h.update(f.encode())
            with open(os.path.join(d, f), "rb") as fh:
                h.update(fh.read())
    with open(os.path.join(ROOT, "include", "gsi_hip.h"), "rb") as fh:
        h.update(fh.read())
    return h.hexdigest()[:16]


def lrcm_bytes(n, Ns, l, q):
    P = 2 * q + 2
    return P * (16.0 * n * Ns + 16.0 * n * l) + (2 * q + 2) * 16.0 * n * l


def dense_bytes(n, l, q):
    P = 2 * q + 2
    return P * (8.0 * n * n + 16.0 * n * l)


def fft_pair_bytes(Ns, Ms):
    """Bytes the 2d - 1 passes of the FFT operator must move per column PAIR (the zero padding is neither
    stored nor read): a forward pass along axis a < d-1 reads prod_{b<a} M_b * N_a * prod_{b>a} N_b complex values and
    writes the same with M_a, the inverse passes mirror that; the fused last-axis pass reads and writes
    prod_{b<d-1} M_b * N_{d-1}; plus 8 B per embedded point for the spectrum."""
    units, d = 0.0, len(Ns)
    for ax in range(d - 1):
        lo, hi = 1.0, 1.0
        for b in range(ax):
            lo *= Ms[b]
        for b in range(ax + 1, d):
            hi *= Ns[b]
        units += 2.0 * lo * (Ns[ax] + Ms[ax]) * hi
    lo = 1.0
    for b in range(d - 1):
        lo *= Ms[b]
    units += 2.0 * lo * Ns[-1]
    M = 1.0
    for m in Ms:
        M *= m
    return 16.0 * units + 8.0 * M


def phases_over_ranks(ctx, phases, steps):
    """max / min over the ranks of every phase's ms per step (collective: every rank calls it)."""
    import numpy as np
    names = sorted(phases)
    allp = ctx.host_allgather([phases[k][0] / max(steps, 1) for k in names])
    return {k: {"max": float(allp[:, i].max()), "min": float(allp[:, i].min())} for i, k in enumerate(names)}


def run_steps(gsi, ctx, op, n, K, p, q, steps, warmup, barrier, seed=1234, keep=None, rows=False, profile=1):
    """warmup + timed randsvd steps on device-resident inputs; returns (elapsed_s, phases, counters).  profile = 2 (several
    ranks): skew barriers in front of every collective / panel LU, so that waiting for peers is `comm_wait`, not LU time."""
    l = K + p
    lib = ctx.lib
    rank, world = ctx.rank()
    if rows:
        # several ranks: Omega and Z as ROW SHARDS (gsi_randsvd_rows) -- no GPU ever holds an n x l panel of them, and the
        # result is not all-gathered (what a multi-GPU consumer wants: a row-sharded xi-basis)
        _, nloc = ctx.shard(n)
        Omega = gsi.DeviceMatrix(ctx, nloc, l).randn(seed + 7919 * rank)
        Z = gsi.DeviceMatrix(ctx, nloc, l)
    else:
        Omega = gsi.DeviceMatrix(ctx, n, l).randn(seed)
        Z = gsi.DeviceMatrix(ctx, n, l)
    S = gsi.DeviceMatrix(ctx, l, 1)

    def step():
        if rows:
            gsi._lib.check(lib.gsi_randsvd_rows(ctx.h, op.h, Omega.h, K, p, q, Z.h, S.h), lib)
        else:
            gsi._lib.check(lib.gsi_randsvd_dev(ctx.h, op.h, Omega.h, K, p, q, Z.h, S.h), lib)

    for _ in range(warmup):
        step()
    ctx.profile(profile)
    ctx.phase_reset()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    phases = ctx.phase_times()
    run_steps.last_path_info = ctx.path_info()      # LU forms / collectives of exactly the timed steps (reset by phase_reset)
    ctx.profile(False)
    Sh = S.to_host()[:, 0].copy()
    if keep is not None:                 # the caller compares this very step with the oracle (full-size parity leg)
        keep["Omega"], keep["Z"] = Omega, Z
        S.close()
        return elapsed, phases, Sh
    for m in (Omega, Z, S):
        m.close()
    return elapsed, phases, Sh


run_steps.last_path_info = None


def toolchain_info():
    """What compiled the kernels (the resource usage of the hot kernels is pinned per compiler version,
    profiles/isa_resources.json) and what runs them."""
    info = {}
    try:                                                    # written by build.py when it compiled the library: no compiler is started here
        with open(os.path.join(ROOT, "geostatinversion.jl_amd", "build", "hipcc_version.txt")) as f:
            info["hipcc_version"] = f.read().strip()
    except OSError:
        info["hipcc_version"] = "unrecorded (library built without build.py)"
    try:
        info["isa_resources"] = json.load(open(os.path.join(ROOT, "profiles", "isa_resources.json"))).get("hipcc_version")
    except Exception:                                       # noqa: BLE001
        info["isa_resources"] = None
    return info


def host_threads():
    try:
        from threadpoolctl import threadpool_info
        return int(max([i.get("num_threads", 1) for i in threadpool_info()] + [1]))
    except Exception:
        return int(os.cpu_count() or 1)


def cpu_baseline_and_parity(gsi, ctx, Ns, K, p, q, n_s, decay):
    """Oracle timed on the host on a bounded sample of the same operator class; GPU on the same inputs for parity."""
    import numpy as np
    from oracle import oracle as orc
    from helpers import rel_sv_err
    l = K + p
    rng = np.random.default_rng(0)
    S = rng.standard_normal((Ns, n_s)) * (np.arange(1, Ns + 1.0) ** -decay)[:, None]     # sample j = row j
    Omega = rng.standard_normal((n_s, l))
    A = orc.LowRankCovMatrix(S)
    t0 = time.perf_counter()
    Zref, Sref, _ = orc.randsvd_full(A, K, p, q, Omega)
    t_cpu = time.perf_counter() - t0
    lr = gsi.LowRankCovMatrix(S, ctx=ctx)
    Z, Sv = gsi.randsvd(lr, K, p, q, Omega=Omega, return_S=True)
    lr.close()
    err = rel_sv_err(Sv, Sref, K)
    xerr = orc.xis_error_up_to_sign(Z, Zref, K)
    return {
        "value": lrcm_bytes(n_s, Ns, l, q) / t_cpu / 1e9, "unit": "GB/s", "cores": host_threads(), "kind": "port",
        "sample": f"same operator class at n={n_s}: LowRankCovMatrix over {Ns} samples (decay {decay}), K={K}, p={p}, "
                  f"q={q}; numpy/scipy oracle = the reference's algorithm (lowrank.jl:115-121: {Ns} rank-1 "
                  f"gemv/ger sweeps per product; dgetrf/dgeqp3/dgesdd panels, OpenBLAS) {t_cpu:.2f} s",
        "seconds": t_cpu,
    }, err, xerr


def secondary_c1(gsi, ctx, barrier):
    """BASELINE.json configs[0] -- the reference's own CPU-runnable case: dense 2000 x 2000 Gaussian covariance (50 x 40 grid,
    ell = 5), K = 32, p = 16, q = 1 (test/testrmf.jl path).  HIP (matrix and Omega resident in HBM, median of 50 steps)
    beside the oracle on this box's host cores, and the distance between the two on the same Omega."""
    import numpy as np
    from oracle import oracle as orc
    from helpers import gaussian_cov, rel_sv_err
    nx, ny, K, p, q = 50, 40, 32, 16, 1
    n, l = nx * ny, K + p
    A = gaussian_cov(nx, ny, 5.0)
    Om = np.asfortranarray(np.random.default_rng(1).standard_normal((n, l)))
    op = gsi.dense_operator(ctx, A)
    Omd = gsi.DeviceMatrix.from_host(ctx, Om)
    Z = gsi.DeviceMatrix(ctx, n, l)
    S = gsi.DeviceMatrix(ctx, l, 1)
    step = lambda: gsi._lib.check(ctx.lib.gsi_randsvd_dev(ctx.h, op.h, Omd.h, K, p, q, Z.h, S.h), ctx.lib)
    for _ in range(3):
        step()
    ts = []
    for _ in range(50):
        barrier()
        t0 = time.perf_counter()
        step()
        barrier()
        ts.append(time.perf_counter() - t0)
    Zh, Sh = Z.to_host(), S.to_host()[:, 0].copy()
    for m in (Omd, Z, S, op):
        m.close()
    tc = []
    for _ in range(5):
        t0 = time.perf_counter()
        Zr, Sr, _ = orc.randsvd_full(A, K, p, q, Om)
        tc.append(time.perf_counter() - t0)
    return {"workload": f"dense fp64 {n}x{n} Gaussian covariance (50x40 grid, ell=5), K={K}, p={p}, q={q} (BASELINE.json configs[0])",
            "ms_per_step": 1e3 * sorted(ts)[len(ts) // 2], "GB/s": dense_bytes(n, l, q) / sorted(ts)[len(ts) // 2] / 1e9,
            "oracle_ms_per_step": 1e3 * sorted(tc)[len(tc) // 2], "oracle_threads": host_threads(),
            "sv_rel_err": rel_sv_err(Sh, Sr, K), "xis_err_up_to_sign": orc.xis_error_up_to_sign(Zh, Zr, K)}


def secondary_headline_p0(gsi, ctx, barrier, n, Ns, K, q, decay):
    """The headline operator with NO oversampling (p = 0, l = K = 256: "rank=256" read literally; SURVEY.md 8 pins p = 64 for
    the headline and asks for this one beside it)."""
    op = gsi.lowrank_synthetic_operator(ctx, n, Ns, seed=0, decay=decay)
    e, ph, _ = run_steps(gsi, ctx, op, n, K, 0, q, 5, 1, barrier)
    op.close()
    return {"workload": f"the headline LowRankCovMatrix (n={n}, N_s={Ns}), K={K}, p=0 (l={K}), q={q}", "steps": 5,
            "ms_per_step": 1e3 * e / 5, "GB/s": lrcm_bytes(n, Ns, K, q) * 5 / e / 1e9,
            "phases_ms_per_step": {k: v[0] / 5 for k, v in ph.items()}}


def secondary_c2(gsi, ctx, barrier):
    """BASELINE.json configs[1]: dense fp64 65536^2 Gaussian covariance, K = 128, p = 32, q = 2."""
    g, K2, p2, q2 = 256, 128, 32, 2
    n2, l2 = g * g, K2 + p2
    op2 = gsi.gridcov_operator(ctx, g, g, 16.0, 0)
    e2, ph2, _ = run_steps(gsi, ctx, op2, n2, K2, p2, q2, 5, 1, barrier)
    op2.close()
    gm = ph2["gemm_n"][0] + ph2["gemm_t"][0]
    gc = ph2["gemm_n"][1] + ph2["gemm_t"][1]
    tf = 2.0 * n2 * n2 * l2 / (gm / gc * 1e-3) / 1e12
    return {
        "workload": f"dense fp64 {n2}x{n2} Gaussian covariance (256x256 grid, ell=16), K={K2}, p={p2}, q={q2} "
                    "(BASELINE.json configs[1])",
        "steps": 5, "ms_per_step": 1e3 * e2 / 5, "GB/s": dense_bytes(n2, l2, q2) * 5 / e2 / 1e9,
        "gemm_TFLOP/s": tf, "gemm_frac_of_mfma_peak": tf / PEAK_FP64_MFMA_TFLOPS,
        "hbm_frac_of_A_stream": 8.0 * n2 * n2 / (gm / gc * 1e-3) / 1e9 / PEAK_HBM_GBS,
        "phases_ms_per_step": {k: v[0] / 5 for k, v in ph2.items()}}


def one_gpu_implicit_reference():
    """The 1-GPU time of the implicit 10^6 step as the driver last recorded it (BENCH_rNN.json of an earlier round, or the
    builder's committed line): carried beside the N-rank step, never re-measured inside an N > 1 run (61 s of one GPU)."""
    cands = sorted([f for f in os.listdir(ROOT) if f.startswith("BENCH_r") and f.endswith(".json")], reverse=True)
    cands = [os.path.join(ROOT, f) for f in cands] + [os.path.join(ROOT, "profiles", f) for f in ("r04_bench_line.json", "r03_bench_line.json")]
    for path in cands:
        try:
            d = json.load(open(path))
            d = d.get("parsed", d)
            if d.get("n_gpus", 1) != 1:
                continue
            ms = d["secondary"]["implicit_dense_1e6"]["ms_per_step"]
            return {"ms_per_step": float(ms), "kernel_source_hash": d.get("kernel_source_hash"),
                    "source": os.path.relpath(path, ROOT) + " (recorded 1-GPU line, not re-measured in this run)"}
        except Exception:                                   # noqa: BLE001
            continue
    return None


def secondary_implicit(gsi, ctx, barrier, rows=False, max_over_ranks=None):
    """n = 1e6 dense covariance, never stored (north_star "10^6 x 10^6-implicit"): ONE step, no warm-up.  rows=True (N > 1):
    the operator row-sharded over the ranks, gsi_randsvd_rows (Omega gathered for the products, panel LUs replicated on the
    gathered panels, TSQR, Z as row shards) -- the step north_star's ">= 6x at 8 GPUs" is about."""
    import numpy as np
    gi, K3, p3, q3 = 1000, 256, 64, 2
    n3, l3 = gi * gi, K3 + p3
    op3 = gsi.gridcov_implicit_operator(ctx, gi, gi, 100.0, kind=1)      # exponential kernel, ell = 100
    keep = {} if rows else None
    # N > 1: profile level 2 -- ~20 one-double all-reduces in a step of seconds, and the phases say what was work and what was
    # waiting for the slowest rank
    e3, ph3, S3 = run_steps(gsi, ctx, op3, n3, K3, p3, q3, 1, 0, barrier, keep=keep, rows=rows, profile=2 if rows else 1)
    pinfo = run_steps.last_path_info
    op3.close()
    extra = {}
    if rows:
        e3 = max_over_ranks(e3)
        # size-independent check of the sharded result: Z = V sqrt(S) has |Z[:, i]|^2 = S_i, columns beyond K zero -- the column
        # norms summed over the ranks' row blocks against the replicated singular values
        Zl = keep["Z"].to_host()
        keep["Omega"].close()
        keep["Z"].close()
        part = np.concatenate([(Zl[:, :K3] ** 2).sum(axis=0), [float(np.abs(Zl[:, K3:]).max()) if Zl.shape[0] else 0.0]])
        tot = ctx.host_allgather(part)
        col2 = tot[:, :K3].sum(axis=0)
        world = ctx.rank()[1]
        extra = {"n_gpus": world, "parallelism": f"row-shard x{world} (gsi_randsvd_rows)",
                 "ZtZ_diag_vs_S_max_rel": float(np.max(np.abs(col2 - S3[:K3]) / S3[:K3])),
                 "trailing_p_columns_max_abs": float(tot[:, K3].max()),
                 "sv_descending_positive": bool(all(S3[i] >= S3[i + 1] for i in range(l3 - 1)) and S3[K3 - 1] > 0),
                 "lu_form": pinfo["lu_form"], "collectives_per_step": pinfo["collectives"],
                 "phases_ms_max_min_over_ranks": phases_over_ranks(ctx, ph3, 1),
                 "phases_note": "profile level 2: comm_wait = arrival skew in front of collectives / panel LUs (one-double all-reduces)",
                 "one_gpu_reference": one_gpu_implicit_reference()}
        ref = extra["one_gpu_reference"]
        # a ratio only between runs of the SAME kernels: the recorded line must carry this build's source hash
        if ref and ref.get("kernel_source_hash") == kernel_source_hash():
            extra["speedup_vs_recorded_one_gpu"] = ref["ms_per_step"] / (1e3 * e3)
        else:
            extra["speedup_vs_recorded_one_gpu"] = None
            extra["speedup_note"] = ("no recorded 1-GPU line" if not ref else
                                     "the recorded 1-GPU line is from another build (kernel_source_hash %s, this build %s): "
                                     "divide by a 1-GPU run of THIS build" % (ref.get("kernel_source_hash"), kernel_source_hash()))
    gm = ph3["gemm_n"][0] + ph3["gemm_t"][0]
    gc = ph3["gemm_n"][1] + ph3["gemm_t"][1]
    nloc3 = ctx.shard(n3)[1]
    tf = 2.0 * nloc3 * n3 * l3 / (gm / gc * 1e-3) / 1e12              # this rank's contraction launches (its rows of A)
    return {
        "workload": f"implicit dense fp64 {n3}x{n3} exponential grid covariance exp(-d/100) (1000x1000 grid; 8 TB if "
                    f"stored), K={K3}, p={p3}, q={q3}: entries generated inside the contraction kernel from an 8 MB "
                    "table of the kernel over grid offsets",
        "steps": 1, "ms_per_step": 1e3 * e3, "equivalent_stored_GB/s": dense_bytes(n3, l3, q3) / e3 / 1e9,
        "gemm_TFLOP/s": tf, "gemm_frac_of_mfma_peak": tf / PEAK_FP64_MFMA_TFLOPS,
        "phases_ms_per_step": {k: v[0] for k, v in ph3.items()}, **extra}


def secondary_fft_1000sq(gsi, ctx, barrier):
    """BASELINE.json configs[2] (C3) at the size whose rank-256 panels fit one GPU: matrix-free FFT power-law covariance of
    a 1000 x 1000 grid (n = 1e6), K = 205, p = 51 (l = 256), q = 2.  HBM-bound operator."""
    import numpy as np
    gf, K4, p4, q4 = 1000, 205, 51, 2
    n4, l4 = gf * gf, K4 + p4
    op4 = gsi.fft_powerlaw_operator(ctx, [gf, gf], -3.5)
    e4, ph4, _ = run_steps(gsi, ctx, op4, n4, K4, p4, q4, 5, 1, barrier)
    op4.close()
    Mf = 1 << int(np.ceil(np.log2(2 * gf)))
    pair_bytes = fft_pair_bytes([gf, gf], [Mf, Mf])
    prod_ms = (ph4["gemm_n"][0] + ph4["gemm_t"][0]) / (ph4["gemm_n"][1] + ph4["gemm_t"][1])
    return {
        "workload": f"matrix-free FFT power-law covariance (beta = -3.5) of a {gf}x{gf} grid, embedding {Mf}x{Mf}, "
                    f"K={K4}, p={p4}, q={q4} (BASELINE.json configs[2] at n = 1e6)",
        "steps": 5, "ms_per_step": 1e3 * e4 / 5, "ms_per_product": prod_ms,
        "product_algorithmic_GB/s": (l4 // 2) * pair_bytes / (prod_ms * 1e-3) / 1e9,
        "product_frac_of_hbm_peak": (l4 // 2) * pair_bytes / (prod_ms * 1e-3) / 1e9 / PEAK_HBM_GBS,
        "phases_ms_per_step": {k: v[0] / 5 for k, v in ph4.items()}}


def secondary_pointcov(gsi, ctx, barrier):
    """The covariance of SCATTERED points as a row-streamed implicit operator (SURVEY.md 8b "coords + kernel id + params"):
    one product A*X at n = 202500 points (the nodes of a 450 x 450 grid handed over as coordinates), l = 320, exponential
    kernel -- against the table-based implicit grid operator on the same points (the same matrix: their difference is the
    check)."""
    import numpy as np
    g, l = 450, 320
    n = g * g
    xs, ys = np.meshgrid(np.arange(g, dtype=np.float64), np.arange(g, dtype=np.float64), indexing="ij")
    P = np.stack([xs.ravel(), ys.ravel()])
    lib = ctx.lib
    X = gsi.DeviceMatrix(ctx, n, l).randn(1)
    Y = gsi.DeviceMatrix(ctx, n, l)
    out = {}
    cols = {}
    for tag, op in (("table", gsi.gridcov_implicit_operator(ctx, g, g, 45.0, kind=1)),
                    ("points", gsi.pointcov_implicit_operator(ctx, P, "exponential", ell=45.0))):
        gsi._lib.check(lib.gsi_op_mul_dev(ctx.h, op.h, 0, X.h, Y.h), lib)
        barrier()
        t0 = time.perf_counter()
        gsi._lib.check(lib.gsi_op_mul_dev(ctx.h, op.h, 0, X.h, Y.h), lib)
        barrier()
        out[tag] = time.perf_counter() - t0
        cols[tag] = Y.to_host()[:, :2]
        op.close()
    X.close()
    Y.close()
    return {
        "workload": f"implicit covariance exp(-d/45) of n = {n} SCATTERED points (given as coordinates), l = {l}: one product; every "
                    "entry generated once, inside the contraction's tile loader (96 x 320 output tiles, DESIGN.md 4.9)",
        "contraction_frac_of_mfma_peak": 2.0 * n * n * l / out["points"] / 1e12 / PEAK_FP64_MFMA_TFLOPS,
        "ms_per_product": 1e3 * out["points"], "contraction_TFLOP/s": 2.0 * n * n * l / out["points"] / 1e12,
        "table_based_grid_operator_ms_per_product": 1e3 * out["table"],
        "table_based_grid_operator_TFLOP/s": 2.0 * n * n * l / out["table"] / 1e12,
        "max_rel_diff_vs_table_operator": float(np.abs(cols["points"] - cols["table"]).max() / np.abs(cols["table"]).max())}


def secondary_pcgalsqr_c5(gsi, ctx, barrier):
    """BASELINE.json configs[4] on one GPU: a 10^6-parameter synthetic inversion by pcgalsqr (lsqr.jl:20-63) with the xi-basis
    resident in HBM -- K = 256 vectors from randsvd of a LowRankCovMatrix -- stored in fp64 and in fp32 ("fp32 mixed
    precision, tolerance vs fp64 reference"): two PCGA iterations each, forward model h(s) = (s .* x)[observed points]
    (test/testrpcga.jl:110-112) on the host, nobs = 4096.  The time is end-to-end, i.e. mostly the K + 3 = 259 forward runs
    per iteration and the 2 GB of perturbed fields that travel to the host for them (user code in the reference too)."""
    import numpy as np
    import scipy.sparse as sp
    n, Ns, K, p, q, nobs = 1000000, 256, 256, 64, 1, 4096
    op = gsi.lowrank_synthetic_operator(ctx, n, Ns, seed=3, decay=0.75)
    Om = gsi.DeviceMatrix(ctx, n, K + p).randn(9)
    Z = gsi.DeviceMatrix(ctx, n, K + p)
    barrier()
    t0 = time.perf_counter()
    gsi._lib.check(ctx.lib.gsi_randsvd_dev(ctx.h, op.h, Om.h, K, p, q, Z.h, None), ctx.lib)
    barrier()
    t_basis = time.perf_counter() - t0
    Om.close()
    op.close()
    b64 = gsi.DeviceBasis(Z, K)
    b32 = gsi.DeviceBasis(Z, K, precision=32)
    rng = np.random.default_rng(8)
    idx = np.arange(nobs) * (n // nobs) + 17
    xw = 1.0 + 0.1 * rng.standard_normal(n)
    forward = lambda sv: (sv * xw)[idx]
    X = np.full(n, 2.0)
    coef = rng.standard_normal(6) * 3.0
    truth = X + sum(c * b64[i] for i, c in enumerate(coef))
    noise = 1e-4
    y = forward(truth) + noise * rng.standard_normal(nobs)
    R = noise ** 2 * sp.identity(nobs, format="csc")
    out = {}
    sols = {}
    for tag, basis in (("fp64", b64), ("fp32", b32)):
        barrier()
        t0 = time.perf_counter()
        sols[tag] = gsi.pcgalsqr(forward, X.copy(), X, basis, R, y, maxiters=2)
        barrier()
        out[f"seconds_2_iterations_{tag}_basis"] = time.perf_counter() - t0
    mis0 = float(np.linalg.norm(forward(X) - y))
    res = {"workload": f"pcgalsqr, n = {n} parameters, K = {K} xi-vectors resident in HBM (randsvd of a LowRankCovMatrix over {Ns} samples, "
                       f"p = {p}, q = {q}), nobs = {nobs}, 2 iterations, host forward model (BASELINE.json configs[4] on one GPU)",
           "basis_randsvd_ms": 1e3 * t_basis,
           "fp32_vs_fp64_rel_diff": float(np.linalg.norm(sols["fp32"] - sols["fp64"]) / np.linalg.norm(sols["fp64"])),
           "fp32_vs_fp64_tolerance": 1e-5,
           "misfit_reduction_fp64": float(np.linalg.norm(forward(sols["fp64"]) - y)) / mis0,
           "misfit_reduction_fp32": float(np.linalg.norm(forward(sols["fp32"]) - y)) / mis0,
           "rel_error_vs_truth_fp64": float(np.linalg.norm(sols["fp64"] - truth) / np.linalg.norm(truth - X)),
           "basis_bytes_fp64": 8.0 * n * K, "basis_bytes_fp32": 4.0 * n * K}
    res.update(out)
    b32.close()
    b64.close()
    Z.close()
    return res


def secondary_fft_512cube(gsi, ctx, barrier):
    """BASELINE.json configs[2]'s own grid: 512^3 points (n = 1.34e8), FFTRF convention (512 is a power of two: exactly
    FFTRF.jl:83-90's 1024^3 embedding), at the sketch width one GPU's 288 GB hold (four n x l fp64 panels of 51 GB +
    spectrum + work array); rank 256 needs the panels spread over GPUs.  One warm-up step (a third of a
    first step is the allocation of the panels), ONE timed step."""
    gc3, K5, p5, q5 = 512, 39, 9, 2
    n5, l5 = gc3 ** 3, K5 + p5
    ctx.release_cache()      # 232 of 288 GB are about to be used: what the earlier workloads left cached goes back first (untimed)
    op5 = gsi.fft_powerlaw_operator(ctx, [gc3, gc3, gc3], -3.5, fftrf=True)
    e5, ph5, _ = run_steps(gsi, ctx, op5, n5, K5, p5, q5, 1, 1, barrier)      # one warm-up step: the 51 GB panels exist afterwards
    peak_bytes5 = ctx.device_bytes()
    op5.close()
    M5 = 2 * gc3
    pair5 = fft_pair_bytes([gc3] * 3, [M5] * 3)
    prod5 = (ph5["gemm_n"][0] + ph5["gemm_t"][0]) / (ph5["gemm_n"][1] + ph5["gemm_t"][1])
    return {
        "workload": f"matrix-free FFTRF-convention power-law covariance (beta = -3.5) of a {gc3}^3 grid "
                    f"(n = {n5}), embedding {M5}^3, K={K5}, p={p5} (l={l5}: what one GPU's HBM holds), q={q5} "
                    "(BASELINE.json configs[2]'s grid; rank 256 needs column-sharded panels over GPUs)",
        "steps": 1, "warmup": 1, "ms_per_step": 1e3 * e5, "ms_per_product": prod5,
        "product_algorithmic_GB/s": (l5 // 2) * pair5 / (prod5 * 1e-3) / 1e9,
        "product_frac_of_hbm_peak": (l5 // 2) * pair5 / (prod5 * 1e-3) / 1e9 / PEAK_HBM_GBS,
        "device_bytes_in_use_after_step": peak_bytes5,
        "phases_ms_per_step": {k: v[0] for k, v in ph5.items()}}


def _timed(fn, reps=1):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2]


def settle(ctx, seconds=6.0):
    """Give the context's cached device memory back and let the driver finish with it before anything is timed: after tens
    of GB are freed (232 GB by the 512^3 workload, 34 GB by C2's matrix) host<->device copies ran at half their rate for a
    few seconds -- the DMA engines that move our chunks were busy with the freed pages (28 against 53 GB/s on the headline's
    Omega / Z inside a full run, profiles/r05_boundary_after_free.log).  Untimed."""
    ctx.release_cache()
    ctx.sync()
    time.sleep(seconds)


def boundary_block(gsi, ctx, barrier, headline_ms, host, Ns, K, p, q, c1_step_ms, c2_step_ms, c1_oracle_ms=None):
    """"upload and Omega generation timed and reported separately".  Every figure of the line above is
    device-resident; what the reference's callers invoke hands over HOST memory -- getxis(Q::Matrix, numxis, p, q, seed)
    (GeostatInversion.jl:63-70), LowRankCovMatrix(samples) (lowrank.jl:14-30), randn(n, l) on the host (RandMatFact.jl:54).
    Timed here, around the host-pointer entry points the Julia shim binds, at C1, C2 and the headline: seconds, GB/s against
    this box's own pinned-copy rate (gsi_ctx_pinned_copy_rate, measured first) and against the 63 GB/s of PCIe Gen5 x16, and
    each as a multiple of its device-resident step.  Host arrays are fresh pageable numpy memory (what a Julia array is)."""
    import ctypes as C
    import numpy as np
    from helpers import gaussian_cov, rel_sv_err
    L = gsi._lib
    lib = ctx.lib
    out = {}
    h2d, d2h = C.c_double(), C.c_double()
    L.check(lib.gsi_ctx_pinned_copy_rate(ctx.h, 1 << 30, C.byref(h2d), C.byref(d2h)), lib)
    out["pinned_copy_GB/s"] = {"h2d": h2d.value, "d2h": d2h.value, "bytes": 1 << 30, "pcie_gen5_x16_GB/s": 63.0}
    pin = max(h2d.value, 1e-9)

    def rate(nbytes, secs, ceiling=pin):
        return {"seconds": secs, "GB/s": nbytes / secs / 1e9, "frac_of_pinned_copy_rate": nbytes / secs / 1e9 / ceiling,
                "frac_of_pcie_gen5_x16": nbytes / secs / 1e9 / 63.0, "bytes": float(nbytes)}

    outs = {}

    def out_arrays(n_, l_):
        # result arrays that exist already (written once).  A FRESH array (what the Julia shim's Matrix{Float64}(undef, n, l)
        # is) costs the staged download 2-5 ms more: the workers' first-touch faults hide under their DMA waits
        # (tools/prefault_probe.py, profiles/r05_prefault_probe.log); reported below as "fresh_output_array", with the host's own
        # cost of creating and unmapping such an array beside it -- which an earlier version of this block timed as the call's.
        if (n_, l_) not in outs:
            outs[(n_, l_)] = (np.zeros((n_, l_), order="F"), np.zeros(l_))
        return outs[(n_, l_)]

    def host_randsvd(op, Om, K_, p_, q_, fresh=False):
        n_, l_ = Om.shape
        Z, S = (np.empty((n_, l_), order="F"), np.empty(l_)) if fresh else out_arrays(n_, l_)
        L.check(lib.gsi_randsvd(ctx.h, op.h, L.dptr(Om), K_, p_, q_, L.dptr(Z), S.ctypes.data_as(L.c_dp)), lib)
        return Z, S

    def dense_host(A, Om, K_, p_, q_):
        m_, n_ = A.shape
        l_ = K_ + p_
        Z, S = out_arrays(n_, l_)
        L.check(lib.gsi_randsvd_dense_host(ctx.h, L.dptr(A), m_, n_, m_, L.dptr(Om), K_, p_, q_, L.dptr(Z),
                                           S.ctypes.data_as(L.c_dp), None), lib)
        return Z, S

    # ---- C1: 2000 x 2000 (32 MB), K = 32, p = 16, q = 1 ------------------------------------------------------------
    A1 = np.asfortranarray(gaussian_cov(50, 40, 5.0))
    Om1 = np.asfortranarray(np.random.default_rng(1).standard_normal((2000, 48)))
    ops = []
    t_up = _timed(lambda: ops.append(gsi.dense_operator(ctx, A1)), reps=9)
    t_rs = _timed(lambda: host_randsvd(ops[0], Om1, 32, 16, 1), reps=21)
    t_all = _timed(lambda: dense_host(A1, Om1, 32, 16, 1), reps=21)
    for o in ops:
        o.close()
    out["c1_dense_2000"] = {
        "gsi_op_dense_upload": rate(A1.nbytes, t_up),
        "gsi_randsvd_host_Omega_in_Z_out_ms": 1e3 * t_rs,
        "gsi_randsvd_dense_host_ms": 1e3 * t_all,
        "what": "gsi_randsvd_dense_host = what getxis(Q::Matrix, ...) costs per call: matrix up, Omega up, randsvd, Z and S down",
        "device_resident_step_ms": c1_step_ms,
        "host_api_over_device_resident": (1e3 * t_all / c1_step_ms) if c1_step_ms else None,
        "oracle_ms_per_step": c1_oracle_ms, "oracle_note": "the numpy/scipy oracle on this box's host cores, same matrix (secondary.c1_dense_2000)"}

    # ---- C2: 65536 x 65536 (34.4 GB), K = 128, p = 32, q = 2 ---------------------------------------------------------
    # the Gaussian covariance of the 256 x 256 grid is a Kronecker product: built on the host in seconds
    g, K2, p2, q2 = 256, 128, 32, 2
    n2, l2 = g * g, K2 + p2
    d = np.arange(g, dtype=np.float64)
    k1 = np.exp(-((d[:, None] - d[None, :]) ** 2) / (2.0 * 16.0 ** 2))
    t0 = time.perf_counter()
    A2 = np.kron(k1, k1).T                               # symmetric: the transposed view is the column-major matrix, no copy
    t_build = time.perf_counter() - t0
    Om2 = np.asfortranarray(np.random.default_rng(2).standard_normal((n2, l2)))
    settle(ctx)
    ops = []
    t_up2 = _timed(lambda: ops.append(gsi.dense_operator(ctx, A2)))
    Z2r, S2r = host_randsvd(ops[0], Om2, K2, p2, q2)     # warm (workspaces exist afterwards)
    Z2r, S2r = Z2r.copy(), S2r.copy()
    t_rs2 = _timed(lambda: host_randsvd(ops[0], Om2, K2, p2, q2), reps=3)
    ops[0].close()
    t_all2 = []
    for _ in range(2):
        t0 = time.perf_counter()
        Z2, S2 = dense_host(A2, Om2, K2, p2, q2)
        t_all2.append(time.perf_counter() - t0)
    t_all2 = min(t_all2)
    out["c2_dense_65536"] = {
        "host_matrix_build_seconds": t_build,
        "gsi_op_dense_upload": rate(A2.nbytes, t_up2),
        "gsi_randsvd_host_Omega_in_Z_out_ms": 1e3 * t_rs2,
        "gsi_randsvd_dense_host": dict(rate(A2.nbytes + 2 * Om2.nbytes, t_all2), what="matrix and Omega up in row blocks, the "
                                       "sketch A*Omega under the upload (RandMatFact.jl:55), power iterations, Z and S down"),
        "upload_then_randsvd_seconds": t_up2 + t_rs2,
        "overlap_gain_ms": 1e3 * (t_up2 + t_rs2 - t_all2),
        "overlapped_equals_resident_bitwise": bool(np.array_equal(Z2, Z2r) and np.array_equal(S2, S2r)),
        "device_resident_step_ms": c2_step_ms,
        "host_api_over_device_resident": (1e3 * t_all2 / c2_step_ms) if c2_step_ms else None}
    del A2, Z2, Z2r
    settle(ctx)

    # ---- headline: LowRankCovMatrix(samples) with 8.2 GB of samples, Omega 2.56 GB in, Z 2.56 GB out -------------------
    if host is not None and "samples" in host:
        Sh = host["samples"]                               # N_s x n, one sample per row = n x N_s column-major
        n = Sh.shape[1]
        l = K + p
        hs = []

        def up():
            h = C.c_void_p()
            L.check(lib.gsi_op_lowrank(ctx.h, C.byref(h), Sh.ctypes.data_as(L.c_dp), n, Ns, n, 1, 0, n), lib)
            hs.append(gsi.Operator(ctx, h))
        t_ups = _timed(up)
        Om = host["Omega"]
        Zh, Svh = host_randsvd(hs[0], Om, K, p, q)         # warm
        Svh = Svh.copy()
        t_rsh = _timed(lambda: host_randsvd(hs[0], Om, K, p, q), reps=3)
        t0 = time.perf_counter()
        Zf, Sf = np.empty((n, l), order="F"), np.empty(l)          # fresh pages, never written
        t_alloc = time.perf_counter() - t0
        t_fresh = _timed(lambda: L.check(lib.gsi_randsvd(ctx.h, hs[0].h, L.dptr(Om), K, p, q, L.dptr(Zf), Sf.ctypes.data_as(L.c_dp)), lib))
        fresh_same = bool(np.array_equal(Zf, Zh))
        t0 = time.perf_counter()
        del Zf
        t_free = time.perf_counter() - t0
        back = np.zeros_like(Sh)
        t_dn = _timed(lambda: L.check(lib.gsi_op_lowrank_samples(ctx.h, hs[0].h, back.ctypes.data_as(L.c_dp), n), lib))
        hs[0].close()
        out["headline_lowrank_1e6"] = {
            "gsi_op_lowrank_upload_and_centre": rate(Sh.nbytes, t_ups),
            "gsi_op_lowrank_samples_download": rate(Sh.nbytes, t_dn, max(d2h.value, 1e-9)),
            "gsi_randsvd_host_Omega_in_Z_out_ms": 1e3 * t_rsh,
            "gsi_randsvd_host_fresh_output_array_ms": 1e3 * t_fresh,
            "fresh_output_equals_reused_output_bitwise": fresh_same,
            "host_side_np_empty_ms": 1e3 * t_alloc, "host_side_unmap_of_a_written_result_ms": 1e3 * t_free,
            "host_transfer_bytes_per_call": float(2 * Om.nbytes),
            "device_resident_step_ms": headline_ms,
            "host_api_over_device_resident": 1e3 * t_rsh / headline_ms,
            "transfers_ms": 1e3 * t_rsh - headline_ms,
            "transfers_GB/s": 2 * Om.nbytes / max(t_rsh - headline_ms * 1e-3, 1e-9) / 1e9,
            "sv_rel_err_vs_device_resident_step": rel_sv_err(Svh, host["S"], K) if "S" in host else None}
    return out


def full_size_parity(host, Ns, K, p, q, Sv):
    """The metric's rel-err AT the metric's size: the oracle (RandMatFact.jl:83-90 over lowrank.jl's operator, dgetrf /
    dgeqp3 / dgesdd panels) on the very operator, Omega and step the HIP path was timed on -- the centred samples, Omega
    and Z of the last timed step downloaded from HBM.  The oracle's products run in GEMM form S'(S X)/(N-1) here (the
    ger! loop of lowrank.jl:115-121 would take hours at n = 1e6; rounding-order difference only); the n = 16384 sample
    of cpu_baseline keeps the reference's loop."""
    from oracle import oracle as orc
    from helpers import rel_sv_err
    n = host["Omega"].shape[0]
    t0 = time.perf_counter()
    A = orc.LowRankCovMatrix(host.pop("samples"), gemm_form=True)
    Zref, Sref, _ = orc.randsvd_full(A, K, p, q, host["Omega"])
    t_cpu = time.perf_counter() - t0
    err = rel_sv_err(Sv, Sref, K)
    xerr = orc.xis_error_up_to_sign(host["Z"], Zref, K)
    tail_zero = bool((host["Z"][:, K:] == 0.0).all())
    return {"n": int(n), "samples": int(Ns), "K": K, "p": p, "q": q, "sv_rel_err": err, "xis_err_up_to_sign": xerr,
            "trailing_p_columns_zero": tail_zero, "oracle_seconds": t_cpu, "oracle_threads": host_threads(),
            "oracle_products": "gemm form S'(S X)/(N-1) (rounding-order difference from lowrank.jl:115-121's ger! loop)",
            "inputs": "centred samples, Omega and the last timed step's Z/S downloaded from HBM: the timed operator itself"}


def same_numbers_as_one_gpu(gsi, device, n, Ns, decay, K, p, q, world, seed, Sv, Z0_rows):
    """Rank 0, after the timed region of an N > 1 run: the SAME randsvd on ONE GPU (a second context without a communicator:
    whole operator, the global Omega = the ranks' row blocks stacked, gsi_randsvd's replicated code path) and the distance
    of the N-rank result from it.  The one-GPU path is the one the oracle checks at this size in the N = 1 line."""
    from oracle import oracle as orc
    from helpers import rel_sv_err
    import numpy as np
    l = K + p
    t0 = time.perf_counter()
    ctx1 = gsi.Context(device)
    pad = (n + world - 1) // world
    Om = np.empty((n, l), order="F")
    for r in range(world):                                   # the ranks' Omega blocks, regenerated from their seeds
        r0 = min(r * pad, n)
        nl = min(pad, n - r0)
        if nl > 0:
            blk = gsi.DeviceMatrix(ctx1, nl, l).randn(seed + 7919 * r)
            Om[r0:r0 + nl] = blk.to_host()
            blk.close()
    op1 = gsi.lowrank_synthetic_operator(ctx1, n, Ns, seed=0, decay=decay)
    Omega = gsi.DeviceMatrix.from_host(ctx1, Om)
    del Om
    Z = gsi.DeviceMatrix(ctx1, n, l)
    S = gsi.DeviceMatrix(ctx1, l, 1)
    gsi._lib.check(ctx1.lib.gsi_randsvd_dev(ctx1.h, op1.h, Omega.h, K, p, q, Z.h, S.h), ctx1.lib)
    S1 = S.to_host()[:, 0].copy()
    Z1 = Z.to_host()[:Z0_rows.shape[0]]
    for m in (Omega, Z, S, op1):
        m.close()
    ctx1.close()
    return {"n": int(n), "ranks": int(world), "sv_rel_err_vs_one_gpu": rel_sv_err(Sv, S1, K),
            "xis_err_up_to_sign_vs_one_gpu_rank0_rows": orc.xis_error_up_to_sign(Z0_rows, Z1, K),
            "seconds": time.perf_counter() - t0,
            "what": "rank 0 re-runs the step on one GPU (no communicator) from the same samples and the stacked Omega blocks"}


def spawn_ranks(n):
    """`python3 bench.py --gpus N` without a launcher: THIS process becomes the launcher.  It never touches the GPU (no HIP
    call, the product library is not even loaded here) -- N fresh children are started with RANK / LOCAL_RANK / WORLD_SIZE
    and a rendezvous directory in their environment, rank 0's stdout (the one JSON line) is relayed, and the exit status is
    non-zero if any child's is."""
    base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
    rdv = tempfile.mkdtemp(prefix="gsi-bench-", dir=base)
    procs = []
    try:
        for r in range(n):
            env = dict(os.environ)
            env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "GSI_BENCH_RDV": rdv})
            env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                          stdout=subprocess.PIPE if r == 0 else sys.stderr))
        import threading
        got = []
        reader = threading.Thread(target=lambda: got.append(procs[0].stdout.read()), daemon=True)
        reader.start()                           # rank 0's one line arrives at the very end; never block the watch loop on it
        failed = None
        deadline = None
        while any([pr.poll() is None for pr in procs]):      # (a list: every child is polled on every round)
            for r, pr in enumerate(procs):
                if pr.returncode not in (None, 0) and failed is None:
                    failed = r
                    open(os.path.join(rdv, "failed"), "w").write("rank %d exited with %d\n" % (r, pr.returncode))
                    # the others see the marker at their next rendezvous wait, or are ended after a grace period
                    deadline = time.time() + float(os.environ.get("GSI_BENCH_FAIL_GRACE_S", "60"))
            if deadline is not None and time.time() > deadline:
                for pr in procs:
                    if pr.poll() is None:
                        pr.kill()                    # exactly the children started above
            time.sleep(0.05)
        rcs = [pr.returncode for pr in procs]
        if any(rc != 0 for rc in rcs):
            sys.stderr.write("bench.py: rank exit codes %s\n" % rcs)
            rc = rcs[failed] if failed is not None else next(rc for rc in rcs if rc != 0)
            return rc if 0 < rc < 256 else 1          # the rank that failed first, not the ones ended because of it
        reader.join(timeout=30)
        sys.stdout.buffer.write(got[0] if got else b"")
        sys.stdout.flush()
        return 0
    finally:
        for pr in procs:
            if pr.poll() is None:
                pr.kill()
        shutil.rmtree(rdv, ignore_errors=True)


class Rendezvous:
    """How the rank processes find each other before the communicator exists: files in a directory all of them can see
    (spawn_ranks creates it; under torch.distributed.run it is derived from the launcher's pid and port)."""

    def __init__(self, rank, world):
        self.rank, self.world = rank, world
        self.dir = os.environ.get("GSI_BENCH_RDV")
        self.own = False
        if not self.dir:
            base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else tempfile.gettempdir()
            # one directory per incarnation of the job: a restarted worker group (max_restarts > 0) must not read the id file of
            # the group before it
            self.dir = os.path.join(base, "gsi-bench-%s-%s-%d-r%s" % (os.environ.get("MASTER_PORT", "0"),
                                                                      os.environ.get("TORCHELASTIC_RUN_ID", "none"), os.getppid(),
                                                                      os.environ.get("TORCHELASTIC_RESTART_COUNT", "0")))
            os.makedirs(self.dir, exist_ok=True)
            self.own = True

    def broadcast(self, name, data=None, timeout=900):
        path = os.path.join(self.dir, name)
        if self.rank == 0:
            with open(path + ".tmp", "wb") as f:
                f.write(data)
            os.rename(path + ".tmp", path)
            return data
        t0 = time.time()
        while not os.path.exists(path):
            if os.path.exists(os.path.join(self.dir, "failed")):
                raise SystemExit("bench.py: another rank failed: " + open(os.path.join(self.dir, "failed")).read().strip())
            if time.time() - t0 > timeout:
                raise SystemExit("bench.py: timed out waiting for rank 0's " + name)
            time.sleep(0.01)
        with open(path, "rb") as f:
            return f.read()

    def fail(self, why):
        """Tell the other ranks (they look at every rendezvous wait; the launcher ends them after its grace period)."""
        try:
            with open(os.path.join(self.dir, "failed"), "w") as f:
                f.write("rank %d: %s\n" % (self.rank, why))
        except OSError:
            pass

    def close(self):
        if self.own and self.rank == 0:
            shutil.rmtree(self.dir, ignore_errors=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--n", type=int, default=1000000, help="grid points of the headline LowRankCovMatrix")
    ap.add_argument("--samples", type=int, default=1024, help="sample fields N_s")
    ap.add_argument("--decay", type=float, default=0.75, help="sample j is scaled by (j+1)^-decay")
    ap.add_argument("--K", type=int, default=256)
    ap.add_argument("--p", type=int, default=64)
    ap.add_argument("--q", type=int, default=2)
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the C2 dense and the implicit 10^6 steps")
    ap.add_argument("--no-fft-512cube", action="store_true", help="skip the one 512^3 FFT-operator step (~230 GB of HBM)")
    ap.add_argument("--cpu-sample-n", type=int, default=16384)
    ap.add_argument("--no-full-parity", action="store_true",
                    help="skip the oracle run at the headline size (about 1-2 min of host LAPACK, ~40 GB of host memory)")
    ap.add_argument("--no-boundary", action="store_true",
                    help="skip the host-boundary block (host-pointer entry points at C1, C2 -- a 34 GB host matrix -- and the headline)")
    ap.add_argument("--no-implicit", action="store_true",
                    help="N > 1: skip the one row-sharded step of the implicit 10^6 x 10^6 covariance")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(spawn_ranks(args.gpus))            # the launcher: no GPU call before or after this line
    # stdout carries exactly ONE line (the JSON result of rank 0): libraries that chat on fd 1 (RCCL's version banner)
    # are sent to stderr for the whole run.
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    if os.environ.get("GSI_BENCH_LAUNCH_TEST"):          # tests/test_bench_launcher.py: the launcher alone, on a box without a GPU
        rv = Rendezvous(rank, world)
        token = rv.broadcast("uid", b"launch-test" if rank == 0 else None)
        if os.environ["GSI_BENCH_LAUNCH_TEST"] == "fail" and rank == 1:
            raise SystemExit(3)
        if os.environ["GSI_BENCH_LAUNCH_TEST"] == "fail" and rank == 0:
            rv.broadcast("never-written-by-anyone" if False else "uid2", b"x")
            time.sleep(600)                              # a rank stuck in a collective its failed peer never enters
        if rank == 0:
            os.write(result_fd, (json.dumps({"launch_test": world, "token": token.decode(), "gpu_modules_loaded":
                                             any("gsi_amd" in m or m == "torch" for m in sys.modules)}) + "\n").encode())
        return
    import gsi_amd as gsi

    # rehearsal of the N > 1 run on a ONE-GPU box: every rank a process on device 0, joined by the library's shared-memory
    # communicator (RCCL refuses two ranks on one device).  The ranks share the GPU, so the figure is not a scaling number;
    # the code path -- row shards, gsi_randsvd_rows, the cross-process pivot exchange of the sharded LU -- is the real one.
    one_gpu = bool(os.environ.get("GSI_BENCH_ONE_GPU"))
    if one_gpu:
        local_rank = 0
        os.environ["GSI_SHM_COMM"] = "1"
    use_dist = world > 1 or bool(os.environ.get("GSI_BENCH_FORCE_DIST"))   # the latter: rehearse the N > 1 code on one GPU
    rdv = None
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

    ctx = gsi.Context(local_rank)
    if use_dist:
        # rank 0 makes the communicator id and ships it as a file; from comm_init on, every barrier and reduction of this
        # harness goes through the library's communicator (RCCL over xGMI; shared memory in the one-GPU rehearsal)
        rdv = Rendezvous(rank, world)
        uid = rdv.broadcast("uid", bytes(ctx.unique_id()) if rank == 0 else None)
        ctx.comm_init(world, rank, uid)

    def barrier():
        ctx.sync()
        if use_dist:
            ctx.barrier()

    def max_over_ranks(x):
        if not use_dist:
            return x
        return float(ctx.host_allgather([x]).max())

    # ---------------- headline: n = 1e6 LowRankCovMatrix, rank 256 -------------------------------------------
    n = args.n if args.scaling == "strong" else args.n * world
    Ns, K, p, q = args.samples, args.K, args.p, args.q
    l = K + p
    op = gsi.lowrank_synthetic_operator(ctx, n, Ns, seed=0, decay=args.decay)      # samples generated + centred in HBM
    full_parity = world == 1 and not args.no_cpu_baseline and not args.no_full_parity
    # N > 1 (same problem row-sharded): rank 0 afterwards re-runs the step on one GPU and reports how far the N-rank numbers
    # are from it -- when the whole problem fits one GPU comfortably (n <= 2e6: 16 GB of samples, 5 GB panels)
    vs_one_gpu = world > 1 and rank == 0 and n <= 2000000 and not args.no_full_parity
    want_host = full_parity or (world == 1 and not use_dist and not args.no_boundary)     # the boundary block re-uploads the samples
    keep = {} if (want_host or vs_one_gpu) else None
    elapsed, phases, Sv = run_steps(gsi, ctx, op, n, K, p, q, args.steps, args.warmup, barrier, keep=keep, rows=use_dist)
    elapsed = max_over_ranks(elapsed)
    pinfo = run_steps.last_path_info                     # of exactly the timed steps
    ranks_phases = skew_split = None
    if use_dist:
        # the same phases per rank (max / min), and ONE more step with skew barriers (profile level 2): what is waiting for the
        # slowest rank and what is work -- outside the timed region, the barriers are extra collectives
        ranks_phases = phases_over_ranks(ctx, phases, args.steps)
        _, ph2, _ = run_steps(gsi, ctx, op, n, K, p, q, 1, 0, barrier, rows=True, profile=2)
        skew_split = phases_over_ranks(ctx, ph2, 1)
    counters = ctx.counters()
    counters.update({"lu_form": pinfo["lu_form"], "lu_forms_run_in_timed_steps": pinfo["lu_forms_run"],
                     "lu_selftest_mask": pinfo["lu_selftest_mask"],
                     "collectives_per_step": pinfo["collectives"] / max(args.steps, 1), "n_ranks_seen": pinfo["n_ranks_seen"],
                     "lu_timeouts": pinfo["lu_timeouts"], "lu_timeouts_recovered": pinfo["lu_timeouts_recovered"],
                     "svd_sweep_cap_hits": pinfo["svd_sweep_cap_hits"]})
    if use_dist:                         # every rank must have run the same form: a rank that fell back alone is a bug
        forms = ctx.host_allgather([float(gsi.Context.LU_FORMS.index(pinfo["lu_form"])), float(pinfo["lu_selftest_mask"])])
        counters["lu_form_same_on_all_ranks"] = bool((forms == forms[0]).all())
    dev_bytes = ctx.device_bytes()
    host = None
    Z0_rows = None
    if vs_one_gpu:
        Z0_rows = keep["Z"].to_host()
        keep["Omega"].close()
        keep["Z"].close()
    if want_host:                        # after the timed region: what the oracle / the boundary block need, off the device
        try:
            host = {"samples": gsi.device_samples(op, Ns), "Omega": keep["Omega"].to_host(), "Z": keep["Z"].to_host()}
        except Exception:                # noqa: BLE001 -- not enough host memory: the n = 16384 sample is what remains
            host = None
        keep["Omega"].close()
        keep["Z"].close()
    op.close()

    ms_per_step = 1e3 * elapsed / args.steps
    value = lrcm_bytes(n, Ns, l, q) * args.steps / elapsed / 1e9
    row0, nloc = ctx.shard(n)
    g_ms = phases["gemm_n"][0] + phases["gemm_t"][0]
    g_cnt = phases["gemm_n"][1] + phases["gemm_t"][1]
    avg_ms = g_ms / max(g_cnt, 1)
    flops_per_launch = 2.0 * nloc * Ns * l                   # both S'X (N_s x l, K = n) and S T (n x l, K = N_s)
    bytes_per_launch = 8.0 * nloc * (Ns + l)                 # the sample shard once + the tall panel once
    achieved = flops_per_launch / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
    hbm_gbs = bytes_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    traffic = None
    panel_traffic = {}
    traffic_source = None
    for tname in TRAFFIC_FILES if world == 1 and not use_dist else []:
        tpath = os.path.join(ROOT, "profiles", tname)
        if not os.path.exists(tpath):
            continue
        try:
            t = json.load(open(tpath))
            if (t.get("csrc_hash") == kernel_source_hash() and t.get("n") == n and t.get("samples") == Ns
                    and t.get("l") == l):
                traffic = t.get("hbm_bytes_per_launch")
                for k in ("lu", "qr"):
                    if t.get(k, {}).get("hbm_bytes_per_factorization"):
                        panel_traffic[k] = float(t[k]["hbm_bytes_per_factorization"])
                traffic_source = "replayed from profiles/%s (rocprofv3 --pmc passes of this command on this build: " \
                                 "csrc hash %s); not measured in this run" % (tname, t.get("csrc_hash"))
                break
        except Exception:
            traffic = None
            panel_traffic = {}
    roofline = {
        "bound": "mfma", "achieved": achieved, "peak": PEAK_FP64_MFMA_TFLOPS, "unit": "TFLOP/s",
        "frac": achieved / PEAK_FP64_MFMA_TFLOPS, "traffic": traffic, "traffic_source": traffic_source,
        "kernel": "gemm_f64_kernel<NT,TRANS_A,GEN,XMODE> (v_mfma_f64_16x16x4_f64) + its fixed-order split-K slab reduction",
        "avg_launch_ms": avg_ms, "launches": int(g_cnt),
        "flops_per_launch": flops_per_launch, "algorithmic_bytes_per_launch": bytes_per_launch,
        "hbm": {"achieved": hbm_gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": hbm_gbs / PEAK_HBM_GBS,
                "note": "intensity 2 N_s l / (8 (N_s + l)) = %.0f flop/B >> ridge 9.8: the contraction is MFMA-bound; "
                        "at full MFMA peak this stream would reach %.2f of the HBM peak" % (
                            2.0 * Ns * l / (8.0 * (Ns + l)),
                            (bytes_per_launch / (flops_per_launch / (PEAK_FP64_MFMA_TFLOPS * 1e12))) / 1e9 / PEAK_HBM_GBS)},
    }
    panel_bytes = 16.0 * n * l                               # one read + one write of the (replicated) n x l panel
    phases_hbm = {}
    for name, key in (("lu", "lu"), ("qr", "qr"), ("z_product", "small_gemm")):
        ms, cnt = phases[key]
        if cnt > 0 and ms > 0:
            per = ms / cnt
            phases_hbm[name] = {"ms_per_factorization": per, "factorizations_per_step": cnt / args.steps,
                                "algorithmic_bytes": panel_bytes, "GB/s": panel_bytes / (per * 1e-3) / 1e9,
                                "frac_of_hbm_peak": panel_bytes / (per * 1e-3) / 1e9 / PEAK_HBM_GBS}
            if name in panel_traffic:        # bytes the factorization actually moved (PMC passes of this command)
                phases_hbm[name]["traffic"] = panel_traffic[name]
                phases_hbm[name]["traffic_GB/s"] = panel_traffic[name] / (per * 1e-3) / 1e9
                phases_hbm[name]["traffic_frac_of_hbm_peak"] = panel_traffic[name] / (per * 1e-3) / 1e9 / PEAK_HBM_GBS
    phases_hbm["svd_small_jacobi_ms_per_step"] = phases["svd"][0] / args.steps

    if rank == 0:
        out = {
            "metric": "randSVD GB/s + top-k singular-value rel-err, n=1e6 rank=256", "value": value, "unit": "GB/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"n={n} grid points, LowRankCovMatrix over N_s={Ns} synthetic sample fields "
                                   f"({8.0 * n * Ns / 1e9:.1f} GB in HBM), rank K={K}, p={p} (l={l}), q={q} "
                                   f"(BASELINE.json metric config; SURVEY.md 8d C4-ii)"
                                   + ("" if world == 1 else (", same problem row-sharded" if args.scaling == "strong"
                                                              else ", 1e6 rows per GPU")),
                       "n": n, "samples": Ns, "K": K, "p": p, "q": q, "parallelism": f"row-shard x{world}" + (" (Omega and Z as row shards: gsi_randsvd_rows)" if use_dist else "")
                       + (" -- REHEARSAL: all ranks share GPU 0 (GSI_BENCH_ONE_GPU)" if one_gpu and world > 1 else ""),
                       "operator_bytes_per_gpu": 8.0 * nloc * Ns, "device_bytes_in_use": dev_bytes},
            "roofline": roofline,
            "phases_hbm": phases_hbm,
            "phases_ms_per_step": {k: v[0] / args.steps for k, v in phases.items()},
            "phase_launch_groups_per_step": {k: v[1] / args.steps for k, v in phases.items()},
            "phases_ms_max_min_over_ranks": ranks_phases,
            "phases_ms_skew_split_one_step": skew_split,
            "path_counters": counters,
            "algorithmic_bytes_per_step": lrcm_bytes(n, Ns, l, q),
            "kernel_source_hash": kernel_source_hash(),
        }
        # size-independent property at the full size: the trailing p singular values exist, descending, positive
        out["sv_descending_positive"] = bool(all(Sv[i] >= Sv[i + 1] for i in range(l - 1)) and Sv[K - 1] > 0)
        if vs_one_gpu:
            try:
                out["multi_rank_vs_one_gpu"] = same_numbers_as_one_gpu(gsi, local_rank, n, Ns, args.decay, K, p, q, world, 1234, Sv,
                                                                       Z0_rows)
            except Exception as exc:                        # noqa: BLE001 -- the check must not take the line down
                out["multi_rank_vs_one_gpu"] = {"error": f"{type(exc).__name__}: {exc}"}

    # ---------------- N > 1: ONE row-sharded step of the operator north_star's ">= 6x at 8 GPUs" names ---------------
    if world > 1 and not args.no_implicit and not args.no_secondary:
        try:
            imp = secondary_implicit(gsi, ctx, barrier, rows=True, max_over_ranks=max_over_ranks)
        except gsi.GsiError as exc:
            # a library error inside a collective section: the peers may be in a different collective, so meeting them at the
            # final barrier could hang until the driver's time-out -- mark the job failed and leave non-zero
            sys.stderr.write(f"bench.py: rank {rank}: {type(exc).__name__}: {exc}\n")
            if rdv is not None:
                rdv.fail(f"{type(exc).__name__} in the row-sharded implicit step")
            os._exit(4)
        if rank == 0:
            out["secondary"] = {"implicit_dense_1e6": imp}

    # ---------------- secondary workloads (one GPU only) -------------------------------------------------------
    if world == 1 and not args.no_secondary:
        sec = {}

        def guarded(name, fn):
            # a secondary workload must never take the headline line down with it: its failure is recorded instead
            try:
                sec[name] = fn()
            except Exception as exc:                        # noqa: BLE001
                sec[name] = {"error": f"{type(exc).__name__}: {exc}"}
                for ch in list(ctx._children):             # whatever the failed workload left on the device
                    ch.close()

        guarded("headline_p0", lambda: secondary_headline_p0(gsi, ctx, barrier, n, Ns, K, q, args.decay))
        if not args.no_cpu_baseline:                        # (it times the oracle beside the GPU)
            guarded("c1_dense_2000", lambda: secondary_c1(gsi, ctx, barrier))
        guarded("c2_dense_65536", lambda: secondary_c2(gsi, ctx, barrier))
        guarded("implicit_dense_1e6", lambda: secondary_implicit(gsi, ctx, barrier))
        guarded("fft_powerlaw_1000sq", lambda: secondary_fft_1000sq(gsi, ctx, barrier))
        guarded("pointcov_implicit_2e5", lambda: secondary_pointcov(gsi, ctx, barrier))
        guarded("pcgalsqr_c5_1e6", lambda: secondary_pcgalsqr_c5(gsi, ctx, barrier))
        if not args.no_fft_512cube:
            guarded("fft_powerlaw_512cube", lambda: secondary_fft_512cube(gsi, ctx, barrier))
        out["secondary"] = sec

    if rank == 0 and world == 1 and not args.no_boundary:
        try:
            sec_ = out.get("secondary", {})
            hb = dict(host, S=Sv) if host is not None else None
            out["boundary"] = boundary_block(gsi, ctx, barrier, ms_per_step, hb, Ns, K, p, q,
                                             sec_.get("c1_dense_2000", {}).get("ms_per_step"),
                                             sec_.get("c2_dense_65536", {}).get("ms_per_step"),
                                             sec_.get("c1_dense_2000", {}).get("oracle_ms_per_step"))
        except Exception as exc:                            # noqa: BLE001 -- must not cost the line
            out["boundary"] = {"error": f"{type(exc).__name__}: {exc}"}
            for ch in list(ctx._children):
                ch.close()
    if rank == 0:
        out["toolchain"] = toolchain_info()
        if world == 1 and not args.no_cpu_baseline:
            cb, err, xerr = cpu_baseline_and_parity(gsi, ctx, Ns, K, p, q, args.cpu_sample_n, args.decay)
            out["cpu_baseline"] = cb
            small = {"n": args.cpu_sample_n, "sv_rel_err": err, "xis_err_up_to_sign": xerr,
                     "oracle_products": "the reference's ger!/gemv loop (lowrank.jl:115-121)"}
            full = None
            if host is not None and full_parity:
                # the metric's rel-err on the metric's configuration: HIP vs oracle on the timed operator itself
                try:
                    full = full_size_parity(host, Ns, K, p, q, Sv)
                except Exception as exc:                     # noqa: BLE001 -- host memory / LAPACK trouble must not cost the line
                    out["parity_full_size"] = {"error": f"{type(exc).__name__}: {exc}"}
            if full is not None:
                # the same algorithm at the metric's own size on this box's host cores, beside the bounded sample (VERDICT r4
                # weak 14): products in GEMM form (the reference's ger!/gemv loop would take hours at n = 1e6)
                cb["full_size_oracle"] = {"seconds": full["oracle_seconds"], "n": full["n"], "threads": full["oracle_threads"],
                                          "GB/s": lrcm_bytes(n, Ns, l, q) / full["oracle_seconds"] / 1e9,
                                          "gpu_step_speedup": full["oracle_seconds"] / (ms_per_step * 1e-3),
                                          "products": full["oracle_products"]}
                out["sv_rel_err"] = {"value": full["sv_rel_err"], "n": full["n"], "K": K, "tolerance": 1e-5}
                out["xis_err_up_to_sign"] = {"value": full["xis_err_up_to_sign"], "n": full["n"], "K": K,
                                             "tolerance": 1e-6}
                out["parity_full_size"] = full
                out["parity_cpu_sample"] = small
            else:
                out["sv_rel_err"] = {"value": err, "n": args.cpu_sample_n, "K": K, "tolerance": 1e-5}
                out["xis_err_up_to_sign"] = {"value": xerr, "n": args.cpu_sample_n, "K": K, "tolerance": 1e-6}
                out["parity_cpu_sample"] = small
        else:
            out["cpu_baseline"] = None
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    if use_dist:
        ctx.barrier()
        rdv.close()
    ctx.close()


if __name__ == "__main__":
    main()
