"""N > 1 path on CPU: the row-sharded pipeline (pipeline.cpp, exactly the code in libgsi_hip.so)
on the CPU reference backend, one process per rank, collectives over torch.distributed/gloo.
On the GPU the same calls go to RCCL; here they are ctypes callbacks -- the order of operations,
the shard arithmetic and the collective sequence are what is under test."""
import multiprocessing as mp
import os
import socket
import sys
import traceback

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    try:
        sys.path.insert(0, ROOT)
        sys.path.insert(0, HERE)
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        os.environ["OMP_NUM_THREADS"] = "2"
        import torch
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world, init_method=f"tcp://127.0.0.1:{port}")
        import gsi_amd as gsi
        import cpuref
        from oracle import oracle as orc
        from helpers import gaussian_cov, exponential_cov, powerlaw_fields, rel_sv_err

        lib = cpuref.load_cpuref()

        def allreduce(buf, count):
            t = torch.from_numpy(np.ctypeslib.as_array(buf, shape=(count,)))
            dist.all_reduce(t)                                     # in place on the library's buffer

        def allgather(send, recv, count):
            s = torch.from_numpy(np.ctypeslib.as_array(send, shape=(count,)).copy())
            outs = [torch.empty(count, dtype=torch.float64) for _ in range(world)]
            dist.all_gather(outs, s)
            np.ctypeslib.as_array(recv, shape=(world * count,))[:] = torch.cat(outs).numpy()

        cbs = (cpuref.ALLREDUCE_FN(allreduce), cpuref.ALLGATHER_FN(allgather))
        lib.gsi_cpuref_set_collectives(*cbs)
        ctx = gsi.Context(0, lib=lib)
        ctx.comm_init(world, rank, b"\0" * 128)
        assert ctx.rank() == (rank, world)
        results = {}
        rng = np.random.default_rng(7)                              # same stream on every rank

        # dense, n not divisible by the world size; q = 0 exercises the TSQR tree, q = 2 the LU exchange
        A = gaussian_cov(13, 11, 3.0)                               # n = 143
        for qq in (0, 2):
            K, p = 10, 6
            Om = rng.standard_normal((143, K + p))
            Z, S = gsi.randsvd(A, K, p, qq, Omega=Om, return_S=True, ctx=ctx)
            Zr, Sr, Qr = orc.randsvd_full(A, K, p, qq, Om)
            results[f"dense_q{qq}_sv"] = rel_sv_err(S, Sr, K)
            results[f"dense_q{qq}_xis"] = orc.xis_error_up_to_sign(Z, Zr, K)
            Q = gsi.rangefinder(A, K + p, qq, Omega=Om, ctx=ctx)
            results[f"dense_q{qq}_orth"] = float(np.abs(Q.T @ Q - np.eye(K + p)).max())
            s1 = np.linalg.svd(Q.T @ A, compute_uv=False)
            s2 = np.linalg.svd(Qr.T @ A, compute_uv=False)
            results[f"dense_q{qq}_range"] = rel_sv_err(s1, s2, K)
        # profile level 2 (gsi_ctx_profile): every collective and every panel LU is preceded by a one-double all-reduce counted
        # as the phase `comm_wait`; the numbers are the same, the extra collectives show in gsi_ctx_path_info
        Om = rng.standard_normal((143, 16))
        ctx.profile(1)
        ctx.phase_reset()
        Z1, S1 = gsi.randsvd(A, 10, 6, 2, Omega=Om, return_S=True, ctx=ctx)
        ph1, col1 = ctx.phase_times(), ctx.path_info()["collectives"]
        ctx.profile(2)
        ctx.phase_reset()
        Z2, S2 = gsi.randsvd(A, 10, 6, 2, Omega=Om, return_S=True, ctx=ctx)
        ph2, col2 = ctx.phase_times(), ctx.path_info()["collectives"]
        ctx.profile(0)
        results["skew_profile_same_numbers"] = 0.0 if (np.array_equal(Z1, Z2) and np.array_equal(S1, S2)) else 1.0
        results["skew_profile_counts"] = 0.0 if (ph1["comm_wait"][1] == 0 and ph2["comm_wait"][1] > 0
                                                 and col2 - col1 == ph2["comm_wait"][1] and ph2["lu"][1] == ph1["lu"][1]) else 1.0
        # non-symmetric operator (Jacobian-like), shards shorter than... still >= l
        B = rng.standard_normal((90, 60)) @ np.diag(np.logspace(0, -4, 60)) @ rng.standard_normal((60, 60))
        Om = rng.standard_normal((60, 12))
        Q = gsi.rangefinder(B, 12, 1, Omega=Om, ctx=ctx)
        Qr = orc.rangefinder(B, 12, 1, Om)
        results["rect_range"] = rel_sv_err(np.linalg.svd(Q.T @ B, compute_uv=False),
                                           np.linalg.svd(Qr.T @ B, compute_uv=False), 8)
        # a shard shorter than the sketch width -> gathered-QR fallback
        C = exponential_cov(6, 5, 3.0)                              # n = 30, l = 14: shards of 15/15 (world 2) or 10 (world 3)
        Om = rng.standard_normal((30, 14))
        Z, S = gsi.randsvd(C, 10, 4, 1, Omega=Om, return_S=True, ctx=ctx)
        Zr, Sr, _ = orc.randsvd_full(C, 10, 4, 1, Om)
        results["short_sv"] = rel_sv_err(S, Sr, 10)
        # LowRankCovMatrix, sample rows sharded
        fields = powerlaw_fields(rng, (9, 9), 20)
        Om = rng.standard_normal((81, 9))
        lr = gsi.LowRankCovMatrix(fields, ctx=ctx)
        Z = gsi.randsvd(lr, 6, 3, 3, Omega=Om)
        xr, _ = orc.getxis_fields(fields, 6, 3, 3, Om)
        results["lowrank_xis"] = orc.xis_error_up_to_sign(Z, np.array(xr).T, 6)
        X = rng.standard_normal((81, 4))
        results["lowrank_mul"] = float(np.abs(lr.matmul(X) - orc.LowRankCovMatrix(fields).matmul(X)).max())
        lr.close()
        # LowRankCovMatrix whose last rank holds NO rows (n = 4: shards 2/2/0 at world 3, 2/2 at world 2): the
        # empty rank must still enter the all-reduce of S'X inside every product (a skipped collective hangs RCCL)
        small = [rng.standard_normal(4) for _ in range(6)]
        lr4 = gsi.LowRankCovMatrix(small, ctx=ctx)
        X4 = rng.standard_normal((4, 3))
        results["lowrank_empty_rank_mul"] = float(np.abs(lr4.matmul(X4) - orc.LowRankCovMatrix(small).matmul(X4)).max())
        Om4 = rng.standard_normal((4, 3))
        Z4, S4 = gsi.randsvd(lr4, 2, 1, 1, Omega=Om4, return_S=True)
        x4, _ = orc.getxis_fields(small, 2, 1, 1, Om4)
        results["lowrank_empty_rank_xis"] = orc.xis_error_up_to_sign(Z4, np.array(x4).T, 2)
        lr4.close()
        # row-sharded partial-pivot LU (SURVEY 8e "sharded alternative"): dgetrf's pivots, L equal to the single-rank
        # factorization EXACTLY, on every rank; ties (duplicate rows across the shard boundary) included
        ctx1 = gsi.Context(0, lib=lib)                               # a second context without a communicator
        for (mm, ll) in [(143, 16), (90, 30), (64, 21)]:
            Yp = rng.standard_normal((mm, ll))
            if mm == 90:
                Yp[60:75] = Yp[5:20]                                # exact ties between rows of different ranks
            Ls, ps = gsi.lu_L_sharded(Yp, return_pivots=True, ctx=ctx)
            L1, p1 = gsi.lu_L(Yp, return_pivots=True, ctx=ctx1)
            results[f"lu_sharded_{mm}_pivots_equal"] = 0.0 if (np.array_equal(ps, p1) and np.array_equal(ps, orc.lu_pivots(Yp))) else 1.0
            results[f"lu_sharded_{mm}_L_exact"] = 0.0 if np.array_equal(Ls, L1) else 1.0
        ctx1.close()
        # operator products gathered to every rank
        op = gsi.dense_operator(ctx, B)
        X = rng.standard_normal((60, 3))
        results["mul"] = float(np.abs(op.matmul(X) - B @ X).max())
        Y = rng.standard_normal((90, 3))
        results["mul_t"] = float(np.abs(op.rmatmul_t(Y) - B.T @ Y).max())
        op.close()
        # implicit grid covariance (never stored): sharded generated products against the dense matrix
        G = gaussian_cov(9, 7, 2.5)                                  # n = 63
        gop = gsi.gridcov_implicit_operator(ctx, 9, 7, 2.5)
        X = rng.standard_normal((63, 5))
        results["implicit_mul"] = float(np.abs(gop.matmul(X) - G @ X).max())
        results["implicit_mul_t"] = float(np.abs(gop.rmatmul_t(X) - G.T @ X).max())
        Om = rng.standard_normal((63, 12))
        Z, S = gsi.randsvd(gop, 8, 4, 2, Omega=Om, return_S=True)
        Zr, Sr, _ = orc.randsvd_full(G, 8, 4, 2, Om)
        results["implicit_sv"] = rel_sv_err(S, Sr, 8)
        results["implicit_xis"] = orc.xis_error_up_to_sign(Z, Zr, 8)
        gop.close()
        # scattered-point covariance (entries generated panel by panel): sharded products and randsvd against the dense matrix
        Pp = rng.uniform(0.0, 20.0, size=(2, 70))
        dd = np.sqrt(((Pp[:, :, None] - Pp[:, None, :]) ** 2).sum(axis=0)) / 5.0
        Ap = (1.0 + np.sqrt(3.0) * dd) * np.exp(-np.sqrt(3.0) * dd)
        pop = gsi.pointcov_implicit_operator(ctx, Pp, "matern32", ell=5.0)
        X = rng.standard_normal((70, 4))
        results["pointcov_mul"] = float(np.abs(pop.matmul(X) - Ap @ X).max())
        results["pointcov_mul_t"] = float(np.abs(pop.rmatmul_t(X) - Ap @ X).max())
        Om = rng.standard_normal((70, 12))
        Z, S = gsi.randsvd(pop, 8, 4, 2, Omega=Om, return_S=True)
        Zr, Sr, _ = orc.randsvd_full(Ap, 8, 4, 2, Om)
        results["pointcov_sv"] = rel_sv_err(S, Sr, 8)
        results["pointcov_xis"] = orc.xis_error_up_to_sign(Z, Zr, 8)
        pop.close()
        # ---- matrix-free FFT covariance on several ranks (BASELINE configs[2] as configured needs the panels spread over
        #      GPUs): every rank transforms its own columns, panels are row shards between the products (all-to-all)
        Ns, beta = [12, 9], -3.0                                     # n = 108; FFTRF convention, 2N not a power of two on axis 1
        nf = 108
        Af = orc.fft_powerlaw_apply(np.eye(nf), Ns, beta, fftrf=True)
        fop = gsi.fft_powerlaw_operator(ctx, Ns, beta, fftrf=True)
        X = rng.standard_normal((nf, 7))
        results["fft_mul"] = float(np.abs(fop.matmul(X) - Af @ X).max())
        results["fft_mul_t"] = float(np.abs(fop.rmatmul_t(X) - Af @ X).max())
        K, p, qq = 9, 5, 2
        Om = rng.standard_normal((nf, K + p))
        Zf, Sf = gsi.randsvd(fop, K, p, qq, Omega=Om, return_S=True)          # Omega replicated, Z gathered
        Zr, Sr, _ = orc.randsvd_full(Af, K, p, qq, Om)
        results["fft_sv"] = rel_sv_err(Sf, Sr, K)
        results["fft_xis"] = orc.xis_error_up_to_sign(Zf, Zr, K)

        def gather_rows(loc):                                         # host-side all-gather of row blocks (test harness)
            parts = [None] * world
            dist.all_gather_object(parts, np.ascontiguousarray(loc))
            return np.concatenate(parts, axis=0)

        def my_rows(full):
            r0, nl = ctx.shard(full.shape[0])
            return np.asfortranarray(full[r0:r0 + nl])

        # gsi_randsvd_rows: Omega and Z as row shards, nothing n x l on any rank (FFT, LowRankCovMatrix); dense gathers Omega
        Zrows, S2 = gsi.randsvd_rows(fop, K, p, qq, my_rows(Om), return_S=True)
        Zfull = gather_rows(Zrows.to_host())
        results["fft_rows_sv"] = rel_sv_err(S2, Sr, K)
        results["fft_rows_xis"] = orc.xis_error_up_to_sign(Zfull, Zr, K)
        Zrows.close()
        Z0 = gather_rows(gsi.randsvd_rows(fop, K, p, 0, my_rows(Om)).to_host())   # q = 0: sketch + TSQR only
        Zr0, Sr0, _ = orc.randsvd_full(Af, K, p, 0, Om)
        results["fft_rows_q0_xis"] = orc.xis_error_up_to_sign(Z0, Zr0, K)
        fop.close()
        fields = powerlaw_fields(rng, (10, 9), 25)
        Om = rng.standard_normal((90, 12))
        lr = gsi.LowRankCovMatrix(fields, ctx=ctx)
        Zl = gather_rows(gsi.randsvd_rows(lr._device_operator(), 8, 4, 3, my_rows(Om)).to_host())
        xr, _ = orc.getxis_fields(fields, 8, 4, 3, Om)
        results["lowrank_rows_xis"] = orc.xis_error_up_to_sign(Zl, np.array(xr).T, 8)
        lr.close()
        Om = rng.standard_normal((143, 16))
        dop = gsi.dense_operator(ctx, A)
        Zd, Sd = gsi.randsvd_rows(dop, 10, 6, 2, my_rows(Om), return_S=True)
        Zr, Sr, _ = orc.randsvd_full(A, 10, 6, 2, Om)
        results["dense_rows_sv"] = rel_sv_err(Sd, Sr, 10)
        results["dense_rows_xis"] = orc.xis_error_up_to_sign(gather_rows(Zd.to_host()), Zr, 10)
        dop.close()
        # ---- BASELINE configs[4]: pcgalsqr over a ROW-SHARDED xi-basis (every rank keeps its rows of Z, s, X and of the
        #      perturbation batch; the forward model is host code and sees gathered vectors) against the oracle's pcgalsqr
        Np, Mp = 96, 6
        xs = rng.standard_normal(Np)
        Q0 = rng.standard_normal((Mp, Np))
        Qc = Q0.T @ Q0
        truep = np.real(np.linalg.cholesky(Qc + 1e-9 * np.eye(Np)) @ rng.standard_normal(Np)) + 1.0
        forward = lambda pv: pv * xs
        noise = 1e-4
        yobs = forward(truep) + noise * rng.standard_normal(Np)
        import scipy.sparse as sp
        Rn = noise ** 2 * sp.identity(Np, format="csc")
        Omp = rng.standard_normal((Np, Mp + 2))
        qop = gsi.dense_operator(ctx, Qc)
        Zp = gsi.randsvd_rows(qop, Mp, 2, 3, my_rows(Omp))
        basis = gsi.ShardedDeviceBasis(Zp, Mp, gather_rows)
        X0 = np.full(Np, 1.0)
        r0p, nlp = ctx.shard(Np)
        s_loc = gsi.pcgalsqr(forward, X0[r0p:r0p + nlp], X0[r0p:r0p + nlp], basis, Rn, yobs, ctx=ctx)
        s_full = gather_rows(s_loc)
        xis_ref = orc.getxis_dense(Qc, Mp, 2, 3, Omp)
        s_ref = orc.pcgalsqr(forward, X0, X0, xis_ref, Rn, yobs)
        results["pcgalsqr_sharded_basis"] = float(np.linalg.norm(s_full - s_ref) / np.linalg.norm(s_ref))
        results["pcgalsqr_sharded_fit"] = 0.0 if np.linalg.norm(s_full - truep) / np.linalg.norm(truep) < 2e-2 else 1.0
        basis.close(); Zp.close(); qop.close()
        # round 4: host values through the library's communicator, and what gsi_ctx_path_info says about the run so far
        got = ctx.host_allgather([rank + 0.5, 7.0])
        results["host_allgather"] = float(np.abs(got - np.array([[r + 0.5, 7.0] for r in range(world)])).max())
        pi = ctx.path_info()
        results["path_ranks_seen"] = 0.0 if pi["n_ranks_seen"] == world else 1.0
        # the CPU reference backend has no in-kernel exchange: sharded LUs ran per step, gathered ones replicated; the
        # self-test was tried and found nothing (mask 0); no time-outs
        ok = set(pi["lu_forms_run"]) <= {"replicated", "per-step"} and pi["lu_form"] in ("replicated", "per-step") \
            and pi["lu_selftest_mask"] == 0 and pi["lu_timeouts"] == 0 and pi["collectives"] > 0
        results["path_info_consistent"] = 0.0 if ok else 1.0
        ctx.phase_reset()
        results["path_reset"] = float(ctx.path_info()["collectives"]) + float(sum(ctx.path_info()["lu_forms_run"].values()))
        ctx.close()
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, "ok", results))
    except Exception:
        q.put((rank, "error", traceback.format_exc()))


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_pipeline_gloo(world):
    sys.path.insert(0, HERE)
    import cpuref
    cpuref.build()
    mpctx = mp.get_context("spawn")
    q = mpctx.Queue()
    port = _free_port()
    procs = [mpctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = []
    try:
        for _ in range(world):
            out.append(q.get(timeout=240))
    finally:
        for p in procs:
            p.join(timeout=30)
            if p.is_alive():
                p.kill()
    for rank, status, payload in out:
        assert status == "ok", f"rank {rank}:\n{payload}"
    for rank, _, res in out:
        for k, v in res.items():
            tol = 1e-6 if k.endswith("xis") else 1e-9
            if k.endswith("orth") or k in ("mul", "mul_t", "lowrank_mul", "lowrank_empty_rank_mul", "implicit_mul", "implicit_mul_t",
                                            "fft_mul", "fft_mul_t", "pointcov_mul", "pointcov_mul_t"):
                tol = 1e-11
            if k == "pcgalsqr_sharded_basis":
                tol = 1e-6
            assert v < tol, (rank, k, v)
    # every rank computed the same replicated result
    r0 = out[0][2]
    for _, _, res in out[1:]:
        for k in r0:
            assert abs(res[k] - r0[k]) < 1e-12 + 1e-6 * abs(r0[k])
