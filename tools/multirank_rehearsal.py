#!/usr/bin/env python3
"""The multi-rank pipeline at real sizes on ONE GPU: `world` rank PROCESSES on device 0 over the shared-memory communicator
(GSI_SHM_COMM=1; RCCL refuses two ranks on one device) run gsi_randsvd_rows on an operator; a 1-rank run of gsi_randsvd on
the same operator and the same Omega (the ranks' blocks stacked) is the reference.  Prints one JSON line per case.
The ranks share the chip: times are sums of the ranks' work, not scaling figures.

    python tools/multirank_rehearsal.py [--worlds 2 4] [--case fft3d|fft2d|lowrank|implicit ...]
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = {
    # name: (operator spec, K, p, q)
    "fft3d": (("fft", [256, 256, 256]), 39, 9, 2),          # n = 1.7e7, FFTRF convention, 1 GB per column
    "fft2d": (("fft", [1000, 1000]), 205, 51, 2),            # BASELINE configs[2] at n = 1e6 (re-embedded spectrum: 1000 is no power of two)
    "lowrank": (("lowrank", 2000000, 512), 256, 64, 2),      # taller than the bench's headline: 1e6 rows per rank at world 2
    "implicit": (("implicit", 300, 300), 128, 32, 1),        # 90 000 x 90 000 never stored, transposed products row-sharded
    # the same operators on grids WITHOUT axis symmetry: |k| has no multiplicities beyond +-k, so single xi-vectors are
    # well determined and the N-rank xis can be compared with the one-rank ones column by column (on the cubic / square
    # grids above eigenvalues come in multiplets and a vector is only determined up to a rotation inside its multiplet)
    "fft3d_asym": (("fft", [256, 240, 200]), 39, 9, 2),
    "fft2d_asym": (("fft", [1000, 900]), 205, 51, 2),
}


def child(args):
    sys.path.insert(0, ROOT)
    import numpy as np
    import gsi_amd as gsi
    world, rank, tmp, case = args.world, args.rank, args.tmp, args.case
    spec, K, p, q = CASES[case]
    l = K + p
    ctx = gsi.Context(0)

    def wait_for(path):
        t0 = time.time()
        while not os.path.exists(path):
            if time.time() - t0 > 600:
                raise RuntimeError("timed out waiting for " + path)
            time.sleep(0.01)

    if world > 1:
        idfile = os.path.join(tmp, "uid")
        if rank == 0:
            with open(idfile + ".tmp", "wb") as f:
                f.write(bytes(ctx.unique_id()))
            os.rename(idfile + ".tmp", idfile)
        wait_for(idfile)
        with open(idfile, "rb") as f:
            ctx.comm_init(world, rank, f.read())
    if spec[0] == "fft":
        n = int(np.prod(spec[1]))
        op = gsi.fft_powerlaw_operator(ctx, spec[1], -3.5, fftrf=True)
    elif spec[0] == "lowrank":
        n = spec[1]
        op = gsi.lowrank_synthetic_operator(ctx, n, spec[2], seed=0, decay=0.75)
    else:
        n = spec[1] * spec[2]
        op = gsi.gridcov_implicit_operator(ctx, spec[1], spec[2], 30.0, kind=1)
    pad = (n + args.ranks_of_omega - 1) // args.ranks_of_omega

    def omega_block(r):                     # Omega is DEFINED as args.ranks_of_omega row blocks, block r from seed 1234 + 7919 r
        r0 = min(r * pad, n)
        return r0, gsi.DeviceMatrix(ctx, min(pad, n - r0), l).randn(1234 + 7919 * r)

    lib = ctx.lib
    S = gsi.DeviceMatrix(ctx, l, 1)
    if world > 1:
        assert world == args.ranks_of_omega
        _, Om = omega_block(rank)
        Z = gsi.DeviceMatrix(ctx, Om.shape[0], l)
        step = lambda: gsi._lib.check(lib.gsi_randsvd_rows(ctx.h, op.h, Om.h, K, p, q, Z.h, S.h), lib)
    else:
        Omh = np.empty((n, l), order="F")
        for r in range(args.ranks_of_omega):
            r0, blk = omega_block(r)
            Omh[r0:r0 + blk.shape[0]] = blk.to_host()
            blk.close()
        Om = gsi.DeviceMatrix.from_host(ctx, Omh)
        del Omh
        Z = gsi.DeviceMatrix(ctx, n, l)
        step = lambda: gsi._lib.check(lib.gsi_randsvd_dev(ctx.h, op.h, Om.h, K, p, q, Z.h, S.h), lib)
    step()
    ctx.sync()
    ctx.profile(True)
    ctx.phase_reset()
    t0 = time.perf_counter()
    step()
    ctx.sync()
    dt = time.perf_counter() - t0
    ph = {k: v[0] for k, v in ctx.phase_times().items()}
    ctx.profile(False)
    Sh = S.to_host()[:, 0]
    r0 = min(rank * ((n + world - 1) // world), n)
    Zh = Z.to_host()[:1000] if world == 1 else Z.to_host()[:max(0, min(1000 - r0, Z.shape[0]))]     # the first 1000 rows of Z
    np.savez(os.path.join(tmp, f"out_{world}_{rank}.npz"), S=Sh, Z=Zh, ms=1e3 * dt, phases=json.dumps(ph), n=n)
    for m in (Om, Z, S, op):
        m.close()
    ctx.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--worlds", type=int, nargs="*", default=[2, 4])
    ap.add_argument("--case", nargs="*", default=["fft3d", "fft2d", "lowrank", "implicit"])
    ap.add_argument("--world", type=int, default=0)
    ap.add_argument("--rank", type=int, default=0)
    ap.add_argument("--ranks-of-omega", type=int, default=1)
    ap.add_argument("--tmp", default="")
    args = ap.parse_args()
    if args.world > 0:
        args.case = args.case[0]
        return child(args)
    import numpy as np
    for case in args.case:
        _, K, p, q = CASES[case]
        for world in args.worlds:
            with tempfile.TemporaryDirectory() as tmp:
                outs = {}
                for w in (1, world):        # the reference: ONE rank, no communicator, the same Omega
                    env = dict(os.environ)
                    if w > 1:
                        env["GSI_SHM_COMM"] = "1"
                        env["GSI_SHM_TIMEOUT_S"] = "300"
                    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--world", str(w), "--rank", str(r),
                                               "--ranks-of-omega", str(world), "--tmp", tmp, "--case", case], env=env)
                             for r in range(w)]
                    rcs = [pr.wait(timeout=900) for pr in procs]
                    if any(rcs):
                        print(json.dumps({"case": case, "world": w, "error": f"exit codes {rcs}"}), flush=True)
                        outs = None
                        break
                    outs[w] = [np.load(os.path.join(tmp, f"out_{w}_{r}.npz")) for r in range(w)]
                if not outs:
                    continue
                S1, SN = outs[1][0]["S"], outs[world][0]["S"]
                keep = S1[:K] > 1e-10 * S1[0]
                sv = float(np.max(np.abs(SN[:K][keep] - S1[:K][keep]) / S1[:K][keep]))
                Z1 = outs[1][0]["Z"]
                ZN = np.concatenate([o["Z"] for o in outs[world]], axis=0)[:Z1.shape[0]]
                zerr = max(min(np.linalg.norm(ZN[:, i] - Z1[:, i]), np.linalg.norm(ZN[:, i] + Z1[:, i])) for i in range(K))
                # degeneracy-robust: Z Z' restricted to the sampled rows is the rank-K' approximation of A there, the same for
                # every basis of a degenerate eigenspace -- K' = the last index <= K with a spectral gap behind it (a cut
                # through a multiplet would make the truncated sum itself ambiguous)
                Kc = K
                while Kc > 1 and (S1[Kc - 1] - S1[Kc]) < 1e-6 * S1[0]:
                    Kc -= 1
                P1, PN = Z1[:, :Kc] @ Z1[:, :Kc].T, ZN[:, :Kc] @ ZN[:, :Kc].T
                proj = float(np.linalg.norm(PN - P1) / np.linalg.norm(P1))
                mult = int(sum(1 for i in range(K - 1) if abs(S1[i] - S1[i + 1]) < 1e-9 * S1[0]))
                print(json.dumps({"case": case, "n": int(outs[1][0]["n"]), "K": K, "p": p, "q": q, "ranks_on_one_gpu": world,
                                  "sv_rel_diff_vs_one_rank": sv, "xis_diff_up_to_sign_first_1000_rows": float(zerr),
                                  "ZZt_rel_diff_first_1000_rows": proj, "ZZt_columns": Kc,
                                  "near_equal_neighbouring_singular_values": mult,
                                  "all_ranks_same_S": bool(all(np.array_equal(o["S"], SN) for o in outs[world])),
                                  "ms_per_step_one_rank": float(outs[1][0]["ms"]),
                                  "ms_per_step_ranks_sharing_the_gpu": max(float(o["ms"]) for o in outs[world]),
                                  "phases_ms_one_rank": json.loads(str(outs[1][0]["phases"])),
                                  "phases_ms_rank0": json.loads(str(outs[world][0]["phases"]))}), flush=True)


if __name__ == "__main__":
    main()
