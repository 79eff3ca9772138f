#!/usr/bin/env python3
"""S'X of the LowRankCovMatrix product two ways: as it runs now -- the TN contraction over the n x N_s sample matrix (columns
8 MB apart) -- and as an NN contraction over a TRANSPOSED copy (N_s x n: a grid point's N_s values contiguous).  Same flops,
same bytes; only the operand's layout differs.   python tools/tn_vs_nn_layout.py [n] [N_s] [l]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gsi_amd as gsi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
Ns = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
l = int(sys.argv[3]) if len(sys.argv) > 3 else 320
ctx = gsi.Context(0)
lib = ctx.lib
rng = np.random.default_rng(0)
St = np.asfortranarray(rng.standard_normal((Ns, n)))                 # N_s x n column-major = the transposed copy
opT = gsi.dense_operator(ctx, St)                                    # A = St: A*X is the NN contraction, M = N_s, K = n
X = gsi.DeviceMatrix(ctx, n, l).randn(1)
Y = gsi.DeviceMatrix(ctx, Ns, l)
def time_it(op, trans, Xm, Ym, key):
    gsi._lib.check(lib.gsi_op_mul_dev(ctx.h, op.h, trans, Xm.h, Ym.h), lib); ctx.sync()
    ctx.profile(True); ctx.phase_reset()
    for _ in range(5):
        gsi._lib.check(lib.gsi_op_mul_dev(ctx.h, op.h, trans, Xm.h, Ym.h), lib)
    ph = ctx.phase_times(); ctx.profile(False)
    return ph[key][0] / ph[key][1]
ms_nn = time_it(opT, 0, X, Y, "gemm_n")
Ynn = Y.to_host()
opT.close()
S = np.asfortranarray(St.T)                                           # n x N_s column-major: today's layout
del St
op = gsi.dense_operator(ctx, S)                                      # A = S: A'*X is the TN contraction
ms_tn = time_it(op, 1, X, Y, "gemm_t")
Ytn = Y.to_host()
fl = 2.0 * n * Ns * l
print(f"n={n} N_s={Ns} l={l}: TN over the n x N_s matrix {ms_tn:.3f} ms ({fl/ms_tn/1e9:.1f} TFLOP/s); "
      f"NN over the transposed copy {ms_nn:.3f} ms ({fl/ms_nn/1e9:.1f} TFLOP/s); max diff {np.abs(Ynn-Ytn).max():.2e}", flush=True)
