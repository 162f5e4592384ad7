"""Per-level kernel timing of a BoomerAMG hierarchy (HIP events on the library's stream).

    python tools/bench_levels.py [n] [reps]

Builds the benchmark hierarchy (n^3 7-pt Laplacian, PMIS / ext+i(4) / l1-Jacobi) and times, level
by level, y = A x, the fused l1-Jacobi sweep, the prolongation P x and the restriction P^T x,
printing algorithmic GB/s (SURVEY.md 8d byte formulas).  Tuning knobs (HYPRE_AMD_SPMV_*) are read
once per process, so A/B runs are separate invocations.
"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hypre_amd import binding as B, ij   # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
L = B.load_library()
opt = ij.IJOptions(n=(n, n, n), coarsen_type=8, interp_type=6, P_max_elmts=4, relax_type=18, num_sweeps=1)
t0 = time.time()
A = ij.build_matrix(opt)
s = ij.create_amg(opt, memory_location=B.HYPRE_MEMORY_DEVICE)
L.HYPRE_BoomerAMGSetup(s, A, None, None)
B.check()
L.hypre_ParCSRMatrixMigrate(A, B.HYPRE_MEMORY_DEVICE)     # level 0 is the caller's matrix
print("setup %.1fs" % (time.time() - t0), flush=True)
L.hypre_SetSyncCudaCompute(0)
nl = L.hypre_amd_BoomerAMGGetNumLevels(s)


def timed(fn):
    for _ in range(3):
        fn()
    L.hypre_SyncComputeStream()
    L.hypre_amd_EventTimerStart()
    for _ in range(reps):
        fn()
    return L.hypre_amd_EventTimerStopMs() / reps


tot = 0.0
for l in range(nl):
    Al = C.cast(L.hypre_amd_BoomerAMGGetA(s, l), C.POINTER(B.ParCSRMatrix))
    d = Al.contents.diag.contents
    nr, nnz = d.num_rows, d.num_nonzeros
    x = B.parvec_from_numpy(np.random.default_rng(l).uniform(-1, 1, nr))
    y = B.parvec_from_numpy(np.zeros(nr))
    f = B.parvec_from_numpy(np.ones(nr))
    v = B.parvec_from_numpy(np.zeros(nr))
    l1p = L.hypre_amd_BoomerAMGGetL1Norms(s, l)
    l1 = C.cast(l1p, C.POINTER(B.Vector)).contents.data if l1p else None
    ms_a = timed(lambda: L.hypre_ParCSRMatrixMatvec(1.0, Al, x, 0.0, y))
    by_a = nnz * 12 + (nr + 1) * 4 + nr * 16
    line = "L%d rows=%9d nnz=%10d (%.1f/row)  A x: %.4f ms %6.0f GB/s" % (l, nr, nnz, nnz / max(nr, 1), ms_a, by_a / ms_a / 1e6)
    if l1 is not None and l < nl - 1:
        ms_j = timed(lambda: L.hypre_BoomerAMGRelax(Al, f, None, 18, 0, 1.0, 1.0, l1, x, v, v))
        by_j = nnz * 12 + (nr + 1) * 4 + nr * 32
        line += " | l1-Jacobi (+copy back): %.4f ms %6.0f GB/s" % (ms_j, (by_j + nr * 16) / ms_j / 1e6)
        tot += ms_j
    if l < nl - 1:
        Pl = C.cast(L.hypre_amd_BoomerAMGGetP(s, l), C.POINTER(B.ParCSRMatrix))
        pd = Pl.contents.diag.contents
        nc = pd.num_cols
        xc = B.parvec_from_numpy(np.random.default_rng(l + 50).uniform(-1, 1, nc))
        yc = B.parvec_from_numpy(np.zeros(nc))
        ms_p = timed(lambda: L.hypre_ParCSRMatrixMatvec(1.0, Pl, xc, 1.0, y))
        ms_r = timed(lambda: L.hypre_ParCSRMatrixMatvecT(1.0, Pl, x, 0.0, yc))
        by_p = pd.num_nonzeros * 12 + (nr + 1) * 4 + nc * 8 + nr * 16
        by_r = pd.num_nonzeros * 12 + (nc + 1) * 4 + nr * 8 + nc * 8
        line += " | P x: %.4f ms %6.0f GB/s | P^T x: %.4f ms %6.0f GB/s" % (ms_p, by_p / ms_p / 1e6, ms_r, by_r / ms_r / 1e6)
        tot += ms_p + ms_r
    tot += ms_a
    print(line, flush=True)
B.check()
print("sum of timed kernels %.3f ms" % tot)
