"""Controlled version of sorted_rows.py: the SAME allocation before and after (the setup leaves the first-touch order,
HYPRE_AMD_SORT_COARSE_ROWS=0; the rows are then sorted IN PLACE), several timing rounds each: does the column order inside a
row matter to the x-staged kernel, apart from where an allocation happens to land in memory?"""
import ctypes as C
import os
import sys

import numpy as np

os.environ["HYPRE_AMD_SORT_COARSE_ROWS"] = "0"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hypre_amd import binding as B, ij   # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
L = B.load_library()
opt = ij.IJOptions(n=(n, n, n), coarsen_type=8, interp_type=6, P_max_elmts=4, relax_type=18, num_sweeps=1)
A = ij.build_matrix(opt)
L.hypre_ParCSRMatrixMigrate(A, B.HYPRE_MEMORY_DEVICE)
s = ij.create_amg(opt, memory_location=B.HYPRE_MEMORY_DEVICE)
L.HYPRE_BoomerAMGSetup(s, A, None, None)
B.check()
L.hypre_SetSyncCudaCompute(0)


def timed(fn, reps=30):
    for _ in range(5):
        fn()
    L.hypre_SyncComputeStream()
    L.hypre_amd_EventTimerStart()
    for _ in range(reps):
        fn()
    return L.hypre_amd_EventTimerStopMs() / reps


for level in (1, 2, 3):
    Al = C.cast(L.hypre_amd_BoomerAMGGetA(s, level), C.POINTER(B.ParCSRMatrix))
    d = Al.contents.diag
    nr = d.contents.num_rows
    x = B.vec_from_numpy(np.random.default_rng(level).uniform(-1, 1, nr))
    y = B.vec_from_numpy(np.zeros(nr))
    before = [timed(lambda: L.hypre_CSRMatrixMatvec(1.0, d, x, 0.0, y)) for _ in range(4)]
    y0 = B.vec_to_numpy(y)
    L.hypre_amd_CSRMatrixSortRows(d, 1)
    B.check()
    after = [timed(lambda: L.hypre_CSRMatrixMatvec(1.0, d, x, 0.0, y)) for _ in range(4)]
    err = float(np.max(np.abs(B.vec_to_numpy(y) - y0)) / np.max(np.abs(y0)))
    print("level %d rows %d: first-touch order %s ms | sorted in place %s ms | products differ by %.1e"
          % (level, nr, " ".join("%.4f" % t for t in before), " ".join("%.4f" % t for t in after), err), flush=True)
