"""y = A_l x of one level of the benchmark hierarchy, `reps` times back to back (for rocprofv3 --pmc runs).

    python tools/bench_level_spmv.py [n] [level] [reps]
"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hypre_amd import binding as B, ij   # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
level = int(sys.argv[2]) if len(sys.argv) > 2 else 1
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
L = B.load_library()
opt = ij.IJOptions(n=(n, n, n), coarsen_type=8, interp_type=6, P_max_elmts=4, relax_type=18, num_sweeps=1)
A = ij.build_matrix(opt)
s = ij.create_amg(opt, memory_location=B.HYPRE_MEMORY_DEVICE)
L.HYPRE_BoomerAMGSetup(s, A, None, None)
B.check()
L.hypre_ParCSRMatrixMigrate(A, B.HYPRE_MEMORY_DEVICE)
L.hypre_SetSyncCudaCompute(0)
Al = C.cast(L.hypre_amd_BoomerAMGGetA(s, level), C.POINTER(B.ParCSRMatrix))
d = Al.contents.diag.contents
nr, nnz = d.num_rows, d.num_nonzeros
x = B.parvec_from_numpy(np.random.default_rng(level).uniform(-1, 1, nr))
y = B.parvec_from_numpy(np.zeros(nr))
for _ in range(3):
    L.hypre_ParCSRMatrixMatvec(1.0, Al, x, 0.0, y)
L.hypre_SyncComputeStream()
L.hypre_amd_EventTimerStart()
for _ in range(reps):
    L.hypre_ParCSRMatrixMatvec(1.0, Al, x, 0.0, y)
ms = L.hypre_amd_EventTimerStopMs() / reps
by = nnz * 12 + (nr + 1) * 4 + nr * 16
print("level %d rows %d nnz %d (%.1f/row): %.4f ms  %.0f GB/s" % (level, nr, nnz, nnz / nr, ms, by / ms / 1e6), flush=True)
