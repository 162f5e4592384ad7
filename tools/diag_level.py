"""Diagnostic: how much of a level's SpMV time is the x gather?  Times y = A x on level L's diag block
as it is, with every column replaced by the row's own index (perfect gather locality, same stream), and
with columns sorted-random within a +-4096 band (no structure).   python tools/diag_level.py [n] [level]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hypre_amd import binding as B, ij   # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
lev = int(sys.argv[2]) if len(sys.argv) > 2 else 1
L = B.load_library()
opt = ij.IJOptions(n=(n, n, n), coarsen_type=8, interp_type=6, P_max_elmts=4, relax_type=18)
A = ij.build_matrix(opt)
s = ij.create_amg(opt, memory_location=B.HYPRE_MEMORY_DEVICE)
L.HYPRE_BoomerAMGSetup(s, A, None, None)
B.check()
L.hypre_ParCSRMatrixMigrate(A, B.HYPRE_MEMORY_DEVICE)
L.hypre_SetSyncCudaCompute(0)
Al = C.cast(L.hypre_amd_BoomerAMGGetA(s, lev), C.POINTER(B.ParCSRMatrix))
ii, jj, aa = B.csr_to_arrays(Al.contents.diag)
nr = len(ii) - 1
nnz = len(jj)
rows = np.repeat(np.arange(nr, dtype=np.int32), np.diff(ii))


def timed(mat, reps=20):
    x = B.vec_from_numpy(np.random.default_rng(0).uniform(-1, 1, nr))
    y = B.vec_from_numpy(np.zeros(nr))
    for _ in range(3):
        L.hypre_CSRMatrixMatvec(1.0, mat, x, 0.0, y)
    L.hypre_SyncComputeStream()
    L.hypre_amd_EventTimerStart()
    for _ in range(reps):
        L.hypre_CSRMatrixMatvec(1.0, mat, x, 0.0, y)
    return L.hypre_amd_EventTimerStopMs() / reps


byt = nnz * 12 + (nr + 1) * 4 + nr * 16
print("level %d: rows %d nnz %d (%.1f/row)" % (lev, nr, nnz, nnz / nr))
for name, cols in (("as is", jj), ("own row", rows),
                   ("row + k (consecutive)", np.minimum(rows + (np.arange(nnz) - ii[rows]).astype(np.int32), nr - 1))):
    m = B.csr_from_arrays(nr, nr, ii, cols.astype(np.int32), aa)
    ms = timed(m)
    print("  %-24s %.4f ms  %6.0f GB/s" % (name, ms, byt / ms / 1e6), flush=True)
B.check()
