"""Experiment: does sorting the columns inside each row of the level-1 operator (diagonal kept in front) make the x-staged
kernel faster?  (A lane's consecutive entries would then read consecutive staged positions: fewer LDS bank conflicts.)"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hypre_amd import binding as B, ij   # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
level = int(sys.argv[2]) if len(sys.argv) > 2 else 1
L = B.load_library()
opt = ij.IJOptions(n=(n, n, n), coarsen_type=8, interp_type=6, P_max_elmts=4, relax_type=18, num_sweeps=1)
A = ij.build_matrix(opt)
L.hypre_ParCSRMatrixMigrate(A, B.HYPRE_MEMORY_DEVICE)
s = ij.create_amg(opt, memory_location=B.HYPRE_MEMORY_DEVICE)
L.HYPRE_BoomerAMGSetup(s, A, None, None)
B.check()
L.hypre_SetSyncCudaCompute(0)
Al = C.cast(L.hypre_amd_BoomerAMGGetA(s, level), C.POINTER(B.ParCSRMatrix))
ii, jj, aa = B.csr_to_arrays(Al.contents.diag)
nr = len(ii) - 1
print("level", level, "rows", nr, "nnz", len(jj), flush=True)


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    L.hypre_SyncComputeStream()
    L.hypre_amd_EventTimerStart()
    for _ in range(reps):
        fn()
    return L.hypre_amd_EventTimerStopMs() / reps


x = B.vec_from_numpy(np.random.default_rng(1).uniform(-1, 1, nr))
y = B.vec_from_numpy(np.zeros(nr))
d0 = Al.contents.diag
print("stored order: %.4f ms" % timed(lambda: L.hypre_CSRMatrixMatvec(1.0, d0, x, 0.0, y)), flush=True)
y0 = B.vec_to_numpy(y)
# sort columns inside each row, diagonal (first entry) stays in front
t0 = time.time()
rows = np.repeat(np.arange(nr, dtype=np.int64), np.diff(ii))
key = jj.astype(np.int64).copy()
key[ii[:-1]] = -1                                     # diagonal first
order = np.lexsort((key, rows))
js, as_ = jj[order], aa[order]
print("sorted on the host in %.1fs" % (time.time() - t0), flush=True)
dS = B.csr_from_arrays(nr, nr, ii, js, as_)
print("sorted columns: %.4f ms" % timed(lambda: L.hypre_CSRMatrixMatvec(1.0, dS, x, 0.0, y)), flush=True)
print("difference of the products: %.2e" % float(np.max(np.abs(B.vec_to_numpy(y) - y0)) / np.max(np.abs(y0))))
nt, mp = C.c_int(), C.c_double()
for name, m in (("stored", d0), ("sorted", dS)):
    st = L.hypre_amd_CSRMatrixPlanStaging(m, C.byref(nt), C.byref(mp))
    print(name, "tiles", nt.value, "staged", st, "pieces %.1f" % mp.value)
