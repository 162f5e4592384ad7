"""y = A_l x of the first levels of the benchmark hierarchy, `reps` launches each, back to back (for rocprofv3 runs:
one process covers every level, the launches of a level are told apart by their grid size).

    python tools/bench_levels_spmv.py [n] [levels] [reps] [variant[:wgs]]
"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hypre_amd import binding as B, ij   # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
levels = int(sys.argv[2]) if len(sys.argv) > 2 else 3
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
var = sys.argv[4].split(":") if len(sys.argv) > 4 else ["0"]
L = B.load_library()
L.hypre_amd_SpmvSetVariant(int(var[0]), int(var[1]) if len(var) > 1 else 0)
opt = ij.IJOptions(n=(n, n, n), coarsen_type=8, interp_type=6, P_max_elmts=4, relax_type=18, num_sweeps=1)
A = ij.build_matrix(opt)
s = ij.create_amg(opt, memory_location=B.HYPRE_MEMORY_DEVICE)
L.HYPRE_BoomerAMGSetup(s, A, None, None)
B.check()
L.hypre_ParCSRMatrixMigrate(A, B.HYPRE_MEMORY_DEVICE)
L.hypre_SetSyncCudaCompute(0)
for level in range(levels):
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
    nt = C.c_int()
    L.hypre_amd_CSRMatrixPlanInfo(Al.contents.diag, C.byref(nt), None)
    lanes = L.hypre_amd_CSRMatrixPlanSliceForm(Al.contents.diag)
    if lanes:                                      # slice form: one workgroup per 256 / lanes rows
        nt.value = -(-nr // (256 // lanes))
    rsr = C.c_int()
    if L.hypre_amd_CSRMatrixPlanRowSlices(Al.contents.diag, C.byref(rsr), None):      # row slices: one workgroup per block of rows
        nt.value = -(-nr // rsr.value)
    print("LEVEL %d rows %d nnz %d tiles %d bytes %d : %.4f ms  %.0f GB/s" % (level, nr, nnz, nt.value, by, ms, by / ms / 1e6), flush=True)
