"""Where a tile of spmv_xs_kernel spends its life (library built with -DXS_TIMING=1, tools/experiments/build_variant.sh):
mean ticks (10 ns) per tile between the kernel's entry, the return of the scalar batch, the first barrier (stream and x
pieces landed), the products parked and the end, for A x on levels 0 and 1 of the 256^3 hierarchy."""
import ctypes as C
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hypre_amd import binding as B, ij   # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
problem = sys.argv[2] if len(sys.argv) > 2 else "laplacian"
L = B.load_library()
L.hypre_amd_XsTiming.argtypes = [C.c_void_p, C.c_int]
opt = ij.IJOptions(n=(n, n, n), coarsen_type=8, interp_type=6, P_max_elmts=4, relax_type=18, num_sweeps=1, problem=problem)
A = ij.build_matrix(opt)
s = ij.create_amg(opt, memory_location=B.HYPRE_MEMORY_DEVICE)
L.hypre_ParCSRMatrixMigrate(A, B.HYPRE_MEMORY_DEVICE)
L.HYPRE_BoomerAMGSetup(s, A, None, None)
B.check()
L.hypre_SetSyncCudaCompute(0)
for l in range(3):
    Al = C.cast(L.hypre_amd_BoomerAMGGetA(s, l), C.POINTER(B.ParCSRMatrix))
    nr = Al.contents.diag.contents.num_rows
    x = B.parvec_from_numpy(np.random.default_rng(l).uniform(-1, 1, nr))
    y = B.parvec_from_numpy(np.zeros(nr))
    nt = C.c_int()
    mp = C.c_double(); L.hypre_amd_CSRMatrixPlanStaging(Al.contents.diag, C.byref(nt), C.byref(mp))
    for _ in range(3):
        L.hypre_ParCSRMatrixMatvec(1.0, Al, x, 0.0, y)
    L.hypre_amd_XsTiming(None, nt.value)
    L.hypre_SyncComputeStream()
    L.hypre_amd_EventTimerStart()
    for _ in range(20):
        L.hypre_ParCSRMatrixMatvec(1.0, Al, x, 0.0, y)
    ms = L.hypre_amd_EventTimerStopMs() / 20
    t = np.zeros((nt.value, 4), dtype=np.uint32)
    L.hypre_amd_XsTiming(t.ctypes.data, 0)
    t = t[t.sum(axis=1) > 0].astype(np.float64) * 10.0
    ph = t.mean(axis=0)
    p90 = np.percentile(t, 90, axis=0)
    print("L%d rows %d: %.4f ms per product, %d tiles; per tile, ns, mean (p90): scalar batch %.0f (%.0f) | stream + pieces %.0f (%.0f) | "
          "gathers, products %.0f (%.0f) | row sums, epilogue %.0f (%.0f) | life %.0f" % (l, nr, ms, len(t), ph[0], p90[0], ph[1], p90[1], ph[2], p90[2], ph[3], p90[3], ph.sum()), flush=True)
