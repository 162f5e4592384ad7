"""Wall time of HYPRE_BoomerAMGSetup alone on the benchmark problem, matrix handed over in host or device memory.

    HYPRE_AMD_SETUP_TIMING=1 python tools/setup_time.py [n] [host|device] [repeats]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hypre_amd import binding as B, ij   # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
where = sys.argv[2] if len(sys.argv) > 2 else "device"
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
L = B.load_library()
opt = ij.IJOptions(n=(n, n, n), coarsen_type=8, interp_type=6, P_max_elmts=4, relax_type=18, num_sweeps=1)
t0 = time.time()
A = ij.build_matrix(opt)
t1 = time.time()
if where == "device":
    L.hypre_ParCSRMatrixMigrate(A, B.HYPRE_MEMORY_DEVICE)
    L.hypre_SyncComputeStream()
t2 = time.time()
print("matrix generation %.2f s, upload %.2f s" % (t1 - t0, t2 - t1), flush=True)
for r in range(reps):
    s = ij.create_amg(opt, memory_location=B.HYPRE_MEMORY_DEVICE)
    t = time.time()
    L.HYPRE_BoomerAMGSetup(s, A, None, None)
    L.hypre_SyncComputeStream()
    B.check()
    print("setup %d (matrix in %s memory): %.3f s, %d levels" % (r, where, time.time() - t, L.hypre_amd_BoomerAMGGetNumLevels(s)), flush=True)
    L.HYPRE_BoomerAMGDestroy(s)
