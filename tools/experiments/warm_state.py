"""How long does the device take to reach its steady rate?  The V-cycle of the benchmark timed in consecutive blocks of 20."""
import ctypes as C
import os
import sys
import time

import numpy as np

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
nn = n ** 3
b = B.parvec_from_numpy(np.ones(nn))
u = B.parvec_from_numpy(np.zeros(nn))
L.HYPRE_BoomerAMGSetTol(s, 0.0)
L.HYPRE_BoomerAMGSetMaxIter(s, 1)


def step():
    L.hypre_ParVectorSetZeros(u)
    L.HYPRE_BoomerAMGSolve(s, A, b, u)


for _ in range(3):
    step()
L.hypre_SyncComputeStream()
out = []
for blk in range(12):
    t0 = time.perf_counter()
    for _ in range(20):
        step()
    L.hypre_SyncComputeStream()
    out.append(1e3 * (time.perf_counter() - t0) / 20)
print("ms per cycle, consecutive blocks of 20:", " ".join("%.3f" % t for t in out), flush=True)
time.sleep(2.0)
out = []
for blk in range(4):
    t0 = time.perf_counter()
    for _ in range(20):
        step()
    L.hypre_SyncComputeStream()
    out.append(1e3 * (time.perf_counter() - t0) / 20)
print("after 2 s of idling:", " ".join("%.3f" % t for t in out), flush=True)
