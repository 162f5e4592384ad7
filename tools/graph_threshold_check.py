"""V-cycle time of the benchmark hierarchy with the coarse tail recorded as a HIP graph from different levels down.

    python tools/graph_threshold_check.py [n] [rows...]      rows: hypre_amd_BoomerAMGSetGraphThreshold values (0 = no graph)
"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hypre_amd import binding as B, ij   # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
relax = [int(x) for x in os.environ.get("RELAX", "18,18").split(",")]
rows = [int(a) for a in sys.argv[2:]] or [0, 100000, 1000000, 100000000]
L = B.load_library()
opt = ij.IJOptions(n=(n, n, n), coarsen_type=8, interp_type=6, P_max_elmts=4, relax_type=relax[0], num_sweeps=1)
opt.relax_down, opt.relax_up = relax[0], relax[1]
A = ij.build_matrix(opt)
L.hypre_ParCSRMatrixMigrate(A, B.HYPRE_MEMORY_DEVICE)
s = ij.create_amg(opt, memory_location=B.HYPRE_MEMORY_DEVICE)
L.HYPRE_BoomerAMGSetup(s, A, None, None)
B.check()
L.hypre_SetSyncCudaCompute(0)
L.HYPRE_BoomerAMGSetTol(s, 0.0)
L.HYPRE_BoomerAMGSetMaxIter(s, 1)
b = B.parvec_from_numpy(np.ones(n ** 3))
u = B.parvec_from_numpy(np.zeros(n ** 3))


def cycle():
    L.hypre_ParVectorSetZeros(u)
    L.HYPRE_BoomerAMGSolve(s, A, b, u)


for r in rows:
    L.hypre_amd_BoomerAMGSetGraphThreshold(s, r)
    for _ in range(4):
        cycle()
    L.hypre_SyncComputeStream()
    L.hypre_amd_EventTimerStart()
    for _ in range(20):
        cycle()
    ms = L.hypre_amd_EventTimerStopMs() / 20
    lev, nodes = C.c_int(), C.c_int()
    L.hypre_amd_BoomerAMGGetGraphInfo(s, C.byref(lev), C.byref(nodes))
    print("graph threshold %10d rows: from level %2d, %3d nodes, V-cycle %.4f ms" % (r, lev.value, nodes.value, ms), flush=True)
