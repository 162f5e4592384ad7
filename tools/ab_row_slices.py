"""A/B of the V(1,1) cycle with and without the row-slice form (or the cycle's fusions: --toggle fusion; or the one-workgroup tail: --toggle smalltail) in ONE process on ONE box (the HBM-bound launches move by
4-7 % from box to box and drift with the clocks inside a run: the two forms are measured alternately, four times each).

    python tools/ab_row_slices.py [n] [cycles] [--problem laplacian|27pt|difconv] [--relax 18]
"""
import argparse
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hypre_amd import binding as B, ij   # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("n", type=int, nargs="?", default=256)
ap.add_argument("cycles", type=int, nargs="?", default=30)
ap.add_argument("--problem", default="laplacian")
ap.add_argument("--relax", type=int, default=18)
ap.add_argument("--codes", type=int, default=1)
ap.add_argument("--toggle", default="rowslices", choices=["rowslices", "fusion", "smalltail"], help="what is switched between the two samples")
args = ap.parse_args()
L = B.load_library()
n = args.n
opt = ij.IJOptions(n=(n, n, n), coarsen_type=8, interp_type=6, P_max_elmts=4, relax_type=args.relax, num_sweeps=1, problem=args.problem)
A = ij.build_matrix(opt)
L.hypre_ParCSRMatrixMigrate(A, B.HYPRE_MEMORY_DEVICE)
L.hypre_amd_SpmvSetValueCodes(args.codes)
b = B.parvec_from_numpy(np.ones(n ** 3))
u = B.parvec_from_numpy(np.zeros(n ** 3))
res = {0: [], 1: []}
for rep in range(4):
    for mode in (1, 0):
        if args.toggle == "rowslices":
            L.hypre_amd_SpmvSetRowSlices(mode)
        elif args.toggle == "smalltail":
            L.hypre_amd_SetSmallTail(mode)
        else:
            L.hypre_amd_SetCycleFusion(mode)
        s = ij.create_amg(opt, memory_location=B.HYPRE_MEMORY_DEVICE)
        L.HYPRE_BoomerAMGSetup(s, A, None, None)
        B.check()
        L.HYPRE_BoomerAMGSetTol(s, 0.0)
        L.HYPRE_BoomerAMGSetMaxIter(s, 1)
        L.hypre_SetSyncCudaCompute(0)
        for _ in range(5):
            L.hypre_ParVectorSetZeros(u)
            L.HYPRE_BoomerAMGSolve(s, A, b, u)
        L.hypre_SyncComputeStream()
        t0 = time.perf_counter()
        for _ in range(args.cycles):
            L.hypre_ParVectorSetZeros(u)
            L.HYPRE_BoomerAMGSolve(s, A, b, u)
        L.hypre_SyncComputeStream()
        ms = 1e3 * (time.perf_counter() - t0) / args.cycles
        res[mode].append(ms)
        L.hypre_SetSyncCudaCompute(1)
        L.HYPRE_BoomerAMGDestroy(s)
        B.check()
        print("rep %d %s %d: %.4f ms per cycle" % (rep, args.toggle, mode, ms), flush=True)
for mode in (1, 0):
    v = sorted(res[mode])
    print("%s %s: median %.4f ms  (min %.4f, max %.4f)" % (args.toggle, "on " if mode else "off", (v[1] + v[2]) / 2, v[0], v[-1]))
