"""Where the one-workgroup tail spends its time (library variant built with -DTAIL_TIMING=1:
tools/build_variant_lib.sh tailtiming tail_kernels.hip -DTAIL_TIMING=1; HYPRE_AMD_LIB=hypre_amd/lib/libhypre_amd_tailtiming.so)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from hypre_amd import binding as B, ij
L = B.load_library()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
opt = ij.IJOptions(n=(n, n, n), coarsen_type=8, interp_type=6, P_max_elmts=4, relax_type=18, num_sweeps=1)
A = ij.build_matrix(opt)
L.hypre_ParCSRMatrixMigrate(A, B.HYPRE_MEMORY_DEVICE)
s = ij.create_amg(opt, memory_location=B.HYPRE_MEMORY_DEVICE)
L.HYPRE_BoomerAMGSetup(s, A, None, None)
L.HYPRE_BoomerAMGSetTol(s, 0.0); L.HYPRE_BoomerAMGSetMaxIter(s, 1)
b = B.parvec_from_numpy(np.ones(n ** 3)); u = B.parvec_from_numpy(np.zeros(n ** 3))
raw = C.CDLL(os.environ["HYPRE_AMD_LIB"])
for it in range(6):
    L.hypre_ParVectorSetZeros(u)
    L.HYPRE_BoomerAMGSolve(s, A, b, u)
    st = (C.c_ulonglong * 64)()
    raw.hypre_amd_TailTiming(st)
    v = [int(x) for x in st]
    t0 = v[0]
    print("cycle %d tail level %d:" % (it, L.hypre_amd_BoomerAMGGetSmallTailLevel(s)),
          " ".join("%d:%.2f" % (k, (x - t0) / 100.0) for k, x in enumerate(v) if x >= t0 and x - t0 < 10 ** 7), flush=True)
B.check()
