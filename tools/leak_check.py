import ctypes as C, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from hypre_amd import binding as B, ij
L = B.load_library()
hip = C.CDLL("libamdhip64.so")
def free_mb():
    f, t = C.c_size_t(), C.c_size_t()
    hip.hipMemGetInfo(C.byref(f), C.byref(t))
    return f.value / 2**20
for relax, up in ((18, 18), (11, 11), (21, 22), (13, 14)):
    base = None
    for it in range(int(os.environ.get("LEAK_ROUNDS", "8"))):
        opt = ij.IJOptions(n=(48, 48, 48), coarsen_type=8, interp_type=6, P_max_elmts=4, relax_type=relax, num_sweeps=1)
        opt.relax_down, opt.relax_up = relax, up
        A = ij.build_matrix(opt)
        if it % 2: L.hypre_ParCSRMatrixMigrate(A, B.HYPRE_MEMORY_DEVICE)
        s = ij.create_amg(opt, memory_location=B.HYPRE_MEMORY_DEVICE)
        L.HYPRE_BoomerAMGSetup(s, A, None, None)
        L.hypre_ParCSRMatrixMigrate(A, B.HYPRE_MEMORY_DEVICE)
        n = 48 ** 3
        b = B.parvec_from_numpy(np.ones(n)); u = B.parvec_from_numpy(np.zeros(n))
        L.HYPRE_BoomerAMGSetTol(s, 1e-8); L.HYPRE_BoomerAMGSetMaxIter(s, 30)
        L.HYPRE_BoomerAMGSolve(s, A, b, u)
        L.HYPRE_ClearAllErrors()
        L.HYPRE_BoomerAMGDestroy(s)
        L.hypre_ParVectorDestroy(b); L.hypre_ParVectorDestroy(u)
        L.hypre_ParCSRMatrixDestroy(A)
        L.hypre_SyncComputeStream()
        f = free_mb()
        if it == 2: base = f
        if base is None or it % 10 == 9 or it < 4:
            print("relax %d/%d round %d: free %.1f MiB%s" % (relax, up, it, f, "" if base is None else "  (drift %.1f)" % (base - f)), flush=True)
    print("relax %d/%d: drift of free device memory between round 2 and the last: %.1f MiB" % (relax, up, base - f), flush=True)
