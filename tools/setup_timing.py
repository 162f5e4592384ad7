"""Host setup time per level and phase (HYPRE_AMD_SETUP_TIMING) for an n^3 7-point problem, single rank."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("HYPRE_AMD_SETUP_TIMING", "1")
from hypre_amd import binding as B, ij  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
L = B.load_library()
opt = ij.IJOptions(n=(n, n, n), coarsen_type=8, relax_type=18)
t = time.time()
A = ij.build_matrix(opt)
print("generate %.2f s" % (time.time() - t), flush=True)
s = ij.create_amg(opt, memory_location=B.HYPRE_MEMORY_HOST)
t = time.time()
L.HYPRE_BoomerAMGSetup(s, A, None, None)
B.check()
print("setup %.2f s" % (time.time() - t), flush=True)
