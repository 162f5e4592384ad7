"""Fine-level SpMV micro-benchmark (HIP events on the library's compute stream)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hypre_amd import binding as B

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
L = B.load_library()
t0 = time.time()
A = B.laplacian(n, n, n)
print("generate %.1fs" % (time.time() - t0), flush=True)
if os.environ.get("SPMV_LOCAL_COLS"):
    # diagnostic: every entry points at its own row (no x re-use distance), same stream bytes
    d = A.contents.diag.contents
    ii = np.ctypeslib.as_array(d.i, shape=(d.num_rows + 1,))
    jj = np.ctypeslib.as_array(d.j, shape=(d.num_nonzeros,))
    jj[:] = np.repeat(np.arange(d.num_rows, dtype=np.int32), np.diff(ii))
L.hypre_ParCSRMatrixMigrate(A, B.HYPRE_MEMORY_DEVICE)
diag = A.contents.diag
nr, nnz = diag.contents.num_rows, diag.contents.num_nonzeros
x = B.vec_from_numpy(np.random.default_rng(0).uniform(-1, 1, nr))
y = B.vec_from_numpy(np.zeros(nr))
L.hypre_SetSyncCudaCompute(0)
for _ in range(5):
    L.hypre_CSRMatrixMatvec(1.0, diag, x, 0.0, y)
L.hypre_SyncComputeStream()
B.check()
L.hypre_amd_EventTimerStart()
for _ in range(reps):
    L.hypre_CSRMatrixMatvec(1.0, diag, x, 0.0, y)
ms = L.hypre_amd_EventTimerStopMs() / reps
byt = nnz * 12 + (nr + 1) * 4 + nr * 8 + nr * 8
print("n=%d rows=%d nnz=%d  %.4f ms/SpMV  %.1f GB/s algorithmic  (%.1f%% of 8 TB/s)" %
      (n, nr, nnz, ms, byt / ms / 1e6, byt / ms / 1e6 / 80.0), flush=True)
import hashlib
print("y digest", hashlib.sha1(B.fetch(y.contents.data, nr, np.float64, B.HYPRE_MEMORY_DEVICE).tobytes()).hexdigest()[:16], flush=True)
# out-of-place residual form
b = B.vec_from_numpy(np.ones(nr))
L.hypre_amd_EventTimerStart()
for _ in range(reps):
    L.hypre_CSRMatrixMatvecOutOfPlace(-1.0, diag, x, 1.0, b, y, 0)
ms = L.hypre_amd_EventTimerStopMs() / reps
print("residual form %.4f ms  %.1f GB/s" % (ms, (byt + nr * 8) / ms / 1e6), flush=True)
B.check()
