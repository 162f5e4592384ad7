"""Does the rate of a repeated launch follow a clock?  Level-1 SpMV repeated in rounds of 30 while a thread samples the
device's clock files (sysfs pp_dpm_*), then the same right after 20 V-cycles."""
import ctypes as C
import glob
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hypre_amd import binding as B, ij   # noqa: E402


def active(path):
    try:
        for line in open(path):
            if "*" in line:
                return line.strip()
    except OSError:
        return None
    return None


files = sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_*clk"))
print("clock files:", files, flush=True)
samples, stop = [], False


def watch():
    while not stop:
        samples.append((time.perf_counter(), tuple(active(f) for f in files)))
        time.sleep(0.004)


n = 256
L = B.load_library()
opt = ij.IJOptions(n=(n, n, n), coarsen_type=8, interp_type=6, P_max_elmts=4, relax_type=18, num_sweeps=1)
A = ij.build_matrix(opt)
L.hypre_ParCSRMatrixMigrate(A, B.HYPRE_MEMORY_DEVICE)
s = ij.create_amg(opt, memory_location=B.HYPRE_MEMORY_DEVICE)
L.HYPRE_BoomerAMGSetup(s, A, None, None)
B.check()
L.hypre_SetSyncCudaCompute(0)
Al = C.cast(L.hypre_amd_BoomerAMGGetA(s, 1), C.POINTER(B.ParCSRMatrix))
d = Al.contents.diag
nr = d.contents.num_rows
x = B.vec_from_numpy(np.random.default_rng(1).uniform(-1, 1, nr))
y = B.vec_from_numpy(np.zeros(nr))
nn = n ** 3
b = B.parvec_from_numpy(np.ones(nn))
u = B.parvec_from_numpy(np.zeros(nn))
L.HYPRE_BoomerAMGSetTol(s, 0.0)
L.HYPRE_BoomerAMGSetMaxIter(s, 1)


def rounds(k, reps=30):
    out = []
    for _ in range(k):
        L.hypre_SyncComputeStream()
        L.hypre_amd_EventTimerStart()
        for _ in range(reps):
            L.hypre_CSRMatrixMatvec(1.0, d, x, 0.0, y)
        out.append(L.hypre_amd_EventTimerStopMs() / reps)
    return out


th = threading.Thread(target=watch)
th.start()
time.sleep(0.05)
t0 = time.perf_counter()
print("level-1 SpMV, rounds of 30:", " ".join("%.4f" % t for t in rounds(6)), flush=True)
t1 = time.perf_counter()
for _ in range(40):
    L.hypre_ParVectorSetZeros(u)
    L.HYPRE_BoomerAMGSolve(s, A, b, u)
L.hypre_SyncComputeStream()
t2 = time.perf_counter()
print("after 40 V-cycles:", " ".join("%.4f" % t for t in rounds(4)), flush=True)
t3 = time.perf_counter()
stop = True
th.join()
seen = []
for t, v in samples:
    if not seen or seen[-1][1] != v:
        seen.append((t - t0, v))
print("clock states over time (s since the first round):")
for t, v in seen[:40]:
    print("  %.3f %s" % (t, v))
print("phases: spmv rounds 0 .. %.3f, cycles .. %.3f, spmv rounds .. %.3f" % (t1 - t0, t2 - t0, t3 - t0))
