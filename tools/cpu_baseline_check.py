"""the repo's CPU restatement (oracle/oracle.c, OpenMP row loops) timed in THIS container
against the reference's own CPU numbers recorded by the survey in the same container (8 MPI ranks
on 8 cores, 256^3 7-pt: 37.8 ms per SpMV, 0.44 s per AMG-PCG iteration, 23 iterations to 5.77e-09).

    python tools/cpu_baseline_check.py [n] [threads]

The hierarchy is the library's host setup on one rank (the reference ran 8 ranks; the C/F splitting and hence the
iteration count differ slightly with the rank count, as they do in the reference).  Writes profiles/<tag>.json.
"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
from hypre_amd import binding as B, ij   # noqa: E402
import pyoracle as O                      # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
threads = int(sys.argv[2]) if len(sys.argv) > 2 else len(os.sched_getaffinity(0))
L = B.load_library()
opt = ij.IJOptions(n=(n, n, n), coarsen_type=8, interp_type=6, P_max_elmts=4, relax_type=18, num_sweeps=1, solver=1)
t0 = time.time()
A = ij.build_matrix(opt)
s = ij.create_amg(opt, memory_location=B.HYPRE_MEMORY_HOST)
L.HYPRE_BoomerAMGSetup(s, A, None, None)
B.check()
setup_s = time.time() - t0
amg = O.amg_from_solvers([s])
A0 = amg.A_levels[0]
N = A0.nrows
nnz = int(A.contents.diag.contents.num_nonzeros)
O.set_num_threads(threads)
x = np.random.default_rng(0).uniform(-1, 1, N)
y = np.zeros(N)
O.par_matvec(1.0, A0, x, 0.0, y, y)
reps = 100
t0 = time.perf_counter()
for _ in range(reps):
    O.par_matvec(1.0, A0, x, 0.0, y, y)
spmv_ms = 1e3 * (time.perf_counter() - t0) / reps
by = nnz * 12 + (N + 1) * 4 + N * 16
f = np.ones(N)
u = np.zeros(N)
amg.cycle(f, u, u_all_zeros=True)
cyc = []
for _ in range(5):
    u[:] = 0.0
    t0 = time.perf_counter()
    amg.cycle(f, u, u_all_zeros=True)
    cyc.append(time.perf_counter() - t0)
xs = np.zeros(N)
t0 = time.perf_counter()
its, rel, _ = amg.pcg(f, xs, tol=1e-8, max_iter=100, two_norm=1)
pcg_s = time.perf_counter() - t0
out = {"container": "build container (8 vCPU Xeon 2.1 GHz, the survey's host)", "threads": threads, "n": n,
       "host_setup_seconds": setup_s,
       "port": {"spmv_ms": spmv_ms, "spmv_GBps": by / spmv_ms / 1e6, "vcycle_ms_median": 1e3 * float(np.median(cyc)),
                "vcycle_MDOFps": N / float(np.median(cyc)) / 1e6, "pcg_iterations": its, "pcg_final_rel_resid": rel,
                "pcg_s_per_iteration": pcg_s / max(its, 1), "pcg_MDOFps_per_iteration": N * its / pcg_s / 1e6},
       "reference_cpu_survey_container": {"spmv_ms": 37.8, "spmv_GBps": 46.0, "pcg_iterations": 23,
                                          "pcg_final_rel_resid": 5.768447e-09, "pcg_s_per_iteration": 0.44,
                                          "pcg_MDOFps_per_iteration": 38.0,
                                          "source": "BASELINE.md section 2 (ij -n 256 256 256 -P 2 2 2, 8 MPI ranks, no OpenMP)"}}
out["port_over_reference"] = {"spmv_time": spmv_ms / 37.8, "pcg_time_per_iteration": out["port"]["pcg_s_per_iteration"] / 0.44}
print(json.dumps(out, indent=1))
if n == 256:
    with open(os.path.join(ROOT, "profiles", "r02_cpu_port_vs_reference_survey_container.json"), "w") as fh:
        json.dump(out, fh, indent=1)
