"""Per-level kernel timing of a BoomerAMG hierarchy (HIP events on the library's stream).

    python tools/bench_levels.py [n] [reps] [--variants 0,1:4,1:6] [--problem laplacian|27pt] [--levels 4]

Builds the benchmark hierarchy (n^3 7-pt Laplacian, PMIS / ext+i(4) / l1-Jacobi) once and times, level by level and
for every kernel variant asked for (hypre_amd_SpmvSetVariant: `v` or `v:workgroups-per-CU`), y = A x, the fused
l1-Jacobi sweep, the prolongation P x and the restriction P^T x, printing algorithmic GB/s (SURVEY.md 8d byte formulas).
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
ap.add_argument("reps", type=int, nargs="?", default=20)
ap.add_argument("--variants", default="0")
ap.add_argument("--problem", default="laplacian")
ap.add_argument("--levels", type=int, default=99, help="time the first LEVELS levels only")
ap.add_argument("--relax", type=int, default=18)
ap.add_argument("--mixed", action="store_true")
ap.add_argument("--json", default="", help="write the table (per level and operation: ms, SURVEY 8(d) bytes, GB/s, fraction of 8 TB/s) here")
args = ap.parse_args()
table = {"problem": args.problem, "n": args.n, "relax": args.relax, "mixed": bool(args.mixed), "reps": args.reps,
         "bytes": "SURVEY.md 8(d): CSR entries at 12 bytes, row pointers, every vector operand once", "peak_GBps": 8000.0, "levels": []}
n, reps = args.n, args.reps
L = B.load_library()
opt = ij.IJOptions(n=(n, n, n), coarsen_type=8, interp_type=6, P_max_elmts=4, relax_type=args.relax, num_sweeps=1,
                   problem=args.problem)
t0 = time.time()
A = ij.build_matrix(opt)
s = ij.create_amg(opt, memory_location=B.HYPRE_MEMORY_DEVICE)
if args.mixed:
    L.hypre_amd_BoomerAMGSetMixedPrecision(s, 1)
L.hypre_ParCSRMatrixMigrate(A, B.HYPRE_MEMORY_DEVICE)
L.HYPRE_BoomerAMGSetup(s, A, None, None)
B.check()
L.hypre_ParCSRMatrixMigrate(A, B.HYPRE_MEMORY_DEVICE)     # level 0 is the caller's matrix
print("setup %.1fs" % (time.time() - t0), flush=True)
L.hypre_SetSyncCudaCompute(0)
nl = L.hypre_amd_BoomerAMGGetNumLevels(s)


def timed(fn):
    for _ in range(3):
        fn()
    L.hypre_SyncComputeStream()
    L.hypre_amd_EventTimerStart()
    for _ in range(reps):
        fn()
    return L.hypre_amd_EventTimerStopMs() / reps


variants = []
for tok in args.variants.split(","):
    v = tok.split(":")
    variants.append((int(v[0]), int(v[1]) if len(v) > 1 else 0))

# one V(1,1) cycle per variant first (what the benchmark times)
b = B.parvec_from_numpy(np.ones(n ** 3))
u = B.parvec_from_numpy(np.zeros(n ** 3))
L.HYPRE_BoomerAMGSetTol(s, 0.0)
L.HYPRE_BoomerAMGSetMaxIter(s, 1)


def cycle():
    L.hypre_ParVectorSetZeros(u)
    L.HYPRE_BoomerAMGSolve(s, A, b, u)


for var, wgs in variants:
    L.hypre_amd_SpmvSetVariant(var, wgs)
    ms_c = timed(cycle)
    print("variant %d:%d  V-cycle %.4f ms" % (var, wgs, ms_c), flush=True)
    table.setdefault("vcycle_ms", {})["%d:%d" % (var, wgs)] = ms_c

for l in range(min(nl, args.levels)):
    Al = C.cast(L.hypre_amd_BoomerAMGGetA(s, l), C.POINTER(B.ParCSRMatrix))
    d = Al.contents.diag.contents
    nr, nnz = d.num_rows, d.num_nonzeros
    x = B.parvec_from_numpy(np.random.default_rng(l).uniform(-1, 1, nr))
    y = B.parvec_from_numpy(np.zeros(nr))
    f = B.parvec_from_numpy(np.ones(nr))
    v = B.parvec_from_numpy(np.zeros(nr))
    l1p = L.hypre_amd_BoomerAMGGetL1Norms(s, l)
    l1 = C.cast(l1p, C.POINTER(B.Vector)).contents.data if l1p else None
    nt, mp = C.c_int(), C.c_double()
    st = L.hypre_amd_CSRMatrixPlanStaging(Al.contents.diag, C.byref(nt), C.byref(mp))
    print("L%d rows=%9d nnz=%10d (%.1f/row)  tiles %d, x-staged %d (%.1f pieces each)" % (l, nr, nnz, nnz / max(nr, 1), nt.value, st, mp.value), flush=True)
    if l < nl - 1:
        Pq = C.cast(L.hypre_amd_BoomerAMGGetP(s, l), C.POINTER(B.ParCSRMatrix))
        stp = L.hypre_amd_CSRMatrixPlanStaging(Pq.contents.diag, C.byref(nt), C.byref(mp))
        line = "   P: tiles %d, x-staged %d (%.1f pieces)" % (nt.value, stp, mp.value)
        if Pq.contents.diagT:
            stt = L.hypre_amd_CSRMatrixPlanStaging(Pq.contents.diagT, C.byref(nt), C.byref(mp))
            line += " | P^T: tiles %d, x-staged %d (%.1f pieces)" % (nt.value, stt, mp.value)
        print(line, flush=True)
    rs_rows, rs_kp = C.c_int(), C.c_int()
    rs_w = L.hypre_amd_CSRMatrixPlanRowSlices(Al.contents.diag, C.byref(rs_rows), C.byref(rs_kp))
    form = L.hypre_amd_CSRMatrixPlanForm(Al.contents.diag)
    print("   A: plan form %d%s" % (form, ("  (row slices: %d lanes a row, %d rows a block, %d entries a lane at most)" % (rs_w, rs_rows.value, rs_kp.value)) if rs_w else ""), flush=True)
    lev = {"level": l, "rows": nr, "nnz": nnz, "tiles": nt.value, "x_staged_tiles": st, "plan_form": form, "row_slice_lanes": rs_w,
           "row_slice_rows": rs_rows.value, "row_slice_entries_per_lane": rs_kp.value, "ops": {}}
    table["levels"].append(lev)

    def rec(name, var, wgs, ms, by):
        lev["ops"].setdefault(name, {})["%d:%d" % (var, wgs)] = {"ms": ms, "bytes": by, "GBps": by / ms / 1e6, "frac_of_peak": by / ms / 1e6 / 8000.0}

    z = B.parvec_from_numpy(np.zeros(nr))
    for var, wgs in variants:
        L.hypre_amd_SpmvSetVariant(var, wgs)
        ms_a = timed(lambda: L.hypre_ParCSRMatrixMatvec(1.0, Al, x, 0.0, y))
        by_a = nnz * 12 + (nr + 1) * 4 + nr * 16
        rec("A x", var, wgs, ms_a, by_a)
        line = "   v%d:%d  A x: %.4f ms %6.0f GB/s" % (var, wgs, ms_a, by_a / ms_a / 1e6)
        if l1 is not None and l < nl - 1 and args.relax in (7, 18):
            ms_j = timed(lambda: L.hypre_BoomerAMGRelax(Al, f, None, args.relax, 0, 1.0, 1.0, l1, x, v, v))
            by_j = nnz * 12 + (nr + 1) * 4 + nr * 32
            rec("l1-Jacobi sweep (public entry: + copy back)", var, wgs, ms_j, by_j + nr * 16)
            line += " | l1-Jacobi (+copy back): %.4f ms %6.0f GB/s" % (ms_j, (by_j + nr * 16) / ms_j / 1e6)
        if l1 is not None and l < nl - 1 and args.relax in (11, 12):
            # two-stage GS: residual pass over A, then (relax - 10) passes over the strict lower triangle (about half the
            # entries) with their vector traffic (par_relax_device.c:97-155)
            inner = args.relax - 10
            ms_j = timed(lambda: L.hypre_BoomerAMGRelax(Al, f, None, args.relax, 0, 1.0, 1.0, l1, x, v, z))
            nl_low = (nnz - nr) // 2
            by_j = (nnz * 12 + (nr + 1) * 4 + nr * 24) + nr * 40 + inner * (nl_low * 12 + (nr + 1) * 4 + nr * 40)
            rec("two-stage GS sweep (%d inner)" % inner, var, wgs, ms_j, by_j)
            line += " | two-stage GS(%d): %.4f ms %6.0f GB/s" % (inner, ms_j, by_j / ms_j / 1e6)
        if l < nl - 1:
            Pl = C.cast(L.hypre_amd_BoomerAMGGetP(s, l), C.POINTER(B.ParCSRMatrix))
            pd = Pl.contents.diag.contents
            nc = pd.num_cols
            xc = B.parvec_from_numpy(np.random.default_rng(l + 50).uniform(-1, 1, nc))
            yc = B.parvec_from_numpy(np.zeros(nc))
            ms_p = timed(lambda: L.hypre_ParCSRMatrixMatvec(1.0, Pl, xc, 1.0, y))
            ms_r = timed(lambda: L.hypre_ParCSRMatrixMatvecT(1.0, Pl, x, 0.0, yc))
            by_p = pd.num_nonzeros * 12 + (nr + 1) * 4 + nc * 8 + nr * 16
            by_r = pd.num_nonzeros * 12 + (nc + 1) * 4 + nr * 8 + nc * 8
            rec("P x (u += P e)", var, wgs, ms_p, by_p)
            rec("P^T x", var, wgs, ms_r, by_r)
            line += " | P x: %.4f ms %6.0f GB/s | P^T x: %.4f ms %6.0f GB/s" % (ms_p, by_p / ms_p / 1e6, ms_r, by_r / ms_r / 1e6)
        print(line, flush=True)
B.check()
if args.json:
    import json
    with open(args.json, "w") as fh:
        json.dump(table, fh, indent=1)
