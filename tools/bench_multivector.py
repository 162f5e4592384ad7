"""Products with a multivector: one pass over the matrix for up to four columns (spmv_xs_mv_kernel) against one pass
per column, on the fine-level 7-point (or 27-point) operator, value codes on and off.  HIP events on the library's
compute stream.  usage: bench_multivector.py [n] [reps] [--stencil 27] [--json out.json]"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hypre_amd import binding as B

args = [a for a in sys.argv[1:] if not a.startswith("--")]
n = int(args[0]) if len(args) > 0 else 256
reps = int(args[1]) if len(args) > 1 else 30
stencil = 27 if "--stencil" in sys.argv and sys.argv[sys.argv.index("--stencil") + 1] == "27" else 7
out_json = sys.argv[sys.argv.index("--json") + 1] if "--json" in sys.argv else None
L = B.load_library()
L.hypre_SetSyncCudaCompute(0)


def multivector(nr, nv, seed):
    flat = np.random.default_rng(seed).uniform(-1, 1, nr * nv)
    v = B.vec_from_numpy(flat)
    v.contents.size, v.contents.num_vectors, v.contents.vecstride, v.contents.idxstride = nr, nv, nr, 1
    return v


def timed(diag, x, y, beta):
    for _ in range(3):
        L.hypre_CSRMatrixMatvecOutOfPlace(1.0, diag, x, beta, y, y, 0)
    L.hypre_SyncComputeStream()
    B.check()
    L.hypre_amd_EventTimerStart()
    for _ in range(reps):
        L.hypre_CSRMatrixMatvecOutOfPlace(1.0, diag, x, beta, y, y, 0)
    return L.hypre_amd_EventTimerStopMs() / reps


results = []
for codes in (1, 0):
    L.hypre_amd_SpmvSetValueCodes(codes)
    A = B.laplacian(n, n, n, kind="7pt" if stencil == 7 else "27pt")
    L.hypre_ParCSRMatrixMigrate(A, B.HYPRE_MEMORY_DEVICE)
    diag = A.contents.diag
    nr, nnz = diag.contents.num_rows, diag.contents.num_nonzeros
    base = None
    for nv in (1, 2, 3, 4, 8):
        x, y = multivector(nr, nv, nv), multivector(nr, nv, 100 + nv)
        row = {"value_codes": codes, "nv": nv, "rows": nr, "nnz": nnz}
        for fused in (1, 0):
            if nv == 1 and not fused:
                continue
            L.hypre_amd_SpmvSetFusedMultivectors(fused)
            before = L.hypre_amd_SpmvFusedMultivectorLaunches()
            ms = timed(diag, x, y, 0.0)
            row["fused_ms" if fused else "column_loop_ms"] = round(ms, 4)
            if fused:
                row["fused_launches_per_product"] = (L.hypre_amd_SpmvFusedMultivectorLaunches() - before) // (reps + 3)
        if nv == 1:
            base = row["fused_ms"]
            row["single_vector_ms"] = base
        else:
            row["fused_over_single"] = round(row["fused_ms"] / base, 3)
            row["loop_over_single"] = round(row["column_loop_ms"] / base, 3)
            # CSR count of the fused product: matrix once, nv x and y columns
            byt = nnz * 12 + (nr + 1) * 4 + nv * 16 * nr
            row["fused_csr_GBps"] = round(byt / row["fused_ms"] / 1e6, 1)
        results.append(row)
        print(json.dumps(row), flush=True)
        for v in (x, y):
            L.hypre_SeqVectorDestroy(v)
    L.hypre_ParCSRMatrixDestroy(A)
L.hypre_amd_SpmvSetValueCodes(1)
L.hypre_amd_SpmvSetFusedMultivectors(1)
B.check()
if out_json:
    with open(out_json, "w") as f:
        json.dump({"workload": "%d^3 %d-point fine-level operator, Y = A X, nv columns" % (n, stencil), "reps": reps, "rows": results}, f, indent=1)
