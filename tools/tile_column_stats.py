"""How the columns of a 2048-entry tile cluster, per level of the benchmark hierarchy (host-side analysis of sampled tiles):
distinct 2-column units, and for gap thresholds T (in units) the number of segments, of 128-double pieces and the covered
length.  Sizes the x-staging tables of spmv_xs_kernel.

    python tools/tile_column_stats.py [n] [problem]
"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hypre_amd import binding as B, ij   # noqa: E402

L = B.load_library()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
prob = sys.argv[2] if len(sys.argv) > 2 else "laplacian"
opt = ij.IJOptions(n=(n, n, n), coarsen_type=8, interp_type=6, P_max_elmts=4, relax_type=18, problem=prob)
A = ij.build_matrix(opt)
s = ij.create_amg(opt, memory_location=B.HYPRE_MEMORY_HOST)
L.HYPRE_BoomerAMGSetup(s, A, None, None)
B.check()
nl = L.hypre_amd_BoomerAMGGetNumLevels(s)


def stats(name, M):
    m = M.contents
    nnz = m.num_nonzeros
    jj = np.ctypeslib.as_array(m.j, shape=(nnz,))
    T = 2048
    nt = (nnz + T - 1) // T
    rows = []
    for t in np.linspace(0, nt - 1, min(nt, 300)).astype(int):
        u = np.unique(jj[t * T:(t + 1) * T] >> 1)
        gaps = np.diff(u) - 1
        rec = [len(u)]
        for thr in (0, 1, 2, 4, 8, 16, 32, 64):
            cut = gaps > thr
            nseg = int(cut.sum()) + 1
            covered = int(u[-1] - u[0] + 1 - gaps[cut].sum())
            starts = np.concatenate([[0], np.nonzero(cut)[0] + 1])
            ends = np.concatenate([np.nonzero(cut)[0], [len(u) - 1]])
            pos = u - u[0] - np.concatenate([[0], np.cumsum(np.where(cut, gaps, 0))])
            lens = pos[ends] - pos[starts] + 1
            pieces64 = int(np.sum((lens + 63) // 64))
            pieces32 = int(np.sum((lens + 31) // 32))
            rec += [nseg, pieces64, pieces32, covered]
        rows.append(rec)
    r = np.array(rows, dtype=float)
    print("%s  nnz/row %.1f tiles %d  distinct units mean %.0f max %d" % (name, nnz / max(m.num_rows, 1), nt, r[:, 0].mean(), r[:, 0].max()))
    for k, thr in enumerate((0, 1, 2, 4, 8, 16, 32, 64)):
        c = r[:, 1 + 4 * k:5 + 4 * k]
        print("     T=%2d: segs mean %.0f p90 %.0f max %.0f | pieces(128 dbl) mean %.0f p90 %.0f max %.0f | pieces(64 dbl) mean %.0f p90 %.0f | covered units mean %.0f p90 %.0f max %.0f"
              % (thr, c[:, 0].mean(), np.percentile(c[:, 0], 90), c[:, 0].max(), c[:, 1].mean(), np.percentile(c[:, 1], 90), c[:, 1].max(),
                 c[:, 2].mean(), np.percentile(c[:, 2], 90), c[:, 3].mean(), np.percentile(c[:, 3], 90), c[:, 3].max()))


for l in range(min(nl, 4)):
    Al = C.cast(L.hypre_amd_BoomerAMGGetA(s, l), C.POINTER(B.ParCSRMatrix))
    stats("A L%d" % l, Al.contents.diag)
    if l < nl - 1:
        Pl = C.cast(L.hypre_amd_BoomerAMGGetP(s, l), C.POINTER(B.ParCSRMatrix))
        if Pl.contents.diagT:
            stats("P^T L%d" % l, Pl.contents.diagT)
