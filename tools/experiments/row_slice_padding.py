"""Host-side count (round 4): how much a sliced-ELL layout of the coarse operators would pad — W lanes per row, slices of 64 lane-rows,
rows in matrix order or sorted by length inside a 256 / W-row block, chunk = 1, 2 or 4 entries — and what the jagged layout that was
built instead (no padding) is compared with.  python tools/experiments/row_slice_padding.py [n]  ("Row slices")."""
import ctypes as C, os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hypre_amd import binding as B, ij
L = B.load_library()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 96
prob = sys.argv[2] if len(sys.argv) > 2 else "laplacian"
opt = ij.IJOptions(n=(n, n, n), coarsen_type=8, interp_type=6, P_max_elmts=4, relax_type=18, problem=prob)
A = ij.build_matrix(opt)
s = ij.create_amg(opt, memory_location=B.HYPRE_MEMORY_HOST)
L.HYPRE_BoomerAMGSetup(s, A, None, None)
B.check()
nl = L.hypre_amd_BoomerAMGGetNumLevels(s)
def pad(lens, W, chunk, sort_block):
    # rows per block R=256/W, rows per wave 64/W, lane entries ceil((len-sub)/W) -> max = ceil(len/W)
    R = 256 // W; rw = 64 // W
    nb = -(-len(lens) // R)
    l = np.zeros(nb * R, dtype=np.int64); l[:len(lens)] = lens
    l = l.reshape(nb, R)
    if sort_block: l = -np.sort(-l, axis=1)
    per_lane = -(-l // W)
    wmax = per_lane.reshape(nb, 4, rw).max(axis=2)
    wch = -(-wmax // chunk) * chunk
    return (wch.sum() * 64) / lens.sum(), int(per_lane.max())
for l in range(min(nl, 4)):
    Al = C.cast(L.hypre_amd_BoomerAMGGetA(s, l), C.POINTER(B.ParCSRMatrix)).contents.diag.contents
    ii = np.ctypeslib.as_array(Al.i, shape=(Al.num_rows + 1,))
    lens = np.diff(ii)
    print("A L%d rows %d nnz/row %.1f min %d p10 %d p50 %d p90 %d max %d" % (l, Al.num_rows, lens.mean(), lens.min(), *np.percentile(lens, [10, 50, 90]).astype(int), lens.max()))
    for W in (1, 2, 4, 8, 16):
        for chunk in (2, 4):
            a, m = pad(lens, W, chunk, False); b, _ = pad(lens, W, chunk, True)
            print("    W=%2d chunk=%d: padding unsorted %.3f sorted-in-block %.3f  max entries/lane %d  entries/block %.0f" % (W, chunk, a, b, m, lens.mean() * 256 / W))
    if l < nl - 1:
        Pl = C.cast(L.hypre_amd_BoomerAMGGetP(s, l), C.POINTER(B.ParCSRMatrix)).contents
        for nm, M in (("P", Pl.diag), ("PT", Pl.diagT)):
            if not M: continue
            M = M.contents
            ii = np.ctypeslib.as_array(M.i, shape=(M.num_rows + 1,)); lens = np.diff(ii)
            print("%s L%d rows %d nnz/row %.1f p10 %d p50 %d p90 %d max %d" % (nm, l, M.num_rows, lens.mean(), *np.percentile(lens, [10, 50, 90]).astype(int), lens.max()))
            for W in (1, 2, 4):
                a, m = pad(lens, W, 2, False); b, _ = pad(lens, W, 2, True)
                print("    W=%2d chunk=2: padding unsorted %.3f sorted %.3f max/lane %d" % (W, a, b, m))
print("---- chunk=1 and variable-lanes packing")
def pack_T(lens, T):
    # rows packed into waves of 64 lanes: ceil(len/T) lanes per row, no straddling; returns padded slots / nnz
    lanes = -(-lens // T)
    lanes = np.maximum(lanes, 1)
    waves = 0; used = 0
    for w in lanes:
        if used + w > 64: waves += 1; used = 0
        used += w
    waves += 1
    return waves * 64 * T / lens.sum()
for l in range(1, min(nl, 4)):
    Al = C.cast(L.hypre_amd_BoomerAMGGetA(s, l), C.POINTER(B.ParCSRMatrix)).contents.diag.contents
    ii = np.ctypeslib.as_array(Al.i, shape=(Al.num_rows + 1,)); lens = np.diff(ii)
    for W in (1, 2, 4, 8):
        print("A L%d W=%d chunk=1 sorted %.3f unsorted %.3f" % (l, W, pad(lens, W, 1, True)[0], pad(lens, W, 1, False)[0]))
    for T in (2, 4, 6, 8, 12, 16):
        print("A L%d T=%d variable lanes: %.3f" % (l, T, pack_T(lens, T)))
