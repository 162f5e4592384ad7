"""Host-side simulation (round 4): what a blocked numbering of the coarse unknowns would buy the x staging.  Builds the n^3
hierarchy on the host, renumbers levels 1 and 2 tile-major (tiles of the fine grid) or in Morton order, and reports per block
of R rows the distinct 2-column units, the staged volume and how many blocks fit the staging limits (48 pieces, 2048 units) —
for A_l and for the restriction operators.  python tools/experiments/renumbering_sim.py [n]  (DESIGN.md section 7)."""
import ctypes as C, os, sys, numpy as np, scipy.sparse as sp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hypre_amd import binding as B, ij
L = B.load_library()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 96
opt = ij.IJOptions(n=(n, n, n), coarsen_type=8, interp_type=6, P_max_elmts=4, relax_type=18)
A = ij.build_matrix(opt)
s = ij.create_amg(opt, memory_location=B.HYPRE_MEMORY_HOST)
L.HYPRE_BoomerAMGSetup(s, A, None, None)
B.check()
def csr(M):
    m = M.contents
    return sp.csr_matrix((np.ctypeslib.as_array(m.data, shape=(m.num_nonzeros,)).copy(), np.ctypeslib.as_array(m.j, shape=(m.num_nonzeros,)).copy(),
                          np.ctypeslib.as_array(m.i, shape=(m.num_rows + 1,)).copy()), shape=(m.num_rows, m.num_cols))
def covered(cols):
    u = np.unique(cols >> 1)
    gaps = np.diff(u) - 1
    for thr in (0, 1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 1 << 30):
        cut = gaps > thr
        starts = np.concatenate([[0], np.nonzero(cut)[0] + 1]); ends = np.concatenate([np.nonzero(cut)[0], [len(u) - 1]])
        pos = u - u[0] - np.concatenate([[0], np.cumsum(np.where(cut, gaps, 0))])
        lens = pos[ends] - pos[starts] + 1
        pieces = int(np.sum((lens + 63) // 64))
        cov = int(u[-1] - u[0] + 1 - gaps[cut].sum())
        if cov > 2048: return len(u), None, pieces
        if pieces <= 48: return len(u), cov, pieces
    return len(u), None, 0
def stats(name, M, R):
    ii, jj = M.indptr, M.indices
    nb = -(-M.shape[0] // R)
    rec = []
    for b in np.linspace(0, nb - 1, min(nb, 300)).astype(int):
        lo, hi = ii[b * R], ii[min((b + 1) * R, M.shape[0])]
        if hi > lo: rec.append(covered(jj[lo:hi]) + (hi - lo,))
    du = np.array([r[0] for r in rec]); cov = np.array([r[1] if r[1] is not None else -1 for r in rec]); ent = np.array([r[3] for r in rec]); pc = np.array([r[2] for r in rec])
    ok = cov >= 0
    print("%-28s R=%3d entries %.0f distinct units %.0f | staged %.0f%%: units p50 %d p99 %d pieces %.0f | distinct doubles/entry %.3f staged doubles/entry %.3f" % (
        name, R, ent.mean(), du.mean(), 100 * ok.mean(), *(np.percentile(cov[ok], [50, 99]).astype(int) if ok.any() else (0, 0)), pc[ok].mean() if ok.any() else 0,
        2 * du.mean() / ent.mean(), (2 * cov[ok].mean() / ent[ok].mean()) if ok.any() else 0))
# coordinates of level-l points in the fine grid, via the CF markers
def cf(l):
    p = L.hypre_amd_BoomerAMGGetCFMarker(s, l)
    ia = C.cast(p, C.POINTER(B.IntArray)).contents
    return np.ctypeslib.as_array(ia.data, shape=(ia.size,)).copy()
fine_of = [np.arange(n ** 3)]
for l in range(3):
    m = cf(l)
    fine_of.append(fine_of[l][m == 1])
def tile_perm(l, t):
    p = fine_of[l]
    x, y, z = p % n, (p // n) % n, p // (n * n)
    key = (((z // t) * (n // t + 1) + (y // t)) * (n // t + 1) + (x // t))
    order = np.lexsort((x, y, z, key))          # tile-major, lexicographic inside the tile
    perm = np.empty(len(p), dtype=np.int64); perm[order] = np.arange(len(p))     # new index of old point
    return order, perm
def morton_perm(l):
    p = fine_of[l]
    x, y, z = (p % n).astype(np.uint64), ((p // n) % n).astype(np.uint64), (p // (n * n)).astype(np.uint64)
    def spread(v):
        r = np.zeros_like(v)
        for b in range(10): r |= ((v >> np.uint64(b)) & np.uint64(1)) << np.uint64(3 * b)
        return r
    key = spread(x) | (spread(y) << np.uint64(1)) | (spread(z) << np.uint64(2))
    order = np.argsort(key, kind="stable")
    perm = np.empty(len(p), dtype=np.int64); perm[order] = np.arange(len(p))
    return order, perm
for l in (1, 2):
    Al = csr(C.cast(L.hypre_amd_BoomerAMGGetA(s, l), C.POINTER(B.ParCSRMatrix)).contents.diag)
    Pm = C.cast(L.hypre_amd_BoomerAMGGetP(s, l - 1), C.POINTER(B.ParCSRMatrix)).contents
    PT = csr(Pm.diagT)
    for R in (64, 128):
        stats("A L%d lexicographic" % l, Al, R)
    stats("PT L%d lexicographic" % (l - 1), PT, 64)
    for t in (8, 16, 32):
        order, perm = tile_perm(l, t)
        Ap = Al[order][:, :]
        Ap = sp.csr_matrix((Ap.data, perm[Ap.indices].astype(np.int32), Ap.indptr), shape=Ap.shape)
        for R in (64, 128):
            stats("A L%d tiles %d^3 (fine)" % (l, t), Ap, R)
        # restriction: rows = coarse points (permuted), columns = fine points of level l-1 (permuted too if l-1 >= 1)
        PTp = PT[order]
        if l - 1 >= 1:
            o2, p2 = tile_perm(l - 1, t)
            PTp = sp.csr_matrix((PTp.data, p2[PTp.indices].astype(np.int32), PTp.indptr), shape=PTp.shape)
        stats("PT L%d tiles %d^3" % (l - 1, t), PTp, 64)
    order, perm = morton_perm(l)
    Ap = Al[order]
    Ap = sp.csr_matrix((Ap.data, perm[Ap.indices].astype(np.int32), Ap.indptr), shape=Ap.shape)
    for R in (64, 128):
        stats("A L%d morton" % l, Ap, R)
