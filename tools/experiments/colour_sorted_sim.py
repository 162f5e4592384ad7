"""Host-side simulation (round 4): can a block of rows of ONE colour stage its x, in the level's own numbering or with the
columns renumbered colour by colour?  (No: rows of a colour are never neighbours and share no columns.)  DESIGN.md section 7."""
import ctypes as C, os, sys, numpy as np, scipy.sparse as sp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'renumbering_sim.py')).read().split("# coordinates of level-l points")[0])
def greedy(A):
    S = (A + A.T).tocsr(); n = S.shape[0]
    col = -np.ones(n, dtype=np.int64)
    ip, ix = S.indptr, S.indices
    for i in range(n):
        nb = col[ix[ip[i]:ip[i+1]]]
        used = set(nb[nb >= 0].tolist())
        c = 0
        while c in used: c += 1
        col[i] = c
    return col
for l in (1,):
    Al = csr(C.cast(L.hypre_amd_BoomerAMGGetA(s, l), C.POINTER(B.ParCSRMatrix)).contents.diag)
    col = greedy(Al)
    Cn = col.max() + 1
    order = np.lexsort((np.arange(len(col)), col))
    inv = np.empty(len(col), dtype=np.int64); inv[order] = np.arange(len(col))
    cnt = np.bincount(col)
    print("colours", Cn, cnt[:25])
    c = 3
    rows = order[np.cumsum(cnt)[c-1] if c else 0: np.cumsum(cnt)[c]]
    M_orig = Al[rows]
    M_sorted = sp.csr_matrix((M_orig.data, inv[M_orig.indices].astype(np.int32), M_orig.indptr), shape=M_orig.shape)
    for R in (64, 128):
        stats("colour %d, original columns" % c, M_orig, R)
        stats("colour %d, sorted columns" % c, M_sorted, R)
