"""ctypes view of the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg, never by the hypre_amd package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "liboracle.so")

IntP = C.POINTER(C.c_int)
RealP = C.POINTER(C.c_double)
LLP = C.POINTER(C.c_longlong)


class OCSR(C.Structure):
    _fields_ = [("nrows", C.c_int), ("ncols", C.c_int), ("i", IntP), ("j", IntP), ("a", RealP),
                ("rownnz", IntP), ("num_rownnz", C.c_int)]


class OPAR(C.Structure):
    _fields_ = [("nranks", C.c_int), ("diag", C.POINTER(OCSR)), ("offd", C.POINTER(OCSR)),
                ("col_map_offd", C.POINTER(LLP)), ("row_starts", LLP), ("col_starts", LLP)]


class OAMG(C.Structure):
    _fields_ = [("num_levels", C.c_int), ("max_levels", C.c_int), ("A", C.POINTER(OPAR)), ("P", C.POINTER(OPAR)),
                ("cf_marker", C.POINTER(IntP)), ("l1_norms", C.POINTER(RealP)), ("F", C.POINTER(RealP)),
                ("U", C.POINTER(RealP)), ("vtemp", RealP), ("num_grid_sweeps", C.c_int * 4),
                ("grid_relax_type", C.c_int * 4), ("grid_relax_points", C.POINTER(IntP)),
                ("relax_order", C.c_int), ("user_relax_type", C.c_int), ("relax_weight", RealP),
                ("omega", RealP), ("cycle_type", C.c_int), ("fcycle", C.c_int), ("num_threads", C.c_int),
                ("cheby_order", C.c_int), ("cheby_scale", C.c_int), ("cheby_coefs", C.POINTER(RealP)),
                ("cheby_ds", C.POINTER(RealP)), ("A_outer", C.POINTER(OPAR)), ("colors", C.POINTER(IntP))]


def build():
    subprocess.run(["make", "-C", HERE, "-s"], check=True)
    return LIB


_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(os.path.join(HERE, "oracle.c")):
        build()
    L = C.CDLL(LIB)
    P = C.POINTER
    L.oracle_set_num_threads.restype = None
    L.oracle_set_num_threads.argtypes = [C.c_int]
    L.oracle_get_num_threads.restype = C.c_int
    L.oracle_drop_transposes.restype = None
    L.oracle_csr_matvec.restype = C.c_int
    L.oracle_csr_matvec.argtypes = [C.c_double, P(OCSR), RealP, C.c_int, C.c_double, RealP, C.c_int, RealP,
                                    C.c_int, C.c_int]
    L.oracle_csr_matvecT.restype = C.c_int
    L.oracle_csr_matvecT.argtypes = [C.c_double, P(OCSR), RealP, C.c_int, C.c_double, RealP, C.c_int]
    L.oracle_inner_prod.restype = C.c_double
    L.oracle_inner_prod.argtypes = [RealP, RealP, C.c_longlong]
    L.oracle_axpy.restype = None
    L.oracle_axpy.argtypes = [C.c_double, RealP, RealP, C.c_longlong]
    L.oracle_par_matvec.restype = C.c_int
    L.oracle_par_matvec.argtypes = [C.c_double, P(OPAR), RealP, C.c_double, RealP, RealP]
    L.oracle_par_matvecT.restype = C.c_int
    L.oracle_par_matvecT.argtypes = [C.c_double, P(OPAR), RealP, C.c_double, RealP]
    L.oracle_l1_norms.restype = C.c_int
    L.oracle_l1_norms.argtypes = [P(OPAR), C.c_int, IntP, RealP]
    L.oracle_relax.restype = C.c_int
    L.oracle_relax.argtypes = [P(OPAR), RealP, IntP, C.c_int, C.c_int, C.c_double, C.c_double, RealP, RealP,
                               RealP, C.c_int, IntP]
    L.oracle_set_multicolor.restype = None
    L.oracle_set_multicolor.argtypes = [IntP]
    L.oracle_relax_if.restype = C.c_int
    L.oracle_relax_if.argtypes = [P(OPAR), RealP, IntP, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double,
                                  RealP, RealP, RealP, C.c_int, IntP]
    L.oracle_gselim.restype = C.c_int
    L.oracle_gselim.argtypes = [RealP, RealP, C.c_int]
    L.oracle_amg_cycle.restype = C.c_int
    L.oracle_amg_cycle.argtypes = [P(OAMG), P(RealP), P(RealP), IntP]
    L.oracle_amg_solve.restype = C.c_int
    L.oracle_amg_solve.argtypes = [P(OAMG), RealP, RealP, C.c_double, C.c_int, C.c_int, C.c_int, C.c_int,
                                   RealP, IntP, RealP]
    L.oracle_pcg_amg.restype = C.c_int
    L.oracle_pcg_amg.argtypes = [P(OAMG), RealP, RealP, C.c_double, C.c_double, C.c_int, C.c_int, C.c_int,
                                 RealP, IntP]
    L.oracle_pcg_amg_flex.restype = C.c_int
    L.oracle_pcg_amg_flex.argtypes = [P(OAMG), RealP, RealP, C.c_double, C.c_double, C.c_int, C.c_int, C.c_int, C.c_int,
                                      RealP, IntP]
    L.oracle_gmres_amg.restype = C.c_int
    L.oracle_gmres_amg.argtypes = [P(OAMG), RealP, RealP, C.c_double, C.c_double, C.c_int, C.c_int, C.c_int,
                                   RealP, IntP]
    L.oracle_gmres_ds_multi.restype = C.c_int
    L.oracle_gmres_ds_multi.argtypes = [P(OPAR), RealP, RealP, C.c_int, C.c_double, C.c_double, C.c_int, C.c_int, RealP, IntP]
    L.oracle_pcg_ds_multi.restype = C.c_int
    L.oracle_pcg_ds_multi.argtypes = [P(OPAR), RealP, RealP, C.c_int, C.c_double, C.c_double, C.c_int, C.c_int, RealP, IntP]
    _lib = L
    return L


def set_num_threads(n):
    """OpenMP threads of the independent row loops (bit-identical results for any count)."""
    load().oracle_set_num_threads(int(n))


def drop_transposes():
    load().oracle_drop_transposes()


def _ip(a):
    return a.ctypes.data_as(IntP)


def _rp(a):
    return a.ctypes.data_as(RealP) if a is not None else None


class Csr:
    """Keeps the numpy arrays alive next to the C struct."""

    def __init__(self, nrows, ncols, indptr, indices, data, rownnz=None):
        self.indptr = np.ascontiguousarray(indptr, dtype=np.int32)
        self.indices = np.ascontiguousarray(indices, dtype=np.int32)
        self.data = np.ascontiguousarray(data, dtype=np.float64)
        self.rownnz = None if rownnz is None else np.ascontiguousarray(rownnz, dtype=np.int32)
        self.c = OCSR(int(nrows), int(ncols), _ip(self.indptr), _ip(self.indices), _rp(self.data),
                      _ip(self.rownnz) if self.rownnz is not None else None,
                      int(len(self.rownnz)) if self.rownnz is not None else int(nrows))

    @classmethod
    def from_scipy(cls, A, with_rownnz=False):
        A = A.tocsr()
        rn = None
        if with_rownnz:
            rn = np.nonzero(np.diff(A.indptr))[0].astype(np.int32)
            if len(rn) == A.shape[0] or len(rn) == 0:
                rn = None
        return cls(A.shape[0], A.shape[1], A.indptr, A.indices, A.data, rn)


class Par:
    """Distributed matrix as virtual ranks.  blocks = list of
    (diag Csr, offd Csr, col_map_offd int64 array); row_starts/col_starts global."""

    def __init__(self, blocks, row_starts, col_starts):
        self.blocks = blocks
        n = len(blocks)
        self.row_starts = np.ascontiguousarray(row_starts, dtype=np.int64)
        self.col_starts = np.ascontiguousarray(col_starts, dtype=np.int64)
        self._diag = (OCSR * n)(*[b[0].c for b in blocks])
        self._offd = (OCSR * n)(*[b[1].c for b in blocks])
        self._cmaps = [np.ascontiguousarray(b[2], dtype=np.int64) for b in blocks]
        self._cmap_ptrs = (LLP * n)(*[m.ctypes.data_as(LLP) for m in self._cmaps])
        self.c = OPAR(n, self._diag, self._offd, self._cmap_ptrs, self.row_starts.ctypes.data_as(LLP),
                      self.col_starts.ctypes.data_as(LLP))

    @property
    def nrows(self):
        return int(self.row_starts[-1])

    @property
    def ncols(self):
        return int(self.col_starts[-1])

    @classmethod
    def from_scipy_single(cls, A):
        """One virtual rank holding the whole matrix (no ghost block)."""
        import scipy.sparse as sp
        A = A.tocsr()
        offd = Csr(A.shape[0], 0, np.zeros(A.shape[0] + 1, np.int32), np.zeros(0, np.int32), np.zeros(0))
        return cls([(Csr.from_scipy(A), offd, np.zeros(0, np.int64))], [0, A.shape[0]], [0, A.shape[1]])


class Amg:
    """oamg built from python-side level lists."""

    def __init__(self, A_levels, P_levels, cf_markers, l1_norms, num_grid_sweeps, grid_relax_type,
                 relax_order=0, relax_weight=None, omega=None, cycle_type=1, fcycle=0, num_threads=1,
                 max_levels=25, user_relax_type=-1, grid_relax_points=None, cheby=None, A_outer=None, colors=None):
        """cheby: None or dict(order=, scale=, coefs=[per level array or None], ds=[per level array or None]).
        A_outer: the exact fine-level operator when A_levels hold fp32-rounded values (mixed precision)."""
        L = len(A_levels)
        self.A_outer = A_outer
        self.A_levels, self.P_levels = A_levels, P_levels
        self.cf = [None if c is None else np.ascontiguousarray(c, dtype=np.int32) for c in cf_markers]
        self.l1 = [None if v is None else np.ascontiguousarray(v, dtype=np.float64) for v in l1_norms]
        sizes = [a.nrows for a in A_levels]
        self.F = [np.zeros(max(n, 1)) for n in sizes]
        self.U = [np.zeros(max(n, 1)) for n in sizes]
        self.vtemp = np.zeros(max(sizes[0], 1))
        self.rw = np.ascontiguousarray(relax_weight if relax_weight is not None else np.ones(L), dtype=np.float64)
        self.om = np.ascontiguousarray(omega if omega is not None else np.ones(L), dtype=np.float64)
        self._A = (OPAR * L)(*[a.c for a in A_levels])
        self._P = (OPAR * max(L - 1, 1))(*[p.c for p in P_levels]) if L > 1 else (OPAR * 1)()
        self._cf = (IntP * L)(*[_ip(c) if c is not None else None for c in self.cf])
        self._l1 = (RealP * L)(*[_rp(v) if v is not None else None for v in self.l1])
        self._F = (RealP * L)(*[_rp(v) for v in self.F])
        self._U = (RealP * L)(*[_rp(v) for v in self.U])
        self._grp = None
        self._grp_rows = None
        if grid_relax_points is not None:
            self._grp_rows = [np.ascontiguousarray(r, dtype=np.int32) for r in grid_relax_points]
            self._grp = (IntP * 4)(*[_ip(r) for r in self._grp_rows])
        self.c = OAMG(L, max_levels, self._A, self._P, self._cf, self._l1, self._F, self._U, _rp(self.vtemp),
                      (C.c_int * 4)(*num_grid_sweeps), (C.c_int * 4)(*grid_relax_type), self._grp, relax_order,
                      user_relax_type, _rp(self.rw), _rp(self.om), cycle_type, fcycle, num_threads)
        if A_outer is not None:
            self._Ao = (OPAR * 1)(A_outer.c)
            self.c.A_outer = self._Ao
        self.colors = None
        if colors is not None:
            # multicolour Gauss-Seidel (relax 21 / 22): colour of every row, per level
            self.colors = [None if c is None else np.ascontiguousarray(c, dtype=np.int32) for c in colors]
            self._colors = (IntP * L)(*[_ip(c) if c is not None else None for c in self.colors])
            self.c.colors = self._colors
        if cheby is not None:
            self.cheby_coefs = [None if v is None else np.ascontiguousarray(v, dtype=np.float64) for v in cheby["coefs"]]
            self.cheby_ds = [None if v is None else np.ascontiguousarray(v, dtype=np.float64) for v in cheby["ds"]]
            self._cc = (RealP * L)(*[_rp(v) if v is not None else None for v in self.cheby_coefs])
            self._cd = (RealP * L)(*[_rp(v) if v is not None else None for v in self.cheby_ds])
            self.c.cheby_order, self.c.cheby_scale = int(cheby["order"]), int(cheby["scale"])
            self.c.cheby_coefs, self.c.cheby_ds = self._cc, self._cd

    def cycle(self, f, u, u_all_zeros=False):
        L = load()
        f = np.ascontiguousarray(f, dtype=np.float64)
        self._F[0] = _rp(f)
        self._U[0] = _rp(u)
        az = C.c_int(1 if u_all_zeros else 0)
        err = L.oracle_amg_cycle(C.byref(self.c), self._F, self._U, C.byref(az))
        self._F[0] = _rp(self.F[0])
        self._U[0] = _rp(self.U[0])
        return err

    def solve(self, f, u, tol=1e-8, min_iter=0, max_iter=20, converge_type=0, u_all_zeros=False):
        L = load()
        f = np.ascontiguousarray(f, dtype=np.float64)
        rel = C.c_double(0.0)
        conv = C.c_int(0)
        hist = np.zeros(max_iter + 2)
        its = L.oracle_amg_solve(C.byref(self.c), _rp(f), _rp(u), tol, min_iter, max_iter, converge_type,
                                 1 if u_all_zeros else 0, C.byref(rel), C.byref(conv), _rp(hist))
        return its, rel.value, conv.value, hist[:its + 1]

    def pcg(self, b, x, tol=1e-8, atol=0.0, max_iter=1000, two_norm=1, precond_cycles=1, flex=0):
        L = load()
        b = np.ascontiguousarray(b, dtype=np.float64)
        rel = C.c_double(0.0)
        conv = C.c_int(0)
        its = L.oracle_pcg_amg_flex(C.byref(self.c), _rp(b), _rp(x), tol, atol, max_iter, two_norm, precond_cycles,
                                    int(flex), C.byref(rel), C.byref(conv))
        return its, rel.value, conv.value

    def gmres(self, b, x, tol=1e-8, atol=0.0, max_iter=1000, k_dim=5, precond_cycles=1):
        L = load()
        b = np.ascontiguousarray(b, dtype=np.float64)
        rel = C.c_double(0.0)
        conv = C.c_int(0)
        its = L.oracle_gmres_amg(C.byref(self.c), _rp(b), _rp(x), tol, atol, max_iter, k_dim, precond_cycles,
                                 C.byref(rel), C.byref(conv))
        return its, rel.value, conv.value


def csr_matvec(alpha, A, x, beta, b, y, offset=0):
    L = load()
    return L.oracle_csr_matvec(alpha, C.byref(A.c), _rp(x), len(x), beta, _rp(b), len(b), _rp(y), len(y), offset)


def csr_matvecT(alpha, A, x, beta, y):
    L = load()
    return L.oracle_csr_matvecT(alpha, C.byref(A.c), _rp(x), len(x), beta, _rp(y), len(y))


def par_matvec(alpha, A, x, beta, b, y):
    return load().oracle_par_matvec(alpha, C.byref(A.c), _rp(x), beta, _rp(b), _rp(y))


def pcg_ds_multi(A, B_cols, X_cols, tol=1e-8, atol=0.0, max_iter=1000, two_norm=1):
    """DS-PCG on the multivector whose columns are the columns of B_cols (n x nv), started from X_cols (n x nv, overwritten
    with the solution): `ij -solver 2 -nc nv`.  Returns (iterations, final relative residual, converged)."""
    n, nv = B_cols.shape
    b = np.ascontiguousarray(B_cols.T, dtype=np.float64).ravel()
    x = np.ascontiguousarray(X_cols.T, dtype=np.float64).ravel()
    rel = C.c_double(0.0)
    conv = C.c_int(0)
    its = load().oracle_pcg_ds_multi(C.byref(A.c), _rp(b), _rp(x), nv, tol, atol, max_iter, two_norm, C.byref(rel), C.byref(conv))
    X_cols[:, :] = x.reshape(nv, n).T
    return its, rel.value, conv.value


def gmres_ds_multi(A, B_cols, X_cols, tol=1e-8, atol=0.0, max_iter=1000, k_dim=5):
    """DS-GMRES(k_dim) on a multivector (`ij -solver 4 -nc nv`); arguments as pcg_ds_multi."""
    n, nv = B_cols.shape
    b = np.ascontiguousarray(B_cols.T, dtype=np.float64).ravel()
    x = np.ascontiguousarray(X_cols.T, dtype=np.float64).ravel()
    rel = C.c_double(0.0)
    conv = C.c_int(0)
    its = load().oracle_gmres_ds_multi(C.byref(A.c), _rp(b), _rp(x), nv, tol, atol, max_iter, k_dim, C.byref(rel), C.byref(conv))
    X_cols[:, :] = x.reshape(nv, n).T
    return its, rel.value, conv.value


def par_matvecT(alpha, A, x, beta, y):
    return load().oracle_par_matvecT(alpha, C.byref(A.c), _rp(x), beta, _rp(y))


def l1_norms(A, option, cf_marker=None):
    out = np.zeros(A.nrows)
    cf = None if cf_marker is None else np.ascontiguousarray(cf_marker, dtype=np.int32)
    bad = load().oracle_l1_norms(C.byref(A.c), option, _ip(cf) if cf is not None else None, _rp(out))
    return out, bad


def relax(A, f, cf_marker, relax_type, relax_points, w, omega, l1, u, num_threads=1, all_zeros=False, colors=None):
    """colors: colour of every row, for the multicolour sweeps 21 / 22."""
    cf = None if cf_marker is None else np.ascontiguousarray(cf_marker, dtype=np.int32)
    if colors is not None:
        colors = np.ascontiguousarray(colors, dtype=np.int32)
    load().oracle_set_multicolor(_ip(colors) if colors is not None else None)
    vtemp = np.zeros(max(A.nrows, 1))
    az = C.c_int(1 if all_zeros else 0)
    f = np.ascontiguousarray(f, dtype=np.float64)
    l1 = None if l1 is None else np.ascontiguousarray(l1, dtype=np.float64)
    err = load().oracle_relax(C.byref(A.c), _rp(f), _ip(cf) if cf is not None else None, relax_type,
                              relax_points, w, omega, _rp(l1), _rp(u), _rp(vtemp), num_threads, C.byref(az))
    return err


# ---------------------------------------------------------------------------
# bridges from the product's objects (fetched through the C ABI) to oracle structs
# ---------------------------------------------------------------------------
def par_from_handles(handles, fp32_diag_values=False):
    """handles: list (one per rank, rank order) of hypre_ParCSRMatrix pointers
    (ctypes POINTER(ParCSRMatrix) or raw addresses) living in this process.
    fp32_diag_values: round the matrix values (diagonal and ghost block) to fp32 (the product's mixed-precision mode
    streams fp32 copies of those values and accumulates in fp64, which is this matrix in exact arithmetic)."""
    import ctypes as C
    from hypre_amd import binding as B
    blocks, rs, cs = [], [], []
    last_r = last_c = 0
    for h in handles:
        if not isinstance(h, C.POINTER(B.ParCSRMatrix)):
            h = C.cast(h, C.POINTER(B.ParCSRMatrix))
        m = h.contents
        di, dj, da = B.csr_to_arrays(m.diag)
        if fp32_diag_values:
            da = da.astype(np.float32).astype(np.float64)
        oi, oj, oa = B.csr_to_arrays(m.offd)
        if fp32_diag_values:
            oa = oa.astype(np.float32).astype(np.float64)
        nco = m.offd.contents.num_cols
        cmap = np.array([m.col_map_offd[k] for k in range(nco)], dtype=np.int64)
        rn = np.nonzero(np.diff(oi))[0].astype(np.int32)
        with_rn = rn if (0 < len(rn) < m.offd.contents.num_rows) else None
        d = Csr(m.diag.contents.num_rows, m.diag.contents.num_cols, di, dj, da)
        o = Csr(m.offd.contents.num_rows, nco, oi, oj, oa, with_rn)
        blocks.append((d, o, cmap))
        rs.append(int(m.row_starts[0])); cs.append(int(m.col_starts[0]))
        last_r, last_c = int(m.row_starts[1]), int(m.col_starts[1])
    rs.append(last_r); cs.append(last_c)
    return Par(blocks, rs, cs)


def _export_cheby(s, nl):
    """Chebyshev level data of one solver: (order, scale, [coefs or None], [ds or None])."""
    import ctypes as C
    from hypre_amd import binding as B
    L = B.load_library()
    order, scale = C.c_int(0), C.c_int(0)
    L.hypre_amd_BoomerAMGGetChebyOrderScale(s, C.byref(order), C.byref(scale))
    k = min(max(order.value, 1), 4) + 1
    coefs, ds = [], []
    for l in range(nl):
        cp = L.hypre_amd_BoomerAMGGetChebyCoefs(s, l)
        coefs.append(np.array([cp[i] for i in range(k)]) if cp else None)
        dp = L.hypre_amd_BoomerAMGGetChebyDS(s, l)
        if dp:
            v = C.cast(dp, C.POINTER(B.Vector)).contents
            ds.append(B.fetch(v.data, v.size, np.float64, v.memory_location))
        else:
            ds.append(None)
    if all(c is None for c in coefs):
        return None
    return dict(order=order.value, scale=scale.value, coefs=coefs, ds=ds)


def _merge_cheby(parts):
    """Per-rank Chebyshev exports -> one global description (coefficients are rank-invariant)."""
    if not parts or any(p is None for p in parts):
        return None
    nl = len(parts[0]["coefs"])
    ds = []
    for l in range(nl):
        v = [p["ds"][l] for p in parts]
        ds.append(np.concatenate(v) if all(x is not None for x in v) else None)
    return dict(order=parts[0]["order"], scale=parts[0]["scale"], coefs=parts[0]["coefs"], ds=ds)


def amg_from_solvers(solvers, num_threads=1, mixed_precision=False):
    """Build the oracle's hierarchy from one product solver per (virtual) rank.
    mixed_precision: model the product's fp32-matrix-value mode (operators and interpolation of every
    level but the coarsest, whose dense solve stays fp64)."""
    import ctypes as C
    from hypre_amd import binding as B
    L = B.load_library()
    nl = L.hypre_amd_BoomerAMGGetNumLevels(solvers[0])
    A_levels, P_levels, cfs, l1s = [], [], [], []
    for l in range(nl):
        A_levels.append(par_from_handles([L.hypre_amd_BoomerAMGGetA(s, l) for s in solvers],
                                         fp32_diag_values=mixed_precision and l < nl - 1))
        if l < nl - 1:
            P_levels.append(par_from_handles([L.hypre_amd_BoomerAMGGetP(s, l) for s in solvers],
                                             fp32_diag_values=mixed_precision))
        cf_parts, l1_parts = [], []
        for s in solvers:
            cfp = L.hypre_amd_BoomerAMGGetCFMarker(s, l)
            if cfp:
                ia = C.cast(cfp, C.POINTER(B.IntArray)).contents
                cf_parts.append(B.fetch(ia.data, ia.size, np.int32, ia.memory_location))
            lp = L.hypre_amd_BoomerAMGGetL1Norms(s, l)
            if lp:
                v = C.cast(lp, C.POINTER(B.Vector)).contents
                l1_parts.append(B.fetch(v.data, v.size, np.float64, v.memory_location))
        cfs.append(np.concatenate(cf_parts) if cf_parts else None)
        l1s.append(np.concatenate(l1_parts) if l1_parts else None)
    s0 = solvers[0]
    sweeps = [L.hypre_amd_BoomerAMGGetNumGridSweeps(s0, k) for k in range(4)]
    types = [L.hypre_amd_BoomerAMGGetGridRelaxType(s0, k) for k in range(4)]
    d = C.cast(s0, C.POINTER(AmgDataView)).contents
    rw = np.array([d.relax_weight[k] for k in range(nl)])
    om = np.array([d.omega[k] for k in range(nl)])
    cheby = _merge_cheby([_export_cheby(s, nl) for s in solvers])
    A_outer = par_from_handles([L.hypre_amd_BoomerAMGGetA(s, 0) for s in solvers]) if mixed_precision else None
    colors = None
    if any(t in (21, 22) for t in types):
        colors = [np.concatenate([level_colors(L.hypre_amd_BoomerAMGGetA(s, l)) for s in solvers]) for l in range(nl)]
    return Amg(A_levels, P_levels, cfs, l1s, sweeps, types, relax_order=d.relax_order, relax_weight=rw, omega=om,
               cycle_type=d.cycle_type, fcycle=d.fcycle, num_threads=num_threads, max_levels=d.max_levels,
               user_relax_type=d.user_relax_type, cheby=cheby, A_outer=A_outer, colors=colors)


def level_colors(handle):
    """The product's colouring of one rank's diagonal block (device matrices only)."""
    import ctypes as C
    from hypre_amd import binding as B
    L = B.load_library()
    h = handle if isinstance(handle, C.POINTER(B.ParCSRMatrix)) else C.cast(handle, C.POINTER(B.ParCSRMatrix))
    n = h.contents.diag.contents.num_rows
    out = np.zeros(max(n, 1), dtype=np.int32)
    L.hypre_amd_ParCSRMatrixMultiColoring(h, _ip(out))
    B.check()
    return out[:n]


def export_par(h):
    """Picklable numpy view of one rank's block of a hypre_ParCSRMatrix."""
    import ctypes as C
    from hypre_amd import binding as B
    if not isinstance(h, C.POINTER(B.ParCSRMatrix)):
        h = C.cast(h, C.POINTER(B.ParCSRMatrix))
    m = h.contents
    di, dj, da = B.csr_to_arrays(m.diag)
    oi, oj, oa = B.csr_to_arrays(m.offd)
    nco = m.offd.contents.num_cols
    cmap = np.array([m.col_map_offd[k] for k in range(nco)], dtype=np.int64)
    return dict(di=di, dj=dj, da=da, oi=oi, oj=oj, oa=oa, cmap=cmap, nrows=m.diag.contents.num_rows,
                ncols=m.diag.contents.num_cols, nco=nco, rs=(int(m.row_starts[0]), int(m.row_starts[1])),
                cs=(int(m.col_starts[0]), int(m.col_starts[1])))


def par_from_exports(parts, fp32_diag_values=False):
    blocks, rs, cs = [], [], []
    for d in parts:
        rn = np.nonzero(np.diff(d["oi"]))[0].astype(np.int32)
        with_rn = rn if (0 < len(rn) < d["nrows"]) else None
        da = d["da"].astype(np.float32).astype(np.float64) if fp32_diag_values else d["da"]
        oa = d["oa"].astype(np.float32).astype(np.float64) if fp32_diag_values else d["oa"]
        blocks.append((Csr(d["nrows"], d["ncols"], d["di"], d["dj"], da),
                       Csr(d["nrows"], d["nco"], d["oi"], d["oj"], oa, with_rn), d["cmap"]))
        rs.append(d["rs"][0]); cs.append(d["cs"][0])
    rs.append(parts[-1]["rs"][1]); cs.append(parts[-1]["cs"][1])
    return Par(blocks, rs, cs)


def export_solver(s):
    """This rank's share of the hierarchy as plain numpy (for gathering to one process)."""
    import ctypes as C
    from hypre_amd import binding as B
    L = B.load_library()
    nl = L.hypre_amd_BoomerAMGGetNumLevels(s)
    out = dict(num_levels=nl, A=[], P=[], cf=[], l1=[])
    for l in range(nl):
        out["A"].append(export_par(L.hypre_amd_BoomerAMGGetA(s, l)))
        if l < nl - 1:
            out["P"].append(export_par(L.hypre_amd_BoomerAMGGetP(s, l)))
        cfp = L.hypre_amd_BoomerAMGGetCFMarker(s, l)
        if cfp:
            ia = C.cast(cfp, C.POINTER(B.IntArray)).contents
            out["cf"].append(B.fetch(ia.data, ia.size, np.int32, ia.memory_location))
        else:
            out["cf"].append(None)
        lp = L.hypre_amd_BoomerAMGGetL1Norms(s, l)
        if lp:
            v = C.cast(lp, C.POINTER(B.Vector)).contents
            out["l1"].append(B.fetch(v.data, v.size, np.float64, v.memory_location))
        else:
            out["l1"].append(None)
    d = C.cast(s, C.POINTER(AmgDataView)).contents
    out["sweeps"] = [L.hypre_amd_BoomerAMGGetNumGridSweeps(s, k) for k in range(4)]
    out["types"] = [L.hypre_amd_BoomerAMGGetGridRelaxType(s, k) for k in range(4)]
    out["rw"] = np.array([d.relax_weight[k] for k in range(nl)])
    out["om"] = np.array([d.omega[k] for k in range(nl)])
    out["relax_order"], out["cycle_type"], out["fcycle"] = d.relax_order, d.cycle_type, d.fcycle
    out["max_levels"], out["user_relax_type"] = d.max_levels, d.user_relax_type
    out["cheby"] = _export_cheby(s, nl)
    out["colors"] = None
    if any(t in (21, 22) for t in out["types"]):
        out["colors"] = [level_colors(L.hypre_amd_BoomerAMGGetA(s, l)) for l in range(nl)]
    return out


def amg_from_exports(parts, num_threads=1, mixed_precision=False):
    """parts: export_solver() dicts of all ranks in rank order.  mixed_precision as in amg_from_solvers."""
    p0 = parts[0]
    nl = p0["num_levels"]
    A_levels = [par_from_exports([p["A"][l] for p in parts], fp32_diag_values=mixed_precision and l < nl - 1)
                for l in range(nl)]
    P_levels = [par_from_exports([p["P"][l] for p in parts], fp32_diag_values=mixed_precision) for l in range(nl - 1)]
    cfs, l1s = [], []
    for l in range(nl):
        c = [p["cf"][l] for p in parts]
        cfs.append(np.concatenate(c) if all(x is not None for x in c) else None)
        v = [p["l1"][l] for p in parts]
        l1s.append(np.concatenate(v) if all(x is not None for x in v) else None)
    A_outer = par_from_exports([p["A"][0] for p in parts]) if mixed_precision else None
    return Amg(A_levels, P_levels, cfs, l1s, p0["sweeps"], p0["types"], relax_order=p0["relax_order"],
               relax_weight=p0["rw"], omega=p0["om"], cycle_type=p0["cycle_type"], fcycle=p0["fcycle"],
               num_threads=num_threads, max_levels=p0["max_levels"], user_relax_type=p0["user_relax_type"],
               cheby=_merge_cheby([p.get("cheby") for p in parts]), A_outer=A_outer,
               colors=None if p0.get("colors") is None else [np.concatenate([p["colors"][l] for p in parts]) for l in range(nl)])


class AmgDataView(C.Structure):
    """Leading members of hypre_ParAMGData (include/hypre_amd_parcsr_ls.h) up to omega."""
    _fields_ = [("setup", C.c_void_p), ("solve", C.c_void_p), ("destroy", C.c_void_p), ("memory_location", C.c_int),
                ("max_levels", C.c_int), ("strong_threshold", C.c_double), ("max_row_sum", C.c_double),
                ("trunc_factor", C.c_double), ("measure_type", C.c_int), ("coarsen_type", C.c_int),
                ("P_max_elmts", C.c_int), ("interp_type", C.c_int), ("agg_num_levels", C.c_int),
                ("max_coarse_size", C.c_int), ("min_coarse_size", C.c_int), ("keepTranspose", C.c_int),
                ("num_functions", C.c_int), ("max_iter", C.c_int), ("min_iter", C.c_int), ("fcycle", C.c_int),
                ("cycle_type", C.c_int), ("num_grid_sweeps", IntP), ("grid_relax_type", IntP),
                ("grid_relax_points", C.c_void_p), ("relax_order", C.c_int), ("user_coarse_relax_type", C.c_int),
                ("user_relax_type", C.c_int), ("user_num_sweeps", C.c_int), ("user_relax_weight", C.c_double),
                ("outer_wt", C.c_double), ("relax_weight", RealP), ("omega", RealP)]
