"""Multicolour Gauss-Seidel on the device (relax 21 / 22; BASELINE north_star) against the oracle.

The reference has no colouring, so parity is stated through what it does have: the oracle's relax 21 / 22
are its hybrid Gauss-Seidel sweeps on the colour-permuted ordering (pinned on the CPU side by
tests/test_oracle_basic.py::test_multicolor_sweep_is_hybrid_gauss_seidel_on_the_colour_permuted_system), and the device
sweep — one fused pass of the tiled SpMV kernel per colour, in place — is compared with them here on the colouring the
library itself computed.  fp64, 1e-12 relative max-norm for a sweep, 1e-11 for a cycle."""
import ctypes as C

import numpy as np
import pytest

from util import rand_vector

pytestmark = pytest.mark.gpu


def _setup(lib, **kw):
    from hypre_amd import binding as B, ij
    opt = ij.IJOptions(**kw)
    A = ij.build_matrix(opt)
    s = ij.create_amg(opt, memory_location=B.HYPRE_MEMORY_DEVICE)
    lib.HYPRE_BoomerAMGSetup(s, A, None, None)
    B.check()
    lib.hypre_ParCSRMatrixMigrate(A, B.HYPRE_MEMORY_DEVICE)
    return opt, A, s


def _level(lib, s, l):
    from hypre_amd import binding as B
    A = C.cast(lib.hypre_amd_BoomerAMGGetA(s, l), C.POINTER(B.ParCSRMatrix))
    cfp = lib.hypre_amd_BoomerAMGGetCFMarker(s, l)
    l1p = lib.hypre_amd_BoomerAMGGetL1Norms(s, l)
    cf = C.cast(cfp, C.POINTER(B.IntArray)).contents.data if cfp else None
    l1 = C.cast(l1p, C.POINTER(B.Vector)).contents.data if l1p else None
    return A, cf, l1


@pytest.mark.parametrize("problem,ncolors", [("laplacian", 2), ("27pt", 8)])
def test_colouring_is_proper_and_small(gpu_lib, oracle, problem, ncolors):
    """Greedy first-fit in row order: red-black on the 7-point grid, 8 colours on the 27-point one; on every level no
    two coupled rows share a colour."""
    from hypre_amd import binding as B
    lib = gpu_lib
    opt, A0, s = _setup(lib, n=(12, 11, 10), problem=problem, relax_type=21, coarsen_type=8)
    nl = lib.hypre_amd_BoomerAMGGetNumLevels(s)
    for l in range(nl):
        A, cf, l1 = _level(lib, s, l)
        col = oracle.level_colors(A)
        M = B.csr_to_scipy(A.contents.diag).tocoo()
        off = M.row != M.col
        assert np.all(col[M.row[off]] != col[M.col[off]])
        if l == 0:
            assert col.max() + 1 == ncolors
    lib.HYPRE_BoomerAMGDestroy(s)


@pytest.mark.parametrize("relax_type", [21, 22])
@pytest.mark.parametrize("relax_points,w,zero", [(0, 1.0, False), (0, 1.0, True), (1, 1.0, False), (-1, 1.0, False), (0, 0.8, False)])
@pytest.mark.parametrize("level", [0, 1, 2])
@pytest.mark.parametrize("problem", ["laplacian", "27pt"])
def test_single_sweep(gpu_lib, oracle, relax_type, relax_points, w, zero, level, problem):
    from hypre_amd import binding as B
    lib = gpu_lib
    opt, A0, s = _setup(lib, n=(14, 13, 12), problem=problem, relax_type=relax_type, coarsen_type=8,
                        relax_order=1 if relax_points else 0)
    A, cf, l1 = _level(lib, s, level)
    amg = oracle.amg_from_solvers([s])
    n = amg.A_levels[level].nrows
    f = rand_vector(n, 3)
    u0 = np.zeros(n) if zero else rand_vector(n, 4)
    du, df, dv = B.parvec_from_numpy(u0), B.parvec_from_numpy(f), B.parvec_from_numpy(np.zeros(n))
    if zero:
        lib.hypre_ParVectorSetZeros(du)
    err = lib.hypre_BoomerAMGRelax(A, df, cf, relax_type, relax_points, w, 1.0, l1, du, dv, dv)
    B.check()
    assert err == 0 and du.contents.all_zeros == 0
    u = B.parvec_to_numpy(du)
    ur = u0.copy()
    assert oracle.relax(amg.A_levels[level], f, amg.cf[level], relax_type, relax_points, w, 1.0, amg.l1[level], ur,
                        all_zeros=zero, colors=amg.colors[level]) == 0
    assert np.max(np.abs(u - ur)) <= 1e-12 * max(1.0, np.max(np.abs(ur)))
    lib.HYPRE_BoomerAMGDestroy(s)


@pytest.mark.parametrize("problem,n,level", [("laplacian", (14, 13, 12), 1), ("27pt", (14, 13, 12), 1), ("laplacian", (40, 40, 36), 2),
                                             ("laplacian", (40, 40, 36), 3), ("27pt", (30, 28, 26), 1)])
@pytest.mark.parametrize("relax_type,relax_points", [(21, 0), (22, 0), (21, 1), (22, -1)])
def test_look_ahead_sweeps_are_the_plain_sweeps(gpu_lib, oracle, problem, n, level, relax_type, relax_points):
    """hypre_amd_SetMcLookAhead: the one-workgroup sweeps (whole small levels; the tails of small colours of larger levels)
    with everything that does not depend on the iterate requested a pass ahead (and 32 lanes a row instead of 8) are the plain
    sweeps up to the order of a row's sum: forward and backward, all points and C / F points only, rows longer than what the
    lanes fetch ahead."""
    from hypre_amd import binding as B
    lib = gpu_lib
    opt, A0, s = _setup(lib, n=n, problem=problem, relax_type=relax_type, coarsen_type=8, relax_order=1 if relax_points else 0)
    nl = lib.hypre_amd_BoomerAMGGetNumLevels(s)
    if level >= nl - 1:
        level = nl - 2
    A, cf, l1 = _level(lib, s, level)
    nrow = A.contents.diag.contents.num_rows
    f = rand_vector(nrow, 3)
    u0 = rand_vector(nrow, 4)
    out = {}
    try:
        for on in (1, 0):
            assert lib.hypre_amd_SetMcLookAhead(on) == on
            du, df, dv = B.parvec_from_numpy(u0), B.parvec_from_numpy(f), B.parvec_from_numpy(np.zeros(nrow))
            for _ in range(2):
                err = lib.hypre_BoomerAMGRelax(A, df, cf, relax_type, relax_points, 0.9, 1.0, l1, du, dv, dv)
                assert err == 0
            B.check()
            out[on] = B.parvec_to_numpy(du)
            for v in (du, df, dv):
                lib.hypre_ParVectorDestroy(v)
    finally:
        lib.hypre_amd_SetMcLookAhead(1)
    # (the same products; a row's sum is added by 32 lanes instead of 8: rounding)
    assert np.max(np.abs(out[1] - out[0])) <= 1e-13 * np.max(np.abs(out[0]))
    lib.HYPRE_BoomerAMGDestroy(s)


def test_sweep_equals_the_level_scheduled_gauss_seidel_on_the_permuted_matrix(gpu_lib):
    """The parity statement end to end on the device: multicolour sweep of A == the library's own (bit-exact, golden-
    pinned) hybrid Gauss-Seidel sweep, relax 3, of P A P^T."""
    import scipy.sparse as sp
    from hypre_amd import binding as B
    lib = gpu_lib
    P0 = B.laplacian(9, 8, 7, kind="27pt")
    A = B.csr_to_scipy(P0.contents.diag)
    n = A.shape[0]
    lib.hypre_ParCSRMatrixMigrate(P0, B.HYPRE_MEMORY_DEVICE)
    colors = np.zeros(n, dtype=np.int32)
    nc = lib.hypre_amd_ParCSRMatrixMultiColoring(P0, colors.ctypes.data_as(C.POINTER(C.c_int)))
    B.check()
    assert nc == colors.max() + 1 == 8
    order = np.lexsort((np.arange(n), colors))
    inv = np.empty(n, dtype=np.int64); inv[order] = np.arange(n)
    cols, vals, indptr = [], [], [0]
    for pi in range(n):
        i = order[pi]
        js = A.indices[A.indptr[i]:A.indptr[i + 1]]; vs = A.data[A.indptr[i]:A.indptr[i + 1]]
        k = list(js).index(i)
        cols += [pi] + [inv[j] for q, j in enumerate(js) if q != k]
        vals += [vs[k]] + [v for q, v in enumerate(vs) if q != k]
        indptr.append(len(cols))
    ii, jj, aa = np.array(indptr, dtype=np.int32), np.array(cols, dtype=np.int32), np.array(vals)
    part = np.array([0, n], dtype=np.int64)
    z = np.zeros(n + 1, dtype=np.int32)
    Pp = lib.hypre_amd_ParCSRMatrixFromArrays(0, n, n, B._bp(part), B._bp(part), 0, None, B._ip(ii), B._ip(jj), B._rp(aa),
                                             B._ip(z), None, None, B.HYPRE_MEMORY_DEVICE)
    B.check()
    f, u0 = rand_vector(n, 3), rand_vector(n, 4)
    du, df, dv = B.parvec_from_numpy(u0), B.parvec_from_numpy(f), B.parvec_from_numpy(np.zeros(n))
    lib.hypre_BoomerAMGRelax(P0, df, None, 21, 0, 1.0, 1.0, None, du, dv, dv)
    dup, dfp = B.parvec_from_numpy(u0[order]), B.parvec_from_numpy(f[order])
    dw = B.parvec_from_numpy(np.zeros(n))
    lib.hypre_BoomerAMGRelax(Pp, dfp, None, 3, 0, 1.0, 1.0, None, dup, dv, dw)
    B.check()
    u, up = B.parvec_to_numpy(du), B.parvec_to_numpy(dup)
    assert np.max(np.abs(u[order] - up)) <= 1e-13 * np.max(np.abs(up))


@pytest.mark.parametrize("kw", [
    dict(relax_type=21),
    dict(relax_down=21, relax_up=22),
    dict(relax_down=21, relax_up=22, problem="27pt"),
    dict(relax_type=21, relax_order=1),
    dict(relax_down=21, relax_up=22, cycle_type=2),
    dict(relax_down=21, relax_up=22, relax_wt=0.9, num_sweeps=2),
])
def test_one_cycle_matches_oracle(gpu_lib, oracle, kw):
    from hypre_amd import binding as B
    lib = gpu_lib
    opt, A, s = _setup(lib, n=(16, 15, 14), coarsen_type=8, **kw)
    amg = oracle.amg_from_solvers([s])
    n = amg.A_levels[0].nrows
    f = rand_vector(n, 5)
    for zero in (True, False):
        u0 = np.zeros(n) if zero else rand_vector(n, 6)
        du, df = B.parvec_from_numpy(u0), B.parvec_from_numpy(f)
        if zero:
            lib.hypre_ParVectorSetZeros(du)
        lib.HYPRE_BoomerAMGSetTol(s, 0.0)
        lib.HYPRE_BoomerAMGSetMaxIter(s, 1)
        lib.HYPRE_BoomerAMGSolve(s, A, df, du)
        B.check()
        u = B.parvec_to_numpy(du)
        ur = u0.copy()
        amg.cycle(f, ur, u_all_zeros=zero)
        assert np.max(np.abs(u - ur)) <= 1e-11 * np.max(np.abs(ur)), (kw, zero)
    lib.HYPRE_BoomerAMGDestroy(s)


def test_pcg_with_symmetric_multicolour_smoothing(gpu_lib, oracle):
    """Colours ascending on the way down, descending on the way up: a symmetric preconditioner, so PCG applies; same
    iteration count and final residual as the oracle, and fewer iterations than with l1-Jacobi."""
    from hypre_amd import binding as B, ij
    lib = gpu_lib

    def pcg(**kw):
        opt, A, s = _setup(lib, n=(32, 32, 32), coarsen_type=8, solver=1, **kw)
        lib.HYPRE_BoomerAMGSetTol(s, 0.0)
        lib.HYPRE_BoomerAMGSetMaxIter(s, 1)
        b, x0 = ij.build_rhs_host(opt, A)
        dx, db = B.parvec_from_numpy(x0), B.parvec_from_numpy(b)
        h = C.c_void_p()
        lib.HYPRE_ParCSRPCGCreate(0, C.byref(h))
        lib.HYPRE_PCGSetTol(h, opt.tol)
        lib.HYPRE_PCGSetMaxIter(h, opt.max_iter)
        lib.HYPRE_PCGSetTwoNorm(h, 1)
        lib.HYPRE_PCGSetPrecond(h, C.cast(lib.HYPRE_BoomerAMGSolve, C.c_void_p), None, s)
        lib.HYPRE_ParCSRPCGSetup(h, A, db, dx)
        lib.HYPRE_ParCSRPCGSolve(h, A, db, dx)
        its, rel = C.c_int(), C.c_double()
        lib.HYPRE_PCGGetNumIterations(h, C.byref(its))
        lib.HYPRE_PCGGetFinalRelativeResidualNorm(h, C.byref(rel))
        B.check()
        amg = oracle.amg_from_solvers([s])
        xo = x0.copy()
        oits, orel, _ = amg.pcg(b, xo, tol=opt.tol, max_iter=opt.max_iter, two_norm=1)
        lib.HYPRE_ParCSRPCGDestroy(h)
        lib.HYPRE_BoomerAMGDestroy(s)
        return its.value, rel.value, oits, orel

    its, rel, oits, orel = pcg(relax_down=21, relax_up=22)
    assert its == oits and abs(rel - orel) <= 1e-6 * orel
    jits = pcg(relax_type=18)[0]
    assert its < jits


@pytest.mark.parametrize("relax_type", [21, 3])
def test_caches_keyed_by_the_matrix_address_notice_another_matrix(gpu_lib, relax_type):
    """The colour classes of relax 21 hold a colour-sorted COPY of the matrix, the level schedule of relax 3 the dependency
    levels of its pattern; both are found again by the matrix's address.  A different matrix at the same address (the caller
    freed the first with hypre's own destroy routine and built the next one of the same size) must not be swept with the old
    one's classes silently: every sweep compares a sampled fingerprint of the matrix with the one taken when the cache was
    built; a mismatch raises HYPRE_ERROR_GENERIC at the next call, which rebuilds, and the sweep after that is the new
    matrix's (checked against a fresh copy of it at another address)."""
    import scipy.sparse as sp
    from hypre_amd import binding as B
    lib = gpu_lib
    A1 = B.laplacian(10, 9, 8, kind="27pt")
    M1 = B.csr_to_scipy(A1.contents.diag)
    n = M1.shape[0]
    # relax 21 (the classes copy the values): other values on the same pattern
    M2 = sp.csr_matrix((M1.data * 1.5, M1.indices.copy(), M1.indptr), shape=M1.shape)
    if relax_type == 3:
        # another PATTERN of the same size for the level schedule: reverse the off-diagonal part of every row
        idx = M1.indices.copy()
        for r in range(n):
            b, e = M1.indptr[r], M1.indptr[r + 1]
            idx[b + 1:e] = (n - 1 - M1.indices[b + 1:e])[::-1] if r % 2 else M1.indices[b + 1:e]
            # a reflected column may coincide with the diagonal: keep such rows as they were
            if r % 2 and (r in idx[b + 1:e] or len(set(idx[b + 1:e])) != e - b - 1):
                idx[b + 1:e] = M1.indices[b + 1:e]
        M2 = sp.csr_matrix((M1.data.copy(), idx.astype(np.int32), M1.indptr), shape=M1.shape)
    lib.hypre_ParCSRMatrixMigrate(A1, B.HYPRE_MEMORY_DEVICE)
    f = rand_vector(n, 7)
    u0 = rand_vector(n, 8)

    def sweep(Apar):
        du, df, dv, dz = (B.parvec_from_numpy(u0), B.parvec_from_numpy(f), B.parvec_from_numpy(np.zeros(n)), B.parvec_from_numpy(np.zeros(n)))
        lib.hypre_BoomerAMGRelax(Apar, df, None, relax_type, 0, 1.0, 1.0, None, du, dv, dz)
        lib.hypre_SyncComputeStream()
        return B.parvec_to_numpy(du)

    sweep(A1)                                              # builds the cache for the matrix at this address
    B.check()
    d = A1.contents.diag.contents
    for dst, src in ((d.j, np.ascontiguousarray(M2.indices, dtype=np.int32)), (d.data, np.ascontiguousarray(M2.data))):
        lib.hypre_Memcpy(C.cast(dst, C.c_void_p), src.ctypes.data_as(C.c_void_p), src.nbytes, B.HYPRE_MEMORY_DEVICE, B.HYPRE_MEMORY_HOST)
    sweep(A1)                                              # swept with the old cache: the launch notices ...
    sweep(A1)                                              # ... this call reports it and rebuilds
    assert lib.HYPRE_GetError() & 1
    lib.HYPRE_ClearAllErrors()
    got = sweep(A1)
    B.check()
    # the same matrix, built fresh at another address
    A2 = B.laplacian(10, 9, 8, kind="27pt")
    d2 = A2.contents.diag.contents
    for dst, src in ((d2.j, np.ascontiguousarray(M2.indices, dtype=np.int32)), (d2.data, np.ascontiguousarray(M2.data))):
        lib.hypre_Memcpy(C.cast(dst, C.c_void_p), src.ctypes.data_as(C.c_void_p), src.nbytes, B.HYPRE_MEMORY_HOST, B.HYPRE_MEMORY_HOST)
    lib.hypre_ParCSRMatrixMigrate(A2, B.HYPRE_MEMORY_DEVICE)
    want = sweep(A2)
    B.check()
    assert np.array_equal(got, want) if relax_type == 3 else np.max(np.abs(got - want)) <= 1e-13 * np.max(np.abs(want))
    lib.hypre_ParCSRMatrixDestroy(A1)
    lib.hypre_ParCSRMatrixDestroy(A2)
    B.check()
