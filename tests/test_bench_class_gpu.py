"""The code path the benchmark runs, under test: placement tables (tile -> XCD, `tile_perm`) with a tile count
that is not a multiple of 64, grids above 2048 tiles, the 27-point operator with two-stage Gauss-Seidel (config C4's
kernel mix) and with fp32 matrix values (C5's), each against the CPU oracle on the same hierarchy.

Round 1 shipped a kernel whose padded workgroups read past the placement table (a tile ran twice: harmless for the
idempotent epilogues, wrong for the accumulating ones) and no test reached it: band placement needs >= 2048 tiles and
the largest matrix of the suite had 896.  Two remedies here: cases large enough to get a table on their own, and
`hypre_amd_SpmvSetBandPolicy(1, 8, 1)`, which gives small matrices the same table."""
import ctypes as C

import numpy as np
import pytest

from util import rand_vector

pytestmark = pytest.mark.gpu

DEFAULT_VARIANT = 2                           # what the library starts with (seq_mv.cpp: spmv_variant): x staged through LDS


@pytest.fixture(params=["x-staged", "x-gathered"])
def forced_tables(gpu_lib, request):
    """Placement tables forced on every matrix of at least 8 tiles, under both kernels of the tiled family: x staged
    through LDS from the plan's chunk lists (the default) and x gathered through the cache."""
    gpu_lib.hypre_amd_SpmvSetBandPolicy(1, 8, 1)
    gpu_lib.hypre_amd_SpmvSetVariant(2 if request.param == "x-staged" else 0, 0)
    yield gpu_lib
    gpu_lib.hypre_amd_SpmvSetBandPolicy(1, 2048, 0)
    gpu_lib.hypre_amd_SpmvSetVariant(DEFAULT_VARIANT, 0)


def _setup(lib, **kw):
    from hypre_amd import binding as B, ij
    opt = ij.IJOptions(**kw)
    A = ij.build_matrix(opt)
    s = ij.create_amg(opt, memory_location=B.HYPRE_MEMORY_DEVICE)
    lib.HYPRE_BoomerAMGSetup(s, A, None, None)
    B.check()
    lib.hypre_ParCSRMatrixMigrate(A, B.HYPRE_MEMORY_DEVICE)
    return opt, A, s


def _plan_info(lib, csr):
    nt, band = C.c_int(), C.c_int()
    has = lib.hypre_amd_CSRMatrixPlanInfo(csr, C.byref(nt), C.byref(band))
    return bool(has), nt.value, band.value


def _bound(A, x, alpha, beta, b):
    return 1e-13 * (abs(alpha) * (abs(A) @ np.abs(x)) + abs(beta) * np.abs(b)) + 1e-300


@pytest.mark.parametrize("kind,n", [("7pt", (40, 40, 40)), ("27pt", (30, 30, 30)), ("7pt", (31, 29, 23))])
@pytest.mark.parametrize("alpha,beta,inplace", [(1.0, 0.0, False), (-1.0, 1.0, False), (0.7, 0.3, True), (1.0, 1.0, True),
                                                (-1.0, -1.0, True), (2.5, -2.5, False)])
def test_spmv_through_a_placement_table(forced_tables, oracle, kind, n, alpha, beta, inplace):
    """y = alpha A x + beta b with the tiles dealt to the XCDs by a table, out of place and in place (b == y with
    beta != 0: a tile visited twice would apply beta twice)."""
    from hypre_amd import binding as B
    lib = forced_tables
    P = B.laplacian(*n, kind=kind)
    A = B.csr_to_scipy(P.contents.diag)
    dA = B.csr_from_scipy(A)
    has, nt, band = _plan_info(lib, dA)
    assert has and nt % 64 != 0 and band > 0, (has, nt, band)
    x, b = rand_vector(A.shape[1], 1), rand_vector(A.shape[0], 2)
    dx, db, dy = B.vec_from_numpy(x), B.vec_from_numpy(b), B.vec_from_numpy(b if inplace else np.zeros(A.shape[0]))
    if inplace:
        lib.hypre_CSRMatrixMatvec(alpha, dA, dx, beta, dy)
    else:
        lib.hypre_CSRMatrixMatvecOutOfPlace(alpha, dA, dx, beta, db, dy, 0)
    B.check()
    y = B.vec_to_numpy(dy)
    yr = b.copy()
    oracle.csr_matvec(alpha, oracle.Csr.from_scipy(A), x, beta, b, yr)
    assert np.all(np.abs(y - yr) <= _bound(A, x, alpha, beta, b)), float(np.abs(y - yr).max())
    for v in (dx, db, dy):
        lib.hypre_SeqVectorDestroy(v)
    lib.hypre_CSRMatrixDestroy(dA)
    lib.hypre_ParCSRMatrixDestroy(P)


@pytest.mark.parametrize("relax_type,relax_points,zero", [(18, 0, False), (18, 1, False), (18, -1, False), (7, 0, False),
                                                          (0, 0, False), (11, 0, False), (12, 0, False), (11, 0, True),
                                                          (12, 0, True)])
@pytest.mark.parametrize("problem", ["laplacian", "27pt"])
def test_every_epilogue_through_a_placement_table(forced_tables, oracle, relax_type, relax_points, zero, problem):
    """One sweep of every smoother the tiled kernel serves (fused Jacobi, CF-masked Jacobi, the accumulating
    two-stage Gauss-Seidel inner step on the strictly lower copy) with tables on the operator AND on its triangle."""
    from hypre_amd import binding as B
    lib = forced_tables
    opt, A0, s = _setup(lib, n=(26, 25, 24), problem=problem, relax_type=relax_type if relax_type != 0 else 18,
                        coarsen_type=8, relax_order=1 if relax_points else 0)
    Ap = C.cast(lib.hypre_amd_BoomerAMGGetA(s, 0), C.POINTER(B.ParCSRMatrix))
    has, nt, band = _plan_info(lib, Ap.contents.diag)
    assert has and nt % 64 != 0
    cfp = lib.hypre_amd_BoomerAMGGetCFMarker(s, 0)
    l1p = lib.hypre_amd_BoomerAMGGetL1Norms(s, 0)
    cf = C.cast(cfp, C.POINTER(B.IntArray)).contents.data if cfp else None
    l1 = C.cast(l1p, C.POINTER(B.Vector)).contents.data if l1p else None
    amg = oracle.amg_from_solvers([s])
    n = amg.A_levels[0].nrows
    f = rand_vector(n, 3)
    u0 = np.zeros(n) if zero else rand_vector(n, 4)
    du, df = B.parvec_from_numpy(u0), B.parvec_from_numpy(f)
    dv, dz = B.parvec_from_numpy(np.zeros(n)), B.parvec_from_numpy(np.zeros(n))
    if zero:
        lib.hypre_ParVectorSetZeros(du)
    uses_l1 = relax_type in (7, 18, 11, 12)
    err = lib.hypre_BoomerAMGRelax(Ap, df, cf, relax_type, relax_points, 0.9, 1.0, l1 if uses_l1 else None, du, dv, dz)
    B.check()
    assert err == 0
    u = B.parvec_to_numpy(du)
    ur = u0.copy()
    oracle.relax(amg.A_levels[0], f, amg.cf[0], relax_type, relax_points, 0.9, 1.0, amg.l1[0] if uses_l1 else None, ur,
                 all_zeros=zero)
    assert np.max(np.abs(u - ur)) <= 1e-12 * max(1.0, np.max(np.abs(ur)))
    lib.HYPRE_BoomerAMGDestroy(s)


def _one_cycle(lib, oracle, mixed=False, check_table=True, **kw):
    from hypre_amd import binding as B
    opt, A, s = _setup(lib, **kw)
    if mixed:
        lib.hypre_amd_BoomerAMGSetMixedPrecision(s, 1)
    Ap = C.cast(lib.hypre_amd_BoomerAMGGetA(s, 0), C.POINTER(B.ParCSRMatrix))
    has, nt, band = _plan_info(lib, Ap.contents.diag)
    if check_table:
        assert has and nt >= 2048 and nt % 64 != 0, (has, nt, band)
    amg = oracle.amg_from_solvers([s], mixed_precision=mixed)
    n = amg.A_levels[0].nrows
    f = np.ones(n)
    worst = 0.0
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
        amg.solve(f, ur, tol=0.0, max_iter=1, u_all_zeros=zero)      # = one cycle (mixed: in correction form)
        worst = max(worst, float(np.max(np.abs(u - ur)) / np.max(np.abs(ur))))
        lib.hypre_ParVectorDestroy(du)
        lib.hypre_ParVectorDestroy(df)
    oracle.drop_transposes()
    lib.HYPRE_BoomerAMGDestroy(s)
    lib.hypre_ParCSRMatrixDestroy(A)
    return worst


@pytest.mark.parametrize("kw", [
    dict(n=(88, 88, 88), relax_type=18),                       # 7-pt: 2 300 tiles, table only when forced
    dict(n=(80, 80, 80), problem="27pt", relax_type=11),       # 27-pt: 6 500 tiles on the operator, 3 250 on its triangle
    dict(n=(80, 80, 80), problem="27pt", relax_type=12),
    dict(n=(80, 80, 80), problem="27pt", relax_type=18, mixed=True),
    dict(n=(80, 80, 80), problem="27pt", relax_type=11, mixed=True),
])
def test_benchmark_class_cycle_matches_oracle_forced_tables(forced_tables, oracle, kw):
    """One V(1,1) cycle from a zero and from a non-zero guess at sizes whose fine level crosses 2048 tiles, every
    level large enough carrying a placement table."""
    kw = dict(kw)
    mixed = kw.pop("mixed", False)
    assert _one_cycle(forced_tables, oracle, mixed=mixed, coarsen_type=8, **kw) <= 1e-11


@pytest.mark.parametrize("kw", [
    dict(n=(80, 80, 80), problem="27pt", relax_type=11),
    dict(n=(80, 80, 80), problem="27pt", relax_type=18, mixed=True),
    dict(n=(100, 100, 100), problem="27pt", relax_type=12),    # the triangle crosses the 8-tiles-per-slab rule on its own
])
def test_benchmark_class_cycle_matches_oracle_default_policy(gpu_lib, oracle, kw):
    """The same with the policy the benchmark runs under (nothing forced): the 27-point operator of an 80^3 grid gets
    its table by itself."""
    kw = dict(kw)
    mixed = kw.pop("mixed", False)
    gpu_lib.hypre_amd_SpmvSetBandPolicy(1, 2048, 0)
    for variant in (2, 0):
        gpu_lib.hypre_amd_SpmvSetVariant(variant, 0)
        try:
            assert _one_cycle(gpu_lib, oracle, mixed=mixed, coarsen_type=8, **kw) <= 1e-11, variant
        finally:
            gpu_lib.hypre_amd_SpmvSetVariant(DEFAULT_VARIANT, 0)


@pytest.mark.parametrize("kw", [
    dict(relax_type=18), dict(relax_type=18, relax_order=1), dict(relax_type=11), dict(relax_type=12),
    dict(relax_type=7, relax_wt=0.8), dict(relax_type=18, cycle_type=2), dict(relax_type=16),
    dict(relax_type=11, problem="27pt"), dict(relax_type=18, problem="27pt", mixed=True),
])
def test_small_cycles_with_forced_tables(forced_tables, oracle, kw):
    """The configurations of test_amg_gpu.py::test_one_cycle_matches_oracle that run through the tiled kernel, on a
    24^3 grid with placement tables forced on every level of at least 8 tiles."""
    kw = dict(kw)
    mixed = kw.pop("mixed", False)
    assert _one_cycle(forced_tables, oracle, mixed=mixed, check_table=False, n=(24, 23, 22), coarsen_type=8, **kw) <= 1e-11


def test_solver_iterations_27pt_two_stage_gs_pcg(forced_tables, oracle):
    """Config C4's solver at a size the oracle solves in seconds: AMG(two-stage GS)-PCG on the 27-point operator,
    same iteration count and final residual as the oracle's PCG on the same hierarchy."""
    from hypre_amd import binding as B, ij
    lib = forced_tables
    opt, A, s = _setup(lib, n=(40, 40, 40), problem="27pt", relax_type=11, coarsen_type=8, solver=1)
    lib.HYPRE_BoomerAMGSetTol(s, 0.0)
    lib.HYPRE_BoomerAMGSetMaxIter(s, 1)
    amg = oracle.amg_from_solvers([s])
    b, x0 = ij.build_rhs_host(opt, A)
    dx, db = B.parvec_from_numpy(x0), B.parvec_from_numpy(b)
    pcg = C.c_void_p()
    lib.HYPRE_ParCSRPCGCreate(0, C.byref(pcg))
    lib.HYPRE_PCGSetTol(pcg, opt.tol)
    lib.HYPRE_PCGSetMaxIter(pcg, opt.max_iter)
    lib.HYPRE_PCGSetTwoNorm(pcg, 1)
    lib.HYPRE_PCGSetPrecond(pcg, C.cast(lib.HYPRE_BoomerAMGSolve, C.c_void_p), None, s)
    lib.HYPRE_ParCSRPCGSetup(pcg, A, db, dx)
    lib.HYPRE_ParCSRPCGSolve(pcg, A, db, dx)
    its, rel = C.c_int(), C.c_double()
    lib.HYPRE_PCGGetNumIterations(pcg, C.byref(its))
    lib.HYPRE_PCGGetFinalRelativeResidualNorm(pcg, C.byref(rel))
    B.check()
    xo = x0.copy()
    oits, orel, _ = amg.pcg(b, xo, tol=opt.tol, max_iter=opt.max_iter, two_norm=1)
    assert its.value == oits
    assert abs(rel.value - orel) <= 1e-6 * orel
    lib.HYPRE_ParCSRPCGDestroy(pcg)
    lib.HYPRE_BoomerAMGDestroy(s)
