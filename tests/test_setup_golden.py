"""CPU: the reference's own regression goldens (test/TEST_ij/*.saved) reproduced by
the library's host setup + the CPU oracle's solve phase.  This is what pins the
oracle: iteration counts, final residuals, convergence factors and complexities
printed by the reference driver for the same command lines."""
import ctypes as C
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, "golden", "ij_saved.json")))


def _run(lib, oracle, case):
    from hypre_amd import binding as B, ij
    opt = ij.IJOptions(**{k: (tuple(v) if isinstance(v, list) else v) for k, v in case["options"].items()})
    A = ij.build_matrix(opt)
    if opt.solver in (2, 4):
        # DS-PCG / DS-GMRES on a multivector (test/TEST_ij/vector.jobs): no hierarchy; the oracle's Krylov loop over all columns
        Ao = oracle.par_from_handles([A])
        b, x = ij.build_rhs_host(opt, A)
        nv = opt.num_components
        X = np.repeat(x[:, None], nv, axis=1)
        if opt.solver == 2:
            its, rel, conv = oracle.pcg_ds_multi(Ao, np.repeat(b[:, None], nv, axis=1), X, tol=opt.tol, max_iter=opt.max_iter,
                                                 two_norm=opt.two_norm)
        else:
            its, rel, conv = oracle.gmres_ds_multi(Ao, np.repeat(b[:, None], nv, axis=1), X, tol=opt.tol, max_iter=opt.max_iter,
                                                   k_dim=opt.k_dim)
        # every column is the single-vector solve (the same right-hand side in every component)
        assert np.max(np.abs(X - X[:, :1])) <= 1e-9 * np.max(np.abs(X))
        lib.hypre_ParCSRMatrixDestroy(A)
        return {"iterations": its, "rel_resid": rel}
    s = ij.create_amg(opt, memory_location=B.HYPRE_MEMORY_HOST)
    lib.HYPRE_BoomerAMGSetup(s, A, None, None)
    B.check()
    g, o = C.c_double(), C.c_double()
    lib.hypre_amd_BoomerAMGGetComplexities(s, C.byref(g), C.byref(o))
    amg = oracle.amg_from_solvers([s], num_threads=opt.num_threads)
    n = amg.A_levels[0].nrows
    b, x = ij.build_rhs_host(opt, A)
    if b is None:
        b = np.zeros(n)
        oracle.par_matvec(1.0, amg.A_levels[0], np.ones(n), 0.0, b, b)
    out = {"grid": g.value, "operator": o.value}
    if opt.solver == 0:
        its, rel, conv, hist = amg.solve(b, x, tol=opt.tol, max_iter=opt.mg_max_iter)
        out.update(iterations=its, rel_resid=rel, conv_factor=(hist[-1] / hist[0]) ** (1.0 / its))
    else:
        its, rel, conv = amg.pcg(b, x, tol=opt.tol, max_iter=opt.max_iter, two_norm=opt.two_norm,
                                 precond_cycles=opt.precon_cycles)
        out.update(iterations=its, rel_resid=rel)
    lib.HYPRE_BoomerAMGDestroy(s)
    return out


@pytest.mark.parametrize("name", sorted(k for k, v in GOLD.items() if v.get("ranks", 1) == 1))
def test_single_rank_goldens(lib, oracle, name):
    case = GOLD[name]
    out = _run(lib, oracle, case)
    exp = case["expect"]
    if "iterations" in exp:
        assert out["iterations"] == exp["iterations"]
    if "rel_resid" in exp:
        # the .saved files print 7 significant digits
        assert abs(out["rel_resid"] - exp["rel_resid"]) <= 5e-7 * exp["rel_resid"]
    for key in ("conv_factor", "grid", "operator"):
        if key in exp:
            assert abs(out[key] - exp[key]) < 5.1e-7, (key, out[key], exp[key])


def test_oracle_thread_count_does_not_change_bits(lib, oracle):
    """The timed CPU baseline runs the oracle's row loops under OpenMP: same bits as one thread."""
    from hypre_amd import binding as B, ij
    opt = ij.IJOptions(n=(14, 13, 12), coarsen_type=8, interp_type=6, P_max_elmts=4, relax_type=18, num_sweeps=1)
    A = ij.build_matrix(opt)
    s = ij.create_amg(opt, memory_location=B.HYPRE_MEMORY_HOST)
    lib.HYPRE_BoomerAMGSetup(s, A, None, None)
    B.check()
    amg = oracle.amg_from_solvers([s])
    n = amg.A_levels[0].nrows
    f = np.random.default_rng(3).uniform(-1, 1, n)
    outs = []
    for threads in (1, 3):
        oracle.set_num_threads(threads)
        u = np.zeros(n)
        amg.cycle(f, u, u_all_zeros=True)
        amg.cycle(f, u, u_all_zeros=False)
        outs.append(u)
    oracle.set_num_threads(1)
    oracle.drop_transposes()
    lib.HYPRE_BoomerAMGDestroy(s)
    assert np.array_equal(outs[0], outs[1])


def test_host_setup_does_not_depend_on_the_thread_count(lib):
    """The host setup runs its row loops on OpenMP threads above 100 000 rows — strength, the independent-set sweeps of
    PMIS (order-independent: a point's fate depends on the measures and on which neighbours are in the set), interpolation
    and the Galerkin product (per-row, insertion order kept).  One thread and eight threads give the same hierarchy,
    array for array."""
    import ctypes as C
    from hypre_amd import binding as B, ij
    hier = []
    for threads in (1, 8):
        lib.hypre_amd_SetHostThreads(threads)
        opt = ij.IJOptions(n=(64, 60, 56), coarsen_type=8, relax_type=18)
        A = ij.build_matrix(opt)
        s = ij.create_amg(opt, memory_location=B.HYPRE_MEMORY_HOST)
        lib.HYPRE_BoomerAMGSetup(s, A, None, None)
        B.check()
        nl = lib.hypre_amd_BoomerAMGGetNumLevels(s)
        lv = []
        for l in range(nl):
            Al = C.cast(lib.hypre_amd_BoomerAMGGetA(s, l), C.POINTER(B.ParCSRMatrix))
            lv.append(B.csr_to_arrays(Al.contents.diag))
            cfp = lib.hypre_amd_BoomerAMGGetCFMarker(s, l)
            if cfp:
                ia = C.cast(cfp, C.POINTER(B.IntArray)).contents
                lv.append((B.fetch(ia.data, ia.size, np.int32, ia.memory_location),))
        hier.append(lv)
        lib.HYPRE_BoomerAMGDestroy(s)
    lib.hypre_amd_SetHostThreads(lib.hypre_amd_HostCpuShare())
    assert len(hier[0]) == len(hier[1]) and len(hier[0]) > 6
    for m0, m1 in zip(hier[0], hier[1]):
        for a, b in zip(m0, m1):
            assert np.array_equal(a, b)
