"""The BASELINE configurations at their full per-GPU sizes (256^3 unknowns: C2's 7-point operator, C4's 27-point
operator with two-stage Gauss-Seidel, C5's anisotropic operator with fp32 matrix values), checked through properties
that do not need a CPU run of the same size:

* y = A x against closed forms — x = 1 and x = i + 2j + 3k give integers, so every row is exact whatever the order of the
  sums; the expected value of a row follows from which of its stencil neighbours exist;
* symmetry of A (x.Ay = y.Ax), the transpose product against the product;
* the V-cycle from a zero guess is a LINEAR operator B (and, with symmetric smoothing, a symmetric one): B(a f1 + f2) =
  a B f1 + B f2, f1.B f2 = f2.B f1;
* AMG-PCG to 1e-8: the true residual, formed by a product of its own, is what the solver reports; the iteration count is
  the one the small-size oracle runs and the reference's benchmark file pin (22 / 15 / 16: bench.py);
* the hierarchy: 9 levels, grid complexity 1.354, operator complexity 2.767 (the reference's own
  run of this problem).

* one V-cycle against the CPU oracle's on the same hierarchy (every level copied to the host, the oracle's row loops on the
  host's cores): 1e-11 relative in the max norm — the full-size counterpart of the small-size cycle tests.

The same hierarchies are compared entry by entry with the CPU oracle at sizes the oracle finishes in seconds
(test_amg_gpu.py, test_bench_class_gpu.py), and bench.py compares one full-size cycle with the oracle on every run."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N = 256


def _coords(n):
    idx = np.arange(n ** 3, dtype=np.int64)
    return idx % n, (idx // n) % n, idx // (n * n)


def _closed_forms(n, kind):
    """(A 1, A (i + 2j + 3k)) of the 7-point (diagonal 6) or 27-point (diagonal 26) operator with -1 couplings on an
    n^3 grid whose boundary neighbours are simply absent."""
    i, j, k = _coords(n)
    lin = (i + 2 * j + 3 * k).astype(np.float64)
    if kind == "7pt":
        offs = [(-1, 0, 0), (1, 0, 0), (0, -1, 0), (0, 1, 0), (0, 0, -1), (0, 0, 1)]
        diag = 6.0
    else:
        offs = [(a, b, c) for a in (-1, 0, 1) for b in (-1, 0, 1) for c in (-1, 0, 1) if (a, b, c) != (0, 0, 0)]
        diag = 26.0
    ones = np.full(n ** 3, diag)
    y_lin = diag * lin
    for a, b, c in offs:
        there = (i + a >= 0) & (i + a < n) & (j + b >= 0) & (j + b < n) & (k + c >= 0) & (k + c < n)
        ones -= there
        y_lin -= np.where(there, lin + (a + 2 * b + 3 * c), 0.0)
    return ones, lin, y_lin


class Problem:
    def __init__(self, lib, **kw):
        from hypre_amd import binding as B, ij
        self.lib, self.B = lib, B
        self.opt = ij.IJOptions(n=(N, N, N), coarsen_type=8, interp_type=6, P_max_elmts=4, num_sweeps=1, **kw)
        self.A = ij.build_matrix(self.opt)
        lib.hypre_ParCSRMatrixMigrate(self.A, B.HYPRE_MEMORY_DEVICE)       # handed over in device memory, set up there
        self.n = N ** 3

    def setup(self, mixed=False):
        from hypre_amd import ij
        B, lib = self.B, self.lib
        self.s = ij.create_amg(self.opt, memory_location=B.HYPRE_MEMORY_DEVICE)
        if mixed:
            lib.hypre_amd_BoomerAMGSetMixedPrecision(self.s, 1)
        lib.HYPRE_BoomerAMGSetup(self.s, self.A, None, None)
        B.check()
        lib.HYPRE_BoomerAMGSetTol(self.s, 0.0)
        lib.HYPRE_BoomerAMGSetMaxIter(self.s, 1)
        return self

    def vec(self, x):
        return self.B.parvec_from_numpy(np.ascontiguousarray(x, dtype=np.float64))

    def matvec(self, x, transpose=False):
        dx, dy = self.vec(x), self.vec(np.zeros(self.n))
        (self.lib.hypre_ParCSRMatrixMatvecT if transpose else self.lib.hypre_ParCSRMatrixMatvec)(1.0, self.A, dx, 0.0, dy)
        self.B.check()
        y = self.B.parvec_to_numpy(dy)
        self.lib.hypre_ParVectorDestroy(dx); self.lib.hypre_ParVectorDestroy(dy)
        return y

    def matvec_columns(self, X):
        """Y = A X for the columns of X handed over as one multivector (stored column by column)."""
        B, lib = self.B, self.lib
        n, nv = X.shape

        def mv(M):
            pv = B.parvec_from_numpy(np.ascontiguousarray(M.T).ravel(), global_size=n)
            pv.contents.partitioning[1] = n
            pv.contents.last_index = n - 1
            pv.contents.actual_local_size = n
            v = pv.contents.local_vector.contents
            v.size, v.num_vectors, v.vecstride, v.idxstride = n, nv, n, 1
            return pv
        dx, dy = mv(X), mv(np.zeros((n, nv)))
        before = lib.hypre_amd_SpmvFusedMultivectorLaunches()
        lib.hypre_ParCSRMatrixMatvec(1.0, self.A, dx, 0.0, dy)
        B.check()
        assert lib.hypre_amd_SpmvFusedMultivectorLaunches() > before          # the fused kernel served it
        v = dy.contents.local_vector.contents
        Y = B.fetch(v.data, n * nv, np.float64, v.memory_location).reshape(nv, n).T
        lib.hypre_ParVectorDestroy(dx); lib.hypre_ParVectorDestroy(dy)
        return Y

    def cycle(self, f):
        """u = B f: one V-cycle from a zero guess, as a preconditioner call."""
        df, du = self.vec(f), self.vec(np.zeros(self.n))
        self.lib.hypre_ParVectorSetZeros(du)
        self.lib.HYPRE_BoomerAMGSolve(self.s, self.A, df, du)
        self.B.check()
        u = self.B.parvec_to_numpy(du)
        self.lib.hypre_ParVectorDestroy(df); self.lib.hypre_ParVectorDestroy(du)
        return u

    def pcg(self, b, tol=1e-8):
        lib, B = self.lib, self.B
        db, du = self.vec(b), self.vec(np.zeros(self.n))
        pcg = C.c_void_p()
        lib.HYPRE_ParCSRPCGCreate(0, C.byref(pcg))
        lib.HYPRE_PCGSetTol(pcg, tol)
        lib.HYPRE_PCGSetMaxIter(pcg, 100)
        lib.HYPRE_PCGSetTwoNorm(pcg, 1)
        lib.HYPRE_PCGSetPrecond(pcg, C.cast(lib.HYPRE_BoomerAMGSolve, C.c_void_p), None, self.s)
        lib.hypre_ParVectorSetZeros(du)
        lib.HYPRE_ParCSRPCGSetup(pcg, self.A, db, du)
        lib.HYPRE_ParCSRPCGSolve(pcg, self.A, db, du)
        its, rel = C.c_int(), C.c_double()
        lib.HYPRE_PCGGetNumIterations(pcg, C.byref(its))
        lib.HYPRE_PCGGetFinalRelativeResidualNorm(pcg, C.byref(rel))
        lib.HYPRE_ParCSRPCGDestroy(pcg)
        B.check()
        x = B.parvec_to_numpy(du)
        lib.hypre_ParVectorDestroy(db); lib.hypre_ParVectorDestroy(du)
        return x, its.value, rel.value

    def close(self):
        if getattr(self, "s", None):
            self.lib.HYPRE_BoomerAMGDestroy(self.s)
        self.lib.hypre_ParCSRMatrixDestroy(self.A)


def _rand(n, seed):
    return np.random.default_rng(seed).uniform(-1.0, 1.0, n)


def _check_operator(p, kind):
    ones, lin, y_lin = _closed_forms(N, kind)
    assert np.array_equal(p.matvec(np.ones(p.n)), ones)
    assert np.array_equal(p.matvec(lin), y_lin)
    assert np.array_equal(p.matvec(lin, transpose=True), y_lin)          # A is symmetric: the stored transpose is A
    x, y = _rand(p.n, 1), _rand(p.n, 2)
    Ax, Ay = p.matvec(x), p.matvec(y)
    scale = np.linalg.norm(x) * np.linalg.norm(Ay)
    assert abs(np.dot(x, Ay) - np.dot(y, Ax)) <= 1e-13 * scale
    assert np.max(np.abs(p.matvec(x, transpose=True) - Ax)) <= 1e-13 * np.max(np.abs(Ax))
    # the four vectors as ONE multivector (one pass over the matrix for its columns): the closed forms again, and the bits
    # of the single-vector products
    Y = p.matvec_columns(np.stack([np.ones(p.n), lin, x, y], axis=1))
    assert np.array_equal(Y[:, 0], ones) and np.array_equal(Y[:, 1], y_lin)
    assert np.array_equal(Y[:, 2].view(np.int64), Ax.view(np.int64)) and np.array_equal(Y[:, 3].view(np.int64), Ay.view(np.int64))
    return x, y


def _check_cycle_is_linear(p, f1, f2, tol, symmetric):
    a = 0.75                                             # a power of two times three: scaling f1 by it rounds, on purpose
    B1, B2, B12 = p.cycle(f1), p.cycle(f2), p.cycle(a * f1 + f2)
    ref = a * B1 + B2
    assert np.max(np.abs(B12 - ref)) <= tol * np.max(np.abs(ref))
    if symmetric:
        lhs, rhs = np.dot(f1, B2), np.dot(f2, B1)
        assert abs(lhs - rhs) <= 100 * tol * np.linalg.norm(f1) * np.linalg.norm(B2)
    # and it does reduce the error: ||f - A B f|| < ||f|| by the factor a V(1,1) cycle is good for
    r = f1 - p.matvec(B1)
    assert np.linalg.norm(r) <= 0.5 * np.linalg.norm(f1)


def _check_cycle_against_the_oracle(p, oracle, f, mixed=False):
    """one V-cycle from a zero guess on the device against the CPU oracle's on the same hierarchy — every level copied to the
    host, the oracle's row loops on the host's cores (the bits do not depend on the thread count): the full-size counterpart
    of the small-size cycle tests, 1e-11 relative in the max norm"""
    u = p.cycle(f)
    amg = oracle.amg_from_solvers([p.s], mixed_precision=mixed)
    import os
    oracle.set_num_threads(max(1, min(16, len(os.sched_getaffinity(0)))))
    try:
        ur = np.zeros(p.n)
        amg.cycle(f, ur, u_all_zeros=True)
    finally:
        oracle.set_num_threads(1)
        oracle.drop_transposes()
    assert np.max(np.abs(u - ur)) <= 1e-11 * np.max(np.abs(ur)), np.max(np.abs(u - ur)) / np.max(np.abs(ur))


def _levels_and_complexities(p):
    lib = p.lib
    nl = lib.hypre_amd_BoomerAMGGetNumLevels(p.s)
    rows = nnz = 0
    for l in range(nl):
        Al = C.cast(lib.hypre_amd_BoomerAMGGetA(p.s, l), C.POINTER(p.B.ParCSRMatrix)).contents
        rows += Al.diag.contents.num_rows
        nnz += Al.diag.contents.num_nonzeros
    A0 = p.A.contents.diag.contents
    return nl, rows / A0.num_rows, nnz / A0.num_nonzeros


def test_c2_7pt_l1_jacobi_at_full_size(gpu_lib, oracle):
    """Config C2: 256^3 7-point Laplacian, PMIS / ext+i(4) / l1-Jacobi V(1,1), set up on the device."""
    p = Problem(gpu_lib, relax_type=18)
    try:
        x, y = _check_operator(p, "7pt")
        p.setup()
        nl, gc, oc = _levels_and_complexities(p)
        assert nl == 9 and abs(gc - 1.354) < 2e-3 and abs(oc - 2.767) < 2e-3, (nl, gc, oc)
        _check_cycle_is_linear(p, x, y, 1e-12, symmetric=True)
        _check_cycle_against_the_oracle(p, oracle, x)
        b = np.ones(p.n)
        sol, its, rel = p.pcg(b)
        assert its == 22 and rel <= 1e-8, (its, rel)
        true_rel = np.linalg.norm(b - p.matvec(sol)) / np.linalg.norm(b)
        assert abs(true_rel - rel) <= 1e-2 * rel + 1e-13, (true_rel, rel)
    finally:
        p.close()


def test_c4_27pt_two_stage_gs_at_full_size(gpu_lib, oracle):
    """Config C4's per-GPU share: 256^3 27-point operator (453 M entries), two-stage Gauss-Seidel (relax 11), whose
    accumulating epilogue is the one that must see every tile exactly once."""
    p = Problem(gpu_lib, relax_type=11, problem="27pt")
    try:
        x, y = _check_operator(p, "27pt")
        p.setup()
        _check_cycle_is_linear(p, x, y, 1e-12, symmetric=False)       # forward sweeps down and up: not a symmetric cycle
        _check_cycle_against_the_oracle(p, oracle, x)
        b = np.ones(p.n)
        sol, its, rel = p.pcg(b)
        assert its <= 20 and rel <= 1e-8, (its, rel)
        true_rel = np.linalg.norm(b - p.matvec(sol)) / np.linalg.norm(b)
        assert abs(true_rel - rel) <= 1e-2 * rel + 1e-13, (true_rel, rel)
    finally:
        p.close()


def test_c5_anisotropic_mixed_precision_at_full_size(gpu_lib, oracle):
    """Config C5's per-GPU share: 256^3 anisotropic diffusion (1, 1, 0.001), matrix values streamed as fp32 inside the
    cycle, residuals and corrections in fp64: the cycle is still linear (to fp64 accuracy: rounding the matrix once does
    not depend on the right-hand side), and PCG reaches 1e-8 on the TRUE (fp64) residual."""
    p = Problem(gpu_lib, relax_type=18, problem="difconv", c=(1.0, 1.0, 0.001), a=(0.0, 0.0, 0.0))
    try:
        x, y = _rand(p.n, 1), _rand(p.n, 2)
        Ax, Ay = p.matvec(x), p.matvec(y)
        assert abs(np.dot(x, Ay) - np.dot(y, Ax)) <= 1e-13 * np.linalg.norm(x) * np.linalg.norm(Ay)
        p.setup(mixed=True)
        _check_cycle_is_linear(p, x, y, 1e-12, symmetric=True)
        _check_cycle_against_the_oracle(p, oracle, x, mixed=True)
        b = np.ones(p.n)
        sol, its, rel = p.pcg(b)
        assert its <= 20 and rel <= 1e-8, (its, rel)
        true_rel = np.linalg.norm(b - p.matvec(sol)) / np.linalg.norm(b)
        assert abs(true_rel - rel) <= 1e-2 * rel + 1e-13, (true_rel, rel)
    finally:
        p.close()
