"""CPU: the oracle's products against scipy (sanity of the restatement itself)."""
import numpy as np
import pytest

from util import laplace_3d, random_csr, rand_vector


@pytest.mark.parametrize("alpha,beta", [(1.0, 0.0), (-1.0, 1.0), (2.0, -2.0), (2.0, 2.0), (0.5, 0.25), (0.0, 3.0)])
def test_csr_matvec_matches_scipy(oracle, alpha, beta):
    A = random_csr(300, 200, 0, 9, seed=4, empty_frac=0.1)
    x, b = rand_vector(200, 1), rand_vector(300, 2)
    for with_rownnz in (False, True):
        y = np.zeros(300)
        assert oracle.csr_matvec(alpha, oracle.Csr.from_scipy(A, with_rownnz), x, beta, b, y) == 0
        assert np.allclose(y, alpha * (A @ x) + beta * b, rtol=1e-13, atol=1e-13)


def test_csr_matvec_alias_and_ierr(oracle):
    A = laplace_3d(4, 4, 4)
    oA = oracle.Csr.from_scipy(A)
    x = rand_vector(64, 1)
    y = x.copy()
    oracle.csr_matvec(1.0, oA, y, 0.0, y, y)          # x == y is deep-cloned
    assert np.allclose(y, A @ x)
    assert oracle.csr_matvec(1.0, oA, np.ones(70), 0.0, np.ones(70), np.zeros(70)) == 3
    assert oracle.csr_matvec(1.0, oA, np.ones(70), 0.0, np.ones(64), np.zeros(64)) == 1
    assert oracle.csr_matvec(1.0, oA, np.ones(64), 0.0, np.ones(70), np.zeros(70)) == 2


def test_csr_matvecT(oracle):
    A = random_csr(120, 80, 1, 5, seed=2)
    x, y0 = rand_vector(120, 3), rand_vector(80, 4)
    y = y0.copy()
    oracle.csr_matvecT(-1.5, oracle.Csr.from_scipy(A), x, 0.5, y)
    assert np.allclose(y, -1.5 * (A.T @ x) + 0.5 * y0, rtol=1e-13, atol=1e-13)


def test_gselim(oracle):
    import ctypes as C
    rng = np.random.default_rng(0)
    M = rng.uniform(-1, 1, (7, 7)) + 7 * np.eye(7)
    b = rng.uniform(-1, 1, 7)
    Mc, x = M.copy().ravel(), b.copy()
    L = oracle.load()
    assert L.oracle_gselim(Mc.ctypes.data_as(oracle.RealP), x.ctypes.data_as(oracle.RealP), 7) == 0
    assert np.allclose(x, np.linalg.solve(M, b), rtol=1e-12)


def test_host_transpose_large_matrix_is_the_stable_counting_sort(lib):
    """hypre_CSRMatrixTranspose on the host switches to row blocks counted in parallel above 2^20 entries; the
    result must be the sequential stable sort (rows of A^T list their entries by ascending row of A)."""
    import ctypes as C
    import scipy.sparse as sp
    from hypre_amd import binding as B
    rng = np.random.default_rng(11)
    nr, nc, per = 150000, 90000, 9
    cols = rng.integers(0, nc, size=(nr, per))
    cols.sort(axis=1)
    indptr = np.arange(0, nr * per + 1, per)
    A = sp.csr_matrix((rng.uniform(-1, 1, nr * per), cols.ravel(), indptr), shape=(nr, nc))   # duplicates kept
    A.has_canonical_format = False
    hA = B.csr_from_arrays(nr, nc, indptr, cols.ravel(), A.data, location=B.HYPRE_MEMORY_HOST)
    hT = C.POINTER(B.CSRMatrix)()
    lib.hypre_CSRMatrixTranspose(hA, C.byref(hT), 1)
    B.check()
    ti, tj, ta = B.csr_to_arrays(hT)
    # reference: stable argsort by column
    order = np.argsort(cols.ravel(), kind="stable")
    rows = np.repeat(np.arange(nr), per)
    assert np.array_equal(tj, rows[order].astype(np.int32))
    assert np.array_equal(ta, A.data[order])
    assert np.array_equal(ti, np.concatenate([[0], np.cumsum(np.bincount(cols.ravel(), minlength=nc))]).astype(np.int32))
    lib.hypre_CSRMatrixDestroy(hA)
    lib.hypre_CSRMatrixDestroy(hT)


def test_multicolor_sweep_is_hybrid_gauss_seidel_on_the_colour_permuted_system(oracle):
    """the reference has no multicolour Gauss-Seidel; the oracle's relax 21 / 22 are pinned to its
    golden-pinned hybrid Gauss-Seidel sweeps (relax 3 forward, 4 backward): on P A P^T, with P the permutation that sorts
    the rows by (colour, row), the sequential sweep visits the unknowns in the multicolour order."""
    import scipy.sparse as sp
    from util import laplace_3d, rand_vector
    for stencil, dims in ((7, (7, 6, 5)), (27, (6, 5, 4))):
        A = laplace_3d(*dims, stencil=stencil)
        n = A.shape[0]
        # greedy first-fit colouring in row order
        colors = -np.ones(n, dtype=np.int32)
        for i in range(n):
            taken = {colors[j] for j in A.indices[A.indptr[i]:A.indptr[i + 1]] if j != i and colors[j] >= 0}
            c = 0
            while c in taken:
                c += 1
            colors[i] = c
        assert colors.max() + 1 == (2 if stencil == 7 else 8)
        order = np.lexsort((np.arange(n), colors))            # rows by (colour, row)
        inv = np.empty(n, dtype=np.int64); inv[order] = np.arange(n)
        # P A P^T with the diagonal entry first in every row (the relaxation loops rely on it)
        rows, cols, vals = [], [], []
        indptr = [0]
        for pi in range(n):
            i = order[pi]
            js = A.indices[A.indptr[i]:A.indptr[i + 1]]; vs = A.data[A.indptr[i]:A.indptr[i + 1]]
            k = list(js).index(i)
            cols += [pi] + [inv[j] for q, j in enumerate(js) if q != k]
            vals += [vs[k]] + [v for q, v in enumerate(vs) if q != k]
            indptr.append(len(cols))
        Ap = sp.csr_matrix((np.array(vals), np.array(cols, dtype=np.int32), np.array(indptr, dtype=np.int32)), shape=(n, n))
        f, u0 = rand_vector(n, 3), rand_vector(n, 4)
        for rt_mc, rt_gs in ((21, 3), (22, 4)):
            u = u0.copy()
            assert oracle.relax(oracle.Par.from_scipy_single(A), f, None, rt_mc, 0, 1.0, 1.0, None, u, colors=colors) == 0
            up = u0[order].copy()
            assert oracle.relax(oracle.Par.from_scipy_single(Ap), f[order].copy(), None, rt_gs, 0, 1.0, 1.0, None, up) == 0
            assert np.max(np.abs(u[order] - up)) <= 1e-13 * np.max(np.abs(up)), (stencil, rt_mc)
