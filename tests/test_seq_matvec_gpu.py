"""Parity of the HIP sequential SpMV (hypre_CSRMatrixMatvec* through the C ABI)
against the CPU oracle, on seeded inputs.  fp64; tolerance: the GPU sums each
row in a different association than the reference's serial loop, so results
agree to a few ulps of the row's absolute sum: |y_gpu - y_ref| <= 1e-13 * (|alpha| |A| |x| + |beta b|)."""
import numpy as np
import pytest
import scipy.sparse as sp

from util import banded_csr, laplace_3d, random_csr, rand_vector

pytestmark = pytest.mark.gpu


def _bound(A, x, alpha, beta, b):
    return 1e-13 * (abs(alpha) * (abs(A) @ np.abs(x)) + abs(beta) * np.abs(b)) + 1e-300


def _run(lib, oracle, A, alpha, beta, seed=0, inplace=False):
    from hypre_amd import binding as B
    x = rand_vector(A.shape[1], seed + 1)
    b = rand_vector(A.shape[0], seed + 2)
    y0 = rand_vector(A.shape[0], seed + 3)
    dA = B.csr_from_scipy(A)
    dx, db = B.vec_from_numpy(x), B.vec_from_numpy(b)
    dy = B.vec_from_numpy(b if inplace else y0)
    if inplace:
        ierr = lib.hypre_CSRMatrixMatvec(alpha, dA, dx, beta, dy)
    else:
        ierr = lib.hypre_CSRMatrixMatvecOutOfPlace(alpha, dA, dx, beta, db, dy, 0)
    B.check()
    y = B.vec_to_numpy(dy)
    oA = oracle.Csr.from_scipy(A)
    yr = (b if inplace else y0).copy()
    ierr_ref = oracle.csr_matvec(alpha, oA, x, beta, b, yr)
    assert ierr == ierr_ref == 0
    assert np.all(np.abs(y - yr) <= _bound(A, x, alpha, beta, b)), np.abs(y - yr).max()
    for o in (dx, db, dy):
        lib.hypre_SeqVectorDestroy(o)
    lib.hypre_CSRMatrixDestroy(dA)


@pytest.mark.parametrize("alpha,beta", [(1.0, 0.0), (-1.0, 1.0), (1.0, 1.0), (1.0, -1.0), (-1.0, -1.0),
                                        (2.5, 0.0), (2.5, -2.5), (2.5, 2.5), (0.7, 0.3), (-1.0, 0.4),
                                        (1.0, 0.4), (0.0, 0.5), (-1.0, 0.0)])
def test_laplacian_all_branches(gpu_lib, oracle, alpha, beta):
    _run(gpu_lib, oracle, laplace_3d(8, 8, 8), alpha, beta)


def test_27pt_and_inplace(gpu_lib, oracle):
    _run(gpu_lib, oracle, laplace_3d(6, 6, 6, 27), 1.0, 0.0)
    _run(gpu_lib, oracle, laplace_3d(6, 6, 6, 27), -1.0, 1.0, inplace=True)


@pytest.mark.parametrize("lo,hi", [(0, 4), (1, 12), (13, 48), (49, 160), (200, 900)])
def test_row_length_bins(gpu_lib, oracle, lo, hi):
    """Every reduction width of the tiled kernel (1, 8, 32 lanes per row) plus ragged tiles."""
    A = random_csr(3000, 2500, lo, hi, seed=hi)
    _run(gpu_lib, oracle, A, 1.0, 0.0, seed=hi)
    _run(gpu_lib, oracle, A, -0.5, 2.0, seed=hi + 1)


def test_rectangular_with_empty_rows(gpu_lib, oracle):
    A = random_csr(5000, 700, 0, 4, seed=3, empty_frac=0.4)
    _run(gpu_lib, oracle, A, 1.0, 1.0, seed=5)


def test_long_rows_fall_back_to_wave_kernel(gpu_lib, oracle):
    A = random_csr(40, 5000, 1500, 3000, seed=9)
    _run(gpu_lib, oracle, A, 1.0, 0.0, seed=7)
    _run(gpu_lib, oracle, A, 1.0, -1.0, seed=8)


def test_empty_matrix_and_zero_rows(gpu_lib, oracle):
    A = sp.csr_matrix((50, 30))
    _run(gpu_lib, oracle, A, 1.0, 0.5)


def test_rownnz_path(gpu_lib, oracle):
    """Off-diagonal-like block: few non-empty rows, rownnz list set -> sparse-row kernel."""
    from hypre_amd import binding as B
    A = random_csr(4000, 300, 1, 3, seed=11, empty_frac=0.95)
    x = rand_vector(300, 1); y0 = rand_vector(4000, 2)
    dA = B.csr_from_scipy(A)
    gpu_lib.hypre_CSRMatrixSetRownnz(dA)
    assert dA.contents.num_rownnz < 0.7 * 4000 and bool(dA.contents.rownnz)
    dx, dy = B.vec_from_numpy(x), B.vec_from_numpy(y0)
    gpu_lib.hypre_CSRMatrixMatvec(-1.0, dA, dx, 1.0, dy)
    B.check()
    y = B.vec_to_numpy(dy)
    oA = oracle.Csr.from_scipy(A, with_rownnz=True)
    yr = y0.copy()
    oracle.csr_matvec(-1.0, oA, x, 1.0, y0.copy(), yr)
    assert np.all(np.abs(y - yr) <= _bound(A, x, -1.0, 1.0, y0))


def test_transpose(gpu_lib, oracle):
    from hypre_amd import binding as B
    A = random_csr(900, 400, 1, 6, seed=21)
    x = rand_vector(900, 1); y0 = rand_vector(400, 2)
    dA = B.csr_from_scipy(A)
    dx, dy = B.vec_from_numpy(x), B.vec_from_numpy(y0)
    ierr = gpu_lib.hypre_CSRMatrixMatvecT(2.0, dA, dx, -1.0, dy)
    B.check()
    y = B.vec_to_numpy(dy)
    yr = y0.copy()
    assert oracle.csr_matvecT(2.0, oracle.Csr.from_scipy(A), x, -1.0, yr) == ierr == 0
    assert np.all(np.abs(y - yr) <= 1e-13 * (2.0 * (abs(A).T @ np.abs(x)) + np.abs(y0)) + 1e-300)


def test_size_mismatch_is_informational(gpu_lib, oracle):
    """ierr 1/2/3 is returned and the product is still formed (csr_matvec.c:57-86)."""
    from hypre_amd import binding as B
    A = laplace_3d(4, 4, 4)
    dA = B.csr_from_scipy(A)
    dx = B.vec_from_numpy(np.ones(70)); dy = B.vec_from_numpy(np.zeros(70))
    assert gpu_lib.hypre_CSRMatrixMatvec(1.0, dA, dx, 0.0, dy) == 3
    B.check()
    y = B.vec_to_numpy(dy)[:64]
    assert np.allclose(y, A @ np.ones(64))


def test_host_operands_fail_loudly(gpu_lib):
    from hypre_amd import binding as B
    A = laplace_3d(4, 4, 4)
    hA = B.csr_from_scipy(A, B.HYPRE_MEMORY_HOST)
    hx = B.vec_from_numpy(np.ones(64), B.HYPRE_MEMORY_HOST)
    hy = B.vec_from_numpy(np.zeros(64), B.HYPRE_MEMORY_HOST)
    gpu_lib.hypre_CSRMatrixMatvec(1.0, hA, hx, 0.0, hy)
    with pytest.raises(B.HypreAmdError, match="host execution is not part of this library"):
        B.check()


def test_blas1(gpu_lib, oracle):
    from hypre_amd import binding as B
    n = 100003
    x, y = rand_vector(n, 1), rand_vector(n, 2)
    dx, dy = B.vec_from_numpy(x), B.vec_from_numpy(y)
    dot = gpu_lib.hypre_SeqVectorInnerProd(dx, dy)
    assert abs(dot - float(np.dot(x, y))) <= 1e-12 * float(np.dot(np.abs(x), np.abs(y)))
    gpu_lib.hypre_SeqVectorAxpy(0.3, dx, dy)
    assert np.allclose(B.vec_to_numpy(dy), y + 0.3 * x, rtol=0, atol=1e-15)
    gpu_lib.hypre_SeqVectorScale(-2.0, dy)
    assert np.allclose(B.vec_to_numpy(dy), -2.0 * (y + 0.3 * x), rtol=0, atol=1e-15)
    dz = B.vec_from_numpy(np.zeros(n))
    gpu_lib.hypre_SeqVectorAxpyz(2.0, dx, -1.0, dy, dz)
    assert np.allclose(B.vec_to_numpy(dz), 2.0 * x + 2.0 * (y + 0.3 * x), rtol=0, atol=1e-14)
    d = np.abs(rand_vector(n, 3)) + 0.5
    dd = B.vec_from_numpy(d)
    ybefore = B.vec_to_numpy(dy)
    gpu_lib.hypre_SeqVectorElmdivpy(dx, dd, dy)
    assert np.allclose(B.vec_to_numpy(dy), ybefore + x / d, rtol=0, atol=1e-14)
    B.check()


def _multivector(B, X, par=False):
    """Column-major device multivector (seq_mv/vector.h:22-40: vecstride = size, idxstride = 1) holding the columns
    of X; built as one long vector whose header is then re-shaped."""
    n, nv = X.shape
    flat = np.ascontiguousarray(X.T).ravel()
    if par:
        pv = B.parvec_from_numpy(flat, global_size=n)
        pv.contents.partitioning[1] = n
        pv.contents.last_index = n - 1
        pv.contents.actual_local_size = n
        v = pv.contents.local_vector
    else:
        pv = v = B.vec_from_numpy(flat)
    v.contents.size, v.contents.num_vectors, v.contents.vecstride, v.contents.idxstride = n, nv, n, 1
    return pv, v


def _columns(B, v):
    s = v.contents
    flat = B.fetch(s.data, s.size * s.num_vectors, np.float64, s.memory_location)
    return flat.reshape(s.num_vectors, s.size).T


@pytest.mark.parametrize("par", [False, True])
def test_multivectors_column_by_column(gpu_lib, oracle, par):
    """csr_matvec.c:117-380 / par_csr_matvec.c:146-165: NV = 3 right-hand sides in one call, y = alpha A X + beta B and
    Y = alpha A^T X + beta Y, each column equal to the single-vector product."""
    from hypre_amd import binding as B
    lib = gpu_lib
    A = laplace_3d(7, 6, 5)
    n, nv = A.shape[0], 3
    X = np.stack([rand_vector(n, 10 + k) for k in range(nv)], axis=1)
    Bm = np.stack([rand_vector(n, 20 + k) for k in range(nv)], axis=1)
    px, vx = _multivector(B, X, par)
    pb, vb = _multivector(B, Bm, par)
    py, vy = _multivector(B, np.zeros((n, nv)), par)
    if par:
        ii, jj, aa = A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data
        lapl = B.laplacian(7, 6, 5)
        lib.hypre_ParCSRMatrixMigrate(lapl, B.HYPRE_MEMORY_DEVICE)
        lib.hypre_ParCSRMatrixMatvecOutOfPlace(0.7, lapl, px, -1.3, pb, py)
    else:
        dA = B.csr_from_scipy(A)
        lib.hypre_CSRMatrixMatvecOutOfPlace(0.7, dA, vx, -1.3, vb, vy, 0)
    B.check()
    Y = _columns(B, vy)
    oA = oracle.Csr.from_scipy(A)
    for k in range(nv):
        yr = np.zeros(n)
        oracle.csr_matvec(0.7, oA, X[:, k].copy(), -1.3, Bm[:, k].copy(), yr)
        assert np.all(np.abs(Y[:, k] - yr) <= _bound(A, X[:, k], 0.7, -1.3, Bm[:, k]))
    # transpose product, in place
    if par:
        lib.hypre_ParCSRMatrixMatvecT(-0.4, lapl, px, 0.5, pb)
    else:
        lib.hypre_CSRMatrixMatvecT(-0.4, dA, vx, 0.5, vb)
    B.check()
    Z = _columns(B, vb)
    for k in range(nv):
        zr = Bm[:, k].copy()
        oracle.csr_matvecT(-0.4, oA, X[:, k].copy(), 0.5, zr)
        assert np.all(np.abs(Z[:, k] - zr) <= _bound(A.T.tocsr(), X[:, k], -0.4, 0.5, Bm[:, k]))


def _mv_matrix(what):
    if what == "7pt":                       # coded (two values), rows of 4-7 entries: a lane per row
        return laplace_3d(24, 22, 20), True
    if what == "27pt":                      # coded, up to 27 entries per row: several lanes per row
        return laplace_3d(16, 15, 14, stencil=27), True
    if what == "coded_unequal":             # five values, rows of 3-40 entries: coded tiles (no slice form)
        A = banded_csr(20000, 20000, 3, 40, 600, seed=8)
        A.data[:] = np.array([6.0, -1.0, 0.25, -0.0, 3e-300])[np.random.default_rng(9).integers(0, 5, A.nnz)]
        return A, True
    if what == "banded_short":              # all values distinct, 3-11 entries per row
        return banded_csr(30000, 30000, 3, 11, 700, seed=3, empty_frac=0.02), True
    if what == "banded_long":               # 20-60 per row: 2 to 8 lanes per row, tile by tile
        return banded_csr(9000, 9000, 20, 60, 500, seed=4), True
    if what == "banded_rect":               # rectangular, rows up to the 256 the fused kernel takes, spills past the window
        return banded_csr(3000, 5000, 100, 256, 900, seed=5), True
    if what == "rows_too_long":             # rows of up to 400 entries: column by column
        return banded_csr(3000, 5000, 100, 400, 900, seed=6), False
    if what == "scattered":                 # columns all over a wide x: most tiles cannot be staged, column by column
        return random_csr(4000, 400000, 10, 40, seed=7), False
    if what == "odd_rows":                  # an odd number of rows: the second column of y and b is not 16-byte aligned (fine),
        return laplace_3d(9, 7, 5), True    # of x as well (315 columns): column by column
    raise ValueError(what)


@pytest.mark.parametrize("what", ["7pt", "27pt", "coded_unequal", "banded_short", "banded_long", "banded_rect", "rows_too_long", "scattered", "odd_rows"])
@pytest.mark.parametrize("nv,alpha,beta", [(2, 1.0, 0.0), (3, 0.7, -1.3), (4, -1.0, 1.0), (7, 2.5, 0.5)])
def test_fused_multivector_products_have_the_bits_of_the_column_loop(gpu_lib, oracle, what, nv, alpha, beta):
    """One pass over the matrix for up to four columns at a time (spmv_xs_mv_kernel; reference: csr_matvec.c:117-380,
    csr_spmv_device.c:37-134) against one pass per column: the same bits in every column, for coded and fp64 matrices, a
    lane or several lanes per row, spilling rows, 2 / 3 / 4 columns and 7 = 4 + 3; operands the fused kernel does not take
    (rows over 256 entries, unstaged tiles, columns of x that are not 16-byte aligned) fall back to the loop; and every
    column is the oracle's single-vector product."""
    from hypre_amd import binding as B
    lib = gpu_lib
    A, fused_expected = _mv_matrix(what)
    if what == "odd_rows":
        fused_expected = False
    n, m = A.shape
    X = np.stack([rand_vector(m, 10 + k) for k in range(nv)], axis=1)
    Bm = np.stack([rand_vector(n, 20 + k) for k in range(nv)], axis=1)
    dA = B.csr_from_scipy(A)
    out = {}
    try:
        for on in (1, 0):
            lib.hypre_amd_SpmvSetFusedMultivectors(on)
            _, vx = _multivector(B, X)
            _, vb = _multivector(B, Bm)
            _, vy = _multivector(B, np.full((n, nv), 7.0))
            before = lib.hypre_amd_SpmvFusedMultivectorLaunches()
            lib.hypre_CSRMatrixMatvecOutOfPlace(alpha, dA, vx, beta, vb, vy, 0)
            B.check()
            launches = lib.hypre_amd_SpmvFusedMultivectorLaunches() - before
            # passes: four columns at a time over the tiles; threes and twos over the slice form (plan form 4)
            passes = {2: 1, 3: 1, 4: 2, 7: 3}[nv] if lib.hypre_amd_CSRMatrixPlanForm(dA) == 4 else (nv + 3) // 4
            assert launches == (passes if (on and fused_expected) else 0), (what, on, launches)
            out[on] = _columns(B, vy)
            # in place: Y = alpha A X + beta Y
            lib.hypre_CSRMatrixMatvec(alpha, dA, vx, beta, vb)
            B.check()
            out[on, "inplace"] = _columns(B, vb)
            if n == m:
                # x is y (the host routine clones x: csr_matvec.c:109-113): X <- alpha A X + beta X
                lib.hypre_CSRMatrixMatvec(alpha, dA, vx, beta, vx)
                B.check()
                out[on, "aliased"] = _columns(B, vx)
            for o in (vx, vb, vy):
                lib.hypre_SeqVectorDestroy(o)
    finally:
        lib.hypre_amd_SpmvSetFusedMultivectors(1)
    lib.hypre_CSRMatrixDestroy(dA)
    assert np.array_equal(out[1].view(np.int64), out[0].view(np.int64))
    assert np.array_equal(out[1, "inplace"].view(np.int64), out[0, "inplace"].view(np.int64))
    assert np.array_equal(out[1].view(np.int64), out[1, "inplace"].view(np.int64))
    if n == m:
        assert np.array_equal(out[1, "aliased"].view(np.int64), out[0, "aliased"].view(np.int64))
        ref = alpha * (A @ X) + beta * X
        assert np.all(np.abs(out[1, "aliased"] - ref) <= np.stack([_bound(A, X[:, k], alpha, beta, X[:, k]) for k in range(nv)], axis=1))
    oA = oracle.Csr.from_scipy(A)
    for k in range(nv):
        yr = np.zeros(n)
        oracle.csr_matvec(alpha, oA, X[:, k].copy(), beta, Bm[:, k].copy(), yr)
        assert np.all(np.abs(out[1][:, k] - yr) <= _bound(A, X[:, k], alpha, beta, Bm[:, k]))


@pytest.mark.parametrize("total", [2049, 2050, 2051, 2052, 2053, 4099, 2048 + 255, 2048 + 256, 2048 + 257, 2048 + 600])
def test_last_row_spills_past_the_streamed_window(gpu_lib, oracle, total):
    """The tiled kernel streams 2048 entries per tile and fetches what the tile's last row has beyond that window
    separately (one entry per lane, then a strided loop): the spill may end in the matrix's last, partial 16-byte quad
    and may be longer than the 256 lanes."""
    rng = np.random.default_rng(total)
    n_short = 340
    lens = [6] * n_short                     # 2040 entries in short rows, then one long row to `total`
    lens.append(total - 6 * n_short)
    ncols = 5000
    indptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    cols = np.concatenate([np.sort(rng.choice(ncols, size=l, replace=False)) for l in lens]).astype(np.int32)
    vals = rng.uniform(-1.0, 1.0, size=indptr[-1])
    A = sp.csr_matrix((vals, cols, indptr), shape=(len(lens), ncols))
    _run(gpu_lib, oracle, A, 1.0, 0.0, seed=total)
    _run(gpu_lib, oracle, A, -0.7, 1.3, seed=total + 1)


@pytest.mark.parametrize("shape,lo,hi,empty", [((900, 400), 1, 6, 0.0), ((5000, 700), 0, 4, 0.4), ((300, 3000), 20, 90, 0.0),
                                               ((64, 64), 1, 3, 0.2), ((40, 5000), 1500, 3000, 0.0)])
def test_device_transpose_is_the_host_transpose(gpu_lib, shape, lo, hi, empty):
    """hypre_CSRMatrixTranspose of a device matrix runs on the device (count, scan, scatter, order: kernels.hip) and gives
    what the host routine gives (seq_mv/csr_matop.c:1043-1270): row c of A^T lists the rows of A holding column c in
    ascending order, values alongside — array for array."""
    import ctypes as C
    from hypre_amd import binding as B
    lib = gpu_lib
    A = random_csr(shape[0], shape[1], lo, hi, seed=shape[0] + hi, empty_frac=empty)
    dA, hA = B.csr_from_scipy(A), B.csr_from_scipy(A, B.HYPRE_MEMORY_HOST)
    dT, hT = C.POINTER(B.CSRMatrix)(), C.POINTER(B.CSRMatrix)()
    lib.hypre_CSRMatrixTranspose(dA, C.byref(dT), 1)
    lib.hypre_CSRMatrixTranspose(hA, C.byref(hT), 1)
    B.check()
    assert dT.contents.memory_location == B.HYPRE_MEMORY_DEVICE and hT.contents.memory_location == B.HYPRE_MEMORY_HOST
    for a, b in zip(B.csr_to_arrays(dT), B.csr_to_arrays(hT)):
        assert np.array_equal(a, b)
    # pattern only
    dP = C.POINTER(B.CSRMatrix)()
    lib.hypre_CSRMatrixTranspose(dA, C.byref(dP), 0)
    B.check()
    ii, jj, _ = B.csr_to_arrays(dP)
    ri, rj, _ = B.csr_to_arrays(hT)
    assert np.array_equal(ii, ri) and np.array_equal(jj, rj)
    for m in (dA, hA, dT, hT, dP):
        lib.hypre_CSRMatrixDestroy(m)


def test_comm_pkg_update_vec_starts(gpu_lib):
    """par_csr_communication.c:1054-1154 on a hand-made package: 1 -> 3 components and back."""
    import ctypes as C
    from hypre_amd import binding as B
    lib = gpu_lib
    pkg = B.CommPkg()
    starts = np.array([0, 2, 5], dtype=np.int32)
    elmts = np.array([4, 7, 1, 2, 9], dtype=np.int32)
    rstarts = np.array([0, 3, 4], dtype=np.int32)

    def host_copy(a):
        p = lib.hypre_CAlloc(len(a), 4, B.HYPRE_MEMORY_HOST)
        C.memmove(p, a.ctypes.data, a.nbytes)
        return C.cast(p, C.POINTER(C.c_int))

    pkg.comm, pkg.num_components, pkg.num_sends, pkg.num_recvs = 0, 1, 2, 2
    pkg.send_map_starts, pkg.send_map_elmts, pkg.recv_vec_starts = host_copy(starts), host_copy(elmts), host_copy(rstarts)
    lib.hypre_ParCSRCommPkgUpdateVecStarts(C.byref(pkg), 3, 100, 1)
    B.check()
    assert pkg.num_components == 3
    assert [pkg.send_map_starts[i] for i in range(3)] == [0, 6, 15]
    assert [pkg.recv_vec_starts[i] for i in range(3)] == [0, 9, 12]
    assert [pkg.send_map_elmts[i] for i in range(15)] == [e + 100 * j for e in elmts for j in range(3)]
    lib.hypre_ParCSRCommPkgUpdateVecStarts(C.byref(pkg), 1, 100, 1)
    B.check()
    assert pkg.num_components == 1
    assert [pkg.send_map_starts[i] for i in range(3)] == [0, 2, 5]
    assert [pkg.recv_vec_starts[i] for i in range(3)] == [0, 3, 4]
    assert [pkg.send_map_elmts[i] for i in range(5)] == list(elmts)


def _overwrite(lib, dA, A):
    """the device arrays of dA now hold A (same shape, same entry count): what a caller does who frees a matrix with
    hypre's own hypre_CSRMatrixDestroy — which knows nothing of this library's plans — and builds the next one of the
    same size, which lands on the same addresses"""
    import ctypes as C
    from hypre_amd import binding as B
    s = dA.contents
    ii = np.ascontiguousarray(A.indptr, dtype=np.int32)
    jj = np.ascontiguousarray(A.indices, dtype=np.int32)
    aa = np.ascontiguousarray(A.data, dtype=np.float64)
    assert len(ii) == s.num_rows + 1 and len(jj) == s.num_nonzeros
    for dst, src in ((s.i, ii), (s.j, jj), (s.data, aa)):
        lib.hypre_Memcpy(C.cast(dst, C.c_void_p), src.ctypes.data_as(C.c_void_p), src.nbytes, B.HYPRE_MEMORY_DEVICE, B.HYPRE_MEMORY_HOST)


@pytest.mark.parametrize("how", ["columns", "rows"])
def test_a_plan_that_outlived_its_matrix_is_found_out(gpu_lib, oracle, how):
    """The x-staged kernel reads per-entry indices from its plan, not the column array: a plan that survives its matrix
    (same struct address, same array addresses, same sizes: the identity test of get_plan passes) would multiply by the
    OLD pattern.  Every tile therefore compares two entries of the column array with the fingerprint the plan took and
    its first row pointer with the tile table; a mismatch raises a flag in pinned memory, the synchronous public product
    raises HYPRE_ERROR_GENERIC, rebuilds the plan and repeats itself: the caller reads the right product."""
    from hypre_amd import binding as B
    lib = gpu_lib
    A1 = random_csr(9000, 9000, 6, 40, seed=11)
    if how == "columns":
        # same row pointers, other columns
        perm = np.random.default_rng(5).permutation(9000)
        A2 = sp.csr_matrix((A1.data * 0.5, perm[A1.indices].astype(np.int32), A1.indptr), shape=A1.shape)
    else:
        # other row lengths (the rows in reverse order), same entry count
        A2 = sp.csr_matrix(A1[::-1, :])
    x = rand_vector(9000, 3)
    dA = B.csr_from_scipy(A1)
    dx, dy = B.vec_from_numpy(x), B.vec_from_numpy(np.zeros(9000))
    lib.hypre_CSRMatrixMatvec(1.0, dA, dx, 0.0, dy)
    B.check()
    assert np.all(np.abs(B.vec_to_numpy(dy) - A1 @ x) <= _bound(A1, x, 1.0, 0.0, x))
    _overwrite(lib, dA, A2)
    lib.hypre_CSRMatrixMatvec(1.0, dA, dx, 0.0, dy)
    with pytest.raises(B.HypreAmdError):
        B.check()                         # the stale plan was noticed and reported ...
    lib.HYPRE_ClearAllErrors()
    assert np.all(np.abs(B.vec_to_numpy(dy) - A2 @ x) <= _bound(A2, x, 1.0, 0.0, x))      # ... and the product is A2's
    lib.hypre_CSRMatrixMatvec(1.0, dA, dx, 0.0, dy)
    B.check()                             # the rebuilt plan is A2's: no complaint
    assert np.all(np.abs(B.vec_to_numpy(dy) - A2 @ x) <= _bound(A2, x, 1.0, 0.0, x))
    # the announced way: tell the library, and nothing is raised
    _overwrite(lib, dA, A1)
    lib.hypre_amd_CSRMatrixInvalidatePlan(dA)
    lib.hypre_CSRMatrixMatvec(1.0, dA, dx, 0.0, dy)
    B.check()
    assert np.all(np.abs(B.vec_to_numpy(dy) - A1 @ x) <= _bound(A1, x, 1.0, 0.0, x))
    for o in (dx, dy):
        lib.hypre_SeqVectorDestroy(o)
    lib.hypre_CSRMatrixDestroy(dA)


def test_sort_rows_and_byte_counters(gpu_lib, oracle):
    """hypre_amd_CSRMatrixSortRows: columns ascending inside every row, a row's first entry kept in front, the product
    unchanged up to the order of a row's sum.  hypre_amd_ByteCounters: one product accounts for the SURVEY 8(d) count of its
    matrix (12 bytes per entry, row pointers, x once, y once) and, as streamed, for what the launched kernel's format requires."""
    import ctypes as C
    from hypre_amd import binding as B
    lib = gpu_lib
    A = random_csr(7000, 7000, 3, 60, seed=21)
    x = rand_vector(7000, 4)
    dA = B.csr_from_scipy(A)
    dx, dy = B.vec_from_numpy(x), B.vec_from_numpy(np.zeros(7000))
    lib.hypre_CSRMatrixMatvec(1.0, dA, dx, 0.0, dy)          # builds the plan
    csr, streamed = C.c_double(), C.c_double()
    lib.hypre_amd_ByteCounters(None, None, 1)
    lib.hypre_CSRMatrixMatvec(1.0, dA, dx, 0.0, dy)
    lib.hypre_amd_ByteCounters(C.byref(csr), C.byref(streamed), 1)
    nnz, n = A.nnz, 7000
    assert csr.value == nnz * 12 + (n + 1) * 4 + n * 8 + n * 8
    # streamed: what the launched kernel's format requires — 2 bytes less per entry where x is staged (16-bit local indices,
    # no column array), plus the per-tile tables (bounds, 96 piece descriptors, fingerprint: 408 bytes per 2048-entry tile)
    tiles = (nnz + 2047) // 2048
    assert streamed.value in (csr.value + 8 * tiles, csr.value - 2 * nnz + 408 * tiles)
    y0 = B.vec_to_numpy(dy)
    lib.hypre_amd_CSRMatrixSortRows(dA, 1)
    B.check()
    ii, jj, aa = B.csr_to_arrays(dA)
    assert np.array_equal(ii, A.indptr)
    for r in (0, 1, 17, 3500, 6999):
        b, e = ii[r], ii[r + 1]
        assert jj[b] == A.indices[b] and aa[b] == A.data[b]                        # first entry stays
        assert np.all(np.diff(jj[b + 1:e]) > 0)                                     # the rest ascends
        assert sorted(zip(jj[b:e], aa[b:e])) == sorted(zip(A.indices[b:e], A.data[b:e]))
    lib.hypre_CSRMatrixMatvec(1.0, dA, dx, 0.0, dy)
    B.check()
    assert np.all(np.abs(B.vec_to_numpy(dy) - y0) <= _bound(A, x, 1.0, 0.0, x))
    for o in (dx, dy):
        lib.hypre_SeqVectorDestroy(o)
    lib.hypre_CSRMatrixDestroy(dA)


def _table_matrix(shape, lo, hi, nvals, seed):
    """random pattern whose values are drawn from a table of nvals distinct doubles (signed zeros, a subnormal and an
    infinity-free mix of magnitudes among them), every table entry used at least once"""
    A = random_csr(shape[0], shape[1], lo, hi, seed=seed)
    rng = np.random.default_rng(seed + 100)
    table = np.unique(np.concatenate([[0.0, 6.0, -1.0, 5e-324, 1e300, -1e-300], rng.uniform(-3, 3, nvals + 8)]))
    table = table[:nvals].copy()
    if nvals >= 2:
        table[-1] = -0.0 if 0.0 in table[:-1] else table[-1]        # 0.0 and -0.0 are two table entries (bit patterns)
    pick = rng.integers(0, len(table), A.nnz)
    pick[:len(table)] = np.arange(len(table))
    A.data[:] = table[pick]
    return A, len(np.unique(table.view(np.int64)))


@pytest.mark.parametrize("nvals,shape,lo,hi", [(1, (6000, 6000), 3, 30), (2, (6000, 6000), 3, 30), (7, (20000, 20000), 5, 9),
                                               (255, (6000, 6000), 20, 60), (256, (6000, 6000), 20, 60),
                                               (257, (6000, 6000), 20, 60), (5000, (6000, 6000), 20, 60),
                                               (3, (3000, 2500), 200, 900), (3, (4000, 400000), 10, 40)])
def test_value_codes_give_the_same_bits(gpu_lib, oracle, nvals, shape, lo, hi):
    """A matrix with at most 256 distinct values (bit patterns) is streamed by the x-staged kernel as one byte per entry
    plus a table staged in LDS: the product must be the SAME BITS as with the fp64 stream (same values, same order of
    every row's sum) — staged tiles, tiles that gather (400 000 columns), spilling last rows (up to 900 entries) — and a
    matrix with more values must not be coded at all."""
    from hypre_amd import binding as B
    lib = gpu_lib
    A, distinct = _table_matrix(shape, lo, hi, nvals, seed=nvals)
    x = rand_vector(shape[1], 7)
    b = rand_vector(shape[0], 8)
    out = {}
    try:
        for on in (1, 0):
            lib.hypre_amd_SpmvSetValueCodes(on)
            dA = B.csr_from_scipy(A)
            dx, db, dy = B.vec_from_numpy(x), B.vec_from_numpy(b), B.vec_from_numpy(np.zeros(shape[0]))
            lib.hypre_CSRMatrixMatvecOutOfPlace(-0.75, dA, dx, 1.5, db, dy, 0)
            B.check()
            coded = lib.hypre_amd_CSRMatrixPlanValueCodes(dA)
            assert coded == (distinct if (on and distinct <= 256) else 0), (coded, distinct)
            out[on] = B.vec_to_numpy(dy)
            for o in (dx, db, dy):
                lib.hypre_SeqVectorDestroy(o)
            lib.hypre_CSRMatrixDestroy(dA)
    finally:
        lib.hypre_amd_SpmvSetValueCodes(1)
    assert np.array_equal(out[1].view(np.int64), out[0].view(np.int64))
    with np.errstate(over="ignore", invalid="ignore"):
        bound = _bound(A, x, -0.75, 1.5, b)
    ok = np.isfinite(bound)
    assert np.all(np.abs(out[1] - (-0.75 * (A @ x) + 1.5 * b))[ok] <= bound[ok])


def test_a_matrix_that_only_starts_like_a_stencil_is_not_coded(gpu_lib, oracle):
    """The search for value codes looks at the first 16 384 values before it scans them all: a matrix whose head holds three
    values and whose tail holds thousands must come out uncoded (and multiplied right)."""
    from hypre_amd import binding as B
    lib = gpu_lib
    A = random_csr(6000, 6000, 8, 12, seed=77)
    rng = np.random.default_rng(78)
    A.data[:] = rng.choice([6.0, -1.0, 0.5], A.nnz)
    A.data[40000:] = rng.uniform(-1, 1, A.nnz - 40000)
    x = rand_vector(6000, 5)
    dA = B.csr_from_scipy(A)
    dx, dy = B.vec_from_numpy(x), B.vec_from_numpy(np.zeros(6000))
    lib.hypre_CSRMatrixMatvec(1.0, dA, dx, 0.0, dy)
    B.check()
    assert lib.hypre_amd_CSRMatrixPlanValueCodes(dA) == 0
    assert np.all(np.abs(B.vec_to_numpy(dy) - A @ x) <= _bound(A, x, 1.0, 0.0, x))
    for o in (dx, dy):
        lib.hypre_SeqVectorDestroy(o)
    lib.hypre_CSRMatrixDestroy(dA)


def test_values_changed_behind_a_coded_plan_are_found_out(gpu_lib, oracle):
    """The coded kernel never reads the fp64 values: every tile compares the first value it decodes with the fp64 original,
    so values changed in place without hypre_amd_CSRMatrixInvalidatePlan raise HYPRE_ERROR_GENERIC and the synchronous
    product repeats itself with a fresh plan."""
    from hypre_amd import binding as B
    lib = gpu_lib
    A1 = laplace_3d(20, 20, 20)
    A2 = sp.csr_matrix((A1.data * 0.5, A1.indices, A1.indptr), shape=A1.shape)
    n = A1.shape[0]
    x = rand_vector(n, 3)
    dA = B.csr_from_scipy(A1)
    dx, dy = B.vec_from_numpy(x), B.vec_from_numpy(np.zeros(n))
    lib.hypre_CSRMatrixMatvec(1.0, dA, dx, 0.0, dy)
    B.check()
    assert lib.hypre_amd_CSRMatrixPlanValueCodes(dA) == 2
    assert np.all(np.abs(B.vec_to_numpy(dy) - A1 @ x) <= _bound(A1, x, 1.0, 0.0, x))
    _overwrite(lib, dA, A2)
    lib.hypre_CSRMatrixMatvec(1.0, dA, dx, 0.0, dy)
    with pytest.raises(B.HypreAmdError):
        B.check()
    lib.HYPRE_ClearAllErrors()
    assert np.all(np.abs(B.vec_to_numpy(dy) - A2 @ x) <= _bound(A2, x, 1.0, 0.0, x))
    lib.hypre_CSRMatrixMatvec(1.0, dA, dx, 0.0, dy)
    B.check()
    for o in (dx, dy):
        lib.hypre_SeqVectorDestroy(o)
    lib.hypre_CSRMatrixDestroy(dA)


def _fixed_rows_matrix(n, K, nvals, seed, short_every=0):
    """every row K entries (or, every `short_every`-th row, K // 2) at random columns of a band around the diagonal, the diagonal
    first; values from a table of nvals doubles"""
    rng = np.random.default_rng(seed)
    table = np.concatenate([[6.0, -1.0], rng.uniform(-2, 2, max(nvals - 2, 0))])[:nvals]
    counts = np.full(n, K, dtype=np.int64)
    if short_every:
        counts[::short_every] = max(K // 2, 1)
    indptr = np.zeros(n + 1, dtype=np.int32)
    indptr[1:] = np.cumsum(counts)
    indices = np.empty(indptr[-1], dtype=np.int32)
    for r in range(n):
        lo, hi = max(0, r - 300), min(n, r + 300)
        others = rng.choice(np.setdiff1d(np.arange(lo, hi), [r]), counts[r] - 1, replace=False)
        indices[indptr[r]] = r
        indices[indptr[r] + 1:indptr[r + 1]] = others
    data = table[rng.integers(0, len(table), indptr[-1])]
    return sp.csr_matrix((data, indices, indptr), shape=(n, n))


@pytest.mark.parametrize("what,lanes", [("7pt", 1), ("27pt", 2), ("K8", 1), ("K13", 2), ("K16", 2), ("K27", 2), ("K32", 2),
                                        ("K7short", 1), ("K3", 0), ("K20", 0), ("K40", 0)])
def test_slice_form_of_coded_stencils(gpu_lib, oracle, what, lanes):
    """A coded matrix with short, equally long rows is multiplied by spmv_sl_kernel (a lane per row, or per half row): the
    same products as the tiled kernel — bit for bit where both sum a row in stored order (rows of at most 8 entries), within
    the tolerance of this file otherwise — for every epilogue of y = alpha A x + beta b; matrices the form does not fit
    (rows much shorter than 8 / 16 / 32 entries, rows longer than 32) do not get it."""
    from hypre_amd import binding as B
    lib = gpu_lib
    if what == "7pt":
        A = laplace_3d(24, 20, 18)
    elif what == "27pt":
        A = laplace_3d(30, 30, 30, 27)
    elif what == "K7short":
        A = _fixed_rows_matrix(5000, 7, 3, 5, short_every=9)
    else:
        A = _fixed_rows_matrix(5000, int(what[1:]), 5, int(what[1:]))
    n = A.shape[0]
    x, b = rand_vector(n, 1), rand_vector(n, 2)
    out = {}
    try:
        for on in (1, 0):
            lib.hypre_amd_SpmvSetSliceForm(on)
            dA = B.csr_from_scipy(A)
            dx, db, dy = B.vec_from_numpy(x), B.vec_from_numpy(b), B.vec_from_numpy(np.zeros(n))
            res = []
            for alpha, beta in ((1.0, 0.0), (-1.0, 1.0), (0.7, -0.3)):
                lib.hypre_CSRMatrixMatvecOutOfPlace(alpha, dA, dx, beta, db, dy, 0)
                B.check()
                res.append(B.vec_to_numpy(dy))
            assert lib.hypre_amd_CSRMatrixPlanValueCodes(dA) > 0
            assert lib.hypre_amd_CSRMatrixPlanSliceForm(dA) == (lanes if on else 0)
            out[on] = res
            for o in (dx, db, dy):
                lib.hypre_SeqVectorDestroy(o)
            lib.hypre_CSRMatrixDestroy(dA)
    finally:
        lib.hypre_amd_SpmvSetSliceForm(1)
    for (alpha, beta), y1, y0 in zip(((1.0, 0.0), (-1.0, 1.0), (0.7, -0.3)), out[1], out[0]):
        ref = alpha * (A @ x) + beta * b
        assert np.all(np.abs(y1 - ref) <= _bound(A, x, alpha, beta, b))
        if lanes == 1:
            assert np.array_equal(y1.view(np.int64), y0.view(np.int64))
        else:
            assert np.all(np.abs(y1 - y0) <= _bound(A, x, alpha, beta, b))


def test_the_slice_kernel_notices_another_matrix(gpu_lib, oracle):
    """spmv_sl_kernel reads neither the column array nor the values: per block it compares two columns with the plan's
    fingerprint, the first row pointer with the plan's and one decoded value with its fp64 original."""
    from hypre_amd import binding as B
    lib = gpu_lib
    A1 = _fixed_rows_matrix(6000, 8, 4, 3)
    perm = np.arange(6000)
    perm[1::2], perm[0:-1:2] = np.arange(0, 5999, 2), np.arange(1, 6000, 2)          # neighbours swapped: the band stays
    A2 = sp.csr_matrix((A1.data, perm[A1.indices].astype(np.int32), A1.indptr), shape=A1.shape)
    x = rand_vector(6000, 3)
    dA = B.csr_from_scipy(A1)
    dx, dy = B.vec_from_numpy(x), B.vec_from_numpy(np.zeros(6000))
    lib.hypre_CSRMatrixMatvec(1.0, dA, dx, 0.0, dy)
    B.check()
    assert lib.hypre_amd_CSRMatrixPlanSliceForm(dA) == 1
    assert np.all(np.abs(B.vec_to_numpy(dy) - A1 @ x) <= _bound(A1, x, 1.0, 0.0, x))
    _overwrite(lib, dA, A2)
    lib.hypre_CSRMatrixMatvec(1.0, dA, dx, 0.0, dy)
    with pytest.raises(B.HypreAmdError):
        B.check()
    lib.HYPRE_ClearAllErrors()
    assert np.all(np.abs(B.vec_to_numpy(dy) - A2 @ x) <= _bound(A2, x, 1.0, 0.0, x))
    for o in (dx, dy):
        lib.hypre_SeqVectorDestroy(o)
    lib.hypre_CSRMatrixDestroy(dA)


def _poke(lib, dA, positions, values):
    """entries `positions` of the device value array of dA now hold `values`: an in-place edit of some coefficients, the
    hypre idiom (HYPRE_IJMatrixSetValues on the same pattern) — nothing is told to the library"""
    import ctypes as C
    from hypre_amd import binding as B
    base = C.cast(dA.contents.data, C.c_void_p).value
    for k, v in zip(positions, values):
        src = np.array([v], dtype=np.float64)
        lib.hypre_Memcpy(C.c_void_p(base + 8 * int(k)), src.ctypes.data_as(C.c_void_p), 8, B.HYPRE_MEMORY_DEVICE, B.HYPRE_MEMORY_HOST)


@pytest.mark.parametrize("form,bound", [("slice", 128), ("coded tiles", 64), ("fp32 copy", 64)])
@pytest.mark.parametrize("edit", ["one coefficient in the middle of a tile", "one row scaled"])
def test_any_coefficient_edited_in_place_is_found_within_the_bound(gpu_lib, oracle, form, bound, edit):
    """The kernels that multiply by a private copy of the values — value codes, the slice form, the fp32 copy of the mixed-
    precision mode — compare a ROTATING sample of the copy with the caller's fp64 array: eight consecutive entries per wave
    at a position that moves with the plan's launch counter.  So ONE coefficient changed in place, anywhere, without
    hypre_amd_CSRMatrixInvalidatePlan, is found within 64 (tiled kernel) or 128 (slice kernel) products: HYPRE_ERROR_GENERIC
    is raised, the plan rebuilt, and the synchronous product repeats itself so that the caller reads the NEW matrix's
    product.  (Round 3 sampled one fixed entry per tile: only a wholesale replacement was noticed.)"""
    import ctypes as C
    from hypre_amd import binding as B
    lib = gpu_lib
    A = laplace_3d(24, 22, 20).tocsr()
    A.sort_indices()
    n = A.shape[0]
    x = rand_vector(n, 9)
    if edit.startswith("one coefficient"):
        ks = [2048 * 7 + 1001]                               # the middle of tile 7
        new = [0.375]
    else:
        r = n // 2 + 3
        ks = list(range(A.indptr[r], A.indptr[r + 1]))
        new = list(1.5 * A.data[ks])
    A2 = A.copy()
    A2.data[ks] = new
    try:
        lib.hypre_amd_SpmvSetSliceForm(1 if form == "slice" else 0)
        dA = B.csr_from_scipy(A)
        dx, dy = B.vec_from_numpy(x), B.vec_from_numpy(np.zeros(n))
        if form == "fp32 copy":
            lib.hypre_amd_SpmvSetValueCodes(0)
            lib.hypre_amd_SetMixedPrecisionValues(1)
        lib.hypre_CSRMatrixMatvec(1.0, dA, dx, 0.0, dy)
        B.check()
        assert lib.hypre_amd_CSRMatrixPlanForm(dA) == {"slice": 4, "coded tiles": 3, "fp32 copy": 2}[form]
        _poke(lib, dA, ks, new)
        found = None
        for launch in range(1, bound + 1):
            lib.hypre_CSRMatrixMatvec(1.0, dA, dx, 0.0, dy)
            if lib.HYPRE_GetError():
                found = launch
                break
        assert found is not None, "an edited coefficient went unnoticed for %d products" % bound
        lib.HYPRE_ClearAllErrors()
        tol = _bound(A2, x, 1.0, 0.0, x) if form != "fp32 copy" else 1e-6 * (abs(A2) @ np.abs(x))
        assert np.all(np.abs(B.vec_to_numpy(dy) - A2 @ x) <= tol)          # the repeated product is the new matrix's
        lib.hypre_CSRMatrixMatvec(1.0, dA, dx, 0.0, dy)
        B.check()                                                          # the rebuilt plan stands
        for o in (dx, dy):
            lib.hypre_SeqVectorDestroy(o)
        lib.hypre_CSRMatrixDestroy(dA)
    finally:
        lib.hypre_amd_SpmvSetSliceForm(1)
        lib.hypre_amd_SpmvSetValueCodes(1)
        lib.hypre_amd_SetMixedPrecisionValues(0)


def test_an_in_place_product_with_beta_is_not_repeated(gpu_lib, oracle):
    """hypre_CSRMatrixMatvec(alpha, A, x, beta, y) with beta != 0 overwrites the operand a repeat would need: when the kernels
    find the plan out of date the error is raised, the plan rebuilt — and y is documented as NOT valid (the next call is right)."""
    from hypre_amd import binding as B
    lib = gpu_lib
    A1 = laplace_3d(20, 20, 20)
    A2 = sp.csr_matrix((A1.data * 0.5, A1.indices, A1.indptr), shape=A1.shape)
    n = A1.shape[0]
    x, y0 = rand_vector(n, 3), rand_vector(n, 4)
    dA = B.csr_from_scipy(A1)
    dx, dy = B.vec_from_numpy(x), B.vec_from_numpy(y0)
    lib.hypre_CSRMatrixMatvec(1.0, dA, dx, 0.5, dy)
    B.check()
    y1 = B.vec_to_numpy(dy)
    assert np.all(np.abs(y1 - (A1 @ x + 0.5 * y0)) <= _bound(A1, x, 1.0, 0.5, y0))
    _overwrite(lib, dA, A2)
    lib.hypre_CSRMatrixMatvec(1.0, dA, dx, 0.5, dy)
    with pytest.raises(B.HypreAmdError):
        B.check()
    lib.HYPRE_ClearAllErrors()
    # y now holds alpha A1 x + beta y1 (the stale product): reported, not repaired; the next call multiplies by A2
    dz = B.vec_from_numpy(y0)
    lib.hypre_CSRMatrixMatvec(1.0, dA, dx, 0.5, dz)
    B.check()
    assert np.all(np.abs(B.vec_to_numpy(dz) - (A2 @ x + 0.5 * y0)) <= _bound(A2, x, 1.0, 0.5, y0))
    for o in (dx, dy, dz):
        lib.hypre_SeqVectorDestroy(o)
    lib.hypre_CSRMatrixDestroy(dA)


def test_verify_plan_finds_any_edit_at_once(gpu_lib, oracle):
    """hypre_amd_CSRMatrixVerifyPlan: the checksum of row pointers, columns and value bit patterns taken when the plan was
    built against the arrays as they are now — one changed coefficient, one changed column: the plan is dropped silently and
    the next product is the new matrix's, with no error raised."""
    from hypre_amd import binding as B
    lib = gpu_lib
    A = random_csr(5000, 5000, 5, 40, seed=11)
    x = rand_vector(5000, 1)
    dA = B.csr_from_scipy(A)
    dx, dy = B.vec_from_numpy(x), B.vec_from_numpy(np.zeros(5000))
    lib.hypre_CSRMatrixMatvec(1.0, dA, dx, 0.0, dy)
    B.check()
    assert lib.hypre_amd_CSRMatrixVerifyPlan(dA) == 1
    A2 = A.copy()
    A2.data[12345] *= 1.0 + 2.0 ** -40                      # a few ulps
    _overwrite(lib, dA, A2)
    assert lib.hypre_amd_CSRMatrixVerifyPlan(dA) == 0
    assert lib.hypre_amd_CSRMatrixVerifyPlan(dA) == 1       # no plan left to object to
    lib.hypre_CSRMatrixMatvec(1.0, dA, dx, 0.0, dy)
    B.check()
    assert np.all(np.abs(B.vec_to_numpy(dy) - A2 @ x) <= _bound(A2, x, 1.0, 0.0, x))
    A3 = A2.copy()
    A3.indices[777] = (A3.indices[777] + 1) % 5000 if (A3.indices[777] + 1) % 5000 not in A3.indices[A3.indptr[np.searchsorted(A3.indptr, 777, side="right") - 1]:A3.indptr[np.searchsorted(A3.indptr, 777, side="right")]] else A3.indices[777]
    _overwrite(lib, dA, A3)
    changed = not np.array_equal(A3.indices, A2.indices)
    assert lib.hypre_amd_CSRMatrixVerifyPlan(dA) == (0 if changed else 1)
    lib.hypre_CSRMatrixMatvec(1.0, dA, dx, 0.0, dy)
    B.check()
    assert np.all(np.abs(B.vec_to_numpy(dy) - A3 @ x) <= _bound(A3, x, 1.0, 0.0, x))
    for o in (dx, dy):
        lib.hypre_SeqVectorDestroy(o)
    lib.hypre_CSRMatrixDestroy(dA)


@pytest.mark.parametrize("site,lowest", [(1, 0), (2, 1), (3, 2), (4, 3)])
@pytest.mark.parametrize("nth", [1, 2, 3, 4, 5, 6])
def test_a_failed_plan_allocation_falls_back_one_form(gpu_lib, oracle, site, lowest, nth):
    """Every allocation a plan builder makes is checked: with the nth allocation of a site made to fail — tile tables,
    x-staging tables, value codes, slice form — the builder frees what the step had obtained, leaves hypre_error_flag clean
    and the matrix is multiplied by the form below (a wave per row, tiles gathering x, uncoded tiles, coded tiles)."""
    from hypre_amd import binding as B
    lib = gpu_lib
    A = laplace_3d(22, 20, 18)
    n = A.shape[0]
    x, b = rand_vector(n, 2), rand_vector(n, 3)
    dA = B.csr_from_scipy(A)
    dx, db, dy = B.vec_from_numpy(x), B.vec_from_numpy(b), B.vec_from_numpy(np.zeros(n))
    try:
        lib.hypre_amd_PlanTestFailAlloc(site, nth)
        lib.hypre_CSRMatrixMatvecOutOfPlace(-0.5, dA, dx, 2.0, db, dy, 0)
        B.check()                                            # nothing raised
        form = lib.hypre_amd_CSRMatrixPlanForm(dA)
        happened = lib.hypre_amd_PlanTestFailAlloc(0, 0) == 0     # (a site makes two to seven allocations for this matrix)
        assert nth > 2 or happened
        assert (form <= lowest) if happened else (form == 4), (form, lowest, happened)
        assert np.all(np.abs(B.vec_to_numpy(dy) - (-0.5 * (A @ x) + 2.0 * b)) <= _bound(A, x, -0.5, 2.0, b))
        # the next plan of the same matrix gets everything again
        lib.hypre_amd_CSRMatrixInvalidatePlan(dA)
        lib.hypre_CSRMatrixMatvecOutOfPlace(-0.5, dA, dx, 2.0, db, dy, 0)
        B.check()
        assert lib.hypre_amd_CSRMatrixPlanForm(dA) == 4
        assert np.all(np.abs(B.vec_to_numpy(dy) - (-0.5 * (A @ x) + 2.0 * b)) <= _bound(A, x, -0.5, 2.0, b))
    finally:
        lib.hypre_amd_PlanTestFailAlloc(0, 0)
    for o in (dx, db, dy):
        lib.hypre_SeqVectorDestroy(o)
    lib.hypre_CSRMatrixDestroy(dA)


@pytest.mark.parametrize("lo,hi,n,lanes", [(4, 12, 6000, 0), (13, 48, 5000, 2), (20, 40, 9000, 2), (49, 160, 3000, 8), (60, 90, 2500, 4),
                                           (200, 900, 1500, 32), (0, 30, 4000, 1), (1, 3, 6000, 0), (33, 64, 2000, 4), (10, 20, 300, 1)])
def test_row_slice_form(gpu_lib, oracle, lo, hi, n, lanes):
    """An uncoded matrix that cannot change behind its plan is multiplied from jagged row slices (spmv_rs_kernel: 256 / W rows a
    workgroup, W lanes a row, a lane's entries summed in stored order from registers, the W partial sums of a row added in
    lane order): every epilogue of y = alpha A x + beta b against scipy within the tolerance of this file and against the
    tiled kernel's result; every width W = 1 ... 32 and every register form (8 ... 32 entries a lane); empty rows; rows too
    short for the form to pay (a dozen entries or fewer on average) keep the tiles.  The form is for matrices the library owns (coarse levels) or the caller
    declared immutable: here hypre_amd_CSRMatrixSetImmutable."""
    import ctypes as C
    from hypre_amd import binding as B
    lib = gpu_lib
    A = banded_csr(n, n - 100, lo, hi, max(hi + 100, 500), seed=hi + n, empty_frac=0.02 if lo == 0 else 0.0)
    x, b = rand_vector(n - 100, 1), rand_vector(n, 2)
    out = {}
    for immutable in (1, 0):
        dA = B.csr_from_scipy(A)
        lib.hypre_amd_CSRMatrixSetImmutable(dA, immutable)
        dx, db, dy = B.vec_from_numpy(x), B.vec_from_numpy(b), B.vec_from_numpy(np.zeros(n))
        res = []
        for alpha, beta in ((1.0, 0.0), (-1.0, 1.0), (0.7, -0.3)):
            lib.hypre_CSRMatrixMatvecOutOfPlace(alpha, dA, dx, beta, db, dy, 0)
            B.check()
            res.append(B.vec_to_numpy(dy))
        rows, per = C.c_int(), C.c_int()
        w = lib.hypre_amd_CSRMatrixPlanRowSlices(dA, C.byref(rows), C.byref(per))
        form = lib.hypre_amd_CSRMatrixPlanForm(dA)
        if immutable and lanes:
            assert form == 5 and w == lanes, (form, w)
            assert rows.value == 256 // w and per.value in (8, 16, 24, 32) and per.value * w >= np.diff(A.indptr).max()
        else:
            assert form == 2 and w == 0
        out[immutable] = res
        for o in (dx, db, dy):
            lib.hypre_SeqVectorDestroy(o)
        lib.hypre_CSRMatrixDestroy(dA)
    for (alpha, beta), y1, y0 in zip(((1.0, 0.0), (-1.0, 1.0), (0.7, -0.3)), out[1], out[0]):
        ref = alpha * (A @ x) + beta * b
        assert np.all(np.abs(y1 - ref) <= _bound(A, x, alpha, beta, b))
        assert np.all(np.abs(y1 - y0) <= _bound(A, x, alpha, beta, b))


def test_row_slices_of_every_matrix_and_their_fallbacks(gpu_lib, oracle):
    """hypre_amd_SpmvSetRowSlices(2): every uncoded matrix gets the form (the caller then owes InvalidatePlan after a change);
    a failed allocation of the form's tables (site 5) leaves the tiles; a coded matrix keeps its codes; in-place products and
    the transposed product go through it; mixed precision multiplies by the fp32 copy of the slices."""
    import ctypes as C
    from hypre_amd import binding as B
    lib = gpu_lib
    A = banded_csr(7000, 7000, 15, 45, 400, seed=5)
    x, y0 = rand_vector(7000, 1), rand_vector(7000, 2)
    try:
        lib.hypre_amd_SpmvSetRowSlices(2)
        for nth in (0, 1, 4, 5, 7):
            dA = B.csr_from_scipy(A)
            dx, dy = B.vec_from_numpy(x), B.vec_from_numpy(y0)
            lib.hypre_amd_PlanTestFailAlloc(5, nth)
            lib.hypre_CSRMatrixMatvec(-0.5, dA, dx, 1.5, dy)
            B.check()
            happened = nth > 0 and lib.hypre_amd_PlanTestFailAlloc(0, 0) == 0
            assert lib.hypre_amd_CSRMatrixPlanForm(dA) == (2 if happened else 5), (nth, happened)
            assert np.all(np.abs(B.vec_to_numpy(dy) - (-0.5 * (A @ x) + 1.5 * y0)) <= _bound(A, x, -0.5, 1.5, y0))
            # the transposed product: the cached transpose is a matrix of its own
            dz = B.vec_from_numpy(np.zeros(7000))
            lib.hypre_CSRMatrixMatvecT(1.0, dA, dx, 0.0, dz)
            B.check()
            assert np.all(np.abs(B.vec_to_numpy(dz) - A.T @ x) <= _bound(A.T.tocsr(), x, 1.0, 0.0, x))
            # mixed precision: the values rounded through fp32
            lib.hypre_amd_SetMixedPrecisionValues(1)
            lib.hypre_CSRMatrixMatvec(1.0, dA, dx, 0.0, dz)
            lib.hypre_amd_SetMixedPrecisionValues(0)
            B.check()
            A32 = sp.csr_matrix((A.data.astype(np.float32).astype(np.float64), A.indices, A.indptr), shape=A.shape)
            assert np.all(np.abs(B.vec_to_numpy(dz) - A32 @ x) <= _bound(A32, x, 1.0, 0.0, x))
            for o in (dx, dy, dz):
                lib.hypre_SeqVectorDestroy(o)
            lib.hypre_CSRMatrixDestroy(dA)
        # short rows: taken under mode 2 only (8 entries a lane, a lane a row)
        Sh = banded_csr(5000, 5000, 4, 12, 300, seed=9)
        dSh = B.csr_from_scipy(Sh)
        xh, yh = B.vec_from_numpy(rand_vector(5000, 3)), B.vec_from_numpy(np.zeros(5000))
        lib.hypre_CSRMatrixMatvec(1.0, dSh, xh, 0.0, yh)
        B.check()
        assert lib.hypre_amd_CSRMatrixPlanForm(dSh) == 5 and lib.hypre_amd_CSRMatrixPlanRowSlices(dSh, None, None) == 1
        assert np.all(np.abs(B.vec_to_numpy(yh) - Sh @ rand_vector(5000, 3)) <= _bound(Sh, rand_vector(5000, 3), 1.0, 0.0, rand_vector(5000, 3)))
        for o in (xh, yh):
            lib.hypre_SeqVectorDestroy(o)
        lib.hypre_CSRMatrixDestroy(dSh)
        S = laplace_3d(16, 16, 16)
        dS = B.csr_from_scipy(S)
        xs_, ys_ = B.vec_from_numpy(rand_vector(4096, 3)), B.vec_from_numpy(np.zeros(4096))
        lib.hypre_CSRMatrixMatvec(1.0, dS, xs_, 0.0, ys_)
        B.check()
        assert lib.hypre_amd_CSRMatrixPlanForm(dS) == 4          # a stencil: codes in slice form, not row slices
        for o in (xs_, ys_):
            lib.hypre_SeqVectorDestroy(o)
        lib.hypre_CSRMatrixDestroy(dS)
    finally:
        lib.hypre_amd_SpmvSetRowSlices(1)
        lib.hypre_amd_PlanTestFailAlloc(0, 0)
        lib.hypre_amd_SetMixedPrecisionValues(0)
