"""Shared helpers for the tests: seeded inputs and small sparse matrices."""
import numpy as np
import scipy.sparse as sp


def lcg_vector(n, seed=1):
    """Deterministic values in (-1, 1) from a fixed LCG (no platform RNG)."""
    out = np.empty(n)
    s = np.uint64(seed * 2654435761 % (2 ** 32) + 12345)
    a = np.uint64(6364136223846793005)
    c = np.uint64(1442695040888963407)
    with np.errstate(over="ignore"):
        for i in range(n):
            s = s * a + c
            out[i] = (float(s >> np.uint64(11)) / float(1 << 53)) * 2.0 - 1.0
    return out


def rand_vector(n, seed=1):
    return np.random.default_rng(seed).uniform(-1.0, 1.0, n)


def laplace_3d(nx, ny, nz, stencil=7):
    """Global 7- or 27-point Laplacian, diagonal first in every row."""
    N = nx * ny * nz
    rows, cols, vals = [], [], []
    idx = lambda x, y, z: (z * ny + y) * nx + x
    for z in range(nz):
        for y in range(ny):
            for x in range(nx):
                r = idx(x, y, z)
                if stencil == 7:
                    nb = [(0, 0, -1), (0, -1, 0), (-1, 0, 0), (1, 0, 0), (0, 1, 0), (0, 0, 1)]
                    d = 6.0
                else:
                    nb = [(dx, dy, dz) for dz in (-1, 0, 1) for dy in (-1, 0, 1) for dx in (-1, 0, 1)
                          if (dx, dy, dz) != (0, 0, 0)]
                    d = 26.0
                rows.append(r); cols.append(r); vals.append(d)
                for dx, dy, dz in nb:
                    X, Y, Z = x + dx, y + dy, z + dz
                    if 0 <= X < nx and 0 <= Y < ny and 0 <= Z < nz:
                        rows.append(r); cols.append(idx(X, Y, Z)); vals.append(-1.0)
    # build CSR by hand to keep the stored order (scipy's coo->csr would sort/merge)
    rows = np.array(rows); cols = np.array(cols, dtype=np.int32); vals = np.array(vals)
    indptr = np.zeros(N + 1, dtype=np.int32)
    np.add.at(indptr, rows + 1, 1)
    indptr = np.cumsum(indptr).astype(np.int32)
    A = sp.csr_matrix((vals, cols, indptr), shape=(N, N))
    return A


def random_csr(nrows, ncols, min_nnz, max_nnz, seed=0, empty_frac=0.0):
    """Rows with a uniformly random number of entries in [min_nnz, max_nnz], unsorted columns."""
    rng = np.random.default_rng(seed)
    counts = rng.integers(min_nnz, max_nnz + 1, nrows)
    if empty_frac > 0:
        counts[rng.random(nrows) < empty_frac] = 0
    counts = np.minimum(counts, ncols)
    indptr = np.zeros(nrows + 1, dtype=np.int32)
    indptr[1:] = np.cumsum(counts)
    indices = np.empty(indptr[-1], dtype=np.int32)
    for r in range(nrows):
        indices[indptr[r]:indptr[r + 1]] = rng.choice(ncols, counts[r], replace=False)
    data = rng.uniform(-1, 1, indptr[-1])
    return sp.csr_matrix((data, indices, indptr), shape=(nrows, ncols))


def banded_csr(nrows, ncols, min_nnz, max_nnz, band, seed=0, empty_frac=0.0):
    """Rows with a uniformly random number of entries in [min_nnz, max_nnz] at random, unsorted columns within `band` of the
    row's own position (scaled to the column range): the locality of an operator on a mesh, all values distinct."""
    rng = np.random.default_rng(seed)
    counts = rng.integers(min_nnz, max_nnz + 1, nrows)
    if empty_frac > 0:
        counts[rng.random(nrows) < empty_frac] = 0
    indptr = np.zeros(nrows + 1, dtype=np.int32)
    lo = np.empty(nrows, dtype=np.int64)
    hi = np.empty(nrows, dtype=np.int64)
    for r in range(nrows):
        c = int(r * (ncols - 1) / max(nrows - 1, 1))
        lo[r], hi[r] = max(0, c - band), min(ncols, c + band + 1)
        counts[r] = min(counts[r], hi[r] - lo[r])
    indptr[1:] = np.cumsum(counts)
    indices = np.empty(indptr[-1], dtype=np.int32)
    for r in range(nrows):
        indices[indptr[r]:indptr[r + 1]] = lo[r] + rng.choice(hi[r] - lo[r], counts[r], replace=False)
    data = rng.uniform(-1, 1, indptr[-1])
    return sp.csr_matrix((data, indices, indptr), shape=(nrows, ncols))
