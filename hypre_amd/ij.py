"""Host-side mirror of the reference's `ij` driver for the in-scope options
(test/ij.c:521-2245 flags, :4440-4780 solver 0, :5007-5200 solver 1).

Everything numerical happens in libhypre_amd.so; this module only translates
driver options into HYPRE_* calls, the way test/ij.c does.
"""
import ctypes as C

import numpy as np

from . import binding as B

HOST, DEVICE = B.HYPRE_MEMORY_HOST, B.HYPRE_MEMORY_DEVICE


class IJOptions:
    """Defaults of test/ij.c (:140-420, :1697-1712)."""

    def __init__(self, **kw):
        self.n = (10, 10, 10)
        self.P = None                 # (P, Q, R); default (1, nprocs, 1)
        self.problem = "laplacian"    # laplacian | 27pt | difconv
        self.c = (1.0, 1.0, 1.0)      # -c cx cy cz
        self.a = (1.0, 1.0, 1.0)      # -a ax ay az (difconv)
        self.solver = 0               # 0 AMG, 1 AMG-PCG
        self.rhs = "one"              # one (-rhsisone default) | rand (-rhsrand) | xisone
        self.coarsen_type = 10
        self.interp_type = 6
        self.P_max_elmts = 4
        self.trunc_factor = 0.0
        self.strong_threshold = 0.25
        self.max_row_sum = 1.0
        self.relax_type = -1          # -rlx
        self.relax_down = -1
        self.relax_up = -1
        self.relax_coarse = -1
        self.relax_order = 0          # -CF
        self.num_sweeps = 1           # -ns
        self.relax_wt = 1.0           # -w
        self.outer_wt = 1.0           # -ow
        self.cycle_type = 1           # -mu
        self.fcycle = 0
        self.max_levels = 25
        self.coarse_threshold = 9
        self.tol = 1.0e-8
        self.mg_max_iter = 100
        self.max_iter = 1000
        self.two_norm = 1
        self.precon_cycles = 1
        self.keep_transpose = 1
        self.num_threads = 1
        for k, v in kw.items():
            if not hasattr(self, k):
                raise TypeError("unknown ij option %r" % k)
            setattr(self, k, v)


def stencil_values(opt):
    """test/ij.c:9703-9719 (laplacian), :10984-10993 (27pt), :10184-10215 (difconv)."""
    nx, ny, nz = opt.n
    cx, cy, cz = opt.c
    if opt.problem == "laplacian":
        v = np.zeros(4)
        v[1], v[2], v[3] = -cx, -cy, -cz
        if nx > 1:
            v[0] += 2.0 * cx
        if ny > 1:
            v[0] += 2.0 * cy
        if nz > 1:
            v[0] += 2.0 * cz
        return v
    if opt.problem == "27pt":
        v = np.zeros(2)
        v[0] = 26.0
        if nx == 1 or ny == 1 or nz == 1:
            v[0] = 8.0
        if nx * ny == 1 or nx * nz == 1 or ny * nz == 1:
            v[0] = 2.0
        v[1] = -1.0
        return v
    if opt.problem == "difconv":
        ax, ay, az = opt.a
        hinx, hiny, hinz = 1.0 / (nx + 1), 1.0 / (ny + 1), 1.0 / (nz + 1)
        v = np.zeros(7)
        # atype 0: forward differencing of the convection term (ij.c:10184-10215)
        if nx > 1:
            v[0] += 2.0 * cx / (hinx * hinx) - 1.0 * ax / hinx
        if ny > 1:
            v[0] += 2.0 * cy / (hiny * hiny) - 1.0 * ay / hiny
        if nz > 1:
            v[0] += 2.0 * cz / (hinz * hinz) - 1.0 * az / hinz
        v[1] = -cx / (hinx * hinx)
        v[2] = -cy / (hiny * hiny)
        v[3] = -cz / (hinz * hinz)
        v[4] = -cx / (hinx * hinx) + ax / hinx
        v[5] = -cy / (hiny * hiny) + ay / hiny
        v[6] = -cz / (hinz * hinz) + az / hinz
        return v
    raise ValueError(opt.problem)


def build_matrix(opt, comm=0, rank=0, nprocs=1):
    """Rank (p,q,r) = (id % P, (id / P) % Q, id / (P*Q))   (test/ij.c:9693-9695)."""
    P, Q, R = opt.P if opt.P else (1, nprocs, 1)     # test/ij.c BuildParLaplacian: P = 1, Q = num_procs, R = 1
    if P * Q * R != nprocs:
        raise ValueError("P*Q*R must equal the number of ranks")
    p, q, r = rank % P, (rank // P) % Q, rank // (P * Q)
    kind = {"laplacian": "7pt", "27pt": "27pt", "difconv": "difconv"}[opt.problem]
    nx, ny, nz = opt.n
    return B.laplacian(nx, ny, nz, P, Q, R, p, q, r, comm=comm, values=stencil_values(opt), kind=kind)


class HypreRand:
    """utilities/random.c (Park-Miller), used by -rhsrand."""

    def __init__(self, seed):
        m = 2147483647
        self.seed = min(max(seed, 1), m - 1)

    def next(self):
        a, m, q, r = 16807, 2147483647, 127773, 2836
        high, low = divmod(self.seed, q)
        test = a * low - r * high
        self.seed = test if test > 0 else test + m
        return self.seed / m


def build_rhs_host(opt, A, rank=0, allreduce=None):
    """(b, x0) as numpy arrays for this rank (test/ij.c:3465-3600)."""
    n = A.contents.diag.contents.num_rows
    if opt.rhs == "one":
        return np.ones(n), np.zeros(n)
    if opt.rhs == "rand":
        # HYPRE_ParVectorSetRandomValues(b, 22775): seed * (rank + 1) on every rank, then normalise
        rng = HypreRand(22775 * (rank + 1))
        b = np.array([2.0 * rng.next() - 1.0 for _ in range(n)])
        # the reference accumulates the dot product serially
        nrm2 = 0.0
        for v in b:
            nrm2 += v * v
        if allreduce is not None:
            nrm2 = allreduce(nrm2)
        return b * (1.0 / np.sqrt(nrm2)), np.zeros(n)
    if opt.rhs == "xisone":
        return None, np.zeros(n)      # b = A * ones is formed by the caller (needs a product)
    raise ValueError(opt.rhs)


def create_amg(opt, memory_location=DEVICE):
    """test/ij.c:4440-4660: the setter sequence of solver 0 / the preconditioner."""
    L = B.load_library()
    s = C.c_void_p()
    L.HYPRE_BoomerAMGCreate(C.byref(s))
    L.hypre_amd_BoomerAMGSetMemoryLocation(s, memory_location)
    L.hypre_amd_BoomerAMGSetNumThreads(s, opt.num_threads)
    L.HYPRE_BoomerAMGSetInterpType(s, opt.interp_type)
    L.HYPRE_BoomerAMGSetCoarsenType(s, opt.coarsen_type)
    L.HYPRE_BoomerAMGSetTol(s, opt.tol)
    L.HYPRE_BoomerAMGSetStrongThreshold(s, opt.strong_threshold)
    L.HYPRE_BoomerAMGSetMaxCoarseSize(s, opt.coarse_threshold)
    L.HYPRE_BoomerAMGSetTruncFactor(s, opt.trunc_factor)
    L.HYPRE_BoomerAMGSetPMaxElmts(s, opt.P_max_elmts)
    L.HYPRE_BoomerAMGSetCycleType(s, opt.cycle_type)
    L.HYPRE_BoomerAMGSetFCycle(s, opt.fcycle)
    L.HYPRE_BoomerAMGSetNumSweeps(s, opt.num_sweeps)
    if opt.relax_type > -1:
        L.HYPRE_BoomerAMGSetRelaxType(s, opt.relax_type)
    if opt.relax_down > -1:
        L.HYPRE_BoomerAMGSetCycleRelaxType(s, opt.relax_down, 1)
    if opt.relax_up > -1:
        L.HYPRE_BoomerAMGSetCycleRelaxType(s, opt.relax_up, 2)
    if opt.relax_coarse > -1:
        L.HYPRE_BoomerAMGSetCycleRelaxType(s, opt.relax_coarse, 3)
    L.HYPRE_BoomerAMGSetRelaxOrder(s, opt.relax_order)
    L.HYPRE_BoomerAMGSetRelaxWt(s, opt.relax_wt)
    L.HYPRE_BoomerAMGSetOuterWt(s, opt.outer_wt)
    L.HYPRE_BoomerAMGSetMaxLevels(s, opt.max_levels)
    L.HYPRE_BoomerAMGSetMaxRowSum(s, opt.max_row_sum)
    L.HYPRE_BoomerAMGSetMaxIter(s, opt.mg_max_iter)
    L.HYPRE_BoomerAMGSetKeepTranspose(s, opt.keep_transpose)
    B.check()
    return s
