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
        self.problem = "laplacian"    # laplacian | 27pt | difconv | rotate
        self.alpha = 1.0              # -alpha (rotate: angle in degrees; test/ij.c:11147)
        self.eps = None               # -eps   (rotate: anisotropy, default 0; vardifconv: diffusion scale, default 1)
        self.sys_num_fun = 1          # -sysL <num functions>: systems version of the 7-point operator
        self.c = (1.0, 1.0, 1.0)      # -c cx cy cz
        self.a = (1.0, 1.0, 1.0)      # -a ax ay az (difconv)
        self.solver = 0               # 0 AMG, 1 AMG-PCG, 2 DS-PCG (diagonal scaling), 3 AMG-GMRES, 4 DS-GMRES
        self.neg_a = 0                # -negA 1: solve with -A (test/ij.c:4014-4017)
        self.num_components = 1       # -nc: columns of b and x (multivectors; test/ij.c:874-878, 3400-3404)
        self.k_dim = 5                # -k (GMRES restart length, test/ij.c:1731)
        self.flex = 0                 # -flex (flexible PCG: Polak-Ribiere beta)
        self.rhs = "one"              # one (-rhsisone default) | rand (-rhsrand) | xisone
        self.fromfile = None          # -fromfile <name>: matrix from IJ text files <name>.<rank %05d>
        self.rhsfromfile = None       # -rhsfromfile <name>: right-hand side from IJ text files
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
        self.num_functions = 1        # -nf (systems of PDEs, unknown approach)
        self.filter_functions = 0     # -ff
        self.ns_down = self.ns_up = self.ns_coarse = -1   # -ns_down / -ns_up / -ns_coarse
        self.level_w = None           # -wl  value level
        self.level_ow = None          # -owl value level
        self.mixed = False            # fp32 matrix values inside the cycle (extension)
        self.cheby_order = 2          # -cheby_order    (test/ij.c:330-336 defaults)
        self.cheby_eig_est = 10       # -cheby_eig_est
        self.cheby_variant = 0        # -cheby_variant
        self.cheby_scale = 1          # -cheby_scale
        self.cheby_fraction = 0.3     # -cheby_fraction
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
    """Rank (p,q,r) = (id % P, (id / P) % Q, id / (P*Q))   (test/ij.c:9693-9695); -negA 1: the operator times -1
    (test/ij.c:4014-4017 hypre_ParCSRMatrixScale(A, -1): test/TEST_ij/posneg.jobs)."""
    A = _build_matrix(opt, comm=comm, rank=rank, nprocs=nprocs)
    if opt.neg_a:
        for blk in (A.contents.diag.contents, A.contents.offd.contents):
            if blk.num_nonzeros > 0:
                np.ctypeslib.as_array(blk.data, shape=(blk.num_nonzeros,))[:] *= -1.0
    return A


def _build_matrix(opt, comm=0, rank=0, nprocs=1):
    if opt.fromfile:
        return read_matrix(opt.fromfile, comm=comm)
    P, Q, R = opt.P if opt.P else (1, nprocs, 1)     # test/ij.c BuildParLaplacian: P = 1, Q = num_procs, R = 1
    if P * Q * R != nprocs:
        raise ValueError("P*Q*R must equal the number of ranks")
    p, q, r = rank % P, (rank // P) % Q, rank // (P * Q)
    if opt.problem == "rotate":
        # test/ij.c:11130-11240 BuildParRotate7pt: a 2-D problem, -n and -P read two values each
        if R != 1:
            raise ValueError("-rotate is two-dimensional: R must be 1")
        A = B.load_library().GenerateRotate7pt(comm, opt.n[0], opt.n[1], P, Q, p, q, opt.alpha,
                                               0.0 if opt.eps is None else opt.eps)
        B.check()
        return A
    if opt.problem == "vardifconv":
        # test/ij.c:11260-11370 BuildParVarDifConv; the right-hand side it returns is all ones (build_rhs_host)
        nx, ny, nz = opt.n
        A = B.load_library().GenerateVarDifConv(comm, nx, ny, nz, P, Q, R, p, q, r, 1.0 if opt.eps is None else opt.eps, None)
        B.check()
        return A
    kind = {"laplacian": "7pt", "27pt": "27pt", "difconv": "difconv"}[opt.problem]
    nx, ny, nz = opt.n
    if opt.sys_num_fun > 1:
        # test/ij.c:9718-9870: coupling matrix of -sysL_opt 0
        mtrx = {2: [2.0, 1.0, 1.0, 2.0], 3: [2.0, 1.0, 0.0, 1.0, 2.0, 1.0, 0.0, 1.0, 2.0]}.get(opt.sys_num_fun)
        if mtrx is None or opt.problem != "laplacian":
            raise ValueError("-sysL supports 2 or 3 functions of the 7-point operator")
        L = B.load_library()
        vals = np.ascontiguousarray(stencil_values(opt), dtype=np.float64)
        m = np.ascontiguousarray(mtrx, dtype=np.float64)
        A = L.GenerateSysLaplacian(comm, nx, ny, nz, P, Q, R, p, q, r, opt.sys_num_fun,
                                   m.ctypes.data_as(C.POINTER(C.c_double)), vals.ctypes.data_as(C.POINTER(C.c_double)))
        B.check()
        return A
    return B.laplacian(nx, ny, nz, P, Q, R, p, q, r, comm=comm, values=stencil_values(opt), kind=kind)


def read_matrix(name, comm=0):
    """test/ij.c:2824-2833 (build_matrix_type -1): HYPRE_IJMatrixRead + GetObject; the IJ shell is dropped and
    the ParCSR matrix lives on (host memory)."""
    L = B.load_library()
    ij = C.POINTER(B.IJMatrix)()
    L.HYPRE_IJMatrixRead(name.encode(), comm, B.HYPRE_PARCSR, C.byref(ij))
    B.check()
    obj = C.c_void_p()
    L.HYPRE_IJMatrixGetObject(ij, C.byref(obj))
    A = C.cast(obj, C.POINTER(B.ParCSRMatrix))
    ij.contents.object = None          # keep the matrix, free the shell
    L.HYPRE_IJMatrixDestroy(ij)
    return A


def read_vector(name, comm=0):
    """test/ij.c:3406-3432 (build_rhs_type 0): the local slice as a numpy array."""
    L = B.load_library()
    ij = C.POINTER(B.IJVector)()
    L.HYPRE_IJVectorRead(name.encode(), comm, B.HYPRE_PARCSR, C.byref(ij))
    B.check()
    obj = C.c_void_p()
    L.HYPRE_IJVectorGetObject(ij, C.byref(obj))
    vals = B.parvec_to_numpy(C.cast(obj, C.POINTER(B.ParVector)))
    L.HYPRE_IJVectorDestroy(ij)
    return vals


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
    if opt.rhsfromfile:
        b = read_vector(opt.rhsfromfile, comm=A.contents.comm)
        if len(b) != n:
            raise ValueError("right-hand side file holds %d entries for %d local rows" % (len(b), n))
        return b, np.zeros(n)
    if opt.problem == "vardifconv":
        # test/ij.c:2877-2879, 3856-3887: b from the generator (ones), random initial guess seeded with the rank
        rng = HypreRand(rank)
        return np.ones(n), np.array([rng.next() for _ in range(n)])
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
    L.HYPRE_BoomerAMGSetNumFunctions(s, opt.num_functions)
    L.HYPRE_BoomerAMGSetFilterFunctions(s, opt.filter_functions)
    for k, sweeps in ((1, opt.ns_down), (2, opt.ns_up), (3, opt.ns_coarse)):
        if sweeps > -1:
            L.HYPRE_BoomerAMGSetCycleNumSweeps(s, sweeps, k)
    if opt.level_w is not None:
        L.HYPRE_BoomerAMGSetLevelRelaxWt(s, opt.level_w[0], opt.level_w[1])
    if opt.level_ow is not None:
        L.HYPRE_BoomerAMGSetLevelOuterWt(s, opt.level_ow[0], opt.level_ow[1])
    if opt.mixed:
        L.hypre_amd_BoomerAMGSetMixedPrecision(s, 1)
    L.HYPRE_BoomerAMGSetChebyOrder(s, opt.cheby_order)
    L.HYPRE_BoomerAMGSetChebyFraction(s, opt.cheby_fraction)
    L.HYPRE_BoomerAMGSetChebyEigEst(s, opt.cheby_eig_est)
    L.HYPRE_BoomerAMGSetChebyVariant(s, opt.cheby_variant)
    L.HYPRE_BoomerAMGSetChebyScale(s, opt.cheby_scale)
    B.check()
    return s


# ---------------------------------------------------------------------------
# command line of the reference driver (test/ij.c:521-2260), in-scope subset
# ---------------------------------------------------------------------------
# flag -> (attribute, converter, number of values); tuples of floats/ints for multi-valued flags
_VALUE_FLAGS = {
    "-solver": ("solver", int, 1), "-rlx": ("relax_type", int, 1), "-rlx_down": ("relax_down", int, 1),
    "-rlx_up": ("relax_up", int, 1), "-rlx_coarse": ("relax_coarse", int, 1), "-ns": ("num_sweeps", int, 1),
    "-ns_down": ("ns_down", int, 1), "-ns_up": ("ns_up", int, 1), "-ns_coarse": ("ns_coarse", int, 1),
    "-w": ("relax_wt", float, 1), "-ow": ("outer_wt", float, 1), "-CF": ("relax_order", int, 1),
    "-mu": ("cycle_type", int, 1), "-th": ("strong_threshold", float, 1), "-mxrs": ("max_row_sum", float, 1),
    "-tr": ("trunc_factor", float, 1), "-Pmx": ("P_max_elmts", int, 1), "-interptype": ("interp_type", int, 1),
    "-tol": ("tol", float, 1), "-max_iter": ("max_iter", int, 1), "-mg_max_iter": ("mg_max_iter", int, 1),
    "-mxl": ("max_levels", int, 1), "-coarse_th": ("coarse_threshold", int, 1), "-keepT": ("keep_transpose", int, 1),
    "-precon_cycles": ("precon_cycles", int, 1), "-k": ("k_dim", int, 1), "-nc": ("num_components", int, 1), "-negA": ("neg_a", int, 1), "-nf": ("num_functions", int, 1), "-flex": ("flex", int, 1),
    "-alpha": ("alpha", float, 1), "-eps": ("eps", float, 1), "-sysL": ("sys_num_fun", int, 1), "-ff": ("filter_functions", int, 1),
    "-cheby_order": ("cheby_order", int, 1), "-cheby_eig_est": ("cheby_eig_est", int, 1),
    "-cheby_variant": ("cheby_variant", int, 1), "-cheby_scale": ("cheby_scale", int, 1),
    "-cheby_fraction": ("cheby_fraction", float, 1),
    "-n": ("n", int, 3), "-P": ("P", int, 3), "-c": ("c", float, 3), "-a": ("a", float, 3),
    # extensions of this driver (no reference counterpart)
    "-amd_threads": ("num_threads", int, 1),
}
_SWITCH_FLAGS = {
    "-laplacian": ("problem", "laplacian"), "-27pt": ("problem", "27pt"), "-difconv": ("problem", "difconv"),
    "-rotate": ("problem", "rotate"), "-vardifconv": ("problem", "vardifconv"), "-rhsrand": ("rhs", "rand"), "-rhsisone": ("rhs", "one"), "-xisone": ("rhs", "xisone"),
    "-pmis": ("coarsen_type", 8), "-pmis1": ("coarsen_type", 9), "-hmis": ("coarsen_type", 10),
    "-fmg": ("fcycle", 1), "-amd_mixed": ("mixed", True),
}


def parse_cli(argv):
    """Reference `ij` flags -> IJOptions.  Flags outside the scope of this library are an error, never
    silently dropped."""
    opt = IJOptions()
    i = 0
    while i < len(argv):
        flag = argv[i]
        if flag in _SWITCH_FLAGS:
            attr, val = _SWITCH_FLAGS[flag]
            setattr(opt, attr, val)
            i += 1
        elif flag in _VALUE_FLAGS:
            attr, conv, count = _VALUE_FLAGS[flag]
            vals = [conv(v) for v in argv[i + 1:i + 1 + count]]
            if len(vals) != count:
                raise SystemExit("ij: %s needs %d value(s)" % (flag, count))
            setattr(opt, attr, vals[0] if count == 1 else tuple(vals))
            i += 1 + count
        elif flag in ("-fromfile", "-rhsfromfile"):
            if i + 1 >= len(argv):
                raise SystemExit("ij: %s needs a file name" % flag)
            setattr(opt, flag[1:], argv[i + 1])
            i += 2
        elif flag == "-rap":
            # test/ij.c:2157-2161 rap2: 0 = the Galerkin triple product this library builds (the default)
            if i + 1 >= len(argv) or int(argv[i + 1]) != 0:
                raise SystemExit("ij: -rap 1 (two-product coarse operator) is outside the scope of this driver")
            i += 2
        elif flag in ("-wl", "-owl"):
            val, lev = float(argv[i + 1]), int(argv[i + 2])
            if lev > -1:        # test/ij.c:4543-4550 applies the level weight only for level > -1
                setattr(opt, "level_w" if flag == "-wl" else "level_ow", (val, lev))
            i += 3
        else:
            raise SystemExit("ij: option %s is outside the scope of this driver" % flag)
    if opt.solver not in (0, 1, 2, 3, 4):
        raise SystemExit("ij: -solver %d is outside the scope of this driver (0 AMG, 1 AMG-PCG, 2 DS-PCG, 3 AMG-GMRES, 4 DS-GMRES)" % opt.solver)
    if opt.num_components < 1 or (opt.num_components > 1 and (opt.solver not in (2, 4) or opt.rhs != "one" or opt.rhsfromfile)):
        # test/ij.c:3400-3404 takes several components with the constant right-hand sides only; of the solvers that accept
        # multivectors (test/TEST_ij/vector.jobs) this driver has the diagonally scaled PCG
        raise SystemExit("ij: -nc %d needs -solver 2 or 4 and -rhsisone in this driver" % opt.num_components)
    if opt.interp_type not in (6, 3):
        raise SystemExit("ij: -interptype %d is outside the scope of this driver (6 ext+i, 3 direct)" % opt.interp_type)
    smoothers = (-1, 0, 3, 4, 6, 7, 8, 11, 12, 13, 14, 15, 16, 17, 18, 88, 89)
    for name in ("relax_type", "relax_down", "relax_up"):
        if getattr(opt, name) not in smoothers:
            raise SystemExit("ij: smoother %d is outside the scope of this driver" % getattr(opt, name))
    if opt.relax_coarse not in smoothers + (9, 19, 98, 99):
        raise SystemExit("ij: coarse solver %d is outside the scope of this driver" % opt.relax_coarse)
    weights = [opt.relax_wt, opt.outer_wt] + [w[0] for w in (opt.level_w, opt.level_ow) if w is not None]
    if any(w < 0 for w in weights):
        raise SystemExit("ij: negative (automatically estimated) relaxation weights are outside the scope of this driver")
    return opt


def run(opt, comm=0, rank=0, nprocs=1, allreduce=None, out=None):
    """test/ij.c: build the problem, set up, solve on the device, print the driver's closing lines."""
    import sys
    out = out or sys.stdout
    L = B.load_library()
    A = build_matrix(opt, comm=comm, rank=rank, nprocs=nprocs)
    if opt.solver in (2, 4):
        return run_ds_pcg(opt, A, comm=comm, rank=rank, allreduce=allreduce, out=out)
    s = create_amg(opt, memory_location=DEVICE)
    L.HYPRE_BoomerAMGSetup(s, A, None, None)
    B.check()
    L.hypre_ParCSRMatrixMigrate(A, DEVICE)
    Am = A.contents
    first, nglob = int(Am.row_starts[0]), int(Am.global_num_rows)
    b, x0 = build_rhs_host(opt, A, rank=rank, allreduce=allreduce)
    dx = B.parvec_from_numpy(x0, comm=comm, global_size=nglob, first=first)
    if b is None:
        ones = B.parvec_from_numpy(np.ones(len(x0)), comm=comm, global_size=nglob, first=first)
        db = B.parvec_from_numpy(np.zeros(len(x0)), comm=comm, global_size=nglob, first=first)
        L.hypre_ParCSRMatrixMatvec(1.0, A, ones, 0.0, db)
    else:
        db = B.parvec_from_numpy(b, comm=comm, global_size=nglob, first=first)
    its, rel = C.c_int(), C.c_double()
    lines = []
    if opt.solver == 0:
        # ||b - A x0|| for the average convergence factor (par_amg_solve.c:237-256, 347-355)
        r0 = B.parvec_from_numpy(np.zeros(len(x0)), comm=comm, global_size=nglob, first=first)
        L.hypre_ParCSRMatrixMatvecOutOfPlace(-1.0, A, dx, 1.0, db, r0)
        L.hypre_ParVectorInnerProd.restype = C.c_double
        nrm0 = float(np.sqrt(L.hypre_ParVectorInnerProd(r0, r0)))
        nrmb = float(np.sqrt(L.hypre_ParVectorInnerProd(db, db)))
        L.HYPRE_BoomerAMGSolve(s, A, db, dx)
        L.HYPRE_BoomerAMGGetNumIterations(s, C.byref(its))
        L.HYPRE_BoomerAMGGetFinalRelativeResidualNorm(s, C.byref(rel))
        g, o, cyc = C.c_double(), C.c_double(), C.c_double()
        L.hypre_amd_BoomerAMGGetComplexities(s, C.byref(g), C.byref(o))
        L.hypre_amd_BoomerAMGGetCycleOpCount(s, C.byref(cyc))
        resid = rel.value * (nrmb if nrmb != 0.0 else 1.0)
        conv = (resid / nrm0) ** (1.0 / its.value) if its.value > 0 and nrm0 != 0.0 else 1.0
        nnz0 = float(Am.d_num_nonzeros)
        lines += ["", " Average Convergence Factor = %f" % conv, "",
                  "     Complexity:    grid = %f" % g.value,
                  "                operator = %f" % o.value,
                  "                   cycle = %f" % (cyc.value / nnz0 if nnz0 else 0.0), "", "", "",
                  "BoomerAMG Iterations = %d" % its.value,
                  "Final Relative Residual Norm = %e" % rel.value, ""]
    elif opt.solver == 3:
        its.value, rel.value = solve_gmres(opt, s, A, db, dx, comm=comm)
        lines += ["", "GMRES Iterations = %d" % its.value, "Final GMRES Relative Residual Norm = %e" % rel.value, ""]
    else:
        L.HYPRE_BoomerAMGSetTol(s, 0.0)
        L.HYPRE_BoomerAMGSetMaxIter(s, opt.precon_cycles)
        pcg = C.c_void_p()
        L.HYPRE_ParCSRPCGCreate(comm, C.byref(pcg))
        L.HYPRE_PCGSetTol(pcg, opt.tol)
        L.HYPRE_PCGSetMaxIter(pcg, opt.max_iter)
        L.HYPRE_PCGSetTwoNorm(pcg, opt.two_norm)
        L.HYPRE_PCGSetFlex(pcg, opt.flex)
        L.HYPRE_PCGSetPrecond(pcg, C.cast(L.HYPRE_BoomerAMGSolve, C.c_void_p), None, s)
        L.HYPRE_ParCSRPCGSetup(pcg, A, db, dx)
        L.HYPRE_ParCSRPCGSolve(pcg, A, db, dx)
        L.HYPRE_PCGGetNumIterations(pcg, C.byref(its))
        L.HYPRE_PCGGetFinalRelativeResidualNorm(pcg, C.byref(rel))
        L.HYPRE_ParCSRPCGDestroy(pcg)
        lines += ["", "Iterations = %d" % its.value, "Final Relative Residual Norm = %e" % rel.value, ""]
    L.HYPRE_ClearError(256)          # HYPRE_ERROR_CONV is reported through the iteration count, as the driver does
    B.check()
    L.HYPRE_BoomerAMGDestroy(s)
    if rank == 0:
        out.write("\n".join(lines) + "\n")
        out.flush()
    return its.value, rel.value


def solve_ds_pcg(opt, A, db, dx, comm=0):
    """test/ij.c:5007-5027, 5179-5191: PCG with the diagonal scaling preconditioner (solver 2); db and dx may be
    multivectors (-nc N)."""
    L = B.load_library()
    pcg = C.c_void_p()
    L.HYPRE_ParCSRPCGCreate(comm, C.byref(pcg))
    L.HYPRE_PCGSetTol(pcg, opt.tol)
    L.HYPRE_PCGSetMaxIter(pcg, opt.max_iter)
    L.HYPRE_PCGSetTwoNorm(pcg, opt.two_norm)
    L.HYPRE_PCGSetFlex(pcg, opt.flex)
    L.HYPRE_PCGSetPrecond(pcg, C.cast(L.HYPRE_ParCSRDiagScale, C.c_void_p), C.cast(L.HYPRE_ParCSRDiagScaleSetup, C.c_void_p), None)
    L.HYPRE_ParCSRPCGSetup(pcg, A, db, dx)
    L.HYPRE_ParCSRPCGSolve(pcg, A, db, dx)
    its, rel = C.c_int(), C.c_double()
    L.HYPRE_PCGGetNumIterations(pcg, C.byref(its))
    L.HYPRE_PCGGetFinalRelativeResidualNorm(pcg, C.byref(rel))
    L.HYPRE_ParCSRPCGDestroy(pcg)
    return its.value, rel.value


def solve_ds_gmres(opt, A, db, dx, comm=0):
    """test/ij.c:6710-6727, 6932-6942: GMRES(k) with the diagonal scaling preconditioner (solver 4); multivectors as above."""
    L = B.load_library()
    g = C.c_void_p()
    L.HYPRE_ParCSRGMRESCreate(comm, C.byref(g))
    L.HYPRE_GMRESSetKDim(g, opt.k_dim)
    L.HYPRE_GMRESSetMaxIter(g, opt.max_iter)
    L.HYPRE_GMRESSetTol(g, opt.tol)
    L.HYPRE_GMRESSetAbsoluteTol(g, 0.0)
    L.HYPRE_GMRESSetPrecond(g, C.cast(L.HYPRE_ParCSRDiagScale, C.c_void_p), C.cast(L.HYPRE_ParCSRDiagScaleSetup, C.c_void_p), None)
    L.HYPRE_ParCSRGMRESSetup(g, A, db, dx)
    L.HYPRE_ParCSRGMRESSolve(g, A, db, dx)
    its, rel = C.c_int(), C.c_double()
    L.HYPRE_GMRESGetNumIterations(g, C.byref(its))
    L.HYPRE_GMRESGetFinalRelativeResidualNorm(g, C.byref(rel))
    L.HYPRE_ParCSRGMRESDestroy(g)
    return its.value, rel.value


def run_ds_pcg(opt, A, comm=0, rank=0, allreduce=None, out=None):
    """`ij -solver 2 | 4 [-nc N]` on the device: no hierarchy; the N columns of b (test/ij.c:3483-3512: the same values in
    every component) and of the zero initial guess as multivectors."""
    L = B.load_library()
    L.hypre_ParCSRMatrixMigrate(A, DEVICE)
    Am = A.contents
    first, nglob = int(Am.row_starts[0]), int(Am.global_num_rows)
    b, x0 = build_rhs_host(opt, A, rank=rank, allreduce=allreduce)
    nv = opt.num_components
    if b is None:
        ones = B.parvec_from_numpy(np.ones(len(x0)), comm=comm, global_size=nglob, first=first)
        db1 = B.parvec_from_numpy(np.zeros(len(x0)), comm=comm, global_size=nglob, first=first)
        L.hypre_ParCSRMatrixMatvec(1.0, A, ones, 0.0, db1)
        b = B.parvec_to_numpy(db1)
        L.hypre_ParVectorDestroy(ones); L.hypre_ParVectorDestroy(db1)
    if nv > 1:
        db = B.parmultivec_from_numpy(np.repeat(b[:, None], nv, axis=1), comm=comm, global_size=nglob, first=first)
        dx = B.parmultivec_from_numpy(np.repeat(x0[:, None], nv, axis=1), comm=comm, global_size=nglob, first=first)
    else:
        db = B.parvec_from_numpy(b, comm=comm, global_size=nglob, first=first)
        dx = B.parvec_from_numpy(x0, comm=comm, global_size=nglob, first=first)
    its, rel = (solve_ds_gmres if opt.solver == 4 else solve_ds_pcg)(opt, A, db, dx, comm=comm)
    L.HYPRE_ClearError(256)
    B.check()
    L.hypre_ParVectorDestroy(db); L.hypre_ParVectorDestroy(dx)
    if rank == 0:
        tag = "GMRES " if opt.solver == 4 else ""
        out.write("\n".join(["", "%sIterations = %d" % (tag, its), "Final %sRelative Residual Norm = %e" % (tag, rel), ""]) + "\n")
        out.flush()
    return its, rel


def solve_gmres(opt, amg, A, b, x, comm=0):
    """test/ij.c:6715-6790, 6919-6925, 7190-7215: AMG-GMRES (solver 3)."""
    L = B.load_library()
    L.HYPRE_BoomerAMGSetTol(amg, 0.0)
    L.HYPRE_BoomerAMGSetMaxIter(amg, opt.precon_cycles)
    g = C.c_void_p()
    L.HYPRE_ParCSRGMRESCreate(comm, C.byref(g))
    L.HYPRE_GMRESSetKDim(g, opt.k_dim)
    L.HYPRE_GMRESSetMaxIter(g, opt.max_iter)
    L.HYPRE_GMRESSetTol(g, opt.tol)
    L.HYPRE_GMRESSetAbsoluteTol(g, 0.0)
    L.HYPRE_GMRESSetPrecond(g, C.cast(L.HYPRE_BoomerAMGSolve, C.c_void_p), None, amg)
    L.HYPRE_ParCSRGMRESSetup(g, A, b, x)
    L.HYPRE_ParCSRGMRESSolve(g, A, b, x)
    its, rel = C.c_int(), C.c_double()
    L.HYPRE_GMRESGetNumIterations(g, C.byref(its))
    L.HYPRE_GMRESGetFinalRelativeResidualNorm(g, C.byref(rel))
    L.HYPRE_ParCSRGMRESDestroy(g)
    return its.value, rel.value


def main(argv=None):
    """`python -m hypre_amd.ij <reference ij flags>`; under torch.distributed.run one rank per process
    (ranks may share a GPU: the halo then travels over gloo, staged through the host)."""
    import os
    import sys
    argv = sys.argv[1:] if argv is None else argv
    opt = parse_cli(argv)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1:
        run(opt)
        return 0
    import torch
    import torch.distributed as dist
    from . import distributed
    dist.init_process_group(backend="gloo")
    rank = dist.get_rank()

    def allsum(v):
        t = torch.tensor([v], dtype=torch.float64)
        dist.all_reduce(t)
        return float(t.item())

    comm = distributed.create_callback_comm(dist, rank, world)
    run(opt, comm=comm, rank=rank, nprocs=world, allreduce=allsum)
    dist.barrier()
    dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
