"""Worker of the multi-process CPU tests (gloo): run under torch.distributed.run.
Builds the distributed problem, runs the library's HOST setup on every rank, gathers the
hierarchy on rank 0 and lets the CPU oracle solve it.  Prints one JSON line on rank 0."""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def main():
    import torch.distributed as dist
    from hypre_amd import binding as B, ij, distributed
    import pyoracle as O

    spec = json.loads(sys.argv[1])
    # {"batch": [case, ...]}: several cases behind one launch (process start-up and the torch import dominate a case)
    cases = spec["batch"] if "batch" in spec else [spec]
    # a rank that stops making progress dumps its Python stack and exits instead of hanging the suite
    import faulthandler
    faulthandler.dump_traceback_later(int(os.environ.get("HYPRE_AMD_TEST_WATCHDOG", "300")), exit=True)
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    L = B.load_library()
    # transport "staged": device-buffer communicator (the library's production halo flow: pack, event, exchange enqueued
    # on the communication stream, event; device all-reduces) whose bytes travel over gloo — ranks may share the card
    if world > 1 and spec.get("transport") == "staged":
        comm = distributed.create_stream_staged_comm(dist, rank, world)
    else:
        comm = distributed.create_callback_comm(dist, rank, world) if world > 1 else 0     # 0: hypre_MPI_COMM_WORLD of one rank
    if world > 1 and L.hypre_amd_CommSelfTest(comm, 4099) != 0:
        raise SystemExit("communicator self-test failed on rank %d" % rank)
    for case in cases:
        run_case(case, L, B, ij, O, dist, comm, rank, world)
    dist.barrier()
    dist.destroy_process_group()


def _blocks(B, m):
    """arrays of one rank's share of a ParCSR matrix, stored transposes included"""
    out = list(B.csr_to_arrays(m.diag)) + list(B.csr_to_arrays(m.offd))
    nco = m.offd.contents.num_cols
    out.append(np.array([m.col_map_offd[k] for k in range(nco)], dtype=np.int64))
    for t in (m.diagT, m.offdT):
        if t:
            out += list(B.csr_to_arrays(t))
    return out


def compare_setup(case, L, B, ij, O, dist, comm, rank, world):
    """HYPRE_BoomerAMGSetup twice on the same distributed problem, hierarchy in device memory: the distributed levels worked
    on by the host routines (OpenMP loops) and by the device kernels.  Every array of every level of this rank's share must
    be identical: operators (both blocks, ghost column maps), interpolation operators and their stored transposes, C/F
    markers, smoother diagonals."""
    opt = ij.IJOptions(**{k: (tuple(v) if isinstance(v, list) else v) for k, v in case["options"].items()})
    hier, counts = [], []
    for on in (0, 1):
        L.hypre_amd_SetSetupDeviceDist(on)
        L.hypre_amd_SetSetupDeviceRAP(1, int(case.get("min_rows", 50)))
        L.hypre_amd_SetSetupDeviceInterp(1 + int(case.get("rung", 0)))
        L.hypre_amd_SetSetupDeviceCoarsen(1)
        A = ij.build_matrix(opt, comm=comm, rank=rank, nprocs=world)
        if on and case.get("matrix_on_device"):
            L.hypre_ParCSRMatrixMigrate(A, B.HYPRE_MEMORY_DEVICE)
        s = ij.create_amg(opt, memory_location=B.HYPRE_MEMORY_DEVICE)
        if on and "decline" in case:
            # one rank's kernel "declines" a step once: all ranks repeat that step of that level with the host routine
            L.hypre_amd_SetupDistTestDecline(int(case["decline"][0]), int(case["decline"][1]), 1)
        L.HYPRE_BoomerAMGSetup(s, A, None, None)
        L.hypre_amd_SetupDistTestDecline(0, -1, 0)
        B.check()
        counts.append((L.hypre_amd_SetSetupDeviceCoarsen(-1), L.hypre_amd_SetSetupDeviceInterp(-1), L.hypre_amd_SetSetupDeviceRAP(-1, -1)))
        nl = L.hypre_amd_BoomerAMGGetNumLevels(s)
        lv = []
        for l in range(nl):
            Al = C.cast(L.hypre_amd_BoomerAMGGetA(s, l), C.POINTER(B.ParCSRMatrix)).contents
            lv.append(("A%d" % l, _blocks(B, Al)))
            if l < nl - 1:
                Pl = C.cast(L.hypre_amd_BoomerAMGGetP(s, l), C.POINTER(B.ParCSRMatrix)).contents
                lv.append(("P%d" % l, _blocks(B, Pl)))
            cfp = L.hypre_amd_BoomerAMGGetCFMarker(s, l)
            if cfp:
                ia = C.cast(cfp, C.POINTER(B.IntArray)).contents
                lv.append(("cf%d" % l, [B.fetch(ia.data, ia.size, np.int32, ia.memory_location)]))
            lp = L.hypre_amd_BoomerAMGGetL1Norms(s, l)
            if lp:
                v = C.cast(lp, C.POINTER(B.Vector)).contents
                lv.append(("l1_%d" % l, [B.fetch(v.data, v.size, np.float64, v.memory_location)]))
        hier.append(lv)
        L.HYPRE_BoomerAMGDestroy(s)
        B.check()
        L.hypre_ParCSRMatrixDestroy(A)
        B.check()
    L.hypre_amd_SetSetupDeviceDist(1)
    L.hypre_amd_SetSetupDeviceRAP(1, 20000)
    L.hypre_amd_SetSetupDeviceInterp(1)
    L.hypre_amd_SetSetupDeviceCoarsen(1)
    mismatch = None
    if len(hier[0]) != len(hier[1]):
        mismatch = "hierarchies of %d and %d pieces" % (len(hier[0]), len(hier[1]))
    else:
        for (n0, a0), (n1, a1) in zip(hier[0], hier[1]):
            if n0 != n1 or len(a0) != len(a1):
                mismatch = "%s / %s: %d and %d arrays" % (n0, n1, len(a0), len(a1))
                break
            for k, (x, y) in enumerate(zip(a0, a1)):
                if x.shape != y.shape or not np.array_equal(x, y):
                    where = int(np.flatnonzero(x != y)[0]) if x.shape == y.shape else -1
                    mismatch = "rank %d, %s array %d (shapes %s %s, first difference at %d)" % (rank, n0, k, x.shape, y.shape, where)
                    break
            if mismatch:
                break
    mine = dict(mismatch=mismatch, host_counts=counts[0], device_counts=counts[1], pieces=len(hier[1]),
                sizes=[int(a[1][0].shape[0]) - 1 for a in hier[1] if a[0].startswith("A")])
    parts = [None] * world if rank == 0 else None
    dist.gather_object(mine, parts, dst=0)
    if rank == 0:
        bad = [p["mismatch"] for p in parts if p["mismatch"]]
        out = {"name": case.get("name"), "setup_equal": not bad, "mismatch": bad[:3],
               "host_counts": [list(p["host_counts"]) for p in parts], "device_counts": [list(p["device_counts"]) for p in parts],
               "local_sizes": [p["sizes"] for p in parts]}
        print("RESULT " + json.dumps(out), flush=True)


def run_ds_pcg_case(case, opt, A, device, L, B, ij, O, dist, comm, rank, world):
    """`ij -solver 2 -nc N` (test/TEST_ij/vector.jobs): PCG with diagonal scaling on a multivector of N columns — no
    hierarchy.  The CPU oracle solves the gathered problem; with "device" the library solves it too, distributed, on the
    GPU (multivector products with one halo exchange for all columns, fused over the columns on the local block)."""
    nv = opt.num_components

    def allsum(v):
        import torch
        t = torch.tensor([v], dtype=torch.float64)
        dist.all_reduce(t)
        return float(t.item())

    b, x = ij.build_rhs_host(opt, A, rank=rank, allreduce=allsum)
    mine = dict(h=O.export_par(A), b=b, x=x)
    if device:
        Am = A.contents
        first, nglob = int(Am.row_starts[0]), int(Am.global_num_rows)
        L.hypre_ParCSRMatrixMigrate(A, B.HYPRE_MEMORY_DEVICE)
        db = B.parmultivec_from_numpy(np.repeat(b[:, None], nv, axis=1), comm=comm, global_size=nglob, first=first)
        dx = B.parmultivec_from_numpy(np.repeat(x[:, None], nv, axis=1), comm=comm, global_size=nglob, first=first)
        fused_before = L.hypre_amd_SpmvFusedMultivectorLaunches()
        its, rel = (ij.solve_ds_gmres if opt.solver == 4 else ij.solve_ds_pcg)(opt, A, db, dx, comm=comm)
        L.HYPRE_ClearError(256)
        B.check()
        mine.update(dev_its=its, dev_rel=rel, dev_x=B.parmultivec_to_numpy(dx),
                    fused=int(L.hypre_amd_SpmvFusedMultivectorLaunches() - fused_before))
        L.hypre_ParVectorDestroy(db); L.hypre_ParVectorDestroy(dx)
    parts = [None] * world if rank == 0 else None
    dist.gather_object(mine, parts, dst=0)
    if rank == 0:
        Ao = O.par_from_exports([p["h"] for p in parts])
        bg = np.concatenate([p["b"] for p in parts]); xg = np.concatenate([p["x"] for p in parts])
        X = np.repeat(xg[:, None], nv, axis=1)
        if opt.solver == 2:
            its, rel, conv = O.pcg_ds_multi(Ao, np.repeat(bg[:, None], nv, axis=1), X, tol=opt.tol, max_iter=opt.max_iter,
                                            two_norm=opt.two_norm)
        else:
            its, rel, conv = O.gmres_ds_multi(Ao, np.repeat(bg[:, None], nv, axis=1), X, tol=opt.tol, max_iter=opt.max_iter,
                                              k_dim=opt.k_dim)
        out = {"iterations": its, "rel_resid": rel}
        if device:
            Xd = np.concatenate([p["dev_x"] for p in parts])
            out.update(dev_iterations=parts[0]["dev_its"], dev_rel_resid=parts[0]["dev_rel"],
                       x_err=float(np.max(np.abs(Xd - X)) / np.max(np.abs(X))), fused=[p["fused"] for p in parts])
        if "name" in case:
            out["name"] = case["name"]
        print("RESULT " + json.dumps(out), flush=True)
    L.hypre_ParCSRMatrixDestroy(A)
    B.check()


def run_case(case, L, B, ij, O, dist, comm, rank, world):
    if case.get("compare_setup"):
        return compare_setup(case, L, B, ij, O, dist, comm, rank, world)
    opt = ij.IJOptions(**{k: (tuple(v) if isinstance(v, list) else v) for k, v in case["options"].items()})
    for name in ("fromfile", "rhsfromfile"):          # the reference's input files live beside the goldens
        if getattr(opt, name):
            setattr(opt, name, os.path.join(ROOT, "tests", "golden", "ij_files", getattr(opt, name)))
    A = ij.build_matrix(opt, comm=comm, rank=rank, nprocs=world)
    device = bool(case.get("device", 0))
    if opt.solver in (2, 4):
        return run_ds_pcg_case(case, opt, A, device, L, B, ij, O, dist, comm, rank, world)
    s = ij.create_amg(opt, memory_location=B.HYPRE_MEMORY_DEVICE if device else B.HYPRE_MEMORY_HOST)
    if "replicate" in case:
        L.hypre_amd_BoomerAMGSetReplicateThreshold(s, int(case["replicate"]))
    mixed = bool(case.get("mixed", 0))
    if mixed:
        L.hypre_amd_BoomerAMGSetMixedPrecision(s, 1)
    if "min_rows" in case:
        # distributed levels of at least that many rows per rank are set up by the device kernels (default 20000)
        L.hypre_amd_SetSetupDeviceRAP(1, int(case["min_rows"]))
        if case.get("matrix_on_device"):
            L.hypre_ParCSRMatrixMigrate(A, B.HYPRE_MEMORY_DEVICE)
    L.HYPRE_BoomerAMGSetup(s, A, None, None)
    B.check()
    device_levels = L.hypre_amd_SetSetupDeviceCoarsen(-1)
    if "min_rows" in case:
        L.hypre_amd_SetSetupDeviceRAP(1, 20000)
        if case.get("matrix_on_device"):
            L.hypre_ParCSRMatrixMigrate(A, B.HYPRE_MEMORY_HOST)
    g, o = C.c_double(), C.c_double()
    L.hypre_amd_BoomerAMGGetComplexities(s, C.byref(g), C.byref(o))

    def allsum(v):
        import torch
        t = torch.tensor([v], dtype=torch.float64)
        dist.all_reduce(t)
        return float(t.item())

    b, x = ij.build_rhs_host(opt, A, rank=rank, allreduce=allsum)
    mine = dict(h=O.export_solver(s), b=b, x=x)
    if device:
        # the product path: distributed solve on the GPU (two ranks may share one card in the
        # test; halo traffic goes through the callback communicator, staged over the host)
        Am = A.contents
        first, nglob = int(Am.row_starts[0]), int(Am.global_num_rows)
        L.hypre_ParCSRMatrixMigrate(A, B.HYPRE_MEMORY_DEVICE)
        if b is None:
            ones = B.parvec_from_numpy(np.ones(len(x)), comm=comm, global_size=nglob, first=first)
            db = B.parvec_from_numpy(np.zeros(len(x)), comm=comm, global_size=nglob, first=first)
            L.hypre_ParCSRMatrixMatvec(1.0, A, ones, 0.0, db)
        else:
            db = B.parvec_from_numpy(b, comm=comm, global_size=nglob, first=first)
        dx = B.parvec_from_numpy(x, comm=comm, global_size=nglob, first=first)
        # y = A x and z = A^T y through the distributed device products, for the oracle to check
        xt = np.cos(np.arange(first, first + len(x)) * 0.37)
        dxt = B.parvec_from_numpy(xt, comm=comm, global_size=nglob, first=first)
        dy = B.parvec_from_numpy(np.zeros(len(x)), comm=comm, global_size=nglob, first=first)
        dz = B.parvec_from_numpy(np.ones(len(x)), comm=comm, global_size=nglob, first=first)
        L.hypre_ParCSRMatrixMatvec(2.0, A, dxt, 0.0, dy)
        L.hypre_ParCSRMatrixMatvecT(1.0, A, dy, -0.5, dz)
        dot = L.hypre_ParVectorInnerProd(dxt, dy)
        # the transpose product twice: its unpack adds the neighbours' contributions in a fixed order (y += E * buf), so the
        # bits repeat
        dz2 = B.parvec_from_numpy(np.ones(len(x)), comm=comm, global_size=nglob, first=first)
        L.hypre_ParCSRMatrixMatvecT(1.0, A, dy, -0.5, dz2)
        zt_repeat = bool(np.array_equal(B.parvec_to_numpy(dz), B.parvec_to_numpy(dz2)))
        # multivectors (3 columns, stored column by column): every column through ONE halo exchange
        # (hypre_ParCSRCommPkgUpdateVecStarts), compared below with the single-vector products
        nloc, nv = len(x), 3
        Xm = np.stack([xt * (k + 1.0) + 0.1 * k for k in range(nv)], axis=1)

        def multivec(M):
            pv = B.parvec_from_numpy(np.ascontiguousarray(M.T).ravel(), comm=comm, global_size=nglob * nv, first=first * nv)
            pv.contents.global_size = nglob
            pv.contents.partitioning[0], pv.contents.partitioning[1] = first, first + nloc
            pv.contents.first_index, pv.contents.last_index = first, first + nloc - 1
            pv.contents.actual_local_size = nloc * nv
            v = pv.contents.local_vector.contents
            v.size, v.num_vectors, v.vecstride, v.idxstride = nloc, nv, nloc, 1
            return pv

        def columns(pv):
            v = pv.contents.local_vector.contents
            return B.fetch(v.data, nloc * nv, np.float64, v.memory_location).reshape(nv, nloc).T.copy()

        mx, my, mz = multivec(Xm), multivec(np.zeros((nloc, nv))), multivec(np.ones((nloc, nv)))
        L.hypre_ParCSRMatrixMatvec(2.0, A, mx, 0.0, my)
        L.hypre_ParCSRMatrixMatvecT(1.0, A, my, -0.5, mz)
        B.check()
        mv_y, mv_z = columns(my), columns(mz)
        # ... and the package is back in single-vector shape
        L.hypre_ParCSRMatrixMatvec(2.0, A, dxt, 0.0, dy)
        B.check()
        # the solve with timing events on every exchange (hypre_amd_CommSetTiming): how many there were per AMG level, how long
        # the transfers took and how much of that the compute stream spent waiting (bench.py --gpus N reports these)
        L.hypre_amd_CommSetTiming(1)
        if opt.solver == 0:
            L.HYPRE_BoomerAMGSolve(s, A, db, dx)
            its, rel = C.c_int(), C.c_double()
            L.HYPRE_BoomerAMGGetNumIterations(s, C.byref(its))
            L.HYPRE_BoomerAMGGetFinalRelativeResidualNorm(s, C.byref(rel))
        elif opt.solver == 3:
            its, rel = C.c_int(), C.c_double()
            its.value, rel.value = ij.solve_gmres(opt, s, A, db, dx, comm=comm)
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
            its, rel = C.c_int(), C.c_double()
            L.HYPRE_PCGGetNumIterations(pcg, C.byref(its))
            L.HYPRE_PCGGetFinalRelativeResidualNorm(pcg, C.byref(rel))
        B.check()
        MT = 32
        t_ex, t_ar = (C.c_int * MT)(), (C.c_int * MT)()
        t_eu, t_tu = (C.c_double * MT)(), (C.c_double * MT)()
        t_hu = (C.c_double * MT)()
        L.hypre_amd_CommExposedTimes(MT, t_ex, t_ar, t_eu, t_tu, t_hu)
        L.hypre_amd_CommSetTiming(0)
        B.check()
        mine.update(timed=dict(exchanges=sum(t_ex), allreduces=sum(t_ar), in_cycle=sum(t_ex[:MT - 1]), exposed_us=sum(t_eu), transfer_us=sum(t_tu), host_us=sum(t_hu),
                               worst=max((t_eu[k] - t_tu[k]) for k in range(MT))))
        mine.update(replicated_level=int(L.hypre_amd_BoomerAMGGetReplicatedLevel(s)), device_levels=int(device_levels))
        mine.update(dev_its=its.value, dev_rel=rel.value, dev_x=B.parvec_to_numpy(dx), xt=xt,
                    dev_y=B.parvec_to_numpy(dy), dev_z=B.parvec_to_numpy(dz), dev_dot=dot, zt_repeat=zt_repeat,
                    mv_x=Xm, mv_y=mv_y, mv_z=mv_z)
    parts = [None] * world if rank == 0 else None
    dist.gather_object(mine, parts, dst=0)
    if rank == 0:
        amg = O.amg_from_exports([p["h"] for p in parts], num_threads=opt.num_threads, mixed_precision=mixed)
        n = amg.A_levels[0].nrows
        xg = np.concatenate([p["x"] for p in parts])
        if parts[0]["b"] is None:
            bg = np.zeros(n)
            O.par_matvec(1.0, amg.A_outer or amg.A_levels[0], np.ones(n), 0.0, bg, bg)
        else:
            bg = np.concatenate([p["b"] for p in parts])
        out = {"grid": g.value, "operator": o.value, "levels": amg.c.num_levels,
               "sizes": [a.nrows for a in amg.A_levels]}
        if opt.solver == 0:
            its, rel, conv, hist = amg.solve(bg, xg, tol=opt.tol, max_iter=opt.mg_max_iter)
            out.update(iterations=its, rel_resid=rel, conv_factor=(hist[-1] / hist[0]) ** (1.0 / max(its, 1)))
        elif opt.solver == 3:
            its, rel, conv = amg.gmres(bg, xg, tol=opt.tol, max_iter=opt.max_iter, k_dim=opt.k_dim,
                                       precond_cycles=opt.precon_cycles)
            out.update(iterations=its, rel_resid=rel)
        else:
            its, rel, conv = amg.pcg(bg, xg, tol=opt.tol, max_iter=opt.max_iter, two_norm=opt.two_norm,
                                     precond_cycles=opt.precon_cycles, flex=opt.flex)
            out.update(iterations=its, rel_resid=rel)
        if device:
            A0 = amg.A_outer or amg.A_levels[0]        # mixed precision: products outside the cycle use the exact operator
            xt = np.concatenate([p["xt"] for p in parts])
            yr = np.zeros(n)
            O.par_matvec(2.0, A0, xt, 0.0, yr, yr)
            zr = np.ones(n)
            O.par_matvecT(1.0, A0, yr, -0.5, zr)
            yd = np.concatenate([p["dev_y"] for p in parts])
            zd = np.concatenate([p["dev_z"] for p in parts])
            xd = np.concatenate([p["dev_x"] for p in parts])
            out.update(replicated_level=parts[0]["replicated_level"], device_levels=min(p["device_levels"] for p in parts))
            out.update(timed=parts[0]["timed"])
            # multivector products, column by column, against the oracle
            Xg = np.concatenate([p["mv_x"] for p in parts]); Yg = np.concatenate([p["mv_y"] for p in parts])
            Zg = np.concatenate([p["mv_z"] for p in parts])
            mv_err = 0.0
            for k in range(Xg.shape[1]):
                yk = np.zeros(n)
                O.par_matvec(2.0, A0, Xg[:, k].copy(), 0.0, yk, yk)
                zk = np.ones(n)
                O.par_matvecT(1.0, A0, yk, -0.5, zk)
                mv_err = max(mv_err, float(np.max(np.abs(Yg[:, k] - yk)) / np.max(np.abs(yk))),
                             float(np.max(np.abs(Zg[:, k] - zk)) / np.max(np.abs(zk))))
            out.update(mv_err=mv_err, matvecT_repeats=all(p["zt_repeat"] for p in parts))
            out.update(dev_iterations=parts[0]["dev_its"], dev_rel_resid=parts[0]["dev_rel"],
                       matvec_err=float(np.max(np.abs(yd - yr)) / np.max(np.abs(yr))),
                       matvecT_err=float(np.max(np.abs(zd - zr)) / np.max(np.abs(zr))),
                       dot_err=float(abs(parts[0]["dev_dot"] - float(np.dot(xt, yr))) / abs(float(np.dot(xt, yr)))),
                       x_err=float(np.max(np.abs(xd - xg)) / np.max(np.abs(xg))))
        if "name" in case:
            out["name"] = case["name"]
        print("RESULT " + json.dumps(out), flush=True)
    L.HYPRE_BoomerAMGDestroy(s)
    B.check()                      # a free with the wrong memory space or a double free shows up here
    L.hypre_ParCSRMatrixDestroy(A)
    B.check()


if __name__ == "__main__":
    main()
