#!/usr/bin/env python
"""BoomerAMG V-cycle benchmark on synthetic 7-point Laplacians (BASELINE.json metric).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One step = one application of the hot path: one V(1,1) BoomerAMG cycle
(PMIS / extended+i (4 per row) / l1-Jacobi, fp64) from a zero initial guess, as PCG applies it
as a preconditioner, through HYPRE_BoomerAMGSolve of the C-ABI library.  Inputs
(matrix hierarchy, right-hand side) are resident in HBM before the timed region.
Rank 0 prints ONE JSON line.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
PARITY_TOL = 1.0e-11       # device V-cycle vs the oracle's on the same hierarchy, relative max-norm (fp64)
# the reference's own CPU path, 256^3 7-pt, 8 MPI ranks on the 8 cores of the survey container
REFERENCE_CPU_SURVEY = {"config": "ij -n 256 256 256 -P 2 2 2 -solver 1 -rlx 18 -pmis -interptype 6 -keepT 1; 8 MPI ranks, 8 cores",
                        "spmv_ms": 37.8, "spmv_GBps": 46.0, "pcg_s_per_iteration": 0.44, "pcg_iterations": 23,
                        "MDOF_per_s_per_pcg_iteration": 38.0,
                        "port_in_same_container": "profiles/r02_cpu_port_vs_reference_survey_container.json"}
# the reference's own GPU run of this problem (regression-perf files of test/TEST_bench, job #14:
# 256^3 7-pt, 1 GPU, PMIS / ext+i / l1-Jacobi, AMG-PCG to 1e-8).  OTHER hardware and the reference's device PMIS (other
# random numbers: 21 iterations there, 22 with the host routine's here): context for the solve and setup times below,
# not a baseline for the metric (which is the V-cycle rate).
REFERENCE_GPU_PUBLISHED = {"job": "test/TEST_bench/benchmark_ij.jobs:52 (#14)", "hardware": "1 GCD of AMD MI250X (tioga), ROCm 5.2",
                           "pcg_solve_s": 0.680926, "pcg_iterations": 21, "pcg_setup_s": 0.953187,
                           "source": "test/TEST_bench/benchmark_ij.perf.saved.tioga:40-42"}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--grid", dest="n", type=int, default=256, help="grid points per dimension PER GPU (weak scaling)")
    # the default is BASELINE config C2/C3; the switches below select the other device configs
    ap.add_argument("--problem", choices=["laplacian", "27pt", "difconv"], default="laplacian")
    ap.add_argument("--relax", type=int, default=18, help="smoother (18 l1-Jacobi, 11/12 two-stage GS, 13 l1-GS ...)")
    ap.add_argument("--relax-up", type=int, default=-1,
                    help="smoother of the up leg (e.g. --relax 13 --relax-up 14, the symmetric pair PCG needs)")
    ap.add_argument("--mixed", action="store_true", help="fp32 matrix values inside the cycle (config C5)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-cycles", type=int, default=2)
    ap.add_argument("--cpu-threads", type=int, default=16, help="threads of the CPU baseline (capped by affinity)")
    return ap.parse_args()


def proc_grid(n):
    """P x Q x R box decomposition used by the reference's benchmark jobs (1, 2x1x1, 2x2x1, 2x2x2)."""
    P = Q = R = 1
    k = 0
    while P * Q * R < n:
        if k % 3 == 0:
            P *= 2
        elif k % 3 == 1:
            Q *= 2
        else:
            R *= 2
        k += 1
    if P * Q * R != n:
        raise SystemExit("--gpus must be a power of two")
    return P, Q, R


def main():
    # Host threads are deliberately NOT pinned: the box's share of its host is a CFS quota over all hardware threads (256
    # visible, a quota of 16 cores' worth), not a set of cores, so binding to particular cores would only collide with
    # the other tenants' bindings; and a bound master thread makes every later affinity query (the library's own thread
    # count for the host setup) see one core.  The CPU baseline instead takes the median of five samples.
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N>1 with torch.distributed.run (one process per GPU)")
    dist = None
    if world > 1:
        # a rank that stops making progress (a peer died, a collective never completes) dumps its Python stack and
        # exits instead of hanging the launcher; HYPRE_AMD_BENCH_WATCHDOG seconds, default 20 minutes
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ.get("HYPRE_AMD_BENCH_WATCHDOG", "1200")), exit=True)
    # HYPRE_AMD_BENCH_TRANSPORT=gloo: rehearsal mode (ranks may share one GPU, halo traffic staged
    # over the host through torch.distributed/gloo).  Default: RCCL over xGMI, one GPU per rank.
    transport = os.environ.get("HYPRE_AMD_BENCH_TRANSPORT", "rccl")
    if world > 1:
        import torch
        import torch.distributed as dist
        if transport == "gloo":
            torch.cuda.set_device(local_rank % max(torch.cuda.device_count(), 1))
            dist.init_process_group(backend="gloo")
        else:
            # HYPRE_AMD_BENCH_SHARE_GPU=1 (rehearsal on a one-GPU box): ranks share device 0; RCCL refuses that,
            # which exercises the fallback below
            share = os.environ.get("HYPRE_AMD_BENCH_SHARE_GPU") == "1"
            torch.cuda.set_device(0 if share else local_rank)
            # nccl (= RCCL) for device tensors; gloo beside it only so that the ranks can agree on
            # the fallback below through host tensors should the RCCL communicator be unusable
            dist.init_process_group(backend="cpu:gloo,cuda:nccl",
                                    device_id=None if share else torch.device("cuda", local_rank))

    from hypre_amd import binding as B, ij
    L = B.load_library()          # raises when the HIP library is missing: no fallback exists
    if world > 1 and "HYPRE_AMD_BENCH_KEEP_OMP" not in os.environ:
        # torch.distributed.run pins every rank to one OpenMP thread; the host-side AMG setup of this rank may use its
        # share of the node's cores instead
        L.hypre_amd_SetHostThreads(max(1, L.hypre_amd_HostCpuShare() // world))
    if not L.hypre_amd_DeviceAvailable():
        raise SystemExit("bench.py needs a HIP device")

    comm = 0
    if world > 1:
        from hypre_amd import distributed
        if transport == "gloo":
            # rehearsal: the library's device-buffer halo flow (events, communication stream), bytes over gloo
            comm = distributed.create_stream_staged_comm(dist, rank, world)
        else:
            import torch
            ok = 1
            try:
                comm = distributed.create_rccl_comm(dist, rank, world)
                if L.hypre_amd_CommSelfTest(comm, 1 << 16) != 0:
                    ok = 0
                B.check()
            except Exception as exc:                  # noqa: BLE001 - any failure means "no RCCL transport"
                print("rank %d: RCCL communicator unusable (%s)" % (rank, exc), file=sys.stderr, flush=True)
                L.HYPRE_ClearAllErrors()
                ok = 0
            flag = torch.tensor([ok], dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 0:
                # loud, and recorded in the JSON line: halo traffic staged over the host instead of xGMI
                print("rank %d: falling back to the host-staged gloo transport" % rank, file=sys.stderr, flush=True)
                transport = "gloo-fallback (RCCL self-test failed)"
                comm = distributed.create_stream_staged_comm(dist, rank, world)

    P, Q, R = proc_grid(world)
    n1 = args.n
    opt = ij.IJOptions(n=(n1 * P, n1 * Q, n1 * R), P=(P, Q, R), coarsen_type=8, interp_type=6, P_max_elmts=4,
                       relax_type=args.relax, num_sweeps=1, problem=args.problem)
    if args.relax_up > -1:
        opt.relax_down, opt.relax_up = args.relax, args.relax_up
    if args.problem == "difconv":
        opt.c, opt.a = (1.0, 1.0, 0.001), (0.0, 0.0, 0.0)       # anisotropic diffusion (config C5)
    t0 = time.time()
    A = ij.build_matrix(opt, comm=comm, rank=rank, nprocs=world)
    matrix_s = time.time() - t0
    s = ij.create_amg(opt, memory_location=B.HYPRE_MEMORY_DEVICE)
    if args.mixed:
        L.hypre_amd_BoomerAMGSetMixedPrecision(s, 1)
    # the matrix is handed to the setup in device memory, as the reference's driver does with -exec device (ij.c migrates
    # the IJ matrix first), and the whole setup runs there — on several ranks too (the distributed levels run the
    # single-rank kernels on the extended numbering; HYPRE_AMD_SETUP_DEVICE_DIST=0 keeps them on the host)
    L.hypre_ParCSRMatrixMigrate(A, B.HYPRE_MEMORY_DEVICE)
    L.hypre_SyncComputeStream()
    t1 = time.time()
    L.HYPRE_BoomerAMGSetup(s, A, None, None)
    L.hypre_SyncComputeStream()
    setup_path = "device"
    if dist is not None:
        # several ranks: should the device form of the distributed setup fail on any rank (it has run between ranks that
        # share a card, over the stream-staged transport; RCCL between eight cards is the driver's run), every rank falls
        # back — loudly, and recorded in the JSON line — to the host routines instead of losing the run
        import torch
        if os.environ.get("HYPRE_AMD_BENCH_FAKE_SETUP_ERROR") == str(rank):
            L.HYPRE_BoomerAMGSolve(None, None, None, None)        # rehearsal of this branch: an error on one rank
        bad = torch.tensor([1 if L.HYPRE_GetError() else 0], dtype=torch.int32)
        dist.all_reduce(bad, op=dist.ReduceOp.MAX)
        if int(bad.item()):
            msg = L.hypre_amd_LastErrorMessage().decode() if L.HYPRE_GetError() else "(another rank)"
            print("rank %d: device setup of the distributed hierarchy failed (%s): repeating it with the host routines" % (rank, msg),
                  file=sys.stderr, flush=True)
            L.HYPRE_ClearAllErrors()
            L.HYPRE_BoomerAMGDestroy(s)
            L.hypre_amd_SetSetupDeviceDist(0)
            s = ij.create_amg(opt, memory_location=B.HYPRE_MEMORY_DEVICE)
            if args.mixed:
                L.hypre_amd_BoomerAMGSetMixedPrecision(s, 1)
            t1 = time.time()
            L.HYPRE_BoomerAMGSetup(s, A, None, None)
            L.hypre_SyncComputeStream()
            setup_path = "host (fallback after a failed device setup)"
    B.check()
    setup_s = time.time() - t1            # HYPRE_BoomerAMGSetup alone
    L.hypre_ParCSRMatrixMigrate(A, B.HYPRE_MEMORY_DEVICE)
    Am = A.contents
    nloc = Am.diag.contents.num_rows
    nglob = int(Am.global_num_rows)
    b = B.parvec_from_numpy(np.ones(nloc), comm=comm, global_size=nglob, first=int(Am.row_starts[0]))
    u = B.parvec_from_numpy(np.zeros(nloc), comm=comm, global_size=nglob, first=int(Am.row_starts[0]))
    L.HYPRE_BoomerAMGSetTol(s, 0.0)
    L.HYPRE_BoomerAMGSetMaxIter(s, 1)
    L.hypre_SetSyncCudaCompute(0)

    def step():
        L.hypre_ParVectorSetZeros(u)                 # PCG's ClearVector before each preconditioner call
        L.HYPRE_BoomerAMGSolve(s, A, b, u)

    def fence():
        L.hypre_SyncComputeStream()
        if dist is not None:
            import torch
            torch.cuda.synchronize()
            if transport == "rccl":
                dist.barrier()
            else:
                dist.all_reduce(torch.zeros(1))      # host-side barrier over gloo
            torch.cuda.synchronize()

    warm = max(args.warmup, 2)                           # at least two: the coarse-tail graph is recorded on the second cycle
    for _ in range(warm):
        step()
    fence()
    B.check()
    L.hypre_amd_CommCounters(None, None, 1)
    L.hypre_amd_ByteCounters(None, None, 1)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    b_csr, b_str = C.c_double(), C.c_double()
    L.hypre_amd_ByteCounters(C.byref(b_csr), C.byref(b_str), 0)      # this rank's launches of the timed cycles
    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if transport == "rccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    B.check()
    n_exch, n_allr, x_bytes, r_bytes = C.c_longlong(), C.c_longlong(), C.c_longlong(), C.c_longlong()
    L.hypre_amd_CommCounters(C.byref(n_exch), C.byref(n_allr), 0)
    L.hypre_amd_CommBytes(C.byref(x_bytes), C.byref(r_bytes))
    # the exchanges of one cycle level by level, from the hierarchy's communication packages (this rank's share): a
    # distributed level exchanges u twice through A's package (residual, post-smoothing), the coarse correction once
    # through P's (prolongation) and the restricted residual's off-rank part once through P's in reverse
    exchange_plan = []
    if world > 1:
        nlv = L.hypre_amd_BoomerAMGGetNumLevels(s)
        rep = int(L.hypre_amd_BoomerAMGGetReplicatedLevel(s))
        ns, se, nr, re_ = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        for l in range(nlv - 1 if rep < 0 else min(rep, nlv - 1)):
            Al = C.cast(L.hypre_amd_BoomerAMGGetA(s, l), C.POINTER(B.ParCSRMatrix))
            Pl = C.cast(L.hypre_amd_BoomerAMGGetP(s, l), C.POINTER(B.ParCSRMatrix))
            L.hypre_amd_ParCSRMatrixHaloInfo(Al, C.byref(ns), C.byref(se), C.byref(nr), C.byref(re_))
            row = {"level": l, "local_rows": int(Al.contents.diag.contents.num_rows), "A_neighbours": ns.value,
                   "A_send_bytes": 8 * se.value}
            L.hypre_amd_ParCSRMatrixHaloInfo(Pl, C.byref(ns), C.byref(se), C.byref(nr), C.byref(re_))
            row.update(P_neighbours=ns.value, P_send_bytes=8 * se.value, PT_send_bytes=8 * re_.value,
                       exchanges=4, bytes_sent=2 * row["A_send_bytes"] + 8 * se.value + 8 * re_.value)
            exchange_plan.append(row)
    # ---- how much of every exchange the interior work hid (N > 1): a few more cycles with timing events on every
    # exchange — buffer packed, transfers done, compute stream about to wait for them — read out per AMG level: the EXPOSED
    # time is what the compute stream spent waiting at the halo event.  Rank 0's figures; per cycle.
    overlap = None
    if world > 1:
        MT = 32
        kd = max(1, min(args.steps, 5))
        L.hypre_amd_CommSetTiming(1)
        for _ in range(kd):
            step()
        fence()
        ex, ar = (C.c_int * MT)(), (C.c_int * MT)()
        eu, tu, hu = (C.c_double * MT)(), (C.c_double * MT)(), (C.c_double * MT)()
        L.hypre_amd_CommExposedTimes(MT, ex, ar, eu, tu, hu)
        L.hypre_amd_CommSetTiming(0)
        B.check()
        planned = {row["level"] for row in exchange_plan}
        for row in exchange_plan:
            l = row["level"]
            row.update(timed_exchanges=ex[l] / kd, exposed_us=eu[l] / kd, transfer_us=tu[l] / kd, host_in_transport_us=hu[l] / kd)
        others = [{"level": (l if l < MT - 1 else "outside a cycle"), "exchanges": ex[l] / kd, "allreduces": ar[l] / kd,
                   "exposed_us": eu[l] / kd, "transfer_us": tu[l] / kd, "host_in_transport_us": hu[l] / kd}
                  for l in range(MT) if l not in planned and (ex[l] or ar[l])]
        overlap = {"cycles_timed": kd, "exposed_us_per_cycle": sum(eu) / kd, "transfer_us_per_cycle": sum(tu) / kd,
                   "host_in_transport_us_per_cycle": sum(hu) / kd,
                   "exchanges_per_cycle": sum(ex) / kd, "allreduces_per_cycle": sum(ar) / kd,
                   "levels_without_a_package_row": others,
                   "what": "exposed = compute stream waiting at the halo event (not hidden behind the interior product / sweep); "
                           "transfer = send buffer packed -> transfers done on the communication stream; host_in_transport = "
                           "wall-clock time of the host inside the transport's calls (RCCL: an enqueue; a host-staged transport "
                           "blocks there for the whole transfer and enqueues nothing meanwhile: the compute stream then runs dry "
                           "without waiting at an event, so ms_per_step - single-rank ms ~ exposed + host_in_transport there); "
                           "the first sweep of a cycle starts from zero and exchanges nothing (3 of the 4 planned on level 0); rank 0"}
    g_level, g_nodes = C.c_int(), C.c_int()
    L.hypre_amd_BoomerAMGGetGraphInfo(s, C.byref(g_level), C.byref(g_nodes))
    ms_per_step = 1e3 * elapsed / args.steps
    dof_per_s = nglob / (elapsed / args.steps)

    # ---- kernel figures: y = A x of the fine-level operator (as launched) and of level 1 ------------------
    # Every figure carries two byte counts, both taken from the library's own launch accounting (hypre_amd_ByteCounters
    # around one launch): `csr` = SURVEY.md 8(d)'s count (12 bytes per entry, row pointers, vectors once) and `streamed` =
    # what the FORMAT the launched kernel reads requires (16-bit local indices instead of columns; one-byte value codes;
    # padded slices).  The roofline fraction is streamed bytes / time / peak — at most 1 by construction, cross-checked
    # against PMC traffic (profiles/) —; the CSR count over the same time is reported as `effective_csr_GBps`, a rate, not
    # a fraction of anything.
    diag = Am.diag
    nnz = diag.contents.num_nonzeros
    reps = 50
    FORMS = {0: "spmv_wave_kernel (a wave per row)", 1: "spmv_tiled_kernel (x gathered through the cache)",
             2: "spmv_xs_kernel (2048-entry tiles, x staged through LDS, fp64 values + 16-bit local indices)",
             3: "spmv_xs_kernel, coded (2048-entry tiles, x staged through LDS, one-byte value codes + 16-bit local indices)",
             4: "spmv_sl_kernel (coded stencil in slice form: a lane per row or half row, codes and local indices in registers)",
             5: "spmv_rs_kernel (row-slice form: jagged slices of fp64 values + 16-bit local indices, sums in registers)"}

    def spmv_figure(M, seed):
        """y = M x through the public entry: HIP-event time per launch, the library's two byte counts of one launch"""
        mc = M.contents
        xv = B.vec_from_numpy(np.random.default_rng(seed).uniform(-1, 1, mc.num_cols))
        yv = B.vec_from_numpy(np.zeros(mc.num_rows))
        for _ in range(5):
            L.hypre_CSRMatrixMatvec(1.0, M, xv, 0.0, yv)
        L.hypre_SyncComputeStream()
        L.hypre_amd_ByteCounters(None, None, 1)
        L.hypre_CSRMatrixMatvec(1.0, M, xv, 0.0, yv)
        c1, s1 = C.c_double(), C.c_double()
        L.hypre_amd_ByteCounters(C.byref(c1), C.byref(s1), 1)
        L.hypre_SyncComputeStream()
        L.hypre_amd_EventTimerStart()
        for _ in range(reps):
            L.hypre_CSRMatrixMatvec(1.0, M, xv, 0.0, yv)
        ms = L.hypre_amd_EventTimerStopMs() / reps
        L.hypre_SeqVectorDestroy(xv)
        L.hypre_SeqVectorDestroy(yv)
        B.check()
        form = int(L.hypre_amd_CSRMatrixPlanForm(M))
        csr_b = mc.num_nonzeros * (4 + value_width) + (mc.num_rows + 1) * 4 + mc.num_cols * 8 + mc.num_rows * 8
        streamed = s1.value
        frac = streamed / ms / 1e6 / HBM_PEAK_GBS
        # (the bytes are what the launched kernel's format makes it read, so a fraction above 1 can only mean that they did not
        # come from HBM: a working set that fits the 256 MB Infinity Cache — small --grid runs; never the benchmark sizes)
        note = None if frac <= 1.0 else "above 1: working set of %.0f MB served from cache, not an HBM figure" % (streamed / 1e6)
        return {"note": note,"kernel": FORMS.get(form, "?"), "form": form, "rows": mc.num_rows, "nnz": mc.num_nonzeros, "ms_per_launch": ms,
                "streamed_bytes_per_launch": streamed, "achieved": streamed / ms / 1e6, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": frac, "csr_bytes_per_launch": csr_b, "library_csr_count": c1.value,
                "effective_csr_GBps": csr_b / ms / 1e6,
                "value_codes": int(L.hypre_amd_CSRMatrixPlanValueCodes(M))}

    value_width = 4 if args.mixed else 8
    fine = spmv_figure(diag, rank)
    ndict = fine["value_codes"]
    spmv_bytes = fine["csr_bytes_per_launch"]
    # Bytes of one cycle on this rank (every rank carries the same share: weak scaling): counted by the launch wrappers
    # over the timed cycles, so the figure follows the smoother, the value width and the levels actually run (two-stage
    # GS: residual pass + inner passes over the strict lower triangle; fp32 values: 4 instead of 8 bytes per entry).
    cycle_bytes = b_csr.value / args.steps
    cycle_streamed = b_str.value / args.steps
    cycle_formula = L.hypre_amd_BoomerAMGCycleBytes(s)
    B.check()

    # ---- what this box's memory sustains: a plain device copy (read + write), next to which box-to-box spread of the
    # HBM-bound numbers above can be read (compute-bound kernels repeat to 0.5 % between boxes, the big SpMV launches to 4-7 %)
    nb = 1 << 27                                                       # 1 GiB per vector
    cx, cy = B.vec_from_numpy(np.zeros(nb)), B.vec_from_numpy(np.zeros(nb))
    for _ in range(3):
        L.hypre_SeqVectorCopy(cx, cy)
    L.hypre_SyncComputeStream()
    L.hypre_amd_EventTimerStart()
    for _ in range(10):
        L.hypre_SeqVectorCopy(cx, cy)
    copy_gbs = 2.0 * 8.0 * nb / (L.hypre_amd_EventTimerStopMs() / 10) / 1e6
    L.hypre_SeqVectorDestroy(cx)
    L.hypre_SeqVectorDestroy(cy)
    B.check()

    # ---- the caller of the path: AMG-preconditioned CG to 1e-8 (BASELINE configs solve with it) ----
    pcg_info = None
    try:
        pcg = C.c_void_p()
        L.HYPRE_ParCSRPCGCreate(comm, C.byref(pcg))
        L.HYPRE_PCGSetTol(pcg, 1.0e-8)
        L.HYPRE_PCGSetMaxIter(pcg, 100)
        L.HYPRE_PCGSetTwoNorm(pcg, 1)
        L.HYPRE_PCGSetPrecond(pcg, C.cast(L.HYPRE_BoomerAMGSolve, C.c_void_p), None, s)
        L.hypre_ParVectorSetZeros(u)
        L.HYPRE_ParCSRPCGSetup(pcg, A, b, u)
        fence()
        t0 = time.perf_counter()
        L.HYPRE_ParCSRPCGSolve(pcg, A, b, u)
        fence()
        pcg_s = time.perf_counter() - t0
        its, rel = C.c_int(), C.c_double()
        L.HYPRE_PCGGetNumIterations(pcg, C.byref(its))
        L.HYPRE_PCGGetFinalRelativeResidualNorm(pcg, C.byref(rel))
        L.HYPRE_ParCSRPCGDestroy(pcg)
        B.check()
        pcg_info = {"iterations": its.value, "final_rel_resid": rel.value, "solve_ms": 1e3 * pcg_s,
                    "ms_per_iteration": 1e3 * pcg_s / max(its.value, 1),
                    "dof_per_s_per_iteration": nglob * max(its.value, 1) / pcg_s}
        if world == 1 and args.problem == "laplacian" and args.n == 256 and args.relax == 18 and not args.mixed:
            pcg_info["reference_published_other_hardware"] = REFERENCE_GPU_PUBLISHED
    except Exception as exc:      # noqa: BLE001 - the headline metric above stands on its own
        pcg_info = {"error": str(exc)}
        L.HYPRE_ClearAllErrors()
    # ---- after the solves (dropping the plan of the fine-level matrix also drops what else the library cached for it — the
    # strictly lower copy of the two-stage sweeps, colour classes, level schedules —, which the next solve would rebuild)
    # the same cycle and the same fine-level launch with the value codes OFF: what a variable-coefficient operator of this
    # size gets (fp64 values streamed by the tiled kernel) — the general-CSR figures of this run
    fine_general, ms_per_step_codes_off = fine, ms_per_step
    if ndict:
        L.hypre_amd_SpmvSetValueCodes(0)
        L.hypre_amd_CSRMatrixInvalidatePlan(diag)
        fine_general = spmv_figure(diag, rank)
        for _ in range(warm):
            step()
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        fence()
        el = time.perf_counter() - t0
        if dist is not None:
            import torch
            t = torch.tensor([el], dtype=torch.float64, device="cuda" if transport == "rccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        ms_per_step_codes_off = 1e3 * el / args.steps
        L.hypre_amd_SpmvSetValueCodes(1)
        L.hypre_amd_CSRMatrixInvalidatePlan(diag)
        step()                                                         # the coded plans again, for what follows
        fence()
        B.check()
    # the largest launch of the cycle below the fine level: y = A_1 x on this rank's block of level 1 (values all distinct)
    level1 = None
    if int(L.hypre_amd_BoomerAMGGetNumLevels(s)) > 2:
        lv1_A = C.cast(L.hypre_amd_BoomerAMGGetA(s, 1), C.POINTER(B.ParCSRMatrix)).contents.diag
        if lv1_A.contents.num_rows > 0 and lv1_A.contents.num_nonzeros > 0:
            level1 = spmv_figure(lv1_A, rank + 7)
    # the kernel with the largest share of the timed cycle: each level's operator is passed over twice per V(1,1) cycle
    # (residual, post-smoothing sweep; the pre-smoothing starts from zero), so the larger of the two launches decides
    dominant, dominant_level = fine, 0
    if level1 is not None and level1["ms_per_launch"] > fine["ms_per_launch"]:
        dominant, dominant_level = level1, 1
    # HBM traffic per launch cannot be read from inside the process (PMC counters need rocprofv3 around it): the figure is
    # taken from the newest committed --pmc pass over this same kernel and matrix (profiles/, tools/pmc_levels.sh) when
    # the workload matches, and labelled as what it is: a profile-derived number of an earlier run of this kernel — with
    # the hash of the kernel source it was taken from, so that a changed kernel shows
    traffic = None
    traffic_src = None
    try:
        import hashlib
        with open(os.path.join(ROOT, "hypre_amd", "csrc", "spmv_kernels.hip"), "rb") as fh:
            kernel_sha = hashlib.sha256(fh.read()).hexdigest()[:16]
        pmc_files = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if "levels_" in f and f.endswith("_summary.json")
                           and "gather" not in f and (("codes" in f) == bool(dominant["value_codes"])))
        if pmc_files:
            with open(os.path.join(ROOT, "profiles", pmc_files[-1])) as fh:
                summary = json.load(fh)
            pmc = summary["levels"][str(dominant_level)]
            if int(pmc.get("algorithmic_bytes_per_launch", -1)) == int(dominant["csr_bytes_per_launch"]) and "hbm_traffic_bytes_per_launch" in pmc:
                traffic = pmc["hbm_traffic_bytes_per_launch"]
                same = summary.get("kernel_source_sha16") == kernel_sha
                traffic_src = "profile-derived, not of this run: profiles/%s (%s)" % (
                    pmc_files[-1], "same kernel source" if same else "taken from an EARLIER version of the kernel source")
    except (OSError, KeyError, ValueError):
        pass

    # ---- CPU baseline: the oracle's V-cycle on the same hierarchy, host cores ------------
    # N = 1: the hierarchy just timed.  N > 1 (rank 0 only, the others wait at the fence below): the per-GPU block of the
    # job — the n^3 problem of one rank — set up as a single-rank hierarchy on rank 0's GPU, its cycle checked against
    # and timed with the oracle; the job is N such blocks (weak scaling), so DOF/s is the comparable figure.
    cpu = None
    if rank == 0 and not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import pyoracle as O
        if world == 1:
            s1, A1, n_cpu, b1, u1 = s, A, nloc, b, u
        else:
            opt1 = ij.IJOptions(n=(n1, n1, n1), P=(1, 1, 1), coarsen_type=8, interp_type=6, P_max_elmts=4,
                                relax_type=args.relax, num_sweeps=1, problem=args.problem)
            if args.relax_up > -1:
                opt1.relax_down, opt1.relax_up = args.relax, args.relax_up
            if args.problem == "difconv":
                opt1.c, opt1.a = (1.0, 1.0, 0.001), (0.0, 0.0, 0.0)
            A1 = ij.build_matrix(opt1)
            s1 = ij.create_amg(opt1, memory_location=B.HYPRE_MEMORY_DEVICE)
            if args.mixed:
                L.hypre_amd_BoomerAMGSetMixedPrecision(s1, 1)
            L.hypre_ParCSRMatrixMigrate(A1, B.HYPRE_MEMORY_DEVICE)
            L.HYPRE_BoomerAMGSetup(s1, A1, None, None)
            B.check()
            n_cpu = n1 ** 3
            b1, u1 = B.parvec_from_numpy(np.ones(n_cpu)), B.parvec_from_numpy(np.zeros(n_cpu))
            L.HYPRE_BoomerAMGSetTol(s1, 0.0)
            L.HYPRE_BoomerAMGSetMaxIter(s1, 1)
        if world > 1:
            # the same per-GPU block as ONE rank's job: what a cycle costs without any exchange (the other ranks wait)
            def step1():
                L.hypre_ParVectorSetZeros(u1)
                L.HYPRE_BoomerAMGSolve(s1, A1, b1, u1)
            for _ in range(warm):
                step1()
            L.hypre_SyncComputeStream()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                step1()
            L.hypre_SyncComputeStream()
            single_rank_ms = 1e3 * (time.perf_counter() - t0) / args.steps
            if overlap is not None:
                overlap["single_rank_ms_per_step_same_block"] = single_rank_ms
                overlap["ms_per_step_minus_single_rank_ms"] = ms_per_step - single_rank_ms
        amg = O.amg_from_solvers([s1], mixed_precision=args.mixed)
        f = np.ones(n_cpu)
        ur = np.zeros(n_cpu)

        def cpu_cycles(threads, cycles):
            O.set_num_threads(threads)
            ur[:] = 0.0
            amg.cycle(f, ur, u_all_zeros=True)          # warm the caches / page in / build transposes
            t0 = time.perf_counter()
            for _ in range(cycles):
                ur[:] = 0.0
                amg.cycle(f, ur, u_all_zeros=True)
            return (time.perf_counter() - t0) / cycles

        cores = max(1, min(len(os.sched_getaffinity(0)), args.cpu_threads))
        cpu_1 = cpu_cycles(1, args.cpu_cycles)
        # threaded: median of 5 samples of 4 cycles each (threads float, see main()); run to run the mean of one long
        # sample moved by +-25 % on the GPU box's shared host
        samples = sorted(cpu_cycles(cores, 2 * args.cpu_cycles) for _ in range(5)) if cores > 1 else [cpu_1]
        cpu_s = samples[len(samples) // 2]
        O.set_num_threads(1)
        O.drop_transposes()
        L.hypre_ParVectorSetZeros(u1)                # one cycle from zero again (the PCG run above reused u)
        L.HYPRE_BoomerAMGSolve(s1, A1, b1, u1)
        L.hypre_SyncComputeStream()
        B.check()
        ug = B.parvec_to_numpy(u1)
        parity = float(np.max(np.abs(ug - ur)) / np.max(np.abs(ur)))
        which = "the same %d^3 hierarchy" % n1 if world == 1 else \
            "the per-GPU block of the job (%d^3, set up as a single-rank hierarchy on rank 0; the job is %d such blocks)" % (n1, world)
        cpu = {"value": n_cpu / cpu_s, "unit": "DOF/s", "cores": cores, "kind": "port",
               "sample": "median of %d samples of %d full V(1,1) cycles of %s (oracle/oracle.c, "
                         "OpenMP row loops, %d threads, not pinned: OMP_PROC_BIND=%s, the host share is a CFS quota)"
                         % (len(samples), 2 * args.cpu_cycles if cores > 1 else args.cpu_cycles, which, cores,
                            os.environ.get("OMP_PROC_BIND", "unset")),
               "samples_DOF_per_s": [n_cpu / t for t in samples],
               "single_thread_value": n_cpu / cpu_1,
               # the reference itself cannot run on the GPU box; what it did in the survey's container (8 cores), and the
               # port timed beside it there (tools/cpu_baseline_check.py -> profiles/r02_cpu_port_vs_reference_*.json)
               "reference_cpu_survey_container": REFERENCE_CPU_SURVEY,
               "gpu_vs_cpu_cycle_rel_max_diff": parity}
        if world > 1:
            L.HYPRE_BoomerAMGDestroy(s1)
            L.hypre_ParCSRMatrixDestroy(A1)
            B.check()

    # parity gate: a fast cycle whose result differs from the oracle's is not a result
    parity_failed = bool(cpu is not None and not (cpu["gpu_vs_cpu_cycle_rel_max_diff"] <= PARITY_TOL))
    if rank == 0:
        g, o = C.c_double(), C.c_double()
        L.hypre_amd_BoomerAMGGetComplexities(s, C.byref(g), C.byref(o))
        stencil = {"laplacian": "7-pt Laplacian", "27pt": "27-pt Laplacian",
                   "difconv": "7-pt anisotropic diffusion (1, 1, 0.001)"}[args.problem]
        smoother = {18: "l1-Jacobi", 7: "Jacobi", 0: "weighted Jacobi", 11: "two-stage GS (1 inner)",
                    12: "two-stage GS (2 inner)"}.get(args.relax, "relax %d" % args.relax)
        if args.relax_up > -1:
            smoother += " down / relax %d up" % args.relax_up
        arith = "fp32 matrix values / fp64 vectors" if args.mixed else "fp64"
        out = {
            "metric": "BoomerAMG V-cycle DOF/s (%d^3 %s per GPU, %s V(1,1), %s)" % (n1, stencil, smoother, arith),
            "value": dof_per_s, "unit": "DOF/s", "n_gpus": world, "steps": args.steps, "warmup": warm,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%dx%dx%d %s, %d rank(s) %dx%dx%d, PMIS + ext+i(4) + %s V(1,1)%s"
                                   % (n1 * P, n1 * Q, n1 * R, stencil, world, P, Q, R, smoother,
                                      ", fp32 matrix values in the cycle" if args.mixed else ""),
                       "transport": transport if world > 1 else "none",
                       "replicated_from_level": int(L.hypre_amd_BoomerAMGGetReplicatedLevel(s)),
                       "halo_exchanges_per_cycle": n_exch.value / args.steps, "allreduces_per_cycle": n_allr.value / args.steps,
                       "halo_bytes_sent_per_cycle": x_bytes.value / args.steps, "allreduce_bytes_per_cycle": r_bytes.value / args.steps,
                       "exchanges_by_level": exchange_plan, "overlap": overlap,
                       "rccl_self_test": (None if world == 1 else ("passed" if transport == "rccl" else "not run (transport %s)" % transport)),
                       "coarse_tail_graph_from_level": g_level.value, "coarse_tail_graph_nodes": g_nodes.value,
                       "one_workgroup_tail_from_level": int(L.hypre_amd_BoomerAMGGetSmallTailLevel(s)),
                       "levels": int(L.hypre_amd_BoomerAMGGetNumLevels(s)), "grid_complexity": g.value,
                       "operator_complexity": o.value, "setup_seconds": setup_s, "setup_path": setup_path,
                       "matrix_generation_seconds": matrix_s},
            # the kernel with the largest share of the timed cycle; frac = streamed bytes / time / peak (<= 1 by construction)
            "roofline": {"bound": "hbm", "kernel": "y = A_%d x: %s" % (dominant_level, dominant["kernel"]), "level": dominant_level,
                         "achieved": dominant["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": dominant["frac"],
                         "traffic": traffic, "traffic_source": traffic_src,
                         "bytes_per_launch": dominant["streamed_bytes_per_launch"], "ms_per_launch": dominant["ms_per_launch"],
                         "bytes_are": "what the launched kernel's format requires (library accounting of one launch), not the CSR count",
                         "effective_csr_GBps": dominant["effective_csr_GBps"], "csr_bytes_per_launch": dominant["csr_bytes_per_launch"],
                         "share_of_cycle": 2.0 * dominant["ms_per_launch"] / ms_per_step,
                         "device_copy_GBps_this_box": copy_gbs},
            # top-level so that a parser that keeps only first-level keys keeps them: the fine-level operator as launched
            # (coded stencil), the same operator as general CSR (codes off: fp64 values), level 1, the cycle with codes off
            "spmv_fine_as_launched": fine,
            "spmv_fine_general_csr": fine_general,
            "spmv_level1": level1,
            "ms_per_step_codes_off": ms_per_step_codes_off,
            "value_codes_off_DOF_per_s": nglob / (ms_per_step_codes_off * 1e-3),
            "vcycle": {"streamed_bytes": cycle_streamed, "achieved_GBps": cycle_streamed / ms_per_step / 1e6,
                       "frac_of_hbm_peak": cycle_streamed / ms_per_step / 1e6 / HBM_PEAK_GBS,
                       "bytes_are": "what the launched kernels' formats require; the CSR count is effective_csr_*",
                       "effective_csr_bytes": cycle_bytes, "effective_csr_GBps": cycle_bytes / ms_per_step / 1e6,
                       "per": "GPU (rank 0's launches; every rank carries the same share)",
                       "counted_by": "the library's launch wrappers over the timed cycles (hypre_amd_ByteCounters)",
                       "jacobi_csr_formula_bytes": cycle_formula},
            "pcg": pcg_info,
            "cpu_baseline": cpu,
        }
        if parity_failed:
            out["parity_failed"] = True
        print(json.dumps(out), flush=True)
    if dist is not None:
        if transport == "rccl":
            dist.barrier()
        dist.destroy_process_group()
    if parity_failed:
        print("bench.py: the device V-cycle differs from the oracle's by %.3e (> %.1e, relative max-norm): the numbers "
              "above are INVALID" % (cpu["gpu_vs_cpu_cycle_rel_max_diff"], PARITY_TOL), file=sys.stderr, flush=True)
        sys.exit(3)


if __name__ == "__main__":
    main()
