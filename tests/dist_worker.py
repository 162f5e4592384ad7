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

    case = json.loads(sys.argv[1])
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    L = B.load_library()
    comm = distributed.create_callback_comm(dist, rank, world)
    opt = ij.IJOptions(**{k: (tuple(v) if isinstance(v, list) else v) for k, v in case["options"].items()})
    A = ij.build_matrix(opt, comm=comm, rank=rank, nprocs=world)
    s = ij.create_amg(opt, memory_location=B.HYPRE_MEMORY_HOST)
    L.HYPRE_BoomerAMGSetup(s, A, None, None)
    B.check()
    g, o = C.c_double(), C.c_double()
    L.hypre_amd_BoomerAMGGetComplexities(s, C.byref(g), C.byref(o))

    def allsum(v):
        import torch
        t = torch.tensor([v], dtype=torch.float64)
        dist.all_reduce(t)
        return float(t.item())

    b, x = ij.build_rhs_host(opt, A, rank=rank, allreduce=allsum)
    mine = dict(h=O.export_solver(s), b=b, x=x)
    parts = [None] * world if rank == 0 else None
    dist.gather_object(mine, parts, dst=0)
    if rank == 0:
        amg = O.amg_from_exports([p["h"] for p in parts], num_threads=opt.num_threads)
        n = amg.A_levels[0].nrows
        xg = np.concatenate([p["x"] for p in parts])
        if parts[0]["b"] is None:
            bg = np.zeros(n)
            O.par_matvec(1.0, amg.A_levels[0], np.ones(n), 0.0, bg, bg)
        else:
            bg = np.concatenate([p["b"] for p in parts])
        out = {"grid": g.value, "operator": o.value, "levels": amg.c.num_levels,
               "sizes": [a.nrows for a in amg.A_levels]}
        if opt.solver == 0:
            its, rel, conv, hist = amg.solve(bg, xg, tol=opt.tol, max_iter=opt.mg_max_iter)
            out.update(iterations=its, rel_resid=rel, conv_factor=(hist[-1] / hist[0]) ** (1.0 / max(its, 1)))
        else:
            its, rel, conv = amg.pcg(bg, xg, tol=opt.tol, max_iter=opt.max_iter, two_norm=opt.two_norm,
                                     precond_cycles=opt.precon_cycles)
            out.update(iterations=its, rel_resid=rel)
        print("RESULT " + json.dumps(out), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
