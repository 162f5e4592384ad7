"""Communicator providers for one-process-per-GPU runs.

* create_rccl_comm  — the production path: RCCL send/recv + all-reduce over xGMI
  inside libhypre_amd.so (hypre_amd_CommCreateRCCL).  torch.distributed is only
  the out-of-band channel that ships the 128-byte RCCL id to the other ranks.
* create_callback_comm — a hypre_amd_CommOps table whose entries call back into
  torch.distributed (any backend).  Used by the CPU test-suite with gloo to
  exercise the distributed host logic (comm packages, setup) without a GPU; an
  MPI application would fill the same table with MPI_Isend/Irecv/Waitall.
"""
import ctypes as C

import numpy as np

from . import binding as B

_keepalive = []


def create_rccl_comm(dist, rank, world):
    import torch
    L = B.load_library()
    ident = (C.c_ubyte * 128)()
    if rank == 0:
        L.hypre_amd_RCCLGetUniqueId(C.cast(ident, C.c_void_p))
    t = torch.tensor(list(bytes(ident)), dtype=torch.uint8, device="cuda")
    dist.broadcast(t, src=0)
    raw = bytes(t.cpu().tolist())
    buf = (C.c_ubyte * 128).from_buffer_copy(raw)
    comm = L.hypre_amd_CommCreateRCCL(C.cast(buf, C.c_void_p), rank, world)
    B.check()
    if comm < 0:
        raise B.HypreAmdError("hypre_amd_CommCreateRCCL failed")
    return comm


def create_stream_staged_comm(dist, rank, world):
    """Device-buffer communicator over torch.distributed host traffic (hypre_amd_CommCreateStreamStaged on top of
    create_callback_comm): the library runs its production halo flow — pack kernel, event, exchange enqueued on the
    communication stream, event, ghost product; all-reduces on device buffers — while the bytes travel through host
    memory.  Ranks may share one GPU, which RCCL does not allow: this is how the multi-rank device path is tested on
    a one-GPU box, and the rehearsal transport of bench.py."""
    L = B.load_library()
    inner = create_callback_comm(dist, rank, world)
    comm = L.hypre_amd_CommCreateStreamStaged(inner)
    B.check()
    if comm < 0:
        raise B.HypreAmdError("hypre_amd_CommCreateStreamStaged failed")
    return comm


def _view(ptr, nbytes):
    return np.frombuffer((C.c_ubyte * nbytes).from_address(ptr), dtype=np.uint8)


def create_callback_comm(dist, rank, world):
    """Host-buffer communicator backed by torch.distributed (gloo on CPU)."""
    import torch
    L = B.load_library()

    def exchange(ctx, ns, dest, sbuf, sbytes, nr, src, rbuf, rbytes, on_device, stream):
        ops, keep = [], []
        for i in range(nr):
            if rbytes[i]:
                t = torch.from_numpy(_view(rbuf[i], rbytes[i]))
                keep.append(t)
                ops.append(dist.P2POp(dist.irecv, t, src[i]))
        for i in range(ns):
            if sbytes[i]:
                t = torch.from_numpy(_view(sbuf[i], sbytes[i]).copy())
                keep.append(t)
                ops.append(dist.P2POp(dist.isend, t, dest[i]))
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        return 0

    def allreduce(ctx, buf, count, on_device, stream):
        a = np.frombuffer((C.c_double * count).from_address(C.addressof(buf.contents)), dtype=np.float64)
        t = torch.from_numpy(a.copy())
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        a[:] = t.numpy()
        return 0

    def allgather(ctx, sbuf, rbuf, nbytes):
        mine = torch.from_numpy(_view(sbuf, nbytes).copy())
        out = [torch.empty(nbytes, dtype=torch.uint8) for _ in range(world)]
        dist.all_gather(out, mine)
        dst = _view(rbuf, nbytes * world)
        for r in range(world):
            dst[r * nbytes:(r + 1) * nbytes] = out[r].numpy()
        return 0

    def barrier(ctx):
        dist.barrier()
        return 0

    ops = B.CommOps()
    ops.ctx = None
    ops.rank, ops.size = rank, world
    cbs = (B.EXCHANGE_FN(exchange), B.ALLREDUCE_FN(allreduce), B.ALLGATHER_FN(allgather), B.BARRIER_FN(barrier))
    ops.exchange, ops.allreduce_sum, ops.allgather, ops.barrier = cbs
    ops.destroy = C.cast(None, B.DESTROY_FN)
    ops.device_buffers = 0
    _keepalive.append((ops, cbs))
    comm = L.hypre_amd_CommCreate(C.byref(ops))
    B.check()
    return comm
