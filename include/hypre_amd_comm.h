/*
 * hypre_amd — communicator shim.
 *
 * The reference reaches MPI through hypre_MPI_* wrappers (utilities/mpistubs.[ch];
 * sequential builds replace them with stubs and MPI_Comm becomes an integer,
 * mpistubs.h:142).  This library keeps that integer-handle convention: an
 * MPI_Comm is an index into a table of communicator objects, each of which is
 * a small table of function pointers.  Three providers exist:
 *
 *   - handle 0 (hypre_MPI_COMM_WORLD): one rank, no communication;
 *   - hypre_amd_CommCreateRCCL: one process per GPU, neighbour exchange as
 *     grouped ncclSend/ncclRecv over xGMI, reductions as ncclAllReduce;
 *   - hypre_amd_CommCreate: caller-supplied callbacks (an MPI application
 *     would wrap MPI_Isend/Irecv/Waitall here; the CPU test-suite wraps
 *     torch.distributed/gloo).
 *
 * Only the collective shapes that occur on the BoomerAMG solve path are
 * modelled: the neighbour halo exchange of
 * parcsr_mv/par_csr_communication.c:483-526, the scalar all-reduce of
 * parcsr_mv/par_vector.c:526 and the small all-gather(v) of
 * parcsr_ls/par_gauss_elim.c:575 / par_csr_communication.c:805-840.
 */
#ifndef HYPRE_AMD_COMM_H
#define HYPRE_AMD_COMM_H

#include "HYPRE_amd_utilities.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct
{
   void *ctx;
   int   rank;
   int   size;
   /* Post every receive and every send of one neighbour exchange and complete
    * them.  on_device != 0: buffers are device pointers and the call only has
    * to ENQUEUE the transfers on `stream` (a hipStream_t); on_device == 0:
    * buffers are host pointers and the call returns when they are complete. */
   int (*exchange)(void *ctx,
                   int num_sends, const int *dest, void *const *send_buf, const size_t *send_bytes,
                   int num_recvs, const int *src,  void *const *recv_buf, const size_t *recv_bytes,
                   int on_device, void *stream);
   /* in-place sum over all ranks of `count` doubles (same on_device rule) */
   int (*allreduce_sum)(void *ctx, double *buf, int count, int on_device, void *stream);
   /* host all-gather of a fixed-size record from every rank, rank order */
   int (*allgather)(void *ctx, const void *send_buf, void *recv_buf, size_t bytes_per_rank);
   /* optional; may be NULL */
   int (*barrier)(void *ctx);
   void (*destroy)(void *ctx);
   /* 1 if exchange/allreduce_sum accept device pointers (RCCL does); 0 makes
    * the library stage halo buffers through host memory around the call */
   int   device_buffers;
} hypre_amd_CommOps;

MPI_Comm  hypre_amd_CommCreate(const hypre_amd_CommOps *ops);
HYPRE_Int hypre_amd_CommDestroy(MPI_Comm comm);

/* RCCL provider.  Rank 0 obtains an id (128 bytes) with ..GetUniqueId, ships it
 * to the other ranks by any out-of-band means (the launcher's store), and
 * every rank calls ..CreateRCCL.  The calling process must already have
 * selected its GPU (hipSetDevice / torch.cuda.set_device). */
#define HYPRE_AMD_RCCL_ID_BYTES 128
HYPRE_Int hypre_amd_RCCLGetUniqueId(void *id_out);
MPI_Comm  hypre_amd_CommCreateRCCL(const void *id, int rank, int size);

/* Device-buffer semantics on top of a host-buffer communicator `inner` (one made by
 * hypre_amd_CommCreate from MPI or torch.distributed callbacks; it must outlive the new one):
 * halo buffers are copied device -> pinned host on the stream the exchange is given, the host
 * transport runs once that stream has drained, and the received data is copied back
 * asynchronously on the same stream.  The library then drives it exactly like the RCCL provider
 * (pack kernel -> event -> exchange on the communication stream -> event -> ghost product), which is
 * what an application without a device-aware MPI wants, and what lets several ranks share one GPU. */
MPI_Comm  hypre_amd_CommCreateStreamStaged(MPI_Comm inner);

/* Collective health check of a communicator (no reference counterpart; the
 * reference trusts MPI): ring shift of nbytes through exchange() with host and,
 * if the provider takes them, device buffers, a sum all-reduce and an
 * all-gather, each verified.  Returns the number of failed checks. */
HYPRE_Int hypre_amd_CommSelfTest(MPI_Comm comm, HYPRE_Int nbytes);

/* neighbour exchanges and all-reduces this process has started since the last reset (benchmark reporting) */
HYPRE_Int hypre_amd_CommCounters(HYPRE_BigInt *exchanges, HYPRE_BigInt *allreduces, HYPRE_Int reset);
/* Diagnosis of the overlap of halo exchanges with interior work (the reference overlaps them by hand per product,
 * parcsr_mv/par_csr_matvec_device.c:218-256).  With timing on, every device-buffer exchange and device all-reduce started
 * from now on carries timing events of its own — send buffer packed (compute stream), transfers done (communication
 * stream), compute stream about to wait for the transfers — tagged with hypre_amd_CommSetTag's value (the AMG cycle sets
 * the level it is on; -1 outside a cycle).  hypre_amd_CommExposedTimes synchronises both streams, reads the exchanges since
 * the last call out and forgets them: per tag t < max_tags (others in the last slot) the number of exchanges and of
 * all-reduces, the EXPOSED time (compute stream waiting at the halo event: what the interior product or sweep did not
 * hide), the whole time from "buffer packed" to "transfers done", and (host_us, may be NULL) the wall-clock time the host
 * spent inside the transport's calls — an enqueue for RCCL, the whole transfer for a host-staged transport, during which it
 * enqueues nothing else, so that the compute stream runs dry without ever waiting at an event —, in microseconds. */
HYPRE_Int hypre_amd_CommSetTiming(HYPRE_Int on);
HYPRE_Int hypre_amd_CommSetTag(HYPRE_Int tag);
HYPRE_Int hypre_amd_CommExposedTimes(HYPRE_Int max_tags, HYPRE_Int *exchanges, HYPRE_Int *allreduces, HYPRE_Real *exposed_us,
                                     HYPRE_Real *transfer_us, HYPRE_Real *host_us);
/* bytes this process sent in those exchanges / contributed to device all-reduces (cleared with the counters above) */
HYPRE_Int hypre_amd_CommBytes(HYPRE_BigInt *exchange_bytes, HYPRE_BigInt *allreduce_bytes);

HYPRE_Int hypre_MPI_Comm_rank(MPI_Comm comm, HYPRE_Int *rank);
HYPRE_Int hypre_MPI_Comm_size(MPI_Comm comm, HYPRE_Int *size);
HYPRE_Int hypre_MPI_Barrier(MPI_Comm comm);

#ifdef __cplusplus
}
#endif
#endif
