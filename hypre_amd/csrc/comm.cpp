// hypre_amd — communicator table and the RCCL provider.
//
// Reference counterpart: utilities/mpistubs.[ch] (the hypre_MPI_* shim) and the
// MPI calls of parcsr_mv/par_csr_communication.c:483-526.  Here the neighbour
// exchange of a halo update is one ncclGroupStart/ncclSend.../ncclRecv.../
// ncclGroupEnd batch on the communication stream: every neighbour pair is a
// direct xGMI link on an MI355X node, so the batch is link-parallel.
#include "internal.hpp"
#include "hypre_amd_comm.h"
#include <rccl/rccl.h>
#include <vector>
#include <mutex>

namespace {

struct CommObj
{
   bool              live = false;
   hypre_amd_CommOps ops{};
};

std::vector<CommObj> &table()
{
   static std::vector<CommObj> t(1);   // slot 0 = single-rank world
   if (!t[0].live) { t[0].live = true; t[0].ops.rank = 0; t[0].ops.size = 1; }
   return t;
}

#define NCCL_CHECK(call)                                                            \
   do {                                                                             \
      ncclResult_t r_ = (call);                                                     \
      if (r_ != ncclSuccess) {                                                      \
         char msg_[512];                                                            \
         snprintf(msg_, sizeof(msg_), "RCCL error %d (%s) in %s", (int) r_,         \
                  ncclGetErrorString(r_), #call);                                   \
         hypre_error_handler(__FILE__, __LINE__, HYPRE_ERROR_GENERIC, msg_);        \
         return 1;                                                                  \
      }                                                                             \
   } while (0)

struct RcclCtx
{
   ncclComm_t comm = nullptr;
   int rank = 0, size = 1;
   // device staging for host-buffer traffic (setup-time metadata, tiny vectors)
   char  *d_stage = nullptr;
   size_t d_stage_len = 0;
   char *stage(size_t n)
   {
      if (d_stage_len < n)
      {
         if (d_stage) { (void) hipFree(d_stage); }
         size_t len = n < (1u << 20) ? (1u << 20) : n;
         if (hipMalloc((void **) &d_stage, len) != hipSuccess) { return nullptr; }
         d_stage_len = len;
      }
      return d_stage;
   }
};

int rccl_exchange(void *vctx, int ns, const int *dest, void *const *sbuf, const size_t *sbytes,
                  int nr, const int *src, void *const *rbuf, const size_t *rbytes, int on_device, void *vstream)
{
   RcclCtx *c = (RcclCtx *) vctx;
   hipStream_t s = (hipStream_t) vstream;
   if (!s) { s = hamd::handle().comm_stream; }
   if (on_device)
   {
      NCCL_CHECK(ncclGroupStart());
      for (int i = 0; i < nr; i++) { if (rbytes[i]) NCCL_CHECK(ncclRecv(rbuf[i], rbytes[i], ncclChar, src[i], c->comm, s)); }
      for (int i = 0; i < ns; i++) { if (sbytes[i]) NCCL_CHECK(ncclSend(sbuf[i], sbytes[i], ncclChar, dest[i], c->comm, s)); }
      NCCL_CHECK(ncclGroupEnd());
      return 0;
   }
   // host buffers: stage through device memory
   size_t tot_s = 0, tot_r = 0;
   for (int i = 0; i < ns; i++) { tot_s += (sbytes[i] + 15) & ~(size_t) 15; }
   for (int i = 0; i < nr; i++) { tot_r += (rbytes[i] + 15) & ~(size_t) 15; }
   char *st = c->stage(tot_s + tot_r + 16);
   if (!st) { hypre_error_w_msg(HYPRE_ERROR_MEMORY, "RCCL staging allocation failed"); return 1; }
   size_t off = 0;
   std::vector<char *> ds((size_t) ns), dr((size_t) nr);
   for (int i = 0; i < ns; i++)
   {
      ds[(size_t) i] = st + off;
      if (sbytes[i]) { HIP_CHECK(hipMemcpyAsync(st + off, sbuf[i], sbytes[i], hipMemcpyHostToDevice, s)); }
      off += (sbytes[i] + 15) & ~(size_t) 15;
   }
   for (int i = 0; i < nr; i++) { dr[(size_t) i] = st + off; off += (rbytes[i] + 15) & ~(size_t) 15; }
   NCCL_CHECK(ncclGroupStart());
   for (int i = 0; i < nr; i++) { if (rbytes[i]) NCCL_CHECK(ncclRecv(dr[(size_t) i], rbytes[i], ncclChar, src[i], c->comm, s)); }
   for (int i = 0; i < ns; i++) { if (sbytes[i]) NCCL_CHECK(ncclSend(ds[(size_t) i], sbytes[i], ncclChar, dest[i], c->comm, s)); }
   NCCL_CHECK(ncclGroupEnd());
   for (int i = 0; i < nr; i++)
   {
      if (rbytes[i]) { HIP_CHECK(hipMemcpyAsync(rbuf[i], dr[(size_t) i], rbytes[i], hipMemcpyDeviceToHost, s)); }
   }
   HIP_CHECK(hipStreamSynchronize(s));
   return 0;
}

int rccl_allreduce(void *vctx, double *buf, int count, int on_device, void *vstream)
{
   RcclCtx *c = (RcclCtx *) vctx;
   hipStream_t s = (hipStream_t) vstream;
   if (!s) { s = hamd::handle().comm_stream; }
   if (on_device)
   {
      NCCL_CHECK(ncclAllReduce(buf, buf, (size_t) count, ncclDouble, ncclSum, c->comm, s));
      return 0;
   }
   char *st = c->stage(sizeof(double) * (size_t) count);
   if (!st) { return 1; }
   HIP_CHECK(hipMemcpyAsync(st, buf, sizeof(double) * (size_t) count, hipMemcpyHostToDevice, s));
   NCCL_CHECK(ncclAllReduce(st, st, (size_t) count, ncclDouble, ncclSum, c->comm, s));
   HIP_CHECK(hipMemcpyAsync(buf, st, sizeof(double) * (size_t) count, hipMemcpyDeviceToHost, s));
   HIP_CHECK(hipStreamSynchronize(s));
   return 0;
}

int rccl_allgather(void *vctx, const void *sbuf, void *rbuf, size_t bytes)
{
   RcclCtx *c = (RcclCtx *) vctx;
   hipStream_t s = hamd::handle().comm_stream;
   const size_t pad = (bytes + 15) & ~(size_t) 15;
   char *st = c->stage(pad * (size_t) (c->size + 1));
   if (!st) { return 1; }
   char *d_send = st, *d_recv = st + pad;
   HIP_CHECK(hipMemcpyAsync(d_send, sbuf, bytes, hipMemcpyHostToDevice, s));
   NCCL_CHECK(ncclAllGather(d_send, d_recv, pad, ncclChar, c->comm, s));
   std::vector<char> tmp(pad * (size_t) c->size);
   HIP_CHECK(hipMemcpyAsync(tmp.data(), d_recv, pad * (size_t) c->size, hipMemcpyDeviceToHost, s));
   HIP_CHECK(hipStreamSynchronize(s));
   for (int r = 0; r < c->size; r++) { memcpy((char *) rbuf + bytes * (size_t) r, tmp.data() + pad * (size_t) r, bytes); }
   return 0;
}

int rccl_barrier(void *vctx)
{
   double one = 1.0;
   return rccl_allreduce(vctx, &one, 1, 0, nullptr);
}

void rccl_destroy(void *vctx)
{
   RcclCtx *c = (RcclCtx *) vctx;
   if (c->comm) { ncclCommDestroy(c->comm); }
   if (c->d_stage) { (void) hipFree(c->d_stage); }
   delete c;
}


// ---------------------------------------------------------------------------
// Stream-staged provider: device-buffer semantics on top of a host-buffer transport.
//
// What an application whose MPI cannot take device pointers plugs in (the reference's own device path
// stages its halo buffers the same way, par_csr_communication.c:483-526 with hypre_TMemcpy around the
// MPI calls) — and what lets the library's device-buffer flow (pack -> event -> exchange on the
// communication stream -> event -> ghost product) run between ranks that share one GPU, where RCCL
// refuses to form a communicator.  exchange(on_device = 1, stream):
//    device -> pinned host on `stream`; wait for `stream` (and with it for everything the caller
//    ordered before the exchange, i.e. the pack event); inner host exchange; pinned host -> device
//    enqueued on `stream`, NOT waited for: the caller's event after the call covers it.
// ---------------------------------------------------------------------------
struct StagedCtx
{
   hypre_amd_CommOps inner{};
   char  *h_stage = nullptr;     // pinned
   size_t h_len = 0;
   hipEvent_t done = nullptr;    // recorded behind the last host -> device copies: the staging buffer is busy until then
   char *stage(size_t n)
   {
      if (done) { (void) hipEventSynchronize(done); }
      if (h_len < n)
      {
         if (h_stage) { (void) hipHostFree(h_stage); }
         size_t len = n < (1u << 16) ? (1u << 16) : n;
         if (hipHostMalloc((void **) &h_stage, len, hipHostMallocDefault) != hipSuccess) { h_stage = nullptr; h_len = 0; return nullptr; }
         h_len = len;
      }
      return h_stage;
   }
};

int staged_exchange(void *vctx, int ns, const int *dest, void *const *sbuf, const size_t *sbytes,
                    int nr, const int *src, void *const *rbuf, const size_t *rbytes, int on_device, void *vstream)
{
   StagedCtx *c = (StagedCtx *) vctx;
   if (!on_device) { return c->inner.exchange(c->inner.ctx, ns, dest, sbuf, sbytes, nr, src, rbuf, rbytes, 0, nullptr); }
   hipStream_t s = (hipStream_t) vstream;
   if (!s) { s = hamd::handle().comm_stream; }
   size_t tot = 0;
   for (int i = 0; i < ns; i++) { tot += (sbytes[i] + 15) & ~(size_t) 15; }
   for (int i = 0; i < nr; i++) { tot += (rbytes[i] + 15) & ~(size_t) 15; }
   char *st = c->stage(tot + 16);
   if (!st) { hypre_error_w_msg(HYPRE_ERROR_MEMORY, "stream-staged transport: pinned staging allocation failed"); return 1; }
   std::vector<void *> hs((size_t) ns), hr((size_t) nr);
   size_t off = 0;
   for (int i = 0; i < ns; i++)
   {
      hs[(size_t) i] = st + off;
      if (sbytes[i]) { HIP_CHECK(hipMemcpyAsync(st + off, sbuf[i], sbytes[i], hipMemcpyDeviceToHost, s)); }
      off += (sbytes[i] + 15) & ~(size_t) 15;
   }
   for (int i = 0; i < nr; i++) { hr[(size_t) i] = st + off; off += (rbytes[i] + 15) & ~(size_t) 15; }
   HIP_CHECK(hipStreamSynchronize(s));
   const int rc = c->inner.exchange(c->inner.ctx, ns, dest, hs.data(), sbytes, nr, src, hr.data(), rbytes, 0, nullptr);
   for (int i = 0; i < nr; i++)
   {
      if (rbytes[i]) { HIP_CHECK(hipMemcpyAsync(rbuf[i], hr[(size_t) i], rbytes[i], hipMemcpyHostToDevice, s)); }
   }
   if (!c->done) { HIP_CHECK(hipEventCreateWithFlags(&c->done, hipEventDisableTiming)); }
   HIP_CHECK(hipEventRecord(c->done, s));
   return rc;
}

int staged_allreduce(void *vctx, double *buf, int count, int on_device, void *vstream)
{
   StagedCtx *c = (StagedCtx *) vctx;
   if (!on_device) { return c->inner.allreduce_sum(c->inner.ctx, buf, count, 0, nullptr); }
   hipStream_t s = (hipStream_t) vstream;
   if (!s) { s = hamd::handle().comm_stream; }
   const size_t bytes = sizeof(double) * (size_t) count;
   char *st = c->stage(bytes + 16);
   if (!st) { return 1; }
   HIP_CHECK(hipMemcpyAsync(st, buf, bytes, hipMemcpyDeviceToHost, s));
   HIP_CHECK(hipStreamSynchronize(s));
   const int rc = c->inner.allreduce_sum(c->inner.ctx, (double *) st, count, 0, nullptr);
   HIP_CHECK(hipMemcpyAsync(buf, st, bytes, hipMemcpyHostToDevice, s));
   if (!c->done) { HIP_CHECK(hipEventCreateWithFlags(&c->done, hipEventDisableTiming)); }
   HIP_CHECK(hipEventRecord(c->done, s));
   return rc;
}

int staged_allgather(void *vctx, const void *sbuf, void *rbuf, size_t bytes)
{
   StagedCtx *c = (StagedCtx *) vctx;
   return c->inner.allgather(c->inner.ctx, sbuf, rbuf, bytes);
}

int staged_barrier(void *vctx)
{
   StagedCtx *c = (StagedCtx *) vctx;
   return c->inner.barrier ? c->inner.barrier(c->inner.ctx) : 0;
}

void staged_destroy(void *vctx)
{
   StagedCtx *c = (StagedCtx *) vctx;
   if (c->done) { (void) hipEventSynchronize(c->done); (void) hipEventDestroy(c->done); }
   if (c->h_stage) { (void) hipHostFree(c->h_stage); }
   delete c;      // the inner communicator stays the caller's
}

}  // namespace

namespace hamd {
const hypre_amd_CommOps *comm_ops(MPI_Comm comm)
{
   auto &t = table();
   if (comm < 0 || (size_t) comm >= t.size() || !t[(size_t) comm].live) { return nullptr; }
   return &t[(size_t) comm].ops;
}
}  // namespace hamd

extern "C" {

MPI_Comm hypre_amd_CommCreate(const hypre_amd_CommOps *ops)
{
   auto &t = table();
   CommObj o;
   o.live = true;
   o.ops = *ops;
   for (size_t k = 1; k < t.size(); k++)
   {
      if (!t[k].live) { t[k] = o; return (MPI_Comm) k; }
   }
   t.push_back(o);
   return (MPI_Comm) (t.size() - 1);
}

HYPRE_Int hypre_amd_CommDestroy(MPI_Comm comm)
{
   auto &t = table();
   if (comm <= 0 || (size_t) comm >= t.size() || !t[(size_t) comm].live) { return hypre_error_flag; }
   if (t[(size_t) comm].ops.destroy) { t[(size_t) comm].ops.destroy(t[(size_t) comm].ops.ctx); }
   t[(size_t) comm] = CommObj();
   return hypre_error_flag;
}

HYPRE_Int hypre_amd_RCCLGetUniqueId(void *id_out)
{
   static_assert(sizeof(ncclUniqueId) == HYPRE_AMD_RCCL_ID_BYTES, "RCCL id size");
   ncclUniqueId id;
   if (ncclGetUniqueId(&id) != ncclSuccess)
   {
      hypre_error_w_msg(HYPRE_ERROR_GENERIC, "ncclGetUniqueId failed");
      return hypre_error_flag;
   }
   memcpy(id_out, &id, sizeof(id));
   return hypre_error_flag;
}

MPI_Comm hypre_amd_CommCreateRCCL(const void *id_bytes, int rank, int size)
{
   if (!hamd::ensure_device())
   {
      hypre_error_w_msg(HYPRE_ERROR_GENERIC, "hypre_amd_CommCreateRCCL: no HIP device");
      return hypre_MPI_COMM_NULL;
   }
   ncclUniqueId id;
   memcpy(&id, id_bytes, sizeof(id));
   RcclCtx *c = new RcclCtx();
   c->rank = rank; c->size = size;
   if (ncclCommInitRank(&c->comm, size, id, rank) != ncclSuccess)
   {
      hypre_error_w_msg(HYPRE_ERROR_GENERIC, "ncclCommInitRank failed");
      delete c;
      return hypre_MPI_COMM_NULL;
   }
   hypre_amd_CommOps ops{};
   ops.ctx = c; ops.rank = rank; ops.size = size;
   ops.exchange = rccl_exchange;
   ops.allreduce_sum = rccl_allreduce;
   ops.allgather = rccl_allgather;
   ops.barrier = rccl_barrier;
   ops.destroy = rccl_destroy;
   ops.device_buffers = 1;
   return hypre_amd_CommCreate(&ops);
}

MPI_Comm hypre_amd_CommCreateStreamStaged(MPI_Comm inner)
{
   const hypre_amd_CommOps *io = hamd::comm_ops(inner);
   if (!io || !io->exchange || !io->allreduce_sum || !io->allgather)
   {
      hypre_error_w_msg(HYPRE_ERROR_ARG, "hypre_amd_CommCreateStreamStaged: the inner communicator lacks a host transport");
      return hypre_MPI_COMM_NULL;
   }
   if (!hamd::ensure_device())
   {
      hypre_error_w_msg(HYPRE_ERROR_GENERIC, "hypre_amd_CommCreateStreamStaged: no HIP device");
      return hypre_MPI_COMM_NULL;
   }
   StagedCtx *c = new StagedCtx();
   c->inner = *io;
   hypre_amd_CommOps ops{};
   ops.ctx = c; ops.rank = io->rank; ops.size = io->size;
   ops.exchange = staged_exchange;
   ops.allreduce_sum = staged_allreduce;
   ops.allgather = staged_allgather;
   ops.barrier = staged_barrier;
   ops.destroy = staged_destroy;
   ops.device_buffers = 1;
   return hypre_amd_CommCreate(&ops);
}

// Round trip over every entry of a communicator's table: a ring shift of a byte
// pattern (host buffers, then device buffers when the provider takes them), a
// sum all-reduce and an all-gather, each checked against the expected answer.
// Returns the number of failed checks (0 = healthy).  Collective.
HYPRE_Int hypre_amd_CommSelfTest(MPI_Comm comm, HYPRE_Int nbytes)
{
   const hypre_amd_CommOps *o = hamd::comm_ops(comm);
   if (!o) { return 1; }
   if (!o->exchange || !o->allreduce_sum || !o->allgather) { return o->size > 1 ? 1 : 0; }
   const int    rank = o->rank, size = o->size;
   const size_t n = nbytes > 0 ? (size_t) nbytes : 1;
   int          next = (rank + 1) % size, prev = (rank + size - 1) % size;
   int          fails = 0;
   auto pattern = [](int r, size_t i) { return (unsigned char) ((i * 131u + 7u * (unsigned) r + 1u) & 255u); };

   std::vector<unsigned char> hs(n), hr(n, 0);
   for (size_t i = 0; i < n; i++) { hs[i] = pattern(rank, i); }
   {
      void *sb = hs.data(), *rb = hr.data();
      size_t nb = n;
      if (o->exchange(o->ctx, 1, &next, &sb, &nb, 1, &prev, &rb, &nb, 0, nullptr)) { fails++; }
      for (size_t i = 0; i < n; i++) { if (hr[i] != pattern(prev, i)) { fails++; break; } }
   }
   if (o->device_buffers && hamd::ensure_device())
   {
      hipStream_t s = hamd::handle().comm_stream;
      unsigned char *ds = nullptr, *dr = nullptr;
      if (hipMalloc((void **) &ds, n) != hipSuccess || hipMalloc((void **) &dr, n) != hipSuccess) { return fails + 1; }
      (void) hipMemcpyAsync(ds, hs.data(), n, hipMemcpyHostToDevice, s);
      (void) hipMemsetAsync(dr, 0, n, s);
      void *sb = ds, *rb = dr;
      size_t nb = n;
      if (o->exchange(o->ctx, 1, &next, &sb, &nb, 1, &prev, &rb, &nb, 1, s)) { fails++; }
      std::fill(hr.begin(), hr.end(), 0);
      (void) hipMemcpyAsync(hr.data(), dr, n, hipMemcpyDeviceToHost, s);
      if (hipStreamSynchronize(s) != hipSuccess) { fails++; }
      for (size_t i = 0; i < n; i++) { if (hr[i] != pattern(prev, i)) { fails++; break; } }

      double h2[2] = {1.0, (double) rank}, *d2 = (double *) ds;
      if (n >= sizeof(h2))
      {
         (void) hipMemcpyAsync(d2, h2, sizeof(h2), hipMemcpyHostToDevice, s);
         if (o->allreduce_sum(o->ctx, d2, 2, 1, s)) { fails++; }
         (void) hipMemcpyAsync(h2, d2, sizeof(h2), hipMemcpyDeviceToHost, s);
         if (hipStreamSynchronize(s) != hipSuccess) { fails++; }
         if (h2[0] != (double) size || h2[1] != 0.5 * size * (size - 1)) { fails++; }
      }
      (void) hipFree(ds);
      (void) hipFree(dr);
   }
   {
      double h2[2] = {1.0, (double) rank};
      if (o->allreduce_sum(o->ctx, h2, 2, 0, nullptr)) { fails++; }
      if (h2[0] != (double) size || h2[1] != 0.5 * size * (size - 1)) { fails++; }
   }
   {
      std::vector<int> all((size_t) size, -1);
      int mine = 1000 + rank;
      if (o->allgather(o->ctx, &mine, all.data(), sizeof(int))) { fails++; }
      for (int r = 0; r < size; r++) { if (all[(size_t) r] != 1000 + r) { fails++; break; } }
   }
   return fails;
}

HYPRE_Int hypre_MPI_Comm_rank(MPI_Comm comm, HYPRE_Int *rank)
{
   const hypre_amd_CommOps *o = hamd::comm_ops(comm);
   *rank = o ? o->rank : 0;
   return 0;
}
HYPRE_Int hypre_MPI_Comm_size(MPI_Comm comm, HYPRE_Int *size)
{
   const hypre_amd_CommOps *o = hamd::comm_ops(comm);
   *size = o ? o->size : 1;
   return 0;
}
HYPRE_Int hypre_MPI_Barrier(MPI_Comm comm)
{
   const hypre_amd_CommOps *o = hamd::comm_ops(comm);
   if (o && o->size > 1 && o->barrier) { return o->barrier(o->ctx); }
   return 0;
}

}  // extern "C"
