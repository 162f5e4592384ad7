// hypre_amd — distributed CSR matrix / vector objects, the halo-exchange
// package, and the ParCSR matrix-vector products (one rank per GPU).
//
// Reference counterparts:
//   parcsr_mv/par_csr_matrix.c:37-330          object life cycle, migrate
//   parcsr_mv/par_vector.c:28-575              ParVector + BLAS-1 wrappers
//   parcsr_mv/par_csr_communication.c:358-699  CommHandleCreate_v2 / Destroy
//   parcsr_mv/par_csr_communication.c:713-943  CommPkgCreate_core
//   parcsr_mv/par_csr_matvec.c:21-546          Matvec / MatvecT (host flow)
//   parcsr_mv/par_csr_matvec_device.c:25-591   Matvec / MatvecT (device flow)
//
// Device flow of y = alpha*A*x + beta*b (halo overlapped with the interior
// product; no host synchronisation inside):
//   compute stream : pack x[send_map] -> buf            ; record ev_pack
//   comm stream    : wait ev_pack ; grouped send/recv   ; record ev_halo
//   compute stream : y = alpha*diag*x + beta*b          (interior, overlaps)
//   compute stream : wait ev_halo ; y += alpha*offd*x_ghost
#include "internal.hpp"
#include <algorithm>
#include <omp.h>

using namespace hamd;

namespace hamd {
// traffic counters of this process (reported by the benchmark): neighbour exchanges and all-reduces started
static long long g_exchanges = 0, g_allreduces = 0, g_exchange_bytes = 0, g_allreduce_bytes = 0;
}
extern "C" HYPRE_Int hypre_amd_CommCounters(HYPRE_BigInt *exchanges, HYPRE_BigInt *allreduces, HYPRE_Int reset)
{
   if (exchanges) { *exchanges = hamd::g_exchanges; }
   if (allreduces) { *allreduces = hamd::g_allreduces; }
   if (reset) { hamd::g_exchanges = 0; hamd::g_allreduces = 0; hamd::g_exchange_bytes = 0; hamd::g_allreduce_bytes = 0; }
   return hypre_error_flag;
}
// bytes this process SENT in neighbour exchanges / contributed to all-reduces since the last reset of the counters above
extern "C" HYPRE_Int hypre_amd_CommBytes(HYPRE_BigInt *exchange_bytes, HYPRE_BigInt *allreduce_bytes)
{
   if (exchange_bytes) { *exchange_bytes = hamd::g_exchange_bytes; }
   if (allreduce_bytes) { *allreduce_bytes = hamd::g_allreduce_bytes; }
   return hypre_error_flag;
}
// shape of a matrix's halo exchange (its communication package, built on demand): neighbours this rank sends to /
// receives from, entries sent / received per exchange
extern "C" HYPRE_Int hypre_amd_ParCSRMatrixHaloInfo(hypre_ParCSRMatrix *A, HYPRE_Int *num_sends, HYPRE_Int *send_entries,
                                                    HYPRE_Int *num_recvs, HYPRE_Int *recv_entries)
{
   HYPRE_Int np;
   hypre_MPI_Comm_size(A->comm, &np);
   if (np > 1 && !A->comm_pkg) { hypre_MatvecCommPkgCreate(A); }
   hypre_ParCSRCommPkg *pk = A->comm_pkg;
   if (num_sends) { *num_sends = pk ? pk->num_sends : 0; }
   if (send_entries) { *send_entries = pk ? pk->send_map_starts[pk->num_sends] : 0; }
   if (num_recvs) { *num_recvs = pk ? pk->num_recvs : 0; }
   if (recv_entries) { *recv_entries = pk ? pk->recv_vec_starts[pk->num_recvs] : 0; }
   return hypre_error_flag;
}
namespace hamd {
double global_sum(MPI_Comm comm, double v)
{
   const hypre_amd_CommOps *o = comm_ops(comm);
   if (o && o->size > 1 && o->allreduce_sum) { g_allreduces++; o->allreduce_sum(o->ctx, &v, 1, 0, nullptr); }
   return v;
}
}  // namespace hamd

namespace {

// Event pairs that order the compute and the communication stream around one exchange.  Every exchange in
// flight owns a pair (handed out round-robin from a pool deep enough for the few a cycle can have open at once,
// and remembered in its hypre_ParCSRCommHandle): `pack` is recorded on the compute stream when the send buffer is
// ready, `halo` on the communication stream when the transfers have been enqueued.
struct EventPair
{
   hipEvent_t pack = nullptr, halo = nullptr;
   // diagnosis (hypre_amd_CommSetTiming): `ready` is recorded on the compute stream where it is about to wait for `halo` —
   // what lies between the two is the time the exchange was NOT hidden behind the interior work
   hipEvent_t ready = nullptr;
   int tag = -1, reduce = 0;
   double host_us = 0.0;        // wall-clock time the HOST spent inside the transport's call (RCCL: an enqueue; a host-staged
                                // transport: the whole transfer, during which it enqueues nothing else)
};
constexpr int EVENT_POOL = 16;
// Diagnosis mode: every exchange (and device all-reduce) gets timing events of its own, kept until they are read out.
int g_comm_timing = 0, g_comm_tag = -1;
std::vector<EventPair *> g_timed;
EventPair *next_event_pair()
{
   if (g_comm_timing && g_timed.size() < 65536)      // (a caller that never reads the times out gets the pool back after 65 536 exchanges)
   {
      EventPair *e = new EventPair();
      HIP_CHECK(hipEventCreate(&e->pack)); HIP_CHECK(hipEventCreate(&e->halo)); HIP_CHECK(hipEventCreate(&e->ready));
      e->tag = g_comm_tag;
      g_timed.push_back(e);
      return e;
   }
   static EventPair pool[EVENT_POOL];
   static int next = 0;
   EventPair *e = &pool[next];
   next = (next + 1) % EVENT_POOL;
   if (!e->pack)
   {
      HIP_CHECK(hipEventCreateWithFlags(&e->pack, hipEventDisableTiming));
      HIP_CHECK(hipEventCreateWithFlags(&e->halo, hipEventDisableTiming));
   }
   return e;
}

}  // namespace

// Diagnosis of the overlap (bench.py --gpus N): with timing on, every device-buffer exchange and device all-reduce from now
// on carries timing events — buffer packed (compute stream), transfers done (communication stream), compute stream about
// to wait for them — tagged with the AMG level the cycle is on (hypre_amd_CommSetTag; -1 outside a cycle).
extern "C" HYPRE_Int hypre_amd_CommSetTiming(HYPRE_Int on)
{
   g_comm_timing = on != 0;
   return hypre_error_flag;
}
extern "C" HYPRE_Int hypre_amd_CommSetTag(HYPRE_Int tag)
{
   g_comm_tag = tag;
   return hypre_error_flag;
}
// Reads out and forgets the timed exchanges since the last call: per tag t in [0, max_tags) (tag -1 and tags beyond go to
// the last slot) the number of exchanges, of all-reduces, the EXPOSED time — compute stream waiting at the halo event: the
// part of the transfer that the interior product / sweep did not hide — and the whole time from "send buffer packed" to
// "transfers done", and the wall-clock time the host spent inside the transport's calls (an enqueue for RCCL; the whole
// transfer for a host-staged transport, which enqueues nothing meanwhile), in microseconds.  Synchronises both streams.
extern "C" HYPRE_Int hypre_amd_CommExposedTimes(HYPRE_Int max_tags, HYPRE_Int *exchanges, HYPRE_Int *allreduces, HYPRE_Real *exposed_us,
                                                HYPRE_Real *transfer_us, HYPRE_Real *host_us)
{
   Handle &hd = handle();
   if (hd.compute_stream) { HIP_CHECK(hipStreamSynchronize(hd.compute_stream)); }
   if (hd.comm_stream) { HIP_CHECK(hipStreamSynchronize(hd.comm_stream)); }
   for (HYPRE_Int t = 0; t < max_tags; t++) { exchanges[t] = 0; allreduces[t] = 0; exposed_us[t] = 0.0; transfer_us[t] = 0.0; if (host_us) { host_us[t] = 0.0; } }
   for (EventPair *e : g_timed)
   {
      const int t = (e->tag >= 0 && e->tag < max_tags - 1) ? e->tag : max_tags - 1;
      float gap = 0.f, whole = 0.f;
      const bool have_ready = hipEventQuery(e->ready) == hipSuccess;     // (an exchange whose handle was never destroyed has none)
      (void) hipGetLastError();
      if (max_tags > 0)
      {
         if (have_ready && hipEventElapsedTime(&gap, e->ready, e->halo) != hipSuccess) { (void) hipGetLastError(); gap = 0.f; }
         if (hipEventElapsedTime(&whole, e->pack, e->halo) != hipSuccess) { (void) hipGetLastError(); whole = 0.f; }
         if (e->reduce) { allreduces[t]++; } else { exchanges[t]++; }
         exposed_us[t] += 1e3 * (gap > 0.f ? gap : 0.f);
         transfer_us[t] += 1e3 * (whole > 0.f ? whole : 0.f);
         if (host_us) { host_us[t] += e->host_us; }
      }
      HIP_CHECK(hipEventDestroy(e->pack)); HIP_CHECK(hipEventDestroy(e->halo)); HIP_CHECK(hipEventDestroy(e->ready));
      delete e;
   }
   g_timed.clear();
   return hypre_error_flag;
}

namespace hamd {
// In-place sum over the ranks of `comm` of n doubles in DEVICE memory, ordered behind the work already queued on the
// compute stream and ahead of what is queued after the call; no host synchronisation when the transport takes
// device buffers.  All traffic of a communicator goes through the communication stream (one stream per RCCL
// communicator: halo exchanges and reductions never race on it).
void dev_allreduce_sum(MPI_Comm comm, double *d_buf, int n)
{
   const hypre_amd_CommOps *o = comm_ops(comm);
   if (!o || o->size <= 1 || n <= 0) { return; }
   g_allreduces++;
   g_allreduce_bytes += 8LL * n;
   Handle &hd = handle();
   if (o->device_buffers)
   {
      EventPair *ev = next_event_pair();
      HIP_CHECK(hipEventRecord(ev->pack, hd.compute_stream));
      HIP_CHECK(hipStreamWaitEvent(hd.comm_stream, ev->pack, 0));
      const double t_host = ev->ready ? omp_get_wtime() : 0.0;
      o->allreduce_sum(o->ctx, d_buf, n, 1, (void *) hd.comm_stream);
      if (ev->ready) { ev->host_us = 1e6 * (omp_get_wtime() - t_host); }
      HIP_CHECK(hipEventRecord(ev->halo, hd.comm_stream));
      if (ev->ready) { ev->reduce = 1; HIP_CHECK(hipEventRecord(ev->ready, hd.compute_stream)); }
      HIP_CHECK(hipStreamWaitEvent(hd.compute_stream, ev->halo, 0));
   }
   else
   {
      std::vector<double> h((size_t) n);
      hypre_TMemcpy(h.data(), d_buf, double, (size_t) n, HYPRE_MEMORY_HOST, HYPRE_MEMORY_DEVICE);
      o->allreduce_sum(o->ctx, h.data(), n, 0, nullptr);
      hypre_TMemcpy(d_buf, h.data(), double, (size_t) n, HYPRE_MEMORY_DEVICE, HYPRE_MEMORY_HOST);
   }
}

// Sum over the ranks of n (<= 8) doubles that sit in device memory, returned on the host: one device all-reduce and
// one read-back when the transport takes device buffers (otherwise read back, then reduce on the host).
void dev_global_sums(MPI_Comm comm, double *d_vals, int n, double *h_out)
{
   const hypre_amd_CommOps *o = comm_ops(comm);
   Handle &hd = handle();
   double *h = hd.h_reduce;
   if (o && o->size > 1 && o->device_buffers) { dev_allreduce_sum(comm, d_vals, n); }
   HIP_CHECK(hipMemcpyAsync(h, d_vals, sizeof(double) * (size_t) n, hipMemcpyDeviceToHost, hd.compute_stream));
   HIP_CHECK(hipStreamSynchronize(hd.compute_stream));
   for (int k = 0; k < n; k++) { h_out[k] = h[k]; }
   if (o && o->size > 1 && !o->device_buffers && o->allreduce_sum) { o->allreduce_sum(o->ctx, h_out, n, 0, nullptr); }
}
}  // namespace hamd

extern "C" {

// ===========================================================================
// ParCSR matrix object
// ===========================================================================
static void local_partition(HYPRE_BigInt length, HYPRE_Int num_procs, HYPRE_Int myid, HYPRE_BigInt *part)
{
   // seq_mv/genpart.c:50-75: even split, the first `rest` ranks get one extra
   const HYPRE_BigInt size = length / num_procs;
   const HYPRE_BigInt rest = length - size * num_procs;
   part[0] = size * myid + std::min<HYPRE_BigInt>(myid, rest);
   part[1] = size * (myid + 1) + std::min<HYPRE_BigInt>(myid + 1, rest);
}

hypre_ParCSRMatrix *hypre_ParCSRMatrixCreate(MPI_Comm comm, HYPRE_BigInt global_num_rows,
                                             HYPRE_BigInt global_num_cols, HYPRE_BigInt *row_starts_in,
                                             HYPRE_BigInt *col_starts_in, HYPRE_Int num_cols_offd,
                                             HYPRE_Int num_nonzeros_diag, HYPRE_Int num_nonzeros_offd)
{
   hypre_ParCSRMatrix *m = (hypre_ParCSRMatrix *) calloc(1, sizeof(hypre_ParCSRMatrix));
   HYPRE_Int num_procs, my_id;
   hypre_MPI_Comm_rank(comm, &my_id);
   hypre_MPI_Comm_size(comm, &num_procs);
   HYPRE_BigInt rs[2], cs[2];
   if (row_starts_in) { rs[0] = row_starts_in[0]; rs[1] = row_starts_in[1]; }
   else { local_partition(global_num_rows, num_procs, my_id, rs); }
   if (col_starts_in) { cs[0] = col_starts_in[0]; cs[1] = col_starts_in[1]; }
   else { local_partition(global_num_cols, num_procs, my_id, cs); }
   const HYPRE_Int nr = (HYPRE_Int) (rs[1] - rs[0]);
   const HYPRE_Int nc = (HYPRE_Int) (cs[1] - cs[0]);
   m->comm = comm;
   m->diag = hypre_CSRMatrixCreate(nr, nc, num_nonzeros_diag);
   m->offd = hypre_CSRMatrixCreate(nr, num_cols_offd, num_nonzeros_offd);
   m->global_num_rows = global_num_rows;
   m->global_num_cols = global_num_cols;
   m->global_num_rownnz = global_num_rows;
   m->num_nonzeros = -1;
   m->d_num_nonzeros = -1.0;
   m->first_row_index = rs[0];
   m->first_col_diag = cs[0];
   m->last_row_index = rs[0] + nr - 1;
   m->last_col_diag = cs[0] + nc - 1;
   m->row_starts[0] = rs[0]; m->row_starts[1] = rs[1];
   m->col_starts[0] = cs[0]; m->col_starts[1] = cs[1];
   m->owns_data = 1;
   m->owns_assumed_partition = 1;
   m->bdiag_size = -1;
   return m;
}

HYPRE_Int hypre_ParCSRMatrixInitialize_v2(hypre_ParCSRMatrix *m, HYPRE_MemoryLocation loc)
{
   hypre_CSRMatrixInitialize_v2(m->diag, 0, loc);
   hypre_CSRMatrixInitialize_v2(m->offd, 0, loc);
   if (!m->col_map_offd && m->offd->num_cols)
   {
      m->col_map_offd = hypre_CTAlloc(HYPRE_BigInt, m->offd->num_cols, HYPRE_MEMORY_HOST);
   }
   return hypre_error_flag;
}

HYPRE_Int hypre_ParCSRMatrixDestroy(hypre_ParCSRMatrix *m)
{
   if (!m) { return hypre_error_flag; }
   if (m->owns_data)
   {
      hypre_CSRMatrixDestroy(m->diag);
      hypre_CSRMatrixDestroy(m->offd);
      hypre_Free(m->col_map_offd, HYPRE_MEMORY_HOST);
      hypre_Free(m->device_col_map_offd, HYPRE_MEMORY_DEVICE);
      if (m->comm_pkg) { hypre_MatvecCommPkgDestroy(m->comm_pkg); }
      if (m->comm_pkgT) { hypre_MatvecCommPkgDestroy(m->comm_pkgT); }
   }
   if (m->diagT) { hypre_CSRMatrixDestroy(m->diagT); }
   if (m->offdT) { hypre_CSRMatrixDestroy(m->offdT); }
   free(m);
   return hypre_error_flag;
}

HYPRE_Int hypre_ParCSRMatrixMigrate(hypre_ParCSRMatrix *A, HYPRE_MemoryLocation loc)
{
   if (!A) { return hypre_error_flag; }
   hypre_CSRMatrixMigrate(A->diag, loc);
   hypre_CSRMatrixMigrate(A->offd, loc);
   if (A->diagT) { hypre_CSRMatrixMigrate(A->diagT, loc); }
   if (A->offdT) { hypre_CSRMatrixMigrate(A->offdT, loc); }
   return hypre_error_flag;
}

hypre_ParCSRMatrix *hypre_ParCSRMatrixClone_v2(hypre_ParCSRMatrix *A, HYPRE_Int copy_data,
                                               HYPRE_MemoryLocation loc)
{
   hypre_ParCSRMatrix *B = hypre_ParCSRMatrixCreate(A->comm, A->global_num_rows, A->global_num_cols,
                                                    A->row_starts, A->col_starts, A->offd->num_cols,
                                                    A->diag->num_nonzeros, A->offd->num_nonzeros);
   hypre_CSRMatrixDestroy(B->diag);
   hypre_CSRMatrixDestroy(B->offd);
   B->diag = hypre_CSRMatrixClone_v2(A->diag, copy_data, loc);
   B->offd = hypre_CSRMatrixClone_v2(A->offd, copy_data, loc);
   B->num_nonzeros = A->num_nonzeros;
   B->d_num_nonzeros = A->d_num_nonzeros;
   if (A->offd->num_cols)
   {
      B->col_map_offd = hypre_TAlloc(HYPRE_BigInt, A->offd->num_cols, HYPRE_MEMORY_HOST);
      memcpy(B->col_map_offd, A->col_map_offd, sizeof(HYPRE_BigInt) * (size_t) A->offd->num_cols);
   }
   return B;
}


HYPRE_Int hypre_ParCSRMatrixSetDNumNonzeros(hypre_ParCSRMatrix *m)
{
   const double local = (double) m->diag->num_nonzeros + (double) m->offd->num_nonzeros;
   m->d_num_nonzeros = global_sum(m->comm, local);
   return hypre_error_flag;
}

HYPRE_Int hypre_ParCSRMatrixSetNumNonzeros(hypre_ParCSRMatrix *m)
{
   const double local = (double) m->diag->num_nonzeros + (double) m->offd->num_nonzeros;
   m->num_nonzeros = (HYPRE_BigInt) global_sum(m->comm, local);
   return hypre_error_flag;
}

HYPRE_Int hypre_amd_ParCSRMatrixKeepTranspose(hypre_ParCSRMatrix *A)
{
   if (!A->diagT) { hypre_CSRMatrixTranspose(A->diag, &A->diagT, 1); }
   if (!A->offdT && A->offd->num_cols) { hypre_CSRMatrixTranspose(A->offd, &A->offdT, 1); }
   return hypre_error_flag;
}

// ===========================================================================
// ParVector object
// ===========================================================================
hypre_ParVector *hypre_ParVectorCreate(MPI_Comm comm, HYPRE_BigInt global_size, HYPRE_BigInt *partitioning_in)
{
   hypre_ParVector *v = (hypre_ParVector *) calloc(1, sizeof(hypre_ParVector));
   HYPRE_Int num_procs, my_id;
   hypre_MPI_Comm_rank(comm, &my_id);
   hypre_MPI_Comm_size(comm, &num_procs);
   HYPRE_BigInt part[2];
   if (partitioning_in) { part[0] = partitioning_in[0]; part[1] = partitioning_in[1]; }
   else { local_partition(global_size, num_procs, my_id, part); }
   const HYPRE_Int local_size = (HYPRE_Int) (part[1] - part[0]);
   v->comm = comm;
   v->global_size = global_size;
   v->partitioning[0] = part[0];
   v->partitioning[1] = part[1];
   v->first_index = part[0];
   v->last_index = part[1] - 1;
   v->local_vector = hypre_SeqVectorCreate(local_size);
   v->actual_local_size = 0;
   v->owns_data = 1;
   v->all_zeros = 0;
   return v;
}

// par_vector.c:77-87: global_size is the global length of ONE column
hypre_ParVector *hypre_ParMultiVectorCreate(MPI_Comm comm, HYPRE_BigInt global_size, HYPRE_BigInt *partitioning_in, HYPRE_Int num_vectors)
{
   hypre_ParVector *v = hypre_ParVectorCreate(comm, global_size, partitioning_in);
   v->local_vector->num_vectors = num_vectors;
   return v;
}

HYPRE_Int hypre_ParVectorInitialize_v2(hypre_ParVector *v, HYPRE_MemoryLocation loc)
{
   hypre_SeqVectorInitialize_v2(v->local_vector, loc);
   v->actual_local_size = v->local_vector->size;
   return hypre_error_flag;
}

HYPRE_Int hypre_ParVectorInitialize(hypre_ParVector *v)
{
   return hypre_ParVectorInitialize_v2(v, v->local_vector->memory_location);
}

HYPRE_Int hypre_ParVectorDestroy(hypre_ParVector *v)
{
   if (!v) { return hypre_error_flag; }
   if (v->owns_data) { hypre_SeqVectorDestroy(v->local_vector); }
   free(v);
   return hypre_error_flag;
}

// Work vectors are allocated at fine-grid size and re-sized in place per level
// (parcsr_ls/par_cycle.c:290-291); only the logical size changes.
HYPRE_Int hypre_ParVectorSetLocalSize(hypre_ParVector *v, HYPRE_Int local_size)
{
   hypre_Vector *l = v->local_vector;
   l->size = local_size;
   if (l->multivec_storage_method == 0) { l->vecstride = local_size; }
   return hypre_error_flag;
}

HYPRE_Int hypre_ParVectorMigrate(hypre_ParVector *x, HYPRE_MemoryLocation loc)
{
   if (!x) { return hypre_error_flag; }
   // migrate the whole allocation, not just the logical size
   hypre_Vector *l = x->local_vector;
   const HYPRE_Int logical = l->size;
   if (x->actual_local_size > logical) { l->size = x->actual_local_size; }
   hypre_SeqVectorMigrate(l, loc);
   l->size = logical;
   return hypre_error_flag;
}

// ---- BLAS-1 wrappers (par_vector.c:322-575) ----
HYPRE_Int hypre_ParVectorSetConstantValues(hypre_ParVector *v, HYPRE_Complex value)
{
   return hypre_SeqVectorSetConstantValues(v->local_vector, value);
}
HYPRE_Int hypre_ParVectorSetZeros(hypre_ParVector *v)
{
   v->all_zeros = 1;
   return hypre_SeqVectorSetConstantValues(v->local_vector, 0.0);
}
HYPRE_Int hypre_ParVectorCopy(hypre_ParVector *x, hypre_ParVector *y)
{
   return hypre_SeqVectorCopy(x->local_vector, y->local_vector);
}
HYPRE_Int hypre_ParVectorScale(HYPRE_Complex alpha, hypre_ParVector *y)
{
   return hypre_SeqVectorScale(alpha, y->local_vector);
}
HYPRE_Int hypre_ParVectorAxpy(HYPRE_Complex alpha, hypre_ParVector *x, hypre_ParVector *y)
{
   return hypre_SeqVectorAxpy(alpha, x->local_vector, y->local_vector);
}
HYPRE_Int hypre_ParVectorAxpyz(HYPRE_Complex alpha, hypre_ParVector *x, HYPRE_Complex beta,
                               hypre_ParVector *y, hypre_ParVector *z)
{
   return hypre_SeqVectorAxpyz(alpha, x->local_vector, beta, y->local_vector, z->local_vector);
}
HYPRE_Real hypre_ParVectorInnerProd(hypre_ParVector *x, hypre_ParVector *y)
{
   // local dot + one scalar all-reduce (par_vector.c:513-533); with a device-buffer transport the partial sum is
   // reduced where it is and read back once
   hypre_Vector *xl = x->local_vector, *yl = y->local_vector;
   if (xl->memory_location != HYPRE_MEMORY_DEVICE || yl->memory_location != HYPRE_MEMORY_DEVICE)
   {
      hypre_error_w_msg(HYPRE_ERROR_GENERIC, "hypre_ParVectorInnerProd: operand is not in device memory; host execution is not part of this library");
      return 0.0;
   }
   double *d_out = reduce_scratch(2048);
   launch_dot(xl->data, yl->data, (size_t) xl->size * (size_t) xl->num_vectors, d_out, stream());
   double r = 0.0;
   dev_global_sums(x->comm, d_out, 1, &r);
   return r;
}
// x = y ./ diag(A), column by column of a multivector (par_csr_matop.c:6479-6658: the first entry of every row of the local
// block is its diagonal)
HYPRE_Int hypre_ParCSRDiagScaleVector(hypre_ParCSRMatrix *A, hypre_ParVector *par_y, hypre_ParVector *par_x)
{
   hypre_CSRMatrix *diag = A->diag;
   hypre_Vector *x = par_x->local_vector, *y = par_y->local_vector;
   if (x->num_vectors != y->num_vectors) { hypre_error_w_msg(HYPRE_ERROR_GENERIC, "Error! incompatible number of vectors!\n"); return hypre_error_flag; }
   if (diag->num_rows != x->size) { hypre_error_w_msg(HYPRE_ERROR_GENERIC, "Error! incompatible x size!\n"); return hypre_error_flag; }
   if (x->size > 0 && x->vecstride <= 0) { hypre_error_w_msg(HYPRE_ERROR_GENERIC, "Error! non-positive x vector stride!\n"); return hypre_error_flag; }
   if (y->size > 0 && y->vecstride <= 0) { hypre_error_w_msg(HYPRE_ERROR_GENERIC, "Error! non-positive y vector stride!\n"); return hypre_error_flag; }
   if (diag->num_rows != y->size) { hypre_error_w_msg(HYPRE_ERROR_GENERIC, "Error! incompatible y size!\n"); return hypre_error_flag; }
   HYPRE_AMD_REQUIRE_DEVICE(diag->memory_location, "hypre_ParCSRDiagScaleVector(A)");
   HYPRE_AMD_REQUIRE_DEVICE(x->memory_location, "hypre_ParCSRDiagScaleVector(x)");
   HYPRE_AMD_REQUIRE_DEVICE(y->memory_location, "hypre_ParCSRDiagScaleVector(y)");
   if (x->num_vectors > 1 && (x->idxstride != 1 || y->idxstride != 1))
   {
      hypre_error_w_msg(HYPRE_ERROR_GENERIC, "hypre_ParCSRDiagScaleVector: row-wise multivector storage is not supported");
      return hypre_error_flag;
   }
   launch_diag_first_scale(diag->i, diag->data, y->data, x->data, (size_t) diag->num_rows, x->num_vectors,
                           (size_t) y->vecstride, (size_t) x->vecstride, stream());
   par_x->all_zeros = 0;
   maybe_sync();
   return hypre_error_flag;
}
// parcsr_ls/HYPRE_parcsr_pcg.c: the diagonal-scaling preconditioner of the Krylov drivers (`ij -solver 2`)
HYPRE_Int HYPRE_ParCSRDiagScaleSetup(HYPRE_Solver solver, HYPRE_ParCSRMatrix A, HYPRE_ParVector y, HYPRE_ParVector x)
{
   (void) solver; (void) A; (void) y; (void) x;
   return hypre_error_flag;
}
HYPRE_Int HYPRE_ParCSRDiagScale(HYPRE_Solver solver, HYPRE_ParCSRMatrix HA, HYPRE_ParVector Hy, HYPRE_ParVector Hx)
{
   (void) solver;
   return hypre_ParCSRDiagScaleVector(HA, Hy, Hx);
}

HYPRE_Int hypre_ParVectorElmdivpy(hypre_ParVector *x, hypre_ParVector *b, hypre_ParVector *y)
{
   return hypre_SeqVectorElmdivpy(x->local_vector, b->local_vector, y->local_vector);
}
HYPRE_Int hypre_ParVectorElmdivpyMarked(hypre_ParVector *x, hypre_ParVector *b, hypre_ParVector *y,
                                        HYPRE_Int *marker, HYPRE_Int marker_val)
{
   return hypre_SeqVectorElmdivpyMarked(x->local_vector, b->local_vector, y->local_vector, marker, marker_val);
}

// ===========================================================================
// halo-exchange package
// ===========================================================================
// Build the neighbour lists from the (ascending) ghost-column map.  Because
// col_map_offd is sorted and ownership ranges are contiguous, the ghosts owned
// by one neighbour form one contiguous run (par_csr_communication.c:765-801).
HYPRE_Int hypre_ParCSRCommPkgCreate_core(MPI_Comm comm, HYPRE_BigInt *col_map_offd,
                                         HYPRE_BigInt first_col_diag, HYPRE_BigInt *col_starts,
                                         HYPRE_Int num_cols_diag, HYPRE_Int num_cols_offd,
                                         HYPRE_Int *p_num_recvs, HYPRE_Int **p_recv_procs,
                                         HYPRE_Int **p_recv_vec_starts, HYPRE_Int *p_num_sends,
                                         HYPRE_Int **p_send_procs, HYPRE_Int **p_send_map_starts,
                                         HYPRE_Int **p_send_map_elmts)
{
   (void) num_cols_diag;
   const hypre_amd_CommOps *o = comm_ops(comm);
   const int size = o ? o->size : 1;
   const int rank = o ? o->rank : 0;

   // every rank's first owned column (+ global end)
   std::vector<HYPRE_BigInt> starts((size_t) size + 1, 0);
   if (size > 1)
   {
      HYPRE_BigInt mine[2] = {col_starts[0], col_starts[1]};
      std::vector<HYPRE_BigInt> all((size_t) 2 * size);
      o->allgather(o->ctx, mine, all.data(), sizeof(mine));
      for (int r = 0; r < size; r++) { starts[(size_t) r] = all[(size_t) 2 * r]; }
      starts[(size_t) size] = all[(size_t) 2 * size - 1];
   }
   else
   {
      starts[0] = col_starts[0]; starts[1] = col_starts[1];
   }
   (void) first_col_diag;

   // receive side: runs of ghosts per owner
   std::vector<HYPRE_Int> recv_procs, recv_vec_starts;
   {
      HYPRE_Int k = 0;
      while (k < num_cols_offd)
      {
         const HYPRE_BigInt g = col_map_offd[k];
         const int owner = (int) (std::upper_bound(starts.begin(), starts.end(), g) - starts.begin()) - 1;
         recv_procs.push_back(owner);
         recv_vec_starts.push_back(k);
         const HYPRE_BigInt end = starts[(size_t) owner + 1];
         while (k < num_cols_offd && col_map_offd[k] < end) { k++; }
      }
      recv_vec_starts.push_back(num_cols_offd);
   }
   const HYPRE_Int num_recvs = (HYPRE_Int) recv_procs.size();

   // tell every owner how many of its entries we need: one row of a size x size table
   std::vector<HYPRE_Int> want((size_t) size, 0), table((size_t) size * size, 0);
   for (HYPRE_Int i = 0; i < num_recvs; i++)
   {
      want[(size_t) recv_procs[(size_t) i]] = recv_vec_starts[(size_t) i + 1] - recv_vec_starts[(size_t) i];
   }
   if (size > 1) { o->allgather(o->ctx, want.data(), table.data(), sizeof(HYPRE_Int) * (size_t) size); }

   std::vector<HYPRE_Int> send_procs, send_map_starts(1, 0);
   for (int r = 0; r < size; r++)
   {
      const HYPRE_Int cnt = (size > 1) ? table[(size_t) r * size + rank] : 0;
      if (cnt > 0)
      {
         send_procs.push_back(r);
         send_map_starts.push_back(send_map_starts.back() + cnt);
      }
   }
   const HYPRE_Int num_sends = (HYPRE_Int) send_procs.size();
   const HYPRE_Int tot_send = send_map_starts.back();

   // ship the wanted global ids to their owners; they become local gather ids
   std::vector<HYPRE_BigInt> req((size_t) std::max(tot_send, 1));
   if (size > 1 && (num_sends || num_recvs))
   {
      std::vector<void *> sb((size_t) num_recvs), rb((size_t) num_sends);
      std::vector<size_t> sbytes((size_t) num_recvs), rbytes((size_t) num_sends);
      for (HYPRE_Int i = 0; i < num_recvs; i++)
      {
         sb[(size_t) i] = (void *) (col_map_offd + recv_vec_starts[(size_t) i]);
         sbytes[(size_t) i] = sizeof(HYPRE_BigInt) * (size_t) (recv_vec_starts[(size_t) i + 1] - recv_vec_starts[(size_t) i]);
      }
      for (HYPRE_Int i = 0; i < num_sends; i++)
      {
         rb[(size_t) i] = (void *) (req.data() + send_map_starts[(size_t) i]);
         rbytes[(size_t) i] = sizeof(HYPRE_BigInt) * (size_t) (send_map_starts[(size_t) i + 1] - send_map_starts[(size_t) i]);
      }
      o->exchange(o->ctx, num_recvs, recv_procs.data(), sb.data(), sbytes.data(),
                  num_sends, send_procs.data(), rb.data(), rbytes.data(), 0, nullptr);
   }

   *p_num_recvs = num_recvs;
   *p_recv_procs = hypre_CTAlloc(HYPRE_Int, std::max(num_recvs, 1), HYPRE_MEMORY_HOST);
   *p_recv_vec_starts = hypre_CTAlloc(HYPRE_Int, num_recvs + 1, HYPRE_MEMORY_HOST);
   for (HYPRE_Int i = 0; i < num_recvs; i++) { (*p_recv_procs)[i] = recv_procs[(size_t) i]; }
   for (HYPRE_Int i = 0; i <= num_recvs; i++) { (*p_recv_vec_starts)[i] = recv_vec_starts[(size_t) i]; }
   *p_num_sends = num_sends;
   *p_send_procs = hypre_CTAlloc(HYPRE_Int, std::max(num_sends, 1), HYPRE_MEMORY_HOST);
   *p_send_map_starts = hypre_CTAlloc(HYPRE_Int, num_sends + 1, HYPRE_MEMORY_HOST);
   *p_send_map_elmts = hypre_CTAlloc(HYPRE_Int, std::max(tot_send, 1), HYPRE_MEMORY_HOST);
   for (HYPRE_Int i = 0; i < num_sends; i++) { (*p_send_procs)[i] = send_procs[(size_t) i]; }
   for (HYPRE_Int i = 0; i <= num_sends; i++) { (*p_send_map_starts)[i] = send_map_starts[(size_t) i]; }
   for (HYPRE_Int i = 0; i < tot_send; i++) { (*p_send_map_elmts)[i] = (HYPRE_Int) (req[(size_t) i] - col_starts[0]); }
   return hypre_error_flag;
}

HYPRE_Int hypre_MatvecCommPkgCreate(hypre_ParCSRMatrix *A)
{
   if (A->comm_pkg) { return hypre_error_flag; }
   hypre_ParCSRCommPkg *pkg = (hypre_ParCSRCommPkg *) calloc(1, sizeof(hypre_ParCSRCommPkg));
   pkg->comm = A->comm;
   pkg->num_components = 1;
   hypre_ParCSRCommPkgCreate_core(A->comm, A->col_map_offd, A->first_col_diag, A->col_starts,
                                  A->diag->num_cols, A->offd->num_cols,
                                  &pkg->num_recvs, &pkg->recv_procs, &pkg->recv_vec_starts,
                                  &pkg->num_sends, &pkg->send_procs, &pkg->send_map_starts,
                                  &pkg->send_map_elmts);
   A->comm_pkg = pkg;
   return hypre_error_flag;
}

HYPRE_Int hypre_MatvecCommPkgDestroy(hypre_ParCSRCommPkg *pkg)
{
   if (!pkg) { return hypre_error_flag; }
   hypre_Free(pkg->send_procs, HYPRE_MEMORY_HOST);
   hypre_Free(pkg->send_map_starts, HYPRE_MEMORY_HOST);
   hypre_Free(pkg->send_map_elmts, HYPRE_MEMORY_HOST);
   hypre_Free(pkg->recv_procs, HYPRE_MEMORY_HOST);
   hypre_Free(pkg->recv_vec_starts, HYPRE_MEMORY_HOST);
   hypre_Free(pkg->device_send_map_elmts, HYPRE_MEMORY_DEVICE);
   hypre_Free(pkg->tmp_data, HYPRE_MEMORY_DEVICE);
   hypre_Free(pkg->buf_data, HYPRE_MEMORY_DEVICE);
   if (pkg->matrix_E) { hypre_CSRMatrixDestroy(pkg->matrix_E); }
   free(pkg);
   return hypre_error_flag;
}

// Switch a package between single vectors and multivectors of num_components_in columns (par_csr_communication.c:1054-1154):
// entry i of a send list becomes num_components_in consecutive entries, component j addressing
// send_map_elmts[i] * idxstride + j * vecstride of the multivector's data, and every message grows by that factor —
// one exchange then carries all columns.  The device copies of the lists and the work buffers are dropped (rebuilt at
// the new size on the next product).
HYPRE_Int hypre_ParCSRCommPkgUpdateVecStarts(hypre_ParCSRCommPkg *pkg, HYPRE_Int num_components_in, HYPRE_Int vecstride,
                                             HYPRE_Int idxstride)
{
   const HYPRE_Int nc = pkg->num_components > 0 ? pkg->num_components : 1;
   if (num_components_in < 1) { hypre_error_in_arg(2); return hypre_error_flag; }
   if (num_components_in == nc) { return hypre_error_flag; }
   const HYPRE_Int ns = pkg->num_sends, nr = pkg->num_recvs;
   const HYPRE_Int tot = pkg->send_map_starts[ns];                 // in units of the current component count
   HYPRE_Int *elmts_new = hypre_CTAlloc(HYPRE_Int, (size_t) std::max(tot / nc * num_components_in, 1), HYPRE_MEMORY_HOST);
   const HYPRE_Int entries = tot / nc;
   for (HYPRE_Int i = 0; i < entries; i++)
   {
      if (num_components_in > nc)
      {
         // growing always starts from the single-vector list (as the reference: index * idxstride + j * vecstride)
         for (HYPRE_Int j = 0; j < num_components_in; j++)
         {
            elmts_new[(size_t) i * num_components_in + j] = pkg->send_map_elmts[(size_t) i * nc] * idxstride + j * vecstride;
         }
      }
      else
      {
         for (HYPRE_Int j = 0; j < num_components_in; j++)
         {
            elmts_new[(size_t) i * num_components_in + j] = pkg->send_map_elmts[(size_t) i * nc + j];
         }
      }
   }
   hypre_Free(pkg->send_map_elmts, HYPRE_MEMORY_HOST);
   pkg->send_map_elmts = elmts_new;
   hypre_Free(pkg->device_send_map_elmts, HYPRE_MEMORY_DEVICE);
   pkg->device_send_map_elmts = nullptr;
   hypre_Free(pkg->tmp_data, HYPRE_MEMORY_DEVICE); pkg->tmp_data = nullptr;
   hypre_Free(pkg->buf_data, HYPRE_MEMORY_DEVICE); pkg->buf_data = nullptr;
   // matrix_E is kept: it is defined in single-vector terms (ensure_matrix_E), whatever the component count
   for (HYPRE_Int i = 0; i <= ns; i++) { pkg->send_map_starts[i] = pkg->send_map_starts[i] / nc * num_components_in; }
   for (HYPRE_Int i = 0; i <= nr; i++) { pkg->recv_vec_starts[i] = pkg->recv_vec_starts[i] / nc * num_components_in; }
   pkg->num_components = num_components_in;
   return hypre_error_flag;
}

// Start a neighbour exchange.  job 1/11: owner -> ghost; job 2/12: ghost -> owner.
// Device buffers: the transfers are enqueued on the communication stream behind
// everything already queued on the compute stream; Destroy makes the compute
// stream wait for them (no host synchronisation).  Host buffers: blocking.
hypre_ParCSRCommHandle *hypre_ParCSRCommHandleCreate_v2(HYPRE_Int job, hypre_ParCSRCommPkg *pkg,
                                                        HYPRE_MemoryLocation send_loc, void *send_data,
                                                        HYPRE_MemoryLocation recv_loc, void *recv_data)
{
   hypre_ParCSRCommHandle *h = (hypre_ParCSRCommHandle *) calloc(1, sizeof(hypre_ParCSRCommHandle));
   h->comm_pkg = pkg;
   h->send_memory_location = send_loc;
   h->recv_memory_location = recv_loc;
   h->send_data = send_data;
   h->recv_data = recv_data;
   const hypre_amd_CommOps *o = comm_ops(pkg->comm);
   if (!o || o->size <= 1) { return h; }
   hamd::g_exchanges++;

   // jobs 1/2: HYPRE_Complex, 11/12: HYPRE_Int, 21/22: HYPRE_BigInt (par_csr_communication.c:483-614)
   const size_t esz = (job == 11 || job == 12) ? sizeof(HYPRE_Int) : sizeof(HYPRE_Complex);
   const bool forward = (job % 10 == 1);
   const HYPRE_Int ns = forward ? pkg->num_sends : pkg->num_recvs;
   const HYPRE_Int nr = forward ? pkg->num_recvs : pkg->num_sends;
   const HYPRE_Int *sprocs  = forward ? pkg->send_procs : pkg->recv_procs;
   const HYPRE_Int *rprocs  = forward ? pkg->recv_procs : pkg->send_procs;
   const HYPRE_Int *sstarts = forward ? pkg->send_map_starts : pkg->recv_vec_starts;
   const HYPRE_Int *rstarts = forward ? pkg->recv_vec_starts : pkg->send_map_starts;

   const bool dev = (send_loc == HYPRE_MEMORY_DEVICE) || (recv_loc == HYPRE_MEMORY_DEVICE);
   char *sbase = (char *) send_data, *rbase = (char *) recv_data;
   std::vector<char> hs, hr;
   const size_t stot = esz * (size_t) sstarts[ns], rtot = esz * (size_t) rstarts[nr];
   const bool stage = dev && !o->device_buffers;
   hamd::g_exchange_bytes += (long long) stot;
   if (stage)
   {
      // provider cannot take device pointers: bounce through host memory
      hs.resize(stot ? stot : 1); hr.resize(rtot ? rtot : 1);
      if (send_loc == HYPRE_MEMORY_DEVICE) { hypre_Memcpy(hs.data(), send_data, stot, HYPRE_MEMORY_HOST, HYPRE_MEMORY_DEVICE); sbase = hs.data(); }
      if (recv_loc == HYPRE_MEMORY_DEVICE) { rbase = hr.data(); }
   }
   std::vector<void *> sb((size_t) ns), rb((size_t) nr);
   std::vector<size_t> sbytes((size_t) ns), rbytes((size_t) nr);
   for (HYPRE_Int i = 0; i < ns; i++)
   {
      sb[(size_t) i] = sbase + esz * (size_t) sstarts[i];
      sbytes[(size_t) i] = esz * (size_t) (sstarts[i + 1] - sstarts[i]);
   }
   for (HYPRE_Int i = 0; i < nr; i++)
   {
      rb[(size_t) i] = rbase + esz * (size_t) rstarts[i];
      rbytes[(size_t) i] = esz * (size_t) (rstarts[i + 1] - rstarts[i]);
   }
   if (dev && !stage)
   {
      EventPair *ev = next_event_pair();
      Handle &hd = handle();
      HIP_CHECK(hipEventRecord(ev->pack, hd.compute_stream));
      HIP_CHECK(hipStreamWaitEvent(hd.comm_stream, ev->pack, 0));
      const double t_host = ev->ready ? omp_get_wtime() : 0.0;
      o->exchange(o->ctx, ns, sprocs, sb.data(), sbytes.data(), nr, rprocs, rb.data(), rbytes.data(), 1,
                  (void *) hd.comm_stream);
      if (ev->ready) { ev->host_us = 1e6 * (omp_get_wtime() - t_host); }
      HIP_CHECK(hipEventRecord(ev->halo, hd.comm_stream));
      h->num_requests = 1;   // marks "device exchange in flight"
      h->requests = ev;      // ... and the event pair that belongs to it
   }
   else
   {
      o->exchange(o->ctx, ns, sprocs, sb.data(), sbytes.data(), nr, rprocs, rb.data(), rbytes.data(), 0, nullptr);
      if (stage && recv_loc == HYPRE_MEMORY_DEVICE)
      {
         hypre_Memcpy(recv_data, hr.data(), rtot, HYPRE_MEMORY_DEVICE, HYPRE_MEMORY_HOST);
      }
   }
   return h;
}

hypre_ParCSRCommHandle *hypre_ParCSRCommHandleCreate(HYPRE_Int job, hypre_ParCSRCommPkg *pkg,
                                                     void *send_data, void *recv_data)
{
   return hypre_ParCSRCommHandleCreate_v2(job, pkg, HYPRE_MEMORY_HOST, send_data, HYPRE_MEMORY_HOST, recv_data);
}

HYPRE_Int hypre_ParCSRCommHandleDestroy(hypre_ParCSRCommHandle *h)
{
   if (!h) { return hypre_error_flag; }
   if (h->num_requests && h->requests)
   {
      EventPair *ev = (EventPair *) h->requests;
      if (ev->ready) { HIP_CHECK(hipEventRecord(ev->ready, handle().compute_stream)); }     // (diagnosis: the compute stream is here, the halo may not be)
      HIP_CHECK(hipStreamWaitEvent(handle().compute_stream, ev->halo, 0));
   }
   free(h);
   return hypre_error_flag;
}

// grow-only device scratch of the multivector products (two independent buffers)
static double *mv_scratch(size_t n, int which)
{
   static double *buf[2] = {nullptr, nullptr};
   static size_t len[2] = {0, 0};
   if (len[which] < n)
   {
      if (buf[which]) { hypre_Free(buf[which], HYPRE_MEMORY_DEVICE); }
      buf[which] = hypre_TAlloc(double, n, HYPRE_MEMORY_DEVICE);
      len[which] = n;
   }
   return buf[which];
}

// device work space of a package (allocated on first use, never in the hot loop)
static void ensure_pkg_device(hypre_ParCSRCommPkg *pkg, HYPRE_Int num_cols_offd)
{
   const HYPRE_Int tot_send = pkg->send_map_starts[pkg->num_sends];
   if (!pkg->device_send_map_elmts && tot_send)
   {
      pkg->device_send_map_elmts = hypre_TAlloc(HYPRE_Int, tot_send, HYPRE_MEMORY_DEVICE);
      hypre_TMemcpy(pkg->device_send_map_elmts, pkg->send_map_elmts, HYPRE_Int, tot_send,
                    HYPRE_MEMORY_DEVICE, HYPRE_MEMORY_HOST);
   }
   if (!pkg->buf_data && tot_send) { pkg->buf_data = hypre_TAlloc(HYPRE_Complex, tot_send, HYPRE_MEMORY_DEVICE); }
   const size_t ghosts = (size_t) num_cols_offd * (size_t) std::max(pkg->num_components, 1);
   if (!pkg->tmp_data && ghosts) { pkg->tmp_data = hypre_TAlloc(HYPRE_Complex, ghosts, HYPRE_MEMORY_DEVICE); }
}

// The unpack operator of the transpose product (the reference's matrix_E, par_csr_matvec_device.c:470-560): the
// num_rows x tot_send 0/1 matrix with a one at (send_map_elmts[i], i), kept as a device CSR matrix with its non-empty
// rows listed, so that  y += E * buf  adds every row's contributions in ONE fixed order (ascending position in the
// receive buffer) — an atomicAdd per received value would add them in whatever order the lanes happen to run.
static hypre_CSRMatrix *ensure_matrix_E(hypre_ParCSRCommPkg *pkg, HYPRE_Int num_rows)
{
   if (pkg->matrix_E && pkg->matrix_E->num_rows == num_rows) { return pkg->matrix_E; }
   if (pkg->matrix_E) { hypre_CSRMatrixDestroy(pkg->matrix_E); pkg->matrix_E = nullptr; }
   // always in single-vector terms: entry i of the lists is component 0 of slot i * num_components
   const HYPRE_Int nc = std::max(pkg->num_components, 1);
   const HYPRE_Int tot = pkg->send_map_starts[pkg->num_sends] / nc;
   hypre_CSRMatrix *E = hypre_CSRMatrixCreate(num_rows, tot, tot);
   hypre_CSRMatrixInitialize_v2(E, 0, HYPRE_MEMORY_HOST);
   for (HYPRE_Int i = 0; i < tot; i++) { E->i[pkg->send_map_elmts[(size_t) i * nc] + 1]++; }
   for (HYPRE_Int r = 0; r < num_rows; r++) { E->i[r + 1] += E->i[r]; }
   std::vector<HYPRE_Int> pos(E->i, E->i + num_rows);
   for (HYPRE_Int i = 0; i < tot; i++)
   {
      const HYPRE_Int q = pos[(size_t) pkg->send_map_elmts[(size_t) i * nc]]++;
      E->j[q] = i; E->data[q] = 1.0;
   }
   hypre_CSRMatrixSetRownnz(E);
   hypre_CSRMatrixMigrate(E, HYPRE_MEMORY_DEVICE);
   pkg->matrix_E = E;
   return E;
}

}  // extern "C"

namespace hamd {
// Start the owner -> ghost exchange of a device vector (collective over A's
// communicator).  Ghost values land in A->comm_pkg->tmp_data once dev_halo_end
// has made the compute stream wait for them.  Returns nullptr on one rank.
hypre_ParCSRCommHandle *dev_halo_begin(hypre_ParCSRMatrix *A, const double *x_local)
{
   HYPRE_Int nprocs;
   hypre_MPI_Comm_size(A->comm, &nprocs);
   if (nprocs <= 1) { return nullptr; }
   if (!A->comm_pkg) { hypre_MatvecCommPkgCreate(A); }
   hypre_ParCSRCommPkg *pkg = A->comm_pkg;
   if (!pkg->num_sends && !pkg->num_recvs) { return nullptr; }
   ensure_pkg_device(pkg, A->offd->num_cols);
   const HYPRE_Int tot_send = pkg->send_map_starts[pkg->num_sends];
   launch_gather(x_local, pkg->device_send_map_elmts, pkg->buf_data, (size_t) tot_send, stream());
   return hypre_ParCSRCommHandleCreate_v2(1, pkg, HYPRE_MEMORY_DEVICE, pkg->buf_data, HYPRE_MEMORY_DEVICE, pkg->tmp_data);
}
void dev_halo_end(hypre_ParCSRCommHandle *h)
{
   if (h) { hypre_ParCSRCommHandleDestroy(h); }
}
}  // namespace hamd

extern "C" {

// ===========================================================================
// ParCSR SpMV
// ===========================================================================
HYPRE_Int hypre_ParCSRMatrixMatvecOutOfPlaceDevice(HYPRE_Complex alpha, hypre_ParCSRMatrix *A,
                                                   hypre_ParVector *x, HYPRE_Complex beta,
                                                   hypre_ParVector *b, hypre_ParVector *y)
{
   hypre_CSRMatrix *diag = A->diag, *offd = A->offd;
   hypre_Vector *xl = x->local_vector, *bl = b->local_vector, *yl = y->local_vector;
   HYPRE_AMD_REQUIRE_DEVICE(diag->memory_location, "hypre_ParCSRMatrixMatvec(A)");
   HYPRE_AMD_REQUIRE_DEVICE(xl->memory_location, "hypre_ParCSRMatrixMatvec(x)");
   HYPRE_AMD_REQUIRE_DEVICE(yl->memory_location, "hypre_ParCSRMatrixMatvec(y)");
   HYPRE_AMD_REQUIRE_DEVICE(bl->memory_location, "hypre_ParCSRMatrixMatvec(b)");
   if (xl->num_vectors != 1 || yl->num_vectors != 1 || bl->num_vectors != 1)
   {
      // column-major multivectors (par_csr_matvec.c:146-165 asserts idxstride == 1 as well): one halo exchange for all
      // columns, the local block's product fused over the columns, the ghost block's column by column
      const HYPRE_Int nv = xl->num_vectors;
      if (yl->num_vectors != nv || bl->num_vectors != nv || xl->idxstride != 1 || yl->idxstride != 1 || bl->idxstride != 1)
      {
         hypre_error_w_msg(HYPRE_ERROR_GENERIC, "hypre_ParCSRMatrixMatvec: multivectors must agree in num_vectors and be stored column by column");
         return hypre_error_flag;
      }
      const int saved = handle().sync_compute;
      handle().sync_compute = 0;
      HYPRE_Int np;
      hypre_MPI_Comm_size(A->comm, &np);
      if (np > 1 && !A->comm_pkg) { hypre_MatvecCommPkgCreate(A); }
      hypre_ParCSRCommPkg *pkg = np > 1 ? A->comm_pkg : nullptr;
      const HYPRE_Int nco = offd->num_cols;
      hypre_ParCSRCommHandle *ch = nullptr;
      if (pkg && (pkg->num_sends || pkg->num_recvs))
      {
         // ONE halo exchange for all columns (par_csr_matvec.c:89-160): the package is switched to nv components, the
         // send buffer holds [entry][column], the ghost data arrives in the same shape and is put column by column
         hypre_ParCSRCommPkgUpdateVecStarts(pkg, nv, xl->vecstride, xl->idxstride);
         ensure_pkg_device(pkg, nco);
         const HYPRE_Int tot_send = pkg->send_map_starts[pkg->num_sends];
         launch_gather(xl->data, pkg->device_send_map_elmts, pkg->buf_data, (size_t) tot_send, stream());
         ch = hypre_ParCSRCommHandleCreate_v2(1, pkg, HYPRE_MEMORY_DEVICE, pkg->buf_data, HYPRE_MEMORY_DEVICE, pkg->tmp_data);
      }
      // the local block: all columns in one pass over the matrix where its plan allows it (seq_mv.cpp: spmv_device_columns)
      hypre_CSRMatrixMatvecDevice(0, alpha, diag, xl, beta, b == y ? yl : bl, yl, 0);
      if (ch)
      {
         hypre_ParCSRCommHandleDestroy(ch);
         if (nco > 0)
         {
            double *cols = mv_scratch((size_t) nco * (size_t) nv, 0);
            launch_deinterleave(pkg->tmp_data, cols, nco, nv, stream());
            for (HYPRE_Int v = 0; v < nv; v++)
            {
               hypre_Vector gv{}, yv = *yl;
               gv.data = cols + (size_t) v * nco; gv.size = nco; gv.num_vectors = 1; gv.vecstride = nco; gv.idxstride = 1;
               gv.memory_location = HYPRE_MEMORY_DEVICE;
               yv.data += (size_t) v * yl->vecstride; yv.num_vectors = 1;
               hypre_CSRMatrixMatvecDevice(0, alpha, offd, &gv, 1.0, &yv, &yv, 0);
            }
         }
         HIP_CHECK(hipStreamSynchronize(stream()));          // the buffers go away with the switch back
         hypre_ParCSRCommPkgUpdateVecStarts(pkg, 1, xl->vecstride, xl->idxstride);
      }
      handle().sync_compute = saved;
      maybe_sync();
      return hypre_error_flag;
   }
   const HYPRE_Int num_cols_offd = offd->num_cols;
   HYPRE_Int nprocs;
   hypre_MPI_Comm_size(A->comm, &nprocs);

   const int saved_sync = handle().sync_compute;
   handle().sync_compute = 0;

   // x halo: pack on the compute stream, grouped send/recv on the comm stream
   hypre_ParCSRCommHandle *ch = hamd::dev_halo_begin(A, xl->data);
   hypre_Vector x_ghost{};
   if (nprocs > 1 && num_cols_offd > 0)
   {
      x_ghost.data = A->comm_pkg->tmp_data;
      x_ghost.size = num_cols_offd;
      x_ghost.num_vectors = 1; x_ghost.vecstride = num_cols_offd; x_ghost.idxstride = 1;
      x_ghost.memory_location = HYPRE_MEMORY_DEVICE;
   }

   // interior product, overlapped with the exchange
   hypre_CSRMatrixMatvecDevice(0, alpha, diag, xl, beta, bl, yl, 0);

   hamd::dev_halo_end(ch);
   if (num_cols_offd > 0 && x_ghost.data)
   {
      hypre_CSRMatrixMatvecDevice(0, alpha, offd, &x_ghost, 1.0, yl, yl, 0);
   }
   handle().sync_compute = saved_sync;
   maybe_sync();
   return hypre_error_flag;
}

static HYPRE_Int par_ierr(HYPRE_BigInt nrows, HYPRE_BigInt ncols, hypre_ParVector *x, hypre_ParVector *b,
                          hypre_ParVector *y)
{
   // par_csr_matvec.c:57-82 — informational only
   HYPRE_Int ierr = 0;
   const bool badx = ncols != x->global_size;
   const bool bady = nrows != y->global_size || nrows != b->global_size;
   if (badx) { ierr = 11; }
   if (bady) { ierr = 12; }
   if (badx && bady) { ierr = 13; }
   return ierr;
}

HYPRE_Int hypre_ParCSRMatrixMatvecOutOfPlace(HYPRE_Complex alpha, hypre_ParCSRMatrix *A, hypre_ParVector *x,
                                             HYPRE_Complex beta, hypre_ParVector *b, hypre_ParVector *y)
{
   const HYPRE_Int ierr = par_ierr(A->global_num_rows, A->global_num_cols, x, b, y);
   hypre_ParCSRMatrixMatvecOutOfPlaceDevice(alpha, A, x, beta, b, y);
   return ierr;
}

HYPRE_Int hypre_ParCSRMatrixMatvec(HYPRE_Complex alpha, hypre_ParCSRMatrix *A, hypre_ParVector *x,
                                   HYPRE_Complex beta, hypre_ParVector *y)
{
   return hypre_ParCSRMatrixMatvecOutOfPlace(alpha, A, x, beta, y, y);
}

HYPRE_Int HYPRE_ParCSRMatrixMatvec(HYPRE_Complex alpha, HYPRE_ParCSRMatrix A, HYPRE_ParVector x,
                                   HYPRE_Complex beta, HYPRE_ParVector y)
{
   return hypre_ParCSRMatrixMatvec(alpha, A, x, beta, y);
}

// y = alpha*A^T*x + beta*y : ghost-row contributions travel back to their owners
HYPRE_Int hypre_ParCSRMatrixMatvecTDevice(HYPRE_Complex alpha, hypre_ParCSRMatrix *A, hypre_ParVector *x,
                                          HYPRE_Complex beta, hypre_ParVector *y)
{
   hypre_CSRMatrix *diag = A->diag, *offd = A->offd;
   hypre_Vector *xl = x->local_vector, *yl = y->local_vector;
   HYPRE_AMD_REQUIRE_DEVICE(diag->memory_location, "hypre_ParCSRMatrixMatvecT(A)");
   HYPRE_AMD_REQUIRE_DEVICE(xl->memory_location, "hypre_ParCSRMatrixMatvecT(x)");
   HYPRE_AMD_REQUIRE_DEVICE(yl->memory_location, "hypre_ParCSRMatrixMatvecT(y)");
   if (xl->num_vectors != 1 || yl->num_vectors != 1)
   {
      const HYPRE_Int nv = xl->num_vectors;
      if (yl->num_vectors != nv || xl->idxstride != 1 || yl->idxstride != 1)
      {
         hypre_error_w_msg(HYPRE_ERROR_GENERIC, "hypre_ParCSRMatrixMatvecT: multivectors must agree in num_vectors and be stored column by column");
         return hypre_error_flag;
      }
      const int saved = handle().sync_compute;
      handle().sync_compute = 0;
      HYPRE_Int np;
      hypre_MPI_Comm_size(A->comm, &np);
      if (np > 1 && !A->comm_pkg) { hypre_MatvecCommPkgCreate(A); }
      hypre_ParCSRCommPkg *pkg = np > 1 ? A->comm_pkg : nullptr;
      const HYPRE_Int nco = offd->num_cols;
      hypre_ParCSRCommHandle *ch = nullptr;
      hypre_CSRMatrix *E = nullptr;
      HYPRE_Int entries = 0;
      auto column = [](hypre_Vector *l, HYPRE_Int v) { hypre_Vector c = *l; c.data += (size_t) v * l->vecstride; c.num_vectors = 1; return c; };
      if (pkg && (pkg->num_sends || pkg->num_recvs))
      {
         // ghost-row contributions of all columns travel back in ONE reverse exchange
         entries = pkg->send_map_starts[pkg->num_sends] / std::max(pkg->num_components, 1);
         if (entries > 0) { E = ensure_matrix_E(pkg, diag->num_cols); }
         hypre_ParCSRCommPkgUpdateVecStarts(pkg, nv, yl->vecstride, yl->idxstride);
         ensure_pkg_device(pkg, nco);
         if (nco > 0)
         {
            double *cols = mv_scratch((size_t) nco * (size_t) nv, 0);
            for (HYPRE_Int v = 0; v < nv; v++)
            {
               hypre_Vector xv = column(xl, v), gv{};
               gv.data = cols + (size_t) v * nco; gv.size = nco; gv.num_vectors = 1; gv.vecstride = nco; gv.idxstride = 1;
               gv.memory_location = HYPRE_MEMORY_DEVICE;
               if (A->offdT) { hypre_CSRMatrixMatvecDevice(0, alpha, A->offdT, &xv, 0.0, &gv, &gv, 0); }
               else          { hypre_CSRMatrixMatvecDevice(1, alpha, offd, &xv, 0.0, &gv, &gv, 0); }
            }
            launch_interleave(cols, pkg->tmp_data, nco, nv, stream());
         }
         ch = hypre_ParCSRCommHandleCreate_v2(2, pkg, HYPRE_MEMORY_DEVICE, pkg->tmp_data, HYPRE_MEMORY_DEVICE, pkg->buf_data);
      }
      // the local block: all columns in one pass over the (stored or cached) transpose where its plan allows it
      if (A->diagT) { hypre_CSRMatrixMatvecDevice(0, alpha, A->diagT, xl, beta, yl, yl, 0); }
      else          { hypre_CSRMatrixMatvecDevice(1, alpha, diag, xl, beta, yl, yl, 0); }
      if (ch)
      {
         hypre_ParCSRCommHandleDestroy(ch);
         if (entries > 0)
         {
            double *recv = mv_scratch((size_t) entries * (size_t) nv, 1);
            launch_deinterleave(pkg->buf_data, recv, entries, nv, stream());
            for (HYPRE_Int v = 0; v < nv; v++)
            {
               hypre_Vector bv{}, yv = column(yl, v);
               bv.data = recv + (size_t) v * entries; bv.size = entries; bv.num_vectors = 1; bv.vecstride = entries; bv.idxstride = 1;
               bv.memory_location = HYPRE_MEMORY_DEVICE;
               hypre_CSRMatrixMatvecDevice(0, 1.0, E, &bv, 1.0, &yv, &yv, 0);
            }
         }
         HIP_CHECK(hipStreamSynchronize(stream()));
         hypre_ParCSRCommPkgUpdateVecStarts(pkg, 1, yl->vecstride, yl->idxstride);
      }
      handle().sync_compute = saved;
      maybe_sync();
      return hypre_error_flag;
   }
   const HYPRE_Int num_cols_offd = offd->num_cols;
   HYPRE_Int nprocs;
   hypre_MPI_Comm_size(A->comm, &nprocs);

   const int saved_sync = handle().sync_compute;
   handle().sync_compute = 0;

   hypre_ParCSRCommHandle *ch = nullptr;
   hypre_ParCSRCommPkg *pkg = nullptr;
   if (nprocs > 1)
   {
      if (!A->comm_pkg) { hypre_MatvecCommPkgCreate(A); }
      pkg = A->comm_pkg;
      ensure_pkg_device(pkg, num_cols_offd);
      if (num_cols_offd > 0)
      {
         hypre_Vector y_ghost{};
         y_ghost.data = pkg->tmp_data; y_ghost.size = num_cols_offd; y_ghost.num_vectors = 1;
         y_ghost.vecstride = num_cols_offd; y_ghost.idxstride = 1; y_ghost.memory_location = HYPRE_MEMORY_DEVICE;
         if (A->offdT) { hypre_CSRMatrixMatvecDevice(0, alpha, A->offdT, xl, 0.0, &y_ghost, &y_ghost, 0); }
         else          { hypre_CSRMatrixMatvecDevice(1, alpha, offd, xl, 0.0, &y_ghost, &y_ghost, 0); }
      }
      if (pkg->num_sends || pkg->num_recvs)
      {
         ch = hypre_ParCSRCommHandleCreate_v2(2, pkg, HYPRE_MEMORY_DEVICE, pkg->tmp_data,
                                              HYPRE_MEMORY_DEVICE, pkg->buf_data);
      }
   }
   if (A->diagT) { hypre_CSRMatrixMatvecDevice(0, alpha, A->diagT, xl, beta, yl, yl, 0); }
   else          { hypre_CSRMatrixMatvecDevice(1, alpha, diag, xl, beta, yl, yl, 0); }
   if (ch)
   {
      hypre_ParCSRCommHandleDestroy(ch);
      const HYPRE_Int tot_send = pkg->send_map_starts[pkg->num_sends];
      if (tot_send > 0)
      {
         // y += E * buf: every boundary row sums what its neighbours sent in a fixed order
         hypre_CSRMatrix *E = ensure_matrix_E(pkg, diag->num_cols);
         hypre_Vector bufv{};
         bufv.data = pkg->buf_data; bufv.size = tot_send; bufv.num_vectors = 1; bufv.vecstride = tot_send; bufv.idxstride = 1;
         bufv.memory_location = HYPRE_MEMORY_DEVICE;
         const bool mp = handle().fp32_values;
         handle().fp32_values = false;                  // E holds exact ones; nothing to gain from an fp32 copy
         hypre_CSRMatrixMatvecDevice(0, 1.0, E, &bufv, 1.0, yl, yl, 0);
         handle().fp32_values = mp;
      }
   }
   handle().sync_compute = saved_sync;
   maybe_sync();
   return hypre_error_flag;
}

HYPRE_Int hypre_ParCSRMatrixMatvecT(HYPRE_Complex alpha, hypre_ParCSRMatrix *A, hypre_ParVector *x,
                                    HYPRE_Complex beta, hypre_ParVector *y)
{
   HYPRE_Int ierr = 0;
   const bool badx = A->global_num_rows != x->global_size;
   const bool bady = A->global_num_cols != y->global_size;
   if (badx) { ierr = 1; }
   if (bady) { ierr = 2; }
   if (badx && bady) { ierr = 3; }
   hypre_ParCSRMatrixMatvecTDevice(alpha, A, x, beta, y);
   return ierr;
}

// ===========================================================================
// binding conveniences
// ===========================================================================
hypre_CSRMatrix *hypre_amd_CSRMatrixFromArrays(HYPRE_Int num_rows, HYPRE_Int num_cols, HYPRE_Int nnz,
                                               const HYPRE_Int *i, const HYPRE_Int *j,
                                               const HYPRE_Complex *data, HYPRE_MemoryLocation loc)
{
   hypre_CSRMatrix *m = hypre_CSRMatrixCreate(num_rows, num_cols, nnz);
   hypre_CSRMatrixInitialize_v2(m, 0, loc);
   hypre_TMemcpy(m->i, i, HYPRE_Int, num_rows + 1, loc, HYPRE_MEMORY_HOST);
   if (nnz)
   {
      hypre_TMemcpy(m->j, j, HYPRE_Int, nnz, loc, HYPRE_MEMORY_HOST);
      if (data) { hypre_TMemcpy(m->data, data, HYPRE_Complex, nnz, loc, HYPRE_MEMORY_HOST); }
   }
   return m;
}

hypre_Vector *hypre_amd_SeqVectorFromArray(HYPRE_Int size, const HYPRE_Complex *data, HYPRE_MemoryLocation loc)
{
   hypre_Vector *v = hypre_SeqVectorCreate(size);
   hypre_SeqVectorInitialize_v2(v, loc);
   if (data && size) { hypre_TMemcpy(v->data, data, HYPRE_Complex, size, loc, HYPRE_MEMORY_HOST); }
   return v;
}

HYPRE_Int hypre_amd_SeqVectorToArray(hypre_Vector *v, HYPRE_Complex *out)
{
   hypre_TMemcpy(out, v->data, HYPRE_Complex, (size_t) v->size * v->num_vectors, HYPRE_MEMORY_HOST, v->memory_location);
   return hypre_error_flag;
}

HYPRE_Int hypre_amd_CopyToHost(void *dst, const void *src, size_t bytes, HYPRE_MemoryLocation loc)
{
   hypre_Memcpy(dst, src, bytes, HYPRE_MEMORY_HOST, loc);
   return hypre_error_flag;
}

hypre_ParCSRMatrix *hypre_amd_ParCSRMatrixFromArrays(MPI_Comm comm, HYPRE_BigInt global_num_rows,
                                                     HYPRE_BigInt global_num_cols,
                                                     const HYPRE_BigInt *row_starts,
                                                     const HYPRE_BigInt *col_starts,
                                                     HYPRE_Int num_cols_offd, const HYPRE_BigInt *col_map_offd,
                                                     const HYPRE_Int *diag_i, const HYPRE_Int *diag_j,
                                                     const HYPRE_Complex *diag_data,
                                                     const HYPRE_Int *offd_i, const HYPRE_Int *offd_j,
                                                     const HYPRE_Complex *offd_data, HYPRE_MemoryLocation loc)
{
   HYPRE_BigInt rs[2] = {row_starts[0], row_starts[1]}, cs[2] = {col_starts[0], col_starts[1]};
   const HYPRE_Int nr = (HYPRE_Int) (rs[1] - rs[0]);
   const HYPRE_Int nnz_d = diag_i[nr];
   const HYPRE_Int nnz_o = offd_i ? offd_i[nr] : 0;
   hypre_ParCSRMatrix *A = hypre_ParCSRMatrixCreate(comm, global_num_rows, global_num_cols, rs, cs,
                                                    num_cols_offd, nnz_d, nnz_o);
   hypre_ParCSRMatrixInitialize_v2(A, loc);
   hypre_TMemcpy(A->diag->i, diag_i, HYPRE_Int, nr + 1, loc, HYPRE_MEMORY_HOST);
   if (nnz_d)
   {
      hypre_TMemcpy(A->diag->j, diag_j, HYPRE_Int, nnz_d, loc, HYPRE_MEMORY_HOST);
      hypre_TMemcpy(A->diag->data, diag_data, HYPRE_Complex, nnz_d, loc, HYPRE_MEMORY_HOST);
   }
   if (offd_i) { hypre_TMemcpy(A->offd->i, offd_i, HYPRE_Int, nr + 1, loc, HYPRE_MEMORY_HOST); }
   if (nnz_o)
   {
      hypre_TMemcpy(A->offd->j, offd_j, HYPRE_Int, nnz_o, loc, HYPRE_MEMORY_HOST);
      hypre_TMemcpy(A->offd->data, offd_data, HYPRE_Complex, nnz_o, loc, HYPRE_MEMORY_HOST);
   }
   for (HYPRE_Int k = 0; k < num_cols_offd; k++) { A->col_map_offd[k] = col_map_offd[k]; }
   hypre_CSRMatrixSetRownnz(A->offd);
   hypre_ParCSRMatrixSetNumNonzeros(A);
   hypre_ParCSRMatrixSetDNumNonzeros(A);
   return A;
}

hypre_ParVector *hypre_amd_ParVectorFromArray(MPI_Comm comm, HYPRE_BigInt global_size,
                                              const HYPRE_BigInt *partitioning, const HYPRE_Complex *data,
                                              HYPRE_MemoryLocation loc)
{
   HYPRE_BigInt part[2] = {partitioning[0], partitioning[1]};
   hypre_ParVector *v = hypre_ParVectorCreate(comm, global_size, part);
   hypre_ParVectorInitialize_v2(v, loc);
   const HYPRE_Int n = v->local_vector->size;
   if (data && n) { hypre_TMemcpy(v->local_vector->data, data, HYPRE_Complex, n, loc, HYPRE_MEMORY_HOST); }
   return v;
}

HYPRE_Int hypre_amd_ParVectorToArray(hypre_ParVector *v, HYPRE_Complex *out)
{
   return hypre_amd_SeqVectorToArray(v->local_vector, out);
}

}  // extern "C"
