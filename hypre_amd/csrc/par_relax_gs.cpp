// hypre_amd — hybrid Gauss-Seidel / SOR on the device (relax 3, 4, 6, 8, 13, 14, 88, 89).
//
// Reference: parcsr_ls/par_relax.c:691-945 (hypre_BoomerAMGRelaxHybridGaussSeidel_core,
// hypre_BoomerAMGRelaxHybridSOR) and the row bodies of par_relax.h:13-457.  The
// reference's own device entry (par_relax_device.c:19-90) solves the triangular
// system with rocSPARSE and ignores cf_marker / relax_points / omega; this one keeps
// the host semantics (CF point sets, SOR weights, thread-block "hybrid" structure)
// and reproduces the sequential sweep bit for bit by level scheduling
// (gs_kernels.hip).  The level sets of a matrix are computed once, on the host,
// from its sparsity pattern and cached beside the SpMV plan.
#include "amg_internal.hpp"
#include <algorithm>
#include <cstdlib>
#include <unordered_map>
#include <vector>

using namespace hamd;

namespace {

struct GsDirection
{
   std::vector<int> lev_start;      // [nlev + 1] offsets into sched
   int4            *d_sched = nullptr;   // rows in level order: {row, first entry, end of row, -}
   int             *d_lev_start = nullptr;
   int              nlev = 0;
};

struct GsSchedule
{
   const HYPRE_Int *key_i = nullptr, *key_j = nullptr;
   int              n = 0, nnz = 0, threads = 1;
   int              lanes = 8;      // lanes that share a row (power of two covering the average row)
   GsDirection      dir[2];         // 0 forward, 1 backward
   MatrixWatch      watch;          // the levels come from the pattern: every sweep checks it is still the same
};

std::unordered_map<const hypre_CSRMatrix *, GsSchedule *> &gs_table()
{
   static std::unordered_map<const hypre_CSRMatrix *, GsSchedule *> t;
   return t;
}

void free_schedule(GsSchedule *g)
{
   if (!g) { return; }
   for (int d = 0; d < 2; d++)
   {
      if (g->dir[d].d_sched) { (void) hipFree(g->dir[d].d_sched); }
      if (g->dir[d].d_lev_start) { (void) hipFree(g->dir[d].d_lev_start); }
   }
   watch_release(g->watch);
   delete g;
}

void partition1d(int n, int p, int j, int &s, int &e)
{
   // utilities/threading.c hypre_partition1D
   if (p <= 1) { s = 0; e = n; return; }
   const int size = n / p, rest = n - size * p;
   if (j < rest) { s = j * (size + 1); e = (j + 1) * (size + 1); }
   else { s = j * size + rest; e = (j + 1) * size + rest; }
}

// level of every row for one sweep direction, then rows bucketed by level
void build_direction(GsDirection &D, int n, int threads, const int *Ai, const int *Aj, bool forward)
{
   std::vector<int> level((size_t) std::max(n, 1), 0);
   int nlev = n > 0 ? 1 : 0;
   for (int t = 0; t < threads; t++)
   {
      int ns, ne;
      partition1d(n, threads, t, ns, ne);
      if (forward)
      {
         for (int i = ns; i < ne; i++)
         {
            int lev = 0;
            for (int jj = Ai[i]; jj < Ai[i + 1]; jj++)
            {
               const int j = Aj[jj];
               if (j >= ns && j < i) { lev = std::max(lev, level[(size_t) j] + 1); }
            }
            level[(size_t) i] = lev;
            nlev = std::max(nlev, lev + 1);
         }
      }
      else
      {
         for (int i = ne - 1; i >= ns; i--)
         {
            int lev = 0;
            for (int jj = Ai[i]; jj < Ai[i + 1]; jj++)
            {
               const int j = Aj[jj];
               if (j > i && j < ne) { lev = std::max(lev, level[(size_t) j] + 1); }
            }
            level[(size_t) i] = lev;
            nlev = std::max(nlev, lev + 1);
         }
      }
   }
   D.nlev = nlev;
   D.lev_start.assign((size_t) nlev + 1, 0);
   for (int i = 0; i < n; i++) { D.lev_start[(size_t) level[(size_t) i] + 1]++; }
   for (int l = 0; l < nlev; l++) { D.lev_start[(size_t) l + 1] += D.lev_start[(size_t) l]; }
   std::vector<int> pos(D.lev_start.begin(), D.lev_start.end() - (nlev > 0 ? 1 : 0));
   std::vector<int4> rows((size_t) std::max(n, 1));
   for (int i = 0; i < n; i++) { rows[(size_t) pos[(size_t) level[(size_t) i]]++] = make_int4(i, Ai[i], Ai[i + 1], 0); }
   HIP_CHECK(hipMalloc((void **) &D.d_sched, sizeof(int4) * (size_t) std::max(n, 1)));
   HIP_CHECK(hipMalloc((void **) &D.d_lev_start, sizeof(int) * ((size_t) nlev + 1)));
   HIP_CHECK(hipMemcpy(D.d_sched, rows.data(), sizeof(int4) * (size_t) n, hipMemcpyHostToDevice));
   HIP_CHECK(hipMemcpy(D.d_lev_start, D.lev_start.data(), sizeof(int) * ((size_t) nlev + 1), hipMemcpyHostToDevice));
}

GsSchedule *get_schedule(hypre_CSRMatrix *A, int threads)
{
   auto &t = gs_table();
   auto it = t.find(A);
   if (it != t.end())
   {
      GsSchedule *g = it->second;
      const bool flagged = watch_flagged(g->watch);
      if (!flagged && g->key_i == A->i && g->key_j == A->j && g->n == A->num_rows && g->nnz == A->num_nonzeros && g->threads == threads)
      {
         return g;
      }
      if (flagged)
      {
         hypre_error_w_msg(HYPRE_ERROR_GENERIC, "hybrid Gauss-Seidel: the level schedule was built from another pattern than the one at this "
                                                "address now: the sweeps since then are wrong; the schedule is rebuilt");
      }
      free_schedule(g);
      t.erase(it);
   }
   GsSchedule *g = new GsSchedule();
   g->key_i = A->i; g->key_j = A->j; g->n = A->num_rows; g->nnz = A->num_nonzeros; g->threads = threads;
   const int n = A->num_rows, nnz = A->num_nonzeros;
   std::vector<int> hi((size_t) n + 1, 0), hj((size_t) std::max(nnz, 1));
   HIP_CHECK(hipStreamSynchronize(stream()));
   hypre_TMemcpy(hi.data(), A->i, HYPRE_Int, (size_t) n + 1, HYPRE_MEMORY_HOST, A->memory_location);
   if (nnz > 0) { hypre_TMemcpy(hj.data(), A->j, HYPRE_Int, (size_t) nnz, HYPRE_MEMORY_HOST, A->memory_location); }
   const double avg = n > 0 ? (double) nnz / n : 1.0;
   g->lanes = avg <= 4.5 ? 4 : avg <= 9.0 ? 8 : avg <= 18.0 ? 16 : 32;
   build_direction(g->dir[0], n, threads, hi.data(), hj.data(), true);
   build_direction(g->dir[1], n, threads, hi.data(), hj.data(), false);
   if (n > 0) { watch_record(g->watch, A, false, stream()); }
   t[A] = g;
   return g;
}

// Levels that fit one pass of a workgroup (1024 / lanes rows) share one single-workgroup launch whose
// steps are pipelined (gs_kernels.hip: ~2 us per level); a larger level gets a grid of its own (~5 us).
// Measured on the 64^3 7-pt hierarchy: 1 pass 16.7 ms per V(1,1) cycle, 2: 17.0, 4: 18.8, 8: 36.5.
void run_direction(const GsDirection &D, int lanes, GsArgs a, hipStream_t s)
{
   static const int passes = [] { const char *e = getenv("HYPRE_AMD_GS_RUN_PASSES"); return std::max(1, e ? atoi(e) : 1); }();
   const int small = passes * (1024 / lanes);
   constexpr int MAX_RUN = 12288;                // level offsets of a run sit in LDS
   a.sched = D.d_sched;
   int lev = 0;
   while (lev < D.nlev)
   {
      const int cnt = D.lev_start[(size_t) lev + 1] - D.lev_start[(size_t) lev];
      if (cnt <= small)
      {
         int end = lev + 1;
         while (end < D.nlev && end - lev < MAX_RUN && D.lev_start[(size_t) end + 1] - D.lev_start[(size_t) end] <= small) { end++; }
         launch_gs_run(a, lanes, D.d_lev_start, lev, end, s);
         lev = end;
      }
      else
      {
         launch_gs_level(a, lanes, D.lev_start[(size_t) lev], cnt, s);
         lev++;
      }
   }
}

}  // namespace

namespace hamd {
void drop_gs_schedule(const hypre_CSRMatrix *A)
{
   auto &t = gs_table();
   auto it = t.find(A);
   if (it != t.end()) { free_schedule(it->second); t.erase(it); }
}
}  // namespace hamd

extern "C" HYPRE_Int hypre_BoomerAMGRelaxHybridGaussSeidelDevice(hypre_ParCSRMatrix *A, hypre_ParVector *f,
                                                                 HYPRE_Int *cf_marker, HYPRE_Int relax_points,
                                                                 HYPRE_Real relax_weight, HYPRE_Real omega,
                                                                 HYPRE_Real *l1_norms, hypre_ParVector *u,
                                                                 hypre_ParVector *Vtemp, hypre_ParVector *Ztemp,
                                                                 HYPRE_Int GS_order, HYPRE_Int Symm)
{
   HYPRE_AMD_REQUIRE_DEVICE(A->diag->memory_location, "hypre_BoomerAMGRelaxHybridGaussSeidelDevice(A)");
   HYPRE_AMD_REQUIRE_DEVICE(u->local_vector->memory_location, "hypre_BoomerAMGRelaxHybridGaussSeidelDevice(u)");
   hypre_CSRMatrix *diag = A->diag, *offd = A->offd;
   const int n = diag->num_rows;
   if (n <= 0) { return hypre_error_flag; }
   if (!Vtemp || !Ztemp || Vtemp->local_vector->size < n || Ztemp->local_vector->size < n)
   {
      hypre_error_w_msg(HYPRE_ERROR_GENERIC, "hypre_BoomerAMGRelaxHybridGaussSeidelDevice: Vtemp and Ztemp must hold a local vector each");
      return hypre_error_flag;
   }
   if (relax_points != 0 && !cf_marker)
   {
      hypre_error_w_msg(HYPRE_ERROR_GENERIC, "hypre_BoomerAMGRelaxHybridGaussSeidelDevice: relax_points without a CF marker");
      return hypre_error_flag;
   }
   hipStream_t s = stream();
   const int saved = handle().sync_compute;
   handle().sync_compute = 0;
   double *ud = u->local_vector->data;
   double *vt = Vtemp->local_vector->data, *zt = Ztemp->local_vector->data;
   HYPRE_Int nprocs;
   hypre_MPI_Comm_size(A->comm, &nprocs);

   // ghost values of the incoming iterate (par_relax.c:735-754), overlapped with the schedule lookup
   // an iterate known to be zero has zero ghosts: no exchange, no ghost terms (same bits: res - a*0 = res)
   const bool zero_guess = u->all_zeros != 0;
   hypre_ParCSRCommHandle *ch = (nprocs > 1 && !zero_guess) ? dev_halo_begin(A, ud) : nullptr;
   GsSchedule *g = get_schedule(diag, std::max(1, std::min(handle().gs_threads, n)));
   if (!is_owned(diag)) { watch_check(g->watch, diag, false, s); }
   dev_halo_end(ch);

   GsArgs a{};
   a.Di = diag->i; a.Dj = diag->j; a.Da = diag->data;
   const bool has_offd = nprocs > 1 && !zero_guess && offd && offd->num_cols > 0 && offd->num_nonzeros > 0;
   a.Oi = has_offd ? offd->i : nullptr; a.Oj = has_offd ? offd->j : nullptr; a.Oa = has_offd ? offd->data : nullptr;
   a.vext = has_offd ? A->comm_pkg->tmp_data : nullptr;
   a.f = f->local_vector->data;
   a.cf = cf_marker; a.relax_points = relax_points;
   a.l1 = l1_norms;
   a.u = ud;
   a.w = relax_weight; a.omega = omega;
   a.non_scale = (relax_weight == 1.0 && omega == 1.0) ? 1 : 0;
   // par_relax.c:1256-1330: plain GS/SOR skips the diagonal entry; the l1 variants keep it in
   // the row sum unless weights are in play (relax 8/13/14 call the core with Skip_diag = !non_scale)
   a.skip_diag = l1_norms ? (a.non_scale ? 0 : 1) : 1;
   a.n = n; a.threads = g->threads;

   // Vtemp: state at the start of the call (off-block and SOR terms read it)
   launch_copy(vt, ud, (size_t) n, s);
   a.vtemp = vt;
   const int nsweeps = Symm ? 2 : 1;
   for (int sweep = 0; sweep < nsweeps; sweep++)
   {
      const int dirn = Symm ? (sweep == 0 ? 1 : -1) : (GS_order > 0 ? 1 : -1);
      if (sweep == 0) { a.uold = vt; }
      else { launch_copy(zt, ud, (size_t) n, s); a.uold = zt; }
      a.dir = dirn;
      run_direction(g->dir[dirn > 0 ? 0 : 1], g->lanes, a, s);
   }
   u->all_zeros = 0;               // the sweep has written u (relax 89 calls this twice: the second sweep must fetch ghosts)
   handle().sync_compute = saved;
   maybe_sync();
   return hypre_error_flag;
}
