// hypre_amd — multicolour Gauss-Seidel on the device (relax 21 forward, 22 backward colour order).
//
// No counterpart in the reference (parcsr_ls/par_relax*.c has no colouring; BASELINE's north_star names it).  What the
// reference does have is the hybrid Gauss-Seidel family (par_relax.c:691-945): Jacobi across ranks, a sequential
// Gauss-Seidel sweep inside a rank — inherently serial, and on the GPU a chain of ~thousands of dependency levels
// (par_relax_gs.cpp).  Re-ordering the local unknowns by colour makes the same sweep parallel: rows of one colour do not
// couple to each other, so all of them can be relaxed at once from the current iterate, and colour after colour that
// IS the sequential Gauss-Seidel sweep of the colour-permuted system.  Parity statement: relax 21 equals
// the oracle's hybrid forward sweep (relax 3) applied to P A P^T with P the colour permutation, relax 22 the backward
// one (relax 4); tests/test_multicolor_gpu.py.
//
// Setup (inside HYPRE_BoomerAMGSetup for the levels of a hierarchy smoothed with relax 21 / 22; on the first sweep for a
// matrix handed to the public entry point; cached beside the SpMV plan), all of it on the device (mc_kernels.hip): greedy
// first-fit colouring in row order of the pattern of diag + diag^T, then
//   * large levels: one CSR matrix per colour holding that colour's rows (original column numbering) and the list of
//     their row numbers; a sweep is one pass of the tiled SpMV kernel per colour with the Jacobi epilogue written IN PLACE
//     through the row list: u[i] += w (f[i] - (A u)[i]) / a_ii for the rows i of the colour.  The whole matrix is
//     streamed once per sweep, as in the fused Jacobi sweep; the price is one launch per colour and the strided vector
//     traffic of the epilogue;
//   * small levels (at most 1.5 M entries): the rows listed colour by colour, and ONE kernel of one workgroup that sweeps
//     all colours with a workgroup barrier between them, on the level's own matrix: the coarse levels have 15 - 40
//     colours of a few hundred rows each, and a launch per colour is 5 us of latency around nothing (346 launches per
//     cycle at 256^3 before, a third of the cycle).
#include "amg_internal.hpp"
#include <algorithm>
#include <omp.h>
#include <unordered_map>
#include <vector>

using namespace hamd;

namespace {

struct McPlan
{
   const HYPRE_Int *key_i = nullptr, *key_j = nullptr;
   const HYPRE_Complex *key_a = nullptr;
   int n = 0, nnz = 0;
   int num_colors = 0;
   bool small = false;                         // swept by the one-workgroup kernel: no per-colour matrices
   int  tail_from = -1;                        // large levels: the colours from here on hold a few dozen rows each (first-fit leaves
                                               // many such) and are swept together by the one-workgroup kernel; -1: none
   std::vector<hypre_CSRMatrix *> rows_of;     // [num_colors] device CSR views: the rows of one colour (large levels)
   std::vector<int>               count;       // rows per colour
   std::vector<int>               cstart;      // [num_colors + 1] where every colour starts in d_order
   int    *d_cstart = nullptr;                 // the same on the device
   int    *d_order = nullptr;                  // [n] the rows colour by colour, ascending inside a colour
   void   *d_rowinfo = nullptr;                // [n] (row, first entry, last entry, 0) for every position of d_order (look-ahead sweep)
   int    *d_ptr = nullptr, *d_cj = nullptr;   // large levels: the colour-sorted copy of the matrix the views point into
   double *d_ca = nullptr;
   int    *d_color = nullptr;                  // [n] colour of every row (device; for the tests / oracle)
   double *d_diag = nullptr;                   // [n] first entry of every row (the diagonal), zero replaced by one
   MatrixWatch watch;                          // the classes hold COPIES of the matrix: every sweep checks it is still the same
   const int *rowmap(int c) const { return d_order + cstart[(size_t) c]; }
};

// the one-workgroup sweeps fetch what does not depend on the iterate one pass ahead (mc_small_sweep_ahead_kernel; default on;
// HYPRE_AMD_MC_LOOK_AHEAD=0 / hypre_amd_SetMcLookAhead: the plain kernel, for comparisons)
int &mc_look_ahead()
{
   static int on = [] { const char *e = getenv("HYPRE_AMD_MC_LOOK_AHEAD"); return e ? atoi(e) : 1; }();
   return on;
}

std::unordered_map<const hypre_CSRMatrix *, McPlan *> &mc_table()
{
   static std::unordered_map<const hypre_CSRMatrix *, McPlan *> t;
   return t;
}

void free_mc(McPlan *m)
{
   if (!m) { return; }
   for (hypre_CSRMatrix *c : m->rows_of) { if (c) { hypre_CSRMatrixDestroy(c); } }        // views: drops their plans only
   if (m->d_cstart) { HIP_CHECK(hipFree(m->d_cstart)); }
   if (m->d_order) { HIP_CHECK(hipFree(m->d_order)); }
   if (m->d_rowinfo) { HIP_CHECK(hipFree(m->d_rowinfo)); }
   if (m->d_ptr) { HIP_CHECK(hipFree(m->d_ptr)); }
   if (m->d_cj) { HIP_CHECK(hipFree(m->d_cj)); }
   if (m->d_ca) { HIP_CHECK(hipFree(m->d_ca)); }
   if (m->d_color) { hypre_Free(m->d_color, HYPRE_MEMORY_DEVICE); }
   if (m->d_diag) { hypre_Free(m->d_diag, HYPRE_MEMORY_DEVICE); }
   watch_release(m->watch);
   delete m;
}

constexpr int MC_SMALL_NNZ = 20000;            // levels of at most this many entries are swept by one workgroup (a level of
                                               // 77 000 entries and 25 colours: 270 us per sweep that way, 125 us as 25 launches)

McPlan *get_mc(hypre_CSRMatrix *A)
{
   auto &t = mc_table();
   auto it = t.find(A);
   if (it != t.end())
   {
      McPlan *m = it->second;
      const bool flagged = watch_flagged(m->watch);
      if (!flagged && m->key_i == A->i && m->key_j == A->j && m->key_a == A->data && m->n == A->num_rows && m->nnz == A->num_nonzeros) { return m; }
      if (flagged)
      {
         hypre_error_w_msg(HYPRE_ERROR_GENERIC, "multicolour Gauss-Seidel: the colour classes were built from another matrix than the one at "
                                                "this address now (or its values changed without hypre_amd_CSRMatrixInvalidatePlan): the "
                                                "sweeps since then are wrong; the classes are rebuilt");
      }
      free_mc(m);
      t.erase(it);
      bump_plan_generation();
   }
   McPlan *m = new McPlan();
   m->key_i = A->i; m->key_j = A->j; m->key_a = A->data; m->n = A->num_rows; m->nnz = A->num_nonzeros;
   const int n = A->num_rows;
   hipStream_t s = stream();
   const bool timing = getenv("HYPRE_AMD_SETUP_TIMING") != nullptr;
   const double t0 = omp_get_wtime();
   int rounds = 0;
   m->d_color = hypre_TAlloc(int, (size_t) std::max(n, 1), HYPRE_MEMORY_DEVICE);
   m->d_diag = hypre_TAlloc(double, (size_t) std::max(n, 1), HYPRE_MEMORY_DEVICE);
   if (n > 0)
   {
      // pattern of A^T (device transpose), then the colouring.  Row order for as long as its dependency chains are those
      // of a grid (a 256^3 grid: 766 rounds); past 1500 rounds the hashed order takes over (irregular coarse levels, whose
      // chains run into the thousands: their colour counts are the same either way)
      hypre_CSRMatrix *AT = nullptr;
      hypre_CSRMatrixTranspose(A, &AT, 0);
      static const int budget = [] { const char *e = getenv("HYPRE_AMD_MC_ROUND_BUDGET"); return e ? atoi(e) : 1500; }();
      m->num_colors = device_greedy_coloring(n, A->i, A->j, AT->i, AT->j, m->d_color, budget, s, &rounds);
      hypre_CSRMatrixDestroy(AT);
      launch_mc_diag(n, A->i, A->data, m->d_diag, s);
   }
   const int C = m->num_colors;
   const double t1 = omp_get_wtime();
   device_color_order(n, C, m->d_color, &m->d_order, m->cstart, s);
   m->count.assign((size_t) C, 0);
   for (int c = 0; c < C; c++) { m->count[(size_t) c] = m->cstart[(size_t) c + 1] - m->cstart[(size_t) c]; }
   HIP_CHECK(hipMalloc((void **) &m->d_cstart, sizeof(int) * ((size_t) C + 1)));
   HIP_CHECK(hipMemcpyAsync(m->d_cstart, m->cstart.data(), sizeof(int) * ((size_t) C + 1), hipMemcpyHostToDevice, s));
   HIP_CHECK(hipStreamSynchronize(s));
   if (n > 0 && hipMalloc(&m->d_rowinfo, sizeof(int) * 4 * (size_t) n) == hipSuccess) { launch_mc_rowinfo(n, m->d_order, A->i, m->d_rowinfo, s); }
   else { (void) hipGetLastError(); m->d_rowinfo = nullptr; }
   m->small = A->num_nonzeros <= MC_SMALL_NNZ;
   if (!m->small)
   {
      // the tail of small colours: every colour from tail_from on fits one pass of the one-workgroup kernel (128 rows), and
      // there are at least three of them (a colour there costs a chain of loads and a barrier, about 3.5 us, against the
      // 5 - 6 us of a launch of its own)
      int c0 = C;
      while (c0 > 0 && m->count[(size_t) c0 - 1] <= 128) { c0--; }
      m->tail_from = (C - c0 >= 3) ? c0 : -1;
      std::vector<int> slice0, slice_nnz;
      device_color_matrices(n, C, m->d_color, m->d_order, m->cstart, A->i, A->j, A->data, &m->d_ptr, &m->d_cj, &m->d_ca, slice0, slice_nnz, s);
      m->rows_of.assign((size_t) C, nullptr);
      for (int c = 0; c < C; c++)
      {
         hypre_CSRMatrix *M = hypre_CSRMatrixCreate(m->count[(size_t) c], A->num_cols, slice_nnz[(size_t) c]);
         M->i = m->d_ptr + m->cstart[(size_t) c] + c;
         M->j = m->d_cj + slice0[(size_t) c];
         M->data = m->d_ca + slice0[(size_t) c];
         M->memory_location = HYPRE_MEMORY_DEVICE;
         M->owns_data = 0;
         // (a view into the library's own colour-sorted copy; not marked as owned: rows of one colour are not neighbours, they
         // share no columns, and a block of them cannot stage its x — measured, also with the columns renumbered colour by
         // colour: 7.4 ms per cycle at 256^3 against 6.9 ms — so no row slices are attempted for them)
         m->rows_of[(size_t) c] = M;
         if (M->num_rows > 0 && M->num_nonzeros > 0) { (void) get_plan(M); }    // the colour's SpMV plan: part of the setup
      }
   }
   if (timing)
   {
      fprintf(stderr, "   multicolour plan: %d rows, %d colours, colouring %.3fs (%d rounds), classes%s %.3fs", n, C, t1 - t0, rounds,
              m->small ? "" : " + matrices + plans", omp_get_wtime() - t1);
      if (m->small) { fprintf(stderr, "; one workgroup sweeps all colours\n"); }
      else if (m->tail_from >= 0) { fprintf(stderr, "; colours %d .. %d (%d rows) swept by one workgroup\n", m->tail_from, C - 1, m->cstart[(size_t) C] - m->cstart[(size_t) m->tail_from]); }
      else { fprintf(stderr, "\n"); }
   }
   if (n > 0) { watch_record(m->watch, A, true, s); }
   t[A] = m;
   return m;
}

double *mc_scratch(size_t n)
{
   static double *buf = nullptr;
   static size_t len = 0;
   if (len < n)
   {
      if (buf) { hypre_Free(buf, HYPRE_MEMORY_DEVICE); }
      buf = hypre_TAlloc(double, n, HYPRE_MEMORY_DEVICE);
      len = n;
   }
   return buf;
}

}  // namespace

namespace hamd {
// colouring and colour classes of a device matrix now rather than at its first sweep (the setup calls this for the levels
// a hierarchy smooths with relax 21 / 22)
void prepare_mc_plan(hypre_CSRMatrix *A)
{
   if (A && A->memory_location == HYPRE_MEMORY_DEVICE && A->num_rows > 0) { (void) get_mc(A); }
}
void drop_mc_plan(const hypre_CSRMatrix *A)
{
   auto &t = mc_table();
   auto it = t.find(A);
   if (it != t.end()) { free_mc(it->second); t.erase(it); bump_plan_generation(); }
}
}  // namespace hamd

extern "C" {

// The colouring the sweeps of A's diagonal block use (built on demand): number of colours; colors_out (host, one
// entry per local row) is filled when given.
HYPRE_Int hypre_amd_ParCSRMatrixMultiColoring(hypre_ParCSRMatrix *A, HYPRE_Int *colors_out)
{
   HYPRE_AMD_REQUIRE_DEVICE(A->diag->memory_location, "hypre_amd_ParCSRMatrixMultiColoring(A)");
   McPlan *m = get_mc(A->diag);
   if (colors_out && m->n > 0) { hypre_TMemcpy(colors_out, m->d_color, int, (size_t) m->n, HYPRE_MEMORY_HOST, HYPRE_MEMORY_DEVICE); }
   return m->num_colors;
}

// One multicolour Gauss-Seidel sweep, colours in ascending (direction > 0) or descending order:
//    for every colour c, for all rows i of c at once:  u_i += w (f_i - sum_j a_ij u_j - sum_g o_ig u_g^old) / a_ii
// (u_g^old: ghost values of the iterate the call started from, as in the hybrid sweeps of par_relax.c:735-754).
// cf_marker / relax_points restrict the rows as in hypre_BoomerAMGRelax.  diag: smoother diagonal (the cycle hands
// the l1_norms option-5 vector = a_ii with 0 -> 1), or NULL for the stored diagonal entries.
// The one-workgroup multicolour sweeps with (1, default) or without (0) the look-ahead; < 0: unchanged.  Returns the setting.
// Same results either way (a switch for comparisons and for the recorded coarse tail's signature).
HYPRE_Int hypre_amd_SetMcLookAhead(HYPRE_Int on)
{
   if (on >= 0) { mc_look_ahead() = on != 0; }
   return mc_look_ahead();
}

HYPRE_Int hypre_BoomerAMGRelaxMultiColorGaussSeidelDevice(hypre_ParCSRMatrix *A, hypre_ParVector *f, HYPRE_Int *cf_marker,
                                                          HYPRE_Int relax_points, HYPRE_Real relax_weight, HYPRE_Real *diag,
                                                          hypre_ParVector *u, hypre_ParVector *Vtemp, HYPRE_Int direction)
{
   HYPRE_AMD_REQUIRE_DEVICE(A->diag->memory_location, "hypre_BoomerAMGRelaxMultiColorGaussSeidelDevice(A)");
   HYPRE_AMD_REQUIRE_DEVICE(u->local_vector->memory_location, "hypre_BoomerAMGRelaxMultiColorGaussSeidelDevice(u)");
   hypre_CSRMatrix *dg = A->diag, *offd = A->offd;
   const int n = dg->num_rows;
   if (n <= 0) { return hypre_error_flag; }
   if (relax_points != 0 && !cf_marker)
   {
      hypre_error_w_msg(HYPRE_ERROR_GENERIC, "hypre_BoomerAMGRelaxMultiColorGaussSeidelDevice: relax_points without a CF marker");
      return hypre_error_flag;
   }
   hipStream_t s = stream();
   const int saved = handle().sync_compute;
   handle().sync_compute = 0;
   double *ud = u->local_vector->data;
   const double *fd = f->local_vector->data;
   HYPRE_Int nprocs;
   hypre_MPI_Comm_size(A->comm, &nprocs);
   const bool zero_guess = u->all_zeros != 0;
   hypre_ParCSRCommHandle *ch = (nprocs > 1 && !zero_guess) ? dev_halo_begin(A, ud) : nullptr;
   McPlan *m = get_mc(dg);
   // (a mismatch is reported by the next call that asks for the classes; a matrix the library made itself cannot change
   // behind its classes: no launch spent on it)
   if (!is_owned(dg)) { watch_check(m->watch, dg, true, s); }
   dev_halo_end(ch);
   // right-hand side with the ghost couplings folded in: ft = f - A_offd u_ghost
   const double *ft = fd;
   if (nprocs > 1 && !zero_guess && offd && offd->num_cols > 0 && offd->num_nonzeros > 0)
   {
      double *tmp = (Vtemp && Vtemp->local_vector->size >= n) ? Vtemp->local_vector->data : mc_scratch((size_t) n);
      launch_copy(tmp, fd, (size_t) n, s);
      SpmvArgs o{};
      o.Ai = offd->i; o.Aj = offd->j; o.Aa = offd->data; o.Aa32 = nullptr;
      o.x = A->comm_pkg->tmp_data; o.y = tmp; o.d = nullptr; o.marker = nullptr; o.alpha = -1.0;
      if (offd->rownnz) { launch_spmv_rownnz(offd->rownnz, offd->num_rownnz, o, s); }
      else { launch_spmv_allrows_update(offd->num_rows, o, s); }
      ft = tmp;
   }
   const double *d = diag ? diag : m->d_diag;
   const int C = m->num_colors;
   if (m->small)
   {
      const bool masked = relax_points != 0 && cf_marker;
      if (m->d_rowinfo && mc_look_ahead())
      {
         launch_mc_small_sweep_ahead(C, direction, m->d_cstart, m->d_rowinfo, n, dg->j, dg->data, handle().fp32_values ? fp32_values_of(dg) : nullptr,
                                     dg->num_nonzeros, ft, d, masked ? cf_marker : nullptr, relax_points, relax_weight, ud, n, dg->num_nonzeros, 0, s);
      }
      else
      {
      launch_mc_small_sweep(C, direction, m->d_cstart, m->d_order, dg->i, dg->j, dg->data, handle().fp32_values ? fp32_values_of(dg) : nullptr,
                            ft, d, masked ? cf_marker : nullptr, relax_points, relax_weight, ud, n, dg->num_nonzeros, s);
      }
   }
   else
   {
   // colours [0, big) by a launch each, colours [big, C) — the small ones — by the one-workgroup kernel, in sweep order
   const int big = m->tail_from >= 0 ? m->tail_from : C;
   auto sweep_tail = [&]()
   {
      if (big >= C) { return; }
      const bool masked = relax_points != 0 && cf_marker;
      const int trows = m->cstart[(size_t) C] - m->cstart[(size_t) big];
      const int tnnz = (int) ((long long) trows * dg->num_nonzeros / std::max(n, 1));
      if (m->d_rowinfo && mc_look_ahead())
      {
         launch_mc_small_sweep_ahead(C - big, direction, m->d_cstart + big, m->d_rowinfo, n, dg->j, dg->data,
                                     handle().fp32_values ? fp32_values_of(dg) : nullptr, dg->num_nonzeros, ft, d, masked ? cf_marker : nullptr,
                                     relax_points, relax_weight, ud, trows, tnnz, m->cstart[(size_t) big], s);
         return;
      }
      launch_mc_small_sweep(C - big, direction, m->d_cstart + big, m->d_order, dg->i, dg->j, dg->data,
                            handle().fp32_values ? fp32_values_of(dg) : nullptr, ft, d, masked ? cf_marker : nullptr, relax_points,
                            relax_weight, ud, trows, tnnz, s);
   };
   if (direction <= 0) { sweep_tail(); }
   for (int q = 0; q < big; q++)
   {
      const int c = direction > 0 ? q : big - 1 - q;
      hypre_CSRMatrix *M = m->rows_of[(size_t) c];
      if (M->num_rows <= 0) { continue; }
      SpmvPlan *plan = get_plan(M);
      SpmvArgs a{};
      a.Ai = M->i; a.Aj = M->j; a.Aa = M->data; a.Aa32 = nullptr;
      a.x = ud; a.b = ft; a.y = ud; a.aux = nullptr; a.d = d;
      a.marker = cf_marker; a.marker_val = relax_points;
      a.alpha = relax_weight; a.beta = 0.0; a.fill = HYPRE_SPMV_FILL_WHOLE; a.row_offset = 0;
      a.rowmap = m->rowmap(c);
      spmv_default_flags(a);
      if (!(relax_points != 0 && cf_marker)) { a.marker = nullptr; }
      launch_spmv(plan, a, OP_JACOBI_MAP, s);
   }
   if (direction > 0) { sweep_tail(); }
   }
   u->all_zeros = 0;
   handle().sync_compute = saved;
   maybe_sync();
   return hypre_error_flag;
}

}  // extern "C"
