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
// Setup (first sweep on a matrix, cached beside its SpMV plan): greedy first-fit colouring of the pattern of
// diag + diag^T on the host, then one CSR matrix per colour holding that colour's rows (original column numbering)
// and the list of their row numbers.  A sweep is one pass of the tiled SpMV kernel per colour with the Jacobi epilogue
// written IN PLACE through the row list: u[i] += w (f[i] - (A u)[i]) / a_ii for the rows i of the colour.  The whole
// matrix is streamed once per sweep, as in the fused Jacobi sweep; the price is one launch per colour and the
// strided vector traffic of the epilogue.
#include "amg_internal.hpp"
#include <algorithm>
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
   std::vector<hypre_CSRMatrix *> rows_of;     // [num_colors] device CSR: the rows of one colour
   std::vector<int *>             rowmap;      // [num_colors] device: original row number of every row of rows_of[c]
   std::vector<int>               count;       // rows per colour
   int    *d_color = nullptr;                  // [n] colour of every row (device; for the tests / oracle)
   double *d_diag = nullptr;                   // [n] first entry of every row (the diagonal), zero replaced by one
};

std::unordered_map<const hypre_CSRMatrix *, McPlan *> &mc_table()
{
   static std::unordered_map<const hypre_CSRMatrix *, McPlan *> t;
   return t;
}

void free_mc(McPlan *m)
{
   if (!m) { return; }
   for (hypre_CSRMatrix *c : m->rows_of) { if (c) { hypre_CSRMatrixDestroy(c); } }
   for (int *r : m->rowmap) { if (r) { hypre_Free(r, HYPRE_MEMORY_DEVICE); } }
   if (m->d_color) { hypre_Free(m->d_color, HYPRE_MEMORY_DEVICE); }
   if (m->d_diag) { hypre_Free(m->d_diag, HYPRE_MEMORY_DEVICE); }
   delete m;
}

// Greedy first-fit colouring in row order over the symmetrised pattern: colour[i] = smallest colour no neighbour
// j (a_ij != 0 or a_ji != 0 stored, j != i) carries yet.
void greedy_coloring(int n, const int *Ai, const int *Aj, std::vector<int> &color, int &num_colors)
{
   // transpose pattern (only needed where the pattern is not symmetric; building it is cheaper than testing)
   std::vector<int> ti((size_t) n + 1, 0);
   for (int i = 0; i < n; i++) { for (int k = Ai[i]; k < Ai[i + 1]; k++) { const int j = Aj[k]; if (j >= 0 && j < n && j != i) { ti[(size_t) j + 1]++; } } }
   for (int i = 0; i < n; i++) { ti[(size_t) i + 1] += ti[(size_t) i]; }
   std::vector<int> tj((size_t) std::max(ti[(size_t) n], 1)), pos(ti.begin(), ti.end() - 1);
   for (int i = 0; i < n; i++) { for (int k = Ai[i]; k < Ai[i + 1]; k++) { const int j = Aj[k]; if (j >= 0 && j < n && j != i) { tj[(size_t) pos[(size_t) j]++] = i; } } }
   color.assign((size_t) std::max(n, 1), -1);
   std::vector<int> mark;          // mark[c] == i: colour c is taken by a neighbour of row i
   num_colors = 0;
   for (int i = 0; i < n; i++)
   {
      for (int k = Ai[i]; k < Ai[i + 1]; k++)
      {
         const int j = Aj[k];
         if (j >= 0 && j < n && j != i && color[(size_t) j] >= 0) { mark[(size_t) color[(size_t) j]] = i; }
      }
      for (int k = ti[(size_t) i]; k < ti[(size_t) i + 1]; k++)
      {
         const int j = tj[(size_t) k];
         if (color[(size_t) j] >= 0) { mark[(size_t) color[(size_t) j]] = i; }
      }
      int c = 0;
      while (c < num_colors && mark[(size_t) c] == i) { c++; }
      if (c == num_colors) { num_colors++; mark.push_back(-1); }
      color[(size_t) i] = c;
   }
}

McPlan *get_mc(hypre_CSRMatrix *A)
{
   auto &t = mc_table();
   auto it = t.find(A);
   if (it != t.end())
   {
      McPlan *m = it->second;
      if (m->key_i == A->i && m->key_j == A->j && m->key_a == A->data && m->n == A->num_rows && m->nnz == A->num_nonzeros) { return m; }
      free_mc(m);
      t.erase(it);
   }
   McPlan *m = new McPlan();
   m->key_i = A->i; m->key_j = A->j; m->key_a = A->data; m->n = A->num_rows; m->nnz = A->num_nonzeros;
   const int n = A->num_rows, nnz = A->num_nonzeros;
   std::vector<int> hi((size_t) n + 1, 0), hj((size_t) std::max(nnz, 1));
   std::vector<double> ha((size_t) std::max(nnz, 1));
   HIP_CHECK(hipStreamSynchronize(stream()));
   hypre_TMemcpy(hi.data(), A->i, HYPRE_Int, (size_t) n + 1, HYPRE_MEMORY_HOST, A->memory_location);
   if (nnz > 0)
   {
      hypre_TMemcpy(hj.data(), A->j, HYPRE_Int, (size_t) nnz, HYPRE_MEMORY_HOST, A->memory_location);
      hypre_TMemcpy(ha.data(), A->data, HYPRE_Complex, (size_t) nnz, HYPRE_MEMORY_HOST, A->memory_location);
   }
   std::vector<int> color;
   greedy_coloring(n, hi.data(), hj.data(), color, m->num_colors);
   const int C = m->num_colors;
   m->count.assign((size_t) C, 0);
   std::vector<long long> cnnz((size_t) C, 0);
   for (int i = 0; i < n; i++) { m->count[(size_t) color[(size_t) i]]++; cnnz[(size_t) color[(size_t) i]] += hi[(size_t) i + 1] - hi[(size_t) i]; }
   m->rows_of.assign((size_t) C, nullptr);
   m->rowmap.assign((size_t) C, nullptr);
   std::vector<std::vector<int>> ci((size_t) C), cj((size_t) C), cr((size_t) C);
   std::vector<std::vector<double>> ca((size_t) C);
   for (int c = 0; c < C; c++)
   {
      ci[(size_t) c].reserve((size_t) m->count[(size_t) c] + 1); ci[(size_t) c].push_back(0);
      cj[(size_t) c].reserve((size_t) cnnz[(size_t) c]); ca[(size_t) c].reserve((size_t) cnnz[(size_t) c]);
      cr[(size_t) c].reserve((size_t) m->count[(size_t) c]);
   }
   std::vector<double> diag((size_t) std::max(n, 1), 1.0);
   for (int i = 0; i < n; i++)
   {
      const size_t c = (size_t) color[(size_t) i];
      for (int k = hi[(size_t) i]; k < hi[(size_t) i + 1]; k++) { cj[c].push_back(hj[(size_t) k]); ca[c].push_back(ha[(size_t) k]); }
      ci[c].push_back((int) cj[c].size());
      cr[c].push_back(i);
      if (hi[(size_t) i + 1] > hi[(size_t) i] && ha[(size_t) hi[(size_t) i]] != 0.0) { diag[(size_t) i] = ha[(size_t) hi[(size_t) i]]; }
   }
   for (int c = 0; c < C; c++)
   {
      const int nr = m->count[(size_t) c], nz = (int) cj[(size_t) c].size();
      hypre_CSRMatrix *M = hypre_CSRMatrixCreate(nr, A->num_cols, nz);
      hypre_CSRMatrixInitialize_v2(M, 0, HYPRE_MEMORY_DEVICE);
      hypre_TMemcpy(M->i, ci[(size_t) c].data(), HYPRE_Int, (size_t) nr + 1, HYPRE_MEMORY_DEVICE, HYPRE_MEMORY_HOST);
      if (nz > 0)
      {
         hypre_TMemcpy(M->j, cj[(size_t) c].data(), HYPRE_Int, (size_t) nz, HYPRE_MEMORY_DEVICE, HYPRE_MEMORY_HOST);
         hypre_TMemcpy(M->data, ca[(size_t) c].data(), HYPRE_Complex, (size_t) nz, HYPRE_MEMORY_DEVICE, HYPRE_MEMORY_HOST);
      }
      m->rows_of[(size_t) c] = M;
      m->rowmap[(size_t) c] = hypre_TAlloc(int, (size_t) std::max(nr, 1), HYPRE_MEMORY_DEVICE);
      hypre_TMemcpy(m->rowmap[(size_t) c], cr[(size_t) c].data(), int, (size_t) nr, HYPRE_MEMORY_DEVICE, HYPRE_MEMORY_HOST);
   }
   m->d_color = hypre_TAlloc(int, (size_t) std::max(n, 1), HYPRE_MEMORY_DEVICE);
   hypre_TMemcpy(m->d_color, color.data(), int, (size_t) n, HYPRE_MEMORY_DEVICE, HYPRE_MEMORY_HOST);
   m->d_diag = hypre_TAlloc(double, (size_t) std::max(n, 1), HYPRE_MEMORY_DEVICE);
   hypre_TMemcpy(m->d_diag, diag.data(), double, (size_t) n, HYPRE_MEMORY_DEVICE, HYPRE_MEMORY_HOST);
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
   for (int q = 0; q < C; q++)
   {
      const int c = direction > 0 ? q : C - 1 - q;
      hypre_CSRMatrix *M = m->rows_of[(size_t) c];
      if (M->num_rows <= 0) { continue; }
      SpmvPlan *plan = get_plan(M);
      SpmvArgs a{};
      a.Ai = M->i; a.Aj = M->j; a.Aa = M->data; a.Aa32 = nullptr;
      a.x = ud; a.b = ft; a.y = ud; a.aux = nullptr; a.d = d;
      a.marker = cf_marker; a.marker_val = relax_points;
      a.alpha = relax_weight; a.beta = 0.0; a.fill = HYPRE_SPMV_FILL_WHOLE; a.row_offset = 0;
      a.rowmap = m->rowmap[(size_t) c];
      spmv_default_flags(a);
      if (!(relax_points != 0 && cf_marker)) { a.marker = nullptr; }
      launch_spmv(plan, a, OP_JACOBI_MAP, s);
   }
   u->all_zeros = 0;
   handle().sync_compute = saved;
   maybe_sync();
   return hypre_error_flag;
}

}  // extern "C"
