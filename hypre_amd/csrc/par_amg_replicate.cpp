// hypre_amd — replicated tail of a distributed hierarchy.
//
// On the coarse levels of a multi-GPU V-cycle every product and sweep is a few
// microseconds of work behind a halo exchange that costs tens of microseconds:
// four exchanges per level, pure latency.  Once a level is small enough its
// operators are cheap to hold on every rank, so setup gathers the levels from
// there down into an ordinary single-rank hierarchy (same operators, same
// smoother diagonals, same coarse factors) that every rank owns.  In the cycle
// the right-hand side of the first replicated level is summed into place with
// ONE all-reduce, the rest of the V-cycle runs locally and redundantly through
// the same kernels, and each rank keeps its slice of the correction.
//
// The reference has a relative in hypre_seqAMGSetup / hypre_seqAMGCycle
// (parcsr_ls/par_amg_setup.c:3160, par_coordinates... "seq_threshold"): it
// gathers the coarse OPERATOR and builds a new sequential hierarchy under it.
// Here the distributed hierarchy itself is gathered, so the arithmetic is that
// of the distributed cycle (Jacobi-type smoothers do not depend on how rows are
// spread over ranks; the two-stage Gauss-Seidel sweeps keep the triangle of every
// rank's own diagonal block); only the association of row sums changes (a row is
// summed in one piece instead of diag block + ghost block).  In mixed precision
// every matrix value of the cycle — diagonal and ghost blocks alike — is the
// fp32-rounded one, so the gathered rows carry the same values as the distributed ones.
#include "amg_internal.hpp"
#include <algorithm>
#include <vector>

using namespace hamd;

namespace {

// every rank contributes `n` items of T; result: all items in rank order
template <class T>
std::vector<T> allgather_var(const hypre_amd_CommOps *o, const T *mine, int n, std::vector<int> *counts_out = nullptr)
{
   std::vector<int> counts((size_t) o->size, 0);
   o->allgather(o->ctx, &n, counts.data(), sizeof(int));
   const int maxn = std::max(1, *std::max_element(counts.begin(), counts.end()));
   std::vector<T> pad((size_t) maxn), all((size_t) maxn * (size_t) o->size);
   for (int k = 0; k < n; k++) { pad[(size_t) k] = mine[k]; }
   o->allgather(o->ctx, pad.data(), all.data(), sizeof(T) * (size_t) maxn);
   std::vector<T> out;
   size_t total = 0;
   for (int c : counts) { total += (size_t) c; }
   out.reserve(total);
   for (int r = 0; r < o->size; r++)
   {
      out.insert(out.end(), all.begin() + (size_t) r * (size_t) maxn, all.begin() + (size_t) r * (size_t) maxn + (size_t) counts[(size_t) r]);
   }
   if (counts_out) { *counts_out = counts; }
   return out;
}

// distributed host matrix -> the same matrix as a single-rank (communicator 0) host ParCSR on every
// rank.  A row keeps its entry order: diag block first (so the diagonal stays in front), ghost block after.
hypre_ParCSRMatrix *replicate_matrix(hypre_ParCSRMatrix *M)
{
   const hypre_amd_CommOps *o = comm_ops(M->comm);
   // a level the setup worked on in device memory: host copies for the gather
   hypre_CSRMatrix *D = M->diag, *O = M->offd, *Dh = nullptr, *Oh = nullptr;
   if (D->memory_location == HYPRE_MEMORY_DEVICE) { Dh = hypre_CSRMatrixClone_v2(D, 1, HYPRE_MEMORY_HOST); D = Dh; }
   if (O->memory_location == HYPRE_MEMORY_DEVICE) { Oh = hypre_CSRMatrixClone_v2(O, 1, HYPRE_MEMORY_HOST); O = Oh; }
   const HYPRE_Int nloc = D->num_rows;
   std::vector<int> rowlen((size_t) std::max(nloc, 1));
   std::vector<int> cols;
   std::vector<double> vals;
   cols.reserve((size_t) D->num_nonzeros + (size_t) O->num_nonzeros);
   vals.reserve(cols.capacity());
   for (HYPRE_Int i = 0; i < nloc; i++)
   {
      for (HYPRE_Int k = D->i[i]; k < D->i[i + 1]; k++) { cols.push_back((int) (M->first_col_diag + D->j[k])); vals.push_back(D->data[k]); }
      if (O->num_cols > 0)
      {
         for (HYPRE_Int k = O->i[i]; k < O->i[i + 1]; k++) { cols.push_back((int) M->col_map_offd[O->j[k]]); vals.push_back(O->data[k]); }
      }
      rowlen[(size_t) i] = (D->i[i + 1] - D->i[i]) + (O->num_cols > 0 ? O->i[i + 1] - O->i[i] : 0);
   }
   if (Dh) { hypre_CSRMatrixDestroy(Dh); }
   if (Oh) { hypre_CSRMatrixDestroy(Oh); }
   std::vector<int> all_len = allgather_var<int>(o, rowlen.data(), nloc);
   std::vector<int> all_col = allgather_var<int>(o, cols.data(), (int) cols.size());
   std::vector<double> all_val = allgather_var<double>(o, vals.data(), (int) vals.size());
   const HYPRE_Int n = (HYPRE_Int) all_len.size(), nnz = (HYPRE_Int) all_col.size();
   HYPRE_BigInt rs[2] = {0, (HYPRE_BigInt) n}, cs[2] = {0, M->global_num_cols};
   hypre_ParCSRMatrix *R = hypre_ParCSRMatrixCreate(0, (HYPRE_BigInt) n, M->global_num_cols, rs, cs, 0, nnz, 0);
   hypre_ParCSRMatrixInitialize_v2(R, HYPRE_MEMORY_HOST);
   R->diag->i[0] = 0;
   for (HYPRE_Int i = 0; i < n; i++) { R->diag->i[i + 1] = R->diag->i[i] + all_len[(size_t) i]; }
   if (nnz > 0)
   {
      memcpy(R->diag->j, all_col.data(), sizeof(int) * (size_t) nnz);
      memcpy(R->diag->data, all_val.data(), sizeof(double) * (size_t) nnz);
   }
   hypre_CSRMatrixSetRownnz(R->offd);
   hypre_ParCSRMatrixSetNumNonzeros(R);
   hypre_ParCSRMatrixSetDNumNonzeros(R);
   return R;
}

// smoothers whose result does not depend on how the rows are spread over ranks — or, for the two-stage
// Gauss-Seidel sweeps 11 / 12 (par_relax.c:1506-1588: residual with the whole operator, inner steps with the strict
// lower triangle of the rank's OWN diagonal block), depends on it only through that triangle, which the tail keeps
// rank block by rank block (block_strict_lower below)
bool jacobi_like(int t) { return t == 0 || t == 7 || t == 18 || t == 16 || t == 11 || t == 12; }
bool ge_like(int t) { return t == 9 || t == 19 || t == 98 || t == 99 || t == 198 || t == 199; }

// {a_ij : j < i, j owned by the rank that owns i} of a replicated host matrix, as a device CSR matrix
hypre_CSRMatrix *block_strict_lower(hypre_CSRMatrix *R, const std::vector<int> &first_row /* [ranks + 1] */)
{
   const HYPRE_Int n = R->num_rows;
   std::vector<HYPRE_Int> li((size_t) n + 1, 0), lj;
   std::vector<HYPRE_Complex> la;
   size_t b = 0;
   for (HYPRE_Int i = 0; i < n; i++)
   {
      while (b + 1 < first_row.size() && i >= first_row[b + 1]) { b++; }
      const HYPRE_Int lo = first_row[b];
      for (HYPRE_Int k = R->i[i]; k < R->i[i + 1]; k++)
      {
         const HYPRE_Int c = R->j[k];
         if (c >= lo && c < i) { lj.push_back(c); la.push_back(R->data[k]); }
      }
      li[(size_t) i + 1] = (HYPRE_Int) lj.size();
   }
   hypre_CSRMatrix *L = hypre_CSRMatrixCreate(n, R->num_cols, (HYPRE_Int) lj.size());
   hypre_CSRMatrixInitialize_v2(L, 0, HYPRE_MEMORY_DEVICE);
   hypre_TMemcpy(L->i, li.data(), HYPRE_Int, (size_t) n + 1, HYPRE_MEMORY_DEVICE, HYPRE_MEMORY_HOST);
   if (!lj.empty())
   {
      hypre_TMemcpy(L->j, lj.data(), HYPRE_Int, lj.size(), HYPRE_MEMORY_DEVICE, HYPRE_MEMORY_HOST);
      hypre_TMemcpy(L->data, la.data(), HYPRE_Complex, la.size(), HYPRE_MEMORY_DEVICE, HYPRE_MEMORY_HOST);
   }
   return L;
}

hypre_ParVector *self_vec(HYPRE_BigInt n, HYPRE_MemoryLocation loc)
{
   HYPRE_BigInt part[2] = {0, n};
   hypre_ParVector *v = hypre_ParVectorCreate(0, n, part);
   hypre_ParVectorInitialize_v2(v, loc);
   return v;
}

}  // namespace

namespace hamd {

// Called at the end of the host setup, before the hierarchy moves to the device.  hostA[l] are the host
// operators of all levels.  Leaves pv->tail == nullptr when the configuration does not qualify.
void build_replicated_tail(hypre_ParAMGData *d, const std::vector<hypre_ParCSRMatrix *> &hostA)
{
   AmgPrivate *pv = (AmgPrivate *) d->amd_private;
   const hypre_amd_CommOps *o = comm_ops(hostA[0]->comm);
   const int L = d->num_levels;
   if (!o || o->size <= 1 || pv->replicate_rows <= 0 || L < 2) { return; }
   // what the replicated cycle reproduces exactly: V-cycles with smoothers whose result does not depend on
   // the row distribution, and a coarsest level that is either eliminated or smoothed the same way
   if (d->cycle_type != 1 || d->fcycle || d->grid_relax_points) { return; }
   const HYPRE_Int *gt = d->grid_relax_type;
   if (!jacobi_like(gt[1]) || !jacobi_like(gt[2]) || !(ge_like(gt[3]) || jacobi_like(gt[3]))) { return; }
   int Lr = -1;
   for (int l = 1; l < L; l++) { if (hostA[(size_t) l]->global_num_rows <= (HYPRE_BigInt) pv->replicate_rows) { Lr = l; break; } }
   if (Lr < 0) { return; }

   HYPRE_Solver ts = nullptr;
   HYPRE_BoomerAMGCreate(&ts);
   hypre_ParAMGData *t = (hypre_ParAMGData *) ts;
   AmgPrivate *tp = (AmgPrivate *) t->amd_private;
   tp->replica = true;
   tp->replicate_rows = 0;
   tp->emulated_threads = pv->emulated_threads;
   tp->mixed_precision = pv->mixed_precision;
   const int TL = L - Lr;
   const bool two_stage = gt[1] == 11 || gt[1] == 12 || gt[2] == 11 || gt[2] == 12 || gt[3] == 11 || gt[3] == 12;
   std::vector<hypre_CSRMatrix *> tail_lower((size_t) TL, nullptr);
   t->memory_location = d->memory_location;
   t->max_levels = std::max(TL, 1);
   t->num_levels = TL;
   t->cycle_type = 1; t->fcycle = 0; t->relax_order = d->relax_order;
   t->user_relax_type = d->user_relax_type; t->user_coarse_relax_type = d->user_coarse_relax_type;
   t->max_coarse_size = d->max_coarse_size; t->min_coarse_size = d->min_coarse_size;
   for (int k = 0; k < 4; k++) { t->grid_relax_type[k] = d->grid_relax_type[k]; t->num_grid_sweeps[k] = d->num_grid_sweeps[k]; }
   // a one-level tail is just the coarsest level: the cycle uses slot 0 of the tables for it
   if (TL == 1) { t->grid_relax_type[0] = d->grid_relax_type[3]; t->num_grid_sweeps[0] = d->num_grid_sweeps[3]; t->user_relax_type = d->grid_relax_type[3]; }
   t->A_array = (hypre_ParCSRMatrix **) calloc((size_t) TL, sizeof(void *));
   t->P_array = (hypre_ParCSRMatrix **) calloc((size_t) TL, sizeof(void *));
   t->R_array = t->P_array;
   t->F_array = (hypre_ParVector **) calloc((size_t) TL, sizeof(void *));
   t->U_array = (hypre_ParVector **) calloc((size_t) TL, sizeof(void *));
   t->CF_marker_array = (hypre_IntArray **) calloc((size_t) TL, sizeof(void *));
   t->l1_norms = (hypre_Vector **) calloc((size_t) TL, sizeof(void *));
   t->relax_weight = (HYPRE_Real *) calloc((size_t) TL, sizeof(HYPRE_Real));
   t->omega = (HYPRE_Real *) calloc((size_t) TL, sizeof(HYPRE_Real));
   for (int l = 0; l < TL; l++)
   {
      const int g = Lr + l;
      t->relax_weight[l] = d->relax_weight[g];
      t->omega[l] = d->omega[g];
      t->A_array[l] = replicate_matrix(hostA[(size_t) g]);
      if (g < L - 1) { t->P_array[l] = replicate_matrix(d->P_array[g]); }
      if (two_stage && d->memory_location == HYPRE_MEMORY_DEVICE)
      {
         int mine = (int) hostA[(size_t) g]->row_starts[0];
         std::vector<int> first((size_t) o->size + 1, 0);
         o->allgather(o->ctx, &mine, first.data(), sizeof(int));
         first[(size_t) o->size] = (int) hostA[(size_t) g]->global_num_rows;
         tail_lower[(size_t) l] = block_strict_lower(t->A_array[l]->diag, first);
      }
      const HYPRE_BigInt n = t->A_array[l]->global_num_rows;
      t->F_array[l] = self_vec(n, HYPRE_MEMORY_HOST);
      t->U_array[l] = self_vec(n, HYPRE_MEMORY_HOST);
      if (d->l1_norms[g])
      {
         std::vector<double> mine((size_t) std::max(d->l1_norms[g]->size, 1));
         hypre_Memcpy(mine.data(), d->l1_norms[g]->data, sizeof(double) * (size_t) d->l1_norms[g]->size, HYPRE_MEMORY_HOST,
                      d->l1_norms[g]->memory_location);
         std::vector<double> all = allgather_var<double>(o, mine.data(), d->l1_norms[g]->size);
         t->l1_norms[l] = hypre_SeqVectorCreate((HYPRE_Int) all.size());
         hypre_SeqVectorInitialize_v2(t->l1_norms[l], HYPRE_MEMORY_HOST);
         memcpy(t->l1_norms[l]->data, all.data(), sizeof(double) * all.size());
      }
      if (d->CF_marker_array[g])
      {
         std::vector<int> all = allgather_var<int>(o, d->CF_marker_array[g]->data, d->CF_marker_array[g]->size);
         t->CF_marker_array[l] = hypre_IntArrayCreate((HYPRE_Int) all.size());
         hypre_IntArrayInitialize_v2(t->CF_marker_array[l], HYPRE_MEMORY_HOST);
         memcpy(t->CF_marker_array[l]->data, all.data(), sizeof(int) * all.size());
      }
   }
   t->A = t->A_array[0];
   t->Vtemp = self_vec(t->A_array[0]->global_num_rows, HYPRE_MEMORY_HOST);
   t->Ztemp = self_vec(t->A_array[0]->global_num_rows, HYPRE_MEMORY_HOST);
   if (d->cheby_coefs)
   {
      // Chebyshev: coefficients are the same on every rank, the scaling vector is gathered
      t->cheby_order = d->cheby_order; t->cheby_scale = d->cheby_scale; t->cheby_variant = d->cheby_variant;
      t->cheby_eig_est = d->cheby_eig_est; t->cheby_fraction = d->cheby_fraction;
      t->cheby_coefs = (HYPRE_Real **) calloc((size_t) TL, sizeof(void *));
      t->cheby_ds = (hypre_Vector **) calloc((size_t) TL, sizeof(void *));
      t->max_eig_est = (HYPRE_Real *) calloc((size_t) TL, sizeof(HYPRE_Real));
      t->min_eig_est = (HYPRE_Real *) calloc((size_t) TL, sizeof(HYPRE_Real));
      const int nco = std::min(std::max((int) d->cheby_order, 1), 4) + 1;
      for (int l = 0; l < TL; l++)
      {
         const int g = Lr + l;
         t->max_eig_est[l] = d->max_eig_est[g]; t->min_eig_est[l] = d->min_eig_est[g];
         if (d->cheby_coefs[g])
         {
            t->cheby_coefs[l] = hypre_CTAlloc(HYPRE_Real, (size_t) nco, HYPRE_MEMORY_HOST);
            memcpy(t->cheby_coefs[l], d->cheby_coefs[g], sizeof(HYPRE_Real) * (size_t) nco);
         }
         if (d->cheby_ds && d->cheby_ds[g])
         {
            std::vector<double> all = allgather_var<double>(o, d->cheby_ds[g]->data, d->cheby_ds[g]->size);
            t->cheby_ds[l] = hypre_SeqVectorCreate((HYPRE_Int) all.size());
            hypre_SeqVectorInitialize_v2(t->cheby_ds[l], HYPRE_MEMORY_HOST);
            memcpy(t->cheby_ds[l]->data, all.data(), sizeof(double) * all.size());
         }
      }
      t->Ptemp = self_vec(t->A_array[0]->global_num_rows, HYPRE_MEMORY_HOST);
      t->Rtemp = self_vec(t->A_array[0]->global_num_rows, HYPRE_MEMORY_HOST);
   }
   // the dense coarse operator was gathered by hypre_GaussElimSetup already: same matrix on every rank
   if (d->A_mat && d->gs_setup)
   {
      const size_t n = (size_t) hostA[(size_t) L - 1]->global_num_rows;
      t->A_mat = (HYPRE_Real *) malloc(sizeof(HYPRE_Real) * n * n);
      memcpy(t->A_mat, d->A_mat, sizeof(HYPRE_Real) * n * n);
      t->b_vec = (HYPRE_Real *) calloc(n, sizeof(HYPRE_Real));
      t->gs_setup = 1;
   }

   // to the device, like the main hierarchy
   if (d->memory_location == HYPRE_MEMORY_DEVICE)
   {
      for (int l = 0; l < TL; l++)
      {
         hypre_ParCSRMatrixMigrate(t->A_array[l], HYPRE_MEMORY_DEVICE);
         // the tail is the library's own copy of these levels: its matrices cannot change behind their plans
         auto own = [](hypre_ParCSRMatrix *M) { for (hypre_CSRMatrix *B : {M->diag, M->offd, M->diagT, M->offdT}) { if (B) { mark_owned(B); } } };
         own(t->A_array[l]);
         if (tail_lower[(size_t) l]) { mark_owned(tail_lower[(size_t) l]); set_strict_lower(t->A_array[l]->diag, tail_lower[(size_t) l]); }
         if (t->P_array[l])
         {
            hypre_amd_ParCSRMatrixKeepTranspose(t->P_array[l]);
            hypre_ParCSRMatrixMigrate(t->P_array[l], HYPRE_MEMORY_DEVICE);
            own(t->P_array[l]);
         }
         hypre_ParVectorMigrate(t->F_array[l], HYPRE_MEMORY_DEVICE);
         hypre_ParVectorMigrate(t->U_array[l], HYPRE_MEMORY_DEVICE);
         if (t->l1_norms[l]) { hypre_SeqVectorMigrate(t->l1_norms[l], HYPRE_MEMORY_DEVICE); }
         if (t->CF_marker_array[l])
         {
            hypre_IntArray *a = t->CF_marker_array[l];
            HYPRE_Int *dd = hypre_TAlloc(HYPRE_Int, (size_t) std::max(a->size, 1), HYPRE_MEMORY_DEVICE);
            hypre_TMemcpy(dd, a->data, HYPRE_Int, (size_t) a->size, HYPRE_MEMORY_DEVICE, HYPRE_MEMORY_HOST);
            hypre_Free(a->data, HYPRE_MEMORY_HOST);
            a->data = dd; a->memory_location = HYPRE_MEMORY_DEVICE;
         }
      }
      hypre_ParVectorMigrate(t->Vtemp, HYPRE_MEMORY_DEVICE);
      hypre_ParVectorMigrate(t->Ztemp, HYPRE_MEMORY_DEVICE);
      if (t->Ptemp) { hypre_ParVectorMigrate(t->Ptemp, HYPRE_MEMORY_DEVICE); }
      if (t->Rtemp) { hypre_ParVectorMigrate(t->Rtemp, HYPRE_MEMORY_DEVICE); }
      if (t->cheby_ds) { for (int l = 0; l < TL; l++) { if (t->cheby_ds[l]) { hypre_SeqVectorMigrate(t->cheby_ds[l], HYPRE_MEMORY_DEVICE); } } }
      pv->d_tail_f = hypre_TAlloc(double, (size_t) std::max<HYPRE_BigInt>(t->A_array[0]->global_num_rows, 1), HYPRE_MEMORY_DEVICE);
   }
   else
   {
      for (int l = 0; l < TL - 1; l++) { hypre_amd_ParCSRMatrixKeepTranspose(t->P_array[l]); }
   }
   pv->tail = t;
   pv->tail_level = Lr;
}

void destroy_replicated_tail(hypre_ParAMGData *d)
{
   AmgPrivate *pv = (AmgPrivate *) d->amd_private;
   if (!pv || !pv->tail) { return; }
   hypre_ParAMGData *t = pv->tail;
   // the tail owns its level 0 too (amg_free_hierarchy leaves level 0 to the caller)
   hypre_ParCSRMatrix *A0 = t->A_array ? t->A_array[0] : nullptr;
   hypre_ParVector *F0 = t->F_array ? t->F_array[0] : nullptr, *U0 = t->U_array ? t->U_array[0] : nullptr;
   HYPRE_BoomerAMGDestroy((HYPRE_Solver) t);
   hypre_ParCSRMatrixDestroy(A0);
   hypre_ParVectorDestroy(F0);
   hypre_ParVectorDestroy(U0);
   if (pv->d_tail_f) { hypre_Free(pv->d_tail_f, HYPRE_MEMORY_DEVICE); pv->d_tail_f = nullptr; }
   pv->tail = nullptr;
   pv->tail_level = -1;
}

}  // namespace hamd
