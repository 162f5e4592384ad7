// hypre_amd — distributed (multi-rank) pieces of the BoomerAMG host setup:
// fetching ghost rows, extended+i interpolation across rank boundaries and
// the Galerkin product with off-rank contributions.  Called by the entry
// points in par_amg_setup.cpp when the communicator has more than one rank.
//
// Reference counterparts:
//   parcsr_mv/par_csr_matop.c:1236-1720    ExtractBExt (ghost rows, optional filters)
//   parcsr_mv/par_csr_matop.c (ExchangeExternalRows)   rows travelling ghost -> owner
//   parcsr_ls/aux_interp.c:351-776         new off-rank nodes, extended comm package
//   parcsr_ls/par_lr_interp.c:1024-1700    extended+i interpolation, off-rank branches
//   parcsr_ls/aux_interp.c:777-900         column map of P's off-rank block
//   parcsr_ls/par_rap.c:30-2000            RAP with P_ext and RAP_ext
#include "amg_internal.hpp"
#include <omp.h>
#include <algorithm>
#include <cmath>

using namespace hamd;

namespace {

int comm_size(MPI_Comm c) { HYPRE_Int n; hypre_MPI_Comm_size(c, &n); return n; }

// ---------------------------------------------------------------------------
// variable-length rows through a comm package
// ---------------------------------------------------------------------------
struct ExtCSR
{
   std::vector<HYPRE_Int>    i;       // [nrows+1]
   std::vector<HYPRE_BigInt> j;       // global column ids (or encoded local/ghost ids later)
   std::vector<HYPRE_Real>   a;       // may stay empty
   HYPRE_Int nrows() const { return (HYPRE_Int) i.size() - 1; }
};

// forward = owner -> ghost: `rows` holds one row per send_map entry (in send_map
// order); the result holds one row per ghost column.  reverse = ghost -> owner.
ExtCSR exchange_rows(hypre_ParCSRCommPkg *pkg, const ExtCSR &rows, bool forward, bool with_data)
{
   const HYPRE_Int ns = pkg->num_sends, nr = pkg->num_recvs;
   const HYPRE_Int nsend_slots = pkg->send_map_starts[ns], nrecv_slots = pkg->recv_vec_starts[nr];
   const HYPRE_Int n_in = forward ? nsend_slots : nrecv_slots;
   const HYPRE_Int n_out = forward ? nrecv_slots : nsend_slots;
   std::vector<HYPRE_Int> len_in((size_t) std::max(n_in, 1)), len_out((size_t) std::max(n_out, 1));
   for (HYPRE_Int k = 0; k < n_in; k++) { len_in[(size_t) k] = rows.i[(size_t) k + 1] - rows.i[(size_t) k]; }
   {
      hypre_ParCSRCommHandle *h = hypre_ParCSRCommHandleCreate(forward ? 11 : 12, pkg, len_in.data(), len_out.data());
      hypre_ParCSRCommHandleDestroy(h);
   }
   ExtCSR out;
   out.i.assign((size_t) n_out + 1, 0);
   for (HYPRE_Int k = 0; k < n_out; k++) { out.i[(size_t) k + 1] = out.i[(size_t) k] + len_out[(size_t) k]; }
   // temporary package whose "map" counts entries instead of rows
   hypre_ParCSRCommPkg tmp;
   memset(&tmp, 0, sizeof(tmp));
   tmp.comm = pkg->comm;
   tmp.num_sends = ns; tmp.num_recvs = nr;
   tmp.send_procs = pkg->send_procs; tmp.recv_procs = pkg->recv_procs;
   std::vector<HYPRE_Int> s_starts((size_t) ns + 1, 0), r_starts((size_t) nr + 1, 0);
   const std::vector<HYPRE_Int> &owner_i = forward ? rows.i : out.i;    // indexed by send_map slot
   const std::vector<HYPRE_Int> &ghost_i = forward ? out.i : rows.i;    // indexed by ghost column
   for (HYPRE_Int p = 0; p <= ns; p++) { s_starts[(size_t) p] = owner_i[(size_t) pkg->send_map_starts[p]]; }
   for (HYPRE_Int p = 0; p <= nr; p++) { r_starts[(size_t) p] = ghost_i[(size_t) pkg->recv_vec_starts[p]]; }
   tmp.send_map_starts = s_starts.data();
   tmp.recv_vec_starts = r_starts.data();
   const size_t nnz_out = (size_t) out.i[(size_t) n_out];
   out.j.resize(std::max<size_t>(nnz_out, 1));
   {
      hypre_ParCSRCommHandle *h = hypre_ParCSRCommHandleCreate(forward ? 21 : 22, &tmp, (void *) rows.j.data(), out.j.data());
      hypre_ParCSRCommHandleDestroy(h);
   }
   if (with_data)
   {
      out.a.resize(std::max<size_t>(nnz_out, 1));
      hypre_ParCSRCommHandle *h = hypre_ParCSRCommHandleCreate(forward ? 1 : 2, &tmp, (void *) rows.a.data(), out.a.data());
      hypre_ParCSRCommHandleDestroy(h);
   }
   out.j.resize(nnz_out);
   if (with_data) { out.a.resize(nnz_out); }
   return out;
}

// Rows of B for the ghost columns of the package (ExtractBExt).  filter:
//   0 everything; 1 entries whose column is a C point (skip_fine);
//   2 off-diagonal entries of sign opposite to the diagonal, local ones only if C (skip_fine && skip_same_sign)
ExtCSR extract_ext(hypre_ParCSRMatrix *B, hypre_ParCSRCommPkg *pkg, bool with_data, int filter,
                   const HYPRE_Int *CF, const HYPRE_Int *CF_offd)
{
   hypre_CSRMatrix *D = B->diag, *O = B->offd;
   const HYPRE_BigInt first_col = B->first_col_diag;
   const HYPRE_Int tot = pkg->send_map_starts[pkg->num_sends];
   ExtCSR rows;
   rows.i.assign((size_t) tot + 1, 0);
   for (HYPRE_Int s = 0; s < tot; s++)
   {
      const HYPRE_Int r = pkg->send_map_elmts[s];
      if (filter == 2)
      {
         const bool pos = D->data[D->i[r]] >= 0;
         for (HYPRE_Int k = D->i[r] + 1; k < D->i[r + 1]; k++)
         {
            const bool opp = pos ? D->data[k] < 0 : D->data[k] > 0;
            if (opp && CF[D->j[k]] >= 0) { rows.j.push_back((HYPRE_BigInt) D->j[k] + first_col); if (with_data) { rows.a.push_back(D->data[k]); } }
         }
         for (HYPRE_Int k = O->i[r]; k < O->i[r + 1]; k++)
         {
            const bool opp = pos ? O->data[k] < 0 : O->data[k] > 0;
            if (opp) { rows.j.push_back(B->col_map_offd[O->j[k]]); if (with_data) { rows.a.push_back(O->data[k]); } }
         }
      }
      else
      {
         for (HYPRE_Int k = D->i[r]; k < D->i[r + 1]; k++)
         {
            if (filter == 1 && !(CF[D->j[k]] >= 0)) { continue; }
            rows.j.push_back((HYPRE_BigInt) D->j[k] + first_col);
            if (with_data) { rows.a.push_back(D->data[k]); }
         }
         for (HYPRE_Int k = O->i[r]; k < O->i[r + 1]; k++)
         {
            if (filter == 1 && !(CF_offd[O->j[k]] >= 0)) { continue; }
            rows.j.push_back(B->col_map_offd[O->j[k]]);
            if (with_data) { rows.a.push_back(O->data[k]); }
         }
      }
      rows.i[(size_t) s + 1] = (HYPRE_Int) rows.j.size();
   }
   if (rows.j.empty()) { rows.j.push_back(0); if (with_data) { rows.a.push_back(0.0); } }
   return exchange_rows(pkg, rows, true, with_data);
}

template <class T>
void halo_fwd(hypre_ParCSRCommPkg *pkg, const T *local, T *ghost, int job)
{
   const HYPRE_Int tot = pkg->send_map_starts[pkg->num_sends];
   std::vector<T> buf((size_t) std::max(tot, 1));
   for (HYPRE_Int k = 0; k < tot; k++) { buf[(size_t) k] = local[pkg->send_map_elmts[k]]; }
   hypre_ParCSRCommHandle *h = hypre_ParCSRCommHandleCreate(job, pkg, buf.data(), ghost);
   hypre_ParCSRCommHandleDestroy(h);
}

HYPRE_Int bsearch_big(const HYPRE_BigInt *list, HYPRE_BigInt v, HYPRE_Int n)
{
   const HYPRE_BigInt *p = std::lower_bound(list, list + n, v);
   return (p != list + n && *p == v) ? (HYPRE_Int) (p - list) : -1;
}

// New off-rank nodes of the extended+i interpolation (aux_interp.c:351-560): the columns of ghost F rows (of the
// fetched rows Aext and Sop) that are neither local nor ghosts of A become `found` (ascending); every off-rank column
// of those rows is re-encoded as -(k) - 1 with k its position among A's ghosts, or nco + its position in `found`.
void ghost_row_numbering(const HYPRE_BigInt *col_map_offd, HYPRE_Int nco, HYPRE_BigInt col_1, HYPRE_BigInt col_n,
                         const std::vector<HYPRE_Int> &CF_offd, ExtCSR &Aext, ExtCSR &Sop, std::vector<HYPRE_BigInt> &found)
{
   for (HYPRE_Int i = 0; i < nco; i++)
   {
      if (CF_offd[(size_t) i] < 0)
      {
         for (HYPRE_Int k = Aext.i[(size_t) i]; k < Aext.i[(size_t) i + 1]; k++)
         {
            const HYPRE_BigInt g = Aext.j[(size_t) k];
            if (g < col_1 || g >= col_n)
            {
               const HYPRE_Int f = bsearch_big(col_map_offd, g, nco);
               if (f == -1) { found.push_back(g); } else { Aext.j[(size_t) k] = (HYPRE_BigInt) (-f - 1); }
            }
         }
         for (HYPRE_Int k = Sop.i[(size_t) i]; k < Sop.i[(size_t) i + 1]; k++)
         {
            const HYPRE_BigInt g = Sop.j[(size_t) k];
            if (g < col_1 || g >= col_n)
            {
               const HYPRE_Int f = bsearch_big(col_map_offd, g, nco);
               if (f == -1) { found.push_back(g); } else { Sop.j[(size_t) k] = (HYPRE_BigInt) (-f - 1); }
            }
         }
      }
   }
   std::sort(found.begin(), found.end());
   found.erase(std::unique(found.begin(), found.end()), found.end());
   const HYPRE_Int newoff = (HYPRE_Int) found.size();
   for (HYPRE_Int i = 0; i < nco; i++)
   {
      if (CF_offd[(size_t) i] < 0)
      {
         for (HYPRE_Int k = Sop.i[(size_t) i]; k < Sop.i[(size_t) i + 1]; k++)
         {
            const HYPRE_BigInt g = Sop.j[(size_t) k];
            if (g > -1 && (g < col_1 || g >= col_n))
            {
               const HYPRE_Int loc = bsearch_big(found.data(), g, newoff);
               if (loc > -1) { Sop.j[(size_t) k] = (HYPRE_BigInt) (-(loc + nco) - 1); }
            }
         }
         for (HYPRE_Int k = Aext.i[(size_t) i]; k < Aext.i[(size_t) i + 1]; k++)
         {
            const HYPRE_BigInt g = Aext.j[(size_t) k];
            if (g > -1 && (g < col_1 || g >= col_n))
            {
               const HYPRE_Int loc = bsearch_big(found.data(), g, newoff);
               if (loc > -1) { Aext.j[(size_t) k] = (HYPRE_BigInt) (-(loc + nco) - 1); }
            }
         }
      }
   }
}

}  // namespace

namespace hamd {

// ===========================================================================
// extended+i interpolation, distributed (par_lr_interp.c:1024-1700)
// ===========================================================================
HYPRE_Int dist_build_extpi_interp(hypre_ParCSRMatrix *A, HYPRE_Int *CF_marker, hypre_ParCSRMatrix *S,
                                  HYPRE_BigInt *num_cpts_global, HYPRE_BigInt total_global_cpts,
                                  const HYPRE_Int *dof_func, HYPRE_Real trunc_factor, HYPRE_Int max_elmts,
                                  hypre_ParCSRMatrix **P_ptr)
{
   MPI_Comm comm = A->comm;
   if (!A->comm_pkg) { hypre_MatvecCommPkgCreate(A); }
   hypre_ParCSRCommPkg *pkg = A->comm_pkg;
   // systems: functions of the ghost columns of A (par_lr_interp.c:1706-1713, 1781-1787 need no more)
   std::vector<HYPRE_Int> dof_offd;
   if (dof_func && A->offd->num_cols)
   {
      dof_offd.resize((size_t) A->offd->num_cols);
      halo_fwd<HYPRE_Int>(pkg, dof_func, dof_offd.data(), 11);
   }
   hypre_CSRMatrix *Ad = A->diag, *Ao = A->offd;
   const HYPRE_Int *Adi = Ad->i, *Adj = Ad->j, *Aoi = Ao->i, *Aoj = Ao->j;
   const HYPRE_Real *Ada = Ad->data, *Aoa = Ao->data;
   const HYPRE_Int *Sdi = S->diag->i, *Sdj = S->diag->j, *Soi = S->offd->i, *Soj = S->offd->j;
   const HYPRE_Int n = Ad->num_rows, nco = Ao->num_cols;
   const HYPRE_BigInt col_1 = A->first_row_index, col_n = col_1 + n;
   const HYPRE_BigInt my_first_cpt = num_cpts_global[0];

   // ---- hypre_exchange_interp_data ----------------------------------------------------
   std::vector<HYPRE_Int> CF_offd((size_t) std::max(nco, 1));
   halo_fwd<HYPRE_Int>(pkg, CF_marker, CF_offd.data(), 11);
   ExtCSR Aext = extract_ext(A, pkg, true, 2, CF_marker, CF_offd.data());
   ExtCSR Sop  = extract_ext(S, pkg, false, 1, CF_marker, CF_offd.data());
   std::vector<HYPRE_BigInt> found;
   ghost_row_numbering(A->col_map_offd, nco, col_1, col_n, CF_offd, Aext, Sop, found);
   const HYPRE_Int newoff = (HYPRE_Int) found.size();
   const HYPRE_Int full_off = nco + newoff;
   // package for the new nodes only (collective: every rank builds one, possibly empty)
   hypre_ParCSRCommPkg ext_pkg;
   memset(&ext_pkg, 0, sizeof(ext_pkg));
   ext_pkg.comm = comm;
   hypre_ParCSRCommPkgCreate_core(comm, found.empty() ? nullptr : found.data(), A->first_col_diag, A->col_starts,
                                  Ad->num_cols, newoff, &ext_pkg.num_recvs, &ext_pkg.recv_procs,
                                  &ext_pkg.recv_vec_starts, &ext_pkg.num_sends, &ext_pkg.send_procs,
                                  &ext_pkg.send_map_starts, &ext_pkg.send_map_elmts);
   CF_offd.resize((size_t) std::max(full_off, 1));
   halo_fwd<HYPRE_Int>(&ext_pkg, CF_marker, CF_offd.data() + nco, 11);

   // ---- fine -> coarse numbering, local and for every (extended) ghost ----------------
   std::vector<HYPRE_Int> f2c((size_t) std::max(n, 1), -1);
   {
      HYPRE_Int c = 0;
      for (HYPRE_Int i = 0; i < n; i++) { if (CF_marker[i] >= 0) { f2c[(size_t) i] = c++; } }
   }
   std::vector<HYPRE_BigInt> f2c_big((size_t) std::max(n, 1)), f2c_offd((size_t) std::max(full_off, 1), -1);
   for (HYPRE_Int i = 0; i < n; i++) { f2c_big[(size_t) i] = (HYPRE_BigInt) f2c[(size_t) i] + my_first_cpt; }
   halo_fwd<HYPRE_BigInt>(pkg, f2c_big.data(), f2c_offd.data(), 21);
   halo_fwd<HYPRE_BigInt>(&ext_pkg, f2c_big.data(), f2c_offd.data() + nco, 21);

   // ---- rows of P: contiguous row blocks, one per thread, each with its own markers and output (a row only
   // reads A, S, the ghost rows and the CF / coarse-index arrays) ------------------------------------------------
   std::vector<HYPRE_Int> Pdi((size_t) n + 1, 0), Poi((size_t) n + 1, 0);
   const int T = std::max(1, std::min(omp_get_max_threads(), n / 2048 + 1));
   std::vector<std::vector<HYPRE_Int>> t_pdj((size_t) T), t_poj((size_t) T);
   std::vector<std::vector<HYPRE_Real>> t_pda((size_t) T), t_poa((size_t) T);
#pragma omp parallel num_threads(T)
   {
   const int tid = omp_get_thread_num();
   const HYPRE_Int row_b = (HYPRE_Int) ((long long) n * tid / T), row_e = (HYPRE_Int) ((long long) n * (tid + 1) / T);
   std::vector<HYPRE_Int> &pdj = t_pdj[(size_t) tid], &poj = t_poj[(size_t) tid];
   std::vector<HYPRE_Real> &pda = t_pda[(size_t) tid], &poa = t_poa[(size_t) tid];
   std::vector<long long> mk((size_t) std::max(n, 1), -1), mko((size_t) std::max(full_off, 1), -1);
   long long strong_f = -2;
   for (HYPRE_Int i = row_b; i < row_e; i++)
   {
      const long long bd = (long long) pdj.size(), bo = (long long) poj.size();
      if (CF_marker[i] >= 0) { pdj.push_back(f2c[(size_t) i]); pda.push_back(1.0); }
      else if (CF_marker[i] != -3)
      {
         strong_f--;
         for (HYPRE_Int jj = Sdi[i]; jj < Sdi[i + 1]; jj++)
         {
            const HYPRE_Int i1 = Sdj[jj];
            if (CF_marker[i1] >= 0)
            {
               if (mk[(size_t) i1] < bd) { mk[(size_t) i1] = (long long) pdj.size(); pdj.push_back(f2c[(size_t) i1]); pda.push_back(0.0); }
            }
            else if (CF_marker[i1] != -3)
            {
               mk[(size_t) i1] = strong_f;
               for (HYPRE_Int kk = Sdi[i1]; kk < Sdi[i1 + 1]; kk++)
               {
                  const HYPRE_Int k1 = Sdj[kk];
                  if (CF_marker[k1] >= 0 && mk[(size_t) k1] < bd) { mk[(size_t) k1] = (long long) pdj.size(); pdj.push_back(f2c[(size_t) k1]); pda.push_back(0.0); }
               }
               for (HYPRE_Int kk = Soi[i1]; kk < Soi[i1 + 1]; kk++)
               {
                  const HYPRE_Int k1 = Soj[kk];
                  if (CF_offd[(size_t) k1] >= 0 && mko[(size_t) k1] < bo) { mko[(size_t) k1] = (long long) poj.size(); poj.push_back(k1); poa.push_back(0.0); }
               }
            }
         }
         for (HYPRE_Int jj = Soi[i]; jj < Soi[i + 1]; jj++)
         {
            const HYPRE_Int i1 = Soj[jj];
            if (CF_offd[(size_t) i1] >= 0)
            {
               if (mko[(size_t) i1] < bo) { mko[(size_t) i1] = (long long) poj.size(); poj.push_back(i1); poa.push_back(0.0); }
            }
            else if (CF_offd[(size_t) i1] != -3)
            {
               mko[(size_t) i1] = strong_f;
               for (HYPRE_Int kk = Sop.i[(size_t) i1]; kk < Sop.i[(size_t) i1 + 1]; kk++)
               {
                  const HYPRE_BigInt g = Sop.j[(size_t) kk];
                  if (g >= col_1 && g < col_n)
                  {
                     const HYPRE_Int lc = (HYPRE_Int) (g - col_1);
                     if (mk[(size_t) lc] < bd) { mk[(size_t) lc] = (long long) pdj.size(); pdj.push_back(f2c[(size_t) lc]); pda.push_back(0.0); }
                  }
                  else
                  {
                     const HYPRE_Int lc = (HYPRE_Int) (-g - 1);
                     if (mko[(size_t) lc] < bo) { mko[(size_t) lc] = (long long) poj.size(); poj.push_back(lc); poa.push_back(0.0); }
                  }
               }
            }
         }
         HYPRE_Real diagonal = Ada[Adi[i]];
         for (HYPRE_Int jj = Adi[i] + 1; jj < Adi[i + 1]; jj++)
         {
            const HYPRE_Int i1 = Adj[jj];
            if (mk[(size_t) i1] >= bd) { pda[(size_t) mk[(size_t) i1]] += Ada[jj]; }
            else if (mk[(size_t) i1] == strong_f)
            {
               HYPRE_Real sum = 0.0;
               const int sgn = Ada[Adi[i1]] < 0 ? -1 : 1;
               for (HYPRE_Int j1 = Adi[i1] + 1; j1 < Adi[i1 + 1]; j1++)
               {
                  const HYPRE_Int i2 = Adj[j1];
                  if ((mk[(size_t) i2] >= bd || i2 == i) && (sgn * Ada[j1]) < 0) { sum += Ada[j1]; }
               }
               for (HYPRE_Int j1 = Aoi[i1]; j1 < Aoi[i1 + 1]; j1++)
               {
                  const HYPRE_Int i2 = Aoj[j1];
                  if (mko[(size_t) i2] >= bo && (sgn * Aoa[j1]) < 0) { sum += Aoa[j1]; }
               }
               if (sum != 0)
               {
                  const HYPRE_Real distribute = Ada[jj] / sum;
                  for (HYPRE_Int j1 = Adi[i1] + 1; j1 < Adi[i1 + 1]; j1++)
                  {
                     const HYPRE_Int i2 = Adj[j1];
                     if (mk[(size_t) i2] >= bd && (sgn * Ada[j1]) < 0) { pda[(size_t) mk[(size_t) i2]] += distribute * Ada[j1]; }
                     if (i2 == i && (sgn * Ada[j1]) < 0) { diagonal += distribute * Ada[j1]; }
                  }
                  for (HYPRE_Int j1 = Aoi[i1]; j1 < Aoi[i1 + 1]; j1++)
                  {
                     const HYPRE_Int i2 = Aoj[j1];
                     if (mko[(size_t) i2] >= bo && (sgn * Aoa[j1]) < 0) { poa[(size_t) mko[(size_t) i2]] += distribute * Aoa[j1]; }
                  }
               }
               else { diagonal += Ada[jj]; }
            }
            else if (CF_marker[i1] != -3) { if (!dof_func || dof_func[i] == dof_func[i1]) { diagonal += Ada[jj]; } }
         }
         for (HYPRE_Int jj = Aoi[i]; jj < Aoi[i + 1]; jj++)
         {
            const HYPRE_Int i1 = Aoj[jj];
            if (mko[(size_t) i1] >= bo) { poa[(size_t) mko[(size_t) i1]] += Aoa[jj]; }
            else if (mko[(size_t) i1] == strong_f)
            {
               HYPRE_Real sum = 0.0;
               for (HYPRE_Int j1 = Aext.i[(size_t) i1]; j1 < Aext.i[(size_t) i1 + 1]; j1++)
               {
                  const HYPRE_BigInt g = Aext.j[(size_t) j1];
                  if (g >= col_1 && g < col_n)
                  {
                     const HYPRE_Int lc = (HYPRE_Int) (g - col_1);
                     if (mk[(size_t) lc] >= bd || lc == i) { sum += Aext.a[(size_t) j1]; }
                  }
                  else
                  {
                     const HYPRE_Int lc = (HYPRE_Int) (-g - 1);
                     if (mko[(size_t) lc] >= bo) { sum += Aext.a[(size_t) j1]; }
                  }
               }
               if (sum != 0)
               {
                  const HYPRE_Real distribute = Aoa[jj] / sum;
                  for (HYPRE_Int j1 = Aext.i[(size_t) i1]; j1 < Aext.i[(size_t) i1 + 1]; j1++)
                  {
                     const HYPRE_BigInt g = Aext.j[(size_t) j1];
                     if (g >= col_1 && g < col_n)
                     {
                        const HYPRE_Int lc = (HYPRE_Int) (g - col_1);
                        if (mk[(size_t) lc] >= bd) { pda[(size_t) mk[(size_t) lc]] += distribute * Aext.a[(size_t) j1]; }
                        if (lc == i) { diagonal += distribute * Aext.a[(size_t) j1]; }
                     }
                     else
                     {
                        const HYPRE_Int lc = (HYPRE_Int) (-g - 1);
                        if (mko[(size_t) lc] >= bo) { poa[(size_t) mko[(size_t) lc]] += distribute * Aext.a[(size_t) j1]; }
                     }
                  }
               }
               else { diagonal += Aoa[jj]; }
            }
            else if (CF_offd[(size_t) i1] != -3) { if (!dof_func || dof_func[i] == dof_offd[(size_t) i1]) { diagonal += Aoa[jj]; } }
         }
         if (diagonal)
         {
            for (size_t k = (size_t) bd; k < pdj.size(); k++) { pda[k] /= -diagonal; }
            for (size_t k = (size_t) bo; k < poj.size(); k++) { poa[k] /= -diagonal; }
         }
         strong_f--;
      }
      Pdi[(size_t) i + 1] = (HYPRE_Int) ((long long) pdj.size() - bd);      // row lengths; offsets after the region
      Poi[(size_t) i + 1] = (HYPRE_Int) ((long long) poj.size() - bo);
   }
   }  // parallel region
   for (HYPRE_Int i = 0; i < n; i++) { Pdi[(size_t) i + 1] += Pdi[(size_t) i]; Poi[(size_t) i + 1] += Poi[(size_t) i]; }

   HYPRE_BigInt cs[2] = {num_cpts_global[0], num_cpts_global[1]};
   hypre_ParCSRMatrix *P = hypre_ParCSRMatrixCreate(comm, A->global_num_rows, total_global_cpts, A->col_starts, cs,
                                                    full_off, Pdi[(size_t) n], Poi[(size_t) n]);
   hypre_CSRMatrixInitialize_v2(P->diag, 0, HYPRE_MEMORY_HOST);
   hypre_CSRMatrixInitialize_v2(P->offd, 0, HYPRE_MEMORY_HOST);
   memcpy(P->diag->i, Pdi.data(), sizeof(HYPRE_Int) * ((size_t) n + 1));
   memcpy(P->offd->i, Poi.data(), sizeof(HYPRE_Int) * ((size_t) n + 1));
#pragma omp parallel for num_threads(T) schedule(static, 1)
   for (int t = 0; t < T; t++)
   {
      const HYPRE_Int row_b = (HYPRE_Int) ((long long) n * t / T);
      const size_t od = (size_t) Pdi[(size_t) row_b], oo = (size_t) Poi[(size_t) row_b];
      if (!t_pdj[(size_t) t].empty())
      {
         memcpy(P->diag->j + od, t_pdj[(size_t) t].data(), sizeof(HYPRE_Int) * t_pdj[(size_t) t].size());
         memcpy(P->diag->data + od, t_pda[(size_t) t].data(), sizeof(HYPRE_Real) * t_pda[(size_t) t].size());
      }
      if (!t_poj[(size_t) t].empty())
      {
         memcpy(P->offd->j + oo, t_poj[(size_t) t].data(), sizeof(HYPRE_Int) * t_poj[(size_t) t].size());
         memcpy(P->offd->data + oo, t_poa[(size_t) t].data(), sizeof(HYPRE_Real) * t_poa[(size_t) t].size());
      }
   }
   if (trunc_factor != 0.0 || max_elmts > 0) { hypre_BoomerAMGInterpTruncation(P, trunc_factor, max_elmts); }

   // ---- column map of the off-rank block (aux_interp.c:777-900): used ghosts, ascending coarse id
   {
      const HYPRE_Int nnz_o = P->offd->i[n];
      std::vector<char> used((size_t) std::max(full_off, 1), 0);
      for (HYPRE_Int k = 0; k < nnz_o; k++) { used[(size_t) P->offd->j[k]] = 1; }
      std::vector<HYPRE_BigInt> cmap;
      for (HYPRE_Int g = 0; g < full_off; g++) { if (used[(size_t) g]) { cmap.push_back(f2c_offd[(size_t) g]); } }
      std::sort(cmap.begin(), cmap.end());
      cmap.erase(std::unique(cmap.begin(), cmap.end()), cmap.end());
      for (HYPRE_Int k = 0; k < nnz_o; k++)
      {
         P->offd->j[k] = (HYPRE_Int) (std::lower_bound(cmap.begin(), cmap.end(), f2c_offd[(size_t) P->offd->j[k]]) - cmap.begin());
      }
      P->offd->num_cols = (HYPRE_Int) cmap.size();
      if (!cmap.empty())
      {
         P->col_map_offd = hypre_TAlloc(HYPRE_BigInt, cmap.size(), HYPRE_MEMORY_HOST);
         memcpy(P->col_map_offd, cmap.data(), sizeof(HYPRE_BigInt) * cmap.size());
      }
   }
   hypre_CSRMatrixSetRownnz(P->offd);
   hypre_MatvecCommPkgCreate(P);
   hypre_Free(ext_pkg.recv_procs, HYPRE_MEMORY_HOST); hypre_Free(ext_pkg.recv_vec_starts, HYPRE_MEMORY_HOST);
   hypre_Free(ext_pkg.send_procs, HYPRE_MEMORY_HOST); hypre_Free(ext_pkg.send_map_starts, HYPRE_MEMORY_HOST);
   hypre_Free(ext_pkg.send_map_elmts, HYPRE_MEMORY_HOST);
   *P_ptr = P;
   return hypre_error_flag;
}

// ===========================================================================
// Galerkin product, distributed (par_rap.c:30-2000)
// ===========================================================================
HYPRE_Int dist_build_coarse_operator(hypre_ParCSRMatrix *RT, hypre_ParCSRMatrix *A, hypre_ParCSRMatrix *P,
                                     HYPRE_Int keepTranspose, hypre_ParCSRMatrix **RAP_ptr)
{
   MPI_Comm comm = A->comm;
   if (!A->comm_pkg) { hypre_MatvecCommPkgCreate(A); }
   if (!RT->comm_pkg) { hypre_MatvecCommPkgCreate(RT); }
   hypre_ParCSRCommPkg *pkgA = A->comm_pkg, *pkgRT = RT->comm_pkg;
   hypre_CSRMatrix *Ad = A->diag, *Ao = A->offd, *Pd = P->diag, *Po = P->offd;
   const HYPRE_Int ncP = Pd->num_cols, ncoP = Po->num_cols, ncoA = Ao->num_cols, ncoRT = RT->offd->num_cols;
   const HYPRE_Int ncRT = RT->diag->num_cols;
   const HYPRE_BigInt first_c = P->first_col_diag, last_c = first_c + ncP - 1;
   const bool square = (ncRT == ncP);
   hypre_CSRMatrix *Rd = nullptr, *Ro = nullptr;
   hypre_CSRMatrixTranspose(RT->diag, &Rd, 1);
   if (ncoRT) { hypre_CSRMatrixTranspose(RT->offd, &Ro, 1); }

   // ---- P_ext: rows of P for A's ghost columns, split into local-coarse and off-rank-coarse parts
   ExtCSR Ps = extract_ext(P, pkgA, true, 0, nullptr, nullptr);
   std::vector<HYPRE_Int> Pedi((size_t) ncoA + 1, 0), Peoi((size_t) ncoA + 1, 0), Pedj, Peoj;
   std::vector<HYPRE_Real> Peda, Peoa;
   std::vector<HYPRE_BigInt> Pe_big;
   for (HYPRE_Int i = 0; i < ncoA; i++)
   {
      for (HYPRE_Int k = Ps.i[(size_t) i]; k < Ps.i[(size_t) i + 1]; k++)
      {
         const HYPRE_BigInt g = Ps.j[(size_t) k];
         if (g < first_c || g > last_c) { Pe_big.push_back(g); Peoa.push_back(Ps.a[(size_t) k]); }
         else { Pedj.push_back((HYPRE_Int) (g - first_c)); Peda.push_back(Ps.a[(size_t) k]); }
      }
      Pedi[(size_t) i + 1] = (HYPRE_Int) Pedj.size();
      Peoi[(size_t) i + 1] = (HYPRE_Int) Pe_big.size();
   }
   std::vector<HYPRE_BigInt> cmapPext(Pe_big);
   for (HYPRE_Int k = 0; k < ncoP; k++) { cmapPext.push_back(P->col_map_offd[k]); }
   std::sort(cmapPext.begin(), cmapPext.end());
   cmapPext.erase(std::unique(cmapPext.begin(), cmapPext.end()), cmapPext.end());
   const HYPRE_Int ncoPext = (HYPRE_Int) cmapPext.size();
   Peoj.resize(Pe_big.size());
   for (size_t k = 0; k < Pe_big.size(); k++) { Peoj[k] = bsearch_big(cmapPext.data(), Pe_big[k], ncoPext); }
   std::vector<HYPRE_Int> mapP2Pext((size_t) std::max(ncoP, 1));
   for (HYPRE_Int k = 0; k < ncoP; k++) { mapP2Pext[(size_t) k] = bsearch_big(cmapPext.data(), P->col_map_offd[k], ncoPext); }

   // ---- RAP_int: rows of the product that belong to other ranks (one per ghost coarse column of RT)
   ExtCSR Rint;
   Rint.i.assign((size_t) ncoRT + 1, 0);
   {
      std::vector<long long> Pm((size_t) std::max(ncP + ncoPext, 1), -1);
      for (HYPRE_Int ic = 0; ic < ncoRT; ic++)
      {
         const long long begin = (long long) Rint.j.size();
         for (HYPRE_Int j1 = Ro->i[ic]; j1 < Ro->i[ic + 1]; j1++)
         {
            const HYPRE_Int i1 = Ro->j[j1];
            const HYPRE_Real r = Ro->data[j1];
            for (HYPRE_Int j2 = Ao->i[i1]; j2 < Ao->i[i1 + 1]; j2++)
            {
               const HYPRE_Int i2 = Ao->j[j2];
               const HYPRE_Real ra = r * Ao->data[j2];
               for (HYPRE_Int j3 = Pedi[(size_t) i2]; j3 < Pedi[(size_t) i2 + 1]; j3++)
               {
                  const HYPRE_Int i3 = Pedj[(size_t) j3];
                  const HYPRE_Real v = ra * Peda[(size_t) j3];
                  if (Pm[(size_t) i3] < begin) { Pm[(size_t) i3] = (long long) Rint.j.size(); Rint.j.push_back((HYPRE_BigInt) i3 + first_c); Rint.a.push_back(v); }
                  else { Rint.a[(size_t) Pm[(size_t) i3]] += v; }
               }
               for (HYPRE_Int j3 = Peoi[(size_t) i2]; j3 < Peoi[(size_t) i2 + 1]; j3++)
               {
                  const HYPRE_Int i3 = Peoj[(size_t) j3] + ncP;
                  const HYPRE_Real v = ra * Peoa[(size_t) j3];
                  if (Pm[(size_t) i3] < begin) { Pm[(size_t) i3] = (long long) Rint.j.size(); Rint.j.push_back(cmapPext[(size_t) (i3 - ncP)]); Rint.a.push_back(v); }
                  else { Rint.a[(size_t) Pm[(size_t) i3]] += v; }
               }
            }
            for (HYPRE_Int j2 = Ad->i[i1]; j2 < Ad->i[i1 + 1]; j2++)
            {
               const HYPRE_Int i2 = Ad->j[j2];
               const HYPRE_Real ra = r * Ad->data[j2];
               for (HYPRE_Int j3 = Pd->i[i2]; j3 < Pd->i[i2 + 1]; j3++)
               {
                  const HYPRE_Int i3 = Pd->j[j3];
                  const HYPRE_Real v = ra * Pd->data[j3];
                  if (Pm[(size_t) i3] < begin) { Pm[(size_t) i3] = (long long) Rint.j.size(); Rint.j.push_back((HYPRE_BigInt) i3 + first_c); Rint.a.push_back(v); }
                  else { Rint.a[(size_t) Pm[(size_t) i3]] += v; }
               }
               for (HYPRE_Int j3 = Po->i[i2]; j3 < Po->i[i2 + 1]; j3++)
               {
                  const HYPRE_Int i3 = mapP2Pext[(size_t) Po->j[j3]] + ncP;
                  const HYPRE_Real v = ra * Po->data[j3];
                  if (Pm[(size_t) i3] < begin) { Pm[(size_t) i3] = (long long) Rint.j.size(); Rint.j.push_back(cmapPext[(size_t) (i3 - ncP)]); Rint.a.push_back(v); }
                  else { Rint.a[(size_t) Pm[(size_t) i3]] += v; }
               }
            }
         }
         Rint.i[(size_t) ic + 1] = (HYPRE_Int) Rint.j.size();
      }
   }
   if (Rint.j.empty()) { Rint.j.push_back(0); Rint.a.push_back(0.0); }
   // ghost -> owner: one received row per send_map entry of RT's package
   ExtCSR Rext = exchange_rows(pkgRT, Rint, false, true);
   const HYPRE_Int nsendRT = pkgRT->send_map_starts[pkgRT->num_sends];

   // ---- column map of the off-rank block of RAP
   std::vector<HYPRE_BigInt> cmapRAP;
   for (size_t k = 0; k < Rext.j.size(); k++) { if (Rext.j[k] < first_c || Rext.j[k] > last_c) { cmapRAP.push_back(Rext.j[k]); } }
   for (HYPRE_Int k = 0; k < ncoPext; k++) { cmapRAP.push_back(cmapPext[(size_t) k]); }
   std::sort(cmapRAP.begin(), cmapRAP.end());
   cmapRAP.erase(std::unique(cmapRAP.begin(), cmapRAP.end()), cmapRAP.end());
   const HYPRE_Int ncoRAP = (HYPRE_Int) cmapRAP.size();
   std::vector<HYPRE_Int> mapP2RAP((size_t) std::max(ncoP, 1)), mapPext2RAP((size_t) std::max(ncoPext, 1));
   for (HYPRE_Int k = 0; k < ncoP; k++) { mapP2RAP[(size_t) k] = bsearch_big(cmapRAP.data(), P->col_map_offd[k], ncoRAP); }
   for (HYPRE_Int k = 0; k < ncoPext; k++) { mapPext2RAP[(size_t) k] = bsearch_big(cmapRAP.data(), cmapPext[(size_t) k], ncoRAP); }
   for (size_t k = 0; k < Rext.j.size(); k++)
   {
      const HYPRE_BigInt g = Rext.j[k];
      Rext.j[k] = (g < first_c || g > last_c) ? (HYPRE_BigInt) ncP + bsearch_big(cmapRAP.data(), g, ncoRAP) : g - first_c;
   }
   // which received rows feed local coarse row ic (in package order)
   std::vector<std::vector<HYPRE_Int>> feeds((size_t) std::max(ncRT, 1));
   for (HYPRE_Int s = 0; s < nsendRT; s++) { feeds[(size_t) pkgRT->send_map_elmts[s]].push_back(s); }

   // ---- local rows: diagonal slot, received contributions, RA_offd*P_ext, RA_diag*P
   std::vector<HYPRE_Int> Cdi((size_t) ncRT + 1, 0), Coi((size_t) ncRT + 1, 0);
   const int T = std::max(1, std::min(omp_get_max_threads(), ncRT / 2048 + 1));
   std::vector<std::vector<HYPRE_Int>> t_cdj((size_t) T), t_coj((size_t) T);
   std::vector<std::vector<HYPRE_Real>> t_cda((size_t) T), t_coa((size_t) T);
#pragma omp parallel num_threads(T)
   {
      const int tid = omp_get_thread_num();
      const HYPRE_Int ic_b = (HYPRE_Int) ((long long) ncRT * tid / T), ic_e = (HYPRE_Int) ((long long) ncRT * (tid + 1) / T);
      std::vector<HYPRE_Int> &cdj = t_cdj[(size_t) tid], &coj = t_coj[(size_t) tid];
      std::vector<HYPRE_Real> &cda = t_cda[(size_t) tid], &coa = t_coa[(size_t) tid];
      std::vector<long long> Pm((size_t) std::max(ncP + ncoRAP, 1), -1);
      std::vector<HYPRE_Int> Am((size_t) std::max(ncoA + Ad->num_cols, 1), -1);
      std::vector<HYPRE_Int> radj, raoj;
      std::vector<HYPRE_Real> rada, raoa;
      auto add_d = [&](HYPRE_Int col, HYPRE_Real v, long long begin)
      {
         if (Pm[(size_t) col] < begin) { Pm[(size_t) col] = (long long) cdj.size(); cdj.push_back(col); cda.push_back(v); }
         else { cda[(size_t) Pm[(size_t) col]] += v; }
      };
      auto add_o = [&](HYPRE_Int col_in_rap, HYPRE_Real v, long long begin)
      {
         const HYPRE_Int slot = col_in_rap + ncP;
         if (Pm[(size_t) slot] < begin) { Pm[(size_t) slot] = (long long) coj.size(); coj.push_back(col_in_rap); coa.push_back(v); }
         else { coa[(size_t) Pm[(size_t) slot]] += v; }
      };
      for (HYPRE_Int ic = ic_b; ic < ic_e; ic++)
      {
         const long long bd = (long long) cdj.size(), bo = (long long) coj.size();
         if (square) { Pm[(size_t) ic] = bd; cdj.push_back(ic); cda.push_back(0.0); }
         for (HYPRE_Int s : feeds[(size_t) ic])
         {
            for (HYPRE_Int k = Rext.i[(size_t) s]; k < Rext.i[(size_t) s + 1]; k++)
            {
               const HYPRE_Int jc = (HYPRE_Int) Rext.j[(size_t) k];
               if (jc < ncP) { add_d(jc, Rext.a[(size_t) k], bd); } else { add_o(jc - ncP, Rext.a[(size_t) k], bo); }
            }
         }
         radj.clear(); rada.clear(); raoj.clear(); raoa.clear();
         for (HYPRE_Int j1 = Rd->i[ic]; j1 < Rd->i[ic + 1]; j1++)
         {
            const HYPRE_Int i1 = Rd->j[j1];
            const HYPRE_Real r = Rd->data[j1];
            for (HYPRE_Int j2 = Ao->i[i1]; j2 < Ao->i[i1 + 1]; j2++)
            {
               const HYPRE_Int i2 = Ao->j[j2];
               const HYPRE_Int m = Am[(size_t) i2];
               if (m < 0 || m >= (HYPRE_Int) raoj.size() || raoj[(size_t) m] != i2) { Am[(size_t) i2] = (HYPRE_Int) raoj.size(); raoj.push_back(i2); raoa.push_back(r * Ao->data[j2]); }
               else { raoa[(size_t) m] += r * Ao->data[j2]; }
            }
            for (HYPRE_Int j2 = Ad->i[i1]; j2 < Ad->i[i1 + 1]; j2++)
            {
               const HYPRE_Int i2 = Ad->j[j2];
               const HYPRE_Int m = Am[(size_t) i2 + ncoA];
               if (m < 0 || m >= (HYPRE_Int) radj.size() || radj[(size_t) m] != i2) { Am[(size_t) i2 + ncoA] = (HYPRE_Int) radj.size(); radj.push_back(i2); rada.push_back(r * Ad->data[j2]); }
               else { rada[(size_t) m] += r * Ad->data[j2]; }
            }
         }
         for (size_t q = 0; q < raoj.size(); q++)
         {
            const HYPRE_Int i1 = raoj[q];
            const HYPRE_Real rap = raoa[q];
            for (HYPRE_Int j2 = Pedi[(size_t) i1]; j2 < Pedi[(size_t) i1 + 1]; j2++) { add_d(Pedj[(size_t) j2], rap * Peda[(size_t) j2], bd); }
            for (HYPRE_Int j2 = Peoi[(size_t) i1]; j2 < Peoi[(size_t) i1 + 1]; j2++) { add_o(mapPext2RAP[(size_t) Peoj[(size_t) j2]], rap * Peoa[(size_t) j2], bo); }
         }
         for (size_t q = 0; q < radj.size(); q++)
         {
            const HYPRE_Int i1 = radj[q];
            const HYPRE_Real rap = rada[q];
            for (HYPRE_Int j2 = Pd->i[i1]; j2 < Pd->i[i1 + 1]; j2++) { add_d(Pd->j[j2], rap * Pd->data[j2], bd); }
            for (HYPRE_Int j2 = Po->i[i1]; j2 < Po->i[i1 + 1]; j2++) { add_o(mapP2RAP[(size_t) Po->j[j2]], rap * Po->data[j2], bo); }
         }
         Cdi[(size_t) ic + 1] = (HYPRE_Int) ((long long) cdj.size() - bd);      // row lengths; offsets below
         Coi[(size_t) ic + 1] = (HYPRE_Int) ((long long) coj.size() - bo);
      }
   }
   for (HYPRE_Int ic = 0; ic < ncRT; ic++) { Cdi[(size_t) ic + 1] += Cdi[(size_t) ic]; Coi[(size_t) ic + 1] += Coi[(size_t) ic]; }
   hypre_ParCSRMatrix *C = hypre_ParCSRMatrixCreate(comm, RT->global_num_cols, P->global_num_cols, RT->col_starts,
                                                    P->col_starts, ncoRAP, Cdi[(size_t) ncRT], Coi[(size_t) ncRT]);
   hypre_CSRMatrixInitialize_v2(C->diag, 0, HYPRE_MEMORY_HOST);
   hypre_CSRMatrixInitialize_v2(C->offd, 0, HYPRE_MEMORY_HOST);
   memcpy(C->diag->i, Cdi.data(), sizeof(HYPRE_Int) * ((size_t) ncRT + 1));
   memcpy(C->offd->i, Coi.data(), sizeof(HYPRE_Int) * ((size_t) ncRT + 1));
#pragma omp parallel for num_threads(T) schedule(static, 1)
   for (int t = 0; t < T; t++)
   {
      const HYPRE_Int ic_b = (HYPRE_Int) ((long long) ncRT * t / T);
      const size_t od = (size_t) Cdi[(size_t) ic_b], oo = (size_t) Coi[(size_t) ic_b];
      if (!t_cdj[(size_t) t].empty())
      {
         memcpy(C->diag->j + od, t_cdj[(size_t) t].data(), sizeof(HYPRE_Int) * t_cdj[(size_t) t].size());
         memcpy(C->diag->data + od, t_cda[(size_t) t].data(), sizeof(HYPRE_Real) * t_cda[(size_t) t].size());
      }
      if (!t_coj[(size_t) t].empty())
      {
         memcpy(C->offd->j + oo, t_coj[(size_t) t].data(), sizeof(HYPRE_Int) * t_coj[(size_t) t].size());
         memcpy(C->offd->data + oo, t_coa[(size_t) t].data(), sizeof(HYPRE_Real) * t_coa[(size_t) t].size());
      }
   }
   if (ncoRAP)
   {
      C->col_map_offd = hypre_TAlloc(HYPRE_BigInt, ncoRAP, HYPRE_MEMORY_HOST);
      memcpy(C->col_map_offd, cmapRAP.data(), sizeof(HYPRE_BigInt) * (size_t) ncoRAP);
   }
   if (keepTranspose) { RT->diagT = Rd; RT->offdT = Ro; }
   else { hypre_CSRMatrixDestroy(Rd); if (Ro) { hypre_CSRMatrixDestroy(Ro); } }
   hypre_CSRMatrixSetRownnz(C->offd);
   hypre_ParCSRMatrixSetNumNonzeros(C);
   hypre_ParCSRMatrixSetDNumNonzeros(C);
   hypre_MatvecCommPkgCreate(C);
   *RAP_ptr = C;
   return hypre_error_flag;
}


// ===========================================================================
// The same steps for a level that lives in DEVICE memory: the single-rank kernels on the extended numbering
// [local points | ghost points] (dist_setup_kernels.hip).  What travels between ranks is what the host routines above
// exchange — boundary-sized row sets, extracted on the device, exchanged and renumbered on the host, appended on the
// device as rows n .. of the extended matrices.  The results equal the host routines' array for array.
// Reference: par_coarsen_device.c:30, par_lr_interp_device.c:1001, par_csr_triplemat.c:938-960 (other formulations).
// ===========================================================================
namespace {

template <class T>
T *upload(const T *h, size_t n)
{
   T *d = nullptr;
   HIP_CHECK(hipMalloc((void **) &d, sizeof(T) * std::max<size_t>(n, 1)));
   if (n)
   {
      HIP_CHECK(hipMemcpyAsync(d, h, sizeof(T) * n, hipMemcpyHostToDevice, stream()));
      HIP_CHECK(hipStreamSynchronize(stream()));
   }
   return d;
}
template <class T>
void fetch(T *h, const T *d, size_t n)
{
   if (!n) { return; }
   HIP_CHECK(hipMemcpyAsync(h, d, sizeof(T) * n, hipMemcpyDeviceToHost, stream()));
   HIP_CHECK(hipStreamSynchronize(stream()));
}
void dfree(void *p) { if (p) { HIP_CHECK(hipFree(p)); } }

// a host CSR (rows of ghost points, extended column numbers) on the device
struct DevCSR
{
   int *i = nullptr, *j = nullptr;
   double *a = nullptr;
   void put(const std::vector<HYPRE_Int> &hi, const std::vector<HYPRE_Int> &hj, const std::vector<HYPRE_Real> *ha)
   {
      i = upload(hi.data(), hi.size());
      j = upload(hj.data(), hj.size());
      if (ha) { a = upload(ha->data(), ha->size()); }
   }
   ~DevCSR() { dfree(i); dfree(j); dfree(a); }
};

const int *send_elmts_on_device(hypre_ParCSRCommPkg *pkg)
{
   const HYPRE_Int tot = pkg->send_map_starts[pkg->num_sends];
   if (!pkg->device_send_map_elmts && tot)
   {
      pkg->device_send_map_elmts = hypre_TAlloc(HYPRE_Int, tot, HYPRE_MEMORY_DEVICE);
      hypre_TMemcpy(pkg->device_send_map_elmts, pkg->send_map_elmts, HYPRE_Int, tot, HYPRE_MEMORY_DEVICE, HYPRE_MEMORY_HOST);
   }
   return pkg->device_send_map_elmts;
}

}  // namespace

bool all_ranks_agree(MPI_Comm comm, bool mine)
{
   return global_sum(comm, mine ? 0.0 : 1.0) == 0.0;
}

// Test hook: rank `rank` pretends that the device kernel of step `what` (1: interpolation, 2: product rows for the
// neighbours, 3: product) declined, `count` times — so that the agreed fall-back to the host routines runs in a test.
static int g_decline_what = 0, g_decline_rank = -1, g_decline_count = 0;
extern "C" HYPRE_Int hypre_amd_SetupDistTestDecline(HYPRE_Int what, HYPRE_Int rank, HYPRE_Int count)
{
   g_decline_what = what; g_decline_rank = rank; g_decline_count = count;
   return hypre_error_flag;
}
static bool test_declines(MPI_Comm comm, int what)
{
   HYPRE_Int me;
   hypre_MPI_Comm_rank(comm, &me);
   if (g_decline_count > 0 && g_decline_what == what && me == g_decline_rank) { g_decline_count--; return true; }
   return false;
}

HYPRE_Int dist_device_create_S(hypre_ParCSRMatrix *A, HYPRE_Real theta, HYPRE_Real max_row_sum, hypre_ParCSRMatrix **S_ptr)
{
   hypre_CSRMatrix *dD = setup_device_twin_of(A->diag, 1), *dO = setup_device_twin_of(A->offd, 1);
   const HYPRE_Int n = A->diag->num_rows, nco = A->offd->num_cols;
   int *Si = nullptr, *Sj = nullptr, snnz = 0;
   device_strength_blocks(n, dD->i, dD->j, dD->data, nco ? dO->i : nullptr, dO->j, dO->data, theta, max_row_sum, &Si, &Sj, &snnz, stream());
   hypre_ParCSRMatrix *S = hypre_ParCSRMatrixCreate(A->comm, A->global_num_rows, A->global_num_rows, A->row_starts, A->row_starts, 0, snnz, 0);
   hypre_CSRMatrixDestroy(S->diag);
   S->diag = setup_wrap_device_csr(n, n + nco, snnz, Si, Sj, nullptr);      // one pattern over the extended numbering
   hypre_CSRMatrixInitialize_v2(S->offd, 0, HYPRE_MEMORY_HOST);
   *S_ptr = S;
   return hypre_error_flag;
}

HYPRE_Int dist_device_pmis(hypre_ParCSRMatrix *S, hypre_ParCSRMatrix *A, HYPRE_Int CF_init, HYPRE_Int *CF_host)
{
   MPI_Comm comm = A->comm;
   HYPRE_Int my_id;
   hypre_MPI_Comm_rank(comm, &my_id);
   if (!A->comm_pkg) { hypre_MatvecCommPkgCreate(A); }
   hypre_ParCSRCommPkg *pkg = A->comm_pkg;
   const HYPRE_Int n = A->diag->num_rows, nco = A->offd->num_cols;
   const HYPRE_Int tot = pkg->send_map_starts[pkg->num_sends];
   const int *d_elmts = send_elmts_on_device(pkg);
   // previous slot of the send list that names the same point (a point on an edge or a corner goes to several neighbours)
   std::vector<int> prev((size_t) std::max(tot, 1), -1);
   {
      std::vector<std::pair<int, int>> order((size_t) tot);
      for (HYPRE_Int k = 0; k < tot; k++) { order[(size_t) k] = {pkg->send_map_elmts[k], k}; }
      std::sort(order.begin(), order.end());
      for (HYPRE_Int q = 1; q < tot; q++) { if (order[(size_t) q].first == order[(size_t) q - 1].first) { prev[(size_t) order[(size_t) q].second] = order[(size_t) q - 1].second; } }
   }
   int *d_prev = upload(prev.data(), (size_t) tot);
   HYPRE_Int *dCF = hypre_TAlloc(HYPRE_Int, (size_t) std::max(n + nco, 1), HYPRE_MEMORY_DEVICE);
   hypre_CSRMatrix *Sd = S->diag;
   device_pmis_dist(n, nco, Sd->i, Sd->j, Sd->num_nonzeros, 2747u + (CF_init == 2 ? 0u : (unsigned) my_id),
                    CF_init == 2 ? (unsigned long long) S->first_row_index : 0ull, pkg, d_elmts, d_prev, comm, dCF, stream());
   fetch(CF_host, dCF, (size_t) n);
   dfree(d_prev);
   setup_register_device_marker(CF_host, dCF);
   return hypre_error_flag;
}

HYPRE_Int dist_device_extpi_interp(hypre_ParCSRMatrix *A, HYPRE_Int *CF_marker, hypre_ParCSRMatrix *S,
                                   HYPRE_BigInt *num_cpts_global, HYPRE_BigInt total_global_cpts, HYPRE_Real trunc_factor,
                                   HYPRE_Int max_elmts, HYPRE_Int first_rung, hypre_ParCSRMatrix **P_ptr)
{
   *P_ptr = nullptr;
   MPI_Comm comm = A->comm;
   hipStream_t st = stream();
   if (!A->comm_pkg) { hypre_MatvecCommPkgCreate(A); }
   hypre_ParCSRCommPkg *pkg = A->comm_pkg;
   hypre_CSRMatrix *dD = setup_device_twin_of(A->diag, 1), *dO = setup_device_twin_of(A->offd, 1), *Sx = S->diag;
   const HYPRE_Int n = A->diag->num_rows, nco = A->offd->num_cols;
   const HYPRE_BigInt col_1 = A->first_row_index, col_n = col_1 + n;
   const HYPRE_BigInt my_first_cpt = num_cpts_global[0];
   const HYPRE_Int ncP = (HYPRE_Int) (num_cpts_global[1] - num_cpts_global[0]);
   const HYPRE_Int tot = pkg->send_map_starts[pkg->num_sends];
   const int *d_elmts = send_elmts_on_device(pkg);
   HYPRE_Int *dCF = setup_device_marker_of(CF_marker, n);          // left by the device coarsening (or uploaded now)

   // ---- markers of the ghost points, the rows of A and S the neighbours hold for them
   std::vector<HYPRE_Int> CF_offd((size_t) std::max(nco, 1));
   halo_fwd<HYPRE_Int>(pkg, CF_marker, CF_offd.data(), 11);
   long long *d_cmap = upload((const long long *) A->col_map_offd, (size_t) nco);
   int *dCFx = nullptr;                                            // markers over [local | ghosts of A]
   HIP_CHECK(hipMalloc((void **) &dCFx, sizeof(int) * (size_t) std::max(n + nco, 1)));
   if (n) { HIP_CHECK(hipMemcpyAsync(dCFx, dCF, sizeof(int) * (size_t) n, hipMemcpyDeviceToDevice, st)); }
   if (nco) { HIP_CHECK(hipMemcpyAsync(dCFx + n, CF_offd.data(), sizeof(int) * (size_t) nco, hipMemcpyHostToDevice, st)); }
   HIP_CHECK(hipStreamSynchronize(st));
   ExtCSR Aext, Sop;
   {
      ExtCSR rows;
      device_extract_rows(2, tot, d_elmts, dD->i, dD->j, dD->data, nco ? dO->i : nullptr, dO->j, dO->data, (long long) A->first_col_diag,
                          d_cmap, dCF, rows.i, rows.j, rows.a, st);
      if (rows.j.empty()) { rows.j.push_back(0); rows.a.push_back(0.0); }
      Aext = exchange_rows(pkg, rows, true, true);
   }
   {
      ExtCSR rows;
      device_extract_S_rows(tot, d_elmts, n, Sx->i, Sx->j, (long long) A->first_col_diag, d_cmap, dCFx, rows.i, rows.j, st);
      if (rows.j.empty()) { rows.j.push_back(0); }
      Sop = exchange_rows(pkg, rows, true, false);
   }
   dfree(d_cmap); dfree(dCFx);
   std::vector<HYPRE_BigInt> found;
   ghost_row_numbering(A->col_map_offd, nco, col_1, col_n, CF_offd, Aext, Sop, found);
   const HYPRE_Int newoff = (HYPRE_Int) found.size(), full_off = nco + newoff;
   hypre_ParCSRCommPkg ext_pkg;
   memset(&ext_pkg, 0, sizeof(ext_pkg));
   ext_pkg.comm = comm;
   hypre_ParCSRCommPkgCreate_core(comm, found.empty() ? nullptr : found.data(), A->first_col_diag, A->col_starts,
                                  A->diag->num_cols, newoff, &ext_pkg.num_recvs, &ext_pkg.recv_procs,
                                  &ext_pkg.recv_vec_starts, &ext_pkg.num_sends, &ext_pkg.send_procs,
                                  &ext_pkg.send_map_starts, &ext_pkg.send_map_elmts);
   CF_offd.resize((size_t) std::max(full_off, 1));
   halo_fwd<HYPRE_Int>(&ext_pkg, CF_marker, CF_offd.data() + nco, 11);

   // ---- coarse numbers: local ones on the device; the ghosts' global ones through the two packages
   int *dF2Cx = nullptr;                                          // over [local | ghosts | new nodes]
   HIP_CHECK(hipMalloc((void **) &dF2Cx, sizeof(int) * (size_t) std::max(n + full_off, 1)));
   device_coarse_numbering(n, dCF, dF2Cx, st);
   std::vector<HYPRE_BigInt> f2c_offd((size_t) std::max(full_off, 1), -1);
   auto coarse_ids_out = [&](hypre_ParCSRCommPkg *pk, const int *d_idx, HYPRE_BigInt *ghost)
   {
      const HYPRE_Int t = pk->send_map_starts[pk->num_sends];
      std::vector<HYPRE_Int> loc((size_t) std::max(t, 1));
      std::vector<HYPRE_BigInt> buf((size_t) std::max(t, 1));
      if (t)
      {
         int *d_out = nullptr;
         HIP_CHECK(hipMalloc((void **) &d_out, sizeof(int) * (size_t) t));
         launch_gather_int(dF2Cx, d_idx, d_out, (size_t) t, st);
         fetch(loc.data(), d_out, (size_t) t);
         dfree(d_out);
      }
      for (HYPRE_Int k = 0; k < t; k++) { buf[(size_t) k] = (HYPRE_BigInt) loc[(size_t) k] + my_first_cpt; }
      hypre_ParCSRCommHandle *h = hypre_ParCSRCommHandleCreate(21, pk, buf.data(), ghost);
      hypre_ParCSRCommHandleDestroy(h);
   };
   coarse_ids_out(pkg, d_elmts, f2c_offd.data());
   {
      const HYPRE_Int t = ext_pkg.send_map_starts[ext_pkg.num_sends];
      int *d_idx = upload(ext_pkg.send_map_elmts, (size_t) t);
      coarse_ids_out(&ext_pkg, d_idx, f2c_offd.data() + nco);
      dfree(d_idx);
   }

   // ---- extended matrices: the ghost F points' rows behind the local ones.  A fetched row of A holds only the entries the
   // weights can use (sign opposite to its diagonal: the owner filtered them) and no diagonal: a stand-in of the right
   // sign goes in front, where the kernel expects a row's diagonal
   std::vector<HYPRE_Int> gi((size_t) full_off + 1, 0), gj, si((size_t) full_off + 1, 0), sj;
   std::vector<HYPRE_Real> ga;
   auto ext_id = [&](HYPRE_BigInt g) { return g >= 0 ? (HYPRE_Int) (g - col_1) : n + (HYPRE_Int) (-g - 1); };
   for (HYPRE_Int k = 0; k < nco; k++)
   {
      if (CF_offd[(size_t) k] < 0)
      {
         const HYPRE_Int b = Aext.i[(size_t) k], e = Aext.i[(size_t) k + 1];
         gj.push_back(n + k);
         ga.push_back(b < e && Aext.a[(size_t) b] > 0 ? -1.0 : 1.0);
         for (HYPRE_Int q = b; q < e; q++) { gj.push_back(ext_id(Aext.j[(size_t) q])); ga.push_back(Aext.a[(size_t) q]); }
         for (HYPRE_Int q = Sop.i[(size_t) k]; q < Sop.i[(size_t) k + 1]; q++) { sj.push_back(ext_id(Sop.j[(size_t) q])); }
      }
      gi[(size_t) k + 1] = (HYPRE_Int) gj.size();
      si[(size_t) k + 1] = (HYPRE_Int) sj.size();
   }
   for (HYPRE_Int k = nco; k < full_off; k++) { gi[(size_t) k + 1] = gi[(size_t) k]; si[(size_t) k + 1] = si[(size_t) k]; }
   bool ok = true;
   int *Pdi = nullptr, *Pdj = nullptr, *Poi = nullptr, *Poj = nullptr, dnnz = 0, onnz = 0;
   double *Pda = nullptr, *Poa = nullptr;
   {
      DevCSR G, GS;
      G.put(gi, gj, &ga);
      GS.put(si, sj, nullptr);
      int *Ei = nullptr, *Ej = nullptr, *SEi = nullptr, *SEj = nullptr, ennz = 0, sennz = 0;
      double *Ea = nullptr;
      device_extend_csr(n, dD->i, dD->j, dD->data, nco ? dO->i : nullptr, dO->j, dO->data, n, full_off, G.i, G.j, G.a, &Ei, &Ej, &Ea, &ennz, st);
      device_extend_csr(n, Sx->i, Sx->j, nullptr, nullptr, nullptr, nullptr, 0, full_off, GS.i, GS.j, nullptr, &SEi, &SEj, nullptr, &sennz, st);
      // markers and coarse numbers of the ghost points: an off-rank C point's column is ncP + its ghost number
      int *dCFf = nullptr;
      HIP_CHECK(hipMalloc((void **) &dCFf, sizeof(int) * (size_t) std::max(n + full_off, 1)));
      std::vector<int> f2c_g((size_t) std::max(full_off, 1));
      for (HYPRE_Int k = 0; k < full_off; k++) { f2c_g[(size_t) k] = CF_offd[(size_t) k] >= 0 ? ncP + k : -1; }
      if (n) { HIP_CHECK(hipMemcpyAsync(dCFf, dCF, sizeof(int) * (size_t) n, hipMemcpyDeviceToDevice, st)); }
      if (full_off)
      {
         HIP_CHECK(hipMemcpyAsync(dCFf + n, CF_offd.data(), sizeof(int) * (size_t) full_off, hipMemcpyHostToDevice, st));
         HIP_CHECK(hipMemcpyAsync(dF2Cx + n, f2c_g.data(), sizeof(int) * (size_t) full_off, hipMemcpyHostToDevice, st));
      }
      HIP_CHECK(hipStreamSynchronize(st));
      ok = device_extpi(n, Ei, Ej, Ea, SEi, SEj, dCFf, dF2Cx, trunc_factor, max_elmts, first_rung, &Pdi, &Pdj, &Pda, &dnnz, st,
                        ncP, &Poi, &Poj, &Poa, &onnz);
      dfree(Ei); dfree(Ej); dfree(Ea); dfree(SEi); dfree(SEj); dfree(dCFf);
   }
   dfree(dF2Cx);
   auto free_ext_pkg = [&]()
   {
      hypre_Free(ext_pkg.recv_procs, HYPRE_MEMORY_HOST); hypre_Free(ext_pkg.recv_vec_starts, HYPRE_MEMORY_HOST);
      hypre_Free(ext_pkg.send_procs, HYPRE_MEMORY_HOST); hypre_Free(ext_pkg.send_map_starts, HYPRE_MEMORY_HOST);
      hypre_Free(ext_pkg.send_map_elmts, HYPRE_MEMORY_HOST);
   };
   if (ok && test_declines(comm, 1)) { dfree(Pdi); dfree(Pdj); dfree(Pda); dfree(Poi); dfree(Poj); dfree(Poa); ok = false; }
   if (!all_ranks_agree(comm, ok))
   {
      if (ok) { dfree(Pdi); dfree(Pdj); dfree(Pda); dfree(Poi); dfree(Poj); dfree(Poa); }
      free_ext_pkg();
      return hypre_error_flag;
   }

   // ---- column map of the off-rank block (aux_interp.c:777-900): the ghosts in use, ascending coarse number
   std::vector<HYPRE_BigInt> cmap;
   {
      int *d_used = nullptr;
      HIP_CHECK(hipMalloc((void **) &d_used, sizeof(int) * (size_t) std::max(full_off, 1)));
      HIP_CHECK(hipMemsetAsync(d_used, 0, sizeof(int) * (size_t) std::max(full_off, 1), st));
      launch_mark_used(Poj, (size_t) onnz, d_used, st);
      std::vector<int> used((size_t) std::max(full_off, 1), 0);
      fetch(used.data(), d_used, (size_t) full_off);
      for (HYPRE_Int g = 0; g < full_off; g++) { if (used[(size_t) g]) { cmap.push_back(f2c_offd[(size_t) g]); } }
      std::sort(cmap.begin(), cmap.end());
      cmap.erase(std::unique(cmap.begin(), cmap.end()), cmap.end());
      std::vector<int> renum((size_t) std::max(full_off, 1), 0);
      for (HYPRE_Int g = 0; g < full_off; g++)
      {
         if (used[(size_t) g]) { renum[(size_t) g] = (int) (std::lower_bound(cmap.begin(), cmap.end(), f2c_offd[(size_t) g]) - cmap.begin()); }
      }
      HIP_CHECK(hipMemcpyAsync(d_used, renum.data(), sizeof(int) * (size_t) full_off, hipMemcpyHostToDevice, st));
      launch_renumber(Poj, (size_t) onnz, 0, d_used, st);
      HIP_CHECK(hipStreamSynchronize(st));
      dfree(d_used);
   }
   HYPRE_BigInt cs[2] = {num_cpts_global[0], num_cpts_global[1]};
   hypre_ParCSRMatrix *P = hypre_ParCSRMatrixCreate(comm, A->global_num_rows, total_global_cpts, A->col_starts, cs,
                                                    (HYPRE_Int) cmap.size(), dnnz, onnz);
   hypre_CSRMatrixDestroy(P->diag); hypre_CSRMatrixDestroy(P->offd);
   P->diag = setup_wrap_device_csr(n, ncP, dnnz, Pdi, Pdj, Pda);
   P->offd = setup_wrap_device_csr(n, (HYPRE_Int) cmap.size(), onnz, Poi, Poj, Poa);
   if (!cmap.empty())
   {
      P->col_map_offd = hypre_TAlloc(HYPRE_BigInt, cmap.size(), HYPRE_MEMORY_HOST);
      memcpy(P->col_map_offd, cmap.data(), sizeof(HYPRE_BigInt) * cmap.size());
   }
   hypre_CSRMatrixSetRownnz(P->offd);
   hypre_MatvecCommPkgCreate(P);
   free_ext_pkg();
   *P_ptr = P;
   return hypre_error_flag;
}

HYPRE_Int dist_device_coarse_operator(hypre_ParCSRMatrix *RT, hypre_ParCSRMatrix *A, hypre_ParCSRMatrix *P,
                                      HYPRE_Int keepTranspose, hypre_ParCSRMatrix **RAP_ptr)
{
   *RAP_ptr = nullptr;
   MPI_Comm comm = A->comm;
   hipStream_t st = stream();
   if (!A->comm_pkg) { hypre_MatvecCommPkgCreate(A); }
   if (!RT->comm_pkg) { hypre_MatvecCommPkgCreate(RT); }
   hypre_ParCSRCommPkg *pkgA = A->comm_pkg, *pkgRT = RT->comm_pkg;
   hypre_CSRMatrix *dD = setup_device_twin_of(A->diag, 1), *dO = setup_device_twin_of(A->offd, 1);
   hypre_CSRMatrix *Pd = setup_device_twin_of(P->diag, 1), *Po = setup_device_twin_of(P->offd, 1);
   const HYPRE_Int n = A->diag->num_rows, ncoA = A->offd->num_cols;
   const HYPRE_Int ncP = P->diag->num_cols, ncoP = P->offd->num_cols, ncoRT = RT->offd->num_cols, ncRT = RT->diag->num_cols;
   const HYPRE_BigInt first_c = P->first_col_diag, last_c = first_c + ncP - 1;
   hypre_CSRMatrix *Rd = nullptr, *Ro = nullptr;
   hypre_CSRMatrixTranspose(Pd, &Rd, 1);
   if (ncoRT) { hypre_CSRMatrixTranspose(Po, &Ro, 1); }

   // ---- P_ext: the rows of P of A's ghost columns, local coarse columns first, off-rank ones behind
   ExtCSR Ps;
   {
      const HYPRE_Int totA = pkgA->send_map_starts[pkgA->num_sends];
      long long *d_cmapP = upload((const long long *) P->col_map_offd, (size_t) ncoP);
      ExtCSR rows;
      device_extract_rows(0, totA, send_elmts_on_device(pkgA), Pd->i, Pd->j, Pd->data, ncoP ? Po->i : nullptr, Po->j, Po->data,
                          (long long) first_c, d_cmapP, nullptr, rows.i, rows.j, rows.a, st);
      dfree(d_cmapP);
      if (rows.j.empty()) { rows.j.push_back(0); rows.a.push_back(0.0); }
      Ps = exchange_rows(pkgA, rows, true, true);
   }
   std::vector<HYPRE_Int> Pedi((size_t) ncoA + 1, 0), Peoi((size_t) ncoA + 1, 0), Pedj, Peoj;
   std::vector<HYPRE_Real> Peda, Peoa;
   std::vector<HYPRE_BigInt> Pe_big;
   for (HYPRE_Int i = 0; i < ncoA; i++)
   {
      for (HYPRE_Int k = Ps.i[(size_t) i]; k < Ps.i[(size_t) i + 1]; k++)
      {
         const HYPRE_BigInt g = Ps.j[(size_t) k];
         if (g < first_c || g > last_c) { Pe_big.push_back(g); Peoa.push_back(Ps.a[(size_t) k]); }
         else { Pedj.push_back((HYPRE_Int) (g - first_c)); Peda.push_back(Ps.a[(size_t) k]); }
      }
      Pedi[(size_t) i + 1] = (HYPRE_Int) Pedj.size();
      Peoi[(size_t) i + 1] = (HYPRE_Int) Pe_big.size();
   }
   std::vector<HYPRE_BigInt> cmapPext(Pe_big);
   for (HYPRE_Int k = 0; k < ncoP; k++) { cmapPext.push_back(P->col_map_offd[k]); }
   std::sort(cmapPext.begin(), cmapPext.end());
   cmapPext.erase(std::unique(cmapPext.begin(), cmapPext.end()), cmapPext.end());
   const HYPRE_Int ncoPext = (HYPRE_Int) cmapPext.size();
   Peoj.resize(Pe_big.size());
   for (size_t k = 0; k < Pe_big.size(); k++) { Peoj[k] = bsearch_big(cmapPext.data(), Pe_big[k], ncoPext); }
   std::vector<int> mapP2Pext((size_t) std::max(ncoP, 1));
   for (HYPRE_Int k = 0; k < ncoP; k++) { mapP2Pext[(size_t) k] = bsearch_big(cmapPext.data(), P->col_map_offd[k], ncoPext); }

   // ---- the extended interpolation operator: rows of the local points [P_diag | ncP + P_offd in P_ext's numbering], then
   // the ghost points' rows
   int *Xi = nullptr, *Xj = nullptr, xnnz = 0;
   double *Xa = nullptr;
   {
      std::vector<HYPRE_Int> gi((size_t) ncoA + 1, 0), gj;
      std::vector<HYPRE_Real> ga;
      for (HYPRE_Int i = 0; i < ncoA; i++)
      {
         for (HYPRE_Int k = Pedi[(size_t) i]; k < Pedi[(size_t) i + 1]; k++) { gj.push_back(Pedj[(size_t) k]); ga.push_back(Peda[(size_t) k]); }
         for (HYPRE_Int k = Peoi[(size_t) i]; k < Peoi[(size_t) i + 1]; k++) { gj.push_back(ncP + Peoj[(size_t) k]); ga.push_back(Peoa[(size_t) k]); }
         gi[(size_t) i + 1] = (HYPRE_Int) gj.size();
      }
      DevCSR G;
      G.put(gi, gj, &ga);
      int *d_map = upload(mapP2Pext.data(), (size_t) ncoP), *oj = nullptr;
      const HYPRE_Int onz = Po->num_nonzeros;
      HIP_CHECK(hipMalloc((void **) &oj, sizeof(int) * (size_t) std::max(onz, 1)));
      if (onz) { HIP_CHECK(hipMemcpyAsync(oj, Po->j, sizeof(int) * (size_t) onz, hipMemcpyDeviceToDevice, st)); }
      launch_renumber(oj, (size_t) onz, 0, d_map, st);
      device_extend_csr(n, Pd->i, Pd->j, Pd->data, ncoP ? Po->i : nullptr, oj, Po->data, ncP, ncoA, G.i, G.j, G.a, &Xi, &Xj, &Xa, &xnnz, st);
      dfree(d_map); dfree(oj);
   }
   const int maxP = device_max_row_nnz(Xi, n + ncoA, st);
   auto give_up = [&]()
   {
      dfree(Xi); dfree(Xj); dfree(Xa);
      hypre_CSRMatrixDestroy(Rd);
      if (Ro) { hypre_CSRMatrixDestroy(Ro); }
      return hypre_error_flag;
   };

   // ---- the rows of the product that belong to the neighbours (one per ghost coarse column of P), sent home
   ExtCSR Rint;
   {
      int *Ii = nullptr, *Ij = nullptr, innz = 0;
      double *Ia = nullptr;
      const bool ok = device_rap_dist(true, ncoRT, 0, ncP + ncoPext, maxP, 0, Ro ? Ro->i : nullptr, Ro ? Ro->j : nullptr, Ro ? Ro->data : nullptr,
                                      dD->i, dD->j, dD->data, ncoA ? dO->i : nullptr, dO->j, dO->data, n, n, Xi, Xj, Xa,
                                      nullptr, nullptr, nullptr, nullptr, nullptr, ncP, &Ii, &Ij, &Ia, &innz, nullptr, nullptr, nullptr, nullptr, st);
      bool okk = ok;
      if (okk && test_declines(comm, 2)) { dfree(Ii); dfree(Ij); dfree(Ia); okk = false; }
      if (!all_ranks_agree(comm, okk))
      {
         if (okk) { dfree(Ii); dfree(Ij); dfree(Ia); }
         return give_up();
      }
      Rint.i.assign((size_t) ncoRT + 1, 0);
      std::vector<int> hj((size_t) std::max(innz, 1));
      Rint.a.resize((size_t) std::max(innz, 1));
      fetch(Rint.i.data(), Ii, (size_t) ncoRT + 1);
      fetch(hj.data(), Ij, (size_t) innz);
      fetch(Rint.a.data(), Ia, (size_t) innz);
      dfree(Ii); dfree(Ij); dfree(Ia);
      Rint.j.resize((size_t) std::max(innz, 1), 0);
      for (HYPRE_Int k = 0; k < innz; k++)
      {
         const int c = hj[(size_t) k];
         Rint.j[(size_t) k] = c < ncP ? (HYPRE_BigInt) c + first_c : cmapPext[(size_t) (c - ncP)];
      }
      if (innz == 0) { Rint.j.assign(1, 0); Rint.a.assign(1, 0.0); }
      else { Rint.j.resize((size_t) innz); Rint.a.resize((size_t) innz); }
   }
   ExtCSR Rext = exchange_rows(pkgRT, Rint, false, true);
   const HYPRE_Int nsendRT = pkgRT->send_map_starts[pkgRT->num_sends];

   // ---- column map of the off-rank block of the product
   std::vector<HYPRE_BigInt> cmapRAP;
   for (size_t k = 0; k < Rext.j.size(); k++) { if (Rext.j[k] < first_c || Rext.j[k] > last_c) { cmapRAP.push_back(Rext.j[k]); } }
   for (HYPRE_Int k = 0; k < ncoPext; k++) { cmapRAP.push_back(cmapPext[(size_t) k]); }
   std::sort(cmapRAP.begin(), cmapRAP.end());
   cmapRAP.erase(std::unique(cmapRAP.begin(), cmapRAP.end()), cmapRAP.end());
   const HYPRE_Int ncoRAP = (HYPRE_Int) cmapRAP.size();
   std::vector<int> mapPext2RAP((size_t) std::max(ncoPext, 1));
   for (HYPRE_Int k = 0; k < ncoPext; k++) { mapPext2RAP[(size_t) k] = bsearch_big(cmapRAP.data(), cmapPext[(size_t) k], ncoRAP); }
   // what the neighbours computed for this rank's rows: the received rows in the extended coarse numbering, and per local
   // coarse row the received rows that feed it, in package order
   std::vector<HYPRE_Int> xj(Rext.j.size());
   for (size_t k = 0; k < Rext.j.size(); k++)
   {
      const HYPRE_BigInt g = Rext.j[k];
      xj[k] = (g < first_c || g > last_c) ? ncP + bsearch_big(cmapRAP.data(), g, ncoRAP) : (HYPRE_Int) (g - first_c);
   }
   std::vector<HYPRE_Int> Fi((size_t) ncRT + 1, 0), Fj((size_t) std::max(nsendRT, 1));
   for (HYPRE_Int s = 0; s < nsendRT; s++) { Fi[(size_t) pkgRT->send_map_elmts[s] + 1]++; }
   for (HYPRE_Int ic = 0; ic < ncRT; ic++) { Fi[(size_t) ic + 1] += Fi[(size_t) ic]; }
   int max_seed = 0;
   {
      std::vector<HYPRE_Int> pos(Fi.begin(), Fi.end() - 1), seed((size_t) std::max(ncRT, 1), 0);
      for (HYPRE_Int s = 0; s < nsendRT; s++)
      {
         const HYPRE_Int ic = pkgRT->send_map_elmts[s];
         Fj[(size_t) pos[(size_t) ic]++] = s;
         seed[(size_t) ic] += Rext.i[(size_t) s + 1] - Rext.i[(size_t) s];
         max_seed = std::max(max_seed, (int) seed[(size_t) ic]);
      }
   }
   int *Cdi = nullptr, *Cdj = nullptr, *Coi = nullptr, *Coj = nullptr, cdnnz = 0, connz = 0;
   double *Cda = nullptr, *Coa = nullptr;
   bool ok;
   {
      DevCSR X;
      X.put(Rext.i, xj, &Rext.a);
      int *dFi = upload(Fi.data(), Fi.size()), *dFj = upload(Fj.data(), (size_t) nsendRT);
      int *d_map = upload(mapPext2RAP.data(), (size_t) ncoPext);
      launch_renumber(Xj, (size_t) xnnz, ncP, d_map, st);            // P_ext's numbering of the ghost coarse points -> the product's
      ok = device_rap_dist(false, ncRT, ncRT == ncP ? 1 : 0, ncP + ncoRAP, maxP, max_seed, Rd->i, Rd->j, Rd->data,
                           dD->i, dD->j, dD->data, ncoA ? dO->i : nullptr, dO->j, dO->data, n, n, Xi, Xj, Xa,
                           nsendRT ? dFi : nullptr, dFj, X.i, X.j, X.a, ncP, &Cdi, &Cdj, &Cda, &cdnnz, &Coi, &Coj, &Coa, &connz, st);
      dfree(dFi); dfree(dFj); dfree(d_map);
   }
   if (ok && test_declines(comm, 3)) { dfree(Cdi); dfree(Cdj); dfree(Cda); dfree(Coi); dfree(Coj); dfree(Coa); ok = false; }
   if (!all_ranks_agree(comm, ok))
   {
      if (ok) { dfree(Cdi); dfree(Cdj); dfree(Cda); dfree(Coi); dfree(Coj); dfree(Coa); }
      return give_up();
   }
   dfree(Xi); dfree(Xj); dfree(Xa);
   hypre_ParCSRMatrix *C = hypre_ParCSRMatrixCreate(comm, RT->global_num_cols, P->global_num_cols, RT->col_starts,
                                                    P->col_starts, ncoRAP, cdnnz, connz);
   hypre_CSRMatrixDestroy(C->diag); hypre_CSRMatrixDestroy(C->offd);
   C->diag = setup_wrap_device_csr(ncRT, ncP, cdnnz, Cdi, Cdj, Cda);
   C->offd = setup_wrap_device_csr(ncRT, ncoRAP, connz, Coi, Coj, Coa);
   if (ncoRAP)
   {
      C->col_map_offd = hypre_TAlloc(HYPRE_BigInt, ncoRAP, HYPRE_MEMORY_HOST);
      memcpy(C->col_map_offd, cmapRAP.data(), sizeof(HYPRE_BigInt) * (size_t) ncoRAP);
   }
   if (keepTranspose) { RT->diagT = Rd; RT->offdT = Ro; }
   else { hypre_CSRMatrixDestroy(Rd); if (Ro) { hypre_CSRMatrixDestroy(Ro); } }
   hypre_CSRMatrixSetRownnz(C->offd);
   hypre_ParCSRMatrixSetNumNonzeros(C);
   hypre_ParCSRMatrixSetDNumNonzeros(C);
   hypre_MatvecCommPkgCreate(C);
   *RAP_ptr = C;
   return hypre_error_flag;
}

}  // namespace hamd
