// hypre_amd — IJ text files in and out (the reference's regression inputs).
//
// Reference: IJ_mv/IJMatrix.c:112-247 (hypre_IJMatrixRead), IJ_mv/IJMatrix_parcsr.c:2690-2950
// (assembly of the auxiliary rows into diag / offd), IJ_mv/HYPRE_IJMatrix.c:21-117 (global first
// row / column and sizes), parcsr_mv/par_csr_matrix.c:888-1047 (PrintIJ), IJ_mv/HYPRE_IJVector.c:
// 641-782 (vector read / print).  Host code on both sides of the solve path: a matrix read here is
// an ordinary host ParCSR matrix that setup consumes and hypre_ParCSRMatrixMigrate moves to the GPU.
#include "internal.hpp"
#include "hypre_amd_IJ_mv.h"
#include <algorithm>
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

using namespace hamd;

namespace {

// whole file in memory; number parsing with strtoll / strtod walks it once
bool slurp(const std::string &path, std::string &out)
{
   FILE *f = fopen(path.c_str(), "rb");
   if (!f) { return false; }
   fseek(f, 0, SEEK_END);
   const long n = ftell(f);
   fseek(f, 0, SEEK_SET);
   out.resize((size_t) std::max(0L, n));
   const size_t got = n > 0 ? fread(&out[0], 1, (size_t) n, f) : 0;
   fclose(f);
   return got == (size_t) std::max(0L, n);
}

struct Cursor
{
   const char *p, *end;
   void skip_ws() { while (p < end && isspace((unsigned char) *p)) { p++; } }
   bool at_end() { skip_ws(); return p >= end; }
   bool big(HYPRE_BigInt &v)
   {
      skip_ws();
      if (p >= end) { return false; }
      char *q = nullptr;
      const long long x = strtoll(p, &q, 10);
      if (q == p) { return false; }
      p = q; v = (HYPRE_BigInt) x;
      return true;
   }
   // the reference's "%*[ \t]%le": at least one blank or tab must separate an index from the value,
   // so that a line holding a lone decimal number is not read as "index value" (HYPRE_IJVector.c:666)
   bool blank_then_real(double &v)
   {
      if (p >= end || (*p != ' ' && *p != '\t')) { return false; }
      while (p < end && (*p == ' ' || *p == '\t')) { p++; }
      char *q = nullptr;
      const double x = strtod(p, &q);
      if (q == p) { return false; }
      p = q; v = x;
      return true;
   }
};

std::string rank_file(const char *filename, int rank)
{
   char buf[32];
   snprintf(buf, sizeof buf, ".%05d", rank);
   return std::string(filename) + buf;
}

// first index of rank 0 and last index of the last rank (HYPRE_IJMatrix.c:88-113: two broadcasts)
void global_span(MPI_Comm comm, HYPRE_BigInt lower, HYPRE_BigInt upper, HYPRE_BigInt &first, HYPRE_BigInt &count)
{
   HYPRE_Int np = 1;
   hypre_MPI_Comm_size(comm, &np);
   if (np <= 1) { first = lower; count = upper - lower + 1; return; }
   const hypre_amd_CommOps *o = comm_ops(comm);
   HYPRE_BigInt mine[2] = {lower, upper};
   std::vector<HYPRE_BigInt> all((size_t) 2 * np);
   o->allgather(o->ctx, mine, all.data(), sizeof mine);
   first = all[0];
   count = all[(size_t) 2 * (np - 1) + 1] - first + 1;
}

struct OffProcEntry { HYPRE_BigInt I, J; double v; };

// entries for rows of other ranks, from every rank, in rank order (the reference ships them to the
// owners through hypre_IJMatrixAssembleOffProcValsParCSR, IJMatrix_parcsr.c:1770-2530; the files
// that need this are a few dozen lines, so everybody gets everything)
std::vector<OffProcEntry> gather_off_proc(MPI_Comm comm, const std::vector<OffProcEntry> &mine)
{
   HYPRE_Int np = 1;
   hypre_MPI_Comm_size(comm, &np);
   if (np <= 1) { return mine; }
   const hypre_amd_CommOps *o = comm_ops(comm);
   int n = (int) mine.size();
   std::vector<int> counts((size_t) np, 0);
   o->allgather(o->ctx, &n, counts.data(), sizeof(int));
   const int maxn = std::max(1, *std::max_element(counts.begin(), counts.end()));
   std::vector<OffProcEntry> pad((size_t) maxn, OffProcEntry{0, 0, 0.0}), all((size_t) maxn * (size_t) np);
   std::copy(mine.begin(), mine.end(), pad.begin());
   o->allgather(o->ctx, pad.data(), all.data(), sizeof(OffProcEntry) * (size_t) maxn);
   std::vector<OffProcEntry> out;
   for (int r = 0; r < np; r++)
   {
      out.insert(out.end(), all.begin() + (size_t) r * (size_t) maxn, all.begin() + (size_t) r * (size_t) maxn + counts[(size_t) r]);
   }
   return out;
}

// every rank learns whether any rank failed, so that a collective read fails everywhere
bool any_rank(MPI_Comm comm, bool mine)
{
   HYPRE_Int np = 1;
   hypre_MPI_Comm_size(comm, &np);
   if (np <= 1) { return mine; }
   const hypre_amd_CommOps *o = comm_ops(comm);
   int flag = mine ? 1 : 0;
   std::vector<int> all((size_t) np);
   o->allgather(o->ctx, &flag, all.data(), sizeof(int));
   for (int f : all) { if (f) { return true; } }
   return false;
}

}  // namespace

extern "C" {

HYPRE_Int HYPRE_IJMatrixRead(const char *filename, MPI_Comm comm, HYPRE_Int type, HYPRE_IJMatrix *matrix_ptr)
{
   if (matrix_ptr) { *matrix_ptr = nullptr; }
   if (!filename || !matrix_ptr || !comm_ops(comm)) { hypre_error_in_arg(1); return hypre_error_flag; }
   if (type != HYPRE_PARCSR) { hypre_error_in_arg(3); return hypre_error_flag; }
   HYPRE_Int rank = 0;
   hypre_MPI_Comm_rank(comm, &rank);

   std::string text;
   bool bad_open = !slurp(rank_file(filename, rank), text);
   Cursor c{text.data(), text.data() + text.size()};
   HYPRE_BigInt ilower = 0, iupper = -1, jlower = 0, jupper = -1;
   bool bad_format = false;
   if (!bad_open)
   {
      bad_format = !(c.big(ilower) && c.big(iupper) && c.big(jlower) && c.big(jupper));
      // HYPRE_IJMatrixCreate's argument checks (HYPRE_IJMatrix.c:52-78)
      if (!bad_format && (ilower > iupper + 1 || ilower < 0 || iupper < -1 || jlower > jupper + 1 || jlower < 0 || jupper < -1))
      {
         bad_format = true;
      }
   }
   if (any_rank(comm, bad_open)) { hypre_error_in_arg(1); return hypre_error_flag; }

   const HYPRE_Int nrows = bad_format ? 0 : (HYPRE_Int) (iupper - ilower + 1);
   // rows as the auxiliary matrix holds them: entries in arrival order, a repeated column overwrites
   std::vector<std::vector<HYPRE_BigInt>> rj((size_t) nrows);
   std::vector<std::vector<double>>       ra((size_t) nrows);
   std::vector<OffProcEntry>              off_proc;
   while (!bad_format && !c.at_end())
   {
      HYPRE_BigInt I, J; double v;
      if (!(c.big(I) && c.big(J) && c.blank_then_real(v))) { bad_format = true; break; }
      if (I < ilower || I > iupper)
      {
         off_proc.push_back(OffProcEntry{I, J, v});     // AddToValues on the owner, at assembly (IJMatrix.c:214-217)
         continue;
      }
      std::vector<HYPRE_BigInt> &cols = rj[(size_t) (I - ilower)];
      std::vector<double>       &vals = ra[(size_t) (I - ilower)];
      size_t k = 0;
      while (k < cols.size() && cols[k] != J) { k++; }
      if (k < cols.size()) { vals[k] = v; }
      else { cols.push_back(J); vals.push_back(v); }
   }
   if (any_rank(comm, bad_format))
   {
      hypre_error_w_msg(HYPRE_ERROR_GENERIC, "Error in IJ matrix input file.");
      return hypre_error_flag;
   }

   // contributions to rows of other ranks are added on the owner (an entry that is not there yet is
   // appended), after the owner's own entries
   for (const OffProcEntry &e : gather_off_proc(comm, off_proc))
   {
      if (e.I < ilower || e.I > iupper) { continue; }
      std::vector<HYPRE_BigInt> &cols = rj[(size_t) (e.I - ilower)];
      std::vector<double>       &vals = ra[(size_t) (e.I - ilower)];
      size_t k = 0;
      while (k < cols.size() && cols[k] != e.J) { k++; }
      if (k < cols.size()) { vals[k] += e.v; }
      else { cols.push_back(e.J); vals.push_back(e.v); }
   }

   hypre_IJMatrix *ij = (hypre_IJMatrix *) calloc(1, sizeof(hypre_IJMatrix));
   ij->comm = comm;
   ij->row_partitioning[0] = ilower; ij->row_partitioning[1] = iupper + 1;
   ij->col_partitioning[0] = jlower; ij->col_partitioning[1] = jupper + 1;
   ij->object_type = HYPRE_PARCSR;
   global_span(comm, ilower, iupper, ij->global_first_row, ij->global_num_rows);
   global_span(comm, jlower, jupper, ij->global_first_col, ij->global_num_cols);

   // split into diag / offd (IJMatrix_parcsr.c:2724-2821)
   const HYPRE_BigInt col_0 = jlower, col_n = jupper, base = ij->global_first_col;
   std::vector<HYPRE_Int> di((size_t) nrows + 1, 0), oi((size_t) nrows + 1, 0);
   std::vector<HYPRE_Int> dj; std::vector<double> da, oa; std::vector<HYPRE_BigInt> obig;
   for (HYPRE_Int i = 0; i < nrows; i++)
   {
      const std::vector<HYPRE_BigInt> &cols = rj[(size_t) i];
      const std::vector<double>       &vals = ra[(size_t) i];
      int diag_pos = -1;
      for (size_t k = 0; k < cols.size(); k++)
      {
         if (cols[k] >= col_0 && cols[k] <= col_n && (HYPRE_Int) (cols[k] - col_0) == i) { diag_pos = (int) k; }
      }
      if (diag_pos >= 0) { dj.push_back(i); da.push_back(vals[(size_t) diag_pos]); }
      for (size_t k = 0; k < cols.size(); k++)
      {
         if (cols[k] < col_0 || cols[k] > col_n) { obig.push_back(cols[k]); oa.push_back(vals[k]); }
         else if ((int) k != diag_pos) { dj.push_back((HYPRE_Int) (cols[k] - col_0)); da.push_back(vals[k]); }
      }
      di[(size_t) i + 1] = (HYPRE_Int) dj.size();
      oi[(size_t) i + 1] = (HYPRE_Int) obig.size();
   }
   // ghost columns: sorted, unique, relative to the global first column (:2906-2944)
   std::vector<HYPRE_BigInt> cmap(obig);
   std::sort(cmap.begin(), cmap.end());
   cmap.erase(std::unique(cmap.begin(), cmap.end()), cmap.end());
   std::vector<HYPRE_Int> oj(obig.size());
   for (size_t k = 0; k < obig.size(); k++)
   {
      oj[k] = (HYPRE_Int) (std::lower_bound(cmap.begin(), cmap.end(), obig[k]) - cmap.begin());
   }
   for (HYPRE_BigInt &g : cmap) { g -= base; }

   HYPRE_BigInt rs[2] = {ilower - ij->global_first_row, iupper + 1 - ij->global_first_row};
   HYPRE_BigInt cs[2] = {jlower - ij->global_first_col, jupper + 1 - ij->global_first_col};
   hypre_ParCSRMatrix *A = hypre_amd_ParCSRMatrixFromArrays(comm, ij->global_num_rows, ij->global_num_cols, rs, cs,
                                                            (HYPRE_Int) cmap.size(), cmap.data(), di.data(), dj.data(),
                                                            da.data(), oi.data(), oj.data(), oa.data(),
                                                            HYPRE_MEMORY_HOST);
   ij->object = A;
   ij->assemble_flag = 1;
   *matrix_ptr = ij;
   return hypre_error_flag;
}

HYPRE_Int hypre_ParCSRMatrixPrintIJ(const hypre_ParCSRMatrix *matrix, const HYPRE_Int base_i, const HYPRE_Int base_j,
                                    const char *filename)
{
   if (!matrix) { hypre_error_in_arg(1); return hypre_error_flag; }
   hypre_ParCSRMatrix *h = (hypre_ParCSRMatrix *) matrix;
   const bool on_device = matrix->diag->memory_location != HYPRE_MEMORY_HOST;
   if (on_device) { h = hypre_ParCSRMatrixClone_v2((hypre_ParCSRMatrix *) matrix, 1, HYPRE_MEMORY_HOST); }
   HYPRE_Int rank = 0;
   hypre_MPI_Comm_rank(h->comm, &rank);
   FILE *f = fopen(rank_file(filename, rank).c_str(), "w");
   if (!f)
   {
      hypre_error_w_msg(HYPRE_ERROR_GENERIC, "Error: can't open output file %s\n");
      if (on_device) { hypre_ParCSRMatrixDestroy(h); }
      return hypre_error_flag;
   }
   const hypre_CSRMatrix *diag = h->diag, *offd = h->offd;
   const HYPRE_Int nrows = diag->num_rows;
   fprintf(f, "%lld %lld %lld %lld\n", (long long) (h->row_starts[0] + base_i), (long long) (h->row_starts[1] + base_i - 1),
           (long long) (h->col_starts[0] + base_j), (long long) (h->col_starts[1] + base_j - 1));
   const bool has_offd = offd && offd->num_nonzeros > 0;
   for (HYPRE_Int i = 0; i < nrows; i++)
   {
      const long long I = (long long) (h->first_row_index + i + base_i);
      for (HYPRE_Int k = diag->i[i]; k < diag->i[i + 1]; k++)
      {
         const long long J = (long long) (h->first_col_diag + diag->j[k] + base_j);
         if (diag->data) { fprintf(f, "%lld %lld %.14e\n", I, J, diag->data[k]); }
         else { fprintf(f, "%lld %lld\n", I, J); }
      }
      if (has_offd)
      {
         for (HYPRE_Int k = offd->i[i]; k < offd->i[i + 1]; k++)
         {
            const long long J = (long long) (h->col_map_offd[offd->j[k]] + base_j);
            if (offd->data) { fprintf(f, "%lld %lld %.14e\n", I, J, offd->data[k]); }
            else { fprintf(f, "%lld %lld\n", I, J); }
         }
      }
   }
   fclose(f);
   if (on_device) { hypre_ParCSRMatrixDestroy(h); }
   return hypre_error_flag;
}

HYPRE_Int HYPRE_IJMatrixPrint(HYPRE_IJMatrix matrix, const char *filename)
{
   if (!matrix || matrix->object_type != HYPRE_PARCSR) { hypre_error_in_arg(1); return hypre_error_flag; }
   return hypre_ParCSRMatrixPrintIJ((hypre_ParCSRMatrix *) matrix->object, 0, 0, filename);
}

HYPRE_Int HYPRE_IJMatrixGetObject(HYPRE_IJMatrix matrix, void **object)
{
   if (!matrix) { hypre_error_in_arg(1); return hypre_error_flag; }
   *object = matrix->object;
   return hypre_error_flag;
}

HYPRE_Int HYPRE_IJMatrixDestroy(HYPRE_IJMatrix matrix)
{
   if (!matrix) { hypre_error_in_arg(1); return hypre_error_flag; }
   if (matrix->object_type == HYPRE_PARCSR)
   {
      if (matrix->object && !matrix->translator) { hypre_ParCSRMatrixDestroy((hypre_ParCSRMatrix *) matrix->object); }
   }
   else if (matrix->object_type != -1) { hypre_error_in_arg(1); return hypre_error_flag; }
   free(matrix);
   return hypre_error_flag;
}

// `translator` is the reference's assembly scratch, never live on a finished matrix; a wrapper marks
// itself there so that Destroy leaves the borrowed object alone
static char borrowed_tag;

HYPRE_IJMatrix hypre_amd_IJMatrixWrap(hypre_ParCSRMatrix *A)
{
   if (!A) { hypre_error_in_arg(1); return nullptr; }
   hypre_IJMatrix *ij = (hypre_IJMatrix *) calloc(1, sizeof(hypre_IJMatrix));
   ij->comm = A->comm;
   ij->row_partitioning[0] = A->row_starts[0]; ij->row_partitioning[1] = A->row_starts[1];
   ij->col_partitioning[0] = A->col_starts[0]; ij->col_partitioning[1] = A->col_starts[1];
   ij->object_type = HYPRE_PARCSR;
   ij->object = A;
   ij->translator = &borrowed_tag;
   ij->assemble_flag = 1;
   ij->global_num_rows = A->global_num_rows;
   ij->global_num_cols = A->global_num_cols;
   return ij;
}

HYPRE_Int HYPRE_IJVectorRead(const char *filename, MPI_Comm comm, HYPRE_Int type, HYPRE_IJVector *vector_ptr)
{
   if (vector_ptr) { *vector_ptr = nullptr; }
   if (!filename || !vector_ptr || !comm_ops(comm)) { hypre_error_in_arg(1); return hypre_error_flag; }
   if (type != HYPRE_PARCSR) { hypre_error_in_arg(3); return hypre_error_flag; }
   HYPRE_Int rank = 0;
   hypre_MPI_Comm_rank(comm, &rank);
   std::string text;
   const bool bad_open = !slurp(rank_file(filename, rank), text);
   if (any_rank(comm, bad_open)) { hypre_error_in_arg(1); return hypre_error_flag; }
   Cursor c{text.data(), text.data() + text.size()};
   HYPRE_BigInt jlower = 0, jupper = -1;
   bool bad = !(c.big(jlower) && c.big(jupper)) || jlower > jupper + 1 || jlower < 0;
   const HYPRE_Int n = bad ? 0 : (HYPRE_Int) (jupper - jlower + 1);
   std::vector<double> vals((size_t) n, 0.0);
   std::vector<OffProcEntry> off_proc;
   while (!bad && !c.at_end())
   {
      HYPRE_BigInt j; double v;
      if (!(c.big(j) && c.blank_then_real(v))) { bad = true; break; }
      if (j < jlower || j > jupper) { off_proc.push_back(OffProcEntry{j, 0, v}); continue; }   // AddToValues on the owner
      vals[(size_t) (j - jlower)] = v;
   }
   if (any_rank(comm, bad))
   {
      hypre_error_w_msg(HYPRE_ERROR_GENERIC, "Error in IJ vector input file.");
      return hypre_error_flag;
   }
   for (const OffProcEntry &e : gather_off_proc(comm, off_proc))
   {
      if (e.I >= jlower && e.I <= jupper) { vals[(size_t) (e.I - jlower)] += e.v; }
   }
   hypre_IJVector *ij = (hypre_IJVector *) calloc(1, sizeof(hypre_IJVector));
   ij->comm = comm;
   ij->partitioning[0] = jlower; ij->partitioning[1] = jupper + 1;
   ij->num_components = 1;
   ij->object_type = HYPRE_PARCSR;
   global_span(comm, jlower, jupper, ij->global_first_row, ij->global_num_rows);
   HYPRE_BigInt part[2] = {jlower - ij->global_first_row, jupper + 1 - ij->global_first_row};
   ij->object = hypre_amd_ParVectorFromArray(comm, ij->global_num_rows, part, vals.data(), HYPRE_MEMORY_HOST);
   *vector_ptr = ij;
   return hypre_error_flag;
}

HYPRE_Int HYPRE_IJVectorPrint(HYPRE_IJVector vector, const char *filename)
{
   if (!vector) { hypre_error_in_arg(1); return hypre_error_flag; }
   HYPRE_Int rank = 0;
   hypre_MPI_Comm_rank(vector->comm, &rank);
   FILE *f = fopen(rank_file(filename, rank).c_str(), "w");
   if (!f) { hypre_error_in_arg(2); return hypre_error_flag; }
   hypre_ParVector *v = (hypre_ParVector *) vector->object;
   const HYPRE_BigInt jlower = vector->partitioning[0], jupper = vector->partitioning[1] - 1;
   const HYPRE_Int n = (HYPRE_Int) (jupper - jlower + 1);
   std::vector<double> vals((size_t) std::max(n, 1));
   hypre_amd_ParVectorToArray(v, vals.data());
   fprintf(f, "%lld %lld\n", (long long) jlower, (long long) jupper);
   for (HYPRE_Int k = 0; k < n; k++) { fprintf(f, "%lld %.14e\n", (long long) (jlower + k), vals[(size_t) k]); }
   fclose(f);
   return hypre_error_flag;
}

HYPRE_Int HYPRE_IJVectorGetObject(HYPRE_IJVector vector, void **object)
{
   if (!vector) { hypre_error_in_arg(1); return hypre_error_flag; }
   *object = vector->object;
   return hypre_error_flag;
}

HYPRE_Int HYPRE_IJVectorDestroy(HYPRE_IJVector vector)
{
   if (!vector) { hypre_error_in_arg(1); return hypre_error_flag; }
   if (vector->object && !vector->translator) { hypre_ParVectorDestroy((hypre_ParVector *) vector->object); }
   free(vector);
   return hypre_error_flag;
}

HYPRE_IJVector hypre_amd_IJVectorWrap(hypre_ParVector *v)
{
   if (!v) { hypre_error_in_arg(1); return nullptr; }
   hypre_IJVector *ij = (hypre_IJVector *) calloc(1, sizeof(hypre_IJVector));
   ij->comm = v->comm;
   ij->partitioning[0] = v->partitioning[0]; ij->partitioning[1] = v->partitioning[1];
   ij->num_components = 1;
   ij->object_type = HYPRE_PARCSR;
   ij->object = v;
   ij->translator = &borrowed_tag;
   ij->global_num_rows = v->global_size;
   return ij;
}

}  // extern "C"
