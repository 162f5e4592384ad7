// hypre_amd — the BoomerAMG solve phase on the device: relaxation sweeps, the
// V/W/F cycle, the outer solve loop and the PCG caller.
//
// Reference counterparts (behaviour; the structure is MI355X-first):
//   parcsr_ls/par_relax.c:24-173,1178-1254,1506-1666   relaxation dispatcher, Jacobi, two-stage GS
//   parcsr_ls/par_relax_device.c:19-155                device smoothers
//   parcsr_ls/par_relax_interface.c:20-117             CF-ordered passes
//   parcsr_ls/par_gauss_elim.c:457-697                 coarsest-level solve
//   parcsr_ls/par_cycle.c:23-803                       cycle state machine
//   parcsr_ls/par_amg_solve.c:22-424                   outer loop, convergence test
//   krylov/pcg.c:318-1000                              PCG
//
// What is different from the reference's device path:
//   * a Jacobi / l1-Jacobi sweep is ONE pass over the matrix (SpMV with the
//     update fused into the row epilogue) instead of copy + SpMV + elementwise;
//   * sweeps are out-of-place and every level ping-pongs between two solution
//     buffers, so no copy-back pass exists inside a cycle;
//   * nothing inside a cycle synchronises with the host (the reference syncs
//     after every vector op and SpMV); one stream sync ends the cycle;
//   * the coarsest-level elimination runs on the device from precomputed
//     factors instead of a device->host->device round trip per cycle.
#include "amg_internal.hpp"
#include <algorithm>
#include <cmath>

using namespace hamd;

namespace hamd {

void AmgPrivate::drop_graph()
{
   if (graph_exec) { (void) hipGraphExecDestroy(graph_exec); graph_exec = nullptr; }
   if (graph) { (void) hipGraphDestroy(graph); graph = nullptr; }
   graph_state = 0; graph_sig = 0; graph_launches = 0;
   graph_cur.clear();
}

void AmgPrivate::release_device()
{
   drop_graph();
   graph_level = -1;
   for (double *p : u_alt) { if (p) { hypre_Free(p, HYPRE_MEMORY_DEVICE); } }
   u_alt.clear(); u_alt_len.clear();
   for (double *p : diag_buf) { if (p) { hypre_Free(p, HYPRE_MEMORY_DEVICE); } }
   diag_buf.clear(); diag_len.clear();
   if (d_coarse_lu) { hypre_Free(d_coarse_lu, HYPRE_MEMORY_DEVICE); d_coarse_lu = nullptr; }
   if (d_coarse_rhs) { hypre_Free(d_coarse_rhs, HYPRE_MEMORY_DEVICE); d_coarse_rhs = nullptr; }
   coarse_n = 0;
   if (tail_image) { (void) hipFree(tail_image); tail_image = nullptr; }
   tail_image_sig = 0; small_tail_level = -2; small_tail_used = -2; tail_outside = -1;
   if (mp_r) { hypre_ParVectorDestroy(mp_r); mp_r = nullptr; }
   if (mp_e) { hypre_ParVectorDestroy(mp_e); mp_e = nullptr; }
}

double *AmgPrivate::level_diag(int level, int n)
{
   if ((int) diag_buf.size() <= level) { diag_buf.resize((size_t) level + 1, nullptr); diag_len.resize((size_t) level + 1, 0); }
   if (diag_len[(size_t) level] < n || !diag_buf[(size_t) level])
   {
      if (diag_buf[(size_t) level]) { hypre_Free(diag_buf[(size_t) level], HYPRE_MEMORY_DEVICE); drop_graph(); graph_state = 0; }
      diag_buf[(size_t) level] = hypre_TAlloc(double, (size_t) std::max(n, 1), HYPRE_MEMORY_DEVICE);
      diag_len[(size_t) level] = n;
   }
   return diag_buf[(size_t) level];
}

// ---------------------------------------------------------------------------
// raw-pointer cores
// ---------------------------------------------------------------------------
static inline void wrap(hypre_Vector *v, const double *data, int size)
{
   memset(v, 0, sizeof(*v));
   v->data = const_cast<double *>(data);
   v->size = size; v->num_vectors = 1; v->vecstride = size; v->idxstride = 1;
   v->memory_location = HYPRE_MEMORY_DEVICE;
}
static inline void wrap_par(hypre_ParVector *pv, hypre_Vector *lv, MPI_Comm comm, HYPRE_BigInt gsize)
{
   memset(pv, 0, sizeof(*pv));
   pv->comm = comm; pv->global_size = gsize; pv->local_vector = lv;
}

void dev_par_matvec(HYPRE_Complex alpha, hypre_ParCSRMatrix *A, const double *x, HYPRE_Complex beta,
                    const double *b, double *y)
{
   hypre_Vector xv, bv, yv;
   hypre_ParVector xp, bp, yp;
   wrap(&xv, x, A->diag->num_cols); wrap(&bv, b, A->diag->num_rows); wrap(&yv, y, A->diag->num_rows);
   wrap_par(&xp, &xv, A->comm, A->global_num_cols);
   wrap_par(&bp, &bv, A->comm, A->global_num_rows);
   wrap_par(&yp, &yv, A->comm, A->global_num_rows);
   hypre_ParCSRMatrixMatvecOutOfPlaceDevice(alpha, A, &xp, beta, &bp, &yp);
}

void dev_par_matvecT(HYPRE_Complex alpha, hypre_ParCSRMatrix *A, const double *x, HYPRE_Complex beta, double *y)
{
   hypre_Vector xv, yv;
   hypre_ParVector xp, yp;
   wrap(&xv, x, A->diag->num_rows); wrap(&yv, y, A->diag->num_cols);
   wrap_par(&xp, &xv, A->comm, A->global_num_rows);
   wrap_par(&yp, &yv, A->comm, A->global_num_cols);
   hypre_ParCSRMatrixMatvecTDevice(alpha, A, &xp, beta, &yp);
}

// u_out = u_in + (w f - w A u_in)./d  on rows with cf_marker == relax_points
// (all rows when relax_points == 0); other rows are copied.  Single rank: one
// fused pass.  Distributed: the interior pass is fused as well and the ghost
// couplings are folded in by a second, halo-sized pass.
void dev_jacobi_sweep(hypre_ParCSRMatrix *A, const double *f, const int *cf_marker, int relax_points,
                      double w, const double *d, const double *u_in, double *u_out)
{
   hipStream_t s = stream();
   hypre_CSRMatrix *diag = A->diag;
   const int n = diag->num_rows;
   if (n <= 0) { return; }
   HYPRE_Int nprocs;
   hypre_MPI_Comm_size(A->comm, &nprocs);
   // distributed: start the halo of u_in, run the fused interior sweep while it
   // travels, then fold the ghost couplings into the boundary rows
   //    u_out[row] -= w * (A_offd u_ghost)[row] / d[row]        (halo-sized pass)
   hypre_ParCSRCommHandle *ch = nprocs > 1 ? dev_halo_begin(A, u_in) : nullptr;
   SpmvPlan *plan = get_plan(diag);
   SpmvArgs a{};
   a.Ai = diag->i; a.Aj = diag->j; a.Aa = diag->data; a.Aa32 = nullptr;
   a.x = u_in; a.b = f; a.y = u_out; a.aux = nullptr; a.d = d;
   a.marker = cf_marker; a.marker_val = relax_points;
   a.alpha = w; a.beta = 0.0; a.fill = HYPRE_SPMV_FILL_WHOLE; a.row_offset = 0;
   spmv_default_flags(a);
   launch_spmv(plan, a, (relax_points != 0 && cf_marker) ? OP_JACOBI_CF : OP_JACOBI, s);
   dev_halo_end(ch);
   hypre_CSRMatrix *offd = A->offd;
   if (nprocs > 1 && offd->num_cols > 0 && offd->num_nonzeros > 0)
   {
      SpmvArgs o{};
      o.Ai = offd->i; o.Aj = offd->j; o.Aa = offd->data;
      o.Aa32 = handle().fp32_values ? fp32_values_of(offd) : nullptr;     // mixed precision covers the ghost block too
      o.x = A->comm_pkg->tmp_data; o.y = u_out; o.d = d;
      o.marker = (relax_points != 0) ? cf_marker : nullptr; o.marker_val = relax_points;
      o.alpha = -w;
      if (offd->rownnz) { launch_spmv_rownnz(offd->rownnz, offd->num_rownnz, o, s); }
      else { launch_spmv_allrows_update(offd->num_rows, o, s); }
   }
}

}  // namespace hamd

// per-matrix scratch for the in-place public relax entry (result lands in the
// caller's u, so one extra vector is needed for the out-of-place sweep)
static double *relax_scratch(size_t n)
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
static double *diag_scratch(size_t n)
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

namespace { int &cycle_fusion(); int &small_tail_on(); int &small_tail_forced_form(); }     // (defined with the cycle below)

extern "C" {

// ===========================================================================
// relaxation (public, in-place on u as the reference)
// ===========================================================================
HYPRE_Int hypre_BoomerAMGRelaxTwoStageGaussSeidelDevice(hypre_ParCSRMatrix *A, hypre_ParVector *f,
                                                        HYPRE_Real relax_weight, HYPRE_Real omega,
                                                        HYPRE_Real *A_diag_diag, hypre_ParVector *u,
                                                        hypre_ParVector *r, hypre_ParVector *z,
                                                        HYPRE_Int num_inner_iters)
{
   (void) omega;
   hipStream_t s = stream();
   hypre_CSRMatrix *diag = A->diag;
   const int n = diag->num_rows;
   double *ud = u->local_vector->data, *rd = r->local_vector->data, *zd = z->local_vector->data;
   const int saved = handle().sync_compute;
   handle().sync_compute = 0;
   // 0) r = w (f - A u)     1) z = r./D ; u += z
   // u known to be zero (par_vector.h all_zeros; set by SetZeros and by the cycle on the way down):
   // A u = 0 exactly, so r = w f without touching the matrix.  The reference uses the flag this way
   // only in its Jacobi sweep (par_relax.c:1221-1228); the result here is the same bits either way.
   HYPRE_Int nprocs = 1;
   hypre_MPI_Comm_size(A->comm, &nprocs);
   hypre_CSRMatrix *Lm = (n > 0 && diag->num_nonzeros > 0) ? strict_lower_of(diag) : nullptr;
   // At least one inner step over a non-empty triangle (hypre_amd_SetCycleFusion): on one rank the sweep in two or three
   // passes instead of four or five — stage 0 and the scaling of stage 1 in one kernel (z = (w f - w A u) .* (1 ./ D): an
   // epilogue of the residual's pass; from a zero iterate one vector kernel), and the "u += z" of stage 1 folded into the
   // first inner step's epilogue (u = (u + z) + mult z'): every rounding where it was, 24 - 32 bytes per row less.
   // Several ranks: the residual needs the ghost block's part before it can be scaled, so it stays a pass of its own, the
   // scaling a vector kernel (z = r .* (1 ./ D), without the "u += z"), and the first inner step — local by construction: the
   // triangle of this rank's own diagonal block — folds the "u += z" in all the same.
   const bool fused = cycle_fusion() && Lm && Lm->num_nonzeros > 0 && num_inner_iters >= 1;
   const bool one_rank = nprocs == 1 && A->offd->num_nonzeros == 0;
   const bool from_zero = u->all_zeros != 0;
   if (fused)
   {
      if (from_zero) { launch_scaled_recip(relax_weight, f->local_vector->data, A_diag_diag, zd, (size_t) n, s); }
      else if (!one_rank)
      {
         dev_par_matvec(-relax_weight, A, ud, relax_weight, f->local_vector->data, rd);
         launch_scaled_recip(1.0, rd, A_diag_diag, zd, (size_t) n, s);           // (1.0 * r is r: the product with the reciprocal as before)
      }
      else
      {
         SpmvArgs a{};
         a.Ai = diag->i; a.Aj = diag->j; a.Aa = diag->data; a.Aa32 = nullptr;
         a.x = ud; a.b = f->local_vector->data; a.y = zd; a.aux = nullptr; a.d = A_diag_diag; a.marker = nullptr;
         a.alpha = -relax_weight; a.beta = relax_weight; a.fill = HYPRE_SPMV_FILL_WHOLE;
         spmv_default_flags(a);
         launch_spmv(get_plan(diag), a, OP_RESID_RD, s);
      }
   }
   else
   {
      if (from_zero) { launch_scale_copy(relax_weight, f->local_vector->data, rd, (size_t) n, s); }
      else { dev_par_matvec(-relax_weight, A, ud, relax_weight, f->local_vector->data, rd); }
      launch_diagscale2(A_diag_diag, rd, 1.0, zd, ud, 1, (size_t) n, s);
   }
   double mult = -1.0;
   double *zin = zd, *zout = rd;
   if (Lm)
   {
      // the inner steps run on a cached copy of the strictly lower triangle (half the bytes of
      // masking the full matrix, which is what the reference's fill-mode SpMV does)
      SpmvPlan *plan = get_plan(Lm);
      for (int k = 0; k < num_inner_iters && Lm->num_nonzeros > 0; k++)
      {
         if (fused && k == 0)
         {
            // z' = (L z)./D ; u = (u + z) + mult z'   (u is not read when it is known to be zero)
            SpmvArgs a{};
            a.Ai = Lm->i; a.Aj = Lm->j; a.Aa = Lm->data; a.Aa32 = nullptr;
            a.x = zin; a.b = nullptr; a.y = zout; a.aux = ud; a.d = A_diag_diag; a.marker = nullptr;
            a.alpha = mult; a.beta = from_zero ? 0.0 : 1.0; a.fill = HYPRE_SPMV_FILL_WHOLE;
            spmv_default_flags(a);
            launch_spmv(plan, a, OP_TSGS_FIRST, s);
            std::swap(zin, zout);
            mult *= -1.0;
            continue;
         }
         // 2+3) z_out = (L_strict z_in)./D ; u += mult * z_out   — one fused pass
         SpmvArgs a{};
         a.Ai = Lm->i; a.Aj = Lm->j; a.Aa = Lm->data; a.Aa32 = nullptr;
         a.x = zin; a.b = nullptr; a.y = zout; a.aux = ud; a.d = A_diag_diag; a.marker = nullptr;
         a.alpha = mult; a.beta = 0.0; a.fill = HYPRE_SPMV_FILL_WHOLE;
         spmv_default_flags(a);
         launch_spmv(plan, a, OP_TSGS, s);
         std::swap(zin, zout);
         mult *= -1.0;
      }
   }
   u->all_zeros = 0;
   handle().sync_compute = saved;
   maybe_sync();
   return hypre_error_flag;
}

HYPRE_Int hypre_BoomerAMGRelax(hypre_ParCSRMatrix *A, hypre_ParVector *f, HYPRE_Int *cf_marker,
                               HYPRE_Int relax_type, HYPRE_Int relax_points, HYPRE_Real relax_weight,
                               HYPRE_Real omega, HYPRE_Real *l1_norms, hypre_ParVector *u,
                               hypre_ParVector *Vtemp, hypre_ParVector *Ztemp)
{
   HYPRE_AMD_REQUIRE_DEVICE(A->diag->memory_location, "hypre_BoomerAMGRelax(A)");
   HYPRE_AMD_REQUIRE_DEVICE(u->local_vector->memory_location, "hypre_BoomerAMGRelax(u)");
   HYPRE_AMD_REQUIRE_DEVICE(f->local_vector->memory_location, "hypre_BoomerAMGRelax(f)");
   hipStream_t s = stream();
   const int n = A->diag->num_rows;
   const int saved = handle().sync_compute;
   handle().sync_compute = 0;
   double *ud = u->local_vector->data;
   const double *fd = f->local_vector->data;
   switch (relax_type)
   {
      case 0: case 7: case 18:
      {
         // smoother diagonal: l1 norms (18), stored diagonal vector (7), first row entry (0)
         const double *d = l1_norms;
         if (relax_type == 0 || !d)
         {
            double *dd = diag_scratch((size_t) std::max(n, 1));
            launch_diag_first(A->diag->i, A->diag->data, dd, n, s);
            d = dd;
         }
         if (u->all_zeros && relax_type != 0)
         {
            // par_relax.c:1221-1228: u == 0 => u = (w f)./d, no SpMV
            launch_scaled_div(relax_weight, fd, d, ud, relax_points ? cf_marker : nullptr, relax_points, (size_t) n, s);
         }
         else
         {
            double *tmp = relax_scratch((size_t) std::max(n, 1));
            dev_jacobi_sweep(A, fd, cf_marker, relax_points, relax_weight, d, ud, tmp);
            launch_copy(ud, tmp, (size_t) n, s);
         }
         break;
      }
      case 11:
         hypre_BoomerAMGRelaxTwoStageGaussSeidelDevice(A, f, relax_weight, omega, l1_norms, u, Vtemp, Ztemp, 1);
         break;
      case 12:
         hypre_BoomerAMGRelaxTwoStageGaussSeidelDevice(A, f, relax_weight, omega, l1_norms, u, Vtemp, Ztemp, 2);
         break;
      // hybrid Gauss-Seidel / SOR family (par_relax.c:1256-1377): plain 3 forward, 4 backward,
      // 6 symmetric; l1-scaled 13 forward, 14 backward, 8/88 symmetric, 89 = 13 then 14
      case 3:
         hypre_BoomerAMGRelaxHybridGaussSeidelDevice(A, f, cf_marker, relax_points, relax_weight, omega, nullptr, u, Vtemp, Ztemp, 1, 0);
         break;
      case 4:
         hypre_BoomerAMGRelaxHybridGaussSeidelDevice(A, f, cf_marker, relax_points, relax_weight, omega, nullptr, u, Vtemp, Ztemp, -1, 0);
         break;
      case 6:
         hypre_BoomerAMGRelaxHybridGaussSeidelDevice(A, f, cf_marker, relax_points, relax_weight, omega, nullptr, u, Vtemp, Ztemp, 1, 1);
         break;
      case 8: case 88: case 13: case 14: case 89:
         if (!l1_norms)
         {
            hypre_error_w_msg(HYPRE_ERROR_GENERIC, "hypre_BoomerAMGRelax: the l1 Gauss-Seidel variants need l1_norms");
            break;
         }
         if (relax_type == 8 || relax_type == 88) { hypre_BoomerAMGRelaxHybridGaussSeidelDevice(A, f, cf_marker, relax_points, relax_weight, omega, l1_norms, u, Vtemp, Ztemp, 1, 1); }
         if (relax_type == 13 || relax_type == 89) { hypre_BoomerAMGRelaxHybridGaussSeidelDevice(A, f, cf_marker, relax_points, relax_weight, omega, l1_norms, u, Vtemp, Ztemp, 1, 0); }
         if (relax_type == 14 || relax_type == 89) { hypre_BoomerAMGRelaxHybridGaussSeidelDevice(A, f, cf_marker, relax_points, relax_weight, omega, l1_norms, u, Vtemp, Ztemp, -1, 0); }
         break;
      // multicolour Gauss-Seidel (no reference counterpart; par_relax_mc.cpp): 21 colours ascending, 22 descending
      case 21:
         hypre_BoomerAMGRelaxMultiColorGaussSeidelDevice(A, f, cf_marker, relax_points, relax_weight, l1_norms, u, Vtemp, 1);
         break;
      case 22:
         hypre_BoomerAMGRelaxMultiColorGaussSeidelDevice(A, f, cf_marker, relax_points, relax_weight, l1_norms, u, Vtemp, -1);
         break;
      default:
         hypre_error_w_msg(HYPRE_ERROR_GENERIC, "hypre_BoomerAMGRelax: relax_type is outside the scope of this library");
         break;
   }
   u->all_zeros = 0;
   handle().sync_compute = saved;
   maybe_sync();
   return 0;     // relax_error (par_relax.c:36,172)
}

HYPRE_Int hypre_BoomerAMGRelaxIF(hypre_ParCSRMatrix *A, hypre_ParVector *f, HYPRE_Int *cf_marker,
                                 HYPRE_Int relax_type, HYPRE_Int relax_order, HYPRE_Int cycle_param,
                                 HYPRE_Real relax_weight, HYPRE_Real omega, HYPRE_Real *l1_norms,
                                 hypre_ParVector *u, hypre_ParVector *Vtemp, hypre_ParVector *Ztemp)
{
   HYPRE_Int err = 0;
   if (relax_order == 1 && cycle_param < 3)
   {
      const HYPRE_Int pts[2] = {cycle_param < 2 ? 1 : -1, cycle_param < 2 ? -1 : 1};
      for (int i = 0; i < 2; i++)
      {
         err = hypre_BoomerAMGRelax(A, f, cf_marker, relax_type, pts[i], relax_weight, omega, l1_norms, u, Vtemp, Ztemp);
      }
   }
   else
   {
      err = hypre_BoomerAMGRelax(A, f, cf_marker, relax_type, 0, relax_weight, omega, l1_norms, u, Vtemp, Ztemp);
   }
   return err;
}

HYPRE_Int hypre_ParCSRRelax_L1_Jacobi(hypre_ParCSRMatrix *A, hypre_ParVector *f, HYPRE_Int *cf_marker,
                                      HYPRE_Int relax_points, HYPRE_Real relax_weight, HYPRE_Real *l1_norms,
                                      hypre_ParVector *u, hypre_ParVector *Vtemp)
{
   return hypre_BoomerAMGRelax(A, f, cf_marker, 18, relax_points, relax_weight, 0.0, l1_norms, u, Vtemp, nullptr);
}

HYPRE_Int hypre_BoomerAMGRelax_FCFJacobi(hypre_ParCSRMatrix *A, hypre_ParVector *f, HYPRE_Int *cf_marker,
                                         HYPRE_Real relax_weight, hypre_ParVector *u, hypre_ParVector *Vtemp)
{
   const HYPRE_Int pts[3] = {-1, 1, -1};
   for (int i = 0; i < 3; i++) { hypre_BoomerAMGRelax(A, f, cf_marker, 0, pts[i], relax_weight, 0.0, nullptr, u, Vtemp, nullptr); }
   return hypre_error_flag;
}

// ===========================================================================
// coarsest level
// ===========================================================================
// Factor the dense operator once with the reference's elimination order; the
// device kernel then replays the right-hand-side operations of hypre_gselim.
static void ensure_coarse_factors(hypre_ParAMGData *d)
{
   AmgPrivate *pv = (AmgPrivate *) d->amd_private;
   if (pv->d_coarse_lu || !d->A_mat) { return; }
   hypre_ParCSRMatrix *A = d->A_array[d->num_levels - 1];
   const int n = (int) A->global_num_rows;
   std::vector<double> M(d->A_mat, d->A_mat + (size_t) n * n);
   for (int k = 0; k < n - 1; k++)
   {
      if (M[(size_t) k * n + k] != 0.0)
      {
         const double divA = 1.0 / M[(size_t) k * n + k];
         for (int j = k + 1; j < n; j++)
         {
            if (M[(size_t) j * n + k] != 0.0)
            {
               const double factor = M[(size_t) j * n + k] * divA;
               for (int m = k + 1; m < n; m++) { M[(size_t) j * n + m] -= factor * M[(size_t) k * n + m]; }
               M[(size_t) j * n + k] = factor;      // keep the multiplier in the eliminated slot
            }
         }
      }
      else
      {
         for (int j = k + 1; j < n; j++) { M[(size_t) j * n + k] = 0.0; }
      }
   }
   pv->coarse_n = n;
   pv->coarse_first_row = (int) A->first_row_index;
   pv->d_coarse_lu = hypre_TAlloc(double, (size_t) std::max(n * n, 1), HYPRE_MEMORY_DEVICE);
   pv->d_coarse_rhs = hypre_TAlloc(double, (size_t) std::max(n, 1), HYPRE_MEMORY_DEVICE);
   hypre_TMemcpy(pv->d_coarse_lu, M.data(), double, (size_t) n * n, HYPRE_MEMORY_DEVICE, HYPRE_MEMORY_HOST);
}

HYPRE_Int hypre_GaussElimSolve(hypre_ParAMGData *d, HYPRE_Int level, HYPRE_Int relax_type)
{
   (void) relax_type;
   AmgPrivate *pv = (AmgPrivate *) d->amd_private;
   hypre_ParCSRMatrix *A = d->A_array[level];
   if (!d->gs_setup) { hypre_GaussElimSetup(d, level, relax_type); }
   ensure_coarse_factors(d);
   hipStream_t s = stream();
   const int n = pv->coarse_n, nloc = A->diag->num_rows;
   double *fd = d->F_array[level]->local_vector->data;
   double *ud = d->U_array[level]->local_vector->data;
   HYPRE_Int nprocs;
   hypre_MPI_Comm_size(A->comm, &nprocs);
   if (nprocs == 1)
   {
      HIP_CHECK(hipMemcpyAsync(pv->d_coarse_rhs, fd, sizeof(double) * (size_t) n, hipMemcpyDeviceToDevice, s));
      launch_coarse_solve(pv->d_coarse_lu, pv->d_coarse_rhs, n, s);
      HIP_CHECK(hipMemcpyAsync(ud, pv->d_coarse_rhs, sizeof(double) * (size_t) n, hipMemcpyDeviceToDevice, s));
   }
   else
   {
      // gather the (<= max_coarse_size) right-hand side on every rank: zero-fill,
      // drop the own slice in, sum over ranks (one tiny all-reduce instead of an
      // Allgatherv through the host, par_gauss_elim.c:575)
      HIP_CHECK(hipMemsetAsync(pv->d_coarse_rhs, 0, sizeof(double) * (size_t) n, s));
      if (nloc) { HIP_CHECK(hipMemcpyAsync(pv->d_coarse_rhs + pv->coarse_first_row, fd, sizeof(double) * (size_t) nloc, hipMemcpyDeviceToDevice, s)); }
      dev_allreduce_sum(A->comm, pv->d_coarse_rhs, n);
      launch_coarse_solve(pv->d_coarse_lu, pv->d_coarse_rhs, n, s);
      if (nloc) { HIP_CHECK(hipMemcpyAsync(ud, pv->d_coarse_rhs + pv->coarse_first_row, sizeof(double) * (size_t) nloc, hipMemcpyDeviceToDevice, s)); }
   }
   return hypre_error_flag;
}

// ===========================================================================
// the cycle
// ===========================================================================
namespace {

// where a level's current iterate lives: its home vector or the alternate
struct LevelVec
{
   double *home = nullptr, *alt = nullptr, *cur = nullptr;
   double *other() const { return cur == home ? alt : home; }
   void flip() { cur = other(); }
};

// Levels pv->tail_level and below on this rank's replica (par_amg_replicate.cpp): gather f, cycle, keep own slice.
int run_replicated_tail(hypre_ParAMGData *d, AmgPrivate *pv, const double *f_local, double *u_local, double *op_count)
{
   hypre_ParAMGData *t = pv->tail;
   hypre_ParCSRMatrix *Al = d->A_array[pv->tail_level];
   const int nloc = Al->diag->num_rows;
   const size_t first = (size_t) Al->row_starts[0];
   const int ng = (int) t->A_array[0]->global_num_rows;
   hipStream_t s = stream();
   double *g = pv->d_tail_f;
   HIP_CHECK(hipMemsetAsync(g, 0, sizeof(double) * (size_t) ng, s));
   if (nloc) { HIP_CHECK(hipMemcpyAsync(g + first, f_local, sizeof(double) * (size_t) nloc, hipMemcpyDeviceToDevice, s)); }
   dev_allreduce_sum(Al->comm, g, ng);
   HIP_CHECK(hipMemcpyAsync(t->F_array[0]->local_vector->data, g, sizeof(double) * (size_t) ng, hipMemcpyDeviceToDevice, s));
   hypre_ParVectorSetZeros(t->U_array[0]);
   ((AmgPrivate *) t->amd_private)->mixed_precision = pv->mixed_precision;     // may have been switched after setup
   t->cycle_op_count = 0;
   const int err = hypre_BoomerAMGCycle(t, t->F_array, t->U_array);
   if (nloc)
   {
      HIP_CHECK(hipMemcpyAsync(u_local, t->U_array[0]->local_vector->data + first, sizeof(double) * (size_t) nloc,
                               hipMemcpyDeviceToDevice, s));
   }
   if (op_count) { *op_count += t->cycle_op_count; }
   return err;
}

// fusions across the cycle's steps (hypre_amd_SetCycleFusion; HYPRE_AMD_CYCLE_FUSION=0): same results, fewer passes
int &cycle_fusion()
{
   static int on = [] { const char *e = getenv("HYPRE_AMD_CYCLE_FUSION"); return e ? atoi(e) : 1; }();
   return on;
}
// the smallest levels of a Jacobi V(1,1) cycle in one kernel of one workgroup (tail_kernels.hip; hypre_amd_SetSmallTail)
int &small_tail_on()
{
   static int on = [] { const char *e = getenv("HYPRE_AMD_SMALL_TAIL"); return e ? atoi(e) : 1; }();
   return on;
}
// The image of levels st .. L-1 the one-workgroup tail copies into LDS (tail_kernels.hip): every array the walk reads, at the
// offset the kernel addresses it, and behind it the room of the work vectors.  Built when the hierarchy (its arrays'
// addresses) is new; false: the levels do not fit the LDS of one workgroup.
// form: 0 every array in LDS; 1 the first level's operator in the lanes' registers (its rows times their lanes fit the
// workgroup's 768 lanes, a lane's share is at most 32 entries); 2 the first level's operator streamed from where it lies
bool build_small_tail_image(hypre_ParAMGData *d, AmgPrivate *pv, int st, hipStream_t s, int form = 0)
{
   const bool first_operator_outside = form != 0;
   const int L = d->num_levels;
   hypre_ParCSRMatrix **A = d->A_array, **P = d->P_array;
   unsigned long long h = 1469598103934665603ull;
   auto mix = [&](unsigned long long v) { h ^= v; h *= 1099511628211ull; };
   mix((unsigned long long) st); mix((unsigned long long) L); mix((unsigned long long) (uintptr_t) pv->d_coarse_lu); mix((unsigned long long) pv->coarse_n);
   mix((unsigned long long) form);
   for (int l = st; l < L; l++)
   {
      mix((unsigned long long) (uintptr_t) A[l]->diag->i); mix((unsigned long long) (uintptr_t) A[l]->diag->data); mix((unsigned long long) A[l]->diag->num_nonzeros);
      if (l < L - 1)
      {
         mix((unsigned long long) (uintptr_t) P[l]->diag->data); mix((unsigned long long) (uintptr_t) P[l]->diagT->data);
         mix((unsigned long long) (uintptr_t) d->l1_norms[l]->data);
      }
   }
   if (pv->tail_image && pv->tail_image_sig == h) { return true; }
   if (pv->tail_image) { (void) hipFree(pv->tail_image); pv->tail_image = nullptr; pv->tail_image_sig = 0; }
   int reg_lanes = 0;
   if (form == 1)
   {
      // lanes per row of the register form: enough that a lane holds at most 16 entries of the longest row, few enough that
      // all rows fit the workgroup at once
      hypre_CSRMatrix *Ad = A[st]->diag;
      std::vector<HYPRE_Int> ri((size_t) Ad->num_rows + 1);
      HIP_CHECK(hipStreamSynchronize(s));
      hypre_TMemcpy(ri.data(), Ad->i, HYPRE_Int, Ad->num_rows + 1, HYPRE_MEMORY_HOST, HYPRE_MEMORY_DEVICE);
      int longest = 0;
      for (HYPRE_Int r = 0; r < Ad->num_rows; r++) { longest = std::max(longest, (int) (ri[(size_t) r + 1] - ri[(size_t) r])); }
      // (the lanes per row the other forms take, where a lane's share then fits its registers: the same order of the sums)
      // (the register form runs 768 lanes, a lane holding at most 32 entries: tail_kernels.hip)
      int wmax = 1, wlen = 1;
      while (2 * wmax <= 64 && 2LL * wmax * std::max((int) Ad->num_rows, 1) <= 768) { wmax *= 2; }
      const int avg = Ad->num_rows > 0 ? (int) ((Ad->num_nonzeros + Ad->num_rows - 1) / Ad->num_rows) : 1;
      while (wlen < avg && wlen < 64) { wlen *= 2; }
      int W = std::min(wmax, wlen);
      while (32 * W < longest && W < 64) { W *= 2; }
      if (32 * W < longest || (long long) W * Ad->num_rows > 768 || Ad->num_nonzeros < 1) { return false; }
      reg_lanes = W;
   }
   SmallTailArgs &ta = pv->tail_args;
   ta = SmallTailArgs{};
   int off = 0;
   auto take = [&](size_t bytes) { const int o = off; off += (int) ((bytes + 15) & ~(size_t) 15); return o; };
   auto lanes = [](int rows, int entries)
   {
      int wmax = 1, wlen = 1;
      while (2 * wmax <= 64 && 2 * wmax * std::max(rows, 1) <= 1024) { wmax *= 2; }
      const int avg = rows > 0 ? (entries + rows - 1) / rows : 1;
      while (wlen < avg && wlen < 64) { wlen *= 2; }
      return std::min(wmax, wlen);
   };
   struct Piece { const void *src; int off; size_t bytes; };
   std::vector<Piece> pieces;
   auto put = [&](const void *src, size_t bytes) { const int o = take(bytes); pieces.push_back({src, o, bytes}); return o; };
   ta.nl = L - st;
   for (int l = st; l < L - 1; l++)
   {
      SmallTailLevel &q = ta.lv[l - st];
      hypre_CSRMatrix *Ad = A[l]->diag, *Pd = P[l]->diag, *Rd = P[l]->diagT;
      q.n = Ad->num_rows;
      q.Ai = put(Ad->i, sizeof(int) * ((size_t) Ad->num_rows + 1));
      if (l == st && first_operator_outside) { q.gAj = Ad->j; q.gAa = Ad->data; ta.gAi = Ad->i; ta.nnz0 = Ad->num_nonzeros; }
      else
      {
         q.Aj = put(Ad->j, sizeof(int) * (size_t) Ad->num_nonzeros);
         q.Aa = put(Ad->data, sizeof(double) * (size_t) Ad->num_nonzeros);
      }
      q.Pi = put(Pd->i, sizeof(int) * ((size_t) Pd->num_rows + 1)); q.Pj = put(Pd->j, sizeof(int) * (size_t) Pd->num_nonzeros);
      q.Pa = put(Pd->data, sizeof(double) * (size_t) Pd->num_nonzeros);
      q.Ri = put(Rd->i, sizeof(int) * ((size_t) Rd->num_rows + 1)); q.Rj = put(Rd->j, sizeof(int) * (size_t) Rd->num_nonzeros);
      q.Ra = put(Rd->data, sizeof(double) * (size_t) Rd->num_nonzeros);
      q.d = put(d->l1_norms[l]->data, sizeof(double) * (size_t) Ad->num_rows);
      q.wA = lanes(Ad->num_rows, Ad->num_nonzeros); q.wP = lanes(Pd->num_rows, Pd->num_nonzeros); q.wR = lanes(Rd->num_rows, Rd->num_nonzeros);
      if (l == st && form == 1) { q.wA = reg_lanes; ta.reg_first = 1; }
   }
   ta.lv[L - 1 - st].n = A[L - 1]->diag->num_rows;
   ta.ncoarse = pv->coarse_n;
   ta.lu_off = put(pv->d_coarse_lu, sizeof(double) * (size_t) pv->coarse_n * (size_t) pv->coarse_n);
   ta.image_bytes = off;
   for (int l = st; l < L; l++)
   {
      SmallTailLevel &q = ta.lv[l - st];
      q.f = take(sizeof(double) * (size_t) q.n); q.u = take(sizeof(double) * (size_t) q.n); q.alt = take(sizeof(double) * (size_t) q.n);
   }
   ta.vt_off = take(sizeof(double) * (size_t) ta.lv[0].n);
   ta.lds_bytes = off;
   if (off > 158 * 1024) { ta = SmallTailArgs{}; return false; }
   if (hipMalloc(&pv->tail_image, (size_t) std::max(ta.image_bytes, 16)) != hipSuccess) { (void) hipGetLastError(); pv->tail_image = nullptr; return false; }
   for (const Piece &pc : pieces)
   {
      if (pc.bytes) { HIP_CHECK(hipMemcpyAsync((char *) pv->tail_image + pc.off, pc.src, pc.bytes, hipMemcpyDeviceToDevice, s)); }
   }
   ta.image = pv->tail_image;
   pv->tail_image_sig = h;
   return true;
}

// tests: the form of the tail's image (-1: the first that fits)
int &small_tail_forced_form() { static int f = -1; return f; }
// everything the launches of the sub-cycle below level gl depend on
unsigned long long tail_signature(hypre_ParAMGData *d, AmgPrivate *pv, int gl, hypre_ParVector **F, hypre_ParVector **U)
{
   unsigned long long h = 1469598103934665603ull;
   auto mix = [&](unsigned long long v) { h ^= v; h *= 1099511628211ull; };
   auto mixd = [&](double v) { unsigned long long b; memcpy(&b, &v, sizeof b); mix(b); };
   mix((unsigned long long) d->num_levels); mix((unsigned long long) gl);
   for (int k = 0; k < 4; k++) { mix((unsigned long long) d->grid_relax_type[k]); mix((unsigned long long) d->num_grid_sweeps[k]); }
   mix((unsigned long long) d->relax_order); mix((unsigned long long) d->cycle_type); mix((unsigned long long) d->fcycle);
   mix((unsigned long long) d->user_relax_type); mix((unsigned long long) (d->grid_relax_points != nullptr));
   mix((unsigned long long) pv->mixed_precision); mix((unsigned long long) pv->emulated_threads);
   mix((unsigned long long) (uintptr_t) d->Vtemp->local_vector->data);
   for (int l = gl; l < d->num_levels; l++)
   {
      mixd(d->relax_weight[l]); mixd(d->omega[l]);
      mix((unsigned long long) (uintptr_t) d->A_array[l]->diag->data); mix((unsigned long long) (uintptr_t) d->A_array[l]->diag->j);
      mix((unsigned long long) d->A_array[l]->diag->num_nonzeros);
      if (l < d->num_levels - 1 && d->P_array[l])
      {
         mix((unsigned long long) (uintptr_t) d->P_array[l]->diag->data);
         mix((unsigned long long) (uintptr_t) (d->P_array[l]->diagT ? d->P_array[l]->diagT->data : nullptr));
      }
      mix((unsigned long long) (uintptr_t) F[l]->local_vector->data); mix((unsigned long long) (uintptr_t) U[l]->local_vector->data);
      mix((unsigned long long) (uintptr_t) (d->l1_norms[l] ? d->l1_norms[l]->data : nullptr));
      mix((unsigned long long) (uintptr_t) (l < (int) pv->u_alt.size() ? pv->u_alt[(size_t) l] : nullptr));
      mix((unsigned long long) (uintptr_t) (l < (int) pv->diag_buf.size() ? pv->diag_buf[(size_t) l] : nullptr));
   }
   mix(plan_generation());            // a plan dropped or rebuilt anywhere (the recorded kernels point into plans' tables)
   mix((unsigned long long) cycle_fusion());     // (the recorded tail leaves out what a fused restriction into it did)
   mix((unsigned long long) (small_tail_on() ? pv->small_tail_nnz : 0)); mix((unsigned long long) (small_tail_forced_form() + 1));
   mix((unsigned long long) hypre_amd_SetMcLookAhead(-1));     // (one kernel instead of a dozen launches in the recorded tail)
   SpmvArgs a{};
   spmv_default_flags(a);
   mix((unsigned long long) a.variant); mix((unsigned long long) a.gather_t); mix((unsigned long long) a.xcd_map);
   return h;
}

bool is_jacobi_type(int t) { return t == 0 || t == 7 || t == 18; }
bool is_ge_type(int t) { return t == 9 || t == 19 || t == 98 || t == 99 || t == 198 || t == 199; }

}  // namespace

// Fusions across the steps of a cycle from now on (default 1): the restriction also writes the coarse level's first sweep
// from zero.  Same bits either way; a switch for comparisons.  on < 0: unchanged; returns the setting.
HYPRE_Int hypre_amd_SetCycleFusion(HYPRE_Int on)
{
   if (on >= 0) { cycle_fusion() = on != 0; }
   return cycle_fusion();
}

// The smallest levels of a V(1,1) cycle with Jacobi-type smoothing in one kernel of one workgroup from now on (default 1):
// the levels whose operators hold at most `max_entries` entries each (default 20 000; < 0: unchanged).  The order of a
// row's sum differs from the per-level kernels' (rounding only).  on < 0: unchanged; returns the setting.
HYPRE_Int hypre_amd_SetSmallTail(HYPRE_Int on)
{
   if (on >= 0) { small_tail_on() = on != 0; }
   return small_tail_on();
}
// Test hook: the form the tail's image takes from now on — 0 every array in LDS, 1 the first level's operator in the lanes'
// registers, 2 streamed from where it lies; -1 the first of these that fits (default).  A form that does not fit a
// hierarchy moves the tail a level down.
HYPRE_Int hypre_amd_SetSmallTailForm(HYPRE_Int form)
{
   small_tail_forced_form() = (form >= 0 && form <= 2) ? form : -1;
   return hypre_error_flag;
}

HYPRE_Int hypre_BoomerAMGCycle(void *amg_vdata, hypre_ParVector **F_array, hypre_ParVector **U_array)
{
   hypre_ParAMGData *d = (hypre_ParAMGData *) amg_vdata;
   AmgPrivate *pv = (AmgPrivate *) d->amd_private;
   const int L = d->num_levels;
   hypre_ParCSRMatrix **A = d->A_array, **P = d->P_array;
   HYPRE_AMD_REQUIRE_DEVICE(A[0]->diag->memory_location, "hypre_BoomerAMGCycle(A)");
   HYPRE_AMD_REQUIRE_DEVICE(U_array[0]->local_vector->memory_location, "hypre_BoomerAMGCycle(u)");
   HYPRE_AMD_REQUIRE_DEVICE(F_array[0]->local_vector->memory_location, "hypre_BoomerAMGCycle(f)");
   hipStream_t s = stream();
   const int saved_sync = handle().sync_compute;
   handle().sync_compute = 0;
   const int saved_gs_threads = handle().gs_threads;
   handle().gs_threads = pv->emulated_threads;
   const bool saved_fp32 = handle().fp32_values;
   handle().fp32_values = pv->mixed_precision;

   // alternate solution buffers (allocated on the first cycle, reused afterwards)
   if ((int) pv->u_alt.size() != L) { pv->u_alt.assign((size_t) L, nullptr); pv->u_alt_len.assign((size_t) L, 0); }
   std::vector<LevelVec> lv((size_t) L);
   for (int l = 0; l < L; l++)
   {
      const int n = A[l]->diag->num_rows;
      if (pv->u_alt_len[(size_t) l] < n)
      {
         if (pv->u_alt[(size_t) l]) { hypre_Free(pv->u_alt[(size_t) l], HYPRE_MEMORY_DEVICE); }
         pv->u_alt[(size_t) l] = hypre_TAlloc(double, (size_t) std::max(n, 1), HYPRE_MEMORY_DEVICE);
         pv->u_alt_len[(size_t) l] = n;
      }
      lv[(size_t) l].home = U_array[l]->local_vector->data;
      lv[(size_t) l].alt = pv->u_alt[(size_t) l];
      lv[(size_t) l].cur = lv[(size_t) l].home;
   }
   std::vector<int> lev_counter((size_t) L, 0), zeros((size_t) L, 0);
   // presmoothed[l]: the restriction into level l also wrote the result of the level's first sweep from zero, u = (w f) ./ d
   std::vector<int> presmoothed((size_t) L, 0);
   zeros[0] = U_array[0]->all_zeros;
   lev_counter[0] = 1;
   for (int k = 1; k < L; k++) { lev_counter[(size_t) k] = d->fcycle ? 1 : d->cycle_type; }
   int fcycle_lev = L - 2, level = 0, cycle_param = 1, err = 0;
   bool not_finished = true;
   double *vtemp = d->Vtemp->local_vector->data;
   double *ztemp = d->Ztemp->local_vector->data;
   const bool old_version = d->grid_relax_points != nullptr;
   double cycle_op_count = d->cycle_op_count;

   HYPRE_Int cycle_nprocs = 1;
   hypre_MPI_Comm_size(A[0]->comm, &cycle_nprocs);
   // ---- the smallest levels in one kernel of one workgroup (tail_kernels.hip): one rank, a plain V(1,1) cycle, Jacobi or
   // l1-Jacobi over all points with the smoother diagonals at hand, a direct solve on the coarsest level ----
   int st = -1;
   {
      const int t1 = d->grid_relax_type[1], t2 = d->grid_relax_type[2], t3 = d->grid_relax_type[3];
      const bool ok = small_tail_on() && pv->small_tail_nnz > 0 && cycle_nprocs == 1 && d->cycle_type == 1 && !d->fcycle && !old_version &&
                      L >= 3 && (t1 == 7 || t1 == 18 || t1 == 11 || t1 == 12) && (t2 == 7 || t2 == 18 || t2 == 11 || t2 == 12) &&
                      !(pv->replica && (t1 == 11 || t1 == 12 || t2 == 11 || t2 == 12)) && is_ge_type(t3) && d->relax_order == 0 &&
                      d->num_grid_sweeps[1] == 1 && d->num_grid_sweeps[2] == 1 && d->num_grid_sweeps[3] == 1 && !pv->tail;
      if (ok)
      {
         if (pv->small_tail_level == -2)
         {
            int first = L - 1;
            while (first - 1 >= 1 && A[first - 1]->diag->num_nonzeros <= pv->small_tail_nnz && L - (first - 1) <= SMALL_TAIL_MAX_LEVELS) { first--; }
            pv->small_tail_level = (first <= L - 2) ? first : -1;
         }
         st = pv->small_tail_level;
         for (int l = st; st >= 0 && l < L; l++)
         {
            const bool last = l == L - 1;
            const bool here = A[l]->diag->memory_location == HYPRE_MEMORY_DEVICE && A[l]->offd->num_nonzeros == 0 &&
                              (last || (d->l1_norms[l] && d->l1_norms[l]->data && P[l] && P[l]->diagT &&
                                        P[l]->diagT->memory_location == HYPRE_MEMORY_DEVICE && P[l]->offd->num_nonzeros == 0));
            if (!here) { st = -1; }
         }
         if (st >= 0)
         {
            if (!d->gs_setup) { hypre_GaussElimSetup(d, L - 1, t3); }
            ensure_coarse_factors(d);
            if (pv->coarse_n != A[L - 1]->diag->num_rows || pv->coarse_n > SMALL_TAIL_MAX_COARSE || pv->coarse_n < 1) { st = -1; }
         }
         // the levels' arrays as the kernel holds them in LDS; a tail that does not fit starts a level further down
         auto image_of = [&](int lvl)
         {
            // (the form that worked last time first: with an unchanged hierarchy that is a comparison of two numbers)
            const int forced = small_tail_forced_form();
            if (forced >= 0) { pv->tail_outside = forced; return build_small_tail_image(d, pv, lvl, s, forced); }
            if (pv->tail_outside >= 0 && build_small_tail_image(d, pv, lvl, s, pv->tail_outside)) { return true; }
            for (int form = 0; form < 3; form++)
            {
               if (build_small_tail_image(d, pv, lvl, s, form)) { pv->tail_outside = form; return true; }
            }
            pv->tail_outside = -1;
            return false;
         };
         while (st >= 0 && !image_of(st))
         {
            st = (st + 1 <= L - 2) ? st + 1 : -1;
            pv->small_tail_level = st;
         }
      }
   }
   pv->small_tail_used = st;
   // ---- coarse tail as a HIP graph (single rank, plain V-cycle, smoothers whose launches do not depend on host state
   // that changes from cycle to cycle) ----
   int gl = -1;
   {
      auto graphable = [&](int t) { return is_jacobi_type(t) || is_ge_type(t) || t == 11 || t == 12 || t == 16 || t == 21 || t == 22; };
      static const int graph_env = [] { const char *e = getenv("HYPRE_AMD_CYCLE_GRAPH"); return e ? atoi(e) : 1; }();
      const bool ok = graph_env && pv->graph_rows > 0 && cycle_nprocs == 1 && d->cycle_type == 1 && !d->fcycle && !old_version &&
                      L >= 3 && graphable(d->grid_relax_type[1]) && graphable(d->grid_relax_type[2]) && graphable(d->grid_relax_type[3]);
      if (ok)
      {
         if (pv->graph_level < 0)
         {
            for (int l = 1; l < L - 1; l++) { if (A[l]->global_num_rows <= (HYPRE_BigInt) pv->graph_rows) { pv->graph_level = l; break; } }
         }
         gl = pv->graph_level;
         if (gl >= L - 1) { gl = -1; }
      }
      if (gl < 0 && pv->graph_state) { pv->drop_graph(); }
      if (gl >= 0)
      {
         const unsigned long long sig = tail_signature(d, pv, gl, F_array, U_array);
         if (pv->graph_state && sig != pv->graph_sig) { pv->drop_graph(); }
         pv->graph_sig = sig;
      }
   }
   bool arrived_down = false, capturing = false;
   double op_count_at_capture = 0.0, csr_at_capture = 0.0, stream_at_capture = 0.0;

   while (not_finished)
   {
      const int n = A[level]->diag->num_rows;
      hypre_amd_CommSetTag(level);             // (diagnosis: exchanges from here on belong to this level)
      // the tail: replay, or record on the second visit
      bool replayed = false;
      if (gl >= 0 && level == gl && arrived_down)
      {
         if (pv->graph_state == 2)
         {
            HIP_CHECK(hipGraphLaunch(pv->graph_exec, s));
            for (int l = gl; l < L; l++) { lv[(size_t) l].cur = pv->graph_cur[(size_t) (l - gl)]; zeros[(size_t) l] = 0; lev_counter[(size_t) l] = -1; }
            cycle_op_count += pv->graph_op_count;
            account_bytes(pv->graph_bytes_csr, pv->graph_bytes_stream);      // what the recorded launches accounted for
            replayed = true;
         }
         else if (pv->graph_state == 1)
         {
            if (hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) == hipSuccess) { capturing = true; op_count_at_capture = cycle_op_count; csr_at_capture = handle().bytes_csr; stream_at_capture = handle().bytes_stream; }
            else { (void) hipGetLastError(); pv->graph_state = 0; gl = -1; }
         }
      }
      if (st >= 0 && level == st && arrived_down && !replayed)
      {
         // everything from here down and back up to this level's post-smoothing sweep: one launch
         SmallTailArgs ta = pv->tail_args;
         double csr_bytes = 0.0;
         for (int l = st; l < L; l++)
         {
            ta.lv[l - st].w = d->relax_weight[l];
            if (l < L - 1)
            {
               cycle_op_count += 2.0 * A[l]->d_num_nonzeros;
               csr_bytes += 2.0 * (12.0 * A[l]->diag->num_nonzeros + 36.0 * A[l]->diag->num_rows) +
                            2.0 * (12.0 * P[l]->diag->num_nonzeros + 20.0 * P[l]->diag->num_rows);
            }
            else { cycle_op_count += A[l]->d_num_nonzeros; }
         }
         ta.f_in = F_array[st]->local_vector->data;
         ta.u_io = lv[(size_t) st].home;
         {
            const int t1 = d->grid_relax_type[1], t2 = d->grid_relax_type[2];
            ta.kind_down = (t1 == 11 || t1 == 12) ? 1 : 0; ta.inner_down = t1 == 12 ? 2 : 1;
            ta.kind_up = (t2 == 11 || t2 == 12) ? 1 : 0; ta.inner_up = t2 == 12 ? 2 : 1;
         }
         ta.first_presmoothed = presmoothed[(size_t) st];
         presmoothed[(size_t) st] = 0;
         ta.round32 = handle().fp32_values ? 1 : 0;
         account_bytes(csr_bytes);
         launch_small_tail(ta, s);
         for (int l = st; l < L; l++) { lv[(size_t) l].cur = lv[(size_t) l].home; zeros[(size_t) l] = 0; lev_counter[(size_t) l] = -1; }
         replayed = true;
      }
      arrived_down = false;
      int num_sweep, relax_type;
      if (L > 1) { num_sweep = d->num_grid_sweeps[cycle_param]; relax_type = d->grid_relax_type[cycle_param]; }
      else
      {
         num_sweep = d->num_grid_sweeps[0];
         relax_type = d->user_relax_type;
         if (relax_type == -1) { relax_type = 6; }
      }
      const int *cf = d->CF_marker_array[level] ? d->CF_marker_array[level]->data : nullptr;
      const double *l1 = d->l1_norms[level] ? d->l1_norms[level]->data : nullptr;
      const double *fd = F_array[level]->local_vector->data;
      LevelVec &u = lv[(size_t) level];
      const double w = d->relax_weight[level];

      // replicated tail: from this level down the V-cycle runs on this rank's own copy of the operators
      // (one all-reduce of the right-hand side instead of four halo exchanges per level)
      const bool tail_here = pv->tail != nullptr && level == pv->tail_level;
      if (tail_here)
      {
         err = run_replicated_tail(d, pv, fd, u.home, &cycle_op_count);
         u.cur = u.home;
         zeros[(size_t) level] = 0;
         if (err) { break; }
      }
      for (int j = 0; j < ((tail_here || replayed) ? 0 : num_sweep); j++)
      {
         int relax_points = 0, relax_local = d->relax_order;
         if (L == 1 && d->max_levels > 1) { relax_points = 0; relax_local = 0; }
         else if (old_version) { relax_points = d->grid_relax_points[cycle_param][j]; }
         // par_cycle.c:413-430: the reference's ("VERY sloppy") operation count behind the printed cycle complexity
         if (old_version && level < L - 1)
         {
            if (relax_points == 1) { cycle_op_count += A[level + 1]->d_num_nonzeros; }
            else if (relax_points == -1) { cycle_op_count += A[level]->d_num_nonzeros - A[level + 1]->d_num_nonzeros; }
         }
         else { cycle_op_count += A[level]->d_num_nonzeros; }
         if (is_ge_type(relax_type))
         {
            // the dense solve reads F[level] and writes the home vector
            hypre_GaussElimSolve(d, level, relax_type);
            u.cur = u.home;
            zeros[(size_t) level] = 0;
         }
         else if (is_jacobi_type(relax_type))
         {
            // point sets of this sweep: all points, or C then F / F then C
            int pts[2] = {relax_points, 0}, npass = 1;
            if (!old_version && relax_local == 1 && cycle_param < 3)
            {
               npass = 2;
               pts[0] = cycle_param < 2 ? 1 : -1; pts[1] = -pts[0];
            }
            const double *dg = l1;
            if (relax_type == 0 || !dg)
            {
               double *dd = pv->level_diag(level, n);
               launch_diag_first(A[level]->diag->i, A[level]->diag->data, dd, n, s);
               dg = dd;
            }
            for (int pss = 0; pss < npass; pss++)
            {
               if (zeros[(size_t) level] && relax_type != 0)
               {
                  if (presmoothed[(size_t) level]) { presmoothed[(size_t) level] = 0; }       // the restriction's epilogue did it
                  else { launch_scaled_div(w, fd, dg, u.cur, pts[pss] ? cf : nullptr, pts[pss], (size_t) n, s); }
               }
               else
               {
                  dev_jacobi_sweep(A[level], fd, cf, pts[pss], w, dg, u.cur, u.other());
                  u.flip();
               }
               zeros[(size_t) level] = 0;
            }
         }
         else if (relax_type == 15)
         {
            // CG smoother (par_cycle.c:517-528, par_relax_more.c:464-493, setup par_amg_setup.c:3551-3567): num_sweep
            // iterations of unpreconditioned CG from the current iterate, once per relaxation call
            if (j == 0)
            {
               if (pv->cg_smoothers.size() < (size_t) L) { pv->cg_smoothers.resize((size_t) L, nullptr); }
               hypre_Vector uv; hypre_ParVector up;
               wrap(&uv, u.cur, n);
               wrap_par(&up, &uv, A[level]->comm, A[level]->global_num_rows);
               up.all_zeros = zeros[(size_t) level];
               HYPRE_Solver &cg = pv->cg_smoothers[(size_t) level];
               if (!cg)
               {
                  HYPRE_ParCSRPCGCreate(A[level]->comm, &cg);
                  HYPRE_PCGSetTwoNorm(cg, 1);
                  HYPRE_ParCSRPCGSetup(cg, A[level], F_array[level], &up);
               }
               HYPRE_PCGSetMaxIter(cg, num_sweep);
               HYPRE_PCGSetTol(cg, 0.0);
               HYPRE_ParCSRPCGSolve(cg, A[level], F_array[level], &up);
               zeros[(size_t) level] = 0;
            }
         }
         else if (relax_type == 17)
         {
            // FCF-Jacobi (par_cycle.c:539-556, par_relax_interface.c:83-117): weighted Jacobi on the F, the C and again
            // the F points; one plain Jacobi sweep on the coarsest level, which has no C/F splitting
            static const int fcf[3] = {-1, 1, -1};
            const int npass = (level == L - 1) ? 1 : 3;
            double *dd = pv->level_diag(level, n);
            launch_diag_first(A[level]->diag->i, A[level]->diag->data, dd, n, s);
            for (int pss = 0; pss < npass; pss++)
            {
               dev_jacobi_sweep(A[level], fd, cf, npass == 1 ? 0 : fcf[pss], w, dd, u.cur, u.other());
               u.flip();
            }
            zeros[(size_t) level] = 0;
         }
         else if (relax_type == 16)
         {
            // Chebyshev polynomial smoothing (par_cycle.c:529-537), in place on the current buffer
            if (!d->cheby_coefs || !d->cheby_coefs[level] || !d->Ptemp || !d->Rtemp)
            {
               hypre_error_w_msg(HYPRE_ERROR_GENERIC, "hypre_BoomerAMGCycle: relax 16 on a level the setup did not prepare for it");
               err = 1;
            }
            else
            {
               hypre_Vector uv; hypre_ParVector up;
               wrap(&uv, u.cur, n);
               wrap_par(&up, &uv, A[level]->comm, A[level]->global_num_rows);
               up.all_zeros = zeros[(size_t) level];
               hypre_ParVectorSetLocalSize(d->Vtemp, n);
               hypre_ParVectorSetLocalSize(d->Ztemp, n);
               hypre_ParVectorSetLocalSize(d->Ptemp, n);
               hypre_ParVectorSetLocalSize(d->Rtemp, n);
               hypre_ParCSRRelax_Cheby_Solve(A[level], F_array[level],
                                             d->cheby_ds && d->cheby_ds[level] ? d->cheby_ds[level]->data : nullptr,
                                             d->cheby_coefs[level], d->cheby_order, d->cheby_scale, d->cheby_variant,
                                             &up, d->Vtemp, d->Ztemp, d->Ptemp, d->Rtemp);
               zeros[(size_t) level] = 0;
               if (hypre_error_flag) { err = 1; }
            }
         }
         else
         {
            // in-place smoothers go through the public entry on a view of the current buffer
            hypre_Vector uv; hypre_ParVector up;
            wrap(&uv, u.cur, n);
            wrap_par(&up, &uv, A[level]->comm, A[level]->global_num_rows);
            up.all_zeros = zeros[(size_t) level];
            hypre_ParVectorSetLocalSize(d->Vtemp, n);
            hypre_ParVectorSetLocalSize(d->Ztemp, n);
            if (old_version) { err = hypre_BoomerAMGRelax(A[level], F_array[level], (HYPRE_Int *) cf, relax_type, relax_points, w, d->omega[level], (HYPRE_Real *) l1, &up, d->Vtemp, d->Ztemp); }
            else { err = hypre_BoomerAMGRelaxIF(A[level], F_array[level], (HYPRE_Int *) cf, relax_type, relax_local, cycle_param, w, d->omega[level], (HYPRE_Real *) l1, &up, d->Vtemp, d->Ztemp); }
            zeros[(size_t) level] = 0;
            if (hypre_error_flag) { err = 1; }
         }
         if (err) { break; }
      }
      if (err) { break; }

      if (!replayed) { --lev_counter[(size_t) level]; }
      if (!replayed && lev_counter[(size_t) level] >= 0 && level != L - 1 && !tail_here)
      {
         arrived_down = true;
         // descend: u_c = 0 ; r = f - A u ; f_c = P^T r
         const int fine = level, coarse = level + 1;
         LevelVec &uc = lv[(size_t) coarse];
         uc.cur = uc.home;
         const int nc = A[coarse]->diag->num_rows;
         // the zero-guess sweep on the coarse level overwrites every row it
         // relaxes; rows it skips (CF passes, zero diagonals never occur) must
         // read as zero, so the vector is cleared unless the first coarse
         // operation is known to overwrite all of it
         const int ctype = d->grid_relax_type[coarse == L - 1 ? 3 : 1];
         const bool overwrites_all = (is_jacobi_type(ctype) && ctype != 0 && !(d->relax_order == 1 && coarse != L - 1) &&
                                      !old_version && d->num_grid_sweeps[coarse == L - 1 ? 3 : 1] > 0) || is_ge_type(ctype);
         if (!overwrites_all) { launch_set(uc.cur, 0.0, (size_t) nc, s); }
         zeros[(size_t) coarse] = 1;
         dev_par_matvec(-1.0, A[fine], u.cur, 1.0, fd, vtemp);
         // One rank, the coarse level starts with a Jacobi-type sweep from zero over all its points (u_c = w f_c ./ d_c, its
         // smoother diagonal at hand): the restriction's epilogue writes that as well — f_c is in its registers — instead of a
         // kernel of its own reading f_c and d_c again (one launch and 8 bytes per coarse row less on every level below the
         // finest; same bits: (w f) / d either way).  Several ranks add the neighbours' contributions to f_c afterwards: not there.
         const int fuse_on = cycle_fusion();
         presmoothed[(size_t) coarse] = 0;
         bool restricted = false;
         if (fuse_on && cycle_nprocs == 1 && overwrites_all && is_jacobi_type(ctype) && d->l1_norms[coarse] && d->l1_norms[coarse]->data &&
             P[fine]->diagT && P[fine]->diagT->memory_location == HYPRE_MEMORY_DEVICE && P[fine]->offd->num_cols == 0)
         {
            restricted = spmv_with_scaled_quotient(P[fine]->diagT, vtemp, F_array[coarse]->local_vector->data, d->relax_weight[coarse],
                                                   d->l1_norms[coarse]->data, uc.cur);
            if (restricted) { presmoothed[(size_t) coarse] = 1; F_array[coarse]->all_zeros = 0; }
         }
         if (!restricted) { dev_par_matvecT(1.0, P[fine], vtemp, 0.0, F_array[coarse]->local_vector->data); }
         ++level;
         lev_counter[(size_t) level] = std::max(lev_counter[(size_t) level], (int) d->cycle_type);
         cycle_param = (level == L - 1) ? 3 : 1;
      }
      else if (level != 0)
      {
         if (capturing && level == gl)
         {
            // the sub-cycle is over: everything enqueued since the tail was entered becomes the graph
            capturing = false;
            hipGraph_t g = nullptr;
            if (hipStreamEndCapture(s, &g) == hipSuccess && g)
            {
               hipGraphExec_t ex = nullptr;
               if (hipGraphInstantiate(&ex, g, nullptr, nullptr, 0) == hipSuccess)
               {
                  pv->graph = g; pv->graph_exec = ex; pv->graph_state = 2;
                  pv->graph_cur.assign((size_t) (L - gl), nullptr);
                  for (int l = gl; l < L; l++) { pv->graph_cur[(size_t) (l - gl)] = lv[(size_t) l].cur; }
                  pv->graph_op_count = cycle_op_count - op_count_at_capture;
                  pv->graph_bytes_csr = handle().bytes_csr - csr_at_capture;
                  pv->graph_bytes_stream = handle().bytes_stream - stream_at_capture;
                  size_t nn = 0;
                  if (hipGraphGetNodes(g, nullptr, &nn) == hipSuccess) { pv->graph_launches = (int) nn; }
                  HIP_CHECK(hipGraphLaunch(ex, s));       // the capture recorded the work, it did not run it
               }
               else { (void) hipGetLastError(); (void) hipGraphDestroy(g); pv->graph_state = 0; err = 1; hypre_error_w_msg(HYPRE_ERROR_GENERIC, "hypre_BoomerAMGCycle: the coarse-tail graph could not be instantiated"); }
            }
            else { (void) hipGetLastError(); pv->graph_state = 0; err = 1; hypre_error_w_msg(HYPRE_ERROR_GENERIC, "hypre_BoomerAMGCycle: capturing the coarse tail failed"); }
            if (err) { break; }
         }
         // ascend: u_f += P u_c, written to whichever buffer lets the
         // post-smoothing sweeps end in the level's home vector
         const int fine = level - 1;
         LevelVec &uf = lv[(size_t) fine];
         int flips = 0;
         {
            const int t = d->grid_relax_type[2];
            if (is_jacobi_type(t))
            {
               const int passes = (!old_version && d->relax_order == 1) ? 2 : 1;
               flips = d->num_grid_sweeps[2] * passes;
            }
         }
         double *target = (flips % 2 == 0) ? uf.home : uf.alt;
         dev_par_matvec(1.0, P[fine], u.cur, 1.0, uf.cur, target);
         uf.cur = target;
         zeros[(size_t) fine] = 0;
         --level;
         cycle_param = 2;
         if (d->fcycle && fcycle_lev == level)
         {
            lev_counter[(size_t) level] = std::max(lev_counter[(size_t) level], 1);
            fcycle_lev--;
         }
      }
      else
      {
         not_finished = false;
      }
   }
   if (capturing)
   {
      // left the loop (an error) while recording: close the capture so that the stream is usable again
      hipGraph_t g = nullptr;
      (void) hipStreamEndCapture(s, &g);
      if (g) { (void) hipGraphDestroy(g); }
      (void) hipGetLastError();
      pv->graph_state = 0;
   }
   else if (gl >= 0 && pv->graph_state == 0 && !err) { pv->graph_state = 1; }      // warmed up: record at the next cycle
   // the caller's vector must hold the result
   if (lv[0].cur != lv[0].home)
   {
      launch_copy(lv[0].home, lv[0].cur, (size_t) A[0]->diag->num_rows, s);
   }
   U_array[0]->all_zeros = 0;
   d->cycle_op_count = cycle_op_count;
   (void) ztemp;
   handle().fp32_values = saved_fp32;
   hypre_amd_CommSetTag(-1);
   handle().gs_threads = saved_gs_threads;
   handle().sync_compute = saved_sync;
   maybe_sync();
   return err;
}

// ===========================================================================
// outer solve loop (par_amg_solve.c:22-424)
// ===========================================================================
HYPRE_Int hypre_BoomerAMGSolve(void *amg_vdata, hypre_ParCSRMatrix *A, hypre_ParVector *f, hypre_ParVector *u)
{
   hypre_ParAMGData *d = (hypre_ParAMGData *) amg_vdata;
   if (!d) { hypre_error_in_arg(1); return hypre_error_flag; }
   if (d->num_levels < 1 || !d->A_array)
   {
      hypre_error_w_msg(HYPRE_ERROR_GENERIC, "hypre_BoomerAMGSolve: setup has not been run");
      return hypre_error_flag;
   }
   HYPRE_AMD_REQUIRE_DEVICE(A->diag->memory_location, "hypre_BoomerAMGSolve(A)");
   HYPRE_AMD_REQUIRE_DEVICE(f->local_vector->memory_location, "hypre_BoomerAMGSolve(f)");
   HYPRE_AMD_REQUIRE_DEVICE(u->local_vector->memory_location, "hypre_BoomerAMGSolve(u)");
   const HYPRE_Real tol = d->tol;
   const int saved_sync = handle().sync_compute;
   handle().sync_compute = 0;
   // stand-alone solver (more than one cycle allowed): the fine-level plans are verified against the caller's arrays once
   // per solve (one pass over the CSR arrays; a Krylov method that applies one cycle per iteration does it itself, once per
   // solve — HYPRE_ParCSRPCGSolve / GMRESSolve)
   if (d->max_iter > 1) { verify_par_plans(A); }
   d->A_array[0] = A; d->F_array[0] = f; d->U_array[0] = u;
   hypre_ParVector *Vtemp = d->Vtemp;
   hypre_ParVectorSetLocalSize(Vtemp, A->diag->num_rows);

   HYPRE_Real resid_nrm = 1.0, resid_nrm_init = 1.0, rhs_norm = 0.0, relative_resid = 1.0, old_resid, conv_factor = 0.0;
   HYPRE_Int cycle_count = 0;
   if (d->print_level > 1 || d->logging > 1 || tol > 0.)
   {
      hypre_ParVectorCopy(f, Vtemp);
      // r0 = A u - f: alpha = +1, beta = -1 exactly as the reference forms it (:170-189)
      if (tol > 0) { hypre_ParCSRMatrixMatvec(1.0, A, u, -1.0, Vtemp); }
      resid_nrm = std::sqrt(hypre_ParVectorInnerProd(Vtemp, Vtemp));
      HYPRE_Real ieee_check = 0.;
      if (resid_nrm != 0.) { ieee_check = resid_nrm / resid_nrm; }
      if (ieee_check != ieee_check)
      {
         if (d->print_level > 0)
         {
            fprintf(stderr, "ERROR -- hypre_BoomerAMGSolve: INFs and/or NaNs detected in input.\n");
         }
         handle().sync_compute = saved_sync;
         hypre_error(HYPRE_ERROR_GENERIC);
         return hypre_error_flag;
      }
      resid_nrm_init = resid_nrm;
      if (d->converge_type == 0)
      {
         rhs_norm = std::sqrt(hypre_ParVectorInnerProd(f, f));
         relative_resid = rhs_norm ? resid_nrm_init / rhs_norm : resid_nrm_init;
      }
      else { relative_resid = 1.0; }
   }
   else { relative_resid = 1.; }

   HYPRE_Int Solve_err_flag = 0;
   while ((relative_resid >= tol || cycle_count < d->min_iter) && cycle_count < d->max_iter)
   {
      d->cycle_op_count = 0;
      AmgPrivate *pv = (AmgPrivate *) d->amd_private;
      if (pv->mixed_precision && !u->all_zeros)
      {
         // Mixed precision from an iterate that is not known to be zero: the cycle's operators carry fp32-rounded
         // values, so it is applied to the error equation — residual with the exact operator in fp64, cycle from a zero
         // guess, correction added in fp64 ("fp32 SpMV / fp64 residual").  From a zero guess (every preconditioner
         // call) the residual is f itself and the cycle runs on (f, u) directly.
         const HYPRE_Int nloc = A->diag->num_rows;
         if (pv->mp_r && pv->mp_r->local_vector->size != nloc)
         {
            hypre_ParVectorDestroy(pv->mp_r); hypre_ParVectorDestroy(pv->mp_e);
            pv->mp_r = pv->mp_e = nullptr;
         }
         if (!pv->mp_r)
         {
            pv->mp_r = hypre_ParVectorCreate(A->comm, A->global_num_rows, A->row_starts);
            pv->mp_e = hypre_ParVectorCreate(A->comm, A->global_num_rows, A->row_starts);
            hypre_ParVectorInitialize_v2(pv->mp_r, HYPRE_MEMORY_DEVICE);
            hypre_ParVectorInitialize_v2(pv->mp_e, HYPRE_MEMORY_DEVICE);
         }
         hypre_ParCSRMatrixMatvecOutOfPlace(-1.0, A, u, 1.0, f, pv->mp_r);
         hypre_ParVectorSetZeros(pv->mp_e);
         d->F_array[0] = pv->mp_r; d->U_array[0] = pv->mp_e;
         hypre_BoomerAMGCycle(d, d->F_array, d->U_array);
         d->F_array[0] = f; d->U_array[0] = u;
         hypre_ParVectorAxpy(1.0, pv->mp_e, u);
      }
      else { hypre_BoomerAMGCycle(d, d->F_array, d->U_array); }
      if (d->print_level > 1 || d->logging > 1 || tol > 0.)
      {
         old_resid = resid_nrm;
         hypre_ParVectorSetLocalSize(Vtemp, A->diag->num_rows);
         hypre_ParCSRMatrixMatvecOutOfPlace(1.0, A, u, -1.0, f, Vtemp);
         resid_nrm = std::sqrt(hypre_ParVectorInnerProd(Vtemp, Vtemp));
         conv_factor = old_resid ? resid_nrm / old_resid : resid_nrm;
         if (d->converge_type == 0) { relative_resid = rhs_norm ? resid_nrm / rhs_norm : resid_nrm; }
         else { relative_resid = resid_nrm / resid_nrm_init; }
         d->rel_resid_norm = relative_resid;
      }
      ++cycle_count;
      d->num_iterations = cycle_count;
      if (d->print_level > 1)
      {
         HYPRE_Int rank; hypre_MPI_Comm_rank(A->comm, &rank);
         if (rank == 0) { printf("    Cycle %2d   %e    %f     %e \n", cycle_count, resid_nrm, conv_factor, relative_resid); }
      }
   }
   if (cycle_count == d->max_iter && tol > 0.)
   {
      Solve_err_flag = 1;
      hypre_error(HYPRE_ERROR_CONV);
   }
   (void) Solve_err_flag;
   handle().sync_compute = saved_sync;
   maybe_sync();
   return hypre_error_flag;
}

// ===========================================================================
// PCG (krylov/pcg.c:318-1000 with the defaults flex = rel_change = 0,
// recompute_residual = 0, stop_crit = 0, atolf = 0, cf_tol = 0)
// ===========================================================================
struct hypre_amd_PCGData
{
   hypre_Solver base;
   MPI_Comm comm;
   HYPRE_Real tol = 1e-6, a_tol = 0.0;
   HYPRE_Int max_iter = 1000, two_norm = 0, flex = 0;
   HYPRE_PtrToSolverFcn precond = nullptr, precond_setup = nullptr;
   HYPRE_Solver precond_data = nullptr;
   hypre_ParVector *p = nullptr, *s = nullptr, *r = nullptr, *r_old = nullptr;
   double *d_rr = nullptr;       // device scalar of this solver: <r,r> of the fused update survives the preconditioner call
                                 // (which may itself run a PCG: the CG smoother)
   HYPRE_Int num_iterations = 0, converged = 0;
   HYPRE_Real rel_residual_norm = 0.0;
};

HYPRE_Int HYPRE_ParCSRPCGCreate(MPI_Comm comm, HYPRE_Solver *solver)
{
   hypre_amd_PCGData *d = new hypre_amd_PCGData();
   memset(&d->base, 0, sizeof(d->base));
   d->comm = comm;
   *solver = (HYPRE_Solver) d;
   return hypre_error_flag;
}
HYPRE_Int HYPRE_ParCSRPCGDestroy(HYPRE_Solver solver)
{
   hypre_amd_PCGData *d = (hypre_amd_PCGData *) solver;
   if (!d) { return hypre_error_flag; }
   hypre_ParVectorDestroy(d->p); hypre_ParVectorDestroy(d->s); hypre_ParVectorDestroy(d->r); hypre_ParVectorDestroy(d->r_old);
   if (d->d_rr) { hypre_Free(d->d_rr, HYPRE_MEMORY_DEVICE); }
   delete d;
   return hypre_error_flag;
}
HYPRE_Int HYPRE_PCGSetTol(HYPRE_Solver s, HYPRE_Real v) { ((hypre_amd_PCGData *) s)->tol = v; return hypre_error_flag; }
HYPRE_Int HYPRE_PCGSetAbsoluteTol(HYPRE_Solver s, HYPRE_Real v) { ((hypre_amd_PCGData *) s)->a_tol = v; return hypre_error_flag; }
HYPRE_Int HYPRE_PCGSetMaxIter(HYPRE_Solver s, HYPRE_Int v) { ((hypre_amd_PCGData *) s)->max_iter = v; return hypre_error_flag; }
HYPRE_Int HYPRE_PCGSetTwoNorm(HYPRE_Solver s, HYPRE_Int v) { ((hypre_amd_PCGData *) s)->two_norm = v; return hypre_error_flag; }
// pcg.c:339-345: Polak-Ribiere beta instead of Fletcher-Reeves, for preconditioners that change between iterations
HYPRE_Int HYPRE_PCGSetFlex(HYPRE_Solver s, HYPRE_Int v) { ((hypre_amd_PCGData *) s)->flex = v; return hypre_error_flag; }
HYPRE_Int HYPRE_PCGSetPrecond(HYPRE_Solver s, HYPRE_PtrToSolverFcn precond, HYPRE_PtrToSolverFcn precond_setup,
                              HYPRE_Solver precond_solver)
{
   hypre_amd_PCGData *d = (hypre_amd_PCGData *) s;
   d->precond = precond; d->precond_setup = precond_setup; d->precond_data = precond_solver;
   return hypre_error_flag;
}
// options of krylov/HYPRE_pcg.c an application sets routinely: logging and printing are accepted and inert (this PCG
// keeps no residual history and prints nothing); the relative-change and recomputed-residual variants are not built
HYPRE_Int HYPRE_PCGSetLogging(HYPRE_Solver s, HYPRE_Int v) { (void) s; (void) v; return hypre_error_flag; }
HYPRE_Int HYPRE_PCGSetPrintLevel(HYPRE_Solver s, HYPRE_Int v) { (void) s; (void) v; return hypre_error_flag; }
HYPRE_Int HYPRE_PCGSetRelChange(HYPRE_Solver s, HYPRE_Int v)
{
   (void) s;
   if (v != 0) { hypre_error_in_arg(2); hypre_error_w_msg(HYPRE_ERROR_GENERIC, "HYPRE_PCGSetRelChange: the relative-change stopping test is not built (0 only)"); }
   return hypre_error_flag;
}
HYPRE_Int HYPRE_PCGSetRecomputeResidual(HYPRE_Solver s, HYPRE_Int v)
{
   (void) s;
   if (v != 0) { hypre_error_in_arg(2); hypre_error_w_msg(HYPRE_ERROR_GENERIC, "HYPRE_PCGSetRecomputeResidual: 0 only"); }
   return hypre_error_flag;
}
HYPRE_Int HYPRE_PCGGetNumIterations(HYPRE_Solver s, HYPRE_Int *v) { *v = ((hypre_amd_PCGData *) s)->num_iterations; return hypre_error_flag; }
HYPRE_Int HYPRE_PCGGetFinalRelativeResidualNorm(HYPRE_Solver s, HYPRE_Real *v) { *v = ((hypre_amd_PCGData *) s)->rel_residual_norm; return hypre_error_flag; }

HYPRE_Int HYPRE_ParCSRPCGSetup(HYPRE_Solver solver, HYPRE_ParCSRMatrix A, HYPRE_ParVector b, HYPRE_ParVector x)
{
   hypre_amd_PCGData *d = (hypre_amd_PCGData *) solver;
   hypre_ParVectorDestroy(d->p); hypre_ParVectorDestroy(d->s); hypre_ParVectorDestroy(d->r); hypre_ParVectorDestroy(d->r_old);
   const HYPRE_MemoryLocation loc = x->local_vector->memory_location;
   // work vectors shaped like x (krylov/pcg.c:226-260 CreateVector(x)): as many columns as a multivector x has (`ij -nc N`)
   const HYPRE_Int nv = x->local_vector->num_vectors;
   auto mk = [&]() { hypre_ParVector *v = hypre_ParMultiVectorCreate(A->comm, A->global_num_rows, A->row_starts, nv); hypre_ParVectorInitialize_v2(v, loc); return v; };
   d->p = mk(); d->s = mk(); d->r = mk();
   d->r_old = d->flex ? mk() : nullptr;
   if (!d->d_rr && loc == HYPRE_MEMORY_DEVICE) { d->d_rr = hypre_TAlloc(double, 2, HYPRE_MEMORY_DEVICE); }
   if (d->precond_setup) { d->precond_setup(d->precond_data, A, b, x); }
   return hypre_error_flag;
}

HYPRE_Int HYPRE_ParCSRPCGSolve(HYPRE_Solver solver, HYPRE_ParCSRMatrix A, HYPRE_ParVector b, HYPRE_ParVector x)
{
   hypre_amd_PCGData *d = (hypre_amd_PCGData *) solver;
   hypre_ParVector *p = d->p, *s = d->s, *r = d->r;
   const HYPRE_Real r_tol = d->tol, a_tol = d->a_tol;
   HYPRE_Real alpha, beta, gamma, gamma_old, bi_prod, eps, sdotp, i_prod = 0.0, i_prod_0 = 0.0, delta = 0.0;
   HYPRE_Int i = 0;
   d->converged = 0;
   if (d->flex && !d->r_old)
   {
      hypre_error_w_msg(HYPRE_ERROR_GENERIC, "HYPRE_ParCSRPCGSolve: call HYPRE_ParCSRPCGSetup after HYPRE_PCGSetFlex");
      return hypre_error_flag;
   }
   // multivectors (`ij -nc N`, test/TEST_ij/vector.jobs): the reference's PCG works on whatever its vector functions take,
   // and hypre_ParVector's take every column — one Krylov iteration over the columns together.  The fused updates below
   // run over the columns' common storage, which has to be one piece.
   const HYPRE_Int nvec = x->local_vector->num_vectors;
   for (hypre_ParVector *v : {b, x, p, s, r})
   {
      const hypre_Vector *l = v->local_vector;
      if (l->num_vectors != nvec || (nvec > 1 && (l->idxstride != 1 || l->vecstride != l->size)))
      {
         hypre_error_w_msg(HYPRE_ERROR_GENERIC, "HYPRE_ParCSRPCGSolve: b, x and the work vectors of HYPRE_ParCSRPCGSetup must have the same number of "
                                                "columns, stored one after the other");
         return hypre_error_flag;
      }
   }
   const size_t vec_len = (size_t) r->local_vector->size * (size_t) nvec;
   const int saved_sync = handle().sync_compute;
   handle().sync_compute = 0;
   verify_par_plans(A);                    // a solve never starts from a plan its matrix has moved away from
   auto precond = [&](hypre_ParVector *rhs, hypre_ParVector *sol)
   {
      hypre_ParVectorSetZeros(sol);       // ClearVector: all_zeros = 1 (par_vector.c:335-340)
      if (d->precond) { d->precond(d->precond_data, A, rhs, sol); }
      else { hypre_ParVectorCopy(rhs, sol); }
   };
   if (d->two_norm) { bi_prod = hypre_ParVectorInnerProd(b, b); }
   else { precond(b, p); bi_prod = hypre_ParVectorInnerProd(p, b); }
   HYPRE_Real ieee_check = 0.;
   if (bi_prod != 0.) { ieee_check = bi_prod / bi_prod; }
   if (ieee_check != ieee_check) { handle().sync_compute = saved_sync; hypre_error(HYPRE_ERROR_GENERIC); return hypre_error_flag; }
   eps = r_tol * r_tol;
   if (bi_prod > 0.0) { eps = std::max(r_tol * r_tol, a_tol * a_tol / bi_prod); }
   else
   {
      hypre_ParVectorCopy(b, x);
      d->num_iterations = 0; d->rel_residual_norm = 0.0;
      handle().sync_compute = saved_sync;
      return hypre_error_flag;
   }
   hypre_ParVectorCopy(b, r);
   hypre_ParCSRMatrixMatvec(-1.0, A, x, 1.0, r);
   precond(r, p);
   gamma = hypre_ParVectorInnerProd(r, p);
   if (gamma != 0.) { ieee_check = gamma / gamma; }
   if (ieee_check != ieee_check) { handle().sync_compute = saved_sync; hypre_error(HYPRE_ERROR_GENERIC); return hypre_error_flag; }
   i_prod_0 = d->two_norm ? hypre_ParVectorInnerProd(r, r) : gamma;
   while ((i + 1) <= d->max_iter)
   {
      i++;
      hypre_ParCSRMatrixMatvec(1.0, A, p, 0.0, s);
      sdotp = hypre_ParVectorInnerProd(s, p);
      if (sdotp == 0.0) { hypre_error_w_msg(HYPRE_ERROR_CONV, "Zero sdotp value in PCG"); if (i == 1) { i_prod = i_prod_0; } break; }
      alpha = gamma / sdotp;
      if (alpha <= 0.0) { hypre_error_w_msg(HYPRE_ERROR_CONV, "Negative or zero alpha value in PCG"); if (i == 1) { i_prod = i_prod_0; } break; }
      gamma_old = gamma;
      // x += alpha p ; r -= alpha s ; <r,r> of the new residual: one pass (pcg.c:716-760 makes
      // three vector calls of it; element values and the norm's summation order are unchanged)
      if (d->flex) { hypre_ParVectorCopy(r, d->r_old); }          // pcg.c:636-639
      double *d_rr = d->d_rr;
      launch_pcg_update(alpha, -alpha, p->local_vector->data, s->local_vector->data, x->local_vector->data,
                        r->local_vector->data, vec_len, d_rr, stream());
      x->all_zeros = 0; r->all_zeros = 0;
      precond(r, s);
      // gamma = <r, s> and, for the two-norm test, <r, r> (left on the device by the fused update): ONE all-reduce of
      // two values and one read-back per iteration (pcg.c:716-760 reduces them separately)
      launch_dot(r->local_vector->data, s->local_vector->data, vec_len, d_rr + 1, stream());
      {
         double sums[2];
         dev_global_sums(r->comm, d_rr, 2, sums);
         gamma = sums[1];
         i_prod = d->two_norm ? sums[0] : gamma;
      }
      if (d->flex) { delta = gamma - hypre_ParVectorInnerProd(d->r_old, s); }      // pcg.c:720-723
      if (i_prod / bi_prod < eps) { d->converged = 1; break; }
      if (gamma <= 0.0) { hypre_error_w_msg(HYPRE_ERROR_CONV, "Negative or zero gamma value in PCG"); break; }
      beta = (d->flex ? delta : gamma) / gamma_old;                 // pcg.c:957-965
      // p = beta p + s in one pass (Scale then Axpy in the reference)
      launch_pcg_direction(beta, s->local_vector->data, p->local_vector->data, vec_len, stream());
   }
   if (i >= d->max_iter && (i_prod / bi_prod) >= eps && eps > 0)
   {
      hypre_error_w_msg(HYPRE_ERROR_CONV, "Reached max iterations in PCG before convergence");
   }
   d->num_iterations = i;
   d->rel_residual_norm = std::sqrt(i_prod / bi_prod);
   handle().sync_compute = saved_sync;
   maybe_sync();
   return hypre_error_flag;
}

}  // extern "C"
