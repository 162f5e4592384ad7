// hypre_amd — right-preconditioned restarted GMRES, the second Krylov caller of the
// BoomerAMG solve phase (`ij -solver 3`).
//
// Reference: krylov/gmres.c:274-1000 (hypre_GMRESSolve) with parcsr_ls/HYPRE_parcsr_gmres.c and
// krylov/HYPRE_gmres.c for the entry points; defaults of hypre_GMRESCreate (gmres.c:68-110):
// k_dim 5, tol 1e-6, a_tol 0, min_iter 0, max_iter 1000, rel_change 0, skip_real_r_check 0.
// Modified Gram-Schmidt, Givens rotations on the host, the true residual is recomputed before
// convergence is accepted; the relative-change and convergence-factor exits of the reference
// (rel_change, cf_tol) are not carried.  Vectors stay on the device; every inner product ends in
// one scalar read-back, as in the reference.
#include "amg_internal.hpp"
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

using namespace hamd;

struct hypre_amd_GMRESData
{
   hypre_Solver base;
   MPI_Comm comm;
   HYPRE_Int k_dim = 5, min_iter = 0, max_iter = 1000, skip_real_r_check = 0;
   HYPRE_Real tol = 1e-6, a_tol = 0.0;
   HYPRE_PtrToSolverFcn precond = nullptr, precond_setup = nullptr;
   HYPRE_Solver precond_data = nullptr;
   hypre_ParVector *r = nullptr, *w = nullptr;
   std::vector<hypre_ParVector *> p;
   HYPRE_Int num_iterations = 0, converged = 0;
   HYPRE_Real rel_residual_norm = 0.0;
};

namespace {
void free_vectors(hypre_amd_GMRESData *d)
{
   hypre_ParVectorDestroy(d->r); hypre_ParVectorDestroy(d->w);
   d->r = d->w = nullptr;
   for (hypre_ParVector *v : d->p) { hypre_ParVectorDestroy(v); }
   d->p.clear();
}
}  // namespace

extern "C" {

HYPRE_Int HYPRE_ParCSRGMRESCreate(MPI_Comm comm, HYPRE_Solver *solver)
{
   hypre_amd_GMRESData *d = new hypre_amd_GMRESData();
   memset(&d->base, 0, sizeof(d->base));
   d->comm = comm;
   *solver = (HYPRE_Solver) d;
   return hypre_error_flag;
}

HYPRE_Int HYPRE_ParCSRGMRESDestroy(HYPRE_Solver solver)
{
   hypre_amd_GMRESData *d = (hypre_amd_GMRESData *) solver;
   if (!d) { return hypre_error_flag; }
   free_vectors(d);
   delete d;
   return hypre_error_flag;
}

HYPRE_Int HYPRE_GMRESSetKDim(HYPRE_Solver s, HYPRE_Int v)
{
   if (v < 1) { hypre_error_in_arg(2); return hypre_error_flag; }
   ((hypre_amd_GMRESData *) s)->k_dim = v;
   return hypre_error_flag;
}
HYPRE_Int HYPRE_GMRESSetTol(HYPRE_Solver s, HYPRE_Real v) { ((hypre_amd_GMRESData *) s)->tol = v; return hypre_error_flag; }
HYPRE_Int HYPRE_GMRESSetAbsoluteTol(HYPRE_Solver s, HYPRE_Real v) { ((hypre_amd_GMRESData *) s)->a_tol = v; return hypre_error_flag; }
HYPRE_Int HYPRE_GMRESSetMinIter(HYPRE_Solver s, HYPRE_Int v) { ((hypre_amd_GMRESData *) s)->min_iter = v; return hypre_error_flag; }
HYPRE_Int HYPRE_GMRESSetMaxIter(HYPRE_Solver s, HYPRE_Int v) { ((hypre_amd_GMRESData *) s)->max_iter = v; return hypre_error_flag; }
HYPRE_Int HYPRE_GMRESSetSkipRealResidualCheck(HYPRE_Solver s, HYPRE_Int v) { ((hypre_amd_GMRESData *) s)->skip_real_r_check = v; return hypre_error_flag; }
HYPRE_Int HYPRE_GMRESSetPrecond(HYPRE_Solver s, HYPRE_PtrToSolverFcn precond, HYPRE_PtrToSolverFcn precond_setup,
                                HYPRE_Solver precond_solver)
{
   hypre_amd_GMRESData *d = (hypre_amd_GMRESData *) s;
   d->precond = precond; d->precond_setup = precond_setup; d->precond_data = precond_solver;
   return hypre_error_flag;
}
HYPRE_Int HYPRE_GMRESSetLogging(HYPRE_Solver s, HYPRE_Int v) { (void) s; (void) v; return hypre_error_flag; }       // accepted, inert
HYPRE_Int HYPRE_GMRESSetPrintLevel(HYPRE_Solver s, HYPRE_Int v) { (void) s; (void) v; return hypre_error_flag; }
HYPRE_Int HYPRE_GMRESSetRelChange(HYPRE_Solver s, HYPRE_Int v)
{
   (void) s;
   if (v != 0) { hypre_error_in_arg(2); hypre_error_w_msg(HYPRE_ERROR_GENERIC, "HYPRE_GMRESSetRelChange: the relative-change stopping test is not built (0 only)"); }
   return hypre_error_flag;
}
HYPRE_Int HYPRE_GMRESGetNumIterations(HYPRE_Solver s, HYPRE_Int *v) { *v = ((hypre_amd_GMRESData *) s)->num_iterations; return hypre_error_flag; }
HYPRE_Int HYPRE_GMRESGetFinalRelativeResidualNorm(HYPRE_Solver s, HYPRE_Real *v) { *v = ((hypre_amd_GMRESData *) s)->rel_residual_norm; return hypre_error_flag; }
HYPRE_Int HYPRE_GMRESGetConverged(HYPRE_Solver s, HYPRE_Int *v) { *v = ((hypre_amd_GMRESData *) s)->converged; return hypre_error_flag; }

HYPRE_Int HYPRE_ParCSRGMRESSetup(HYPRE_Solver solver, HYPRE_ParCSRMatrix A, HYPRE_ParVector b, HYPRE_ParVector x)
{
   hypre_amd_GMRESData *d = (hypre_amd_GMRESData *) solver;
   free_vectors(d);
   const HYPRE_MemoryLocation loc = x->local_vector->memory_location;
   // work vectors shaped like x (gmres.c:204-215 CreateVector / CreateVectorArray): the columns of a multivector x
   const HYPRE_Int nv = x->local_vector->num_vectors;
   auto mk = [&]() { hypre_ParVector *v = hypre_ParMultiVectorCreate(A->comm, A->global_num_rows, A->row_starts, nv); hypre_ParVectorInitialize_v2(v, loc); return v; };
   d->r = mk(); d->w = mk();
   for (HYPRE_Int i = 0; i <= d->k_dim; i++) { d->p.push_back(mk()); }
   if (d->precond_setup) { d->precond_setup(d->precond_data, A, b, x); }
   return hypre_error_flag;
}

HYPRE_Int HYPRE_ParCSRGMRESSolve(HYPRE_Solver solver, HYPRE_ParCSRMatrix A, HYPRE_ParVector b, HYPRE_ParVector x)
{
   hypre_amd_GMRESData *d = (hypre_amd_GMRESData *) solver;
   if ((HYPRE_Int) d->p.size() != d->k_dim + 1)
   {
      hypre_error_w_msg(HYPRE_ERROR_GENERIC, "HYPRE_ParCSRGMRESSolve: call HYPRE_ParCSRGMRESSetup after HYPRE_GMRESSetKDim");
      return hypre_error_flag;
   }
   const HYPRE_Int k_dim = d->k_dim, min_iter = d->min_iter, max_iter = d->max_iter;
   hypre_ParVector *r = d->r, *w = d->w;
   std::vector<hypre_ParVector *> &p = d->p;
   const HYPRE_Real epsmac = 1.e-16;
   std::vector<HYPRE_Real> rs((size_t) k_dim + 1, 0.0), c((size_t) k_dim, 0.0), s((size_t) k_dim, 0.0);
   std::vector<std::vector<HYPRE_Real>> hh((size_t) k_dim + 1, std::vector<HYPRE_Real>((size_t) k_dim, 0.0));
   HYPRE_Int i = 0, j, k, iter = 0;
   HYPRE_Real t, gamma, r_norm, b_norm, den_norm, epsilon, ieee_check = 0., real_r_norm_old, real_r_norm_new;
   d->converged = 0;
   const int saved_sync = handle().sync_compute;
   handle().sync_compute = 0;
   verify_par_plans(A);                    // a solve never starts from a plan its matrix has moved away from
   auto leave = [&]() { handle().sync_compute = saved_sync; maybe_sync(); return hypre_error_flag; };
   auto precond = [&](hypre_ParVector *rhs, hypre_ParVector *sol)
   {
      hypre_ParVectorSetZeros(sol);       // ClearVector (gmres.c:569): all_zeros = 1
      if (d->precond) { d->precond(d->precond_data, A, rhs, sol); }
      else { hypre_ParVectorCopy(rhs, sol); }
   };
   auto norm = [&](hypre_ParVector *v) { return std::sqrt(hypre_ParVectorInnerProd(v, v)); };

   hypre_ParVectorCopy(b, p[0]);
   hypre_ParCSRMatrixMatvec(-1.0, A, x, 1.0, p[0]);
   b_norm = norm(b);
   real_r_norm_old = b_norm;
   if (b_norm != 0.) { ieee_check = b_norm / b_norm; }
   if (ieee_check != ieee_check) { hypre_error(HYPRE_ERROR_GENERIC); return leave(); }
   r_norm = norm(p[0]);
   if (r_norm != 0.) { ieee_check = r_norm / r_norm; }
   if (ieee_check != ieee_check) { hypre_error(HYPRE_ERROR_GENERIC); return leave(); }
   den_norm = (b_norm > 0.0) ? b_norm : r_norm;
   epsilon = std::max(d->a_tol, d->tol * den_norm);

   while (iter < max_iter)
   {
      rs[0] = r_norm;
      if (r_norm == 0.0)
      {
         d->num_iterations = iter;            // gmres.c:500-512 returns here, rel_residual_norm untouched
         return leave();
      }
      if (r_norm <= epsilon && iter >= min_iter)
      {
         hypre_ParVectorCopy(b, r);
         hypre_ParCSRMatrixMatvec(-1.0, A, x, 1.0, r);
         r_norm = norm(r);
         if (r_norm <= epsilon) { break; }
      }
      t = 1.0 / r_norm;
      hypre_ParVectorScale(t, p[0]);
      i = 0;
      while (i < k_dim && iter < max_iter)
      {
         i++;
         iter++;
         precond(p[(size_t) i - 1], r);
         hypre_ParCSRMatrixMatvec(1.0, A, r, 0.0, p[(size_t) i]);
         for (j = 0; j < i; j++)
         {
            hh[(size_t) j][(size_t) i - 1] = hypre_ParVectorInnerProd(p[(size_t) j], p[(size_t) i]);
            hypre_ParVectorAxpy(-hh[(size_t) j][(size_t) i - 1], p[(size_t) j], p[(size_t) i]);
         }
         t = norm(p[(size_t) i]);
         hh[(size_t) i][(size_t) i - 1] = t;
         if (t != 0.0) { t = 1.0 / t; hypre_ParVectorScale(t, p[(size_t) i]); }
         for (j = 1; j < i; j++)
         {
            t = hh[(size_t) j - 1][(size_t) i - 1];
            hh[(size_t) j - 1][(size_t) i - 1] = s[(size_t) j - 1] * hh[(size_t) j][(size_t) i - 1] + c[(size_t) j - 1] * t;
            hh[(size_t) j][(size_t) i - 1] = -s[(size_t) j - 1] * t + c[(size_t) j - 1] * hh[(size_t) j][(size_t) i - 1];
         }
         t = hh[(size_t) i][(size_t) i - 1] * hh[(size_t) i][(size_t) i - 1];
         t += hh[(size_t) i - 1][(size_t) i - 1] * hh[(size_t) i - 1][(size_t) i - 1];
         gamma = std::sqrt(t);
         if (gamma == 0.0) { gamma = epsmac; }
         c[(size_t) i - 1] = hh[(size_t) i - 1][(size_t) i - 1] / gamma;
         s[(size_t) i - 1] = hh[(size_t) i][(size_t) i - 1] / gamma;
         rs[(size_t) i] = -hh[(size_t) i][(size_t) i - 1] * rs[(size_t) i - 1];
         rs[(size_t) i] /= gamma;
         rs[(size_t) i - 1] = c[(size_t) i - 1] * rs[(size_t) i - 1];
         hh[(size_t) i - 1][(size_t) i - 1] = s[(size_t) i - 1] * hh[(size_t) i][(size_t) i - 1] + c[(size_t) i - 1] * hh[(size_t) i - 1][(size_t) i - 1];
         r_norm = std::fabs(rs[(size_t) i]);
         if (r_norm <= epsilon && iter >= min_iter) { break; }
      }
      // upper triangular solve, then the update through the preconditioner
      rs[(size_t) i - 1] = rs[(size_t) i - 1] / hh[(size_t) i - 1][(size_t) i - 1];
      for (k = i - 2; k >= 0; k--)
      {
         t = 0.0;
         for (j = k + 1; j < i; j++) { t -= hh[(size_t) k][(size_t) j] * rs[(size_t) j]; }
         t += rs[(size_t) k];
         rs[(size_t) k] = t / hh[(size_t) k][(size_t) k];
      }
      hypre_ParVectorCopy(p[(size_t) i - 1], w);
      hypre_ParVectorScale(rs[(size_t) i - 1], w);
      for (j = i - 2; j >= 0; j--) { hypre_ParVectorAxpy(rs[(size_t) j], p[(size_t) j], w); }
      precond(w, r);
      hypre_ParVectorAxpy(1.0, r, x);
      x->all_zeros = 0;
      if (r_norm <= epsilon && iter >= min_iter)
      {
         if (d->skip_real_r_check) { d->converged = 1; break; }
         hypre_ParVectorCopy(b, r);
         hypre_ParCSRMatrixMatvec(-1.0, A, x, 1.0, r);
         real_r_norm_new = r_norm = norm(r);
         if (r_norm <= epsilon) { d->converged = 1; break; }
         if (real_r_norm_new >= real_r_norm_old) { d->converged = 1; break; }     // gmres.c:903-912
         hypre_ParVectorCopy(r, p[0]);
         i = 0;
         real_r_norm_old = real_r_norm_new;
      }
      // residual vector of the restart (gmres.c:930-950)
      for (j = i; j > 0; j--)
      {
         rs[(size_t) j - 1] = -s[(size_t) j - 1] * rs[(size_t) j];
         rs[(size_t) j] = c[(size_t) j - 1] * rs[(size_t) j];
      }
      if (i) { hypre_ParVectorAxpy(rs[(size_t) i] - 1.0, p[(size_t) i], p[(size_t) i]); }
      for (j = i - 1; j > 0; j--) { hypre_ParVectorAxpy(rs[(size_t) j], p[(size_t) j], p[(size_t) i]); }
      if (i)
      {
         hypre_ParVectorAxpy(rs[0] - 1.0, p[0], p[0]);
         hypre_ParVectorAxpy(1.0, p[(size_t) i], p[0]);
      }
   }
   d->num_iterations = iter;
   d->rel_residual_norm = (b_norm > 0.0) ? r_norm / b_norm : r_norm;
   if (iter >= max_iter && r_norm > epsilon && epsilon > 0) { hypre_error(HYPRE_ERROR_CONV); }
   return leave();
}

}  // extern "C"
