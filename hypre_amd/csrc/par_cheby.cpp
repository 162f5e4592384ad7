// hypre_amd — Chebyshev polynomial smoothing (relax 16).
//
// Reference counterparts:
//   parcsr_ls/par_relax_more.c:34-135    hypre_ParCSRMaxEigEstimateHost   (Gershgorin discs)
//   parcsr_ls/par_relax_more.c:203-400   hypre_ParCSRMaxEigEstimateCGHost (Lanczos tridiagonal of k CG steps)
//   parcsr_ls/par_relax_more.c:506-750   the EISPACK tql1 transcription used for its eigenvalues
//   parcsr_ls/par_cheby.c:57-222         hypre_ParCSRRelax_Cheby_Setup
//   parcsr_ls/par_cheby.c:224-446, par_cheby_device.c:119-294   hypre_ParCSRRelax_Cheby_Solve
//
// The spectrum estimates and the coefficients are setup-time host work, like the
// rest of the AMG setup.  The smoother itself runs on the device: order-1 SpMVs
// of the tiled kernel family plus two fused elementwise passes per application
// (the reference: up to four elementwise passes per polynomial term).
#include "amg_internal.hpp"
#include <algorithm>
#include <cmath>
#include <vector>

using namespace hamd;

namespace {

// utilities/random.c: Park-Miller minimal standard generator
struct ParkMiller
{
   HYPRE_Int seed = 13579;
   explicit ParkMiller(HYPRE_Int s)
   {
      const HYPRE_Int m = 2147483647;
      seed = s < 1 ? 1 : (s >= m ? m - 1 : s);
   }
   double next()
   {
      const HYPRE_Int a = 16807, m = 2147483647, q = 127773, r = 2836;
      const HYPRE_Int high = seed / q, low = seed % q;
      const HYPRE_Int test = a * low - r * high;
      seed = test > 0 ? test : test + m;
      return (double) seed / (double) m;
   }
};

// y = A x on host arrays (setup-time only: the Lanczos steps of the spectrum estimate)
void host_par_matvec(hypre_ParCSRMatrix *A, const double *x, double *y, std::vector<double> &ghost,
                     std::vector<double> &sendbuf)
{
   hypre_CSRMatrix *D = A->diag, *O = A->offd;
   const HYPRE_Int n = D->num_rows;
   hypre_ParCSRCommPkg *pkg = A->comm_pkg;
   if (O->num_cols > 0 && pkg)
   {
      const HYPRE_Int tot = pkg->send_map_starts[pkg->num_sends];
      sendbuf.resize((size_t) std::max(tot, 1));
      ghost.resize((size_t) O->num_cols);
      for (HYPRE_Int k = 0; k < tot; k++) { sendbuf[(size_t) k] = x[pkg->send_map_elmts[k]]; }
      hypre_ParCSRCommHandle *h = hypre_ParCSRCommHandleCreate(1, pkg, sendbuf.data(), ghost.data());
      hypre_ParCSRCommHandleDestroy(h);
   }
   else if (pkg && pkg->num_sends > 0)
   {
      // a rank without ghost columns still serves its neighbours
      const HYPRE_Int tot = pkg->send_map_starts[pkg->num_sends];
      sendbuf.resize((size_t) std::max(tot, 1));
      for (HYPRE_Int k = 0; k < tot; k++) { sendbuf[(size_t) k] = x[pkg->send_map_elmts[k]]; }
      hypre_ParCSRCommHandle *h = hypre_ParCSRCommHandleCreate(1, pkg, sendbuf.data(), ghost.data());
      hypre_ParCSRCommHandleDestroy(h);
   }
#pragma omp parallel for schedule(static)
   for (HYPRE_Int i = 0; i < n; i++)
   {
      double s = 0.0;
      for (HYPRE_Int k = D->i[i]; k < D->i[i + 1]; k++) { s += D->data[k] * x[D->j[k]]; }
      if (O->num_cols > 0) { for (HYPRE_Int k = O->i[i]; k < O->i[i + 1]; k++) { s += O->data[k] * ghost[(size_t) O->j[k]]; } }
      y[i] = s;
   }
}

double host_dot(MPI_Comm comm, const double *x, const double *y, HYPRE_Int n)
{
   double r = 0.0;
   for (HYPRE_Int i = 0; i < n; i++) { r += y[i] * x[i]; }       // seq_mv/vector.c:1070-1090 order
   const hypre_amd_CommOps *o = comm_ops(comm);
   if (o && o->size > 1) { o->allreduce_sum(o->ctx, &r, 1, 0, nullptr); }
   return r;
}

// Eigenvalues of the symmetric tridiagonal matrix with diagonal d[0..n) and
// couplings e[i] between i-1 and i (e[0] unused), ascending in d on return.
// Implicit QL with Wilkinson shifts (the algorithm EISPACK's tql1 implements).
int tridiag_eigenvalues(int n, double *d, double *e)
{
   if (n <= 0) { return 0; }
   for (int i = 1; i < n; i++) { e[i - 1] = e[i]; }
   e[n - 1] = 0.0;
   for (int l = 0; l < n; l++)
   {
      int iter = 0, m;
      do
      {
         for (m = l; m < n - 1; m++)
         {
            const double dd = std::fabs(d[m]) + std::fabs(d[m + 1]);
            if (std::fabs(e[m]) <= 2.220446049250313e-16 * dd) { break; }
         }
         if (m != l)
         {
            if (iter++ == 60) { return l + 1; }
            double g = (d[l + 1] - d[l]) / (2.0 * e[l]);
            double r = std::hypot(g, 1.0);
            g = d[m] - d[l] + e[l] / (g + (g >= 0.0 ? std::fabs(r) : -std::fabs(r)));
            double s = 1.0, c = 1.0, p = 0.0;
            int i;
            for (i = m - 1; i >= l; i--)
            {
               double f = s * e[i];
               const double b = c * e[i];
               r = std::hypot(f, g);
               e[i + 1] = r;
               if (r == 0.0)
               {
                  d[i + 1] -= p;
                  e[m] = 0.0;
                  break;
               }
               s = f / r;
               c = g / r;
               g = d[i + 1] - p;
               r = (d[i] - g) * s + 2.0 * c * b;
               p = s * r;
               d[i + 1] = g + p;
               g = c * r - b;
            }
            if (r == 0.0 && i >= l) { continue; }
            d[l] -= p;
            e[l] = g;
            e[m] = 0.0;
         }
      } while (m != l);
   }
   std::sort(d, d + n);
   return 0;
}

}  // namespace

extern "C" {

HYPRE_Int hypre_ParCSRMaxEigEstimate(hypre_ParCSRMatrix *A, HYPRE_Int scale, HYPRE_Real *max_eig, HYPRE_Real *min_eig)
{
   hypre_CSRMatrix *D = A->diag, *O = A->offd;
   if (D->memory_location != HYPRE_MEMORY_HOST)
   {
      hypre_error_w_msg(HYPRE_ERROR_GENERIC, "hypre_ParCSRMaxEigEstimate: setup-time routine, expects host matrices");
      return hypre_error_flag;
   }
   const HYPRE_Int n = D->num_rows;
   double e_max = 0.0, e_min = 0.0;
   for (HYPRE_Int i = 0; i < n; i++)
   {
      double a_ii = 0.0, r_i = 0.0;
      for (HYPRE_Int j = D->i[i]; j < D->i[i + 1]; j++)
      {
         if (D->j[j] == i) { a_ii = D->data[j]; } else { r_i += std::fabs(D->data[j]); }
      }
      for (HYPRE_Int j = O->i[i]; j < O->i[i + 1]; j++) { r_i += std::fabs(O->data[j]); }
      double lower = a_ii - r_i, upper = a_ii + r_i;
      if (scale == 1) { lower /= std::fabs(a_ii); upper /= std::fabs(a_ii); }
      if (i) { e_max = std::max(e_max, upper); e_min = std::min(e_min, lower); }
      else { e_max = upper; e_min = lower; }
   }
   // max over the ranks of (-e_min, e_max)
   const hypre_amd_CommOps *o = comm_ops(A->comm);
   if (o && o->size > 1)
   {
      double mine[2] = {-e_min, e_max};
      std::vector<double> all((size_t) 2 * (size_t) o->size);
      o->allgather(o->ctx, mine, all.data(), sizeof(mine));
      double m0 = all[0], m1 = all[1];
      for (int r = 1; r < o->size; r++) { m0 = std::max(m0, all[(size_t) 2 * r]); m1 = std::max(m1, all[(size_t) 2 * r + 1]); }
      e_min = -m0; e_max = m1;
   }
   if (std::fabs(e_min) > std::fabs(e_max)) { *min_eig = e_min; *max_eig = std::min(0.0, e_max); }
   else { *min_eig = std::max(e_min, 0.0); *max_eig = e_max; }
   return hypre_error_flag;
}

HYPRE_Int hypre_ParCSRMaxEigEstimateCG(hypre_ParCSRMatrix *A, HYPRE_Int scale, HYPRE_Int max_iter,
                                       HYPRE_Real *max_eig, HYPRE_Real *min_eig)
{
   hypre_CSRMatrix *D = A->diag;
   if (D->memory_location != HYPRE_MEMORY_HOST)
   {
      hypre_error_w_msg(HYPRE_ERROR_GENERIC, "hypre_ParCSRMaxEigEstimateCG: setup-time routine, expects host matrices");
      return hypre_error_flag;
   }
   const HYPRE_Int n = D->num_rows;
   if (A->global_num_rows < (HYPRE_BigInt) max_iter) { max_iter = (HYPRE_Int) A->global_num_rows; }
   if (!A->comm_pkg) { hypre_MatvecCommPkgCreate(A); }
   const size_t nn = (size_t) std::max(n, 1);
   std::vector<double> p(nn, 0.0), s(nn, 0.0), r(nn, 0.0), ds(nn, 1.0), u(nn, 0.0), ghost, sendbuf;
   std::vector<double> tridiag((size_t) max_iter + 1, 0.0), trioffd((size_t) max_iter + 1, 0.0);

   // residual = random vector: seed 1 scaled by (rank + 1)   (par_vector.c:347-359)
   {
      HYPRE_Int my_id;
      hypre_MPI_Comm_rank(A->comm, &my_id);
      ParkMiller rng(1 * (my_id + 1));
      for (HYPRE_Int i = 0; i < n; i++) { r[(size_t) i] = 2.0 * rng.next() - 1.0; }
   }
   if (scale)
   {
      // csr_matop.c:1919-1970 type 4: 1/sqrt(|a_ii|)
      for (HYPRE_Int i = 0; i < n; i++)
      {
         double d_i = 0.0;
         for (HYPRE_Int j = D->i[i]; j < D->i[i + 1]; j++)
         {
            if (D->j[j] == i)
            {
               if (D->data[j] == 0.0) { hypre_error_w_msg(HYPRE_ERROR_GENERIC, "Zero diagonal found!"); }
               else { d_i = 1.0 / std::sqrt(std::fabs(D->data[j])); }
               break;
            }
         }
         ds[(size_t) i] = d_i;
      }
   }
   double gamma = host_dot(A->comm, r.data(), p.data(), n), gamma_old, beta = 1.0;
   HYPRE_Int i = 0;
   while (i < max_iter)
   {
      s = r;                                           // s = C r with C = I
      gamma_old = gamma;
      gamma = host_dot(A->comm, r.data(), s.data(), n);
      if (gamma < 2.220446049250313e-16) { break; }
      if (i == 0) { beta = 1.0; p = s; }
      else
      {
         beta = gamma / gamma_old;
         for (HYPRE_Int j = 0; j < n; j++) { p[(size_t) j] = s[(size_t) j] + beta * p[(size_t) j]; }
      }
      if (scale)
      {
         for (HYPRE_Int j = 0; j < n; j++) { u[(size_t) j] = ds[(size_t) j] * p[(size_t) j]; }
         host_par_matvec(A, u.data(), s.data(), ghost, sendbuf);
         for (HYPRE_Int j = 0; j < n; j++) { s[(size_t) j] = ds[(size_t) j] * s[(size_t) j]; }
      }
      else { host_par_matvec(A, p.data(), s.data(), ghost, sendbuf); }
      const double sdotp = host_dot(A->comm, s.data(), p.data(), n);
      const double alpha = gamma / sdotp;
      const double alphainv = 1.0 / alpha;
      tridiag[(size_t) i + 1] = alphainv;
      tridiag[(size_t) i] *= beta;
      tridiag[(size_t) i] += alphainv;
      trioffd[(size_t) i + 1] = alphainv;
      trioffd[(size_t) i] *= std::sqrt(beta);
      for (HYPRE_Int j = 0; j < n; j++) { r[(size_t) j] += -alpha * s[(size_t) j]; }
      i++;
   }
   if (i == 0) { *max_eig = 0.0; *min_eig = 0.0; return hypre_error_flag; }
   tridiag_eigenvalues(i, tridiag.data(), trioffd.data());
   *max_eig = tridiag[(size_t) i - 1];
   *min_eig = tridiag[0];
   return hypre_error_flag;
}

HYPRE_Int hypre_ParCSRRelax_Cheby_Setup(hypre_ParCSRMatrix *A, HYPRE_Real max_eig, HYPRE_Real min_eig,
                                        HYPRE_Real fraction, HYPRE_Int order, HYPRE_Int scale, HYPRE_Int variant,
                                        HYPRE_Real **coefs_ptr, HYPRE_Real **ds_ptr)
{
   hypre_CSRMatrix *D = A->diag;
   if (D->memory_location != HYPRE_MEMORY_HOST)
   {
      hypre_error_w_msg(HYPRE_ERROR_GENERIC, "hypre_ParCSRRelax_Cheby_Setup: setup-time routine, expects host matrices");
      return hypre_error_flag;
   }
   order = std::min(std::max(order, 1), 4);
   HYPRE_Real *coefs = hypre_CTAlloc(HYPRE_Real, (size_t) order + 1, HYPRE_MEMORY_HOST);
   const int cheby_order = order - 1;               // degree of p in u += p(A) r
   double upper_bound, lower_bound;
   if (max_eig <= 0.0)
   {
      upper_bound = min_eig * 1.1;
      lower_bound = max_eig - (max_eig - upper_bound) * fraction;
   }
   else
   {
      upper_bound = max_eig * 1.1;
      lower_bound = (upper_bound - min_eig) * fraction + min_eig;
   }
   const double theta = (upper_bound + lower_bound) / 2, delta = (upper_bound - lower_bound) / 2;
   double den;
   if (variant == 1)
   {
      switch (cheby_order)
      {
         case 0: coefs[0] = 1.0 / theta; break;
         case 1:
            den = (theta * theta + delta * theta);
            coefs[0] = (delta + 2 * theta) / den;
            coefs[1] = -1.0 / den;
            break;
         case 2:
            den = 2 * delta * theta * theta - delta * delta * theta - std::pow(delta, 3) + 2 * std::pow(theta, 3);
            coefs[0] = (4 * delta * theta - std::pow(delta, 2) + 6 * std::pow(theta, 2)) / den;
            coefs[1] = -(2 * delta + 6 * theta) / den;
            coefs[2] = 2 / den;
            break;
         case 3:
            den = -4 * delta * std::pow(theta, 3) + 3 * std::pow(delta, 2) * std::pow(theta, 2) +
                  3 * std::pow(delta, 3) * theta - 4 * std::pow(theta, 4);
            coefs[0] = (6 * std::pow(delta, 2) * theta - 12 * delta * std::pow(theta, 2) + 3 * std::pow(delta, 3) -
                        16 * std::pow(theta, 3)) / den;
            coefs[1] = (12 * delta * theta - 3 * std::pow(delta, 2) + 24 * std::pow(theta, 2)) / den;
            coefs[2] = -(4 * delta + 16 * theta) / den;
            coefs[3] = 4 / den;
            break;
      }
   }
   else
   {
      switch (cheby_order)
      {
         case 0: coefs[0] = 1.0 / theta; break;
         case 1:
            den = delta * delta - 2 * theta * theta;
            coefs[0] = -4 * theta / den;
            coefs[1] = 2 / den;
            break;
         case 2:
            den = 3 * (delta * delta) * theta - 4 * (theta * theta * theta);
            coefs[0] = (3 * delta * delta - 12 * theta * theta) / den;
            coefs[1] = 12 * theta / den;
            coefs[2] = -4 / den;
            break;
         case 3:
            den = std::pow(delta, 4) - 8 * delta * delta * theta * theta + 8 * std::pow(theta, 4);
            coefs[0] = (32 * std::pow(theta, 3) - 16 * delta * delta * theta) / den;
            coefs[1] = (8 * delta * delta - 48 * theta * theta) / den;
            coefs[2] = 32 * theta / den;
            coefs[3] = -8 / den;
            break;
      }
   }
   *coefs_ptr = coefs;
   HYPRE_Real *ds = nullptr;
   if (scale)
   {
      const HYPRE_Int n = D->num_rows;
      ds = hypre_CTAlloc(HYPRE_Real, (size_t) std::max(n, 1), HYPRE_MEMORY_HOST);
      for (HYPRE_Int i = 0; i < n; i++)
      {
         double d_i = 0.0;
         for (HYPRE_Int j = D->i[i]; j < D->i[i + 1]; j++)
         {
            if (D->j[j] == i)
            {
               if (D->data[j] == 0.0) { hypre_error_w_msg(HYPRE_ERROR_GENERIC, "Zero diagonal found!"); }
               else { d_i = 1.0 / std::sqrt(std::fabs(D->data[j])); }
               break;
            }
         }
         ds[i] = d_i;
      }
   }
   *ds_ptr = ds;
   return hypre_error_flag;
}

// u += p(A) (f - A u), p from the coefficients above, optionally on D^-1/2 A D^-1/2.
//   r = ds .* (f - A u) ; orig = u ; u = c_k r
//   for i = k-1 .. 0 :  v = A (ds .* u) ; u = c_i r + ds .* v
//   u = orig + ds .* u
// The elementwise steps between the SpMVs are fused into one kernel each (the one after the
// residual also saves u and starts the Horner recurrence; the last one adds the correction).
HYPRE_Int hypre_ParCSRRelax_Cheby_Solve(hypre_ParCSRMatrix *A, hypre_ParVector *f, HYPRE_Real *ds_data,
                                        HYPRE_Real *coefs, HYPRE_Int order, HYPRE_Int scale, HYPRE_Int variant,
                                        hypre_ParVector *u, hypre_ParVector *v, hypre_ParVector *r,
                                        hypre_ParVector *orig_u_vec, hypre_ParVector *tmp_vec)
{
   (void) variant;
   HYPRE_AMD_REQUIRE_DEVICE(A->diag->memory_location, "hypre_ParCSRRelax_Cheby_Solve(A)");
   HYPRE_AMD_REQUIRE_DEVICE(u->local_vector->memory_location, "hypre_ParCSRRelax_Cheby_Solve(u)");
   HYPRE_AMD_REQUIRE_DEVICE(f->local_vector->memory_location, "hypre_ParCSRRelax_Cheby_Solve(f)");
   if (f->local_vector->num_vectors > 1)
   {
      hypre_error_w_msg(HYPRE_ERROR_GENERIC, "Requested relaxation type doesn't support multicomponent vectors");
      return hypre_error_flag;
   }
   const int n = A->diag->num_rows;
   if (!v || !r || !orig_u_vec || (scale && (!tmp_vec || !ds_data)) || !coefs)
   {
      hypre_error_w_msg(HYPRE_ERROR_GENERIC, "hypre_ParCSRRelax_Cheby_Solve: missing work vector, coefficients or scaling");
      return hypre_error_flag;
   }
   order = std::min(std::max(order, 1), 4);
   const int cheby_order = order - 1;
   hipStream_t s = stream();
   const int saved = handle().sync_compute;
   handle().sync_compute = 0;
   double *ud = u->local_vector->data, *vd = v->local_vector->data, *rd = r->local_vector->data;
   double *od = orig_u_vec->local_vector->data;
   const double *fd = f->local_vector->data;
   double *td = scale ? tmp_vec->local_vector->data : nullptr;
   const double *ds = scale ? ds_data : nullptr;

   // tmp (or r) = -A u ; then r = ds.*(f + tmp), orig = u, u = c_k r, tmp = ds.*u
   double *first = scale ? td : rd;
   // u known to be zero (all_zeros flag): -A u = 0 without touching the matrix, same bits as the product
   if (u->all_zeros) { launch_set(first, 0.0, (size_t) n, s); }
   else { dev_par_matvec(-1.0, A, ud, 0.0, first, first); }
   launch_cheby_start(fd, first, ds, coefs[cheby_order], cheby_order == 0, ud, od, rd, td, (size_t) n, s);
   for (int i = cheby_order - 1; i >= 0; i--)
   {
      // v = A (ds .* u)      (unscaled: v = A u)
      dev_par_matvec(1.0, A, scale ? td : ud, 0.0, vd, vd);
      launch_cheby_step(rd, vd, ds, od, coefs[i], i == 0, ud, td, (size_t) n, s);
   }
   u->all_zeros = 0;
   handle().sync_compute = saved;
   maybe_sync();
   return hypre_error_flag;
}

}  // extern "C"
