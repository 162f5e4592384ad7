// hypre_amd — synthetic problem generators: the inputs of the benchmark
// configurations.  Each rank (p,q,r) of a P x Q x R box
// decomposition assembles its own block in host memory; rows are numbered
// lexicographically inside the rank's box (x fastest) and boxes are
// concatenated in rank order, exactly as the reference driver does so that
// matrices, partitions and hence AMG hierarchies are comparable.
//
// Reference: parcsr_ls/par_laplace.c:15-375 (7-point), par_laplace_27pt.c,
// par_difconv.c; seq_mv/genpart.c:18-40 (partitioning).
#include "internal.hpp"
#include <algorithm>
#include <cmath>

namespace {

std::vector<HYPRE_BigInt> gen_part(HYPRE_BigInt length, HYPRE_Int nprocs)
{
   std::vector<HYPRE_BigInt> part((size_t) nprocs + 1, 0);
   const HYPRE_BigInt size = length / nprocs;
   const HYPRE_BigInt rest = length - size * nprocs;
   for (HYPRE_Int i = 0; i < nprocs; i++) { part[(size_t) i + 1] = part[(size_t) i] + size + (i < rest ? 1 : 0); }
   return part;
}

struct Box
{
   HYPRE_BigInt nx, ny, nz;
   std::vector<HYPRE_BigInt> xp, yp, zp;
   // global index of grid point (ix,iy,iz) that lives in the box of rank (p,q,r)
   HYPRE_BigInt map(HYPRE_BigInt ix, HYPRE_BigInt iy, HYPRE_BigInt iz, HYPRE_Int p, HYPRE_Int q, HYPRE_Int r) const
   {
      const HYPRE_BigInt nxl = xp[(size_t) p + 1] - xp[(size_t) p];
      const HYPRE_BigInt nyl = yp[(size_t) q + 1] - yp[(size_t) q];
      const HYPRE_BigInt nzl = zp[(size_t) r + 1] - zp[(size_t) r];
      HYPRE_BigInt g = zp[(size_t) r] * nx * ny + yp[(size_t) q] * nx * nzl + xp[(size_t) p] * (nyl * nzl);
      g += ((iz - zp[(size_t) r]) * nyl + (iy - yp[(size_t) q])) * nxl + (ix - xp[(size_t) p]);
      return g;
   }
};

// Generic stencil assembler.  `stencil` lists (dx,dy,dz,value-index) in the
// order entries are stored in each row; entry 0 must be the centre.
struct StencilPt { int dx, dy, dz, vi; };

// `point_values`, when given, fills value[] for the grid point whose row is being assembled (variable coefficients)
HYPRE_ParCSRMatrix assemble(MPI_Comm comm, HYPRE_BigInt nx, HYPRE_BigInt ny, HYPRE_BigInt nz,
                            HYPRE_Int P, HYPRE_Int Q, HYPRE_Int R, HYPRE_Int ip, HYPRE_Int iq, HYPRE_Int ir,
                            const std::vector<StencilPt> &stencil, HYPRE_Real *value,
                            void (*point_values)(void *, HYPRE_BigInt, HYPRE_BigInt, HYPRE_BigInt, HYPRE_Real *) = nullptr,
                            void *point_ctx = nullptr)
{
   Box bx;
   bx.nx = nx; bx.ny = ny; bx.nz = nz;
   bx.xp = gen_part(nx, P); bx.yp = gen_part(ny, Q); bx.zp = gen_part(nz, R);
   const HYPRE_BigInt x0 = bx.xp[(size_t) ip], x1 = bx.xp[(size_t) ip + 1];
   const HYPRE_BigInt y0 = bx.yp[(size_t) iq], y1 = bx.yp[(size_t) iq + 1];
   const HYPRE_BigInt z0 = bx.zp[(size_t) ir], z1 = bx.zp[(size_t) ir + 1];
   const HYPRE_Int nxl = (HYPRE_Int) (x1 - x0), nyl = (HYPRE_Int) (y1 - y0), nzl = (HYPRE_Int) (z1 - z0);
   const HYPRE_Int nloc = nxl * nyl * nzl;
   HYPRE_BigInt part[2];
   part[0] = z0 * nx * ny + (y0 * nx + x0 * nyl) * nzl;
   part[1] = part[0] + nloc;

   std::vector<HYPRE_Int> di((size_t) nloc + 1, 0), oi((size_t) nloc + 1, 0);
   std::vector<HYPRE_Int> dj;
   std::vector<HYPRE_Real> da, oa;
   std::vector<HYPRE_BigInt> obig;
   dj.reserve((size_t) nloc * stencil.size());
   da.reserve((size_t) nloc * stencil.size());

   HYPRE_Int row = 0;
   for (HYPRE_BigInt iz = z0; iz < z1; iz++)
      for (HYPRE_BigInt iy = y0; iy < y1; iy++)
         for (HYPRE_BigInt ix = x0; ix < x1; ix++)
         {
            if (point_values) { point_values(point_ctx, ix, iy, iz, value); }
            for (const StencilPt &s : stencil)
            {
               const HYPRE_BigInt jx = ix + s.dx, jy = iy + s.dy, jz = iz + s.dz;
               if (jx < 0 || jx >= nx || jy < 0 || jy >= ny || jz < 0 || jz >= nz) { continue; }
               const bool in = jx >= x0 && jx < x1 && jy >= y0 && jy < y1 && jz >= z0 && jz < z1;
               if (in)
               {
                  dj.push_back(row + s.dx + nxl * (s.dy + nyl * s.dz));
                  da.push_back(value[s.vi]);
               }
               else
               {
                  const HYPRE_Int jp = ip + (jx < x0 ? -1 : (jx >= x1 ? 1 : 0));
                  const HYPRE_Int jq = iq + (jy < y0 ? -1 : (jy >= y1 ? 1 : 0));
                  const HYPRE_Int jr = ir + (jz < z0 ? -1 : (jz >= z1 ? 1 : 0));
                  obig.push_back(bx.map(jx, jy, jz, jp, jq, jr));
                  oa.push_back(value[s.vi]);
               }
            }
            row++;
            di[(size_t) row] = (HYPRE_Int) dj.size();
            oi[(size_t) row] = (HYPRE_Int) obig.size();
         }

   // ghost columns: ascending global ids, compressed
   std::vector<HYPRE_BigInt> cmap(obig);
   std::sort(cmap.begin(), cmap.end());
   cmap.erase(std::unique(cmap.begin(), cmap.end()), cmap.end());
   std::vector<HYPRE_Int> oj(obig.size());
   for (size_t k = 0; k < obig.size(); k++)
   {
      oj[k] = (HYPRE_Int) (std::lower_bound(cmap.begin(), cmap.end(), obig[k]) - cmap.begin());
   }
   const HYPRE_BigInt gsize = nx * ny * nz;
   return hypre_amd_ParCSRMatrixFromArrays(comm, gsize, gsize, part, part, (HYPRE_Int) cmap.size(), cmap.data(),
                                           di.data(), dj.data(), da.data(), oi.data(), oj.data(), oa.data(),
                                           HYPRE_MEMORY_HOST);
}

}  // namespace

extern "C" {

// 7-point stencil; stored order per row = [centre, -z, -y, -x, +x, +y, +z]
// (par_laplace.c:199-300); value = {centre, x-coupling, y-coupling, z-coupling}
HYPRE_ParCSRMatrix GenerateLaplacian(MPI_Comm comm, HYPRE_BigInt nx, HYPRE_BigInt ny, HYPRE_BigInt nz,
                                     HYPRE_Int P, HYPRE_Int Q, HYPRE_Int R, HYPRE_Int p, HYPRE_Int q,
                                     HYPRE_Int r, HYPRE_Real *value)
{
   static const std::vector<StencilPt> st = {
      {0, 0, 0, 0}, {0, 0, -1, 3}, {0, -1, 0, 2}, {-1, 0, 0, 1}, {1, 0, 0, 1}, {0, 1, 0, 2}, {0, 0, 1, 3}};
   return assemble(comm, nx, ny, nz, P, Q, R, p, q, r, st, value);
}

// 27-point stencil; stored order = centre first, then the remaining 26 points
// in lexicographic (z, y, x) order (par_laplace_27pt.c); value = {centre, off}
HYPRE_ParCSRMatrix GenerateLaplacian27pt(MPI_Comm comm, HYPRE_BigInt nx, HYPRE_BigInt ny, HYPRE_BigInt nz,
                                         HYPRE_Int P, HYPRE_Int Q, HYPRE_Int R, HYPRE_Int p, HYPRE_Int q,
                                         HYPRE_Int r, HYPRE_Real *value)
{
   std::vector<StencilPt> st;
   st.push_back({0, 0, 0, 0});
   for (int dz = -1; dz <= 1; dz++)
      for (int dy = -1; dy <= 1; dy++)
         for (int dx = -1; dx <= 1; dx++)
         {
            if (dx || dy || dz) { st.push_back({dx, dy, dz, 1}); }
         }
   return assemble(comm, nx, ny, nz, P, Q, R, p, q, r, st, value);
}

// convection-diffusion 7-point operator; value = {centre, -x, -y, -z, +x, +y, +z}
// (par_difconv.c:203-285; values prepared by the driver as in test/ij.c:10184-10215)
HYPRE_ParCSRMatrix GenerateDifConv(MPI_Comm comm, HYPRE_BigInt nx, HYPRE_BigInt ny, HYPRE_BigInt nz,
                                   HYPRE_Int P, HYPRE_Int Q, HYPRE_Int R, HYPRE_Int p, HYPRE_Int q,
                                   HYPRE_Int r, HYPRE_Real *value)
{
   static const std::vector<StencilPt> st = {
      {0, 0, 0, 0}, {0, 0, -1, 3}, {0, -1, 0, 2}, {-1, 0, 0, 1}, {1, 0, 0, 4}, {0, 1, 0, 5}, {0, 0, 1, 6}};
   return assemble(comm, nx, ny, nz, P, Q, R, p, q, r, st, value);
}

// Variable-coefficient diffusion -eps div(k grad u) on the unit cube, 7-point flux form with h = 1/(n+1)
// (par_vardifconv.c:15-565): k = 0.01 in the eight 0.1-corners, 1000 in the inner [0.1, 0.9]^3 box, 1 elsewhere; the
// reference's convection and reaction coefficients d, e, f, g are zero, its right-hand side function is 1 and its
// boundary function 0, so the right-hand side it returns is the vector of ones (the driver builds that itself).
namespace {
struct VarDifCtx { HYPRE_Real eps, hx, hy, hz; };
HYPRE_Real vardif_k(HYPRE_Real xx, HYPRE_Real yy, HYPRE_Real zz)
{
   const bool lx = xx < 0.1, hx = xx > 0.9, ly = yy < 0.1, hy = yy > 0.9, lz = zz < 0.1, hz = zz > 0.9;
   if ((lx || hx) && (ly || hy) && (lz || hz)) { return 0.01; }
   if (xx >= 0.1 && xx <= 0.9 && yy >= 0.1 && yy <= 0.9 && zz >= 0.1 && zz <= 0.9) { return 1000.0; }
   return 1.0;
}
void vardif_values(void *vctx, HYPRE_BigInt ix, HYPRE_BigInt iy, HYPRE_BigInt iz, HYPRE_Real *v)
{
   const VarDifCtx *c = (const VarDifCtx *) vctx;
   const HYPRE_Real xx = (HYPRE_Real) (ix + 1) * c->hx, yy = (HYPRE_Real) (iy + 1) * c->hy, zz = (HYPRE_Real) (iz + 1) * c->hz;
   const HYPRE_Real afp = c->eps * vardif_k(xx + 0.5 * c->hx, yy, zz) / c->hx / c->hx;
   const HYPRE_Real afm = c->eps * vardif_k(xx - 0.5 * c->hx, yy, zz) / c->hx / c->hx;
   const HYPRE_Real bfp = c->eps * vardif_k(xx, yy + 0.5 * c->hy, zz) / c->hy / c->hy;
   const HYPRE_Real bfm = c->eps * vardif_k(xx, yy - 0.5 * c->hy, zz) / c->hy / c->hy;
   const HYPRE_Real cfp = c->eps * vardif_k(xx, yy, zz + 0.5 * c->hz) / c->hz / c->hz;
   const HYPRE_Real cfm = c->eps * vardif_k(xx, yy, zz - 0.5 * c->hz) / c->hz / c->hz;
   const HYPRE_Real df = 0.0 / c->hx, ef = 0.0 / c->hy, ff = 0.0 / c->hz, gf = 0.0;     // dfun = efun = ffun = gfun = 0
   v[0] = afp + afm + bfp + bfm + cfp + cfm + gf - df - ef - ff;
   v[1] = -cfm; v[2] = -bfm; v[3] = -afm; v[4] = -afp + df; v[5] = -bfp + ef; v[6] = -cfp + ff;
}
}  // namespace

HYPRE_ParCSRMatrix GenerateVarDifConv(MPI_Comm comm, HYPRE_BigInt nx, HYPRE_BigInt ny, HYPRE_BigInt nz, HYPRE_Int P,
                                      HYPRE_Int Q, HYPRE_Int R, HYPRE_Int p, HYPRE_Int q, HYPRE_Int r, HYPRE_Real eps,
                                      HYPRE_ParVector *rhs_ptr)
{
   VarDifCtx ctx{eps, 1.0 / (HYPRE_Real) (nx + 1), 1.0 / (HYPRE_Real) (ny + 1), 1.0 / (HYPRE_Real) (nz + 1)};
   HYPRE_Real value[7];
   static const std::vector<StencilPt> st = {
      {0, 0, 0, 0}, {0, 0, -1, 1}, {0, -1, 0, 2}, {-1, 0, 0, 3}, {1, 0, 0, 4}, {0, 1, 0, 5}, {0, 0, 1, 6}};
   hypre_ParCSRMatrix *A = assemble(comm, nx, ny, nz, P, Q, R, p, q, r, st, value, vardif_values, &ctx);
   if (A && rhs_ptr)
   {
      hypre_ParVector *b = hypre_ParVectorCreate(comm, A->global_num_rows, A->row_starts);
      hypre_ParVectorInitialize_v2(b, HYPRE_MEMORY_HOST);
      for (HYPRE_Int i = 0; i < b->local_vector->size; i++) { b->local_vector->data[i] = 1.0; }      // rfun = 1, bndfun = 0
      *rhs_ptr = b;
   }
   return A;
}

// Rotated anisotropic diffusion in 2-D, 7-point stencil [centre, SW, S, W, E, N, NE] (par_rotate_7pt.c:15-397):
// -(c^2 + eps s^2) u_xx + 2 (1 - eps) s c u_xy - (s^2 + eps c^2) u_yy with s, c = sin, cos of alpha degrees.
HYPRE_ParCSRMatrix GenerateRotate7pt(MPI_Comm comm, HYPRE_BigInt nx, HYPRE_BigInt ny, HYPRE_Int P, HYPRE_Int Q,
                                     HYPRE_Int p, HYPRE_Int q, HYPRE_Real alpha, HYPRE_Real eps)
{
   const HYPRE_Real pi = 4.0 * std::atan(1.0), x = pi * alpha / 180.0, s = std::sin(x), c = std::cos(x);
   const HYPRE_Real ac = -(c * c + eps * s * s), bc = 2.0 * (1.0 - eps) * s * c, cc = -(s * s + eps * c * c);
   HYPRE_Real value[4] = {-2 * (2 * ac + bc + 2 * cc), 2 * ac + bc, bc + 2 * cc, -bc};
   static const std::vector<StencilPt> st = {
      {0, 0, 0, 0}, {-1, -1, 0, 3}, {0, -1, 0, 2}, {-1, 0, 0, 1}, {1, 0, 0, 1}, {0, 1, 0, 2}, {1, 1, 0, 3}};
   return assemble(comm, nx, ny, 1, P, Q, 1, p, q, 0, st, value);
}

// Systems version of the 7-point operator: num_fun unknowns per grid point, A = L (x) mtrx with the unknowns
// of a point numbered consecutively (par_laplace.c:380-848).  Every row keeps the scalar row's block order
// [centre, -z, -y, -x, +x, +y, +z], num_fun entries per block (zeros of mtrx are stored); in the rows of
// function j > 0 the first entry of the centre block and the diagonal entry trade places (:818-832).
HYPRE_ParCSRMatrix GenerateSysLaplacian(MPI_Comm comm, HYPRE_BigInt nx, HYPRE_BigInt ny, HYPRE_BigInt nz,
                                        HYPRE_Int P, HYPRE_Int Q, HYPRE_Int R, HYPRE_Int p, HYPRE_Int q,
                                        HYPRE_Int r, HYPRE_Int num_fun, HYPRE_Real *mtrx, HYPRE_Real *value)
{
   if (num_fun < 1 || !mtrx) { hypre_error_in_arg(11); return nullptr; }
   hypre_ParCSRMatrix *L = GenerateLaplacian(comm, nx, ny, nz, P, Q, R, p, q, r, value);
   if (!L) { return nullptr; }
   const HYPRE_Int nf = num_fun, ng = L->diag->num_rows, nco = L->offd->num_cols;
   const HYPRE_Int *Ldi = L->diag->i, *Ldj = L->diag->j, *Loi = L->offd->i, *Loj = L->offd->j;
   const HYPRE_Real *Lda = L->diag->data, *Loa = L->offd->data;
   std::vector<HYPRE_Int> di((size_t) ng * nf + 1, 0), oi((size_t) ng * nf + 1, 0);
   std::vector<HYPRE_Int> dj((size_t) Ldi[ng] * nf * nf), oj((size_t) Loi[ng] * nf * nf);
   std::vector<HYPRE_Real> da(dj.size()), oa(oj.size());
   size_t dp = 0, op = 0;
   for (HYPRE_Int g = 0; g < ng; g++)
   {
      for (HYPRE_Int i = 0; i < nf; i++)
      {
         const size_t row_begin = dp;
         for (HYPRE_Int k = Ldi[g]; k < Ldi[g + 1]; k++)
         {
            for (HYPRE_Int j = 0; j < nf; j++) { dj[dp] = Ldj[k] * nf + j; da[dp++] = Lda[k] * mtrx[i * nf + j]; }
         }
         if (i > 0) { std::swap(dj[row_begin], dj[row_begin + (size_t) i]); std::swap(da[row_begin], da[row_begin + (size_t) i]); }
         for (HYPRE_Int k = Loi[g]; k < Loi[g + 1]; k++)
         {
            for (HYPRE_Int j = 0; j < nf; j++) { oj[op] = Loj[k] * nf + j; oa[op++] = Loa[k] * mtrx[i * nf + j]; }
         }
         di[(size_t) g * nf + i + 1] = (HYPRE_Int) dp;
         oi[(size_t) g * nf + i + 1] = (HYPRE_Int) op;
      }
   }
   std::vector<HYPRE_BigInt> cmap((size_t) nco * nf);
   for (HYPRE_Int c = 0; c < nco; c++)
   {
      for (HYPRE_Int j = 0; j < nf; j++) { cmap[(size_t) c * nf + j] = L->col_map_offd[c] * nf + j; }
   }
   HYPRE_BigInt part[2] = {L->row_starts[0] * nf, L->row_starts[1] * nf};
   const HYPRE_BigInt gsize = L->global_num_rows * nf;
   hypre_ParCSRMatrix *A = hypre_amd_ParCSRMatrixFromArrays(comm, gsize, gsize, part, part, (HYPRE_Int) cmap.size(), cmap.data(),
                                                            di.data(), dj.data(), da.data(), oi.data(), oj.data(), oa.data(),
                                                            HYPRE_MEMORY_HOST);
   hypre_ParCSRMatrixDestroy(L);
   return A;
}

}  // extern "C"
