// hypre_amd — level-scheduled hybrid Gauss-Seidel / SOR row kernels.
//
// The reference's host sweep (parcsr_ls/par_relax.c:691-945 with the row bodies
// of par_relax.h:13-457) walks the rows of a thread's block in order and reads
// whatever u holds at that moment.  Here the rows of a block are grouped into
// levels (a row's level is one more than the deepest in-block row it reads that
// precedes it in sweep order); a level's rows are independent and run
// concurrently, levels run in order.  A row reads
//    * in-block columns that precede it in sweep order from u (already final:
//      they sit in an earlier level),
//    * in-block columns that follow it from a copy of u taken when the
//      directional sweep started (the sequential loop has not reached them),
//    * off-block columns from the copy taken when the relaxation call started
//      (par_relax.c:756-760: Vtemp), ghost columns from the halo of that state,
// and sums its row in stored order without fused multiply-adds, so the result
// equals the sequential sweep bit for bit.
#include "amg_internal.hpp"

// the host loop rounds every product and every sum: no fused multiply-add in this file
// (plain operators below; the __dmul_rn-style wrappers of the HIP headers are themselves
// contractible once inlined)
#pragma clang fp contract(off)

namespace hamd {

// block of row i under hypre_partition1D(n, p, ...) (utilities/threading.c)
__device__ __forceinline__ void gs_block_of(int n, int p, int i, int &ns, int &ne)
{
   if (p <= 1) { ns = 0; ne = n; return; }
   const int size = n / p, rest = n - size * p;
   const int cut = rest * (size + 1);
   if (i < cut) { const int t = i / (size + 1); ns = t * (size + 1); ne = ns + size + 1; }
   else { const int t = (i - cut) / size; ns = cut + t * size; ne = ns + size; }
}

__device__ __forceinline__ void gs_row(const GsArgs &a, int i)
{
   const double di = a.l1 ? a.l1[i] : a.Da[a.Di[i]];
   if (!((a.relax_points == 0 || a.cf[i] == a.relax_points) && di != 0.0)) { return; }
   int ns, ne;
   gs_block_of(a.n, a.threads, i, ns, ne);
   const int s = a.Di[i] + a.skip_diag, e = a.Di[i + 1];
   if (a.non_scale)
   {
      double res = a.f[i];
      for (int jj = s; jj < e; jj++)
      {
         const int ii = a.Dj[jj];
         double v;
         if (ii >= ns && ii < ne) { v = ((a.dir > 0) ? (ii < i) : (ii > i)) ? a.u[ii] : a.uold[ii]; }
         else { v = a.vtemp[ii]; }
         res -= a.Da[jj] * v;
      }
      if (a.Oi)
      {
         for (int jj = a.Oi[i]; jj < a.Oi[i + 1]; jj++) { res -= a.Oa[jj] * a.vext[a.Oj[jj]]; }
      }
      const double q = res / di;
      a.u[i] = a.skip_diag ? q : a.uold[i] + q;
   }
   else
   {
      const double one_minus_omega = 1.0 - a.omega, prod = 1.0 - a.w * a.omega;
      double res = a.f[i], res0 = 0.0, res2 = 0.0;
      for (int jj = s; jj < e; jj++)
      {
         const int ii = a.Dj[jj];
         if (ii >= ns && ii < ne)
         {
            const double v = ((a.dir > 0) ? (ii < i) : (ii > i)) ? a.u[ii] : a.uold[ii];
            res0 -= a.Da[jj] * v;
            res2 += a.Da[jj] * a.vtemp[ii];
         }
         else { res -= a.Da[jj] * a.vtemp[ii]; }
      }
      if (a.Oi)
      {
         for (int jj = a.Oi[i]; jj < a.Oi[i + 1]; jj++) { res -= a.Oa[jj] * a.vext[a.Oj[jj]]; }
      }
      double un = a.uold[i];
      if (a.skip_diag) { un *= prod; }
      un += a.w * (a.omega * res + res0 + one_minus_omega * res2) / di;
      a.u[i] = un;
   }
}

// one level, one lane per row
__global__ __launch_bounds__(256)
void gs_level_kernel(GsArgs a, int start, int count)
{
   const int k = blockIdx.x * blockDim.x + threadIdx.x;
   if (k < count) { gs_row(a, a.rows[start + k]); }
}

// a run of small levels inside one workgroup: barrier between levels
__global__ __launch_bounds__(1024)
void gs_multilevel_kernel(GsArgs a, const int *__restrict__ lev_start, int lev_begin, int lev_end)
{
   for (int lev = lev_begin; lev < lev_end; lev++)
   {
      const int s = lev_start[lev], e = lev_start[lev + 1];
      for (int k = s + (int) threadIdx.x; k < e; k += (int) blockDim.x) { gs_row(a, a.rows[k]); }
      __threadfence_block();
      __syncthreads();
   }
}

void launch_gs_level(const GsArgs &a, int start, int count, hipStream_t s)
{
   if (count <= 0) { return; }
   hipLaunchKernelGGL(gs_level_kernel, dim3((count + 255) / 256), dim3(256), 0, s, a, start, count);
}

void launch_gs_multilevel(const GsArgs &a, const int *d_lev_start, int lev_begin, int lev_end, hipStream_t s)
{
   if (lev_end <= lev_begin) { return; }
   hipLaunchKernelGGL(gs_multilevel_kernel, dim3(1), dim3(1024), 0, s, a, d_lev_start, lev_begin, lev_end);
}

}  // namespace hamd
