// hypre_amd — elementwise passes of the Chebyshev smoother between its SpMVs
// (parcsr_ls/par_cheby.c:272-399 restated; one fused pass where the reference has
// two to four).  Products and sums are rounded one by one as in the host loops.
#include "amg_internal.hpp"

#pragma clang fp contract(off)

namespace hamd {

// t = -A u on entry.  r = ds.*(f + t) [ds == nullptr: r = f + t, t and r may be the same array],
// orig = u, u = c r; when further terms follow also tmp = ds.*u, otherwise the correction is added
// right away: u = orig + ds.*u.
__global__ __launch_bounds__(256)
void cheby_start_kernel(const double *__restrict__ f, const double *t, const double *__restrict__ ds, double c,
                        int last, double *__restrict__ u, double *__restrict__ orig, double *r, double *tmp, size_t n)
{
   for (size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x)
   {
      const double rr = ds ? ds[i] * (f[i] + t[i]) : f[i] + t[i];
      const double uo = u[i];
      const double un = rr * c;
      r[i] = rr;
      orig[i] = uo;
      if (last) { u[i] = ds ? uo + ds[i] * un : uo + un; }
      else
      {
         u[i] = un;
         if (ds) { tmp[i] = ds[i] * un; }
      }
   }
}

// v = A (ds.*u) on entry.  u = mult r + ds.*v, then as above.
__global__ __launch_bounds__(256)
void cheby_step_kernel(const double *__restrict__ r, const double *__restrict__ v, const double *__restrict__ ds,
                       const double *__restrict__ orig, double mult, int last, double *__restrict__ u,
                       double *__restrict__ tmp, size_t n)
{
   for (size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x)
   {
      const double un = ds ? mult * r[i] + ds[i] * v[i] : mult * r[i] + v[i];
      if (last) { u[i] = ds ? orig[i] + ds[i] * un : orig[i] + un; }
      else
      {
         u[i] = un;
         if (ds) { tmp[i] = ds[i] * un; }
      }
   }
}

static inline int ew_grid(size_t n)
{
   size_t g = (n + 255) / 256;
   if (g > 65536) { g = 65536; }
   return (int) (g < 1 ? 1 : g);
}

void launch_cheby_start(const double *f, const double *t, const double *ds, double c, bool last, double *u,
                        double *orig, double *r, double *tmp, size_t n, hipStream_t s)
{
   if (n == 0) { return; }
   hipLaunchKernelGGL(cheby_start_kernel, dim3(ew_grid(n)), dim3(256), 0, s, f, t, ds, c, last ? 1 : 0, u, orig, r, tmp, n);
}

void launch_cheby_step(const double *r, const double *v, const double *ds, const double *orig, double mult, bool last,
                       double *u, double *tmp, size_t n, hipStream_t s)
{
   if (n == 0) { return; }
   hipLaunchKernelGGL(cheby_step_kernel, dim3(ew_grid(n)), dim3(256), 0, s, r, v, ds, orig, mult, last ? 1 : 0, u, tmp, n);
}

// The code object of this file is loaded when one of its kernels is first asked for: ensure_device() asks here, so that
// the load (tens of milliseconds per file) is part of bringing the device up, not of the first setup or solve.
void preload_cheby_kernels() { hipFuncAttributes at; (void) hipFuncGetAttributes(&at, (const void *) cheby_start_kernel); (void) hipGetLastError(); }

}  // namespace hamd
