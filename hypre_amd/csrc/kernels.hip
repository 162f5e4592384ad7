// hypre_amd — hand-written gfx950 (CDNA4) kernels for the BoomerAMG solve phase.
//
// This file: BLAS-1 and small utility kernels (the SpMV family lives in
// spmv_kernels.hip).  All of it is HBM-bound streaming work: 16-byte loads and
// stores, grid-stride loops capped at ~2048 workgroups.
//
// Replaces (behaviourally, not structurally) the reference kernels of
//   utilities/device_utils.c:649-720    IVAXPY / IVAXPYMarked
//   utilities/device_utils.c:2422-2468  DiagScaleVector2
//   seq_mv/vector_device.c              axpy / scale / dot via rocBLAS+thrust

#include "amg_internal.hpp"

namespace hamd {

// ---------------------------------------------------------------------------
// small device helpers
// ---------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v)
{
   // 64-lane butterfly; __shfl_xor lowers to ds_swizzle / DPP on gfx950
#pragma unroll
   for (int off = 32; off > 0; off >>= 1) { v += __shfl_xor(v, off, 64); }
   return v;
}

// ---------------------------------------------------------------------------
// BLAS-1.  All are grid-stride with two doubles (16 B) per lane per step.
// ---------------------------------------------------------------------------
static inline int vec_grid(size_t n)
{
   size_t g = (n / 2 + 255) / 256;
   if (g > 2048) { g = 2048; }
   if (g < 1) { g = 1; }
   return (int) g;
}

#define VEC_LOOP_BEGIN                                                                   \
   const size_t n2 = n >> 1;                                                             \
   const size_t stride = (size_t) gridDim.x * blockDim.x;                                \
   for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += stride) {
#define VEC_LOOP_END }

__global__ void set_kernel(double *__restrict__ y, double v, size_t n)
{
   VEC_LOOP_BEGIN
      reinterpret_cast<double2 *>(y)[i] = make_double2(v, v);
   VEC_LOOP_END
   if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) { y[n - 1] = v; }
}

__global__ void copy_kernel(double *__restrict__ y, const double *__restrict__ x, size_t n)
{
   VEC_LOOP_BEGIN
      reinterpret_cast<double2 *>(y)[i] = reinterpret_cast<const double2 *>(x)[i];
   VEC_LOOP_END
   if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) { y[n - 1] = x[n - 1]; }
}

__global__ void scale_kernel(double *__restrict__ y, double a, size_t n)
{
   VEC_LOOP_BEGIN
      double2 t = reinterpret_cast<double2 *>(y)[i];
      t.x *= a; t.y *= a;
      reinterpret_cast<double2 *>(y)[i] = t;
   VEC_LOOP_END
   if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) { y[n - 1] *= a; }
}

__global__ void scale_copy_kernel(double b, const double *__restrict__ x, double *__restrict__ y, size_t n)
{
   VEC_LOOP_BEGIN
      const double2 xv = reinterpret_cast<const double2 *>(x)[i];
      reinterpret_cast<double2 *>(y)[i] = make_double2(b * xv.x, b * xv.y);
   VEC_LOOP_END
   if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) { y[n - 1] = b * x[n - 1]; }
}

// x may alias y (GMRES forms its restart residual with Axpy(a, p, p), gmres.c:939-947): no restrict
__global__ void axpy_kernel(double a, const double *x, double *y, size_t n)
{
   VEC_LOOP_BEGIN
      const double2 xv = reinterpret_cast<const double2 *>(x)[i];
      double2 t = reinterpret_cast<double2 *>(y)[i];
      t.x += a * xv.x; t.y += a * xv.y;
      reinterpret_cast<double2 *>(y)[i] = t;
   VEC_LOOP_END
   if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) { y[n - 1] += a * x[n - 1]; }
}

__global__ void axpyz_kernel(double a, const double *__restrict__ x, double b,
                             const double *__restrict__ y, double *__restrict__ z, size_t n)
{
   VEC_LOOP_BEGIN
      const double2 xv = reinterpret_cast<const double2 *>(x)[i];
      const double2 yv = reinterpret_cast<const double2 *>(y)[i];
      reinterpret_cast<double2 *>(z)[i] = make_double2(a * xv.x + b * yv.x, a * xv.y + b * yv.y);
   VEC_LOOP_END
   if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) { z[n - 1] = a * x[n - 1] + b * y[n - 1]; }
}

// y += x ./ d, optionally only where marker == mval
__global__ void elmdivpy_kernel(const double *__restrict__ x, const double *__restrict__ d,
                                double *__restrict__ y, const int *__restrict__ marker, int mval, size_t n)
{
   const size_t stride = (size_t) gridDim.x * blockDim.x;
   for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
   {
      if (marker == nullptr || marker[i] == mval) { y[i] += x[i] / d[i]; }
   }
}

// u = (w*f) ./ d : Jacobi sweep from a zero initial guess (par_relax.c:1221-1228)
// u is known to be all zeros, so u += (w f)./d is written as a plain store.
__global__ void scaled_div_kernel(double w, const double *__restrict__ f, const double *__restrict__ d,
                                  double *__restrict__ u, const int *__restrict__ marker, int mval, size_t n)
{
   const size_t stride = (size_t) gridDim.x * blockDim.x;
   for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
   {
      if (marker == nullptr || marker[i] == mval) { u[i] = (w * f[i]) / d[i]; }
   }
}

// the same, two rows a lane (16-byte loads and stores: unmarked rows, aligned vectors)
__global__ void scaled_div2_kernel(double w, const double *__restrict__ f, const double *__restrict__ d, double *__restrict__ u, size_t n)
{
   VEC_LOOP_BEGIN
      const double2 fv = reinterpret_cast<const double2 *>(f)[i], dv = reinterpret_cast<const double2 *>(d)[i];
      reinterpret_cast<double2 *>(u)[i] = make_double2((w * fv.x) / dv.x, (w * fv.y) / dv.y);
   VEC_LOOP_END
   if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) { u[n - 1] = (w * f[n - 1]) / d[n - 1]; }
}
__global__ void scaled_recip2_kernel(double w, const double *__restrict__ f, const double *__restrict__ d, double *__restrict__ z, size_t n)
{
   VEC_LOOP_BEGIN
      const double2 fv = reinterpret_cast<const double2 *>(f)[i], dv = reinterpret_cast<const double2 *>(d)[i];
      reinterpret_cast<double2 *>(z)[i] = make_double2(__dmul_rn(__dmul_rn(w, fv.x), 1.0 / dv.x), __dmul_rn(__dmul_rn(w, fv.y), 1.0 / dv.y));
   VEC_LOOP_END
   if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) { z[n - 1] = __dmul_rn(__dmul_rn(w, f[n - 1]), 1.0 / d[n - 1]); }
}

// z = (w f) .* (1 ./ d): the start of a two-stage Gauss-Seidel sweep from a zero iterate (what scale_copy + diagscale2 give, in one pass)
__global__ void scaled_recip_kernel(double w, const double *__restrict__ f, const double *__restrict__ d, double *__restrict__ z, size_t n)
{
   const size_t stride = (size_t) gridDim.x * blockDim.x;
   for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
   {
      z[i] = __dmul_rn(__dmul_rn(w, f[i]), 1.0 / d[i]);
   }
}

// u_out = u_in + r./d on marked rows, u_out = u_in elsewhere (r may alias u_out)
__global__ void jacobi_update_kernel(const double *__restrict__ u_in, const double *r, const double *__restrict__ d,
                                     const int *__restrict__ marker, int mval, double *u_out, size_t n)
{
   const size_t stride = (size_t) gridDim.x * blockDim.x;
   for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
   {
      const double ui = u_in[i];
      u_out[i] = (marker == nullptr || mval == 0 || marker[i] == mval) ? ui + r[i] / d[i] : ui;
   }
}

// y = x ./ diag ; z += beta * (x ./ diag)     (device_utils.c:2422-2468, NV = 1)
__global__ void diagscale2_kernel(const double *__restrict__ diag, const double *__restrict__ x, double beta,
                                  double *__restrict__ y, double *__restrict__ z, int computeY, size_t n)
{
   const size_t stride = (size_t) gridDim.x * blockDim.x;
   for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
   {
      const double t = x[i] * (1.0 / diag[i]);
      if (computeY) { y[i] = t; }
      z[i] += beta * t;
   }
}

// dot product: per-workgroup partials, then one workgroup folds them in a fixed
// order (bitwise reproducible run to run for a fixed n).
constexpr int DOT_BLOCKS = 1024;
__global__ __launch_bounds__(256)
void dot_partial_kernel(const double *__restrict__ x, const double *__restrict__ y, size_t n,
                        double *__restrict__ partial)
{
   __shared__ double wsum[4];
   double acc = 0.0;
   VEC_LOOP_BEGIN
      const double2 xv = reinterpret_cast<const double2 *>(x)[i];
      const double2 yv = reinterpret_cast<const double2 *>(y)[i];
      acc += xv.x * yv.x + xv.y * yv.y;
   VEC_LOOP_END
   if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) { acc += x[n - 1] * y[n - 1]; }
   acc = wave_sum(acc);
   if ((threadIdx.x & 63) == 0) { wsum[threadIdx.x >> 6] = acc; }
   __syncthreads();
   if (threadIdx.x == 0) { partial[blockIdx.x] = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]); }
}

__global__ __launch_bounds__(256)
void dot_final_kernel(const double *__restrict__ partial, int m, double *__restrict__ out)
{
   __shared__ double wsum[4];
   double acc = 0.0;
   for (int i = threadIdx.x; i < m; i += 256) { acc += partial[i]; }
   acc = wave_sum(acc);
   if ((threadIdx.x & 63) == 0) { wsum[threadIdx.x >> 6] = acc; }
   __syncthreads();
   if (threadIdx.x == 0) { out[0] = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]); }
}

// PCG, fused (krylov/pcg.c:716-760 does these as separate vector calls): x += a p, r += na s and the
// per-workgroup partials of <r, r> of the updated r, accumulated exactly like dot_partial_kernel does
// (same grid, same per-lane order), so the fused norm has the bits of a separate dot.
__global__ __launch_bounds__(256)
void pcg_update_kernel(double a, double na, const double *__restrict__ p, const double *__restrict__ s,
                       double *__restrict__ x, double *__restrict__ r, size_t n, double *__restrict__ partial)
{
   __shared__ double wsum[4];
   double acc = 0.0;
   VEC_LOOP_BEGIN
      const double2 pv = reinterpret_cast<const double2 *>(p)[i];
      const double2 sv = reinterpret_cast<const double2 *>(s)[i];
      double2 xv = reinterpret_cast<double2 *>(x)[i];
      double2 rv = reinterpret_cast<double2 *>(r)[i];
      xv.x += a * pv.x; xv.y += a * pv.y;
      rv.x += na * sv.x; rv.y += na * sv.y;
      reinterpret_cast<double2 *>(x)[i] = xv;
      reinterpret_cast<double2 *>(r)[i] = rv;
      acc += rv.x * rv.x + rv.y * rv.y;
   VEC_LOOP_END
   if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0)
   {
      x[n - 1] += a * p[n - 1];
      const double rl = r[n - 1] + na * s[n - 1];
      r[n - 1] = rl;
      acc += rl * rl;
   }
   acc = wave_sum(acc);
   if ((threadIdx.x & 63) == 0) { wsum[threadIdx.x >> 6] = acc; }
   __syncthreads();
   if (threadIdx.x == 0) { partial[blockIdx.x] = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]); }
}

// p = beta p + s, the product rounded before the sum as in Scale followed by Axpy(1, s, p)
__global__ void pcg_direction_kernel(double beta, const double *__restrict__ s, double *__restrict__ p, size_t n)
{
#pragma clang fp contract(off)
   VEC_LOOP_BEGIN
      const double2 sv = reinterpret_cast<const double2 *>(s)[i];
      double2 t = reinterpret_cast<double2 *>(p)[i];
      t.x *= beta; t.y *= beta;
      t.x += sv.x; t.y += sv.y;
      reinterpret_cast<double2 *>(p)[i] = t;
   VEC_LOOP_END
   if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) { const double t = p[n - 1] * beta; p[n - 1] = t + s[n - 1]; }
}

// strictly lower triangular part of a CSR matrix, entries kept in their stored order
__global__ void count_lower_kernel(const int *__restrict__ Ai, const int *__restrict__ Aj, int n, int *__restrict__ cnt)
{
   const int i = blockIdx.x * blockDim.x + threadIdx.x;
   if (i >= n) { return; }
   int c = 0;
   for (int k = Ai[i]; k < Ai[i + 1]; k++) { if (Aj[k] < i) { c++; } }
   cnt[i] = c;
}
__global__ void fill_lower_kernel(const int *__restrict__ Ai, const int *__restrict__ Aj, const double *__restrict__ Aa,
                                  const int *__restrict__ Li, int *__restrict__ Lj, double *__restrict__ La, int n)
{
   const int i = blockIdx.x * blockDim.x + threadIdx.x;
   if (i >= n) { return; }
   int o = Li[i];
   for (int k = Ai[i]; k < Ai[i + 1]; k++) { if (Aj[k] < i) { Lj[o] = Aj[k]; La[o] = Aa[k]; o++; } }
}

__global__ void gather_kernel(const double *__restrict__ x, const int *__restrict__ idx,
                              double *__restrict__ out, size_t n)
{
   const size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
   if (i < n) { out[i] = x[idx[i]]; }
}

// y[idx[i]] += in[i]; indices may repeat across i (several neighbours can own
// the same boundary row) -> atomics.  Used only on halo-sized arrays.
__global__ void scatter_add_kernel(const double *__restrict__ in, const int *__restrict__ idx,
                                   double *__restrict__ y, size_t n)
{
   const size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
   if (i < n) { atomicAdd(&y[idx[i]], in[i]); }
}

// ---------------------------------------------------------------------------
// CSR transpose on the device (seq_mv/csr_matop.c:1043-1270 gives the host result this reproduces: row c of A^T lists
// the rows of A that hold column c in ascending order).  Count per column, exclusive scan, scatter with per-column
// cursors (order within a column arbitrary), then every row of A^T is put in order by ranking its entries.
// ---------------------------------------------------------------------------
__global__ void count_columns_kernel(const int *__restrict__ Aj, int nnz, int *__restrict__ cnt)
{
   const size_t stride = (size_t) gridDim.x * blockDim.x;
   for (size_t k = (size_t) blockIdx.x * blockDim.x + threadIdx.x; k < (size_t) nnz; k += stride) { atomicAdd(&cnt[Aj[k]], 1); }
}

// in-place exclusive scan of data[0..n) with the total in data[n]; one workgroup walks the array (setup-time utility)
__global__ __launch_bounds__(1024) void scan_exclusive_kernel(int *__restrict__ data, int n)
{
   __shared__ int wsum[16];
   __shared__ int carry;
   const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
   if (tid == 0) { carry = 0; }
   __syncthreads();
   for (int base = 0; base < n; base += 4096)
   {
      int v[4], t = 0;
      for (int q = 0; q < 4; q++) { const int i = base + 4 * tid + q; v[q] = i < n ? data[i] : 0; t += v[q]; }
      int inc = t;
      for (int off = 1; off < 64; off <<= 1) { const int u = __shfl_up(inc, off, 64); if (lane >= off) { inc += u; } }
      if (lane == 63) { wsum[wave] = inc; }
      __syncthreads();
      int before = carry;
      for (int w = 0; w < wave; w++) { before += wsum[w]; }
      int run = before + inc - t;
      for (int q = 0; q < 4; q++) { const int i = base + 4 * tid + q; if (i < n) { data[i] = run; } run += v[q]; }
      __syncthreads();
      if (tid == 1023) { carry = run; }
      __syncthreads();
   }
   if (tid == 0) { data[n] = carry; }
}

// long arrays (launch_scan_exclusive): sum of every block of 4096 elements ...
__global__ __launch_bounds__(1024) void scan_block_sums_kernel(const int *__restrict__ data, int n, int *__restrict__ sums)
{
   __shared__ int wsum[16];
   const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
   const size_t base = (size_t) blockIdx.x * 4096;
   int t = 0;
   for (int q = 0; q < 4; q++) { const size_t i = base + 4 * (size_t) tid + q; if (i < (size_t) n) { t += data[i]; } }
   for (int off = 32; off > 0; off >>= 1) { t += __shfl_xor(t, off, 64); }
   if (lane == 0) { wsum[wave] = t; }
   __syncthreads();
   if (tid == 0) { int a = 0; for (int w = 0; w < 16; w++) { a += wsum[w]; } sums[blockIdx.x] = a; }
}
// ... and, the block sums scanned, every block's own exclusive scan from its offset; the last block writes the total
__global__ __launch_bounds__(1024) void scan_blocks_kernel(int *__restrict__ data, int n, const int *__restrict__ sums, int nb)
{
   __shared__ int wsum[16];
   const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
   const size_t base = (size_t) blockIdx.x * 4096;
   int v[4], t = 0;
   for (int q = 0; q < 4; q++) { const size_t i = base + 4 * (size_t) tid + q; v[q] = i < (size_t) n ? data[i] : 0; t += v[q]; }
   int inc = t;
   for (int off = 1; off < 64; off <<= 1) { const int u = __shfl_up(inc, off, 64); if (lane >= off) { inc += u; } }
   if (lane == 63) { wsum[wave] = inc; }
   __syncthreads();
   int before = sums[blockIdx.x];
   for (int w = 0; w < wave; w++) { before += wsum[w]; }
   int run = before + inc - t;
   for (int q = 0; q < 4; q++) { const size_t i = base + 4 * (size_t) tid + q; if (i < (size_t) n) { data[i] = run; } run += v[q]; }
   if (blockIdx.x == nb - 1 && tid == 0) { data[n] = sums[nb]; }
}

__global__ void scatter_transpose_kernel(const int *__restrict__ Ai, const int *__restrict__ Aj, const double *__restrict__ Aa,
                                         int nrows, int *__restrict__ cursor, int *__restrict__ tj, double *__restrict__ ta)
{
   // 8 lanes per row
   const int g = (blockIdx.x * blockDim.x + threadIdx.x) >> 3, sub = threadIdx.x & 7;
   if (g >= nrows) { return; }
   for (int k = Ai[g] + sub; k < Ai[g + 1]; k += 8)
   {
      const int q = atomicAdd(&cursor[Aj[k]], 1);
      tj[q] = g;
      if (Aa) { ta[q] = Aa[k]; }
   }
}

// row c of the scattered transpose -> ascending source rows: entry i goes to slot (number of entries of the row with a
// smaller source row); source rows are distinct within a row.  One wave per row.
__global__ void order_rows_kernel(const int *__restrict__ Ti, int ncols, const int *__restrict__ tj_in, const double *__restrict__ ta_in,
                                  int *__restrict__ tj, double *__restrict__ ta)
{
   const int row = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
   if (row >= ncols) { return; }
   const int s = Ti[row], e = Ti[row + 1];
   for (int i = s + lane; i < e; i += 64)
   {
      const int mine = tj_in[i];
      int rank = 0;
      for (int k = s; k < e; k++) { rank += tj_in[k] < mine ? 1 : 0; }
      tj[s + rank] = mine;
      if (ta_in) { ta[s + rank] = ta_in[i]; }
   }
}

// ghost data of a multivector exchange arrives entry by entry ([ghost][component]); the products want it column by column
__global__ void deinterleave_kernel(const double *__restrict__ in, double *__restrict__ out, int n, int nv)
{
   const size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
   if (i < (size_t) n * nv) { const int g = (int) (i / nv), j = (int) (i % nv); out[(size_t) j * n + g] = in[i]; }
}
__global__ void interleave_kernel(const double *__restrict__ in, double *__restrict__ out, int n, int nv)
{
   const size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
   if (i < (size_t) n * nv) { const int g = (int) (i / nv), j = (int) (i % nv); out[i] = in[(size_t) j * n + g]; }
}

__global__ void f64_to_f32_kernel(const double *__restrict__ x, float *__restrict__ y, size_t n)
{
   const size_t stride = (size_t) gridDim.x * blockDim.x;
   for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) { y[i] = (float) x[i]; }
}

// d[i] = first stored entry of row i (the diagonal, by the diagonal-first invariant)
__global__ void diag_first_kernel(const int *__restrict__ Ai, const double *__restrict__ Aa,
                                  double *__restrict__ d, int n)
{
   const int i = blockIdx.x * blockDim.x + threadIdx.x;
   if (i < n) { d[i] = (Ai[i + 1] > Ai[i]) ? Aa[Ai[i]] : 0.0; }
}

// Coarsest level (<= a few dozen unknowns): forward elimination / back
// substitution with the factors of the reference's pivot-free elimination
// (utilities/gselim.h), one lane, operations in the reference's order and
// without fused multiply-adds so the result matches the host loop bit for bit.
// Systems of at most 32 unknowns: the factors staged in LDS, lane j of the wave holds x[j]; step k of the elimination
// updates every j > k at once (x[k] from lane k), the back substitution every j < k — each x[j] sees the operations of the
// one-lane loop below in its order, so the bits are the same; 2 n steps of an LDS read instead of n^2 dependent trips to
// memory (16 us -> 3 us for the 8 unknowns of the benchmark hierarchy).
__global__ __launch_bounds__(64) void coarse_solve_wave_kernel(const double *__restrict__ lu_g, double *__restrict__ xg, int n)
{
   __shared__ double lu[32 * 32];
   const int j = threadIdx.x;
   for (int i = j; i < n * n; i += 64) { lu[i] = lu_g[i]; }
   double x = j < n ? xg[j] : 0.0;
   __syncthreads();
   if (n == 1) { if (j == 0 && lu[0] != 0.0) { x = x / lu[0]; } }
   else
   {
      for (int k = 0; k < n - 1; k++)
      {
         const double xk = __shfl(x, k, 64);
         if (lu[k * n + k] != 0.0 && j > k && j < n)
         {
            const double factor = lu[j * n + k];
            if (factor != 0.0) { x = __dsub_rn(x, __dmul_rn(factor, xk)); }
         }
      }
      for (int k = n - 1; k > 0; --k)
      {
         const double piv = lu[k * n + k];
         if (piv != 0.0 && j == k) { x = x / piv; }
         const double xk = __shfl(x, k, 64);
         if (piv != 0.0 && j < k)
         {
            const double c = lu[j * n + k];
            if (c != 0.0) { x = __dsub_rn(x, __dmul_rn(xk, c)); }
         }
      }
      if (j == 0 && lu[0] != 0.0) { x = x / lu[0]; }
   }
   if (j < n) { xg[j] = x; }
}

__global__ void coarse_solve_kernel(const double *__restrict__ lu, double *__restrict__ x, int n)
{
   if (threadIdx.x != 0 || blockIdx.x != 0) { return; }
   if (n == 1) { if (lu[0] != 0.0) { x[0] = x[0] / lu[0]; } return; }
   for (int k = 0; k < n - 1; k++)
   {
      if (lu[k * n + k] != 0.0)
      {
         for (int j = k + 1; j < n; j++)
         {
            const double factor = lu[j * n + k];        // multiplier stored below the diagonal
            if (factor != 0.0) { x[j] = __dsub_rn(x[j], __dmul_rn(factor, x[k])); }
         }
      }
   }
   for (int k = n - 1; k > 0; --k)
   {
      if (lu[k * n + k] != 0.0)
      {
         x[k] = x[k] / lu[k * n + k];
         for (int j = 0; j < k; j++)
         {
            if (lu[j * n + k] != 0.0) { x[j] = __dsub_rn(x[j], __dmul_rn(x[k], lu[j * n + k])); }
         }
      }
   }
   if (lu[0] != 0.0) { x[0] = x[0] / lu[0]; }
}

static inline int lin_grid(size_t n)
{
   size_t g = (n + 255) / 256;
   if (g > 4096) { g = 4096; }
   if (g < 1) { g = 1; }
   return (int) g;
}

void launch_set(double *y, double v, size_t n, hipStream_t s)
{ account_bytes(8.0 * n); if (n) hipLaunchKernelGGL(set_kernel, dim3(vec_grid(n)), dim3(256), 0, s, y, v, n); }
void launch_copy(double *y, const double *x, size_t n, hipStream_t s)
{ account_bytes(16.0 * n); if (n) hipLaunchKernelGGL(copy_kernel, dim3(vec_grid(n)), dim3(256), 0, s, y, x, n); }
void launch_scale(double *y, double a, size_t n, hipStream_t s)
{ account_bytes(16.0 * n); if (n) hipLaunchKernelGGL(scale_kernel, dim3(vec_grid(n)), dim3(256), 0, s, y, a, n); }
void launch_scale_copy(double b, const double *x, double *y, size_t n, hipStream_t s)
{ account_bytes(16.0 * n); if (n) hipLaunchKernelGGL(scale_copy_kernel, dim3(vec_grid(n)), dim3(256), 0, s, b, x, y, n); }
void launch_axpy(double a, const double *x, double *y, size_t n, hipStream_t s)
{ account_bytes(24.0 * n); if (n) hipLaunchKernelGGL(axpy_kernel, dim3(vec_grid(n)), dim3(256), 0, s, a, x, y, n); }
void launch_axpyz(double a, const double *x, double b, const double *y, double *z, size_t n, hipStream_t s)
{ account_bytes(24.0 * n); if (n) hipLaunchKernelGGL(axpyz_kernel, dim3(vec_grid(n)), dim3(256), 0, s, a, x, b, y, z, n); }
void launch_elmdivpy(const double *x, const double *d, double *y, const int *marker, int mval, size_t n, hipStream_t s)
{ account_bytes((32.0 + (marker ? 4.0 : 0.0)) * n); if (n) hipLaunchKernelGGL(elmdivpy_kernel, dim3(lin_grid(n)), dim3(256), 0, s, x, d, y, marker, mval, n); }
void launch_scaled_div(double w, const double *f, const double *d, double *u, const int *marker, int mval, size_t n, hipStream_t s)
{
   account_bytes((24.0 + (marker ? 4.0 : 0.0)) * n);
   if (!n) { return; }
   const bool aligned = ((((uintptr_t) f) | ((uintptr_t) d) | ((uintptr_t) u)) & 15) == 0;
   if (!marker && aligned) { hipLaunchKernelGGL(scaled_div2_kernel, dim3(vec_grid(n)), dim3(256), 0, s, w, f, d, u, n); }
   else { hipLaunchKernelGGL(scaled_div_kernel, dim3(lin_grid(n)), dim3(256), 0, s, w, f, d, u, marker, mval, n); }
}
// x(:, v) = y(:, v) ./ (first entry of every row of A): hypre_ParCSRDiagScaleVector (par_csr_matop.c:6479-6575), the
// preconditioner of the reference driver's DS-PCG / DS-GMRES; nv columns ys / xs doubles apart
__global__ void diag_first_scale_kernel(const int *__restrict__ Ai, const double *__restrict__ Aa, const double *__restrict__ y,
                                        double *__restrict__ x, size_t n, int nv, size_t ys, size_t xs)
{
   for (size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x)
   {
      const double d = Aa[Ai[i]];
      for (int v = 0; v < nv; v++) { x[(size_t) v * xs + i] = y[(size_t) v * ys + i] / d; }
   }
}
void launch_diag_first_scale(const int *Ai, const double *Aa, const double *y, double *x, size_t n, int nv, size_t ys, size_t xs, hipStream_t s)
{
   account_bytes((12.0 + 16.0 * nv) * n);
   if (n && nv > 0) { hipLaunchKernelGGL(diag_first_scale_kernel, dim3(lin_grid(n)), dim3(256), 0, s, Ai, Aa, y, x, n, nv, ys, xs); }
}
void launch_scaled_recip(double w, const double *f, const double *d, double *z, size_t n, hipStream_t s)
{
   account_bytes(24.0 * n);
   if (!n) { return; }
   const bool aligned = ((((uintptr_t) f) | ((uintptr_t) d) | ((uintptr_t) z)) & 15) == 0;
   if (aligned) { hipLaunchKernelGGL(scaled_recip2_kernel, dim3(vec_grid(n)), dim3(256), 0, s, w, f, d, z, n); }
   else { hipLaunchKernelGGL(scaled_recip_kernel, dim3(lin_grid(n)), dim3(256), 0, s, w, f, d, z, n); }
}
void launch_diagscale2(const double *diag, const double *x, double beta, double *y, double *z, int computeY, size_t n, hipStream_t s)
{ account_bytes(40.0 * n); if (n) hipLaunchKernelGGL(diagscale2_kernel, dim3(lin_grid(n)), dim3(256), 0, s, diag, x, beta, y, z, computeY, n); }
void launch_dot(const double *x, const double *y, size_t n, double *d_out, hipStream_t s)
{
   account_bytes(16.0 * n);
   int nb = vec_grid(n);
   if (nb > DOT_BLOCKS) { nb = DOT_BLOCKS; }
   double *partial = reduce_scratch(DOT_BLOCKS + 16) + 16;
   hipLaunchKernelGGL(dot_partial_kernel, dim3(nb), dim3(256), 0, s, x, y, n, partial);
   hipLaunchKernelGGL(dot_final_kernel, dim3(1), dim3(256), 0, s, partial, nb, d_out);
}
void launch_pcg_update(double a, double na, const double *p, const double *sv, double *x, double *r, size_t n,
                       double *d_out, hipStream_t s)
{
   account_bytes(48.0 * n);
   int nb = vec_grid(n);
   if (nb > DOT_BLOCKS) { nb = DOT_BLOCKS; }
   double *partial = reduce_scratch(DOT_BLOCKS + 16) + 16;
   hipLaunchKernelGGL(pcg_update_kernel, dim3(nb), dim3(256), 0, s, a, na, p, sv, x, r, n, partial);
   hipLaunchKernelGGL(dot_final_kernel, dim3(1), dim3(256), 0, s, partial, nb, d_out);
}
void launch_pcg_direction(double beta, const double *sv, double *p, size_t n, hipStream_t s)
{ account_bytes(24.0 * n); if (n) hipLaunchKernelGGL(pcg_direction_kernel, dim3(vec_grid(n)), dim3(256), 0, s, beta, sv, p, n); }
void launch_count_lower(const HYPRE_Int *Ai, const HYPRE_Int *Aj, int n, int *cnt, hipStream_t s)
{ if (n > 0) hipLaunchKernelGGL(count_lower_kernel, dim3((n + 255) / 256), dim3(256), 0, s, Ai, Aj, n, cnt); }
void launch_fill_lower(const HYPRE_Int *Ai, const HYPRE_Int *Aj, const double *Aa, const HYPRE_Int *Li, HYPRE_Int *Lj,
                       double *La, int n, hipStream_t s)
{ if (n > 0) hipLaunchKernelGGL(fill_lower_kernel, dim3((n + 255) / 256), dim3(256), 0, s, Ai, Aj, Aa, Li, Lj, La, n); }
void launch_gather(const double *x, const int *idx, double *out, size_t n, hipStream_t s)
{ account_bytes(20.0 * n); if (n) hipLaunchKernelGGL(gather_kernel, dim3((n + 255) / 256), dim3(256), 0, s, x, idx, out, n); }
void launch_scatter_add(const double *in, const int *idx, double *y, size_t n, hipStream_t s)
{ account_bytes(28.0 * n); if (n) hipLaunchKernelGGL(scatter_add_kernel, dim3((n + 255) / 256), dim3(256), 0, s, in, idx, y, n); }
void launch_jacobi_update(const double *u_in, const double *r, const double *d, const int *marker, int mval,
                          double *u_out, size_t n, hipStream_t s)
{ account_bytes((32.0 + (marker ? 4.0 : 0.0)) * n); if (n) hipLaunchKernelGGL(jacobi_update_kernel, dim3(lin_grid(n)), dim3(256), 0, s, u_in, r, d, marker, mval, u_out, n); }
void launch_diag_first(const int *Ai, const double *Aa, double *d, int n, hipStream_t s)
{ account_bytes(20.0 * n); if (n > 0) hipLaunchKernelGGL(diag_first_kernel, dim3((n + 255) / 256), dim3(256), 0, s, Ai, Aa, d, n); }
void launch_coarse_solve(const double *lu, double *x, int n, hipStream_t s)
{
   if (n <= 0) { return; }
   if (n <= 32) { hipLaunchKernelGGL(coarse_solve_wave_kernel, dim3(1), dim3(64), 0, s, lu, x, n); }
   else { hipLaunchKernelGGL(coarse_solve_kernel, dim3(1), dim3(64), 0, s, lu, x, n); }
}
// A (nrows x ncols, device) -> Ti[ncols + 1], tj[nnz], ta[nnz] (device, allocated by the caller); Aa / ta may be null
void launch_transpose(const int *Ai, const int *Aj, const double *Aa, int nrows, int ncols, int nnz, int *Ti, int *tj, double *ta,
                      hipStream_t s)
{
   HIP_CHECK(hipMemsetAsync(Ti, 0, sizeof(int) * ((size_t) ncols + 1), s));
   if (nnz <= 0 || nrows <= 0) { return; }
   int grid = (int) std::min<size_t>(((size_t) nnz + 255) / 256, 65536);
   hipLaunchKernelGGL(count_columns_kernel, dim3(grid), dim3(256), 0, s, Aj, nnz, Ti);
   launch_scan_exclusive(Ti, ncols, s);
   int *cursor = nullptr, *tj0 = nullptr;
   double *ta0 = nullptr;
   HIP_CHECK(hipMalloc((void **) &cursor, sizeof(int) * ((size_t) ncols + 1)));
   HIP_CHECK(hipMalloc((void **) &tj0, sizeof(int) * (size_t) nnz));
   if (Aa) { HIP_CHECK(hipMalloc((void **) &ta0, sizeof(double) * (size_t) nnz)); }
   HIP_CHECK(hipMemcpyAsync(cursor, Ti, sizeof(int) * ((size_t) ncols + 1), hipMemcpyDeviceToDevice, s));
   hipLaunchKernelGGL(scatter_transpose_kernel, dim3(((size_t) nrows * 8 + 255) / 256), dim3(256), 0, s, Ai, Aj, Aa, nrows, cursor, tj0, ta0);
   hipLaunchKernelGGL(order_rows_kernel, dim3(((size_t) ncols * 64 + 255) / 256), dim3(256), 0, s, Ti, ncols, tj0, ta0, tj, ta);
   HIP_CHECK(hipStreamSynchronize(s));
   HIP_CHECK(hipFree(cursor));
   HIP_CHECK(hipFree(tj0));
   if (ta0) { HIP_CHECK(hipFree(ta0)); }
}
// in-place exclusive scan of data[0..n), total in data[n] (the array holds n + 1 ints).  Short arrays: one workgroup
// walks them; long ones: sums of 4096-element blocks, their scan (this routine again), then every block scans itself
// from its offset.
void launch_scan_exclusive(int *data, int n, hipStream_t s)
{
   if (n <= 16384) { hipLaunchKernelGGL(scan_exclusive_kernel, dim3(1), dim3(1024), 0, s, data, n); return; }
   static int *sums[2] = {nullptr, nullptr};
   static size_t room[2] = {0, 0};
   static int depth = 0;
   const int nb = (n + 4095) / 4096;
   const int d = depth;
   if (d >= 2) { hipLaunchKernelGGL(scan_exclusive_kernel, dim3(1), dim3(1024), 0, s, data, n); return; }
   if (room[d] < (size_t) nb + 1)
   {
      if (sums[d]) { HIP_CHECK(hipFree(sums[d])); }
      room[d] = (size_t) nb + 1 + 1024;
      HIP_CHECK(hipMalloc((void **) &sums[d], sizeof(int) * room[d]));
   }
   hipLaunchKernelGGL(scan_block_sums_kernel, dim3(nb), dim3(1024), 0, s, data, n, sums[d]);
   depth++;
   launch_scan_exclusive(sums[d], nb, s);
   depth--;
   hipLaunchKernelGGL(scan_blocks_kernel, dim3(nb), dim3(1024), 0, s, data, n, sums[d], nb);
}
// Columns of every row in ascending order, the first entry (the diagonal of a square block) left in front.  One wave per
// row: bitonic network over the row in LDS.  Rows longer than the network (2048 entries) are left as they are.
// (A utility, hypre_amd_CSRMatrixSortRows: the x-staged SpMV runs no faster on sorted rows — measured on one and the same
// allocation, tools/experiments/sorted_rows_inplace.py — so the setup leaves the Galerkin products' first-touch order.)
constexpr int SORT_CAP = 2048;
__global__ __launch_bounds__(64)
void sort_rows_kernel(int n, const int *__restrict__ Ai, int *__restrict__ Aj, double *__restrict__ Aa, int keep_first)
{
   __shared__ int key[SORT_CAP];
   __shared__ double val[SORT_CAP];
   const int lane = threadIdx.x;
   for (int row = blockIdx.x; row < n; row += gridDim.x)
   {
      const int e = Ai[row + 1];
      int b = Ai[row];
      if (keep_first && e > b) { b++; }
      const int len = e - b;
      if (len <= 1 || len > SORT_CAP) { continue; }
      // already ascending?  (restriction operators, sorted inputs)
      bool asc = true;
      for (int k = lane; k + 1 < len; k += 64) { if (Aj[b + k] > Aj[b + k + 1]) { asc = false; } }
      if (__ballot(!asc) == 0ull) { continue; }
      int m = 2;
      while (m < len) { m <<= 1; }
      for (int k = lane; k < m; k += 64)
      {
         key[k] = k < len ? Aj[b + k] : 0x7fffffff;
         val[k] = k < len ? Aa[b + k] : 0.0;
      }
      __syncthreads();
      for (int size = 2; size <= m; size <<= 1)
      {
         for (int stride = size >> 1; stride > 0; stride >>= 1)
         {
            for (int t = lane; t < (m >> 1); t += 64)
            {
               const int i = 2 * t - (t & (stride - 1)), j = i + stride;
               const bool up = (i & size) == 0;
               const int ki = key[i], kj = key[j];
               if ((ki > kj) == up)
               {
                  key[i] = kj; key[j] = ki;
                  const double vi = val[i]; val[i] = val[j]; val[j] = vi;
               }
            }
            __syncthreads();
         }
      }
      for (int k = lane; k < len; k += 64) { Aj[b + k] = key[k]; Aa[b + k] = val[k]; }
      __syncthreads();
   }
}
void launch_sort_rows(const int *Ai, int *Aj, double *Aa, int n, int keep_first, hipStream_t s)
{
   if (n <= 0) { return; }
   const int grid = std::min(n, handle().num_cus * 64);
   hipLaunchKernelGGL(sort_rows_kernel, dim3(grid), dim3(64), 0, s, n, Ai, Aj, Aa, keep_first);
}

// fingerprint of a CSR matrix over 4096 positions spread over its rows and entries (MatrixWatch, internal.hpp): sixteen
// workgroups, a position per lane, a partial fingerprint per workgroup — the positions are cold cache lines, and one
// workgroup fetching all 12 288 of them took 26 us per check, 0.4 ms of a multicolour cycle
constexpr int FP_BLOCKS = 16;
__global__ __launch_bounds__(256)
void matrix_fingerprint_kernel(const int *__restrict__ Ai, const int *__restrict__ Aj, const double *__restrict__ Aa, int n, int nnz,
                               unsigned long long *fp, int *stale, int record)
{
   auto mix = [](unsigned long long x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return x; };
   unsigned long long h = 0ull;
   {
      const long long k = (long long) blockIdx.x * 256 + threadIdx.x;
      if (n > 0) { const int r = (int) (k * n / 4096); h ^= mix((unsigned long long) (unsigned) Ai[r + 1] + ((unsigned long long) k << 32)); }
      if (nnz > 0)
      {
         const int p = (int) (k * nnz / 4096);
         h ^= mix((unsigned long long) (unsigned) Aj[p] + ((unsigned long long) (k + 4096) << 32));
         if (Aa) { h ^= mix((unsigned long long) __double_as_longlong(Aa[p]) + (unsigned long long) k); }
      }
   }
   for (int off = 32; off > 0; off >>= 1)
   {
      const unsigned lo = __shfl_xor((unsigned) h, off, 64), hi = __shfl_xor((unsigned) (h >> 32), off, 64);
      h ^= ((unsigned long long) hi << 32) | lo;
   }
   __shared__ unsigned long long part[4];
   if ((threadIdx.x & 63) == 0) { part[threadIdx.x >> 6] = h; }
   __syncthreads();
   if (threadIdx.x == 0)
   {
      h = part[0] ^ part[1] ^ part[2] ^ part[3];
      if (record) { fp[blockIdx.x] = h; }
      else if (fp[blockIdx.x] != h) { __hip_atomic_fetch_or(stale, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
   }
}
// fp: MATRIX_FP_WORDS values
void launch_matrix_fingerprint(const int *Ai, const int *Aj, const double *Aa, int n, int nnz, unsigned long long *fp, int *stale,
                               int record, hipStream_t s)
{
   static_assert(FP_BLOCKS == MATRIX_FP_WORDS, "one partial fingerprint per workgroup");
   hipLaunchKernelGGL(matrix_fingerprint_kernel, dim3(FP_BLOCKS), dim3(256), 0, s, Ai, Aj, Aa, n, nnz, fp, stale, record);
}

void launch_deinterleave(const double *in, double *out, int n, int nv, hipStream_t s)
{ if (n > 0 && nv > 0) hipLaunchKernelGGL(deinterleave_kernel, dim3(((size_t) n * nv + 255) / 256), dim3(256), 0, s, in, out, n, nv); }
void launch_interleave(const double *in, double *out, int n, int nv, hipStream_t s)
{ if (n > 0 && nv > 0) hipLaunchKernelGGL(interleave_kernel, dim3(((size_t) n * nv + 255) / 256), dim3(256), 0, s, in, out, n, nv); }
void launch_f64_to_f32(const double *x, float *y, size_t n, hipStream_t s)
{ if (n) hipLaunchKernelGGL(f64_to_f32_kernel, dim3(lin_grid(n)), dim3(256), 0, s, x, y, n); }

// The code object of this file is loaded when one of its kernels is first asked for: ensure_device() asks here, so that
// the load (tens of milliseconds per file) is part of bringing the device up, not of the first setup or solve.
void preload_vector_kernels() { hipFuncAttributes at; (void) hipFuncGetAttributes(&at, (const void *) set_kernel); (void) hipGetLastError(); }

}  // namespace hamd
