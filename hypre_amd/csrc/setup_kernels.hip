// hypre_amd — strength of connection, PMIS coarsening and the smoother diagonals on the device (single rank), with the
// results of the host setup array for array.
//
// Reference: parcsr_ls/par_strength.c:75-530 (hypre_BoomerAMGCreateS), parcsr_ls/par_coarsen.c:2101-2810
// (hypre_BoomerAMGCoarsenPMIS) with parcsr_ls/par_indepset.c (measures) and utilities/random.c (Park-Miller generator),
// parcsr_ls/ams.c:527-830 (hypre_ParCSRComputeL1Norms).  The reference's own device versions
// (par_strength_device.c, par_coarsen_device.c:30) draw their random numbers from the vendor generator and therefore
// produce a different — equally valid — splitting than the host code; here the device follows the HOST routine, whose
// hierarchies the reference's regression files pin: the same measures (the generator's k-th value is seed * 16807^k mod
// 2^31-1: every row computes its own by repeated squaring), and sweeps that are independent of the order in which the
// points are visited (a point's fate depends on the measures, which a sweep does not change, and on which neighbours
// are in the set, which only grows), so one thread per row gives the sequential result.
//
// Rows are short (7 - 90 entries) and the work is a fraction of a second a level on the host: one thread per row keeps
// every per-row sum in the host's order, which is what makes the strength test's row sum and the l1 norms bit-equal.
#include "internal.hpp"
#include <algorithm>

#pragma clang fp contract(off)

namespace hamd {

namespace {

constexpr int TB = 256;
// one thread per item; the strided kernels below (column_count) take fewer workgroups than that happily
inline int grid_for(size_t n) { return (int) std::min<size_t>((n + TB - 1) / TB, (size_t) 0x7fffffff); }

// ---- strength ----------------------------------------------------------------------------------------------------
// the first stored entry of a row is its diagonal (hypre's convention for square ParCSR blocks)
template <bool FILL>
__global__ __launch_bounds__(TB)
void strength_kernel(int n, const int *__restrict__ Ai, const int *__restrict__ Aj, const double *__restrict__ Aa,
                     double theta, double max_row_sum, int *__restrict__ cnt, const int *__restrict__ Si, int *__restrict__ Sj)
{
   const int i = blockIdx.x * TB + threadIdx.x;
   if (i >= n) { return; }
   const int b = Ai[i], e = Ai[i + 1];
   if (e <= b) { if (!FILL) { cnt[i] = 0; } return; }
   const double diag = Aa[b];
   double row_scale = 0.0, row_sum = diag;
   if (diag < 0) { for (int k = b + 1; k < e; k++) { const double a = Aa[k]; row_scale = fmax(row_scale, a); row_sum += a; } }
   else          { for (int k = b + 1; k < e; k++) { const double a = Aa[k]; row_scale = fmin(row_scale, a); row_sum += a; } }
   const bool all_weak = (fabs(row_sum) > fabs(diag) * max_row_sum) && (max_row_sum < 1.0);
   const double bar = theta * row_scale;
   int c = 0, p = FILL ? Si[i] : 0;
   if (!all_weak)
   {
      for (int k = b + 1; k < e; k++)
      {
         const double a = Aa[k];
         const bool strong = diag < 0 ? !(a <= bar) : !(a >= bar);
         if (strong) { if (FILL) { Sj[p++] = Aj[k]; } else { c++; } }
      }
   }
   if (!FILL) { cnt[i] = c; }
}

// A rank's two blocks (par_strength.c:75-530 with ghost columns): the row is "diag entries, then offd entries" for the
// scale, the row sum and the strong couplings alike; a strong ghost column g is stored as n + g (extended numbering)
template <bool FILL>
__global__ __launch_bounds__(TB)
void strength_blocks_kernel(int n, const int *__restrict__ Di, const int *__restrict__ Dj, const double *__restrict__ Da,
                            const int *__restrict__ Oi, const int *__restrict__ Oj, const double *__restrict__ Oa,
                            double theta, double max_row_sum, int *__restrict__ cnt, const int *__restrict__ Si, int *__restrict__ Sj)
{
   const int i = blockIdx.x * TB + threadIdx.x;
   if (i >= n) { return; }
   const int b = Di[i], e = Di[i + 1], ob = Oi ? Oi[i] : 0, oe = Oi ? Oi[i + 1] : 0;
   if (e <= b) { if (!FILL) { cnt[i] = 0; } return; }
   const double diag = Da[b];
   double row_scale = 0.0, row_sum = diag;
   if (diag < 0)
   {
      for (int k = b + 1; k < e; k++) { const double a = Da[k]; row_scale = fmax(row_scale, a); row_sum += a; }
      for (int k = ob; k < oe; k++) { const double a = Oa[k]; row_scale = fmax(row_scale, a); row_sum += a; }
   }
   else
   {
      for (int k = b + 1; k < e; k++) { const double a = Da[k]; row_scale = fmin(row_scale, a); row_sum += a; }
      for (int k = ob; k < oe; k++) { const double a = Oa[k]; row_scale = fmin(row_scale, a); row_sum += a; }
   }
   const bool all_weak = (fabs(row_sum) > fabs(diag) * max_row_sum) && (max_row_sum < 1.0);
   const double bar = theta * row_scale;
   int c = 0, p = FILL ? Si[i] : 0;
   if (!all_weak)
   {
      for (int k = b + 1; k < e; k++)
      {
         const double a = Da[k];
         const bool strong = diag < 0 ? !(a <= bar) : !(a >= bar);
         if (strong) { if (FILL) { Sj[p++] = Dj[k]; } else { c++; } }
      }
      for (int k = ob; k < oe; k++)
      {
         const double a = Oa[k];
         const bool strong = diag < 0 ? !(a <= bar) : !(a >= bar);
         if (strong) { if (FILL) { Sj[p++] = n + Oj[k]; } else { c++; } }
      }
   }
   if (!FILL) { cnt[i] = c; }
}

// ---- PMIS ----------------------------------------------------------------------------------------------------------
constexpr int C_PT = 1, F_PT = -1, SF_PT = -3;

__global__ __launch_bounds__(TB)
void column_count_kernel(const int *__restrict__ Sj, size_t nnz, int *__restrict__ cnt)
{
   const size_t stride = (size_t) gridDim.x * TB;
   for (size_t k = (size_t) blockIdx.x * TB + threadIdx.x; k < nnz; k += stride) { atomicAdd(&cnt[Sj[k]], 1); }
}

// utilities/random.c: seed <- 16807 * seed mod (2^31 - 1); the value is seed / (2^31 - 1).  Row i takes the (i + 1 + skip)-th.
__device__ __forceinline__ unsigned long long mulmod31(unsigned long long a, unsigned long long b)
{
   return (a * b) % 2147483647ull;       // both below 2^31: the product fits 62 bits
}
__global__ __launch_bounds__(TB)
void pmis_init_kernel(int n, const int *__restrict__ Si, const int *__restrict__ colcnt, unsigned seed0, unsigned long long skip,
                      double *__restrict__ measure, int *__restrict__ CF)
{
   const int i = blockIdx.x * TB + threadIdx.x;
   if (i >= n) { return; }
   if (Si[i + 1] - Si[i] == 0)
   {
      // a point that depends on nobody (par_coarsen.c:2271-2285) is out before the iteration starts
      CF[i] = SF_PT; measure[i] = 0.0;
      return;
   }
   unsigned long long p = 1, base = 16807, k = (unsigned long long) i + 1 + skip;
   while (k) { if (k & 1) { p = mulmod31(p, base); } base = mulmod31(base, base); k >>= 1; }
   const unsigned long long s = mulmod31((unsigned long long) seed0, p);
   CF[i] = 0;
   measure[i] = (double) colcnt[i] + (double) (int) s / (double) 2147483647;
}

// candidates: every undecided point that influences someone
__global__ __launch_bounds__(TB)
void pmis_candidates_kernel(int n, const double *__restrict__ measure, int *__restrict__ CF)
{
   const int i = blockIdx.x * TB + threadIdx.x;
   if (i < n && measure[i] > 1) { CF[i] = 1; }
}
// of two candidates joined by a strong connection the smaller leaves the set (the only writes are zeros)
__global__ __launch_bounds__(TB)
void pmis_knockout_kernel(int n, const int *__restrict__ Si, const int *__restrict__ Sj, const double *__restrict__ measure,
                          int *__restrict__ CF)
{
   const int i = blockIdx.x * TB + threadIdx.x;
   if (i >= n) { return; }
   const double mi = measure[i];
   if (!(mi > 1)) { return; }
   for (int k = Si[i]; k < Si[i + 1]; k++)
   {
      const int j = Sj[k];
      const double mj = measure[j];
      if (mj > 1)
      {
         if (mi > mj) { CF[j] = 0; }
         else if (mj > mi) { CF[i] = 0; }
      }
   }
}
// undecided points (measure > 0): in the set -> C; influencing nobody, or depending on a point of the set -> F
__global__ __launch_bounds__(TB)
void pmis_settle_kernel(int n, const int *__restrict__ Si, const int *__restrict__ Sj, const double *__restrict__ measure,
                        int *CF)
{
   const int i = blockIdx.x * TB + threadIdx.x;
   if (i >= n) { return; }
   const double mi = measure[i];
   if (!(mi > 0)) { return; }
   int mine = __hip_atomic_load(&CF[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
   if (mi < 1) { mine = F_PT; }
   if (mine > 0) { mine = C_PT; }
   else
   {
      for (int k = Si[i]; k < Si[i + 1]; k++)
      {
         if (__hip_atomic_load(&CF[Sj[k]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > 0) { mine = F_PT; }
      }
   }
   __hip_atomic_store(&CF[i], mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// decided points leave the graph (measure 0); counts the ones that stay
__global__ __launch_bounds__(TB)
void pmis_retire_kernel(int n, double *__restrict__ measure, const int *__restrict__ CF, int *__restrict__ left)
{
   const int i = blockIdx.x * TB + threadIdx.x;
   int stay = 0;
   if (i < n && measure[i] > 0)
   {
      if (CF[i] != 0) { measure[i] = 0.0; } else { stay = 1; }
   }
   // one atomic per workgroup (16.8 M rows are 262 144 waves: their atomics on one counter took 0.4 ms a sweep)
   __shared__ int wave_count[TB / 64];
   const unsigned long long b = __ballot(stay);
   if ((threadIdx.x & 63) == 0) { wave_count[threadIdx.x >> 6] = __popcll(b); }
   __syncthreads();
   if (threadIdx.x == 0)
   {
      int c = 0;
      for (int w = 0; w < TB / 64; w++) { c += wave_count[w]; }
      if (c) { atomicAdd(left, c); }
   }
}

// ---- PMIS across ranks (par_coarsen.c:2101-2810 with ghost points) --------------------------------------------------------
// ghost points carry the owner's measure; decided ghosts leave the graph like local points
__global__ __launch_bounds__(TB)
void pmis_retire_ghost_kernel(int nco, double *__restrict__ measure_g, const int *__restrict__ CF_g)
{
   const int i = blockIdx.x * TB + threadIdx.x;
   if (i < nco && CF_g[i] != 0) { measure_g[i] = 0.0; }
}
__global__ __launch_bounds__(TB)
void add_counts_kernel(int tot, const int *__restrict__ elmts, const int *__restrict__ buf, int *__restrict__ cnt)
{
   const int k = blockIdx.x * TB + threadIdx.x;
   if (k < tot && buf[k]) { atomicAdd(&cnt[elmts[k]], buf[k]); }
}
__global__ __launch_bounds__(TB)
void gather_double_kernel(int tot, const double *__restrict__ x, const int *__restrict__ elmts, double *__restrict__ out)
{
   const int k = blockIdx.x * TB + threadIdx.x;
   if (k < tot) { out[k] = x[elmts[k]]; }
}
__global__ __launch_bounds__(TB)
void gather_marker_kernel(int tot, const int *__restrict__ x, const int *__restrict__ elmts, int *__restrict__ out)
{
   const int k = blockIdx.x * TB + threadIdx.x;
   if (k < tot) { out[k] = x[elmts[k]]; }
}
// What the neighbours decided about this rank's boundary points (par_coarsen.c:2485-2500): the host walks the send list
//    for k: if (!recv[k] && CF[elmt] > 0) CF[elmt] = 0; else recv[k] = CF[elmt];
// IN ORDER, and a point listed for several neighbours is knocked out from the first slot on that says so — the slots
// before it still report the old marker.  out[k] = what slot k reports; prev[k] = the previous slot of the same point.
__global__ __launch_bounds__(TB)
void pmis_verdict_kernel(int tot, const int *__restrict__ elmts, const int *__restrict__ prev, const int *__restrict__ recv,
                         const int *__restrict__ CF, int *__restrict__ out)
{
   const int k = blockIdx.x * TB + threadIdx.x;
   if (k >= tot) { return; }
   const int orig = CF[elmts[k]];
   bool knocked = false;
   if (orig > 0) { for (int q = k; q >= 0; q = prev[q]) { if (recv[q] == 0) { knocked = true; break; } } }
   out[k] = knocked ? 0 : orig;
}
__global__ __launch_bounds__(TB)
void pmis_apply_verdict_kernel(int tot, const int *__restrict__ elmts, const int *__restrict__ out, int *__restrict__ CF)
{
   const int k = blockIdx.x * TB + threadIdx.x;
   if (k < tot && out[k] == 0 && CF[elmts[k]] > 0) { CF[elmts[k]] = 0; }
}

// ---- coarse numbering: f2c[i] = number of C points before i, or -1 ---------------------------------------------------
__global__ __launch_bounds__(TB)
void cpt_flag_kernel(int n, const int *__restrict__ CF, int *__restrict__ flag)
{
   const int i = blockIdx.x * TB + threadIdx.x;
   if (i < n) { flag[i] = CF[i] >= 0 ? 1 : 0; }
}
__global__ __launch_bounds__(TB)
void cpt_number_kernel(int n, const int *__restrict__ CF, int *__restrict__ f2c)
{
   const int i = blockIdx.x * TB + threadIdx.x;
   if (i < n && CF[i] < 0) { f2c[i] = -1; }
}

// ---- smoother diagonals (ams.c:527-830 without ghost columns) --------------------------------------------------------
__global__ __launch_bounds__(TB)
void l1_norms_kernel(int n, const int *__restrict__ Ai, const int *__restrict__ Aj, const double *__restrict__ Aa, int option,
                     const int *__restrict__ cf, double *__restrict__ out, int *__restrict__ zero_seen)
{
   const int i = blockIdx.x * TB + threadIdx.x;
   if (i >= n) { return; }
   const int b = Ai[i], e = Ai[i + 1];
   double diag = 0.0;
   for (int k = b; k < e; k++) { if (Aj[k] == i) { diag = Aa[k]; break; } }
   double v;
   if (option == 5) { out[i] = diag == 0.0 ? 1.0 : diag; return; }
   if (option == 1)
   {
      v = 0.0;
      for (int k = b; k < e; k++)
      {
         if (cf && cf[i] != cf[Aj[k]]) { continue; }
         v += 1.0 * fabs(Aa[k]);
      }
   }
   else { v = fabs(diag); }          // options 4 and 6 with no ghost columns
   if (diag < 0.0) { v = -v; }
   if (fabs(v) == 0.0) { *zero_seen = 1; }
   out[i] = v;
}

}  // namespace

void launch_scan_exclusive(int *data, int n, hipStream_t s);     // kernels.hip

// S = strong off-diagonal couplings of A, columns in A's stored order.  Si (n + 1) and Sj are allocated here.
void device_strength(int n, const int *Ai, const int *Aj, const double *Aa, double theta, double max_row_sum,
                     int **Si_out, int **Sj_out, int *nnz_out, hipStream_t s)
{
   int *Si = nullptr, *Sj = nullptr;
   HIP_CHECK(hipMalloc((void **) &Si, sizeof(int) * ((size_t) n + 1)));
   hipLaunchKernelGGL((strength_kernel<false>), dim3(grid_for((size_t) n)), dim3(TB), 0, s, n, Ai, Aj, Aa, theta, max_row_sum, Si, nullptr, nullptr);
   launch_scan_exclusive(Si, n, s);
   int nnz = 0;
   HIP_CHECK(hipMemcpyAsync(&nnz, Si + n, sizeof(int), hipMemcpyDeviceToHost, s));
   HIP_CHECK(hipStreamSynchronize(s));
   HIP_CHECK(hipMalloc((void **) &Sj, sizeof(int) * (size_t) std::max(nnz, 1)));
   hipLaunchKernelGGL((strength_kernel<true>), dim3(grid_for((size_t) n)), dim3(TB), 0, s, n, Ai, Aj, Aa, theta, max_row_sum, nullptr, Si, Sj);
   *Si_out = Si; *Sj_out = Sj; *nnz_out = nnz;
}

// PMIS on the graph of S (n x n, device): CF (device, n ints) comes back with 1 (C), -1 (F) or -3 (isolated).
// seed / skip: the generator's seed and the number of values drawn before row 0's (par_indepset.c).
// Returns the number of sweeps.
int device_pmis(int n, const int *Si, const int *Sj, int snnz, unsigned seed, unsigned long long skip, int *CF, hipStream_t s)
{
   if (n <= 0) { return 0; }
   double *measure = nullptr;
   int *cnt = nullptr;
   HIP_CHECK(hipMalloc((void **) &measure, sizeof(double) * (size_t) n));
   HIP_CHECK(hipMalloc((void **) &cnt, sizeof(int) * ((size_t) n + 1)));
   HIP_CHECK(hipMemsetAsync(cnt, 0, sizeof(int) * ((size_t) n + 1), s));
   if (snnz > 0) { hipLaunchKernelGGL(column_count_kernel, dim3(grid_for((size_t) snnz)), dim3(TB), 0, s, Sj, (size_t) snnz, cnt); }
   const int g = grid_for((size_t) n);
   hipLaunchKernelGGL(pmis_init_kernel, dim3(g), dim3(TB), 0, s, n, Si, cnt, seed, skip, measure, CF);
   int *left = cnt + n, sweeps = 0;
   // the number of undecided points before the first sweep
   HIP_CHECK(hipMemsetAsync(left, 0, sizeof(int), s));
   hipLaunchKernelGGL(pmis_retire_kernel, dim3(g), dim3(TB), 0, s, n, measure, CF, left);
   int h_left = 0;
   HIP_CHECK(hipMemcpyAsync(&h_left, left, sizeof(int), hipMemcpyDeviceToHost, s));
   HIP_CHECK(hipStreamSynchronize(s));
   while (h_left > 0)
   {
      hipLaunchKernelGGL(pmis_candidates_kernel, dim3(g), dim3(TB), 0, s, n, measure, CF);
      hipLaunchKernelGGL(pmis_knockout_kernel, dim3(g), dim3(TB), 0, s, n, Si, Sj, measure, CF);
      hipLaunchKernelGGL(pmis_settle_kernel, dim3(g), dim3(TB), 0, s, n, Si, Sj, measure, CF);
      HIP_CHECK(hipMemsetAsync(left, 0, sizeof(int), s));
      hipLaunchKernelGGL(pmis_retire_kernel, dim3(g), dim3(TB), 0, s, n, measure, CF, left);
      HIP_CHECK(hipMemcpyAsync(&h_left, left, sizeof(int), hipMemcpyDeviceToHost, s));
      HIP_CHECK(hipStreamSynchronize(s));
      sweeps++;
      if (sweeps > 10000) { break; }       // cannot happen: every sweep settles the undecided point of largest measure
   }
   HIP_CHECK(hipFree(measure));
   HIP_CHECK(hipFree(cnt));
   return sweeps;
}

// S of a rank's two blocks, as ONE pattern over the extended numbering (local columns, then n + ghost column)
void device_strength_blocks(int n, const int *Di, const int *Dj, const double *Da, const int *Oi, const int *Oj, const double *Oa,
                            double theta, double max_row_sum, int **Si_out, int **Sj_out, int *nnz_out, hipStream_t s)
{
   int *Si = nullptr, *Sj = nullptr;
   HIP_CHECK(hipMalloc((void **) &Si, sizeof(int) * ((size_t) n + 1)));
   HIP_CHECK(hipMemsetAsync(Si, 0, sizeof(int) * ((size_t) n + 1), s));
   if (n > 0) { hipLaunchKernelGGL((strength_blocks_kernel<false>), dim3(grid_for((size_t) n)), dim3(TB), 0, s, n, Di, Dj, Da, Oi, Oj, Oa, theta, max_row_sum, Si, nullptr, nullptr); }
   launch_scan_exclusive(Si, n, s);
   int nnz = 0;
   HIP_CHECK(hipMemcpyAsync(&nnz, Si + n, sizeof(int), hipMemcpyDeviceToHost, s));
   HIP_CHECK(hipStreamSynchronize(s));
   HIP_CHECK(hipMalloc((void **) &Sj, sizeof(int) * (size_t) std::max(nnz, 1)));
   if (n > 0) { hipLaunchKernelGGL((strength_blocks_kernel<true>), dim3(grid_for((size_t) n)), dim3(TB), 0, s, n, Di, Dj, Da, Oi, Oj, Oa, theta, max_row_sum, nullptr, Si, Sj); }
   *Si_out = Si; *Sj_out = Sj; *nnz_out = nnz;
}

// PMIS of a distributed level on the extended graph (n rows; columns below n local, n + g ghost g of the package).
// CF has n + nco entries: the local markers, then the ghosts' (current when the routine returns).  d_elmts / d_prev: the
// package's send list and, per slot, the previous slot naming the same point (-1: none), both on the device.
// Every exchange is the host routine's (par_coarsen.c:2101-2810): column counts of ghost columns to their owners, the
// measures out to the ghosts, and per sweep ghost verdicts in / markers out / settled markers out; the sweep count is
// agreed on through one all-reduce per sweep.  Returns the number of sweeps.
int device_pmis_dist(int n, int nco, const int *Si, const int *Sj, int snnz, unsigned seed, unsigned long long skip,
                     hypre_ParCSRCommPkg *pkg, const int *d_elmts, const int *d_prev, MPI_Comm comm, int *CF, hipStream_t s)
{
   const int tot = pkg ? pkg->send_map_starts[pkg->num_sends] : 0;
   const int next = n + nco;
   double *measure = nullptr, *dbuf = nullptr;
   int *cnt = nullptr, *ibuf = nullptr, *obuf = nullptr;
   HIP_CHECK(hipMalloc((void **) &measure, sizeof(double) * (size_t) std::max(next, 1)));
   HIP_CHECK(hipMalloc((void **) &cnt, sizeof(int) * ((size_t) next + 1)));
   HIP_CHECK(hipMalloc((void **) &dbuf, sizeof(double) * (size_t) std::max(tot, 1)));
   HIP_CHECK(hipMalloc((void **) &ibuf, sizeof(int) * (size_t) std::max(tot, 1)));
   HIP_CHECK(hipMalloc((void **) &obuf, sizeof(int) * (size_t) std::max(tot, 1)));
   HIP_CHECK(hipMemsetAsync(cnt, 0, sizeof(int) * ((size_t) next + 1), s));
   HIP_CHECK(hipMemsetAsync(measure, 0, sizeof(double) * (size_t) std::max(next, 1), s));
   HIP_CHECK(hipMemsetAsync(CF, 0, sizeof(int) * (size_t) std::max(next, 1), s));
   auto exchange = [&](int job, void *send, void *recv)
   {
      if (!pkg) { return; }
      hypre_ParCSRCommHandle *h = hypre_ParCSRCommHandleCreate_v2(job, pkg, HYPRE_MEMORY_DEVICE, send, HYPRE_MEMORY_DEVICE, recv);
      hypre_ParCSRCommHandleDestroy(h);
   };
   if (snnz > 0) { hipLaunchKernelGGL(column_count_kernel, dim3(grid_for((size_t) snnz)), dim3(TB), 0, s, Sj, (size_t) snnz, cnt); }
   // what the neighbours' rows add to this rank's points
   exchange(12, cnt + n, ibuf);
   if (tot > 0) { hipLaunchKernelGGL(add_counts_kernel, dim3(grid_for((size_t) tot)), dim3(TB), 0, s, tot, d_elmts, ibuf, cnt); }
   const int g = grid_for((size_t) std::max(n, 1)), gx = grid_for((size_t) std::max(next, 1)), gg = grid_for((size_t) std::max(nco, 1));
   const int gt = grid_for((size_t) std::max(tot, 1));
   if (n > 0) { hipLaunchKernelGGL(pmis_init_kernel, dim3(g), dim3(TB), 0, s, n, Si, cnt, seed, skip, measure, CF); }
   if (tot > 0) { hipLaunchKernelGGL(gather_double_kernel, dim3(gt), dim3(TB), 0, s, tot, measure, d_elmts, dbuf); }
   exchange(1, dbuf, measure + n);
   int *left = cnt + next, sweeps = 0;
   auto undecided = [&]()
   {
      HIP_CHECK(hipMemsetAsync(left, 0, sizeof(int), s));
      if (n > 0) { hipLaunchKernelGGL(pmis_retire_kernel, dim3(g), dim3(TB), 0, s, n, measure, CF, left); }
      if (nco > 0) { hipLaunchKernelGGL(pmis_retire_ghost_kernel, dim3(gg), dim3(TB), 0, s, nco, measure + n, CF + n); }
      int h_left = 0;
      HIP_CHECK(hipMemcpyAsync(&h_left, left, sizeof(int), hipMemcpyDeviceToHost, s));
      HIP_CHECK(hipStreamSynchronize(s));
      return (long long) llround(global_sum(comm, (double) h_left));
   };
   while (undecided() > 0)
   {
      hipLaunchKernelGGL(pmis_candidates_kernel, dim3(gx), dim3(TB), 0, s, next, measure, CF);
      if (n > 0) { hipLaunchKernelGGL(pmis_knockout_kernel, dim3(g), dim3(TB), 0, s, n, Si, Sj, measure, CF); }
      exchange(12, CF + n, ibuf);
      if (tot > 0)
      {
         hipLaunchKernelGGL(pmis_verdict_kernel, dim3(gt), dim3(TB), 0, s, tot, d_elmts, d_prev, ibuf, CF, obuf);
         hipLaunchKernelGGL(pmis_apply_verdict_kernel, dim3(gt), dim3(TB), 0, s, tot, d_elmts, obuf, CF);
      }
      exchange(11, obuf, CF + n);
      if (n > 0) { hipLaunchKernelGGL(pmis_settle_kernel, dim3(g), dim3(TB), 0, s, n, Si, Sj, measure, CF); }
      if (tot > 0) { hipLaunchKernelGGL(gather_marker_kernel, dim3(gt), dim3(TB), 0, s, tot, CF, d_elmts, obuf); }
      exchange(11, obuf, CF + n);
      sweeps++;
      if (sweeps > 10000) { break; }
   }
   HIP_CHECK(hipStreamSynchronize(s));
   HIP_CHECK(hipFree(measure)); HIP_CHECK(hipFree(cnt)); HIP_CHECK(hipFree(dbuf)); HIP_CHECK(hipFree(ibuf)); HIP_CHECK(hipFree(obuf));
   return sweeps;
}

// f2c[i] = coarse number of point i (CF[i] >= 0) or -1; returns the number of coarse points
int device_coarse_numbering(int n, const int *CF, int *f2c, hipStream_t s)
{
   if (n <= 0) { return 0; }
   int *tmp = nullptr;
   HIP_CHECK(hipMalloc((void **) &tmp, sizeof(int) * ((size_t) n + 1)));
   const int g = grid_for((size_t) n);
   hipLaunchKernelGGL(cpt_flag_kernel, dim3(g), dim3(TB), 0, s, n, CF, tmp);
   launch_scan_exclusive(tmp, n, s);
   int nc = 0;
   HIP_CHECK(hipMemcpyAsync(&nc, tmp + n, sizeof(int), hipMemcpyDeviceToHost, s));
   HIP_CHECK(hipMemcpyAsync(f2c, tmp, sizeof(int) * (size_t) n, hipMemcpyDeviceToDevice, s));
   hipLaunchKernelGGL(cpt_number_kernel, dim3(g), dim3(TB), 0, s, n, CF, f2c);
   HIP_CHECK(hipStreamSynchronize(s));
   HIP_CHECK(hipFree(tmp));
   return nc;
}

// smoother diagonal of a matrix without ghost columns; option as hypre_ParCSRComputeL1Norms (1, 4, 5, 6).
// Returns false when a row came out zero (the host routine flags the argument).
bool device_l1_norms(int n, const int *Ai, const int *Aj, const double *Aa, int option, const int *cf, double *out, hipStream_t s)
{
   if (n <= 0) { return true; }
   int *flag = reinterpret_cast<int *>(reduce_scratch(2));
   HIP_CHECK(hipMemsetAsync(flag, 0, sizeof(int), s));
   hipLaunchKernelGGL(l1_norms_kernel, dim3(grid_for((size_t) n)), dim3(TB), 0, s, n, Ai, Aj, Aa, option, cf, out, flag);
   int h = 0;
   HIP_CHECK(hipMemcpyAsync(&h, flag, sizeof(int), hipMemcpyDeviceToHost, s));
   HIP_CHECK(hipStreamSynchronize(s));
   return h == 0;
}

// The code object of this file is loaded when one of its kernels is first asked for: ensure_device() asks here, so that
// the load (tens of milliseconds per file) is part of bringing the device up, not of the first setup or solve.
void preload_setup_kernels() { hipFuncAttributes at; (void) hipFuncGetAttributes(&at, (const void *) strength_kernel<false>); (void) hipGetLastError(); }

}  // namespace hamd
