// The smallest levels of a V(1,1) cycle with Jacobi-type smoothing in ONE kernel of one workgroup.
//
// From the level where an operator holds a few thousand entries down to the coarsest and back up, every step of the cycle
// (par_cycle.c:23-803: residual, restriction, sweep from zero, the dense solve, interpolation, sweep) is a launch of 5 us
// for a fraction of a microsecond of work — 13 launches and 76 us of the 1.94 ms of a 256^3 cycle (trace of the bench
// command, round 4), a third of a 128^3 cycle.  The steps depend on each other, so a graph does not shorten them; one
// workgroup that walks them with a barrier in between does: the operators of these levels are a few hundred KB, one CU
// streams them in a microsecond or two.
//
// Arithmetic: every product rounded, a row's products added by W lanes (W a power of two chosen per level so that the rows
// of a pass fill the workgroup): lane t of a row adds entries t, t + W, ... in stored order, the W partial sums go through
// the xor tree; W = 1 is the stored order of the reference's host loop.  Epilogues as the row epilogues of
// spmv_kernels.hip (OP_AXPBY, OP_AXPBY_DIV, OP_JACOBI) and scaled_div_kernel; the dense solve as coarse_solve_kernel
// (utilities/gselim.h order, no fused multiply-adds).  The order of a row's sum differs from the tiled / row-slice kernels'
// — by rounding only.
#include "internal.hpp"
#include "amg_internal.hpp"

#include <hip/hip_runtime.h>

namespace hamd {

namespace {

constexpr int TAIL_THREADS = 1024;         // lanes of the workgroup; the register form runs 768 (twelve waves: 170 registers a lane)
constexpr int TAIL_THREADS_REG = 768;
#ifndef TAIL_TIMING
#define TAIL_TIMING 0          // experiment: ticks of the 100 MHz wall clock at the steps of the walk (hypre_amd_TailTiming)
#endif
#if TAIL_TIMING
__device__ unsigned long long tail_stamps[64];
#define TAIL_STAMP(k) do { if (threadIdx.x == 0) { tail_stamps[k] = wall_clock64(); } } while (0)
#else
#define TAIL_STAMP(k)
#endif

__device__ __forceinline__ double tail_value(const double *a, int k, int r32)
{
   const double v = a[k];
   return r32 ? (double) (float) v : v;
}

// sum of row [s, e) of (j, a) times x with W lanes (this lane: sub), NB entries in flight: 4 out of LDS, more when the
// operator streams from global memory: 24 (a lane's share of a row is then one trip, or two)
enum { T_RESID = 0, T_RESTRICT = 1, T_RESTRICT_DIV = 2, T_INTERP_ADD = 3, T_JACOBI = 4, T_RESID_RD = 5, T_L_FIRST = 6, T_L_NEXT = 7 };
// the passes over the strictly lower triangle (the inner steps of the two-stage Gauss-Seidel sweep): entries left of the diagonal only
template <int OPK> struct tail_lower { static constexpr bool value = OPK == T_L_FIRST || OPK == T_L_NEXT; };

// what a row's sum becomes: the row epilogues of spmv_kernels.hip (and scaled_div / scaled_recip), rounding for rounding
template <int OPK>
__device__ __forceinline__ void tail_epilogue(int row, double sum, const double *x, const double *f, const double *d, double w,
                                              double *y, double *u, int flag)
{
   if (OPK == T_RESID) { y[row] = __fma_rn(1.0, f[row], __dmul_rn(-1.0, sum)); }                  // OP_AXPBY, alpha -1, beta 1
   else if (OPK == T_RESTRICT) { y[row] = __dmul_rn(1.0, sum); }                                     // OP_AXPBY, alpha 1, beta 0
   else if (OPK == T_RESTRICT_DIV) { const double r = __dmul_rn(1.0, sum); y[row] = r; u[row] = __dmul_rn(w, r) / d[row]; }
   else if (OPK == T_INTERP_ADD) { y[row] = __fma_rn(1.0, y[row], __dmul_rn(1.0, sum)); }            // OP_AXPBY in place, beta 1
   else if (OPK == T_JACOBI)
   {
      // OP_JACOBI: y = x + (w f - w (A x)) / d, out of place
      const double t = __fma_rn(w, f[row], -__dmul_rn(w, sum));
      y[row] = __dadd_rn(x[row], t / d[row]);
   }
   else if (OPK == T_RESID_RD)
   {
      // OP_RESID_RD, alpha -w, beta w: z = (w f - w A u) .* (1 ./ D)
      double r = __dmul_rn(-w, sum);
      r = __fma_rn(w, f[row], r);
      y[row] = __dmul_rn(r, 1.0 / d[row]);
   }
   else if (OPK == T_L_FIRST)
   {
      // OP_TSGS_FIRST: z' = (L z) .* (1 ./ D) ; u = (u + z) + mult z'   (flag: u is known to be zero; w is mult)
      const double z = __dmul_rn(sum, 1.0 / d[row]);
      y[row] = z;
      u[row] = __fma_rn(w, z, __dadd_rn(flag ? 0.0 : u[row], x[row]));
   }
   else
   {
      // OP_TSGS: z'' = (L z') .* (1 ./ D) ; u += mult z''
      const double z = __dmul_rn(sum, 1.0 / d[row]);
      y[row] = z;
      u[row] = __fma_rn(w, z, u[row]);
   }
}

// sum of row [s, e) of (j, a) times x with W lanes (this lane: sub), NB entries in flight: 4 out of LDS, more when the
// operator streams from global memory: 24 (a lane's share of a row is then one trip, or two); LOWER: columns below `row` only
template <int NB, bool LOWER>
__device__ __forceinline__ double tail_row_sum(const int *__restrict__ j, const double *__restrict__ a, int s, int e, int sub, int W,
                                               const double *x, int r32, int row)
{
   double sum = 0.0;
   for (int k = s + sub; k < e; k += NB * W)
   {
      double v[NB];
      int c[NB];
#pragma unroll
      for (int i = 0; i < NB; i++)
      {
         const int q = min(k + i * W, e - 1);
         v[i] = tail_value(a, q, r32);
         c[i] = j[q];
      }
#pragma unroll
      for (int i = 0; i < NB; i++)
      {
         if (k + i * W < e && (!LOWER || c[i] < row))
         {
#pragma clang fp contract(off)
            const double pr = v[i] * x[c[i]];
            sum = sum + pr;
         }
      }
   }
   for (int off = W >> 1; off > 0; off >>= 1) { sum += __shfl_xor(sum, off, 64); }
   return sum;
}

// one matrix pass over n rows: y (and u) from the row sums, W lanes per row.  Row pointers, x, f, d, y, u in LDS; the
// columns and values in LDS (NB = 4) or where the matrix lies (NB = 24)
template <int OPK, int NB>
__device__ __forceinline__ void tail_pass(const int *__restrict__ Mi, const int *__restrict__ Mj, const double *__restrict__ Ma, int n, int W,
                                          const double *x, const double *f, const double *d, double w, double *y, double *u, int r32,
                                          int flag = 0)
{
   const int tid = threadIdx.x, sub = tid & (W - 1), G = (int) blockDim.x / W;
   for (int base = 0; base < n; base += G)
   {
      const int row = base + tid / W;
      const bool live = row < n;
      const int s = live ? Mi[row] : 0, e = live ? Mi[row + 1] : 0;
      const double sum = tail_row_sum<NB, tail_lower<OPK>::value>(Mj, Ma, s, e, sub, W, x, r32, row);
      if (live && sub == 0) { tail_epilogue<OPK>(row, sum, x, f, d, w, y, u, flag); }
   }
}

// The first level's operator out of REGISTERS: when its rows times their lanes fill the workgroup once (n W <= 1024) and a
// lane's share of a row is at most TAIL_REG entries (the workgroup then runs twelve waves instead of sixteen, so that a lane
// may keep 3 x 32 registers of entries), the lane fetches that share once — at the start, beside the image —
// and every pass over the operator (residual, sweeps) multiplies from registers: the largest operator of the tail costs no
// LDS and no second trip.
constexpr int TAIL_REG = 32;
template <int OPK, int NR>
__device__ __forceinline__ void tail_pass_reg(const double (&av)[NR], const int (&ac)[NR], int s, int e, int n, int W,
                                              const double *x, const double *f, const double *d, double w, double *y, double *u, int flag = 0)
{
   const int tid = threadIdx.x, sub = tid & (W - 1), row = tid / W;
   double sum = 0.0;
#pragma unroll
   for (int i = 0; i < NR; i++)
   {
      if (s + sub + i * W < e && (!tail_lower<OPK>::value || ac[i] < row))
      {
#pragma clang fp contract(off)
         const double pr = av[i] * x[ac[i]];
         sum = sum + pr;
      }
   }
   for (int off = W >> 1; off > 0; off >>= 1) { sum += __shfl_xor(sum, off, 64); }
   if (row < n && sub == 0) { tail_epilogue<OPK>(row, sum, x, f, d, w, y, u, flag); }
}

// Everything the walk reads lies in LDS: the plan's image of the levels' arrays (operators, interpolation, restriction,
// smoother diagonals, the coarse factors; built once per hierarchy, laid out as the kernel addresses it) is copied in with
// one batch of 16-byte loads per lane — one trip to memory — and every later step costs LDS latencies, not a round trip to
// L2 or HBM per dependent load (a first version that walked the arrays in global memory took as long as the launches it
// replaced: three dependent loads a pass, a dozen passes).
template <bool REG>
__global__ __launch_bounds__(REG ? TAIL_THREADS_REG : TAIL_THREADS)
void small_tail_kernel(SmallTailArgs t)
{
   constexpr int NT = REG ? TAIL_THREADS_REG : TAIL_THREADS;
   extern __shared__ __align__(16) unsigned char smem[];
   const int tid = threadIdx.x, nl = t.nl, r32 = t.round32;
   TAIL_STAMP(0);
   // the first operator's share of this lane (register form): row bounds now, the entries behind the image's loads
   double av[REG ? TAIL_REG : 1];
   int ac[REG ? TAIL_REG : 1], rs0 = 0, re0 = 0;
   if (REG)
   {
      const int W0 = t.lv[0].wA, row0 = tid / W0;
      if (row0 < t.lv[0].n) { rs0 = t.gAi[row0]; re0 = t.gAi[row0 + 1]; }
   }
   {
      const uint4 *src = reinterpret_cast<const uint4 *>(t.image);
      uint4 *dst = reinterpret_cast<uint4 *>(smem);
      const int quads = t.image_bytes >> 4;
      for (int base = 0; base < quads; base += 8 * NT)
      {
         uint4 v[8];
#pragma unroll
         for (int i = 0; i < 8; i++) { v[i] = src[min(base + i * NT + tid, quads - 1)]; }
#pragma unroll
         for (int i = 0; i < 8; i++) { if (base + i * NT + tid < quads) { dst[base + i * NT + tid] = v[i]; } }
      }
   }
   if (!REG && t.lv[0].gAj)
   {
      // a streamed operator: one load per 128-byte line of its columns and values now, with the image's loads, so that the
      // two passes over it find the lines (and their page) in the L2 instead of paying for them on the critical path
      const int last = t.nnz0 - 1;
      for (int q = 16 * tid; q <= last; q += 16 * NT)
      {
         const double v0 = t.lv[0].gAa[q];
         const int c0 = t.lv[0].gAj[min(2 * q, last)], c1 = t.lv[0].gAj[min(2 * q + 32 * NT, last)];
         asm volatile("" :: "v"(v0), "v"(c0), "v"(c1));
      }
   }
   if (REG)
   {
      const int W0 = t.lv[0].wA, sub0 = tid & (W0 - 1), last = t.nnz0 - 1;
#pragma unroll
      for (int i = 0; i < (REG ? TAIL_REG : 1); i++)
      {
         const int q = min(max(rs0 + sub0 + i * W0, 0), last);
         av[i] = tail_value(t.lv[0].gAa, q, r32);
         ac[i] = t.lv[0].gAj[q];
      }
   }
#define ip(off) (reinterpret_cast<const int *>(smem + (off)))
#define dp(off) (reinterpret_cast<const double *>(smem + (off)))
#define wp(off) (reinterpret_cast<double *>(smem + (off)))
   double *vt = wp(t.vt_off);
   {
      // the first level's right-hand side and, when the restriction into it wrote it, its sweep from zero
      const SmallTailLevel &L0 = t.lv[0];
      double *f0 = wp(L0.f), *u0 = wp(L0.u);
      for (int i = tid; i < L0.n; i += NT) { f0[i] = t.f_in[i]; if (t.first_presmoothed) { u0[i] = t.u_io[i]; } }
   }
   __syncthreads();
   TAIL_STAMP(1);
   if (!t.first_presmoothed && t.kind_down == 0)
   {
      // u = (w f) / d (scaled_div_kernel)
      const SmallTailLevel &L0 = t.lv[0];
      const double *f0 = dp(L0.f), *d0 = dp(L0.d);
      double *u0 = wp(L0.u);
      for (int i = tid; i < L0.n; i += NT) { u0[i] = (L0.w * f0[i]) / d0[i]; }
      __syncthreads();
   }
   // a pass over the operator of level l in whatever form the kernel holds it
#define A_PASS(OPK, l, F, x, f, d, w, y, u, flag)                                                                                     \
   do {                                                                                                                               \
      if (REG && (l) == 0) { tail_pass_reg<OPK, (REG ? TAIL_REG : 1)>(av, ac, rs0, re0, (F).n, (F).wA, x, f, d, w, y, u, flag); }       \
      else if (!REG && (F).gAj) { tail_pass<OPK, 24>(ip((F).Ai), (F).gAj, (F).gAa, (F).n, (F).wA, x, f, d, w, y, u, r32, flag); }      \
      else { tail_pass<OPK, 4>(ip((F).Ai), ip((F).Aj), dp((F).Aa), (F).n, (F).wA, x, f, d, w, y, u, r32, flag); }                     \
   } while (0)
   // two-stage Gauss-Seidel sweep of level l in place (hypre_BoomerAMGRelaxTwoStageGaussSeidelDevice, fused form):
   //   z = (w f - w A u) .* (1 ./ D)  [from zero: (w f) .* (1 ./ D)] ; z' = (L z) .* (1 ./ D), u = (u + z) - z' ; [z'' = (L z') .* (1 ./ D), u += z'']
   auto tsgs = [&](int l, bool from_zero, int inner)
   {
      const SmallTailLevel &F = t.lv[l];
      double *z0 = wp(F.alt), *z1 = vt;
      if (from_zero)
      {
         const double *f = dp(F.f), *d = dp(F.d);
         for (int i = tid; i < F.n; i += NT) { z0[i] = __dmul_rn(__dmul_rn(F.w, f[i]), 1.0 / d[i]); }      // scaled_recip_kernel
      }
      else { A_PASS(T_RESID_RD, l, F, dp(F.u), dp(F.f), dp(F.d), F.w, z0, nullptr, 0); }
      __syncthreads();
      A_PASS(T_L_FIRST, l, F, z0, nullptr, dp(F.d), -1.0, z1, wp(F.u), from_zero ? 1 : 0);
      __syncthreads();
      if (inner > 1)
      {
         A_PASS(T_L_NEXT, l, F, z1, nullptr, dp(F.d), 1.0, z0, wp(F.u), 0);
         __syncthreads();
      }
   };
   if (t.kind_down == 1) { tsgs(0, true, t.inner_down); }
   // down: residual, restriction (+ the next level's sweep from zero)
   for (int l = 0; l < nl - 1; l++)
   {
      const SmallTailLevel &F = t.lv[l], &C = t.lv[l + 1];
      TAIL_STAMP(2 + 2 * l);
      A_PASS(T_RESID, l, F, dp(F.u), dp(F.f), nullptr, 0.0, vt, nullptr, 0);
      __syncthreads();
      TAIL_STAMP(3 + 2 * l);
      if (l + 1 < nl - 1 && t.kind_down == 0) { tail_pass<T_RESTRICT_DIV, 4>(ip(F.Ri), ip(F.Rj), dp(F.Ra), C.n, F.wR, vt, nullptr, dp(C.d), C.w, wp(C.f), wp(C.u), r32); }
      else { tail_pass<T_RESTRICT, 4>(ip(F.Ri), ip(F.Rj), dp(F.Ra), C.n, F.wR, vt, nullptr, nullptr, 0.0, wp(C.f), nullptr, r32); }
      __syncthreads();
      if (l + 1 < nl - 1 && t.kind_down == 1) { tsgs(l + 1, true, t.inner_down); }
   }
   // the coarsest level: substitution with the factors of the pivot-free elimination (coarse_solve_kernel)
   TAIL_STAMP(20);
   {
      const SmallTailLevel &C = t.lv[nl - 1];
      const int n = t.ncoarse;
      const double *lu = dp(t.lu_off);
      double *xc = wp(C.u);
      for (int i = tid; i < n; i += NT) { xc[i] = dp(C.f)[i]; }
      __syncthreads();
      // lane j of the first wave holds x[j]: step k of the elimination updates every j > k at once (x[k] from lane k), the
      // back substitution every j < k — each x[j] sees the operations of the one-lane loop in its order (the same bits)
      if (tid < 64)
      {
         const int j = tid;
         double x = j < n ? xc[j] : 0.0;
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
         if (j < n) { xc[j] = x; }
      }
      __syncthreads();
   }
   // up: interpolation, sweep (Jacobi: out of place into the level's second buffer; two-stage Gauss-Seidel: in place)
   TAIL_STAMP(21);
   for (int l = nl - 2; l >= 0; l--)
   {
      const SmallTailLevel &F = t.lv[l], &C = t.lv[l + 1];
      TAIL_STAMP(22 + 2 * l);
      tail_pass<T_INTERP_ADD, 4>(ip(F.Pi), ip(F.Pj), dp(F.Pa), F.n, F.wP, dp(C.u), nullptr, nullptr, 0.0, wp(F.u), nullptr, r32);
      __syncthreads();
      TAIL_STAMP(23 + 2 * l);
      if (t.kind_up == 1) { tsgs(l, false, t.inner_up); }
      else
      {
         A_PASS(T_JACOBI, l, F, dp(F.u), dp(F.f), dp(F.d), F.w, wp(F.alt), nullptr, 0);
         __syncthreads();
         if (l > 0)
         {
            double *u = wp(F.u);
            const double *a = dp(F.alt);
            for (int i = tid; i < F.n; i += NT) { u[i] = a[i]; }
            __syncthreads();
         }
      }
   }
   {
      const SmallTailLevel &L0 = t.lv[0];
      const double *a = t.kind_up == 1 ? dp(L0.u) : dp(L0.alt);
      for (int i = tid; i < L0.n; i += NT) { t.u_io[i] = a[i]; }
   }
#undef A_PASS
   TAIL_STAMP(40);
#undef ip
#undef dp
#undef wp
}

}  // namespace

#if TAIL_TIMING
extern "C" void hypre_amd_TailTiming(unsigned long long *out)
{
   HIP_CHECK(hipDeviceSynchronize());
   HIP_CHECK(hipMemcpyFromSymbol(out, HIP_SYMBOL(tail_stamps), sizeof(unsigned long long) * 64));
}
#endif
void launch_small_tail(const SmallTailArgs &t, hipStream_t s)
{
   static bool raised = false;
   if (!raised)
   {
      (void) hipFuncSetAttribute((const void *) small_tail_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      (void) hipFuncSetAttribute((const void *) small_tail_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      (void) hipGetLastError();
      raised = true;
   }
   if (t.reg_first) { hipLaunchKernelGGL(small_tail_kernel<true>, dim3(1), dim3(TAIL_THREADS_REG), (size_t) t.lds_bytes, s, t); }
   else { hipLaunchKernelGGL(small_tail_kernel<false>, dim3(1), dim3(TAIL_THREADS), (size_t) t.lds_bytes, s, t); }
}

}  // namespace hamd
