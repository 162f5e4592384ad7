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

// One row, G lanes.  Every lane of the group holds the same (i, s, e, di, fi); lane gl loads entry
// s + gl of each G-wide chunk (jj0/a0: the first chunk, fetched ahead by the caller when PRE) and
// forms its product; the group then folds the products in stored order, every lane running the same
// chain (shuffles inside the group), so the sum is the sequential one bit for bit.
// entries of a row fetched ahead of its turn: one chunk, two for the 32-lane groups (coarse-level rows
// often run past 32 entries and a late chunk costs two more dependent round trips)
template <int G> struct GsPre { static constexpr int N = (G >= 32) ? 2 : 1; int ii[N]; double av[N]; };

template <int G>
__device__ __forceinline__ void gs_row_group(const GsArgs &a, int i, int s, int e, double di, double fi, int gl,
                                             const GsPre<G> &pre)
{
   int ns, ne;
   gs_block_of(a.n, a.threads, i, ns, ne);
   double res = fi, res0 = 0.0, res2 = 0.0;
   for (int base = s; base < e; base += G)
   {
      const int jj = base + gl;
      double pa = 0.0, pb = 0.0;       // padding lanes contribute x - 0.0 = x
      int    inblk = 0;
      if (jj < e)
      {
         int ii; double av;
         if (base == s) { ii = pre.ii[0]; av = pre.av[0]; }
         else if (GsPre<G>::N > 1 && base == s + G) { ii = pre.ii[GsPre<G>::N - 1]; av = pre.av[GsPre<G>::N - 1]; }
         else { ii = a.Dj[jj]; av = a.Da[jj]; }
         if (ii >= ns && ii < ne)
         {
            const double v = ((a.dir > 0) ? (ii < i) : (ii > i)) ? a.u[ii] : a.uold[ii];
            pa = av * v;
            inblk = 1;
            if (!a.non_scale) { pb = av * a.vtemp[ii]; }
         }
         else { pa = av * a.vtemp[ii]; }
      }
      if (a.non_scale)
      {
#pragma unroll
         for (int k = 0; k < G; k++) { res -= __shfl(pa, k, G); }
      }
      else
      {
#pragma unroll
         for (int k = 0; k < G; k++)
         {
            const double qa = __shfl(pa, k, G), qb = __shfl(pb, k, G);
            if (__shfl(inblk, k, G)) { res0 -= qa; res2 += qb; }
            else { res -= qa; }
         }
      }
   }
   if (a.Oi)
   {
      const int os = a.Oi[i], oe = a.Oi[i + 1];
      for (int base = os; base < oe; base += G)
      {
         const int jj = base + gl;
         const double pa = (jj < oe) ? a.Oa[jj] * a.vext[a.Oj[jj]] : 0.0;
#pragma unroll
         for (int k = 0; k < G; k++) { res -= __shfl(pa, k, G); }
      }
   }
   if (gl != 0) { return; }
   if (a.non_scale)
   {
      const double q = res / di;
      a.u[i] = a.skip_diag ? q : a.uold[i] + q;
   }
   else
   {
      const double one_minus_omega = 1.0 - a.omega, prod = 1.0 - a.w * a.omega;
      double un = a.uold[i];
      if (a.skip_diag) { un *= prod; }
      un += a.w * (a.omega * res + res0 + one_minus_omega * res2) / di;
      a.u[i] = un;
   }
}

// what a row needs before it can gather: its span, smoother diagonal, right-hand side, and whether
// the sweep touches it at all (par_relax.h: relax_points / zero diagonal)
template <int G> struct GsRowHead { int i, s, e, live; double di, fi; GsPre<G> pre; };

template <int G>
__device__ __forceinline__ GsRowHead<G> gs_row_head(const GsArgs &a, int4 sc, int gl, bool valid)
{
   GsRowHead<G> h;
   h.i = sc.x; h.s = sc.y + a.skip_diag; h.e = sc.z; h.live = 0; h.di = 1.0; h.fi = 0.0;
#pragma unroll
   for (int c = 0; c < GsPre<G>::N; c++) { h.pre.ii[c] = 0; h.pre.av[c] = 0.0; }
   if (!valid) { return h; }
   h.di = a.l1 ? a.l1[h.i] : a.Da[sc.y];
   h.live = ((a.relax_points == 0 || a.cf[h.i] == a.relax_points) && h.di != 0.0) ? 1 : 0;
   h.fi = a.f[h.i];
#pragma unroll
   for (int c = 0; c < GsPre<G>::N; c++)
   {
      const int jj = h.s + c * G + gl;
      if (jj < h.e) { h.pre.ii[c] = a.Dj[jj]; h.pre.av[c] = a.Da[jj]; }
   }
   return h;
}

// one large level: G lanes per row, as many workgroups as the level needs
template <int G>
__global__ __launch_bounds__(256)
void gs_level_kernel(GsArgs a, int start, int count)
{
   const int slot = (int) ((blockIdx.x * blockDim.x + threadIdx.x) / G), gl = (int) (threadIdx.x % G);
   if (slot >= count) { return; }
   const GsRowHead<G> h = gs_row_head<G>(a, a.sched[start + slot], gl, true);
   if (h.live) { gs_row_group<G>(a, h.i, h.s, h.e, h.di, h.fi, gl, h.pre); }
}

// A run of levels inside one workgroup.  The run is cut into steps of blockDim/G rows that never
// straddle a level; a barrier closes every level.  Three steps are in flight: the schedule entry of
// step t+2 and the row heads of step t+1 (matrix data only: nothing a sweep writes) are fetched
// while step t gathers u, so a step costs one memory round trip instead of four.
template <int G>
__global__ __launch_bounds__(1024)
void gs_run_kernel(GsArgs a, const int *__restrict__ lev_start, int lev_begin, int lev_end)
{
   extern __shared__ int s_lev[];                  // lev_start[lev_begin .. lev_end]
   const int nl = lev_end - lev_begin;
   for (int t = (int) threadIdx.x; t <= nl; t += (int) blockDim.x) { s_lev[t] = lev_start[lev_begin + t]; }
   __syncthreads();
   const int slots = (int) blockDim.x / G, slot = (int) threadIdx.x / G, gl = (int) threadIdx.x % G;
   const int last = s_lev[nl];

   // step = rows [pos, end) of level lev (relative index); pos == last: past the run
   int posA = s_lev[0], levA = 0;
   auto step_end = [&](int pos, int lev) { return min(pos + slots, s_lev[lev + 1]); };
   auto advance = [&](int &pos, int &lev) {
      pos = step_end(pos, lev);
      while (lev < nl && pos >= s_lev[lev + 1]) { lev++; }
   };
   while (levA < nl && posA >= s_lev[levA + 1]) { levA++; }          // leading empty levels
   auto fetch_sched = [&](int pos, int lev) {
      int4 sc = make_int4(0, 0, 0, 0);
      if (lev < nl && pos + slot < step_end(pos, lev)) { sc = a.sched[pos + slot]; sc.w = 1; }
      return sc;
   };

   int4 scA = fetch_sched(posA, levA);
   int  posB = posA, levB = levA; int4 scB = scA;
   advance(posA, levA); scA = fetch_sched(posA, levA);
   GsRowHead<G> hB = gs_row_head<G>(a, scB, gl, scB.w != 0);
   while (posB < last && levB < nl)
   {
      const int posD = posB, levD = levB; const GsRowHead<G> hD = hB;
      posB = posA; levB = levA; scB = scA;
      advance(posA, levA); scA = fetch_sched(posA, levA);
      hB = gs_row_head<G>(a, scB, gl, scB.w != 0);
      if (hD.live) { gs_row_group<G>(a, hD.i, hD.s, hD.e, hD.di, hD.fi, gl, hD.pre); }
      if (step_end(posD, levD) >= s_lev[levD + 1])
      {
         __threadfence_block();
         __syncthreads();
      }
   }
}

template <int G>
static void launch_level_g(const GsArgs &a, int start, int count, hipStream_t s)
{
   const long threads = (long) count * G;
   hipLaunchKernelGGL(gs_level_kernel<G>, dim3((unsigned) ((threads + 255) / 256)), dim3(256), 0, s, a, start, count);
}

void launch_gs_level(const GsArgs &a, int lanes, int start, int count, hipStream_t s)
{
   if (count <= 0) { return; }
   switch (lanes)
   {
      case 4: launch_level_g<4>(a, start, count, s); break;
      case 8: launch_level_g<8>(a, start, count, s); break;
      case 16: launch_level_g<16>(a, start, count, s); break;
      default: launch_level_g<32>(a, start, count, s); break;
   }
}

void launch_gs_run(const GsArgs &a, int lanes, const int *d_lev_start, int lev_begin, int lev_end, hipStream_t s)
{
   if (lev_end <= lev_begin) { return; }
   const size_t lds = sizeof(int) * (size_t) (lev_end - lev_begin + 1);
   switch (lanes)
   {
      case 4: hipLaunchKernelGGL(gs_run_kernel<4>, dim3(1), dim3(1024), lds, s, a, d_lev_start, lev_begin, lev_end); break;
      case 8: hipLaunchKernelGGL(gs_run_kernel<8>, dim3(1), dim3(1024), lds, s, a, d_lev_start, lev_begin, lev_end); break;
      case 16: hipLaunchKernelGGL(gs_run_kernel<16>, dim3(1), dim3(1024), lds, s, a, d_lev_start, lev_begin, lev_end); break;
      default: hipLaunchKernelGGL(gs_run_kernel<32>, dim3(1), dim3(1024), lds, s, a, d_lev_start, lev_begin, lev_end); break;
   }
}

// The code object of this file is loaded when one of its kernels is first asked for: ensure_device() asks here, so that
// the load (tens of milliseconds per file) is part of bringing the device up, not of the first setup or solve.
void preload_gs_kernels() { hipFuncAttributes at; (void) hipFuncGetAttributes(&at, (const void *) gs_level_kernel<8>); (void) hipGetLastError(); }

}  // namespace hamd
