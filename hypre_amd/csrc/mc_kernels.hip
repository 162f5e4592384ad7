// hypre_amd — multicolour Gauss-Seidel: colouring and colour classes on the device, and the one-workgroup sweep of
// small levels.
//
// No counterpart in the reference (par_relax_mc.cpp has the parity statement: relax 21 / 22 are the hybrid Gauss-Seidel
// sweeps 3 / 4 of parcsr_ls/par_relax.c:691-945 on the colour-permuted ordering).  The colouring is the greedy
// first-fit one in row order over the pattern of A + A^T,
//      colour(i) = the smallest colour none of i's neighbours j < i carries,
// which gives the two colours of a 7-point grid, eight of a 27-point one, 15 - 40 on the coarse levels — fewer, and
// with better-shaped classes on grids (every other point: half of every cache line used), than colourings by random
// priority (four to six classes of a 7-point grid, a fifth of a line used).  It is a sequential definition, made
// parallel the way the level-scheduled sweeps are: a point is ready once all its lower neighbours are coloured, the
// ready points of a round are coloured at once, and rounds follow each other on the device (each round's list is filled
// by the round before; the host looks in every 32 rounds to see whether everybody is done).  The result does not depend
// on the order in which a round's points are handled.
#include "internal.hpp"
#include <algorithm>

namespace hamd {

namespace {

constexpr int TB = 256;
inline int grid_for(size_t n) { return (int) std::max<size_t>(1, std::min<size_t>((n + TB - 1) / TB, (size_t) 0x7fffffff)); }

// j is coloured before i: in row order, or — when that order makes the rounds too many (a chain of dependencies as long
// as the matrix: one-dimensional problems) — in the order of a hash of the row numbers, whose chains are short
__device__ __forceinline__ bool before(int j, int i, int hashed)
{
   if (!hashed) { return j < i; }
   auto mix = [](unsigned x) { x ^= x >> 16; x *= 0x85ebca6bu; x ^= x >> 13; x *= 0xc2b2ae35u; x ^= x >> 16; return x; };
   const unsigned hj = mix((unsigned) j), hi = mix((unsigned) i);
   return hj < hi || (hj == hi && j < i);
}

// lower neighbours of every point (an entry of A and its mirror in A^T count once each: the releases below count the same way)
__global__ __launch_bounds__(TB)
void mc_count_lower_kernel(int n, const int *__restrict__ Ai, const int *__restrict__ Aj, const int *__restrict__ Ti,
                           const int *__restrict__ Tj, int *__restrict__ cnt, int *__restrict__ color, int *__restrict__ list,
                           int *__restrict__ sizes, int hashed)
{
   const int i = blockIdx.x * TB + threadIdx.x;
   if (i >= n) { return; }
   int c = 0;
   for (int k = Ai[i]; k < Ai[i + 1]; k++) { const int j = Aj[k]; if (j >= 0 && j < n && j != i && before(j, i, hashed)) { c++; } }
   for (int k = Ti[i]; k < Ti[i + 1]; k++) { const int j = Tj[k]; if (j != i && before(j, i, hashed)) { c++; } }
   cnt[i] = c;
   color[i] = -1;
   if (c == 0) { list[atomicAdd(&sizes[0], 1)] = i; }
}

// One round: colour the ready points, release their later neighbours.  G lanes share a point (8 for short rows, a whole
// wave for the 30 - 90 entries of a coarse level's row): a lane per point walked its two neighbour lists one dependent load
// after the other — 180 entries at 0.5 us each, 200 us a round, a second a hierarchy.
template <int G>
__global__ __launch_bounds__(TB)
void mc_round_kernel(int n, const int *__restrict__ Ai, const int *__restrict__ Aj, const int *__restrict__ Ti,
                     const int *__restrict__ Tj, int *cnt, int *color, const int *__restrict__ cur, int *__restrict__ next,
                     const int *__restrict__ size_cur, int *size_next, int *num_colors, int *done, int hashed)
{
   const int m = *size_cur;
   const int lane = threadIdx.x & 63, gl = threadIdx.x & (G - 1);
   const int groups = gridDim.x * (TB / G), g0 = (blockIdx.x * TB + threadIdx.x) / G;
   const int passes = (m + groups - 1) / groups;
   int cmax = 0, ndone = 0;
   for (int pss = 0; pss < passes; pss++)
   {
      const int t = pss * groups + g0;
      const bool have = t < m;
      const int i = have ? cur[t] : 0;
      const int a0 = have ? Ai[i] : 0, na = have ? Ai[i + 1] - a0 : 0;
      const int b0 = have ? Ti[i] : 0, nb = have ? Ti[i + 1] - b0 : 0;
      // the smallest colour no earlier neighbour carries: 64 colours at a time, the group's lanes over the entries
      int c = -1;
      for (int base = 0; have && c < 0; base += 64)
      {
         unsigned long long used = 0ull;
         for (int k = gl; k < na + nb; k += G)
         {
            const int j = k < na ? Aj[a0 + k] : Tj[b0 + k - na];
            if (j >= 0 && j < n && j != i && before(j, i, hashed)) { const int cj = color[j] - base; if (cj >= 0 && cj < 64) { used |= 1ull << cj; } }
         }
         for (int off = G >> 1; off > 0; off >>= 1)
         {
            const unsigned lo = __shfl_xor((unsigned) used, off, 64), hi = __shfl_xor((unsigned) (used >> 32), off, 64);
            used |= ((unsigned long long) hi << 32) | lo;
         }
         if (~used) { c = base + __ffsll((long long) ~used) - 1; }
      }
      if (have && gl == 0) { color[i] = c; cmax = max(cmax, c + 1); ndone++; }
      // release the later neighbours: the group's lanes over the entries, one reservation in the next list per wave and step
      int steps = (na + nb + G - 1) / G;
      for (int off = 32; off > 0; off >>= 1) { steps = max(steps, __shfl_xor(steps, off, 64)); }
      for (int st = 0; st < steps; st++)
      {
         const int k = st * G + gl;
         int w = -1;
         bool rel = false;
         if (k < na + nb)
         {
            const int v = k < na ? Aj[a0 + k] : Tj[b0 + k - na];
            if (v >= 0 && v < n && v != i && before(i, v, hashed)) { w = v; rel = atomicSub(&cnt[v], 1) == 1; }
         }
         const unsigned long long rb = __ballot(rel);
         if (rb)
         {
            int base = 0;
            if (lane == 0) { base = atomicAdd(size_next, __popcll(rb)); }
            base = __shfl(base, 0, 64);
            if (rel) { next[base + __popcll(rb & ((1ull << lane) - 1ull))] = w; }
         }
      }
   }
   // (a point coloured here is read by the NEXT round only: kernel boundary)
   for (int off = 32; off > 0; off >>= 1) { cmax = max(cmax, __shfl_xor(cmax, off, 64)); ndone += __shfl_xor(ndone, off, 64); }
   if (lane == 0 && ndone) { atomicMax(num_colors, cmax); atomicAdd(done, ndone); }
}

// Rows listed colour by colour, ascending inside a colour, in two passes over the colours x segments of the row range:
// pass 0 counts the rows of colour c in segment g, pass 1 (bases known) writes them.
__global__ __launch_bounds__(TB)
void mc_classes_kernel(int n, const int *__restrict__ color, int seg_len, int nseg, int pass, int *__restrict__ counts,
                       const int *__restrict__ bases, int *__restrict__ order)
{
   const int g = blockIdx.x, c = blockIdx.y;
   const int lo = g * seg_len, hi = min(n, lo + seg_len);
   __shared__ int wave_cnt[TB / 64];
   __shared__ int running;
   if (threadIdx.x == 0) { running = pass ? bases[c * nseg + g] : 0; }
   __syncthreads();
   const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
   for (int i0 = lo; i0 < hi; i0 += TB)
   {
      const int i = i0 + threadIdx.x;
      const bool mine = i < hi && color[i] == c;
      const unsigned long long b = __ballot(mine);
      if (lane == 0) { wave_cnt[wave] = __popcll(b); }
      __syncthreads();
      int before_me = running;
      for (int w = 0; w < wave; w++) { before_me += wave_cnt[w]; }
      if (pass && mine) { order[before_me + __popcll(b & ((1ull << lane) - 1ull))] = i; }
      __syncthreads();
      if (threadIdx.x == 0) { int t = 0; for (int w = 0; w < TB / 64; w++) { t += wave_cnt[w]; } running += t; }
      __syncthreads();
   }
   if (!pass && threadIdx.x == 0) { counts[c * nseg + g] = running; }
}
// lengths of the rows in that order
__global__ __launch_bounds__(TB)
void mc_lens_kernel(int n, const int *__restrict__ order, const int *__restrict__ Ai, int *__restrict__ len)
{
   const int r = blockIdx.x * TB + threadIdx.x;
   if (r < n) { const int i = order[r]; len[r] = Ai[i + 1] - Ai[i]; }
}
// colour-sorted copy of the matrix: row r of the order goes to entries G[r] + shift[colour of r] .. (each colour's slice
// starts at a multiple of four entries); per-colour row pointers, zero-based, colour c's at ptr + cstart[c] + c
__global__ __launch_bounds__(TB)
void mc_copy_kernel(int n, const int *__restrict__ order, const int *__restrict__ color, const int *__restrict__ cstart,
                    const int *__restrict__ G, const int *__restrict__ slice0, const int *__restrict__ Ai, const int *__restrict__ Aj,
                    const double *__restrict__ Aa, int *__restrict__ ptr, int *__restrict__ Cj, double *__restrict__ Ca)
{
   const int r = blockIdx.x * TB + threadIdx.x;
   if (r >= n) { return; }
   const int i = order[r], c = color[i];
   const int rel = G[r] - G[cstart[c]];                   // zero-based inside the colour
   ptr[r + c] = rel;
   if (r + 1 == cstart[c + 1]) { ptr[r + c + 1] = G[r + 1] - G[cstart[c]]; }
   int p = slice0[c] + rel;
   for (int k = Ai[i]; k < Ai[i + 1]; k++) { Cj[p] = Aj[k]; Ca[p] = Aa[k]; p++; }
}
__global__ __launch_bounds__(TB)
void mc_diag_kernel(int n, const int *__restrict__ Ai, const double *__restrict__ Aa, double *__restrict__ d)
{
   const int i = blockIdx.x * TB + threadIdx.x;
   if (i >= n) { return; }
   double v = 1.0;
   if (Ai[i + 1] > Ai[i] && Aa[Ai[i]] != 0.0) { v = Aa[Ai[i]]; }
   d[i] = v;
}

// All colours of a small level in ONE workgroup: colour after colour, eight lanes per row (128 rows at a time), a
// workgroup barrier between colours.  order[] lists the rows colour by colour (cstart[c] .. cstart[c + 1]); the sweep works
// on the level's own matrix.  What a colour costs is one chain of dependent loads (row, entries, u) and the barrier —
// about half of what the launch of a kernel of its own costs, which is all there is to win on levels whose colours hold a
// few dozen rows.
constexpr int SMALL_TB = 1024;
__global__ __launch_bounds__(SMALL_TB)
void mc_small_sweep_kernel(int num_colors, int direction, const int *__restrict__ cstart, const int *__restrict__ order,
                           const int *__restrict__ Ai, const int *__restrict__ Aj, const double *__restrict__ Aa,
                           const float *__restrict__ Aa32, const double *__restrict__ f, const double *__restrict__ d,
                           const int *__restrict__ marker, int marker_val, double w, double *u)
{
   constexpr int GL = 8, NG = SMALL_TB / GL;
   const int gl = threadIdx.x & (GL - 1), grp = threadIdx.x / GL;
   for (int q = 0; q < num_colors; q++)
   {
      const int c = direction > 0 ? q : num_colors - 1 - q;
      const int r0 = cstart[c], r1 = cstart[c + 1];
      for (int rb = r0; rb < r1; rb += NG)
      {
         const int r = rb + grp;
         const bool have = r < r1;
         const int i = have ? order[r] : 0;
         const bool on = have && !(marker && marker[i] != marker_val);
         double s = 0.0;
         if (on)
         {
            const int b = Ai[i], e = Ai[i + 1];
            if (Aa32) { for (int k = b + gl; k < e; k += GL) { s += (double) Aa32[k] * u[Aj[k]]; } }
            else { for (int k = b + gl; k < e; k += GL) { s += Aa[k] * u[Aj[k]]; } }
         }
         s += __shfl_xor(s, 4, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 1, 64);
         if (on && gl == 0) { u[i] = u[i] + w * (f[i] - s) / d[i]; }
      }
      __syncthreads();
   }
}

// The same sweep with what does not depend on the iterate fetched AHEAD.  In the kernel above a colour is a chain of five
// dependent loads (colour bounds -> row -> row pointers -> entries -> u) and a barrier: ~8 us a colour, 158 us a launch on
// the benchmark hierarchy (round 4 trace), 1.3 ms of a 6.9 ms cycle.  Of that chain only the gathers of u have to wait for
// the colour before: a row's place (mc_rowinfo: row, first and last entry, one 16-byte record per sorted position), its
// right-hand side, diagonal, own iterate value (only the row itself writes it) and its first entries are requested one
// pass ahead, while the current pass is summed — a colour then costs the gathers and the barrier.
// And a row gets 32 lanes instead of 8: the colours of these levels hold a few dozen rows at most (1003 rows in 44 colours,
// 185 in 47 on the benchmark hierarchy), so 8 lanes a row left most of the workgroup idle and ten entries a lane to walk;
// with 32 the four entries a lane fetches ahead cover rows of 128 entries — no second trip for the rest of a row.
// Same arithmetic; a row's sum is added by 32 lanes (entries gl, gl + 32, ... per lane, then the xor tree) instead of 8.
template <int PB>
struct McAhead
{
   int    i, b, e, on;
   double f, d, uo;
   double v[PB];
   int    c[PB];
};
struct McPass { int q, rb, r1; };      // colour (in sweep order), first row of the pass, end of the colour; q >= num_colors: none
// The look-ahead is two passes deep: a row's record (mc_rowinfo) is requested two passes ahead, what hangs on it (right-hand
// side, diagonal, own value, first entries) one pass ahead — by then the record has arrived with the loads of the pass
// before, so nothing in the loop waits for a load it has just issued except the gathers of u, the one thing that must.
// ULDS: the level's iterate (at most 8192 rows) lives in LDS for the length of the sweep — copied in at the start, written
// back at the end —, so what a colour hands to the next (a store that has to reach the L2, a barrier, gathers that come
// back from the L2: two round trips a colour) becomes LDS traffic.
template <int MC_GL, int MC_PB, bool ULDS>
__global__ __launch_bounds__(SMALL_TB)
void mc_small_sweep_ahead_kernel(int num_colors, int direction, const int *__restrict__ cstart, const int4 *__restrict__ rowinfo, int nrows,
                                 const int *__restrict__ Aj, const double *__restrict__ Aa, const float *__restrict__ Aa32, int nnz,
                                 const double *__restrict__ f, const double *__restrict__ d, const int *__restrict__ marker, int marker_val,
                                 double w, double *ug)
{
   extern __shared__ __align__(16) double mc_ulds[];
   constexpr int GL = MC_GL, NG = SMALL_TB / GL;
   const int gl = threadIdx.x & (GL - 1), grp = threadIdx.x / GL;
   double *u = ug;
   if (ULDS)
   {
      for (int i = threadIdx.x; i < nrows; i += SMALL_TB) { mc_ulds[i] = ug[i]; }
      u = mc_ulds;
   }
   auto colour = [&](int q) { return direction > 0 ? q : num_colors - 1 - q; };
   // the pass after p (the next rows of its colour, or the first of the next colour that holds rows)
   auto after = [&](McPass p) -> McPass
   {
      if (p.q >= num_colors) { return p; }
      p.rb += NG;
      while (p.rb >= p.r1)
      {
         p.q++;
         if (p.q >= num_colors) { break; }
         const int c = colour(p.q);
         p.rb = cstart[c]; p.r1 = cstart[c + 1];
      }
      return p;
   };
   auto record = [&](const McPass &p) -> int4 { return rowinfo[min(max(p.rb + grp, 0), nrows - 1)]; };
   // what a pass needs of this lane's row beyond its record
   auto fetch = [&](const McPass &p, const int4 &ri) -> McAhead<MC_PB>
   {
      McAhead<MC_PB> a;
      const bool have = p.rb + grp < p.r1;
      a.i = ri.x; a.b = ri.y; a.e = ri.z;
      a.on = (have && !(marker && marker[a.i] != marker_val)) ? 1 : 0;
      a.f = f[a.i]; a.d = d[a.i]; a.uo = ULDS ? 0.0 : ug[a.i];
#pragma unroll
      for (int j = 0; j < MC_PB; j++)
      {
         const int k = min(max(a.b + gl + j * GL, 0), nnz - 1);
         a.v[j] = Aa32 ? (double) Aa32[k] : Aa[k];
         a.c[j] = Aj[k];
      }
      return a;
   };
   McPass p0 = after(McPass{-1, 0, 0});     // (from q = -1 with no rows left, "after" walks to the first colour that holds rows)
   if (p0.q >= num_colors) { return; }
   if (ULDS) { __syncthreads(); }
   McPass p1 = after(p0), p2 = after(p1);
   McAhead<MC_PB> cur = fetch(p0, record(p0));
   int4 ri1 = p1.q < num_colors ? record(p1) : make_int4(0, 0, 0, 0);
   while (true)
   {
      int4 ri2 = make_int4(0, 0, 0, 0);
      if (p2.q < num_colors) { ri2 = record(p2); }
      McAhead<MC_PB> nxt = cur;
      if (p1.q < num_colors) { nxt = fetch(p1, ri1); }
      // this pass: gathers, sums, the row's update
      double s = 0.0;
      if (cur.on)
      {
#pragma unroll
         for (int j = 0; j < MC_PB; j++) { if (cur.b + gl + j * GL < cur.e) { s += cur.v[j] * u[cur.c[j]]; } }
         if (Aa32) { for (int k = cur.b + gl + MC_PB * GL; k < cur.e; k += GL) { s += (double) Aa32[k] * u[Aj[k]]; } }
         else { for (int k = cur.b + gl + MC_PB * GL; k < cur.e; k += GL) { s += Aa[k] * u[Aj[k]]; } }
      }
#pragma unroll
      for (int off = GL >> 1; off > 0; off >>= 1) { s += __shfl_xor(s, off, 64); }
      if (cur.on && gl == 0) { u[cur.i] = (ULDS ? u[cur.i] : cur.uo) + w * (cur.f - s) / cur.d; }
      if (p1.q >= num_colors) { break; }
      if (p1.q != p0.q) { __syncthreads(); }          // the next colour reads what this one wrote
      cur = nxt; p0 = p1; p1 = p2; ri1 = ri2; p2 = after(p2);
   }
   if (ULDS)
   {
      __syncthreads();
      for (int i = threadIdx.x; i < nrows; i += SMALL_TB) { ug[i] = mc_ulds[i]; }
   }
}

// The sweep of a small level out of LDS: the iterate (all n rows of the level), and for the swept rows their record, right-hand
// side and diagonal live in LDS for the length of the launch — filled in two trips at the start — so a colour hands over to the
// next through LDS and the only global loads of the loop are the rows' entries, requested three passes ahead (they depend on
// nothing the sweep changes).  A colour then costs a barrier and LDS latencies instead of two round trips to the L2.
template <int GL, int PB>
__global__ __launch_bounds__(SMALL_TB)
void mc_small_sweep_lds_kernel(int num_colors, int direction, const int *__restrict__ cstart, const int4 *__restrict__ rowinfo, int nrows,
                               int r_begin, int m, const int *__restrict__ Aj, const double *__restrict__ Aa, const float *__restrict__ Aa32, int nnz,
                               const double *__restrict__ f, const double *__restrict__ d, const int *__restrict__ marker, int marker_val,
                               double w, double *ug)
{
   extern __shared__ __align__(16) unsigned char mc_smem[];
   constexpr int NG = SMALL_TB / GL, DEPTH = 3;
   double *uL = reinterpret_cast<double *>(mc_smem);
   double *fL = uL + nrows, *dL = fL + m;
   int4 *infoL = reinterpret_cast<int4 *>(dL + m + ((nrows + 2 * m) & 1));      // (16-byte aligned)
   int *csL = reinterpret_cast<int *>(infoL + m);
   const int tid = threadIdx.x, gl = tid & (GL - 1), grp = tid / GL;
   for (int i = tid; i < nrows; i += SMALL_TB) { uL[i] = ug[i]; }
   for (int i = tid; i <= num_colors; i += SMALL_TB) { csL[i] = cstart[i]; }
   for (int r = tid; r < m; r += SMALL_TB)
   {
      int4 ri = rowinfo[min(r_begin + r, nrows - 1)];
      ri.w = (marker && marker[ri.x] != marker_val) ? 0 : 1;
      infoL[r] = ri;
      fL[r] = f[ri.x];
      dL[r] = d[ri.x];
   }
   __syncthreads();
   auto colour = [&](int q) { return direction > 0 ? q : num_colors - 1 - q; };
   struct Pass { int q, rb, r1; };
   auto after = [&](Pass p) -> Pass
   {
      if (p.q >= num_colors) { return p; }
      p.rb += NG;
      while (p.rb >= p.r1)
      {
         p.q++;
         if (p.q >= num_colors) { break; }
         const int c = colour(p.q);
         p.rb = csL[c]; p.r1 = csL[c + 1];
      }
      return p;
   };
   struct Entries { double v[PB]; int c[PB]; };
   // the first entries of this lane's row in pass p (any valid address for lanes without a row: never used)
   auto entries = [&](const Pass &p) -> Entries
   {
      Entries e;
      const int r = min(max(p.rb + grp - r_begin, 0), m - 1);
      const int b = (p.q < num_colors) ? infoL[r].y : 0;
#pragma unroll
      for (int j = 0; j < PB; j++)
      {
         const int k = min(max(b + gl + j * GL, 0), nnz - 1);
         e.v[j] = Aa32 ? (double) Aa32[k] : Aa[k];
         e.c[j] = Aj[k];
      }
      return e;
   };
   Pass p[DEPTH + 1];
   p[0] = after(Pass{-1, 0, 0});
   if (p[0].q < num_colors)
   {
#pragma unroll
      for (int k = 1; k <= DEPTH; k++) { p[k] = after(p[k - 1]); }
      // three slots of entries, each consumed in place and then asked for again (for the pass three ahead): no register of
      // a slot is copied, so nothing waits for a load that was only just issued
      Entries E0 = entries(p[0]), E1 = entries(p[1]), E2 = entries(p[2]);
      auto step = [&](Entries &E) -> bool
      {
         const int r = p[0].rb + grp;
         const bool have = r < p[0].r1;
         const int rl = min(max(r - r_begin, 0), m - 1);
         const int4 ri = infoL[rl];
         const bool on = have && ri.w != 0;
         double s = 0.0;
         if (on)
         {
#pragma unroll
            for (int j = 0; j < PB; j++) { if (ri.y + gl + j * GL < ri.z) { s += E.v[j] * uL[E.c[j]]; } }
            if (Aa32) { for (int k = ri.y + gl + PB * GL; k < ri.z; k += GL) { s += (double) Aa32[k] * uL[Aj[k]]; } }
            else { for (int k = ri.y + gl + PB * GL; k < ri.z; k += GL) { s += Aa[k] * uL[Aj[k]]; } }
         }
#pragma unroll
         for (int off = GL >> 1; off > 0; off >>= 1) { s += __shfl_xor(s, off, 64); }
         if (on && gl == 0) { uL[ri.x] = uL[ri.x] + w * (fL[rl] - s) / dL[rl]; }
         if (p[1].q >= num_colors) { return false; }
         if (p[1].q != p[0].q) { __syncthreads(); }       // the next colour reads what this one wrote
         E = entries(p[DEPTH]);
#pragma unroll
         for (int k = 0; k < DEPTH; k++) { p[k] = p[k + 1]; }
         p[DEPTH] = after(p[DEPTH]);
         return true;
      };
      while (step(E0) && step(E1) && step(E2)) { }
   }
   __syncthreads();
   for (int i = tid; i < nrows; i += SMALL_TB) { ug[i] = uL[i]; }
}

__global__ void mc_rowinfo_kernel(int n, const int *__restrict__ order, const int *__restrict__ Ai, int4 *__restrict__ info)
{
   const int r = blockIdx.x * blockDim.x + threadIdx.x;
   if (r >= n) { return; }
   const int i = order[r];
   info[r] = make_int4(i, Ai[i], Ai[i + 1], 0);
}

}  // namespace

// rows colour by colour -> (row, first entry, last entry) records of the look-ahead sweep
void launch_mc_rowinfo(int n, const int *order, const int *Ai, void *info, hipStream_t s)
{
   if (n > 0) { hipLaunchKernelGGL(mc_rowinfo_kernel, dim3((n + 255) / 256), dim3(256), 0, s, n, order, Ai, (int4 *) info); }
}

// Greedy first-fit colouring in row order of the pattern of A + A^T (both given as device CSR patterns).  color: n ints
// (device).  Returns the number of colours.  round_budget: rounds after which the row order is given up for the hashed one.
int device_greedy_coloring(int n, const int *Ai, const int *Aj, const int *Ti, const int *Tj, int *color, int round_budget, hipStream_t s,
                           int *rounds_out)
{
   if (rounds_out) { *rounds_out = 0; }
   if (n <= 0) { return 0; }
   int *cnt = nullptr, *lists = nullptr, *sizes = nullptr, *scal = nullptr;
   HIP_CHECK(hipMalloc((void **) &cnt, sizeof(int) * (size_t) n));
   HIP_CHECK(hipMalloc((void **) &lists, sizeof(int) * 2 * (size_t) n));
   HIP_CHECK(hipMalloc((void **) &scal, sizeof(int) * 2));
   // a round's list holds a diagonal plane of a grid at most (tens of thousands of points), usually far fewer: a small grid
   // with a strided loop keeps a round at the cost of a launch
   const int grid = std::min(grid_for((size_t) n), 256);
   // lanes per point by the mean row length (the two lists of a point together are twice that)
   int nnz_h = 0;
   HIP_CHECK(hipMemcpyAsync(&nnz_h, Ai + n, sizeof(int), hipMemcpyDeviceToHost, s));
   HIP_CHECK(hipStreamSynchronize(s));
   const double mean = (double) nnz_h / (double) n;
   const int width = mean <= 10.0 ? 8 : (mean <= 40.0 ? 32 : 64);
   int h[2] = {0, 0};
   for (int hashed = 0; hashed < 2; hashed++)
   {
      const int max_rounds = (hashed ? n : std::min(n, round_budget)) + 64;
      HIP_CHECK(hipMalloc((void **) &sizes, sizeof(int) * (size_t) max_rounds));
      HIP_CHECK(hipMemsetAsync(sizes, 0, sizeof(int) * (size_t) max_rounds, s));
      HIP_CHECK(hipMemsetAsync(scal, 0, sizeof(int) * 2, s));
      hipLaunchKernelGGL(mc_count_lower_kernel, dim3(grid_for((size_t) n)), dim3(TB), 0, s, n, Ai, Aj, Ti, Tj, cnt, color, lists, sizes, hashed);
      int round = 0;
      h[0] = h[1] = 0;
      while (round < max_rounds - 1)
      {
         for (int b = 0; b < 32 && round < max_rounds - 1; b++, round++)
         {
            int *cur = lists + (size_t) (round & 1) * (size_t) n, *nxt = lists + (size_t) ((round + 1) & 1) * (size_t) n;
            if (width == 64)
            {
               hipLaunchKernelGGL((mc_round_kernel<64>), dim3(grid), dim3(TB), 0, s, n, Ai, Aj, Ti, Tj, cnt, color, cur, nxt, sizes + round,
                                  sizes + round + 1, scal, scal + 1, hashed);
            }
            else if (width == 32)
            {
               hipLaunchKernelGGL((mc_round_kernel<32>), dim3(grid), dim3(TB), 0, s, n, Ai, Aj, Ti, Tj, cnt, color, cur, nxt, sizes + round,
                                  sizes + round + 1, scal, scal + 1, hashed);
            }
            else
            {
               hipLaunchKernelGGL((mc_round_kernel<8>), dim3(grid), dim3(TB), 0, s, n, Ai, Aj, Ti, Tj, cnt, color, cur, nxt, sizes + round,
                                  sizes + round + 1, scal, scal + 1, hashed);
            }
         }
         HIP_CHECK(hipMemcpyAsync(h, scal, sizeof(int) * 2, hipMemcpyDeviceToHost, s));
         HIP_CHECK(hipStreamSynchronize(s));
         if (h[1] >= n) { break; }
      }
      HIP_CHECK(hipFree(sizes));
      if (rounds_out) { *rounds_out += round; }
      if (h[1] >= n) { break; }
   }
   HIP_CHECK(hipFree(cnt)); HIP_CHECK(hipFree(lists)); HIP_CHECK(hipFree(scal));
   if (h[1] < n) { hypre_error_w_msg(HYPRE_ERROR_GENERIC, "multicolour Gauss-Seidel: the colouring did not reach every row"); }
   return h[0];
}

// The rows colour by colour (ascending inside a colour): order (device, n ints, allocated here) and the start of every
// colour in it (host).
void device_color_order(int n, int C, const int *color, int **order_out, std::vector<int> &cstart, hipStream_t s)
{
   int *order = nullptr;
   HIP_CHECK(hipMalloc((void **) &order, sizeof(int) * (size_t) std::max(n, 1)));
   cstart.assign((size_t) C + 1, 0);
   if (n > 0 && C > 0)
   {
      const int nseg = std::max(1, std::min(1024, (n + 4095) / 4096));
      const int seg_len = (n + nseg - 1) / nseg;
      int *d_tab = nullptr;
      HIP_CHECK(hipMalloc((void **) &d_tab, sizeof(int) * (size_t) C * (size_t) nseg));
      hipLaunchKernelGGL(mc_classes_kernel, dim3(nseg, C), dim3(TB), 0, s, n, color, seg_len, nseg, 0, d_tab, (const int *) nullptr, (int *) nullptr);
      std::vector<int> tab((size_t) C * (size_t) nseg);
      HIP_CHECK(hipMemcpyAsync(tab.data(), d_tab, sizeof(int) * tab.size(), hipMemcpyDeviceToHost, s));
      HIP_CHECK(hipStreamSynchronize(s));
      int run = 0;
      for (int c = 0; c < C; c++)
      {
         cstart[(size_t) c] = run;
         for (int g = 0; g < nseg; g++) { const int k = tab[(size_t) c * nseg + g]; tab[(size_t) c * nseg + g] = run; run += k; }
      }
      cstart[(size_t) C] = run;
      HIP_CHECK(hipMemcpyAsync(d_tab, tab.data(), sizeof(int) * tab.size(), hipMemcpyHostToDevice, s));
      hipLaunchKernelGGL(mc_classes_kernel, dim3(nseg, C), dim3(TB), 0, s, n, color, seg_len, nseg, 1, (int *) nullptr, d_tab, order);
      HIP_CHECK(hipStreamSynchronize(s));
      HIP_CHECK(hipFree(d_tab));
   }
   *order_out = order;
}

// The colour classes as matrices: ONE colour-sorted copy of A (every colour's slice starting at a multiple of four
// entries) and ONE array of zero-based row pointers (colour c's at ptr + cstart[c] + c).  slice0 (host): first entry of
// every colour's slice.  The arrays are device allocations the caller owns.
void device_color_matrices(int n, int C, const int *color, const int *order, const std::vector<int> &cstart, const int *Ai, const int *Aj,
                           const double *Aa, int **ptr_out, int **Cj_out, double **Ca_out, std::vector<int> &slice0, std::vector<int> &slice_nnz,
                           hipStream_t s)
{
   int *G = nullptr, *ptr = nullptr, *Cj = nullptr, *d_cstart = nullptr, *d_slice0 = nullptr;
   double *Ca = nullptr;
   HIP_CHECK(hipMalloc((void **) &G, sizeof(int) * ((size_t) n + 1)));
   HIP_CHECK(hipMemsetAsync(G, 0, sizeof(int) * ((size_t) n + 1), s));
   hipLaunchKernelGGL(mc_lens_kernel, dim3(grid_for((size_t) n)), dim3(TB), 0, s, n, order, Ai, G);
   launch_scan_exclusive(G, n, s);
   // entries before every colour: G at the colours' starts
   std::vector<int> gat((size_t) C + 1, 0);
   for (int c = 0; c <= C; c++) { HIP_CHECK(hipMemcpyAsync(&gat[(size_t) c], G + cstart[(size_t) c], sizeof(int), hipMemcpyDeviceToHost, s)); }
   HIP_CHECK(hipStreamSynchronize(s));
   slice0.assign((size_t) C, 0); slice_nnz.assign((size_t) C, 0);
   int run = 0;
   for (int c = 0; c < C; c++)
   {
      slice0[(size_t) c] = run;
      slice_nnz[(size_t) c] = gat[(size_t) c + 1] - gat[(size_t) c];
      run = (run + slice_nnz[(size_t) c] + 3) & ~3;
   }
   HIP_CHECK(hipMalloc((void **) &ptr, sizeof(int) * ((size_t) n + (size_t) C + 1)));
   HIP_CHECK(hipMalloc((void **) &Cj, sizeof(int) * (size_t) std::max(run, 4)));
   HIP_CHECK(hipMalloc((void **) &Ca, sizeof(double) * (size_t) std::max(run, 4)));
   HIP_CHECK(hipMemsetAsync(Cj, 0, sizeof(int) * (size_t) std::max(run, 4), s));
   HIP_CHECK(hipMemsetAsync(Ca, 0, sizeof(double) * (size_t) std::max(run, 4), s));
   HIP_CHECK(hipMemsetAsync(ptr, 0, sizeof(int) * ((size_t) n + (size_t) C + 1), s));
   HIP_CHECK(hipMalloc((void **) &d_cstart, sizeof(int) * ((size_t) C + 1)));
   HIP_CHECK(hipMalloc((void **) &d_slice0, sizeof(int) * (size_t) std::max(C, 1)));
   HIP_CHECK(hipMemcpyAsync(d_cstart, cstart.data(), sizeof(int) * ((size_t) C + 1), hipMemcpyHostToDevice, s));
   HIP_CHECK(hipMemcpyAsync(d_slice0, slice0.data(), sizeof(int) * (size_t) C, hipMemcpyHostToDevice, s));
   if (n > 0) { hipLaunchKernelGGL(mc_copy_kernel, dim3(grid_for((size_t) n)), dim3(TB), 0, s, n, order, color, d_cstart, G, d_slice0, Ai, Aj, Aa, ptr, Cj, Ca); }
   HIP_CHECK(hipStreamSynchronize(s));
   HIP_CHECK(hipFree(G)); HIP_CHECK(hipFree(d_cstart)); HIP_CHECK(hipFree(d_slice0));
   *ptr_out = ptr; *Cj_out = Cj; *Ca_out = Ca;
}

void launch_mc_diag(int n, const int *Ai, const double *Aa, double *d, hipStream_t s)
{
   if (n > 0) { hipLaunchKernelGGL(mc_diag_kernel, dim3(grid_for((size_t) n)), dim3(TB), 0, s, n, Ai, Aa, d); }
}

void launch_mc_small_sweep(int num_colors, int direction, const int *cstart, const int *order, const int *Ai, const int *Aj,
                           const double *Aa, const float *Aa32, const double *f, const double *d, const int *marker, int marker_val,
                           double w, double *u, int n, int nnz, hipStream_t s)
{
   if (num_colors <= 0) { return; }
   account_bytes((double) nnz * (Aa32 ? 8.0 : 12.0) + (double) n * 44.0);
   hipLaunchKernelGGL(mc_small_sweep_kernel, dim3(1), dim3(SMALL_TB), 0, s, num_colors, direction, cstart, order, Ai, Aj, Aa, Aa32, f, d,
                      marker, marker_val, w, u);
}

// the same with the look-ahead kernels: rowinfo holds (row, first entry, last entry) for every position of the colour order
// (nrows of them: the level's rows), the matrix nnz_matrix entries; the launch sweeps the n rows from position r_first on
void launch_mc_small_sweep_ahead(int num_colors, int direction, const int *cstart, const void *rowinfo, int nrows, const int *Aj,
                                 const double *Aa, const float *Aa32, int nnz_matrix, const double *f, const double *d, const int *marker,
                                 int marker_val, double w, double *u, int n, int nnz, int r_first, hipStream_t s)
{
   if (num_colors <= 0 || nrows <= 0 || nnz_matrix <= 0 || n <= 0) { return; }
   account_bytes((double) nnz * (Aa32 ? 8.0 : 12.0) + (double) n * 44.0);
   // a level of at most 12 288 rows whose swept rows' records fit beside its iterate: the sweep out of LDS
   const int r_begin = r_first;
   const size_t lds = sizeof(double) * ((size_t) nrows + 2 * (size_t) n + 2) + 16 * (size_t) n + sizeof(int) * ((size_t) num_colors + 2);
   if (nrows <= 12288 && lds <= (size_t) 150 * 1024)
   {
      static bool raised = false;
      if (!raised)
      {
         (void) hipFuncSetAttribute((const void *) (mc_small_sweep_lds_kernel<32, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
         (void) hipGetLastError();
         raised = true;
      }
      hipLaunchKernelGGL((mc_small_sweep_lds_kernel<32, 4>), dim3(1), dim3(SMALL_TB), lds, s, num_colors, direction, cstart, (const int4 *) rowinfo, nrows,
                         r_begin, n, Aj, Aa, Aa32, nnz_matrix, f, d, marker, marker_val, w, u);
      return;
   }
   if (nrows <= 8192)
   {
      hipLaunchKernelGGL((mc_small_sweep_ahead_kernel<32, 4, true>), dim3(1), dim3(SMALL_TB), sizeof(double) * (size_t) nrows, s, num_colors, direction, cstart,
                         (const int4 *) rowinfo, nrows, Aj, Aa, Aa32, nnz_matrix, f, d, marker, marker_val, w, u);
   }
   else
   {
      hipLaunchKernelGGL((mc_small_sweep_ahead_kernel<32, 4, false>), dim3(1), dim3(SMALL_TB), 0, s, num_colors, direction, cstart,
                         (const int4 *) rowinfo, nrows, Aj, Aa, Aa32, nnz_matrix, f, d, marker, marker_val, w, u);
   }
}

void preload_mc_kernels() { hipFuncAttributes at; (void) hipFuncGetAttributes(&at, (const void *) mc_round_kernel<64>); (void) hipGetLastError(); }

}  // namespace hamd
