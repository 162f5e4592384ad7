// hypre_amd — Galerkin product A_c = R A P on the device (single rank), bit-identical to the host setup.
//
// Reference: parcsr_ls/par_rap.c:30-2000 (hypre_BoomerAMGBuildCoarseOperatorKT) — row ic of A_c is formed as
//    RA  = sum_{i1 in R(ic,:)} r * A(i1,:)        columns kept in first-touch order,
//    row = RA * P                                  columns in first-touch order behind the diagonal slot,
// every value accumulated in exactly that nesting order.  The host setup of this library (par_amg_setup.cpp) keeps that
// order because the C/F splittings of the coarser levels — and with them the iteration counts and complexities the
// reference's regression files pin — depend on the bits of A_c.  So does this kernel: one WAVE owns a coarse row and
// walks R(ic,:) and RA in order; the lanes share the work INSIDE a step (the entries of one row of A, or of one row of
// P, have distinct columns: they can be looked up, appended — in entry order, by ballot and prefix count — and
// accumulated side by side without changing any sum).  The two column -> position maps of a row live in LDS as
// open-addressing tables whose slots carry the row's number, so nothing is cleared between rows.  Two passes over the
// rows: lengths, then (after a scan) columns and values.
//
// The reference's own device product (parcsr_mv/par_csr_triplemat_device.c, hypre's spgemm hash kernels) is a general
// row-wise SpGEMM whose summation order is whatever the hash probing yields; it is what this has to beat in time, not
// in form.
#include "internal.hpp"
#include <algorithm>
#include <omp.h>

// r * a and the accumulation are separate roundings on the host (no fused multiply-add in the host build): keep them so
#pragma clang fp contract(off)

namespace hamd {

namespace {

// (row tag << 32 | key) in one 64-bit LDS word; a slot whose tag is not the current row's is free
__device__ __forceinline__ int table_find(const unsigned long long *tab, const int *pos, int mask, unsigned tag, int key)
{
   unsigned h = ((unsigned) key * 2654435761u) >> 7;
   while (true)
   {
      const unsigned long long e = tab[h & mask];
      if ((unsigned) (e >> 32) != tag) { return -1; }
      if ((int) (unsigned) e == key) { return pos[h & mask]; }
      h++;
   }
}

// insert a key known to be absent (lanes insert distinct keys side by side)
__device__ __forceinline__ void table_insert(unsigned long long *tab, int *pos, int mask, unsigned tag, int key, int p)
{
   unsigned h = ((unsigned) key * 2654435761u) >> 7;
   const unsigned long long mine = ((unsigned long long) tag << 32) | (unsigned) key;
   while (true)
   {
      const unsigned long long e = tab[h & mask];
      if ((unsigned) (e >> 32) != tag)
      {
         if (atomicCAS(&tab[h & mask], e, mine) == e) { pos[h & mask] = p; return; }
         continue;          // somebody took the slot: look at it again
      }
      h++;
   }
}

__device__ __forceinline__ int lanes_below(unsigned long long ballot, int lane)
{
   return __popcll(ballot & ((1ull << lane) - 1ull));
}

}  // namespace

// inclusive scan of one int per lane over the wave
__device__ __forceinline__ int wave_scan_incl(int v, int lane)
{
   for (int off = 1; off < 64; off <<= 1) { const int u = __shfl_up(v, off, 64); if (lane >= off) { v += u; } }
   return v;
}

// Which of the (up to 64) source rows whose inclusive entry counts are incl[0..63] holds flattened entry t: the first s
// with incl[s] > t.
__device__ __forceinline__ int owner_of(const int *incl, int t)
{
   int lo = 0;
#pragma unroll
   for (int step = 32; step > 0; step >>= 1) { if (incl[lo + step - 1] <= t) { lo += step; } }
   return lo;
}

// FILL = false: row lengths only.  One wave per workgroup; workgroups walk the rows with stride gridDim.
//
// The host loop nests "for every entry of R(ic,:) — for every entry of that row of A" and "for every entry of RA — for
// every entry of that row of P".  Walking it that way costs a round trip to memory per outer entry (pointer, then row),
// 160 of them for a row of the fine level's product and thousands further down, and the kernel spent its time waiting.
// Here the outer entries are taken 64 at a time: their row pointers in one request, then the rows' entries FLATTENED —
// 64 consecutive entries of the concatenated rows per request, whichever rows they belong to — and only the table
// work (look-up, append in entry order, accumulate) still goes row by row, on what already sits in registers.  The
// order of every append and of every sum is the host's.
template <bool FILL>
__global__ __launch_bounds__(64)
void rap_rows_kernel(int nc, int square,
                     const int *__restrict__ Ri, const int *__restrict__ Rj, const double *__restrict__ Ra,
                     const int *__restrict__ Ai, const int *__restrict__ Aj, const double *__restrict__ Aa,
                     const int *__restrict__ Pi, const int *__restrict__ Pj, const double *__restrict__ Pa,
                     int capA, int capRA, int capP, int capO,
                     int *__restrict__ rowlen, const int *__restrict__ Ci, int *__restrict__ Cj, double *__restrict__ Ca,
                     int *overflow, int row_step, int *__restrict__ maxima)
{
   extern __shared__ __align__(16) unsigned char smem[];
   unsigned long long *Akey = reinterpret_cast<unsigned long long *>(smem);
   unsigned long long *Pkey = Akey + capA;
   double *raa = reinterpret_cast<double *>(Pkey + capP);
   double *oa  = raa + capRA;
   double *sval = oa + capO;                   // [64] multiplier of every source row of the batch
   int *Apos = reinterpret_cast<int *>(sval + 64);
   int *Ppos = Apos + capA;
   int *raj  = Ppos + capP;
   int *oj   = raj + capRA;
   int *sincl = oj + capO;                     // [64] inclusive entry counts of the batch's source rows
   int *sbeg  = sincl + 64;                    // [64] where every source row starts in its matrix
   const int lane = threadIdx.x;
   for (int i = lane; i < capA; i += 64) { Akey[i] = 0; }
   for (int i = lane; i < capP; i += 64) { Pkey[i] = 0; }
   __syncthreads();

   unsigned tag = 0;
   // row_step > 1: a pilot over every row_step-th row that only records how long RA and the row get (maxima[0], [1])
   for (int ic = blockIdx.x * row_step; ic < nc; ic += gridDim.x * row_step)
   {
      // some row did not fit the tables: this attempt is lost, leave it to the next one
      if (__hip_atomic_load(overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { break; }
      tag++;                                   // tag 0 = the cleared table
      // ---- RA = sum r * A(i1,:), first-touch order
      int nRA = 0;
      bool bad = false;
      const int r0 = Ri[ic], r1 = Ri[ic + 1];
      for (int rb = r0; rb < r1 && !bad; rb += 64)
      {
         const int j1 = rb + lane;
         const bool src_here = j1 < r1;
         const int i1 = src_here ? Rj[j1] : 0;
         const int a0 = src_here ? Ai[i1] : 0;
         const int len = src_here ? Ai[i1 + 1] - a0 : 0;
         const int incl = wave_scan_incl(len, lane);
         sincl[lane] = incl; sbeg[lane] = a0;
         if (FILL) { sval[lane] = src_here ? Ra[j1] : 0.0; }
         __syncthreads();
         const int total = sincl[63];
         for (int cb = 0; cb < total && !bad; cb += 64)
         {
            const int t = cb + lane;
            const bool have = t < total;
            const int src = have ? owner_of(sincl, t) : 64;
            const int j2 = have ? sbeg[src] + (t - (src ? sincl[src - 1] : 0)) : 0;
            const int i2 = have ? Aj[j2] : -1;
            const double v = (FILL && have) ? sval[src] * Aa[j2] : 0.0;
            const int s_lo = __builtin_amdgcn_readfirstlane(src);
            const int s_hi = __builtin_amdgcn_readfirstlane(owner_of(sincl, min(cb + 63, total - 1)));
            for (int sr = s_lo; sr <= s_hi; sr++)
            {
               const bool mine = have && src == sr;
               if (__ballot(mine) == 0ull) { continue; }
               const int m = mine ? table_find(Akey, Apos, capA - 1, tag, i2) : 0;
               const bool fresh = mine && m < 0;
               const unsigned long long ball = __ballot(fresh);
               const int p = nRA + lanes_below(ball, lane);
               if (nRA + __popcll(ball) > capRA) { bad = true; break; }
               if (fresh)
               {
                  table_insert(Akey, Apos, capA - 1, tag, i2, p);
                  raj[p] = i2;
                  if (FILL) { raa[p] = v; }
               }
               else if (FILL && mine) { raa[m] += v; }
               nRA += __popcll(ball);
               __syncthreads();
            }
         }
         __syncthreads();
      }
      // ---- row = RA * P, first-touch order behind the diagonal slot
      int nO = 0;
      if (!bad)
      {
         if (square)
         {
            if (lane == 0) { table_insert(Pkey, Ppos, capP - 1, tag, ic, 0); oj[0] = ic; if (FILL) { oa[0] = 0.0; } }
            nO = 1;
            __syncthreads();
         }
         for (int qb = 0; qb < nRA && !bad; qb += 64)
         {
            const int q = qb + lane;
            const bool src_here = q < nRA;
            const int i1 = src_here ? raj[q] : 0;
            const int p0 = src_here ? Pi[i1] : 0;
            const int len = src_here ? Pi[i1 + 1] - p0 : 0;
            const int incl = wave_scan_incl(len, lane);
            sincl[lane] = incl; sbeg[lane] = p0;
            if (FILL) { sval[lane] = src_here ? raa[q] : 0.0; }
            __syncthreads();
            const int total = sincl[63];
            for (int cb = 0; cb < total && !bad; cb += 64)
            {
               const int t = cb + lane;
               const bool have = t < total;
               const int src = have ? owner_of(sincl, t) : 64;
               const int j2 = have ? sbeg[src] + (t - (src ? sincl[src - 1] : 0)) : 0;
               const int i2 = have ? Pj[j2] : -1;
               const double v = (FILL && have) ? sval[src] * Pa[j2] : 0.0;
               const int s_lo = __builtin_amdgcn_readfirstlane(src);
               const int s_hi = __builtin_amdgcn_readfirstlane(owner_of(sincl, min(cb + 63, total - 1)));
               for (int sr = s_lo; sr <= s_hi; sr++)
               {
                  const bool mine = have && src == sr;
                  if (__ballot(mine) == 0ull) { continue; }
                  const int m = mine ? table_find(Pkey, Ppos, capP - 1, tag, i2) : 0;
                  const bool fresh = mine && m < 0;
                  const unsigned long long ball = __ballot(fresh);
                  const int p = nO + lanes_below(ball, lane);
                  if (nO + __popcll(ball) > capO) { bad = true; break; }
                  if (fresh)
                  {
                     table_insert(Pkey, Ppos, capP - 1, tag, i2, p);
                     oj[p] = i2;
                     if (FILL) { oa[p] = v; }
                  }
                  else if (FILL && mine) { oa[m] += v; }
                  nO += __popcll(ball);
                  __syncthreads();
               }
            }
            __syncthreads();
         }
      }
      if (bad) { if (lane == 0) { atomicExch(overflow, 1); } nO = 0; }
      if (maxima) { if (lane == 0) { atomicMax(&maxima[0], nRA); atomicMax(&maxima[1], nO); } }
      else if (!FILL) { if (lane == 0) { rowlen[ic] = nO; } }
      else
      {
         // exact offsets (second pass of two) or a fixed stride per row (single pass: lengths come out as well)
         const size_t c0 = Ci ? (size_t) Ci[ic] : (size_t) ic * (size_t) capO;
         if (!Ci && lane == 0) { rowlen[ic] = nO; }
         for (int k = lane; k < nO; k += 64) { Cj[c0 + k] = oj[k]; Ca[c0 + k] = oa[k]; }
      }
      __syncthreads();
   }
}

// ---------------------------------------------------------------------------------------------------------------------
// The product of a DISTRIBUTED level (par_rap.c:30-2000 with ghost columns; the host restatement is
// par_amg_setup_dist.cpp:dist_build_coarse_operator).  Same walk, on the extended numbering (dist_setup_kernels.hip):
//   * a fine row of A is its ghost block (columns + a2_off) FOLLOWED BY its diagonal block — the host visits A_offd
//     before A_diag, and keeps two RA lists, the ghost one applied first: here one list, applied in two passes by class;
//   * P has the neighbours' rows (P_ext) appended as rows nfine .. and ghost coarse columns numbered from `split`;
//   * a row starts (behind its diagonal slot) with what the neighbours computed for it: rows of X named by F;
//   * the finished row is written with its diagonal-block columns (< split) first, the others behind, both in
//     first-touch order — the host appends to two lists;
//   * DIRECT: rows made FOR the neighbours (RAP_int) are triple loops without the intermediate RA (r*a is not summed over
//     the rows of R before it meets P): every (r, a) pair is its own RA entry.
// ---------------------------------------------------------------------------------------------------------------------
struct RapDist
{
   const int *A2i, *A2j; const double *A2a; int a2_off;
   int nfine;
   const int *Fi, *Fj;
   const int *Xi, *Xj; const double *Xa;
   int split;
   int *rowlen_d;
};

// rows M(srcrow[q], :) scaled by srcmul[q] (nullptr: taken as they are), q = 0 .. count-1, into the output list in
// first-touch order.  cls >= 0: only the sources whose number is >= thr (cls 1) or < thr (cls 0).  false: table overflow.
template <bool FILL>
__device__ __forceinline__ bool rap_apply_rows(const int *srcrow, const double *srcmul, int count, int cls, int thr,
                                               const int *__restrict__ Mi, const int *__restrict__ Mj, const double *__restrict__ Ma,
                                               unsigned long long *Pkey, int *Ppos, int capP, unsigned tag,
                                               int *oj, double *oa, int capO, int &nO,
                                               int *sincl, int *sbeg, double *sval, int lane)
{
   for (int qb = 0; qb < count; qb += 64)
   {
      const int q = qb + lane;
      bool src_here = q < count;
      const int i1 = src_here ? srcrow[q] : 0;
      if (cls >= 0 && src_here && ((i1 >= thr) != (cls == 1))) { src_here = false; }
      const int p0 = src_here ? Mi[i1] : 0;
      const int len = src_here ? Mi[i1 + 1] - p0 : 0;
      const int incl = wave_scan_incl(len, lane);
      sincl[lane] = incl; sbeg[lane] = p0;
      if (FILL) { sval[lane] = src_here ? (srcmul ? srcmul[q] : 1.0) : 0.0; }
      __syncthreads();
      const int total = sincl[63];
      for (int cb = 0; cb < total; cb += 64)
      {
         const int t = cb + lane;
         const bool have = t < total;
         const int src = have ? owner_of(sincl, t) : 64;
         const int j2 = have ? sbeg[src] + (t - (src ? sincl[src - 1] : 0)) : 0;
         const int i2 = have ? Mj[j2] : -1;
         const double v = (FILL && have) ? (srcmul ? sval[src] * Ma[j2] : Ma[j2]) : 0.0;
         const int s_lo = __builtin_amdgcn_readfirstlane(src);
         const int s_hi = __builtin_amdgcn_readfirstlane(owner_of(sincl, min(cb + 63, total - 1)));
         for (int sr = s_lo; sr <= s_hi; sr++)
         {
            const bool mine = have && src == sr;
            if (__ballot(mine) == 0ull) { continue; }
            const int m = mine ? table_find(Pkey, Ppos, capP - 1, tag, i2) : 0;
            const bool fresh = mine && m < 0;
            const unsigned long long ball = __ballot(fresh);
            const int p = nO + lanes_below(ball, lane);
            if (nO + __popcll(ball) > capO) { return false; }
            if (fresh)
            {
               table_insert(Pkey, Ppos, capP - 1, tag, i2, p);
               oj[p] = i2;
               if (FILL) { oa[p] = v; }
            }
            else if (FILL && mine) { oa[m] += v; }
            nO += __popcll(ball);
            __syncthreads();
         }
      }
      __syncthreads();
   }
   return true;
}

template <bool FILL, bool DIRECT>
__global__ __launch_bounds__(64)
void rap_rows_dist_kernel(int nc, int square,
                          const int *__restrict__ Ri, const int *__restrict__ Rj, const double *__restrict__ Ra,
                          const int *__restrict__ Ai, const int *__restrict__ Aj, const double *__restrict__ Aa,
                          const int *__restrict__ Pi, const int *__restrict__ Pj, const double *__restrict__ Pa,
                          RapDist dd, int capA, int capRA, int capP, int capO,
                          int *__restrict__ rowlen, int *__restrict__ Cj, double *__restrict__ Ca,
                          int *overflow, int row_step, int *__restrict__ maxima)
{
   extern __shared__ __align__(16) unsigned char smem[];
   unsigned long long *Akey = reinterpret_cast<unsigned long long *>(smem);
   unsigned long long *Pkey = Akey + capA;
   double *raa = reinterpret_cast<double *>(Pkey + capP);
   double *oa  = raa + capRA;
   double *sval = oa + capO;
   int *Apos = reinterpret_cast<int *>(sval + 64);
   int *Ppos = Apos + capA;
   int *raj  = Ppos + capP;
   int *oj   = raj + capRA;
   int *sincl = oj + capO;
   int *sbeg  = sincl + 64;
   const int lane = threadIdx.x;
   for (int i = lane; i < capA; i += 64) { Akey[i] = 0; }
   for (int i = lane; i < capP; i += 64) { Pkey[i] = 0; }
   __syncthreads();

   unsigned tag = 0;
   for (int ic = blockIdx.x * row_step; ic < nc; ic += gridDim.x * row_step)
   {
      if (__hip_atomic_load(overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { break; }
      tag++;
      // ---- RA: the rows of R(ic,:), 32 at a time, each as two pieces (ghost block of A, then diagonal block)
      int nRA = 0;
      bool bad = false, has_ghost = false;
      const int r0 = Ri[ic], r1 = Ri[ic + 1];
      for (int rb = r0; rb < r1 && !bad; rb += 32)
      {
         const int j1 = rb + (lane >> 1);
         const int seg = lane & 1;                     // 0: ghost block, 1: diagonal block
         const bool src_here = j1 < r1 && (seg == 1 || dd.A2i != nullptr);
         const int i1 = src_here ? Rj[j1] : 0;
         const int a0 = src_here ? (seg ? Ai[i1] : dd.A2i[i1]) : 0;
         const int len = src_here ? (seg ? Ai[i1 + 1] : dd.A2i[i1 + 1]) - a0 : 0;
         const int incl = wave_scan_incl(len, lane);
         sincl[lane] = incl; sbeg[lane] = a0;
         if (FILL) { sval[lane] = src_here ? Ra[j1] : 0.0; }
         __syncthreads();
         const int total = sincl[63];
         for (int cb = 0; cb < total && !bad; cb += 64)
         {
            const int t = cb + lane;
            const bool have = t < total;
            const int src = have ? owner_of(sincl, t) : 64;
            const int j2 = have ? sbeg[src] + (t - (src ? sincl[src - 1] : 0)) : 0;
            const bool dblock = (src & 1) != 0;
            const int i2 = have ? (dblock ? Aj[j2] : dd.A2j[j2] + dd.a2_off) : -1;
            const double v = (FILL && have) ? sval[src] * (dblock ? Aa[j2] : dd.A2a[j2]) : 0.0;
            if (DIRECT)
            {
               const int cnt = min(64, total - cb);
               if (nRA + cnt > capRA) { bad = true; break; }
               if (have) { raj[nRA + lane] = i2; if (FILL) { raa[nRA + lane] = v; } }
               nRA += cnt;
               __syncthreads();
               continue;
            }
            const int s_lo = __builtin_amdgcn_readfirstlane(src);
            const int s_hi = __builtin_amdgcn_readfirstlane(owner_of(sincl, min(cb + 63, total - 1)));
            for (int sr = s_lo; sr <= s_hi; sr++)
            {
               const bool mine = have && src == sr;
               if (__ballot(mine) == 0ull) { continue; }
               const int m = mine ? table_find(Akey, Apos, capA - 1, tag, i2) : 0;
               const bool fresh = mine && m < 0;
               const unsigned long long ball = __ballot(fresh);
               const int p = nRA + lanes_below(ball, lane);
               if (nRA + __popcll(ball) > capRA) { bad = true; break; }
               if (fresh)
               {
                  table_insert(Akey, Apos, capA - 1, tag, i2, p);
                  raj[p] = i2;
                  if (FILL) { raa[p] = v; }
               }
               else if (FILL && mine) { raa[m] += v; }
               if (__ballot(fresh && i2 >= dd.nfine) != 0ull) { has_ghost = true; }
               nRA += __popcll(ball);
               __syncthreads();
            }
         }
         __syncthreads();
      }
      // ---- the row: diagonal slot, the neighbours' contributions, RA_ghost * P_ext, RA_local * P
      int nO = 0;
      if (!bad)
      {
         if (square)
         {
            if (lane == 0) { table_insert(Pkey, Ppos, capP - 1, tag, ic, 0); oj[0] = ic; if (FILL) { oa[0] = 0.0; } }
            nO = 1;
            __syncthreads();
         }
         if (dd.Fi)
         {
            const int f0 = dd.Fi[ic];
            bad = !rap_apply_rows<FILL>(dd.Fj + f0, nullptr, dd.Fi[ic + 1] - f0, -1, 0, dd.Xi, dd.Xj, dd.Xa, Pkey, Ppos, capP, tag, oj, oa, capO, nO, sincl, sbeg, sval, lane);
         }
         if (!bad)
         {
            if (DIRECT || !has_ghost)
            {
               bad = !rap_apply_rows<FILL>(raj, raa, nRA, -1, 0, Pi, Pj, Pa, Pkey, Ppos, capP, tag, oj, oa, capO, nO, sincl, sbeg, sval, lane);
            }
            else
            {
               bad = !rap_apply_rows<FILL>(raj, raa, nRA, 1, dd.nfine, Pi, Pj, Pa, Pkey, Ppos, capP, tag, oj, oa, capO, nO, sincl, sbeg, sval, lane);
               if (!bad) { bad = !rap_apply_rows<FILL>(raj, raa, nRA, 0, dd.nfine, Pi, Pj, Pa, Pkey, Ppos, capP, tag, oj, oa, capO, nO, sincl, sbeg, sval, lane); }
            }
         }
      }
      if (bad) { if (lane == 0) { atomicExch(overflow, 1); } nO = 0; }
      if (maxima) { if (lane == 0) { atomicMax(&maxima[0], nRA); atomicMax(&maxima[1], nO); } }
      else if (!FILL) { if (lane == 0) { rowlen[ic] = nO; } }
      else
      {
         const size_t c0 = (size_t) ic * (size_t) capO;
         if (DIRECT)
         {
            if (lane == 0) { rowlen[ic] = nO; }
            for (int k = lane; k < nO; k += 64) { Cj[c0 + k] = oj[k]; Ca[c0 + k] = oa[k]; }
         }
         else
         {
            // diagonal-block columns first, the others behind, each class in the order it was met
            int nd = 0;
            for (int kb = 0; kb < nO; kb += 64) { const int k = kb + lane; nd += __popcll(__ballot(k < nO && oj[k] < dd.split)); }
            int wd = 0, wo = nd;
            for (int kb = 0; kb < nO; kb += 64)
            {
               const int k = kb + lane;
               const bool have = k < nO;
               const int c = have ? oj[k] : 0;
               const bool isd = have && c < dd.split, iso = have && !isd;
               const unsigned long long bd = __ballot(isd), bo = __ballot(iso);
               if (isd) { const int p = wd + lanes_below(bd, lane); Cj[c0 + p] = c; Ca[c0 + p] = oa[k]; }
               if (iso) { const int p = wo + lanes_below(bo, lane); Cj[c0 + p] = c; Ca[c0 + p] = oa[k]; }
               wd += __popcll(bd); wo += __popcll(bo);
            }
            if (lane == 0) { rowlen[ic] = nO; dd.rowlen_d[ic] = nd; }
         }
      }
      __syncthreads();
   }
}

// rows written with a fixed stride -> CSR
__global__ void rap_compact_kernel(int n, int stride, const int *__restrict__ Ci, const int *__restrict__ sj, const double *__restrict__ sa,
                                   int *__restrict__ Cj, double *__restrict__ Ca)
{
   const size_t t = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
   const int i = (int) (t / (size_t) stride), k = (int) (t % (size_t) stride);
   if (i < n && k < Ci[i + 1] - Ci[i]) { Cj[Ci[i] + k] = sj[t]; Ca[Ci[i] + k] = sa[t]; }
}

// per coarse row: sum of the lengths of the rows of A it touches (bounds the length of RA), maximum over all rows
__global__ void rap_bound_kernel(int nc, const int *__restrict__ Ri, const int *__restrict__ Rj, const int *__restrict__ Ai,
                                 int *__restrict__ max_out)
{
   int m = 0;
   for (int ic = blockIdx.x * blockDim.x + threadIdx.x; ic < nc; ic += gridDim.x * blockDim.x)
   {
      int ub = 0;
      for (int j1 = Ri[ic]; j1 < Ri[ic + 1]; j1++) { const int i1 = Rj[j1]; ub += Ai[i1 + 1] - Ai[i1]; }
      m = max(m, ub);
   }
   for (int off = 32; off > 0; off >>= 1) { m = max(m, __shfl_xor(m, off, 64)); }
   if ((threadIdx.x & 63) == 0) { atomicMax(max_out, m); }
}

static int pow2_at_least(int v) { int p = 8; while (p < v) { p <<= 1; } return p; }

// R (nc x nf), A (nf x nf), P (nf x ncP) as device CSR.  Allocates *Ci / *Cj / *Ca (device) for the product.  Returns false
// when a row does not fit the LDS budget of a workgroup (the caller then forms this product on the host).
bool device_rap(int nc, int ncP, int maxP,
                const int *Ri, const int *Rj, const double *Ra, const int *Ai, const int *Aj, const double *Aa,
                const int *Pi, const int *Pj, const double *Pa, int **Ci_out, int **Cj_out, double **Ca_out, int *nnz_out,
                hipStream_t s)
{
   const bool square = (nc == ncP);
   const bool timing = getenv("HYPRE_AMD_SETUP_TIMING") != nullptr;
   const double t_begin = omp_get_wtime();
   double t_alloc = 0.0, t_walk = 0.0;
   int *d_scr = nullptr;
   HIP_CHECK(hipMalloc((void **) &d_scr, sizeof(int) * 2));
   HIP_CHECK(hipMemsetAsync(d_scr, 0, sizeof(int) * 2, s));
   int grid = (nc + 255) / 256;
   if (grid > 4096) { grid = 4096; }
   hipLaunchKernelGGL(rap_bound_kernel, dim3(grid), dim3(256), 0, s, nc, Ri, Rj, Ai, d_scr);
   int h_scr[2] = {0, 0};
   HIP_CHECK(hipMemcpyAsync(h_scr, d_scr, sizeof(int) * 2, hipMemcpyDeviceToHost, s));
   HIP_CHECK(hipStreamSynchronize(s));
   const int ubA = std::max(h_scr[0], 1);
   auto lds_bytes = [](int capA, int capRA, int capP, int capO)
   { return (size_t) 8 * capA + 8 * capP + 8 * capRA + 8 * capO + 8 * 64 + 4 * capA + 4 * capP + 4 * capRA + 4 * capO + 4 * 128 + 64; };
   // Pass 1 (lengths).  The tables are sized by what rows of such products usually need, not by the worst case the
   // bounds allow (RA <= sum of the touched rows of A; the row <= RA x longest row of P): a first attempt with small
   // tables, the next with four times the room if any row overflowed, the host after the last.
   const long long ubO = std::min<long long>((long long) ubA * std::max(maxP, 1) + 1, (long long) ncP);
   const size_t budget = 150 * 1024;
   (void) hipFuncSetAttribute((const void *) rap_rows_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) budget);
   (void) hipFuncSetAttribute((const void *) rap_rows_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) budget);
   int *rowlen = nullptr;
   HIP_CHECK(hipMalloc((void **) &rowlen, sizeof(int) * ((size_t) nc + 1)));
   const int waves = std::min(nc, handle().num_cus * 32);
   int capRA = 0, capA = 0, capO = 0, capP = 0;
   bool done = false, single = false;
   int *sj = nullptr;
   double *sa = nullptr;
   // The rows are walked entry by entry: the kernel lives on the number of rows in flight, i.e. on SMALL tables, and the
   // bounds (RA <= sum of the touched rows of A; the row <= RA x longest row of P) are several times what rows need.
   // So a pilot walks every 61st row with the largest tables that fit and reports the longest RA and the longest row it
   // met; the tables of the real walk take that plus a quarter.  A row that overflows them sends everyone to tables
   // half as large again (the waves leave the lost attempt at their next row), the host loop after the last.
   // The walk is the cost, so it is done ONCE where memory allows: rows go to a scratch array with room for capO entries
   // each, their lengths come out of the same pass, a copy packs them.  (With less room: lengths first, then a second
   // walk that writes at the exact offsets.)
   auto table_for = [](int entries) { return pow2_at_least((entries * 29 + 19) / 20); };      // load factor <= 0.69
   int needRA = std::min(ubA, 384), needO = (int) std::min<long long>(ubO, 192);      // without a pilot (few rows)
   {
      int pRA = std::min(ubA, 1536), pO = (int) std::min<long long>(ubO, 768);
      pRA = (pRA + 1) & ~1; pO = (pO + 1) & ~1;
      const int pA = table_for(pRA), pP = table_for(pO);
      if (lds_bytes(pA, pRA, pP, pO) <= budget && nc >= 4096)
      {
         int *d_max = nullptr;
         HIP_CHECK(hipMalloc((void **) &d_max, sizeof(int) * 2));
         HIP_CHECK(hipMemsetAsync(d_max, 0, sizeof(int) * 2, s));
         HIP_CHECK(hipMemsetAsync(d_scr + 1, 0, sizeof(int), s));
         const int step = 61, rows = (nc + step - 1) / step;
         hipLaunchKernelGGL((rap_rows_kernel<false>), dim3(std::min(rows, handle().num_cus * 8)), dim3(64), lds_bytes(pA, pRA, pP, pO), s,
                            nc, square ? 1 : 0, Ri, Rj, Ra, Ai, Aj, Aa, Pi, Pj, Pa, pA, pRA, pP, pO, (int *) nullptr, (const int *) nullptr,
                            (int *) nullptr, (double *) nullptr, d_scr + 1, step, d_max);
         int h_max[2] = {0, 0};
         HIP_CHECK(hipMemcpyAsync(h_max, d_max, sizeof(int) * 2, hipMemcpyDeviceToHost, s));
         HIP_CHECK(hipMemcpyAsync(h_scr, d_scr, sizeof(int) * 2, hipMemcpyDeviceToHost, s));
         HIP_CHECK(hipStreamSynchronize(s));
         HIP_CHECK(hipFree(d_max));
         if (h_scr[1] == 0)
         {
            needRA = std::min(ubA, h_max[0] + h_max[0] / 4 + 16);
            needO = (int) std::min<long long>(ubO, (long long) h_max[1] + h_max[1] / 4 + 8);
         }
      }
   }
   // the strided scratch of the one-walk form: at most 24 GB and at most half of what the device has free right now
   // (other hierarchies, other tenants); an allocation that fails all the same sends this product to the two-walk form,
   // which needs none
   size_t free_b = 0, total_b = 0;
   if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void) hipGetLastError(); free_b = (size_t) 48 << 30; }
   const size_t scratch_limit = std::min<size_t>((size_t) 24 << 30, free_b / 2);
   for (int attempt = 0; attempt < 6 && !done; attempt++)
   {
      capRA = (std::min(ubA, needRA) + 1) & ~1;
      capA = table_for(capRA);
      capO = (int) ((std::min<long long>(ubO, needO) + 1) & ~1LL);
      capP = table_for(capO);
      if (lds_bytes(capA, capRA, capP, capO) > budget) { break; }
      HIP_CHECK(hipMemsetAsync(d_scr + 1, 0, sizeof(int), s));
      const size_t slots = (size_t) nc * (size_t) capO;
      single = slots * 12 <= scratch_limit;
      if (single)
      {
         const double ta = omp_get_wtime();
         if (hipMalloc((void **) &sj, sizeof(int) * slots) != hipSuccess) { (void) hipGetLastError(); sj = nullptr; single = false; }
         if (single && hipMalloc((void **) &sa, sizeof(double) * slots) != hipSuccess)
         {
            (void) hipGetLastError();
            HIP_CHECK(hipFree(sj)); sj = nullptr; sa = nullptr; single = false;
         }
         t_alloc += omp_get_wtime() - ta;
      }
      if (single)
      {
         hipLaunchKernelGGL((rap_rows_kernel<true>), dim3(waves), dim3(64), lds_bytes(capA, capRA, capP, capO), s, nc, square ? 1 : 0,
                            Ri, Rj, Ra, Ai, Aj, Aa, Pi, Pj, Pa, capA, capRA, capP, capO, rowlen, (const int *) nullptr, sj, sa, d_scr + 1,
                            1, (int *) nullptr);
      }
      else
      {
         hipLaunchKernelGGL((rap_rows_kernel<false>), dim3(waves), dim3(64), lds_bytes(capA, capRA, capP, capO), s, nc, square ? 1 : 0,
                            Ri, Rj, Ra, Ai, Aj, Aa, Pi, Pj, Pa, capA, capRA, capP, capO, rowlen, (const int *) nullptr, (int *) nullptr,
                            (double *) nullptr, d_scr + 1, 1, (int *) nullptr);
      }
      const double tw = omp_get_wtime();
      HIP_CHECK(hipMemcpyAsync(h_scr, d_scr, sizeof(int) * 2, hipMemcpyDeviceToHost, s));
      HIP_CHECK(hipStreamSynchronize(s));
      t_walk += omp_get_wtime() - tw;
      done = h_scr[1] == 0;
      if (!done && single) { HIP_CHECK(hipFree(sj)); HIP_CHECK(hipFree(sa)); sj = nullptr; sa = nullptr; }
      if (capRA >= ubA && capO >= ubO) { break; }        // the bounds themselves fitted: nothing larger to try
      needRA = std::min(ubA, needRA + needRA / 2 + 16);
      needO = (int) std::min<long long>(ubO, (long long) needO + needO / 2 + 8);
   }
   auto give_up = [&]() { HIP_CHECK(hipFree(rowlen)); HIP_CHECK(hipFree(d_scr)); if (sj) { HIP_CHECK(hipFree(sj)); } if (sa) { HIP_CHECK(hipFree(sa)); } return false; };
   if (!done) { return give_up(); }
   // row pointers: exclusive scan of the lengths, in place
   launch_scan_exclusive(rowlen, nc, s);
   int nnz = 0;
   HIP_CHECK(hipMemcpyAsync(&nnz, rowlen + nc, sizeof(int), hipMemcpyDeviceToHost, s));
   HIP_CHECK(hipStreamSynchronize(s));
   if (nnz < 0) { return give_up(); }               // more than 2^31 - 1 entries
   int *Ci = rowlen, *Cj = nullptr;
   double *Ca = nullptr;
   if (hipMalloc((void **) &Cj, sizeof(int) * (size_t) std::max(nnz, 1)) != hipSuccess ||
       hipMalloc((void **) &Ca, sizeof(double) * (size_t) std::max(nnz, 1)) != hipSuccess)
   {
      // no room for the product on the device: the host loop takes over
      (void) hipGetLastError();
      if (Cj) { HIP_CHECK(hipFree(Cj)); }
      return give_up();
   }
   if (single)
   {
      const size_t slots = (size_t) nc * (size_t) capO;
      hipLaunchKernelGGL(rap_compact_kernel, dim3((unsigned) ((slots + 255) / 256)), dim3(256), 0, s, nc, capO, Ci, sj, sa, Cj, Ca);
      HIP_CHECK(hipStreamSynchronize(s));
      HIP_CHECK(hipFree(sj)); HIP_CHECK(hipFree(sa));
      HIP_CHECK(hipFree(d_scr));
   }
   else
   {
      // pass 2 (columns and values): the output tables need only hold the longest row found
      const int maxlen = device_max_row_nnz(Ci, nc, s);
      capO = (std::max(maxlen, 1) + 1) & ~1;
      capP = table_for(capO);
      HIP_CHECK(hipMemsetAsync(d_scr + 1, 0, sizeof(int), s));
      hipLaunchKernelGGL((rap_rows_kernel<true>), dim3(waves), dim3(64), lds_bytes(capA, capRA, capP, capO), s, nc, square ? 1 : 0,
                         Ri, Rj, Ra, Ai, Aj, Aa, Pi, Pj, Pa, capA, capRA, capP, capO, (int *) nullptr, Ci, Cj, Ca, d_scr + 1, 1, (int *) nullptr);
      HIP_CHECK(hipMemcpyAsync(h_scr, d_scr, sizeof(int) * 2, hipMemcpyDeviceToHost, s));
      HIP_CHECK(hipStreamSynchronize(s));
      HIP_CHECK(hipFree(d_scr));
      if (h_scr[1]) { HIP_CHECK(hipFree(Ci)); HIP_CHECK(hipFree(Cj)); HIP_CHECK(hipFree(Ca)); return false; }
   }
   if (timing)
   {
      fprintf(stderr, "   product: %d rows, RA <= %d, tables %d / %d, %s: scratch allocation %.3fs, walk %.3fs, all %.3fs\n", nc, ubA, capRA, capO,
              single ? "one walk" : "two walks", t_alloc, t_walk, omp_get_wtime() - t_begin);
   }
   *Ci_out = Ci; *Cj_out = Cj; *Ca_out = Ca; *nnz_out = nnz;
   return true;
}

// bound of RA for rows whose fine rows have a ghost block too
__global__ void rap_bound2_kernel(int nc, const int *__restrict__ Ri, const int *__restrict__ Rj, const int *__restrict__ Ai,
                                  const int *__restrict__ A2i, int *__restrict__ max_out)
{
   int m = 0;
   for (int ic = blockIdx.x * blockDim.x + threadIdx.x; ic < nc; ic += gridDim.x * blockDim.x)
   {
      int ub = 0;
      for (int j1 = Ri[ic]; j1 < Ri[ic + 1]; j1++)
      {
         const int i1 = Rj[j1];
         ub += Ai[i1 + 1] - Ai[i1];
         if (A2i) { ub += A2i[i1 + 1] - A2i[i1]; }
      }
      m = max(m, ub);
   }
   for (int off = 32; off > 0; off >>= 1) { m = max(m, __shfl_xor(m, off, 64)); }
   if ((threadIdx.x & 63) == 0) { atomicMax(max_out, m); }
}

void device_split_strided(int n, int stride, int *len, int *nd, const int *sj, const double *sa, int split,
                          int **Di_out, int **Dj_out, double **Da_out, int *dnnz, int **Oi_out, int **Oj_out, double **Oa_out, int *onnz,
                          hipStream_t s);

// The product of a distributed level (see rap_rows_dist_kernel).  R: nc rows over the local fine points; A / A2: the
// diagonal and the ghost block of A (A2 may be null); P: the extended interpolation operator (local rows, then P_ext).
// direct: rows for the neighbours — one CSR over the extended coarse numbering comes back in (Di, Dj, Da); otherwise the
// two blocks of the coarse operator, split at `split`.  ncols_out bounds a row's length, max_seed the entries a row
// receives from X.  false: a row does not fit the tables (the caller forms the product on the host).
bool device_rap_dist(bool direct, int nc, int square, int ncols_out, int maxP, int max_seed,
                     const int *Ri, const int *Rj, const double *Ra, const int *Ai, const int *Aj, const double *Aa,
                     const int *A2i, const int *A2j, const double *A2a, int a2_off, int nfine,
                     const int *Pi, const int *Pj, const double *Pa,
                     const int *Fi, const int *Fj, const int *Xi, const int *Xj, const double *Xa, int split,
                     int **Di_out, int **Dj_out, double **Da_out, int *dnnz, int **Oi_out, int **Oj_out, double **Oa_out, int *onnz,
                     hipStream_t s)
{
   *dnnz = 0;
   if (onnz) { *onnz = 0; }
   if (nc <= 0)
   {
      // a rank without rows on this level: empty blocks
      int *Di = nullptr, *Oi = nullptr;
      HIP_CHECK(hipMalloc((void **) &Di, sizeof(int) * 2)); HIP_CHECK(hipMemsetAsync(Di, 0, sizeof(int) * 2, s));
      *Di_out = Di; *Dj_out = nullptr; *Da_out = nullptr;
      if (!direct) { HIP_CHECK(hipMalloc((void **) &Oi, sizeof(int) * 2)); HIP_CHECK(hipMemsetAsync(Oi, 0, sizeof(int) * 2, s)); *Oi_out = Oi; *Oj_out = nullptr; *Oa_out = nullptr; }
      HIP_CHECK(hipStreamSynchronize(s));
      return true;
   }
   RapDist dd;
   dd.A2i = A2i; dd.A2j = A2j; dd.A2a = A2a; dd.a2_off = a2_off; dd.nfine = nfine;
   dd.Fi = Fi; dd.Fj = Fj; dd.Xi = Xi; dd.Xj = Xj; dd.Xa = Xa; dd.split = split; dd.rowlen_d = nullptr;
   int *d_scr = nullptr;
   HIP_CHECK(hipMalloc((void **) &d_scr, sizeof(int) * 4));
   HIP_CHECK(hipMemsetAsync(d_scr, 0, sizeof(int) * 4, s));
   int grid = std::min((nc + 255) / 256, 4096);
   hipLaunchKernelGGL(rap_bound2_kernel, dim3(grid), dim3(256), 0, s, nc, Ri, Rj, Ai, A2i, d_scr);
   int h_scr[4] = {0, 0, 0, 0};
   HIP_CHECK(hipMemcpyAsync(h_scr, d_scr, sizeof(int) * 4, hipMemcpyDeviceToHost, s));
   HIP_CHECK(hipStreamSynchronize(s));
   const int ubA = std::max(h_scr[0], 1);
   auto lds_bytes = [](int capA, int capRA, int capP, int capO)
   { return (size_t) 8 * capA + 8 * capP + 8 * capRA + 8 * capO + 8 * 64 + 4 * capA + 4 * capP + 4 * capRA + 4 * capO + 4 * 128 + 64; };
   const long long ubO = std::max<long long>(1, std::min<long long>((long long) ubA * std::max(maxP, 1) + 1 + max_seed, (long long) ncols_out));
   const size_t budget = 150 * 1024;
   (void) hipFuncSetAttribute((const void *) rap_rows_dist_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) budget);
   (void) hipFuncSetAttribute((const void *) rap_rows_dist_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) budget);
   (void) hipFuncSetAttribute((const void *) rap_rows_dist_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) budget);
   (void) hipFuncSetAttribute((const void *) rap_rows_dist_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) budget);
   int *rowlen = nullptr, *rowlen_d = nullptr;
   HIP_CHECK(hipMalloc((void **) &rowlen, sizeof(int) * ((size_t) nc + 1)));
   HIP_CHECK(hipMalloc((void **) &rowlen_d, sizeof(int) * ((size_t) nc + 1)));
   dd.rowlen_d = rowlen_d;
   const int waves = std::min(nc, handle().num_cus * 32);
   auto table_for = [](int entries) { return pow2_at_least((entries * 29 + 19) / 20); };
   int needRA = std::min(ubA, 384), needO = (int) std::min<long long>(ubO, 192);
   {
      int pRA = std::min(ubA, 1536), pO = (int) std::min<long long>(ubO, 768);
      pRA = (pRA + 1) & ~1; pO = (pO + 1) & ~1;
      const int pA = direct ? 8 : table_for(pRA), pP = table_for(pO);
      if (lds_bytes(pA, pRA, pP, pO) <= budget && nc >= 4096)
      {
         const int step = 61, rows = (nc + step - 1) / step;
         const dim3 g(std::min(rows, handle().num_cus * 8));
         if (direct)
         {
            hipLaunchKernelGGL((rap_rows_dist_kernel<false, true>), g, dim3(64), lds_bytes(pA, pRA, pP, pO), s, nc, square, Ri, Rj, Ra, Ai, Aj, Aa,
                               Pi, Pj, Pa, dd, pA, pRA, pP, pO, (int *) nullptr, (int *) nullptr, (double *) nullptr, d_scr + 1, step, d_scr + 2);
         }
         else
         {
            hipLaunchKernelGGL((rap_rows_dist_kernel<false, false>), g, dim3(64), lds_bytes(pA, pRA, pP, pO), s, nc, square, Ri, Rj, Ra, Ai, Aj, Aa,
                               Pi, Pj, Pa, dd, pA, pRA, pP, pO, (int *) nullptr, (int *) nullptr, (double *) nullptr, d_scr + 1, step, d_scr + 2);
         }
         HIP_CHECK(hipMemcpyAsync(h_scr, d_scr, sizeof(int) * 4, hipMemcpyDeviceToHost, s));
         HIP_CHECK(hipStreamSynchronize(s));
         if (h_scr[1] == 0)
         {
            needRA = std::min(ubA, h_scr[2] + h_scr[2] / 4 + 16);
            needO = (int) std::min<long long>(ubO, (long long) h_scr[3] + h_scr[3] / 4 + 8);
         }
      }
   }
   int capRA = 0, capA = 0, capO = 0, capP = 0;
   bool done = false;
   int *sj = nullptr;
   double *sa = nullptr;
   size_t free_b = 0, total_b = 0;
   (void) hipMemGetInfo(&free_b, &total_b);
   const size_t scratch_limit = std::min<size_t>((size_t) 24 << 30, free_b / 2);
   for (int attempt = 0; attempt < 8 && !done; attempt++)
   {
      capRA = (std::min(ubA, needRA) + 1) & ~1;
      capA = direct ? 8 : table_for(capRA);
      capO = (int) ((std::min<long long>(ubO, needO) + 1) & ~1LL);
      capP = table_for(capO);
      if (lds_bytes(capA, capRA, capP, capO) > budget) { break; }
      const size_t slots = (size_t) nc * (size_t) capO;
      if (slots * 12 > scratch_limit) { break; }
      HIP_CHECK(hipMemsetAsync(d_scr + 1, 0, sizeof(int), s));
      if (hipMalloc((void **) &sj, sizeof(int) * slots) != hipSuccess) { sj = nullptr; (void) hipGetLastError(); break; }
      if (hipMalloc((void **) &sa, sizeof(double) * slots) != hipSuccess) { sa = nullptr; (void) hipGetLastError(); break; }
      if (direct)
      {
         hipLaunchKernelGGL((rap_rows_dist_kernel<true, true>), dim3(waves), dim3(64), lds_bytes(capA, capRA, capP, capO), s, nc, square, Ri, Rj, Ra,
                            Ai, Aj, Aa, Pi, Pj, Pa, dd, capA, capRA, capP, capO, rowlen, sj, sa, d_scr + 1, 1, (int *) nullptr);
      }
      else
      {
         hipLaunchKernelGGL((rap_rows_dist_kernel<true, false>), dim3(waves), dim3(64), lds_bytes(capA, capRA, capP, capO), s, nc, square, Ri, Rj, Ra,
                            Ai, Aj, Aa, Pi, Pj, Pa, dd, capA, capRA, capP, capO, rowlen, sj, sa, d_scr + 1, 1, (int *) nullptr);
      }
      HIP_CHECK(hipMemcpyAsync(h_scr, d_scr, sizeof(int) * 4, hipMemcpyDeviceToHost, s));
      HIP_CHECK(hipStreamSynchronize(s));
      done = h_scr[1] == 0;
      if (!done) { HIP_CHECK(hipFree(sj)); HIP_CHECK(hipFree(sa)); sj = nullptr; sa = nullptr; }
      if (capRA >= ubA && capO >= ubO) { break; }
      needRA = std::min(ubA, needRA + needRA / 2 + 16);
      needO = (int) std::min<long long>(ubO, (long long) needO + needO / 2 + 8);
   }
   HIP_CHECK(hipFree(d_scr));
   if (!done)
   {
      HIP_CHECK(hipFree(rowlen)); HIP_CHECK(hipFree(rowlen_d));
      if (sj) { HIP_CHECK(hipFree(sj)); }
      if (sa) { HIP_CHECK(hipFree(sa)); }
      return false;
   }
   if (direct)
   {
      launch_scan_exclusive(rowlen, nc, s);
      int nnz = 0;
      HIP_CHECK(hipMemcpyAsync(&nnz, rowlen + nc, sizeof(int), hipMemcpyDeviceToHost, s));
      HIP_CHECK(hipStreamSynchronize(s));
      int *Cj = nullptr;
      double *Ca = nullptr;
      HIP_CHECK(hipMalloc((void **) &Cj, sizeof(int) * (size_t) std::max(nnz, 1)));
      HIP_CHECK(hipMalloc((void **) &Ca, sizeof(double) * (size_t) std::max(nnz, 1)));
      const size_t slots = (size_t) nc * (size_t) capO;
      hipLaunchKernelGGL(rap_compact_kernel, dim3((unsigned) ((slots + 255) / 256)), dim3(256), 0, s, nc, capO, rowlen, sj, sa, Cj, Ca);
      HIP_CHECK(hipStreamSynchronize(s));
      HIP_CHECK(hipFree(rowlen_d));
      *Di_out = rowlen; *Dj_out = Cj; *Da_out = Ca; *dnnz = nnz;
   }
   else
   {
      device_split_strided(nc, capO, rowlen, rowlen_d, sj, sa, split, Di_out, Dj_out, Da_out, dnnz, Oi_out, Oj_out, Oa_out, onnz, s);
      HIP_CHECK(hipFree(rowlen)); HIP_CHECK(hipFree(rowlen_d));
   }
   HIP_CHECK(hipFree(sj)); HIP_CHECK(hipFree(sa));
   return true;
}

// The code object of this file is loaded when one of its kernels is first asked for: ensure_device() asks here, so that
// the load (tens of milliseconds per file) is part of bringing the device up, not of the first setup or solve.
void preload_rap_kernels() { hipFuncAttributes at; (void) hipFuncGetAttributes(&at, (const void *) rap_rows_kernel<true>); (void) hipGetLastError(); }

}  // namespace hamd
