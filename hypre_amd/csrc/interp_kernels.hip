// hypre_amd — extended+i interpolation on the device (single rank, scalar problems), bit-identical to the host setup.
//
// Reference: parcsr_ls/par_lr_interp.c:1024-1700 (hypre_BoomerAMGBuildExtPIInterpHost) and the truncation of
// parcsr_mv/par_csr_matrix.c:2874-3400; restated for the host in par_amg_setup.cpp (hypre_BoomerAMGBuildExtPIInterp,
// hypre_BoomerAMGInterpTruncation), whose loop order — the order in which a row's interpolatory set is discovered, the
// order of every sum, the quicksort that picks the entries a truncated row keeps, ties included — this kernel follows
// statement for statement, because the hierarchies the reference's regression files pin depend on those bits.
// The reference's own device routine: parcsr_ls/par_lr_interp_device.c:1001 (a different formulation of the same weights).
//
// One wave owns a row.  What is sequential in the host loop stays sequential (the walk over the row's strong neighbours,
// over the row's entries, every running sum); what the host does for the entries of ONE neighbour's row — look its
// columns up in the row's column -> position map, append the new ones, add the distributed shares — the 64 lanes do side
// by side, since those columns are distinct: appended in entry order (ballot + prefix count), sums over a neighbour's
// row folded in entry order (one lane, values read lane by lane).  The map is an open-addressing table in LDS whose slots
// carry the row's tag.
#include "internal.hpp"
#include <algorithm>

#pragma clang fp contract(off)

namespace hamd {

namespace {

constexpr int ABSENT = -1, STRONG_F = -2;

__device__ __forceinline__ int map_get(const unsigned long long *tab, const int *val, int mask, unsigned tag, int key)
{
   unsigned h = ((unsigned) key * 2654435761u) >> 7;
   while (true)
   {
      const unsigned long long e = tab[h & mask];
      if ((unsigned) (e >> 32) != tag) { return ABSENT; }
      if ((int) (unsigned) e == key) { return val[h & mask]; }
      h++;
   }
}

// insert a key known to be absent, or overwrite the value of a present one (lanes work on distinct keys)
__device__ __forceinline__ void map_set(unsigned long long *tab, int *val, int mask, unsigned tag, int key, int v)
{
   unsigned h = ((unsigned) key * 2654435761u) >> 7;
   const unsigned long long mine = ((unsigned long long) tag << 32) | (unsigned) key;
   while (true)
   {
      const unsigned long long e = tab[h & mask];
      if ((unsigned) (e >> 32) != tag)
      {
         if (atomicCAS(&tab[h & mask], e, mine) == e) { val[h & mask] = v; return; }
         continue;
      }
      if ((int) (unsigned) e == key) { val[h & mask] = v; return; }
      h++;
   }
}

__device__ __forceinline__ int below(unsigned long long ballot, int lane) { return __popcll(ballot & ((1ull << lane) - 1ull)); }

// v of lane l, l the same for the whole wave (v_readlane: no trip through the LDS crossbar)
__device__ __forceinline__ double lane_value(double v, int l)
{
   const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
   return __hiloint2double(hi, lo);
}
// inclusive scan of one int per lane over the wave
__device__ __forceinline__ int wave_scan_incl(int v, int lane)
{
   for (int off = 1; off < 64; off <<= 1) { const int u = __shfl_up(v, off, 64); if (lane >= off) { v += u; } }
   return v;
}
// which of the batch's source rows (inclusive entry counts incl[0..63]) holds flattened entry t: the first s with incl[s] > t
__device__ __forceinline__ int owner_of(const int *incl, int t)
{
   int lo = 0;
#pragma unroll
   for (int step = 32; step > 0; step >>= 1) { if (incl[lo + step - 1] <= t) { lo += step; } }
   return lo;
}

// utilities/qsort.c:395-417 (decreasing |w|; the tie order is part of the contract), recursion unrolled on a stack:
// the left part of a split is sorted before the right one, as the recursive routine does
// Only the first `keep` entries of the sorted row are used afterwards, in the order the sort leaves them: a part that
// lies entirely behind them is not sorted (a partition permutes its own range only, so the front does not notice).
__device__ void qsort2_abs_dev(int *v, double *w, int n, int keep, int *stack)
{
   int top = 0;
   stack[top++] = 0; stack[top++] = n - 1;
   while (top > 0)
   {
      const int right = stack[--top], left = stack[--top];
      if (left >= right || left >= keep) { continue; }
      const int mid = (left + right) / 2;
      { const int tv = v[left]; v[left] = v[mid]; v[mid] = tv; const double tw = w[left]; w[left] = w[mid]; w[mid] = tw; }
      int last = left;
      for (int i = left + 1; i <= right; i++)
      {
         if (fabs(w[i]) > fabs(w[left]))
         {
            ++last;
            const int tv = v[last]; v[last] = v[i]; v[i] = tv; const double tw = w[last]; w[last] = w[i]; w[i] = tw;
         }
      }
      { const int tv = v[left]; v[left] = v[last]; v[last] = tv; const double tw = w[left]; w[left] = w[last]; w[last] = tw; }
      // push the right part first so that the left one is popped (sorted) first
      stack[top++] = last + 1; stack[top++] = right;
      stack[top++] = left;     stack[top++] = last - 1;
   }
}

}  // namespace

// MODE 0: row lengths only (no truncation by count is possible then: max_elmts == 0, tol == 0 handled by the caller);
// MODE 1: write rows at out_off[i] (exact offsets) ; MODE 2: write rows at i * stride and their lengths to rowlen.
template <int MODE>
__global__ __launch_bounds__(64)
void extpi_rows_kernel(int n, const int *__restrict__ Ai, const int *__restrict__ Aj, const double *__restrict__ Aa,
                       const int *__restrict__ Si, const int *__restrict__ Sj, const int *__restrict__ CF,
                       const int *__restrict__ f2c, double trunc_tol, int max_elmts, int capM, int capR,
                       const int *__restrict__ out_off, int stride, int *__restrict__ rowlen,
                       int *__restrict__ Pj, double *__restrict__ Pa, int *__restrict__ overflow,
                       int split, int *__restrict__ rowlen_d)
{
   extern __shared__ __align__(16) unsigned char smem[];
   unsigned long long *Mkey = reinterpret_cast<unsigned long long *>(smem);
   double *pa = reinterpret_cast<double *>(Mkey + capM);
   double *s_a = pa + capR;              // [64] value of every source entry of the batch
   int *Mval = reinterpret_cast<int *>(s_a + 64);
   int *pj = Mval + capM;
   int *stack = pj + capR;               // 4 * capR + 8 ints
   int *sincl = stack + 4 * capR + 8;    // [64] inclusive entry counts of the batch's source rows
   int *sbeg  = sincl + 64;              // [64] where every source row starts in its matrix
   int *s_i1  = sbeg + 64;               // [64] the source entries' columns
   int *s_cf  = s_i1 + 64;               // [64] their C/F markers
   int *s_x   = s_cf + 64;               // [64] their coarse numbers (set discovery) / map values (weights)
   const int lane = threadIdx.x;
   for (int i = lane; i < capM; i += 64) { Mkey[i] = 0; }
   __syncthreads();
   const int mask = capM - 1;
   unsigned tag = 0;

   for (int i = blockIdx.x; i < n; i += gridDim.x)
   {
      // some row did not fit the tables: this attempt is lost, leave it to the next one
      if (__hip_atomic_load(overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { break; }
      tag++;
      int len = 0;
      bool bad = false;
      const int cf_i = CF[i];
      if (cf_i >= 0)
      {
         if (lane == 0) { pj[0] = f2c[i]; pa[0] = 1.0; }
         len = 1;
         __syncthreads();
      }
      else if (cf_i != -3)
      {
         // The host loop walks the strong neighbours one by one and, for an F neighbour, that neighbour's own strong
         // neighbours: a round trip to memory per step.  Here the row's neighbours are taken 64 at a time (columns,
         // markers, coarse numbers, row pointers: two requests), the F neighbours' rows FLATTENED (64 consecutive entries
         // of their concatenation per request, with markers and coarse numbers), and the walk itself — look-ups, appends
         // in entry order — runs over what sits in registers and LDS, in the host's order.
         // ---- the interpolatory set, in first-touch order
         const int s0 = Si[i], s1 = Si[i + 1];
         for (int sb = s0; sb < s1 && !bad; sb += 64)
         {
            const int jj = sb + lane;
            const bool src_here = jj < s1;
            const int i1 = src_here ? Sj[jj] : 0;
            const int cf1 = src_here ? CF[i1] : -3;
            const bool ftype = src_here && cf1 < 0 && cf1 != -3;
            const int b0 = ftype ? Si[i1] : 0;
            const int len1 = ftype ? Si[i1 + 1] - b0 : 0;
            const int fc = src_here ? f2c[i1] : 0;
            const int incl = wave_scan_incl(len1, lane);
            sincl[lane] = incl; sbeg[lane] = b0; s_i1[lane] = i1; s_cf[lane] = cf1; s_x[lane] = fc;
            __syncthreads();
            const int nsrc = min(64, s1 - sb), total = sincl[63];
            int loaded = -1, k1 = 0, fck = 0;
            bool isc = false;
            for (int sl = 0; sl < nsrc && !bad; sl++)
            {
               const int ci1 = s_i1[sl], ccf = s_cf[sl];
               if (ccf >= 0)
               {
                  const int m = map_get(Mkey, Mval, mask, tag, ci1);
                  if (m < 0)
                  {
                     if (len + 1 > capR) { bad = true; break; }
                     if (lane == 0) { map_set(Mkey, Mval, mask, tag, ci1, len); pj[len] = s_x[sl]; pa[len] = 0.0; }
                     len++;
                  }
                  __syncthreads();
               }
               else if (ccf != -3)
               {
                  if (lane == 0) { map_set(Mkey, Mval, mask, tag, ci1, STRONG_F); }
                  __syncthreads();
                  const int e0 = sl ? sincl[sl - 1] : 0, e1 = sincl[sl];
                  for (int c = e0 >> 6; (c << 6) < e1; c++)
                  {
                     const int t = (c << 6) + lane;
                     if (c != loaded)
                     {
                        const bool have = t < total;
                        const int src = have ? owner_of(sincl, t) : 0;
                        const int kk = have ? sbeg[src] + (t - (src ? sincl[src - 1] : 0)) : 0;
                        k1 = have ? Sj[kk] : 0;
                        fck = have ? f2c[k1] : -1;           // the coarse number says it all: -1 for every point that is not C
                        isc = fck >= 0;
                        loaded = c;
                     }
                     const bool mine = t >= e0 && t < e1;
                     const bool fresh = mine && isc && map_get(Mkey, Mval, mask, tag, k1) < 0;
                     const unsigned long long ball = __ballot(fresh);
                     if (len + __popcll(ball) > capR) { bad = true; break; }
                     if (fresh)
                     {
                        const int p = len + below(ball, lane);
                        map_set(Mkey, Mval, mask, tag, k1, p);
                        pj[p] = fck; pa[p] = 0.0;
                     }
                     len += __popcll(ball);
                     __syncthreads();
                  }
               }
            }
            __syncthreads();
         }
         // ---- the weights
         if (!bad && MODE != 0)
         {
            const int a0 = Ai[i], a1 = Ai[i + 1];
            double diagonal = Aa[a0];
            for (int ab = a0 + 1; ab < a1; ab += 64)
            {
               const int jj = ab + lane;
               const bool src_here = jj < a1;
               const int i1 = src_here ? Aj[jj] : 0;
               const double a = src_here ? Aa[jj] : 0.0;
               const int m = src_here ? map_get(Mkey, Mval, mask, tag, i1) : ABSENT;
               const bool sf = src_here && m == STRONG_F;
               // a strong F neighbour's row is taken whole, diagonal entry (its first) included: its sign is needed
               const int b0 = sf ? Ai[i1] : 0;
               const int len1 = sf ? Ai[i1 + 1] - b0 : 0;
               const int cfo = (src_here && m == ABSENT) ? CF[i1] : 0;
               const int incl = wave_scan_incl(len1, lane);
               sincl[lane] = incl; sbeg[lane] = b0; s_x[lane] = m; s_cf[lane] = cfo; s_a[lane] = a;
               __syncthreads();
               const int nsrc = min(64, a1 - ab), total = sincl[63];
               int loaded = -1, i2 = 0;
               double v = 0.0;
               auto load_chunk = [&](int c)
               {
                  const int t = (c << 6) + lane;
                  const bool have = t < total;
                  const int src = have ? owner_of(sincl, t) : 0;
                  const int kk = have ? sbeg[src] + (t - (src ? sincl[src - 1] : 0)) : 0;
                  i2 = have ? Aj[kk] : 0;
                  v = have ? Aa[kk] : 0.0;
                  loaded = c;
               };
               for (int sl = 0; sl < nsrc; sl++)
               {
                  const int cm = s_x[sl];
                  const double ca = s_a[sl];
                  if (cm >= 0)
                  {
                     if (lane == 0) { pa[cm] += ca; }
                     __syncthreads();
                  }
                  else if (cm == STRONG_F)
                  {
                     const int e0 = sl ? sincl[sl - 1] : 0, e1 = sincl[sl];     // entry e0 is the neighbour's diagonal
                     // sum of the couplings of i1 to the set (and to i itself) that have the sign opposite to its diagonal,
                     // added in entry order
                     double sum = 0.0;
                     int sgn = 1;
                     for (int c = e0 >> 6; (c << 6) < e1; c++)
                     {
                        if (c != loaded) { load_chunk(c); }
                        if (c == (e0 >> 6)) { sgn = lane_value(v, e0 & 63) < 0 ? -1 : 1; }
                        const int t = (c << 6) + lane;
                        const bool mine = t > e0 && t < e1;
                        const bool take = mine && (sgn * v) < 0 && (i2 == i || map_get(Mkey, Mval, mask, tag, i2) >= 0);
                        unsigned long long ball = __ballot(take);
                        while (ball)
                        {
                           const int b = __ffsll((long long) ball) - 1;
                           sum += lane_value(v, b);
                           ball &= ball - 1;
                        }
                     }
                     if (sum != 0)
                     {
                        const double distribute = ca / sum;
                        for (int c = e0 >> 6; (c << 6) < e1; c++)
                        {
                           if (c != loaded) { load_chunk(c); }
                           const int t = (c << 6) + lane;
                           const bool mine = t > e0 && t < e1;
                           const bool neg = mine && (sgn * v) < 0;
                           const int m2 = neg ? map_get(Mkey, Mval, mask, tag, i2) : ABSENT;
                           if (m2 >= 0) { pa[m2] += distribute * v; }
                           // at most one entry of the row is i itself
                           const unsigned long long self = __ballot(neg && i2 == i);
                           if (self) { diagonal += distribute * lane_value(v, __ffsll((long long) self) - 1); }
                        }
                        __syncthreads();
                     }
                     else { diagonal += ca; }
                  }
                  else if (s_cf[sl] != -3) { diagonal += ca; }       // weak neighbour: lumped into the diagonal
               }
               __syncthreads();
            }
            if (diagonal != 0.0) { for (int k = lane; k < len; k += 64) { pa[k] /= -diagonal; } }
            __syncthreads();
         }
      }
      if (bad) { if (lane == 0) { atomicExch(overflow, 1); } len = 0; }

      // ---- truncation (par_csr_matrix.c:2874-3400 with rescale, inf-norm): relative threshold, then the max_elmts
      // largest; one lane, the host's statement order.
      // split >= 0 (a distributed level, extended numbering): columns >= split are off-rank coarse points.  The host keeps
      // a row's diagonal-block and off-rank entries in two lists, truncates their concatenation and deals the survivors
      // back to the two lists in the order the sort left them: a stable partition of the row by class before the
      // truncation, and another one after it.
      int nd = len;
      const bool f_row = !bad && cf_i < 0 && cf_i != -3;
      const bool trunc = MODE != 0 && f_row && (trunc_tol > 0.0 || (max_elmts > 0 && max_elmts < len));
      const bool dist_row = MODE != 0 && split >= 0 && f_row && len > 0;
      if (trunc || dist_row)
      {
         if (lane == 0)
         {
            auto partition_row = [&](int cnt)
            {
               int *tj = stack;
               double *ta = reinterpret_cast<double *>(stack + capR);
               int w = 0, ng = 0;
               for (int k = 0; k < cnt; k++)
               {
                  if (pj[k] >= split) { tj[ng] = pj[k]; ta[ng] = pa[k]; ng++; }
                  else { pj[w] = pj[k]; pa[w] = pa[k]; w++; }
               }
               for (int q = 0; q < ng; q++) { pj[w + q] = tj[q]; pa[w + q] = ta[q]; }
               return w;
            };
            int d1 = len, ndl = len;
            if (dist_row) { ndl = partition_row(len); }
            if (trunc)
            {
               if (trunc_tol > 0.0)
               {
                  double row_nrm = 0.0;
                  for (int k = 0; k < len; k++) { row_nrm = fmax(row_nrm, fabs(pa[k])); }
                  const double drop = trunc_tol * row_nrm;
                  double row_sum = 0.0, scale = 0.0;
                  int w = 0;
                  for (int k = 0; k < len; k++)
                  {
                     row_sum += pa[k];
                     if (!(fabs(pa[k]) < drop)) { scale += pa[k]; pa[w] = pa[k]; pj[w] = pj[k]; w++; }
                  }
                  d1 = w;
                  if (scale != 0. && scale != row_sum)
                  {
                     scale = row_sum / scale;
                     for (int k = 0; k < d1; k++) { pa[k] *= scale; }
                  }
               }
               if (max_elmts > 0 && max_elmts < d1)
               {
                  double row_sum = 0.0;
                  for (int k = 0; k < d1; k++) { row_sum += pa[k]; }
                  qsort2_abs_dev(pj, pa, d1, max_elmts, stack);
                  double scale = 0.0;
                  for (int k = 0; k < max_elmts; k++) { scale += pa[k]; }
                  d1 = max_elmts;
                  if (scale != 0. && scale != row_sum)
                  {
                     scale = row_sum / scale;
                     for (int k = 0; k < d1; k++) { pa[k] *= scale; }
                  }
               }
               if (dist_row) { ndl = partition_row(d1); }
            }
            stack[0] = d1; stack[1] = ndl;
         }
         __syncthreads();
         len = stack[0]; nd = stack[1];
         __syncthreads();
      }

      if (MODE == 0) { if (lane == 0) { rowlen[i] = len; } }
      else
      {
         const size_t o = MODE == 1 ? (size_t) out_off[i] : (size_t) i * (size_t) stride;
         if (MODE == 2 && lane == 0) { rowlen[i] = len; if (rowlen_d) { rowlen_d[i] = nd; } }
         for (int k = lane; k < len; k += 64) { Pj[o + k] = pj[k]; Pa[o + k] = pa[k]; }
      }
      __syncthreads();
   }
}

// rows written with a fixed stride -> CSR
__global__ void compact_rows_kernel(int n, int stride, const int *__restrict__ Pi, const int *__restrict__ sj, const double *__restrict__ sa,
                                    int *__restrict__ Pj, double *__restrict__ Pa)
{
   const size_t t = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
   const int i = (int) (t / (size_t) stride), k = (int) (t % (size_t) stride);
   if (i < n && k < Pi[i + 1] - Pi[i]) { Pj[Pi[i] + k] = sj[t]; Pa[Pi[i] + k] = sa[t]; }
}

static int pow2_ge(int v) { int p = 16; while (p < v) { p <<= 1; } return p; }

void device_split_strided(int n, int stride, int *len, int *nd, const int *sj, const double *sa, int split,
                          int **Di_out, int **Dj_out, double **Da_out, int *dnnz, int **Oi_out, int **Oj_out, double **Oa_out, int *onnz,
                          hipStream_t s);

// A (n x n, diagonal first), S (pattern), CF marker and fine -> coarse numbering on the device.  Returns false when a row's
// interpolatory set does not fit LDS (the caller then uses the host loop).  The result arrays are device allocations.
// split >= 0: a distributed level on the extended numbering (dist_setup_kernels.hip): rows 0 .. n-1 are built, the arrays
// cover the ghost points as well, coarse numbers >= split are off-rank; the operator comes back as two blocks
// (Pi/Pj/Pa: columns < split; Oi/Oj/Oa: the others, minus split).
bool device_extpi(int n, const int *Ai, const int *Aj, const double *Aa, const int *Si, const int *Sj, const int *CF,
                  const int *f2c, double trunc_tol, int max_elmts, int first_rung, int **Pi_out, int **Pj_out, double **Pa_out,
                  int *nnz_out, hipStream_t s, int split, int **Oi_out, int **Oj_out, double **Oa_out, int *onnz_out)
{
   if (max_elmts <= 0 && trunc_tol > 0.0) { return false; }     // lengths would depend on the weights: not built here
   const bool dist = split >= 0;
   if (dist && n <= 0)
   {
      int *Pi = nullptr, *Oi = nullptr;
      HIP_CHECK(hipMalloc((void **) &Pi, sizeof(int) * 2)); HIP_CHECK(hipMemsetAsync(Pi, 0, sizeof(int) * 2, s));
      HIP_CHECK(hipMalloc((void **) &Oi, sizeof(int) * 2)); HIP_CHECK(hipMemsetAsync(Oi, 0, sizeof(int) * 2, s));
      HIP_CHECK(hipStreamSynchronize(s));
      *Pi_out = Pi; *Pj_out = nullptr; *Pa_out = nullptr; *nnz_out = 0;
      *Oi_out = Oi; *Oj_out = nullptr; *Oa_out = nullptr; *onnz_out = 0;
      return true;
   }
   auto lds = [](int capM, int capR) { return (size_t) 8 * capM + 8 * capR + 8 * 64 + 4 * capM + 4 * capR + 4 * (4 * capR + 8) + 4 * 64 * 5 + 64; };
   const size_t budget = 150 * 1024;
   (void) hipFuncSetAttribute((const void *) extpi_rows_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) budget);
   (void) hipFuncSetAttribute((const void *) extpi_rows_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) budget);
   (void) hipFuncSetAttribute((const void *) extpi_rows_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) budget);
   int *d_flag = nullptr, *rowlen = nullptr, *rowlen_d = nullptr;
   HIP_CHECK(hipMalloc((void **) &d_flag, sizeof(int)));
   HIP_CHECK(hipMalloc((void **) &rowlen, sizeof(int) * ((size_t) n + 1)));
   if (dist) { HIP_CHECK(hipMalloc((void **) &rowlen_d, sizeof(int) * ((size_t) n + 1))); }
   const int waves = std::min(n, handle().num_cus * 32);
   // rows have at most max_elmts entries: one pass into strided storage (a distributed level always works that way, with
   // room for a whole table's worth of entries per row when the rows are not cut)
   const bool fixed = max_elmts > 0 || dist;
   int *sj = nullptr;
   double *sa = nullptr;
   int stride = max_elmts;
   auto alloc_strided = [&](int st)
   {
      if (sj) { HIP_CHECK(hipFree(sj)); sj = nullptr; }
      if (sa) { HIP_CHECK(hipFree(sa)); sa = nullptr; }
      stride = st;
      if (hipMalloc((void **) &sj, sizeof(int) * (size_t) n * (size_t) st) != hipSuccess) { sj = nullptr; (void) hipGetLastError(); return false; }
      if (hipMalloc((void **) &sa, sizeof(double) * (size_t) n * (size_t) st) != hipSuccess) { sa = nullptr; (void) hipGetLastError(); return false; }
      return true;
   };
   auto give_up = [&]()
   {
      HIP_CHECK(hipFree(d_flag)); HIP_CHECK(hipFree(rowlen));
      if (rowlen_d) { HIP_CHECK(hipFree(rowlen_d)); }
      if (sj) { HIP_CHECK(hipFree(sj)); }
      if (sa) { HIP_CHECK(hipFree(sa)); }
      return false;
   };
   if (fixed && max_elmts > 0 && !alloc_strided(max_elmts)) { return give_up(); }
   // Tables sized by what rows of such operators need, smallest first: the kernel lives on the number of rows in flight
   // (tables for 64 entries: 23 waves a CU; for 256: 7), and the interpolatory sets of the benchmark hierarchy stay below
   // 64 points on every level although the rows of A grow to 134 entries.  An overflowing row sends everyone to the next
   // rung (the waves leave the lost attempt at their next row).
   const int room[4] = {64, 128, 256, 1024};
   int capR = 0, capM = 0, h_flag = 1;
   const bool verbose = getenv("HYPRE_AMD_SETUP_TIMING") != nullptr;
   const int maxA = verbose ? device_max_row_nnz(Ai, n, s) : 0;
   for (int attempt = std::min(std::max(first_rung, 0), 3); attempt < 4 && h_flag; attempt++)
   {
      capR = room[attempt];
      if (verbose) { fprintf(stderr, "   interpolation: %d rows, longest row of A %d, tables for %d entries\n", n, maxA, capR); }
      // the map also holds the strong F neighbours of the row
      capM = pow2_ge(4 * capR);
      if (lds(capM, capR) > budget) { break; }
      HIP_CHECK(hipMemsetAsync(d_flag, 0, sizeof(int), s));
      if (fixed)
      {
         if (max_elmts <= 0 && !alloc_strided(capR)) { return give_up(); }
         hipLaunchKernelGGL((extpi_rows_kernel<2>), dim3(waves), dim3(64), lds(capM, capR), s, n, Ai, Aj, Aa, Si, Sj, CF, f2c, trunc_tol,
                            max_elmts, capM, capR, (const int *) nullptr, stride, rowlen, sj, sa, d_flag, split, rowlen_d);
      }
      else
      {
         hipLaunchKernelGGL((extpi_rows_kernel<0>), dim3(waves), dim3(64), lds(capM, capR), s, n, Ai, Aj, Aa, Si, Sj, CF, f2c, trunc_tol,
                            max_elmts, capM, capR, (const int *) nullptr, 0, rowlen, (int *) nullptr, (double *) nullptr, d_flag, -1, (int *) nullptr);
      }
      HIP_CHECK(hipMemcpyAsync(&h_flag, d_flag, sizeof(int), hipMemcpyDeviceToHost, s));
      HIP_CHECK(hipStreamSynchronize(s));
   }
   if (h_flag) { return give_up(); }
   if (dist)
   {
      device_split_strided(n, stride, rowlen, rowlen_d, sj, sa, split, Pi_out, Pj_out, Pa_out, nnz_out, Oi_out, Oj_out, Oa_out, onnz_out, s);
      (void) give_up();       // frees the work arrays
      return true;
   }
   // row pointers: exclusive scan of the lengths, in place
   launch_scan_exclusive(rowlen, n, s);
   int nnz = 0;
   HIP_CHECK(hipMemcpyAsync(&nnz, rowlen + n, sizeof(int), hipMemcpyDeviceToHost, s));
   HIP_CHECK(hipStreamSynchronize(s));
   if (nnz < 0) { return give_up(); }
   int *Pi = rowlen, *Pj = nullptr;
   double *Pa = nullptr;
   HIP_CHECK(hipMalloc((void **) &Pj, sizeof(int) * (size_t) std::max(nnz, 1)));
   HIP_CHECK(hipMalloc((void **) &Pa, sizeof(double) * (size_t) std::max(nnz, 1)));
   if (fixed)
   {
      const size_t tot = (size_t) n * (size_t) max_elmts;
      hipLaunchKernelGGL(compact_rows_kernel, dim3((unsigned) ((tot + 255) / 256)), dim3(256), 0, s, n, max_elmts, Pi, sj, sa, Pj, Pa);
   }
   else
   {
      HIP_CHECK(hipMemsetAsync(d_flag, 0, sizeof(int), s));
      hipLaunchKernelGGL((extpi_rows_kernel<1>), dim3(waves), dim3(64), lds(capM, capR), s, n, Ai, Aj, Aa, Si, Sj, CF, f2c, trunc_tol,
                         max_elmts, capM, capR, Pi, 0, (int *) nullptr, Pj, Pa, d_flag, -1, (int *) nullptr);
   }
   HIP_CHECK(hipStreamSynchronize(s));
   HIP_CHECK(hipFree(d_flag));
   if (sj) { HIP_CHECK(hipFree(sj)); }
   if (sa) { HIP_CHECK(hipFree(sa)); }
   *Pi_out = Pi; *Pj_out = Pj; *Pa_out = Pa; *nnz_out = nnz;
   return true;
}

// The code object of this file is loaded when one of its kernels is first asked for: ensure_device() asks here, so that
// the load (tens of milliseconds per file) is part of bringing the device up, not of the first setup or solve.
void preload_interp_kernels() { hipFuncAttributes at; (void) hipFuncGetAttributes(&at, (const void *) extpi_rows_kernel<2>); (void) hipGetLastError(); }

}  // namespace hamd
