// hypre_amd — device pieces of the DISTRIBUTED BoomerAMG setup (several ranks, one per GPU).
//
// The single-rank device setup (setup_kernels.hip, interp_kernels.hip, rap_kernels.hip) reproduces the host loops bit for
// bit on one block.  A rank of a distributed matrix owns two blocks — diag (local columns) and offd (ghost columns) — and
// the host routines (par_amg_setup_dist.cpp, restating parcsr_ls/par_lr_interp.c:1024-1700, par_rap.c:30-2000,
// par_coarsen.c:2101-2810, par_strength.c:75-530) walk "diag entries, then offd entries" wherever the single-rank loop
// walks a row.  So the distributed setup runs the SAME kernels on an EXTENDED local index space
//      [0, n)  local points      [n, n + g)  ghost points (columns of offd, then the extra nodes the interpolation finds)
// in which a row is the concatenation [diag row | offd row with columns + n] and the rows the neighbours send (A_ext,
// S_ext, P_ext: parcsr_mv/par_csr_matop.c:1236-1720) are appended as rows n .. n + g - 1.  This file holds what that
// needs around the big kernels: building extended matrices, extracting the rows a neighbour asks for, splitting results
// back into diag / offd blocks, and the smoother diagonals of a level with ghost columns.
// The reference's own device setup (par_coarsen_device.c:30, par_lr_interp_device.c:1001, par_csr_triplemat.c:938-960)
// is a different formulation (vendor random numbers, generic SpGEMM); the results to match are the HOST routines'.
#include "internal.hpp"
#include <algorithm>

#pragma clang fp contract(off)

namespace hamd {

namespace {

constexpr int TB = 256;
inline int grid_for(size_t n) { return (int) std::max<size_t>(1, std::min<size_t>((n + TB - 1) / TB, (size_t) 0x7fffffff)); }

// ---- extended matrices -----------------------------------------------------------------------------------------------
__global__ __launch_bounds__(TB)
void ext_len_kernel(int n, const int *__restrict__ Di, const int *__restrict__ Oi, int ng, const int *__restrict__ Gi,
                    int *__restrict__ len)
{
   const int i = blockIdx.x * TB + threadIdx.x;
   if (i < n) { len[i] = (Di[i + 1] - Di[i]) + (Oi ? Oi[i + 1] - Oi[i] : 0); }
   else if (i < n + ng) { len[i] = Gi ? Gi[i - n + 1] - Gi[i - n] : 0; }
}

template <bool DATA>
__global__ __launch_bounds__(TB)
void ext_fill_kernel(int n, const int *__restrict__ Di, const int *__restrict__ Dj, const double *__restrict__ Da,
                     const int *__restrict__ Oi, const int *__restrict__ Oj, const double *__restrict__ Oa, int ooff,
                     int ng, const int *__restrict__ Gi, const int *__restrict__ Gj, const double *__restrict__ Ga,
                     const int *__restrict__ Ei, int *__restrict__ Ej, double *__restrict__ Ea)
{
   const int i = blockIdx.x * TB + threadIdx.x;
   if (i < n)
   {
      int p = Ei[i];
      for (int k = Di[i]; k < Di[i + 1]; k++) { Ej[p] = Dj[k]; if (DATA) { Ea[p] = Da[k]; } p++; }
      if (Oi) { for (int k = Oi[i]; k < Oi[i + 1]; k++) { Ej[p] = Oj[k] + ooff; if (DATA) { Ea[p] = Oa[k]; } p++; } }
   }
   else if (i < n + ng && Gi)
   {
      int p = Ei[i];
      for (int k = Gi[i - n]; k < Gi[i - n + 1]; k++) { Ej[p] = Gj[k]; if (DATA) { Ea[p] = Ga[k]; } p++; }
   }
}

// ---- rows a neighbour asks for ---------------------------------------------------------------------------------------
// FILTER 0: every entry of [D | O] (rows of P); FILTER 2: the off-diagonal entries of A whose sign is opposite to the
// diagonal's, local ones only when they are C points (par_csr_matop.c ExtractBExt with skip_fine && skip_same_sign)
template <int FILTER, bool FILL>
__global__ __launch_bounds__(TB)
void extract_rows_kernel(int tot, const int *__restrict__ elmts,
                         const int *__restrict__ Di, const int *__restrict__ Dj, const double *__restrict__ Da,
                         const int *__restrict__ Oi, const int *__restrict__ Oj, const double *__restrict__ Oa,
                         long long first_col, const long long *__restrict__ cmap, const int *__restrict__ CF,
                         int *__restrict__ cnt, const int *__restrict__ off, long long *__restrict__ outj, double *__restrict__ outa)
{
   const int s = blockIdx.x * TB + threadIdx.x;
   if (s >= tot) { return; }
   const int r = elmts[s];
   int c = 0, p = FILL ? off[s] : 0;
   if (FILTER == 2)
   {
      const int d0 = Di[r], d1 = Di[r + 1];
      const bool pos = d1 > d0 ? Da[d0] >= 0 : true;
      for (int k = d0 + 1; k < d1; k++)
      {
         const double a = Da[k];
         const bool opp = pos ? a < 0 : a > 0;
         if (opp && CF[Dj[k]] >= 0) { if (FILL) { outj[p] = (long long) Dj[k] + first_col; outa[p] = a; p++; } else { c++; } }
      }
      if (Oi)
      {
         for (int k = Oi[r]; k < Oi[r + 1]; k++)
         {
            const double a = Oa[k];
            const bool opp = pos ? a < 0 : a > 0;
            if (opp) { if (FILL) { outj[p] = cmap[Oj[k]]; outa[p] = a; p++; } else { c++; } }
         }
      }
   }
   else
   {
      for (int k = Di[r]; k < Di[r + 1]; k++) { if (FILL) { outj[p] = (long long) Dj[k] + first_col; outa[p] = Da[k]; p++; } else { c++; } }
      if (Oi) { for (int k = Oi[r]; k < Oi[r + 1]; k++) { if (FILL) { outj[p] = cmap[Oj[k]]; outa[p] = Oa[k]; p++; } else { c++; } } }
   }
   if (!FILL) { cnt[s] = c; }
}

// rows of the extended strength matrix, C columns only (ExtractBExt with skip_fine): a column below n is local, the
// others are ghosts of A
template <bool FILL>
__global__ __launch_bounds__(TB)
void extract_S_kernel(int tot, const int *__restrict__ elmts, int n, const int *__restrict__ Si, const int *__restrict__ Sj,
                      long long first_col, const long long *__restrict__ cmap, const int *__restrict__ CFext,
                      int *__restrict__ cnt, const int *__restrict__ off, long long *__restrict__ outj)
{
   const int s = blockIdx.x * TB + threadIdx.x;
   if (s >= tot) { return; }
   const int r = elmts[s];
   int c = 0, p = FILL ? off[s] : 0;
   for (int k = Si[r]; k < Si[r + 1]; k++)
   {
      const int col = Sj[k];
      if (!(CFext[col] >= 0)) { continue; }
      if (FILL) { outj[p++] = col < n ? (long long) col + first_col : cmap[col - n]; } else { c++; }
   }
   if (!FILL) { cnt[s] = c; }
}

// ---- results back into diag / offd blocks ------------------------------------------------------------------------------
// rows stored with a fixed stride whose first nd entries are the diagonal block's (columns < split): per row the two counts
__global__ __launch_bounds__(TB)
void split_counts_kernel(int n, const int *__restrict__ len, const int *__restrict__ nd, int *__restrict__ cd, int *__restrict__ co)
{
   const int i = blockIdx.x * TB + threadIdx.x;
   if (i < n) { cd[i] = nd[i]; co[i] = len[i] - nd[i]; }
}
__global__ __launch_bounds__(TB)
void split_compact_kernel(int n, int stride, const int *__restrict__ Di, const int *__restrict__ Oi,
                          const int *__restrict__ sj, const double *__restrict__ sa, int split,
                          int *__restrict__ Dj, double *__restrict__ Da, int *__restrict__ Oj, double *__restrict__ Oa)
{
   const size_t t = (size_t) blockIdx.x * TB + threadIdx.x;
   const int i = (int) (t / (size_t) stride), k = (int) (t % (size_t) stride);
   if (i >= n) { return; }
   const int nd = Di[i + 1] - Di[i], no = Oi[i + 1] - Oi[i];
   if (k < nd) { Dj[Di[i] + k] = sj[t]; Da[Di[i] + k] = sa[t]; }
   else if (k < nd + no) { Oj[Oi[i] + k - nd] = sj[t] - split; Oa[Oi[i] + k - nd] = sa[t]; }
}
// pattern only, from a CSR whose rows hold [columns < split ... | columns >= split ...]: the extended strength matrix
__global__ __launch_bounds__(TB)
void split_pattern_counts_kernel(int n, const int *__restrict__ Si, const int *__restrict__ Sj, int split, int *__restrict__ cd, int *__restrict__ co)
{
   const int i = blockIdx.x * TB + threadIdx.x;
   if (i >= n) { return; }
   int d = 0;
   for (int k = Si[i]; k < Si[i + 1]; k++) { if (Sj[k] < split) { d++; } }
   cd[i] = d; co[i] = Si[i + 1] - Si[i] - d;
}
__global__ __launch_bounds__(TB)
void split_pattern_fill_kernel(int n, const int *__restrict__ Si, const int *__restrict__ Sj, int split,
                               const int *__restrict__ Di, const int *__restrict__ Oi, int *__restrict__ Dj, int *__restrict__ Oj)
{
   const int i = blockIdx.x * TB + threadIdx.x;
   if (i >= n) { return; }
   int pd = Di[i], po = Oi[i];
   for (int k = Si[i]; k < Si[i + 1]; k++) { const int c = Sj[k]; if (c < split) { Dj[pd++] = c; } else { Oj[po++] = c - split; } }
}

__global__ __launch_bounds__(TB)
void mark_used_kernel(size_t nnz, const int *__restrict__ j, int *__restrict__ used)
{
   const size_t k = (size_t) blockIdx.x * TB + threadIdx.x;
   if (k < nnz) { used[j[k]] = 1; }
}
__global__ __launch_bounds__(TB)
void renumber_kernel(size_t nnz, int *__restrict__ j, int split, const int *__restrict__ map)
{
   const size_t k = (size_t) blockIdx.x * TB + threadIdx.x;
   if (k < nnz) { const int c = j[k]; if (c >= split) { j[k] = split + map[c - split]; } }
}
__global__ __launch_bounds__(TB)
void gather_int_kernel(size_t n, const int *__restrict__ x, const int *__restrict__ idx, int *__restrict__ out)
{
   const size_t k = (size_t) blockIdx.x * TB + threadIdx.x;
   if (k < n) { out[k] = x[idx[k]]; }
}

// ---- smoother diagonals of a level with ghost columns (ams.c:527-830) --------------------------------------------------
__global__ __launch_bounds__(TB)
void l1_norms_blocks_kernel(int n, const int *__restrict__ Di, const int *__restrict__ Dj, const double *__restrict__ Da,
                            const int *__restrict__ Oi, const int *__restrict__ Oj, const double *__restrict__ Oa,
                            int option, const int *__restrict__ cf, const int *__restrict__ cfo,
                            double *__restrict__ out, int *__restrict__ zero_seen)
{
   const int i = blockIdx.x * TB + threadIdx.x;
   if (i >= n) { return; }
   const int b = Di[i], e = Di[i + 1];
   double diag = 0.0;
   for (int k = b; k < e; k++) { if (Dj[k] == i) { diag = Da[k]; break; } }
   if (option == 5) { out[i] = diag == 0.0 ? 1.0 : diag; return; }
   auto ghost_sum = [&](double s, double scal)
   {
      for (int k = Oi[i]; k < Oi[i + 1]; k++)
      {
         if (cf && cfo && cf[i] != cfo[Oj[k]]) { continue; }
         s += scal * fabs(Oa[k]);
      }
      return s;
   };
   double v;
   if (option == 1)
   {
      v = 0.0;
      for (int k = b; k < e; k++)
      {
         if (cf && cf[i] != cf[Dj[k]]) { continue; }
         v += 1.0 * fabs(Da[k]);
      }
      if (Oi) { v = ghost_sum(v, 1.0); }
   }
   else if (option == 4)
   {
      const double d = fabs(diag);
      v = d;
      if (Oi) { v = ghost_sum(v, 0.5); }
      if (v <= 4.0 / 3.0 * d) { v = d; }
   }
   else       // 6
   {
      v = fabs(diag);
      if (Oi) { const double t = ghost_sum(0.0, 1.0); v = 0.5 * (t + v + sqrt(t * t + v * v)); }
   }
   if (diag < 0.0) { v = -v; }
   if (fabs(v) == 0.0) { *zero_seen = 1; }
   out[i] = v;
}

}  // namespace

// [D row | O row, columns + ooff] for rows 0 .. n-1, then the ng rows of G as they are.  Ea_out == nullptr: pattern only.
// The arrays are device allocations the caller frees with hipFree.
void device_extend_csr(int n, const int *Di, const int *Dj, const double *Da, const int *Oi, const int *Oj, const double *Oa, int ooff,
                       int ng, const int *Gi, const int *Gj, const double *Ga, int **Ei_out, int **Ej_out, double **Ea_out,
                       int *nnz_out, hipStream_t s)
{
   const int rows = n + ng;
   int *Ei = nullptr, *Ej = nullptr;
   double *Ea = nullptr;
   HIP_CHECK(hipMalloc((void **) &Ei, sizeof(int) * ((size_t) rows + 1)));
   HIP_CHECK(hipMemsetAsync(Ei, 0, sizeof(int) * ((size_t) rows + 1), s));
   if (rows > 0) { hipLaunchKernelGGL(ext_len_kernel, dim3(grid_for((size_t) rows)), dim3(TB), 0, s, n, Di, Oi, ng, Gi, Ei); }
   launch_scan_exclusive(Ei, rows, s);
   int nnz = 0;
   HIP_CHECK(hipMemcpyAsync(&nnz, Ei + rows, sizeof(int), hipMemcpyDeviceToHost, s));
   HIP_CHECK(hipStreamSynchronize(s));
   HIP_CHECK(hipMalloc((void **) &Ej, sizeof(int) * (size_t) std::max(nnz, 1)));
   if (Ea_out) { HIP_CHECK(hipMalloc((void **) &Ea, sizeof(double) * (size_t) std::max(nnz, 1))); }
   if (rows > 0)
   {
      if (Ea_out) { hipLaunchKernelGGL((ext_fill_kernel<true>), dim3(grid_for((size_t) rows)), dim3(TB), 0, s, n, Di, Dj, Da, Oi, Oj, Oa, ooff, ng, Gi, Gj, Ga, Ei, Ej, Ea); }
      else { hipLaunchKernelGGL((ext_fill_kernel<false>), dim3(grid_for((size_t) rows)), dim3(TB), 0, s, n, Di, Dj, Da, Oi, Oj, Oa, ooff, ng, Gi, Gj, Ga, Ei, Ej, Ea); }
   }
   *Ei_out = Ei; *Ej_out = Ej; if (Ea_out) { *Ea_out = Ea; }
   *nnz_out = nnz;
}

// The rows of [D | O] named by elmts[0 .. tot) with GLOBAL column numbers, on the host (they travel to the neighbours
// through the host-side row exchange).  filter as extract_rows_kernel; CF (device, local points) is read by filter 2.
void device_extract_rows(int filter, int tot, const int *d_elmts, const int *Di, const int *Dj, const double *Da,
                         const int *Oi, const int *Oj, const double *Oa, long long first_col, const long long *d_cmap, const int *CF,
                         std::vector<int> &hi, std::vector<long long> &hj, std::vector<double> &ha, hipStream_t s)
{
   hi.assign((size_t) tot + 1, 0);
   hj.clear(); ha.clear();
   if (tot <= 0) { return; }
   int *cnt = nullptr;
   HIP_CHECK(hipMalloc((void **) &cnt, sizeof(int) * ((size_t) tot + 1)));
   const int g = grid_for((size_t) tot);
   if (filter == 2) { hipLaunchKernelGGL((extract_rows_kernel<2, false>), dim3(g), dim3(TB), 0, s, tot, d_elmts, Di, Dj, Da, Oi, Oj, Oa, first_col, d_cmap, CF, cnt, nullptr, nullptr, nullptr); }
   else { hipLaunchKernelGGL((extract_rows_kernel<0, false>), dim3(g), dim3(TB), 0, s, tot, d_elmts, Di, Dj, Da, Oi, Oj, Oa, first_col, d_cmap, CF, cnt, nullptr, nullptr, nullptr); }
   launch_scan_exclusive(cnt, tot, s);
   HIP_CHECK(hipMemcpyAsync(hi.data(), cnt, sizeof(int) * ((size_t) tot + 1), hipMemcpyDeviceToHost, s));
   HIP_CHECK(hipStreamSynchronize(s));
   const int nnz = hi[(size_t) tot];
   long long *dj = nullptr;
   double *da = nullptr;
   HIP_CHECK(hipMalloc((void **) &dj, sizeof(long long) * (size_t) std::max(nnz, 1)));
   HIP_CHECK(hipMalloc((void **) &da, sizeof(double) * (size_t) std::max(nnz, 1)));
   if (filter == 2) { hipLaunchKernelGGL((extract_rows_kernel<2, true>), dim3(g), dim3(TB), 0, s, tot, d_elmts, Di, Dj, Da, Oi, Oj, Oa, first_col, d_cmap, CF, nullptr, cnt, dj, da); }
   else { hipLaunchKernelGGL((extract_rows_kernel<0, true>), dim3(g), dim3(TB), 0, s, tot, d_elmts, Di, Dj, Da, Oi, Oj, Oa, first_col, d_cmap, CF, nullptr, cnt, dj, da); }
   hj.resize((size_t) nnz); ha.resize((size_t) nnz);
   if (nnz > 0)
   {
      HIP_CHECK(hipMemcpyAsync(hj.data(), dj, sizeof(long long) * (size_t) nnz, hipMemcpyDeviceToHost, s));
      HIP_CHECK(hipMemcpyAsync(ha.data(), da, sizeof(double) * (size_t) nnz, hipMemcpyDeviceToHost, s));
   }
   HIP_CHECK(hipStreamSynchronize(s));
   HIP_CHECK(hipFree(cnt)); HIP_CHECK(hipFree(dj)); HIP_CHECK(hipFree(da));
}

// same for the extended strength matrix (pattern, C columns only)
void device_extract_S_rows(int tot, const int *d_elmts, int n, const int *Si, const int *Sj, long long first_col, const long long *d_cmap,
                           const int *CFext, std::vector<int> &hi, std::vector<long long> &hj, hipStream_t s)
{
   hi.assign((size_t) tot + 1, 0);
   hj.clear();
   if (tot <= 0) { return; }
   int *cnt = nullptr;
   HIP_CHECK(hipMalloc((void **) &cnt, sizeof(int) * ((size_t) tot + 1)));
   const int g = grid_for((size_t) tot);
   hipLaunchKernelGGL((extract_S_kernel<false>), dim3(g), dim3(TB), 0, s, tot, d_elmts, n, Si, Sj, first_col, d_cmap, CFext, cnt, nullptr, nullptr);
   launch_scan_exclusive(cnt, tot, s);
   HIP_CHECK(hipMemcpyAsync(hi.data(), cnt, sizeof(int) * ((size_t) tot + 1), hipMemcpyDeviceToHost, s));
   HIP_CHECK(hipStreamSynchronize(s));
   const int nnz = hi[(size_t) tot];
   long long *dj = nullptr;
   HIP_CHECK(hipMalloc((void **) &dj, sizeof(long long) * (size_t) std::max(nnz, 1)));
   hipLaunchKernelGGL((extract_S_kernel<true>), dim3(g), dim3(TB), 0, s, tot, d_elmts, n, Si, Sj, first_col, d_cmap, CFext, nullptr, cnt, dj);
   hj.resize((size_t) nnz);
   if (nnz > 0) { HIP_CHECK(hipMemcpyAsync(hj.data(), dj, sizeof(long long) * (size_t) nnz, hipMemcpyDeviceToHost, s)); }
   HIP_CHECK(hipStreamSynchronize(s));
   HIP_CHECK(hipFree(cnt)); HIP_CHECK(hipFree(dj));
}

// Rows stored with stride `stride` (sj / sa), row i holding len[i] entries of which the first nd[i] have columns < split:
// two CSR blocks, the second with columns - split.  len and nd are consumed (they become the row pointers' storage).
void device_split_strided(int n, int stride, int *len, int *nd, const int *sj, const double *sa, int split,
                          int **Di_out, int **Dj_out, double **Da_out, int *dnnz, int **Oi_out, int **Oj_out, double **Oa_out, int *onnz,
                          hipStream_t s)
{
   int *Di = nullptr, *Oi = nullptr;
   HIP_CHECK(hipMalloc((void **) &Di, sizeof(int) * ((size_t) n + 1)));
   HIP_CHECK(hipMalloc((void **) &Oi, sizeof(int) * ((size_t) n + 1)));
   HIP_CHECK(hipMemsetAsync(Di, 0, sizeof(int) * ((size_t) n + 1), s));
   HIP_CHECK(hipMemsetAsync(Oi, 0, sizeof(int) * ((size_t) n + 1), s));
   if (n > 0) { hipLaunchKernelGGL(split_counts_kernel, dim3(grid_for((size_t) n)), dim3(TB), 0, s, n, len, nd, Di, Oi); }
   launch_scan_exclusive(Di, n, s);
   launch_scan_exclusive(Oi, n, s);
   int nn[2] = {0, 0};
   HIP_CHECK(hipMemcpyAsync(&nn[0], Di + n, sizeof(int), hipMemcpyDeviceToHost, s));
   HIP_CHECK(hipMemcpyAsync(&nn[1], Oi + n, sizeof(int), hipMemcpyDeviceToHost, s));
   HIP_CHECK(hipStreamSynchronize(s));
   int *Dj = nullptr, *Oj = nullptr;
   double *Da = nullptr, *Oa = nullptr;
   HIP_CHECK(hipMalloc((void **) &Dj, sizeof(int) * (size_t) std::max(nn[0], 1)));
   HIP_CHECK(hipMalloc((void **) &Da, sizeof(double) * (size_t) std::max(nn[0], 1)));
   HIP_CHECK(hipMalloc((void **) &Oj, sizeof(int) * (size_t) std::max(nn[1], 1)));
   HIP_CHECK(hipMalloc((void **) &Oa, sizeof(double) * (size_t) std::max(nn[1], 1)));
   const size_t tot = (size_t) n * (size_t) stride;
   if (tot > 0) { hipLaunchKernelGGL(split_compact_kernel, dim3(grid_for(tot)), dim3(TB), 0, s, n, stride, Di, Oi, sj, sa, split, Dj, Da, Oj, Oa); }
   HIP_CHECK(hipStreamSynchronize(s));
   *Di_out = Di; *Dj_out = Dj; *Da_out = Da; *dnnz = nn[0];
   *Oi_out = Oi; *Oj_out = Oj; *Oa_out = Oa; *onnz = nn[1];
}

// extended strength pattern (n rows, columns < split local, others ghosts) -> the two blocks' patterns
void device_split_pattern(int n, const int *Si, const int *Sj, int split, int **Di_out, int **Dj_out, int *dnnz,
                          int **Oi_out, int **Oj_out, int *onnz, hipStream_t s)
{
   int *Di = nullptr, *Oi = nullptr;
   HIP_CHECK(hipMalloc((void **) &Di, sizeof(int) * ((size_t) n + 1)));
   HIP_CHECK(hipMalloc((void **) &Oi, sizeof(int) * ((size_t) n + 1)));
   HIP_CHECK(hipMemsetAsync(Di, 0, sizeof(int) * ((size_t) n + 1), s));
   HIP_CHECK(hipMemsetAsync(Oi, 0, sizeof(int) * ((size_t) n + 1), s));
   if (n > 0) { hipLaunchKernelGGL(split_pattern_counts_kernel, dim3(grid_for((size_t) n)), dim3(TB), 0, s, n, Si, Sj, split, Di, Oi); }
   launch_scan_exclusive(Di, n, s);
   launch_scan_exclusive(Oi, n, s);
   int nn[2] = {0, 0};
   HIP_CHECK(hipMemcpyAsync(&nn[0], Di + n, sizeof(int), hipMemcpyDeviceToHost, s));
   HIP_CHECK(hipMemcpyAsync(&nn[1], Oi + n, sizeof(int), hipMemcpyDeviceToHost, s));
   HIP_CHECK(hipStreamSynchronize(s));
   int *Dj = nullptr, *Oj = nullptr;
   HIP_CHECK(hipMalloc((void **) &Dj, sizeof(int) * (size_t) std::max(nn[0], 1)));
   HIP_CHECK(hipMalloc((void **) &Oj, sizeof(int) * (size_t) std::max(nn[1], 1)));
   if (n > 0) { hipLaunchKernelGGL(split_pattern_fill_kernel, dim3(grid_for((size_t) n)), dim3(TB), 0, s, n, Si, Sj, split, Di, Oi, Dj, Oj); }
   HIP_CHECK(hipStreamSynchronize(s));
   *Di_out = Di; *Dj_out = Dj; *dnnz = nn[0]; *Oi_out = Oi; *Oj_out = Oj; *onnz = nn[1];
}

// used[c] = 1 for every column c of j[0 .. nnz)
void launch_mark_used(const int *j, size_t nnz, int *used, hipStream_t s)
{
   if (nnz > 0) { hipLaunchKernelGGL(mark_used_kernel, dim3(grid_for(nnz)), dim3(TB), 0, s, nnz, j, used); }
}
// columns c >= split become split + map[c - split]
void launch_renumber(int *j, size_t nnz, int split, const int *map, hipStream_t s)
{
   if (nnz > 0) { hipLaunchKernelGGL(renumber_kernel, dim3(grid_for(nnz)), dim3(TB), 0, s, nnz, j, split, map); }
}
void launch_gather_int(const int *x, const int *idx, int *out, size_t n, hipStream_t s)
{
   if (n > 0) { hipLaunchKernelGGL(gather_int_kernel, dim3(grid_for(n)), dim3(TB), 0, s, n, x, idx, out); }
}

// smoother diagonal of a level with ghost columns; cf / cfo: C/F markers of the local and of the ghost points (CF-ordered
// relaxation) or nullptr.  Returns false when a row came out zero.
bool device_l1_norms_blocks(int n, const int *Di, const int *Dj, const double *Da, const int *Oi, const int *Oj, const double *Oa,
                            int option, const int *cf, const int *cfo, double *out, hipStream_t s)
{
   if (n <= 0) { return true; }
   int *flag = reinterpret_cast<int *>(reduce_scratch(2));
   HIP_CHECK(hipMemsetAsync(flag, 0, sizeof(int), s));
   hipLaunchKernelGGL(l1_norms_blocks_kernel, dim3(grid_for((size_t) n)), dim3(TB), 0, s, n, Di, Dj, Da, Oi, Oj, Oa, option, cf, cfo, out, flag);
   int h = 0;
   HIP_CHECK(hipMemcpyAsync(&h, flag, sizeof(int), hipMemcpyDeviceToHost, s));
   HIP_CHECK(hipStreamSynchronize(s));
   return h == 0;
}

void preload_dist_setup_kernels() { hipFuncAttributes at; (void) hipFuncGetAttributes(&at, (const void *) ext_len_kernel); (void) hipGetLastError(); }

}  // namespace hamd
