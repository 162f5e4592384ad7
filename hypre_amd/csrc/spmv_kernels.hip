// hypre_amd — CSR SpMV kernel family for gfx950 (CDNA4), fp64 values / int32 indices.
//
// One kernel template serves y = alpha*A*x + beta*b, the fused Jacobi / l1-Jacobi
// sweep, and the two-stage Gauss-Seidel inner step (row epilogues), so a
// smoother sweep is a single pass over the matrix.
//
// Tiled ("stream") kernel, one 256-thread workgroup per tile of 2048 stored
// entries (rows are binned into tiles by the position of their first entry),
// eight workgroups resident per CU:
//   phase 1  (col, val) streamed with unconditional 16-byte loads (2 x 4 entries
//            per lane in flight), issued before the tile's bounds are known;
//   phase 0  row pointers of the tile -> LDS, epilogue operands (b, d, x of the
//            row each lane will finish) -> registers, while the stream is in flight;
//   phase 1' every wave transposes its column indices through its own slice of
//            LDS so that one gather instruction covers 64 consecutive entries
//            (few x cache lines), parks the gathered x in LDS in entry order and
//            reads back the four values it owns; products stay in LDS;
//   phase 2  per-row reduction from LDS with 1, 8 or 32 lanes per row (chosen
//            per tile from its mean row length), row sums of the multi-lane
//            paths go through LDS so that the epilogue is one coalesced pass.
// Tiles are dealt to the 8 XCDs (each with its own L2) in runs of 8 consecutive
// tiles; the measured alternatives are listed at spmv_default_flags.
//
// Replaces (behaviourally) seq_mv/csr_spmv_device.c:35-260 of the reference,
// which uses K lanes per row chosen from the matrix-wide average row length,
// no LDS, and one kernel per operation (copy, SpMV, elementwise update).
#include "amg_internal.hpp"

namespace hamd {

typedef int    v4i __attribute__((ext_vector_type(4)));
typedef int    v2i __attribute__((ext_vector_type(2)));
typedef double v2d __attribute__((ext_vector_type(2)));
typedef float  v4f __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double wave_sum64(double v)
{
#pragma unroll
   for (int off = 32; off > 0; off >>= 1) { v += __shfl_xor(v, off, 64); }
   return v;
}

template <int W>
__device__ __forceinline__ double subwave_sum(double v)
{
#pragma unroll
   for (int off = W / 2; off > 0; off >>= 1) { v += __shfl_xor(v, off, 64); }
   return v;
}

__device__ __forceinline__ bool fill_keep(int fill, int row, int col)
{
   switch (fill)
   {
      case HYPRE_SPMV_FILL_STRICT_LOWER: return col <  row;
      case HYPRE_SPMV_FILL_LOWER:        return col <= row;
      case HYPRE_SPMV_FILL_UPPER:        return col >= row;
      case HYPRE_SPMV_FILL_STRICT_UPPER: return col >  row;
      default:                           return true;
   }
}

// operands of one row's epilogue, fetched early
struct RowOps
{
   double b, d, x;
   int    m;
   int    g;      // row of the vectors the epilogue reads and writes (= the matrix row unless a row map is given)
};

template <int OP>
__device__ __forceinline__ RowOps load_row_ops(const SpmvArgs &p, int row)
{
   RowOps o;
   o.b = 0.0; o.d = 1.0; o.x = 0.0; o.m = 0;
   // the row map is its own operation: a test on p.rowmap here merges the two paths and the compiler then waits for
   // every load in flight (the tile's whole stream) before it asks for b, d and x — one more round trip for every sweep
   if (OP == OP_JACOBI_MAP) { row = p.rowmap[row]; }
   o.g = row;
   if (OP == OP_AXPBY) { if (p.beta != 0.0) { o.b = p.b[row]; } }
   else if (OP == OP_AXPBY_DIV || OP == OP_RESID_RD) { if (p.beta != 0.0) { o.b = p.b[row]; } o.d = p.d[row]; }
   else if (OP == OP_TSGS_FIRST) { o.d = p.d[row]; o.b = p.x[row]; if (p.beta != 0.0) { o.x = p.aux[row]; } }      // (x is z_in: its own row as well)
   else if (OP == OP_TSGS) { o.d = p.d[row]; o.x = p.aux[row]; }      // aux: read here, with the other operands, not in the epilogue
   else
   {
      o.b = p.b[row]; o.d = p.d[row]; o.x = p.x[row];
      if (OP == OP_JACOBI_CF) { o.m = p.marker[row]; }
      if (OP == OP_JACOBI_MAP) { o.m = p.marker ? p.marker[row] : p.marker_val; }
   }
   return o;
}

template <int OP>
__device__ __forceinline__ void row_epilogue(const SpmvArgs &p, int, double sum, const RowOps &o)
{
   const int row = o.g;
   if (OP == OP_AXPBY || OP == OP_AXPBY_DIV)
   {
      // y = alpha*(A x) + beta*b   (roundings spelled out: every kernel of the family gives the same bits for the same row sum)
      double r = __dmul_rn(p.alpha, sum);
      if (p.beta != 0.0) { r = __fma_rn(p.beta, o.b, r); }
      p.y[row] = r;
      // the restriction that also starts the coarse level's sweep from zero: u = (w f) ./ d, as scaled_div_kernel rounds it
      // (an operation of its own: its operand costs the plain product of the x-staged kernel its eighth wave per SIMD)
      if (OP == OP_AXPBY_DIV) { p.aux[row] = __dmul_rn(p.scale2, r) / o.d; }
   }
   else if (OP == OP_RESID_RD)
   {
      // the two-stage sweep's first stage in one pass: z = (w f - w A u) .* (1 ./ D) — the residual's roundings (as OP_AXPBY
      // forms it), then the product with the reciprocal as diagscale2_kernel forms it
      double r = __dmul_rn(p.alpha, sum);
      if (p.beta != 0.0) { r = __fma_rn(p.beta, o.b, r); }
      p.y[row] = __dmul_rn(r, 1.0 / o.d);
   }
   else if (OP == OP_TSGS_FIRST)
   {
      // first inner step, with the "u += z" of the first stage folded in: z' = (L z) .* (1 ./ D) ; u = (u + z) + mult z'
      // (u + z rounded first: what the separate pass stored; from a zero iterate o.x is 0)
      const double z = __dmul_rn(sum, 1.0 / o.d);
      p.y[row] = z;
      p.aux[row] = __fma_rn(p.alpha, z, __dadd_rn(o.x, o.b));
   }
   else if (OP == OP_TSGS)
   {
      // inner step of the two-stage Gauss-Seidel sweep (par_relax_device.c:139-150):
      //    z_out = (L_strict z_in) ./ D ;  u += mult * z_out
      const double z = __dmul_rn(sum, 1.0 / o.d);
      p.y[row] = z;
      p.aux[row] = __fma_rn(p.alpha, z, o.x);
   }
   else
   {
      // Jacobi / l1-Jacobi sweep fused into the SpMV pass (par_relax.c:1216-1244):
      //    y = x + (w f - w (A x)) ./ d      on the marked rows, y = x elsewhere
      if ((OP == OP_JACOBI_CF || OP == OP_JACOBI_MAP) && o.m != p.marker_val) { p.y[row] = o.x; return; }
      const double t = __fma_rn(p.alpha, o.b, -__dmul_rn(p.alpha, sum));
      p.y[row] = __dadd_rn(o.x, t / o.d);
   }
}

#ifndef SPMV_REDUCE_BATCH
#define SPMV_REDUCE_BATCH 4
#endif
constexpr int RB = SPMV_REDUCE_BATCH;   // products a lane of the row reduction has in flight
constexpr int RP_CAP = 640;      // upper bound of the row pointers staged in LDS per tile (the plan asks for fewer
                                 // when no tile of the matrix holds that many rows)

// a workgroup barrier that orders LDS accesses only: __syncthreads() also waits for every global load in flight
__device__ __forceinline__ void lds_barrier()
{
   __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
   __builtin_amdgcn_s_barrier();
   __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// row sums of a tile with W lanes per row (products in LDS, sums to rowsum[]); W is a compile-time width: the shuffle
// tree and the lane arithmetic unroll (a run-time W cost levels 2+ of the benchmark hierarchy 5 %)
template <bool HASFILL, int W>
__device__ __forceinline__ void tile_row_sums(const SpmvArgs &p, int r0, int nrows, int ka, const double *prod, double *rowsum,
                                              const int *rp)
{
   constexpr int G = SPMV_THREADS / W;
   const int tid = threadIdx.x, sub = tid & (W - 1);
   for (int base = 0; base < nrows; base += G)
   {
      const int rr = base + tid / W;
      double sum = 0.0;
      if (rr < nrows)
      {
         const int row = r0 + rr;
         const int s = rp[rr], e = rp[rr + 1];
         // RB products at a time: the LDS reads of a batch are in flight together (clamped addresses, so that they are
         // unconditional), the additions keep the order of the one-by-one loop
         for (int k = s + sub; k < e; k += RB * W)
         {
            double t[RB];
#pragma unroll
            for (int i = 0; i < RB; i++) { t[i] = prod[min(k + i * W, e - 1) - ka]; }
#pragma unroll
            for (int i = 0; i < RB; i++)
            {
               if (k + i * W < e)
               {
                  if (HASFILL) { if (!fill_keep(p.fill, row, p.Aj[k + i * W])) { t[i] = 0.0; } }
                  sum += t[i];
               }
            }
         }
      }
      sum = subwave_sum<W>(sum);
      if (rr < nrows && sub == 0) { rowsum[rr] = sum; }
   }
}

// Per-row reduction of the products parked in LDS and the row epilogue.
// prod[k - ka] holds entry k of the tile, rp[rr] the row pointer of row r0 + rr
// (the first rp_cap + 1 of them), ops the epilogue operands of row r0 + tid.
// OPS2: the caller also fetched the operands of row r0 + SPMV_THREADS + tid early (tiles of short rows hold more rows than
// the workgroup has lanes: a 7-point tile 293, an interpolation tile ~512): without them the second pass asks for its
// operands when it needs them, a third trip to memory at the end of the tile's life.
template <int OP, bool HASFILL, bool OPS2 = false>
__device__ __forceinline__ void tile_reduce(const SpmvArgs &p, int r0, int nrows, int k0, int k1, int ka,
                                            const double *prod, double *rowsum, const int *rp, int rp_cap,
                                            const RowOps &ops, const RowOps *ops2 = nullptr)
{
   const int tid = threadIdx.x;
   const int avg = (k1 - k0) / nrows;
   if (avg <= 12)
   {
      // one lane per row; entries are summed in stored order
      for (int rr = tid; rr < nrows; rr += SPMV_THREADS)
      {
         const int row = r0 + rr;
         const int s = (rr     <= rp_cap) ? rp[rr]     : p.Ai[row];
         const int e = (rr + 1 <= rp_cap) ? rp[rr + 1] : p.Ai[row + 1];
         double sum = 0.0;
         // RB products at a time (see tile_row_sums)
         for (int k = s; k < e; k += RB)
         {
            double t[RB];
#pragma unroll
            for (int i = 0; i < RB; i++) { t[i] = prod[min(k + i, e - 1) - ka]; }
#pragma unroll
            for (int i = 0; i < RB; i++)
            {
               if (k + i < e)
               {
                  if (HASFILL) { if (!fill_keep(p.fill, row, p.Aj[k + i])) { t[i] = 0.0; } }
                  sum += t[i];
               }
            }
         }
         if (rr == tid) { row_epilogue<OP>(p, row, sum, ops); }
         else if (OPS2 && rr == tid + SPMV_THREADS) { row_epilogue<OP>(p, row, sum, *ops2); }
         else { const RowOps o = load_row_ops<OP>(p, row); row_epilogue<OP>(p, row, sum, o); }
      }
   }
   else
   {
      // avg > 12  =>  nrows <= (TILE + MAXROW)/13 < SPMV_THREADS: row sums fit rowsum[]
      // W lanes per row, W the largest power of two (at most 32) that takes all rows of the tile in ONE pass: a second
      // pass with most lanes idle costs as much as the first (level 1 of the benchmark hierarchy: 70 rows a tile were
      // three passes of 32 rows with 8 lanes each; two lanes a row take them at once)
      int W = p.reduce_w;                                   // > 0: fixed (tuning knob)
      if (W <= 0) { W = 32; while (W > 1 && nrows * W > SPMV_THREADS) { W >>= 1; } }
      switch (W)
      {
         case 1:  tile_row_sums<HASFILL, 1>(p, r0, nrows, ka, prod, rowsum, rp); break;
         case 2:  tile_row_sums<HASFILL, 2>(p, r0, nrows, ka, prod, rowsum, rp); break;
         case 4:  tile_row_sums<HASFILL, 4>(p, r0, nrows, ka, prod, rowsum, rp); break;
         case 8:  tile_row_sums<HASFILL, 8>(p, r0, nrows, ka, prod, rowsum, rp); break;
         case 16: tile_row_sums<HASFILL, 16>(p, r0, nrows, ka, prod, rowsum, rp); break;
         default: tile_row_sums<HASFILL, 32>(p, r0, nrows, ka, prod, rowsum, rp); break;
      }
      lds_barrier();                 // row sums are in LDS
      if (tid < nrows) { row_epilogue<OP>(p, r0 + tid, rowsum[tid], ops); }
   }
}


// The slice of the (col, val) streams one lane holds in registers: two quads.
// Loads are issued unconditionally (lanes past the tile's end re-read its first
// quad) so that no register merge forces a wait right behind the loads.
struct TileStream
{
   v4i cA, cB;
   v2d vA01, vA23, vB01, vB23;     // fp64 values
   v4f fA, fB;                     // fp32 values (mixed precision)
   // the spill of the tile's last row past the streamed window, one entry per lane (entry ka + TILE + tid), requested
   // as soon as the tile's bounds are known so that it travels with the stream instead of after it
   int    cC;
   double vC;
};

template <typename T>
__device__ __forceinline__ T stream_load(const void *p)
{
   return *reinterpret_cast<const T *>(p);
}

// issue the stream loads of the tile whose entries are [k0, k1), ka = k0 & ~3
template <bool F32>
__device__ __forceinline__ void stream_issue(const SpmvArgs &p, int ka, int k1, TileStream &s)
{
   const int kA = ka + 4 * (int) threadIdx.x;
   const int kB = kA + 4 * SPMV_THREADS;
   const int qA = min(kA < k1 ? kA : ka, p.last_quad);
   const int qB = min(kB < k1 ? kB : ka, p.last_quad);
   s.cA = stream_load<v4i>(p.Aj + qA);
   s.cB = stream_load<v4i>(p.Aj + qB);
   if (F32)
   {
      s.fA = stream_load<v4f>(p.Aa32 + qA);
      s.fB = stream_load<v4f>(p.Aa32 + qB);
   }
   else
   {
      s.vA01 = stream_load<v2d>(p.Aa + qA);
      s.vA23 = stream_load<v2d>(p.Aa + qA + 2);
      s.vB01 = stream_load<v2d>(p.Aa + qB);
      s.vB23 = stream_load<v2d>(p.Aa + qB + 2);
   }
}

// the spill entries: lanes without one all re-read the first spill entry (one cache line)
template <bool F32>
__device__ __forceinline__ void stream_issue_spill(const SpmvArgs &p, int ka, int k1, TileStream &s)
{
   const int kC = ka + 8 * SPMV_THREADS + (int) threadIdx.x;
   // a spill entry is an entry of the matrix (kC < k1 <= nnz): only the stand-in address needs the clamp
   const int qC = kC < k1 ? kC : min(ka + 8 * SPMV_THREADS, p.last_quad);
   s.cC = p.Aj[qC];
   s.vC = F32 ? (double) p.Aa32[qC] : p.Aa[qC];
}

// gather x for the entries held in s, park the products in LDS (prod[k - ka])
template <bool F32>
__device__ __forceinline__ void stream_consume(const SpmvArgs &p, int k0, int k1, int ka, const TileStream &s,
                                               double *prod)
{
   const int kA = ka + 4 * (int) threadIdx.x;
   const int kB = kA + 4 * SPMV_THREADS;
   const int kC = ka + 8 * SPMV_THREADS + (int) threadIdx.x;
   const double xC = (kC < k1) ? p.x[s.cC] : 0.0;
   double vA0, vA1, vA2, vA3, vB0, vB1, vB2, vB3;
   if (F32)
   {
      vA0 = s.fA.x; vA1 = s.fA.y; vA2 = s.fA.z; vA3 = s.fA.w;
      vB0 = s.fB.x; vB1 = s.fB.y; vB2 = s.fB.z; vB3 = s.fB.w;
   }
   else
   {
      vA0 = s.vA01.x; vA1 = s.vA01.y; vA2 = s.vA23.x; vA3 = s.vA23.y;
      vB0 = s.vB01.x; vB1 = s.vB01.y; vB2 = s.vB23.x; vB3 = s.vB23.y;
   }
   if (kA < k1)
   {
      double *dst = prod + (kA - ka);
      if (kA >= k0 && kA + 4 <= k1)
      {
         const double x0 = p.x[s.cA.x], x1 = p.x[s.cA.y], x2 = p.x[s.cA.z], x3 = p.x[s.cA.w];
         *reinterpret_cast<double2 *>(dst)     = make_double2(vA0 * x0, vA1 * x1);
         *reinterpret_cast<double2 *>(dst + 2) = make_double2(vA2 * x2, vA3 * x3);
      }
      else
      {
         if (kA     >= k0 && kA     < k1) { dst[0] = vA0 * p.x[s.cA.x]; }
         if (kA + 1 >= k0 && kA + 1 < k1) { dst[1] = vA1 * p.x[s.cA.y]; }
         if (kA + 2 >= k0 && kA + 2 < k1) { dst[2] = vA2 * p.x[s.cA.z]; }
         if (kA + 3 >= k0 && kA + 3 < k1) { dst[3] = vA3 * p.x[s.cA.w]; }
      }
   }
   if (kB < k1)
   {
      double *dst = prod + (kB - ka);
      if (kB + 4 <= k1)
      {
         const double x0 = p.x[s.cB.x], x1 = p.x[s.cB.y], x2 = p.x[s.cB.z], x3 = p.x[s.cB.w];
         *reinterpret_cast<double2 *>(dst)     = make_double2(vB0 * x0, vB1 * x1);
         *reinterpret_cast<double2 *>(dst + 2) = make_double2(vB2 * x2, vB3 * x3);
      }
      else
      {
         if (kB     < k1) { dst[0] = vB0 * p.x[s.cB.x]; }
         if (kB + 1 < k1) { dst[1] = vB1 * p.x[s.cB.y]; }
         if (kB + 2 < k1) { dst[2] = vB2 * p.x[s.cB.z]; }
         if (kB + 3 < k1) { dst[3] = vB3 * p.x[s.cB.w]; }
      }
   }
   if (kC < k1) { prod[kC - ka] = s.vC * xC; }
   // tail of a tile whose last row runs past the spill entries as well
   for (int k = ka + 9 * SPMV_THREADS + 4 * (int) threadIdx.x; k < k1; k += 4 * SPMV_THREADS)
   {
      const v4i c = stream_load<v4i>(p.Aj + k);
      double v0, v1, v2, v3;
      if (F32) { const v4f f = stream_load<v4f>(p.Aa32 + k); v0 = f.x; v1 = f.y; v2 = f.z; v3 = f.w; }
      else
      {
         const v2d lo = stream_load<v2d>(p.Aa + k), hi = stream_load<v2d>(p.Aa + k + 2);
         v0 = lo.x; v1 = lo.y; v2 = hi.x; v3 = hi.y;
         if (p.dict_rounded) { v0 = (double) (float) v0; v1 = (double) (float) v1; v2 = (double) (float) v2; v3 = (double) (float) v3; }
      }
      double *dst = prod + (k - ka);
      if (k     < k1) { dst[0] = v0 * p.x[c.x]; }
      if (k + 1 < k1) { dst[1] = v1 * p.x[c.y]; }
      if (k + 2 < k1) { dst[2] = v2 * p.x[c.z]; }
      if (k + 3 < k1) { dst[3] = v3 * p.x[c.w]; }
   }
}

// Same result, different lane <-> entry pairing for the gathers.  A lane streams
// 4 consecutive entries (one 16-byte load per array), so the k-th gather of a wave
// touches every 4th entry of a 256-entry chunk: the x lines of up to 256/rowlen rows.
// Here the wave passes its column indices through LDS so that gather c covers the 64
// consecutive entries 64c..64c+63 of the chunk (the x lines of a few neighbouring rows
// only), parks the gathered x values in LDS in entry order, and every lane reads back
// the 4 it owns.  All of it happens inside the wave's own 2 KB of the product area:
// no workgroup barrier.  Slots outside the tile's [k0, k1) gather x[0] and their products are
// never read.
template <bool F32>
__device__ __forceinline__ void stream_consume_gt(const SpmvArgs &p, int k0, int k1, int ka, const TileStream &s,
                                                  double *prod)
{
   const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
   const int kC = ka + 8 * SPMV_THREADS + (int) threadIdx.x;
   const double xC = (kC < k1) ? p.x[s.cC] : 0.0;
#pragma unroll
   for (int half = 0; half < 2; half++)
   {
      double *chunk = prod + half * (4 * SPMV_THREADS) + 256 * wave;     // entries ka + half*1024 + 256*wave ...
      int    *ci = reinterpret_cast<int *>(chunk);
      *reinterpret_cast<v4i *>(ci + 4 * lane) = half ? s.cB : s.cA;
      __builtin_amdgcn_wave_barrier();
      // entries outside [k0, k1) may lie past the end of the arrays (the tail of the last
      // 16-byte quad holds no entry): their "columns" must not be dereferenced
      const int e0 = ka + half * (4 * SPMV_THREADS) + 256 * wave + lane;
      const int c0 = (e0       >= k0 && e0       < k1) ? ci[lane]       : 0;
      const int c1 = (e0 + 64  >= k0 && e0 + 64  < k1) ? ci[64 + lane]  : 0;
      const int c2 = (e0 + 128 >= k0 && e0 + 128 < k1) ? ci[128 + lane] : 0;
      const int c3 = (e0 + 192 >= k0 && e0 + 192 < k1) ? ci[192 + lane] : 0;
      const double x0 = p.x[c0], x1 = p.x[c1], x2 = p.x[c2], x3 = p.x[c3];
      __builtin_amdgcn_wave_barrier();
      chunk[lane] = x0; chunk[64 + lane] = x1; chunk[128 + lane] = x2; chunk[192 + lane] = x3;
      __builtin_amdgcn_wave_barrier();
      const v2d xa = *reinterpret_cast<const v2d *>(chunk + 4 * lane);
      const v2d xb = *reinterpret_cast<const v2d *>(chunk + 4 * lane + 2);
      v2d lo, hi;
      if (F32)
      {
         const v4f f = half ? s.fB : s.fA;
         lo.x = (double) f.x * xa.x; lo.y = (double) f.y * xa.y; hi.x = (double) f.z * xb.x; hi.y = (double) f.w * xb.y;
      }
      else
      {
         const v2d v01 = half ? s.vB01 : s.vA01, v23 = half ? s.vB23 : s.vA23;
         lo = v01 * xa; hi = v23 * xb;
      }
      *reinterpret_cast<v2d *>(chunk + 4 * lane)     = lo;
      *reinterpret_cast<v2d *>(chunk + 4 * lane + 2) = hi;
   }
   if (kC < k1) { prod[kC - ka] = s.vC * xC; }
   // tail of a tile whose last row runs past the spill entries as well
   for (int k = ka + 9 * SPMV_THREADS + 4 * (int) threadIdx.x; k < k1; k += 4 * SPMV_THREADS)
   {
      const v4i c = stream_load<v4i>(p.Aj + k);
      double v0, v1, v2, v3;
      if (F32) { const v4f f = stream_load<v4f>(p.Aa32 + k); v0 = f.x; v1 = f.y; v2 = f.z; v3 = f.w; }
      else
      {
         const v2d lo = stream_load<v2d>(p.Aa + k), hi = stream_load<v2d>(p.Aa + k + 2);
         v0 = lo.x; v1 = lo.y; v2 = hi.x; v3 = hi.y;
         if (p.dict_rounded) { v0 = (double) (float) v0; v1 = (double) (float) v1; v2 = (double) (float) v2; v3 = (double) (float) v3; }
      }
      double *dst = prod + (k - ka);
      if (k     < k1) { dst[0] = v0 * p.x[c.x]; }
      if (k + 1 < k1) { dst[1] = v1 * p.x[c.y]; }
      if (k + 2 < k1) { dst[2] = v2 * p.x[c.z]; }
      if (k + 3 < k1) { dst[3] = v3 * p.x[c.w]; }
   }
}

// epilogue operands of row r0 + tid, clamped into the tile so the loads are unconditional
template <int OP>
__device__ __forceinline__ RowOps tile_row_ops(const SpmvArgs &p, int r0, int nrows)
{
   const int rr = min((int) threadIdx.x, nrows - 1);
   return load_row_ops<OP>(p, max(r0 + rr, 0));
}

// One workgroup per tile, in dispatch order (or a re-mapped order, see xcd_map).
template <int OP, bool F32, bool HASFILL, bool GT>
__global__ __launch_bounds__(SPMV_THREADS)
void spmv_tiled_kernel(SpmvArgs p, const int *__restrict__ tile_row, const int *__restrict__ tile_k,
                       int num_tiles, int prod_elems, int rowsum_elems, int rp_cap)
{
   extern __shared__ __align__(16) unsigned char smem_raw[];
   double *prod   = reinterpret_cast<double *>(smem_raw);
   double *rowsum = prod + prod_elems;                                 // [rowsum_elems]
   int    *rp     = reinterpret_cast<int *>(rowsum + rowsum_elems);    // [rp_cap + 1]

   // Workgroups are dealt round-robin over the 8 XCDs (workgroup g -> XCD g % 8),
   // each XCD with its own L2.  xcd_map > 0: every XCD takes chunks of xcd_map
   // consecutive tiles; xcd_map < 0: one contiguous eighth of the tiles per XCD.
   // Speed only: any placement is correct.
   // The grid may be padded past num_tiles (launch_tiled_gt): a padding workgroup must leave before it touches the
   // placement table, which holds num_tiles entries.  Every tile is visited exactly once — the in-place epilogues
   // (OP_TSGS adds into aux, OP_AXPBY with b == y) rely on it — so a table entry outside [0, num_tiles) is dropped too.
   int tile = (int) blockIdx.x;
   if (p.tile_perm)
   {
      if (tile >= num_tiles) { return; }
      tile = p.tile_perm[tile];
   }
   else if (p.xcd_map > 0)
   {
      const int g = blockIdx.x >> 3, c = blockIdx.x & 7, C = p.xcd_map;
      tile = (g / C) * (8 * C) + c * C + (g % C);
   }
   else if (p.xcd_map < 0)
   {
      tile = (blockIdx.x & 7) * ((num_tiles + 7) >> 3) + (blockIdx.x >> 3);
   }
   if ((unsigned) tile >= (unsigned) num_tiles) { return; }

   // The tile's entries [k0, k1) start inside [tile*TILE, tile*TILE + longest row): the first
   // TILE entries from tile*TILE on are requested before the tile's bounds are known, so the
   // matrix stream is in flight while the bounds, then the row pointers and this lane's
   // epilogue operands arrive.  Entries before k0 belong to the previous tile and are dropped;
   // entries past tile*TILE + TILE (the spill of the last row) are picked up by the tail loop.
   const int ka = tile * SPMV_TILE;
   TileStream S;
   stream_issue<F32>(p, ka, 0x7fffffff, S);

   const int r0 = tile_row[tile];
   const int r1 = tile_row[tile + 1];
   if (r1 <= r0) { return; }
   const int k0 = tile_k[tile];
   const int k1 = tile_k[tile + 1];
   const int tid = threadIdx.x;
   const int nrows = r1 - r0;
   stream_issue_spill<F32>(p, ka, k1, S);
   // fixed trip count (rp_cap <= RP_CAP): an open-ended loop here is unrolled into a register-hungry
   // load pipeline that costs the kernel its eighth wave per SIMD
#pragma unroll
   for (int j = 0; j < (RP_CAP + SPMV_THREADS) / SPMV_THREADS; j++)
   {
      const int t = tid + j * SPMV_THREADS;
      if (t <= nrows && t <= rp_cap) { rp[t] = p.Ai[r0 + t]; }
   }
   const RowOps ops = tile_row_ops<OP>(p, r0, nrows);

   if (GT) { stream_consume_gt<F32>(p, k0, k1, ka, S, prod); } else { stream_consume<F32>(p, k0, k1, ka, S, prod); }
   __syncthreads();
   tile_reduce<OP, HASFILL>(p, r0, nrows, k0, k1, ka, prod, rowsum, rp, rp_cap, ops);
}

// ---------------------------------------------------------------------------
// x-staged form of the tiled kernel.
//
// Measured on the 256^3 hierarchy (profiles/r02_*): with the x gathers taken out, the tiled kernel streams levels 0 / 1 / 2
// at 6.2 / 5.6 / 5.7 TB/s; with them, at 5.4 / 4.1 / 3.3.  What the gathers cost is not their bytes (L2 hits) and not
// their instructions (a first x-staged kernel that fetched per-tile chunk lists and then the chunks ran no faster): a
// CU's vector-memory pipeline returns data in order, so a load that DEPENDS on an earlier load — a gather on its column
// index, a chunk on its list entry — queues behind the HBM-bound matrix stream of the eight resident workgroups and
// costs a whole second trip through it.  A tile's life is the number of such dependent trips, and with a fixed number
// of resident tiles that sets the rate.
//
// So the x values a tile needs are fetched in the SAME trip as its matrix stream.  The plan records per tile up to 48
// pieces of x, 128 doubles at most each, that cover the columns of its entries (a 2048-entry tile touches 700 - 1400
// distinct 2-column units in a handful of clusters: rows that are neighbours in the matrix share most of their
// columns), and per entry a 16-bit index into the concatenation of those pieces.  The piece descriptors are
// wave-uniform and arrive through the scalar cache in one batch with the tile's bounds; each wave then issues one
// 16-byte load per lane for each of its twelve pieces right behind the (value, index) stream loads, the copy lands in
// LDS (global_load_lds), and the "gathers" are eight LDS reads per lane.  The column array is not read at all: 8 + 2
// bytes per entry instead of 12.  Tiles whose columns do not fit 48 pieces / 4096 doubles (most tiles of the
// restriction operators below level 0: a row of P^T reaches far) take the gather path of spmv_tiled_kernel inside the
// same launch.  The staged copy and the products share LDS.
// Tried and dropped: pieces of up to two loads (256 doubles) — fewer, longer pieces, but 24 predicated load slots per
// wave instead of 12: levels 1 - 2 of the benchmark hierarchy 4 - 7 % slower, the restriction operators no better
// (what keeps their tiles out is the staged volume, not the number of pieces).
// Reference counterpart of the whole family: seq_mv/csr_spmv_device.c:35-260 (no LDS, gathers through the cache).
// ---------------------------------------------------------------------------
constexpr int XS_SEGS  = SPMV_XS_SEGS;      // pieces per tile: 4 waves x XS_WSEG
constexpr int XS_WSEG  = XS_SEGS / 4;       // pieces a wave fetches
constexpr int XS_CAP   = SPMV_XS_CAP;       // doubles a tile may stage at most (the launch sizes LDS by what the matrix needs)
#ifndef XS_EARLY_STREAM
#define XS_EARLY_STREAM 1
#endif
constexpr int XS_DESC  = 2 * XS_SEGS;       // ints per tile in the plan: first column of every piece, then (offset << 16 | length)

#ifndef XS_TIMING
#define XS_TIMING 0
#endif
#ifndef XS_OPS2
#define XS_OPS2 1              // operands of a second row per lane fetched with the first (0: as before, for comparison)
#endif
// experiments (wrong products, timing only; tools/experiments/build_variant.sh): XS_EXP_SLOTS < XS_WSEG stages only the first
// pieces of every wave, XS_EXP_NOREDUCE replaces the row sums by one store per lane
#ifndef XS_EXP_SLOTS
#define XS_EXP_SLOTS 1000
#endif
#ifndef XS_EXP_NOREDUCE
#define XS_EXP_NOREDUCE 0
#endif
#if XS_TIMING
// experiment (tools/experiments/tile_phases.sh): where a tile's life goes, in ticks of the 100 MHz wall clock, summed over
// the tiles of the last launch, four numbers per tile: entry -> scalar batch back, -> stream and x pieces landed (first
// barrier), -> products parked, -> row sums and epilogue issued.  Plain stores into a buffer the host hands over.
__device__ unsigned *xs_timing_buf;
#define XS_STAMP(n) const unsigned long long ts##n = wall_clock64()
#else
#define XS_STAMP(n)
#endif
// VF, the form the matrix values are streamed in: 0 fp64, 1 fp32 (mixed precision), 2 one-byte codes into a table of at
// most 256 values that every tile stages in LDS with its x pieces (SpmvPlan::d_codes).
constexpr int VF_F64 = 0, VF_F32 = 1, VF_CODE = 2;
constexpr int DICT_CAP = 256;
// bytes of LDS in front of the value table: products, row sums, row pointers, rounded up to the table's 16-byte alignment
__host__ __device__ inline size_t xs_dict_offset(int prod_elems, int rowsum_elems, int rp_cap)
{
   return (sizeof(double) * (size_t) (prod_elems + rowsum_elems) + sizeof(int) * (size_t) (rp_cap + 4) + 15) & ~(size_t) 15;
}

template <int OP, int VF, bool HASFILL>
__global__ __launch_bounds__(SPMV_THREADS)
void spmv_xs_kernel(SpmvArgs p, const int *__restrict__ tile_row, const int *__restrict__ tile_k,
                    const int *__restrict__ xs_cnt, const int *__restrict__ xs_desc,
                    const unsigned short *__restrict__ lidx,
                    int num_tiles, int prod_elems, int rowsum_elems, int rp_cap, int xs_units)
{
   constexpr bool F32 = VF == VF_F32, CODED = VF == VF_CODE;
   extern __shared__ __align__(16) unsigned char smem_raw[];
   double *prod   = reinterpret_cast<double *>(smem_raw);     // staged x first, products after the second barrier
   double *rowsum = prod + prod_elems;
   int    *rp     = reinterpret_cast<int *>(rowsum + rowsum_elems);
   const double *dictl = reinterpret_cast<const double *>(smem_raw + xs_dict_offset(prod_elems, rowsum_elems, rp_cap));   // CODED only

   int tile = (int) blockIdx.x;
   if (p.tile_perm)
   {
      if (tile >= num_tiles) { return; }
      tile = p.tile_perm[tile];
   }
   else if (p.xcd_map > 0)
   {
      const int g = blockIdx.x >> 3, c = blockIdx.x & 7, C = p.xcd_map;
      tile = (g / C) * (8 * C) + c * C + (g % C);
   }
   else if (p.xcd_map < 0) { tile = (blockIdx.x & 7) * ((num_tiles + 7) >> 3) + (blockIdx.x >> 3); }
   if ((unsigned) tile >= (unsigned) num_tiles) { return; }

   const int tid = threadIdx.x, lane = tid & 63;
   const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
   const int ka = tile * SPMV_TILE;
   XS_STAMP(0);
   // Values and local indices of the tile's window, then — wave-uniform, through the scalar cache — the tile's bounds
   // and this wave's twelve piece descriptors: ONE batch of requests issued before anything about the tile is known.
   // Left alone, the compiler sinks every one of these loads below the first branch that does not need it (the stream
   // below the empty-tile test, the descriptors below the fallback test), and each such move puts a whole round trip
   // to memory in front of the x pieces: a tile's life is the sum of its round trips.  The empty asm after the batch
   // claims it may write what the pointers point to, so no load of the batch can be carried across it; it touches no
   // loaded value, so nothing waits there.
   TileStream S;
   const int kA = ka + 4 * tid, kB = kA + 4 * SPMV_THREADS;
   const int qA = min(kA, p.last_quad), qB = min(kB, p.last_quad);
   unsigned cdA = 0, cdB = 0;                   // CODED: four value codes each
#if XS_EARLY_STREAM
   if (CODED)
   {
      cdA = stream_load<unsigned>(p.Ac8 + qA);
      cdB = stream_load<unsigned>(p.Ac8 + qB);
   }
   else if (F32)
   {
      S.fA = stream_load<v4f>(p.Aa32 + qA);
      S.fB = stream_load<v4f>(p.Aa32 + qB);
   }
   else
   {
      S.vA01 = stream_load<v2d>(p.Aa + qA); S.vA23 = stream_load<v2d>(p.Aa + qA + 2);
      S.vB01 = stream_load<v2d>(p.Aa + qB); S.vB23 = stream_load<v2d>(p.Aa + qB + 2);
   }
   const v2i lA = stream_load<v2i>(lidx + qA), lB = stream_load<v2i>(lidx + qB);
#endif
   const int *dsc = xs_desc + (size_t) tile * XS_DESC + XS_WSEG * wave;
   int seg_start[XS_WSEG], seg_ol[XS_WSEG];
#pragma unroll
   for (int j = 0; j < XS_WSEG; j++) { seg_start[j] = dsc[j]; seg_ol[j] = dsc[XS_SEGS + j]; }
   const int r0 = tile_row[tile], r1 = tile_row[tile + 1];
   const int k0 = tile_k[tile], k1 = tile_k[tile + 1];
   const int xc = xs_cnt[tile];                // (covered units << 8) | pieces
   // staleness watch (SpmvPlan): two entries of the column array this kernel otherwise never reads, against the
   // fingerprint the plan took of them; in mixed precision one fp64 value against its fp32 copy.  Wave-uniform
   // addresses: they ride in the scalar batch above.
   // (unconditional loads: the launch always hands over a fingerprint table and a flag word — a test on a pointer here
   // puts a scalar round trip of its own in front of the batch)
   const int qs0 = min(ka + 5, p.last_quad), qs1 = min(ka + 1029, p.last_quad);
   const int fp_plan = p.tile_fp[tile];
   const int fc0 = p.Aj[qs0], fc1 = p.Aj[qs1];
   // value codes: the kernel multiplies by a private copy of the values, so every launch compares a ROTATING sample of the
   // copy with the caller's array — eight consecutive entries per wave (one 64-byte request), at a position that moves with
   // the plan's launch counter: the four waves of a tile cover its 2048 entries in 64 launches, so any coefficient edited
   // in place is found within 64 products, whichever it is.  Loaded here, in the tile's one trip; compared once the table
   // has landed.
   // (mixed precision: the fp32 copy of the values is checked the same way)
   double   ckv = 0.0;
   unsigned ckc = 0;
   float    ckf = 0.0f;
   if (CODED || F32)
   {
      const int ck = min(ka + 8 * (int) ((4u * p.rot + (unsigned) wave) & 255u) + (lane & 7), p.nnz - 1);
      ckv = p.Aa[ck];
      if (CODED) { ckc = p.Ac8[ck]; } else { ckf = p.Aa32[ck]; }
   }
   asm volatile("" :: "s"(tile_row), "s"(tile_k), "s"(xs_cnt), "s"(xs_desc), "s"(lidx), "s"(p.Aa), "s"(p.Aa32), "s"(p.Aj), "s"(p.tile_fp), "s"(p.Ac8) : "memory");
   if (r1 <= r0) { return; }
   {
      const bool off = fp_plan != (int) ((unsigned) fc0 * 2654435761u + (unsigned) fc1);
      if (off && tid == 0) { __hip_atomic_fetch_or(p.stale, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
   }
   if (CODED)
   {
      // the value table: 16 bytes per lane straight into LDS, in the same trip as the stream (the table is allocated, and
      // readable, up to DICT_CAP entries whatever it holds)
      const int dl = (p.ndict + 1) >> 1;                   // lanes the table takes
      const int mine = dl - 64 * wave;
      if (lane < mine)
      {
         __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *) (p.dict + 128 * wave + 2 * lane),
                                          (__attribute__((address_space(3))) void *) (reinterpret_cast<char *>(const_cast<double *>(dictl)) + 1024 * wave), 16, 0, 0);
      }
   }
#if !XS_EARLY_STREAM
   if (CODED)
   {
      cdA = stream_load<unsigned>(p.Ac8 + qA);
      cdB = stream_load<unsigned>(p.Ac8 + qB);
   }
   else if (F32)
   {
      S.fA = stream_load<v4f>(p.Aa32 + qA);
      S.fB = stream_load<v4f>(p.Aa32 + qB);
   }
   else
   {
      S.vA01 = stream_load<v2d>(p.Aa + qA); S.vA23 = stream_load<v2d>(p.Aa + qA + 2);
      S.vB01 = stream_load<v2d>(p.Aa + qB); S.vB23 = stream_load<v2d>(p.Aa + qB + 2);
   }
   const v2i lA = stream_load<v2i>(lidx + qA), lB = stream_load<v2i>(lidx + qB);
#endif
   const int nrows = r1 - r0;
   const int nseg = (xc >> 8) <= xs_units ? (xc & 0xff) : 0;     // the launch may stage less than the plan allows (occupancy)
   XS_STAMP(1);

   if (nseg == 0)
   {
      // this tile's columns do not fit the staging area: gathers through the cache, as spmv_tiled_kernel
      S.cA = stream_load<v4i>(p.Aj + qA);
      S.cB = stream_load<v4i>(p.Aj + qB);
      stream_issue_spill<F32>(p, ka, k1, S);
      // row pointers: loaded unconditionally (clamped) and stored afterwards — a conditional load-and-store here makes
      // the compiler wait for everything in flight before each store: three more round trips per tile
      {
         constexpr int RPJ = (RP_CAP + SPMV_THREADS) / SPMV_THREADS;
         const int lim = min(nrows, rp_cap);
         int rpv[RPJ];
#pragma unroll
         for (int j = 0; j < RPJ; j++) { rpv[j] = p.Ai[r0 + min(tid + j * SPMV_THREADS, lim)]; }
#pragma unroll
         for (int j = 0; j < RPJ; j++) { rp[min(tid + j * SPMV_THREADS, lim)] = rpv[j]; }
      }
      const RowOps ops = tile_row_ops<OP>(p, r0, nrows);
      if (CODED)
      {
         // decode into the fp64 slots of the stream, then as the uncoded tile; the spill and the tail of a very long last
         // row were read from the fp64 array by the calls above and below
         __syncthreads();
         S.vA01.x = dictl[cdA & 0xff]; S.vA01.y = dictl[(cdA >> 8) & 0xff]; S.vA23.x = dictl[(cdA >> 16) & 0xff]; S.vA23.y = dictl[cdA >> 24];
         S.vB01.x = dictl[cdB & 0xff]; S.vB01.y = dictl[(cdB >> 8) & 0xff]; S.vB23.x = dictl[(cdB >> 16) & 0xff]; S.vB23.y = dictl[cdB >> 24];
         if (__double_as_longlong(dictl[ckc]) != __double_as_longlong(p.dict_rounded ? (double) (float) ckv : ckv))
         {
            __hip_atomic_fetch_or(p.stale, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
         }
         if (p.dict_rounded) { S.vC = (double) (float) S.vC; }
      }
      stream_consume_gt<F32>(p, k0, k1, ka, S, prod);
      __syncthreads();
      if (F32 && (float) ckv != ckf) { __hip_atomic_fetch_or(p.stale, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
      tile_reduce<OP, HASFILL>(p, r0, nrows, k0, k1, ka, prod, rowsum, rp, rp_cap, ops);
      return;
   }

   // x pieces (at most 128 doubles each: one 16-byte load per lane), requested in the same trip through the memory
   // pipeline as the stream above and written straight into LDS (global_load_lds: no registers, no ds_write pass; the
   // LDS address is the piece's base + 16 * lane, which is exactly how a piece is laid out).  A piece starts at an even
   // column and x is 16-byte aligned, so a lane's 16 bytes never straddle a page: when x has an odd length the upper
   // half of its last pair is read (not used) but cannot fault.
#pragma unroll
   for (int j = 0; j < (XS_WSEG < XS_EXP_SLOTS ? XS_WSEG : XS_EXP_SLOTS); j++)
   {
      // (the plan keeps what the slot needs as it needs it: the piece's LDS offset in bytes, its length in lanes, an
      // unsigned first column — a slot is a dozen instructions per wave whether it loads or not)
      const unsigned offb = (unsigned) seg_ol[j] >> 16, lanes = (unsigned) seg_ol[j] & 0xffffu;
      if ((unsigned) lane < lanes)
      {
         __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *) (p.x + (size_t) (unsigned) seg_start[j] + 2 * lane),
                                          (__attribute__((address_space(3))) void *) (reinterpret_cast<char *>(prod) + offb), 16, 0, 0);
      }
   }
   // spill of the tile's last row past the window (one entry per lane), row pointers, epilogue operands: same trip
   const int kC = ka + 8 * SPMV_THREADS + tid;
   const int qC = kC < k1 ? kC : min(ka + 8 * SPMV_THREADS, p.last_quad);
   const unsigned lC = lidx[qC];
   unsigned cdC = 0;
   if (CODED) { cdC = p.Ac8[qC]; } else { S.vC = F32 ? (double) p.Aa32[qC] : p.Aa[qC]; }
   constexpr int RPJ = (RP_CAP + SPMV_THREADS) / SPMV_THREADS;
   const int lim = min(nrows, rp_cap);
   int rpv[RPJ];
#pragma unroll
   for (int j = 0; j < RPJ; j++) { rpv[j] = p.Ai[r0 + min(tid + j * SPMV_THREADS, lim)]; }
   const RowOps ops = tile_row_ops<OP>(p, r0, nrows);
   // the operands of a second row per lane (tiles of short rows), where the registers allow: unconditional, clamped
   // (y = alpha A x + beta b: one operand; the sweeps' three do not fit 64 registers beside the fp64 or coded stream)
   constexpr bool OPS2 = XS_OPS2 && !HASFILL && (OP == OP_AXPBY || (OP == OP_JACOBI && VF == VF_F32));
   RowOps ops2 = ops;
   if (OPS2) { ops2 = load_row_ops<OP>(p, max(r0 + min(tid + SPMV_THREADS, nrows - 1), 0)); }
   // lanes past the tile's rows repeat its last pointer into the last slot
#pragma unroll
   for (int j = 0; j < RPJ; j++) { rp[min(tid + j * SPMV_THREADS, lim)] = rpv[j]; }
   if (tid == 0 && rpv[0] != k0) { __hip_atomic_fetch_or(p.stale, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }   // the tile table is not this matrix's
   __syncthreads();
   XS_STAMP(2);

   // "gathers": eight LDS reads per lane.  Entries of the window that belong to a neighbouring tile carry that tile's
   // numbering: any index below the staging capacity reads initialised-or-not LDS and the product is never used.
   const double xa0 = prod[lA.x & 0xffff], xa1 = prod[(unsigned) lA.x >> 16], xa2 = prod[lA.y & 0xffff], xa3 = prod[(unsigned) lA.y >> 16];
   const double xb0 = prod[lB.x & 0xffff], xb1 = prod[(unsigned) lB.x >> 16], xb2 = prod[lB.y & 0xffff], xb3 = prod[(unsigned) lB.y >> 16];
   const double xC = prod[lC];
   v2d lo0, hi0, lo1, hi1;
   if (CODED)
   {
      // the values: eight more LDS reads (a stencil's few values: broadcasts)
      lo0.x = dictl[cdA & 0xff] * xa0; lo0.y = dictl[(cdA >> 8) & 0xff] * xa1; hi0.x = dictl[(cdA >> 16) & 0xff] * xa2; hi0.y = dictl[cdA >> 24] * xa3;
      lo1.x = dictl[cdB & 0xff] * xb0; lo1.y = dictl[(cdB >> 8) & 0xff] * xb1; hi1.x = dictl[(cdB >> 16) & 0xff] * xb2; hi1.y = dictl[cdB >> 24] * xb3;
      S.vC = dictl[cdC];
      if (__double_as_longlong(dictl[ckc]) != __double_as_longlong(p.dict_rounded ? (double) (float) ckv : ckv))
      {
         __hip_atomic_fetch_or(p.stale, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
   }
   else if (F32)
   {
      if ((float) ckv != ckf) { __hip_atomic_fetch_or(p.stale, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
      lo0.x = (double) S.fA.x * xa0; lo0.y = (double) S.fA.y * xa1; hi0.x = (double) S.fA.z * xa2; hi0.y = (double) S.fA.w * xa3;
      lo1.x = (double) S.fB.x * xb0; lo1.y = (double) S.fB.y * xb1; hi1.x = (double) S.fB.z * xb2; hi1.y = (double) S.fB.w * xb3;
   }
   else
   {
      lo0.x = S.vA01.x * xa0; lo0.y = S.vA01.y * xa1; hi0.x = S.vA23.x * xa2; hi0.y = S.vA23.y * xa3;
      lo1.x = S.vB01.x * xb0; lo1.y = S.vB01.y * xb1; hi1.x = S.vB23.x * xb2; hi1.y = S.vB23.y * xb3;
   }
   const double pC = S.vC * xC;
   __syncthreads();               // every lane has read its x values: the products may overwrite the staged copy
   *reinterpret_cast<v2d *>(prod + (kA - ka))     = lo0;
   *reinterpret_cast<v2d *>(prod + (kA - ka) + 2) = hi0;
   *reinterpret_cast<v2d *>(prod + (kB - ka))     = lo1;
   *reinterpret_cast<v2d *>(prod + (kB - ka) + 2) = hi1;
   if (kC < k1) { prod[kC - ka] = pC; }
   // entries past the spill (a last row longer than 256 + what is left of the window): their x comes through the cache
   for (int k = ka + 9 * SPMV_THREADS + 4 * tid; k < k1; k += 4 * SPMV_THREADS)
   {
      const v4i c = stream_load<v4i>(p.Aj + k);
      double v0, v1, v2, v3;
      if (F32) { const v4f f = stream_load<v4f>(p.Aa32 + k); v0 = f.x; v1 = f.y; v2 = f.z; v3 = f.w; }
      else
      {
         const v2d lo = stream_load<v2d>(p.Aa + k), hi = stream_load<v2d>(p.Aa + k + 2);
         v0 = lo.x; v1 = lo.y; v2 = hi.x; v3 = hi.y;
         if (p.dict_rounded) { v0 = (double) (float) v0; v1 = (double) (float) v1; v2 = (double) (float) v2; v3 = (double) (float) v3; }
      }
      double *dst = prod + (k - ka);
      if (k     < k1) { dst[0] = v0 * p.x[c.x]; }
      if (k + 1 < k1) { dst[1] = v1 * p.x[c.y]; }
      if (k + 2 < k1) { dst[2] = v2 * p.x[c.z]; }
      if (k + 3 < k1) { dst[3] = v3 * p.x[c.w]; }
   }
   __syncthreads();
   XS_STAMP(3);
#if XS_EXP_NOREDUCE
   if (tid < nrows) { row_epilogue<OP>(p, r0 + tid, prod[tid], ops); }
#else
   tile_reduce<OP, HASFILL, OPS2>(p, r0, nrows, k0, k1, ka, prod, rowsum, rp, rp_cap, ops, &ops2);
#endif
#if XS_TIMING
   {
      const unsigned long long ts4 = wall_clock64();
      if (tid == 0 && xs_timing_buf)
      {
         *reinterpret_cast<uint4 *>(xs_timing_buf + 4 * (size_t) tile) = make_uint4((unsigned) (ts1 - ts0), (unsigned) (ts2 - ts1), (unsigned) (ts3 - ts2), (unsigned) (ts4 - ts3));
      }
   }
#endif
}

// ---------------------------------------------------------------------------
// Several right-hand sides in ONE pass over the matrix: Y(:, v) = alpha A X(:, v) + beta B(:, v), v < NV, multivectors stored
// column by column (seq_mv/vector.h:22-40).  The reference sums NV products per entry in registers (host:
// seq_mv/csr_matvec.c:117-380, NV = 2, 3, 4; device: csr_spmv_device.c:37-134); the column loop of seq_mv.cpp streams the
// matrix NV times.  Here a tile of the x-staged form fetches its window of values and local indices ONCE and the x pieces of
// all NV columns in the same trip (NV staged copies side by side in LDS), parks values and indices in LDS — not products: a
// parked product serves one column, a parked value all of them — and a lane per row (W lanes per row where the rows are
// long) sums its entries for the NV columns out of LDS: 1.25 + NV reads per entry instead of NV x (read x, write the
// product, read it again), one barrier instead of two per column.
// Same bits as NV single-vector products: every product rounded, then the additions of tile_reduce in its order (stored
// order on one lane; W interleaved partial sums and the xor tree on W lanes, W chosen as tile_reduce chooses it).
// Serves fp64 and coded matrices (VF_F64 / VF_CODE) multiplied as a whole; everything else keeps the column loop.
// ---------------------------------------------------------------------------
template <int NV, bool CODED, int W>
__device__ __forceinline__ void mv_row_sums(const SpmvArgs &p, int r0, int nrows, int ka, int rp_cap, bool staged, int stage_elems,
                                            const double *xsv, const double *valS, const unsigned char *cdS, const double *dictl,
                                            const unsigned short *liS, const int *rp, long xstride, long bstride, long ystride,
                                            const double *bv)
{
   constexpr int MB = 4;                     // entries a lane has in flight (values, indices, then NV x each): the additions keep their order
   const int G = (int) blockDim.x / W;       // rows a pass takes (the workgroup may have more lanes than the 256 that stream the tile)
   const int tid = threadIdx.x, sub = tid & (W - 1);
   for (int base = 0; base < nrows; base += G)
   {
      const int rr = base + tid / W;
      const bool live = rr < nrows;
      const int row = r0 + min(rr, nrows - 1);
      int s = 0, e = 0;
      if (live)
      {
         s = (rr     <= rp_cap) ? rp[rr]     : p.Ai[row];
         e = (rr + 1 <= rp_cap) ? rp[rr + 1] : p.Ai[row + 1];
      }
      double sum[NV];
#pragma unroll
      for (int v = 0; v < NV; v++) { sum[v] = 0.0; }
      for (int k = s + sub; k < e; k += MB * W)
      {
         double a[MB], xv[MB][NV];
         if (staged)
         {
            unsigned li[MB];
#pragma unroll
            for (int i = 0; i < MB; i++)
            {
               const int q = min(k + i * W, e - 1) - ka;
               a[i] = CODED ? dictl[cdS[q]] : valS[q];
               li[i] = liS[q];
            }
#pragma unroll
            for (int i = 0; i < MB; i++)
            {
#pragma unroll
               for (int v = 0; v < NV; v++) { xv[i][v] = xsv[v * stage_elems + li[i]]; }
            }
         }
         else
         {
            // a tile whose columns do not fit the staging area: x through the cache
            int c[MB];
#pragma unroll
            for (int i = 0; i < MB; i++)
            {
               const int q = min(k + i * W, e - 1);
               a[i] = CODED ? dictl[cdS[q - ka]] : valS[q - ka];
               c[i] = p.Aj[q];
            }
#pragma unroll
            for (int i = 0; i < MB; i++)
            {
#pragma unroll
               for (int v = 0; v < NV; v++) { xv[i][v] = p.x[(size_t) v * xstride + c[i]]; }
            }
         }
#pragma unroll
         for (int i = 0; i < MB; i++)
         {
            if (k + i * W < e)
            {
#pragma unroll
               for (int v = 0; v < NV; v++)
               {
#pragma clang fp contract(off)
                  const double pr = a[i] * xv[i][v];
                  sum[v] = sum[v] + pr;
               }
            }
         }
      }
      if (W > 1)
      {
#pragma unroll
         for (int v = 0; v < NV; v++) { sum[v] = subwave_sum<W>(sum[v]); }
      }
      if (live && sub == 0)
      {
#pragma unroll
         for (int v = 0; v < NV; v++)
         {
            // the epilogue of OP_AXPBY (row_epilogue)
            double r = __dmul_rn(p.alpha, sum[v]);
            if (p.beta != 0.0) { r = __fma_rn(p.beta, (W == 1 && rr == tid) ? bv[v] : p.b[(size_t) v * bstride + row], r); }
            p.y[(size_t) v * ystride + row] = r;
         }
      }
   }
}

constexpr int MV_THREADS_MAX = 384;       // tiles of short rows hold more rows than 256: a fifth and sixth wave take them in the same pass
template <int NV, bool CODED>
__global__ __launch_bounds__(MV_THREADS_MAX)
void spmv_xs_mv_kernel(SpmvArgs p, const int *__restrict__ tile_row, const int *__restrict__ tile_k,
                       const int *__restrict__ xs_cnt, const int *__restrict__ xs_desc, const unsigned short *__restrict__ lidx,
                       int num_tiles, int stage_elems, int win_elems, int rp_cap, int xs_units, long xstride, long bstride, long ystride)
{
   extern __shared__ __align__(16) unsigned char smem_raw[];
   double *xsv  = reinterpret_cast<double *>(smem_raw);                               // NV staged copies of x, stage_elems each
   double *valS = xsv + (size_t) NV * stage_elems;                                    // the window's values (CODED: its codes, one byte each)
   unsigned char *cdS = reinterpret_cast<unsigned char *>(valS);
   unsigned short *liS = reinterpret_cast<unsigned short *>(reinterpret_cast<unsigned char *>(valS) + (CODED ? (size_t) win_elems : sizeof(double) * (size_t) win_elems));
   int *rp = reinterpret_cast<int *>(liS + win_elems);
   const double *dictl = reinterpret_cast<const double *>(rp + ((rp_cap + 4) & ~3));  // CODED only

   int tile = (int) blockIdx.x;
   if (p.tile_perm)
   {
      if (tile >= num_tiles) { return; }
      tile = p.tile_perm[tile];
   }
   else if (p.xcd_map > 0)
   {
      const int g = blockIdx.x >> 3, c = blockIdx.x & 7, C = p.xcd_map;
      tile = (g / C) * (8 * C) + c * C + (g % C);
   }
   if ((unsigned) tile >= (unsigned) num_tiles) { return; }

   const int tid = threadIdx.x, lane = tid & 63;
   const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
   const bool streamer = wave < 4;            // waves 4 and 5 (tiles of more than 256 rows) only sum rows
   const int ka = tile * SPMV_TILE;
   // one batch of requests, as spmv_xs_kernel: the window's values (or codes) and local indices, the tile's bounds and
   // this wave's piece descriptors
   const int kA = ka + 4 * tid, kB = kA + 4 * SPMV_THREADS;
   const int qA = min(kA, p.last_quad), qB = min(kB, p.last_quad);
   unsigned cdA = 0, cdB = 0;
   v2d vA01 = {0.0, 0.0}, vA23 = {0.0, 0.0}, vB01 = {0.0, 0.0}, vB23 = {0.0, 0.0};
   v2i lA = {0, 0}, lB = {0, 0};
   int seg_start[XS_WSEG], seg_ol[XS_WSEG];
   int fp_plan = 0, fc0 = 0, fc1 = 0;
   double   ckv = 0.0;
   unsigned ckc = 0;
   if (streamer)
   {
      if (CODED)
      {
         cdA = stream_load<unsigned>(p.Ac8 + qA);
         cdB = stream_load<unsigned>(p.Ac8 + qB);
      }
      else
      {
         vA01 = stream_load<v2d>(p.Aa + qA); vA23 = stream_load<v2d>(p.Aa + qA + 2);
         vB01 = stream_load<v2d>(p.Aa + qB); vB23 = stream_load<v2d>(p.Aa + qB + 2);
      }
      lA = stream_load<v2i>(lidx + qA); lB = stream_load<v2i>(lidx + qB);
      const int *dsc = xs_desc + (size_t) tile * XS_DESC + XS_WSEG * wave;
#pragma unroll
      for (int j = 0; j < XS_WSEG; j++) { seg_start[j] = dsc[j]; seg_ol[j] = dsc[XS_SEGS + j]; }
      const int qs0 = min(ka + 5, p.last_quad), qs1 = min(ka + 1029, p.last_quad);
      fp_plan = p.tile_fp[tile];
      fc0 = p.Aj[qs0]; fc1 = p.Aj[qs1];
      // the rotating value check of a coded matrix (see spmv_xs_kernel)
      if (CODED)
      {
         const int ck = min(ka + 8 * (int) ((4u * p.rot + (unsigned) wave) & 255u) + (lane & 7), p.nnz - 1);
         ckv = p.Aa[ck];
         ckc = p.Ac8[ck];
      }
   }
   const int r0 = tile_row[tile], r1 = tile_row[tile + 1];
   const int k0 = tile_k[tile], k1 = tile_k[tile + 1];
   const int xc = xs_cnt[tile];
   asm volatile("" :: "s"(tile_row), "s"(tile_k), "s"(xs_cnt), "s"(xs_desc), "s"(lidx), "s"(p.Aa), "s"(p.Aj), "s"(p.tile_fp), "s"(p.Ac8) : "memory");
   if (r1 <= r0) { return; }
   const int nrows = r1 - r0;
   const int nseg = (xc >> 8) <= xs_units ? (xc & 0xff) : 0;
   const bool staged = nseg != 0;
   if (streamer)
   {
      {
         const bool off = fp_plan != (int) ((unsigned) fc0 * 2654435761u + (unsigned) fc1);
         if (off && tid == 0) { __hip_atomic_fetch_or(p.stale, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
      }
      if (CODED)
      {
         const int dl = (p.ndict + 1) >> 1;
         if (lane < dl - 64 * wave)
         {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *) (p.dict + 128 * wave + 2 * lane),
                                             (__attribute__((address_space(3))) void *) (reinterpret_cast<char *>(const_cast<double *>(dictl)) + 1024 * wave), 16, 0, 0);
         }
      }
      if (staged)
      {
         // the x pieces of every column: NV copies of the tile's staging area, one after the other
#pragma unroll
         for (int v = 0; v < NV; v++)
         {
#pragma unroll
            for (int j = 0; j < XS_WSEG; j++)
            {
               const unsigned offb = (unsigned) seg_ol[j] >> 16, lanes = (unsigned) seg_ol[j] & 0xffffu;
               if ((unsigned) lane < lanes)
               {
                  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *) (p.x + (size_t) v * xstride + (size_t) (unsigned) seg_start[j] + 2 * lane),
                                                   (__attribute__((address_space(3))) void *) (reinterpret_cast<char *>(xsv + (size_t) v * stage_elems) + offb), 16, 0, 0);
               }
            }
         }
      }
   }
   // the first row's operands of every lane: same trip
   double bv[NV];
#pragma unroll
   for (int v = 0; v < NV; v++) { bv[v] = 0.0; }
   if (p.beta != 0.0)
   {
      const int rb = max(r0 + min(tid, nrows - 1), 0);
#pragma unroll
      for (int v = 0; v < NV; v++) { bv[v] = p.b[(size_t) v * bstride + rb]; }
   }
   if (streamer)
   {
      // the spill of the tile's last row past the window (rows of at most SPMV_THREADS entries: the launcher's condition) and
      // the row pointers
      const int kC = ka + 8 * SPMV_THREADS + tid;
      const int qC = kC < k1 ? kC : min(ka + 8 * SPMV_THREADS, p.last_quad);
      const unsigned lC = lidx[qC];
      unsigned cdC = 0;
      double vC = 0.0;
      if (CODED) { cdC = p.Ac8[qC]; } else { vC = p.Aa[qC]; }
      constexpr int RPJ = (RP_CAP + SPMV_THREADS) / SPMV_THREADS;
      const int lim = min(nrows, rp_cap);
      int rpv[RPJ];
#pragma unroll
      for (int j = 0; j < RPJ; j++) { rpv[j] = p.Ai[r0 + min(tid + j * SPMV_THREADS, lim)]; }
#pragma unroll
      for (int j = 0; j < RPJ; j++) { rp[min(tid + j * SPMV_THREADS, lim)] = rpv[j]; }
      if (tid == 0 && rpv[0] != k0) { __hip_atomic_fetch_or(p.stale, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
      // park the window
      if (CODED)
      {
         *reinterpret_cast<unsigned *>(cdS + (kA - ka)) = cdA;
         *reinterpret_cast<unsigned *>(cdS + (kB - ka)) = cdB;
         if (kC < k1) { cdS[kC - ka] = (unsigned char) cdC; }
      }
      else
      {
         *reinterpret_cast<v2d *>(valS + (kA - ka))     = vA01;
         *reinterpret_cast<v2d *>(valS + (kA - ka) + 2) = vA23;
         *reinterpret_cast<v2d *>(valS + (kB - ka))     = vB01;
         *reinterpret_cast<v2d *>(valS + (kB - ka) + 2) = vB23;
         if (kC < k1) { valS[kC - ka] = vC; }
      }
      *reinterpret_cast<v2i *>(liS + (kA - ka)) = lA;
      *reinterpret_cast<v2i *>(liS + (kB - ka)) = lB;
      if (kC < k1) { liS[kC - ka] = (unsigned short) lC; }
   }
   __syncthreads();
   if (CODED && streamer)
   {
      if (__double_as_longlong(dictl[ckc]) != __double_as_longlong(ckv)) { __hip_atomic_fetch_or(p.stale, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
   }

   const int avg = (k1 - k0) / nrows;
   int Wd = 1;
   if (avg > 12)
   {
      Wd = p.reduce_w;
      if (Wd <= 0) { Wd = 32; while (Wd > 1 && nrows * Wd > SPMV_THREADS) { Wd >>= 1; } }
   }
#define MV_SUMS(WW) mv_row_sums<NV, CODED, WW>(p, r0, nrows, ka, rp_cap, staged, stage_elems, xsv, valS, cdS, dictl, liS, rp, xstride, bstride, ystride, bv)
   switch (Wd)
   {
      case 1:  MV_SUMS(1); break;
      case 2:  MV_SUMS(2); break;
      case 4:  MV_SUMS(4); break;
      case 8:  MV_SUMS(8); break;
      case 16: MV_SUMS(16); break;
      default: MV_SUMS(32); break;
   }
#undef MV_SUMS
}

// ---- plan construction: per tile, the segments of x its entries touch and per entry the index of its column in
// their concatenation.  One workgroup per tile.  Columns are handled in units of two (16 bytes of x): sort the units
// (bitonic, LDS), keep the distinct ones, cut where the gap to the next unit exceeds a threshold T — the smallest power
// of two that leaves at most 32 segments — and give up on the tile if the segments hold more than the staging capacity.
constexpr int XS_SORT  = 8192;              // >= SPMV_TILE + SPMV_MAXROW entries of a tile, power of two (a row-slice block: up to 8192)
constexpr int XS_SORT_BITS = 13;
constexpr int XS_UNITS = XS_CAP / 2;        // distinct 2-column units a staged tile can hold at most
__global__ __launch_bounds__(SPMV_THREADS)
void build_xs_kernel(const int *__restrict__ Aj, const int *__restrict__ tile_k, int num_tiles,
                     int *__restrict__ xs_cnt, int *__restrict__ xs_desc, unsigned short *__restrict__ lidx)
{
   __shared__ int key[XS_SORT];
   __shared__ int uniq[XS_UNITS + 1];       // distinct units, ascending
   __shared__ int pos[XS_UNITS + 1];        // staged position (in units) of every distinct unit
   __shared__ int part[SPMV_THREADS];
   const int tile = blockIdx.x, tid = threadIdx.x;
   const int k0 = tile_k[tile], k1 = tile_k[tile + 1];
   const int m = k1 - k0;
   int *desc = xs_desc + (size_t) tile * XS_DESC;
   for (int i = tid; i < XS_DESC; i += SPMV_THREADS) { desc[i] = 0; }
   if (m <= 0 || m > XS_SORT) { if (tid == 0) { xs_cnt[tile] = 0; } return; }
   // Distinct units first (a hash set in LDS: a tile's 2048 - 3000 entries name 700 - 1400 units), then a sort of those
   // only: the bitonic network over all entries took 140 us a tile, 85 ms for the 24 matrices of the benchmark hierarchy.
   constexpr int EMPTY = 0x7fffffff;
   for (int i = tid; i < XS_SORT; i += SPMV_THREADS) { key[i] = EMPTY; }
   __syncthreads();
   for (int i = tid; i < m; i += SPMV_THREADS)
   {
      const int q = Aj[k0 + i] >> 1;
      unsigned h = ((unsigned) q * 2654435761u) >> (32 - XS_SORT_BITS);
      while (true)
      {
         const int old = atomicCAS(&key[h & (XS_SORT - 1)], EMPTY, q);
         if (old == EMPTY || old == q) { break; }
         h++;
      }
   }
   __syncthreads();
   // block-wide exclusive scan of one value per lane (part[] in, part[] out inclusive; returns the total)
   auto scan = [&](int mine) -> int
   {
      part[tid] = mine;
      __syncthreads();
      for (int off = 1; off < SPMV_THREADS; off <<= 1)
      {
         const int v = tid >= off ? part[tid - off] : 0;
         __syncthreads();
         part[tid] += v;
         __syncthreads();
      }
      return part[SPMV_THREADS - 1];
   };
   // the set, packed (every lane looks at 16 consecutive slots) ...
   constexpr int PER = XS_SORT / SPMV_THREADS;
   int heads = 0;
   for (int i = tid * PER; i < (tid + 1) * PER; i++) { if (key[i] != EMPTY) { heads++; } }
   const int U = scan(heads);
   if (U > XS_UNITS) { if (tid == 0) { xs_cnt[tile] = 0; } return; }
   {
      int r = part[tid] - heads;
      for (int i = tid * PER; i < (tid + 1) * PER; i++) { if (key[i] != EMPTY) { uniq[r++] = key[i]; } }
   }
   int P2 = 2;
   while (P2 < U) { P2 <<= 1; }
   for (int i = U + tid; i < P2; i += SPMV_THREADS) { uniq[i] = EMPTY; }
   __syncthreads();
   // ... and put in ascending order
   for (int k = 2; k <= P2; k <<= 1)
   {
      for (int j = k >> 1; j > 0; j >>= 1)
      {
         for (int i = tid; i < P2; i += SPMV_THREADS)
         {
            const int l = i ^ j;
            if (l > i)
            {
               const int a = uniq[i], b = uniq[l];
               const bool up = (i & k) == 0;
               if ((a > b) == up) { uniq[i] = b; uniq[l] = a; }
            }
         }
         __syncthreads();
      }
   }
   // Cut threshold T in {0, 1, 2, 4, ...}: a gap of more than T units between two distinct units ends a segment; a
   // segment is staged in pieces of at most XS_PIECE units.  The smallest T whose pieces fit the descriptor table wins
   // (the covered length grows with T; once it exceeds the staging capacity the tile is given up).
   constexpr int UPER = (XS_UNITS + SPMV_THREADS - 1) / SPMV_THREADS;
   constexpr int XS_PIECE = 64;                 // units (2 doubles) per piece: one 16-byte load per lane of a wave
   int T = 0, npieces = 0, covered_units = 0;
   bool fits = false;
   // Where to start: a gap g is cut at thresholds 0, 1, 2, 4, ... below g; a histogram of the gaps by the last threshold
   // that cuts them gives, for every threshold at once, the number of segments and the covered length.  A threshold that
   // leaves more segments than descriptors, or more 64-unit pieces' worth of length, cannot fit: the exact evaluation
   // below starts at the first one that can (level 1 of the benchmark hierarchy: at 8 instead of 0), and a tile no
   // threshold can serve (most tiles of a restriction operator) is given up here.
   __shared__ int hcnt[24], hsum[24];
   if (tid < 24) { hcnt[tid] = 0; hsum[tid] = 0; }
   __syncthreads();
   for (int i = tid * UPER; i < (tid + 1) * UPER && i < U - 1; i++)
   {
      const int g = uniq[i + 1] - uniq[i] - 1;
      if (g > 0)
      {
         const int last = g >= 2 ? min(32 - __clz(g - 1), 23) : 0;      // cut at thresholds 0 .. last
         atomicAdd(&hcnt[last], 1); atomicAdd(&hsum[last], g);
      }
   }
   __syncthreads();
   int first = 24;
   {
      const int span = uniq[U - 1] - uniq[0] + 1;
      int ncut = 0, cut = 0;
      for (int k = 23; k >= 0; k--)
      {
         ncut += hcnt[k]; cut += hsum[k];
         const int covered = span - cut, nseg = ncut + 1;
         if (covered <= XS_UNITS && max(nseg, (covered + XS_PIECE - 1) / XS_PIECE) <= XS_SEGS) { first = k; }
      }
   }
   if (first >= 24) { if (tid == 0) { xs_cnt[tile] = 0; } return; }
   T = first == 0 ? 0 : 1 << (first - 1);
   for (int step = first; step < 24; step++)
   {
      // staged position of unit i = uniq[i] - uniq[0] - (sum of the cut gaps before it); a lane owns UPER consecutive units
      int cut = 0;
      for (int i = tid * UPER; i < (tid + 1) * UPER && i < U - 1; i++) { const int g = uniq[i + 1] - uniq[i] - 1; if (g > T) { cut += g; } }
      const int cut_total = scan(cut);
      const int covered = uniq[U - 1] - uniq[0] + 1 - cut_total;
      {
         int before = part[tid] - cut;           // cut gaps in front of this lane's first unit
         for (int i = tid * UPER; i < (tid + 1) * UPER && i < U; i++)
         {
            pos[i] = uniq[i] - uniq[0] - before;
            if (i < U - 1) { const int g = uniq[i + 1] - uniq[i] - 1; if (g > T) { before += g; } }
         }
      }
      __syncthreads();
      if (covered > XS_UNITS) { break; }
      covered_units = covered;
      int np = 0;
      for (int i = tid * UPER; i < (tid + 1) * UPER && i < U; i++)
      {
         if (i == 0 || uniq[i] - uniq[i - 1] - 1 > T)
         {
            int e = i + 1;
            while (e < U && uniq[e] - uniq[e - 1] - 1 <= T) { e++; }
            np += (pos[e - 1] - pos[i] + XS_PIECE) / XS_PIECE;      // ceil(length / XS_PIECE)
         }
      }
      npieces = scan(np);
      __syncthreads();
      if (npieces <= XS_SEGS) { fits = true; break; }
      T = T == 0 ? 1 : 2 * T;
   }
   if (!fits) { if (tid == 0) { xs_cnt[tile] = 0; } return; }
   // descriptors: part[] still holds the inclusive scan of the pieces per lane
   {
      int np = 0;
      for (int i = tid * UPER; i < (tid + 1) * UPER && i < U; i++)
      {
         if (i == 0 || uniq[i] - uniq[i - 1] - 1 > T)
         {
            int e = i + 1;
            while (e < U && uniq[e] - uniq[e - 1] - 1 <= T) { e++; }
            np += (pos[e - 1] - pos[i] + XS_PIECE) / XS_PIECE;
         }
      }
      int pidx = part[tid] - np;
      for (int i = tid * UPER; i < (tid + 1) * UPER && i < U; i++)
      {
         if (i == 0 || uniq[i] - uniq[i - 1] - 1 > T)
         {
            int e = i + 1;
            while (e < U && uniq[e] - uniq[e - 1] - 1 <= T) { e++; }
            const int len_units = pos[e - 1] - pos[i] + 1;
            for (int o = 0; o < len_units; o += XS_PIECE)
            {
               const int slot = (pidx & 3) * XS_WSEG + (pidx >> 2);  // pieces dealt to the four waves, a wave's own contiguous
               desc[slot] = 2 * (uniq[i] + o);
               desc[XS_SEGS + slot] = ((16 * (pos[i] + o)) << 16) | min(XS_PIECE, len_units - o);     // LDS byte offset | lanes
               pidx++;
            }
         }
      }
   }
   const int nseg = npieces;
   if (tid == 0) { xs_cnt[tile] = (covered_units << 8) | nseg; }
   for (int i = tid; i < m; i += SPMV_THREADS)
   {
      const int col = Aj[k0 + i], q = col >> 1;
      int lo = 0, hi = U - 1;
      while (lo < hi) { const int mid = (lo + hi) >> 1; if (uniq[mid] < q) { lo = mid + 1; } else { hi = mid; } }
      lidx[k0 + i] = (unsigned short) (2 * pos[lo] + (col & 1));
   }
}

void launch_build_xs(const HYPRE_Int *Aj, const int *d_tile_k, int num_tiles, int *xs_cnt, int *xs_desc, unsigned short *lidx,
                     hipStream_t s)
{
   if (num_tiles > 0)
   {
      hipLaunchKernelGGL(build_xs_kernel, dim3(num_tiles), dim3(SPMV_THREADS), 0, s, Aj, d_tile_k, num_tiles, xs_cnt, xs_desc, lidx);
   }
}

// ---------------------------------------------------------------------------
// Slice form of coded short-row matrices (SpmvPlan::d_sl_data).
//
// With the values coded, a tile of the fine-level operators is bound by its own instruction stream (DESIGN.md section 4,
// round 3): half of it parks products in LDS and sums them up again.  A stencil's rows are short and equally long, so a
// lane can own a row: the codes and local indices of the 64 lane-rows of a wave lie side by side, entry j of every
// lane-row in one word position (a sliced ELL layout of what the plan already keeps privately: one byte and two bytes
// per entry), a workgroup takes 256 / W consecutive rows (W = 1: rows of at most 8 entries; W = 2: at most 32, a lane
// takes every second entry), stages their x pieces exactly as spmv_xs_kernel stages a tile's, and every lane multiplies
// and adds its entries in stored order out of registers: one barrier, no products in LDS, no reduction.  Summation:
// W = 1 is the stored order of the row (what spmv_xs_kernel does for rows of at most 12 entries: same bits; and what the
// reference's host loop does); W = 2 is two interleaved partial sums added at the end (spmv_xs_kernel's two-lane path,
// which it takes for tiles of 65 to 128 rows).
// The kernel reads neither the column array nor the values: the staleness watch samples both per block.
// ---------------------------------------------------------------------------
// KP, the entries a lane holds: 8 or 16 (a compile-time count: every load of the stream is then unconditional and the
// sum a straight line; with run-time word counts the compiler put a wait behind every load)
template <int OP, int W, int KP>
__global__ __launch_bounds__(SPMV_THREADS)
void spmv_sl_kernel(SpmvArgs p, const int *__restrict__ sl_cnt, const int *__restrict__ sl_desc, const int *__restrict__ sl_k0,
                    const int *__restrict__ sl_fp, const int *__restrict__ sl_perm, const unsigned *__restrict__ sl_data,
                    int blocks, int num_rows, int nnz, int stage_elems)
{
   constexpr int wc = KP / 4, wl = KP / 2;
   extern __shared__ __align__(16) unsigned char smem_raw[];
   double *xs = reinterpret_cast<double *>(smem_raw);
   const double *dictl = xs + stage_elems;

   int block = (int) blockIdx.x;
   if (sl_perm)
   {
      if (block >= blocks) { return; }
      block = sl_perm[block];
   }
   else { const int g = block >> 3, c = block & 7; block = (g >> 3) * 64 + c * 8 + (g & 7); }     // runs of 8 blocks per XCD
   if ((unsigned) block >= (unsigned) blocks) { return; }

   const int tid = threadIdx.x, lane = tid & 63;
   const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
   constexpr int R = SPMV_THREADS / W;
   // this lane's words of the slice: one trip, issued before anything about the block is known
   const unsigned *sd = sl_data + ((size_t) (block * 4 + wave) * (size_t) (wc + wl)) * 64 + lane;
   unsigned cw[wc], lw[wl];
#pragma unroll
   for (int w = 0; w < wc; w++) { cw[w] = sd[w * 64]; }
#pragma unroll
   for (int w = 0; w < wl; w++) { lw[w] = sd[(wc + w) * 64]; }
   const int r = block * R + tid / W, sub = tid % W;
   const int rs = p.Ai[min(r, num_rows)], re = p.Ai[min(r + 1, num_rows)];
   const RowOps ops = load_row_ops<OP>(p, max(min(r, num_rows - 1), 0));
   // wave-uniform, through the scalar cache: this wave's piece descriptors, the block's first entry and fingerprint
   const int *dsc = sl_desc + (size_t) block * XS_DESC + XS_WSEG * wave;
   int seg_start[XS_WSEG], seg_ol[XS_WSEG];
#pragma unroll
   for (int j = 0; j < XS_WSEG; j++) { seg_start[j] = dsc[j]; seg_ol[j] = dsc[XS_SEGS + j]; }
   const int k0 = sl_k0[block], k1b = sl_k0[block + 1], fp_plan = sl_fp[block];
   asm volatile("" :: "s"(sl_desc), "s"(sl_k0), "s"(sl_fp), "s"(sl_data), "s"(p.Ai) : "memory");
#pragma unroll
   for (int j = 0; j < XS_WSEG; j++)
   {
      const unsigned offb = (unsigned) seg_ol[j] >> 16, lanes = (unsigned) seg_ol[j] & 0xffffu;
      if ((unsigned) lane < lanes)
      {
         __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *) (p.x + (size_t) (unsigned) seg_start[j] + 2 * lane),
                                          (__attribute__((address_space(3))) void *) (reinterpret_cast<char *>(xs) + offb), 16, 0, 0);
      }
   }
   {
      const int dl = (p.ndict + 1) >> 1;
      if (lane < dl - 64 * wave)
      {
         __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *) (p.dict + 128 * wave + 2 * lane),
                                          (__attribute__((address_space(3))) void *) (reinterpret_cast<char *>(const_cast<double *>(dictl)) + 1024 * wave), 16, 0, 0);
      }
   }
   // the watch's samples: two columns and one value of the block's first entries (a second scalar trip, back long before
   // the vector loads above)
   const int q0 = min(max(k0, 0), nnz - 1), q1 = min(max(k0, 0) + 1, nnz - 1);
   const int fc0 = p.Aj[q0], fc1 = p.Aj[q1];
   // the rotating value check (see spmv_xs_kernel): eight consecutive entries of the block per wave, through the code array
   // the slices were packed from; a block holds at most 4096 entries, so four waves cover it within 128 launches
   const int ck = min(max(min(k0 + 8 * (int) ((4u * p.rot + (unsigned) wave) & 511u) + (lane & 7), k1b - 1), 0), nnz - 1);
   const double ckv = p.Aa[ck];
   const unsigned ckc = p.Ac8[ck];
   __syncthreads();

   const int mylen = (re - rs - sub + W - 1) / W;
   double sum = 0.0;
#pragma unroll
   for (int j = 0; j < KP; j++)
   {
      const unsigned code = (cw[j >> 2] >> (8 * (j & 3))) & 0xffu;
      const unsigned li = (lw[j >> 1] >> (16 * (j & 1))) & 0xffffu;
      {
#pragma clang fp contract(off)                                  // no fused multiply-add: the tiled kernel rounds every product
         const double pr = dictl[code] * xs[li];
         if (j < mylen) { sum = sum + pr; }
      }
   }
   if (W == 2) { sum += __shfl_xor(sum, 1, 64); }
   {
      bool off = __double_as_longlong(dictl[ckc]) != __double_as_longlong(p.dict_rounded ? (double) (float) ckv : ckv);
      if (tid == 0) { off = off || fp_plan != (int) ((unsigned) fc0 * 2654435761u + (unsigned) fc1) || rs != k0; }
      if (off) { __hip_atomic_fetch_or(p.stale, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
   }
   if (r < num_rows && sub == 0) { row_epilogue<OP>(p, r, sum, ops); }
}

// The slice form with a multivector: a lane's codes and local indices are in registers, so the columns cost their x pieces
// (NV staged copies), NV reads per entry and NV sums — the matrix words are read once.  Same bits as spmv_sl_kernel column
// by column (and so as every other form).  y = alpha A x + beta b only.
template <int NV, int W, int KP>
__global__ __launch_bounds__(SPMV_THREADS)
void spmv_sl_mv_kernel(SpmvArgs p, const int *__restrict__ sl_desc, const int *__restrict__ sl_k0,
                       const int *__restrict__ sl_fp, const int *__restrict__ sl_perm, const unsigned *__restrict__ sl_data,
                       int blocks, int num_rows, int nnz, int stage_elems, long xstride, long bstride, long ystride)
{
   constexpr int wc = KP / 4, wl = KP / 2;
   extern __shared__ __align__(16) unsigned char smem_raw[];
   double *xs = reinterpret_cast<double *>(smem_raw);
   const double *dictl = xs + (size_t) NV * stage_elems;

   int block = (int) blockIdx.x;
   if (sl_perm)
   {
      if (block >= blocks) { return; }
      block = sl_perm[block];
   }
   else { const int g = block >> 3, c = block & 7; block = (g >> 3) * 64 + c * 8 + (g & 7); }
   if ((unsigned) block >= (unsigned) blocks) { return; }

   const int tid = threadIdx.x, lane = tid & 63;
   const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
   constexpr int R = SPMV_THREADS / W;
   const unsigned *sd = sl_data + ((size_t) (block * 4 + wave) * (size_t) (wc + wl)) * 64 + lane;
   unsigned cw[wc], lw[wl];
#pragma unroll
   for (int w = 0; w < wc; w++) { cw[w] = sd[w * 64]; }
#pragma unroll
   for (int w = 0; w < wl; w++) { lw[w] = sd[(wc + w) * 64]; }
   const int r = block * R + tid / W, sub = tid % W;
   const int rs = p.Ai[min(r, num_rows)], re = p.Ai[min(r + 1, num_rows)];
   const int rc = max(min(r, num_rows - 1), 0);
   double bv[NV];
#pragma unroll
   for (int v = 0; v < NV; v++) { bv[v] = 0.0; }
   if (p.beta != 0.0)
   {
#pragma unroll
      for (int v = 0; v < NV; v++) { bv[v] = p.b[(size_t) v * bstride + rc]; }
   }
   const int *dsc = sl_desc + (size_t) block * XS_DESC + XS_WSEG * wave;
   int seg_start[XS_WSEG], seg_ol[XS_WSEG];
#pragma unroll
   for (int j = 0; j < XS_WSEG; j++) { seg_start[j] = dsc[j]; seg_ol[j] = dsc[XS_SEGS + j]; }
   const int k0 = sl_k0[block], k1b = sl_k0[block + 1], fp_plan = sl_fp[block];
   asm volatile("" :: "s"(sl_desc), "s"(sl_k0), "s"(sl_fp), "s"(sl_data), "s"(p.Ai) : "memory");
#pragma unroll
   for (int v = 0; v < NV; v++)
   {
#pragma unroll
      for (int j = 0; j < XS_WSEG; j++)
      {
         const unsigned offb = (unsigned) seg_ol[j] >> 16, lanes = (unsigned) seg_ol[j] & 0xffffu;
         if ((unsigned) lane < lanes)
         {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *) (p.x + (size_t) v * xstride + (size_t) (unsigned) seg_start[j] + 2 * lane),
                                             (__attribute__((address_space(3))) void *) (reinterpret_cast<char *>(xs + (size_t) v * stage_elems) + offb), 16, 0, 0);
         }
      }
   }
   {
      const int dl = (p.ndict + 1) >> 1;
      if (lane < dl - 64 * wave)
      {
         __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *) (p.dict + 128 * wave + 2 * lane),
                                          (__attribute__((address_space(3))) void *) (reinterpret_cast<char *>(const_cast<double *>(dictl)) + 1024 * wave), 16, 0, 0);
      }
   }
   // the watch's samples and the rotating value check, as spmv_sl_kernel
   const int q0 = min(max(k0, 0), nnz - 1), q1 = min(max(k0, 0) + 1, nnz - 1);
   const int fc0 = p.Aj[q0], fc1 = p.Aj[q1];
   const int ck = min(max(min(k0 + 8 * (int) ((4u * p.rot + (unsigned) wave) & 511u) + (lane & 7), k1b - 1), 0), nnz - 1);
   const double ckv = p.Aa[ck];
   const unsigned ckc = p.Ac8[ck];
   __syncthreads();

   const int mylen = (re - rs - sub + W - 1) / W;
   double sum[NV];
#pragma unroll
   for (int v = 0; v < NV; v++) { sum[v] = 0.0; }
#pragma unroll
   for (int j = 0; j < KP; j++)
   {
      const unsigned code = (cw[j >> 2] >> (8 * (j & 3))) & 0xffu;
      const unsigned li = (lw[j >> 1] >> (16 * (j & 1))) & 0xffffu;
      const double a = dictl[code];
#pragma unroll
      for (int v = 0; v < NV; v++)
      {
#pragma clang fp contract(off)
         const double pr = a * xs[v * stage_elems + li];
         if (j < mylen) { sum[v] = sum[v] + pr; }
      }
   }
   if (W == 2)
   {
#pragma unroll
      for (int v = 0; v < NV; v++) { sum[v] += __shfl_xor(sum[v], 1, 64); }
   }
   {
      bool off = __double_as_longlong(dictl[ckc]) != __double_as_longlong(ckv);
      if (tid == 0) { off = off || fp_plan != (int) ((unsigned) fc0 * 2654435761u + (unsigned) fc1) || rs != k0; }
      if (off) { __hip_atomic_fetch_or(p.stale, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
   }
   if (r < num_rows && sub == 0)
   {
#pragma unroll
      for (int v = 0; v < NV; v++)
      {
         double t = __dmul_rn(p.alpha, sum[v]);
         if (p.beta != 0.0) { t = __fma_rn(p.beta, bv[v], t); }
         p.y[(size_t) v * ystride + r] = t;
      }
   }
}

// ---- slice form construction
__global__ void sl_block_starts_kernel(const int *__restrict__ Ai, int num_rows, int R, int blocks, int *__restrict__ k0)
{
   const int b = blockIdx.x * blockDim.x + threadIdx.x;
   if (b <= blocks) { k0[b] = Ai[min((long long) b * R, (long long) num_rows)]; }
}
__global__ void sl_fp_kernel(const int *__restrict__ Aj, const int *__restrict__ k0, int blocks, int nnz, int *__restrict__ fp)
{
   const int b = blockIdx.x * blockDim.x + threadIdx.x;
   if (b >= blocks) { return; }
   const int q0 = min(max(k0[b], 0), nnz - 1), q1 = min(max(k0[b], 0) + 1, nnz - 1);
   fp[b] = (int) ((unsigned) Aj[q0] * 2654435761u + (unsigned) Aj[q1]);
}
template <int W>
__global__ __launch_bounds__(SPMV_THREADS)
void sl_pack_kernel(const int *__restrict__ Ai, const unsigned char *__restrict__ codes, const unsigned short *__restrict__ lidx,
                    int num_rows, int kper, int wc, int wl, unsigned *__restrict__ data)
{
   constexpr int R = SPMV_THREADS / W;
   const int block = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
   const int r = block * R + tid / W, sub = tid % W;
   const int rs = r < num_rows ? Ai[r] : 0, re = r < num_rows ? Ai[r + 1] : 0;
   unsigned *out = data + ((size_t) (block * 4 + wave) * (size_t) (wc + wl)) * 64 + lane;
   for (int w = 0; w < wc; w++)
   {
      unsigned word = 0;
      for (int q = 0; q < 4; q++)
      {
         const int j = 4 * w + q, k = rs + sub + j * W;
         if (j < kper && k < re) { word |= (unsigned) codes[k] << (8 * q); }
      }
      out[w * 64] = word;
   }
   for (int w = 0; w < wl; w++)
   {
      unsigned word = 0;
      for (int q = 0; q < 2; q++)
      {
         const int j = 2 * w + q, k = rs + sub + j * W;
         if (j < kper && k < re) { word |= (unsigned) lidx[k] << (16 * q); }
      }
      out[(wc + w) * 64] = word;
   }
}

bool device_build_slice_form(SpmvPlan *p, const hypre_CSRMatrix *A, hipStream_t s)
{
   const int K = p->max_row_nnz, n = A->num_rows, nnz = A->num_nonzeros;
   if (!p->d_codes || K < 1 || K > 32 || n < 1 || nnz < 1) { return false; }
   const int W = K <= 8 ? 1 : 2, R = SPMV_THREADS / W;                 // (a lane with 16 entries of its own needs 80 registers)
   const int kper = (K + W - 1) / W <= 8 ? 8 : 16, wc = kper / 4, wl = kper / 2;      // entries a lane holds: the kernel's KP
   const long long blocks_ll = ((long long) n + R - 1) / R;
   // rows about equally long and near 8, 16 or 32 entries: the padded form must not hold more than 1.3 x the entries
   if ((double) blocks_ll * R * W * kper > 1.3 * (double) nnz + 4096.0 || blocks_ll > (1LL << 28)) { return false; }
   const int blocks = (int) blocks_ll;
   int *d_k0 = nullptr, *d_cnt = nullptr, *d_desc = nullptr, *d_fp = nullptr, *d_perm = nullptr;
   unsigned short *d_li = nullptr;
   unsigned *d_data = nullptr;
   // every allocation is checked (PLAN_SITE_SLICE): on failure what was obtained is freed and the matrix keeps the coded tiles
   auto fail = [&]() -> bool
   {
      HIP_CHECK(hipStreamSynchronize(s));
      for (void *q : {(void *) d_k0, (void *) d_cnt, (void *) d_desc, (void *) d_fp, (void *) d_li, (void *) d_data, (void *) d_perm}) { plan_free(q); }
      return false;
   };
   const size_t nl = ((size_t) nnz + 15) & ~(size_t) 7;
   if (!plan_alloc((void **) &d_k0, sizeof(int) * ((size_t) blocks + 1), PLAN_SITE_SLICE) ||
       !plan_alloc((void **) &d_cnt, sizeof(int) * (size_t) blocks, PLAN_SITE_SLICE) ||
       !plan_alloc((void **) &d_desc, sizeof(int) * (size_t) blocks * XS_DESC, PLAN_SITE_SLICE) ||
       !plan_alloc((void **) &d_li, sizeof(unsigned short) * nl, PLAN_SITE_SLICE)) { return fail(); }
   hipLaunchKernelGGL(sl_block_starts_kernel, dim3((blocks + 256) / 256), dim3(256), 0, s, A->i, n, R, blocks, d_k0);
   HIP_CHECK(hipMemsetAsync(d_li, 0, sizeof(unsigned short) * nl, s));
   launch_build_xs(A->j, d_k0, blocks, d_cnt, d_desc, d_li, s);      // a block's entries are at most R * K <= 4096: what the builder sorts
   std::vector<int> cnt((size_t) blocks);
   HIP_CHECK(hipMemcpyAsync(cnt.data(), d_cnt, sizeof(int) * (size_t) blocks, hipMemcpyDeviceToHost, s));
   HIP_CHECK(hipStreamSynchronize(s));
   int units = 0;
   for (int c : cnt)
   {
      if ((c & 0xff) == 0) { return fail(); }            // a block that cannot be staged (or holds no entry): no slice form
      units = std::max(units, c >> 8);
   }
   // LDS: the staged copy and the value table; at least four workgroups per CU
   if (16 * (size_t) units + 8 * DICT_CAP > 40 * 1024) { return fail(); }
   const size_t words = (size_t) blocks * 4 * (size_t) (wc + wl) * 64;
   if (!plan_alloc((void **) &d_data, sizeof(unsigned) * words, PLAN_SITE_SLICE) ||
       !plan_alloc((void **) &d_fp, sizeof(int) * (size_t) blocks, PLAN_SITE_SLICE)) { return fail(); }
   if (W == 1) { hipLaunchKernelGGL((sl_pack_kernel<1>), dim3(blocks), dim3(SPMV_THREADS), 0, s, A->i, p->d_codes, d_li, n, kper, wc, wl, d_data); }
   else        { hipLaunchKernelGGL((sl_pack_kernel<2>), dim3(blocks), dim3(SPMV_THREADS), 0, s, A->i, p->d_codes, d_li, n, kper, wc, wl, d_data); }
   hipLaunchKernelGGL(sl_fp_kernel, dim3((blocks + 255) / 256), dim3(256), 0, s, A->j, d_k0, blocks, nnz, d_fp);
   // band-aware placement, as for the tiles (seq_mv.cpp: build_band_placement): XCD c takes the blocks of slab c
   // (speed only: without the table the blocks go to the XCDs in runs of 8)
   if (p->band > 0 && blocks >= 64 && plan_alloc((void **) &d_perm, sizeof(int) * (size_t) blocks, PLAN_SITE_SLICE))
   {
      const int B = p->band;
      std::vector<std::vector<int>> cls(8);
      for (int b = 0; b < blocks; b++)
      {
         const int c = (int) (((long long) (((long long) b * R) % B) * 8) / B);
         cls[(size_t) std::min(std::max(c, 0), 7)].push_back(b);
      }
      std::vector<int> perm((size_t) blocks, -1), leftovers;
      std::vector<size_t> next(8, 0);
      for (int g = 0; g < blocks; g++) { const size_t c = (size_t) (g & 7); if (next[c] < cls[c].size()) { perm[(size_t) g] = cls[c][next[c]++]; } }
      for (size_t c = 0; c < 8; c++) { for (size_t k = next[c]; k < cls[c].size(); k++) { leftovers.push_back(cls[c][k]); } }
      size_t lo = 0;
      for (int g = 0; g < blocks; g++) { if (perm[(size_t) g] < 0) { perm[(size_t) g] = leftovers[lo++]; } }
      HIP_CHECK(hipMemcpyAsync(d_perm, perm.data(), sizeof(int) * (size_t) blocks, hipMemcpyHostToDevice, s));
      HIP_CHECK(hipStreamSynchronize(s));
   }
   HIP_CHECK(hipStreamSynchronize(s));
   plan_free(d_li);
   p->sl_w = W; p->sl_rows = R; p->sl_k = kper; p->sl_wc = wc; p->sl_wl = wl;
   p->sl_blocks = blocks; p->sl_launch_units = units;
   p->d_sl_cnt = d_cnt; p->d_sl_desc = d_desc; p->d_sl_k0 = d_k0; p->d_sl_fp = d_fp; p->d_sl_perm = d_perm; p->d_sl_data = d_data;
   return true;
}

// ---------------------------------------------------------------------------
// Row-slice form of owned, uncoded matrices (SpmvPlan::d_rs_val): the coarse-level operators.
//
// What the tiled kernel costs on them (level 1 of the benchmark hierarchy, 29 entries per
// row, every value distinct): 550 instructions per wave and tile, half of them to park 2048 products in LDS and to sum them up
// again with a lane tree — the level runs at 0.65 of the HBM peak with 52 % of its LDS-active cycles lost to bank conflicts.
// The slice form of coded stencils showed the way out (a lane owns a row, sums from registers); these rows are neither
// short nor equally long, so the slices are JAGGED: a workgroup takes R = 256 / W consecutive rows with W lanes per row, lane
// `sub` of a row owns its entries sub, sub + W, ...; the 64 lane-tasks of a wave are sorted by their entry counts and entry c
// of every task that has one lies side by side — chunk c of a wave holds exactly as many entries as it has tasks longer than
// c.  No padding at all: 8 + 2 bytes per entry (fp64 value, 16-bit staged position of its column as a byte offset), no row
// pointers, no column array; what a wave needs to find its chunks — the position of its first entry and the number of
// active lanes per chunk — are 16 scalars.  Every lane multiplies and adds its entries in stored order out of registers
// (one LDS read and one fused multiply-add per entry), writes ONE partial sum to LDS, and the lane that finishes a row adds
// its W partial sums in lane order: no products parked, no tree, one barrier after the loads and one before the epilogue.
// x is staged exactly as in spmv_xs_kernel (same builder over the blocks' entries, same descriptors, same LDS-DMA trip).
// Only matrices that cannot change behind their plans get the form (it keeps a private copy of the values): no watch.
// Reference counterpart: seq_mv/csr_spmv_device.c:149-260 (K lanes per row from the matrix-wide mean row length, shuffle
// tree, x through the cache).
// ---------------------------------------------------------------------------
constexpr int RS_HDR = 16;          // ints per wave: [0] position of its first entry, [1] chunks, [4 + c / 4] byte c & 3: active lanes of chunk c
// (A persistent form — a workgroup walking blocks v, v + G, ... with the NEXT block's header and descriptors requested behind
// the current block's vector loads, so that no block waits for a cold scalar round trip — was built and measured: level 1
// of the benchmark hierarchy 0.377 ms against 0.332 ms, whatever the number of workgroups; the doubled scalars spill.  Removed.)
template <int OP, int KP, bool F32>
__global__ __launch_bounds__(SPMV_THREADS)
void spmv_rs_kernel(SpmvArgs p, const int *__restrict__ rs_desc, const int *__restrict__ rs_perm, const int *__restrict__ rs_hdr,
                    const unsigned *__restrict__ rs_meta, const double *__restrict__ rs_val, const float *__restrict__ rs_val32,
                    const unsigned *__restrict__ rs_idx, int blocks, int num_rows, int R, int W, int stage_elems)
{
   extern __shared__ __align__(16) unsigned char smem_raw[];
   double *xs = reinterpret_cast<double *>(smem_raw);
   double *part = xs + stage_elems;                 // [SPMV_THREADS] partial sums, lane-of-the-row major

   int block = (int) blockIdx.x;
   if (rs_perm)
   {
      if (block >= blocks) { return; }
      block = rs_perm[block];
   }
   else { const int g = block >> 3, c = block & 7; block = (g >> 3) * 64 + c * 8 + (g & 7); }     // runs of 8 blocks per XCD
   if ((unsigned) block >= (unsigned) blocks) { return; }

   const int tid = threadIdx.x, lane = tid & 63;
   const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
   // known before anything about the block is: this lane's entry count and slot, the operands of the row it will finish
   const unsigned meta = rs_meta[(size_t) block * SPMV_THREADS + tid];
   const int r = block * R + tid;
   const RowOps ops = load_row_ops<OP>(p, max(min(r, num_rows - 1), 0));
   // wave-uniform, through the scalar cache: the wave's header and its piece descriptors
   constexpr int NH = 4 + KP / 4;
   const int *hdr = rs_hdr + (size_t) (block * 4 + wave) * RS_HDR;
   int h[NH];
#pragma unroll
   for (int i = 0; i < NH; i++) { h[i] = hdr[i]; }
   const int *dsc = rs_desc + (size_t) block * XS_DESC + XS_WSEG * wave;
   int seg_start[XS_WSEG], seg_ol[XS_WSEG];
#pragma unroll
   for (int j = 0; j < XS_WSEG; j++) { seg_start[j] = dsc[j]; seg_ol[j] = dsc[XS_SEGS + j]; }
   asm volatile("" :: "s"(rs_desc), "s"(rs_hdr), "s"(rs_meta) : "memory");
   const int nch = h[1];
   // The stream, in groups of four chunks (a group the wave has none of is skipped: wave-uniform): chunk c of this wave
   // starts where chunk c - 1 ended; lanes past a chunk's last entry — and chunks past the wave's last one inside a group —
   // re-read the entry in front (same cache line, no traffic), so that every load of a group is unconditional.  The 16-bit
   // positions come in pairs, one 32-bit word per lane for chunks 2p and 2p + 1 (as many words as chunk 2p has lanes): no
   // 16-bit loads — those are widened on arrival, i.e. waited for where they are issued.
   const double *wv = rs_val + h[0];
   const float *wf = rs_val32 + h[0];
   const unsigned *wi = rs_idx + h[2];
   double   vv[KP];
   float    vf[KP];
   unsigned iw[KP / 2];                  // staged positions (byte offsets) of the lane's entries of chunks 2p (low half) and 2p + 1
   {
      int off = 0, offi = 0;
#pragma unroll
      for (int g = 0; g < KP / 4; g++)
      {
         if (4 * g < nch)
         {
#pragma unroll
            for (int c = 4 * g; c < 4 * g + 4; c++)
            {
               const int na = (h[4 + (c >> 2)] >> (8 * (c & 3))) & 0xff;
               const unsigned q = (unsigned) (off + min(lane, na - 1));
               if (F32) { vf[c] = wf[q]; } else { vv[c] = wv[q]; }
               if ((c & 1) == 0) { iw[c >> 1] = wi[(unsigned) (offi + min(lane, na - 1))]; offi += na; }
               off += na;
            }
         }
      }
   }
   // the x pieces, straight into LDS (see spmv_xs_kernel)
#pragma unroll
   for (int j = 0; j < XS_WSEG; j++)
   {
      const unsigned offb = (unsigned) seg_ol[j] >> 16, lanes = (unsigned) seg_ol[j] & 0xffffu;
      if ((unsigned) lane < lanes)
      {
         __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *) (p.x + (size_t) (unsigned) seg_start[j] + 2 * lane),
                                          (__attribute__((address_space(3))) void *) (reinterpret_cast<char *>(xs) + offb), 16, 0, 0);
      }
   }
   __syncthreads();

   // a lane's sum, four entries at a time: the LDS reads of a group in flight together, then the multiply-adds in stored
   // order; an entry the lane does not have leaves the sum as it is (a select on the result: its x may be anything)
   const int cnt = (int) (meta >> 10);          // (not byte-aligned on purpose: a byte would be folded into an SDWA compare against a
                                                // REGISTER, and the constants 1 ... KP - 1 would then live in 31 VGPRs)
   const char *xb = reinterpret_cast<const char *>(xs);
   double sum = 0.0;
   // (two at a time where the registers are tight: the sweeps' three row operands on top of 32 entries a lane would cost the
   // fifth wave per SIMD)
#ifndef RS_FG_SWEEPS
#define RS_FG_SWEEPS 2            // (experiment switch: 4 = the sweeps' LDS reads four at a time too)
#endif
   constexpr int FG = (KP >= 32 && !F32 && OP != OP_AXPBY) ? RS_FG_SWEEPS : 4;
#pragma unroll
   for (int g = 0; g < KP / FG; g++)
   {
      if (FG * g < nch)
      {
         double xv[FG];
#pragma unroll
         for (int j = 0; j < FG; j++)
         {
            const int c = FG * g + j;
            const unsigned pos = (c & 1) ? (iw[c >> 1] >> 16) : (iw[c >> 1] & 0xffffu);
            xv[j] = *reinterpret_cast<const double *>(xb + pos);
         }
#pragma unroll
         for (int j = 0; j < FG; j++)
         {
            const int c = FG * g + j;
            const double a = F32 ? (double) vf[c] : vv[c];
            const double t = __fma_rn(a, xv[j], sum);
            sum = (c < cnt) ? t : sum;
         }
      }
   }
   part[meta & 0x3ffu] = sum;
   lds_barrier();
   if (tid < R && r < num_rows)
   {
      double total = part[tid];
      for (int j = 1; j < W; j++) { total += part[j * R + tid]; }
      row_epilogue<OP>(p, r, total, ops);
   }
}

// ---- row-slice construction: one workgroup per block
__global__ __launch_bounds__(SPMV_THREADS)
void rs_pack_kernel(const int *__restrict__ Ai, const double *__restrict__ Aa, const unsigned short *__restrict__ lidx, int n, int R, int W,
                    int kp, int *__restrict__ hdr, unsigned *__restrict__ meta, double *__restrict__ val, unsigned *__restrict__ idx,
                    int *__restrict__ fail)
{
   const int block = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
   const int RW = R >> 2;                                   // rows of a wave (R >= 4)
   const int q = wave * RW + lane / W, sub = lane % W;      // this thread's task: row q of the block, lane sub of the row
   const int r = block * R + q;
   const int s = r < n ? Ai[r] : 0, e = r < n ? Ai[r + 1] : 0, len = e - s;
   int cnt = len > sub ? (len - sub + W - 1) / W : 0;
   if (cnt > kp) { atomicExch(fail, 1); cnt = kp; }         // (the builder chose kp from the longest row: cannot happen)
   // position of the task in its wave: entry counts descending, ties in task order
   int rank = 0, above = 0;
   for (int vv = kp; vv >= 0; vv--)
   {
      const unsigned long long m = __ballot(cnt == vv);
      if (cnt == vv) { rank = above + __popcll(m & ((1ull << lane) - 1ull)); }
      above += __popcll(m);
   }
   const int first = Ai[min(block * R + wave * RW, n)];     // position of the wave's first entry
   // position of the wave's first word of staged positions: a wave with T entries holds between T / 2 and T / 2 + 32 words
   // (one per lane and pair of chunks, as many as the even chunk has lanes), so waves 33 words apart beyond half their first
   // entry never overlap
   const int firsti = (first >> 1) + 33 * (block * 4 + wave);
   int *hw = hdr + (size_t) (block * 4 + wave) * RS_HDR;
   int off = 0, offi = 0, nch = 0, word = 0;
   for (int c = 0; c < kp; c += 2)
   {
      const int na0 = __popcll(__ballot(cnt > c)), na1 = __popcll(__ballot(cnt > c + 1));
      if (c < cnt)
      {
         const int k = s + sub + c * W;
         val[(size_t) first + off + rank] = Aa[k];
         unsigned w2 = 8u * lidx[k];
         if (c + 1 < cnt)
         {
            val[(size_t) first + off + na0 + rank] = Aa[k + W];
            w2 |= (8u * lidx[k + W]) << 16;
         }
         idx[(size_t) firsti + offi + rank] = w2;
      }
      if (na0 > 0) { nch = c + 1; }
      if (na1 > 0) { nch = c + 2; }
      word |= (na0 << (8 * (c & 3))) | (na1 << (8 * ((c + 1) & 3)));
      if ((c & 3) == 2 || c + 2 >= kp) { if (lane == 0) { hw[4 + (c >> 2)] = word; } word = 0; }
      off += na0 + na1;
      offi += na0;
   }
   if (lane == 0) { hw[0] = first; hw[1] = nch; hw[2] = firsti; hw[3] = 0; }
   meta[(size_t) block * SPMV_THREADS + wave * 64 + rank] = (unsigned) (sub * R + q) | ((unsigned) cnt << 10);
}

int &spmv_row_slices()
{
   static int mode = [] { const char *e = getenv("HYPRE_AMD_SPMV_ROW_SLICES"); return e ? atoi(e) : 1; }();
   return mode;
}

bool device_build_row_slices(SpmvPlan *p, const hypre_CSRMatrix *A, hipStream_t s)
{
   const int n = A->num_rows, nnz = A->num_nonzeros, K = p->max_row_nnz;
   if (n < 1 || nnz < 1 || K < 1) { return false; }
   static const int force_w = [] { const char *e = getenv("HYPRE_AMD_SPMV_RS_W"); return e ? atoi(e) : 0; }();
   const double mean = (double) nnz / (double) n;
   // Rows of a dozen entries or fewer stay with the tiles (hypre_amd_SpmvSetRowSlices(2) takes them too): there a lane of the
   // tiled kernel sums a row by itself as well, and its stream is on its way before the tile's scalars are back, whereas a
   // block's slices are found through its wave headers — measured on the fine-level restriction operator of the benchmark
   // hierarchy (11.7 entries a row): 0.147 ms from tiles, 0.18 - 0.21 ms from slices.
   if (mean < 13.0 && force_w <= 0 && spmv_row_slices() < 2) { return false; }
   // lanes per row: a lane holds at most 32 entries; a block's entries are what the staging builder sorts (8192 at most:
   // a mean of 6144 or less); the more rows a block holds the more of the x it stages is shared between them
   int W = 1;
   while (W < 64 && (K + W - 1) / W > 32) { W <<= 1; }
   while (W < 64 && mean * (SPMV_THREADS / W) > 6144.0) { W <<= 1; }
   if (force_w > 0) { W = force_w; }
   for (; W <= 64; W <<= 1)
   {
      const int R = SPMV_THREADS / W;
      const int per = (K + W - 1) / W;
      if (per > 32) { continue; }
      // blocks worth their overhead: a thousand entries or more on average
      if (mean * R < 1024.0 && force_w <= 0) { return false; }
      const int kp = per <= 8 ? 8 : (per <= 16 ? 16 : (per <= 24 ? 24 : 32));
      const long long blocks_ll = ((long long) n + R - 1) / R;
      if (blocks_ll > (1LL << 28)) { return false; }
      const int blocks = (int) blocks_ll;
      int *d_k0 = nullptr, *d_cnt = nullptr, *d_desc = nullptr, *d_hdr = nullptr, *d_perm = nullptr, *d_fail = nullptr;
      unsigned *d_meta = nullptr;
      unsigned short *d_li = nullptr;
      unsigned *d_idx = nullptr;
      double *d_val = nullptr;
      auto release = [&]()
      {
         HIP_CHECK(hipStreamSynchronize(s));
         for (void *q : {(void *) d_k0, (void *) d_cnt, (void *) d_desc, (void *) d_hdr, (void *) d_perm, (void *) d_fail, (void *) d_meta,
                         (void *) d_li, (void *) d_idx, (void *) d_val}) { plan_free(q); }
      };
      const size_t nl = ((size_t) nnz + 15) & ~(size_t) 7;
      if (!plan_alloc((void **) &d_k0, sizeof(int) * ((size_t) blocks + 1), PLAN_SITE_ROWSLICE) ||
          !plan_alloc((void **) &d_cnt, sizeof(int) * (size_t) blocks, PLAN_SITE_ROWSLICE) ||
          !plan_alloc((void **) &d_desc, sizeof(int) * (size_t) blocks * XS_DESC, PLAN_SITE_ROWSLICE) ||
          !plan_alloc((void **) &d_li, sizeof(unsigned short) * nl, PLAN_SITE_ROWSLICE)) { release(); return false; }
      hipLaunchKernelGGL(sl_block_starts_kernel, dim3((blocks + 256) / 256), dim3(256), 0, s, A->i, n, R, blocks, d_k0);
      HIP_CHECK(hipMemsetAsync(d_li, 0, sizeof(unsigned short) * nl, s));
      launch_build_xs(A->j, d_k0, blocks, d_cnt, d_desc, d_li, s);
      std::vector<int> cnt((size_t) blocks), k0((size_t) blocks + 1);
      HIP_CHECK(hipMemcpyAsync(cnt.data(), d_cnt, sizeof(int) * (size_t) blocks, hipMemcpyDeviceToHost, s));
      HIP_CHECK(hipMemcpyAsync(k0.data(), d_k0, sizeof(int) * ((size_t) blocks + 1), hipMemcpyDeviceToHost, s));
      HIP_CHECK(hipStreamSynchronize(s));
      int units = 0;
      bool staged = true;
      for (int b = 0; b < blocks; b++)
      {
         const int c = cnt[(size_t) b];
         if ((c & 0xff) == 0 && k0[(size_t) b + 1] > k0[(size_t) b]) { staged = false; break; }     // (a block without entries stages nothing)
         units = std::max(units, c >> 8);
      }
      // every block must be staged (the kernel has no gathering path), in at most 40 KB of LDS (four workgroups per CU)
      if (!staged || 16 * (size_t) units + 8 * SPMV_THREADS > 40 * 1024)
      {
         release();
         if (force_w > 0) { return false; }
         continue;                                  // shorter blocks: more lanes per row
      }
      if (!plan_alloc((void **) &d_hdr, sizeof(int) * (size_t) blocks * 4 * RS_HDR, PLAN_SITE_ROWSLICE) ||
          !plan_alloc((void **) &d_meta, sizeof(unsigned) * (size_t) blocks * SPMV_THREADS, PLAN_SITE_ROWSLICE) ||
          !plan_alloc((void **) &d_val, sizeof(double) * ((size_t) nnz + 64), PLAN_SITE_ROWSLICE) ||
          !plan_alloc((void **) &d_idx, sizeof(unsigned) * ((size_t) nnz / 2 + 33 * 4 * (size_t) blocks + 128), PLAN_SITE_ROWSLICE) ||
          !plan_alloc((void **) &d_fail, sizeof(int), PLAN_SITE_ROWSLICE)) { release(); return false; }
      HIP_CHECK(hipMemsetAsync(d_fail, 0, sizeof(int), s));
      hipLaunchKernelGGL(rs_pack_kernel, dim3(blocks), dim3(SPMV_THREADS), 0, s, A->i, A->data, d_li, n, R, W, kp, d_hdr, d_meta, d_val, d_idx, d_fail);
      int failed = 0;
      HIP_CHECK(hipMemcpyAsync(&failed, d_fail, sizeof(int), hipMemcpyDeviceToHost, s));
      // band-aware placement, as for the tiles (speed only)
      if (p->band > 0 && blocks >= 64 && plan_alloc((void **) &d_perm, sizeof(int) * (size_t) blocks, PLAN_SITE_ROWSLICE))
      {
         const int B = p->band;
         std::vector<std::vector<int>> cls(8);
         for (int b = 0; b < blocks; b++)
         {
            const int c = (int) (((long long) (((long long) b * R) % B) * 8) / B);
            cls[(size_t) std::min(std::max(c, 0), 7)].push_back(b);
         }
         std::vector<int> perm((size_t) blocks, -1), leftovers;
         std::vector<size_t> next(8, 0);
         for (int g = 0; g < blocks; g++) { const size_t c = (size_t) (g & 7); if (next[c] < cls[c].size()) { perm[(size_t) g] = cls[c][next[c]++]; } }
         for (size_t c = 0; c < 8; c++) { for (size_t k = next[c]; k < cls[c].size(); k++) { leftovers.push_back(cls[c][k]); } }
         size_t lo = 0;
         for (int g = 0; g < blocks; g++) { if (perm[(size_t) g] < 0) { perm[(size_t) g] = leftovers[lo++]; } }
         HIP_CHECK(hipMemcpyAsync(d_perm, perm.data(), sizeof(int) * (size_t) blocks, hipMemcpyHostToDevice, s));
      }
      HIP_CHECK(hipStreamSynchronize(s));
      if (failed) { release(); return false; }
      plan_free(d_li); plan_free(d_k0); plan_free(d_cnt); plan_free(d_fail);
      p->rs_w = W; p->rs_rows = R; p->rs_kp = kp; p->rs_blocks = blocks; p->rs_units = units;
      p->d_rs_desc = d_desc; p->d_rs_perm = d_perm; p->d_rs_hdr = d_hdr; p->d_rs_meta = d_meta; p->d_rs_val = d_val; p->d_rs_idx = d_idx;
      return true;
   }
   return false;
}

// ---------------------------------------------------------------------------
// Wave-per-row SpMV: fallback for matrices with rows longer than SPMV_MAXROW
// or misaligned arrays (never hit by the AMG hierarchies of the benchmark).
// ---------------------------------------------------------------------------
template <int OP, bool F32, bool HASFILL>
__global__ __launch_bounds__(SPMV_THREADS)
void spmv_wave_kernel(SpmvArgs p, int num_rows)
{
   const int lane   = threadIdx.x & 63;
   const int wave   = (blockIdx.x * SPMV_THREADS + threadIdx.x) >> 6;
   const int nwaves = (gridDim.x * SPMV_THREADS) >> 6;
   for (int row = wave; row < num_rows; row += nwaves)
   {
      const int s = p.Ai[row], e = p.Ai[row + 1];
      double sum = 0.0;
      for (int k = s + lane; k < e; k += 64)
      {
         const int c = p.Aj[k];
         double v = F32 ? (double) p.Aa32[k] : p.Aa[k];
         if (HASFILL) { if (!fill_keep(p.fill, row, c)) { v = 0.0; } }
         sum += v * p.x[c];
      }
      sum = wave_sum64(sum);
      if (lane == 0) { const RowOps o = load_row_ops<OP>(p, row); row_epilogue<OP>(p, row, sum, o); }
   }
}

// y[row] += alpha * (A x)[row] (./ d[row] when d is given, only rows with
// marker == marker_val when a marker is given) over the listed non-empty rows
// (offd blocks: seq_mv/csr_matvec.c:381-670 rownnz path); rownnz == nullptr
// means all rows.  8 lanes per row.
__global__ __launch_bounds__(SPMV_THREADS)
void spmv_rownnz_kernel(SpmvArgs p, const int *__restrict__ rownnz, int num_rownnz)
{
   const int g   = (blockIdx.x * SPMV_THREADS + threadIdx.x) >> 3;
   const int sub = threadIdx.x & 7;
   double sum = 0.0;
   int row = 0;
   if (g < num_rownnz)
   {
      row = rownnz ? rownnz[g] : g;
      const int s = p.Ai[row], e = p.Ai[row + 1];
      for (int k = s + sub; k < e; k += 8)
      {
         const double v = p.Aa32 ? (double) p.Aa32[k] : p.Aa[k];
         sum += v * p.x[p.Aj[k]];
      }
   }
   sum = subwave_sum<8>(sum);
   if (g < num_rownnz && sub == 0)
   {
      if (p.marker && p.marker[row] != p.marker_val) { return; }
      double upd = p.alpha * sum;
      if (p.d) { upd /= p.d[row]; }
      p.y[row] += upd;
   }
}

// ---------------------------------------------------------------------------
// plan construction
// ---------------------------------------------------------------------------
// tile_row[b] = first row r with Ai[r] >= b*TILE (a row belongs to the tile its
// first entry falls into); tile_k[b] = Ai[tile_row[b]].
__global__ void build_tiles_kernel(const int *__restrict__ Ai, int num_rows, int num_tiles,
                                   int *__restrict__ tile_row, int *__restrict__ tile_k)
{
   const int b = blockIdx.x * blockDim.x + threadIdx.x;
   if (b > num_tiles) { return; }
   if (b == num_tiles) { tile_row[b] = num_rows; tile_k[b] = Ai[num_rows]; return; }
   const long long target = (long long) b * SPMV_TILE;
   int lo = 0, hi = num_rows;
   while (lo < hi)
   {
      const int mid = (lo + hi) >> 1;
      if ((long long) Ai[mid] >= target) { hi = mid; } else { lo = mid + 1; }
   }
   tile_row[b] = lo;
   tile_k[b] = Ai[lo];
}

__global__ void max_row_nnz_kernel(const int *__restrict__ Ai, int num_rows, int *__restrict__ out)
{
   int m = 0;
   for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < num_rows; r += gridDim.x * blockDim.x)
   {
      m = max(m, Ai[r + 1] - Ai[r]);
   }
#pragma unroll
   for (int off = 32; off > 0; off >>= 1) { m = max(m, __shfl_xor(m, off, 64)); }
   if ((threadIdx.x & 63) == 0) { atomicMax(out, m); }
}

// per tile: two entries of the column array (the positions spmv_xs_kernel samples), mixed
__global__ void build_fp_kernel(const HYPRE_Int *__restrict__ Aj, int last_quad, int num_tiles, int *__restrict__ fp)
{
   const int t = blockIdx.x * blockDim.x + threadIdx.x;
   if (t >= num_tiles) { return; }
   const int ka = t * SPMV_TILE;
   const int c0 = Aj[min(ka + 5, last_quad)], c1 = Aj[min(ka + 1029, last_quad)];
   fp[t] = (int) ((unsigned) c0 * 2654435761u + (unsigned) c1);
}
void launch_build_fp(const HYPRE_Int *Aj, int nnz, int num_tiles, int *fp, hipStream_t s)
{
   if (num_tiles <= 0) { return; }
   hipLaunchKernelGGL(build_fp_kernel, dim3((num_tiles + 255) / 256), dim3(256), 0, s, Aj, (nnz > 0 ? nnz - 1 : 0) & ~3, num_tiles, fp);
}

void launch_build_tiles(const HYPRE_Int *Ai, int num_rows, int nnz, int num_tiles, int *d_tile_row,
                        int *d_tile_k, hipStream_t s)
{
   (void) nnz;
   const int n = num_tiles + 1;
   hipLaunchKernelGGL(build_tiles_kernel, dim3((n + 255) / 256), dim3(256), 0, s, Ai, num_rows,
                      num_tiles, d_tile_row, d_tile_k);
}

__global__ void sample_row_bands_kernel(const HYPRE_Int *__restrict__ Ai, const HYPRE_Int *__restrict__ Aj, int num_rows,
                                        int nsamples, int *__restrict__ out)
{
   const int t = blockIdx.x * blockDim.x + threadIdx.x;
   if (t >= nsamples) { return; }
   const int row = (int) ((long long) num_rows * t / nsamples);
   int far = 0;
   for (int k = Ai[row]; k < Ai[row + 1]; k++) { far = max(far, abs(Aj[k] - row)); }
   out[t] = far;
}

void sample_row_bands(const HYPRE_Int *Ai, const HYPRE_Int *Aj, int num_rows, int nsamples, int *host_out, hipStream_t s)
{
   int *d_out = reinterpret_cast<int *>(reduce_scratch(((size_t) nsamples + 1) / 2));
   hipLaunchKernelGGL(sample_row_bands_kernel, dim3((nsamples + 255) / 256), dim3(256), 0, s, Ai, Aj, num_rows, nsamples, d_out);
   HIP_CHECK(hipMemcpyAsync(host_out, d_out, sizeof(int) * (size_t) nsamples, hipMemcpyDeviceToHost, s));
   HIP_CHECK(hipStreamSynchronize(s));
}

int device_max_row_nnz(const HYPRE_Int *Ai, int num_rows, hipStream_t s)
{
   if (num_rows <= 0) { return 0; }
   int *d_out = reinterpret_cast<int *>(reduce_scratch(2));
   HIP_CHECK(hipMemsetAsync(d_out, 0, sizeof(int), s));
   int grid = (num_rows + 255) / 256;
   if (grid > 2048) { grid = 2048; }
   hipLaunchKernelGGL(max_row_nnz_kernel, dim3(grid), dim3(256), 0, s, Ai, num_rows, d_out);
   int h = 0;
   HIP_CHECK(hipMemcpyAsync(&h, d_out, sizeof(int), hipMemcpyDeviceToHost, s));
   HIP_CHECK(hipStreamSynchronize(s));
   return h;
}

// ---- checksum of a CSR matrix's arrays (SpmvPlan::checksum): a sum (mod 2^64, so the order of the additions does not
// matter) of a mixed word per row pointer and per entry — position, column and the value's bit pattern.
__device__ __forceinline__ unsigned long long mix64(unsigned long long x)
{
   x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ull; x ^= x >> 27; x *= 0x94d049bb133111ebull; x ^= x >> 31;
   return x;
}
__global__ __launch_bounds__(256)
void csr_checksum_kernel(const int *__restrict__ Ai, const int *__restrict__ Aj, const unsigned long long *__restrict__ Aa,
                         int n, int nnz, unsigned long long *out)
{
   unsigned long long acc = 0;
   const size_t stride = (size_t) gridDim.x * blockDim.x, t = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
   for (size_t r = t; r <= (size_t) n; r += stride) { acc += mix64(((unsigned long long) r << 32) ^ (unsigned) Ai[r] ^ 0x9e3779b97f4a7c15ull); }
   for (size_t k = t; k < (size_t) nnz; k += stride)
   {
      acc += mix64(Aa[k] ^ mix64(((unsigned long long) k << 32) | (unsigned) Aj[k]));
   }
#pragma unroll
   for (int off = 32; off > 0; off >>= 1) { acc += __shfl_xor(acc, off, 64); }
   if ((threadIdx.x & 63) == 0) { atomicAdd(out, acc); }
}
unsigned long long device_csr_checksum(const int *Ai, const int *Aj, const double *Aa, int n, int nnz, hipStream_t s)
{
   unsigned long long *d_out = reinterpret_cast<unsigned long long *>(reduce_scratch(2));
   HIP_CHECK(hipMemsetAsync(d_out, 0, sizeof(unsigned long long), s));
   const size_t work = (size_t) std::max(nnz, n + 1);
   const int grid = (int) std::min<size_t>((work + 255) / 256, (size_t) 16 * handle().num_cus);
   hipLaunchKernelGGL(csr_checksum_kernel, dim3(std::max(grid, 1)), dim3(256), 0, s, Ai, Aj, reinterpret_cast<const unsigned long long *>(Aa), n, nnz, d_out);
   unsigned long long h = 0;
   HIP_CHECK(hipMemcpyAsync(&h, d_out, sizeof(h), hipMemcpyDeviceToHost, s));
   HIP_CHECK(hipStreamSynchronize(s));
   return h;
}

// ---- value codes (SpmvPlan::d_codes).  Distinct values are collected as bit patterns in an open-addressed table with
// room for four times what a code can name; the pass ends early once more than DICT_CAP patterns are in (a matrix that is
// not a stencil overflows within its first few thousand entries).  The all-ones pattern marks an empty slot: a matrix that
// holds that NaN is not coded.
constexpr int DICT_SLOTS = 4 * DICT_CAP;
constexpr unsigned long long DICT_EMPTY = ~0ull;
__device__ __forceinline__ unsigned dict_hash(unsigned long long v)
{
   v ^= v >> 33; v *= 0xff51afd7ed558ccdull; v ^= v >> 29;
   return (unsigned) v & (DICT_SLOTS - 1);
}
__global__ __launch_bounds__(256)
void dict_collect_kernel(const unsigned long long *__restrict__ vals, size_t n, unsigned long long *table, int *count)
{
   unsigned long long last = DICT_EMPTY;
   for (size_t k = (size_t) blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (size_t) gridDim.x * blockDim.x)
   {
      const unsigned long long v = vals[k];
      if (v == last) { continue; }
      if (v == DICT_EMPTY || __hip_atomic_load(count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > DICT_CAP)
      {
         __hip_atomic_store(count, DICT_CAP + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
         return;
      }
      unsigned slot = dict_hash(v);
      for (;;)
      {
         unsigned long long cur = __hip_atomic_load(table + slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
         if (cur == DICT_EMPTY)
         {
            cur = atomicCAS(table + slot, DICT_EMPTY, v);
            if (cur == DICT_EMPTY) { atomicAdd(count, 1); break; }
         }
         if (cur == v) { break; }
         // threads in flight may overshoot DICT_CAP before they see the count: a full table ends here, not in an endless probe
         if (__hip_atomic_load(count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > DICT_CAP) { return; }
         slot = (slot + 1) & (DICT_SLOTS - 1);
      }
      last = v;
   }
}
__global__ __launch_bounds__(256)
void dict_encode_kernel(const unsigned long long *__restrict__ vals, size_t n, const unsigned long long *__restrict__ table,
                        const unsigned char *__restrict__ code_of_slot, unsigned char *__restrict__ codes)
{
   __shared__ unsigned long long t[DICT_SLOTS];
   __shared__ unsigned char c[DICT_SLOTS];
   for (int i = threadIdx.x; i < DICT_SLOTS; i += blockDim.x) { t[i] = table[i]; c[i] = code_of_slot[i]; }
   __syncthreads();
   for (size_t k = (size_t) blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (size_t) gridDim.x * blockDim.x)
   {
      const unsigned long long v = vals[k];
      unsigned slot = dict_hash(v);
      while (t[slot] != v) { slot = (slot + 1) & (DICT_SLOTS - 1); }
      codes[k] = c[slot];
   }
}

bool device_value_codes(const double *Aa, size_t nnz, unsigned char **codes_out, double **dict_out, double **dict32_out, int *ndict_out,
                        hipStream_t s)
{
   *codes_out = nullptr; *dict_out = nullptr; *dict32_out = nullptr; *ndict_out = 0;
   if (nnz < 8) { return false; }           // (the kernels' rotating check reads eight consecutive entries)
   // every allocation is checked (PLAN_SITE_CODES): on failure the matrix is simply not coded
   unsigned long long *d_table = nullptr;
   if (!plan_alloc((void **) &d_table, sizeof(unsigned long long) * DICT_SLOTS + sizeof(int) * 4, PLAN_SITE_CODES)) { return false; }
   int *d_count = reinterpret_cast<int *>(d_table + DICT_SLOTS);
   HIP_CHECK(hipMemsetAsync(d_table, 0xff, sizeof(unsigned long long) * DICT_SLOTS, s));
   HIP_CHECK(hipMemsetAsync(d_count, 0, sizeof(int) * 4, s));
   const unsigned long long *bits = reinterpret_cast<const unsigned long long *>(Aa);
   // a first look at the head of the array settles the matrices that are not stencils (every thread of a full-size launch
   // would fight over the table's 1024 slots first: half a millisecond per matrix of a hierarchy)
   int count = 0;
   const size_t head = std::min<size_t>(nnz, 16384);
   hipLaunchKernelGGL(dict_collect_kernel, dim3((unsigned) ((head + 255) / 256)), dim3(256), 0, s, bits, head, d_table, d_count);
   HIP_CHECK(hipMemcpyAsync(&count, d_count, sizeof(int), hipMemcpyDeviceToHost, s));
   HIP_CHECK(hipStreamSynchronize(s));
   if (count > DICT_CAP) { plan_free(d_table); return false; }
   int grid = (int) std::min<size_t>((nnz + 255) / 256, (size_t) 8 * handle().num_cus);
   if (nnz > head) { hipLaunchKernelGGL(dict_collect_kernel, dim3(grid), dim3(256), 0, s, bits, nnz, d_table, d_count); }
   std::vector<unsigned long long> table(DICT_SLOTS);
   HIP_CHECK(hipMemcpyAsync(&count, d_count, sizeof(int), hipMemcpyDeviceToHost, s));
   HIP_CHECK(hipMemcpyAsync(table.data(), d_table, sizeof(unsigned long long) * DICT_SLOTS, hipMemcpyDeviceToHost, s));
   HIP_CHECK(hipStreamSynchronize(s));
   if (count > DICT_CAP || count <= 0) { plan_free(d_table); return false; }
   // codes in the order of the bit patterns: the same matrix gets the same codes whatever order the table filled in
   std::vector<unsigned long long> vals;
   for (unsigned long long v : table) { if (v != DICT_EMPTY) { vals.push_back(v); } }
   std::sort(vals.begin(), vals.end());
   if ((int) vals.size() != count) { plan_free(d_table); hypre_error_w_msg(HYPRE_ERROR_GENERIC, "value table: count and contents disagree"); return false; }
   std::vector<unsigned char> code_of_slot(DICT_SLOTS, 0);
   for (int i = 0; i < DICT_SLOTS; i++)
   {
      if (table[(size_t) i] != DICT_EMPTY) { code_of_slot[(size_t) i] = (unsigned char) (std::lower_bound(vals.begin(), vals.end(), table[(size_t) i]) - vals.begin()); }
   }
   std::vector<double> dict(DICT_CAP, 0.0), dict32(DICT_CAP, 0.0);
   for (size_t i = 0; i < vals.size(); i++)
   {
      double d; memcpy(&d, &vals[i], sizeof(double));
      dict[i] = d; dict32[i] = (double) (float) d;
   }
   unsigned char *d_cos = nullptr, *d_codes = nullptr;
   double *d_dict = nullptr;
   const size_t ncodes = (nnz + 31) & ~(size_t) 15;
   if (!plan_alloc((void **) &d_cos, DICT_SLOTS, PLAN_SITE_CODES) || !plan_alloc((void **) &d_codes, ncodes, PLAN_SITE_CODES) ||
       !plan_alloc((void **) &d_dict, sizeof(double) * 2 * DICT_CAP, PLAN_SITE_CODES))
   {
      plan_free(d_cos); plan_free(d_codes); plan_free(d_dict); plan_free(d_table);
      return false;
   }
   HIP_CHECK(hipMemcpyAsync(d_cos, code_of_slot.data(), DICT_SLOTS, hipMemcpyHostToDevice, s));
   HIP_CHECK(hipMemcpyAsync(d_dict, dict.data(), sizeof(double) * DICT_CAP, hipMemcpyHostToDevice, s));
   HIP_CHECK(hipMemcpyAsync(d_dict + DICT_CAP, dict32.data(), sizeof(double) * DICT_CAP, hipMemcpyHostToDevice, s));
   HIP_CHECK(hipMemsetAsync(d_codes + (ncodes - 32), 0, 32, s));
   grid = (int) std::min<size_t>((nnz + 255) / 256, (size_t) 16 * handle().num_cus);
   hipLaunchKernelGGL(dict_encode_kernel, dim3(grid), dim3(256), 0, s, bits, nnz, d_table, d_cos, d_codes);
   HIP_CHECK(hipStreamSynchronize(s));
   plan_free(d_cos);
   plan_free(d_table);
   *codes_out = d_codes; *dict_out = d_dict; *dict32_out = d_dict + DICT_CAP; *ndict_out = count;
   return true;
}

// ---------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------
static inline size_t tiled_lds_bytes(const SpmvPlan *plan, int &rowsum_elems, int &rp_cap)
{
   // LDS beside the products is sized by the most rows any tile of this matrix holds: row pointers
   // for all of them (up to RP_CAP), row sums only on the multi-lane paths (mean row length of a
   // tile > 12, which also means fewer than SPMV_THREADS rows in that tile)
   static int fit = -1;
   if (fit < 0) { const char *e = getenv("HYPRE_AMD_SPMV_LDSFIT"); fit = e ? atoi(e) : 1; }
   const int rows = (fit && plan->max_tile_rows > 0) ? plan->max_tile_rows : RP_CAP;
   rp_cap = rows < RP_CAP ? rows : RP_CAP;
   rowsum_elems = plan->max_row_nnz > 12 ? ((rows < SPMV_THREADS ? rows : SPMV_THREADS) + 1) & ~1 : 0;
   return sizeof(double) * (size_t) (plan->prod_elems + rowsum_elems) + sizeof(int) * (size_t) (rp_cap + 4);
}

template <int OP, bool F32, bool FILL, bool GT>
static void launch_tiled_gt(const SpmvPlan *plan, const SpmvArgs &a, hipStream_t s)
{
   int rowsum_elems, rp_cap;
   const size_t lds = tiled_lds_bytes(plan, rowsum_elems, rp_cap);
   // the chunked tile -> XCD map covers whole groups of 8 * xcd_map workgroups; a placement table is one workgroup per tile
   const int unit = a.tile_perm ? 1 : (a.xcd_map > 0 ? 8 * a.xcd_map : 8);
   const int grid = ((plan->num_tiles + unit - 1) / unit) * unit;
   hipLaunchKernelGGL((spmv_tiled_kernel<OP, F32, FILL, GT>), dim3(grid), dim3(SPMV_THREADS), lds, s, a,
                      plan->d_tile_row, plan->d_tile_k, plan->num_tiles, plan->prod_elems, rowsum_elems, rp_cap);
}

template <int OP, int VF, bool FILL>
static void launch_xs(const SpmvPlan *plan, const SpmvArgs &a, hipStream_t s)
{
   int rowsum_elems, rp_cap;
   size_t lds = tiled_lds_bytes(plan, rowsum_elems, rp_cap);
   if (VF == VF_CODE) { lds = xs_dict_offset(plan->prod_elems, rowsum_elems, rp_cap) + sizeof(double) * (size_t) ((a.ndict + 1) & ~1); }
   const int unit = a.tile_perm ? 1 : (a.xcd_map > 0 ? 8 * a.xcd_map : 8);
   const int grid = ((plan->num_tiles + unit - 1) / unit) * unit;
   hipLaunchKernelGGL((spmv_xs_kernel<OP, VF, FILL>), dim3(grid), dim3(SPMV_THREADS), lds, s, a,
                      plan->d_tile_row, plan->d_tile_k, plan->d_xs_cnt, plan->d_xs_desc, plan->d_lidx,
                      plan->num_tiles, plan->prod_elems, rowsum_elems, rp_cap, plan->xs_launch_units);
}

template <int OP, int W, int KP>
static void launch_sl_form(const SpmvPlan *plan, const SpmvArgs &a, hipStream_t s)
{
   const int stage_elems = 2 * plan->sl_launch_units + 8;
   const size_t lds = sizeof(double) * (size_t) (stage_elems + ((a.ndict + 1) & ~1));
   const int grid = plan->d_sl_perm ? plan->sl_blocks : ((plan->sl_blocks + 63) / 64) * 64;
   hipLaunchKernelGGL((spmv_sl_kernel<OP, W, KP>), dim3(grid), dim3(SPMV_THREADS), lds, s, a, plan->d_sl_cnt, plan->d_sl_desc, plan->d_sl_k0,
                      plan->d_sl_fp, plan->d_sl_perm, plan->d_sl_data, plan->sl_blocks, plan->num_rows, plan->nnz, stage_elems);
}
template <int OP>
static void launch_sl(const SpmvPlan *plan, const SpmvArgs &a, hipStream_t s)
{
   if (plan->sl_w == 1) { launch_sl_form<OP, 1, 8>(plan, a, s); }
   else if (plan->sl_k == 8) { launch_sl_form<OP, 2, 8>(plan, a, s); }
   else { launch_sl_form<OP, 2, 16>(plan, a, s); }
}

template <int OP, int KP, bool F32>
static void launch_rs_form(const SpmvPlan *plan, const SpmvArgs &a, hipStream_t s)
{
   const int stage_elems = 2 * plan->rs_units + 8;
   const size_t lds = sizeof(double) * (size_t) (stage_elems + SPMV_THREADS);
   const int grid = plan->d_rs_perm ? plan->rs_blocks : ((plan->rs_blocks + 63) / 64) * 64;
   hipLaunchKernelGGL((spmv_rs_kernel<OP, KP, F32>), dim3(grid), dim3(SPMV_THREADS), lds, s, a, plan->d_rs_desc, plan->d_rs_perm, plan->d_rs_hdr,
                      plan->d_rs_meta, plan->d_rs_val, plan->d_rs_val32, plan->d_rs_idx, plan->rs_blocks, plan->num_rows, plan->rs_rows, plan->rs_w,
                      stage_elems);
}
template <int OP, bool F32>
static void launch_rs(const SpmvPlan *plan, const SpmvArgs &a, hipStream_t s)
{
   switch (plan->rs_kp)
   {
      case 8:  launch_rs_form<OP, 8, F32>(plan, a, s); break;
      case 16: launch_rs_form<OP, 16, F32>(plan, a, s); break;
      case 24: launch_rs_form<OP, 24, F32>(plan, a, s); break;
      default: launch_rs_form<OP, 32, F32>(plan, a, s); break;
   }
}
// the row-slice kernel serves this launch: the plan has the form, the whole matrix is multiplied, x can be read in 16-byte pieces
static inline bool takes_rs(const SpmvPlan *plan, const SpmvArgs &a)
{
   return plan->d_rs_val && a.variant == 2 && a.fill == HYPRE_SPMV_FILL_WHOLE && (((uintptr_t) a.x) & 15) == 0;
}

// the x-staged kernel serves this launch: the plan carries the per-tile piece lists and x can be read in 16-byte pieces
static inline bool takes_xs(const SpmvPlan *plan, const SpmvArgs &a)
{
   return plan->tiled && a.variant == 2 && a.gather_t && plan->d_lidx && (((uintptr_t) a.x) & 15) == 0;
}

template <int OP, bool F32, bool FILL>
static void launch_tiled(const SpmvPlan *plan, const SpmvArgs &a, hipStream_t s)
{
   if (!FILL && a.use_rs) { launch_rs<OP, F32>(plan, a, s); return; }
   // x staged through LDS (variant 2, the default)
   if (takes_xs(plan, a))
   {
      if (a.Ac8 && !FILL && plan->d_sl_data) { launch_sl<OP>(plan, a, s); }      // slice form: a lane per row (or half row)
      else if (a.Ac8) { launch_xs<OP, VF_CODE, FILL>(plan, a, s); }
      else { launch_xs<OP, F32 ? VF_F32 : VF_F64, FILL>(plan, a, s); }
      return;
   }
   if (a.gather_t) { launch_tiled_gt<OP, F32, FILL, true>(plan, a, s); }
   else { launch_tiled_gt<OP, F32, FILL, false>(plan, a, s); }
}

template <int OP>
static void launch_spmv_op(const SpmvPlan *plan, const SpmvArgs &a, hipStream_t s)
{
   const bool f32  = a.Aa32 != nullptr;
   const bool fill = a.fill != HYPRE_SPMV_FILL_WHOLE;
   if (plan->tiled)
   {
      if (f32) { if (fill) launch_tiled<OP, true, true>(plan, a, s); else launch_tiled<OP, true, false>(plan, a, s); }
      else     { if (fill) launch_tiled<OP, false, true>(plan, a, s); else launch_tiled<OP, false, false>(plan, a, s); }
   }
   else
   {
      int grid = (plan->num_rows + 3) / 4;
      if (grid > 4096) { grid = 4096; }
      if (grid < 1) { grid = 1; }
      dim3 g(grid), b(SPMV_THREADS);
      if (f32)
      {
         if (fill) hipLaunchKernelGGL((spmv_wave_kernel<OP, true, true>), g, b, 0, s, a, plan->num_rows);
         else      hipLaunchKernelGGL((spmv_wave_kernel<OP, true, false>), g, b, 0, s, a, plan->num_rows);
      }
      else
      {
         if (fill) hipLaunchKernelGGL((spmv_wave_kernel<OP, false, true>), g, b, 0, s, a, plan->num_rows);
         else      hipLaunchKernelGGL((spmv_wave_kernel<OP, false, false>), g, b, 0, s, a, plan->num_rows);
      }
   }
}

void launch_spmv(const SpmvPlan *plan, const SpmvArgs &args, SpmvOp op, hipStream_t s)
{
   if (plan->num_rows <= 0) { return; }
   SpmvArgs a = args;
   a.last_quad = (plan->nnz > 0 ? (int) (plan->nnz - 1) : 0) & ~3;
   a.x_last = plan->num_cols > 0 ? plan->num_cols - 1 : 0;
   a.tile_perm = plan->d_tile_perm;
   a.tile_fp = plan->d_tile_fp;
   a.stale = plan->d_stale;
   a.nnz = plan->nnz;
   a.rot = const_cast<SpmvPlan *>(plan)->launches++;
   // value codes (the x-staged kernel only): one byte per entry and the table of the matrix's values
   a.Ac8 = nullptr; a.dict = nullptr; a.ndict = 0; a.dict_rounded = 0;
   if (plan->d_codes && takes_xs(plan, a))
   {
      const bool rounded = handle().fp32_values || a.Aa32 != nullptr;
      a.Ac8 = plan->d_codes; a.ndict = plan->ndict;
      a.dict = rounded ? plan->d_dict32 : plan->d_dict;
      a.dict_rounded = rounded ? 1 : 0;
   }
   a.use_rs = takes_rs(plan, a) ? 1 : 0;
   if (a.use_rs && (handle().fp32_values || a.Aa32))
   {
      // mixed precision on a row-slice matrix: the fp32 copy of the values in slice order (same positions), made once;
      // Aa32 only selects the fp32 kernel (the row-slice kernel reads the plan's copy)
      SpmvPlan *mp = const_cast<SpmvPlan *>(plan);
      if (!mp->d_rs_val32 && plan_alloc((void **) &mp->d_rs_val32, sizeof(float) * ((size_t) plan->nnz + 64), PLAN_SITE_ROWSLICE))
      {
         launch_f64_to_f32(plan->d_rs_val, mp->d_rs_val32, (size_t) plan->nnz, s);
      }
      if (mp->d_rs_val32) { a.Aa32 = mp->d_rs_val32; } else { a.use_rs = 0; }
   }
   if (handle().fp32_values && !a.Aa32 && !a.Ac8 && plan->nnz > 0)
   {
      // mixed precision: matrix values stream as fp32 (converted once per matrix), vectors and
      // accumulation stay fp64
      SpmvPlan *mp = const_cast<SpmvPlan *>(plan);
      if (!mp->a32)
      {
         HIP_CHECK(hipMalloc((void **) &mp->a32, sizeof(float) * (((size_t) plan->nnz + 3) & ~(size_t) 3)));
         launch_f64_to_f32(plan->a, mp->a32, (size_t) plan->nnz, s);
      }
      a.Aa32 = mp->a32;
   }
   if (a.rowmap && op != OP_JACOBI_MAP) { hypre_error_w_msg(HYPRE_ERROR_ARG, "row map given to an operation that does not read it"); return; }
   {
      // byte accounting (Handle::bytes_csr / bytes_stream): the matrix pass, the compulsory read of x (every column once,
      // or every entry's column when the matrix has fewer entries than columns: a colour's rows) and the row operands
      const double nz = (double) plan->nnz, nr = (double) plan->num_rows, vw = (a.Aa32 || a.dict_rounded) ? 4.0 : 8.0;
      const double xcols = 8.0 * std::min((double) plan->num_cols, nz);
      double rowb = 0.0;
      switch (op)
      {
         case OP_AXPBY:      rowb = 8.0 + ((a.beta != 0.0) ? 8.0 : 0.0); break;              // y (+ b)
         case OP_AXPBY_DIV:  rowb = 24.0 + ((a.beta != 0.0) ? 8.0 : 0.0); break;             // y, d read, u written (+ b)
         case OP_RESID_RD:   rowb = 16.0 + ((a.beta != 0.0) ? 8.0 : 0.0); break;             // d, z (+ f)
         case OP_TSGS_FIRST: rowb = 32.0 + ((a.beta != 0.0) ? 8.0 : 0.0); break;             // d, z row, z' write, u write (+ u read)
         case OP_JACOBI:     rowb = 24.0; break;                                              // f, d, u'   (u is the x gather)
         case OP_JACOBI_CF:  rowb = 28.0; break;                                              // + marker
         case OP_JACOBI_MAP: rowb = 28.0 + (a.marker ? 4.0 : 0.0); break;                     // + row map
         case OP_TSGS:       rowb = 32.0; break;                                              // d, z', u read and write
      }
      const double rows_b = 4.0 * (nr + 1.0) + rowb * nr + xcols;
      // what the format of the kernel that serves this launch requires: the matrix stream, the per-tile (per-block) tables
      // — bounds, piece descriptors, fingerprint —, the value table and the rotating value check of a coded matrix
      const bool staged = takes_xs(plan, a);
      const bool slice = staged && a.Ac8 && a.fill == HYPRE_SPMV_FILL_WHOLE && plan->d_sl_data != nullptr;
      double streamed;
      if (a.use_rs)
      {
         // row slices: value + 16-bit position per entry, no row pointers; per block the wave headers, lane words and descriptors
         streamed = nz * (vw + 2.0) + (double) plan->rs_blocks * (4.0 * (4 * RS_HDR + SPMV_THREADS + XS_DESC)) + rowb * nr + xcols;
      }
      else if (slice)
      {
         streamed = (double) plan->sl_blocks * ((double) SPMV_THREADS * plan->sl_k * 3.0 + 4.0 * (XS_DESC + 4) + 8.0 * a.ndict + 4.0 * 72.0) + rows_b;
      }
      else if (staged)
      {
         streamed = nz * ((a.Ac8 ? 1.0 : vw) + 2.0) + rows_b + (double) plan->num_tiles * (4.0 * (XS_DESC + 6) + (a.Ac8 ? 8.0 * a.ndict : 0.0) + ((a.Ac8 || a.Aa32) ? 4.0 * 72.0 : 0.0));
      }
      else { streamed = nz * (vw + 4.0) + rows_b + (plan->tiled ? 8.0 * plan->num_tiles : 0.0); }
      account_bytes(nz * (vw + 4.0) + rows_b, streamed);
   }
   switch (op)
   {
      case OP_AXPBY:     launch_spmv_op<OP_AXPBY>(plan, a, s); break;
      case OP_AXPBY_DIV: launch_spmv_op<OP_AXPBY_DIV>(plan, a, s); break;
      case OP_RESID_RD:  launch_spmv_op<OP_RESID_RD>(plan, a, s); break;
      case OP_TSGS_FIRST: launch_spmv_op<OP_TSGS_FIRST>(plan, a, s); break;
      case OP_JACOBI:    launch_spmv_op<OP_JACOBI>(plan, a, s); break;
      case OP_JACOBI_CF: launch_spmv_op<OP_JACOBI_CF>(plan, a, s); break;
      case OP_JACOBI_MAP: launch_spmv_op<OP_JACOBI_MAP>(plan, a, s); break;
      case OP_TSGS:      launch_spmv_op<OP_TSGS>(plan, a, s); break;
   }
}

// Fused product with a multivector (spmv_xs_mv_kernel): columns v < nv of x, b, y lie xstride / bstride / ystride doubles
// apart.  false: this plan or these operands are not served (the caller multiplies column by column); nothing was launched.
template <int NV, bool CODED>
static void launch_xs_mv(const SpmvPlan *plan, const SpmvArgs &a, int stage_elems, int win_elems, int rp_cap, size_t lds,
                         long xstride, long bstride, long ystride, hipStream_t s)
{
   static bool raised = false;
   if (!raised)
   {
      (void) hipFuncSetAttribute((const void *) (spmv_xs_mv_kernel<NV, CODED>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      (void) hipGetLastError();
      raised = true;
   }
   const int unit = a.tile_perm ? 1 : (a.xcd_map > 0 ? 8 * a.xcd_map : 8);
   const int grid = ((plan->num_tiles + unit - 1) / unit) * unit;
   // a lane per row where the rows are short: a fifth and a sixth wave take rows 256 .. 383 of a tile in the same pass
   const int threads = (plan->max_tile_rows > SPMV_THREADS && plan->max_tile_rows <= 320) ? 320 :
                       (plan->max_tile_rows > 320 ? MV_THREADS_MAX : SPMV_THREADS);
   hipLaunchKernelGGL((spmv_xs_mv_kernel<NV, CODED>), dim3(grid), dim3(threads), lds, s, a,
                      plan->d_tile_row, plan->d_tile_k, plan->d_xs_cnt, plan->d_xs_desc, plan->d_lidx,
                      plan->num_tiles, stage_elems, win_elems, rp_cap, plan->xs_launch_units, xstride, bstride, ystride);
}

long &spmv_mv_launches() { static long n = 0; return n; }
template <int NV, int W, int KP>
static void launch_sl_mv(const SpmvPlan *plan, const SpmvArgs &a, int stage_elems, size_t lds, long xstride, long bstride, long ystride, hipStream_t s)
{
   static bool raised = false;
   if (!raised)
   {
      (void) hipFuncSetAttribute((const void *) (spmv_sl_mv_kernel<NV, W, KP>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      (void) hipGetLastError();
      raised = true;
   }
   const int grid = plan->d_sl_perm ? plan->sl_blocks : ((plan->sl_blocks + 63) / 64) * 64;
   hipLaunchKernelGGL((spmv_sl_mv_kernel<NV, W, KP>), dim3(grid), dim3(SPMV_THREADS), lds, s, a, plan->d_sl_desc, plan->d_sl_k0,
                      plan->d_sl_fp, plan->d_sl_perm, plan->d_sl_data, plan->sl_blocks, plan->num_rows, plan->nnz, stage_elems, xstride, bstride, ystride);
}
template <int NV>
static void launch_sl_mv_nv(const SpmvPlan *plan, const SpmvArgs &a, int stage_elems, size_t lds, long xstride, long bstride, long ystride, hipStream_t s)
{
   if (plan->sl_w == 1) { launch_sl_mv<NV, 1, 8>(plan, a, stage_elems, lds, xstride, bstride, ystride, s); }
   else if (plan->sl_k == 8) { launch_sl_mv<NV, 2, 8>(plan, a, stage_elems, lds, xstride, bstride, ystride, s); }
   else { launch_sl_mv<NV, 2, 16>(plan, a, stage_elems, lds, xstride, bstride, ystride, s); }
}
static inline size_t sl_mv_lds_bytes(const SpmvPlan *plan, int nv, int ndict, int &stage_elems)
{
   stage_elems = 2 * plan->sl_launch_units + 8;
   return sizeof(double) * ((size_t) nv * stage_elems + (size_t) ((ndict + 1) & ~1));
}

constexpr int MV_WIN = 2320;        // SPMV_TILE + SPMV_THREADS entries of a window and its spill, a multiple of 16
static inline size_t mv_lds_bytes(const SpmvPlan *plan, int nv, bool coded, int ndict, int &stage_elems, int &rp_cap)
{
   int rowsum_elems;
   (void) tiled_lds_bytes(plan, rowsum_elems, rp_cap);
   stage_elems = 2 * plan->xs_launch_units + 8;
   return sizeof(double) * (size_t) nv * stage_elems + (coded ? (size_t) MV_WIN : sizeof(double) * (size_t) MV_WIN) + sizeof(unsigned short) * (size_t) MV_WIN +
          sizeof(int) * (size_t) ((rp_cap + 4) & ~3) + (coded ? sizeof(double) * (size_t) ((ndict + 1) & ~1) : 0);
}

bool spmv_mv_serves(const SpmvPlan *plan, const SpmvArgs &args, long xstride)
{
   SpmvArgs a = args;
   if (!plan->tiled || !takes_xs(plan, a) || a.fill != HYPRE_SPMV_FILL_WHOLE || a.Aa32 || handle().fp32_values) { return false; }
   if (plan->max_row_nnz > SPMV_THREADS || (xstride & 1) != 0 || !plan->d_tile_fp || !plan->d_stale) { return false; }
   // nearly all tiles staged: the others gather NV columns through the cache, entry by entry
   if ((long long) plan->xs_tiles * 10 < (long long) plan->num_tiles * 9) { return false; }
   int stage_elems, rp_cap;
   return mv_lds_bytes(plan, 2, plan->d_codes != nullptr, plan->ndict, stage_elems, rp_cap) <= (size_t) 160 * 1024;
}

bool launch_spmv_mv(const SpmvPlan *plan, const SpmvArgs &args, int nv, long xstride, long bstride, long ystride, hipStream_t s)
{
   if (plan->num_rows <= 0 || nv < 2 || !spmv_mv_serves(plan, args, xstride)) { return false; }
   SpmvArgs a = args;
   a.last_quad = (plan->nnz > 0 ? (int) (plan->nnz - 1) : 0) & ~3;
   a.x_last = plan->num_cols > 0 ? plan->num_cols - 1 : 0;
   a.tile_perm = plan->d_tile_perm;
   a.tile_fp = plan->d_tile_fp;
   a.stale = plan->d_stale;
   a.nnz = plan->nnz;
   a.Ac8 = nullptr; a.dict = nullptr; a.ndict = 0; a.dict_rounded = 0; a.use_rs = 0;
   const bool coded = plan->d_codes != nullptr;
   if (coded) { a.Ac8 = plan->d_codes; a.ndict = plan->ndict; a.dict = plan->d_dict; }
   const double nz = (double) plan->nnz, nr = (double) plan->num_rows;
   const double vecs = (8.0 + ((a.beta != 0.0) ? 8.0 : 0.0)) * nr + 8.0 * std::min((double) plan->num_cols, nz);     // per column: y (+ b), x
   for (int v0 = 0; v0 < nv; )
   {
      const bool slice = coded && plan->d_sl_data != nullptr;      // a lane per row (or half row), the matrix words in registers
      // columns per pass: four over the tiles (the matrix stream is what the pass saves); the slice form's matrix words are
      // few, its passes are bound by the x pieces, and a fourth staged copy costs a resident workgroup per CU: threes and twos
      // (measured per column on the 256^3 7-point operator: 0.108 / 0.104 / 0.118 ms at 2 / 3 / 4 columns, 0.150 alone)
      const int left = nv - v0;
      int g = slice ? (left == 4 ? 2 : std::min(3, left)) : std::min(4, left), stage_elems = 0, rp_cap = 0;
      size_t lds = 0;
      if (slice) { while (g > 1 && (lds = sl_mv_lds_bytes(plan, g, a.ndict, stage_elems)) > (size_t) 160 * 1024) { g--; } }
      else { while (g > 1 && (lds = mv_lds_bytes(plan, g, coded, a.ndict, stage_elems, rp_cap)) > (size_t) 160 * 1024) { g--; } }
      SpmvArgs c = a;
      c.x = a.x + (size_t) v0 * xstride;
      c.y = a.y + (size_t) v0 * ystride;
      c.b = a.b ? a.b + (size_t) v0 * bstride : nullptr;
      if (g == 1) { launch_spmv(plan, c, OP_AXPBY, s); v0 += 1; continue; }
      c.rot = const_cast<SpmvPlan *>(plan)->launches++;
      spmv_mv_launches()++;
      // bytes: the matrix once (CSR count: 12 bytes per entry and the row pointers; streamed: what the x-staged form holds
      // per entry and per tile), the vectors of every column
      if (slice)
      {
         account_bytes(nz * 12.0 + 4.0 * (nr + 1.0) + g * vecs,
                       (double) plan->sl_blocks * ((double) SPMV_THREADS * plan->sl_k * 3.0 + 4.0 * (XS_DESC + 4) + 8.0 * a.ndict + 4.0 * 72.0) +
                       4.0 * (nr + 1.0) + g * vecs);
         switch (g)
         {
            case 2:  launch_sl_mv_nv<2>(plan, c, stage_elems, lds, xstride, bstride, ystride, s); break;
            case 3:  launch_sl_mv_nv<3>(plan, c, stage_elems, lds, xstride, bstride, ystride, s); break;
            default: launch_sl_mv_nv<4>(plan, c, stage_elems, lds, xstride, bstride, ystride, s); break;
         }
         v0 += g;
         continue;
      }
      account_bytes(nz * 12.0 + 4.0 * (nr + 1.0) + g * vecs,
                    nz * ((coded ? 1.0 : 8.0) + 2.0) + 4.0 * (nr + 1.0) + g * vecs +
                    (double) plan->num_tiles * (4.0 * (XS_DESC + 6) + (coded ? 8.0 * a.ndict + 4.0 * 72.0 : 0.0)));
      switch (g)
      {
         case 2: if (coded) launch_xs_mv<2, true>(plan, c, stage_elems, MV_WIN, rp_cap, lds, xstride, bstride, ystride, s);
                 else       launch_xs_mv<2, false>(plan, c, stage_elems, MV_WIN, rp_cap, lds, xstride, bstride, ystride, s); break;
         case 3: if (coded) launch_xs_mv<3, true>(plan, c, stage_elems, MV_WIN, rp_cap, lds, xstride, bstride, ystride, s);
                 else       launch_xs_mv<3, false>(plan, c, stage_elems, MV_WIN, rp_cap, lds, xstride, bstride, ystride, s); break;
         default: if (coded) launch_xs_mv<4, true>(plan, c, stage_elems, MV_WIN, rp_cap, lds, xstride, bstride, ystride, s);
                  else       launch_xs_mv<4, false>(plan, c, stage_elems, MV_WIN, rp_cap, lds, xstride, bstride, ystride, s); break;
      }
      v0 += g;
   }
   return true;
}

void launch_spmv_allrows_update(int num_rows, const SpmvArgs &args, hipStream_t s)
{
   if (num_rows <= 0) { return; }
   const int grid = (num_rows * 8 + SPMV_THREADS - 1) / SPMV_THREADS;
   hipLaunchKernelGGL(spmv_rownnz_kernel, dim3(grid), dim3(SPMV_THREADS), 0, s, args, (const int *) nullptr, num_rows);
}

void launch_spmv_rownnz(const HYPRE_Int *rownnz, int num_rownnz, const SpmvArgs &args, hipStream_t s)
{
   if (num_rownnz <= 0) { return; }
   account_bytes(36.0 * num_rownnz);        // latency-bound ghost-block pass: listed row, its pointers, about one entry, y read and write
   const int grid = (num_rownnz * 8 + SPMV_THREADS - 1) / SPMV_THREADS;
   hipLaunchKernelGGL(spmv_rownnz_kernel, dim3(grid), dim3(SPMV_THREADS), 0, s, args, rownnz, num_rownnz);
}

// The code object of this file is loaded when one of its kernels is first asked for: ensure_device() asks here, so that
// the load (tens of milliseconds per file) is part of bringing the device up, not of the first setup or solve.
#if XS_TIMING
// tiles > 0: the launches from now on record into a fresh buffer of 4 * tiles numbers; out != null: the buffer's contents
extern "C" void hypre_amd_XsTiming(unsigned *out, int tiles)
{
   static unsigned *buf = nullptr;
   static int len = 0;
   HIP_CHECK(hipDeviceSynchronize());
   if (out && buf) { HIP_CHECK(hipMemcpy(out, buf, sizeof(unsigned) * 4 * (size_t) len, hipMemcpyDeviceToHost)); }
   if (tiles > 0)
   {
      if (buf) { HIP_CHECK(hipFree(buf)); }
      HIP_CHECK(hipMalloc((void **) &buf, sizeof(unsigned) * 4 * (size_t) tiles));
      HIP_CHECK(hipMemset(buf, 0, sizeof(unsigned) * 4 * (size_t) tiles));
      len = tiles;
      HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(xs_timing_buf), &buf, sizeof(buf)));
   }
}
#endif
void preload_spmv_kernels() { hipFuncAttributes at; (void) hipFuncGetAttributes(&at, (const void *) (spmv_xs_kernel<0, 0, false>)); (void) hipGetLastError(); }

}  // namespace hamd
