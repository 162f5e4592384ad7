// hypre_amd internal declarations shared by the host-side translation units
// and the HIP kernel file.  Nothing here is part of the C ABI (see include/).
#pragma once

#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdint>
#include <vector>

#include "HYPRE_amd_utilities.h"
#include "hypre_amd_seq_mv.h"
#include "hypre_amd_comm.h"
#include "hypre_amd_parcsr_mv.h"

// ---------------------------------------------------------------------------
// error plumbing
// ---------------------------------------------------------------------------
#define HIP_CHECK(call)                                                          \
   do {                                                                          \
      hipError_t e_ = (call);                                                    \
      if (e_ != hipSuccess) {                                                    \
         char msg_[512];                                                         \
         snprintf(msg_, sizeof(msg_), "HIP error %d (%s) in %s", (int) e_,       \
                  hipGetErrorString(e_), #call);                                 \
         hypre_error_handler(__FILE__, __LINE__, HYPRE_ERROR_GENERIC, msg_);     \
      }                                                                          \
   } while (0)

// The compute entry points exist only for device-resident operands.
#define HYPRE_AMD_REQUIRE_DEVICE(loc, what)                                      \
   do {                                                                          \
      if ((loc) != HYPRE_MEMORY_DEVICE) {                                        \
         hypre_error_w_msg(HYPRE_ERROR_GENERIC,                                  \
                           what ": operand is not in device memory; host "       \
                           "execution is not part of this library");             \
         return hypre_error_flag;                                                \
      }                                                                          \
   } while (0)

namespace hamd {

// ---------------------------------------------------------------------------
// library handle (utilities/handle.h:34-81 in the reference)
// ---------------------------------------------------------------------------
struct Handle
{
   bool                  device_ok      = false;
   bool                  device_probed  = false;
   hipStream_t           compute_stream = nullptr;
   hipStream_t           comm_stream    = nullptr;
   int                   sync_compute   = 1;       // sync at the end of public ops
   HYPRE_MemoryLocation  memory_location = HYPRE_MEMORY_DEVICE;
   HYPRE_ExecutionPolicy exec_policy     = HYPRE_EXEC_DEVICE;
   // scratch for reductions: device partials + pinned host landing zone
   double               *d_reduce      = nullptr;
   size_t                d_reduce_len  = 0;
   double               *h_reduce      = nullptr;  // pinned, 16 doubles
   int                   num_cus       = 256;
   // OpenMP thread count the hybrid Gauss-Seidel sweeps emulate (row blocks of
   // hypre_partition1D; 1 = one block per rank); set by the cycle from the solver
   int                   gs_threads    = 1;
   // mixed precision (set by the cycle from the solver): SpMV-class kernels read fp32 copies
   // of the matrix values
   bool                  fp32_values   = false;
   // Algorithmic bytes of everything launched since the last reset (hypre_amd_ByteCounters): what SURVEY 8(d) counts —
   // CSR entries at 4 bytes of index + the value width in use, row pointers, every vector operand once — and, beside it,
   // what the kernels are designed to stream (the x-staged kernel reads a 16-bit index and no column array).  Filled
   // by the launch wrappers, so the numbers follow the smoother, the precision and the rank's share actually run;
   // a replayed graph adds what its recording added.
   double                bytes_csr     = 0.0;
   double                bytes_stream  = 0.0;
};
Handle &handle();
inline void account_bytes(double csr, double stream) { Handle &h = handle(); h.bytes_csr += csr; h.bytes_stream += stream; }
inline void account_bytes(double b) { account_bytes(b, b); }
const hypre_amd_CommOps *comm_ops(MPI_Comm comm);   // nullptr for an invalid handle
bool    ensure_device();                 // lazily creates streams; false if no GPU
int     host_cpu_share();                // cores this process may use: affinity mask cut by a cgroup CPU quota
hipStream_t stream();                    // compute stream
void    maybe_sync();                    // honours hypre_SetSyncCudaCompute
double *reduce_scratch(size_t n);        // >= n doubles of device scratch

// ---------------------------------------------------------------------------
// SpMV plan: rows are binned into fixed-nnz tiles ("row blocks").  Tile b owns
// the rows whose first stored entry lies in [b*TILE, (b+1)*TILE).
// ---------------------------------------------------------------------------
#ifndef HYPRE_AMD_SPMV_TILE_NNZ          // experiment hooks: -DHYPRE_AMD_SPMV_TILE_NNZ=.. -DHYPRE_AMD_SPMV_WG=..
#define HYPRE_AMD_SPMV_TILE_NNZ 2048
#endif
#ifndef HYPRE_AMD_SPMV_WG
#define HYPRE_AMD_SPMV_WG 256
#endif
constexpr int SPMV_TILE    = HYPRE_AMD_SPMV_TILE_NNZ;   // nnz per tile (8 per lane)
constexpr int SPMV_MAXROW  = 1024;   // longest row the tiled path accepts
constexpr int SPMV_THREADS = HYPRE_AMD_SPMV_WG;
static_assert(SPMV_TILE == 8 * SPMV_THREADS, "a lane streams two quads of the tile");

struct SpmvPlan
{
   // identity of the matrix arrays the plan was built for
   const HYPRE_Int     *i = nullptr;
   const HYPRE_Int     *j = nullptr;
   const HYPRE_Complex *a = nullptr;
   int   num_rows = 0, num_cols = 0, nnz = 0;
   int   max_row_nnz = 0;
   int   max_tile_rows = 0;          // most rows any tile holds (sizes the row-pointer / row-sum LDS)
   int   num_tiles = 0;
   int  *d_tile_row = nullptr;       // [num_tiles+1] first row of every tile
   int  *d_tile_k   = nullptr;       // [num_tiles+1] Ai[tile_row[b]]
   int  *d_tile_perm = nullptr;      // [num_tiles] workgroup -> tile for band-aware XCD placement (or null)
   int   band = 0;                   // the far-coupling distance that placement was built for (0: none)
   int   prod_elems = 0;             // LDS product slots per tile: TILE + max row + pad
   bool  tiled = false;              // false -> wave-per-row kernel (long rows)
   // cached explicit transpose (built on first MatvecT)
   hypre_CSRMatrix *AT = nullptr;
   // cached strictly lower triangular part (built on the first two-stage Gauss-Seidel sweep: its inner
   // steps stream half the entries instead of masking the upper half of the full matrix)
   hypre_CSRMatrix *Lstrict = nullptr;
   // fp32 copy of the values for the mixed-precision path (lazily built)
   float *a32 = nullptr;
   // Value codes (x-staged kernel only).  A matrix that holds at most 256 distinct values — a constant-coefficient stencil
   // holds a handful — is streamed as one byte per entry plus the table of its values (sorted by bit pattern), which every
   // tile stages in LDS beside its x pieces: 1 + 2 bytes per entry instead of 8 + 2, the products bit for bit the same.
   // d_dict32 is the same table rounded through fp32 (what the mixed-precision path multiplies by).  Built with the plan
   // (spmv_value_codes(): on by default); the staleness watch compares one decoded value per tile with the fp64 original.
   unsigned char *d_codes = nullptr;   // [nnz rounded up to 16]
   double        *d_dict = nullptr, *d_dict32 = nullptr;   // [256]
   int            ndict = 0;
   // Slice form (spmv_sl_kernel): a coded matrix whose rows are short (at most 32 entries) and about equally long — a
   // stencil — is stored a third time with the codes and local indices of a ROW (sl_w = 1: rows of at most 8 entries) or
   // of every second entry of a row (sl_w = 2) in a lane's own words, entry j of the 64 lane-rows of a wave side by side:
   // a workgroup takes sl_rows = 256 / sl_w consecutive rows, stages their x pieces as the tiled kernel does, and every
   // lane sums its entries in stored order from registers — no products parked in LDS, no reduction, one barrier.
   int       sl_w = 0, sl_rows = 0, sl_k = 0;   // lanes per row; rows per workgroup; entries per lane (every row padded to sl_w * sl_k)
   int       sl_wc = 0, sl_wl = 0;               // 32-bit words per lane: ceil(sl_k / 4) of codes, ceil(sl_k / 2) of indices
   int       sl_blocks = 0, sl_launch_units = 0;
   int      *d_sl_cnt = nullptr, *d_sl_desc = nullptr;    // per block, as d_xs_cnt / d_xs_desc per tile
   int      *d_sl_k0 = nullptr, *d_sl_fp = nullptr;       // per block: position of its first entry; fingerprint of two of its columns
   int      *d_sl_perm = nullptr;                          // workgroup -> block (band-aware placement), or null
   unsigned *d_sl_data = nullptr;                          // [sl_blocks][4 waves][sl_wc + sl_wl][64 lanes]
   // Row-slice form (spmv_rs_kernel): matrices that cannot change behind their plans (owned) and are not coded — the Galerkin
   // operators of the coarse levels, 30 - 90 entries per row, every value distinct — are stored once more as JAGGED SLICES:
   // a workgroup takes rs_rows = 256 / rs_w consecutive rows, rs_w lanes per row (lane `sub` takes entries sub, sub + rs_w,
   // ... of its row); the 64 lane-tasks of a wave are sorted by their entry counts, and entry c of every task that has one
   // is stored side by side (jagged diagonals: chunk c of a wave holds as many entries as tasks are longer than c) — fp64
   // value and 16-bit staged position (byte offset) in two arrays at the same positions, NO padding: 10 bytes per entry
   // and no row pointers.  Every lane multiplies and adds its entries in stored order out of registers, the rs_w partial
   // sums of a row meet in LDS (added in lane order by the lane that finishes the row): no products parked, no tree.
   int             rs_w = 0, rs_rows = 0, rs_kp = 0;     // lanes per row; rows per workgroup; most entries a lane holds (8, 16, 24, 32: the kernel's KP)
   int             rs_blocks = 0, rs_units = 0;          // workgroups; longest staged copy of x (2-column units)
   int            *d_rs_desc = nullptr;                  // per block: x piece descriptors, as d_xs_desc per tile
   int            *d_rs_perm = nullptr;                  // workgroup -> block (band-aware placement), or null
   int            *d_rs_hdr = nullptr;                   // per wave 16 ints: position of its first entry, chunks, active lanes per chunk (bytes)
   unsigned       *d_rs_meta = nullptr;                  // per lane: slot of its partial sum | (entries << 10)
   double         *d_rs_val = nullptr;                   // [nnz] values in slice order
   float          *d_rs_val32 = nullptr;                 // the same in fp32 (mixed precision; made at the first such launch)
   unsigned       *d_rs_idx = nullptr;                   // staged positions (byte offsets) of the entries' columns, two to a word: per wave and
                                                         // pair of chunks 2p, 2p + 1 one word for every lane that has an entry of chunk 2p
   // x staging (spmv_xs_kernel): per tile the number of column segments that cover its entries (0: they do not fit, gather
   // instead) and their descriptors (2 * SPMV_XS_SEGS ints per tile); per entry the index of its column in the tile's
   // staged copy
   int            *d_xs_cnt = nullptr;
   int            *d_xs_desc = nullptr;
   unsigned short *d_lidx   = nullptr;
   int             xs_tiles = 0;       // tiles with a piece list
   int             xs_max_units = 0;   // longest staged copy of any tile, in 2-column units
   int             xs_launch_units = 0;  // what a launch stages at most (tiles above it gather): sizes the launch's LDS
   // Staleness watch.  The x-staged kernel never reads the column array and the mixed-precision kernels never read the
   // fp64 values: a plan that outlived its matrix (the caller freed it with its own hypre_CSRMatrixDestroy and a new
   // matrix of the same shape landed on the same addresses) would give a wrong product silently.  So the plan keeps a
   // fingerprint per tile (two entries of the column array), every launch compares it — and Ai[first row] against the
   // tile table, and one fp32 value against its fp64 original — for 12 bytes per 20 KB tile, and a mismatch raises
   // the flag: a word of pinned host memory the host reads without synchronising (get_plan drops a flagged plan and
   // raises HYPRE_ERROR_GENERIC; a synchronous public product then repeats itself with a fresh plan).
   int            *d_tile_fp = nullptr;  // [num_tiles]
   int            *h_stale = nullptr;    // pinned, mapped
   int            *d_stale = nullptr;    // the same word as the device sees it
   // Launch counter: the kernels that multiply by a private copy of the values (value codes, slice form) compare a ROTATING
   // sample of the copy with the caller's fp64 array — eight consecutive entries per wave, at a position that moves with
   // this counter — so that ANY in-place edit of a coefficient is found within a bounded number of launches (64 for the
   // tiled kernel, 128 for the slice kernel), not only a wholesale replacement.
   unsigned        launches = 0;
   // 64-bit checksum of the arrays the plan was built from (row pointers, columns, value bit patterns), taken when the plan
   // is built for a matrix the library does not own: plan_verify() recomputes it (one pass over the CSR arrays) — called
   // where a solve begins and by hypre_amd_CSRMatrixVerifyPlan.
   unsigned long long checksum = 0;
   bool            has_checksum = false;
   // The matrix cannot change behind the plan: the library made it (hierarchy levels below the finest, interpolation and
   // restriction operators, triangles, colour classes, cached transposes), or the caller said so
   // (hypre_amd_CSRMatrixSetImmutable).  Only such matrices get forms that keep a private copy of fp64 values (the row-slice
   // form); their launches carry no watch.
   bool            owned = false;
};
// library-owned / caller-declared immutable matrices (see SpmvPlan::owned); forgotten when the matrix is destroyed
void mark_owned(const hypre_CSRMatrix *A, bool on = true);
bool is_owned(const hypre_CSRMatrix *A);
// recompute the checksum of A's arrays and compare it with the one its plan was built with; a plan that fails is dropped
// (the next product builds a fresh one) and false returned.  No plan, or no checksum: true.
bool plan_verify(hypre_CSRMatrix *A);
// checked device allocation of the plan builders (seq_mv.cpp): false — nothing allocated, no HIP error left behind — when the
// memory is not to be had, when the request exceeds half of what is free, or when a test armed this site
enum PlanAllocSite { PLAN_SITE_TILES = 1, PLAN_SITE_XS = 2, PLAN_SITE_CODES = 3, PLAN_SITE_SLICE = 4, PLAN_SITE_ROWSLICE = 5 };
bool plan_alloc(void **ptr, size_t bytes, int site);
void plan_free(void *ptr);
// where a solve begins: the plans of the caller's matrix are verified against its arrays (plan_verify) — a plan that
// fails is dropped silently, nothing has used it since
inline void verify_par_plans(hypre_ParCSRMatrix *A)
{
   if (!A) { return; }
   if (A->diag && A->diag->memory_location == HYPRE_MEMORY_DEVICE) { (void) plan_verify(A->diag); }
   if (A->offd && A->offd->memory_location == HYPRE_MEMORY_DEVICE) { (void) plan_verify(A->offd); }
}
unsigned long long device_csr_checksum(const int *Ai, const int *Aj, const double *Aa, int n, int nnz, hipStream_t s);
void bump_plan_generation();
// A sampled fingerprint of a device CSR matrix — row pointers, columns and, if asked, values at 4096 positions spread over
// it — kept on the device, and a pinned flag that a checking launch raises when the matrix no longer matches.  For the
// caches that pre-digest a matrix and are found again by its address (colour classes: a colour-sorted COPY of the values;
// level schedules of the hybrid sweeps: the dependency levels of the pattern): the same exposure as the SpMV plan's — the
// caller frees the matrix with hypre's own destroy routine, the next one lands on the same addresses.
constexpr int MATRIX_FP_WORDS = 16;       // partial fingerprints (one per workgroup of the checking launch)
struct MatrixWatch
{
   unsigned long long *d_fp = nullptr;    // [MATRIX_FP_WORDS]
   int *h_stale = nullptr, *d_stale = nullptr;
};
void watch_record(MatrixWatch &w, const hypre_CSRMatrix *A, bool with_values, hipStream_t s);
void watch_check(const MatrixWatch &w, const hypre_CSRMatrix *A, bool with_values, hipStream_t s);    // one small launch
bool watch_flagged(const MatrixWatch &w);
void watch_release(MatrixWatch &w);
int *take_stale_slot(int **device_view);      // a word of pinned, device-visible host memory (nullptr: none to be had)
void give_stale_slot(int *slot);
void launch_matrix_fingerprint(const int *Ai, const int *Aj, const double *Aa, int n, int nnz, unsigned long long *fp, int *stale,
                               int record, hipStream_t s);
unsigned long long plan_generation();    // bumped whenever a plan (or a colour plan) is freed: recorded graphs watch it
bool plan_is_stale(const hypre_CSRMatrix *A);   // the plan of A, if one exists, was flagged by a kernel
void launch_build_fp(const HYPRE_Int *Aj, int nnz, int num_tiles, int *fp, hipStream_t s);
SpmvPlan *get_plan(hypre_CSRMatrix *A);
hypre_CSRMatrix *strict_lower_of(hypre_CSRMatrix *A);   // device CSR of {a_ij : j < i}, cached in A's plan
void      drop_plan(hypre_CSRMatrix *A);
const float *fp32_values_of(hypre_CSRMatrix *A);         // fp32 copy of the values, cached in A's plan (mixed precision)
void      set_strict_lower(hypre_CSRMatrix *A, hypre_CSRMatrix *L);   // A's plan takes ownership of L

// epilogue selector of the tiled SpMV family
enum SpmvOp
{
   OP_AXPBY   = 0,  // y = alpha*(A x) + beta*b
   OP_JACOBI  = 1,  // y = x + w*(b - A x)./d          (b=f, d=l1 or diag vector)
   OP_JACOBI_CF = 2, // same, rows with marker!=pts copy x
   OP_TSGS    = 3,  // y = (A_fill x)./d ; aux += alpha*y   (two-stage GS inner step)
   OP_JACOBI_MAP = 4, // OP_JACOBI on the rows rowmap[] names (marker optional): sweeps over one colour's rows
   OP_AXPBY_DIV = 5, // OP_AXPBY and aux = (scale2 * y) ./ d: the restriction that also starts the coarse level's sweep from zero
   OP_RESID_RD = 6,  // y = (alpha*(A x) + beta*b) .* (1 ./ d): the two-stage sweep's residual already scaled by the diagonal
   OP_TSGS_FIRST = 7 // first inner step of the two-stage sweep: z' = (A_fill z) .* (1 ./ d) ; aux = (aux_or_0 + z) + alpha*z'  (beta != 0: aux is read)
};

struct SpmvArgs
{
   const HYPRE_Int     *Ai;
   const HYPRE_Int     *Aj;
   const HYPRE_Complex *Aa;
   const float         *Aa32;     // non-null selects fp32 matrix values
   const unsigned char *Ac8;      // x-staged kernel: non-null selects value codes (filled from the plan by launch_spmv) ...
   const double        *dict;     // ... and the table they index (ndict entries; allocated and readable up to 256)
   int                  ndict;
   int                  dict_rounded;   // the table was rounded through fp32 (mixed precision)
   const HYPRE_Complex *x;
   const HYPRE_Complex *b;        // may be null when beta == 0
   HYPRE_Complex       *y;
   HYPRE_Complex       *aux;      // second output of the OP_TSGS epilogue
   const HYPRE_Complex *d;        // diagonal / l1 norms for the relax epilogues
   const HYPRE_Int     *marker;   // CF marker or null
   int                  marker_val;
   HYPRE_Complex        alpha, beta;
   HYPRE_Complex        scale2;   // OP_AXPBY_DIV: aux[row] = (scale2 * y[row]) / d[row] as well
   int                  fill;     // HYPRE_SPMV_FILL_*
   int                  row_offset;
   int                  last_quad;   // (nnz - 1) & ~3: last 16-byte quad of the (col, val) arrays holding an entry
   int                  x_last;      // num_cols - 1: largest valid index into x
   int                  reduce_w;      // lanes per row of the tile reduction: 0 = per tile (all its rows in one pass), else fixed (power of two)
   int                  gather_t;    // x gathers paired with consecutive entries per wave (columns transposed through LDS)
   int                  xcd_map;     // tile -> XCD placement: 0 dispatch order, C > 0 chunks of C tiles, < 0 contiguous eighths
   const int           *tile_perm;   // workgroup -> tile table (overrides xcd_map), or null
   const int           *tile_fp;     // per-tile fingerprint of the column array the plan was built from, or null
   int                 *stale;       // raised (system scope) by a tile whose fingerprint / row pointer / fp32 value disagrees
   const int           *rowmap;      // epilogue row indirection (multicolour sweeps: row r of the matrix is row rowmap[r] of
                                     // the vectors b, d, x, y, marker), or null
   int                  variant;     // 0: x gathered through the cache; 2: x staged through LDS from the plan's chunk lists
   int                  use_rs;      // the row-slice kernel serves this launch (decided by launch_spmv)
   unsigned             rot;         // launch counter of the plan (filled by launch_spmv): position of the rotating value check
   int                  nnz;         // entries of the matrix (filled by launch_spmv)
};
void spmv_default_flags(SpmvArgs &a);   // fills gather_t / xcd_map from the tuning knobs

void launch_spmv(const SpmvPlan *plan, const SpmvArgs &args, SpmvOp op, hipStream_t s);
// y = M x and, in the same pass, u = (w y) ./ d  (seq_mv.cpp): the restriction f_c = P^T r fused with the zero-guess Jacobi
// sweep u_c = w f_c ./ d_c that follows it on the coarse level; false: not served (empty matrix), nothing was launched
bool spmv_with_scaled_quotient(hypre_CSRMatrix *M, const double *x, double *y, double w, const double *d, double *u);
// y(:, v) = alpha A x(:, v) + beta b(:, v), v < nv, in one pass over the matrix (columns xstride / bstride / ystride doubles
// apart); false: not served by this plan or these operands, nothing launched — the caller loops over the columns
bool launch_spmv_mv(const SpmvPlan *plan, const SpmvArgs &args, int nv, long xstride, long bstride, long ystride, hipStream_t s);
long &spmv_mv_launches();               // fused launches so far (tests)
bool &spmv_fused_multivectors();       // default on; HYPRE_AMD_SPMV_FUSED_MV=0 / hypre_amd_SpmvSetFusedMultivectors
void launch_spmv_rownnz(const HYPRE_Int *rownnz, int num_rownnz, const SpmvArgs &args, hipStream_t s);
void launch_spmv_allrows_update(int num_rows, const SpmvArgs &args, hipStream_t s);
// value codes of a matrix (nullptr / 0 when it holds more than 256 distinct values): codes[nnz], the sorted table, its fp32-rounded twin
bool device_value_codes(const double *Aa, size_t nnz, unsigned char **codes_out, double **dict_out, double **dict32_out, int *ndict_out,
                        hipStream_t s);
// slice form of a coded matrix (false: not applicable — long or very unequal rows, a block that cannot be staged)
bool device_build_slice_form(SpmvPlan *p, const hypre_CSRMatrix *A, hipStream_t s);
// row-slice form of an owned, uncoded matrix (false: not applicable, or a table was not to be had: the tiles serve)
bool device_build_row_slices(SpmvPlan *p, const hypre_CSRMatrix *A, hipStream_t s);
int  &spmv_row_slices();              // 0 off, 1 matrices the library owns (default), 2 every matrix (tests); HYPRE_AMD_SPMV_ROW_SLICES
bool &spmv_slice_form();               // plans built from now on get the slice form where it applies (default: on; HYPRE_AMD_SPMV_SLICE_FORM=0)
bool &spmv_value_codes();              // plans built from now on look for value codes (default: on; HYPRE_AMD_SPMV_VALUE_CODES=0)
void launch_build_tiles(const HYPRE_Int *Ai, int num_rows, int nnz, int num_tiles, int *d_tile_row,
                        int *d_tile_k, hipStream_t s);
void launch_build_xs(const HYPRE_Int *Aj, const int *d_tile_k, int num_tiles, int *xs_cnt, int *xs_desc, unsigned short *lidx,
                     hipStream_t s);
constexpr int SPMV_XS_SEGS = 48;     // pieces of x (<= 128 doubles each) a staged tile may have
constexpr int SPMV_XS_CAP  = 4096;   // doubles a tile may stage at most
int  device_max_row_nnz(const HYPRE_Int *Ai, int num_rows, hipStream_t s);
// largest |col - row| of `nsamples` evenly spaced rows, copied to the host (structure probe of the plan builder)
void sample_row_bands(const HYPRE_Int *Ai, const HYPRE_Int *Aj, int num_rows, int nsamples, int *host_out, hipStream_t s);

// BLAS-1 kernels
void launch_set(double *y, double v, size_t n, hipStream_t s);
void launch_copy(double *y, const double *x, size_t n, hipStream_t s);
void launch_scale(double *y, double a, size_t n, hipStream_t s);
void launch_axpy(double a, const double *x, double *y, size_t n, hipStream_t s);
void launch_axpyz(double a, const double *x, double b, const double *y, double *z, size_t n, hipStream_t s);
void launch_elmdivpy(const double *x, const double *d, double *y, const int *marker, int mval,
                     size_t n, hipStream_t s);
void launch_scaled_div(double w, const double *f, const double *d, double *u, const int *marker,
                       int mval, size_t n, hipStream_t s);   // u = w*f./d (zero-guess Jacobi)
void launch_diag_first_scale(const int *Ai, const double *Aa, const double *y, double *x, size_t n, int nv, size_t ys, size_t xs, hipStream_t s);
void launch_scaled_recip(double w, const double *f, const double *d, double *z, size_t n, hipStream_t s);   // z = (w*f) .* (1 ./ d)
void launch_diagscale2(const double *diag, const double *x, double beta, double *y, double *z,
                       int computeY, size_t n, hipStream_t s);
void launch_dot(const double *x, const double *y, size_t n, double *d_out, hipStream_t s);
void launch_gather(const double *x, const int *idx, double *out, size_t n, hipStream_t s);
void launch_scatter_add(const double *in, const int *idx, double *y, size_t n, hipStream_t s);
void launch_f64_to_f32(const double *x, float *y, size_t n, hipStream_t s);
void launch_transpose(const int *Ai, const int *Aj, const double *Aa, int nrows, int ncols, int nnz, int *Ti, int *tj, double *ta,
                      hipStream_t s);     // device CSR transpose, rows of the result in ascending source-row order
void launch_scan_exclusive(int *data, int n, hipStream_t s);
void launch_sort_rows(const int *Ai, int *Aj, double *Aa, int n, int keep_first, hipStream_t s);   // columns ascending inside every row
// multicolour Gauss-Seidel on the device (mc_kernels.hip, par_relax_mc.cpp)
void preload_mc_kernels();
void prepare_mc_plan(hypre_CSRMatrix *A);
int  device_greedy_coloring(int n, const int *Ai, const int *Aj, const int *Ti, const int *Tj, int *color, int round_budget, hipStream_t s,
                            int *rounds_out = nullptr);
void device_color_order(int n, int C, const int *color, int **order_out, std::vector<int> &cstart, hipStream_t s);
void device_color_matrices(int n, int C, const int *color, const int *order, const std::vector<int> &cstart, const int *Ai, const int *Aj,
                           const double *Aa, int **ptr_out, int **Cj_out, double **Ca_out, std::vector<int> &slice0, std::vector<int> &slice_nnz,
                           hipStream_t s);
void launch_mc_diag(int n, const int *Ai, const double *Aa, double *d, hipStream_t s);
void launch_mc_rowinfo(int n, const int *order, const int *Ai, void *info, hipStream_t s);
void launch_mc_small_sweep_ahead(int num_colors, int direction, const int *cstart, const void *rowinfo, int nrows, const int *Aj,
                                 const double *Aa, const float *Aa32, int nnz_matrix, const double *f, const double *d, const int *marker,
                                 int marker_val, double w, double *u, int n, int nnz, int r_first, hipStream_t s);
void launch_mc_small_sweep(int num_colors, int direction, const int *cstart, const int *order, const int *Ai, const int *Aj,
                           const double *Aa, const float *Aa32, const double *f, const double *d, const int *marker, int marker_val,
                           double w, double *u, int n, int nnz, hipStream_t s);
// one per kernel file: loads its code object (runtime.cpp: ensure_device)
void preload_cheby_kernels(); void preload_gs_kernels(); void preload_interp_kernels(); void preload_vector_kernels();
void preload_rap_kernels(); void preload_setup_kernels(); void preload_spmv_kernels();
// setup_kernels.hip: strength of connection, PMIS, coarse numbering and smoother diagonals of a single-rank level
void device_strength(int n, const int *Ai, const int *Aj, const double *Aa, double theta, double max_row_sum,
                     int **Si_out, int **Sj_out, int *nnz_out, hipStream_t s);
int  device_pmis(int n, const int *Si, const int *Sj, int snnz, unsigned seed, unsigned long long skip, int *CF, hipStream_t s);
int  device_coarse_numbering(int n, const int *CF, int *f2c, hipStream_t s);
bool device_l1_norms(int n, const int *Ai, const int *Aj, const double *Aa, int option, const int *cf, double *out, hipStream_t s);
// Galerkin product R A P on the device, bit-identical to the host setup's (rap_kernels.hip); false: does not fit, use the host
bool device_rap(int nc, int ncP, int maxP, const int *Ri, const int *Rj, const double *Ra, const int *Ai, const int *Aj,
                const double *Aa, const int *Pi, const int *Pj, const double *Pa, int **Ci_out, int **Cj_out, double **Ca_out,
                int *nnz_out, hipStream_t s);
// extended+i interpolation rows on the device, bit-identical to the host setup's (interp_kernels.hip); false: use the host
bool device_extpi(int n, const int *Ai, const int *Aj, const double *Aa, const int *Si, const int *Sj, const int *CF,
                  const int *f2c, double trunc_tol, int max_elmts, int first_rung, int **Pi_out, int **Pj_out, double **Pa_out,
                  int *nnz_out, hipStream_t s, int split = -1, int **Oi_out = nullptr, int **Oj_out = nullptr,
                  double **Oa_out = nullptr, int *onnz_out = nullptr);
// ---- distributed levels on the device (dist_setup_kernels.hip, setup_kernels.hip, rap_kernels.hip): the single-rank kernels
// on the extended numbering [local points | ghost points]
void preload_dist_setup_kernels();
void device_extend_csr(int n, const int *Di, const int *Dj, const double *Da, const int *Oi, const int *Oj, const double *Oa, int ooff,
                       int ng, const int *Gi, const int *Gj, const double *Ga, int **Ei_out, int **Ej_out, double **Ea_out,
                       int *nnz_out, hipStream_t s);
void device_extract_rows(int filter, int tot, const int *d_elmts, const int *Di, const int *Dj, const double *Da,
                         const int *Oi, const int *Oj, const double *Oa, long long first_col, const long long *d_cmap, const int *CF,
                         std::vector<int> &hi, std::vector<long long> &hj, std::vector<double> &ha, hipStream_t s);
void device_extract_S_rows(int tot, const int *d_elmts, int n, const int *Si, const int *Sj, long long first_col, const long long *d_cmap,
                           const int *CFext, std::vector<int> &hi, std::vector<long long> &hj, hipStream_t s);
void device_split_strided(int n, int stride, int *len, int *nd, const int *sj, const double *sa, int split,
                          int **Di_out, int **Dj_out, double **Da_out, int *dnnz, int **Oi_out, int **Oj_out, double **Oa_out, int *onnz,
                          hipStream_t s);
void device_split_pattern(int n, const int *Si, const int *Sj, int split, int **Di_out, int **Dj_out, int *dnnz,
                          int **Oi_out, int **Oj_out, int *onnz, hipStream_t s);
void launch_mark_used(const int *j, size_t nnz, int *used, hipStream_t s);
void launch_renumber(int *j, size_t nnz, int split, const int *map, hipStream_t s);
void launch_gather_int(const int *x, const int *idx, int *out, size_t n, hipStream_t s);
bool device_l1_norms_blocks(int n, const int *Di, const int *Dj, const double *Da, const int *Oi, const int *Oj, const double *Oa,
                            int option, const int *cf, const int *cfo, double *out, hipStream_t s);
void device_strength_blocks(int n, const int *Di, const int *Dj, const double *Da, const int *Oi, const int *Oj, const double *Oa,
                            double theta, double max_row_sum, int **Si_out, int **Sj_out, int *nnz_out, hipStream_t s);
int  device_pmis_dist(int n, int nco, const int *Si, const int *Sj, int snnz, unsigned seed, unsigned long long skip,
                      hypre_ParCSRCommPkg *pkg, const int *d_elmts, const int *d_prev, MPI_Comm comm, int *CF, hipStream_t s);
bool device_rap_dist(bool direct, int nc, int square, int ncols_out, int maxP, int max_seed,
                     const int *Ri, const int *Rj, const double *Ra, const int *Ai, const int *Aj, const double *Aa,
                     const int *A2i, const int *A2j, const double *A2a, int a2_off, int nfine,
                     const int *Pi, const int *Pj, const double *Pa,
                     const int *Fi, const int *Fj, const int *Xi, const int *Xj, const double *Xa, int split,
                     int **Di_out, int **Dj_out, double **Da_out, int *dnnz, int **Oi_out, int **Oj_out, double **Oa_out, int *onnz,
                     hipStream_t s);
void launch_deinterleave(const double *in, double *out, int n, int nv, hipStream_t s);   // [entry][component] -> column by column
void launch_interleave(const double *in, double *out, int n, int nv, hipStream_t s);
void launch_count_lower(const HYPRE_Int *Ai, const HYPRE_Int *Aj, int n, int *cnt, hipStream_t s);
void launch_fill_lower(const HYPRE_Int *Ai, const HYPRE_Int *Aj, const double *Aa, const HYPRE_Int *Li, HYPRE_Int *Lj,
                       double *La, int n, hipStream_t s);
// fused PCG vector updates: x += a p, r += na s, *d_out = <r, r> of the new r; p = beta p + s
void launch_pcg_update(double a, double na, const double *p, const double *s, double *x, double *r, size_t n,
                       double *d_out, hipStream_t stream);
void launch_pcg_direction(double beta, const double *s, double *p, size_t n, hipStream_t stream);
double global_sum(MPI_Comm comm, double v);     // scalar all-reduce over a communicator (identity on one rank)
void   dev_allreduce_sum(MPI_Comm comm, double *d_buf, int n);                 // in place, device memory, stream-ordered
void   dev_global_sums(MPI_Comm comm, double *d_vals, int n, double *h_out);   // n <= 8 device partials -> global sums on the host
void launch_scale_copy(double b, const double *x, double *y, size_t n, hipStream_t s);   // y = b*x

}  // namespace hamd
