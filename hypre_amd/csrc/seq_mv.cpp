// hypre_amd — local CSR matrix / dense vector objects and the sequential
// (per-rank) matrix-vector and BLAS-1 entry points.
//
// Reference counterparts:
//   seq_mv/csr_matrix.c:25-160,350-420,917-1040   object life cycle, rownnz, migrate, clone
//   seq_mv/csr_matop.c:1043-1300,1537-1604        transpose, diagonal-first reorder
//   seq_mv/csr_matvec.c:860-901,1142-1171         Matvec dispatchers
//   seq_mv/csr_matvec_device.c:37-172             device wrapper
//   seq_mv/csr_spmv_device.c:381-557              SpMV with triangular fill modes
//   seq_mv/vector.c / vector_device.c             BLAS-1
#include "internal.hpp"
#include <unordered_map>
#include <unordered_set>
#include <omp.h>
#include <algorithm>

using namespace hamd;

// ===========================================================================
// plan cache
// ===========================================================================
namespace hamd {

// kernel variant of the tiled family (hypre_amd_SpmvSetVariant, or HYPRE_AMD_SPMV_VARIANT read once): 2 = x staged through
// LDS (spmv_xs_kernel; plans carry the chunk lists), 0 = x gathered through the cache (spmv_tiled_kernel)
struct SpmvVariant { int variant; };
static SpmvVariant &spmv_variant()
{
   static SpmvVariant v = {-1};
   if (v.variant < 0)
   {
      const char *e = getenv("HYPRE_AMD_SPMV_VARIANT");
      v.variant = e ? atoi(e) : 2;
   }
   return v;
}

// Tuning knobs (A/B-testable without rebuilding): HYPRE_AMD_SPMV_GT, HYPRE_AMD_SPMV_XCD.
void spmv_default_flags(SpmvArgs &a)
{
   static int gt = -1, xcd = 8;
   if (gt < 0)
   {
      const char *e = getenv("HYPRE_AMD_SPMV_GT");
      // measured on MI355X, 256^3 hierarchy: level 0 (7/row) 4.82 -> 5.01 TB/s, level 1 (29/row)
      // 3.76 -> 4.15 TB/s, level 2 (70/row) 3.17 -> 3.41 TB/s: fewer cache lines per gather instruction
      gt = e ? atoi(e) : 1;
      e = getenv("HYPRE_AMD_SPMV_XCD");
      // workgroup g runs on XCD g % 8 (own L2): handing every XCD runs of 8 consecutive tiles keeps rows that
      // are neighbours in the matrix in one L2.  Measured on the 256^3 hierarchy: fine-level y = A x unchanged,
      // restriction (P^T, scattered fine-vector gathers) 0.185 -> 0.167 ms, whole V-cycle 2.849 -> 2.821 ms;
      // chunks of 4..16 the same, 128 hurts the small levels, one contiguous eighth per XCD is 8 % slower.
      xcd = e ? atoi(e) : 8;
   }
   a.gather_t = gt; a.xcd_map = xcd;
   a.variant = spmv_variant().variant;
   // Lanes per row of the multi-lane reduction: 0 = chosen per tile (spmv_kernels.hip: tile_reduce), else fixed.
   { static int w = -1; if (w < 0) { const char *e = getenv("HYPRE_AMD_SPMV_REDUCE_W"); w = e ? atoi(e) : 0; } a.reduce_w = w; }
}

// Band-aware XCD placement.  Workgroup g runs on XCD g % 8 and every XCD has its own L2.  A matrix from a
// structured grid couples row i to rows i +- B (the next grid plane, B rows away): with tiles dealt to the XCDs in
// runs, the three planes that share an x value are streamed by different XCDs and every XCD fetches its own copy
// (measured on the 256^3 7-point operator: 2.0 GB of L2 misses per product against 1.74 GB of compulsory traffic).
// When the far couplings of sampled rows agree on one distance B, the rows are cut into 8 slabs by (row mod B) and
// XCD c is handed the tiles of slab c, plane after plane: the x values it fetched as the upper neighbours of plane p
// are still in its L2 when plane p + 1 and p + 2 read them again.  Irregular matrices (coarse levels, interpolation)
// do not pass the agreement test and keep the run-of-8 placement.  HYPRE_AMD_SPMV_BAND=0 switches it off.
// Policy (hypre_amd_SpmvSetBandPolicy, or HYPRE_AMD_SPMV_BAND / HYPRE_AMD_SPMV_BAND_MIN_TILES / HYPRE_AMD_SPMV_BAND_FORCE
// read once): `min_tiles` is the size below which the table is not worth building; `force` also drops the lower bound
// on the slab size, so that small test matrices go through the same code as the benchmark's.
struct BandPolicy { int enabled, min_tiles, force; };
static BandPolicy &band_policy()
{
   static BandPolicy bp = {-1, 2048, 0};
   if (bp.enabled < 0)
   {
      const char *e = getenv("HYPRE_AMD_SPMV_BAND");
      bp.enabled = e ? atoi(e) : 1;
      if ((e = getenv("HYPRE_AMD_SPMV_BAND_MIN_TILES"))) { bp.min_tiles = atoi(e); }
      if ((e = getenv("HYPRE_AMD_SPMV_BAND_FORCE"))) { bp.force = atoi(e); }
   }
   return bp;
}

// ---- checked allocations of the plan builders.  A plan is an accelerator, never a necessity: every table it holds has a
// form below it that does without (slice form -> coded tiles -> fp64 stream with staged x -> x gathered through the cache ->
// a wave per row, which needs nothing).  So an allocation that fails — no memory, more than the plan's share of what is
// free, or a test's request (hypre_amd_PlanTestFailAlloc) — frees what its step had obtained, leaves no HIP error behind
// and the plan goes on one form lower.
int g_fail_site = 0, g_fail_nth = 0;
bool plan_alloc(void **ptr, size_t bytes, int site)
{
   *ptr = nullptr;
   if (g_fail_site == site && g_fail_nth > 0 && --g_fail_nth == 0) { g_fail_site = 0; return false; }
   // no single table of a plan may take more than half of what is free (a plan lives beside the matrix it serves)
   size_t free_b = 0, total_b = 0;
   if (bytes > ((size_t) 64 << 20))
   {
      if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void) hipGetLastError(); return false; }
      if (bytes > free_b / 2) { return false; }
   }
   if (hipMalloc(ptr, bytes ? bytes : 16) != hipSuccess) { (void) hipGetLastError(); *ptr = nullptr; return false; }
   return true;
}
void plan_free(void *ptr) { if (ptr) { HIP_CHECK(hipFree(ptr)); } }

// ---- matrices that cannot change behind their plans (SpmvPlan::owned)
static std::unordered_set<const hypre_CSRMatrix *> &owned_table()
{
   static std::unordered_set<const hypre_CSRMatrix *> t;
   return t;
}
void mark_owned(const hypre_CSRMatrix *A, bool on)
{
   if (!A) { return; }
   if (on) { owned_table().insert(A); } else { owned_table().erase(A); }
}
bool is_owned(const hypre_CSRMatrix *A) { return A && owned_table().count(A) != 0; }

static void build_band_placement(SpmvPlan *p, const hypre_CSRMatrix *A, hipStream_t s)
{
   const BandPolicy &bp = band_policy();
   if (!bp.enabled || p->num_tiles < bp.min_tiles || p->num_tiles < 8 || A->num_rows != A->num_cols) { return; }
   const int ns = 1024;
   std::vector<int> far((size_t) ns);
   sample_row_bands(A->i, A->j, A->num_rows, ns, far.data(), s);
   std::vector<int> sorted(far);
   std::sort(sorted.begin(), sorted.end());
   const int B = sorted[(size_t) ns / 2];
   int agree = 0;
   for (int f : far) { if (std::abs(f - B) * 50 <= B) { agree++; } }
   const long long rows_per_tile = std::max(1LL, (long long) A->num_rows / p->num_tiles);
   // one distance for most rows, at least four planes, slabs of at least 8 tiles
   if (B <= 0 || agree * 10 < ns * 6 || 4LL * B > A->num_rows) { return; }
   if (!bp.force && (long long) B < 8 * 8 * rows_per_tile) { return; }
   std::vector<int> trow((size_t) p->num_tiles + 1);
   HIP_CHECK(hipMemcpyAsync(trow.data(), p->d_tile_row, sizeof(int) * ((size_t) p->num_tiles + 1), hipMemcpyDeviceToHost, s));
   HIP_CHECK(hipStreamSynchronize(s));
   std::vector<std::vector<int>> cls(8);
   for (int t = 0; t < p->num_tiles; t++)
   {
      const int c = (int) (((long long) (trow[(size_t) t] % B) * 8) / B);
      cls[(size_t) std::min(std::max(c, 0), 7)].push_back(t);
   }
   // workgroup 8k + c -> k-th tile of slab c; slabs that run out borrow from the tail of the longest one
   std::vector<int> perm((size_t) p->num_tiles, -1);
   std::vector<size_t> next(8, 0);
   std::vector<int> leftovers;
   for (int g = 0; g < p->num_tiles; g++)
   {
      const size_t c = (size_t) (g & 7);
      if (next[c] < cls[c].size()) { perm[(size_t) g] = cls[c][next[c]++]; }
   }
   for (size_t c = 0; c < 8; c++) { for (size_t k = next[c]; k < cls[c].size(); k++) { leftovers.push_back(cls[c][k]); } }
   size_t lo = 0;
   for (int g = 0; g < p->num_tiles; g++) { if (perm[(size_t) g] < 0) { perm[(size_t) g] = leftovers[lo++]; } }
   if (!plan_alloc((void **) &p->d_tile_perm, sizeof(int) * (size_t) p->num_tiles, PLAN_SITE_TILES)) { return; }     // speed only: runs of 8 instead
   HIP_CHECK(hipMemcpyAsync(p->d_tile_perm, perm.data(), sizeof(int) * (size_t) p->num_tiles, hipMemcpyHostToDevice, s));
   HIP_CHECK(hipStreamSynchronize(s));
   p->band = B;
}

constexpr int RP_CAP_HOST = 640;     // = RP_CAP of spmv_kernels.hip

static std::unordered_map<const hypre_CSRMatrix *, SpmvPlan *> &plan_table()
{
   static std::unordered_map<const hypre_CSRMatrix *, SpmvPlan *> t;
   return t;
}

static unsigned long long g_plan_generation = 1;
unsigned long long plan_generation() { return g_plan_generation; }
void bump_plan_generation() { g_plan_generation++; }

// words of pinned, device-visible host memory for the plans' staleness flags, handed out from pages of 1024
namespace {
struct StaleSlots
{
   std::vector<int *> pages;
   std::vector<int *> free_list;
   int used_in_last = 1024;
   int *take()
   {
      if (!free_list.empty()) { int *p = free_list.back(); free_list.pop_back(); *p = 0; return p; }
      if (used_in_last >= 1024)
      {
         int *page = nullptr;
         if (hipHostMalloc((void **) &page, sizeof(int) * 1024, hipHostMallocMapped) != hipSuccess) { (void) hipGetLastError(); return nullptr; }
         memset(page, 0, sizeof(int) * 1024);
         pages.push_back(page);
         used_in_last = 0;
      }
      return pages.back() + used_in_last++;
   }
   void give(int *p) { if (p) { free_list.push_back(p); } }
};
StaleSlots &stale_slots() { static StaleSlots s; return s; }
}  // namespace

int *take_stale_slot(int **device_view)
{
   int *h = stale_slots().take();
   int *d = nullptr;
   if (h && hipHostGetDevicePointer((void **) &d, h, 0) != hipSuccess) { (void) hipGetLastError(); d = nullptr; }
   if (!d) { stale_slots().give(h); h = nullptr; }
   if (device_view) { *device_view = d; }
   return h;
}
void give_stale_slot(int *slot) { stale_slots().give(slot); }

void watch_record(MatrixWatch &w, const hypre_CSRMatrix *A, bool with_values, hipStream_t s)
{
   watch_release(w);
   w.h_stale = take_stale_slot(&w.d_stale);
   if (!w.h_stale) { return; }
   HIP_CHECK(hipMalloc((void **) &w.d_fp, sizeof(unsigned long long) * MATRIX_FP_WORDS));
   launch_matrix_fingerprint(A->i, A->j, with_values ? A->data : nullptr, A->num_rows, A->num_nonzeros, w.d_fp, w.d_stale, 1, s);
}
void watch_check(const MatrixWatch &w, const hypre_CSRMatrix *A, bool with_values, hipStream_t s)
{
   if (!w.d_fp) { return; }
   launch_matrix_fingerprint(A->i, A->j, with_values ? A->data : nullptr, A->num_rows, A->num_nonzeros, w.d_fp, w.d_stale, 0, s);
}
bool watch_flagged(const MatrixWatch &w) { return w.h_stale && __atomic_load_n(w.h_stale, __ATOMIC_RELAXED) != 0; }
void watch_release(MatrixWatch &w)
{
   if (w.d_fp) { HIP_CHECK(hipFree(w.d_fp)); }
   give_stale_slot(w.h_stale);
   w = MatrixWatch();
}

static void free_plan(SpmvPlan *p)
{
   if (!p) { return; }
   g_plan_generation++;
   if (p->d_tile_fp) { HIP_CHECK(hipFree(p->d_tile_fp)); }
   stale_slots().give(p->h_stale);
   if (p->d_tile_row) { HIP_CHECK(hipFree(p->d_tile_row)); }
   if (p->d_tile_k) { HIP_CHECK(hipFree(p->d_tile_k)); }
   if (p->d_tile_perm) { HIP_CHECK(hipFree(p->d_tile_perm)); }
   if (p->d_xs_cnt) { HIP_CHECK(hipFree(p->d_xs_cnt)); }
   if (p->d_xs_desc) { HIP_CHECK(hipFree(p->d_xs_desc)); }
   if (p->d_lidx) { HIP_CHECK(hipFree(p->d_lidx)); }
   if (p->a32) { HIP_CHECK(hipFree(p->a32)); }
   if (p->d_codes) { HIP_CHECK(hipFree(p->d_codes)); }
   if (p->d_dict) { HIP_CHECK(hipFree(p->d_dict)); }       // d_dict32 is its second half
   if (p->d_sl_cnt) { HIP_CHECK(hipFree(p->d_sl_cnt)); }
   if (p->d_sl_desc) { HIP_CHECK(hipFree(p->d_sl_desc)); }
   if (p->d_sl_k0) { HIP_CHECK(hipFree(p->d_sl_k0)); }
   if (p->d_sl_fp) { HIP_CHECK(hipFree(p->d_sl_fp)); }
   if (p->d_sl_perm) { HIP_CHECK(hipFree(p->d_sl_perm)); }
   if (p->d_sl_data) { HIP_CHECK(hipFree(p->d_sl_data)); }
   for (void *q : {(void *) p->d_rs_desc, (void *) p->d_rs_perm, (void *) p->d_rs_hdr, (void *) p->d_rs_meta, (void *) p->d_rs_val,
                   (void *) p->d_rs_val32, (void *) p->d_rs_idx}) { plan_free(q); }
   if (p->AT) { hypre_CSRMatrixDestroy(p->AT); }
   if (p->Lstrict) { hypre_CSRMatrixDestroy(p->Lstrict); }
   delete p;
}

void drop_gs_schedule(const hypre_CSRMatrix *A);    // par_relax_gs.cpp
void drop_mc_plan(const hypre_CSRMatrix *A);        // par_relax_mc.cpp

void drop_plan(hypre_CSRMatrix *A)
{
   drop_gs_schedule(A);
   drop_mc_plan(A);
   auto &t = plan_table();
   auto it = t.find(A);
   if (it != t.end())
   {
      free_plan(it->second);
      t.erase(it);
   }
}

// tile bounds of the tiled kernel family (PLAN_SITE_TILES).  false: the tables are not to be had; a wave per row serves.
static bool build_tile_tables(SpmvPlan *p, const hypre_CSRMatrix *A, hipStream_t s)
{
   p->num_tiles = (int) (((long long) A->num_nonzeros + SPMV_TILE - 1) / SPMV_TILE);
   if (!plan_alloc((void **) &p->d_tile_row, sizeof(int) * (size_t) (p->num_tiles + 1), PLAN_SITE_TILES) ||
       !plan_alloc((void **) &p->d_tile_k, sizeof(int) * (size_t) (p->num_tiles + 1), PLAN_SITE_TILES))
   {
      plan_free(p->d_tile_row); plan_free(p->d_tile_k);
      p->d_tile_row = p->d_tile_k = nullptr;
      p->num_tiles = 0;
      return false;
   }
   p->prod_elems = SPMV_TILE + ((p->max_row_nnz + 3) & ~3) + 8;
   launch_build_tiles(A->i, A->num_rows, A->num_nonzeros, p->num_tiles, p->d_tile_row, p->d_tile_k, s);
   p->max_tile_rows = device_max_row_nnz(p->d_tile_row, p->num_tiles, s);     // max over tiles of tile_row[b+1] - tile_row[b]
   // staleness flag (see SpmvPlan): a word of pinned host memory, or — none to be had — somewhere for the kernels to write
   p->h_stale = stale_slots().take();
   if (p->h_stale && hipHostGetDevicePointer((void **) &p->d_stale, p->h_stale, 0) != hipSuccess) { (void) hipGetLastError(); p->d_stale = nullptr; }
   if (!p->d_stale)
   {
      static int *sink = nullptr;
      if (!sink && hipMalloc((void **) &sink, sizeof(int)) != hipSuccess) { (void) hipGetLastError(); sink = nullptr; }
      if (sink) { HIP_CHECK(hipMemsetAsync(sink, 0, sizeof(int), s)); }
      p->d_stale = sink;
      stale_slots().give(p->h_stale); p->h_stale = nullptr;
   }
   return true;
}

// x staging of the tiled kernel (PLAN_SITE_XS: fingerprints, piece lists, local indices), then — each with a site of its
// own, each optional — value codes and the slice form.  On failure the plan keeps what the step below needs: without the
// staging tables the tiles gather x through the cache.
static void build_staging(SpmvPlan *p, const hypre_CSRMatrix *A, hipStream_t s)
{
   // value codes first: a coded matrix (a stencil) has better forms than any fp64 one, and the table the codes stage is
   // part of the launch's LDS, which decides how much x a launch stages
   if (spmv_value_codes() && p->d_stale) { device_value_codes(A->data, (size_t) A->num_nonzeros, &p->d_codes, &p->d_dict, &p->d_dict32, &p->ndict, s); }
   // a matrix that cannot change behind its plan and is not coded: row slices (a private copy of the fp64 values); the
   // launches the form does not serve (triangular fills, an unaligned x) gather x through the cache from the tile tables
   const int rs_mode = spmv_row_slices();
   if (!p->d_codes && (rs_mode >= 2 || (rs_mode == 1 && p->owned)) && device_build_row_slices(p, A, s)) { return; }
   if (!p->d_stale) { plan_free(p->d_codes); plan_free(p->d_dict); p->d_codes = nullptr; p->d_dict = p->d_dict32 = nullptr; p->ndict = 0; return; }   // the x-staged kernels write their verdict somewhere
   const size_t nl = ((size_t) A->num_nonzeros + 15) & ~(size_t) 7;
   if (!plan_alloc((void **) &p->d_tile_fp, sizeof(int) * (size_t) p->num_tiles, PLAN_SITE_XS) ||
       !plan_alloc((void **) &p->d_xs_cnt, sizeof(int) * (size_t) p->num_tiles, PLAN_SITE_XS) ||
       !plan_alloc((void **) &p->d_xs_desc, sizeof(int) * (size_t) p->num_tiles * 2 * SPMV_XS_SEGS, PLAN_SITE_XS) ||
       !plan_alloc((void **) &p->d_lidx, sizeof(unsigned short) * nl, PLAN_SITE_XS))
   {
      plan_free(p->d_tile_fp); plan_free(p->d_xs_cnt); plan_free(p->d_xs_desc); plan_free(p->d_lidx);
      p->d_tile_fp = nullptr; p->d_xs_cnt = nullptr; p->d_xs_desc = nullptr; p->d_lidx = nullptr;
      plan_free(p->d_codes); plan_free(p->d_dict);           // (only the x-staged kernels read codes)
      p->d_codes = nullptr; p->d_dict = p->d_dict32 = nullptr; p->ndict = 0;
      return;
   }
   launch_build_fp(A->j, A->num_nonzeros, p->num_tiles, p->d_tile_fp, s);
   HIP_CHECK(hipMemsetAsync(p->d_lidx, 0, sizeof(unsigned short) * nl, s));
   launch_build_xs(A->j, p->d_tile_k, p->num_tiles, p->d_xs_cnt, p->d_xs_desc, p->d_lidx, s);
   // the product area doubles as the staging area: size it by the longest staged copy of this matrix
   std::vector<int> cnt((size_t) p->num_tiles);
   HIP_CHECK(hipMemcpyAsync(cnt.data(), p->d_xs_cnt, sizeof(int) * (size_t) p->num_tiles, hipMemcpyDeviceToHost, s));
   HIP_CHECK(hipStreamSynchronize(s));
   std::vector<int> cov;
   cov.reserve(cnt.size());
   for (int c : cnt) { if (c & 0xff) { p->xs_tiles++; p->xs_max_units = std::max(p->xs_max_units, c >> 8); cov.push_back(c >> 8); } }
   // LDS per workgroup decides how many tiles a CU holds (160 KB): stage what 99 % of the tiles need, rounded up
   // to the next occupancy step; the few tiles above that gather through the cache
   if (!cov.empty())
   {
      std::sort(cov.begin(), cov.end());
      static int pct = -1, verbose = 0;
      if (pct < 0)
      {
         const char *e = getenv("HYPRE_AMD_SPMV_XS_PCT");
         pct = e ? std::min(100, std::max(1, atoi(e))) : 99;
         verbose = getenv("HYPRE_AMD_PLAN_VERBOSE") != nullptr;
      }
      const int need = cov[(size_t) ((cov.size() - 1) * (size_t) pct / 100)];
      // what the launch puts beside the products (tiled_lds_bytes / launch_xs of spmv_kernels.hip): row sums and row pointers
      // for the most rows a tile of THIS matrix holds, and the value table of a coded matrix
      const int rows = p->max_tile_rows > 0 ? std::min(p->max_tile_rows, RP_CAP_HOST) : RP_CAP_HOST;
      const int other = 8 * (p->max_row_nnz > 12 ? ((std::min(rows, SPMV_THREADS) + 1) & ~1) : 0) + 4 * (rows + 4) + 64 +
                        (p->d_codes ? 8 * ((p->ndict + 1) & ~1) + 16 : 0);
      int units = p->xs_max_units;
      for (int wgs = 8; wgs >= 3; wgs--)
      {
         const int room = ((160 * 1024) / wgs) / 1280 * 1280;           // LDS is handed out in blocks (1280 bytes on gfx950)
         const int fit = (room - other) / 16;                           // units (16 bytes) that leave room for wgs workgroups
         if (fit >= need) { units = std::min(fit, p->xs_max_units); break; }
      }
      p->xs_launch_units = std::max(units, need);
      if (verbose)
      {
         fprintf(stderr, "[plan] %d x %d, %d tiles: staged units p50 %d p90 %d p99 %d max %d -> launch %d (%d bytes of LDS)\n",
                 A->num_rows, A->num_cols, p->num_tiles, cov[cov.size() / 2], cov[(cov.size() - 1) * 9 / 10],
                 cov[(cov.size() - 1) * 99 / 100], cov.back(), p->xs_launch_units, 16 * p->xs_launch_units + other);
      }
   }
   p->prod_elems = std::max(p->prod_elems, 2 * p->xs_launch_units + 8);
   if (p->d_codes && spmv_slice_form()) { device_build_slice_form(p, A, s); }
}

SpmvPlan *get_plan(hypre_CSRMatrix *A)
{
   auto &t = plan_table();
   auto it = t.find(A);
   if (it != t.end())
   {
      SpmvPlan *p = it->second;
      const bool flagged = p->h_stale && __atomic_load_n(p->h_stale, __ATOMIC_RELAXED) != 0;
      if (!flagged && p->i == A->i && p->j == A->j && p->a == A->data && p->nnz == A->num_nonzeros &&
          p->num_rows == A->num_rows && p->num_cols == A->num_cols)
      {
         return p;
      }
      if (flagged)
      {
         hypre_error_w_msg(HYPRE_ERROR_GENERIC, "a kernel found its SpMV plan out of date (the matrix at this address is not the one the plan "
                                                "was built for, or its values changed without hypre_amd_CSRMatrixInvalidatePlan): the products "
                                                "since then are wrong; the plan is rebuilt");
         drop_gs_schedule(A);
         drop_mc_plan(A);
      }
      free_plan(p);
      t.erase(it);
   }
   SpmvPlan *p = new SpmvPlan();
   p->i = A->i; p->j = A->j; p->a = A->data;
   p->num_rows = A->num_rows; p->num_cols = A->num_cols; p->nnz = A->num_nonzeros;
   p->owned = is_owned(A);
   hipStream_t s = stream();
   if (A->num_rows > 0 && A->num_nonzeros > 0)
   {
      p->max_row_nnz = device_max_row_nnz(A->i, A->num_rows, s);
      const bool aligned = (((uintptr_t) A->j) & 15) == 0 && (((uintptr_t) A->data) & 15) == 0;
      p->tiled = aligned && p->max_row_nnz <= SPMV_MAXROW && build_tile_tables(p, A, s);     // false: a wave per row
      if (p->tiled)
      {
         build_band_placement(p, A, s);
         if (spmv_variant().variant == 2) { build_staging(p, A, s); }
      }
      // a matrix somebody else may write to: the checksum its plan can be verified against (plan_verify)
      if (p->tiled && !p->owned)
      {
         p->checksum = device_csr_checksum(A->i, A->j, A->data, A->num_rows, A->num_nonzeros, s);
         p->has_checksum = true;
      }
   }
   t[A] = p;
   return p;
}

bool plan_verify(hypre_CSRMatrix *A)
{
   auto &t = plan_table();
   auto it = t.find(A);
   if (it == t.end()) { return true; }
   SpmvPlan *p = it->second;
   if (!p->has_checksum) { return true; }
   const bool same_arrays = p->i == A->i && p->j == A->j && p->a == A->data && p->nnz == A->num_nonzeros &&
                            p->num_rows == A->num_rows && p->num_cols == A->num_cols;
   if (same_arrays && device_csr_checksum(A->i, A->j, A->data, A->num_rows, A->num_nonzeros, stream()) == p->checksum) { return true; }
   drop_plan(A);
   return false;
}

bool &spmv_slice_form()
{
   static bool on = [] { const char *e = getenv("HYPRE_AMD_SPMV_SLICE_FORM"); return !(e && atoi(e) == 0); }();
   return on;
}

bool &spmv_fused_multivectors()
{
   static bool on = [] { const char *e = getenv("HYPRE_AMD_SPMV_FUSED_MV"); return !(e && atoi(e) == 0); }();
   return on;
}

bool &spmv_value_codes()
{
   static bool on = [] { const char *e = getenv("HYPRE_AMD_SPMV_VALUE_CODES"); return !(e && atoi(e) == 0); }();
   return on;
}

bool plan_is_stale(const hypre_CSRMatrix *A)
{
   auto &t = plan_table();
   auto it = t.find(A);
   return it != t.end() && it->second->h_stale && __atomic_load_n(it->second->h_stale, __ATOMIC_RELAXED) != 0;
}

// fp32 copy of a device matrix's values (mixed precision), converted once and cached in the plan
const float *fp32_values_of(hypre_CSRMatrix *A)
{
   if (!A || A->num_nonzeros <= 0 || !A->data) { return nullptr; }
   SpmvPlan *p = get_plan(A);
   if (!p->a32)
   {
      HIP_CHECK(hipMalloc((void **) &p->a32, sizeof(float) * (((size_t) A->num_nonzeros + 3) & ~(size_t) 3)));
      launch_f64_to_f32(A->data, p->a32, (size_t) A->num_nonzeros, stream());
   }
   return p->a32;
}

// Hand a matrix its strictly-lower copy ready made (replicated tail: the triangle of every rank's own diagonal
// block, which is what the distributed two-stage sweep multiplies by, not the triangle of the gathered matrix).
void set_strict_lower(hypre_CSRMatrix *A, hypre_CSRMatrix *L)
{
   SpmvPlan *plan = get_plan(A);
   if (plan->Lstrict && plan->Lstrict != L) { hypre_CSRMatrixDestroy(plan->Lstrict); }
   plan->Lstrict = L;
}

// Strictly lower triangular part of a device matrix as its own CSR matrix (entries in stored
// order), cached in the plan: counted and filled by one lane per row, row offsets scanned on the host.
hypre_CSRMatrix *strict_lower_of(hypre_CSRMatrix *A)
{
   SpmvPlan *plan = get_plan(A);
   if (plan->Lstrict) { return plan->Lstrict; }
   const int n = A->num_rows;
   hipStream_t s = stream();
   std::vector<int> cnt((size_t) std::max(n, 1)), off((size_t) n + 1, 0);
   int *d_cnt = hypre_TAlloc(int, (size_t) std::max(n, 1), HYPRE_MEMORY_DEVICE);
   launch_count_lower(A->i, A->j, n, d_cnt, s);
   HIP_CHECK(hipMemcpyAsync(cnt.data(), d_cnt, sizeof(int) * (size_t) n, hipMemcpyDeviceToHost, s));
   HIP_CHECK(hipStreamSynchronize(s));
   for (int i = 0; i < n; i++) { off[(size_t) i + 1] = off[(size_t) i] + cnt[(size_t) i]; }
   hypre_Free(d_cnt, HYPRE_MEMORY_DEVICE);
   hypre_CSRMatrix *L = hypre_CSRMatrixCreate(n, A->num_cols, off[(size_t) n]);
   hypre_CSRMatrixInitialize_v2(L, 0, HYPRE_MEMORY_DEVICE);
   hypre_TMemcpy(L->i, off.data(), HYPRE_Int, (size_t) n + 1, HYPRE_MEMORY_DEVICE, HYPRE_MEMORY_HOST);
   if (off[(size_t) n] > 0) { launch_fill_lower(A->i, A->j, A->data, L->i, L->j, L->data, n, s); }
   if (plan->owned) { mark_owned(L); }            // a copy of a matrix that cannot change (a copy of the caller's matrix is watched with it)
   plan->Lstrict = L;
   return L;
}

bool spmv_with_scaled_quotient(hypre_CSRMatrix *M, const double *x, double *y, double w, const double *d, double *u)
{
   if (!M || M->num_rows <= 0 || M->num_nonzeros <= 0 || !d || !u || M->memory_location != HYPRE_MEMORY_DEVICE) { return false; }
   SpmvArgs a{};
   a.Ai = M->i; a.Aj = M->j; a.Aa = M->data; a.Aa32 = nullptr;
   a.x = x; a.b = nullptr; a.y = y; a.d = d; a.aux = u; a.marker = nullptr; a.marker_val = 0;
   a.alpha = 1.0; a.beta = 0.0; a.scale2 = w; a.fill = HYPRE_SPMV_FILL_WHOLE; a.row_offset = 0;
   spmv_default_flags(a);
   launch_spmv(get_plan(M), a, OP_AXPBY_DIV, stream());
   return true;
}
}  // namespace hamd

// Band-aware tile placement policy of the plans built from now on (existing plans keep theirs): enabled 0/1,
// min_tiles = smallest matrix (in 2048-entry tiles) that gets a table, force != 0 drops the minimum slab size.
// Negative arguments leave a field as it is.  Speed only — except that tests use it to put small matrices through
// the placement-table code the benchmark sizes run.
extern "C" HYPRE_Int hypre_amd_SpmvSetBandPolicy(HYPRE_Int enabled, HYPRE_Int min_tiles, HYPRE_Int force)
{
   hamd::BandPolicy &bp = hamd::band_policy();
   if (enabled >= 0) { bp.enabled = enabled; }
   if (min_tiles >= 0) { bp.min_tiles = min_tiles; }
   if (force >= 0) { bp.force = force; }
   return hypre_error_flag;
}

// Kernel variant of the tiled SpMV family from now on: 2 = x staged through LDS (plans built from now on carry the
// chunk lists), 0 = x gathered through the cache.  Speed only: same products, same summation order.
extern "C" HYPRE_Int hypre_amd_SpmvSetVariant(HYPRE_Int variant, HYPRE_Int unused)
{
   (void) unused;
   hamd::SpmvVariant &v = hamd::spmv_variant();
   if (variant >= 0) { v.variant = variant; }
   return hypre_error_flag;
}

// Value codes for the plans built from now on (existing plans keep theirs): a matrix that holds at most 256 distinct values
// is streamed by the x-staged kernel as one byte per entry and a table.  Speed only: same products, bit for bit.
extern "C" HYPRE_Int hypre_amd_SpmvSetValueCodes(HYPRE_Int on)
{
   if (on >= 0) { hamd::spmv_value_codes() = on != 0; }
   return hypre_error_flag;
}

// Slice form of coded short-row matrices for the plans built from now on.  Speed only.
extern "C" HYPRE_Int hypre_amd_SpmvSetSliceForm(HYPRE_Int on)
{
   if (on >= 0) { hamd::spmv_slice_form() = on != 0; }
   return hypre_error_flag;
}

// Row-slice form for the plans built from now on: 0 off, 1 matrices the library owns or the caller declared immutable
// (default), 2 every matrix (tests: the caller then owes hypre_amd_CSRMatrixInvalidatePlan after any change).  Speed only.
extern "C" HYPRE_Int hypre_amd_SpmvSetRowSlices(HYPRE_Int mode)
{
   if (mode >= 0) { hamd::spmv_row_slices() = mode; }
   return hypre_error_flag;
}
// lanes per row of the row-slice form in A's plan (built on demand), the rows of a block and the entries a lane holds at
// most; 0: the matrix has none
extern "C" HYPRE_Int hypre_amd_CSRMatrixPlanRowSlices(hypre_CSRMatrix *A, HYPRE_Int *rows_per_block, HYPRE_Int *entries_per_lane)
{
   if (rows_per_block) { *rows_per_block = 0; }
   if (entries_per_lane) { *entries_per_lane = 0; }
   if (!A || A->memory_location != HYPRE_MEMORY_DEVICE) { return 0; }
   hamd::SpmvPlan *p = hamd::get_plan(A);
   if (!p->d_rs_val) { return 0; }
   if (rows_per_block) { *rows_per_block = p->rs_rows; }
   if (entries_per_lane) { *entries_per_lane = p->rs_kp; }
   return p->rs_w;
}

// lanes per row (1 or 2) of the slice form in A's plan (built on demand); 0: the matrix has none
extern "C" HYPRE_Int hypre_amd_CSRMatrixPlanSliceForm(hypre_CSRMatrix *A)
{
   if (A->memory_location != HYPRE_MEMORY_DEVICE) { return 0; }
   hamd::SpmvPlan *p = hamd::get_plan(A);
   return p->d_sl_data ? p->sl_w : 0;
}

// number of distinct values in the value table of A's plan (built on demand); 0: the matrix is not coded
extern "C" HYPRE_Int hypre_amd_CSRMatrixPlanValueCodes(hypre_CSRMatrix *A)
{
   if (A->memory_location != HYPRE_MEMORY_DEVICE) { return 0; }
   hamd::SpmvPlan *p = hamd::get_plan(A);
   return p->d_codes ? p->ndict : 0;
}

// 1 when the plan of A (built on demand) carries a placement table, with the band distance it was built for
extern "C" HYPRE_Int hypre_amd_CSRMatrixPlanInfo(hypre_CSRMatrix *A, HYPRE_Int *num_tiles, HYPRE_Int *band)
{
   if (A->memory_location != HYPRE_MEMORY_DEVICE) { if (num_tiles) { *num_tiles = 0; } if (band) { *band = 0; } return 0; }
   hamd::SpmvPlan *p = hamd::get_plan(A);
   if (num_tiles) { *num_tiles = p->tiled ? p->num_tiles : 0; }
   if (band) { *band = p->band; }
   return p->d_tile_perm != nullptr;
}

// x staging of A's plan (built on demand): returns the number of tiles whose columns fit the staging area (they take
// the LDS path of spmv_xs_kernel, the others gather), fills the tile count and the mean number of pieces per staged tile
extern "C" HYPRE_Int hypre_amd_CSRMatrixPlanStaging(hypre_CSRMatrix *A, HYPRE_Int *num_tiles, HYPRE_Real *mean_pieces)
{
   if (num_tiles) { *num_tiles = 0; }
   if (mean_pieces) { *mean_pieces = 0.0; }
   if (A->memory_location != HYPRE_MEMORY_DEVICE) { return 0; }
   hamd::SpmvPlan *p = hamd::get_plan(A);
   if (!p->tiled || !p->d_xs_cnt) { return 0; }
   std::vector<int> cnt((size_t) p->num_tiles);
   HIP_CHECK(hipStreamSynchronize(hamd::stream()));
   HIP_CHECK(hipMemcpy(cnt.data(), p->d_xs_cnt, sizeof(int) * (size_t) p->num_tiles, hipMemcpyDeviceToHost));
   long long staged = 0, pieces = 0;
   for (int c : cnt) { if (c & 0xff) { staged++; pieces += c & 0xff; } }
   if (num_tiles) { *num_tiles = p->num_tiles; }
   if (mean_pieces && staged) { *mean_pieces = (double) pieces / (double) staged; }
   return (HYPRE_Int) staged;
}

// Columns ascending inside every row of a device matrix, in place (keep_first != 0: the first entry of a row — the
// diagonal of a square block — stays in front); the matrix's plan is dropped.
extern "C" HYPRE_Int hypre_amd_CSRMatrixSortRows(hypre_CSRMatrix *A, HYPRE_Int keep_first)
{
   HYPRE_AMD_REQUIRE_DEVICE(A->memory_location, "hypre_amd_CSRMatrixSortRows(A)");
   hamd::drop_plan(A);
   if (A->num_nonzeros > 0) { hamd::launch_sort_rows(A->i, A->j, A->data, A->num_rows, keep_first, hamd::stream()); }
   hamd::maybe_sync();
   return hypre_error_flag;
}

extern "C" HYPRE_Int hypre_amd_CSRMatrixInvalidatePlan(hypre_CSRMatrix *A)
{
   drop_plan(A);
   return hypre_error_flag;
}

// 1: the arrays of A are what its plan was built from (checksum over row pointers, columns and value bit patterns: one pass
// over the CSR arrays); 0: they are not — the plan has been dropped, the next product builds a fresh one, nothing is raised
// (no product has used the stale plan since this call).  A matrix without a plan, or one the library owns: 1.
extern "C" HYPRE_Int hypre_amd_CSRMatrixVerifyPlan(hypre_CSRMatrix *A)
{
   if (!A || A->memory_location != HYPRE_MEMORY_DEVICE) { return 1; }
   return hamd::plan_verify(A) ? 1 : 0;
}

// The caller's promise that the arrays of A do not change until it calls hypre_amd_CSRMatrixInvalidatePlan (or this function
// with on = 0) or destroys A: the plan may then keep private copies of the values in whatever form multiplies fastest (the
// row-slice form), and its launches carry no watch.  The library sets this itself for the matrices it makes (the levels of
// a hierarchy below the finest, interpolation and restriction operators).  Changing the promise drops the plan.
extern "C" HYPRE_Int hypre_amd_CSRMatrixSetImmutable(hypre_CSRMatrix *A, HYPRE_Int on)
{
   if (!A) { hypre_error_in_arg(1); return hypre_error_flag; }
   if (hamd::is_owned(A) != (on != 0)) { drop_plan(A); hamd::mark_owned(A, on != 0); }
   return hypre_error_flag;
}

// Test hook: the nth allocation (1 = the next) that a plan builder makes at `site` fails, once — 1 tile tables, 2 x-staging
// tables, 3 value codes, 4 slice form, 5 row-slice form.  nth <= 0 disarms.  Returns what was still pending of the request
// before (0: it happened, or nothing was armed).
extern "C" HYPRE_Int hypre_amd_PlanTestFailAlloc(HYPRE_Int site, HYPRE_Int nth)
{
   const HYPRE_Int pending = hamd::g_fail_site ? hamd::g_fail_nth : 0;      // > 0: the failure armed before has not happened
   hamd::g_fail_site = nth > 0 ? site : 0;
   hamd::g_fail_nth = nth > 0 ? nth : 0;
   return pending;
}

// Mixed precision for the products called directly (the AMG cycle sets this from its solver: hypre_amd_BoomerAMGSetMixedPrecision):
// from now on the SpMV-class kernels stream an fp32 copy of the matrix values, vectors and sums stay fp64.
extern "C" HYPRE_Int hypre_amd_SetMixedPrecisionValues(HYPRE_Int on)
{
   hamd::handle().fp32_values = on != 0;
   return hypre_error_flag;
}

// which kernel form the plan of the device matrix A (built on demand) multiplies with: 0 a wave per row, 1 tiles with x
// gathered through the cache, 2 tiles with x staged through LDS, 3 coded tiles, 4 slice form of a coded stencil, 5 row-slice form
extern "C" HYPRE_Int hypre_amd_CSRMatrixPlanForm(hypre_CSRMatrix *A)
{
   if (!A || A->memory_location != HYPRE_MEMORY_DEVICE) { return -1; }
   hamd::SpmvPlan *p = hamd::get_plan(A);
   if (!p->tiled) { return 0; }
   if (p->d_rs_val) { return 5; }
   if (!p->d_lidx) { return 1; }
   if (p->d_sl_data) { return 4; }
   if (p->d_codes) { return 3; }
   return 2;
}

// ===========================================================================
// CSR matrix object
// ===========================================================================
extern "C" {

hypre_CSRMatrix *hypre_CSRMatrixCreate(HYPRE_Int num_rows, HYPRE_Int num_cols, HYPRE_Int num_nonzeros)
{
   hypre_CSRMatrix *m = (hypre_CSRMatrix *) calloc(1, sizeof(hypre_CSRMatrix));
   mark_owned(m, false);              // (an address handed out again starts as nobody's)
   m->num_rows = num_rows;
   m->num_cols = num_cols;
   m->num_nonzeros = num_nonzeros;
   m->num_rownnz = num_rows;
   m->owns_data = 1;
   m->memory_location = handle().memory_location;
   return m;
}

HYPRE_Int hypre_CSRMatrixInitialize_v2(hypre_CSRMatrix *m, HYPRE_Int bigInit, HYPRE_MemoryLocation loc)
{
   m->memory_location = loc;
   if (!m->data && m->num_nonzeros) { m->data = hypre_CTAlloc(HYPRE_Complex, m->num_nonzeros, loc); }
   if (!m->i) { m->i = hypre_CTAlloc(HYPRE_Int, m->num_rows + 1, loc); }
   if (bigInit)
   {
      if (!m->big_j && m->num_nonzeros) { m->big_j = hypre_CTAlloc(HYPRE_BigInt, m->num_nonzeros, loc); }
   }
   else
   {
      if (!m->j && m->num_nonzeros) { m->j = hypre_CTAlloc(HYPRE_Int, m->num_nonzeros, loc); }
   }
   return hypre_error_flag;
}

HYPRE_Int hypre_CSRMatrixInitialize(hypre_CSRMatrix *m)
{
   return hypre_CSRMatrixInitialize_v2(m, 0, m->memory_location);
}

HYPRE_Int hypre_CSRMatrixDestroy(hypre_CSRMatrix *m)
{
   if (!m) { return hypre_error_flag; }
   drop_plan(m);
   mark_owned(m, false);
   const HYPRE_MemoryLocation loc = m->memory_location;
   hypre_Free(m->rownnz, loc);
   if (m->owns_data)
   {
      hypre_Free(m->data, loc);
      hypre_Free(m->i, loc);
      hypre_Free(m->j, loc);
      hypre_Free(m->big_j, loc);
   }
   free(m);
   return hypre_error_flag;
}

// List the rows that hold at least one entry (seq_mv/csr_matrix.c:350-400).
HYPRE_Int hypre_CSRMatrixSetRownnz(hypre_CSRMatrix *m)
{
   const HYPRE_MemoryLocation loc = m->memory_location;
   const HYPRE_Int n = m->num_rows;
   std::vector<HYPRE_Int> hi;
   const HYPRE_Int *Ai = m->i;
   if (loc == HYPRE_MEMORY_DEVICE)
   {
      hi.resize((size_t) n + 1);
      hypre_TMemcpy(hi.data(), m->i, HYPRE_Int, n + 1, HYPRE_MEMORY_HOST, HYPRE_MEMORY_DEVICE);
      Ai = hi.data();
   }
   HYPRE_Int cnt = 0;
   for (HYPRE_Int r = 0; r < n; r++) { if (Ai[r + 1] > Ai[r]) { cnt++; } }
   hypre_Free(m->rownnz, loc);
   m->rownnz = nullptr;
   m->num_rownnz = cnt;
   if (cnt == 0 || cnt == n) { return hypre_error_flag; }
   std::vector<HYPRE_Int> list((size_t) cnt);
   cnt = 0;
   for (HYPRE_Int r = 0; r < n; r++) { if (Ai[r + 1] > Ai[r]) { list[(size_t) cnt++] = r; } }
   m->rownnz = hypre_TAlloc(HYPRE_Int, cnt, loc);
   hypre_TMemcpy(m->rownnz, list.data(), HYPRE_Int, cnt, loc, HYPRE_MEMORY_HOST);
   return hypre_error_flag;
}

static void migrate_array(void **p, size_t bytes, HYPRE_MemoryLocation from, HYPRE_MemoryLocation to)
{
   if (!*p) { return; }
   // an empty array may still own a (minimal) allocation: it has to change memory space with its header, or it is
   // later freed in the wrong one (a rank without rows on a coarse level: hipFree of a host pointer)
   void *q = hypre_MAlloc(bytes ? bytes : 8, to);
   if (bytes) { hypre_Memcpy(q, *p, bytes, to, from); }
   hypre_Free(*p, from);
   *p = q;
}

HYPRE_Int hypre_CSRMatrixMigrate(hypre_CSRMatrix *A, HYPRE_MemoryLocation to)
{
   const HYPRE_MemoryLocation from = A->memory_location;
   if (from == to) { return hypre_error_flag; }
   drop_plan(A);
   migrate_array((void **) &A->i, sizeof(HYPRE_Int) * (size_t) (A->num_rows + 1), from, to);
   migrate_array((void **) &A->j, sizeof(HYPRE_Int) * (size_t) A->num_nonzeros, from, to);
   migrate_array((void **) &A->big_j, sizeof(HYPRE_BigInt) * (size_t) A->num_nonzeros, from, to);
   migrate_array((void **) &A->data, sizeof(HYPRE_Complex) * (size_t) A->num_nonzeros, from, to);
   if (A->rownnz) { migrate_array((void **) &A->rownnz, sizeof(HYPRE_Int) * (size_t) A->num_rownnz, from, to); }
   A->memory_location = to;
   return hypre_error_flag;
}

hypre_CSRMatrix *hypre_CSRMatrixClone_v2(hypre_CSRMatrix *A, HYPRE_Int copy_data, HYPRE_MemoryLocation loc)
{
   hypre_CSRMatrix *B = hypre_CSRMatrixCreate(A->num_rows, A->num_cols, A->num_nonzeros);
   hypre_CSRMatrixInitialize_v2(B, A->big_j != nullptr && A->j == nullptr, loc);
   const HYPRE_MemoryLocation src = A->memory_location;
   hypre_TMemcpy(B->i, A->i, HYPRE_Int, A->num_rows + 1, loc, src);
   if (A->j) { hypre_TMemcpy(B->j, A->j, HYPRE_Int, A->num_nonzeros, loc, src); }
   if (A->big_j && B->big_j) { hypre_TMemcpy(B->big_j, A->big_j, HYPRE_BigInt, A->num_nonzeros, loc, src); }
   if (copy_data && A->data) { hypre_TMemcpy(B->data, A->data, HYPRE_Complex, A->num_nonzeros, loc, src); }
   B->num_rownnz = A->num_rownnz;
   if (A->rownnz)
   {
      B->rownnz = hypre_TAlloc(HYPRE_Int, A->num_rownnz, loc);
      hypre_TMemcpy(B->rownnz, A->rownnz, HYPRE_Int, A->num_rownnz, loc, src);
   }
   return B;
}

// Explicit transpose (setup-time utility; counting sort on the host).
// Column order inside each row of AT is ascending source row, as produced by
// the reference's host transpose (seq_mv/csr_matop.c:1043-1270).
HYPRE_Int hypre_CSRMatrixTranspose(hypre_CSRMatrix *A, hypre_CSRMatrix **AT_ptr, HYPRE_Int data)
{
   const HYPRE_MemoryLocation loc = A->memory_location;
   const HYPRE_Int nr = A->num_rows, nc = A->num_cols, nnz = A->num_nonzeros;
   if (loc == HYPRE_MEMORY_DEVICE)
   {
      // device matrices are transposed where they are (count / scan / scatter / order: kernels.hip), same result
      hypre_CSRMatrix *AT = hypre_CSRMatrixCreate(nc, nr, nnz);
      const bool with_data = data && A->data;
      hypre_CSRMatrixInitialize_v2(AT, 0, loc);
      if (!with_data && AT->data) { hypre_Free(AT->data, loc); AT->data = nullptr; }
      launch_transpose(A->i, A->j, with_data ? A->data : nullptr, nr, nc, nnz, AT->i, AT->j, with_data ? AT->data : nullptr, stream());
      HIP_CHECK(hipStreamSynchronize(stream()));
      *AT_ptr = AT;
      return hypre_error_flag;
   }
   const HYPRE_Int *Ai = A->i, *Aj = A->j;
   const HYPRE_Complex *Aa = A->data;
   const bool with_data = data && A->data;
   std::vector<HYPRE_Int> ti((size_t) nc + 1, 0), tj((size_t) nnz);
   std::vector<HYPRE_Complex> ta(with_data ? (size_t) nnz : 0);
   // Stable counting sort by column (rows of A^T list their entries by ascending row of A).  Large
   // matrices: T row blocks count their columns separately (seq_mv/csr_matop.c:1060-1230 does the
   // same with one bucket array per OpenMP thread); block t's entries of a column go behind those of
   // the blocks before it, so the result is the sequential one.
   int T = 1;
   if (nnz > (1 << 20))
   {
      T = std::min(omp_get_max_threads(), 16);
      while (T > 1 && (size_t) T * (size_t) nc > (size_t) 1 << 30) { T /= 2; }
   }
   if (T <= 1)
   {
      for (HYPRE_Int k = 0; k < nnz; k++) { ti[(size_t) Aj[k] + 1]++; }
      for (HYPRE_Int c = 0; c < nc; c++) { ti[(size_t) c + 1] += ti[(size_t) c]; }
      std::vector<HYPRE_Int> pos(ti.begin(), ti.end() - 1);
      for (HYPRE_Int r = 0; r < nr; r++)
      {
         for (HYPRE_Int k = Ai[r]; k < Ai[r + 1]; k++)
         {
            const HYPRE_Int q = pos[(size_t) Aj[k]]++;
            tj[(size_t) q] = r;
            if (with_data) { ta[(size_t) q] = Aa[k]; }
         }
      }
   }
   else
   {
      std::vector<HYPRE_Int> cnt((size_t) T * (size_t) nc, 0);
      std::vector<HYPRE_Int> rbeg((size_t) T + 1, 0);
      for (int t = 0; t <= T; t++)
      {
         // row blocks balanced by entries
         const long long target = (long long) nnz * t / T;
         rbeg[(size_t) t] = (HYPRE_Int) (std::lower_bound(Ai, Ai + nr + 1, (HYPRE_Int) target) - Ai);
      }
      rbeg[0] = 0; rbeg[(size_t) T] = nr;
#pragma omp parallel num_threads(T)
      {
         const int t = omp_get_thread_num();
         HYPRE_Int *c = cnt.data() + (size_t) t * (size_t) nc;
         for (HYPRE_Int k = Ai[rbeg[(size_t) t]]; k < Ai[rbeg[(size_t) t + 1]]; k++) { c[Aj[k]]++; }
#pragma omp barrier
         // column totals, then (after the scan) each block's first slot per column
#pragma omp for schedule(static)
         for (HYPRE_Int col = 0; col < nc; col++)
         {
            HYPRE_Int tot = 0;
            for (int b = 0; b < T; b++) { tot += cnt[(size_t) b * (size_t) nc + (size_t) col]; }
            ti[(size_t) col + 1] = tot;
         }
#pragma omp single
         { for (HYPRE_Int col = 0; col < nc; col++) { ti[(size_t) col + 1] += ti[(size_t) col]; } }
#pragma omp for schedule(static)
         for (HYPRE_Int col = 0; col < nc; col++)
         {
            HYPRE_Int off = ti[(size_t) col];
            for (int b = 0; b < T; b++)
            {
               const HYPRE_Int m = cnt[(size_t) b * (size_t) nc + (size_t) col];
               cnt[(size_t) b * (size_t) nc + (size_t) col] = off;
               off += m;
            }
         }
         for (HYPRE_Int r = rbeg[(size_t) t]; r < rbeg[(size_t) t + 1]; r++)
         {
            for (HYPRE_Int k = Ai[r]; k < Ai[r + 1]; k++)
            {
               const HYPRE_Int q = c[Aj[k]]++;
               tj[(size_t) q] = r;
               if (with_data) { ta[(size_t) q] = Aa[k]; }
            }
         }
      }
   }
   hypre_CSRMatrix *AT = hypre_CSRMatrixCreate(nc, nr, nnz);
   hypre_CSRMatrixInitialize_v2(AT, 0, loc);
   hypre_TMemcpy(AT->i, ti.data(), HYPRE_Int, nc + 1, loc, HYPRE_MEMORY_HOST);
   if (nnz)
   {
      hypre_TMemcpy(AT->j, tj.data(), HYPRE_Int, nnz, loc, HYPRE_MEMORY_HOST);
      if (with_data) { hypre_TMemcpy(AT->data, ta.data(), HYPRE_Complex, nnz, loc, HYPRE_MEMORY_HOST); }
   }
   *AT_ptr = AT;
   return hypre_error_flag;
}

// Move the diagonal entry of every row of a square host matrix to the front
// (seq_mv/csr_matop.c:1537-1584); the relaxation kernels rely on it.
HYPRE_Int hypre_CSRMatrixReorder(hypre_CSRMatrix *A)
{
   if (A->num_rows != A->num_cols) { return -1; }
   if (A->memory_location != HYPRE_MEMORY_HOST)
   {
      hypre_error_w_msg(HYPRE_ERROR_GENERIC, "hypre_CSRMatrixReorder: host matrices only (setup-time utility)");
      return hypre_error_flag;
   }
   for (HYPRE_Int r = 0; r < A->num_rows; r++)
   {
      const HYPRE_Int s = A->i[r], e = A->i[r + 1];
      for (HYPRE_Int k = s; k < e; k++)
      {
         if (A->j[k] == r)
         {
            if (k != s)
            {
               std::swap(A->j[s], A->j[k]);
               std::swap(A->data[s], A->data[k]);
            }
            break;
         }
      }
   }
   return hypre_error_flag;
}

// ===========================================================================
// dense vector object
// ===========================================================================
hypre_Vector *hypre_SeqMultiVectorCreate(HYPRE_Int size, HYPRE_Int num_vectors)
{
   hypre_Vector *v = (hypre_Vector *) calloc(1, sizeof(hypre_Vector));
   v->size = size;
   v->num_vectors = num_vectors;
   v->owns_data = 1;
   v->multivec_storage_method = 0;
   v->vecstride = size;
   v->idxstride = 1;
   v->memory_location = handle().memory_location;
   return v;
}

hypre_Vector *hypre_SeqVectorCreate(HYPRE_Int size) { return hypre_SeqMultiVectorCreate(size, 1); }

HYPRE_Int hypre_SeqVectorInitialize_v2(hypre_Vector *v, HYPRE_MemoryLocation loc)
{
   v->memory_location = loc;
   if (!v->data) { v->data = hypre_CTAlloc(HYPRE_Complex, (size_t) v->size * v->num_vectors, loc); }
   if (v->multivec_storage_method == 0) { v->vecstride = v->size; v->idxstride = 1; }
   else { v->vecstride = 1; v->idxstride = v->num_vectors; }
   return hypre_error_flag;
}

HYPRE_Int hypre_SeqVectorInitialize(hypre_Vector *v) { return hypre_SeqVectorInitialize_v2(v, v->memory_location); }

HYPRE_Int hypre_SeqVectorDestroy(hypre_Vector *v)
{
   if (!v) { return hypre_error_flag; }
   if (v->owns_data) { hypre_Free(v->data, v->memory_location); }
   free(v);
   return hypre_error_flag;
}

HYPRE_Int hypre_SeqVectorMigrate(hypre_Vector *x, HYPRE_MemoryLocation to)
{
   if (x->memory_location == to) { return hypre_error_flag; }
   migrate_array((void **) &x->data, sizeof(HYPRE_Complex) * (size_t) x->size * x->num_vectors, x->memory_location, to);
   x->memory_location = to;
   return hypre_error_flag;
}

hypre_Vector *hypre_SeqVectorCloneDeep_v2(hypre_Vector *x, HYPRE_MemoryLocation loc)
{
   hypre_Vector *y = hypre_SeqMultiVectorCreate(x->size, x->num_vectors);
   y->multivec_storage_method = x->multivec_storage_method;
   hypre_SeqVectorInitialize_v2(y, loc);
   hypre_TMemcpy(y->data, x->data, HYPRE_Complex, (size_t) x->size * x->num_vectors, loc, x->memory_location);
   return y;
}

hypre_Vector *hypre_SeqVectorCloneDeep(hypre_Vector *x) { return hypre_SeqVectorCloneDeep_v2(x, x->memory_location); }

// ===========================================================================
// SpMV
// ===========================================================================
static HYPRE_Int matvec_ierr(HYPRE_Int num_rows, HYPRE_Int num_cols, hypre_Vector *x, hypre_Vector *b,
                             hypre_Vector *y)
{
   // informational only; the product is still formed (seq_mv/csr_matvec.c:57-86)
   HYPRE_Int ierr = 0;
   const bool badx = num_cols != x->size;
   const bool bady = num_rows != y->size || num_rows != b->size;
   if (badx) { ierr = 1; }
   if (bady) { ierr = 2; }
   if (badx && bady) { ierr = 3; }
   return ierr;
}

// y(:,v) = alpha * A * x(:,v) + beta * b(:,v) on the compute stream, no sync.
static void spmv_device_core(HYPRE_Complex alpha, hypre_CSRMatrix *A, const HYPRE_Complex *x,
                             HYPRE_Complex beta, const HYPRE_Complex *b, HYPRE_Complex *y, HYPRE_Int fill)
{
   hipStream_t s = stream();
   const HYPRE_Int nr = A->num_rows;
   if (nr <= 0) { return; }
   if (A->num_nonzeros <= 0 || alpha == 0.0)
   {
      // y = beta*b
      if (beta == 0.0) { launch_set(y, 0.0, (size_t) nr, s); }
      else if (b == y) { if (beta != 1.0) { launch_scale(y, beta, (size_t) nr, s); } }
      else { launch_scale_copy(beta, b, y, (size_t) nr, s); }
      return;
   }
   SpmvArgs a{};
   a.Ai = A->i; a.Aj = A->j; a.Aa = A->data; a.Aa32 = nullptr;
   a.x = x; a.b = b; a.y = y; a.d = nullptr; a.marker = nullptr; a.marker_val = 0;
   a.alpha = alpha; a.beta = beta; a.fill = fill; a.row_offset = 0;
   spmv_default_flags(a);

   // sparse-row path: only a few rows hold entries (off-diagonal blocks)
   if (fill == HYPRE_SPMV_FILL_WHOLE && A->rownnz && (double) A->num_rownnz < 0.7 * (double) nr)
   {
      if (handle().fp32_values) { a.Aa32 = fp32_values_of(A); }
      if (beta == 0.0) { launch_set(y, 0.0, (size_t) nr, s); }
      else if (b == y) { if (beta != 1.0) { launch_scale(y, beta, (size_t) nr, s); } }
      else { launch_scale_copy(beta, b, y, (size_t) nr, s); }
      launch_spmv_rownnz(A->rownnz, A->num_rownnz, a, s);
      return;
   }
   SpmvPlan *plan = get_plan(A);
   launch_spmv(plan, a, OP_AXPBY, s);
}

// Y = alpha A X + beta B for the nv columns of a multivector: one pass over the matrix where the plan's form allows it
// (spmv_xs_mv_kernel; the reference sums NV products per entry too: csr_matvec.c:117-380, csr_spmv_device.c:37-134),
// else column by column.  Same bits either way.
static void spmv_device_columns(HYPRE_Complex alpha, hypre_CSRMatrix *A, const HYPRE_Complex *x, size_t xstride,
                                HYPRE_Complex beta, const HYPRE_Complex *b, size_t bstride, HYPRE_Complex *y, size_t ystride,
                                HYPRE_Int nv, HYPRE_Int fill)
{
   const bool plain = A->num_rows > 0 && A->num_nonzeros > 0 && alpha != 0.0 && fill == HYPRE_SPMV_FILL_WHOLE &&
                      !(A->rownnz && (double) A->num_rownnz < 0.7 * (double) A->num_rows);
   if (nv > 1 && plain && spmv_fused_multivectors())
   {
      SpmvArgs a{};
      a.Ai = A->i; a.Aj = A->j; a.Aa = A->data; a.Aa32 = nullptr;
      a.x = x; a.b = b; a.y = y; a.d = nullptr; a.marker = nullptr; a.marker_val = 0;
      a.alpha = alpha; a.beta = beta; a.fill = fill; a.row_offset = 0;
      spmv_default_flags(a);
      if (launch_spmv_mv(get_plan(A), a, nv, (long) xstride, (long) bstride, (long) ystride, stream())) { return; }
   }
   for (HYPRE_Int v = 0; v < nv; v++)
   {
      spmv_device_core(alpha, A, x + (size_t) v * xstride, beta, b + (size_t) v * bstride, y + (size_t) v * ystride, fill);
   }
}

// Multivector products in one pass over the matrix (default) or column by column: speed only, same bits.
HYPRE_Int hypre_amd_SpmvSetFusedMultivectors(HYPRE_Int on)
{
   hamd::spmv_fused_multivectors() = on != 0;
   return hypre_error_flag;
}
// launches of the fused multivector kernel since the library was loaded (tests: which path served a product)
HYPRE_Int hypre_amd_SpmvFusedMultivectorLaunches(void) { return (HYPRE_Int) hamd::spmv_mv_launches(); }

HYPRE_Int hypre_CSRMatrixMatvecDevice(HYPRE_Int trans, HYPRE_Complex alpha, hypre_CSRMatrix *A,
                                      hypre_Vector *x, HYPRE_Complex beta, hypre_Vector *b,
                                      hypre_Vector *y, HYPRE_Int offset)
{
   HYPRE_AMD_REQUIRE_DEVICE(A->memory_location, "hypre_CSRMatrixMatvecDevice(A)");
   HYPRE_AMD_REQUIRE_DEVICE(x->memory_location, "hypre_CSRMatrixMatvecDevice(x)");
   HYPRE_AMD_REQUIRE_DEVICE(y->memory_location, "hypre_CSRMatrixMatvecDevice(y)");
   HYPRE_AMD_REQUIRE_DEVICE(b->memory_location, "hypre_CSRMatrixMatvecDevice(b)");
   if (offset != 0)
   {
      // the reference's device path asserts offset == 0 (csr_matvec_device.c:121)
      hypre_error_w_msg(HYPRE_ERROR_GENERIC, "hypre_CSRMatrixMatvecDevice: offset != 0 is not supported on the device");
      return hypre_error_flag;
   }
   if (x->num_vectors != y->num_vectors || x->num_vectors != b->num_vectors)
   {
      hypre_error_w_msg(HYPRE_ERROR_GENERIC, "hypre_CSRMatrixMatvecDevice: num_vectors mismatch");
      return hypre_error_flag;
   }
   hypre_CSRMatrix *M = A;
   if (trans)
   {
      SpmvPlan *plan = get_plan(A);
      if (!plan->AT) { hypre_CSRMatrixTranspose(A, &plan->AT, 1); if (plan->owned) { mark_owned(plan->AT); } }
      M = plan->AT;
   }
   hypre_Vector *x_tmp = nullptr;
   const HYPRE_Complex *xd = x->data;
   if (x->data == y->data)
   {
      // aliasing: the host reference deep-clones x (csr_matvec.c:109-113)
      x_tmp = hypre_SeqVectorCloneDeep(x);
      xd = x_tmp->data;
   }
   const bool by_columns = x->num_vectors == 1 || (x->idxstride == 1 && y->idxstride == 1 && b->idxstride == 1);
   if (!by_columns)
   {
      hypre_error_w_msg(HYPRE_ERROR_GENERIC, "hypre_CSRMatrixMatvecDevice: row-wise multivector storage is not supported");
   }
   else
   {
      spmv_device_columns(alpha, M, xd, (size_t) x->vecstride, beta, b->data, (size_t) b->vecstride, y->data, (size_t) y->vecstride,
                          x->num_vectors, HYPRE_SPMV_FILL_WHOLE);
   }
   if (handle().sync_compute && !(x->data == y->data))
   {
      // synchronous public product: a plan the kernels found out of date is rebuilt (get_plan raises the error) and the
      // product repeated, so that what the caller reads is right.  (Without the end-of-call synchronisation the flag is
      // seen by the next call that asks for the plan.)  In place with beta != 0 (hypre_CSRMatrixMatvec: b is y) the first,
      // wrong product has already overwritten the operand the repeat would need: then nothing is repeated, the error is
      // raised here and y is NOT valid.
      HIP_CHECK(hipStreamSynchronize(stream()));
      if (plan_is_stale(M) && b->data == y->data && beta != 0.0)
      {
         (void) get_plan(M);                 // raises HYPRE_ERROR_GENERIC, rebuilds the plan for the next call
      }
      else if (plan_is_stale(M))
      {
         if (by_columns)
         {
            spmv_device_columns(alpha, M, xd, (size_t) x->vecstride, beta, b->data, (size_t) b->vecstride, y->data, (size_t) y->vecstride,
                                x->num_vectors, HYPRE_SPMV_FILL_WHOLE);
         }
      }
   }
   if (x_tmp) { HIP_CHECK(hipStreamSynchronize(stream())); hypre_SeqVectorDestroy(x_tmp); }
   maybe_sync();
   return hypre_error_flag;
}

HYPRE_Int hypre_CSRMatrixMatvecOutOfPlace(HYPRE_Complex alpha, hypre_CSRMatrix *A, hypre_Vector *x,
                                          HYPRE_Complex beta, hypre_Vector *b, hypre_Vector *y,
                                          HYPRE_Int offset)
{
   const HYPRE_Int ierr = matvec_ierr(A->num_rows - offset, A->num_cols, x, b, y);
   hypre_CSRMatrixMatvecDevice(0, alpha, A, x, beta, b, y, offset);
   return ierr;
}

HYPRE_Int hypre_CSRMatrixMatvec(HYPRE_Complex alpha, hypre_CSRMatrix *A, hypre_Vector *x,
                                HYPRE_Complex beta, hypre_Vector *y)
{
   return hypre_CSRMatrixMatvecOutOfPlace(alpha, A, x, beta, y, y, 0);
}

HYPRE_Int hypre_CSRMatrixMatvecT(HYPRE_Complex alpha, hypre_CSRMatrix *A, hypre_Vector *x,
                                 HYPRE_Complex beta, hypre_Vector *y)
{
   // ierr as seq_mv/csr_matvec.c:952-966 (roles of rows/cols swapped)
   HYPRE_Int ierr = 0;
   const bool badx = A->num_rows != x->size;
   const bool bady = A->num_cols != y->size;
   if (badx) { ierr = 1; }
   if (bady) { ierr = 2; }
   if (badx && bady) { ierr = 3; }
   hypre_CSRMatrixMatvecDevice(1, alpha, A, x, beta, y, y, 0);
   return ierr;
}

HYPRE_Int hypre_CSRMatrixSpMVDevice(HYPRE_Int trans, HYPRE_Complex alpha, hypre_CSRMatrix *B,
                                    hypre_Vector *x, HYPRE_Complex beta, hypre_Vector *y, HYPRE_Int fill)
{
   HYPRE_AMD_REQUIRE_DEVICE(B->memory_location, "hypre_CSRMatrixSpMVDevice(B)");
   HYPRE_AMD_REQUIRE_DEVICE(x->memory_location, "hypre_CSRMatrixSpMVDevice(x)");
   HYPRE_AMD_REQUIRE_DEVICE(y->memory_location, "hypre_CSRMatrixSpMVDevice(y)");
   if (x->num_vectors != y->num_vectors)
   {
      hypre_error_w_msg(HYPRE_ERROR_GENERIC, "num_vectors_x != num_vectors_y");
      return hypre_error_flag;
   }
   hypre_CSRMatrix *M = B;
   if (trans)
   {
      SpmvPlan *plan = get_plan(B);
      if (!plan->AT) { hypre_CSRMatrixTranspose(B, &plan->AT, 1); if (plan->owned) { mark_owned(plan->AT); } }
      M = plan->AT;
   }
   spmv_device_columns(alpha, M, x->data, (size_t) x->vecstride, beta, y->data, (size_t) y->vecstride, y->data, (size_t) y->vecstride,
                       x->num_vectors, fill);
   return hypre_error_flag;
}

// ===========================================================================
// BLAS-1
// ===========================================================================
static inline size_t vlen(hypre_Vector *v) { return (size_t) v->size * (size_t) v->num_vectors; }

HYPRE_Int hypre_SeqVectorSetConstantValuesDevice(hypre_Vector *v, HYPRE_Complex value)
{
   launch_set(v->data, value, vlen(v), stream());
   maybe_sync();
   return hypre_error_flag;
}
HYPRE_Int hypre_SeqVectorSetConstantValues(hypre_Vector *v, HYPRE_Complex value)
{
   if (v->memory_location == HYPRE_MEMORY_HOST)
   {
      // vectors are filled on the host while problems are assembled; this is
      // data preparation, not part of the solve path
      for (size_t i = 0; i < vlen(v); i++) { v->data[i] = value; }
      return hypre_error_flag;
   }
   return hypre_SeqVectorSetConstantValuesDevice(v, value);
}

HYPRE_Int hypre_SeqVectorCopy(hypre_Vector *x, hypre_Vector *y)
{
   const size_t n = std::min(vlen(x), vlen(y));
   if (x->memory_location == HYPRE_MEMORY_DEVICE && y->memory_location == HYPRE_MEMORY_DEVICE)
   {
      launch_copy(y->data, x->data, n, stream());
      maybe_sync();
   }
   else
   {
      hypre_TMemcpy(y->data, x->data, HYPRE_Complex, n, y->memory_location, x->memory_location);
   }
   return hypre_error_flag;
}

HYPRE_Int hypre_SeqVectorScaleDevice(HYPRE_Complex alpha, hypre_Vector *y)
{
   launch_scale(y->data, alpha, vlen(y), stream());
   maybe_sync();
   return hypre_error_flag;
}
HYPRE_Int hypre_SeqVectorScale(HYPRE_Complex alpha, hypre_Vector *y)
{
   // shortcuts of seq_mv/vector.c:661-669
   if (alpha == 1.0) { return hypre_error_flag; }
   if (alpha == 0.0) { return hypre_SeqVectorSetConstantValues(y, 0.0); }
   HYPRE_AMD_REQUIRE_DEVICE(y->memory_location, "hypre_SeqVectorScale");
   return hypre_SeqVectorScaleDevice(alpha, y);
}

HYPRE_Int hypre_SeqVectorAxpyDevice(HYPRE_Complex alpha, hypre_Vector *x, hypre_Vector *y)
{
   launch_axpy(alpha, x->data, y->data, vlen(x), stream());
   maybe_sync();
   return hypre_error_flag;
}
HYPRE_Int hypre_SeqVectorAxpy(HYPRE_Complex alpha, hypre_Vector *x, hypre_Vector *y)
{
   HYPRE_AMD_REQUIRE_DEVICE(x->memory_location, "hypre_SeqVectorAxpy(x)");
   HYPRE_AMD_REQUIRE_DEVICE(y->memory_location, "hypre_SeqVectorAxpy(y)");
   return hypre_SeqVectorAxpyDevice(alpha, x, y);
}

HYPRE_Int hypre_SeqVectorAxpyzDevice(HYPRE_Complex alpha, hypre_Vector *x, HYPRE_Complex beta,
                                     hypre_Vector *y, hypre_Vector *z)
{
   launch_axpyz(alpha, x->data, beta, y->data, z->data, vlen(x), stream());
   maybe_sync();
   return hypre_error_flag;
}
HYPRE_Int hypre_SeqVectorAxpyz(HYPRE_Complex alpha, hypre_Vector *x, HYPRE_Complex beta,
                               hypre_Vector *y, hypre_Vector *z)
{
   HYPRE_AMD_REQUIRE_DEVICE(x->memory_location, "hypre_SeqVectorAxpyz(x)");
   HYPRE_AMD_REQUIRE_DEVICE(y->memory_location, "hypre_SeqVectorAxpyz(y)");
   HYPRE_AMD_REQUIRE_DEVICE(z->memory_location, "hypre_SeqVectorAxpyz(z)");
   return hypre_SeqVectorAxpyzDevice(alpha, x, beta, y, z);
}

HYPRE_Real hypre_SeqVectorInnerProdDevice(hypre_Vector *x, hypre_Vector *y)
{
   hipStream_t s = stream();
   double *d_out = reduce_scratch(2048);
   launch_dot(x->data, y->data, vlen(x), d_out, s);
   double *h = handle().h_reduce;
   HIP_CHECK(hipMemcpyAsync(h, d_out, sizeof(double), hipMemcpyDeviceToHost, s));
   HIP_CHECK(hipStreamSynchronize(s));
   return h[0];
}
HYPRE_Real hypre_SeqVectorInnerProd(hypre_Vector *x, hypre_Vector *y)
{
   if (x->memory_location != HYPRE_MEMORY_DEVICE || y->memory_location != HYPRE_MEMORY_DEVICE)
   {
      hypre_error_w_msg(HYPRE_ERROR_GENERIC, "hypre_SeqVectorInnerProd: operand is not in device memory; host execution is not part of this library");
      return 0.0;
   }
   return hypre_SeqVectorInnerProdDevice(x, y);
}

HYPRE_Int hypre_SeqVectorElmdivpyDevice(hypre_Vector *x, hypre_Vector *b, hypre_Vector *y,
                                        HYPRE_Int *marker, HYPRE_Int marker_val)
{
   launch_elmdivpy(x->data, b->data, y->data, marker, marker_val, (size_t) b->size, stream());
   maybe_sync();
   return hypre_error_flag;
}
HYPRE_Int hypre_SeqVectorElmdivpy(hypre_Vector *x, hypre_Vector *b, hypre_Vector *y)
{
   HYPRE_AMD_REQUIRE_DEVICE(x->memory_location, "hypre_SeqVectorElmdivpy(x)");
   HYPRE_AMD_REQUIRE_DEVICE(y->memory_location, "hypre_SeqVectorElmdivpy(y)");
   return hypre_SeqVectorElmdivpyDevice(x, b, y, nullptr, -1);
}
HYPRE_Int hypre_SeqVectorElmdivpyMarked(hypre_Vector *x, hypre_Vector *b, hypre_Vector *y,
                                        HYPRE_Int *marker, HYPRE_Int marker_val)
{
   HYPRE_AMD_REQUIRE_DEVICE(x->memory_location, "hypre_SeqVectorElmdivpyMarked(x)");
   HYPRE_AMD_REQUIRE_DEVICE(y->memory_location, "hypre_SeqVectorElmdivpyMarked(y)");
   return hypre_SeqVectorElmdivpyDevice(x, b, y, marker, marker_val);
}

HYPRE_Int hypreDevice_IVAXPY(HYPRE_Int n, HYPRE_Complex *a, HYPRE_Complex *x, HYPRE_Complex *y)
{
   if (n > 0) { launch_elmdivpy(x, a, y, nullptr, 0, (size_t) n, stream()); }
   return hypre_error_flag;
}
HYPRE_Int hypreDevice_IVAXPYMarked(HYPRE_Int n, HYPRE_Complex *a, HYPRE_Complex *x, HYPRE_Complex *y,
                                   HYPRE_Int *marker, HYPRE_Int marker_val)
{
   if (n > 0) { launch_elmdivpy(x, a, y, marker, marker_val, (size_t) n, stream()); }
   return hypre_error_flag;
}
HYPRE_Int hypreDevice_DiagScaleVector2(HYPRE_Int num_vectors, HYPRE_Int num_rows, HYPRE_Complex *diag,
                                       HYPRE_Complex *x, HYPRE_Complex beta, HYPRE_Complex *y,
                                       HYPRE_Complex *z, HYPRE_Int computeY)
{
   for (HYPRE_Int v = 0; v < num_vectors; v++)
   {
      const size_t o = (size_t) v * (size_t) num_rows;
      launch_diagscale2(diag, x + o, beta, y ? y + o : nullptr, z + o, computeY, (size_t) num_rows, stream());
   }
   return hypre_error_flag;
}

}  // extern "C"
