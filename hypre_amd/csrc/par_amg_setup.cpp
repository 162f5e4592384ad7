// hypre_amd — BoomerAMG setup (hierarchy construction) on the host.
//
// The solve phase is the GPU hot path; this file is the step before it (scope
// table row "next"): it exists so that a 256^3 hierarchy can be produced where
// the benchmark runs.  Every algorithm follows the reference's host code in
// loop order and tie-breaking so that hierarchies (and therefore iteration
// counts and complexities) match the reference driver's on the same input:
//   parcsr_ls/par_strength.c:75-530        classical strength of connection
//   parcsr_ls/par_coarsen.c:2101-2810      PMIS (measures, independent sets)
//   parcsr_ls/par_coarsen.c:911-1400       Ruge-Stueben first pass (HMIS = RS pass + PMIS)
//   parcsr_ls/par_indepset.c               random tie-breakers (utilities/random.c LCG)
//   parcsr_ls/par_coarse_parms.c           coarse numbering
//   parcsr_ls/par_lr_interp.c:1024-1700    extended+i interpolation
//   parcsr_mv/par_csr_matrix.c:2874-3400   truncation of P (max elements / threshold, rescaled)
//   parcsr_ls/par_rap.c:30-2000            Galerkin product RAP
//   parcsr_ls/ams.c:527-830                smoother diagonals
//   parcsr_ls/par_gauss_elim.c:25-300      dense coarsest-level operator
//   parcsr_ls/par_amg_setup.c:28-3560      level loop, stopping rules, work vectors
//
// Rows are processed in parallel with OpenMP where every row's result is
// independent of the others (strength, interpolation, RAP); the per-row
// arithmetic and entry order are those of the sequential reference loops.
#include "amg_internal.hpp"
#include <algorithm>
#include <cmath>
#include <omp.h>
#include <unordered_map>

using namespace hamd;

namespace {

// ---------------------------------------------------------------------------
// host halo helpers over a comm package
// ---------------------------------------------------------------------------
template <class T> struct HaloJob;
template <> struct HaloJob<double>       { static constexpr int fwd = 1,  rev = 2; };
template <> struct HaloJob<HYPRE_Int>    { static constexpr int fwd = 11, rev = 12; };
template <> struct HaloJob<HYPRE_BigInt> { static constexpr int fwd = 21; };      // ghost -> owner is never needed for indices

// owner values -> ghost array (offd column order)
template <class T>
void halo_forward(hypre_ParCSRCommPkg *pkg, const T *local, T *ghost)
{
   if (!pkg) { return; }
   const HYPRE_Int tot = pkg->send_map_starts[pkg->num_sends];
   std::vector<T> buf((size_t) std::max(tot, 1));
   for (HYPRE_Int k = 0; k < tot; k++) { buf[(size_t) k] = local[pkg->send_map_elmts[k]]; }
   hypre_ParCSRCommHandle *h = hypre_ParCSRCommHandleCreate(HaloJob<T>::fwd, pkg, buf.data(), ghost);
   hypre_ParCSRCommHandleDestroy(h);
}
// ghost values -> owners' buffer laid out like send_map_elmts
template <class T>
void halo_reverse(hypre_ParCSRCommPkg *pkg, const T *ghost, T *buf)
{
   if (!pkg) { return; }
   hypre_ParCSRCommHandle *h = hypre_ParCSRCommHandleCreate(HaloJob<T>::rev, pkg, (void *) ghost, buf);
   hypre_ParCSRCommHandleDestroy(h);
}

int comm_size(MPI_Comm c) { HYPRE_Int n; hypre_MPI_Comm_size(c, &n); return n; }
int comm_rank(MPI_Comm c) { HYPRE_Int r; hypre_MPI_Comm_rank(c, &r); return r; }

HYPRE_BigInt global_sum_big(MPI_Comm comm, HYPRE_BigInt v)
{
   const hypre_amd_CommOps *o = comm_ops(comm);
   if (o && o->size > 1)
   {
      double d = (double) v;
      o->allreduce_sum(o->ctx, &d, 1, 0, nullptr);
      return (HYPRE_BigInt) llround(d);
   }
   return v;
}

// utilities/random.c: Park-Miller minimal standard generator
struct HypreRand
{
   HYPRE_Int seed = 13579;
   void reseed(HYPRE_Int s)
   {
      const HYPRE_Int m = 2147483647;
      if (s < 1) { s = 1; } else if (s >= m) { s = m - 1; }
      seed = s;
   }
   double next()
   {
      const HYPRE_Int a = 16807, m = 2147483647, q = 127773, r = 2836;
      const HYPRE_Int high = seed / q, low = seed % q;
      const HYPRE_Int test = a * low - r * high;
      seed = test > 0 ? test : test + m;
      return (double) seed / (double) m;
   }
};

inline int num_threads_avail() { return omp_get_max_threads(); }


// contiguous row chunk of thread t out of T
inline void chunk(int n, int T, int t, int *b, int *e)
{
   const long long per = n / T, rest = n % T;
   *b = (int) (t * per + std::min<long long>(t, rest));
   *e = (int) (*b + per + (t < rest ? 1 : 0));
}

// Sparse accumulator index for one matrix row at a time: integer key -> 64-bit value, emptied
// after every row by revisiting only the slots that were used.  It stands in for the reference's
// per-thread marker arrays of full vector length (par_rap.c P_marker / A_marker, par_lr_interp.c
// P_marker): those cost num_threads x n words to allocate and fill, which dominates the setup on a
// many-core host; the insertion ORDER of a row's entries, which fixes the layout of RAP and P, is
// unchanged.
struct RowMap
{
   std::vector<HYPRE_Int> key;
   std::vector<long long> val;
   std::vector<HYPRE_Int> used;
   HYPRE_Int              mask;
   explicit RowMap(HYPRE_Int cap = 1024) : key((size_t) cap, -1), val((size_t) cap, 0), mask(cap - 1) { used.reserve((size_t) cap / 2); }
   static inline HYPRE_Int hash(HYPRE_Int k) { return (HYPRE_Int) (((unsigned) k * 2654435761u) >> 7); }
   // value stored for k, or `absent`
   inline long long get(HYPRE_Int k, long long absent) const
   {
      HYPRE_Int h = hash(k) & mask;
      while (key[(size_t) h] != -1)
      {
         if (key[(size_t) h] == k) { return val[(size_t) h]; }
         h = (h + 1) & mask;
      }
      return absent;
   }
   // set (insert or overwrite)
   inline void set(HYPRE_Int k, long long v)
   {
      HYPRE_Int h = hash(k) & mask;
      while (key[(size_t) h] != -1)
      {
         if (key[(size_t) h] == k) { val[(size_t) h] = v; return; }
         h = (h + 1) & mask;
      }
      key[(size_t) h] = k; val[(size_t) h] = v; used.push_back(h);
      if (2 * used.size() > key.size()) { grow(); }
   }
   void grow()
   {
      std::vector<HYPRE_Int> ok; std::vector<long long> ov;
      ok.reserve(used.size()); ov.reserve(used.size());
      for (HYPRE_Int h : used) { ok.push_back(key[(size_t) h]); ov.push_back(val[(size_t) h]); }
      const size_t cap = key.size() * 2;
      key.assign(cap, -1); val.assign(cap, 0); mask = (HYPRE_Int) cap - 1; used.clear();
      for (size_t q = 0; q < ok.size(); q++)
      {
         HYPRE_Int h = hash(ok[q]) & mask;
         while (key[(size_t) h] != -1) { h = (h + 1) & mask; }
         key[(size_t) h] = ok[q]; val[(size_t) h] = ov[q]; used.push_back(h);
      }
   }
   inline void clear()
   {
      for (HYPRE_Int h : used) { key[(size_t) h] = -1; }
      used.clear();
   }
};

// Marker for the interpolation rows of one thread.  When everything a thread's rows can reach lies
// in a narrow index window (banded orderings: grids), a plain array over that window is used and
// keeps the reference's semantics of stale entries (older positions / tags never match); otherwise
// the per-row hash map, emptied after every row (absent reads as -1, which never matches either).
struct RowMarker
{
   bool                   windowed = false;
   HYPRE_Int              lo = 0;
   std::vector<long long> w;
   RowMap                 h;
   RowMarker() : h(1024) {}
   void use_window(HYPRE_Int lo_, HYPRE_Int hi_) { windowed = true; lo = lo_; w.assign((size_t) (hi_ - lo_ + 1), -1); }
   inline long long get(HYPRE_Int k) const { return windowed ? w[(size_t) (k - lo)] : h.get(k, -1); }
   inline void set(HYPRE_Int k, long long v) { if (windowed) { w[(size_t) (k - lo)] = v; } else { h.set(k, v); } }
   inline void end_row() { if (!windowed) { h.clear(); } }
};

}  // namespace

// set by hypre_BoomerAMGSetup while it builds a hierarchy whose home is device memory
static bool g_setup_targets_device = false;
static int  g_device_rap_on = [] { const char *e = getenv("HYPRE_AMD_SETUP_DEVICE_RAP"); return e ? atoi(e) : 1; }();
static int  g_device_rap_min_rows = 20000;
static int  g_device_rap_count = 0;          // products formed on the device since the last query

// Galerkin products of single-rank setups whose hierarchy lives in device memory are formed on the device (same bits as
// the host loop): on = 0 keeps the host loop, min_rows = smallest fine level worth the transfers (negative: unchanged).
// Returns the number of products formed on the device since the previous call.
extern "C" HYPRE_Int hypre_amd_SetSetupDeviceRAP(HYPRE_Int on, HYPRE_Int min_rows)
{
   if (on >= 0) { g_device_rap_on = on; }
   if (min_rows >= 0) { g_device_rap_min_rows = min_rows; }
   const int c = g_device_rap_count;
   g_device_rap_count = 0;
   return c;
}
static bool device_rap_allowed() { return g_setup_targets_device; }
static bool device_setup_allowed() { return g_setup_targets_device; }

// Device twins of host matrices made during a device-targeted setup (operators uploaded for, or produced by, the device
// kernels).  A later step that needs the same matrix on the device takes the twin instead of uploading again, and when
// the finished hierarchy moves to device memory a matrix with a twin simply adopts its arrays.  Twins nobody adopted are
// freed when the setup ends.
static std::unordered_map<const hypre_CSRMatrix *, hypre_CSRMatrix *> &device_twins()
{
   static std::unordered_map<const hypre_CSRMatrix *, hypre_CSRMatrix *> t;
   return t;
}
static hypre_CSRMatrix *device_twin_of(hypre_CSRMatrix *host, int with_data)
{
   if (host->memory_location == HYPRE_MEMORY_DEVICE) { return host; }      // born on the device (an interpolation operator)
   auto &t = device_twins();
   auto it = t.find(host);
   if (it != t.end()) { return it->second; }
   hypre_CSRMatrix *d = hypre_CSRMatrixClone_v2(host, with_data, HYPRE_MEMORY_DEVICE);
   t[host] = d;
   return d;
}
static void drop_device_markers();
static void drop_device_twins()
{
   for (auto &kv : device_twins()) { hypre_CSRMatrixDestroy(kv.second); }
   device_twins().clear();
   drop_device_markers();
}
// device CSR from arrays that already live on the device (ownership passes to the matrix)
static hypre_CSRMatrix *wrap_device_csr(HYPRE_Int nr, HYPRE_Int ncl, HYPRE_Int nnz, int *i, int *j, double *a)
{
   hypre_CSRMatrix *m = hypre_CSRMatrixCreate(nr, ncl, nnz);
   m->i = i; m->j = j; m->data = a; m->memory_location = HYPRE_MEMORY_DEVICE; m->owns_data = 1;
   return m;
}
// host -> device for one block of the finished hierarchy: adopt the twin's arrays when there is one
static void place_on_device(hypre_CSRMatrix *M)
{
   auto &t = device_twins();
   auto it = M ? t.find(M) : t.end();
   if (it == t.end() || M->memory_location != HYPRE_MEMORY_HOST) { if (M) { hypre_CSRMatrixMigrate(M, HYPRE_MEMORY_DEVICE); } return; }
   hypre_CSRMatrix *d = it->second;
   t.erase(it);
   hypre_Free(M->i, HYPRE_MEMORY_HOST); hypre_Free(M->j, HYPRE_MEMORY_HOST); hypre_Free(M->data, HYPRE_MEMORY_HOST);
   M->i = d->i; M->j = d->j; M->data = d->data;
   if (M->rownnz)
   {
      // the list of non-empty rows (ghost blocks carry one) moves with the arrays
      HYPRE_Int *dr = hypre_TAlloc(HYPRE_Int, (size_t) std::max(M->num_rownnz, 1), HYPRE_MEMORY_DEVICE);
      hypre_TMemcpy(dr, M->rownnz, HYPRE_Int, (size_t) M->num_rownnz, HYPRE_MEMORY_DEVICE, HYPRE_MEMORY_HOST);
      hypre_Free(M->rownnz, HYPRE_MEMORY_HOST);
      M->rownnz = dr;
   }
   M->memory_location = HYPRE_MEMORY_DEVICE;
   d->i = nullptr; d->j = nullptr; d->data = nullptr;
   hypre_CSRMatrixDestroy(d);
}
// device -> fresh host arrays of n items: the destination pages are touched by all threads first (a copy into untouched
// memory faults them in one by one on one thread: 1.8 GB took 0.4 s that way)
static void download_bytes(void *host, const void *dev, size_t bytes)
{
   if (bytes > (size_t) 1 << 24)
   {
      char *p = (char *) host;
#pragma omp parallel for schedule(static)
      for (long long o = 0; o < (long long) bytes; o += 4096) { p[o] = 0; }
   }
   hypre_Memcpy(host, (void *) dev, bytes, HYPRE_MEMORY_HOST, HYPRE_MEMORY_DEVICE);
}
#define download(host, dev, n) download_bytes((host), (dev), sizeof(*(host)) * (size_t) (n))
static int  g_device_interp_on = [] { const char *e = getenv("HYPRE_AMD_SETUP_DEVICE_INTERP"); return e ? atoi(e) : 1; }();
static int  g_device_interp_count = 0;
static bool g_interp_host_once = false;      // the next interpolation goes to the host loop (the device kernel just declined it)
// same for the extended+i interpolation: on = 0 keeps the host loop; returns the number of operators built on the device
// since the previous call
extern "C" HYPRE_Int hypre_amd_SetSetupDeviceInterp(HYPRE_Int on)
{
   if (on >= 0) { g_device_interp_on = on; }
   const int c = g_device_interp_count;
   g_device_interp_count = 0;
   return c;
}


// Strength of connection and PMIS of single-rank, scalar levels on the device as well (setup_kernels.hip: the host
// routines' results array for array), so that a coarse operator formed on the device never has to come back:
// on = 0 keeps the host loops.  Returns the number of levels coarsened on the device since the previous call.
static int  g_device_coarsen_on = [] { const char *e = getenv("HYPRE_AMD_SETUP_DEVICE_COARSEN"); return e ? atoi(e) : 1; }();
static int  g_device_coarsen_count = 0;
static bool g_level_on_device = false;       // set by the setup loop for the level it is working on
// distributed levels (several ranks) on the device as well: on = 0 keeps them on the host (OpenMP loops)
static int  g_device_dist_on = [] { const char *e = getenv("HYPRE_AMD_SETUP_DEVICE_DIST"); return e ? atoi(e) : 1; }();
extern "C" HYPRE_Int hypre_amd_SetSetupDeviceDist(HYPRE_Int on)
{
   const int was = g_device_dist_on;
   if (on >= 0) { g_device_dist_on = on; }
   return was;
}
extern "C" HYPRE_Int hypre_amd_SetSetupDeviceCoarsen(HYPRE_Int on)
{
   if (on >= 0) { g_device_coarsen_on = on; }
   const int c = g_device_coarsen_count;
   g_device_coarsen_count = 0;
   return c;
}
// device copies of C/F marker arrays made during the setup (host array -> device array); the finished hierarchy adopts them
static std::unordered_map<const HYPRE_Int *, HYPRE_Int *> &device_markers()
{
   static std::unordered_map<const HYPRE_Int *, HYPRE_Int *> t;
   return t;
}
static void drop_device_markers()
{
   for (auto &kv : device_markers()) { hypre_Free(kv.second, HYPRE_MEMORY_DEVICE); }
   device_markers().clear();
}
static HYPRE_Int *device_marker_of(const HYPRE_Int *host, HYPRE_Int n)
{
   auto &t = device_markers();
   auto it = t.find(host);
   if (it != t.end()) { return it->second; }
   HYPRE_Int *dd = hypre_TAlloc(HYPRE_Int, (size_t) std::max(n, 1), HYPRE_MEMORY_DEVICE);
   hypre_TMemcpy(dd, host, HYPRE_Int, (size_t) n, HYPRE_MEMORY_DEVICE, HYPRE_MEMORY_HOST);
   t[host] = dd;
   return dd;
}
namespace hamd {
hypre_CSRMatrix *setup_device_twin_of(hypre_CSRMatrix *host, int with_data) { return device_twin_of(host, with_data); }
hypre_CSRMatrix *setup_wrap_device_csr(HYPRE_Int nr, HYPRE_Int ncl, HYPRE_Int nnz, int *i, int *j, double *a) { return wrap_device_csr(nr, ncl, nnz, i, j, a); }
void setup_register_device_marker(const HYPRE_Int *host, HYPRE_Int *dev)
{
   auto &mk = device_markers();
   auto it = mk.find(host);
   if (it != mk.end()) { hypre_Free(it->second, HYPRE_MEMORY_DEVICE); }
   mk[host] = dev;
}
HYPRE_Int *setup_device_marker_of(const HYPRE_Int *host, HYPRE_Int n) { return device_marker_of(host, n); }
}  // namespace hamd
// a matrix of the setup goes away: so must the twins registered under its blocks' addresses (the next matrix allocated
// there would inherit them)
static void destroy_with_twins(hypre_ParCSRMatrix *M)
{
   if (!M) { return; }
   auto &t = device_twins();
   for (hypre_CSRMatrix *blk : {M->diag, M->offd, M->diagT, M->offdT})
   {
      auto it = blk ? t.find(blk) : t.end();
      if (it != t.end()) { hypre_CSRMatrixDestroy(it->second); t.erase(it); }
   }
   hypre_ParCSRMatrixDestroy(M);
}
// A matrix born on the device is needed by a host loop after all: fetch it; its device arrays stay on as its twin.
static void make_host_resident(hypre_CSRMatrix *M)
{
   if (!M || M->memory_location != HYPRE_MEMORY_DEVICE) { return; }
   const HYPRE_Int nr = M->num_rows, nnz = M->num_nonzeros;
   HYPRE_Int *hi = hypre_TAlloc(HYPRE_Int, (size_t) nr + 1, HYPRE_MEMORY_HOST);
   HYPRE_Int *hj = hypre_TAlloc(HYPRE_Int, (size_t) std::max(nnz, 1), HYPRE_MEMORY_HOST);
   HYPRE_Real *ha = M->data ? hypre_TAlloc(HYPRE_Real, (size_t) std::max(nnz, 1), HYPRE_MEMORY_HOST) : nullptr;
   if (M->i) { download(hi, M->i, (size_t) nr + 1); } else { memset(hi, 0, sizeof(HYPRE_Int) * ((size_t) nr + 1)); }
   if (nnz > 0 && M->j) { download(hj, M->j, (size_t) nnz); }
   if (nnz > 0 && ha) { download(ha, M->data, (size_t) nnz); }
   drop_plan(M);
   device_twins()[M] = wrap_device_csr(nr, M->num_cols, nnz, M->i, M->j, M->data);
   M->i = hi; M->j = hj; M->data = ha;
   if (M->rownnz)
   {
      HYPRE_Int *hr = hypre_TAlloc(HYPRE_Int, (size_t) std::max(M->num_rownnz, 1), HYPRE_MEMORY_HOST);
      hypre_TMemcpy(hr, M->rownnz, HYPRE_Int, (size_t) M->num_rownnz, HYPRE_MEMORY_HOST, HYPRE_MEMORY_DEVICE);
      hypre_Free(M->rownnz, HYPRE_MEMORY_DEVICE);
      M->rownnz = hr;
   }
   M->memory_location = HYPRE_MEMORY_HOST;
}

extern "C" {

// ===========================================================================
// strength of connection (par_strength.c:75-530)
// S keeps, per row and in A's stored order, the off-diagonal columns j with
//   a_ij < theta * min_k a_ik   (a_ii >= 0)   or   a_ij > theta * max_k a_ik   (a_ii < 0);
// rows whose |row sum| exceeds max_row_sum*|a_ii| keep nothing.
// ===========================================================================
HYPRE_Int hypre_BoomerAMGCreateS(hypre_ParCSRMatrix *A, HYPRE_Real theta, HYPRE_Real max_row_sum,
                                 HYPRE_Int num_functions, HYPRE_Int *dof_func, hypre_ParCSRMatrix **S_ptr)
{
   if (num_functions > 1 && !dof_func)
   {
      hypre_error_w_msg(HYPRE_ERROR_GENERIC, "hypre_BoomerAMGCreateS: num_functions > 1 needs the function of every row");
      return hypre_error_flag;
   }
   hypre_CSRMatrix *Ad = A->diag, *Ao = A->offd;
   const HYPRE_Int n = Ad->num_rows;
   const HYPRE_Int nco = Ao->num_cols;
   if (g_level_on_device && num_functions <= 1 && comm_size(A->comm) > 1)
   {
      // a distributed level of a device-targeted setup: both blocks, one pattern over [local | ghost] columns
      return dist_device_create_S(A, theta, max_row_sum, S_ptr);
   }
   if (g_level_on_device && num_functions <= 1 && nco == 0 && Ao->num_nonzeros == 0)
   {
      // this level of a device-targeted setup: one thread per row, the loops below statement for statement; S stays there
      hypre_CSRMatrix *dA = device_twin_of(Ad, 1);
      int *Si = nullptr, *Sj = nullptr, snnz = 0;
      device_strength(n, dA->i, dA->j, dA->data, theta, max_row_sum, &Si, &Sj, &snnz, stream());
      hypre_ParCSRMatrix *S = hypre_ParCSRMatrixCreate(A->comm, A->global_num_rows, A->global_num_rows,
                                                       A->row_starts, A->row_starts, 0, snnz, 0);
      hypre_CSRMatrixDestroy(S->diag);
      S->diag = wrap_device_csr(n, n, snnz, Si, Sj, nullptr);
      hypre_CSRMatrixInitialize_v2(S->offd, 0, HYPRE_MEMORY_HOST);
      *S_ptr = S;
      return hypre_error_flag;
   }
   if (Ad->memory_location != HYPRE_MEMORY_HOST)
   {
      hypre_error_w_msg(HYPRE_ERROR_GENERIC, "hypre_BoomerAMGCreateS: the host loop was handed a device matrix");
      return hypre_error_flag;
   }
   const HYPRE_Int *Adi = Ad->i, *Adj = Ad->j, *Aoi = Ao->i, *Aoj = Ao->j;
   const HYPRE_Real *Ada = Ad->data, *Aoa = Ao->data;

   hypre_ParCSRMatrix *S = hypre_ParCSRMatrixCreate(A->comm, A->global_num_rows, A->global_num_rows,
                                                    A->row_starts, A->row_starts, nco, 0, 0);
   std::vector<HYPRE_Int> sdi((size_t) n + 1, 0), soi((size_t) n + 1, 0);
   // systems, unknown approach (par_strength.c:177-222, 248-290, 343-403): couplings between different
   // functions neither scale the threshold nor count as strong
   const bool sys = num_functions > 1;
   std::vector<HYPRE_Int> dof_offd;
   if (sys && nco)
   {
      if (!A->comm_pkg) { hypre_MatvecCommPkgCreate(A); }
      dof_offd.resize((size_t) nco);
      halo_forward<HYPRE_Int>(A->comm_pkg, dof_func, dof_offd.data());
   }
   auto same_d = [&](HYPRE_Int i, HYPRE_Int k) { return !sys || dof_func[i] == dof_func[Adj[k]]; };
   auto same_o = [&](HYPRE_Int i, HYPRE_Int k) { return !sys || dof_func[i] == dof_offd[(size_t) Aoj[k]]; };
   // pass 1: per-row counts, pass 2: fill (both row-parallel)
   std::vector<char> keep_d((size_t) Adi[n]), keep_o((size_t) Aoi[n]);
#pragma omp parallel for schedule(static)
   for (HYPRE_Int i = 0; i < n; i++)
   {
      const HYPRE_Real diag = Ada[Adi[i]];
      HYPRE_Real row_scale = 0.0, row_sum = diag;
      if (diag < 0)
      {
         for (HYPRE_Int k = Adi[i] + 1; k < Adi[i + 1]; k++) { if (same_d(i, k)) { row_scale = std::max(row_scale, Ada[k]); row_sum += Ada[k]; } }
         for (HYPRE_Int k = Aoi[i]; k < Aoi[i + 1]; k++) { if (same_o(i, k)) { row_scale = std::max(row_scale, Aoa[k]); row_sum += Aoa[k]; } }
      }
      else
      {
         for (HYPRE_Int k = Adi[i] + 1; k < Adi[i + 1]; k++) { if (same_d(i, k)) { row_scale = std::min(row_scale, Ada[k]); row_sum += Ada[k]; } }
         for (HYPRE_Int k = Aoi[i]; k < Aoi[i + 1]; k++) { if (same_o(i, k)) { row_scale = std::min(row_scale, Aoa[k]); row_sum += Aoa[k]; } }
      }
      HYPRE_Int cd = 0, co = 0;
      if (Adi[i + 1] > Adi[i]) { keep_d[(size_t) Adi[i]] = 0; }
      const bool all_weak = (std::fabs(row_sum) > std::fabs(diag) * max_row_sum) && (max_row_sum < 1.0);
      for (HYPRE_Int k = Adi[i] + 1; k < Adi[i + 1]; k++)
      {
         bool strong = false;
         if (!all_weak) { strong = (diag < 0 ? !(Ada[k] <= theta * row_scale) : !(Ada[k] >= theta * row_scale)) && same_d(i, k); }
         keep_d[(size_t) k] = strong; cd += strong;
      }
      for (HYPRE_Int k = Aoi[i]; k < Aoi[i + 1]; k++)
      {
         bool strong = false;
         if (!all_weak) { strong = (diag < 0 ? !(Aoa[k] <= theta * row_scale) : !(Aoa[k] >= theta * row_scale)) && same_o(i, k); }
         keep_o[(size_t) k] = strong; co += strong;
      }
      sdi[(size_t) i + 1] = cd; soi[(size_t) i + 1] = co;
   }
   for (HYPRE_Int i = 0; i < n; i++) { sdi[(size_t) i + 1] += sdi[(size_t) i]; soi[(size_t) i + 1] += soi[(size_t) i]; }
   hypre_CSRMatrix *Sd = S->diag, *So = S->offd;
   Sd->num_nonzeros = sdi[(size_t) n]; So->num_nonzeros = soi[(size_t) n];
   Sd->memory_location = So->memory_location = HYPRE_MEMORY_HOST;
   Sd->i = hypre_TAlloc(HYPRE_Int, n + 1, HYPRE_MEMORY_HOST);
   So->i = hypre_TAlloc(HYPRE_Int, n + 1, HYPRE_MEMORY_HOST);
   Sd->j = hypre_TAlloc(HYPRE_Int, std::max(Sd->num_nonzeros, 1), HYPRE_MEMORY_HOST);
   So->j = hypre_TAlloc(HYPRE_Int, std::max(So->num_nonzeros, 1), HYPRE_MEMORY_HOST);
   memcpy(Sd->i, sdi.data(), sizeof(HYPRE_Int) * ((size_t) n + 1));
   memcpy(So->i, soi.data(), sizeof(HYPRE_Int) * ((size_t) n + 1));
#pragma omp parallel for schedule(static)
   for (HYPRE_Int i = 0; i < n; i++)
   {
      HYPRE_Int p = sdi[(size_t) i];
      for (HYPRE_Int k = Adi[i] + 1; k < Adi[i + 1]; k++) { if (keep_d[(size_t) k]) { Sd->j[p++] = Adj[k]; } }
      p = soi[(size_t) i];
      for (HYPRE_Int k = Aoi[i]; k < Aoi[i + 1]; k++) { if (keep_o[(size_t) k]) { So->j[p++] = Aoj[k]; } }
   }
   if (nco)
   {
      S->col_map_offd = hypre_TAlloc(HYPRE_BigInt, nco, HYPRE_MEMORY_HOST);
      memcpy(S->col_map_offd, A->col_map_offd, sizeof(HYPRE_BigInt) * (size_t) nco);
   }
   *S_ptr = S;
   return hypre_error_flag;
}

// ===========================================================================
// PMIS (par_coarsen.c:2101-2810).  CF_init: 0 plain, 1 second stage of HMIS
// (markers pre-set by the RS pass), 2 sequential random numbers ("pmis1").
// ===========================================================================
static const int C_PT = 1, F_PT = -1, SF_PT = -3, Z_PT = -2, SC_PT = 3, UNDECIDED = -9999;

HYPRE_Int hypre_BoomerAMGCoarsenPMIS(hypre_ParCSRMatrix *S, hypre_ParCSRMatrix *A, HYPRE_Int CF_init,
                                     HYPRE_Int debug_flag, hypre_IntArray **CF_marker_ptr)
{
   (void) debug_flag;
   MPI_Comm comm = S->comm;
   const int nprocs = comm_size(comm), my_id = comm_rank(comm);
   hypre_CSRMatrix *Sd = S->diag, *So = S->offd;
   const HYPRE_Int n = Sd->num_rows, nco = So->num_cols;
   if (Sd->memory_location == HYPRE_MEMORY_DEVICE)
   {
      // S was made on the device (a single-rank level of a device-targeted setup): the sweeps below, one thread per
      // row; the markers come back for the host's bookkeeping and stay on the device for the interpolation
      if (nprocs > 1 && (CF_init == 0 || CF_init == 2))
      {
         // distributed level: the sweeps on the extended graph, the host routine's exchanges (setup_kernels.hip)
         if (*CF_marker_ptr == nullptr)
         {
            *CF_marker_ptr = hypre_IntArrayCreate(n);
            hypre_IntArrayInitialize_v2(*CF_marker_ptr, HYPRE_MEMORY_HOST);
         }
         dist_device_pmis(S, A, CF_init, (*CF_marker_ptr)->data);
         g_device_coarsen_count++;
         return hypre_error_flag;
      }
      if (nprocs > 1 || nco > 0 || (CF_init != 0 && CF_init != 2))
      {
         hypre_error_w_msg(HYPRE_ERROR_GENERIC, "hypre_BoomerAMGCoarsenPMIS: device strength matrix outside the single-rank PMIS path");
         return hypre_error_flag;
      }
      if (*CF_marker_ptr == nullptr)
      {
         *CF_marker_ptr = hypre_IntArrayCreate(n);
         hypre_IntArrayInitialize_v2(*CF_marker_ptr, HYPRE_MEMORY_HOST);
      }
      HYPRE_Int *CFh = (*CF_marker_ptr)->data;
      HYPRE_Int *dCF = hypre_TAlloc(HYPRE_Int, (size_t) std::max(n, 1), HYPRE_MEMORY_DEVICE);
      device_pmis(n, Sd->i, Sd->j, Sd->num_nonzeros, 2747u + (CF_init == 2 ? 0u : (unsigned) my_id),
                  CF_init == 2 ? (unsigned long long) S->first_row_index : 0ull, dCF, stream());
      download(CFh, dCF, (size_t) n);
      auto &mk = device_markers();
      auto it = mk.find(CFh);
      if (it != mk.end()) { hypre_Free(it->second, HYPRE_MEMORY_DEVICE); }
      mk[CFh] = dCF;
      g_device_coarsen_count++;
      return hypre_error_flag;
   }
   const HYPRE_Int *Sdi = Sd->i, *Sdj = Sd->j, *Soi = So->i, *Soj = So->j;
   hypre_ParCSRCommPkg *pkg = nullptr;
   if (nprocs > 1)
   {
      if (!A->comm_pkg) { hypre_MatvecCommPkgCreate(A); }
      pkg = A->comm_pkg;
   }
   const HYPRE_Int tot_send = pkg ? pkg->send_map_starts[pkg->num_sends] : 0;
   std::vector<HYPRE_Int> int_buf((size_t) std::max(tot_send, 1));
   std::vector<double> buf((size_t) std::max(tot_send, 1));

   // measure = number of points a point influences (column sums of S) + random in (0,1]
   std::vector<double> measure((size_t) n + nco, 0.0);
   for (HYPRE_Int k = 0; k < Soi[n]; k++) { measure[(size_t) n + Soj[k]] += 1.0; }
   if (pkg) { halo_reverse<double>(pkg, measure.data() + n, buf.data()); }
   // (column counts of S: whole numbers, so the order in which the ones are added does not matter — rows in parallel;
   // every loop of the independent-set iteration below is order-independent as well: a point's fate depends on the
   // measures, which the sweeps do not change, and on which neighbours are in the set, which only grows between sweeps.
   // The reference runs them sequentially, par_coarsen.c:2100-2600; 16.8 M rows took a second that way.)
   if (n > 100000)
   {
#pragma omp parallel for schedule(static)
      for (HYPRE_Int i = 0; i < n; i++)
      {
         for (HYPRE_Int k = Sdi[i]; k < Sdi[i + 1]; k++)
         {
            double &m = measure[(size_t) Sdj[k]];
#pragma omp atomic
            m += 1.0;
         }
      }
   }
   else { for (HYPRE_Int k = 0; k < Sdi[n]; k++) { measure[(size_t) Sdj[k]] += 1.0; } }
   for (HYPRE_Int k = 0; k < tot_send; k++) { measure[(size_t) pkg->send_map_elmts[k]] += buf[(size_t) k]; }
   for (HYPRE_Int k = n; k < n + nco; k++) { measure[(size_t) k] = 0; }
   {
      // par_indepset.c: seed 2747+rank, or one global stream when seq_rand
      HypreRand rng;
      const bool seq_rand = (CF_init == 2 || CF_init == 4);
      rng.reseed(seq_rand ? 2747 : 2747 + my_id);
      if (seq_rand) { for (HYPRE_BigInt q = 0; q < S->first_row_index; q++) { rng.next(); } }
      for (HYPRE_Int i = 0; i < n; i++) { measure[(size_t) i] += rng.next(); }
   }

   std::vector<HYPRE_Int> graph((size_t) std::max(n, 1)), graph2((size_t) std::max(n, 1));
   std::vector<HYPRE_Int> graph_offd((size_t) std::max(nco, 1)), graph_offd2((size_t) std::max(nco, 1));
   for (HYPRE_Int k = 0; k < nco; k++) { graph_offd[(size_t) k] = k; }
   HYPRE_Int graph_offd_size = nco;

   if (*CF_marker_ptr == nullptr)
   {
      *CF_marker_ptr = hypre_IntArrayCreate(n);
      hypre_IntArrayInitialize_v2(*CF_marker_ptr, HYPRE_MEMORY_HOST);
   }
   HYPRE_Int *CF = (*CF_marker_ptr)->data;
   HYPRE_Int cnt = 0;
   if (CF_init == 1)
   {
      for (HYPRE_Int i = 0; i < n; i++)
      {
         if (CF[i] != SF_PT)
         {
            if (Soi[i + 1] - Soi[i] > 0 || CF[i] == -1) { CF[i] = 0; }
            if (CF[i] == Z_PT)
            {
               if (measure[(size_t) i] >= 1.0 || Sdi[i + 1] - Sdi[i] > 0) { CF[i] = 0; graph[(size_t) cnt++] = i; }
               else { CF[i] = F_PT; }
            }
            else { graph[(size_t) cnt++] = i; }
         }
         else { measure[(size_t) i] = 0; }
      }
   }
   else
   {
      for (HYPRE_Int i = 0; i < n; i++)
      {
         CF[i] = 0;
         const HYPRE_Int nnzrow = (Sdi[i + 1] - Sdi[i]) + (Soi[i + 1] - Soi[i]);
         if (nnzrow == 0)
         {
            CF[i] = SF_PT;
            if (CF_init == 3 || CF_init == 4) { CF[i] = C_PT; }
            measure[(size_t) i] = 0;
         }
         else { graph[(size_t) cnt++] = i; }
      }
   }
   HYPRE_Int graph_size = cnt;
   std::vector<HYPRE_Int> CF_offd((size_t) std::max(nco, 1), 0);
   if (pkg) { halo_forward<double>(pkg, measure.data(), measure.data() + n); }

   HYPRE_Int iter = 0;
   while (true)
   {
      if (global_sum_big(comm, graph_size) == 0) { break; }
      if (!CF_init || iter)
      {
#pragma omp parallel for schedule(static) if (graph_size > 100000)
         for (HYPRE_Int ig = 0; ig < graph_size; ig++)
         {
            const HYPRE_Int i = graph[(size_t) ig];
            if (measure[(size_t) i] > 1) { CF[i] = 1; }
         }
         for (HYPRE_Int ig = 0; ig < graph_offd_size; ig++)
         {
            const HYPRE_Int i = graph_offd[(size_t) ig];
            if (measure[(size_t) i + n] > 1) { CF_offd[(size_t) i] = 1; }
         }
         // knock the smaller of two strongly connected candidates out of the set (rows in parallel: the only writes are
         // zeros, decided by measures alone)
#pragma omp parallel for schedule(static) if (graph_size > 100000)
         for (HYPRE_Int ig = 0; ig < graph_size; ig++)
         {
            const HYPRE_Int i = graph[(size_t) ig];
            if (measure[(size_t) i] > 1)
            {
               for (HYPRE_Int jS = Sdi[i]; jS < Sdi[i + 1]; jS++)
               {
                  const HYPRE_Int j = Sdj[jS];
                  if (measure[(size_t) j] > 1)
                  {
                     if (measure[(size_t) i] > measure[(size_t) j]) { __atomic_store_n(&CF[j], 0, __ATOMIC_RELAXED); }
                     else if (measure[(size_t) j] > measure[(size_t) i]) { __atomic_store_n(&CF[i], 0, __ATOMIC_RELAXED); }
                  }
               }
               for (HYPRE_Int jS = Soi[i]; jS < Soi[i + 1]; jS++)
               {
                  const HYPRE_Int jj = Soj[jS];
                  const HYPRE_Int j = n + jj;
                  if (measure[(size_t) j] > 1)
                  {
                     if (measure[(size_t) i] > measure[(size_t) j]) { __atomic_store_n(&CF_offd[(size_t) jj], 0, __ATOMIC_RELAXED); }
                     else if (measure[(size_t) j] > measure[(size_t) i]) { __atomic_store_n(&CF[i], 0, __ATOMIC_RELAXED); }
                  }
               }
            }
         }
         if (pkg)
         {
            halo_reverse<HYPRE_Int>(pkg, CF_offd.data(), int_buf.data());
            for (HYPRE_Int k = 0; k < tot_send; k++)
            {
               const HYPRE_Int elmt = pkg->send_map_elmts[k];
               if (!int_buf[(size_t) k] && CF[elmt] > 0) { CF[elmt] = 0; }
               else { int_buf[(size_t) k] = CF[elmt]; }
            }
            hypre_ParCSRCommHandle *h = hypre_ParCSRCommHandleCreate(11, pkg, int_buf.data(), CF_offd.data());
            hypre_ParCSRCommHandleDestroy(h);
         }
      }
      iter++;
      // (rows in parallel: "CF[j] > 0" — j is in the set — cannot change during this sweep: a point in the set keeps a
      // positive marker, a point outside it can only be written -1)
#pragma omp parallel for schedule(static) if (graph_size > 100000)
      for (HYPRE_Int ig = 0; ig < graph_size; ig++)
      {
         const HYPRE_Int i = graph[(size_t) ig];
         HYPRE_Int mine = __atomic_load_n(&CF[i], __ATOMIC_RELAXED);
         if (measure[(size_t) i] < 1) { mine = F_PT; }
         if (mine > 0) { mine = C_PT; }
         else
         {
            for (HYPRE_Int jS = Sdi[i]; jS < Sdi[i + 1]; jS++) { if (__atomic_load_n(&CF[Sdj[jS]], __ATOMIC_RELAXED) > 0) { mine = F_PT; } }
            for (HYPRE_Int jS = Soi[i]; jS < Soi[i + 1]; jS++) { if (CF_offd[(size_t) Soj[jS]] > 0) { mine = F_PT; } }
         }
         __atomic_store_n(&CF[i], mine, __ATOMIC_RELAXED);
      }
      if (pkg) { halo_forward<HYPRE_Int>(pkg, CF, CF_offd.data()); }
      HYPRE_Int g2 = 0, go2 = 0;
      for (HYPRE_Int ig = 0; ig < graph_size; ig++)
      {
         const HYPRE_Int i = graph[(size_t) ig];
         if (CF[i] != 0) { measure[(size_t) i] = 0; } else { graph2[(size_t) g2++] = i; }
      }
      for (HYPRE_Int ig = 0; ig < graph_offd_size; ig++)
      {
         const HYPRE_Int i = graph_offd[(size_t) ig];
         if (CF_offd[(size_t) i] != 0) { measure[(size_t) i + n] = 0; } else { graph_offd2[(size_t) go2++] = i; }
      }
      graph.swap(graph2); graph_offd.swap(graph_offd2);
      graph_size = g2; graph_offd_size = go2;
   }
   return hypre_error_flag;
}

// ===========================================================================
// Ruge-Stueben first pass on the local graph (par_coarsen.c:911-1400 with
// coarsen_type 10 -> 11, measure_type 0), then PMIS on what it leaves.
// The bucket structure reproduces the reference's list-of-lists: one FIFO
// per measure value; the head of the highest non-empty list is picked.
// ===========================================================================
namespace {
struct Buckets
{
   std::vector<int> head, tail, next, prev, where_meas;
   int max_meas = 0;
   explicit Buckets(int n, int maxm) : head((size_t) maxm + 2, -1), tail((size_t) maxm + 2, -1),
      next((size_t) std::max(n, 1), -1), prev((size_t) std::max(n, 1), -1), where_meas((size_t) std::max(n, 1), -1) {}
   void grow(int m) { if ((size_t) m + 2 > head.size()) { head.resize((size_t) m + 2, -1); tail.resize((size_t) m + 2, -1); } }
   void enter(int m, int i)
   {
      grow(m);
      next[(size_t) i] = -1; prev[(size_t) i] = tail[(size_t) m];
      if (tail[(size_t) m] >= 0) { next[(size_t) tail[(size_t) m]] = i; } else { head[(size_t) m] = i; }
      tail[(size_t) m] = i;
      where_meas[(size_t) i] = m;
      if (m > max_meas) { max_meas = m; }
   }
   void remove(int m, int i)
   {
      const int p = prev[(size_t) i], q = next[(size_t) i];
      if (p >= 0) { next[(size_t) p] = q; } else { head[(size_t) m] = q; }
      if (q >= 0) { prev[(size_t) q] = p; } else { tail[(size_t) m] = p; }
      where_meas[(size_t) i] = -1;
   }
   int top()
   {
      while (max_meas > 0 && head[(size_t) max_meas] < 0) { max_meas--; }
      return head[(size_t) max_meas];
   }
};
}  // namespace

static void ruge_first_pass(hypre_ParCSRMatrix *S, HYPRE_Int *CF)
{
   hypre_CSRMatrix *Sd = S->diag, *So = S->offd;
   const HYPRE_Int n = Sd->num_rows;
   const HYPRE_Int *Si = Sd->i, *Sj = Sd->j, *Soi = So->i;
   const HYPRE_Int nS = Si[n];
   // S^T by counting sort (rows in ascending source order)
   std::vector<HYPRE_Int> STi((size_t) n + 1, 0), STj((size_t) std::max(nS, 1));
   for (HYPRE_Int k = 0; k < nS; k++) { STi[(size_t) Sj[k] + 1]++; }
   for (HYPRE_Int i = 0; i < n; i++) { STi[(size_t) i + 1] += STi[(size_t) i]; }
   {
      std::vector<HYPRE_Int> pos(STi.begin(), STi.end() - 1);
      for (HYPRE_Int i = 0; i < n; i++) { for (HYPRE_Int k = Si[i]; k < Si[i + 1]; k++) { STj[(size_t) pos[(size_t) Sj[k]]++] = i; } }
   }
   std::vector<HYPRE_Int> meas((size_t) std::max(n, 1));
   HYPRE_Int maxm = 0;
   for (HYPRE_Int i = 0; i < n; i++) { meas[(size_t) i] = STi[(size_t) i + 1] - STi[(size_t) i]; maxm = std::max(maxm, meas[(size_t) i]); }
   const int f_pnt = Z_PT;
   HYPRE_Int num_left = 0;
   for (HYPRE_Int j = 0; j < n; j++)
   {
      if (CF[j] == 0)
      {
         const HYPRE_Int nnzrow = (Si[j + 1] - Si[j]) + (Soi[j + 1] - Soi[j]);
         if (nnzrow == 0) { CF[j] = SF_PT; meas[(size_t) j] = 0; }
         else { CF[j] = UNDECIDED; num_left++; }
      }
      else { meas[(size_t) j] = 0; }
   }
   Buckets B(n, 2 * maxm + 2);
   for (HYPRE_Int j = 0; j < n; j++)
   {
      const HYPRE_Int measure = meas[(size_t) j];
      if (CF[j] != SF_PT && CF[j] != SC_PT)
      {
         if (measure > 0) { B.enter(measure, j); }
         else
         {
            CF[j] = f_pnt;
            for (HYPRE_Int k = Si[j]; k < Si[j + 1]; k++)
            {
               const HYPRE_Int nabor = Sj[k];
               if (CF[nabor] != SF_PT && CF[nabor] != SC_PT)
               {
                  if (nabor < j)
                  {
                     HYPRE_Int nm = meas[(size_t) nabor];
                     if (nm > 0) { B.remove(nm, nabor); }
                     nm = ++meas[(size_t) nabor];
                     B.enter(nm, nabor);
                  }
                  else { ++meas[(size_t) nabor]; }
               }
            }
            --num_left;
         }
      }
   }
   while (num_left > 0)
   {
      const HYPRE_Int index = B.top();
      CF[index] = C_PT;
      HYPRE_Int measure = meas[(size_t) index];
      meas[(size_t) index] = 0;
      --num_left;
      B.remove(measure, index);
      for (HYPRE_Int j = STi[(size_t) index]; j < STi[(size_t) index + 1]; j++)
      {
         const HYPRE_Int nabor = STj[(size_t) j];
         if (CF[nabor] == UNDECIDED)
         {
            CF[nabor] = F_PT;
            B.remove(meas[(size_t) nabor], nabor);
            --num_left;
            for (HYPRE_Int k = Si[nabor]; k < Si[nabor + 1]; k++)
            {
               const HYPRE_Int n2 = Sj[k];
               if (CF[n2] == UNDECIDED)
               {
                  B.remove(meas[(size_t) n2], n2);
                  const HYPRE_Int nm = ++meas[(size_t) n2];
                  B.enter(nm, n2);
               }
            }
         }
      }
      for (HYPRE_Int j = Si[index]; j < Si[index + 1]; j++)
      {
         const HYPRE_Int nabor = Sj[j];
         if (CF[nabor] == UNDECIDED)
         {
            measure = meas[(size_t) nabor];
            B.remove(measure, nabor);
            meas[(size_t) nabor] = --measure;
            if (measure > 0) { B.enter(measure, nabor); }
            else
            {
               CF[nabor] = F_PT;
               --num_left;
               for (HYPRE_Int k = Si[nabor]; k < Si[nabor + 1]; k++)
               {
                  const HYPRE_Int n2 = Sj[k];
                  if (CF[n2] == UNDECIDED)
                  {
                     B.remove(meas[(size_t) n2], n2);
                     const HYPRE_Int nm = ++meas[(size_t) n2];
                     B.enter(nm, n2);
                  }
               }
            }
         }
      }
   }
   for (HYPRE_Int i = 0; i < n; i++) { if (CF[i] == SC_PT) { CF[i] = C_PT; } }
}

HYPRE_Int hypre_BoomerAMGCoarsenHMIS(hypre_ParCSRMatrix *S, hypre_ParCSRMatrix *A, HYPRE_Int measure_type,
                                     HYPRE_Int cut_factor, HYPRE_Int debug_flag, hypre_IntArray **CF_marker_ptr)
{
   (void) measure_type; (void) cut_factor;
   const HYPRE_Int n = S->diag->num_rows;
   if (*CF_marker_ptr == nullptr)
   {
      *CF_marker_ptr = hypre_IntArrayCreate(n);
      hypre_IntArrayInitialize_v2(*CF_marker_ptr, HYPRE_MEMORY_HOST);
   }
   ruge_first_pass(S, (*CF_marker_ptr)->data);
   return hypre_BoomerAMGCoarsenPMIS(S, A, 1, debug_flag, CF_marker_ptr);
}

// par_coarse_parms.c: this rank's coarse range [first, first + local)
static void coarse_parms(MPI_Comm comm, HYPRE_Int n, const HYPRE_Int *CF, HYPRE_BigInt *cpts_global,
                         HYPRE_BigInt *total)
{
   HYPRE_BigInt local = 0;
   for (HYPRE_Int i = 0; i < n; i++) { if (CF[i] == 1) { local++; } }
   const hypre_amd_CommOps *o = comm_ops(comm);
   if (o && o->size > 1)
   {
      std::vector<HYPRE_BigInt> all((size_t) o->size);
      o->allgather(o->ctx, &local, all.data(), sizeof(HYPRE_BigInt));
      HYPRE_BigInt first = 0, tot = 0;
      for (int r = 0; r < o->size; r++) { if (r < o->rank) { first += all[(size_t) r]; } tot += all[(size_t) r]; }
      cpts_global[0] = first; cpts_global[1] = first + local; *total = tot;
   }
   else { cpts_global[0] = 0; cpts_global[1] = local; *total = local; }
}

// ===========================================================================
// truncation of P (par_csr_matrix.c:2874-3400 with rescale = 1, inf-norm)
// ===========================================================================
static void qsort2_abs(HYPRE_Int *v, HYPRE_Real *w, HYPRE_Int left, HYPRE_Int right)
{
   // utilities/qsort.c:395-417 (decreasing |w|; tie order is part of the contract)
   if (left >= right) { return; }
   std::swap(v[left], v[(left + right) / 2]); std::swap(w[left], w[(left + right) / 2]);
   HYPRE_Int last = left;
   for (HYPRE_Int i = left + 1; i <= right; i++)
   {
      if (std::fabs(w[i]) > std::fabs(w[left])) { ++last; std::swap(v[last], v[i]); std::swap(w[last], w[i]); }
   }
   std::swap(v[left], v[last]); std::swap(w[left], w[last]);
   qsort2_abs(v, w, left, last - 1);
   qsort2_abs(v, w, last + 1, right);
}

HYPRE_Int hypre_BoomerAMGInterpTruncation(hypre_ParCSRMatrix *P, HYPRE_Real tol, HYPRE_Int max_elmts)
{
   if (tol <= 0.0 && max_elmts == 0) { return hypre_error_flag; }
   hypre_CSRMatrix *Pd = P->diag, *Po = P->offd;
   const HYPRE_Int n = Pd->num_rows, ncols = Pd->num_cols;
   std::vector<HYPRE_Int> ndi((size_t) n + 1, 0), noi((size_t) n + 1, 0);
   // rows are independent: rewrite every row in place (rows only shrink), then compact
   std::vector<HYPRE_Int> cntd((size_t) std::max(n, 1)), cnto((size_t) std::max(n, 1));
#pragma omp parallel
   {
      std::vector<HYPRE_Int> aj;
      std::vector<HYPRE_Real> aa;
#pragma omp for schedule(static)
      for (HYPRE_Int i = 0; i < n; i++)
      {
         HYPRE_Int d0 = Pd->i[i], d1 = Pd->i[i + 1], o0 = Po->i[i], o1 = Po->i[i + 1];
         if (tol > 0)
         {
            HYPRE_Real row_nrm = 0;
            for (HYPRE_Int k = d0; k < d1; k++) { row_nrm = std::max(row_nrm, std::fabs(Pd->data[k])); }
            for (HYPRE_Int k = o0; k < o1; k++) { row_nrm = std::max(row_nrm, std::fabs(Po->data[k])); }
            const HYPRE_Real drop = tol * row_nrm;
            HYPRE_Real row_sum = 0, scale = 0;
            HYPRE_Int w = d0;
            for (HYPRE_Int k = d0; k < d1; k++)
            {
               row_sum += Pd->data[k];
               if (!(std::fabs(Pd->data[k]) < drop)) { scale += Pd->data[k]; Pd->data[w] = Pd->data[k]; Pd->j[w] = Pd->j[k]; w++; }
            }
            d1 = w; w = o0;
            for (HYPRE_Int k = o0; k < o1; k++)
            {
               row_sum += Po->data[k];
               if (!(std::fabs(Po->data[k]) < drop)) { scale += Po->data[k]; Po->data[w] = Po->data[k]; Po->j[w] = Po->j[k]; w++; }
            }
            o1 = w;
            if (scale != 0. && scale != row_sum)
            {
               scale = row_sum / scale;
               for (HYPRE_Int k = d0; k < d1; k++) { Pd->data[k] *= scale; }
               for (HYPRE_Int k = o0; k < o1; k++) { Po->data[k] *= scale; }
            }
         }
         const HYPRE_Int num = (d1 - d0) + (o1 - o0);
         if (max_elmts > 0 && max_elmts < num)
         {
            aj.resize((size_t) num); aa.resize((size_t) num);
            HYPRE_Int c = 0;
            HYPRE_Real row_sum = 0;
            for (HYPRE_Int k = d0; k < d1; k++) { aj[(size_t) c] = Pd->j[k]; aa[(size_t) c++] = Pd->data[k]; row_sum += Pd->data[k]; }
            for (HYPRE_Int k = o0; k < o1; k++) { aj[(size_t) c] = Po->j[k] + ncols; aa[(size_t) c++] = Po->data[k]; row_sum += Po->data[k]; }
            qsort2_abs(aj.data(), aa.data(), 0, c - 1);
            HYPRE_Real scale = 0;
            HYPRE_Int wd = d0, wo = o0;
            for (HYPRE_Int k = 0; k < max_elmts; k++)
            {
               scale += aa[(size_t) k];
               if (aj[(size_t) k] < ncols) { Pd->j[wd] = aj[(size_t) k]; Pd->data[wd++] = aa[(size_t) k]; }
               else { Po->j[wo] = aj[(size_t) k] - ncols; Po->data[wo++] = aa[(size_t) k]; }
            }
            d1 = wd; o1 = wo;
            if (scale != 0. && scale != row_sum)
            {
               scale = row_sum / scale;
               for (HYPRE_Int k = d0; k < d1; k++) { Pd->data[k] *= scale; }
               for (HYPRE_Int k = o0; k < o1; k++) { Po->data[k] *= scale; }
            }
         }
         cntd[(size_t) i] = d1 - d0; cnto[(size_t) i] = o1 - o0;
      }
   }
   for (HYPRE_Int i = 0; i < n; i++) { ndi[(size_t) i + 1] = ndi[(size_t) i] + cntd[(size_t) i]; noi[(size_t) i + 1] = noi[(size_t) i] + cnto[(size_t) i]; }
   // compact: rows move to their new offsets (out of place, rows in parallel)
   auto compact = [&](hypre_CSRMatrix *M, const std::vector<HYPRE_Int> &ni, const std::vector<HYPRE_Int> &cnt)
   {
      const HYPRE_Int nnz_new = ni[(size_t) n];
      if (nnz_new == M->i[n]) { return; }
      HYPRE_Int  *nj = hypre_TAlloc(HYPRE_Int, (size_t) std::max(nnz_new, 1), HYPRE_MEMORY_HOST);
      HYPRE_Real *na = hypre_TAlloc(HYPRE_Real, (size_t) std::max(nnz_new, 1), HYPRE_MEMORY_HOST);
#pragma omp parallel for schedule(static)
      for (HYPRE_Int i = 0; i < n; i++)
      {
         const HYPRE_Int s0 = M->i[i], t0 = ni[(size_t) i];
         for (HYPRE_Int k = 0; k < cnt[(size_t) i]; k++) { nj[t0 + k] = M->j[s0 + k]; na[t0 + k] = M->data[s0 + k]; }
      }
      hypre_Free(M->j, HYPRE_MEMORY_HOST); hypre_Free(M->data, HYPRE_MEMORY_HOST);
      M->j = nj; M->data = na;
   };
   compact(Pd, ndi, cntd);
   compact(Po, noi, cnto);
   memcpy(Pd->i, ndi.data(), sizeof(HYPRE_Int) * ((size_t) n + 1));
   memcpy(Po->i, noi.data(), sizeof(HYPRE_Int) * ((size_t) n + 1));
   Pd->num_nonzeros = ndi[(size_t) n];
   Po->num_nonzeros = noi[(size_t) n];
   return hypre_error_flag;
}

// ===========================================================================
// extended+i interpolation (par_lr_interp.c:1024-1700), single-rank form.
// For an F-point i the interpolatory set C^_i is: strong C neighbours of i and
// strong C neighbours of i's strong F neighbours (distance two).  Weights:
//   w_ij = -( a_ij + sum_{k in F_i^s} a_ik a^-_kj / sum_{l in C^_i u {i}} a^-_kl ) / a~_ii
// with a~_ii = a_ii + weak couplings + the "+i" share of every distributed F neighbour.
// ===========================================================================
HYPRE_Int hypre_BoomerAMGBuildExtPIInterp(hypre_ParCSRMatrix *A, HYPRE_Int *CF_marker, hypre_ParCSRMatrix *S,
                                          HYPRE_BigInt *num_cpts_global, HYPRE_Int num_functions,
                                          HYPRE_Int *dof_func, HYPRE_Int debug_flag, HYPRE_Real trunc_factor,
                                          HYPRE_Int max_elmts, hypre_ParCSRMatrix **P_ptr)
{
   (void) debug_flag;
   if (num_functions > 1 && !dof_func)
   {
      hypre_error_w_msg(HYPRE_ERROR_GENERIC, "hypre_BoomerAMGBuildExtPIInterp: num_functions > 1 needs the function of every row");
      return hypre_error_flag;
   }
   const bool sys = num_functions > 1;
   MPI_Comm comm = A->comm;
   if (comm_size(comm) > 1)
   {
      // total number of coarse points = upper bound of the last rank's range
      const hypre_amd_CommOps *o = comm_ops(comm);
      std::vector<HYPRE_BigInt> ends((size_t) o->size);
      o->allgather(o->ctx, &num_cpts_global[1], ends.data(), sizeof(HYPRE_BigInt));
      if (g_level_on_device && !sys && S->diag->memory_location == HYPRE_MEMORY_DEVICE)
      {
         // P_ptr comes back empty (error flag clear) when some rank's rows did not fit the kernel's tables: the caller
         // repeats the level's interpolation with the host routine
         dist_device_extpi_interp(A, CF_marker, S, num_cpts_global, ends.back(), trunc_factor, max_elmts, g_device_interp_on - 1, P_ptr);
         if (*P_ptr) { g_device_interp_count++; }
         return hypre_error_flag;
      }
      return dist_build_extpi_interp(A, CF_marker, S, num_cpts_global, ends.back(), sys ? dof_func : nullptr, trunc_factor, max_elmts, P_ptr);
   }
   hypre_CSRMatrix *Ad = A->diag;
   const HYPRE_Int n = Ad->num_rows;
   const HYPRE_BigInt total_cpts = num_cpts_global[1];

   // On the device when the hierarchy's home is device memory (interp_kernels.hip: one wave per row, this loop's order,
   // same bits; hypre_amd_SetSetupDeviceInterp(0) keeps the host loop)
   if (!sys && g_device_interp_on && !g_interp_host_once && device_setup_allowed() && n >= g_device_rap_min_rows &&
       Ad->num_nonzeros > 0 && ensure_device())
   {
      hipStream_t st = stream();
      hypre_CSRMatrix *dA = device_twin_of(Ad, 1);
      const bool S_there = S->diag->memory_location == HYPRE_MEMORY_DEVICE;
      hypre_CSRMatrix *dS = S_there ? S->diag : hypre_CSRMatrixClone_v2(S->diag, 0, HYPRE_MEMORY_DEVICE);
      HYPRE_Int *dCF = device_marker_of(CF_marker, n);           // left by the device coarsening, or uploaded now
      HYPRE_Int *dF2C = hypre_TAlloc(HYPRE_Int, (size_t) n, HYPRE_MEMORY_DEVICE);
      device_coarse_numbering(n, dCF, dF2C, st);
      int *dPi = nullptr, *dPj = nullptr, pnnz = 0;
      double *dPa = nullptr;
      const bool ok = device_extpi(n, dA->i, dA->j, dA->data, dS->i, dS->j, dCF, dF2C, trunc_factor, max_elmts, g_device_interp_on - 1, &dPi, &dPj, &dPa, &pnnz, st);
      if (!S_there) { hypre_CSRMatrixDestroy(dS); }
      hypre_Free(dF2C, HYPRE_MEMORY_DEVICE);
      if (ok)
      {
         // the operator stays where it was made: nothing on the host reads it (the Galerkin product that follows runs
         // on the device as well; should that fall back to the host loop, it fetches a copy)
         HYPRE_BigInt cs[2] = {num_cpts_global[0], num_cpts_global[1]};
         hypre_ParCSRMatrix *P = hypre_ParCSRMatrixCreate(comm, A->global_num_rows, total_cpts, A->col_starts, cs, 0, pnnz, 0);
         hypre_CSRMatrixDestroy(P->diag);
         P->diag = wrap_device_csr(n, (HYPRE_Int) (cs[1] - cs[0]), pnnz, dPi, dPj, dPa);
         hypre_CSRMatrixInitialize_v2(P->offd, 0, HYPRE_MEMORY_HOST);
         hypre_CSRMatrixSetRownnz(P->offd);
         *P_ptr = P;
         g_device_interp_count++;
         return hypre_error_flag;
      }
   }
   g_interp_host_once = false;
   if (Ad->memory_location != HYPRE_MEMORY_HOST || S->diag->memory_location != HYPRE_MEMORY_HOST)
   {
      // the host loop below needs host copies: the caller (the setup loop) fetches them and asks again
      *P_ptr = nullptr;
      return hypre_error_flag;
   }
   const HYPRE_Int *Ai = Ad->i, *Aj = Ad->j;
   const HYPRE_Real *Aa = Ad->data;
   const HYPRE_Int *Si = S->diag->i, *Sj = S->diag->j;
   std::vector<HYPRE_Int> f2c((size_t) std::max(n, 1), -1);
   {
      HYPRE_Int c = 0;
      for (HYPRE_Int i = 0; i < n; i++) { if (CF_marker[i] >= 0) { f2c[(size_t) i] = c++; } }
   }
   const int T = num_threads_avail();
   std::vector<std::vector<HYPRE_Int>> tj((size_t) T);
   std::vector<std::vector<HYPRE_Real>> ta((size_t) T);
   std::vector<HYPRE_Int> rowlen((size_t) std::max(n, 1), 0);

#pragma omp parallel num_threads(T)
   {
      const int t = omp_get_thread_num();
      int rb, re;
      chunk(n, T, t, &rb, &re);
      std::vector<HYPRE_Int> &pj = tj[(size_t) t];
      std::vector<HYPRE_Real> &pa = ta[(size_t) t];
      // marker holds, per fine point, either the position of its column in
      // the current row (>= row begin), or the strong-F tag of the current row
      RowMarker marker;
      if (re > rb)
      {
         // index window of everything rows [rb, re) can touch: two hops through A's pattern
         HYPRE_Int lo1 = rb, hi1 = re - 1;
         for (HYPRE_Int k = Ai[rb]; k < Ai[re]; k++) { lo1 = std::min(lo1, Aj[k]); hi1 = std::max(hi1, Aj[k]); }
         HYPRE_Int lo2 = lo1, hi2 = hi1;
         for (HYPRE_Int k = Ai[lo1]; k < Ai[hi1 + 1]; k++) { lo2 = std::min(lo2, Aj[k]); hi2 = std::max(hi2, Aj[k]); }
         if ((long long) hi2 - lo2 + 1 <= 16LL * (re - rb) + 65536) { marker.use_window(lo2, hi2); }
      }
      long long strong_f = -2;
      for (HYPRE_Int i = rb; i < re; i++)
      {
         const long long begin = (long long) pj.size();
         if (CF_marker[i] >= 0)
         {
            pj.push_back(f2c[(size_t) i]); pa.push_back(1.0);
         }
         else if (CF_marker[i] != -3)
         {
            strong_f--;
            for (HYPRE_Int jj = Si[i]; jj < Si[i + 1]; jj++)
            {
               const HYPRE_Int i1 = Sj[jj];
               if (CF_marker[i1] >= 0)
               {
                  if (marker.get(i1) < begin) { marker.set(i1, (long long) pj.size()); pj.push_back(f2c[(size_t) i1]); pa.push_back(0.0); }
               }
               else if (CF_marker[i1] != -3)
               {
                  marker.set(i1, strong_f);
                  for (HYPRE_Int kk = Si[i1]; kk < Si[i1 + 1]; kk++)
                  {
                     const HYPRE_Int k1 = Sj[kk];
                     if (CF_marker[k1] >= 0 && marker.get(k1) < begin)
                     {
                        marker.set(k1, (long long) pj.size()); pj.push_back(f2c[(size_t) k1]); pa.push_back(0.0);
                     }
                  }
               }
            }
            HYPRE_Real diagonal = Aa[Ai[i]];
            for (HYPRE_Int jj = Ai[i] + 1; jj < Ai[i + 1]; jj++)
            {
               const HYPRE_Int i1 = Aj[jj];
               if (marker.get(i1) >= begin) { pa[(size_t) marker.get(i1)] += Aa[jj]; }
               else if (marker.get(i1) == strong_f)
               {
                  HYPRE_Real sum = 0.0;
                  const int sgn = Aa[Ai[i1]] < 0 ? -1 : 1;
                  for (HYPRE_Int j1 = Ai[i1] + 1; j1 < Ai[i1 + 1]; j1++)
                  {
                     const HYPRE_Int i2 = Aj[j1];
                     if ((marker.get(i2) >= begin || i2 == i) && (sgn * Aa[j1]) < 0) { sum += Aa[j1]; }
                  }
                  if (sum != 0)
                  {
                     const HYPRE_Real distribute = Aa[jj] / sum;
                     for (HYPRE_Int j1 = Ai[i1] + 1; j1 < Ai[i1 + 1]; j1++)
                     {
                        const HYPRE_Int i2 = Aj[j1];
                        if (marker.get(i2) >= begin && (sgn * Aa[j1]) < 0) { pa[(size_t) marker.get(i2)] += distribute * Aa[j1]; }
                        if (i2 == i && (sgn * Aa[j1]) < 0) { diagonal += distribute * Aa[j1]; }
                     }
                  }
                  else { diagonal += Aa[jj]; }
               }
               // weak neighbour: lumped into the diagonal, within the same function only (par_lr_interp.c:1706-1713)
               else if (CF_marker[i1] != -3) { if (!sys || dof_func[i] == dof_func[i1]) { diagonal += Aa[jj]; } }
            }
            if (diagonal) { for (size_t k = (size_t) begin; k < pj.size(); k++) { pa[k] /= -diagonal; } }
            strong_f--;
         }
         rowlen[(size_t) i] = (HYPRE_Int) ((long long) pj.size() - begin);
         marker.end_row();
      }
   }
   // stitch
   std::vector<HYPRE_Int> Pi((size_t) n + 1, 0);
   for (HYPRE_Int i = 0; i < n; i++) { Pi[(size_t) i + 1] = Pi[(size_t) i] + rowlen[(size_t) i]; }
   const HYPRE_Int nnz = Pi[(size_t) n];
   HYPRE_BigInt cs[2] = {num_cpts_global[0], num_cpts_global[1]};
   hypre_ParCSRMatrix *P = hypre_ParCSRMatrixCreate(comm, A->global_num_rows, total_cpts, A->col_starts, cs, 0, nnz, 0);
   hypre_ParCSRMatrixInitialize_v2(P, HYPRE_MEMORY_HOST);
   memcpy(P->diag->i, Pi.data(), sizeof(HYPRE_Int) * ((size_t) n + 1));
#pragma omp parallel num_threads(T)
   {
      const int t = omp_get_thread_num();
      int rb, re;
      chunk(n, T, t, &rb, &re);
      if (re > rb)
      {
         const size_t off = (size_t) Pi[(size_t) rb];
         if (!tj[(size_t) t].empty())
         {
            memcpy(P->diag->j + off, tj[(size_t) t].data(), sizeof(HYPRE_Int) * tj[(size_t) t].size());
            memcpy(P->diag->data + off, ta[(size_t) t].data(), sizeof(HYPRE_Real) * ta[(size_t) t].size());
         }
      }
   }
   if (trunc_factor != 0.0 || max_elmts > 0) { hypre_BoomerAMGInterpTruncation(P, trunc_factor, max_elmts); }
   hypre_CSRMatrixSetRownnz(P->offd);
   *P_ptr = P;
   return hypre_error_flag;
}

// Direct interpolation (par_interp.c hypre_BoomerAMGBuildDirInterpHost, interp_type 3):
//   w_ij = -alpha a_ij / a_ii for strong C neighbours, alpha/beta rescale negative
//   and positive couplings so that the row sum of A is reproduced.
HYPRE_Int hypre_BoomerAMGBuildDirInterp(hypre_ParCSRMatrix *A, HYPRE_Int *CF_marker, hypre_ParCSRMatrix *S,
                                        HYPRE_BigInt *num_cpts_global, HYPRE_Int num_functions,
                                        HYPRE_Int *dof_func, HYPRE_Int debug_flag, HYPRE_Real trunc_factor,
                                        HYPRE_Int max_elmts, HYPRE_Int interp_type, hypre_ParCSRMatrix **P_ptr)
{
   (void) dof_func; (void) debug_flag; (void) interp_type;
   if (num_functions > 1)
   {
      hypre_error_w_msg(HYPRE_ERROR_GENERIC, "hypre_BoomerAMGBuildDirInterp: systems (num_functions > 1) are supported with interp_type 6 only");
      return hypre_error_flag;
   }
   MPI_Comm comm = A->comm;
   const bool dist = comm_size(comm) > 1;
   hypre_CSRMatrix *Ad = A->diag, *Ao = A->offd;
   const HYPRE_Int *Ai = Ad->i, *Aj = Ad->j, *Aoi = Ao->i, *Aoj = Ao->j;
   const HYPRE_Real *Aa = Ad->data, *Aoa = Ao->data;
   const HYPRE_Int *Si = S->diag->i, *Sj = S->diag->j, *Soi = S->offd->i, *Soj = S->offd->j;
   const HYPRE_Int n = Ad->num_rows, nco = dist ? Ao->num_cols : 0;
   std::vector<HYPRE_Int> f2c((size_t) std::max(n, 1), -1);
   HYPRE_Int c = 0;
   for (HYPRE_Int i = 0; i < n; i++) { if (CF_marker[i] >= 0) { f2c[(size_t) i] = c++; } }
   // ghost columns: C/F marker and global coarse index (par_interp.c:1990-2060)
   HYPRE_BigInt total_cpts = num_cpts_global[1];
   std::vector<HYPRE_Int> CF_offd((size_t) std::max(nco, 1), -1);
   std::vector<HYPRE_BigInt> f2c_offd((size_t) std::max(nco, 1), -1);
   if (dist)
   {
      const hypre_amd_CommOps *o = comm_ops(comm);
      std::vector<HYPRE_BigInt> ends((size_t) o->size);
      o->allgather(o->ctx, &num_cpts_global[1], ends.data(), sizeof(HYPRE_BigInt));
      total_cpts = ends.back();
      if (!A->comm_pkg) { hypre_MatvecCommPkgCreate(A); }
      if (nco)
      {
         std::vector<HYPRE_BigInt> f2c_big((size_t) std::max(n, 1));
         for (HYPRE_Int i = 0; i < n; i++) { f2c_big[(size_t) i] = (HYPRE_BigInt) f2c[(size_t) i] + num_cpts_global[0]; }
         halo_forward<HYPRE_Int>(A->comm_pkg, CF_marker, CF_offd.data());
         halo_forward<HYPRE_BigInt>(A->comm_pkg, f2c_big.data(), f2c_offd.data());
      }
      else
      {
         // collective all the same: ranks without ghosts still take part in the neighbours' exchanges
         halo_forward<HYPRE_Int>(A->comm_pkg, CF_marker, CF_offd.data());
         std::vector<HYPRE_BigInt> f2c_big((size_t) std::max(n, 1));
         for (HYPRE_Int i = 0; i < n; i++) { f2c_big[(size_t) i] = (HYPRE_BigInt) f2c[(size_t) i] + num_cpts_global[0]; }
         halo_forward<HYPRE_BigInt>(A->comm_pkg, f2c_big.data(), f2c_offd.data());
      }
   }
   std::vector<HYPRE_Int> Pi((size_t) n + 1, 0), Poi((size_t) n + 1, 0), pj, poj;
   std::vector<HYPRE_Real> pa, poa;
   std::vector<HYPRE_Int> marker((size_t) std::max(n, 1), -1), marker_o((size_t) std::max(nco, 1), -1);
   for (HYPRE_Int i = 0; i < n; i++)
   {
      const HYPRE_Int begin = (HYPRE_Int) pj.size(), begin_o = (HYPRE_Int) poj.size();
      if (CF_marker[i] >= 0) { pj.push_back(f2c[(size_t) i]); pa.push_back(1.0); }
      else
      {
         for (HYPRE_Int jj = Si[i]; jj < Si[i + 1]; jj++)
         {
            const HYPRE_Int i1 = Sj[jj];
            if (CF_marker[i1] >= 0) { marker[(size_t) i1] = (HYPRE_Int) pj.size(); pj.push_back(f2c[(size_t) i1]); pa.push_back(0.0); }
         }
         const HYPRE_Int end = (HYPRE_Int) pj.size();
         if (dist)
         {
            for (HYPRE_Int jj = Soi[i]; jj < Soi[i + 1]; jj++)
            {
               const HYPRE_Int i1 = Soj[jj];
               if (CF_offd[(size_t) i1] >= 0) { marker_o[(size_t) i1] = (HYPRE_Int) poj.size(); poj.push_back(i1); poa.push_back(0.0); }
            }
         }
         const HYPRE_Int end_o = (HYPRE_Int) poj.size();
         const HYPRE_Real diagonal = Aa[Ai[i]];
         HYPRE_Real sum_N_pos = 0, sum_N_neg = 0, sum_P_pos = 0, sum_P_neg = 0;
         for (HYPRE_Int jj = Ai[i] + 1; jj < Ai[i + 1]; jj++)
         {
            const HYPRE_Int i1 = Aj[jj];
            if (Aa[jj] > 0) { sum_N_pos += Aa[jj]; } else { sum_N_neg += Aa[jj]; }
            if (marker[(size_t) i1] >= begin)
            {
               pa[(size_t) marker[(size_t) i1]] += Aa[jj];
               if (Aa[jj] > 0) { sum_P_pos += Aa[jj]; } else { sum_P_neg += Aa[jj]; }
            }
         }
         if (dist)
         {
            for (HYPRE_Int jj = Aoi[i]; jj < Aoi[i + 1]; jj++)
            {
               const HYPRE_Int i1 = Aoj[jj];
               if (Aoa[jj] > 0) { sum_N_pos += Aoa[jj]; } else { sum_N_neg += Aoa[jj]; }
               if (marker_o[(size_t) i1] >= begin_o)
               {
                  poa[(size_t) marker_o[(size_t) i1]] += Aoa[jj];
                  if (Aoa[jj] > 0) { sum_P_pos += Aoa[jj]; } else { sum_P_neg += Aoa[jj]; }
               }
            }
         }
         HYPRE_Real alfa = 1.0, beta = 1.0, diag = diagonal;
         if (sum_P_neg) { alfa = sum_N_neg / sum_P_neg / diag; }
         if (sum_P_pos) { beta = sum_N_pos / sum_P_pos / diag; }
         for (HYPRE_Int k = begin; k < end; k++)
         {
            if (pa[(size_t) k] > 0) { pa[(size_t) k] *= -beta; } else { pa[(size_t) k] *= -alfa; }
         }
         for (HYPRE_Int k = begin_o; k < end_o; k++)
         {
            if (poa[(size_t) k] > 0) { poa[(size_t) k] *= -beta; } else { poa[(size_t) k] *= -alfa; }
         }
      }
      Pi[(size_t) i + 1] = (HYPRE_Int) pj.size();
      Poi[(size_t) i + 1] = (HYPRE_Int) poj.size();
   }
   HYPRE_BigInt cs[2] = {num_cpts_global[0], num_cpts_global[1]};
   hypre_ParCSRMatrix *P = hypre_ParCSRMatrixCreate(comm, A->global_num_rows, total_cpts, A->col_starts, cs, nco,
                                                    Pi[(size_t) n], Poi[(size_t) n]);
   hypre_ParCSRMatrixInitialize_v2(P, HYPRE_MEMORY_HOST);
   memcpy(P->diag->i, Pi.data(), sizeof(HYPRE_Int) * ((size_t) n + 1));
   memcpy(P->offd->i, Poi.data(), sizeof(HYPRE_Int) * ((size_t) n + 1));
   if (!pj.empty())
   {
      memcpy(P->diag->j, pj.data(), sizeof(HYPRE_Int) * pj.size());
      memcpy(P->diag->data, pa.data(), sizeof(HYPRE_Real) * pa.size());
   }
   if (!poj.empty())
   {
      memcpy(P->offd->j, poj.data(), sizeof(HYPRE_Int) * poj.size());
      memcpy(P->offd->data, poa.data(), sizeof(HYPRE_Real) * poa.size());
   }
   if (trunc_factor != 0.0 || max_elmts > 0) { hypre_BoomerAMGInterpTruncation(P, trunc_factor, max_elmts); }
   if (dist)
   {
      // ghost C-points that survived: P's own column map, ascending (par_interp.c:2370-2440)
      const HYPRE_Int nnz_o = P->offd->i[n];
      std::vector<HYPRE_Int> renum((size_t) std::max(nco, 1), -1);
      for (HYPRE_Int k = 0; k < nnz_o; k++) { renum[(size_t) P->offd->j[k]] = 0; }
      std::vector<HYPRE_BigInt> cmap;
      for (HYPRE_Int g = 0; g < nco; g++) { if (renum[(size_t) g] == 0) { renum[(size_t) g] = (HYPRE_Int) cmap.size(); cmap.push_back(f2c_offd[(size_t) g]); } }
      for (HYPRE_Int k = 0; k < nnz_o; k++) { P->offd->j[k] = renum[(size_t) P->offd->j[k]]; }
      P->offd->num_cols = (HYPRE_Int) cmap.size();
      if (P->col_map_offd) { hypre_Free(P->col_map_offd, HYPRE_MEMORY_HOST); P->col_map_offd = nullptr; }
      if (!cmap.empty())
      {
         P->col_map_offd = hypre_TAlloc(HYPRE_BigInt, cmap.size(), HYPRE_MEMORY_HOST);
         memcpy(P->col_map_offd, cmap.data(), sizeof(HYPRE_BigInt) * cmap.size());
      }
      hypre_CSRMatrixSetRownnz(P->offd);
      hypre_MatvecCommPkgCreate(P);
      *P_ptr = P;
      return hypre_error_flag;
   }
   hypre_CSRMatrixSetRownnz(P->offd);
   *P_ptr = P;
   return hypre_error_flag;
}

// ===========================================================================
// Galerkin product on the device (rap_kernels.hip), for host operands of a single rank when a GPU is there: the
// operands are uploaded, P is transposed on the device, the product is formed by one wave per coarse row in the host
// loop's order — same columns, same order, same bits — and comes back to the host, where the next level's coarsening
// still runs.  HYPRE_AMD_SETUP_DEVICE_RAP=0 keeps the host loop (tests compare the two).  Returns false when the
// product was not formed here.
// ===========================================================================


static bool device_galerkin_product(hypre_ParCSRMatrix *RT, hypre_ParCSRMatrix *A, hypre_ParCSRMatrix *P,
                                    HYPRE_Int keepTranspose, hypre_ParCSRMatrix **RAP_ptr)
{
   if (!g_device_rap_on || !device_rap_allowed() || RT != P) { return false; }
   const HYPRE_Int nf = A->diag->num_rows, nc = P->diag->num_cols;
   if (nf < g_device_rap_min_rows || A->diag->num_nonzeros <= 0 || P->diag->num_nonzeros <= 0) { return false; }   // small levels: the host loop is quicker than the transfers
   if (!ensure_device()) { return false; }
   hipStream_t s = stream();
   const bool timing = getenv("HYPRE_AMD_SETUP_TIMING") != nullptr;
   const double t0 = omp_get_wtime();
   hypre_CSRMatrix *dA = device_twin_of(A->diag, 1);
   hypre_CSRMatrix *dP = device_twin_of(P->diag, 1);
   const double t1 = omp_get_wtime();
   hypre_CSRMatrix *dR = nullptr;
   hypre_CSRMatrixTranspose(dP, &dR, 1);            // device transpose: rows of R list the fine rows in ascending order, as the host's
   const double t2 = omp_get_wtime();
   int maxP = 0;
   if (P->diag->memory_location == HYPRE_MEMORY_DEVICE) { maxP = device_max_row_nnz(P->diag->i, nf, s); }
   else { for (HYPRE_Int i = 0; i < nf; i++) { maxP = std::max(maxP, (int) (P->diag->i[i + 1] - P->diag->i[i])); } }
   int *Ci = nullptr, *Cj = nullptr, nnz = 0;
   double *Ca = nullptr;
   const bool ok = device_rap(nc, nc, maxP, dR->i, dR->j, dR->data, dA->i, dA->j, dA->data, dP->i, dP->j, dP->data, &Ci, &Cj, &Ca, &nnz, s);
   const double t3 = omp_get_wtime();
   if (!ok) { hypre_CSRMatrixDestroy(dR); return false; }
   hypre_ParCSRMatrix *C = hypre_ParCSRMatrixCreate(A->comm, RT->global_num_cols, P->global_num_cols, RT->col_starts,
                                                    P->col_starts, 0, nnz, 0);
   // the coarse operator stays where it was made; a later step that runs on the host (a small level, a smoother whose
   // setup is a host loop) fetches it then (make_host_resident) and the device arrays stay on as its twin
   hypre_CSRMatrixDestroy(C->diag);
   C->diag = wrap_device_csr(nc, nc, nnz, Ci, Cj, Ca);
   hypre_CSRMatrixInitialize_v2(C->offd, 0, HYPRE_MEMORY_HOST);
   if (keepTranspose) { RT->diagT = dR; }          // the restriction operator is used on the device only
   else { hypre_CSRMatrixDestroy(dR); }
   hypre_CSRMatrixSetRownnz(C->offd);
   hypre_ParCSRMatrixSetNumNonzeros(C);
   hypre_ParCSRMatrixSetDNumNonzeros(C);
   *RAP_ptr = C;
   g_device_rap_count++;
   if (timing) { fprintf(stderr, "   device RAP: upload %.3fs  transpose %.3fs  product %.3fs  rest %.3fs\n", t1 - t0, t2 - t1, t3 - t2, omp_get_wtime() - t3); }
   return true;
}

// ===========================================================================
// Galerkin product A_c = P^T A P (par_rap.c:30-2000), single-rank form.
// Row ic: diagonal slot first, then RA = sum_{i1 in R(ic,:)} r * A(i1,:)
// accumulated in first-touch order, then RA * P in first-touch order.
// ===========================================================================
HYPRE_Int hypre_BoomerAMGBuildCoarseOperatorKT(hypre_ParCSRMatrix *RT, hypre_ParCSRMatrix *A,
                                               hypre_ParCSRMatrix *P, HYPRE_Int keepTranspose,
                                               hypre_ParCSRMatrix **RAP_ptr)
{
   MPI_Comm comm = A->comm;
   if (comm_size(comm) > 1)
   {
      if (g_level_on_device && RT == P && P->diag->memory_location == HYPRE_MEMORY_DEVICE)
      {
         dist_device_coarse_operator(RT, A, P, keepTranspose, RAP_ptr);       // empty result, clear flag: the host routine takes over
         if (*RAP_ptr) { g_device_rap_count++; }
         return hypre_error_flag;
      }
      return dist_build_coarse_operator(RT, A, P, keepTranspose, RAP_ptr);
   }
   if (device_galerkin_product(RT, A, P, keepTranspose, RAP_ptr)) { return hypre_error_flag; }
   if (A->diag->memory_location != HYPRE_MEMORY_HOST)
   {
      // the host loop below needs a host copy of A: the caller (the setup loop) fetches it and asks again
      *RAP_ptr = nullptr;
      return hypre_error_flag;
   }
   if (P->diag->memory_location == HYPRE_MEMORY_DEVICE) { hypre_CSRMatrixMigrate(P->diag, HYPRE_MEMORY_HOST); }     // made on the device, needed here
   hypre_CSRMatrix *R = nullptr;
   hypre_CSRMatrixTranspose(RT->diag, &R, 1);
   const HYPRE_Int nc = R->num_rows, nf = A->diag->num_rows;
   const HYPRE_Int *Ri = R->i, *Rj = R->j; const HYPRE_Real *Ra = R->data;
   const HYPRE_Int *Ai = A->diag->i, *Aj = A->diag->j; const HYPRE_Real *Aa = A->diag->data;
   const HYPRE_Int *Pi = P->diag->i, *Pj = P->diag->j; const HYPRE_Real *Pa = P->diag->data;
   const bool square = (RT->diag->num_cols == P->diag->num_cols);

   const int T = num_threads_avail();
   std::vector<std::vector<HYPRE_Int>> tj((size_t) T);
   std::vector<std::vector<HYPRE_Real>> ta((size_t) T);
   std::vector<HYPRE_Int> rowlen((size_t) std::max(nc, 1), 0);
#pragma omp parallel num_threads(T)
   {
      const int t = omp_get_thread_num();
      int rb, re;
      chunk(nc, T, t, &rb, &re);
      std::vector<HYPRE_Int> &oj = tj[(size_t) t];
      std::vector<HYPRE_Real> &oa = ta[(size_t) t];
      RowMap Pmark(1024), Amark(1024);         // column -> position in the row being formed
      std::vector<HYPRE_Int> raj;
      std::vector<HYPRE_Real> raa;
      (void) nf;
      for (HYPRE_Int ic = rb; ic < re; ic++)
      {
         const long long begin = (long long) oj.size();
         if (square) { Pmark.set(ic, begin); oj.push_back(ic); oa.push_back(0.0); }
         raj.clear(); raa.clear();
         for (HYPRE_Int j1 = Ri[ic]; j1 < Ri[ic + 1]; j1++)
         {
            const HYPRE_Int i1 = Rj[j1];
            const HYPRE_Real r = Ra[j1];
            for (HYPRE_Int j2 = Ai[i1]; j2 < Ai[i1 + 1]; j2++)
            {
               const HYPRE_Int i2 = Aj[j2];
               const long long m = Amark.get(i2, -1);
               if (m < 0)
               {
                  Amark.set(i2, (long long) raj.size());
                  raj.push_back(i2); raa.push_back(r * Aa[j2]);
               }
               else { raa[(size_t) m] += r * Aa[j2]; }
            }
         }
         for (size_t q = 0; q < raj.size(); q++)
         {
            const HYPRE_Int i1 = raj[q];
            const HYPRE_Real rap = raa[q];
            for (HYPRE_Int j2 = Pi[i1]; j2 < Pi[i1 + 1]; j2++)
            {
               const HYPRE_Int i2 = Pj[j2];
               const long long m = Pmark.get(i2, -1);
               if (m < begin) { Pmark.set(i2, (long long) oj.size()); oj.push_back(i2); oa.push_back(rap * Pa[j2]); }
               else { oa[(size_t) m] += rap * Pa[j2]; }
            }
         }
         rowlen[(size_t) ic] = (HYPRE_Int) ((long long) oj.size() - begin);
         Amark.clear(); Pmark.clear();
      }
   }
   std::vector<HYPRE_Int> Ci((size_t) nc + 1, 0);
   for (HYPRE_Int i = 0; i < nc; i++) { Ci[(size_t) i + 1] = Ci[(size_t) i] + rowlen[(size_t) i]; }
   hypre_ParCSRMatrix *C = hypre_ParCSRMatrixCreate(comm, RT->global_num_cols, P->global_num_cols, RT->col_starts,
                                                    P->col_starts, 0, Ci[(size_t) nc], 0);
   hypre_ParCSRMatrixInitialize_v2(C, HYPRE_MEMORY_HOST);
   memcpy(C->diag->i, Ci.data(), sizeof(HYPRE_Int) * ((size_t) nc + 1));
#pragma omp parallel num_threads(T)
   {
      const int t = omp_get_thread_num();
      int rb, re;
      chunk(nc, T, t, &rb, &re);
      if (re > rb && !tj[(size_t) t].empty())
      {
         const size_t off = (size_t) Ci[(size_t) rb];
         memcpy(C->diag->j + off, tj[(size_t) t].data(), sizeof(HYPRE_Int) * tj[(size_t) t].size());
         memcpy(C->diag->data + off, ta[(size_t) t].data(), sizeof(HYPRE_Real) * ta[(size_t) t].size());
      }
   }
   if (keepTranspose) { RT->diagT = R; } else { hypre_CSRMatrixDestroy(R); }
   hypre_CSRMatrixSetRownnz(C->offd);
   hypre_ParCSRMatrixSetNumNonzeros(C);
   hypre_ParCSRMatrixSetDNumNonzeros(C);
   *RAP_ptr = C;
   return hypre_error_flag;
}

// ===========================================================================
// smoother diagonals (ams.c:527-830), host
// ===========================================================================
HYPRE_Int hypre_ParCSRComputeL1Norms(hypre_ParCSRMatrix *A, HYPRE_Int option, HYPRE_Int *cf_marker,
                                     HYPRE_Real **l1_norm_ptr)
{
   hypre_CSRMatrix *D = A->diag, *O = A->offd;
   if (D->memory_location != HYPRE_MEMORY_HOST)
   {
      hypre_error_w_msg(HYPRE_ERROR_GENERIC, "hypre_ParCSRComputeL1Norms: setup-time routine, expects host matrices");
      return hypre_error_flag;
   }
   const HYPRE_Int n = D->num_rows, nco = O->num_cols;
   HYPRE_Real *l1 = hypre_TAlloc(HYPRE_Real, std::max(n, 1), HYPRE_MEMORY_HOST);
   std::vector<HYPRE_Int> cf_offd;
   const HYPRE_Int *cfo = nullptr;
   if (cf_marker && nco)
   {
      cf_offd.resize((size_t) nco);
      if (!A->comm_pkg) { hypre_MatvecCommPkgCreate(A); }
      halo_forward<HYPRE_Int>(A->comm_pkg, cf_marker, cf_offd.data());
      cfo = cf_offd.data();
   }
   auto abs_sum = [&](hypre_CSRMatrix *M, const HYPRE_Int *ci, const HYPRE_Int *cj, HYPRE_Real *out, double scal, bool add)
   {
#pragma omp parallel for schedule(static)
      for (HYPRE_Int i = 0; i < n; i++)
      {
         HYPRE_Real s = add ? out[i] : 0.0;
         for (HYPRE_Int k = M->i[i]; k < M->i[i + 1]; k++)
         {
            if (ci && cj && ci[i] != cj[M->j[k]]) { continue; }
            s += scal * std::fabs(M->data[k]);
         }
         out[i] = s;
      }
   };
   auto get_diag = [&](HYPRE_Real *d, bool take_abs)
   {
#pragma omp parallel for schedule(static)
      for (HYPRE_Int i = 0; i < n; i++)
      {
         HYPRE_Real v = 0.0;
         for (HYPRE_Int k = D->i[i]; k < D->i[i + 1]; k++) { if (D->j[k] == i) { v = take_abs ? std::fabs(D->data[k]) : D->data[k]; break; } }
         d[i] = v;
      }
   };
   std::vector<HYPRE_Real> tmp((size_t) std::max(n, 1));
   if (option == 1)
   {
      abs_sum(D, cf_marker, cf_marker, l1, 1.0, false);
      if (nco) { abs_sum(O, cf_marker, cfo, l1, 1.0, true); }
   }
   else if (option == 4)
   {
      get_diag(l1, true);
      memcpy(tmp.data(), l1, sizeof(HYPRE_Real) * (size_t) n);
      if (nco) { abs_sum(O, cf_marker, cfo, l1, 0.5, true); }
      for (HYPRE_Int i = 0; i < n; i++) { if (l1[i] <= 4.0 / 3.0 * tmp[(size_t) i]) { l1[i] = tmp[(size_t) i]; } }
   }
   else if (option == 5)
   {
      get_diag(l1, false);
      for (HYPRE_Int i = 0; i < n; i++) { if (l1[i] == 0.0) { l1[i] = 1.0; } }
      *l1_norm_ptr = l1;
      return hypre_error_flag;
   }
   else if (option == 6)
   {
      get_diag(l1, true);
      if (nco)
      {
         abs_sum(O, cf_marker, cfo, tmp.data(), 1.0, false);
         for (HYPRE_Int i = 0; i < n; i++)
         {
            l1[i] = 0.5 * (tmp[(size_t) i] + l1[i] + std::sqrt(tmp[(size_t) i] * tmp[(size_t) i] + l1[i] * l1[i]));
         }
      }
   }
   else
   {
      hypre_error_w_msg(HYPRE_ERROR_GENERIC, "hypre_ParCSRComputeL1Norms: option not supported");
   }
   get_diag(tmp.data(), false);
   for (HYPRE_Int i = 0; i < n; i++) { if (tmp[(size_t) i] < 0.0) { l1[i] = -l1[i]; } }
   for (HYPRE_Int i = 0; i < n; i++) { if (std::fabs(l1[i]) == 0.0) { hypre_error_in_arg(1); break; } }
   *l1_norm_ptr = l1;
   return hypre_error_flag;
}

// Smoother diagonals when the hybrid sweeps run in num_threads row blocks
// (ams.c:4535-4915): a diag-block entry whose column lies in another thread's
// block is treated like a ghost coupling (options 4 and 6), sums accumulate in
// stored order, and the sign flip looks at the first entry of the row.
HYPRE_Int hypre_ParCSRComputeL1NormsThreads(hypre_ParCSRMatrix *A, HYPRE_Int option, HYPRE_Int num_threads,
                                            HYPRE_Int *cf_marker, HYPRE_Real **l1_norm_ptr)
{
   hypre_CSRMatrix *D = A->diag, *O = A->offd;
   if (D->memory_location != HYPRE_MEMORY_HOST)
   {
      hypre_error_w_msg(HYPRE_ERROR_GENERIC, "hypre_ParCSRComputeL1NormsThreads: setup-time routine, expects host matrices");
      return hypre_error_flag;
   }
   if (option != 1 && option != 4 && option != 5 && option != 6)
   {
      hypre_error_w_msg(HYPRE_ERROR_GENERIC, "hypre_ParCSRComputeL1NormsThreads: option not supported");
      return hypre_error_flag;
   }
   const HYPRE_Int n = D->num_rows, nco = O->num_cols;
   if (num_threads < 1) { num_threads = 1; }
   HYPRE_Real *l1 = hypre_TAlloc(HYPRE_Real, std::max(n, 1), HYPRE_MEMORY_HOST);
   std::vector<HYPRE_Int> cf_offd;
   const HYPRE_Int *cfo = nullptr;
   if (cf_marker && nco)
   {
      cf_offd.resize((size_t) nco);
      if (!A->comm_pkg) { hypre_MatvecCommPkgCreate(A); }
      halo_forward<HYPRE_Int>(A->comm_pkg, cf_marker, cf_offd.data());
      cfo = cf_offd.data();
   }
   bool zero_norm = false;
   for (HYPRE_Int k = 0; k < num_threads; k++)
   {
      const HYPRE_Int size = n / num_threads, rest = n - size * num_threads;
      const HYPRE_Int ns = k < rest ? k * size + k : k * size + rest;
      const HYPRE_Int ne = k < rest ? (k + 1) * size + k + 1 : (k + 1) * size + rest;
#pragma omp parallel for schedule(static)
      for (HYPRE_Int i = ns; i < ne; i++)
      {
         if (option == 5)
         {
            l1[i] = D->data[D->i[i]];
            if (l1[i] == 0.0) { l1[i] = 1.0; }
            continue;
         }
         const bool use_cf = cf_marker != nullptr;
         const HYPRE_Int cfd = use_cf ? cf_marker[i] : 0;
         HYPRE_Real s = 0.0, dg = 0.0;
         for (HYPRE_Int j = D->i[i]; j < D->i[i + 1]; j++)
         {
            const HYPRE_Int ii = D->j[j];
            if (use_cf && cfd != cf_marker[ii]) { continue; }
            if (option == 1) { s += std::fabs(D->data[j]); continue; }
            if (ii == i)
            {
               dg = std::fabs(D->data[j]);
               if (option == 4) { s += dg; }
            }
            else if (ii < ns || ii >= ne) { s += 0.5 * std::fabs(D->data[j]); }
         }
         if (nco)
         {
            for (HYPRE_Int j = O->i[i]; j < O->i[i + 1]; j++)
            {
               if (use_cf && cfd != cfo[O->j[j]]) { continue; }
               s += (option == 1 ? 1.0 : 0.5) * std::fabs(O->data[j]);
            }
         }
         if (option == 4) { if (s <= (4.0 / 3.0) * dg) { s = dg; } }
         if (option == 6) { s = (dg + s + std::sqrt(dg * dg + s * s)) * 0.5; }
         if (option < 5 && D->data[D->i[i]] < 0) { s = -s; }
         l1[i] = s;
      }
   }
   if (option < 5)
   {
      for (HYPRE_Int i = 0; i < n; i++) { if (std::fabs(l1[i]) == 0.0) { zero_norm = true; break; } }
      if (zero_norm) { hypre_error_in_arg(1); }
   }
   *l1_norm_ptr = l1;
   return hypre_error_flag;
}

// ===========================================================================
// coarsest level: dense copy of the operator (par_gauss_elim.c:25-300)
// ===========================================================================
HYPRE_Int hypre_GaussElimSetup(hypre_ParAMGData *d, HYPRE_Int level, HYPRE_Int relax_type)
{
   (void) relax_type;
   hypre_ParCSRMatrix *A = d->A_array[level];
   const HYPRE_Int n = (HYPRE_Int) A->global_num_rows;
   const HYPRE_Int nloc = A->diag->num_rows;
   const HYPRE_Int first = (HYPRE_Int) A->first_row_index;
   std::vector<double> local((size_t) std::max(nloc, 1) * (size_t) std::max(n, 1), 0.0);
   for (HYPRE_Int i = 0; i < nloc; i++)
   {
      for (HYPRE_Int k = A->diag->i[i]; k < A->diag->i[i + 1]; k++) { local[(size_t) i * n + (size_t) (A->diag->j[k] + first)] = A->diag->data[k]; }
      for (HYPRE_Int k = A->offd->i[i]; k < A->offd->i[i + 1]; k++) { local[(size_t) i * n + (size_t) A->col_map_offd[A->offd->j[k]]] = A->offd->data[k]; }
   }
   free(d->A_mat); free(d->b_vec);
   d->A_mat = (HYPRE_Real *) calloc((size_t) std::max(n, 1) * (size_t) std::max(n, 1), sizeof(HYPRE_Real));
   d->b_vec = (HYPRE_Real *) calloc((size_t) std::max(n, 1), sizeof(HYPRE_Real));
   const hypre_amd_CommOps *o = comm_ops(A->comm);
   if (o && o->size > 1)
   {
      // rows are contiguous per rank in rank order: gather row counts, then padded blocks
      std::vector<HYPRE_Int> counts((size_t) o->size);
      o->allgather(o->ctx, &nloc, counts.data(), sizeof(HYPRE_Int));
      HYPRE_Int maxc = 0;
      for (int r = 0; r < o->size; r++) { maxc = std::max(maxc, counts[(size_t) r]); }
      std::vector<double> sendb((size_t) std::max(maxc, 1) * (size_t) n, 0.0), all((size_t) std::max(maxc, 1) * (size_t) n * (size_t) o->size);
      memcpy(sendb.data(), local.data(), sizeof(double) * (size_t) nloc * (size_t) n);
      o->allgather(o->ctx, sendb.data(), all.data(), sizeof(double) * sendb.size());
      HYPRE_Int row = 0;
      for (int r = 0; r < o->size; r++)
      {
         memcpy(d->A_mat + (size_t) row * n, all.data() + (size_t) r * sendb.size(), sizeof(double) * (size_t) counts[(size_t) r] * (size_t) n);
         row += counts[(size_t) r];
      }
   }
   else
   {
      memcpy(d->A_mat, local.data(), sizeof(double) * (size_t) nloc * (size_t) n);
   }
   d->gs_setup = 1;
   return hypre_error_flag;
}

// ===========================================================================
// the level loop (par_amg_setup.c:28-3560 reduced to the in-scope options)
// ===========================================================================
// A without the couplings between different functions (parcsr_mv/par_csr_filter.c:21-186): entry (i, j) stays
// when i and j are the same unknown of their grid points (equal index modulo block_size); ghost columns that lose
// all their entries leave the column map.
static hypre_ParCSRMatrix *blk_filter(hypre_ParCSRMatrix *A, HYPRE_Int block_size)
{
   if (A->global_num_rows % block_size || A->row_starts[0] % block_size || A->global_num_rows != A->global_num_cols)
   {
      hypre_error_w_msg(HYPRE_ERROR_GENERIC, "block size must divide the number of rows and the first row of every rank (square matrices only)");
      return nullptr;
   }
   const hypre_CSRMatrix *Ad = A->diag, *Ao = A->offd;
   const HYPRE_Int n = Ad->num_rows, nco = Ao->num_cols;
   std::vector<HYPRE_Int> di((size_t) n + 1, 0), oi((size_t) n + 1, 0), dj, oj;
   std::vector<HYPRE_Real> da, oa;
   std::vector<char> used((size_t) std::max(nco, 1), 0);
   for (HYPRE_Int i = 0; i < n; i++)
   {
      const HYPRE_Int c = i % block_size;
      for (HYPRE_Int k = Ad->i[i]; k < Ad->i[i + 1]; k++)
      {
         if (c == Ad->j[k] % block_size) { dj.push_back(Ad->j[k]); da.push_back(Ad->data[k]); }
      }
      for (HYPRE_Int k = Ao->i[i]; k < Ao->i[i + 1]; k++)
      {
         if (c == (HYPRE_Int) (A->col_map_offd[Ao->j[k]] % (HYPRE_BigInt) block_size))
         {
            oj.push_back(Ao->j[k]); oa.push_back(Ao->data[k]); used[(size_t) Ao->j[k]] = 1;
         }
      }
      di[(size_t) i + 1] = (HYPRE_Int) dj.size();
      oi[(size_t) i + 1] = (HYPRE_Int) oj.size();
   }
   std::vector<HYPRE_Int> renum((size_t) std::max(nco, 1), -1);
   std::vector<HYPRE_BigInt> cmap;
   for (HYPRE_Int c = 0; c < nco; c++) { if (used[(size_t) c]) { renum[(size_t) c] = (HYPRE_Int) cmap.size(); cmap.push_back(A->col_map_offd[c]); } }
   for (HYPRE_Int &c : oj) { c = renum[(size_t) c]; }
   hypre_ParCSRMatrix *Bm = hypre_amd_ParCSRMatrixFromArrays(A->comm, A->global_num_rows, A->global_num_cols, A->row_starts,
                                                             A->col_starts, (HYPRE_Int) cmap.size(), cmap.data(), di.data(),
                                                             dj.data(), da.data(), oi.data(), oj.data(), oa.data(), HYPRE_MEMORY_HOST);
   return Bm;
}

static hypre_ParVector *new_vec(MPI_Comm comm, HYPRE_BigInt gsize, HYPRE_BigInt *part, HYPRE_MemoryLocation loc)
{
   hypre_ParVector *v = hypre_ParVectorCreate(comm, gsize, part);
   hypre_ParVectorInitialize_v2(v, loc);
   return v;
}

namespace {
// The flags that route the builders to the device, the twins and the markers belong to ONE run of the setup: set when it
// starts, cleared on every way out (a standalone call of hypre_BoomerAMGBuildExtPIInterp or ..CoarseOperatorKT afterwards
// works where its operands are, as the reference's do).
struct SetupScope
{
   int saved_threads;
   explicit SetupScope(bool device) : saved_threads(omp_get_max_threads())
   {
      g_setup_targets_device = device;
      g_level_on_device = false;
      g_interp_host_once = false;
      drop_device_twins();
   }
   ~SetupScope()
   {
      g_setup_targets_device = false;
      g_level_on_device = false;
      g_interp_host_once = false;
      drop_device_twins();
      omp_set_num_threads(saved_threads);
   }
};
}  // namespace

HYPRE_Int hypre_BoomerAMGSetup(void *amg_vdata, hypre_ParCSRMatrix *A, hypre_ParVector *f, hypre_ParVector *u)
{
   (void) f; (void) u;
   hypre_ParAMGData *d = (hypre_ParAMGData *) amg_vdata;
   if (!d) { hypre_error_in_arg(1); return hypre_error_flag; }
   AmgPrivate *pv = (AmgPrivate *) d->amd_private;
   amg_free_hierarchy(d);
   // A (re-)setup is the caller telling the library that the matrix is what it is NOW: whatever earlier products cached for
   // its two blocks — tile tables, staged column pattern, value codes and slice form, triangles, colour classes, level
   // schedules — is dropped before anything else, so that the hypre idiom "HYPRE_IJMatrixSetValues on the same pattern,
   // then HYPRE_BoomerAMGSetup again" works on the new values on every level, the finest included
   // (the reference reads the caller's arrays in every product: seq_mv/csr_matvec.c:860-901).
   if (A->diag) { drop_plan(A->diag); }
   if (A->offd) { drop_plan(A->offd); }
   MPI_Comm comm = A->comm;
   const HYPRE_MemoryLocation target = d->memory_location;
   if (target == HYPRE_MEMORY_DEVICE && !ensure_device())
   {
      hypre_error_w_msg(HYPRE_ERROR_GENERIC, "hypre_BoomerAMGSetup: hierarchy requested in device memory but no HIP device is available");
      return hypre_error_flag;
   }
   SetupScope scope(target == HYPRE_MEMORY_DEVICE);
   const int max_levels = d->max_levels;
   d->A = A;
   d->A_array = (hypre_ParCSRMatrix **) calloc((size_t) max_levels, sizeof(void *));
   d->P_array = (hypre_ParCSRMatrix **) calloc((size_t) max_levels, sizeof(void *));
   d->R_array = d->P_array;
   d->F_array = (hypre_ParVector **) calloc((size_t) max_levels, sizeof(void *));
   d->U_array = (hypre_ParVector **) calloc((size_t) max_levels, sizeof(void *));
   d->CF_marker_array = (hypre_IntArray **) calloc((size_t) max_levels, sizeof(void *));
   d->l1_norms = (hypre_Vector **) calloc((size_t) max_levels, sizeof(void *));
   d->relax_weight = (HYPRE_Real *) calloc((size_t) max_levels, sizeof(HYPRE_Real));
   d->omega = (HYPRE_Real *) calloc((size_t) max_levels, sizeof(HYPRE_Real));
   for (int l = 0; l < max_levels; l++) { d->relax_weight[l] = d->user_relax_weight; d->omega[l] = d->outer_wt; }
   {
      AmgPrivate *pvw = (AmgPrivate *) d->amd_private;
      for (auto &lw : pvw->level_relax_wt) { if (lw.first < max_levels) { d->relax_weight[lw.first] = lw.second; } }
      for (auto &lw : pvw->level_outer_wt) { if (lw.first < max_levels) { d->omega[lw.first] = lw.second; } }
   }

   // the setup works on host copies; a device-resident A is cloned once
   // (a level that qualifies — single rank, one function, PMIS, extended+i, no small level — is worked on where the
   // hierarchy will live: strength, coarsening, interpolation and the Galerkin product on the device, nothing fetched)
   std::vector<hypre_ParCSRMatrix *> hostA((size_t) max_levels, nullptr);
   const bool A_on_device = A->diag->memory_location == HYPRE_MEMORY_DEVICE;
   const bool single_rank = comm_size(comm) == 1;
   const int num_ranks = comm_size(comm);
   auto level_runs_on_device = [&](hypre_ParCSRMatrix *M, HYPRE_Int ct)
   {
      if (!(target == HYPRE_MEMORY_DEVICE && g_device_coarsen_on && g_device_interp_on && g_device_rap_on &&
            d->num_functions <= 1 && (ct == 8 || ct == 9) && d->interp_type == 6)) { return false; }
      if (single_rank)
      {
         return M->offd->num_cols == 0 && M->offd->num_nonzeros == 0 && M->diag->num_rows >= g_device_rap_min_rows && M->diag->num_nonzeros > 0;
      }
      // several ranks (par_amg_setup_dist.cpp, the device half): every rank has to take the same road, so the test is on
      // the level's global size; a rank with few rows, or none, goes along
      return g_device_dist_on && M->global_num_rows / num_ranks >= (HYPRE_BigInt) g_device_rap_min_rows;
   };
   const bool A_stays = A_on_device && level_runs_on_device(A, d->coarsen_type);
   hostA[0] = (A_on_device && !A_stays) ? hypre_ParCSRMatrixClone_v2(A, 1, HYPRE_MEMORY_HOST) : A;
   bool own_host_A0 = A_on_device && !A_stays;
   // a host loop needs level l after all: the caller's matrix is copied, a matrix of ours is fetched (and keeps its twin)
   auto fetch_level = [&](int l)
   {
      hypre_ParCSRMatrix *M = hostA[(size_t) l];
      if (!M || (M->diag->memory_location != HYPRE_MEMORY_DEVICE && M->offd->memory_location != HYPRE_MEMORY_DEVICE)) { return; }
      if (l == 0 && M == A) { hostA[0] = hypre_ParCSRMatrixClone_v2(A, 1, HYPRE_MEMORY_HOST); own_host_A0 = true; }
      else { make_host_resident(M->diag); make_host_resident(M->offd); }
   };
   if (d->num_functions > 1 && pv->filter_functions)
   {
      // par_amg_setup.c:774-780: the hierarchy (S, P, coarse operators, smoother diagonals) comes from the filtered
      // matrix; the solve phase still smooths with the caller's A on level 0 (par_amg_solve.c:109)
      hypre_ParCSRMatrix *At = blk_filter(hostA[0], d->num_functions);
      if (!At) { if (own_host_A0) { hypre_ParCSRMatrixDestroy(hostA[0]); } return hypre_error_flag; }
      if (own_host_A0) { hypre_ParCSRMatrixDestroy(hostA[0]); }
      hostA[0] = At;
      own_host_A0 = true;
   }
   d->A_array[0] = A;
   if (hostA[0]->d_num_nonzeros < 0) { hypre_ParCSRMatrixSetDNumNonzeros(hostA[0]); A->d_num_nonzeros = hostA[0]->d_num_nonzeros; }

   int level = 0;
   bool not_finished = max_levels > 1;     // the loop always coarsens once (par_amg_setup.c:960-990)
   HYPRE_Int coarsen_type = d->coarsen_type;
   HYPRE_BigInt fine_size = A->global_num_rows, coarse_size = fine_size;
   if (max_levels == 1)
   {
      d->CF_marker_array[0] = hypre_IntArrayCreate(A->diag->num_rows);
      hypre_IntArrayInitialize_v2(d->CF_marker_array[0], HYPRE_MEMORY_HOST);
      for (HYPRE_Int i = 0; i < A->diag->num_rows; i++) { d->CF_marker_array[0]->data[i] = 1; }
   }
   // systems, unknown approach: function of every row; (first row + i) mod num_functions unless the rows
   // say otherwise (par_amg_setup.c:752-771), C-points carry theirs to the next level (par_coarse_parms.c)
   std::vector<HYPRE_Int> dof_func;
   if (d->num_functions > 1)
   {
      const HYPRE_Int n0 = A->diag->num_rows;
      const HYPRE_Int offset = (HYPRE_Int) (A->first_row_index % (HYPRE_BigInt) d->num_functions);
      dof_func.resize((size_t) n0);
      for (HYPRE_Int i = 0; i < n0; i++) { dof_func[(size_t) i] = (i + offset) % d->num_functions; }
   }
   // host threads: no more than the cores this process owns, and no more than a level's rows can keep busy
   const int saved_omp_threads = omp_get_max_threads();
   const int thread_cap = std::max(1, std::min(saved_omp_threads, host_cpu_share()));
   omp_set_num_threads(thread_cap);
   while (not_finished)
   {
      g_level_on_device = level_runs_on_device(hostA[(size_t) level], coarsen_type);
      if (!g_level_on_device) { fetch_level(level); }
      hypre_ParCSRMatrix *Al = hostA[(size_t) level];
      omp_set_num_threads(std::max(1, std::min(thread_cap, Al->diag->num_rows / 4096)));
      fine_size = Al->global_num_rows;
      if (level > 0)
      {
         d->F_array[level] = new_vec(comm, fine_size, Al->row_starts, target);
         d->U_array[level] = new_vec(comm, fine_size, Al->row_starts, target);
      }
      hypre_ParCSRMatrix *S = nullptr;
      const double t_s0 = omp_get_wtime();
      HYPRE_Int *dofs = d->num_functions > 1 ? dof_func.data() : nullptr;
      hypre_BoomerAMGCreateS(Al, d->strong_threshold, d->max_row_sum, d->num_functions, dofs, &S);
      const double t_s1 = omp_get_wtime();
      const HYPRE_Int nloc = Al->diag->num_rows;
      d->CF_marker_array[level] = hypre_IntArrayCreate(nloc);
      hypre_IntArrayInitialize_v2(d->CF_marker_array[level], HYPRE_MEMORY_HOST);
      if (coarsen_type == 8) { hypre_BoomerAMGCoarsenPMIS(S, Al, 0, 0, &d->CF_marker_array[level]); }
      else if (coarsen_type == 9) { hypre_BoomerAMGCoarsenPMIS(S, Al, 2, 0, &d->CF_marker_array[level]); }
      else if (coarsen_type == 10) { hypre_BoomerAMGCoarsenHMIS(S, Al, d->measure_type, 0, 0, &d->CF_marker_array[level]); }
      else
      {
         hypre_error_w_msg(HYPRE_ERROR_GENERIC, "hypre_BoomerAMGSetup: coarsen_type must be 8 (PMIS), 9 (PMIS, sequential random numbers) or 10 (HMIS)");
         destroy_with_twins(S);
         break;
      }
      const double t_c1 = omp_get_wtime();
      HYPRE_Int *CF = d->CF_marker_array[level]->data;
      HYPRE_BigInt cpts[2];
      coarse_parms(comm, nloc, CF, cpts, &coarse_size);
      std::vector<HYPRE_Int> coarse_dof_func;
      if (d->num_functions > 1)
      {
         for (HYPRE_Int i = 0; i < nloc; i++) { if (CF[i] == 1) { coarse_dof_func.push_back(dof_func[(size_t) i]); } }
      }
      if (coarse_size == 0 || coarse_size == fine_size)
      {
         // no coarse grid: one sweep of the default smoother on the last level (par_amg_setup.c:1655-1690)
         HYPRE_Int *gt = d->grid_relax_type;
         if (gt[3] == 9 || gt[3] == 99 || gt[3] == 19 || gt[3] == 98)
         {
            gt[3] = gt[0];
            d->num_grid_sweeps[3] = 1;
            if (d->grid_relax_points) { d->grid_relax_points[3][0] = 0; }
         }
         destroy_with_twins(S);
         if (level > 0)
         {
            hypre_IntArrayDestroy(d->CF_marker_array[level]); d->CF_marker_array[level] = nullptr;
            hypre_ParVectorDestroy(d->F_array[level]); d->F_array[level] = nullptr;
            hypre_ParVectorDestroy(d->U_array[level]); d->U_array[level] = nullptr;
         }
         coarse_size = fine_size;
         break;
      }
      if (coarse_size < (HYPRE_BigInt) d->min_coarse_size)
      {
         destroy_with_twins(S);
         hypre_IntArrayDestroy(d->CF_marker_array[level]); d->CF_marker_array[level] = nullptr;
         if (level > 0)
         {
            hypre_ParVectorDestroy(d->F_array[level]); d->F_array[level] = nullptr;
            hypre_ParVectorDestroy(d->U_array[level]); d->U_array[level] = nullptr;
         }
         coarse_size = fine_size;
         break;
      }
      const double t_i0 = omp_get_wtime();
      hypre_ParCSRMatrix *P = nullptr;
      if (d->interp_type == 6)
      {
         hypre_BoomerAMGBuildExtPIInterp(Al, CF, S, cpts, d->num_functions, dofs, 0, d->trunc_factor, d->P_max_elmts, &P);
         if (!P && !hypre_error_flag)
         {
            // the device kernel declined (a row beyond its tables) and the level lives on the device: host copies, host loop
            fetch_level(level);
            Al = hostA[(size_t) level];
            if (single_rank) { make_host_resident(S->diag); g_interp_host_once = true; }
            else
            {
               // (every rank is here: the verdict was agreed on)  the strength matrix again, as the host routine's two blocks
               destroy_with_twins(S);
               S = nullptr;
               g_level_on_device = false;
               hypre_BoomerAMGCreateS(Al, d->strong_threshold, d->max_row_sum, d->num_functions, dofs, &S);
            }
            hypre_BoomerAMGBuildExtPIInterp(Al, CF, S, cpts, d->num_functions, dofs, 0, d->trunc_factor, d->P_max_elmts, &P);
         }
      }
      else if (d->interp_type == 3)
      {
         hypre_BoomerAMGBuildDirInterp(Al, CF, S, cpts, d->num_functions, dofs, 0, d->trunc_factor, d->P_max_elmts, 0, &P);
      }
      else
      {
         hypre_error_w_msg(HYPRE_ERROR_GENERIC, "hypre_BoomerAMGSetup: interp_type must be 6 (extended+i) or 3 (direct)");
      }
      destroy_with_twins(S);
      if (!P || hypre_error_flag) { break; }
      hypre_ParCSRMatrixSetNumNonzeros(P);
      hypre_ParCSRMatrixSetDNumNonzeros(P);
      d->P_array[level] = P;
      const double t_r0 = omp_get_wtime();
      hypre_ParCSRMatrix *AH = nullptr;
      hypre_BoomerAMGBuildCoarseOperatorKT(P, Al, P, 1, &AH);
      if (!AH && !hypre_error_flag && (Al->diag->memory_location == HYPRE_MEMORY_DEVICE || (!single_rank && g_level_on_device)))
      {
         fetch_level(level);
         Al = hostA[(size_t) level];
         if (!single_rank)
         {
            // (agreed on by every rank)  the host routine reads the interpolation operator's two blocks as well
            make_host_resident(P->diag); make_host_resident(P->offd);
            g_level_on_device = false;
         }
         hypre_BoomerAMGBuildCoarseOperatorKT(P, Al, P, 1, &AH);
      }
      if (getenv("HYPRE_AMD_SETUP_TIMING") && comm_rank(comm) == 0)
      {
         fprintf(stderr, "setup level %d: rows %lld  strength %.2fs  coarsen %.2fs  interp %.2fs  RAP %.2fs  (threads %d)\n", level,
                 (long long) fine_size, t_s1 - t_s0, t_c1 - t_s1, t_r0 - t_i0, omp_get_wtime() - t_r0, omp_get_max_threads());
      }
      if (!AH || hypre_error_flag) { break; }
      ++level;
      dof_func.swap(coarse_dof_func);
      hostA[(size_t) level] = AH;
      d->A_array[level] = AH;
      // (par_amg_setup.c:3128-3136) switch to plain CLJP-free coarsening when the grid barely shrinks
      if (coarsen_type > 0 && (double) coarse_size >= 0.75 * (double) fine_size) { coarsen_type = 0; }
      if (level == max_levels - 1 || coarse_size <= (HYPRE_BigInt) d->max_coarse_size) { not_finished = false; }
      if (coarsen_type == 0)
      {
         hypre_error_w_msg(HYPRE_ERROR_GENERIC, "hypre_BoomerAMGSetup: coarsening stalled (coarse grid >= 75% of fine); CLJP fallback is out of scope");
         not_finished = false;
      }
   }
   g_level_on_device = false;
   omp_set_num_threads(thread_cap);
   const int num_levels = level + 1;
   d->num_levels = num_levels;
   if (num_levels > 1 && !d->F_array[num_levels - 1])
   {
      hypre_ParCSRMatrix *Al = hostA[(size_t) num_levels - 1];
      d->F_array[num_levels - 1] = new_vec(comm, Al->global_num_rows, Al->row_starts, target);
      d->U_array[num_levels - 1] = new_vec(comm, Al->global_num_rows, Al->row_starts, target);
   }

   // coarsest-level solver (par_amg_setup.c:3165-3200)
   {
      HYPRE_Int *gt = d->grid_relax_type;
      if (gt[3] == 9 || gt[3] == 19 || gt[3] == 98 || gt[3] == 99 || gt[3] == 198 || gt[3] == 199)
      {
         hypre_ParCSRMatrix *Ac = hostA[(size_t) num_levels - 1];
         if (Ac->global_num_rows <= (HYPRE_BigInt) d->max_coarse_size)
         {
            fetch_level(num_levels - 1);
            Ac = hostA[(size_t) num_levels - 1];
            // (the dense factorisation reads a host copy; level 0 of a one-level hierarchy stays the caller's matrix)
            hypre_ParCSRMatrix *keep = d->A_array[num_levels - 1];
            d->A_array[num_levels - 1] = Ac;
            hypre_GaussElimSetup(d, num_levels - 1, gt[3]);
            d->A_array[num_levels - 1] = keep;
         }
         else { gt[3] = gt[1]; }
      }
   }

   // smoother diagonals per level (par_amg_setup.c:3296-3500)
   {
      const HYPRE_Int *gt = d->grid_relax_type;
      for (int j = 0; j < num_levels; j++)
      {
         HYPRE_Real *l1 = nullptr;
         bool l1_on_device = false;
         HYPRE_Int *cf = (d->relax_order && d->CF_marker_array[j]) ? d->CF_marker_array[j]->data : nullptr;
         const bool last = (j == num_levels - 1);
         auto any = [&](int a, int b, int c, int e) { return gt[1] == a || gt[1] == b || gt[1] == c || gt[1] == e ||
                                                            gt[2] == a || gt[2] == b || gt[2] == c || gt[2] == e; };
         auto last_is = [&](int a, int b, int c, int e) { return gt[3] == a || gt[3] == b || gt[3] == c || gt[3] == e; };
         // ams.c:541-548: the host routine switches to the thread-block variant when OpenMP has > 1 thread
         const int gs_threads = ((AmgPrivate *) d->amd_private)->emulated_threads;
         auto l1_norms = [&](HYPRE_Int option, HYPRE_Int *cfm)
         {
            if (l1) { hypre_Free(l1, l1_on_device ? HYPRE_MEMORY_DEVICE : HYPRE_MEMORY_HOST); l1 = nullptr; }
            hypre_ParCSRMatrix *M = hostA[(size_t) j];
            if (M->diag->memory_location == HYPRE_MEMORY_DEVICE && gs_threads <= 1 &&
                (M->offd->num_cols == 0 || M->offd->memory_location == HYPRE_MEMORY_DEVICE))
            {
               // a level that never left the device: one thread per row, the host routine's sums in the host's order
               const HYPRE_Int n = M->diag->num_rows, nco = M->offd->num_cols;
               l1 = hypre_TAlloc(HYPRE_Real, (size_t) std::max(n, 1), HYPRE_MEMORY_DEVICE);
               l1_on_device = true;
               const HYPRE_Int *dcf = cfm ? device_marker_of(cfm, n) : nullptr;
               bool fine;
               if (nco == 0) { fine = device_l1_norms(n, M->diag->i, M->diag->j, M->diag->data, option, dcf, l1, stream()); }
               else
               {
                  // ghost columns: their markers come from the owners (CF-ordered relaxation only)
                  HYPRE_Int *dcfo = nullptr;
                  if (cfm)
                  {
                     std::vector<HYPRE_Int> cfo((size_t) nco);
                     if (!M->comm_pkg) { hypre_MatvecCommPkgCreate(M); }
                     halo_forward<HYPRE_Int>(M->comm_pkg, cfm, cfo.data());
                     dcfo = hypre_TAlloc(HYPRE_Int, (size_t) nco, HYPRE_MEMORY_DEVICE);
                     hypre_TMemcpy(dcfo, cfo.data(), HYPRE_Int, (size_t) nco, HYPRE_MEMORY_DEVICE, HYPRE_MEMORY_HOST);
                  }
                  fine = device_l1_norms_blocks(n, M->diag->i, M->diag->j, M->diag->data, M->offd->i, M->offd->j, M->offd->data, option, dcf,
                                                dcfo, l1, stream());
                  if (dcfo) { hypre_Free(dcfo, HYPRE_MEMORY_DEVICE); }
               }
               if (!fine) { hypre_error_in_arg(1); }
               return;
            }
            fetch_level(j);
            M = hostA[(size_t) j];
            l1_on_device = false;
            if (gs_threads > 1) { hypre_ParCSRComputeL1NormsThreads(M, option, gs_threads, cfm, &l1); }
            else { hypre_ParCSRComputeL1Norms(M, option, cfm, &l1); }
         };
         if (!last && any(8, 89, 13, 14)) { l1_norms(4, cf); }
         else if (last && last_is(8, 89, 13, 14)) { l1_norms(4, nullptr); }
         if (!last && (gt[1] == 88 || gt[2] == 88)) { l1_norms(6, cf); }
         else if (last && gt[3] == 88) { l1_norms(6, nullptr); }
         if (!last && (gt[1] == 18 || gt[2] == 18)) { l1_norms(1, cf); }
         else if (last && gt[3] == 18) { l1_norms(1, nullptr); }
         auto diag_smoother = [](HYPRE_Int t) { return t == 7 || t == 11 || t == 12 || t == 21 || t == 22; };
         if (diag_smoother(gt[1]) || diag_smoother(gt[2]) || (diag_smoother(gt[3]) && last)) { l1_norms(5, nullptr); }
         if (l1)
         {
            const HYPRE_MemoryLocation where = l1_on_device ? HYPRE_MEMORY_DEVICE : HYPRE_MEMORY_HOST;
            d->l1_norms[j] = hypre_SeqVectorCreate(hostA[(size_t) j]->diag->num_rows);
            d->l1_norms[j]->data = l1;
            d->l1_norms[j]->memory_location = where;
            hypre_SeqVectorInitialize_v2(d->l1_norms[j], where);
         }
      }
   }

   // Chebyshev smoothing (par_amg_setup.c:3273-3285, 3510-3550): spectrum estimate, polynomial
   // coefficients and the scaling vector of every level that is smoothed with relax 16
   const bool uses_cheby = d->grid_relax_type[0] == 16 || d->grid_relax_type[1] == 16 ||
                           d->grid_relax_type[2] == 16 || d->grid_relax_type[3] == 16;
   if (uses_cheby)
   {
      const HYPRE_Int *gt = d->grid_relax_type;
      d->max_eig_est = (HYPRE_Real *) calloc((size_t) num_levels, sizeof(HYPRE_Real));
      d->min_eig_est = (HYPRE_Real *) calloc((size_t) num_levels, sizeof(HYPRE_Real));
      d->cheby_ds = (hypre_Vector **) calloc((size_t) num_levels, sizeof(void *));
      d->cheby_coefs = (HYPRE_Real **) calloc((size_t) num_levels, sizeof(void *));
      for (int j = 0; j < num_levels; j++)
      {
         const bool last = (j == num_levels - 1);
         if (!(gt[1] == 16 || gt[2] == 16 || (gt[3] == 16 && last))) { continue; }
         fetch_level(j);
         hypre_ParCSRMatrix *Al = hostA[(size_t) j];
         HYPRE_Real max_eig = 0.0, min_eig = 0.0, *coefs = nullptr, *ds = nullptr;
         if (d->cheby_eig_est) { hypre_ParCSRMaxEigEstimateCG(Al, d->cheby_scale, d->cheby_eig_est, &max_eig, &min_eig); }
         else { hypre_ParCSRMaxEigEstimate(Al, d->cheby_scale, &max_eig, &min_eig); }
         d->max_eig_est[j] = max_eig;
         d->min_eig_est[j] = min_eig;
         hypre_ParCSRRelax_Cheby_Setup(Al, max_eig, min_eig, d->cheby_fraction, d->cheby_order, d->cheby_scale,
                                       d->cheby_variant, &coefs, &ds);
         d->cheby_coefs[j] = coefs;
         if (ds)
         {
            d->cheby_ds[j] = hypre_SeqVectorCreate(Al->diag->num_rows);
            d->cheby_ds[j]->data = ds;
            d->cheby_ds[j]->memory_location = HYPRE_MEMORY_HOST;
            hypre_SeqVectorInitialize_v2(d->cheby_ds[j], HYPRE_MEMORY_HOST);
         }
      }
   }

   // work vectors sized for the finest level (par_amg_setup.c:846-880: Chebyshev needs two more)
   d->Vtemp = new_vec(comm, A->global_num_rows, A->row_starts, target);
   d->Ztemp = new_vec(comm, A->global_num_rows, A->row_starts, target);
   d->Rtemp = nullptr; d->Ptemp = nullptr;
   if (uses_cheby)
   {
      d->Ptemp = new_vec(comm, A->global_num_rows, A->row_starts, target);
      d->Rtemp = new_vec(comm, A->global_num_rows, A->row_starts, target);
   }

   // algorithmic bytes of one V(1,1) cycle on this hierarchy: per level below the
   // coarsest 2 passes over A plus one over P and one over P^T
   {
      double bytes = 0.0;
      for (int l = 0; l < num_levels - 1; l++)
      {
         hypre_ParCSRMatrix *Al = hostA[(size_t) l], *Pl = d->P_array[l];
         const double nA = (double) Al->diag->num_nonzeros + Al->offd->num_nonzeros, rA = Al->diag->num_rows;
         const double nP = (double) Pl->diag->num_nonzeros + Pl->offd->num_nonzeros, cP = Pl->diag->num_cols;
         const double sA = nA * 12 + (rA + 1) * 4 + rA * 8 + rA * 8;
         bytes += 2.0 * sA;                                   // residual + post-smoothing pass
         bytes += 3.0 * rA * 8;                               // f read (x2) + l1 read of the fused sweep / residual
         bytes += 3.0 * rA * 8;                               // zero-guess pre-smoothing: f, l1 read, u write
         bytes += nP * 12 + (rA + 1) * 4 + cP * 8 + 2 * rA * 8;   // prolongation u += P e
         bytes += nP * 12 + (cP + 1) * 4 + rA * 8 + cP * 8;       // restriction f_c = P^T r
      }
      pv->cycle_bytes = bytes;
   }

   // multi-rank device runs: small levels are replicated on every rank (par_amg_replicate.cpp)
   if (target == HYPRE_MEMORY_DEVICE) { build_replicated_tail(d, hostA); }

   // place the hierarchy where the solve phase will run
   if (target == HYPRE_MEMORY_DEVICE)
   {
      for (int l = 0; l < num_levels; l++)
      {
         auto place = [&](hypre_ParCSRMatrix *M)
         {
            place_on_device(M->diag);
            place_on_device(M->offd);
            if (M->diagT) { place_on_device(M->diagT); }
            if (M->offdT) { hypre_CSRMatrixMigrate(M->offdT, HYPRE_MEMORY_DEVICE); }
         };
         if (l > 0) { place(d->A_array[l]); }
         if (l < num_levels - 1)
         {
            hypre_amd_ParCSRMatrixKeepTranspose(d->P_array[l]);
            place(d->P_array[l]);
         }
         if (l > 0)
         {
            hypre_ParVectorMigrate(d->F_array[l], HYPRE_MEMORY_DEVICE);
            hypre_ParVectorMigrate(d->U_array[l], HYPRE_MEMORY_DEVICE);
         }
         if (d->l1_norms[l]) { hypre_SeqVectorMigrate(d->l1_norms[l], HYPRE_MEMORY_DEVICE); }
         if (d->cheby_ds && d->cheby_ds[l]) { hypre_SeqVectorMigrate(d->cheby_ds[l], HYPRE_MEMORY_DEVICE); }
         if (d->CF_marker_array[l])
         {
            hypre_IntArray *a = d->CF_marker_array[l];
            HYPRE_Int *dd = device_marker_of(a->data, a->size);       // the copy the device coarsening left, or a fresh one
            device_markers().erase(a->data);
            hypre_Free(a->data, HYPRE_MEMORY_HOST);
            a->data = dd; a->memory_location = HYPRE_MEMORY_DEVICE;
         }
      }
      // the kernels' per-matrix plans (tile bounds, placement tables, x-staging descriptors and local indices) are part of
      // the setup: built here, not inside the first cycle
      // what the setup made cannot change behind its plans (SpmvPlan::owned): every level below the finest, every
      // interpolation operator and its stored transpose; the finest level stays the caller's
      for (int l = 0; l < num_levels; l++)
      {
         auto own = [](hypre_ParCSRMatrix *M)
         {
            if (!M) { return; }
            for (hypre_CSRMatrix *B : {M->diag, M->offd, M->diagT, M->offdT}) { if (B && B->memory_location == HYPRE_MEMORY_DEVICE) { mark_owned(B); } }
         };
         if (l > 0) { own(d->A_array[l]); }
         if (l < num_levels - 1) { own(d->P_array[l]); }
      }
      for (int l = 0; l < num_levels; l++)
      {
         auto plan_of = [](hypre_CSRMatrix *M) { if (M && M->memory_location == HYPRE_MEMORY_DEVICE && M->num_nonzeros > 0) { (void) get_plan(M); } };
         plan_of(d->A_array[l]->diag);
         if (l < num_levels - 1) { plan_of(d->P_array[l]->diag); plan_of(d->P_array[l]->diagT); }
         // two-stage Gauss-Seidel sweeps multiply by the strictly lower copy of the operator: made here as well
         const HYPRE_Int *gt = d->grid_relax_type;
         const bool last = (l == num_levels - 1);
         const bool two_stage = last ? (gt[3] == 11 || gt[3] == 12) : (gt[1] == 11 || gt[1] == 12 || gt[2] == 11 || gt[2] == 12);
         hypre_CSRMatrix *Ad = d->A_array[l]->diag;
         if (two_stage && single_rank && Ad->memory_location == HYPRE_MEMORY_DEVICE && Ad->num_nonzeros > 0) { plan_of(strict_lower_of(Ad)); }
         // multicolour Gauss-Seidel: the colouring and the colour classes (device kernels) belong to the setup as well
         auto mc = [](HYPRE_Int t) { return t == 21 || t == 22; };
         if (last ? mc(gt[3]) : (mc(gt[1]) || mc(gt[2]))) { prepare_mc_plan(Ad); }
      }
      hypre_ParVectorMigrate(d->Vtemp, HYPRE_MEMORY_DEVICE);
      hypre_ParVectorMigrate(d->Ztemp, HYPRE_MEMORY_DEVICE);
      if (d->Ptemp) { hypre_ParVectorMigrate(d->Ptemp, HYPRE_MEMORY_DEVICE); }
      if (d->Rtemp) { hypre_ParVectorMigrate(d->Rtemp, HYPRE_MEMORY_DEVICE); }
   }
   else
   {
      for (int l = 0; l < num_levels - 1; l++) { hypre_amd_ParCSRMatrixKeepTranspose(d->P_array[l]); }
   }
   if (own_host_A0) { hypre_ParCSRMatrixDestroy(hostA[0]); }
   return hypre_error_flag;
}

HYPRE_Real hypre_amd_BoomerAMGCycleBytes(HYPRE_Solver s)
{
   hypre_ParAMGData *d = (hypre_ParAMGData *) s;
   return d ? ((AmgPrivate *) d->amd_private)->cycle_bytes : 0.0;
}

}  // extern "C"
