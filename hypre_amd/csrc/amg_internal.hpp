// hypre_amd — internal declarations of the BoomerAMG translation units.
#pragma once
#include "internal.hpp"
#include "hypre_amd_parcsr_ls.h"
#include <utility>
#include <vector>

namespace hamd {

// The smallest levels of a V(1,1) Jacobi cycle in one kernel of one workgroup (tail_kernels.hip)
constexpr int SMALL_TAIL_MAX_LEVELS = 10;
constexpr int SMALL_TAIL_MAX_COARSE = 32;      // unknowns of the coarsest level's dense solve
struct SmallTailLevel
{
   // byte offsets into the kernel's LDS (the image's arrays first, the work vectors behind them)
   int    Ai, Aj, Aa;                          // the level's operator
   int    Pi, Pj, Pa;                          // interpolation to this level from the next
   int    Ri, Rj, Ra;                          // restriction from this level to the next: the stored transpose of P
   int    d;                                   // smoother diagonal
   int    f, u, alt;                           // right-hand side, iterate, second buffer of the out-of-place sweep
   int    n;                                   // rows
   int    wA, wP, wR;                          // lanes per row of the passes over A, P, R (powers of two <= 64)
   double w;                                   // relaxation weight
   const int    *gAj;                          // the operator's columns and values where they are, when the image leaves them
   const double *gAa;                          // out (the first level of a tail that would not fit LDS otherwise); else null
};
struct SmallTailArgs
{
   SmallTailLevel lv[SMALL_TAIL_MAX_LEVELS];
   int            nl;                          // levels (the last one is solved directly)
   const void    *image;                       // the levels' arrays as the kernel lays them out in LDS (AmgPrivate::tail_image)
   int            image_bytes;                 // a multiple of 16
   int            lds_bytes;                   // image + work vectors
   int            lu_off, ncoarse;             // factors of the coarsest operator
   int            vt_off;                      // residual scratch, as long as the first level
   const double  *f_in;                        // right-hand side of the first level
   double        *u_io;                        // its iterate: read when the restriction into it wrote the sweep from zero; the result
   int            first_presmoothed;
   int            round32;                     // mixed precision: matrix values rounded through fp32
   int            kind_down, kind_up;          // smoother on the way down / up: 0 Jacobi (relax 7 / 18), 1 two-stage Gauss-Seidel (11 / 12)
   int            inner_down, inner_up;        // its inner steps (relax 11: 1, relax 12: 2)
   int            reg_first;                   // the first level's operator is held in registers (lv[0].gAj / gAa, gAi: where it lies)
   const int     *gAi;
   int            nnz0;
};
void launch_small_tail(const SmallTailArgs &t, hipStream_t s);

// Device-side state that rides along with a hypre_ParAMGData (amd_private).
struct AmgPrivate
{
   int  emulated_threads = 1;      // thread count the thread-partitioned host loops emulate
   bool mixed_precision  = false;
   bool filter_functions = false;  // systems: build the hierarchy from A without its inter-function couplings
   // per-level overrides of the smoother weights (HYPRE_BoomerAMGSetLevelRelaxWt / SetLevelOuterWt),
   // applied over the uniform values when setup fills relax_weight[] / omega[]
   std::vector<std::pair<int, double>> level_relax_wt, level_outer_wt;

   // second solution buffer per level: Jacobi-type sweeps are out-of-place
   // (u_new is written while u_old is gathered), so every level ping-pongs
   // between its home vector and this one instead of copying back.
   std::vector<double *> u_alt;     // [num_levels], device
   std::vector<int>      u_alt_len;
   // diagonal of a level's operator for the sweeps that have no smoother-diagonal vector (relax 0, 17, Jacobi without l1):
   // one buffer per level, owned by this solver, so that a recorded coarse-tail graph never points into scratch that
   // another solver's larger level may reallocate
   std::vector<double *> diag_buf;  // [num_levels], device, allocated at the first cycle that needs it
   std::vector<int>      diag_len;
   double *level_diag(int level, int n);

   // mixed precision: fp64 residual and correction of the outer solve loop (a cycle on fp32-rounded operators applied
   // to a non-zero iterate must work on the error equation, or the iteration converges to the rounded system's solution)
   hypre_ParVector *mp_r = nullptr, *mp_e = nullptr;

   // Coarse tail of a single-rank V-cycle as one HIP graph.  From level graph_level down and back up every kernel is a
   // few microseconds of work behind a launch that costs as much; the sub-cycle reads F[graph_level], writes U[graph_level]
   // and touches only buffers the hierarchy owns, so its launches are recorded once (on the second cycle: the first one
   // creates plans and scratch, which a capture cannot) and replayed as one graph.  A signature of everything the recorded
   // launches depend on (options, weights, operator and vector addresses) is checked at every cycle; a mismatch drops
   // the graph and records again.
   // levels with at most this many rows belong to the tail (0: no graph — the default since the end of round 4: with the
   // smallest levels in one kernel the recorded tail holds 13 - 25 launches, the host runs ahead of them anyway, and a graph
   // costs ~8 us where it hands back to the eager stream: eager cycles measured 0.5 - 1.4 % faster on every configuration;
   // HYPRE_AMD_CYCLE_GRAPH_ROWS / hypre_amd_BoomerAMGSetGraphThreshold switch it on)
   int            graph_rows   = [] { const char *e = getenv("HYPRE_AMD_CYCLE_GRAPH_ROWS"); return e ? atoi(e) : 0; }();
   int            graph_level  = -1;          // first level of the tail (>= 1), fixed when the first cycle runs
   int            graph_state  = 0;           // 0: nothing yet, 1: warmed up (record next), 2: graph ready
   unsigned long long graph_sig = 0;
   hipGraph_t     graph        = nullptr;
   hipGraphExec_t graph_exec   = nullptr;
   std::vector<double *> graph_cur;           // where every tail level's iterate lives when the sub-cycle is over
   double         graph_op_count = 0.0;       // what the sub-cycle adds to cycle_op_count
   double         graph_bytes_csr = 0.0, graph_bytes_stream = 0.0;   // ... and to the byte counters (Handle::bytes_*)
   int            graph_launches = 0;         // kernel nodes of the graph (reported by the benchmark)
   void drop_graph();

   // one-workgroup tail (tail_kernels.hip): first level of it, or -1 (decided at the first cycle; levels whose operator
   // holds at most small_tail_nnz entries; 0: off)
   int            small_tail_nnz   = [] { const char *e = getenv("HYPRE_AMD_SMALL_TAIL_NNZ"); return e ? atoi(e) : 20000; }();
   int            small_tail_level = -2;      // -2: not decided yet
   int            small_tail_used  = -2;      // level the last cycle entered the tail at (-1: it did not; tests)
   bool           replica          = false;   // this hierarchy is a replicated tail: its two-stage Gauss-Seidel sweeps keep the
                                              // ranks' own lower triangles (par_amg_replicate.cpp), which the one-workgroup tail does not
   void          *tail_image       = nullptr; // device: the tail levels' arrays in the kernel's LDS layout
   unsigned long long tail_image_sig = 0;     // what the image was built from
   int            tail_outside     = -1;      // form of the image (0 all in LDS, 1 first operator in registers, 2 streamed); -1: not known
   SmallTailArgs  tail_args{};                // offsets of the image's arrays (vectors and flags are filled per cycle)

   // relax 15: one unpreconditioned-CG solver per level, created at the first cycle that needs it
   std::vector<HYPRE_Solver> cg_smoothers;

   // coarsest level: factors of the reference's pivot-free elimination
   // (utilities/gselim.h), computed once on the host, applied on the device
   double *d_coarse_lu = nullptr;   // n*n row-major: U on/above the diagonal, multipliers below
   double *d_coarse_rhs = nullptr;  // n (gathered right-hand side)
   int     coarse_n = 0;
   int     coarse_first_row = 0;    // this rank's offset into the gathered system

   // algorithmic byte count of one cycle (filled by setup, SURVEY §8d formula)
   double cycle_bytes = 0.0;

   // Replicated tail (multi-rank, device): the levels from tail_level down are small enough that
   // their halo exchanges are pure latency.  Setup gathers these operators onto every rank as a
   // single-rank hierarchy; the cycle then gathers the right-hand side of level tail_level with ONE
   // all-reduce, runs the rest of the V-cycle redundantly and locally, and keeps its own slice.
   // Global row count at or below which a level is replicated (0: off).  What a replicated level costs a rank grows with
   // the level's GLOBAL size (every rank sweeps all of it), what a distributed one costs is four latency-bound exchanges
   // (~30 us each over xGMI) whatever its size: with ~90 entries per row the two meet around 100 000 global rows (six
   // passes of 100 000 x 90 entries at ~5 TB/s = 130 us against 4 x 30 us of exchanges plus the local share), and below
   // ~25 000 rows a level is launch-bound either way.  At 8 x 256^3 that replicates from level 4 (66 320 rows): 16
   // exchanges + 1 all-reduce per cycle instead of 20; HYPRE_AMD_REPLICATE_ROWS / ..SetReplicateThreshold change it.
   int               replicate_rows = [] { const char *e = getenv("HYPRE_AMD_REPLICATE_ROWS"); return e ? atoi(e) : 100000; }();
   hypre_ParAMGData *tail = nullptr;
   int               tail_level = -1;
   double           *d_tail_f = nullptr;      // global right-hand side of level tail_level (device)

   void release_device();
   ~AmgPrivate() { release_device(); }
};

}  // namespace hamd

extern "C" void amg_free_hierarchy(hypre_ParAMGData *d);

namespace hamd {

// raw-pointer cores shared by the public entry points and the cycle
void dev_par_matvec(HYPRE_Complex alpha, hypre_ParCSRMatrix *A, const double *x, HYPRE_Complex beta,
                    const double *b, double *y);
void dev_par_matvecT(HYPRE_Complex alpha, hypre_ParCSRMatrix *A, const double *x, HYPRE_Complex beta, double *y);
// Jacobi-type sweep u_out = u_in + w*(f - A u_in)./d on marked rows, u_out = u_in elsewhere
void dev_jacobi_sweep(hypre_ParCSRMatrix *A, const double *f, const int *cf_marker, int relax_points,
                      double w, const double *d, const double *u_in, double *u_out);

hypre_ParCSRCommHandle *dev_halo_begin(hypre_ParCSRMatrix *A, const double *x_local);
void dev_halo_end(hypre_ParCSRCommHandle *h);

// replicated tail (par_amg_replicate.cpp)
void build_replicated_tail(hypre_ParAMGData *d, const std::vector<hypre_ParCSRMatrix *> &hostA);
void destroy_replicated_tail(hypre_ParAMGData *d);

// device twins and markers of the setup in progress (par_amg_setup.cpp)
hypre_CSRMatrix *setup_device_twin_of(hypre_CSRMatrix *host, int with_data);
hypre_CSRMatrix *setup_wrap_device_csr(HYPRE_Int nr, HYPRE_Int ncl, HYPRE_Int nnz, int *i, int *j, double *a);
void setup_register_device_marker(const HYPRE_Int *host, HYPRE_Int *dev);     // the table takes ownership of dev
HYPRE_Int *setup_device_marker_of(const HYPRE_Int *host, HYPRE_Int n);

// distributed levels of a device-targeted setup (par_amg_setup_dist.cpp): the single-rank device kernels on the extended
// numbering.  *_ptr == nullptr with a clear error flag: some rank's tables overflowed — every rank comes back that way
// (the verdict is agreed on), and the caller repeats the step with the host routine.
HYPRE_Int dist_device_create_S(hypre_ParCSRMatrix *A, HYPRE_Real theta, HYPRE_Real max_row_sum, hypre_ParCSRMatrix **S_ptr);
HYPRE_Int dist_device_pmis(hypre_ParCSRMatrix *S, hypre_ParCSRMatrix *A, HYPRE_Int CF_init, HYPRE_Int *CF_host);
HYPRE_Int dist_device_extpi_interp(hypre_ParCSRMatrix *A, HYPRE_Int *CF_marker, hypre_ParCSRMatrix *S,
                                   HYPRE_BigInt *num_cpts_global, HYPRE_BigInt total_global_cpts, HYPRE_Real trunc_factor,
                                   HYPRE_Int max_elmts, HYPRE_Int first_rung, hypre_ParCSRMatrix **P_ptr);
HYPRE_Int dist_device_coarse_operator(hypre_ParCSRMatrix *RT, hypre_ParCSRMatrix *A, hypre_ParCSRMatrix *P,
                                      HYPRE_Int keepTranspose, hypre_ParCSRMatrix **RAP_ptr);
// true on every rank or on none
bool all_ranks_agree(MPI_Comm comm, bool mine);

// distributed setup pieces (par_amg_setup_dist.cpp)
HYPRE_Int dist_build_extpi_interp(hypre_ParCSRMatrix *A, HYPRE_Int *CF_marker, hypre_ParCSRMatrix *S,
                                  HYPRE_BigInt *num_cpts_global, HYPRE_BigInt total_global_cpts,
                                  const HYPRE_Int *dof_func,   // function of every local row, or nullptr (scalar problem)
                                  HYPRE_Real trunc_factor, HYPRE_Int max_elmts, hypre_ParCSRMatrix **P_ptr);
HYPRE_Int dist_build_coarse_operator(hypre_ParCSRMatrix *RT, hypre_ParCSRMatrix *A, hypre_ParCSRMatrix *P,
                                     HYPRE_Int keepTranspose, hypre_ParCSRMatrix **RAP_ptr);

// operands of one directional hybrid Gauss-Seidel / SOR sweep (gs_kernels.hip)
struct GsArgs
{
   const HYPRE_Int *Di, *Dj; const HYPRE_Complex *Da;     // diag block
   const HYPRE_Int *Oi, *Oj; const HYPRE_Complex *Oa;     // offd block (Oi == nullptr: none)
   const double *f;
   const int    *cf; int relax_points;
   const double *l1;           // smoother diagonal, or nullptr for the stored diagonal
   double       *u;
   const double *uold;         // u when this directional sweep started
   const double *vtemp;        // u when the relaxation call started
   const double *vext;         // ghost values of that state
   int    dir;                 // +1 forward, -1 backward
   int    skip_diag, non_scale;
   double w, omega;
   int    n, threads;
   const int4 *sched;          // rows ordered by level: {row, first entry, end of row, -}
};
void launch_gs_level(const GsArgs &a, int lanes, int start, int count, hipStream_t s);
void launch_gs_run(const GsArgs &a, int lanes, const int *d_lev_start, int lev_begin, int lev_end, hipStream_t s);
void drop_gs_schedule(const hypre_CSRMatrix *A);

// fused elementwise passes of the Chebyshev smoother (cheby_kernels.hip)
void launch_cheby_start(const double *f, const double *t, const double *ds, double c, bool last, double *u,
                        double *orig, double *r, double *tmp, size_t n, hipStream_t s);
void launch_cheby_step(const double *r, const double *v, const double *ds, const double *orig, double mult, bool last,
                       double *u, double *tmp, size_t n, hipStream_t s);

void launch_jacobi_update(const double *u_in, const double *r, const double *d, const int *marker, int mval,
                          double *u_out, size_t n, hipStream_t s);
void launch_diag_first(const int *Ai, const double *Aa, double *d, int n, hipStream_t s);
void launch_coarse_solve(const double *lu, double *x, int n, hipStream_t s);

}  // namespace hamd
