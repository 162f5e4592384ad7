/*
 * hypre_amd — BoomerAMG: relaxation, V-cycle, solve (the hot path), plus the
 * host-side setup that produces the hierarchy the cycle runs on, and the PCG
 * caller.
 *
 * Functions replace (paths relative to /root/reference/src):
 *   parcsr_ls/par_relax.c:24-173               hypre_BoomerAMGRelax (dispatcher)
 *   parcsr_ls/par_relax.c:1178-1254,180-369    Jacobi / l1-Jacobi (relax 7, 18, 0)
 *   parcsr_ls/par_relax.c:1506-1666, par_relax_device.c:97-155   two-stage GS (11, 12)
 *   parcsr_ls/par_relax.c:691-1377, par_relax_device.c:19-90     hybrid GS family (3,4,6,8,13,14,88,89)
 *   parcsr_ls/par_relax_interface.c:20-117     hypre_BoomerAMGRelaxIF, L1_Jacobi, FCFJacobi
 *   parcsr_ls/par_cycle.c:23-803               hypre_BoomerAMGCycle
 *   parcsr_ls/par_amg_solve.c:22-424           hypre_BoomerAMGSolve
 *   parcsr_ls/HYPRE_parcsr_amg.c:64-91         HYPRE_BoomerAMGSolve
 *   parcsr_ls/par_gauss_elim.c:457-697         hypre_GaussElimSolve (coarsest level)
 *   parcsr_ls/ams.c:527-830                    hypre_ParCSRComputeL1Norms
 *   parcsr_ls/par_amg.c / HYPRE_parcsr_amg.c   Create / Destroy / Set* / Get*
 *   parcsr_ls/par_amg_setup.c:28-4046          hypre_BoomerAMGSetup (host; "next" row of the scope table)
 *   krylov/pcg.c:318-1000 + parcsr_ls/HYPRE_parcsr_pcg.c   PCG with a BoomerAMG preconditioner
 *
 * hypre_ParAMGData below is the solve- and setup-relevant SLICE of the
 * reference struct (parcsr_ls/par_amg.h:19-295): member names and the
 * accessor macros match, the byte layout does not (the reference struct
 * carries ~250 members for solvers outside this path).  Code that touches
 * the solver object only through the HYPRE_BoomerAMG* calls or the
 * hypre_ParAMGData* macros is source compatible; see INTEGRATION.md.
 */
#ifndef HYPRE_AMD_PARCSR_LS_H
#define HYPRE_AMD_PARCSR_LS_H

#include "hypre_amd_parcsr_mv.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct
{
   hypre_Solver          base;               /* setup / solve / destroy */
   HYPRE_MemoryLocation  memory_location;    /* where the hierarchy lives after setup */

   /* setup parameters (par_amg.c:162-316 defaults) */
   HYPRE_Int      max_levels;
   HYPRE_Real     strong_threshold;
   HYPRE_Real     max_row_sum;
   HYPRE_Real     trunc_factor;
   HYPRE_Int      measure_type;
   HYPRE_Int      coarsen_type;              /* 8 PMIS, 9 PMIS(seq rand), 10 HMIS */
   HYPRE_Int      P_max_elmts;
   HYPRE_Int      interp_type;               /* 6 ext+i, 3 direct */
   HYPRE_Int      agg_num_levels;
   HYPRE_Int      max_coarse_size;
   HYPRE_Int      min_coarse_size;
   HYPRE_Int      keepTranspose;
   HYPRE_Int      num_functions;

   /* solve parameters */
   HYPRE_Int      max_iter;
   HYPRE_Int      min_iter;
   HYPRE_Int      fcycle;
   HYPRE_Int      cycle_type;
   HYPRE_Int     *num_grid_sweeps;           /* [4] */
   HYPRE_Int     *grid_relax_type;           /* [4] */
   HYPRE_Int    **grid_relax_points;         /* NULL unless set by the user */
   HYPRE_Int      relax_order;
   HYPRE_Int      user_coarse_relax_type;
   HYPRE_Int      user_relax_type;
   HYPRE_Int      user_num_sweeps;
   HYPRE_Real     user_relax_weight;
   HYPRE_Real     outer_wt;
   HYPRE_Real    *relax_weight;              /* [max_levels] */
   HYPRE_Real    *omega;                     /* [max_levels] */
   HYPRE_Int      converge_type;
   HYPRE_Real     tol;

   /* hierarchy */
   hypre_ParCSRMatrix  *A;
   hypre_ParCSRMatrix **A_array;
   hypre_ParVector    **F_array;
   hypre_ParVector    **U_array;
   hypre_ParCSRMatrix **P_array;
   hypre_ParCSRMatrix **R_array;             /* == P_array (restriction is P^T) */
   hypre_IntArray     **CF_marker_array;
   HYPRE_Int            num_levels;
   hypre_Vector       **l1_norms;

   /* work vectors (fine-grid sized, re-sized per level in place) */
   hypre_ParVector   *Vtemp;
   hypre_ParVector   *Rtemp;
   hypre_ParVector   *Ptemp;
   hypre_ParVector   *Ztemp;
   HYPRE_Real         cycle_op_count;

   /* coarsest-level dense solve (par_gauss_elim.c) */
   HYPRE_Int          gs_setup;
   HYPRE_Real        *A_mat;                 /* dense coarse operator, row major, host */
   HYPRE_Real        *b_vec;

   /* log */
   HYPRE_Int        logging;
   HYPRE_Int        num_iterations;
   HYPRE_Real       rel_resid_norm;
   HYPRE_Int        print_level;
   HYPRE_Int        debug_flag;

   /* Chebyshev smoothing, relax 16 (par_amg.h:208-217; defaults par_amg.c:273-277) */
   HYPRE_Int        cheby_order;             /* 1..4, default 2 */
   HYPRE_Int        cheby_eig_est;           /* CG iterations of the estimate, 0 = Gershgorin; default 10 */
   HYPRE_Int        cheby_variant;           /* 0 standard, 1 modified */
   HYPRE_Int        cheby_scale;             /* 1: smooth D^-1/2 A D^-1/2 */
   HYPRE_Real       cheby_fraction;          /* part of the spectrum that is damped, default 0.3 */
   HYPRE_Real      *max_eig_est;             /* [num_levels] */
   HYPRE_Real      *min_eig_est;
   hypre_Vector   **cheby_ds;                /* [num_levels] 1/sqrt(|a_ii|) */
   HYPRE_Real     **cheby_coefs;             /* [num_levels][order + 1], host */

   /* library-private state (device plans, graphs, mixed-precision copies) */
   void            *amd_private;
} hypre_ParAMGData;

#define hypre_ParAMGDataMemoryLocation(d)   ((d)->memory_location)
#define hypre_ParAMGDataMaxLevels(d)        ((d)->max_levels)
#define hypre_ParAMGDataStrongThreshold(d)  ((d)->strong_threshold)
#define hypre_ParAMGDataMaxRowSum(d)        ((d)->max_row_sum)
#define hypre_ParAMGDataTruncFactor(d)      ((d)->trunc_factor)
#define hypre_ParAMGDataCoarsenType(d)      ((d)->coarsen_type)
#define hypre_ParAMGDataPMaxElmts(d)        ((d)->P_max_elmts)
#define hypre_ParAMGDataInterpType(d)       ((d)->interp_type)
#define hypre_ParAMGDataMaxCoarseSize(d)    ((d)->max_coarse_size)
#define hypre_ParAMGDataMinCoarseSize(d)    ((d)->min_coarse_size)
#define hypre_ParAMGDataKeepTranspose(d)    ((d)->keepTranspose)
#define hypre_ParAMGDataMaxIter(d)          ((d)->max_iter)
#define hypre_ParAMGDataMinIter(d)          ((d)->min_iter)
#define hypre_ParAMGDataFCycle(d)           ((d)->fcycle)
#define hypre_ParAMGDataCycleType(d)        ((d)->cycle_type)
#define hypre_ParAMGDataNumGridSweeps(d)    ((d)->num_grid_sweeps)
#define hypre_ParAMGDataGridRelaxType(d)    ((d)->grid_relax_type)
#define hypre_ParAMGDataGridRelaxPoints(d)  ((d)->grid_relax_points)
#define hypre_ParAMGDataRelaxOrder(d)       ((d)->relax_order)
#define hypre_ParAMGDataUserRelaxType(d)    ((d)->user_relax_type)
#define hypre_ParAMGDataRelaxWeight(d)      ((d)->relax_weight)
#define hypre_ParAMGDataOmega(d)            ((d)->omega)
#define hypre_ParAMGDataConvergeType(d)     ((d)->converge_type)
#define hypre_ParAMGDataTol(d)              ((d)->tol)
#define hypre_ParAMGDataAArray(d)           ((d)->A_array)
#define hypre_ParAMGDataFArray(d)           ((d)->F_array)
#define hypre_ParAMGDataUArray(d)           ((d)->U_array)
#define hypre_ParAMGDataPArray(d)           ((d)->P_array)
#define hypre_ParAMGDataRArray(d)           ((d)->R_array)
#define hypre_ParAMGDataCFMarkerArray(d)    ((d)->CF_marker_array)
#define hypre_ParAMGDataNumLevels(d)        ((d)->num_levels)
#define hypre_ParAMGDataL1Norms(d)          ((d)->l1_norms)
#define hypre_ParAMGDataVtemp(d)            ((d)->Vtemp)
#define hypre_ParAMGDataRtemp(d)            ((d)->Rtemp)
#define hypre_ParAMGDataPtemp(d)            ((d)->Ptemp)
#define hypre_ParAMGDataZtemp(d)            ((d)->Ztemp)
#define hypre_ParAMGDataCycleOpCount(d)     ((d)->cycle_op_count)
#define hypre_ParAMGDataNumIterations(d)    ((d)->num_iterations)
#define hypre_ParAMGDataRelativeResidualNorm(d) ((d)->rel_resid_norm)
#define hypre_ParAMGDataPrintLevel(d)       ((d)->print_level)
#define hypre_ParAMGDataLogging(d)          ((d)->logging)
#define hypre_ParAMGDataChebyOrder(d)       ((d)->cheby_order)
#define hypre_ParAMGDataChebyEigEst(d)      ((d)->cheby_eig_est)
#define hypre_ParAMGDataChebyVariant(d)     ((d)->cheby_variant)
#define hypre_ParAMGDataChebyScale(d)       ((d)->cheby_scale)
#define hypre_ParAMGDataChebyFraction(d)    ((d)->cheby_fraction)
#define hypre_ParAMGDataMaxEigEst(d)        ((d)->max_eig_est)
#define hypre_ParAMGDataMinEigEst(d)        ((d)->min_eig_est)
#define hypre_ParAMGDataChebyDS(d)          ((d)->cheby_ds)
#define hypre_ParAMGDataChebyCoefs(d)       ((d)->cheby_coefs)

/* ---- life cycle and parameters (HYPRE_parcsr_amg.c) ---- */
HYPRE_Int HYPRE_BoomerAMGCreate(HYPRE_Solver *solver);
HYPRE_Int HYPRE_BoomerAMGDestroy(HYPRE_Solver solver);
HYPRE_Int HYPRE_BoomerAMGSetup(HYPRE_Solver solver, HYPRE_ParCSRMatrix A, HYPRE_ParVector b, HYPRE_ParVector x);
HYPRE_Int HYPRE_BoomerAMGSolve(HYPRE_Solver solver, HYPRE_ParCSRMatrix A, HYPRE_ParVector b, HYPRE_ParVector x);
HYPRE_Int HYPRE_BoomerAMGSetMaxLevels(HYPRE_Solver solver, HYPRE_Int max_levels);
HYPRE_Int HYPRE_BoomerAMGSetMaxCoarseSize(HYPRE_Solver solver, HYPRE_Int max_coarse_size);
HYPRE_Int HYPRE_BoomerAMGSetMinCoarseSize(HYPRE_Solver solver, HYPRE_Int min_coarse_size);
HYPRE_Int HYPRE_BoomerAMGSetStrongThreshold(HYPRE_Solver solver, HYPRE_Real strong_threshold);
HYPRE_Int HYPRE_BoomerAMGSetMaxRowSum(HYPRE_Solver solver, HYPRE_Real max_row_sum);
HYPRE_Int HYPRE_BoomerAMGSetCoarsenType(HYPRE_Solver solver, HYPRE_Int coarsen_type);
HYPRE_Int HYPRE_BoomerAMGSetInterpType(HYPRE_Solver solver, HYPRE_Int interp_type);
HYPRE_Int HYPRE_BoomerAMGSetTruncFactor(HYPRE_Solver solver, HYPRE_Real trunc_factor);
HYPRE_Int HYPRE_BoomerAMGSetPMaxElmts(HYPRE_Solver solver, HYPRE_Int P_max_elmts);
HYPRE_Int HYPRE_BoomerAMGSetKeepTranspose(HYPRE_Solver solver, HYPRE_Int keepTranspose);
/* systems of PDEs, "unknown" approach (HYPRE_parcsr_amg.c HYPRE_BoomerAMGSetNumFunctions; par_strength.c:248-403,
 * par_lr_interp.c:1706-1713): row i belongs to function (global row) mod num_functions, couplings between
 * different functions are neither strong nor lumped.  ext+i interpolation only. */
HYPRE_Int HYPRE_BoomerAMGSetNumFunctions(HYPRE_Solver solver, HYPRE_Int num_functions);
/* par_amg.c:3232-3250, par_amg_setup.c:774-780, parcsr_mv/par_csr_filter.c:21-186: with num_functions > 1, build
 * the hierarchy from A without its inter-function couplings (`ij -ff 1`); level 0 is still smoothed with A */
HYPRE_Int HYPRE_BoomerAMGSetFilterFunctions(HYPRE_Solver solver, HYPRE_Int filter_functions);
/* Chebyshev smoother parameters (HYPRE_parcsr_amg.c:1340-1400 -> par_amg.c:4583-4670) */
HYPRE_Int HYPRE_BoomerAMGSetChebyOrder(HYPRE_Solver solver, HYPRE_Int order);
HYPRE_Int HYPRE_BoomerAMGSetChebyFraction(HYPRE_Solver solver, HYPRE_Real ratio);
HYPRE_Int HYPRE_BoomerAMGSetChebyEigEst(HYPRE_Solver solver, HYPRE_Int eig_est);
HYPRE_Int HYPRE_BoomerAMGSetChebyVariant(HYPRE_Solver solver, HYPRE_Int variant);
HYPRE_Int HYPRE_BoomerAMGSetChebyScale(HYPRE_Solver solver, HYPRE_Int scale);
HYPRE_Int HYPRE_BoomerAMGSetTol(HYPRE_Solver solver, HYPRE_Real tol);
HYPRE_Int HYPRE_BoomerAMGSetMaxIter(HYPRE_Solver solver, HYPRE_Int max_iter);
HYPRE_Int HYPRE_BoomerAMGSetMinIter(HYPRE_Solver solver, HYPRE_Int min_iter);
HYPRE_Int HYPRE_BoomerAMGSetConvergeType(HYPRE_Solver solver, HYPRE_Int type);
HYPRE_Int HYPRE_BoomerAMGSetCycleType(HYPRE_Solver solver, HYPRE_Int cycle_type);
HYPRE_Int HYPRE_BoomerAMGSetFCycle(HYPRE_Solver solver, HYPRE_Int fcycle);
HYPRE_Int HYPRE_BoomerAMGSetNumSweeps(HYPRE_Solver solver, HYPRE_Int num_sweeps);
HYPRE_Int HYPRE_BoomerAMGSetCycleNumSweeps(HYPRE_Solver solver, HYPRE_Int num_sweeps, HYPRE_Int k);
HYPRE_Int HYPRE_BoomerAMGSetRelaxType(HYPRE_Solver solver, HYPRE_Int relax_type);
HYPRE_Int HYPRE_BoomerAMGSetCycleRelaxType(HYPRE_Solver solver, HYPRE_Int relax_type, HYPRE_Int k);
HYPRE_Int HYPRE_BoomerAMGSetRelaxOrder(HYPRE_Solver solver, HYPRE_Int relax_order);
HYPRE_Int HYPRE_BoomerAMGSetRelaxWt(HYPRE_Solver solver, HYPRE_Real relax_weight);
HYPRE_Int HYPRE_BoomerAMGSetOuterWt(HYPRE_Solver solver, HYPRE_Real omega);
/* parcsr_ls/HYPRE_parcsr_amg.c:560-600 -> par_amg.c:2466,2590: weight of one level, overriding the uniform value */
HYPRE_Int HYPRE_BoomerAMGSetLevelRelaxWt(HYPRE_Solver solver, HYPRE_Real relax_weight, HYPRE_Int level);
HYPRE_Int HYPRE_BoomerAMGSetLevelOuterWt(HYPRE_Solver solver, HYPRE_Real omega, HYPRE_Int level);
HYPRE_Int HYPRE_BoomerAMGSetPrintLevel(HYPRE_Solver solver, HYPRE_Int print_level);
HYPRE_Int HYPRE_BoomerAMGSetLogging(HYPRE_Solver solver, HYPRE_Int logging);
/* Option gates (HYPRE_parcsr_amg.c): setters an application written against hypre calls routinely, for options whose
 * other branches are outside this library.  The value that selects what is implemented here is accepted (named in each
 * comment); any other value raises HYPRE_ERROR_ARG(2) with a message.  SetDebugFlag / SetNumPaths / SetSmoothType /
 * SetSmoothNumSweeps are accepted and inert. */
HYPRE_Int HYPRE_BoomerAMGSetMeasureType(HYPRE_Solver solver, HYPRE_Int measure_type);          /* 0 local, 1 global */
HYPRE_Int HYPRE_BoomerAMGSetDebugFlag(HYPRE_Solver solver, HYPRE_Int debug_flag);
HYPRE_Int HYPRE_BoomerAMGSetNumPaths(HYPRE_Solver solver, HYPRE_Int num_paths);
HYPRE_Int HYPRE_BoomerAMGSetAggNumLevels(HYPRE_Solver solver, HYPRE_Int agg_num_levels);       /* 0 */
HYPRE_Int HYPRE_BoomerAMGSetNodal(HYPRE_Solver solver, HYPRE_Int nodal);                       /* 0 */
HYPRE_Int HYPRE_BoomerAMGSetSeqThreshold(HYPRE_Solver solver, HYPRE_Int seq_threshold);        /* 0 */
HYPRE_Int HYPRE_BoomerAMGSetRedundant(HYPRE_Solver solver, HYPRE_Int redundant);               /* 0 */
HYPRE_Int HYPRE_BoomerAMGSetRAP2(HYPRE_Solver solver, HYPRE_Int rap2);                         /* 0 */
HYPRE_Int HYPRE_BoomerAMGSetRestriction(HYPRE_Solver solver, HYPRE_Int restr_par);             /* 0 */
HYPRE_Int HYPRE_BoomerAMGSetSmoothNumLevels(HYPRE_Solver solver, HYPRE_Int smooth_num_levels); /* 0 */
HYPRE_Int HYPRE_BoomerAMGSetSmoothType(HYPRE_Solver solver, HYPRE_Int smooth_type);
HYPRE_Int HYPRE_BoomerAMGSetSmoothNumSweeps(HYPRE_Solver solver, HYPRE_Int smooth_num_sweeps);
HYPRE_Int HYPRE_BoomerAMGSetAdditive(HYPRE_Solver solver, HYPRE_Int addlvl);                   /* -1 */
HYPRE_Int HYPRE_BoomerAMGSetMultAdditive(HYPRE_Solver solver, HYPRE_Int addlvl);               /* -1 */
HYPRE_Int HYPRE_BoomerAMGSetSimple(HYPRE_Solver solver, HYPRE_Int addlvl);                     /* -1 */
HYPRE_Int HYPRE_BoomerAMGSetNonGalerkinTol(HYPRE_Solver solver, HYPRE_Real nongalerkin_tol);   /* 0.0 */
HYPRE_Int HYPRE_BoomerAMGSetADropTol(HYPRE_Solver solver, HYPRE_Real A_drop_tol);              /* 0.0 */
HYPRE_Int HYPRE_BoomerAMGGetNumIterations(HYPRE_Solver solver, HYPRE_Int *num_iterations);
HYPRE_Int HYPRE_BoomerAMGGetFinalRelativeResidualNorm(HYPRE_Solver solver, HYPRE_Real *rel_resid_norm);

/* library extensions (no reference counterpart) */
/* memory location the hierarchy is placed in by Setup (default: device) */
HYPRE_Int hypre_amd_BoomerAMGSetMemoryLocation(HYPRE_Solver solver, HYPRE_MemoryLocation location);
/* OpenMP thread count the host setup emulates in its thread-partitioned loops
 * (0 = 1; results of the hybrid smoothers' block partition depend on it) */
HYPRE_Int hypre_amd_BoomerAMGSetNumThreads(HYPRE_Solver solver, HYPRE_Int num_threads);
/* Where the Galerkin products of HYPRE_BoomerAMGSetup are formed when the hierarchy's home is device memory and there is
 * one rank: on the device by default (rap_kernels.hip; same columns, order and bits as the host loop), on = 0 keeps the
 * host loop; min_rows: smallest fine level sent to the device (default 20000).  Negative arguments leave a setting
 * unchanged.  Returns the number of products formed on the device since the previous call.  The reference's device setup:
 * parcsr_mv/par_csr_triplemat.c:938-960. */
HYPRE_Int hypre_amd_SetSetupDeviceRAP(HYPRE_Int on, HYPRE_Int min_rows);
/* The same for the extended+i interpolation operators (interp_kernels.hip; scalar problems, levels of at least min_rows
 * rows): on = 0 keeps the host loop; on = 1 + k starts the kernel's ladder of table sizes (64, 128, 256, 1024 entries per
 * row; an overflowing row moves everyone up) at rung k — the results do not depend on it, tests walk the rungs; returns
 * the number built on the device since the previous call.  The reference's device routine:
 * parcsr_ls/par_lr_interp_device.c:1001. */
HYPRE_Int hypre_amd_SetSetupDeviceInterp(HYPRE_Int on);
/* And for strength of connection, PMIS coarsening and the smoother diagonals (setup_kernels.hip; the HOST routines'
 * measures and results, not the reference's device variant with its own random numbers): with all three switches on, a
 * single-rank scalar level of at least min_rows rows is set up without leaving the device — the coarse operator is not
 * fetched, and a matrix handed over in device memory is not copied to the host.  on = 0 keeps the host loops; returns the
 * number of levels coarsened on the device since the previous call.  The reference's device routines:
 * parcsr_ls/par_strength_device.c, parcsr_ls/par_coarsen_device.c:30. */
HYPRE_Int hypre_amd_SetSetupDeviceCoarsen(HYPRE_Int on);
/* Levels of a DISTRIBUTED hierarchy (several ranks, one per GPU) are set up on the device as well when the three switches
 * above are on and a level has at least min_rows rows per rank on average: the single-rank kernels run on the extended
 * numbering [local points | ghost points], fed with the rows the host routines exchange (A_ext, S_ext, P_ext, the product
 * rows computed for the neighbours, the PMIS halo rounds); results equal the host routines' array for array.  on = 0 keeps
 * distributed levels on the host (OpenMP loops); negative: unchanged.  Returns the previous setting.  The reference's device
 * routines: par_coarsen_device.c:30, par_lr_interp_device.c:1001, parcsr_mv/par_csr_triplemat.c:938-960. */
HYPRE_Int hypre_amd_SetSetupDeviceDist(HYPRE_Int on);
/* Test hook of the distributed device setup: rank `rank` pretends, `count` times, that the device kernel of step `what`
 * (1 interpolation, 2 product rows made for the neighbours, 3 Galerkin product) could not fit a row into its tables.  Every
 * rank then repeats that step with the host routine — the agreed fall-back, which real problems reach only through rows of
 * more than a thousand entries. */
HYPRE_Int hypre_amd_SetupDistTestDecline(HYPRE_Int what, HYPRE_Int rank, HYPRE_Int count);
/* The coarse tail of a single-rank V-cycle (levels of at most `rows` rows; 0: off) can be recorded once as a HIP graph and
 * replayed: its kernels are a few microseconds each behind launches that cost as much.  Off by default since the end of
 * round 4 (environment HYPRE_AMD_CYCLE_GRAPH_ROWS=100000 or this call switch it on): with the smallest levels in one kernel
 * (hypre_amd_SetSmallTail) the tail is 13 - 25 launches the host runs ahead of anyway, and eager cycles measured 0.5 - 1.4 %
 * faster than replayed ones on every benchmark configuration.  No reference counterpart (the
 * reference launches and synchronises per operation).  GetGraphInfo: first level of the recorded graph (-1: none) and its
 * node count. */
HYPRE_Int hypre_amd_BoomerAMGSetGraphThreshold(HYPRE_Solver solver, HYPRE_Int rows);
HYPRE_Int hypre_amd_BoomerAMGGetGraphInfo(HYPRE_Solver solver, HYPRE_Int *level, HYPRE_Int *nodes);
/* Mixed precision (BASELINE config C5): inside the cycle the SpMV-class kernels stream an fp32 copy of every level's matrix
 * values, diagonal and ghost blocks alike (4 + 2 instead of 8 + 2 bytes per entry); vectors, accumulation, smoother
 * diagonals, the coarse solve and everything outside the cycle stay fp64, and HYPRE_BoomerAMGSolve applies the cycle to the
 * fp64 residual equation from a non-zero iterate.  The reference has only the whole-library HYPRE_SINGLE
 * (utilities/HYPRE_utilities.h:78-89). */
HYPRE_Int hypre_amd_BoomerAMGSetMixedPrecision(HYPRE_Solver solver, HYPRE_Int on);
/* Fusions across the steps of a cycle (default on; environment HYPRE_AMD_CYCLE_FUSION=0): on one rank the restriction
 * f_c = P^T r also writes the result of the coarse level's first Jacobi-type sweep from zero, u_c = (w f_c) ./ d_c, instead
 * of a kernel of its own reading f_c and d_c again (par_cycle.c:340-420 runs the two as separate steps).  Same bits either
 * way; the switch is for comparisons.  on < 0 leaves the setting; returns it. */
HYPRE_Int hypre_amd_SetCycleFusion(HYPRE_Int on);
/* The smallest levels of a V(1,1) cycle with Jacobi / l1-Jacobi or two-stage Gauss-Seidel smoothing (relax 7 / 18 / 11 / 12,
 * no C/F ordering) and a direct
 * coarse solve in ONE kernel of one workgroup (default on; environment HYPRE_AMD_SMALL_TAIL=0): from the first level whose
 * operator — and every coarser one — holds at most 20 000 entries (HYPRE_AMD_SMALL_TAIL_NNZ) down and back up, every step
 * of par_cycle.c:23-803 is a launch of ~5 us for a fraction of a microsecond of work; one workgroup walks them with a
 * barrier in between.  One rank only.  Same arithmetic, the products of a row added in a fixed order of its own (rounding
 * differences against the per-level kernels).  on < 0 leaves the setting; returns it. */
HYPRE_Int hypre_amd_SetSmallTail(HYPRE_Int on);
/* Test hook: how the tail's kernel holds the first level's operator — 0 with everything else in LDS, 1 in the lanes'
 * registers (its rows times their lanes fill the workgroup once), 2 streamed from global memory; -1 (default) the first of
 * these that fits. */
HYPRE_Int hypre_amd_SetSmallTailForm(HYPRE_Int form);
/* first level of the one-workgroup tail in the last cycle of this solver (-1: none, -2: no cycle has run) */
HYPRE_Int hypre_amd_BoomerAMGGetSmallTailLevel(HYPRE_Solver solver);
/* grid / operator complexity of the last setup */
HYPRE_Int hypre_amd_BoomerAMGGetComplexities(HYPRE_Solver solver, HYPRE_Real *grid, HYPRE_Real *op);
/* Multi-rank device hierarchies: levels with at most `rows` global rows are gathered onto every rank at
 * setup and the V-cycle below the first such level runs locally from one all-reduced right-hand side
 * (latency of four halo exchanges per level removed).  Applies to V-cycles with smoothers whose result
 * does not depend on the row distribution (relax 0/7/18 with or without CF ordering, Chebyshev 16); default 16384,
 * 0 disables.  hypre's own relative is the seq_threshold /
 * hypre_seqAMGSetup path (par_amg_setup.c), which re-coarsens the gathered operator instead. */
HYPRE_Int hypre_amd_BoomerAMGSetReplicateThreshold(HYPRE_Solver solver, HYPRE_Int rows);
HYPRE_Int hypre_amd_BoomerAMGGetReplicatedLevel(HYPRE_Solver solver);      /* -1: none */
/* par_amg_solve.c:391-401: operation count of the last cycle (the reference's cycle complexity = this / nnz(A_0)) */
HYPRE_Int hypre_amd_BoomerAMGGetCycleOpCount(HYPRE_Solver solver, HYPRE_Real *count);
/* algorithmic HBM bytes of one cycle on the current hierarchy (SURVEY §8d formula) */
HYPRE_Real hypre_amd_BoomerAMGCycleBytes(HYPRE_Solver solver);
/* level accessors for tests / the oracle harness */
HYPRE_Int hypre_amd_BoomerAMGGetNumLevels(HYPRE_Solver solver);
hypre_ParCSRMatrix *hypre_amd_BoomerAMGGetA(HYPRE_Solver solver, HYPRE_Int level);
hypre_ParCSRMatrix *hypre_amd_BoomerAMGGetP(HYPRE_Solver solver, HYPRE_Int level);
hypre_IntArray     *hypre_amd_BoomerAMGGetCFMarker(HYPRE_Solver solver, HYPRE_Int level);
hypre_Vector       *hypre_amd_BoomerAMGGetL1Norms(HYPRE_Solver solver, HYPRE_Int level);
HYPRE_Int hypre_amd_BoomerAMGGetGridRelaxType(HYPRE_Solver solver, HYPRE_Int k);
HYPRE_Int hypre_amd_BoomerAMGGetNumGridSweeps(HYPRE_Solver solver, HYPRE_Int k);

/* ---- setup building blocks (host) ---- */
HYPRE_Int hypre_BoomerAMGSetup(void *amg_vdata, hypre_ParCSRMatrix *A, hypre_ParVector *f, hypre_ParVector *u);
HYPRE_Int hypre_BoomerAMGCreateS(hypre_ParCSRMatrix *A, HYPRE_Real strength_threshold, HYPRE_Real max_row_sum,
                                 HYPRE_Int num_functions, HYPRE_Int *dof_func, hypre_ParCSRMatrix **S_ptr);
HYPRE_Int hypre_BoomerAMGCoarsenPMIS(hypre_ParCSRMatrix *S, hypre_ParCSRMatrix *A, HYPRE_Int CF_init,
                                     HYPRE_Int debug_flag, hypre_IntArray **CF_marker_ptr);
HYPRE_Int hypre_BoomerAMGCoarsenHMIS(hypre_ParCSRMatrix *S, hypre_ParCSRMatrix *A, HYPRE_Int measure_type,
                                     HYPRE_Int cut_factor, HYPRE_Int debug_flag, hypre_IntArray **CF_marker_ptr);
HYPRE_Int hypre_BoomerAMGBuildExtPIInterp(hypre_ParCSRMatrix *A, HYPRE_Int *CF_marker, hypre_ParCSRMatrix *S,
                                          HYPRE_BigInt *num_cpts_global, HYPRE_Int num_functions,
                                          HYPRE_Int *dof_func, HYPRE_Int debug_flag, HYPRE_Real trunc_factor,
                                          HYPRE_Int max_elmts, hypre_ParCSRMatrix **P_ptr);
HYPRE_Int hypre_BoomerAMGBuildDirInterp(hypre_ParCSRMatrix *A, HYPRE_Int *CF_marker, hypre_ParCSRMatrix *S,
                                        HYPRE_BigInt *num_cpts_global, HYPRE_Int num_functions,
                                        HYPRE_Int *dof_func, HYPRE_Int debug_flag, HYPRE_Real trunc_factor,
                                        HYPRE_Int max_elmts, HYPRE_Int interp_type, hypre_ParCSRMatrix **P_ptr);
HYPRE_Int hypre_BoomerAMGInterpTruncation(hypre_ParCSRMatrix *P, HYPRE_Real trunc_factor, HYPRE_Int max_elmts);
HYPRE_Int hypre_BoomerAMGBuildCoarseOperatorKT(hypre_ParCSRMatrix *RT, hypre_ParCSRMatrix *A,
                                               hypre_ParCSRMatrix *P, HYPRE_Int keepTranspose,
                                               hypre_ParCSRMatrix **RAP_ptr);
HYPRE_Int hypre_ParCSRComputeL1Norms(hypre_ParCSRMatrix *A, HYPRE_Int option, HYPRE_Int *cf_marker,
                                     HYPRE_Real **l1_norm_ptr);
/* parcsr_ls/ams.c:4535-4915: the variant the host routine switches to under OpenMP (options 1, 4, 5, 6) */
HYPRE_Int hypre_ParCSRComputeL1NormsThreads(hypre_ParCSRMatrix *A, HYPRE_Int option, HYPRE_Int num_threads,
                                            HYPRE_Int *cf_marker, HYPRE_Real **l1_norm_ptr);

/* ---- the hot path ---- */
HYPRE_Int hypre_BoomerAMGRelax(hypre_ParCSRMatrix *A, hypre_ParVector *f, HYPRE_Int *cf_marker,
                               HYPRE_Int relax_type, HYPRE_Int relax_points, HYPRE_Real relax_weight,
                               HYPRE_Real omega, HYPRE_Real *l1_norms, hypre_ParVector *u,
                               hypre_ParVector *Vtemp, hypre_ParVector *Ztemp);
HYPRE_Int hypre_BoomerAMGRelaxIF(hypre_ParCSRMatrix *A, hypre_ParVector *f, HYPRE_Int *cf_marker,
                                 HYPRE_Int relax_type, HYPRE_Int relax_order, HYPRE_Int cycle_param,
                                 HYPRE_Real relax_weight, HYPRE_Real omega, HYPRE_Real *l1_norms,
                                 hypre_ParVector *u, hypre_ParVector *Vtemp, hypre_ParVector *Ztemp);
HYPRE_Int hypre_ParCSRRelax_L1_Jacobi(hypre_ParCSRMatrix *A, hypre_ParVector *f, HYPRE_Int *cf_marker,
                                      HYPRE_Int relax_points, HYPRE_Real relax_weight, HYPRE_Real *l1_norms,
                                      hypre_ParVector *u, hypre_ParVector *Vtemp);
HYPRE_Int hypre_BoomerAMGRelax_FCFJacobi(hypre_ParCSRMatrix *A, hypre_ParVector *f, HYPRE_Int *cf_marker,
                                         HYPRE_Real relax_weight, hypre_ParVector *u, hypre_ParVector *Vtemp);
/* Chebyshev polynomial smoothing.
 *   parcsr_ls/par_relax_more.c:34-198   spectrum estimates: Gershgorin discs / Lanczos on k CG steps (host, setup time)
 *   parcsr_ls/par_cheby.c:57-222        coefficients of the degree-(order-1) polynomial and the scaling vector (host)
 *   parcsr_ls/par_cheby.c:405-446, par_cheby_device.c:119-294   u += p(A)(f - A u) (device operands only) */
HYPRE_Int hypre_ParCSRMaxEigEstimate(hypre_ParCSRMatrix *A, HYPRE_Int scale, HYPRE_Real *max_eig, HYPRE_Real *min_eig);
HYPRE_Int hypre_ParCSRMaxEigEstimateCG(hypre_ParCSRMatrix *A, HYPRE_Int scale, HYPRE_Int max_iter,
                                       HYPRE_Real *max_eig, HYPRE_Real *min_eig);
HYPRE_Int hypre_ParCSRRelax_Cheby_Setup(hypre_ParCSRMatrix *A, HYPRE_Real max_eig, HYPRE_Real min_eig,
                                        HYPRE_Real fraction, HYPRE_Int order, HYPRE_Int scale, HYPRE_Int variant,
                                        HYPRE_Real **coefs_ptr, HYPRE_Real **ds_ptr);
HYPRE_Int hypre_ParCSRRelax_Cheby_Solve(hypre_ParCSRMatrix *A, hypre_ParVector *f, HYPRE_Real *ds_data,
                                        HYPRE_Real *coefs, HYPRE_Int order, HYPRE_Int scale, HYPRE_Int variant,
                                        hypre_ParVector *u, hypre_ParVector *v, hypre_ParVector *r,
                                        hypre_ParVector *orig_u_vec, hypre_ParVector *tmp_vec);
/* level data of the Chebyshev smoother for inspection (NULL when relax 16 is not in use) */
HYPRE_Int     hypre_amd_BoomerAMGGetChebyOrderScale(HYPRE_Solver solver, HYPRE_Int *order, HYPRE_Int *scale);
HYPRE_Real   *hypre_amd_BoomerAMGGetChebyCoefs(HYPRE_Solver solver, HYPRE_Int level);
hypre_Vector *hypre_amd_BoomerAMGGetChebyDS(HYPRE_Solver solver, HYPRE_Int level);

HYPRE_Int hypre_BoomerAMGRelaxTwoStageGaussSeidelDevice(hypre_ParCSRMatrix *A, hypre_ParVector *f,
                                                        HYPRE_Real relax_weight, HYPRE_Real omega,
                                                        HYPRE_Real *A_diag_diag, hypre_ParVector *u,
                                                        hypre_ParVector *r, hypre_ParVector *z,
                                                        HYPRE_Int num_inner_iters);
HYPRE_Int hypre_BoomerAMGRelaxHybridGaussSeidelDevice(hypre_ParCSRMatrix *A, hypre_ParVector *f,
                                                      HYPRE_Int *cf_marker, HYPRE_Int relax_points,
                                                      HYPRE_Real relax_weight, HYPRE_Real omega,
                                                      HYPRE_Real *l1_norms, hypre_ParVector *u,
                                                      hypre_ParVector *Vtemp, hypre_ParVector *Ztemp,
                                                      HYPRE_Int GS_order, HYPRE_Int Symm);
/* Multicolour Gauss-Seidel, relax types 21 (colours ascending) and 22 (descending) of hypre_BoomerAMGRelax.  No
 * counterpart in the reference (parcsr_ls/par_relax*.c knows no colouring): it is the hybrid Gauss-Seidel sweep of
 * par_relax.c:691-945 (Jacobi across ranks, Gauss-Seidel inside a rank) on the colour-permuted local ordering, which
 * the GPU can run one colour at a time.  `diag`: smoother diagonal (the l1_norms option-5 vector of relax 7 / 11 / 12)
 * or NULL for the stored diagonal; Vtemp: local work vector (may be NULL on one rank); direction: +1 / -1. */
HYPRE_Int hypre_BoomerAMGRelaxMultiColorGaussSeidelDevice(hypre_ParCSRMatrix *A, hypre_ParVector *f,
                                                          HYPRE_Int *cf_marker, HYPRE_Int relax_points,
                                                          HYPRE_Real relax_weight, HYPRE_Real *diag,
                                                          hypre_ParVector *u, hypre_ParVector *Vtemp,
                                                          HYPRE_Int direction);
/* number of colours of A's diagonal block (greedy first-fit over the symmetrised pattern, built on demand);
 * colors_out: host array of one colour per local row, or NULL */
HYPRE_Int hypre_amd_ParCSRMatrixMultiColoring(hypre_ParCSRMatrix *A, HYPRE_Int *colors_out);
/* The one-workgroup multicolour sweeps (small levels, the tails of small colours of larger ones) request what does not depend
 * on the iterate — a row's place, right-hand side, diagonal, own value and first entries — one pass ahead (default 1), or
 * walk the plain chain of dependent loads (0); < 0 leaves the setting; returns it.  Same results. */
HYPRE_Int hypre_amd_SetMcLookAhead(HYPRE_Int on);
HYPRE_Int hypre_GaussElimSetup(hypre_ParAMGData *amg_data, HYPRE_Int level, HYPRE_Int relax_type);
HYPRE_Int hypre_GaussElimSolve(hypre_ParAMGData *amg_data, HYPRE_Int level, HYPRE_Int relax_type);
HYPRE_Int hypre_BoomerAMGCycle(void *amg_vdata, hypre_ParVector **F_array, hypre_ParVector **U_array);
HYPRE_Int hypre_BoomerAMGSolve(void *amg_vdata, hypre_ParCSRMatrix *A, hypre_ParVector *f, hypre_ParVector *u);

/* ---- PCG with BoomerAMG as preconditioner (the caller in the benchmark configs) ---- */
HYPRE_Int HYPRE_ParCSRPCGCreate(MPI_Comm comm, HYPRE_Solver *solver);
HYPRE_Int HYPRE_ParCSRPCGDestroy(HYPRE_Solver solver);
HYPRE_Int HYPRE_PCGSetTol(HYPRE_Solver solver, HYPRE_Real tol);
HYPRE_Int HYPRE_PCGSetAbsoluteTol(HYPRE_Solver solver, HYPRE_Real a_tol);
HYPRE_Int HYPRE_PCGSetMaxIter(HYPRE_Solver solver, HYPRE_Int max_iter);
HYPRE_Int HYPRE_PCGSetTwoNorm(HYPRE_Solver solver, HYPRE_Int two_norm);
HYPRE_Int HYPRE_PCGSetLogging(HYPRE_Solver solver, HYPRE_Int logging);             /* accepted, inert */
HYPRE_Int HYPRE_PCGSetPrintLevel(HYPRE_Solver solver, HYPRE_Int level);            /* accepted, inert */
HYPRE_Int HYPRE_PCGSetRelChange(HYPRE_Solver solver, HYPRE_Int rel_change);        /* 0 only */
HYPRE_Int HYPRE_PCGSetRecomputeResidual(HYPRE_Solver solver, HYPRE_Int recompute); /* 0 only */
HYPRE_Int HYPRE_PCGSetFlex(HYPRE_Solver solver, HYPRE_Int flex);     /* pcg.c:339-345, 636-639, 720-723, 957-965; before Setup */
HYPRE_Int HYPRE_PCGSetPrecond(HYPRE_Solver solver, HYPRE_PtrToSolverFcn precond,
                              HYPRE_PtrToSolverFcn precond_setup, HYPRE_Solver precond_solver);
/* the diagonal-scaling preconditioner of the reference driver's DS-PCG (`ij -solver 2`; parcsr_ls/HYPRE_parcsr_pcg.c):
 * x = y ./ diag(A); vectors may be multivectors (`-nc N`: test/TEST_ij/vector.jobs) */
HYPRE_Int HYPRE_ParCSRDiagScaleSetup(HYPRE_Solver solver, HYPRE_ParCSRMatrix A, HYPRE_ParVector y, HYPRE_ParVector x);
HYPRE_Int HYPRE_ParCSRDiagScale(HYPRE_Solver solver, HYPRE_ParCSRMatrix HA, HYPRE_ParVector Hy, HYPRE_ParVector Hx);
HYPRE_Int HYPRE_ParCSRPCGSetup(HYPRE_Solver solver, HYPRE_ParCSRMatrix A, HYPRE_ParVector b, HYPRE_ParVector x);
HYPRE_Int HYPRE_ParCSRPCGSolve(HYPRE_Solver solver, HYPRE_ParCSRMatrix A, HYPRE_ParVector b, HYPRE_ParVector x);
HYPRE_Int HYPRE_PCGGetNumIterations(HYPRE_Solver solver, HYPRE_Int *num_iterations);
HYPRE_Int HYPRE_PCGGetFinalRelativeResidualNorm(HYPRE_Solver solver, HYPRE_Real *norm);

/* ---- GMRES with BoomerAMG as right preconditioner (`ij -solver 3`) ----
 * krylov/gmres.c:274-1000, krylov/HYPRE_gmres.c, parcsr_ls/HYPRE_parcsr_gmres.c; defaults of
 * hypre_GMRESCreate (gmres.c:68-110): k_dim 5, tol 1e-6, max_iter 1000.  SetKDim must precede Setup
 * (Setup allocates the k_dim + 1 basis vectors, gmres.c:204-215).  HYPRE_ERROR_CONV when max_iter is
 * reached above the tolerance (gmres.c:982-985). */
HYPRE_Int HYPRE_ParCSRGMRESCreate(MPI_Comm comm, HYPRE_Solver *solver);
HYPRE_Int HYPRE_ParCSRGMRESDestroy(HYPRE_Solver solver);
HYPRE_Int HYPRE_GMRESSetKDim(HYPRE_Solver solver, HYPRE_Int k_dim);
HYPRE_Int HYPRE_GMRESSetTol(HYPRE_Solver solver, HYPRE_Real tol);
HYPRE_Int HYPRE_GMRESSetAbsoluteTol(HYPRE_Solver solver, HYPRE_Real a_tol);
HYPRE_Int HYPRE_GMRESSetMinIter(HYPRE_Solver solver, HYPRE_Int min_iter);
HYPRE_Int HYPRE_GMRESSetMaxIter(HYPRE_Solver solver, HYPRE_Int max_iter);
HYPRE_Int HYPRE_GMRESSetSkipRealResidualCheck(HYPRE_Solver solver, HYPRE_Int skip_real_r_check);
HYPRE_Int HYPRE_GMRESSetLogging(HYPRE_Solver solver, HYPRE_Int logging);           /* accepted, inert */
HYPRE_Int HYPRE_GMRESSetPrintLevel(HYPRE_Solver solver, HYPRE_Int level);          /* accepted, inert */
HYPRE_Int HYPRE_GMRESSetRelChange(HYPRE_Solver solver, HYPRE_Int rel_change);      /* 0 only */
HYPRE_Int HYPRE_GMRESSetPrecond(HYPRE_Solver solver, HYPRE_PtrToSolverFcn precond,
                                HYPRE_PtrToSolverFcn precond_setup, HYPRE_Solver precond_solver);
HYPRE_Int HYPRE_ParCSRGMRESSetup(HYPRE_Solver solver, HYPRE_ParCSRMatrix A, HYPRE_ParVector b, HYPRE_ParVector x);
HYPRE_Int HYPRE_ParCSRGMRESSolve(HYPRE_Solver solver, HYPRE_ParCSRMatrix A, HYPRE_ParVector b, HYPRE_ParVector x);
HYPRE_Int HYPRE_GMRESGetNumIterations(HYPRE_Solver solver, HYPRE_Int *num_iterations);
HYPRE_Int HYPRE_GMRESGetFinalRelativeResidualNorm(HYPRE_Solver solver, HYPRE_Real *norm);
HYPRE_Int HYPRE_GMRESGetConverged(HYPRE_Solver solver, HYPRE_Int *converged);

#ifdef __cplusplus
}
#endif
#endif
