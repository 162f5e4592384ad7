// hypre_amd — BoomerAMG solver object: creation, parameters, destruction.
// Reference: parcsr_ls/par_amg.c:28-520 (defaults at :162-316), HYPRE_parcsr_amg.c.
#include "amg_internal.hpp"

using namespace hamd;

extern "C" {

static HYPRE_Int amg_setup_thunk(HYPRE_Solver s, void *A, void *b, void *x)
{
   return HYPRE_BoomerAMGSetup(s, (HYPRE_ParCSRMatrix) A, (HYPRE_ParVector) b, (HYPRE_ParVector) x);
}
static HYPRE_Int amg_solve_thunk(HYPRE_Solver s, void *A, void *b, void *x)
{
   return HYPRE_BoomerAMGSolve(s, (HYPRE_ParCSRMatrix) A, (HYPRE_ParVector) b, (HYPRE_ParVector) x);
}

HYPRE_Int HYPRE_BoomerAMGCreate(HYPRE_Solver *solver)
{
   if (!solver) { hypre_error_in_arg(1); return hypre_error_flag; }
   hypre_ParAMGData *d = (hypre_ParAMGData *) calloc(1, sizeof(hypre_ParAMGData));
   d->base.setup = amg_setup_thunk;
   d->base.solve = amg_solve_thunk;
   d->base.destroy = HYPRE_BoomerAMGDestroy;
   d->memory_location = handle().memory_location;
   // defaults of par_amg.c:162-316 (host flavour; the driver overrides what it needs)
   d->max_levels = 25;
   d->strong_threshold = 0.25;
   d->max_row_sum = 0.9;
   d->trunc_factor = 0.0;
   d->measure_type = 0;
   d->coarsen_type = 10;
   d->P_max_elmts = 4;
   d->interp_type = 6;
   d->agg_num_levels = 0;
   d->max_coarse_size = 9;
   d->min_coarse_size = 0;
   d->keepTranspose = 1;           // stored P^T: the restriction SpMV never re-transposes
   d->num_functions = 1;
   d->max_iter = 20;
   d->min_iter = 0;
   d->fcycle = 0;
   d->cycle_type = 1;
   d->converge_type = 0;
   d->tol = 1.0e-6;
   d->relax_order = 0;
   d->user_coarse_relax_type = 9;
   d->user_relax_type = -1;
   d->user_num_sweeps = -1;
   d->user_relax_weight = 1.0;
   d->outer_wt = 1.0;
   d->num_grid_sweeps = (HYPRE_Int *) calloc(4, sizeof(HYPRE_Int));
   d->grid_relax_type = (HYPRE_Int *) calloc(4, sizeof(HYPRE_Int));
   for (int k = 0; k < 4; k++) { d->num_grid_sweeps[k] = 1; }
   // SetCycleRelaxType(13,1), (14,2), (9,3) on a fresh array initialised to {3,3,3,9}
   d->grid_relax_type[0] = 3; d->grid_relax_type[1] = 13; d->grid_relax_type[2] = 14; d->grid_relax_type[3] = 9;
   // par_amg.c:273-277
   d->cheby_order = 2; d->cheby_variant = 0; d->cheby_scale = 1; d->cheby_eig_est = 10; d->cheby_fraction = 0.3;
   d->amd_private = new AmgPrivate();
   *solver = (HYPRE_Solver) d;
   return hypre_error_flag;
}

void amg_free_hierarchy(hypre_ParAMGData *d)
{
   AmgPrivate *pv = (AmgPrivate *) d->amd_private;
   destroy_replicated_tail(d);
   if (pv)
   {
      pv->release_device();
      for (HYPRE_Solver cg : pv->cg_smoothers) { if (cg) { HYPRE_ParCSRPCGDestroy(cg); } }
      pv->cg_smoothers.clear();
   }
   const int L = d->num_levels;
   if (d->A_array)
   {
      for (int l = 1; l < L; l++) { hypre_ParCSRMatrixDestroy(d->A_array[l]); }
      free(d->A_array); d->A_array = nullptr;
   }
   if (d->P_array)
   {
      for (int l = 0; l < L - 1; l++) { hypre_ParCSRMatrixDestroy(d->P_array[l]); }
      free(d->P_array); d->P_array = nullptr; d->R_array = nullptr;
   }
   if (d->F_array)
   {
      for (int l = 1; l < L; l++) { hypre_ParVectorDestroy(d->F_array[l]); hypre_ParVectorDestroy(d->U_array[l]); }
      free(d->F_array); free(d->U_array); d->F_array = d->U_array = nullptr;
   }
   if (d->CF_marker_array)
   {
      for (int l = 0; l < L; l++) { hypre_IntArrayDestroy(d->CF_marker_array[l]); }
      free(d->CF_marker_array); d->CF_marker_array = nullptr;
   }
   if (d->l1_norms)
   {
      for (int l = 0; l < L; l++) { hypre_SeqVectorDestroy(d->l1_norms[l]); }
      free(d->l1_norms); d->l1_norms = nullptr;
   }
   if (d->cheby_ds)
   {
      for (int l = 0; l < L; l++) { hypre_SeqVectorDestroy(d->cheby_ds[l]); }
      free(d->cheby_ds); d->cheby_ds = nullptr;
   }
   if (d->cheby_coefs)
   {
      for (int l = 0; l < L; l++) { hypre_Free(d->cheby_coefs[l], HYPRE_MEMORY_HOST); }
      free(d->cheby_coefs); d->cheby_coefs = nullptr;
   }
   free(d->max_eig_est); d->max_eig_est = nullptr;
   free(d->min_eig_est); d->min_eig_est = nullptr;
   hypre_ParVectorDestroy(d->Vtemp); d->Vtemp = nullptr;
   hypre_ParVectorDestroy(d->Ztemp); d->Ztemp = nullptr;
   hypre_ParVectorDestroy(d->Rtemp); d->Rtemp = nullptr;
   hypre_ParVectorDestroy(d->Ptemp); d->Ptemp = nullptr;
   free(d->A_mat); d->A_mat = nullptr;
   free(d->b_vec); d->b_vec = nullptr;
   free(d->relax_weight); d->relax_weight = nullptr;
   free(d->omega); d->omega = nullptr;
   d->num_levels = 0;
   d->gs_setup = 0;
}

HYPRE_Int HYPRE_BoomerAMGDestroy(HYPRE_Solver solver)
{
   hypre_ParAMGData *d = (hypre_ParAMGData *) solver;
   if (!d) { return hypre_error_flag; }
   amg_free_hierarchy(d);
   delete (AmgPrivate *) d->amd_private;
   free(d->num_grid_sweeps);
   free(d->grid_relax_type);
   if (d->grid_relax_points)
   {
      for (int k = 0; k < 4; k++) { free(d->grid_relax_points[k]); }
      free(d->grid_relax_points);
   }
   free(d);
   return hypre_error_flag;
}

#define AMG_DATA(solver, d)                                              \
   hypre_ParAMGData *d = (hypre_ParAMGData *) (solver);                  \
   if (!d) { hypre_error_in_arg(1); return hypre_error_flag; }

HYPRE_Int HYPRE_BoomerAMGSetMaxLevels(HYPRE_Solver s, HYPRE_Int v)
{ AMG_DATA(s, d); if (v < 1) { hypre_error_in_arg(2); return hypre_error_flag; } d->max_levels = v; return hypre_error_flag; }
HYPRE_Int HYPRE_BoomerAMGSetMaxCoarseSize(HYPRE_Solver s, HYPRE_Int v)
{ AMG_DATA(s, d); if (v < 1) { hypre_error_in_arg(2); return hypre_error_flag; } d->max_coarse_size = v; return hypre_error_flag; }
HYPRE_Int HYPRE_BoomerAMGSetMinCoarseSize(HYPRE_Solver s, HYPRE_Int v)
{ AMG_DATA(s, d); if (v < 0) { hypre_error_in_arg(2); return hypre_error_flag; } d->min_coarse_size = v; return hypre_error_flag; }
HYPRE_Int HYPRE_BoomerAMGSetStrongThreshold(HYPRE_Solver s, HYPRE_Real v)
{ AMG_DATA(s, d); if (v < 0 || v > 1) { hypre_error_in_arg(2); return hypre_error_flag; } d->strong_threshold = v; return hypre_error_flag; }
HYPRE_Int HYPRE_BoomerAMGSetMaxRowSum(HYPRE_Solver s, HYPRE_Real v)
{ AMG_DATA(s, d); if (v <= 0 || v > 1) { hypre_error_in_arg(2); return hypre_error_flag; } d->max_row_sum = v; return hypre_error_flag; }
HYPRE_Int HYPRE_BoomerAMGSetCoarsenType(HYPRE_Solver s, HYPRE_Int v) { AMG_DATA(s, d); d->coarsen_type = v; return hypre_error_flag; }
HYPRE_Int HYPRE_BoomerAMGSetInterpType(HYPRE_Solver s, HYPRE_Int v) { AMG_DATA(s, d); d->interp_type = v; return hypre_error_flag; }
HYPRE_Int HYPRE_BoomerAMGSetTruncFactor(HYPRE_Solver s, HYPRE_Real v)
{ AMG_DATA(s, d); if (v < 0 || v >= 1) { hypre_error_in_arg(2); return hypre_error_flag; } d->trunc_factor = v; return hypre_error_flag; }
HYPRE_Int HYPRE_BoomerAMGSetPMaxElmts(HYPRE_Solver s, HYPRE_Int v)
{ AMG_DATA(s, d); if (v < 0) { hypre_error_in_arg(2); return hypre_error_flag; } d->P_max_elmts = v; return hypre_error_flag; }
HYPRE_Int HYPRE_BoomerAMGSetKeepTranspose(HYPRE_Solver s, HYPRE_Int v) { AMG_DATA(s, d); d->keepTranspose = v; return hypre_error_flag; }
HYPRE_Int HYPRE_BoomerAMGSetFilterFunctions(HYPRE_Solver s, HYPRE_Int v)
{
   AMG_DATA(s, d);
   if (v < 0 || v > 1) { hypre_error_in_arg(2); return hypre_error_flag; }      // par_amg.c:3232-3250
   ((AmgPrivate *) d->amd_private)->filter_functions = v != 0;
   return hypre_error_flag;
}
HYPRE_Int HYPRE_BoomerAMGSetNumFunctions(HYPRE_Solver s, HYPRE_Int v)
{
   AMG_DATA(s, d);
   if (v < 1) { hypre_error_in_arg(2); return hypre_error_flag; }      // par_amg.c hypre_BoomerAMGSetNumFunctions
   d->num_functions = v;
   return hypre_error_flag;
}
HYPRE_Int HYPRE_BoomerAMGSetTol(HYPRE_Solver s, HYPRE_Real v)
{ AMG_DATA(s, d); if (v < 0 || v > 1) { hypre_error_in_arg(2); return hypre_error_flag; } d->tol = v; return hypre_error_flag; }
HYPRE_Int HYPRE_BoomerAMGSetMaxIter(HYPRE_Solver s, HYPRE_Int v)
{ AMG_DATA(s, d); if (v < 0) { hypre_error_in_arg(2); return hypre_error_flag; } d->max_iter = v; return hypre_error_flag; }
HYPRE_Int HYPRE_BoomerAMGSetMinIter(HYPRE_Solver s, HYPRE_Int v) { AMG_DATA(s, d); d->min_iter = v; return hypre_error_flag; }
HYPRE_Int HYPRE_BoomerAMGSetConvergeType(HYPRE_Solver s, HYPRE_Int v) { AMG_DATA(s, d); d->converge_type = v; return hypre_error_flag; }
HYPRE_Int HYPRE_BoomerAMGSetCycleType(HYPRE_Solver s, HYPRE_Int v)
{ AMG_DATA(s, d); if (v < 1 || v > 2) { hypre_error_in_arg(2); return hypre_error_flag; } d->cycle_type = v; return hypre_error_flag; }
HYPRE_Int HYPRE_BoomerAMGSetFCycle(HYPRE_Solver s, HYPRE_Int v) { AMG_DATA(s, d); d->fcycle = v != 0; return hypre_error_flag; }

// par_amg.c SetNumSweeps: levels 0..2 get num_sweeps, the coarsest gets 1
HYPRE_Int HYPRE_BoomerAMGSetNumSweeps(HYPRE_Solver s, HYPRE_Int v)
{
   AMG_DATA(s, d);
   if (v < 1) { hypre_error_in_arg(2); return hypre_error_flag; }
   for (int k = 0; k < 3; k++) { d->num_grid_sweeps[k] = v; }
   d->num_grid_sweeps[3] = 1;
   d->user_num_sweeps = v;
   return hypre_error_flag;
}
HYPRE_Int HYPRE_BoomerAMGSetCycleNumSweeps(HYPRE_Solver s, HYPRE_Int v, HYPRE_Int k)
{
   AMG_DATA(s, d);
   if (v < 0) { hypre_error_in_arg(2); return hypre_error_flag; }
   if (k < 1 || k > 3) { hypre_error_in_arg(3); return hypre_error_flag; }
   d->num_grid_sweeps[k] = v;
   return hypre_error_flag;
}
// par_amg.c SetRelaxType: down/up/"fine" get relax_type, the coarsest is reset to 9
HYPRE_Int HYPRE_BoomerAMGSetRelaxType(HYPRE_Solver s, HYPRE_Int v)
{
   AMG_DATA(s, d);
   if (v < 0) { hypre_error_in_arg(2); return hypre_error_flag; }
   for (int k = 0; k < 3; k++) { d->grid_relax_type[k] = v; }
   d->grid_relax_type[3] = 9;
   d->user_coarse_relax_type = 9;
   d->user_relax_type = v;
   return hypre_error_flag;
}
HYPRE_Int HYPRE_BoomerAMGSetCycleRelaxType(HYPRE_Solver s, HYPRE_Int v, HYPRE_Int k)
{
   AMG_DATA(s, d);
   if (k < 1 || k > 3) { hypre_error_in_arg(3); return hypre_error_flag; }
   if (v < 0) { hypre_error_in_arg(2); return hypre_error_flag; }
   d->grid_relax_type[k] = v;
   if (k == 3) { d->user_coarse_relax_type = v; }
   return hypre_error_flag;
}
HYPRE_Int HYPRE_BoomerAMGSetRelaxOrder(HYPRE_Solver s, HYPRE_Int v) { AMG_DATA(s, d); d->relax_order = v; return hypre_error_flag; }
HYPRE_Int HYPRE_BoomerAMGSetRelaxWt(HYPRE_Solver s, HYPRE_Real v)
{
   // par_amg.c:2436-2463: the uniform value replaces every level's weight, earlier level overrides included
   AMG_DATA(s, d);
   d->user_relax_weight = v;
   ((AmgPrivate *) d->amd_private)->level_relax_wt.clear();
   return hypre_error_flag;
}
HYPRE_Int HYPRE_BoomerAMGSetOuterWt(HYPRE_Solver s, HYPRE_Real v)
{
   AMG_DATA(s, d);
   d->outer_wt = v;
   ((AmgPrivate *) d->amd_private)->level_outer_wt.clear();
   return hypre_error_flag;
}
// par_amg.c:2466-2495, 2590-2620: weight of one level (level < max_levels, else HYPRE_ERROR_ARG on argument 3)
HYPRE_Int HYPRE_BoomerAMGSetLevelRelaxWt(HYPRE_Solver s, HYPRE_Real v, HYPRE_Int level)
{
   AMG_DATA(s, d);
   if (level > d->max_levels - 1 || level < 0) { hypre_error_in_arg(3); return hypre_error_flag; }
   ((AmgPrivate *) d->amd_private)->level_relax_wt.emplace_back((int) level, (double) v);
   return hypre_error_flag;
}
HYPRE_Int HYPRE_BoomerAMGSetLevelOuterWt(HYPRE_Solver s, HYPRE_Real v, HYPRE_Int level)
{
   AMG_DATA(s, d);
   if (level > d->max_levels - 1 || level < 0) { hypre_error_in_arg(3); return hypre_error_flag; }
   ((AmgPrivate *) d->amd_private)->level_outer_wt.emplace_back((int) level, (double) v);
   return hypre_error_flag;
}
HYPRE_Int HYPRE_BoomerAMGSetPrintLevel(HYPRE_Solver s, HYPRE_Int v) { AMG_DATA(s, d); d->print_level = v; return hypre_error_flag; }
HYPRE_Int HYPRE_BoomerAMGSetLogging(HYPRE_Solver s, HYPRE_Int v) { AMG_DATA(s, d); d->logging = v; return hypre_error_flag; }

// ---- option gates -------------------------------------------------------------------------------------------------
// Setters of HYPRE_parcsr_amg.c that an application written against hypre calls as a matter of course, for options
// whose non-default branches are not built here.  The value that selects what this library implements is accepted;
// anything else raises HYPRE_ERROR_ARG on argument 2 with a message, it is never dropped silently.
namespace {
HYPRE_Int gate(const char *what, bool ok)
{
   if (!ok)
   {
      hypre_error_in_arg(2);
      hypre_error_w_msg(HYPRE_ERROR_GENERIC, what);
   }
   return hypre_error_flag;
}
}  // namespace
HYPRE_Int HYPRE_BoomerAMGSetMeasureType(HYPRE_Solver s, HYPRE_Int v)       // local (0) or global (1) measures of the first HMIS pass
{
   AMG_DATA(s, d);
   if (v != 0 && v != 1) { hypre_error_in_arg(2); return hypre_error_flag; }
   d->measure_type = v;
   return hypre_error_flag;
}
HYPRE_Int HYPRE_BoomerAMGSetDebugFlag(HYPRE_Solver s, HYPRE_Int v) { AMG_DATA(s, d); (void) d; (void) v; return hypre_error_flag; }
HYPRE_Int HYPRE_BoomerAMGSetNumPaths(HYPRE_Solver s, HYPRE_Int v) { AMG_DATA(s, d); (void) d; (void) v; return hypre_error_flag; }      // aggressive coarsening only
HYPRE_Int HYPRE_BoomerAMGSetAggNumLevels(HYPRE_Solver s, HYPRE_Int v)
{ AMG_DATA(s, d); (void) d; return gate("HYPRE_BoomerAMGSetAggNumLevels: aggressive coarsening is not built (0 only)", v == 0); }
HYPRE_Int HYPRE_BoomerAMGSetNodal(HYPRE_Solver s, HYPRE_Int v)
{ AMG_DATA(s, d); (void) d; return gate("HYPRE_BoomerAMGSetNodal: nodal systems coarsening is not built (0 only)", v == 0); }
HYPRE_Int HYPRE_BoomerAMGSetSeqThreshold(HYPRE_Solver s, HYPRE_Int v)
{ AMG_DATA(s, d); (void) d; return gate("HYPRE_BoomerAMGSetSeqThreshold: the sequential coarse solve is not built (0 only; see hypre_amd_BoomerAMGSetReplicateThreshold)", v == 0); }
HYPRE_Int HYPRE_BoomerAMGSetRedundant(HYPRE_Solver s, HYPRE_Int v)
{ AMG_DATA(s, d); (void) d; return gate("HYPRE_BoomerAMGSetRedundant: 0 only", v == 0); }
HYPRE_Int HYPRE_BoomerAMGSetRAP2(HYPRE_Solver s, HYPRE_Int v)
{ AMG_DATA(s, d); (void) d; return gate("HYPRE_BoomerAMGSetRAP2: the coarse operator is the Galerkin triple product (0 only)", v == 0); }
HYPRE_Int HYPRE_BoomerAMGSetRestriction(HYPRE_Solver s, HYPRE_Int v)
{ AMG_DATA(s, d); (void) d; return gate("HYPRE_BoomerAMGSetRestriction: R = P^T (0 only)", v == 0); }
HYPRE_Int HYPRE_BoomerAMGSetSmoothNumLevels(HYPRE_Solver s, HYPRE_Int v)
{ AMG_DATA(s, d); (void) d; return gate("HYPRE_BoomerAMGSetSmoothNumLevels: Schwarz / ILU / FSAI / ParaSails smoothers are not built (0 only)", v == 0); }
HYPRE_Int HYPRE_BoomerAMGSetSmoothType(HYPRE_Solver s, HYPRE_Int v) { AMG_DATA(s, d); (void) d; (void) v; return hypre_error_flag; }    // inert while SmoothNumLevels is 0
HYPRE_Int HYPRE_BoomerAMGSetSmoothNumSweeps(HYPRE_Solver s, HYPRE_Int v) { AMG_DATA(s, d); (void) d; (void) v; return hypre_error_flag; }
HYPRE_Int HYPRE_BoomerAMGSetAdditive(HYPRE_Solver s, HYPRE_Int v)
{ AMG_DATA(s, d); (void) d; return gate("HYPRE_BoomerAMGSetAdditive: additive cycles are not built (-1 only)", v == -1); }
HYPRE_Int HYPRE_BoomerAMGSetMultAdditive(HYPRE_Solver s, HYPRE_Int v)
{ AMG_DATA(s, d); (void) d; return gate("HYPRE_BoomerAMGSetMultAdditive: additive cycles are not built (-1 only)", v == -1); }
HYPRE_Int HYPRE_BoomerAMGSetSimple(HYPRE_Solver s, HYPRE_Int v)
{ AMG_DATA(s, d); (void) d; return gate("HYPRE_BoomerAMGSetSimple: additive cycles are not built (-1 only)", v == -1); }
HYPRE_Int HYPRE_BoomerAMGSetNonGalerkinTol(HYPRE_Solver s, HYPRE_Real v)
{ AMG_DATA(s, d); (void) d; return gate("HYPRE_BoomerAMGSetNonGalerkinTol: non-Galerkin coarse operators are not built (0.0 only)", v == 0.0); }
HYPRE_Int HYPRE_BoomerAMGSetADropTol(HYPRE_Solver s, HYPRE_Real v)
{ AMG_DATA(s, d); (void) d; return gate("HYPRE_BoomerAMGSetADropTol: dropping from coarse operators is not built (0.0 only)", v == 0.0); }
HYPRE_Int HYPRE_BoomerAMGGetNumIterations(HYPRE_Solver s, HYPRE_Int *v)
{ AMG_DATA(s, d); *v = d->num_iterations; return hypre_error_flag; }
HYPRE_Int HYPRE_BoomerAMGGetFinalRelativeResidualNorm(HYPRE_Solver s, HYPRE_Real *v)
{ AMG_DATA(s, d); *v = d->rel_resid_norm; return hypre_error_flag; }

// par_amg.c:4583-4670
HYPRE_Int HYPRE_BoomerAMGSetChebyOrder(HYPRE_Solver s, HYPRE_Int v)
{ AMG_DATA(s, d); if (v < 1) { hypre_error_in_arg(2); return hypre_error_flag; } d->cheby_order = v; return hypre_error_flag; }
HYPRE_Int HYPRE_BoomerAMGSetChebyFraction(HYPRE_Solver s, HYPRE_Real v)
{ AMG_DATA(s, d); if (v <= 0.0 || v > 1.0) { hypre_error_in_arg(2); return hypre_error_flag; } d->cheby_fraction = v; return hypre_error_flag; }
HYPRE_Int HYPRE_BoomerAMGSetChebyEigEst(HYPRE_Solver s, HYPRE_Int v)
{ AMG_DATA(s, d); if (v < 0) { hypre_error_in_arg(2); return hypre_error_flag; } d->cheby_eig_est = v; return hypre_error_flag; }
HYPRE_Int HYPRE_BoomerAMGSetChebyVariant(HYPRE_Solver s, HYPRE_Int v) { AMG_DATA(s, d); d->cheby_variant = v; return hypre_error_flag; }
HYPRE_Int HYPRE_BoomerAMGSetChebyScale(HYPRE_Solver s, HYPRE_Int v) { AMG_DATA(s, d); d->cheby_scale = v; return hypre_error_flag; }
HYPRE_Int hypre_amd_BoomerAMGGetChebyOrderScale(HYPRE_Solver s, HYPRE_Int *order, HYPRE_Int *scale)
{ AMG_DATA(s, d); if (order) { *order = d->cheby_order; } if (scale) { *scale = d->cheby_scale; } return hypre_error_flag; }
HYPRE_Real *hypre_amd_BoomerAMGGetChebyCoefs(HYPRE_Solver s, HYPRE_Int l)
{ hypre_ParAMGData *d = (hypre_ParAMGData *) s; return (d && d->cheby_coefs && l >= 0 && l < d->num_levels) ? d->cheby_coefs[l] : nullptr; }
hypre_Vector *hypre_amd_BoomerAMGGetChebyDS(HYPRE_Solver s, HYPRE_Int l)
{ hypre_ParAMGData *d = (hypre_ParAMGData *) s; return (d && d->cheby_ds && l >= 0 && l < d->num_levels) ? d->cheby_ds[l] : nullptr; }

// global row count at or below which the levels of a multi-rank device hierarchy are replicated on every
// rank (0 disables; see par_amg_replicate.cpp)
HYPRE_Int hypre_amd_BoomerAMGSetReplicateThreshold(HYPRE_Solver s, HYPRE_Int rows)
{ AMG_DATA(s, d); ((AmgPrivate *) d->amd_private)->replicate_rows = rows < 0 ? 0 : rows; return hypre_error_flag; }
HYPRE_Int hypre_amd_BoomerAMGGetSmallTailLevel(HYPRE_Solver s)
{
   hypre_ParAMGData *d = (hypre_ParAMGData *) s;
   return d && d->amd_private ? ((AmgPrivate *) d->amd_private)->small_tail_used : -2;
}
HYPRE_Int hypre_amd_BoomerAMGGetReplicatedLevel(HYPRE_Solver s)
{ hypre_ParAMGData *d = (hypre_ParAMGData *) s; return (d && ((AmgPrivate *) d->amd_private)->tail) ? ((AmgPrivate *) d->amd_private)->tail_level : -1; }

HYPRE_Int hypre_amd_BoomerAMGSetMemoryLocation(HYPRE_Solver s, HYPRE_MemoryLocation loc)
{ AMG_DATA(s, d); d->memory_location = loc; return hypre_error_flag; }
HYPRE_Int hypre_amd_BoomerAMGSetNumThreads(HYPRE_Solver s, HYPRE_Int n)
{ AMG_DATA(s, d); ((AmgPrivate *) d->amd_private)->emulated_threads = n < 1 ? 1 : n; return hypre_error_flag; }
// Coarse tail of the single-rank V-cycle as one HIP graph: levels with at most `rows` rows (0 switches the graph off;
// default 100000).  Speed only.
HYPRE_Int hypre_amd_BoomerAMGSetGraphThreshold(HYPRE_Solver s, HYPRE_Int rows)
{
   AMG_DATA(s, d);
   AmgPrivate *pv = (AmgPrivate *) d->amd_private;
   pv->drop_graph();
   pv->graph_level = -1;
   pv->graph_rows = rows > 0 ? rows : 0;
   return hypre_error_flag;
}
// first level of the graph (-1: none recorded) and the number of nodes it holds
HYPRE_Int hypre_amd_BoomerAMGGetGraphInfo(HYPRE_Solver s, HYPRE_Int *level, HYPRE_Int *nodes)
{
   AMG_DATA(s, d);
   AmgPrivate *pv = (AmgPrivate *) d->amd_private;
   if (level) { *level = pv->graph_state == 2 ? pv->graph_level : -1; }
   if (nodes) { *nodes = pv->graph_state == 2 ? pv->graph_launches : 0; }
   return hypre_error_flag;
}
HYPRE_Int hypre_amd_BoomerAMGSetMixedPrecision(HYPRE_Solver s, HYPRE_Int on)
{ AMG_DATA(s, d); ((AmgPrivate *) d->amd_private)->mixed_precision = on != 0; return hypre_error_flag; }

HYPRE_Int hypre_amd_BoomerAMGGetComplexities(HYPRE_Solver s, HYPRE_Real *grid, HYPRE_Real *op)
{
   AMG_DATA(s, d);
   double rows = 0, nnz = 0;
   for (int l = 0; l < d->num_levels; l++)
   {
      rows += (double) d->A_array[l]->global_num_rows;
      nnz += d->A_array[l]->d_num_nonzeros;
   }
   if (d->num_levels > 0)
   {
      if (grid) { *grid = rows / (double) d->A_array[0]->global_num_rows; }
      if (op) { *op = nnz / d->A_array[0]->d_num_nonzeros; }
   }
   return hypre_error_flag;
}

// operations of the last cycle as the reference counts them (par_cycle.c:413-430); divided by the
// fine operator's nonzeros this is the "cycle" complexity the reference prints
HYPRE_Int hypre_amd_BoomerAMGGetCycleOpCount(HYPRE_Solver s, HYPRE_Real *count)
{ AMG_DATA(s, d); if (count) { *count = d->cycle_op_count; } return hypre_error_flag; }

HYPRE_Int hypre_amd_BoomerAMGGetNumLevels(HYPRE_Solver s) { return s ? ((hypre_ParAMGData *) s)->num_levels : 0; }
hypre_ParCSRMatrix *hypre_amd_BoomerAMGGetA(HYPRE_Solver s, HYPRE_Int l)
{ hypre_ParAMGData *d = (hypre_ParAMGData *) s; return (d && l >= 0 && l < d->num_levels) ? d->A_array[l] : nullptr; }
hypre_ParCSRMatrix *hypre_amd_BoomerAMGGetP(HYPRE_Solver s, HYPRE_Int l)
{ hypre_ParAMGData *d = (hypre_ParAMGData *) s; return (d && l >= 0 && l < d->num_levels - 1) ? d->P_array[l] : nullptr; }
hypre_IntArray *hypre_amd_BoomerAMGGetCFMarker(HYPRE_Solver s, HYPRE_Int l)
{ hypre_ParAMGData *d = (hypre_ParAMGData *) s; return (d && d->CF_marker_array && l >= 0 && l < d->num_levels) ? d->CF_marker_array[l] : nullptr; }
hypre_Vector *hypre_amd_BoomerAMGGetL1Norms(HYPRE_Solver s, HYPRE_Int l)
{ hypre_ParAMGData *d = (hypre_ParAMGData *) s; return (d && d->l1_norms && l >= 0 && l < d->num_levels) ? d->l1_norms[l] : nullptr; }
HYPRE_Int hypre_amd_BoomerAMGGetGridRelaxType(HYPRE_Solver s, HYPRE_Int k) { return ((hypre_ParAMGData *) s)->grid_relax_type[k]; }
HYPRE_Int hypre_amd_BoomerAMGGetNumGridSweeps(HYPRE_Solver s, HYPRE_Int k) { return ((hypre_ParAMGData *) s)->num_grid_sweeps[k]; }

HYPRE_Int HYPRE_BoomerAMGSetup(HYPRE_Solver solver, HYPRE_ParCSRMatrix A, HYPRE_ParVector b, HYPRE_ParVector x)
{
   if (!A) { hypre_error_in_arg(2); return hypre_error_flag; }
   return hypre_BoomerAMGSetup((void *) solver, A, b, x);
}

// HYPRE_parcsr_amg.c:64-91
HYPRE_Int HYPRE_BoomerAMGSolve(HYPRE_Solver solver, HYPRE_ParCSRMatrix A, HYPRE_ParVector b, HYPRE_ParVector x)
{
   if (!A) { hypre_error_in_arg(2); return hypre_error_flag; }
   if (!b) { hypre_error_in_arg(3); return hypre_error_flag; }
   if (!x) { hypre_error_in_arg(4); return hypre_error_flag; }
   return hypre_BoomerAMGSolve((void *) solver, A, b, x);
}

}  // extern "C"
