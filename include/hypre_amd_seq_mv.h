/*
 * hypre_amd — local (per-rank) CSR matrix and dense vector, and the
 * sequential matrix-vector / BLAS-1 entry points of the hot path.
 *
 * Struct layouts follow the reference in the configuration fixed in
 * HYPRE_amd_utilities.h (no vendor sparse library => the four trailing
 * vendor members of hypre_CSRMatrix are absent):
 *   seq_mv/csr_matrix.h:33-58   hypre_CSRMatrix
 *   seq_mv/vector.h:22-40       hypre_Vector
 * Functions replace:
 *   seq_mv/csr_matvec.c:860,894,1142      Matvec / MatvecOutOfPlace / MatvecT
 *   seq_mv/csr_matvec_device.c:109        hypre_CSRMatrixMatvecDevice
 *   seq_mv/csr_spmv_device.c:381          hypre_CSRMatrixSpMVDevice
 *   seq_mv/vector.c:351,436,653,723,789,953,1030,1070  BLAS-1 dispatchers
 *   seq_mv/vector_device.c:26-316         their *Device bodies
 *   utilities/device_utils.c:667,713,2481 IVAXPY / IVAXPYMarked / DiagScaleVector2
 *
 * Execution: the compute functions run on the GPU only.  Operands whose
 * memory_location is HYPRE_MEMORY_HOST make them raise HYPRE_ERROR_GENERIC
 * ("host execution is not part of this library") — there is no CPU path.
 */
#ifndef HYPRE_AMD_SEQ_MV_H
#define HYPRE_AMD_SEQ_MV_H

#include "HYPRE_amd_utilities.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct
{
   HYPRE_Int            *i;
   HYPRE_Int            *j;
   HYPRE_BigInt         *big_j;
   HYPRE_Int             num_rows;
   HYPRE_Int             num_cols;
   HYPRE_Int             num_nonzeros;
   hypre_int            *i_short;
   hypre_int            *j_short;
   HYPRE_Int             owns_data;
   HYPRE_Int             pattern_only;
   HYPRE_Complex        *data;
   HYPRE_Int            *rownnz;      /* ids of the non-empty rows (offd blocks) */
   HYPRE_Int             num_rownnz;
   HYPRE_MemoryLocation  memory_location;
} hypre_CSRMatrix;

#define hypre_CSRMatrixData(matrix)            ((matrix) -> data)
#define hypre_CSRMatrixI(matrix)               ((matrix) -> i)
#define hypre_CSRMatrixJ(matrix)               ((matrix) -> j)
#define hypre_CSRMatrixBigJ(matrix)            ((matrix) -> big_j)
#define hypre_CSRMatrixNumRows(matrix)         ((matrix) -> num_rows)
#define hypre_CSRMatrixNumCols(matrix)         ((matrix) -> num_cols)
#define hypre_CSRMatrixNumNonzeros(matrix)     ((matrix) -> num_nonzeros)
#define hypre_CSRMatrixRownnz(matrix)          ((matrix) -> rownnz)
#define hypre_CSRMatrixNumRownnz(matrix)       ((matrix) -> num_rownnz)
#define hypre_CSRMatrixOwnsData(matrix)        ((matrix) -> owns_data)
#define hypre_CSRMatrixPatternOnly(matrix)     ((matrix) -> pattern_only)
#define hypre_CSRMatrixMemoryLocation(matrix)  ((matrix) -> memory_location)

typedef struct
{
   HYPRE_Complex        *data;
   HYPRE_Int             size;
   HYPRE_Int             component;
   HYPRE_Int             owns_data;
   HYPRE_MemoryLocation  memory_location;
   HYPRE_Int             num_vectors;
   HYPRE_Int             multivec_storage_method;
   HYPRE_Int             vecstride, idxstride;   /* v_j[i] = data[j*vecstride + i*idxstride] */
} hypre_Vector;

#define hypre_VectorData(vector)                  ((vector) -> data)
#define hypre_VectorSize(vector)                  ((vector) -> size)
#define hypre_VectorComponent(vector)             ((vector) -> component)
#define hypre_VectorOwnsData(vector)              ((vector) -> owns_data)
#define hypre_VectorMemoryLocation(vector)        ((vector) -> memory_location)
#define hypre_VectorNumVectors(vector)            ((vector) -> num_vectors)
#define hypre_VectorMultiVecStorageMethod(vector) ((vector) -> multivec_storage_method)
#define hypre_VectorVectorStride(vector)          ((vector) -> vecstride)
#define hypre_VectorIndexStride(vector)           ((vector) -> idxstride)

/* triangular-part selectors of hypre_CSRMatrixSpMVDevice — seq_mv/csr_spmv_device.h:13-17 */
#define HYPRE_SPMV_FILL_STRICT_LOWER -2
#define HYPRE_SPMV_FILL_LOWER        -1
#define HYPRE_SPMV_FILL_WHOLE         0
#define HYPRE_SPMV_FILL_UPPER         1
#define HYPRE_SPMV_FILL_STRICT_UPPER  2

/* ---- objects (seq_mv/csr_matrix.c, seq_mv/vector.c) ---- */
hypre_CSRMatrix *hypre_CSRMatrixCreate(HYPRE_Int num_rows, HYPRE_Int num_cols, HYPRE_Int num_nonzeros);
HYPRE_Int hypre_CSRMatrixInitialize_v2(hypre_CSRMatrix *matrix, HYPRE_Int bigInit,
                                       HYPRE_MemoryLocation memory_location);
HYPRE_Int hypre_CSRMatrixInitialize(hypre_CSRMatrix *matrix);
HYPRE_Int hypre_CSRMatrixDestroy(hypre_CSRMatrix *matrix);
HYPRE_Int hypre_CSRMatrixSetRownnz(hypre_CSRMatrix *matrix);
HYPRE_Int hypre_CSRMatrixMigrate(hypre_CSRMatrix *A, HYPRE_MemoryLocation memory_location);
hypre_CSRMatrix *hypre_CSRMatrixClone_v2(hypre_CSRMatrix *A, HYPRE_Int copy_data,
                                         HYPRE_MemoryLocation memory_location);
HYPRE_Int hypre_CSRMatrixTranspose(hypre_CSRMatrix *A, hypre_CSRMatrix **AT, HYPRE_Int data);
HYPRE_Int hypre_CSRMatrixReorder(hypre_CSRMatrix *A);   /* diagonal entry first; host matrices */

hypre_Vector *hypre_SeqVectorCreate(HYPRE_Int size);
hypre_Vector *hypre_SeqMultiVectorCreate(HYPRE_Int size, HYPRE_Int num_vectors);
HYPRE_Int hypre_SeqVectorInitialize_v2(hypre_Vector *vector, HYPRE_MemoryLocation memory_location);
HYPRE_Int hypre_SeqVectorInitialize(hypre_Vector *vector);
HYPRE_Int hypre_SeqVectorDestroy(hypre_Vector *vector);
HYPRE_Int hypre_SeqVectorMigrate(hypre_Vector *x, HYPRE_MemoryLocation memory_location);
hypre_Vector *hypre_SeqVectorCloneDeep_v2(hypre_Vector *x, HYPRE_MemoryLocation memory_location);
hypre_Vector *hypre_SeqVectorCloneDeep(hypre_Vector *x);

/* ---- SpMV: y = alpha*A*x + beta*b  (b == y for the in-place forms) ----
 * Return value: informational ierr 1/2/3 on dimension mismatch, exactly as
 * seq_mv/csr_matvec.c:57-86 (the product is still formed). */
HYPRE_Int hypre_CSRMatrixMatvecOutOfPlace(HYPRE_Complex alpha, hypre_CSRMatrix *A, hypre_Vector *x,
                                          HYPRE_Complex beta, hypre_Vector *b, hypre_Vector *y,
                                          HYPRE_Int offset);
HYPRE_Int hypre_CSRMatrixMatvec(HYPRE_Complex alpha, hypre_CSRMatrix *A, hypre_Vector *x,
                                HYPRE_Complex beta, hypre_Vector *y);
HYPRE_Int hypre_CSRMatrixMatvecT(HYPRE_Complex alpha, hypre_CSRMatrix *A, hypre_Vector *x,
                                 HYPRE_Complex beta, hypre_Vector *y);
HYPRE_Int hypre_CSRMatrixMatvecDevice(HYPRE_Int trans, HYPRE_Complex alpha, hypre_CSRMatrix *A,
                                      hypre_Vector *x, HYPRE_Complex beta, hypre_Vector *b,
                                      hypre_Vector *y, HYPRE_Int offset);
HYPRE_Int hypre_CSRMatrixSpMVDevice(HYPRE_Int trans, HYPRE_Complex alpha, hypre_CSRMatrix *B,
                                    hypre_Vector *x, HYPRE_Complex beta, hypre_Vector *y,
                                    HYPRE_Int fill);

/* The row-binning plan and cached transpose the kernels use are kept in a
 * side table keyed by the matrix address (the struct layout is untouched).
 * Call after changing i/j/data in place or before freeing arrays that the
 * library does not own. */
HYPRE_Int hypre_amd_CSRMatrixInvalidatePlan(hypre_CSRMatrix *A);
/* What protects a caller who does NOT call it (the reference's hypre_CSRMatrixMatvec, seq_mv/csr_matvec.c:860-901, reads the
 * caller's arrays on every call; a plan caches things derived from them):
 *  - HYPRE_BoomerAMGSetup drops the plans of the matrix it is handed before anything else: a re-setup after
 *    HYPRE_IJMatrixSetValues on the same pattern works on the new values on every level;
 *  - the kernels that multiply by a private copy of the VALUES (value codes, slice form, fp32 copy) compare a rotating
 *    sample of the copy with the fp64 array in every launch — eight consecutive entries per wave at a position that moves
 *    with the plan's launch counter — so ANY coefficient edited in place is found within 64 (tiled kernel) or 128 (slice
 *    kernel) products of that matrix: the pinned flag is raised, the next call raises HYPRE_ERROR_GENERIC and rebuilds the
 *    plan, and a synchronous out-of-place product (or an in-place one with beta == 0) repeats itself with the fresh plan
 *    so that the caller reads the right result.  An in-place product with beta != 0 (hypre_CSRMatrixMatvec) cannot be
 *    repeated — its operand is already overwritten: the error is raised and y is NOT valid;
 *  - the column PATTERN is sampled at two fixed positions per tile (found out: another matrix on the same addresses);
 *  - hypre_amd_CSRMatrixVerifyPlan compares a 64-bit checksum of all three arrays with the one taken when the plan was
 *    built (one pass over the CSR arrays) and silently drops a plan that fails; HYPRE_ParCSRPCGSolve / GMRESSolve and a
 *    stand-alone HYPRE_BoomerAMGSolve (max_iter > 1) call it for the matrix they are handed, so a solve never STARTS from
 *    stale values.  Returns 1 (the plan stands, or there is none) or 0 (dropped). */
HYPRE_Int hypre_amd_CSRMatrixVerifyPlan(hypre_CSRMatrix *A);
/* The caller's promise that the arrays of the device matrix A stay as they are until hypre_amd_CSRMatrixInvalidatePlan,
 * hypre_amd_CSRMatrixSetImmutable(A, 0) or hypre_CSRMatrixDestroy: the plan may then keep a private copy of the fp64 values
 * in the layout that multiplies fastest (the row-slice form) and its launches carry no watch.  The library sets this for
 * the matrices it makes itself (hierarchy levels below the finest, interpolation / restriction operators, triangles,
 * colour classes).  Changing the promise drops the plan.  No reference counterpart. */
HYPRE_Int hypre_amd_CSRMatrixSetImmutable(hypre_CSRMatrix *A, HYPRE_Int on);
/* Mixed precision for products called directly (the AMG cycle sets it from its solver, hypre_amd_BoomerAMGSetMixedPrecision):
 * from now on the SpMV-class kernels stream an fp32 copy of the matrix values; vectors and sums stay fp64. */
HYPRE_Int hypre_amd_SetMixedPrecisionValues(HYPRE_Int on);
/* Which kernel form the plan of the device matrix A (built on demand) multiplies with: 0 a wave per row, 1 tiles with x
 * gathered through the cache, 2 tiles with x staged through LDS, 3 coded tiles, 4 slice form of a coded stencil, 5 row-slice
 * form; -1: not a device matrix. */
HYPRE_Int hypre_amd_CSRMatrixPlanForm(hypre_CSRMatrix *A);
/* Test hook: the nth allocation (1 = the next one) a plan builder makes at `site` fails, once — 1 tile tables, 2 x-staging
 * tables, 3 value codes, 4 slice form, 5 row-slice form; nth <= 0 disarms.  A plan is an accelerator: the builder frees
 * what the step had obtained, leaves no error behind and the matrix is multiplied one form lower.  Returns what was still
 * pending of the request before (0: that failure happened, or nothing was armed). */
HYPRE_Int hypre_amd_PlanTestFailAlloc(HYPRE_Int site, HYPRE_Int nth);
/* Columns ascending inside every row of a device matrix, in place (keep_first != 0: a row's first entry — the diagonal of a
 * square block, seq_mv/csr_matop.c:1536-1604 — stays in front); the plan is dropped.  A utility (the reference's
 * hypre_CSRMatrixSortRow, seq_mv/csr_matop.c, for device matrices); nothing on the solve path needs sorted rows. */
HYPRE_Int hypre_amd_CSRMatrixSortRows(hypre_CSRMatrix *A, HYPRE_Int keep_first);
/* Tile -> XCD placement policy of the plans built from now on (speed only; no reference
 * counterpart): enabled 0/1, min_tiles = smallest matrix, in 2048-entry tiles, that gets a
 * placement table, force != 0 also drops the minimum slab size so that small (test) matrices
 * take the code path of the benchmark sizes.  A negative argument leaves its field unchanged. */
HYPRE_Int hypre_amd_SpmvSetBandPolicy(HYPRE_Int enabled, HYPRE_Int min_tiles, HYPRE_Int force);
/* Kernel variant of the tiled SpMV family (speed only; results are the same bits): 2 (default) stages the x values a
 * tile needs through LDS from per-tile chunk lists kept in the plan, 0 gathers x through the cache.  Takes effect for
 * plans built afterwards (a plan built under 0 has no chunk lists and keeps gathering).  variant < 0: unchanged;
 * the second argument is unused. */
HYPRE_Int hypre_amd_SpmvSetVariant(HYPRE_Int variant, HYPRE_Int unused);
/* Value codes (speed only; results are the same bits; no reference counterpart): when a matrix holds at most 256 distinct
 * values — constant-coefficient stencils hold a handful — the x-staged kernel streams one byte per entry and looks the
 * value up in a table each tile keeps in LDS (1 + 2 bytes per entry instead of 8 + 2).  Found when the plan is built; on by
 * default (environment: HYPRE_AMD_SPMV_VALUE_CODES=0).  on < 0: unchanged.  Takes effect for plans built afterwards.
 * A caller that changes the values of a device matrix in place calls hypre_amd_CSRMatrixInvalidatePlan, as for the other
 * things a plan caches; one that does not is found out by the rotating value check (see hypre_amd_CSRMatrixInvalidatePlan). */
HYPRE_Int hypre_amd_SpmvSetValueCodes(HYPRE_Int on);
/* Slice form (speed only; no reference counterpart): a coded matrix whose rows hold at most 32 entries and are about equally
 * long (a stencil) is stored in the plan once more, a row's codes and local indices in one lane's words, and multiplied by a
 * kernel in which every lane sums its row from registers — no products parked in LDS, no reduction.  Rows of at most 8
 * entries give the same bits as the tiled kernel; longer ones are summed as two interleaved partial sums (what the tiled
 * kernel does where a tile holds 65 to 128 rows), within an ulp or two of the row's absolute sum.  On by default (environment:
 * HYPRE_AMD_SPMV_SLICE_FORM=0); on < 0: unchanged; takes effect for plans built afterwards. */
HYPRE_Int hypre_amd_SpmvSetSliceForm(HYPRE_Int on);
/* Row-slice form (speed only; no reference counterpart — the reference multiplies every matrix with K lanes per row and a
 * shuffle tree, seq_mv/csr_spmv_device.c:149-260): a matrix that cannot change behind its plan (made by the library, or
 * declared by hypre_amd_CSRMatrixSetImmutable) and is not coded is stored in the plan as jagged slices — 256 / W rows per
 * workgroup, W lanes per row, the entries of the 64 lane-tasks of a wave side by side with no padding: fp64 value + 16-bit
 * staged position per entry — and multiplied by a kernel in which a lane sums its entries in stored order from registers and
 * the W partial sums of a row are added in lane order: within a few ulps of the row's absolute sum of any other summation
 * order.  mode 0 off, 1 owned / immutable matrices (default; environment HYPRE_AMD_SPMV_ROW_SLICES), 2 every matrix (the
 * caller then owes hypre_amd_CSRMatrixInvalidatePlan after any change of the arrays); < 0 unchanged.  Plans built afterwards. */
HYPRE_Int hypre_amd_SpmvSetRowSlices(HYPRE_Int mode);
/* Products with a multivector (num_vectors > 1, stored column by column): one pass over the matrix for up to four
 * columns at a time where the matrix's plan has the x-staged form (default), or one pass per column (on = 0).  Speed
 * only: the same bits.  Replaces the NV = 2, 3, 4 paths of seq_mv/csr_matvec.c:117-380 and the NV sums per row of
 * seq_mv/csr_spmv_device.c:37-134. */
HYPRE_Int hypre_amd_SpmvSetFusedMultivectors(HYPRE_Int on);
/* Launches of the fused multivector kernel since the library was loaded (tests: which path served a product). */
HYPRE_Int hypre_amd_SpmvFusedMultivectorLaunches(void);
/* Lanes per row of the row-slice form in the plan of the device matrix A (0: none); fills the rows of a block and the most
 * entries a lane holds. */
HYPRE_Int hypre_amd_CSRMatrixPlanRowSlices(hypre_CSRMatrix *A, HYPRE_Int *rows_per_block, HYPRE_Int *entries_per_lane);
/* Lanes per row (1 or 2) of the slice form in the plan of the device matrix A; 0 when it has none. */
HYPRE_Int hypre_amd_CSRMatrixPlanSliceForm(hypre_CSRMatrix *A);
/* Number of distinct values in the value table of the plan of the device matrix A; 0 when A is not coded. */
HYPRE_Int hypre_amd_CSRMatrixPlanValueCodes(hypre_CSRMatrix *A);
/* x staging of the plan of the device matrix A: returns the number of tiles that take the LDS-staged path of the tiled
 * kernel; fills the plan's tile count and the mean number of x pieces of a staged tile. */
HYPRE_Int hypre_amd_CSRMatrixPlanStaging(hypre_CSRMatrix *A, HYPRE_Int *num_tiles, HYPRE_Real *mean_pieces);
/* Returns 1 when the plan of the device matrix A carries a placement table; fills the tile
 * count of the plan and the band distance the table was built for (0: none). */
HYPRE_Int hypre_amd_CSRMatrixPlanInfo(hypre_CSRMatrix *A, HYPRE_Int *num_tiles, HYPRE_Int *band);

/* ---- BLAS-1 (seq_mv/vector.c dispatchers + vector_device.c bodies) ---- */
HYPRE_Int hypre_SeqVectorSetConstantValues(hypre_Vector *v, HYPRE_Complex value);
HYPRE_Int hypre_SeqVectorCopy(hypre_Vector *x, hypre_Vector *y);
HYPRE_Int hypre_SeqVectorScale(HYPRE_Complex alpha, hypre_Vector *y);
HYPRE_Int hypre_SeqVectorAxpy(HYPRE_Complex alpha, hypre_Vector *x, hypre_Vector *y);
HYPRE_Int hypre_SeqVectorAxpyz(HYPRE_Complex alpha, hypre_Vector *x, HYPRE_Complex beta,
                               hypre_Vector *y, hypre_Vector *z);
HYPRE_Real hypre_SeqVectorInnerProd(hypre_Vector *x, hypre_Vector *y);
HYPRE_Int hypre_SeqVectorElmdivpy(hypre_Vector *x, hypre_Vector *b, hypre_Vector *y);
HYPRE_Int hypre_SeqVectorElmdivpyMarked(hypre_Vector *x, hypre_Vector *b, hypre_Vector *y,
                                        HYPRE_Int *marker, HYPRE_Int marker_val);
HYPRE_Int hypre_SeqVectorSetConstantValuesDevice(hypre_Vector *v, HYPRE_Complex value);
HYPRE_Int hypre_SeqVectorScaleDevice(HYPRE_Complex alpha, hypre_Vector *y);
HYPRE_Int hypre_SeqVectorAxpyDevice(HYPRE_Complex alpha, hypre_Vector *x, hypre_Vector *y);
HYPRE_Int hypre_SeqVectorAxpyzDevice(HYPRE_Complex alpha, hypre_Vector *x, HYPRE_Complex beta,
                                     hypre_Vector *y, hypre_Vector *z);
HYPRE_Real hypre_SeqVectorInnerProdDevice(hypre_Vector *x, hypre_Vector *y);
HYPRE_Int hypre_SeqVectorElmdivpyDevice(hypre_Vector *x, hypre_Vector *b, hypre_Vector *y,
                                        HYPRE_Int *marker, HYPRE_Int marker_val);
HYPRE_Int hypreDevice_IVAXPY(HYPRE_Int n, HYPRE_Complex *a, HYPRE_Complex *x, HYPRE_Complex *y);
HYPRE_Int hypreDevice_IVAXPYMarked(HYPRE_Int n, HYPRE_Complex *a, HYPRE_Complex *x, HYPRE_Complex *y,
                                   HYPRE_Int *marker, HYPRE_Int marker_val);
HYPRE_Int hypreDevice_DiagScaleVector2(HYPRE_Int num_vectors, HYPRE_Int num_rows,
                                       HYPRE_Complex *diag, HYPRE_Complex *x, HYPRE_Complex beta,
                                       HYPRE_Complex *y, HYPRE_Complex *z, HYPRE_Int computeY);

#ifdef __cplusplus
}
#endif
#endif
