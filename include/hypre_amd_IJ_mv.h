/* hypre_amd — the file-format slice of hypre's IJ interface.
 *
 * The reference's regression inputs (test/TEST_ij: A.0000N, test.A.0000N, data/tucker21935/IJ.A.0000N ...) are
 * text files in hypre's IJ format, one per rank:
 *
 *    matrix  <name>.<rank %05d>:  "ilower iupper jlower jupper"  then  "I J value" per entry
 *    vector  <name>.<rank %05d>:  "jlower jupper"                then  "j value"   per entry
 *
 * (written by hypre_ParCSRMatrixPrintIJ, parcsr_mv/par_csr_matrix.c:888-1047, and
 * HYPRE_IJVectorPrint, IJ_mv/HYPRE_IJVector.c:718-782; read by hypre_IJMatrixRead,
 * IJ_mv/IJMatrix.c:112-247, and HYPRE_IJVectorRead, HYPRE_IJVector.c:641-699).
 * This header carries the read / print / get-object / destroy entry points of the
 * IJ interface so that `ij -fromfile A -rhsfromfile b` lines can be replayed.  As in
 * the reference, an entry of a row (vector index) another rank owns is added to that
 * rank's value at assembly (test/TEST_ij/A_tstoffd.* exercises this).  The programmatic
 * assembly interface (SetValues / AddToValues calls) is not part of the solve path and
 * is not provided.
 *
 * Struct layouts follow IJ_mv/IJ_matrix.h:20-45 and IJ_mv/IJ_vector.h:20-35.
 */
#ifndef HYPRE_AMD_IJ_MV_H
#define HYPRE_AMD_IJ_MV_H

#include "hypre_amd_parcsr_mv.h"

#ifdef __cplusplus
extern "C" {
#endif

#define HYPRE_PARCSR       5555     /* HYPRE_utilities.h / HYPRE_IJ_mv.h object type */
#define HYPRE_UNITIALIZED  -999

typedef struct hypre_IJMatrix_struct
{
   MPI_Comm      comm;
   HYPRE_BigInt  row_partitioning[2];
   HYPRE_BigInt  col_partitioning[2];
   HYPRE_Int     object_type;
   void         *object;              /* hypre_ParCSRMatrix* */
   void         *translator;          /* unused here (the reference's auxiliary assembly matrix) */
   void         *assumed_part;
   HYPRE_Int     assemble_flag;
   HYPRE_BigInt  global_first_row;    /* indices in the file are relative to these */
   HYPRE_BigInt  global_first_col;
   HYPRE_BigInt  global_num_rows;
   HYPRE_BigInt  global_num_cols;
   HYPRE_Int     omp_flag;
   HYPRE_Int     print_level;
} hypre_IJMatrix;

typedef struct hypre_IJVector_struct
{
   MPI_Comm      comm;
   HYPRE_BigInt  partitioning[2];
   HYPRE_Int     num_components;
   HYPRE_Int     object_type;
   void         *object;              /* hypre_ParVector* */
   void         *translator;
   void         *assumed_part;
   HYPRE_BigInt  global_first_row;
   HYPRE_BigInt  global_num_rows;
   HYPRE_Int     print_level;
} hypre_IJVector;

typedef hypre_IJMatrix *HYPRE_IJMatrix;
typedef hypre_IJVector *HYPRE_IJVector;

/* IJ_mv/HYPRE_IJMatrix.c:1267 -> IJMatrix.c:112.  Collective.  The matrix is assembled in host
 * memory the way hypre_IJMatrixAssembleParCSR does (IJMatrix_parcsr.c:2690-2950): entries keep their
 * file order, the entry whose local column equals the local row moves to the front of its row, ghost
 * columns are compressed through a sorted col_map_offd, a repeated (I, J) overwrites the earlier value
 * (SetValues, IJMatrix_parcsr.c:868-880).  Errors: HYPRE_ERROR_ARG(1) when the file cannot be opened,
 * HYPRE_ERROR_GENERIC "Error in IJ matrix input file." on a malformed line. */
HYPRE_Int HYPRE_IJMatrixRead(const char *filename, MPI_Comm comm, HYPRE_Int type, HYPRE_IJMatrix *matrix_ptr);
/* HYPRE_IJMatrix.c:1318 -> par_csr_matrix.c:888 with base 0 (indices relative to the global first row /
 * column, "%.14e" values, diag block entries before ghost entries in every row) */
HYPRE_Int HYPRE_IJMatrixPrint(HYPRE_IJMatrix matrix, const char *filename);
HYPRE_Int HYPRE_IJMatrixGetObject(HYPRE_IJMatrix matrix, void **object);          /* HYPRE_IJMatrix.c:1083 */
HYPRE_Int HYPRE_IJMatrixDestroy(HYPRE_IJMatrix matrix);                           /* HYPRE_IJMatrix.c:184 */
HYPRE_Int hypre_ParCSRMatrixPrintIJ(const hypre_ParCSRMatrix *matrix, const HYPRE_Int base_i,
                                    const HYPRE_Int base_j, const char *filename);  /* par_csr_matrix.c:888 */

/* HYPRE_IJVector.c:641, :718, :603, :122 */
HYPRE_Int HYPRE_IJVectorRead(const char *filename, MPI_Comm comm, HYPRE_Int type, HYPRE_IJVector *vector_ptr);
HYPRE_Int HYPRE_IJVectorPrint(HYPRE_IJVector vector, const char *filename);
HYPRE_Int HYPRE_IJVectorGetObject(HYPRE_IJVector vector, void **object);
HYPRE_Int HYPRE_IJVectorDestroy(HYPRE_IJVector vector);

/* wrap existing objects so that they can be printed through the IJ entry points (the object is
 * borrowed: Destroy of the wrapper leaves it alone) */
HYPRE_IJMatrix hypre_amd_IJMatrixWrap(hypre_ParCSRMatrix *A);
HYPRE_IJVector hypre_amd_IJVectorWrap(hypre_ParVector *v);

#ifdef __cplusplus
}
#endif
#endif
