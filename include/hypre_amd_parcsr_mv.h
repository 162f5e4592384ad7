/*
 * hypre_amd — distributed (one rank per GPU) CSR matrix, vector, halo-exchange
 * package and the ParCSR matrix-vector products.
 *
 * Struct layouts follow the reference in the configuration fixed in
 * HYPRE_amd_utilities.h (HYPRE_USING_GPU members present, persistent-comm
 * members absent, MPI_Comm an integer):
 *   parcsr_mv/par_csr_matrix.h:27-86          hypre_ParCSRMatrix
 *   parcsr_mv/par_vector.h:25-45              hypre_ParVector
 *   parcsr_mv/par_csr_communication.h:34-75   hypre_ParCSRCommHandle / CommPkg
 * Functions replace:
 *   parcsr_mv/par_csr_matvec.c:241,274,523            Matvec[OutOfPlace|T] dispatchers
 *   parcsr_mv/par_csr_matvec_device.c:25,277          their *Device bodies
 *   parcsr_mv/par_csr_communication.c:358-699         CommHandleCreate_v2 / Destroy
 *   parcsr_mv/par_csr_communication.c:713-943,1163    CommPkgCreate_core / MatvecCommPkgCreate
 *   parcsr_mv/par_vector.c:322-575                    ParVector BLAS-1 wrappers
 */
#ifndef HYPRE_AMD_PARCSR_MV_H
#define HYPRE_AMD_PARCSR_MV_H

#include "hypre_amd_seq_mv.h"
#include "hypre_amd_comm.h"

#ifdef __cplusplus
extern "C" {
#endif

struct _hypre_ParCSRCommPkg;

typedef struct
{
   struct _hypre_ParCSRCommPkg *comm_pkg;
   HYPRE_MemoryLocation  send_memory_location;
   HYPRE_MemoryLocation  recv_memory_location;
   HYPRE_Int             num_send_bytes;
   HYPRE_Int             num_recv_bytes;
   void                 *send_data;
   void                 *recv_data;
   void                 *send_data_buffer;
   void                 *recv_data_buffer;
   HYPRE_Int             num_requests;
   void                 *requests;     /* HIP event recorded after the exchange was enqueued */
} hypre_ParCSRCommHandle;

typedef struct _hypre_ParCSRCommPkg
{
   MPI_Comm          comm;
   HYPRE_Int         num_components;
   HYPRE_Int         num_sends;
   HYPRE_Int        *send_procs;
   HYPRE_Int        *send_map_starts;
   HYPRE_Int        *send_map_elmts;          /* local row ids to gather, host */
   HYPRE_Int        *device_send_map_elmts;   /* same list in device memory */
   HYPRE_Int         num_recvs;
   HYPRE_Int        *recv_procs;
   HYPRE_Int        *recv_vec_starts;         /* offsets into the ghost vector */
   void             *send_mpi_types;
   void             *recv_mpi_types;
   /* device work space, allocated once per package */
   HYPRE_Complex    *tmp_data;                /* ghost vector (num_cols_offd)   */
   HYPRE_Complex    *buf_data;                /* packed send buffer              */
   hypre_CSRMatrix  *matrix_E;                /* MatvecT unpack operator: y += E * buf (device, built on demand) */
} hypre_ParCSRCommPkg;

#define hypre_ParCSRCommPkgComm(comm_pkg)               (comm_pkg -> comm)
#define hypre_ParCSRCommPkgNumSends(comm_pkg)           (comm_pkg -> num_sends)
#define hypre_ParCSRCommPkgSendProcs(comm_pkg)          (comm_pkg -> send_procs)
#define hypre_ParCSRCommPkgSendProc(comm_pkg, i)        (comm_pkg -> send_procs[i])
#define hypre_ParCSRCommPkgSendMapStarts(comm_pkg)      (comm_pkg -> send_map_starts)
#define hypre_ParCSRCommPkgSendMapStart(comm_pkg,i)     (comm_pkg -> send_map_starts[i])
#define hypre_ParCSRCommPkgSendMapElmts(comm_pkg)       (comm_pkg -> send_map_elmts)
#define hypre_ParCSRCommPkgDeviceSendMapElmts(comm_pkg) (comm_pkg -> device_send_map_elmts)
#define hypre_ParCSRCommPkgSendMapElmt(comm_pkg,i)      (comm_pkg -> send_map_elmts[i])
#define hypre_ParCSRCommPkgNumRecvs(comm_pkg)           (comm_pkg -> num_recvs)
#define hypre_ParCSRCommPkgRecvProcs(comm_pkg)          (comm_pkg -> recv_procs)
#define hypre_ParCSRCommPkgRecvProc(comm_pkg, i)        (comm_pkg -> recv_procs[i])
#define hypre_ParCSRCommPkgRecvVecStarts(comm_pkg)      (comm_pkg -> recv_vec_starts)
#define hypre_ParCSRCommPkgRecvVecStart(comm_pkg,i)     (comm_pkg -> recv_vec_starts[i])
#define hypre_ParCSRCommPkgTmpData(comm_pkg)            (comm_pkg -> tmp_data)
#define hypre_ParCSRCommPkgBufData(comm_pkg)            (comm_pkg -> buf_data)

typedef struct hypre_ParCSRMatrix_struct
{
   MPI_Comm              comm;
   HYPRE_BigInt          global_num_rows;
   HYPRE_BigInt          global_num_cols;
   HYPRE_BigInt          global_num_rownnz;
   HYPRE_BigInt          num_nonzeros;
   HYPRE_Real            d_num_nonzeros;
   HYPRE_BigInt          first_row_index;
   HYPRE_BigInt          first_col_diag;
   HYPRE_BigInt          last_row_index;
   HYPRE_BigInt          last_col_diag;
   hypre_CSRMatrix      *diag;                /* local columns; diagonal entry first in every row */
   hypre_CSRMatrix      *offd;                /* columns = compressed ghost ids */
   hypre_CSRMatrix      *diagT, *offdT;       /* optional stored transposes (keepTranspose) */
   HYPRE_BigInt         *col_map_offd;        /* ghost id -> global column, ascending, host */
   HYPRE_BigInt         *device_col_map_offd;
   HYPRE_BigInt          row_starts[2];
   HYPRE_BigInt          col_starts[2];
   hypre_ParCSRCommPkg  *comm_pkg;
   hypre_ParCSRCommPkg  *comm_pkgT;
   HYPRE_Int             owns_data;
   HYPRE_BigInt         *rowindices;
   HYPRE_Complex        *rowvalues;
   HYPRE_Int             getrowactive;
   void                 *assumed_partition;
   HYPRE_Int             owns_assumed_partition;
   HYPRE_Int            *proc_ordering;
   HYPRE_Int             bdiag_size;
   HYPRE_Complex        *bdiaginv;
   hypre_ParCSRCommPkg  *bdiaginv_comm_pkg;
   HYPRE_Int            *soc_diag_j;
   HYPRE_Int            *soc_offd_j;
} hypre_ParCSRMatrix;

#define hypre_ParCSRMatrixComm(matrix)             ((matrix) -> comm)
#define hypre_ParCSRMatrixGlobalNumRows(matrix)    ((matrix) -> global_num_rows)
#define hypre_ParCSRMatrixGlobalNumCols(matrix)    ((matrix) -> global_num_cols)
#define hypre_ParCSRMatrixNumNonzeros(matrix)      ((matrix) -> num_nonzeros)
#define hypre_ParCSRMatrixDNumNonzeros(matrix)     ((matrix) -> d_num_nonzeros)
#define hypre_ParCSRMatrixFirstRowIndex(matrix)    ((matrix) -> first_row_index)
#define hypre_ParCSRMatrixFirstColDiag(matrix)     ((matrix) -> first_col_diag)
#define hypre_ParCSRMatrixLastRowIndex(matrix)     ((matrix) -> last_row_index)
#define hypre_ParCSRMatrixLastColDiag(matrix)      ((matrix) -> last_col_diag)
#define hypre_ParCSRMatrixDiag(matrix)             ((matrix) -> diag)
#define hypre_ParCSRMatrixOffd(matrix)             ((matrix) -> offd)
#define hypre_ParCSRMatrixDiagT(matrix)            ((matrix) -> diagT)
#define hypre_ParCSRMatrixOffdT(matrix)            ((matrix) -> offdT)
#define hypre_ParCSRMatrixColMapOffd(matrix)       ((matrix) -> col_map_offd)
#define hypre_ParCSRMatrixRowStarts(matrix)        ((matrix) -> row_starts)
#define hypre_ParCSRMatrixColStarts(matrix)        ((matrix) -> col_starts)
#define hypre_ParCSRMatrixCommPkg(matrix)          ((matrix) -> comm_pkg)
#define hypre_ParCSRMatrixCommPkgT(matrix)         ((matrix) -> comm_pkgT)
#define hypre_ParCSRMatrixOwnsData(matrix)         ((matrix) -> owns_data)
#define hypre_ParCSRMatrixNumRows(matrix)          hypre_CSRMatrixNumRows(hypre_ParCSRMatrixDiag(matrix))
#define hypre_ParCSRMatrixNumCols(matrix)          hypre_CSRMatrixNumCols(hypre_ParCSRMatrixDiag(matrix))
#define hypre_ParCSRMatrixMemoryLocation(matrix)   hypre_CSRMatrixMemoryLocation(hypre_ParCSRMatrixDiag(matrix))

typedef struct hypre_ParVector_struct
{
   MPI_Comm       comm;
   HYPRE_BigInt   global_size;
   HYPRE_BigInt   first_index;
   HYPRE_BigInt   last_index;
   HYPRE_BigInt   partitioning[2];
   HYPRE_Int      actual_local_size;   /* allocated length; work vectors are re-sized in place */
   hypre_Vector  *local_vector;
   HYPRE_Int      owns_data;
   HYPRE_Int      all_zeros;           /* set by SetZeros; lets a Jacobi sweep skip its SpMV */
   void          *assumed_partition;
} hypre_ParVector;

#define hypre_ParVectorComm(vector)             ((vector) -> comm)
#define hypre_ParVectorGlobalSize(vector)       ((vector) -> global_size)
#define hypre_ParVectorFirstIndex(vector)       ((vector) -> first_index)
#define hypre_ParVectorLastIndex(vector)        ((vector) -> last_index)
#define hypre_ParVectorPartitioning(vector)     ((vector) -> partitioning)
#define hypre_ParVectorActualLocalSize(vector)  ((vector) -> actual_local_size)
#define hypre_ParVectorLocalVector(vector)      ((vector) -> local_vector)
#define hypre_ParVectorOwnsData(vector)         ((vector) -> owns_data)
#define hypre_ParVectorAllZeros(vector)         ((vector) -> all_zeros)
#define hypre_ParVectorNumVectors(vector)       (hypre_VectorNumVectors(hypre_ParVectorLocalVector(vector)))
#define hypre_ParVectorMemoryLocation(vector)   hypre_VectorMemoryLocation(hypre_ParVectorLocalVector(vector))

typedef hypre_ParCSRMatrix *HYPRE_ParCSRMatrix;
typedef hypre_ParVector    *HYPRE_ParVector;

/* ---- objects ---- */
hypre_ParCSRMatrix *hypre_ParCSRMatrixCreate(MPI_Comm comm, HYPRE_BigInt global_num_rows,
                                             HYPRE_BigInt global_num_cols, HYPRE_BigInt *row_starts_in,
                                             HYPRE_BigInt *col_starts_in, HYPRE_Int num_cols_offd,
                                             HYPRE_Int num_nonzeros_diag, HYPRE_Int num_nonzeros_offd);
HYPRE_Int hypre_ParCSRMatrixInitialize_v2(hypre_ParCSRMatrix *matrix, HYPRE_MemoryLocation memory_location);
HYPRE_Int hypre_ParCSRMatrixDestroy(hypre_ParCSRMatrix *matrix);
HYPRE_Int hypre_ParCSRMatrixMigrate(hypre_ParCSRMatrix *A, HYPRE_MemoryLocation memory_location);
hypre_ParCSRMatrix *hypre_ParCSRMatrixClone_v2(hypre_ParCSRMatrix *A, HYPRE_Int copy_data,
                                               HYPRE_MemoryLocation memory_location);
HYPRE_Int hypre_ParCSRMatrixSetNumNonzeros(hypre_ParCSRMatrix *matrix);
HYPRE_Int hypre_ParCSRMatrixSetDNumNonzeros(hypre_ParCSRMatrix *matrix);
/* build and cache local transposes of diag and offd (par_csr_triplemat.c:364-372 keepTranspose) */
HYPRE_Int hypre_amd_ParCSRMatrixKeepTranspose(hypre_ParCSRMatrix *A);

hypre_ParVector *hypre_ParVectorCreate(MPI_Comm comm, HYPRE_BigInt global_size, HYPRE_BigInt *partitioning_in);
/* parcsr_mv/par_vector.c:77-87 (global_size: the global length of one column; columns stored one after the other) */
hypre_ParVector *hypre_ParMultiVectorCreate(MPI_Comm comm, HYPRE_BigInt global_size, HYPRE_BigInt *partitioning_in,
                                            HYPRE_Int num_vectors);
HYPRE_Int hypre_ParVectorInitialize_v2(hypre_ParVector *vector, HYPRE_MemoryLocation memory_location);
/* x = y ./ diag(A), every column of a multivector (parcsr_mv/par_csr_matop.c:6479-6658) */
HYPRE_Int hypre_ParCSRDiagScaleVector(hypre_ParCSRMatrix *par_A, hypre_ParVector *par_y, hypre_ParVector *par_x);
HYPRE_Int hypre_ParVectorInitialize(hypre_ParVector *vector);
HYPRE_Int hypre_ParVectorDestroy(hypre_ParVector *vector);
HYPRE_Int hypre_ParVectorSetLocalSize(hypre_ParVector *vector, HYPRE_Int local_size);
HYPRE_Int hypre_ParVectorMigrate(hypre_ParVector *x, HYPRE_MemoryLocation memory_location);

/* ---- halo exchange ---- */
HYPRE_Int hypre_MatvecCommPkgCreate(hypre_ParCSRMatrix *A);
HYPRE_Int hypre_MatvecCommPkgDestroy(hypre_ParCSRCommPkg *comm_pkg);
/* par_csr_communication.c:1054-1154: switch the package to multivectors of num_components_in columns (one exchange then
 * carries every column) or back to single vectors */
HYPRE_Int hypre_ParCSRCommPkgUpdateVecStarts(hypre_ParCSRCommPkg *comm_pkg, HYPRE_Int num_components_in,
                                             HYPRE_Int vecstride, HYPRE_Int idxstride);
HYPRE_Int hypre_ParCSRCommPkgCreate_core(MPI_Comm comm, HYPRE_BigInt *col_map_offd,
                                         HYPRE_BigInt first_col_diag, HYPRE_BigInt *col_starts,
                                         HYPRE_Int num_cols_diag, HYPRE_Int num_cols_offd,
                                         HYPRE_Int *p_num_recvs, HYPRE_Int **p_recv_procs,
                                         HYPRE_Int **p_recv_vec_starts, HYPRE_Int *p_num_sends,
                                         HYPRE_Int **p_send_procs, HYPRE_Int **p_send_map_starts,
                                         HYPRE_Int **p_send_map_elmts);
/* job 1: owner -> ghost (x halo of the SpMV); job 2: ghost -> owner (transpose
 * product); 11/12: the same for HYPRE_Int payloads.  Buffers may live in host
 * or device memory (both must agree with the communicator's capabilities). */
hypre_ParCSRCommHandle *hypre_ParCSRCommHandleCreate_v2(HYPRE_Int job, hypre_ParCSRCommPkg *comm_pkg,
                                                        HYPRE_MemoryLocation send_memory_location,
                                                        void *send_data,
                                                        HYPRE_MemoryLocation recv_memory_location,
                                                        void *recv_data);
hypre_ParCSRCommHandle *hypre_ParCSRCommHandleCreate(HYPRE_Int job, hypre_ParCSRCommPkg *comm_pkg,
                                                     void *send_data, void *recv_data);
HYPRE_Int hypre_ParCSRCommHandleDestroy(hypre_ParCSRCommHandle *comm_handle);

/* ---- ParCSR SpMV ---- (return: informational ierr 11/12/13, par_csr_matvec.c:57-82) */
HYPRE_Int hypre_ParCSRMatrixMatvecOutOfPlace(HYPRE_Complex alpha, hypre_ParCSRMatrix *A, hypre_ParVector *x,
                                             HYPRE_Complex beta, hypre_ParVector *b, hypre_ParVector *y);
HYPRE_Int hypre_ParCSRMatrixMatvec(HYPRE_Complex alpha, hypre_ParCSRMatrix *A, hypre_ParVector *x,
                                   HYPRE_Complex beta, hypre_ParVector *y);
HYPRE_Int hypre_ParCSRMatrixMatvecT(HYPRE_Complex alpha, hypre_ParCSRMatrix *A, hypre_ParVector *x,
                                    HYPRE_Complex beta, hypre_ParVector *y);
HYPRE_Int hypre_ParCSRMatrixMatvecOutOfPlaceDevice(HYPRE_Complex alpha, hypre_ParCSRMatrix *A,
                                                   hypre_ParVector *x, HYPRE_Complex beta,
                                                   hypre_ParVector *b, hypre_ParVector *y);
HYPRE_Int hypre_ParCSRMatrixMatvecTDevice(HYPRE_Complex alpha, hypre_ParCSRMatrix *A, hypre_ParVector *x,
                                          HYPRE_Complex beta, hypre_ParVector *y);
HYPRE_Int HYPRE_ParCSRMatrixMatvec(HYPRE_Complex alpha, HYPRE_ParCSRMatrix A, HYPRE_ParVector x,
                                   HYPRE_Complex beta, HYPRE_ParVector y);

/* ---- ParVector BLAS-1 ---- */
HYPRE_Int  hypre_ParVectorSetConstantValues(hypre_ParVector *v, HYPRE_Complex value);
HYPRE_Int  hypre_ParVectorSetZeros(hypre_ParVector *v);
HYPRE_Int  hypre_ParVectorCopy(hypre_ParVector *x, hypre_ParVector *y);
HYPRE_Int  hypre_ParVectorScale(HYPRE_Complex alpha, hypre_ParVector *y);
HYPRE_Int  hypre_ParVectorAxpy(HYPRE_Complex alpha, hypre_ParVector *x, hypre_ParVector *y);
HYPRE_Int  hypre_ParVectorAxpyz(HYPRE_Complex alpha, hypre_ParVector *x, HYPRE_Complex beta,
                                hypre_ParVector *y, hypre_ParVector *z);
HYPRE_Real hypre_ParVectorInnerProd(hypre_ParVector *x, hypre_ParVector *y);
HYPRE_Int  hypre_ParVectorElmdivpy(hypre_ParVector *x, hypre_ParVector *b, hypre_ParVector *y);
HYPRE_Int  hypre_ParVectorElmdivpyMarked(hypre_ParVector *x, hypre_ParVector *b, hypre_ParVector *y,
                                         HYPRE_Int *marker, HYPRE_Int marker_val);

/* ---- synthetic problem generators (inputs of the benchmark configurations) ----
 * parcsr_ls/par_laplace.c:15-340, par_laplace_27pt.c, par_difconv.c; rank
 * (p,q,r) of a P x Q x R box decomposition builds its own block. */
HYPRE_ParCSRMatrix GenerateLaplacian(MPI_Comm comm, HYPRE_BigInt nx, HYPRE_BigInt ny, HYPRE_BigInt nz,
                                     HYPRE_Int P, HYPRE_Int Q, HYPRE_Int R, HYPRE_Int p, HYPRE_Int q,
                                     HYPRE_Int r, HYPRE_Real *value);
HYPRE_ParCSRMatrix GenerateLaplacian27pt(MPI_Comm comm, HYPRE_BigInt nx, HYPRE_BigInt ny, HYPRE_BigInt nz,
                                         HYPRE_Int P, HYPRE_Int Q, HYPRE_Int R, HYPRE_Int p, HYPRE_Int q,
                                         HYPRE_Int r, HYPRE_Real *value);
HYPRE_ParCSRMatrix GenerateDifConv(MPI_Comm comm, HYPRE_BigInt nx, HYPRE_BigInt ny, HYPRE_BigInt nz,
                                   HYPRE_Int P, HYPRE_Int Q, HYPRE_Int R, HYPRE_Int p, HYPRE_Int q,
                                   HYPRE_Int r, HYPRE_Real *value);

/* par_vardifconv.c:15-565 (`ij -vardifconv -eps e`): variable-coefficient diffusion; *rhs_ptr (optional) receives the
 * right-hand side the reference generator returns with its coefficient functions (all ones) */
HYPRE_ParCSRMatrix GenerateVarDifConv(MPI_Comm comm, HYPRE_BigInt nx, HYPRE_BigInt ny, HYPRE_BigInt nz, HYPRE_Int P,
                                      HYPRE_Int Q, HYPRE_Int R, HYPRE_Int p, HYPRE_Int q, HYPRE_Int r, HYPRE_Real eps,
                                      HYPRE_ParVector *rhs_ptr);
/* par_rotate_7pt.c:15-397 (`ij -rotate -alpha a -eps e -n nx ny 1 -P P Q 1`) */
HYPRE_ParCSRMatrix GenerateRotate7pt(MPI_Comm comm, HYPRE_BigInt nx, HYPRE_BigInt ny, HYPRE_Int P, HYPRE_Int Q,
                                     HYPRE_Int p, HYPRE_Int q, HYPRE_Real alpha, HYPRE_Real eps);
/* par_laplace.c:380-848: num_fun unknowns per grid point, A = (7-point operator) (x) mtrx[num_fun x num_fun]
 * (`ij -sysL num_fun`; the driver's coupling matrices are in test/ij.c:9718-9870) */
HYPRE_ParCSRMatrix GenerateSysLaplacian(MPI_Comm comm, HYPRE_BigInt nx, HYPRE_BigInt ny, HYPRE_BigInt nz,
                                        HYPRE_Int P, HYPRE_Int Q, HYPRE_Int R, HYPRE_Int p, HYPRE_Int q,
                                        HYPRE_Int r, HYPRE_Int num_fun, HYPRE_Real *mtrx, HYPRE_Real *value);

/* ---- binding conveniences (plain pointers in / out; used by the Python host
 * layer and the tests, the way an application would use the IJ interface) ---- */
hypre_CSRMatrix *hypre_amd_CSRMatrixFromArrays(HYPRE_Int num_rows, HYPRE_Int num_cols, HYPRE_Int nnz,
                                               const HYPRE_Int *i, const HYPRE_Int *j,
                                               const HYPRE_Complex *data, HYPRE_MemoryLocation location);
hypre_Vector    *hypre_amd_SeqVectorFromArray(HYPRE_Int size, const HYPRE_Complex *data,
                                              HYPRE_MemoryLocation location);
HYPRE_Int        hypre_amd_SeqVectorToArray(hypre_Vector *v, HYPRE_Complex *out);
/* shape of a matrix's halo exchange (its communication package, built on demand — collective): neighbours this rank sends
 * to / receives from and the entries per exchange (par_csr_communication.h:51-75 send_map_starts / recv_vec_starts) */
HYPRE_Int        hypre_amd_ParCSRMatrixHaloInfo(hypre_ParCSRMatrix *A, HYPRE_Int *num_sends, HYPRE_Int *send_entries,
                                                HYPRE_Int *num_recvs, HYPRE_Int *recv_entries);
HYPRE_Int        hypre_amd_CopyToHost(void *dst_host, const void *src, size_t bytes,
                                      HYPRE_MemoryLocation src_location);
/* assemble one rank's block from local diag/offd CSR arrays (host pointers) */
hypre_ParCSRMatrix *hypre_amd_ParCSRMatrixFromArrays(MPI_Comm comm, HYPRE_BigInt global_num_rows,
                                                     HYPRE_BigInt global_num_cols,
                                                     const HYPRE_BigInt *row_starts,
                                                     const HYPRE_BigInt *col_starts,
                                                     HYPRE_Int num_cols_offd, const HYPRE_BigInt *col_map_offd,
                                                     const HYPRE_Int *diag_i, const HYPRE_Int *diag_j,
                                                     const HYPRE_Complex *diag_data,
                                                     const HYPRE_Int *offd_i, const HYPRE_Int *offd_j,
                                                     const HYPRE_Complex *offd_data,
                                                     HYPRE_MemoryLocation location);
hypre_ParVector *hypre_amd_ParVectorFromArray(MPI_Comm comm, HYPRE_BigInt global_size,
                                              const HYPRE_BigInt *partitioning, const HYPRE_Complex *data,
                                              HYPRE_MemoryLocation location);
HYPRE_Int hypre_amd_ParVectorToArray(hypre_ParVector *v, HYPRE_Complex *out);

#ifdef __cplusplus
}
#endif
#endif
