/*
 * hypre_amd — MI355X-native BoomerAMG solve phase.
 *
 * Scalar types, enums and the error convention of the drop-in boundary.
 * These restate (they do not include) the reference's public utilities
 * interface so that the library can be built and tested on a machine that
 * has no hypre installation.  One build configuration is fixed:
 *
 *     HYPRE_MIXEDINT  (HYPRE_Int = 32-bit, HYPRE_BigInt = 64-bit)
 *     HYPRE_Real = HYPRE_Complex = double
 *     HYPRE_USING_GPU + HYPRE_USING_HIP, no vendor sparse library,
 *     no persistent-communication handles, MPI_Comm = integer handle
 *
 * Reference interface replaced (all paths relative to /root/reference/src):
 *   utilities/HYPRE_utilities.h:30-131  (scalar typedefs)
 *   utilities/HYPRE_utilities.h:147-151 (error codes)
 *   utilities/HYPRE_utilities.h:316-346 (memory location / exec policy)
 *   utilities/error.h:18-44             (sticky global error word)
 *   utilities/mpistubs.h:142            (integer communicator handle)
 */
#ifndef HYPRE_AMD_UTILITIES_H
#define HYPRE_AMD_UTILITIES_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int        HYPRE_Int;
typedef long long  HYPRE_BigInt;
typedef int        hypre_int;
typedef double     HYPRE_Real;
typedef double     HYPRE_Complex;

/* Communicator: an index into the library's communicator table
 * (see hypre_amd_comm.h).  Handle 0 is the single-rank world. */
typedef HYPRE_Int  MPI_Comm;
typedef MPI_Comm   hypre_MPI_Comm;
#define hypre_MPI_COMM_WORLD 0
#define hypre_MPI_COMM_NULL  (-1)

/* error word bits — utilities/HYPRE_utilities.h:147-151 */
#define HYPRE_ERROR_GENERIC   1
#define HYPRE_ERROR_MEMORY    2
#define HYPRE_ERROR_ARG       4    /* bits 3..7 carry the argument index */
#define HYPRE_ERROR_CONV    256

typedef enum _HYPRE_MemoryLocation
{
   HYPRE_MEMORY_UNDEFINED = -1,
   HYPRE_MEMORY_HOST      =  0,
   HYPRE_MEMORY_DEVICE    =  1
} HYPRE_MemoryLocation;

typedef enum _HYPRE_ExecutionPolicy
{
   HYPRE_EXEC_UNDEFINED = -1,
   HYPRE_EXEC_HOST      =  0,
   HYPRE_EXEC_DEVICE    =  1
} HYPRE_ExecutionPolicy;

/* Sticky error word — utilities/error.h:18-31.  Only error_flag is part of
 * the contract; the remaining members keep the reference's struct size. */
typedef struct
{
   HYPRE_Int  error_flag;
   HYPRE_Int  temp_error_flag;
   HYPRE_Int  print_to_memory;
   HYPRE_Int  verbosity;
   char      *memory;
   HYPRE_Int  mem_sz;
   HYPRE_Int  msg_sz;
} hypre_Error;

extern hypre_Error hypre__global_error;
#define hypre_error_flag hypre__global_error.error_flag

void hypre_error_handler(const char *filename, HYPRE_Int line, HYPRE_Int ierr, const char *msg);
#define hypre_error(IERR)            hypre_error_handler(__FILE__, __LINE__, IERR, NULL)
#define hypre_error_w_msg(IERR, msg) hypre_error_handler(__FILE__, __LINE__, IERR, msg)
#define hypre_error_in_arg(IARG)     hypre_error(HYPRE_ERROR_ARG | (IARG) << 3)

HYPRE_Int HYPRE_GetError(void);
HYPRE_Int HYPRE_ClearAllErrors(void);
HYPRE_Int HYPRE_GetErrorArg(void);
/* utilities/error.c:161-164, 221-225 */
HYPRE_Int HYPRE_CheckError(HYPRE_Int hypre_ierr, HYPRE_Int hypre_error_code);
HYPRE_Int HYPRE_ClearError(HYPRE_Int hypre_error_code);
/* last message passed to hypre_error_w_msg (empty string if none) */
const char *hypre_amd_LastErrorMessage(void);

/* library life cycle — utilities/general.c (HYPRE_Initialize / HYPRE_Finalize) */
HYPRE_Int HYPRE_Initialize(void);
HYPRE_Int HYPRE_Finalize(void);
HYPRE_Int HYPRE_SetMemoryLocation(HYPRE_MemoryLocation memory_location);
HYPRE_Int HYPRE_GetMemoryLocation(HYPRE_MemoryLocation *memory_location);
HYPRE_Int HYPRE_SetExecutionPolicy(HYPRE_ExecutionPolicy exec_policy);
HYPRE_Int HYPRE_GetExecutionPolicy(HYPRE_ExecutionPolicy *exec_policy);
/* 1 if a HIP device is usable by this process, 0 otherwise */
HYPRE_Int hypre_amd_DeviceAvailable(void);
/* whether public device ops end with a stream synchronise (default 1);
 * utilities/general.c hypre_SetSyncCudaCompute */
/* Host threads of the library's OpenMP loops (AMG setup, cached transposes).  At load the library limits itself to
 * the cores the process owns (affinity mask cut by a cgroup CPU quota); a launcher that runs several ranks per node
 * gives each its part with hypre_amd_SetHostThreads(hypre_amd_HostCpuShare() / ranks_per_node).  No reference
 * counterpart (hypre takes OMP_NUM_THREADS as it finds it). */
HYPRE_Int hypre_amd_HostCpuShare(void);
HYPRE_Int hypre_amd_SetHostThreads(HYPRE_Int num_threads);
HYPRE_Int hypre_SetSyncCudaCompute(HYPRE_Int action);
HYPRE_Int hypre_GetSyncCudaCompute(HYPRE_Int *cuda_compute_stream_sync_ptr);
HYPRE_Int hypre_SyncComputeStream(void);
/* Algorithmic bytes of everything the library launched since the last reset, counted by the launch wrappers (so they follow
 * the smoother, the value width and the rank's share actually run; a replayed HIP graph adds what its recording added):
 * csr_bytes = the count of SURVEY.md 8(d) — CSR entries at 4 bytes of index + the value width in use, row pointers, every
 * vector operand once; streamed_bytes = what the kernels are designed to read (the x-staged SpMV streams a 16-bit staged
 * index per entry and never the column array).  Either pointer may be NULL; reset != 0 clears both.  No reference
 * counterpart (the reference reports flop-free wall-clock times only). */
HYPRE_Int hypre_amd_ByteCounters(HYPRE_Real *csr_bytes, HYPRE_Real *streamed_bytes, HYPRE_Int reset);
/* the HIP stream (hipStream_t as void*) every kernel of the library is
 * launched on; utilities/handle.h hypre_HandleComputeStream */
void     *hypre_amd_ComputeStream(void);
void     *hypre_amd_CommStream(void);
/* HIP-event stopwatch on the compute stream (torch.cuda.Event only sees
 * torch's own stream): Start records an event, StopMs records another,
 * waits for it and returns the elapsed milliseconds between the two. */
HYPRE_Int hypre_amd_EventTimerStart(void);
HYPRE_Real hypre_amd_EventTimerStopMs(void);

/* memory model — utilities/memory.h:131 (hypre_TAlloc / hypre_TMemcpy) */
void *hypre_MAlloc(size_t size, HYPRE_MemoryLocation location);
void *hypre_CAlloc(size_t count, size_t elt_size, HYPRE_MemoryLocation location);
void  hypre_Free(void *ptr, HYPRE_MemoryLocation location);
void  hypre_Memcpy(void *dst, const void *src, size_t size,
                   HYPRE_MemoryLocation loc_dst, HYPRE_MemoryLocation loc_src);
void  hypre_Memset(void *ptr, HYPRE_Int value, size_t num, HYPRE_MemoryLocation location);

#define hypre_TAlloc(type, count, location)  ((type *) hypre_MAlloc((size_t)(sizeof(type) * (count)), location))
#define hypre_CTAlloc(type, count, location) ((type *) hypre_CAlloc((size_t)(count), (size_t) sizeof(type), location))
#define hypre_TFree(ptr, location)           (hypre_Free((void *)(ptr), location), (ptr) = NULL)
#define hypre_TMemcpy(dst, src, type, count, locdst, locsrc) \
   (hypre_Memcpy((void *)(dst), (const void *)(src), (size_t)(sizeof(type) * (count)), locdst, locsrc))

HYPRE_ExecutionPolicy hypre_GetExecPolicy1(HYPRE_MemoryLocation location);
HYPRE_ExecutionPolicy hypre_GetExecPolicy2(HYPRE_MemoryLocation location1,
                                           HYPRE_MemoryLocation location2);

/* integer array — utilities/int_array.h (CF markers are carried in these) */
typedef struct
{
   HYPRE_Int            *data;
   HYPRE_Int             size;
   HYPRE_MemoryLocation  memory_location;
} hypre_IntArray;
#define hypre_IntArrayData(array)            ((array) -> data)
#define hypre_IntArraySize(array)            ((array) -> size)
#define hypre_IntArrayMemoryLocation(array)  ((array) -> memory_location)
hypre_IntArray *hypre_IntArrayCreate(HYPRE_Int size);
HYPRE_Int hypre_IntArrayInitialize_v2(hypre_IntArray *array, HYPRE_MemoryLocation memory_location);
HYPRE_Int hypre_IntArrayDestroy(hypre_IntArray *array);

/* generic solver vtable — utilities/base.h:16-23 */
struct hypre_Solver_struct;
typedef struct hypre_Solver_struct *HYPRE_Solver;
typedef HYPRE_Int (*HYPRE_PtrToSolverFcn)(HYPRE_Solver, void *, void *, void *);
typedef HYPRE_Int (*HYPRE_PtrToDestroyFcn)(HYPRE_Solver);
typedef struct
{
   HYPRE_PtrToSolverFcn   setup;
   HYPRE_PtrToSolverFcn   solve;
   HYPRE_PtrToDestroyFcn  destroy;
} hypre_Solver;

#ifdef __cplusplus
}
#endif
#endif
