"""ctypes binding of libhypre_amd.so (the C-ABI boundary declared in include/*.h).

Struct classes mirror the headers field by field; function names are hypre's.
No compute happens in Python: every call below lands in the HIP library.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# HYPRE_AMD_LIB: load another build of the same library (A/B timing of two builds in one GPU call)
LIB_PATH = os.environ.get("HYPRE_AMD_LIB") or os.path.join(HERE, "lib", "libhypre_amd.so")

HYPRE_MEMORY_HOST = 0
HYPRE_MEMORY_DEVICE = 1

Int = C.c_int
BigInt = C.c_longlong
Real = C.c_double
IntP = C.POINTER(C.c_int)
BigIntP = C.POINTER(C.c_longlong)
RealP = C.POINTER(C.c_double)


class HypreAmdError(RuntimeError):
    pass


class CSRMatrix(C.Structure):
    _fields_ = [("i", IntP), ("j", IntP), ("big_j", BigIntP), ("num_rows", Int), ("num_cols", Int),
                ("num_nonzeros", Int), ("i_short", IntP), ("j_short", IntP), ("owns_data", Int),
                ("pattern_only", Int), ("data", RealP), ("rownnz", IntP), ("num_rownnz", Int),
                ("memory_location", Int)]


class Vector(C.Structure):
    _fields_ = [("data", RealP), ("size", Int), ("component", Int), ("owns_data", Int),
                ("memory_location", Int), ("num_vectors", Int), ("multivec_storage_method", Int),
                ("vecstride", Int), ("idxstride", Int)]


class CommPkg(C.Structure):
    _fields_ = [("comm", Int), ("num_components", Int), ("num_sends", Int), ("send_procs", IntP),
                ("send_map_starts", IntP), ("send_map_elmts", IntP), ("device_send_map_elmts", IntP),
                ("num_recvs", Int), ("recv_procs", IntP), ("recv_vec_starts", IntP),
                ("send_mpi_types", C.c_void_p), ("recv_mpi_types", C.c_void_p), ("tmp_data", RealP),
                ("buf_data", RealP), ("matrix_E", C.c_void_p)]


class ParCSRMatrix(C.Structure):
    _fields_ = [("comm", Int), ("global_num_rows", BigInt), ("global_num_cols", BigInt),
                ("global_num_rownnz", BigInt), ("num_nonzeros", BigInt), ("d_num_nonzeros", Real),
                ("first_row_index", BigInt), ("first_col_diag", BigInt), ("last_row_index", BigInt),
                ("last_col_diag", BigInt), ("diag", C.POINTER(CSRMatrix)), ("offd", C.POINTER(CSRMatrix)),
                ("diagT", C.POINTER(CSRMatrix)), ("offdT", C.POINTER(CSRMatrix)), ("col_map_offd", BigIntP),
                ("device_col_map_offd", BigIntP), ("row_starts", BigInt * 2), ("col_starts", BigInt * 2),
                ("comm_pkg", C.POINTER(CommPkg)), ("comm_pkgT", C.POINTER(CommPkg)), ("owns_data", Int),
                ("rowindices", BigIntP), ("rowvalues", RealP), ("getrowactive", Int),
                ("assumed_partition", C.c_void_p), ("owns_assumed_partition", Int), ("proc_ordering", IntP),
                ("bdiag_size", Int), ("bdiaginv", RealP), ("bdiaginv_comm_pkg", C.c_void_p),
                ("soc_diag_j", IntP), ("soc_offd_j", IntP)]


class ParVector(C.Structure):
    _fields_ = [("comm", Int), ("global_size", BigInt), ("first_index", BigInt), ("last_index", BigInt),
                ("partitioning", BigInt * 2), ("actual_local_size", Int), ("local_vector", C.POINTER(Vector)),
                ("owns_data", Int), ("all_zeros", Int), ("assumed_partition", C.c_void_p)]


class IJMatrix(C.Structure):
    _fields_ = [("comm", Int), ("row_partitioning", BigInt * 2), ("col_partitioning", BigInt * 2),
                ("object_type", Int), ("object", C.c_void_p), ("translator", C.c_void_p),
                ("assumed_part", C.c_void_p), ("assemble_flag", Int), ("global_first_row", BigInt),
                ("global_first_col", BigInt), ("global_num_rows", BigInt), ("global_num_cols", BigInt),
                ("omp_flag", Int), ("print_level", Int)]


class IJVector(C.Structure):
    _fields_ = [("comm", Int), ("partitioning", BigInt * 2), ("num_components", Int), ("object_type", Int),
                ("object", C.c_void_p), ("translator", C.c_void_p), ("assumed_part", C.c_void_p),
                ("global_first_row", BigInt), ("global_num_rows", BigInt), ("print_level", Int)]


HYPRE_PARCSR = 5555


class IntArray(C.Structure):
    _fields_ = [("data", IntP), ("size", Int), ("memory_location", Int)]


class ErrorStruct(C.Structure):
    _fields_ = [("error_flag", Int), ("temp_error_flag", Int), ("print_to_memory", Int), ("verbosity", Int),
                ("memory", C.c_char_p), ("mem_sz", Int), ("msg_sz", Int)]


EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, IntP, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t),
                          C.c_int, IntP, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_int, C.c_void_p)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, RealP, C.c_int, C.c_int, C.c_void_p)
ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t)
BARRIER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p)
DESTROY_FN = C.CFUNCTYPE(None, C.c_void_p)


class CommOps(C.Structure):
    _fields_ = [("ctx", C.c_void_p), ("rank", C.c_int), ("size", C.c_int), ("exchange", EXCHANGE_FN),
                ("allreduce_sum", ALLREDUCE_FN), ("allgather", ALLGATHER_FN), ("barrier", BARRIER_FN),
                ("destroy", DESTROY_FN), ("device_buffers", C.c_int)]


CSRp = C.POINTER(CSRMatrix)
Vecp = C.POINTER(Vector)
ParCSRp = C.POINTER(ParCSRMatrix)
ParVecp = C.POINTER(ParVector)

# name -> (restype, argtypes); every symbol declared in include/*.h appears here
# or in parcsr_ls_binding.PROTOTYPES (tests check the union against the headers).
PROTOTYPES = {
    # utilities
    "hypre_error_handler": (None, [C.c_char_p, Int, Int, C.c_char_p]),
    "HYPRE_GetError": (Int, []),
    "HYPRE_CheckError": (Int, [Int, Int]),
    "HYPRE_ClearError": (Int, [Int]),
    "HYPRE_ClearAllErrors": (Int, []),
    "HYPRE_GetErrorArg": (Int, []),
    "hypre_amd_LastErrorMessage": (C.c_char_p, []),
    "HYPRE_Initialize": (Int, []),
    "hypre_amd_HostCpuShare": (Int, []),
    "hypre_amd_SetHostThreads": (Int, [Int]),
    "HYPRE_Finalize": (Int, []),
    "HYPRE_SetMemoryLocation": (Int, [Int]),
    "HYPRE_GetMemoryLocation": (Int, [IntP]),
    "HYPRE_SetExecutionPolicy": (Int, [Int]),
    "HYPRE_GetExecutionPolicy": (Int, [IntP]),
    "hypre_amd_DeviceAvailable": (Int, []),
    "hypre_SetSyncCudaCompute": (Int, [Int]),
    "hypre_GetSyncCudaCompute": (Int, [IntP]),
    "hypre_SyncComputeStream": (Int, []),
    "hypre_amd_ByteCounters": (Int, [C.POINTER(C.c_double), C.POINTER(C.c_double), Int]),
    "hypre_amd_ComputeStream": (C.c_void_p, []),
    "hypre_amd_CommStream": (C.c_void_p, []),
    "hypre_amd_EventTimerStart": (Int, []),
    "hypre_amd_EventTimerStopMs": (Real, []),
    "hypre_MAlloc": (C.c_void_p, [C.c_size_t, Int]),
    "hypre_CAlloc": (C.c_void_p, [C.c_size_t, C.c_size_t, Int]),
    "hypre_Free": (None, [C.c_void_p, Int]),
    "hypre_Memcpy": (None, [C.c_void_p, C.c_void_p, C.c_size_t, Int, Int]),
    "hypre_Memset": (None, [C.c_void_p, Int, C.c_size_t, Int]),
    "hypre_GetExecPolicy1": (Int, [Int]),
    "hypre_GetExecPolicy2": (Int, [Int, Int]),
    "hypre_IntArrayCreate": (C.POINTER(IntArray), [Int]),
    "hypre_IntArrayInitialize_v2": (Int, [C.POINTER(IntArray), Int]),
    "hypre_IntArrayDestroy": (Int, [C.POINTER(IntArray)]),
    # comm
    "hypre_amd_CommCreate": (Int, [C.POINTER(CommOps)]),
    "hypre_amd_CommDestroy": (Int, [Int]),
    "hypre_amd_RCCLGetUniqueId": (Int, [C.c_void_p]),
    "hypre_amd_CommCreateRCCL": (Int, [C.c_void_p, C.c_int, C.c_int]),
    "hypre_amd_CommCreateStreamStaged": (Int, [Int]),
    "hypre_MPI_Comm_rank": (Int, [Int, IntP]),
    "hypre_MPI_Comm_size": (Int, [Int, IntP]),
    "hypre_amd_CommSelfTest": (Int, [Int, Int]),
    "hypre_MPI_Barrier": (Int, [Int]),
    "hypre_amd_CommCounters": (Int, [BigIntP, BigIntP, Int]),
    "hypre_amd_CommSetTiming": (Int, [Int]),
    "hypre_amd_CommSetTag": (Int, [Int]),
    "hypre_amd_CommExposedTimes": (Int, [Int, IntP, IntP, RealP, RealP, RealP]),
    "hypre_amd_CommBytes": (Int, [C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]),
    "hypre_amd_ParCSRMatrixHaloInfo": (Int, [ParCSRp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    # seq_mv
    "hypre_CSRMatrixCreate": (CSRp, [Int, Int, Int]),
    "hypre_CSRMatrixInitialize_v2": (Int, [CSRp, Int, Int]),
    "hypre_CSRMatrixInitialize": (Int, [CSRp]),
    "hypre_CSRMatrixDestroy": (Int, [CSRp]),
    "hypre_CSRMatrixSetRownnz": (Int, [CSRp]),
    "hypre_CSRMatrixMigrate": (Int, [CSRp, Int]),
    "hypre_CSRMatrixClone_v2": (CSRp, [CSRp, Int, Int]),
    "hypre_CSRMatrixTranspose": (Int, [CSRp, C.POINTER(CSRp), Int]),
    "hypre_CSRMatrixReorder": (Int, [CSRp]),
    "hypre_SeqVectorCreate": (Vecp, [Int]),
    "hypre_SeqMultiVectorCreate": (Vecp, [Int, Int]),
    "hypre_SeqVectorInitialize_v2": (Int, [Vecp, Int]),
    "hypre_SeqVectorInitialize": (Int, [Vecp]),
    "hypre_SeqVectorDestroy": (Int, [Vecp]),
    "hypre_SeqVectorMigrate": (Int, [Vecp, Int]),
    "hypre_SeqVectorCloneDeep_v2": (Vecp, [Vecp, Int]),
    "hypre_SeqVectorCloneDeep": (Vecp, [Vecp]),
    "hypre_CSRMatrixMatvecOutOfPlace": (Int, [Real, CSRp, Vecp, Real, Vecp, Vecp, Int]),
    "hypre_CSRMatrixMatvec": (Int, [Real, CSRp, Vecp, Real, Vecp]),
    "hypre_CSRMatrixMatvecT": (Int, [Real, CSRp, Vecp, Real, Vecp]),
    "hypre_CSRMatrixMatvecDevice": (Int, [Int, Real, CSRp, Vecp, Real, Vecp, Vecp, Int]),
    "hypre_CSRMatrixSpMVDevice": (Int, [Int, Real, CSRp, Vecp, Real, Vecp, Int]),
    "hypre_amd_CSRMatrixInvalidatePlan": (Int, [CSRp]),
    "hypre_amd_CSRMatrixVerifyPlan": (Int, [CSRp]),
    "hypre_amd_CSRMatrixSetImmutable": (Int, [CSRp, Int]),
    "hypre_amd_CSRMatrixPlanForm": (Int, [CSRp]),
    "hypre_amd_PlanTestFailAlloc": (Int, [Int, Int]),
    "hypre_amd_SetMixedPrecisionValues": (Int, [Int]),
    "hypre_amd_SpmvSetRowSlices": (Int, [Int]),
    "hypre_amd_SpmvSetFusedMultivectors": (Int, [Int]),
    "hypre_amd_SpmvFusedMultivectorLaunches": (Int, []),
    "hypre_amd_CSRMatrixPlanRowSlices": (Int, [CSRp, IntP, IntP]),
    "hypre_amd_CSRMatrixSortRows": (Int, [CSRp, Int]),
    "hypre_amd_SpmvSetBandPolicy": (Int, [Int, Int, Int]),
    "hypre_amd_SpmvSetVariant": (Int, [Int, Int]),
    "hypre_amd_CSRMatrixPlanInfo": (Int, [CSRp, IntP, IntP]),
    "hypre_amd_CSRMatrixPlanStaging": (Int, [CSRp, IntP, RealP]),
    "hypre_amd_SpmvSetValueCodes": (Int, [Int]),
    "hypre_amd_SpmvSetSliceForm": (Int, [Int]),
    "hypre_amd_CSRMatrixPlanSliceForm": (Int, [CSRp]),
    "hypre_amd_CSRMatrixPlanValueCodes": (Int, [CSRp]),
    "hypre_SeqVectorSetConstantValues": (Int, [Vecp, Real]),
    "hypre_SeqVectorCopy": (Int, [Vecp, Vecp]),
    "hypre_SeqVectorScale": (Int, [Real, Vecp]),
    "hypre_SeqVectorAxpy": (Int, [Real, Vecp, Vecp]),
    "hypre_SeqVectorAxpyz": (Int, [Real, Vecp, Real, Vecp, Vecp]),
    "hypre_SeqVectorInnerProd": (Real, [Vecp, Vecp]),
    "hypre_SeqVectorElmdivpy": (Int, [Vecp, Vecp, Vecp]),
    "hypre_SeqVectorElmdivpyMarked": (Int, [Vecp, Vecp, Vecp, IntP, Int]),
    "hypre_SeqVectorSetConstantValuesDevice": (Int, [Vecp, Real]),
    "hypre_SeqVectorScaleDevice": (Int, [Real, Vecp]),
    "hypre_SeqVectorAxpyDevice": (Int, [Real, Vecp, Vecp]),
    "hypre_SeqVectorAxpyzDevice": (Int, [Real, Vecp, Real, Vecp, Vecp]),
    "hypre_SeqVectorInnerProdDevice": (Real, [Vecp, Vecp]),
    "hypre_SeqVectorElmdivpyDevice": (Int, [Vecp, Vecp, Vecp, IntP, Int]),
    "hypreDevice_IVAXPY": (Int, [Int, RealP, RealP, RealP]),
    "hypreDevice_IVAXPYMarked": (Int, [Int, RealP, RealP, RealP, IntP, Int]),
    "hypreDevice_DiagScaleVector2": (Int, [Int, Int, RealP, RealP, Real, RealP, RealP, Int]),
    # parcsr_mv
    "hypre_ParCSRMatrixCreate": (ParCSRp, [Int, BigInt, BigInt, BigIntP, BigIntP, Int, Int, Int]),
    "hypre_ParCSRMatrixInitialize_v2": (Int, [ParCSRp, Int]),
    "hypre_ParCSRMatrixDestroy": (Int, [ParCSRp]),
    "hypre_ParCSRMatrixMigrate": (Int, [ParCSRp, Int]),
    "hypre_ParCSRMatrixClone_v2": (ParCSRp, [ParCSRp, Int, Int]),
    "hypre_ParCSRMatrixSetNumNonzeros": (Int, [ParCSRp]),
    "hypre_ParCSRMatrixSetDNumNonzeros": (Int, [ParCSRp]),
    "hypre_amd_ParCSRMatrixKeepTranspose": (Int, [ParCSRp]),
    "hypre_ParVectorCreate": (ParVecp, [Int, BigInt, BigIntP]),
    "hypre_ParMultiVectorCreate": (ParVecp, [Int, BigInt, BigIntP, Int]),
    "hypre_ParCSRDiagScaleVector": (Int, [ParCSRp, ParVecp, ParVecp]),
    "hypre_ParVectorInitialize_v2": (Int, [ParVecp, Int]),
    "hypre_ParVectorInitialize": (Int, [ParVecp]),
    "hypre_ParVectorDestroy": (Int, [ParVecp]),
    "hypre_ParVectorSetLocalSize": (Int, [ParVecp, Int]),
    "hypre_ParVectorMigrate": (Int, [ParVecp, Int]),
    "hypre_MatvecCommPkgCreate": (Int, [ParCSRp]),
    "hypre_MatvecCommPkgDestroy": (Int, [C.POINTER(CommPkg)]),
    "hypre_ParCSRCommPkgUpdateVecStarts": (Int, [C.POINTER(CommPkg), Int, Int, Int]),
    "hypre_ParCSRCommPkgCreate_core": (Int, [Int, BigIntP, BigInt, BigIntP, Int, Int, IntP, C.POINTER(IntP),
                                             C.POINTER(IntP), IntP, C.POINTER(IntP), C.POINTER(IntP),
                                             C.POINTER(IntP)]),
    "hypre_ParCSRCommHandleCreate_v2": (C.c_void_p, [Int, C.POINTER(CommPkg), Int, C.c_void_p, Int, C.c_void_p]),
    "hypre_ParCSRCommHandleCreate": (C.c_void_p, [Int, C.POINTER(CommPkg), C.c_void_p, C.c_void_p]),
    "hypre_ParCSRCommHandleDestroy": (Int, [C.c_void_p]),
    "hypre_ParCSRMatrixMatvecOutOfPlace": (Int, [Real, ParCSRp, ParVecp, Real, ParVecp, ParVecp]),
    "hypre_ParCSRMatrixMatvec": (Int, [Real, ParCSRp, ParVecp, Real, ParVecp]),
    "hypre_ParCSRMatrixMatvecT": (Int, [Real, ParCSRp, ParVecp, Real, ParVecp]),
    "hypre_ParCSRMatrixMatvecOutOfPlaceDevice": (Int, [Real, ParCSRp, ParVecp, Real, ParVecp, ParVecp]),
    "hypre_ParCSRMatrixMatvecTDevice": (Int, [Real, ParCSRp, ParVecp, Real, ParVecp]),
    "HYPRE_ParCSRMatrixMatvec": (Int, [Real, ParCSRp, ParVecp, Real, ParVecp]),
    "hypre_ParVectorSetConstantValues": (Int, [ParVecp, Real]),
    "hypre_ParVectorSetZeros": (Int, [ParVecp]),
    "hypre_ParVectorCopy": (Int, [ParVecp, ParVecp]),
    "hypre_ParVectorScale": (Int, [Real, ParVecp]),
    "hypre_ParVectorAxpy": (Int, [Real, ParVecp, ParVecp]),
    "hypre_ParVectorAxpyz": (Int, [Real, ParVecp, Real, ParVecp, ParVecp]),
    "hypre_ParVectorInnerProd": (Real, [ParVecp, ParVecp]),
    "hypre_ParVectorElmdivpy": (Int, [ParVecp, ParVecp, ParVecp]),
    "hypre_ParVectorElmdivpyMarked": (Int, [ParVecp, ParVecp, ParVecp, IntP, Int]),
    "GenerateLaplacian": (ParCSRp, [Int, BigInt, BigInt, BigInt, Int, Int, Int, Int, Int, Int, RealP]),
    "GenerateLaplacian27pt": (ParCSRp, [Int, BigInt, BigInt, BigInt, Int, Int, Int, Int, Int, Int, RealP]),
    "GenerateDifConv": (ParCSRp, [Int, BigInt, BigInt, BigInt, Int, Int, Int, Int, Int, Int, RealP]),
    "GenerateVarDifConv": (ParCSRp, [Int, BigInt, BigInt, BigInt, Int, Int, Int, Int, Int, Int, Real, C.POINTER(ParVecp)]),
    "GenerateRotate7pt": (ParCSRp, [Int, BigInt, BigInt, Int, Int, Int, Int, Real, Real]),
    "GenerateSysLaplacian": (ParCSRp, [Int, BigInt, BigInt, BigInt, Int, Int, Int, Int, Int, Int, Int, RealP, RealP]),
    "hypre_amd_CSRMatrixFromArrays": (CSRp, [Int, Int, Int, IntP, IntP, RealP, Int]),
    "hypre_amd_SeqVectorFromArray": (Vecp, [Int, RealP, Int]),
    "hypre_amd_SeqVectorToArray": (Int, [Vecp, RealP]),
    "hypre_amd_CopyToHost": (Int, [C.c_void_p, C.c_void_p, C.c_size_t, Int]),
    "hypre_amd_ParCSRMatrixFromArrays": (ParCSRp, [Int, BigInt, BigInt, BigIntP, BigIntP, Int, BigIntP, IntP, IntP,
                                                   RealP, IntP, IntP, RealP, Int]),
    "hypre_amd_ParVectorFromArray": (ParVecp, [Int, BigInt, BigIntP, RealP, Int]),
    "hypre_amd_ParVectorToArray": (Int, [ParVecp, RealP]),
    # IJ text files
    "HYPRE_IJMatrixRead": (Int, [C.c_char_p, Int, Int, C.POINTER(C.POINTER(IJMatrix))]),
    "HYPRE_IJMatrixPrint": (Int, [C.POINTER(IJMatrix), C.c_char_p]),
    "HYPRE_IJMatrixGetObject": (Int, [C.POINTER(IJMatrix), C.POINTER(C.c_void_p)]),
    "HYPRE_IJMatrixDestroy": (Int, [C.POINTER(IJMatrix)]),
    "hypre_ParCSRMatrixPrintIJ": (Int, [ParCSRp, Int, Int, C.c_char_p]),
    "HYPRE_IJVectorRead": (Int, [C.c_char_p, Int, Int, C.POINTER(C.POINTER(IJVector))]),
    "HYPRE_IJVectorPrint": (Int, [C.POINTER(IJVector), C.c_char_p]),
    "HYPRE_IJVectorGetObject": (Int, [C.POINTER(IJVector), C.POINTER(C.c_void_p)]),
    "HYPRE_IJVectorDestroy": (Int, [C.POINTER(IJVector)]),
    "hypre_amd_IJMatrixWrap": (C.POINTER(IJMatrix), [ParCSRp]),
    "hypre_amd_IJVectorWrap": (C.POINTER(IJVector), [ParVecp]),
}

_lib = None


def load_library(build_if_missing=False):
    """Load libhypre_amd.so.  Fails loudly when the HIP library is absent: there
    is no Python or CPU implementation of the compute path to fall back to."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        if build_if_missing:
            from . import build as _b
            _b.build()
        else:
            raise HypreAmdError("HIP library %s is missing — run `python -m hypre_amd.build` "
                                "(there is no fallback implementation)" % LIB_PATH)
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    from . import parcsr_ls_binding as _ls
    for table in (PROTOTYPES, _ls.PROTOTYPES):
        for name, (res, args) in table.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
    _lib = lib
    return lib


class _LazyLib:
    def __getattr__(self, name):
        return getattr(load_library(), name)


lib = _LazyLib()


def check():
    """Raise if the library's sticky error word is set (then clear it)."""
    L = load_library()
    flag = L.HYPRE_GetError()
    if flag:
        msg = L.hypre_amd_LastErrorMessage().decode()
        L.HYPRE_ClearAllErrors()
        raise HypreAmdError("hypre error flag 0x%x: %s" % (flag, msg))


# ---------------------------------------------------------------------------
# numpy helpers
# ---------------------------------------------------------------------------
def _ip(a):
    return a.ctypes.data_as(IntP)


def _bp(a):
    return a.ctypes.data_as(BigIntP)


def _rp(a):
    return a.ctypes.data_as(RealP)


def csr_from_scipy(A, location=HYPRE_MEMORY_DEVICE):
    A = A.tocsr()
    ii = np.ascontiguousarray(A.indptr, dtype=np.int32)
    jj = np.ascontiguousarray(A.indices, dtype=np.int32)
    aa = np.ascontiguousarray(A.data, dtype=np.float64)
    m = lib.hypre_amd_CSRMatrixFromArrays(A.shape[0], A.shape[1], int(A.nnz), _ip(ii), _ip(jj), _rp(aa), location)
    check()
    return m


def csr_from_arrays(nrows, ncols, indptr, indices, data, location=HYPRE_MEMORY_DEVICE):
    ii = np.ascontiguousarray(indptr, dtype=np.int32)
    jj = np.ascontiguousarray(indices, dtype=np.int32)
    aa = np.ascontiguousarray(data, dtype=np.float64)
    m = lib.hypre_amd_CSRMatrixFromArrays(nrows, ncols, int(len(jj)), _ip(ii), _ip(jj), _rp(aa), location)
    check()
    return m


def fetch(ptr, count, dtype, location):
    """Copy `count` items behind a (host or device) pointer into a numpy array."""
    out = np.empty(int(count), dtype=dtype)
    if count:
        addr = C.cast(ptr, C.c_void_p)
        lib.hypre_amd_CopyToHost(out.ctypes.data_as(C.c_void_p), addr, out.nbytes, location)
    return out


def csr_to_arrays(m):
    """(indptr, indices, data) of a hypre_CSRMatrix* wherever it lives."""
    s = m.contents
    loc = s.memory_location
    ii = fetch(s.i, s.num_rows + 1, np.int32, loc)
    nnz = int(ii[-1]) if s.num_rows > 0 else 0
    jj = fetch(s.j, nnz, np.int32, loc)
    aa = fetch(s.data, nnz, np.float64, loc) if s.data else np.ones(nnz)
    return ii, jj, aa


def csr_to_scipy(m):
    import scipy.sparse as sp
    ii, jj, aa = csr_to_arrays(m)
    return sp.csr_matrix((aa, jj, ii), shape=(m.contents.num_rows, m.contents.num_cols))


def vec_from_numpy(x, location=HYPRE_MEMORY_DEVICE):
    x = np.ascontiguousarray(x, dtype=np.float64)
    v = lib.hypre_amd_SeqVectorFromArray(int(x.size), _rp(x), location)
    check()
    return v


def vec_to_numpy(v):
    s = v.contents
    out = np.empty(s.size * s.num_vectors, dtype=np.float64)
    lib.hypre_amd_SeqVectorToArray(v, _rp(out))
    return out


def parvec_from_numpy(x, comm=0, global_size=None, first=0, location=HYPRE_MEMORY_DEVICE):
    x = np.ascontiguousarray(x, dtype=np.float64)
    part = np.array([first, first + x.size], dtype=np.int64)
    gs = int(global_size if global_size is not None else x.size)
    v = lib.hypre_amd_ParVectorFromArray(comm, gs, _bp(part), _rp(x), location)
    check()
    return v


def parmultivec_from_numpy(M, comm=0, global_size=None, first=0, location=HYPRE_MEMORY_DEVICE):
    """ParVector of M.shape[1] columns (stored one after the other: parcsr_mv/par_vector.c:77-87) holding the local rows
    M[:, v] of every column; global_size and first are the global length and this rank's first row of ONE column."""
    M = np.asarray(M, dtype=np.float64)
    n, nv = M.shape
    part = np.array([first, first + n], dtype=np.int64)
    gs = int(global_size if global_size is not None else n)
    v = lib.hypre_ParMultiVectorCreate(comm, gs, _bp(part), nv)
    lib.hypre_ParVectorInitialize_v2(v, location)
    flat = np.ascontiguousarray(M.T).ravel()
    if flat.size:
        lib.hypre_Memcpy(C.cast(v.contents.local_vector.contents.data, C.c_void_p), flat.ctypes.data_as(C.c_void_p), flat.nbytes,
                         location, HYPRE_MEMORY_HOST)
    check()
    return v


def parmultivec_to_numpy(v):
    l = v.contents.local_vector.contents
    return fetch(l.data, l.size * l.num_vectors, np.float64, l.memory_location).reshape(l.num_vectors, l.size).T.copy()


def parvec_to_numpy(v):
    n = v.contents.local_vector.contents.size
    out = np.empty(n, dtype=np.float64)
    lib.hypre_amd_ParVectorToArray(v, _rp(out))
    return out


def laplacian(nx, ny, nz, P=1, Q=1, R=1, p=0, q=0, r=0, comm=0, values=None, kind="7pt"):
    """Host-resident ParCSR block of rank (p,q,r); mirrors test/ij.c:9703-9719."""
    if values is None:
        if kind == "27pt":
            values = np.array([26.0 if nz > 1 else (8.0 if ny > 1 else 2.0), -1.0])
        else:
            cx = cy = cz = 1.0
            d = 0.0
            if nx > 1:
                d += 2.0 * cx
            if ny > 1:
                d += 2.0 * cy
            if nz > 1:
                d += 2.0 * cz
            values = np.array([d, -cx, -cy, -cz])
    values = np.ascontiguousarray(values, dtype=np.float64)
    fn = {"7pt": lib.GenerateLaplacian, "27pt": lib.GenerateLaplacian27pt, "difconv": lib.GenerateDifConv}[kind]
    A = fn(comm, nx, ny, nz, P, Q, R, p, q, r, _rp(values))
    check()
    return A
