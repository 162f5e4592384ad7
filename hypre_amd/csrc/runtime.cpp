// hypre_amd runtime: error word, library handle (streams, scratch), memory model.
// Reference counterparts: utilities/error.c, utilities/handle.h:34-81,
// utilities/memory.c (hypre_MAlloc/hypre_Free/hypre_Memcpy), utilities/general.c
// (HYPRE_Initialize, HYPRE_SetMemoryLocation, hypre_SetSyncCudaCompute).
#include "internal.hpp"
#include <execinfo.h>
#include <omp.h>
#include <sched.h>
#include <unistd.h>
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <cstdio>
#include <string>

extern "C" {
hypre_Error hypre__global_error = {0, 0, 0, 1, nullptr, 0, 0};
}

static std::string g_last_msg;

extern "C" void hypre_error_handler(const char *filename, HYPRE_Int line, HYPRE_Int ierr, const char *msg)
{
   hypre__global_error.error_flag |= ierr;
   if (msg)
   {
      g_last_msg = msg;
      if (hypre__global_error.verbosity)
      {
         fprintf(stderr, "hypre_amd error 0x%x at %s:%d: %s\n", ierr, filename, line, msg);
      }
   }
}

extern "C" const char *hypre_amd_LastErrorMessage(void) { return g_last_msg.c_str(); }
extern "C" HYPRE_Int HYPRE_GetError(void) { return hypre_error_flag; }
extern "C" HYPRE_Int HYPRE_ClearAllErrors(void) { hypre_error_flag = 0; g_last_msg.clear(); return 0; }
extern "C" HYPRE_Int HYPRE_GetErrorArg(void) { return (hypre_error_flag >> 3) & 31; }
// utilities/error.c:221-225: clear the given bits (the value returned is the masked flag after clearing)
extern "C" HYPRE_Int HYPRE_ClearError(HYPRE_Int hypre_error_code)
{
   hypre_error_flag &= ~hypre_error_code;
   return (hypre_error_flag & hypre_error_code);
}
// utilities/error.c:161-164
extern "C" HYPRE_Int HYPRE_CheckError(HYPRE_Int hypre_ierr, HYPRE_Int hypre_error_code) { return hypre_ierr & hypre_error_code; }

namespace hamd {

Handle &handle()
{
   static Handle h;
   return h;
}

// Cores this process may really use: the affinity mask, cut by a cgroup CPU quota when one is set (a container on
// a 256-thread host is often limited to a handful of cores while OpenMP still sees all of them; 256 threads on 16
// cores made the setup of a 37-row level cost 0.9 s).  HYPRE_AMD_SETUP_THREADS overrides.
int host_cpu_share()
{
   static int cached = 0;
   if (cached > 0) { return cached; }
   int n = (int) sysconf(_SC_NPROCESSORS_ONLN);
   cpu_set_t set;
   if (sched_getaffinity(0, sizeof set, &set) == 0) { n = std::min(n, CPU_COUNT(&set)); }
   long long quota = -1, period = 0;
   if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r"))                      // cgroup v2: "max 100000" or "<quota> <period>"
   {
      char q[64] = {0};
      if (fscanf(f, "%63s %lld", q, &period) == 2 && strcmp(q, "max") != 0) { quota = atoll(q); }
      fclose(f);
   }
   else
   {
      if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(g, "%lld", &quota) != 1) { quota = -1; } fclose(g); }
      if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(g, "%lld", &period) != 1) { period = 0; } fclose(g); }
   }
   if (quota > 0 && period > 0) { n = (int) std::min<long long>(n, std::max<long long>(1, (quota + period - 1) / period)); }
   if (const char *e = getenv("HYPRE_AMD_SETUP_THREADS")) { if (atoi(e) > 0) { n = atoi(e); } }
   cached = std::max(1, n);
   return cached;
}

namespace {
// host loops (setup, cached transposes, comm packages) never use more OpenMP threads than the process owns cores
struct OmpCap { OmpCap() { if (omp_get_max_threads() > host_cpu_share()) { omp_set_num_threads(host_cpu_share()); } } } omp_cap_at_load;
}

bool ensure_device()
{
   Handle &h = handle();
   if (h.device_probed) { return h.device_ok; }
   h.device_probed = true;
   int n = 0;
   if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
   {
      (void) hipGetLastError();
      h.device_ok = false;
      return false;
   }
   // one process per GPU: honour the launcher's LOCAL_RANK unless the caller
   // already chose a device with hipSetDevice / torch.cuda.set_device
   int dev = 0;
   if (hipGetDevice(&dev) != hipSuccess) { dev = 0; }
   hipDeviceProp_t prop;
   if (hipGetDeviceProperties(&prop, dev) == hipSuccess) { h.num_cus = prop.multiProcessorCount; }
   if (hipStreamCreateWithFlags(&h.compute_stream, hipStreamNonBlocking) != hipSuccess ||
       hipStreamCreateWithFlags(&h.comm_stream, hipStreamNonBlocking) != hipSuccess)
   {
      h.device_ok = false;
      return false;
   }
   if (hipHostMalloc((void **) &h.h_reduce, 16 * sizeof(double), hipHostMallocDefault) != hipSuccess)
   {
      h.device_ok = false;
      return false;
   }
   h.device_ok = true;
   preload_cheby_kernels(); preload_gs_kernels(); preload_interp_kernels(); preload_vector_kernels();
   preload_rap_kernels(); preload_setup_kernels(); preload_spmv_kernels(); preload_dist_setup_kernels(); preload_mc_kernels();
   return true;
}

hipStream_t stream()
{
   ensure_device();
   return handle().compute_stream;
}

void maybe_sync()
{
   Handle &h = handle();
   if (h.device_ok && h.sync_compute) { HIP_CHECK(hipStreamSynchronize(h.compute_stream)); }
}

double *reduce_scratch(size_t n)
{
   Handle &h = handle();
   if (h.d_reduce_len < n)
   {
      if (h.d_reduce) { HIP_CHECK(hipFree(h.d_reduce)); }
      size_t len = n < 4096 ? 4096 : n;
      HIP_CHECK(hipMalloc((void **) &h.d_reduce, len * sizeof(double)));
      h.d_reduce_len = len;
   }
   return h.d_reduce;
}

}  // namespace hamd

using namespace hamd;

extern "C" {

HYPRE_Int HYPRE_Initialize(void) { ensure_device(); return hypre_error_flag; }
HYPRE_Int HYPRE_Finalize(void) { return hypre_error_flag; }
HYPRE_Int hypre_amd_DeviceAvailable(void) { return ensure_device() ? 1 : 0; }

HYPRE_Int HYPRE_SetMemoryLocation(HYPRE_MemoryLocation loc) { handle().memory_location = loc; return hypre_error_flag; }
HYPRE_Int HYPRE_GetMemoryLocation(HYPRE_MemoryLocation *loc) { *loc = handle().memory_location; return hypre_error_flag; }
HYPRE_Int HYPRE_SetExecutionPolicy(HYPRE_ExecutionPolicy p) { handle().exec_policy = p; return hypre_error_flag; }
HYPRE_Int HYPRE_GetExecutionPolicy(HYPRE_ExecutionPolicy *p) { *p = handle().exec_policy; return hypre_error_flag; }

// host threads of the library's OpenMP loops (AMG setup, cached transposes): the share of cores this process
// owns, and a setter for launchers that split a node between several ranks
HYPRE_Int hypre_amd_HostCpuShare(void) { return hamd::host_cpu_share(); }
HYPRE_Int hypre_amd_SetHostThreads(HYPRE_Int n)
{
   if (n < 1) { hypre_error_in_arg(1); return hypre_error_flag; }
   omp_set_num_threads(n);
   return hypre_error_flag;
}
HYPRE_Int hypre_SetSyncCudaCompute(HYPRE_Int action) { handle().sync_compute = action; return hypre_error_flag; }
HYPRE_Int hypre_GetSyncCudaCompute(HYPRE_Int *p) { *p = handle().sync_compute; return hypre_error_flag; }
HYPRE_Int hypre_SyncComputeStream(void)
{
   Handle &h = handle();
   if (h.device_ok) { HIP_CHECK(hipStreamSynchronize(h.compute_stream)); }
   return hypre_error_flag;
}
// Algorithmic bytes of the launches since the last reset (Handle::bytes_csr / bytes_stream): the CSR count of SURVEY 8(d)
// and what the kernels are designed to stream.
HYPRE_Int hypre_amd_ByteCounters(HYPRE_Real *csr_bytes, HYPRE_Real *streamed_bytes, HYPRE_Int reset)
{
   Handle &h = handle();
   if (csr_bytes) { *csr_bytes = h.bytes_csr; }
   if (streamed_bytes) { *streamed_bytes = h.bytes_stream; }
   if (reset) { h.bytes_csr = 0.0; h.bytes_stream = 0.0; }
   return hypre_error_flag;
}
static hipEvent_t g_ev0 = nullptr, g_ev1 = nullptr;
HYPRE_Int hypre_amd_EventTimerStart(void)
{
   if (!ensure_device()) { hypre_error_w_msg(HYPRE_ERROR_GENERIC, "no HIP device"); return hypre_error_flag; }
   if (!g_ev0) { HIP_CHECK(hipEventCreate(&g_ev0)); HIP_CHECK(hipEventCreate(&g_ev1)); }
   HIP_CHECK(hipEventRecord(g_ev0, handle().compute_stream));
   return hypre_error_flag;
}
HYPRE_Real hypre_amd_EventTimerStopMs(void)
{
   float ms = 0.f;
   if (!g_ev0) { return 0.0; }
   HIP_CHECK(hipEventRecord(g_ev1, handle().compute_stream));
   HIP_CHECK(hipEventSynchronize(g_ev1));
   HIP_CHECK(hipEventElapsedTime(&ms, g_ev0, g_ev1));
   return (HYPRE_Real) ms;
}
void *hypre_amd_ComputeStream(void) { ensure_device(); return (void *) handle().compute_stream; }
void *hypre_amd_CommStream(void) { ensure_device(); return (void *) handle().comm_stream; }

HYPRE_ExecutionPolicy hypre_GetExecPolicy1(HYPRE_MemoryLocation location)
{
   // utilities/memory.c:1070 — host memory runs on the host, device memory on the device
   return location == HYPRE_MEMORY_DEVICE ? HYPRE_EXEC_DEVICE : HYPRE_EXEC_HOST;
}
HYPRE_ExecutionPolicy hypre_GetExecPolicy2(HYPRE_MemoryLocation l1, HYPRE_MemoryLocation l2)
{
   if (l1 == HYPRE_MEMORY_DEVICE && l2 == HYPRE_MEMORY_DEVICE) { return HYPRE_EXEC_DEVICE; }
   if (l1 == HYPRE_MEMORY_HOST && l2 == HYPRE_MEMORY_HOST) { return HYPRE_EXEC_HOST; }
   return HYPRE_EXEC_UNDEFINED;
}

void *hypre_MAlloc(size_t size, HYPRE_MemoryLocation location)
{
   if (size == 0) { return nullptr; }
   void *p = nullptr;
   if (location == HYPRE_MEMORY_DEVICE)
   {
      if (!ensure_device())
      {
         hypre_error_w_msg(HYPRE_ERROR_MEMORY, "device allocation requested but no HIP device is available");
         return nullptr;
      }
      if (hipMalloc(&p, size) != hipSuccess)
      {
         (void) hipGetLastError();
         hypre_error_w_msg(HYPRE_ERROR_MEMORY, "hipMalloc failed");
         return nullptr;
      }
   }
   else
   {
      // 64-byte alignment keeps host buffers friendly to pinned/async copies
      if (posix_memalign(&p, 64, size) != 0) { p = nullptr; }
      if (!p) { hypre_error_w_msg(HYPRE_ERROR_MEMORY, "host allocation failed"); }
   }
   return p;
}

void *hypre_CAlloc(size_t count, size_t elt_size, HYPRE_MemoryLocation location)
{
   const size_t size = count * elt_size;
   void *p = hypre_MAlloc(size, location);
   if (p) { hypre_Memset(p, 0, size, location); }
   return p;
}

void hypre_Free(void *ptr, HYPRE_MemoryLocation location)
{
   if (!ptr) { return; }
   if (location == HYPRE_MEMORY_DEVICE)
   {
      const hipError_t e = hipFree(ptr);
      if (e != hipSuccess)
      {
         (void) hipGetLastError();
         // a host pointer freed as device memory (or a double free): say who asked, once the caller wants to know
         if (getenv("HYPRE_AMD_DEBUG_FREE"))
         {
            void *frames[24];
            const int nf = backtrace(frames, 24);
            backtrace_symbols_fd(frames, nf, 2);
         }
         char msg[256];
         snprintf(msg, sizeof(msg), "HIP error %d (%s) in hipFree(%p)", (int) e, hipGetErrorString(e), ptr);
         hypre_error_handler(__FILE__, __LINE__, HYPRE_ERROR_GENERIC, msg);
      }
   }
   else { free(ptr); }
}

void hypre_Memset(void *ptr, HYPRE_Int value, size_t num, HYPRE_MemoryLocation location)
{
   if (!ptr || !num) { return; }
   if (location == HYPRE_MEMORY_DEVICE)
   {
      HIP_CHECK(hipMemsetAsync(ptr, value, num, stream()));
      HIP_CHECK(hipStreamSynchronize(stream()));
   }
   else { memset(ptr, value, num); }
}

void hypre_Memcpy(void *dst, const void *src, size_t size, HYPRE_MemoryLocation loc_dst, HYPRE_MemoryLocation loc_src)
{
   if (!size || dst == src) { return; }
   if (loc_dst == HYPRE_MEMORY_HOST && loc_src == HYPRE_MEMORY_HOST) { memcpy(dst, src, size); return; }
   hipMemcpyKind kind = hipMemcpyDeviceToDevice;
   if (loc_dst == HYPRE_MEMORY_HOST) { kind = hipMemcpyDeviceToHost; }
   else if (loc_src == HYPRE_MEMORY_HOST) { kind = hipMemcpyHostToDevice; }
   // ordered after everything already queued on the compute stream
   HIP_CHECK(hipMemcpyAsync(dst, src, size, kind, stream()));
   HIP_CHECK(hipStreamSynchronize(stream()));
}

hypre_IntArray *hypre_IntArrayCreate(HYPRE_Int size)
{
   hypre_IntArray *a = (hypre_IntArray *) calloc(1, sizeof(hypre_IntArray));
   a->size = size;
   a->memory_location = handle().memory_location;
   return a;
}
HYPRE_Int hypre_IntArrayInitialize_v2(hypre_IntArray *a, HYPRE_MemoryLocation loc)
{
   a->memory_location = loc;
   if (!a->data) { a->data = hypre_CTAlloc(HYPRE_Int, a->size, loc); }
   return hypre_error_flag;
}
HYPRE_Int hypre_IntArrayDestroy(hypre_IntArray *a)
{
   if (a) { hypre_Free(a->data, a->memory_location); free(a); }
   return hypre_error_flag;
}

}  // extern "C"
