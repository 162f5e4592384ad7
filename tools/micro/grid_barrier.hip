// Cost of a software grid barrier on MI355X (one atomic counter, agent scope), to decide whether a persistent
// multi-workgroup Gauss-Seidel kernel could beat one launch per dependency level.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/grid_barrier.hip -o /tmp/grid_barrier && /tmp/grid_barrier
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void barrier_loop(unsigned *bar, int rounds, double *sink, const double *src, int n)
{
   const unsigned nwg = gridDim.x;
   double acc = 0.0;
   for (int r = 0; r < rounds; r++)
   {
      // a little dependent work per round: one gather, like a level step
      acc += src[(threadIdx.x * 977 + blockIdx.x * 131 + r * 17) % n];
      __syncthreads();
      if (threadIdx.x == 0)
      {
         __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
         const unsigned target = (unsigned) (r + 1) * nwg;
         unsigned spins = 0;
         while ((int) (__hip_atomic_load(bar, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) - target) < 0)
         {
            if (++spins > (1u << 24)) { break; }       // never wait forever
            __builtin_amdgcn_s_sleep(1);
         }
      }
      __syncthreads();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
   }
   if (acc == 12345.678) { sink[0] = acc; }
}

int main()
{
   unsigned *bar; double *sink, *src; const int n = 1 << 20;
   hipMalloc(&bar, 4); hipMalloc(&sink, 8); hipMalloc(&src, 8 * n); hipMemset(src, 0, 8 * n);
   hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
   const int rounds = 2000;
   for (int nwg : {1, 4, 8, 16, 32, 64, 128, 256})
   {
      for (int rep = 0; rep < 2; rep++)
      {
         hipMemset(bar, 0, 4);
         hipEventRecord(e0, 0);
         hipLaunchKernelGGL(barrier_loop, dim3(nwg), dim3(256), 0, 0, bar, rounds, sink, src, n);
         hipEventRecord(e1, 0);
         hipEventSynchronize(e1);
         float ms = 0; hipEventElapsedTime(&ms, e0, e1);
         if (rep) { printf("workgroups %3d: %.2f us per round (gather + barrier)\n", nwg, 1000.0 * ms / rounds); }
      }
   }
   return 0;
}
