// Development aid: cost of a grid-wide barrier (agent-scope atomics) inside one resident kernel, for the serial
// 'previous' chain.  usage: scripts/gridbar <workgroups> <threads> <barriers>
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// every workgroup arrives, the last one opens the next generation; spins are bounded (a grid that is not resident
// gives up and reports instead of hanging)
__device__ __forceinline__ bool grid_barrier(unsigned* count, unsigned* gen, unsigned nwg, unsigned& my_gen, int* fail) {
  __syncthreads();
  bool ok = true;
  if (threadIdx.x == 0) {
    __threadfence();
    const unsigned target = my_gen + 1;
    if (__hip_atomic_fetch_add(count, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == nwg - 1) {
      __hip_atomic_store(count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(gen, target, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      long spins = 0;
      while (__hip_atomic_load(gen, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != target) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > 20000000) { *fail = 1; ok = false; break; }
      }
    }
    __threadfence();
  }
  my_gen += 1;
  __syncthreads();
  return ok;
}

__global__ void bar_kernel(unsigned* count, unsigned* gen, int nbar, float* data, int work, int* fail) {
  unsigned my_gen = 0;
  float acc = 0.f;
  for (int b = 0; b < nbar; ++b) {
    // a little work that crosses workgroups: write a line, read the neighbour's after the barrier
    if (work) data[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = acc + b;
    if (!grid_barrier(count, gen, gridDim.x, my_gen, fail)) return;
    if (work) acc += __builtin_nontemporal_load(&data[(size_t)((blockIdx.x + 1) % gridDim.x) * blockDim.x + threadIdx.x]);
  }
  if (work && acc == -1.f) data[0] = acc;
}

int main(int argc, char** argv) {
  const int nwg = argc > 1 ? atoi(argv[1]) : 64, nth = argc > 2 ? atoi(argv[2]) : 256, nbar = argc > 3 ? atoi(argv[3]) : 1000;
  unsigned* sync;
  float* data;
  int* fail;
  CK(hipMalloc(&sync, 512));
  CK(hipMalloc(&data, (size_t)nwg * nth * 4));
  CK(hipMalloc(&fail, 4));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int work = 0; work < 2; ++work) {
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipMemset(sync, 0, 512));
      CK(hipMemset(fail, 0, 4));
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(bar_kernel, dim3(nwg), dim3(nth), 0, 0, sync, sync + 64, nbar, data, work, fail);
      CK(hipEventRecord(e1));
      CK(hipDeviceSynchronize());
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      int f;
      CK(hipMemcpy(&f, fail, 4, hipMemcpyDeviceToHost));
      printf("{\"workgroups\": %d, \"threads\": %d, \"barriers\": %d, \"work\": %d, \"us_per_barrier\": %.3f, \"gave_up\": %d}\n",
             nwg, nth, nbar, work, ms * 1e3 / nbar, f);
      if (f) return 2;
    }
  }
  return 0;
}
