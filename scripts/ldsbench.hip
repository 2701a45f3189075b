// Development micro-benchmark: LDS read/write bandwidth per CU for 8-byte and 16-byte accesses.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v2 __attribute__((ext_vector_type(2)));
typedef float v4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
  __shared__ __align__(16) float lds[16384];   // 64 KB
  const int tid = threadIdx.x;
  for (int i = tid; i < 16384; i += 512) lds[i] = (float)i;
  __syncthreads();
  float acc = 0.f;
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {            // 8-byte reads, lanes contiguous
#pragma unroll
      for (int q = 0; q < 16; ++q) { v2 a = *(const v2*)&lds[(((tid + q * 512 + it) & 8191)) * 2]; acc += a.x + a.y; }
    } else if (MODE == 1) {     // 16-byte reads, lanes contiguous
#pragma unroll
      for (int q = 0; q < 8; ++q) { v4 a = *(const v4*)&lds[(((tid + q * 512 + it) & 4095)) * 4]; acc += a.x + a.y + a.z + a.w; }
    } else if (MODE == 2) {     // 8-byte writes
#pragma unroll
      for (int q = 0; q < 16; ++q) *(v2*)&lds[(((tid + q * 512 + it) & 8191)) * 2] = (v2){acc, (float)q};
    } else {                    // 16-byte writes
#pragma unroll
      for (int q = 0; q < 8; ++q) *(v4*)&lds[(((tid + q * 512 + it) & 4095)) * 4] = (v4){acc, (float)q, 1.f, 2.f};
    }
  }
  if (acc == 12345.f || MODE >= 2) out[blockIdx.x * 512 + tid] = acc + lds[tid];
}
template <int MODE> void run(const char* name, float* out) {
  const int iters = 2000, wgs = 512;   // 2 workgroups per CU
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(k<MODE>, dim3(wgs), dim3(512), 0, 0, out, 10);
  hipEventRecord(a);
  hipLaunchKernelGGL(k<MODE>, dim3(wgs), dim3(512), 0, 0, out, iters);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  double bytes = (double)wgs * 512 * iters * 128.0;   // 128 bytes per thread and iteration in every mode
  printf("%-16s %.2f ms  %.1f TB/s aggregate  %.1f B/clk/CU at 2.2 GHz\n", name, ms, bytes / ms / 1e9, bytes / (ms * 1e-3) / 256 / 2.2e9);
}
int main() {
  float* out; hipMalloc(&out, 512 * 512 * 4);
  run<0>("read 8 B", out); run<1>("read 16 B", out); run<2>("write 8 B", out); run<3>("write 16 B", out);
  return 0;
}
