// Development harness: runs iap_bg_smo_u16 through the C-ABI on synthetic planes (for rocprofv3 --pmc passes,
// which crash under python/torch on this image).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include "../include/iap.h"
__global__ void fill(uint16_t* p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    unsigned h = (unsigned)i * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    p[i] = (uint16_t)(480 + (h & 63));   // flat noisy background
  }
}
int main(int argc, char** argv) {
  int n = argc > 1 ? atoi(argv[1]) : 240, Y = 2048, X = 2048;
  uint16_t *in, *out; void* ws; int *med2, *nbg;
  size_t px = (size_t)n * Y * X;
  hipMalloc(&in, px * 2); hipMalloc(&out, px * 2); hipMalloc(&med2, n * 4); hipMalloc(&nbg, n * 4);
  size_t wsb = iap_bg_smo_ws_bytes(n, Y, X, 0.f); hipMalloc(&ws, wsb);
  fill<<<4096, 256>>>(in, px);
  hipDeviceSynchronize();
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int k = 0; k < 2; ++k) {
    hipEventRecord(a);
    int rc = iap_bg_smo_u16(in, out, n, Y, X, 0.f, 7, 0.045f, med2, nbg, ws, wsb, nullptr);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("rc %d: %.2f us/plane\n", rc, ms * 1e3 / n);
  }
  return 0;
}
