// Development micro-benchmark for apply_shift_kernel: time per plane and, with -DSTAMPS, the
// per-phase cycle counters of thread 0.   apbench [nty] [planes] [dy dx]
#ifdef STAMPS
#define IAP_AP_STAMPS 1
#endif
#include "../imageanalysis_pipeline_amd/csrc/reg.hip"
using namespace iap;
int main(int argc, char** argv) {
  const int nty = argc > 1 ? atoi(argv[1]) : 8, n = argc > 2 ? atoi(argv[2]) : 128, Y = 2048, X = 2048;
  const double dy = argc > 4 ? atof(argv[3]) : 1.3, dx = argc > 4 ? atof(argv[4]) : -2.6;
  uint16_t *in, *out; double* sh;
  if (hipMalloc(&in, (size_t)n * Y * X * 2) || hipMalloc(&out, (size_t)n * Y * X * 2) || hipMalloc(&sh, n * 16)) return 2;
  hipMemset(in, 1, (size_t)n * Y * X * 2);
  std::vector<double> h(2 * n);
  for (int i = 0; i < n; ++i) { h[2 * i] = dy + 0.01 * i; h[2 * i + 1] = dx + 0.02 * i; }
  hipMemcpy(sh, h.data(), n * 16, hipMemcpyHostToDevice);
  dim3 grid(X / 64, (Y / 64 + nty - 1) / nty, n);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int k = 0; k < 4; ++k) {
#ifdef STAMPS
    unsigned long long z[8] = {0};
    hipMemcpyToSymbol(HIP_SYMBOL(g_ap_stamps), z, sizeof(z));
#endif
    hipEventRecord(a);
    hipLaunchKernelGGL(apply_shift_kernel, grid, dim3(256), 0, 0, in, out, 1LL, Y, X, sh, nty, (const int*)nullptr);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("nty=%d  %.2f us/plane", nty, ms * 1e3 / n);
#ifdef STAMPS
    hipMemcpyFromSymbol(z, HIP_SYMBOL(g_ap_stamps), sizeof(z));
    const double steps = (double)n * (Y / 64) * (X / 64);
    printf(" | ticks per step: lead-in %.0f syncA %.0f xpass %.0f syncB %.0f stage+preload %.0f ypass %.0f syncC+carry %.0f",
           z[0] / steps, z[1] / steps, z[2] / steps, z[3] / steps, z[4] / steps, z[5] / steps, z[6] / steps);
#endif
    printf("\n");
  }
  return 0;
}
