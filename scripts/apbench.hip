// Development micro-benchmark for apply_shift_kernel (per-phase cycle counters of thread 0).
#ifndef NO_STAMPS
#define IAP_AP_STAMPS 1
#endif
#include "../imageanalysis_pipeline_amd/csrc/reg.hip"
using namespace iap;
int main(int argc, char** argv) {
  const int n = argc > 4 ? atoi(argv[4]) : 256, Y = 2048, X = 2048;
  double dy = argc > 1 ? atof(argv[1]) : 1.3, dx = argc > 2 ? atof(argv[2]) : -2.6;
  int ntx = argc > 3 ? atoi(argv[3]) : 16;
  uint16_t *in, *out; double* sh;
  hipMalloc(&in, (size_t)n * Y * X * 2); hipMalloc(&out, (size_t)n * Y * X * 2); hipMalloc(&sh, n * 16);
  hipMemset(in, 1, (size_t)n * Y * X * 2);
  std::vector<double> h(2 * n); for (int i = 0; i < n; ++i) { h[2 * i] = dy; h[2 * i + 1] = dx; }
  hipMemcpy(sh, h.data(), n * 16, hipMemcpyHostToDevice);
  int TY = argc > 5 ? atoi(argv[5]) : 32;
  dim3 grid((X / 64 + ntx - 1) / ntx, Y / TY, n);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int k = 0; k < 4; ++k) {
    unsigned long long z[8] = {0};
#ifdef IAP_AP_STAMPS
    hipMemcpyToSymbol(HIP_SYMBOL(g_ap_stamps), z, sizeof(z));
#endif
    hipEventRecord(a);
    hipLaunchKernelGGL(apply_shift_kernel<32>, grid, dim3(256), 0, 0, in, out, 1LL, Y, X, sh, ntx);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
#ifdef IAP_AP_STAMPS
    hipMemcpyFromSymbol(z, HIP_SYMBOL(g_ap_stamps), sizeof(z));
#endif
    double tiles = (double)n * (Y / TY) * (X / 64);
    printf("%.2f us/plane | cycles per tile: prologue %.0f syncA %.0f xpass %.0f syncB %.0f stage %.0f ypass+store %.0f\n", ms * 1e3 / n,
           z[0] / tiles, z[1] / tiles, z[2] / tiles, z[3] / tiles, z[4] / tiles, z[5] / tiles);
  }
  return 0;
}
