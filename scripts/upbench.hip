// Development micro-benchmark for upsample_partial_kernel (per-workgroup time stamps).
#define IAP_UP_STAMPS 1
#include "../imageanalysis_pipeline_amd/csrc/reg.hip"
#include <vector>
#include <algorithm>
#include <map>
using namespace iap;
int main(int argc, char** argv) {
  int n = 16, Py = 2048, Px = 2048, Wp = 1040, ncg = 2;
  int nuc = argc > 1 ? atoi(argv[1]) : 12;
  int ub = (((Py + nuc - 1) / nuc) + 3) & ~3; nuc = (Py + ub - 1) / ub;
  size_t fs = (size_t)Py * Wp;
  cf *R; float *ey, *ex; float* part; long long* st; int2* coarse;
  hipMalloc(&coarse, n * 8); hipMemset(coarse, 0, n * 8);
  hipMalloc(&R, n * fs * 8); hipMemset(R, 0, n * fs * 8);
  hipMalloc(&ey, (size_t)n * Py * 16 * 4); hipMemset(ey, 0, (size_t)n * Py * 16 * 4);
  hipMalloc(&ex, (size_t)n * Wp * 20 * 4); hipMemset(ex, 0, (size_t)n * Wp * 20 * 4);
  int nch = ncg * nuc;
  hipMalloc(&part, (size_t)n * (nch + 1) * 225 * 4);
  hipMalloc(&st, (size_t)n * nch * 32);
  hipMemcpyToSymbol(HIP_SYMBOL(g_up_stamps), &st, sizeof(st));
  int per_cu = 0; hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, upsample_partial_kernel, 256, 0);
  printf("occupancy %d WG/CU, grid %d x %d = %d WGs, ub %d\n", per_cu, nch, n, nch * n, ub);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int k = 0; k < 3; ++k) {
    hipEventRecord(a);
    hipLaunchKernelGGL(upsample_partial_kernel, dim3(nch, n), dim3(256), 0, 0, R, (long long)fs, Wp, Py, Px, ncg, ub, (const int2*)coarse, ey, ex, part, nch);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("launch %d: %.1f us total, %.2f us/frame\n", k, ms * 1e3, ms * 1e3 / n);
  }
  std::vector<long long> h((size_t)n * nch * 4);
  hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost);
  long long t0 = h[0], t1 = 0;
  for (size_t i = 0; i < h.size() / 4; ++i) { t0 = std::min(t0, h[4 * i]); t1 = std::max(t1, h[4 * i + 2]); }
  printf("wall_clock span %lld ticks (100 MHz => %.1f us)\n", t1 - t0, (t1 - t0) / 100.0);
  std::vector<double> dm, dt, s0;
  std::map<long long, int> per_cu_count;
  for (size_t i = 0; i < h.size() / 4; ++i) {
    dm.push_back((h[4 * i + 1] - h[4 * i]) / 100.0); dt.push_back((h[4 * i + 2] - h[4 * i + 1]) / 100.0);
    s0.push_back((h[4 * i] - t0) / 100.0);
    unsigned hw = (unsigned)h[4 * i + 3]; unsigned xcc = (unsigned)(h[4 * i + 3] >> 32) & 0xf;
    unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
    per_cu_count[((long long)xcc << 16) | (se << 8) | (sh << 4) | cu]++;
  }
  auto pct = [](std::vector<double> v, double q) { std::sort(v.begin(), v.end()); return v[(size_t)(q * (v.size() - 1))]; };
  printf("main  us: min %.1f med %.1f max %.1f\n", pct(dm, 0), pct(dm, .5), pct(dm, 1));
  printf("tail  us: min %.1f med %.1f max %.1f\n", pct(dt, 0), pct(dt, .5), pct(dt, 1));
  printf("start us: med %.1f p90 %.1f max %.1f\n", pct(s0, .5), pct(s0, .9), pct(s0, 1));
  std::map<int, int> hist; for (auto& kv : per_cu_count) hist[kv.second]++;
  printf("distinct CUs %zu; WGs-per-CU histogram:", per_cu_count.size());
  for (auto& kv : hist) printf(" %d:%d", kv.first, kv.second);
  printf("\n");
  return 0;
}
