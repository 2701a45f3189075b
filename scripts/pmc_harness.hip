// Native harness for rocprofv3 counter passes (rocprofv3 --pmc crashes python/torch on this image): drives the
// stages of the hot path through the C-ABI of the in-tree libiap_hip.so -- the very library bench.py runs -- on
// synthetic planes of the benchmark's shape.  The only kernels of its own are the data generators.
//   pmc_harness apply    <frames> <C> [Y X]     iap_apply_shift_u16, one launch, sub-pixel shifts per frame
//   pmc_harness bg       <planes> [Y X]         iap_bg_smo_u16, levels only (what bench.py times)
//   pmc_harness estimate <frames> [Y X]         iap_reg_estimate_u16 (r2c, cols, c2r, upsample kernels), 64-frame groups
//   pmc_harness measure  <frames> <C> [Y X] [dense]   iap_measure_labels_u16; 300 discs per frame, or the 5041-label grid
//   pmc_harness blur     <planes> [Y X]         iap_gauss_blur_u16, the reference's default (15, 15), sigma 5 (experiments.py:337)
//   pmc_harness smosigma <planes> [Y X]         iap_bg_smo_u16 with SMO's gaussian pre-filter, sigma 1 (levels only)
// Prints the mean wall time per call (HIP events) so a run without the profiler doubles as a micro-benchmark.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "../include/iap.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define IAP(x) do { int r_ = (x); if (r_) { fprintf(stderr, "%s -> %d: %s\n", #x, r_, iap_last_error()); return 3; } } while (0)

__device__ __forceinline__ unsigned mix(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

// background ~ 400 + N(100, 10^2) (sum of four uniforms) plus smooth blobs: the content class of bench.py's stacks
__global__ void fill_planes(uint16_t* p, long long n_planes, int Y, int X, int nblob, const float4* blobs) {
  const long long px = (long long)Y * X;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_planes * px;
       i += (long long)gridDim.x * blockDim.x) {
    const long long pl = i / px;
    const int y = (int)((i - pl * px) / X), x = (int)(i - pl * px - (long long)y * X);
    unsigned h = mix((unsigned)i * 2654435761u + (unsigned)(i >> 32) * 40503u + 17u);
    float u = 0.f;
    for (int k = 0; k < 4; ++k) { u += (float)(h & 0xffffu) * (1.0f / 65536.0f); h = mix(h + 0x9e3779b9u); }
    float v = 500.f + (u - 2.f) * 17.3f;
    for (int b = 0; b < nblob; ++b) {
      const float4 bl = blobs[b];
      const float dy = (float)y - bl.x - 0.37f * (float)(pl % 7), dx = (float)x - bl.y + 0.21f * (float)(pl % 5);
      const float r2 = dy * dy + dx * dx;
      if (r2 < 16.f * bl.z * bl.z) v += bl.w * __expf(-r2 / (2.f * bl.z * bl.z));
    }
    p[i] = (uint16_t)fminf(fmaxf(rintf(v), 0.f), 65535.f);
  }
}

// discs at the blob centres (label b + 1) or, dense, a jittered grid x grid of discs
__global__ void fill_labels(uint16_t* lab, long long frames, int Y, int X, int nblob, const float4* blobs, int grid) {
  const long long px = (long long)Y * X;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < frames * px;
       i += (long long)gridDim.x * blockDim.x) {
    const long long f = i / px;
    const int y = (int)((i - f * px) / X), x = (int)(i - f * px - (long long)y * X);
    unsigned v = 0;
    if (grid > 0) {
      const float py = (float)Y / grid, pxx = (float)X / grid;
      const int gy = min(grid - 1, (int)(y / py)), gx = min(grid - 1, (int)(x / pxx));
      const unsigned id = (unsigned)(gy * grid + gx);
      const unsigned h = mix(id * 747796405u + 1u);
      const float cy = (gy + 0.5f + 0.3f * ((h & 255) / 255.f - 0.5f)) * py, cx = (gx + 0.5f + 0.3f * (((h >> 8) & 255) / 255.f - 0.5f)) * pxx;
      const float r = (0.28f + 0.14f * (((h >> 16) & 255) / 255.f)) * fminf(py, pxx);
      if ((y - cy) * (y - cy) + (x - cx) * (x - cx) <= r * r) v = 1u + (mix(id + 99u) % (unsigned)(grid * grid));
    } else {
      for (int b = 0; b < nblob; ++b) {
        const float4 bl = blobs[b];
        const float dy = (float)y - bl.x, dx = (float)x - bl.y, r = 1.2f * bl.z;
        if (dy * dy + dx * dx <= r * r) v = (unsigned)b + 1u;
      }
    }
    lab[i] = (uint16_t)v;
  }
}

static std::vector<float4> make_blobs(int n, int Y, int X) {
  std::vector<float4> b(n);
  unsigned s = 12345u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)(s >> 8) / 16777216.0f; };
  for (int i = 0; i < n; ++i)
    b[i] = make_float4((0.08f + 0.84f * rnd()) * Y, (0.08f + 0.84f * rnd()) * X, 6.f + 6.f * rnd(), 2000.f + 18000.f * rnd());
  return b;
}

int main(int argc, char** argv) {
  if (argc < 3) { fprintf(stderr, "usage: pmc_harness apply|bg|estimate|measure <n> ...\n"); return 1; }
  const char* stage = argv[1];
  const long long n = atoll(argv[2]);
  const int reps = getenv("IAP_HARNESS_REPS") ? atoi(getenv("IAP_HARNESS_REPS")) : 3;
  int C = 1, Y = 2048, X = 2048, dense = 0, a = 3;
  if (!strcmp(stage, "apply") || !strcmp(stage, "measure")) C = argc > a ? atoi(argv[a++]) : 4;
  if (argc > a + 1 && atoi(argv[a]) > 0) { Y = atoi(argv[a]); X = atoi(argv[a + 1]); a += 2; }
  if (argc > a && !strcmp(argv[a], "dense")) dense = 1;
  if (n <= 0 || C <= 0 || Y < 64 || X < 64) { fprintf(stderr, "bad sizes\n"); return 1; }
  const long long px = (long long)Y * X, planes = !strcmp(stage, "apply") || !strcmp(stage, "measure") ? n * C : n;
  const int nblob = (int)fmax(4.0, 300.0 * (double)px / (2048.0 * 2048.0));
  std::vector<float4> hb = make_blobs(nblob, Y, X);
  float4* blobs;
  CK(hipMalloc(&blobs, hb.size() * sizeof(float4)));
  CK(hipMemcpy(blobs, hb.data(), hb.size() * sizeof(float4), hipMemcpyHostToDevice));
  uint16_t* in;
  CK(hipMalloc(&in, (size_t)planes * px * 2));
  // blobs only where they matter (the estimate's peak, the measured labels); 24 of them keep the fill quick
  hipLaunchKernelGGL(fill_planes, dim3(8192), dim3(256), 0, 0, in, planes, Y, X, nblob < 24 ? nblob : 24, (const float4*)blobs);
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  float ms = 0.f;
  if (!strcmp(stage, "apply")) {
    uint16_t* out;
    double* sh;
    CK(hipMalloc(&out, (size_t)planes * px * 2));
    CK(hipMalloc(&sh, (size_t)n * 16));
    std::vector<double> h(2 * n);
    unsigned s = 7u;
    double dy = 0, dx = 0;
    for (long long i = 0; i < n; ++i) {   // a random walk of sub-pixel steps, as bench.py's drift
      s = s * 1664525u + 1013904223u; dy += ((double)(s >> 8) / 16777216.0 - 0.5) * 2.4;
      s = s * 1664525u + 1013904223u; dx += ((double)(s >> 8) / 16777216.0 - 0.5) * 2.4;
      h[2 * i] = dy; h[2 * i + 1] = dx;
    }
    h[0] = h[1] = 0.0;
    if (const char* m = getenv("IAP_HARNESS_SHIFT"))   // experiments: "y", "x" or "yx" makes those axes' shifts whole numbers
      for (long long i = 0; i < n; ++i) {
        if (strchr(m, 'y')) h[2 * i] = floor(h[2 * i]);
        if (strchr(m, 'x')) h[2 * i + 1] = floor(h[2 * i + 1]);
      }
    CK(hipMemcpy(sh, h.data(), (size_t)n * 16, hipMemcpyHostToDevice));
    IAP(iap_apply_shift_u16(in, out, n, C, Y, X, sh, nullptr));
    CK(hipEventRecord(e0));
    for (int k = 0; k < reps; ++k) IAP(iap_apply_shift_u16(in, out, n, C, Y, X, sh, nullptr));
    CK(hipEventRecord(e1));
  } else if (!strcmp(stage, "bg")) {
    const size_t wsb = iap_bg_smo_ws_bytes(n, Y, X, 0.f);
    void* ws;
    int32_t* med2;
    CK(hipMalloc(&ws, wsb));
    CK(hipMalloc(&med2, (size_t)n * 4));
    IAP(iap_bg_smo_u16(in, nullptr, n, Y, X, 0.f, 7, 0.0455f, med2, nullptr, ws, wsb, nullptr));
    CK(hipEventRecord(e0));
    for (int k = 0; k < reps; ++k) IAP(iap_bg_smo_u16(in, nullptr, n, Y, X, 0.f, 7, 0.0455f, med2, nullptr, ws, wsb, nullptr));
    CK(hipEventRecord(e1));
  } else if (!strcmp(stage, "blur")) {
    uint16_t* out;
    CK(hipMalloc(&out, (size_t)planes * px * 2));
    const int k = getenv("IAP_HARNESS_BLUR_K") ? atoi(getenv("IAP_HARNESS_BLUR_K")) : 15;
    IAP(iap_gauss_blur_u16(in, out, n, Y, X, k, k, 5.0, nullptr));
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) IAP(iap_gauss_blur_u16(in, out, n, Y, X, k, k, 5.0, nullptr));
    CK(hipEventRecord(e1));
  } else if (!strcmp(stage, "smosigma")) {
    const size_t wsb = iap_bg_smo_ws_bytes(n, Y, X, 1.0f);
    void* ws;
    int32_t* med2;
    CK(hipMalloc(&ws, wsb));
    CK(hipMalloc(&med2, (size_t)n * 4));
    IAP(iap_bg_smo_u16(in, nullptr, n, Y, X, 1.0f, 7, 0.0455f, med2, nullptr, ws, wsb, nullptr));
    CK(hipEventRecord(e0));
    for (int k = 0; k < reps; ++k) IAP(iap_bg_smo_u16(in, nullptr, n, Y, X, 1.0f, 7, 0.0455f, med2, nullptr, ws, wsb, nullptr));
    CK(hipEventRecord(e1));
  } else if (!strcmp(stage, "estimate")) {
    const int bmax = getenv("IAP_HARNESS_BATCH") ? atoi(getenv("IAP_HARNESS_BATCH")) : 128;   // frames per launch group (workspace size; ops.py picks 128 at 2048^2)
    const int batch = (int)(n < bmax ? n : bmax);
    const size_t wsb = iap_reg_ws_bytes(Y, X, 1, batch);
    void* ws;
    double* sh;
    CK(hipMalloc(&ws, wsb));
    CK(hipMalloc(&sh, (size_t)n * 16));
    IAP(iap_reg_estimate_u16(in, 0, in, px, n, Y, X, IAP_REG_UPSAMPLE, sh, ws, wsb, nullptr));
    CK(hipEventRecord(e0));
    for (int k = 0; k < reps; ++k) IAP(iap_reg_estimate_u16(in, 0, in, px, n, Y, X, IAP_REG_UPSAMPLE, sh, ws, wsb, nullptr));
    CK(hipEventRecord(e1));
  } else if (!strcmp(stage, "measure")) {
    uint16_t* lab;
    CK(hipMalloc(&lab, (size_t)n * px * 2));
    hipLaunchKernelGGL(fill_labels, dim3(8192), dim3(256), 0, 0, lab, n, Y, X, nblob, (const float4*)blobs, dense ? 71 : 0);
    const int L = dense ? 71 * 71 + 1 : nblob + 1;
    uint64_t* acc;
    CK(hipMalloc(&acc, (size_t)n * L * (4 + C) * 8));
    CK(hipDeviceSynchronize());
    IAP(iap_measure_labels_u16(in, lab, n, C, 1, Y, X, L, acc, nullptr));
    CK(hipEventRecord(e0));
    for (int k = 0; k < reps; ++k) IAP(iap_measure_labels_u16(in, lab, n, C, 1, Y, X, L, acc, nullptr));
    CK(hipEventRecord(e1));
  } else {
    fprintf(stderr, "unknown stage %s\n", stage);
    return 1;
  }
  CK(hipEventSynchronize(e1));
  CK(hipEventElapsedTime(&ms, e0, e1));
  printf("{\"stage\": \"%s\", \"n\": %lld, \"C\": %d, \"Y\": %d, \"X\": %d, \"dense\": %d, \"ms_per_call\": %.4f, \"reps\": %d}\n",
         stage, n, C, Y, X, dense, ms / reps, reps);
  return 0;
}
