// Development micro-benchmark: VALU issue rate (v_fma_f32, independent accumulators) vs waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int PK>
__global__ __launch_bounds__(256) void k(float* out, int iters, long long* cyc) {
  typedef float v2f __attribute__((ext_vector_type(2)));
  float a[16]; v2f p[8];
  for (int i = 0; i < 16; ++i) a[i] = threadIdx.x * 0.001f + i;
  for (int i = 0; i < 8; ++i) p[i] = (v2f){a[2 * i], a[2 * i + 1]};
  float m = out[0], c = out[1];
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
    if (PK) {
#pragma unroll
      for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) p[i] = __builtin_elementwise_fma(p[i], (v2f){m, m}, (v2f){c, c});
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) a[i] = fmaf(a[i], m, c);
    }
  }
  long long t1 = clock64();
  float s = 0; for (int i = 0; i < 16; ++i) s += a[i]; for (int i = 0; i < 8; ++i) s += p[i].x + p[i].y;
  out[2 + blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
int main() {
  float* out; long long* cyc; hipMalloc(&out, 64 << 20); hipMalloc(&cyc, 8);
  float init[2] = {0.999f, 0.001f}; hipMemcpy(out, init, 8, hipMemcpyHostToDevice);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const int iters = 20000;
  for (int pk = 0; pk < 2; ++pk)
    for (int wgs = 1; wgs <= 8; wgs *= 2) {   // workgroups of 4 waves per CU -> waves per SIMD
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(a);
        if (pk) k<1><<<256 * wgs, 256>>>(out, iters, cyc); else k<0><<<256 * wgs, 256>>>(out, iters, cyc);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        double instr_per_simd = (double)iters * 64 * wgs;   // 64 VALU per iteration per wave, wgs waves per SIMD
        if (rep) printf("%s waves/SIMD %d: %.3f ms, clock64 ticks %lld (%.0f MHz tick rate), %.2f ns per instr per SIMD, %.2f ticks/instr/SIMD\n",
               pk ? "v_pk_fma_f32" : "v_fma_f32   ", wgs, ms, h, h / (ms * 1e3), ms * 1e6 / instr_per_simd, (double)h / instr_per_simd);
      }
    }
  return 0;
}
