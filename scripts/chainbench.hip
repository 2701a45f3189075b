// Development micro-benchmark: issue rate of DEPENDENT v_pk_fma_f32 chains on gfx950 -- C independent chains per
// wave, W waves per SIMD (the cubic-spline recursions of apply_shift_kernel are one chain per thread at 4 waves).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v2f __attribute__((ext_vector_type(2)));

template <int C, bool PK>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  v2f r[C];
  for (int i = 0; i < C; ++i) r[i] = (v2f){threadIdx.x * 0.001f + i + 1.f, 2.f + i};
  v2f m = {out[0], out[0]}, c = {out[1], out[1]};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int rep = 0; rep < 64 / C; ++rep) {
#pragma unroll
      for (int i = 0; i < C; ++i) {
        if (PK) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(m), "v"(c));
        else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[i].x) : "v"(m.x), "v"(c.x));
      }
    }
  }
  float s = 0;
  for (int i = 0; i < C; ++i) s += r[i].x + r[i].y;
  out[2 + blockIdx.x * 256 + threadIdx.x] = s;
}

template <int C, bool PK>
static void run(float* out, int w) {
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  const int iters = 4000;
  float ms = 0;
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(a);
    k<C, PK><<<256 * w, 256>>>(out, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    hipEventElapsedTime(&ms, a, b);
  }
  // per SIMD: w waves each issue iters * 64 instructions
  const double ns = ms * 1e6 / ((double)iters * 64 * w);
  printf("{\"op\": \"%s\", \"chains_per_wave\": %d, \"waves_per_simd\": %d, \"ns_per_instruction_per_simd\": %.3f, \"ns_per_dependent_step\": %.3f}\n",
         PK ? "v_pk_fma_f32" : "v_fma_f32", C, w, ns, ns * w * C);
}

int main() {
  float* out;
  hipMalloc(&out, (2 + 256 * 8 * 256) * 4);
  float h[2] = {0.999f, 0.001f};
  hipMemcpy(out, h, 8, hipMemcpyHostToDevice);
  for (int w : {1, 2, 4, 8}) {
    run<1, true>(out, w);
    run<2, true>(out, w);
    run<4, true>(out, w);
    run<1, false>(out, w);
    run<2, false>(out, w);
    run<4, false>(out, w);
  }
  return 0;
}
