// Development micro-benchmark: does the length of a loop body (instruction fetch) limit VALU issue on gfx950?
// The same independent v_pk_fma_f32 (8-byte encoding) / v_fma via VOP2 v_fmac_f32 (4-byte) stream, 3 waves per SIMD on
// every CU, as a loop whose body is 0.5, 4, 16 or 48 KB of code.   hipcc -O2 --offload-arch=gfx950
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v2f __attribute__((ext_vector_type(2)));
#define I8(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(m), "v"(c));
#define I4(i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(r[i].x) : "v"(m.x), "v"(c.x));
#define R16(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11) M(12) M(13) M(14) M(15)
#define R64(M) R16(M) R16(M) R16(M) R16(M)
#define R512(M) R64(M) R64(M) R64(M) R64(M) R64(M) R64(M) R64(M) R64(M)
#define R2K(M) R512(M) R512(M) R512(M) R512(M)
#define R6K(M) R2K(M) R2K(M) R2K(M)

template <int BODY, int WIDE>   // BODY: instructions per loop iteration
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  v2f r[16];
  for (int i = 0; i < 16; ++i) r[i] = (v2f){threadIdx.x * 0.001f + i + 1.f, 2.f + i};
  v2f m = {out[0], out[0]}, c = {out[1], out[1]};
  for (int it = 0; it < iters; ++it) {
    if (WIDE) {
      if (BODY == 64) { R64(I8) } else if (BODY == 512) { R512(I8) } else if (BODY == 2048) { R2K(I8) } else { R6K(I8) }
    } else {
      if (BODY == 64) { R64(I4) } else if (BODY == 512) { R512(I4) } else if (BODY == 2048) { R2K(I4) } else { R6K(I4) }
    }
  }
  float s = 0;
  for (int i = 0; i < 16; ++i) s += r[i].x + r[i].y;
  out[2 + blockIdx.x * 256 + threadIdx.x] = s;
}

template <int BODY, int WIDE>
static void run(float* out) {
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  const long long total = 1 << 21;            // instructions per wave
  const int iters = (int)(total / BODY), wgs = 3;
  float ms = 0;
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(a);
    k<BODY, WIDE><<<256 * wgs, 256>>>(out, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    hipEventElapsedTime(&ms, a, b);
  }
  printf("%s, body %5d instructions (%6d bytes): %.2f ns per instruction per SIMD\n", WIDE ? "v_pk_fma_f32 (8 B)" : "v_fmac_f32   (4 B)",
         BODY, BODY * (WIDE ? 8 : 4), ms * 1e6 / ((double)iters * BODY * wgs));
}

int main() {
  float* out;
  hipMalloc(&out, 64 << 20);
  float init[2] = {0.999f, 0.001f};
  hipMemcpy(out, init, 8, hipMemcpyHostToDevice);
  run<64, 1>(out); run<512, 1>(out); run<2048, 1>(out); run<6144, 1>(out);
  run<64, 0>(out); run<512, 0>(out); run<2048, 0>(out); run<6144, 0>(out);
  return 0;
}
