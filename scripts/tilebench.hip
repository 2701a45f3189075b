// Development micro-benchmark: how fast can 2-D tiles of a 2048x2048 u16 plane stack be copied?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
struct alignas(16) v8 { uint16_t v[8]; };
template <int TX, int TY>
__global__ __launch_bounds__(256) void tilecopy(const uint16_t* in, uint16_t* out, int Y, int X, int remap) {
  unsigned bx = blockIdx.x, by = blockIdx.y;
  if (remap) {  // group 8 consecutive linear ids per XCD
    unsigned lin = by * gridDim.x + bx, n = gridDim.x * gridDim.y;
    unsigned q = n >> 3, x = lin & 7u;
    lin = x * q + (lin >> 3);
    bx = lin % gridDim.x; by = lin / gridDim.x;
  }
  const uint16_t* p = in + (size_t)blockIdx.z * Y * X;
  uint16_t* o = out + (size_t)blockIdx.z * Y * X;
  constexpr int VPR = TX / 8;
  for (int i = threadIdx.x; i < TY * VPR; i += 256) {
    int r = i / VPR, v = i % VPR;
    size_t off = (size_t)(by * TY + r) * X + bx * TX + v * 8;
    *(v8*)(o + off) = *(const v8*)(p + off);
  }
}
__global__ __launch_bounds__(256) void lincopy(const uint16_t* in, uint16_t* out, size_t nvec) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (size_t)gridDim.x * 256)
    ((v8*)out)[i] = ((const v8*)in)[i];
}
template <int TX, int TY>
float run(const uint16_t* in, uint16_t* out, int n, int remap) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  dim3 g(2048 / TX, 2048 / TY, n);
  tilecopy<TX, TY><<<g, 256>>>(in, out, 2048, 2048, remap);
  hipEventRecord(a);
  for (int k = 0; k < 5; ++k) tilecopy<TX, TY><<<g, 256>>>(in, out, 2048, 2048, remap);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); return ms / 5;
}
int main() {
  int n = 256; size_t bytes = (size_t)n * 2048 * 2048 * 2;
  uint16_t *in, *out; hipMalloc(&in, bytes); hipMalloc(&out, bytes); hipMemset(in, 1, bytes);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  lincopy<<<8192, 256>>>(in, out, bytes / 16);
  hipEventRecord(a); for (int k = 0; k < 5; ++k) lincopy<<<8192, 256>>>(in, out, bytes / 16); hipEventRecord(b);
  hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
  printf("linear        %7.3f ms %6.2f TB/s\n", ms, 2.0 * bytes / ms / 1e9);
#define R(TX, TY) for (int rm = 0; rm < 2; ++rm) { float t = run<TX, TY>(in, out, n, rm); printf("tile %4dx%-3d remap=%d %7.3f ms %6.2f TB/s\n", TX, TY, rm, t, 2.0 * bytes / t / 1e9); }
  R(64, 64) R(64, 32) R(128, 32) R(256, 16) R(512, 8) R(2048, 2) R(128, 64) R(256, 32)
  return 0;
}
