// Development micro-benchmark: issue cost of single VALU operations on gfx950 at 3 waves per SIMD, relative to
// v_fma_f32 (16 independent registers per wave, no memory traffic).   hipcc -O2 --offload-arch=gfx950
#include <hip/hip_runtime.h>
#include <stdio.h>
#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)
#define OPS(F)                                                                                                  \
  F(fma, "v_fma_f32 %0, %0, %1, %2")                                                                           \
  F(pk_fma, "v_pk_fma_f32 %0, %0, %1, %2")                                                                     \
  F(pk_add, "v_pk_add_f32 %0, %0, %1")                                                                         \
  F(pk_mul, "v_pk_mul_f32 %0, %0, %1")                                                                         \
  F(rsq, "v_rsq_f32 %0, %0")                                                                                   \
  F(mov_wave_shr, "v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1")                     \
  F(add_wave_shr, "v_add_f32_dpp %0, %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1")                 \
  F(add_wave_shl, "v_add_f32_dpp %0, %0, %1 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1")                 \
  F(add_row_shr, "v_add_f32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1")                   \
  F(cvt_sdwa, "v_cvt_f32_u32_sdwa %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1")                    \
  F(sub_sdwa, "v_sub_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD")    \
  F(cmp_sgpr, "v_cmp_gt_f32_e64 s[20:21], %0, %1")                                                             \
  F(lshl_add_u64, "v_lshl_add_u64 %0, %0, 1, %1")                                                              \
  F(mad_i64_i32, "v_mad_i64_i32 %0, s[20:21], %1, %2, %0")

typedef float v2f __attribute__((ext_vector_type(2)));
enum {
#define E(n, s) OP_##n,
  OPS(E)
#undef E
  OP_COUNT
};
static const char* names[] = {
#define N(n, s) #n,
    OPS(N)
#undef N
};

template <int OP>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  v2f r[16];
  for (int i = 0; i < 16; ++i) r[i] = (v2f){threadIdx.x * 0.001f + i + 1.f, 2.f + i};
  v2f m = {out[0], out[0]}, c = {out[1], out[1]};
  int mi = (int)out[0];
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int rep = 0; rep < 4; ++rep) {
#define ONE(i)                                                                                                  \
  if (OP == OP_fma) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[i].x) : "v"(m.x), "v"(c.x));               \
  if (OP == OP_pk_fma) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(m), "v"(c));                \
  if (OP == OP_pk_add) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(r[i]) : "v"(c));                            \
  if (OP == OP_pk_mul) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(r[i]) : "v"(m));                            \
  if (OP == OP_rsq) asm volatile("v_rsq_f32 %0, %0" : "+v"(r[i].x));                                             \
  if (OP == OP_mov_wave_shr) asm volatile("v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(r[i].x)); \
  if (OP == OP_add_wave_shr) asm volatile("v_add_f32_dpp %0, %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(r[i].x) : "v"(c.x)); \
  if (OP == OP_add_wave_shl) asm volatile("v_add_f32_dpp %0, %0, %1 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(r[i].x) : "v"(c.x)); \
  if (OP == OP_add_row_shr) asm volatile("v_add_f32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(r[i].x) : "v"(c.x)); \
  if (OP == OP_cvt_sdwa) asm volatile("v_cvt_f32_u32_sdwa %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "+v"(r[i].x)); \
  if (OP == OP_sub_sdwa) asm volatile("v_sub_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD" : "+v"(r[i].x) : "v"(mi)); \
  if (OP == OP_cmp_sgpr) asm volatile("v_cmp_gt_f32_e64 s[20:21], %0, %1" : : "v"(r[i].x), "v"(c.x) : "s20", "s21"); \
  if (OP == OP_lshl_add_u64) asm volatile("v_lshl_add_u64 %0, %0, 1, %1" : "+v"(r[i]) : "v"(c));                 \
  if (OP == OP_mad_i64_i32) asm volatile("v_mad_i64_i32 %0, s[20:21], %1, %2, %0" : "+v"(r[i]) : "v"(mi), "v"(mi) : "s20", "s21");
      REP16(ONE)
#undef ONE
    }
  }
  float s = 0;
  for (int i = 0; i < 16; ++i) s += r[i].x + r[i].y;
  out[2 + blockIdx.x * 256 + threadIdx.x] = s;
}

template <int OP>
static void run(float* out, double base_ns, double* mine) {
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  const int iters = 20000, wgs = 3;
  float ms = 0;
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(a);
    k<OP><<<256 * wgs, 256>>>(out, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    hipEventElapsedTime(&ms, a, b);
  }
  double ns = ms * 1e6 / ((double)iters * 64 * wgs);
  *mine = ns;
  printf("%-14s %.3f ms  %.2f ns per instruction per SIMD  (%.2f x v_fma_f32)\n", names[OP], ms, ns, base_ns > 0 ? ns / base_ns : 1.0);
}

int main() {
  float* out;
  hipMalloc(&out, 64 << 20);
  float init[2] = {0.999f, 0.001f};
  hipMemcpy(out, init, 8, hipMemcpyHostToDevice);
  double base = 0, t;
  run<OP_fma>(out, 0, &base);
#define R(n, s) if (OP_##n != OP_fma) run<OP_##n>(out, base, &t);
  OPS(R)
#undef R
  return 0;
}
