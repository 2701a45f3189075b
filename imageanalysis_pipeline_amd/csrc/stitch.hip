// Label-pair overlap, relabelling and the temporal "master mask" for mask stitching
// (reference: Utility.modif_stitch3D utility.py:700-738, which calls Cellpose's
// stitch3D / metrics._intersection_over_union and scipy.stats.mode).
//
//   overlap_kernel   overlap[a[i]][b[i]] += 1 over a plane pair: 8 pixels per lane, equal
//                    neighbouring keys merged in registers, the background pair (0, 0) counted by
//                    a wave reduction, the rest by global atomics (integer: order-free, exact)
//   relabel_kernel   out[i] = lut[frame][in[i]]                      (masks[i] = istitch[masks[i]])
//   mode_kernel      per pixel, the most frequent non-zero label over T frames, ties to the smaller
//                    label, 0 if the pixel is background in every frame (scipy.stats.mode with
//                    nan_policy='omit' on the zero->nan stack, then astype(int)); one thread per
//                    pixel with a small value/count table in registers; pixels with more distinct
//                    labels than the table holds are flagged and finished by the host
#include <algorithm>
#include "common.h"

namespace iap {

constexpr int MODE_SLOTS = 8;

__global__ __launch_bounds__(256) void overlap_kernel(const uint16_t* __restrict__ a, const uint16_t* __restrict__ b,
                                                      long long px, int na1, int nb1, long long mat_stride,
                                                      unsigned* __restrict__ overlap, int vec) {
  const long long pair = blockIdx.y;
  const uint16_t* pa = a + pair * px;
  const uint16_t* pb = b + pair * px;
  unsigned* mat = overlap + pair * mat_stride;
  const long long nv = vec ? px >> 3 : 0;   // planes off a 16-byte boundary (odd sizes, frames of such stacks): all scalar
  unsigned zero = 0;
  // labels beyond the caller's bounds are dropped rather than written outside the matrix
  auto add = [&](unsigned key, unsigned run) {
    const unsigned la = key >> 16, lb = key & 0xffffu;
    if (la < (unsigned)na1 && lb < (unsigned)nb1) atomicAdd(mat + (size_t)la * nb1 + lb, run);
  };
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nv; i += (long long)gridDim.x * 256) {
    const u16x8 va = load8(pa + i * 8), vb = load8(pb + i * 8);
    unsigned key = ((unsigned)va.v[0] << 16) | vb.v[0], run = 1;
#pragma unroll
    for (int k = 1; k < 8; ++k) {
      const unsigned kk = ((unsigned)va.v[k] << 16) | vb.v[k];
      if (kk == key) {
        ++run;
      } else {
        if (key == 0) zero += run;
        else add(key, run);
        key = kk;
        run = 1;
      }
    }
    if (key == 0) zero += run;
    else add(key, run);
  }
  for (long long t = nv * 8 + (long long)blockIdx.x * 256 + threadIdx.x; t < px; t += (long long)gridDim.x * 256) {
    const unsigned la = pa[t], lb = pb[t];   // pixels past the last whole vector
    if ((la | lb) == 0) ++zero;
    else add((la << 16) | lb, 1u);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) zero += __shfl_down(zero, o, 64);
  if ((threadIdx.x & 63) == 0 && zero) atomicAdd(mat, zero);
}

__global__ __launch_bounds__(256) void relabel_kernel(const uint16_t* __restrict__ in, uint16_t* __restrict__ out,
                                                      long long px, const uint16_t* __restrict__ lut, int lut_stride,
                                                      int vec) {
  const long long frame = blockIdx.y;
  const uint16_t* l = lut + frame * lut_stride;
  const uint16_t* pi = in + frame * px;
  uint16_t* po = out + frame * px;
  const long long nv = vec ? px >> 3 : 0;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nv; i += (long long)gridDim.x * 256) {
    u16x8 v = load8(pi + i * 8);
#pragma unroll
    for (int k = 0; k < 8; ++k) v.v[k] = l[v.v[k]];
    store8(po + i * 8, v);
  }
  for (long long t = nv * 8 + (long long)blockIdx.x * 256 + threadIdx.x; t < px; t += (long long)gridDim.x * 256)
    po[t] = l[pi[t]];
}

__global__ __launch_bounds__(256) void mode_kernel(const uint16_t* __restrict__ masks, int T, long long px,
                                                   uint16_t* __restrict__ out, unsigned* __restrict__ n_overflow) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= px) return;
  unsigned val[MODE_SLOTS], cnt[MODE_SLOTS];
#pragma unroll
  for (int s = 0; s < MODE_SLOTS; ++s) val[s] = cnt[s] = 0;
  bool over = false;
  for (int t = 0; t < T; ++t) {
    const unsigned v = masks[(long long)t * px + i];
    if (v == 0) continue;
    bool placed = false;
#pragma unroll
    for (int s = 0; s < MODE_SLOTS; ++s) {
      if (!placed && (cnt[s] == 0 || val[s] == v)) {
        val[s] = v;
        ++cnt[s];
        placed = true;
      }
    }
    over |= !placed;
  }
  unsigned best = 0, bc = 0;
#pragma unroll
  for (int s = 0; s < MODE_SLOTS; ++s)
    if (cnt[s] > bc || (cnt[s] == bc && bc > 0 && val[s] < best)) {
      bc = cnt[s];
      best = val[s];
    }
  if (over) {
    best = 0xffffu;  // more distinct labels than slots: the host recounts this pixel
    atomicAdd(n_overflow, 1u);
  }
  out[i] = (uint16_t)best;
}

}  // namespace iap

using namespace iap;

extern "C" int iap_label_overlap_u16(const uint16_t* a, const uint16_t* b, int64_t n, int64_t px, int na_max, int nb_max,
                                     uint32_t* overlap, void* stream) {
  IAP_REQUIRE(a && b && overlap, IAP_E_ARG, "iap_label_overlap_u16: null pointer");
  IAP_REQUIRE(n > 0 && px > 0, IAP_E_ARG, "iap_label_overlap_u16: non-positive size");
  IAP_REQUIRE(na_max >= 0 && na_max <= 65535 && nb_max >= 0 && nb_max <= 65535, IAP_E_ARG,
              "iap_label_overlap_u16: label bounds must be in 0..65535");
  const int vec = aligned16(a) && aligned16(b) && (px % 8 == 0 || n == 1);   // else the kernel's scalar path
  IAP_REQUIRE(n <= 65535, IAP_E_UNSUPPORTED, "iap_label_overlap_u16: n must be <= 65535");
  hipStream_t st = as_stream(stream);
  const size_t mat = (size_t)(na_max + 1) * (nb_max + 1);
  IAP_HIP(hipMemsetAsync(overlap, 0, (size_t)n * mat * 4, st));
  unsigned gx = (unsigned)std::min<int64_t>(((vec ? px / 8 : px) + 255) / 256, 1024);
  if (gx == 0) gx = 1;
  { ProfScope ps_(K_OTHER, st);
    hipLaunchKernelGGL(overlap_kernel, dim3(gx, (unsigned)n), dim3(256), 0, st, a, b, (long long)px, na_max + 1, nb_max + 1,
                       (long long)mat, (unsigned*)overlap, vec); }
  IAP_LAUNCH_CHECK();
  return IAP_OK;
}

extern "C" int iap_relabel_u16(const uint16_t* in, uint16_t* out, int64_t n, int64_t px, const uint16_t* lut,
                               int lut_stride, void* stream) {
  IAP_REQUIRE(in && out && lut, IAP_E_ARG, "iap_relabel_u16: null pointer");
  IAP_REQUIRE(n > 0 && px > 0 && lut_stride >= 0, IAP_E_ARG, "iap_relabel_u16: bad size");
  const int vec = aligned16(in) && aligned16(out) && (px % 8 == 0 || n == 1);   // else the kernel's scalar path
  IAP_REQUIRE(n <= 65535, IAP_E_UNSUPPORTED, "iap_relabel_u16: n must be <= 65535");
  hipStream_t st = as_stream(stream);
  unsigned gx = (unsigned)std::min<int64_t>(((vec ? px / 8 : px) + 255) / 256, 1024);
  if (gx == 0) gx = 1;
  { ProfScope ps_(K_OTHER, st);
    hipLaunchKernelGGL(relabel_kernel, dim3(gx, (unsigned)n), dim3(256), 0, st, in, out, (long long)px, lut, lut_stride,
                       vec); }
  IAP_LAUNCH_CHECK();
  return IAP_OK;
}

extern "C" int iap_label_mode_u16(const uint16_t* masks, int T, int64_t px, uint16_t* out, uint32_t* n_overflow,
                                  void* stream) {
  IAP_REQUIRE(masks && out && n_overflow, IAP_E_ARG, "iap_label_mode_u16: null pointer");
  IAP_REQUIRE(T > 0 && px > 0, IAP_E_ARG, "iap_label_mode_u16: non-positive size");
  IAP_REQUIRE((px + 255) / 256 <= 0x7fffffff, IAP_E_UNSUPPORTED, "iap_label_mode_u16: plane too large");
  hipStream_t st = as_stream(stream);
  IAP_HIP(hipMemsetAsync(n_overflow, 0, 4, st));
  { ProfScope ps_(K_OTHER, st);
    hipLaunchKernelGGL(mode_kernel, dim3((unsigned)((px + 255) / 256)), dim3(256), 0, st, masks, T, (long long)px, out,
                       (unsigned*)n_overflow); }
  IAP_LAUNCH_CHECK();
  return IAP_OK;
}
