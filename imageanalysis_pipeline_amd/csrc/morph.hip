// Label-preserving erosion and dilation with a disk for the compartment masks
// (reference: Utility.erode_mask / dilate_mask utility.py:776-826, used by
// Exp_Indiv.cell_compart experiments.py:236-271).
//
// erode_mask sums cv2.erode(mask == obj ? obj : 0, disk(r)) over the objects: a pixel keeps its label
// iff every pixel of the disk round it that lies inside the image carries the same label (cv2's default
// border value for erosion is +inf: the image border does not erode).  dilate_mask is cv2.dilate of the
// label image: the largest label inside the disk (border -inf).  disk(r) is skimage's: dx^2 + dy^2 <= r^2.
// One workgroup computes a 32 x 32 tile from an LDS copy with an r-pixel halo; every thread walks
// the row spans of the disk (exact, integer).
#include <algorithm>
#include "common.h"

namespace iap {

constexpr int MO_T = 32;        // tile edge
constexpr int MO_RMAX = 31;     // largest radius (LDS tile (32 + 62)^2 u16 = 17.7 KB)

template <bool ERODE>
__global__ __launch_bounds__(256) void label_morph_kernel(const uint16_t* __restrict__ in, uint16_t* __restrict__ out,
                                                          int Y, int X, int r) {
  __shared__ uint16_t tile[(MO_T + 2 * MO_RMAX) * (MO_T + 2 * MO_RMAX)];
  __shared__ int halfw[2 * MO_RMAX + 1];
  const int W = MO_T + 2 * r;
  const int oy = blockIdx.y * MO_T, ox = blockIdx.x * MO_T;
  const uint16_t* p = in + (size_t)blockIdx.z * Y * X;
  uint16_t* po = out + (size_t)blockIdx.z * Y * X;
  // positions outside the image are never tested below (the spans are clipped); they are filled with 0
  for (int i = threadIdx.x; i < W * W; i += 256) {
    const int ty = i / W, tx = i - ty * W;
    const int yy = oy + ty - r, xx = ox + tx - r;
    tile[i] = (yy >= 0 && yy < Y && xx >= 0 && xx < X) ? p[(size_t)yy * X + xx] : (uint16_t)0;
  }
  for (int i = threadIdx.x; i <= 2 * r; i += 256) {
    const int dy = i - r;
    int w = 0;
    while ((w + 1) * (w + 1) + dy * dy <= r * r) ++w;
    halfw[i] = w;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < MO_T * MO_T; i += 256) {
    const int ty = i / MO_T, tx = i - ty * MO_T;
    const int yy = oy + ty, xx = ox + tx;
    if (yy >= Y || xx >= X) continue;
    const unsigned self = tile[(ty + r) * W + tx + r];
    unsigned res;
    if (ERODE) {
      bool keep = self != 0;
      if (keep) {
        for (int dy = -r; dy <= r; ++dy) {
          if (yy + dy < 0 || yy + dy >= Y) continue;     // outside the image: does not erode
          const int w = halfw[dy + r];
          const uint16_t* row = tile + (ty + r + dy) * W + tx + r;
          const int lo = max(-w, -xx), hi = min(w, X - 1 - xx);
          for (int dx = lo; dx <= hi; ++dx) keep &= row[dx] == self;
        }
      }
      res = keep ? self : 0u;
    } else {
      unsigned m = self;
      for (int dy = -r; dy <= r; ++dy) {
        if (yy + dy < 0 || yy + dy >= Y) continue;
        const int w = halfw[dy + r];
        const uint16_t* row = tile + (ty + r + dy) * W + tx + r;
        const int lo = max(-w, -xx), hi = min(w, X - 1 - xx);
        for (int dx = lo; dx <= hi; ++dx) m = max(m, (unsigned)row[dx]);
      }
      res = m;
    }
    po[(size_t)yy * X + xx] = (uint16_t)res;
  }
}

template <bool ERODE>
static int launch_morph(const uint16_t* in, uint16_t* out, int64_t n, int Y, int X, int radius, int iterations, void* ws,
                        size_t ws_bytes, hipStream_t st, const char* fn) {
  IAP_REQUIRE(in && out, IAP_E_ARG, "%s: null pointer", fn);
  IAP_REQUIRE(n > 0 && Y > 0 && X > 0, IAP_E_ARG, "%s: non-positive size", fn);
  IAP_REQUIRE(radius >= 0 && radius <= MO_RMAX, IAP_E_UNSUPPORTED, "%s: radius must be in 0..%d", fn, MO_RMAX);
  IAP_REQUIRE(iterations >= 1, IAP_E_ARG, "%s: iterations must be >= 1", fn);
  IAP_REQUIRE(n <= 65535, IAP_E_UNSUPPORTED, "%s: n must be <= 65535", fn);
  const size_t bytes = (size_t)n * Y * X * 2;
  IAP_REQUIRE(in != out, IAP_E_ARG, "%s: in-place operation is not supported", fn);
  IAP_REQUIRE(iterations == 1 || (ws && ws != (const void*)in && ws != (void*)out && ws_bytes >= bytes), IAP_E_WS,
              "%s: %d iterations need a separate workspace of %zu bytes", fn, iterations, bytes);
  dim3 grid((X + MO_T - 1) / MO_T, (Y + MO_T - 1) / MO_T, (unsigned)n);
  // passes alternate between `out` and the workspace so that the last one lands in `out`
  const uint16_t* src = in;
  for (int it = 0; it < iterations; ++it) {
    uint16_t* dst = ((iterations - 1 - it) % 2 == 0) ? out : (uint16_t*)ws;
    { ProfScope ps_(K_OTHER, st);
      hipLaunchKernelGGL((label_morph_kernel<ERODE>), grid, dim3(256), 0, st, src, dst, Y, X, radius); }
    src = dst;
  }
  IAP_LAUNCH_CHECK();
  return IAP_OK;
}

// out = a - b on u16 label planes (membrane = full mask - eroded mask, experiments.py:259), 8 px per lane
__global__ __launch_bounds__(256) void label_sub_kernel(const uint16_t* __restrict__ a, const uint16_t* __restrict__ b,
                                                        uint16_t* __restrict__ out, long long n) {
  const long long nv = n >> 3;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nv; i += (long long)gridDim.x * 256) {
    const u16x8 va = load8(a + i * 8), vb = load8(b + i * 8);
    u16x8 vo;
#pragma unroll
    for (int k = 0; k < 8; ++k) vo.v[k] = (uint16_t)(va.v[k] - vb.v[k]);
    store8(out + i * 8, vo);
  }
  if (blockIdx.x == 0) {
    const long long t = nv * 8 + threadIdx.x;
    if (t < n) out[t] = (uint16_t)(a[t] - b[t]);
  }
}

}  // namespace iap

using namespace iap;

extern "C" int iap_label_sub_u16(const uint16_t* a, const uint16_t* b, uint16_t* out, int64_t n, void* stream) {
  IAP_REQUIRE(a && b && out, IAP_E_ARG, "iap_label_sub_u16: null pointer");
  IAP_REQUIRE(n > 0, IAP_E_ARG, "iap_label_sub_u16: non-positive size");
  IAP_REQUIRE(aligned16(a) && aligned16(b) && aligned16(out), IAP_E_ALIGN, "iap_label_sub_u16: buffers must be 16-byte aligned");
  hipStream_t st = as_stream(stream);
  unsigned g = (unsigned)std::min<int64_t>((n / 8 + 255) / 256, 4096);
  if (g == 0) g = 1;
  { ProfScope ps_(K_OTHER, st);
    hipLaunchKernelGGL(label_sub_kernel, dim3(g), dim3(256), 0, st, a, b, out, (long long)n); }
  IAP_LAUNCH_CHECK();
  return IAP_OK;
}

extern "C" int iap_label_erode_u16(const uint16_t* in, uint16_t* out, int64_t n, int Y, int X, int radius, int iterations,
                                   void* ws, size_t ws_bytes, void* stream) {
  return launch_morph<true>(in, out, n, Y, X, radius, iterations, ws, ws_bytes, as_stream(stream), "iap_label_erode_u16");
}

extern "C" int iap_label_dilate_u16(const uint16_t* in, uint16_t* out, int64_t n, int Y, int X, int radius, int iterations,
                                    void* ws, size_t ws_bytes, void* stream) {
  return launch_morph<false>(in, out, n, Y, X, radius, iterations, ws, ws_bytes, as_stream(stream), "iap_label_dilate_u16");
}
