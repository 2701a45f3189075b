// Distance map of a reference mask and the per-pixel table of masked pixels
// (reference: Analysis.pixel_distance analysis.py:262-304 and
// Utility.apply_dmap utility.py:871-876 = mahotas.distance(~mask, 'euclidean')).
//
// Exact Euclidean distance transform, squared, in integers:
//   E1 edt_cols_kernel  g[y][x] = distance along the column to the nearest mask pixel (two sweeps,
//                       one thread per column, lanes along x)
//   E2 edt_rows_kernel  sq[y][x] = min_j g[y][j]^2 + (x-j)^2, one workgroup per row with the row of
//                       g in LDS; the search walks outwards from j = x and stops as soon as the
//                       horizontal offset alone exceeds the best value found (exact, no parabola
//                       stack; cost ~ 2 * final distance per pixel)
// Masked gather (numpy boolean indexing order = row-major), deterministic:
//   G1 mask_count_kernel   selected pixels per 4096-pixel chunk
//   G2 mask_scan_kernel    exclusive scan of the chunk counts, one workgroup per frame
//   G3 mask_gather_kernel  wave ballots + chunk offset -> position; writes rint(sqrt(sq) * pix) and
//                          the channel values
#include <math.h>
#include "common.h"

namespace iap {

constexpr int EDT_INF = 30000;        // "no mask pixel in this column": 30000^2 + 4096^2 < 2^31
constexpr int GATHER_CHUNK = 4096;    // pixels per workgroup of the count / gather kernels

__global__ __launch_bounds__(256) void edt_cols_kernel(const uint16_t* __restrict__ mask, int* __restrict__ g, int Y,
                                                       int X) {
  const int x = blockIdx.x * 256 + threadIdx.x;
  if (x >= X) return;
  const uint16_t* m = mask + (size_t)blockIdx.y * Y * X + x;
  int* gp = g + (size_t)blockIdx.y * Y * X + x;
  int d = EDT_INF;
  for (int y = 0; y < Y; ++y) {
    d = m[(size_t)y * X] ? 0 : min(d + 1, EDT_INF);
    gp[(size_t)y * X] = d;
  }
  d = EDT_INF;
  for (int y = Y - 1; y >= 0; --y) {
    const int f = gp[(size_t)y * X];
    d = min(f, min(d + 1, EDT_INF));
    gp[(size_t)y * X] = d;
  }
}

__global__ __launch_bounds__(256) void edt_rows_kernel(const int* __restrict__ g, int* __restrict__ sq, int X) {
  extern __shared__ int grow[];
  const size_t row = blockIdx.x;
  const int* gr = g + row * X;
  for (int i = threadIdx.x; i < X; i += 256) grow[i] = gr[i];
  __syncthreads();
  for (int x = threadIdx.x; x < X; x += 256) {
    const int g0 = grow[x];
    int best = g0 * g0;
    const int kmax = max(x, X - 1 - x);
    for (int k = 1; k <= kmax && k * k < best; ++k) {
      if (x - k >= 0) {
        const int a = grow[x - k];
        best = min(best, a * a + k * k);
      }
      if (x + k < X) {
        const int b = grow[x + k];
        best = min(best, b * b + k * k);
      }
    }
    sq[row * X + x] = best;
  }
}

__global__ __launch_bounds__(256) void mask_count_kernel(const uint16_t* __restrict__ mask, long long px,
                                                         int nchunks, int* __restrict__ counts) {
  __shared__ int wsum[4];
  const long long base = (long long)blockIdx.y * px + (long long)blockIdx.x * GATHER_CHUNK;
  const long long end = min((long long)(blockIdx.y + 1) * px, base + GATHER_CHUNK);
  int c = 0;
  for (long long i = base + threadIdx.x; i < end; i += 256) c += mask[i] != 0;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o, 64);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) counts[(size_t)blockIdx.y * nchunks + blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// counts[frame][chunk] -> exclusive prefix in place; totals[frame] = number of selected pixels
__global__ __launch_bounds__(256) void mask_scan_kernel(int* __restrict__ counts, int nchunks,
                                                        long long* __restrict__ totals) {
  __shared__ int part[256];
  int* c = counts + (size_t)blockIdx.x * nchunks;
  const int per = (nchunks + 255) / 256;
  const int lo = threadIdx.x * per, hi = min(lo + per, nchunks);
  int s = 0;
  for (int i = lo; i < hi; ++i) s += c[i];
  part[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    int run = 0;
    for (int i = 0; i < 256; ++i) {
      const int v = part[i];
      part[i] = run;
      run += v;
    }
    totals[blockIdx.x] = run;
  }
  __syncthreads();
  int run = part[threadIdx.x];
  for (int i = lo; i < hi; ++i) {
    const int v = c[i];
    c[i] = run;
    run += v;
  }
}

__global__ __launch_bounds__(256) void mask_gather_kernel(const uint16_t* __restrict__ mask, const int* __restrict__ sq,
                                                          const uint16_t* __restrict__ img, long long img_frame_stride,
                                                          long long img_chan_stride, int C, long long px, int nchunks,
                                                          const int* __restrict__ offs,
                                                          const long long* __restrict__ frame_base, double pix_size,
                                                          long long total, int* __restrict__ out_d,
                                                          uint16_t* __restrict__ out_v) {
  __shared__ int wcnt[4];
  const int frame = blockIdx.y;
  const long long pbase = (long long)frame * px;
  const long long cbase = (long long)blockIdx.x * GATHER_CHUNK;
  long long pos = frame_base[frame] + offs[(size_t)frame * nchunks + blockIdx.x];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int it = 0; it < GATHER_CHUNK / 256; ++it) {     // 256 consecutive pixels per iteration keep row-major order
    const long long i = cbase + it * 256 + threadIdx.x;
    const bool sel = i < px && mask[pbase + i] != 0;
    const unsigned long long b = __ballot(sel);
    if (lane == 0) wcnt[wave] = __popcll(b);
    __syncthreads();
    int before = 0, all = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      before += w < wave ? wcnt[w] : 0;
      all += wcnt[w];
    }
    if (sel) {
      const long long o = pos + before + __popcll(b & ((1ull << lane) - 1ull));
      if (out_d) out_d[o] = sq ? (int)rint(sqrt((double)sq[pbase + i]) * pix_size) : 0;
      for (int c = 0; c < C; ++c)
        out_v[(long long)c * total + o] = img[(long long)frame * img_frame_stride + (long long)c * img_chan_stride + i];
    }
    pos += all;
    __syncthreads();
  }
}

}  // namespace iap

using namespace iap;

extern "C" size_t iap_edt_ws_bytes(int64_t n, int Y, int X) {
  if (n <= 0 || Y <= 0 || X <= 0) return 0;
  return (size_t)n * Y * X * 4 + 256;
}

extern "C" int iap_edt_sq_u16(const uint16_t* mask, int32_t* sq_out, int64_t n, int Y, int X, void* ws,
                              size_t ws_bytes, void* stream) {
  IAP_REQUIRE(mask && sq_out, IAP_E_ARG, "iap_edt_sq_u16: null pointer");
  IAP_REQUIRE(n > 0 && Y > 0 && X > 0, IAP_E_ARG, "iap_edt_sq_u16: non-positive size");
  IAP_REQUIRE(Y <= 4096 && X <= 4096, IAP_E_UNSUPPORTED, "iap_edt_sq_u16: planes up to 4096 x 4096");
  IAP_REQUIRE(n <= 65535 && (int64_t)n * Y <= 0x7fffffff, IAP_E_UNSUPPORTED, "iap_edt_sq_u16: too many planes per call");
  IAP_REQUIRE(ws && ws_bytes >= iap_edt_ws_bytes(n, Y, X), IAP_E_WS, "iap_edt_sq_u16: workspace of %zu bytes, need %zu",
              ws_bytes, iap_edt_ws_bytes(n, Y, X));
  hipStream_t st = as_stream(stream);
  int* g = (int*)ws;
  { ProfScope ps_(K_OTHER, st);
    hipLaunchKernelGGL(edt_cols_kernel, dim3((X + 255) / 256, (unsigned)n), dim3(256), 0, st, mask, g, Y, X);
    hipLaunchKernelGGL(edt_rows_kernel, dim3((unsigned)(n * Y)), dim3(256), (size_t)X * 4, st, (const int*)g, sq_out, X); }
  IAP_LAUNCH_CHECK();
  return IAP_OK;
}

extern "C" int iap_mask_chunks(int Y, int X) { return (int)(((int64_t)Y * X + GATHER_CHUNK - 1) / GATHER_CHUNK); }

extern "C" int iap_mask_count_u16(const uint16_t* mask, int64_t n, int Y, int X, int32_t* chunk_offsets,
                                  int64_t* totals, void* stream) {
  IAP_REQUIRE(mask && chunk_offsets && totals, IAP_E_ARG, "iap_mask_count_u16: null pointer");
  IAP_REQUIRE(n > 0 && Y > 0 && X > 0, IAP_E_ARG, "iap_mask_count_u16: non-positive size");
  IAP_REQUIRE(n <= 65535 && (int64_t)Y * X < (1ll << 31), IAP_E_UNSUPPORTED, "iap_mask_count_u16: size too large");
  hipStream_t st = as_stream(stream);
  const int nch = iap_mask_chunks(Y, X);
  { ProfScope ps_(K_OTHER, st);
    hipLaunchKernelGGL(mask_count_kernel, dim3(nch, (unsigned)n), dim3(256), 0, st, mask, (long long)Y * X, nch,
                       chunk_offsets);
    hipLaunchKernelGGL(mask_scan_kernel, dim3((unsigned)n), dim3(256), 0, st, chunk_offsets, nch, (long long*)totals); }
  IAP_LAUNCH_CHECK();
  return IAP_OK;
}

extern "C" int iap_masked_gather_u16(const uint16_t* mask, const int32_t* sq, const uint16_t* img, int64_t img_frame_stride,
                                     int64_t img_chan_stride, int C, int64_t n, int Y, int X, const int32_t* chunk_offsets,
                                     const int64_t* frame_base, double pix_size, int64_t total, int32_t* out_d,
                                     uint16_t* out_v, void* stream) {
  IAP_REQUIRE(mask && chunk_offsets && frame_base, IAP_E_ARG, "iap_masked_gather_u16: null pointer");
  IAP_REQUIRE(C >= 0 && (C == 0 || (img && out_v)), IAP_E_ARG, "iap_masked_gather_u16: channels without image or output");
  IAP_REQUIRE(n > 0 && Y > 0 && X > 0 && total >= 0, IAP_E_ARG, "iap_masked_gather_u16: bad size");
  IAP_REQUIRE(n <= 65535 && (int64_t)Y * X < (1ll << 31), IAP_E_UNSUPPORTED, "iap_masked_gather_u16: size too large");
  if (total == 0) return IAP_OK;
  hipStream_t st = as_stream(stream);
  const int nch = iap_mask_chunks(Y, X);
  { ProfScope ps_(K_OTHER, st);
    hipLaunchKernelGGL(mask_gather_kernel, dim3(nch, (unsigned)n), dim3(256), 0, st, mask, (const int*)sq, img,
                       (long long)img_frame_stride, (long long)img_chan_stride, C, (long long)Y * X, nch,
                       (const int*)chunk_offsets, (const long long*)frame_base, pix_size, (long long)total, out_d, out_v); }
  IAP_LAUNCH_CHECK();
  return IAP_OK;
}
