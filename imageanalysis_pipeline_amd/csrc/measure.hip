// Per-label measurement: LDS-privatised segmented reduction keyed by label id.
//
// Replaces the O(labels x channels) full-frame compares of analysis.py:80-98 (mean
// intensity), analysis.py:164-173 (centroid) and experiments.py:200-201 (area) by ONE pass
// over the label plane and the C image planes.  All sums are exact integers (u32 partials
// per workgroup chunk, u64 in HBM), so mean = (f64)sum/(f64)count reproduces numpy's
// nanmean(where=) bit for bit.
//
// Kernel shape (HBM-bound, 2 + 2/C algorithmic bytes per stack pixel):
//   grid = (chunks per frame, T); one 256-thread workgroup owns CHUNK consecutive pixels of
//   the flattened (z,y,x) index space of one frame.  Each lane loads its 8 x 8 labels (16 B at a time, all
//   requested at once); vectors of eight background pixels are only counted (their image pixels are never
//   loaded), the others are queued in LDS and then visited by all lanes, one vector each (a round of a wave in
//   which half the lanes hold a labelled vector is visited in place: dense masks): 8 pixels per channel,
//   run-length aggregation of equal neighbouring labels in registers, the run added to an open-addressing hash
//   table in LDS (ds_add_u32); the table is flushed with one 64-bit global atomic per (label, field) per
//   workgroup.  Label 0 (background, the bulk of the pixels) never touches the table: it is counted per lane and
//   reduced with wave shuffles.
#include "common.h"

namespace iap {

constexpr int MEAS_THREADS = 256;
constexpr int MEAS_ITERS = 8;
constexpr int MEAS_HASH_LOG2 = 9;
constexpr int MEAS_HASH = 1 << MEAS_HASH_LOG2;  // slots per workgroup
constexpr int MEAS_DIRECT = 32;   // labelled vectors in a wave's round from which the round is visited in place (16 .. 48: no difference)

template <int NC>
struct MeasTable {
  unsigned key[MEAS_HASH];
  unsigned cnt[MEAS_HASH];
  unsigned sz[MEAS_HASH];
  unsigned sy[MEAS_HASH];
  unsigned sx[MEAS_HASH];
  unsigned si[NC][MEAS_HASH];
  unsigned bg;
  unsigned mx;   // largest label the workgroup saw (only kept when the caller asks for the frame maxima)
  unsigned qn;   // queued vectors
  unsigned short queue[MEAS_THREADS * MEAS_ITERS];   // label-carrying vectors of the chunk (index within the chunk)
};

template <int NC, bool COORDS>
__device__ __forceinline__ void meas_flush_run(MeasTable<NC>& tb, unsigned lab, unsigned cnt, unsigned sz,
                                               unsigned sy, unsigned sx, const unsigned (&si)[NC], int L,
                                               unsigned long long* __restrict__ acc_frame, int F, int c0,
                                               unsigned& bg) {
  if (cnt == 0) return;
  if (lab == 0) {
    bg += cnt;
    return;
  }
  if (lab >= (unsigned)L) return;
  unsigned h = (lab * 2654435761u) >> (32 - MEAS_HASH_LOG2);
  int probes = 0;
  for (; probes < MEAS_HASH; ++probes) {
    unsigned k = atomicCAS(&tb.key[h], 0u, lab);
    if (k == 0u || k == lab) break;
    h = (h + 1) & (MEAS_HASH - 1);
  }
  if (probes < MEAS_HASH) {
    if (COORDS) {
      atomicAdd(&tb.cnt[h], cnt);
      atomicAdd(&tb.sz[h], sz);
      atomicAdd(&tb.sy[h], sy);
      atomicAdd(&tb.sx[h], sx);
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) atomicAdd(&tb.si[c][h], si[c]);
  } else {  // table full (more than MEAS_HASH distinct labels in one chunk): straight to HBM
    unsigned long long* a = acc_frame + (size_t)lab * F;
    if (COORDS) {
      atomicAdd(a + 0, (unsigned long long)cnt);
      atomicAdd(a + 1, (unsigned long long)sz);
      atomicAdd(a + 2, (unsigned long long)sy);
      atomicAdd(a + 3, (unsigned long long)sx);
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) atomicAdd(a + 4 + c0 + c, (unsigned long long)si[c]);
  }
}

// VEC = 8: 16-byte lane loads (needs ZYX % 8 == 0); VEC = 1: scalar fallback for odd sizes.
template <int NC, bool COORDS, int VEC>
__global__ __launch_bounds__(MEAS_THREADS) void measure_kernel(
    const uint16_t* __restrict__ img, const uint16_t* __restrict__ labels, unsigned zyx, int C, int c0,
    int Y, int X, int L, unsigned long long* __restrict__ acc, unsigned* __restrict__ frame_max) {
  __shared__ MeasTable<NC> tb;
  const int tid = threadIdx.x;
  const unsigned t = blockIdx.y;
  const int F = 4 + C;
  for (int s = tid; s < MEAS_HASH; s += MEAS_THREADS) {
    tb.key[s] = 0;
    tb.cnt[s] = 0;
    tb.sz[s] = 0;
    tb.sy[s] = 0;
    tb.sx[s] = 0;
#pragma unroll
    for (int c = 0; c < NC; ++c) tb.si[c][s] = 0;
  }
  if (tid == 0) {
    tb.bg = 0;
    tb.mx = 0;
    tb.qn = 0;
  }
  __syncthreads();

  const uint16_t* lab_f = labels + (size_t)t * zyx;
  const uint16_t* img_f = img + ((size_t)t * C + c0) * zyx;
  unsigned long long* acc_f = acc + (size_t)t * L * F;
  const unsigned chunk = MEAS_THREADS * MEAS_ITERS * VEC;
  const unsigned chunk0 = blockIdx.x * chunk;
  const unsigned yx = (unsigned)Y * (unsigned)X;
  unsigned bg = 0, mx = 0;

  // One vector (VEC pixels from flat index p on) into the table: image loads, run-length aggregation, table adds.
  auto visit = [&](unsigned p) {
    uint16_t lv[VEC];
    uint16_t iv[NC][VEC];
    if (VEC == 8) {
      u16x8 l8 = load8(lab_f + p);   // (again: a cache hit, and only for the vectors with a label)
#pragma unroll
      for (int i = 0; i < 8; ++i) lv[i] = l8.v[i];
      if (frame_max) {
#pragma unroll
        for (int i = 0; i < 8; ++i) mx = max(mx, (unsigned)lv[i]);
      }
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        u16x8 i8 = load8_nt(img_f + (size_t)c * zyx + p);   // image planes stream through once
#pragma unroll
        for (int i = 0; i < 8; ++i) iv[c][i] = i8.v[i];
      }
    } else {
      lv[0] = lab_f[p];
      if (frame_max) mx = max(mx, (unsigned)lv[0]);
#pragma unroll
      for (int c = 0; c < NC; ++c) iv[c][0] = img_f[(size_t)c * zyx + p];
    }
    unsigned z = p / yx;
    unsigned rem = p - z * yx;
    unsigned y = rem / (unsigned)X;
    unsigned x = rem - y * (unsigned)X;

    unsigned cur = lv[0], cnt = 0, sz = 0, sy = 0, sx = 0;
    unsigned si[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) si[c] = 0;
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      unsigned lab = lv[i];
      if (lab != cur) {
        meas_flush_run<NC, COORDS>(tb, cur, cnt, sz, sy, sx, si, L, acc_f, F, c0, bg);
        cur = lab;
        cnt = sz = sy = sx = 0;
#pragma unroll
        for (int c = 0; c < NC; ++c) si[c] = 0;
      }
      cnt += 1;
      sz += z;
      sy += y;
      sx += x;
#pragma unroll
      for (int c = 0; c < NC; ++c) si[c] += iv[c][i];
      if (++x == (unsigned)X) {  // a vector may straddle a row (X % 8 != 0) or a plane
        x = 0;
        if (++y == (unsigned)Y) {
          y = 0;
          ++z;
        }
      }
    }
    meas_flush_run<NC, COORDS>(tb, cur, cnt, sz, sy, sx, si, L, acc_f, F, c0, bg);
  };

  if (VEC == 8) {
    // The lane's label vectors of the whole chunk are requested at once (MEAS_ITERS 16-byte loads in flight per
    // lane).  Eight background pixels need nothing of the image, so it is not loaded either -- on sparse masks
    // (C2: 2.5 % of the pixels carry a label) most image lines are never fetched.  The vectors that do carry a
    // label are queued in LDS and visited afterwards by ALL lanes, one vector each: the label-carrying vectors
    // cluster (a blob covers two or three neighbouring lanes of a wave), and visiting them where they were found
    // ran the long path -- dependent image loads, eight steps of run-length bookkeeping, table atomics -- with two
    // or three lanes of a wave active, once per round of almost every wave.
    unsigned todo = 0;       // bit `it`: the lane's vector of round `it` carries a label
    {
      u16x8 l8[MEAS_ITERS];
#pragma unroll
      for (int it = 0; it < MEAS_ITERS; ++it) {
        const unsigned p = chunk0 + (it * MEAS_THREADS + tid) * VEC;
        if (p < zyx) l8[it] = load8(lab_f + p);
      }
#pragma unroll
      for (int it = 0; it < MEAS_ITERS; ++it) {
        const unsigned p = chunk0 + (it * MEAS_THREADS + tid) * VEC;
        if (p < zyx) {
          iap_u32x4 lw;
          __builtin_memcpy(&lw, &l8[it], 16);
          if ((lw.x | lw.y | lw.z | lw.w) == 0u) bg += 8;
          else todo |= 1u << it;
        }
      }
    }
    // dense masks: a round in which at least MEAS_DIRECT lanes of the wave hold a labelled vector is visited in
    // place (the lanes are busy anyway, and the queue would only add its traffic)
#pragma unroll 1
    for (int it = 0; it < MEAS_ITERS; ++it) {
      const bool mine = (todo >> it) & 1u;
      if (__builtin_popcountll(__ballot(mine)) >= MEAS_DIRECT) {   // wave-uniform
        if (mine) visit(chunk0 + (it * MEAS_THREADS + tid) * VEC);
        todo &= ~(1u << it);
      }
    }
    if (todo) {
      unsigned k = atomicAdd(&tb.qn, (unsigned)__builtin_popcount(todo));
      while (todo) {
        tb.queue[k++] = (unsigned short)(__builtin_ctz(todo) * MEAS_THREADS + tid);
        todo &= todo - 1u;
      }
    }
    __syncthreads();
    const unsigned nq = tb.qn;
    for (unsigned i = tid; i < nq; i += MEAS_THREADS) visit(chunk0 + (unsigned)tb.queue[i] * VEC);
  } else {
    for (int it = 0; it < MEAS_ITERS; ++it) {
      const unsigned p = chunk0 + (it * MEAS_THREADS + tid) * VEC;
      if (p >= zyx) break;
      visit(p);
    }
  }

  if (COORDS) {  // background count: wave shuffle reduction, one LDS add per wave
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) bg += __shfl_down(bg, o, 64);
    if ((tid & 63) == 0 && bg) atomicAdd(&tb.bg, bg);
  }
  if (frame_max) {   // uniform
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = max(mx, (unsigned)__shfl_down(mx, o, 64));
    if ((tid & 63) == 0 && mx) atomicMax(&tb.mx, mx);
  }
  __syncthreads();
  if (frame_max && tid == 0 && tb.mx) atomicMax(frame_max + t, tb.mx);
  for (int s = tid; s < MEAS_HASH; s += MEAS_THREADS) {
    unsigned lab = tb.key[s];
    if (lab) {
      unsigned long long* a = acc_f + (size_t)lab * F;
      if (COORDS) {
        atomicAdd(a + 0, (unsigned long long)tb.cnt[s]);
        if (tb.sz[s]) atomicAdd(a + 1, (unsigned long long)tb.sz[s]);
        atomicAdd(a + 2, (unsigned long long)tb.sy[s]);
        atomicAdd(a + 3, (unsigned long long)tb.sx[s]);
      }
#pragma unroll
      for (int c = 0; c < NC; ++c) atomicAdd(a + 4 + c0 + c, (unsigned long long)tb.si[c][s]);
    }
  }
  if (COORDS && tid == 0 && tb.bg && L > 0) atomicAdd(acc_f + 0, (unsigned long long)tb.bg);
}

__global__ __launch_bounds__(256) void measure_finalize_kernel(
    const unsigned long long* __restrict__ acc, long long rows, int C, long long* __restrict__ area,
    double* __restrict__ mean, double* __restrict__ cen, long long* __restrict__ cen_r) {
  long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  const int F = 4 + C;
  const unsigned long long* a = acc + r * F;
  unsigned long long n = a[0];
  area[r] = (long long)n;
  double dn = (double)n;
  for (int c = 0; c < C; ++c) mean[r * C + c] = n ? (double)a[4 + c] / dn : 0.0;
  for (int d = 0; d < 3; ++d) {
    double m = n ? (double)a[1 + d] / dn : 0.0;
    cen[r * 3 + d] = m;
    cen_r[r * 3 + d] = (long long)rint(m);  // round-half-even == Python round(np.float64)
  }
}

// `head` leading elements (fewer than 8) bring the rest of the array to a 16-byte boundary; they and the tail are
// read one by one
__global__ __launch_bounds__(256) void max_u16_kernel(const uint16_t* __restrict__ p, long long n, int head,
                                                      unsigned* __restrict__ out) {
  unsigned m = 0;
  const uint16_t* body = p + head;
  const long long nb = n - head, nv = nb >> 3;
  long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += stride) {
    u16x8 v = load8(body + i * 8);
#pragma unroll
    for (int k = 0; k < 8; ++k) m = max(m, (unsigned)v.v[k]);
  }
  if (blockIdx.x == 0) {
    if ((int)threadIdx.x < head) m = max(m, (unsigned)p[threadIdx.x]);
    if (threadIdx.x < (nb & 7)) m = max(m, (unsigned)body[nv * 8 + threadIdx.x]);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_down(m, o, 64));
  if ((threadIdx.x & 63) == 0 && m) atomicMax(out, m);
}

template <int NC, bool COORDS>
static int launch_measure(const uint16_t* img, const uint16_t* labels, int64_t T, int C, int c0, int Z,
                          int Y, int X, int L, uint64_t* acc, unsigned* frame_max, hipStream_t st) {
  unsigned zyx = (unsigned)((int64_t)Z * Y * X);
  auto* a = reinterpret_cast<unsigned long long*>(acc);
  if (zyx % 8 == 0) {
    unsigned chunk = MEAS_THREADS * MEAS_ITERS * 8;
    dim3 grid((zyx + chunk - 1) / chunk, (unsigned)T);
    { ProfScope ps_(K_MEASURE, st); hipLaunchKernelGGL((measure_kernel<NC, COORDS, 8>), grid, dim3(MEAS_THREADS), 0, st, img, labels, zyx, C,
                       c0, Y, X, L, a, frame_max); }
  } else {
    unsigned chunk = MEAS_THREADS * MEAS_ITERS;
    dim3 grid((zyx + chunk - 1) / chunk, (unsigned)T);
    { ProfScope ps_(K_MEASURE, st); hipLaunchKernelGGL((measure_kernel<NC, COORDS, 1>), grid, dim3(MEAS_THREADS), 0, st, img, labels, zyx, C,
                       c0, Y, X, L, a, frame_max); }
  }
  IAP_LAUNCH_CHECK();
  return IAP_OK;
}

}  // namespace iap

using namespace iap;

static int measure_labels(const uint16_t* img, const uint16_t* labels, int64_t T, int C, int Z, int Y, int X, int L,
                          uint64_t* acc, uint32_t* frame_max, void* stream) {
  IAP_REQUIRE(img && labels && acc, IAP_E_ARG, "iap_measure_labels_u16: null pointer");
  IAP_REQUIRE(T > 0 && C > 0 && Z > 0 && Y > 0 && X > 0 && L > 0, IAP_E_ARG,
              "iap_measure_labels_u16: non-positive size (T=%lld C=%d Z=%d Y=%d X=%d L=%d)", (long long)T, C,
              Z, Y, X, L);
  IAP_REQUIRE(L <= 65536, IAP_E_ARG, "iap_measure_labels_u16: L=%d exceeds the u16 label range", L);
  IAP_REQUIRE((int64_t)Z * Y * X < (1ll << 31), IAP_E_UNSUPPORTED,
              "iap_measure_labels_u16: Z*Y*X must be below 2^31");
  IAP_REQUIRE(T <= 65535, IAP_E_UNSUPPORTED, "iap_measure_labels_u16: T must be <= 65535 per call");
  IAP_REQUIRE(Y <= 65536, IAP_E_UNSUPPORTED, "iap_measure_labels_u16: Y must be <= 65536");
  IAP_REQUIRE(aligned16(img) && aligned16(labels) && (reinterpret_cast<uintptr_t>(acc) & 7u) == 0, IAP_E_ALIGN,
              "iap_measure_labels_u16: img/labels must be 16-byte and acc 8-byte aligned");
  hipStream_t st = as_stream(stream);
  IAP_HIP(hipMemsetAsync(acc, 0, (size_t)T * L * (4 + C) * sizeof(uint64_t), st));
  if (frame_max) IAP_HIP(hipMemsetAsync(frame_max, 0, (size_t)T * sizeof(uint32_t), st));
  // channel groups of <= 4 keep every accumulator in registers; the first group also
  // carries count and coordinate sums.
  for (int c0 = 0; c0 < C; c0 += 4) {
    int nc = C - c0 < 4 ? C - c0 : 4;
    int rc;
    bool first = c0 == 0;
#define IAP_MEAS_CASE(N)                                                                                  \
  case N:                                                                                                 \
    rc = first ? launch_measure<N, true>(img, labels, T, C, c0, Z, Y, X, L, acc, frame_max, st)           \
               : launch_measure<N, false>(img, labels, T, C, c0, Z, Y, X, L, acc, nullptr, st);           \
    break;
    switch (nc) {
      IAP_MEAS_CASE(1)
      IAP_MEAS_CASE(2)
      IAP_MEAS_CASE(3)
      IAP_MEAS_CASE(4)
      default:
        rc = IAP_E_ARG;
    }
#undef IAP_MEAS_CASE
    if (rc) return rc;
  }
  return IAP_OK;
}

extern "C" int iap_measure_labels_u16(const uint16_t* img, const uint16_t* labels, int64_t T, int C, int Z,
                                      int Y, int X, int L, uint64_t* acc, void* stream) {
  return measure_labels(img, labels, T, C, Z, Y, X, L, acc, nullptr, stream);
}

extern "C" int iap_measure_labels_max_u16(const uint16_t* img, const uint16_t* labels, int64_t T, int C, int Z,
                                          int Y, int X, int L, uint64_t* acc, uint32_t* frame_max, void* stream) {
  IAP_REQUIRE(frame_max && (reinterpret_cast<uintptr_t>(frame_max) & 3u) == 0, IAP_E_ARG,
              "iap_measure_labels_max_u16: frame_max must be a 4-byte aligned device pointer");
  return measure_labels(img, labels, T, C, Z, Y, X, L, acc, frame_max, stream);
}

extern "C" int iap_measure_finalize(const uint64_t* acc, int64_t T, int L, int C, int64_t* area, double* mean,
                                    double* centroid, int64_t* centroid_rne, void* stream) {
  IAP_REQUIRE(acc && area && mean && centroid && centroid_rne, IAP_E_ARG, "iap_measure_finalize: null pointer");
  IAP_REQUIRE(T > 0 && L > 0 && C > 0, IAP_E_ARG, "iap_measure_finalize: non-positive size");
  long long rows = (long long)T * L;
  unsigned blocks = (unsigned)((rows + 255) / 256);
  { ProfScope ps_(K_MEASURE_FIN, as_stream(stream)); hipLaunchKernelGGL(measure_finalize_kernel, dim3(blocks), dim3(256), 0, as_stream(stream),
                     reinterpret_cast<const unsigned long long*>(acc), rows, C,
                     reinterpret_cast<long long*>(area), mean, centroid, reinterpret_cast<long long*>(centroid_rne)); }
  IAP_LAUNCH_CHECK();
  return IAP_OK;
}

extern "C" int iap_max_u16(const uint16_t* labels, int64_t n, uint32_t* out, void* stream) {
  IAP_REQUIRE(labels && out, IAP_E_ARG, "iap_max_u16: null pointer");
  IAP_REQUIRE(n > 0, IAP_E_ARG, "iap_max_u16: n must be positive");
  IAP_REQUIRE((reinterpret_cast<uintptr_t>(labels) & 1u) == 0, IAP_E_ALIGN, "iap_max_u16: labels must be 2-byte aligned");
  hipStream_t st = as_stream(stream);
  IAP_HIP(hipMemsetAsync(out, 0, sizeof(uint32_t), st));
  // a frame of a stack whose planes are not a multiple of 8 pixels starts off a 16-byte boundary
  int head = (int)(((16u - (unsigned)(reinterpret_cast<uintptr_t>(labels) & 15u)) & 15u) / 2u);
  if (head > n) head = (int)n;
  long long nv = (n - head) >> 3;
  unsigned blocks = (unsigned)((nv + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  if (blocks == 0) blocks = 1;
  { ProfScope ps_(K_MAXLABEL, st); hipLaunchKernelGGL(max_u16_kernel, dim3(blocks), dim3(256), 0, st, labels, (long long)n, head, out); }
  IAP_LAUNCH_CHECK();
  return IAP_OK;
}
