// Background subtraction and smoothing kernels (reference: pipeline/utility.py:213-270, 433-468).
//
// SMO background (utility.py:252-264):  per plane
//   pass 1  smo_march_kernel<HIST>   u16 plane -> SMO value per pixel -> pixels whose SMO value is
//           below the threshold go into a 65 536-bin histogram of the plane (+ 256-bin coarse one)
//   pass 2  median_kernel            2*median from the two histograms (half-integers possible)
//   pass 3  sub_med_kernel           out = max(2*v - med2, 0) >> 1   (== trunc(max(v - med, 0)))
// Planes are processed in batches small enough that pass 3 re-reads its planes from the
// 256 MiB Infinity Cache; algorithmic HBM bytes are 4 per pixel (one read + one write).
//
// smo_march_kernel: one WAVE marches down a strip of 256 columns (4 per lane; lanes 0 and 63
// are halo, 248 columns are produced) x `band` rows.  Image rows live in registers, the
// horizontal 7-tap sums use cross-lane shuffles (no LDS, no barriers), the vertical window is
// a 7-slot register ring that is indexed statically by unrolling the march 7x.  scipy's
// 'reflect' boundary of uniform_filter is reproduced by seeding / feeding the ring with the
// mirrored rows and by mirroring the halo lanes' columns; np.gradient's one-sided edge
// differences are reproduced as well.  The per-wave histogram is privatised in LDS over a
// 1024-value window around the strip's first pixels (background is narrow); anything outside
// goes straight to HBM atomics.
#include <stdlib.h>
#include <mutex>
#include "common.h"

namespace iap {

constexpr int SMO_COLS = 248;        // columns produced per wave
constexpr int SMO_WIN = 1024;        // LDS histogram window (values) per wave
constexpr int SMO_WAVES = 4;         // waves per workgroup (independent of each other)
constexpr int HIST_FINE = 65536;
constexpr int HIST_COARSE = 256;
constexpr int HIST_WORDS = HIST_FINE + HIST_COARSE;

enum { SMO_MODE_IMAGE = 0, SMO_MODE_HIST = 1 };

// Four neighbouring pixels of one lane as two packed pairs (px 0,1 and px 2,3): element-wise
// arithmetic on them is one v_pk_*_f32 per pair (the march kernel is VALU-issue bound).
typedef float v2f __attribute__((ext_vector_type(2)));
struct f4 {
  v2f lo, hi;
};
__device__ __forceinline__ v2f pk_fma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }

// raw (unconverted) row segment of one lane: 4 pixels
template <typename T> struct Raw4;
template <> struct Raw4<uint16_t> { uint2 w; };
template <> struct Raw4<float> { float4 w; };

template <typename T>
__device__ __forceinline__ Raw4<T> load_raw4(const T* __restrict__ row, int x, bool ok);
template <>
__device__ __forceinline__ Raw4<uint16_t> load_raw4<uint16_t>(const uint16_t* __restrict__ row, int x, bool ok) {
  Raw4<uint16_t> r;
  r.w = ok ? *reinterpret_cast<const uint2*>(row + x) : make_uint2(0u, 0u);
  return r;
}
template <>
__device__ __forceinline__ Raw4<float> load_raw4<float>(const float* __restrict__ row, int x, bool ok) {
  Raw4<float> r;
  r.w = ok ? *reinterpret_cast<const float4*>(row + x) : make_float4(0.f, 0.f, 0.f, 0.f);
  return r;
}
__device__ __forceinline__ f4 to_f4(const Raw4<uint16_t>& r) {
  f4 o;
  o.lo = (v2f){(float)(r.w.x & 0xffffu), (float)(r.w.x >> 16)};
  o.hi = (v2f){(float)(r.w.y & 0xffffu), (float)(r.w.y >> 16)};
  return o;
}
__device__ __forceinline__ f4 to_f4(const Raw4<float>& r) {
  f4 o;
  o.lo = (v2f){r.w.x, r.w.y};
  o.hi = (v2f){r.w.z, r.w.w};
  return o;
}
__device__ __forceinline__ uint2 as_u2(const Raw4<uint16_t>& r) { return r.w; }
__device__ __forceinline__ uint2 as_u2(const Raw4<float>&) { return make_uint2(0u, 0u); }
template <typename T>
__device__ __forceinline__ f4 load_row4(const T* __restrict__ row, int x, bool ok) {
  return to_f4(load_raw4<T>(row, x, ok));
}

// value of lane-1 / lane+1 through DPP wave shifts (one VALU move each, no LDS traffic); the end
// lanes, which have no neighbour, receive 0 (they are halo lanes whose results are not used)
__device__ __forceinline__ float from_left(float x) {   // lane i receives lane i-1
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x138, 0xf, 0xf, true));  // wave_shr:1, lane 0 <- 0
}
__device__ __forceinline__ float from_right(float x) {  // lane i receives lane i+1
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x130, 0xf, 0xf, true));  // wave_shl:1, lane 63 <- 0
}

struct SmoGeom {
  int Y, X;
  int nstrips, nbands, band;
  int tasks_per_plane;  // nstrips * nbands rounded up to a multiple of SMO_WAVES
};

// Per-lane constants of the marching kernel's inner step.
struct SmoLane {
  bool edge_l, edge_r;  // this lane holds image column 0 (as its col 0) / X-1 (as its col 3)
  bool mir_l, mir_r;    // lane 1 of the leftmost / lane 62 of the rightmost strip: mirrored halo
  bool edge_strip;      // wave-uniform: leftmost or rightmost strip
  float fx_l, fx_r;     // 2 where the x difference of col 0 / col 3 is one-sided, else 1
};

// One row of horizontally 7-summed unit-gradient components from image rows v-1, v, v+1.
// first_row / last_row (wave-uniform): np.gradient's one-sided differences at the plane edge.
// Gradients are kept at twice np.gradient's scale (central difference without the 1/2, one-sided
// differences doubled): the factor cancels exactly in the normalisation.
__device__ __forceinline__ void smo_row(const f4& A, const f4& B, const f4& Cn, bool first_row, bool last_row,
                                        const SmoLane& ln, f4& hy, f4& hx) {
  const float bl = from_left(B.hi.y);
  const float br = from_right(B.lo.x);
  v2f gy_lo, gy_hi;
  if (first_row || last_row) {  // wave-uniform
    const f4& up = first_row ? B : A;
    const f4& dn = last_row ? B : Cn;
    gy_lo = 2.0f * (dn.lo - up.lo);
    gy_hi = 2.0f * (dn.hi - up.hi);
  } else {
    gy_lo = Cn.lo - A.lo;
    gy_hi = Cn.hi - A.hi;
  }
  // x neighbours: l = (bl, B0, B1, B2), r = (B1, B2, B3, br); one-sided (doubled) at the image edge
  const float gx0 = ln.fx_l * (B.lo.y - (ln.edge_l ? B.lo.x : bl));
  const float gx3 = ln.fx_r * ((ln.edge_r ? B.hi.y : br) - B.hi.x);
  const v2f gx_lo = {gx0, B.hi.x - B.lo.x};
  const v2f gx_hi = {B.hi.y - B.lo.y, gx3};
  // zero gradient -> 0 * rsq(tiny) = 0: the floor rides in the fused multiply-adds (n2 + 1e-30 == n2 for every
  // non-zero n2 of integer pixels, which is >= 1)
  const v2f tiny = {1e-30f, 1e-30f};
  const v2f n2_lo = pk_fma(gx_lo, gx_lo, pk_fma(gy_lo, gy_lo, tiny));
  const v2f n2_hi = pk_fma(gx_hi, gx_hi, pk_fma(gy_hi, gy_hi, tiny));
  const v2f inv_lo = {__builtin_amdgcn_rsqf(n2_lo.x), __builtin_amdgcn_rsqf(n2_lo.y)};
  const v2f inv_hi = {__builtin_amdgcn_rsqf(n2_hi.x), __builtin_amdgcn_rsqf(n2_hi.y)};
  const v2f ny_lo = gy_lo * inv_lo, ny_hi = gy_hi * inv_hi;
  const v2f nx_lo = gx_lo * inv_lo, nx_hi = gx_hi * inv_hi;
  // neighbours' columns: left lane's cols 1..3, right lane's cols 0..2
  float ly1 = from_left(ny_lo.y), ly2 = from_left(ny_hi.x), ly3 = from_left(ny_hi.y);
  float lx1 = from_left(nx_lo.y), lx2 = from_left(nx_hi.x), lx3 = from_left(nx_hi.y);
  float ry0 = from_right(ny_lo.x), ry1 = from_right(ny_lo.y), ry2 = from_right(ny_hi.x);
  float rx0 = from_right(nx_lo.x), rx1 = from_right(nx_lo.y), rx2 = from_right(nx_hi.x);
  if (ln.edge_strip) {  // wave-uniform: only the first and the last strip of a row mirror a halo
    if (ln.mir_l) {  // columns -3,-2,-1 mirror columns 2,1,0 (scipy 'reflect')
      ly1 = ny_hi.x; ly2 = ny_lo.y; ly3 = ny_lo.x;
      lx1 = nx_hi.x; lx2 = nx_lo.y; lx3 = nx_lo.x;
    }
    if (ln.mir_r) {  // columns X,X+1,X+2 mirror X-1,X-2,X-3
      ry0 = ny_hi.y; ry1 = ny_hi.x; ry2 = ny_lo.y;
      rx0 = nx_hi.y; rx1 = nx_hi.x; rx2 = nx_lo.y;
    }
  }
  // 7-sums: out0 = own + l1+l2+l3, out1 = own + l2+l3+r0, out2 = own + l3+r0+r1, out3 = own + r0+r1+r2
  const v2f s2y = ny_lo + ny_hi, s2x = nx_lo + nx_hi;
  const float sy = s2y.x + s2y.y, sx = s2x.x + s2x.y;
  const float ty = ly2 + ly3, uy = ry0 + ry1, tx = lx2 + lx3, ux = rx0 + rx1;
  hy.lo = ((v2f){ly1, ry0} + (v2f){ty, ty}) + (v2f){sy, sy};
  hy.hi = ((v2f){ly3, ry2} + (v2f){uy, uy}) + (v2f){sy, sy};
  hx.lo = ((v2f){lx1, rx0} + (v2f){tx, tx}) + (v2f){sx, sx};
  hx.hi = ((v2f){lx3, rx2} + (v2f){ux, ux}) + (v2f){sx, sx};
}

template <typename T, int MODE>
__global__ __launch_bounds__(SMO_WAVES * 64, 3) void smo_march_kernel(
    const T* __restrict__ src, const uint16_t* __restrict__ orig, float* __restrict__ smo_out,
    unsigned* __restrict__ hist, SmoGeom g, float thr2 /* (threshold*49)^2 */, float inv_area) {
  __shared__ unsigned lds_fine[SMO_WAVES][SMO_WIN];
  __shared__ unsigned lds_coarse[SMO_WAVES][HIST_COARSE];
  const int lane = threadIdx.x & 63;
  // (read back through a scalar register: the compiler cannot know that threadIdx.x >> 6 is the same for a whole
  // wave, and would otherwise carry the strip, the band, the row counter and every row address in vector registers)
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int Y = g.Y, X = g.X;
  const long long task_all = (long long)blockIdx.x * SMO_WAVES + wave;
  const long long plane = task_all / g.tasks_per_plane;
  const int task = (int)(task_all - plane * g.tasks_per_plane);
  if (task >= g.nstrips * g.nbands) return;  // padding wave (no barriers in this kernel)
  const int strip = task % g.nstrips;
  const int bandi = task / g.nstrips;
  const bool left_strip = strip == 0;
  const bool right_strip = strip == g.nstrips - 1;
  const int x0 = right_strip ? (X - 252) : (strip * SMO_COLS - 4);
  const int own_lo = strip * SMO_COLS;
  const int own_hi = right_strip ? X : own_lo + SMO_COLS;
  const int xl = x0 + 4 * lane;                 // first column of this lane
  const bool col_ok = xl >= 0 && xl + 3 < X;    // X % 4 == 0: a lane is entirely in or out
  // A lane outside the image (lane 0 of the leftmost strip, lane 63 of the rightmost) loads the image's edge columns
  // instead of nothing: its values are never used -- its neighbour mirrors its own columns (mir_l / mir_r, edge_l /
  // edge_r) and it owns no output -- and an unconditional load needs no zeroed destination and no branch per row.
  const int xl_ld = min(max(xl, 0), X - 4);
  const int y0 = bandi * g.band;
  const int yend = min(Y, y0 + g.band);
  const bool top = y0 == 0;
  const T* __restrict__ p = src + (size_t)plane * Y * X;
  const uint16_t* __restrict__ po = orig ? orig + (size_t)plane * Y * X : nullptr;
  unsigned* __restrict__ hfine = hist ? hist + (size_t)plane * HIST_WORDS : nullptr;
  unsigned* __restrict__ hcoarse = hfine ? hfine + HIST_FINE : nullptr;

  unsigned win_base = 0;
  if (MODE == SMO_MODE_HIST) {
    for (int i = lane; i < SMO_WIN; i += 64) lds_fine[wave][i] = 0;
    for (int i = lane; i < HIST_COARSE; i += 64) lds_coarse[wave][i] = 0;
    // window base from the minimum of the strip's first row (wave-uniform)
    unsigned m = 0xffffu;
    if (col_ok) {
      uint2 w = *reinterpret_cast<const uint2*>(po + (size_t)y0 * X + xl);
      m = min(min(w.x & 0xffffu, w.x >> 16), min(w.y & 0xffffu, w.y >> 16));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = min(m, (unsigned)__shfl_xor(m, o, 64));
    win_base = m > 128u ? m - 128u : 0u;
    if (win_base > (unsigned)(HIST_FINE - SMO_WIN)) win_base = HIST_FINE - SMO_WIN;
  }

  v2f ry[7][2], rx[7][2];    // vertical ring (pixel pairs), statically indexed
  v2f vy[2], vx[2];          // running vertical 7-row sums (re-summed exactly every 7th row)
  const int vbase = y0 - 3;
  const int vfirst = top ? 0 : vbase;
  SmoLane ln;
  ln.edge_l = xl == 0;
  ln.edge_r = xl + 3 == X - 1;
  ln.mir_l = left_strip && lane == 1;
  ln.mir_r = right_strip && lane == 62;
  ln.edge_strip = left_strip || right_strip;
  ln.fx_l = ln.edge_l ? 2.0f : 1.0f;
  ln.fx_r = ln.edge_r ? 2.0f : 1.0f;
  // rows v-1, v, v+1 of the row being computed; rows v+2 and v+3 are in flight, unconverted
  f4 A = load_row4<T>(p + (size_t)max(vfirst - 1, 0) * X, xl_ld, true);
  f4 B = load_row4<T>(p + (size_t)vfirst * X, xl_ld, true);
  f4 Cn = load_row4<T>(p + (size_t)min(vfirst + 1, Y - 1) * X, xl_ld, true);
  Raw4<T> Dn = load_raw4<T>(p + (size_t)min(vfirst + 2, Y - 1) * X, xl_ld, true);
  Raw4<T> En = load_raw4<T>(p + (size_t)min(vfirst + 3, Y - 1) * X, xl_ld, true);
  // u16 source == histogram source: the unconverted rows are kept in a 7-slot ring (same slot
  // numbering as ry/rx) so that emit() never re-reads the image
  constexpr bool RAWRING = (MODE == SMO_MODE_HIST) && sizeof(T) == 2;
  uint2 rawr[7];
#pragma unroll
  for (int q = 0; q < 7; ++q) rawr[q] = make_uint2(0u, 0u);
  if (RAWRING && top) {  // rows 0 and 1 live in slots 3 and 4
    rawr[3] = as_u2(load_raw4<T>(p, xl_ld, true));
    rawr[4] = as_u2(load_raw4<T>(p + (size_t)X, xl_ld, true));
  }

  // Put virtual row v into ring slot PH (PH is a compile-time constant after unrolling).
  // Rows Y, Y+1, Y+2 mirror Y-1, Y-2, Y-3, i.e. the slots filled 1, 3 and 5 steps ago.
#define SMO_FILL(PH, VROW)                                                                              \
  do {                                                                                               \
    if ((VROW) < Y) {                                                                                   \
      f4 hy_, hx_;                                                                                   \
      smo_row(A, B, Cn, (VROW) == 0, (VROW) == Y - 1, ln, hy_, hx_);                                    \
      A = B;                                                                                         \
      B = Cn;                                                                                        \
      Cn = to_f4(Dn);                                                                                \
      if (RAWRING) rawr[((PH) + 2) % 7] = as_u2(Dn);                      \
      Dn = En;                                                                                       \
      En = load_raw4<T>(p + (size_t)min((VROW) + 4, Y - 1) * X, xl_ld, true);                            \
      ry[PH][0] = hy_.lo; ry[PH][1] = hy_.hi;                                                        \
      rx[PH][0] = hx_.lo; rx[PH][1] = hx_.hi;                                                        \
    } else {                                                                                         \
      const int j_ = (VROW) - Y;                                                                        \
      _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                \
        v2f ty_ = j_ == 0 ? ry[((PH) + 6) % 7][i] : (j_ == 1 ? ry[((PH) + 4) % 7][i] : ry[((PH) + 2) % 7][i]); \
        v2f tx_ = j_ == 0 ? rx[((PH) + 6) % 7][i] : (j_ == 1 ? rx[((PH) + 4) % 7][i] : rx[((PH) + 2) % 7][i]); \
        ry[PH][i] = ty_;                                                                             \
        rx[PH][i] = tx_;                                                                             \
      }                                                                                              \
    }                                                                                                \
  } while (0)

#define SMO_RESUM()                                                                                    \
  _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                      \
    vy[i] = ((ry[0][i] + ry[1][i]) + (ry[2][i] + ry[3][i])) + ((ry[4][i] + ry[5][i]) + ry[6][i]);      \
    vx[i] = ((rx[0][i] + rx[1][i]) + (rx[2][i] + rx[3][i])) + ((rx[4][i] + rx[5][i]) + rx[6][i]);      \
  }

  // ---- prologue: fill the ring with virtual rows vbase .. vbase+6
#pragma unroll
  for (int s = 0; s < 7; ++s) {
    if (top && s < 3) {  // rows -3..-1: mirrored copies, filled below
#pragma unroll
      for (int i = 0; i < 2; ++i) ry[s][i] = rx[s][i] = (v2f){0.f, 0.f};
      continue;
    }
    SMO_FILL(s, vbase + s);
  }
  if (top) {  // virtual rows -3,-2,-1 = rows 2,1,0
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      ry[0][i] = ry[5][i]; rx[0][i] = rx[5][i];
      ry[1][i] = ry[4][i]; rx[1][i] = rx[4][i];
      ry[2][i] = ry[3][i]; rx[2][i] = rx[3][i];
    }
  }
  SMO_RESUM();

  const bool lane_own = lane >= 1 && lane <= 62 && xl >= own_lo && xl < own_hi;
  const float thr2_lane = lane_own ? thr2 : -1.0f;   // no squared magnitude is below -1: halo lanes never count
  auto emit = [&](int oy, const uint2& rawrow) {
    if (MODE == SMO_MODE_IMAGE) {
      if (lane_own) {
        const v2f m0 = pk_fma(vx[0], vx[0], vy[0] * vy[0]), m1 = pk_fma(vx[1], vx[1], vy[1] * vy[1]);
        float4 o;
        o.x = sqrtf(m0.x) * inv_area;
        o.y = sqrtf(m0.y) * inv_area;
        o.z = sqrtf(m1.x) * inv_area;
        o.w = sqrtf(m1.y) * inv_area;
        *reinterpret_cast<float4*>(smo_out + ((size_t)plane * Y + oy) * X + xl) = o;
      }
    } else {
      const v2f m0 = pk_fma(vx[0], vx[0], vy[0] * vy[0]), m1 = pk_fma(vx[1], vx[1], vy[1] * vy[1]);
      const float mag2[4] = {m0.x, m0.y, m1.x, m1.y};
      bool bgp[4];
      bool any = false;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        bgp[i] = mag2[i] < thr2_lane;
        any |= bgp[i];
      }
      if (any) {  // ~5 % of the background pixels pass: most lanes skip the whole block
        uint2 w = RAWRING ? rawrow : *reinterpret_cast<const uint2*>(po + (size_t)oy * X + xl);
        unsigned px[4] = {w.x & 0xffffu, w.x >> 16, w.y & 0xffffu, w.y >> 16};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (bgp[i]) {
            unsigned val = px[i];
            unsigned rel = val - win_base;
            if (rel < (unsigned)SMO_WIN) {
              atomicAdd(&lds_fine[wave][rel], 1u);
            } else {
              atomicAdd(hfine + val, 1u);
              atomicAdd(hcoarse + (val >> 8), 1u);
            }
          }
        }
      }
    }
  };

  emit(y0, rawr[3]);
  // ---- march: output row oy needs virtual row oy+3, which goes into ring slot ph.
  // SMO_STEP(ph, FILL): replace slot ph, update the running sums, emit.
#define SMO_STEP(PH, FILLER)                                                                           \
  do {                                                                                               \
    v2f oy_[2], ox_[2];                                                                              \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                  \
      oy_[i] = ry[PH][i];                                                                            \
      ox_[i] = rx[PH][i];                                                                            \
    }                                                                                                \
    FILLER(PH, oy + 3);                                                                              \
    if ((PH) == 0) {                                                                                 \
      SMO_RESUM();                                                                                   \
    } else {                                                                                         \
      _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                \
        vy[i] += ry[PH][i] - oy_[i];                                                                 \
        vx[i] += rx[PH][i] - ox_[i];                                                                 \
      }                                                                                              \
    }                                                                                                \
    emit(oy, rawr[((PH) + 4) % 7]);                                                                  \
    ++oy;                                                                                            \
  } while (0)
  // interior rows: no per-row conditions, so the 7x unrolled body is straight-line code
#define SMO_FILL_FAST(PH, VROW)                                                                       \
  do {                                                                                               \
    f4 hy_, hx_;                                                                                     \
    smo_row(A, B, Cn, false, false, ln, hy_, hx_);                                                   \
    A = B;                                                                                           \
    B = Cn;                                                                                          \
    Cn = to_f4(Dn);                                                                                  \
    if (RAWRING) rawr[((PH) + 2) % 7] = as_u2(Dn);                        \
    Dn = En;                                                                                         \
    En = load_raw4<T>(p + (size_t)((VROW) + 4) * X, xl_ld, true);                                      \
    ry[PH][0] = hy_.lo; ry[PH][1] = hy_.hi;                                                          \
    rx[PH][0] = hx_.lo; rx[PH][1] = hx_.hi;                                                          \
  } while (0)

  int oy = y0 + 1;
  // fast groups need rows oy+3 .. oy+9 strictly inside the plane and the prefetch row oy+13 <= Y-1
  // three ring rounds per trip: the three row registers (A, B, Cn) rotate with period 3, the ring with period 7, and
  // only after 21 steps is every value back in the register it started in (no copies at the back edge)
  while (oy + 21 <= yend && oy + 27 < Y && oy + 3 > 0) {
    SMO_STEP(0, SMO_FILL_FAST);
    SMO_STEP(1, SMO_FILL_FAST);
    SMO_STEP(2, SMO_FILL_FAST);
    SMO_STEP(3, SMO_FILL_FAST);
    SMO_STEP(4, SMO_FILL_FAST);
    SMO_STEP(5, SMO_FILL_FAST);
    SMO_STEP(6, SMO_FILL_FAST);
    SMO_STEP(0, SMO_FILL_FAST);
    SMO_STEP(1, SMO_FILL_FAST);
    SMO_STEP(2, SMO_FILL_FAST);
    SMO_STEP(3, SMO_FILL_FAST);
    SMO_STEP(4, SMO_FILL_FAST);
    SMO_STEP(5, SMO_FILL_FAST);
    SMO_STEP(6, SMO_FILL_FAST);
    SMO_STEP(0, SMO_FILL_FAST);
    SMO_STEP(1, SMO_FILL_FAST);
    SMO_STEP(2, SMO_FILL_FAST);
    SMO_STEP(3, SMO_FILL_FAST);
    SMO_STEP(4, SMO_FILL_FAST);
    SMO_STEP(5, SMO_FILL_FAST);
    SMO_STEP(6, SMO_FILL_FAST);
  }
  while (oy + 7 <= yend && oy + 13 < Y && oy + 3 > 0) {
    SMO_STEP(0, SMO_FILL_FAST);
    SMO_STEP(1, SMO_FILL_FAST);
    SMO_STEP(2, SMO_FILL_FAST);
    SMO_STEP(3, SMO_FILL_FAST);
    SMO_STEP(4, SMO_FILL_FAST);
    SMO_STEP(5, SMO_FILL_FAST);
    SMO_STEP(6, SMO_FILL_FAST);
  }
  while (oy < yend) {
#pragma unroll
    for (int ph = 0; ph < 7; ++ph) {
      if (oy < yend) SMO_STEP(ph, SMO_FILL);
    }
  }
#undef SMO_STEP
#undef SMO_FILL_FAST
#undef SMO_FILL
#undef SMO_RESUM

  if (MODE == SMO_MODE_HIST) {  // flush this wave's private histogram (+ its coarse 256-bin view)
    for (int i = lane; i < SMO_WIN; i += 64) {
      unsigned c = lds_fine[wave][i];
      if (c) {
        atomicAdd(hfine + win_base + i, c);
        atomicAdd(&lds_coarse[wave][(win_base + i) >> 8], c);
      }
    }
    for (int i = lane; i < HIST_COARSE; i += 64) {
      unsigned c = lds_coarse[wave][i];
      if (c) atomicAdd(hcoarse + i, c);
    }
  }
}

// Generic fallback (any size, any odd `size`): one thread per pixel, brute force with
// reflect indexing.  Only used when the marching kernel's shape constraints do not hold.
__device__ __forceinline__ int reflect_idx(int i, int n) {
  int p = 2 * n;
  i %= p;
  if (i < 0) i += p;
  return i >= n ? p - 1 - i : i;
}

template <typename T>
__device__ __forceinline__ void unit_grad(const T* __restrict__ p, int y, int x, int Y, int X, float& ny,
                                          float& nx) {
  float c = (float)p[(size_t)y * X + x];
  float gy, gx;
  if (Y == 1) gy = 0.f;
  else if (y == 0) gy = (float)p[(size_t)X + x] - c;
  else if (y == Y - 1) gy = c - (float)p[(size_t)(y - 1) * X + x];
  else gy = 0.5f * ((float)p[(size_t)(y + 1) * X + x] - (float)p[(size_t)(y - 1) * X + x]);
  if (X == 1) gx = 0.f;
  else if (x == 0) gx = (float)p[(size_t)y * X + 1] - c;
  else if (x == X - 1) gx = c - (float)p[(size_t)y * X + x - 1];
  else gx = 0.5f * ((float)p[(size_t)y * X + x + 1] - (float)p[(size_t)y * X + x - 1]);
  float n2 = gy * gy + gx * gx;
  float inv = n2 > 0.f ? rsqrtf(n2) : 0.f;
  ny = gy * inv;
  nx = gx * inv;
}

template <typename T, int MODE>
__global__ __launch_bounds__(256) void smo_generic_kernel(const T* __restrict__ src,
                                                          const uint16_t* __restrict__ orig,
                                                          float* __restrict__ smo_out,
                                                          unsigned* __restrict__ hist, int Y, int X, int size,
                                                          float thr2, float inv_area) {
  long long plane = blockIdx.y;
  long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long long)Y * X) return;
  int y = (int)(idx / X), x = (int)(idx - (long long)y * X);
  const T* p = src + (size_t)plane * Y * X;
  // scipy uniform_filter window for odd and even sizes: offsets -(size/2) .. size-1-size/2
  int lo = -(size / 2), hi = size - 1 - size / 2;
  float sy = 0.f, sx = 0.f;
  for (int dy = lo; dy <= hi; ++dy) {
    int yy = reflect_idx(y + dy, Y);
    float rsy = 0.f, rsx = 0.f;
    for (int dx = lo; dx <= hi; ++dx) {
      int xx = reflect_idx(x + dx, X);
      float ny, nx;
      unit_grad<T>(p, yy, xx, Y, X, ny, nx);
      rsy += ny;
      rsx += nx;
    }
    sy += rsy;
    sx += rsx;
  }
  float s2 = sy * sy + sx * sx;
  if (MODE == SMO_MODE_IMAGE) {
    smo_out[(size_t)plane * Y * X + idx] = sqrtf(s2) * inv_area;
  } else if (s2 < thr2) {
    unsigned val = orig[(size_t)plane * Y * X + idx];
    unsigned* h = hist + (size_t)plane * HIST_WORDS;
    atomicAdd(h + val, 1u);
    atomicAdd(h + HIST_FINE + (val >> 8), 1u);
  }
}

// ---- median from the two-level histogram: one workgroup per plane -------------------------
__device__ __forceinline__ unsigned block_find_rank(unsigned cnt, unsigned long long rank, unsigned* sh,
                                                    unsigned long long& before) {
  // returns the index i (0..255) of the bin with prefix(i) <= rank < prefix(i)+cnt(i); `before` = prefix(i)
  const int tid = threadIdx.x;
  sh[tid] = cnt;
  __syncthreads();
  for (int off = 1; off < 256; off <<= 1) {  // Hillis-Steele inclusive scan
    unsigned v = tid >= off ? sh[tid - off] : 0u;
    __syncthreads();
    sh[tid] += v;
    __syncthreads();
  }
  unsigned incl = sh[tid];
  unsigned excl = incl - cnt;
  __shared__ unsigned found_i, found_before;
  if (cnt && (unsigned long long)excl <= rank && rank < (unsigned long long)incl) {
    found_i = tid;
    found_before = excl;
  }
  __syncthreads();
  unsigned r = found_i;
  before = found_before;
  __syncthreads();
  return r;
}

// Leaves the plane's histogram all zero again: every fine bin that was counted into lies under a non-zero coarse
// bin (both the window flush and the direct adds keep the coarse view), so the few 256-bin segments under those are
// cleared and the next batch of planes needs no 257 KiB memset per plane (0.15 ms of a C2 step).
__global__ __launch_bounds__(256) void median_kernel(unsigned* __restrict__ hist, int* __restrict__ med2,
                                                     int* __restrict__ nbg, int* __restrict__ med2_user,
                                                     int* __restrict__ nbg_user) {
  __shared__ unsigned sh[256];
  __shared__ unsigned total_s;
  __shared__ unsigned used[256];
  const int tid = threadIdx.x;
  unsigned* fine = hist + (size_t)blockIdx.x * HIST_WORDS;
  unsigned* coarse = fine + HIST_FINE;
  unsigned c = coarse[tid];
  // total
  unsigned t = c;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) t += __shfl_down(t, o, 64);
  if (tid == 0) total_s = 0;
  __syncthreads();
  if ((tid & 63) == 0) atomicAdd(&total_s, t);
  __syncthreads();
  const unsigned n = total_s;
  int result = 0;
  if (n > 0) {
    unsigned long long k[2] = {(unsigned long long)(n - 1) / 2, (unsigned long long)n / 2};
    for (int q = 0; q < 2; ++q) {
      unsigned long long before;
      unsigned cb = block_find_rank(c, k[q], sh, before);
      unsigned f = fine[cb * 256 + tid];
      unsigned long long before2;
      unsigned fb = block_find_rank(f, k[q] - before, sh, before2);
      result += (int)(cb * 256 + fb);
    }
  }
  if (tid == 0) {
    med2[blockIdx.x] = result;
    nbg[blockIdx.x] = (int)n;
    if (med2_user) med2_user[blockIdx.x] = result;
    if (nbg_user) nbg_user[blockIdx.x] = (int)n;
  }
  used[tid] = c;
  __syncthreads();   // (also: every read of the histogram above is done)
  for (int b = 0; b < HIST_COARSE; ++b)
    if (used[b]) fine[b * 256 + tid] = 0u;
  coarse[tid] = 0u;
}

// out = max(2 v - sub2[plane], 0) >> 1  : trunc(max(v - sub2/2, 0)), exact integer arithmetic
__global__ __launch_bounds__(256) void sub_half_kernel(const uint16_t* __restrict__ in,
                                                       uint16_t* __restrict__ out, long long plane_px,
                                                       const int* __restrict__ sub2) {
  const long long plane = blockIdx.y;
  const int s2 = sub2[plane];
  const uint16_t* pi = in + plane * plane_px;
  uint16_t* po = out + plane * plane_px;
  const long long nv = plane_px >> 3;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nv;
       i += (long long)gridDim.x * blockDim.x) {
    u16x8 v = load8_nt(pi + i * 8);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      int d = 2 * (int)v.v[k] - s2;
      v.v[k] = (uint16_t)(d > 0 ? (d >> 1) : 0);
    }
    store8_nt(po + i * 8, v);
  }
  if (blockIdx.x == 0) {  // tail of planes whose pixel count is not a multiple of 8
    long long i = nv * 8 + threadIdx.x;
    if (i < plane_px) {
      int d = 2 * (int)pi[i] - s2;
      po[i] = (uint16_t)(d > 0 ? (d >> 1) : 0);
    }
  }
}

// ---- manual background: integer ROI sum per plane, bg = rint(sum / n) ----------------------
__global__ __launch_bounds__(256) void roi_sum_kernel(const uint16_t* __restrict__ in, long long plane_px, int X,
                                                      int r0, int c0, int r1, int c1,
                                                      unsigned long long* __restrict__ sums) {
  const long long plane = blockIdx.y;
  const uint16_t* p = in + plane * plane_px;
  const int w = c1 - c0 + 1;
  const long long n = (long long)(r1 - r0 + 1) * w;
  unsigned long long s = 0;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x) {
    int r = (int)(i / w), c = (int)(i - (long long)r * w);
    s += p[(size_t)(r0 + r) * X + c0 + c];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
  if ((threadIdx.x & 63) == 0 && s) atomicAdd(sums + plane, s);
}

__global__ void roi_bg_kernel(const unsigned long long* __restrict__ sums, long long n_roi, long long planes,
                              int* __restrict__ bg2, int* __restrict__ bg_user) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= planes) return;
  int bg = (int)rint((double)sums[i] / (double)n_roi);  // np.rint: round-half-even
  bg2[i] = 2 * bg;
  if (bg_user) bg_user[i] = bg;
}

// ---- scipy-style gaussian pre-smoothing for SMO(sigma > 0): plain, not on the default path ----
template <typename T>
__global__ __launch_bounds__(256) void gauss_reflect_kernel(const T* __restrict__ in, float* __restrict__ out,
                                                            int Y, int X, int axis, int radius, float sigma) {
  long long plane = blockIdx.y;
  long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long long)Y * X) return;
  int y = (int)(idx / X), x = (int)(idx - (long long)y * X);
  const T* p = in + (size_t)plane * Y * X;
  // Integer planes are filtered relative to their first pixel (exact differences): only gradients of the result are
  // used, a constant does not reach them, and the float32 rounding then scales with the plane's contrast instead of
  // its level.
  const float base = sizeof(T) == 2 ? (float)p[0] : 0.f;
  float acc = 0.f, wsum = 0.f;
  for (int k = -radius; k <= radius; ++k) {
    float w = __expf(-0.5f * (float)(k * k) / (sigma * sigma));
    int yy = axis == 0 ? reflect_idx(y + k, Y) : y;
    int xx = axis == 1 ? reflect_idx(x + k, X) : x;
    acc += w * ((float)p[(size_t)yy * X + xx] - base);
    wsum += w;
  }
  out[(size_t)plane * Y * X + idx] = acc / wsum;
}

// The same filter as one LDS-staged separable tile (radius <= 31, i.e. sigma <= 7.8; the plain kernel above stays for
// wider ones): a 64 x 32 tile with its halo is staged once (values relative to the plane's first pixel), filtered
// along y and then along x with precomputed, normalised taps -- no exponential per tap, no float plane between the
// passes.
struct ReflectTaps {
  float w[32];   // w[k] = exp(-k^2 / 2 sigma^2) / sum, k = 0 .. radius
};
template <typename T>
__global__ __launch_bounds__(256) void gauss_reflect_tile_kernel(const T* __restrict__ in, float* __restrict__ out,
                                                                 int Y, int X, int radius, ReflectTaps taps) {
  constexpr int TX = 64, TY = 32;
  extern __shared__ float lds[];
  const int tid = threadIdx.x;
  const int wx = TX + 2 * radius, hy = TY + 2 * radius;
  float* tile = lds;                 // hy x wx
  float* colp = lds + hy * wx;       // TY x wx after the y pass
  const long long plane = blockIdx.z;
  const int ox = blockIdx.x * TX, oy = blockIdx.y * TY;
  const T* p = in + (size_t)plane * Y * X;
  const float base = sizeof(T) == 2 ? (float)p[0] : 0.f;
  for (int i = tid; i < hy * wx; i += 256) {
    const int r = i / wx, c = i - r * wx;
    const int yy = reflect_idx(oy + r - radius, Y), xx = reflect_idx(ox + c - radius, X);
    tile[i] = (float)p[(size_t)yy * X + xx] - base;
  }
  __syncthreads();
  for (int i = tid; i < TY * wx; i += 256) {   // y pass (scipy filters axis 0 first)
    const int r = i / wx, c = i - r * wx;
    const float* t = tile + (r + radius) * wx + c;
    float s_ = taps.w[0] * t[0];
    for (int k = 1; k <= radius; ++k) s_ += taps.w[k] * (t[-k * wx] + t[k * wx]);
    colp[i] = s_;
  }
  __syncthreads();
  for (int i = tid; i < TY * TX; i += 256) {
    const int r = i / TX, c = i - r * TX;
    const int yy = oy + r, xx = ox + c;
    if (yy < Y && xx < X) {
      const float* t = colp + r * wx + c + radius;
      float s_ = taps.w[0] * t[0];
      for (int k = 1; k <= radius; ++k) s_ += taps.w[k] * (t[-k] + t[k]);
      out[(size_t)plane * Y * X + (size_t)yy * X + xx] = s_;
    }
  }
}

// ---- cv2.GaussianBlur semantics: LDS-staged separable tile, fp32, reflect-101, RNE+saturate ----
constexpr int GB_TX = 64, GB_TY = 32, GB_MAXR = 31;


struct GaussTaps {
  float x[GB_MAXR + 1], y[GB_MAXR + 1];
};

__global__ __launch_bounds__(256) void gauss_blur_kernel(const uint16_t* __restrict__ in,
                                                         uint16_t* __restrict__ out, int Y, int X, int rx,
                                                         int ry, GaussTaps taps) {
  extern __shared__ float lds[];
  const int tid = threadIdx.x;
  const int wx = GB_TX + 2 * rx, hy = GB_TY + 2 * ry;
  float* tile = lds;                 // hy x wx   input as float
  float* rowp = lds + hy * wx;       // hy x GB_TX after the x pass
  float* tx = rowp + hy * GB_TX;     // taps
  float* ty = tx + (rx + 1);
  const long long plane = blockIdx.z;
  const int ox = blockIdx.x * GB_TX, oy = blockIdx.y * GB_TY;
  const uint16_t* p = in + (size_t)plane * Y * X;
  if (tid == 0) {
    for (int i = 0; i <= rx; ++i) tx[i] = taps.x[i];
    for (int i = 0; i <= ry; ++i) ty[i] = taps.y[i];
  }
  for (int i = tid; i < hy * wx; i += 256) {
    int r = i / wx, c = i - r * wx;
    int yy = reflect101_idx(oy + r - ry, Y), xx = reflect101_idx(ox + c - rx, X);
    tile[i] = (float)p[(size_t)yy * X + xx];
  }
  __syncthreads();
  for (int i = tid; i < hy * GB_TX; i += 256) {
    int r = i / GB_TX, c = i - r * GB_TX;
    const float* t = tile + r * wx + c + rx;
    float s = __fmul_rn(tx[0], t[0]);
    for (int k = 1; k <= rx; ++k) s = __fadd_rn(s, __fmul_rn(tx[k], __fadd_rn(t[-k], t[k])));
    rowp[i] = s;
  }
  __syncthreads();
  for (int i = tid; i < GB_TY * GB_TX; i += 256) {
    int r = i / GB_TX, c = i - r * GB_TX;
    int yy = oy + r, xx = ox + c;
    if (yy < Y && xx < X) {
      const float* t = rowp + (r + ry) * GB_TX + c;
      float s = __fmul_rn(ty[0], t[0]);
      for (int k = 1; k <= ry; ++k)
        s = __fadd_rn(s, __fmul_rn(ty[k], __fadd_rn(t[-k * GB_TX], t[k * GB_TX])));
      float r_ = rintf(s);
      r_ = fminf(fmaxf(r_, 0.f), 65535.f);
      out[(size_t)plane * Y * X + (size_t)yy * X + xx] = (uint16_t)r_;
    }
  }
}

// The same arithmetic for the common square kernels (radius R <= 7: the reference's default is 15 x 15), marching:
// a workgroup owns a strip of 128 columns and walks down it 32 rows a step.  The x-filtered rows live in an LDS
// ring of 32 + 2R rows, so every source row is fetched and x-filtered once per strip (the tile kernel above
// re-reads and re-filters its 2R halo rows for every 32-row tile: 44 % more at R = 7, and stages pixel by pixel).
// A step's source rows arrive as unaligned 16-byte loads (8 pixels), requested a step ahead and kept in registers;
// the x pass reads 8 + 2R neighbouring samples as three aligned 16-byte LDS vectors and produces 8 outputs from
// registers, the y pass reads 8 + 2R ring rows of 2 columns and produces 2 x 8 outputs, stored 4 bytes per lane.  Every product and sum is rounded on its own (no FMA), in OpenCV's order: the bits of the
// tile kernel.
constexpr int GM_TX = 128, GM_TY = 32;
template <int R> struct GmGeom {
  static constexpr int SW = (GM_TX + 2 * R + 7) / 8 * 8;   // staged columns, from image column ox - R on
  static constexpr int VPR = SW / 8;                        // 16-byte vectors per staged row
  static constexpr int RH = GM_TY + 2 * R;                  // ring rows
  static constexpr int NV = (GM_TY * VPR + 255) / 256;      // vectors a thread stages per step
  static constexpr int XV = (8 + 2 * R + 7) / 8;            // vectors an x-pass item reads
};
typedef unsigned gm_u32x4 __attribute__((ext_vector_type(4)));
typedef gm_u32x4 gm_u32x4_px __attribute__((aligned(2)));   // 16-byte load at any pixel address

template <int R>
__global__ __launch_bounds__(256, 4) void gauss_march_kernel(const uint16_t* __restrict__ in, uint16_t* __restrict__ out,
                                                          int Y, int X, int band, GaussTaps taps) {
  using G = GmGeom<R>;
  __shared__ gm_u32x4 raw[GM_TY * G::VPR];                       // the step's source rows, u16
  __shared__ __align__(16) float ring[G::RH * GM_TX];             // x-filtered rows, row = source row mod RH (from the band's first)
  const int tid = threadIdx.x;
  const long long plane = blockIdx.z;
  const uint16_t* __restrict__ p = in + (size_t)plane * Y * X;
  uint16_t* __restrict__ po = out + (size_t)plane * Y * X;
  // strips next to each other (consecutive ids) run on one XCD: the lines they share come from one L2
  const unsigned lin = xcd_remap(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
  const int ox = (lin % gridDim.x) * GM_TX;
  const int y0 = (lin / gridDim.x) * band, y1 = min(Y, y0 + band);
  const int first = ox - R;                                       // image column of staged column 0
  const bool cols_inside = first >= 0 && first + G::SW <= X;      // (uniform) no mirrored column in this strip
  float tx[R + 1], ty[R + 1];
#pragma unroll
  for (int k = 0; k <= R; ++k) { tx[k] = taps.x[k]; ty[k] = taps.y[k]; }

  // staged vector i of a step: row i / VPR, vector i % VPR; source row `src0 + row` (mirrored outside the plane)
  auto fetch = [&](int i, int src0) -> gm_u32x4 {
    const int r = i / G::VPR, v = i - r * G::VPR;
    const int yy = src0 + r;
    const size_t ro = (size_t)((unsigned)yy < (unsigned)Y ? yy : reflect101_idx(yy, Y)) * X;
    const int c0 = first + 8 * v;
    if (cols_inside) return *reinterpret_cast<const gm_u32x4_px*>(p + ro + c0);
    unsigned e[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) e[k] = p[ro + reflect101_idx(c0 + k, X)];
    return (gm_u32x4){e[0] | (e[1] << 16), e[2] | (e[3] << 16), e[4] | (e[5] << 16), e[6] | (e[7] << 16)};
  };
  // x pass over `nrows` staged rows; staged row r is source row src0 + r and goes to ring row (src0 + r - (y0 - R)) mod RH.
  // An item is 8 outputs of TWO rows (r and r + nrows / 2) worked as packed pairs: v_pk_mul / v_pk_add round each
  // product and sum like their scalar forms and cost 1.4x one scalar instruction for two.
  auto xpass = [&](int nrows, int src0) {
    const int half = nrows / 2;
    for (int i = tid; i < half * (GM_TX / 8); i += 256) {
      const int r = i / (GM_TX / 8), g = i - r * (GM_TX / 8);
      const gm_u32x4* ta = raw + r * G::VPR + g;
      const gm_u32x4* tb = ta + half * G::VPR;
      v2f smp[8 * G::XV];
#pragma unroll
      for (int v = 0; v < G::XV; ++v) {
        const gm_u32x4 qa = ta[v], qb = tb[v];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          smp[8 * v + 2 * k] = (v2f){(float)(qa[k] & 0xffffu), (float)(qb[k] & 0xffffu)};
          smp[8 * v + 2 * k + 1] = (v2f){(float)(qa[k] >> 16), (float)(qb[k] >> 16)};
        }
      }
      v2f o[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        v2f s_ = (v2f){tx[0], tx[0]} * smp[c + R];
#pragma unroll
        for (int k = 1; k <= R; ++k) s_ = s_ + (v2f){tx[k], tx[k]} * (smp[c + R - k] + smp[c + R + k]);
        o[c] = s_;
      }
      const int ra = (src0 + r - (y0 - R)) % G::RH, rb = (src0 + r + half - (y0 - R)) % G::RH;
      float4* da = reinterpret_cast<float4*>(ring + ra * GM_TX + 8 * g);
      float4* db = reinterpret_cast<float4*>(ring + rb * GM_TX + 8 * g);
      da[0] = make_float4(o[0].x, o[1].x, o[2].x, o[3].x);
      da[1] = make_float4(o[4].x, o[5].x, o[6].x, o[7].x);
      db[0] = make_float4(o[0].y, o[1].y, o[2].y, o[3].y);
      db[1] = make_float4(o[4].y, o[5].y, o[6].y, o[7].y);
    }
  };

  // lead-in: the 2R rows above the first step's new rows
  gm_u32x4 q[G::NV];
  {
    if (tid < 2 * R * G::VPR) raw[tid] = fetch(tid, y0 - R);
    static_assert(2 * R * G::VPR <= 256, "the lead-in is one round of staging");
#pragma unroll
    for (int jv = 0; jv < G::NV; ++jv) {
      const int i = tid + jv * 256;
      if (i < GM_TY * G::VPR) q[jv] = fetch(i, y0 + R);
    }
    __syncthreads();
    xpass(2 * R, y0 - R);
    __syncthreads();
  }
  for (int oy = y0; oy < y1; oy += GM_TY) {
    // the step's new source rows oy + R .. oy + R + 31 (already in registers) -> LDS
#pragma unroll
    for (int jv = 0; jv < G::NV; ++jv) {
      const int i = tid + jv * 256;
      if (i < GM_TY * G::VPR) raw[i] = q[jv];
    }
    __syncthreads();
    if (oy + GM_TY < y1) {
#pragma unroll
      for (int jv = 0; jv < G::NV; ++jv) {
        const int i = tid + jv * 256;
        if (i < GM_TY * G::VPR) q[jv] = fetch(i, oy + GM_TY + R);
      }
    }
    xpass(GM_TY, oy + R);
    __syncthreads();
    // y pass: 2 columns x 8 rows per thread (a wave = 128 columns of one 8-row group: 256-byte stores)
    {
      const int cp = tid & 63, rg = tid >> 6;
      const int ya = oy + 8 * rg;                       // first output row of this thread
      int rr = (ya - R - (y0 - R)) % G::RH;             // ring row of source row ya - R
      v2f col[8 + 2 * R];
#pragma unroll
      for (int i = 0; i < 8 + 2 * R; ++i) {
        col[i] = *reinterpret_cast<const v2f*>(ring + rr * GM_TX + 2 * cp);
        rr = rr + 1 == G::RH ? 0 : rr + 1;
      }
      const int xx = ox + 2 * cp;
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        v2f s_ = (v2f){ty[0], ty[0]} * col[r + R];
#pragma unroll
        for (int k = 1; k <= R; ++k) s_ = s_ + (v2f){ty[k], ty[k]} * (col[r + R - k] + col[r + R + k]);
        const float res[2] = {fminf(fmaxf(rintf(s_.x), 0.f), 65535.f), fminf(fmaxf(rintf(s_.y), 0.f), 65535.f)};
        const int yy = ya + r;
        if (yy < y1 && xx < X)   // X is even: both columns of the pair are inside
          __builtin_nontemporal_store((unsigned)res[0] | ((unsigned)res[1] << 16), reinterpret_cast<unsigned*>(po + (size_t)yy * X + xx));
      }
    }
    __syncthreads();   // the ring rows this step read may be overwritten by the next
  }
}

static bool smo_fast_ok(int Y, int X, int size) { return size == 7 && X % 4 == 0 && X >= 248 && Y >= 8; }

static SmoGeom smo_geom(int Y, int X, int band) {
  SmoGeom g;
  g.Y = Y;
  g.X = X;
  g.band = band;
  g.nstrips = (X + SMO_COLS - 1) / SMO_COLS;
  g.nbands = (Y + band - 1) / band;
  int t = g.nstrips * g.nbands;
  g.tasks_per_plane = (t + SMO_WAVES - 1) / SMO_WAVES * SMO_WAVES;
  return g;
}

static int pick_band(int Y, int64_t n) {
  // every band re-primes its 7-row ring (6 rows: 4.7 % at 128 rows, 9 % at 64); ~3072 waves are resident
  // (3 per SIMD), so bands shrink only while fewer than ~4 rounds of waves would be in flight.
  // Measured on C2 (128 planes a launch): 64 rows 14.85, 128 rows 14.26, 256 rows 14.26, 512 rows 15.4 ms/step
  // Launches over a whole stack (levels-only calls: 2000 planes on C2) have rounds to spare and take taller bands:
  // 128 rows 8.36, 256 rows 8.04, 512 rows 7.99 ms/step.
  int band = 128;
  while (band > 16 && n * ((Y + band - 1) / band) * 9 < 12000) band >>= 1;
  while (band < 512 && n * ((Y + 2 * band - 1) / (2 * band)) * 9 >= 60000) band <<= 1;
  return band;
}

template <typename T, int MODE>
static int launch_smo(const T* src, const uint16_t* orig, float* smo_out, unsigned* hist, int64_t n, int Y,
                      int X, int size, float threshold, hipStream_t st) {
  double tscaled = (double)threshold * size * size;
  float thr2 = (float)(tscaled * tscaled);
  if (threshold <= 0) thr2 = 0.f;
  float inv_area = 1.0f / (float)(size * size);
  if (smo_fast_ok(Y, X, size)) {
    SmoGeom g = smo_geom(Y, X, pick_band(Y, n));
    long long blocks = (long long)n * g.tasks_per_plane / SMO_WAVES;
    { ProfScope ps_(K_SMO_MARCH, st); hipLaunchKernelGGL((smo_march_kernel<T, MODE>), dim3((unsigned)blocks), dim3(SMO_WAVES * 64), 0, st, src,
                       orig, smo_out, hist, g, thr2, inv_area); }
  } else {
    unsigned bx = (unsigned)(((long long)Y * X + 255) / 256);
    { ProfScope ps_(K_SMO_GENERIC, st); hipLaunchKernelGGL((smo_generic_kernel<T, MODE>), dim3(bx, (unsigned)n), dim3(256), 0, st, src, orig,
                       smo_out, hist, Y, X, size, thr2, inv_area); }
  }
  IAP_LAUNCH_CHECK();
  return IAP_OK;
}

template <typename T>
static int launch_gauss_reflect(const T* in, float* tmp, float* out, int64_t n, int Y, int X, float sigma,
                                hipStream_t st) {
  int radius = (int)(4.0f * sigma + 0.5f);
  if (radius <= 31) {   // one tile kernel, taps worked out here
    ReflectTaps h;
    double sum = 1.0;
    for (int k = 1; k <= radius; ++k) sum += 2.0 * exp(-0.5 * (double)(k * k) / ((double)sigma * sigma));
    for (int k = 0; k < 32; ++k) h.w[k] = k <= radius ? (float)(exp(-0.5 * (double)(k * k) / ((double)sigma * sigma)) / sum) : 0.f;
    const size_t lds = ((size_t)(32 + 2 * radius) * (64 + 2 * radius) + (size_t)32 * (64 + 2 * radius)) * 4;
    if (lds > 48 * 1024)
      IAP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gauss_reflect_tile_kernel<T>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    dim3 grid((X + 63) / 64, (Y + 31) / 32, (unsigned)n);
    { ProfScope ps_(K_GAUSS, st); hipLaunchKernelGGL((gauss_reflect_tile_kernel<T>), grid, dim3(256), lds, st, in, out, Y, X, radius, h); }
    IAP_LAUNCH_CHECK();
    return IAP_OK;
  }
  unsigned bx = (unsigned)(((long long)Y * X + 255) / 256);
  { ProfScope ps_(K_GAUSS, st); hipLaunchKernelGGL((gauss_reflect_kernel<T>), dim3(bx, (unsigned)n), dim3(256), 0, st, in, tmp, Y, X, 0,
                     radius, sigma); }
  { ProfScope ps_(K_GAUSS, st); hipLaunchKernelGGL((gauss_reflect_kernel<float>), dim3(bx, (unsigned)n), dim3(256), 0, st,
                     (const float*)tmp, out, Y, X, 1, radius, sigma); }
  IAP_LAUNCH_CHECK();
  return IAP_OK;
}

// Scratch of iap_bg_const_static_u16 (the entry point that allocates nothing): the ROI sums (u64) and doubled
// levels (int) of up to IAP_BG_CONST_STATIC_MAX_PLANES planes, 16 bytes a plane.  Every GPU the library runs on
// holds its own copy until the process ends.  Host code never passes the array's name to a kernel or a memset: it
// is a host-side stand-in there; the device address comes from hipGetSymbolAddress, once per device.
static __device__ __align__(256) unsigned char scratch[IAP_BG_CONST_STATIC_MAX_PLANES * 16];

constexpr int64_t BG_BATCH_BYTES = 2048ll << 20;  // planes per batch: long launches hide the march tail (C2 bg stage, levels only: 128 MB 18.2, 1 GB 8.95, 2 GB 8.5 ms)

}  // namespace iap

using namespace iap;

static int check_plane_args(const char* fn, const void* in, const void* out, int64_t n, int Y, int X) {
  IAP_REQUIRE(in && out, IAP_E_ARG, "%s: null pointer", fn);
  IAP_REQUIRE(n > 0 && Y > 0 && X > 0, IAP_E_ARG, "%s: non-positive size (n=%lld Y=%d X=%d)", fn, (long long)n,
              Y, X);
  IAP_REQUIRE(aligned16(in) && aligned16(out), IAP_E_ALIGN, "%s: pointers must be 16-byte aligned", fn);
  IAP_REQUIRE(n <= 0x7fffffff && (int64_t)Y * X < (1ll << 31), IAP_E_UNSUPPORTED, "%s: size too large", fn);
  return IAP_OK;
}

extern "C" size_t iap_smo_image_ws_bytes(int64_t n, int Y, int X, float sigma) {
  if (n <= 0 || Y <= 0 || X <= 0 || !(sigma > 0.f)) return 0;
  return (size_t)n * Y * X * 8 + 256;   // two float planes per plane: after the y pass, after the x pass
}

extern "C" int iap_smo_image(const void* in, int in_kind, float* out, int64_t n, int Y, int X, float sigma,
                             int size, void* ws, size_t ws_bytes, void* stream) {
  int rc = check_plane_args("iap_smo_image", in, out, n, Y, X);
  if (rc) return rc;
  IAP_REQUIRE(in_kind == 0 || in_kind == 1, IAP_E_ARG, "iap_smo_image: in_kind must be 0 (u16) or 1 (f32)");
  IAP_REQUIRE(size >= 1 && size <= 63, IAP_E_ARG, "iap_smo_image: size must be in 1..63");
  IAP_REQUIRE(size & 1, IAP_E_ARG, "iap_smo_image: size must be odd");
  IAP_REQUIRE(sigma >= 0.f, IAP_E_ARG, "iap_smo_image: sigma must be >= 0");
  IAP_REQUIRE(n <= 65535, IAP_E_UNSUPPORTED, "iap_smo_image: n must be <= 65535");
  hipStream_t st = as_stream(stream);
  if (sigma > 0.f) {   // scipy gaussian_filter (reflect, truncate 4) first, as SMO(sigma=...) does
    IAP_REQUIRE(ws && aligned16(ws) && ws_bytes >= iap_smo_image_ws_bytes(n, Y, X, sigma), IAP_E_WS,
                "iap_smo_image: sigma > 0 needs a 16-byte aligned workspace of %zu bytes (got %zu)",
                iap_smo_image_ws_bytes(n, Y, X, sigma), ws_bytes);
    float* ftmp = (float*)ws;
    float* fsm = ftmp + (size_t)n * Y * X;
    rc = in_kind == 0 ? launch_gauss_reflect<uint16_t>((const uint16_t*)in, ftmp, fsm, n, Y, X, sigma, st)
                      : launch_gauss_reflect<float>((const float*)in, ftmp, fsm, n, Y, X, sigma, st);
    if (rc) return rc;
    return launch_smo<float, SMO_MODE_IMAGE>(fsm, nullptr, out, nullptr, n, Y, X, size, 0.f, st);
  }
  if (in_kind == 0)
    return launch_smo<uint16_t, SMO_MODE_IMAGE>((const uint16_t*)in, nullptr, out, nullptr, n, Y, X, size, 0.f,
                                                st);
  return launch_smo<float, SMO_MODE_IMAGE>((const float*)in, nullptr, out, nullptr, n, Y, X, size, 0.f, st);
}

static int64_t bg_batch_planes(int64_t n, int Y, int X) {
  int64_t per = (int64_t)Y * X * 2;
  int64_t bytes = BG_BATCH_BYTES;
  int64_t b = bytes / per;
  if (b < 1) b = 1;
  if (b > 256) b = 256;  // bounds the histogram workspace (257 KiB per plane in flight)
  if (b > n) b = n;
  return b;
}

extern "C" size_t iap_bg_smo_ws_bytes(int64_t n, int Y, int X, float sigma) {
  if (n <= 0 || Y <= 0 || X <= 0) return 0;
  int64_t b = bg_batch_planes(n, Y, X);
  // per batch plane: histogram words + med2 + nbg; with sigma > 0 two float planes on top (none at sigma = 0)
  size_t per = (size_t)HIST_WORDS * 4 + 16 + (sigma > 0.f ? (size_t)Y * X * 8 : 0);
  return (size_t)b * per + 512;
}

extern "C" int iap_bg_smo_u16(const uint16_t* in, uint16_t* out, int64_t n, int Y, int X, float sigma, int size,
                              float threshold, int32_t* med2_out, int32_t* nbg_out, void* ws, size_t ws_bytes,
                              void* stream) {
  // out == NULL: levels only (med2_out), for callers that subtract on load (iap_im_reg_sub_u16)
  int rc = check_plane_args("iap_bg_smo_u16", in, out ? (const void*)out : (const void*)in, n, Y, X);
  if (rc) return rc;
  IAP_REQUIRE(out || med2_out, IAP_E_ARG, "iap_bg_smo_u16: without an output stack med2_out must be given");
  IAP_REQUIRE(size >= 1 && size <= 63, IAP_E_ARG, "iap_bg_smo_u16: size must be in 1..63");
  IAP_REQUIRE(size & 1, IAP_E_ARG, "iap_bg_smo_u16: size must be odd");
  IAP_REQUIRE(sigma >= 0.f, IAP_E_ARG, "iap_bg_smo_u16: sigma must be >= 0");
  IAP_REQUIRE(ws && aligned16(ws), IAP_E_ARG, "iap_bg_smo_u16: workspace missing or not 16-byte aligned");
  IAP_REQUIRE(ws_bytes >= iap_bg_smo_ws_bytes(n, Y, X, sigma), IAP_E_WS,
              "iap_bg_smo_u16: workspace of %zu bytes, need %zu", ws_bytes, iap_bg_smo_ws_bytes(n, Y, X, sigma));
  hipStream_t st = as_stream(stream);
  int64_t B = bg_batch_planes(n, Y, X);
  if (sigma == 0.f && !out) {
    // levels only and no float planes to keep: the whole workspace can hold histograms, so one launch takes as
    // many planes as fit and the march kernel's tail is paid once instead of once per batch (C2: 8.66 -> 8.3 ms;
    // with the subtraction pass behind every batch the smaller batches measured 0.2 ms better)
    const int64_t fit = (int64_t)((ws_bytes - 256) / ((size_t)HIST_WORDS * 4 + 16));
    B = std::max(B, std::min(std::min(n, fit), (int64_t)65535));
  }
  const int64_t px = (int64_t)Y * X;
  char* w = (char*)ws;
  unsigned* hist = (unsigned*)w;
  w += (size_t)B * HIST_WORDS * 4;
  int* med2 = (int*)w;
  w += (size_t)B * 8;
  int* nbg = (int*)w;
  w += (size_t)B * 8;
  w = (char*)(((uintptr_t)w + 255) & ~(uintptr_t)255);
  float* ftmp = (float*)w;
  float* fsm = ftmp + (size_t)B * px;
  // the histograms are zeroed once; the median kernel hands them back clean for the next batch
  IAP_HIP(hipMemsetAsync(hist, 0, (size_t)std::min<int64_t>(B, n) * HIST_WORDS * 4, st));
  for (int64_t p0 = 0; p0 < n; p0 += B) {
    int64_t nb = n - p0 < B ? n - p0 : B;
    const uint16_t* bin = in + p0 * px;
    uint16_t* bout = out ? out + p0 * px : nullptr;
    if (sigma > 0.f) {
      rc = launch_gauss_reflect<uint16_t>(bin, ftmp, fsm, nb, Y, X, sigma, st);
      if (rc) return rc;
      rc = launch_smo<float, SMO_MODE_HIST>(fsm, bin, nullptr, hist, nb, Y, X, size, threshold, st);
    } else {
      rc = launch_smo<uint16_t, SMO_MODE_HIST>(bin, bin, nullptr, hist, nb, Y, X, size, threshold, st);
    }
    if (rc) return rc;
    { ProfScope ps_(K_MEDIAN, st); hipLaunchKernelGGL(median_kernel, dim3((unsigned)nb), dim3(256), 0, st, hist, med2, nbg,
                       med2_out ? med2_out + p0 : nullptr, nbg_out ? nbg_out + p0 : nullptr); }
    unsigned gx = (unsigned)((px / 8 + 255) / 256);
    if (gx > 256) gx = 256;
    if (gx == 0) gx = 1;
    if (bout) { ProfScope ps_(K_SUB, st); hipLaunchKernelGGL(sub_half_kernel, dim3(gx, (unsigned)nb), dim3(256), 0, st, bin, bout, (long long)px,
                       (const int*)med2); }
    IAP_LAUNCH_CHECK();
  }
  return IAP_OK;
}

// ROI sums -> level per plane -> subtraction; `sums` (u64[n]) and `bg2` (int[n]) are device scratch
static int bg_const_run(const uint16_t* in, uint16_t* out, int64_t n, int Y, int X, int r0, int c0, int r1, int c1,
                        int32_t* bg_out, unsigned long long* sums, int* bg2, hipStream_t st) {
  IAP_HIP(hipMemsetAsync(sums, 0, (size_t)n * 8, st));
  long long n_roi = (long long)(r1 - r0 + 1) * (c1 - c0 + 1);
  unsigned gx = (unsigned)((n_roi + 255) / 256);
  if (gx > 64) gx = 64;
  { ProfScope ps_(K_ROI, st); hipLaunchKernelGGL(roi_sum_kernel, dim3(gx, (unsigned)n), dim3(256), 0, st, in, (long long)Y * X, X, r0, c0, r1,
                     c1, sums); }
  { ProfScope ps_(K_ROI, st); hipLaunchKernelGGL(roi_bg_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st,
                     (const unsigned long long*)sums, n_roi, (long long)n, bg2, bg_out); }
  long long px = (long long)Y * X;
  unsigned sx = (unsigned)((px / 8 + 255) / 256);
  if (sx > 256) sx = 256;
  if (sx == 0) sx = 1;
  { ProfScope ps_(K_SUB, st); hipLaunchKernelGGL(sub_half_kernel, dim3(sx, (unsigned)n), dim3(256), 0, st, in, out, px, (const int*)bg2); }
  IAP_LAUNCH_CHECK();
  return IAP_OK;
}

static int bg_const_check(const char* fn, const uint16_t* in, uint16_t* out, int64_t n, int Y, int X, int r0, int c0,
                          int r1, int c1) {
  int rc = check_plane_args(fn, in, out, n, Y, X);
  if (rc) return rc;
  IAP_REQUIRE(0 <= r0 && r0 <= r1 && r1 < Y && 0 <= c0 && c0 <= c1 && c1 < X, IAP_E_ARG,
              "%s: ROI (%d,%d)-(%d,%d) outside the %dx%d plane", fn, r0, c0, r1, c1, Y, X);
  return IAP_OK;
}

extern "C" int iap_bg_const_u16(const uint16_t* in, uint16_t* out, int64_t n, int Y, int X, int r0, int c0,
                                int r1, int c1, int32_t* bg_out, void* ws, size_t ws_bytes, void* stream) {
  int rc = bg_const_check("iap_bg_const_u16", in, out, n, Y, X, r0, c0, r1, c1);
  if (rc) return rc;
  IAP_REQUIRE(ws && aligned16(ws) && ws_bytes >= (size_t)n * 16, IAP_E_WS,
              "iap_bg_const_u16: workspace must hold n*16 bytes");
  IAP_REQUIRE(n <= 65535, IAP_E_UNSUPPORTED, "iap_bg_const_u16: n must be <= 65535 per call");
  unsigned long long* sums = (unsigned long long*)ws;
  return bg_const_run(in, out, n, Y, X, r0, c0, r1, c1, bg_out, sums, (int*)(sums + n), as_stream(stream));
}

// Device address of `scratch` on the current device, resolved once per device (with that device current) and kept.
static int static_scratch(unsigned char** ptr) {
  static std::mutex mu;
  static unsigned char* per_device[64] = {nullptr};
  int dev = -1;
  IAP_HIP(hipGetDevice(&dev));
  IAP_REQUIRE(dev >= 0 && dev < 64, IAP_E_UNSUPPORTED, "iap_bg_const_static_u16: device index %d out of range", dev);
  std::lock_guard<std::mutex> lk(mu);
  if (!per_device[dev]) {
    void* p = nullptr;
    IAP_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
    IAP_REQUIRE(p != nullptr, IAP_E_UNSUPPORTED, "iap_bg_const_static_u16: hipGetSymbolAddress returned a null address");
    per_device[dev] = (unsigned char*)p;
  }
  *ptr = per_device[dev];
  return IAP_OK;
}

extern "C" int iap_bg_const_static_u16(const uint16_t* in, uint16_t* out, int64_t n, int Y, int X, int r0, int c0,
                                       int r1, int c1, int32_t* bg_out, void* stream) {
  int rc = bg_const_check("iap_bg_const_static_u16", in, out, n, Y, X, r0, c0, r1, c1);
  if (rc) return rc;
  IAP_REQUIRE(n <= IAP_BG_CONST_STATIC_MAX_PLANES, IAP_E_UNSUPPORTED,
              "iap_bg_const_static_u16: %lld planes exceed the static scratch (at most %d per call; use "
              "iap_bg_const_u16 with a workspace, or split the call)", (long long)n, IAP_BG_CONST_STATIC_MAX_PLANES);
  unsigned char* ptr = nullptr;
  rc = static_scratch(&ptr);
  if (rc) return rc;
  // the array keeps the last call's contents: bg_const_run zeroes the sums it adds into; the levels are overwritten
  unsigned long long* sums = (unsigned long long*)ptr;
  return bg_const_run(in, out, n, Y, X, r0, c0, r1, c1, bg_out, sums, (int*)(sums + n), as_stream(stream));
}

extern "C" int iap_gauss_blur_u16(const uint16_t* in, uint16_t* out, int64_t n, int Y, int X, int kx, int ky,
                                  double sigma, void* stream) {
  int rc = check_plane_args("iap_gauss_blur_u16", in, out, n, Y, X);
  if (rc) return rc;
  IAP_REQUIRE(in != out, IAP_E_ARG, "iap_gauss_blur_u16: in-place operation is not supported");
  IAP_REQUIRE(kx >= 1 && ky >= 1 && (kx & 1) && (ky & 1) && kx <= 2 * GB_MAXR + 1 && ky <= 2 * GB_MAXR + 1,
              IAP_E_ARG, "iap_gauss_blur_u16: kernel sizes must be odd and <= %d", 2 * GB_MAXR + 1);
  IAP_REQUIRE(n <= 65535, IAP_E_UNSUPPORTED, "iap_gauss_blur_u16: n must be <= 65535 per call");
  hipStream_t st = as_stream(stream);
  // taps as cv2.getGaussianKernel(k, sigma, CV_32F); they ride to the kernel by value
  GaussTaps h;
  auto fill = [](int k, double sg, float* t) {
    if (!(sg > 0) && k <= 7) {   // OpenCV's fixed table for small kernels without a sigma (small_gaussian_tab)
      static const float tab[4][4] = {{1.f, 0.f, 0.f, 0.f}, {0.5f, 0.25f, 0.f, 0.f}, {0.375f, 0.25f, 0.0625f, 0.f},
                                      {0.28125f, 0.21875f, 0.109375f, 0.03125f}};
      for (int i = 0; i <= (k - 1) / 2; ++i) t[i] = tab[(k - 1) / 2][i];
      return;
    }
    // OpenCV 4.1.2 (the reference's pin), getGaussianKernel with CV_32F: float values of exp(scale2X x^2), their sum
    // taken in index order in a double, normalised by a multiplication with the sum's reciprocal
    const double s = sg > 0 ? sg : 0.3 * ((k - 1) * 0.5 - 1) + 0.8;
    const double scale2x = -0.5 / (s * s);
    float cf[2 * GB_MAXR + 1];
    double sum = 0;
    for (int i = 0; i < k; ++i) {
      const double x = i - (k - 1) * 0.5;
      cf[i] = (float)exp(scale2x * x * x);
      sum += cf[i];
    }
    sum = 1. / sum;
    const int r = (k - 1) / 2;
    for (int i = 0; i <= r; ++i) t[i] = (float)(cf[r + i] * sum);
  };
  fill(kx, sigma, h.x);
  fill(ky, sigma, h.y);
  int rx = (kx - 1) / 2, ry = (ky - 1) / 2;
  if (rx == ry && rx >= 1 && rx <= 7 && X % 2 == 0 && X >= 8) {   // marching kernel (4-byte stores need even rows)
    // bands of 256 rows (8 steps): the lead-in is 2R of them; shorter bands only while the launch would not fill the chip
    int band = 256;
    while (band > GM_TY && (long long)((X + GM_TX - 1) / GM_TX) * ((Y + band - 1) / band) * n < 2048) band >>= 1;
    dim3 grid((X + GM_TX - 1) / GM_TX, (Y + band - 1) / band, (unsigned)n);
    ProfScope ps_(K_GAUSS, st);
    switch (rx) {
#define IAP_GM(R_) case R_: hipLaunchKernelGGL(gauss_march_kernel<R_>, grid, dim3(256), 0, st, in, out, Y, X, band, h); break;
      IAP_GM(1) IAP_GM(2) IAP_GM(3) IAP_GM(4) IAP_GM(5) IAP_GM(6) IAP_GM(7)
#undef IAP_GM
    }
    IAP_LAUNCH_CHECK();
    return IAP_OK;
  }
  size_t lds = ((size_t)(GB_TY + 2 * ry) * (GB_TX + 2 * rx) + (size_t)(GB_TY + 2 * ry) * GB_TX + rx + ry + 2) * 4;
  dim3 grid((X + GB_TX - 1) / GB_TX, (Y + GB_TY - 1) / GB_TY, (unsigned)n);
  if (lds > 48 * 1024)   // kernels of 57 and up pass the default 64 KB of dynamic LDS (71 KB at 63): ask for it
    IAP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gauss_blur_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  { ProfScope ps_(K_GAUSS, st); hipLaunchKernelGGL(gauss_blur_kernel, grid, dim3(256), lds, st, in, out, Y, X, rx, ry, h); }
  IAP_LAUNCH_CHECK();
  return IAP_OK;
}
