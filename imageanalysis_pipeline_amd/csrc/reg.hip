// Registration: FFT cross-correlation shift estimate + fused apply-shift
// (reference: pipeline/utility.py:274-430 im_reg, which calls pystackreg's TurboReg; the
// estimator here is the FFT cross-power scheme north_star asks for).
//
// Estimate, per moving plane (all on device, no host round trip):
//   K1 rows_r2c      u16 rows -> half spectrum, two real rows per complex FFT (LDS Stockham); stored as
//                    packed half pairs (the reference plane's: float)
//   K2 cols_xpower   column FFT, * conj-multiply with the reference spectrum (DC zeroed,
//                    optional phase normalisation), inverse column FFT in the same LDS tile,
//                    written as float Q[y][v], spatial along y, spectral along x; the reference's tile stays
//                    in registers while a workgroup walks its frames, the next frame's tile is in flight
//   K3 rows_c2r_argmax  inverse row FFT + wave-shuffle / LDS arg-max (correlation surface is
//                    never written to HBM)
//   K4 coarse peak      arg-max over the per-workgroup partials -> signed integer shift: done by K5's table
//                    kernel (whole-pixel mode: coarse_finish, a launch of its own)
//   K5 upsample_tables / upsample_partial  matrix-DFT of the cross-power spectrum on a 15x15 grid of
//                    0.1 px offsets, evaluated from Q (the buffer K3 reads anyway, so the spectrum
//                    itself is never written): periodic-sinc weights along y, +-m symmetric phases
//                    along x, per-chunk partials (the Nyquist column rides in the table kernel)
//   K6 upsample_finish  sum partials, arg-max, parabolic vertex -> (dy, dx) as f64
// Apply: apply_shift_kernel -- separable cubic B-spline shift = scipy.ndimage.shift(order=3):
// mirror-boundary prefilter as two one-multiply recursions started from 12 explicit taps
// (truncation 0.268^12), then the 4-tap evaluation, as packed f32 pairs; a workgroup marches down a
// 64-column strip and keeps the x-filtered rows two steps share in LDS; zero outside, clip,
// truncate to u16.  An axis with an integer shift bypasses the filter: exact copies.
#include <math.h>
#include <stdlib.h>
#include <algorithm>
#include <atomic>
#include "common.h"

namespace iap {

// complex f32 as a packed pair: the FFT butterflies run on v_pk_add/mul/fma_f32, which issue at the
// rate of their scalar forms on this part (half the VALU instructions of float2 code)
typedef float cf __attribute__((ext_vector_type(2)));
__device__ __forceinline__ cf mkc(float x, float y) { return (cf){x, y}; }

// streaming 8-byte load of one complex value (spectra that are read exactly once by a kernel)
__device__ __forceinline__ cf load_cf_nt(const cf* p) { return __builtin_nontemporal_load(p); }

// a * b: (a.x b.x, a.x b.y) then + (a.y (-b.y), a.y b.x); the half selects and the sign ride in the
// second instruction (the compiler materialises such a swizzle with two extra moves)
__device__ __forceinline__ cf cmul(cf a, cf b) {
  const cf t = (cf){a.x, a.x} * b;
  cf r;
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "=v"(r) : "v"(a), "v"(b), "v"(t));
  return r;
}
// c + i d = (c.x - d.y, c.y + d.x) and c - i d = (c.x + d.y, c.y - d.x) as one packed add each
__device__ __forceinline__ cf addi(cf c, cf d) {
  cf r;
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(c), "v"(d));
  return r;
}
__device__ __forceinline__ cf subi(cf c, cf d) {
  cf r;
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(c), "v"(d));
  return r;
}
// c + (SIGN i) d: SIGN < 0 is the forward transform
template <int SIGN> __device__ __forceinline__ cf add_si(cf c, cf d) { return SIGN < 0 ? subi(c, d) : addi(c, d); }
template <int SIGN> __device__ __forceinline__ cf sub_si(cf c, cf d) { return SIGN < 0 ? addi(c, d) : subi(c, d); }

// LDS index of sequence element i.  8 points per thread (the row kernels): an XOR swizzle of the low five bits,
// conflict-free for the radix-8 passes in a 32-lane / 64-bank model.  16 or 32 points per thread (the column
// kernels): one pad element per 16, i + i / 16 -- every access of a pass is then one per-thread base plus a
// compile-time offset (the DS instructions' immediate), where the swizzle cost three VALU instructions per
// access, 40 % of all the column kernel issued; the padded passes stay conflict-free but for one lane pair.
template <int E>
__device__ __forceinline__ int padi(int i) { return E >= 16 ? (i + (i >> 4)) : (i ^ ((i >> 3) & 31)); }
template <int E>
constexpr int padk(int k) { return E >= 16 ? k + k / 16 : k; }   // padi(a + k) - padi(a) for 16 | k (and for k < 16 when 16 | a)

template <int SIGN>
__device__ __forceinline__ void dft2(cf* v) {
  const cf a = v[0], b = v[1];
  v[0] = a + b;
  v[1] = a - b;
}

template <int SIGN>
__device__ __forceinline__ void dft4(cf* v) {
  const cf e0 = v[0] + v[2], e1 = v[0] - v[2];
  const cf o0 = v[1] + v[3], d = v[1] - v[3];
  v[0] = e0 + o0;
  v[1] = add_si<SIGN>(e1, d);
  v[2] = e0 - o0;
  v[3] = sub_si<SIGN>(e1, d);
}

template <int SIGN>
__device__ __forceinline__ void dft8(cf* v) {
  const float h = 0.70710678118654752f;
  cf a[4], b[4];
#pragma unroll
  for (int n = 0; n < 4; ++n) {
    a[n] = v[n] + v[n + 4];
    b[n] = v[n] - v[n + 4];
  }
  dft4<SIGN>(a);
  // b[n] *= W8^n, W8 = exp(SIGN 2 pi i / 8), folded into the 4-point transform of b:
  //   b1' = h (b1 + SIGN i b1),  b2' = SIGN i b2,  b3' = -h (b3 - SIGN i b3)
  const cf b1 = add_si<SIGN>(b[1], b[1]) * (cf){h, h};
  const cf x3 = sub_si<SIGN>(b[3], b[3]);
  const cf e0 = add_si<SIGN>(b[0], b[2]), e1 = sub_si<SIGN>(b[0], b[2]);
  const cf o0 = __builtin_elementwise_fma(x3, (cf){-h, -h}, b1), d = __builtin_elementwise_fma(x3, (cf){h, h}, b1);
  v[0] = a[0];
  v[1] = e0 + o0;
  v[2] = a[1];
  v[3] = add_si<SIGN>(e1, d);
  v[4] = a[2];
  v[5] = e0 - o0;
  v[6] = a[3];
  v[7] = sub_si<SIGN>(e1, d);
}

// 16 points as 4 x 4: four 4-point transforms over the stride-4 groups, twiddles W16^(q k), four 4-point
// transforms across the groups; X[k + 4 m] comes out of transform k at position m
template <int SIGN>
__device__ __forceinline__ void dft16(cf* v) {
  constexpr float c1 = 0.92387953251128674f, s1 = 0.38268343236508977f, h = 0.70710678118654752f;
  constexpr float sg = SIGN < 0 ? -1.f : 1.f;
  cf a[4][4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    a[q][0] = v[q]; a[q][1] = v[q + 4]; a[q][2] = v[q + 8]; a[q][3] = v[q + 12];
    dft4<SIGN>(a[q]);
  }
  // a[q][k] *= W16^(q k), W16 = exp(SIGN 2 pi i / 16)
  a[1][1] = cmul(a[1][1], mkc(c1, sg * s1));            // W^1
  a[1][2] = cmul(a[1][2], mkc(h, sg * h));              // W^2
  a[1][3] = cmul(a[1][3], mkc(s1, sg * c1));            // W^3
  a[2][1] = cmul(a[2][1], mkc(h, sg * h));              // W^2
  a[2][2] = add_si<SIGN>(mkc(0.f, 0.f), a[2][2]);       // W^4 = SIGN i
  a[2][3] = cmul(a[2][3], mkc(-h, sg * h));             // W^6
  a[3][1] = cmul(a[3][1], mkc(s1, sg * c1));            // W^3
  a[3][2] = cmul(a[3][2], mkc(-h, sg * h));             // W^6
  a[3][3] = cmul(a[3][3], mkc(-c1, -sg * s1));          // W^9 = -W^1
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    cf b[4] = {a[0][k], a[1][k], a[2][k], a[3][k]};
    dft4<SIGN>(b);
#pragma unroll
    for (int m = 0; m < 4; ++m) v[k + 4 * m] = b[m];
  }
}

template <int R, int SIGN>
__device__ __forceinline__ void dftR(cf* v) {
  if constexpr (R == 16) dft16<SIGN>(v);
  else if constexpr (R == 8) dft8<SIGN>(v);
  else if constexpr (R == 4) dft4<SIGN>(v);
  else dft2<SIGN>(v);
}

// Hand-over between the read half and the write half of a pass.  An FFT that one wave does on its own (N / E = 64)
// needs no workgroup barrier: the LDS operations of a wave complete in order, the fence only pins the compiler.
template <int NT>
__device__ __forceinline__ void fft_sync() {
  if constexpr (NT == 64) {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  } else {
    __syncthreads();
  }
}

// After a wave-private FFT, before other waves read its column: the workgroup barrier fft_sync left out.
template <int NT>
__device__ __forceinline__ void fft_publish() {
  if constexpr (NT == 64) __syncthreads();
}

// One Stockham radix-R pass over an N-point sequence held in LDS (padded index), done by
// N/E threads (E points each).  tw[k] = exp(-2 pi i k / N).  The read half leaves the twiddled, transformed
// values in registers; the write half puts them back (autosort order) and ends with a barrier (fft_sync).
template <int N, int E, int R, int SIGN, int Ns>
__device__ __forceinline__ void fft_pass_read(const cf* s, int j, const cf* __restrict__ tw, cf (&v)[E / R][R]) {
  constexpr int NT = N / E;
  constexpr int NB = E / R;
  cf w1[NB];
  // every read of the pass is issued before anything waits for one: left alone, the scheduler threads the reads
  // one by one into the twiddle chain and each then costs a full LDS round trip
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const int jj = j + b * NT;
    if constexpr (E >= 16 && (N / R) % 16 == 0) {
      const cf* sr = s + padi<E>(jj);
#pragma unroll
      for (int r = 0; r < R; ++r) v[b][r] = sr[padk<E>(r * (N / R))];
    } else {
#pragma unroll
      for (int r = 0; r < R; ++r) v[b][r] = s[padi<E>(jj + r * (N / R))];
    }
    if (Ns > 1) {
      static_assert((Ns - 1) * (N / (Ns * R)) < ((N / 2) < 256 ? (N / 2) : 256) || E == 8, "column passes stay inside the LDS twiddle table (FftGeom::TWL)");
      w1[b] = tw[(jj & (Ns - 1)) * (N / (Ns * R))];
    }
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    if (Ns > 1) {
      if (SIGN > 0) w1[b].y = -w1[b].y;
      // powers of w1 by a running product: two live twiddle registers instead of seven
      cf w = w1[b];
#pragma unroll
      for (int r = 1; r < R; ++r) {
        v[b][r] = cmul(v[b][r], w);
        if (r + 1 < R) w = cmul(w, w1[b]);
      }
    }
    dftR<R, SIGN>(v[b]);
  }
}

template <int N, int E, int R, int Ns>
__device__ __forceinline__ void fft_pass_write(cf* s, int j, const cf (&v)[E / R][R]) {
  constexpr int NT = N / E;
  constexpr int NB = E / R;
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const int jj = j + b * NT;
    const int base = (jj / Ns) * (Ns * R) + (jj & (Ns - 1));
    if constexpr (E >= 16 && (Ns % 16 == 0 || (Ns == 1 && R == 16))) {
      cf* sw = s + padi<E>(base);          // Ns = 1: base = 16 jj, the r < 16 behind it share its pad
#pragma unroll
      for (int r = 0; r < R; ++r) sw[Ns == 1 ? r : padk<E>(r * Ns)] = v[b][r];
    } else {
#pragma unroll
      for (int r = 0; r < R; ++r) s[padi<E>(base + r * Ns)] = v[b][r];
    }
  }
  fft_sync<N / E>();
}

template <int N, int E, int R, int SIGN, int Ns>
__device__ __forceinline__ void fft_pass(cf* s, int j, const cf* __restrict__ tw) {
  cf v[E / R][R];
  fft_pass_read<N, E, R, SIGN, Ns>(s, j, tw, v);
  fft_sync<N / E>();
  fft_pass_write<N, E, R, Ns>(s, j, v);
}

// 16-point passes where a thread holds 16 points (a pass less through LDS for N >= 256), else 8
template <int N, int E, int Ns>
constexpr int pass_radix() {
  constexpr int rem = N / Ns;
  return (E >= 16 && rem >= 16) ? 16 : (rem >= 8 ? 8 : rem);
}

template <int N, int E, int SIGN, int Ns>
__device__ __forceinline__ void fft_rec(cf* s, int j, const cf* __restrict__ tw) {
  if constexpr (Ns < N) {
    constexpr int R = pass_radix<N, E, Ns>();
    fft_pass<N, E, R, SIGN, Ns>(s, j, tw);
    fft_rec<N, E, SIGN, Ns * R>(s, j, tw);
  }
}

// In-LDS N-point FFT by N/E threads (j = thread index within the FFT); the data must be in place
// and the workgroup synchronised on entry.
template <int N, int SIGN, int E = 8>
__device__ __forceinline__ void fft_lds(cf* s, int j, const cf* __restrict__ tw) {
  fft_rec<N, E, SIGN, 1>(s, j, tw);
}

// Forward FFT whose LAST pass stays in registers: x[m] = X[j + m N/E], m < E.  That is exactly the set the first
// pass of an E-radix transform of the same length reads (its thread j takes the elements j + r N/E), so a
// pointwise product followed by the inverse transform needs no LDS round trip in between: the last forward
// pass's stores, the product's loads and stores and the first inverse pass's loads -- a quarter of the column
// kernel's LDS traffic at 2048 points -- and three of its barriers go.
template <int N, int E, int Ns>
__device__ __forceinline__ void fft_fwd_open_rec(cf* s, int j, const cf* __restrict__ tw, cf (&x)[E]) {
  constexpr int R = pass_radix<N, E, Ns>();
  if constexpr (Ns * R == N) {
    constexpr int NB = E / R;
    cf v[NB][R];
    fft_pass_read<N, E, R, -1, Ns>(s, j, tw, v);
    // thread j, batch b, output r is element (j + b N/E) + r Ns = j + (b + NB r) N/E
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int r = 0; r < R; ++r) x[b + NB * r] = v[b][r];
  } else {
    fft_pass<N, E, R, -1, Ns>(s, j, tw);
    fft_fwd_open_rec<N, E, Ns * R>(s, j, tw, x);
  }
}
// Transform whose LAST pass stays in registers, either direction, for a consumer that only looks at the values:
// x[m] is element pos[m] of the result, pos[] increasing (the inverse row pass takes the arg-max straight from them -- the last
// pass's stores, the barrier behind them and the read-back are a fifth of that kernel's LDS traffic).
template <int N, int E, int SIGN, int Ns>
__device__ __forceinline__ void fft_open_rec(cf* s, int j, const cf* __restrict__ tw, cf (&x)[E], int (&pos)[E]) {
  constexpr int R = pass_radix<N, E, Ns>();
  if constexpr (Ns * R == N) {
    constexpr int NB = E / R, NT = N / E;
    cf v[NB][R];
    fft_pass_read<N, E, R, SIGN, Ns>(s, j, tw, v);
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int r = 0; r < R; ++r) {
        x[r * NB + b] = v[b][r];               // in increasing order of position (NB NT = Ns)
        pos[r * NB + b] = j + b * NT + r * Ns;   // butterfly jj = j + b NT < Ns: its outputs are jj + r Ns
      }
  } else {
    fft_pass<N, E, R, SIGN, Ns>(s, j, tw);
    fft_open_rec<N, E, SIGN, Ns * R>(s, j, tw, x, pos);
  }
}
template <int N, int SIGN, int E = 8>
__device__ __forceinline__ void fft_lds_open(cf* s, int j, const cf* __restrict__ tw, cf (&x)[E], int (&pos)[E]) {
  static_assert(pass_radix<N, E, 1>() < N, "at least two passes");
  fft_open_rec<N, E, SIGN, 1>(s, j, tw, x, pos);
}
template <int N, int E>
__device__ __forceinline__ void fft_fwd_open(cf* s, int j, const cf* __restrict__ tw, cf (&x)[E]) {
  static_assert(pass_radix<N, E, 1>() < N, "at least two passes");
  fft_fwd_open_rec<N, E, 1>(s, j, tw, x);
}
// Inverse FFT whose first pass takes its input from registers (x[m] = element j + m N/E); the result is in LDS.
// Every thread of the FFT must be past its last LDS read of the forward transform before the first store:
// the barrier between the two halves of the pass does that, as in any other pass.
template <int N, int E>
__device__ __forceinline__ void fft_inv_regs(cf* s, int j, const cf* __restrict__ tw, cf (&x)[E]) {
  constexpr int R = pass_radix<N, E, 1>();
  static_assert(R == E, "the first pass is one E-point transform per thread");
  cf v[1][R];
#pragma unroll
  for (int r = 0; r < R; ++r) v[0][r] = x[r];
  dftR<R, +1>(v[0]);
  fft_sync<N / E>();
  fft_pass_write<N, E, R, 1>(s, j, v);
  fft_rec<N, E, +1, R>(s, j, tw);
}

// Columns per workgroup of the column kernels, for N-point columns.  4 columns x 8 (4) rows of the row-interleaved
// spectra are one 128-byte line, so 4 is the smallest tile that moves whole lines; larger tiles mean fewer, bigger
// workgroups.  Measured per size (round 3, scripts/pmc_harness estimate): 2048 points 4 (74 KB of LDS per tile as it
// is); 1024 points 8 (4: +5 %); 512 points 4 (16: +10 %; a 512 x 512 frame is 65 tiles instead of 17, which is
// also what a single-frame launch of the serial chain lives on); 256 points no difference between 4, 8 and 16;
// 128 points 8 (16: +2.5 %).
constexpr int col_tile(int N) { return N >= 2048 ? 4 : (N == 1024 ? 8 : (N >= 256 ? 4 : 8)); }

template <int N>
struct FftGeom {
  static constexpr int NT = N / 8;                            // threads per FFT
  static constexpr int PADN = N + N / 8;                      // padded LDS points per FFT
  static constexpr int G = (256 / NT) > 1 ? (256 / NT) : 1;   // row FFTs per workgroup
  static constexpr int TC = col_tile(N);                      // columns per workgroup
  static constexpr int CE = N >= 128 ? 16 : 8;                // points per thread in the column kernels
  static constexpr int CNT = N / CE;                          // threads per column FFT
  static constexpr int CWPS = 4;                              // waves per SIMD the column kernels are compiled for (LDS allows no more)
  static constexpr int TWL = (N / 2) < 256 ? (N / 2) : 256;   // twiddles the column passes index (k N / (Ns R) < N / R of the last pass)
  // LDS stride between columns: the padded column (N + N/16 for 16 points per thread) rounded up to 1 mod 16, so
  // that the 16 lanes that stage 4-row pieces of 4 columns x 2 half blocks x 2 blocks hit 16 different bank pairs
  static constexpr int CS = ((CE >= 16 ? N + N / 16 : PADN) + 15) / 16 * 16 + 1;
  // row pairs per workgroup of the inverse row kernel: at least the 4 rows of Q's interleave (below)
  static constexpr int RPQ = G >= 2 ? G : 2;
};

// Spectra in HBM.  The moving planes' row spectra F (half pairs) and the column pass's output Q (float pairs) are
// ROW-INTERLEAVED: the same column of 8 (F) or 4 (Q) consecutive rows is contiguous,
//   F[((y / 8) * Wp + v) * 8 + y % 8],   Q[((y / 4) * Wp + v) * 4 + y % 4],
// so that a tile of 4 columns x 8 (4) rows is one 128-byte line.  The column kernel owns whole columns: with
// row-major spectra every line of F was fetched by 8 workgroups and every line of Q written by 4, 16 and 32 bytes at
// a time, and the kernel ran at the rate of those requests, not of its FFTs (measured with the addresses faked
// contiguous: 798 -> 539 us per 64 frames of 2048^2).  The row kernels take 8 (4) rows per workgroup and stream whole
// blocks; the refinement reads 4 rows of a column as 32 contiguous bytes.  The reference plane's float row spectrum
// (one per call) stays row-major.
__device__ __forceinline__ size_t f_at(int y, int v, int Wp) { return ((size_t)(y >> 3) * Wp + v) * 8 + (y & 7); }
__device__ __forceinline__ size_t q_at(int y, int v, int Wp) { return ((size_t)(y >> 2) * Wp + v) * 4 + (y & 3); }

__global__ void twiddle_kernel(cf* tw, int n) {
  int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < n) {
    double s, c;
    sincospi(-2.0 * (double)k / (double)n, &s, &c);
    tw[k] = mkc((float)c, (float)s);
  }
}

// Background level of a plane that is subtracted on load: a kernel that reads raw u16 pixels v of plane i
// sees max(v - level, 0), which is what sub_half_kernel (bg.hip) would have stored for sub2 = 2 * background:
// max(2 v - sub2, 0) >> 1 == max(v - ((sub2 + 1) >> 1), 0) for integers.  sub2 == nullptr: level 0.
__device__ __forceinline__ int plane_level(const int* __restrict__ sub2, long long i) {
  if (!sub2) return 0;
  const int s = (sub2[i] + 1) >> 1;
  return min(max(s, 0), 65535);
}

__device__ __forceinline__ float load_src(const void* __restrict__ p, int kind, size_t i) {
  if (kind == 0) return (float)reinterpret_cast<const uint16_t*>(p)[i];
  if (kind == 1) return (float)reinterpret_cast<const uint32_t*>(p)[i];
  return reinterpret_cast<const float*>(p)[i];
}

#ifndef ROWS_E16_MIN
#define ROWS_E16_MIN 1024   // shortest row that the row kernels transform with 16 points per thread (512: C1 estimate 0.213 against 0.206 ms)
#endif
// K1 ---------------------------------------------------------------------------------------
// Row spectra are stored in one of two forms.  The moving planes' (always u16 data, so |F| <= Px * 65535) go
// to HBM as packed half pairs, scaled by FS16<N> = 0.5 / Px: half the bytes of the path's largest scratch
// stream, and small enough for the column kernel to keep the next frame's tile in flight in registers.  The
// rounding is harmless here (relative 2^-12 on coefficients that carry 1e-2 .. 1 of photon noise each;
// measured on the synthetic stacks: 3e-5 px, scripts/f16_spectra_sim.py); the same rounding on Q is not
// (1e-2 px: the refinement differences neighbouring values of a broad peak), so Q stays float.  The reference
// plane's row spectrum (any of the three source kinds, unbounded for float) stays float on its way to mode 0.
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ h2 to_h2(cf v) { return __builtin_convertvector(v, h2); }
__device__ __forceinline__ cf from_h2(h2 v) { return __builtin_convertvector(v, cf); }
template <int N> constexpr float fs16() { return 0.5f / (float)N; }

// Storage of the spectra F and G: half pairs, or float pairs (F32) for FFT windows below 256 points on a side.
// There few coefficients average the rounding of half spectra (relative 2^-12 each) and a correlation peak between
// two different channels can be flat; such planes are small, so their spectra cost nothing as floats.  (The 0.02 px
// misses of the randomised channel-shift run that round 2 put down to this rounding were the Nyquist row's factor in
// the refinement, see up_row_sum; with both changes 1500 randomised planes stay within 0.01 px, most within 1e-5.)
// Both forms move in pieces of four values: P4.
template <bool F32> struct Sp;
template <> struct Sp<false> {
  typedef h2 E;
  typedef h8 P4;
  static __device__ __forceinline__ P4 pack(cf a, cf b, cf c, cf d) {
    const h2 h0 = to_h2(a), h1 = to_h2(b), h2_ = to_h2(c), h3 = to_h2(d);
    return (h8){h0.x, h0.y, h1.x, h1.y, h2_.x, h2_.y, h3.x, h3.y};
  }
  static __device__ __forceinline__ cf get(const P4& v, int i) { return from_h2((h2){v[2 * i], v[2 * i + 1]}); }
  static __device__ __forceinline__ P4 load(const E* p) { return *reinterpret_cast<const P4*>(p); }
  static __device__ __forceinline__ P4 load_nt(const E* p) { return __builtin_nontemporal_load(reinterpret_cast<const P4*>(p)); }
  static __device__ __forceinline__ void store(E* p, const P4& v) { *reinterpret_cast<P4*>(p) = v; }
};
template <> struct Sp<true> {
  typedef cf E;
  struct P4 { float4 lo, hi; };
  static __device__ __forceinline__ P4 pack(cf a, cf b, cf c, cf d) {
    return P4{make_float4(a.x, a.y, b.x, b.y), make_float4(c.x, c.y, d.x, d.y)};
  }
  static __device__ __forceinline__ cf get(const P4& v, int i) {
    const float4& q = i < 2 ? v.lo : v.hi;
    return (i & 1) ? mkc(q.z, q.w) : mkc(q.x, q.y);
  }
  static __device__ __forceinline__ P4 load(const E* p) {
    return P4{reinterpret_cast<const float4*>(p)[0], reinterpret_cast<const float4*>(p)[1]};
  }
  static __device__ __forceinline__ P4 load_nt(const E* p) { return load(p); }
  static __device__ __forceinline__ void store(E* p, const P4& v) {
    reinterpret_cast<float4*>(p)[0] = v.lo;
    reinterpret_cast<float4*>(p)[1] = v.hi;
  }
};

// two real rows per complex transform: Fa[k] = (Z[k] + conj Z[N-k]) / 2, Fb[k] = (Z[k] - conj Z[N-k]) / 2i
template <int N, int E = 8>
__device__ __forceinline__ void split_rows(const cf* sp, int k, cf& fa, cf& fb) {
  const cf zk = sp[padi<E>(k)], zn = sp[padi<E>((N - k) & (N - 1))];
  fa = mkc(0.5f * (zk.x + zn.x), 0.5f * (zk.y - zn.y));
  fb = mkc(0.5f * (zk.y + zn.y), -0.5f * (zk.x - zn.x));
}

// the reference plane (any source kind): float row spectrum, row-major
template <int N>
__global__ __launch_bounds__(FftGeom<N>::G* FftGeom<N>::NT) void rows_r2c_kernel(
    const void* __restrict__ src, int kind, long long plane_stride, int srcX, int oy, int ox, float in_scale,
    cf* __restrict__ F, long long F_stride, int Wp, const cf* __restrict__ tw, const int* __restrict__ sub2,
    long long sub_stride) {
  using Gm = FftGeom<N>;
  extern __shared__ cf lds_c[];
  const int g = threadIdx.x / Gm::NT, j = threadIdx.x % Gm::NT;
  cf* s = lds_c + g * Gm::PADN;
  const int ra = 2 * (blockIdx.x * Gm::G + g);
  const int lv = kind == 0 ? plane_level(sub2, (long long)blockIdx.y * sub_stride) : 0;   // u16 sources only
  const size_t offa = (size_t)blockIdx.y * plane_stride + (size_t)(oy + ra) * srcX + ox;
  const size_t offb = offa + srcX;
#pragma unroll
  for (int m = 0; m < 8; ++m) {
    int n = j + m * Gm::NT;
    // (exact in f32: pixel values and levels are integers below 2^16)
    float a = load_src(src, kind, offa + n), b = load_src(src, kind, offb + n);
    if (kind == 0) {
      a = fmaxf(a - (float)lv, 0.f);
      b = fmaxf(b - (float)lv, 0.f);
    }
    s[padi<8>(n)] = mkc(a * in_scale, b * in_scale);
  }
  __syncthreads();
  fft_lds<N, -1>(s, j, tw);
  cf* Fa = F + (size_t)blockIdx.y * F_stride + (size_t)ra * Wp;
  cf* Fb = Fa + Wp;
  for (int k = j; k <= N / 2; k += Gm::NT) {
    cf fa, fb;
    split_rows<N>(s, k, fa, fb);
    Fa[k] = fa;
    Fb[k] = fb;
  }
}

// The moving planes (u16): half-pair row spectra in the row-interleaved layout.  A workgroup produces whole 8-row
// blocks.  Where a transform takes 128 threads or more it does so in two steps of two row pairs each: the first
// step's results wait in registers (16 bytes per thread and round) while the second step's rows, requested before
// the first transform started, go through the same LDS, and both leave together as 32 contiguous bytes per column --
// whole lines from one wave, with the workgroups as small as the transforms allow (four per CU at 2048 points).
// Rows of 1024 points and more are transformed with 16 points per thread, as in the inverse row pass (three LDS
// passes instead of four, half the threads, padded index): 533-542 against 594-601 us per 128 frames of 2048^2.
#ifndef R2C_E
#define R2C_E 16
#endif
template <int N>
struct R16Geom {
  using Gm = FftGeom<N>;
  static constexpr int E = N >= ROWS_E16_MIN ? R2C_E : 8;        // points per thread of the transform
  static constexpr int NT = N / E;                       // threads per transform
  static constexpr int PADN = E >= 16 ? N + N / 16 : Gm::PADN;   // LDS points per transform (padi<E>)
  static constexpr int G = (256 / NT) > 1 ? (256 / NT) : 1;
  static constexpr int PS = G >= 4 ? G : 2;              // row pairs per step
  static constexpr int STEPS = G >= 4 ? 1 : 2;
  static constexpr int NTH = PS * NT;
  static constexpr int ROWS = 2 * PS * STEPS;            // rows per workgroup: a multiple of 8
  static constexpr int NQ = PS / 2;                      // row quads per step
  static constexpr int LG = NTH / NQ < 32 ? NTH / NQ : 32;
  static constexpr int KSTEP = NTH / NQ;                 // bins a round of stores covers
  static constexpr int NR = (N / 2) / KSTEP;             // rounds that cover the bins below Nyquist (2 at 8 points per thread)
  static constexpr int WPS = E >= 16 ? 4 : 8;            // waves per SIMD the kernel is compiled for
  static_assert(ROWS % 8 == 0 && NTH % (LG * NQ) == 0 && KSTEP * NR == N / 2, "row blocks, lane groups, rounds");
};
template <int N, bool vec, bool F32>
__global__ __launch_bounds__(R16Geom<N>::NTH, R16Geom<N>::WPS) void rows_r2c16_kernel(
    const uint16_t* __restrict__ src, long long plane_stride, int srcX, int oy, int ox, typename Sp<F32>::E* __restrict__ F,
    long long F_stride, int Wp, const cf* __restrict__ tw, const int* __restrict__ sub2, long long sub_stride) {
  using Rg = R16Geom<N>;
  constexpr int E = Rg::E, NV = E / 8;
  extern __shared__ cf lds_c[];
  const int t = threadIdx.x;
  const int g = t / Rg::NT;
  int j = t % Rg::NT;
  cf* s = lds_c + g * Rg::PADN;
  const int r0 = blockIdx.x * Rg::ROWS;
  const int lv = plane_level(sub2, (long long)blockIdx.y * sub_stride);
  const float in_scale = fs16<N>();
  const uint16_t* p = src + (size_t)blockIdx.y * plane_stride + (size_t)oy * srcX + ox;
  auto row_of = [&](int step) { return r0 + step * 2 * Rg::PS + 2 * g; };
  // 16-byte aligned rows: E / 8 8-pixel vectors per row and thread (vector j + v NT: a wave reads whole lines),
  // fetched a step ahead
  auto fetch = [&](int step, u16x8 (&qa)[NV], u16x8 (&qb)[NV]) {
    const uint16_t* pa = p + (size_t)row_of(step) * srcX + 8 * j;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      qa[v] = load8(pa + 8 * v * Rg::NT);
      qb[v] = load8(pa + 8 * v * Rg::NT + srcX);
    }
  };
  auto stage = [&](int step, const u16x8 (&qa)[NV], const u16x8 (&qb)[NV]) {
    if (vec) {
#pragma unroll
      for (int v = 0; v < NV; ++v)
#pragma unroll
        for (int k = 0; k < 8; ++k)
          s[padi<E>(8 * (j + v * Rg::NT) + k)] =
              mkc((float)max((int)qa[v].v[k] - lv, 0) * in_scale, (float)max((int)qb[v].v[k] - lv, 0) * in_scale);
    } else {
      const uint16_t* pa = p + (size_t)row_of(step) * srcX;
#pragma unroll
      for (int m = 0; m < E; ++m) {
        const int n = j + m * Rg::NT;
        s[padi<E>(n)] = mkc((float)max((int)pa[n] - lv, 0) * in_scale, (float)max((int)pa[srcX + n] - lv, 0) * in_scale);
      }
    }
  };
  // 4 rows (two pairs) of one column as 8 halves; 32 consecutive lanes take consecutive columns of one row quad
  // (conflict-free LDS reads)
  const int q = (t / Rg::LG) % Rg::NQ;
  const int k0 = (t % Rg::LG) + Rg::LG * (t / (Rg::LG * Rg::NQ));   // < KSTEP
  const cf* sa = lds_c + (2 * q) * Rg::PADN;
  const cf* sb = sa + Rg::PADN;
  auto quad = [&](int k) {
    cf f0, f1, f2, f3;
    split_rows<N, E>(sa, k, f0, f1);
    split_rows<N, E>(sb, k, f2, f3);
    return Sp<F32>::pack(f0, f1, f2, f3);
  };
  typedef typename Sp<F32>::E FE;
  typedef typename Sp<F32>::P4 P4;
  FE* Fo = F + (size_t)blockIdx.y * F_stride;
  u16x8 qa[NV], qb[NV];
  if (vec) fetch(0, qa, qb);
  stage(0, qa, qb);
  __syncthreads();
  if (Rg::STEPS == 2 && vec) fetch(1, qa, qb);
  fft_lds<N, -1, E>(s, j, tw);
  fft_publish<Rg::NT>();   // the quads are read across the transforms' own thread groups
  if constexpr (Rg::STEPS == 1) {
    const int y0 = r0 + 4 * q;
#pragma unroll
    for (int r = 0; r < Rg::NR; ++r) Sp<F32>::store(Fo + f_at(y0, k0 + r * Rg::KSTEP, Wp), quad(k0 + r * Rg::KSTEP));
    if (k0 == 0) Sp<F32>::store(Fo + f_at(y0, N / 2, Wp), quad(N / 2));
  } else {
    static_assert(Rg::STEPS == 1 || Rg::NQ == 1, "two steps: one quad per step");
    P4 a[Rg::NR];
#pragma unroll
    for (int r = 0; r < Rg::NR; ++r) a[r] = quad(k0 + r * Rg::KSTEP);
    FE* o = Fo + f_at(r0, k0, Wp);          // rows r0 .. r0 + 3 of column k0, rows r0 + 4 .. + 7 behind them
    if (k0 == 0) Sp<F32>::store(o + (size_t)(N / 2) * 8, quad(N / 2));   // (the Nyquist column: half lines)
    __syncthreads();   // the first step's spectra are in registers
    stage(1, qa, qb);
    __syncthreads();
    asm volatile("" : "+v"(j));   // the second transform works out its LDS offsets anew instead of holding the first one's
    fft_lds<N, -1, E>(s, j, tw);
    fft_publish<Rg::NT>();
#pragma unroll
    for (int r = 0; r < Rg::NR; ++r) {
      Sp<F32>::store(o + (size_t)r * Rg::KSTEP * 8, a[r]);
      Sp<F32>::store(o + (size_t)r * Rg::KSTEP * 8 + 4, quad(k0 + r * Rg::KSTEP));
    }
    if (k0 == 0) Sp<F32>::store(o + (size_t)(N / 2) * 8 + 4, quad(N / 2));
  }
}

// K2 ---------------------------------------------------------------------------------------
// Column pass on tiles of TC columns (XCD-remapped so that neighbouring tiles, which share cache lines of the
// spectra, run on one L2).  Tiles of F and Q move as whole 128-byte lines of the row-interleaved layouts.
// Thread tid = c * CNT + j works on column c of the tile; the last forward pass leaves it the spectrum values
// u = j + m * CNT, m < CE, in registers (fft_fwd_open), where they meet the reference's and go straight into the
// inverse transform.
//   cols_ref_kernel     the reference: float row spectrum -> column FFT -> G as half pairs in the order the threads
//                       of the cross-power kernel use them: chunk q (16 bytes = elements m = 4 q .. 4 q + 3) of
//                       thread tid at (q * NTH + tid) * 4, so that a wave's load is one contiguous KiB; scaled per
//                       tile by a power of two so that any float reference fits (gscale[tile] = the factor that
//                       undoes it).
//   cols_xpower_kernel  the moving frames, one (tile, frame) per workgroup: half tile -> LDS, column FFT,
//                       P = G conj(F) (DC zeroed, optional / |P|; G one for all frames or one per frame: channel
//                       shift pairs), inverse FFT, Q[y][v] stored as float (spatial along y, spectral along x).
//   cols_fused_kernel   the serial 'previous' chain: the reference arrives as its ROW spectrum (half pairs, F's
//                       layout) and gets its column FFT in a second LDS tile of the same workgroup.
template <int N>
struct ColGeom {
  using Gm = FftGeom<N>;
  static constexpr int TC = Gm::TC, CE = Gm::CE, CNT = Gm::CNT, CS = Gm::CS;
  static constexpr int NTH = TC * CNT;
  static constexpr int NLG = CE / 4;             // 16-byte chunks of G per thread
  static constexpr int NL4 = CE / 4;             // 16-byte chunks (4 rows of a column) of F per thread
  static_assert(CE % 4 == 0 && N % 8 == 0, "whole chunks per thread, whole row blocks");
};
// the column kernels keep the twiddles their passes use in LDS: a pass then waits for an LDS read, not for L2
template <int N>
__device__ __forceinline__ void stage_twiddles(cf* tw_s, const cf* __restrict__ tw, int tid) {
  for (int i = tid; i < FftGeom<N>::TWL; i += ColGeom<N>::NTH) tw_s[i] = tw[i];
}
// Half tile of the row-interleaved spectrum F -> registers: 16-byte chunks of 4 rows of one column.  The tile's
// share of an 8-row block is TC * 2 contiguous chunks (column c, rows 0..3 or 4..7 of the block): a thread owns the
// chunks idx = tid + m * NTH, m < NL4, of the tile in that order, a wave's load is a run of whole lines.
template <int N, bool F32>
__device__ __forceinline__ void load_half_tile(const typename Sp<F32>::E* __restrict__ Ff, int Wp, int v0, int tid,
                                               typename Sp<F32>::P4 (&w)[ColGeom<N>::NL4]) {
  using Cg = ColGeom<N>;
#pragma unroll
  for (int m = 0; m < Cg::NL4; ++m) {
    const int idx = tid + m * Cg::NTH;
    // (streaming: a line of F is used once, by this tile)
    w[m] = Sp<F32>::load_nt(Ff + ((size_t)(idx / (2 * Cg::TC)) * Wp + v0) * 8 + (idx % (2 * Cg::TC)) * 4);
  }
}
template <int N, bool F32>
__device__ __forceinline__ void stage_half_tile(cf* lds, int v0, int nvalid, int tid,
                                                const typename Sp<F32>::P4 (&w)[ColGeom<N>::NL4]) {
  using Cg = ColGeom<N>;
#pragma unroll
  for (int m = 0; m < Cg::NL4; ++m) {
    const int idx = tid + m * Cg::NTH;
    const int wi = idx % (2 * Cg::TC), c = wi >> 1, u0 = (idx / (2 * Cg::TC)) * 8 + (wi & 1) * 4;
    cf* d = lds + c * Cg::CS;
    if (v0 + Cg::TC <= nvalid) {   // (uniform) every tile but the last
#pragma unroll
      for (int h = 0; h < 4; ++h)   // (16 points per thread: 4 | u0, the four rows share a pad group and are contiguous)
        d[padi<Cg::CE>(u0 + h)] = Sp<F32>::get(w[m], h);
    } else {
      const bool ok = v0 + c < nvalid;   // padding columns hold whatever the workspace held
#pragma unroll
      for (int h = 0; h < 4; ++h) d[padi<Cg::CE>(u0 + h)] = ok ? Sp<F32>::get(w[m], h) : mkc(0.f, 0.f);
    }
  }
}
// Q tile (LDS) -> row-interleaved float spectrum: two rows of one column per 16-byte store, the tile's share of a
// 4-row block is TC * 2 contiguous chunks.  A wave stores 64 consecutive chunks (whole lines); which lane takes
// which of them is free, and for 4-column tiles the lanes are dealt so that each half wave reads two columns x
// 8 blocks x 2 row pairs = 32 different bank pairs (in chunk order columns 0 and 2 would collide).
template <int N>
__device__ __forceinline__ void store_float_tile(const cf* lds, cf* __restrict__ Qf, int Wp, int v0, int tid) {
  using Cg = ColGeom<N>;
#pragma unroll 4
  for (int m = 0; m < Cg::CE / 2; ++m) {
    int idx = tid + m * Cg::NTH;
    if constexpr (Cg::TC == 4) {
      const int l = idx & 63;
      idx = (idx & ~63) + ((l >> 2) & 7) * 8 + ((l >> 5) * 2 + (l & 1)) * 2 + ((l >> 1) & 1);
    }
    const int wi = idx % (2 * Cg::TC), c = wi >> 1, u = (idx / (2 * Cg::TC)) * 4 + (wi & 1) * 2;
    const cf* sp = lds + c * Cg::CS;
    const cf a = sp[padi<Cg::CE>(u)], b2 = sp[padi<Cg::CE>(u + 1)];
    *reinterpret_cast<float4*>(Qf + ((size_t)(idx / (2 * Cg::TC)) * Wp + v0) * 4 + wi * 2) = make_float4(a.x, a.y, b2.x, b2.y);
  }
}
// f conj(g) * scale (g = the moving frame's value, f = the reference's) as three packed instructions:
// (g.x, g.x) f, then + (g.y, g.y) (f.y, -f.x) -- the half selects and the sign ride in the second one
__device__ __forceinline__ cf xpower(cf g, cf f, float scale) {
  const cf t = (cf){g.x, g.x} * f;
  cf r;
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_hi:[0,1,0]" : "=v"(r) : "v"(g), "v"(f), "v"(t));
  return r * (cf){scale, scale};
}
// optional whitening P / |P| (IAP_REG_NORMALIZE)
__device__ __forceinline__ cf unit_or_zero(cf p) {
  const float m2 = p.x * p.x + p.y * p.y;
  const float inv = m2 > 0.f ? rsqrtf(m2) : 0.f;
  return p * (cf){inv, inv};
}
// One column of the tile, already staged in LDS: forward FFT, times the reference (ref(m) = its value at
// u = j + m * CNT), DC zeroed, inverse FFT back into the same LDS column.
template <int N, typename REF>
__device__ __forceinline__ void column_correlate(cf* s, int j, const cf* __restrict__ tw_s, REF ref, float sc,
                                                 int normalize, bool dc_column) {
  using Cg = ColGeom<N>;
  cf x[Cg::CE];
  fft_fwd_open<N, Cg::CE>(s, j, tw_s, x);
#pragma unroll
  for (int m = 0; m < Cg::CE; ++m) x[m] = xpower(x[m], ref(m), sc);
  if (normalize) {   // (uniform)
#pragma unroll
    for (int m = 0; m < Cg::CE; ++m) x[m] = unit_or_zero(x[m]);
  }
  if (dc_column && j == 0) x[0] = mkc(0.f, 0.f);
  fft_inv_regs<N, Cg::CE>(s, j, tw_s, x);
}

template <int N, bool F32>
__global__ __launch_bounds__(FftGeom<N>::TC* FftGeom<N>::CNT, FftGeom<N>::CWPS) void cols_ref_kernel(
    const cf* __restrict__ F, long long F_stride, typename Sp<F32>::E* __restrict__ G, long long G_stride, float* __restrict__ gscale,
    int gs_stride, int Wp, int nvalid, const cf* __restrict__ tw) {
  using Cg = ColGeom<N>;
  extern __shared__ cf lds_c[];
  __shared__ float wmax[16];
  __shared__ cf tw_s[FftGeom<N>::TWL];
  const int tid = threadIdx.x;
  stage_twiddles<N>(tw_s, tw, tid);
  const int c = tid / Cg::CNT, j = tid % Cg::CNT;
  const unsigned frame = blockIdx.y;
  const int tile = xcd_remap(blockIdx.x, gridDim.x), v0 = tile * Cg::TC;
  const cf* Ff = F + (size_t)frame * F_stride;
  constexpr int HP = Cg::TC / 2;
  {
    float4 w[Cg::CE / 2];
#pragma unroll
    for (int m = 0; m < Cg::CE / 2; ++m) {
      const int idx = tid + m * Cg::NTH;
      w[m] = *reinterpret_cast<const float4*>(Ff + (size_t)(idx / HP) * Wp + v0 + 2 * (idx % HP));
    }
#pragma unroll
    for (int m = 0; m < Cg::CE / 2; ++m) {
      const int idx = tid + m * Cg::NTH;
      const int pc = idx % HP, u = idx / HP;
      const bool oka = v0 + 2 * pc < nvalid, okb = v0 + 2 * pc + 1 < nvalid;
      lds_c[(2 * pc) * Cg::CS + padi<Cg::CE>(u)] = oka ? mkc(w[m].x, w[m].y) : mkc(0.f, 0.f);
      lds_c[(2 * pc + 1) * Cg::CS + padi<Cg::CE>(u)] = okb ? mkc(w[m].z, w[m].w) : mkc(0.f, 0.f);
    }
  }
  __syncthreads();
  cf v[Cg::CE];    // this thread's column at u = j + m * CNT
  fft_fwd_open<N, Cg::CE>(lds_c + c * Cg::CS, j, tw_s, v);
  // largest component of the tile -> power-of-two scale that brings it into [16384, 32768)
  float mx = 0.f;
#pragma unroll
  for (int m = 0; m < Cg::CE; ++m) mx = fmaxf(mx, fmaxf(fabsf(v[m].x), fabsf(v[m].y)));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  if ((tid & 63) == 0) wmax[tid >> 6] = mx;
  __syncthreads();
  mx = 0.f;
  for (int k = 0; k < (Cg::NTH + 63) / 64; ++k) mx = fmaxf(mx, wmax[k]);
  int ex = 0;
  float sc = 1.f;
  if (!F32 && mx > 0.f && mx < INFINITY) {   // (float spectra are stored as they are)
    (void)frexpf(mx, &ex);          // mx = m 2^ex, m in [0.5, 1)
    sc = ldexpf(1.f, 15 - ex);
  }
  if (tid == 0) gscale[(size_t)frame * gs_stride + tile] = 1.0f / sc;
  typename Sp<F32>::E* Gt = G + (size_t)frame * G_stride + (size_t)v0 * N;
  const cf s2 = {sc, sc};
#pragma unroll
  for (int q = 0; q < Cg::NLG; ++q)
    Sp<F32>::store(Gt + ((size_t)q * Cg::NTH + tid) * 4,
                   Sp<F32>::pack(v[4 * q] * s2, v[4 * q + 1] * s2, v[4 * q + 2] * s2, v[4 * q + 3] * s2));
}

template <int N, bool F32>
__global__ __launch_bounds__(FftGeom<N>::TC* FftGeom<N>::CNT, FftGeom<N>::CWPS) void cols_xpower_kernel(
    const typename Sp<F32>::E* __restrict__ F, long long F_stride, const typename Sp<F32>::E* __restrict__ G, long long G_stride,
    const float* __restrict__ gscale, int gs_stride, cf* __restrict__ Q, long long Q_stride, int Wp, int nvalid,
    const cf* __restrict__ tw, int normalize, float scale, int ntiles, int nitems) {
  using Cg = ColGeom<N>;
  extern __shared__ cf lds_c[];
  const int tid0 = threadIdx.x;
  __shared__ cf tw_s[FftGeom<N>::TWL];
  stage_twiddles<N>(tw_s, tw, tid0);
  // Persistent workgroups: item i = frame * ntiles + tile; a workgroup takes i = b, b + grid, ...  The workgroups of
  // one XCD hold consecutive b, so at any moment an XCD works on neighbouring tiles of a frame and F's 128-byte lines
  // are used up by one L2.  The next item's half tile and this item's chunks of the reference are requested before
  // the forward FFT: the memory latency (most of a one-shot workgroup's life: of its 56 kcycles, 20 went to waiting
  // for F and 10 for G) runs under the FFTs.
  // The staging of an item's half tile sits at the END of the item before it, behind that item's stores: there the
  // compiler knows how many younger operations follow the prefetch (vmcnt counts in order) and waits for the loads
  // alone.  At the top of the loop it merged the loop's entry with its back edge and waited for everything: every
  // item then began by draining the stores of the one before (~60 of 570 us per 64 frames of 2048^2).
  const int b = xcd_remap(blockIdx.x, gridDim.x);
  typedef typename Sp<F32>::P4 P4;
  P4 fa[Cg::NL4];
  if (b < nitems) {
    load_half_tile<N, F32>(F + (size_t)(b / ntiles) * F_stride, Wp, (b % ntiles) * Cg::TC, tid0, fa);
    stage_half_tile<N, F32>(lds_c, (b % ntiles) * Cg::TC, nvalid, tid0, fa);
  }
  for (int item = b; item < nitems; item += gridDim.x) {
    const int f = item / ntiles, tile = item - f * ntiles, v0 = tile * Cg::TC;
    // the thread index is made opaque per item: the LDS offsets of all passes depend on nothing else, and hoisted
    // out of this loop they would fill the register file (spills wait on the same counter as the prefetch)
    int tid = tid0;
    asm volatile("" : "+v"(tid));
    const int c = tid / Cg::CNT, j = tid % Cg::CNT;
    __syncthreads();   // the tile is staged
    const int nx = item + gridDim.x;
    const int nv0 = nx < nitems ? (nx % ntiles) * Cg::TC : 0;
    if (nx < nitems) load_half_tile<N, F32>(F + (size_t)(nx / ntiles) * F_stride, Wp, nv0, tid, fa);
    // the reference's chunks: contiguous 16 bytes per lane, served by L2 / the Infinity Cache (every frame re-reads them)
    const int gf = G_stride ? f : 0;
    const typename Sp<F32>::E* Gt = G + (size_t)gf * G_stride + (size_t)v0 * N;
    const float sc = scale * gscale[(size_t)gf * gs_stride + tile];
    P4 g[Cg::NLG];
#pragma unroll
    for (int q = 0; q < Cg::NLG; ++q) g[q] = Sp<F32>::load(Gt + ((size_t)q * Cg::NTH + tid) * 4);
    column_correlate<N>(lds_c + c * Cg::CS, j, tw_s, [&](int m) { return Sp<F32>::get(g[m >> 2], m & 3); }, sc,
                        normalize, v0 + c == 0);
    fft_publish<Cg::CNT>();
    store_float_tile<N>(lds_c, Q + (size_t)f * Q_stride, Wp, v0, tid);
    __syncthreads();   // the tile is free for the next item
    if (nx < nitems) stage_half_tile<N, F32>(lds_c, nv0, nvalid, tid, fa);
  }
}

template <int N, bool F32>
__global__ __launch_bounds__(FftGeom<N>::TC* FftGeom<N>::CNT, 2) void cols_fused_kernel(
    const typename Sp<F32>::E* __restrict__ F, long long F_stride, const typename Sp<F32>::E* __restrict__ Gref, long long G_stride,
    cf* __restrict__ Q, long long Q_stride, int Wp, int nvalid, const cf* __restrict__ tw, int normalize, float scale) {
  using Cg = ColGeom<N>;
  extern __shared__ cf lds_c[];
  const int tid = threadIdx.x;
  const int c = tid / Cg::CNT, j = tid % Cg::CNT;
  const unsigned frame = blockIdx.y;
  const int v0 = xcd_remap(blockIdx.x, gridDim.x) * Cg::TC;
  cf* const lds_g = lds_c + Cg::TC * Cg::CS;     // the reference's tile
  __shared__ cf tw_s[FftGeom<N>::TWL];
  stage_twiddles<N>(tw_s, tw, tid);
  {
    typename Sp<F32>::P4 w[Cg::NL4];
    load_half_tile<N, F32>(F + (size_t)frame * F_stride, Wp, v0, tid, w);
    stage_half_tile<N, F32>(lds_c, v0, nvalid, tid, w);
    load_half_tile<N, F32>(Gref + (size_t)frame * G_stride, Wp, v0, tid, w);
    stage_half_tile<N, F32>(lds_g, v0, nvalid, tid, w);
  }
  __syncthreads();
  // the reference's column transform ends in LDS; thread (c, j) then reads back its own column's values only
  const cf* sg = lds_g + c * Cg::CS;
  fft_lds<N, -1, Cg::CE>(lds_g + c * Cg::CS, j, tw_s);
  column_correlate<N>(lds_c + c * Cg::CS, j, tw_s, [&](int m) { return sg[padi<Cg::CE>(j + m * Cg::CNT)]; }, scale,
                      normalize, v0 + c == 0);
  fft_publish<Cg::CNT>();
  store_float_tile<N>(lds_c, Q + (size_t)frame * Q_stride, Wp, v0, tid);
}

// K3 ---------------------------------------------------------------------------------------
__device__ __forceinline__ void amax_take(float& v, int& i, float ov, int oi) {
  if (ov > v || (ov == v && oi < i)) {
    v = ov;
    i = oi;
  }
}

__device__ __forceinline__ void block_argmax(float& v, int& i, float* shv, int* shi) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    float ov = __shfl_down(v, o, 64);
    int oi = __shfl_down(i, o, 64);
    amax_take(v, i, ov, oi);
  }
  const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  if ((threadIdx.x & 63) == 0) {
    shv[w] = v;
    shi[w] = i;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int k = 1; k < nw; ++k) amax_take(v, i, shv[k], shi[k]);
  }
}

// The inverse row pass.  Rows of 1024 points and more are transformed with 16 points per thread (three passes
// 16 x 16 x 8 through LDS instead of four, padded index, half the threads): 630-660 against 720-740 us per 128 frames
// of 2048^2 under the profiler, estimate -0.3 ms per C2 step in alternating runs.
#ifndef C2R_E
#define C2R_E 16
#endif
template <int N>
struct C2rGeom {
  static constexpr int E = N >= ROWS_E16_MIN ? C2R_E : 8;
  static constexpr int NT = N / E;                                            // threads per transform
  static constexpr int PADN = E >= 16 ? N + N / 16 : FftGeom<N>::PADN;        // LDS points per transform (padi<E>)
  static constexpr int RPQ = FftGeom<N>::RPQ;
};

template <int N>
__global__ __launch_bounds__(C2rGeom<N>::RPQ* C2rGeom<N>::NT) void rows_c2r_argmax_kernel(
    const cf* __restrict__ Q, long long Q_stride, int Wp, const cf* __restrict__ tw, float* __restrict__ pval,
    int* __restrict__ pidx) {
  using Gm = C2rGeom<N>;
  constexpr int E = Gm::E;
  constexpr int RP = Gm::RPQ, NTH = RP * Gm::NT, LG = Gm::NT < 32 ? Gm::NT : 32;
  static_assert(RP % 2 == 0 && NTH % (LG * RP) == 0, "whole row quads, whole lane groups per pair");
  extern __shared__ cf lds_c[];
  __shared__ float shv[16];
  __shared__ int shi[16];
  const int t = threadIdx.x;
  const int g = t / Gm::NT, j = t % Gm::NT;
  cf* s = lds_c + g * Gm::PADN;
  const int ra = 2 * ((int)blockIdx.x * RP + g), rb = ra + 1;
  {
    // Q[.][v] holds the rows of a pair next to each other: one 16-byte streaming load brings both.  32 consecutive
    // lanes take consecutive bins of one pair (conflict-free LDS stores), the groups of a wave the pairs of a quad,
    // so that a wave reads whole lines.
    const int lg = (t / LG) % RP;
    const int yl = 2 * ((int)blockIdx.x * RP + lg);
    cf* sl = lds_c + lg * Gm::PADN;
    const cf* Qf = Q + (size_t)blockIdx.y * Q_stride;
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    auto put = [&](int k, f32x4 ab) {
      cf a = mkc(ab.x, ab.y), b = mkc(ab.z, ab.w);
      if (k == 0 || k == N / 2) {  // c2r convention: the imaginary parts of the self-conjugate bins are ignored
        a.y = 0.f;
        b.y = 0.f;
      }
      sl[padi<E>(k)] = mkc(a.x - b.y, a.y + b.x);
      if (k > 0 && k < N / 2) sl[padi<E>(N - k)] = mkc(a.x + b.y, b.x - a.y);
    };
    const int k0 = (t % LG) + LG * (t / (LG * RP));   // < NT = N / E: E / 2 rounds cover the bins below Nyquist
    f32x4 ab[E / 2];
#pragma unroll
    for (int i = 0; i < E / 2; ++i)
      ab[i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(Qf + q_at(yl, k0 + i * Gm::NT, Wp)));
#pragma unroll
    for (int i = 0; i < E / 2; ++i) put(k0 + i * Gm::NT, ab[i]);
    if (k0 == 0) put(N / 2, __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(Qf + q_at(yl, N / 2, Wp))));
  }
  __syncthreads();
  cf z[E];
  int zn[E];
  fft_lds_open<N, +1, E>(s, j, tw, z, zn);   // the surface rows never go back to LDS
  float bv = -INFINITY;
  int bi = 0x7fffffff;
  // the thread's candidates in increasing order of index (row ra before rb, positions increasing): a strictly
  // larger value takes over, so ties keep the smaller index as amax_take would, at half its instructions
#pragma unroll
  for (int m = 0; m < E; ++m)
    if (z[m].x > bv) {
      bv = z[m].x;
      bi = ra * N + zn[m];
    }
#pragma unroll
  for (int m = 0; m < E; ++m)
    if (z[m].y > bv) {
      bv = z[m].y;
      bi = rb * N + zn[m];
    }
  block_argmax(bv, bi, shv, shi);
  if (threadIdx.x == 0) {
    pval[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = bv;
    pidx[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = bi;
  }
}

// K4 ---------------------------------------------------------------------------------------
// flat index of the peak -> signed integer shift (wrap at the half size, phase_cross_correlation's rule)
__device__ __forceinline__ int2 coarse_from_index(int bi, int Py, int Px) {
  if (bi == 0x7fffffff) bi = 0;  // all-NaN / empty surface: no shift
  const int py = bi / Px, px = bi - py * Px;
  return make_int2(py > Py / 2 ? py - Py : py, px > Px / 2 ? px - Px : px);
}

// whole-pixel mode only: with upsampling the table kernel of K5 does this reduction itself
__global__ __launch_bounds__(256) void coarse_finish_kernel(const float* __restrict__ pval,
                                                            const int* __restrict__ pidx, int nparts, int Py, int Px,
                                                            int2* __restrict__ coarse, double* __restrict__ shifts,
                                                            int write_shifts) {
  __shared__ float shv[16];
  __shared__ int shi[16];
  float bv = -INFINITY;
  int bi = 0x7fffffff;
  for (int k = threadIdx.x; k < nparts; k += 256)
    amax_take(bv, bi, pval[(size_t)blockIdx.x * nparts + k], pidx[(size_t)blockIdx.x * nparts + k]);
  block_argmax(bv, bi, shv, shi);
  if (threadIdx.x == 0) {
    const int2 cs = coarse_from_index(bi, Py, Px);
    coarse[blockIdx.x] = cs;
    if (write_shifts) {
      shifts[2 * blockIdx.x] = (double)cs.x;
      shifts[2 * blockIdx.x + 1] = (double)cs.y;
    }
  }
}

// K5 ---------------------------------------------------------------------------------------
// Matrix-DFT of the cross-power spectrum R on the 15 x 15 grid of 0.1 px offsets round the coarse
// peak:  V[j][k] = Re sum_{u,v} w_v R[u][v] ey[u][j] ex[v][k],  ey[u][j] = exp(2 pi i fu y'_j / Py),
// y'_j = sy + 0.1 (j - 7), fu the signed frequency of row u (the Nyquist row counts as -Py/2), the
// same along x.  R itself is never stored: the kernels read Q[y][v] = sum_u R[u][v] exp(2 pi i u y / Py),
// which K2 leaves for K3, and use
//   T[j][v] = sum_u R[u][v] ey[u][j] = sum_y Q[y][v] K(y'_j - y),
//   K(t) = 1/Py sum_fu exp(2 pi i fu t / Py) = Ks(t) + exp(-i pi t) / Py,
//   Ks(t) = sin(pi t (Py-1)/Py) / (Py sin(pi t / Py))          (real: the frequencies |fu| < Py/2)
// so a row contributes 15 real weights Ks(y'_j - y) and the term of the Nyquist row,
// cos(pi y'_j) * sum_y (-1)^y Q[y][v] / Py (the cosine, not the exponential: see up_row_sum), rides as a 16th real
// weight.  Along x the +-m symmetry
// halves the work: ex[v][7 +- m] = base_v (a_m +- i b_m), a_m + i b_m = exp(2 pi i fv 0.1 m / Px).
constexpr int UP_N = 15;       // grid points per axis: offsets -0.7 .. 0.7
constexpr int UP_HALF = 7;
constexpr float UP_STEP = 0.1f;
constexpr int UP_UB_MIN = 16;  // fewest spectrum rows a workgroup takes (reached when few frames are in flight)
constexpr int UP_YS = 16;      // floats per row record   {Ks(y'_0 - y) .. Ks(y'_14 - y), (-1)^y / Py}  (one 64-byte scalar load)
constexpr int UP_XS = 20;      // floats per column record {w base.x, w base.y, 0, 0 | 1, a1-1..a7-1 | b1..b7, 0}
constexpr int UP_NV = UP_N * UP_N;  // 225 grid values
constexpr int UP_NQ = 32;      // floats per Nyquist-column partial: the 16 weighted row sums as complex pairs
typedef cf v2f;

__device__ __forceinline__ cf unit_phase(int freq, int ishift, int n, float frac_shift) {
  // exp(2 pi i * freq * (ishift + frac_shift) / n); the integer part is reduced exactly
  int a = (int)(((long long)freq * ishift) % n);
  float rev = (float)a / (float)n + (float)freq * frac_shift / (float)n;
  float sn, cs;
  sincospif(2.0f * rev, &sn, &cs);
  return mkc(cs, sn);
}


// Ks(t) for t = n + f_j (n integer, f_j = 0.1 (j - 7)), in double.  With alpha = pi n / Py,
// gamma_j = pi f_j (1 - 1/Py) and delta_j = pi f_j / Py:
//   Ks = (-1)^n sin(gamma_j - alpha) / (Py sin(alpha + delta_j)),
// so a row costs one sincospi; k = {sin gamma_j, cos gamma_j, sin delta_j, cos delta_j}.
__device__ __forceinline__ void row_weight_consts(int j, int Py, double* k) {
  const double f = 0.1 * (double)(j - UP_HALF);
  sincospi(f * (1.0 - 1.0 / (double)Py), &k[0], &k[1]);
  sincospi(f / (double)Py, &k[2], &k[3]);
}
__device__ __forceinline__ double row_weight(double sa, double ca, bool odd, const double* k, int Py) {
  const double den = sa * k[3] + ca * k[2];
  if (den == 0.0) return (double)(Py - 1) / (double)Py;   // t = 0 (mod Py)
  const double num = k[0] * ca - k[1] * sa;
  return (odd ? -num : num) / ((double)Py * den);
}

// The grid is evaluated as DIFFERENCES from its centre.  The refinement only looks at how the 225 values differ from
// one another (arg-max, then a parabola through three neighbours), and on a broad correlation peak neighbouring values
// differ by 1e-5 .. 1e-6 of their size: summed as they are in float32 they carry about that much rounding, and the
// vertex moved by up to 0.02 px against a float64 evaluation (small windows, broad blobs).  So the row weights of grid
// row j != 7 are K(y'_j - y) - K(y'_7 - y), the column factors of grid column 7 +- m are exp(i phi m) - 1, both
// formed before rounding to float; the sums are then small numbers with full relative precision:
//   U[7][7] = V77,  U[j][7] = Vj7 - V77,  U[7][k] = V7k - V77,  U[j][k] = (Vjk - Vj7) - (V7k - V77)
// and the finish kernel works on D[j][k] = Vjk - V77 = U[j][7] + U[7][k] + U[j][k] in double (V77 itself is not needed).
//
// T'[j] (j = 0..14) from the weighted row sums: acc[j] + e_j acc[15], y'_j = sy + 0.1 (j - 7).  acc[15] is the
// Nyquist row R[Py/2][v], whose frequency counts as -Py/2 (fftfreq's convention, skimage's).  Its
// factor is e_j = cos(pi y'_j) (- cos(pi y'_7) for j != 7), NOT exp(-i pi y'_j): the refinement sums the half
// spectrum v <= Px/2 and doubles, which stands for the pair (u, v), (-u, -v) -- but the Nyquist row is its own
// mirror image, its pair is (Py/2, v), (Py/2, -v), both with the SAME factor exp(-i pi y'), and their sum is
// exp(-i pi y') 2 Re(R E_v): the real part keeps the cosine only.  With the sine in, the grid carried an extra term
// antisymmetric in the offset, proportional to the energy in that one row: nothing on 2048 rows, but on a 64-row
// window of two different channels it moved one vertex in ~600 planes by 0.03 px (round 3; until then put down to
// half-precision spectra).
__device__ __forceinline__ v2f up_row_sum(int j, int sy, const v2f* acc) {
  // cos of pi m / 10, m = 0..7
  constexpr float C[8] = {1.f, 0.95105651629515357f, 0.80901699437494742f, 0.58778525229247313f,
                          0.30901699437494742f, 0.f, -0.30901699437494742f, -0.58778525229247313f};
  const int m = j >= UP_HALF ? j - UP_HALF : UP_HALF - j;
  const float sg = (sy & 1) ? -1.f : 1.f;
  // cos(pi y') = (-1)^sy cos(pi (j-7)/10); off the centre row minus its value there, (-1)^sy
  const float pc = sg * (m ? C[m] - 1.f : 1.f);
  const v2f n = acc[UP_N];
  return (v2f){acc[j].x + n.x * pc, acc[j].y + n.y * pc};
}

// Row and column records of every frame of the batch (the coarse shift enters the row weights
// and the column base).  Every workgroup first reduces K3's per-workgroup maxima to the coarse peak
// (a few hundred values; cheaper than a launch of its own) and the first one of a frame records it.  The threads that build row records also sum the Nyquist column
// v = Px/2, which the main kernel's power-of-two column split leaves out: one partial (16 weighted
// row sums) per workgroup.
__global__ __launch_bounds__(256) void upsample_tables_kernel(int Py, int Px, int Wp, const float* __restrict__ pval,
                                                              const int* __restrict__ pidx, int nparts,
                                                              int2* __restrict__ coarse,
                                                              const cf* __restrict__ Q, long long Q_stride,
                                                              float* __restrict__ yrec, float* __restrict__ xrec,
                                                              float* __restrict__ nyq, int nyb) {
  __shared__ float wsum[4][UP_NQ];
  __shared__ double wk[UP_N][4];
  const int frame = blockIdx.y;
  const int tid = threadIdx.x;
  const int i = blockIdx.x * 256 + tid;
  __shared__ float shv[16];
  __shared__ int shi[16];
  __shared__ int2 peak;
  {
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int k = tid; k < nparts; k += 256)
      amax_take(bv, bi, pval[(size_t)frame * nparts + k], pidx[(size_t)frame * nparts + k]);
    block_argmax(bv, bi, shv, shi);
    if (tid == 0) {
      peak = coarse_from_index(bi, Py, Px);
      if (blockIdx.x == 0) coarse[frame] = peak;
    }
  }
  if (tid < UP_N) row_weight_consts(tid, Py, wk[tid]);
  __syncthreads();
  const int2 cs = peak;
  v2f acc[UP_N + 1];
#pragma unroll
  for (int j = 0; j <= UP_N; ++j) acc[j] = (v2f){0.f, 0.f};
  if (i < Py) {
    const cf q = Q[(size_t)frame * Q_stride + q_at(i, Px / 2, Wp)];
    const int n = cs.x - i;
    double sa, ca;
    sincospi((double)n / (double)Py, &sa, &ca);
    float wgt[UP_YS];
    const double w7 = row_weight(sa, ca, n & 1, wk[UP_HALF], Py);
#pragma unroll
    for (int j = 0; j < UP_N; ++j)
      wgt[j] = j == UP_HALF ? (float)w7 : (float)(row_weight(sa, ca, n & 1, wk[j], Py) - w7);
    wgt[UP_N] = ((i & 1) ? -1.f : 1.f) / (float)Py;
    float4* o = reinterpret_cast<float4*>(yrec + ((size_t)frame * Py + i) * UP_YS);
#pragma unroll
    for (int k = 0; k < UP_YS / 4; ++k) o[k] = make_float4(wgt[4 * k], wgt[4 * k + 1], wgt[4 * k + 2], wgt[4 * k + 3]);
#pragma unroll
    for (int j = 0; j <= UP_N; ++j) acc[j] = (v2f){q.x * wgt[j], q.y * wgt[j]};
  } else if (i - Py < Wp) {
    const int v = i - Py;
    const int fv = v < Px / 2 ? v : v - Px;
    const float w = v > Px / 2 ? 0.f : ((v == 0 || v == Px / 2) ? 1.f : 2.f);  // Hermitian doubling of the half spectrum
    float* o = xrec + ((size_t)frame * Wp + v) * UP_XS;
    const cf bs = unit_phase(fv, cs.y, Px, 0.f);
    o[0] = w * bs.x; o[1] = w * bs.y; o[2] = 0.f; o[3] = 0.f; o[4] = 1.f; o[UP_XS - 1] = 0.f;
#pragma unroll
    for (int m = 1; m <= UP_HALF; ++m) {   // exp(i phi) - 1 = -2 sin^2(phi/2) + i sin(phi), from the half angle
      float sh, ch;
      sincospif((float)fv * ((float)m * UP_STEP) / (float)Px, &sh, &ch);
      o[4 + m] = -2.f * sh * sh; o[11 + m] = 2.f * sh * ch;
    }
  }
  if ((int)blockIdx.x < nyb) {  // uniform: this workgroup holds rows
#pragma unroll
    for (int q = 0; q < UP_NQ; ++q) {
      float x = (q & 1) ? acc[q >> 1].y : acc[q >> 1].x;
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) x += __shfl_down(x, d, 64);
      if ((tid & 63) == 0) wsum[tid >> 6][q] = x;
    }
    __syncthreads();
    if (tid < UP_NQ)
      nyq[((size_t)frame * nyb + blockIdx.x) * UP_NQ + tid] =
          (wsum[0][tid] + wsum[1][tid]) + (wsum[2][tid] + wsum[3][tid]);
  }
}

// grid value (j, k) from the P and Q sums, pq[j * 8 + m] = P[j][m] (m = 0..7) and
// pq[128 + j * 8 + m - 1] = Q[j][m] (m = 1..7):  V[j][7] = P[j][0],  V[j][7 +- m] = P[j][m] -+ Q[j][m]
__device__ __forceinline__ float up_grid_value(const float* pq, int j, int k) {
  const int m = k >= UP_HALF ? k - UP_HALF : UP_HALF - k;
  const float pv = pq[j * 8 + m];
  if (m == 0) return pv;
  const float qv = pq[128 + j * 8 + (m - 1)];
  return k > UP_HALF ? pv - qv : pv + qv;
}

// One workgroup: 512 columns (two per thread, v < Px/2) x ub rows of Q.  Row records are
// wave-uniform and arrive through one 64-byte scalar load per row.  The column sums T[j][v],
// multiplied by the column base, are then contracted over v in LDS: a lane owns a 4 x 4 block of
// the P (Re T x a) or Q (Im T x b) sums for every fourth column of its wave's 64, so that two
// 16-byte LDS reads feed 16 FMAs; lanes, then waves, are summed in fixed order (deterministic).
__global__ __launch_bounds__(256) void upsample_partial_kernel(const cf* __restrict__ R, long long R_stride, int Wp,
                                                               int Py, int Px, int ncg, int ub,
                                                               const int2* __restrict__ coarse,
                                                               const float* __restrict__ yrec,
                                                               const float* __restrict__ xrec,
                                                               float* __restrict__ part, int slots) {
  __shared__ __align__(16) float Tl[256][36];  // [column][plane * 16 + j]: Re, Im of T[j][v] base_v; j = 15 is zero
  __shared__ __align__(16) float Xl[256][16];  // [column]{1, a1..a7, b1..b7, 0}
  __shared__ float red[4][256];
  const int tid = threadIdx.x;
  const int cg = blockIdx.x % ncg, uc = blockIdx.x / ncg;
  const int frame = blockIdx.y;
  const int u0 = uc * ub;
  const int len = min(ub, Py - u0);  // multiple of 4 (host)
  const int half = Px / 2;
  int vcol[2];
  bool ok[2];
#pragma unroll
  for (int c = 0; c < 2; ++c) { vcol[c] = cg * 512 + c * 256 + tid; ok[c] = vcol[c] < half; }
  v2f acc[2][UP_N + 1];   // the 15 weighted row sums and the Nyquist-row sum, per column
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int j = 0; j <= UP_N; ++j) acc[c][j] = (v2f){0.f, 0.f};
  // this thread's columns in the row-interleaved Q: the 4 rows of a block are 32 contiguous bytes
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  const f32x4* Rp[2];
#pragma unroll
  for (int c = 0; c < 2; ++c)
    Rp[c] = reinterpret_cast<const f32x4*>(R + (size_t)frame * R_stride + q_at(u0, ok[c] ? vcol[c] : 0, Wp));   // 4 | u0
  const size_t rblk = (size_t)Wp * 2;   // f32x4 units per 4-row block
  const float* __restrict__ yr = yrec + ((size_t)frame * Py + u0) * UP_YS;
  constexpr int PF = 4;  // rows in flight = one block: the next block's loads are issued before this block's FMAs
  f32x4 r[2][2], rn[2][2];
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    r[c][0] = Rp[c][0];
    r[c][1] = Rp[c][1];
  }
  for (int ul = 0; ul < len; ul += PF) {
    const int un = min(ul + PF, len - PF);
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      rn[c][0] = Rp[c][(size_t)(un >> 2) * rblk];
      rn[c][1] = Rp[c][(size_t)(un >> 2) * rblk + 1];
    }
#pragma unroll
    for (int i = 0; i < PF; ++i) {
      const float* __restrict__ rec = yr + (ul + i) * UP_YS;
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const f32x4 two = r[c][i >> 1];
        const v2f q = (i & 1) ? (v2f){two.z, two.w} : (v2f){two.x, two.y};
#pragma unroll
        for (int j = 0; j <= UP_N; ++j) acc[c][j] = __builtin_elementwise_fma(q, (v2f){rec[j], rec[j]}, acc[c][j]);
      }
    }
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      r[c][0] = rn[c][0];
      r[c][1] = rn[c][1];
    }
  }
  // contraction over columns
  const int sy = coarse[frame].x;
  const int wv = tid >> 6, lane = tid & 63, vsub = lane >> 4, blk = lane & 15;
  const int isq = blk >> 3, j0 = ((blk & 7) >> 1) * 4, m0 = (blk & 1) * 4;
  float sum[4][4];
#pragma unroll
  for (int jj = 0; jj < 4; ++jj)
#pragma unroll
    for (int mm = 0; mm < 4; ++mm) sum[jj][mm] = 0.f;
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    if (cg * 512 + c * 256 >= half) break;  // uniform
    if (c) __syncthreads();                  // the first half's reads are done
    {
      const float4* xr = (const float4*)(xrec + ((size_t)frame * Wp + (ok[c] ? vcol[c] : 0)) * UP_XS);
      const float4 x0 = xr[0];
      const float bx = ok[c] ? x0.x : 0.f, by = ok[c] ? x0.y : 0.f;  // columns past Px/2 contribute zeros
#pragma unroll
      for (int q = 0; q < 4; ++q) *(float4*)&Xl[tid][4 * q] = xr[1 + q];
      float tre[16], tim[16];
#pragma unroll
      for (int j = 0; j < UP_N; ++j) {
        const v2f t = up_row_sum(j, sy, acc[c]);
        tre[j] = fmaf(t.x, bx, -(t.y * by));
        tim[j] = fmaf(t.x, by, t.y * bx);
      }
      tre[15] = tim[15] = 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        *(float4*)&Tl[tid][4 * q] = make_float4(tre[4 * q], tre[4 * q + 1], tre[4 * q + 2], tre[4 * q + 3]);
        *(float4*)&Tl[tid][16 + 4 * q] = make_float4(tim[4 * q], tim[4 * q + 1], tim[4 * q + 2], tim[4 * q + 3]);
      }
    }
    __syncthreads();
#pragma unroll 4
    for (int it = 0; it < 16; ++it) {
      const int vv = wv * 64 + it * 4 + vsub;
      const float4 t = *(const float4*)&Tl[vv][isq * 16 + j0];
      const float4 xc = *(const float4*)&Xl[vv][isq * 8 + m0];
      const float tv[4] = {t.x, t.y, t.z, t.w}, xv[4] = {xc.x, xc.y, xc.z, xc.w};
#pragma unroll
      for (int jj = 0; jj < 4; ++jj)
#pragma unroll
        for (int mm = 0; mm < 4; ++mm) sum[jj][mm] = fmaf(tv[jj], xv[mm], sum[jj][mm]);
    }
  }
#pragma unroll
  for (int jj = 0; jj < 4; ++jj)
#pragma unroll
    for (int mm = 0; mm < 4; ++mm) {
      float x = sum[jj][mm];
      x += __shfl_xor(x, 16, 64);
      x += __shfl_xor(x, 32, 64);
      if (vsub == 0) red[wv][isq * 128 + (j0 + jj) * 8 + m0 + mm] = x;
    }
  __syncthreads();
  red[0][tid] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
  __syncthreads();
  if (tid < UP_NV) {
    const int j = tid / UP_N, k = tid - j * UP_N;
    part[((size_t)frame * slots + blockIdx.x) * UP_NV + tid] = up_grid_value(&red[0][0], j, k);
  }
}

// K6 ---------------------------------------------------------------------------------------
__device__ __forceinline__ double parabola(double vm, double v0, double vp) {
  double den = vm - 2.0 * v0 + vp;
  if (!(den < 0.0)) return 0.0;
  double d = 0.5 * (vm - vp) / den;
  return fmin(0.5, fmax(-0.5, d));
}

__global__ __launch_bounds__(256) void upsample_finish_kernel(const float* __restrict__ part, int nchunks,
                                                              const float* __restrict__ nyq, int nyb,
                                                              const float* __restrict__ xrec, int Wp, int Px,
                                                              const int2* __restrict__ coarse,
                                                              double* __restrict__ shifts) {
  __shared__ float U[UP_N * UP_N];
  __shared__ double V[UP_N * UP_N];   // the grid minus its centre value (see up_row_sum)
  __shared__ float nq[UP_NQ];
  const int frame = blockIdx.x, tid = threadIdx.x;
  if (tid < UP_NQ) {
    float s = 0.f;
    for (int b = 0; b < nyb; ++b) s += nyq[((size_t)frame * nyb + b) * UP_NQ + tid];
    nq[tid] = s;
  }
  __syncthreads();
  if (tid < UP_N * UP_N) {
    // fixed summation order (deterministic); loads are issued 8 at a time to hide their latency
    const float* pp = part + (size_t)frame * nchunks * (UP_N * UP_N) + tid;
    float s = 0.f;
    int c = 0;
    for (; c + 8 <= nchunks; c += 8) {
      float t[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) t[q] = pp[(size_t)(c + q) * (UP_N * UP_N)];
#pragma unroll
      for (int q = 0; q < 8; ++q) s += t[q];
    }
    for (; c < nchunks; ++c) s += pp[(size_t)c * (UP_N * UP_N)];
    {  // the Nyquist column (weight 1): its row sums come from the table kernel; single-term P and Q sums
      const int j = tid / UP_N, k = tid - j * UP_N;
      const float* xr = xrec + ((size_t)frame * Wp + Px / 2) * UP_XS;
      const v2f t = up_row_sum(j, coarse[frame].x, (const v2f*)nq);
      const float cx = fmaf(t.x, xr[0], -(t.y * xr[1])), cy = fmaf(t.x, xr[1], t.y * xr[0]);
      const int m = k >= UP_HALF ? k - UP_HALF : UP_HALF - k;
      const float pv = cx * xr[4 + m], qv = m ? cy * xr[11 + m] : 0.f;
      s += k > UP_HALF ? pv - qv : pv + qv;
    }
    U[tid] = s;
  }
  __syncthreads();
  if (tid < UP_N * UP_N) {
    const int j = tid / UP_N, k = tid - j * UP_N;
    double d = 0.0;
    if (j != UP_HALF) d += (double)U[j * UP_N + UP_HALF];
    if (k != UP_HALF) d += (double)U[UP_HALF * UP_N + k];
    if (j != UP_HALF && k != UP_HALF) d += (double)U[tid];
    V[tid] = d;
  }
  __syncthreads();
  __shared__ float shv[16];
  __shared__ int shi[16];
  float bv = tid < UP_N * UP_N ? (float)V[tid] : -INFINITY;
  int best = tid < UP_N * UP_N ? tid : 0x7fffffff;
  block_argmax(bv, best, shv, shi);   // first maximum in row-major order, as np.argmax
  if (tid == 0) {
    if (best == 0x7fffffff) best = UP_HALF * UP_N + UP_HALF;
    int iy = best / UP_N, ix = best - iy * UP_N;
    int2 cs = coarse[frame];
    double cy = (double)cs.x + (double)(iy - UP_HALF) * 0.1;
    double cx = (double)cs.y + (double)(ix - UP_HALF) * 0.1;
    if (iy > 0 && iy < UP_N - 1)
      cy += 0.1 * parabola(V[(iy - 1) * UP_N + ix], V[best], V[(iy + 1) * UP_N + ix]);
    if (ix > 0 && ix < UP_N - 1)
      cx += 0.1 * parabola(V[iy * UP_N + ix - 1], V[best], V[iy * UP_N + ix + 1]);
    shifts[2 * frame] = cy;
    shifts[2 * frame + 1] = cx;
  }
}

// MIP / plane sums -------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mip_kernel(const uint16_t* __restrict__ in, long long in_stride,
                                                  uint16_t* __restrict__ out, int Z, long long px,
                                                  const int* __restrict__ sub2, long long sub_stride) {
  // sub2: plane z of item i is read as max(v - level(sub2[i * sub_stride + z]), 0)
  const long long item = blockIdx.y;
  const uint16_t* p = in + item * in_stride;
  uint16_t* o = out + item * px;
  // 16-byte accesses need every plane z * px of the item to start aligned: px % 8 == 0; else all scalar
  const long long nv = (px & 7) ? 0 : (px >> 3);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nv;
       i += (long long)gridDim.x * blockDim.x) {
    u16x8 m;
#pragma unroll
    for (int k = 0; k < 8; ++k) m.v[k] = 0;
    for (int z = 0; z < Z; ++z) {
      const u16x8 v = load8(p + z * px + i * 8);
      const int lv = plane_level(sub2, item * sub_stride + z);
#pragma unroll
      for (int k = 0; k < 8; ++k) m.v[k] = (uint16_t)max((int)m.v[k], (int)v.v[k] - lv);
    }
    store8(o + i * 8, m);
  }
  for (long long i = nv * 8 + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < px;
       i += (long long)gridDim.x * blockDim.x) {   // scalar: nothing when px % 8 == 0, the whole plane otherwise
    int m = 0;
    for (int z = 0; z < Z; ++z) m = max(m, (int)p[z * px + i] - plane_level(sub2, item * sub_stride + z));
    o[i] = (uint16_t)m;
  }
}

// sum[i] += sum over n planes (stride apart) of plane[i]; vec: 8 pixels per lane as 16-byte loads, four
// planes in flight
__global__ __launch_bounds__(256) void accum_planes_kernel(const uint16_t* __restrict__ in, long long stride,
                                                           long long n, long long px, uint32_t* __restrict__ sum,
                                                           int vec, const int* __restrict__ sub2, long long sub_stride) {
  // sub2: plane k is read as max(v - level(sub2[k * sub_stride]), 0)
  if (vec) {
    const long long nv = px >> 3;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nv;
         i += (long long)gridDim.x * blockDim.x) {
      uint32_t s[8];
      const uint4 s0 = reinterpret_cast<const uint4*>(sum)[2 * i], s1 = reinterpret_cast<const uint4*>(sum)[2 * i + 1];
      s[0] = s0.x; s[1] = s0.y; s[2] = s0.z; s[3] = s0.w; s[4] = s1.x; s[5] = s1.y; s[6] = s1.z; s[7] = s1.w;
      long long k = 0;
      for (; k + 4 <= n; k += 4) {
        u16x8 v[4];
        int lv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = load8_nt(in + (k + q) * stride + i * 8);
#pragma unroll
        for (int q = 0; q < 4; ++q) lv[q] = plane_level(sub2, (k + q) * sub_stride);
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int e = 0; e < 8; ++e) s[e] += (uint32_t)max((int)v[q].v[e] - lv[q], 0);
      }
      for (; k < n; ++k) {
        const u16x8 v = load8_nt(in + k * stride + i * 8);
        const int lv = plane_level(sub2, k * sub_stride);
#pragma unroll
        for (int e = 0; e < 8; ++e) s[e] += (uint32_t)max((int)v.v[e] - lv, 0);
      }
      reinterpret_cast<uint4*>(sum)[2 * i] = make_uint4(s[0], s[1], s[2], s[3]);
      reinterpret_cast<uint4*>(sum)[2 * i + 1] = make_uint4(s[4], s[5], s[6], s[7]);
    }
    return;
  }
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < px;
       i += (long long)gridDim.x * blockDim.x) {
    uint32_t s = sum[i];
    for (long long k = 0; k < n; ++k) s += (uint32_t)max((int)in[k * stride + i] - plane_level(sub2, k * sub_stride), 0);
    sum[i] = s;
  }
}

// Apply ------------------------------------------------------------------------------------
// out(y,x) = cubic B-spline interpolation of `in` at (y - dy, x - dx): scipy.ndimage.shift(order=3,
// mode='constant', cval=0), i.e. spline coefficients c = (direct B-spline filter)^-1 s with mirror
// boundary, then a 4-tap evaluation.  The inverse filter is the two-sided exponential
// c[k] = c0 * sum_m z1^|m| s[k-m]  (z1 = sqrt(3)-2).  Per axis a thread evaluates runs of 8
// neighbouring outputs: the causal and anti-causal sums are started with R+1 explicit taps
// (truncation z1^(R+1) ~ 1.4e-7) and continued by the exact one-multiply recursions
// C+[k+1] = s[k+1] + z1 C+[k],  C-[k-1] = s[k-1] + z1 C-[k]  -> ~11 FMA per sample instead of a
// 24-tap FIR.  An axis whose shift is an integer bypasses the filter, so integer shifts are exact
// copies.  Source rows are staged through LDS: unaligned 16-byte loads inside the image, mirrored
// scalar loads at its border.
//
// A workgroup owns a strip of 64 output columns and walks down it 64 rows a step.  The x-filtered
// rows the next step needs again (2R+3 = 25) stay in LDS, so every source row is staged and
// x-filtered once per strip.  Both passes work on packed f32 pairs: a thread runs the columns c and
// c+32 together, which halves the instruction count (not the arithmetic time: per SIMD a
// v_pk_fma_f32 costs two v_fma_f32 at 4 waves, scripts/chainbench.hip) and the per-run bookkeeping.
constexpr int AP_R = 11;
constexpr int AP_TX = 64;      // output columns per strip
constexpr int AP_TW = 96;      // staged columns, starting at the first sample the strip needs: 12 u16x8 vectors
constexpr int AP_PITCH = 104;  // u16 per staged row = 52 dwords = 4 * odd: 16-byte reads by lanes on consecutive rows hit distinct banks
constexpr float AP_Z1 = -0.26794919243112270647f;

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef u32x4 u32x4_px __attribute__((aligned(2)));  // 16-byte load at any pixel address

struct AxisParams {
  int sh;      // integer part of the source offset: i0 = i + sh
  int lo, hi;  // output indices whose source coordinate lies inside [0, n-1]
  int whole;   // 1: integer shift on this axis
  float w[4];  // c0 * beta3(frac - j), j = -1..2
};

__device__ __forceinline__ AxisParams axis_params(double d, int n) {
  AxisParams a;
  double nf = floor(-d);
  double frac = -d - nf;
  // |d| beyond the int range: every output is outside; any in-range sh will do
  a.sh = nf > 1.0e9 ? 1000000000 : (nf < -1.0e9 ? -1000000000 : (int)nf);
  a.whole = frac == 0.0;
  double lo = ceil(d), hi = floor((double)(n - 1) + d);   // i - d >= 0 ; i - d <= n - 1
  a.lo = lo < 0.0 ? 0 : (lo > (double)n ? n : (int)lo);
  a.hi = hi > (double)(n - 1) ? n - 1 : (hi < -1.0 ? -1 : (int)hi);
  // cubic B-spline weights beta3(frac - j), j = -1..2, times c0 = -6 z1 / (1 - z1^2) = sqrt(3)
  const float c0 = 1.7320508075688772f / 6.0f;
  float f = (float)frac, g = 1.0f - f;
  a.w[0] = c0 * (g * g * g);
  a.w[1] = c0 * (4.0f - 6.0f * f * f + 3.0f * f * f * f);
  a.w[2] = c0 * (4.0f - 6.0f * g * g + 3.0f * g * g * g);
  a.w[3] = c0 * (f * f * f);
  return a;
}

constexpr int AM_TY = 64;                    // output rows per step
constexpr int AM_HALO = 2 * AP_R + 3;        // x-filtered rows carried from step to step
constexpr int AM_CP = 90;                    // rows per column pair: 180 dwords = 4 * odd -> conflict-free 16-byte reads
constexpr int AM_G = 8;                      // outputs per run (x and y)
static_assert(AM_TY + AM_HALO + 1 <= AM_CP, "a step's rows fit one column of colp");

__device__ __forceinline__ v2f sp_splat(float v) { return (v2f){v, v}; }
__device__ __forceinline__ v2f sp_fma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }

// NOUT outputs of two runs (the halves of a pair) from NOUT + 2*AP_R + 3 consecutive samples each:
// `s(i)` returns sample i, the sample under output k is i = AP_R + 1 + k, coefficients k-1 .. k+2
// feed output k.  Two sweeps, so that few values are live at once: the causal sums upwards (kept),
// then downwards the anti-causal sum, the coefficient and its four contributions to the outputs;
// `mid()` runs between them (the caller fetches what only the second sweep reads).
template <int NOUT, typename F, typename M>
__device__ __forceinline__ void spline_run2(F s, M mid, const float (&w)[4], v2f* out) {
  constexpr int NC = NOUT + 3;           // coefficients q = -1 .. NOUT+1
  const v2f z = sp_splat(AP_Z1);
  const v2f w0 = sp_splat(w[0]), w1 = sp_splat(w[1]), w2 = sp_splat(w[2]), w3 = sp_splat(w[3]);
  v2f cp[NC];
  {
    v2f a = s(0);
#pragma unroll
    for (int m = 1; m <= AP_R; ++m) a = sp_fma(a, z, s(m));
    cp[0] = a;
  }
#pragma unroll
  for (int q = 1; q < NC; ++q) cp[q] = sp_fma(cp[q - 1], z, s(AP_R + q));
  mid();
  v2f cm = s(NC - 1 + 2 * AP_R);         // anti-causal sum of coefficient q + 1 while q is worked on
#pragma unroll
  for (int m = 1; m <= AP_R; ++m) cm = sp_fma(cm, z, s(NC - 1 + 2 * AP_R - m));
#pragma unroll
  for (int q = NC - 1; q >= 0; --q) {
    v2f c;                               // coefficient / c0 = C+ + C- - s, and C-[q] - s[q] = z1 C-[q+1]
    if (q == NC - 1) {
      c = (cp[q] + cm) - s(AP_R + q);
    } else {
      c = sp_fma(cm, z, cp[q]);
      if (q > 0) cm = sp_fma(cm, z, s(AP_R + q));
    }
    if (q - 3 >= 0 && q - 3 < NOUT) out[q - 3] = w3 * c;
    if (q - 2 >= 0 && q - 2 < NOUT) out[q - 2] = sp_fma(w2, c, out[q - 2]);
    if (q - 1 >= 0 && q - 1 < NOUT) out[q - 1] = sp_fma(w1, c, out[q - 1]);
    if (q < NOUT) out[q] = sp_fma(w0, c, out[q]);
  }
}

// float -> u32, truncating toward zero, negatives to 0 (the instruction's own clamp; a C cast leaves them undefined)
__device__ __forceinline__ unsigned cvt_u32_floor0(float v) {
  unsigned r;
  asm("v_cvt_u32_f32 %0, %1" : "=v"(r) : "v"(v));
  return r;
}

// mirror index for the common single reflection, the general form otherwise
__device__ __forceinline__ int reflect_near(int i, int n) {
  int a = i < 0 ? -i : i;
  a = a >= n ? 2 * (n - 1) - a : a;
  if ((unsigned)a >= (unsigned)n) a = reflect101_idx(i, n);
  return a;
}

__global__ __launch_bounds__(256, 4) void apply_shift_kernel(const uint16_t* __restrict__ in,
                                                                        uint16_t* __restrict__ out,
                                                                        long long planes_per_frame, int Y, int X,
                                                                        const double* __restrict__ shifts, int nty,
                                                                        const int* __restrict__ sub2) {
  constexpr int VPR = AP_TW / 8;                 // 16-byte vectors per staged row
  constexpr int LP = AP_PITCH / 8;               // ... per LDS row
  constexpr int NV = AM_TY * VPR / 256;          // vectors a thread stages per step
  static_assert(AM_TY * VPR % 256 == 0, "whole rounds of staging");
  static_assert((AP_TX / 2 / AM_G - 1) * AM_G + 32 + AM_G + 2 * AP_R + 3 <= AP_TW, "x runs stay inside the staged columns");
  __shared__ u32x4 tile[AM_TY * LP];             // the step's new source rows, raw u16
  __shared__ __align__(16) v2f colp[(AP_TX / 2) * AM_CP];   // x-filtered rows: [column pair (c, c+32)][row]
  __shared__ AxisParams pay, pax;
  const int tid = threadIdx.x;
  const long long plane = blockIdx.z;
  const long long frame = plane / planes_per_frame;
  // background level subtracted as the samples are staged (two pixels per saturating packed subtract); 0 without sub2
  const unsigned short lv = (unsigned short)__builtin_amdgcn_readfirstlane(plane_level(sub2, plane));
  auto lower = [&](u32x4 v) {
    typedef unsigned short us8 __attribute__((ext_vector_type(8)));
    return __builtin_bit_cast(u32x4, __builtin_elementwise_sub_sat(__builtin_bit_cast(us8, v), (us8)(lv)));
  };
  if (tid < 2) {
    AxisParams a = axis_params(shifts[2 * frame + tid], tid == 0 ? Y : X);
    if (tid == 0) pay = a;
    else pax = a;
  }
  __syncthreads();
  // wave-uniform: keep the parameters in scalar registers
  auto uniform = [](const AxisParams& a) {
    AxisParams u;
    u.sh = __builtin_amdgcn_readfirstlane(a.sh);
    u.lo = __builtin_amdgcn_readfirstlane(a.lo);
    u.hi = __builtin_amdgcn_readfirstlane(a.hi);
    u.whole = __builtin_amdgcn_readfirstlane(a.whole);
#pragma unroll
    for (int i = 0; i < 4; ++i) u.w[i] = a.w[i];   // the weights stay in vector registers (scalar ones are short here)
    return u;
  };
  const AxisParams ay = uniform(pay), ax = uniform(pax);
  // strips next to each other (consecutive ids) run on one XCD: the 32 columns they both stage come from one L2
  const unsigned lin = xcd_remap(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
  const int ox = (lin % gridDim.x) * AP_TX;
  const int tiles_y = (Y + AM_TY - 1) / AM_TY;
  const int ty0 = (lin / gridDim.x) * nty, ty1 = min(ty0 + nty, tiles_y);
  const uint16_t* __restrict__ p = in + (size_t)plane * Y * X;
  uint16_t* __restrict__ po = out + (size_t)plane * Y * X;
  const int xend = min(ox + AP_TX, X), yend = min(ty1 * AM_TY, Y);
  if (ox > ax.hi || xend - 1 < ax.lo || ty0 * AM_TY > ay.hi || yend - 1 < ay.lo) {
    // no output of this workgroup has its source inside the image
    const int w = xend - ox, h = yend - ty0 * AM_TY;
    for (int i = tid; i < w * h; i += 256) po[(size_t)(ty0 * AM_TY + i / w) * X + ox + i % w] = 0;
    return;
  }
  const int first = ox + ax.sh - 1 - AP_R;       // source column of staged column 0
  const bool wide = X >= 8;                      // a 16-byte load fits in a row
  const bool cols_inside = wide && first >= 0 && first + AP_TW <= X;

  // this thread's staged vectors: (row in the step, vector) never change
  int cl[NV], col0[NV], lds_off[NV], rr[NV];
#pragma unroll
  for (int jv = 0; jv < NV; ++jv) {
    const int i = tid + jv * 256;
    const int r = i / VPR, v = i - r * VPR;
    rr[jv] = r;
    col0[jv] = first + 8 * v;
    cl[jv] = min(max(col0[jv], 0), X - 8);
    lds_off[jv] = r * LP + v;
  }
  u32x4 q[NV];  // kept opaque so that the loads stay in flight until the rows are staged
  auto row_off = [&](int yy, bool inside) -> size_t {
    if (!inside) yy = reflect_near(yy, Y);
    return (size_t)yy * X;
  };
  auto border_vec = [&](size_t ro, int c0) {     // image border: mirrored scalar loads
    unsigned e[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) e[k] = p[ro + reflect_near(c0 + k, X)];
    return (u32x4){e[0] | (e[1] << 16), e[2] | (e[3] << 16), e[4] | (e[5] << 16), e[6] | (e[7] << 16)};
  };
  auto preload = [&](int src0) {                 // AM_TY rows from source row src0 on
    const bool inside = src0 >= 0 && src0 + AM_TY <= Y;
#pragma unroll
    for (int jv = 0; jv < NV; ++jv)
      q[jv] = *reinterpret_cast<const u32x4_px*>(p + row_off(src0 + rr[jv], inside) + cl[jv]);
  };
  auto stage = [&](int src0) {
    if (cols_inside) {
#pragma unroll
      for (int jv = 0; jv < NV; ++jv) tile[lds_off[jv]] = lower(q[jv]);
      return;
    }
    const bool inside = src0 >= 0 && src0 + AM_TY <= Y;
    // the mirrored columns are the same at every step; worked out here each time (an opaque zero keeps
    // them from being hoisted), they do not hold registers through the interior strips' loop
    int zero;
    asm volatile("v_mov_b32 %0, 0" : "=v"(zero));
#pragma unroll
    for (int jv = 0; jv < NV; ++jv) {
      u32x4 w8 = q[jv];
      const int c0 = col0[jv] + zero;
      if (!(wide && c0 >= 0 && c0 + 8 <= X)) w8 = border_vec(row_off(src0 + rr[jv], inside), c0);
      tile[lds_off[jv]] = lower(w8);
    }
  };
  // x pass over `nrows` staged rows -> colp rows jbase ..: one wave per run pair (m, m+4) when nrows = 64
  auto xpass = [&](int nrows, int jbase) {
    for (int i = tid; i < nrows * (AP_TX / 2 / AM_G); i += 256) {
      const int m = i / nrows, r = i - m * nrows;
      const u32x4* t = tile + r * LP + m;
      u32x4 ra[5], rb[5];   // samples 0..21 (vectors 0..2) feed the upward sweep, 11..32 the downward one
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        ra[k] = t[k];
        rb[k] = t[k + 4];
      }
      auto rest = [&] {
#pragma unroll
        for (int k = 3; k < 5; ++k) {
          ra[k] = t[k];
          rb[k] = t[k + 4];
        }
      };
      auto smp = [&](int i) {            // sample i of both runs: u16 -> f32
        const unsigned a = ra[i >> 3][(i >> 1) & 3], b = rb[i >> 3][(i >> 1) & 3];
        return (i & 1) ? (v2f){(float)(a >> 16), (float)(b >> 16)} : (v2f){(float)(a & 0xffffu), (float)(b & 0xffffu)};
      };
      v2f res[AM_G];
      if (ax.whole) {
#pragma unroll
        for (int k = 0; k < AM_G; ++k) res[k] = smp(AP_R + 1 + k);
      } else {
        spline_run2<AM_G>(smp, rest, ax.w, res);
      }
      // colp row p holds the NEIGHBOURING columns (2p, 2p+1): the y pass then ends in one 4-byte store per lane and
      // row.  Run m's eight outputs are the pairs 4m .. 4m+3, run m+4's the pairs 16 + 4m ..
      v2f* d = colp + (AM_G / 2 * m) * AM_CP + jbase + r;
#pragma unroll
      for (int q = 0; q < AM_G / 2; ++q) {
        d[q * AM_CP] = (v2f){res[2 * q].x, res[2 * q + 1].x};
        d[(AP_TX / 4 + q) * AM_CP] = (v2f){res[2 * q].y, res[2 * q + 1].y};
      }
    }
  };

  // prologue: the 25 rows above the first step's new rows, staged and x-filtered on their own
  {
    const int src0 = ty0 * AM_TY + ay.sh - 1 - AP_R;
    for (int i = tid; i < AM_HALO * VPR; i += 256) {
      const int r = i / VPR, v = i - r * VPR;
      const size_t ro = row_off(src0 + r, false);
      const int c0 = first + 8 * v;
      u32x4 w8;
      if (wide && c0 >= 0 && c0 + 8 <= X) w8 = *reinterpret_cast<const u32x4_px*>(p + ro + c0);
      else w8 = border_vec(ro, c0);
      tile[r * LP + v] = lower(w8);
    }
    if (wide) preload(src0 + AM_HALO);
    __syncthreads();
    xpass(AM_HALO, 0);
    __syncthreads();
    stage(src0 + AM_HALO);
    if (ty0 + 1 < ty1 && wide) preload(src0 + AM_HALO + AM_TY);
  }

  for (int ty = ty0; ty < ty1; ++ty) {
    const int oy = ty * AM_TY;
    const int nsrc = oy + AM_TY + ay.sh + AP_R + 2;   // first new source row of the next step
    const bool more = ty + 1 < ty1;
    __syncthreads();   // A: staged rows and the carried rows are visible; colp rows 25.. are free
    xpass(AM_TY, AM_HALO);
    __syncthreads();   // B: colp is complete, the tile is free
    if (more) stage(nsrc);
    // the loads of the step after that fly during this y pass and the next x pass; they are older than
    // the stores below, so waiting for them (vmcnt counts in order) does not wait for a store
    if (ty + 2 < ty1 && wide) preload(nsrc + AM_TY);
    // ---- y pass: a column pair per lane, 8 rows per thread; stores go straight to the plane
    {
      const int pr = tid & 31, y0 = (tid >> 5) * AM_G;
      const float4* t = reinterpret_cast<const float4*>(colp + pr * AM_CP + y0);
      float4 tv[17];   // rows 0..21 feed the upward sweep, rows 11..32 the downward one
#pragma unroll
      for (int k = 0; k < 11; ++k) tv[k] = t[k];
      auto rest = [&] {
#pragma unroll
        for (int k = 11; k < 17; ++k) tv[k] = t[k];
      };
      auto smp = [&](int i) { return (i & 1) ? (v2f){tv[i >> 1].z, tv[i >> 1].w} : (v2f){tv[i >> 1].x, tv[i >> 1].y}; };
      v2f res[AM_G];
      if (ay.whole) {
#pragma unroll
        for (int k = 0; k < AM_G; ++k) res[k] = smp(AP_R + 1 + k);
      } else {
        spline_run2<AM_G>(smp, rest, ay.w, res);
      }
      const int xa = ox + 2 * pr, xb = xa + 1;
      const bool pair_store = (X & 1) == 0 && (reinterpret_cast<size_t>(out) & 3) == 0;   // both pixels of a lane in one aligned dword
      const bool interior = ox >= ax.lo && ox + AP_TX - 1 <= ax.hi && oy >= ay.lo && oy + AM_TY - 1 <= ay.hi &&
                            ox + AP_TX <= X && oy + AM_TY <= Y && pair_store;
      if (interior) {   // uniform: every output has its source inside the image and lies inside the plane
        // a row's address = a wave-uniform base (scalar arithmetic) + this lane's fixed 32-bit offset
        const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
        const size_t rowb = ((size_t)(oy + 2 * AM_G * wv) * X + ox) * 2;
        const unsigned lane_off = (unsigned)(((tid >> 5) & 1) * AM_G * X + 2 * pr) * 2u;
#pragma unroll
        for (int k = 0; k < AM_G; ++k) {   // truncation toward zero and the clamp to 0 .. 65535 ride in the two conversions
          // (v_cvt_u32_f32 sends negatives to 0, v_cvt_pk_u16_u32 saturates at 65535); written once, not re-read here
          typedef unsigned short us2 __attribute__((ext_vector_type(2)));
          const us2 ab = __builtin_amdgcn_cvt_pk_u16(cvt_u32_floor0(res[k].x), cvt_u32_floor0(res[k].y));
          typedef __attribute__((address_space(1))) char gchar;
          typedef __attribute__((address_space(1))) unsigned gu32;
          gchar* rk = (gchar*)(reinterpret_cast<char*>(po) + rowb + (size_t)k * X * 2);
          asm("" : "+s"(rk));   // stays a scalar base: the store then takes base + 32-bit lane offset, no vector address arithmetic
          unsigned lo = lane_off;
          asm("" : "+v"(lo));   // (kept 32 bits wide up to the store)
          __builtin_nontemporal_store(__builtin_bit_cast(unsigned, ab), (gu32*)(rk + lo));
        }
      } else {
        const bool ain = xa >= ax.lo && xa <= ax.hi, bin = xb >= ax.lo && xb <= ax.hi;
#pragma unroll
        for (int k = 0; k < AM_G; ++k) {
          const int yy = oy + y0 + k;
          const bool yin = yy >= ay.lo && yy <= ay.hi;
          if (yy < Y) {
            const float va = ain && yin ? fminf(fmaxf(res[k].x, 0.f), 65535.f) : 0.f;
            const float vb = bin && yin ? fminf(fmaxf(res[k].y, 0.f), 65535.f) : 0.f;
            if (pair_store) {
              if (xa < X) __builtin_nontemporal_store((unsigned)va | ((unsigned)vb << 16), reinterpret_cast<unsigned*>(po + (size_t)yy * X + xa));
            } else {
              if (xa < X) __builtin_nontemporal_store((uint16_t)(unsigned)va, po + (size_t)yy * X + xa);
              if (xb < X) __builtin_nontemporal_store((uint16_t)(unsigned)vb, po + (size_t)yy * X + xb);
            }
          }
        }
      }
    }
    if (more) {
      __syncthreads();   // C: the y pass has read colp; carry rows 64..89 up to rows 0..25
      for (int i = tid; i < (AP_TX / 2) * 13; i += 256) {
        const int pr = i / 13, k = i - pr * 13;
        float4* c = reinterpret_cast<float4*>(colp + pr * AM_CP);
        c[k] = c[AM_TY / 2 + k];
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------

static int fft_window(int n) {
  int p = 1;
  while (p * 2 <= n) p *= 2;
  return p > 4096 ? 4096 : p;
}

struct RegPlan {
  int Y, X, Py, Px, oy, ox, Wp;
  size_t spec;      // complex elements per spectrum
  int row_blocks;   // workgroups of the inverse row kernel per frame (= arg-max partials)
  int col_blocks;
  int ncg, nuc;     // upsample chunks: column groups x row chunks
  int fb;           // bytes per value of the spectra F and G: 4 (half pairs), 8 (float pairs: windows below 256 points)
};

// row pairs per workgroup of the inverse row kernel (FftGeom<N>::RPQ)
static int geom_rows_RPQ(int N) {
  int nt = N / 8;
  int g = 256 / nt;
  return g > 2 ? g : 2;
}
static int geom_TC(int N) { return col_tile(N); }

static bool make_plan(int Y, int X, RegPlan& p) {
  p.Y = Y;
  p.X = X;
  p.Py = fft_window(Y);
  p.Px = fft_window(X);
  if (p.Py < 64 || p.Px < 64) return false;
  p.oy = (Y - p.Py) / 2;
  p.ox = (X - p.Px) / 2;
  p.Wp = (p.Px / 2 + 1 + 15) / 16 * 16;
  p.spec = (size_t)p.Py * p.Wp;
  p.fb = (p.Py < 256 || p.Px < 256) ? 8 : 4;
  p.row_blocks = p.Py / 2 / geom_rows_RPQ(p.Px);
  p.col_blocks = p.Wp / geom_TC(p.Py);
  p.ncg = (p.Px / 2 + 511) / 512;
  p.nuc = std::max(1, p.Py / UP_UB_MIN);  // most row chunks a launch may use (sizes the partials)
  return true;
}

struct RegWs {
  cf *tw_x, *tw_y;
  char* refspec;   // the reference's full spectrum: p.fb bytes per value, in the column kernel's order (cols_ref_kernel)
  float* gscale;   // ... and the factor per tile that undoes its scale
  char* F;         // row spectra of the batch's moving planes, p.fb bytes per value
  cf* Q;           // float: the batch's Q[y][v]; its slots also take the float row spectra on their way to cols_ref
  float* pval;
  int* pidx;
  int2* coarse;
  float* upart;
  float *yrec, *xrec;  // upsample row / column records per frame
  float* nyq;          // Nyquist-column partial sums per frame
  uint16_t *mip, *mip_ref;
  uint32_t* sum;
  int batch;
};

static size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

static size_t reg_fixed_bytes(const RegPlan& p) {
  return align256(4096 * 8) * 2 + align256(p.spec * p.fb) + align256((size_t)p.col_blocks * 4) +
         align256((size_t)p.Y * p.X * 2) + align256((size_t)p.Y * p.X * 4) + 1024;
}
static size_t reg_frame_bytes(const RegPlan& p, int Z) {
  size_t b = align256(p.spec * p.fb) + align256(p.spec * 8) + 2 * align256((size_t)p.row_blocks * 4) + 256 +
             align256((size_t)p.ncg * p.nuc * UP_NV * 4) + align256((size_t)p.Py * UP_YS * 4) +
             align256((size_t)p.Wp * UP_XS * 4) + align256((size_t)((p.Py + 255) / 256) * UP_NQ * 4);
  if (Z > 1) b += align256((size_t)p.Y * p.X * 2);
  return b;
}

static bool carve(void* ws, size_t bytes, const RegPlan& p, int Z, RegWs& w) {
  size_t fixed = reg_fixed_bytes(p), per = reg_frame_bytes(p, Z);
  if (bytes < fixed + per) return false;
  int batch = (int)((bytes - fixed) / per);
  if (batch > 4096) batch = 4096;
  w.batch = batch;
  char* c = (char*)(((uintptr_t)ws + 255) & ~(uintptr_t)255);
  auto take = [&](size_t n) {
    char* r = c;
    c += align256(n);
    return r;
  };
  w.tw_x = (cf*)take(4096 * 8);
  w.tw_y = (cf*)take(4096 * 8);
  w.refspec = take(p.spec * p.fb);
  w.gscale = (float*)take((size_t)p.col_blocks * 4);
  w.mip_ref = (uint16_t*)take((size_t)p.Y * p.X * 2);
  w.sum = (uint32_t*)take((size_t)p.Y * p.X * 4);
  w.F = take((size_t)batch * align256(p.spec * p.fb));
  w.Q = (cf*)take((size_t)batch * align256(p.spec * 8));
  w.pval = (float*)take((size_t)batch * p.row_blocks * 4);
  w.pidx = (int*)take((size_t)batch * p.row_blocks * 4);
  w.coarse = (int2*)take((size_t)batch * 8);
  w.upart = (float*)take((size_t)batch * p.ncg * p.nuc * UP_NV * 4);
  w.nyq = (float*)take((size_t)batch * ((p.Py + 255) / 256) * UP_NQ * 4);
  w.yrec = (float*)take((size_t)batch * p.Py * UP_YS * 4);
  w.xrec = (float*)take((size_t)batch * p.Wp * UP_XS * 4);
  w.mip = Z > 1 ? (uint16_t*)take((size_t)batch * p.Y * p.X * 2) : nullptr;
  return true;
}

#define IAP_FOR_N(N_, BODY)          \
  switch (N_) {                      \
    case 64: { constexpr int N = 64; BODY; } break;      \
    case 128: { constexpr int N = 128; BODY; } break;    \
    case 256: { constexpr int N = 256; BODY; } break;    \
    case 512: { constexpr int N = 512; BODY; } break;    \
    case 1024: { constexpr int N = 1024; BODY; } break;  \
    case 2048: { constexpr int N = 2048; BODY; } break;  \
    case 4096: { constexpr int N = 4096; BODY; } break;  \
    default: break;                  \
  }

template <typename K>
static hipError_t allow_lds(K kernel, size_t bytes) {
  if (bytes <= 48 * 1024) return hipSuccess;
  return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                             (int)bytes);
}

// Levels subtracted on load (plane_level): plane i of a launch takes sub2[i * stride]; none by default.
struct PlaneSub {
  const int* sub2 = nullptr;
  long long stride = 0;
};

// half16: the row spectra go out as packed half pairs (moving planes; F_stride in half pairs), else as float
// (the reference plane on its way to cols_ref_kernel; F_stride in float pairs)
#define IAP_R2C16(VEC, F32)                                                                                          \
  do {                                                                                                               \
    IAP_HIP(allow_lds((rows_r2c16_kernel<N, VEC, F32>), lds));                                                       \
    hipLaunchKernelGGL((rows_r2c16_kernel<N, VEC, F32>), dim3(p.Py / Rg::ROWS, n), dim3(Rg::NTH), lds, st,           \
                       (const uint16_t*)src, plane_stride, p.X, p.oy, p.ox, (typename Sp<F32>::E*)F, F_stride, p.Wp, \
                       tw_x, sub.sub2, sub.stride);                                                                  \
  } while (0)
static int launch_rows_r2c(const RegPlan& p, const void* src, int kind, long long plane_stride, float in_scale,
                           void* F, long long F_stride, bool half16, int n, const cf* tw_x, hipStream_t st,
                           PlaneSub sub = {}) {
  const int vec = kind == 0 && aligned16(src) && p.X % 8 == 0 && p.ox % 8 == 0 && plane_stride % 8 == 0;
  IAP_FOR_N(p.Px, {
    using Gm = FftGeom<N>;
    ProfScope ps_(K_ROWS_R2C, st);
    if (half16) {
      static_assert(FftGeom<N>::RPQ == (256 / (N / 8) > 2 ? 256 / (N / 8) : 2), "geom_rows_RPQ");
      using Rg = R16Geom<N>;
      if (kind != 0 || in_scale != 1.0f) return IAP_E_ARG;   // the moving planes are u16
      const size_t lds = (size_t)Rg::PS * Rg::PADN * sizeof(cf);
      if (p.fb == 8) {
        if (vec) IAP_R2C16(true, true); else IAP_R2C16(false, true);
      } else {
        if (vec) IAP_R2C16(true, false); else IAP_R2C16(false, false);
      }
    } else {
      const size_t lds = (size_t)Gm::G * Gm::PADN * sizeof(cf);
      IAP_HIP(allow_lds(rows_r2c_kernel<N>, lds));
      hipLaunchKernelGGL(rows_r2c_kernel<N>, dim3(p.Py / 2 / Gm::G, n), dim3(Gm::G * Gm::NT), lds, st, src, kind,
                         plane_stride, p.X, p.oy, p.ox, in_scale, (cf*)F, F_stride, p.Wp, tw_x, sub.sub2, sub.stride);
    }
  });
  IAP_LAUNCH_CHECK();
  return IAP_OK;
}

static long long spec16_stride(const RegPlan& p) { return (long long)(align256(p.spec * p.fb) / p.fb); }   // values of F / G
static long long spec32_stride(const RegPlan& p) { return (long long)(align256(p.spec * 8) / 8); }   // float pairs
static int valid_tiles(const RegPlan& p) { return (p.Px / 2 + 1 + geom_TC(p.Py) - 1) / geom_TC(p.Py); }

// n references: float row spectra F -> column FFT -> G (half pairs, tile-major chunks) + gscale per tile
static int launch_cols_ref(const RegPlan& p, const cf* F, long long F_stride, void* G, long long G_stride, float* gscale,
                           int n, const cf* tw_y, hipStream_t st) {
  IAP_FOR_N(p.Py, {
    using Gm = FftGeom<N>;
    size_t lds = (size_t)Gm::TC * Gm::CS * sizeof(cf);
    ProfScope ps_(K_COLS, st);
    if (p.fb == 8) {
      IAP_HIP(allow_lds((cols_ref_kernel<N, true>), lds));
      hipLaunchKernelGGL((cols_ref_kernel<N, true>), dim3(valid_tiles(p), n), dim3(Gm::TC * Gm::CNT), lds, st, F, F_stride,
                         (cf*)G, G_stride, gscale, p.col_blocks, p.Wp, p.Px / 2 + 1, tw_y);
    } else {
      IAP_HIP(allow_lds((cols_ref_kernel<N, false>), lds));
      hipLaunchKernelGGL((cols_ref_kernel<N, false>), dim3(valid_tiles(p), n), dim3(Gm::TC * Gm::CNT), lds, st, F, F_stride,
                         (h2*)G, G_stride, gscale, p.col_blocks, p.Wp, p.Px / 2 + 1, tw_y);
    }
  });
  IAP_LAUNCH_CHECK();
  return IAP_OK;
}

// n moving frames (half row spectra F) against G (one for all: G_stride 0, or one per frame) -> Q (float)
// workgroups of cols_xpower_kernel<N> that are resident at once on this device (0 on error); one cache per plane
// size and device, the queries are slow
template <int N, bool F32>
static int cols_resident(int threads, size_t lds) {
  static std::atomic<int> known[64];
  int per_cu = 0, dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  if (dev >= 0 && dev < 64 && known[dev].load() > 0) return known[dev].load();
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, cols_xpower_kernel<N, F32>, threads, lds) != hipSuccess) return 0;
  if (dev >= 0 && dev < 64) known[dev].store(per_cu * cus);
  return per_cu * cus;
}

#define IAP_XPOWER(F32)                                                                                                \
  do {                                                                                                                 \
    typedef typename Sp<F32>::E FE;                                                                                    \
    IAP_HIP(allow_lds((cols_xpower_kernel<N, F32>), lds));                                                             \
    const int resident = cols_resident<N, F32>(Gm::TC * Gm::CNT, lds);                                                 \
    if (resident < 1) return (int)hipErrorUnknown;                                                                     \
    hipLaunchKernelGGL((cols_xpower_kernel<N, F32>), dim3((unsigned)std::min<long long>(items, resident)),             \
                       dim3(Gm::TC * Gm::CNT), lds, st, (const FE*)F, spec16_stride(p), (const FE*)G, G_stride, gscale, \
                       p.col_blocks, Q, spec32_stride(p), p.Wp, p.Px / 2 + 1, tw_y, normalize, scale, valid_tiles(p),  \
                       (int)items);                                                                                    \
  } while (0)
static int launch_cols_xpower(const RegPlan& p, const void* F, const void* G, long long G_stride, const float* gscale,
                              cf* Q, int n, const cf* tw_y, int normalize, hipStream_t st) {
  // F holds F_true * 0.5 / Px (fs16), G holds G_true / gscale: P = G conj(F) / (Py Px) needs 2 Px / (Py Px)
  const float scale = 2.0f / (float)p.Py;
  IAP_FOR_N(p.Py, {
    using Gm = FftGeom<N>;
    size_t lds = (size_t)Gm::TC * Gm::CS * sizeof(cf);
    ProfScope ps_(K_COLS, st);
    const long long items = (long long)valid_tiles(p) * n;
    if (items > 0x7fffffffLL) return IAP_E_UNSUPPORTED;
    if (p.fb == 8) IAP_XPOWER(true); else IAP_XPOWER(false);
  });
  IAP_LAUNCH_CHECK();
  return IAP_OK;
}

// n moving frames (half row spectra F) against half ROW spectra of their references (G_stride 0: one for all)
static int launch_cols_fused(const RegPlan& p, const void* F, const void* Grows, long long G_stride, cf* Q, int n,
                             const cf* tw_y, int normalize, hipStream_t st) {
  const float scale = 4.0f * (float)p.Px / (float)p.Py;   // both factors carry fs16 = 0.5 / Px
  IAP_FOR_N(p.Py, {
    using Gm = FftGeom<N>;
    size_t lds = (size_t)Gm::TC * Gm::CS * sizeof(cf) * 2;
    ProfScope ps_(K_COLS, st);
    if (p.fb == 8) {
      IAP_HIP(allow_lds((cols_fused_kernel<N, true>), lds));
      hipLaunchKernelGGL((cols_fused_kernel<N, true>), dim3(valid_tiles(p), n), dim3(Gm::TC * Gm::CNT), lds, st, (const cf*)F,
                         spec16_stride(p), (const cf*)Grows, G_stride, Q, spec32_stride(p), p.Wp, p.Px / 2 + 1, tw_y, normalize, scale);
    } else {
      IAP_HIP(allow_lds((cols_fused_kernel<N, false>), lds));
      hipLaunchKernelGGL((cols_fused_kernel<N, false>), dim3(valid_tiles(p), n), dim3(Gm::TC * Gm::CNT), lds, st, (const h2*)F,
                         spec16_stride(p), (const h2*)Grows, G_stride, Q, spec32_stride(p), p.Wp, p.Px / 2 + 1, tw_y, normalize, scale);
    }
  });
  IAP_LAUNCH_CHECK();
  return IAP_OK;
}

static int launch_rows_c2r(const RegPlan& p, const cf* Q, long long Q_stride, int n, const cf* tw_x, float* pval,
                           int* pidx, hipStream_t st) {
  IAP_FOR_N(p.Px, {
    using Cg = C2rGeom<N>;
    size_t lds = (size_t)Cg::RPQ * Cg::PADN * sizeof(cf);
    IAP_HIP(allow_lds(rows_c2r_argmax_kernel<N>, lds));
    { ProfScope ps_(K_ROWS_C2R, st); hipLaunchKernelGGL(rows_c2r_argmax_kernel<N>, dim3(p.row_blocks, n), dim3(Cg::RPQ * Cg::NT), lds, st, Q, Q_stride,
                       p.Wp, tw_x, pval, pidx); }
  });
  IAP_LAUNCH_CHECK();
  return IAP_OK;
}

static int reg_init_tables(const RegPlan& p, RegWs& w, hipStream_t st) {
  { ProfScope ps_(K_OTHER, st); hipLaunchKernelGGL(twiddle_kernel, dim3((p.Px + 255) / 256), dim3(256), 0, st, w.tw_x, p.Px); }
  { ProfScope ps_(K_OTHER, st); hipLaunchKernelGGL(twiddle_kernel, dim3((p.Py + 255) / 256), dim3(256), 0, st, w.tw_y, p.Py); }
  IAP_LAUNCH_CHECK();
  return IAP_OK;
}

// full 2-D spectrum of the reference plane into w.refspec / w.gscale; the float row pass goes through the
// first frame slot of w.Q, which is free between batches
static int reg_ref_spectrum(const RegPlan& p, RegWs& w, const void* ref, int kind, float in_scale, hipStream_t st,
                            PlaneSub sub = {}) {
  int rc = launch_rows_r2c(p, ref, kind, 0, in_scale, w.Q, 0, false, 1, w.tw_x, st, sub);
  if (rc) return rc;
  return launch_cols_ref(p, w.Q, 0, w.refspec, 0, w.gscale, 1, w.tw_y, st);
}

// The reference of an estimate: w.refspec (default), one spectrum per frame (gref/gscale, channel-shift pairs), or
// the half ROW spectrum of one reference plane (ref_rows: the serial 'previous' chain, cols_fused_kernel).
struct RegRef {
  const void* gref = nullptr;
  long long g_stride = 0;
  const float* gscale = nullptr;
  const void* ref_rows = nullptr;
};

// shifts for n moving planes (n <= w.batch)
static int reg_estimate_batch(const RegPlan& p, RegWs& w, const uint16_t* mov, long long mov_stride, int n, int flags,
                              double* shifts, hipStream_t st, RegRef ref = {}, const void* rows = nullptr,
                              PlaneSub msub = {}) {
  // rows != nullptr: the n row-pass spectra are already there (slots of w.F filled ahead of a serial loop)
  const long long fs = spec32_stride(p);
  const bool up = flags & IAP_REG_UPSAMPLE;
  const int normalize = (flags & IAP_REG_NORMALIZE) ? 1 : 0;
  const void* F16 = rows ? rows : w.F;
  int rc = rows ? IAP_OK : launch_rows_r2c(p, mov, 0, mov_stride, 1.0f, w.F, spec16_stride(p), true, n, w.tw_x, st, msub);
  if (rc) return rc;
  if (ref.ref_rows) rc = launch_cols_fused(p, F16, ref.ref_rows, 0, w.Q, n, w.tw_y, normalize, st);
  else if (ref.gref) rc = launch_cols_xpower(p, F16, ref.gref, ref.g_stride, ref.gscale, w.Q, n, w.tw_y, normalize, st);
  else rc = launch_cols_xpower(p, F16, w.refspec, 0, w.gscale, w.Q, n, w.tw_y, normalize, st);
  if (rc) return rc;
  const cf* F = w.Q;
  rc = launch_rows_c2r(p, F, fs, n, w.tw_x, w.pval, w.pidx, st);
  if (rc) return rc;
  if (!up) {
    ProfScope ps_(K_COARSE, st);
    hipLaunchKernelGGL(coarse_finish_kernel, dim3(n), dim3(256), 0, st, (const float*)w.pval, (const int*)w.pidx,
                       p.row_blocks, p.Py, p.Px, w.coarse, shifts, 1);
  }
  if (up) {
    // Row chunks of a size that depends on the plane alone: an 8th of the rows from 2048 rows on (64 frames x 2 column
    // groups x 8 chunks = 1024 workgroups, measured 207 us against 200 for 4 chunks and 215-250 for 6, 10, 12, 16),
    // a 16th below.  The partial sums are float: were the chunking sized by the number of frames in the launch, a
    // frame's shift would depend (at the 1e-4 px level) on how many others were estimated with it -- on the workspace
    // the caller happened to pass, or on how a stack is split over ranks.
    const int parts = p.Py >= 2048 ? 8 : 16;
    const int ub = std::max(UP_UB_MIN, ((p.Py + parts - 1) / parts + 3) & ~3);
    const int nuc = (p.Py + ub - 1) / ub;
    const int nch = p.ncg * nuc, nyb = (p.Py + 255) / 256;
    { ProfScope ps_(K_OTHER, st);
      hipLaunchKernelGGL(upsample_tables_kernel, dim3((p.Py + p.Wp + 255) / 256, n), dim3(256), 0, st, p.Py, p.Px, p.Wp,
                         (const float*)w.pval, (const int*)w.pidx, p.row_blocks, w.coarse, (const cf*)F, fs, w.yrec,
                         w.xrec, w.nyq, nyb); }
    { ProfScope ps_(K_UPSAMPLE, st);
      hipLaunchKernelGGL(upsample_partial_kernel, dim3(nch, n), dim3(256), 0, st, (const cf*)F, fs, p.Wp, p.Py, p.Px,
                         p.ncg, ub, (const int2*)w.coarse, (const float*)w.yrec, (const float*)w.xrec, w.upart, nch); }
    { ProfScope ps_(K_UPFINISH, st); hipLaunchKernelGGL(upsample_finish_kernel, dim3(n), dim3(256), 0, st, (const float*)w.upart, nch,
                         (const float*)w.nyq, nyb, (const float*)w.xrec, p.Wp, p.Px,
                       (const int2*)w.coarse, shifts); }
  }
  IAP_LAUNCH_CHECK();
  return IAP_OK;
}

static int launch_mip(const uint16_t* in, long long in_stride, uint16_t* out, long long n, int Z, long long px,
                      hipStream_t st, PlaneSub sub = {}) {
  unsigned gx = (unsigned)((px / 8 + 255) / 256);
  if (gx > 512) gx = 512;
  if (gx == 0) gx = 1;
  for (long long i0 = 0; i0 < n; i0 += 32768) {
    long long nb = n - i0 < 32768 ? n - i0 : 32768;
    { ProfScope ps_(K_MIP, st); hipLaunchKernelGGL(mip_kernel, dim3(gx, (unsigned)nb), dim3(256), 0, st, in + i0 * in_stride, in_stride,
                       out + i0 * px, Z, px, sub.sub2 ? sub.sub2 + i0 * sub.stride : nullptr, sub.stride); }
  }
  IAP_LAUNCH_CHECK();
  return IAP_OK;
}

static int launch_apply(const uint16_t* in, uint16_t* out, long long n_frames, long long ppf, int Y, int X,
                        const double* shifts, hipStream_t st, const int* sub2 = nullptr) {
  const int tiles_x = (X + AP_TX - 1) / AP_TX, tiles_y = (Y + AM_TY - 1) / AM_TY;
  // each workgroup walks `nty` steps down its strip.  Long walks amortise the 25-row lead-in, short ones keep
  // neighbouring strips close enough in time for the columns they share to be served by L2: 8 measured best
  int nty = 8;
  while (nty > 1 && (long long)tiles_x * ((tiles_y + nty - 1) / nty) * ppf * n_frames < 2048) nty >>= 1;
  dim3 grid(tiles_x, (tiles_y + nty - 1) / nty, 1);
  // blockIdx.z is limited to 65535: chunk by whole frames
  long long fpc = 65535 / ppf;
  if (fpc < 1) return IAP_E_UNSUPPORTED;
  for (long long f0 = 0; f0 < n_frames; f0 += fpc) {
    long long nf = n_frames - f0 < fpc ? n_frames - f0 : fpc;
    grid.z = (unsigned)(nf * ppf);
    ProfScope ps_(K_APPLY, st);
    hipLaunchKernelGGL(apply_shift_kernel, grid, dim3(256), 0, st, in + (size_t)f0 * ppf * Y * X,
                       out + (size_t)f0 * ppf * Y * X, ppf, Y, X, shifts + 2 * f0, nty,
                       sub2 ? sub2 + f0 * ppf : nullptr);
  }
  IAP_LAUNCH_CHECK();
  return IAP_OK;
}

}  // namespace iap

using namespace iap;


extern "C" int iap_mip_u16(const uint16_t* in, int64_t in_stride, uint16_t* out, int64_t n, int Z, int Y, int X,
                           void* stream) {
  IAP_REQUIRE(in && out, IAP_E_ARG, "iap_mip_u16: null pointer");
  IAP_REQUIRE(n > 0 && Z > 0 && Y > 0 && X > 0, IAP_E_ARG, "iap_mip_u16: non-positive size");
  long long px = (long long)Y * X;
  IAP_REQUIRE(in_stride >= (int64_t)Z * px, IAP_E_ARG, "iap_mip_u16: in_stride smaller than one item");
  // planes of a multiple of 8 pixels go through 16-byte accesses (every plane of every item must then start on a
  // 16-byte boundary); any other plane size takes the kernel's scalar path and needs no alignment at all
  IAP_REQUIRE(px % 8 != 0 || (aligned16(in) && aligned16(out) && (in_stride % 8 == 0 || n == 1)), IAP_E_ALIGN,
              "iap_mip_u16: with Y*X a multiple of 8, pointers must be 16-byte aligned and in_stride a multiple of 8");
  return launch_mip(in, in_stride, out, n, Z, px, as_stream(stream));
}

extern "C" int iap_mip_sub_u16(const uint16_t* in, int64_t in_stride, uint16_t* out, int64_t n, int Z, int Y, int X,
                               const int32_t* sub2, int64_t sub_stride, void* stream) {
  IAP_REQUIRE(in && out && sub2, IAP_E_ARG, "iap_mip_sub_u16: null pointer");
  IAP_REQUIRE(n > 0 && Z > 0 && Y > 0 && X > 0, IAP_E_ARG, "iap_mip_sub_u16: non-positive size");
  long long px = (long long)Y * X;
  IAP_REQUIRE(in_stride >= (int64_t)Z * px && sub_stride >= Z, IAP_E_ARG, "iap_mip_sub_u16: stride smaller than one item");
  // planes of a multiple of 8 pixels go through 16-byte accesses (every plane of every item must then start on a
  // 16-byte boundary); any other plane size takes the kernel's scalar path and needs no alignment at all
  IAP_REQUIRE(px % 8 != 0 || (aligned16(in) && aligned16(out) && (in_stride % 8 == 0 || n == 1)), IAP_E_ALIGN,
              "iap_mip_sub_u16: with Y*X a multiple of 8, pointers must be 16-byte aligned and in_stride a multiple of 8");
  PlaneSub s;
  s.sub2 = (const int*)sub2;
  s.stride = sub_stride;
  return launch_mip(in, in_stride, out, n, Z, px, as_stream(stream), s);
}

extern "C" size_t iap_reg_ws_bytes(int Y, int X, int Z, int batch) {
  RegPlan p;
  if (Y <= 0 || X <= 0 || Z <= 0 || !make_plan(Y, X, p)) return 0;
  if (batch < 1) batch = 1;
  return reg_fixed_bytes(p) + (size_t)batch * reg_frame_bytes(p, Z) + 512;
}

static int reg_check(const char* fn, int Y, int X, RegPlan& p) {
  IAP_REQUIRE(Y > 0 && X > 0, IAP_E_ARG, "%s: non-positive plane size", fn);
  IAP_REQUIRE(make_plan(Y, X, p), IAP_E_UNSUPPORTED, "%s: planes must be at least 64x64 (got %dx%d)", fn, Y, X);
  return IAP_OK;
}

extern "C" int iap_reg_estimate_u16(const void* ref, int ref_kind, const uint16_t* mov, int64_t mov_stride, int64_t n,
                                    int Y, int X, int flags, double* shifts_out, void* ws, size_t ws_bytes,
                                    void* stream) {
  IAP_REQUIRE(ref && mov && shifts_out && ws, IAP_E_ARG, "iap_reg_estimate_u16: null pointer");
  IAP_REQUIRE(n > 0, IAP_E_ARG, "iap_reg_estimate_u16: n must be positive");
  IAP_REQUIRE(ref_kind >= 0 && ref_kind <= 2, IAP_E_ARG, "iap_reg_estimate_u16: ref_kind must be 0, 1 or 2");
  IAP_REQUIRE(mov_stride >= (int64_t)Y * X || n == 1, IAP_E_ARG, "iap_reg_estimate_u16: mov_stride too small");
  RegPlan p;
  int rc = reg_check("iap_reg_estimate_u16", Y, X, p);
  if (rc) return rc;
  RegWs w;
  IAP_REQUIRE(carve(ws, ws_bytes, p, 1, w), IAP_E_WS, "iap_reg_estimate_u16: workspace of %zu bytes, need %zu",
              ws_bytes, iap_reg_ws_bytes(Y, X, 1, 1));
  hipStream_t st = as_stream(stream);
  rc = reg_init_tables(p, w, st);
  if (rc) return rc;
  rc = reg_ref_spectrum(p, w, ref, ref_kind, 1.0f, st);
  if (rc) return rc;
  for (int64_t f0 = 0; f0 < n; f0 += w.batch) {
    int nb = (int)(n - f0 < w.batch ? n - f0 : w.batch);
    rc = reg_estimate_batch(p, w, mov + f0 * mov_stride, mov_stride, nb, flags, shifts_out + 2 * f0, st);
    if (rc) return rc;
  }
  return IAP_OK;
}

// Pairwise estimate (channel shift, utility.py:313-330: every plane against the same frame's plane of the
// registration channel): the workspace is that of the plain estimate plus one reference spectrum per
// frame of the batch.
// one reference spectrum (half pairs) and its per-tile scales per frame of the batch
static size_t pair_ref_bytes(const RegPlan& p) { return align256(p.spec * p.fb) + align256((size_t)p.col_blocks * 4); }

extern "C" size_t iap_reg_pairs_ws_bytes(int Y, int X, int batch) {
  RegPlan p;
  if (!make_plan(Y, X, p)) return 0;
  if (batch < 1) batch = 1;
  return reg_fixed_bytes(p) + (size_t)batch * (reg_frame_bytes(p, 1) + pair_ref_bytes(p)) + 512;
}

extern "C" int iap_reg_estimate_pairs_u16(const uint16_t* ref, int64_t ref_stride, const uint16_t* mov,
                                          int64_t mov_stride, int64_t n, int Y, int X, int flags, double* shifts_out,
                                          void* ws, size_t ws_bytes, void* stream) {
  IAP_REQUIRE(ref && mov && shifts_out && ws, IAP_E_ARG, "iap_reg_estimate_pairs_u16: null pointer");
  IAP_REQUIRE(n > 0, IAP_E_ARG, "iap_reg_estimate_pairs_u16: n must be positive");
  IAP_REQUIRE(n == 1 || (mov_stride >= (int64_t)Y * X && ref_stride >= (int64_t)Y * X), IAP_E_ARG,
              "iap_reg_estimate_pairs_u16: stride smaller than a plane");
  RegPlan p;
  int rc = reg_check("iap_reg_estimate_pairs_u16", Y, X, p);
  if (rc) return rc;
  // split the workspace: per-frame reference spectra and their scales first, the plain estimate's layout in the rest
  const size_t gs = align256(p.spec * p.fb), ss = align256((size_t)p.col_blocks * 4), per = reg_frame_bytes(p, 1) + gs + ss,
               fixed = reg_fixed_bytes(p);
  IAP_REQUIRE(ws_bytes >= fixed + per + 512, IAP_E_WS, "iap_reg_estimate_pairs_u16: workspace of %zu bytes, need %zu",
              ws_bytes, iap_reg_pairs_ws_bytes(Y, X, 1));
  int batch = (int)((ws_bytes - fixed - 512) / per);
  if (batch > 4096) batch = 4096;
  char* base = (char*)(((uintptr_t)ws + 255) & ~(uintptr_t)255);
  char* gref = base;
  float* gscale = (float*)(base + (size_t)batch * gs);
  RegWs w;
  IAP_REQUIRE(carve(base + (size_t)batch * (gs + ss), fixed + (size_t)batch * reg_frame_bytes(p, 1) + 256, p, 1, w) && w.batch >= 1,
              IAP_E_WS, "iap_reg_estimate_pairs_u16: workspace too small");
  if (w.batch > batch) w.batch = batch;
  hipStream_t st = as_stream(stream);
  rc = reg_init_tables(p, w, st);
  if (rc) return rc;
  RegRef rr;
  rr.gref = gref;
  rr.g_stride = (long long)(gs / p.fb);
  rr.gscale = gscale;
  for (int64_t f0 = 0; f0 < n; f0 += w.batch) {
    const int nb = (int)(n - f0 < w.batch ? n - f0 : w.batch);
    // reference planes -> float row spectra in the Q slots -> full spectra (half pairs), one per frame
    rc = launch_rows_r2c(p, ref + f0 * ref_stride, 0, ref_stride, 1.0f, w.Q, spec32_stride(p), false, nb, w.tw_x, st);
    if (rc) return rc;
    rc = launch_cols_ref(p, w.Q, spec32_stride(p), gref, rr.g_stride, gscale, nb, w.tw_y, st);
    if (rc) return rc;
    rc = reg_estimate_batch(p, w, mov + f0 * mov_stride, mov_stride, nb, flags, shifts_out + 2 * f0, st, rr);
    if (rc) return rc;
  }
  return IAP_OK;
}

extern "C" int iap_apply_shift_u16(const uint16_t* in, uint16_t* out, int64_t n_frames, int64_t planes_per_frame,
                                   int Y, int X, const double* shifts, void* stream) {
  IAP_REQUIRE(in && out && shifts, IAP_E_ARG, "iap_apply_shift_u16: null pointer");
  IAP_REQUIRE(in != out, IAP_E_ARG, "iap_apply_shift_u16: in-place operation is not supported");
  IAP_REQUIRE(n_frames > 0 && planes_per_frame > 0 && Y > 0 && X > 0, IAP_E_ARG,
              "iap_apply_shift_u16: non-positive size");
  IAP_REQUIRE(planes_per_frame <= 65535, IAP_E_UNSUPPORTED, "iap_apply_shift_u16: more than 65535 planes per frame");
  return launch_apply(in, out, n_frames, planes_per_frame, Y, X, shifts, as_stream(stream));
}

// sub2 (device, one int per plane of the stack, or nullptr): every read of `in` sees max(v - sub2/2, 0)
static int im_reg_impl(const uint16_t* in, uint16_t* out, int64_t T, int C, int Z, int Y, int X,
                       int reg_channel, int ref_mode, int flags, const int* sub2, double* shifts_out, void* ws,
                       size_t ws_bytes, void* stream) {
  IAP_REQUIRE(in && out && shifts_out && ws, IAP_E_ARG, "iap_im_reg_u16: null pointer");
  IAP_REQUIRE(in != out, IAP_E_ARG, "iap_im_reg_u16: out must not alias in");
  IAP_REQUIRE(T > 0 && C > 0 && Z > 0, IAP_E_ARG, "iap_im_reg_u16: non-positive size");
  IAP_REQUIRE(reg_channel >= 0 && reg_channel < C, IAP_E_ARG, "iap_im_reg_u16: reg_channel %d not in 0..%d",
              reg_channel, C - 1);
  IAP_REQUIRE(ref_mode >= 0 && ref_mode <= 2, IAP_E_ARG,
              "iap_im_reg_u16: ref_mode must be IAP_REF_FIRST, _MEAN or _PREVIOUS");
  IAP_REQUIRE((int64_t)C * Z <= 65535 && T <= 65535, IAP_E_UNSUPPORTED, "iap_im_reg_u16: T and C*Z must be <= 65535");
  RegPlan p;
  int rc = reg_check("iap_im_reg_u16", Y, X, p);
  if (rc) return rc;
  const long long px = (long long)Y * X;
  // (planes of Y*X %% 8 != 0 pixels take the MIP kernel's scalar path, which needs no alignment)
  IAP_REQUIRE(Z == 1 || px % 8 != 0 || (aligned16(in) && aligned16(out)), IAP_E_ALIGN,
              "iap_im_reg_u16: z-stacks of planes with Y*X %% 8 == 0 need 16-byte aligned buffers");
  RegWs w;
  IAP_REQUIRE(carve(ws, ws_bytes, p, Z, w), IAP_E_WS, "iap_im_reg_u16: workspace of %zu bytes, need %zu", ws_bytes,
              iap_reg_ws_bytes(Y, X, Z, 1));
  hipStream_t st = as_stream(stream);
  rc = reg_init_tables(p, w, st);
  if (rc) return rc;
  const long long ppf = (long long)C * Z;
  const long long frame_stride = ppf * px;
  const long long chan_off = (long long)reg_channel * Z * px;

  // levels of the registration channel's planes from frame f0 on
  auto chan_sub = [&](long long f0) {
    PlaneSub s;
    if (sub2) {
      s.sub2 = sub2 + f0 * ppf + (long long)reg_channel * Z;
      s.stride = ppf;
    }
    return s;
  };
  // moving planes of frames [f0, f0+nb): either straight from the stack (Z == 1; `msub`: what is left to
  // subtract when they are read) or z-MIPs of the lowered planes
  auto mov_planes = [&](const uint16_t* stack, long long f0, int nb, const uint16_t*& mp, long long& ms,
                        PlaneSub& msub) -> int {
    if (Z == 1) {
      mp = stack + f0 * frame_stride + chan_off;
      ms = frame_stride;
      msub = chan_sub(f0);
      return IAP_OK;
    }
    int r = launch_mip(stack + f0 * frame_stride + chan_off, frame_stride, w.mip, nb, Z, px, st, chan_sub(f0));
    mp = w.mip;
    ms = px;
    msub = PlaneSub{};
    return r;
  };

  if (ref_mode == IAP_REF_PREVIOUS) {
    IAP_HIP(hipMemsetAsync(shifts_out, 0, 16, st));
    if (sub2) {   // frame 0 is a copy of the lowered planes: the warp with a zero shift is exactly that
      rc = launch_apply(in, out, 1, ppf, Y, X, shifts_out, st, sub2);
      if (rc) return rc;
    } else {
      IAP_HIP(hipMemcpyAsync(out, in, (size_t)frame_stride * 2, hipMemcpyDeviceToDevice, st));
    }
    // The chain reference -> estimate -> warp is serial in time, the row pass over the raw moving planes is
    // not: it runs ahead, one launch per group of frames, into slots 1.. of w.F (slot 0 is the reference's).
    const long long fs = spec16_stride(p);
    const int ahead = w.batch - 1;
    for (long long f0 = 1; f0 < T; f0 += std::max(ahead, 1)) {
      const int nb = (int)std::min<long long>(std::max(ahead, 1), T - f0);
      const uint16_t* mp;
      long long ms;
      PlaneSub msub;
      rc = mov_planes(in, f0, nb, mp, ms, msub);
      if (rc) return rc;
      if (ahead) {
        rc = launch_rows_r2c(p, mp, 0, ms, 1.0f, w.F + fs * p.fb, fs, true, nb, w.tw_x, st, msub);
        if (rc) return rc;
      }
      for (int k = 0; k < nb; ++k) {
        const long long f = f0 + k;
        const uint16_t* refp;
        if (Z == 1) {
          refp = out + (f - 1) * frame_stride + chan_off;
        } else {
          rc = launch_mip(out + (f - 1) * frame_stride + chan_off, frame_stride, w.mip_ref, 1, Z, px, st);
          if (rc) return rc;
          refp = w.mip_ref;
        }
        if (ahead) {   // the reference's row pass into slot 0; its column FFT rides in the cross-power launch
          rc = launch_rows_r2c(p, refp, 0, 0, 1.0f, w.F, 0, true, 1, w.tw_x, st);
          if (rc) return rc;
          RegRef rr;
          rr.ref_rows = w.F;
          rc = reg_estimate_batch(p, w, mp, ms, 1, flags, shifts_out + 2 * f, st, rr, w.F + (size_t)(k + 1) * fs * p.fb);
        } else {   // a one-frame workspace: the same arithmetic, the reference's row spectrum in w.refspec's place
          rc = launch_rows_r2c(p, refp, 0, 0, 1.0f, w.refspec, 0, true, 1, w.tw_x, st);
          if (rc) return rc;
          RegRef rr;
          rr.ref_rows = w.refspec;
          rc = reg_estimate_batch(p, w, mp, ms, 1, flags, shifts_out + 2 * f, st, rr, nullptr, msub);
        }
        if (rc) return rc;
        rc = launch_apply(in + f * frame_stride, out + f * frame_stride, 1, ppf, Y, X, shifts_out + 2 * f, st,
                          sub2 ? sub2 + f * ppf : nullptr);
        if (rc) return rc;
      }
    }
    return IAP_OK;
  }

  if (ref_mode == IAP_REF_FIRST) {
    const uint16_t* mp;
    PlaneSub rsub;
    if (Z == 1) {
      mp = in + chan_off;
      rsub = chan_sub(0);
    } else {
      rc = launch_mip(in + chan_off, frame_stride, w.mip_ref, 1, Z, px, st, chan_sub(0));
      if (rc) return rc;
      mp = w.mip_ref;
    }
    rc = reg_ref_spectrum(p, w, mp, 0, 1.0f, st, rsub);
    if (rc) return rc;
  } else {  // mean over T of the (MIP) planes: exact u32 sum, scaled by 1/T on load
    IAP_HIP(hipMemsetAsync(w.sum, 0, (size_t)px * 4, st));
    unsigned gx = (unsigned)((px + 255) / 256);
    if (gx > 4096) gx = 4096;
    for (long long f0 = 0; f0 < T; f0 += w.batch) {
      int nb = (int)(T - f0 < w.batch ? T - f0 : w.batch);
      const uint16_t* mp;
      long long ms;
      PlaneSub msub;
      rc = mov_planes(in, f0, nb, mp, ms, msub);
      if (rc) return rc;
      const int vec = px % 8 == 0 && ms % 8 == 0 && aligned16(mp) && aligned16(w.sum);
      { ProfScope ps_(K_ACCUM, st); hipLaunchKernelGGL(accum_planes_kernel, dim3(vec ? (unsigned)std::min<long long>((px / 8 + 255) / 256, 4096) : gx), dim3(256), 0, st, mp, ms, (long long)nb, px, w.sum, vec, msub.sub2, msub.stride); }
    }
    IAP_LAUNCH_CHECK();
    rc = reg_ref_spectrum(p, w, w.sum, 1, 1.0f / (float)T, st);
    if (rc) return rc;
  }
  for (long long f0 = 0; f0 < T; f0 += w.batch) {
    int nb = (int)(T - f0 < w.batch ? T - f0 : w.batch);
    const uint16_t* mp;
    long long ms;
    PlaneSub msub;
    rc = mov_planes(in, f0, nb, mp, ms, msub);
    if (rc) return rc;
    rc = reg_estimate_batch(p, w, mp, ms, nb, flags, shifts_out + 2 * f0, st, RegRef{}, nullptr, msub);
    if (rc) return rc;
  }
  // 'first': frame 0 is the reference itself, its shift is zero by definition (the estimate, summed in f32
  // from Q, lands within 1e-4 px of it; a non-zero remainder would push the plane's last row or column out)
  if (ref_mode == IAP_REF_FIRST) IAP_HIP(hipMemsetAsync(shifts_out, 0, 16, st));
  return launch_apply(in, out, T, ppf, Y, X, shifts_out, st, sub2);
}

extern "C" int iap_apply_shift_sub_u16(const uint16_t* in, uint16_t* out, int64_t n_frames, int64_t planes_per_frame,
                                       int Y, int X, const double* shifts, const int32_t* sub2, void* stream) {
  IAP_REQUIRE(in && out && shifts && sub2, IAP_E_ARG, "iap_apply_shift_sub_u16: null pointer");
  IAP_REQUIRE(in != out, IAP_E_ARG, "iap_apply_shift_sub_u16: in-place operation is not supported");
  IAP_REQUIRE(n_frames > 0 && planes_per_frame > 0 && Y > 0 && X > 0, IAP_E_ARG,
              "iap_apply_shift_sub_u16: non-positive size");
  IAP_REQUIRE(planes_per_frame <= 65535, IAP_E_UNSUPPORTED, "iap_apply_shift_sub_u16: more than 65535 planes per frame");
  return launch_apply(in, out, n_frames, planes_per_frame, Y, X, shifts, as_stream(stream), (const int*)sub2);
}

extern "C" int iap_im_reg_u16(const uint16_t* in, uint16_t* out, int64_t T, int C, int Z, int Y, int X,
                              int reg_channel, int ref_mode, int flags, double* shifts_out, void* ws, size_t ws_bytes,
                              void* stream) {
  return im_reg_impl(in, out, T, C, Z, Y, X, reg_channel, ref_mode, flags, nullptr, shifts_out, ws, ws_bytes, stream);
}

extern "C" int iap_im_reg_sub_u16(const uint16_t* in, uint16_t* out, int64_t T, int C, int Z, int Y, int X,
                                  int reg_channel, int ref_mode, int flags, const int32_t* sub2, double* shifts_out,
                                  void* ws, size_t ws_bytes, void* stream) {
  IAP_REQUIRE(sub2, IAP_E_ARG, "iap_im_reg_sub_u16: sub2 is null (use iap_im_reg_u16)");
  return im_reg_impl(in, out, T, C, Z, Y, X, reg_channel, ref_mode, flags, (const int*)sub2, shifts_out, ws, ws_bytes,
                     stream);
}
