// Internal helpers shared by the HIP translation units of libiap_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include <stdio.h>
#include "../../include/iap.h"

namespace iap {

void set_error(const char* fmt, ...);

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

#define IAP_REQUIRE(cond, code, ...)          \
  do {                                        \
    if (!(cond)) {                            \
      iap::set_error(__VA_ARGS__);            \
      return (code);                          \
    }                                         \
  } while (0)

#define IAP_HIP(call)                                                                   \
  do {                                                                                  \
    hipError_t e__ = (call);                                                            \
    if (e__ != hipSuccess) {                                                            \
      iap::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__,  \
                     __LINE__);                                                         \
      return static_cast<int>(e__);                                                     \
    }                                                                                   \
  } while (0)

#define IAP_LAUNCH_CHECK()                                                              \
  do {                                                                                  \
    hipError_t e__ = hipGetLastError();                                                 \
    if (e__ != hipSuccess) {                                                            \
      iap::set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(e__),        \
                     __FILE__, __LINE__);                                               \
      return static_cast<int>(e__);                                                     \
    }                                                                                   \
  } while (0)

// Optional per-kernel timing with HIP events on the launch stream (iap_prof_* in iap.h).
enum KernelId {
  K_SMO_MARCH = 0, K_SMO_GENERIC, K_MEDIAN, K_SUB, K_ROWS_R2C, K_COLS, K_ROWS_C2R, K_COARSE, K_UPSAMPLE,
  K_UPFINISH, K_APPLY, K_MEASURE, K_MEASURE_FIN, K_MIP, K_ACCUM, K_GAUSS, K_ROI, K_MAXLABEL, K_OTHER, K_COUNT
};
// (atomics: host threads may be inside the library at once, each on its own stream; a scope closes the record it
// opened -- the last open one of its kernel id ON ITS STREAM -- whatever other threads recorded in between)
extern std::atomic<bool> g_prof_on;
extern std::atomic<int> g_prof_only;   // kernel id the recording is restricted to, -1: every kernel
void prof_mark(int kid, hipStream_t st, bool start);
struct ProfScope {
  int kid;
  hipStream_t st;
  bool on;
  ProfScope(int k, hipStream_t s) : kid(k), st(s) {
    const int only = g_prof_only.load(std::memory_order_relaxed);
    on = g_prof_on.load(std::memory_order_relaxed) && (only < 0 || only == kid);
    if (on) prof_mark(kid, st, true);
  }
  ~ProfScope() {
    if (on) prof_mark(kid, st, false);
  }
};

// 8 x u16 = one 16-byte lane access (guide G13: always vectorise 16-bit streams).
struct alignas(16) u16x8 {
  uint16_t v[8];
};

__device__ __forceinline__ u16x8 load8(const uint16_t* p) { return *reinterpret_cast<const u16x8*>(p); }
__device__ __forceinline__ void store8(uint16_t* p, const u16x8& v) { *reinterpret_cast<u16x8*>(p) = v; }
// streaming variants: data that is not touched again soon (keeps it from displacing reusable lines)
typedef unsigned iap_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u16x8 load8_nt(const uint16_t* p) {
  iap_u32x4 w = __builtin_nontemporal_load(reinterpret_cast<const iap_u32x4*>(p));
  u16x8 v;
  __builtin_memcpy(&v, &w, 16);
  return v;
}
__device__ __forceinline__ void store8_nt(uint16_t* p, const u16x8& v) {
  iap_u32x4 w;
  __builtin_memcpy(&w, &v, 16);
  __builtin_nontemporal_store(w, reinterpret_cast<iap_u32x4*>(p));
}

// XCD-aware block remap (guide T1): blocks b and b+8 share an XCD/L2; give each XCD a
// contiguous range of logical tiles.  Bijective for any n.
__device__ __forceinline__ unsigned xcd_remap(unsigned b, unsigned n) {
  unsigned q = n >> 3, r = n & 7u, x = b & 7u;
  unsigned base = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
  return base + (b >> 3);
}

// OpenCV BORDER_REFLECT_101 / scipy 'mirror' index: d c b | a b c d | c b a
__device__ __forceinline__ int reflect101_idx(int i, int n) {
  if (n == 1) return 0;
  int p = 2 * (n - 1);
  i %= p;
  if (i < 0) i += p;
  return i >= n ? p - i : i;
}

}  // namespace iap
