// Library-wide host state: version string and the thread-local error message.
#include <stdarg.h>
#include <mutex>
#include <vector>
#include "common.h"

namespace iap {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace iap

namespace iap {
std::atomic<bool> g_prof_on{false};
std::atomic<int> g_prof_only{-1};
namespace {
struct ProfRec {
  int kid;
  hipStream_t st;
  hipEvent_t a, b;
};
std::vector<ProfRec> g_recs;
std::vector<hipEvent_t> g_pool;
std::mutex g_prof_mu;
hipEvent_t take_event() {
  if (!g_pool.empty()) {
    hipEvent_t e = g_pool.back();
    g_pool.pop_back();
    return e;
  }
  hipEvent_t e = nullptr;
  (void)hipEventCreate(&e);
  return e;
}
const char* const k_names[K_COUNT] = {"smo_march_kernel", "smo_generic_kernel", "median_kernel", "sub_half_kernel",
                                      "rows_r2c16_kernel", "cols_xpower_kernel", "rows_c2r_argmax_kernel",
                                      "coarse_finish_kernel", "upsample_partial_kernel", "upsample_finish_kernel",
                                      "apply_shift_kernel", "measure_kernel", "measure_finalize_kernel", "mip_kernel",
                                      "accum_planes_kernel", "gauss_kernels", "roi_kernels", "max_u16_kernel", "other"};
}  // namespace

void prof_mark(int kid, hipStream_t st, bool start) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  if (start) {
    ProfRec r{kid, st, take_event(), nullptr};
    if (r.a) (void)hipEventRecord(r.a, st);
    g_recs.push_back(r);
  } else {
    for (size_t i = g_recs.size(); i-- > 0;) {
      if (g_recs[i].kid == kid && g_recs[i].st == st && g_recs[i].b == nullptr) {
        g_recs[i].b = take_event();
        if (g_recs[i].b) (void)hipEventRecord(g_recs[i].b, st);
        break;
      }
    }
  }
}
}  // namespace iap

extern "C" void iap_prof_enable(int on) {
  std::lock_guard<std::mutex> lk(iap::g_prof_mu);
  for (auto& r : iap::g_recs) {
    if (r.a) iap::g_pool.push_back(r.a);
    if (r.b) iap::g_pool.push_back(r.b);
  }
  iap::g_recs.clear();
  iap::g_prof_on.store(on != 0);
}

extern "C" void iap_prof_select(int kernel_id) {
  std::lock_guard<std::mutex> lk(iap::g_prof_mu);
  iap::g_prof_only.store((kernel_id >= 0 && kernel_id < iap::K_COUNT) ? kernel_id : -1);
}

extern "C" int iap_prof_kernel_count(void) { return iap::K_COUNT; }

extern "C" const char* iap_prof_kernel_name(int kid) {
  return (kid >= 0 && kid < iap::K_COUNT) ? iap::k_names[kid] : "";
}

extern "C" int iap_prof_collect(int kid, double* total_ms, int64_t* launches) {
  if (!total_ms || !launches || kid < 0 || kid >= iap::K_COUNT) {
    iap::set_error("iap_prof_collect: bad argument");
    return IAP_E_ARG;
  }
  std::lock_guard<std::mutex> lk(iap::g_prof_mu);
  double tot = 0;
  int64_t n = 0;
  for (auto& r : iap::g_recs) {
    if (r.kid != kid || !r.a || !r.b) continue;
    hipError_t e = hipEventSynchronize(r.b);
    if (e != hipSuccess) {
      iap::set_error("iap_prof_collect: hipEventSynchronize: %s", hipGetErrorString(e));
      return (int)e;
    }
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
      tot += ms;
      ++n;
    }
  }
  *total_ms = tot;
  *launches = n;
  return IAP_OK;
}

extern "C" const char* iap_version(void) { return "libiap_hip 0.1 (gfx950)"; }
extern "C" const char* iap_last_error(void) { return iap::g_err; }
