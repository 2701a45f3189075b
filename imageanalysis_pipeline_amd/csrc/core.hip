// Library-wide host state: version string and the thread-local error message.
#include <stdarg.h>
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

extern "C" const char* iap_version(void) { return "libiap_hip 0.1 (gfx950)"; }
extern "C" const char* iap_last_error(void) { return iap::g_err; }
