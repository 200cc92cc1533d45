// Thread-local last-error string of libcreste_hip.so (never throws across the C ABI).
#include <stdarg.h>
#include <stdio.h>

#include "../../include/creste_hip.h"

namespace creste {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace creste

extern "C" const char* creste_last_error(void) { return creste::g_err; }
extern "C" int creste_abi_version(void) { return 12; }
