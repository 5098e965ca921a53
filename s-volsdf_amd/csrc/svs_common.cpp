// Error reporting shared by every C-ABI entry point (thread-local last-error string).
#include "svs_common.h"

#include <stdarg.h>
#include <stdio.h>

namespace svs {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace svs

extern "C" {
int svs_version(void) { return 100; }
const char* svs_last_error_string(void) { return svs::g_err; }
}
