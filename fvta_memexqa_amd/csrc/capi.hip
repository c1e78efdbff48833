// Library-level entry points: version and per-thread error string.
#include <stdarg.h>

#include "fvta_common.h"

static thread_local char g_err[512] = "";

void fvta_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int fvta_version(void) { return 100; }
extern "C" const char* fvta_last_error(void) { return g_err; }
