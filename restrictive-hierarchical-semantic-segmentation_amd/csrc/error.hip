// Thread-local error string + ABI version for libhrseg_hip.so.
#include <stdarg.h>
#include <string.h>

#include "common.h"

static thread_local char g_err[512] = "";
int hrseg_g_deterministic = 0;

void hrseg_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* hrseg_last_error_string(void) { return g_err; }
extern "C" int hrseg_abi_version(void) { return 15; }
