// Error plumbing + version of libsfod_hip.so
#include <stdarg.h>
#include <stdio.h>

#include "../../include/sfod_hip.h"

static thread_local char g_err[512] = "";

void sfod_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int sfod_version(void) { return 100; }
extern "C" const char* sfod_last_error(void) { return g_err; }
