// Library-level C ABI plumbing: version string and per-thread error message.
#include "common.h"
#include <string.h>

static thread_local char g_err[512] = "";

void gs_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* gs_version(void) { return "goslam_hip 0.1.0 (gfx950)"; }
extern "C" const char* gs_last_error(void) { return g_err; }
