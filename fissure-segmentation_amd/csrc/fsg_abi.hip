// Version / error plumbing of the C ABI (include/fsg_hip.h).
#include <stdarg.h>
#include <string.h>

#include "fsg_common.h"

static thread_local char g_err[512] = "";

void fsg_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int fsg_version(void) { return 100; }
extern "C" const char *fsg_last_error(void) { return g_err; }
