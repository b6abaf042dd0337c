// Library-wide plumbing: error string, version.
#include <stdarg.h>
#include "durf_common.h"

static thread_local char g_err[512] = "";

void durf_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" {
const char* durf_last_error(void) { return g_err; }
int durf_version(void) { return 25; }      // bump with every kernel change: bench.py quotes PMC traffic per version
}
