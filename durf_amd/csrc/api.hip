// Library-wide plumbing: error string, version.
#include <stdarg.h>
#include <atomic>
#include "durf_common.h"

static thread_local char g_err[512] = "";

void durf_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

static std::atomic<unsigned> g_dispatch{0u};
void durf::note_dispatch(unsigned bits) { g_dispatch.fetch_or(bits, std::memory_order_relaxed); }

extern "C" {
int durf_dispatch_seen(void) { return (int)g_dispatch.load(std::memory_order_relaxed); }
int durf_dispatch_reset(void) { g_dispatch.store(0u, std::memory_order_relaxed); return 0; }
const char* durf_last_error(void) { return g_err; }
int durf_version(void) { return 35; }      // bump with every kernel change: bench.py quotes PMC traffic per version
}
