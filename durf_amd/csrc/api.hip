// Library-wide plumbing: error string, version.
#include <stdarg.h>
#include <atomic>
#include <mutex>
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

// Item counters of the mixed launches (k_mlp_fwd / k_mlp_bwd <.., MIX>): a launch needs ONE zeroed int and leaves it zeroed
// (the workgroup that draws the last ticket resets it), so a ring of them per device, zero filled once, serves every launch:
// two launches in flight on different streams never share a counter (the ring is 16 384 launches long).
int* durf::next_ticket() {
    constexpr int RING = 16384;
    static std::atomic<int*> tab[64];
    static std::atomic<unsigned> next[64];
    static std::mutex mu;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    int* base = tab[dev].load(std::memory_order_acquire);
    if (base == nullptr) {
        std::lock_guard<std::mutex> lock(mu);
        base = tab[dev].load(std::memory_order_relaxed);
        if (base == nullptr) {
            int* p = nullptr;
            if (hipMalloc((void**)&p, RING * sizeof(int)) != hipSuccess) return nullptr;
            if (hipMemset(p, 0, RING * sizeof(int)) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
                (void)hipFree(p);
                return nullptr;
            }
            tab[dev].store(p, std::memory_order_release);
            base = p;
        }
    }
    return base + (next[dev].fetch_add(1u, std::memory_order_relaxed) % RING);
}

extern "C" {
int durf_dispatch_seen(void) { return (int)g_dispatch.load(std::memory_order_relaxed); }
int durf_dispatch_reset(void) { g_dispatch.store(0u, std::memory_order_relaxed); return 0; }
const char* durf_last_error(void) { return g_err; }
int durf_version(void) { return 40; }      // bump with every kernel change: bench.py quotes PMC traffic per version
}
