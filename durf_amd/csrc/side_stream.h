// The object side stream of the one-call entry points (csrc/train.hip, csrc/forward.hip).
#pragma once
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <hip/hip_runtime.h>

namespace durf {

// The K object MLPs of a LARGE step run beside the background MLP's kernels on a second stream, as durf_amd/obbpose_model.py
// and train_boxpose.py place them (ops.overlap_mode: from 2048 x 128 sample rows per level; below that every kernel is one
// latency-bound round and a fork / join is one more dependency in the chain): the forward's object launches are issued
// BEFORE the persistent background launch takes every CU, the object backward runs in its shadow, the objects' weight
// gradients (their own split-K launch + finalize) beside the background's.  One side stream + two events per device,
// created on first use; DURF_OVERLAP_OBJECTS=0 keeps everything on the caller's stream.  No result depends on it (no atomics).
// pending_trunk: the buffer an outstanding cross-step prefetch (durf_train_step: prefetch_const_trunk) is writing on this stream
// -- it also READS the parameters -- or nullptr; whoever uses, recomputes or is asked about that buffer joins first (join_prefetch).
struct SideStream { hipStream_t s; hipEvent_t forked, joined; bool ok; void* pending_trunk; };
inline std::mutex& side_mutex() { static std::mutex m; return m; }

inline SideStream* side_stream_of_device(bool create = true) {      // (inline: ONE table for the library, whichever file asks)
    static SideStream tab[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    std::lock_guard<std::mutex> lock(side_mutex());
    SideStream& t = tab[dev];
    if (!t.ok && !create) return nullptr;
    if (!t.ok) {
        if (hipStreamCreateWithFlags(&t.s, hipStreamNonBlocking) != hipSuccess) return nullptr;
        if (hipEventCreateWithFlags(&t.forked, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&t.joined, hipEventDisableTiming) != hipSuccess) return nullptr;
        t.ok = true;
    }
    return &t;
}

struct Overlap {
    hipStream_t main;
    SideStream* sd;                       // nullptr: one stream
    void* obj() const { return sd ? (void*)sd->s : (void*)main; }
    // fork: the side stream waits for everything issued so far on the caller's stream; join: the reverse
    int fork() const {
        if (!sd) return 0;
        std::lock_guard<std::mutex> lock(side_mutex());          // (record + wait as a pair: the events are per device)
        if (hipEventRecord(sd->forked, main) != hipSuccess || hipStreamWaitEvent(sd->s, sd->forked, 0) != hipSuccess) return 1;
        return 0;
    }
    int join() const {
        if (!sd) return 0;
        std::lock_guard<std::mutex> lock(side_mutex());
        if (hipEventRecord(sd->joined, sd->s) != hipSuccess || hipStreamWaitEvent(main, sd->joined, 0) != hipSuccess) return 1;
        return 0;
    }
};

// Orders `stream` behind an outstanding prefetch of this device, if there is one (no stream is created for the question).
inline int join_prefetch(void* stream) {
    SideStream* sd = side_stream_of_device(false);
    if (sd == nullptr) return 0;
    {
        std::lock_guard<std::mutex> lock(side_mutex());
        if (sd->pending_trunk == nullptr) return 0;
        sd->pending_trunk = nullptr;
    }
    return (Overlap{(hipStream_t)stream, sd}).join();
}
inline void note_prefetch(SideStream* sd, void* dst) {
    std::lock_guard<std::mutex> lock(side_mutex());
    sd->pending_trunk = dst;
}

// DURF_OVERLAP_OBJECTS as the one-call entry points read it: unset / "auto" = by size, "0" = one stream, anything else = "2"
// (forward, backward and weight gradients of the objects on the side stream).  The Python-issued path's experiment modes "1"
// (forward only) and "3" (forward + backward) exist there only (ops.overlap_mode): an A/B of those through the C call measures
// mode 2.  Read per call (the tests toggle it); the host must not call setenv concurrently with a step.
inline Overlap overlap_for(void* stream, size_t rows, int Kb) {
    Overlap o{(hipStream_t)stream, nullptr};
    const char* e = getenv("DURF_OVERLAP_OBJECTS");
    const bool want = (e == nullptr || !strcmp(e, "auto")) ? rows >= (size_t)2048 * 128 : strcmp(e, "0") != 0;
    if (Kb > 0 && want) o.sd = side_stream_of_device();
    return o;
}

}  // namespace durf
