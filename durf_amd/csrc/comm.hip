// The data-parallel exchange of a step (jax.lax.pmean(grad, 'batch'), train_boxpose.py:253) issued by THIS library on the
// caller's compute stream: one in-place RCCL all-reduce (sum) of the flat fp32 gradient.  Why here as well as in the host's
// own collective layer (torch.distributed, which durf_amd/train_boxpose.py uses by default):
//   * a host that is not Python gets the whole data-parallel step through the C ABI (durf_train_args.comm);
//   * the call sits IN the stream between the weight gradients and the optimizer -- no hop to a communication stream and
//     back (the two event waits cost a world-size-1 step ~22 us of idle GPU, profiles/r05_rccl_instream.txt).
// RCCL is resolved at run time: first among the libraries the process has already loaded (a PyTorch host has its own
// librccl in memory: the communicator and the calls must come from ONE copy), then librccl.so.1 / librccl.so.  No link-time
// dependency: a single-GPU host never needs it, and a missing library is a clear error from durf_comm_init.
#include <dlfcn.h>
#include <link.h>
#include <string.h>
#include <rccl/rccl.h>
#include "durf_common.h"
#include "../../include/durf_hip.h"

namespace {

struct Rccl {
    ncclResult_t (*get_unique_id)(ncclUniqueId*);
    ncclResult_t (*comm_init_rank)(ncclComm_t*, int, ncclUniqueId, int);
    ncclResult_t (*comm_destroy)(ncclComm_t);
    ncclResult_t (*all_reduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);
    const char* (*error_string)(ncclResult_t);
    bool ok;
};

// how many distinct loaded objects are called like `stem` (libamdhip64 / librccl), and the path of the first
struct Loaded { const char* stem; int n; char first[512]; };
int count_loaded(struct dl_phdr_info* info, size_t, void* data) {
    Loaded* L = (Loaded*)data;
    const char* base = info->dlpi_name ? strrchr(info->dlpi_name, '/') : nullptr;
    base = base ? base + 1 : (info->dlpi_name ? info->dlpi_name : "");
    if (strncmp(base, L->stem, strlen(L->stem)) == 0) {
        if (L->n == 0) { strncpy(L->first, info->dlpi_name, sizeof(L->first) - 1); L->first[sizeof(L->first) - 1] = 0; }
        L->n++;
    }
    return 0;
}

const Rccl& rccl() {
    static const Rccl r = [] {
        Rccl x{};
        void* h = RTLD_DEFAULT;
        if (!dlsym(h, "ncclAllReduce")) {
            // A host that loaded its RCCL privately (Python imports are RTLD_LOCAL: a PyTorch process has librccl in memory but
            // not in the global scope): take THAT object, by the path it was loaded from -- the communicator, the collective
            // and the streams they are handed must all belong to the one HIP runtime the process runs on
            Loaded have{"librccl", 0, ""};
            dl_iterate_phdr(count_loaded, &have);
            h = have.n > 0 ? dlopen(have.first, RTLD_NOW | RTLD_NOLOAD) : nullptr;
            if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);      // (by soname: a copy already loaded under it is returned)
            if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        }
        if (!h && !dlsym(RTLD_DEFAULT, "ncclAllReduce")) return x;
        // ... and refuse when resolving it brought a SECOND HIP runtime into the process (an RCCL bound to another
        // libamdhip64 than the host's: its calls would be handed streams of a foreign runtime)
        Loaded hip{"libamdhip64", 0, ""};
        dl_iterate_phdr(count_loaded, &hip);
        if (hip.n > 1) {
            durf_set_error("durf_comm: %d HIP runtimes are loaded (first: %s): the RCCL found does not share the host's", hip.n, hip.first);
            return x;
        }
        x.get_unique_id = (decltype(x.get_unique_id))dlsym(h, "ncclGetUniqueId");
        x.comm_init_rank = (decltype(x.comm_init_rank))dlsym(h, "ncclCommInitRank");
        x.comm_destroy = (decltype(x.comm_destroy))dlsym(h, "ncclCommDestroy");
        x.all_reduce = (decltype(x.all_reduce))dlsym(h, "ncclAllReduce");
        x.error_string = (decltype(x.error_string))dlsym(h, "ncclGetErrorString");
        x.ok = x.get_unique_id && x.comm_init_rank && x.comm_destroy && x.all_reduce && x.error_string;
        return x;
    }();
    return r;
}

#define DURF_RCCL(call, what)                                                                           \
    do {                                                                                                \
        const ncclResult_t r__ = (call);                                                                \
        if (r__ != ncclSuccess) {                                                                       \
            durf_set_error("%s: %s failed: %s", __func__, what, rccl().error_string(r__));               \
            return -1;                                                                                  \
        }                                                                                               \
    } while (0)

}  // namespace

extern "C" {

int durf_comm_available(void) { return rccl().ok ? 1 : 0; }

int durf_comm_unique_id(void* id_out) {
    DURF_REQUIRE(id_out != nullptr, "a host buffer of DURF_COMM_ID_BYTES bytes");
    DURF_REQUIRE(rccl().ok, "no RCCL in this process and none could be loaded (librccl.so.1)");
    static_assert(sizeof(ncclUniqueId) == DURF_COMM_ID_BYTES, "DURF_COMM_ID_BYTES");
    ncclUniqueId id;
    DURF_RCCL(rccl().get_unique_id(&id), "ncclGetUniqueId");
    memcpy(id_out, &id, sizeof(id));
    return 0;
}

int durf_comm_init(int world, int rank, const void* id, void** comm_out) {
    DURF_REQUIRE(world >= 1 && rank >= 0 && rank < world && id != nullptr && comm_out != nullptr, "0 <= rank < world, id, comm_out");
    DURF_REQUIRE(rccl().ok, "no RCCL in this process and none could be loaded (librccl.so.1)");
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof(uid));
    ncclComm_t c = nullptr;
    DURF_RCCL(rccl().comm_init_rank(&c, world, uid, rank), "ncclCommInitRank");
    *comm_out = (void*)c;
    return 0;
}

int durf_comm_destroy(void* comm) {
    if (comm == nullptr) return 0;
    DURF_REQUIRE(rccl().ok, "no RCCL in this process");
    DURF_RCCL(rccl().comm_destroy((ncclComm_t)comm), "ncclCommDestroy");
    return 0;
}

int durf_allreduce_sum(void* stream, void* comm, float* buf, size_t n) {
    DURF_REQUIRE(comm != nullptr && buf != nullptr, "communicator and buffer");
    if (n == 0) return 0;
    DURF_RCCL(rccl().all_reduce(buf, buf, n, ncclFloat, ncclSum, (ncclComm_t)comm, (hipStream_t)stream), "ncclAllReduce");
    return 0;
}

}  // extern "C"
