// Per-object work of one level as ONE C-ABI call each: the K BoxMLPs (obbpose_model.py:174-201) are
// small, latency-bound launches (5 % of the rays hit a box), so driving them one by one from the host
// language costs more host time than GPU time at K = 8.  These entry points loop over the objects
// in C and spread them over a few side streams (fork/join with events around the caller's stream),
// so independent objects overlap on the GPU.  Buffers are [K, ...] slabs with the per-object
// strides documented in include/durf_hip.h; every kernel is the one the per-object entry points launch.
#include "durf_common.h"
#include "mlp_spec.h"

namespace {
constexpr int NSIDE = 4;
struct Side {
    hipStream_t s[NSIDE];
    hipEvent_t fork, join[NSIDE];
    bool ok = false;
};
thread_local Side g_side[16];

Side* side_streams() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
    Side& S = g_side[dev];
    if (!S.ok) {
        for (int i = 0; i < NSIDE; i++) {
            if (hipStreamCreateWithFlags(&S.s[i], hipStreamNonBlocking) != hipSuccess) return nullptr;
            if (hipEventCreateWithFlags(&S.join[i], hipEventDisableTiming) != hipSuccess) return nullptr;
        }
        if (hipEventCreateWithFlags(&S.fork, hipEventDisableTiming) != hipSuccess) return nullptr;
        S.ok = true;
    }
    return &S;
}

// run body(k, stream) for k in [0,K) on the side streams, ordered after / before `main`
template <class F>
int fan_out(void* main_stream, int K, F body) {
    hipStream_t main = (hipStream_t)main_stream;
    Side* S = K > 1 ? side_streams() : nullptr;
    if (!S) {                                         // one object (or no side streams): stay on the caller's stream
        for (int k = 0; k < K; k++) { const int rc = body(k, main_stream); if (rc) return rc; }
        return 0;
    }
    const int used = K < NSIDE ? K : NSIDE;
    if (hipEventRecord(S->fork, main) != hipSuccess) return -1;
    for (int i = 0; i < used; i++) if (hipStreamWaitEvent(S->s[i], S->fork, 0) != hipSuccess) return -1;
    int rc = 0;
    for (int k = 0; k < K && rc == 0; k++) rc = body(k, (void*)S->s[k % used]);
    for (int i = 0; i < used; i++) {
        if (hipEventRecord(S->join[i], S->s[i]) != hipSuccess) return -1;
        if (hipStreamWaitEvent(main, S->join[i], 0) != hipSuccess) return -1;
    }
    return rc;
}
size_t tile_rows(size_t rows) { return (rows + 31) / 32 * 32; }
}  // namespace

extern "C" {

size_t durf_obj_enc_stride(int B, int N) { return tile_rows((size_t)B * N) * DURF_ENC_DIM * 2; }
size_t durf_obj_view_stride(int B, int N) { return tile_rows((size_t)B * N) * DURF_VIEW_DIM * 2; }
size_t durf_obj_dzout_stride(int B, int N) { return tile_rows((size_t)B * N) * 16 * 2; }

int durf_pack_weights_batch(void* stream, int width, int in_dim, int K, const float* mlp_params,
                            size_t param_stride, void* wpack_fwd, void* wpack_bwd) {
    const size_t sf = durf_wpack_fwd_bytes(width), sb = durf_wpack_bwd_bytes(width);
    for (int k = 0; k < K; k++) {
        const int rc = durf_pack_weights(stream, width, in_dim, mlp_params + (size_t)k * param_stride,
                                         (char*)wpack_fwd + k * sf, wpack_bwd ? (char*)wpack_bwd + k * sb : nullptr);
        if (rc) return rc;
    }
    return 0;
}

int durf_obj_fwd_batch(void* stream, int K, int B, int N, const int32_t* idx, const int32_t* count,
                       const float* t_vals, const float* origins_s, const float* dirs_s, const float* radii,
                       const float* barf_w, int flags, const void* view_bf16, const void* wpack_fwd,
                       void* enc, float* raw, void* stash, void* relu_mask, void* view_tile) {
    const size_t rows = (size_t)B * N;
    const size_t s_enc = durf_obj_enc_stride(B, N), s_wf = durf_wpack_fwd_bytes(DURF_W_OBJ);
    const size_t s_st = durf_mlp_stash_bytes(DURF_W_OBJ, rows), s_mk = durf_mlp_mask_bytes(rows);
    const size_t s_vt = durf_obj_view_stride(B, N);
    return fan_out(stream, K, [&](int k, void* st) -> int {
        const int32_t* idx_k = idx + (size_t)k * B;
        const int32_t* cnt_k = count + k;
        char* enc_k = (char*)enc + k * s_enc;
        int rc = durf_encode_obj(st, B, N, idx_k, cnt_k, t_vals, origins_s, dirs_s, radii, barf_w, flags, enc_k, nullptr);
        if (rc) return rc;
        rc = durf_mlp_fwd(st, DURF_W_OBJ, rows, N, enc_k, view_bf16, idx_k, cnt_k, (const char*)wpack_fwd + k * s_wf,
                          raw + (size_t)k * rows * 4, stash ? (char*)stash + k * s_st : nullptr,
                          relu_mask ? (char*)relu_mask + k * s_mk : nullptr);
        if (rc) return rc;
        if (view_tile) rc = durf_expand_view(st, rows, N, view_bf16, idx_k, cnt_k, (char*)view_tile + k * s_vt);
        return rc;
    });
}

int durf_obj_bwd_batch(void* stream, int K, int B, int N, const int32_t* idx, const int32_t* count,
                       const float* draw, const void* wpack_bwd, const void* relu_mask, void* dz, void* dz_out,
                       float* d_enc) {
    const size_t rows = (size_t)B * N;
    const size_t s_wb = durf_wpack_bwd_bytes(DURF_W_OBJ), s_st = durf_mlp_stash_bytes(DURF_W_OBJ, rows);
    const size_t s_mk = durf_mlp_mask_bytes(rows), s_do = durf_obj_dzout_stride(B, N);
    return fan_out(stream, K, [&](int k, void* st) -> int {
        return durf_mlp_bwd(st, DURF_W_OBJ, rows, N, draw, idx + (size_t)k * B, count + k,
                            (const char*)wpack_bwd + k * s_wb, (const char*)relu_mask + k * s_mk,
                            (char*)dz + k * s_st, (char*)dz_out + k * s_do,
                            d_enc ? d_enc + (size_t)k * rows * DURF_ENC_DIM : nullptr);
    });
}

int durf_obj_dw_batch(void* stream, int K, int B, int N, const int32_t* count, int nlevels,
                      const void* const* enc, const void* const* view_tile, const void* const* stash,
                      const void* const* dz, const void* const* dz_out, int in_dim, float* part, float* bpart,
                      float* grad_mlp, size_t grad_stride) {
    DURF_REQUIRE(nlevels >= 1 && nlevels <= DURF_MAX_LEVELS, "1 <= nlevels <= DURF_MAX_LEVELS");
    const size_t rows = (size_t)B * N;
    const size_t s_enc = durf_obj_enc_stride(B, N), s_vt = durf_obj_view_stride(B, N);
    const size_t s_st = durf_mlp_stash_bytes(DURF_W_OBJ, rows), s_do = durf_obj_dzout_stride(B, N);
    const size_t s_part = durf_dw_part_floats(DURF_W_OBJ), s_bpart = durf_dw_bpart_floats(DURF_W_OBJ);
    return fan_out(stream, K, [&](int k, void* st) -> int {
        const void *e[DURF_MAX_LEVELS], *v[DURF_MAX_LEVELS], *s[DURF_MAX_LEVELS], *d[DURF_MAX_LEVELS], *o[DURF_MAX_LEVELS];
        for (int l = 0; l < nlevels; l++) {
            e[l] = (const char*)enc[l] + k * s_enc;
            v[l] = (const char*)view_tile[l] + k * s_vt;
            s[l] = (const char*)stash[l] + k * s_st;
            d[l] = (const char*)dz[l] + k * s_st;
            o[l] = (const char*)dz_out[l] + k * s_do;
        }
        int rc = durf_mlp_dw(st, DURF_W_OBJ, rows, N, count + k, nlevels, e, v, s, d, o, part + k * s_part,
                             bpart + k * s_bpart);
        if (rc) return rc;
        return durf_mlp_dw_finalize(st, DURF_W_OBJ, in_dim, part + k * s_part, bpart + k * s_bpart,
                                    grad_mlp + (size_t)k * grad_stride);
    });
}

}  // extern "C"
