// Per-object work of one level as ONE launch per kernel: the K BoxMLPs (obbpose_model.py:174-201) are
// small, latency-bound launches (5 % of the rays hit a box); driven one by one they leave the GPU idle
// behind the host (K = 8: 16 objects x levels x ~10 launches).  Here every kernel of the per-object path
// runs once with the object index in blockIdx.y (k_dw_finalize: blockIdx.z) over [K, ...] slabs whose
// strides are documented in include/durf_hip.h, so all objects' workgroups are resident together.
#include <stdlib.h>
#include "durf_common.h"
#include "mlp_spec.h"

namespace {
size_t tile_rows(size_t rows) { return (rows + 31) / 32 * 32; }
}  // namespace

extern "C" {

size_t durf_obj_enc_stride(int B, int N) { return tile_rows((size_t)B * N) * DURF_ENC_DIM * 2; }
size_t durf_obj_view_stride(int B, int N) { return tile_rows((size_t)B * N) * DURF_VIEW_DIM * 2; }
size_t durf_obj_dzout_stride(int B, int N) { return tile_rows((size_t)B * N) * 16 * 2; }

int durf_pack_weights_batch(void* stream, int width, int in_dim, int K, const float* mlp_params,
                            size_t param_stride, void* wpack_fwd, void* wpack_bwd) {
    return durf::launch_pack(stream, width, in_dim, K, mlp_params, param_stride, wpack_fwd, wpack_bwd);
}

int durf_obj_fwd_batch(void* stream, int K, int B, int N, const int32_t* idx, const int32_t* count,
                       const float* t_vals, const float* origins_s, const float* dirs_s, const float* radii,
                       const float* barf_w, int flags, const void* view_bf16, const void* wpack_fwd,
                       void* enc, float* raw, void* stash, void* relu_mask, void* view_tile) {
    const size_t rows = (size_t)B * N;
    FwdStrides st;
    st.enc = durf_obj_enc_stride(B, N); st.idx = (size_t)B; st.wpack = durf_wpack_fwd_bytes(DURF_W_OBJ);
    st.raw = rows * 4 * sizeof(float); st.stash = durf_mlp_stash_bytes(DURF_W_OBJ, rows); st.mask = durf_mlp_mask_bytes(rows);
    // ONE launch: the forward encodes its own tiles (enc_lane.h: k_encode_lane<true>'s body, bit-identical) and, in
    // training, writes the view-direction tile from the fragment it holds for the view layer (durf_expand_view's output).
    // (The three separate launches -- durf_encode_obj, durf_mlp_fwd, durf_expand_view -- remain as entry points of their own.)
    int rc;
    EncIn ei{};
    ei.t_vals = t_vals; ei.origins_s = origins_s; ei.dirs_s = dirs_s; ei.radii = radii;
    ei.flags = flags & (DURF_ENC_NO_INTEGRATION | DURF_ENC_CYLINDER); ei.obj = 1;
    for (int i = 0; i < 10; i++) ei.w[i] = barf_w[i];
    ei.view_tile = stash ? view_tile : nullptr; ei.view_stride = durf_obj_view_stride(B, N);
    rc = durf::launch_mlp_fwd(stream, DURF_W_OBJ, rows, N, enc, view_bf16, idx, count, wpack_fwd, raw, stash, relu_mask, K, st,
                              nullptr, nullptr, &ei);
    if (rc == 0 && view_tile && !stash)        // (inference callers that still ask for the tile)
        rc = durf::launch_expand_view(stream, rows, N, view_bf16, idx, count, view_tile, K, (size_t)B, durf_obj_view_stride(B, N));
    return rc;
}

int durf_obj_bwd_batch(void* stream, int K, int B, int N, const int32_t* idx, const int32_t* count,
                       const float* draw, const void* wpack_bwd, const void* relu_mask, void* dz, void* dz_out,
                       float* d_enc) {
    const size_t rows = (size_t)B * N;
    BwdStrides st;
    st.idx = (size_t)B; st.wpack = durf_wpack_bwd_bytes(DURF_W_OBJ); st.mask = durf_mlp_mask_bytes(rows);
    st.dz = durf_mlp_stash_bytes(DURF_W_OBJ, rows); st.dz_out = durf_obj_dzout_stride(B, N);
    st.d_enc = rows * DURF_ENC_DIM * sizeof(float);
    return durf::launch_mlp_bwd(stream, DURF_W_OBJ, rows, N, draw, idx, count, wpack_bwd, relu_mask, dz, dz_out, d_enc,
                                K, st);
}

int durf_obj_bwd_batch_levels(void* stream, int K, int B, int N, int nlevels, const int32_t* idx, const int32_t* count,
                              const float* const* draw, const void* wpack_bwd, const void* const* relu_mask, void* const* dz,
                              void* const* dz_out) {
    DURF_REQUIRE(nlevels >= 1 && nlevels <= DURF_MAX_LEVELS, "1 <= nlevels <= DURF_MAX_LEVELS");
    const size_t rows = (size_t)B * N;
    if (!durf::obj_msplit(rows)) {          // large batches: the sample-split kernel, level by level (as durf_obj_bwd_batch)
        for (int l = 0; l < nlevels; l++) {
            const int rc = durf_obj_bwd_batch(stream, K, B, N, idx, count, draw[l], wpack_bwd, relu_mask[l], dz[l], dz_out[l], nullptr);
            if (rc) return rc;
        }
        return 0;
    }
    BwdStrides st;
    st.idx = (size_t)B; st.wpack = durf_wpack_bwd_bytes(DURF_W_OBJ); st.mask = durf_mlp_mask_bytes(rows);
    st.dz = durf_mlp_stash_bytes(DURF_W_OBJ, rows); st.dz_out = durf_obj_dzout_stride(B, N);
    st.d_enc = rows * DURF_ENC_DIM * sizeof(float);
    return durf::launch_mlp_bwd_ms_levels(stream, rows, N, nlevels, draw, idx, count, wpack_bwd, relu_mask, dz, dz_out, K, st);
}

static DwStrides obj_dw_strides(int B, int N) {
    const size_t rows = (size_t)B * N;
    DwStrides st;
    st.enc = durf_obj_enc_stride(B, N); st.view = durf_obj_view_stride(B, N);
    st.stash = durf_mlp_stash_bytes(DURF_W_OBJ, rows); st.dz_out = durf_obj_dzout_stride(B, N);
    st.part = durf_dw_part_floats(DURF_W_OBJ); st.bpart = durf_dw_bpart_floats(DURF_W_OBJ);
    return st;
}

int durf_obj_dw_partials(void* stream, int K, int B, int N, const int32_t* count, int nlevels,
                         const void* const* enc, const void* const* view_tile, const void* const* stash,
                         const void* const* dz, const void* const* dz_out, float* part, float* bpart) {
    DURF_REQUIRE(nlevels >= 1 && nlevels <= DURF_MAX_LEVELS, "1 <= nlevels <= DURF_MAX_LEVELS");
    const durf::DwLevels lv = durf::uniform_levels((size_t)B * N, N, count, nlevels);
    return durf::launch_mlp_dw(stream, DURF_W_OBJ, lv, enc, view_tile, stash, dz, dz_out, part, bpart, K, obj_dw_strides(B, N));
}

int durf_obj_dw_batch(void* stream, int K, int B, int N, const int32_t* count, int nlevels,
                      const void* const* enc, const void* const* view_tile, const void* const* stash,
                      const void* const* dz, const void* const* dz_out, int in_dim, float* part, float* bpart,
                      float* grad_mlp, size_t grad_stride, const float* mlp_params) {
    int rc = durf_obj_dw_partials(stream, K, B, N, count, nlevels, enc, view_tile, stash, dz, dz_out, part, bpart);
    if (rc) return rc;
    const DwStrides st = obj_dw_strides(B, N);
    return durf::launch_dw_finalize(stream, DURF_W_OBJ, in_dim, durf::uniform_levels((size_t)B * N, N, count, nlevels), part, bpart,
                                    grad_mlp, K, st.part, st.bpart, grad_stride, mlp_params, grad_stride);
}

int durf_dw_finalize_all(void* stream, int in_bkgd, int nseg, const size_t* rows, const int* rows_per_ray,
                         const int32_t* const* seg_count, const float* part_bkgd, const float* bpart_bkgd,
                         float* grad_bkgd, const float* bkgd_params, int K, int B, int N, const int32_t* obj_count,
                         int nlevels, int in_obj, const float* part_obj, const float* bpart_obj, float* grad_obj,
                         size_t obj_grad_stride, const float* obj_params) {
    durf::DwFinSpec a, b;
    a.width = 256; a.in_dim = in_bkgd; a.K = 1;
    DURF_REQUIRE(nseg >= 1 && nseg <= DURF_MAX_LEVELS, "1 <= segments <= DURF_MAX_LEVELS");
    a.lv.nlevels = nseg;
    for (int l = 0; l < DURF_MAX_LEVELS; l++) {
        const int ll = l < nseg ? l : 0;
        DURF_REQUIRE(rows_per_ray[ll] >= 1, "rows_per_ray >= 1");
        a.lv.rows[l] = rows[ll]; a.lv.n[l] = rows_per_ray[ll]; a.lv.count[l] = seg_count ? seg_count[ll] : nullptr;
    }
    a.part = part_bkgd; a.bpart = bpart_bkgd; a.grad = grad_bkgd; a.params = bkgd_params;
    a.part_stride = a.bpart_stride = a.grad_stride = a.param_stride = 0;
    b.width = DURF_W_OBJ; b.in_dim = in_obj; b.K = K;
    if (K > 0) {
        DURF_REQUIRE(nlevels >= 1 && nlevels <= DURF_MAX_LEVELS, "1 <= nlevels <= DURF_MAX_LEVELS");
        b.lv = durf::uniform_levels((size_t)B * N, N, obj_count, nlevels);
        b.part = part_obj; b.bpart = bpart_obj; b.grad = grad_obj; b.params = obj_params;
        b.part_stride = durf_dw_part_floats(DURF_W_OBJ); b.bpart_stride = durf_dw_bpart_floats(DURF_W_OBJ);
        b.grad_stride = b.param_stride = obj_grad_stride;
    }
    return durf::launch_dw_finalize2(stream, a, b);
}

}  // extern "C"
