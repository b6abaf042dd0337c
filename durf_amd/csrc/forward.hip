// durf_forward: MipNerfModel.__call__ in inference (obbpose_model.py:68-261 as render_eval_fn runs it, train_boxpose.py:377-390)
// as ONE C call -- the orchestration durf_amd/obbpose_model.py does in Python for `train=False`, for hosts that are not Python
// (SURVEY 8b proposed it; INTEGRATION.md shows the binding).  No kernel of its own: the stage entry points of this library
// in the order the Python path issues them, on one stream, with every intermediate carved out of a caller-owned workspace.
// bf16 MLPs; rays that hit exactly one box take the de-duplicated background evaluation (durf_expand_raw); K = 0 is the
// static model.  Results are bit-identical to MipNerfModel.apply (tests/test_gpu_forward_call.py).
#include "durf_common.h"
#include "side_stream.h"
#include "../../include/durf_hip.h"

namespace {

struct Carver {                      // 256-byte aligned sub-buffers of the workspace (or just their total with base == null)
    char* base;
    size_t off;
    void* take(size_t bytes) {
        off = (off + 255) & ~(size_t)255;
        void* p = base ? base + off : nullptr;
        off += bytes;
        return p;
    }
};

struct FwdWs {
    float *o_s, *d_s, *raw_b, *obj_raw, *t_scratch, *u_rand;
    int32_t *hit, *idx_obj, *count_obj, *slot_obj, *idx_cls, *count_cls, *slot_cls;
    void *view, *wf_bkgd, *wf_obj, *enc, *obj_enc;
    size_t total;
};

FwdWs carve(void* workspace, int B, int N, int K) {
    Carver c{(char*)workspace, 0};
    FwdWs w{};
    const size_t rows = (size_t)B * N, Kc = K > 0 ? K : 1;
    w.o_s = (float*)c.take((size_t)B * 3 * 4);
    w.d_s = (float*)c.take((size_t)B * 3 * 4);
    w.hit = (int32_t*)c.take((size_t)B * Kc * 4);
    w.view = c.take((size_t)B * 32 * 2);
    w.idx_obj = (int32_t*)c.take(Kc * B * 4);
    w.count_obj = (int32_t*)c.take(Kc * 4);
    w.slot_obj = (int32_t*)c.take((size_t)B * Kc * 4);
    w.idx_cls = (int32_t*)c.take((size_t)2 * B * 4);
    w.count_cls = (int32_t*)c.take(8 * 4);
    w.slot_cls = (int32_t*)c.take((size_t)2 * B * 4);
    w.wf_bkgd = c.take(durf_wpack_fwd_bytes(256));
    w.wf_obj = c.take(Kc * durf_wpack_fwd_bytes(128));
    w.enc = c.take(((rows + 31) / 32 * 32) * 64 * 2);
    w.raw_b = (float*)c.take(rows * 4 * 4);
    w.obj_enc = c.take(K > 0 ? (size_t)K * durf_obj_enc_stride(B, N) : 0);
    w.obj_raw = (float*)c.take(K > 0 ? (size_t)K * rows * 4 * 4 : 0);
    w.u_rand = (float*)c.take((size_t)B * (N + 1) * 4);              // draw_noise: the resampling draws of the prologue
    w.total = (c.off + 255) & ~(size_t)255;
    return w;
}

}  // namespace

extern "C" {

size_t durf_forward_workspace_bytes(int B, int N, int K) { return carve(nullptr, B, N, K).total; }

int durf_forward(void* stream, const durf_forward_args* a, void* workspace, size_t workspace_bytes) {
    DURF_REQUIRE(a != nullptr && workspace != nullptr, "arguments and workspace");
    const int B = a->B, N = a->N, K = a->K, L = a->num_levels;
    DURF_REQUIRE(B > 0 && N % 32 == 0 && N >= 32 && N <= 256, "B > 0, num_samples a multiple of 32 in [32, 256]");
    DURF_REQUIRE(K >= 0 && K <= DURF_MAX_OBJ, "0 <= K <= DURF_MAX_OBJ");
    DURF_REQUIRE(L >= 1 && L <= DURF_FORWARD_MAX_LEVELS, "1 <= num_levels <= DURF_FORWARD_MAX_LEVELS");
    DURF_REQUIRE(((size_t)workspace & 255) == 0, "workspace aligned to 256 bytes");
    DURF_REQUIRE(!a->draw_noise || (a->t_rand == nullptr && a->u_rand == nullptr), "draw_noise: the library makes the draws");
    for (int l = 0; l < L && a->density_noise != 0.0f; l++)
        DURF_REQUIRE(a->density_rand[l] != nullptr || a->draw_noise, "density_noise: density_rand[level] or draw_noise");
    const FwdWs w = carve(workspace, B, N, K);
    if (workspace_bytes < w.total) {
        durf_set_error("durf_forward: workspace of %zu bytes, durf_forward_workspace_bytes(%d, %d, %d) = %zu", workspace_bytes, B, N, K, w.total);
        return -1;
    }
    const size_t rows = (size_t)B * N;
    int rc;
#define STEP(call) do { rc = (call); if (rc != 0) return rc; } while (0)
    // ray setup + view encoding + level-0 sample positions (obbpose_model.py:99-131, mip.py:330-370) + the bf16 weight
    // streams of every MLP: one launch
    STEP(durf_ray_prologue_pack(stream, B, K, N, a->origins, a->directions, a->pose, a->ext, w.o_s, w.d_s, w.hit, a->zo, a->viewdirs,
                                w.view, a->near, a->far, a->t_rand, a->lindisp, a->t_vals[0], nullptr, nullptr, 0, a->seed_lo, a->seed_hi,
                                a->draw_noise ? w.u_rand : nullptr, a->bkgd_params, 60, w.wf_bkgd, nullptr, K, a->obj_params,
                                a->obj_param_stride, 63, w.wf_obj, nullptr, K == 0 ? (float*)a->dyn_mask : nullptr, K == 0 ? (size_t)B : 0));
    if (K > 0)      // per-object hit lists + the ray classes of the de-duplicated background evaluation: one launch
        STEP(durf_compact_all(stream, B, K, N, w.hit, w.idx_obj, w.count_obj, w.slot_obj, w.idx_cls, w.count_cls, w.slot_cls,
                              a->dyn_mask));
    const float* raw_obj[DURF_MAX_OBJ > 0 ? DURF_MAX_OBJ : 1];
    for (int k = 0; k < K; k++) raw_obj[k] = w.obj_raw + (size_t)k * rows * 4;
    // (a large chunk's object MLPs on the library's side stream, issued before the persistent background launch: side_stream.h)
    const durf::Overlap ov = durf::overlap_for(stream, rows, K);
    for (int lvl = 0; lvl < L; lvl++) {
        float* t_vals = a->t_vals[lvl];
        if (K > 0) {
            STEP(ov.fork());
            if (ov.sd)
                STEP(durf_obj_fwd_batch(ov.obj(), K, B, N, w.idx_obj, w.count_obj, t_vals, w.o_s, w.d_s, a->radii, a->barf_w,
                                        a->enc_flags & (DURF_ENC_NO_INTEGRATION | DURF_ENC_CYLINDER), w.view, w.wf_obj, w.obj_enc,
                                        w.obj_raw, nullptr, nullptr, nullptr));
            STEP(durf_mlp_fwd_enc(stream, rows, N, t_vals, w.o_s, w.d_s, a->radii, w.hit, K, a->enc_flags | DURF_FWD_RAW_FULL, w.enc, w.view, w.idx_cls, w.count_cls,
                                  w.wf_bkgd, w.raw_b, nullptr, nullptr, w.idx_cls + B, w.count_cls + 1, nullptr));
            if (!ov.sd)
                STEP(durf_obj_fwd_batch(stream, K, B, N, w.idx_obj, w.count_obj, t_vals, w.o_s, w.d_s, a->radii, a->barf_w,
                                        a->enc_flags & (DURF_ENC_NO_INTEGRATION | DURF_ENC_CYLINDER), w.view, w.wf_obj, w.obj_enc,
                                        w.obj_raw, nullptr, nullptr, nullptr));
            STEP(ov.join());
        } else {
            STEP(durf_mlp_fwd_enc(stream, rows, N, t_vals, w.o_s, w.d_s, a->radii, nullptr, 0, a->enc_flags, w.enc, w.view, nullptr, nullptr, w.wf_bkgd, w.raw_b, nullptr, nullptr, nullptr,
                              nullptr, nullptr));
        }
        if (a->density_noise != 0.0f)      // obbpose_model.py:236-240
            STEP(durf_density_noise(stream, rows, w.raw_b, a->density_noise, a->density_rand[lvl], a->seed_lo, a->seed_hi, lvl));
        STEP(durf_composite_fwd(stream, B, N, K, w.raw_b, raw_obj, w.slot_obj, t_vals, w.d_s, a->density_bias, a->bkgd_mode,
                                a->rgb[lvl], a->depth[lvl], a->acc[lvl], a->weights[lvl], a->t_mids[lvl], a->t_dists[lvl]));
        if (lvl + 1 < L)
            STEP(durf_resample(stream, B, N, t_vals, a->weights[lvl], a->resample_padding, a->draw_noise ? w.u_rand : a->u_rand, a->t_vals[lvl + 1]));
    }
#undef STEP
    return 0;
}

}  // extern "C"
