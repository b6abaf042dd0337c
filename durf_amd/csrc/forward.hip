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
    w.u_rand = (float*)c.take((size_t)3 * B * (N + 1) * 4);              // draw_noise: the resampling draws of the prologue
    w.total = (c.off + 255) & ~(size_t)255;
    return w;
}

}  // namespace

extern "C" {

size_t durf_forward_workspace_bytes(int B, int N, int K) { return carve(nullptr, B, N, K).total; }

static int check_forward_args(const durf_forward_args* a, void* workspace) {
    DURF_REQUIRE(a != nullptr && workspace != nullptr, "arguments and workspace");
    const int B = a->B, N = a->N, K = a->K, L = a->num_levels;
    DURF_REQUIRE(B > 0 && N % 32 == 0 && N >= 32 && N <= 256, "B > 0, num_samples a multiple of 32 in [32, 256]");
    DURF_REQUIRE(K >= 0 && K <= DURF_MAX_OBJ, "0 <= K <= DURF_MAX_OBJ");
    DURF_REQUIRE(L >= 1 && L <= DURF_FORWARD_MAX_LEVELS, "1 <= num_levels <= DURF_FORWARD_MAX_LEVELS");
    DURF_REQUIRE(((size_t)workspace & 255) == 0, "workspace aligned to 256 bytes");
    DURF_REQUIRE(!a->draw_noise || (a->t_rand == nullptr && a->u_rand == nullptr), "draw_noise: the library makes the draws");
    for (int l = 0; l < L && a->density_noise != 0.0f; l++)
        DURF_REQUIRE(a->density_rand[l] != nullptr || a->draw_noise, "density_noise: density_rand[level] or draw_noise");
    return 0;
}

// the launches of one chunk (arguments checked, workspace carved by the caller)
static int forward_launches(void* stream, const durf_forward_args* a, const FwdWs& w) {
    const int B = a->B, N = a->N, K = a->K, L = a->num_levels;
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
            STEP(durf_resample(stream, B, N, t_vals, a->weights[lvl], a->resample_padding, a->draw_noise ? w.u_rand + (size_t)lvl * B * (N + 1) : a->u_rand, a->t_vals[lvl + 1]));
    }
#undef STEP
    return 0;
}

int durf_forward(void* stream, const durf_forward_args* a, void* workspace, size_t workspace_bytes) {
    int rc = check_forward_args(a, workspace);
    if (rc != 0) return rc;
    const FwdWs w = carve(workspace, a->B, a->N, a->K);
    if (workspace_bytes < w.total) {
        durf_set_error("durf_forward: workspace of %zu bytes, durf_forward_workspace_bytes(%d, %d, %d) = %zu", workspace_bytes, a->B, a->N,
                       a->K, w.total);
        return -1;
    }
    return forward_launches(stream, a, w);
}

// ---- one C call per IMAGE (obbpose_model.py:421-479 render_image on one device) --------------------------------------
// The reference walks the image in chunks from Python, one pmapped call and one host round trip per chunk; here the chunk
// loop is this function: the rays of the whole image stay where they are on the device, every chunk runs durf_forward's
// launch sequence on slices of them, the levels' per-chunk outputs live in the workspace, and the LAST level's rgb /
// distance / acc land in the image planes in place -- what render_image returns (:476-479).
namespace {
struct ImgWs { FwdWs f; float *rgb[DURF_FORWARD_MAX_LEVELS], *depth[DURF_FORWARD_MAX_LEVELS], *acc[DURF_FORWARD_MAX_LEVELS],
               *weights[DURF_FORWARD_MAX_LEVELS], *t_vals[DURF_FORWARD_MAX_LEVELS], *t_mids[DURF_FORWARD_MAX_LEVELS],
               *t_dists[DURF_FORWARD_MAX_LEVELS], *zo; int32_t* dyn; size_t total; };
ImgWs carve_image(void* workspace, int chunk, int N, int K, int L) {
    ImgWs w{};
    w.f = carve(workspace, chunk, N, K);
    Carver c{(char*)workspace, w.f.total};
    for (int l = 0; l < L; l++) {
        w.rgb[l] = (float*)c.take((size_t)chunk * 3 * 4); w.depth[l] = (float*)c.take((size_t)chunk * 4);
        w.acc[l] = (float*)c.take((size_t)chunk * 4); w.weights[l] = (float*)c.take((size_t)chunk * N * 4);
        w.t_vals[l] = (float*)c.take((size_t)chunk * (N + 1) * 4); w.t_mids[l] = (float*)c.take((size_t)chunk * N * 4);
        w.t_dists[l] = (float*)c.take((size_t)chunk * N * 4);
    }
    w.zo = (float*)c.take((size_t)chunk * 4);
    w.dyn = (int32_t*)c.take((size_t)chunk * 4);
    w.total = (c.off + 255) & ~(size_t)255;
    return w;
}
}  // namespace

size_t durf_render_image_workspace_bytes(int chunk, int N, int K, int num_levels) {
    return carve_image(nullptr, chunk, N, K, num_levels).total;
}

int durf_render_image(void* stream, const durf_forward_args* a, size_t n_rays, int chunk, float* rgb, float* distance, float* acc,
                      void* workspace, size_t workspace_bytes) {
    DURF_REQUIRE(a != nullptr && rgb && distance && acc, "arguments and the three image planes");
    DURF_REQUIRE(chunk > 0 && n_rays > 0, "chunk > 0, n_rays > 0");
    DURF_REQUIRE(a->t_rand == nullptr && a->u_rand == nullptr && !a->draw_noise && a->density_noise == 0.0f,
                 "render_image is test mode: randomized = False (obbpose_model.py:421-479)");
    const int L = a->num_levels;
    DURF_REQUIRE(L >= 1 && L <= DURF_FORWARD_MAX_LEVELS, "1 <= num_levels <= DURF_FORWARD_MAX_LEVELS");
    const ImgWs w = carve_image(workspace, chunk, a->N, a->K, L);
    if (workspace_bytes < w.total) {
        durf_set_error("durf_render_image: workspace of %zu bytes, durf_render_image_workspace_bytes(%d, %d, %d, %d) = %zu",
                       workspace_bytes, chunk, a->N, a->K, L, w.total);
        return -1;
    }
    for (size_t i = 0; i < n_rays; i += (size_t)chunk) {
        durf_forward_args c = *a;
        c.B = (int)(n_rays - i < (size_t)chunk ? n_rays - i : (size_t)chunk);        // (the last chunk is the remainder, :451-453)
        c.origins = a->origins + i * 3; c.directions = a->directions + i * 3; c.viewdirs = a->viewdirs + i * 3;
        c.radii = a->radii + i; c.near = a->near + i; c.far = a->far + i;
        for (int l = 0; l < L; l++) {
            const bool last = l == L - 1;
            c.rgb[l] = last ? rgb + i * 3 : w.rgb[l]; c.depth[l] = last ? distance + i : w.depth[l]; c.acc[l] = last ? acc + i : w.acc[l];
            c.weights[l] = w.weights[l]; c.t_vals[l] = w.t_vals[l]; c.t_mids[l] = w.t_mids[l]; c.t_dists[l] = w.t_dists[l];
        }
        c.zo = w.zo; c.dyn_mask = w.dyn;
        int rc = check_forward_args(&c, workspace);
        if (rc != 0) return rc;
        rc = forward_launches(stream, &c, w.f);
        if (rc != 0) return rc;
    }
    return 0;
}

}  // extern "C"
