// durf_loss_backward / durf_train_step: value_and_grad(loss_fn) of train_step (train_boxpose.py:49-321) for one shard as ONE
// C call -- the orchestration durf_amd/train_boxpose.py (loss_and_grad, train_step) and obbpose_model.py (_forward, train)
// do in Python, for hosts that are not Python (durf_forward / durf_loss_backward / durf_clip_adam).  No kernel
// of its own: the stage entry points of this library in the order the Python path issues them on ONE stream, every
// intermediate carved out of a caller-owned workspace.  Scope: every BASELINE.json training configuration -- bf16 background
// MLP; object MLPs on the bf16 kernels (frozen poses: cfg2 / cfg3 / cfg5) or, with obj_fp32, on the exact-fp32 kernels with
// the box-pose gradient behind them (cfg4: want_pos / want_rot, the TV prior); >= 2 levels; density noise (f.density_noise),
// weight decay (weight_decay_mult) and every background mode included.  Results are bit-identical to
// train_boxpose.train_step (tests/test_gpu_train_call.py).
#include "durf_common.h"
#include "../../include/durf_hip.h"
#include "side_stream.h"

namespace {

struct Carver {
    char* base;
    size_t off;
    void* take(size_t bytes) {
        // (large buffers start on 2 MB boundaries, as the host allocator's own blocks do: the forward ran 2-3 % slower on
        // streams that started at arbitrary 256-byte offsets of one big block)
        const size_t al = bytes >= ((size_t)1 << 20) ? ((size_t)2 << 20) : (size_t)256;
        off = (off + al - 1) & ~(al - 1);
        void* p = base ? base + off : nullptr;
        off += bytes;
        return p;
    }
};

constexpr int ML = DURF_FORWARD_MAX_LEVELS;
constexpr int OBJ32_NSPLIT = 8;        // ops.objf32_dw_batch's default: same split-K partial order as the Python path

// (the one kernel of this file) TV prior on the box positions (train_boxpose.py:136,219) added to this timestep's rows of the
// gradient: g[k, 0:3] += c * (pose[k, 0:3] - prev[k, 0:3]), in the operation order of the Python path's tensor expression
__global__ void k_tv_rows(int K, const float* __restrict__ pose, const float* __restrict__ prev, float c, float* __restrict__ g) {
    const int i = threadIdx.x, k = i / 3, q = i % 3;
    if (k >= K) return;
    const float t = pose[k * 6 + q] - prev[k * 6 + q];
    const float t2 = c * t;
    g[k * 6 + q] = g[k * 6 + q] + t2;
}

struct TrainWs {
    float *o_s, *d_s, *norms, *prep, *ray_sums, *sums, *part, *bpart, *opart, *obpart, *scratch, *u_rand, *weight_l2;
    float* draw[ML];      // d(loss)/d(raw) of every level (ONE loss launch fills them all: durf_loss_bwd_levels)
    float *raw_c[ML], *raw_b[ML], *obj_raw[ML], *terms[ML];
    int32_t *hit, *idx_obj, *count_obj, *slot_obj, *idx_cls, *count_cls, *slot_cls;
    void *view, *wf_bkgd, *wb_bkgd, *wf_obj, *wb_obj, *view_tile, *obj_view_tile;
    void *enc[ML], *stash[ML], *mask[ML], *dz[ML], *dz_out[ML];
    void *obj_enc[ML], *obj_stash[ML], *obj_mask[ML], *obj_dz[ML], *obj_dz_out[ML];
    // object branch on the exact-fp32 kernels (obj_fp32): view features, the background's fp32 evaluation of the box-hit
    // rays, weight streams, per-level records, the pose sums
    float *view27, *trunk, *raw_tail, *obj_ws, *dw32_scratch, *pose_sums, *pose_scratch;
    float *act32[ML], *dz32[ML], *d_enc32[ML];
    size_t total;
};

TrainWs carve(void* workspace, int B, int N, int K, int L, size_t n_params, int flags = 0) {
    const bool f32o = K > 0 && (flags & DURF_TRAIN_OBJ_FP32) != 0, pose = f32o && (flags & DURF_TRAIN_POSE_OPT) != 0;
    Carver c{(char*)workspace, 0};
    TrainWs w{};
    const size_t rows = (size_t)B * N, Kc = K > 0 ? K : 1, trows = (rows + 31) / 32 * 32;
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
    w.wb_bkgd = c.take(durf_wpack_bwd_bytes(256));
    w.wf_obj = c.take(Kc * durf_wpack_fwd_bytes(128));
    w.wb_obj = c.take(Kc * durf_wpack_bwd_bytes(128));
    w.view_tile = c.take(trows * 32 * 2);
    w.obj_view_tile = c.take(K > 0 ? (size_t)K * durf_obj_view_stride(B, N) : 0);
    w.norms = (float*)c.take((size_t)L * 5 * 4);
    w.prep = (float*)c.take((size_t)2 * 5 * B * 4);
    w.ray_sums = (float*)c.take((size_t)L * B * 4 * 4);
    w.sums = (float*)c.take((size_t)L * 7 * 4);
    for (int l = 0; l < L; l++) w.draw[l] = (float*)c.take(rows * 4 * 4);
    w.part = (float*)c.take(durf_dw_part_floats(256) * 4);
    w.bpart = (float*)c.take(durf_dw_bpart_floats(256) * 4);
    w.opart = (float*)c.take(K > 0 ? (size_t)K * durf_dw_part_floats(128) * 4 : 0);
    w.obpart = (float*)c.take(K > 0 ? (size_t)K * durf_dw_bpart_floats(128) * 4 : 0);
    w.scratch = (float*)c.take(durf_optim_scratch_floats(n_params) * 4);
    w.u_rand = (float*)c.take((size_t)3 * B * (N + 1) * 4);              // f.draw_noise: the resampling draws of the prologue
    w.weight_l2 = (float*)c.take(4);                                 // weight_decay_mult * mean(theta^2) (durf_weight_decay)
    for (int l = 0; l < L; l++) {
        w.terms[l] = (float*)c.take((size_t)7 * B * 4);
        w.enc[l] = c.take(trows * 64 * 2);
        w.raw_c[l] = (float*)c.take(rows * 4 * 4);
        w.raw_b[l] = (float*)c.take(rows * 4 * 4);
        w.stash[l] = c.take(durf_mlp_stash_bytes(256, rows));
        w.mask[l] = c.take(durf_mlp_mask_bytes(rows));
        w.dz[l] = c.take(durf_mlp_stash_bytes(256, rows));
        w.dz_out[l] = c.take(trows * 16 * 2);
        if (K > 0 && !f32o) {
            w.obj_enc[l] = c.take((size_t)K * durf_obj_enc_stride(B, N));
            w.obj_raw[l] = (float*)c.take((size_t)K * rows * 4 * 4);
            w.obj_stash[l] = c.take((size_t)K * durf_mlp_stash_bytes(128, rows));
            w.obj_mask[l] = c.take((size_t)K * durf_mlp_mask_bytes(rows));
            w.obj_dz[l] = c.take((size_t)K * durf_mlp_stash_bytes(128, rows));
            w.obj_dz_out[l] = c.take((size_t)K * durf_obj_dzout_stride(B, N));
        } else if (f32o) {
            w.obj_raw[l] = (float*)c.take((size_t)K * rows * 4 * 4);
            w.act32[l] = (float*)c.take((size_t)K * durf_objf32_act_stride(B, N) * 4);
            w.dz32[l] = (float*)c.take((size_t)K * durf_objf32_dz_stride(B, N) * 4);
            w.d_enc32[l] = (float*)c.take(pose ? (size_t)K * rows * 64 * 4 : 0);
        }
    }
    if (f32o) {
        w.view27 = (float*)c.take((size_t)B * 27 * 4);
        w.trunk = (float*)c.take(264 * 4);
        w.raw_tail = (float*)c.take((size_t)B * 4 * 4);
        w.obj_ws = (float*)c.take((size_t)K * durf_mlp_f32_wstream_floats(128) * 4);
        w.dw32_scratch = (float*)c.take((size_t)K * durf_mlp_f32_dw_scratch_floats(128, 63, OBJ32_NSPLIT) * 4);
        w.pose_sums = (float*)c.take((size_t)K * 21 * 4);
        w.pose_scratch = (float*)c.take(pose ? (size_t)L * K * 21 * B * 4 : 0);      // (every level's rows: one launch pair)
    }
    w.total = (c.off + 255) & ~(size_t)255;
    return w;
}

// the workspace of one call; the class counts live in the caller's buffer when it asked for them (durf_train_args.cls_count)
TrainWs workspace_of(const durf_train_args* a, void* workspace) {
    TrainWs w = carve(workspace, a->f.B, a->f.N, a->f.K, a->f.num_levels, a->n_params, a->flags);
    if (a->cls_count != nullptr) w.count_cls = a->cls_count;
    return w;
}

// a workspace the caller sized for another shape is refused, not overrun (the intermediates of a step are up to ~10 GB)
int check_workspace(const char* who, const durf_train_args* a, const TrainWs& w, size_t workspace_bytes) {
    if (workspace_bytes >= w.total) return 0;
    durf_set_error("%s: workspace of %zu bytes, durf_train_workspace_bytes_flags(%d, %d, %d, %d, %zu, %d) = %zu", who, workspace_bytes,
                   a->f.B, a->f.N, a->f.K, a->f.num_levels, a->n_params, a->flags, w.total);
    return -1;
}

int check_args(const durf_train_args* a, void* workspace) {
    DURF_REQUIRE(a != nullptr && workspace != nullptr, "arguments and workspace");
    const durf_forward_args& f = a->f;
    DURF_REQUIRE(f.B > 0 && f.N % 32 == 0 && f.N >= 32 && f.N <= 256, "B > 0, num_samples a multiple of 32 in [32, 256]");
    DURF_REQUIRE(f.K >= 0 && f.K <= DURF_MAX_OBJ, "0 <= K <= DURF_MAX_OBJ");
    DURF_REQUIRE(f.num_levels >= 2 && f.num_levels <= DURF_FORWARD_MAX_LEVELS, "2 <= num_levels <= DURF_FORWARD_MAX_LEVELS");
    DURF_REQUIRE(((size_t)workspace & 255) == 0, "workspace aligned to 256 bytes");
    DURF_REQUIRE(a->box_floats + a->mlp0_floats + (size_t)f.K * a->obj_floats == a->n_params,
                 "flat layout: box_centers | MLP_0 | K object MLPs");
    DURF_REQUIRE(f.bkgd_params == a->params + a->box_floats &&
                 (f.K == 0 || (f.obj_params == a->params + a->box_floats + a->mlp0_floats && f.obj_param_stride == a->obj_floats)),
                 "f.bkgd_params / f.obj_params point into params");
    DURF_REQUIRE(f.bkgd_mode >= 0 && f.bkgd_mode <= 2, "bkgd_mode: 0 grey / 1 white / 2 none (rand_bkgd: bg = 0)");
    for (int l = 0; l < f.num_levels && f.density_noise != 0.0f; l++)
        DURF_REQUIRE(f.density_rand[l] != nullptr || f.draw_noise, "density_noise: density_rand[level] or draw_noise");
    DURF_REQUIRE(!f.draw_noise || (f.t_rand == nullptr && f.u_rand == nullptr), "draw_noise: the library makes the draws");
    DURF_REQUIRE((a->flags & ~(DURF_TRAIN_OBJ_FP32 | DURF_TRAIN_POSE_OPT | DURF_TRAIN_OBJ_X3)) == 0, "unknown flags");
    DURF_REQUIRE(!(a->flags & DURF_TRAIN_OBJ_X3) || (a->flags & DURF_TRAIN_OBJ_FP32), "DURF_TRAIN_OBJ_X3 is a variant of the fp32 object branch");
    if (a->flags & DURF_TRAIN_POSE_OPT) {
        DURF_REQUIRE(f.K > 0 && (a->flags & DURF_TRAIN_OBJ_FP32), "box-pose optimisation runs behind the fp32 object branch");
        DURF_REQUIRE(a->want_pos || a->want_rot, "pose optimisation without a pose gradient to compute");
        DURF_REQUIRE(f.pose >= a->params && f.pose + (size_t)f.K * 6 <= a->params + a->box_floats,
                     "f.pose must be this timestep's rows of box_centers inside params (their gradient goes to the same rows of grad)");
        DURF_REQUIRE(a->prev6 != nullptr || a->tv_loss_mult == 0.0f, "the TV prior needs prev6");
    }
    DURF_REQUIRE(a->params && a->grad && a->stats, "params, grad and stats buffers");
    DURF_REQUIRE(((size_t)a->grad & 15) == 0, "grad aligned to 16 bytes");
    for (int l = 0; l < f.num_levels; l++)
        DURF_REQUIRE(f.t_vals[l] && f.weights[l] && f.rgb[l] && f.depth[l] && f.acc[l] && f.t_mids[l] && f.t_dists[l],
                     "per-level output buffers (t_vals, rgb, depth, acc, weights, t_mids, t_dists)");
    return 0;
}

#define STEP(call) do { rc = (call); if (rc != 0) return rc; } while (0)
// a launch bracketed by the caller's timing events (durf_train_args.timing), recorded on the stream it is issued to
#define TIMED(id, call) do {                                                                       \
        if (tm && tm->begin[id]) (void)hipEventRecord((hipEvent_t)tm->begin[id], hs);               \
        STEP(call);                                                                                \
        if (tm && tm->end[id]) (void)hipEventRecord((hipEvent_t)tm->end[id], hs);                   \
    } while (0)

__global__ void k_scale(int n, float* __restrict__ x, float c) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] *= c;
}

// tail: the multi-hit outcome and the logged scalars as launches of their own (durf_loss_backward: the caller gets the
// un-post-processed gradient); durf_train_step folds both into the optimizer's first launch (durf_stats_scrub)
int loss_backward(void* stream, const durf_train_args* a, const TrainWs& w, bool tail = true) {
    const durf_forward_args& f = a->f;
    const int B = f.B, N = f.N, K = f.K, L = f.num_levels;
    const size_t rows = (size_t)B * N;
    hipStream_t hs = (hipStream_t)stream;
    const durf_step_timing* tm = a->timing;
    int rc;
    // ---- forward (obbpose_model.py:68-261), activations stashed ----
    const bool f32o = K > 0 && (a->flags & DURF_TRAIN_OBJ_FP32), pose_opt = f32o && (a->flags & DURF_TRAIN_POSE_OPT);
    const bool x3 = f32o && (a->flags & DURF_TRAIN_OBJ_X3);      // ... their forward / backward GEMMs on split bf16 operands
    const int Kb = f32o ? 0 : K;                          // objects on the bf16 kernels
    // (ray setup, view encoding, level-0 samples, the step's draws, the gradient's zero fill AND every bf16 weight stream: one launch)
    STEP(durf_ray_prologue_pack(stream, B, K, N, f.origins, f.directions, f.pose, f.ext, w.o_s, w.d_s, w.hit, f.zo, f.viewdirs, w.view,
                                f.near, f.far, f.t_rand, f.lindisp, f.t_vals[0], K > 0 ? a->pose_used : nullptr, a->grad, a->n_params, f.seed_lo, f.seed_hi,
                                f.draw_noise ? w.u_rand : nullptr, f.bkgd_params, 60, w.wf_bkgd, w.wb_bkgd, Kb, f.obj_params,
                                f.obj_param_stride, 63, w.wf_obj, w.wb_obj,
                                // (zero filled on the way: dyn_mask of a model without boxes, or the pose sums)
                                K == 0 ? (float*)f.dyn_mask : (pose_opt ? w.pose_sums : nullptr),
                                K == 0 ? (size_t)B : (pose_opt ? (size_t)K * 21 : (size_t)0)));
    if (K > 0)
        STEP(durf_compact_all(stream, B, K, N, w.hit, w.idx_obj, w.count_obj, w.slot_obj, w.idx_cls, w.count_cls, w.slot_cls,
                              f.dyn_mask));
    if (f32o) {
        // the object branch in exact fp32 (MipNerfModel.object_precision, obbpose_model._forward): fp32 view features and
        // weight streams, and the background MLP's ONE evaluation of every box-hit ray redone in fp32 -- the constant
        // trunk once, the view layer + rgb head per ray (the Python path runs these beside the prologue on a side stream)
        STEP(durf_view_enc(stream, B, f.viewdirs, nullptr, w.view27));
        if (x3) STEP(durf_mlp_f32_pack_x3(stream, K, f.obj_params, f.obj_param_stride, w.obj_ws));
        else STEP(durf_mlp_f32_pack(stream, 128, 63, K, f.obj_params, f.obj_param_stride, w.obj_ws));
        // (the constant trunk + the box-hit rays' view layer and rgb head: behind the level-0 forward, below)
    }
    const float* raw_obj[ML][DURF_MAX_OBJ];
    for (int l = 0; l < L; l++)
        for (int k = 0; k < K; k++) raw_obj[l][k] = w.obj_raw[l] + (size_t)k * rows * 4;
    const int obj_flags = f.enc_flags & (DURF_ENC_NO_INTEGRATION | DURF_ENC_CYLINDER);
    const durf::Overlap ov = durf::overlap_for(stream, rows, Kb);
    // Small steps (one stream): the bf16 object MLPs' forward and backward are items of the background MLP's persistent launches
    // (durf_mlp_fwd_enc_obj / durf_mlp_bwd_obj, round 6; those fall back to two launches each where the mix does not apply)
    const bool mix = Kb > 0 && !ov.sd && durf::obj_mix(rows) && N % 32 == 0;
    for (int lvl = 0; lvl < L; lvl++) {
        float* t_vals = f.t_vals[lvl];
        if (ov.sd) {              // bf16 objects of a large step: issued first, on the side stream (joined before the composite)
            STEP(ov.fork());
            STEP(durf_obj_fwd_batch(ov.obj(), K, B, N, w.idx_obj, w.count_obj, t_vals, w.o_s, w.d_s, f.radii, f.barf_w, obj_flags,
                                    w.view, w.wf_obj, w.obj_enc[lvl], w.obj_raw[lvl], w.obj_stash[lvl], w.obj_mask[lvl],
                                    lvl == 0 ? w.obj_view_tile : nullptr));
        }
        if (K > 0) {
            // (bf16 objects: the forward writes raw in the full layout itself, DURF_FWD_RAW_FULL; fp32 objects: the box-hit
            // rays' rows come from k_bkgd_hit_rays' fp32 evaluation instead, through durf_expand_raw)
            if (mix)          // small step, bf16 objects: the object MLPs' items ride in the background launch (k_mlp_fwd<.., MIX>)
                TIMED(DURF_TIMED_FWD + lvl,
                      durf_mlp_fwd_enc_obj(stream, rows, N, t_vals, w.o_s, w.d_s, f.radii, w.hit, K, f.enc_flags | DURF_FWD_RAW_FULL,
                                           w.enc[lvl], w.view, w.idx_cls, w.count_cls, w.wf_bkgd, w.raw_b[lvl], w.stash[lvl], w.mask[lvl],
                                           w.idx_cls + B, w.count_cls + 1, lvl == 0 ? w.view_tile : nullptr, B, w.idx_obj, w.count_obj,
                                           f.barf_w, obj_flags, w.wf_obj, w.obj_enc[lvl], w.obj_raw[lvl], w.obj_stash[lvl], w.obj_mask[lvl],
                                           lvl == 0 ? w.obj_view_tile : nullptr));
            else
            TIMED(DURF_TIMED_FWD + lvl,
                  durf_mlp_fwd_enc(stream, rows, N, t_vals, w.o_s, w.d_s, f.radii, w.hit, K, f.enc_flags | (f32o ? 0 : DURF_FWD_RAW_FULL),
                                   w.enc[lvl], w.view, w.idx_cls, w.count_cls, w.wf_bkgd, f32o ? w.raw_c[lvl] : w.raw_b[lvl], w.stash[lvl],
                                   w.mask[lvl], w.idx_cls + B, w.count_cls + 1, lvl == 0 ? w.view_tile : nullptr));
            if (f32o && lvl == 0) {
                // the background MLP's ONE fp32 evaluation of every box-hit ray (same input at both levels: once per step), HERE,
                // behind the level-0 forward: a trunk prefetched behind the previous step's update (durf_train_step below) has
                // had that launch's time to land, and beside the persistent launch these small ones would wait for a CU
                float* trunk = a->const_trunk ? a->const_trunk : w.trunk;
                // A prefetch the previous durf_train_step left running on the side stream (it reads the parameters and writes the
                // caller's const_trunk) is joined HERE whatever this call was told about the buffer: a recomputation into the same
                // buffer must not race with it, and this step's optimizer must not update parameters it is still reading.
                STEP(durf::join_prefetch(stream));
                if (!(a->const_trunk && a->const_trunk_valid))
                    STEP(durf_bkgd_const_trunk_f32(stream, f.bkgd_params, trunk));
                STEP(durf_bkgd_hit_rays_f32(stream, B, w.view27, f.bkgd_params, w.idx_cls + B, w.count_cls + 1, trunk, w.raw_tail));
            }
            if (f32o) STEP(durf_expand_raw(stream, B, N, w.raw_c[lvl], w.count_cls, w.slot_cls, w.raw_b[lvl], w.raw_tail));
            if (f32o)
                STEP(durf_objf32_fwd_batch(stream, K, B, N, w.idx_obj, w.count_obj, nullptr, w.view27, f.obj_params,
                                           f.obj_param_stride, w.obj_ws, w.obj_raw[lvl], w.act32[lvl], t_vals, w.o_s, w.d_s,
                                           f.radii, f.barf_w, obj_flags | (x3 ? DURF_F32_X3 : 0)));
            else if (!ov.sd && !mix)
                STEP(durf_obj_fwd_batch(stream, K, B, N, w.idx_obj, w.count_obj, t_vals, w.o_s, w.d_s, f.radii, f.barf_w, obj_flags,
                                        w.view, w.wf_obj, w.obj_enc[lvl], w.obj_raw[lvl], w.obj_stash[lvl], w.obj_mask[lvl],
                                        lvl == 0 ? w.obj_view_tile : nullptr));
            STEP(ov.join());
        } else {
            TIMED(DURF_TIMED_FWD + lvl,
                  durf_mlp_fwd_enc(stream, rows, N, t_vals, w.o_s, w.d_s, f.radii, nullptr, 0, f.enc_flags, w.enc[lvl], w.view, nullptr,
                                   nullptr, w.wf_bkgd, w.raw_b[lvl], w.stash[lvl], w.mask[lvl], nullptr, nullptr,
                                   lvl == 0 ? w.view_tile : nullptr));
        }
        if (f.density_noise != 0.0f)       // obbpose_model.py:236-240, on the background's raw density as the Python path adds it
            STEP(durf_density_noise(stream, rows, w.raw_b[lvl], f.density_noise, f.density_rand[lvl], f.seed_lo, f.seed_hi, lvl));
        if (lvl + 1 < L)        // composite + resample + the loss normalisers of this (level 0 only) and the next level: one launch
            TIMED(DURF_TIMED_COMPOSITE + lvl,
                 durf_composite_resample(stream, B, N, K, w.raw_b[lvl], raw_obj[lvl], w.slot_obj, t_vals, w.d_s, f.density_bias,
                                         f.bkgd_mode, f.rgb[lvl], f.depth[lvl], f.acc[lvl], f.weights[lvl], f.t_mids[lvl],
                                         f.t_dists[lvl], f.resample_padding, f.draw_noise ? w.u_rand + (size_t)lvl * B * (N + 1) : f.u_rand, f.t_vals[lvl + 1], a->lossmult, a->gt_depth,
                                         a->sky, f.dyn_mask, f.zo, a->eps, a->box_loss_mult, lvl, a->disable_multiscale,
                                         lvl == 0 ? w.prep : nullptr, lvl == 0 ? w.norms : nullptr, w.prep + (size_t)5 * B,
                                         w.norms + (size_t)(lvl + 1) * 5));
        // (the last level launches no composite: durf_loss_bwd recomputes it and fills its rendered outputs)
    }
    // ---- losses + backward (train_boxpose.py:67-252), last level first ----
    // (the view-direction tile of the weight-gradient launch was written by the level-0 forward)
    {   // every level's loss + composite backward: ONE launch (stop_level_grad: functions of the forward alone)
        durf_loss_level ll[ML] = {};
        for (int lvl = 0; lvl < L; lvl++) {
            const bool last = lvl == L - 1;
            durf_loss_level& h = ll[lvl];
            h.raw_bkgd = w.raw_b[lvl]; h.t_vals = f.t_vals[lvl]; h.norm = w.norms + (size_t)lvl * 5; h.level = lvl;
            for (int k = 0; k < K; k++) h.raw_obj[k] = raw_obj[lvl][k];
            for (int i = 0; i < 6; i++) h.mults[i] = a->level_mults[lvl][i];
            h.draw = w.draw[lvl]; h.terms = w.terms[lvl];
            if (last) { h.rgb_out = f.rgb[lvl]; h.depth_out = f.depth[lvl]; h.acc_out = f.acc[lvl]; h.weights_out = f.weights[lvl];
                        h.t_mids_out = f.t_mids[lvl]; h.t_dists_out = f.t_dists[lvl]; }
            h.draw_ray_sum = K > 0 ? w.ray_sums + (size_t)lvl * B * 4 : nullptr;
        }
        STEP(durf_loss_bwd_levels(stream, B, N, K, L, ll, w.slot_obj, w.d_s, a->pixels, a->lossmult, a->gt_depth, a->sky, f.dyn_mask,
                                  f.zo, a->eps, a->box_loss_mult, a->disable_multiscale, a->bg, f.density_bias));
    }
    STEP(ov.fork());          // (ONE fork for the whole backward: the object launches read d(raw), which the loss launch above wrote
                              // for every level, and what their own forward left on the side stream)
    if (Kb > 0 && !mix) {     // the object backward of EVERY level: level by level on the side stream, in the shadow of the background
                              // backward (large steps; a small step's rides in the background launches below)
        const float* dr[ML]; const void* mk[ML]; void* dzl[ML]; void* dzo[ML];
        for (int l = 0; l < L; l++) { dr[l] = w.draw[L - 1 - l]; mk[l] = w.obj_mask[L - 1 - l]; dzl[l] = w.obj_dz[L - 1 - l]; dzo[l] = w.obj_dz_out[L - 1 - l]; }
        STEP(durf_obj_bwd_batch_levels(ov.obj(), K, B, N, L, w.idx_obj, w.count_obj, dr, w.wb_obj, mk, dzl, dzo));
    }
    if (f32o) {           // the object branch in fp32: backward of every level, all K at once; then d(enc) -> the 21 pose sums per
                          // object for EVERY level as one launch pair (levels added last level first, as one pair per level would)
        const float *de[ML], *tv[ML];
        for (int lvl = L - 1; lvl >= 0; lvl--) {
            STEP((x3 ? durf_objf32_bwd_batch_x3 : durf_objf32_bwd_batch)(stream, K, B, N, w.idx_obj, w.count_obj, w.draw[lvl], f.obj_params,
                                                                         f.obj_param_stride, w.obj_ws, w.act32[lvl], w.dz32[lvl],
                                                                         pose_opt ? w.d_enc32[lvl] : nullptr));
            de[L - 1 - lvl] = w.d_enc32[lvl]; tv[L - 1 - lvl] = f.t_vals[lvl];
        }
        if (pose_opt)
            STEP(durf_encode_obj_bwd_levels(stream, K, B, N, L, w.idx_obj, w.count_obj, de, tv, w.o_s, w.d_s, f.radii, f.origins,
                                            f.directions, f.pose, f.barf_w, w.pose_scratch, w.pose_sums, 1, obj_flags));
    }
    for (int lvl = L - 1; lvl >= 0; lvl--) {
        float* rs = K > 0 ? w.ray_sums + (size_t)lvl * B * 4 : nullptr;
        if (mix) {            // this level's background backward + the object MLPs' backward of the same level: one launch
            const float* dr1[1] = {w.draw[lvl]}; const void* mk1[1] = {w.obj_mask[lvl]};
            void* dz1[1] = {w.obj_dz[lvl]}; void* dzo1[1] = {w.obj_dz_out[lvl]};
            TIMED(DURF_TIMED_BWD + lvl,
                  durf_mlp_bwd_obj(stream, rows, N, w.draw[lvl], w.idx_cls, w.count_cls, w.wb_bkgd, w.mask[lvl], w.dz[lvl], w.dz_out[lvl],
                                   w.idx_cls + B, w.count_cls + 1, rs, K, B, 1, w.idx_obj, w.count_obj, dr1, w.wb_obj, mk1, dz1, dzo1));
        } else if (K > 0) {
            TIMED(DURF_TIMED_BWD + lvl,
                  durf_mlp_bwd(stream, 256, rows, N, w.draw[lvl], w.idx_cls, w.count_cls, w.wb_bkgd, w.mask[lvl], w.dz[lvl], w.dz_out[lvl],
                               nullptr, w.idx_cls + B, w.count_cls + 1, rs));
        } else {
            TIMED(DURF_TIMED_BWD + lvl,
                  durf_mlp_bwd(stream, 256, rows, N, w.draw[lvl], nullptr, nullptr, w.wb_bkgd, w.mask[lvl], w.dz[lvl], w.dz_out[lvl], nullptr,
                               nullptr, nullptr, nullptr));
        }
    }
    // ---- weight gradients of every MLP over every level: the objects' split-K partials, the background's, one finalize ----
    // (the gradient buffer was zero filled by the prologue launch)
    const void *enc[ML], *vt[ML], *stash[ML], *dz[ML], *dzo[ML], *ovt[ML];
    size_t seg_rows[ML];
    int per_ray[ML];
    const int32_t* seg_count[ML];
    for (int l = 0; l < L; l++) {
        vt[l] = w.view_tile; ovt[l] = w.obj_view_tile;
        seg_rows[l] = rows;
        per_ray[l] = K > 0 ? 1 : N;                           // de-duplicated: one segment of count_cls[2] valid rows
        seg_count[l] = K > 0 ? w.count_cls + 2 : nullptr;
    }
    float* g_bkgd = a->grad + a->box_floats;
    float* g_obj = g_bkgd + a->mlp0_floats;
    if (f32o) {                 // fp32 object branch: its own weight-gradient launch pair over every level
        const float *act[ML], *dz32[ML];
        for (int l = 0; l < L; l++) { act[l] = w.act32[l]; dz32[l] = w.dz32[l]; }
        STEP(durf_objf32_dw_batch(stream, K, B, N, w.count_obj, L, act, dz32, OBJ32_NSPLIT, w.dw32_scratch, g_obj, a->obj_floats));
    } else if (K > 0) {
        for (int l = 0; l < L; l++) { enc[l] = w.obj_enc[l]; stash[l] = w.obj_stash[l]; dz[l] = w.obj_dz[l]; dzo[l] = w.obj_dz_out[l]; }
        if (ov.sd)            // large step: split-K launch + finalize of their own, beside the background's (same sums: the finalize
                              // launches add each MLP's partials in the same order whether they are merged or not)
            STEP(durf_obj_dw_batch(ov.obj(), K, B, N, w.count_obj, L, enc, ovt, stash, dz, dzo, 63, w.opart, w.obpart, g_obj,
                                   a->obj_floats, f.obj_params));
        else
            STEP(durf_obj_dw_partials(stream, K, B, N, w.count_obj, L, enc, ovt, stash, dz, dzo, w.opart, w.obpart));
    }
    const int Km = ov.sd ? 0 : Kb;          // object MLPs finalized together with the background MLP
    for (int l = 0; l < L; l++) { enc[l] = w.enc[l]; stash[l] = w.stash[l]; dz[l] = w.dz[l]; dzo[l] = w.dz_out[l]; }
    TIMED(DURF_TIMED_DW, durf_mlp_dw_levels(stream, 256, L, seg_rows, per_ray, seg_count, enc, vt, stash, dz, dzo, w.part, w.bpart));
    STEP(durf_dw_finalize_all(stream, 60, L, seg_rows, per_ray, seg_count, w.part, w.bpart, g_bkgd, f.bkgd_params, Km, Km ? B : 0,
                              Km ? N : 0, Km ? w.count_obj : nullptr, Km ? L : 1, 63, Km ? w.opart : nullptr, Km ? w.obpart : nullptr,
                              Km ? g_obj : nullptr, Km ? a->obj_floats : 0, Km ? f.obj_params : nullptr));
    STEP(ov.join());
    const float* wl2 = nullptr;
    if (a->weight_decay_mult != 0.0f) {       // train_boxpose.py:73-75 (in front of the pose rows' additions, as the Python path orders it)
        STEP(durf_weight_decay(stream, a->n_params, a->params, a->grad, 0, a->n_params, a->weight_decay_mult, w.scratch, w.weight_l2));
        wl2 = w.weight_l2;
    }
    if (pose_opt) {             // d(loss)/d(box_centers[ts]) (obbpose_model.py:99-131) + the TV prior, into this timestep's rows
        float* g_rows = a->grad + (f.pose - a->params);
        STEP(durf_pose_finish(stream, K, f.pose, w.pose_sums, a->want_pos, a->want_rot, g_rows));
        if (a->want_pos && a->tv_loss_mult != 0.0f) {
            const float c = (float)((double)a->tv_loss_mult * (1.0 + 0.1 * (double)(L - 1)) * 2.0);
            hipLaunchKernelGGL(k_tv_rows, dim3(1), dim3(64), 0, hs, K, f.pose, a->prev6, c, g_rows);
            DURF_CHECK_LAUNCH("durf_loss_backward (TV prior rows)");
        }
    }
    if (!tail) return 0;
    if (K > 1)                  // rays that hit two boxes: the reference's NaN -> zero update (durf_poison_multi_hit)
        STEP(durf_poison_multi_hit(stream, a->n_params, a->grad, w.count_cls, a->box_floats, K, a->mlp0_floats, a->obj_floats));
    // ---- the logged scalars (utils.Stats), one launch ----
    const float* tv[ML];
    const float* terms[ML];
    for (int l = 0; l < L; l++) { tv[l] = f.t_vals[l]; terms[l] = w.terms[l]; }
    STEP(durf_train_stats(stream, L, K, N, w.norms, w.sums, wl2, K ? f.pose : nullptr, K ? a->prev6 : nullptr,
                          K ? a->target6 : nullptr, tv, a->stat_mults, 3 /* assemble | psnr */, a->stats, terms, B));
    return 0;
}

}  // namespace

extern "C" {

size_t durf_train_workspace_bytes(int B, int N, int K, int num_levels, size_t n_params) {
    return carve(nullptr, B, N, K, num_levels, n_params).total;
}
size_t durf_train_workspace_bytes_flags(int B, int N, int K, int num_levels, size_t n_params, int flags) {
    return carve(nullptr, B, N, K, num_levels, n_params, flags).total;
}

int durf_loss_backward(void* stream, const durf_train_args* a, void* workspace, size_t workspace_bytes) {
    int rc = check_args(a, workspace);
    if (rc != 0) return rc;
    const TrainWs w = workspace_of(a, workspace);
    STEP(check_workspace("durf_loss_backward", a, w, workspace_bytes));
    return loss_backward(stream, a, w);
}

static int train_step_body(void* stream, const durf_train_args* a, void* workspace, size_t workspace_bytes);

int durf_train_step(void* stream, const durf_train_args* a, void* workspace, size_t workspace_bytes) {
    int rc = train_step_body(stream, a, workspace, workspace_bytes);
    if (rc != 0) return rc;
    if ((a->flags & DURF_TRAIN_OBJ_FP32) && a->f.K > 0 && a->prefetch_const_trunk && a->const_trunk) {
        // the NEXT step's constant trunk, from the parameters just updated, beside whatever the caller does between the steps
        durf::SideStream* sd = durf::side_stream_of_device();
        if (sd != nullptr) {
            const durf::Overlap ov{(hipStream_t)stream, sd};
            STEP(ov.fork());
            STEP(durf_bkgd_const_trunk_f32(ov.obj(), a->f.bkgd_params, a->const_trunk));
            durf::note_prefetch(sd, a->const_trunk);
        }
    }
    return 0;
}

static int train_step_body(void* stream, const durf_train_args* a, void* workspace, size_t workspace_bytes) {
    int rc = check_args(a, workspace);
    if (rc != 0) return rc;
    DURF_REQUIRE(a->adam_m && a->adam_v && a->grad_stats, "Adam moments and grad_stats");
    const TrainWs w = workspace_of(a, workspace);
    STEP(check_workspace("durf_train_step", a, w, workspace_bytes));
    STEP(loss_backward(stream, a, w, false));
    STEP(durf::join_prefetch(stream));       // (a step without the fp32 object branch behind one with it: nothing has joined yet)
    if (a->comm != nullptr) {
        // One rank's share of a data-parallel step (train_boxpose.py:253-255; durf_amd/train_boxpose.py train_step): the
        // multi-hit NaNs have to exist BEFORE the exchange (the reference's pmean sees them), then ONE all-reduce of the flat
        // gradient in this stream, and the optimizer on the mean.
        DURF_REQUIRE(a->world >= 1, "world: the ranks of comm");
        const durf_forward_args& f = a->f;
        const int K = f.K, L = f.num_levels;
        const float inv_world = 1.0f / (float)a->world;
        const float* tv[ML];
        const float* terms[ML];
        for (int l = 0; l < L; l++) { tv[l] = f.t_vals[l]; terms[l] = w.terms[l]; }
        const float* wl2 = a->weight_decay_mult != 0.0f ? w.weight_l2 : nullptr;
        if (K > 1)
            STEP(durf_poison_multi_hit(stream, a->n_params, a->grad, w.count_cls, a->box_floats, K, a->mlp0_floats, a->obj_floats));
        STEP(durf_allreduce_sum(stream, a->comm, a->grad, a->n_params));
        if (!a->reduce_stats) {          // shard-local scalars (the reference reads them every print_every steps only)
            STEP(durf_stats_scrub(stream, L, K, f.N, w.norms, w.sums, wl2, K ? f.pose : nullptr, K ? a->prev6 : nullptr,
                                  K ? a->target6 : nullptr, tv, a->stat_mults, 3, a->stats, terms, f.B, a->n_params, a->grad,
                                  inv_world, a->max_val, w.scratch, nullptr, 0, 0, 0, 0));
            return durf_adam_apply(stream, a->n_params, a->params, a->adam_m, a->adam_v, a->grad, a->max_norm, a->lr, a->step,
                                   w.scratch, a->grad_stats);
        }
        // lax.pmean(stats) (:255), then the PSNRs from the averaged losses (:291-292)
        const int ns = 2 + 17 * L;
        STEP(durf_train_stats(stream, L, K, f.N, w.norms, w.sums, wl2, K ? f.pose : nullptr, K ? a->prev6 : nullptr,
                              K ? a->target6 : nullptr, tv, a->stat_mults, 1, a->stats, terms, f.B));
        STEP(durf_allreduce_sum(stream, a->comm, a->stats, (size_t)ns));
        hipLaunchKernelGGL(k_scale, dim3(1), dim3(256), 0, (hipStream_t)stream, ns, a->stats, inv_world);
        DURF_CHECK_LAUNCH("durf_train_step (pmean of the logged scalars)");
        STEP(durf_train_stats(stream, L, K, f.N, w.norms, w.sums, nullptr, nullptr, nullptr, nullptr, tv, a->stat_mults, 2, a->stats,
                              nullptr, 0));
        return durf_clip_adam(stream, a->n_params, a->params, a->adam_m, a->adam_v, a->grad, inv_world, a->max_val, a->max_norm,
                              a->lr, a->step, w.scratch, a->grad_stats);
    }
    // the step's tail in two launches (as durf_amd/train_boxpose.py issues it on one device): the logged scalars + the
    // multi-hit outcome + the optimizer's scrub pass, then Adam
    const durf_forward_args& f = a->f;
    const int K = f.K, L = f.num_levels;
    const float* tv[ML];
    const float* terms[ML];
    for (int l = 0; l < L; l++) { tv[l] = f.t_vals[l]; terms[l] = w.terms[l]; }
    STEP(durf_stats_scrub(stream, L, K, f.N, w.norms, w.sums, a->weight_decay_mult != 0.0f ? w.weight_l2 : nullptr, K ? f.pose : nullptr, K ? a->prev6 : nullptr,
                          K ? a->target6 : nullptr, tv, a->stat_mults, 3 /* assemble | psnr */, a->stats, terms, f.B, a->n_params,
                          a->grad, 1.0f, a->max_val, w.scratch, K > 1 ? w.count_cls : nullptr, a->box_floats, K > 1 ? K : 0,
                          a->mlp0_floats, a->obj_floats));
    return durf_adam_apply(stream, a->n_params, a->params, a->adam_m, a->adam_v, a->grad, a->max_norm, a->lr, a->step, w.scratch,
                           a->grad_stats);
}

int durf_prefetch_join(void* stream) { return durf::join_prefetch(stream); }

}  // extern "C"
