/* durf_hip.h -- C ABI of libdurf_hip.so, the MI355X (gfx950) ray-pipeline library.
 *
 * The reference (FelTris/durf) has NO FFI/plugin boundary: its hot path is plain
 * Python/JAX.  This ABI is therefore what a reference-side
 * binding (ctypes stub) would call to replace the body of
 *   MipNerfModel.__call__     internal/obbpose_model.py:68-261
 *   train_step / loss_fn      train_boxpose.py:49-321
 * Each entry point cites the reference lines it replaces.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer owned by the caller (a torch allocation);
 *    the library never frees or retains it past the call;
 *  - every call is asynchronous on `stream` (a hipStream_t passed as void*);
 *  - return value: 0 = ok, otherwise a hipError_t / negative durf error;
 *    durf_last_error() returns a thread-local message;
 *  - tensors are row-major fp32 with the reference's shapes unless stated;
 *  - "tile layout" = bf16 MFMA B-fragment-major layout of a [rows, 16*nks] matrix:
 *        elem(row, f) at  (((row/32)*nks + f/16) * 64 + ((f/8)&1)*32 + row%32) * 8 + f%8
 *    i.e. per 32-row tile, per 16-feature k-step, lanes (hi=(f/8)&1, n=row%32) hold 8
 *    consecutive features (one 16-byte vector per lane).
 */
#ifndef DURF_HIP_H
#define DURF_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DURF_MAX_OBJ 16
#define DURF_MAX_LEVELS 8
#define DURF_MAX_CAMS 8        /* cameras of one timestep (Waymo rig: 5) */
#define DURF_W_BKGD 256      /* MLP.net_width (obbpose_model.py:296) */
#define DURF_W_OBJ 128       /* BoxMLP.net_width (:360) */
/* flags of durf_encode_bkgd (`contraction` argument) / durf_encode_obj */
#define DURF_ENC_CONTRACT 1        /* MipNerfModel.contraction (mip360.new_space) */
#define DURF_ENC_NO_INTEGRATION 2  /* MipNerfModel.disable_integration: PE instead of IPE (covariances zeroed) */
#define DURF_ENC_CYLINDER 4        /* MipNerfModel.ray_shape == 'cylinder' (mip.cylinder_to_gaussian) */
#define DURF_FWD_RAW_FULL 8        /* durf_mlp_fwd_enc only (not an encoder flag): see there */
#define DURF_ENC_DIM 64      /* 60 (bkgd IPE) / 63 (object IPE) features padded to 64 */
#define DURF_VIEW_DIM 32     /* 27 view-direction features padded to 32 */

const char* durf_last_error(void);
int durf_version(void);
/* Which of the size- or switch-selected kernel variants have been launched by this process since durf_dispatch_reset():
 * a bit mask of DURF_DISPATCH_*.  Test instrumentation of the launchers (one relaxed atomic OR per launch): the parity
 * suite asserts that every variant below ran inside an oracle-compared test (tests/test_gpu_dispatch_matrix.py). */
#define DURF_DISPATCH_FWD256_8W 0x1        /* k_mlp_fwd<256>: 256-sample blocks of 8 waves */
#define DURF_DISPATCH_FWD256_4W 0x2        /* k_mlp_fwd<256>: 128-sample blocks of 4 waves (launches of <= 128 blocks) */
#define DURF_DISPATCH_FWD128_SAMPLE 0x4    /* k_mlp_fwd<128>: the object MLPs, one wave per 32 samples */
#define DURF_DISPATCH_FWD128_MSPLIT 0x8    /* k_mlp_fwd_ms: the object MLPs, one wave per output tile (< 2048 x 128 rows) */
#define DURF_DISPATCH_BWD256_8W 0x10
#define DURF_DISPATCH_BWD256_4W 0x20
#define DURF_DISPATCH_BWD128_SAMPLE 0x40
#define DURF_DISPATCH_BWD128_MSPLIT 0x80
#define DURF_DISPATCH_DW256_256WG 0x100    /* k_dw_all<256>: one round of 256 workgroups (< 3072 x 256 rows over all levels) */
#define DURF_DISPATCH_DW256_512WG 0x200    /* k_dw_all<256>: two rounds */
#define DURF_DISPATCH_DW128_128WG 0x400    /* k_dw_all<128> per object */
#define DURF_DISPATCH_DW128_256WG 0x800
#define DURF_DISPATCH_FWD_ENC 0x1000       /* the forward encodes its own tiles (durf_mlp_fwd_enc / durf_obj_fwd_batch) */
#define DURF_DISPATCH_FWD_RAW_FULL 0x2000  /* ... and writes raw in the full [B*N,4] layout (DURF_FWD_RAW_FULL) */
#define DURF_DISPATCH_FWD_TAIL 0x4000      /* tail rows: the de-duplicated background evaluation */
#define DURF_DISPATCH_F32_DW_TILE 0x8000   /* k_mlp_dw_f32: one output tile per workgroup (the fp32 object branch) */
#define DURF_DISPATCH_F32_DW_B2 0x10000    /* k_mlp_dw_f32_b2: 2 x 2 blocks (W = 256) */
#define DURF_DISPATCH_BWD_POSE 0x20000     /* k_mlp_bwd<.., POSE>: d(enc) for the box-pose gradient */
#define DURF_DISPATCH_FWD_MIX 0x40000      /* k_mlp_fwd<256, .., MIX>: background blocks + the object MLPs' items in ONE launch */
#define DURF_DISPATCH_BWD_MIX 0x80000      /* k_mlp_bwd<256, .., MIX>: the same for the backward */
int durf_dispatch_seen(void);
int durf_dispatch_reset(void);

/* ---- parameter layout -------------------------------------------------------
 * One contiguous fp32 buffer (so the data-parallel gradient exchange is a single
 * all-reduce, train_boxpose.py:253):
 *   [ box_centers T*K*6 | MLP_0: Dense_0.kernel[in,out], Dense_0.bias, ... Dense_11 |
 *     BoxMLP_0 ... | BoxMLP_{K-1} ... ]
 * Dense order/shape as flax creates them (obbpose_model.py:329-353): 8 trunk layers,
 * density head, bottleneck, view layer, rgb head.  kernels are [in,out] row-major. */
size_t durf_mlp_param_count(int width, int in_dim);          /* floats in one MLP */
size_t durf_mlp_layer_offset(int width, int in_dim, int layer, int want_bias);
/* bf16 fragment-ordered weight streams consumed by the fused MLP kernels */
size_t durf_wpack_fwd_bytes(int width);
size_t durf_wpack_bwd_bytes(int width);
int durf_pack_weights(void* stream, int width, int in_dim, const float* mlp_params,
                      void* wpack_fwd, void* wpack_bwd /* nullable */);
/* Every weight stream of the model in ONE launch (a training step re-packs after each optimizer update): the
 * background MLP (width 256; bkgd_params nullable = skip) and K object MLPs (width 128, parameters obj_param_stride
 * floats apart, streams durf_wpack_{fwd,bwd}_bytes(128) apart).  The backward streams are nullable (inference). */
int durf_pack_weights_all(void* stream, const float* bkgd_params, int in_bkgd, void* bkgd_fwd, void* bkgd_bwd,
                          int K, const float* obj_params, size_t obj_param_stride, int in_obj, void* obj_fwd,
                          void* obj_bwd);

/* ---- stage-level entry points (each one kernel; used by parity + roofline tests) */

/* K1 box setup: aa2matrix, world2object_rpy, ray_box_intersection, ray select.
 * box_helpers.py:148-167,170-181,286-341,59-106; obbpose_model.py:99-131.
 * pose = box_centers[ts] [K,6]; ext [K,3]; hit [B,K] int32; zo[B]; */
int durf_ray_setup(void* stream, int B, int K, const float* origins, const float* dirs,
                   const float* pose, const float* ext, float* origins_s, float* dirs_s,
                   int32_t* hit, float* zo);

/* deterministic stream compaction of hit[:,k]: idx[k*B + j] = j-th ray hitting k,
 * count[k] = number of such rays, slot[b*K+k] = position of ray b in list k or -1. */
int durf_compact_hits(void* stream, int B, int K, const int32_t* hit, int32_t* idx,
                      int32_t* count, int32_t* slot);
/* The two ray classes of the de-duplicated background evaluation (durf_expand_raw) from hit [B,K] in one launch:
 * class 0 = rays that hit no box or several, class 1 = rays that hit exactly one.  idx [2,B], slot [B,2] as above;
 * count [5] = {class-0 rays, class-1 rays, class0 * N + class1 (valid rows of the compacted buffers), rays that hit
 * several boxes, bit k set: box k is hit by such a ray}; dyn [B] = boxes hit per ray (obbpose_model.py:257 `jnp.array(ret_masks).sum(axis=0)`). */
int durf_compact_classes(void* stream, int B, int K, int N, const int32_t* hit, int32_t* idx, int32_t* count,
                         int32_t* slot, int32_t* dyn);
/* Both compactions in one launch (a training step needs both): the per-object lists of durf_compact_hits and the two
 * ray classes of durf_compact_classes. */
int durf_compact_all(void* stream, int B, int K, int N, const int32_t* hit, int32_t* idx_obj, int32_t* count_obj,
                     int32_t* slot_obj, int32_t* idx_cls, int32_t* count_cls, int32_t* slot_cls, int32_t* dyn);
/* The three preparations of a step that depend on the batch only, as ONE launch: durf_ray_setup, durf_view_enc (bf16
 * output) and durf_sample_t -- same kernels' code, same results.  Two chores of a training step ride along (each a launch of
 * its own otherwise): pose_copy [K,6] receives a snapshot of `pose` (train_step returns the poses it rendered with,
 * train_boxpose.py:315, while the optimizer updates the parameters in place), and zero_buf[0..zero_count) -- the flat
 * gradient buffer, 16-byte aligned -- is zero filled.
 * u_rand_out (nullable [3, B, N+1]; t_rand must then be NULL): the launch DRAWS the step's stratified-sampling noise itself, as
 * the reference draws inside its program (mip.py:364, math.py:257-260): Philox4x32-10 with counter (i, 0, 0, 0) and key
 * (seed_lo, seed_hi) for i in [0, B (N+1)); its output word 0, as (x >> 8) 2^-24 in [0, 1), jitters level-0 sample
 * position i exactly as t_rand[i] would; words 1, 2, 3 are written to planes 0, 1, 2 of u_rand_out -- plane l is the u_rand of
 * the durf_composite_resample / durf_resample that turns level l into level l + 1 (round 6: until then ONE plane served every
 * level, which correlated the draws of num_levels > 2).  No generator launch in front of the step; oracle/philox_ref.py
 * restates it. */
int durf_ray_prologue(void* stream, int B, int K, int N, const float* origins, const float* dirs, const float* pose,
                      const float* ext, float* origins_s, float* dirs_s, int32_t* hit, float* zo,
                      const float* viewdirs, void* view_bf16, const float* near, const float* far,
                      const float* t_rand /* nullable */, int lindisp, float* t_vals, float* pose_copy /* nullable */,
                      float* zero_buf /* nullable */, size_t zero_count, uint32_t seed_lo, uint32_t seed_hi,
                      float* u_rand_out /* nullable */);
/* durf_ray_prologue + durf_pack_weights_all as ONE launch (same results of both): the step's bf16 weight streams depend on
 * the parameters only, so their packing rides in workgroups behind the prologue's own.  Arguments: durf_ray_prologue's, then
 * durf_pack_weights_all's (K_pack = object MLPs to pack: 0 when the object branch runs on the fp32 kernels). */
int durf_ray_prologue_pack(void* stream, int B, int K, int N, const float* origins, const float* dirs, const float* pose,
                           const float* ext, float* origins_s, float* dirs_s, int32_t* hit, float* zo,
                           const float* viewdirs, void* view_bf16, const float* near, const float* far,
                           const float* t_rand /* nullable */, int lindisp, float* t_vals, float* pose_copy /* nullable */,
                           float* zero_buf /* nullable */, size_t zero_count, uint32_t seed_lo, uint32_t seed_hi,
                           float* u_rand_out /* nullable */, const float* bkgd_params, int in_bkgd, void* bkgd_fwd,
                           void* bkgd_bwd /* nullable */, int K_pack, const float* obj_params, size_t obj_param_stride,
                           int in_obj, void* obj_fwd, void* obj_bwd /* nullable */,
                           float* zero_buf2 /* nullable: a second, small region to zero fill (4-byte words) */, size_t zero_count2);

/* mip.sample_along_rays t_vals (mip.py:353-368). t_rand nullable (randomized=False). */
int durf_sample_t(void* stream, int B, int N, const float* near, const float* far,
                  const float* t_rand, int lindisp /* MipNerfModel.lindisp, mip.py:354-356 */, float* t_vals);

/* MipNerfModel.density_noise on the randomized path (obbpose_model.py:236-240): raw[:, 3] += scale * z over the [rows, 4] raw
 * outputs of one level, z ~ N(0, 1) the caller's (`normal` [rows]) or, when NULL, drawn here: Philox4x32-10 block
 * (row, 1 + level, 0, 0) under (seed_lo, seed_hi) through Box-Muller (restated in oracle/philox_ref.py). */
int durf_density_noise(void* stream, size_t rows, float* raw, float scale, const float* normal /* nullable */,
                       uint32_t seed_lo, uint32_t seed_hi, int level);

/* view-direction encoding mip.pos_enc(viewdirs,0,4,True) (mip.py:36-45) -> [B,32]
 * bf16 (27 features, zero padded) and/or fp32 [B,27]. */
int durf_view_enc(void* stream, int B, const float* viewdirs, void* out_bf16, float* out_f32);

/* K2+K3 background encoding: cast_rays (mip.py:155-179,99-130,76-96), bkgd masking
 * (obbpose_model.py:205-210), mip360.new_space (mip360.py:47-79), integrated_pos_enc
 * (mip.py:226-282) -> 60 features/sample.  out_tile: bf16 tile layout [B*N, 64];
 * out_f32: [B*N, 60] row-major (either may be null). */
int durf_encode_bkgd(void* stream, int B, int N, const float* t_vals, const float* origins_s,
                     const float* dirs_s, const float* radii, const int32_t* hit, int K,
                     int contraction /* DURF_ENC_* flags */, void* out_tile, float* out_f32,
                     const int32_t* idx /* nullable */, const int32_t* count /* nullable: with idx, encode only the
                     rays idx[0..*count) into compacted rows (row r <-> ray idx[r/N]); bf16 tile output only */);

/* K4 object encoding for compacted hit rays of object k: weighted_ipe (mip.py:182-223),
 * no contraction, xyz prepended -> 63 features.  idx/count from durf_compact_hits
 * (device-side count; launch covers max_rays).  barf_w: 10 host floats (mip.py:217-218). */
int durf_encode_obj(void* stream, int max_rays, int N, const int32_t* idx, const int32_t* count,
                    const float* t_vals, const float* origins_s, const float* dirs_s,
                    const float* radii, const float* barf_w, int flags /* DURF_ENC_NO_INTEGRATION | DURF_ENC_CYLINDER */,
                    void* out_tile, float* out_f32);

/* K6/K7 fused MLP forward (obbpose_model.py:305-354 / :369-418), bf16 MFMA, fp32
 * accumulate.  rows = number of samples (multiple of N); enc_tile: [rows,64] tile layout;
 * view_bf16 [B,32]; ray_idx nullable (object MLPs: row r belongs to ray ray_idx[r/N]);
 * count nullable device int (valid rays; rows beyond count*N are skipped);
 * raw out [rows,4] = (raw_rgb[3], raw_density).  Training (both nullable): stash = bf16
 * activations (operands of the weight-gradient GEMMs), durf_mlp_stash_bytes(width, rows);
 * relu_mask = one bit per activation (all the backward data path needs),
 * durf_mlp_mask_bytes(rows). */
size_t durf_mlp_stash_bytes(int width, size_t rows);
size_t durf_mlp_mask_bytes(size_t rows);
int durf_mlp_fwd(void* stream, int width, size_t rows, int N, const void* enc_tile,
                 const void* view_bf16, const int32_t* ray_idx, const int32_t* count,
                 const void* wpack_fwd, float* raw, void* stash, void* relu_mask,
                 const int32_t* tail_idx /* nullable */, const int32_t* tail_count /* nullable: "tail rows", the
                 once-per-ray evaluations of a de-duplicated batch (durf_expand_raw): rows [count*N, count*N +
                 *tail_count) hold ONE sample of ray tail_idx[i] each, evaluated on the constant encoding of a
                 zero-masked Gaussian ([0 x 30, 1 x 30]; enc_tile is not read for them) with that ray's view
                 direction; raw / stash / relu_mask rows follow the same numbering */);

/* The background forward that ENCODES ITS OWN TILES: durf_encode_bkgd (bf16 tile output) + durf_mlp_fwd(width = 256) as
 * ONE launch.  Every workgroup computes the 60 features of its 256 samples from the ray data at the head of the block
 * (the body of the stand-alone encoder: bit-identical features, obbpose_model.py:205-210, mip.py:155-179,226-282,
 * mip360.py:47-79), keeps them as its first MFMA operand and writes the tile to enc_tile -- an OUTPUT here, [rows,64]
 * tile layout as durf_encode_bkgd writes it: the skip connection re-reads it, and in training the weight-gradient GEMMs
 * of Dense_0 / Dense_5 do.  t_vals [B,N+1], origins_s / dirs_s [B,3], radii [B], hit [B,K] (nullable with K = 0: no
 * object masking), enc_flags as durf_encode_bkgd's `contraction`; row r is sample r % N of ray r / N, or of ray
 * ray_idx[r / N] when a compacted list is given (with count); everything else as durf_mlp_fwd.  raw / stash / relu_mask
 * are bit-identical to the two separate calls'.  view_tile (nullable, training): the launch also writes the per-sample
 * view-direction tile [rows,32] that durf_expand_view would (the fragment each lane holds for the view layer IS its tile
 * layout) -- one launch less per step.  enc_flags | DURF_FWD_RAW_FULL with a compacted list + tail (a de-duplicated batch,
 * N % 32 == 0): raw is written in the FULL [B*N,4] layout -- row ray_idx[j]*N + n for the compacted rows, and the one
 * evaluation of tail ray tail_idx[i] at all N samples of that ray -- i.e. exactly what durf_expand_raw makes of the compacted
 * raw (bit-identical), which then is not called: one launch less per level. */
int durf_mlp_fwd_enc(void* stream, size_t rows, int N, const float* t_vals, const float* origins_s, const float* dirs_s,
                     const float* radii, const int32_t* hit /* nullable */, int K, int enc_flags, void* enc_tile,
                     const void* view_bf16, const int32_t* ray_idx /* nullable */, const int32_t* count /* nullable */,
                     const void* wpack_fwd, float* raw, void* stash /* nullable */, void* relu_mask /* nullable */,
                     const int32_t* tail_idx /* nullable */, const int32_t* tail_count /* nullable */,
                     void* view_tile /* nullable */);
/* One level's forward of BOTH MLP classes of a small training step as ONE launch (round 6): durf_mlp_fwd_enc (the background
 * MLP on the de-duplicated ray classes: arguments as there, ray_idx / count / tail_* required) + durf_obj_fwd_batch (the K
 * BoxMLPs on their compacted hit lists, obbpose_model.py:174-201: B rays, obj_idx [K,B] / obj_count [K] from
 * durf_compact_hits, slabs as there).  A heterogeneous persistent grid: every workgroup walks its background blocks, then
 * becomes two 4-wave groups that take (object, tile pair) items off an atomic ticket counter -- the ~10 % of the workgroups
 * the de-duplication leaves without a block (512 rays) or with one block fewer (1024 rays, K = 8) absorb the object MLPs
 * inside the background launch's own duration, where a second launch cost 22 us per level and a second stream delayed the
 * persistent workgroups.  Every output (raw, encoding tiles, stashes, masks, view tiles of both classes) is bit-identical
 * to the two separate calls'.  Applies in training below 2048 x 128 sample rows (the M-split regime of the object
 * kernels); otherwise, with inference buffers (stash == NULL) or under DURF_OBJ_MIX=0, the call issues the two launches. */
int durf_mlp_fwd_enc_obj(void* stream, size_t rows, int N, const float* t_vals, const float* origins_s, const float* dirs_s,
                         const float* radii, const int32_t* hit, int K, int enc_flags, void* enc_tile, const void* view_bf16,
                         const int32_t* ray_idx, const int32_t* count, const void* wpack_fwd, float* raw,
                         void* stash /* nullable */, void* relu_mask /* nullable */, const int32_t* tail_idx /* nullable */,
                         const int32_t* tail_count /* nullable */, void* view_tile /* nullable */, int B,
                         const int32_t* obj_idx, const int32_t* obj_count, const float* barf_w /* host float[10] */,
                         int obj_flags, const void* obj_wpack_fwd, void* obj_enc, float* obj_raw,
                         void* obj_stash /* nullable */, void* obj_relu_mask /* nullable */, void* obj_view_tile /* nullable */);

/* K8 merge + activations + volumetric_rendering (obbpose_model.py:232-254, mip.py:285-327).
 * raw_bkgd [B*N,4]; raw_obj[k] [count_k*N,4] compacted, slot from durf_compact_hits.
 * bkgd_mode: 0 = grey 0.5 (rand_bkgd=False, white_bkgd=False), 1 = white, 2 = rand_bkgd
 * (adds randint(0,1)==0, mip.py:324).  Outputs nullable except weights. */
int durf_composite_fwd(void* stream, int B, int N, int K, const float* raw_bkgd,
                       const float* const* raw_obj /* host array of K device ptrs */,
                       const int32_t* slot, const float* t_vals, const float* dirs_s,
                       float density_bias, int bkgd_mode, float* rgb, float* depth, float* acc,
                       float* weights, float* t_mids, float* t_dists);

/* K8 + K9 in one launch for a level that is followed by another (obbpose_model.py:143-151,232-254):
 * durf_composite_fwd, then durf_resample of its weights (handed over through LDS) -> t_vals_out [B,N+1];
 * outputs bit-identical to the two separate calls.  With lossmult != NULL it also does durf_loss_prep's job
 * for level+1 (from t_vals_out) into prep_next/norm_next and, if prep_this != NULL, for `level` (from t_vals)
 * into prep_this/norm_this, so that a training step launches nothing else for the normalisers. */
int durf_composite_resample(void* stream, int B, int N, int K, const float* raw_bkgd, const float* const* raw_obj,
                            const int32_t* slot, const float* t_vals, const float* dirs_s, float density_bias,
                            int bkgd_mode, float* rgb, float* depth, float* acc, float* weights, float* t_mids,
                            float* t_dists, float resample_padding, const float* u_rand, float* t_vals_out,
                            const float* lossmult /* nullable: no loss prep */, const float* gt_depth, const float* sky,
                            const int32_t* dyn, const float* zo, float eps, float box_loss_mult, int level,
                            int disable_multiscale, float* prep_this /* nullable */, float* norm_this,
                            float* prep_next, float* norm_next);

/* K9 resample: mip.resample_along_rays + math.sorted_piecewise_constant_pdf
 * (mip.py:373-416, math.py:222-284). u_rand nullable. */
int durf_resample(void* stream, int B, int N, const float* t_vals, const float* weights,
                  float resample_padding, const float* u_rand, float* t_vals_out);

/* math.sorted_piecewise_constant_pdf(key, bins, weights, num_samples = N + 1, randomized) (math.py:222-284) on its
 * own -- the routine inside durf_resample without the blur-pool / padding that mip.resample_along_rays applies first
 * (mip.py:393-404).  bins [B,N+1], weights [B,N], u_rand [B,N+1] nullable (randomized=False), samples [B,N+1].
 * The reference's own properties for it (internal/math_test.py:183-346) run against this entry point. */
int durf_sorted_piecewise_constant_pdf(void* stream, int B, int N, const float* bins, const float* weights,
                                       const float* u_rand, float* samples);

/* ---- training: losses, backward, optimizer ----------------------------------- */

/* per-level normalisers of loss_fn (train_boxpose.py:94-102,138-140,164): writes
 * prep[5,B] = per-ray {lossmult mask, depth_mask, sky_mask, min near-dist^2, dyn_mask} and
 * norm[5] = their sums (row 3: min).  dyn[B] = sum_k hit; zo from durf_ray_setup. */
int durf_loss_prep(void* stream, int B, int N, const float* t_vals, const float* lossmult,
                   const float* gt_depth, const float* sky, const int32_t* dyn, const float* zo,
                   float eps, float box_loss_mult, int level, int disable_multiscale, float* prep,
                   float* norm);

/* K10 + composite backward: per-level loss terms of train_boxpose.py:123-192 and
 * d(loss)/d(raw) [B*N,4] through mip.volumetric_rendering / sigmoid / softplus.
 * mults[6] = this level's multipliers of (rgb, sky, depth, near, empty, distortion) in the
 * total loss (:211-220).  terms[7,B] per-ray numerators, term_sums[7] their sums:
 * {rgb, obj_rgb, depth, near, empty, sky, distortion}. */
int durf_loss_bwd(void* stream, int B, int N, int K, const float* raw_bkgd, const float* const* raw_obj,
                  const int32_t* slot, const float* t_vals, const float* dirs_s, const float* pixels,
                  const float* lossmult, const float* gt_depth, const float* sky, const int32_t* dyn,
                  const float* zo, const float* norm, float eps, const float* mults,
                  float box_loss_mult, int level, int disable_multiscale, float bg, float density_bias,
                  float* draw, float* terms, float* term_sums /* nullable: see durf_train_stats */,
                  float* rgb_out /* nullable [B,3] */, float* depth_out /* nullable [B] */,
                  float* acc_out /* nullable [B] */, float* weights_out /* nullable [B,N] */,
                  float* t_mids_out, float* t_dists_out /* nullable [B,N], written with weights_out: the level's
                  rendered outputs, bit-identical to durf_composite_fwd's, for a level nothing is resampled from */,
                  float* draw_ray_sum /* nullable [B,4]: sum over each ray's samples of draw (the head gradient of the
                  ONE background sample a box-hit ray is evaluated with, see durf_expand_raw) */);

/* durf_loss_bwd for EVERY level in one launch (stop_level_grad: each level's loss gradient depends on the forward only;
 * on one stream the levels below the last otherwise sit between two backward kernels as launches of their own).
 * levels: HOST array of L descriptions -- the per-level arguments of durf_loss_bwd (no term_sums: durf_train_stats reduces
 * the terms); everything else is shared.  Outputs bit-identical to L calls of durf_loss_bwd. */
typedef struct durf_loss_level {
    const float* raw_bkgd;                 /* [B*N,4] */
    const float* raw_obj[DURF_MAX_OBJ];    /* device pointers, the first K used */
    const float* t_vals;                   /* [B,N+1] */
    const float* norm;                     /* [5] this level's normalisers (durf_loss_prep / durf_composite_resample) */
    float mults[6];                        /* rgb, sky, depth, near, empty, distortion multipliers of this level */
    int level;
    float *draw, *terms;                   /* out: [B*N,4], [7,B] */
    float *rgb_out, *depth_out, *acc_out, *weights_out, *t_mids_out, *t_dists_out;   /* nullable: the level's rendered outputs */
    float* draw_ray_sum;                   /* nullable [B,4] */
} durf_loss_level;
int durf_loss_bwd_levels(void* stream, int B, int N, int K, int L, const durf_loss_level* levels /* HOST */,
                         const int32_t* slot, const float* dirs_s, const float* pixels, const float* lossmult,
                         const float* gt_depth, const float* sky, const int32_t* dyn, const float* zo, float eps,
                         float box_loss_mult, int disable_multiscale, float bg, float density_bias);

/* Scalars of utils.Stats from the per-level sums (train_boxpose.py:123-249,291-292) in one launch.
 * norms [L,5] (durf_loss_prep), sums [L,7] (durf_loss_bwd), weight_l2 nullable device scalar,
 * pose6/prev6/target6 [K,6] (box_centers[ts], prev[0], batch target), t_vals: L host-side device
 * pointers, mults (host) {coarse, sky, depth, near, empty, tv}_loss_mult.
 * out [2 + 17 L]: loss | 15 rows of L (losses, obj_losses, d, n, e, s, distr, tv, offsets,
 * offset_x, offset_y, offset_z, offset_yaw, psnrs, obj_psnrs) | 2L sampling stats | weight_l2.
 * mode bit 0: everything but the PSNRs; bit 1: PSNRs from out's (possibly all-reduced) losses. */
int durf_train_stats(void* stream, int L, int K, int N, const float* norms, float* sums,
                     const float* weight_l2, const float* pose6, const float* prev6, const float* target6,
                     const float* const* t_vals, const float* mults, int mode, float* out,
                     const float* const* terms /* nullable: L device pointers to durf_loss_bwd's per-ray terms [7,B];
                     given (with mode bit 0), the launch reduces them itself into sums [L,7] -- in durf_loss_bwd's own
                     order -- and durf_loss_bwd may be called with term_sums = NULL (two launches fewer per step) */,
                     int B);

/* K11 fused MLP backward (data path).  draw [*,4] fp32 head gradients (object MLPs gather
 * rows through ray_idx); relu_mask from durf_mlp_fwd; dz: same size/layout as the stash,
 * receives every pre-activation gradient; dz_out: tile layout [rows,16] (slots 0-2 d rgb,
 * 3 d density). */
int durf_mlp_bwd(void* stream, int width, size_t rows, int N, const float* draw, const int32_t* ray_idx,
                 const int32_t* count, const void* wpack_bwd, const void* relu_mask, void* dz, void* dz_out,
                 float* d_enc /* nullable: [rows,64] fp32 d(loss)/d(encoding), for box-pose gradients */,
                 const int32_t* tail_idx, const int32_t* tail_count, const float* draw_ray_sum /* all nullable, together:
                 the tail rows of durf_mlp_fwd take their head gradient from draw_ray_sum[tail_idx[i]] (durf_loss_bwd) */);
int durf_expand_view(void* stream, size_t rows, int N, const void* view_bf16, const int32_t* ray_idx,
                     const int32_t* count, void* out_tile /* tile layout [rows,32] */,
                     const int32_t* tail_idx, const int32_t* tail_count /* nullable: tail rows as in durf_mlp_fwd */);

/* K11 weight gradients of one MLP: ONE grouped launch of the 12 split-K GEMMs (one per Dense)
 * whose K axis runs over the `rows` samples of EVERY level; enc_tile/view_tile/stash/dz/dz_out
 * are host arrays of nlevels device pointers (the per-level buffers of durf_mlp_fwd /
 * durf_mlp_bwd / durf_expand_view).  fp32 partials per (job, split); the finalize call sums
 * the splits in a fixed order into grad_mlp (flax layout of one MLP, overwritten).
 * part / bpart: durf_dw_part_floats / durf_dw_bpart_floats (width) floats. */
size_t durf_dw_part_floats(int width);
size_t durf_dw_bpart_floats(int width);
int durf_mlp_dw(void* stream, int width, size_t rows, int N, const int32_t* count, int nlevels,
                const void* const* enc_tile, const void* const* view_tile, const void* const* stash,
                const void* const* dz, const void* const* dz_out, float* part, float* bpart);
int durf_mlp_dw_finalize(void* stream, int width, int in_dim, size_t rows, int N, const int32_t* count,
                         int nlevels /* the same rows, N, count, nlevels as the durf_mlp_dw call */,
                         const float* part, const float* bpart, float* grad_mlp,
                         const float* mlp_params /* fp32 flax-layout parameters of this MLP: the bottleneck Dense_9 is
                         linear, so its gradients (and the bottleneck-fed rows of Dense_10's) are derived from ONE
                         sample-axis product h7^T dz10 and the weights, see k_bottleneck_grads in csrc/mlp_bwd.hip --
                         neither the bottleneck activations nor their gradients are exchanged through memory */);
/* The same with per-segment geometry: segment l of the sample axis has row capacity rows[l] (a multiple of 32: the
 * layout stride of its buffers), rows_per_ray[l] rows per ray and a nullable device ray count count[l] (valid rows =
 * count * rows_per_ray; a partial last tile is fine when the rows beyond the count carry zero dz and finite
 * operands).  Used for a de-duplicated batch: two sampling levels of the compacted rays + two one-row-per-ray
 * segments of the box-hit rays (durf_expand_raw).  Host arrays of nlevels entries. */
int durf_mlp_dw_levels(void* stream, int width, int nlevels, const size_t* rows, const int* rows_per_ray,
                       const int32_t* const* count, const void* const* enc_tile, const void* const* view_tile,
                       const void* const* stash, const void* const* dz, const void* const* dz_out, float* part,
                       float* bpart);
int durf_mlp_dw_finalize_levels(void* stream, int width, int in_dim, int nlevels, const size_t* rows,
                                const int* rows_per_ray, const int32_t* const* count, const float* part,
                                const float* bpart, float* grad_mlp, const float* mlp_params);

/* Background MLP without redundant work.  A ray that hits exactly one box feeds the background MLP the SAME trunk
 * input at every sample (obbpose_model.py:205-210 masks its Gaussians to zero -> encoding [0 x 30, 1 x 30]); only the
 * view direction differs, per ray.  So the background MLP is evaluated sample by sample on the other rays only
 * (compacted: durf_encode_bkgd / durf_mlp_fwd with idx, count) and ONCE per box-hit ray (durf_mlp_fwd with N = 1 on a
 * constant encoding: the "tail rows" of the same durf_mlp_fwd launch); this call rebuilds the reference's [B*N,4]
 * raw layout from the two: slot [B,2] from
 * durf_compact_hits on the two ray classes (column 0: position among the sample-by-sample rays or -1, column 1:
 * position among the once-per-ray rays).  Backward: durf_loss_bwd's draw_ray_sum is the head gradient of the single
 * evaluation (the MLP is the same function at every sample of such a ray, so the sum of the per-sample output
 * gradients is exactly what reaches its weights), and durf_mlp_dw_levels takes both kinds of segment. */
int durf_expand_raw(void* stream, int B, int N, const float* raw_c /* compacted rows, then the tail rows */,
                    const int32_t* count /* device int[2] of durf_compact_hits on the two classes */,
                    const int32_t* slot /* [B,2] */, float* raw_full /* [B*N,4] */,
                    const float* raw_tail /* nullable [B,4]: row slot[b,1] replaces the tail row of box-hit ray b -- the
                       fp32 evaluation of durf_mlp_fwd_f32(enc = NULL) when the object branch runs in fp32 */);

/* ---- the whole inference forward as ONE call (csrc/forward.hip) ----------------------------------------
 * MipNerfModel.__call__ (obbpose_model.py:68-261) as render_eval_fn runs it (train_boxpose.py:377-390): ray setup, per-object
 * hit lists, weight packing, and per level the background encoding + 8x256 MLP (rays that hit exactly one box evaluated
 * once: durf_expand_raw), the K object MLPs on their hit rays, merge + volumetric rendering, and the resampling that
 * feeds the next level -- the stage entry points above in the order durf_amd/obbpose_model.py issues them for
 * train=False, on `stream`, bf16 MLPs, results bit-identical to that path.  For hosts that are not Python: SURVEY 8b's
 * `durf_forward`.  All pointers are device pointers of caller-owned buffers except barf_w (host values inside the
 * struct); `workspace` holds every intermediate: durf_forward_workspace_bytes(B, N, K) bytes, 256-byte aligned, free
 * to reuse once the stream has passed the call.  density_noise is not applied (inference: randomized=False; with
 * t_rand / u_rand given the sampling is stratified as in training). */
#define DURF_FORWARD_MAX_LEVELS 4
typedef struct durf_forward_args {
    int B, N, K, num_levels;            /* rays, MipNerfModel.num_samples, boxes (0: static model), num_levels */
    int enc_flags;                      /* DURF_ENC_CONTRACT | DURF_ENC_NO_INTEGRATION | DURF_ENC_CYLINDER */
    int lindisp, bkgd_mode;             /* MipNerfModel.lindisp; 0 grey 0.5 / 1 white / 2 none (rand_bkgd: mip.py:324 draws the colour 0) */
    float density_bias, resample_padding;
    float barf_w[10];                   /* weighted_ipe's per-degree weights for this step's alpha (mip.py:217-218) */
    const float *origins, *directions, *viewdirs, *radii, *near, *far;     /* Rays fields, [B,3] x3, [B] x3 */
    const float *pose, *ext;            /* box_centers[ts] [K,6], half extents [K,3] */
    const float *bkgd_params;           /* MLP_0: Dense_0..11 (kernel[in,out], bias) flat */
    const float *obj_params;            /* BoxMLP_0 .. BoxMLP_{K-1}, obj_param_stride floats apart */
    size_t obj_param_stride;
    const float *t_rand, *u_rand;       /* nullable [B,N+1] each: randomized=False (or draw_noise) */
    /* outputs, per level: rgb [B,3], depth / acc [B], weights / t_mids / t_dists [B,N], t_vals [B,N+1] */
    float *rgb[DURF_FORWARD_MAX_LEVELS], *depth[DURF_FORWARD_MAX_LEVELS], *acc[DURF_FORWARD_MAX_LEVELS];
    float *weights[DURF_FORWARD_MAX_LEVELS], *t_vals[DURF_FORWARD_MAX_LEVELS], *t_mids[DURF_FORWARD_MAX_LEVELS];
    float *t_dists[DURF_FORWARD_MAX_LEVELS];
    int32_t* dyn_mask;                  /* [B] boxes hit per ray (the 10-tuple's dyn_mask) */
    float* zo;                          /* [B] */
    int draw_noise;                     /* != 0 (t_rand = u_rand = NULL): randomized=True with the draws made by the library */
    uint32_t seed_lo, seed_hi;          /*   -- durf_ray_prologue's Philox stream under this key (the host's PRNG key) */
    float density_noise;                /* randomized: MipNerfModel.density_noise, the std of the normal noise on the background's raw
                                           density (obbpose_model.py:236-240; durf_density_noise); 0: none */
    const float* density_rand[DURF_FORWARD_MAX_LEVELS];   /*   its standard-normal draws, [B,N] per level; NULL: the library's own,
                                                               under (seed_lo, seed_hi) */
} durf_forward_args;
size_t durf_forward_workspace_bytes(int B, int N, int K);
/* workspace_bytes: the size of the buffer `workspace` points to.  The call refuses (-1, durf_last_error names both sizes)
 * a buffer smaller than durf_forward_workspace_bytes(B, N, K) instead of carving its intermediates out of memory the
 * caller does not own. */
int durf_forward(void* stream, const durf_forward_args* args, void* workspace, size_t workspace_bytes);
/* One C call per IMAGE: render_image (obbpose_model.py:421-479) on one device -- the chunk loop the reference runs from Python
 * (one pmapped call + one host round trip per chunk, :446-475) over a ray buffer RESIDENT on the device.  `args` as for
 * durf_forward with the ray fields (origins, directions, viewdirs [n_rays,3]; radii, near, far [n_rays]) pointing at the whole
 * image and B, the per-level output pointers, dyn_mask and zo ignored (they live in the workspace, per chunk); test mode only
 * (no draws).  Chunks of `chunk` rays (the last one the remainder) run durf_forward's launch sequence; the LAST level's rgb
 * [n_rays,3], distance [n_rays] and acc [n_rays] are written in place -- bit-identical to render_image over durf_forward
 * chunks.  workspace: durf_render_image_workspace_bytes(chunk, N, K, num_levels) bytes, 256-byte aligned, size checked. */
size_t durf_render_image_workspace_bytes(int chunk, int N, int K, int num_levels);
int durf_render_image(void* stream, const durf_forward_args* args, size_t n_rays, int chunk, float* rgb, float* distance,
                      float* acc, void* workspace, size_t workspace_bytes);

/* ---- one shard's training step as ONE call (csrc/train.hip) ------------------------------------------------
 * durf_loss_backward: value_and_grad(loss_fn) of train_step (train_boxpose.py:67-252) -- the forward with activations
 * stashed, the losses, the backward, the weight gradients of every MLP, the reference's multi-hit outcome
 * (durf_poison_multi_hit) and the logged scalars -- in the order durf_amd/train_boxpose.py issues them, on `stream`,
 * bit-identical to that path.  A data-parallel host all-reduces `grad` (and `stats` when it logs) and calls
 * durf_clip_adam (lax.pmean, train_boxpose.py:253-255); durf_train_step = durf_loss_backward + durf_clip_adam with
 * inv_world = 1 for a single device.  Scope: every BASELINE.json training configuration -- bf16 background MLP; the K
 * object MLPs on the bf16 kernels with frozen box poses (flags = 0: box_centers get a zero gradient; cfg2 / cfg3 / cfg5),
 * or on the exact-fp32 kernels (DURF_TRAIN_OBJ_FP32: MipNerfModel.object_precision() == 'f32' -- the box-hit rays'
 * object MLPs, their encodings and the background MLP's one evaluation of those rays in fp32) with, under
 * DURF_TRAIN_POSE_OPT, the box-pose gradient behind them (cfg4: obbpose_model.py:99-131; want_pos = !no_pose_opt,
 * want_rot = !no_yaw_opt, the TV prior tv_loss_mult of train_boxpose.py:136,219 on the positions; f.pose must then be this
 * timestep's rows of box_centers INSIDE params: their gradient lands in the same rows of grad); >= 2 levels.  The knobs
 * off the shipped configs ride along: f.density_noise (obbpose_model.py:236-240), weight_decay_mult (train_boxpose.py:73-75),
 * f.bkgd_mode 0 / 1 / 2 with bg = 0.5 / 1.0 / 0 (2: Config.rand_bkgd, no background colour).  `f` carries the rays, boxes,
 * draws and -- as outputs -- each level's rendered values; f.bkgd_params / f.obj_params must point into `params`.
 * workspace: durf_train_workspace_bytes_flags(B, N, K, num_levels, n_params, flags) bytes, 256-byte aligned
 * (durf_train_workspace_bytes = flags 0).
 * Streams: everything is ordered on `stream`, whose device must be the calling thread's current device.  The bf16 object
 * MLPs of a large step (>= 2048 x 128 sample rows per level) run on a second, non-blocking stream the library creates per
 * device on first use, forked from / joined to `stream` with events inside the call (DURF_OVERLAP_OBJECTS=0: one stream);
 * when the call returns, all of its work is ordered before whatever the caller issues to `stream` next -- with ONE exception,
 * prefetch_const_trunk (below): that launch is left running on the side stream, reading `params` and writing const_trunk.  The
 * next durf_train_step / durf_loss_backward on the device joins it before it uses or recomputes a trunk and before its own
 * optimizer update; a caller that overwrites or frees `params` or `const_trunk` between two steps calls
 * durf_prefetch_join(stream) first.
 * Threading: the side stream and its two events are ONE set per DEVICE for the whole process (created on first use, never
 * destroyed).  Calls for two models on one device from two host threads are correct -- fork / join pairs are issued under a
 * mutex -- but each join waits for everything on the shared side stream, i.e. the two models' object launches serialise;
 * the entry points themselves keep no other state between calls (durf_last_error is thread-local). */
#define DURF_TRAIN_OBJ_FP32 1
#define DURF_TRAIN_POSE_OPT 2
#define DURF_TRAIN_OBJ_X3 4        /* with DURF_TRAIN_OBJ_FP32: that branch's forward / backward on split bf16 operands (durf_objf32_*_x3) */
/* Live timing of the step's dominant launches for a roofline line (bench.py): hipEvent_t handles (created with timing
 * enabled, any may be NULL) that the call records on `stream` right before / after the background MLP's forward and backward
 * launch of level l (DURF_TIMED_FWD + l, DURF_TIMED_BWD + l), the fused per-ray launch behind level l's forward
 * (DURF_TIMED_COMPOSITE + l) and the weight-gradient launch (DURF_TIMED_DW). */
#define DURF_TIMED_FWD 0
#define DURF_TIMED_BWD 4
#define DURF_TIMED_COMPOSITE 8
#define DURF_TIMED_DW 12
#define DURF_TIMED_STAGES 13
typedef struct durf_step_timing { void* begin[DURF_TIMED_STAGES]; void* end[DURF_TIMED_STAGES]; } durf_step_timing;
typedef struct durf_train_args {
    durf_forward_args f;
    const float *lossmult, *pixels, *gt_depth, *sky;     /* Rays.lossmult [B], batch pixels [B,3], depth [B], sky [B] */
    const float *target6, *prev6;                         /* batch target [K,6], prev[0] [K,6]: offsets / TV statistics */
    float eps, box_loss_mult, bg;                         /* near-loss interval; Config.box_loss_mult; background colour */
    int disable_multiscale;                               /* Config.disable_multiscale_loss */
    float level_mults[DURF_FORWARD_MAX_LEVELS][6];        /* per level: rgb, sky, depth, near, empty, distortion (train_boxpose.py:211-220) */
    float stat_mults[6];                                  /* coarse, sky, depth, near, empty, tv loss multipliers (durf_train_stats) */
    float* params;                                        /* flat parameters: box_centers | MLP_0 | K x BoxMLP */
    size_t n_params, box_floats, mlp0_floats, obj_floats;
    float* grad;                                          /* [n_params] out: d(loss)/d(params) of this shard, not post-processed */
    float* stats;                                         /* [2 + 17 num_levels] out: durf_train_stats layout */
    float *adam_m, *adam_v;                               /* durf_train_step only: Adam moments [n_params] */
    float lr, max_val, max_norm;                          /*   learning rate, Config.grad_max_val, Config.grad_max_norm */
    int step;                                             /*   optimizer step count (0 for the first update) */
    float* grad_stats;                                    /*   [4] out: grad_norm, grad_abs_max, clip multiplier, grad_norm_clipped */
    int flags;                                            /* DURF_TRAIN_OBJ_FP32 | DURF_TRAIN_POSE_OPT (0: the bf16 object branch, frozen poses) */
    int want_pos, want_rot;                               /* DURF_TRAIN_POSE_OPT: !no_pose_opt, !no_yaw_opt (obbpose_model.py:100-104) */
    float tv_loss_mult;                                   /*   Config.tv_loss_mult (position prior against prev6) */
    void* comm;                                           /* durf_train_step only, nullable: a durf_comm_init communicator -- the step then is
                                                             one rank's share of a data-parallel step: the multi-hit outcome, ONE in-stream
                                                             all-reduce of grad (durf_allreduce_sum), clip + Adam on the mean (1 / world) */
    int world;                                            /*   ranks of comm (1 / world scales the summed gradient; lax.pmean) */
    int reduce_stats;                                     /*   != 0: the logged scalars are averaged over the ranks too (lax.pmean(stats), :255) */
    float weight_decay_mult;                              /* Config.weight_decay_mult (train_boxpose.py:73-75; durf_weight_decay); 0: none */
    float* pose_used;                                     /* nullable out [K,6]: f.pose as the step rendered with it (snapshot taken by the first
                                                             launch; durf_train_step updates box_centers in place under pose optimisation) */
    int32_t* cls_count;                                   /* nullable out [8]: durf_compact_all's class counts, kept where the caller can read
                                                             them ([3] = rays that hit two boxes, utils.Stats.multi_hit_rays) instead of in the workspace */
    const durf_step_timing* timing;                       /* nullable: events to record around the dominant launches (above) */
    float* const_trunk;                                   /* DURF_TRAIN_OBJ_FP32, nullable [264]: the background trunk on the box-hit rays' constant
                                                             encoding (durf_bkgd_const_trunk_f32: a function of the parameters alone, one workgroup,
                                                             40-50 us), kept by the CALLER across steps so that it leaves the critical path: */
    int const_trunk_valid;                                /*   != 0: const_trunk holds the trunk of `params` as they are NOW -- the caller vouches that
                                                             nothing wrote them since the durf_train_step that prefetched it -- and the step uses it;
                                                             0: the step computes it (into const_trunk when given) */
    int prefetch_const_trunk;                             /*   != 0 (durf_train_step): behind the optimizer update the call starts the NEXT step's trunk
                                                             on its side stream into const_trunk; the next call waits for it before using it */
} durf_train_args;
size_t durf_train_workspace_bytes(int B, int N, int K, int num_levels, size_t n_params);
size_t durf_train_workspace_bytes_flags(int B, int N, int K, int num_levels, size_t n_params, int flags);
/* workspace_bytes: the size of the buffer behind `workspace`; a buffer smaller than
 * durf_train_workspace_bytes_flags(f.B, f.N, f.K, f.num_levels, n_params, flags) is refused (-1, durf_last_error). */
/* durf_prefetch_join: orders `stream` behind a prefetch_const_trunk launch still outstanding on this device's side stream
 * (no-op when there is none). */
int durf_prefetch_join(void* stream);
int durf_loss_backward(void* stream, const durf_train_args* args, void* workspace, size_t workspace_bytes);
int durf_train_step(void* stream, const durf_train_args* args, void* workspace, size_t workspace_bytes);

/* ---- the data-parallel exchange on the caller's stream (csrc/comm.hip) -------------------------------------
 * jax.lax.pmean(grad, 'batch') (train_boxpose.py:253) as ONE in-place RCCL all-reduce (sum; durf_clip_adam / durf_stats_scrub
 * fold the 1 / world into their scrub pass) of the flat fp32 gradient, issued by this library IN `stream` -- for hosts that
 * are not Python (durf_train_args.comm: the whole data-parallel step is then one C call per rank) and for hosts that do
 * not want the hop to a communication stream and back.  One process per GPU: rank 0 calls durf_comm_unique_id and hands the
 * DURF_COMM_ID_BYTES bytes to every rank by its own means (file, socket, its launcher's store); every rank then calls
 * durf_comm_init (collective: ncclCommInitRank) with its device current.  RCCL is resolved at run time, from the copy the
 * process has loaded already if there is one (durf_comm_available() == 0: none found). */
#define DURF_COMM_ID_BYTES 128
int durf_comm_available(void);
int durf_comm_unique_id(void* id_out /* HOST, DURF_COMM_ID_BYTES bytes */);
int durf_comm_init(int world, int rank, const void* id /* HOST */, void** comm_out /* HOST: receives the communicator */);
int durf_comm_destroy(void* comm);
int durf_allreduce_sum(void* stream, void* comm, float* buf /* device, in place */, size_t n);

/* ---- exact-fp32 MLP (csrc/mlp_f32.hip) -----------------------------------------------------------
 * The reference's Dense layers are fp32 (obbpose_model.py:326-327; HIGHEST-precision matmul, internal/math.py:22-24).
 * These entry points evaluate the same stack with v_mfma_f32_32x32x2_f32 (exact fp32: bitwise an fmaf chain over the
 * input features, bias first; 1/16 of the bf16 MFMA rate), reading the fp32 flax-layout parameters directly.  Two users:
 * the OBJECT BRANCH of a step with box-pose optimisation (durf_objf32_*: the box-hit rays of cfg4, whose pose gradient
 * does not survive bf16 rounding) and the parity instrument MipNerfModel(mlp_precision='f32').
 *   enc [rows,in_dim] row-major fp32 (durf_encode_*'s out_f32); NULL (width 256, in_dim 60 only): every row is the
 *       constant encoding of a zero-masked Gaussian [0 x 30, 1 x 30] -- the background MLP's ONE evaluation of a
 *       box-hit ray (durf_expand_raw), with N = 1 and ray_idx / count = that ray class;
 *   view27 [B,27] (durf_view_enc's out_f32), raw [rows,4];
 *   act / dz: per-sample records of the forward (the input of every Dense) / backward (d loss / d pre-activation of
 *       every Dense), durf_mlp_f32_{act,dz}_floats floats per row, ROWS ROUNDED UP TO 32, stored per 32-row tile as
 *       [tile][float index][32 rows];  opaque to the caller, only passed between these calls;
 *   wstream: durf_mlp_f32_pack's output, durf_mlp_f32_wstream_floats(width) floats per MLP -- the kernels of the ten
 *       wide Dense layers as the chunks the forward / the backward (transposed) consume, in order, zero-padded to whole
 *       tiles (re-packed after every optimizer step, like the bf16 weight streams); mlp_params still supplies biases and
 *       the 1- / 3-wide heads;  (width, in_dim) must be (256, 60) or (128, 63), the two MLPs of the model;
 *   d_enc [rows,64] row-major (nullable; every valid row is overwritten);  ray_idx / count as in durf_mlp_fwd.
 * Weight gradients are split over `nsplit` sample shares and summed in a fixed order (deterministic); scratch:
 * durf_mlp_f32_dw_scratch_floats floats.  grad_mlp: flax layout of one MLP, overwritten. */
size_t durf_mlp_f32_act_floats(int width, int in_dim);
size_t durf_mlp_f32_dz_floats(int width, int in_dim);
size_t durf_mlp_f32_dw_scratch_floats(int width, int in_dim, int nsplit);
size_t durf_mlp_f32_wstream_floats(int width);
int durf_mlp_f32_pack(void* stream, int width, int in_dim, int K, const float* mlp_params, size_t param_stride,
                      float* wstream /* [K, durf_mlp_f32_wstream_floats] */);
int durf_mlp_fwd_f32(void* stream, int width, int in_dim, size_t rows, int N, const float* enc, const float* view27,
                     const int32_t* ray_idx, const int32_t* count, const float* mlp_params, const float* wstream,
                     float* raw, float* act /* nullable: inference */);
int durf_mlp_bwd_f32(void* stream, int width, int in_dim, size_t rows, int N, const float* draw,
                     const int32_t* ray_idx, const int32_t* count, const float* mlp_params, const float* wstream,
                     const float* act, float* dz, float* d_enc /* nullable */);
int durf_mlp_dw_f32(void* stream, int width, int in_dim, size_t rows, int N, const int32_t* count, const float* act,
                    const float* dz, int nsplit, float* scratch, float* grad_mlp);
/* The background MLP (width 256, in_dim 60) on the box-hit rays in fp32 -- what durf_mlp_fwd_f32(enc = NULL) computes, in
 * two parts: Dense_0 .. Dense_9, whose input is the same for every such ray, ONCE per step (durf_bkgd_const_trunk_f32:
 * parameters -> trunk [257] = bottleneck, density), then the view layer and the rgb head per ray (durf_bkgd_hit_rays_f32:
 * idx / count = that ray class (durf_compact_classes, class 1); raw_tail [B,4], row j = ray idx[j]: durf_expand_raw's
 * raw_tail). */
int durf_bkgd_const_trunk_f32(void* stream, const float* bkgd_params, float* trunk);
int durf_bkgd_hit_rays_f32(void* stream, int B, const float* view27, const float* bkgd_params, const int32_t* idx,
                           const int32_t* count, const float* trunk, float* raw_tail);
/* The K object MLPs (width 128, in_dim 63) of one level on the fp32 kernels, ONE launch per phase with the object
 * index in the grid: idx [K,B] / count [K] from durf_compact_hits; enc [K, B*N, 63]; raw [K, B*N, 4]; act / dz
 * [K, durf_objf32_{act,dz}_stride floats]; d_enc [K, B*N, 64]; obj_params: BoxMLP_0 .. BoxMLP_{K-1}, param_stride
 * floats apart; wstream: durf_mlp_f32_pack(128, 63, K, ...)'s output.  durf_objf32_dw_batch takes the records of every level (host arrays of nlevels device
 * pointers) and writes grad_obj [K, grad_stride]; scratch: K * durf_mlp_f32_dw_scratch_floats(128, 63, nsplit). */
size_t durf_objf32_act_stride(int B, int N);
size_t durf_objf32_dz_stride(int B, int N);
int durf_encode_obj_f32_batch(void* stream, int K, int B, int N, const int32_t* idx, const int32_t* count,
                              const float* t_vals, const float* origins_s, const float* dirs_s, const float* radii,
                              const float* barf_w /* host float[10] */, int flags, float* enc /* [K, B*N, 63] */);
int durf_objf32_fwd_batch(void* stream, int K, int B, int N, const int32_t* idx, const int32_t* count,
                          const float* enc /* nullable: the kernel then encodes its own tiles from the ray data below,
                                              bit-identical to durf_encode_obj_f32_batch, one launch less per level */,
                          const float* view27, const float* obj_params, size_t param_stride, const float* wstream,
                          float* raw, float* act /* nullable: inference */, const float* t_vals, const float* origins_s,
                          const float* dirs_s, const float* radii, const float* barf_w /* host float[10] */, int flags);
int durf_objf32_bwd_batch(void* stream, int K, int B, int N, const int32_t* idx, const int32_t* count, const float* draw,
                          const float* obj_params, size_t param_stride, const float* wstream, const float* act,
                          float* dz, float* d_enc /* nullable */);
/* "bf16x3" (round 6): the same object forward / backward on the bf16 matrix pipe -- every MFMA operand a (hi, lo) pair of
 * bf16 (x = hi + lo: 16-17 significant bits), a product three v_mfma_f32_32x32x16_bf16 (hi.hi + hi.lo + lo.hi) in place of eight
 * v_mfma_f32_32x32x2_f32, fp32 accumulation, bias, ReLU, heads; records, streams' sizes and every other argument as the exact
 * kernels'.  ~2^-16 relative error per product instead of exact fp32 (the reference's HIGHEST-precision matmul,
 * internal/math.py:22-24): MipNerfModel.obj_precision = 'bf16x3'.  durf_mlp_f32_pack_x3 writes the W = 128 weight streams for
 * them; the forward is durf_objf32_fwd_batch with DURF_F32_X3 in `flags` (enc must be NULL: the self-encoding form). */
#define DURF_F32_X3 16
int durf_mlp_f32_pack_x3(void* stream, int K, const float* obj_params, size_t param_stride, float* wstream);
int durf_objf32_bwd_batch_x3(void* stream, int K, int B, int N, const int32_t* idx, const int32_t* count, const float* draw,
                             const float* obj_params, size_t param_stride, const float* wstream, const float* act,
                             float* dz, float* d_enc /* nullable */);
int durf_objf32_dw_batch(void* stream, int K, int B, int N, const int32_t* count, int nlevels, const float* const* act,
                         const float* const* dz, int nsplit, float* scratch, float* grad_obj, size_t grad_stride);

/* The K per-object BoxMLPs of one level as one call each (obbpose_model.py:174-201): every kernel of the
 * per-object path runs ONCE with the object index in blockIdx.y (csrc/objects.hip), on `stream`.
 * Slabs are [K, ...] with per-object strides: enc durf_obj_enc_stride, view_tile durf_obj_view_stride,
 * dz_out durf_obj_dzout_stride, stash/dz durf_mlp_stash_bytes(128, B*N), mask durf_mlp_mask_bytes(B*N),
 * raw B*N*4 floats, d_enc B*N*64 floats, weight packs durf_wpack_{fwd,bwd}_bytes(128), params / grads
 * `*_stride` floats (the flat buffer keeps BoxMLP_0, BoxMLP_1, ... back to back).  idx [K,B], count [K]
 * from durf_compact_hits.  The kernels are the ones the per-object entry points launch. */
size_t durf_obj_enc_stride(int B, int N);
size_t durf_obj_view_stride(int B, int N);
size_t durf_obj_dzout_stride(int B, int N);
int durf_pack_weights_batch(void* stream, int width, int in_dim, int K, const float* mlp_params,
                            size_t param_stride, void* wpack_fwd, void* wpack_bwd /* nullable */);
int durf_obj_fwd_batch(void* stream, int K, int B, int N, const int32_t* idx, const int32_t* count,
                       const float* t_vals, const float* origins_s, const float* dirs_s, const float* radii,
                       const float* barf_w, int flags, const void* view_bf16, const void* wpack_fwd,
                       void* enc, float* raw, void* stash /* nullable */, void* relu_mask /* nullable */,
                       void* view_tile /* nullable: durf_expand_view per object */);
int durf_obj_bwd_batch(void* stream, int K, int B, int N, const int32_t* idx, const int32_t* count,
                       const float* draw, const void* wpack_bwd, const void* relu_mask, void* dz, void* dz_out,
                       float* d_enc /* nullable */);
/* The same for EVERY level of a step at once (no d(enc)): stop_level_grad makes each level's d(raw) a function of the forward
 * alone, so all of them exist before the first backward launch; at small batches (the M-split kernel, < 2048 x 128 sample rows)
 * the step's object backward is then ONE latency-bound launch instead of one per level.  draw / relu_mask / dz / dz_out: host
 * arrays [nlevels] of the per-level device buffers durf_obj_bwd_batch takes.  Bit-identical to the per-level calls. */
int durf_obj_bwd_batch_levels(void* stream, int K, int B, int N, int nlevels, const int32_t* idx, const int32_t* count,
                              const float* const* draw, const void* wpack_bwd, const void* const* relu_mask, void* const* dz,
                              void* const* dz_out);
/* The backward counterpart of durf_mlp_fwd_enc_obj (round 6): durf_mlp_bwd(width 256, no d(enc)) of the background MLP +
 * durf_obj_bwd_batch_levels of the K object MLPs over `nlevels` levels (host arrays of per-level device buffers) as ONE
 * heterogeneous persistent launch -- background blocks, then (level, object, tile pair) items off a ticket counter on two
 * 4-wave groups per workgroup.  dz / dz_out of both classes bit-identical to the two calls, which it falls back to above
 * 2048 x 128 sample rows or under DURF_OBJ_MIX=0. */
int durf_mlp_bwd_obj(void* stream, size_t rows, int N, const float* draw, const int32_t* ray_idx, const int32_t* count,
                     const void* wpack_bwd, const void* relu_mask, void* dz, void* dz_out, const int32_t* tail_idx /* nullable */,
                     const int32_t* tail_count /* nullable */, const float* draw_ray_sum /* nullable */, int K, int B, int nlevels,
                     const int32_t* obj_idx, const int32_t* obj_count, const float* const* obj_draw, const void* obj_wpack_bwd,
                     const void* const* obj_relu_mask, void* const* obj_dz, void* const* obj_dz_out);
int durf_obj_dw_batch(void* stream, int K, int B, int N, const int32_t* count, int nlevels,
                      const void* const* enc, const void* const* view_tile, const void* const* stash,
                      const void* const* dz, const void* const* dz_out, int in_dim, float* part, float* bpart,
                      float* grad_mlp, size_t grad_stride, const float* mlp_params /* K MLPs, grad_stride floats apart */);
/* The same in two halves, so that a training step finalizes ALL its MLPs with one pair of launches: durf_obj_dw_partials
 * = the grouped split-K launch of durf_obj_dw_batch only; durf_dw_finalize_all = durf_mlp_dw_finalize_levels of the
 * background MLP (segments as in durf_mlp_dw_levels) and the finalize half of durf_obj_dw_batch (K may be 0) as ONE
 * k_dw_finalize + ONE k_bottleneck_grads launch (blockIdx.z / .y: the background MLP, then the K object MLPs). */
int durf_obj_dw_partials(void* stream, int K, int B, int N, const int32_t* count, int nlevels,
                         const void* const* enc, const void* const* view_tile, const void* const* stash,
                         const void* const* dz, const void* const* dz_out, float* part, float* bpart);
int durf_dw_finalize_all(void* stream, int in_bkgd, int nseg, const size_t* rows, const int* rows_per_ray,
                         const int32_t* const* seg_count, const float* part_bkgd, const float* bpart_bkgd,
                         float* grad_bkgd, const float* bkgd_params, int K, int B, int N, const int32_t* obj_count,
                         int nlevels, int in_obj, const float* part_obj, const float* bpart_obj, float* grad_obj,
                         size_t obj_grad_stride, const float* obj_params);

/* Box-pose gradients (cfg4): reverse of weighted_ipe / cast_rays / world2object_rpy / aa2matrix
 * (mip.py:182-223,155-179; box_helpers.py:286-341,148-167).  Per level and object:
 * durf_mlp_bwd(..., d_enc) then durf_encode_obj_bwd accumulates 21 per-object sums
 * (scratch: 21*B floats; sums [K,21], zeroed by the caller once per step); durf_pose_finish
 * turns them into d(loss)/d(box_centers[ts]) added to grad6 [K,6]
 * (want_pos = !no_pose_opt, want_rot = !no_yaw_opt, obbpose_model.py:100-104).  precise != 0: libm exp / sin / cos
 * (behind the fp32 object branch); 0: the hardware transcendentals (behind the bf16 object MLPs).  enc_flags: the
 * forward's DURF_ENC_CYLINDER (mip.py:133-152) | DURF_ENC_NO_INTEGRATION (obbpose_model.py:163-164: means only). */
int durf_encode_obj_bwd(void* stream, int B, int N, int k_obj, const int32_t* idx, const int32_t* count,
                        const float* d_enc, const float* t_vals, const float* origins_s,
                        const float* dirs_s, const float* radii, const float* origins, const float* dirs,
                        const float* pose, const float* barf_w, float* scratch, float* sums, int precise, int enc_flags);
/* The same for all K objects of a level in one launch pair (blockIdx.y = object): idx [K,B], count [K], d_enc [K, B*N, 64]
 * (the slab durf_obj_bwd_batch fills), pose [K,6], scratch K*21*B floats, sums [K,21] accumulated over levels. */
int durf_encode_obj_bwd_batch(void* stream, int K, int B, int N, const int32_t* idx, const int32_t* count,
                              const float* d_enc, const float* t_vals, const float* origins_s,
                              const float* dirs_s, const float* radii, const float* origins, const float* dirs,
                              const float* pose, const float* barf_w /* host float[10] */, float* scratch, float* sums,
                              int precise, int enc_flags);
/* ... and for EVERY level of a step in one launch pair (blockIdx.z = level; the reduction adds the levels' row sums in the
 * order given, the bits of one call per level): d_enc / t_vals host arrays [nlevels] of the per-level device buffers,
 * scratch nlevels * K*21*B floats. */
int durf_encode_obj_bwd_levels(void* stream, int K, int B, int N, int nlevels, const int32_t* idx, const int32_t* count,
                               const float* const* d_enc, const float* const* t_vals, const float* origins_s,
                               const float* dirs_s, const float* radii, const float* origins, const float* dirs,
                               const float* pose, const float* barf_w /* host float[10] */, float* scratch, float* sums,
                               int precise, int enc_flags);
int durf_pose_finish(void* stream, int K, const float* pose, const float* sums, int want_pos, int want_rot,
                     float* grad6);

/* K12 gradient post-processing + Adam on the flat buffers (train_boxpose.py:257-289;
 * flax.optim.Adam).  grad is scaled by inv_world (pmean), scrubbed and clipped in place;
 * stats[4] = {grad_norm, grad_abs_max, clip multiplier, grad_norm_clipped}. */
size_t durf_optim_scratch_floats(size_t n);
/* The tail of a training step in TWO launches instead of four: durf_stats_scrub = durf_train_stats (same arguments, same
 * scalars in `out`) + the first pass of durf_clip_adam over grad[0..n) (pmean scale inv_world, nan_to_num, value clip, the
 * partials of the global norm in `scratch`, durf_optim_scratch_floats(n) floats) as ONE launch, with the multi-hit outcome
 * of durf_poison_multi_hit folded into that pass when cls_count is given (single device: with a data-parallel all-reduce
 * the NaNs have to exist BEFORE it, i.e. durf_poison_multi_hit stays a call of its own there); durf_adam_apply = the Adam
 * pass of durf_clip_adam behind it.  Parameters, moments, gradient, scalars: bit-identical to the four separate calls. */
int durf_stats_scrub(void* stream, int L, int K, int N, const float* norms, float* sums, const float* weight_l2 /* nullable */,
                     const float* pose6, const float* prev6, const float* target6, const float* const* t_vals,
                     const float* mults, int mode, float* out, const float* const* terms /* nullable */, int B, size_t n,
                     float* grad, float inv_world, float max_val, float* scratch, const int32_t* cls_count /* nullable */,
                     size_t box_floats, int K_boxes, size_t mlp0_floats, size_t obj_floats);
int durf_adam_apply(void* stream, size_t n, float* params, float* m, float* v, const float* grad, float max_norm, float lr,
                    int step, const float* scratch, float* stats);
/* Reference semantics of rays that hit two boxes (obbpose_model.py:120-122: NaN colours -> NaN loss -> NaN gradient of
 * everything their path touches -> nan_to_num -> 0, train_boxpose.py:263): when cls_count[3] (durf_compact_classes) is
 * non-zero, the gradient segments of MLP_0 and of the boxes in cls_count[4] (their BoxMLP and their box_centers columns)
 * are set to NaN -- call it on the complete local gradient BEFORE the data-parallel all-reduce; durf_clip_adam scrubs.
 * Flat layout: box_centers (box_floats = T*K*6) | MLP_0 | K object MLPs of obj_floats each; n = the floats of grad this
 * call covers: the whole buffer, or a PREFIX of it (e.g. box_centers | MLP_0 when the objects' slice was poisoned earlier
 * and is already being all-reduced).  Exits at once when no ray hits two boxes. */
int durf_poison_multi_hit(void* stream, size_t n, float* grad, const int32_t* cls_count, size_t box_floats, int K,
                          size_t mlp0_floats, size_t obj_floats);
int durf_clip_adam(void* stream, size_t n, float* params, float* m, float* v, float* grad, float inv_world,
                   float max_val, float max_norm, float lr, int step, float* scratch, float* stats);
/* Config.weight_decay_mult (train_boxpose.py:73-75: loss += mult * mean(theta^2) over every parameter): grad[lo:hi) +=
 * (2 mult / n) params[lo:hi) -- a range, because a bucketed exchange hands the object MLPs' slice over before the rest of the
 * gradient exists -- and, when weight_l2 != NULL, weight_l2[0] = mult * mean(params^2) over all n (summed in a fixed order;
 * durf_train_stats adds it to the loss).  scratch: durf_optim_scratch_floats(n) floats. */
int durf_weight_decay(void* stream, size_t n, const float* params, float* grad, size_t lo, size_t hi, float mult,
                      float* scratch /* nullable with weight_l2 */, float* weight_l2 /* nullable */);

/* ---- callers / data either side of the hot path ---------------------------------
 * Rays of a 'timestep' batch generated on the device (obbpose_dataset.py:1868-1916 pinhole rays with
 * un-normalised directions and row-neighbour radii; :1551-1583 gather from the timestep's concatenated
 * cameras).  cams_host: n_cams x 17 HOST floats {camtoworld 3x4 row-major, focal, cx, cy, h, w};
 * ray_idx [B] device indices into the concatenation (NULL: 0..B-1); images [P,img_channels] / depth [P] /
 * sky [P] device arrays of the same concatenation (nullable, with their outputs). */
int durf_gen_batch(void* stream, int B, int n_cams, const float* cams_host, const int32_t* ray_idx, float near,
                   float far, const float* images, const float* depth, const float* sky, int img_channels,
                   float* origins, float* directions, float* viewdirs, float* radii, float* lossmult,
                   float* near_out, float* far_out, float* pixels, float* depth_out, float* sky_out);
/* SSIM of two [H,W,C] fp32 images (internal/math.py:66-137): filt_dev = the normalised 1-D Gaussian
 * window (filter_size device floats), scratch durf_ssim_scratch_floats floats, ssim_map nullable
 * [(H-fs+1),(W-fs+1),C], ssim_mean 1 device float. */
size_t durf_ssim_scratch_floats(int H, int W, int C, int filter_size);
int durf_ssim(void* stream, int H, int W, int C, const float* img0, const float* img1, float max_val,
              int filter_size, const float* filt_dev, float k1, float k2, float* ssim_map, float* scratch,
              float* ssim_mean);

#ifdef __cplusplus
}
#endif
#endif
