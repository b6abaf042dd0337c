// Per-ray reductions: alpha compositing (K8) and hierarchical resampling (K9).
// One 64-lane wavefront owns one ray; samples are split into contiguous runs of P per
// lane, so the transmittance prefix is a lane-local scan plus one wave-level scan.
// Both kernels are HBM-bound (3.1 KB / 1.5 KB per ray-level) and read each input once.
#include "loss_common.h"

struct ObjPtrs { const float* p[DURF_MAX_OBJ]; };


// ---------------------------------------------------------------------------
// K8 composite forward: obbpose_model.py:232-245 + mip.volumetric_rendering (mip.py:285-327)
// One ray per wave; lane l owns samples [l*P, l*P+P).  Shared by k_composite_fwd and the fused
// k_composite_resample, so both produce the same bits.
// ---------------------------------------------------------------------------
template <int P>
__device__ __forceinline__ void composite_ray(int b, int lane, int N, int K, const float* __restrict__ raw_bkgd,
                                              const ObjPtrs& raw_obj, const int32_t* __restrict__ slot,
                                              const float* __restrict__ tv, float dnorm, float density_bias,
                                              float (&w)[P], float (&tm)[P], float (&td)[P], float (&s_rgb)[3],
                                              float& s_acc, float& s_dep) {
    float a[P], c[P][3];
    float run = 0.0f;
#pragma unroll
    for (int p = 0; p < P; p++) {
        const int n = lane * P + p;
        a[p] = 0.0f; c[p][0] = c[p][1] = c[p][2] = 0.0f; tm[p] = 0.0f; td[p] = 0.0f;
        if (n < N) {
            const f32x4 rb = *(const f32x4*)(raw_bkgd + ((size_t)b * N + n) * 4);
            float o0 = 0.f, o1 = 0.f, o2 = 0.f, o3 = 0.f;
            for (int k = 0; k < K; k++) {
                const int s = slot[b * K + k];
                if (s >= 0) {
                    const f32x4 ro = *(const f32x4*)(raw_obj.p[k] + ((size_t)s * N + n) * 4);
                    o0 += ro[0]; o1 += ro[1]; o2 += ro[2]; o3 += ro[3];
                }
            }
            const float t0 = tv[n], t1 = tv[n + 1];
            tm[p] = 0.5f * (t0 + t1);
            td[p] = t1 - t0;
            c[p][0] = sigmoidf_(rb[0] + o0);
            c[p][1] = sigmoidf_(rb[1] + o1);
            c[p][2] = sigmoidf_(rb[2] + o2);
            const float dens = softplusf_((rb[3] + o3) + density_bias);
            a[p] = dens * (td[p] * dnorm);
        }
        run += a[p];
    }
    // exclusive prefix of a over the whole ray
    const float incl = wave_incl_scan(run, lane);
    float pre = incl - run;
    s_rgb[0] = s_rgb[1] = s_rgb[2] = 0.f; s_acc = 0.f; s_dep = 0.f;
#pragma unroll
    for (int p = 0; p < P; p++) {
        const int n = lane * P + p;
        const float alpha = 1.0f - expf(-a[p]);
        const float trans = expf(-pre);
        w[p] = nan_to_num(alpha * trans);
        pre += a[p];
        if (n < N) {
            s_rgb[0] += w[p] * c[p][0]; s_rgb[1] += w[p] * c[p][1]; s_rgb[2] += w[p] * c[p][2];
            s_acc += w[p];
            s_dep += w[p] * tm[p];
        }
    }
    s_rgb[0] = wave_sum(s_rgb[0]); s_rgb[1] = wave_sum(s_rgb[1]); s_rgb[2] = wave_sum(s_rgb[2]);
    s_acc = wave_sum(s_acc);
    s_dep = wave_sum(s_dep);
}

template <int P>
__device__ __forceinline__ void composite_store(int b, int lane, int N, int bkgd_mode, const float (&w)[P],
                                                const float (&tm)[P], const float (&td)[P], const float (&s_rgb)[3],
                                                float s_acc, float s_dep, float* __restrict__ rgb_out,
                                                float* __restrict__ depth_out, float* __restrict__ acc_out,
                                                float* __restrict__ weights, float* __restrict__ t_mids,
                                                float* __restrict__ t_dists) {
#pragma unroll
    for (int p = 0; p < P; p++) {
        const int n = lane * P + p;
        if (n < N) {
            if (weights) weights[(size_t)b * N + n] = w[p];
            if (t_mids) t_mids[(size_t)b * N + n] = tm[p];
            if (t_dists) t_dists[(size_t)b * N + n] = td[p];
        }
    }
    if (lane == 0) {
        float bg = 0.0f;                       // rand_bkgd: randint(key,(1,3),0,1) == 0 (mip.py:324)
        if (bkgd_mode == 0) bg = 0.5f;
        else if (bkgd_mode == 1) bg = 1.0f;
        const float rem = 1.0f - s_acc;
        if (rgb_out) {
            rgb_out[b * 3 + 0] = s_rgb[0] + bg * rem;
            rgb_out[b * 3 + 1] = s_rgb[1] + bg * rem;
            rgb_out[b * 3 + 2] = s_rgb[2] + bg * rem;
        }
        if (depth_out) depth_out[b] = s_dep;
        if (acc_out) acc_out[b] = s_acc;
    }
}

template <int P>
__global__ void __launch_bounds__(256)
k_composite_fwd(int B, int N, int K, const float* __restrict__ raw_bkgd, ObjPtrs raw_obj,
                const int32_t* __restrict__ slot, const float* __restrict__ t_vals,
                const float* __restrict__ dirs_s, float density_bias, int bkgd_mode,
                float* __restrict__ rgb_out, float* __restrict__ depth_out,
                float* __restrict__ acc_out, float* __restrict__ weights,
                float* __restrict__ t_mids, float* __restrict__ t_dists) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (b >= B) return;
    const float dx = dirs_s[b * 3], dy = dirs_s[b * 3 + 1], dz = dirs_s[b * 3 + 2];
    const float dnorm = sqrtf(dx * dx + dy * dy + dz * dz);
    float w[P], tm[P], td[P], s_rgb[3], s_acc, s_dep;
    composite_ray<P>(b, lane, N, K, raw_bkgd, raw_obj, slot, t_vals + (size_t)b * (N + 1), dnorm, density_bias,
                     w, tm, td, s_rgb, s_acc, s_dep);
    composite_store<P>(b, lane, N, bkgd_mode, w, tm, td, s_rgb, s_acc, s_dep, rgb_out, depth_out, acc_out, weights,
                       t_mids, t_dists);
}

// ---------------------------------------------------------------------------
// K9 resample: blur-pool + padding (mip.py:393-404), sorted_piecewise_constant_pdf
// (math.py:222-284).  The reference's [B,N+1,N+1] mask compare is a search in a sorted
// CDF; here each lane binary-searches the wave's CDF held in LDS.
// sw (the ray's weights) and bins (its t_vals) are in LDS when this is entered (all four waves of the
// workgroup call it: it synchronises).  Returns, when gt_eps_dm is given, the minimum of near_d2 over the
// first N resampled values (the NEXT level's interval starts, loss_common.h).
// ---------------------------------------------------------------------------
#define RS_MAXN 256
template <int P>
__device__ __forceinline__ float resample_ray(bool active, int b, int lane, int N, const float* sw, float* cdf,
                                              const float* bins, float padding, const float* __restrict__ u_rand,
                                              float* __restrict__ t_out, const float* gt_eps_dm, bool blur = true) {
    float pw[P];
    float tot = 0.0f;
    // (this lane's draws, requested before the pdf / cdf arithmetic and the barrier below instead of inside the search loop)
    constexpr int NU = (RS_MAXN + 1 + 63) / 64;
    float uv[NU];
#pragma unroll
    for (int q = 0; q < NU; q++) {
        const int j = lane + 64 * q;
        uv[q] = (u_rand && active && j <= N) ? u_rand[(size_t)b * (N + 1) + j] : 0.0f;
    }
#pragma unroll
    for (int p = 0; p < P; p++) {
        const int n = lane * P + p;
        pw[p] = 0.0f;
        if (active && n < N) {
            const float wl = sw[n > 0 ? n - 1 : 0], wc = sw[n], wr = sw[n < N - 1 ? n + 1 : N - 1];
            // blur-pool + padding of mip.resample_along_rays (mip.py:393-404); without it this is
            // math.sorted_piecewise_constant_pdf on the given weights
            pw[p] = blur ? 0.5f * (fmaxf(wl, wc) + fmaxf(wc, wr)) + padding : wc;
            tot += pw[p];
        }
    }
    float wsum = wave_sum(tot);
    const float pad = fmaxf(0.0f, 1e-5f - wsum);                        // math.py:237-241
    wsum += pad;
    float run = 0.0f;
#pragma unroll
    for (int p = 0; p < P; p++) {
        const int n = lane * P + p;
        pw[p] = (n < N) ? (pw[p] + pad / (float)N) / wsum : 0.0f;       // pdf
        run += pw[p];
    }
    float pre = wave_incl_scan(run, lane) - run;
    if (active) {
#pragma unroll
        for (int p = 0; p < P; p++) {
            const int n = lane * P + p;
            pre += pw[p];
            if (n < N - 1) cdf[n + 1] = fminf(1.0f, pre);               // math.py:246
        }
        if (lane == 0) { cdf[0] = 0.0f; cdf[N] = 1.0f; }
    }
    __syncthreads();
    float mind2 = __builtin_inff();
    if (!active) return mind2;
    const float one_m_eps = 0.99999988079071045f;                        // 1 - finfo(float32).eps
    const int num = N + 1;
#pragma unroll
    for (int q = 0; q < NU; q++) {
        const int j = lane + 64 * q;
        if (j >= num) break;
        float u;
        if (u_rand) {                                                    // math.py:254-262
            const float s = 1.0f / (float)num;
            u = (float)j * s + uv[q] * (s - 1.1920928955078125e-07f);
            u = fminf(u, one_m_eps);
        } else {                                                         // linspace(0, 1-eps, num)
            u = (j == num - 1) ? one_m_eps : one_m_eps * ((float)j / (float)(num - 1));
        }
        // largest i in [0,N] with cdf[i] <= u  (cdf[0] = 0 <= u, cdf[N] = 1 > u)
        int lo = 0, hi = N;
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (cdf[mid] <= u) lo = mid; else hi = mid;
        }
        const float c0 = cdf[lo], c1 = cdf[lo + 1];
        float t = nan_to_num((u - c0) / (c1 - c0));
        t = fminf(fmaxf(t, 0.0f), 1.0f);
        const float b0 = bins[lo], b1 = bins[lo + 1];
        const float tn = b0 + t * (b1 - b0);
        t_out[(size_t)b * num + j] = tn;
        if (gt_eps_dm && j < N) mind2 = fminf(mind2, near_d2(tn, gt_eps_dm[0], gt_eps_dm[1], gt_eps_dm[2]));
    }
    return mind2;
}

template <int P>
__global__ void __launch_bounds__(256)
k_resample(int B, int N, const float* __restrict__ t_vals, const float* __restrict__ w_in,
           float padding, const float* __restrict__ u_rand, float* __restrict__ t_out, int blur) {
    __shared__ float s_w[4][RS_MAXN + 2];
    __shared__ float s_cdf[4][RS_MAXN + 2];
    __shared__ float s_bins[4][RS_MAXN + 2];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int b = blockIdx.x * 4 + wv;
    const bool active = b < B;
    float* sw = s_w[wv];
    float* bins = s_bins[wv];
    if (active) {
        for (int n = lane; n < N; n += 64) sw[n] = w_in[(size_t)b * N + n];
        for (int n = lane; n <= N; n += 64) bins[n] = t_vals[(size_t)b * (N + 1) + n];
    }
    __syncthreads();
    resample_ray<P>(active, b, lane, N, sw, s_cdf[wv], bins, padding, u_rand, t_out, nullptr, blur != 0);
}

// ---------------------------------------------------------------------------
// K8 + K9 (+ the per-ray part of loss_fn's normalisers) in ONE launch for a level that is followed by
// another: composite, hand the ray's weights to the resampler through LDS, emit the next level's t_vals.
// With `lossmult` given it also writes the loss-prep rows (loss_common.h) of the NEXT level (from the
// resampled t_vals) and, if prep_this is given, of this level -- what durf_loss_prep computes, so a
// training step needs no separate launch for them.
// ---------------------------------------------------------------------------
struct PrepArgs {
    const float* lossmult; const float* gt_depth; const float* sky; const int32_t* dyn; const float* zo;
    float eps, box_loss_mult; int level, disable_multiscale;
    float* prep_this; float* prep_next;
    // round 6: the batch reduction of the prep rows (the loss normalisers: k_reduce_rows, a launch of its own until then) is
    // done by whichever workgroup finishes LAST -- counter: a zeroed int the launch leaves zeroed (durf::next_ticket)
    float* norm_this; float* norm_next; int* counter;
};

template <int P>
__global__ void __launch_bounds__(256)
k_composite_resample(int B, int N, int K, const float* __restrict__ raw_bkgd, ObjPtrs raw_obj,
                     const int32_t* __restrict__ slot, const float* __restrict__ t_vals,
                     const float* __restrict__ dirs_s, float density_bias, int bkgd_mode,
                     float* __restrict__ rgb_out, float* __restrict__ depth_out, float* __restrict__ acc_out,
                     float* __restrict__ weights, float* __restrict__ t_mids, float* __restrict__ t_dists,
                     float padding, const float* __restrict__ u_rand, float* __restrict__ t_out, PrepArgs pa) {
    __shared__ float s_w[4][RS_MAXN + 2];
    __shared__ float s_cdf[4][RS_MAXN + 2];
    __shared__ float s_bins[4][RS_MAXN + 2];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int b = blockIdx.x * 4 + wv;
    const bool active = b < B;
    float* sw = s_w[wv];
    float* bins = s_bins[wv];
    float gt_eps_dm[3] = {0.f, 0.f, 0.f};
    RayMasks rm_next = {};
    float dynf = 0.0f;
    // (the per-ray scalars of the loss prep, requested HERE with the ray's other inputs: they used to be fetched behind the
    // composite's stores -- one more dependent round trip in a kernel that is a chain of them)
    float p_gt = 0.f, p_lm = 0.f, p_sky = 0.f, p_zo = 0.f;
    int p_dyn = 0;
    if (active && pa.lossmult) { p_gt = pa.gt_depth[b]; p_lm = pa.lossmult[b]; p_sky = pa.sky[b]; p_zo = pa.zo[b]; p_dyn = pa.dyn[b]; }
    if (active) {
        const float dx = dirs_s[b * 3], dy = dirs_s[b * 3 + 1], dz = dirs_s[b * 3 + 2];
        const float dnorm = sqrtf(dx * dx + dy * dy + dz * dz);
        const float* tv = t_vals + (size_t)b * (N + 1);
        float w[P], tm[P], td[P], s_rgb[3], s_acc, s_dep;
        composite_ray<P>(b, lane, N, K, raw_bkgd, raw_obj, slot, tv, dnorm, density_bias, w, tm, td, s_rgb, s_acc, s_dep);
        composite_store<P>(b, lane, N, bkgd_mode, w, tm, td, s_rgb, s_acc, s_dep, rgb_out, depth_out, acc_out, weights,
                           t_mids, t_dists);
#pragma unroll
        for (int p = 0; p < P; p++) {
            const int n = lane * P + p;
            if (n < N) sw[n] = w[p];
        }
        for (int n = lane; n <= N; n += 64) bins[n] = tv[n];
        if (pa.lossmult) {
            LossCfg c = {};
            c.eps = pa.eps; c.box_loss_mult = pa.box_loss_mult; c.disable_multiscale = pa.disable_multiscale;
            const float gt = p_gt;
            dynf = (float)p_dyn;
            if (pa.prep_this) {
                c.level = pa.level;
                const RayMasks r = ray_masks(c, p_lm, gt, p_sky, dynf, p_zo);
                float mind2 = __builtin_inff();
                for (int n = lane; n < N; n += 64) mind2 = fminf(mind2, near_d2(tv[n], gt, c.eps, r.dm));
                mind2 = wave_min(mind2);
                if (lane == 0) write_prep(pa.prep_this, B, b, r, mind2, dynf);
            }
            c.level = pa.level + 1;
            rm_next = ray_masks(c, p_lm, gt, p_sky, dynf, p_zo);
            gt_eps_dm[0] = gt; gt_eps_dm[1] = c.eps; gt_eps_dm[2] = rm_next.dm;
        }
    }
    __syncthreads();
    float mind2 = resample_ray<P>(active, b, lane, N, sw, s_cdf[wv], bins, padding, u_rand, t_out,
                                  pa.lossmult ? gt_eps_dm : nullptr);
    if (active && pa.lossmult) {
        mind2 = wave_min(mind2);
        if (lane == 0) write_prep(pa.prep_next, B, b, rm_next, mind2, dynf);
    }
    if (pa.counter) {
        // the LAST workgroup to get here reduces the prep rows over the batch (k_reduce_rows' sums, bit for bit): every
        // workgroup publishes its rows (release), takes a number; the last one sees them all (acquire)
        __shared__ int s_last;
        __shared__ float s16[16];
        __syncthreads();
        if (threadIdx.x == 0) {
            __threadfence();
            s_last = (__hip_atomic_fetch_add(pa.counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (int)gridDim.x - 1);
        }
        __syncthreads();
        if (s_last) {
            __threadfence();
            if (pa.prep_this && pa.prep_next == pa.prep_this + (size_t)PREP_ROWS * B && pa.norm_next == pa.norm_this + PREP_ROWS) {
                reduce_rows_256(2 * PREP_ROWS, B, (int)PREP_MIND2, pa.prep_this, pa.norm_this, s16);
            } else {
                if (pa.prep_this) reduce_rows_256(PREP_ROWS, B, (int)PREP_MIND2, pa.prep_this, pa.norm_this, s16);
                reduce_rows_256(PREP_ROWS, B, (int)PREP_MIND2, pa.prep_next, pa.norm_next, s16);
            }
            if (threadIdx.x == 0) __hip_atomic_store(pa.counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// De-duplicated background evaluation (see durf_expand_raw in include/durf_hip.h): raw of the full [B*N,4] layout
// from the compacted rows of the rays evaluated sample by sample and, after them, the ONE row of each box-hit ray.
__global__ void __launch_bounds__(256)
k_expand_raw(int B, int N, const float* __restrict__ raw_c, const int32_t* __restrict__ count,
             const int32_t* __restrict__ slot, float* __restrict__ raw_full, const float* __restrict__ raw_tail) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)B * N) return;
    const int b = (int)(i / N), n = (int)(i % N);
    const int s0 = slot[b * 2], s1 = slot[b * 2 + 1];
    const size_t src = s0 >= 0 ? (size_t)s0 * N + n : (size_t)count[0] * N + (size_t)s1;
    // raw_tail: the box-hit rays' one background evaluation redone in fp32 (object branch in fp32)
    const float* from = (s0 < 0 && raw_tail) ? raw_tail + (size_t)s1 * 4 : raw_c + src * 4;
    *(f32x4*)(raw_full + i * 4) = *(const f32x4*)from;
}

extern "C" {

int durf_expand_raw(void* stream, int B, int N, const float* raw_c, const int32_t* count, const int32_t* slot,
                    float* raw_full, const float* raw_tail) {
    if (B <= 0) return 0;
    hipLaunchKernelGGL(k_expand_raw, dim3(durf_cdiv((size_t)B * N, 256)), dim3(256), 0, (hipStream_t)stream, B, N, raw_c,
                       count, slot, raw_full, raw_tail);
    DURF_CHECK_LAUNCH("durf_expand_raw");
    return 0;
}

int durf_composite_fwd(void* stream, int B, int N, int K, const float* raw_bkgd,
                       const float* const* raw_obj, const int32_t* slot, const float* t_vals,
                       const float* dirs_s, float density_bias, int bkgd_mode, float* rgb,
                       float* depth, float* acc, float* weights, float* t_mids, float* t_dists) {
    DURF_REQUIRE(N >= 1 && N <= 256, "1 <= N <= 256");
    DURF_REQUIRE(K >= 0 && K <= DURF_MAX_OBJ, "K <= DURF_MAX_OBJ");
    if (B <= 0) return 0;
    ObjPtrs op;
    for (int k = 0; k < DURF_MAX_OBJ; k++) op.p[k] = (k < K) ? raw_obj[k] : nullptr;
    dim3 grid(durf_cdiv(B, 4)), block(256);
    hipStream_t s = (hipStream_t)stream;
#define LAUNCH_C(P)                                                                              \
    hipLaunchKernelGGL(k_composite_fwd<P>, grid, block, 0, s, B, N, K, raw_bkgd, op, slot, t_vals, \
                       dirs_s, density_bias, bkgd_mode, rgb, depth, acc, weights, t_mids, t_dists)
    if (N <= 64) LAUNCH_C(1); else if (N <= 128) LAUNCH_C(2); else LAUNCH_C(4);
#undef LAUNCH_C
    DURF_CHECK_LAUNCH("durf_composite_fwd");
    return 0;
}

int durf_composite_resample(void* stream, int B, int N, int K, const float* raw_bkgd, const float* const* raw_obj,
                            const int32_t* slot, const float* t_vals, const float* dirs_s, float density_bias,
                            int bkgd_mode, float* rgb, float* depth, float* acc, float* weights, float* t_mids,
                            float* t_dists, float resample_padding, const float* u_rand, float* t_vals_out,
                            const float* lossmult, const float* gt_depth, const float* sky, const int32_t* dyn,
                            const float* zo, float eps, float box_loss_mult, int level, int disable_multiscale,
                            float* prep_this, float* norm_this, float* prep_next, float* norm_next) {
    DURF_REQUIRE(N >= 2 && N <= RS_MAXN, "2 <= N <= 256");
    DURF_REQUIRE(K >= 0 && K <= DURF_MAX_OBJ, "K <= DURF_MAX_OBJ");
    DURF_REQUIRE(!lossmult || (gt_depth && sky && dyn && zo && prep_next && norm_next), "loss prep needs all its inputs");
    DURF_REQUIRE(!prep_this || (lossmult && norm_this), "prep_this needs the loss-prep inputs and norm_this");
    if (B <= 0) return 0;
    ObjPtrs op;
    for (int k = 0; k < DURF_MAX_OBJ; k++) op.p[k] = (k < K) ? raw_obj[k] : nullptr;
    PrepArgs pa = {lossmult, gt_depth, sky, dyn, zo, eps, box_loss_mult, level, disable_multiscale, prep_this, prep_next,
                   norm_this, norm_next, nullptr};
    {   // DURF_PREP_REDUCE_INKERNEL=1: the batch reduction by the last workgroup of this launch instead of k_reduce_rows
        // launches behind it (the same sums).  Measured and NOT the default (profiles/r06_mix.txt): the release / acquire
        // fences every workgroup pays (L2 write-back + invalidate at device scope) and the one-workgroup reduction tail cost
        // 26-30 us a launch against 8-12 + 5.5 for the two launches.
        const char* e = getenv("DURF_PREP_REDUCE_INKERNEL");
        if (lossmult && e && e[0] == '1') pa.counter = durf::next_ticket();
    }
    dim3 grid(durf_cdiv(B, 4)), block(256);
    hipStream_t s = (hipStream_t)stream;
#define LAUNCH_CR(P)                                                                                       \
    hipLaunchKernelGGL(k_composite_resample<P>, grid, block, 0, s, B, N, K, raw_bkgd, op, slot, t_vals, dirs_s, \
                       density_bias, bkgd_mode, rgb, depth, acc, weights, t_mids, t_dists, resample_padding, \
                       u_rand, t_vals_out, pa)
    if (N <= 64) LAUNCH_CR(1); else if (N <= 128) LAUNCH_CR(2); else LAUNCH_CR(4);
#undef LAUNCH_CR
    if (lossmult && pa.counter == nullptr) {
        // prep_this and prep_next may be adjacent ([2, PREP_ROWS, B]) with adjacent norms: one reduction launch
        if (prep_this && prep_next == prep_this + (size_t)PREP_ROWS * B && norm_next == norm_this + PREP_ROWS) {
            durf::launch_reduce_rows(s, 2 * PREP_ROWS, B, (int)PREP_MIND2, prep_this, norm_this);
        } else {
            if (prep_this) durf::launch_reduce_rows(s, PREP_ROWS, B, (int)PREP_MIND2, prep_this, norm_this);
            durf::launch_reduce_rows(s, PREP_ROWS, B, (int)PREP_MIND2, prep_next, norm_next);
        }
    }
    DURF_CHECK_LAUNCH("durf_composite_resample");
    return 0;
}

int durf_resample(void* stream, int B, int N, const float* t_vals, const float* weights,
                  float resample_padding, const float* u_rand, float* t_vals_out) {
    DURF_REQUIRE(N >= 2 && N <= RS_MAXN, "2 <= N <= 256");
    if (B <= 0) return 0;
    dim3 grid(durf_cdiv(B, 4)), block(256);
    hipStream_t s = (hipStream_t)stream;
#define LAUNCH_R(P)                                                                              \
    hipLaunchKernelGGL(k_resample<P>, grid, block, 0, s, B, N, t_vals, weights, resample_padding, \
                       u_rand, t_vals_out, 1)
    if (N <= 64) LAUNCH_R(1); else if (N <= 128) LAUNCH_R(2); else LAUNCH_R(4);
#undef LAUNCH_R
    DURF_CHECK_LAUNCH("durf_resample");
    return 0;
}

int durf_sorted_piecewise_constant_pdf(void* stream, int B, int N, const float* bins, const float* weights,
                                       const float* u_rand, float* samples) {
    DURF_REQUIRE(N >= 2 && N <= RS_MAXN, "2 <= N <= 256");
    if (B <= 0) return 0;
    dim3 grid(durf_cdiv(B, 4)), block(256);
    hipStream_t s = (hipStream_t)stream;
#define LAUNCH_R(P) hipLaunchKernelGGL(k_resample<P>, grid, block, 0, s, B, N, bins, weights, 0.0f, u_rand, samples, 0)
    if (N <= 64) LAUNCH_R(1); else if (N <= 128) LAUNCH_R(2); else LAUNCH_R(4);
#undef LAUNCH_R
    DURF_CHECK_LAUNCH("durf_sorted_piecewise_constant_pdf");
    return 0;
}

}  // extern "C"
