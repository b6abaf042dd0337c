// Losses (K10) fused with the backward of activations + alpha compositing (part of K11).
// train_boxpose.py:94-220 (loss_fn) and the reverse-mode of mip.volumetric_rendering
// (mip.py:285-327) / obbpose_model.py:243-245.
//
// One wavefront per ray, same sample->lane split as the forward composite: the
// transmittance prefix and the gradient's suffix sum are wave-level scans.  The reference's
// O(N^2) distortion term  sum_ij w_i w_j |s_i - s_j|  is evaluated in O(N) with two more
// prefix sums (s is sorted along a ray).
//
// Flow per level:  durf_loss_prep  -> per-ray {m, dm, sm, min dist^2, dyn}  -> durf_reduce_rows
//                  -> norm[] (device)  -> durf_loss_bwd -> d(raw) [B*N,4] + per-ray loss terms
#include "loss_common.h"
#include "optim_scrub.h"

struct ObjPtrsL { const float* p[DURF_MAX_OBJ]; };

// rows of the per-ray loss-term buffer
enum { LT_RGB = 0, LT_OBJ = 1, LT_DEPTH = 2, LT_NEAR = 3, LT_EMPTY = 4, LT_SKY = 5, LT_DIST = 6, LT_ROWS = 7 };

__global__ void __launch_bounds__(256)
k_loss_prep(int B, int N, const float* __restrict__ t_vals, const float* __restrict__ lossmult,
            const float* __restrict__ gt_depth, const float* __restrict__ sky,
            const int32_t* __restrict__ dyn, const float* __restrict__ zo, LossCfg c,
            float* __restrict__ prep) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    const float gt = gt_depth[b];
    const float dynf = (float)dyn[b];
    const RayMasks r = ray_masks(c, lossmult[b], gt, sky[b], dynf, zo[b]);
    float mind2 = __builtin_inff();
    for (int n = lane; n < N; n += 64) mind2 = fminf(mind2, near_d2(t_vals[(size_t)b * (N + 1) + n], gt, c.eps, r.dm));
    mind2 = wave_min(mind2);
    if (lane == 0) write_prep(prep, B, b, r, mind2, dynf);
}

// deterministic row reduction: out[r] = sum (or min if r % PREP_ROWS == min_row) of in[r*n .. r*n+n)
__global__ void __launch_bounds__(1024)
k_reduce_rows(int n, int min_row, const float* __restrict__ in, float* __restrict__ out) {
    __shared__ float s[16];
    const int r = blockIdx.x;
    const bool is_min = (min_row >= 0 && r % PREP_ROWS == min_row);
    const float* p = in + (size_t)r * n;
    float v = is_min ? __builtin_inff() : 0.0f;
    for (int i = threadIdx.x; i < n; i += 1024) v = is_min ? fminf(v, p[i]) : v + p[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float t = __shfl_xor(v, o, 64);
        v = is_min ? fminf(v, t) : v + t;
    }
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        float a = s[0];
        for (int w = 1; w < 16; w++) a = is_min ? fminf(a, s[w]) : a + s[w];
        out[r] = a;
    }
}

// one ray (one wave) of one level: the body of k_loss_bwd and of k_loss_bwd_levels
template <int P>
__device__ __forceinline__ void
loss_bwd_ray(int b, int lane, int B, int N, int K, const float* __restrict__ raw_bkgd, const ObjPtrsL& raw_obj,
             const int32_t* __restrict__ slot, const float* __restrict__ t_vals,
             const float* __restrict__ dirs_s, const float* __restrict__ pixels,
             const float* __restrict__ lossmult, const float* __restrict__ gt_depth,
             const float* __restrict__ sky, const int32_t* __restrict__ dyn,
             const float* __restrict__ zo, const float* __restrict__ norm, const LossCfg& c,
             float* __restrict__ draw, float* __restrict__ terms, float* __restrict__ rgb_out,
             float* __restrict__ depth_out, float* __restrict__ acc_out, float* __restrict__ weights_out,
             float* __restrict__ t_mids_out, float* __restrict__ t_dists_out, float* __restrict__ draw_ray_sum) {
    const float dx = dirs_s[b * 3], dy = dirs_s[b * 3 + 1], dz = dirs_s[b * 3 + 2];
    const float dnorm = sqrtf(dx * dx + dy * dy + dz * dz);
    const float* tv = t_vals + (size_t)b * (N + 1);
    const float gt = gt_depth[b], skyv = sky[b], dynf = (float)dyn[b];
    const RayMasks rm = ray_masks(c, lossmult[b], gt, skyv, dynf, zo[b]);
    const float sum_m = norm[PREP_M];
    const float D = fmaxf(norm[PREP_DM], 1.0f);
    const float Ssky = fmaxf(norm[PREP_SM], 1.0f);
    const float sig = (c.eps / 3.0f) * (c.eps / 3.0f);            // :156
    const float two_sig2 = 2.0f * sig * sig;
    const float inv_max = 1.0f / expf(-(norm[PREP_MIND2] / two_sig2));   // distr /= distr.max() (:164)

    // ---- recompute the forward composite (identical arithmetic to k_composite_fwd) ----
    float a[P], col[P][3], rawd[P], tm[P], td[P], tl[P];
    float run = 0.0f;
#pragma unroll
    for (int p = 0; p < P; p++) {
        const int n = lane * P + p;
        a[p] = 0.f; col[p][0] = col[p][1] = col[p][2] = 0.f; rawd[p] = 0.f; tm[p] = 0.f; td[p] = 0.f; tl[p] = 0.f;
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
            tl[p] = t0;
            tm[p] = 0.5f * (t0 + t1);
            td[p] = t1 - t0;
            col[p][0] = sigmoidf_(rb[0] + o0);
            col[p][1] = sigmoidf_(rb[1] + o1);
            col[p][2] = sigmoidf_(rb[2] + o2);
            rawd[p] = (rb[3] + o3) + c.density_bias;
            a[p] = softplusf_(rawd[p]) * (td[p] * dnorm);
        }
        run += a[p];
    }
    float pre = wave_incl_scan(run, lane) - run;
    float w[P], Tn[P];
    float s_rgb[3] = {0.f, 0.f, 0.f}, s_acc = 0.f, s_dep = 0.f;
    float wrun = 0.f, wsrun = 0.f;
#pragma unroll
    for (int p = 0; p < P; p++) {
        const int n = lane * P + p;
        Tn[p] = expf(-pre);
        const float alpha = 1.0f - expf(-a[p]);
        w[p] = (n < N) ? nan_to_num(alpha * Tn[p]) : 0.0f;
        pre += a[p];
        s_rgb[0] += w[p] * col[p][0]; s_rgb[1] += w[p] * col[p][1]; s_rgb[2] += w[p] * col[p][2];
        s_acc += w[p];
        s_dep += w[p] * tm[p];
        wrun += w[p];
        wsrun += w[p] * tm[p];
    }
#pragma unroll
    for (int i = 0; i < 3; i++) s_rgb[i] = wave_sum(s_rgb[i]);
    s_acc = wave_sum(s_acc);
    s_dep = wave_sum(s_dep);
    const float rem = 1.0f - s_acc;
    const float rgb[3] = {s_rgb[0] + c.bg * rem, s_rgb[1] + c.bg * rem, s_rgb[2] + c.bg * rem};
    const float depth = s_dep;
    // the level's rendered outputs (what k_composite_fwd would write: same arithmetic), so that a training step
    // needs no separate composite launch for a level that is not resampled from
    if (weights_out) {
#pragma unroll
        for (int p = 0; p < P; p++) {
            const int n = lane * P + p;
            if (n < N) {
                weights_out[(size_t)b * N + n] = w[p];
                if (t_mids_out) t_mids_out[(size_t)b * N + n] = tm[p];
                if (t_dists_out) t_dists_out[(size_t)b * N + n] = td[p];
            }
        }
    }
    if (lane == 0) {
        if (rgb_out) { rgb_out[b * 3 + 0] = rgb[0]; rgb_out[b * 3 + 1] = rgb[1]; rgb_out[b * 3 + 2] = rgb[2]; }
        if (depth_out) depth_out[b] = depth;
        if (acc_out) acc_out[b] = s_acc;
    }

    // ---- per-ray loss terms and their gradients wrt rgb / depth ----
    const float m_eff = rm.m + c.box_loss_mult * dynf * rm.box;                     // :191
    float g_rgb[3], sq = 0.f;
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const float e = rgb[i] - pixels[b * 3 + i];
        sq += e * e;
        g_rgb[i] = c.c_rgb * 2.0f * m_eff * e / sum_m;
    }
    const float ed = depth - gt;
    float g_depth = c.c_depth * 2.0f * rm.dm * ed / D;                                // :174-175
    const float xs = rm.sm * depth;
    const float mx = fmaxf(xs, 1.0f);
    const float sd = rm.sm * (1.0f - 1.0f / mx);                                      // :186
    const float es = sd - skyv;
    const float dsd = (xs > 1.0f) ? rm.sm * rm.sm / (xs * xs) : 0.0f;
    g_depth += c.c_sky * 2.0f * rm.sm * es * dsd / Ssky;

    // ---- distortion: prefix sums of w and w*s over the ray ----
    const float W_incl = wave_incl_scan(wrun, lane), WS_incl = wave_incl_scan(wsrun, lane);
    const float W_tot = __shfl(W_incl, 63, 64), WS_tot = __shfl(WS_incl, 63, 64);
    float Wlt = W_incl - wrun, WSlt = WS_incl - wsrun;    // sums over samples before this lane's run

    float gw[P];
    float t_near = 0.f, t_empty = 0.f, t_dist = 0.f, gwsum = 0.f;
#pragma unroll
    for (int p = 0; p < P; p++) {
        const int n = lane * P + p;
        gw[p] = 0.0f;
        if (n < N) {
            const float t = tl[p];
            const float ind = (t > gt - c.eps && t < gt + c.eps) ? 1.0f : 0.0f;      // :158-161
            const float near = ind * rm.dm;
            const float empty = ((t > gt + c.eps) ? 1.0f : 0.0f) * rm.dm;
            const float dist = near * (t - gt);
            const float gval = (expf(-(dist * dist / two_sig2)) * inv_max) * near;    // :163-165
            const float rn = near * w[p] - gval;
            const float re = empty * w[p];
            t_near += rn * rn;
            t_empty += re * re;
            float g = c.c_near * 2.0f * near * rn / D + c.c_empty * 2.0f * empty * re / D;
            // distortion (:146-153): sum_ij w_i w_j |s_i - s_j| + (1/3) sum w_i^2 dt_i
            const float s = tm[p];
            const float left = s * Wlt - WSlt;                                  // sum_{j<i} w_j (s_i - s_j)
            const float Wgt = W_tot - Wlt - w[p], WSgt = WS_tot - WSlt - w[p] * s;
            const float right = WSgt - s * Wgt;                                 // sum_{j>i} w_j (s_j - s_i)
            t_dist += 2.0f * w[p] * left + (1.0f / 3.0f) * w[p] * w[p] * td[p];
            g += c.c_dist * (2.0f * (left + right) + (2.0f / 3.0f) * w[p] * td[p]);
            // through rgb, depth (acc only enters through the background term of rgb)
            g += g_rgb[0] * (col[p][0] - c.bg) + g_rgb[1] * (col[p][1] - c.bg) + g_rgb[2] * (col[p][2] - c.bg);
            g += g_depth * tm[p];
            gw[p] = g;
            Wlt += w[p];
            WSlt += w[p] * s;
        }
        gwsum += gw[p] * w[p];
    }
    // suffix sum of g_m w_m over m > n
    float suf = wave_incl_rscan(gwsum, lane) - gwsum;     // samples after this lane's run
    float loc[P];
    {
        float acc_loc = 0.f;
#pragma unroll
        for (int p = P - 1; p >= 0; p--) { loc[p] = acc_loc; acc_loc += gw[p] * w[p]; }
    }
    float rs[4] = {0.f, 0.f, 0.f, 0.f};       // sum over the ray's samples of d(loss)/d(raw): see draw_ray_sum
#pragma unroll
    for (int p = 0; p < P; p++) {
        const int n = lane * P + p;
        if (n < N) {
            // w_n = (1 - e^{-a_n}) T_n ; T_m = exp(-sum_{k<m} a_k)
            const float da = gw[p] * Tn[p] * expf(-a[p]) - (suf + loc[p]);
            const float ddens = da * (td[p] * dnorm);
            f32x4 o;
            o[0] = g_rgb[0] * w[p] * col[p][0] * (1.0f - col[p][0]);
            o[1] = g_rgb[1] * w[p] * col[p][1] * (1.0f - col[p][1]);
            o[2] = g_rgb[2] * w[p] * col[p][2] * (1.0f - col[p][2]);
            o[3] = ddens * sigmoidf_(rawd[p]);                 // softplus' = sigmoid
            *(f32x4*)(draw + ((size_t)b * N + n) * 4) = o;
            rs[0] += o[0]; rs[1] += o[1]; rs[2] += o[2]; rs[3] += o[3];
        }
    }
    // A ray whose background samples all feed the MLP the same input (a box-hit ray: obbpose_model.py:205-210 masks
    // their Gaussians to zero) contributes to the background MLP's gradients only through this per-ray sum.
    if (draw_ray_sum) {
#pragma unroll
        for (int i = 0; i < 4; i++) rs[i] = wave_sum(rs[i]);
        if (lane == 0) {
            const f32x4 o = {rs[0], rs[1], rs[2], rs[3]};
            *(f32x4*)(draw_ray_sum + (size_t)b * 4) = o;
        }
    }
    t_near = wave_sum(t_near);
    t_empty = wave_sum(t_empty);
    t_dist = wave_sum(t_dist);
    if (lane == 0) {
        terms[(size_t)LT_RGB * B + b] = m_eff * sq;
        terms[(size_t)LT_OBJ * B + b] = dynf * sq;
        terms[(size_t)LT_DEPTH * B + b] = rm.dm * ed * ed;
        terms[(size_t)LT_NEAR * B + b] = t_near;
        terms[(size_t)LT_EMPTY * B + b] = t_empty;
        terms[(size_t)LT_SKY * B + b] = rm.sm * es * es;
        terms[(size_t)LT_DIST * B + b] = t_dist;
    }
}

template <int P>
__global__ void __launch_bounds__(256)
k_loss_bwd(int B, int N, int K, const float* __restrict__ raw_bkgd, ObjPtrsL raw_obj,
           const int32_t* __restrict__ slot, const float* __restrict__ t_vals,
           const float* __restrict__ dirs_s, const float* __restrict__ pixels,
           const float* __restrict__ lossmult, const float* __restrict__ gt_depth,
           const float* __restrict__ sky, const int32_t* __restrict__ dyn,
           const float* __restrict__ zo, const float* __restrict__ norm, LossCfg c,
           float* __restrict__ draw, float* __restrict__ terms, float* __restrict__ rgb_out,
           float* __restrict__ depth_out, float* __restrict__ acc_out, float* __restrict__ weights_out,
           float* __restrict__ t_mids_out, float* __restrict__ t_dists_out, float* __restrict__ draw_ray_sum) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    loss_bwd_ray<P>(b, lane, B, N, K, raw_bkgd, raw_obj, slot, t_vals, dirs_s, pixels, lossmult, gt_depth, sky, dyn, zo, norm, c,
                    draw, terms, rgb_out, depth_out, acc_out, weights_out, t_mids_out, t_dists_out, draw_ray_sum);
}

// Every level's loss + composite backward as ONE launch (blockIdx.y = level): stop_level_grad makes each level's loss
// gradient a function of the forward alone, so on one stream the launches of the levels below the last only sat between
// two backward kernels (a 512-ray step: two launches of ~8 us).  Same body per (level, ray): bit-identical outputs.
struct LossLevelK {
    const float* raw_bkgd; ObjPtrsL raw_obj; const float* t_vals; const float* norm; LossCfg c;
    float *draw, *terms, *rgb_out, *depth_out, *acc_out, *weights_out, *t_mids_out, *t_dists_out, *draw_ray_sum;
};
struct LossLevelsK { LossLevelK l[DURF_MAX_LEVELS]; };

template <int P>
__global__ void __launch_bounds__(256)
k_loss_bwd_levels(int B, int N, int K, const int32_t* __restrict__ slot, const float* __restrict__ dirs_s,
                  const float* __restrict__ pixels, const float* __restrict__ lossmult, const float* __restrict__ gt_depth,
                  const float* __restrict__ sky, const int32_t* __restrict__ dyn, const float* __restrict__ zo, LossLevelsK A) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    const LossLevelK& a = A.l[blockIdx.y];
    loss_bwd_ray<P>(b, lane, B, N, K, a.raw_bkgd, a.raw_obj, slot, a.t_vals, dirs_s, pixels, lossmult, gt_depth, sky, dyn, zo,
                    a.norm, a.c, a.draw, a.terms, a.rgb_out, a.depth_out, a.acc_out, a.weights_out, a.t_mids_out, a.t_dists_out,
                    a.draw_ray_sum);
}


// ---------------------------------------------------------------------------
// Scalars of utils.Stats from the per-level sums (train_boxpose.py:123-249, 291-292): one launch
// instead of ~70 tiny elementwise/reduction kernels.  Layout of `out` (L = levels):
//   [0] loss | 15 rows of L: losses, obj_losses, d, n, e, s, distr, tv, offsets, offset_x/y/z,
//   offset_yaw, psnrs, obj_psnrs | 2L sampling stats (t_vals[0,0], t_vals[0,N] per level) | weight_l2
// mode bit 0: assemble everything but the PSNRs; bit 1: PSNRs from the (possibly all-reduced) losses.
// ---------------------------------------------------------------------------
struct TvalPtrs { const float* p[DURF_MAX_LEVELS]; };
struct StatMults { float coarse, sky, depth, near, empty, tv; };

__device__ __forceinline__ void
train_stats_body(int L, int K, int N, const float* __restrict__ norms, float* __restrict__ sums,
                 const float* __restrict__ weight_l2, const float* __restrict__ pose6,
                 const float* __restrict__ prev6, const float* __restrict__ target6, const TvalPtrs& tv, const TvalPtrs& terms, int B,
                 const StatMults& m, int mode, float* __restrict__ out) {
    // terms given: the per-ray loss terms [LT_ROWS, B] of every level are reduced HERE (the order of k_reduce_rows, so
    // the sums are the ones durf_loss_bwd's own reduction launch would have produced) and left in sums [L, LT_ROWS]
    // (all rows at once: the 1024 threads form the same strided partial sums, xor-shuffle trees and in-order sum of the 16
    // waves as k_reduce_rows -- bit-identical sums -- but row by row that was 2 barriers and a dependent round trip to
    // memory per row, 34 us a launch; here every thread has the loads of 14 rows in flight together and there is one barrier)
    __shared__ float s_sum[DURF_MAX_LEVELS * LT_ROWS];
    __shared__ float s_w[14][16];
    const bool reduce_here = (mode & 1) && terms.p[0] != nullptr;
    // the small inputs of thread 0's assembly below, fetched by many threads at once and under the reduction's loads (thread 0
    // alone walked them as a chain of dependent round trips)
    __shared__ float s_norm[DURF_MAX_LEVELS * 5], s_pose[DURF_MAX_OBJ * 6], s_prev[DURF_MAX_OBJ * 6], s_tgt[DURF_MAX_OBJ * 6],
        s_tv[DURF_MAX_LEVELS * 2];
    if (mode & 1) {
        const int t = threadIdx.x;
        if (t < L * 5) s_norm[t] = norms[t];
        if (t >= 64 && t < 64 + K * 6) { s_pose[t - 64] = pose6[t - 64]; s_prev[t - 64] = prev6[t - 64]; s_tgt[t - 64] = target6[t - 64]; }
        if (t >= 128 && t < 128 + 2 * L) s_tv[t - 128] = tv.p[(t - 128) >> 1][((t - 128) & 1) ? N : 0];
    }
    if (reduce_here) {
        const int nrow = L * LT_ROWS;
        for (int r0 = 0; r0 < nrow; r0 += 14) {
            float v[14];
            const float* p[14];
#pragma unroll
            for (int r = 0; r < 14; r++) {
                const int row = r0 + r < nrow ? r0 + r : nrow - 1;
                p[r] = terms.p[row / LT_ROWS] + (size_t)(row % LT_ROWS) * B;
                v[r] = 0.0f;
            }
            // (four strides of the batch per trip: 56 loads in flight, added in the same order -- one stride at a time a
            // 4096-ray batch was four round trips to memory in sequence)
            int i = threadIdx.x;
            for (; i + 3 * 1024 < B; i += 4 * 1024) {
                float x[4][14];
#pragma unroll
                for (int u = 0; u < 4; u++)
#pragma unroll
                    for (int r = 0; r < 14; r++) x[u][r] = p[r][i + u * 1024];
#pragma unroll
                for (int u = 0; u < 4; u++)
#pragma unroll
                    for (int r = 0; r < 14; r++) v[r] += x[u][r];
            }
            for (; i < B; i += 1024) {
#pragma unroll
                for (int r = 0; r < 14; r++) v[r] += p[r][i];
            }
#pragma unroll
            for (int r = 0; r < 14; r++) {
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) v[r] += __shfl_xor(v[r], o, 64);
                if ((threadIdx.x & 63) == 0) s_w[r][threadIdx.x >> 6] = v[r];
            }
            __syncthreads();
            if (threadIdx.x < 14 && r0 + (int)threadIdx.x < nrow) {
                float a = s_w[threadIdx.x][0];
                for (int w = 1; w < 16; w++) a += s_w[threadIdx.x][w];
                s_sum[r0 + threadIdx.x] = a;
                sums[r0 + threadIdx.x] = a;
            }
            __syncthreads();
        }
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    float* rows = out + 1;
    if (mode & 1) {
        const float* pose6 = s_pose; const float* prev6 = s_prev; const float* target6 = s_tgt; const float* norms = s_norm;
        float sq_prev = 0.f, sq_t = 0.f, sx = 0.f, sy = 0.f, sz = 0.f, syaw = 0.f;
        for (int k = 0; k < K; k++) {
            for (int j = 0; j < 3; j++) {
                const float dp = pose6[k * 6 + j] - prev6[k * 6 + j];
                sq_prev += dp * dp;
                const float dt = pose6[k * 6 + j] - target6[k * 6 + j];
                sq_t += dt * dt;
                if (j == 0) sx += dt * dt;
                if (j == 1) sy += dt * dt;
                if (j == 2) sz += dt * dt;
                const float dy = pose6[3 + j] - target6[k * 6 + 3 + j];      // box_rot of object 0 vs every target
                syaw += dy * dy;
            }
        }
        const float wl2 = weight_l2 ? *weight_l2 : 0.0f;
        float loss = wl2;
        for (int l = 0; l < L; l++) {
            const float* nr = norms + l * 5;
            const float* sm = reduce_here ? s_sum + l * 7 : sums + l * 7;
            const float D = fmaxf(nr[1], 1.0f), S = fmaxf(nr[2], 1.0f);
            const float v[13] = {sm[0] / nr[0], sm[1] / nr[4], sm[2] / D, sm[3] / D, sm[4] / D, sm[5] / S, sm[6],
                                 sq_prev, sq_t, sx, sy, sz, syaw};
            for (int r = 0; r < 13; r++) rows[r * L + l] = (K > 0 || r < 7) ? v[r] : 0.0f;
            if (K == 0) rows[7 * L + l] = 0.0f;
            const bool last = l == L - 1;
            loss += (last ? 1.0f : m.coarse) * v[0] + (last ? 10.0f : 1.0f) * m.sky * v[5]
                  + (last ? 1.0f : 0.1f) * (m.depth * v[2] + m.near * v[3] + m.empty * v[4] + m.tv * (K > 0 ? sq_prev : 0.0f))
                  + 0.000001f * v[6];
            out[1 + 15 * L + 2 * l] = s_tv[2 * l];
            out[1 + 15 * L + 2 * l + 1] = s_tv[2 * l + 1];
        }
        out[0] = loss;
        out[1 + 17 * L] = wl2;
    }
    if (mode & 2) {
        for (int l = 0; l < L; l++) {
            rows[13 * L + l] = -4.342944819032518f * logf(rows[0 * L + l]);      // -10/ln(10) * ln(mse)
            rows[14 * L + l] = -4.342944819032518f * logf(rows[1 * L + l]);
        }
    }
}

__global__ void __launch_bounds__(1024)
k_train_stats(int L, int K, int N, const float* __restrict__ norms, float* __restrict__ sums,
              const float* __restrict__ weight_l2, const float* __restrict__ pose6,
              const float* __restrict__ prev6, const float* __restrict__ target6, TvalPtrs tv, TvalPtrs terms, int B,
              StatMults m, int mode, float* __restrict__ out) {
    train_stats_body(L, K, N, norms, sums, weight_l2, pose6, prev6, target6, tv, terms, B, m, mode, out);
}

// The tail of a training step as ONE launch instead of three: workgroup 0 assembles the logged scalars (k_train_stats'
// body), the others run the optimizer's scrub pass over the flat gradient, four 256-thread virtual blocks each
// (optim_scrub.h: k_grad_scrub's body, same partials), with the multi-hit outcome (k_poison_multi_hit) folded in.  The
// two halves share nothing; as launches of their own they were 4.7 + 10 + 6 us of a 0.45-0.7 ms step.
__global__ void __launch_bounds__(1024)
k_stats_scrub(int L, int K, int N, const float* __restrict__ norms, float* __restrict__ sums,
              const float* __restrict__ weight_l2, const float* __restrict__ pose6,
              const float* __restrict__ prev6, const float* __restrict__ target6, TvalPtrs tv, TvalPtrs terms, int B,
              StatMults m, int mode, float* __restrict__ out, size_t n, float* __restrict__ g, float inv_world, float max_val,
              float* __restrict__ part, PoisonArgs pa) {
    if (blockIdx.x == 0) {
        train_stats_body(L, K, N, norms, sums, weight_l2, pose6, prev6, target6, tv, terms, B, m, mode, out);
        return;
    }
    __shared__ float s_sq[16], s_mx[16];
    const int sub = threadIdx.x >> 8, tid = threadIdx.x & 255;
    const size_t vb = (size_t)(blockIdx.x - 1) * 4 + sub;
    const size_t nvb = (n + OPT_BLOCK * OPT_PER_THREAD - 1) / (OPT_BLOCK * OPT_PER_THREAD);
    scrub_vblock(n, g, inv_world, max_val, part, vb, tid, s_sq + 4 * sub, s_mx + 4 * sub, pa, vb < nvb);
}

namespace durf {
void launch_reduce_rows(hipStream_t s, int rows, int n, int min_row, const float* in, float* out) {
    hipLaunchKernelGGL(k_reduce_rows, dim3(rows), dim3(1024), 0, s, n, min_row, in, out);
}
}  // namespace durf

extern "C" {

int durf_loss_prep(void* stream, int B, int N, const float* t_vals, const float* lossmult,
                   const float* gt_depth, const float* sky, const int32_t* dyn, const float* zo,
                   float eps, float box_loss_mult, int level, int disable_multiscale, float* prep,
                   float* norm) {
    if (B <= 0) return 0;
    LossCfg c = {};
    c.eps = eps; c.box_loss_mult = box_loss_mult; c.level = level; c.disable_multiscale = disable_multiscale;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_loss_prep, dim3(durf_cdiv(B, 4)), dim3(256), 0, s, B, N, t_vals, lossmult, gt_depth,
                       sky, dyn, zo, c, prep);
    durf::launch_reduce_rows(s, PREP_ROWS, B, (int)PREP_MIND2, prep, norm);
    DURF_CHECK_LAUNCH("durf_loss_prep");
    return 0;
}

int durf_loss_bwd(void* stream, int B, int N, int K, const float* raw_bkgd, const float* const* raw_obj,
                  const int32_t* slot, const float* t_vals, const float* dirs_s, const float* pixels,
                  const float* lossmult, const float* gt_depth, const float* sky, const int32_t* dyn,
                  const float* zo, const float* norm, float eps, const float* mults /*6: rgb,sky,depth,near,empty,dist*/,
                  float box_loss_mult, int level, int disable_multiscale, float bg, float density_bias,
                  float* draw, float* terms, float* term_sums, float* rgb_out, float* depth_out, float* acc_out,
                  float* weights_out, float* t_mids_out, float* t_dists_out, float* draw_ray_sum) {
    DURF_REQUIRE(N >= 1 && N <= 256, "1 <= N <= 256");
    if (B <= 0) return 0;
    LossCfg c;
    c.eps = eps; c.c_rgb = mults[0]; c.c_sky = mults[1]; c.c_depth = mults[2]; c.c_near = mults[3];
    c.c_empty = mults[4]; c.c_dist = mults[5]; c.box_loss_mult = box_loss_mult; c.level = level;
    c.bg = bg; c.density_bias = density_bias; c.disable_multiscale = disable_multiscale;
    ObjPtrsL op;
    for (int k = 0; k < DURF_MAX_OBJ; k++) op.p[k] = (k < K) ? raw_obj[k] : nullptr;
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(durf_cdiv(B, 4)), block(256);
#define LAUNCH_L(P)                                                                                   \
    hipLaunchKernelGGL(k_loss_bwd<P>, grid, block, 0, s, B, N, K, raw_bkgd, op, slot, t_vals, dirs_s,  \
                       pixels, lossmult, gt_depth, sky, dyn, zo, norm, c, draw, terms, rgb_out, depth_out, acc_out, \
                       weights_out, t_mids_out, t_dists_out, draw_ray_sum)
    if (N <= 64) LAUNCH_L(1); else if (N <= 128) LAUNCH_L(2); else LAUNCH_L(4);
#undef LAUNCH_L
    if (term_sums) hipLaunchKernelGGL(k_reduce_rows, dim3(LT_ROWS), dim3(1024), 0, s, B, -1, terms, term_sums);
    DURF_CHECK_LAUNCH("durf_loss_bwd");
    return 0;
}

int durf_loss_bwd_levels(void* stream, int B, int N, int K, int L, const durf_loss_level* levels, const int32_t* slot,
                         const float* dirs_s, const float* pixels, const float* lossmult, const float* gt_depth,
                         const float* sky, const int32_t* dyn, const float* zo, float eps, float box_loss_mult,
                         int disable_multiscale, float bg, float density_bias) {
    DURF_REQUIRE(N >= 1 && N <= 256, "1 <= N <= 256");
    DURF_REQUIRE(L >= 1 && L <= DURF_MAX_LEVELS && levels != nullptr, "1 <= L <= DURF_MAX_LEVELS level descriptions");
    if (B <= 0) return 0;
    LossLevelsK A{};
    for (int l = 0; l < L; l++) {
        const durf_loss_level& h = levels[l];
        LossLevelK& a = A.l[l];
        a.raw_bkgd = h.raw_bkgd; a.t_vals = h.t_vals; a.norm = h.norm;
        for (int k = 0; k < DURF_MAX_OBJ; k++) a.raw_obj.p[k] = (k < K) ? h.raw_obj[k] : nullptr;
        a.c.eps = eps; a.c.c_rgb = h.mults[0]; a.c.c_sky = h.mults[1]; a.c.c_depth = h.mults[2]; a.c.c_near = h.mults[3];
        a.c.c_empty = h.mults[4]; a.c.c_dist = h.mults[5]; a.c.box_loss_mult = box_loss_mult; a.c.level = h.level;
        a.c.bg = bg; a.c.density_bias = density_bias; a.c.disable_multiscale = disable_multiscale;
        a.draw = h.draw; a.terms = h.terms; a.rgb_out = h.rgb_out; a.depth_out = h.depth_out; a.acc_out = h.acc_out;
        a.weights_out = h.weights_out; a.t_mids_out = h.t_mids_out; a.t_dists_out = h.t_dists_out; a.draw_ray_sum = h.draw_ray_sum;
        DURF_REQUIRE(a.raw_bkgd && a.t_vals && a.norm && a.draw && a.terms, "raw, t_vals, norm, draw and terms of every level");
    }
    dim3 grid(durf_cdiv(B, 4), L), block(256);
#define LAUNCH_LL(P) hipLaunchKernelGGL(k_loss_bwd_levels<P>, grid, block, 0, (hipStream_t)stream, B, N, K, slot, dirs_s, pixels, \
                                        lossmult, gt_depth, sky, dyn, zo, A)
    if (N <= 64) LAUNCH_LL(1); else if (N <= 128) LAUNCH_LL(2); else LAUNCH_LL(4);
#undef LAUNCH_LL
    DURF_CHECK_LAUNCH("durf_loss_bwd_levels");
    return 0;
}

int durf_train_stats(void* stream, int L, int K, int N, const float* norms, float* sums,
                     const float* weight_l2, const float* pose6, const float* prev6, const float* target6,
                     const float* const* t_vals, const float* mults, int mode, float* out,
                     const float* const* terms, int B) {
    DURF_REQUIRE(L >= 1 && L <= DURF_MAX_LEVELS, "1 <= num_levels <= DURF_MAX_LEVELS");
    TvalPtrs tv, tm;
    for (int l = 0; l < DURF_MAX_LEVELS; l++) {
        tv.p[l] = l < L ? t_vals[l] : nullptr;
        tm.p[l] = (terms && l < L) ? terms[l] : nullptr;
    }
    const StatMults m = {mults[0], mults[1], mults[2], mults[3], mults[4], mults[5]};
    hipLaunchKernelGGL(k_train_stats, dim3(1), dim3(1024), 0, (hipStream_t)stream, L, K, N, norms, sums, weight_l2,
                       pose6, prev6, target6, tv, tm, B, m, mode, out);
    DURF_CHECK_LAUNCH("durf_train_stats");
    return 0;
}

int durf_stats_scrub(void* stream, int L, int K, int N, const float* norms, float* sums, const float* weight_l2,
                     const float* pose6, const float* prev6, const float* target6, const float* const* t_vals,
                     const float* mults, int mode, float* out, const float* const* terms, int B, size_t n, float* grad,
                     float inv_world, float max_val, float* scratch, const int32_t* cls_count, size_t box_floats, int K_boxes,
                     size_t mlp0_floats, size_t obj_floats) {
    DURF_REQUIRE(L >= 1 && L <= DURF_MAX_LEVELS, "1 <= num_levels <= DURF_MAX_LEVELS");
    DURF_REQUIRE(n > 0 && grad != nullptr && scratch != nullptr, "the flat gradient and the scrub partials");
    DURF_REQUIRE(cls_count == nullptr || (K_boxes >= 1 && K_boxes <= DURF_MAX_OBJ && obj_floats > 0 &&
                                          n == box_floats + mlp0_floats + (size_t)K_boxes * obj_floats),
                 "multi-hit outcome: flat layout box_centers | MLP_0 | K object MLPs, the WHOLE buffer");
    TvalPtrs tv, tm;
    for (int l = 0; l < DURF_MAX_LEVELS; l++) {
        tv.p[l] = l < L ? t_vals[l] : nullptr;
        tm.p[l] = (terms && l < L) ? terms[l] : nullptr;
    }
    const StatMults m = {mults[0], mults[1], mults[2], mults[3], mults[4], mults[5]};
    PoisonArgs pa{};
    pa.cls_count = cls_count; pa.box_floats = box_floats; pa.mlp0_floats = mlp0_floats; pa.obj_floats = obj_floats; pa.K = K_boxes;
    const size_t nvb = durf_cdiv(n, OPT_BLOCK * OPT_PER_THREAD);
    hipLaunchKernelGGL(k_stats_scrub, dim3(1 + durf_cdiv(nvb, 4)), dim3(1024), 0, (hipStream_t)stream, L, K, N, norms, sums,
                       weight_l2, pose6, prev6, target6, tv, tm, B, m, mode, out, n, grad, inv_world, max_val, scratch, pa);
    DURF_CHECK_LAUNCH("durf_stats_scrub");
    return 0;
}

}  // extern "C"
