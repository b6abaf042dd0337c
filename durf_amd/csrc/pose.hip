// Box-pose gradients (part of K11): reverse-mode of the object branch of MipNerfModel.__call__
// from d(loss)/d(object encoding) back to box_centers[ts, k, :6]:
//   weighted_ipe (mip.py:182-223)  ->  cast_rays / conical frustum (mip.py:155-179,99-130,76-96)
//   ->  world2object_rpy (box_helpers.py:286-341)  ->  aa2matrix (box_helpers.py:148-167).
// `hit` is stop-gradient (obbpose_model.py:113) and level-1 t_vals are stop-gradient
// (mip.py:413-414), so the only path is (o', d') -> sample Gaussians -> encoding.
//
// k_encode_obj_bwd: one wavefront per hit ray; per sample 60 sin/cos/exp (VALU-bound, but only
// hit rays are processed); reduces over the ray's samples to d(o'), d(d'), then to the 21
// per-ray sums that determine dL/dR and dL/dc.  Reduction over rays is a fixed-order row sum.
#include "gauss.h"

enum { POSE_ROWS = 21 };   // [0..2] sum g_o ; [3..11] sum g_o (x) o ; [12..20] sum g_u (x) d

// One WORKGROUP per hit ray and object (blockIdx.x = compact ray, blockIdx.y = object of a batched call): the four waves
// split the 60 encoding features (15 each: the per-sample chain of 60 sin / cos / exp is what a launch waits for -- with
// one wave per ray it was 50 us per object and level, K host-driven launches per level), lanes take the samples, and the
// per-wave partial d(o'), d(d') meet in LDS in a fixed order.
// PRECISE: libm exp / sin / cos (the fp32 object branch and the parity instrument); otherwise the hardware
// transcendentals (bf16 object branch: |z| < 314.16 after the wrap, where v_sin's z / 2 pi scaling alone loses
// ~|z| 2^-24 = 2e-5 rad -- two orders below the bf16 rounding of the d(enc) it multiplies).
// the per-level operands of one launch over several levels (blockIdx.z = level: durf_encode_obj_bwd_levels)
struct EncBwdLevels {
    const float* d_enc[DURF_MAX_LEVELS];
    const float* t_vals[DURF_MAX_LEVELS];
    float* rows_out[DURF_MAX_LEVELS];      // [21][B] per object, column j = compact ray index
};

template <int P, bool PRECISE>
__global__ void __launch_bounds__(256)
k_encode_obj_bwd(int B, int N, int k_obj, const int32_t* __restrict__ idx, const int32_t* __restrict__ count,
                 EncBwdLevels lv,
                 const float* __restrict__ origins_s, const float* __restrict__ dirs_s,
                 const float* __restrict__ radii, const float* __restrict__ origins,
                 const float* __restrict__ dirs, const float* __restrict__ pose, BarfW bw,
                 size_t idx_stride, size_t denc_stride, size_t rows_stride, int enc_flags) {
    const float* __restrict__ d_enc = lv.d_enc[blockIdx.z];
    const float* __restrict__ t_vals = lv.t_vals[blockIdx.z];
    float* __restrict__ rows_out = lv.rows_out[blockIdx.z];
    // MipNerfModel.ray_shape = 'cylinder' (mip.cylinder_to_gaussian, mip.py:133-152) and disable_integration
    // (obbpose_model.py:163-164: the variances are zeroed, so only the means carry a gradient)
    const bool cyl = (enc_flags & DURF_ENC_CYLINDER) != 0, noint = (enc_flags & DURF_ENC_NO_INTEGRATION) != 0;
    __shared__ float part[4][6];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int kb = blockIdx.y;                      // object within this call
    k_obj += kb;
    idx += kb * idx_stride; count += kb; d_enc += kb * denc_stride; rows_out += kb * rows_stride;
    // capped grid (the hit count lives on the device; thousands of workgroups that only exit cost their dispatch)
    const int nj = *count < B ? *count : B;
    for (int j = blockIdx.x; j < nj; j += gridDim.x) {
    const int b = idx[j];
    const float o[3] = {origins_s[b * 3], origins_s[b * 3 + 1], origins_s[b * 3 + 2]};
    const float d[3] = {dirs_s[b * 3], dirs_s[b * 3 + 1], dirs_s[b * 3 + 2]};
    const float radius = radii[b];
    const float dsum = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
    const float m = fmaxf(1e-10f, dsum);
    float go[3] = {0.f, 0.f, 0.f}, gd[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int p = 0; p < P; p++) {
        const int n = lane * P + p;
        if (n >= N) continue;
        const float t0 = t_vals[(size_t)b * (N + 1) + n], t1 = t_vals[(size_t)b * (N + 1) + n + 1];
        // forward quantities (same formulas as frustum_gaussian)
        const float mu = (t0 + t1) / 2.0f, hw = (t1 - t0) / 2.0f;
        const float mu2 = mu * mu, hw2 = hw * hw, den = 3.0f * mu2 + hw2, hw4 = hw2 * hw2;
        float t_mean = mu + (2.0f * mu * hw2) / den;
        float t_var = hw2 / 3.0f - (4.0f / 15.0f) * ((hw4 * (12.0f * mu2 - hw2)) / (den * den));
        float r_var = (radius * radius) * (mu2 / 4.0f + (5.0f / 12.0f) * hw2 - (4.0f / 15.0f) * hw4 / den);
        if (cyl) { t_mean = mu; r_var = (radius * radius) / 4.0f; t_var = ((t1 - t0) * (t1 - t0)) / 12.0f; }
        float x[3], var[3];
#pragma unroll
        for (int i = 0; i < 3; i++) {
            x[i] = d[i] * t_mean + o[i];
            var[i] = noint ? 0.0f : t_var * (d[i] * d[i]) + r_var * (1.0f - d[i] * (d[i] / m));
        }
        const float* ge = d_enc + ((size_t)j * N + n) * DURF_ENC_DIM;
        float gx[3] = {0.f, 0.f, 0.f}, gv[3] = {0.f, 0.f, 0.f};
        if (wv == 0) { gx[0] = ge[0]; gx[1] = ge[1]; gx[2] = ge[2]; }     // identity features (mip.py:222)
        for (int f = 15 * wv; f < 15 * wv + 15; f++) {
            const int c = f / 30, r = f - c * 30, deg = r / 3, i = r - deg * 3;
            const float sc = (float)(1 << deg);
            float z = x[i] * sc;
            if (c) z = z + 1.5707963705062866f;
            const float t = 314.15927124023438f;                          // safe_sin wrap (math.py:35-46)
            if (!(fabsf(z) < t)) { float q = fmodf(z, t); if (q != 0.0f && q < 0.0f) q += t; z = q; }
            const float e = PRECISE ? expf(-0.5f * (var[i] * sc * sc)) : __expf(-0.5f * (var[i] * sc * sc));
            const float cz = PRECISE ? cosf(z) : __cosf(z), sz = PRECISE ? sinf(z) : __sinf(z);
            const float g = ge[3 + f] * bw.w[f / 6];
            const float gxf = g * e * sc * cz, gvf = noint ? 0.0f : g * (-0.5f * sc * sc) * e * sz;
            if (i == 0) { gx[0] += gxf; gv[0] += gvf; }
            else if (i == 1) { gx[1] += gxf; gv[1] += gvf; }
            else { gx[2] += gxf; gv[2] += gvf; }
        }
        // x_i = o_i + d_i t_mean ; var_i = t_var d_i^2 + r_var (1 - d_i^2 / m)     (linear in gx, gv: per-wave partials add up)
        float s_gv = 0.0f;
#pragma unroll
        for (int i = 0; i < 3; i++) s_gv += gv[i] * r_var * (d[i] * d[i]) / (m * m);
#pragma unroll
        for (int i = 0; i < 3; i++) {
            go[i] += gx[i];
            float g = gx[i] * t_mean + gv[i] * (2.0f * t_var * d[i]) - gv[i] * r_var * (2.0f * d[i] / m);
            if (dsum > 1e-10f) g += s_gv * 2.0f * d[i];                   // through m = max(1e-10, |d|^2)
            gd[i] += g;
        }
    }
#pragma unroll
    for (int i = 0; i < 3; i++) { go[i] = wave_sum(go[i]); gd[i] = wave_sum(gd[i]); }
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < 3; i++) { part[wv][i] = go[i]; part[wv][3 + i] = gd[i]; }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
    for (int i = 0; i < 3; i++) {
        go[i] = ((part[0][i] + part[1][i]) + part[2][i]) + part[3][i];
        gd[i] = ((part[0][3 + i] + part[1][3 + i]) + part[2][3 + i]) + part[3][3 + i];
    }
    // o' = R (o_w - c), u = R d_w, d' = u / |u|  (box_helpers.py:323-340)
    const float* pk = pose + k_obj * 6;
    const float rx = pk[3], ry = pk[4], rz = pk[5];
    float s = rx * rx + ry * ry + rz * rz;
    s = (s < 1e-12f) ? 1e-12f : s;
    const float th = sqrtf(s) + 1e-12f;
    const float a = sinf(th) / th, bb = (1.0f - cosf(th)) / (th * th);
    const float S[9] = {0.f, -rz, ry, rz, 0.f, -rx, -ry, rx, 0.f};
    float R[9];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int q = 0; q < 3; q++) {
            const float s2 = S[i * 3] * S[q] + S[i * 3 + 1] * S[3 + q] + S[i * 3 + 2] * S[6 + q];
            R[i * 3 + q] = ((i == q) ? 1.0f : 0.0f) + a * S[i * 3 + q] + bb * s2;
        }
    const float ow[3] = {origins[b * 3], origins[b * 3 + 1], origins[b * 3 + 2]};
    const float dw[3] = {dirs[b * 3], dirs[b * 3 + 1], dirs[b * 3 + 2]};
    float u[3];
#pragma unroll
    for (int i = 0; i < 3; i++) u[i] = R[i * 3] * dw[0] + R[i * 3 + 1] * dw[1] + R[i * 3 + 2] * dw[2];
    const float nrm = sqrtf(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
    const float dp[3] = {u[0] / nrm, u[1] / nrm, u[2] / nrm};
    const float dot = dp[0] * gd[0] + dp[1] * gd[1] + dp[2] * gd[2];
    float gu[3];
#pragma unroll
    for (int i = 0; i < 3; i++) gu[i] = (gd[i] - dp[i] * dot) / nrm;
#pragma unroll
    for (int i = 0; i < 3; i++) {
        rows_out[(size_t)i * B + j] = go[i];
#pragma unroll
        for (int q = 0; q < 3; q++) {
            rows_out[(size_t)(3 + i * 3 + q) * B + j] = go[i] * ow[q];
            rows_out[(size_t)(12 + i * 3 + q) * B + j] = gu[i] * dw[q];
        }
    }
    }   // thread 0
    __syncthreads();                                // part[] is reused by the next ray
    }   // rays
}

// (nlev levels in the order given: each level's row sum is formed and added exactly as a launch of its own would -- the same bits
// as one launch per level)
__global__ void __launch_bounds__(1024)
k_pose_reduce(int n, const int32_t* __restrict__ count, EncBwdLevels lv, int nlev, float* __restrict__ out) {
    __shared__ float sh[16];
    const int r = blockIdx.x;
    count += blockIdx.y;                               // batched call: blockIdx.y = object
    out += blockIdx.y * POSE_ROWS;
    const int c = *count < n ? *count : n;
    for (int l = 0; l < nlev; l++) {
        const float* p = lv.rows_out[l] + (size_t)blockIdx.y * POSE_ROWS * n + (size_t)r * n;
        float v = 0.0f;
        for (int i = threadIdx.x; i < c; i += 1024) v += p[i];
        v = wave_sum(v);
        __syncthreads();                   // (sh[] of the previous level has been read)
        if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
        __syncthreads();
        if (threadIdx.x == 0) {
            float a = sh[0];
            for (int w = 1; w < 16; w++) a += sh[w];
            out[r] += a;                   // accumulates over levels (caller zeroes)
        }
    }
}

// sums[K][21] -> d(loss)/d(box_centers[ts, k, 0:6]) (added into grad6[K][6])
__global__ void k_pose_finish(int K, const float* __restrict__ pose, const float* __restrict__ sums,
                              int want_pos, int want_rot, float* __restrict__ grad6) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    const float* pk = pose + k * 6;
    const float* sm = sums + k * POSE_ROWS;
    const float c[3] = {pk[0], pk[1], pk[2]};
    const float r[3] = {pk[3], pk[4], pk[5]};
    const float s0 = r[0] * r[0] + r[1] * r[1] + r[2] * r[2];
    const bool tiny = s0 < 1e-12f;
    const float s = tiny ? 1e-12f : s0;
    const float rt = sqrtf(s);
    const float th = rt + 1e-12f;
    const float sn = sinf(th), cs = cosf(th);
    const float a = sn / th, b = (1.0f - cs) / (th * th);
    const float da = (th * cs - sn) / (th * th);
    const float db = (th * sn - 2.0f * (1.0f - cs)) / (th * th * th);
    float Km[9] = {0.f, -r[2], r[1], r[2], 0.f, -r[0], -r[1], r[0], 0.f};
    float K2[9], R[9], G[9];
    for (int i = 0; i < 3; i++)
        for (int q = 0; q < 3; q++) {
            K2[i * 3 + q] = Km[i * 3] * Km[q] + Km[i * 3 + 1] * Km[3 + q] + Km[i * 3 + 2] * Km[6 + q];
            R[i * 3 + q] = ((i == q) ? 1.0f : 0.0f) + a * Km[i * 3 + q] + b * K2[i * 3 + q];
            // dL/dR = sum g_o (x) (o - c) + sum g_u (x) d
            G[i * 3 + q] = sm[3 + i * 3 + q] - sm[i] * c[q] + sm[12 + i * 3 + q];
        }
    if (want_pos)
        for (int q = 0; q < 3; q++)            // dL/dc = -R^T sum g_o
            grad6[k * 6 + q] += -(R[q] * sm[0] + R[3 + q] * sm[1] + R[6 + q] * sm[2]);
    if (want_rot)
        for (int i = 0; i < 3; i++) {
            float E[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};    // skew(e_i)
            if (i == 0) { E[5] = -1.f; E[7] = 1.f; }
            if (i == 1) { E[2] = 1.f; E[6] = -1.f; }
            if (i == 2) { E[1] = -1.f; E[3] = 1.f; }
            const float dth = tiny ? 0.0f : r[i] / rt;
            float acc = 0.0f;
            for (int p = 0; p < 3; p++)
                for (int q = 0; q < 3; q++) {
                    const float ek = E[p * 3] * Km[q] + E[p * 3 + 1] * Km[3 + q] + E[p * 3 + 2] * Km[6 + q];
                    const float ke = Km[p * 3] * E[q] + Km[p * 3 + 1] * E[3 + q] + Km[p * 3 + 2] * E[6 + q];
                    const float dR = da * dth * Km[p * 3 + q] + a * E[p * 3 + q] + db * dth * K2[p * 3 + q] + b * (ek + ke);
                    acc += G[p * 3 + q] * dR;
                }
            grad6[k * 6 + 3 + i] += acc;
        }
}

extern "C" {

// One level: d_enc [K][count*N, 64] -> accumulates sums[(k0 + k)*21 .. +21) for the K objects of the call (caller
// zeroes sums once per step).  scratch: K*21*B floats.  idx [K,B] / count [K] / d_enc slabs denc_stride floats apart.
static int encode_obj_bwd_launch(void* stream, int K, int B, int N, int k0, const int32_t* idx, const int32_t* count,
                                 const float* d_enc, size_t denc_stride, const float* t_vals, const float* origins_s,
                                 const float* dirs_s, const float* radii, const float* origins, const float* dirs,
                                 const float* pose, const float* barf_w, float* scratch, float* sums, int precise, int enc_flags,
                                 int nlev = 1, const float* const* d_enc_l = nullptr, const float* const* t_vals_l = nullptr) {
    DURF_REQUIRE(N >= 1 && N <= 256, "1 <= N <= 256");
    DURF_REQUIRE(nlev >= 1 && nlev <= DURF_MAX_LEVELS, "1 <= nlevels <= DURF_MAX_LEVELS");
    DURF_REQUIRE((enc_flags & ~(DURF_ENC_CYLINDER | DURF_ENC_NO_INTEGRATION)) == 0, "enc_flags: DURF_ENC_CYLINDER | DURF_ENC_NO_INTEGRATION");
    if (B <= 0 || K <= 0) return 0;
    BarfW bw;
    for (int i = 0; i < 10; i++) bw.w[i] = barf_w[i];
    hipStream_t s = (hipStream_t)stream;
    // (levels: scratch holds nlev blocks of K * 21 * B floats)
    EncBwdLevels lv{};
    for (int l = 0; l < nlev; l++) {
        lv.d_enc[l] = d_enc_l ? d_enc_l[l] : d_enc;
        lv.t_vals[l] = t_vals_l ? t_vals_l[l] : t_vals;
        lv.rows_out[l] = scratch + (size_t)l * K * POSE_ROWS * B;
    }
    dim3 grid(B < 256 ? B : 256, K, nlev), block(256);
#define LAUNCH_E2(P, PR)                                                                                  \
    hipLaunchKernelGGL((k_encode_obj_bwd<P, PR>), grid, block, 0, s, B, N, k0, idx, count, lv,            \
                       origins_s, dirs_s, radii, origins, dirs, pose, bw, (size_t)B, denc_stride,          \
                       (size_t)POSE_ROWS * B, enc_flags)
#define LAUNCH_E(P) { if (precise) LAUNCH_E2(P, true); else LAUNCH_E2(P, false); }
    if (N <= 64) LAUNCH_E(1) else if (N <= 128) LAUNCH_E(2) else LAUNCH_E(4)
#undef LAUNCH_E2
#undef LAUNCH_E
    hipLaunchKernelGGL(k_pose_reduce, dim3(POSE_ROWS, K), dim3(1024), 0, s, B, count, lv, nlev, sums + k0 * POSE_ROWS);
    DURF_CHECK_LAUNCH("durf_encode_obj_bwd");
    return 0;
}

int durf_encode_obj_bwd(void* stream, int B, int N, int k_obj, const int32_t* idx, const int32_t* count,
                        const float* d_enc, const float* t_vals, const float* origins_s,
                        const float* dirs_s, const float* radii, const float* origins, const float* dirs,
                        const float* pose, const float* barf_w, float* scratch, float* sums, int precise, int enc_flags) {
    return encode_obj_bwd_launch(stream, 1, B, N, k_obj, idx, count, d_enc, 0, t_vals, origins_s, dirs_s, radii, origins,
                                 dirs, pose, barf_w, scratch, sums, precise, enc_flags);
}

int durf_encode_obj_bwd_batch(void* stream, int K, int B, int N, const int32_t* idx, const int32_t* count,
                              const float* d_enc, const float* t_vals, const float* origins_s,
                              const float* dirs_s, const float* radii, const float* origins, const float* dirs,
                              const float* pose, const float* barf_w, float* scratch, float* sums, int precise, int enc_flags) {
    return encode_obj_bwd_launch(stream, K, B, N, 0, idx, count, d_enc, (size_t)B * N * DURF_ENC_DIM, t_vals, origins_s,
                                 dirs_s, radii, origins, dirs, pose, barf_w, scratch, sums, precise, enc_flags);
}

int durf_encode_obj_bwd_levels(void* stream, int K, int B, int N, int nlevels, const int32_t* idx, const int32_t* count,
                               const float* const* d_enc, const float* const* t_vals, const float* origins_s,
                               const float* dirs_s, const float* radii, const float* origins, const float* dirs,
                               const float* pose, const float* barf_w, float* scratch, float* sums, int precise, int enc_flags) {
    return encode_obj_bwd_launch(stream, K, B, N, 0, idx, count, nullptr, (size_t)B * N * DURF_ENC_DIM, nullptr, origins_s,
                                 dirs_s, radii, origins, dirs, pose, barf_w, scratch, sums, precise, enc_flags, nlevels, d_enc, t_vals);
}

// sums [K,21] (all levels accumulated) -> adds d(loss)/d(box_centers[ts]) into grad6 [K,6]
int durf_pose_finish(void* stream, int K, const float* pose, const float* sums, int want_pos, int want_rot,
                     float* grad6) {
    if (K <= 0) return 0;
    hipLaunchKernelGGL(k_pose_finish, dim3(1), dim3(64), 0, (hipStream_t)stream, K, pose, sums, want_pos, want_rot,
                       grad6);
    DURF_CHECK_LAUNCH("durf_pose_finish");
    return 0;
}

}  // extern "C"
