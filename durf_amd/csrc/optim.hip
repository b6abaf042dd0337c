// Gradient post-processing + Adam (K12): train_boxpose.py:257-289 and flax.optim.Adam
// (apply_param_gradient: bias-corrected moments, eps outside the sqrt, no weight decay).
// Two passes over the flat fp32 buffers (0.6-2 M floats: latency-, not bandwidth-bound):
//   1. scrub (nan/+inf -> 0, -inf -> -FLT_MAX), value clip, per-block sum of squares / abs max
//   2. Adam update with the scaled gradient; every workgroup first derives the global norm, abs max and clip multiplier
//      from the per-block partials (a separate single-workgroup launch in round 1)
#include "optim_scrub.h"

__global__ void __launch_bounds__(OPT_BLOCK)
k_grad_scrub(size_t n, float* __restrict__ g, float inv_world, float max_val, float* __restrict__ part) {
    __shared__ float s_sq[OPT_BLOCK / 64], s_mx[OPT_BLOCK / 64];
    scrub_vblock(n, g, inv_world, max_val, part, blockIdx.x, threadIdx.x, s_sq, s_mx, PoisonArgs{}, true);
}

// The global norm / clip multiplier from the per-block partials.  Every workgroup of the Adam launch recomputes it (537
// pairs at 1.1 M parameters: a few L2 reads per thread) instead of a separate single-workgroup launch in between; the
// order of additions is fixed, so every workgroup gets the same bits.
// out[0] = grad_norm, out[1] = grad_abs_max, out[2] = clip multiplier, out[3] = grad_norm_clipped
__device__ __forceinline__ float grad_clip_mult(int nblocks, const float* __restrict__ part, float max_norm,
                                                float* __restrict__ out, bool write) {
    __shared__ float s_sq[4], s_mx[4], s_mult;
    float sq = 0.0f, mx = 0.0f;
    for (int i = threadIdx.x; i < nblocks; i += 256) { sq += part[2 * i]; mx = fmaxf(mx, part[2 * i + 1]); }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { sq += __shfl_xor(sq, o, 64); mx = fmaxf(mx, __shfl_xor(mx, o, 64)); }
    if ((threadIdx.x & 63) == 0) { s_sq[threadIdx.x >> 6] = sq; s_mx[threadIdx.x >> 6] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float a = 0.0f, m = 0.0f;
        for (int w = 0; w < 4; w++) { a += s_sq[w]; m = fmaxf(m, s_mx[w]); }
        const float norm = sqrtf(a);
        float mult = 1.0f;
        if (max_norm > 0.0f) mult = fminf(1.0f, max_norm / (1e-7f + norm));       // :283-285
        s_mult = mult;
        if (write) { out[0] = norm; out[1] = m; out[2] = mult; out[3] = norm * mult; }
    }
    __syncthreads();
    return s_mult;
}

__global__ void __launch_bounds__(256)
k_adam(size_t n, float* __restrict__ p, float* __restrict__ m, float* __restrict__ v,
       const float* __restrict__ g, int nblocks, const float* __restrict__ part, float max_norm,
       float* __restrict__ stats, float lr, float beta1, float beta2, float eps, float bc1, float bc2) {
    const float mult = grad_clip_mult(nblocks, part, max_norm, stats, blockIdx.x == 0);
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float gi = mult * g[i];
    const float mi = beta1 * m[i] + (1.0f - beta1) * gi;
    const float vi = beta2 * v[i] + (1.0f - beta2) * (gi * gi);
    m[i] = mi;
    v[i] = vi;
    const float mh = mi / bc1, vh = vi / bc2;              // bc = 1 - beta^t
    p[i] = p[i] - lr * mh / (sqrtf(vh) + eps);
}

// A ray that hits two boxes is garbage in the reference (obbpose_model.py:120-122): its colours are NaN, the loss is NaN,
// and d(loss)/d(theta) is NaN wherever its samples reach -- all of the background MLP, and in the object MLPs every weight
// column whose unit is active on one of them -- which jnp.nan_to_num then turns into a zero gradient
// (train_boxpose.py:263).  This path never evaluates such a ray (it is on no box's compacted list), so the outcome is
// stated per segment of the flat gradient: MLP_0, and the MLP and pose columns of every box such a ray hits, are set to
// NaN -- before the data-parallel all-reduce, where the reference's pmean sees its NaNs -- and durf_clip_adam's scrub does
// the rest.  A handful of workgroups that exit at once when no such ray exists.
__global__ void __launch_bounds__(256)
k_poison_multi_hit(size_t n, float* __restrict__ g, const int32_t* __restrict__ cls_count, size_t box_floats, int K,
                   size_t mlp0_floats, size_t obj_floats) {
    if (cls_count[3] == 0) return;
    const unsigned bits = (unsigned)cls_count[4];
    const float qn = __builtin_nanf("");
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        bool p;
        if (i < box_floats) p = (bits >> ((i / 6) % K)) & 1u;                           // box_centers [T, K, 6]
        else if (i < box_floats + mlp0_floats) p = true;                                // MLP_0: every ray runs through it
        else p = (bits >> ((i - box_floats - mlp0_floats) / obj_floats)) & 1u;          // BoxMLP_k
        if (p) g[i] = qn;
    }
}

// Config.weight_decay_mult (train_boxpose.py:73-75): loss += mult * mean(theta^2) over EVERY parameter, so
// d(loss)/d(theta) += (2 mult / n) theta.  One pass: the gradient term on [lo, hi) (a bucketed exchange hands the object
// MLPs' slice over before the rest exists) and, when asked for, per-block sums of squares over all n; a single-workgroup
// launch adds those in index order (deterministic) into the scalar durf_train_stats logs and adds to the loss.
__global__ void __launch_bounds__(OPT_BLOCK)
k_weight_decay(size_t n, const float* __restrict__ p, float* __restrict__ g, size_t lo, size_t hi, float c,
               float* __restrict__ part) {
    __shared__ float s_sq[OPT_BLOCK / 64];
    const size_t base = (size_t)blockIdx.x * (OPT_BLOCK * OPT_PER_THREAD);
    float sq = 0.0f;
#pragma unroll
    for (int j = 0; j < OPT_PER_THREAD; j++) {
        const size_t i = base + (size_t)j * OPT_BLOCK + threadIdx.x;
        if (i < n) {
            const float t = p[i];
            sq = __fadd_rn(sq, __fmul_rn(t, t));
            if (i >= lo && i < hi) g[i] = __fadd_rn(g[i], __fmul_rn(c, t));
        }
    }
    if (part == nullptr) return;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o, 64);
    if ((threadIdx.x & 63) == 0) s_sq[threadIdx.x >> 6] = sq;
    __syncthreads();
    if (threadIdx.x == 0) {
        float a = 0.0f;
        for (int w = 0; w < OPT_BLOCK / 64; w++) a += s_sq[w];
        part[blockIdx.x] = a;
    }
}

__global__ void __launch_bounds__(256)
k_weight_l2(int nblocks, const float* __restrict__ part, float mult, float inv_n, float* __restrict__ out) {
    __shared__ float s_sq[4];
    float sq = 0.0f;
    for (int i = threadIdx.x; i < nblocks; i += 256) sq += part[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o, 64);
    if ((threadIdx.x & 63) == 0) s_sq[threadIdx.x >> 6] = sq;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = mult * (((s_sq[0] + s_sq[1]) + (s_sq[2] + s_sq[3])) * inv_n);
}

extern "C" {

int durf_poison_multi_hit(void* stream, size_t n, float* grad, const int32_t* cls_count, size_t box_floats, int K,
                          size_t mlp0_floats, size_t obj_floats) {
    DURF_REQUIRE(K >= 1 && K <= DURF_MAX_OBJ && obj_floats > 0 && n <= box_floats + mlp0_floats + (size_t)K * obj_floats,
                 "flat layout: box_centers | MLP_0 | K object MLPs (n: the whole buffer or a prefix of it)");
    hipLaunchKernelGGL(k_poison_multi_hit, dim3(32), dim3(256), 0, (hipStream_t)stream, n, grad, cls_count, box_floats, K,
                       mlp0_floats, obj_floats);
    DURF_CHECK_LAUNCH("durf_poison_multi_hit");
    return 0;
}

// Config.weight_decay_mult: grad[lo:hi) += (2 mult / n) params[lo:hi); weight_l2 (nullable) <- mult * mean(params^2) over
// all n.  scratch: durf_optim_scratch_floats(n) floats (only read between this call's two launches).
int durf_weight_decay(void* stream, size_t n, const float* params, float* grad, size_t lo, size_t hi, float mult,
                      float* scratch, float* weight_l2) {
    if (n == 0) return 0;
    DURF_REQUIRE(params != nullptr && grad != nullptr && lo <= hi && hi <= n, "params, grad, 0 <= lo <= hi <= n");
    DURF_REQUIRE(weight_l2 == nullptr || scratch != nullptr, "weight_l2 needs the scratch of durf_optim_scratch_floats(n)");
    hipStream_t s = (hipStream_t)stream;
    const int nb = (int)durf_cdiv(n, OPT_BLOCK * OPT_PER_THREAD);
    const float c = (float)(2.0 * (double)mult / (double)n);
    hipLaunchKernelGGL(k_weight_decay, dim3(nb), dim3(OPT_BLOCK), 0, s, n, params, grad, lo, hi, c,
                       weight_l2 != nullptr ? scratch : nullptr);
    if (weight_l2 != nullptr)
        hipLaunchKernelGGL(k_weight_l2, dim3(1), dim3(256), 0, s, nb, scratch, mult, (float)(1.0 / (double)n), weight_l2);
    DURF_CHECK_LAUNCH("durf_weight_decay");
    return 0;
}

size_t durf_optim_scratch_floats(size_t n) {
    return 2 * (size_t)durf_cdiv(n, OPT_BLOCK * OPT_PER_THREAD) + 4;
}

// The Adam half of durf_clip_adam alone, behind a scrub that ran elsewhere (durf_stats_scrub): `scratch` holds that pass's
// partials of the same n.
int durf_adam_apply(void* stream, size_t n, float* params, float* m, float* v, const float* grad, float max_norm, float lr,
                    int step, const float* scratch, float* stats) {
    if (n == 0) return 0;
    const int nb = (int)durf_cdiv(n, OPT_BLOCK * OPT_PER_THREAD);
    const double b1 = 0.9, b2 = 0.999;
    const double t = (double)step + 1.0;
    hipLaunchKernelGGL(k_adam, dim3(durf_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, n, params, m, v, grad, nb, scratch,
                       max_norm, stats, lr, (float)b1, (float)b2, 1e-8f, (float)(1.0 - pow(b1, t)), (float)(1.0 - pow(b2, t)));
    DURF_CHECK_LAUNCH("durf_adam_apply");
    return 0;
}

// grad is modified in place (mean over world, scrub, value clip); stats[4] receives
// grad_norm, grad_abs_max, clip multiplier, grad_norm_clipped; step is 0-based (t = step+1).
int durf_clip_adam(void* stream, size_t n, float* params, float* m, float* v, float* grad, float inv_world,
                   float max_val, float max_norm, float lr, int step, float* scratch, float* stats) {
    if (n == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    const int nb = (int)durf_cdiv(n, OPT_BLOCK * OPT_PER_THREAD);
    hipLaunchKernelGGL(k_grad_scrub, dim3(nb), dim3(OPT_BLOCK), 0, s, n, grad, inv_world, max_val, scratch);
    const double b1 = 0.9, b2 = 0.999;
    const double t = (double)step + 1.0;
    hipLaunchKernelGGL(k_adam, dim3(durf_cdiv(n, 256)), dim3(256), 0, s, n, params, m, v, grad, nb, scratch, max_norm,
                       stats, lr, (float)b1, (float)b2, 1e-8f, (float)(1.0 - pow(b1, t)), (float)(1.0 - pow(b2, t)));
    DURF_CHECK_LAUNCH("durf_clip_adam");
    return 0;
}

}  // extern "C"
