// Per-ray masks and normaliser rows of loss_fn (train_boxpose.py:94-102,138-140,164), shared by the loss
// kernels (loss.hip) and the fused composite + resample + loss-prep kernel (render.hip).
#pragma once
#include "durf_common.h"

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }
// jax.nn.softplus = logaddexp(x, 0)
__device__ __forceinline__ float softplusf_(float x) { return fmaxf(x, 0.0f) + log1pf(expf(-fabsf(x))); }

struct LossCfg {
    float eps;              // near-loss half width (schedule value)
    float c_rgb, c_sky, c_depth, c_near, c_empty, c_dist;   // multipliers of this level's terms
    float box_loss_mult;
    int level;              // depth_mask accumulates box_loss_mult*dyn*box once per level (:140)
    float bg;               // background colour added with (1-acc): 0.5 / 1 / 0
    float density_bias;
    int disable_multiscale;
};

// per-ray masks (train_boxpose.py:94-102,138-140)
struct RayMasks { float m, dm, sm, box; };
__device__ __forceinline__ RayMasks ray_masks(const LossCfg& c, float lossmult, float gt, float sky,
                                              float dyn, float zo) {
    RayMasks r;
    r.m = c.disable_multiscale ? 1.0f : lossmult;
    const float dm0 = gt > 0.0f ? 1.0f : 0.0f;
    const float s0 = sky > 0.0f ? 1.0f : 0.0f;
    r.sm = s0 - dm0 * s0;
    r.box = gt < zo ? 1.0f : 0.0f;
    r.dm = dm0 + (float)(c.level + 1) * (c.box_loss_mult * dyn * r.box);
    return r;
}

// rows of the per-ray prep buffer
enum { PREP_M = 0, PREP_DM = 1, PREP_SM = 2, PREP_MIND2 = 3, PREP_DYN = 4, PREP_ROWS = 5 };

// squared distance of one interval start t to the LIDAR depth if it lies in the near band (:158-163); the
// per-ray minimum over the N interval starts, reduced over the batch, is the argument of distr.max() (:164)
__device__ __forceinline__ float near_d2(float t, float gt, float eps, float dm) {
    const float ind = (t > gt - eps && t < gt + eps) ? 1.0f : 0.0f;
    const float d = (ind * dm) * (t - gt);
    return d * d;
}
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ void write_prep(float* __restrict__ prep, int B, int b, const RayMasks& r, float mind2,
                                           float dynf) {
    prep[(size_t)PREP_M * B + b] = r.m;
    prep[(size_t)PREP_DM * B + b] = r.dm;
    prep[(size_t)PREP_SM * B + b] = r.sm;
    prep[(size_t)PREP_MIND2 * B + b] = mind2;
    prep[(size_t)PREP_DYN * B + b] = dynf;
}

// k_reduce_rows' row reduction (loss.hip: 1024 threads -- strided partial sums, xor-shuffle trees, the 16 waves' sums in
// order) carried out by a 256-thread workgroup with the SAME order of additions, i.e. the same bits: thread t plays the
// virtual threads t + 256 j (j = 0..3), whose wave is (t >> 6) + 4 j and whose lane is this thread's own.  s16: 16 floats of
// shared memory.  Every thread of the workgroup must call it (two __syncthreads per row).
__device__ __forceinline__ void reduce_rows_256(int rows, int n, int min_row, const float* __restrict__ in, float* __restrict__ out,
                                                float* s16) {
    const int t = threadIdx.x;
    for (int r = 0; r < rows; r++) {
        const bool is_min = (min_row >= 0 && r % PREP_ROWS == min_row);
        const float* p = in + (size_t)r * n;
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            v[j] = is_min ? __builtin_inff() : 0.0f;
            for (int i = t + 256 * j; i < n; i += 1024) v[j] = is_min ? fminf(v[j], p[i]) : v[j] + p[i];
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const float x = __shfl_xor(v[j], o, 64);
                v[j] = is_min ? fminf(v[j], x) : v[j] + x;
            }
        if ((t & 63) == 0)
#pragma unroll
            for (int j = 0; j < 4; j++) s16[(t >> 6) + 4 * j] = v[j];
        __syncthreads();
        if (t == 0) {
            float a = s16[0];
            for (int w = 1; w < 16; w++) a = is_min ? fminf(a, s16[w]) : a + s16[w];
            out[r] = a;
        }
        __syncthreads();
    }
}

namespace durf {
// out[r] = sum over in[r*n .. r*n+n), or the minimum for rows with r % PREP_ROWS == min_row (min_row < 0: none)
void launch_reduce_rows(hipStream_t s, int rows, int n, int min_row, const float* in, float* out);
}
