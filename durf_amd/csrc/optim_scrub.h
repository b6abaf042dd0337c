// The scrub pass of the optimizer step (train_boxpose.py:253-277: pmean scale, nan_to_num, value clip, the partial sums of
// the global norm) as a device function over one VIRTUAL block of 256 threads x 8 elements, shared by k_grad_scrub
// (optim.hip: one virtual block per workgroup) and k_stats_scrub (loss.hip: four per 1024-thread workgroup, beside the
// workgroup that assembles the logged scalars) -- one body, so both leave the same bits in grad and in the partials.
#pragma once
#include "durf_common.h"

#define OPT_BLOCK 256
#define OPT_PER_THREAD 8

// the reference's outcome of rays that hit two boxes (durf_poison_multi_hit), folded into the scrub: a poisoned element
// is NaN in the reference's gradient and therefore 0 after nan_to_num (train_boxpose.py:263)
struct PoisonArgs {
    const int32_t* cls_count;       // durf_compact_classes' count[5], or null: nothing to poison
    size_t box_floats, mlp0_floats, obj_floats;
    int K;
};
__device__ __forceinline__ bool poisoned_index(const PoisonArgs& p, unsigned bits, size_t i) {
    if (i < p.box_floats) return (bits >> ((i / 6) % p.K)) & 1u;                           // box_centers [T, K, 6]
    if (i < p.box_floats + p.mlp0_floats) return true;                                     // MLP_0: every ray runs through it
    return (bits >> ((i - p.box_floats - p.mlp0_floats) / p.obj_floats)) & 1u;             // BoxMLP_k
}

// vb: virtual block index; tid in [0, 256); s_sq / s_mx: 4 floats of shared memory each, this virtual block's own.
// Every thread of the workgroup must call it (one __syncthreads inside).
__device__ __forceinline__ void scrub_vblock(size_t n, float* __restrict__ g, float inv_world, float max_val,
                                             float* __restrict__ part, size_t vb, int tid, float* s_sq, float* s_mx,
                                             const PoisonArgs& pa, bool live) {
    const size_t base = vb * OPT_BLOCK * OPT_PER_THREAD;
    const bool poison = pa.cls_count != nullptr && pa.cls_count[3] != 0;
    const unsigned bits = poison ? (unsigned)pa.cls_count[4] : 0u;
    float sq = 0.0f, mx = 0.0f;
#pragma unroll
    for (int i = 0; i < OPT_PER_THREAD; i++) {
        const size_t idx = base + (size_t)i * OPT_BLOCK + tid;
        if (live && idx < n) {
            float v = g[idx] * inv_world;                       // pmean over devices (:253)
            if (poison && poisoned_index(pa, bits, idx)) v = 0.0f;      // NaN in the reference -> 0 (:263)
            if (v != v || v == __builtin_inff()) v = 0.0f;      // nan_to_num(g, posinf=0.0) (:263)
            else if (v == -__builtin_inff()) v = -3.4028234663852886e+38f;
            if (max_val > 0.0f) v = fminf(fmaxf(v, -max_val), max_val);   // :275-277
            g[idx] = v;
            sq += v * v;
            mx = fmaxf(mx, fabsf(v));
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { sq += __shfl_xor(sq, o, 64); mx = fmaxf(mx, __shfl_xor(mx, o, 64)); }
    if ((tid & 63) == 0) { s_sq[tid >> 6] = sq; s_mx[tid >> 6] = mx; }
    __syncthreads();
    if (live && tid == 0) {
        float a = 0.0f, m = 0.0f;
        for (int w = 0; w < OPT_BLOCK / 64; w++) { a += s_sq[w]; m = fmaxf(m, s_mx[w]); }
        part[2 * vb] = a;
        part[2 * vb + 1] = m;
    }
}
