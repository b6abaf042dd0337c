// Conical-frustum Gaussian shared by the encoders (rays.hip) and their backward (pose.hip).
#pragma once
#include "durf_common.h"

struct Gauss { float x[3]; float var[3]; };
struct BarfW { float w[10]; };

// mip.cast_rays / conical_frustum_to_gaussian / lift_gaussian (mip.py:155-179,99-130,76-96);
// only diag(cov) is ever consumed downstream.
__device__ __forceinline__ Gauss frustum_gaussian(float t0, float t1, const float* o,
                                                  const float* d, float radius, bool cylinder = false) {
    if (cylinder) {                                   // mip.cylinder_to_gaussian (mip.py:133-152)
        const float t_mean = (t0 + t1) / 2.0f;
        const float r_var = (radius * radius) / 4.0f;
        const float t_var = ((t1 - t0) * (t1 - t0)) / 12.0f;
        const float dmag = fmaxf(1e-10f, d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
        Gauss g;
#pragma unroll
        for (int j = 0; j < 3; j++) {
            g.x[j] = d[j] * t_mean + o[j];
            g.var[j] = t_var * (d[j] * d[j]) + r_var * (1.0f - d[j] * (d[j] / dmag));
        }
        return g;
    }
    const float mu = (t0 + t1) / 2.0f;
    const float hw = (t1 - t0) / 2.0f;
    const float mu2 = mu * mu, hw2 = hw * hw;
    const float den = 3.0f * mu2 + hw2;
    const float hw4 = hw2 * hw2;
    const float t_mean = mu + (2.0f * mu * hw2) / den;
    const float t_var = hw2 / 3.0f - (4.0f / 15.0f) * ((hw4 * (12.0f * mu2 - hw2)) / (den * den));
    const float r_var = (radius * radius) * (mu2 / 4.0f + (5.0f / 12.0f) * hw2 - (4.0f / 15.0f) * hw4 / den);
    const float dmag = fmaxf(1e-10f, d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    Gauss g;
#pragma unroll
    for (int j = 0; j < 3; j++) {
        g.x[j] = d[j] * t_mean + o[j];
        const float dd = d[j] * d[j];
        const float null_d = 1.0f - d[j] * (d[j] / dmag);
        g.var[j] = t_var * dd + r_var * null_d;
    }
    return g;
}


// one IPE feature f in [0,60): mip.py:273-282 with basis [2^i I3] (accurate libm path, used when
// the caller asks for fp32 features; the bf16-only path is k_encode_lane below).
__device__ __forceinline__ float ipe_feature(const Gauss& g, int f) {
    const int c = f / 30, r = f - c * 30, deg = r / 3, j = r - deg * 3;
    const float sc = (float)(1 << deg);
    float y = g.x[j] * sc;
    if (c) y = y + 1.5707963705062866f;
    const float yv = g.var[j] * sc * sc;
    return expf(-0.5f * yv) * safe_sin(y);
}

// The 8 consecutive features [8 q, 8 q + 8) of one sample's OBJECT encoding (63 features padded to 64): the sample's
// coordinates, then the BARF-weighted IPE (mip.weighted_ipe, mip.py:182-223; weight index = feature / 6).  Shared by
// k_encode<true> (rays.hip) and the fp32 forward that encodes its own tiles (mlp_f32.hip): bit-identical features.
__device__ __forceinline__ void obj_features8(const Gauss& g, const BarfW& w, int q, float (&v)[8]) {
#pragma unroll
    for (int e = 0; e < 8; e++) {
        const int p = q * 8 + e;
        float val;
        if (p < 3) val = g.x[p];
        else if (p < 63) { const int f = p - 3; val = w.w[f / 6] * ipe_feature(g, f); }
        else val = 0.0f;
        v[e] = val;
    }
}
