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

