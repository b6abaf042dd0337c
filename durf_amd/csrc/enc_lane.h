// The bf16 fast-path encoder of ONE sample per lane (K3/K4: integrated positional encoding, mip.py:226-282, and its
// BARF-weighted object form, mip.py:182-223), shared by the stand-alone encode kernels (rays.hip: k_encode_lane) and by
// the fused background forward that encodes its own tiles (mlp_fwd.hip: k_mlp_fwd<256, .., ENC>) -- one body, so the two
// produce bit-identical features.
#pragma once
#include <type_traits>
#include "gauss.h"

template <int I, int N, class F>
__device__ __forceinline__ void enc_static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        enc_static_for<I + 1, N>(f);
    }
}

// mip360.new_space (mip360.py:47-79): x -> contract(x); var_j -> (var_j * v_j) * v_j with
// v = JVP of contract along (1,1,1), written in the op order reverse-free JVP produces.
__device__ __forceinline__ void contract_gaussian(Gauss& g) {
    const float s0 = g.x[0] * g.x[0] + g.x[1] * g.x[1] + g.x[2] * g.x[2];
    const bool tiny = s0 < 1e-12f;
    const float n = sqrtf(tiny ? 1e-12f : s0);
    if (n <= 0.1f) return;                                     // x_smaller branch: x' = x, v = 1
    const float dn = tiny ? 0.0f : (2.0f * (g.x[0] + g.x[1] + g.x[2])) / (2.0f * n);
    const float inv = 1.0f / n;
    const float fac = 2.0f - inv;
#pragma unroll
    for (int j = 0; j < 3; j++) {
        const float xn = g.x[j] / n;
        // d(2 - 1/n) = dn/n^2 ; d(x/n) = 1/n - x*dn/n^2
        const float v = (dn / (n * n)) * xn + fac * (inv - g.x[j] * dn / (n * n));
        g.x[j] = fac * xn;
        g.var[j] = (g.var[j] * v) * v;
    }
}

// Same arithmetic as ipe_feature<true>; the safe_sin wrap is an exact fmod done with one fma
// (y - floor(y/t)*t is representable, so the single rounding of the fma returns it exactly).
__device__ __forceinline__ float wrap_100pi(float y) {
    const float t = 314.15927124023438f;
    if (!(fabsf(y) < t)) {
        const float q = floorf(y * (1.0f / 314.15927124023438f));
        float m = fmaf(-q, t, y);
        if (m < 0.0f) m += t;
        if (m >= t) m -= t;
        y = m;
    }
    return y;
}

// The Gaussian of sample n of ray b as the BACKGROUND encoder sees it: cast_rays (mip.py:155-179), the object mask
// (obbpose_model.py:205-210: multiplied by 1 - number of boxes hit), mip360.new_space.
__device__ __forceinline__ Gauss bkgd_sample_gaussian(int b, int n, int N, const float* __restrict__ t_vals,
                                                      const float* __restrict__ origins_s, const float* __restrict__ dirs_s,
                                                      const float* __restrict__ radii, const int32_t* __restrict__ hit, int K,
                                                      int flags) {
    const float t0 = t_vals[(size_t)b * (N + 1) + n], t1 = t_vals[(size_t)b * (N + 1) + n + 1];
    float o[3] = {origins_s[b * 3], origins_s[b * 3 + 1], origins_s[b * 3 + 2]};
    float d[3] = {dirs_s[b * 3], dirs_s[b * 3 + 1], dirs_s[b * 3 + 2]};
    Gauss g = frustum_gaussian(t0, t1, o, d, radii[b], (flags & DURF_ENC_CYLINDER) != 0);
    if (flags & DURF_ENC_NO_INTEGRATION) g.var[0] = g.var[1] = g.var[2] = 0.0f;      // obbpose_model.py:164-165
    int nh = 0;
    for (int k = 0; k < K; k++) nh += hit[b * K + k];
    if (nh != 0) {
        const float m = 1.0f - (float)nh;
#pragma unroll
        for (int i = 0; i < 3; i++) { g.x[i] *= m; g.var[i] *= m; }
    }
    if (flags & DURF_ENC_CONTRACT) contract_gaussian(g);
    return g;
}

// The 64 feature slots of one sample (OBJ: 3 coordinates + 60 weighted IPE features + pad; else 60 IPE features + pad),
// handed to `sink(integral_constant<q>, bf16x8)` as the eight 16-byte vectors [8 q, 8 q + 8) in the order they complete.
// QMASK: the vectors q this call produces (bit q).  Every feature is a function of the sample alone, so a caller may deal the
// eight vectors to several waves (the M-split object forward: four waves x two vectors); what a call does not hand out is
// dead code, the values it does hand out are the full call's, bit for bit.
template <bool OBJ, unsigned QMASK = 0xFFu, class Sink>
__device__ __forceinline__ void lane_features(const Gauss& g, const BarfW& barf_w, Sink&& sink) {
    float feat[64];
#pragma unroll
    for (int i = 0; i < 64; i++) feat[i] = 0.0f;
    constexpr int OFF = OBJ ? 3 : 0;
    if (OBJ) { feat[0] = g.x[0]; feat[1] = g.x[1]; feat[2] = g.x[2]; }
    // Degrees are the OUTER loop and every 16-byte output vector is stored as soon as its last feature exists
    // (all indices are compile-time constants): the stores are spread through the arithmetic instead of bursting
    // at the end of the wave, and at most ~3 vectors of features are live at a time, so the kernel fits 64 VGPRs
    // (8 waves per SIMD: the 8192 waves of a 4096-ray launch are all resident at once).
    auto flush = [&](int deg) {
        enc_static_for<0, 8>([&](auto q_) {
            constexpr int q = decltype(q_)::value;
            // last degree that contributes to features [8q, 8q+8): sin features OFF + 3 deg + a, cos OFF + 30 + 3 deg + a
            int last = 0;
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const int f = q * 8 + e - OFF;
                if (f >= 0 && f < 60) { const int d = (f % 30) / 3; last = d > last ? d : last; }
            }
            if (last == deg && ((QMASK >> q) & 1u)) {
                bf16x8 o8;
#pragma unroll
                for (int e = 0; e < 8; e++) o8[e] = (__bf16)feat[q * 8 + e];
                sink(q_, o8);
            }
        });
    };
    // Range reduction of the 60 sine arguments.  safe_sin wraps |y| >= 100 pi by an exact fmod (wrap_100pi below the
    // sine).  v_sin_f32 works in revolutions and any integer may be dropped, so v_fract(y / 2 pi) does the same job in
    // one op when the arguments are moderate.  Differences from the exact wrap: it ignores that the reference's 100 pi
    // is rounded to fp32 (5.9e-6 rad per wrap) and it rounds y / 2 pi in fp32 (<= 2e-4 rad at |y| = 2048); both are far
    // below the bf16 quantum 4e-3 this path writes (measured: max / mean abs error vs the oracle unchanged to three
    // digits, tests/encode_error.py).  A sample with max|x| * 512 + pi/2 >= 2048 takes the exact wrap, unchanged (a
    // wave with both kinds runs both paths under their lane masks) -- that covers uncontracted coordinates (|y| up to 1e5, where the fp32 rounding of y / 2 pi
    // would reach the quantum), object-frame rays, and also contracted points: the reference's contraction switches
    // at norm 0.1, so norms just above 0.1 map to |2 - 1/n| up to 8, not <= 2.  How many samples that is depends on
    // the scene (level 0 of the bench.py batch: none).
    const float amax = fmaxf(fmaxf(fabsf(g.x[0]), fabsf(g.x[1])), fabsf(g.x[2]));
    const bool big = !(amax * 512.0f + 1.5707963705062866f < 2048.0f);         // also true for NaN
    if (big) {        // per lane: a sample's features never depend on which samples share its wave
#pragma unroll
        for (int deg = 0; deg < 10; deg++) {
            const float sc = (float)(1 << deg);
#pragma unroll
            for (int a = 0; a < 3; a++) {
                const float y = g.x[a] * sc;
                const float yc = y + 1.5707963705062866f;
                const float yv = g.var[a] * sc * sc;
                const float e = __expf(-0.5f * yv);
                float fs = e * __sinf(wrap_100pi(y));
                float fc = e * __sinf(wrap_100pi(yc));
                const int f = deg * 3 + a;
                if (OBJ) { fs = barf_w.w[f / 6] * fs; fc = barf_w.w[(f + 30) / 6] * fc; }   // mip.py:217-222
                feat[OFF + f] = fs;
                feat[OFF + 30 + f] = fc;
            }
            flush(deg);
        }
    } else {
        // Octave recurrences (the arithmetic is transcendental-heavy: 60 v_sin + 30 v_exp at quarter rate): per axis
        // the sine / cosine of degree 0 and of degree 5 come from the hardware (v_sin / v_cos of v_fract(y / 2 pi)),
        // the degrees in between from the double-angle formulas s' = 2 s c, c' = 1 - 2 s^2, so an angle or amplitude
        // error at most doubles per octave (<= 16 x the hardware error ~1e-6, far below the bf16 quantum 4e-3 this
        // path writes).  cos(y) stands in for the reference's sin(y + fl(pi/2)) (they differ by the fp32 rounding of
        // the sum, <= 3e-5 rad at |y| = 1000).  The Gaussian damping exp(-0.5 var 4^deg) is evaluated at degrees
        // 0, 3, 6, 9 and raised to the 4th power in between (the relative error quadruples per octave: two chained
        // steps stay below 2e-6).  12 + 12 transcendentals instead of 90.  The exact fp32 path (k_encode, libm) is
        // untouched; error vs the oracle: tests/encode_error.py.
        const float inv2pi = 0.15915494309189535f;
        float sn[3], cs[3], ev[3];
#pragma unroll
        for (int deg = 0; deg < 10; deg++) {
            const float sc = (float)(1 << deg);
#pragma unroll
            for (int a = 0; a < 3; a++) {
                if (deg % 5 == 0) {
                    const float r = __builtin_amdgcn_fractf((g.x[a] * sc) * inv2pi);
                    sn[a] = __builtin_amdgcn_sinf(r);
                    cs[a] = __builtin_amdgcn_cosf(r);
                } else {
                    const float s2 = sn[a] + sn[a];
                    const float cn = fmaf(-s2, sn[a], 1.0f);
                    sn[a] = s2 * cs[a];
                    cs[a] = cn;
                }
                if (deg % 3 == 0) {
                    ev[a] = __expf(-0.5f * (g.var[a] * sc * sc));
                } else {
                    const float e2 = ev[a] * ev[a];
                    ev[a] = e2 * e2;
                }
                float fs = ev[a] * sn[a], fc = ev[a] * cs[a];
                const int f = deg * 3 + a;
                if (OBJ) { fs = barf_w.w[f / 6] * fs; fc = barf_w.w[(f + 30) / 6] * fc; }   // mip.py:217-222
                feat[OFF + f] = fs;
                feat[OFF + 30 + f] = fc;
            }
            flush(deg);
        }
    }
}
