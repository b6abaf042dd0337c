// Ray generation + 'timestep' batch gather on the device (obbpose_dataset.py:1868-1916,1551-1583) and
// the SSIM evaluation metric (internal/math.py:66-137): the callers / data formats either side of the
// hot path.  HBM-bound gathers and small stencils; fp32 like the reference.
#include "durf_common.h"

struct CamTable { float v[DURF_MAX_CAMS][17]; int first[DURF_MAX_CAMS + 1]; int n; };

// one thread per ray: flat index into the timestep's concatenated cameras -> (camera, x, y)
__global__ void __launch_bounds__(256)
k_gen_batch(int B, CamTable cams, const int32_t* __restrict__ ray_idx, float near, float far,
            const float* __restrict__ images, const float* __restrict__ depth, const float* __restrict__ sky,
            int img_channels, float* __restrict__ origins, float* __restrict__ dirs, float* __restrict__ viewdirs,
            float* __restrict__ radii, float* __restrict__ lossmult, float* __restrict__ near_o,
            float* __restrict__ far_o, float* __restrict__ pixels, float* __restrict__ depth_o,
            float* __restrict__ sky_o) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B) return;
    const int r = ray_idx ? ray_idx[i] : i;
    int c = 0;
    for (int k = 1; k < cams.n; k++) c += (r >= cams.first[k]) ? 1 : 0;
    const float* cam = cams.v[c];
    const int w = (int)cam[16], h = (int)cam[15];
    const int p = r - cams.first[c];
    const int y = p / w, x = p - y * w;
    auto dir = [&](int yy, float* d) {           // :1882-1889: d = sum_j cam_dirs_j * R[:, j], in that order
        const float cd[3] = {((float)x - cam[13]) / cam[12], -((float)yy - cam[14]) / cam[12], -1.0f};
#pragma unroll
        for (int a = 0; a < 3; a++) d[a] = (cd[0] * cam[4 * a] + cd[1] * cam[4 * a + 1]) + cd[2] * cam[4 * a + 2];
    };
    float d[3], dn[3];
    dir(y, d);
    // radius: distance to the next row's direction; the last row repeats the previous one (:1896-1902)
    const int y0 = (y < h - 1) ? y : h - 2;
    float d0[3];
    dir(y0, d0);
    dir(y0 + 1, dn);
    const float dx = sqrtf(((d0[0] - dn[0]) * (d0[0] - dn[0]) + (d0[1] - dn[1]) * (d0[1] - dn[1])) +
                           (d0[2] - dn[2]) * (d0[2] - dn[2]));
    const float nrm = sqrtf((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]);
#pragma unroll
    for (int a = 0; a < 3; a++) {
        origins[i * 3 + a] = cam[4 * a + 3];
        dirs[i * 3 + a] = d[a];
        viewdirs[i * 3 + a] = d[a] / nrm;
    }
    radii[i] = dx * 2.0f / 3.4641016151377544f;       // 2 / sqrt(12)
    lossmult[i] = 1.0f;
    near_o[i] = near;
    far_o[i] = far;
    if (pixels) for (int a = 0; a < img_channels; a++) pixels[(size_t)i * img_channels + a] = images[(size_t)r * img_channels + a];
    if (depth_o) depth_o[i] = depth[r];
    if (sky_o) sky_o[i] = sky[r];
}

// SSIM map: one thread per output pixel-channel; 'valid' separable Gaussian window (x first, then y)
__global__ void __launch_bounds__(256)
k_ssim(int H, int W, int C, int fs, const float* __restrict__ filt, const float* __restrict__ a,
       const float* __restrict__ b, float c1, float c2, float* __restrict__ ssim_map, float* __restrict__ block_sums) {
    __shared__ float red[256];
    const int Ho = H - fs + 1, Wo = W - fs + 1;
    const int tot = Ho * Wo * C;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    float val = 0.0f;
    if (i < tot) {
        const int c = i % C, xo = (i / C) % Wo, yo = i / (C * Wo);
        float m0 = 0.f, m1 = 0.f, s00 = 0.f, s11 = 0.f, s01 = 0.f;
        for (int dy = 0; dy < fs; dy++) {
            float r0 = 0.f, r1 = 0.f, r00 = 0.f, r11 = 0.f, r01 = 0.f;
            const size_t row = ((size_t)(yo + dy) * W + xo) * C + c;
            for (int dxx = 0; dxx < fs; dxx++) {
                const float f = filt[fs - 1 - dxx], p = a[row + (size_t)dxx * C], q = b[row + (size_t)dxx * C];
                r0 += f * p; r1 += f * q; r00 += f * (p * p); r11 += f * (q * q); r01 += f * (p * q);
            }
            const float g = filt[fs - 1 - dy];
            m0 += g * r0; m1 += g * r1; s00 += g * r00; s11 += g * r11; s01 += g * r01;
        }
        const float mu00 = m0 * m0, mu11 = m1 * m1, mu01 = m0 * m1;
        const float v00 = fmaxf(0.0f, s00 - mu00), v11 = fmaxf(0.0f, s11 - mu11);
        float v01 = s01 - mu01;
        const float lim = sqrtf(v00 * v11);
        const float av = fminf(lim, fabsf(v01));
        v01 = v01 > 0.0f ? av : (v01 < 0.0f ? -av : 0.0f);
        val = ((2.0f * mu01 + c1) * (2.0f * v01 + c2)) / ((mu00 + mu11 + c1) * (v00 + v11 + c2));
        if (ssim_map) ssim_map[i] = val;
    }
    red[threadIdx.x] = val;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) block_sums[blockIdx.x] = red[0];
}

__global__ void k_ssim_mean(int nblocks, int count, const float* __restrict__ block_sums, float* __restrict__ out) {
    __shared__ float red[256];
    float s = 0.0f;
    for (int i = threadIdx.x; i < nblocks; i += 256) s += block_sums[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x == 0) *out = red[0] / (float)count;
}

extern "C" {

int durf_gen_batch(void* stream, int B, int n_cams, const float* cams_host, const int32_t* ray_idx, float near,
                   float far, const float* images, const float* depth, const float* sky, int img_channels,
                   float* origins, float* directions, float* viewdirs, float* radii, float* lossmult,
                   float* near_out, float* far_out, float* pixels, float* depth_out, float* sky_out) {
    DURF_REQUIRE(n_cams >= 1 && n_cams <= DURF_MAX_CAMS, "1 <= n_cams <= DURF_MAX_CAMS");
    if (B <= 0) return 0;
    CamTable t;
    t.n = n_cams;
    int first = 0;
    for (int c = 0; c < n_cams; c++) {
        for (int j = 0; j < 17; j++) t.v[c][j] = cams_host[c * 17 + j];
        DURF_REQUIRE(t.v[c][15] >= 2 && t.v[c][16] >= 1, "camera height >= 2, width >= 1");
        t.first[c] = first;
        first += (int)t.v[c][15] * (int)t.v[c][16];
    }
    t.first[n_cams] = first;
    hipLaunchKernelGGL(k_gen_batch, dim3(durf_cdiv(B, 256)), dim3(256), 0, (hipStream_t)stream, B, t, ray_idx, near, far,
                       images, depth, sky, img_channels, origins, directions, viewdirs, radii, lossmult, near_out,
                       far_out, images ? pixels : nullptr, depth ? depth_out : nullptr, sky ? sky_out : nullptr);
    DURF_CHECK_LAUNCH("durf_gen_batch");
    return 0;
}

size_t durf_ssim_scratch_floats(int H, int W, int C, int filter_size) {
    const long n = (long)(H - filter_size + 1) * (W - filter_size + 1) * C;
    return n > 0 ? (size_t)durf_cdiv((size_t)n, 256) : 0;
}

int durf_ssim(void* stream, int H, int W, int C, const float* img0, const float* img1, float max_val,
              int filter_size, const float* filt_dev, float k1, float k2, float* ssim_map, float* scratch,
              float* ssim_mean) {
    DURF_REQUIRE(filter_size >= 1 && H >= filter_size && W >= filter_size && C >= 1, "image smaller than the window");
    const int n = (H - filter_size + 1) * (W - filter_size + 1) * C;
    const int nb = (int)durf_cdiv((size_t)n, 256);
    const float c1 = (k1 * max_val) * (k1 * max_val), c2 = (k2 * max_val) * (k2 * max_val);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_ssim, dim3(nb), dim3(256), 0, s, H, W, C, filter_size, filt_dev, img0, img1, c1, c2, ssim_map,
                       scratch);
    hipLaunchKernelGGL(k_ssim_mean, dim3(1), dim3(256), 0, s, nb, n, scratch, ssim_mean);
    DURF_CHECK_LAUNCH("durf_ssim");
    return 0;
}

}  // extern "C"
