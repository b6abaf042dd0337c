// Exact-fp32 MLP: the PARITY INSTRUMENT of the fused bf16 kernels (SURVEY.md 7.4 / 8c "F32_EXACT").
//
// The reference evaluates its Dense layers in fp32 (obbpose_model.py:326-327, HIGHEST-precision matmul,
// internal/math.py:22-24).  These kernels evaluate the same stack with v_mfma_f32_32x32x2_f32 (exact fp32
// products and accumulation: bitwise an fmaf chain over k), reading the fp32 flax-layout parameters directly
// (no packing) and exchanging row-major fp32 tensors.  They are 1/16 of the bf16 MFMA rate by construction and
// are not tuned: one wave owns 32 samples, activations pass between layers through LDS as [feature][sample].
// Used by MipNerfModel(mlp_precision='f32') for end-to-end fp32 parity tests; never by bench.py.
//
// Per-sample record of the forward ("act", ACT floats): the INPUT of every Dense, concatenations included,
//   x0 = enc | x1..x4 = h0..h3 | x5 = [h4, enc] | x6, x7 = h5, h6 | x8 = h7 (density head and bottleneck) |
//   x10 = [bottleneck, view] | x11 = hc
// so that every weight gradient is one GEMM  dW_l = x_l^T dz_l  over contiguous columns (the bias is the row of
// ones appended to x_l).  Per-sample record of the backward ("dz", DZ floats): d(loss)/d(pre-activation) of
// every Dense output, in Dense order.
#include "mlp_spec.h"

struct F32Layer { int fi, fo, x_off, dz_off, relu; size_t w_off; };
struct F32Spec {
    F32Layer L[12];
    int act, dz, W, in_dim;
};

__host__ __device__ inline F32Spec f32_spec(int W, int in_dim) {
    F32Spec s;
    s.W = W; s.in_dim = in_dim;
    int x = 0, d = 0;
    for (int l = 0; l < 12; l++) {
        int fi, fo;
        durf_layer_shape(W, in_dim, l, &fi, &fo);
        s.L[l].fi = fi; s.L[l].fo = fo;
        s.L[l].w_off = durf_layer_offset(W, in_dim, l, 0);
        s.L[l].relu = (l <= 7 || l == 10) ? 1 : 0;
        s.L[l].dz_off = d; d += fo;
        if (l == 9) s.L[l].x_off = s.L[8].x_off;           // bottleneck reads h7 like the density head
        else { s.L[l].x_off = x; x += fi; }
    }
    s.act = x; s.dz = d;
    return s;
}

#define F32_MAXF 320                 // >= widest Dense input (316) ; rows of one LDS activation buffer
#define F32_BUF (F32_MAXF * 32)      // floats per buffer: [feature][sample]

__device__ __forceinline__ int c_row(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

// acc[mo] (+)= W^T x for one Dense: out tiles mo < nmt, input rows k < fi from LDS x[k][n]
__device__ __forceinline__ void dense_fwd(const float* __restrict__ Wl, int fi, int fo, const float* x, int lane,
                                          f32x16 (&acc)[8]) {
    const int m = lane & 31, kk = lane >> 5;
    const int nmt = (fo + 31) >> 5;
    const float* bias = Wl + (size_t)fi * fo;
#pragma unroll
    for (int mo = 0; mo < 8; mo++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int o = 32 * mo + c_row(r, kk);
            acc[mo][r] = (mo < nmt && o < fo) ? bias[o] : 0.0f;
        }
    for (int ks = 0; ks < (fi + 1) / 2; ks++) {
        const int k = 2 * ks + kk;
        const float b = k < fi ? x[k * 32 + m] : 0.0f;
#pragma unroll
        for (int mo = 0; mo < 8; mo++) {
            if (mo < nmt) {
                const int o = 32 * mo + m;
                const float a = (k < fi && o < fo) ? Wl[(size_t)k * fo + o] : 0.0f;
                acc[mo] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[mo], 0, 0, 0);
            }
        }
    }
}

// d_x[ki] (+)= W dz for one Dense: input-feature tiles ki < nkt, dz rows m < fo from LDS dz[m][n]
__device__ __forceinline__ void dense_bwd(const float* __restrict__ Wl, int fi, int fo, const float* dzl, int lane,
                                          f32x16 (&acc)[10], bool accumulate) {
    const int i = lane & 31, kk = lane >> 5;
    const int nkt = (fi + 31) >> 5;
    if (!accumulate) {
#pragma unroll
        for (int ki = 0; ki < 10; ki++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[ki][r] = 0.0f;
    }
    for (int ms = 0; ms < (fo + 1) / 2; ms++) {
        const int mm = 2 * ms + kk;
        const float b = mm < fo ? dzl[mm * 32 + i] : 0.0f;
#pragma unroll
        for (int ki = 0; ki < 10; ki++) {
            if (ki < nkt) {
                const int k = 32 * ki + i;
                const float a = (k < fi && mm < fo) ? Wl[(size_t)k * fo + mm] : 0.0f;
                acc[ki] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[ki], 0, 0, 0);
            }
        }
    }
}

__device__ __forceinline__ size_t f32_rows(size_t rows, int N, const int32_t* count) {
    if (!count) return rows;
    const size_t c = (size_t)(*count) * (size_t)N;
    return c < rows ? c : rows;
}

// ---------------------------------------------------------------------------------------------
// forward: obbpose_model.py:305-354 / :369-418 in fp32
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64)
k_mlp_fwd_f32(F32Spec S, size_t rows, int N, const float* __restrict__ enc, const float* __restrict__ view,
              const int32_t* __restrict__ ray_idx, const int32_t* __restrict__ count,
              const float* __restrict__ P, float* __restrict__ raw, float* __restrict__ act) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* bufA = lds;
    float* bufB = lds + F32_BUF;
    const size_t nrows = f32_rows(rows, N, count);
    const size_t row0 = (size_t)blockIdx.x * 32;
    if (row0 >= nrows) return;
    const int lane = threadIdx.x, n = lane & 31, hi = lane >> 5;
    const size_t row = row0 + n;
    const bool valid = row < nrows;
    const int W = S.W, in_dim = S.in_dim;
    float* arow = act ? act + row * (size_t)S.act : nullptr;

    // x0 = enc
    for (int f = hi; f < in_dim; f += 2) {
        const float v = valid ? enc[row * (size_t)in_dim + f] : 0.0f;
        bufA[f * 32 + n] = v;
        if (arow && valid) arow[S.L[0].x_off + f] = v;
    }
    __syncthreads();
    f32x16 acc[8];
    float* cur = bufA;
    float* nxt = bufB;
    float dens = 0.0f;
    for (int l = 0; l < 12; l++) {
        const F32Layer& Ly = S.L[l];
        dense_fwd(P + Ly.w_off, Ly.fi, Ly.fo, cur, lane, acc);
        if (l == 8) {                       // density head: keeps the input buffer for the bottleneck
            dens = acc[0][0];               // out feature 0 lives in reg 0 of the hi = 0 half
            dens = __shfl(dens, n, 64);
            continue;
        }
        if (l == 11) break;
        // the next Dense's input: this output (ReLU) [+ skip / view concatenation]
        const int nmt = (Ly.fo + 31) >> 5;
        const int lx = l + 1 == 8 ? 8 : (l == 9 ? 10 : l + 1);       // which x record receives it (h7 -> x8, bott -> x10)
        const int xo = S.L[lx].x_off;
#pragma unroll
        for (int mo = 0; mo < 8; mo++) {
            if (mo < nmt) {
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int o = 32 * mo + c_row(r, hi);
                    float v = acc[mo][r];
                    // jnp.maximum(x, 0) propagates NaN; fmaxf would drop it
                    if (Ly.relu) v = (v != v) ? v : fmaxf(v, 0.0f);
                    if (o < Ly.fo) {
                        nxt[o * 32 + n] = v;
                        if (arow && valid) arow[xo + o] = v;
                    }
                }
            }
        }
        if (l == 4) {                        // x5 = [h4, enc]   (obbpose_model.py:333-334)
            for (int f = hi; f < in_dim; f += 2) {
                const float v = valid ? enc[row * (size_t)in_dim + f] : 0.0f;
                nxt[(W + f) * 32 + n] = v;
                if (arow && valid) arow[xo + W + f] = v;
            }
        }
        if (l == 9) {                        // x10 = [bottleneck, view]   (:346-347)
            size_t ray = row / (size_t)N;
            if (ray_idx && valid) ray = (size_t)ray_idx[ray];
            for (int f = hi; f < 27; f += 2) {
                const float v = valid ? view[ray * 27 + f] : 0.0f;
                nxt[(W + f) * 32 + n] = v;
                if (arow && valid) arow[xo + W + f] = v;
            }
        }
        __syncthreads();
        float* t = cur; cur = nxt; nxt = t;
    }
    // rgb head output: features 0..2 are regs 0..2 of the hi = 0 half
    if (valid && hi == 0) {
        const f32x4 o = {acc[0][0], acc[0][1], acc[0][2], dens};
        *(f32x4*)(raw + row * 4) = o;
    }
}

// ---------------------------------------------------------------------------------------------
// backward (data path): d(loss)/d(pre-activation) of every Dense, optionally d(loss)/d(enc)
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64)
k_mlp_bwd_f32(F32Spec S, size_t rows, int N, const float* __restrict__ draw, const int32_t* __restrict__ ray_idx,
              const int32_t* __restrict__ count, const float* __restrict__ P, const float* __restrict__ act,
              float* __restrict__ dz, float* __restrict__ d_enc) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* bufA = lds;                 // dz of the layer being propagated: [out feature][sample]
    float* bufB = lds + F32_BUF;
    const size_t nrows = f32_rows(rows, N, count);
    const size_t row0 = (size_t)blockIdx.x * 32;
    if (row0 >= nrows) return;
    const int lane = threadIdx.x, n = lane & 31, hi = lane >> 5;
    const size_t row = row0 + n;
    const bool valid = row < nrows;
    const int W = S.W, in_dim = S.in_dim;
    const float* arow = act + row * (size_t)S.act;
    float* zrow = dz + row * (size_t)S.dz;
    // head gradients (object MLPs gather their rows of the [B*N,4] buffer through ray_idx)
    size_t src = row;
    if (ray_idx && valid) src = (size_t)ray_idx[row / (size_t)N] * (size_t)N + row % (size_t)N;
    f32x4 g = {0.f, 0.f, 0.f, 0.f};
    if (valid) g = *(const f32x4*)(draw + src * 4);
    if (hi == 0) {
        bufA[0 * 32 + n] = g[0]; bufA[1 * 32 + n] = g[1]; bufA[2 * 32 + n] = g[2];      // dz11 (rgb head, linear)
        if (valid) {
            zrow[S.L[11].dz_off + 0] = g[0]; zrow[S.L[11].dz_off + 1] = g[1]; zrow[S.L[11].dz_off + 2] = g[2];
            zrow[S.L[8].dz_off] = g[3];                                                   // dz8 (density head, linear)
        }
    }
    __syncthreads();
    f32x16 acc[10];
    float* cur = bufA;
    float* nxt = bufB;
    // order of propagation: 11 -> 10 -> 9 (+ 8) -> 7 -> 6 -> 5 -> 4 ... -> 0
    const int order[11] = {11, 10, 9, 7, 6, 5, 4, 3, 2, 1, 0};
    for (int oi = 0; oi < 11; oi++) {
        const int l = order[oi];
        const F32Layer& Ly = S.L[l];
        dense_bwd(P + Ly.w_off, Ly.fi, Ly.fo, cur, lane, acc, false);
        if (l == 9) {                    // h7 feeds the bottleneck AND the density head (fo = 1): add W8 * dz8
            const float* W8 = P + S.L[8].w_off;
            const float g3 = __shfl(g[3], n, 64);        // this sample's d sigma (held by the hi = 0 half too)
#pragma unroll
            for (int ki = 0; ki < 10; ki++)
                if (ki < (W >> 5))
#pragma unroll
                    for (int r = 0; r < 16; r++) acc[ki][r] += W8[32 * ki + c_row(r, hi)] * g3;
        }
        if (l == 0) {
            if (d_enc && valid) {
#pragma unroll
                for (int ki = 0; ki < 2; ki++)
#pragma unroll
                    for (int r = 0; r < 16; r++) {
                        const int k = 32 * ki + c_row(r, hi);
                        if (k < in_dim) d_enc[row * 64 + k] += acc[ki][r];
                    }
            }
            break;
        }
        // d x_l -> dz of the Dense that produced x_l's first W_prev features (masked by its ReLU)
        const int lp = l == 11 ? 10 : (l == 10 ? 9 : (l == 9 ? 7 : l - 1));     // producer of x_l's leading features
        const F32Layer& Lp = S.L[lp];
        const int nkt = (Lp.fo + 31) >> 5;
#pragma unroll
        for (int ki = 0; ki < 10; ki++) {
            if (ki < nkt) {
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int k = 32 * ki + c_row(r, hi);
                    if (k < Lp.fo) {
                        float v = acc[ki][r];
                        if (Lp.relu) {
                            const float h = valid ? arow[Ly.x_off + k] : 0.0f;       // the producer's ReLU output
                            v = h > 0.0f ? v : 0.0f;
                        }
                        nxt[k * 32 + n] = v;
                        if (valid) zrow[Lp.dz_off + k] = v;
                    }
                }
            }
        }
        if (l == 5 && d_enc && valid) {                 // skip connection: rows W.. of d x5 are d enc
#pragma unroll
            for (int ki = 0; ki < 10; ki++)
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int k = 32 * ki + c_row(r, hi) - W;
                    if (ki >= (W >> 5) && k >= 0 && k < in_dim) d_enc[row * 64 + k] = acc[ki][r];
                }
        }
        __syncthreads();
        float* t = cur; cur = nxt; nxt = t;
    }
}

// ---------------------------------------------------------------------------------------------
// weight gradients: dW_l[k, m] = sum_n x_l[n, k] dz_l[n, m]  (row fi of x_l = 1: the bias), split over
// the samples, partials summed in a fixed order by k_dw_f32_reduce (deterministic, no atomics)
// ---------------------------------------------------------------------------------------------
struct F32Tile { int layer, ki, mj; };

__global__ void __launch_bounds__(64)
k_mlp_dw_f32(F32Spec S, size_t rows, int N, const int32_t* __restrict__ count, const float* __restrict__ act,
             const float* __restrict__ dz, const int* __restrict__ tiles, int nsplit, size_t params,
             float* __restrict__ part) {
    const size_t nrows = f32_rows(rows, N, count);
    const int t = blockIdx.x, sp = blockIdx.y;
    const int l = tiles[3 * t], ki = tiles[3 * t + 1], mj = tiles[3 * t + 2];
    const F32Layer& Ly = S.L[l];
    const int lane = threadIdx.x, i = lane & 31, kk = lane >> 5;
    const size_t per = ((nrows + nsplit - 1) / nsplit + 1) & ~(size_t)1;
    const size_t n0 = (size_t)sp * per, n1 = n0 + per < nrows ? n0 + per : nrows;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0.0f;
    const int k = 32 * ki + i, m = 32 * mj + i;
    for (size_t nn = n0; nn < n1; nn += 2) {
        const size_t s = nn + kk;
        float a = 0.0f, b = 0.0f;
        if (s < n1) {
            a = k < Ly.fi ? act[s * (size_t)S.act + Ly.x_off + k] : (k == Ly.fi ? 1.0f : 0.0f);
            b = m < Ly.fo ? dz[s * (size_t)S.dz + Ly.dz_off + m] : 0.0f;
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    float* p = part + (size_t)sp * params + Ly.w_off;
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const int kr = 32 * ki + c_row(r, kk);
        if (kr <= Ly.fi && m < Ly.fo) p[(size_t)kr * Ly.fo + m] = acc[r];
    }
}

__global__ void __launch_bounds__(256)
k_dw_f32_reduce(size_t params, int nsplit, const float* __restrict__ part, float* __restrict__ grad) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= params) return;
    float s = 0.0f;
    for (int sp = 0; sp < nsplit; sp++) s += part[(size_t)sp * params + i];
    grad[i] = s;
}

extern "C" {

size_t durf_mlp_f32_act_floats(int width, int in_dim) { return (size_t)f32_spec(width, in_dim).act; }
size_t durf_mlp_f32_dz_floats(int width, int in_dim) { return (size_t)f32_spec(width, in_dim).dz; }
size_t durf_mlp_f32_dw_scratch_floats(int width, int in_dim, int nsplit) {
    return (size_t)nsplit * durf_layer_offset(width, in_dim, 12, 0);
}

int durf_mlp_fwd_f32(void* stream, int width, int in_dim, size_t rows, int N, const float* enc, const float* view27,
                     const int32_t* ray_idx, const int32_t* count, const float* mlp_params, float* raw, float* act) {
    DURF_REQUIRE(width == 256 || width == 128, "width must be 256 or 128");
    DURF_REQUIRE(in_dim > 0 && in_dim <= DURF_ENC_DIM, "in_dim <= 64");
    if (rows == 0) return 0;
    const F32Spec S = f32_spec(width, in_dim);
    const int lds = 2 * F32_BUF * (int)sizeof(float);
    (void)hipFuncSetAttribute((const void*)k_mlp_fwd_f32, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL(k_mlp_fwd_f32, dim3(durf_cdiv(rows, 32)), dim3(64), lds, (hipStream_t)stream, S, rows, N, enc,
                       view27, ray_idx, count, mlp_params, raw, act);
    DURF_CHECK_LAUNCH("durf_mlp_fwd_f32");
    return 0;
}

int durf_mlp_bwd_f32(void* stream, int width, int in_dim, size_t rows, int N, const float* draw,
                     const int32_t* ray_idx, const int32_t* count, const float* mlp_params, const float* act,
                     float* dz, float* d_enc) {
    DURF_REQUIRE(width == 256 || width == 128, "width must be 256 or 128");
    DURF_REQUIRE(in_dim > 0 && in_dim <= DURF_ENC_DIM, "in_dim <= 64");
    if (rows == 0) return 0;
    const F32Spec S = f32_spec(width, in_dim);
    const int lds = 2 * F32_BUF * (int)sizeof(float);
    (void)hipFuncSetAttribute((const void*)k_mlp_bwd_f32, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL(k_mlp_bwd_f32, dim3(durf_cdiv(rows, 32)), dim3(64), lds, (hipStream_t)stream, S, rows, N, draw,
                       ray_idx, count, mlp_params, act, dz, d_enc);
    DURF_CHECK_LAUNCH("durf_mlp_bwd_f32");
    return 0;
}

int durf_mlp_dw_f32(void* stream, int width, int in_dim, size_t rows, int N, const int32_t* count, const float* act,
                    const float* dz, int nsplit, float* scratch, int32_t* tiles_dev, float* grad_mlp) {
    DURF_REQUIRE(width == 256 || width == 128, "width must be 256 or 128");
    DURF_REQUIRE(nsplit >= 1 && nsplit <= 1024, "1 <= nsplit <= 1024");
    const F32Spec S = f32_spec(width, in_dim);
    const size_t params = durf_layer_offset(width, in_dim, 12, 0);
    hipStream_t s = (hipStream_t)stream;
    // tile list (layer, in-feature tile incl. the bias row, out-feature tile): <= 12 * 10 * 8 entries
    static thread_local int tiles[3 * 1024];
    int nt = 0;
    for (int l = 0; l < 12; l++)
        for (int ki = 0; ki < (S.L[l].fi + 1 + 31) / 32; ki++)
            for (int mj = 0; mj < (S.L[l].fo + 31) / 32; mj++) {
                tiles[3 * nt] = l; tiles[3 * nt + 1] = ki; tiles[3 * nt + 2] = mj; nt++;
            }
    hipError_t e = hipMemcpyAsync(tiles_dev, tiles, sizeof(int) * 3 * nt, hipMemcpyHostToDevice, s);
    if (e != hipSuccess) { durf_set_error("durf_mlp_dw_f32: %s", hipGetErrorString(e)); return (int)e; }
    e = hipMemsetAsync(scratch, 0, sizeof(float) * params * nsplit, s);
    if (e != hipSuccess) { durf_set_error("durf_mlp_dw_f32: %s", hipGetErrorString(e)); return (int)e; }
    if (rows > 0)
        hipLaunchKernelGGL(k_mlp_dw_f32, dim3(nt, nsplit), dim3(64), 0, s, S, rows, N, count, act, dz, (const int*)tiles_dev,
                           nsplit, params, scratch);
    hipLaunchKernelGGL(k_dw_f32_reduce, dim3(durf_cdiv(params, 256)), dim3(256), 0, s, params, nsplit, scratch, grad_mlp);
    DURF_CHECK_LAUNCH("durf_mlp_dw_f32");
    return 0;
}

}  // extern "C"
