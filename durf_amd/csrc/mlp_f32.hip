// Exact-fp32 MLP on v_mfma_f32_32x32x2_f32 (exact fp32 products and accumulation: bitwise an fmaf chain over k).
//
// Two users:
//  * the OBJECT BRANCH of a step with box-pose optimisation on (MipNerfModel.object_precision() == 'f32', cfg4):
//    d(loss)/d(box pose) is a sum over the box-hit rays that cancels to ~1 % of its summed magnitudes, and bf16
//    rounding anywhere on those rays (object MLP or the background MLP's one evaluation per hit ray) shows up as
//    tens of per cent on it (tools/pose_grad_ablate.py).  So the hit rays -- 5-15 % of a batch -- are
//    evaluated in the reference's own arithmetic type (obbpose_model.py:326-327, internal/math.py:22-24);
//  * the PARITY INSTRUMENT (MipNerfModel.mlp_precision = 'f32', SURVEY.md 8c "F32_EXACT"): every MLP of the model.
//
// One workgroup = 32 samples (one MFMA N tile) x W/32 waves; wave w owns output tile w of every Dense (M split), the
// activations pass between layers through LDS as x[feature][sample], and the fp32 flax-layout weights stream
// L2 -> registers -> LDS in chunks of KC input rows, double buffered, the next chunk (also across layer boundaries)
// in flight behind the current chunk's MFMAs.  The 1- and 3-wide heads (density, rgb) are VALU dot products.
// The backward runs the same loop on per-layer TRANSPOSED weights (durf_mlp_f32_transpose, once per step).
//
// Per-sample records, stored per 32-sample tile as [tile][float index][32 samples] ("tile-transposed", so that a
// wave's accumulator registers store and load them as full 128-byte lines):
//   act (ACT floats): the INPUT of every Dense, concatenations included,
//     x0 = enc | x1..x4 = h0..h3 | x5 = [h4, enc] | x6, x7 = h5, h6 | x8 = h7 (density head and bottleneck) |
//     x10 = [bottleneck, view] | x11 = hc
//   dz (DZ floats): d(loss)/d(pre-activation) of every Dense output, in Dense order.
// Weight gradients: dW_l = x_l^T dz_l (the bias is a row of ones appended to x_l), split over the samples, partials
// summed in a fixed order (deterministic, no atomics).
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

#define F32_XS 33                    // LDS row stride of the activation buffers (floats): odd, so that both the
                                     // [feature][sample] and the transposing accesses are bank-conflict-free
template <int W>
struct F32Cfg {
    static constexpr int NW = W / 32;            // waves per workgroup = output tiles of a W-wide Dense
    static constexpr int NT = NW * 64;
    static constexpr int KC = W == 128 ? 32 : 16; // input rows per weight chunk
    static constexpr int XROWS = W + 64;         // widest Dense input (W + 63) padded
    static constexpr int CMAX = W + 64;          // widest chunk (backward of Dense_5: W + in_dim outputs)
    static constexpr int PF = KC * CMAX / NT;    // prefetch registers per thread
    static constexpr int LDS_FLOATS = 2 * XROWS * F32_XS + 2 * 64 * F32_XS + 2 * KC * CMAX + 8 * 4 * 32 + 4 * 32;
};

__device__ __forceinline__ int c_row(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

__device__ __forceinline__ size_t f32_rows(size_t rows, int N, const int32_t* count) {
    if (!count) return rows;
    const size_t c = (size_t)(*count) * (size_t)N;
    return c < rows ? c : rows;
}

// ---- weight chunks: global -> registers (issue) -> LDS (commit) ---------------------------------------------------
// chunk = rows [r0, r0 + KC) x columns [0, C) of a row-major matrix with `stride` floats per row, `R` valid rows and
// `cv` valid columns (the rest is zero-filled: partial last chunk, padded output tiles)
template <int C, int KC, int NT, int PF>
__device__ __forceinline__ void chunk_issue(float (&pf)[PF], const float* __restrict__ M, int stride, int R, int cv, int r0,
                                            int tid) {
    constexpr int E = KC * C / NT;
    static_assert(E * NT == KC * C && E <= PF, "chunk does not divide over the workgroup");
#pragma unroll
    for (int j = 0; j < E; j++) {
        const int idx = tid + NT * j;
        const int r = idx / C, c = idx - r * C;
        pf[j] = (r0 + r < R && c < cv) ? M[(size_t)(r0 + r) * stride + c] : 0.0f;
    }
}
template <int C, int KC, int NT, int PF>
__device__ __forceinline__ void chunk_commit(const float (&pf)[PF], float* wb, int tid) {
    constexpr int E = KC * C / NT;
#pragma unroll
    for (int j = 0; j < E; j++) wb[tid + NT * j] = pf[j];
}

// acc[j] += chunk^T x for this wave's output tiles mo = wave + NW j  (A = chunk[k][32 mo + m], B = x[k][n])
template <int C, int KC, int NW, int TPW>
__device__ __forceinline__ void chunk_mma(const float* wb, const float* xr, int wave, int lane, f32x16 (&acc)[TPW]) {
    const int m = lane & 31, kk = lane >> 5;
    const bool two = TPW > 1 && wave + NW < C / 32;
    if (wave >= C / 32) return;
    if (two) {
#pragma unroll
        for (int ks = 0; ks < KC / 2; ks++) {
            const float b = xr[(2 * ks + kk) * F32_XS + m];
            const float a0 = wb[(2 * ks + kk) * C + 32 * wave + m];
            const float a1 = wb[(2 * ks + kk) * C + 32 * (wave + NW) + m];
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b, acc[0], 0, 0, 0);
            acc[TPW - 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b, acc[TPW - 1], 0, 0, 0);
        }
    } else {
#pragma unroll
        for (int ks = 0; ks < KC / 2; ks++) {
            const float b = xr[(2 * ks + kk) * F32_XS + m];
            const float a0 = wb[(2 * ks + kk) * C + 32 * wave + m];
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b, acc[0], 0, 0, 0);
        }
    }
}

// State of the weight pipeline: which LDS buffer holds the next chunk to consume
struct F32Pipe { float* wbuf; int par; int tid, wave, lane; };

// One Dense through the chunk pipeline.  On entry the registers `pf` hold chunk 0 of THIS matrix (issued by the
// previous call / the prologue); on exit they hold chunk 0 of the NEXT matrix (Mn; nullptr: none).
//   M [R][stride] row-major, cv valid columns of C;  x: LDS activations [>= ceil(R / KC) KC][F32_XS]
template <int W, int C, int CN, int TPW>
__device__ __forceinline__ void dense_f32(F32Pipe& p, float (&pf)[F32Cfg<W>::PF], const float* __restrict__ M, int stride,
                                          int R, int cv, const float* x, f32x16 (&acc)[TPW],
                                          const float* __restrict__ Mn, int stride_n, int Rn, int cvn) {
    using Cf = F32Cfg<W>;
    constexpr int KC = Cf::KC, NT = Cf::NT, NW = Cf::NW, PF = Cf::PF;
    constexpr int CB = KC * Cf::CMAX;                   // floats per LDS weight buffer
    const int nchunk = (R + KC - 1) / KC;
    chunk_commit<C, KC, NT, PF>(pf, p.wbuf + p.par * CB, p.tid);
    __syncthreads();                                    // chunk 0 and the previous layer's x are visible
    for (int c = 0; c < nchunk; c++) {
        const bool last = c + 1 == nchunk;
        if (!last) chunk_issue<C, KC, NT, PF>(pf, M, stride, R, cv, (c + 1) * KC, p.tid);
        else if (Mn) chunk_issue<CN, KC, NT, PF>(pf, Mn, stride_n, Rn, cvn, 0, p.tid);
        chunk_mma<C, KC, NW, TPW>(p.wbuf + p.par * CB, x + c * KC * F32_XS, p.wave, p.lane, acc);
        p.par ^= 1;
        if (!last) {
            chunk_commit<C, KC, NT, PF>(pf, p.wbuf + p.par * CB, p.tid);
            __syncthreads();
        }
    }
}

// Every workgroup reads every weight of its MLP once, chunk by chunk, one chunk ahead -- and a launch's workgroups all
// start together on an L2 that last saw these weights before the optimizer rewrote them, so every chunk would cost one
// L2 miss (~2 us) on every CU at once.  Instead each workgroup first touches one line in 128 bytes of a 1/8 slice of
// the parameters (workgroups are dealt round-robin to the 8 XCDs: id / 8 walks one XCD's workgroups), so that each
// XCD's L2 is filled once, early, by its own workgroups while the first layers run.  Returns a value that depends on
// the loads (the caller keeps it alive to the end of the kernel so that the wait for them is never on the critical path).
__device__ __forceinline__ float f32_warm_l2(const float* __restrict__ P, size_t params, int tid, int nt) {
    const unsigned wg = blockIdx.x + gridDim.x * blockIdx.y;
    const size_t lines = (params + 31) / 32, per = (lines + 7) / 8;
    const size_t l0 = (size_t)((wg >> 3) & 7) * per;
    float w = 0.0f;
    for (size_t j = tid; j < per; j += nt) {
        const size_t l = l0 + j;
        if (l < lines) w += P[l * 32];
    }
    return w;
}
#ifndef F32_WARM
#define F32_WARM 1
#endif

struct F32FwdBatch { size_t enc, idx, params, raw, act; };        // per-object strides (floats; idx: int32 elements)
struct F32BwdBatch { size_t idx, params, act, dz, d_enc; };

// ---------------------------------------------------------------------------------------------
// forward: obbpose_model.py:305-354 / :369-418 in fp32
//   enc == nullptr: every row is evaluated on the constant encoding of a zero-masked Gaussian ([0 x 30, 1 x 30]): the
//   background MLP's single evaluation of a box-hit ray (obbpose_model.py:205-210; include/durf_hip.h durf_expand_raw)
// ---------------------------------------------------------------------------------------------
template <int W>
__global__ void __launch_bounds__(W * 2)
k_mlp_fwd_f32(F32Spec S, size_t rows, int N, const float* __restrict__ enc, const float* __restrict__ view,
              const int32_t* __restrict__ ray_idx, const int32_t* __restrict__ count,
              const float* __restrict__ P, float* __restrict__ raw, float* __restrict__ act, F32FwdBatch bs) {
    using Cf = F32Cfg<W>;
    constexpr int NT = Cf::NT, XR = Cf::XROWS;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* xa = lds;
    float* xb = xa + XR * F32_XS;
    float* encs = xb + XR * F32_XS;                     // the block's encoding, kept for the skip concatenation
    float* wbuf = encs + 2 * 64 * F32_XS;               // (second half of the enc region: backward only)
    float* red = wbuf + 2 * Cf::KC * Cf::CMAX;          // [8 parts][4 outputs][32 samples] head partial sums
    if (gridDim.y > 1) {
        const size_t k = blockIdx.y;
        if (enc) enc += k * bs.enc;
        ray_idx += k * bs.idx; count += k; P += k * bs.params; raw += k * bs.raw;
        if (act) act += k * bs.act;
    }
    const size_t nrows = f32_rows(rows, N, count);
    const size_t row0 = (size_t)blockIdx.x * 32;
    if (row0 >= nrows) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 31, hi = lane >> 5;
    const int in_dim = S.in_dim;
    float* at = act ? act + (size_t)blockIdx.x * S.act * 32 : nullptr;         // this tile's record block
    F32Pipe p{wbuf, 0, tid, wave, lane};
    float pf[Cf::PF];

    // x0 = enc (rows in_dim..63 zero)
    for (int idx = tid; idx < 64 * 32; idx += NT) {
        const int f = idx & 63, nn = idx >> 6;
        float v = 0.0f;
        if (f < in_dim && row0 + nn < nrows) v = enc ? enc[(row0 + nn) * (size_t)in_dim + f] : ((f >= 30 && f < 60) ? 1.0f : 0.0f);
        encs[f * F32_XS + nn] = v;
        xa[f * F32_XS + nn] = v;
    }
    chunk_issue<W, Cf::KC, NT, Cf::PF>(pf, P + S.L[0].w_off, W, in_dim, W, 0, tid);
    const float warm = F32_WARM ? f32_warm_l2(P, S.L[11].w_off + 128 * 3 + 3, tid, NT) : 0.0f;
    __syncthreads();
    if (at)
        for (int idx = tid; idx < in_dim * 32; idx += NT) at[(S.L[0].x_off + (idx >> 5)) * 32 + (idx & 31)] = encs[(idx >> 5) * F32_XS + (idx & 31)];

    float* cur = xa;
    float* nxt = xb;
    f32x16 acc[1];
    auto init_bias = [&](const float* bias, int fo) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int o = 32 * wave + c_row(r, hi);
            acc[0][r] = (wave < fo / 32) ? bias[o] : 0.0f;
        }
    };
    // this wave's output tile -> the next Dense's input (LDS) and its record (global)
    auto store_out = [&](int fo, int relu, int x_off_next) {
        if (wave < fo / 32) {
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int o = 32 * wave + c_row(r, hi);
                float v = acc[0][r];
                if (relu) v = (v != v) ? v : fmaxf(v, 0.0f);        // jnp.maximum(x, 0) propagates NaN; fmaxf would drop it
                nxt[o * F32_XS + n] = v;
                if (at) at[(x_off_next + o) * 32 + n] = v;
            }
        }
    };
    // 1- / 3-wide heads: out[c][n] = b[c] + sum_k Wl[k][c] x[k][n] as 8 interleaved partial chains, summed in order
    auto head = [&](const float* Wl, int fi, int fo, float (&out)[3]) {
        const int part = tid >> 5, nn = tid & 31;
        if (part < 8) {
            float s[3] = {0.0f, 0.0f, 0.0f};
            for (int k = part; k < fi; k += 8) {
                const float xv = cur[k * F32_XS + nn];
                for (int c = 0; c < fo; c++) s[c] = fmaf(Wl[k * fo + c], xv, s[c]);
            }
            for (int c = 0; c < fo; c++) red[(part * 4 + c) * 32 + nn] = s[c];
        }
        __syncthreads();
        if (tid < 32)
            for (int c = 0; c < fo; c++) {
                float s = Wl[fi * fo + c];
                for (int q = 0; q < 8; q++) s += red[(q * 4 + c) * 32 + nn];
                out[c] = s;
            }
        __syncthreads();
    };
    const auto& L = S.L;
#define WL(l) (P + L[l].w_off)
#define SWAP() { float* t_ = cur; cur = nxt; nxt = t_; }
    // Dense_0 .. Dense_3
    init_bias(WL(0) + (size_t)L[0].fi * W, W);
    dense_f32<W, W, W, 1>(p, pf, WL(0), W, L[0].fi, W, cur, acc, WL(1), W, W, W);
    store_out(W, 1, L[1].x_off); SWAP();
    for (int l = 1; l <= 3; l++) {
        init_bias(WL(l) + (size_t)W * W, W);
        dense_f32<W, W, W, 1>(p, pf, WL(l), W, W, W, cur, acc, WL(l + 1), W, L[l + 1].fi, W);
        store_out(W, 1, L[l + 1].x_off); SWAP();
    }
    // Dense_4 -> x5 = [h4, enc]   (obbpose_model.py:333-334)
    init_bias(WL(4) + (size_t)W * W, W);
    dense_f32<W, W, W, 1>(p, pf, WL(4), W, W, W, cur, acc, WL(5), W, L[5].fi, W);
    store_out(W, 1, L[5].x_off);
    for (int idx = tid; idx < 64 * 32; idx += NT) {
        const int f = idx >> 5, nn = idx & 31;
        const float v = encs[f * F32_XS + nn];
        nxt[(W + f) * F32_XS + nn] = v;
        if (at && f < in_dim) at[(L[5].x_off + W + f) * 32 + nn] = v;
    }
    SWAP();
    // Dense_5 .. Dense_7
    for (int l = 5; l <= 7; l++) {
        init_bias(WL(l) + (size_t)L[l].fi * W, W);
        // after Dense_7 comes the bottleneck (Dense_9): the density head (Dense_8) is a VALU dot product
        const int ln = l == 7 ? 9 : l + 1;
        dense_f32<W, W, W, 1>(p, pf, WL(l), W, L[l].fi, W, cur, acc, WL(ln), W, W, W);
        store_out(W, 1, L[ln == 9 ? 8 : ln].x_off); SWAP();
    }
    __syncthreads();
    float dens3[3], rgb[3];
    head(WL(8), W, 1, dens3);                          // density head on h7
    // Dense_9 (bottleneck, linear) -> x10 = [bottleneck, view]   (:339, :346-347)
    init_bias(WL(9) + (size_t)W * W, W);
    dense_f32<W, W, 128, 1>(p, pf, WL(9), W, W, W, cur, acc, WL(10), 128, L[10].fi, 128);
    store_out(W, 0, L[10].x_off);
    for (int idx = tid; idx < 32 * 32; idx += NT) {
        const int f = idx >> 5, nn = idx & 31;
        float v = 0.0f;
        if (f < 27 && row0 + nn < nrows) {
            size_t ray = (row0 + nn) / (size_t)N;
            if (ray_idx) ray = (size_t)ray_idx[ray];
            v = view[ray * 27 + f];
        }
        nxt[(W + f) * F32_XS + nn] = v;
        if (at && f < 27) at[(L[10].x_off + W + f) * 32 + nn] = v;
    }
    SWAP();
    // Dense_10 (view layer, 128 wide, relu) -> hc
    init_bias(WL(10) + (size_t)L[10].fi * 128, 128);
    dense_f32<W, 128, 128, 1>(p, pf, WL(10), 128, L[10].fi, 128, cur, acc, nullptr, 0, 0, 0);
    store_out(128, 1, L[11].x_off); SWAP();
    __syncthreads();
    head(WL(11), 128, 3, rgb);                         // rgb head on hc
    if (tid < 32 && row0 + tid < nrows) {
        const f32x4 o = {rgb[0], rgb[1], rgb[2], dens3[0]};
        *(f32x4*)(raw + (row0 + tid) * 4) = o;
    }
    if (warm == 1.2345e-33f) raw[0] = warm;            // never true for real parameters; keeps the warm-up loads alive
#undef WL
#undef SWAP
}

// ---------------------------------------------------------------------------------------------
// per-layer transposed weights for the backward: PT[w_off(l) + m * fi + k] = P[w_off(l) + k * fo + m]
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_f32_transpose(F32Spec S, size_t params, const float* __restrict__ P, float* __restrict__ PT, size_t p_stride,
                size_t pt_stride) {
    P += blockIdx.y * p_stride; PT += blockIdx.y * pt_stride;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= params) return;
    int l = 11;
    while (l > 0 && i < S.L[l].w_off) l--;
    const size_t j = i - S.L[l].w_off;
    const int fi = S.L[l].fi, fo = S.L[l].fo;
    if (j >= (size_t)fi * fo) { PT[i] = P[i]; return; }         // bias: copied
    const int m = (int)(j / fi), k = (int)(j - (size_t)m * fi);
    PT[i] = P[S.L[l].w_off + (size_t)k * fo + m];
}

// ---------------------------------------------------------------------------------------------
// backward (data path): d(loss)/d(pre-activation) of every Dense, optionally d(loss)/d(enc)
// ---------------------------------------------------------------------------------------------
template <int W>
__global__ void __launch_bounds__(W * 2)
k_mlp_bwd_f32(F32Spec S, size_t rows, int N, const float* __restrict__ draw, const int32_t* __restrict__ ray_idx,
              const int32_t* __restrict__ count, const float* __restrict__ P, const float* __restrict__ PT,
              const float* __restrict__ act, float* __restrict__ dz, float* __restrict__ d_enc, F32BwdBatch bs) {
    using Cf = F32Cfg<W>;
    constexpr int NT = Cf::NT, XR = Cf::XROWS, CE = W + 64;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* xa = lds;
    float* xb = xa + XR * F32_XS;
    float* denc = xb + XR * F32_XS;                     // [64][XS] d(enc), accumulated over Dense_5 and Dense_0
    float* wbuf = denc + 2 * 64 * F32_XS;
    float* gsm = wbuf + 2 * Cf::KC * Cf::CMAX + 8 * 4 * 32;     // [4][32] head gradients of the block's samples
    if (gridDim.y > 1) {
        const size_t k = blockIdx.y;
        ray_idx += k * bs.idx; count += k; P += k * bs.params; PT += k * bs.params; act += k * bs.act; dz += k * bs.dz;
        if (d_enc) d_enc += k * bs.d_enc;
    }
    const size_t nrows = f32_rows(rows, N, count);
    const size_t row0 = (size_t)blockIdx.x * 32;
    if (row0 >= nrows) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 31, hi = lane >> 5;
    const int in_dim = S.in_dim;
    const float* at = act + (size_t)blockIdx.x * S.act * 32;
    float* zt = dz + (size_t)blockIdx.x * S.dz * 32;
    const auto& L = S.L;
    F32Pipe p{wbuf, 0, tid, wave, lane};
    float pf[Cf::PF];
#define WT(l) (PT + L[l].w_off)
    // Dense_10's transposed kernel [128][W + 27]: only the W bottleneck-fed columns are propagated
    chunk_issue<W, Cf::KC, NT, Cf::PF>(pf, WT(10), L[10].fi, 128, W, 0, tid);
    const float warm = F32_WARM ? f32_warm_l2(PT, L[11].w_off, tid, NT) : 0.0f;
    // head gradients (object MLPs gather their rows of the [B*N,4] buffer through ray_idx); rows past nrows: zero
    if (tid < 32) {
        const size_t row = row0 + tid;
        f32x4 g = {0.f, 0.f, 0.f, 0.f};
        if (row < nrows) {
            size_t src = row;
            if (ray_idx) src = (size_t)ray_idx[row / (size_t)N] * (size_t)N + row % (size_t)N;
            g = *(const f32x4*)(draw + src * 4);
        }
#pragma unroll
        for (int c = 0; c < 4; c++) gsm[c * 32 + tid] = g[c];
        zt[(L[11].dz_off + 0) * 32 + tid] = g[0]; zt[(L[11].dz_off + 1) * 32 + tid] = g[1];
        zt[(L[11].dz_off + 2) * 32 + tid] = g[2]; zt[L[8].dz_off * 32 + tid] = g[3];
    }
    for (int idx = tid; idx < 64 * 32; idx += NT) denc[(idx >> 5) * F32_XS + (idx & 31)] = 0.0f;
    __syncthreads();
    float* cur = xa;
    float* nxt = xb;
    // Dense_11 (rgb head, 128 -> 3): d hc = W11 dz11, masked by Dense_10's ReLU -> dz10
    {
        const float* W11 = P + L[11].w_off;
        for (int idx = tid; idx < 128 * 32; idx += NT) {
            const int k = idx >> 5, nn = idx & 31;
            float v = 0.0f;
#pragma unroll
            for (int c = 0; c < 3; c++) v = fmaf(W11[k * 3 + c], gsm[c * 32 + nn], v);
            const float h = at[(L[11].x_off + k) * 32 + nn];
            v = h > 0.0f ? v : 0.0f;
            cur[k * F32_XS + nn] = v;
            zt[(L[10].dz_off + k) * 32 + nn] = v;
        }
    }
    f32x16 acc[2];
    auto zero_acc = [&]() {
#pragma unroll
        for (int r = 0; r < 16; r++) { acc[0][r] = 0.0f; acc[1][r] = 0.0f; }
    };
    // d x (first Wp features) -> dz of the producing Dense (masked by its ReLU output, read from the act record at h_off)
    auto store_dz = [&](int Wp, int relu, int h_off, int dz_off, const float* extra_w) {
        if (wave < Wp / 32) {
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int k = 32 * wave + c_row(r, hi);
                float v = acc[0][r];
                if (extra_w) v += extra_w[k] * gsm[3 * 32 + n];          // h7 also feeds the density head: + W8 dz8
                if (relu) { const float h = at[(h_off + k) * 32 + n]; v = h > 0.0f ? v : 0.0f; }
                nxt[k * F32_XS + n] = v;
                zt[(dz_off + k) * 32 + n] = v;
            }
        }
    };
#define SWAP() { float* t_ = cur; cur = nxt; nxt = t_; }
    // Dense_10: d [bottleneck] = W10[:W] dz10 -> dz9 (linear)
    zero_acc();
    dense_f32<W, W, W, 2>(p, pf, WT(10), L[10].fi, 128, W, cur, acc, WT(9), W, W, W);
    store_dz(W, 0, 0, L[9].dz_off, nullptr); SWAP();
    // Dense_9 (+ Dense_8): d h7 -> dz7
    zero_acc();
    dense_f32<W, W, W, 2>(p, pf, WT(9), W, W, W, cur, acc, WT(7), W, W, W);
    store_dz(W, 1, L[8].x_off, L[7].dz_off, P + L[8].w_off); SWAP();
    // Dense_7, Dense_6: d h6 -> dz6, d h5 -> dz5
    zero_acc();
    dense_f32<W, W, W, 2>(p, pf, WT(7), W, W, W, cur, acc, WT(6), W, W, W);
    store_dz(W, 1, L[7].x_off, L[6].dz_off, nullptr); SWAP();
    zero_acc();
    if (d_enc) dense_f32<W, W, CE, 2>(p, pf, WT(6), W, W, W, cur, acc, WT(5), L[5].fi, W, L[5].fi);
    else dense_f32<W, W, W, 2>(p, pf, WT(6), W, W, W, cur, acc, WT(5), L[5].fi, W, W);
    store_dz(W, 1, L[6].x_off, L[5].dz_off, nullptr); SWAP();
    // Dense_5: d [h4, enc] -> dz4 and (skip connection) d enc
    zero_acc();
    if (d_enc) {
        dense_f32<W, CE, W, 2>(p, pf, WT(5), L[5].fi, W, L[5].fi, cur, acc, WT(4), W, W, W);
        // tiles W/32 and W/32 + 1 are the d enc part: wave t of the second round owns tile NW + t
        if (wave < 2) {
#pragma unroll
            for (int r = 0; r < 16; r++) denc[(32 * wave + c_row(r, hi)) * F32_XS + n] = acc[1][r];
        }
    } else {
        dense_f32<W, W, W, 2>(p, pf, WT(5), L[5].fi, W, W, cur, acc, WT(4), W, W, W);
    }
    store_dz(W, 1, L[5].x_off, L[4].dz_off, nullptr); SWAP();
    // Dense_4 .. Dense_1
    for (int l = 4; l >= 1; l--) {
        zero_acc();
        if (l > 1) dense_f32<W, W, W, 2>(p, pf, WT(l), W, W, W, cur, acc, WT(l - 1), W, W, W);
        else if (d_enc) dense_f32<W, W, 64, 2>(p, pf, WT(1), W, W, W, cur, acc, WT(0), in_dim, W, in_dim);
        else dense_f32<W, W, W, 2>(p, pf, WT(1), W, W, W, cur, acc, nullptr, 0, 0, 0);
        store_dz(W, 1, L[l].x_off, L[l - 1].dz_off, nullptr); SWAP();
    }
    // Dense_0: d enc += W0 dz0
    if (d_enc) {
        zero_acc();
        dense_f32<W, 64, W, 2>(p, pf, WT(0), in_dim, W, in_dim, cur, acc, nullptr, 0, 0, 0);
        if (wave < 2) {
#pragma unroll
            for (int r = 0; r < 16; r++) denc[(32 * wave + c_row(r, hi)) * F32_XS + n] += acc[0][r];
        }
        __syncthreads();
        for (int idx = tid; idx < 32 * 64; idx += NT) {
            const int nn = idx >> 6, k = idx & 63;
            if (row0 + nn < nrows) d_enc[(row0 + nn) * 64 + k] = denc[k * F32_XS + nn];
        }
    }
    if (warm == 1.2345e-33f) zt[0] = warm;             // never true for real parameters; keeps the warm-up loads alive
#undef WT
#undef SWAP
}

// ---------------------------------------------------------------------------------------------
// weight gradients: dW_l[k, m] = sum_n x_l[n, k] dz_l[n, m]  (row fi of x_l = 1: the bias).  One workgroup (4 waves)
// per (32 x 32 output tile, sample split, object); a wave walks its share of the 32-sample tiles -- the A / B
// operands of a tile's 16 MFMAs are 4 + 4 float4 per lane straight from the tile-transposed records (a lane's 16
// samples are the contiguous half [16 kk, 16 kk + 16) of a 128-byte line) -- then the four waves' accumulators are
// summed in order through LDS.  Partials are summed over the splits in a fixed order by k_dw_f32_reduce.
// ---------------------------------------------------------------------------------------------
#define F32_MAX_SEG 4
struct F32DwSeg { const float* act; const float* dz; const int32_t* count; size_t rows; int N; };
struct F32DwArgs { F32DwSeg seg[F32_MAX_SEG]; int nseg; size_t act_stride, dz_stride, part_stride; };
struct F32TileTab { int base[13]; };          // output tiles of Dense_l: [base[l], base[l + 1]), k-tile major

__global__ void __launch_bounds__(256)
k_mlp_dw_f32(F32Spec S, F32TileTab T, F32DwArgs a, int nsplit, size_t params, float* __restrict__ part) {
    __shared__ float red[3][16][64];
    const int t = blockIdx.x, sp = blockIdx.y;
    int l = 11;
    while (l > 0 && t < T.base[l]) l--;
    const F32Layer Ly = S.L[l];
    const int ntm = (Ly.fo + 31) / 32;
    const int ki = (t - T.base[l]) / ntm, mj = (t - T.base[l]) % ntm;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, kk = lane >> 5;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0.0f;
    const int k = 32 * ki + i, m = 32 * mj + i;
    const int krow = Ly.x_off + (k < Ly.fi ? k : 0);            // clamped: rows >= fi are the bias row / padding
    const int mrow = Ly.dz_off + (m < Ly.fo ? m : 0);
    for (int sgi = 0; sgi < a.nseg; sgi++) {
        const F32DwSeg sg = a.seg[sgi];
        const size_t nrows = f32_rows(sg.rows, sg.N, sg.count ? sg.count + blockIdx.z : nullptr);
        const size_t ntile = (nrows + 31) / 32;
        const float* act = sg.act + blockIdx.z * a.act_stride;
        const float* dz = sg.dz + blockIdx.z * a.dz_stride;
        // tiles are dealt round-robin to (split, wave): every split of every segment gets its share; the next tile's
        // operands are in flight behind the current tile's 16 MFMAs
        const size_t step = (size_t)nsplit * 4;
        size_t tl = (size_t)sp * 4 + wave;
        f32x4 av[4], bv[4], an[4], bn[4];
        auto load = [&](size_t t_, f32x4 (&a_)[4], f32x4 (&b_)[4]) {
            const float* xa = act + (t_ * S.act + krow) * 32 + 16 * kk;
            const float* za = dz + (t_ * S.dz + mrow) * 32 + 16 * kk;
#pragma unroll
            for (int q = 0; q < 4; q++) { a_[q] = *(const f32x4*)(xa + 4 * q); b_[q] = *(const f32x4*)(za + 4 * q); }
        };
        if (tl < ntile) load(tl, av, bv);
        for (; tl < ntile; tl += step) {
            const bool more = tl + step < ntile;
            if (more) load(tl + step, an, bn);
            const size_t s0 = tl * 32 + 16 * kk;
#pragma unroll
            for (int j = 0; j < 16; j++) {
                const bool ok = s0 + j < nrows;
                float x = av[j >> 2][j & 3], z = bv[j >> 2][j & 3];
                x = k < Ly.fi ? x : (k == Ly.fi ? 1.0f : 0.0f);
                x = ok ? x : 0.0f;
                z = (ok && m < Ly.fo) ? z : 0.0f;
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x, z, acc, 0, 0, 0);
            }
            if (more) {
#pragma unroll
                for (int q = 0; q < 4; q++) { av[q] = an[q]; bv[q] = bn[q]; }
            }
        }
    }
    if (wave > 0) {
#pragma unroll
        for (int r = 0; r < 16; r++) red[wave - 1][r][lane] = acc[r];
    }
    __syncthreads();
    if (wave == 0) {
        float* pp = part + blockIdx.z * a.part_stride + (size_t)sp * params + Ly.w_off;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const float v = ((acc[r] + red[0][r][lane]) + red[1][r][lane]) + red[2][r][lane];
            const int kr = 32 * ki + c_row(r, kk);
            if (kr <= Ly.fi && m < Ly.fo) pp[(size_t)kr * Ly.fo + m] = v;
        }
    }
}

__global__ void __launch_bounds__(256)
k_dw_f32_reduce(size_t params, int nsplit, const float* __restrict__ part, size_t part_stride, float* __restrict__ grad,
                size_t grad_stride, int accumulate) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= params) return;
    part += blockIdx.y * part_stride; grad += blockIdx.y * grad_stride;
    float s = 0.0f;
    for (int sp = 0; sp < nsplit; sp++) s += part[(size_t)sp * params + i];
    grad[i] = accumulate ? grad[i] + s : s;
}

// ---------------------------------------------------------------------------------------------
// The background MLP's evaluation of the box-hit rays in fp32.  Every such ray feeds the trunk the SAME input (the
// encoding of a zero-masked Gaussian, [0 x 30, 1 x 30]: obbpose_model.py:205-210), so Dense_0 .. Dense_9 are evaluated
// ONCE per step (k_bkgd_const_trunk: one workgroup, matrix-vector products, 4 interleaved fmaf chains per output) and
// only the view layer and the rgb head per ray (k_bkgd_hit_rays: 36 k MACs per ray instead of 592 k).
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024)
k_bkgd_const_trunk(F32Spec S, const float* __restrict__ P, float* __restrict__ out /* [257]: bottleneck, density */) {
    __shared__ float x[2][320];
    __shared__ float part[4][256];
    const int tid = threadIdx.x, o = tid & 255, g = tid >> 8;
    if (tid < 60) { const float v = tid >= 30 ? 1.0f : 0.0f; x[0][tid] = v; x[1][256 + tid] = v; }     // x0; enc half of x5
    __syncthreads();
    int cur = 0;
    for (int l = 0; l <= 9; l++) {
        if (l == 8) continue;
        const F32Layer Ly = S.L[l];
        const float* Wl = P + Ly.w_off;
        const float* xin = x[cur];
        float s = 0.0f;
#pragma unroll 16
        for (int k = g; k < Ly.fi; k += 4) s = fmaf(Wl[(size_t)k * 256 + o], xin[k], s);
        part[g][o] = s;
        __syncthreads();
        if (g == 0) {
            float v = (((Wl[(size_t)Ly.fi * 256 + o] + part[0][o]) + part[1][o]) + part[2][o]) + part[3][o];
            if (Ly.relu) v = (v != v) ? v : fmaxf(v, 0.0f);
            if (l == 9) out[o] = v;
            else x[cur ^ 1][o] = v;
        }
        cur ^= 1;          // ping-pong; Dense_4's output lands in x[1], whose tail already holds the encoding: x5 = [h4, enc]
        __syncthreads();
        if (l == 7 && tid < 64) {                      // density head on h7 (Dense_8): one wave, lane-strided chains
            const float* W8 = P + S.L[8].w_off;
            float d = 0.0f;
            for (int k = tid; k < 256; k += 64) d = fmaf(W8[k], x[cur][k], d);
            d = wave_sum(d);
            if (tid == 0) out[256] = d + W8[256];
        }
    }
}

#define HITRAYS_PER_WG 4
__global__ void __launch_bounds__(128)
k_bkgd_hit_rays(F32Spec S, const float* __restrict__ P, const float* __restrict__ trunk, const float* __restrict__ view27,
                const int32_t* __restrict__ idx, const int32_t* __restrict__ count, float* __restrict__ raw_tail) {
    __shared__ float x10[HITRAYS_PER_WG][288];
    __shared__ float hc[HITRAYS_PER_WG][128];
    const int n = *count;
    const int j0 = blockIdx.x * HITRAYS_PER_WG;
    if (j0 >= n) return;
    const int tid = threadIdx.x;
    for (int i = tid; i < HITRAYS_PER_WG * 283; i += 128) {
        const int r = i / 283, f = i - r * 283;
        float v = 0.0f;
        if (j0 + r < n) v = f < 256 ? trunk[f] : view27[(size_t)idx[j0 + r] * 27 + (f - 256)];
        x10[r][f] = v;
    }
    __syncthreads();
    const float* W10 = P + S.L[10].w_off;
    float acc[HITRAYS_PER_WG];
#pragma unroll
    for (int r = 0; r < HITRAYS_PER_WG; r++) acc[r] = W10[283 * 128 + tid];
#pragma unroll 8
    for (int k = 0; k < 283; k++) {
        const float w = W10[k * 128 + tid];
#pragma unroll
        for (int r = 0; r < HITRAYS_PER_WG; r++) acc[r] = fmaf(w, x10[r][k], acc[r]);
    }
#pragma unroll
    for (int r = 0; r < HITRAYS_PER_WG; r++) { const float v = acc[r]; hc[r][tid] = (v != v) ? v : fmaxf(v, 0.0f); }
    __syncthreads();
    if (tid < HITRAYS_PER_WG * 4) {
        const int r = tid >> 2, c = tid & 3;
        if (j0 + r < n) {
            float v;
            if (c < 3) {
                const float* W11 = P + S.L[11].w_off;
                v = W11[128 * 3 + c];
                for (int k = 0; k < 128; k++) v = fmaf(W11[k * 3 + c], hc[r][k], v);
            } else {
                v = trunk[256];
            }
            raw_tail[(size_t)(j0 + r) * 4 + c] = v;
        }
    }
}

namespace {

F32TileTab tile_table(const F32Spec& S) {
    F32TileTab T;
    int nt = 0;
    for (int l = 0; l < 12; l++) {            // k tiles cover fi inputs + the bias row
        T.base[l] = nt;
        nt += ((S.L[l].fi + 1 + 31) / 32) * ((S.L[l].fo + 31) / 32);
    }
    T.base[12] = nt;
    return T;
}

template <int W>
int launch_fwd(hipStream_t s, const F32Spec& S, size_t rows, int N, const float* enc, const float* view27,
               const int32_t* ray_idx, const int32_t* count, const float* P, float* raw, float* act, int K,
               const F32FwdBatch& bs) {
    constexpr int lds = F32Cfg<W>::LDS_FLOATS * (int)sizeof(float);
    (void)hipFuncSetAttribute((const void*)k_mlp_fwd_f32<W>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL(k_mlp_fwd_f32<W>, dim3(durf_cdiv(rows, 32), K), dim3(W * 2), lds, s, S, rows, N, enc, view27, ray_idx,
                       count, P, raw, act, bs);
    return 0;
}
template <int W>
int launch_bwd(hipStream_t s, const F32Spec& S, size_t rows, int N, const float* draw, const int32_t* ray_idx,
               const int32_t* count, const float* P, const float* PT, const float* act, float* dz, float* d_enc, int K,
               const F32BwdBatch& bs) {
    constexpr int lds = F32Cfg<W>::LDS_FLOATS * (int)sizeof(float);
    (void)hipFuncSetAttribute((const void*)k_mlp_bwd_f32<W>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL(k_mlp_bwd_f32<W>, dim3(durf_cdiv(rows, 32), K), dim3(W * 2), lds, s, S, rows, N, draw, ray_idx, count,
                       P, PT, act, dz, d_enc, bs);
    return 0;
}
size_t tile_rows(size_t rows) { return (rows + 31) / 32 * 32; }

}  // namespace

extern "C" {

size_t durf_mlp_f32_act_floats(int width, int in_dim) { return (size_t)f32_spec(width, in_dim).act; }
size_t durf_mlp_f32_dz_floats(int width, int in_dim) { return (size_t)f32_spec(width, in_dim).dz; }
size_t durf_mlp_f32_dw_scratch_floats(int width, int in_dim, int nsplit) {
    return (size_t)nsplit * durf_layer_offset(width, in_dim, 12, 0);
}

int durf_mlp_f32_transpose(void* stream, int width, int in_dim, int K, const float* mlp_params, size_t param_stride,
                           float* params_t) {
    DURF_REQUIRE(width == 256 || width == 128, "width must be 256 or 128");
    if (K <= 0) return 0;
    const F32Spec S = f32_spec(width, in_dim);
    const size_t params = durf_layer_offset(width, in_dim, 12, 0);
    hipLaunchKernelGGL(k_f32_transpose, dim3(durf_cdiv(params, 256), K), dim3(256), 0, (hipStream_t)stream, S, params,
                       mlp_params, params_t, param_stride, param_stride);
    DURF_CHECK_LAUNCH("durf_mlp_f32_transpose");
    return 0;
}

int durf_mlp_fwd_f32(void* stream, int width, int in_dim, size_t rows, int N, const float* enc, const float* view27,
                     const int32_t* ray_idx, const int32_t* count, const float* mlp_params, float* raw, float* act) {
    DURF_REQUIRE(width == 256 || width == 128, "width must be 256 or 128");
    DURF_REQUIRE(in_dim > 0 && in_dim <= DURF_ENC_DIM, "in_dim <= 64");
    DURF_REQUIRE(enc != nullptr || in_dim == 60, "the constant encoding (enc == NULL) is the background MLP's");
    if (rows == 0) return 0;
    const F32Spec S = f32_spec(width, in_dim);
    if (width == 256) launch_fwd<256>((hipStream_t)stream, S, rows, N, enc, view27, ray_idx, count, mlp_params, raw, act, 1, F32FwdBatch{});
    else launch_fwd<128>((hipStream_t)stream, S, rows, N, enc, view27, ray_idx, count, mlp_params, raw, act, 1, F32FwdBatch{});
    DURF_CHECK_LAUNCH("durf_mlp_fwd_f32");
    return 0;
}

int durf_mlp_bwd_f32(void* stream, int width, int in_dim, size_t rows, int N, const float* draw,
                     const int32_t* ray_idx, const int32_t* count, const float* mlp_params, const float* params_t,
                     const float* act, float* dz, float* d_enc) {
    DURF_REQUIRE(width == 256 || width == 128, "width must be 256 or 128");
    DURF_REQUIRE(in_dim > 0 && in_dim <= DURF_ENC_DIM, "in_dim <= 64");
    if (rows == 0) return 0;
    const F32Spec S = f32_spec(width, in_dim);
    if (width == 256) launch_bwd<256>((hipStream_t)stream, S, rows, N, draw, ray_idx, count, mlp_params, params_t, act, dz, d_enc, 1, F32BwdBatch{});
    else launch_bwd<128>((hipStream_t)stream, S, rows, N, draw, ray_idx, count, mlp_params, params_t, act, dz, d_enc, 1, F32BwdBatch{});
    DURF_CHECK_LAUNCH("durf_mlp_bwd_f32");
    return 0;
}

int durf_mlp_dw_f32(void* stream, int width, int in_dim, size_t rows, int N, const int32_t* count, const float* act,
                    const float* dz, int nsplit, float* scratch, float* grad_mlp) {
    DURF_REQUIRE(width == 256 || width == 128, "width must be 256 or 128");
    DURF_REQUIRE(nsplit >= 1 && nsplit <= 1024, "1 <= nsplit <= 1024");
    const F32Spec S = f32_spec(width, in_dim);
    const size_t params = durf_layer_offset(width, in_dim, 12, 0);
    const F32TileTab T = tile_table(S);
    F32DwArgs a{};
    a.nseg = 1;
    a.seg[0] = F32DwSeg{act, dz, count, rows, N};
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_mlp_dw_f32, dim3(T.base[12], nsplit, 1), dim3(256), 0, s, S, T, a, nsplit, params, scratch);
    hipLaunchKernelGGL(k_dw_f32_reduce, dim3(durf_cdiv(params, 256), 1), dim3(256), 0, s, params, nsplit, scratch, (size_t)0,
                       grad_mlp, (size_t)0, 0);
    DURF_CHECK_LAUNCH("durf_mlp_dw_f32");
    return 0;
}

int durf_bkgd_hit_rays_f32(void* stream, int B, const float* view27, const float* bkgd_params, const int32_t* idx,
                           const int32_t* count, float* trunk /* [257] scratch */, float* raw_tail) {
    if (B <= 0) return 0;
    const F32Spec S = f32_spec(DURF_W_BKGD, 60);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_bkgd_const_trunk, dim3(1), dim3(1024), 0, s, S, bkgd_params, trunk);
    hipLaunchKernelGGL(k_bkgd_hit_rays, dim3(durf_cdiv(B, HITRAYS_PER_WG)), dim3(128), 0, s, S, bkgd_params, trunk, view27, idx,
                       count, raw_tail);
    DURF_CHECK_LAUNCH("durf_bkgd_hit_rays_f32");
    return 0;
}

/* ---- the K object MLPs of a step on the fp32 kernels, one launch per phase (objects in blockIdx.y / .z) ---------- */
size_t durf_objf32_act_stride(int B, int N) { return tile_rows((size_t)B * N) * f32_spec(DURF_W_OBJ, 63).act; }
size_t durf_objf32_dz_stride(int B, int N) { return tile_rows((size_t)B * N) * f32_spec(DURF_W_OBJ, 63).dz; }

int durf_objf32_fwd_batch(void* stream, int K, int B, int N, const int32_t* idx, const int32_t* count, const float* enc,
                          const float* view27, const float* obj_params, size_t param_stride, float* raw, float* act) {
    DURF_REQUIRE(K >= 1 && K <= DURF_MAX_OBJ, "1 <= K <= DURF_MAX_OBJ");
    if (B <= 0) return 0;
    const size_t rows = (size_t)B * N;
    const F32Spec S = f32_spec(DURF_W_OBJ, 63);
    F32FwdBatch bs{rows * 63, (size_t)B, param_stride, rows * 4, durf_objf32_act_stride(B, N)};
    launch_fwd<DURF_W_OBJ>((hipStream_t)stream, S, rows, N, enc, view27, idx, count, obj_params, raw, act, K, bs);
    DURF_CHECK_LAUNCH("durf_objf32_fwd_batch");
    return 0;
}

int durf_objf32_bwd_batch(void* stream, int K, int B, int N, const int32_t* idx, const int32_t* count, const float* draw,
                          const float* obj_params, const float* obj_params_t, size_t param_stride, const float* act,
                          float* dz, float* d_enc) {
    DURF_REQUIRE(K >= 1 && K <= DURF_MAX_OBJ, "1 <= K <= DURF_MAX_OBJ");
    if (B <= 0) return 0;
    const size_t rows = (size_t)B * N;
    const F32Spec S = f32_spec(DURF_W_OBJ, 63);
    F32BwdBatch bs{(size_t)B, param_stride, durf_objf32_act_stride(B, N), durf_objf32_dz_stride(B, N), rows * DURF_ENC_DIM};
    launch_bwd<DURF_W_OBJ>((hipStream_t)stream, S, rows, N, draw, idx, count, obj_params, obj_params_t, act, dz, d_enc, K, bs);
    DURF_CHECK_LAUNCH("durf_objf32_bwd_batch");
    return 0;
}

int durf_objf32_dw_batch(void* stream, int K, int B, int N, const int32_t* count, int nlevels, const float* const* act,
                         const float* const* dz, int nsplit, float* scratch, float* grad_obj, size_t grad_stride) {
    DURF_REQUIRE(K >= 1 && K <= DURF_MAX_OBJ, "1 <= K <= DURF_MAX_OBJ");
    DURF_REQUIRE(nlevels >= 1 && nlevels <= F32_MAX_SEG, "1 <= nlevels <= 4");
    DURF_REQUIRE(nsplit >= 1 && nsplit <= 1024, "1 <= nsplit <= 1024");
    if (B <= 0) return 0;
    const F32Spec S = f32_spec(DURF_W_OBJ, 63);
    const size_t params = durf_layer_offset(DURF_W_OBJ, 63, 12, 0);
    const F32TileTab T = tile_table(S);
    F32DwArgs a{};
    a.nseg = nlevels;
    for (int l = 0; l < nlevels; l++) a.seg[l] = F32DwSeg{act[l], dz[l], count, (size_t)B * N, N};
    a.act_stride = durf_objf32_act_stride(B, N); a.dz_stride = durf_objf32_dz_stride(B, N);
    a.part_stride = (size_t)nsplit * params;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_mlp_dw_f32, dim3(T.base[12], nsplit, K), dim3(256), 0, s, S, T, a, nsplit, params, scratch);
    hipLaunchKernelGGL(k_dw_f32_reduce, dim3(durf_cdiv(params, 256), K), dim3(256), 0, s, params, nsplit, scratch,
                       a.part_stride, grad_obj, grad_stride, 0);
    DURF_CHECK_LAUNCH("durf_objf32_dw_batch");
    return 0;
}

}  // extern "C"
