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
// activations pass between layers through LDS as x[feature][sample], and the weights stream L2 -> registers -> LDS as
// CHUNKS of KC input rows from a packed stream (durf_mlp_f32_pack, once per step: the chunks of a whole pass in the
// order they are consumed, zero-padded to whole tiles, so a chunk is one contiguous, 16-byte aligned block and a
// load is `descriptor + lane offset + immediate`), double buffered in LDS, DEPTH chunks ahead in registers.  The
// backward walks a second stream of per-layer TRANSPOSED kernels.  The 1- / 3-wide heads are VALU dot products.
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
#include "gauss.h"

struct F32Layer { int fi, fo, x_off, dz_off, relu; size_t w_off; };
struct F32Spec {
    F32Layer L[12];
    int act, dz, W, in_dim;
};

// (constexpr: the fused kernels are instantiated for the two MLPs of the model and see every offset as an immediate)
__host__ __device__ constexpr F32Spec f32_spec(int W, int in_dim) {
    F32Spec s{};
    s.W = W; s.in_dim = in_dim;
    int x = 0, d = 0;
    size_t woff = 0;
    for (int l = 0; l < 12; l++) {
        const int fi = l == 0 ? in_dim : (l == 5 ? W + in_dim : (l == 10 ? W + 27 : (l == 11 ? 128 : W)));
        const int fo = l == 8 ? 1 : (l == 10 ? 128 : (l == 11 ? 3 : W));
        s.L[l].fi = fi; s.L[l].fo = fo;
        s.L[l].w_off = woff; woff += (size_t)fi * fo + fo;          // == durf_layer_offset(W, in_dim, l, 0)
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
    static constexpr int PFV = (KC * CMAX / 4 + NT - 1) / NT;      // prefetch registers (float4) per thread and chunk
    // constants fetched once per tile: biases of the 10 MFMA layers [10][W], Dense_8 kernel + bias [W + 1],
    // Dense_11 kernel + bias [128 * 3 + 3] (padded to 392)
    static constexpr int CONST_FLOATS = 10 * W + (W + 8) + 392;
    static constexpr int LDS_FLOATS = 2 * XROWS * F32_XS + 2 * 64 * F32_XS + 2 * KC * CMAX + 8 * 4 * 32 + 4 * 32 + CONST_FLOATS;
    static constexpr int DEPTH = W == 128 ? 4 : 2;       // register sets of the weight prefetch (F32Sched)
};

__device__ __forceinline__ int c_row(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

__device__ __forceinline__ size_t f32_rows(size_t rows, int N, const int32_t* count) {
    if (!count) return rows;
    const size_t c = (size_t)(*count) * (size_t)N;
    return c < rows ? c : rows;
}

// ---- static chunk schedule ----------------------------------------------------------------------------------------
// A pass over an MLP is a fixed sequence of weight chunks (KC input rows x C output columns of one Dense after the
// other), known at compile time.  Chunk g is consumed from LDS buffer g & 1; at step g the workgroup commits chunk
// g + 1 from its registers to the other buffer and issues the loads of chunk g + DEPTH into the register set that
// chunk g occupied, so a chunk's loads have DEPTH - 1 chunks of MFMA time (across layer boundaries) to arrive.
template <int W, bool BWD>
struct F32Sched {
    static constexpr int KC = F32Cfg<W>::KC;
    static constexpr int NL = 10;
    // forward: Dense_0..7, 9, 10 (the 1- and 3-wide heads are VALU work); backward: Dense_10, 9, 7, .. 1, 0
    __host__ __device__ static constexpr int layer(int i) {
        if (!BWD) return i <= 7 ? i : i + 1;
        return i == 0 ? 10 : (i == 1 ? 9 : 9 - i);
    }
    // output columns of the chunk matrix, padded to whole 32-wide tiles (backward: the inputs whose gradient is
    // propagated -- Dense_10: the W bottleneck-fed ones; Dense_5: trunk + encoding; Dense_0: the encoding)
    __host__ __device__ static constexpr int cols(int i) {
        const int l = layer(i);
        if (!BWD) return l == 10 ? 128 : W;
        return l == 5 ? W + 64 : (l == 0 ? 64 : W);
    }
    // reduction rows covered by the schedule (the stream is zero-padded beyond the actual count)
    __host__ __device__ static constexpr int rows(int i) {
        const int l = layer(i);
        if (BWD) return l == 10 ? 128 : W;
        return l == 0 ? 64 : (l == 5 ? W + 64 : (l == 10 ? W + 32 : W));
    }
    __host__ __device__ static constexpr int nchunk(int i) { return (rows(i) + KC - 1) / KC; }
    __host__ __device__ static constexpr int total() { int t = 0; for (int i = 0; i < NL; i++) t += nchunk(i); return t; }
    __host__ __device__ static constexpr int layer_of(int g) { int i = 0; while (g >= nchunk(i)) { g -= nchunk(i); i++; } return i; }
    __host__ __device__ static constexpr int chunk_of(int g) { int i = 0; while (g >= nchunk(i)) { g -= nchunk(i); i++; } return g; }
    // float offset of chunk g in the packed stream
    __host__ __device__ static constexpr size_t chunk_off(int g) {
        size_t o = 0;
        for (int h = 0; h < g; h++) o += (size_t)KC * cols(layer_of(h));
        return o;
    }
    __host__ __device__ static constexpr size_t stream_floats() { return chunk_off(total()); }
};

// packed streams of one MLP: [forward stream | backward stream]; element (chunk g, row rr, column c) of the forward
// stream = kernel_l[c_g KC + rr][c], of the backward stream = kernel_l[c][c_g KC + rr] (transposed), zero outside
// operand packing of the "bf16x3" variants (see chunk_mma_x3 below)
__device__ __forceinline__ float x3_pack(float x) {
    const __bf16 h = (__bf16)x;
    const __bf16 l = (__bf16)(x - (float)h);
    const unsigned w = ((unsigned)__builtin_bit_cast(unsigned short, h) << 16) | (unsigned)__builtin_bit_cast(unsigned short, l);
    return __builtin_bit_cast(float, w);
}
__device__ __forceinline__ float x3_unpack(float p) {
    const unsigned w = __builtin_bit_cast(unsigned, p);
    return __builtin_bit_cast(float, w & 0xffff0000u) + __builtin_bit_cast(float, w << 16);
}
template <bool X3> __device__ __forceinline__ float opk(float x) { if constexpr (X3) return x3_pack(x); else return x; }
template <bool X3> __device__ __forceinline__ float oup(float p) { if constexpr (X3) return x3_unpack(p); else return p; }

template <int W, int IN, bool X3 = false>
__global__ void __launch_bounds__(256)
k_f32_pack(const float* __restrict__ P, float* __restrict__ ws, size_t p_stride, size_t ws_stride) {
    constexpr F32Spec S = f32_spec(W, IN);
    using SF = F32Sched<W, false>;
    using SB = F32Sched<W, true>;
    constexpr size_t NF = SF::stream_floats(), NB = SB::stream_floats();
    P += blockIdx.y * p_stride; ws += blockIdx.y * ws_stride;
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= NF + NB) return;
    const bool bwd = e >= NF;
    size_t rem = bwd ? e - NF : e;
    float v = 0.0f;
    bool done = false;
    static_for<0, 10>([&](auto i_) {
        constexpr int i = decltype(i_)::value;
        if (done) return;
        // same decode for both streams, with their own tables
        const size_t sz = bwd ? (size_t)SB::nchunk(i) * SB::KC * SB::cols(i) : (size_t)SF::nchunk(i) * SF::KC * SF::cols(i);
        if (rem >= sz) { rem -= sz; return; }
        done = true;
        const int C = bwd ? SB::cols(i) : SF::cols(i);
        const int l = bwd ? SB::layer(i) : SF::layer(i);
        const int r = (int)(rem / C), c = (int)(rem % C);
        const int fi = S.L[l].fi, fo = S.L[l].fo;
        if (!bwd) { if (r < fi && c < fo) v = P[S.L[l].w_off + (size_t)r * fo + c]; }
        else {
            const int cv = l == 10 ? W : fi;                  // the view-fed inputs of Dense_10 are not propagated
            if (r < fo && c < cv) v = P[S.L[l].w_off + (size_t)c * fo + r];
        }
    });
    ws[e] = opk<X3>(v);          // (bf16x3: the stream holds (hi, lo) bf16 pairs in the words of the fp32 values)
}

// ---- weight chunks: stream -> registers (issue) -> LDS (commit), 16 bytes per lane ------------------------------
template <int C, int KC, int NT, int PFV>
__device__ __forceinline__ void chunk_issue(f32x4 (&pf)[PFV], __amdgpu_buffer_rsrc_t rs, unsigned byte_off, int tid) {
    constexpr int V4 = KC * C / 4, E = (V4 + NT - 1) / NT;
    static_assert(E <= PFV, "prefetch registers");
#pragma unroll
    for (int j = 0; j < E; j++) {
        const int idx = tid + NT * j;
        if (E * NT == V4 || idx < V4) pf[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, idx * 16, byte_off, 0));
    }
}
template <int C, int KC, int NT, int PFV>
__device__ __forceinline__ void chunk_commit(const f32x4 (&pf)[PFV], float* wb, int tid) {
    constexpr int V4 = KC * C / 4, E = (V4 + NT - 1) / NT;
#pragma unroll
    for (int j = 0; j < E; j++) {
        const int idx = tid + NT * j;
        if (E * NT == V4 || idx < V4) *(f32x4*)(wb + idx * 4) = pf[j];
    }
}

// acc[j] += chunk^T x for this wave's output tiles mo = wave + NW j  (A = chunk[k][32 mo + m], B = x[k][n]).
// The operand reads are issued from inline asm RING k-steps ahead of the MFMA that consumes them, with counted
// s_waitcnt lgkmcnt (LDS reads return in order): left to itself hipcc issues each pair of reads right before its
// MFMA and waits lgkmcnt(0), i.e. every 64-cycle MFMA also pays an LDS round trip.
template <int OFF>
__device__ __forceinline__ float lds_read4(unsigned addr) {
    float r;
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
    return r;
}
// the operands are valid once at most N younger LDS reads are outstanding; every register an MFMA after the wait reads
// is tied to it, so that hipcc cannot move that MFMA above the wait
template <int N>
__device__ __forceinline__ void lds_wait4(float& u, float& v) {
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(u), "+v"(v) : "n"(N));
}
template <int N>
__device__ __forceinline__ void lds_wait4(float& u, float& v, float& w) {
    asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(u), "+v"(v), "+v"(w) : "n"(N));
}
#define F32_MMA(a, b, c) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0)
template <int C, int KC, int NW, int TPW>
__device__ __forceinline__ void chunk_mma(const float* wb, const float* xr, int wave, int lane, f32x16 (&acc)[TPW]) {
    const int m = lane & 31, kk = lane >> 5;
    if (wave >= C / 32) return;
    constexpr int KS = KC / 2, RING = KS < 4 ? KS : 4;
    const unsigned xb_ = lds_addr_of((const char*)(xr + kk * F32_XS + m));
    const unsigned a0_ = lds_addr_of((const char*)(wb + kk * C + 32 * wave + m));
    const bool two = TPW > 1 && wave + NW < C / 32;
    if (two) {
        float b[RING], a0[RING], a1[RING];
        static_for<0, RING>([&](auto j_) {
            constexpr int j = decltype(j_)::value;
            b[j] = lds_read4<j * 2 * F32_XS * 4>(xb_);
            a0[j] = lds_read4<j * 2 * C * 4>(a0_);
            a1[j] = lds_read4<(j * 2 * C + 32 * NW) * 4>(a0_);
        });
        static_for<0, KS>([&](auto k_) {
            constexpr int ks = decltype(k_)::value, later = (KS - 1 - ks) < (RING - 1) ? (KS - 1 - ks) : (RING - 1);
            lds_wait4<3 * later>(b[ks % RING], a0[ks % RING], a1[ks % RING]);
            F32_MMA(a0[ks % RING], b[ks % RING], acc[0]);
            F32_MMA(a1[ks % RING], b[ks % RING], acc[TPW - 1]);
            if constexpr (ks + RING < KS) {
                b[ks % RING] = lds_read4<(ks + RING) * 2 * F32_XS * 4>(xb_);
                a0[ks % RING] = lds_read4<(ks + RING) * 2 * C * 4>(a0_);
                a1[ks % RING] = lds_read4<((ks + RING) * 2 * C + 32 * NW) * 4>(a0_);
            }
        });
    } else {
        float b[RING], a0[RING];
        static_for<0, RING>([&](auto j_) {
            constexpr int j = decltype(j_)::value;
            b[j] = lds_read4<j * 2 * F32_XS * 4>(xb_);
            a0[j] = lds_read4<j * 2 * C * 4>(a0_);
        });
        static_for<0, KS>([&](auto k_) {
            constexpr int ks = decltype(k_)::value, later = (KS - 1 - ks) < (RING - 1) ? (KS - 1 - ks) : (RING - 1);
            lds_wait4<2 * later>(b[ks % RING], a0[ks % RING]);
            F32_MMA(a0[ks % RING], b[ks % RING], acc[0]);
            if constexpr (ks + RING < KS) {
                b[ks % RING] = lds_read4<(ks + RING) * 2 * F32_XS * 4>(xb_);
                a0[ks % RING] = lds_read4<(ks + RING) * 2 * C * 4>(a0_);
            }
        });
    }
}

// ---- "bf16x3" (round 6): the same data flow on the bf16 matrix pipe ------------------------------------------------
// Every MFMA operand is kept as a PAIR of bf16 -- x = hi + lo, hi = bf16(x), lo = bf16(x - hi): 16-17 significant bits --
// packed in the 32-bit word the fp32 value would occupy (hi in the upper half), so streams, LDS layouts and the chunk
// schedule are unchanged; a product is three v_mfma_f32_32x32x16_bf16 (hi.hi + hi.lo + lo.hi; lo.lo is below 2^-17 of
// it) in place of eight v_mfma_f32_32x32x2_f32: 96 instead of 512 matrix-pipe cycles per 16 k values.  The packing is
// done where an operand is WRITTEN (the weight stream by k_f32_pack, the activations by the layer epilogues), so the
// inner loop only reads words and sorts halves (v_perm_b32).  fp32 accumulation, bias, ReLU, records and heads as before.
typedef unsigned u32x8_ __attribute__((ext_vector_type(8)));
typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
template <int OFF>
__device__ __forceinline__ unsigned lds_read_u32(unsigned addr) {
    unsigned r;
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
    return r;
}
// eight packed words (k = k0 .. k0 + 7 of one row / column) -> the hi and the lo MFMA fragment (8 bf16 each)
__device__ __forceinline__ void x3_frags(const unsigned (&w)[8], bf16x8& fh, bf16x8& fl) {
    u32x4_ h, l;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        h[j] = __builtin_amdgcn_perm(w[2 * j + 1], w[2 * j], 0x07060302u);
        l[j] = __builtin_amdgcn_perm(w[2 * j + 1], w[2 * j], 0x05040100u);
    }
    fh = __builtin_bit_cast(bf16x8, h);
    fl = __builtin_bit_cast(bf16x8, l);
}
// the eight words of a fragment are valid once at most N younger LDS reads are outstanding.  Every one of them is tied to the
// wait as a SCALAR register -- the asm reads must land in the very registers the v_perm below reads: gathered into a vector
// first, hipcc copies them (a v_mov right behind the ds_read, i.e. BEFORE the data has arrived: seen as run-to-run different
// losses in the first build of this routine)
template <int N>
__device__ __forceinline__ void x3_wait(unsigned (&w)[8]) {
    asm volatile("s_waitcnt lgkmcnt(%8)" : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]), "+v"(w[4]), "+v"(w[5]), "+v"(w[6]), "+v"(w[7]) : "n"(N));
}
#define X3_MMA(a, b, c) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)
// chunk_mma on packed operands: lane (m, kk) holds k = 16 s + 8 kk .. + 7 of A row / B column m.  The reads go out in GROUPS of
// eight (one fragment's words), one group ahead of the one being consumed: never more than eight in flight -- lgkmcnt is a
// 4-bit counter, and a first build that issued a whole chunk's 32-48 reads at once was not bit-reproducible in the backward --
// and every group is waited for (lgkmcnt(0)) right before its words are sorted into fragments, which the next group's reads
// and this group's perms / MFMAs overlap.
template <int C, int KC, int NW, int TPW>
__device__ __forceinline__ void chunk_mma_x3(const float* wb, const float* xr, int wave, int lane, f32x16 (&acc)[TPW]) {
    static_assert(KC % 16 == 0, "whole 16-k MFMA steps per chunk");
    const int m = lane & 31, kk = lane >> 5;
    if (wave >= C / 32) return;
    constexpr int NS = KC / 16;
    const unsigned xb_ = lds_addr_of((const char*)(xr + 8 * kk * F32_XS + m));
    const unsigned a0_ = lds_addr_of((const char*)(wb + 8 * kk * C + 32 * wave + m));
    const bool two = TPW > 1 && wave + NW < C / 32;
    // group q of a step: 0 = B (activations), 1 = A (this wave's tile), 2 = the second tile's A
    auto issue = [&](auto s_, auto q_, unsigned (&w)[8]) {
        constexpr int s = decltype(s_)::value, q = decltype(q_)::value;
        static_for<0, 8>([&](auto j_) {
            constexpr int j = decltype(j_)::value;
            if constexpr (q == 0) w[j] = lds_read_u32<(16 * s + j) * F32_XS * 4>(xb_);
            else if constexpr (q == 1) w[j] = lds_read_u32<(16 * s + j) * C * 4>(a0_);
            else w[j] = lds_read_u32<((16 * s + j) * C + 32 * NW) * 4>(a0_);
        });
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    unsigned g0[8], g1[8];                 // two word buffers; their roles swap every step (never copied: a copy of a register
                                           // whose read is still in flight reads the old value)
    if (two) {
        issue(I0{}, I0{}, g0);
        auto step = [&](auto s_, unsigned (&X)[8], unsigned (&Y)[8]) {        // on entry: B_s under way into X
            constexpr int s = decltype(s_)::value;
            bf16x8 bh, bl, ah, al;
            x3_wait<0>(X); issue(s_, I1{}, Y); x3_frags(X, bh, bl);                        // B_s ready; A_s under way
            x3_wait<0>(Y); issue(s_, I2{}, X); x3_frags(Y, ah, al);                        // A_s ready; the second tile's A_s under way
            X3_MMA(ah, bh, acc[0]);
            X3_MMA(ah, bl, acc[0]);
            X3_MMA(al, bh, acc[0]);
            x3_wait<0>(X);
            if constexpr (s + 1 < NS) issue(std::integral_constant<int, s + 1>{}, I0{}, Y);    // B_{s+1} under way into Y
            x3_frags(X, ah, al);
            X3_MMA(ah, bh, acc[TPW - 1]);
            X3_MMA(ah, bl, acc[TPW - 1]);
            X3_MMA(al, bh, acc[TPW - 1]);
        };
        static_for<0, NS>([&](auto s_) {
            if constexpr (decltype(s_)::value % 2 == 0) step(s_, g0, g1); else step(s_, g1, g0);
        });
    } else {
        issue(I0{}, I0{}, g0);
        static_for<0, NS>([&](auto s_) {            // B_s is always under way into g0 on entry
            constexpr int s = decltype(s_)::value;
            bf16x8 bh, bl, ah, al;
            x3_wait<0>(g0); issue(s_, I1{}, g1); x3_frags(g0, bh, bl);
            x3_wait<0>(g1);
            if constexpr (s + 1 < NS) issue(std::integral_constant<int, s + 1>{}, I0{}, g0);
            x3_frags(g1, ah, al);
            X3_MMA(ah, bh, acc[0]);
            X3_MMA(ah, bl, acc[0]);
            X3_MMA(al, bh, acc[0]);
        });
    }
}
template <bool X3, int C, int KC, int NW, int TPW>
__device__ __forceinline__ void chunk_mma_sel(const float* wb, const float* xr, int wave, int lane, f32x16 (&acc)[TPW]) {
    if constexpr (X3) chunk_mma_x3<C, KC, NW, TPW>(wb, xr, wave, lane, acc);
    else chunk_mma<C, KC, NW, TPW>(wb, xr, wave, lane, acc);
}

// Record accesses (tile-transposed act / dz blocks) through a buffer descriptor rebuilt per tile: element (float index f,
// sample n) sits at byte (f * 32 + n) * 4, so the 16 accesses of an epilogue are `descriptor + one lane offset + an
// immediate` with the record's offset in an SGPR -- as plain pointers they are 16 64-bit addresses per layer that hipcc
// keeps live across the whole unrolled pass (hundreds of spills).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t f32_rsrc(const float* p) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ void rec_store(__amdgpu_buffer_rsrc_t rs, int f, int n, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs, (unsigned)((f * 32 + n) * 4), 0, 0);
}
// lane part (wave tile, half, sample) in a register, (record offset + in-tile row) * 128 bytes as scalar / immediate
template <int F0>
__device__ __forceinline__ void rec_store_t(__amdgpu_buffer_rsrc_t rs, unsigned lane_off, int r, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs, lane_off + (unsigned)(((r & 3) + 8 * (r >> 2)) * 128),
                                          F0 * 128, 0);
}
template <int F0>
__device__ __forceinline__ float rec_load_t(__amdgpu_buffer_rsrc_t rs, unsigned lane_off, int r) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, lane_off + (unsigned)(((r & 3) + 8 * (r >> 2)) * 128),
                                                                         F0 * 128, 0));
}

struct F32FwdBatch { size_t enc, idx, params, ws, raw, act; };        // per-object strides (floats; idx: int32 elements)
// the object forward can encode its own tiles (accurate-libm object IPE of rays.hip's k_encode<true>, bit-identical): the
// inputs of durf_encode_obj
struct F32Enc { const float* t_vals; const float* origins_s; const float* dirs_s; const float* radii; BarfW w; int flags; };
struct F32BwdBatch { size_t idx, params, ws, act, dz, d_enc; };

// ---------------------------------------------------------------------------------------------
// forward: obbpose_model.py:305-354 / :369-418 in fp32
//   enc == nullptr: every row is evaluated on the constant encoding of a zero-masked Gaussian ([0 x 30, 1 x 30]): the
//   background MLP's single evaluation of a box-hit ray (obbpose_model.py:205-210; include/durf_hip.h durf_expand_raw)
// ---------------------------------------------------------------------------------------------
template <int W, int IN, bool TRAIN, bool ENC, bool X3 = false>
__global__ void __launch_bounds__(W * 2)
k_mlp_fwd_f32(size_t rows, int N, const float* __restrict__ enc, const float* __restrict__ view,
              const int32_t* __restrict__ ray_idx, const int32_t* __restrict__ count,
              const float* __restrict__ P, const float* __restrict__ ws, float* __restrict__ raw, float* __restrict__ act,
              F32FwdBatch bs, F32Enc ei) {
    using Cf = F32Cfg<W>;
    using Sc = F32Sched<W, false>;
    constexpr F32Spec S = f32_spec(W, IN);
    constexpr int NT = Cf::NT, XR = Cf::XROWS;
    constexpr int D = Cf::DEPTH, KC = Cf::KC, PFV = Cf::PFV, CB = KC * Cf::CMAX, TOTAL = Sc::total();
    constexpr int in_dim = IN;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* xa = lds;
    float* xb = xa + XR * F32_XS;
    float* encs = xb + XR * F32_XS;                     // the block's encoding, kept for the skip concatenation
    float* vws = encs + 64 * F32_XS;                    // the block's view directions [27 (32)][32]
    float* wbuf = encs + 2 * 64 * F32_XS;
    float* red = wbuf + 2 * Cf::KC * Cf::CMAX;          // [8 parts][4 outputs][32 samples] head partial sums
    float* cst = red + 8 * 4 * 32 + 4 * 32;             // biases [10][W] | Dense_8 [W + 8] | Dense_11 [392]
    float* w8s = cst + 10 * W;
    float* w11s = w8s + W + 8;
    if (gridDim.y > 1) {
        const size_t k = blockIdx.y;
        if (enc) enc += k * bs.enc;
        ray_idx += k * bs.idx; count += k; P += k * bs.params; ws += k * bs.ws; raw += k * bs.raw;
        if (TRAIN) act += k * bs.act;
    }
    const size_t nrows = f32_rows(rows, N, count);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 31, hi = lane >> 5;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)ws, 0, 0x7fffffff, 0x00020000);
    // constants of the MLP: biases (the initial accumulators) and the head kernels, once per workgroup
    static_for<0, 10>([&](auto i_) {
        constexpr int i = decltype(i_)::value, l = i <= 7 ? i : i + 1;
        for (int o = tid; o < W; o += NT) cst[i * W + o] = o < S.L[l].fo ? P[S.L[l].w_off + (size_t)S.L[l].fi * S.L[l].fo + o] : 0.0f;
    });
    for (int idx = tid; idx < W + 1; idx += NT) w8s[idx] = P[S.L[8].w_off + idx];
    for (int idx = tid; idx < 128 * 3 + 3; idx += NT) w11s[idx] = P[S.L[11].w_off + idx];
    // The grid is capped (the row count lives on the device: a grid sized for the capacity would be thousands of
    // workgroups that only exit, and an early-exit workgroup of a 120 KB-LDS kernel still costs its dispatch: measured
    // 48 us for 12 288 of them at cfg4); a workgroup walks its 32-row tiles.
    for (size_t tile = blockIdx.x; tile * 32 < nrows; tile += gridDim.x) {
        const size_t row0 = tile * 32;
        const __amdgpu_buffer_rsrc_t ra = f32_rsrc(TRAIN ? act + tile * S.act * 32 : nullptr);   // this tile's record block
        const unsigned lane_off = (unsigned)(((32 * wave + 4 * hi) * 32 + n) * 4);
        __syncthreads();                                                       // the previous tile is done with the LDS
        f32x4 pf[D][PFV];
        static_for<0, D>([&](auto g_) {
            constexpr int g = decltype(g_)::value;
            if constexpr (g < TOTAL) chunk_issue<Sc::cols(Sc::layer_of(g)), KC, NT, PFV>(pf[g % D], rs, (unsigned)(Sc::chunk_off(g) * 4), tid);
        });
        // the samples' view directions and encodings (x0 = enc, rows in_dim..63 zero)
        for (int idx = tid; idx < 32 * 32; idx += NT) {
            const int f = idx & 31, nn = idx >> 5;
            float v = 0.0f;
            if (f < 27 && row0 + nn < nrows) {
                size_t ray = (row0 + nn) / (size_t)N;
                if (ray_idx) ray = (size_t)ray_idx[ray];
                v = view[ray * 27 + f];
            }
            vws[f * F32_XS + nn] = opk<X3>(v);
        }
        if constexpr (ENC) {
            // thread = (sample tid >> 3, 8-feature vector tid & 7): the tile's 32 x 64 encoding in one pass of 256 threads
            static_assert(!ENC || NT == 256, "in-kernel encoding is the object MLP's (W = 128)");
            const int nn = tid >> 3, q = tid & 7;
            float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (row0 + nn < nrows) {
                const size_t row = row0 + nn;
                const int b = ray_idx[row / (size_t)N], n_ = (int)(row % (size_t)N);
                const float t0 = ei.t_vals[(size_t)b * (N + 1) + n_], t1 = ei.t_vals[(size_t)b * (N + 1) + n_ + 1];
                const float o[3] = {ei.origins_s[b * 3], ei.origins_s[b * 3 + 1], ei.origins_s[b * 3 + 2]};
                const float d[3] = {ei.dirs_s[b * 3], ei.dirs_s[b * 3 + 1], ei.dirs_s[b * 3 + 2]};
                Gauss g = frustum_gaussian(t0, t1, o, d, ei.radii[b], (ei.flags & DURF_ENC_CYLINDER) != 0);
                if (ei.flags & DURF_ENC_NO_INTEGRATION) g.var[0] = g.var[1] = g.var[2] = 0.0f;      // obbpose_model.py:164-165
                obj_features8(g, ei.w, q, v);
            }
#pragma unroll
            for (int e = 0; e < 8; e++) { encs[(8 * q + e) * F32_XS + nn] = opk<X3>(v[e]); xa[(8 * q + e) * F32_XS + nn] = opk<X3>(v[e]); }
        } else {
            for (int idx = tid; idx < 64 * 32; idx += NT) {
                const int f = idx & 63, nn = idx >> 6;
                float v = 0.0f;
                if (f < in_dim && row0 + nn < nrows) v = enc ? enc[(row0 + nn) * (size_t)in_dim + f] : ((f >= 30 && f < 60) ? 1.0f : 0.0f);
                encs[f * F32_XS + nn] = opk<X3>(v);
                xa[f * F32_XS + nn] = opk<X3>(v);
            }
        }
        chunk_commit<Sc::cols(0), KC, NT, PFV>(pf[0], wbuf, tid);
        __syncthreads();
        if (TRAIN)
            for (int idx = tid; idx < in_dim * 32; idx += NT) rec_store(ra, S.L[0].x_off + (idx >> 5), idx & 31, oup<X3>(encs[(idx >> 5) * F32_XS + (idx & 31)]));

        f32x16 acc[1];
        float dens3[3] = {0.f, 0.f, 0.f}, rgb[3] = {0.f, 0.f, 0.f};
        // 1- / 3-wide heads: out[c][n] = b[c] + sum_k Wl[k][c] x[k][n] as 8 interleaved partial chains, summed in order
        auto head = [&](const float* Wl, const float* x, int fi, int fo, float (&out)[3]) {
            const int part = tid >> 5, nn = tid & 31;
            if (part < 8) {
                float sm[3] = {0.0f, 0.0f, 0.0f};
                for (int k = part; k < fi; k += 8) {
                    const float xv = oup<X3>(x[k * F32_XS + nn]);
                    for (int c = 0; c < fo; c++) sm[c] = fmaf(Wl[k * fo + c], xv, sm[c]);
                }
                for (int c = 0; c < fo; c++) red[(part * 4 + c) * 32 + nn] = sm[c];
            }
            __syncthreads();
            if (tid < 32)
                for (int c = 0; c < fo; c++) {
                    float sm = Wl[fi * fo + c];
                    for (int q = 0; q < 8; q++) sm += red[(q * 4 + c) * 32 + nn];
                    out[c] = sm;
                }
            __syncthreads();
        };
        if constexpr (W == 128) {
            static_for<0, TOTAL>([&](auto g_) {
                constexpr int g = decltype(g_)::value;
                constexpr int i = Sc::layer_of(g), c = Sc::chunk_of(g), l = Sc::layer(i), C = Sc::cols(i);
                const float* xin = (i & 1) ? xb : xa;
                float* xout = (i & 1) ? xa : xb;
                if constexpr (g + 1 < TOTAL) chunk_commit<Sc::cols(Sc::layer_of(g + 1)), KC, NT, PFV>(pf[(g + 1) % D], wbuf + ((g + 1) & 1) * CB, tid);
                if constexpr (g + D < TOTAL) chunk_issue<Sc::cols(Sc::layer_of(g + D)), KC, NT, PFV>(pf[(g + D) % D], rs, (unsigned)(Sc::chunk_off(g + D) * 4), tid);
                if constexpr (c == 0) {                                    // the bias is the initial accumulator
    #pragma unroll
                    for (int r = 0; r < 16; r++) acc[0][r] = cst[i * W + ((32 * wave + c_row(r, hi)) & (W - 1))];
                }
                chunk_mma_sel<X3, C, KC, Cf::NW, 1>(wbuf + (g & 1) * CB, xin + c * KC * F32_XS, wave, lane, acc);
                if constexpr (c == Sc::nchunk(i) - 1) {
                    // this wave's output tile -> the next Dense's input (LDS) and its record (global)
                    constexpr int lrec = l == 7 ? 8 : (l == 9 ? 10 : l + 1);    // h7 is x8 (density head and bottleneck); bottleneck -> x10
                    if (wave < C / 32) {
    #pragma unroll
                        for (int r = 0; r < 16; r++) {
                            const int o = 32 * wave + c_row(r, hi);
                            float v = acc[0][r];
                            if (l != 9) v = (v != v) ? v : fmaxf(v, 0.0f);     // jnp.maximum(x, 0) propagates NaN; fmaxf would drop it
                            xout[o * F32_XS + n] = opk<X3>(v);
                            if (TRAIN) rec_store_t<S.L[lrec].x_off>(ra, lane_off, r, v);
                        }
                    }
                    if constexpr (l == 4) {                                 // x5 = [h4, enc]   (obbpose_model.py:333-334)
                        for (int idx = tid; idx < 64 * 32; idx += NT) {
                            const int f = idx >> 5, nn = idx & 31;
                            const float v = encs[f * F32_XS + nn];
                            xout[(W + f) * F32_XS + nn] = v;
                            if (TRAIN && f < in_dim) rec_store(ra, S.L[5].x_off + W + f, nn, oup<X3>(v));
                        }
                    }
                    if constexpr (l == 9) {                                 // x10 = [bottleneck, view]   (:339, :346-347)
                        for (int idx = tid; idx < 32 * 32; idx += NT) {
                            const int f = idx >> 5, nn = idx & 31;
                            const float v = vws[f * F32_XS + nn];
                            xout[(W + f) * F32_XS + nn] = v;
                            if (TRAIN && f < 27) rec_store(ra, S.L[10].x_off + W + f, nn, oup<X3>(v));
                        }
                    }
                }
                __syncthreads();
                if constexpr (c == Sc::nchunk(i) - 1 && l == 7) head(w8s, xout, W, 1, dens3);           // density head on h7
                if constexpr (c == Sc::nchunk(i) - 1 && l == 10) head(w11s, xout, 128, 3, rgb);         // rgb head on hc
            });
        } else {
            // W = 256: the same schedule with the chunks of a layer walked by a RUNTIME loop (pairs of chunks: the LDS
            // buffer and the prefetch register set alternate), the last pair of every layer peeled for the hand-over to
            // the next layer.  Fully unrolled (as for W = 128, whose 35-50 KB fit) the pass is 154 chunk bodies = 170 KB
            // of straight-line code, three times the 64 KB instruction cache two CUs share: the kernel ran at 1.4 TFLOP/s,
            // 1 % of the fp32 MFMA rate, waiting for instructions (tools/time_f32_bkgd.py).
            static_assert(W == 128 || D == 2, "the pairwise walk assumes two prefetch register sets");
            static_for<0, Sc::NL>([&](auto i_) {
                constexpr int i = decltype(i_)::value, l = Sc::layer(i), C = Sc::cols(i), NCH = Sc::nchunk(i);
                constexpr bool has_next = i + 1 < Sc::NL;
                constexpr int CN = Sc::cols(has_next ? i + 1 : i);                   // the next layer's chunk width
                constexpr int G0 = TOTAL - [] { int t = 0; for (int j = i; j < Sc::NL; j++) t += Sc::nchunk(j); return t; }();
                static_assert(NCH % 2 == 0 && G0 % 2 == 0, "whole pairs of chunks per layer");
                constexpr unsigned OFF0 = (unsigned)(Sc::chunk_off(G0) * 4), OFFN = (unsigned)(Sc::chunk_off(G0 + NCH) * 4);
                const float* xin = (i & 1) ? xb : xa;
                float* xout = (i & 1) ? xa : xb;
#pragma unroll
                for (int r = 0; r < 16; r++) acc[0][r] = cst[i * W + ((32 * wave + c_row(r, hi)) & (W - 1))];     // bias
                // chunk c (parity PAR) of this layer: commit chunk c + 1, issue chunk c + 2, multiply
                auto same_layer_step = [&](auto par_, int c) {
                    constexpr int PAR = decltype(par_)::value;
                    chunk_commit<C, KC, NT, PFV>(pf[PAR ^ 1], wbuf + (PAR ^ 1) * CB, tid);
                    chunk_issue<C, KC, NT, PFV>(pf[PAR], rs, OFF0 + (unsigned)((c + 2) * KC * C * 4), tid);
                    chunk_mma_sel<X3, C, KC, Cf::NW, 1>(wbuf + PAR * CB, xin + c * KC * F32_XS, wave, lane, acc);
                    __syncthreads();
                };
#pragma unroll 1
                for (int c = 0; c < NCH - 2; c += 2) {
                    same_layer_step(std::integral_constant<int, 0>{}, c);
                    same_layer_step(std::integral_constant<int, 1>{}, c + 1);
                }
                // chunk NCH - 2: commit NCH - 1, issue the next layer's chunk 0
                chunk_commit<C, KC, NT, PFV>(pf[1], wbuf + CB, tid);
                if constexpr (has_next) chunk_issue<CN, KC, NT, PFV>(pf[0], rs, OFFN, tid);
                chunk_mma_sel<X3, C, KC, Cf::NW, 1>(wbuf, xin + (NCH - 2) * KC * F32_XS, wave, lane, acc);
                __syncthreads();
                // chunk NCH - 1: commit the next layer's chunk 0, issue its chunk 1, multiply, hand the tile over
                if constexpr (has_next) {
                    chunk_commit<CN, KC, NT, PFV>(pf[0], wbuf, tid);
                    chunk_issue<CN, KC, NT, PFV>(pf[1], rs, OFFN + (unsigned)(KC * CN * 4), tid);
                }
                chunk_mma_sel<X3, C, KC, Cf::NW, 1>(wbuf + CB, xin + (NCH - 1) * KC * F32_XS, wave, lane, acc);
                {
                    // this wave's output tile -> the next Dense's input (LDS) and its record (global)
                    constexpr int lrec = l == 7 ? 8 : (l == 9 ? 10 : l + 1);    // h7 is x8 (density head and bottleneck); bottleneck -> x10
                    if (wave < C / 32) {
    #pragma unroll
                        for (int r = 0; r < 16; r++) {
                            const int o = 32 * wave + c_row(r, hi);
                            float v = acc[0][r];
                            if (l != 9) v = (v != v) ? v : fmaxf(v, 0.0f);     // jnp.maximum(x, 0) propagates NaN; fmaxf would drop it
                            xout[o * F32_XS + n] = opk<X3>(v);
                            if (TRAIN) rec_store_t<S.L[lrec].x_off>(ra, lane_off, r, v);
                        }
                    }
                    if constexpr (l == 4) {                                 // x5 = [h4, enc]   (obbpose_model.py:333-334)
                        for (int idx = tid; idx < 64 * 32; idx += NT) {
                            const int f = idx >> 5, nn = idx & 31;
                            const float v = encs[f * F32_XS + nn];
                            xout[(W + f) * F32_XS + nn] = v;
                            if (TRAIN && f < in_dim) rec_store(ra, S.L[5].x_off + W + f, nn, oup<X3>(v));
                        }
                    }
                    if constexpr (l == 9) {                                 // x10 = [bottleneck, view]   (:339, :346-347)
                        for (int idx = tid; idx < 32 * 32; idx += NT) {
                            const int f = idx >> 5, nn = idx & 31;
                            const float v = vws[f * F32_XS + nn];
                            xout[(W + f) * F32_XS + nn] = v;
                            if (TRAIN && f < 27) rec_store(ra, S.L[10].x_off + W + f, nn, oup<X3>(v));
                        }
                    }
            
                }
                __syncthreads();
                if constexpr (l == 7) head(w8s, xout, W, 1, dens3);           // density head on h7
                if constexpr (l == 10) head(w11s, xout, 128, 3, rgb);         // rgb head on hc
            });
        }
        if (tid < 32 && row0 + tid < nrows) {
            const f32x4 o = {rgb[0], rgb[1], rgb[2], dens3[0]};
            *(f32x4*)(raw + (row0 + tid) * 4) = o;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// backward (data path): d(loss)/d(pre-activation) of every Dense, d(loss)/d(enc) (stored when d_enc != nullptr)
// ---------------------------------------------------------------------------------------------
template <int W, int IN, bool X3 = false>
__global__ void __launch_bounds__(W * 2)
k_mlp_bwd_f32(size_t rows, int N, const float* __restrict__ draw, const int32_t* __restrict__ ray_idx,
              const int32_t* __restrict__ count, const float* __restrict__ P, const float* __restrict__ ws,
              const float* __restrict__ act, float* __restrict__ dz, float* __restrict__ d_enc, F32BwdBatch bs) {
    using Cf = F32Cfg<W>;
    using Sc = F32Sched<W, true>;
    constexpr F32Spec S = f32_spec(W, IN);
    constexpr int NT = Cf::NT, XR = Cf::XROWS;
    constexpr int D = Cf::DEPTH, KC = Cf::KC, PFV = Cf::PFV, CB = KC * Cf::CMAX, TOTAL = Sc::total();
    constexpr int in_dim = IN;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* xa = lds;
    float* xb = xa + XR * F32_XS;
    float* denc = xb + XR * F32_XS;                     // [64][XS] d(enc), accumulated over Dense_5 and Dense_0
    float* wbuf = denc + 2 * 64 * F32_XS;
    float* gsm = wbuf + 2 * Cf::KC * Cf::CMAX + 8 * 4 * 32;     // [4][32] head gradients of the block's samples
    float* w8s = gsm + 4 * 32 + 10 * W;                          // Dense_8 / Dense_11 kernels (same slots as the forward's)
    float* w11s = w8s + W + 8;
    if (gridDim.y > 1) {
        const size_t k = blockIdx.y;
        ray_idx += k * bs.idx; count += k; P += k * bs.params; ws += k * bs.ws; act += k * bs.act; dz += k * bs.dz;
        if (d_enc) d_enc += k * bs.d_enc;
    }
    ws += F32Sched<W, false>::stream_floats();                   // the backward stream follows the forward one
    const size_t nrows = f32_rows(rows, N, count);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 31, hi = lane >> 5;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)ws, 0, 0x7fffffff, 0x00020000);
    for (int idx = tid; idx < W; idx += NT) w8s[idx] = P[S.L[8].w_off + idx];
    for (int idx = tid; idx < 128 * 3; idx += NT) w11s[idx] = P[S.L[11].w_off + idx];
    for (size_t tile = blockIdx.x; tile * 32 < nrows; tile += gridDim.x) {      // capped grid, see the forward
        const size_t row0 = tile * 32;
        const __amdgpu_buffer_rsrc_t ra = f32_rsrc(act + tile * S.act * 32), rz = f32_rsrc(dz + tile * S.dz * 32);
        const unsigned lane_off = (unsigned)(((32 * wave + 4 * hi) * 32 + n) * 4);
        __syncthreads();                                                       // the previous tile is done with the LDS
        f32x4 pf[D][PFV];
        static_for<0, D>([&](auto g_) {
            constexpr int g = decltype(g_)::value;
            if constexpr (g < TOTAL) chunk_issue<Sc::cols(Sc::layer_of(g)), KC, NT, PFV>(pf[g % D], rs, (unsigned)(Sc::chunk_off(g) * 4), tid);
        });
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
            rec_store(rz, S.L[11].dz_off + 0, tid, g[0]); rec_store(rz, S.L[11].dz_off + 1, tid, g[1]);
            rec_store(rz, S.L[11].dz_off + 2, tid, g[2]); rec_store(rz, S.L[8].dz_off, tid, g[3]);
        }
        __syncthreads();
        // Dense_11 (rgb head, 128 -> 3): d hc = W11 dz11, masked by Dense_10's ReLU -> dz10
        for (int idx = tid; idx < 128 * 32; idx += NT) {
            const int k = idx >> 5, nn = idx & 31;
            float v = 0.0f;
#pragma unroll
            for (int c = 0; c < 3; c++) v = fmaf(w11s[k * 3 + c], gsm[c * 32 + nn], v);
            const float h = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ra, (unsigned)(((S.L[11].x_off + k) * 32 + nn) * 4), 0, 0));
            v = h > 0.0f ? v : 0.0f;
            xa[k * F32_XS + nn] = opk<X3>(v);
            rec_store(rz, S.L[10].dz_off + k, nn, v);
        }
        chunk_commit<Sc::cols(0), KC, NT, PFV>(pf[0], wbuf, tid);
        __syncthreads();
        f32x16 acc[2];
        float hm[16];                      // the producer's ReLU outputs at this lane's 16 positions: fetched at the first
                                           // chunk of a layer, applied in its epilogue (a round trip hidden behind the MFMAs)
        if constexpr (W == 128) {
            static_for<0, TOTAL>([&](auto g_) {
                constexpr int g = decltype(g_)::value;
                constexpr int i = Sc::layer_of(g), c = Sc::chunk_of(g), l = Sc::layer(i), C = Sc::cols(i);
                const float* xin = (i & 1) ? xb : xa;
                float* xout = (i & 1) ? xa : xb;
                if constexpr (g + 1 < TOTAL) chunk_commit<Sc::cols(Sc::layer_of(g + 1)), KC, NT, PFV>(pf[(g + 1) % D], wbuf + ((g + 1) & 1) * CB, tid);
                if constexpr (g + D < TOTAL) chunk_issue<Sc::cols(Sc::layer_of(g + D)), KC, NT, PFV>(pf[(g + D) % D], rs, (unsigned)(Sc::chunk_off(g + D) * 4), tid);
                if constexpr (c == 0) {
    #pragma unroll
                    for (int r = 0; r < 16; r++) { acc[0][r] = 0.0f; acc[1][r] = 0.0f; }
                    if constexpr (l != 0 && l != 10) {
                        constexpr int h_off = S.L[l == 9 ? 8 : l].x_off;
    #pragma unroll
                        for (int r = 0; r < 16; r++) hm[r] = rec_load_t<h_off>(ra, lane_off, r);
                    }
                }
                chunk_mma_sel<X3, C, KC, Cf::NW, 2>(wbuf + (g & 1) * CB, xin + c * KC * F32_XS, wave, lane, acc);
                if constexpr (c == Sc::nchunk(i) - 1) {
                    if constexpr (l != 0) {
                        // d x_l (its first W features) -> dz of the Dense that produced them, masked by that Dense's ReLU output
                        // (the record x_l itself; the bottleneck, Dense_9, is linear; h7 = x8 also feeds the density head)
                        constexpr int lp = l == 10 ? 9 : (l == 9 ? 7 : l - 1);                  // the producer
                        constexpr bool relu = l != 10;
    #pragma unroll
                        for (int r = 0; r < 16; r++) {
                            const int k = 32 * wave + c_row(r, hi);
                            float v = acc[0][r];
                            if (l == 9) v += w8s[k] * gsm[3 * 32 + n];
                            if (relu) v = hm[r] > 0.0f ? v : 0.0f;
                            xout[k * F32_XS + n] = opk<X3>(v);
                            rec_store_t<S.L[lp].dz_off>(rz, lane_off, r, v);
                        }
                        if constexpr (l == 5) {            // skip connection: output tiles W/32 and W/32 + 1 are d enc
                            if (wave < 2) {
    #pragma unroll
                                for (int r = 0; r < 16; r++) denc[(32 * wave + c_row(r, hi)) * F32_XS + n] = acc[1][r];
                            }
                        }
                    } else {                               // Dense_0: d enc += W0 dz0
                        if (wave < 2) {
    #pragma unroll
                            for (int r = 0; r < 16; r++) denc[(32 * wave + c_row(r, hi)) * F32_XS + n] += acc[0][r];
                        }
                    }
                }
                __syncthreads();
            });
        } else {
            // W = 256: a runtime loop over the chunk pairs of every layer, the last pair peeled (see k_mlp_fwd_f32: the fully
            // unrolled pass does not fit the instruction cache)
            static_assert(W == 128 || D == 2, "the pairwise walk assumes two prefetch register sets");
            static_for<0, Sc::NL>([&](auto i_) {
                constexpr int i = decltype(i_)::value, l = Sc::layer(i), C = Sc::cols(i), NCH = Sc::nchunk(i);
                constexpr bool has_next = i + 1 < Sc::NL;
                constexpr int CN = Sc::cols(has_next ? i + 1 : i);
                constexpr int G0 = TOTAL - [] { int t = 0; for (int j = i; j < Sc::NL; j++) t += Sc::nchunk(j); return t; }();
                static_assert(NCH % 2 == 0 && G0 % 2 == 0, "whole pairs of chunks per layer");
                constexpr unsigned OFF0 = (unsigned)(Sc::chunk_off(G0) * 4), OFFN = (unsigned)(Sc::chunk_off(G0 + NCH) * 4);
                const float* xin = (i & 1) ? xb : xa;
                float* xout = (i & 1) ? xa : xb;
#pragma unroll
                for (int r = 0; r < 16; r++) { acc[0][r] = 0.0f; acc[1][r] = 0.0f; }
                if constexpr (l != 0 && l != 10) {
                    constexpr int h_off = S.L[l == 9 ? 8 : l].x_off;
#pragma unroll
                    for (int r = 0; r < 16; r++) hm[r] = rec_load_t<h_off>(ra, lane_off, r);
                }
            
                auto same_layer_step = [&](auto par_, int c) {
                    constexpr int PAR = decltype(par_)::value;
                    chunk_commit<C, KC, NT, PFV>(pf[PAR ^ 1], wbuf + (PAR ^ 1) * CB, tid);
                    chunk_issue<C, KC, NT, PFV>(pf[PAR], rs, OFF0 + (unsigned)((c + 2) * KC * C * 4), tid);
                    chunk_mma_sel<X3, C, KC, Cf::NW, 2>(wbuf + PAR * CB, xin + c * KC * F32_XS, wave, lane, acc);
                    __syncthreads();
                };
#pragma unroll 1
                for (int c = 0; c < NCH - 2; c += 2) {
                    same_layer_step(std::integral_constant<int, 0>{}, c);
                    same_layer_step(std::integral_constant<int, 1>{}, c + 1);
                }
                chunk_commit<C, KC, NT, PFV>(pf[1], wbuf + CB, tid);
                if constexpr (has_next) chunk_issue<CN, KC, NT, PFV>(pf[0], rs, OFFN, tid);
                chunk_mma_sel<X3, C, KC, Cf::NW, 2>(wbuf, xin + (NCH - 2) * KC * F32_XS, wave, lane, acc);
                __syncthreads();
                if constexpr (has_next) {
                    chunk_commit<CN, KC, NT, PFV>(pf[0], wbuf, tid);
                    chunk_issue<CN, KC, NT, PFV>(pf[1], rs, OFFN + (unsigned)(KC * CN * 4), tid);
                }
                chunk_mma_sel<X3, C, KC, Cf::NW, 2>(wbuf + CB, xin + (NCH - 1) * KC * F32_XS, wave, lane, acc);
                {
                    if constexpr (l != 0) {
                        // d x_l (its first W features) -> dz of the Dense that produced them, masked by that Dense's ReLU output
                        // (the record x_l itself; the bottleneck, Dense_9, is linear; h7 = x8 also feeds the density head)
                        constexpr int lp = l == 10 ? 9 : (l == 9 ? 7 : l - 1);                  // the producer
                        constexpr bool relu = l != 10;
    #pragma unroll
                        for (int r = 0; r < 16; r++) {
                            const int k = 32 * wave + c_row(r, hi);
                            float v = acc[0][r];
                            if (l == 9) v += w8s[k] * gsm[3 * 32 + n];
                            if (relu) v = hm[r] > 0.0f ? v : 0.0f;
                            xout[k * F32_XS + n] = opk<X3>(v);
                            rec_store_t<S.L[lp].dz_off>(rz, lane_off, r, v);
                        }
                        if constexpr (l == 5) {            // skip connection: output tiles W/32 and W/32 + 1 are d enc
                            if (wave < 2) {
    #pragma unroll
                                for (int r = 0; r < 16; r++) denc[(32 * wave + c_row(r, hi)) * F32_XS + n] = acc[1][r];
                            }
                        }
                    } else {                               // Dense_0: d enc += W0 dz0
                        if (wave < 2) {
    #pragma unroll
                            for (int r = 0; r < 16; r++) denc[(32 * wave + c_row(r, hi)) * F32_XS + n] += acc[0][r];
                        }
                    }
            
                }
                __syncthreads();
            });
        }
        if (d_enc) {
            for (int idx = tid; idx < 32 * 64; idx += NT) {
                const int nn = idx >> 6, k = idx & 63;
                if (row0 + nn < nrows) d_enc[(row0 + nn) * 64 + k] = k < in_dim ? denc[k * F32_XS + nn] : 0.0f;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// weight gradients: dW_l[k, m] = sum_n x_l[n, k] dz_l[n, m]; db_l[m] = sum_n dz_l[n, m] (VALU column sums, carried by
// the first k tile of every column tile: a bias row appended to x_l would cost a whole extra k tile of MFMAs per layer,
// +25 % on a 128-wide Dense).  One workgroup (4 waves)
// per (32 x 32 output tile, sample split, object); a wave walks its share of the 32-sample tiles -- the A / B
// operands of a tile's 16 MFMAs are 4 + 4 float4 per lane straight from the tile-transposed records (a lane's 16
// samples are the contiguous half [16 kk, 16 kk + 16) of a 128-byte line) -- then the four waves' accumulators are
// summed in order through LDS.  Partials are summed over the splits in a fixed order by k_dw_f32_reduce.
// (Measured at cfg4, 85 us for 33 us of MFMA work: the records are re-read by every column / k tile -- 630 MB per launch for
// 160 MB of records.  Three other layouts were built and measured slower: a wave owning a column tile and all k tiles of a
// strip straight from memory (150-230 us), the same through cooperative LDS staging (130-320 us) -- both need the operand
// reads ahead of the MFMAs the way chunk_mma issues them, hipcc serialises each MFMA behind its LDS read; with such an
// inline-asm operand ring and zero-filled staging the strip still took 130 us (fewer, longer workgroups: 13 units x
// splits x objects against 172 tiles) -- and this layout with coalesced loads staged through wave-private LDS (94 us) or
// 4 tiles of prefetch (192 us: occupancy 1).)
// ---------------------------------------------------------------------------------------------
#define F32_MAX_SEG 4
struct F32DwSeg { const float* act; const float* dz; const int32_t* count; size_t rows; int N; };
struct F32DwArgs { F32DwSeg seg[F32_MAX_SEG]; int nseg; size_t act_stride, dz_stride, part_stride; };
struct F32TileTab { int base[13]; };          // output tiles of Dense_l: [base[l], base[l + 1]), k-tile major

__global__ void __launch_bounds__(256)
k_mlp_dw_f32(F32Spec S, F32TileTab T, F32DwArgs a, int nsplit, size_t params, float* __restrict__ part) {
    __shared__ float red[3][16][64];
    __shared__ float redb[3][32];
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
    const int krow = Ly.x_off + (k < Ly.fi ? k : 0);            // clamped: rows >= fi are padding
    float bsum = 0.0f;                                           // this lane's share of db[m] (ki == 0 tiles)
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
                x = (ok && k < Ly.fi) ? x : 0.0f;
                z = (ok && m < Ly.fo) ? z : 0.0f;
                bsum += z;
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x, z, acc, 0, 0, 0);
            }
            if (more) {
#pragma unroll
                for (int q = 0; q < 4; q++) { av[q] = an[q]; bv[q] = bn[q]; }
            }
        }
    }
    bsum += __shfl_xor(bsum, 32, 64);                            // the two sample halves of column m
    if (wave > 0) {
#pragma unroll
        for (int r = 0; r < 16; r++) red[wave - 1][r][lane] = acc[r];
        if (lane < 32) redb[wave - 1][lane] = bsum;
    }
    __syncthreads();
    if (wave == 0) {
        float* pp = part + blockIdx.z * a.part_stride + (size_t)sp * params + Ly.w_off;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const float v = ((acc[r] + red[0][r][lane]) + red[1][r][lane]) + red[2][r][lane];
            const int kr = 32 * ki + c_row(r, kk);
            if (kr < Ly.fi && m < Ly.fo) pp[(size_t)kr * Ly.fo + m] = v;
        }
        if (ki == 0 && lane < 32 && m < Ly.fo) pp[(size_t)Ly.fi * Ly.fo + m] = ((bsum + redb[0][lane]) + redb[1][lane]) + redb[2][lane];
    }
}

// The same sums with a 2 x 2 block of output tiles per workgroup (round 4): a wave's 32-sample tile feeds 64 MFMAs from
// 8 + 8 float4 per lane instead of 16 from 4 + 4 -- every record strip is read by half as many workgroups (the launch above
// re-reads the records 4-5 x and runs at ~40 % of its MFMA time: one tile of prefetch hides ~0.4 us of compute per load
// round trip, this one 1.7 us).  Tiles are dealt to (split, wave) exactly as above and the four waves' accumulators are
// summed in the same order, so every partial is BIT-identical to k_mlp_dw_f32's.  Blocks of a layer: ceil(k tiles / 2) x
// ceil(column tiles / 2), k-block major; the absent half of an edge block is skipped (wave-uniform).
struct F32BlockTab { int base[13]; };
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
k_mlp_dw_f32_b2(F32Spec S, F32BlockTab T, F32DwArgs a, int nsplit, size_t params, float* __restrict__ part) {
    __shared__ float red[4][3][16][64];
    __shared__ float redb[2][3][32];
    const int t = blockIdx.x, sp = blockIdx.y;
    int l = 11;
    while (l > 0 && t < T.base[l]) l--;
    const F32Layer Ly = S.L[l];
    const int ntk = (Ly.fi + 31) / 32, ntm = (Ly.fo + 31) / 32;
    const int nbm = (ntm + 1) / 2;
    const int kb = (t - T.base[l]) / nbm, mb = (t - T.base[l]) % nbm;
    const int ki0 = 2 * kb, mj0 = 2 * mb;
    const bool k1 = ki0 + 1 < ntk, m1 = mj0 + 1 < ntm;          // wave-uniform: the block's second k / column tile exists
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, kk = lane >> 5;
    f32x16 acc[2][2];
#pragma unroll
    for (int x = 0; x < 2; x++)
#pragma unroll
        for (int y = 0; y < 2; y++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[x][y][r] = 0.0f;
    int kq[2], mq[2], krow[2], mrow[2];
#pragma unroll
    for (int x = 0; x < 2; x++) {
        kq[x] = 32 * (ki0 + x) + i; mq[x] = 32 * (mj0 + x) + i;
        krow[x] = Ly.x_off + (kq[x] < Ly.fi ? kq[x] : 0);       // clamped: rows >= fi are padding
        mrow[x] = Ly.dz_off + (mq[x] < Ly.fo ? mq[x] : 0);
    }
    float bsum[2] = {0.0f, 0.0f};
    for (int sgi = 0; sgi < a.nseg; sgi++) {
        const F32DwSeg sg = a.seg[sgi];
        const size_t nrows = f32_rows(sg.rows, sg.N, sg.count ? sg.count + blockIdx.z : nullptr);
        const size_t ntile = (nrows + 31) / 32;
        const float* act = sg.act + blockIdx.z * a.act_stride;
        const float* dz = sg.dz + blockIdx.z * a.dz_stride;
        const size_t step = (size_t)nsplit * 4;
        size_t tl = (size_t)sp * 4 + wave;
        f32x4 av[2][4], bv[2][4], an[2][4], bn[2][4];
        auto load = [&](size_t t_, f32x4 (&a_)[2][4], f32x4 (&b_)[2][4]) {
#pragma unroll
            for (int x = 0; x < 2; x++) {
                const float* xa = act + (t_ * S.act + krow[x]) * 32 + 16 * kk;
                const float* za = dz + (t_ * S.dz + mrow[x]) * 32 + 16 * kk;
#pragma unroll
                for (int q = 0; q < 4; q++) { a_[x][q] = *(const f32x4*)(xa + 4 * q); b_[x][q] = *(const f32x4*)(za + 4 * q); }
            }
        };
        // padding rows (k >= fi, m >= fo: edge blocks only) and the samples beyond the count (last tile only) enter as zeros:
        // masked IN PLACE once per tile and only where they occur (wave-uniform tests), not with four selects per MFMA
        const bool edge = 32 * (ki0 + 2) > Ly.fi || 32 * (mj0 + 2) > Ly.fo;
        auto mask_tile = [&](size_t t_, f32x4 (&a_)[2][4], f32x4 (&b_)[2][4]) {
            const bool ragged = (t_ + 1) * 32 > nrows;
            if (!edge && !ragged) return;
            const size_t s0 = t_ * 32 + 16 * kk;
#pragma unroll
            for (int x = 0; x < 2; x++)
#pragma unroll
                for (int j = 0; j < 16; j++) {
                    const bool ok = s0 + j < nrows;
                    a_[x][j >> 2][j & 3] = (ok && kq[x] < Ly.fi) ? a_[x][j >> 2][j & 3] : 0.0f;
                    b_[x][j >> 2][j & 3] = (ok && mq[x] < Ly.fo) ? b_[x][j >> 2][j & 3] : 0.0f;
                }
        };
        auto tile_mma = [&](const f32x4 (&a_)[2][4], const f32x4 (&b_)[2][4]) {
#pragma unroll
            for (int j = 0; j < 16; j++) {
                const float x0 = a_[0][j >> 2][j & 3], x1 = a_[1][j >> 2][j & 3];
                const float z0 = b_[0][j >> 2][j & 3], z1 = b_[1][j >> 2][j & 3];
                bsum[0] += z0; bsum[1] += z1;
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(x0, z0, acc[0][0], 0, 0, 0);
                if (m1) acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(x0, z1, acc[0][1], 0, 0, 0);
                if (k1) acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(x1, z0, acc[1][0], 0, 0, 0);
                if (k1 && m1) acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(x1, z1, acc[1][1], 0, 0, 0);
            }
        };
        // two tiles per trip, ping-pong between the operand sets (no register copies)
        if (tl < ntile) load(tl, av, bv);
        while (tl < ntile) {
            if (tl + step < ntile) load(tl + step, an, bn);
            mask_tile(tl, av, bv);
            tile_mma(av, bv);
            tl += step;
            if (tl >= ntile) break;
            if (tl + step < ntile) load(tl + step, av, bv);
            mask_tile(tl, an, bn);
            tile_mma(an, bn);
            tl += step;
        }
    }
#pragma unroll
    for (int x = 0; x < 2; x++) bsum[x] += __shfl_xor(bsum[x], 32, 64);          // the two sample halves of column m
    if (wave > 0) {
#pragma unroll
        for (int x = 0; x < 2; x++)
#pragma unroll
            for (int y = 0; y < 2; y++)
#pragma unroll
                for (int r = 0; r < 16; r++) red[2 * x + y][wave - 1][r][lane] = acc[x][y][r];
        if (lane < 32) { redb[0][wave - 1][lane] = bsum[0]; redb[1][wave - 1][lane] = bsum[1]; }
    }
    __syncthreads();
    if (wave == 0) {
        float* pp = part + blockIdx.z * a.part_stride + (size_t)sp * params + Ly.w_off;
#pragma unroll
        for (int x = 0; x < 2; x++)
#pragma unroll
            for (int y = 0; y < 2; y++) {
                if ((x && !k1) || (y && !m1)) continue;
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const float v = ((acc[x][y][r] + red[2 * x + y][0][r][lane]) + red[2 * x + y][1][r][lane]) + red[2 * x + y][2][r][lane];
                    const int kr = 32 * (ki0 + x) + c_row(r, kk);
                    if (kr < Ly.fi && mq[y] < Ly.fo) pp[(size_t)kr * Ly.fo + mq[y]] = v;
                }
            }
        if (ki0 == 0 && lane < 32) {
#pragma unroll
            for (int y = 0; y < 2; y++)
                if ((!y || m1) && mq[y] < Ly.fo)
                    pp[(size_t)Ly.fi * Ly.fo + mq[y]] = ((bsum[y] + redb[y][0][lane]) + redb[y][1][lane]) + redb[y][2][lane];
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
        // (measured: unroll 16 -- 16 loads in flight per thread -- is the fastest form hipcc makes of this loop; fully
        // unrolled float4 variants spill at the register cap of a 512 / 1024-thread workgroup and run 3 x slower)
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
// 512 threads: 128 outputs of the view layer x 4 groups that split its 283 inputs (k = g, g + 4, ...: one 128-float row of
// the kernel per load, coalesced), partial sums joined in a fixed order through LDS.  (Until round 4: 128 threads, each
// walking all 283 rows -- 36 rounds of 8 dependent-latency loads, 40 us standalone; this form: ~10 us.)
__global__ void __launch_bounds__(512)
k_bkgd_hit_rays(F32Spec S, const float* __restrict__ P, const float* __restrict__ trunk, const float* __restrict__ view27,
                const int32_t* __restrict__ idx, const int32_t* __restrict__ count, float* __restrict__ raw_tail) {
    __shared__ float x10[HITRAYS_PER_WG][288];
    __shared__ float hc[HITRAYS_PER_WG][128];
    __shared__ float part[3][HITRAYS_PER_WG][128];
    const int n = *count;
    const int j0 = blockIdx.x * HITRAYS_PER_WG;
    if (j0 >= n) return;
    const int tid = threadIdx.x, o = tid & 127, g = tid >> 7;
    for (int i = tid; i < HITRAYS_PER_WG * 283; i += 512) {
        const int r = i / 283, f = i - r * 283;
        float v = 0.0f;
        if (j0 + r < n) v = f < 256 ? trunk[f] : view27[(size_t)idx[j0 + r] * 27 + (f - 256)];
        x10[r][f] = v;
    }
    __syncthreads();
    const float* W10 = P + S.L[10].w_off;
    float acc[HITRAYS_PER_WG];
#pragma unroll
    for (int r = 0; r < HITRAYS_PER_WG; r++) acc[r] = 0.0f;
#pragma unroll 8
    for (int k = g; k < 283; k += 4) {
        const float w = W10[k * 128 + o];
#pragma unroll
        for (int r = 0; r < HITRAYS_PER_WG; r++) acc[r] = fmaf(w, x10[r][k], acc[r]);
    }
    if (g > 0) {
#pragma unroll
        for (int r = 0; r < HITRAYS_PER_WG; r++) part[g - 1][r][o] = acc[r];
    }
    __syncthreads();
    if (g == 0) {
        const float b = W10[283 * 128 + o];
#pragma unroll
        for (int r = 0; r < HITRAYS_PER_WG; r++) {
            const float v = (((b + acc[r]) + part[0][r][o]) + part[1][r][o]) + part[2][r][o];
            hc[r][o] = (v != v) ? v : fmaxf(v, 0.0f);
        }
    }
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
    for (int l = 0; l < 12; l++) {
        T.base[l] = nt;
        nt += ((S.L[l].fi + 31) / 32) * ((S.L[l].fo + 31) / 32);
    }
    T.base[12] = nt;
    return T;
}

F32BlockTab block_table(const F32Spec& S) {
    F32BlockTab T;
    int nb = 0;
    for (int l = 0; l < 12; l++) {
        T.base[l] = nb;
        nb += (((S.L[l].fi + 31) / 32 + 1) / 2) * (((S.L[l].fo + 31) / 32 + 1) / 2);
    }
    T.base[12] = nb;
    return T;
}
// The 2 x 2 blocks pay where a wave has many sample tiles to walk: the W = 256 instrument / --precision f32 (8.52 -> 8.05 ms
// per launch at 4096 rays).  On the K object MLPs of a pose-optimisation step (cfg4: ~5 tiles per wave, 42 blocks x 8 splits
// x K workgroups at occupancy 2 instead of 172 x 8 x K at 4) they LOSE: 87 -> 101-103 us (tools/time_objf32.py), cfg4
// 636 -> 629 k rays/s -- the sixth layout of that launch measured slower than one tile per workgroup (review item 6).
// (Bit-identical partials either way.)
bool dw_b2_enabled(bool wide) { return wide; }
void launch_dw_f32(hipStream_t s, const F32Spec& S, const F32DwArgs& a, int nsplit, int K, size_t params, float* scratch) {
    if (dw_b2_enabled(S.W == 256)) {
        const F32BlockTab T = block_table(S);
        hipLaunchKernelGGL(k_mlp_dw_f32_b2, dim3(T.base[12], nsplit, K), dim3(256), 0, s, S, T, a, nsplit, params, scratch);
        durf::note_dispatch(DURF_DISPATCH_F32_DW_B2);
    } else {
        const F32TileTab T = tile_table(S);
        hipLaunchKernelGGL(k_mlp_dw_f32, dim3(T.base[12], nsplit, K), dim3(256), 0, s, S, T, a, nsplit, params, scratch);
        durf::note_dispatch(DURF_DISPATCH_F32_DW_TILE);
    }
}

template <int W, int IN, bool TRAIN, bool ENC, bool X3 = false>
void launch_fwd_k(hipStream_t s, dim3 grid, size_t rows, int N, const float* enc, const float* view27, const int32_t* ray_idx,
                  const int32_t* count, const float* P, const float* ws, float* raw, float* act, const F32FwdBatch& bs,
                  const F32Enc& ei) {
    constexpr int lds = F32Cfg<W>::LDS_FLOATS * (int)sizeof(float);
    (void)hipFuncSetAttribute((const void*)k_mlp_fwd_f32<W, IN, TRAIN, ENC, X3>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL((k_mlp_fwd_f32<W, IN, TRAIN, ENC, X3>), grid, dim3(W * 2), lds, s, rows, N, enc, view27, ray_idx, count, P, ws,
                       raw, act, bs, ei);
}
// ei != nullptr (W = 128 only): the kernel encodes its own tiles from the ray data instead of reading `enc`
template <int W, int IN>
int launch_fwd(hipStream_t s, size_t rows, int N, const float* enc, const float* view27, const int32_t* ray_idx,
               const int32_t* count, const float* P, const float* ws, float* raw, float* act, int K, const F32FwdBatch& bs,
               const F32Enc* ei = nullptr, bool x3 = false) {
    const unsigned nt_ = durf_cdiv(rows, 32), cap = K > 1 ? 128u : 512u;          // workgroups per MLP (see k_mlp_fwd_f32)
    const dim3 grid(nt_ < cap ? nt_ : cap, K);
    if constexpr (W == 128) {
        if (ei && x3) {          // the object MLPs on the bf16 matrix pipe with split operands (wstream: durf_mlp_f32_pack_x3)
            if (act) launch_fwd_k<W, IN, true, true, true>(s, grid, rows, N, enc, view27, ray_idx, count, P, ws, raw, act, bs, *ei);
            else launch_fwd_k<W, IN, false, true, true>(s, grid, rows, N, enc, view27, ray_idx, count, P, ws, raw, act, bs, *ei);
            return 0;
        }
        if (ei) {
            if (act) launch_fwd_k<W, IN, true, true>(s, grid, rows, N, enc, view27, ray_idx, count, P, ws, raw, act, bs, *ei);
            else launch_fwd_k<W, IN, false, true>(s, grid, rows, N, enc, view27, ray_idx, count, P, ws, raw, act, bs, *ei);
            return 0;
        }
    }
    if (act) launch_fwd_k<W, IN, true, false>(s, grid, rows, N, enc, view27, ray_idx, count, P, ws, raw, act, bs, F32Enc{});
    else launch_fwd_k<W, IN, false, false>(s, grid, rows, N, enc, view27, ray_idx, count, P, ws, raw, act, bs, F32Enc{});
    return 0;
}
template <int W, int IN>
int launch_bwd(hipStream_t s, size_t rows, int N, const float* draw, const int32_t* ray_idx, const int32_t* count,
               const float* P, const float* ws, const float* act, float* dz, float* d_enc, int K, const F32BwdBatch& bs,
               bool x3 = false) {
    constexpr int lds = F32Cfg<W>::LDS_FLOATS * (int)sizeof(float);
    const unsigned nt_ = durf_cdiv(rows, 32), cap = K > 1 ? 128u : 512u;
    const dim3 grid(nt_ < cap ? nt_ : cap, K), block(W * 2);
    if constexpr (W == 128) {
        if (x3) {
            (void)hipFuncSetAttribute((const void*)k_mlp_bwd_f32<W, IN, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            hipLaunchKernelGGL((k_mlp_bwd_f32<W, IN, true>), grid, block, lds, s, rows, N, draw, ray_idx, count, P, ws, act, dz, d_enc, bs);
            return 0;
        }
    }
    (void)hipFuncSetAttribute((const void*)k_mlp_bwd_f32<W, IN>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL((k_mlp_bwd_f32<W, IN>), grid, block, lds, s, rows, N, draw, ray_idx, count, P, ws, act, dz, d_enc, bs);
    return 0;
}
template <int W, int IN>
size_t wstream_floats() { return F32Sched<W, false>::stream_floats() + F32Sched<W, true>::stream_floats(); }
size_t tile_rows(size_t rows) { return (rows + 31) / 32 * 32; }

}  // namespace

extern "C" {

size_t durf_mlp_f32_act_floats(int width, int in_dim) { return (size_t)f32_spec(width, in_dim).act; }
size_t durf_mlp_f32_dz_floats(int width, int in_dim) { return (size_t)f32_spec(width, in_dim).dz; }
size_t durf_mlp_f32_dw_scratch_floats(int width, int in_dim, int nsplit) {
    return (size_t)nsplit * durf_layer_offset(width, in_dim, 12, 0);
}
size_t durf_mlp_f32_wstream_floats(int width) { return width == 256 ? wstream_floats<256, 60>() : wstream_floats<128, 63>(); }

#define F32_REQUIRE_MLP(width, in_dim)                                                               \
    DURF_REQUIRE((width == 256 && in_dim == 60) || (width == 128 && in_dim == 63),                   \
                 "built for the two MLPs of the model: (width, in_dim) = (256, 60) or (128, 63)")

int durf_mlp_f32_pack(void* stream, int width, int in_dim, int K, const float* mlp_params, size_t param_stride,
                      float* wstream) {
    F32_REQUIRE_MLP(width, in_dim);
    if (K <= 0) return 0;
    const size_t n = durf_mlp_f32_wstream_floats(width);
    if (width == 256)
        hipLaunchKernelGGL((k_f32_pack<256, 60>), dim3(durf_cdiv(n, 256), K), dim3(256), 0, (hipStream_t)stream, mlp_params, wstream, param_stride, n);
    else
        hipLaunchKernelGGL((k_f32_pack<128, 63>), dim3(durf_cdiv(n, 256), K), dim3(256), 0, (hipStream_t)stream, mlp_params, wstream, param_stride, n);
    DURF_CHECK_LAUNCH("durf_mlp_f32_pack");
    return 0;
}

// the W = 128 stream with every weight as a (hi, lo) bf16 pair: what the DURF_F32_X3 object kernels consume
int durf_mlp_f32_pack_x3(void* stream, int K, const float* obj_params, size_t param_stride, float* wstream) {
    if (K <= 0) return 0;
    const size_t n = durf_mlp_f32_wstream_floats(DURF_W_OBJ);
    hipLaunchKernelGGL((k_f32_pack<128, 63, true>), dim3(durf_cdiv(n, 256), K), dim3(256), 0, (hipStream_t)stream, obj_params, wstream,
                       param_stride, n);
    DURF_CHECK_LAUNCH("durf_mlp_f32_pack_x3");
    return 0;
}

int durf_mlp_fwd_f32(void* stream, int width, int in_dim, size_t rows, int N, const float* enc, const float* view27,
                     const int32_t* ray_idx, const int32_t* count, const float* mlp_params, const float* wstream,
                     float* raw, float* act) {
    F32_REQUIRE_MLP(width, in_dim);
    DURF_REQUIRE(enc != nullptr || width == 256, "the constant encoding (enc == NULL) is the background MLP's");
    if (rows == 0) return 0;
    if (width == 256) launch_fwd<256, 60>((hipStream_t)stream, rows, N, enc, view27, ray_idx, count, mlp_params, wstream, raw, act, 1, F32FwdBatch{});
    else launch_fwd<128, 63>((hipStream_t)stream, rows, N, enc, view27, ray_idx, count, mlp_params, wstream, raw, act, 1, F32FwdBatch{});
    DURF_CHECK_LAUNCH("durf_mlp_fwd_f32");
    return 0;
}

int durf_mlp_bwd_f32(void* stream, int width, int in_dim, size_t rows, int N, const float* draw,
                     const int32_t* ray_idx, const int32_t* count, const float* mlp_params, const float* wstream,
                     const float* act, float* dz, float* d_enc) {
    F32_REQUIRE_MLP(width, in_dim);
    if (rows == 0) return 0;
    if (width == 256) launch_bwd<256, 60>((hipStream_t)stream, rows, N, draw, ray_idx, count, mlp_params, wstream, act, dz, d_enc, 1, F32BwdBatch{});
    else launch_bwd<128, 63>((hipStream_t)stream, rows, N, draw, ray_idx, count, mlp_params, wstream, act, dz, d_enc, 1, F32BwdBatch{});
    DURF_CHECK_LAUNCH("durf_mlp_bwd_f32");
    return 0;
}

int durf_mlp_dw_f32(void* stream, int width, int in_dim, size_t rows, int N, const int32_t* count, const float* act,
                    const float* dz, int nsplit, float* scratch, float* grad_mlp) {
    DURF_REQUIRE(width == 256 || width == 128, "width must be 256 or 128");
    DURF_REQUIRE(nsplit >= 1 && nsplit <= 1024, "1 <= nsplit <= 1024");
    const F32Spec S = f32_spec(width, in_dim);
    const size_t params = durf_layer_offset(width, in_dim, 12, 0);
    F32DwArgs a{};
    a.nseg = 1;
    a.seg[0] = F32DwSeg{act, dz, count, rows, N};
    hipStream_t s = (hipStream_t)stream;
    launch_dw_f32(s, S, a, nsplit, 1, params, scratch);
    hipLaunchKernelGGL(k_dw_f32_reduce, dim3(durf_cdiv(params, 256), 1), dim3(256), 0, s, params, nsplit, scratch, (size_t)0,
                       grad_mlp, (size_t)0, 0);
    DURF_CHECK_LAUNCH("durf_mlp_dw_f32");
    return 0;
}

int durf_bkgd_const_trunk_f32(void* stream, const float* bkgd_params, float* trunk) {
    const F32Spec S = f32_spec(DURF_W_BKGD, 60);
    hipLaunchKernelGGL(k_bkgd_const_trunk, dim3(1), dim3(1024), 0, (hipStream_t)stream, S, bkgd_params, trunk);
    DURF_CHECK_LAUNCH("durf_bkgd_const_trunk_f32");
    return 0;
}

int durf_bkgd_hit_rays_f32(void* stream, int B, const float* view27, const float* bkgd_params, const int32_t* idx,
                           const int32_t* count, const float* trunk, float* raw_tail) {
    if (B <= 0) return 0;
    const F32Spec S = f32_spec(DURF_W_BKGD, 60);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_bkgd_hit_rays, dim3(durf_cdiv(B, HITRAYS_PER_WG)), dim3(512), 0, s, S, bkgd_params, trunk, view27, idx,
                       count, raw_tail);
    DURF_CHECK_LAUNCH("durf_bkgd_hit_rays_f32");
    return 0;
}

/* ---- the K object MLPs of a step on the fp32 kernels, one launch per phase (objects in blockIdx.y / .z) ---------- */
size_t durf_objf32_act_stride(int B, int N) { return tile_rows((size_t)B * N) * f32_spec(DURF_W_OBJ, 63).act; }
size_t durf_objf32_dz_stride(int B, int N) { return tile_rows((size_t)B * N) * f32_spec(DURF_W_OBJ, 63).dz; }

int durf_objf32_fwd_batch(void* stream, int K, int B, int N, const int32_t* idx, const int32_t* count, const float* enc,
                          const float* view27, const float* obj_params, size_t param_stride, const float* wstream,
                          float* raw, float* act, const float* t_vals, const float* origins_s, const float* dirs_s,
                          const float* radii, const float* barf_w, int flags) {
    DURF_REQUIRE(K >= 1 && K <= DURF_MAX_OBJ, "1 <= K <= DURF_MAX_OBJ");
    if (B <= 0) return 0;
    const size_t rows = (size_t)B * N;
    DURF_REQUIRE(enc != nullptr || (t_vals && origins_s && dirs_s && radii && barf_w), "enc, or the ray data to encode from");
    F32FwdBatch bs{rows * 63, (size_t)B, param_stride, durf_mlp_f32_wstream_floats(DURF_W_OBJ), rows * 4, durf_objf32_act_stride(B, N)};
    F32Enc ei{t_vals, origins_s, dirs_s, radii, BarfW{}, flags & (DURF_ENC_NO_INTEGRATION | DURF_ENC_CYLINDER)};
    if (!enc) for (int i = 0; i < 10; i++) ei.w.w[i] = barf_w[i];
    DURF_REQUIRE(!(flags & DURF_F32_X3) || enc == nullptr, "DURF_F32_X3: the self-encoding forward");
    launch_fwd<DURF_W_OBJ, 63>((hipStream_t)stream, rows, N, enc, view27, idx, count, obj_params, wstream, raw, act, K, bs,
                               enc ? nullptr : &ei, (flags & DURF_F32_X3) != 0);
    DURF_CHECK_LAUNCH("durf_objf32_fwd_batch");
    return 0;
}

int durf_objf32_bwd_batch(void* stream, int K, int B, int N, const int32_t* idx, const int32_t* count, const float* draw,
                          const float* obj_params, size_t param_stride, const float* wstream, const float* act,
                          float* dz, float* d_enc) {
    DURF_REQUIRE(K >= 1 && K <= DURF_MAX_OBJ, "1 <= K <= DURF_MAX_OBJ");
    if (B <= 0) return 0;
    const size_t rows = (size_t)B * N;
    F32BwdBatch bs{(size_t)B, param_stride, durf_mlp_f32_wstream_floats(DURF_W_OBJ), durf_objf32_act_stride(B, N),
                   durf_objf32_dz_stride(B, N), rows * DURF_ENC_DIM};
    launch_bwd<DURF_W_OBJ, 63>((hipStream_t)stream, rows, N, draw, idx, count, obj_params, wstream, act, dz, d_enc, K, bs);
    DURF_CHECK_LAUNCH("durf_objf32_bwd_batch");
    return 0;
}

int durf_objf32_bwd_batch_x3(void* stream, int K, int B, int N, const int32_t* idx, const int32_t* count, const float* draw,
                             const float* obj_params, size_t param_stride, const float* wstream, const float* act,
                             float* dz, float* d_enc) {
    DURF_REQUIRE(K >= 1 && K <= DURF_MAX_OBJ, "1 <= K <= DURF_MAX_OBJ");
    if (B <= 0) return 0;
    const size_t rows = (size_t)B * N;
    F32BwdBatch bs{(size_t)B, param_stride, durf_mlp_f32_wstream_floats(DURF_W_OBJ), durf_objf32_act_stride(B, N),
                   durf_objf32_dz_stride(B, N), rows * DURF_ENC_DIM};
    launch_bwd<DURF_W_OBJ, 63>((hipStream_t)stream, rows, N, draw, idx, count, obj_params, wstream, act, dz, d_enc, K, bs, true);
    DURF_CHECK_LAUNCH("durf_objf32_bwd_batch_x3");
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
    F32DwArgs a{};
    a.nseg = nlevels;
    for (int l = 0; l < nlevels; l++) a.seg[l] = F32DwSeg{act[l], dz[l], count, (size_t)B * N, N};
    a.act_stride = durf_objf32_act_stride(B, N); a.dz_stride = durf_objf32_dz_stride(B, N);
    a.part_stride = (size_t)nsplit * params;
    hipStream_t s = (hipStream_t)stream;
    launch_dw_f32(s, S, a, nsplit, K, params, scratch);
    hipLaunchKernelGGL(k_dw_f32_reduce, dim3(durf_cdiv(params, 256), K), dim3(256), 0, s, params, nsplit, scratch,
                       a.part_stride, grad_obj, grad_stride, 0);
    DURF_CHECK_LAUNCH("durf_objf32_dw_batch");
    return 0;
}

}  // extern "C"
