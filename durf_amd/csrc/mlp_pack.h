// Weight packers: fp32 flax parameters -> the bf16 fragment streams the fused kernels read (one thread per 16-byte
// vector).  pack_fwd_vec / pack_bwd_vec write vector `vec` of one MLP's forward / backward stream; the kernels around
// them choose the MLP (k_pack_fwd / k_pack_bwd: blockIdx.y = object of a batched call; k_pack_all: every MLP of the
// model, both streams, ONE launch per training step instead of four).
#pragma once
#include "mlp_spec.h"

// ---------------------------------------------------------------------------
// backward weight stream: for fwd stage s = 10..1, tiles over INPUT features
// ---------------------------------------------------------------------------
// Backward stages in execution order.  b: 0..5 = fwd stages 10,9,8,7,6,5 (trunk rows),
// 6 = input-encoding rows of Dense_5 (skip connection; only needed for box-pose gradients),
// 7..10 = fwd stages 4,3,2,1, 11 = Dense_0 -> d encoding (box-pose gradients only).
template <int W>
struct BwdSpec {
    using S = MlpSpec<W>;
    static constexpr int NB = 12;
    __host__ __device__ static constexpr int fwd_stage(int b) { return b <= 5 ? 10 - b : (b == 6 ? 5 : (b <= 10 ? 11 - b : 0)); }
    __host__ __device__ static constexpr bool is_enc(int b) { return b == 6 || b == 11; }
    __host__ __device__ static constexpr int n_mt(int b) { return is_enc(b) ? S::KE / 2 : (b == 0 ? S::CT : S::WT); }
    __host__ __device__ static constexpr int n_ks(int b) {
        return b == 0 ? 1 : (b == 1 ? S::KC : (b == 2 ? S::KW + 1 : S::KW));
    }
    __host__ __device__ static constexpr int chunk_base(int b) {
        int c = 0;
        for (int i = 0; i < b; i++) c += n_mt(i) * n_ks(i);
        return c;
    }
    static constexpr int TOTAL_CHUNKS = chunk_base(NB);
    static constexpr int MAX_TILE_CHUNKS = S::KW + 1;
};

template <int W>
__device__ __forceinline__ void pack_fwd_vec(int in_dim, const float* __restrict__ P, bf16x8* __restrict__ out, int vec) {
    using S = MlpSpec<W>;
    if (vec >= S::TOTAL_CHUNKS * 64) return;
    const int chunk = vec >> 6, lane = vec & 63;
    int s = 0, base = 0;
    for (; s < S::NSTAGE; s++) {
        const int cnt = S::n_mt(s) * S::tile_chunks(s);
        if (chunk < base + cnt) break;
        base += cnt;
    }
    const int rel = chunk - base;
    const int mo = rel / S::tile_chunks(s), ck = rel % S::tile_chunks(s);
    bf16x8 v;
    if (ck < S::n_ks(s)) {
        const int i = lane & 31, hi = lane >> 5;
        int L, col;
        durf_fwd_out_col<W>(s, mo, i, &L, &col);
        int fi = 0, fo = 0;
        if (L >= 0) durf_layer_shape(W, in_dim, L, &fi, &fo);
        const size_t koff = (L >= 0) ? durf_layer_offset(W, in_dim, L, 0) : 0;
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const int row = durf_fwd_in_row<W>(s, ck, hi, e, in_dim);
            float val = 0.0f;
            if (L >= 0 && col < fo && row >= 0 && row < fi) val = P[koff + (size_t)row * fo + col];
            v[e] = (__bf16)val;
        }
        out[vec] = v;
    } else {
        // bias chunk: floats [hi][r] = bias[out feature 32*mo + (r&3) + 8*(r>>2) + 4*hi]
        float f[4] = {0.f, 0.f, 0.f, 0.f};
        if (lane < 8) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int idx = lane * 4 + j, hi = idx >> 4, r = idx & 15;
                const int i = (r & 3) + 8 * (r >> 2) + 4 * hi;
                int L, col;
                durf_fwd_out_col<W>(s, mo, i, &L, &col);
                if (L >= 0) {
                    int fi, fo;
                    durf_layer_shape(W, in_dim, L, &fi, &fo);
                    if (col < fo) f[j] = P[durf_layer_offset(W, in_dim, L, 1) + col];
                }
            }
        }
        f32x4 fv = {f[0], f[1], f[2], f[3]};
        *(f32x4*)&out[vec] = fv;
    }
}

template <int W>
__device__ __forceinline__ void pack_bwd_vec(int in_dim, const float* __restrict__ P, bf16x8* __restrict__ out, int vec) {
    using S = MlpSpec<W>;
    using Bs = BwdSpec<W>;
    if (vec >= Bs::TOTAL_CHUNKS * 64) return;
    const int chunk = vec >> 6, lane = vec & 63;
    int b = 0, base = 0;
    for (; b < Bs::NB; b++) {
        const int cnt = Bs::n_mt(b) * Bs::n_ks(b);
        if (chunk < base + cnt) break;
        base += cnt;
    }
    const int s = Bs::fwd_stage(b);
    const bool enc_rows = Bs::is_enc(b);
    const int rel = chunk - base;
    const int mo = rel / Bs::n_ks(b), ks = rel % Bs::n_ks(b);
    const int i = lane & 31, hi = lane >> 5;
    // input feature (kernel row) of fwd stage s this A-fragment row stands for
    const int row = enc_rows ? (s == 5 ? W : 0) + 32 * mo + i : 32 * mo + i;
    bf16x8 v;
#pragma unroll
    for (int e = 0; e < 8; e++) {
        const int perm = 16 * ks + (e & 3) + 8 * (e >> 2) + 4 * hi;
        const int nat = 8 * hi + e;
        int L = -1, col = 0;
        if (s == 10) { if (nat < 3) { L = 11; col = nat; } }
        else if (s == 9) { L = 10; col = perm; }
        else if (s == 8) { if (ks < S::KW) { L = 9; col = perm; } else if (nat == 0) { L = 8; col = 0; } }
        else { L = s; col = perm; }
        float val = 0.0f;
        if (L >= 0) {
            int fi, fo;
            durf_layer_shape(W, in_dim, L, &fi, &fo);
            const int row_lim = enc_rows ? fi : ((s == 10) ? 128 : W);   // trunk/bottleneck rows, or the encoding rows
            if (row < row_lim && row < fi && col < fo)
                val = P[durf_layer_offset(W, in_dim, L, 0) + (size_t)row * fo + col];
        }
        v[e] = (__bf16)val;
    }
    out[vec] = v;
}

// Every weight stream of the model as ONE index space: MLP 0 = the background MLP (W = 256), 1..K the object MLPs (W = 128);
// `vec` runs over an MLP's forward vectors, then its backward ones (bwd streams nullable: inference).  Shared by k_pack_all
// (mlp_fwd.hip: its own launch) and by the step prologue (rays.hip: the packing rides in the ray-setup launch).
struct PackAll {
    const float* p_bkgd; bf16x8* f_bkgd; bf16x8* b_bkgd; int in_bkgd;
    const float* p_obj; bf16x8* f_obj; bf16x8* b_obj; int in_obj;
    size_t p_stride, f_stride, b_stride;
};
__device__ __forceinline__ void pack_all_vec(const PackAll& a, int mlp, int vec) {
    if (mlp == 0) {
        constexpr int NF = MlpSpec<256>::TOTAL_CHUNKS * 64, NB = BwdSpec<256>::TOTAL_CHUNKS * 64;
        if (a.p_bkgd == nullptr) return;
        if (vec < NF) pack_fwd_vec<256>(a.in_bkgd, a.p_bkgd, a.f_bkgd, vec);
        else if (a.b_bkgd && vec < NF + NB) pack_bwd_vec<256>(a.in_bkgd, a.p_bkgd, a.b_bkgd, vec - NF);
    } else {
        constexpr int NF = MlpSpec<128>::TOTAL_CHUNKS * 64, NB = BwdSpec<128>::TOTAL_CHUNKS * 64;
        const size_t k = mlp - 1;
        const float* P = a.p_obj + k * a.p_stride;
        if (vec < NF) pack_fwd_vec<128>(a.in_obj, P, (bf16x8*)((char*)a.f_obj + k * a.f_stride), vec);
        else if (a.b_obj && vec < NF + NB) pack_bwd_vec<128>(a.in_obj, P, (bf16x8*)((char*)a.b_obj + k * a.b_stride), vec - NF);
    }
}
