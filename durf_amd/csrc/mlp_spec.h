// Static description of the two MLPs of the ray pipeline (obbpose_model.py:293-418) as the
// fused kernels see them: 11 "stages", each a dense layer (or a fused pair) evaluated as
//      D[out_feature, sample] = W^T[out_feature, k] * X[k, sample]
// with v_mfma_f32_32x32x16_bf16, A = weights (LDS), B = activations (registers).
//
// Orientation trick: with samples on the MFMA N axis, the C/D layout (lane = sample,
// regs = 16 of 32 output features) IS the B-operand layout of the next layer up to a fixed
// permutation of the k index, which is folded into the packed weights.  Activations never
// leave registers between layers; only weights stream through LDS.
//
//   accumulator reg r of lane (n, hi) holds out-feature 32*mo + (r&3) + 8*(r>>2) + 4*hi
//   -> B fragment of k-step ks = 2*mo + (r>>3), slot e = r&7:
//      feature(ks, hi, e) = 16*ks + (e&3) + 8*(e>>2) + 4*hi              ("C-perm")
//   inputs that come from memory (encoding, view dirs) use the natural order
//      feature(ks, hi, e) = 16*ks + 8*hi + e                              ("natural")
#pragma once
#include <type_traits>
#include "durf_common.h"

template <int W_>
struct MlpSpec {
    static constexpr int W = W_;
    static constexpr int WT = W / 32;     // output M-tiles of a trunk layer
    static constexpr int KW = W / 16;     // k-steps spanning W features
    static constexpr int KE = DURF_ENC_DIM / 16;    // 4
    static constexpr int KV = DURF_VIEW_DIM / 16;   // 2
    static constexpr int WC = 128;        // net_width_condition
    static constexpr int KC = WC / 16;    // 8
    static constexpr int CT = WC / 32;    // 4
    static constexpr int NSTAGE = 11;
    // stage -> (#output M-tiles, #k-steps)
    //  0: enc -> W relu | 1-4,6,7: W -> W relu | 5: [W,enc] -> W relu
    //  8: W -> [bottleneck W (linear) ; density 1] | 9: [bottleneck, view] -> 128 relu | 10: 128 -> rgb 3
    __host__ __device__ static constexpr int n_mt(int s) { return s <= 7 ? WT : (s == 8 ? WT + 1 : (s == 9 ? CT : 1)); }
    __host__ __device__ static constexpr int n_ks(int s) {
        return s == 0 ? KE : (s == 5 ? KW + KE : (s == 9 ? KW + KV : (s == 10 ? KC : KW)));
    }
    // number of leading k-steps whose B fragments come from the previous stage (C-perm order)
    __host__ __device__ static constexpr int n_ks_perm(int s) { return s == 0 ? 0 : (s == 10 ? KC : KW); }
    // one weight tile = one output M-tile: n_ks chunks of 1 KB (64 lanes x 8 bf16) + 1 bias chunk
    __host__ __device__ static constexpr int tile_chunks(int s) { return n_ks(s) + 1; }
    __host__ __device__ static constexpr int stage_chunk_base(int s) {
        int c = 0;
        for (int i = 0; i < s; i++) c += n_mt(i) * tile_chunks(i);
        return c;
    }
    static constexpr int TOTAL_CHUNKS = stage_chunk_base(NSTAGE);
    static constexpr int MAX_TILE_CHUNKS = KW + KE + 1;
    // activation stash (training): regions 0..7 trunk outputs, 8 bottleneck, 9 view-layer out
    static constexpr int NSTASH = 10;
    __host__ __device__ static constexpr int stash_ks(int j) { return j == 9 ? KC : KW; }
    __host__ __device__ static constexpr int stash_ks_before(int j) { return j * KW; }
    static constexpr int STASH_KS_TOTAL = 9 * KW + KC;   // KB per 32-row tile
};

// flax Dense_l shapes (fan_in, fan_out) for an MLP of width W and input dim `in_dim`
__host__ __device__ inline void durf_layer_shape(int W, int in_dim, int l, int* fin, int* fout) {
    int fi, fo;
    if (l == 0) { fi = in_dim; fo = W; }
    else if (l <= 4) { fi = W; fo = W; }
    else if (l == 5) { fi = W + in_dim; fo = W; }
    else if (l <= 7) { fi = W; fo = W; }
    else if (l == 8) { fi = W; fo = 1; }
    else if (l == 9) { fi = W; fo = W; }
    else if (l == 10) { fi = W + 27; fo = 128; }
    else { fi = 128; fo = 3; }
    *fin = fi; *fout = fo;
}
__host__ __device__ inline size_t durf_layer_offset(int W, int in_dim, int layer, int want_bias) {
    size_t off = 0;
    for (int l = 0; l < layer; l++) {
        int fi, fo;
        durf_layer_shape(W, in_dim, l, &fi, &fo);
        off += (size_t)fi * fo + fo;
    }
    if (want_bias) {
        int fi, fo;
        durf_layer_shape(W, in_dim, layer, &fi, &fo);
        off += (size_t)fi * fo;
    }
    return off;
}

// Forward stage s, output M-tile mo, row i  ->  (flax layer, output column) or layer = -1
template <int W>
__host__ __device__ inline void durf_fwd_out_col(int s, int mo, int i, int* layer, int* col) {
    using S = MlpSpec<W>;
    int L = -1, c = 0;
    if (s <= 7) { L = s; c = 32 * mo + i; }
    else if (s == 8) { if (mo < S::WT) { L = 9; c = 32 * mo + i; } else if (i < 1) { L = 8; c = i; } }
    else if (s == 9) { L = 10; c = 32 * mo + i; }
    else { if (i < 3) { L = 11; c = i; } }
    *layer = L; *col = c;
}
// Forward stage s, k-step ks, lane half hi, slot e -> input row of the flax kernel, or -1 (pad)
template <int W>
__host__ __device__ inline int durf_fwd_in_row(int s, int ks, int hi, int e, int in_dim) {
    using S = MlpSpec<W>;
    const int np = S::n_ks_perm(s);
    if (ks < np) return 16 * ks + (e & 3) + 8 * (e >> 2) + 4 * hi;
    const int f = 16 * (ks - np) + 8 * hi + e;
    if (s == 0) return f < in_dim ? f : -1;
    if (s == 5) return f < in_dim ? W + f : -1;
    if (s == 9) return f < 27 ? W + f : -1;
    return -1;
}

// ---------------------------------------------------------------------------
// Weight fragments LDS -> registers: a REGISTER RING of 2-4 ds_read_b128 in flight, issued from inline asm with
// counted s_waitcnt lgkmcnt(n).  Left to itself hipcc (at the 256-register cap of these kernels) sinks every fragment
// read to just before the MFMA that consumes it -- "ds_read_b128 v[40:43]; s_waitcnt lgkmcnt(0); v_mfma", 1184 times per
// 256-sample block -- so each MFMA waited for a full LDS round trip.  LDS reads return in order, so fragment i is valid
// once at most min(ring - 1, reads after it) younger reads are outstanding; reads hipcc issues by itself in between
// (bias rows, masks) and scalar loads only make that wait conservative (at least one more LDS read has completed than
// the count requires, and the oldest completes first).
// Two output tiles at once (read i: tile i & 1, k-step i >> 1): their MFMAs alternate, so consecutive MFMAs never share
// an accumulator and each B fragment is used twice back to back.  `hook(ks)` runs after the second MFMA of k-step ks.
// ---------------------------------------------------------------------------
// ring depth, measured on MI355X (same-box A/B, 4096 rays x 128 samples; tools/time_fwd.py): forward 2 (690 -> 669 us
// with the training stores, 2-4 alike; 6-8 slower), backward 4 (605 -> 597 us at 2, 582 at 4)
#ifndef FWD_LDS_RING
#define FWD_LDS_RING 2
#endif
#ifndef BWD_LDS_RING
#define BWD_LDS_RING 4
#endif
typedef int v4i_ __attribute__((ext_vector_type(4)));
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}
template <int OFF>
__device__ __forceinline__ v4i_ lds_read16(unsigned addr) {
    v4i_ r;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
    return r;
}
template <int N>
__device__ __forceinline__ void lds_wait(v4i_& frag) {     // frag is valid once at most N younger LDS reads are outstanding
    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(frag) : "n"(N));
}
__device__ __forceinline__ unsigned lds_addr_of(const char* p) {
    return (unsigned)(size_t)(const __attribute__((address_space(3))) char*)p;
}
template <int NA, int NB, int RING, class H>
__device__ __forceinline__ void mma_pair_ring(const char* slot0, const char* slot1, int lane, const bf16x8* inA,
                                              const bf16x8* inB, f32x16& acc0, f32x16& acc1, H&& hook) {
    constexpr int T = NA + NB, NR = 2 * T;
    constexpr int D = RING < NR ? RING : NR;
    const unsigned l0 = lds_addr_of(slot0) + lane * 16, l1 = lds_addr_of(slot1) + lane * 16;
    v4i_ ring[D];
    static_for<0, D>([&](auto i_) {
        constexpr int i = decltype(i_)::value;
        ring[i] = lds_read16<(i >> 1) * 1024>((i & 1) ? l1 : l0);
    });
    static_for<0, NR>([&](auto i_) {
        constexpr int i = decltype(i_)::value, ks = i >> 1;
        constexpr int later = (NR - 1 - i) < (D - 1) ? (NR - 1 - i) : (D - 1);
        lds_wait<later>(ring[i % D]);
        const bf16x8 a = __builtin_bit_cast(bf16x8, ring[i % D]);
        const bf16x8 b = ks < NA ? inA[ks < NA ? ks : 0] : inB[ks < NA ? 0 : ks - NA];
        if constexpr (i & 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc1, 0, 0, 0);
        else acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
        if constexpr (i + D < NR) ring[i % D] = lds_read16<((i + D) >> 1) * 1024>(((i + D) & 1) ? l1 : l0);
        if constexpr (i & 1) hook(std::integral_constant<int, ks>{});
    });
}
