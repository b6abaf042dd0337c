// MLP backward (K11): reverse-mode of obbpose_model.py:305-354 / :369-418.
//
//  (1) k_mlp_bwd  -- fused data path.  Same orientation trick as the forward (mlp_spec.h):
//      dX[in_feature, sample] = W[in_feature, out] * dZ[out, sample], gradients stay in
//      registers from the heads back to layer 1; the ReLU masks come from the forward's
//      bf16 activation stash (same fragment layout, so masking is element-wise), and every
//      dZ is written once, in tile layout, for the weight-gradient GEMM.
//  (2) k_dw       -- dW[out, in] = sum_samples dZ[sample, out] * X[sample, in].  Both
//      operands are stored sample-major (what the fused kernels produce) but the MFMA needs
//      the sample axis in the k-slots of each lane: ds_read_b64_tr_b16 does that transpose
//      on the LDS read path (semantics verified by tools/probes/probe_tr.hip).  Split-K over
//      workgroups with fp32 partials, summed in a fixed order by k_dw_finalize
//      (deterministic; no atomics).  HBM-bound: 1 KB read per sample per 256x256 layer.
#include <stdlib.h>
#include <cstddef>
#include "mlp_pack.h"


// (BwdSpec and the packer bodies: mlp_pack.h)
template <int W>
__global__ void __launch_bounds__(256)
k_pack_bwd(int in_dim, const float* __restrict__ P, bf16x8* __restrict__ out, size_t p_stride, size_t out_stride) {
    P += blockIdx.y * p_stride;                                  // object index (0 for a single MLP)
    out = (bf16x8*)((char*)out + blockIdx.y * out_stride);
    pack_bwd_vec<W>(in_dim, P, out, blockIdx.x * blockDim.x + threadIdx.x);
}

// ---------------------------------------------------------------------------
// fused backward data path
// ---------------------------------------------------------------------------
struct BPipe {        // same protocol as WPipe (mlp_fwd.hip): asm LDS-DMA, counted wait, raw barrier
    i32x4 rsrc;          // the packed (transposed) weight stream
    unsigned gnext;      // byte offset of the next tile group (the persistent loop wraps it to 0)
    unsigned lds0;
    char* lds;
    int slot_bytes, par, wave, lane;
    int nw;              // waves of the workgroup (8, or 4: launch_mlp_bwd)
    int since;           // stores this wave issued after its last weight DMA (lower bound)
    __device__ __forceinline__ void skip(int chunks) { gnext += chunks * 1024u; }
    __device__ __forceinline__ void issue(int slot, int chunks) {
        const unsigned dst = lds0 + (unsigned)(slot * slot_bytes);
        for (int c = wave; c < chunks; c += nw)
            lds_dma16_cached(rsrc, gnext + c * 1024u, lane * 16u, dst + c * 1024u);
        gnext += chunks * 1024u;
        since = 0;
    }
    __device__ __forceinline__ const char* begin(int next_chunks) {
        wait_vmcnt_le(since);
        __builtin_amdgcn_s_barrier();
        issue(par ^ 1, next_chunks);
        const char* cur = lds + par * slot_bytes;
        par ^= 1;
        return cur;
    }
};

template <int NA, int NB>
__device__ __forceinline__ f32x16 bmma_tile(const char* slot, int lane, const bf16x8* inA,
                                            const bf16x8* inB) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0.0f;
    const char* ap = slot + lane * 16;
#pragma unroll
    for (int ks = 0; ks < NA; ks++) {
        const bf16x8 a = *(const bf16x8*)(ap + ks * 1024);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, inA[ks], acc, 0, 0, 0);
    }
#pragma unroll
    for (int ks = 0; ks < NB; ks++) {
        const bf16x8 a = *(const bf16x8*)(ap + (NA + ks) * 1024);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, inB[ks], acc, 0, 0, 0);
    }
    return acc;
}

// two tiles with interleaved, independent accumulators, fragments through the register ring (mma_pair_ring, mlp_spec.h)
template <int NA, int NB>
__device__ __forceinline__ void bmma_tile2(const char* slot0, const char* slot1, int lane, const bf16x8* inA,
                                           const bf16x8* inB, f32x16& acc0, f32x16& acc1) {
#pragma unroll
    for (int r = 0; r < 16; r++) { acc0[r] = 0.0f; acc1[r] = 0.0f; }
    mma_pair_ring<NA, NB, BWD_LDS_RING>(slot0, slot1, lane, inA, inB, acc0, acc1, [](auto) {});
}

// gradient wrt the pre-activation: pass where the stashed post-ReLU activation is non-zero
// `word`: the pair's 32 ReLU flags as the forward filed them (pack_tile in mlp_fwd.hip): tile t of the pair,
// accumulator register r -> bit 8t + (r < 8 ? r/2 : 4 + (r-8)/2) + 16 (r & 1)
template <bool MASK>
__device__ __forceinline__ void bpack_tile(const f32x16& acc, unsigned word, int t, bf16x8& o0, bf16x8& o1) {
#pragma unroll
    for (int e = 0; e < 8; e++) {
        float v0 = acc[e], v1 = acc[8 + e];
        if (MASK) {     // v_bfe_i32 (bit -> 0 / ~0) + v_and_b32; asm: hipcc turns the C form into and + cmp + cndmask
            int m0, m1;
            if (t == 0) {
                asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(m0) : "v"(word), "n"(e / 2 + 16 * (e & 1)));
                asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(m1) : "v"(word), "n"(4 + e / 2 + 16 * (e & 1)));
            } else {
                asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(m0) : "v"(word), "n"(8 + e / 2 + 16 * (e & 1)));
                asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(m1) : "v"(word), "n"(12 + e / 2 + 16 * (e & 1)));
            }
            v0 = __int_as_float(__float_as_int(v0) & m0);
            v1 = __int_as_float(__float_as_int(v1) & m1);
        }
        o0[e] = (__bf16)v0;
        o1[e] = (__bf16)v1;
    }
}

__host__ __device__ constexpr int bgroup_tiles(int nmt, int ch, int slot) {
    int g = nmt < slot / ch ? nmt : slot / ch;
    return (g > 1) ? (g & ~1) : g;          // even, so tiles can be processed in pairs
}

// one backward stage: NMT tiles over the fwd stage's input features, buffered in tile groups
// (one barrier + one prefetch burst per group, see mlp_fwd.hip)
template <int SLOT, int NA, int NB, int NMT, bool MASK, int PREV_NMT, bool STORE_OUT = true>
__device__ __forceinline__ void run_bstage(BPipe& p, const bf16x8* inA, const bf16x8* inB, bf16x8* out,
                                           int next_stage_chunks, const char* mask_src,
                                           char* dz_dst, bool valid, const bf16x8* prev_out, char* prev_dst,
                                           int skip_chunks = 0, bool wrap = false) {
    // mask_src / dz_dst / prev_dst: wave-uniform byte pointers (+ lane*16 per lane), see mlp_fwd.hip
    constexpr int CH = NA + NB;
    constexpr int G = bgroup_tiles(NMT, CH, SLOT);
    uint4 mk = make_uint4(0u, 0u, 0u, 0u);
    if (MASK && valid) mk = *(const uint4*)(mask_src + p.lane * 16);   // 128 ReLU bits of this lane's sample
    const unsigned mw[4] = {mk.x, mk.y, mk.z, mk.w};
    static_assert(NMT % 2 == 0 && G % 2 == 0, "even tile counts");
    const char* slot = nullptr;
#pragma unroll
    for (int mo = 0; mo < NMT; mo += 2) {
        if (mo % G == 0) {
            const int rest = NMT - mo - G;
            if (rest <= 0) p.skip(skip_chunks);      // stages this variant does not run
            if (rest <= 0 && wrap) p.gnext = 0;       // next prefetch = first group of the next block
            slot = p.begin(rest > 0 ? (rest < G ? rest : G) * CH : next_stage_chunks);
        }
        if (valid) {          // stores trail the compute by one pair, also across stages (mlp_fwd.hip)
            if (mo > 0) {
                if (STORE_OUT) {      // false: the linear bottleneck's gradient, which no weight-gradient GEMM reads
#pragma unroll
                    for (int q = 4; q >= 1; q--) STREAM_STORE(dz_dst + (2 * mo - q) * 1024 + p.lane * 16, out[2 * mo - q]);
                    p.since += 4;
                }
            } else if (PREV_NMT > 0 && prev_dst != nullptr) {
#pragma unroll
                for (int q = 4; q >= 1; q--)
                    STREAM_STORE(prev_dst + (2 * PREV_NMT - q) * 1024 + p.lane * 16, prev_out[2 * PREV_NMT - q]);
                p.since += 4;
            }
        }
        f32x16 acc0, acc1;
        bmma_tile2<NA, NB>(slot + (mo % G) * CH * 1024, slot + (mo % G + 1) * CH * 1024, p.lane, inA, inB, acc0, acc1);
        const unsigned w = mw[mo >> 1];
        bpack_tile<MASK>(acc0, w, 0, out[2 * mo], out[2 * mo + 1]);
        bpack_tile<MASK>(acc1, w, 1, out[2 * mo + 2], out[2 * mo + 3]);
    }
}

// d(enc) stage (box-pose gradients): 2 output tiles = the 64 encoding features, fp32 accumulators
template <int NA>
__device__ __forceinline__ void run_enc_stage(BPipe& p, const bf16x8* in, f32x16* denc, bool add,
                                              int next_stage_chunks, bool wrap = false) {
    if (wrap) p.gnext = 0;
    const char* slot = p.begin(next_stage_chunks);           // both tiles arrive as one group
    f32x16 acc0, acc1;
    bmma_tile2<NA, 0>(slot, slot + NA * 1024, p.lane, in, nullptr, acc0, acc1);
#pragma unroll
    for (int r = 0; r < 16; r++) {
        denc[0][r] = add ? denc[0][r] + acc0[r] : acc0[r];
        denc[1][r] = add ? denc[1][r] + acc1[r] : acc1[r];
    }
}

// ---------------------------------------------------------------------------
// M-split backward for the object MLPs (W = 128): the counterpart of k_mlp_fwd_ms (mlp_fwd.hip).  A workgroup is 4 waves x
// 64 samples; wave w owns input-feature tile w of every stage, gradients are exchanged through LDS as the next stage's B
// fragments, the transposed weight tiles come straight from L2, one barrier per stage.  Same MFMA instruction, operands and
// k order per output as k_mlp_bwd<128, false>: dz / dz_out are BIT-identical.  (The box-pose variant -- d(enc) -- stays on
// k_mlp_bwd<128, true>.)
// ---------------------------------------------------------------------------
namespace msb {
using S = MlpSpec<128>;
using B_ = BwdSpec<128>;
constexpr int NT = 2;
constexpr int X_BYTES = NT * S::KW * 1024;
constexpr int OFF_X = 0;                               // two gradient fragment buffers [tile][k-step][lane][16 B]
constexpr int OFF_G = 2 * X_BYTES;                     // head-gradient fragments [tile][2: rgb, density][lane][16 B]
constexpr int LDS_BYTES = OFF_G + NT * 2 * 1024;
}  // namespace msb

// the per-level operands of one launch over SEVERAL levels (durf_obj_bwd_batch_levels: every level's d(raw) exists before the
// first backward launch -- stop_level_grad -- so the object backward of a step is ONE latency-bound round instead of one per level)
struct MsBwdLevels {
    const float* draw[DURF_MAX_LEVELS];
    const uint4* relu_mask[DURF_MAX_LEVELS];
    bf16x8* dz[DURF_MAX_LEVELS];
    bf16x8* dz_out[DURF_MAX_LEVELS];
    int n;
};

// arguments of the M-split object backward (k_mlp_bwd_ms; the object items of the mixed launch k_mlp_bwd<.., MIX>)
struct MsBwd {
    size_t rows; int N; MsBwdLevels lv; const int32_t* ray_idx; const int32_t* count; const char* wpack; BwdStrides bs; int nobj;
    int* ticket;         // mixed launch only (durf::next_ticket)
};
__device__ __forceinline__ size_t msb_pairs_of(const MsBwd& A, int k) {
    const size_t c = (size_t)as_global(A.count)[k] * (size_t)A.N;
    return ((c < A.rows ? c : A.rows) + 32 * msb::NT - 1) / (32 * msb::NT);
}
// item -> (level, object, pair) -- level-major: the items of one level are consecutive; false when item >= total * levels
__device__ __forceinline__ bool msb_item(const MsBwd& A, size_t total, size_t item, int& level_out, size_t& k_out, size_t& pair_out) {
    const bool ok = total > 0 && item < total * (size_t)A.lv.n;
    const size_t it = ok ? item : 0;
    const int level = total ? (int)(it / total) : 0;
    size_t k = 0, pair = it - (size_t)level * total;
    for (; k + 1 < (size_t)A.nobj; k++) {
        const size_t np = msb_pairs_of(A, (int)k);
        if (pair < np) break;
        pair -= np;
    }
    level_out = __builtin_amdgcn_readfirstlane(level);
    k_out = (size_t)__builtin_amdgcn_readfirstlane((unsigned)k);
    pair_out = pair;
    return ok;
}

// Timing probe of one backward item (variant builds only, -DDURF_MS_STAMPS: see the forward's in mlp_fwd.hip; tools/experiments/ms_stamps.py --bwd)
#if defined(DURF_MS_STAMPS)
__device__ unsigned long long g_msb_stamps[4096 * 32];
__device__ unsigned g_msb_n;
#define MSB_STAMP(i) do { if (msb_probe) g_msb_stamps[msb_slot * 32 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
extern "C" int durf_debug_msb_stamps(void* dst, int reset) {
    unsigned n = 0;
    if (hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_msb_n), sizeof(n)) != hipSuccess) return -1;
    if (dst && hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_msb_stamps), sizeof(unsigned long long) * 4096 * 32) != hipSuccess) return -1;
    if (reset) { const unsigned z = 0; if (hipMemcpyToSymbol(HIP_SYMBOL(g_msb_n), &z, sizeof(z)) != hipSuccess) return -1; }
    return (int)n;
}
#else
#define MSB_STAMP(i) do { } while (0)
#endif
// One (level, object, tile pair) item on FOUR waves and msb::LDS_BYTES of LDS at `smem` (every barrier inside is the workgroup's)
// (the level's operands are resolved by the caller: draw [B*N,4], relu_mask / dz / dz_out [K, ...] slabs of that level)
__device__ __forceinline__ void msb_bwd_pair(const MsBwd& A, const float* __restrict__ draw, const uint4* relu_mask_g, bf16x8* dz_g,
                                             bf16x8* dz_out_g, char* smem, int lane, int wave, bool live, size_t k, size_t pair) {
    using S = msb::S;
    using BS = msb::B_;
    constexpr int NT = msb::NT;
    const int n = lane & 31;
    const size_t ntile32 = A.rows >> 5;
    const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    char* const X0 = smem + msb::OFF_X;
    char* const G = smem + msb::OFF_G;
    const size_t rows = A.rows;
    const int N = A.N;
    draw = as_global(draw);
    const int32_t* __restrict__ ray_idx = as_global(A.ray_idx + k * A.bs.idx);
    const char* __restrict__ wpack = A.wpack + k * A.bs.wpack;
    const uint4* __restrict__ relu_mask = as_global((const uint4*)((const char*)relu_mask_g + k * A.bs.mask));
    bf16x8* __restrict__ dz = as_global((bf16x8*)((char*)dz_g + k * A.bs.dz));
    bf16x8* __restrict__ dz_out = as_global((bf16x8*)((char*)dz_out_g + k * A.bs.dz_out));
    const size_t c = (size_t)as_global(A.count)[k] * (size_t)N;
    const size_t nrows = c < rows ? c : rows;
    const size_t t32[NT] = {pair * NT, pair * NT + 1};
    // (live == false: the other group of a mixed workgroup still has an item -- same stages on zeros, every global access off)
    const bool tv[NT] = {live, live && t32[1] * 32 < nrows};
#if defined(DURF_MS_STAMPS)
    const bool msb_probe = live && wave == 0 && lane == 0;
    unsigned msb_slot = 0;
    if (msb_probe) {
        msb_slot = atomicAdd(&g_msb_n, 1u) & 4095u;
        g_msb_stamps[msb_slot * 32 + 30] = ((unsigned long long)blockIdx.x << 32) | (unsigned long long)(k * 100000 + pair);
    }
#endif
    MSB_STAMP(0);
    ms_barrier();
    MSB_STAMP(1);
    // head gradients (fp32 [*,4]: d raw_rgb[3], d raw_density), gathered by ray: waves 0 / 1 build tile 0 / 1's fragments
    if (wave < NT) {
        const int t = wave;
        f32x4 dr = {0.f, 0.f, 0.f, 0.f};
        if (tv[t] && lane < 32) {
            const size_t row = t32[t] * 32 + n;
            const size_t src = (size_t)ray_idx[row / (size_t)N] * (size_t)N + row % (size_t)N;
            dr = *(const f32x4*)(draw + src * 4);
        }
        bf16x8 g10 = zero8, gd = zero8, gout = zero8;
        if (lane < 32) {
            g10[0] = (__bf16)dr[0]; g10[1] = (__bf16)dr[1]; g10[2] = (__bf16)dr[2];
            gd[0] = (__bf16)dr[3];
            gout = g10; gout[3] = gd[0];
        }
        *(bf16x8*)(G + (t * 2 + 0) * 1024 + lane * 16) = g10;
        *(bf16x8*)(G + (t * 2 + 1) * 1024 + lane * 16) = gd;
        if (tv[t]) *(DURF_G(bf16x8)*)(dz_out + t32[t] * 64 + lane) = gout;         // [rows,16] tile: slots 0-2 rgb, 3 density
    }
    f32x16 acc[NT];
    // Weights of (backward stage b, this wave's tile): requested ONE STAGE AHEAD into one of two alternating register sets
    struct WSet { bf16x8 A[S::KW + 1]; };
    auto load_w = [&](auto b_, WSet& w) {
        constexpr int b = decltype(b_)::value, T = BS::n_ks(b);
        typedef const __attribute__((address_space(1))) char* gptr_t;        // (global by type: see k_mlp_fwd_ms)
        gptr_t wt = (gptr_t)(wpack + (size_t)(BS::chunk_base(b) + wave * T) * 1024);
        asm volatile("" : "+s"(wt));               // (see k_mlp_fwd_ms: keeps later stages' weight loads below the barriers)
#pragma unroll
        for (int k = 0; k < T; k++) w.A[k] = *(const __attribute__((address_space(1))) bf16x8*)(wt + k * 1024 + lane * 16);
    };
    // this wave's tile of backward stage b: NX k-steps from Xin, then NG head-gradient fragments (g: 0 rgb, 1 density)
    unsigned mword[NT] = {0u, 0u};                 // this stage's ReLU flags: requested before its MFMAs AND before the next stage's
                                                   // weights (loads return in order: behind the 9 KB of weights the epilogue waited for those too)
    auto load_mask = [&](int jm) {
#pragma unroll
        for (int t = 0; t < NT; t++)
            mword[t] = tv[t] ? ((const unsigned*)((const char*)relu_mask + ((size_t)jm * ntile32 + t32[t]) * 1024 + lane * 16))[wave >> 1] : 0u;
    };
    auto stage_mma = [&](auto b_, const WSet& w, auto nx_, auto ng_, int g, const char* Xin) {
        constexpr int b = decltype(b_)::value, NX = decltype(nx_)::value, NG = decltype(ng_)::value, T = NX + NG;
        static_assert(T == BS::n_ks(b), "k-steps of the backward stage");
#pragma unroll
        for (int t = 0; t < NT; t++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[t][r] = 0.0f;
#if MS_RING > 0
        // (the B fragments through a ring of explicit LDS reads with counted waits, as k_mlp_fwd_ms: same fragments, same order)
        static_assert(NT == 2, "the ring alternates the two sample tiles");
        constexpr int NR = 2 * T, D = MS_RING < NR ? MS_RING : NR;
        const unsigned ax = lds_addr_of(Xin) + lane * 16, ag = lds_addr_of(G) + g * 1024 + lane * 16;
        auto rd = [&](auto m_) -> v4i_ {
            constexpr int m = decltype(m_)::value, k = m >> 1, t = m & 1;
            if constexpr (k < NX) return lds_read16<(t * S::KW + k) * 1024>(ax);
            else return lds_read16<t * 2048>(ag);
        };
        v4i_ ring[D];
        static_for<0, D>([&](auto i_) { ring[decltype(i_)::value] = rd(i_); });
        static_for<0, NR>([&](auto i_) {
            constexpr int i = decltype(i_)::value, k = i >> 1, t = i & 1;
            constexpr int later = (NR - 1 - i) < (D - 1) ? (NR - 1 - i) : (D - 1);
            lds_wait<later>(ring[i % D]);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.A[k], __builtin_bit_cast(bf16x8, ring[i % D]), acc[t], 0, 0, 0);
            if constexpr (i + D < NR) ring[i % D] = rd(std::integral_constant<int, i + D>{});
        });
#else
#pragma unroll
        for (int k = 0; k < T; k++) {
#pragma unroll
            for (int t = 0; t < NT; t++) {
                const char* src = k < NX ? Xin + (t * S::KW + k) * 1024 : G + (t * 2 + g) * 1024;
                const bf16x8 bv = *(const bf16x8*)(src + lane * 16);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.A[k], bv, acc[t], 0, 0, 0);
            }
        }
#endif
    };
    // mask with the forward's ReLU flags of stash region jm (this wave's tile: word wave / 2, parity wave % 2), hand the
    // fragments over, store dz region jd (jd < 0: not stored -- the linear bottleneck's gradient)
    auto hand_over = [&](auto mask_, int /*jm*/, int jd, char* Xout) {
        constexpr bool MASK = decltype(mask_)::value;
#pragma unroll
        for (int t = 0; t < NT; t++) {
            const unsigned word = MASK ? mword[t] : 0u;
            bf16x8 o0, o1;
            if (wave & 1) bpack_tile<MASK>(acc[t], word, 1, o0, o1);
            else bpack_tile<MASK>(acc[t], word, 0, o0, o1);
            *(bf16x8*)(Xout + (t * S::KW + 2 * wave) * 1024 + lane * 16) = o0;
            *(bf16x8*)(Xout + (t * S::KW + 2 * wave + 1) * 1024 + lane * 16) = o1;
            if (jd >= 0 && tv[t]) {
                char* dd = (char*)dz + ((size_t)S::stash_ks_before(jd) * ntile32 + t32[t] * S::stash_ks(jd)) * 1024;
                STREAM_STORE(dd + (2 * wave) * 1024 + lane * 16, o0);
                STREAM_STORE(dd + (2 * wave + 1) * 1024 + lane * 16, o1);
            }
        }
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using IW = std::integral_constant<int, S::KW>;
    char* Xa = X0;
    char* Xb = X0 + msb::X_BYTES;
    WSet w0, w1;
    load_w(std::integral_constant<int, 0>{}, w0);
    MSB_STAMP(2);
    ms_barrier();                               // head-gradient fragments in place
    MSB_STAMP(3);
    // bwd of stage 10 (rgb head): d rgb -> d A9, masked by A9 (mask region 8) -> dz region 9
    load_mask(8);
    load_w(std::integral_constant<int, 1>{}, w1);
    stage_mma(std::integral_constant<int, 0>{}, w0, I0{}, I1{}, 0, Xa);
    hand_over(std::true_type{}, 8, 9, Xa);
    MSB_STAMP(16);
    ms_barrier();
    MSB_STAMP(4);
    // bwd of stage 9 (view layer): d Z9 -> d bottleneck (linear; not stored)
    load_w(std::integral_constant<int, 2>{}, w0);
    stage_mma(std::integral_constant<int, 1>{}, w1, std::integral_constant<int, S::KC>{}, I0{}, 0, Xa);
    hand_over(std::false_type{}, 0, -1, Xb);
    MSB_STAMP(17);
    ms_barrier();
    MSB_STAMP(5);
    // bwd of stage 8 (bottleneck + density head): -> d A7 (mask region 7, dz region 7)
    load_mask(7);
    load_w(std::integral_constant<int, 3>{}, w1);
    stage_mma(std::integral_constant<int, 2>{}, w0, IW{}, I1{}, 1, Xb);
    hand_over(std::true_type{}, 7, 7, Xa);
    MSB_STAMP(18);
    ms_barrier();
    MSB_STAMP(6);
    // bwd of stages 7, 6, 5 (trunk rows) -> d Z6, d Z5, d Z4
    load_mask(6);
    load_w(std::integral_constant<int, 4>{}, w0);
    stage_mma(std::integral_constant<int, 3>{}, w1, IW{}, I0{}, 0, Xa);
    hand_over(std::true_type{}, 6, 6, Xb);
    MSB_STAMP(19);
    ms_barrier();
    MSB_STAMP(7);
    load_mask(5);
    load_w(std::integral_constant<int, 5>{}, w1);
    stage_mma(std::integral_constant<int, 4>{}, w0, IW{}, I0{}, 0, Xb);
    hand_over(std::true_type{}, 5, 5, Xa);
    MSB_STAMP(20);
    ms_barrier();
    MSB_STAMP(8);
    load_mask(4);
    load_w(std::integral_constant<int, 7>{}, w0);
    stage_mma(std::integral_constant<int, 5>{}, w1, IW{}, I0{}, 0, Xa);
    hand_over(std::true_type{}, 4, 4, Xb);
    MSB_STAMP(21);
    ms_barrier();
    MSB_STAMP(9);
    // bwd of stages 4..1 -> d Z3 .. d Z0
    load_mask(3);
    load_w(std::integral_constant<int, 8>{}, w1);
    stage_mma(std::integral_constant<int, 7>{}, w0, IW{}, I0{}, 0, Xb);
    hand_over(std::true_type{}, 3, 3, Xa);
    MSB_STAMP(22);
    ms_barrier();
    MSB_STAMP(10);
    load_mask(2);
    load_w(std::integral_constant<int, 9>{}, w0);
    stage_mma(std::integral_constant<int, 8>{}, w1, IW{}, I0{}, 0, Xa);
    hand_over(std::true_type{}, 2, 2, Xb);
    MSB_STAMP(23);
    ms_barrier();
    MSB_STAMP(11);
    load_mask(1);
    load_w(std::integral_constant<int, 10>{}, w1);
    stage_mma(std::integral_constant<int, 9>{}, w0, IW{}, I0{}, 0, Xb);
    hand_over(std::true_type{}, 1, 1, Xa);
    MSB_STAMP(24);
    ms_barrier();
    MSB_STAMP(12);
    load_mask(0);
    stage_mma(std::integral_constant<int, 10>{}, w1, IW{}, I0{}, 0, Xa);
    hand_over(std::true_type{}, 0, 0, Xb);
    MSB_STAMP(15);
}

__global__ void __launch_bounds__(256)
k_mlp_bwd_ms(MsBwd A) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // (object, pair) items dealt to a 1-D grid by the device-side hit counts: see k_mlp_fwd_ms
    size_t total = 0;
    for (int k = 0; k < A.nobj; k++) total += msb_pairs_of(A, k);
    for (size_t item = blockIdx.x; item < total * (size_t)A.lv.n; item += gridDim.x) {
        int level; size_t k, pair;
        msb_item(A, total, item, level, k, pair);
        msb_bwd_pair(A, A.lv.draw[level], A.lv.relu_mask[level], A.lv.dz[level], A.lv.dz_out[level], smem, lane, wave, true, k, pair);
    }
}

// (the kernel's explicit arguments as the kernarg segment lays them out; see FwdKernArgs in mlp_fwd.hip)
struct BwdKernArgs {
    size_t rows; int N; const float* draw; const int32_t* ray_idx; const int32_t* count; const char* wpack; const uint4* relu_mask;
    bf16x8* dz; bf16x8* dz_out; float* d_enc; BwdStrides bs; const int32_t* tail_idx; const int32_t* tail_count;
    const float* draw_ray_sum; MsBwd ow;
};
// The object phase of a mixed backward workgroup (k_mlp_bwd<.., MIX>): two groups of four waves take (level, object, tile
// pair) items off the ticket counter.  Inlined, arguments from the kernarg segment: see mix_object_items in mlp_fwd.hip.
__device__ __forceinline__ void mix_object_items_bwd(char* smem, int wave, int nwg) {
    typedef const __attribute__((address_space(4))) char* kptr_t;
    kptr_t ka = (kptr_t)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(ka));          // (opaque: the loads below stay below the background loop)
    const unsigned smem_lds = lds_addr_of(smem);
    MsBwd ow;
    load_kernarg(ow, ka + offsetof(BwdKernArgs, ow));
    const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    // (the ticket requests are wave 1's: wave 0 starts every item with the ray loads of the encoding / the head gradients, and
    // a returning atomic ahead of them in its queue would be waited for with them)
    const bool first = wave == MIX_TICKET_WAVE && lane == 0;
    // (waves w and w + 4 share a SIMD: the second group's roles are rotated by two, so that the two groups' role-0 waves --
    // the serial head of an item: the encoding / the head gradients -- run on different SIMDs)
    const int half = wave >> 2, w4 = (wave + MIX_ROLE_ROT * half) & 3;
    char* const lds = smem + half * msb::LDS_BYTES;
    volatile __attribute__((address_space(3))) int* const tk =
        (volatile __attribute__((address_space(3))) int*)(size_t)(__builtin_amdgcn_readfirstlane(smem_lds) + 2u * msb::LDS_BYTES);
    volatile __attribute__((address_space(3))) int* const npl = tk + 4;         // the objects' pair counts, once per workgroup
    if (wave == 0 && lane < ow.nobj) npl[lane] = (int)msb_pairs_of(ow, lane);
    ms_barrier();
    size_t total = 0;
    for (int k = 0; k < ow.nobj; k++) total += (size_t)npl[k];
    total = (size_t)__builtin_amdgcn_readfirstlane((unsigned)total);
    const size_t items = total * (size_t)ow.lv.n;
    const int last = 2 * (int)((items + 1) / 2 + nwg - 1);            // the value the LAST request of the launch returns
    int t = 0;
    // (a GLOBAL atomic: a flat one counts on lgkmcnt, and the first barrier of the item would wait for the request under way)
    DURF_G(int)* const ticket = (DURF_G(int)*)ow.ticket;
    if (first) { t = __hip_atomic_fetch_add(ticket, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); *tk = t; }
    ms_barrier();
    t = __builtin_amdgcn_readfirstlane(*tk);
    while ((size_t)t < items) {
        int tn = 0;
        if (first) tn = __hip_atomic_fetch_add(ticket, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // the next request is under way while this item runs
        const size_t item = (size_t)t + (size_t)half;
        const bool live = item < items;
        const int level = __builtin_amdgcn_readfirstlane(live ? (int)(item / total) : 0);
        size_t k = 0, pair = live ? item - (size_t)level * total : 0;
        for (; live && k + 1 < (size_t)ow.nobj; k++) {
            const size_t np = (size_t)npl[k];
            if (pair < np) break;
            pair -= np;
        }
        k = (size_t)__builtin_amdgcn_readfirstlane((unsigned)(live ? k : 0));
        // (this level's operands straight from the kernarg segment: a dynamically indexed copy would live in scratch)
        const __attribute__((address_space(4))) MsBwd* kp = (const __attribute__((address_space(4))) MsBwd*)(ka + offsetof(BwdKernArgs, ow));
        const float* draw = kp->lv.draw[level];
        const uint4* mk = kp->lv.relu_mask[level];
        bf16x8* dzl = kp->lv.dz[level];
        bf16x8* dzo = kp->lv.dz_out[level];
        msb_bwd_pair(ow, draw, mk, dzl, dzo, lds, lane, w4, live, k, pair);
        if (first) *tk = tn;
        ms_barrier();
        t = __builtin_amdgcn_readfirstlane(*tk);
    }
    if (first && t == last) __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // every other workgroup has made its last request
}

// MIX (W = 256, 8 waves; round 6): the background blocks, then the object MLPs' backward items on two 4-wave groups per
// workgroup -- the counterpart of k_mlp_fwd<.., MIX> (mlp_fwd.hip), dz / dz_out of both classes bit-identical to the launches
// of their own.
template <int W, bool POSE, int NWV = 8, bool MIX = false>
__global__ void __launch_bounds__(512, 2)
k_mlp_bwd(size_t rows, int N, const float* __restrict__ draw, const int32_t* __restrict__ ray_idx,
          const int32_t* __restrict__ count, const char* __restrict__ wpack,
          const uint4* __restrict__ relu_mask, bf16x8* __restrict__ dz, bf16x8* __restrict__ dz_out,
          float* __restrict__ d_enc, BwdStrides bs, const int32_t* __restrict__ tail_idx,
          const int32_t* __restrict__ tail_count, const float* __restrict__ draw_ray_sum, MsBwd ow_arg) {
    static_assert(!MIX || (W == 256 && NWV == 8 && !POSE), "the mixed launch: background blocks of 8 waves + object items on 2 x 4");
    (void)ow_arg;      // (read from the kernarg segment behind the background loop: mix_object_items_bwd)
    using S = MlpSpec<W>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (gridDim.y > 1) {                             // batched object MLPs: this workgroup's object slab
        const size_t k = blockIdx.y;
        ray_idx += k * bs.idx;
        count += k;
        wpack += k * bs.wpack;
        relu_mask = (const uint4*)((const char*)relu_mask + k * bs.mask);
        dz = (bf16x8*)((char*)dz + k * bs.dz);
        dz_out = (bf16x8*)((char*)dz_out + k * bs.dz_out);
        if (POSE) d_enc = (float*)((char*)d_enc + k * bs.d_enc);
    }
    size_t nrows = rows;
    if (count) {
        const size_t c = (size_t)(*count) * (size_t)N;
        nrows = c < rows ? c : rows;
    }
    const size_t nrows_c = nrows;                     // tail rows: one per box-hit ray (see k_mlp_fwd)
    if (tail_count) {
        const size_t t = nrows_c + (size_t)(*tail_count);
        nrows = t < rows ? t : rows;
    }
    const bool has_block = (size_t)blockIdx.x * (32 * NWV) < nrows;
    if (!MIX && !has_block) return;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (wave >= 4 && has_block) __builtin_amdgcn_s_setprio(1);     // stagger SIMD partners (see mlp_fwd.hip)
    const size_t ntile32 = rows >> 5;
    const size_t nblk = (nrows + 32 * NWV - 1) / (32 * NWV);

    BPipe p;
    constexpr int SLOT = 4 * (S::KW + 1);
    p.rsrc = make_rsrc(wpack);
    p.gnext = 0; p.lds = smem; p.slot_bytes = SLOT * 1024; p.par = 0;
    p.lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    p.wave = wave; p.lane = lane; p.nw = NWV;
    constexpr int GB0 = bgroup_tiles(S::CT, 1, SLOT) * 1;                 // all tiles of the rgb-head stage
    if (!MIX || has_block) p.issue(0, GB0);

  // persistent workgroup (see mlp_fwd.hip): loop over this CU's 256-sample blocks
  for (size_t blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
    const bool has_next = blk + gridDim.x < nblk;
    const size_t tile32 = blk * NWV + wave;
    const size_t row = tile32 * 32 + (lane & 31);
    const bool valid = row < nrows;
    const bool tile_valid = tile32 * 32 < nrows;

    // head gradients (fp32 [*,4]: d raw_rgb[3], d raw_density); object MLPs gather by ray
    f32x4 dr = {0.f, 0.f, 0.f, 0.f};
    if (valid) {
        if (tail_count && row >= nrows_c) {           // the single evaluation of a box-hit ray: the ray's summed head gradient
            dr = *(const f32x4*)(draw_ray_sum + (size_t)tail_idx[row - nrows_c] * 4);
        } else {
            size_t src = row;
            if (ray_idx) src = (size_t)ray_idx[row / (size_t)N] * (size_t)N + row % (size_t)N;
            dr = *(const f32x4*)(draw + src * 4);
        }
    }
    const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    const bool lo = lane < 32;
    bf16x8 g10[1], gd[1], gout = zero8;
    g10[0] = zero8; gd[0] = zero8;
    if (lo) {
        g10[0][0] = (__bf16)dr[0]; g10[0][1] = (__bf16)dr[1]; g10[0][2] = (__bf16)dr[2];
        gd[0][0] = (__bf16)dr[3];
        gout = g10[0]; gout[3] = gd[0][0];
    }
    if (tile_valid) dz_out[tile32 * 64 + lane] = gout;      // [rows,16] tile: slots 0-2 rgb, 3 density

    auto stash_at = [&](int j) -> const char* {      // ReLU bit-mask region: stage j (9 -> region 8); wave-uniform
        return (const char*)relu_mask + ((size_t)(j == 9 ? 8 : j) * ntile32 + tile32) * 1024;
    };
    auto dz_at = [&](int j) -> char* {
        return (char*)dz + ((size_t)S::stash_ks_before(j) * ntile32 + tile32 * S::stash_ks(j)) * 1024;
    };
    bf16x8 a[S::KW], b[S::KW], c[S::KC];
    // bwd of stage 10 (rgb head): d rgb -> d A9, masked by A9
    constexpr int GC = bgroup_tiles(S::WT, S::KC, SLOT) * S::KC;           // first group of bwd stage 9
    constexpr int G8 = bgroup_tiles(S::WT, S::KW + 1, SLOT) * (S::KW + 1);
    constexpr int GW = bgroup_tiles(S::WT, S::KW, SLOT) * S::KW;
    constexpr int GE = 2 * S::KW;                                             // d(enc) stage: 2 tiles, one group
    run_bstage<SLOT, 1, 0, S::CT, true, 0>(p, g10, nullptr, c, GC, stash_at(9), dz_at(9), tile_valid, nullptr, nullptr);
    // bwd of stage 9 (view layer): d Z9 -> d bottleneck (linear)
    run_bstage<SLOT, S::KC, 0, S::WT, false, S::CT, false>(p, c, nullptr, a, G8, nullptr, nullptr, tile_valid, c, dz_at(9));
    // bwd of stage 8 (bottleneck + density head): -> d A7
    run_bstage<SLOT, S::KW, 1, S::WT, true, S::WT>(p, a, gd, b, GW, stash_at(7), dz_at(7), tile_valid, a, nullptr);
    // bwd of stages 7, 6 -> d Z6, d Z5
    run_bstage<SLOT, S::KW, 0, S::WT, true, S::WT>(p, b, nullptr, a, GW, stash_at(6), dz_at(6), tile_valid, b, dz_at(7));
    run_bstage<SLOT, S::KW, 0, S::WT, true, S::WT>(p, a, nullptr, b, GW, stash_at(5), dz_at(5), tile_valid, a, dz_at(6));
    // bwd of stage 5: trunk rows -> d Z4 (in a); encoding rows (skip connection) only for POSE
    f32x16 denc[2];
    run_bstage<SLOT, S::KW, 0, S::WT, true, S::WT>(p, b, nullptr, a, POSE ? GE : GW, stash_at(4), dz_at(4), tile_valid,
                                                   b, dz_at(5), POSE ? 0 : GE);
    if (POSE) run_enc_stage<S::KW>(p, b, denc, false, GW);
    // bwd of stages 4..1
    run_bstage<SLOT, S::KW, 0, S::WT, true, S::WT>(p, a, nullptr, b, GW, stash_at(3), dz_at(3), tile_valid, a, dz_at(4));
    run_bstage<SLOT, S::KW, 0, S::WT, true, S::WT>(p, b, nullptr, a, GW, stash_at(2), dz_at(2), tile_valid, b, dz_at(3));
    run_bstage<SLOT, S::KW, 0, S::WT, true, S::WT>(p, a, nullptr, b, GW, stash_at(1), dz_at(1), tile_valid, a, dz_at(2));
    run_bstage<SLOT, S::KW, 0, S::WT, true, S::WT>(p, b, nullptr, a, POSE ? GE : (has_next ? GB0 : 0), stash_at(0),
                                                   dz_at(0), tile_valid, b, dz_at(1), 0, !POSE);
    if (tile_valid) {          // last stage's trailing stores
#pragma unroll
        for (int q = 4; q >= 1; q--) STREAM_STORE(dz_at(0) + (2 * S::WT - q) * 1024 + lane * 16, a[2 * S::WT - q]);
    }
    if (POSE) {
        // Dense_0 -> d(encoding); total d enc = skip-connection part + first-layer part
        run_enc_stage<S::KW>(p, a, denc, true, has_next ? GB0 : 0, true);
        if (valid) {
            const int hi = lane >> 5;
            float* dst = d_enc + row * DURF_ENC_DIM;
#pragma unroll
            for (int mo = 0; mo < 2; mo++)
#pragma unroll
                for (int g4 = 0; g4 < 4; g4++) {
                    const f32x4 o = {denc[mo][4 * g4], denc[mo][4 * g4 + 1], denc[mo][4 * g4 + 2], denc[mo][4 * g4 + 3]};
                    *(f32x4*)(dst + 32 * mo + 8 * g4 + 4 * hi) = o;
                }
        }
    }
  }
  if constexpr (MIX) {
    if (has_block) __builtin_amdgcn_s_setprio(0);
    ms_barrier();                  // every wave is past its last weight read; no DMA is in flight (the last block prefetches none)
    mix_object_items_bwd(smem, wave, (int)gridDim.x);
  }
}

// view-direction features expanded per sample into tile layout [rows, 32] (dW of Dense_10)
__global__ void __launch_bounds__(256)
k_expand_view(size_t rows, int N, const bf16x8* __restrict__ view, const int32_t* __restrict__ ray_idx,
              const int32_t* __restrict__ count, bf16x8* __restrict__ out, size_t idx_stride, size_t out_stride,
              const int32_t* __restrict__ tail_idx, const int32_t* __restrict__ tail_count) {
    if (gridDim.y > 1) {
        ray_idx += blockIdx.y * idx_stride;
        count += blockIdx.y;
        out = (bf16x8*)((char*)out + blockIdx.y * out_stride);
    }
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;     // one 16-byte vector each
    const size_t row = gid >> 2;
    const int q = (int)(gid & 3);
    size_t nrows = rows;
    if (count) { const size_t c = (size_t)(*count) * N; nrows = c < rows ? c : rows; }
    const size_t nrows_c = nrows;                     // tail rows: one per box-hit ray (see k_mlp_fwd)
    if (tail_count) {
        const size_t t = nrows_c + (size_t)(*tail_count);
        nrows = t < rows ? t : rows;
        // rows of the partial last tile beyond the tail: zeros (dW reads whole 32-row tiles; their dz is zero)
        if (row >= nrows && row < ((nrows + 31) & ~(size_t)31)) {
            const bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
            *(bf16x8*)((char*)out + tile_vec_offset(row, q, 2)) = z;
        }
    }
    if (row >= nrows) return;
    size_t ray = row / (size_t)N;
    if (tail_count && row >= nrows_c) ray = (size_t)tail_idx[row - nrows_c];
    else if (ray_idx) ray = (size_t)ray_idx[ray];
    *(bf16x8*)((char*)out + tile_vec_offset(row, q, 2)) = view[ray * 4 + q];
}

// ---------------------------------------------------------------------------
// weight-gradient GEMM
// ---------------------------------------------------------------------------
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ bf16x8 tr_frag(const char* base, int off0, int off1) {
    const s16x4 r0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + off0));
    const s16x4 r1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + off1));
    const s16x8 r = {r0[0], r0[1], r0[2], r0[3], r1[0], r1[1], r1[2], r1[3]};
    return __builtin_bit_cast(bf16x8, r);
}

// NKO: k-steps (16 features) of dZ; NKA/NKB: k-steps of the two input segments.
// The kernel streams 1 KB per sample per 256x256 layer from HBM at an arithmetic intensity of
// 128 FLOP/B, i.e. it is HBM-bound by construction; what matters is bytes in flight per CU.
// An LDS ring of dw_stages(NC) stages (9-36 KB each) is filled by buffer_load...lds with counted
// s_waitcnt vmcnt(N) and raw s_barrier, so 2-3 stages (~100 KB) are always in flight per CU.
// Ring depth: 4 stages of 32 samples (measured on MI355X: 2 stages 1336 us, 3: 1209, 4: 1148,
// 5: 1227, 6+: ~1190 per launch at cfg2 -- deeper rings do not buy bandwidth here).
#ifndef DW_LDS_KB
#define DW_LDS_KB 144
#endif
#ifndef DW_MAX_STAGES
#define DW_MAX_STAGES 4
#endif
#define DW_LDS_BYTES (DW_LDS_KB * 1024)
__host__ __device__ constexpr int dw_stages(int nc) { return DW_LDS_BYTES / (nc * 1024) > DW_MAX_STAGES ? DW_MAX_STAGES : DW_LDS_BYTES / (nc * 1024); }

// LDS-DMA issued from inline asm: hipcc orders every later ds_read behind ALL outstanding
// buffer_load...lds it knows about (s_waitcnt vmcnt(0)), which would drain the ring on every
// tile.  Hidden in asm, the loads are ordered only by this kernel's own counted waits.
// (M0 = LDS byte address of the 1 KB chunk; saved/restored because hipcc owns M0.)
// nt: the operands are read exactly once (measured -6 % per launch vs the default cache policy)
#define DW_DMA_POLICY " nt"
__device__ __forceinline__ void lds_dma16(i32x4 rsrc, unsigned soff, unsigned voff, unsigned lds_addr) {
    unsigned keep;
    asm volatile("s_nop 4\n\t"
                 "s_mov_b32 %0, m0\n\t"
                 "s_mov_b32 m0, %1\n\t"
                 "s_nop 0\n\t"
                 "buffer_load_dwordx4 %2, %3, %4 offen" DW_DMA_POLICY " lds\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff)
                 : "memory");
}

// One launch for the 12 weight-gradient GEMMs of an MLP: blockIdx.y = job (= flax Dense index),
// blockIdx.x = split within the job.  Splits are allotted in proportion to each job's bytes per
// sample so that ~4 rounds of workgroups smooth the tail, with ~3x fewer split-K partials than one
// 256-way launch per job.
struct DwArgs {
    const char* dz[DURF_MAX_LEVELS][12];      // per level (the sample axis of the GEMMs runs over all levels)
    const char* inA[DURF_MAX_LEVELS][12];
    const char* inB[DURF_MAX_LEVELS][12];
    int nlevels;
    const char* dz2[DURF_MAX_LEVELS];          // job 10 only: the [rows,16] head-gradient tile, one more dz k-step (density head)
    size_t ks_dz2;
    size_t ks_dz[12], ks_a[12], ks_b[12];      // per-object strides of the three operand streams (batched object MLPs)
    size_t ks_part, ks_bpart;                  // ... and of the partial buffers (floats)
    float* part[12];
    float* bpart[12];
    int nsplit[12];
    int first_wg[13];        // workgroup ids [first_wg[i], first_wg[i+1]) run job order[i]
    int order[12];
};

// Tiles per split: an even share, but at least DW_MIN_TPS, so that a sparsely hit object (a few hundred
// valid tiles) uses a few dozen splits instead of ~100 per GEMM -- the unused workgroups exit at once and
// neither write nor get summed (k_dw_finalize derives the same count from the device-side ray count).
#ifndef DW_MIN_TPS
#define DW_MIN_TPS 48
#endif
// A launch with few tiles altogether (the objects of a 512-ray batch: ~150 tiles) is bound by the tiles a workgroup walks IN
// SEQUENCE (~1 us each), not by its partials: the floor then drops to 1/16 of the tiles, at least 8 (k_dw_all<128> 56 -> 2x us
// at cfg3's gin-literal 512 rays; unchanged from 768 tiles per MLP up, i.e. at 4096 rays).
__host__ __device__ inline size_t dw_tiles_per_split(size_t nt_all, int nsplit) {
    const size_t even = (nt_all + nsplit - 1) / nsplit;
    size_t floor_ = DW_MIN_TPS;
    const size_t small = (nt_all + 15) / 16;
    if (small < floor_) floor_ = small < 8 ? 8 : small;
    return even > floor_ ? even : floor_;
}

// valid 32-sample tiles of level l (object `obj` of a batched launch): the last tile of a level whose row count is
// not a multiple of 32 (one sample per ray) is partial -- its rows beyond the count carry zero dz
__host__ __device__ inline size_t dw_level_tiles(const durf::DwLevels& lv, int l, size_t obj) {
    size_t nrows = lv.rows[l];
#if defined(__HIP_DEVICE_COMPILE__)
    if (lv.count[l]) { const size_t c = (size_t)lv.count[l][obj] * (size_t)lv.n[l]; nrows = c < nrows ? c : nrows; }
#endif
    return (nrows + 31) >> 5;
}

// NKO2 of the NKO dz k-steps come from a second buffer (a.dz2: the head-gradient tile) after the NKO - NKO2 of a.dz
template <int NKO, int NKA, int NKB, int NKO2 = 0>
__device__ __forceinline__ void dw_job(const durf::DwLevels& lv, const DwArgs& a, int job,
                                       int nsplit, int split_idx, float* __restrict__ part,
                                       float* __restrict__ bpart, char* smem) {
    constexpr int NKI = NKA + NKB;
    constexpr int NC = NKO + NKI;                 // 1 KB chunks per 32-sample stage
    constexpr int CPW = (NC + 7) / 8;             // LDS-DMA instructions per wave per stage
    constexpr int MO = (NKO + 1) / 2, NI = NKI / 2;
    constexpr int RM = (MO + 3) / 4, RN = (NI + 1) / 2;
    constexpr int STAGE = NC * 1024;
    constexpr int S = dw_stages(NC);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const size_t obj = blockIdx.y;                            // batched object MLPs (0 for a single MLP)
    if (gridDim.y > 1) { part += obj * a.ks_part; bpart += obj * a.ks_bpart; }
    size_t nt_all = 0;                                        // the K axis: every level's valid 32-sample tiles
    for (int l = 0; l < a.nlevels; l++) nt_all += dw_level_tiles(lv, l, obj);
    const size_t tps = dw_tiles_per_split(nt_all, nsplit);    // even share of the VALID tiles
    const size_t g0 = (size_t)split_idx * tps;
    if (g0 >= nt_all) return;                                 // unused split: k_dw_finalize skips it too
    size_t g1 = g0 + tps;
    if (g1 > nt_all) g1 = nt_all;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    // LDS-DMA source swizzle: LDS vector position p of half hi_f holds sample n with
    //   p = (n & 16) | ((n + 4*(2*(ks&1) + hi_f)) & 15)   -> conflict-free tr-reads
    const int g_hif = lane >> 5, g_p = lane & 31;
    unsigned voff[2];
#pragma unroll
    for (int par = 0; par < 2; par++) {
        const int c = 2 * par + g_hif;
        const int n = (g_p & 16) | ((g_p - 4 * c) & 15);
        voff[par] = (unsigned)((g_hif * 32 + n) * 16);
    }
    constexpr int NKO1 = NKO - NKO2;
    i32x4 r_dz, r_a, r_b, r_dz2;                      // descriptors based at the current segment's first tile
    auto stage_load = [&](int ti, int slot) {        // ti = tile index relative to t0
        ti = __builtin_amdgcn_readfirstlane(ti);
        const unsigned dst = lds0 + (unsigned)(slot * STAGE);
#pragma unroll
        for (int i = 0; i < CPW; i++) {
            int ci = wave + 8 * i;
            if (ci >= NC) ci -= NC;                    // pad: re-load an early chunk (same bytes, same place)
            if (ci < NKO1) {
                lds_dma16(r_dz, (unsigned)((ti * NKO1 + ci) * 1024), voff[ci & 1], dst + ci * 1024);
            } else if (NKO2 > 0 && ci < NKO) {
                lds_dma16(r_dz2, (unsigned)((ti * NKO2 + ci - NKO1) * 1024), voff[ci & 1], dst + ci * 1024);
            } else if (ci < NKO + NKA) {
                const int ks = ci - NKO;
                lds_dma16(r_a, (unsigned)((ti * NKA + ks) * 1024), voff[ks & 1], dst + ci * 1024);
            } else {
                const int ks = ci - NKO - NKA;
                lds_dma16(r_b, (unsigned)((ti * NKB + ks) * 1024), voff[(ks + NKA) & 1], dst + ci * 1024);
            }
        }
    };
    // per-lane tr-read geometry (see header comment): lane l -> group g, provider index L
    const int g = lane >> 4, L = lane & 15;
    const int ksp = g & 1, hi = g >> 1, j = L >> 2, q = L & 3, hi_f = q >> 1, half = q & 1;
    const int cc = 2 * ksp + hi_f;
    const int lane_off = hi_f * 512 + half * 8;
    const int poff0 = ((8 * hi + 0 + j + 4 * cc) & 15) * 16;
    const int poff1 = ((8 * hi + 4 + j + 4 * cc) & 15) * 16;

    f32x16 acc[RM][RN];
#pragma unroll
    for (int rm = 0; rm < RM; rm++)
#pragma unroll
        for (int rn = 0; rn < RN; rn++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[rm][rn][r] = 0.0f;
    float bsum[RM];
#pragma unroll
    for (int rm = 0; rm < RM; rm++) bsum[rm] = 0.0f;

    // The samples of every level form one K axis; a split that straddles a level boundary runs one
    // segment per level (the ring drains and refills at the boundary, the accumulators carry on).
    size_t lo = 0, hi_t = 0;
    for (int lvl = 0; lvl < a.nlevels; lvl++) {
        lo = hi_t;
        hi_t = lo + dw_level_tiles(lv, lvl, obj);
        const size_t b0 = g0 > lo ? g0 : lo, b1 = g1 < hi_t ? g1 : hi_t;
        if (b1 <= b0) continue;
        const size_t t0 = b0 - lo;                    // first tile of the segment within its level
        const int nt = (int)(b1 - b0);
        r_dz = make_rsrc(a.dz[lvl][job] + obj * a.ks_dz[job] + t0 * NKO1 * 1024);
        r_dz2 = make_rsrc(NKO2 ? a.dz2[lvl] + obj * a.ks_dz2 + t0 * (NKO2 ? NKO2 : 1) * 1024 : a.dz[lvl][job]);
        r_a = make_rsrc(a.inA[lvl][job] + obj * a.ks_a[job] + t0 * NKA * 1024);
        r_b = make_rsrc(NKB ? a.inB[lvl][job] + obj * a.ks_b[job] + t0 * (NKB ? NKB : 1) * 1024 : a.inA[lvl][job]);
        __builtin_amdgcn_s_barrier();                 // everyone is done reading the previous segment's slots
        // prologue: S-1 stages in flight
#pragma unroll
        for (int i = 0; i < S - 1; i++)
            if (i < nt) stage_load(i, i);
        for (int t = 0; t < nt; t++) {
            // this wave's part of tile t has landed when at most min(S-2, nt-1-t) later stages are pending
            const int later = nt - 1 - t;
            if (later >= S - 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((S - 2) * CPW) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // tail: everything was issued, drain once
            __builtin_amdgcn_s_barrier();       // every wave's part landed; everyone is done with tile t-1's slot
            if (t + S - 1 < nt) stage_load(t + S - 1, (t + S - 1) % S);
            const char* st = smem + (t % S) * STAGE;
#pragma unroll
            for (int kk = 0; kk < 2; kk++) {
                bf16x8 af[RM], bf[RN];
#pragma unroll
                for (int rm = 0; rm < RM; rm++) {
                    const int mo = wm + 4 * rm;
                    if (mo < MO) {
                        int ks = 2 * mo + ksp;
                        if (ks > NKO - 1) ks = NKO - 1;            // odd NKO: duplicate, ignored later
                        const char* base = st + ks * 1024 + lane_off + kk * 256;
                        af[rm] = tr_frag(base, poff0, poff1);
                    }
                }
#pragma unroll
                for (int rn = 0; rn < RN; rn++) {
                    const int ni = wn + 2 * rn;
                    if (ni < NI) {
                        const char* base = st + (NKO + 2 * ni + ksp) * 1024 + lane_off + kk * 256;
                        bf[rn] = tr_frag(base, poff0, poff1);
                    }
                }
#pragma unroll
                for (int rm = 0; rm < RM; rm++) {
                    if (wm + 4 * rm < MO) {
#pragma unroll
                        for (int rn = 0; rn < RN; rn++)
                            if (wn + 2 * rn < NI)
                                acc[rm][rn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[rm], bf[rn], acc[rm][rn], 0, 0, 0);
                        if (wn == 0) {
                            float s = 0.0f;
#pragma unroll
                            for (int e = 0; e < 8; e++) s += (float)af[rm][e];
                            bsum[rm] += s;
                        }
                    }
                }
            }
        }
    }
    // partials in fragment coordinates: [split][mo][ni][lane][16]
    const size_t sp = (size_t)split_idx;
#pragma unroll
    for (int rm = 0; rm < RM; rm++) {
        const int mo = wm + 4 * rm;
        if (mo < MO) {
#pragma unroll
            for (int rn = 0; rn < RN; rn++) {
                const int ni = wn + 2 * rn;
                if (ni < NI) {
                    float* dst = part + (((sp * MO + mo) * NI + ni) * 64 + lane) * 16;
#pragma unroll
                    for (int v = 0; v < 4; v++) {
                        const f32x4 o = {acc[rm][rn][4 * v], acc[rm][rn][4 * v + 1], acc[rm][rn][4 * v + 2], acc[rm][rn][4 * v + 3]};
                        *(f32x4*)(dst + 4 * v) = o;
                    }
                }
            }
            if (wn == 0) {
                const float tot = bsum[rm] + __shfl_xor(bsum[rm], 32, 64);
                if (lane < 32) bpart[(sp * MO + mo) * 32 + lane] = tot;
            }
        }
    }
}



template <int W>
__global__ void __launch_bounds__(512, 2)
k_dw_all(durf::DwLevels lv, DwArgs a) {
    using S = MlpSpec<W>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // 1-D grid with no idle workgroups: the dispatcher deals workgroup ids round-robin to the 8
    // XCDs, so padding ids (a 2-D job x max_split grid) left some XCDs a fifth round of work.
    int oi = 0;
#pragma unroll
    for (int i = 1; i < 12; i++) oi += ((int)blockIdx.x >= a.first_wg[i]) ? 1 : 0;
    const int job = a.order[oi], sp = (int)blockIdx.x - a.first_wg[oi];
    const int ns = a.nsplit[job];
#define DW_CALL(...) dw_job<__VA_ARGS__>(lv, a, job, ns, sp, a.part[job], a.bpart[job], smem)
    switch (job) {
        case 0: DW_CALL(S::KW, S::KE, 0); break;
        case 5: DW_CALL(S::KW, S::KW, S::KE); break;
        // W = 256: the density head (d raw x h7) rides in job 10 as one more dz k-step -- both read h7 (512 B/sample).
        // W = 128 keeps it as its own job: the merged variant needs 147 instead of 81 VGPRs and would cost the
        // latency-bound object launches their second workgroup per CU.
        case 8: if (W != 256) DW_CALL(1, S::KW, 0); break;
        case 10:
            if (W == 256) DW_CALL(S::KC + 1, S::KW, S::KV, 1);     // [dz10 | d raw] x [h7 | view]
            else DW_CALL(S::KC, S::KW, S::KV);
            break;
        case 11: DW_CALL(1, S::KC, 0); break;
        default: DW_CALL(S::KW, S::KW, 0); break;      // Dense 1-4, 6, 7, 9
    }
#undef DW_CALL
}

// slot index (position in tile-layout feature order) of feature f
__host__ __device__ inline int cperm_slot(int f) {
    const int w = f & 15;
    return 16 * (f >> 4) + 8 * ((w >> 2) & 1) + (w & 3) + 4 * (w >> 3);
}

struct DwJob {
    int layer;           // flax Dense index this job's gradient belongs to
    int out_nat_off;     // >= 0: out slots are natural, out col c -> slot out_nat_off + c; < 0: C-perm
    int in_perm_rows;    // rows [0, in_perm_rows) use C-perm slots (segment A) ...
    int in_nat_base;     // ... rows beyond map to natural slots starting at this slot index
    int MO, NI;
};

// feature id held at C-perm slot position s (inverse of cperm_slot)
__host__ __device__ inline int cperm_feat(int s) {
    const int e = s & 7;
    return 16 * (s >> 4) + (e & 3) + 8 * (e >> 2) + 4 * ((s >> 3) & 1);
}

// Sum the split-K partials in fragment space (consecutive threads read consecutive floats of a partial: coalesced;
// one thread per element, partials in split order -> deterministic), then scatter each sum to its flax [in,out] position.
// One launch covers all 12 Dense layers of an MLP (blockIdx.y = job).
struct DwJobs { DwJob j[12]; size_t part_off[12], bpart_off[12]; int nparts[12]; };

// One launch finalizes up to TWO classes of MLPs (blockIdx.z: first the `count` MLPs of class 0, then those of class 1):
// a training step has the background MLP (W = 256, its own segment geometry) and the K object MLPs (W = 128, per-object
// ray counts), and finalizing them together saves two dependent launches per step.
struct FinMlp {
    int W, in_dim, count, NI10;
    DwJobs jobs;
    const float* part; const float* bpart; float* grad; const float* params;
    size_t part_stride, bpart_stride, grad_stride, param_stride;
    durf::DwLevels lv;
};
// 1-D grid of exactly the blocks that have work (an early-exit block is not free: the 3 x 12 x 768 padding blocks of a
// (widest job) x 12 x (1 + K) grid cost the merged launch ~80 us): MLP-major, then job, then 256-element block;
// job_blk[c][j] = first block of job j within one MLP of class c, job_blk[c][12] = blocks per MLP.
struct FinArgs { FinMlp m[2]; int job_blk[2][13]; };

__global__ void __launch_bounds__(256)
k_dw_finalize(FinArgs A) {
    const int n0 = A.m[0].count * A.job_blk[0][12];
    const int cls = (int)blockIdx.x < n0 ? 0 : 1;
    const FinMlp& M = A.m[cls];
    const int rel = (int)blockIdx.x - (cls ? n0 : 0);
    const size_t obj = rel / A.job_blk[cls][12];                   // batched object MLPs: the object
    const int rb = rel % A.job_blk[cls][12];
    int jb = 0;
#pragma unroll
    for (int j = 1; j < 12; j++) jb += (rb >= A.job_blk[cls][j]) ? 1 : 0;
    const int xblk = rb - A.job_blk[cls][jb];
    const int W = M.W, in_dim = M.in_dim;
    const DwJobs& jobs = M.jobs;
    const durf::DwLevels& lv = M.lv;
    const float* part_all = M.part + obj * M.part_stride;
    const float* bpart_all = M.bpart + obj * M.bpart_stride;
    float* grad_mlp = M.grad + obj * M.grad_stride;
    size_t nt_all = 0;                               // the splits k_dw_all actually wrote (see dw_tiles_per_split)
    for (int l = 0; l < lv.nlevels; l++) nt_all += dw_level_tiles(lv, l, obj);
    const DwJob job = jobs.j[jb];
    const size_t tps_ = dw_tiles_per_split(nt_all, jobs.nparts[jb]);
    const int nparts = (int)((nt_all + tps_ - 1) / tps_);
    const float* part = part_all + jobs.part_off[jb];
    const float* bpart = bpart_all + jobs.bpart_off[jb];
    int fi, fo;
    durf_layer_shape(W, in_dim, job.layer, &fi, &fo);
    const int nfrag = job.MO * job.NI * 1024;
    const int nb = job.MO * 32;
    // One thread per FOUR consecutive elements (one float4 per partial), 8 partials in flight, each element summed in split
    // order (deterministic, the same sums as one element per thread).  A block covers 1024 elements: the launch is bound by
    // the dispatch of its blocks whenever a job has few partials (small batches, objects -- with 256 elements per block 7500
    // blocks at K = 8, 32 us; with 64 per block and the partials dealt to 4 waves, round 2, ~30 000).
    const int idx0 = (xblk * 256 + (int)threadIdx.x) * 4;
    if (idx0 >= nfrag + nb) return;
    f32x4 s4 = {0.0f, 0.0f, 0.0f, 0.0f};
    {
        const float* p0 = idx0 < nfrag ? part + idx0 : bpart + (idx0 - nfrag);
        const size_t stride = idx0 < nfrag ? (size_t)nfrag : (size_t)nb;
        for (int p = 0; p < nparts; p += 8) {
            f32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] = (p + u < nparts) ? *(const f32x4*)(p0 + (size_t)(p + u) * stride) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int u = 0; u < 8; u++)
                if (p + u < nparts) { s4[0] += v[u][0]; s4[1] += v[u][1]; s4[2] += v[u][2]; s4[3] += v[u][3]; }
        }
    }
#pragma unroll
  for (int e4 = 0; e4 < 4; e4++) {
    const int idx = idx0 + e4;
    const float s = s4[e4];
    if (idx < nfrag) {
        const int r = idx & 15, lane = (idx >> 4) & 63, t = idx >> 10;
        const int ni = t % job.NI, mo = t / job.NI;
        const int hi = lane >> 5, jn = lane & 31;
        const int o_slot = 32 * mo + (r & 3) + 8 * (r >> 2) + 4 * hi;
        const int i_slot = 32 * ni + jn;
        const int col = job.out_nat_off >= 0 ? o_slot - job.out_nat_off : cperm_feat(o_slot);
        int row;
        if (i_slot < job.in_nat_base) row = cperm_feat(i_slot) < job.in_perm_rows ? cperm_feat(i_slot) : -1;
        else row = job.in_perm_rows + (i_slot - job.in_nat_base);
        if (job.layer == 10 && o_slot >= 128) {
            // the head-gradient k-step of job 10 (dz_out: slots 0-2 d rgb, 3 d density) x h7 = the density head's
            // weight gradient (Dense_8, kernel [W,1]); its products with the rgb slots and with the view features mean nothing
            if (o_slot == 128 + 3 && i_slot < job.in_nat_base)
                grad_mlp[durf_layer_offset(W, in_dim, 8, 0) + cperm_feat(i_slot)] = s;
        } else if (job.layer == 10 && i_slot < job.in_nat_base) {
            // view layer, rows fed by the bottleneck: this job multiplied dz10 with h7, not with the bottleneck output.
            // Keep the sum P[h7 feature, dz10 feature] in split 0's slot (only this thread group reads this element)
            // for k_bottleneck_grads, which turns it into the gradients of Dense_9 and of these rows of Dense_10.
            const_cast<float*>(part)[idx] = s;
        } else if (col >= 0 && col < fo && row >= 0 && row < fi)
            grad_mlp[durf_layer_offset(W, in_dim, job.layer, 0) + (size_t)row * fo + col] = s;
    } else {
        const int o_slot = idx - nfrag;
        if (job.layer == 10 && o_slot >= 128) {
            if (o_slot == 128 + 3) grad_mlp[durf_layer_offset(W, in_dim, 8, 1)] = s;        // density head bias
        } else {
            const int col = job.out_nat_off >= 0 ? o_slot - job.out_nat_off : cperm_feat(o_slot);
            if (col >= 0 && col < fo) grad_mlp[durf_layer_offset(W, in_dim, job.layer, 1) + col] = s;
            if (job.layer == 10) const_cast<float*>(bpart)[idx - nfrag] = s;     // db10, kept for k_bottleneck_grads
        }
    }
  }
}

// The bottleneck Dense_9 is LINEAR (obbpose_model.py:339: no activation between it and the view layer), so neither its
// output nor its gradient has to be exchanged through HBM:
//     bott = h7 K9 + b9,   z10 = [bott, view] K10 + b10,   d bott = dz10 K10_top^T      (K10_top: rows of K10 fed by bott)
//     dK9      = h7^T d bott  = (h7^T dz10) K10_top^T                  = P K10_top^T
//     db9      = sum_s d bott = (sum_s dz10) K10_top^T                 = db10 K10_top^T
//     dK10_top = bott^T dz10  = K9^T (h7^T dz10) + b9 (x) sum_s dz10   = K9^T P + b9 (x) db10
// with ONE sample-axis GEMM P = h7^T dz10 [W x 128] (job 10 of k_dw_all, which multiplies dz10 with [h7 | view]).  The
// forward stores no bottleneck activations (512 B/sample), the backward no bottleneck gradients (512 B/sample), and the
// weight-gradient launch reads neither (-768 B/sample together with the dropped job).  K9 / K10 enter bf16-rounded, as
// the MFMA kernels saw them; b9 in fp32, as the forward added it.  P and db10 are read from split 0 of job 10's
// partial buffers (fragment order), where k_dw_finalize left their sums.  blockIdx.y = object of a batched launch.
__device__ __forceinline__ float bf16r(float x) { return (float)(__bf16)x; }
__device__ __forceinline__ size_t frag_index(int W, int i, int j, int NI) {
    // P[h7 feature i, dz10 feature j] in job 10's fragment space [mo][ni][lane][16]
    const int i_slot = cperm_slot(i), o_slot = cperm_slot(j);
    const int ni = i_slot >> 5, jn = i_slot & 31, mo = o_slot >> 5, ol = o_slot & 31;
    const int hi = (ol >> 2) & 1, r = (ol & 3) + 4 * (ol >> 3);
    return (((size_t)mo * NI + ni) * 64 + hi * 32 + jn) * 16 + r;
}
// One workgroup per 32 x 32 output tile (LDS-staged operands: P is gathered from fragment order once per tile):
//   tiles [0, nA):        dK9[i, o]  = sum_j P[i, j] K10[o, j]                    (W/32 x W/32 tiles, K = 128)
//   tiles [nA, nA + nC):  dK10[o, j] = sum_i K9[i, o] P[i, j] + b9[o] db10[j]     (W/32 x 4 tiles,    K = W)
//   (tiles [0, W/32) also:  db9[o]   = sum_j db10[j] K10[o, j])
__global__ void __launch_bounds__(256)
k_bottleneck_grads(FinArgs A) {
    // (row stride 260 floats: 16-byte aligned rows, so the products read four k values per LDS instruction -- the launch's
    // compute phase was 5 ds_read_b32 per multiply-add group, ~1300 LDS instructions per wave at K = 256)
    __shared__ __attribute__((aligned(16))) float sa[32][260];        // rows of the first operand, K (<= 256) values each
    __shared__ __attribute__((aligned(16))) float sb[32][260];        // rows of the second operand
    __shared__ float sdb[128];           // db10 (first row of tiles: db9)
    const int cls = (int)blockIdx.y < A.m[0].count ? 0 : 1;       // blockIdx.y: MLP, classes as in k_dw_finalize
    const FinMlp& M = A.m[cls];
    const size_t obj = blockIdx.y - (cls ? A.m[0].count : 0);
    const int W = M.W, in_dim = M.in_dim, NI = M.NI10;
    if ((int)blockIdx.x >= (W / 32) * (W / 32) + (W / 32) * 4) return;          // the grid is sized for the wider class
    const float* part10 = M.part + M.jobs.part_off[10] + obj * M.part_stride;
    const float* bpart10 = M.bpart + M.jobs.bpart_off[10] + obj * M.bpart_stride;
    const float* params = M.params + obj * M.param_stride;
    float* grad = M.grad + obj * M.grad_stride;
    const size_t off9 = durf_layer_offset(W, in_dim, 9, 0), off10 = durf_layer_offset(W, in_dim, 10, 0);
    const float* K9 = params + off9;                 // [W, W]
    const float* b9 = K9 + (size_t)W * W;
    const float* K10 = params + off10;               // [W + 27, 128]; rows < W are fed by the bottleneck
    const int wt = W / 32, nA = wt * wt, nC = wt * 4;
    const int t = blockIdx.x, tid = threadIdx.x;
    const int tx = tid & 31, ty = tid >> 5;          // output (row = ty + 8 q, col = tx), q = 0..3
    // this thread's float4 of a fragment: lane = tid / 4 = (hi, jn), r = 4 (tid % 4) + e  ->  P row / column (frag_index inverted)
    const int f_hi = tid >> 7, f_jn = (tid >> 2) & 31, f_r0 = (tid & 3) * 4;
    auto frag_col = [&](int mo, int e) { const int r = f_r0 + e; return cperm_feat(32 * mo + (r & 3) + 8 * (r >> 2) + 4 * f_hi); };
    if (t < nA) {
        const int i0 = (t / wt) * 32, o0 = (t % wt) * 32;
        // (operand gathers: ALL of a thread's 16 + 16 loads in flight at once -- P was summed a moment ago on other XCDs, every
        // round of loads is a trip past this XCD's L2; one at a time the tile was a chain of them, 8 + 8 at a time two)
        {
            // P rows i0..i0+31 x all 128 columns = the four WHOLE fragments (mo = 0..3, ni = i0 / 32) of job 10's [mo][ni][lane][16]
            // space: one float4 per thread and fragment, coalesced, scattered into the LDS rows (frag_index inverted) -- as 16
            // gathered dwords per thread every load instruction touched up to 64 lines (round 6: the launch 17 -> 1x us)
            f32x4 pa[4];
            float vb[16];
            const int ni = i0 >> 5;
#pragma unroll
            for (int mo = 0; mo < 4; mo++) pa[mo] = *(const f32x4*)(part10 + ((size_t)(mo * NI + ni)) * 1024 + tid * 4);
#pragma unroll
            for (int u = 0; u < 16; u++) {
                const int e = tid + 256 * u, r = e >> 7, j = e & 127;
                vb[u] = K10[(size_t)(o0 + r) * 128 + j];
            }
            const int rr = cperm_feat(32 * ni + f_jn) - i0;
#pragma unroll
            for (int mo = 0; mo < 4; mo++)
#pragma unroll
                for (int e = 0; e < 4; e++) sa[rr][frag_col(mo, e)] = pa[mo][e];
#pragma unroll
            for (int u = 0; u < 16; u++) {
                const int e = tid + 256 * u, r = e >> 7, j = e & 127;
                sb[r][j] = bf16r(vb[u]);
            }
        }
        if (i0 == 0 && tid < 128) sdb[tid] = bpart10[cperm_slot(tid)];
        __syncthreads();
        float s4[4] = {0.0f, 0.0f, 0.0f, 0.0f};        // four independent chains, LDS reads batched 8 k-values at a time
#pragma unroll 2
        for (int j = 0; j < 128; j += 4) {
            const f32x4 b = *(const f32x4*)&sb[tx][j];
            f32x4 a4[4];
#pragma unroll
            for (int q = 0; q < 4; q++) a4[q] = *(const f32x4*)&sa[ty + 8 * q][j];
#pragma unroll
            for (int e = 0; e < 4; e++)          // (k ascending per chain, as one value at a time)
#pragma unroll
                for (int q = 0; q < 4; q++) s4[q] += a4[q][e] * b[e];
        }
#pragma unroll
        for (int q = 0; q < 4; q++) grad[off9 + (size_t)(i0 + ty + 8 * q) * W + o0 + tx] = s4[q];
        // db9[o] = sum_j db10[j] K10[o, j]: one more row of this product, carried by the first row of tiles (the rows of K10
        // are already in LDS).  (Until round 4 a workgroup of its own walked K10 with 8 dependent rounds of strided loads:
        // ~20 us on cold weights -- the long pole of the launch.)
        if (i0 == 0 && ty == 0) {
            float s = 0.0f;
#pragma unroll 2
            for (int j = 0; j < 128; j += 4) {
                const f32x4 b = *(const f32x4*)&sb[tx][j];
#pragma unroll
                for (int e = 0; e < 4; e++) s += sdb[j + e] * b[e];
            }
            grad[off9 + (size_t)W * W + o0 + tx] = s;
        }
    } else if (t < nA + nC) {
        const int u = t - nA, o0 = (u / 4) * 32, j0 = (u % 4) * 32;
        // P^T rows j0..j0+31 x all W columns = the W / 32 whole fragments (mo = j0 / 32, ni): coalesced float4s as above; K9's
        // columns o0..o0+31 by rows (a lane per column: 128 contiguous bytes per row instead of one line per lane)
#pragma unroll 1
        for (int n0 = 0; n0 < wt; n0 += 4) {                       // W / 32 = 4 or 8 fragments, four at a time
            f32x4 pb[4];
            float va[16];
            const int mo = j0 >> 5;
#pragma unroll
            for (int f = 0; f < 4; f++) pb[f] = *(const f32x4*)(part10 + ((size_t)(mo * NI + n0 + f)) * 1024 + tid * 4);
#pragma unroll
            for (int u = 0; u < 16; u++) {                          // rows i = 32 n0 + (tid >> 5) + 8 u, column o0 + (tid & 31)
                const int i = 32 * n0 + (tid >> 5) + 8 * u;
                va[u] = K9[(size_t)i * W + o0 + (tid & 31)];
            }
#pragma unroll
            for (int f = 0; f < 4; f++) {
                const int i = cperm_feat(32 * (n0 + f) + f_jn);
#pragma unroll
                for (int e = 0; e < 4; e++) sb[frag_col(mo, e) - j0][i] = pb[f][e];
            }
#pragma unroll
            for (int u = 0; u < 16; u++) sa[tid & 31][32 * n0 + (tid >> 5) + 8 * u] = bf16r(va[u]);
        }
        __syncthreads();
        float s4[4];
        const float db = bpart10[cperm_slot(j0 + tx)];
#pragma unroll
        for (int q = 0; q < 4; q++) s4[q] = b9[o0 + ty + 8 * q] * db;
#pragma unroll 2
        for (int i = 0; i < W; i += 4) {
            const f32x4 b = *(const f32x4*)&sb[tx][i];
            f32x4 a4[4];
#pragma unroll
            for (int q = 0; q < 4; q++) a4[q] = *(const f32x4*)&sa[ty + 8 * q][i];
#pragma unroll
            for (int e = 0; e < 4; e++)
#pragma unroll
                for (int q = 0; q < 4; q++) s4[q] += a4[q][e] * b[e];
        }
#pragma unroll
        for (int q = 0; q < 4; q++) grad[off10 + (size_t)(o0 + ty + 8 * q) * 128 + j0 + tx] = s4[q];
    }
}

// split plan of the grouped weight-gradient launch (host side, depends on the width only)
struct DwPlan {
    int nko[12], nki[12], MO[12], NI[12], nsplit[12];
    size_t part_off[12], bpart_off[12], part_total, bpart_total;
    int max_split, total_wgs;
};
// total_rows: the sample rows (all levels) the launch is sized for -- a host-side number both the launch and its finalize
// know; 0 = the largest plan (buffer sizes)
static DwPlan dw_plan(int width, size_t total_rows = 0) {
    DwPlan P;
    const int KW = width / 16;
    int cost = 0, wcost[12];
    for (int j = 0; j < 12; j++) {
        int nko = KW, nki = KW;
        if (j == 0) nki = 4;
        else if (j == 5) nki = KW + 4;
        else if (j == 8) nko = 1;
        else if (j == 10) { nko = width == 256 ? 9 : 8; nki = KW + 2; }        // W = 256: + the head-gradient k-step (density head)
        else if (j == 11) { nko = 1; nki = 8; }
        P.nko[j] = nko; P.nki[j] = nki;
        P.MO[j] = (nko + 1) / 2; P.NI[j] = nki / 2;
        wcost[j] = nko + nki + (nko + nki < 12 ? 3 : 0);     // the 9 KB/stage job is latency-bound: +30 % time per byte (traced)
        if (j == 9 || (j == 8 && width == 256)) wcost[j] = 0;   // no GEMM of their own: the bottleneck (k_bottleneck_grads)
                                                                // and, at W = 256, the density head (rides in job 10)
        cost += wcost[j];
    }
    // splits per job in proportion to its bytes per sample, summing EXACTLY to total_wgs (largest
    // remainder): with one resident workgroup per CU, n x 256 workgroups are n full rounds; two
    // stragglers from plain rounding (1026) cost another round (measured).  W = 256: two rounds -- round 2's
    // sweep at cfg3 (tools/experiments/sweep_dw_wgs.sh): 256 1507 us, 384 1617, 512 1445-1460, 640 1600, 768 1451-1459,
    // 1024 1464-1495, 1280 1490, 1536 1508; fewer workgroups also mean fewer fp32 partials to write and re-read
    // (268 -> 134 MB).  W = 128 (objects; the grid is total_wgs x K and a sparsely hit object leaves most of its
    // workgroups without tiles -- an early-exit workgroup still costs its dispatch): 256 per object (1024 -> 256: the
    // launch takes 137 -> 127 us at cfg3, K = 3, and 139 -> 73 us at cfg5, K = 8; rocprofv3).
    // DURF_DW_WGS / DURF_DW_WGS_OBJ force a plan whatever the size (read per call: the parity suite runs the small oracle
    // cases under the large batches' plan, tests/test_gpu_dispatch_matrix.py; a multiple of 128 up to the
    // largest plan, which is what durf_dw_part_floats sizes the partial buffers for)
    const char* env_s = getenv(width == 256 ? "DURF_DW_WGS" : "DURF_DW_WGS_OBJ");
    int env_w = env_s ? atoi(env_s) : 0;
    if (env_w % 128 != 0 || env_w < 128 || env_w > (width == 256 ? 512 : 256)) env_w = 0;
    // Round 4, small batches: every workgroup writes its fp32 partial tile whatever its share of the samples -- 134 MB per
    // launch with 512 workgroups, written here and read again by k_dw_finalize: ~50 us per step that do not shrink with
    // the batch.  ONE round of 256 workgroups halves that; the longer k loop per workgroup costs less than it saves below
    // ~3000 rays x 128 samples x 2 levels (step, k rays/s, 512 vs 256 workgroups: 512 rays 688-693 -> 714-728, cfg1
    // 1125-1136 -> 1185-1192, cfg5 761-783 -> 783-811, 1024 rays 810 -> 829, 2048 rays 921 -> 941, 4096 rays 966 -> 957;
    // 128 / 192 / 320 / 384 are worse everywhere: partial rounds).
    const bool small = total_rows > 0 && total_rows < (size_t)3072 * 256;
    const int total_wgs = env_w > 0 ? env_w : (width == 256 ? (small ? 256 : 512) : (small ? 128 : 256));     // (objects, small: 512 rays 706 -> 711, cfg5 776 -> 783)
    int base[12], given = 0;
    for (int j = 0; j < 12; j++) {
        base[j] = total_wgs * wcost[j] / cost;
        if (base[j] < 2) base[j] = wcost[j] ? 2 : 0;
        given += base[j];
    }
    for (int left = total_wgs - given; left > 0; left--) {
        int best = 0;
        long best_rem = -1;
        for (int j = 0; j < 12; j++) {
            const long rem = (long)total_wgs * wcost[j] - (long)base[j] * cost;
            if (wcost[j] && rem > best_rem) { best_rem = rem; best = j; }
        }
        base[best]++;
    }
    size_t po = 0, bo = 0;
    P.max_split = 0;
    for (int j = 0; j < 12; j++) {
        const int ns = base[j];
        P.nsplit[j] = ns;
        if (ns > P.max_split) P.max_split = ns;
        P.part_off[j] = po; P.bpart_off[j] = bo;
        po += (size_t)ns * P.MO[j] * P.NI[j] * 1024;
        bo += (size_t)ns * P.MO[j] * 32;
    }
    P.part_total = po; P.bpart_total = bo;
    P.total_wgs = total_wgs;
    return P;
}

// ---------------------------------------------------------------------------
extern "C" {

size_t durf_wpack_bwd_bytes(int width) {
    return (size_t)(width == 256 ? BwdSpec<256>::TOTAL_CHUNKS : BwdSpec<128>::TOTAL_CHUNKS) * 1024;
}

int durf_mlp_bwd(void* stream, int width, size_t rows, int N, const float* draw, const int32_t* ray_idx,
                 const int32_t* count, const void* wpack_bwd, const void* relu_mask, void* dz, void* dz_out,
                 float* d_enc, const int32_t* tail_idx, const int32_t* tail_count, const float* draw_ray_sum) {
    DURF_REQUIRE((tail_idx == nullptr) == (tail_count == nullptr) && (tail_idx == nullptr) == (draw_ray_sum == nullptr),
                 "tail_idx, tail_count and draw_ray_sum go together");
    DURF_REQUIRE(tail_idx == nullptr || (count != nullptr && N % 32 == 0), "tail rows follow a compacted ray list");
    return durf::launch_mlp_bwd(stream, width, rows, N, draw, ray_idx, count, wpack_bwd, relu_mask, dz, dz_out, d_enc,
                                1, BwdStrides{}, tail_idx, tail_count, draw_ray_sum);
}

// durf_mlp_bwd (the background MLP) + durf_obj_bwd_batch_levels (the K object MLPs over `nlevels` levels) as ONE launch where
// that pays (include/durf_hip.h), else as the two launches
int durf_obj_bwd_batch_levels(void* stream, int K, int B, int N, int nlevels, const int32_t* idx, const int32_t* count,
                              const float* const* draw, const void* wpack_bwd, const void* const* relu_mask, void* const* dz,
                              void* const* dz_out);
int durf_mlp_bwd_obj(void* stream, size_t rows, int N, const float* draw, const int32_t* ray_idx, const int32_t* count,
                     const void* wpack_bwd, const void* relu_mask, void* dz, void* dz_out, const int32_t* tail_idx,
                     const int32_t* tail_count, const float* draw_ray_sum, int K, int B, int nlevels, const int32_t* obj_idx,
                     const int32_t* obj_count, const float* const* obj_draw, const void* obj_wpack_bwd,
                     const void* const* obj_relu_mask, void* const* obj_dz, void* const* obj_dz_out) {
    DURF_REQUIRE(K > 0 && B > 0 && (size_t)B * N == rows, "K object MLPs over rows = B * N sample rows");
    DURF_REQUIRE(nlevels >= 1 && nlevels <= DURF_MAX_LEVELS, "1 <= nlevels <= DURF_MAX_LEVELS");
    DURF_REQUIRE(obj_idx && obj_count && obj_draw && obj_wpack_bwd && obj_relu_mask && obj_dz && obj_dz_out, "the object launch's buffers");
    const bool mix = durf::obj_mix(rows) && N % 32 == 0 && ray_idx != nullptr && count != nullptr;
    if (!mix) {
        int rc = durf_mlp_bwd(stream, 256, rows, N, draw, ray_idx, count, wpack_bwd, relu_mask, dz, dz_out, nullptr, tail_idx, tail_count,
                              draw_ray_sum);
        if (rc) return rc;
        return durf_obj_bwd_batch_levels(stream, K, B, N, nlevels, obj_idx, obj_count, obj_draw, obj_wpack_bwd, obj_relu_mask, obj_dz,
                                         obj_dz_out);
    }
    DURF_REQUIRE((tail_idx == nullptr) == (tail_count == nullptr) && (tail_idx == nullptr) == (draw_ray_sum == nullptr),
                 "tail_idx, tail_count and draw_ray_sum go together");
    DURF_REQUIRE(rows % 32 == 0, "rows must be a multiple of 32");
    MsBwd ow{};
    ow.rows = rows; ow.N = N; ow.ray_idx = obj_idx; ow.count = obj_count; ow.wpack = (const char*)obj_wpack_bwd; ow.nobj = K;
    ow.lv.n = nlevels;
    for (int l = 0; l < nlevels; l++) {
        ow.lv.draw[l] = obj_draw[l]; ow.lv.relu_mask[l] = (const uint4*)obj_relu_mask[l]; ow.lv.dz[l] = (bf16x8*)obj_dz[l];
        ow.lv.dz_out[l] = (bf16x8*)obj_dz_out[l];
    }
    ow.bs.idx = (size_t)B; ow.bs.wpack = durf_wpack_bwd_bytes(DURF_W_OBJ); ow.bs.mask = durf_mlp_mask_bytes(rows);
    ow.bs.dz = durf_mlp_stash_bytes(DURF_W_OBJ, rows); ow.bs.dz_out = durf_obj_dzout_stride(B, N);
    ow.bs.d_enc = rows * DURF_ENC_DIM * sizeof(float);
    ow.ticket = durf::next_ticket();
    DURF_REQUIRE(ow.ticket != nullptr, "no item counter for the mixed launch (device allocation failed)");
    const unsigned nblk = durf_cdiv(rows, 256), nobj = durf_cdiv((size_t)nlevels * K * durf_cdiv(rows, 64), 2);
    const unsigned g = nblk + nobj < 256u ? nblk + nobj : 256u;
    constexpr int lds = 2 * 4 * (MlpSpec<256>::KW + 1) * 1024;
    static_assert(2 * msb::LDS_BYTES + 96 <= lds, "two object groups fit the background block's LDS");
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)k_mlp_bwd<256, false, 8, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_set = true;
    }
    hipLaunchKernelGGL((k_mlp_bwd<256, false, 8, true>), dim3(g), dim3(512), lds, (hipStream_t)stream, rows, N, draw, ray_idx, count,
                       (const char*)wpack_bwd, (const uint4*)relu_mask, (bf16x8*)dz, (bf16x8*)dz_out, (float*)nullptr, BwdStrides{},
                       tail_idx, tail_count, draw_ray_sum, ow);
    DURF_CHECK_LAUNCH("durf_mlp_bwd_obj");
    durf::note_dispatch(DURF_DISPATCH_BWD256_8W | DURF_DISPATCH_BWD_MIX);
    return 0;
}

int durf_expand_view(void* stream, size_t rows, int N, const void* view_bf16, const int32_t* ray_idx,
                     const int32_t* count, void* out_tile, const int32_t* tail_idx, const int32_t* tail_count) {
    DURF_REQUIRE((tail_idx == nullptr) == (tail_count == nullptr), "tail_idx and tail_count go together");
    return durf::launch_expand_view(stream, rows, N, view_bf16, ray_idx, count, out_tile, 1, 0, 0, tail_idx, tail_count);
}

size_t durf_dw_part_floats(int width) { return dw_plan(width).part_total; }
size_t durf_dw_bpart_floats(int width) { return dw_plan(width).bpart_total; }

int durf_mlp_dw(void* stream, int width, size_t rows, int N, const int32_t* count, int nlevels,
                const void* const* enc_tile, const void* const* view_tile, const void* const* stash,
                const void* const* dz, const void* const* dz_out, float* part, float* bpart) {
    DURF_REQUIRE(rows % 32 == 0, "rows must be a multiple of 32");
    DURF_REQUIRE(nlevels >= 1 && nlevels <= DURF_MAX_LEVELS, "1 <= nlevels <= DURF_MAX_LEVELS");
    return durf::launch_mlp_dw(stream, width, durf::uniform_levels(rows, N, count, nlevels), enc_tile, view_tile, stash, dz, dz_out, part,
                               bpart, 1, DwStrides{});
}

static int make_levels(int nlevels, const size_t* rows, const int* rows_per_ray, const int32_t* const* count,
                       durf::DwLevels* lv) {
    DURF_REQUIRE(nlevels >= 1 && nlevels <= DURF_MAX_LEVELS, "1 <= nlevels <= DURF_MAX_LEVELS");
    lv->nlevels = nlevels;
    for (int l = 0; l < DURF_MAX_LEVELS; l++) {
        const int ll = l < nlevels ? l : 0;
        DURF_REQUIRE(rows_per_ray[ll] >= 1, "rows_per_ray >= 1");
        lv->rows[l] = rows[ll]; lv->n[l] = rows_per_ray[ll]; lv->count[l] = count ? count[ll] : nullptr;
    }
    return 0;
}

int durf_mlp_dw_levels(void* stream, int width, int nlevels, const size_t* rows, const int* rows_per_ray,
                       const int32_t* const* count, const void* const* enc_tile, const void* const* view_tile,
                       const void* const* stash, const void* const* dz, const void* const* dz_out, float* part,
                       float* bpart) {
    durf::DwLevels lv;
    if (int rc = make_levels(nlevels, rows, rows_per_ray, count, &lv)) return rc;
    return durf::launch_mlp_dw(stream, width, lv, enc_tile, view_tile, stash, dz, dz_out, part, bpart, 1, DwStrides{});
}

int durf_mlp_dw_finalize_levels(void* stream, int width, int in_dim, int nlevels, const size_t* rows,
                                const int* rows_per_ray, const int32_t* const* count, const float* part,
                                const float* bpart, float* grad_mlp, const float* mlp_params) {
    durf::DwLevels lv;
    if (int rc = make_levels(nlevels, rows, rows_per_ray, count, &lv)) return rc;
    return durf::launch_dw_finalize(stream, width, in_dim, lv, part, bpart, grad_mlp, 1, 0, 0, 0, mlp_params, 0);
}

int durf_mlp_dw_finalize(void* stream, int width, int in_dim, size_t rows, int N, const int32_t* count,
                         int nlevels, const float* part, const float* bpart, float* grad_mlp,
                         const float* mlp_params) {
    DURF_REQUIRE(nlevels >= 1 && nlevels <= DURF_MAX_LEVELS, "1 <= nlevels <= DURF_MAX_LEVELS");
    return durf::launch_dw_finalize(stream, width, in_dim, durf::uniform_levels(rows, N, count, nlevels), part, bpart, grad_mlp, 1, 0, 0, 0,
                                    mlp_params, 0);
}

}  // extern "C"

namespace durf {

int pack_bwd_launch(void* stream, int width, int in_dim, int K, const float* params, size_t param_stride, void* wpack_bwd) {
    hipStream_t s = (hipStream_t)stream;
    const size_t ostride = durf_wpack_bwd_bytes(width);
    if (width == 256)
        hipLaunchKernelGGL(k_pack_bwd<256>, dim3(durf_cdiv(BwdSpec<256>::TOTAL_CHUNKS * 64, 256), K), dim3(256), 0, s,
                           in_dim, params, (bf16x8*)wpack_bwd, param_stride, ostride);
    else
        hipLaunchKernelGGL(k_pack_bwd<128>, dim3(durf_cdiv(BwdSpec<128>::TOTAL_CHUNKS * 64, 256), K), dim3(256), 0, s,
                           in_dim, params, (bf16x8*)wpack_bwd, param_stride, ostride);
    DURF_CHECK_LAUNCH("durf_pack_weights (bwd)");
    return 0;
}

int launch_mlp_bwd(void* stream, int width, size_t rows, int N, const float* draw, const int32_t* ray_idx,
                   const int32_t* count, const void* wpack_bwd, const void* relu_mask, void* dz, void* dz_out,
                   float* d_enc, int K, const BwdStrides& st, const int32_t* tail_idx, const int32_t* tail_count,
                   const float* draw_ray_sum) {
    DURF_REQUIRE(width == 256 || width == 128, "width must be 256 or 128");
    DURF_REQUIRE(rows % 32 == 0, "rows must be a multiple of 32");
    DURF_REQUIRE(K == 1 || (ray_idx && count), "batched launches are for compacted object rays");
    if (rows == 0 || K <= 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    // the object MLPs (W = 128 on compacted ray lists, no d(enc)): the M-split kernel (k_mlp_bwd_ms; DURF_OBJ_MSPLIT=0: A/B switch)
    {
        if (width == 128 && ray_idx && count && !d_enc && !tail_idx && obj_msplit(rows)) {
            const size_t items = (size_t)K * durf_cdiv(rows, 64);       // (small batches only; one round of workgroups: launch_mlp_fwd)
            MsBwdLevels lv{};
            lv.draw[0] = draw; lv.relu_mask[0] = (const uint4*)relu_mask; lv.dz[0] = (bf16x8*)dz; lv.dz_out[0] = (bf16x8*)dz_out; lv.n = 1;
            const MsBwd A{rows, N, lv, ray_idx, count, (const char*)wpack_bwd, st, K, nullptr};
            hipLaunchKernelGGL(k_mlp_bwd_ms, dim3((unsigned)(items < 256 ? items : 256)), dim3(256), msb::LDS_BYTES, s, A);
            DURF_CHECK_LAUNCH("durf_mlp_bwd (M-split)");
            note_dispatch(DURF_DISPATCH_BWD128_MSPLIT);
            return 0;
        }
    }
    // (as launch_mlp_fwd: 128-sample blocks of 4 waves when 256-sample blocks would leave half the chip idle)
    const bool half = width == 256 && K == 1 && !d_enc && durf_cdiv(rows, 256) <= 128;
    const unsigned nblk = durf_cdiv(rows, half ? 128u : 256u);
    dim3 grid(nblk < 256u ? nblk : 256u, K), block(half ? 256 : 512);     // persistent: at most one workgroup per CU and object
#define LAUNCH_B(WW, PP, NWV)                                                                              \
    {                                                                                                      \
        constexpr int lds = 2 * 4 * (MlpSpec<WW>::KW + 1) * 1024;                                          \
        (void)hipFuncSetAttribute((const void*)k_mlp_bwd<WW, PP, NWV>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); \
        hipLaunchKernelGGL((k_mlp_bwd<WW, PP, NWV>), grid, block, lds, s, rows, N, draw, ray_idx, count,    \
                           (const char*)wpack_bwd, (const uint4*)relu_mask, (bf16x8*)dz, (bf16x8*)dz_out,  \
                           d_enc, st, tail_idx, tail_count, draw_ray_sum, MsBwd{});                        \
    }
    if (half) LAUNCH_B(256, false, 4)
    else if (width == 256) { if (d_enc) LAUNCH_B(256, true, 8) else LAUNCH_B(256, false, 8) }
    else { if (d_enc) LAUNCH_B(128, true, 8) else LAUNCH_B(128, false, 8) }
#undef LAUNCH_B
    DURF_CHECK_LAUNCH("durf_mlp_bwd");
    note_dispatch((width == 128 ? DURF_DISPATCH_BWD128_SAMPLE : (half ? DURF_DISPATCH_BWD256_4W : DURF_DISPATCH_BWD256_8W)) |
                  (d_enc ? DURF_DISPATCH_BWD_POSE : 0u));
    return 0;
}

// the M-split object backward of SEVERAL levels as one launch (levels x objects x pairs dealt to one 1-D grid); the caller
// checked obj_msplit(rows)
int launch_mlp_bwd_ms_levels(void* stream, size_t rows, int N, int nlevels, const float* const* draw, const int32_t* ray_idx,
                             const int32_t* count, const void* wpack_bwd, const void* const* relu_mask, void* const* dz,
                             void* const* dz_out, int K, const BwdStrides& st) {
    DURF_REQUIRE(nlevels >= 1 && nlevels <= DURF_MAX_LEVELS && rows % 32 == 0 && ray_idx && count, "1 <= nlevels <= DURF_MAX_LEVELS, compacted rays");
    if (rows == 0 || K <= 0) return 0;
    MsBwdLevels lv{};
    lv.n = nlevels;
    for (int l = 0; l < nlevels; l++) {
        lv.draw[l] = draw[l]; lv.relu_mask[l] = (const uint4*)relu_mask[l]; lv.dz[l] = (bf16x8*)dz[l]; lv.dz_out[l] = (bf16x8*)dz_out[l];
    }
    const size_t items = (size_t)nlevels * K * durf_cdiv(rows, 64);
    const MsBwd A{rows, N, lv, ray_idx, count, (const char*)wpack_bwd, st, K, nullptr};
    hipLaunchKernelGGL(k_mlp_bwd_ms, dim3((unsigned)(items < 256 ? items : 256)), dim3(256), msb::LDS_BYTES, (hipStream_t)stream, A);
    DURF_CHECK_LAUNCH("durf_obj_bwd_batch_levels (M-split)");
    note_dispatch(DURF_DISPATCH_BWD128_MSPLIT);
    return 0;
}

int launch_expand_view(void* stream, size_t rows, int N, const void* view_bf16, const int32_t* ray_idx,
                       const int32_t* count, void* out_tile, int K, size_t idx_stride, size_t out_stride,
                       const int32_t* tail_idx, const int32_t* tail_count) {
    if (rows == 0 || K <= 0) return 0;
    hipLaunchKernelGGL(k_expand_view, dim3(durf_cdiv(rows * 4, 256), K), dim3(256), 0, (hipStream_t)stream, rows, N,
                       (const bf16x8*)view_bf16, ray_idx, count, (bf16x8*)out_tile, idx_stride, out_stride, tail_idx,
                       tail_count);
    DURF_CHECK_LAUNCH("durf_expand_view");
    return 0;
}

// All weight gradients of one MLP: ONE grouped launch of the 12 split-K GEMMs whose K axis runs over
// the samples of every level (per-level operand buffers, one set of fp32 partials).
DwLevels uniform_levels(size_t rows, int N, const int32_t* count, int nlevels) {
    DwLevels lv;
    lv.nlevels = nlevels;
    for (int l = 0; l < DURF_MAX_LEVELS; l++) { lv.rows[l] = rows; lv.n[l] = N; lv.count[l] = count; }
    return lv;
}

int launch_mlp_dw(void* stream, int width, const DwLevels& lv,
                  const void* const* enc_tile, const void* const* view_tile, const void* const* stash,
                  const void* const* dz, const void* const* dz_out, float* part, float* bpart, int K,
                  const DwStrides& st) {
    const int nlevels = lv.nlevels;
    DURF_REQUIRE(width == 256 || width == 128, "width must be 256 or 128");
    DURF_REQUIRE(nlevels >= 1 && nlevels <= DURF_MAX_LEVELS, "1 <= nlevels <= DURF_MAX_LEVELS");
    size_t total_rows = 0;
    for (int l = 0; l < nlevels; l++) {
        DURF_REQUIRE(lv.rows[l] % 32 == 0, "rows must be a multiple of 32");
        DURF_REQUIRE(K == 1 || lv.count[l], "batched launches are for compacted object rays");
        total_rows += lv.rows[l];
    }
    if (total_rows == 0 || K <= 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    const DwPlan P = dw_plan(width, total_rows);
    {   // the caller sized the partial buffers with durf_dw_part_floats / durf_dw_bpart_floats = the largest plan
        const DwPlan big = dw_plan(width);
        DURF_REQUIRE(P.part_total <= big.part_total && P.bpart_total <= big.bpart_total, "split plan exceeds the partial buffers");
    }
    const int KW = width / 16;
    auto region = [&](const void* base, int j, int l) { return (const char*)base + ((size_t)j * KW * (lv.rows[l] >> 5)) * 1024; };
    DwArgs a;
    a.nlevels = nlevels;
    a.ks_part = st.part; a.ks_bpart = st.bpart;
    a.ks_dz2 = st.dz_out;
    for (int j = 0; j < 12; j++) {
        a.ks_dz[j] = (j == 8 || j == 11) ? st.dz_out : st.stash;
        a.ks_a[j] = j == 0 ? st.enc : st.stash;
        a.ks_b[j] = j == 5 ? st.enc : (j == 10 ? st.view : 0);
    }
    for (int j = 0; j < 12; j++) {
        a.part[j] = part + P.part_off[j];
        a.bpart[j] = bpart + P.bpart_off[j];
        a.nsplit[j] = P.nsplit[j];
    }
    for (int l = 0; l < DURF_MAX_LEVELS; l++) {
        const int ll = l < nlevels ? l : 0;
        for (int j = 0; j < 12; j++) {
            a.inB[l][j] = nullptr;
            if (j <= 7) {
                a.dz[l][j] = region(dz[ll], j, ll);
                a.inA[l][j] = j == 0 ? (const char*)enc_tile[ll] : region(stash[ll], j - 1, ll);
            }
        }
        a.inB[l][5] = (const char*)enc_tile[ll];
        a.dz[l][8] = (const char*)dz_out[ll]; a.inA[l][8] = region(stash[ll], 7, ll);      // density head (its own job at W = 128)
        a.dz2[l] = (const char*)dz_out[ll];                                                  // ... the last dz k-step of job 10 at W = 256
        a.dz[l][9] = nullptr; a.inA[l][9] = nullptr;            // bottleneck (linear): no job, see k_bottleneck_grads
        // view layer: dz10 x [h7 | view] -- P = h7^T dz10 stands in for the bottleneck's activations and gradients
        a.dz[l][10] = region(dz[ll], 9, ll); a.inA[l][10] = region(stash[ll], 7, ll); a.inB[l][10] = (const char*)view_tile[ll];
        a.dz[l][11] = (const char*)dz_out[ll]; a.inA[l][11] = region(stash[ll], 9, ll);        // rgb head
    }
    // narrow jobs first: their workgroups run longest (less data in flight per stage)
    static const int order[12] = {11, 8, 0, 10, 5, 1, 2, 3, 4, 6, 7, 9};
    int total = 0;
    for (int i = 0; i < 12; i++) {
        a.order[i] = order[i];
        a.first_wg[i] = total;
        total += P.nsplit[order[i]];
    }
    a.first_wg[12] = total;
    dim3 grid(total, K), block(512);
    if (width == 256) {
        constexpr int lds = DW_LDS_BYTES;
        (void)hipFuncSetAttribute((const void*)k_dw_all<256>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        hipLaunchKernelGGL(k_dw_all<256>, grid, block, lds, s, lv, a);
    } else {
        // widest W=128 job: 20 KB per stage x 4 stages = 80 KB, so two workgroups share a CU (80 VGPRs per lane):
        // the object GEMMs are per-tile-latency-bound with one resident workgroup
        constexpr int lds = dw_stages(20) * 20 * 1024;
        (void)hipFuncSetAttribute((const void*)k_dw_all<128>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        hipLaunchKernelGGL(k_dw_all<128>, grid, block, lds, s, lv, a);
    }
    DURF_CHECK_LAUNCH("durf_mlp_dw");
    note_dispatch(width == 256 ? (P.total_wgs <= 256 ? DURF_DISPATCH_DW256_256WG : DURF_DISPATCH_DW256_512WG)
                               : (P.total_wgs <= 128 ? DURF_DISPATCH_DW128_128WG : DURF_DISPATCH_DW128_256WG));
    return 0;
}

// fills one class of a finalize launch
static int fin_class(FinMlp& M, int width, int in_dim, const DwLevels& lv, const float* part, const float* bpart,
                     float* grad_mlp, int K, size_t part_stride, size_t bpart_stride, size_t grad_stride,
                     const float* mlp_params, size_t param_stride, int* max_el, int* max_tiles) {
    DURF_REQUIRE(mlp_params != nullptr, "the bottleneck gradients need the MLP's parameters");
    DURF_REQUIRE(width == 256 || width == 128, "width must be 256 or 128");
    const int W = width, KW = W / 16;
    size_t total_rows = 0;
    for (int l = 0; l < lv.nlevels; l++) total_rows += lv.rows[l];
    const DwPlan P = dw_plan(width, total_rows);
    M.W = W; M.in_dim = in_dim; M.count = K; M.NI10 = P.NI[10];
    M.part = part; M.bpart = bpart; M.grad = grad_mlp; M.params = mlp_params;
    M.part_stride = part_stride; M.bpart_stride = bpart_stride; M.grad_stride = grad_stride; M.param_stride = param_stride;
    M.lv = lv;
    for (int job = 0; job < 12; job++) {
        DwJob& J = M.jobs.j[job];
        J.layer = job;
        J.out_nat_off = -1;
        J.in_perm_rows = W; J.in_nat_base = KW * 16;
        if (job == 0) { J.in_perm_rows = 0; J.in_nat_base = 0; }
        else if (job == 8) { J.out_nat_off = 3; }                 // density head: dz_out slot 3
        else if (job == 11) { J.out_nat_off = 0; J.in_perm_rows = 128; J.in_nat_base = 128; }
        J.MO = P.MO[job]; J.NI = P.NI[job];
        M.jobs.part_off[job] = P.part_off[job];
        M.jobs.bpart_off[job] = P.bpart_off[job];
        M.jobs.nparts[job] = P.nsplit[job];
        const int el = J.MO * J.NI * 1024 + J.MO * 32;
        if (el > *max_el) *max_el = el;
    }
    const int ntiles = (W / 32) * (W / 32) + (W / 32) * 4;
    if (ntiles > *max_tiles) *max_tiles = ntiles;
    return 0;
}

static int launch_fin(void* stream, FinArgs& A, int max_el, int max_tiles) {
    const int n = A.m[0].count + A.m[1].count;
    if (n <= 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    int total = 0;
    for (int c = 0; c < 2; c++) {
        int b = 0;
        for (int j = 0; j < 12; j++) {
            A.job_blk[c][j] = b;
            if (A.m[c].count > 0 && A.m[c].jobs.nparts[j] > 0)          // (the bottleneck layer has no job of its own)
                b += durf_cdiv(A.m[c].jobs.j[j].MO * A.m[c].jobs.j[j].NI * 1024 + A.m[c].jobs.j[j].MO * 32, 1024);
        }
        A.job_blk[c][12] = b > 0 ? b : 1;
        total += A.m[c].count * b;
    }
    (void)max_el;
    hipLaunchKernelGGL(k_dw_finalize, dim3(total), dim3(256), 0, s, A);
    hipLaunchKernelGGL(k_bottleneck_grads, dim3(max_tiles, n), dim3(256), 0, s, A);
    DURF_CHECK_LAUNCH("durf_mlp_dw_finalize");
    return 0;
}

int launch_dw_finalize(void* stream, int width, int in_dim, const DwLevels& lv,
                       const float* part, const float* bpart, float* grad_mlp, int K, size_t part_stride,
                       size_t bpart_stride, size_t grad_stride, const float* mlp_params, size_t param_stride) {
    if (K <= 0) return 0;
    FinArgs A;
    A.m[1].count = 0;
    int max_el = 0, max_tiles = 0;
    if (int rc = fin_class(A.m[0], width, in_dim, lv, part, bpart, grad_mlp, K, part_stride, bpart_stride, grad_stride,
                           mlp_params, param_stride, &max_el, &max_tiles)) return rc;
    return launch_fin(stream, A, max_el, max_tiles);
}

// the background MLP (class 0) and the K object MLPs (class 1) in one finalize launch
int launch_dw_finalize2(void* stream, const DwFinSpec& a, const DwFinSpec& b) {
    FinArgs A;
    A.m[0].count = 0; A.m[1].count = 0;
    int max_el = 0, max_tiles = 0, c = 0;
    for (const DwFinSpec* sp : {&a, &b}) {
        if (sp->K <= 0) continue;
        if (int rc = fin_class(A.m[c], sp->width, sp->in_dim, sp->lv, sp->part, sp->bpart, sp->grad, sp->K, sp->part_stride,
                               sp->bpart_stride, sp->grad_stride, sp->params, sp->param_stride, &max_el, &max_tiles)) return rc;
        c++;
    }
    return launch_fin(stream, A, max_el, max_tiles);
}


}  // namespace durf
