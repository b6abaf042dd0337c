// Fused MLP forward (K6 bkgd 8x256, K7 per-object 8x128): obbpose_model.py:305-354 / :369-418.
//
// One workgroup = 8 waves = 256 samples; each wave owns 32 samples (one MFMA N tile) and
// carries their activations in registers through all 11 stages (see mlp_spec.h for the
// orientation trick).  Weights are pre-packed bf16 A-fragments streamed L2 -> LDS with
// global_load_lds (one 1 KB chunk per wave-instruction, lane-linear = fragment order, so
// ds_read_b128 is conflict-free), double buffered per output M-tile, one barrier per tile.
// MFMA-bound: 2*606 208 padded MAC per sample (591 872 algorithmic) for W=256.
#include <stdlib.h>
#include <cstddef>
#include "mlp_pack.h"
#include "enc_lane.h"

// 16 bytes per lane, global -> LDS (lane-linear destination), as a BUFFER load: descriptor in
// SGPRs, one constant per-lane VGPR offset (lane*16), the chunk offset in an SGPR.  The
// global_load_lds form needs a 64-bit per-lane VGPR address for every tile group; hipcc
// precomputes those, they spill, and a scratch reload next to an in-flight LDS-DMA makes it
// drain the whole weight prefetch (s_waitcnt vmcnt(0)).

// ---------------------------------------------------------------------------
// weight packer: fp32 flax params -> bf16 fragment stream (one thread per 16-byte vector)
// ---------------------------------------------------------------------------
template <int W>
__global__ void __launch_bounds__(256)
k_pack_fwd(int in_dim, const float* __restrict__ P, bf16x8* __restrict__ out, size_t p_stride, size_t out_stride) {
    P += blockIdx.y * p_stride;                                  // object index (0 for a single MLP)
    out = (bf16x8*)((char*)out + blockIdx.y * out_stride);
    pack_fwd_vec<W>(in_dim, P, out, blockIdx.x * blockDim.x + threadIdx.x);
}

// Every weight stream of the model in one launch (PackAll / pack_all_vec: mlp_pack.h): blockIdx.y = MLP
__global__ void __launch_bounds__(256)
k_pack_all(PackAll a) {
    pack_all_vec(a, (int)blockIdx.y, (int)(blockIdx.x * blockDim.x + threadIdx.x));
}

// ---------------------------------------------------------------------------
// fused forward
// ---------------------------------------------------------------------------
struct WPipe {
    i32x4 rsrc;          // the packed weight stream
    unsigned gnext;      // byte offset of the next tile group to prefetch
    unsigned lds0;       // LDS byte address of the two slots
    char* lds;
    int slot_bytes;
    int par;             // slot that holds the tile about to be consumed
    int wave, lane;
    int nw;              // waves of the workgroup (8; 4 for launches that would leave half the chip idle, launch_mlp_fwd)
    int since;           // vector-memory ops (stores) this wave issued after its last weight DMA (lower bound)
    __device__ __forceinline__ void issue(int slot, int chunks) {
        const unsigned dst = lds0 + (unsigned)(slot * slot_bytes);
        for (int c = wave; c < chunks; c += nw)
            lds_dma16_cached(rsrc, gnext + c * 1024u, lane * 16u, dst + c * 1024u);
        gnext += chunks * 1024u;
        since = 0;
    }
    // Make the prefetched tile group visible, start the prefetch of the following one
    // (next_chunks KB, 0 = none) into the other slot, and return the slot to consume.
    // The wait covers this wave's part of the DMA but leaves the stores issued after it in
    // flight; the barrier then publishes every wave's part and frees the other slot.
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
__device__ __forceinline__ f32x16 mma_tile(const char* slot, int lane, const bf16x8* inA,
                                           const bf16x8* inB) {
    const f32x4* bp = (const f32x4*)(slot + (NA + NB) * 1024 + (lane >> 5) * 64);
    f32x16 acc;
#pragma unroll
    for (int g = 0; g < 4; g++) {
        const f32x4 b4 = bp[g];
        acc[4 * g + 0] = b4[0]; acc[4 * g + 1] = b4[1]; acc[4 * g + 2] = b4[2]; acc[4 * g + 3] = b4[3];
    }
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

// Two output tiles at once through the fragment ring (mma_pair_ring, mlp_spec.h); the bias rows are the initial accumulators.
template <int NA, int NB, bool SPREAD = false>
__device__ __forceinline__ void mma_tile2(const char* slot0, const char* slot1, int lane, const bf16x8* inA,
                                          const bf16x8* inB, f32x16& acc0, f32x16& acc1,
                                          const bf16x8* sv = nullptr, char* sp = nullptr, bool do_store = false) {
    // SPREAD: the four 16-byte stores of the previous tile pair (sv[0..3] -> sp + q KB) are issued between the
    // k-steps instead of in one burst before them, so a store that waits for queue space has MFMAs ahead of it.
    constexpr int T = NA + NB;
    const f32x4* bp0 = (const f32x4*)(slot0 + (NA + NB) * 1024 + (lane >> 5) * 64);
    const f32x4* bp1 = (const f32x4*)(slot1 + (NA + NB) * 1024 + (lane >> 5) * 64);
#pragma unroll
    for (int g = 0; g < 4; g++) {
        const f32x4 b0 = bp0[g], b1 = bp1[g];
        acc0[4 * g + 0] = b0[0]; acc0[4 * g + 1] = b0[1]; acc0[4 * g + 2] = b0[2]; acc0[4 * g + 3] = b0[3];
        acc1[4 * g + 0] = b1[0]; acc1[4 * g + 1] = b1[1]; acc1[4 * g + 2] = b1[2]; acc1[4 * g + 3] = b1[3];
    }
    mma_pair_ring<NA, NB, FWD_LDS_RING>(slot0, slot1, lane, inA, inB, acc0, acc1, [&](auto ks_) {
        constexpr int ks = decltype(ks_)::value;
        if constexpr (SPREAD) {
            static_for<0, 4>([&](auto q_) {
                constexpr int q = decltype(q_)::value;
                if constexpr (ks == (q + 1) * T / 4 - 1) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (do_store) STREAM_STORE(sp + q * 1024, sv[q]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            });
        }
    });
}

// packs one output tile to bf16 (ReLU optional) and returns its 16 "activation != 0" flags
template <bool RELU, bool BITS>
__device__ __forceinline__ unsigned pack_tile(const f32x16& acc, bf16x8& o0, bf16x8& o1) {
    static_assert(RELU || !BITS, "mask bits are taken from the ReLU output");
#pragma unroll
    for (int e = 0; e < 8; e++) {
        float v0 = acc[e], v1 = acc[8 + e];
        if (RELU) {     // one v_med3_f32 (fmaxf costs two v_max_f32: it quiets the input first); NaN -> poison flag
            v0 = __builtin_amdgcn_fmed3f(v0, 0.0f, 3.0e38f);
            v1 = __builtin_amdgcn_fmed3f(v1, 0.0f, 3.0e38f);
        }
        o0[e] = (__bf16)v0;
        o1[e] = (__bf16)v1;
    }
    unsigned bits = 0;
    if (BITS) {
        // "activation != 0" of two bf16 at a time: v_pk_min_u16(pair, {1,1}) puts the flags at bits 0 and 16,
        // v_lshl_or_b32 files them at bits j and 16+j (j = dword 0..7 of o0 ++ o1).  One op per activation
        // instead of the v_cmp + v_cndmask + v_or3 (+ VCC stalls) hipcc makes of any C form.  The asm reads
        // the v_cvt_pk results, never an accumulator: hipcc does not pad MFMA -> VALU hazards for asm operands.
        typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
        const u32x4_ d0 = __builtin_bit_cast(u32x4_, o0), d1 = __builtin_bit_cast(u32x4_, o1);
        const unsigned ones = 0x00010001u;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            unsigned t;
            const unsigned d = j < 4 ? d0[j & 3] : d1[j & 3];
            asm("v_pk_min_u16 %0, %1, %2" : "=v"(t) : "v"(d), "s"(ones));
            asm("v_lshl_or_b32 %0, %1, %2, %0" : "+v"(bits) : "v"(t), "n"(j));
        }
    }
    return bits;     // tile-local: flags of acc[2j] at bit j, of acc[2j+1] at bit 16+j (acc[8..15]: j = 4..7)
}

// Tiles are buffered in GROUPS: one barrier + one prefetch burst per group of up to
// SLOT/CH output tiles (4 for a WxW layer), so the 8 waves run unsynchronised for ~64 MFMAs
// each and the next group's weights have a whole group of compute time to arrive.
__host__ __device__ constexpr int group_tiles(int nmt, int ch, int slot) {
    int g = nmt < slot / ch ? nmt : slot / ch;
    return (g > 1) ? (g & ~1) : g;          // even, so tiles can be processed in pairs
}

// One dense stage: NMT output tiles, inputs inA[NA] (C-perm) ++ inB[NB] (natural).
// Training stores (activation stash + ReLU mask) always trail the compute by one tile pair,
// also across stage boundaries (the previous stage's last pair is written after THIS stage's
// first barrier): a store issued right before a barrier would expose its full HBM latency,
// because the barrier's release waits for every outstanding memory operation of the wave.
template <int SLOT, int NA, int NB, int NMT, bool RELU, bool TRAIN, int PREV_NMT, bool STORE_OUT = true>
__device__ __forceinline__ void run_stage(WPipe& p, const bf16x8* inA, const bf16x8* inB,
                                          bf16x8* out, int next_stage_chunks, char* stash_dst,
                                          bool valid, const bf16x8* prev_out, char* prev_dst,
                                          char* prev_mask_dst, uint4& mask_carry) {
    // stash_dst / prev_dst / prev_mask_dst are WAVE-UNIFORM byte pointers (this wave's 32-sample
    // tile of one region); the per-lane part is lane*16.  Keeping them uniform keeps them in
    // SGPRs: as per-lane 64-bit pointers they spill, and a scratch reload next to an in-flight
    // global_load_lds makes hipcc drain the whole weight prefetch (vmcnt(0)).
    constexpr int CH = NA + NB + 1;
    constexpr int G = group_tiles(NMT, CH, SLOT);
    static_assert(NMT % 2 == 0 && G % 2 == 0, "stages processed by run_stage have an even tile count");
    unsigned mb[4] = {0u, 0u, 0u, 0u};
    const char* slot = nullptr;
#pragma unroll
    for (int mo = 0; mo < NMT; mo += 2) {
        if (mo % G == 0) {
            const int rest = NMT - mo - G;                       // tiles after this group
            slot = p.begin(rest > 0 ? (rest < G ? rest : G) * CH : next_stage_chunks);
        }
        f32x16 acc0, acc1;
        const char* s0 = slot + (mo % G) * CH * 1024;
        const char* s1 = slot + (mo % G + 1) * CH * 1024;
        if (TRAIN) {          // the previous pair's stores (this stage's, or the previous stage's last pair) ride inside the k-loop
            if (mo > 0 && STORE_OUT) {
                mma_tile2<NA, NB, true>(s0, s1, p.lane, inA, inB, acc0, acc1, out + 2 * mo - 4,
                                        stash_dst + (2 * mo - 4) * 1024 + p.lane * 16, valid);
                if (valid) p.since += 4;
            } else if (mo == 0 && PREV_NMT > 0) {
                // prev_dst == nullptr: the previous stage's output is not stashed (the linear bottleneck, whose
                // activations no weight-gradient GEMM reads: k_bottleneck_grads in mlp_bwd.hip)
                const bool st = valid && prev_dst != nullptr;
                mma_tile2<NA, NB, true>(s0, s1, p.lane, inA, inB, acc0, acc1, prev_out + 2 * PREV_NMT - 4,
                                        prev_dst + (2 * PREV_NMT - 4) * 1024 + p.lane * 16, st);
                if (st) p.since += 4;
                if (valid && prev_mask_dst) *(uint4*)(prev_mask_dst + p.lane * 16) = mask_carry;
            } else {
                mma_tile2<NA, NB>(s0, s1, p.lane, inA, inB, acc0, acc1);
            }
        } else {
            mma_tile2<NA, NB>(s0, s1, p.lane, inA, inB, acc0, acc1);
        }
        constexpr bool BITS = TRAIN && RELU;
        const unsigned bits0 = pack_tile<RELU, BITS>(acc0, out[2 * mo], out[2 * mo + 1]);
        const unsigned bits1 = pack_tile<RELU, BITS>(acc1, out[2 * mo + 2], out[2 * mo + 3]);
        if (TRAIN && RELU) mb[mo >> 1] |= bits0 | (bits1 << 8);      // second tile of the pair: bits 8-15 and 24-31
    }
    if (TRAIN && RELU) mask_carry = make_uint4(mb[0], mb[1], mb[2], mb[3]);
}


// ---------------------------------------------------------------------------
// M-split forward for the object MLPs (W = 128): latency per tile instead of work per wave.
//
// k_mlp_fwd gives every wave 32 samples and ALL output tiles of a layer: a block is a chain of 11 stages x 32 MFMAs per
// wave behind a workgroup barrier and a weight DMA each, ~2.3 us per stage whatever the occupancy -- and the object launches
// of a step are one round of a few dozen such blocks, i.e. pure latency (25-45 us per launch at the reference's 512-ray
// batch and at K = 8).  Here a workgroup is 4 waves x 64 samples (two 32-sample MFMA tiles); wave w owns output tile w of
// every layer, so a stage is 8-12 k-steps x 2 independent accumulators per wave.  Activations are exchanged through LDS
// as the very fragments the next stage's MFMAs read (the C-layout of tile w IS k-steps 2w, 2w+1 of the next B operand:
// mlp_spec.h), weights come straight from L2 (each wave reads only its own tile's 9-13 KB per stage: no LDS staging, no
// DMA waits), one barrier per stage.  Same MFMA instruction, same operands, same k order per output as k_mlp_fwd<128>:
// raw, encoding tile, stash, masks and view tile are BIT-identical (tests/test_gpu_fused_encode.py).
// ---------------------------------------------------------------------------
namespace ms {
using S = MlpSpec<128>;
constexpr int NT = 2;                                  // 32-sample tiles per workgroup
constexpr int X_BYTES = NT * S::KW * 1024;             // one activation fragment buffer: [tile][k-step][lane][16 B]
constexpr int OFF_X = 0;                               // two of them (written by stage s, read by stage s + 1)
constexpr int OFF_E = 2 * X_BYTES;                     // encoding fragments [tile][KE][lane][16 B] (natural order)
constexpr int OFF_V = OFF_E + NT * S::KE * 1024;       // view fragments     [tile][KV][lane][16 B]
constexpr int OFF_M = OFF_V + NT * S::KV * 1024;       // ReLU-flag pieces   [2][tile][wave][lane] u32
constexpr int LDS_BYTES = OFF_M + 2 * NT * 4 * 64 * 4;
}  // namespace ms

// One (object, pair of 32-sample tiles) item of the M-split forward on FOUR waves (wave = 0..3 within the group) and ms::LDS_BYTES
// of LDS at `smem`: the body of k_mlp_fwd_ms, and of the object half-workgroups of the mixed launch (k_mlp_fwd<.., MIX>).
// Every barrier inside is the WORKGROUP barrier: all groups of a workgroup call this function the same number of times.
struct MsFwd {
    size_t rows; int N; const bf16x8* enc; const bf16x8* view; const int32_t* ray_idx; const int32_t* count; const char* wpack;
    float* raw; bf16x8* stash; uint4* relu_mask; FwdStrides bs; EncIn ei; int nobj;
    int* ticket;         // mixed launch only: the item counter (durf::next_ticket)
};
__device__ __forceinline__ size_t ms_pairs_of(const MsFwd& A, int k) {
    const size_t c = (size_t)as_global(A.count)[k] * (size_t)A.N;
    return ((c < A.rows ? c : A.rows) + 32 * ms::NT - 1) / (32 * ms::NT);
}
// item -> (object, pair); false when item >= total
__device__ __forceinline__ bool ms_item(const MsFwd& A, size_t item, size_t& k_out, size_t& pair_out) {
    size_t k = 0, pair = item;
    for (; k < (size_t)A.nobj; k++) {
        const size_t np = ms_pairs_of(A, (int)k);
        if (pair < np) break;
        pair -= np;
    }
    const bool ok = k < (size_t)A.nobj;
    k_out = (size_t)__builtin_amdgcn_readfirstlane((unsigned)(ok ? k : 0));
    pair_out = ok ? pair : 0;
    return ok;
}

// Timing probe of one item (variant builds only: tools/build_variant.sh stamps -DDURF_MS_STAMPS, one -D per argument): role 0's lane 0 writes the
// 100 MHz s_memrealtime at the marks below into g_ms_stamps[item slot][32]; tools/experiments/ms_stamps.py reads them back.
#ifndef MS_PROBE_ONEW
#define MS_PROBE_ONEW 0          // timing probe only: ONE weight fragment per stage and wave instead of 9-13 (wrong results)
#endif
#ifndef MS_PROBE_NOSTORE
#define MS_PROBE_NOSTORE 0       // timing probe only: the item without its stash / mask stores (wrong results)
#endif
#if defined(DURF_MS_STAMPS)
__device__ unsigned long long g_ms_stamps[4096 * 32];
__device__ unsigned g_ms_n;
#define MS_STAMP(i) do { if (ms_probe) g_ms_stamps[ms_slot * 32 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define MS_STAMP_WAIT(i) do { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); MS_STAMP(i); } while (0)
#if defined(MS_STAMP_STAGE)      // the inside of ONE stage instead of the inside of the input phase (same slots 26-29)
#undef MS_STAMP_WAIT
#define MS_STAMP_WAIT(i) do { } while (0)
#define MS_STAMP_IN(s, i) do { if constexpr ((s) == MS_STAMP_STAGE) MS_STAMP(i); } while (0)
#define MS_STAMP_IN_WAITW(s, i) do { if constexpr ((s) == MS_STAMP_STAGE) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); MS_STAMP(i); } } while (0)
#define MS_STAMP_IN_ACC(s, i) do { if constexpr ((s) == MS_STAMP_STAGE) { float t0_, t1_; asm volatile("v_mov_b32 %0, %2\n\tv_mov_b32 %1, %3" : "=v"(t0_), "=v"(t1_) : "v"(acc[0][15]), "v"(acc[1][15])); MS_STAMP(i); } } while (0)
#define MS_STAMP_HO(j, i) do { if ((j) == MS_STAMP_STAGE) MS_STAMP(i); } while (0)
#else
#define MS_STAMP_IN(s, i) do { } while (0)
#define MS_STAMP_IN_WAITW(s, i) do { } while (0)
#define MS_STAMP_IN_ACC(s, i) do { } while (0)
#define MS_STAMP_HO(j, i) do { } while (0)
#endif
extern "C" int durf_debug_ms_stamps(void* dst, int reset) {
    unsigned n = 0;
    if (hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_ms_n), sizeof(n)) != hipSuccess) return -1;
    if (dst && hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_ms_stamps), sizeof(unsigned long long) * 4096 * 32) != hipSuccess) return -1;
    if (reset) { const unsigned z = 0; if (hipMemcpyToSymbol(HIP_SYMBOL(g_ms_n), &z, sizeof(z)) != hipSuccess) return -1; }
    return (int)n;
}
#else
#define MS_STAMP(i) do { } while (0)
#define MS_STAMP_WAIT(i) do { } while (0)
#define MS_STAMP_IN(s, i) do { } while (0)
#define MS_STAMP_IN_WAITW(s, i) do { } while (0)
#define MS_STAMP_IN_ACC(s, i) do { } while (0)
#define MS_STAMP_HO(j, i) do { } while (0)
#endif

template <bool TRAIN>
__device__ __forceinline__ void ms_fwd_pair(const MsFwd& A, char* smem, int lane, int wave, bool live, size_t k, size_t pair) {
    using S = ms::S;
    constexpr int NT = ms::NT;
    const int n = lane & 31, hi = lane >> 5;
    const size_t ntile32 = A.rows >> 5;
    const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    char* const X0 = smem + ms::OFF_X;
    char* const E = smem + ms::OFF_E;
    char* const V = smem + ms::OFF_V;
    unsigned* const M = (unsigned*)(smem + ms::OFF_M);
    // this object's slabs (0 strides for a single MLP)
    const size_t rows = A.rows;
    const int N = A.N;
    const bf16x8* __restrict__ view = as_global(A.view);
    const bf16x8* __restrict__ enc = as_global((const bf16x8*)((const char*)A.enc + k * A.bs.enc));
    const int32_t* __restrict__ ray_idx = as_global(A.ray_idx + k * A.bs.idx);
    const char* __restrict__ wpack = A.wpack + k * A.bs.wpack;
    float* __restrict__ raw = as_global((float*)((char*)A.raw + k * A.bs.raw));
    EncIn ei = A.ei;
    ei.t_vals = as_global(ei.t_vals); ei.origins_s = as_global(ei.origins_s); ei.dirs_s = as_global(ei.dirs_s); ei.radii = as_global(ei.radii);
    if (ei.view_tile) ei.view_tile = as_global((char*)ei.view_tile + k * ei.view_stride);
    bf16x8* __restrict__ stash = TRAIN ? as_global((bf16x8*)((char*)A.stash + k * A.bs.stash)) : nullptr;
    uint4* __restrict__ relu_mask = TRAIN ? as_global((uint4*)((char*)A.relu_mask + k * A.bs.mask)) : nullptr;
    const size_t c = (size_t)as_global(A.count)[k] * (size_t)N;
    const size_t nrows = c < rows ? c : rows;          // a multiple of 32 (N % 32 == 0)
    const size_t t32[NT] = {pair * NT, pair * NT + 1};
    // (a group without an item -- live == false: the other half of a mixed workgroup still has one -- runs the same stages
    // on zeros with every global load / store predicated off by these flags, so that both halves meet at every barrier)
    const bool tv[NT] = {live, live && t32[1] * 32 < nrows};
#if defined(DURF_MS_STAMPS)
    const bool ms_probe = live && wave == 0 && lane == 0;
    unsigned ms_slot = 0;
    if (ms_probe) {
        ms_slot = atomicAdd(&g_ms_n, 1u) & 4095u;
        g_ms_stamps[ms_slot * 32 + 30] = ((unsigned long long)blockIdx.x << 32) | (unsigned long long)(k * 100000 + pair);
        g_ms_stamps[ms_slot * 32 + 31] = (unsigned long long)lds_addr_of(smem);
    }
#endif
    MS_STAMP(0);
    ms_barrier();                               // the previous pair is done with the LDS
    MS_STAMP(1);
    // ---- inputs: the tiles' encodings (computed here or read), view directions ----
    if (ei.obj) {
        // lane = sample: 64 lanes = both tiles; the stand-alone encoder's body.  Every wave derives the sample's Gaussian (a few
        // loads that hit in L2 + ~0.5 us of arithmetic) and then produces TWO of the eight feature vectors, q = wave and
        // wave + 4 (MS_ENC_SPLIT, round 6: with all eight on wave 0 the features were 4.9 of an item's 21 us, three waves
        // waiting: tools/experiments/ms_stamps.py).  A feature is a function of the sample alone: the same bits whoever computes it.
        if (MS_ENC_SPLIT || wave == 0) {
            const int t = lane >> 5;
            const size_t row = t32[t] * 32 + n;
            if (tv[t]) {
                const int j = (int)(row / (size_t)N), nn = (int)(row % (size_t)N);
                const int b = ray_idx[j];
                MS_STAMP_WAIT(26);
                const float t0 = ei.t_vals[(size_t)b * (N + 1) + nn], t1 = ei.t_vals[(size_t)b * (N + 1) + nn + 1];
                float o[3] = {ei.origins_s[b * 3], ei.origins_s[b * 3 + 1], ei.origins_s[b * 3 + 2]};
                float d[3] = {ei.dirs_s[b * 3], ei.dirs_s[b * 3 + 1], ei.dirs_s[b * 3 + 2]};
                const float rad = ei.radii[b];
                MS_STAMP_WAIT(27);
                Gauss g = frustum_gaussian(t0, t1, o, d, rad, (ei.flags & DURF_ENC_CYLINDER) != 0);
                if (ei.flags & DURF_ENC_NO_INTEGRATION) g.var[0] = g.var[1] = g.var[2] = 0.0f;      // obbpose_model.py:164-165
#if defined(DURF_MS_STAMPS)
                asm volatile("" : "+v"(g.x[0]), "+v"(g.x[1]), "+v"(g.x[2]), "+v"(g.var[0]), "+v"(g.var[1]), "+v"(g.var[2]));
#endif
                MS_STAMP(28);
                BarfW bw;
#pragma unroll
                for (int i = 0; i < 10; i++) bw.w[i] = ei.w[i];
                char* const eg = (char*)enc + t32[t] * (S::KE * 1024);
                auto put = [&](auto q_, const bf16x8& o8) {
                    constexpr int q = decltype(q_)::value;             // features [8 q, 8 q + 8): k-step q / 2, half q % 2
                    const int off = (q >> 1) * 1024 + ((q & 1) * 32 + n) * 16;
                    *(bf16x8*)(E + t * (S::KE * 1024) + off) = o8;
                    *(DURF_G(bf16x8)*)(eg + off) = o8;      // the encoding tile the weight-gradient GEMMs of Dense_0 / Dense_5 read
                };
                if (!MS_ENC_SPLIT) lane_features<true>(g, bw, put);
                else if (wave == 0) lane_features<true, 0x11u>(g, bw, put);
                else if (wave == 1) lane_features<true, 0x22u>(g, bw, put);
                else if (wave == 2) lane_features<true, 0x44u>(g, bw, put);
                else lane_features<true, 0x88u>(g, bw, put);
                MS_STAMP(29);
            } else {
#pragma unroll
                for (int q = 0; q < 8; q++)
                    if (!MS_ENC_SPLIT || (q & 3) == wave) *(bf16x8*)(E + t * (S::KE * 1024) + (q >> 1) * 1024 + ((q & 1) * 32 + n) * 16) = zero8;
            }
        }
    } else {
        for (int ch = wave; ch < NT * S::KE; ch += 4) {                // chunk = (tile, k-step)
            const int t = ch / S::KE, k = ch % S::KE;
            *(bf16x8*)(E + ch * 1024 + lane * 16) =
                tv[t] ? *(const bf16x8*)((const char*)enc + t32[t] * (S::KE * 1024) + k * 1024 + lane * 16) : zero8;
        }
    }
    if (wave >= 2) {                               // waves 2, 3: the view-direction fragments of tile 0, 1
        const int t = wave - 2;
        const size_t row = t32[t] * 32 + n;
        size_t ray = 0;
        if (tv[t]) ray = (size_t)ray_idx[row / (size_t)N];
#pragma unroll
        for (int k = 0; k < S::KV; k++) {
            const bf16x8 v = tv[t] ? view[ray * (DURF_VIEW_DIM / 8) + 2 * k + hi] : zero8;
            *(bf16x8*)(V + (t * S::KV + k) * 1024 + lane * 16) = v;
            if (TRAIN && ei.view_tile && tv[t]) *(DURF_G(bf16x8)*)((char*)ei.view_tile + (t32[t] * S::KV + k) * 1024 + lane * 16) = v;
        }
    }

    // ---- one stage: this wave's output tile `mo` of forward stage s for both sample tiles ----
    // B operands: NX k-steps from the activation buffer Xin, then NE from the encoding, then NV from the view fragments
    f32x16 acc[NT];
    // Weights of (stage s, output tile mo): T A-fragments + the 16 bias values of this lane, straight from L2 into registers.
    // They are requested ONE STAGE AHEAD (two register sets, alternating), so a stage never waits for its own loads.
    struct WSet { bf16x8 A[S::KW + S::KE]; f32x4 b[4]; };         // stages 5 and 9 (12 / 10 k-steps) use the odd set
    struct WSet8 { bf16x8 A[S::KW]; f32x4 b[4]; };                // the even stages have at most KW k-steps
    auto load_w = [&](auto s_, int mo, auto& w) {
        constexpr int s = decltype(s_)::value, T = S::n_ks(s);
        // (a GLOBAL pointer by type, round 6: behind the opaque asm a generic pointer made these FLAT loads, which count on
        // lgkmcnt as well -- every ms_barrier then waited for the next stage's weights it was meant to leave in flight)
        typedef const __attribute__((address_space(1))) char* gptr_t;
        gptr_t wt = (gptr_t)(wpack + (size_t)(S::stage_chunk_base(s) + mo * S::tile_chunks(s)) * 1024);     // wave-uniform
        // (opaque: the stream is read-only, so hipcc would otherwise hoist EVERY later stage's loads above the barriers in
        // between -- the inference instantiation needed all 512 registers and still spilled)
        asm volatile("" : "+s"(wt));
#pragma unroll
        for (int k = 0; k < T; k++) w.A[k] = *(const __attribute__((address_space(1))) bf16x8*)(wt + (MS_PROBE_ONEW ? 0 : k) * 1024 + lane * 16);
        const __attribute__((address_space(1))) f32x4* bp = (const __attribute__((address_space(1))) f32x4*)(wt + T * 1024 + hi * 64);     // the bias rows: the initial accumulators
#pragma unroll
        for (int g = 0; g < 4; g++) w.b[g] = bp[g];
    };
    auto stage_mma = [&](auto s_, const auto& w, auto nx_, auto ne_, auto nv_, const char* Xin) {
        constexpr int s = decltype(s_)::value, NX = decltype(nx_)::value, NE = decltype(ne_)::value, NV = decltype(nv_)::value;
        constexpr int T = NX + NE + NV;
        static_assert(T == S::n_ks(s), "k-steps of the stage");
#pragma unroll
        for (int g = 0; g < 4; g++) {
#pragma unroll
            for (int t = 0; t < NT; t++) { acc[t][4 * g] = w.b[g][0]; acc[t][4 * g + 1] = w.b[g][1]; acc[t][4 * g + 2] = w.b[g][2]; acc[t][4 * g + 3] = w.b[g][3]; }
        }
#if MS_RING > 0
        // The B fragments through a ring of MS_RING explicit LDS reads with counted waits (round 6: left to hipcc, the object
        // half of a mixed workgroup -- at the 256-register cap -- got ONE fragment register: read, wait, MFMA, sixteen times a
        // stage, ~1.3 us of LDS latency per stage: tools/experiments/ms_stamps.py).  Same fragments, same order.
        static_assert(NT == 2, "the ring alternates the two sample tiles");
        constexpr int NR = 2 * T, D = MS_RING < NR ? MS_RING : NR;
        const unsigned ax = lds_addr_of(Xin) + lane * 16, ae = lds_addr_of(E) + lane * 16, av = lds_addr_of(V) + lane * 16;
        auto rd = [&](auto m_) -> v4i_ {
            constexpr int m = decltype(m_)::value, k = m >> 1, t = m & 1;
            if constexpr (k < NX) return lds_read16<(t * S::KW + k) * 1024>(ax);
            else if constexpr (k < NX + NE) return lds_read16<(t * S::KE + (k - NX)) * 1024>(ae);
            else return lds_read16<(t * S::KV + (k - NX - NE)) * 1024>(av);
        };
        v4i_ ring[D];
        static_for<0, D>([&](auto i_) { ring[decltype(i_)::value] = rd(i_); });
        MS_STAMP_IN(s, 26);
        static_for<0, NR>([&](auto i_) {
            constexpr int i = decltype(i_)::value, k = i >> 1, t = i & 1;
            if constexpr (i == 0) MS_STAMP_IN_WAITW(s, 27);
            constexpr int later = (NR - 1 - i) < (D - 1) ? (NR - 1 - i) : (D - 1);
            lds_wait<later>(ring[i % D]);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.A[k], __builtin_bit_cast(bf16x8, ring[i % D]), acc[t], 0, 0, 0);
            if constexpr (i + D < NR) ring[i % D] = rd(std::integral_constant<int, i + D>{});
        });
        MS_STAMP_IN_ACC(s, 28);
#else
#pragma unroll
        for (int k = 0; k < T; k++) {
#pragma unroll
            for (int t = 0; t < NT; t++) {
                const char* src = k < NX ? Xin + (t * S::KW + k) * 1024
                                         : (k < NX + NE ? E + (t * S::KE + (k - NX)) * 1024 : V + (t * S::KV + (k - NX - NE)) * 1024);
                const bf16x8 b = *(const bf16x8*)(src + lane * 16);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.A[k], b, acc[t], 0, 0, 0);
            }
        }
#endif
    };
    // epilogue of a stashed ReLU stage: fragments 2 mo, 2 mo + 1 of the next stage's input, the stash, the flag pieces
    auto hand_over = [&](auto relu_, int mo, char* Xout, int jstash, unsigned* Mbuf) {
        constexpr bool RELU = decltype(relu_)::value;
#pragma unroll
        for (int t = 0; t < NT; t++) {
            bf16x8 o0, o1;
            const unsigned bits = pack_tile<RELU, TRAIN && RELU>(acc[t], o0, o1);
            *(bf16x8*)(Xout + (t * S::KW + 2 * mo) * 1024 + lane * 16) = o0;
            *(bf16x8*)(Xout + (t * S::KW + 2 * mo + 1) * 1024 + lane * 16) = o1;
            if (TRAIN && RELU && tv[t] && !MS_PROBE_NOSTORE) {
                char* sd = (char*)stash + ((size_t)S::stash_ks_before(jstash) * ntile32 + t32[t] * S::stash_ks(jstash)) * 1024;
                STREAM_STORE(sd + (2 * mo) * 1024 + lane * 16, o0);
                STREAM_STORE(sd + (2 * mo + 1) * 1024 + lane * 16, o1);
                Mbuf[(t * 4 + mo) * 64 + lane] = bits << (8 * (mo & 1));
            }
        }
        MS_STAMP_HO(jstash, 29);
    };
    // waves 0 / 1 assemble the previous stage's flags of tile 0 / 1 into k_mlp_fwd's layout (one uint4 per lane: words
    // 0, 1 = tile pairs (0,1), (2,3)) -- after the barrier that made every wave's piece visible
    auto flush_mask = [&](int jmask, const unsigned* Mbuf) {
        if (TRAIN && wave < NT && tv[wave] && !MS_PROBE_NOSTORE) {
            const unsigned* m = Mbuf + (wave * 4) * 64 + lane;
            const uint4 w4 = make_uint4(m[0] | m[64], m[128] | m[192], 0u, 0u);
            typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
            const u32x4_ w4v = {w4.x, w4.y, w4.z, w4.w};
            *(DURF_G(u32x4_)*)((char*)relu_mask + ((size_t)jmask * ntile32 + t32[wave]) * 1024 + lane * 16) = w4v;
        }
    };
    using I0 = std::integral_constant<int, 0>;
    char* Xa = X0;
    char* Xb = X0 + ms::X_BYTES;
    unsigned* Ma = M;
    unsigned* Mb = M + NT * 4 * 64;
    WSet8 w0;                                      // two alternating weight sets
    WSet w1;
    load_w(std::integral_constant<int, 0>{}, wave, w0);
    MS_STAMP(2);
    ms_barrier();                               // encodings and view fragments are in place
    MS_STAMP(3);
    using IE = std::integral_constant<int, S::KE>;
    using IW = std::integral_constant<int, S::KW>;
    auto swap = [&]() { char* tx = Xa; Xa = Xb; Xb = tx; unsigned* tm = Ma; Ma = Mb; Mb = tm; };
    // stage 0: enc -> Xa
    load_w(std::integral_constant<int, 1>{}, wave, w1);
    stage_mma(std::integral_constant<int, 0>{}, w0, I0{}, IE{}, I0{}, Xa);
    hand_over(std::true_type{}, wave, Xa, 0, Ma);
    MS_STAMP(16);
    ms_barrier();
    MS_STAMP(4);
    // stages 1-4 (the next stage's weights are requested before this stage's MFMAs)
    flush_mask(0, Ma);
    load_w(std::integral_constant<int, 2>{}, wave, w0);
    stage_mma(std::integral_constant<int, 1>{}, w1, IW{}, I0{}, I0{}, Xa);
    hand_over(std::true_type{}, wave, Xb, 1, Mb);
    MS_STAMP(17);
    ms_barrier();
    MS_STAMP(5);
    swap();
    flush_mask(1, Ma);
    load_w(std::integral_constant<int, 3>{}, wave, w1);
    stage_mma(std::integral_constant<int, 2>{}, w0, IW{}, I0{}, I0{}, Xa);
    hand_over(std::true_type{}, wave, Xb, 2, Mb);
    MS_STAMP(18);
    ms_barrier();
    MS_STAMP(6);
    swap();
    flush_mask(2, Ma);
    load_w(std::integral_constant<int, 4>{}, wave, w0);
    stage_mma(std::integral_constant<int, 3>{}, w1, IW{}, I0{}, I0{}, Xa);
    hand_over(std::true_type{}, wave, Xb, 3, Mb);
    MS_STAMP(19);
    ms_barrier();
    MS_STAMP(7);
    swap();
    flush_mask(3, Ma);
    load_w(std::integral_constant<int, 5>{}, wave, w1);
    stage_mma(std::integral_constant<int, 4>{}, w0, IW{}, I0{}, I0{}, Xa);
    hand_over(std::true_type{}, wave, Xb, 4, Mb);
    MS_STAMP(20);
    ms_barrier();
    MS_STAMP(8);
    swap();
    // stage 5: [h4, enc] (obbpose_model.py:333-334)
    flush_mask(4, Ma);
    load_w(std::integral_constant<int, 6>{}, wave, w0);
    stage_mma(std::integral_constant<int, 5>{}, w1, IW{}, IE{}, I0{}, Xa);
    hand_over(std::true_type{}, wave, Xb, 5, Mb);
    MS_STAMP(21);
    ms_barrier();
    MS_STAMP(9);
    swap();
    // stages 6, 7
    flush_mask(5, Ma);
    load_w(std::integral_constant<int, 7>{}, wave, w1);
    stage_mma(std::integral_constant<int, 6>{}, w0, IW{}, I0{}, I0{}, Xa);
    hand_over(std::true_type{}, wave, Xb, 6, Mb);
    MS_STAMP(22);
    ms_barrier();
    MS_STAMP(10);
    swap();
    flush_mask(6, Ma);
    load_w(std::integral_constant<int, 8>{}, wave, w0);
    stage_mma(std::integral_constant<int, 7>{}, w1, IW{}, I0{}, I0{}, Xa);
    hand_over(std::true_type{}, wave, Xb, 7, Mb);
    MS_STAMP(23);
    ms_barrier();
    MS_STAMP(11);
    swap();
    // stage 8: h7 -> bottleneck (linear, tile `wave`) and, wave 0, the density head (tile WT)
    flush_mask(7, Ma);
    float dens[NT] = {0.0f, 0.0f};
    if (wave == 0) {
        // Wave 0 also owns the density head's tile (WT).  Its weights take the ODD set's registers for this stage and stage
        // 9's follow behind them (round 6; until then a third set, requested two stages ahead, stayed live across three
        // stages: 302 registers -- with two sets the body needs 182-220, which is what lets it run as the object half of a
        // mixed workgroup beside the persistent background body's 256, k_mlp_fwd<.., MIX>).  Same MFMAs, same operands.
        load_w(std::integral_constant<int, 8>{}, S::WT, w1);
        stage_mma(std::integral_constant<int, 8>{}, w0, IW{}, I0{}, I0{}, Xa);
        hand_over(std::false_type{}, wave, Xb, 8, Mb);
        stage_mma(std::integral_constant<int, 8>{}, w1, IW{}, I0{}, I0{}, Xa);
#pragma unroll
        for (int t = 0; t < NT; t++) dens[t] = acc[t][0];
        load_w(std::integral_constant<int, 9>{}, wave, w1);
    } else {
        load_w(std::integral_constant<int, 9>{}, wave, w1);
        stage_mma(std::integral_constant<int, 8>{}, w0, IW{}, I0{}, I0{}, Xa);
        hand_over(std::false_type{}, wave, Xb, 8, Mb);
    }
    MS_STAMP(24);
    ms_barrier();
    MS_STAMP(12);
    { char* tx = Xa; Xa = Xb; Xb = tx; }
    // stage 9: [bottleneck, view] -> hc (128, relu); its flags go to mask region 8
    if (wave == 0) load_w(std::integral_constant<int, 10>{}, 0, w0);
    stage_mma(std::integral_constant<int, 9>{}, w1, IW{}, I0{}, std::integral_constant<int, S::KV>{}, Xa);
    hand_over(std::true_type{}, wave, Xb, 9, Mb);
    MS_STAMP(25);
    ms_barrier();
    MS_STAMP(13);
    flush_mask(8, Mb);
    // stage 10: hc -> rgb (wave 0), raw = (rgb, density)
    if (wave == 0) {
        // jnp.maximum propagates NaN, v_max_f32 does not: a non-finite encoding poisons the sample's output (as k_mlp_fwd:
        // the lane's 8 features of each k-step, then the sample's other half)
        bool bad[NT] = {false, false};
#pragma unroll
        for (int t = 0; t < NT; t++)
#pragma unroll
            for (int k = 0; k < S::KE; k++) {
                const bf16x8 e8 = *(const bf16x8*)(E + (t * S::KE + k) * 1024 + lane * 16);
#pragma unroll
                for (int e = 0; e < 8; e++) bad[t] |= !(fabsf((float)e8[e]) <= 3.0e38f);
            }
#pragma unroll
        for (int t = 0; t < NT; t++) bad[t] |= (__shfl_xor((int)bad[t], 32, 64) != 0);
        stage_mma(std::integral_constant<int, 10>{}, w0, std::integral_constant<int, S::KC>{}, I0{}, I0{}, Xb);
        if (lane < 32) {
            const float qn = __builtin_nanf("");
#pragma unroll
            for (int t = 0; t < NT; t++) {
                if (!tv[t]) continue;
                const f32x4 o = {bad[t] ? qn : acc[t][0], bad[t] ? qn : acc[t][1], bad[t] ? qn : acc[t][2], bad[t] ? qn : dens[t]};
                *(DURF_G(f32x4)*)(raw + (t32[t] * 32 + n) * 4) = o;
            }
        }
    }
    MS_STAMP(14);
}

template <bool TRAIN>
__global__ void __launch_bounds__(256)      // one wave per SIMD: the register file is this workgroup's (latency, not occupancy)
k_mlp_fwd_ms(MsFwd A) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // The work items are (object, pair of 32-sample tiles); the hit counts live on the device.  A 1-D grid deals them in
    // order -- object 0's pairs, object 1's, ... -- so the workgroups that find work are exactly the first sum(pairs) of the
    // grid.  (With a (object, pair) grid the first 256 workgroups dispatched covered only pairs < 256 / K of every object: at
    // K = 8 an object with more than 32 pairs waited for a second round of CUs although fewer than 256 workgroups had work --
    // 47 us a launch for 26 us of work.)
    size_t total = 0;
    for (int k = 0; k < A.nobj; k++) total += ms_pairs_of(A, k);
    for (size_t item = blockIdx.x; item < total; item += gridDim.x) {
        size_t k, pair;
        ms_item(A, item, k, pair);
        ms_fwd_pair<TRAIN>(A, smem, lane, wave, true, k, pair);
    }
}


// ENC (W = 256): the workgroup ENCODES its own tiles -- each lane computes its sample's 60 features from the ray data
// (enc_lane.h, the body of k_encode_lane<false>: bit-identical features), keeps the half its MFMA fragment holds and
// writes the tile to `enc`, where stage 5 (the skip connection) and the weight-gradient GEMMs of Dense_0 / Dense_5 read
// it.  Replaces the durf_encode_bkgd launch in front of every forward: one launch, one 64 MB write + read and ~30 us of
// launch gaps less per level at 4096 rays.
// MIX (W = 256, 8 waves; round 6): ONE heterogeneous persistent launch for a small step -- every workgroup first walks its
// background blocks as before, then turns into TWO 4-wave groups that take (object, tile pair) items of the K object MLPs
// (ms_fwd_pair: the body of k_mlp_fwd_ms, bit-identical outputs) off an atomic ticket counter until none is left.  At the
// reference's 512-ray batch the de-duplicated background grid leaves ~24 of the 256 workgroups without a block: they start
// on the object items at once and are done with them inside the background blocks' 77 us (a launch of their own behind
// the background's cost 22 us per level; on a second stream the early object workgroups delayed the persistent
// background workgroups that wanted their CUs); at 1024 rays x K = 8 the workgroups with one block instead of two pick
// them up.  `ow.ticket`: a zeroed int the launch leaves zeroed (the workgroup that draws the last ticket resets it).
// (the kernel's explicit arguments as the kernarg segment lays them out -- in order, naturally aligned: the mixed launch reads
// its object arguments from the segment itself, see below)
struct FwdKernArgs {
    size_t rows; int N; const bf16x8* enc; const bf16x8* view; const int32_t* ray_idx; const int32_t* count; const char* wpack;
    float* raw; bf16x8* stash; uint4* relu_mask; FwdStrides bs; const int32_t* tail_idx; const int32_t* tail_count; EncIn ei;
    MsFwd ow;
};
// The object phase of a mixed workgroup, inlined behind the background loop.  Two things keep it from costing that loop --
// which sits at the 256-register cap -- anything: (i) nothing of it lives in a vector register across the loop (the lane number
// is re-derived, everything else is wave-uniform), (ii) its ~70 dwords of arguments are fetched from the kernarg segment
// HERE, behind an opaque pointer, instead of at kernel entry (as ordinary arguments hipcc keeps them in scalar registers
// across the loop: 85 more SGPR spills, two more vector registers reserved for them).  What remains is one more vector
// register of SGPR spill lanes than the plain kernel has: 28 B of scratch instead of 12, six reloads per 256-sample block.
// (As a real CALL -- own register allocation, the loop untouched -- the phase needs a 360-byte frame for the callee-saved
// registers, and a launch with that much scratch per lane took ~23 us longer whatever it did: profiles/r06_mix.txt.)
template <bool TRAIN>
__device__ __forceinline__ void mix_object_items(char* smem, int wave, int nwg) {
    typedef const __attribute__((address_space(4))) char* kptr_t;
    kptr_t ka = (kptr_t)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(ka));          // (opaque: the loads below stay below)
    const unsigned smem_lds = lds_addr_of(smem);
    MsFwd ow;
    load_kernarg(ow, ka + offsetof(FwdKernArgs, ow));      // (scalar loads: a constant-address-space source)
    const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    // (the ticket requests are wave 1's: wave 0 starts every item with the ray loads of the encoding / the head gradients, and
    // a returning atomic ahead of them in its queue would be waited for with them)
    const bool first = wave == MIX_TICKET_WAVE && lane == 0;
    // (waves w and w + 4 share a SIMD: the second group's roles are rotated by two, so that the two groups' role-0 waves --
    // which carry an item's extra work: the density and rgb heads (and, until the encoding was dealt to all four waves, the
    // whole input phase) -- run on different SIMDs)
    const int half = wave >> 2, w4 = (wave + MIX_ROLE_ROT * half) & 3;
    char* const lds = smem + half * ms::LDS_BYTES;
    volatile __attribute__((address_space(3))) int* const tk =
        (volatile __attribute__((address_space(3))) int*)(size_t)(__builtin_amdgcn_readfirstlane(smem_lds) + 2u * ms::LDS_BYTES);
    // (the objects' pair counts once per workgroup, in LDS: re-read from memory for every item they were a chain of K
    // dependent loads in front of it)
    volatile __attribute__((address_space(3))) int* const npl = tk + 4;
    if (wave == 0 && lane < ow.nobj) npl[lane] = (int)ms_pairs_of(ow, lane);
    ms_barrier();
    size_t total = 0;
    for (int k = 0; k < ow.nobj; k++) total += (size_t)npl[k];
    total = (size_t)__builtin_amdgcn_readfirstlane((unsigned)total);
    const int last = 2 * (int)((total + 1) / 2 + nwg - 1);            // the value the LAST request of the launch returns
    int t = 0;
    // (a GLOBAL atomic: a flat one counts on lgkmcnt, and the first barrier of the item would wait for the request under way)
    DURF_G(int)* const ticket = (DURF_G(int)*)ow.ticket;
    if (first) { t = __hip_atomic_fetch_add(ticket, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); *tk = t; }
    ms_barrier();
    t = __builtin_amdgcn_readfirstlane(*tk);
    while ((size_t)t < total) {
        int tn = 0;
        if (first) tn = __hip_atomic_fetch_add(ticket, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // the next request is under way while this item runs
        size_t k = 0, pair = (size_t)t + (size_t)half;
        const bool live = pair < total;
        for (; live && k + 1 < (size_t)ow.nobj; k++) {
            const size_t np = (size_t)npl[k];
            if (pair < np) break;
            pair -= np;
        }
        k = (size_t)__builtin_amdgcn_readfirstlane((unsigned)(live ? k : 0));
        pair = live ? pair : 0;
        ms_fwd_pair<TRAIN>(ow, lds, lane, w4, live, k, pair);
        if (first) *tk = tn;
        ms_barrier();
        t = __builtin_amdgcn_readfirstlane(*tk);
    }
    if (first && t == last) __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // every other workgroup has made its last request
}

template <int W, bool TRAIN, int NWV = 8, bool ENC = false, bool MIX = false>
__global__ void __launch_bounds__(512, 2)
k_mlp_fwd(size_t rows, int N, const bf16x8* __restrict__ enc, const bf16x8* __restrict__ view,
          const int32_t* __restrict__ ray_idx, const int32_t* __restrict__ count,
          const char* __restrict__ wpack, float* __restrict__ raw, bf16x8* __restrict__ stash,
          uint4* __restrict__ relu_mask, FwdStrides bs, const int32_t* __restrict__ tail_idx,
          const int32_t* __restrict__ tail_count, EncIn ei, MsFwd ow_arg) {
    static_assert(!MIX || (W == 256 && NWV == 8 && ENC), "the mixed launch: background blocks of 8 waves + object items on 2 x 4");
    (void)ow_arg;      // (read through the kernarg segment pointer behind the background loop, not held in SGPRs across it)
    using S = MlpSpec<W>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (gridDim.y > 1) {                             // batched object MLPs: this workgroup's object slab
        const size_t k = blockIdx.y;
        enc = (const bf16x8*)((const char*)enc + k * bs.enc);
        ray_idx += k * bs.idx;
        count += k;
        wpack += k * bs.wpack;
        raw = (float*)((char*)raw + k * bs.raw);
        if (ENC && ei.view_tile) ei.view_tile = (char*)ei.view_tile + k * ei.view_stride;
        if (TRAIN) {
            stash = (bf16x8*)((char*)stash + k * bs.stash);
            relu_mask = (uint4*)((char*)relu_mask + k * bs.mask);
        }
    }
    size_t nrows = rows;
    if (count) {
        const size_t c = (size_t)(*count) * (size_t)N;
        nrows = c < rows ? c : rows;
    }
    // Tail rows (de-duplicated background evaluation, include/durf_hip.h durf_expand_raw): after the count*N rows of
    // the rays evaluated sample by sample come *tail_count rows, ONE per box-hit ray tail_idx[i], whose trunk input is
    // the constant encoding of a zero-masked Gaussian ([0 x 30, 1 x 30]) and whose view direction is that ray's.
    const size_t nrows_c = nrows;                             // a multiple of 32 whenever a tail is given (N % 32 == 0)
    if (tail_count) {
        const size_t t = nrows_c + (size_t)(*tail_count);
        nrows = t < rows ? t : rows;
    }
    bool has_block = (size_t)blockIdx.x * (32 * NWV) < nrows;
    if (MIX && (ei.flags & (1 << 30))) { has_block = false; nrows = 0; }      // (DURF_MIX_PROBE=2: timing probe, object items only)
    if (!MIX && !has_block) return;                          // whole workgroup idle
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // Waves w and w+4 share a SIMD; static priority for the younger half staggers them so one
    // wave's epilogue can run under its partner's MFMAs.
    if (wave >= 4 && has_block) __builtin_amdgcn_s_setprio(1);
    const size_t ntile32 = rows >> 5;
    const size_t nblk = (nrows + 32 * NWV - 1) / (32 * NWV);

    WPipe p;
    constexpr int SLOT = 4 * (S::KW + 1);            // chunks per LDS slot (two slots)
    p.rsrc = make_rsrc(wpack);
    p.gnext = 0; p.lds = smem; p.slot_bytes = SLOT * 1024; p.par = 0;
    p.lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    p.wave = wave; p.lane = lane; p.nw = NWV;
    // prologue: first tile group of stage 0 -> slot 0
    constexpr int G0 = group_tiles(S::WT, S::KE + 1, SLOT) * (S::KE + 1);
    if (!MIX || has_block) p.issue(0, G0);

  // Persistent workgroup: one CU holds one workgroup (136 KB of LDS), so looping over the
  // 256-sample blocks here instead of relaunching hides every block's start-up (first weight
  // group + encoding fetch) behind the previous block's last stages.
  for (size_t blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
    const bool has_next = blk + gridDim.x < nblk;
    const size_t tile32 = blk * NWV + wave;
    const size_t row = tile32 * 32 + (lane & 31);
    const bool valid = row < nrows;
    const bool tile_valid = tile32 * 32 < nrows;

    const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    const bool tail = tail_count != nullptr && tile32 * 32 >= nrows_c;   // wave-uniform
    // natural-order fragment of the constant encoding: feature 16 k + 8 hi + e is 1 for the 30 cosine features
    auto const_enc = [&](int k) -> bf16x8 {
        bf16x8 v;
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const int f = 16 * k + 8 * (lane >> 5) + e;
            v[e] = (f >= 30 && f < 60) ? (__bf16)1.0f : (__bf16)0.0f;
        }
        return v;
    };
    bf16x8 encf[S::KE];
    const char* enc_u = (const char*)enc + tile32 * (S::KE * 1024);      // wave-uniform
    if constexpr (ENC) {
#pragma unroll
        for (int k = 0; k < S::KE; k++) encf[k] = (tile_valid && tail) ? const_enc(k) : zero8;
        if (tile_valid && !tail) {               // (a non-tail tile is whole: count * N and rows are multiples of 32)
            const int j = (int)(row / (size_t)N), n = (int)(row % (size_t)N);
            const int b = ray_idx ? ray_idx[j] : j;
            const bool hi = lane >= 32;          // this lane's fragment of k-step k: features [16 k + 8 hi, + 8) = vector q = 2 k + hi
            auto keep = [&](auto q_, const bf16x8& o8) {
                constexpr int q = decltype(q_)::value;
                if (hi == (bool)(q & 1)) encf[q >> 1] = o8;
            };
            if constexpr (W == 128) {            // object MLP: k_encode_lane<true>'s Gaussian (object frame, no masking, no contraction)
                const float t0 = ei.t_vals[(size_t)b * (N + 1) + n], t1 = ei.t_vals[(size_t)b * (N + 1) + n + 1];
                float o[3] = {ei.origins_s[b * 3], ei.origins_s[b * 3 + 1], ei.origins_s[b * 3 + 2]};
                float d[3] = {ei.dirs_s[b * 3], ei.dirs_s[b * 3 + 1], ei.dirs_s[b * 3 + 2]};
                Gauss g = frustum_gaussian(t0, t1, o, d, ei.radii[b], (ei.flags & DURF_ENC_CYLINDER) != 0);
                if (ei.flags & DURF_ENC_NO_INTEGRATION) g.var[0] = g.var[1] = g.var[2] = 0.0f;      // obbpose_model.py:164-165
                BarfW bw;
#pragma unroll
                for (int i = 0; i < 10; i++) bw.w[i] = ei.w[i];
                lane_features<true>(g, bw, keep);
            } else {
                const Gauss g = bkgd_sample_gaussian(b, n, N, ei.t_vals, ei.origins_s, ei.dirs_s, ei.radii, ei.hit, ei.K, ei.flags);
                lane_features<false>(g, BarfW{}, keep);
            }
        }
        if (tile_valid && (TRAIN || !tail)) {    // stage 5 re-reads the tile; training: so do the weight-gradient GEMMs
#pragma unroll
            for (int k = 0; k < S::KE; k++) *(bf16x8*)(const_cast<char*>(enc_u) + k * 1024 + lane * 16) = encf[k];
            p.since += S::KE;
        }
    } else {
#pragma unroll
        for (int k = 0; k < S::KE; k++)
            encf[k] = tile_valid ? (tail ? const_enc(k) : *(const bf16x8*)(enc_u + k * 1024 + lane * 16)) : zero8;
        if (TRAIN && tail && tile_valid) {       // the weight-gradient GEMMs of Dense_0 / Dense_5 read the encoding tile
#pragma unroll
            for (int k = 0; k < S::KE; k++) *(bf16x8*)(const_cast<char*>(enc_u) + k * 1024 + lane * 16) = encf[k];
            p.since += S::KE;
        }
    }
    // jnp.maximum propagates NaN through every ReLU, v_max_f32 does not.  The only source of
    // non-finite values is the encoding of a garbage (multi-hit) ray, and one non-finite
    // feature makes the reference's MLP output NaN: detect it once here and poison the output.
    bool poison = false;
#pragma unroll
    for (int k = 0; k < S::KE; k++)
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const float f = (float)encf[k][e];
            poison |= !(fabsf(f) <= 3.0e38f);
        }
    poison |= (__shfl_xor((int)poison, 32, 64) != 0);      // the sample's other 32 features

    bf16x8 a[S::KW], b[S::KW];
    auto stash_at = [&](int j) -> char* {      // wave-uniform base of this wave's tile in region j
        return (char*)stash + ((size_t)S::stash_ks_before(j) * ntile32 + tile32 * S::stash_ks(j)) * 1024;
    };
    // ReLU bit-masks for the backward: region j (stages 0..7 -> 0..7, stage 9 -> 8), one uint4 per lane
    auto mask_at = [&](int j) -> char* {
        return relu_mask ? (char*)relu_mask + ((size_t)j * ntile32 + tile32) * 1024 : nullptr;
    };
    constexpr int CHW = S::KW + 1;
    constexpr int GW = group_tiles(S::WT, CHW, SLOT) * CHW;                       // first group of a WxW stage
    constexpr int G5 = group_tiles(S::WT, S::KW + S::KE + 1, SLOT) * (S::KW + S::KE + 1);
    constexpr int G9 = group_tiles(S::CT, S::KW + S::KV + 1, SLOT) * (S::KW + S::KV + 1);
    uint4 mcarry = make_uint4(0u, 0u, 0u, 0u);
#define ST(j) (TRAIN ? stash_at(j) : nullptr)
#define MK(j) (TRAIN ? mask_at(j) : nullptr)
    // stage 0: enc -> a
    run_stage<SLOT, 0, S::KE, S::WT, true, TRAIN, 0>(p, nullptr, encf, a, GW, ST(0), tile_valid, nullptr, nullptr, nullptr, mcarry);
    // stages 1-4
    run_stage<SLOT, S::KW, 0, S::WT, true, TRAIN, S::WT>(p, a, nullptr, b, GW, ST(1), tile_valid, a, ST(0), MK(0), mcarry);
    run_stage<SLOT, S::KW, 0, S::WT, true, TRAIN, S::WT>(p, b, nullptr, a, GW, ST(2), tile_valid, b, ST(1), MK(1), mcarry);
    run_stage<SLOT, S::KW, 0, S::WT, true, TRAIN, S::WT>(p, a, nullptr, b, GW, ST(3), tile_valid, a, ST(2), MK(2), mcarry);
    run_stage<SLOT, S::KW, 0, S::WT, true, TRAIN, S::WT>(p, b, nullptr, a, G5, ST(4), tile_valid, b, ST(3), MK(3), mcarry);
    // stage 5: [a, enc] -> b.  The encoding fragments are re-read here (L2-resident) rather than
    // held in 16 VGPRs across stages 1-4, where the kernel sits at the 256-register cap.
    bf16x8 encs[S::KE];
#pragma unroll
    for (int k = 0; k < S::KE; k++)
        encs[k] = tile_valid ? (tail ? const_enc(k) : *(const bf16x8*)(enc_u + k * 1024 + lane * 16)) : zero8;
    run_stage<SLOT, S::KW, S::KE, S::WT, true, TRAIN, S::WT>(p, a, encs, b, GW, ST(5), tile_valid, a, ST(4), MK(4), mcarry);
    // stages 6, 7
    run_stage<SLOT, S::KW, 0, S::WT, true, TRAIN, S::WT>(p, b, nullptr, a, GW, ST(6), tile_valid, b, ST(5), MK(5), mcarry);
    run_stage<SLOT, S::KW, 0, S::WT, true, TRAIN, S::WT>(p, a, nullptr, b, GW, ST(7), tile_valid, a, ST(6), MK(6), mcarry);
    // stage 8: b -> bottleneck (a, linear) + density
    run_stage<SLOT, S::KW, 0, S::WT, false, TRAIN, S::WT, false>(p, b, nullptr, a, CHW, nullptr, tile_valid, b, ST(7), MK(7), mcarry);
    float dens;
    {
        const char* slot = p.begin(G9);
        const f32x16 acc = mma_tile<S::KW, 0>(slot, lane, b, nullptr);
        dens = acc[0];
    }
    // stage 9: [bottleneck, view] -> c (128, relu)
    bf16x8 vf[S::KV];
    {
        size_t ray = row / (size_t)N;
        if (tail) ray = valid ? (size_t)tail_idx[row - nrows_c] : 0;
        else if (ray_idx && valid) ray = (size_t)ray_idx[ray];
        unsigned hi_v = (unsigned)lane;               // opaque: kept per block, not hoisted as a spilled 64-bit pointer
        if (ENC) asm volatile("" : "+v"(hi_v));       // (the lane itself: `lane >> 5` would be hoisted and spilled too)
        hi_v >>= 5;
#pragma unroll
        for (int k = 0; k < S::KV; k++)
            vf[k] = valid ? view[ray * (DURF_VIEW_DIM / 8) + 2 * k + hi_v] : zero8;
        if constexpr (ENC) {     // the view-direction tile of the weight-gradient GEMM: this fragment IS its tile layout
            if (TRAIN && ei.view_tile && tile_valid) {
#pragma unroll
                for (int k = 0; k < S::KV; k++) *(bf16x8*)((char*)ei.view_tile + (tile32 * S::KV + k) * 1024 + lane * 16) = vf[k];
                p.since += S::KV;
            }
        }
    }
    bf16x8 c[S::KC];
    run_stage<SLOT, S::KW, S::KV, S::CT, true, TRAIN, S::WT>(p, a, vf, c, S::KC + 1, ST(9), tile_valid, a, nullptr, nullptr, mcarry);
    // stage 10: c -> rgb; meanwhile the next block's first weight group streams in
    {
        p.gnext = 0;
        const char* slot = p.begin(has_next ? G0 : 0);
        if (TRAIN && tile_valid) {               // trailing stores of stage 9
#pragma unroll
            for (int q = 4; q >= 1; q--) STREAM_STORE(ST(9) + (2 * S::CT - q) * 1024 + lane * 16, c[2 * S::CT - q]);
            p.since += 4;
            if (MK(8)) {
                // (the lane offset is made opaque here: hoisted out of the block loop as a 64-bit per-lane pointer it is
                // the one value the ENC instantiation spills, and a scratch reload waits for vmcnt(0) -- every store in flight)
                unsigned lo = (unsigned)lane * 16u;
                asm volatile("" : "+v"(lo));
                *(uint4*)(MK(8) + lo) = mcarry;
            }
        }
        const f32x16 acc = mma_tile<S::KC, 0>(slot, lane, c, nullptr);
        const float qn = __builtin_nanf("");
        const float o0 = poison ? qn : acc[0], o1 = poison ? qn : acc[1], o2 = poison ? qn : acc[2], o3 = poison ? qn : dens;
        const f32x4 o = {o0, o1, o2, o3};
        bool scattered = false;
        if constexpr (ENC) scattered = (ei.flags & DURF_FWD_RAW_FULL) != 0 && ray_idx != nullptr;
        if (!scattered) {
            if (valid && lane < 32) *(f32x4*)(raw + row * 4) = o;
        } else if (!tail) {
            // de-duplicated batch, raw straight in the full [B*N,4] layout (what durf_expand_raw would make of the compacted
            // rows): a tile's 32 samples are consecutive samples of ONE ray (N % 32 == 0) -- ray and first sample are
            // wave-uniform (scalar unit), the store stays one 512 B run
            const unsigned row0 = __builtin_amdgcn_readfirstlane((unsigned)(tile32 * 32));
            unsigned Nu = (unsigned)N;
            asm volatile("" : "+s"(Nu));         // (a division of its own: sharing the hoisted reciprocal of N costs a spill)
            const unsigned j = row0 / Nu;
            const unsigned b = (unsigned)ray_idx[tile_valid ? j : 0u];
            unsigned l31 = (unsigned)lane;
            asm volatile("" : "+v"(l31));        // (not a hoisted, spilled per-lane pointer: see MK(8) above)
            l31 &= 31u;
            if (valid && lane < 32) *(f32x4*)(raw + ((size_t)(b * Nu + (row0 - j * Nu)) + l31) * 4) = o;
        } else if (tile_valid) {
            // a box-hit ray's ONE evaluation is the background's raw at every sample of that ray: broadcast it
            unsigned l64 = (unsigned)lane;
            asm volatile("" : "+v"(l64));
            const unsigned l31 = l64 & 31u;
            const unsigned ti = __builtin_amdgcn_readfirstlane((unsigned)(tile32 * 32 - nrows_c));
            const int rr = (valid && lane < 32) ? tail_idx[ti + l31] : -1;
#pragma unroll 1
            for (int r = 0; r < 32; r++) {
                const int ray = __builtin_amdgcn_readlane(rr, r);
                if (ray < 0) break;
                const float v0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, o0), r));
                const float v1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, o1), r));
                const float v2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, o2), r));
                const float v3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, o3), r));
                const f32x4 v = {v0, v1, v2, v3};
                float* dst = raw + (size_t)((unsigned)ray * (unsigned)N) * 4;
                for (unsigned n = l64; n < (unsigned)N; n += 64) *(f32x4*)(dst + (size_t)n * 4) = v;
            }
        }
    }
#undef ST
#undef MK
  }
  if constexpr (MIX) {
    // ---- the object items: two groups of four waves, (object, tile pair) items by ticket ----
    // (every weight DMA of the background blocks has been consumed -- the last block prefetches nothing -- and the stores
    // still in flight do not touch the LDS; the barrier makes sure every wave is past its last weight read)
    if (has_block) __builtin_amdgcn_s_setprio(0);
    ms_barrier();
    mix_object_items<TRAIN>(smem, wave, (int)gridDim.x);
  }
}

// ---------------------------------------------------------------------------
extern "C" {

size_t durf_mlp_param_count(int width, int in_dim) { return durf_layer_offset(width, in_dim, 12, 0); }
size_t durf_mlp_layer_offset(int width, int in_dim, int layer, int want_bias) {
    return durf_layer_offset(width, in_dim, layer, want_bias);
}
size_t durf_wpack_fwd_bytes(int width) {
    return (size_t)(width == 256 ? MlpSpec<256>::TOTAL_CHUNKS : MlpSpec<128>::TOTAL_CHUNKS) * 1024;
}
size_t durf_mlp_mask_bytes(size_t rows) { return ((rows + 31) / 32) * 9 * 1024; }
size_t durf_mlp_stash_bytes(int width, size_t rows) {
    const size_t kb = width == 256 ? MlpSpec<256>::STASH_KS_TOTAL : MlpSpec<128>::STASH_KS_TOTAL;
    return ((rows + 31) / 32) * kb * 1024;
}

int durf_pack_weights_all(void* stream, const float* bkgd_params, int in_bkgd, void* bkgd_fwd, void* bkgd_bwd, int K,
                          const float* obj_params, size_t obj_param_stride, int in_obj, void* obj_fwd, void* obj_bwd) {
    DURF_REQUIRE(bkgd_params == nullptr || (bkgd_fwd != nullptr && in_bkgd > 0 && in_bkgd <= DURF_ENC_DIM),
                 "background MLP: forward stream and 1 <= in_dim <= 64");
    DURF_REQUIRE(K == 0 || (obj_params != nullptr && obj_fwd != nullptr && in_obj > 0 && in_obj <= DURF_ENC_DIM),
                 "object MLPs: parameters, forward streams and 1 <= in_dim <= 64");
    if (K < 0 || (bkgd_params == nullptr && K == 0)) return 0;
    PackAll a;
    a.p_bkgd = bkgd_params; a.f_bkgd = (bf16x8*)bkgd_fwd; a.b_bkgd = (bf16x8*)bkgd_bwd; a.in_bkgd = in_bkgd;
    a.p_obj = obj_params; a.f_obj = (bf16x8*)obj_fwd; a.b_obj = (bf16x8*)obj_bwd; a.in_obj = in_obj;
    a.p_stride = obj_param_stride; a.f_stride = durf_wpack_fwd_bytes(128); a.b_stride = durf_wpack_bwd_bytes(128);
    size_t nv = durf_wpack_fwd_bytes(256) / 16 + durf_wpack_bwd_bytes(256) / 16;        // the wider MLP bounds the grid
    if (bkgd_params == nullptr) nv = durf_wpack_fwd_bytes(128) / 16 + durf_wpack_bwd_bytes(128) / 16;
    hipLaunchKernelGGL(k_pack_all, dim3(durf_cdiv(nv, 256), 1 + K), dim3(256), 0, (hipStream_t)stream, a);
    DURF_CHECK_LAUNCH("durf_pack_weights_all");
    return 0;
}

int durf_pack_weights(void* stream, int width, int in_dim, const float* mlp_params, void* wpack_fwd,
                      void* wpack_bwd) {
    return durf::launch_pack(stream, width, in_dim, 1, mlp_params, 0, wpack_fwd, wpack_bwd);
}

int durf_mlp_fwd_enc(void* stream, size_t rows, int N, const float* t_vals, const float* origins_s, const float* dirs_s,
                     const float* radii, const int32_t* hit, int K, int enc_flags, void* enc_tile, const void* view_bf16,
                     const int32_t* ray_idx, const int32_t* count, const void* wpack_fwd, float* raw, void* stash,
                     void* relu_mask, const int32_t* tail_idx, const int32_t* tail_count, void* view_tile) {
    DURF_REQUIRE((tail_idx == nullptr) == (tail_count == nullptr), "tail_idx and tail_count go together");
    DURF_REQUIRE(tail_idx == nullptr || (count != nullptr && N % 32 == 0), "tail rows follow a compacted ray list");
    DURF_REQUIRE((ray_idx == nullptr) == (count == nullptr), "ray_idx and count go together");
    DURF_REQUIRE(view_tile == nullptr || stash != nullptr, "the view-direction tile is a training output");
    DURF_REQUIRE(t_vals && origins_s && dirs_s && radii && enc_tile, "ray data and the encoding tile buffer are required");
    DURF_REQUIRE(K >= 0 && K <= DURF_MAX_OBJ && (K == 0 || hit != nullptr), "0 <= K <= DURF_MAX_OBJ, hit [B,K]");
    DURF_REQUIRE(N > 0 && (count != nullptr ? N % 32 == 0 : rows % 32 == 0), "whole 32-sample tiles");
    DURF_REQUIRE(!(enc_flags & DURF_FWD_RAW_FULL) || rows < ((size_t)1 << 32), "DURF_FWD_RAW_FULL: 32-bit row numbers");
    EncIn ei{};
    ei.t_vals = t_vals; ei.origins_s = origins_s; ei.dirs_s = dirs_s; ei.radii = radii; ei.hit = hit; ei.K = K; ei.flags = enc_flags;
    ei.view_tile = view_tile;
    return durf::launch_mlp_fwd(stream, 256, rows, N, enc_tile, view_bf16, ray_idx, count, wpack_fwd, raw, stash,
                                relu_mask, 1, FwdStrides{}, tail_idx, tail_count, &ei);
}

// durf_mlp_fwd_enc + durf_obj_fwd_batch as ONE launch where that pays (include/durf_hip.h), else as the two launches
int durf_mlp_fwd_enc_obj(void* stream, size_t rows, int N, const float* t_vals, const float* origins_s, const float* dirs_s,
                         const float* radii, const int32_t* hit, int K, int enc_flags, void* enc_tile, const void* view_bf16,
                         const int32_t* ray_idx, const int32_t* count, const void* wpack_fwd, float* raw, void* stash,
                         void* relu_mask, const int32_t* tail_idx, const int32_t* tail_count, void* view_tile,
                         int B, const int32_t* obj_idx, const int32_t* obj_count, const float* barf_w, int obj_flags,
                         const void* obj_wpack_fwd, void* obj_enc, float* obj_raw, void* obj_stash, void* obj_relu_mask,
                         void* obj_view_tile) {
    DURF_REQUIRE(K > 0 && B > 0 && (size_t)B * N == rows, "K object MLPs over rows = B * N sample rows");
    DURF_REQUIRE(obj_idx && obj_count && barf_w && obj_wpack_fwd && obj_enc && obj_raw, "the object launch's buffers");
    DURF_REQUIRE((stash == nullptr) == (obj_stash == nullptr), "training or inference: both MLP classes alike");
    const bool mix = durf::obj_mix(rows) && stash != nullptr && obj_relu_mask != nullptr && N % 32 == 0 && ray_idx != nullptr;
    if (!mix) {
        int rc = durf_mlp_fwd_enc(stream, rows, N, t_vals, origins_s, dirs_s, radii, hit, K, enc_flags, enc_tile, view_bf16, ray_idx,
                                  count, wpack_fwd, raw, stash, relu_mask, tail_idx, tail_count, view_tile);
        if (rc) return rc;
        return durf_obj_fwd_batch(stream, K, B, N, obj_idx, obj_count, t_vals, origins_s, dirs_s, radii, barf_w, obj_flags, view_bf16,
                                  obj_wpack_fwd, obj_enc, obj_raw, obj_stash, obj_relu_mask, obj_view_tile);
    }
    DURF_REQUIRE((tail_idx == nullptr) == (tail_count == nullptr), "tail_idx and tail_count go together");
    DURF_REQUIRE(tail_idx == nullptr || count != nullptr, "tail rows follow a compacted ray list");
    DURF_REQUIRE(t_vals && origins_s && dirs_s && radii && enc_tile && hit, "ray data, hit masks and the encoding tile buffer");
    DURF_REQUIRE(K <= DURF_MAX_OBJ, "K <= DURF_MAX_OBJ");
    DURF_REQUIRE(!(enc_flags & DURF_FWD_RAW_FULL) || rows < ((size_t)1 << 32), "DURF_FWD_RAW_FULL: 32-bit row numbers");
    EncIn ei{};
    ei.t_vals = t_vals; ei.origins_s = origins_s; ei.dirs_s = dirs_s; ei.radii = radii; ei.hit = hit; ei.K = K; ei.flags = enc_flags;
    ei.view_tile = view_tile;
    // the object items: durf_obj_fwd_batch's arguments as the M-split kernel takes them
    MsFwd ow{};
    ow.rows = rows; ow.N = N; ow.enc = (const bf16x8*)obj_enc; ow.view = (const bf16x8*)view_bf16; ow.ray_idx = obj_idx;
    ow.count = obj_count; ow.wpack = (const char*)obj_wpack_fwd; ow.raw = obj_raw; ow.stash = (bf16x8*)obj_stash;
    ow.relu_mask = (uint4*)obj_relu_mask; ow.nobj = K;
    ow.bs.enc = durf_obj_enc_stride(B, N); ow.bs.idx = (size_t)B; ow.bs.wpack = durf_wpack_fwd_bytes(DURF_W_OBJ);
    ow.bs.raw = rows * 4 * sizeof(float); ow.bs.stash = durf_mlp_stash_bytes(DURF_W_OBJ, rows); ow.bs.mask = durf_mlp_mask_bytes(rows);
    ow.ei.t_vals = t_vals; ow.ei.origins_s = origins_s; ow.ei.dirs_s = dirs_s; ow.ei.radii = radii;
    ow.ei.flags = obj_flags & (DURF_ENC_NO_INTEGRATION | DURF_ENC_CYLINDER); ow.ei.obj = 1;
    for (int i = 0; i < 10; i++) ow.ei.w[i] = barf_w[i];
    ow.ei.view_tile = obj_view_tile; ow.ei.view_stride = durf_obj_view_stride(B, N);
    ow.ticket = durf::next_ticket();
    DURF_REQUIRE(ow.ticket != nullptr, "no item counter for the mixed launch (device allocation failed)");
    if (const char* pr = getenv("DURF_MIX_PROBE"))        // timing probe (the results are WRONG): the launch without its object items
    {
        if (pr[0] == '1') ow.nobj = 0;
        if (pr[0] == '2') ei.flags |= 1 << 30;               // ... and without its background blocks
    }
    hipStream_t s = (hipStream_t)stream;
    // one workgroup per CU: the background blocks' (capacity: the counts live on the device), then room for the object items
    const unsigned nblk = durf_cdiv(rows, 256), nobj = durf_cdiv((size_t)K * durf_cdiv(rows, 64), 2);
    const unsigned g = nblk + nobj < 256u ? nblk + nobj : 256u;
    constexpr int lds = 2 * 4 * (MlpSpec<256>::KW + 1) * 1024;
    static_assert(2 * ms::LDS_BYTES + 96 <= lds, "two object groups fit the background block's LDS");
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)k_mlp_fwd<256, true, 8, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_set = true;
    }
    hipLaunchKernelGGL((k_mlp_fwd<256, true, 8, true, true>), dim3(g), dim3(512), lds, s, rows, N, (const bf16x8*)enc_tile,
                       (const bf16x8*)view_bf16, ray_idx, count, (const char*)wpack_fwd, raw, (bf16x8*)stash, (uint4*)relu_mask,
                       FwdStrides{}, tail_idx, tail_count, ei, ow);
    DURF_CHECK_LAUNCH("durf_mlp_fwd_enc_obj");
    durf::note_dispatch(DURF_DISPATCH_FWD256_8W | DURF_DISPATCH_FWD_ENC | DURF_DISPATCH_FWD_MIX | (tail_idx ? DURF_DISPATCH_FWD_TAIL : 0u) |
                        ((enc_flags & DURF_FWD_RAW_FULL) ? DURF_DISPATCH_FWD_RAW_FULL : 0u));
    return 0;
}

int durf_mlp_fwd(void* stream, int width, size_t rows, int N, const void* enc_tile,
                 const void* view_bf16, const int32_t* ray_idx, const int32_t* count,
                 const void* wpack_fwd, float* raw, void* stash, void* relu_mask,
                 const int32_t* tail_idx, const int32_t* tail_count) {
    DURF_REQUIRE((tail_idx == nullptr) == (tail_count == nullptr), "tail_idx and tail_count go together");
    DURF_REQUIRE(tail_idx == nullptr || (count != nullptr && N % 32 == 0), "tail rows follow a compacted ray list");
    return durf::launch_mlp_fwd(stream, width, rows, N, enc_tile, view_bf16, ray_idx, count, wpack_fwd, raw, stash,
                                relu_mask, 1, FwdStrides{}, tail_idx, tail_count);
}

}  // extern "C"

namespace durf {

static bool msplit_enabled() {           // read per call: tests toggle it
    const char* e = getenv("DURF_OBJ_MSPLIT");
    return !(e && e[0] == '0');
}
// whether launch_mlp_fwd / launch_mlp_bwd take the M-split kernels for a W = 128 launch on compacted ray lists of `rows` rows
bool obj_msplit(size_t rows) { return rows < (size_t)2048 * 128 && msplit_enabled(); }
// whether a training step of `rows` sample rows per level issues its bf16 object MLPs as items of the background MLP's launches
// (durf_mlp_fwd_enc_obj / durf_mlp_bwd_obj; DURF_OBJ_MIX=0: A/B switch, bit-identical results)
bool obj_mix(size_t rows) {
    const char* e = getenv("DURF_OBJ_MIX");
    return !(e && e[0] == '0') && obj_msplit(rows);
}

int pack_bwd_launch(void* stream, int width, int in_dim, int K, const float* params, size_t param_stride, void* wpack_bwd);

// fp32 flax params -> bf16 fragment streams for K MLPs laid out `param_stride` floats apart
int launch_pack(void* stream, int width, int in_dim, int K, const float* params, size_t param_stride,
                void* wpack_fwd, void* wpack_bwd) {
    DURF_REQUIRE(width == 256 || width == 128, "width must be 256 or 128");
    DURF_REQUIRE(in_dim > 0 && in_dim <= DURF_ENC_DIM, "in_dim <= 64");
    if (K <= 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    if (wpack_fwd) {
        const size_t ostride = durf_wpack_fwd_bytes(width);
        if (width == 256)
            hipLaunchKernelGGL(k_pack_fwd<256>, dim3(durf_cdiv(MlpSpec<256>::TOTAL_CHUNKS * 64, 256), K), dim3(256), 0, s,
                               in_dim, params, (bf16x8*)wpack_fwd, param_stride, ostride);
        else
            hipLaunchKernelGGL(k_pack_fwd<128>, dim3(durf_cdiv(MlpSpec<128>::TOTAL_CHUNKS * 64, 256), K), dim3(256), 0, s,
                               in_dim, params, (bf16x8*)wpack_fwd, param_stride, ostride);
        DURF_CHECK_LAUNCH("durf_pack_weights (fwd)");
    }
    if (wpack_bwd) return pack_bwd_launch(stream, width, in_dim, K, params, param_stride, wpack_bwd);
    return 0;
}

int launch_mlp_fwd(void* stream, int width, size_t rows, int N, const void* enc_tile, const void* view_bf16,
                   const int32_t* ray_idx, const int32_t* count, const void* wpack_fwd, float* raw, void* stash,
                   void* relu_mask, int K, const FwdStrides& st, const int32_t* tail_idx, const int32_t* tail_count,
                   const EncIn* enc_in) {
    DURF_REQUIRE(width == 256 || width == 128, "width must be 256 or 128");
    DURF_REQUIRE(enc_in == nullptr || (width == 256 && K == 1 && !enc_in->obj) || (width == 128 && ray_idx && count && (enc_in->obj || obj_msplit(rows))),
                 "the self-encoding forward: the background MLP, or the object MLPs on their compacted ray lists");
    DURF_REQUIRE(rows % 32 == 0, "rows must be a multiple of 32");
    DURF_REQUIRE(K == 1 || (ray_idx && count), "batched launches are for compacted object rays");
    if (rows == 0 || K <= 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    // A launch that fills at most half the chip with 256-sample blocks (8 waves) runs as 128-sample blocks (4 waves): twice the
    // workgroups, each with half the dependent work, one per CU as before (cfg1: 128 -> 256 workgroups).
    const bool half = width == 256 && K == 1 && durf_cdiv(rows, 256) <= 128;
    const unsigned per = half ? 128u : 256u;
    const unsigned nblk = durf_cdiv(rows, per);
    // persistent: at most one workgroup per CU and object
    dim3 grid(nblk < 256u ? nblk : 256u, K), block(half ? 256 : 512);
    const EncIn ei = enc_in ? *enc_in : EncIn{};
    // The object MLPs (W = 128 on compacted ray lists): the M-split kernel -- 4 waves x 64 samples, one output tile per wave --
    // whose launch is a few microseconds of latency instead of one 11-stage round of 256-sample blocks.  DURF_OBJ_MSPLIT=0
    // keeps k_mlp_fwd<128> (A/B switch; bit-identical results).
    // Small batches only (below ops.OVERLAP_MIN_ROWS = 2048 x 128 sample rows): there the object launches sit on the critical
    // path; above, they run on a side stream in the shadow of the persistent background kernels, where a launch that spreads
    // over every CU only delays those (measured at cfg3: 4.26 -> 4.42 ms per step).  At most 128 workgroups per object walk the
    // 64-sample pairs: the hit count lives on the device and an early-exit workgroup still costs its dispatch.
    if (width == 128 && ray_idx && count && obj_msplit(rows)) {
        const size_t items = (size_t)K * durf_cdiv(rows, 64);           // capacity; the counts decide (see the kernel)
        dim3 g((unsigned)(items < 256 ? items : 256)), b(256);         // one workgroup per CU at most: one round
        const MsFwd A{rows, N, (const bf16x8*)enc_tile, (const bf16x8*)view_bf16, ray_idx, count, (const char*)wpack_fwd, raw,
                      (bf16x8*)stash, (uint4*)relu_mask, st, ei, K, nullptr};
        if (stash)
            hipLaunchKernelGGL((k_mlp_fwd_ms<true>), g, b, ms::LDS_BYTES, s, A);
        else
            hipLaunchKernelGGL((k_mlp_fwd_ms<false>), g, b, ms::LDS_BYTES, s, A);
        DURF_CHECK_LAUNCH("durf_mlp_fwd (M-split)");
        note_dispatch(DURF_DISPATCH_FWD128_MSPLIT | (ei.obj ? DURF_DISPATCH_FWD_ENC : 0u));
        return 0;
    }
#define LAUNCH_F(WW, TR, NWV, EN)                                                                 \
    {                                                                                             \
        constexpr int lds = 2 * 4 * (MlpSpec<WW>::KW + 1) * 1024;                                 \
        static bool attr_set = false;      /* once per instantiation (the attribute sticks to the function) */ \
        if (!attr_set) {                                                                          \
            (void)hipFuncSetAttribute((const void*)k_mlp_fwd<WW, TR, NWV, EN>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); \
            attr_set = true;                                                                      \
        }                                                                                         \
        hipLaunchKernelGGL((k_mlp_fwd<WW, TR, NWV, EN>), grid, block, lds, s, rows, N, (const bf16x8*)enc_tile, \
                           (const bf16x8*)view_bf16, ray_idx, count, (const char*)wpack_fwd, raw,  \
                           (bf16x8*)stash, (uint4*)relu_mask, st, tail_idx, tail_count, ei, MsFwd{}); \
    }
    if (enc_in && width == 128) { if (stash) LAUNCH_F(128, true, 8, true) else LAUNCH_F(128, false, 8, true) }
    else if (enc_in) {
        if (half) { if (stash) LAUNCH_F(256, true, 4, true) else LAUNCH_F(256, false, 4, true) }
        else { if (stash) LAUNCH_F(256, true, 8, true) else LAUNCH_F(256, false, 8, true) }
    }
    else if (half) { if (stash) LAUNCH_F(256, true, 4, false) else LAUNCH_F(256, false, 4, false) }
    else if (width == 256) { if (stash) LAUNCH_F(256, true, 8, false) else LAUNCH_F(256, false, 8, false) }
    else { if (stash) LAUNCH_F(128, true, 8, false) else LAUNCH_F(128, false, 8, false) }
#undef LAUNCH_F
    DURF_CHECK_LAUNCH("durf_mlp_fwd");
    note_dispatch((width == 128 ? DURF_DISPATCH_FWD128_SAMPLE : (half ? DURF_DISPATCH_FWD256_4W : DURF_DISPATCH_FWD256_8W)) |
                  (enc_in ? DURF_DISPATCH_FWD_ENC : 0u) | (tail_idx ? DURF_DISPATCH_FWD_TAIL : 0u) |
                  ((enc_in && (enc_in->flags & DURF_FWD_RAW_FULL) && ray_idx) ? DURF_DISPATCH_FWD_RAW_FULL : 0u));
    return 0;
}

}  // namespace durf
