// Shared helpers for the gfx950 kernels of libdurf_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>
#include <cstdlib>
#include <stdio.h>
#include "../../include/durf_hip.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define DURF_WAVE 64

void durf_set_error(const char* fmt, ...);
namespace durf {
void note_dispatch(unsigned bits);       // durf_dispatch_seen() (include/durf_hip.h, csrc/api.hip)
int* next_ticket();                      // a zeroed device int for one mixed launch (csrc/api.hip); nullptr: none to be had
}

#define DURF_CHECK_LAUNCH(name)                                              \
    do {                                                                     \
        hipError_t e__ = hipGetLastError();                                  \
        if (e__ != hipSuccess) {                                             \
            durf_set_error("%s: %s", name, hipGetErrorString(e__));          \
            return (int)e__;                                                 \
        }                                                                    \
    } while (0)

#define DURF_REQUIRE(cond, msg)                                              \
    do {                                                                     \
        if (!(cond)) {                                                       \
            durf_set_error("%s: requirement failed: %s", __func__, msg);     \
            return -1;                                                       \
        }                                                                    \
    } while (0)

static inline unsigned durf_cdiv(size_t a, size_t b) { return (unsigned)((a + b - 1) / b); }

// ---- device helpers -------------------------------------------------------
__device__ __forceinline__ float nan_min(float a, float b) {   // jnp.minimum propagates NaN
    return (a != a || b != b) ? __builtin_nanf("") : fminf(a, b);
}
__device__ __forceinline__ float nan_max(float a, float b) {
    return (a != a || b != b) ? __builtin_nanf("") : fmaxf(a, b);
}
// jnp.nan_to_num(x): nan -> 0, +-inf -> +-FLT_MAX
__device__ __forceinline__ float nan_to_num(float x) {
    if (x != x) return 0.0f;
    if (x == __builtin_inff()) return 3.4028234663852886e+38f;
    if (x == -__builtin_inff()) return -3.4028234663852886e+38f;
    return x;
}
// math.safe_sin (internal/math.py:35-46): sin(|x| < 100*pi ? x : x mod 100*pi), the mod
// being jnp.remainder (exact fmod, then shifted to the sign of the divisor).
__device__ __forceinline__ float safe_sin(float x) {
    const float t = 314.15927124023438f;   // float32(100*pi)
    if (!(fabsf(x) < t)) {
        float m = fmodf(x, t);
        if (m != 0.0f && m < 0.0f) m += t;
        x = m;
    }
    return sinf(x);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// inclusive prefix sum across the 64 lanes of a wave
__device__ __forceinline__ float wave_incl_scan(float v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        float t = __shfl_up(v, o, 64);
        if (lane >= o) v += t;
    }
    return v;
}
// inclusive suffix sum across the 64 lanes of a wave
__device__ __forceinline__ float wave_incl_rscan(float v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        float t = __shfl_down(v, o, 64);
        if (lane + o < 64) v += t;
    }
    return v;
}

// byte offset of the 16-byte vector holding features [8q, 8q+8) of `row` in the bf16 tile
// layout of a [rows, 16*nks] matrix (include/durf_hip.h): q = feature/8 in [0, 2*nks)
__device__ __forceinline__ size_t tile_vec_offset(size_t row, int q, int nks) {
    return ((((row >> 5) * nks + (q >> 1)) * 64) + (size_t)((q & 1) * 32) + (row & 31)) * 16;
}

// ---- LDS-DMA with explicit waits --------------------------------------------------------
// hipcc orders every ds_read (and every __syncthreads) behind ALL outstanding buffer_load...lds
// it knows about with s_waitcnt vmcnt(0) -- and on gfx9 stores count on vmcnt too, so that wait
// also drains every store the wave has in flight.  Issued from inline asm the DMA is invisible
// to the compiler and ordered only by the kernel's own counted waits (memory operations of a
// wave retire in issue order: vmcnt <= K means everything but the newest K has completed).
// M0 = LDS byte address of the 1 KB chunk; saved/restored because hipcc owns M0.
typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ i32x4 make_rsrc(const void* p) {
    const unsigned long long a = (unsigned long long)p;
    i32x4 r;
    r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
    r[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32) & 0xffff);
    r[2] = 0x7fffffff;          // num_records (bytes): whole address space above the base
    r[3] = 0x00020000;          // raw buffer, dword data format (as __builtin_amdgcn_make_buffer_rsrc)
    return r;
}
__device__ __forceinline__ void lds_dma16_cached(i32x4 rsrc, unsigned soff, unsigned voff, unsigned lds_addr) {
    unsigned keep;
    asm volatile("s_nop 4\n\t"
                 "s_mov_b32 %0, m0\n\t"
                 "s_mov_b32 m0, %1\n\t"
                 "s_nop 0\n\t"
                 "buffer_load_dwordx4 %2, %3, %4 offen lds\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff)
                 : "memory");
}
// wait until at most min(n, 12) of this wave's vector-memory operations are outstanding (n wave-uniform)
__device__ __forceinline__ void wait_vmcnt_le(int n) {
    switch (n) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
        case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
        case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
    }
}

#ifndef MIX_ROLE_ROT
#define MIX_ROLE_ROT 2
#endif
#ifndef MS_ENC_SPLIT
#define MS_ENC_SPLIT 1          // the M-split object forward's encoding: two feature vectors per wave (0: all eight on wave 0, round 5)
#endif
#ifndef MS_RING
#define MS_RING 4               // B-fragment reads in flight per wave in a stage of the M-split object forward (0: hipcc's schedule)
#endif
#ifndef MIX_TICKET_WAVE
#define MIX_TICKET_WAVE 1        // the wave that requests the mixed launches' tickets (A/B: tools/experiments/r06_mix_ab.sh)
#endif
// A by-value kernel argument read from the kernarg segment where it is needed -- dword by dword through a constant-address-
// space pointer, i.e. scalar loads -- instead of at kernel entry (the mixed launches: their object arguments must not live in
// scalar registers across the background loop)
template <class T>
__device__ __forceinline__ void load_kernarg(T& dst, const __attribute__((address_space(4))) char* src) {
    static_assert(sizeof(T) % 4 == 0, "whole dwords");
    unsigned* d = (unsigned*)&dst;
    const __attribute__((address_space(4))) unsigned* s = (const __attribute__((address_space(4))) unsigned*)src;
#pragma unroll
    for (int i = 0; i < (int)(sizeof(T) / 4); i++) d[i] = s[i];
}

// A device-memory pointer the compiler cannot trace to a kernel argument (read from the kernarg segment by hand, or a field of
// a struct passed to a real call) is GENERIC to it, and flat loads / stores count on lgkmcnt as well as vmcnt: every ms_barrier
// would wait for the stores in flight.  Told that it is neither an LDS nor a scratch address, the compiler emits global accesses.
template <class T>
__device__ __forceinline__ T* as_global(T* p) {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_assume(!__builtin_amdgcn_is_shared((const void*)p) && !__builtin_amdgcn_is_private((const void*)p));
#endif
    return p;
}

// Workgroup barrier of the M-split kernels: this wave's LDS writes have landed (lgkmcnt), then the raw barrier.  __syncthreads()
// also waits for vmcnt(0) -- every global store and every prefetched weight load the wave has in flight -- which made each of
// the 11-13 stages of a tile pay a full HBM round trip (k_mlp_fwd_ms 46 -> 2x us at K = 8).  Nothing the other waves read
// through LDS depends on global memory traffic.
__device__ __forceinline__ void ms_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
}

// 16-byte store of a streamed-once tensor (activation stash, dz): non-temporal, so the stream does
// not compete with the packed weights for L2 (measured on the fused forward / backward: nt 784 / 643 us,
// plain 826 / 700, sc1 816 / 747, sc0 sc1 807 / 745)
// (an explicitly GLOBAL pointer: a generic one the compiler cannot trace to a kernel argument would make it a flat store,
// which counts on lgkmcnt too -- see as_global)
#define DURF_G(T) __attribute__((address_space(1))) T
#define STREAM_STORE(ptr, val) __builtin_nontemporal_store((val), (DURF_G(bf16x8)*)(ptr))

// ---- batched (per-object) launches ---------------------------------------------------------
// The K object MLPs use [K, ...] slabs with uniform strides (include/durf_hip.h, durf_obj_*): the
// kernels take the object index from blockIdx.y (k_dw_finalize: blockIdx.z) and offset their
// pointers by these strides (bytes, except idx in int32 elements and params/grads in floats).
// Strides of 0 with gridDim.y == 1 are the plain per-MLP launches.
struct FwdStrides { size_t enc, idx, wpack, raw, stash, mask; };
// the background encoder's inputs, for the forward that encodes its own tiles (k_mlp_fwd<256, .., ENC>; durf_mlp_fwd_enc)
// (obj: the object form -- no hit masking, no contraction, the coordinates prepended and the BARF weights w, mip.py:182-223;
// view_tile, nullable: the launch also writes the per-sample view-direction tile [rows,32] the weight-gradient GEMM of
// Dense_10 reads (durf_expand_view's output; view_stride bytes apart for batched objects))
struct EncIn {
    const float* t_vals; const float* origins_s; const float* dirs_s; const float* radii; const int32_t* hit; int K; int flags;
    int obj; float w[10]; void* view_tile; size_t view_stride;
};
struct BwdStrides { size_t idx, wpack, mask, dz, dz_out, d_enc; };
struct DwStrides { size_t enc, view, stash, dz_out, part, bpart; };      // stash stride also applies to dz

namespace durf {
int launch_pack(void* stream, int width, int in_dim, int K, const float* params, size_t param_stride,
                void* wpack_fwd, void* wpack_bwd);
int launch_encode_obj(void* stream, int K, int max_rays, int N, const int32_t* idx, const int32_t* count,
                      const float* t_vals, const float* origins_s, const float* dirs_s, const float* radii,
                      const float* barf_w, int flags, void* out_tile, size_t out_stride, float* out_f32);
bool obj_msplit(size_t rows);
bool obj_mix(size_t rows);
int launch_mlp_fwd(void* stream, int width, size_t rows, int N, const void* enc_tile, const void* view_bf16,
                   const int32_t* ray_idx, const int32_t* count, const void* wpack_fwd, float* raw, void* stash,
                   void* relu_mask, int K, const FwdStrides& st, const int32_t* tail_idx = nullptr,
                   const int32_t* tail_count = nullptr, const EncIn* enc_in = nullptr);
int launch_mlp_bwd(void* stream, int width, size_t rows, int N, const float* draw, const int32_t* ray_idx,
                   const int32_t* count, const void* wpack_bwd, const void* relu_mask, void* dz, void* dz_out,
                   float* d_enc, int K, const BwdStrides& st, const int32_t* tail_idx = nullptr,
                   const int32_t* tail_count = nullptr, const float* draw_ray_sum = nullptr);
int launch_mlp_bwd_ms_levels(void* stream, size_t rows, int N, int nlevels, const float* const* draw, const int32_t* ray_idx,
                             const int32_t* count, const void* wpack_bwd, const void* const* relu_mask, void* const* dz,
                             void* const* dz_out, int K, const BwdStrides& st);
int launch_expand_view(void* stream, size_t rows, int N, const void* view_bf16, const int32_t* ray_idx,
                       const int32_t* count, void* out_tile, int K, size_t idx_stride, size_t out_stride,
                       const int32_t* tail_idx = nullptr, const int32_t* tail_count = nullptr);
// Sample axis of the weight-gradient GEMMs: `nlevels` segments, each with its own row capacity (a multiple of 32,
// the layout stride of that level's buffers), rows per ray and (nullable) device-side ray count -- the sampling levels
// of a step plus, for a de-duplicated batch, the one-sample-per-ray evaluations of the box-hit rays.
struct DwLevels {
    size_t rows[DURF_MAX_LEVELS];
    int n[DURF_MAX_LEVELS];
    const int32_t* count[DURF_MAX_LEVELS];
    int nlevels;
};
int launch_mlp_dw(void* stream, int width, const DwLevels& lv,
                  const void* const* enc_tile, const void* const* view_tile, const void* const* stash,
                  const void* const* dz, const void* const* dz_out, float* part, float* bpart, int K,
                  const DwStrides& st);
int launch_dw_finalize(void* stream, int width, int in_dim, const DwLevels& lv,
                       const float* part, const float* bpart, float* grad_mlp, int K, size_t part_stride,
                       size_t bpart_stride, size_t grad_stride, const float* mlp_params, size_t param_stride);
DwLevels uniform_levels(size_t rows, int N, const int32_t* count, int nlevels);
struct DwFinSpec {           // one class of MLPs of a finalize launch (launch_dw_finalize2)
    int width, in_dim, K;
    DwLevels lv;
    const float* part; const float* bpart; float* grad; const float* params;
    size_t part_stride, bpart_stride, grad_stride, param_stride;
};
int launch_dw_finalize2(void* stream, const DwFinSpec& a, const DwFinSpec& b);
}  // namespace durf
