// HBM streaming-read calibration for the weight-gradient kernel (k_dw_all): how fast can one
// MI355X pull a few GB through (a) plain 16-byte global loads, (b) an LDS-DMA ring shaped like
// the dW operand ring (8 waves, NC 1-KB chunks per stage, S stages, one barrier per stage).
//   hipcc --offload-arch=gfx950 -O3 probe_hbm.hip -o probe_hbm && ./probe_hbm
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void __launch_bounds__(256) k_plain(const u32x4* __restrict__ p, size_t n_vec, unsigned* sink) {
    // each workgroup streams a contiguous share; 8 loads in flight per lane
    const size_t per = n_vec / gridDim.x;
    const u32x4* q = p + per * blockIdx.x;
    unsigned acc = 0;
    for (size_t i = threadIdx.x; i + 7 * 256 < per; i += 8 * 256) {
        u32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = __builtin_nontemporal_load(q + i + u * 256);
#pragma unroll
        for (int u = 0; u < 8; u++) acc += v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    if (acc == 0x12345678u) *sink = acc;
}

// streaming stores: every workgroup writes a contiguous share, 1 KB per wave instruction
template <int MODE>
__global__ void __launch_bounds__(256) k_store(u32x4* __restrict__ p, size_t n_vec) {
    const size_t per = n_vec / gridDim.x;
    u32x4* q = p + per * blockIdx.x;
    const u32x4 v = {threadIdx.x, blockIdx.x, 3u, 4u};
    for (size_t i = threadIdx.x; i < per; i += 256) {
        if (MODE == 0) q[i] = v;
        else if (MODE == 1) __builtin_nontemporal_store(v, q + i);
        else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(q + i), "v"(v) : "memory");
    }
}
// interleaved like the fused forward: each wave owns 32-sample blocks and writes 16 KB runs
template <int MODE>
__global__ void __launch_bounds__(512) k_store_waves(char* __restrict__ p, size_t bytes) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const size_t nblk = bytes / (8 * 16384);
    const u32x4 v = {threadIdx.x, blockIdx.x, 3u, 4u};
    for (size_t b = blockIdx.x; b < nblk; b += gridDim.x) {
        char* base = p + (b * 8 + wave) * 16384 + lane * 16;
#pragma unroll
        for (int k = 0; k < 16; k++) {
            if (MODE == 0) *(u32x4*)(base + k * 1024) = v;
            else __builtin_nontemporal_store(v, (u32x4*)(base + k * 1024));
        }
    }
}

template <bool NT>
__device__ __forceinline__ void lds_dma16(i32x4 rsrc, unsigned soff, unsigned voff, unsigned lds_addr) {
    unsigned keep;
    if (NT)
        asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\t"
                     "buffer_load_dwordx4 %2, %3, %4 offen nt lds\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
    else
        asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\t"
                     "buffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}
__device__ __forceinline__ i32x4 make_rsrc(const void* p) {
    const unsigned long long a = (unsigned long long)p;
    i32x4 r;
    r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
    r[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32) & 0xffff);
    r[2] = 0x7fffffff;
    r[3] = 0x00020000;
    return r;
}

// NC chunks (1 KB) per stage, S stages, NSTREAM interleaved source streams (like dz + stash).
template <int NC, int S, int NSTREAM, bool NT>
__global__ void __launch_bounds__(512) k_ring(const char* __restrict__ p, size_t bytes, unsigned* sink) {
    extern __shared__ char smem[];
    constexpr int CPW = NC / 8;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const size_t stream_bytes = bytes / NSTREAM;
    const size_t tiles = stream_bytes / ((size_t)(NC / NSTREAM) * 1024);      // per stream
    const size_t tps = tiles / gridDim.x;
    const size_t t0 = tps * blockIdx.x;
    const int nt = (int)tps;
    i32x4 rs[NSTREAM];
#pragma unroll
    for (int s = 0; s < NSTREAM; s++) rs[s] = make_rsrc(p + s * stream_bytes + t0 * (NC / NSTREAM) * 1024);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const unsigned voff = lane * 16;
    auto stage_load = [&](int ti, int slot) {
        ti = __builtin_amdgcn_readfirstlane(ti);
#pragma unroll
        for (int i = 0; i < CPW; i++) {
            const int ci = wave + 8 * i;
            const int s = ci % NSTREAM, k = ci / NSTREAM;
            lds_dma16<NT>(rs[s], (unsigned)((ti * (NC / NSTREAM) + k) * 1024), voff, lds0 + slot * NC * 1024 + ci * 1024);
        }
    };
    for (int i = 0; i < S - 1; i++) if (i < nt) stage_load(i, i);
    unsigned acc = 0;
    for (int t = 0; t < nt; t++) {
        const int later = nt - 1 - t;
        if (later >= S - 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((S - 2) * CPW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (t + S - 1 < nt) stage_load(t + S - 1, (t + S - 1) % S);
        acc += *(volatile unsigned*)(smem + (t % S) * NC * 1024 + threadIdx.x * 4);
    }
    if (acc == 0x12345678u) *sink = acc;
}

template <typename F>
static double time_ms(F f, int reps = 5) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; i++) f();
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

template <int NC, int S, int NSTREAM, bool NT>
static void run_ring(const char* d, size_t bytes, unsigned* sink, int wgs) {
    const int lds = NC * S * 1024;
    CK(hipFuncSetAttribute((const void*)k_ring<NC, S, NSTREAM, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    double ms = time_ms([&] { hipLaunchKernelGGL((k_ring<NC, S, NSTREAM, NT>), dim3(wgs), dim3(512), lds, 0, d, bytes, sink); });
    CK(hipGetLastError());
    printf("ring NC=%2d S=%d streams=%d nt=%d wgs=%4d (LDS %3d KB): %7.1f us  %.2f TB/s\n", NC, S, NSTREAM, (int)NT, wgs, lds / 1024,
           ms * 1e3, bytes / ms / 1e9);
}

int main() {
    const size_t bytes = (size_t)5 << 30;
    char* d; unsigned* sink;
    CK(hipMalloc(&d, bytes)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(d, 1, bytes));
    for (int wgs : {256, 512, 1024, 2048, 4096}) {
        double ms = time_ms([&] { hipLaunchKernelGGL(k_plain, dim3(wgs), dim3(256), 0, 0, (const u32x4*)d, bytes / 16, sink); });
        printf("plain 16B loads, nt, wgs=%4d: %7.1f us  %.2f TB/s\n", wgs, ms * 1e3, bytes / ms / 1e9);
    }
    for (int wgs : {256, 1024, 4096}) {
        double ms = time_ms([&] { hipLaunchKernelGGL(k_store<0>, dim3(wgs), dim3(256), 0, 0, (u32x4*)d, bytes / 16); });
        printf("plain stores          wgs=%4d: %7.1f us  %.2f TB/s\n", wgs, ms * 1e3, bytes / ms / 1e9);
        ms = time_ms([&] { hipLaunchKernelGGL(k_store<1>, dim3(wgs), dim3(256), 0, 0, (u32x4*)d, bytes / 16); });
        printf("nt stores             wgs=%4d: %7.1f us  %.2f TB/s\n", wgs, ms * 1e3, bytes / ms / 1e9);
        ms = time_ms([&] { hipLaunchKernelGGL(k_store<2>, dim3(wgs), dim3(256), 0, 0, (u32x4*)d, bytes / 16); });
        printf("sc0 sc1 stores        wgs=%4d: %7.1f us  %.2f TB/s\n", wgs, ms * 1e3, bytes / ms / 1e9);
    }
    for (int wgs : {256, 512}) {
        double ms = time_ms([&] { hipLaunchKernelGGL(k_store_waves<0>, dim3(wgs), dim3(512), 0, 0, d, bytes); });
        printf("wave-run stores plain wgs=%4d: %7.1f us  %.2f TB/s\n", wgs, ms * 1e3, bytes / ms / 1e9);
        ms = time_ms([&] { hipLaunchKernelGGL(k_store_waves<1>, dim3(wgs), dim3(512), 0, 0, d, bytes); });
        printf("wave-run stores nt    wgs=%4d: %7.1f us  %.2f TB/s\n", wgs, ms * 1e3, bytes / ms / 1e9);
    }
    for (int wgs : {256, 1024}) {
        run_ring<32, 4, 1, false>(d, bytes, sink, wgs);
        run_ring<32, 4, 2, false>(d, bytes, sink, wgs);
        run_ring<32, 4, 2, true>(d, bytes, sink, wgs);
        run_ring<16, 4, 2, false>(d, bytes, sink, wgs);
        run_ring<16, 8, 2, false>(d, bytes, sink, wgs);
        run_ring<16, 8, 2, true>(d, bytes, sink, wgs);
        run_ring<32, 3, 2, false>(d, bytes, sink, wgs);
        run_ring<16, 4, 1, true>(d, bytes, sink, wgs);
    }
    run_ring<16, 4, 2, false>(d, bytes, sink, 512);
    run_ring<16, 4, 2, true>(d, bytes, sink, 512);
    return 0;
}
