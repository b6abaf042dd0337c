// Which property of a launch makes the ~6 us idle gaps before / after the persistent MLP kernels in the step timeline
// (profiles/r04_step_timeline_512rays.txt)?  Sequences [small, BIG variant, small] x 20 on one stream, run under
//   rocprofv3 --kernel-trace -d out -- ./probe_gaps ; python tools/probes/probe_gaps_report.py out
// BIG variants: dynamic LDS 0 / 136 KB, 512- or 256-thread workgroups, no stores / 256 MB of plain stores / nt stores,
// private scratch.  hipcc --offload-arch=gfx950 -O3 tools/probes/probe_gaps.hip -o probe_gaps
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void __launch_bounds__(256) k_small(float* out) { out[blockIdx.x * 256 + threadIdx.x] = 1.0f; }

__device__ __forceinline__ float spin(float x, int iters) {
    for (int i = 0; i < iters; i++) x = __builtin_fmaf(x, 1.0000001f, 1e-9f);
    return x;
}

// MODE 0: no stores; 1: plain stores; 2: nt stores; 3: scratch
template <int MODE, int THREADS>
__global__ void __launch_bounds__(THREADS) k_big(float* out, size_t floats_per_block, int iters, int sel) {
    extern __shared__ char smem[];
    float x = spin((float)threadIdx.x, iters);
    if (MODE == 3) {
        float priv[64];
        for (int i = 0; i < 64; i++) priv[i] = x + i;
        for (int i = 0; i < 64; i++) priv[(i * sel + 7) & 63] += priv[(i + sel) & 63];
        x = priv[sel & 63];
    }
    if (threadIdx.x == 0 && sel == 12345) smem[0] = 1;
    typedef float f4 __attribute__((ext_vector_type(4)));
    f4* o = (f4*)(out + (size_t)blockIdx.x * floats_per_block);
    if (MODE == 1 || MODE == 2) {
        for (size_t i = threadIdx.x; i < floats_per_block / 4; i += THREADS) {
            f4 v = {x, x, x, x};
            if (MODE == 2) __builtin_nontemporal_store(v, o + i); else o[i] = v;
        }
    } else if (x == 123.456f) o[0] = f4{x, x, x, x};
}

template <int MODE, int THREADS>
static void seq(const char* what, float* small_out, float* big_out, int lds, size_t fpb, int reps) {
    if (lds > 65536) CK(hipFuncSetAttribute((const void*)k_big<MODE, THREADS>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 2; w++) {
        CK(hipEventRecord(e0, 0));
        for (int r = 0; r < reps; r++) {
            hipLaunchKernelGGL(k_small, dim3(64), dim3(256), 0, 0, small_out);
            hipLaunchKernelGGL((k_big<MODE, THREADS>), dim3(256), dim3(THREADS), lds, 0, big_out, fpb, 20000, 1);
            hipLaunchKernelGGL(k_small, dim3(64), dim3(256), 0, 0, small_out);
        }
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
    }
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-50s %.2f us per [small, big, small]\n", what, ms * 1000.0f / reps);
}

int main() {
    float *s, *b;
    CK(hipMalloc(&s, 1 << 20));
    CK(hipMalloc(&b, (size_t)256 << 20));
    const size_t MB1 = (1 << 20) / 4;
    seq<0, 512>("lds 0, 512 threads, no stores", s, b, 0, MB1, 20);
    seq<0, 256>("lds 0, 256 threads, no stores", s, b, 0, MB1, 20);
    seq<0, 512>("lds 136 KB, 512 threads, no stores", s, b, 136 * 1024, MB1, 20);
    seq<1, 512>("lds 0, 512 threads, 256 MB plain stores", s, b, 0, MB1, 20);
    seq<2, 512>("lds 0, 512 threads, 256 MB nt stores", s, b, 0, MB1, 20);
    seq<1, 512>("lds 0, 512 threads, 16 MB plain stores", s, b, 0, MB1 / 16, 20);
    seq<3, 512>("lds 0, 512 threads, scratch", s, b, 0, MB1, 20);
    seq<2, 512>("lds 136 KB, 512 threads, 256 MB nt stores", s, b, 136 * 1024, MB1, 20);
    CK(hipDeviceSynchronize());
    return 0;
}
