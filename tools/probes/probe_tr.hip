// Probe: semantics of ds_read_b64_tr_b16 on gfx950 (used by the weight-gradient GEMM).
// Prints the raw mapping for lane-linear addresses and checks the hypothesis
//   result(lane L, elem j) = chunk[ lane 16*(L/16) + 4*j + ((L%16)>>2) ][ L & 3 ]
// where chunk[l] = the 4 x b16 at lane l's address.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));

__global__ void k(const int* addr_bytes, uint16_t* out) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
    __syncthreads();
    const int a = addr_bytes[threadIdx.x];
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)((char*)lds + a));
    for (int j = 0; j < 4; j++) out[threadIdx.x * 4 + j] = (uint16_t)v[j];
}

int main() {
    int h_addr[64]; uint16_t h_out[256];
    int* d_addr; uint16_t* d_out;
    hipMalloc(&d_addr, sizeof(h_addr)); hipMalloc(&d_out, sizeof(h_out));
    for (int pat = 0; pat < 2; pat++) {
        for (int l = 0; l < 64; l++) h_addr[l] = pat == 0 ? l * 8 : ((l * 37 + 11) % 400) * 8;
        hipMemcpy(d_addr, h_addr, sizeof(h_addr), hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d_addr, d_out);
        hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
        int bad = 0;
        for (int L = 0; L < 64; L++)
            for (int j = 0; j < 4; j++) {
                const int src = 16 * (L / 16) + 4 * j + ((L % 16) >> 2);
                const int want = h_addr[src] / 2 + (L & 3);
                if (h_out[L * 4 + j] != want) bad++;
            }
        printf("pattern %d: hypothesis mismatches = %d\n", pat, bad);
        if (pat == 0 || bad)
            for (int L = 0; L < 64; L++)
                printf("lane %2d: %4d %4d %4d %4d\n", L, h_out[L * 4], h_out[L * 4 + 1], h_out[L * 4 + 2], h_out[L * 4 + 3]);
    }
    return 0;
}
