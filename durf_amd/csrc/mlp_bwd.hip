// Fused MLP backward (data path) + weight-gradient GEMM.  (under construction)
#include "mlp_spec.h"

extern "C" {
size_t durf_wpack_bwd_bytes(int width) { (void)width; return 1024; }
int durf_pack_weights_bwd(void* stream, int width, int in_dim, const float* mlp_params, void* wpack_bwd) {
    (void)stream; (void)width; (void)in_dim; (void)mlp_params; (void)wpack_bwd;
    durf_set_error("durf_pack_weights_bwd: not built yet");
    return -2;
}
}
