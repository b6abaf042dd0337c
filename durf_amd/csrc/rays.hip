// Ray-side kernels: box setup (K1), hit compaction, stratified sampling (K2),
// view-direction encoding (K5), conical-frustum -> Gaussian -> contraction -> IPE (K3/K4).
// All arithmetic is fp32 and follows the op order of the reference lines cited in
// include/durf_hip.h so that results agree with an fp32 evaluation of the reference to
// rounding.  These kernels are HBM-write-bound (31 KB written per ray-level for the
// background encoding); lanes are arranged so every store instruction writes full lines.
#include "durf_common.h"
#include "mlp_pack.h"          // PackAll / pack_all_vec: the weight packing can ride in the step prologue

// ---------------------------------------------------------------------------
// K1: ray_setup.  One thread per ray, K-loop; rotation matrices built once per block.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void ray_setup_block(int blk, int B, int K, const float* __restrict__ origins,
                                                const float* __restrict__ dirs, const float* __restrict__ pose,
                                                const float* __restrict__ ext, float* __restrict__ origins_s,
                                                float* __restrict__ dirs_s, int32_t* __restrict__ hit,
                                                float* __restrict__ zo) {
    __shared__ float sR[DURF_MAX_OBJ][9];
    __shared__ float sT[DURF_MAX_OBJ][3];   // R * (-c)
    __shared__ float sE[DURF_MAX_OBJ][3];
    if (threadIdx.x < K) {
        const int k = threadIdx.x;
        // box_helpers.aa2matrix (:148-167)
        const float rx = pose[k * 6 + 3], ry = pose[k * 6 + 4], rz = pose[k * 6 + 5];
        float s = rx * rx + ry * ry + rz * rz;
        s = (s < 1e-12f) ? 1e-12f : s;                    // math.safe_norm (:27-32)
        const float th = sqrtf(s) + 1e-12f;
        const float a = sinf(th) / th;
        const float b = (1.0f - cosf(th)) / (th * th);
        // skew = [[0,-z,y],[z,0,-x],[-y,x,0]]; skew^2 computed as a matmul
        const float S[9] = {0.f, -rz, ry, rz, 0.f, -rx, -ry, rx, 0.f};
        float R[9];
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 3; j++) {
                float s2 = S[i * 3 + 0] * S[0 * 3 + j] + S[i * 3 + 1] * S[1 * 3 + j] + S[i * 3 + 2] * S[2 * 3 + j];
                R[i * 3 + j] = ((i == j) ? 1.0f : 0.0f) + a * S[i * 3 + j] + b * s2;
            }
        const float cx = -pose[k * 6 + 0], cy = -pose[k * 6 + 1], cz = -pose[k * 6 + 2];
#pragma unroll
        for (int i = 0; i < 9; i++) sR[k][i] = R[i];
#pragma unroll
        for (int i = 0; i < 3; i++) {
            sT[k][i] = R[i * 3 + 0] * cx + R[i * 3 + 1] * cy + R[i * 3 + 2] * cz;   // rotate_matrix(-pose)
            sE[k][i] = ext[k * 3 + i];
        }
    }
    __syncthreads();
    const int b = blk * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const float ox = origins[b * 3 + 0], oy = origins[b * 3 + 1], oz = origins[b * 3 + 2];
    const float dx = dirs[b * 3 + 0], dy = dirs[b * 3 + 1], dz = dirs[b * 3 + 2];
    float so[3] = {0.f, 0.f, 0.f}, sd[3] = {0.f, 0.f, 0.f};
    float zsum = 0.f;
    int nhit = 0;
    for (int k = 0; k < K; k++) {
        const float* R = sR[k];
        float po[3], pd[3];
#pragma unroll
        for (int i = 0; i < 3; i++) {
            po[i] = (R[i * 3 + 0] * ox + R[i * 3 + 1] * oy + R[i * 3 + 2] * oz) + sT[k][i];
            pd[i] = R[i * 3 + 0] * dx + R[i * 3 + 1] * dy + R[i * 3 + 2] * dz;
        }
        const float nrm = sqrtf(pd[0] * pd[0] + pd[1] * pd[1] + pd[2] * pd[2]);   // :340
        float tn = -__builtin_inff(), tf = __builtin_inff();
        bool first = true;
#pragma unroll
        for (int i = 0; i < 3; i++) {
            pd[i] = pd[i] / nrm;
            const float inv = 1.0f / pd[i];                                       // :79
            const float tmin = (-sE[k][i] - po[i]) * inv;
            const float tmax = (sE[k][i] - po[i]) * inv;
            const float t0 = nan_min(tmin, tmax), t1 = nan_max(tmin, tmax);
            tn = first ? t0 : nan_max(tn, t0);
            tf = first ? t1 : nan_min(tf, t1);
            first = false;
        }
        int h = (tf > tn) ? 1 : 0;                                                // :91
        const int pos = (tf * (float)h > 0.0f) ? 1 : 0;                           // :95
        h *= pos;
        const float hf = (float)h;
        hit[b * K + k] = h;
        zsum += hf * (tf * hf);                                                   // :102, model :131
        nhit += h;
#pragma unroll
        for (int i = 0; i < 3; i++) { so[i] += po[i] * hf; sd[i] += pd[i] * hf; } // model :117-118
    }
    const float bk = (nhit == 0) ? 1.0f : 0.0f;                                   // model :115
    origins_s[b * 3 + 0] = so[0] + bk * ox;
    origins_s[b * 3 + 1] = so[1] + bk * oy;
    origins_s[b * 3 + 2] = so[2] + bk * oz;
    dirs_s[b * 3 + 0] = sd[0] + bk * dx;
    dirs_s[b * 3 + 1] = sd[1] + bk * dy;
    dirs_s[b * 3 + 2] = sd[2] + bk * dz;
    zo[b] = zsum;
}

__global__ void __launch_bounds__(256)
k_ray_setup(int B, int K, const float* __restrict__ origins, const float* __restrict__ dirs,
            const float* __restrict__ pose, const float* __restrict__ ext,
            float* __restrict__ origins_s, float* __restrict__ dirs_s,
            int32_t* __restrict__ hit, float* __restrict__ zo) {
    ray_setup_block(blockIdx.x, B, K, origins, dirs, pose, ext, origins_s, dirs_s, hit, zo);
}

// ---------------------------------------------------------------------------
// ordered stream compaction of hit[:,k]; one block per object, wave-ballot scan.
// ---------------------------------------------------------------------------
// CLASSES = false: block k compacts the rays with hit[:,k] != 0.
// CLASSES = true (2 blocks): the two ray classes of the de-duplicated background evaluation (durf_expand_raw):
//   class 0 = rays that hit no box or several (evaluated sample by sample), class 1 = rays that hit exactly one.
//   Block 0 also writes dyn[b] = number of boxes ray b hits, count[2] = count0 * N + count1 (the valid rows of the
//   compacted buffers), count[3] = number of rays that hit several boxes and count[4] = bit k set: box k is hit by such
//   a ray (durf_poison_multi_hit).
template <bool CLASSES>
__device__ __forceinline__ void compact_hits_block(int k, int B, int K, int N, const int32_t* __restrict__ hit,
                                                   int32_t* __restrict__ idx, int32_t* __restrict__ count,
                                                   int32_t* __restrict__ slot, int32_t* __restrict__ dyn) {
    const int KS = CLASSES ? 2 : K;                  // columns of slot
    __shared__ int wave_tot[16];
    __shared__ int base_s, multi_s, bits_s;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) { base_s = 0; multi_s = 0; bits_s = 0; }
    __syncthreads();
    int multi = 0, bits = 0;
    for (int b0 = 0; b0 < B; b0 += 1024) {
        const int b = b0 + threadIdx.x;
        int h = 0;
        if (b < B) {
            if (CLASSES) {
                int nh = 0;
                int hb = 0;
                for (int j = 0; j < K; j++) { const int hj = hit[b * K + j] != 0; nh += hj; hb |= hj << j; }
                h = k == 0 ? (nh != 1) : (nh == 1);
                if (k == 0) { dyn[b] = nh; multi += nh > 1; if (nh > 1) bits |= hb; }
            } else {
                h = hit[b * K + k] != 0;
            }
        }
        const unsigned long long m = __ballot(h);
        const int before = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) wave_tot[wave] = __popcll(m);
        __syncthreads();
        int woff = 0, tot = 0;
        for (int w = 0; w < 16; w++) { const int t = wave_tot[w]; if (w < wave) woff += t; tot += t; }
        const int base = base_s;
        if (b < B) {
            const int pos = base + woff + before;
            slot[b * KS + k] = h ? pos : -1;
            if (h) idx[(size_t)k * B + pos] = b;
        }
        __syncthreads();
        if (threadIdx.x == 0) base_s = base + tot;
        __syncthreads();
    }
    if (CLASSES && k == 0) {
        if (multi) { atomicAdd(&multi_s, multi); atomicOr(&bits_s, bits); }       // integer: order-independent
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        count[k] = base_s;
        if (CLASSES && k == 0) {
            count[2] = base_s * N + (B - base_s);
            count[3] = multi_s;
            count[4] = bits_s;
        }
    }
}

template <bool CLASSES>
__global__ void __launch_bounds__(1024)
k_compact_hits(int B, int K, int N, const int32_t* __restrict__ hit, int32_t* __restrict__ idx,
               int32_t* __restrict__ count, int32_t* __restrict__ slot, int32_t* __restrict__ dyn) {
    compact_hits_block<CLASSES>(blockIdx.x, B, K, N, hit, idx, count, slot, dyn);
}

// both compactions of a step in one launch: blocks [0, K) the per-object ray lists, blocks K, K+1 the two ray classes
__global__ void __launch_bounds__(1024)
k_compact_all(int B, int K, int N, const int32_t* __restrict__ hit, int32_t* __restrict__ idx_obj,
              int32_t* __restrict__ count_obj, int32_t* __restrict__ slot_obj, int32_t* __restrict__ idx_cls,
              int32_t* __restrict__ count_cls, int32_t* __restrict__ slot_cls, int32_t* __restrict__ dyn) {
    if ((int)blockIdx.x < K) compact_hits_block<false>(blockIdx.x, B, K, 0, hit, idx_obj, count_obj, slot_obj, nullptr);
    else compact_hits_block<true>(blockIdx.x - K, B, K, N, hit, idx_cls, count_cls, slot_cls, dyn);
}

// ---------------------------------------------------------------------------
// K2: level-0 t_vals (mip.py:353-368).  linspace(0,1,N+1)[i] == i/N in fp32.
// ---------------------------------------------------------------------------
// Philox4x32-10 (Salmon et al., SC'11: "Parallel random numbers: as easy as 1, 2, 3"), counter (c0, 0, 0, 0), key (k0, k1):
// the stratified-sampling draws of a step made INSIDE its first launch, as the reference makes them inside its program
// (mip.py:364, math.py:257-260: jax.random.uniform on a key) instead of by a generator kernel in front of it.  Word 0 of
// block i jitters level-0 sample position i, words 1 / 2 / 3 are the resampling draw i behind levels 0 / 1 / 2.  Restated on the
// CPU (with the generator's known-answer vectors) in oracle/philox_ref.py; tests/test_gpu_sampling_noise.py.
__device__ __forceinline__ void philox4x32_10_4(unsigned c0, unsigned k0, unsigned k1, unsigned (&x)[4], unsigned c1 = 0u) {
    unsigned c2 = 0u, c3 = 0u;
#pragma unroll
    for (int r = 0; r < 10; r++) {
        const unsigned lo0 = 0xD2511F53u * c0, hi0 = __umulhi(0xD2511F53u, c0);
        const unsigned lo1 = 0xCD9E8D57u * c2, hi1 = __umulhi(0xCD9E8D57u, c2);
        const unsigned n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    x[0] = c0; x[1] = c1; x[2] = c2; x[3] = c3;
}
__device__ __forceinline__ void philox4x32_10(unsigned c0, unsigned k0, unsigned k1, unsigned& x0, unsigned& x1, unsigned c1 = 0u) {
    unsigned x[4];
    philox4x32_10_4(c0, k0, k1, x, c1);
    x0 = x[0]; x1 = x[1];
}
// 24 random bits -> [0, 1) (what jax.random.uniform's fp32 draw resolves, 2^-24 apart here)
__device__ __forceinline__ float u01_24(unsigned x) { return (float)(x >> 8) * 5.9604644775390625e-08f; }

__device__ __forceinline__ void sample_t_block(size_t blk, int B, int N, const float* __restrict__ near,
                                               const float* __restrict__ far, const float* __restrict__ t_rand,
                                               int lindisp, float* __restrict__ t_vals, bool draw = false, unsigned seed_lo = 0u,
                                               unsigned seed_hi = 0u, float* __restrict__ u_rand_out = nullptr) {
    const size_t i = blk * blockDim.x + threadIdx.x;
    const size_t tot = (size_t)B * (N + 1);
    if (i >= tot) return;
    float jitter = 0.0f;
    if (draw) {
        // words 1, 2, 3 of the block: the resampling draws of the resamples behind levels 0, 1, 2 -- one plane [B, N+1] each
        // (round 6: one plane for every level correlated the draws of num_levels > 2; at most DURF_FORWARD_MAX_LEVELS - 1 = 3)
        unsigned x[4];
        philox4x32_10_4((unsigned)i, seed_lo, seed_hi, x);
        jitter = u01_24(x[0]);
        u_rand_out[i] = u01_24(x[1]);
        u_rand_out[tot + i] = u01_24(x[2]);
        u_rand_out[2 * tot + i] = u01_24(x[3]);
    }
    const int b = (int)(i / (N + 1)), n = (int)(i % (N + 1));
    const float nr = near[b], fr = far[b];
    auto tv = [&](int m) {
        const float s = (m == N) ? 1.0f : (float)m / (float)N;
        const float t = nr * (1.0f - s) + fr * s;
        return lindisp ? 1.0f / t : t;                                  // mip.py:354-356
    };
    float t = tv(n);
    if (t_rand || draw) {                                               // :360-365
        const float lower = (n == 0) ? t : 0.5f * (t + tv(n - 1));
        const float upper = (n == N) ? t : 0.5f * (tv(n + 1) + t);
        t = lower + (upper - lower) * (draw ? jitter : t_rand[i]);
    }
    t_vals[i] = t;
}

__global__ void __launch_bounds__(256)
k_sample_t(int B, int N, const float* __restrict__ near, const float* __restrict__ far,
           const float* __restrict__ t_rand, int lindisp, float* __restrict__ t_vals) {
    sample_t_block(blockIdx.x, B, N, near, far, t_rand, lindisp, t_vals);
}

// ---------------------------------------------------------------------------
// K5: view-direction encoding (mip.py:36-45): [v, sin(2^i v_j), sin(2^i v_j + pi/2)]
// ---------------------------------------------------------------------------
__device__ __forceinline__ void view_enc_block(int blk, int B, const float* __restrict__ viewdirs,
                                               __bf16* __restrict__ out_bf16, float* __restrict__ out_f32) {
    const int i = blk * blockDim.x + threadIdx.x;
    if (i >= B * DURF_VIEW_DIM) return;
    const int b = i / DURF_VIEW_DIM, f = i % DURF_VIEW_DIM;
    float val = 0.0f;
    if (f < 3) {
        val = viewdirs[b * 3 + f];
    } else if (f < 27) {
        const int g = f - 3, c = g / 12, r = g % 12, deg = r / 3, j = r % 3;
        float y = viewdirs[b * 3 + j] * (float)(1 << deg);
        if (c) y = y + 1.5707963705062866f;                // float32(0.5*pi)
        val = sinf(y);                                      // plain jnp.sin here (mip.py:41)
    }
    if (out_bf16) out_bf16[i] = (__bf16)val;
    if (out_f32 && f < 27) out_f32[b * 27 + f] = val;
}

__global__ void __launch_bounds__(256)
k_view_enc(int B, const float* __restrict__ viewdirs, __bf16* __restrict__ out_bf16,
           float* __restrict__ out_f32) {
    view_enc_block(blockIdx.x, B, viewdirs, out_bf16, out_f32);
}

// The three per-ray preparations of a step that depend on nothing but the batch -- ray setup (K1), view-direction
// encoding (K5) and the level-0 sample positions (K2) -- as ONE launch: the grid covers the largest of the three index
// spaces (B (N+1) sample positions) and the leading blocks also do the other two.
__global__ void __launch_bounds__(256)
k_ray_prologue(int B, int K, int N, const float* __restrict__ origins, const float* __restrict__ dirs,
               const float* __restrict__ pose, const float* __restrict__ ext, float* __restrict__ origins_s,
               float* __restrict__ dirs_s, int32_t* __restrict__ hit, float* __restrict__ zo,
               const float* __restrict__ viewdirs, __bf16* __restrict__ view_bf16,
               const float* __restrict__ near, const float* __restrict__ far, const float* __restrict__ t_rand,
               int lindisp, float* __restrict__ t_vals, float* __restrict__ pose_copy, float* __restrict__ zero_buf,
               size_t zero_count, unsigned seed_lo, unsigned seed_hi, float* __restrict__ u_rand_out, PackAll pk, int nb_pro,
               int pack_blocks_bkgd, int pack_blocks_obj, float* __restrict__ zero_buf2, size_t zero_count2) {
    // Workgroups behind the nb_pro of the prologue proper pack the step's bf16 weight streams (durf_ray_prologue_pack): the
    // packing depends on the parameters only, so it shares this launch instead of being the next one (12 us of a 0.4-0.7 ms
    // small-batch step).  MLP-major: the background MLP's vectors, then each object's.
    if ((int)blockIdx.x >= nb_pro) {
        const int pb = (int)blockIdx.x - nb_pro;
        if (pb < pack_blocks_bkgd) pack_all_vec(pk, 0, pb * 256 + (int)threadIdx.x);
        else {
            const int rel = pb - pack_blocks_bkgd;
            pack_all_vec(pk, 1 + rel / pack_blocks_obj, (rel % pack_blocks_obj) * 256 + (int)threadIdx.x);
        }
        return;
    }
    // two chores of a training step that cost a launch of their own otherwise (ray-independent; done first so that the
    // stores are in flight under the ray setup): a snapshot of this timestep's poses (the step returns the poses it
    // rendered with, train_boxpose.py:315, and the optimizer updates them in place) and the zero fill of the gradient
    if (pose_copy && blockIdx.x == 0 && (int)threadIdx.x < K * 6) pose_copy[threadIdx.x] = pose[threadIdx.x];
    if (zero_buf) {
        const size_t nthr = (size_t)nb_pro * blockDim.x, gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
        const size_t n4 = zero_count >> 2;                                   // (16-byte aligned buffer: the wrapper checks)
        for (size_t i = gid; i < n4; i += nthr) ((float4*)zero_buf)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (gid < (zero_count & 3)) zero_buf[n4 * 4 + gid] = 0.0f;
    }
    if (zero_buf2) {          // a second, small region (the one-call step: dyn_mask of a model without boxes, or the pose sums)
        const size_t nthr = (size_t)nb_pro * blockDim.x;
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < zero_count2; i += nthr) zero_buf2[i] = 0.0f;
    }
    if ((int)blockIdx.x < (B + 255) / 256)                            // block-uniform: ray_setup_block has a barrier
        ray_setup_block(blockIdx.x, B, K, origins, dirs, pose, ext, origins_s, dirs_s, hit, zo);
    if ((int)blockIdx.x < (B * DURF_VIEW_DIM + 255) / 256) view_enc_block(blockIdx.x, B, viewdirs, view_bf16, nullptr);
    sample_t_block(blockIdx.x, B, N, near, far, t_rand, lindisp, t_vals, u_rand_out != nullptr, seed_lo, seed_hi, u_rand_out);
}

// ---------------------------------------------------------------------------
// K3/K4: per-sample Gaussian + encoding.  8 lanes per sample, each lane produces the 8
// consecutive features of one 16-byte output vector, so a wave writes 8 samples x 128 B.
// ---------------------------------------------------------------------------
#include "enc_lane.h"      // gauss.h + the per-lane encoder shared with the fused forward (mlp_fwd.hip)


template <bool OBJ>
__global__ void __launch_bounds__(256)
k_encode(int rays, int N, const int32_t* __restrict__ idx, const int32_t* __restrict__ count,
         const float* __restrict__ t_vals, const float* __restrict__ origins_s,
         const float* __restrict__ dirs_s, const float* __restrict__ radii,
         const int32_t* __restrict__ hit, int K, int contraction, BarfW barf_w,
         bf16x8* __restrict__ out_tile, float* __restrict__ out_f32, size_t idx_stride, size_t f32_stride) {
    if (OBJ && gridDim.y > 1) {                      // batched objects (fp32 features only): blockIdx.y = object
        idx += blockIdx.y * idx_stride;
        count += blockIdx.y;
        out_f32 += blockIdx.y * f32_stride;
    }
    // grid-stride over (sample, 8-feature vector) pairs: the object launches cap their grid (the hit count lives on the
    // device; a grid sized for the capacity is thousands of workgroups that only exit)
    const int nrays = OBJ ? (*count < rays ? *count : rays) : rays;
    const size_t total = (size_t)nrays * N * 8;
    for (size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x; gid < total; gid += (size_t)gridDim.x * blockDim.x) {
    const size_t row = gid >> 3;          // sample row (ray-major)
    const int q = (int)(gid & 7);         // which 8-feature vector
    const int j = (int)(row / N), n = (int)(row % N);
    const int b = OBJ ? idx[j] : j;
    const float t0 = t_vals[(size_t)b * (N + 1) + n], t1 = t_vals[(size_t)b * (N + 1) + n + 1];
    float o[3] = {origins_s[b * 3], origins_s[b * 3 + 1], origins_s[b * 3 + 2]};
    float d[3] = {dirs_s[b * 3], dirs_s[b * 3 + 1], dirs_s[b * 3 + 2]};
    Gauss g = frustum_gaussian(t0, t1, o, d, radii[b], (contraction & DURF_ENC_CYLINDER) != 0);
    if (contraction & DURF_ENC_NO_INTEGRATION) g.var[0] = g.var[1] = g.var[2] = 0.0f;      // obbpose_model.py:164-165
    float w[10];
    if (OBJ) {
#pragma unroll
        for (int i = 0; i < 10; i++) w[i] = barf_w.w[i];
    } else {
        int nh = 0;
        for (int k = 0; k < K; k++) nh += hit[b * K + k];
        if (nh != 0) {                               // bkgd_mask = 1 - sum(masks) (model :205-210)
            const float m = 1.0f - (float)nh;
#pragma unroll
            for (int i = 0; i < 3; i++) { g.x[i] *= m; g.var[i] *= m; }
        }
        if (contraction & DURF_ENC_CONTRACT) contract_gaussian(g);
    }
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; e++) {
        const int p = q * 8 + e;
        float val;
        if (OBJ) {
            if (p < 3) val = g.x[p];
            else if (p < 63) { const int f = p - 3; val = w[f / 6] * ipe_feature(g, f); }   // mip.py:217-222
            else val = 0.0f;
        } else {
            val = (p < 60) ? ipe_feature(g, p) : 0.0f;
        }
        v[e] = val;
    }
    if (out_tile) {
        bf16x8 o8;
#pragma unroll
        for (int e = 0; e < 8; e++) o8[e] = (__bf16)v[e];
        *(bf16x8*)((char*)out_tile + tile_vec_offset(row, q, 4)) = o8;
    }
    if (out_f32) {
        const int dim = OBJ ? 63 : 60;
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const int p = q * 8 + e;
            if (p < dim) out_f32[row * dim + p] = v[e];
        }
    }
    }   // grid stride
}

// K3/K4, bf16-only fast path (sin/exp on the hardware transcendental units: abs error ~1e-4 at
// |y| ~ 300, far below the bf16 quantum 4e-3): ONE lane per sample (the Gaussian, the contraction and the 30
// exponentials are computed once instead of 8x), every feature index is a compile-time constant,
// and a wave writes, per 16-byte feature vector, two contiguous 512-byte runs of the tile layout.
// Same arithmetic as ipe_feature<true>; the safe_sin wrap is an exact fmod done with one fma
// (y - floor(y/t)*t is representable, so the single rounding of the fma returns it exactly).

#ifndef ENC_BLOCK
#define ENC_BLOCK 256
#endif
template <bool OBJ>
__global__ void __launch_bounds__(ENC_BLOCK)
k_encode_lane(int rays, int N, const int32_t* __restrict__ idx, const int32_t* __restrict__ count,
              const float* __restrict__ t_vals, const float* __restrict__ origins_s,
              const float* __restrict__ dirs_s, const float* __restrict__ radii,
              const int32_t* __restrict__ hit, int K, int contraction, BarfW barf_w,
              char* __restrict__ out_tile, size_t idx_stride, size_t out_stride) {
    if (OBJ && gridDim.y > 1) {                      // batched objects: blockIdx.y = object
        idx += blockIdx.y * idx_stride;
        count += blockIdx.y;
        out_tile += blockIdx.y * out_stride;
    }
    const size_t row = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int j = (int)(row / N), n = (int)(row % N);
    if (j >= rays) return;
    int b = j;
    if (OBJ || idx) {                        // compacted ray list (object rays; background rays of a de-duplicated batch)
        if (j >= *count) return;
        b = idx[j];
    }
    Gauss g;
    if constexpr (OBJ) {
        const float t0 = t_vals[(size_t)b * (N + 1) + n], t1 = t_vals[(size_t)b * (N + 1) + n + 1];
        float o[3] = {origins_s[b * 3], origins_s[b * 3 + 1], origins_s[b * 3 + 2]};
        float d[3] = {dirs_s[b * 3], dirs_s[b * 3 + 1], dirs_s[b * 3 + 2]};
        g = frustum_gaussian(t0, t1, o, d, radii[b], (contraction & DURF_ENC_CYLINDER) != 0);
        if (contraction & DURF_ENC_NO_INTEGRATION) g.var[0] = g.var[1] = g.var[2] = 0.0f;      // obbpose_model.py:164-165
    } else {
        g = bkgd_sample_gaussian(b, n, N, t_vals, origins_s, dirs_s, radii, hit, K, contraction);
    }
    char* base = out_tile + ((row >> 5) * 4 * 64 + (row & 31)) * 16;
    // every 16-byte vector is stored as soon as its last feature exists (enc_lane.h).  Non-temporal: the tile is written
    // once and read once by the fused MLP (measured 18.7 -> 17.9 us at 4096 rays, 125 -> 113 us at 32 768; the kernel is
    // bound by its store path: with the arithmetic compiled out the same stores take 14.8 / 105 us)
    lane_features<OBJ>(g, barf_w, [&](auto q_, const bf16x8& o8) {
        constexpr int q = decltype(q_)::value;
        __builtin_nontemporal_store(o8, (bf16x8*)(base + ((q >> 1) * 64 + (q & 1) * 32) * 16));
    });
}

// ---------------------------------------------------------------------------
// MipNerfModel.density_noise (obbpose_model.py:236-240): raw_density += density_noise * random.normal(key, shape) on the
// randomized path.  The standard-normal draws are the caller's (`normal` [rows]) or made here: Philox block (row, 1 + level,
// 0, 0) under the step's key -- word 1 of the counter keeps it clear of the sampling draws above, which use (i, 0, 0, 0) --
// through Box-Muller, z = sqrt(-2 ln u1) cos(2 pi u2) with u1 in (0, 1], u2 in [0, 1) from 24 bits each
// (oracle/philox_ref.py: density_draws).  The product and the sum are rounded separately, as the tensor expression is.
__global__ void __launch_bounds__(256)
k_density_noise(size_t rows, float* __restrict__ raw, float scale, const float* __restrict__ normal, unsigned seed_lo,
                unsigned seed_hi, unsigned level) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows) return;
    float z;
    if (normal != nullptr) {
        z = normal[i];
    } else {
        unsigned x0, x1;
        philox4x32_10((unsigned)i, seed_lo, seed_hi, x0, x1, 1u + level);
        const float u1 = (float)((x0 >> 8) + 1u) * 5.9604644775390625e-08f, u2 = u01_24(x1);
        z = sqrtf(-2.0f * logf(u1)) * cospif(2.0f * u2);
    }
    raw[i * 4 + 3] = __fadd_rn(raw[i * 4 + 3], __fmul_rn(scale, z));
}

extern "C" {

int durf_ray_setup(void* stream, int B, int K, const float* origins, const float* dirs,
                   const float* pose, const float* ext, float* origins_s, float* dirs_s,
                   int32_t* hit, float* zo) {
    DURF_REQUIRE(K >= 0 && K <= DURF_MAX_OBJ, "0 <= K <= DURF_MAX_OBJ");
    if (B <= 0) return 0;
    hipLaunchKernelGGL(k_ray_setup, dim3(durf_cdiv(B, 256)), dim3(256), 0, (hipStream_t)stream, B, K,
                       origins, dirs, pose, ext, origins_s, dirs_s, hit, zo);
    DURF_CHECK_LAUNCH("durf_ray_setup");
    return 0;
}

static int launch_prologue(void* stream, int B, int K, int N, const float* origins, const float* dirs, const float* pose,
                           const float* ext, float* origins_s, float* dirs_s, int32_t* hit, float* zo,
                           const float* viewdirs, void* view_bf16, const float* near, const float* far, const float* t_rand,
                           int lindisp, float* t_vals, float* pose_copy, float* zero_buf, size_t zero_count,
                           uint32_t seed_lo, uint32_t seed_hi, float* u_rand_out, const PackAll* pack, int K_pack,
                           float* zero_buf2 = nullptr, size_t zero_count2 = 0) {
    DURF_REQUIRE(K >= 0 && K <= DURF_MAX_OBJ, "0 <= K <= DURF_MAX_OBJ");
    DURF_REQUIRE(zero_buf == nullptr || ((size_t)zero_buf & 15) == 0, "zero_buf aligned to 16 bytes");
    DURF_REQUIRE(u_rand_out == nullptr || t_rand == nullptr, "the draws come from t_rand OR from the launch's own generator");
    DURF_REQUIRE(u_rand_out == nullptr || (size_t)B * (N + 1) < ((size_t)1 << 32), "in-kernel draws: 32-bit sample counter");
    PackAll pk{};
    int pb_bkgd = 0, pb_obj = 0;
    if (pack) {
        pk = *pack;
        if (pk.p_bkgd) pb_bkgd = (int)durf_cdiv((size_t)(MlpSpec<256>::TOTAL_CHUNKS + (pk.b_bkgd ? BwdSpec<256>::TOTAL_CHUNKS : 0)) * 64, 256);
        if (K_pack > 0) pb_obj = (int)durf_cdiv((size_t)(MlpSpec<128>::TOTAL_CHUNKS + (pk.b_obj ? BwdSpec<128>::TOTAL_CHUNKS : 0)) * 64, 256);
    }
    const int pack_blocks = pb_bkgd + (K_pack > 0 ? K_pack * pb_obj : 0);
    if (B <= 0) {       // an empty shard still owes its caller the zero-filled gradient (it is all-reduced and fed to clip + Adam)
        if (zero_buf && zero_count) {
            DURF_REQUIRE(hipMemsetAsync(zero_buf, 0, zero_count * sizeof(float), (hipStream_t)stream) == hipSuccess,
                         "zero fill of an empty shard's gradient");
        }
        if (zero_buf2 && zero_count2)
            DURF_REQUIRE(hipMemsetAsync(zero_buf2, 0, zero_count2 * sizeof(float), (hipStream_t)stream) == hipSuccess, "zero fill");
        if (pack_blocks == 0) return 0;
    }
    // the grid covers the largest of the three index spaces (rays, view-encoding features, sample positions)
    const size_t items = B > 0 ? std::max((size_t)B * (N + 1), (size_t)B * DURF_VIEW_DIM) : 0;
    const int nb_pro = (int)durf_cdiv(items, 256);
    hipLaunchKernelGGL(k_ray_prologue, dim3(nb_pro + pack_blocks), dim3(256), 0, (hipStream_t)stream, B,
                       K, N, origins, dirs, pose, ext, origins_s, dirs_s, hit, zo, viewdirs, (__bf16*)view_bf16, near,
                       far, t_rand, lindisp, t_vals, pose_copy, zero_buf, zero_buf ? zero_count : (size_t)0, seed_lo, seed_hi,
                       u_rand_out, pk, nb_pro, pb_bkgd, pb_obj > 0 ? pb_obj : 1, zero_buf2, zero_buf2 ? zero_count2 : (size_t)0);
    DURF_CHECK_LAUNCH("durf_ray_prologue");
    return 0;
}

int durf_ray_prologue(void* stream, int B, int K, int N, const float* origins, const float* dirs, const float* pose,
                      const float* ext, float* origins_s, float* dirs_s, int32_t* hit, float* zo,
                      const float* viewdirs, void* view_bf16, const float* near, const float* far, const float* t_rand,
                      int lindisp, float* t_vals, float* pose_copy, float* zero_buf, size_t zero_count,
                      uint32_t seed_lo, uint32_t seed_hi, float* u_rand_out) {
    return launch_prologue(stream, B, K, N, origins, dirs, pose, ext, origins_s, dirs_s, hit, zo, viewdirs, view_bf16, near, far,
                           t_rand, lindisp, t_vals, pose_copy, zero_buf, zero_count, seed_lo, seed_hi, u_rand_out, nullptr, 0);
}

int durf_ray_prologue_pack(void* stream, int B, int K, int N, const float* origins, const float* dirs, const float* pose,
                           const float* ext, float* origins_s, float* dirs_s, int32_t* hit, float* zo,
                           const float* viewdirs, void* view_bf16, const float* near, const float* far, const float* t_rand,
                           int lindisp, float* t_vals, float* pose_copy, float* zero_buf, size_t zero_count,
                           uint32_t seed_lo, uint32_t seed_hi, float* u_rand_out,
                           const float* bkgd_params, int in_bkgd, void* bkgd_fwd, void* bkgd_bwd, int K_pack,
                           const float* obj_params, size_t obj_param_stride, int in_obj, void* obj_fwd, void* obj_bwd,
                           float* zero_buf2, size_t zero_count2) {
    DURF_REQUIRE(bkgd_params == nullptr || (bkgd_fwd != nullptr && in_bkgd > 0 && in_bkgd <= DURF_ENC_DIM),
                 "background MLP: forward stream and 1 <= in_dim <= 64");
    DURF_REQUIRE(K_pack >= 0 && (K_pack == 0 || (obj_params != nullptr && obj_fwd != nullptr && in_obj > 0 && in_obj <= DURF_ENC_DIM)),
                 "object MLPs: parameters, forward streams and 1 <= in_dim <= 64");
    PackAll a{};
    a.p_bkgd = bkgd_params; a.f_bkgd = (bf16x8*)bkgd_fwd; a.b_bkgd = (bf16x8*)bkgd_bwd; a.in_bkgd = in_bkgd;
    a.p_obj = obj_params; a.f_obj = (bf16x8*)obj_fwd; a.b_obj = (bf16x8*)obj_bwd; a.in_obj = in_obj;
    a.p_stride = obj_param_stride; a.f_stride = durf_wpack_fwd_bytes(128); a.b_stride = durf_wpack_bwd_bytes(128);
    return launch_prologue(stream, B, K, N, origins, dirs, pose, ext, origins_s, dirs_s, hit, zo, viewdirs, view_bf16, near, far,
                           t_rand, lindisp, t_vals, pose_copy, zero_buf, zero_count, seed_lo, seed_hi, u_rand_out, &a, K_pack, zero_buf2,
                           zero_count2);
}

int durf_compact_hits(void* stream, int B, int K, const int32_t* hit, int32_t* idx,
                      int32_t* count, int32_t* slot) {
    if (K <= 0 || B <= 0) return 0;
    hipLaunchKernelGGL(k_compact_hits<false>, dim3(K), dim3(1024), 0, (hipStream_t)stream, B, K, 0, hit, idx,
                       count, slot, (int32_t*)nullptr);
    DURF_CHECK_LAUNCH("durf_compact_hits");
    return 0;
}

int durf_compact_classes(void* stream, int B, int K, int N, const int32_t* hit, int32_t* idx, int32_t* count,
                         int32_t* slot, int32_t* dyn) {
    DURF_REQUIRE(K >= 1 && K <= DURF_MAX_OBJ, "1 <= K <= DURF_MAX_OBJ");
    if (B <= 0) return 0;
    hipLaunchKernelGGL(k_compact_hits<true>, dim3(2), dim3(1024), 0, (hipStream_t)stream, B, K, N, hit, idx, count,
                       slot, dyn);
    DURF_CHECK_LAUNCH("durf_compact_classes");
    return 0;
}

int durf_compact_all(void* stream, int B, int K, int N, const int32_t* hit, int32_t* idx_obj, int32_t* count_obj,
                     int32_t* slot_obj, int32_t* idx_cls, int32_t* count_cls, int32_t* slot_cls, int32_t* dyn) {
    DURF_REQUIRE(K >= 1 && K <= DURF_MAX_OBJ, "1 <= K <= DURF_MAX_OBJ");
    if (B <= 0) return 0;
    hipLaunchKernelGGL(k_compact_all, dim3(K + 2), dim3(1024), 0, (hipStream_t)stream, B, K, N, hit, idx_obj, count_obj,
                       slot_obj, idx_cls, count_cls, slot_cls, dyn);
    DURF_CHECK_LAUNCH("durf_compact_all");
    return 0;
}

int durf_density_noise(void* stream, size_t rows, float* raw, float scale, const float* normal, uint32_t seed_lo,
                       uint32_t seed_hi, int level) {
    if (rows == 0 || scale == 0.0f) return 0;
    DURF_REQUIRE(raw != nullptr && level >= 0 && rows <= 0xffffffffull, "raw [rows,4], level >= 0");
    hipLaunchKernelGGL(k_density_noise, dim3(durf_cdiv(rows, 256)), dim3(256), 0, (hipStream_t)stream, rows, raw, scale, normal,
                       seed_lo, seed_hi, (unsigned)level);
    DURF_CHECK_LAUNCH("durf_density_noise");
    return 0;
}

int durf_sample_t(void* stream, int B, int N, const float* near, const float* far,
                  const float* t_rand, int lindisp, float* t_vals) {
    if (B <= 0) return 0;
    hipLaunchKernelGGL(k_sample_t, dim3(durf_cdiv((size_t)B * (N + 1), 256)), dim3(256), 0,
                       (hipStream_t)stream, B, N, near, far, t_rand, lindisp, t_vals);
    DURF_CHECK_LAUNCH("durf_sample_t");
    return 0;
}

int durf_view_enc(void* stream, int B, const float* viewdirs, void* out_bf16, float* out_f32) {
    if (B <= 0) return 0;
    hipLaunchKernelGGL(k_view_enc, dim3(durf_cdiv((size_t)B * DURF_VIEW_DIM, 256)), dim3(256), 0,
                       (hipStream_t)stream, B, viewdirs, (__bf16*)out_bf16, out_f32);
    DURF_CHECK_LAUNCH("durf_view_enc");
    return 0;
}

int durf_encode_bkgd(void* stream, int B, int N, const float* t_vals, const float* origins_s,
                     const float* dirs_s, const float* radii, const int32_t* hit, int K,
                     int contraction, void* out_tile, float* out_f32, const int32_t* idx, const int32_t* count) {
    if (B <= 0) return 0;
    DURF_REQUIRE(((size_t)B * N) % 32 == 0 || out_tile == nullptr, "B*N must be a multiple of 32");
    DURF_REQUIRE((idx == nullptr) == (count == nullptr), "idx and count go together");
    DURF_REQUIRE(idx == nullptr || out_f32 == nullptr, "the compacted ray list is for the bf16 tile output");
    if (out_f32)
        hipLaunchKernelGGL((k_encode<false>), dim3(durf_cdiv((size_t)B * N * 8, 256)), dim3(256), 0,
                           (hipStream_t)stream, B, N, nullptr, nullptr, t_vals, origins_s, dirs_s, radii,
                           hit, K, contraction, BarfW{}, (bf16x8*)out_tile, out_f32, (size_t)0, (size_t)0);
    else
        hipLaunchKernelGGL((k_encode_lane<false>), dim3(durf_cdiv((size_t)B * N, ENC_BLOCK)), dim3(ENC_BLOCK), 0,
                           (hipStream_t)stream, B, N, idx, count, t_vals, origins_s, dirs_s, radii,
                           hit, K, contraction, BarfW{}, (char*)out_tile, (size_t)0, (size_t)0);
    DURF_CHECK_LAUNCH("durf_encode_bkgd");
    return 0;
}

int durf_encode_obj(void* stream, int max_rays, int N, const int32_t* idx, const int32_t* count,
                    const float* t_vals, const float* origins_s, const float* dirs_s,
                    const float* radii, const float* barf_w, int flags, void* out_tile, float* out_f32) {
    return durf::launch_encode_obj(stream, 1, max_rays, N, idx, count, t_vals, origins_s, dirs_s, radii, barf_w, flags,
                                   out_tile, 0, out_f32);
}

int durf_encode_obj_f32_batch(void* stream, int K, int B, int N, const int32_t* idx, const int32_t* count,
                              const float* t_vals, const float* origins_s, const float* dirs_s, const float* radii,
                              const float* barf_w, int flags, float* enc) {
    DURF_REQUIRE(K >= 1 && K <= DURF_MAX_OBJ, "1 <= K <= DURF_MAX_OBJ");
    return durf::launch_encode_obj(stream, K, B, N, idx, count, t_vals, origins_s, dirs_s, radii, barf_w, flags, nullptr, 0,
                                   enc);
}

}  // extern "C"

namespace durf {

int launch_encode_obj(void* stream, int K, int max_rays, int N, const int32_t* idx, const int32_t* count,
                      const float* t_vals, const float* origins_s, const float* dirs_s, const float* radii,
                      const float* barf_w, int flags, void* out_tile, size_t out_stride, float* out_f32) {
    if (max_rays <= 0 || K <= 0) return 0;
    DURF_REQUIRE(K == 1 || out_f32 == nullptr || out_tile == nullptr, "batched object encoding: bf16 tiles OR fp32 features");
    BarfW bw;
    for (int i = 0; i < 10; i++) bw.w[i] = barf_w[i];
    if (out_f32)      // accurate-libm features, row-major [K, max_rays * N, 63]
        hipLaunchKernelGGL((k_encode<true>), dim3(std::min(durf_cdiv((size_t)max_rays * N * 8, 256), K > 1 ? 512u : 4096u), K), dim3(256), 0,
                           (hipStream_t)stream, max_rays, N, idx, count, t_vals, origins_s, dirs_s, radii,
                           nullptr, 0, flags & (DURF_ENC_NO_INTEGRATION | DURF_ENC_CYLINDER), bw, (bf16x8*)out_tile, out_f32,
                           (size_t)max_rays, (size_t)max_rays * N * 63);
    else
        hipLaunchKernelGGL((k_encode_lane<true>), dim3(durf_cdiv((size_t)max_rays * N, ENC_BLOCK), K), dim3(ENC_BLOCK), 0,
                           (hipStream_t)stream, max_rays, N, idx, count, t_vals, origins_s, dirs_s, radii,
                           nullptr, 0, flags & (DURF_ENC_NO_INTEGRATION | DURF_ENC_CYLINDER), bw, (char*)out_tile, (size_t)max_rays, out_stride);
    DURF_CHECK_LAUNCH("durf_encode_obj");
    return 0;
}

}  // namespace durf
