"""TEST INFRASTRUCTURE (never imported by the product): a numpy restatement of the stratified-sampling draws that
durf_ray_prologue makes in-kernel (include/durf_hip.h; csrc/rays.hip philox4x32_10 / u01_24).

The reference draws this noise inside its program from a jax PRNG key (internal/mip.py:364 `jax.random.uniform(key,
[batch_size, num_samples + 1])`, internal/math.py:257-260 the resampling jitter); jax's threefry stream cannot be reproduced
here (jax is absent), so parity of everything downstream is tested with injected draws.  What this file pins is
the generator itself: Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11) against
the known-answer vectors of its reference implementation (Random123 kat_vectors, rows `philox4x32 10`), and the mapping of
its output words to the two uniform draws of sample position i."""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)


def philox4x32_10(ctr, key):
    """ctr [...,4] uint32, key [...,2] uint32 -> [...,4] uint32"""
    c = [np.asarray(ctr[..., i], dtype=np.uint32) for i in range(4)]
    k0, k1 = np.asarray(key[..., 0], dtype=np.uint32), np.asarray(key[..., 1], dtype=np.uint32)
    with np.errstate(over='ignore'):
        for _ in range(10):
            p0 = c[0].astype(np.uint64) * M0
            p1 = c[2].astype(np.uint64) * M1
            hi0, lo0 = (p0 >> np.uint64(32)).astype(np.uint32), p0.astype(np.uint32)
            hi1, lo1 = (p1 >> np.uint64(32)).astype(np.uint32), p1.astype(np.uint32)
            c = [hi1 ^ c[1] ^ k0, lo1, hi0 ^ c[3] ^ k1, lo0]
            k0 = (k0 + W0).astype(np.uint32)
            k1 = (k1 + W1).astype(np.uint32)
    return np.stack(c, axis=-1)


# Random123 known-answer vectors for philox4x32 with 10 rounds: (counter, key) -> output
KAT = [
    ((0x00000000, 0x00000000, 0x00000000, 0x00000000), (0x00000000, 0x00000000),
     (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
    ((0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff), (0xffffffff, 0xffffffff),
     (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
    ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
     (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
]


def step_draws(seed, B, N):
    """(t_rand [B,N+1], u_rand [B,N+1]) float32 exactly as durf_ray_prologue(seed, u_rand_out != NULL) makes them: block i =
    philox(counter (i, 0, 0, 0), key (seed & 2^32-1, seed >> 32)); word 0 -> level-0 jitter of sample position i, word 1 ->
    resampling draw i; each as (x >> 8) * 2^-24 in [0, 1)."""
    n = B * (N + 1)
    ctr = np.zeros((n, 4), dtype=np.uint32)
    ctr[:, 0] = np.arange(n, dtype=np.uint32)
    s = int(seed) & 0xFFFFFFFFFFFFFFFF
    key = np.tile(np.array([s & 0xFFFFFFFF, s >> 32], dtype=np.uint32), (n, 1))
    x = philox4x32_10(ctr, key)
    u = lambda w: ((w >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24)).reshape(B, N + 1)
    return u(x[:, 0]), u(x[:, 1])


def step_draws_planes(seed, B, N):
    """(t_rand [B,N+1], u_rand [3,B,N+1]): step_draws with every resampling plane -- words 1, 2, 3 of block i are the draws of
    the resamples behind levels 0, 1, 2 (durf_ray_prologue's u_rand_out)"""
    n = B * (N + 1)
    ctr = np.zeros((n, 4), dtype=np.uint32)
    ctr[:, 0] = np.arange(n, dtype=np.uint32)
    s = int(seed) & 0xFFFFFFFFFFFFFFFF
    key = np.tile(np.array([s & 0xFFFFFFFF, s >> 32], dtype=np.uint32), (n, 1))
    x = philox4x32_10(ctr, key)
    u = lambda w: ((w >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24)).reshape(B, N + 1)
    return u(x[:, 0]), np.stack([u(x[:, 1]), u(x[:, 2]), u(x[:, 3])])


def density_draws(seed, rows, level):
    """Standard-normal draws [rows] float32 as durf_density_noise(normal = NULL) makes them (MipNerfModel.density_noise,
    /root/reference/internal/obbpose_model.py:236-240 draws them with jax.random.normal on the step's key): Philox block
    (row, 1 + level, 0, 0) under the step's key -- word 1 of the counter keeps them clear of step_draws' (i, 0, 0, 0) --
    through Box-Muller: u1 = ((x0 >> 8) + 1) 2^-24 in (0, 1], u2 = (x1 >> 8) 2^-24 in [0, 1), z = sqrt(-2 ln u1) cos(2 pi u2).
    Evaluated in float64 here and rounded once: the device's fp32 logf / cospif / sqrtf agree to a few ulp (the tolerance the
    test states)."""
    ctr = np.zeros((rows, 4), dtype=np.uint32)
    ctr[:, 0] = np.arange(rows, dtype=np.uint32)
    ctr[:, 1] = np.uint32(1 + level)
    s = int(seed) & 0xFFFFFFFFFFFFFFFF
    key = np.tile(np.array([s & 0xFFFFFFFF, s >> 32], dtype=np.uint32), (rows, 1))
    x = philox4x32_10(ctr, key)
    u1 = ((x[:, 0] >> np.uint32(8)).astype(np.float64) + 1.0) * 2.0 ** -24
    u2 = (x[:, 1] >> np.uint32(8)).astype(np.float64) * 2.0 ** -24
    return (np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)).astype(np.float32)
