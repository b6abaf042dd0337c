"""Micro-benchmarks of the individual kernels at BASELINE cfg2 shapes (B=4096, N=128)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from durf_amd import ops, synthetic

dev = torch.device('cuda:0')
B, N, K = int(os.environ.get('B', 4096)), 128, 1


def timeit(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


b = synthetic.make_batch(B, K, seed=1)
db = synthetic.device_batch(b, dev)
rays = db['rays']
pose = db['init'][b['ts']].contiguous()
o_s, d_s, hit, zo = ops.ray_setup(rays.origins, rays.directions, pose, db['ext'])
idx, count, slot = ops.compact_hits(hit)
view = ops.view_enc(rays.viewdirs)
radii = rays.radii.reshape(-1).contiguous()
t_vals = ops.sample_t(rays.near.reshape(-1), rays.far.reshape(-1), N)
rows = B * N
print('hit fraction', b['hit_fraction'], 'count', count.tolist())

t = timeit(lambda: ops.ray_setup(rays.origins, rays.directions, pose, db['ext']))
print('ray_setup      %8.1f us' % (t * 1e6))
t = timeit(lambda: ops.encode_bkgd(t_vals, o_s, d_s, radii, hit, True))
print('encode_bkgd    %8.1f us  %.2f TB/s (bf16 out: 15928 B/ray)' % (t * 1e6, B * 15928 / t / 1e12))
enc_b, _ = ops.encode_bkgd(t_vals, o_s, d_s, radii, hit, True)
for width, in_dim in ((256, 60), (128, 63)):
    flat = (torch.rand(ops.mlp_param_count(width, in_dim), device=dev) - 0.5) * 0.1
    wf = ops.pack_weights(width, in_dim, flat)
    macs = 591872 if width == 256 else 167552
    raw = torch.empty(rows, 4, device=dev)
    t = timeit(lambda: ops.mlp_fwd(width, rows, N, enc_b, view, wf, raw=raw), n=10)
    print('mlp_fwd W=%d   %8.1f us  %.1f TFLOP/s (%.1f%% of 2.5 PF)' % (width, t * 1e6, 2 * macs * rows / t / 1e12, 2 * macs * rows / t / 2.5e15 * 100))
    stash = torch.empty(ops.mlp_stash_bytes(width, rows), dtype=torch.uint8, device=dev)
    mask = torch.empty(ops.mlp_mask_bytes(rows), dtype=torch.uint8, device=dev)
    t = timeit(lambda: ops.mlp_fwd(width, rows, N, enc_b, view, wf, raw=raw, stash=stash, relu_mask=mask), n=10)
    print('mlp_fwd W=%d +stash %8.1f us  %.1f TFLOP/s, stash %.2f GB -> %.2f TB/s' % (width, t * 1e6, 2 * macs * rows / t / 1e12, stash.numel() / 1e9, stash.numel() / t / 1e12))
raw_b = torch.randn(rows, 4, device=dev)
t = timeit(lambda: ops.composite_fwd(raw_b, [], slot, t_vals, d_s, -1.0, 0, want_t=False))
print('composite_fwd  %8.1f us  %.2f TB/s (3108 B/ray)' % (t * 1e6, B * 3108 / t / 1e12))
out = ops.composite_fwd(raw_b, [], slot, t_vals, d_s, -1.0, 0)
t = timeit(lambda: ops.resample(t_vals, out[3], 0.01))
print('resample       %8.1f us  %.2f TB/s (1544 B/ray)' % (t * 1e6, B * 1544 / t / 1e12))
