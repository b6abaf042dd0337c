"""How the persistent fused MLP kernels scale with the number of 256-sample blocks (rounds of 256 workgroups):
python tools/time_mlp_rounds.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from durf_amd import ops

dev = torch.device('cuda:0')
B, N, W, IN = 4096, 128, 256, 60
rows = B * N
torch.manual_seed(0)
flat = (torch.rand(ops.mlp_param_count(W, IN), device=dev) - 0.5) * 0.2
wf, wb = ops.pack_weights(W, IN, flat, want_bwd=True)
enc = (torch.randn(rows * 64, device=dev) * 0.5).to(torch.bfloat16)
view = (torch.randn(B * 32, device=dev) * 0.5).to(torch.bfloat16)
stash = torch.empty(ops.mlp_stash_bytes(W, rows), dtype=torch.uint8, device=dev)
mask = torch.empty(ops.mlp_mask_bytes(rows), dtype=torch.uint8, device=dev)
draw = torch.randn(rows, 4, device=dev) * 1e-2
idx = torch.arange(B, dtype=torch.int32, device=dev)

def timeit(fn, n=10):
    for _ in range(3): fn()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    ev[0].record()
    for i in range(n):
        fn(); ev[i + 1].record()
    torch.cuda.synchronize()
    return sorted(ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(n))[n // 2]

for rep in range(2):
    for rays in ((4096, 3072, 3674, 2048, 3584, 1024, 3840, 512) if rep == 0 else (512, 1024, 2048, 3072, 3584, 3674, 3840, 4096)):
        cnt = torch.tensor([rays], dtype=torch.int32, device=dev)
        tf = timeit(lambda: ops.mlp_fwd(W, rows, N, enc, view, wf, ray_idx=idx, count=cnt, stash=stash, relu_mask=mask))
        ti = timeit(lambda: ops.mlp_fwd(W, rows, N, enc, view, wf, ray_idx=idx, count=cnt))
        tb = timeit(lambda: ops.mlp_bwd(W, rows, N, draw, wb, mask, ray_idx=idx, count=cnt))
        print('rays %4d = %4d blocks = %.2f rounds: fwd-train %.1f us  fwd-infer %.1f us  bwd %.1f us' % (
            rays, rays * N // 256, rays * N / 256 / 256, tf, ti, tb))
