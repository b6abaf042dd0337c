"""Times the three training kernels of the 256-wide MLP (fwd+stash, bwd, dW) at cfg2 shapes on
random data, with HIP events, and prints the HBM rate each sustains.  Used for kernel tuning.
    python tools/bench_mlp_train.py [rows]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from durf_amd import ops

dev = torch.device('cuda:0')
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 4096 * 128
N = 128
W, IN = 256, 60
B = rows // N


def timeit(fn, n=10, warm=3):
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


torch.manual_seed(0)
flat = (torch.rand(ops.mlp_param_count(W, IN), device=dev) - 0.5) * 0.2
wf, wb = ops.pack_weights(W, IN, flat, want_bwd=True)
enc = (torch.randn(rows * 64, device=dev) * 0.5).to(torch.bfloat16)
view = (torch.randn(B * 32, device=dev) * 0.5).to(torch.bfloat16)
raw = torch.empty(rows, 4, device=dev)
stash = torch.empty(ops.mlp_stash_bytes(W, rows), dtype=torch.uint8, device=dev)
mask = torch.empty(ops.mlp_mask_bytes(rows), dtype=torch.uint8, device=dev)
flops = 2 * 591872 * rows
t = timeit(lambda: ops.mlp_fwd(W, rows, N, enc, view, wf, raw=raw))
print('fwd (inference) %8.1f us  %6.1f TFLOP/s' % (t * 1e6, flops / t / 1e12))
t = timeit(lambda: ops.mlp_fwd(W, rows, N, enc, view, wf, raw=raw, stash=stash, relu_mask=mask))
by = stash.numel() + mask.numel() + rows * 128
print('fwd (train)     %8.1f us  %6.1f TFLOP/s  %.2f GB -> %.2f TB/s' % (t * 1e6, flops / t / 1e12, by / 1e9, by / t / 1e12))
draw = torch.randn(rows, 4, device=dev) * 1e-3
t = timeit(lambda: ops.mlp_bwd(W, rows, N, draw, wb, mask))
dz, dz_out = ops.mlp_bwd(W, rows, N, draw, wb, mask)
by = dz.numel() * dz.element_size() + dz_out.numel() * dz_out.element_size() + mask.numel()
print('bwd             %8.1f us  %6.1f TFLOP/s  %.2f GB -> %.2f TB/s' % (t * 1e6, flops / t / 1e12, by / 1e9, by / t / 1e12))
view_tile = ops.expand_view(rows, N, view)
part, bpart = ops.dw_buffers(W, dev)
t = timeit(lambda: ops.mlp_dw(W, rows, N, [enc], [view_tile], [stash], [dz], [dz_out], part, bpart))
by = (stash.numel() + dz.numel() * dz.element_size() + dz_out.numel() * dz_out.element_size()
      + enc.numel() * 2 + view_tile.numel() * view_tile.element_size())
print('dW              %8.1f us  %6.1f TFLOP/s  %.2f GB -> %.2f TB/s' % (t * 1e6, flops / t / 1e12, by / 1e9, by / t / 1e12))
grad = torch.empty(ops.mlp_param_count(W, IN), device=dev)
t = timeit(lambda: ops.mlp_dw_finalize(W, IN, rows, N, 1, part, bpart, grad, flat))
print('dW finalize     %8.1f us' % (t * 1e6))
stash2, dz2 = stash.clone(), dz.clone()
t = timeit(lambda: ops.mlp_dw(W, rows, N, [enc, enc], [view_tile] * 2, [stash, stash2], [dz, dz2], [dz_out] * 2, part, bpart))
print('dW (2 levels)   %8.1f us  %6.1f TFLOP/s  %.2f GB -> %.2f TB/s' % (t * 1e6, 2 * flops / t / 1e12, 2 * by / 1e9, 2 * by / t / 1e12))
