"""Times one launch of the object MLP forward (W = 128) at 72, 1 and 8 blocks: the latency of a single 12-layer block.
    python tools/time_fwd128.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from durf_amd import ops
dev = torch.device('cuda:0')
W, IN, N = 128, 63, 128
for rows in (72 * 256, 256, 8 * 256):
    flat = (torch.rand(ops.mlp_param_count(W, IN), device=dev) - 0.5) * 0.2
    wf, wb = ops.pack_weights(W, IN, flat, want_bwd=True)
    enc = (torch.randn(rows * 64, device=dev) * 0.5).to(torch.bfloat16)
    B = max(rows // N, 1)
    view = (torch.randn(B * 32, device=dev) * 0.5).to(torch.bfloat16)
    raw = torch.empty(rows, 4, device=dev)
    stash = torch.empty(ops.mlp_stash_bytes(W, rows), dtype=torch.uint8, device=dev)
    mask = torch.empty(ops.mlp_mask_bytes(rows), dtype=torch.uint8, device=dev)
    def timeit(fn, n=50):
        for _ in range(5): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3
    t = timeit(lambda: ops.mlp_fwd(W, rows, N, enc, view, wf, raw=raw, stash=stash, relu_mask=mask))
    print('W=128 rows %6d (%d blocks): fwd train %.1f us' % (rows, rows // 256, t))
