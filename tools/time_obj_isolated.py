"""One launch of the object MLP forward / backward (W = 128, compacted ray list, weights hot in L2), back to back:
the kernels' own latency at small sizes, M-split (DURF_OBJ_MSPLIT=1) against sample-split (0).
    python tools/time_obj_isolated.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from durf_amd import ops
dev = torch.device('cuda:0')
W, IN, N = 128, 63, 128


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for B, hit in ((512, 2), (512, 16), (512, 52), (1024, 104)):
    rows = B * N
    flat = (torch.rand(ops.mlp_param_count(W, IN), device=dev) - 0.5) * 0.2
    wf, wb = ops.pack_weights(W, IN, flat, want_bwd=True)
    enc = (torch.randn(ops.tile_rows(rows) * 64, device=dev) * 0.5).to(torch.bfloat16)
    view = (torch.randn(B * 32, device=dev) * 0.5).to(torch.bfloat16)
    idx = torch.randperm(B, device=dev)[:B].to(torch.int32).contiguous()
    count = torch.tensor([hit], dtype=torch.int32, device=dev)
    raw = torch.empty(rows, 4, device=dev)
    stash = torch.empty(ops.mlp_stash_bytes(W, rows), dtype=torch.uint8, device=dev)
    mask = torch.empty(ops.mlp_mask_bytes(rows), dtype=torch.uint8, device=dev)
    draw = torch.randn(rows, 4, device=dev) * 1e-3
    for ms in ('1', '0'):
        os.environ['DURF_OBJ_MSPLIT'] = ms
        tf = timeit(lambda: ops.mlp_fwd(W, rows, N, enc, view, wf, ray_idx=idx, count=count, raw=raw, stash=stash, relu_mask=mask))
        ti = timeit(lambda: ops.mlp_fwd(W, rows, N, enc, view, wf, ray_idx=idx, count=count, raw=raw))
        tb = timeit(lambda: ops.mlp_bwd(W, rows, N, draw, wb, mask, ray_idx=idx, count=count))
        print('B %4d  hit rays %3d (%3d pairs)  msplit=%s   fwd train %5.1f us   fwd inference %5.1f us   bwd %5.1f us' %
              (B, hit, hit * N // 64, ms, tf, ti, tb))

# the same with the caches flushed before every launch (a 512 MB fill: what the object launches of a training step meet
# after a background kernel has streamed gigabytes through L2 / the Infinity Cache)
big = torch.empty(128 * 1024 * 1024, device=dev)
B, hit = 1024, 52
rows = B * N
flat = (torch.rand(ops.mlp_param_count(W, IN), device=dev) - 0.5) * 0.2
wf, wb = ops.pack_weights(W, IN, flat, want_bwd=True)
enc = (torch.randn(ops.tile_rows(rows) * 64, device=dev) * 0.5).to(torch.bfloat16)
view = (torch.randn(B * 32, device=dev) * 0.5).to(torch.bfloat16)
idx = torch.randperm(B, device=dev)[:B].to(torch.int32).contiguous()
count = torch.tensor([hit], dtype=torch.int32, device=dev)
raw = torch.empty(rows, 4, device=dev)
stash = torch.empty(ops.mlp_stash_bytes(W, rows), dtype=torch.uint8, device=dev)
mask = torch.empty(ops.mlp_mask_bytes(rows), dtype=torch.uint8, device=dev)
draw = torch.randn(rows, 4, device=dev) * 1e-3


def cold(fn, n=20):
    tot = 0.0
    for _ in range(n + 2):
        big.fill_(1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        if _ >= 2:
            tot += e0.elapsed_time(e1)
    return tot / n * 1e3


for ms in ('1', '0'):
    os.environ['DURF_OBJ_MSPLIT'] = ms
    tf = cold(lambda: ops.mlp_fwd(W, rows, N, enc, view, wf, ray_idx=idx, count=count, raw=raw, stash=stash, relu_mask=mask))
    tb = cold(lambda: ops.mlp_bwd(W, rows, N, draw, wb, mask, ray_idx=idx, count=count))
    print('cold caches, %d hit rays: msplit=%s   fwd train %5.1f us   bwd %5.1f us' % (hit, ms, tf, tb))
