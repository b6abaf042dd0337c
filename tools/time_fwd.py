"""Times the fused MLP forward (inference and training instantiations) at 4096 rays x 128 samples: one line.
Used with tools/ab_variants.sh on dissection builds (FWD_X_* in csrc/mlp_fwd.hip)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from durf_amd import ops
dev = torch.device('cuda:0')
rows, N, W, IN = int(os.environ.get('ROWS', 4096 * 128)), 128, 256, 60
torch.manual_seed(0)
flat = (torch.rand(ops.mlp_param_count(W, IN), device=dev) - 0.5) * 0.2
wf, wb = ops.pack_weights(W, IN, flat, want_bwd=True)
enc = (torch.randn(rows * 64, device=dev) * 0.5).to(torch.bfloat16)
view = (torch.randn(rows // N * 32, device=dev) * 0.5).to(torch.bfloat16)
raw = torch.empty(rows, 4, device=dev)
stash = torch.empty(ops.mlp_stash_bytes(W, rows), dtype=torch.uint8, device=dev)
mask = torch.empty(ops.mlp_mask_bytes(rows), dtype=torch.uint8, device=dev)
draw = torch.randn(rows, 4, device=dev) * 1e-3


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


ti = timeit(lambda: ops.mlp_fwd(W, rows, N, enc, view, wf, raw=raw))
tt = timeit(lambda: ops.mlp_fwd(W, rows, N, enc, view, wf, raw=raw, stash=stash, relu_mask=mask))
tb = timeit(lambda: ops.mlp_bwd(W, rows, N, draw, wb, mask))
print('fwd inference %7.1f us   fwd train %7.1f us   bwd %7.1f us' % (ti, tt, tb))
