"""Per-workgroup / per-block timeline of the fused forward (needs tools/experiments/kernel_instrumentation.patch applied
and a -DFWD_TRACE variant build, see tools/dw_trace.py)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from durf_amd import ops, _lib

dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
train = len(sys.argv) > 2 and sys.argv[2] == 'train'
N, W, IN = 128, 256, 60
rows = B * N
torch.manual_seed(0)
flat = (torch.rand(ops.mlp_param_count(W, IN), device=dev) - 0.5) * 0.2
wf = ops.pack_weights(W, IN, flat)
enc = (torch.randn(rows * 64, device=dev) * 0.5).to(torch.bfloat16)
view = (torch.randn(B * 32, device=dev) * 0.5).to(torch.bfloat16)
raw = torch.empty(rows, 4, device=dev)
stash = torch.empty(ops.mlp_stash_bytes(W, rows), dtype=torch.uint8, device=dev) if train else None
mask = torch.empty(ops.mlp_mask_bytes(rows), dtype=torch.uint8, device=dev) if train else None
for _ in range(4):
    ops.mlp_fwd(W, rows, N, enc, view, wf, raw=raw, stash=stash, relu_mask=mask)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
ops.mlp_fwd(W, rows, N, enc, view, wf, raw=raw, stash=stash, relu_mask=mask)
e1.record()
torch.cuda.synchronize()
buf = np.zeros(256 * 80, dtype=np.uint64)
L = _lib.lib()
L.durf_debug_fwd_trace.argtypes = [C.c_void_p]
assert L.durf_debug_fwd_trace(buf.ctypes.data) == 0
t = buf.reshape(256, 80).astype(np.float64)
nb = min(rows // 256 // 256, 76)
t0 = t[:, 78].min()
start, end = t[:, 78] - t0, t[:, 79] - t0
blk = t[:, :nb] - t0
print('event time %.1f us; span %.1f ticks (100 MHz nominal); WG start min %.1f max %.1f; end min %.1f max %.1f' % (
    e0.elapsed_time(e1) * 1e3, end.max(), start.min(), start.max(), end.min(), end.max()))
d = np.diff(np.concatenate([blk, end[:, None]], 1), axis=1)
print('per-block duration (ticks), mean over WGs:', np.round(d.mean(0), 1).tolist()[:16], '...' if nb > 16 else '')
print('per-block duration max over WGs:       ', np.round(d.max(0), 1).tolist()[:16])
print('first block start - WG start: mean %.2f' % (blk[:, 0] - start).mean())
