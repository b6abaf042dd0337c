"""Per-workgroup timeline of the weight-gradient launch.  The instrumentation is not in the product kernels: apply
tools/experiments/kernel_instrumentation.patch (git apply), then  tools/build_variant.sh trace -DDW_TRACE;
DURF_LIB_PATH=durf_amd/variants/libdurf_trace.so python tools/dw_trace.py;  git checkout durf_amd/csrc."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from durf_amd import ops, _lib

dev = torch.device('cuda:0')
rows, N, W, IN = 4096 * 128, 128, 256, 60
B = rows // N
torch.manual_seed(0)
flat = (torch.rand(ops.mlp_param_count(W, IN), device=dev) - 0.5) * 0.2
wf, wb = ops.pack_weights(W, IN, flat, want_bwd=True)
enc = (torch.randn(rows * 64, device=dev) * 0.5).to(torch.bfloat16)
view = (torch.randn(B * 32, device=dev) * 0.5).to(torch.bfloat16)
raw = torch.empty(rows, 4, device=dev)
stash = torch.empty(ops.mlp_stash_bytes(W, rows), dtype=torch.uint8, device=dev)
mask = torch.empty(ops.mlp_mask_bytes(rows), dtype=torch.uint8, device=dev)
ops.mlp_fwd(W, rows, N, enc, view, wf, raw=raw, stash=stash, relu_mask=mask)
draw = torch.randn(rows, 4, device=dev) * 1e-3
dz, dz_out = ops.mlp_bwd(W, rows, N, draw, wb, mask)
view_tile = ops.expand_view(rows, N, view)
part, bpart = ops.dw_buffers(W, dev)
for _ in range(3):
    ops.mlp_dw(W, rows, N, [enc], [view_tile], [stash], [dz], [dz_out], part, bpart)
torch.cuda.synchronize()
buf = np.zeros(4 * 4096, dtype=np.uint64)
L = _lib.lib()
L.durf_debug_dw_trace.argtypes = [C.c_void_p]
assert L.durf_debug_dw_trace(buf.ctypes.data) == 0
t = buf.reshape(-1, 4)
t = t[t[:, 1] > 0]
t0 = t[:, 0].min()
start = (t[:, 0] - t0).astype(np.float64) / 100.0      # us (100 MHz wall clock)
end = (t[:, 1] - t0).astype(np.float64) / 100.0
job = (t[:, 3] >> np.uint64(32)).astype(int)
hw = (t[:, 2] & np.uint64(0xffffffff)).astype(np.int64)
xcc = (t[:, 2] >> np.uint64(32)).astype(np.int64) & 0xf
cu = (hw >> 8) & 0xf
sh = (hw >> 12) & 0x1
se = (hw >> 13) & 0x7
cuid = ((xcc * 8 + se) * 2 + sh) * 16 + cu
print('workgroups', len(t), ' span %.1f us' % end.max(), ' distinct CUs', len(set(cuid.tolist())))
dur = end - start
for j in sorted(set(job.tolist())):
    m = job == j
    print('job %2d: n=%3d  dur mean %.1f  min %.1f  max %.1f   start min %.1f max %.1f' % (
        j, m.sum(), dur[m].mean(), dur[m].min(), dur[m].max(), start[m].min(), start[m].max()))
# per-CU busy time and finish
ids = sorted(set(cuid.tolist()))
busy = np.array([dur[cuid == c].sum() for c in ids])
fin = np.array([end[cuid == c].max() for c in ids])
cnt = np.array([(cuid == c).sum() for c in ids])
print('per-CU: WGs min %d max %d;  busy mean %.1f min %.1f max %.1f;  finish min %.1f max %.1f' % (
    cnt.min(), cnt.max(), busy.mean(), busy.min(), busy.max(), fin.min(), fin.max()))
h, e = np.histogram(end, bins=12)
print('end-time histogram:', list(zip(e[:-1].round(0).tolist(), h.tolist())))
