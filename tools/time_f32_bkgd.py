"""Times the exact-fp32 background forward / backward / weight gradients (mlp_f32.hip, W = 256) against the row count:
    python tools/time_f32_bkgd.py [rays ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from durf_amd import ops
dev = torch.device('cuda:0')
N, W, IN = 128, 256, 60
torch.manual_seed(0)
flat = (torch.rand(ops.mlp_param_count(W, IN), device=dev) - 0.5) * 0.2
ws = ops.mlp_f32_pack(W, IN, flat)


def timeit(fn, n=3):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for B in [int(a) for a in sys.argv[1:]] or [256, 1024, 4096]:
    rows = B * N
    enc = torch.randn(rows, IN, device=dev) * 0.5
    view = torch.randn(B, 27, device=dev) * 0.5
    draw = torch.randn(rows, 4, device=dev) * 1e-3
    ti = timeit(lambda: ops.mlp_fwd_f32(W, IN, rows, N, enc, view, flat, wstream=ws))
    out = {}
    def ftrain():
        out['r'], out['act'] = ops.mlp_fwd_f32(W, IN, rows, N, enc, view, flat, want_act=True, wstream=ws)
    tt = timeit(ftrain)
    def fb():
        out['dz'] = ops.mlp_bwd_f32(W, IN, rows, N, draw, flat, out['act'], wstream=ws)
    tb = timeit(fb)
    g = torch.zeros_like(flat)
    td = timeit(lambda: ops.mlp_dw_f32(W, IN, rows, N, out['act'], out['dz'], g))
    fl = 2.0 * 591872 * rows
    print('rays %5d: fwd %9.2f ms (%.1f TF)  fwd+records %9.2f ms  bwd %9.2f ms  dW %9.2f ms' % (B, ti, fl / ti / 1e9, tt, tb, td))
