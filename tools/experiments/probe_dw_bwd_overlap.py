"""The review's bounded experiment on the dominant kernels: can the weight-gradient launch of ONE level (HBM-bound, ~6 TB/s)
run beside the fused backward of the other level (MFMA-heavy, ~4 TB/s) on disjoint sets of CUs (both need > 128 KB of LDS,
so they cannot share a CU)?  Streams with CU masks (hipExtStreamCreateWithCUMask), the real kernels on cfg3-sized
buffers (4096 rays x 128 samples), wall time from a common start event to the later of the two finishing.
    python tools/probe_dw_bwd_overlap.py         (kill criterion: >= 2 % of a 4.3 ms step = 86 us saved vs back to back)"""
import ctypes, os, sys
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
ops.mlp_fwd(W, rows, N, enc, view, wf, stash=stash, relu_mask=mask)
draw = torch.randn(rows, 4, device=dev) * 1e-2
dz, dz_out = ops.mlp_bwd(W, rows, N, draw, wb, mask)
view_tile = ops.expand_view(rows, N, view)
part, bpart = ops.dw_buffers(W, dev)
torch.cuda.synchronize()

hip = ctypes.CDLL('libamdhip64.so')


def masked_stream(lo, hi):
    words = (ctypes.c_uint32 * 8)()
    for cu in range(lo, hi):
        words[cu // 32] |= 1 << (cu % 32)
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value, device=dev)


def bwd():
    return ops.mlp_bwd(W, rows, N, draw, wb, mask)


def dw_one_level():
    ops.mlp_dw(W, rows, N, [enc], [view_tile], [stash], [dz], [dz_out], part, bpart)


def wall(fn_a, sa, fn_b, sb, reps=10):
    """both launches issued back to back from the host; time from a common start to both done"""
    main = torch.cuda.current_stream()
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        e0, ea, eb = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record(main)
        sa.wait_event(e0)
        sb.wait_event(e0)
        with torch.cuda.stream(sa):
            keep = fn_a()
            ea.record(sa)
        with torch.cuda.stream(sb):
            fn_b()
            eb.record(sb)
        torch.cuda.synchronize()
        best = min(best, max(e0.elapsed_time(ea), e0.elapsed_time(eb)))
    return best * 1e3


def alone(fn, s, reps=10):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(s):
            e0.record(s)
            keep = fn()
            e1.record(s)
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best * 1e3


full = torch.cuda.Stream(device=dev)
full2 = torch.cuda.Stream(device=dev)
ta, tb = alone(bwd, full), alone(dw_one_level, full)
print('alone on all 256 CUs: backward %.0f us, weight gradients of one level %.0f us, back to back %.0f us' % (ta, tb, ta + tb))
print('two plain streams (no CU mask): %.0f us' % wall(bwd, full, dw_one_level, full2))
for na in (96, 128, 160, 192, 224):
    sa, sb = masked_stream(0, na), masked_stream(na, 256)
    a1, b1 = alone(bwd, sa), alone(dw_one_level, sb)
    both = wall(bwd, sa, dw_one_level, sb)
    print('backward on CUs [0,%d), weight gradients on [%d,256): alone %.0f / %.0f us, together %.0f us  -> vs back to back on the whole GPU %+.0f us' % (
        na, na, a1, b1, both, both - (ta + tb)))
