"""Round 5: what bounds the fused kernels is the board's power cap -- an energy model from measurements.
Loops one kernel at a time for a few seconds while sampling rocm-smi (socket power, sclk): us / launch, W, GHz, J / launch.
    python3 tools/experiments/power_model.py [seconds per kernel]        (DURF_LIB_PATH selects a store variant)"""
import os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import re
import torch
from durf_amd import ops
dev = torch.device('cuda:0')
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
rows, N, W, IN = 4096 * 128, 128, 256, 60
B = rows // N


def smi():
    out = subprocess.run(['rocm-smi', '--showclocks', '--showpower'], capture_output=True, text=True).stdout
    p = re.search(r'Power \(W\): ([\d.]+)', out)
    s = re.search(r'sclk clock level: \d+: \((\d+)Mhz\)|sclk \((\d+)Mhz\)', out)
    return (float(p.group(1)) if p else float('nan'), float(next(g for g in s.groups() if g)) if s else float('nan'))


def run(name, fn, work=None):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    samples, stop = [], [False]

    def sampler():
        time.sleep(0.7)                       # let the power controller settle
        while not stop[0]:
            samples.append(smi())
    th = threading.Thread(target=sampler)
    th.start()
    n, t0 = 0, time.time()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    while time.time() - t0 < secs:
        for _ in range(20):
            fn()
        n += 20
        torch.cuda.synchronize()
    e1.record()
    torch.cuda.synchronize()
    stop[0] = True
    th.join()
    us = e0.elapsed_time(e1) / n * 1e3
    pw = [s[0] for s in samples if s[0] == s[0]]
    ck = [s[1] for s in samples if s[1] == s[1]]
    P = sum(pw) / max(len(pw), 1)
    C = sum(ck) / max(len(ck), 1)
    extra = ''
    if work:
        extra = '  ' + '  '.join('%s=%.3g' % kv for kv in work.items())
    print('%-34s %8.1f us  %6.0f W  %5.0f MHz  %.4f J/launch  (%d samples)%s' % (name, us, P, C, P * us * 1e-6, len(pw), extra), flush=True)
    return us, P


torch.manual_seed(0)
flat = (torch.rand(ops.mlp_param_count(W, IN), device=dev) - 0.5) * 0.2
wf, wb = ops.pack_weights(W, IN, flat, want_bwd=True)
enc = (torch.randn(rows * 64, device=dev) * 0.5).to(torch.bfloat16)
view = (torch.randn(B * 32, device=dev) * 0.5).to(torch.bfloat16)
raw = torch.empty(rows, 4, device=dev)
stash = torch.empty(ops.mlp_stash_bytes(W, rows), dtype=torch.uint8, device=dev)
mask = torch.empty(ops.mlp_mask_bytes(rows), dtype=torch.uint8, device=dev)
draw = torch.randn(rows, 4, device=dev) * 1e-3
print('variant:', os.environ.get('DURF_LIB_PATH', 'shipped'), ' idle: %.0f W %.0f MHz' % smi())
flops = 2 * 606208 * rows
run('fwd inference', lambda: ops.mlp_fwd(W, rows, N, enc, view, wf, raw=raw), dict(TFLOP=flops / 1e12))
run('fwd train', lambda: ops.mlp_fwd(W, rows, N, enc, view, wf, raw=raw, stash=stash, relu_mask=mask),
    dict(TFLOP=flops / 1e12, GB_written=(stash.numel() + mask.numel()) / 1e9))
dz, dz_out = ops.mlp_bwd(W, rows, N, draw, wb, mask)
run('bwd', lambda: ops.mlp_bwd(W, rows, N, draw, wb, mask), dict(TFLOP=flops / 1e12, GB_written=dz.numel() * 2 / 1e9))
if not os.environ.get('DURF_LIB_PATH'):
    view_tile = ops.expand_view(rows, N, view)
    part, bpart = ops.dw_buffers(W, dev)
    by = stash.numel() + dz.numel() * 2 + dz_out.numel() * 2 + enc.numel() * 2 + view_tile.numel() * 2
    run('dW (one level)', lambda: ops.mlp_dw(W, rows, N, [enc], [view_tile], [stash], [dz], [dz_out], part, bpart),
        dict(TFLOP=2 * 591872 * rows / 1e12, GB_read=by / 1e9))
    big = torch.empty(2400 * 1024 * 1024, dtype=torch.uint8, device=dev)
    big2 = torch.empty_like(big)
    run('memset 2.5 GB (torch fill_)', lambda: big.fill_(1), dict(GB_written=big.numel() / 1e9))
    run('copy 2.5 GB (torch copy_)', lambda: big2.copy_(big), dict(GB_read=big.numel() / 1e9, GB_written=big.numel() / 1e9))
    bf = big.view(torch.float32)
    run('read 2.5 GB (torch sum)', lambda: bf.sum(), dict(GB_read=big.numel() / 1e9))
    a = torch.randn(8192, 8192, device=dev).to(torch.bfloat16)
    b = torch.randn(8192, 8192, device=dev).to(torch.bfloat16)
    c = torch.empty(8192, 8192, device=dev, dtype=torch.bfloat16)
    run('vendor GEMM 8192^3 bf16', lambda: torch.matmul(a, b, out=c), dict(TFLOP=2 * 8192 ** 3 / 1e12))
