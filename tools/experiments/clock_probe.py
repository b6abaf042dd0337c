"""Is the fused MLP power/clock-limited?  Times the training forward / backward on random and on all-zero operands, and
samples rocm-smi clocks and power while a kernel loops (tools/clock_probe.py, run on the GPU box)."""
import os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from durf_amd import ops
dev = torch.device('cuda:0')
rows, N, W, IN = 4096 * 128, 128, 256, 60
raw = torch.empty(rows, 4, device=dev)
stash = torch.empty(ops.mlp_stash_bytes(W, rows), dtype=torch.uint8, device=dev)
mask = torch.empty(ops.mlp_mask_bytes(rows), dtype=torch.uint8, device=dev)


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


def smi():
    out = subprocess.run(['rocm-smi', '--showclocks', '--showpower', '--showtemp'], capture_output=True, text=True).stdout
    keep = [l.strip() for l in out.splitlines() if any(k in l for k in ('sclk', 'mclk', 'fclk', 'Power', 'junction', 'Junction'))]
    return ' | '.join(keep)


for name, scale in (('random', 0.2), ('zero', 0.0)):
    torch.manual_seed(0)
    flat = (torch.rand(ops.mlp_param_count(W, IN), device=dev) - 0.5) * scale
    wf, wb = ops.pack_weights(W, IN, flat, want_bwd=True)
    enc = (torch.randn(rows * 64, device=dev) * 0.5 * (scale > 0)).to(torch.bfloat16)
    view = (torch.randn(4096 * 32, device=dev) * 0.5 * (scale > 0)).to(torch.bfloat16)
    draw = torch.randn(rows, 4, device=dev) * 1e-3 * (scale > 0)
    ti = timeit(lambda: ops.mlp_fwd(W, rows, N, enc, view, wf, raw=raw))
    tt = timeit(lambda: ops.mlp_fwd(W, rows, N, enc, view, wf, raw=raw, stash=stash, relu_mask=mask))
    tb = timeit(lambda: ops.mlp_bwd(W, rows, N, draw, wb, mask))
    print('%-7s fwd inference %7.1f us   fwd train %7.1f us   bwd %7.1f us' % (name, ti, tt, tb), flush=True)
    samples = []
    stop = False

    def sampler():
        while not stop:
            samples.append(smi())
            time.sleep(0.5)
    th = threading.Thread(target=sampler)
    th.start()
    t0 = time.time()
    while time.time() - t0 < 4.0:
        for _ in range(50):
            ops.mlp_fwd(W, rows, N, enc, view, wf, raw=raw)
        torch.cuda.synchronize()
    stop = True
    th.join()
    for s in samples[1:6]:
        print('   ', s, flush=True)
print('idle:', smi())

# calibration: what does the vendor GEMM sustain on this board under the same power cap?  (torch.matmul -> hipBLASLt / rocBLAS)
for name, scale in (('random', 1.0), ('zero', 0.0)):
    for (m, n, k) in ((8192, 8192, 8192), (524288, 256, 256)):
        a = (torch.randn(m, k, device=dev) * scale).to(torch.bfloat16)
        b = (torch.randn(k, n, device=dev) * scale).to(torch.bfloat16)
        c = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
        t = timeit(lambda: torch.matmul(a, b, out=c), n=30)
        print('%-7s torch.matmul bf16 %dx%dx%d  %8.1f us  %7.1f TFLOP/s' % (name, m, n, k, t, 2.0 * m * n * k / t / 1e6), flush=True)
    a = (torch.randn(8192, 8192, device=dev) * scale).to(torch.bfloat16)
    b = (torch.randn(8192, 8192, device=dev) * scale).to(torch.bfloat16)
    c = torch.empty(8192, 8192, device=dev, dtype=torch.bfloat16)
    samples = []
    t0 = time.time()
    while time.time() - t0 < 3.0:
        for _ in range(50):
            torch.matmul(a, b, out=c)
        torch.cuda.synchronize()
        samples.append(smi())
    print('   ', samples[-1], flush=True)
