"""The fused per-ray training launch (durf_composite_resample: merge + activations + volumetric rendering + resampling + the
loss normalisers of both levels; obbpose_model.py:232-245, mip.py:285-327,373-416, train_boxpose.py:94-102) alone, at the bench's
4096 rays -- where it is launch-latency-bound -- and at 32 768 rays, where the HBM target can be judged:
    python3 tools/time_composite.py [rays ...]         (under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE for the byte counts)
Bytes per ray (N = 128): reads raw 2048 + t_vals 516 + d 12 + per-ray inputs 24; writes weights 512 + t_mids 512 + t_dists 512
+ next t_vals 516 + rgb / depth / acc 20 + 2 x 5 normaliser rows 40 = 5212 B (SURVEY 8d counted 3624 B: composite 3108 +
next t_vals 516, without the t_mids / t_dists the reference returns and without the loss normalisers)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from durf_amd import ops
dev = torch.device('cuda:0')
N, K = 128, 0
for B in [int(a) for a in sys.argv[1:]] or [4096, 32768]:
    torch.manual_seed(0)
    raw = torch.randn(B * N, 4, device=dev)
    t = torch.sort(torch.rand(B, N + 1, device=dev) * 40.0, dim=1).values
    d = torch.randn(B, 3, device=dev)
    u = torch.rand(B, N + 1, device=dev)
    slot = torch.full((B, 1), -1, dtype=torch.int32, device=dev)
    prep = dict(lossmult=torch.ones(B, device=dev), gt_depth=torch.rand(B, device=dev) * 30, sky=torch.zeros(B, device=dev),
                dyn=torch.zeros(B, dtype=torch.int32, device=dev), zo=torch.zeros(B, device=dev), eps=3.0, box_loss_mult=0.0, level=0,
                disable_multiscale=False, norms=torch.empty(2, ops.PREP_ROWS, device=dev))
    fn = lambda: ops.composite_resample(raw, [], slot, t, d, -1.0, ops.BKGD_GREY, 0.01, u, prep=prep)
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3           # includes the k_reduce_rows launch behind it
    print('%6d rays: composite_resample + reduce_rows %7.1f us per call;  3624 B/ray (SURVEY 8d) -> %.2f TB/s = %.3f of 8 TB/s;  '
          '5212 B/ray (all it reads and writes) -> %.2f TB/s = %.3f' % (B, us, B * 3624 / us / 1e6, B * 3624 / us / 1e6 / 8,
                                                                        B * 5212 / us / 1e6, B * 5212 / us / 1e6 / 8))
