"""Per-step GPU timeline of the bench workload: one HIP event per train_step, read after the final sync,
so the distribution of step times (and of the host-side gaps between steps) is visible without adding
synchronisation inside the timed region.  Used to look into the occasional 15-20 % slow bench run.
    python tools/step_times.py [steps] [warmup]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

import bench
from durf_amd import train_boxpose

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
warmup = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device('cuda:0')
w = bench.setup_workload(os.environ.get('DURF_BENCH_CONFIG', 'cfg3'), dev)
config, model, state, batch, prev, alpha = w['config'], w['model'], w['state'], w['batch'], w['prev'], w['alpha']
rng = 0
host = []
ev = [torch.cuda.Event(enable_timing=True) for _ in range(warmup + steps + 1)]
torch.cuda.synchronize()
ev[0].record()
t_prev = time.perf_counter()
for i in range(warmup + steps):
    state, stats, rng, _ = train_boxpose.train_step(model, config, rng, state, batch, 5e-4, 3.0, alpha, prev)
    ev[i + 1].record()
    t = time.perf_counter(); host.append((t - t_prev) * 1e3); t_prev = t        # host time to ISSUE the step
torch.cuda.synchronize()
gpu = np.array([ev[i].elapsed_time(ev[i + 1]) for i in range(warmup + steps)])
host = np.array(host)
print('step   gpu_ms  host_issue_ms')
for i in range(min(warmup + steps, 12)):
    print('%4d  %7.3f  %7.3f%s' % (i, gpu[i], host[i], '   (warm-up)' if i < warmup else ''))
g = gpu[warmup:]
print('timed steps: mean %.3f  median %.3f  min %.3f  max %.3f  p90 %.3f  -> %.0f rays/s (mean)' % (
    g.mean(), np.median(g), g.min(), g.max(), np.percentile(g, 90), w['B'] / g.mean() * 1e3))
print('host issue time per step: mean %.3f  max %.3f ms (GPU-bound while this stays below the GPU step time)' % (
    host[warmup:].mean(), host[warmup:].max()))
slow = np.nonzero(g > 1.1 * np.median(g))[0]
print('steps >10%% over the median: %d of %d %s' % (len(slow), len(g), (slow + warmup).tolist()[:20]))
