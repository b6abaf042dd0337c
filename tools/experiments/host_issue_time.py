"""How long does the host need to ENQUEUE one training step (ctypes launches + torch allocations), against the GPU's step
time?  If the two are close, the short per-ray kernels run host-bound.   python tools/host_issue_time.py [cfg]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from durf_amd import train_boxpose

cfg = sys.argv[1] if len(sys.argv) > 1 else 'cfg3'
dev = torch.device('cuda:0')
w = bench.setup_workload(cfg, dev)
config, model, state, batch, prev = (w[k] for k in ('config', 'model', 'state', 'batch', 'prev'))
alpha = w['alpha']
rng = 0
for i in range(5):
    state, stats, rng, _ = train_boxpose.train_step(model, config, rng, state, batch, 5e-4, 3.0, alpha, prev, reduce_stats=False)
torch.cuda.synchronize()
n = 100
issue = []
t0 = time.perf_counter()
for i in range(n):
    a = time.perf_counter()
    state, stats, rng, _ = train_boxpose.train_step(model, config, rng, state, batch, 5e-4, 3.0, alpha, prev, reduce_stats=False)
    issue.append(time.perf_counter() - a)
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
issue.sort()
print('%s: host enqueue per step: median %.3f ms, p10 %.3f, p90 %.3f; loop without sync %.3f ms/step; with sync %.3f ms/step'
      % (cfg, issue[n // 2] * 1e3, issue[n // 10] * 1e3, issue[9 * n // 10] * 1e3, t_issue / n * 1e3, t_all / n * 1e3))
# the same with the GPU drained before every step: pure host cost of a step
torch.cuda.synchronize()
pure = []
for i in range(30):
    torch.cuda.synchronize()
    a = time.perf_counter()
    state, stats, rng, _ = train_boxpose.train_step(model, config, rng, state, batch, 5e-4, 3.0, alpha, prev, reduce_stats=False)
    pure.append(time.perf_counter() - a)
pure.sort()
print('%s: host enqueue with an idle GPU: median %.3f ms, min %.3f' % (cfg, pure[15] * 1e3, pure[0] * 1e3))
