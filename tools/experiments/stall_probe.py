"""Where does the one-off ~40 ms step of a fresh process come from?  Times the first 16 steps (GPU events) after different
pre-treatments:  python tools/stall_probe.py none|launches|events|both|sync"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from durf_amd import train_boxpose, ops

mode = sys.argv[1] if len(sys.argv) > 1 else 'none'
dev = torch.device('cuda:0')
w = bench.setup_workload('cfg3', dev)
config, model, state, batch, prev = (w[k] for k in ('config', 'model', 'state', 'batch', 'prev'))
alpha = w['alpha']
x = torch.zeros(1024, device=dev)
if mode in ('launches', 'both'):
    for _ in range(600):
        x.add_(1.0)
if mode in ('events', 'both'):
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(600)]
    for e in evs:
        e.record()
torch.cuda.synchronize()
n = 16
marks = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
rng = 0
host = []
marks[0].record()
for i in range(n):
    a = time.perf_counter()
    state, stats, rng, _ = train_boxpose.train_step(model, config, rng, state, batch, 5e-4, 3.0, alpha, prev, reduce_stats=False)
    host.append((time.perf_counter() - a) * 1e3)
    marks[i + 1].record()
    if mode == 'sync':
        torch.cuda.synchronize()
torch.cuda.synchronize()
print(mode, 'gpu ms:', ' '.join('%.1f' % marks[i].elapsed_time(marks[i + 1]) for i in range(n)))
print(mode, 'host ms:', ' '.join('%.1f' % h for h in host))
