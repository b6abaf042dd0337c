"""Which objects of a training step sit in reference cycles (and so wait for the cyclic collector to give their tensors back)?"""
import collections, gc, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from durf_amd import train_boxpose

dev = torch.device('cuda', 0)
w = bench.setup_workload('cfg3', dev, rays=512)
model, config, state, batch, prev = (w[k] for k in ('model', 'config', 'state', 'batch', 'prev'))
rng = 0
step_fn = train_boxpose.best_step_fn(model, state.variables)
for i in range(5):
    state, stats, rng, _ = step_fn(model, config, rng, state, batch, 5e-4, 3.0, w['alpha'], prev, reduce_stats=False)
torch.cuda.synchronize()
gc.collect()
gc.set_debug(gc.DEBUG_SAVEALL)
m0 = torch.cuda.memory_allocated()
for i in range(20):
    state, stats, rng, _ = step_fn(model, config, rng, state, batch, 5e-4, 3.0, w['alpha'], prev, reduce_stats=(i % 5 == 0))
torch.cuda.synchronize()
m1 = torch.cuda.memory_allocated()
n = gc.collect()
print('unreachable objects found after 20 steps:', n, ' allocated before/after the steps: %.1f / %.1f MB' % (m0 / 1e6, m1 / 1e6))
cnt = collections.Counter(type(o).__name__ for o in gc.garbage)
print(cnt.most_common(15))
tens = [o for o in gc.garbage if isinstance(o, torch.Tensor)]
print('tensors in garbage:', len(tens), ' bytes: %.1f MB' % (sum(t.numel() * t.element_size() for t in tens) / 1e6))
for o in gc.garbage[:400]:
    if not isinstance(o, (torch.Tensor, dict, list, tuple, int, float, str)):
        print('  ', type(o), repr(o)[:120])
