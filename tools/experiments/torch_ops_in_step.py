"""Which torch (aten) operators launch kernels inside one training step?  python tools/torch_ops_in_step.py [cfg]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch.profiler import profile, ProfilerActivity
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
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    state, stats, rng, _ = train_boxpose.train_step(model, config, rng, state, batch, 5e-4, 3.0, alpha, prev, reduce_stats=False)
    torch.cuda.synchronize()
evs = [e for e in prof.events() if e.name.startswith('aten::') and e.device_time_total > 0]
seen = set()
for e in evs:
    if any(c.name.startswith('aten::') and c.device_time_total > 0 for c in (e.cpu_children or [])):
        continue                                   # report the leaf operator only
    st = [s for s in (e.stack or []) if 'durf_amd' in s]
    print('%-28s device %.1f us  %s' % (e.name, e.device_time_total, st[0].strip() if st else ''))
print(len(evs), 'aten events with device time')
