"""Lists the PyTorch (non-durf) device kernels one train_step launches and which line of durf_amd/
launched them -- the glue the fused kernels have not absorbed yet."""
import os
import sys
from collections import Counter

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch.profiler import ProfilerActivity, profile

import bench
from durf_amd import train_boxpose

dev = torch.device('cuda:0')
w = bench.setup_workload(os.environ.get('DURF_BENCH_CONFIG', 'cfg3'), dev)
config, model, state, batch, prev = w['config'], w['model'], w['state'], w['batch'], w['prev']
rng = 0
for _ in range(3):
    state, stats, rng, _ = train_boxpose.train_step(model, config, rng, state, batch, 5e-4, 3.0, 10.0, prev)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    state, stats, rng, _ = train_boxpose.train_step(model, config, rng, state, batch, 5e-4, 3.0, 10.0, prev)
    torch.cuda.synchronize()
by_line = Counter()
NO_KERNEL = ('aten::empty', 'aten::view', 'aten::reshape', 'aten::select', 'aten::slice', 'aten::as_strided', 'aten::expand',
             'aten::_unsafe_view', 'aten::contiguous', 'aten::to', 'aten::squeeze', 'aten::unsqueeze', 'aten::item',
             'aten::_local_scalar_dense', 'aten::empty_like', 'aten::empty_strided', 'aten::detach', 'aten::alias', 'aten::t',
             'aten::transpose', 'aten::permute', 'aten::resize_', 'aten::lift_fresh', 'aten::result_type', 'aten::unbind', 'aten::narrow')
for ev in prof.events():
    if ev.device_type != torch.autograd.DeviceType.CPU or not ev.name.startswith('aten::'):
        continue
    if ev.cpu_parent is not None and ev.cpu_parent.name.startswith('aten::'):
        continue                      # count top-level ops only
    if ev.name in NO_KERNEL:
        continue
    frame = next((f for f in (ev.stack or []) if 'durf_amd/' in f), '?')
    by_line[(frame.split('durf_amd/')[-1].strip(), ev.name)] += 1
print('top-level aten ops that (probably) launch a kernel: %d' % sum(by_line.values()))
for (k, name), n in sorted(by_line.items(), key=lambda kv: kv[0]):
    print('%3d  %-22s %s' % (n, name, k))
