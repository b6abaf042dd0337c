"""Which training steps make the caching allocator go to the driver (hipMalloc / hipFree), and what do they cost?
python tools/alloc_probe.py [cfg]"""
import gc, os, sys, time
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
keys = ('num_device_alloc', 'num_device_free', 'num_alloc_retries')
last = None
gc_before = gc.get_count()
for i in range(40):
    torch.cuda.synchronize()
    a = time.perf_counter()
    state, stats, rng, _ = train_boxpose.train_step(model, config, rng, state, batch, 5e-4, 3.0, alpha, prev, reduce_stats=False)
    host = time.perf_counter() - a
    torch.cuda.synchronize()
    tot = time.perf_counter() - a
    ms = torch.cuda.memory_stats()
    cur = tuple(ms.get(k, 0) for k in keys)
    flag = '' if cur == last else '   <-- allocator went to the driver: %s' % dict(zip(keys, cur))
    print('step %2d  host %.2f ms  total %.2f ms  reserved %.2f GB  gc %s%s' % (
        i, host * 1e3, tot * 1e3, ms['reserved_bytes.all.current'] / 1e9, gc.get_count(), flag))
    last = cur
