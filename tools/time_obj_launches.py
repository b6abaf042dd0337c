"""object forward/backward batch latency at small and large batches: M-split vs sample-split"""
import os, sys
sys.path.insert(0, '/root/repo')
import torch
import bench
from durf_amd import ops, train_boxpose
dev = torch.device('cuda:0')
for cfg, rays in (('cfg3', 512), ('cfg5', 1024), ('cfg3', 4096)):
    for ms in ('1', '0'):
        os.environ['DURF_OBJ_MSPLIT'] = ms
        wl = bench.setup_workload(cfg, dev, rays=rays)
        st, rng = wl['state'], 0
        ops.TIMED_NAMES = {'obj_fwd_batch', 'obj_bwd_batch'}
        for i in range(6):
            st, stats, rng, _ = train_boxpose.train_step(wl['model'], wl['config'], rng, st, wl['batch'], 5e-4, 3.0, wl['alpha'], wl['prev'])
        torch.cuda.synchronize()
        ops.TIMERS = {}
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(50):
            st, stats, rng, _ = train_boxpose.train_step(wl['model'], wl['config'], rng, st, wl['batch'], 5e-4, 3.0, wl['alpha'], wl['prev'])
        e1.record()
        torch.cuda.synchronize()
        tot = ops.timer_totals()
        ops.TIMERS = None
        print('%s %5d rays  msplit=%s  step %.3f ms  %s  loss %.6f' % (cfg, rays, ms, e0.elapsed_time(e1) / 50,
              {k: round(v[1] / v[0] * 1e6, 1) for k, v in tot.items()}, float(stats.loss)))
