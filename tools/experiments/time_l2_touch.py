"""The M-split object kernels with and without their weight-stream warm-up (l2_touch), timed in the step:
DURF_L2_TOUCH = 1 / 0, object forward / backward launch durations (HIP events) and the step."""
import os, sys
sys.path.insert(0, '/root/repo')
import torch
import bench
from durf_amd import ops, train_boxpose
dev = torch.device('cuda:0')
for rep in range(2):
    for cfg, rays in (('cfg3', 512), ('cfg5', 1024), ('cfg3', 1024)):
        for touch in ('1', '0'):
            os.environ['DURF_L2_TOUCH'] = touch
            wl = bench.setup_workload(cfg, dev, rays=rays)
            st, rng = wl['state'], 0
            run = lambda st, rng: train_boxpose.train_step(wl['model'], wl['config'], rng, st, wl['batch'], 5e-4, 3.0, wl['alpha'], wl['prev'])
            res = {}
            for timed in (True, False):          # the launch durations with the events on, the step without them
                ops.TIMED_NAMES = {'obj_fwd_batch', 'obj_bwd_batch'}
                ops.TIMERS = {} if timed else None
                for i in range(6):
                    st, stats, rng, _ = run(st, rng)
                torch.cuda.synchronize()
                if timed:
                    ops.TIMERS = {}
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for i in range(100):
                    st, stats, rng, _ = run(st, rng)
                e1.record()
                torch.cuda.synchronize()
                if timed:
                    res['launch_us'] = {k: round(v[1] / v[0] * 1e6, 1) for k, v in ops.timer_totals().items()}
                    ops.TIMERS = None
                else:
                    res['step_ms'] = e0.elapsed_time(e1) / 100
            print('%s %5d rays  touch=%s  step %.4f ms  %s  loss %.6f' % (cfg, rays, touch, res['step_ms'], res['launch_us'], float(stats.loss)))
