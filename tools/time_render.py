"""Throughput of the inference path: render_image on a 640 x 960 synthetic test case (cfg3 model, K = 3), rays/s per
chunk size.   python tools/time_render.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from durf_amd import obbpose_model as om, train_boxpose, synthetic, utils

dev = torch.device('cuda:0')
w = bench.setup_workload('cfg3', dev)
config, model, state = w['config'], w['model'], w['state']
H, W = 640, 960
b = synthetic.make_batch(H * W, w['K'], seed=7, far=w['far'], allow_multi_hit=True)
db = synthetic.device_batch(b, dev)
rays = utils.namedtuple_map(lambda r: r.reshape(H, W, -1), db['rays'])
fn = train_boxpose.make_render_fn(model, config, state.variables)
for chunk in (4096, 8192, 32768, 131072):
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        rgb, dist, acc = om.render_image(fn, rays, db['init'], db['ext'], b['ts'], 0, 10.0, chunk=chunk)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print('chunk %6d: %.1f ms per %dx%d image = %.2f M rays/s' % (chunk, dt * 1e3, H, W, H * W / dt / 1e6), flush=True)
