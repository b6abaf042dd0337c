"""HIP-event timing of the background encode (and, with 'composite', the composite / resample kernels) on the
bench batch: python tools/time_encode.py [rays] [launches]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from durf_amd import ops, synthetic

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
n = int(sys.argv[2]) if len(sys.argv) > 2 else 50
dev = torch.device('cuda:0')
b = synthetic.make_batch(B, 3, far=40.0, seed=synthetic.SEED)
db = synthetic.device_batch(b, dev)
rays = db['rays']
pose = db['init'][b['ts']].contiguous()
o_s, d_s, hit, zo = ops.ray_setup(rays.origins, rays.directions, pose, db['ext'])
t_vals = ops.sample_t(rays.near.reshape(-1), rays.far.reshape(-1), 128)
radii = rays.radii.reshape(-1).contiguous()
for _ in range(5):
    ops.encode_bkgd(t_vals, o_s, d_s, radii, hit, True)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
ev[0].record()
for i in range(n):
    ops.encode_bkgd(t_vals, o_s, d_s, radii, hit, True)
    ev[i + 1].record()
torch.cuda.synchronize()
ts = sorted(ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(n))
byt = 15928.0 * B
print('encode_bkgd %d rays: median %.2f us  min %.2f us  -> %.2f TB/s = %.1f %% of 8 TB/s (median)' % (
    B, ts[n // 2], ts[0], byt / ts[n // 2] / 1e6, byt / ts[n // 2] / 1e6 / 8 * 100))
