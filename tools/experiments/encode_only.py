"""Launches only the background encode kernel on the bench batch (for rocprofv3 --pmc passes, which serialise
every launch: keep the process tiny).   python tools/encode_only.py [rays] [launches]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import bench
from durf_amd import ops, synthetic

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = torch.device('cuda:0')
b = synthetic.make_batch(B, 1, far=40.0, seed=synthetic.SEED)
db = synthetic.device_batch(b, dev)
rays = db['rays']
pose = db['init'][b['ts']].contiguous()
o_s, d_s, hit, zo = ops.ray_setup(rays.origins, rays.directions, pose, db['ext'])
t_vals = ops.sample_t(rays.near.reshape(-1), rays.far.reshape(-1), 128)
radii = rays.radii.reshape(-1).contiguous()
for _ in range(n):
    ops.encode_bkgd(t_vals, o_s, d_s, radii, hit, True)
torch.cuda.synchronize()
