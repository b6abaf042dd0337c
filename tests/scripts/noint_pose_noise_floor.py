#!/usr/bin/env python3
"""How far the fp32 ORACLE is from the float64 oracle on the box-pose gradient (CPU only, ~1 min):
    python tests/scripts/noint_pose_noise_floor.py > profiles/r05_noint_pose_noise_floor.txt
With disable_integration (obbpose_model.py:163-164) and the full BARF window (alpha = 10) every frequency of the object encoding
is on and none is damped: sin(2^9 x) enters at full weight, and the pose gradient -- a sum over the box-hit rays that cancels
to ~1 % of its summed magnitudes -- is then limited by fp32 itself.  This is the reason tests/test_gpu_train.py holds that one
case of test_box_pose_gradients to the float64 oracle at 0.15 instead of the fp32 oracle at 5e-2."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from durf_amd import obbpose_model, synthetic, utils  # noqa: E402
from oracle import durf_ref as R  # noqa: E402
from tests import helpers as H  # noqa: E402

B, N, K, tv = 1024, 32, 2, 0.01
b = synthetic.make_batch(B, K, seed=77 + K, noise_boxes=0.05)           # the batch of test_box_pose_gradients (K = 2)
utils.clear_gin()
cb = {k: (torch.tensor(v) if isinstance(v, np.ndarray) else v) for k, v in b.items() if k != 'rays'}
model, variables = obbpose_model.construct_mipnerf(5, cb, device='cpu')
print('# d(loss)/d(box_centers[ts]) of the oracle in float32 against the oracle in float64, norm-wise relative difference')
print('# batch: synthetic.make_batch(1024, 2, seed=79, noise_boxes=0.05), 32 samples/ray, tv_loss_mult 0.01')
for alpha in (10.0, 5.5):
    for knobs in ({}, dict(ray_shape='cylinder'), dict(disable_integration=True)):
        res = {}
        for dt in (torch.float32, torch.float64):
            ob = H.oracle_batch(b, dt)
            params = H.oracle_params_from_variables(variables, dt)
            ocfg = dict(R.CONFIG_DEFAULTS, randomized=False, tv_loss_mult=tv)
            mcfg = dict(num_samples=N, no_pose_opt=False, no_yaw_opt=False, **knobs)
            _, _, _, og = R.train_step(params, R.new_opt_state(params), ob, ocfg, mcfg, 5e-4, 3.0, alpha, ob['init'][0:1] + 0.01)
            res[dt] = og[0][b['ts']].double()
        a, c = res[torch.float32], res[torch.float64]
        for k in range(K):
            print('alpha %4.1f  %-28s object %d: position %.2e  rotation %.2e' % (
                alpha, knobs or 'cone, integrated', k, float((a[k, :3] - c[k, :3]).norm() / c[k, :3].norm()),
                float((a[k, 3:] - c[k, 3:]).norm() / c[k, 3:].norm())))
