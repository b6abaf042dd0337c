#!/usr/bin/env python3
"""Generates tests/golden/pose_trajectory_N128.npz: the fp32 CPU oracle's BOX-POSE trajectory with pose optimisation on
(cfg4's path: no_pose_opt = no_yaw_opt = False), on the batches of psnr_trajectory_N128.npz, starting from a box pose
perturbed by PERTURB.  tests/test_gpu_pose_trajectory.py trains the HIP path from the same parameters on the same
batches and compares the pose it arrives at -- the end-to-end check of the box-pose gradient path (object MLP d(enc) ->
durf_encode_obj_bwd_batch -> durf_pose_finish -> Adam) that single-step gradient comparisons cannot give, because the
bf16 gradient noise of that path is large per step.

    python tests/golden/make_pose_trajectory.py        (~6 min on 8 cores; run in the build container)"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
from durf_amd import obbpose_model, utils  # noqa: E402
from oracle import durf_ref as R  # noqa: E402
from tests import helpers as H  # noqa: E402
import make_psnr_trajectory as M  # noqa: E402

STEPS, EVERY = 160, 10
PERTURB = np.array([0.10, 0.0, 0.08, 0.0, 0.05, 0.0], np.float32)      # position (x, y, z), rotation vector
SRC = os.path.join(ROOT, 'tests', 'golden', 'psnr_trajectory_N128.npz')
OUT = os.path.join(ROOT, 'tests', 'golden', 'pose_trajectory_N128.npz')


def gin_text(n):
    return M.gin_text().replace('MipNerfModel.no_pose_opt = True\nMipNerfModel.no_yaw_opt = True\n',
                                'MipNerfModel.no_pose_opt = False\nMipNerfModel.no_yaw_opt = False\n') \
        .replace('Config.max_steps = %d' % M.STEPS, 'Config.max_steps = 4000') \
        .replace('Config.lr_delay_steps = 50', 'Config.lr_delay_steps = 20') \
        .replace('Config.eps_max_steps = %d' % M.STEPS, 'Config.eps_max_steps = %d' % STEPS)      # lr stays ~5e-4: the pose can travel


def perturbed_init(z):
    init = z['init'].copy()
    init[int(z['ts']), 0] += PERTURB
    return init


def main():
    torch.set_num_threads(int(os.environ.get('DURF_ORACLE_THREADS', os.cpu_count() or 1)))
    z = np.load(SRC, allow_pickle=False)
    N, nbatch, far, ts = int(z['num_samples']), int(z['nbatch']), float(z['far']), int(z['ts'])
    init = perturbed_init(z)
    common = dict(init=init, ext=z['ext'], target=z['target'], ts=ts)
    tr = {k[6:]: z[k] for k in z.files if k.startswith('train_')}
    utils.clear_gin()
    utils.parse_gin(gin_text(N))
    config = utils.configured(utils.Config)
    model, var_cpu = obbpose_model.construct_mipnerf(int(z['seed']), {k: torch.tensor(v) if isinstance(v, np.ndarray) else v
                                                                       for k, v in common.items()}, device='cpu')
    params = H.oracle_params_from_variables(var_cpu)
    st = R.new_opt_state(params)
    obs = [H.oracle_batch(dict(M.unpack(tr, i, far), **common)) for i in range(nbatch)]
    ocfg = dict(R.CONFIG_DEFAULTS, randomized=False, tv_loss_mult=0.0)
    mcfg = dict(num_samples=N, no_pose_opt=False, no_yaw_opt=False)
    traj, losses = [params['box_centers'][ts, 0].detach().clone().numpy()], []
    t0 = time.time()
    for step in range(1, STEPS + 1):
        lr, eps = M.schedule(config, step)
        ob = obs[step % nbatch]
        params, st, ostats, _ = R.train_step(params, st, ob, ocfg, mcfg, lr, eps, 10.0, ob['init'][0:1])
        losses.append(float(ostats['loss']))
        if step % EVERY == 0:
            traj.append(params['box_centers'][ts, 0].detach().clone().numpy())
            print('step %d: loss %.5f pose %s  (%.0f s)' % (step, losses[-1], np.round(traj[-1], 4), time.time() - t0), flush=True)
    np.savez_compressed(OUT, steps=STEPS, every=EVERY, perturb=PERTURB, gin=gin_text(N), init=init,
                        oracle_pose=np.stack(traj), oracle_loss=np.array(losses, np.float32), true_pose=np.asarray(z['target']).reshape(-1, 6)[0])
    print('wrote %s in %.0f s' % (OUT, time.time() - t0))


if __name__ == '__main__':
    main()
