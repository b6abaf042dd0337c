#!/usr/bin/env python3
"""Generates tests/golden/psnr_trajectory_N128.npz: the fp32 CPU oracle's training trajectory on the procedural
scene of tests/scripts/psnr_experiment.py at the metric's 128 samples/ray x 2 levels, with the batches it consumed, so that
`pytest -m gpu` (tests/test_gpu_psnr.py) can train the HIP path on the SAME batches from the SAME parameters and
assert |delta PSNR| <= 0.1 dB (BASELINE.json north_star) without running the oracle on the GPU box.

    python tests/golden/make_psnr_trajectory.py        (~20 min on 6 of 8 cores; run in the build container)

Contents: knobs; 8 training batches x 256 rays and 1024 held-out rays (origins, directions, viewdirs, radii, pixels,
depth, sky; box pose / extent / timestep); the oracle's held-out PSNR and train-batch PSNR at the evaluation steps."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from durf_amd import math as dmath, obbpose_model, utils  # noqa: E402
from oracle import durf_ref as R  # noqa: E402
from tests import helpers as H  # noqa: E402
from tests.scripts import psnr_experiment as P  # noqa: E402

N, STEPS, BATCH, NBATCH, NTEST, SEED = 128, 240, 256, 8, 1024, 7
EVAL_AT = (40, 80, 120, 160, 200, 240)
OUT = os.path.join(ROOT, 'tests', 'golden', 'psnr_trajectory_N128.npz')
RAY_KEYS = ('origins', 'directions', 'viewdirs', 'radii')


def gin_text():
    return ('MipNerfModel.num_samples = %d\nMipNerfModel.density_noise = 0.0\n'
            'MipNerfModel.no_pose_opt = True\nMipNerfModel.no_yaw_opt = True\n'
            'Config.randomized = False\nConfig.rand_bkgd = False\nConfig.grad_max_norm = 1.0\n'
            'Config.grad_max_val = 0.1\nConfig.tv_loss_mult = 0.0\nConfig.lr_init = 5e-4\nConfig.lr_final = 5e-6\n'
            'Config.lr_delay_steps = 50\nConfig.max_steps = %d\nConfig.eps_init = 3.0\nConfig.eps_final = 0.2\n'
            'Config.eps_max_steps = %d\n' % (N, STEPS, STEPS))


def schedule(config, s):
    return (dmath.learning_rate_decay(s, config.lr_init, config.lr_final, config.max_steps, config.lr_delay_steps,
                                      config.lr_delay_mult),
            dmath.learning_rate_decay(s, config.eps_init, config.eps_final, config.eps_max_steps, 0, config.lr_delay_mult))


def pack(batches):
    out = {k: np.stack([b['rays'][k] for b in batches]) for k in RAY_KEYS}
    for k in ('pixels', 'depth', 'sky'):
        out[k] = np.stack([b[k] for b in batches])
    return out


def unpack(d, i, far):
    """batch i of a packed set -> the numpy batch schema of durf_amd.synthetic.make_batch"""
    n = d['origins'].shape[1]
    rays = {k: d[k][i] for k in RAY_KEYS}
    rays.update(lossmult=np.ones((n, 1), np.float32), near=np.zeros((n, 1), np.float32), far=np.full((n, 1), far, np.float32))
    return dict(rays=rays, pixels=d['pixels'][i], depth=d['depth'][i], sky=d['sky'][i])


def main():
    P.N_SAMPLES = N
    torch.set_num_threads(int(os.environ.get('DURF_ORACLE_THREADS', os.cpu_count() or 1)))
    batches = [P.make_rays(BATCH, 1000 + i) for i in range(NBATCH)]
    test = P.make_rays(NTEST, 9999)
    utils.clear_gin()
    utils.parse_gin(gin_text())
    config = utils.configured(utils.Config)
    cb0 = {k: (torch.tensor(v) if isinstance(v, np.ndarray) else v) for k, v in batches[0].items() if k != 'rays'}
    model, var_cpu = obbpose_model.construct_mipnerf(SEED, cb0, device='cpu')
    params = H.oracle_params_from_variables(var_cpu)
    st = R.new_opt_state(params)
    obs = [H.oracle_batch(b) for b in batches]
    tb = H.oracle_batch(test)
    gt = torch.tensor(test['pixels'])
    ocfg = dict(R.CONFIG_DEFAULTS, randomized=False, tv_loss_mult=0.0)
    mcfg = dict(num_samples=N)
    test_psnr, train_psnr = [], []
    t0 = time.time()
    for step in range(1, STEPS + 1):
        lr, eps = schedule(config, step)
        ob = obs[step % NBATCH]
        params, st, ostats, _ = R.train_step(params, st, ob, ocfg, mcfg, lr, eps, 10.0, ob['init'][0:1])
        if step in EVAL_AT:
            with torch.no_grad():
                ret = R.model_apply(params, tb['rays'], test['ts'], tb['ext'], False, False, False, 10.0, cfg=mcfg)
            test_psnr.append(float(R.mse_to_psnr(((ret[-1][0] - gt) ** 2).mean())))
            train_psnr.append(float(ostats['psnr']))
            print('step %d: oracle test PSNR %.4f dB, train %.3f dB  (%.0f s)' % (step, test_psnr[-1], train_psnr[-1],
                                                                                 time.time() - t0), flush=True)
    tr = pack(batches)
    te = pack([test])
    np.savez_compressed(
        OUT, num_samples=N, steps=STEPS, nbatch=NBATCH, seed=SEED, far=P.FAR, eval_at=np.array(EVAL_AT), gin=gin_text(),
        oracle_test_psnr=np.array(test_psnr), oracle_train_psnr=np.array(train_psnr),
        init=batches[0]['init'], ext=batches[0]['ext'], target=batches[0]['target'], ts=batches[0]['ts'],
        **{'train_' + k: v for k, v in tr.items()}, **{'test_' + k: v for k, v in te.items()})
    print('wrote %s (%.0f KB) in %.0f s' % (OUT, os.path.getsize(OUT) / 1024, time.time() - t0))


if __name__ == '__main__':
    main()
