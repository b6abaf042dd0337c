#!/usr/bin/env python3
"""PSNR-parity experiment: no dataset exists here, so a procedural scene stands in.

A small analytic radiance field (coloured Gaussian blobs + a ground slab in the world frame, one
blob cluster riding inside a moving oriented box) is rendered to ground-truth pixels / LIDAR depth
/ sky flags by dense quadrature in float64.  Two trainers then start from IDENTICAL parameters and
consume IDENTICAL batches for S steps:
    (i)  the CPU oracle  (oracle/durf_ref.py, fp32 -- the reference's arithmetic), and
    (ii) the HIP build   (durf_amd, bf16 MFMA MLPs) on cuda:0,
and both are evaluated on the same held-out rays:  PSNR = mse_to_psnr(mean((pred - gt)^2))
(train_boxpose.py:562).  Target: |delta| <= 0.1 dB.

    python tests/psnr_experiment.py --steps 300 [--skip-cpu] [--out profiles/r01_psnr.json]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from durf_amd import synthetic  # noqa: E402

N_SAMPLES = 32
FAR = 8.0
BOX_C = np.array([0.3, 0.0, -3.0])
BOX_YAW = 0.6
BOX_EXT = np.array([0.5, 0.4, 0.9])


def _rot_y(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])


def field(x):
    """density [..], rgb [..,3] of the analytic scene at world points x [..,3] (float64)."""
    blobs = [((-0.8, 0.1, -4.0), 0.5, (0.9, 0.2, 0.2), 6.0), ((1.0, -0.2, -5.0), 0.7, (0.2, 0.8, 0.3), 4.0),
             ((0.0, 0.6, -6.0), 0.9, (0.2, 0.3, 0.9), 3.0)]
    sig = np.zeros(x.shape[:-1])
    col = np.zeros(x.shape)
    for c, r, rgb, amp in blobs:
        w = amp * np.exp(-0.5 * ((x - np.array(c)) ** 2).sum(-1) / r ** 2)
        sig += w
        col += w[..., None] * np.array(rgb)
    # ground slab
    g = 8.0 / (1.0 + np.exp((x[..., 1] + 0.9) * 12.0))
    sig += g
    col += g[..., None] * np.array([0.5, 0.45, 0.35])
    # the moving object: a blob in the box frame
    R = _rot_y(BOX_YAW)
    xo = (x - BOX_C) @ R.T
    inside = (np.abs(xo) <= BOX_EXT).all(-1)
    w = 10.0 * np.exp(-0.5 * ((xo / (BOX_EXT * 0.6)) ** 2).sum(-1)) * inside
    sig += w
    col += w[..., None] * np.array([0.9, 0.8, 0.1])
    col = col / np.maximum(sig, 1e-9)[..., None]
    return sig, col


def render_gt(o, d, far=FAR, n=768):
    t = np.linspace(0, far, n + 1)
    tm = 0.5 * (t[1:] + t[:-1])
    dn = np.linalg.norm(d, axis=-1, keepdims=True)
    x = o[:, None, :] + d[:, None, :] * tm[None, :, None]
    sig, col = field(x)
    a = sig * (t[1:] - t[:-1])[None, :] * dn
    T = np.exp(-np.concatenate([np.zeros((o.shape[0], 1)), np.cumsum(a[:, :-1], -1)], -1))
    w = (1 - np.exp(-a)) * T
    acc = w.sum(-1)
    rgb = (w[..., None] * col).sum(-2) + 0.5 * (1 - acc[:, None])
    depth = (w * tm[None]).sum(-1)
    return rgb, depth, acc


def make_rays(n, seed):
    b = synthetic.make_batch(n, 1, far=FAR, seed=seed)
    # one box at a known pose (overrides the generator's placement)
    rot = np.array([0.0, -BOX_YAW, 0.0])        # world->object rotation vector (R_y(-(-yaw)) = R_y(yaw))
    pose = np.concatenate([BOX_C, rot]).astype(np.float32)
    b['init'] = np.tile(pose[None, None], (5, 1, 1)).astype(np.float32)
    b['target'] = pose[None].astype(np.float32)
    b['ext'] = BOX_EXT[None].astype(np.float32)
    b['ts'] = 2
    o, d = b['rays']['origins'].astype(np.float64), b['rays']['directions'].astype(np.float64)
    rgb, depth, acc = render_gt(o, d)
    rs = np.random.default_rng(seed + 1)
    lidar = rs.uniform(0, 1, n) < 0.3
    b['pixels'] = rgb.astype(np.float32)
    b['depth'] = np.where(lidar & (acc > 0.9), depth, 0.0).astype(np.float32)[:, None]
    b['sky'] = np.where(acc < 0.05, 0.975, 0.0).astype(np.float32)[:, None]
    return b


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=300)
    ap.add_argument('--batch', type=int, default=512)
    ap.add_argument('--nbatches', type=int, default=16)
    ap.add_argument('--skip-cpu', action='store_true')
    ap.add_argument('--skip-gpu', action='store_true')
    ap.add_argument('--out', default=None)
    args = ap.parse_args()
    from durf_amd import math as dmath
    from durf_amd import obbpose_model, utils
    from oracle import durf_ref as R
    from tests import helpers as H

    batches = [make_rays(args.batch, 1000 + i) for i in range(args.nbatches)]
    test = make_rays(2048, 9999)
    gin = ('MipNerfModel.num_samples = %d\nMipNerfModel.density_noise = 0.0\n'
           'MipNerfModel.no_pose_opt = True\nMipNerfModel.no_yaw_opt = True\n'
           'Config.randomized = False\nConfig.rand_bkgd = False\nConfig.grad_max_norm = 1.0\n'
           'Config.grad_max_val = 0.1\nConfig.tv_loss_mult = 0.0\nConfig.lr_init = 5e-4\nConfig.lr_final = 5e-6\n'
           'Config.lr_delay_steps = 50\nConfig.max_steps = %d\nConfig.eps_init = 3.0\nConfig.eps_final = 0.2\n'
           'Config.eps_max_steps = %d\n' % (N_SAMPLES, args.steps, args.steps))
    utils.clear_gin()
    utils.parse_gin(gin)
    config = utils.configured(utils.Config)
    sched = lambda s: (dmath.learning_rate_decay(s, config.lr_init, config.lr_final, config.max_steps,
                                                 config.lr_delay_steps, config.lr_delay_mult),
                       dmath.learning_rate_decay(s, config.eps_init, config.eps_final, config.eps_max_steps, 0,
                                                 config.lr_delay_mult))
    cb0 = {k: (torch.tensor(v) if isinstance(v, np.ndarray) else v) for k, v in batches[0].items() if k != 'rays'}
    model, var_cpu = obbpose_model.construct_mipnerf(7, cb0, device='cpu')
    result = dict(steps=args.steps, batch=args.batch, num_samples=N_SAMPLES)
    gt = torch.tensor(test['pixels'])

    if not args.skip_gpu:
        from durf_amd import train_boxpose
        dev = torch.device('cuda:0')
        variables = var_cpu.like(var_cpu.flat.clone().to(dev))
        state = train_boxpose.create_train_state(variables)
        dbs = [H.device_batch(b, dev) for b in batches]
        t0 = time.time()
        for step in range(1, args.steps + 1):
            lr, eps = sched(step)
            db = dbs[step % len(dbs)]
            state, stats, _, _ = train_boxpose.train_step(model, config, 0, state, db, lr, eps, 10.0, db['init'][0:1])
        torch.cuda.synchronize()
        dt = time.time() - t0
        tb = H.device_batch(test, dev)
        ret = model.apply(state.variables, 0, tb['rays'], tb['init'], tb['ext'], test['ts'], randomized=False,
                          rand_bkgd=False, white_bkgd=False, alpha=10.0)
        pred = ret[-1][0].cpu()
        result['gpu_psnr'] = float(dmath.mse_to_psnr(((pred - gt) ** 2).mean()))
        result['gpu_train_psnr_last'] = float(stats.psnr)
        result['gpu_seconds'] = dt
        print('HIP   : test PSNR %.3f dB  (train %.1f s)' % (result['gpu_psnr'], dt), flush=True)

    if not args.skip_cpu:
        torch.set_num_threads(min(os.cpu_count() or 1, 16))
        variables = var_cpu.like(var_cpu.flat.clone())
        params = H.oracle_params_from_variables(variables)
        st = R.new_opt_state(params)
        obs = [H.oracle_batch(b) for b in batches]
        ocfg = dict(R.CONFIG_DEFAULTS, randomized=False, tv_loss_mult=0.0)
        mcfg = dict(num_samples=N_SAMPLES)
        t0 = time.time()
        for step in range(1, args.steps + 1):
            lr, eps = sched(step)
            ob = obs[step % len(obs)]
            params, st, ostats, _ = R.train_step(params, st, ob, ocfg, mcfg, lr, eps, 10.0, ob['init'][0:1])
        dt = time.time() - t0
        tb = H.oracle_batch(test)
        with torch.no_grad():
            ret = R.model_apply(params, tb['rays'], test['ts'], tb['ext'], False, False, False, 10.0, cfg=mcfg)
        pred = ret[-1][0]
        result['cpu_psnr'] = float(R.mse_to_psnr(((pred - gt) ** 2).mean()))
        result['cpu_seconds'] = dt
        print('oracle: test PSNR %.3f dB  (train %.1f s)' % (result['cpu_psnr'], dt), flush=True)
    if 'gpu_psnr' in result and 'cpu_psnr' in result:
        result['delta_db'] = result['gpu_psnr'] - result['cpu_psnr']
        print('delta = %+.3f dB' % result['delta_db'])
    print(json.dumps(result))
    if args.out:
        with open(args.out, 'w') as f:
            json.dump(result, f, indent=1)


if __name__ == '__main__':
    main()
