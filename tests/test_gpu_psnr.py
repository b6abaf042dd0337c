"""PSNR parity at the metric's 128 samples/ray (BASELINE.json north_star: "PSNR within 0.1 dB of the reference").

tests/golden/psnr_trajectory_N128.npz (made by tests/golden/make_psnr_trajectory.py in the build container) holds
the fp32 CPU oracle's training trajectory on the procedural scene, the batches it consumed and its held-out PSNR at
six steps.  Here the HIP path (bf16 MFMA MLPs) trains from the SAME parameters on the SAME batches with the SAME
schedules, and its held-out PSNR must stay within 0.1 dB of the oracle's at the end of training (0.15 dB along the
way, where the PSNR is still moving by several dB per evaluation)."""
import os
import sys

import numpy as np
import pytest
import torch

from durf_amd import math as dmath, obbpose_model, synthetic, train_boxpose, utils

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIX = os.path.join(ROOT, 'tests', 'golden', 'psnr_trajectory_N128.npz')
pytestmark = pytest.mark.gpu


def test_psnr_trajectory_matches_the_oracle_at_128_samples(cuda):
    sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
    import make_psnr_trajectory as M
    z = np.load(FIX, allow_pickle=False)
    N, steps, nbatch, far = int(z['num_samples']), int(z['steps']), int(z['nbatch']), float(z['far'])
    assert N == 128
    utils.clear_gin()
    utils.parse_gin(str(z['gin']))
    config = utils.configured(utils.Config)
    common = dict(init=z['init'], ext=z['ext'], target=z['target'], ts=int(z['ts']))
    tr = {k[6:]: z[k] for k in z.files if k.startswith('train_')}
    te = {k[5:]: z[k] for k in z.files if k.startswith('test_')}
    batches = [synthetic.device_batch(dict(M.unpack(tr, i, far), **common), cuda) for i in range(nbatch)]
    test = synthetic.device_batch(dict(M.unpack(te, 0, far), **common), cuda)
    model, var_cpu = obbpose_model.construct_mipnerf(int(z['seed']), {k: torch.tensor(v) if isinstance(v, np.ndarray) else v
                                                                       for k, v in common.items()}, device='cpu')
    variables = var_cpu.like(var_cpu.flat.clone().to(cuda))
    state = train_boxpose.create_train_state(variables)
    eval_at = [int(s) for s in z['eval_at']]
    got = []
    for step in range(1, steps + 1):
        lr, eps = M.schedule(config, step)
        db = batches[step % nbatch]
        state, stats, _, _ = train_boxpose.train_step(model, config, 0, state, db, lr, eps, 10.0, db['init'][0:1])
        if step in eval_at:
            ret = model.apply(state.variables, 0, test['rays'], test['init'], test['ext'], test['ts'], randomized=False,
                              rand_bkgd=False, white_bkgd=False, alpha=10.0)
            got.append(float(dmath.mse_to_psnr(((ret[-1][0] - test['pixels']) ** 2).mean())))
    want = z['oracle_test_psnr']
    delta = np.array(got) - want
    print('held-out PSNR  HIP %s\n               oracle %s\n               delta %s' % (
        np.round(got, 3), np.round(want, 3), np.round(delta, 3)))
    assert abs(delta[-1]) <= 0.1, 'final PSNR: HIP %.3f dB vs oracle %.3f dB' % (got[-1], want[-1])
    assert np.abs(delta).max() <= 0.15, delta
    assert got[-1] > got[0] + 1.0, 'training must actually improve the held-out PSNR'
