"""Box-pose optimisation end to end (cfg4's path) against the fp32 CPU oracle's trajectory.

tests/golden/pose_trajectory_N128.npz (made by tests/golden/make_pose_trajectory.py in the build container) holds the
box pose the oracle reaches every 10 steps when it trains, with pose optimisation on, from a box pose perturbed by
(0.10, 0, 0.08) in position and 0.05 rad in rotation, on the batches of psnr_trajectory_N128.npz.  The HIP path trains
from the same parameters on the same batches; per step its bf16 pose gradient is noisy, so what is
compared is where the optimiser TAKES the pose: the two trajectories must stay together to a fraction of the distance
travelled."""
import os
import sys

import numpy as np
import pytest
import torch

from durf_amd import obbpose_model, synthetic, train_boxpose, utils

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIX = os.path.join(ROOT, 'tests', 'golden', 'pose_trajectory_N128.npz')
SRC = os.path.join(ROOT, 'tests', 'golden', 'psnr_trajectory_N128.npz')
pytestmark = pytest.mark.gpu


def _train(cuda, precision, z, p, M):
    N, nbatch, far, ts = int(z['num_samples']), int(z['nbatch']), float(z['far']), int(z['ts'])
    steps, every = int(p['steps']), int(p['every'])
    utils.clear_gin()
    utils.parse_gin(str(p['gin']) + ("MipNerfModel.mlp_precision = 'f32'\n" if precision == 'f32' else ''))
    config = utils.configured(utils.Config)
    common = dict(init=p['init'], ext=z['ext'], target=z['target'], ts=ts)
    tr = {k[6:]: z[k] for k in z.files if k.startswith('train_')}
    batches = [synthetic.device_batch(dict(M.unpack(tr, i, far), **common), cuda) for i in range(nbatch)]
    model, var_cpu = obbpose_model.construct_mipnerf(int(z['seed']), {k: torch.tensor(v) if isinstance(v, np.ndarray) else v
                                                                       for k, v in common.items()}, device='cpu')
    assert not model.no_pose_opt and not model.no_yaw_opt
    variables = var_cpu.like(var_cpu.flat.clone().to(cuda))
    state = train_boxpose.create_train_state(variables)
    pose_of = lambda st: st.variables['params']['box_centers'][ts, 0].detach().cpu().numpy().copy()
    got, loss = [pose_of(state)], []
    for step in range(1, steps + 1):
        lr, eps = M.schedule(config, step)
        db = batches[step % nbatch]
        state, stats, _, _ = train_boxpose.train_step(model, config, 0, state, db, lr, eps, 10.0, db['init'][0:1])
        loss.append(float(stats.loss))
        if step % every == 0:
            got.append(pose_of(state))
    return np.stack(got), np.array(loss)


def test_pose_trajectory_follows_the_oracle(cuda):
    """Measured (end-of-round-2 kernels): the oracle takes the pose up to 0.020 away from its start within 160 steps.  For
    the first 40 steps the exact-fp32 instrument reproduces it (<= 2 % of the distance travelled) and the bf16 production
    path stays within 13 %; after that the optimisation amplifies rounding differences (Adam's sign-like steps on a
    gradient that nearly cancels), so even exact fp32 arithmetic in another summation order ends 0.011 from the oracle,
    and bf16 0.005 -- no further than that.  All three end at the same loss (0.182 / 0.184 / 0.186)."""
    sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
    import make_psnr_trajectory as M
    z, p = np.load(SRC, allow_pickle=False), np.load(FIX, allow_pickle=False)
    want = p['oracle_pose']
    moved = np.linalg.norm(want - want[0], axis=1)                  # how far the oracle has taken the pose
    assert moved[-1] > 5e-3, 'the optimiser must actually move the box'
    apart = {}
    for precision in ('f32', 'bf16'):
        got, loss = _train(cuda, precision, z, p, M)
        np.testing.assert_array_equal(got[0], want[0])
        apart[precision] = np.linalg.norm(got - want, axis=1)
        print('%s: oracle moved %s\n      HIP - oracle %s\n      final loss %.4f (oracle %.4f)' % (
            precision, np.array2string(np.round(moved, 4), max_line_width=400), np.array2string(np.round(apart[precision], 4), max_line_width=400),
            loss[-10:].mean(), p['oracle_loss'][-10:].mean()))
        # the loss LEVEL at the end: medians over the last 30 steps (the series has single-step spikes of 1.5-4 x -- the oracle's
        # own last 40 steps hold 0.716, 0.277 and 0.252 among values around 0.19 -- and which step spikes is as chaotic as the
        # trajectory: a mean over 10 steps is at the mercy of one of them)
        med, omed = float(np.median(loss[-30:])), float(np.median(p['oracle_loss'][-30:]))
        print('      median of the last 30 losses %.4f (oracle %.4f)' % (med, omed))
        assert abs(med - omed) < 0.05 * omed          # measured: f32 0.8 %, bf16 2.0 %
    # early on (steps 10-40, before the optimisation has amplified anything) the exact-fp32 path IS the oracle and the
    # bf16 path is within a fifth of the distance travelled
    assert (apart['f32'][1:5] <= 0.05 * moved[1:5] + 4e-4).all(), (apart['f32'], moved)
    assert (apart['bf16'][1:5] <= 0.2 * moved[1:5] + 1e-4).all(), (apart['bf16'], moved)
    # later the trajectories drift apart -- exact fp32 (another summation order) by up to 0.66 of the distance travelled,
    # bf16 by up to 0.54: bounded.  (Until round 4 the test also held bf16 to twice the exact-fp32 drift.  Both drifts are the
    # optimiser's amplification of summation-order differences, not arithmetic error: the SAME bf16 kernels end 0.0054 from
    # the oracle when the weight-gradient partials are summed over 512 workgroups and 0.0086 when over 256 -- round 4's
    # small-batch plan, DURF_DW_WGS=512 restores the other -- while exact fp32 ended 0.011 away in round 2 and ends 0.0024
    # away now.  A relation between two such numbers tests nothing; the bound on each against the distance travelled stays.)
    for precision in ('f32', 'bf16'):
        assert (apart[precision][1:] <= 0.8 * moved[1:] + 1e-3).all(), (precision, apart[precision], moved)
