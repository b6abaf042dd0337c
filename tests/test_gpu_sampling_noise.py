"""The stratified-sampling draws of a step are made INSIDE its first launch (durf_ray_prologue's Philox4x32-10 stream keyed
by the host's PRNG key), as the reference draws inside its program (mip.py:364, math.py:257-260), not by a generator launch
in front of it.  The kernel's stream is compared bit for bit with oracle/philox_ref.py (itself pinned by the generator's
published known-answer vectors, tests/test_oracle_math.py), and a step that draws for itself with a step that is handed
the same draws."""
import numpy as np
import pytest
import torch

from durf_amd import obbpose_model, ops, synthetic, train_boxpose, utils
from oracle import philox_ref
from tests import helpers as H

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('B,N,seed', [(77, 32, 0), (512, 128, 20200823), (1000, 64, (1 << 40) + 12345)])
def test_in_kernel_draws_are_the_oracles_philox_stream(cuda, B, N, seed):
    b = synthetic.make_batch(B, 2, seed=3)
    db = H.device_batch(b, cuda)
    r = db['rays']
    pose, ext = db['init'][b['ts']].contiguous(), db['ext'].reshape(-1, 3).contiguous()
    near, far = r.near.reshape(-1).contiguous(), r.far.reshape(-1).contiguous()
    t_ref, u_ref = philox_ref.step_draws(seed, B, N)
    own = ops.ray_prologue(r.origins, r.directions, pose, ext, r.viewdirs, near, far, N, seed=seed)
    fed = ops.ray_prologue(r.origins, r.directions, pose, ext, r.viewdirs, near, far, N, t_rand=torch.from_numpy(t_ref).to(cuda))
    assert torch.equal(own[6].cpu(), torch.from_numpy(u_ref)), 'resampling draws: word 1 of every Philox block'
    assert torch.equal(own[5], fed[5]), 'level-0 sample positions: jittered by word 0 exactly as t_rand would'
    for a, c in zip(own[:5], fed[:5]):
        assert torch.equal(a, c)
    # uniform on [0, 1), the two words uncorrelated (what mip.py:364 asks of the draws)
    u = own[6].double()
    assert 0.0 <= float(u.min()) and float(u.max()) < 1.0
    n = u.numel()
    assert abs(float(u.mean()) - 0.5) < 4.0 / (12 * n) ** 0.5 and abs(float(u.var()) - 1.0 / 12) < 0.01
    assert abs(float(np.corrcoef(t_ref.ravel(), u_ref.ravel())[0, 1])) < 4.0 / n ** 0.5
    other = ops.ray_prologue(r.origins, r.directions, pose, ext, r.viewdirs, near, far, N, seed=seed + 1)
    assert not torch.equal(other[6], own[6]) and not torch.equal(other[5], own[5])


@pytest.mark.parametrize('K,one_call', [(3, False), (3, True), (0, False)])
def test_a_step_that_draws_for_itself_equals_the_step_that_is_handed_the_draws(cuda, K, one_call):
    B, N, rng = 256, 32, 4242
    utils.clear_gin()
    utils.parse_gin('MipNerfModel.num_samples = %d\nMipNerfModel.density_noise = 0.0\nMipNerfModel.no_pose_opt = True\n'
                    'MipNerfModel.no_yaw_opt = True\nConfig.randomized = True\nConfig.rand_bkgd = False\n' % N)
    config = utils.configured(utils.Config)
    db = H.device_batch(synthetic.make_batch(B, K, seed=11), cuda)
    t_ref, u_ref = philox_ref.step_draws(rng, B, N)
    noise = dict(t_rand=torch.from_numpy(t_ref).to(cuda), u_rand=torch.from_numpy(u_ref).to(cuda))
    step = train_boxpose.train_step_one_call if one_call else train_boxpose.train_step
    out = []
    for nz in (None, noise):
        model, variables = obbpose_model.construct_mipnerf(5, db, device=cuda)
        state = train_boxpose.create_train_state(variables)
        state, stats, new_rng, _ = step(model, config, rng, state, db, 5e-4, 3.0, 10.0, db['init'][0:1], noise=nz)
        torch.cuda.synchronize()
        assert new_rng == rng + 1
        out.append((state.variables.flat.clone(), state.m.clone(), float(stats.loss)))
    assert torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1]) and out[0][2] == out[1][2]
