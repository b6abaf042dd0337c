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
    _, u_all = philox_ref.step_draws_planes(seed, B, N)
    own = ops.ray_prologue(r.origins, r.directions, pose, ext, r.viewdirs, near, far, N, seed=seed)
    fed = ops.ray_prologue(r.origins, r.directions, pose, ext, r.viewdirs, near, far, N, t_rand=torch.from_numpy(t_ref).to(cuda))
    assert torch.equal(own[6].cpu(), torch.from_numpy(u_all)), 'resampling draws: words 1, 2, 3 of every Philox block, a plane per level'
    assert np.array_equal(u_all[0], u_ref) and not np.array_equal(u_all[0], u_all[1]) and not np.array_equal(u_all[1], u_all[2])
    own = own[:6] + (own[6][0],)
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
    assert not torch.equal(other[6][0], own[6]) and not torch.equal(other[5], own[5])


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


@pytest.mark.parametrize('rows,seed,level', [(77 * 32, 0, 0), (512 * 128, 20200823, 1), (100001, (1 << 40) + 12345, 3)])
def test_density_noise_draws_are_the_oracles_normal_stream(cuda, rows, seed, level):
    """durf_density_noise (MipNerfModel.density_noise, obbpose_model.py:236-240): the library's own standard-normal draws
    against oracle/philox_ref.density_draws -- the Philox words bit for bit underneath, Box-Muller in fp32 here and in float64
    there, hence the tolerance: 4 ulp of the largest |z| a 24-bit u1 can give (5.77) -- and injected draws exactly."""
    base = torch.randn(rows, 4, device=cuda)
    got = ops.density_noise(base.clone(), 1.0, seed=seed, level=level)
    z_ref = torch.from_numpy(philox_ref.density_draws(seed, rows, level))
    assert torch.equal(got[:, :3], base[:, :3]), 'the colour channels are not touched'
    z = (got[:, 3].double() - base[:, 3].double()).cpu()
    # (the sum base + z is rounded once in fp32: |base| < 6, |z| < 6 -> half an ulp of 16 on top of the draw's own error)
    torch.testing.assert_close(z, z_ref.double(), rtol=0, atol=4 * 2.0 ** -21 + 2.0 ** -20)
    zero = ops.density_noise(torch.zeros(rows, 4, device=cuda), 1.0, seed=seed, level=level)[:, 3].cpu()
    torch.testing.assert_close(zero, z_ref, rtol=0, atol=4 * 2.0 ** -21)
    assert abs(float(zero.mean())) < 4.0 / rows ** 0.5 and abs(float(zero.std()) - 1.0) < 0.02
    other = ops.density_noise(torch.zeros(rows, 4, device=cuda), 1.0, seed=seed, level=level + 1)[:, 3].cpu()
    assert abs(float(np.corrcoef(zero.numpy(), other.numpy())[0, 1])) < 4.0 / rows ** 0.5, 'levels draw independently'
    # injected draws: scale * z and the sum rounded separately, as the tensor expression raw[:, 3] += scale * z is
    nz = torch.randn(rows, device=cuda)
    fed = ops.density_noise(base.clone(), 0.1, normal=nz)
    assert torch.equal(fed[:, 3], base[:, 3] + 0.1 * nz) and torch.equal(fed[:, :3], base[:, :3])


def test_a_model_that_draws_its_density_noise_equals_the_model_handed_the_oracles_draws(cuda):
    B, K, N, rng = 256, 2, 32, 99
    utils.clear_gin()
    utils.parse_gin('MipNerfModel.num_samples = %d\nMipNerfModel.density_noise = 0.1\nMipNerfModel.no_pose_opt = True\n'
                    'MipNerfModel.no_yaw_opt = True\n' % N)
    b = synthetic.make_batch(B, K, seed=11)
    db = H.device_batch(b, cuda)
    model, variables = obbpose_model.construct_mipnerf(5, db, device=cuda)
    t_ref, u_ref = philox_ref.step_draws(rng, B, N)
    noise = dict(t_rand=torch.from_numpy(t_ref).to(cuda), u_rand=torch.from_numpy(u_ref).to(cuda),
                 density=[torch.from_numpy(philox_ref.density_draws(rng, B * N, lvl)).to(cuda) for lvl in range(2)])
    kw = dict(randomized=True, rand_bkgd=False, white_bkgd=False, alpha=10.0)
    own = model.apply(variables, rng, db['rays'], db['init'], db['ext'], b['ts'], **kw)
    fed = model.apply(variables, rng, db['rays'], db['init'], db['ext'], b['ts'], noise=noise, **kw)
    single = own[0][8].reshape(-1) <= 1          # (a ray that hits two boxes renders NaN on both sides)
    for lvl in range(2):
        for i in (0, 1, 2, 3, 4):
            # (draws a few ulp apart x 0.1 on the raw density; the second level's sample positions amplify that through
            # the inverse of the first level's weight histogram -- still 50 x below what the noise itself does, next line)
            torch.testing.assert_close(own[lvl][i][single], fed[lvl][i][single], rtol=1e-3, atol=2e-4)
    model.density_noise = 0.0
    quiet = model.apply(variables, rng, db['rays'], db['init'], db['ext'], b['ts'], **kw)
    assert (quiet[1][3][single] - own[1][3][single]).abs().max() > 1e-2, 'the noise must matter in this test'


def test_three_levels_draw_a_plane_of_their_own_each(cuda):
    """num_levels = 3: the resample behind level 0 uses Philox word 1, the one behind level 1 word 2 (until round 6 both used
    word 1: correlated draws).  A step that draws for itself == the step handed the oracle's planes; and handing it plane 0 for
    both resamples (the old behaviour) gives a DIFFERENT level-2 sample set."""
    B, N, K, rng = 200, 32, 2, 77
    utils.clear_gin()
    utils.parse_gin('MipNerfModel.num_samples = %d\nMipNerfModel.num_levels = 3\nMipNerfModel.density_noise = 0.0\n'
                    'MipNerfModel.no_pose_opt = True\nMipNerfModel.no_yaw_opt = True\nConfig.randomized = True\n' % N)
    config = utils.configured(utils.Config)
    db = H.device_batch(synthetic.make_batch(B, K, seed=12), cuda)
    t_ref, u_all = philox_ref.step_draws_planes(rng, B, N)
    tt, uu = torch.from_numpy(t_ref).to(cuda), torch.from_numpy(u_all).to(cuda)
    out = []
    for nz in (None, dict(t_rand=tt, u_rand=uu), dict(t_rand=tt, u_rand=uu[0].contiguous())):
        model, variables = obbpose_model.construct_mipnerf(5, db, device=cuda)
        ret = model.apply(variables, rng, db['rays'], db['init'], db['ext'], 0, randomized=True, rand_bkgd=False, white_bkgd=False,
                          alpha=10.0, noise=nz)
        torch.cuda.synchronize()
        out.append([r[4].clone() for r in ret])                 # t_vals of every level
    for a, c in zip(out[0], out[1]):
        assert torch.equal(a, c)
    assert torch.equal(out[0][1], out[2][1]) and not torch.equal(out[0][2], out[2][2])
    # ... and the one C call draws the same planes
    model, variables = obbpose_model.construct_mipnerf(5, db, device=cuda)
    one = model.apply_one_call(variables, rng, db['rays'], db['init'], db['ext'], 0, randomized=True, rand_bkgd=False,
                               white_bkgd=False, alpha=10.0)
    for lvl in range(3):
        assert torch.equal(one[lvl][4], out[0][lvl])
