"""De-duplicated background evaluation (include/durf_hip.h durf_expand_raw; ops.DEDUP_HIT_RAYS): a ray that hits exactly
one box feeds the background MLP the same trunk input at every sample (obbpose_model.py:205-210), so it is evaluated
once per ray.  The reference evaluates it at every sample; the results must be the same:
  * forward: BIT-identical rendered outputs (an MFMA column does not depend on its neighbours),
  * backward: the single evaluation receives the SUM of the per-sample head gradients -- equal in exact arithmetic,
    different only in where bf16 rounding happens (sum first vs round first): gradients within 2e-3 norm-wise of the
    sample-by-sample path and, like it, within the usual 5e-2 of the oracle."""
import pytest
import torch

from durf_amd import obbpose_model, ops, synthetic, train_boxpose, utils
from oracle import durf_ref as R
from tests import helpers as H

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return float((a - b).norm() / (b.norm() + 1e-30))


def _setup(cuda, B, K, N, seed, multi=False):
    utils.clear_gin()
    utils.parse_gin('MipNerfModel.num_samples = %d\nMipNerfModel.density_noise = 0.0\n'
                    'MipNerfModel.no_pose_opt = True\nMipNerfModel.no_yaw_opt = True\n'
                    'Config.randomized = True\nConfig.rand_bkgd = False\nConfig.grad_max_norm = 1.0\n'
                    'Config.grad_max_val = 0.1\nConfig.tv_loss_mult = 0.0\n' % N)
    config = utils.configured(utils.Config)
    b = synthetic.make_batch(B, K, seed=seed, hit_range=(0.2, 0.4), allow_multi_hit=multi)
    db = H.device_batch(b, cuda)
    model, variables = obbpose_model.construct_mipnerf(seed, db, device=cuda)
    g = torch.Generator().manual_seed(seed)
    for name in variables.layout.mlp_names():
        for i in range(12):
            bias = variables['params'][name]['Dense_%d' % i]['bias']
            bias.copy_(((torch.rand(bias.shape, generator=g) - 0.5) * 0.1).to(cuda))
    noise = dict(t_rand=torch.rand(B, N + 1, generator=g).to(cuda), u_rand=torch.rand(B, N + 1, generator=g).to(cuda))
    return config, b, db, model, variables, noise


@pytest.mark.parametrize('B,K,N,multi', [(300, 1, 32, False), (517, 3, 64, False), (256, 8, 32, True), (1024, 3, 128, False)])
def test_forward_is_bit_identical_with_and_without_dedup(cuda, B, K, N, multi):
    config, b, db, model, variables, noise = _setup(cuda, B, K, N, 11 + K, multi)
    outs = {}
    for on in (True, False):
        ops.DEDUP_HIT_RAYS = on
        try:
            outs[on] = model.apply(variables, 0, db['rays'], db['init'], db['ext'], b['ts'], randomized=True,
                                   rand_bkgd=False, white_bkgd=False, alpha=10.0, noise=noise)
        finally:
            ops.DEDUP_HIT_RAYS = True
    nh = outs[True][0][8].reshape(-1)
    assert int((nh == 1).sum()) > 0, 'the batch must contain box-hit rays'
    if multi:
        assert int((nh > 1).sum()) > 0, 'and, here, rays that hit several boxes (they take the full path)'
    for lvl in range(2):
        for i, nm in enumerate(['rgb', 'depth', 'acc', 'weights', 't_vals']):
            a, c = outs[True][lvl][i], outs[False][lvl][i]
            assert torch.equal(torch.isnan(a), torch.isnan(c)), '%s level %d: NaN pattern' % (nm, lvl)
            if multi and lvl > 0:
                # A multi-hit ray has NaN t_vals at level 1, and one NaN sample sends its whole 64-sample wave of the
                # encode kernel down the exact-wrap path instead of the v_fract one (rays.hip).  Compaction changes
                # which rays share a wave with it, so THEIR features may differ by a bf16 rounding flip: close, not equal.
                torch.testing.assert_close(torch.nan_to_num(a), torch.nan_to_num(c), rtol=0, atol=2e-2 if i != 4 else 1e-3,
                                           msg=lambda m: '%s level %d: %s' % (nm, lvl, m))
            else:
                same = (a == c) | (torch.isnan(a) & torch.isnan(c))
                assert bool(same.all()), '%s level %d differs' % (nm, lvl)


@pytest.mark.parametrize('B,K,N', [(300, 1, 32), (517, 3, 32)])
def test_gradients_match_the_sample_by_sample_path_and_the_oracle(cuda, B, K, N):
    config, b, db, model, variables, noise = _setup(cuda, B, K, N, 21 + K)
    prev = db['init'][0:1]
    grads, stats = {}, {}
    for on in (True, False):
        ops.DEDUP_HIT_RAYS = on
        try:
            g, raw, _ = train_boxpose.loss_and_grad(model, config, 0, variables, db, 3.0, 10.0, prev, noise=noise)
            grads[on] = g.clone()
            stats[on] = train_boxpose._assemble_stats(config, db, raw, prev, ops.STATS_ASSEMBLE | ops.STATS_PSNR).clone()
        finally:
            ops.DEDUP_HIT_RAYS = True
    bad = torch.nonzero(~((stats[True] == stats[False]) | (torch.isnan(stats[True]) & torch.isnan(stats[False])))).flatten()
    assert bad.numel() == 0, 'losses come from the (bit-identical) forward; differing entries %s: %s vs %s' % (
        bad.tolist(), stats[True][bad].tolist(), stats[False][bad].tolist())
    lay = variables.layout
    sl = slice(lay.mlp_off['MLP_0'], lay.mlp_off['MLP_0'] + lay.mlp_size[256])
    r = _rel(grads[True][sl], grads[False][sl])
    assert r < 2e-3, 'background MLP gradient, dedup vs sample-by-sample: rel err %g' % r
    for k in range(K):                              # the object MLPs are untouched
        o = lay.mlp_off['BoxMLP_%d' % k]
        assert torch.equal(grads[True][o:o + lay.mlp_size[128]], grads[False][o:o + lay.mlp_size[128]])
    # and against the oracle (bf16-rounded GEMM operands), as tests/test_gpu_train.py::test_train_step
    ob = H.oracle_batch(b)
    params = H.oracle_params_from_variables(variables)
    ocfg = dict(R.CONFIG_DEFAULTS, randomized=True, tv_loss_mult=0.0)
    noise_c = {k: v.cpu() for k, v in noise.items()}
    _, _, ostats, ograds = R.train_step(params, R.new_opt_state(params), ob, ocfg, dict(num_samples=N), 5e-4, 3.0, 10.0,
                                        ob['init'][0:1], noise=noise_c, mlp_hook=R.mlp_apply_bf16)
    og = torch.cat([x.reshape(-1) for x in ograds])
    assert _rel(grads[True].cpu()[sl], og[sl]) < 5e-2


@pytest.mark.parametrize('case', ['all_rays_hit', 'no_ray_hits'])
def test_extreme_hit_fractions(cuda, case):
    """every ray in the once-per-ray class (no compacted rows at all) and none in it (an empty tail): forward
    bit-identical to the sample-by-sample path, a training step stays finite and matches it"""
    B, K, N = 200, 1, 32
    config, b, db, model, variables, noise = _setup(cuda, B, K, N, 5)
    if case == 'all_rays_hit':
        db['ext'] = db['ext'] * 0 + 1.0e3                        # a box that contains every camera
    else:
        db['init'] = db['init'].clone()
        db['init'][:, :, 2] = 1.0e4                              # far behind every camera
        model, variables = obbpose_model.construct_mipnerf(5, db, device=cuda)
    outs, grads = {}, {}
    for on in (True, False):
        ops.DEDUP_HIT_RAYS = on
        try:
            outs[on] = model.apply(variables, 0, db['rays'], db['init'], db['ext'], b['ts'], randomized=True,
                                   rand_bkgd=False, white_bkgd=False, alpha=10.0, noise=noise)
            grads[on], _, _ = train_boxpose.loss_and_grad(model, config, 0, variables, db, 3.0, 10.0, db['init'][0:1],
                                                          noise=noise)
        finally:
            ops.DEDUP_HIT_RAYS = True
    nh = outs[True][0][8].reshape(-1)
    assert bool((nh == (1 if case == 'all_rays_hit' else 0)).all())
    for lvl in range(2):
        for i in range(5):
            assert torch.equal(outs[True][lvl][i], outs[False][lvl][i]), (lvl, i)
    assert torch.isfinite(grads[True]).all()
    lay = variables.layout
    sl = slice(lay.mlp_off['MLP_0'], lay.mlp_off['MLP_0'] + lay.mlp_size[256])
    assert _rel(grads[True][sl], grads[False][sl]) < 2e-3
