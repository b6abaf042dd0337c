"""durf_train_step / durf_loss_backward (csrc/train.hip): one shard's training step as ONE C call -- with durf_forward the
coarse entry points SURVEY 8b proposed for hosts that are not Python.  They issue the same stage kernels in the same order as
durf_amd/train_boxpose.py does, so parameters, Adam moments, the gradient, every logged scalar and every rendered output must
be BIT-identical to train_step / loss_and_grad -- which are the paths the parity tests measure against the oracle and
against the reference's own train_step (tests/test_golden_ref_train.py)."""
import pytest
import torch

from durf_amd import obbpose_model, ops as ops_mod, synthetic, train_boxpose, utils
from tests import helpers as H

pytestmark = pytest.mark.gpu


def _eq(a, b, what):
    assert torch.allclose(a, b, rtol=0, atol=0, equal_nan=True), what


POSE = 'MipNerfModel.no_pose_opt = False\nMipNerfModel.no_yaw_opt = False\n'


@pytest.mark.parametrize('B,K,N,randomized,extra', [
    (4096, 3, 128, True, ''),                                            # the benchmarked shape (cfg3)
    (1000, 1, 64, True, 'Config.white_bkgd = True\nConfig.box_loss_mult = 2\n'),
    (777, 0, 32, False, 'Config.disable_multiscale_loss = True\n'),      # static model, ragged ray count
    (640, 8, 32, True, 'MipNerfModel.ray_shape = "cylinder"\nMipNerfModel.disable_integration = True\n'),
    (512, 2, 32, False, 'MipNerfModel.num_levels = 3\n'),
    # box-pose optimisation (BASELINE cfg4): the object branch and the hit rays' background evaluation in fp32, the pose
    # gradient + TV prior into this timestep's rows of box_centers, the poses moving from step to step
    (1024, 3, 128, True, POSE),                                          # cfg4's per-rank shape
    (600, 2, 32, True, 'MipNerfModel.no_pose_opt = True\nMipNerfModel.no_yaw_opt = False\n'),     # yaw only: no TV term
    (512, 1, 32, False, POSE + 'Config.tv_loss_mult = 0.0\nConfig.white_bkgd = True\n'),
    (512, 3, 32, True, 'MipNerfModel.obj_precision = "f32"\n'),          # fp32 object branch, frozen poses
    (512, 2, 32, True, POSE + 'MipNerfModel.ray_shape = "cylinder"\nMipNerfModel.disable_integration = True\n'),
    # the class defaults no shipped gin file keeps (utils.py:142-144, obbpose_model.py:57): density noise drawn by the library
    # under the step's key, weight decay, no background colour
    (700, 2, 32, True, 'MipNerfModel.density_noise = 0.1\nConfig.weight_decay_mult = 1e-4\nConfig.rand_bkgd = True\n'),
    (512, 0, 64, True, 'MipNerfModel.density_noise = 0.05\nConfig.weight_decay_mult = 1e-2\n'),
    (512, 0, 32, False, 'MipNerfModel.density_noise = 0.1\nConfig.rand_bkgd = True\n'),      # not randomized: no noise either
    (512, 2, 32, True, POSE + 'Config.weight_decay_mult = 1e-2\nMipNerfModel.density_noise = 0.1\n'),   # decay under the pose rows' gradient
])
def test_one_call_train_step_is_bit_identical_to_train_step(cuda, B, K, N, randomized, extra):
    utils.clear_gin()
    base = ('MipNerfModel.num_samples = %d\nMipNerfModel.density_noise = 0.0\nMipNerfModel.no_pose_opt = True\n'
            'MipNerfModel.no_yaw_opt = True\nConfig.randomized = %s\nConfig.rand_bkgd = False\nConfig.grad_max_norm = 1.0\n'
            'Config.grad_max_val = 0.1\nConfig.tv_loss_mult = 0.01\n' % (N, randomized))
    utils.parse_gin(base + extra)
    config = utils.configured(utils.Config)
    b = synthetic.make_batch(B, K, seed=950 + K, allow_multi_hit=K > 1, noise_boxes=0.2)
    db = H.device_batch(b, cuda)
    prev = db['init'][0:1] + 0.01
    results = []
    for fn in (train_boxpose.train_step, train_boxpose.train_step_one_call):
        model, variables = obbpose_model.construct_mipnerf(3, db, device=cuda)
        state = train_boxpose.create_train_state(variables)
        rng, log = 11, []
        for _ in range(3):
            state, stats, rng, pose = fn(model, config, rng, state, db, 5e-4, 0.7, 6.5, prev)
            log.append((stats, pose))
        torch.cuda.synchronize()
        results.append((state, log))
    (s0, l0), (s1, l1) = results
    _eq(s0.variables.flat, s1.variables.flat, 'parameters after 3 steps')
    _eq(s0.m, s1.m, 'Adam m')
    _eq(s0.v, s1.v, 'Adam v')
    assert s0.step == s1.step == 3
    for step, ((a, pa), (c, pc)) in enumerate(zip(l0, l1)):
        _eq(pa, pc, 'pose')
        for name in ('loss', 'losses', 'obj_losses', 'd_losses', 'n_losses', 'e_losses', 's_losses', 'distr_losses', 'tv_losses',
                     'sampling_stats', 'offsets', 'offset_x', 'offset_y', 'offset_z', 'offset_yaw', 'psnr', 'psnrs', 'obj_psnr',
                     'grad_norm', 'grad_abs_max', 'grad_norm_clipped'):
            _eq(getattr(a, name), getattr(c, name), 'step %d: %s' % (step, name))
        assert int(a.multi_hit_rays) == int(c.multi_hit_rays)
        for wa, wc in zip(a.weights + a.samples, c.weights + c.samples):
            _eq(wa, wc, 'step %d: logged weights / samples' % step)


@pytest.mark.parametrize('K', [0, 2])
def test_one_call_train_step_takes_its_draws_from_a_generator_as_train_step_does(cuda, K):
    """`rng` a torch.Generator: sampling draws and density noise come from it, in the same order on both paths"""
    B, N = 512, 32
    utils.clear_gin()
    utils.parse_gin('MipNerfModel.num_samples = %d\nMipNerfModel.density_noise = 0.1\nMipNerfModel.no_pose_opt = True\n'
                    'MipNerfModel.no_yaw_opt = True\nConfig.randomized = True\nConfig.rand_bkgd = True\n'
                    'Config.weight_decay_mult = 1e-3\n' % N)
    config = utils.configured(utils.Config)
    db = H.device_batch(synthetic.make_batch(B, K, seed=977), cuda)
    out = []
    for fn in (train_boxpose.train_step, train_boxpose.train_step_one_call):
        model, variables = obbpose_model.construct_mipnerf(3, db, device=cuda)
        state = train_boxpose.create_train_state(variables)
        rng = torch.Generator(device=cuda).manual_seed(31)
        for _ in range(2):
            state, stats, rng, _ = fn(model, config, rng, state, db, 5e-4, 0.7, 6.5, db['init'][0:1])
        torch.cuda.synchronize()
        out.append((state.variables.flat.clone(), stats))
    _eq(out[0][0], out[1][0], 'parameters after 2 steps')
    _eq(out[0][1].loss, out[1][1].loss, 'loss')
    _eq(out[0][1].weight_l2, out[1][1].weight_l2, 'weight_l2')
    assert float(out[0][1].weight_l2) > 0


def test_loss_backward_gives_the_gradient_of_loss_and_grad(cuda):
    B, K, N = 2048, 3, 64
    utils.clear_gin()
    utils.parse_gin('MipNerfModel.num_samples = %d\nMipNerfModel.density_noise = 0.1\nMipNerfModel.no_pose_opt = True\n'
                    'MipNerfModel.no_yaw_opt = True\nConfig.randomized = True\nConfig.rand_bkgd = False\n'
                    'Config.weight_decay_mult = 1e-4\n' % N)
    config = utils.configured(utils.Config)
    b = synthetic.make_batch(B, K, seed=961)
    db = H.device_batch(b, cuda)
    model, variables = obbpose_model.construct_mipnerf(3, db, device=cuda)
    state = train_boxpose.create_train_state(variables)
    g = torch.Generator().manual_seed(2)
    noise = dict(t_rand=torch.rand(B, N + 1, generator=g).to(cuda), u_rand=torch.rand(B, N + 1, generator=g).to(cuda),
                 density=[torch.randn(B, N, generator=g).to(cuda) for _ in range(2)])
    flat0 = variables.flat.clone()
    want, raw, _ = train_boxpose.loss_and_grad(model, config, 0, variables, db, 3.0, 10.0, db['init'][0:1], noise=noise)
    got, st, ret = train_boxpose.train_step_one_call(model, config, 0, state, db, 5e-4, 3.0, 10.0, db['init'][0:1], noise=noise,
                                                     update=False)
    _eq(got, want, 'gradient')
    _eq(variables.flat, flat0, 'durf_loss_backward leaves the parameters alone')
    for lvl in range(2):
        for i in range(7):
            _eq(ret[lvl][i], raw['ret'][lvl][i], 'level %d output %d' % (lvl, i))


def test_unsupported_configurations_are_refused(cuda):
    utils.clear_gin()
    db = H.device_batch(synthetic.make_batch(256, 1, seed=5), cuda)
    for gin in ('MipNerfModel.no_pose_opt = False\nMipNerfModel.obj_precision = "bf16"\n',     # pose gradient behind bf16 objects
                'MipNerfModel.num_levels = 1\n', 'MipNerfModel.mlp_precision = "f32"\n'):
        utils.clear_gin()
        utils.parse_gin('MipNerfModel.num_samples = 32\n' + gin)
        config = utils.configured(utils.Config)
        model, variables = obbpose_model.construct_mipnerf(3, db, device=cuda)
        with pytest.raises(NotImplementedError):
            train_boxpose.train_step_one_call(model, config, 0, train_boxpose.create_train_state(variables), db, 5e-4, 3.0, 10.0,
                                              db['init'][0:1])


def test_timing_hooks_of_the_one_call_step(cuda):
    """durf_train_args.timing (bench.py's live roofline timers when the step runs through the C call): HIP events recorded
    around the background MLP's forward / backward launch of every level, the fused per-ray launch and the weight-gradient
    launch -- under the names the Python-issued path's timers use -- without changing a bit of the step"""
    from durf_amd import ops
    B, K, N = 1024, 2, 64
    utils.clear_gin()
    utils.parse_gin('MipNerfModel.num_samples = %d\nMipNerfModel.density_noise = 0.0\nMipNerfModel.no_pose_opt = True\n'
                    'MipNerfModel.no_yaw_opt = True\nConfig.randomized = True\nConfig.rand_bkgd = False\n' % N)
    config = utils.configured(utils.Config)
    db = H.device_batch(synthetic.make_batch(B, K, seed=981), cuda)
    out = {}
    for timed in (False, True):
        model, variables = obbpose_model.construct_mipnerf(3, db, device=cuda)
        state = train_boxpose.create_train_state(variables)
        assert train_boxpose.best_step_fn(model, variables) is train_boxpose.train_step_one_call
        ops.TIMERS, ops.TIMED_NAMES, ops.TIMERS_ACTIVE = ({} if timed else None), None, True
        try:
            for i in range(2):
                state, stats, _, _ = train_boxpose.train_step_one_call(model, config, 5 + i, state, db, 5e-4, 3.0, 10.0, db['init'][0:1])
            totals = ops.timer_totals() if timed else None
        finally:
            ops.TIMERS = None
        out[timed] = (state.variables.flat.clone(), float(stats.loss), totals)
    assert torch.equal(out[False][0], out[True][0]) and out[False][1] == out[True][1]
    totals = out[True][2]
    # 2 steps x 2 levels of forward / backward, 2 x 1 fused per-ray launch (the last level's composite is the loss launch's), 2 x 1 dW
    assert {k: v[0] for k, v in totals.items()} == {'mlp_fwd_256_train': 4, 'mlp_bwd_256': 4, 'composite_resample': 2, 'mlp_dw_256': 2,
                                                    'train_call': 2}         # (+ the wrapper's own bracket around the whole call)
    assert totals['train_call'][1] > totals['mlp_dw_256'][1] + totals['mlp_fwd_256_train'][1] + totals['mlp_bwd_256'][1]
    for name, (n, sec) in totals.items():
        assert 2e-6 < sec / n < 5e-3, (name, sec / n)
    assert totals['mlp_dw_256'][1] / 2 > totals['composite_resample'][1] / 2


def test_constant_trunk_kept_across_one_call_steps_follows_the_parameters(cuda):
    """cfg4's fp32 hit-ray branch through the C call keeps the background trunk of the box-hit rays across steps
    (durf_train_args.const_trunk: refilled behind the update, used by the next call while the caller vouches for the
    parameters).  The wrapper's guard -- torch's version counter + the library's generation count -- must drop it when
    anything else writes the parameters between two steps: here torch does, after step 1, on both paths."""
    B, K, N = 512, 2, 32
    utils.clear_gin()
    utils.parse_gin('MipNerfModel.num_samples = %d\nMipNerfModel.density_noise = 0.0\n' % N + POSE +
                    'Config.randomized = True\nConfig.rand_bkgd = False\nConfig.tv_loss_mult = 0.01\n')
    config = utils.configured(utils.Config)
    db = H.device_batch(synthetic.make_batch(B, K, seed=990, noise_boxes=0.2), cuda)
    prev = db['init'][0:1] + 0.01
    out = []
    for fn in (train_boxpose.train_step, train_boxpose.train_step_one_call):
        model, variables = obbpose_model.construct_mipnerf(3, db, device=cuda)
        assert model.object_precision() == 'f32'
        state = train_boxpose.create_train_state(variables)
        rng = 21
        for i in range(4):
            state, stats, rng, _ = fn(model, config, rng, state, db, 5e-4, 0.7, 6.5, prev)
            if i == 1:
                state.variables.flat.mul_(1.001)          # someone else writes the parameters: the kept trunk is stale
            if fn is train_boxpose.train_step_one_call:
                c = state.variables._c_trunk
                fresh = c['key'] == (state.variables.flat._version, ops_mod.param_generation(state.variables.flat))
                assert fresh == (i != 1), 'step %d: the kept trunk is %s' % (i, 'fresh' if fresh else 'stale')
        torch.cuda.synchronize()
        out.append((state.variables.flat.clone(), float(stats.loss)))
    _eq(out[0][0], out[1][0], 'parameters after 4 steps with a foreign write in between')
    assert out[0][1] == out[1][1]
