"""MipNerfModel.use_viewdirs = False on the device (obbpose_model.py:47,221-232,336-352: an MLP without a condition has no
bottleneck and no view layer; durf_amd/noview.py evaluates its 10-Dense tree through the 12-Dense kernels): rendered values,
loss terms, the gradient of the 10-Dense parameters and the optimizer step against the oracle -- whose own no-condition MLP
is pinned by the reference's model run with the knob off (tests/golden/ref_model_K2_N32_static_noview.npz,
tests/test_golden_ref_model.py)."""
import pytest
import torch

from durf_amd import obbpose_model, synthetic, train_boxpose, utils
from oracle import durf_ref as R
from tests import helpers as H

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return float((a - b).norm() / (b.norm() + 1e-30))


def _setup(cuda, B, K, N, dynamics, precision, seed):
    utils.clear_gin()
    utils.parse_gin('MipNerfModel.num_samples = %d\nMipNerfModel.density_noise = 0.0\nMipNerfModel.use_viewdirs = False\n'
                    'MipNerfModel.dynamics = %s\nMipNerfModel.mlp_precision = "%s"\nMipNerfModel.no_pose_opt = True\n'
                    'MipNerfModel.no_yaw_opt = True\nConfig.randomized = True\nConfig.rand_bkgd = False\n'
                    'Config.grad_max_norm = 1.0\nConfig.grad_max_val = 0.1\nConfig.tv_loss_mult = 0.0\n'
                    'Config.weight_decay_mult = 1e-4\n' % (N, dynamics, precision))
    config = utils.configured(utils.Config)
    b = synthetic.make_batch(B, K, seed=seed)
    ob, db = H.oracle_batch(b), H.device_batch(b, cuda)
    model, variables = obbpose_model.construct_mipnerf(1, db, device=cuda)
    g = torch.Generator().manual_seed(4)
    for name in variables.layout.mlp_names():
        for i in range(len(variables.layout.layer_shapes(name))):
            bias = variables['params'][name]['Dense_%d' % i]['bias']
            bias.copy_(((torch.rand(bias.shape, generator=g) - 0.5) * 0.1).to(cuda))
    noise_c = dict(t_rand=torch.rand(B, N + 1, generator=g), u_rand=torch.rand(B, N + 1, generator=g))
    noise_d = {k: v.to(cuda) for k, v in noise_c.items()}
    return config, b, ob, db, model, variables, noise_c, noise_d


@pytest.mark.parametrize('K,dynamics,precision', [(0, True, 'bf16'), (2, False, 'bf16'), (0, True, 'f32')])
def test_forward_without_view_directions(cuda, K, dynamics, precision):
    B, N, far = 256, 32, 40.0
    config, b, ob, db, model, variables, noise_c, noise_d = _setup(cuda, B, K, N, dynamics, precision, seed=41 + K)
    assert len(variables['params']['MLP_0']) == 10 and variables['params']['MLP_0']['Dense_9']['kernel'].shape == (256, 3)
    kw = dict(randomized=True, rand_bkgd=False, white_bkgd=False, alpha=10.0)
    ret = model.apply(variables, 0, db['rays'], db['init'], db['ext'], b['ts'], noise=noise_d, **kw)
    one = model.apply_one_call(variables, 0, db['rays'], db['init'], db['ext'], b['ts'], noise=noise_d, **kw) \
        if model.supports_one_call(variables, True) else ret          # (durf_forward: bf16, no boxes under dynamics=False)
    params = H.oracle_params_from_variables(variables)
    cfg = dict(num_samples=N, density_noise=0.0, use_viewdirs=False, dynamics=dynamics)
    with torch.no_grad():
        ref = R.model_apply(params, ob['rays'], b['ts'], ob['ext'], True, False, False, 10.0, noise=noise_c, cfg=cfg,
                            mlp_hook=R.mlp_apply_bf16 if precision == 'bf16' else None)
        ref_view = R.model_apply(dict(params, MLP_0=params['MLP_0'][:9] + [[torch.zeros(256, 256), torch.zeros(256)],
                                                                         [torch.zeros(283, 128), torch.zeros(128)],
                                                                         [torch.zeros(128, 3), torch.zeros(3)]]),
                                 ob['rays'], b['ts'], ob['ext'], True, False, False, 10.0, noise=noise_c,
                                 cfg=dict(cfg, use_viewdirs=True))
    # bf16: test_gpu_model.py's tolerances (raw_rgb carries one more bf16 rounding here: the head's output, 2^-9 relative);
    # f32: the exact-fp32 kernels, where the embedding costs nothing
    tols = ((0, 3e-3), (2, 3e-3), (3, 3e-3), (4, 2e-3 * far)) if precision == 'bf16' else \
        ((0, 2e-5), (2, 2e-5), (3, 5e-5), (4, 1e-4 * far))
    for lvl in range(2):
        for i, tol in tols:
            torch.testing.assert_close(ret[lvl][i].cpu(), ref[lvl][i], rtol=0, atol=tol, msg=lambda m: 'l%d out %d: %s' % (lvl, i, m))
            assert torch.equal(one[lvl][i], ret[lvl][i]), 'durf_forward on the embedded tree'
    assert (ref[1][0] - ref_view[1][0]).abs().max() > 1e-2, 'the rgb head must matter in this test'


@pytest.mark.parametrize('precision', ['bf16', 'f32'])
def test_train_step_without_view_directions(cuda, precision):
    """one full step: loss terms, weight_l2 over the 10-Dense tree, d(loss)/d(every real parameter), clip + Adam"""
    B, K, N = 256, 0, 32
    config, b, ob, db, model, variables, noise_c, noise_d = _setup(cuda, B, K, N, True, precision, seed=43)
    params = H.oracle_params_from_variables(variables)
    flat0 = variables.flat.clone()
    lr, eps, alpha = 5e-4, 3.0, 10.0
    prev_c, prev_d = ob['init'][0:1], db['init'][0:1]
    grad, raw, _ = train_boxpose.loss_and_grad(model, config, 0, variables, db, eps, alpha, prev_d, noise=noise_d)
    assert grad.shape == variables.flat.shape
    state = train_boxpose.create_train_state(variables)
    new_state, stats, _, _ = train_boxpose.train_step(model, config, 0, state, db, lr, eps, alpha, prev_d, noise=noise_d)
    torch.cuda.synchronize()
    ocfg = dict(R.CONFIG_DEFAULTS, randomized=True, tv_loss_mult=0.0, weight_decay_mult=1e-4)
    p2, _, ostats, ograds = R.train_step(params, R.new_opt_state(params), ob, ocfg, dict(num_samples=N, use_viewdirs=False), lr,
                                         eps, alpha, prev_c, noise=noise_c,
                                         mlp_hook=R.mlp_apply_bf16 if precision == 'bf16' else None)
    rt = 2e-3 if precision == 'bf16' else 2e-5
    for k in ('losses', 'd_losses', 'n_losses', 'e_losses', 's_losses', 'distr_losses'):
        torch.testing.assert_close(getattr(stats, k).cpu(), ostats[k], rtol=rt, atol=1e-6, msg=lambda m: k + ': ' + m)
    torch.testing.assert_close(stats.loss.cpu(), ostats['loss'], rtol=rt, atol=1e-6)
    torch.testing.assert_close(stats.weight_l2.cpu().reshape(()), ostats['weight_l2'].reshape(()).float(), rtol=1e-5, atol=0)
    og = torch.cat([x.reshape(-1) for x in ograds])
    assert og.numel() == grad.numel() == variables.flat.numel()
    gt = 5e-2 if precision == 'bf16' else 2e-4
    assert _rel(grad.cpu(), og) < gt, 'gradient rel err %g' % _rel(grad.cpu(), og)
    lay = variables.layout
    o9 = lay.mlp_off['MLP_0'] + lay.mlp_size[obbpose_model.W_BKGD] - (256 * 3 + 3)
    assert float(og[o9:].norm()) > 0 and _rel(grad.cpu()[o9:], og[o9:]) < gt, 'the rgb head (Dense_9) on its own'
    newflat = torch.cat([x.reshape(-1) for x in R.params_leaves(p2)])
    assert _rel(new_state.variables.flat.cpu() - flat0.cpu(), newflat - flat0.cpu()) < (0.15 if precision == 'bf16' else 0.02)
    assert new_state.step == 1
    # the step the C entry point would take is not offered for this tree (its optimizer runs on the embedded buffer)
    with pytest.raises(NotImplementedError):
        train_boxpose.train_step_one_call(model, config, 0, new_state, db, lr, eps, alpha, prev_d, noise=noise_d)


def test_dynamic_boxes_without_view_directions_are_refused_as_the_reference_fails(cuda):
    config, b, ob, db, model, variables, noise_c, noise_d = _setup(cuda, 64, 2, 32, True, 'bf16', seed=45)
    with pytest.raises(NotImplementedError):
        model.apply(variables, 0, db['rays'], db['init'], db['ext'], b['ts'], randomized=False, rand_bkgd=False,
                    white_bkgd=False, alpha=10.0)
    with pytest.raises(NameError):
        R.model_apply(H.oracle_params_from_variables(variables), ob['rays'], b['ts'], ob['ext'], False, False, False, 10.0,
                      cfg=dict(num_samples=32, use_viewdirs=False))
