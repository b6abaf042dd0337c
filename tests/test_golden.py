"""Golden vectors (tests/golden/*.npz, made by tests/golden/make_golden.py from the float64 oracle).
CPU: the float32 oracle ("reference-precision twin") reproduces them -- this bounds what an fp32
evaluation of the reference may differ by.  GPU: the HIP path reproduces them within the bf16
tolerance SURVEY.md 8c states (rgb <= 2e-2 abs, loss terms <= 2 % rel)."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
import make_golden as MG  # noqa: E402  (the committed generator of the fixtures; imports the oracle, so it lives under tests/)

from durf_amd import obbpose_model, train_boxpose, utils  # noqa: E402
from tests import helpers as H  # noqa: E402


def _load(name):
    return np.load(os.path.join(ROOT, 'tests', 'golden', name + '.npz'))


@pytest.mark.parametrize('name', list(MG.CASES))
def test_generator_reproduces_committed_fixture(name):
    """Re-running the generator (float64 oracle) gives the committed vectors EXACTLY: an oracle edit that moves
    the default path fails here instead of being re-pinned silently by the next `make_golden.py` run."""
    gold = _load(name)
    _, variables, noise, out = MG.oracle_outputs(name)
    for k, v in out.items():
        np.testing.assert_array_equal(np.asarray(v), gold[k], err_msg=k)
    np.testing.assert_array_equal(noise['t_rand'].numpy(), gold['t_rand'])
    np.testing.assert_array_equal(noise['u_rand'].numpy(), gold['u_rand'])
    assert set(gold.files) == set(out) | {'param_checksum', 't_rand', 'u_rand'}


@pytest.mark.parametrize('name', list(MG.CASES))
def test_fp32_oracle_reproduces_golden(name):
    gold = _load(name)
    b, variables, noise, out = MG.oracle_outputs(name, torch.float32)
    flat = variables.flat.double()
    np.testing.assert_allclose([float(flat.sum()), float((flat * flat).sum())], gold['param_checksum'], rtol=1e-12)
    np.testing.assert_array_equal(noise['t_rand'].numpy(), gold['t_rand'])
    far = 40.0
    tol = dict(rgb=2e-5, acc=2e-5, weights=2e-5, depth=1e-4 * far, t_vals=1e-4 * far)
    for lvl in range(2):
        for nm, t in tol.items():
            k = 'l%d_%s' % (lvl, nm)
            np.testing.assert_allclose(out[k], gold[k], rtol=0, atol=t, err_msg=k)
    np.testing.assert_array_equal(out['dyn_mask'], gold['dyn_mask'])
    for k in ('loss', 'losses', 'd_losses', 'n_losses', 'e_losses', 's_losses', 'distr_losses'):
        np.testing.assert_allclose(out['stat_' + k], gold['stat_' + k], rtol=1e-4, atol=1e-7, err_msg=k)
    np.testing.assert_allclose(out['grad_norm'], gold['grad_norm'], rtol=1e-3)
    rel = np.linalg.norm(out['grad_head'] - gold['grad_head']) / np.linalg.norm(gold['grad_head'])
    assert rel < 1e-2, rel      # level-1 samples move with the fp32 coarse weights


@pytest.mark.gpu
@pytest.mark.parametrize('name', list(MG.CASES))
def test_hip_path_reproduces_golden(cuda, name):
    gold = _load(name)
    B, K, N, randomized, alpha, seed = MG.CASES[name]
    b, variables_cpu, noise = MG.build_case(name)
    utils.clear_gin()
    utils.parse_gin('MipNerfModel.num_samples = %d\nMipNerfModel.density_noise = 0.0\n'
                    'MipNerfModel.no_pose_opt = True\nMipNerfModel.no_yaw_opt = True\n'
                    'Config.randomized = %s\nConfig.rand_bkgd = False\nConfig.grad_max_norm = 1.0\n'
                    'Config.grad_max_val = 0.1\nConfig.tv_loss_mult = 0.0\n' % (N, randomized))
    config = utils.configured(utils.Config)
    model = utils.configured(obbpose_model.MipNerfModel)
    variables = variables_cpu.like(variables_cpu.flat.to(cuda))
    db = H.device_batch(b, cuda)
    nz = {k: v.to(cuda) for k, v in noise.items()}
    ret = model.apply(variables, 0, db['rays'], db['init'], db['ext'], b['ts'], randomized=randomized,
                      rand_bkgd=False, white_bkgd=False, alpha=alpha, noise=nz if randomized else None)
    far = 40.0
    tol = dict(rgb=2e-2, acc=2e-2, weights=2e-2, depth=2e-2 * far, t_vals=2e-2 * far)
    for lvl in range(2):
        for i, nm in enumerate(('rgb', 'depth', 'acc', 'weights', 't_vals')):
            k = 'l%d_%s' % (lvl, nm)
            np.testing.assert_allclose(ret[lvl][i].cpu().numpy(), gold[k], rtol=0, atol=tol[nm], err_msg=k)
    np.testing.assert_array_equal(ret[0][8].cpu().numpy(), gold['dyn_mask'])
    grad, _, _ = train_boxpose.loss_and_grad(model, config, 0, variables, db, 3.0, alpha, db['init'][0:1],
                                             noise=nz if randomized else None)
    raw_norm = float(grad.norm())
    state = train_boxpose.create_train_state(variables)
    _, stats, _, _ = train_boxpose.train_step(model, config, 0, state, db, 5e-4, 3.0, alpha, db['init'][0:1],
                                              noise=nz if randomized else None)
    for k in ('loss', 'losses', 'd_losses', 'n_losses', 'e_losses', 's_losses'):
        np.testing.assert_allclose(getattr(stats, k).cpu().numpy(), gold['stat_' + k], rtol=2e-2, atol=1e-6, err_msg=k)
    np.testing.assert_allclose(raw_norm, float(gold['grad_norm']), rtol=5e-2)     # un-clipped gradient norm
