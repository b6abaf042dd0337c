"""tests/golden/ref_train_*.npz: what the REFERENCE's own train_step (train_boxpose.py:49-321) computes.

Made in the build container by running /root/reference/train_boxpose.py unmodified under the numpy stand-ins of
tests/ref_standin.py (tests/golden/make_ref_train_golden.py): every logged scalar of the reference's `loss_fn`, and the
derivative of the reference's loss along seeded random directions (central differences of its own `loss_fn` closure with
`lax.stop_gradient` replayed, three step sizes).  The vectors travel to machines without /root/reference:
  * the float64 oracle must reproduce the scalars (1e-6) and, as <autograd gradient, v>, the derivatives (1e-5 at the best
    step size) -- CPU, everywhere;
  * the HIP train step must reproduce the scalars and the derivatives within its precision's tolerances -- GPU;
  * where the reference tree is present the generator must reproduce the committed vectors."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
import make_ref_train_golden as G  # noqa: E402
from durf_amd import obbpose_model, train_boxpose, utils  # noqa: E402
from oracle import durf_ref as R  # noqa: E402
from tests import helpers as H  # noqa: E402
from tests import ref_standin  # noqa: E402


def _load(case):
    return np.load(os.path.join(ROOT, 'tests', 'golden', 'ref_train_' + case + '.npz'))


def _inputs(case, gold):
    c, b, ob, params, prev, noise, config, model_cfg = G.setup(case)
    flat = torch.cat([z.reshape(-1) for z in R.params_leaves(params)])
    np.testing.assert_allclose([float(flat.sum()), float((flat * flat).sum())], gold['param_checksum'], rtol=1e-12)
    np.testing.assert_array_equal(noise['t_rand'].numpy(), gold['t_rand'])
    np.testing.assert_array_equal(noise['u_rand'].numpy(), gold['u_rand'])
    return c, b, ob, params, prev, noise, config, model_cfg


@pytest.mark.skipif(not ref_standin.available(), reason='reference tree not present')
def test_generator_reproduces_committed_fixtures():
    ref = ref_standin.load(train=True)
    try:
        for case in G.CASES:
            gold = _load(case)
            _, stats, pose, ref_loss, (c, b, ob, params, prev, noise, config, model_cfg, tree) = G.run_reference(ref, case)
            for k in G.SCALARS:
                np.testing.assert_array_equal(np.asarray(getattr(stats, k), dtype=np.float64), gold[k], err_msg=k)
            quot = G.reference_derivatives(ref_loss, params, tree, G.directions(params, b, c['seed']))
            np.testing.assert_array_equal(quot, gold['derivatives'])
    finally:
        ref_standin.unload()


@pytest.mark.parametrize('case', sorted(G.CASES))
def test_oracle_reproduces_the_reference_train_step(case):
    gold = _load(case)
    c, b, ob, params, prev, noise, config, model_cfg = _inputs(case, gold)
    S, grads = G.oracle(params, ob, config, model_cfg, c, prev, noise)
    for k in G.SCALARS:
        want, got = gold[k], S[k].detach().numpy()
        fin = np.isfinite(want)
        assert (np.isfinite(got) == fin).all(), k
        np.testing.assert_allclose(got[fin], want[fin], rtol=0, atol=1e-6 * max(1.0, np.abs(want[fin]).max() if fin.any() else 1.0),
                                   err_msg=case + ' ' + k)
    for (target, vs), q in zip(G.directions(params, b, c['seed']), gold['derivatives']):
        d = float(sum((gr * v).sum() for gr, v in zip(grads, vs)))
        best = min(q, key=lambda x: abs(x - d))
        assert abs(best - d) <= 1e-5 * abs(d) + 2e-7, '%s along %s: reference %s, oracle %.8g' % (case, target, q, d)


@pytest.mark.gpu
@pytest.mark.parametrize('precision', ['f32', 'bf16'])
@pytest.mark.parametrize('case', sorted(G.CASES))
def test_hip_train_step_reproduces_the_reference(cuda, case, precision):
    gold = _load(case)
    c, b, ob, params, prev, noise, config, model_cfg = _inputs(case, gold)
    utils.clear_gin()
    lines = ['MipNerfModel.%s = %r' % (k, v) for k, v in model_cfg.items()] + ['MipNerfModel.mlp_precision = "%s"' % precision]
    utils.parse_gin('\n'.join(lines).replace("'", '"') + '\n')
    try:
        model = utils.configured(obbpose_model.MipNerfModel)
        model._check()
        db = H.device_batch(b, cuda)
        _, variables = obbpose_model.construct_mipnerf(c['seed'], db, device=cuda)
    except NotImplementedError as e:           # knob combinations outside the shipped configs (MipNerfModel._check)
        pytest.skip(str(e))
    leaves = R.params_leaves(params)
    variables.flat.copy_(torch.cat([z.reshape(-1) for z in leaves]).float().to(cuda))
    conf = utils.Config(**{k: v for k, v in config.items() if k in utils.Config.__dataclass_fields__})
    nz = {k: v.float().to(cuda) for k, v in noise.items()} if config['randomized'] else None
    prev_d = prev.float().to(cuda)
    grad, _, _ = train_boxpose.loss_and_grad(model, conf, 0, variables, db, c['eps'], c['alpha'], prev_d, noise=nz)
    grad = grad.double().cpu()
    state = train_boxpose.create_train_state(variables)
    _, stats, _, _ = train_boxpose.train_step(model, conf, 0, state, db, 5e-4, c['eps'], c['alpha'], prev_d, noise=nz)
    rtol = 2e-2 if precision == 'bf16' else 2e-4           # BF16 / F32 loss terms
    for k in ('loss', 'losses', 'd_losses', 'n_losses', 'e_losses', 's_losses', 'distr_losses', 'tv_losses', 'offsets'):
        want, got = gold[k], getattr(stats, k).double().cpu().numpy()
        fin = np.isfinite(want)
        np.testing.assert_allclose(got[fin], want[fin], rtol=rtol, atol=1e-6, err_msg='%s %s %s' % (case, precision, k))
    tol = 5e-2 if precision == 'bf16' else 5e-3            # norm-wise, as every gradient gate of the HIP path
    off = 0
    sizes = [z.numel() for z in leaves]
    for (target, vs), q in zip(G.directions(params, b, c['seed']), gold['derivatives']):
        v = torch.cat([x.reshape(-1) for x in vs])
        d = float((grad * v).sum())
        sel = v != 0
        bound = tol * float(grad[sel].norm()) * float(v.norm())
        best = min(q, key=lambda x: abs(x - d))
        assert abs(best - d) <= bound + 1e-7, '%s %s along %s: reference %s, HIP %.8g (bound %.3g)' % (case, precision, target, q, d, bound)
