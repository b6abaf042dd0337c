"""tests/golden/ref_train_*.npz: what the REFERENCE's own train_step (train_boxpose.py:49-321) computes.

Made in the build container by running /root/reference/train_boxpose.py unmodified under the numpy stand-ins of
tests/ref_standin.py (tests/golden/make_ref_train_golden.py): every logged scalar of the reference's `loss_fn`, and the
derivative of the reference's loss (central differences of its own `loss_fn` closure with `lax.stop_gradient` replayed,
several step sizes) along seeded random directions AND along the gradient's own direction g / |g| per parameter group
and per Dense kernel of MLP_0 -- along those the reference's difference quotient is the NORM of the reference's gradient,
so a zeroed, scaled or mis-directed gradient fails (the negative tests below plant exactly those).  The vectors travel
to machines without /root/reference:
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
            _, grads = G.oracle(params, ob, config, model_cfg, c, prev, noise)
            quot = G.reference_derivatives(ref_loss, params, tree, G.gradient_directions(params, grads)[0])
            np.testing.assert_array_equal(quot, gold['grad_dir_derivatives'])
            quot = G.reference_derivatives(ref_loss, params, tree, G.layer_directions(params, grads)[0], G.LAYER_STEPS)
            np.testing.assert_array_equal(quot, gold['layer_dir_derivatives'])
    finally:
        ref_standin.unload()


def _best(q, d):
    return min(q, key=lambda x: abs(x - d))


def check_gradient(case, gold, params, b, c, oracle_grads, grad_flat, tol, what):
    """Hold a flat gradient (layout = the leaves of R.params_leaves back to back) to the reference's difference quotients.
    tol: norm-wise relative tolerance of the gradient under test.  Raises AssertionError with the measured numbers."""
    grad_flat = grad_flat.double().cpu()
    own = G.owners(params)
    sizes = [z.numel() for z in R.params_leaves(params)]
    flat_of = lambda vs: torch.cat([x.reshape(-1) for x in vs])
    # (1) the gradient's own direction per parameter group: the reference's quotient is |g|
    gdirs, gnorms = G.gradient_directions(params, oracle_grads)
    np.testing.assert_allclose(gnorms, gold['grad_norms'], rtol=1e-9, atol=1e-15,
                               err_msg='the regenerated directions are not the ones the fixture was made with')
    report = []
    for (target, vs), q, nrm in zip(gdirs, gold['grad_dir_derivatives'], gnorms):
        if nrm <= 1e-6:          # a group no ray reaches (or frozen): the gradient under test must vanish there too
            seg = flat_of([torch.ones_like(v) if o == target else torch.zeros_like(v) for v, o in zip(vs, own)]) != 0
            assert float(grad_flat[seg].abs().max()) <= 1e-6, '%s %s: %s has no gradient in the reference' % (case, what, target)
            continue
        d = float((grad_flat * flat_of(vs)).sum())
        ref = _best(q, d)
        report.append('%s %.3e' % (target, abs(d - ref) / abs(ref)))
        assert abs(d - ref) <= tol * abs(ref) + 1e-7, \
            '%s %s: |g| of %s along the reference gradient: reference %s, got %.8g (tol %g)' % (case, what, target, q, d, tol)
    # (2) ... per Dense kernel of MLP_0: the split of the gradient over the layers (2 x tol: a single layer's
    # norm-wise error may exceed the whole group's)
    ldirs, lnorms = G.layer_directions(params, oracle_grads)
    np.testing.assert_allclose(lnorms, gold['layer_grad_norms'], rtol=1e-9, atol=1e-15)
    for (target, vs), q in zip(ldirs, gold['layer_dir_derivatives']):
        d = float((grad_flat * flat_of(vs)).sum())
        ref = _best(q, d)
        assert abs(d - ref) <= 2 * tol * abs(ref) + 1e-7, \
            '%s %s: |g| of %s: reference %s, got %.8g (tol %g)' % (case, what, target, q, d, 2 * tol)
    # (3) seeded random directions: for an error e with |e| <= tol |g|, <e, v> over a Gaussian v of n entries has standard
    # deviation |e| |v| / sqrt(n): gate at 4 sigma (tol |g| |v| itself would be hundreds of times the derivative)
    for (target, vs), q in zip(G.directions(params, b, c['seed']), gold['derivatives']):
        v = flat_of(vs)
        sel = v != 0
        n = int(sel.sum())
        d = float((grad_flat * v).sum())
        g_norm = float(flat_of([g if o == target else torch.zeros_like(g) for g, o in zip(oracle_grads, own)]).norm())
        bound = 4.0 * tol * g_norm * float(v.norm()) / max(n, 1) ** 0.5
        ref = _best(q, d)
        assert abs(ref - d) <= bound + 1e-5 * abs(ref) + 2e-7, \
            '%s %s along a random direction of %s: reference %s, got %.8g (bound %.3g)' % (case, what, target, q, d, bound)
    return report


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
    flat = torch.cat([g.reshape(-1) for g in grads])
    # float64 autograd against float64 central differences: 1e-4 covers the quotients' own O(h) kink error (measured <= 8e-5)
    check_gradient(case, gold, params, b, c, grads, flat, 1e-4, 'oracle')


@pytest.mark.parametrize('tamper', ['zero MLP_0', 'scale MLP_0 by 1.2', 'zero one layer', 'zero a BoxMLP', 'flip the pose gradient'])
def test_the_gradient_gate_rejects_a_wrong_gradient(tamper):
    """the gate must FAIL for a gradient that is zero, scaled or mis-split -- at the bf16 tolerance of the HIP path (5e-2),
    the loosest it is ever used with"""
    case = 'K3_pose_opt_rand'
    gold = _load(case)
    c, b, ob, params, prev, noise, config, model_cfg = _inputs(case, gold)
    _, grads = G.oracle(params, ob, config, model_cfg, c, prev, noise)
    own = G.owners(params)
    bad = [g.clone() for g in grads]
    if tamper == 'zero MLP_0':
        bad = [torch.zeros_like(g) if o == 'MLP_0' else g for g, o in zip(bad, own)]
    elif tamper == 'scale MLP_0 by 1.2':
        bad = [1.2 * g if o == 'MLP_0' else g for g, o in zip(bad, own)]
    elif tamper == 'zero one layer':
        bad[own.index('MLP_0') + 2 * 9].zero_()             # Dense_9's kernel: 0.13 of |g| 3.6 -- invisible to the group norm
    elif tamper == 'zero a BoxMLP':
        bad = [torch.zeros_like(g) if o == 'BoxMLP_1' else g for g, o in zip(bad, own)]
    else:
        bad[0] = -bad[0]
    check_gradient(case, gold, params, b, c, grads, torch.cat([g.reshape(-1) for g in grads]), 5e-2, 'untampered')
    with pytest.raises(AssertionError):
        check_gradient(case, gold, params, b, c, grads, torch.cat([g.reshape(-1) for g in bad]), 5e-2, tamper)


def _hip_setup(cuda, case, precision):
    gold = _load(case)
    c, b, ob, params, prev, noise, config, model_cfg = _inputs(case, gold)
    utils.clear_gin()
    lines = ['MipNerfModel.%s = %r' % (k, v) for k, v in model_cfg.items()] + ['MipNerfModel.mlp_precision = "%s"' % precision]
    utils.parse_gin('\n'.join(lines).replace("'", '"') + '\n')
    model = utils.configured(obbpose_model.MipNerfModel)
    model._check()                           # every case of HIP_CASES is a configuration the product runs: no skip
    db = H.device_batch(b, cuda)
    _, variables = obbpose_model.construct_mipnerf(c['seed'], db, device=cuda)
    leaves = R.params_leaves(params)
    variables.flat.copy_(torch.cat([z.reshape(-1) for z in leaves]).float().to(cuda))
    conf = utils.Config(**{k: v for k, v in config.items() if k in utils.Config.__dataclass_fields__})
    nz = {k: v.float().to(cuda) for k, v in noise.items()} if config['randomized'] else None
    prev_d = prev.float().to(cuda)
    _, oracle_grads = G.oracle(params, ob, config, model_cfg, c, prev, noise)
    return gold, c, b, params, model, conf, variables, db, nz, prev_d, oracle_grads


@pytest.mark.gpu
@pytest.mark.parametrize('precision', ['f32', 'bf16'])
@pytest.mark.parametrize('case', G.HIP_CASES)
def test_hip_train_step_reproduces_the_reference(cuda, case, precision):
    gold, c, b, params, model, conf, variables, db, nz, prev_d, oracle_grads = _hip_setup(cuda, case, precision)
    grad, _, _ = train_boxpose.loss_and_grad(model, conf, 0, variables, db, c['eps'], c['alpha'], prev_d, noise=nz)
    state = train_boxpose.create_train_state(variables)
    _, stats, _, _ = train_boxpose.train_step(model, conf, 0, state, db, 5e-4, c['eps'], c['alpha'], prev_d, noise=nz)
    rtol = 2e-2 if precision == 'bf16' else 2e-4           # BF16 / F32 loss terms
    for k in ('loss', 'losses', 'd_losses', 'n_losses', 'e_losses', 's_losses', 'distr_losses', 'tv_losses', 'offsets'):
        want, got = gold[k], getattr(stats, k).double().cpu().numpy()
        fin = np.isfinite(want)
        np.testing.assert_allclose(got[fin], want[fin], rtol=rtol, atol=1e-6, err_msg='%s %s %s' % (case, precision, k))
    tol = 5e-2 if precision == 'bf16' else 5e-3            # norm-wise, as every gradient gate of the HIP path
    report = check_gradient(case, gold, params, b, c, oracle_grads, grad, tol, 'HIP ' + precision)
    print('%s %s: relative error of |g| along the reference gradient: %s' % (case, precision, ', '.join(report)))


@pytest.mark.gpu
@pytest.mark.parametrize('tamper', ['zero', 'scale'])
def test_the_gate_fails_when_the_hip_gradient_of_mlp0_is_zeroed_or_scaled(cuda, tamper):
    case, precision = 'K3_pose_opt_rand', 'bf16'
    gold, c, b, params, model, conf, variables, db, nz, prev_d, oracle_grads = _hip_setup(cuda, case, precision)
    grad, _, _ = train_boxpose.loss_and_grad(model, conf, 0, variables, db, c['eps'], c['alpha'], prev_d, noise=nz)
    lay = variables.layout
    o0, n0 = lay.mlp_off['MLP_0'], lay.mlp_size[obbpose_model.W_BKGD]
    bad = grad.clone()
    bad[o0:o0 + n0] *= 0.0 if tamper == 'zero' else 1.2
    with pytest.raises(AssertionError):
        check_gradient(case, gold, params, b, c, oracle_grads, bad, 5e-2, 'HIP bf16, MLP_0 %s' % tamper)
