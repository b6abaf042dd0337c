"""End-to-end forward parity of MipNerfModel.apply (HIP path through the C ABI) against the
oracle's model_apply on identical ray batches and parameters.

Tolerances (bf16 MFMA MLP, fp32 everywhere else):
  * vs the oracle with bf16-rounded GEMM operands (same arithmetic): rgb/acc/weights <= 2e-3,
    depth <= 2e-3 * far, level-1 t_vals <= 2e-3 * far;
  * vs the plain fp32 oracle (what an fp32 JAX run computes): rgb <= 2e-2 (SURVEY.md 8c BF16 mode).
"""
import pytest
import torch

from durf_amd import obbpose_model, synthetic, utils
from oracle import durf_ref as R
from tests import helpers as H

pytestmark = pytest.mark.gpu


def _run(cuda, B, K, N, randomized, seed, alpha=10.0, far=40.0, knobs=None, near=None):
    utils.clear_gin()
    utils.parse_gin('MipNerfModel.num_samples = %d\nMipNerfModel.density_noise = 0.0\n'
                    'MipNerfModel.no_pose_opt = True\nMipNerfModel.no_yaw_opt = True\n' % N +
                    ''.join('MipNerfModel.%s = %r\n' % kv for kv in (knobs or {}).items()))
    b = synthetic.make_batch(B, K, seed=seed, far=far, allow_multi_hit=True)
    if near is not None:
        b['rays']['near'][:] = near
    ob, db = H.oracle_batch(b), H.device_batch(b, cuda)
    model, variables = obbpose_model.construct_mipnerf(seed, db, device=cuda)
    # non-zero biases so bias packing is exercised
    g = torch.Generator().manual_seed(seed)
    for name in variables.layout.mlp_names():
        for i in range(12):
            bias = variables['params'][name]['Dense_%d' % i]['bias']
            bias.copy_(((torch.rand(bias.shape, generator=g) - 0.5) * 0.1).to(cuda))
    noise_c = dict(t_rand=torch.rand(B, N + 1, generator=g), u_rand=torch.rand(B, N + 1, generator=g))
    noise_d = {k: v.to(cuda) for k, v in noise_c.items()}
    ret = model.apply(variables, 0, db['rays'], db['init'], db['ext'], b['ts'], randomized=randomized,
                      rand_bkgd=False, white_bkgd=False, alpha=alpha, noise=noise_d if randomized else None)
    torch.cuda.synchronize()
    params = H.oracle_params_from_variables(variables)
    mcfg = dict(num_samples=N, **{k: v for k, v in (knobs or {}).items() if k != 'mlp_precision'})
    with torch.no_grad():
        ref_bf = R.model_apply(params, ob['rays'], b['ts'], ob['ext'], randomized, False, False, alpha,
                               noise=noise_c if randomized else None, cfg=mcfg, mlp_hook=R.mlp_apply_bf16)
        ref_32 = R.model_apply(params, ob['rays'], b['ts'], ob['ext'], randomized, False, False, alpha,
                               noise=noise_c if randomized else None, cfg=mcfg)
    b['_multi'] = (ref_32[0][8].reshape(-1) > 1).numpy().astype('int64')
    return b, ret, ref_bf, ref_32


@pytest.mark.parametrize('K,N,randomized', [(0, 64, False), (1, 32, False), (3, 64, True), (8, 32, False),
                                            (1, 96, True), (2, 256, False)])     # odd multiple of 32, and the maximum N
def test_forward_parity(cuda, K, N, randomized):
    far = 40.0
    b, ret, ref_bf, ref_32 = _run(cuda, 256, K, N, randomized, seed=21 + K, far=far)
    assert len(ret) == 2
    # Rays that hit two boxes at once are garbage-in/garbage-out in the reference
    # (obbpose_model.py:120-122 sums object-frame origins and the bkgd mask becomes -1, so
    # variances go negative and the encoding overflows): outside the tolerance domain, only
    # checked for "non-finite on both sides".
    single = torch.tensor(b['_multi'] == 0)
    for lvl in range(2):
        got, rb, r32 = ret[lvl], ref_bf[lvl], ref_32[lvl]
        names = ['rgb', 'depth', 'acc', 'weights', 't_vals']
        tols = [2e-3, 2e-3 * far, 2e-3, 2e-3, 2e-3 * far]
        for i, (nm, tol) in enumerate(zip(names, tols)):
            torch.testing.assert_close(got[i].cpu()[single], rb[i][single], rtol=0, atol=tol,
                                       msg=lambda m: '%s l%d: %s' % (nm, lvl, m))
        assert (got[0].cpu() - r32[0]).abs()[single].max() < 2e-2, 'rgb vs fp32 oracle'
        assert torch.isfinite(got[0].cpu()[single]).all()
        if (~single).any():
            assert (~torch.isfinite(got[0].cpu()[~single]).all(-1) == ~torch.isfinite(rb[0][~single]).all(-1)).all()
        assert torch.equal(got[8].cpu().long().reshape(-1), rb[8].reshape(-1)), 'dyn_mask'
        torch.testing.assert_close(got[9].cpu(), rb[9], rtol=1e-6, atol=1e-5)
        torch.testing.assert_close(got[7][0].cpu(), rb[7][0])
    if K > 0:
        assert 0.02 < b['hit_fraction'] < 0.3


@pytest.mark.parametrize('knobs', [dict(lindisp=True), dict(disable_integration=True), dict(dynamics=False),
                                   dict(contraction=False), dict(ray_shape='cylinder')])
def test_forward_knobs(cuda, knobs):
    """gin knobs off their shipped values: lindisp (mip.py:354-356; near > 0 so 1/t is finite),
    disable_integration (obbpose_model.py:164-165), dynamics=False (:167,232,257-260), contraction=False."""
    far = 40.0
    b, ret, ref_bf, _ = _run(cuda, 256, 2, 32, False, seed=31, far=far, knobs=knobs,
                             near=0.05 if knobs.get('lindisp') else None)
    single = torch.tensor(b['_multi'] == 0)
    scale = 1.0 / 0.05 if knobs.get('lindisp') else far           # lindisp: t runs over [1/far, 1/near]
    if knobs.get('lindisp'):
        # lindisp makes t_vals DEcreasing (1/near -> 1/far, mip.py:354-356): deltas, alphas and weights go negative
        # and the colours blow up on both sides, so the sample positions of level 0 are the meaningful comparison.
        for i in (4, 5, 6):
            torch.testing.assert_close(ret[0][i].cpu(), ref_bf[0][i], rtol=1e-6, atol=1e-6)
        assert bool((ret[0][4][:, 1:] < ret[0][4][:, :-1]).all())
        return
    for lvl in range(2):
        got, rb = ret[lvl], ref_bf[lvl]
        for i, tol in ((0, 3e-3), (1, 3e-3 * scale), (2, 3e-3), (3, 3e-3), (4, 2e-3 * scale)):
            torch.testing.assert_close(got[i].cpu()[single], rb[i][single], rtol=0, atol=tol,
                                       msg=lambda m: 'output %d l%d: %s' % (i, lvl, m))
        assert torch.equal(got[8].cpu().long().reshape(-1), rb[8].reshape(-1)), 'dyn_mask'


def test_density_noise(cuda):
    """density_noise > 0 with randomized=True (obbpose_model.py:236-240; the class default is 0.1, both shipped
    gin files set 0): the same N(0,1) draws on both sides."""
    B, K, N, far = 256, 1, 32, 40.0
    utils.clear_gin()
    utils.parse_gin('MipNerfModel.num_samples = %d\nMipNerfModel.density_noise = 0.1\n'
                    'MipNerfModel.no_pose_opt = True\nMipNerfModel.no_yaw_opt = True\n' % N)
    b = synthetic.make_batch(B, K, seed=12, far=far)
    ob, db = H.oracle_batch(b), H.device_batch(b, cuda)
    model, variables = obbpose_model.construct_mipnerf(3, db, device=cuda)
    g = torch.Generator().manual_seed(5)
    noise_c = dict(t_rand=torch.rand(B, N + 1, generator=g), u_rand=torch.rand(B, N + 1, generator=g),
                   density=[torch.randn(B, N, 1, generator=g) for _ in range(2)])
    noise_d = dict(t_rand=noise_c['t_rand'].to(cuda), u_rand=noise_c['u_rand'].to(cuda),
                   density=[d.to(cuda) for d in noise_c['density']])
    ret = model.apply(variables, 0, db['rays'], db['init'], db['ext'], b['ts'], randomized=True, rand_bkgd=False,
                      white_bkgd=False, alpha=10.0, noise=noise_d)
    params = H.oracle_params_from_variables(variables)
    with torch.no_grad():
        ref = R.model_apply(params, ob['rays'], b['ts'], ob['ext'], True, False, False, 10.0, noise=noise_c,
                            cfg=dict(num_samples=N, density_noise=0.1), mlp_hook=R.mlp_apply_bf16)
        ref0 = R.model_apply(params, ob['rays'], b['ts'], ob['ext'], True, False, False, 10.0, noise=noise_c,
                             cfg=dict(num_samples=N, density_noise=0.0), mlp_hook=R.mlp_apply_bf16)
    for lvl in range(2):
        for i, tol in ((0, 3e-3), (2, 3e-3), (3, 3e-3), (4, 2e-3 * far)):
            torch.testing.assert_close(ret[lvl][i].cpu(), ref[lvl][i], rtol=0, atol=tol)
    assert (ref[1][3] - ref0[1][3]).abs().max() > 1e-3, 'the noise must matter in this test'
    # without explicit draws the model makes its own (and still runs)
    ret2 = model.apply(variables, 7, db['rays'], db['init'], db['ext'], b['ts'], randomized=True, rand_bkgd=False,
                       white_bkgd=False, alpha=10.0)
    assert torch.isfinite(ret2[1][0]).all()


def test_forward_alpha_ramp(cuda):
    """BARF coarse-to-fine mask active (alpha = 4.5): the feature//6 weight quirk is live."""
    b, ret, ref_bf, _ = _run(cuda, 256, 2, 32, False, seed=5, alpha=4.5)
    single = torch.tensor(b['_multi'] == 0)
    for lvl in range(2):
        torch.testing.assert_close(ret[lvl][0].cpu()[single], ref_bf[lvl][0][single], rtol=0, atol=2e-3)


def test_render_image(cuda):
    utils.clear_gin()
    utils.parse_gin('MipNerfModel.num_samples = 32\nMipNerfModel.density_noise = 0.0\n')
    b = synthetic.make_batch(64, 1, seed=3)
    db = H.device_batch(b, cuda)
    model, variables = obbpose_model.construct_mipnerf(3, db, device=cuda)
    img = synthetic.make_image_rays(12, 20)
    rays = utils.BoxRays(**{k: torch.tensor(v, device=cuda) for k, v in img.items()})

    def render_fn(rng, batch):
        return model.apply(variables, rng, batch['rays'], batch['init'], batch['ext'], batch['ts'],
                           randomized=False, rand_bkgd=False, white_bkgd=False, alpha=batch['alpha'])
    rgb, dist, acc = obbpose_model.render_image(render_fn, rays, db['init'], db['ext'], b['ts'], 0, 10.0, chunk=100)
    assert rgb.shape == (12, 20, 3) and dist.shape == (12, 20) and acc.shape == (12, 20)
    rgb2, _, _ = obbpose_model.render_image(render_fn, rays, db['init'], db['ext'], b['ts'], 0, 10.0, chunk=240)
    torch.testing.assert_close(rgb, rgb2, rtol=0, atol=1e-6)     # chunking must not change results
    params = H.oracle_params_from_variables(variables)
    orays = R.BoxRays(**{k: torch.tensor(v) for k, v in img.items()})
    ref = R.render_image(params, orays, b['ts'], torch.tensor(b['ext']), 10.0, chunk=100,
                         cfg=dict(num_samples=32), mlp_hook=R.mlp_apply_bf16)
    torch.testing.assert_close(rgb.cpu(), ref[0], rtol=0, atol=2e-3)
    torch.testing.assert_close(acc.cpu(), ref[2], rtol=0, atol=2e-3)


def _random_config(seed):
    import random
    r = random.Random(seed)
    K = r.choice([0, 1, 2, 3, 5, 8])
    N = r.choice([32, 64, 96, 128])
    B = r.choice([37, 64, 141, 256, 300])
    randomized = r.random() < 0.5
    far = r.choice([20.0, 40.0, 200.0])
    alpha = r.choice([10.0, 3.3, 0.0])
    knobs = {}
    for name, val in (('disable_integration', True), ('contraction', False), ('ray_shape', 'cylinder'),
                      ('dynamics', False), ('resample_padding', 0.05)):
        if r.random() < 0.3:
            knobs[name] = val
    return K, N, B, randomized, far, alpha, knobs


def test_the_soak_seed_behind_the_level_1_gate_in_exact_fp32(cuda):
    """Seed 1047 of the forward sweep is the case that set the level-1 gate of the bf16 path to 5e-3 (a 68-seed soak run:
    one level-1 weight of 24 576 at 3.14e-3 against the bf16-rounded oracle; attributed to the bf16 noise of level 0's
    weights moving the resampled positions).  The same configuration (K = 0, N = 96, stratified sampling, cylinder
    rays, positional encoding WITHOUT integration, resample padding 0.05) in MipNerfModel.mlp_precision = 'f32' -- same
    kernels around the MLPs, same resampler -- against the plain fp32 oracle: level 0 meets the F32_EXACT tolerances
    (rgb / acc / weights 1e-5, depth and t_vals 1e-4 far); level 1 sits on positions resampled from fp32 weights that
    differ in the last bits, and without the integrated encoding's damping the 2^9-frequency features turn a 1e-6 far
    shift of a sample into ~1e-4 of colour: measured 6.3e-5, gated at 2e-4 -- fifty times below the bf16 path's
    3.14e-3 in the same place, i.e. that deviation is arithmetic, not logic."""
    seed = 1047
    K, N, B, randomized, far, alpha, knobs = _random_config(seed)
    b, ret, ref_bf, ref_32 = _run(cuda, B, K, N, randomized, seed=seed, alpha=alpha, far=far,
                                  knobs=dict(knobs, mlp_precision='f32'))
    single = torch.tensor(b['_multi'] == 0)
    assert single.any()
    worst = {}
    for lvl in range(2):
        got, want = ret[lvl], ref_32[lvl]
        for i, nm in enumerate(('rgb', 'depth', 'acc', 'weights', 't_vals')):
            worst[(lvl, nm)] = float((got[i].cpu()[single] - want[i][single]).abs().max())
    print('seed 1047 in exact fp32, max abs deviation from the fp32 oracle:', {k: '%.2e' % v for k, v in worst.items()})
    for nm, tol in (('rgb', 1e-5), ('acc', 1e-5), ('weights', 1e-5), ('depth', 1e-4 * far), ('t_vals', 1e-4 * far)):
        assert worst[(0, nm)] <= tol, (0, nm, worst[(0, nm)])
    for nm, tol in (('rgb', 2e-4), ('acc', 2e-4), ('weights', 2e-4), ('depth', 1e-4 * far), ('t_vals', 1e-4 * far)):
        assert worst[(1, nm)] <= tol, (1, nm, worst[(1, nm)])


@pytest.mark.parametrize('seed', list(range(100, 108)) + H.extra_fuzz_seeds('FWD'))
def test_forward_random_configurations(cuda, seed):
    """Seeded sweep over combinations the hand-picked cases above do not pair up: ragged ray counts (partial 256-sample
    blocks and compaction rounds), K in 0..8, N in {32, 64, 96, 128}, randomized sampling, and the gin knobs two at a
    time.  Same tolerances as test_forward_parity / test_forward_knobs, single-hit rays only."""
    K, N, B, randomized, far, alpha, knobs = _random_config(seed)
    b, ret, ref_bf, ref_32 = _run(cuda, B, K, N, randomized, seed=seed, alpha=alpha, far=far, knobs=knobs)
    single = torch.tensor(b['_multi'] == 0)
    assert single.any()
    for lvl in range(2):
        got, rb = ret[lvl], ref_bf[lvl]
        # level 1 sits on resampled positions, which carry the bf16 noise of level 0's weights: 5e-3 there (a 68-seed soak
        # run had one weight of 24 576 at 3.14e-3), the hand-picked cases' 3e-3 at level 0
        t3 = 3e-3 if lvl == 0 else 5e-3
        for i, tol in ((0, t3), (1, t3 * far), (2, t3), (3, t3), (4, 2e-3 * far)):
            torch.testing.assert_close(got[i].cpu()[single], rb[i][single], rtol=0, atol=tol,
                                       msg=lambda m: 'seed %d K=%d N=%d B=%d %s: output %d l%d: %s' % (
                                           seed, K, N, B, knobs, i, lvl, m))
        assert (got[0].cpu() - ref_32[lvl][0]).abs()[single].max() < 2e-2, 'rgb vs fp32 oracle'
        assert torch.equal(got[8].cpu().long().reshape(-1), rb[8].reshape(-1)), 'dyn_mask'
    if K > 0:       # the de-duplicated background evaluation renders bit-identically to the plain path
        from durf_amd import ops
        keep = ops.DEDUP_HIT_RAYS
        try:
            ops.DEDUP_HIT_RAYS = not keep
            _, ret2, _, _ = _run(cuda, B, K, N, randomized, seed=seed, alpha=alpha, far=far, knobs=knobs)
        finally:
            ops.DEDUP_HIT_RAYS = keep
        for lvl in range(2):
            for i in range(5):
                a, c = ret[lvl][i].cpu(), ret2[lvl][i].cpu()
                assert torch.equal(a[single], c[single]), 'seed %d: output %d l%d differs with de-duplication toggled' % (seed, i, lvl)
