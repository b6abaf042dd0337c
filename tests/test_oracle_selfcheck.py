"""Self-validation of the CPU oracle (JAX cannot run here, so nothing else can confirm it):
independent Monte-Carlo / closed-form / finite-difference checks of each restated formula
(SURVEY.md 8c).  All in float64 unless stated."""
import math

import numpy as np
import pytest
import torch

from durf_amd import synthetic
from oracle import durf_ref as R
from tests import helpers as H

F64 = torch.float64


def test_conical_frustum_moments_monte_carlo():
    """mip.py:117-124 vs 2M uniform samples of the frustum (density ~ t^2 along the axis)."""
    rs = np.random.RandomState(1)
    t0, t1, r = 1.3, 2.9, 0.07
    n = 2_000_000
    u = rs.uniform(0, 1, n)
    t = (t0 ** 3 + u * (t1 ** 3 - t0 ** 3)) ** (1 / 3)          # pdf ~ t^2
    rad = r * t * np.sqrt(rs.uniform(0, 1, n))
    ang = rs.uniform(0, 2 * np.pi, n)
    x = rad * np.cos(ang)
    d = torch.tensor([[0.0, 0.0, 1.0]], dtype=F64)
    mean, cov = R.conical_frustum_to_gaussian(d, torch.tensor([[t0]], dtype=F64), torch.tensor([[t1]], dtype=F64),
                                              torch.tensor([[r]], dtype=F64))
    assert abs(float(mean[0, 0, 2]) - t.mean()) < 2e-3 * t.mean()
    assert abs(float(cov[0, 0, 2, 2]) - t.var()) < 1e-2 * t.var()
    assert abs(float(cov[0, 0, 0, 0]) - x.var()) < 1e-2 * x.var()
    assert abs(float(cov[0, 0, 0, 1])) < 1e-12


def test_ipe_is_expected_sin_of_gaussian():
    """mip.py:67-73,273-282: exp(-var/2) sin(mu) == E[sin(z)], z ~ N(mu, var), low degrees."""
    rs = np.random.RandomState(2)
    mean = torch.tensor([[[0.3, -0.7, 1.1]]], dtype=F64)
    var = np.array([0.02, 0.05, 0.01])
    cov = torch.diag(torch.tensor(var, dtype=F64))[None, None]
    enc = R.integrated_pos_enc((mean, cov), 0, 10)[0, 0].numpy()
    z = mean.numpy().reshape(3) + rs.normal(size=(400000, 3)) * np.sqrt(var)
    for c in range(2):
        for deg in range(4):
            for j in range(3):
                want = np.sin(z[:, j] * 2 ** deg + c * math.pi / 2).mean()
                assert abs(enc[30 * c + 3 * deg + j] - want) < 6e-3


def test_only_diag_of_cov_matters():
    """integrated_pos_enc with a full covariance == with its diagonal."""
    g = torch.Generator().manual_seed(0)
    a = torch.randn(5, 7, 3, 3, generator=g, dtype=F64)
    cov = a @ a.transpose(-1, -2) * 1e-2
    mean = torch.randn(5, 7, 3, generator=g, dtype=F64)
    full = R.integrated_pos_enc((mean, cov), 0, 10)
    diag = R.integrated_pos_enc((mean, torch.diag_embed(torch.diagonal(cov, dim1=-2, dim2=-1))), 0, 10)
    torch.testing.assert_close(full, diag, rtol=0, atol=1e-14)
    mc, cc = R.new_space((mean, cov))
    v = torch.func.jvp(R.contract, (mean,), (torch.ones_like(mean),))[1]
    torch.testing.assert_close(torch.diagonal(cc, dim1=-2, dim2=-1),
                               torch.diagonal(cov, dim1=-2, dim2=-1) * v ** 2, rtol=1e-12, atol=1e-15)


def test_contraction_jvp_closed_form_and_finite_differences():
    """mip360.py:47-79 / SURVEY.md A.5: v = (2/n - 1/n^2) 1 + (2/n^4 - 2/n^3)(sum x) x for n > 0.1."""
    g = torch.Generator().manual_seed(3)
    for scale in (0.03, 0.3, 1.0, 5.0, 30.0):
        x = torch.randn(64, 1, 3, generator=g, dtype=F64) * scale
        y, v = torch.func.jvp(R.contract, (x,), (torch.ones_like(x),))
        n = x.norm(dim=-1, keepdim=True)
        closed = torch.where(n > 0.1, (2 / n - 1 / n ** 2) + (2 / n ** 4 - 2 / n ** 3) * x.sum(-1, keepdim=True) * x,
                             torch.ones_like(x))
        torch.testing.assert_close(v, closed, rtol=1e-9, atol=1e-12)
        h = 1e-6
        fd = (R.contract(x + h) - R.contract(x - h)) / (2 * h)
        ok = ((n - 0.1).abs() > 1e-3).expand_as(v)       # not across the branch
        torch.testing.assert_close(v[ok], fd[ok], rtol=1e-5, atol=1e-7)
        # the reference's quirk: threshold 0.1, so 0.1 < n < 0.5 flips the sign of x
        flip = ((n > 0.1) & (n < 0.5)).squeeze(-1)
        assert ((y * x).sum(-1)[flip] < 0).all()
        assert ((y * x).sum(-1)[(n > 0.5).squeeze(-1)] > 0).all()


def test_volumetric_rendering_constant_density():
    """mip.py:285-327: for constant sigma, w_n = e^{-sigma s_n} (1 - e^{-sigma delta_n}), acc = 1 - e^{-sigma L}."""
    N, sigma = 40, 0.37
    t = torch.sort(torch.rand(3, N + 1, dtype=F64) * 5, dim=-1).values
    d = torch.tensor([[0.0, 0.6, 0.8], [2.0, 0.0, 0.0], [1.0, 1.0, 1.0]], dtype=F64)
    dn = d.norm(dim=-1, keepdim=True)
    rgb = torch.full((3, N, 3), 0.25, dtype=F64)
    out = R.volumetric_rendering(rgb, torch.full((3, N, 1), sigma, dtype=F64), t, d, False, False)
    s = (t[:, :-1] - t[:, :1]) * dn
    delta = (t[:, 1:] - t[:, :-1]) * dn
    w = torch.exp(-sigma * s) * (1 - torch.exp(-sigma * delta))
    torch.testing.assert_close(out[3], w, rtol=1e-10, atol=1e-14)
    acc = 1 - torch.exp(-sigma * (t[:, -1] - t[:, 0]) * dn[:, 0])
    torch.testing.assert_close(out[2], acc, rtol=1e-10, atol=1e-14)
    torch.testing.assert_close(out[0], (0.25 * acc[:, None] + 0.5 * (1 - acc[:, None])).expand(3, 3), rtol=1e-10, atol=1e-14)
    # rand_bkgd adds randint(0,1) == 0 -> black background (mip.py:324)
    out_r = R.volumetric_rendering(rgb, torch.full((3, N, 1), sigma, dtype=F64), t, d, False, True)
    torch.testing.assert_close(out_r[0], (0.25 * acc[:, None]).expand(3, 3), rtol=1e-10, atol=1e-14)


def test_distortion_linear_time_identity():
    """sum_ij w_i w_j |s_i - s_j| == 2 sum_i w_i (s_i W_<i - WS_<i) for sorted s (used by the HIP kernel)."""
    g = torch.Generator().manual_seed(5)
    w = torch.rand(9, 33, generator=g, dtype=F64)
    s = torch.sort(torch.rand(9, 33, generator=g, dtype=F64) * 7, dim=-1).values
    quad = (w[:, :, None] * w[:, None, :] * (s[:, :, None] - s[:, None, :]).abs()).sum((-1, -2))
    W = torch.cumsum(w, -1) - w
    WS = torch.cumsum(w * s, -1) - w * s
    lin = 2 * (w * (s * W - WS)).sum(-1)
    torch.testing.assert_close(quad, lin, rtol=1e-12, atol=1e-14)


def test_barf_weight_index_is_feature_div_6():
    """mip.py:217-222: weight k multiplies features 6k..6k+5 of the 60, NOT frequency k."""
    mean = torch.tensor([[[0.2, -0.4, 0.9]]], dtype=F64)
    cov = torch.zeros(1, 1, 3, 3, dtype=F64)
    alpha = 3.5
    full = R.weighted_ipe((mean, cov), 0, 10, 10.0)[0, 0, 3:]
    got = R.weighted_ipe((mean, cov), 0, 10, alpha)[0, 0, 3:]
    w = R.barf_weights(alpha, 10, F64)
    torch.testing.assert_close(got, full * w.repeat_interleave(6), rtol=1e-14, atol=0)
    assert float(w[3]) == pytest.approx(0.5) and float(w[4]) == 0.0 and float(w[2]) == 1.0


def test_box_free_model_equals_missed_boxes():
    """A ray batch that misses every box renders exactly like the K=0 scene (masks are 0)."""
    b = synthetic.make_batch(48, 1, seed=4)
    ob = H.oracle_batch(b, F64)
    far_box = ob['init'].clone()
    far_box[:, :, :3] += 1000.0
    p1 = R.init_params(0, far_box, 1, dtype=F64)
    p0 = {'box_centers': far_box[:, :0], 'MLP_0': p1['MLP_0']}
    cfg = dict(num_samples=16)
    with torch.no_grad():
        r1 = R.model_apply(p1, ob['rays'], b['ts'], ob['ext'], False, False, False, 10.0, cfg=cfg)
        r0 = R.model_apply(p0, ob['rays'], b['ts'], ob['ext'][:0], False, False, False, 10.0, cfg=cfg)
    assert int(r1[0][8].sum()) == 0
    for lvl in range(2):
        torch.testing.assert_close(r1[lvl][0], r0[lvl][0], rtol=0, atol=1e-14)
        torch.testing.assert_close(r1[lvl][3], r0[lvl][3], rtol=0, atol=1e-14)


def test_loss_gradient_finite_differences():
    """autograd of the restated loss_fn vs central differences on a handful of parameters.
    One level only: with two, finite differences also see the (stop-gradient) dependence of
    the resampled t_vals on the coarse weights, which autodiff by construction does not."""
    b = synthetic.make_batch(24, 1, seed=6)
    ob = H.oracle_batch(b, F64)
    params = R.init_params(1, ob['init'], 1, dtype=F64)
    cfg = dict(R.CONFIG_DEFAULTS, randomized=False)
    mcfg = dict(num_samples=8, num_levels=1)
    prev = ob['init'][0:1]

    def loss_of(leaves):
        return R.loss_fn(R.set_leaves(params, leaves), ob, cfg, mcfg, 3.0, 10.0, prev)[0]
    leaves = [z.clone().requires_grad_(True) for z in R.params_leaves(params)]
    grads = torch.autograd.grad(loss_of(leaves), leaves, allow_unused=True)
    rs = np.random.RandomState(0)
    for li in (1, 2, 11, 17, 23, 24, 27, 40, 47):          # kernels/biases of both MLPs
        z = leaves[li].detach()
        idx = tuple(rs.randint(0, s) for s in z.shape)
        h = 1e-5
        vals = []
        for sgn in (+1, -1):
            mod = [x.detach().clone() for x in leaves]
            mod[li][idx] += sgn * h
            with torch.no_grad():
                vals.append(float(loss_of(mod)))
        fd = (vals[0] - vals[1]) / (2 * h)
        ga = float(grads[li][idx])
        assert abs(fd - ga) <= 1e-5 * max(1.0, abs(fd)) + 1e-9, (li, fd, ga)


def test_data_parallel_semantics():
    """pmap semantics (train_boxpose.py:253): per-shard losses normalised by LOCAL mask sums,
    gradients averaged over shards -- what each rank + one all-reduce computes."""
    b = synthetic.make_batch(32, 1, seed=8)
    ob = H.oracle_batch(b, F64)
    params = R.init_params(2, ob['init'], 1, dtype=F64)
    cfg = dict(R.CONFIG_DEFAULTS, randomized=False)
    mcfg = dict(num_samples=8)
    prev = ob['init'][0:1]

    def shard(i):
        s = slice(16 * i, 16 * (i + 1))
        out = dict(ob)
        out['rays'] = R.BoxRays(*[r[s] for r in ob['rays']])
        for k in ('pixels', 'depth', 'sky'):
            out[k] = ob[k][s]
        return out
    shards = [shard(0), shard(1)]
    _, _, st, grads = R.train_step(params, R.new_opt_state(params), None, cfg, mcfg, 5e-4, 3.0, 10.0, prev,
                                   shards=shards)
    g0 = R.train_step(params, R.new_opt_state(params), shards[0], cfg, mcfg, 5e-4, 3.0, 10.0, prev)[3]
    g1 = R.train_step(params, R.new_opt_state(params), shards[1], cfg, mcfg, 5e-4, 3.0, 10.0, prev)[3]
    for g, a, c in zip(grads, g0, g1):
        torch.testing.assert_close(g, (a + c) / 2, rtol=1e-12, atol=1e-15)


def test_adam_matches_torch_optim():
    """flax.optim.Adam (eps outside the sqrt, bias-corrected) == torch.optim.Adam's update rule."""
    g = torch.Generator().manual_seed(1)
    p = torch.randn(50, generator=g, dtype=F64)
    tp = p.clone().requires_grad_(True)
    opt = torch.optim.Adam([tp], lr=1e-2, betas=(0.9, 0.999), eps=1e-8)
    st = dict(step=0, m=[torch.zeros_like(p)], v=[torch.zeros_like(p)])
    leaves = [p]
    for _ in range(5):
        gr = torch.randn(50, generator=g, dtype=F64)
        leaves, st = R.adam_update(leaves, [gr], st, 1e-2)
        tp.grad = gr.clone()
        opt.step()
    torch.testing.assert_close(leaves[0], tp.detach(), rtol=1e-10, atol=1e-12)
