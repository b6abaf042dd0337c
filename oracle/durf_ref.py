"""CPU ORACLE for the durf ray pipeline -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A restatement, in plain torch-CPU tensor arithmetic (dtype selectable: float64 =
"truth", float32 = "reference-precision twin"), of the hot path of FelTris/durf:
`MipNerfModel.__call__` + `train_step`.  Every function cites the reference
file:line it follows (paths relative to the reference repository root).

PINNING.  The reference is 100 % Python/JAX; jax/flax/gin are not installable here
or on the GPU box, and it ships no golden vectors for this path (its only test
file, internal/math_test.py, covers internal/math.py).  What pins this restatement:
(a) THE REFERENCE'S OWN SOURCE, RUN HERE: /root/reference/internal/obbpose_model.py
    (with mip, mip360, math, box_helpers, utils) and /root/reference/train_boxpose.py
    are imported unmodified under numpy-backed stand-ins for jax / flax.linen / gin
    (tests/ref_standin.py, float64, PRNG draws replayed) and compared with this
    module: MipNerfModel.__call__ (every entry of every level's 10-tuple: 4e-10 at
    level 0), render_image, every logged scalar of train_step's loss_fn, its
    gradient post-processing entry by entry, and the gradient itself (central
    differences of the reference's loss_fn closure with stop_gradient replayed
    against this module's autograd: 7-8 digits) -- tests/test_reference_*crosscheck.py;
    outputs of those runs are committed as tests/golden/ref_model_*.npz /
    ref_train_*.npz with their generators and checked wherever the tests run;
(b) the framework-agnostic assertions of internal/math_test.py re-expressed in
    tests/test_oracle_math.py;
(c) self-consistency checks (Monte-Carlo frustum moments, sampled E[sin], closed-form
    contraction JVP, finite-difference gradients) in tests/test_oracle_selfcheck.py.
STILL UNPINNED (parity "unpinned" in the sense of the build rules): the third-party
layer the stand-ins replace -- flax.linen.Dense, jax.nn.initializers.glorot_uniform,
flax.optim.Adam, jnp.nan_to_num, jnp.remainder, threefry, XLA's transcendentals --
restated from their public definitions (jax>=0.2.12, flax 0.2.2-0.5.x per
requirements_jax.txt:2-4).  A stand-in fixes the reference's source text as
executed, not JAX's arithmetic.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  The product package (durf_amd/) never does.
"""
import math as _pm
from collections import namedtuple

import torch

# ----------------------------------------------------------------------------
# types (internal/utils.py:77-86)
# ----------------------------------------------------------------------------
Rays = namedtuple('Rays', ('origins', 'directions', 'viewdirs', 'radii',
                           'lossmult', 'near', 'far', 'delta'))
BoxRays = namedtuple('BoxRays', ('origins', 'directions', 'viewdirs', 'radii',
                                 'lossmult', 'near', 'far'))



def batch_from_numpy(b, dt=torch.float32):
    """numpy batch (durf_amd.synthetic.make_batch schema) -> the oracle's batch."""
    import numpy as np
    rays = BoxRays(**{k: torch.tensor(v, dtype=dt) for k, v in b['rays'].items()})
    out = {k: (torch.tensor(v, dtype=dt) if isinstance(v, np.ndarray) else v)
           for k, v in b.items() if k != 'rays'}
    out['rays'] = rays
    return out


F32_EPS = 1.1920928955078125e-07  # jnp.finfo('float32').eps
F32_MAX = 3.4028234663852886e+38


def _nan_to_num(x):
    """jnp.nan_to_num(x): nan->0, +inf->max, -inf->min of x.dtype."""
    return torch.nan_to_num(x)


# ----------------------------------------------------------------------------
# internal/math.py
# ----------------------------------------------------------------------------
def safe_norm(x):
    """internal/math.py:27-32."""
    norm_sqs = torch.sum(x ** 2, dim=-1, keepdim=True)
    norm_save = torch.where(norm_sqs < 1e-12, torch.full_like(norm_sqs, 1e-12), norm_sqs)
    return torch.sqrt(norm_save)


def safe_trig_helper(x, fn, t=100 * _pm.pi):
    """internal/math.py:35-36.  `x % t` is jnp.remainder (sign of divisor)."""
    return fn(torch.where(torch.abs(x) < t, x, torch.remainder(x, t)))


def safe_cos(x):
    """internal/math.py:39-41."""
    return safe_trig_helper(x, torch.cos)


def safe_sin(x):
    """internal/math.py:44-46."""
    return safe_trig_helper(x, torch.sin)


def mse_to_psnr(mse):
    """internal/math.py:49-51."""
    return -10. / _pm.log(10.) * torch.log(torch.as_tensor(mse))


def psnr_to_mse(psnr):
    """internal/math.py:54-56."""
    return torch.exp(-0.1 * _pm.log(10.) * torch.as_tensor(psnr))


def learning_rate_decay(step, lr_init, lr_final, max_steps, lr_delay_steps=0,
                        lr_delay_mult=1):
    """internal/math.py:156-190 (host-side scalar schedule)."""
    if lr_delay_steps > 0:
        delay_rate = lr_delay_mult + (1 - lr_delay_mult) * _pm.sin(
            0.5 * _pm.pi * min(max(step / lr_delay_steps, 0), 1))
    else:
        delay_rate = 1.
    t = min(max(step / max_steps, 0), 1)
    log_lerp = _pm.exp(_pm.log(lr_init) * (1 - t) + _pm.log(lr_final) * t)
    return delay_rate * log_lerp


def freq_alpha_rate(step, alpha_init, alpha_final, alpha_delay_steps,
                    alpha_max_steps):
    """internal/math.py:193-219."""
    if step < alpha_delay_steps:
        return alpha_init
    elif step < alpha_max_steps:
        return (step - alpha_delay_steps) / (alpha_max_steps - alpha_delay_steps) * alpha_final
    else:
        return alpha_final


def sorted_piecewise_constant_pdf(u_rand, bins, weights, num_samples, randomized):
    """internal/math.py:222-284.

    `u_rand` replaces the jax PRNG key: a [..., num_samples] tensor of U[0,1)
    draws (jax.random.uniform(key, shape, maxval=s-eps) == U[0,1)*(s-eps)); it
    is ignored when randomized is False.
    """
    eps = 1e-5
    weight_sum = torch.sum(weights, dim=-1, keepdim=True)
    padding = torch.clamp(eps - weight_sum, min=0)
    weights = weights + padding / weights.shape[-1]
    weight_sum = weight_sum + padding

    pdf = weights / weight_sum
    cdf = torch.clamp(torch.cumsum(pdf[..., :-1], dim=-1), max=1)
    shp = list(cdf.shape[:-1]) + [1]
    cdf = torch.cat([torch.zeros(shp, dtype=cdf.dtype), cdf,
                     torch.ones(shp, dtype=cdf.dtype)], dim=-1)

    dt = bins.dtype
    if randomized:
        s = 1 / num_samples
        u = torch.arange(num_samples, dtype=dt) * s
        u = u + u_rand.to(dt) * (s - F32_EPS)
        u = torch.clamp(u, max=1. - F32_EPS)
    else:
        # jnp.linspace(0., 1. - eps32, num) in float32: start*(1-step) + stop*step with
        # step = iota/(num-1), last element := stop (jax/_src/numpy/lax_numpy.py linspace).
        stop = torch.tensor(1. - F32_EPS, dtype=torch.float32)
        step = torch.arange(num_samples, dtype=torch.float32) / (num_samples - 1)
        u = stop * step
        u[-1] = stop
        u = u.to(dt)
        u = u.expand(list(cdf.shape[:-1]) + [num_samples])

    mask = u[..., None, :] >= cdf[..., :, None]

    def find_interval(x):
        x0 = torch.max(torch.where(mask, x[..., None], x[..., :1, None]), dim=-2).values
        x1 = torch.min(torch.where(~mask, x[..., None], x[..., -1:, None]), dim=-2).values
        return x0, x1

    bins_g0, bins_g1 = find_interval(bins)
    cdf_g0, cdf_g1 = find_interval(cdf)

    t = torch.clamp(_nan_to_num((u - cdf_g0) / (cdf_g1 - cdf_g0)), 0, 1)
    return bins_g0 + t * (bins_g1 - bins_g0)


# ----------------------------------------------------------------------------
# internal/box_helpers.py
# ----------------------------------------------------------------------------
def aa2matrix(angles):
    """internal/box_helpers.py:148-167 (Rodrigues, angles [K,3] -> [K,3,3])."""
    n_frames = angles.shape[0]
    zero = torch.zeros_like(angles[:, 0])
    skew_v0 = torch.stack([zero, -angles[:, 2], angles[:, 1]], dim=-1)
    skew_v1 = torch.stack([angles[:, 2], zero, -angles[:, 0]], dim=-1)
    skew_v2 = torch.stack([-angles[:, 1], angles[:, 0], zero], dim=-1)
    skew_r = torch.stack([skew_v0, skew_v1, skew_v2], dim=-2)
    angles_norm = safe_norm(angles) + 1e-12
    eye = torch.eye(3, dtype=angles.dtype).expand(n_frames, 3, 3)
    R = eye + (torch.sin(angles_norm) / angles_norm)[..., None] * skew_r + \
        ((1 - torch.cos(angles_norm)) / angles_norm ** 2)[..., None] * torch.matmul(skew_r, skew_r)
    return R


def rotate_matrix(p, m):
    """internal/box_helpers.py:170-181."""
    if p.dim() < 4:
        p = p[..., None, :]
    return torch.matmul(m[..., None, :, :], p[..., None]).reshape(p.shape)


def world2object_rpy(pts, dirs, pose, rot):
    """internal/box_helpers.py:286-341, forward branch, dim=None."""
    t_w_o = rotate_matrix(-pose, rot)
    n_obj = rot.shape[1]
    pts_w = pts[:, None, :].expand(-1, n_obj, -1)
    dirs_w = dirs[:, None, :].expand(-1, n_obj, -1)
    pts_o = rotate_matrix(pts_w, rot) + t_w_o
    dirs_o = rotate_matrix(dirs_w, rot)
    dirs_o = dirs_o / torch.sqrt(torch.sum(dirs_o ** 2, dim=3))[..., None, :]
    return pts_o.squeeze(-2), dirs_o.squeeze(-2)


def ray_box_intersection(ray_o, ray_d, aabb_min, aabb_max):
    """internal/box_helpers.py:59-106."""
    inv_d = torch.reciprocal(ray_d)
    t_min = (aabb_min - ray_o) * inv_d
    t_max = (aabb_max - ray_o) * inv_d
    t0 = torch.minimum(t_min, t_max)
    t1 = torch.maximum(t_min, t_max)
    t_near = torch.maximum(torch.maximum(t0[..., 0], t0[..., 1]), t0[..., 2])
    t_far = torch.minimum(torch.minimum(t1[..., 0], t1[..., 1]), t1[..., 2])
    intersection_map = torch.where(t_far > t_near, 1, 0)
    positive_far = torch.where(t_far * intersection_map > 0, 1, 0)
    intersection_map = intersection_map * positive_far
    z_ray_in = t_near * intersection_map
    z_ray_out = t_far * intersection_map
    return z_ray_in, z_ray_out, intersection_map


# ----------------------------------------------------------------------------
# internal/mip.py
# ----------------------------------------------------------------------------
def pos_enc(x, min_deg, max_deg, append_identity=True):
    """internal/mip.py:36-45."""
    scales = torch.tensor([2 ** i for i in range(min_deg, max_deg)], dtype=x.dtype)
    xb = (x[..., None, :] * scales[:, None]).reshape(list(x.shape[:-1]) + [-1])
    four_feat = torch.sin(torch.cat([xb, xb + 0.5 * _pm.pi], dim=-1))
    if append_identity:
        return torch.cat([x, four_feat], dim=-1)
    return four_feat


def expected_sin(x, x_var):
    """internal/mip.py:67-73 (only the mean is consumed by the model)."""
    return torch.exp(-0.5 * x_var) * safe_sin(x)


def lift_gaussian(d, t_mean, t_var, r_var):
    """internal/mip.py:76-96, diag=False (full covariance, as cast_rays calls it)."""
    mean = d[..., None, :] * t_mean[..., None]
    d_mag_sq = torch.clamp(torch.sum(d ** 2, dim=-1, keepdim=True), min=1e-10)
    d_outer = d[..., :, None] * d[..., None, :]
    eye = torch.eye(d.shape[-1], dtype=d.dtype)
    null_outer = eye - d[..., :, None] * (d / d_mag_sq)[..., None, :]
    t_cov = t_var[..., None, None] * d_outer[..., None, :, :]
    xy_cov = r_var[..., None, None] * null_outer[..., None, :, :]
    return mean, t_cov + xy_cov


def conical_frustum_to_gaussian(d, t0, t1, base_radius):
    """internal/mip.py:99-130, stable=True."""
    mu = (t0 + t1) / 2
    hw = (t1 - t0) / 2
    t_mean = mu + (2 * mu * hw ** 2) / (3 * mu ** 2 + hw ** 2)
    t_var = (hw ** 2) / 3 - (4 / 15) * ((hw ** 4 * (12 * mu ** 2 - hw ** 2)) /
                                        (3 * mu ** 2 + hw ** 2) ** 2)
    r_var = base_radius ** 2 * ((mu ** 2) / 4 + (5 / 12) * hw ** 2 - 4 / 15 *
                                (hw ** 4) / (3 * mu ** 2 + hw ** 2))
    return lift_gaussian(d, t_mean, t_var, r_var)


def cylinder_to_gaussian(d, t0, t1, radius):
    """internal/mip.py:133-152."""
    t_mean = (t0 + t1) / 2
    r_var = radius ** 2 / 4
    t_var = (t1 - t0) ** 2 / 12
    return lift_gaussian(d, t_mean, t_var, r_var)


def cast_rays(t_vals, origins, directions, radii, ray_shape='cone'):
    """internal/mip.py:155-179, diag=False."""
    t0 = t_vals[..., :-1]
    t1 = t_vals[..., 1:]
    if ray_shape == 'cone':
        means, covs = conical_frustum_to_gaussian(directions, t0, t1, radii)
    elif ray_shape == 'cylinder':
        means, covs = cylinder_to_gaussian(directions, t0, t1, radii)
    else:
        raise ValueError(ray_shape)
    means = means + origins[..., None, :]
    return means, covs


def _ipe_core(x, x_cov, min_deg, max_deg):
    """internal/mip.py:205-216 / :248-282 (shared body of both encoders)."""
    num_dims = x.shape[-1]
    basis = torch.cat([2 ** i * torch.eye(num_dims, dtype=x.dtype)
                       for i in range(min_deg, max_deg)], dim=1)
    y = torch.matmul(x, basis)
    y_var = torch.sum(torch.matmul(x_cov, basis) * basis, dim=-2)
    return expected_sin(torch.cat([y, y + 0.5 * _pm.pi], dim=-1),
                        torch.cat([y_var] * 2, dim=-1))


def barf_weights(alpha, max_deg, dtype):
    """internal/mip.py:217-218: w_k = (1 - cos(pi * clip(alpha - k, 0, 1))) / 2."""
    k = torch.arange(max_deg, dtype=dtype)
    a = torch.as_tensor(alpha, dtype=dtype)
    return (1 - torch.cos(torch.clamp(a - k, 0, 1) * _pm.pi)) / 2


def weighted_ipe(x_coord, min_deg, max_deg, alpha):
    """internal/mip.py:182-223, diag=False.  NB the weight index is feature//6."""
    x, x_cov = x_coord
    encoding = _ipe_core(x, x_cov, min_deg, max_deg)
    weight = barf_weights(alpha, max_deg, x.dtype)
    B, N, _ = x.shape
    weight = weight[:, None].expand(max_deg, 6).reshape(-1).expand(B, N, -1)
    return torch.cat([x, weight * encoding], dim=-1)


def integrated_pos_enc(x_coord, min_deg, max_deg):
    """internal/mip.py:226-282, diag=False."""
    x, x_cov = x_coord
    return _ipe_core(x, x_cov, min_deg, max_deg)


def volumetric_rendering(rgb, density, t_vals, dirs, white_bkgd, rand_bkgd):
    """internal/mip.py:285-327.  rand_bkgd adds randint(key,(1,3),0,1) == 0."""
    t_mids = 0.5 * (t_vals[..., :-1] + t_vals[..., 1:])
    t_dists = t_vals[..., 1:] - t_vals[..., :-1]
    delta = t_dists * torch.sqrt(torch.sum(dirs[..., None, :] ** 2, dim=-1))
    density_delta = density[..., 0] * delta
    alpha = 1 - torch.exp(-density_delta)
    trans = torch.exp(-torch.cat([
        torch.zeros_like(density_delta[..., :1]),
        torch.cumsum(density_delta[..., :-1], dim=-1)], dim=-1))
    weights = _nan_to_num(alpha * trans)
    comp_rgb = (weights[..., None] * rgb).sum(dim=-2)
    acc = weights.sum(dim=-1)
    depth = (weights * t_mids).sum(dim=-1)
    if white_bkgd:
        comp_rgb = comp_rgb + (1. - acc[..., None])
    if rand_bkgd:
        comp_rgb = comp_rgb + 0.0 * (1.0 - acc[..., None])
    elif not white_bkgd:
        comp_rgb = comp_rgb + 0.5 * (1.0 - acc[..., None])
    return comp_rgb, depth, acc, weights, t_vals, t_mids, t_dists


def sample_along_rays(t_rand, origins, directions, radii, num_samples, near, far,
                      randomized, lindisp=False, ray_shape='cone'):
    """internal/mip.py:330-370.  `t_rand` [B,N+1] U[0,1) replaces the PRNG key."""
    batch_size = origins.shape[0]
    dt = origins.dtype
    t_vals = torch.linspace(0., 1., num_samples + 1, dtype=torch.float32).to(dt)
    if lindisp:
        t_vals = 1. / (near * (1. - t_vals) + far * t_vals)
    else:
        t_vals = near * (1. - t_vals) + far * t_vals
    if randomized:
        mids = 0.5 * (t_vals[..., 1:] + t_vals[..., :-1])
        upper = torch.cat([mids, t_vals[..., -1:]], -1)
        lower = torch.cat([t_vals[..., :1], mids], -1)
        t_vals = lower + (upper - lower) * t_rand.to(dt)
    else:
        t_vals = t_vals.expand(batch_size, num_samples + 1)
    means, covs = cast_rays(t_vals, origins, directions, radii, ray_shape)
    return t_vals, (means, covs)


def resample_along_rays(u_rand, origins, directions, radii, t_vals, weights,
                        randomized, stop_grad, resample_padding, ray_shape='cone'):
    """internal/mip.py:373-416."""
    weights_pad = torch.cat([weights[..., :1], weights, weights[..., -1:]], dim=-1)
    weights_max = torch.maximum(weights_pad[..., :-1], weights_pad[..., 1:])
    weights_blur = 0.5 * (weights_max[..., :-1] + weights_max[..., 1:])
    weights = weights_blur + resample_padding
    new_t_vals = sorted_piecewise_constant_pdf(u_rand, t_vals, weights,
                                               t_vals.shape[-1], randomized)
    if stop_grad:
        new_t_vals = new_t_vals.detach()
    means, covs = cast_rays(new_t_vals, origins, directions, radii, ray_shape)
    return new_t_vals, (means, covs)


# ----------------------------------------------------------------------------
# internal/mip360.py:47-79
# ----------------------------------------------------------------------------
def contract(x):
    """internal/mip360.py:47-60 (threshold 0.1; factor negative on (0.1,0.5))."""
    x_norm = safe_norm(x)
    x_smaller = (x_norm <= 0.1).to(x.dtype)
    x_larger = (x_norm > 0.1).to(x.dtype)
    x_contract = (2.0 - _nan_to_num(1.0 / x_norm)) * _nan_to_num(x / x_norm)
    return x_smaller * x + x_larger * x_contract


def new_space(samples):
    """internal/mip360.py:63-79: mean -> contract(mean); cov -> push-forward with
    the JVP of `contract` along (1,1,1), exactly as the reference composes it."""
    mean, cov = samples
    tangent = torch.ones_like(mean)
    meanc, jt = torch.func.jvp(contract, (mean,), (tangent,))
    eye = torch.eye(3, dtype=mean.dtype).expand(cov.shape)
    je = jt[:, :, :, None] * eye
    covc = torch.matmul(je, torch.matmul(cov, je).transpose(-1, -2)).transpose(-1, -2)
    return meanc, covc


# ----------------------------------------------------------------------------
# internal/obbpose_model.py
# ----------------------------------------------------------------------------
MLP_BKGD = dict(net_depth=8, net_width=256, net_depth_condition=1,
                net_width_condition=128, skip_layer=4, num_rgb_channels=3,
                num_density_channels=1)           # obbpose_model.py:296-303
MLP_BOX = dict(MLP_BKGD, net_width=128)          # obbpose_model.py:360-367


def mlp_layer_shapes(in_dim, view_dim, cfg):
    """(fan_in, fan_out) of Dense_0..Dense_11 in flax creation order
    (obbpose_model.py:329-353): trunk x8, density head, bottleneck, view layer, rgb."""
    W = cfg['net_width']
    shapes = []
    d = in_dim
    for i in range(cfg['net_depth']):
        shapes.append((d, W))
        d = W
        if i % cfg['skip_layer'] == 0 and i > 0:
            d = W + in_dim
    shapes.append((d, cfg['num_density_channels']))
    if view_dim is None:                                                  # no condition: Dense_9 is the rgb head
        shapes.append((d, cfg['num_rgb_channels']))
        return shapes
    shapes.append((d, W))
    dv = W + view_dim
    for _ in range(cfg['net_depth_condition']):
        shapes.append((dv, cfg['net_width_condition']))
        dv = cfg['net_width_condition']
    shapes.append((dv, cfg['num_rgb_channels']))
    return shapes


def mlp_apply(params, x, condition, cfg):
    """obbpose_model.py:305-354 (MLP) == :369-418 (BoxMLP).  params: list of
    (kernel[in,out], bias[out]) in Dense_0..Dense_11 order."""
    feature_dim = x.shape[-1]
    num_samples = x.shape[1]
    x = x.reshape(-1, feature_dim)
    inputs = x
    li = 0
    for i in range(cfg['net_depth']):
        k, b = params[li]; li += 1
        x = torch.relu(x @ k + b)
        if i % cfg['skip_layer'] == 0 and i > 0:
            x = torch.cat([x, inputs], dim=-1)
    k, b = params[li]; li += 1
    raw_density = (x @ k + b).reshape(-1, num_samples, cfg['num_density_channels'])
    if condition is not None:                                             # :336-350 (None: the rgb head reads the trunk)
        k, b = params[li]; li += 1
        bottleneck = x @ k + b
        cond = condition[:, None, :].expand(-1, num_samples, -1).reshape(-1, condition.shape[-1])
        x = torch.cat([bottleneck, cond], dim=-1)
        for _ in range(cfg['net_depth_condition']):
            k, b = params[li]; li += 1
            x = torch.relu(x @ k + b)
    k, b = params[li]; li += 1
    raw_rgb = (x @ k + b).reshape(-1, num_samples, cfg['num_rgb_channels'])
    return raw_rgb, raw_density


MODEL_DEFAULTS = dict(          # obbpose_model.py:45-66 overridden by configs/*.gin:40-54
    num_samples=128, num_levels=2, resample_padding=0.01, stop_level_grad=True,
    use_viewdirs=True, lindisp=False, ray_shape='cone', min_deg_point=0,
    max_deg_point=10, deg_view=4, density_noise=0.0, density_bias=-1.,
    disable_integration=False, contraction=True, dynamics=True,
    no_pose_opt=True, no_yaw_opt=True)


def init_params(seed, init, K, dtype=torch.float32, in_bkgd=60, in_obj=63, view_dim=27):
    """construct_mipnerf (obbpose_model.py:264-291): box_centers := init verbatim
    (:35-39,88); Dense kernels glorot-uniform, biases zero (flax defaults).  The
    threefry stream is not reproducible -> seeded torch generator instead."""
    g = torch.Generator().manual_seed(seed)

    def dense(fi, fo):
        lim = _pm.sqrt(6.0 / (fi + fo))
        k = (torch.rand(fi, fo, generator=g, dtype=torch.float64) * 2 - 1) * lim
        return [k.to(dtype), torch.zeros(fo, dtype=dtype)]

    params = {'box_centers': init.clone().to(dtype),
              'MLP_0': [dense(*s) for s in mlp_layer_shapes(in_bkgd, view_dim, MLP_BKGD)]}
    for k in range(K):
        params['BoxMLP_%d' % k] = [dense(*s) for s in mlp_layer_shapes(in_obj, view_dim, MLP_BOX)]
    return params


def model_apply(params, rays, ts, ext, randomized, rand_bkgd, white_bkgd, alpha,
                noise=None, cfg=None, mlp_hook=None):
    """MipNerfModel.__call__ (obbpose_model.py:68-261).

    params['box_centers'] is the learnable [T,K,6]; `ts` a python int.
    noise = dict(t_rand=[B,N+1], u_rand=[B,N+1]) replaces the PRNG when
    randomized.  mlp_hook(params_list, x, cond, cfg) optionally replaces
    mlp_apply (used to emulate bf16 operand rounding).
    Returns list[num_levels] of the reference's 10-tuples.
    """
    c = dict(MODEL_DEFAULTS)
    if cfg:
        c.update(cfg)
    assert c['density_noise'] == 0.0 or not randomized or (noise is not None and 'density' in noise), \
        'density noise: pass noise["density"] = per-level N(0,1) draws [B,N,1] (they replace the jax PRNG key)'
    mlp = mlp_hook or mlp_apply
    pose_offsets = params['box_centers']
    K = pose_offsets.shape[1]
    origins, dirs = rays.origins, rays.directions
    B = origins.shape[0]
    dt = origins.dtype

    box_pose = pose_offsets[ts, :, :3].expand(B, K, 3)                    # :99
    if c['no_pose_opt']:
        box_pose = box_pose.detach()
    box_rot = pose_offsets[ts, :, 3:]                                     # :102
    if c['no_yaw_opt']:
        box_rot = box_rot.detach()
    ret_pose = [pose_offsets[ts, :, :3] if not c['no_pose_opt'] else pose_offsets[ts, :, :3].detach(),
                box_rot[0] if K > 0 else box_rot.new_zeros(3)]
    if K > 0:
        box_mat = aa2matrix(box_rot).expand(B, K, 3, 3)                   # :105-106
        box_dims = ext.expand(B, K, 3)
        origins_o, dirs_o = world2object_rpy(origins, dirs, box_pose, box_mat)  # :110
        zi, zo, intersection = ray_box_intersection(origins_o, dirs_o, -box_dims, box_dims)
        intersection = intersection.detach()
        inter_f = intersection.to(dt)
        bkgd_mask = (intersection.sum(dim=-1) == 0).to(dt)               # :115
        obj_pts = origins_o * inter_f[..., None]
        obj_dirs = dirs_o * inter_f[..., None]
        origins_s = obj_pts.sum(dim=-2) + bkgd_mask[..., None] * origins  # :121
        dirs_s = obj_dirs.sum(dim=-2) + bkgd_mask[..., None] * dirs       # :122
        zo_ret = (inter_f * zo).sum(dim=-1)                               # :131
    else:
        intersection = torch.zeros(B, 0, dtype=torch.int64)
        origins_s, dirs_s = origins, dirs
        zo_ret = torch.zeros(B, dtype=dt)

    # :193,:222 (use_viewdirs=False: the background MLP gets no condition, :230-231; the object loop of the reference reads
    # `viewdirs_enc` regardless, :199 -- undefined then, so the knob only exists for a model without boxes / dynamics)
    viewdirs_enc = pos_enc(rays.viewdirs, 0, c['deg_view'], True) if c['use_viewdirs'] else None
    if viewdirs_enc is None and c['dynamics'] and K > 0:
        raise NameError("name 'viewdirs_enc' is not defined")             # what the reference does at :199

    ret = []
    t_vals = weights = None
    for i_level in range(c['num_levels']):
        if i_level == 0:
            t_vals, samples = sample_along_rays(
                None if noise is None else noise['t_rand'], origins_s, dirs_s,
                rays.radii, c['num_samples'], rays.near, rays.far, randomized,
                c['lindisp'], c['ray_shape'])
        else:
            t_vals, samples = resample_along_rays(
                None if noise is None else noise['u_rand'], origins_s, dirs_s,
                rays.radii, t_vals, weights, randomized, c['stop_level_grad'],
                c['resample_padding'], c['ray_shape'])
        if c['disable_integration']:
            samples = (samples[0], torch.zeros_like(samples[1]))

        Bn, N, _ = samples[0].shape
        raw_rgbs = torch.zeros(Bn, N, 3, dtype=dt)
        raw_densities = torch.zeros(Bn, N, 1, dtype=dt)
        masks_sum = torch.zeros(Bn, N, 1, dtype=dt)
        ret_masks = torch.zeros(Bn, 1, dtype=torch.int64)
        if not c['dynamics']:                                             # :257-260: mask from the hit test only
            ret_masks = intersection.sum(dim=-1, keepdim=True)
        for i in range(K if c['dynamics'] else 0):                        # :167,:174-201
            mask_i = intersection[:, i].reshape(-1, 1)
            ret_masks = ret_masks + mask_i
            mask = mask_i.to(dt)[:, None, :].expand(Bn, N, 1)
            masks_sum = masks_sum + mask
            obj_mean = mask * samples[0]
            obj_var = mask[..., None] * samples[1]
            enc = weighted_ipe((obj_mean, obj_var), c['min_deg_point'],
                               c['max_deg_point'], alpha)
            obj_rgb, obj_density = mlp(params['BoxMLP_%d' % i], enc, viewdirs_enc, MLP_BOX)
            raw_rgbs = raw_rgbs + mask * obj_rgb
            raw_densities = raw_densities + mask * obj_density
        if c['dynamics']:
            bkgd_mask_s = (1 - masks_sum).detach()                        # :205-206
            samples = (bkgd_mask_s * samples[0], bkgd_mask_s[..., None] * samples[1])

        if c['contraction']:
            samples = new_space(samples)                                  # :212-213
        samples_enc = integrated_pos_enc(samples, c['min_deg_point'], c['max_deg_point'])
        raw_rgb, raw_density = mlp(params['MLP_0'], samples_enc, viewdirs_enc, MLP_BKGD)
        raw_rgb = raw_rgb + raw_rgbs                                      # :232-234
        raw_density = raw_density + raw_densities

        if randomized and c['density_noise'] > 0:                         # :236-240
            raw_density = raw_density + c['density_noise'] * noise['density'][i_level].to(dt)
        rgb = torch.sigmoid(raw_rgb)                                      # :243
        density = torch.nn.functional.softplus(raw_density + c['density_bias'])  # :245
        comp_rgb, distance, acc, weights, t_vals, t_mids, t_dists = volumetric_rendering(
            rgb, density, t_vals, dirs_s, white_bkgd, rand_bkgd)
        ret.append((comp_rgb, distance, acc, weights, t_vals, t_mids, t_dists,
                    ret_pose, ret_masks, zo_ret))
    return ret


# ----------------------------------------------------------------------------
# train_boxpose.py:49-321
# ----------------------------------------------------------------------------
CONFIG_DEFAULTS = dict(     # internal/utils.py:93-144 overridden by configs/waymo.gin:1-39
    randomized=True, rand_bkgd=False, white_bkgd=False, disable_multiscale_loss=False,
    box_loss_mult=0, tv_loss_mult=0.0, depth_loss_mult=1e-4, near_loss_mult=1e-2,
    empty_loss_mult=1.0, sky_loss_mult=1.0, coarse_loss_mult=0.1,
    weight_decay_mult=0.0, grad_max_norm=1.0, grad_max_val=0.1,
    lr_init=5e-4, lr_final=5e-6, lr_delay_steps=2500, lr_delay_mult=0.01,
    eps_init=3.0, eps_final=0.2, eps_max_steps=200000, eps_delay_steps=0,
    alpha_init=10., alpha_final=10., alpha_delay_steps=0, alpha_max_steps=1,
    max_steps=200000, batch_size=512)


def params_leaves(params):
    """Flat list of leaves in a fixed order (box_centers, MLP_0, BoxMLP_0..)."""
    leaves = [params['box_centers']]
    names = ['MLP_0'] + sorted([k for k in params if k.startswith('BoxMLP_')],
                               key=lambda s: int(s.split('_')[1]))
    for n in names:
        for k, b in params[n]:
            leaves += [k, b]
    return leaves


def level_terms(level_ret, batch, config, eps, mask, depth_mask, sky_mask):
    """Body of the per-level loop of loss_fn (train_boxpose.py:123-192) for one level's
    model outputs.  Returns ({term: scalar}, updated depth_mask) -- depth_mask accumulates
    box_loss_mult*dyn*box across levels (:140)."""
    (rgb, depth, _, weights, tvals, tmids, t_dists, off, dyn_mask, zo) = level_ret
    dt = rgb.dtype
    gt_depth = batch['depth'].reshape(-1)
    gt_sky = batch['sky'].reshape(-1)
    pixels = batch['pixels'][..., :3]
    out = {}
    box_mask = (gt_depth < zo).to(dt)                                 # :138
    dyn_f = dyn_mask.to(dt)
    depth_mask = depth_mask + config['box_loss_mult'] * dyn_f.reshape(-1) * box_mask  # :140

    tv = tvals[:, :-1]                                                # :145
    Sij = torch.abs(tmids[:, None, :] - tmids[:, :, None])            # :146-150
    Wij = weights[..., :, None] * weights[..., None, :]
    term1 = (Wij * Sij).sum()
    term2 = (1 / 3) * (weights ** 2 * t_dists).sum()
    out['distr_losses'] = term1 + term2

    depth_t = batch['depth'].reshape(-1, 1).expand(tv.shape)          # :155
    sigma = (eps / 3.) ** 2
    mask_near = ((tv > (depth_t - eps)) & (tv < (depth_t + eps))).to(dt)
    mask_near = mask_near * depth_mask.reshape(tv.shape[0], -1)
    mask_empty = (tv > (depth_t + eps)).to(dt)
    mask_empty = mask_empty * depth_mask.reshape(tv.shape[0], -1)
    dist = mask_near * (tv - depth_t)
    distr = 1.0 / (sigma * _pm.sqrt(2 * _pm.pi)) * torch.exp(-(dist ** 2 / (2 * sigma ** 2)))
    distr = distr / distr.max()
    distr = distr * mask_near
    dm_sum = torch.clamp(depth_mask.sum(), min=1.0)
    out['n_losses'] = ((mask_near * weights - distr) ** 2).sum() / dm_sum   # :166
    out['e_losses'] = ((mask_empty * weights) ** 2).sum() / dm_sum           # :167
    out['d_losses'] = (depth_mask * (depth - gt_depth) ** 2).sum() / dm_sum  # :174-175
    sky_depth = sky_mask * (1.0 - (1.0 / torch.clamp(sky_mask * depth, min=1.0)))  # :186
    out['s_losses'] = (sky_mask * (sky_depth - gt_sky) ** 2).sum() / torch.clamp(sky_mask.sum(), min=1.0)
    out['losses'] = ((mask + config['box_loss_mult'] * dyn_f * box_mask[..., None])
                     * (rgb - pixels) ** 2).sum() / mask.sum()   # :191
    out['obj_losses'] = (dyn_f * (rgb - pixels) ** 2).sum() / dyn_f.sum()   # :192
    return out, depth_mask


def total_loss(S, c, weight_l2=0.0):
    """train_boxpose.py:211-220 from stacked per-level terms."""
    loss = c['coarse_loss_mult'] * S['losses'][:-1].sum() + S['losses'][-1] + weight_l2
    loss = loss + c['sky_loss_mult'] * S['s_losses'][:-1].sum() + 10.0 * c['sky_loss_mult'] * S['s_losses'][-1]
    loss = loss + c['depth_loss_mult'] * S['d_losses'][-1] + 0.1 * c['depth_loss_mult'] * S['d_losses'][:-1].sum()
    loss = loss + c['near_loss_mult'] * S['n_losses'][-1] + 0.1 * c['near_loss_mult'] * S['n_losses'][:-1].sum()
    loss = loss + c['empty_loss_mult'] * S['e_losses'][-1] + 0.1 * c['empty_loss_mult'] * S['e_losses'][:-1].sum()
    loss = loss + c['tv_loss_mult'] * S['tv_losses'][-1] + 0.1 * c['tv_loss_mult'] * S['tv_losses'][:-1].sum()
    loss = loss + 0.000001 * S['distr_losses'][-1] + 0.000001 * S['distr_losses'][:-1].sum()
    return loss


def loss_fn(params, batch, config, model_cfg, eps, alpha, prev, noise=None, mlp_hook=None):
    """train_boxpose.py:67-249.  Returns (loss, stats dict, model ret)."""
    leaves = params_leaves(params)
    dt = leaves[0].dtype
    weight_l2 = config['weight_decay_mult'] * (
        sum((z ** 2).sum() for z in leaves) / sum(z.numel() for z in leaves))   # :73-75

    rays = batch['rays']
    ret = model_apply(params, rays, int(batch['ts']), batch['ext'], config['randomized'],
                      config['rand_bkgd'], config['white_bkgd'], alpha, noise=noise,
                      cfg=model_cfg, mlp_hook=mlp_hook)

    mask = rays.lossmult
    if config['disable_multiscale_loss']:
        mask = torch.ones_like(mask)
    gt_depth = batch['depth'].reshape(-1)
    gt_sky = batch['sky'].reshape(-1)
    depth_mask = (gt_depth > 0.0).to(dt)                                  # :98
    sky_mask = (gt_sky > 0.0).to(dt)
    sky_mask = sky_mask - depth_mask * sky_mask                           # :101-102
    pixels = batch['pixels'][..., :3]
    target = batch['target']

    st = {k: [] for k in ('losses', 'obj_losses', 'd_losses', 'distr_losses', 'tv_losses',
                          's_losses', 'e_losses', 'n_losses', 'sampling_stats', 'offsets',
                          'offset_x', 'offset_y', 'offset_z', 'offset_yaw')}
    pose = None
    for level_ret in ret:
        tvals, off = level_ret[4], level_ret[7]
        st['sampling_stats'] += [tvals[0, 0], tvals[0, -1]]
        pose, yaw = off
        st['offsets'].append(((pose - target[:, :3]) ** 2).sum())         # :130-134
        st['offset_x'].append(((pose[:, 0] - target[:, 0]) ** 2).sum())
        st['offset_y'].append(((pose[:, 1] - target[:, 1]) ** 2).sum())
        st['offset_z'].append(((pose[:, 2] - target[:, 2]) ** 2).sum())
        st['offset_yaw'].append(((yaw - target[:, 3:]) ** 2).sum())
        st['tv_losses'].append(((pose - prev[:, :, :3]) ** 2).sum())      # :136
        terms, depth_mask = level_terms(level_ret, batch, config, eps, mask, depth_mask, sky_mask)
        for k, v in terms.items():
            st[k].append(v)

    S = {k: torch.stack([torch.as_tensor(x, dtype=dt) for x in v]) for k, v in st.items()}
    loss = total_loss(S, config, weight_l2)
    S['loss'] = loss
    S['weight_l2'] = torch.as_tensor(weight_l2, dtype=dt)
    S['pose'] = pose
    return loss, S, ret


def grad_postprocess(grads, config):
    """train_boxpose.py:257-286: nan/+inf -> 0 (-inf -> most negative), value clip,
    global-norm clip.  Returns (grads, grad_abs_max, grad_norm, grad_norm_clipped)."""
    out = []
    for g in grads:
        g = torch.nan_to_num(g, nan=0.0, posinf=0.0)
        if config['grad_max_val'] > 0:
            g = torch.clamp(g, -config['grad_max_val'], config['grad_max_val'])
        out.append(g)
    grad_abs_max = max([g.abs().max() for g in out if g.numel()] + [torch.zeros(())])
    grad_norm = torch.sqrt(sum((g ** 2).sum() for g in out))
    if config['grad_max_norm'] > 0:
        mult = torch.clamp(config['grad_max_norm'] / (1e-7 + grad_norm), max=1.0)
        out = [mult * g for g in out]
    grad_norm_clipped = torch.sqrt(sum((g ** 2).sum() for g in out))
    return out, grad_abs_max, grad_norm, grad_norm_clipped


def adam_update(leaves, grads, opt_state, lr, beta1=0.9, beta2=0.999, eps=1e-8):
    """flax.optim.Adam.apply_param_gradient (flax<=0.5): bias-corrected moments,
    eps outside the sqrt, weight_decay 0.  opt_state = dict(step, m[], v[])."""
    t = opt_state['step'] + 1.
    new_leaves, new_m, new_v = [], [], []
    for p, g, m, v in zip(leaves, grads, opt_state['m'], opt_state['v']):
        m = beta1 * m + (1. - beta1) * g
        v = beta2 * v + (1. - beta2) * g * g
        m_hat = m / (1 - beta1 ** t)
        v_hat = v / (1 - beta2 ** t)
        new_leaves.append(p - lr * m_hat / (torch.sqrt(v_hat) + eps))
        new_m.append(m)
        new_v.append(v)
    return new_leaves, dict(step=opt_state['step'] + 1, m=new_m, v=new_v)


def set_leaves(params, leaves):
    """Inverse of params_leaves: rebuild the params dict from a flat list."""
    it = iter(leaves)
    out = {'box_centers': next(it)}
    names = ['MLP_0'] + sorted([k for k in params if k.startswith('BoxMLP_')],
                               key=lambda s: int(s.split('_')[1]))
    for n in names:
        out[n] = [[next(it), next(it)] for _ in params[n]]
    return out


def new_opt_state(params):
    leaves = params_leaves(params)
    return dict(step=0, m=[torch.zeros_like(z) for z in leaves],
                v=[torch.zeros_like(z) for z in leaves])


def train_step(params, opt_state, batch, config, model_cfg, lr, eps, alpha, prev,
               noise=None, mlp_hook=None, world_size=1, shards=None):
    """train_boxpose.py:49-321 on one device, or (shards=list of batch dicts) the
    pmap semantics of :253-255: per-shard loss/grad, then mean over shards.
    Returns (new_params, new_opt_state, stats, grads_raw)."""
    def one(b, nz):
        leaves = [z.detach().clone().requires_grad_(True) for z in params_leaves(params)]
        p = set_leaves(params, leaves)
        loss, S, _ = loss_fn(p, b, config, model_cfg, eps, alpha, prev, noise=nz, mlp_hook=mlp_hook)
        grads = torch.autograd.grad(loss, leaves, allow_unused=True)
        grads = [torch.zeros_like(z) if g is None else g for g, z in zip(grads, leaves)]
        return S, grads

    if shards is None:
        S, grads = one(batch, noise)
    else:
        outs = [one(b, None if noise is None else noise[i]) for i, b in enumerate(shards)]
        S = outs[0][0]
        S = dict(S, loss=sum(o[0]['loss'] for o in outs) / len(outs))
        grads = [sum(g) / len(outs) for g in zip(*[o[1] for o in outs])]
    g2, gmax, gnorm, gnorm_c = grad_postprocess(grads, config)
    leaves = [z.detach() for z in params_leaves(params)]
    new_leaves, new_state = adam_update(leaves, g2, opt_state, lr)
    stats = {k: (v.detach() if torch.is_tensor(v) else v) for k, v in S.items()}
    stats.update(grad_abs_max=gmax, grad_norm=gnorm, grad_norm_clipped=gnorm_c,
                 psnrs=mse_to_psnr(stats['losses']), psnr=mse_to_psnr(stats['losses'][-1]))
    return set_leaves(params, new_leaves), new_state, stats, grads


def render_image(params, rays_hw, ts, ext, alpha, chunk=8192, cfg=None, mlp_hook=None,
                 white_bkgd=False):
    """obbpose_model.render_image (:421-479) + render_eval_fn (train_boxpose.py:377-390):
    deterministic, rand_bkgd=False; returns last-level (rgb, sum w*t_mid, acc)."""
    H, W = rays_hw[0].shape[:2]
    n = H * W
    flat = type(rays_hw)(*[r.reshape(n, -1) for r in rays_hw])
    outs = []
    with torch.no_grad():
        for i in range(0, n, chunk):
            cr = type(flat)(*[r[i:i + chunk] for r in flat])
            last = model_apply(params, cr, ts, ext, False, False, white_bkgd, alpha,
                               cfg=cfg, mlp_hook=mlp_hook)[-1]
            outs.append(last[:3])
    rgb, dist, acc = [torch.cat(x, dim=0) for x in zip(*outs)]
    return rgb.reshape(H, W, -1), dist.reshape(H, W), acc.reshape(H, W)


# ----------------------------------------------------------------------------
# bf16 operand-rounding emulation (for checking the bf16 MFMA path tightly)
# ----------------------------------------------------------------------------
def _bf(x):
    return x.to(torch.bfloat16).to(x.dtype)


def mlp_apply_bf16(params, x, condition, cfg):
    """mlp_apply with every GEMM operand (activations and kernels) rounded to
    bf16 (RNE) and products accumulated in x.dtype -- the arithmetic the bf16
    MFMA path performs (fp32 accumulate, fp32 bias add)."""
    feature_dim = x.shape[-1]
    num_samples = x.shape[1]
    x = _bf(x.reshape(-1, feature_dim))
    inputs = x
    li = 0
    for i in range(cfg['net_depth']):
        k, b = params[li]; li += 1
        x = _bf(torch.relu(x @ _bf(k) + b))
        if i % cfg['skip_layer'] == 0 and i > 0:
            x = torch.cat([x, inputs], dim=-1)
    k, b = params[li]; li += 1
    raw_density = (x @ _bf(k) + b).reshape(-1, num_samples, cfg['num_density_channels'])
    if condition is not None:
        k, b = params[li]; li += 1
        bottleneck = _bf(x @ _bf(k) + b)
        cond = _bf(condition)[:, None, :].expand(-1, num_samples, -1).reshape(-1, condition.shape[-1])
        x = torch.cat([bottleneck, cond], dim=-1)
        for _ in range(cfg['net_depth_condition']):
            k, b = params[li]; li += 1
            x = _bf(torch.relu(x @ _bf(k) + b))
    k, b = params[li]; li += 1
    raw_rgb = (x @ _bf(k) + b).reshape(-1, num_samples, cfg['num_rgb_channels'])
    return raw_rgb, raw_density
