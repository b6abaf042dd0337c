"""Training-path parity: losses + composite backward (fp32, tight), fused MLP backward and
weight-gradient GEMMs (bf16, norm-wise), clip+Adam, and one full train_step against the oracle."""
import os

import numpy as np
import pytest
import torch

from durf_amd import obbpose_model, ops, synthetic, train_boxpose, utils
from oracle import durf_ref as R
from tests import helpers as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.mark.parametrize('N,level,blm,eps', [(128, 1, 0.0, 3.0), (64, 0, 0.0, 0.5), (32, 1, 2.0, 0.2)])
def test_loss_and_composite_backward(cuda, N, level, blm, eps):
    Bn, L = 384, 2
    g = torch.Generator().manual_seed(11 + N)
    raw = (torch.randn(Bn, N, 4, generator=g) * 1.5).requires_grad_(True)
    t_vals = torch.sort(torch.rand(Bn, N + 1, generator=g) * 40, dim=-1).values
    dirs = torch.randn(Bn, 3, generator=g)
    pixels = torch.rand(Bn, 3, generator=g)
    depth = torch.where(torch.rand(Bn, generator=g) < 0.4, torch.rand(Bn, generator=g) * 30 + 0.5, torch.zeros(Bn))[:, None]
    sky = torch.where(torch.rand(Bn, generator=g) < 0.2, torch.full((Bn,), 0.975), torch.zeros(Bn))[:, None]
    lossmult = torch.ones(Bn, 1)
    dyn = (torch.rand(Bn, generator=g) < 0.2).long()[:, None]
    zo = torch.rand(Bn, generator=g) * 20 * dyn.reshape(-1)
    cfg = dict(R.CONFIG_DEFAULTS, box_loss_mult=blm)
    conf = utils.Config(**{k: v for k, v in cfg.items() if k in utils.Config.__dataclass_fields__})
    mults = train_boxpose.level_multipliers(conf, level, L)
    # ---- oracle: autograd through volumetric_rendering + level_terms ----
    rgb = torch.sigmoid(raw[..., :3])
    dens = torch.nn.functional.softplus(raw[..., 3:] - 1.0)
    vr = R.volumetric_rendering(rgb, dens, t_vals, dirs, False, False)
    level_ret = vr + ([None, None], dyn, zo)
    batch = dict(depth=depth, sky=sky, pixels=pixels)
    dm0 = (depth.reshape(-1) > 0).float()
    sm = (sky.reshape(-1) > 0).float()
    sm = sm - dm0 * sm
    box = (depth.reshape(-1) < zo).float()
    dm_in = dm0 + level * blm * dyn.reshape(-1).float() * box
    terms, _ = R.level_terms(level_ret, batch, cfg, eps, lossmult, dm_in, sm)
    loss = (mults[0] * terms['losses'] + mults[1] * terms['s_losses'] + mults[2] * terms['d_losses'] +
            mults[3] * terms['n_losses'] + mults[4] * terms['e_losses'] + mults[5] * terms['distr_losses'])
    loss.backward()
    # ---- HIP ----
    d = lambda t: t.detach().to(cuda).contiguous()
    dyn_i = d(dyn.reshape(-1).int())
    norm = ops.loss_prep(d(t_vals), d(lossmult.reshape(-1)), d(depth.reshape(-1)), d(sky.reshape(-1)), dyn_i,
                         d(zo), eps, blm, level)
    slot = torch.full((Bn, 1), -1, dtype=torch.int32, device=cuda)
    draw, sums = ops.loss_bwd(d(raw.reshape(-1, 4)), [], slot, d(t_vals), d(dirs), d(pixels), d(lossmult.reshape(-1)),
                              d(depth.reshape(-1)), d(sky.reshape(-1)), dyn_i, d(zo), norm, eps, mults, blm, level, 0.5)
    nrm, sm_ = norm.cpu(), sums.cpu()
    D, S = max(float(nrm[1]), 1.0), max(float(nrm[2]), 1.0)
    got = dict(losses=sm_[0] / nrm[0], obj_losses=sm_[1] / nrm[4], d_losses=sm_[2] / D, n_losses=sm_[3] / D,
               e_losses=sm_[4] / D, s_losses=sm_[5] / S, distr_losses=sm_[6])
    for k, v in got.items():
        torch.testing.assert_close(v, terms[k].detach(), rtol=2e-5, atol=1e-7, msg=lambda m: k + ': ' + m)
    gref = raw.grad.reshape(-1, 4)
    scale = float(gref.abs().max())
    torch.testing.assert_close(draw.cpu(), gref, rtol=1e-4, atol=2e-5 * scale)
    assert _rel(draw.cpu(), gref) < 1e-5


@pytest.mark.parametrize('width,in_dim', [(256, 60), (128, 63)])
def test_mlp_backward_and_weight_grads(cuda, width, in_dim):
    N, Bn = 32, 48
    rows = N * Bn
    g = torch.Generator().manual_seed(3)
    cfg = R.MLP_BKGD if width == 256 else R.MLP_BOX
    shapes = R.mlp_layer_shapes(in_dim, 27, cfg)
    params, flat = [], []
    for fi, fo in shapes:
        lim = (6.0 / (fi + fo)) ** 0.5
        k = ((torch.rand(fi, fo, generator=g) * 2 - 1) * lim).requires_grad_(True)
        bb = ((torch.rand(fo, generator=g) - 0.5) * 0.2).requires_grad_(True)
        params.append([k, bb])
        flat += [k.detach().reshape(-1), bb.detach()]
    flat = torch.cat(flat).to(cuda)
    x = torch.randn(Bn, N, in_dim, generator=g).to(torch.bfloat16).float()
    cond = torch.randn(Bn, 27, generator=g).to(torch.bfloat16).float()
    draw = torch.randn(rows, 4, generator=g) * 0.1
    xp = torch.zeros(rows, 64)
    xp[:, :in_dim] = x.reshape(rows, in_dim)
    enc_tile = H.tile(xp, 4).to(cuda)
    view = torch.zeros(Bn, 32)
    view[:, :27] = cond
    view = view.to(torch.bfloat16).to(cuda)
    wf, wb = ops.pack_weights(width, in_dim, flat, want_bwd=True)
    stash = torch.zeros(ops.mlp_stash_bytes(width, rows), dtype=torch.uint8, device=cuda)
    mask = torch.zeros(ops.mlp_mask_bytes(rows), dtype=torch.uint8, device=cuda)
    ops.mlp_fwd(width, rows, N, enc_tile, view, wf, stash=stash, relu_mask=mask)
    dz, dz_out = ops.mlp_bwd(width, rows, N, draw.to(cuda), wb, mask)
    part, bpart = ops.dw_buffers(width, cuda)
    view_tile = ops.expand_view(rows, N, view)
    # the K axis runs over two levels (here: the same buffers twice, as two separate allocations would be)
    ops.mlp_dw(width, rows, N, [enc_tile] * 2, [view_tile] * 2, [stash, stash.clone()], [dz, dz.clone()],
               [dz_out] * 2, part, bpart)
    grad = torch.zeros_like(flat)
    ops.mlp_dw_finalize(width, in_dim, rows, N, 2, part, bpart, grad, flat)
    grad = grad.cpu() / 2
    # oracle
    rgb, dens = R.mlp_apply_bf16(params, x, cond, cfg)
    out = torch.cat([rgb.reshape(rows, 3), dens.reshape(rows, 1)], -1)
    (out * draw).sum().backward()
    off = 0
    for li, (k, bb) in enumerate(params):
        gk = grad[off:off + k.numel()].reshape(k.shape); off += k.numel()
        gb = grad[off:off + bb.numel()]; off += bb.numel()
        assert _rel(gk, k.grad) < 3e-2, 'dW Dense_%d rel err %g' % (li, _rel(gk, k.grad))
        assert _rel(gb, bb.grad) < 3e-2, 'db Dense_%d rel err %g' % (li, _rel(gb, bb.grad))
    # dz_out tile: slots 0-3 = bf16(draw)
    torch.testing.assert_close(H.untile(dz_out.cpu(), rows, 1)[:, :4], draw.to(torch.bfloat16).float(), rtol=0, atol=0)


def test_clip_adam(cuda):
    g = torch.Generator().manual_seed(9)
    n = 70001
    p = torch.randn(n, generator=g)
    grad = torch.randn(n, generator=g) * 0.3
    grad[5] = float('nan'); grad[6] = float('inf'); grad[7] = float('-inf')
    cfg = dict(grad_max_val=0.1, grad_max_norm=1.0)
    st = dict(step=0, m=[torch.zeros(n)], v=[torch.zeros(n)])
    pd, md, vd = p.to(cuda), torch.zeros(n, device=cuda), torch.zeros(n, device=cuda)
    leaves = [p.clone()]
    for step in range(3):
        gstep = grad * (1 + step)
        g2, gmax, gnorm, gnc = R.grad_postprocess([gstep * 0.5], cfg)     # 0.5 = mean over a world of 2
        leaves, st = R.adam_update(leaves, g2, st, 1e-3)
        stats = ops.clip_adam(pd, md, vd, gstep.to(cuda), 0.5, 0.1, 1.0, 1e-3, step).cpu()
        torch.testing.assert_close(stats[0], gnorm, rtol=1e-5, atol=0)
        torch.testing.assert_close(stats[1], gmax, rtol=1e-6, atol=0)
        torch.testing.assert_close(stats[3], gnc, rtol=1e-5, atol=0)
        torch.testing.assert_close(pd.cpu(), leaves[0], rtol=1e-5, atol=1e-7)
        torch.testing.assert_close(md.cpu(), st['m'][0], rtol=1e-5, atol=1e-9)
        torch.testing.assert_close(vd.cpu(), st['v'][0], rtol=1e-5, atol=1e-12)


@pytest.mark.parametrize('K,N,B', [(1, 32, 256), (0, 64, 256), (3, 32, 256), (2, 32, 141)])
def test_train_step(cuda, K, N, B):
    """One full step (forward, losses, backward, clip, Adam) vs the oracle with bf16-rounded
    GEMM operands.  Loss terms: 1e-3 rel; gradients: 5e-2 norm-wise (bf16 backward).
    B = 141: ragged sizes (partial 256-sample blocks, partial 1024-thread compaction rounds)."""
    utils.clear_gin()
    utils.parse_gin('MipNerfModel.num_samples = %d\nMipNerfModel.density_noise = 0.0\n'
                    'MipNerfModel.no_pose_opt = True\nMipNerfModel.no_yaw_opt = True\n'
                    'Config.randomized = True\nConfig.rand_bkgd = False\nConfig.grad_max_norm = 1.0\n'
                    'Config.grad_max_val = 0.1\nConfig.tv_loss_mult = 0.0\n' % N)
    config = utils.configured(utils.Config)
    b = synthetic.make_batch(B, K, seed=31 + K)
    ob, db = H.oracle_batch(b), H.device_batch(b, cuda)
    model, variables = obbpose_model.construct_mipnerf(1, db, device=cuda)
    g = torch.Generator().manual_seed(4)
    for name in variables.layout.mlp_names():
        for i in range(12):
            bias = variables['params'][name]['Dense_%d' % i]['bias']
            bias.copy_(((torch.rand(bias.shape, generator=g) - 0.5) * 0.1).to(cuda))
    noise_c = dict(t_rand=torch.rand(B, N + 1, generator=g), u_rand=torch.rand(B, N + 1, generator=g))
    noise_d = {k: v.to(cuda) for k, v in noise_c.items()}
    params = H.oracle_params_from_variables(variables)
    flat0 = variables.flat.clone()
    prev_c = ob['init'][0:1]
    prev_d = db['init'][0:1]
    lr, eps, alpha = 5e-4, 3.0, 10.0
    # gradient only (before Adam mutates the parameters)
    grad, raw, pose = train_boxpose.loss_and_grad(model, config, 0, variables, db, eps, alpha, prev_d, noise=noise_d)
    state = train_boxpose.create_train_state(variables)
    new_state, stats, rng, pose = train_boxpose.train_step(model, config, 0, state, db, lr, eps, alpha, prev_d,
                                                           noise=noise_d)
    torch.cuda.synchronize()
    ocfg = dict(R.CONFIG_DEFAULTS, randomized=True, tv_loss_mult=0.0)
    p2, st2, ostats, ograds = R.train_step(params, R.new_opt_state(params), ob, ocfg, dict(num_samples=N), lr, eps,
                                           alpha, prev_c, noise=noise_c, mlp_hook=R.mlp_apply_bf16)
    multi = (ostats['losses'] != ostats['losses']).any()
    assert not multi, 'synthetic batch produced NaN losses (multi-hit ray) -- pick another seed'
    for k in ('losses', 'd_losses', 'n_losses', 'e_losses', 's_losses', 'distr_losses'):
        torch.testing.assert_close(getattr(stats, k).cpu(), ostats[k], rtol=2e-3, atol=1e-6, msg=lambda m: k + ': ' + m)
    torch.testing.assert_close(stats.loss.cpu(), ostats['loss'], rtol=2e-3, atol=1e-6)
    og = torch.cat([x.reshape(-1) for x in ograds])
    assert og.numel() == grad.numel()
    lay = variables.layout
    for name in lay.mlp_names():
        w, _ = lay.mlp_dims(name)
        sl = slice(lay.mlp_off[name], lay.mlp_off[name] + lay.mlp_size[w])
        if float(og[sl].norm()) > 0:
            assert _rel(grad.cpu()[sl], og[sl]) < 5e-2, '%s grad rel err %g' % (name, _rel(grad.cpu()[sl], og[sl]))
    torch.testing.assert_close(stats.grad_norm.cpu(), ostats['grad_norm'], rtol=3e-2, atol=0)
    # post-Adam parameters: the first Adam step moves every weight by ~lr*sign(g)
    newflat = torch.cat([x.reshape(-1) for x in R.params_leaves(p2)])
    step_ref = newflat - flat0.cpu()
    step_got = new_state.variables.flat.cpu() - flat0.cpu()
    assert _rel(step_got, step_ref) < 0.15
    assert new_state.step == 1


@pytest.mark.parametrize('K,alpha,tv,knobs', [
    (2, 4.5, 0.0, {}), (1, 10.0, 0.01, {}),
    # knobs off the shipped configs under pose optimisation: cylinder rays (mip.py:133-152: other Gaussians along the ray) and
    # un-integrated encodings (obbpose_model.py:163-164: the variances are zeroed, the gradient runs through the means alone)
    (2, 6.5, 0.0, dict(ray_shape='cylinder')), (2, 5.5, 0.01, dict(disable_integration=True)),
    (1, 4.5, 0.0, dict(ray_shape='cylinder', disable_integration=True)),
    (2, 10.0, 0.01, dict(disable_integration=True)),
])
def test_box_pose_gradients(cuda, K, alpha, tv, knobs):
    """cfg4: BARF pose optimisation on (no_pose_opt = no_yaw_opt = False) in the PRODUCTION precision (bf16 background
    MLP; the box-hit rays -- object MLPs and the background MLP's one evaluation per hit ray -- in fp32, which is what
    MipNerfModel.obj_precision = 'auto' selects when the pose is optimised).  d(loss)/d(box_centers[ts]) through
    encoding -> frustum Gaussians -> world2object -> Rodrigues vs the autograd of the plain fp32 oracle (no bf16
    rounding on its side): position AND rotation <= 5e-2 norm-wise per object (measured ~1e-4; round 2's all-bf16
    object branch was held to 6e-2 / 0.2 against a bf16-rounded oracle)."""
    B, N = 1024, 32
    utils.clear_gin()
    utils.parse_gin('MipNerfModel.num_samples = %d\nMipNerfModel.density_noise = 0.0\n'
                    'MipNerfModel.no_pose_opt = False\nMipNerfModel.no_yaw_opt = False\n'
                    'Config.randomized = False\nConfig.rand_bkgd = False\nConfig.grad_max_norm = 1.0\n'
                    'Config.grad_max_val = 0.1\nConfig.tv_loss_mult = %g\n' % (N, tv) +
                    ''.join('MipNerfModel.%s = %s\n' % (k, ('"%s"' % v) if isinstance(v, str) else v) for k, v in knobs.items()))
    config = utils.configured(utils.Config)
    b = synthetic.make_batch(B, K, seed=77 + K, noise_boxes=0.05)
    # un-integrated encodings are undamped (sin(2^deg x) at full weight) and fp32 itself becomes the limit of this gradient:
    # the ORACLE in fp32 is 2-11 % away from the oracle in float64 on these batches (profiles/r05_noint_pose_noise_floor.txt,
    # tests/scripts/noint_pose_noise_floor.py), against 1e-6 with integration -- so those cases are held to the float64
    # oracle at 0.15 (measured <= 0.05)
    undamped = bool(knobs.get('disable_integration'))
    odt, tol = (torch.float64, 0.15) if undamped else (torch.float32, 5e-2)
    ob, db = H.oracle_batch(b, odt), H.device_batch(b, cuda)
    model, variables = obbpose_model.construct_mipnerf(5, db, device=cuda)
    assert model.mlp_precision == 'bf16' and model.object_precision() == 'f32'
    params = H.oracle_params_from_variables(variables, odt)
    prev_c, prev_d = ob['init'][0:1] + 0.01, db['init'][0:1] + 0.01
    grad, raw, pose = train_boxpose.loss_and_grad(model, config, 0, variables, db, 3.0, alpha, prev_d)
    torch.cuda.synchronize()
    ocfg = dict(R.CONFIG_DEFAULTS, randomized=False, tv_loss_mult=tv)
    mcfg = dict(num_samples=N, no_pose_opt=False, no_yaw_opt=False, **knobs)
    _, _, ostats, ograds = R.train_step(params, R.new_opt_state(params), ob, ocfg, mcfg, 5e-4, 3.0, alpha, prev_c)
    lay = variables.layout
    got = grad[lay.box[0]:lay.box[1]].view(lay.T, K, 6).cpu().to(odt)
    want = ograds[0]
    ts = b['ts']
    assert float(want[ts].abs().max()) > 0
    other = [t for t in range(lay.T) if t != ts]
    assert float(got[other].abs().max()) == 0.0 and float(want[other].abs().max()) == 0.0
    for k in range(K):
        rp, rr = _rel(got[ts, k, :3], want[ts, k, :3]), _rel(got[ts, k, 3:], want[ts, k, 3:])
        assert rp < tol and rr < tol, 'object %d: position rel err %g, rotation rel err %g\ngot %s\nwant %s' % (
            k, rp, rr, got[ts, k], want[ts, k])
    # the object MLPs' own gradients come out of the fp32 kernels: fp32-class agreement
    for k in range(K):
        so = slice(lay.mlp_off['BoxMLP_%d' % k], lay.mlp_off['BoxMLP_%d' % k] + lay.mlp_size[128])
        og_k = torch.cat([x.reshape(-1) for x in ograds])[so].float()
        # (undamped: the first layer's gradient sees the same fp32 noise of sin(2^9 x); measured 1e-2)
        assert _rel(grad.cpu()[so], og_k) < (3e-2 if undamped else 5e-3), 'BoxMLP_%d grad rel err %g' % (k, _rel(grad.cpu()[so], og_k))
    # the MLP gradients are unaffected by switching pose optimisation on
    og = torch.cat([x.reshape(-1) for x in ograds]).float()
    sl = slice(lay.mlp_off['MLP_0'], lay.mlp_off['MLP_0'] + lay.mlp_size[256])
    assert _rel(grad.cpu()[sl], og[sl]) < 5e-2


def test_pose_chain_fp32(cuda):
    """The fp32 part of the pose-gradient chain in isolation (tight): random d(enc) ->
    d(box_centers[ts]) through weighted_ipe / cast_rays / world2object_rpy / aa2matrix."""
    B, N, K, alpha = 768, 32, 2, 6.3
    b = synthetic.make_batch(B, K, seed=123)
    ob, db = H.oracle_batch(b, torch.float64), H.device_batch(b, cuda)
    ts = b['ts']
    rays = ob['rays']
    pose = ob['init'][ts].clone().requires_grad_(True)           # [K,6]
    box_pose = pose[:, :3].expand(B, K, 3)
    box_mat = R.aa2matrix(pose[:, 3:]).expand(B, K, 3, 3)
    oo, do = R.world2object_rpy(rays.origins, rays.directions, box_pose, box_mat)
    dims = ob['ext'].expand(B, K, 3)
    _, _, inter = R.ray_box_intersection(oo, do, -dims, dims)
    inter = inter.detach()
    f = inter.double()
    bk = (inter.sum(-1) == 0).double()
    o_s = (oo * f[..., None]).sum(-2) + bk[..., None] * rays.origins
    d_s = (do * f[..., None]).sum(-2) + bk[..., None] * rays.directions
    t_vals, samples = R.sample_along_rays(None, o_s, d_s, rays.radii, N, rays.near, rays.far, False)
    g = torch.Generator().manual_seed(0)
    # device side
    pose_d = db['init'][ts].contiguous()
    o_sd, d_sd, hit, zo = ops.ray_setup(db['rays'].origins, db['rays'].directions, pose_d, db['ext'])
    idx, count, slot = ops.compact_hits(hit)
    t_d = ops.sample_t(db['rays'].near.reshape(-1), db['rays'].far.reshape(-1), N)
    sums = torch.zeros(K, 21, device=cuda)
    slab = torch.zeros(K, B * N, 64, device=cuda)                 # the [K, B*N, 64] slab of the batched call
    loss = 0.0
    for k in range(K):
        rows = torch.nonzero(inter[:, k]).flatten()
        d_enc = torch.randn(rows.numel() * N, 64, generator=g, dtype=torch.float64) * 0.1
        d_enc[:, 63] = 0
        enc = R.weighted_ipe((samples[0][rows], samples[1][rows]), 0, 10, alpha).reshape(-1, 63)
        loss = loss + (enc * d_enc[:, :63]).sum()
        buf = torch.zeros(B * N, 64, device=cuda)
        buf[: rows.numel() * N] = d_enc.float().to(cuda)
        slab[k] = buf
        ops.encode_obj_bwd(k, idx[k], count[k:k + 1], buf, t_d, o_sd, d_sd, db['rays'].radii.reshape(-1).contiguous(),
                           db['rays'].origins, db['rays'].directions, pose_d, alpha, sums)
    # all objects in one launch pair: the same kernels, the same sums
    sums_b = torch.zeros(K, 21, device=cuda)
    ops.encode_obj_bwd_batch(K, idx, count, slab, t_d, o_sd, d_sd, db['rays'].radii.reshape(-1).contiguous(),
                             db['rays'].origins, db['rays'].directions, pose_d, alpha, sums_b)
    assert torch.equal(sums_b, sums)
    loss.backward()
    g6 = torch.zeros(K, 6, device=cuda)
    ops.pose_finish(pose_d, sums, True, True, g6)
    want = pose.grad
    for k in range(K):
        assert _rel(g6[k, :3].cpu().double(), want[k, :3]) < 2e-3, (g6[k], want[k])
        assert _rel(g6[k, 3:].cpu().double(), want[k, 3:]) < 2e-3, (g6[k], want[k])
    g6b = torch.zeros(K, 6, device=cuda)
    ops.pose_finish(pose_d, sums, False, True, g6b)               # no_pose_opt=True: position frozen
    assert float(g6b[:, :3].abs().max()) == 0.0 and torch.equal(g6b[:, 3:], g6[:, 3:])


@pytest.mark.parametrize('K,L', [(0, 2), (3, 2), (1, 3)])
def test_train_stats_kernel(cuda, K, L):
    """durf_train_stats against the stat formulas of train_boxpose.py:123-249,291-292 written in torch."""
    g = torch.Generator().manual_seed(11)
    N = 16
    norms = torch.rand(L, 5, generator=g) * 100 + 0.5
    norms[0, 1] = 0.0                                         # no depth rays on level 0 -> max(., 1)
    sums = torch.rand(L, 7, generator=g) * 10
    wl2 = torch.rand((), generator=g)
    pose6, prev6, target6 = (torch.randn(K, 6, generator=g) for _ in range(3))
    t_levels = [torch.rand(4, N + 1, generator=g) for _ in range(L)]
    mults = [0.1, 0.5, 0.7, 0.3, 0.2, 0.05]
    out = ops.train_stats(norms.to(cuda), sums.to(cuda), wl2.to(cuda), pose6.to(cuda) if K else None,
                          prev6.to(cuda) if K else None, target6.to(cuda) if K else None,
                          [t.to(cuda) for t in t_levels], mults, ops.STATS_ASSEMBLE | ops.STATS_PSNR)
    st = {k: v.cpu() for k, v in ops.stats_views(out, L).items()}
    one = torch.ones(())
    D, S = torch.maximum(norms[:, 1], one), torch.maximum(norms[:, 2], one)
    ref = dict(losses=sums[:, 0] / norms[:, 0], obj_losses=sums[:, 1] / norms[:, 4], d_losses=sums[:, 2] / D,
               n_losses=sums[:, 3] / D, e_losses=sums[:, 4] / D, s_losses=sums[:, 5] / S, distr_losses=sums[:, 6])
    pose = pose6[:, :3]
    ref['tv_losses'] = ((pose - prev6[:, :3]) ** 2).sum().expand(L)
    ref['offsets'] = ((pose - target6[:, :3]) ** 2).sum().expand(L)
    for j, nm in enumerate(('offset_x', 'offset_y', 'offset_z')):
        ref[nm] = ((pose[:, j] - target6[:, j]) ** 2).sum().expand(L)
    ref['offset_yaw'] = ((pose6[0, 3:] - target6[:, 3:]) ** 2).sum().expand(L) if K else torch.zeros(L)
    coarse, sky, dep, near, emp, tv = mults
    loss = coarse * ref['losses'][:-1].sum() + ref['losses'][-1] + wl2
    loss = loss + sky * ref['s_losses'][:-1].sum() + 10.0 * sky * ref['s_losses'][-1]
    for m, nm in ((dep, 'd_losses'), (near, 'n_losses'), (emp, 'e_losses'), (tv, 'tv_losses')):
        loss = loss + m * ref[nm][-1] + 0.1 * m * ref[nm][:-1].sum()
    loss = loss + 0.000001 * ref['distr_losses'].sum()
    ref['loss'] = loss
    ref['psnrs'] = R.mse_to_psnr(ref['losses'])
    ref['obj_psnrs'] = R.mse_to_psnr(ref['obj_losses'])
    ref['sampling_stats'] = torch.stack([x for t in t_levels for x in (t[0, 0], t[0, -1])])
    ref['weight_l2'] = wl2
    for k, v in ref.items():
        torch.testing.assert_close(st[k], v.float(), rtol=2e-6, atol=1e-7, msg=lambda m: k + ': ' + m)


def test_main_driver_trains_on_the_synthetic_timestep_dataset(cuda, tmp_path):
    """durf_amd.train_boxpose.main (the reference's main(), train_boxpose.py:324-580): schedules, device-side pose
    feedback, logging, checkpoint, test-image render + PSNR / SSIM -- end to end on the GPU with the device ray generator."""
    from durf_amd import checkpoints
    hist = train_boxpose.main(['--gin_file', os.path.join(ROOT, 'configs', 'waymo.gin'),
                               '--gin_param', 'Config.max_steps = 40', '--gin_param', 'Config.print_every = 10',
                               '--gin_param', 'Config.save_every = 25', '--gin_param', 'Config.batch_size = 512',
                               '--gin_param', 'Config.lr_delay_steps = 0', '--gin_param', 'MipNerfModel.num_samples = 32',
                               '--train_dir', str(tmp_path), '--render_every', '20', '--objects', '3'])
    logs = [r for _, r in hist if 'loss' in r]
    evals = [r for _, r in hist if 'test_psnr' in r]
    assert [s for s, r in hist if 'loss' in r] == [10, 20, 30, 40] and len(evals) == 2
    assert all(np.isfinite(r['loss']) and np.isfinite(r['avg_psnr']) for r in logs)
    assert logs[-1]['avg_loss'] < logs[0]['avg_loss'], 'the loss must go down'
    assert all(np.isfinite(e['test_psnr']) and 0.0 < e['test_ssim'] <= 1.0 for e in evals)
    assert checkpoints._steps(str(tmp_path)) == [25, 40]


@pytest.mark.parametrize('L,K', [(1, 1), (3, 2)])
def test_train_step_with_other_level_counts(cuda, L, K):
    """MipNerfModel.num_levels is a knob (obbpose_model.py:46): 1 level (no resampling, the unfused per-ray launches) and
    3 levels (two fused composite + resample launches, the last level deferred to the loss kernel) against the oracle."""
    N, B = 32, 192
    utils.clear_gin()
    utils.parse_gin('MipNerfModel.num_samples = %d\nMipNerfModel.num_levels = %d\nMipNerfModel.density_noise = 0.0\n'
                    'MipNerfModel.no_pose_opt = True\nMipNerfModel.no_yaw_opt = True\n'
                    'Config.randomized = True\nConfig.rand_bkgd = False\nConfig.grad_max_norm = 1.0\n'
                    'Config.grad_max_val = 0.1\nConfig.tv_loss_mult = 0.0\n' % (N, L))
    config = utils.configured(utils.Config)
    b = synthetic.make_batch(B, K, seed=47 + L)
    ob, db = H.oracle_batch(b), H.device_batch(b, cuda)
    model, variables = obbpose_model.construct_mipnerf(1, db, device=cuda)
    assert model.num_levels == L
    g = torch.Generator().manual_seed(4)
    noise_c = dict(t_rand=torch.rand(B, N + 1, generator=g), u_rand=torch.rand(B, N + 1, generator=g))
    noise_d = {k: v.to(cuda) for k, v in noise_c.items()}
    params = H.oracle_params_from_variables(variables)
    grad, raw, pose = train_boxpose.loss_and_grad(model, config, 0, variables, db, 3.0, 10.0, db['init'][0:1], noise=noise_d)
    stats = ops.stats_views(train_boxpose._assemble_stats(config, db, raw, db['init'][0:1], ops.STATS_ASSEMBLE | ops.STATS_PSNR), L)
    ocfg = dict(R.CONFIG_DEFAULTS, randomized=True, tv_loss_mult=0.0)
    _, _, ostats, ograds = R.train_step(params, R.new_opt_state(params), ob, ocfg, dict(num_samples=N, num_levels=L), 5e-4,
                                        3.0, 10.0, ob['init'][0:1], noise=noise_c, mlp_hook=R.mlp_apply_bf16)
    assert len(raw['ret']) == L and ostats['losses'].shape[0] == L
    for k in ('losses', 'd_losses', 'n_losses', 'e_losses', 's_losses', 'distr_losses'):
        torch.testing.assert_close(stats[k].cpu(), ostats[k], rtol=2e-3, atol=1e-6, msg=lambda m: k + ': ' + m)
    torch.testing.assert_close(stats['loss'].cpu(), ostats['loss'], rtol=2e-3, atol=1e-6)
    og = torch.cat([x.reshape(-1) for x in ograds])
    lay = variables.layout
    for name in lay.mlp_names():
        w, _ = lay.mlp_dims(name)
        sl = slice(lay.mlp_off[name], lay.mlp_off[name] + lay.mlp_size[w])
        if float(og[sl].norm()) > 0:
            assert _rel(grad.cpu()[sl], og[sl]) < 5e-2, '%s grad rel err %g' % (name, _rel(grad.cpu()[sl], og[sl]))


@pytest.mark.parametrize('precision', ['bf16', 'f32'])
@pytest.mark.parametrize('seed', [200, 201, 202, 203, 204, 205] + H.extra_fuzz_seeds('TRAIN'))
def test_train_step_random_configurations(cuda, seed, precision):
    """Seeded sweep over loss-knob / shape combinations of one full step: loss multipliers off their shipped values, box
    loss, weight decay, multiscale loss off, ragged B, K in 0..5.  bf16: the production path against the oracle with
    bf16-rounded GEMM operands (gates of test_train_step).  f32: the exact-fp32 instrument against the fp32 oracle at
    test_train_step_fp32_exact's gates.  DURF_FUZZ_EXTRA=n adds n seeds (soak runs)."""
    import random
    f32 = precision == 'f32'
    r = random.Random(seed)
    K = r.choice([0, 1, 2, 3, 5])
    N = r.choice([32, 64])
    B = r.choice([96, 141, 256])
    knobs = dict(depth_loss_mult=r.choice([1e-4, 1e-2, 0.0]), near_loss_mult=r.choice([1e-2, 0.0, 0.1]),
                 empty_loss_mult=r.choice([1.0, 0.1]), sky_loss_mult=r.choice([1.0, 0.0]),
                 coarse_loss_mult=r.choice([0.1, 1.0]), box_loss_mult=r.choice([0, 0.5]),
                 weight_decay_mult=r.choice([0.0, 1e-4]), disable_multiscale_loss=r.random() < 0.3,
                 grad_max_norm=r.choice([1.0, 0.0, 10.0]), grad_max_val=r.choice([0.1, 0.0]))
    eps, alpha, lr = r.choice([3.0, 0.5]), r.choice([10.0, 2.5]), 5e-4
    # (drawn behind everything else, so that the seeds of earlier rounds keep their other choices) the class defaults no shipped
    # gin file keeps: density noise (obbpose_model.py:57,236-240; injected normal draws on both sides) and rand_bkgd (utils.py:144)
    dnoise, knobs['rand_bkgd'] = r.choice([0.0, 0.0, 0.1]), r.random() < 0.3
    utils.clear_gin()
    utils.parse_gin('MipNerfModel.num_samples = %d\nMipNerfModel.density_noise = %g\n'
                    'MipNerfModel.no_pose_opt = True\nMipNerfModel.no_yaw_opt = True\n'
                    'Config.randomized = True\nConfig.tv_loss_mult = 0.0\n' % (N, dnoise) +
                    ("MipNerfModel.mlp_precision = 'f32'\n" if f32 else '') +
                    ''.join('Config.%s = %r\n' % kv for kv in knobs.items()))
    config = utils.configured(utils.Config)
    b = synthetic.make_batch(B, K, seed=seed)
    ob, db = H.oracle_batch(b), H.device_batch(b, cuda)
    model, variables = obbpose_model.construct_mipnerf(seed, db, device=cuda)
    g = torch.Generator().manual_seed(seed)
    noise_c = dict(t_rand=torch.rand(B, N + 1, generator=g), u_rand=torch.rand(B, N + 1, generator=g))
    if dnoise:
        noise_c['density'] = [torch.randn(B, N, 1, generator=g) for _ in range(2)]
    noise_d = {k: ([x.to(cuda) for x in v] if isinstance(v, list) else v.to(cuda)) for k, v in noise_c.items()}
    params = H.oracle_params_from_variables(variables)
    prev_c, prev_d = ob['init'][0:1], db['init'][0:1]
    grad, raw, pose = train_boxpose.loss_and_grad(model, config, 0, variables, db, eps, alpha, prev_d, noise=noise_d)
    torch.cuda.synchronize()
    ocfg = dict(R.CONFIG_DEFAULTS, randomized=True, tv_loss_mult=0.0, **knobs)
    p2, st2, ostats, ograds = R.train_step(params, R.new_opt_state(params), ob, ocfg, dict(num_samples=N, density_noise=dnoise),
                                           lr, eps, alpha, prev_c, noise=noise_c, mlp_hook=None if f32 else R.mlp_apply_bf16)
    if (ostats['losses'] != ostats['losses']).any():
        pytest.skip('seed %d drew a multi-hit ray (NaN in the reference too)' % seed)
    og = torch.cat([x.reshape(-1) for x in ograds])
    lay = variables.layout
    for name in lay.mlp_names():
        w, _ = lay.mlp_dims(name)
        sl = slice(lay.mlp_off[name], lay.mlp_off[name] + lay.mlp_size[w])
        if float(og[sl].norm()) > 0:
            assert _rel(grad.cpu()[sl], og[sl]) < (1e-3 if f32 else 5e-2), 'seed %d %s %s grad rel err %g' % (
                seed, knobs, name, _rel(grad.cpu()[sl], og[sl]))
    state = train_boxpose.create_train_state(variables)
    new_state, stats, rng, pose = train_boxpose.train_step(model, config, 0, state, db, lr, eps, alpha, prev_d, noise=noise_d)
    for k in ('losses', 'd_losses', 'n_losses', 'e_losses', 's_losses', 'distr_losses'):
        got, want = getattr(stats, k).cpu(), ostats[k]
        tight = 2e-5 if f32 else 2e-3
        # the depth-shaped terms of the RESAMPLED level hinge on where its samples fall relative to the LIDAR depth:
        # a 1e-6 difference in level-1 t_vals moves them by up to 2.4e-4 rel even in exact-fp32 mode (soak run, 46
        # seeds), 5e-3 in bf16 mode -- conditioning of the quantity, not of the kernels
        loose = (1e-3 if f32 else 1e-2) if k in ('n_losses', 'd_losses', 'distr_losses') else tight
        msg = lambda m: 'seed %d %s %s: %s: %s' % (seed, precision, knobs, k, m)
        torch.testing.assert_close(got[:1], want[:1], rtol=tight, atol=1e-7 if f32 else 1e-6, msg=msg)
        torch.testing.assert_close(got[1:], want[1:], rtol=loose, atol=1e-7 if f32 else 1e-6, msg=msg)
    torch.testing.assert_close(stats.loss.cpu(), ostats['loss'], rtol=1e-4 if f32 else 2e-3, atol=1e-7 if f32 else 1e-6)


@pytest.mark.parametrize('precision', ['bf16', 'f32'])
@pytest.mark.parametrize('seed', [300, 302, 303] + H.extra_fuzz_seeds('POSE') + H.slow_fuzz_seeds([301]))
def test_box_pose_gradients_random_configurations(cuda, seed, precision):
    """Seeded sweep of the box-pose gradient (cfg4's path: the batched durf_encode_obj_bwd_batch + durf_pose_finish behind
    the object MLPs' d(enc)): K in 1..5, ragged B, alpha below / at the full BARF window, TV prior on and off, position
    or rotation frozen.  f32: against the fp32 oracle's autograd at test_box_pose_gradients_fp32_exact's 2e-3 (5e-3 here:
    more cancellation with more objects); bf16 = the production precision (bf16 background MLP, box-hit rays in fp32):
    5e-2 norm-wise against the same plain fp32 oracle (round 2, all-bf16 object branch: direction and scale only)."""
    import random
    r = random.Random(seed)
    f32 = precision == 'f32'
    K = r.choice([1, 2, 3, 5])
    N = r.choice([32, 64])
    B = r.choice([300, 512, 1000])
    alpha = r.choice([10.0, 4.5, 2.5])
    tv = r.choice([0.0, 0.01])
    no_pose, no_yaw = r.choice([(False, False), (False, False), (True, False), (False, True)])
    utils.clear_gin()
    utils.parse_gin('MipNerfModel.num_samples = %d\nMipNerfModel.density_noise = 0.0\n'
                    'MipNerfModel.no_pose_opt = %s\nMipNerfModel.no_yaw_opt = %s\n'
                    'Config.randomized = False\nConfig.rand_bkgd = False\nConfig.grad_max_norm = 1.0\n'
                    'Config.grad_max_val = 0.1\nConfig.tv_loss_mult = %g\n' % (N, no_pose, no_yaw, tv) +
                    ("MipNerfModel.mlp_precision = 'f32'\n" if f32 else ''))
    config = utils.configured(utils.Config)
    b = synthetic.make_batch(B, K, seed=seed, noise_boxes=0.05, redraw_noisy_multi_hit=True)
    ob, db = H.oracle_batch(b), H.device_batch(b, cuda)
    model, variables = obbpose_model.construct_mipnerf(seed, db, device=cuda)
    params = H.oracle_params_from_variables(variables)
    prev_c, prev_d = ob['init'][0:1] + 0.01, db['init'][0:1] + 0.01
    grad, raw, pose = train_boxpose.loss_and_grad(model, config, 0, variables, db, 3.0, alpha, prev_d)
    torch.cuda.synchronize()
    ocfg = dict(R.CONFIG_DEFAULTS, randomized=False, tv_loss_mult=tv)
    mcfg = dict(num_samples=N, no_pose_opt=no_pose, no_yaw_opt=no_yaw)
    _, _, ostats, ograds = R.train_step(params, R.new_opt_state(params), ob, ocfg, mcfg, 5e-4, 3.0, alpha, prev_c)
    if (ostats['losses'] != ostats['losses']).any():
        pytest.skip('seed %d drew a multi-hit ray (NaN in the reference too)' % seed)
    lay = variables.layout
    ts = b['ts']
    got = grad[lay.box[0]:lay.box[1]].view(lay.T, K, 6).cpu()[ts].double()
    want = ograds[0][ts].double()
    tag = 'seed %d %s K=%d N=%d B=%d alpha=%g tv=%g no_pose=%s no_yaw=%s' % (seed, precision, K, N, B, alpha, tv, no_pose, no_yaw)
    if no_pose:
        assert float(got[:, :3].abs().max()) == 0.0 and float(want[:, :3].abs().max()) == 0.0, tag
    if no_yaw:
        assert float(got[:, 3:].abs().max()) == 0.0 and float(want[:, 3:].abs().max()) == 0.0, tag
    for sl, frozen in ((slice(0, 3), no_pose), (slice(3, 6), no_yaw)):
        if frozen or float(want[:, sl].norm()) == 0.0:
            continue
        g, w = got[:, sl].reshape(-1), want[:, sl].reshape(-1)
        # (an absolute floor for a component the optimiser has nothing to do with: a gradient that cancels to < 1e-5)
        tol = 5e-3 if f32 else 5e-2
        assert _rel(g, w) < tol or float((g - w).abs().max()) < 1e-6, '%s cols %s rel err %g\ngot %s\nwant %s' % (
            tag, sl, _rel(g, w), g, w)


def test_a_ray_that_hits_two_boxes_poisons_the_step_like_the_reference(cuda):
    """Rays that hit two boxes are garbage in the reference (obbpose_model.py:120-122 sums their object-frame origins, the
    background mask becomes -1, variances go negative): their colours are NaN, the loss is NaN, and d(loss)/d(theta) is NaN
    wherever such a ray's samples reach -- which `jnp.nan_to_num` then turns into a ZERO gradient (train_boxpose.py:263).
    In the oracle (autograd through the reference's formulation) that is every entry of the background MLP here, and in
    an object MLP every weight column whose unit is ReLU-active on one of those samples.  The HIP path never evaluates
    those rays (they belong to no box's compacted list), so it states the outcome at segment granularity instead
    (durf_poison_multi_hit): MLP_0 and the MLPs of the boxes such a ray hits get a zero gradient, a box no such ray hits
    keeps learning.  After one step from zero Adam moments: MLP_0 unchanged on both sides; a hit box unchanged here and
    mostly unchanged in the oracle (the columns of units dead on those samples still move there -- DESIGN.md section 2
    lists this as the one place the step's result is deliberately coarser than the reference's)."""
    B, K, N = 256, 3, 32
    utils.clear_gin()
    utils.parse_gin('MipNerfModel.num_samples = %d\nMipNerfModel.density_noise = 0.0\n'
                    'MipNerfModel.no_pose_opt = True\nMipNerfModel.no_yaw_opt = True\n'
                    'Config.randomized = False\nConfig.rand_bkgd = False\nConfig.grad_max_norm = 1.0\n'
                    'Config.grad_max_val = 0.1\nConfig.tv_loss_mult = 0.0\n' % N)
    config = utils.configured(utils.Config)
    b = None
    for seed in range(400, 460):                      # a batch with at least one two-box ray that leaves a box untouched
        c = synthetic.make_batch(B, K, seed=seed, allow_multi_hit=True)
        ob = H.oracle_batch(c)
        rays = ob['rays']
        pose = ob['init'][c['ts']]
        mats = R.aa2matrix(pose[:, 3:]).expand(B, K, 3, 3)
        oo, do = R.world2object_rpy(rays.origins, rays.directions, pose[:, :3].expand(B, K, 3), mats)
        dims = ob['ext'].expand(B, K, 3)
        hit = R.ray_box_intersection(oo, do, -dims, dims)[2]
        multi = hit.sum(-1) > 1
        touched = hit[multi].sum(0) > 0
        if multi.any() and not touched.all() and (hit[:, ~touched].sum() > 0):
            b, free = c, [k for k in range(K) if not bool(touched[k])]
            break
    assert b is not None, 'no seed produced the wanted batch'
    ob, db = H.oracle_batch(b), H.device_batch(b, cuda)
    model, variables = obbpose_model.construct_mipnerf(7, db, device=cuda)
    params = H.oracle_params_from_variables(variables)
    flat0 = variables.flat.clone()
    state = train_boxpose.create_train_state(variables)
    state, stats, _, _ = train_boxpose.train_step(model, config, 0, state, db, 5e-4, 3.0, 10.0, db['init'][0:1])
    torch.cuda.synchronize()
    ocfg = dict(R.CONFIG_DEFAULTS, randomized=False, tv_loss_mult=0.0)
    p2, _, ostats, _ = R.train_step(params, R.new_opt_state(params), ob, ocfg, dict(num_samples=N), 5e-4, 3.0, 10.0,
                                    ob['init'][0:1], mlp_hook=R.mlp_apply_bf16)
    assert not torch.isfinite(ostats['loss']) and not torch.isfinite(stats.loss.cpu()), 'NaN loss on both sides'
    assert int(stats.multi_hit_rays) == int(multi.sum())
    lay = variables.layout
    new_ref = torch.cat([x.reshape(-1) for x in R.params_leaves(p2)])
    moved = (state.variables.flat - flat0).abs().cpu()
    moved_ref = (new_ref - flat0.cpu()).abs()
    for name in lay.mlp_names():
        w, _ = lay.mlp_dims(name)
        sl = slice(lay.mlp_off[name], lay.mlp_off[name] + lay.mlp_size[w])
        is_free = name != 'MLP_0' and int(name.split('_')[1]) in free
        if is_free:
            assert float(moved_ref[sl].max()) > 0 and float(moved[sl].max()) > 0, name + ': untouched box keeps learning'
        elif name == 'MLP_0':
            assert float(moved_ref[sl].max()) == 0.0, name + ': the reference semantics leave it unchanged'
            assert float(moved[sl].max()) == 0.0, name + ': zero gradient here as well'
        else:
            assert float((moved_ref[sl] > 0).float().mean()) < 0.25, name + ': NaN reaches most of a hit box in the oracle'
            assert float(moved[sl].max()) == 0.0, name + ': zero gradient for the whole box here'
    # the A/B switch DURF_DEDUP_HIT_RAYS=0 (every ray through the full background path) must not change the outcome
    variables.flat.copy_(flat0)
    ops.DEDUP_HIT_RAYS = False
    try:
        st2 = train_boxpose.create_train_state(variables)
        st2, stats2, _, _ = train_boxpose.train_step(model, config, 0, st2, db, 5e-4, 3.0, 10.0, db['init'][0:1])
    finally:
        ops.DEDUP_HIT_RAYS = True
    moved2 = (st2.variables.flat - flat0).abs().cpu()
    assert torch.equal(moved2 == 0, moved == 0), 'the same segments stay put without the de-duplicated forward'


@pytest.mark.parametrize('pose_opt', [False, True])
def test_side_stream_modes_give_the_same_parameters(cuda, monkeypatch, pose_opt):
    """ops.overlap_mode: from 2048 x 128 sample rows per step the object MLP launches go to a side HIP stream ('2'); below,
    everything stays on one stream ('0').  The streams change WHEN a kernel runs, never what it computes: three steps from
    the same state in each mode end in bit-identical parameters, moments and loss (the batch here is far below the
    threshold, so the mode is forced)."""
    B, K, N = 512, 3, 32
    utils.clear_gin()
    utils.parse_gin('MipNerfModel.num_samples = %d\nMipNerfModel.density_noise = 0.0\nMipNerfModel.obj_precision = "bf16"\n'
                    'MipNerfModel.no_pose_opt = %s\nMipNerfModel.no_yaw_opt = %s\n'
                    'Config.randomized = True\nConfig.rand_bkgd = False\n' % (N, not pose_opt, not pose_opt))
    config = utils.configured(utils.Config)
    db = H.device_batch(synthetic.make_batch(B, K, seed=77), cuda)
    results = {}
    for mode in ('0', '2', '3', '1'):
        monkeypatch.setattr(ops, '_MODE', mode)
        assert ops.overlap_mode(B * N) == mode
        model, variables = obbpose_model.construct_mipnerf(5, db, device=cuda)
        state = train_boxpose.create_train_state(variables)
        prev, rng = db['init'][0:1], 3
        for _ in range(3):
            state, stats, rng, pose = train_boxpose.train_step(model, config, rng, state, db, 5e-4, 3.0, 10.0, prev)
        torch.cuda.synchronize()
        results[mode] = (state.variables.flat.clone(), state.m.clone(), state.v.clone(), stats.loss.clone())
    for mode in ('2', '3', '1'):
        for a, b in zip(results['0'], results[mode]):
            assert torch.equal(a, b), 'mode %s differs from the single-stream step' % mode
    monkeypatch.setattr(ops, '_MODE', 'auto')
    assert ops.overlap_mode(2048 * 128) == '2' and ops.overlap_mode(2048 * 128 - 1) == '0'


def test_the_prefetched_trunk_is_dropped_when_the_parameters_change_behind_torch(cuda):
    """MipNerfModel.prefetch_const_trunk caches a function of the parameters for the next pose-optimisation step.  Every
    parameter update of this library goes through raw pointers (durf_clip_adam, durf_train_step) that torch's version
    counter does not see: an update issued between the prefetch and the next forward must invalidate the cache
    (ops.param_generation)."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    wl = bench.setup_workload('cfg4', cuda, rays=256)
    model, config, state, batch, prev = wl['model'], wl['config'], wl['state'], wl['batch'], wl['prev']
    assert model.object_precision() == 'f32'
    state, _, _, _ = train_boxpose.train_step(model, config, 0, state, batch, 5e-4, 3.0, wl['alpha'], prev)   # prefetches
    v = state.variables
    assert getattr(v, '_trunk_cache', None) is not None
    # a second optimizer update that no forward has seen: a large step along a fixed direction
    g = torch.full_like(v.flat, 0.05)
    ops.clip_adam(v.flat, state.m, state.v, g, 1.0, 1.0, 0.0, 1e-2, state.step)
    noise = dict(t_rand=torch.rand(256, model.num_samples + 1, device=cuda), u_rand=torch.rand(256, model.num_samples + 1, device=cuda))
    run = lambda: model.apply(v, 0, batch['rays'], batch['init'], batch['ext'], batch['ts'], randomized=True, rand_bkgd=False,
                              white_bkgd=False, alpha=wl['alpha'], noise=noise)
    got = run()                       # must recompute the trunk from the CURRENT parameters
    v._trunk_cache = None
    want = run()
    for lvl in range(2):
        assert torch.equal(torch.nan_to_num(got[lvl][0]), torch.nan_to_num(want[lvl][0]))


@pytest.mark.parametrize('multi_hit', [False, True])
def test_the_two_launch_tail_equals_the_four_separate_launches(cuda, multi_hit):
    """train_step's tail on one device is {logged scalars + multi-hit outcome + the optimizer's scrub pass} as ONE launch
    (durf_stats_scrub) and Adam (durf_adam_apply) -- round 5's launch diet.  Against durf_poison_multi_hit +
    durf_train_stats + durf_clip_adam (train_boxpose.py:257-289) on the same gradient: parameters, both moments, every logged
    scalar and the gradient statistics are the same bits, with and without rays that hit two boxes."""
    B, K, N = 256, 3, 32
    utils.clear_gin()
    utils.parse_gin('MipNerfModel.num_samples = %d\nMipNerfModel.density_noise = 0.0\n'
                    'MipNerfModel.no_pose_opt = True\nMipNerfModel.no_yaw_opt = True\n'
                    'Config.randomized = False\nConfig.rand_bkgd = False\nConfig.grad_max_norm = 1.0\n'
                    'Config.grad_max_val = 0.1\nConfig.tv_loss_mult = 0.0\n' % N)
    config = utils.configured(utils.Config)
    b = synthetic.make_batch(B, K, seed=405, allow_multi_hit=multi_hit, **(dict(hit_range=(0.2, 0.4)) if multi_hit else {}))
    db = H.device_batch(b, cuda)
    prev = db['init'][0:1]
    results = []
    for merged in (False, True):
        model, variables = obbpose_model.construct_mipnerf(7, db, device=cuda)
        state = train_boxpose.create_train_state(variables)
        state.m.normal_(generator=torch.Generator(device=cuda).manual_seed(1)).mul_(1e-3)
        state.v.uniform_(generator=torch.Generator(device=cuda).manual_seed(2)).mul_(1e-6)
        grad, raw, _ = train_boxpose.loss_and_grad(model, config, 0, variables, db, 3.0, 10.0, prev, defer_poison=merged)
        if multi_hit:
            assert int(raw['multi_hit']) > 0
            assert (raw['poison'] is not None) == merged        # (NaNs of those rays reach MLP_0's gradient either way)
        tvs = [r[4] for r in raw['ret']]
        stat_args = (raw['norms'], raw['sums'], raw['weight_l2'], raw['pose6'], prev[0].contiguous(), db['target'].contiguous(),
                     tvs, train_boxpose._stat_mults(config), ops.STATS_ASSEMBLE | ops.STATS_PSNR)
        if merged:
            out, scratch = ops.stats_scrub(*stat_args, raw['terms'], grad, 1.0, 0.1, poison=raw['poison'])
            gs = ops.adam_apply(variables.flat, state.m, state.v, grad, 1.0, 5e-4, 3, scratch)
        else:
            out = ops.train_stats(*stat_args, terms=raw['terms'])
            gs = ops.clip_adam(variables.flat, state.m, state.v, grad, 1.0, 0.1, 1.0, 5e-4, 3)
        torch.cuda.synchronize()
        results.append([variables.flat.clone(), state.m.clone(), state.v.clone(), grad.clone(), out.clone(), gs.clone(),
                        raw['sums'].clone()])
    for name, a, c in zip(('params', 'm', 'v', 'scrubbed gradient', 'logged scalars', 'gradient statistics', 'term sums'),
                          results[0], results[1]):
        assert torch.equal(a.view(torch.int32), c.view(torch.int32)), name      # (bitwise: the multi-hit loss is NaN, as the reference's)
    assert multi_hit or float(results[0][5][0]) > 0          # (this multi-hit batch touches every box: an all-zero update)
