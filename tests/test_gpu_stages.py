"""Stage-level parity: every kernel of the forward ray pipeline against the CPU oracle on the
same seeded inputs, called through the C ABI.  fp32 stages: tolerance stated per test."""
import numpy as np
import pytest
import torch

from durf_amd import ops, synthetic
from oracle import durf_ref as R
from tests import helpers as H

pytestmark = pytest.mark.gpu


def _setup(B, K, seed, cuda):
    b = synthetic.make_batch(B, K, seed=seed, allow_multi_hit=True)
    return b, H.oracle_batch(b), H.device_batch(b, cuda)


@pytest.mark.parametrize('K', [1, 3, 8])
def test_ray_setup_and_compaction(cuda, K):
    b, ob, db = _setup(2048, K, 7, cuda)
    ts = b['ts']
    pose = db['init'][ts].contiguous()
    o_s, d_s, hit, zo = ops.ray_setup(db['rays'].origins, db['rays'].directions, pose, db['ext'])
    # oracle (obbpose_model.py:99-131)
    rays = ob['rays']
    Bn = rays.origins.shape[0]
    box_pose = ob['init'][ts, :, :3].expand(Bn, K, 3)
    box_mat = R.aa2matrix(ob['init'][ts, :, 3:]).expand(Bn, K, 3, 3)
    oo, do = R.world2object_rpy(rays.origins, rays.directions, box_pose, box_mat)
    dims = ob['ext'].expand(Bn, K, 3)
    zi, zo_r, inter = R.ray_box_intersection(oo, do, -dims, dims)
    assert inter.sum() > 0
    assert torch.equal(hit.cpu().long(), inter), 'hit masks must be bit-exact'
    f = inter.float()
    bk = (inter.sum(-1) == 0).float()
    o_ref = (oo * f[..., None]).sum(-2) + bk[..., None] * rays.origins
    d_ref = (do * f[..., None]).sum(-2) + bk[..., None] * rays.directions
    torch.testing.assert_close(o_s.cpu(), o_ref, rtol=0, atol=2e-6)
    torch.testing.assert_close(d_s.cpu(), d_ref, rtol=0, atol=2e-6)
    torch.testing.assert_close(zo.cpu(), (f * zo_r).sum(-1), rtol=1e-6, atol=1e-5)
    idx, count, slot = ops.compact_hits(hit)
    for k in range(K):
        want = torch.nonzero(inter[:, k]).flatten()
        c = int(count[k])
        assert c == want.numel()
        assert torch.equal(idx[k, :c].cpu().long(), want)
        s = slot[:, k].cpu().long()
        assert torch.equal(s[want], torch.arange(c))
        assert (s[inter[:, k] == 0] == -1).all()


@pytest.mark.parametrize('N,randomized', [(128, False), (64, True), (32, True)])
def test_sample_t(cuda, N, randomized):
    b, ob, db = _setup(512, 0, 3, cuda)
    g = torch.Generator().manual_seed(5)
    t_rand = torch.rand(512, N + 1, generator=g)
    t = ops.sample_t(db['rays'].near.reshape(-1), db['rays'].far.reshape(-1), N,
                     t_rand.to(cuda) if randomized else None)
    rays = ob['rays']
    t_ref, _ = R.sample_along_rays(t_rand, rays.origins, rays.directions, rays.radii, N, rays.near,
                                   rays.far, randomized)
    torch.testing.assert_close(t.cpu(), t_ref.contiguous(), rtol=0, atol=4e-6)
    assert (t[:, 1:] >= t[:, :-1]).all()


def test_view_enc(cuda):
    b, ob, db = _setup(1024, 0, 3, cuda)
    vb, vf = ops.view_enc(db['rays'].viewdirs, want_f32=True)
    ref = R.pos_enc(ob['rays'].viewdirs, 0, 4, True)
    torch.testing.assert_close(vf.cpu(), ref, rtol=0, atol=2e-6)
    torch.testing.assert_close(vb.float().cpu()[:, :27], ref.to(torch.bfloat16).float(), rtol=0, atol=8e-3)
    assert (vb[:, 27:] == 0).all()


def _oracle_samples(ob, K, N, ts, randomized=False):
    rays = ob['rays']
    Bn = rays.origins.shape[0]
    if K > 0:
        box_pose = ob['init'][ts, :, :3].expand(Bn, K, 3)
        box_mat = R.aa2matrix(ob['init'][ts, :, 3:]).expand(Bn, K, 3, 3)
        oo, do = R.world2object_rpy(rays.origins, rays.directions, box_pose, box_mat)
        dims = ob['ext'].expand(Bn, K, 3)
        _, _, inter = R.ray_box_intersection(oo, do, -dims, dims)
        f = inter.float()
        bk = (inter.sum(-1) == 0).float()
        o_s = (oo * f[..., None]).sum(-2) + bk[..., None] * rays.origins
        d_s = (do * f[..., None]).sum(-2) + bk[..., None] * rays.directions
    else:
        inter = torch.zeros(Bn, 0, dtype=torch.long)
        o_s, d_s = rays.origins, rays.directions
    t_vals, samples = R.sample_along_rays(None, o_s, d_s, rays.radii, N, rays.near, rays.far, False)
    return o_s, d_s, inter, t_vals.contiguous(), samples


@pytest.mark.parametrize('K', [0, 3])
def test_encode_bkgd(cuda, K):
    N = 64
    b, ob, db = _setup(256, K, 11, cuda)
    o_s, d_s, inter, t_vals, samples = _oracle_samples(ob, K, N, b['ts'])
    masks = inter.float().sum(-1)
    bm = (1 - masks)[:, None, None]
    s2 = R.new_space((bm * samples[0], bm[..., None] * samples[1]))
    ref = R.integrated_pos_enc(s2, 0, 10).reshape(-1, 60)
    hit = inter.int().to(cuda).contiguous()
    ot, of = ops.encode_bkgd(t_vals.to(cuda), o_s.to(cuda).contiguous(), d_s.to(cuda).contiguous(),
                             db['rays'].radii.reshape(-1), hit, True, tile=True, f32=True)
    # multi-hit rays (bkgd mask -1 -> negative variance -> exp overflow) are outside the domain
    ok = (masks <= 1).repeat_interleave(N)
    # fp32 features: |enc| <= 1; high frequencies amplify fp32 rounding of x by 2^9
    torch.testing.assert_close(of.cpu()[ok], ref[ok], rtol=0, atol=3e-4)
    low = [30 * c + 3 * i + j for c in range(2) for i in range(5) for j in range(3)]
    torch.testing.assert_close(of.cpu()[ok][:, low], ref[ok][:, low], rtol=0, atol=1e-5)
    torch.testing.assert_close(of.cpu()[~ok], ref[~ok], rtol=1e-2, atol=1e-3, equal_nan=True)
    ut = H.untile(ot.cpu(), 256 * N, 4)
    torch.testing.assert_close(ut[:, :60], of.cpu().to(torch.bfloat16).float(), rtol=0, atol=0)
    assert (ut[:, 60:] == 0).all()
    # bf16-only call = the production path (hardware sin/exp): within one bf16 quantum of the accurate one
    ot2, _ = ops.encode_bkgd(t_vals.to(cuda), o_s.to(cuda).contiguous(), d_s.to(cuda).contiguous(),
                             db['rays'].radii.reshape(-1), hit, True, tile=True, f32=False)
    ut2 = H.untile(ot2.cpu(), 256 * N, 4)
    torch.testing.assert_close(ut2[ok][:, :60], ref[ok], rtol=0, atol=6e-3)
    assert (ut2[ok][:, :60] - ref[ok]).abs().mean() < 1.5e-3


def test_encode_obj(cuda):
    N, K = 64, 3
    b, ob, db = _setup(512, K, 13, cuda)
    o_s, d_s, inter, t_vals, samples = _oracle_samples(ob, K, N, b['ts'])
    hit = inter.int().to(cuda).contiguous()
    idx, count, slot = ops.compact_hits(hit)
    for alpha in (10.0, 3.3):
        for k in range(K):
            rows = torch.nonzero(inter[:, k]).flatten()
            if rows.numel() == 0:
                continue
            ref = R.weighted_ipe((samples[0][rows], samples[1][rows]), 0, 10, alpha).reshape(-1, 63)
            ot, of = ops.encode_obj(512, idx[k], count[k:k + 1], t_vals.to(cuda), o_s.to(cuda).contiguous(),
                                    d_s.to(cuda).contiguous(), db['rays'].radii.reshape(-1), alpha,
                                    tile=True, f32=True)
            n = rows.numel() * N
            got = of.cpu()[:n]
            torch.testing.assert_close(got[:, :3], ref[:, :3], rtol=0, atol=1e-5)
            # object-frame coordinates reach t=far: 2^9 * 40 amplifies 1-ulp differences of x
            low = [3 + 30 * c + 3 * i + j for c in range(2) for i in range(4) for j in range(3)]
            torch.testing.assert_close(got[:, low], ref[:, low], rtol=0, atol=2e-4)
            assert (got.abs() <= 40.001).all()
            ut = H.untile(ot.cpu(), (n + 31) // 32 * 32, 4)[:n]
            torch.testing.assert_close(ut[:, :63], got.to(torch.bfloat16).float(), rtol=0, atol=0)


@pytest.mark.parametrize('width,in_dim', [(256, 60), (128, 63)])
def test_mlp_fwd(cuda, width, in_dim):
    """bf16 MFMA path vs the oracle MLP with bf16-rounded operands (fp32 accumulate)."""
    N, Bn = 32, 40
    rows = N * Bn
    g = torch.Generator().manual_seed(1)
    cfg = R.MLP_BKGD if width == 256 else R.MLP_BOX
    shapes = R.mlp_layer_shapes(in_dim, 27, cfg)
    params, flat = [], []
    for fi, fo in shapes:
        lim = (6.0 / (fi + fo)) ** 0.5
        k = (torch.rand(fi, fo, generator=g) * 2 - 1) * lim
        bb = (torch.rand(fo, generator=g) - 0.5) * 0.2
        params.append([k, bb])
        flat += [k.reshape(-1), bb]
    flat = torch.cat(flat).to(cuda)
    assert flat.numel() == ops.mlp_param_count(width, in_dim)
    x = torch.randn(Bn, N, in_dim, generator=g)
    x = x.to(torch.bfloat16).float()
    cond = torch.randn(Bn, 27, generator=g).to(torch.bfloat16).float()
    xp = torch.zeros(rows, 64)
    xp[:, :in_dim] = x.reshape(rows, in_dim)
    enc_tile = H.tile(xp, 4).to(cuda)
    view = torch.zeros(Bn, 32)
    view[:, :27] = cond
    view = view.to(torch.bfloat16).to(cuda)
    wf = ops.pack_weights(width, in_dim, flat)
    stash = torch.zeros(ops.mlp_stash_bytes(width, rows), dtype=torch.uint8, device=cuda)
    raw = ops.mlp_fwd(width, rows, N, enc_tile, view, wf, stash=stash)
    raw2 = ops.mlp_fwd(width, rows, N, enc_tile, view, wf)
    assert torch.equal(raw, raw2), 'training and inference instantiations must agree bitwise'
    rgb_ref, dens_ref = R.mlp_apply_bf16(params, x, cond, cfg)
    ref = torch.cat([rgb_ref.reshape(rows, 3), dens_ref.reshape(rows, 1)], -1)
    torch.testing.assert_close(raw.cpu(), ref, rtol=5e-3, atol=5e-3)   # bf16 re-rounding flips
    # vs the un-rounded fp32 MLP: bf16 noise only
    rgb32, dens32 = R.mlp_apply(params, x, cond, cfg)
    ref32 = torch.cat([rgb32.reshape(rows, 3), dens32.reshape(rows, 1)], -1)
    assert (raw.cpu() - ref32).abs().max() < 5e-2
    # stash region 0 = relu(Dense_0) in C-perm order
    KW = width // 16
    st = stash.view(torch.bfloat16)
    r0 = H.untile(st[: rows * width].cpu(), rows, KW)[:, H.cperm_cols(KW)]
    a0 = torch.relu(x.reshape(rows, in_dim) @ params[0][0].to(torch.bfloat16).float() + params[0][1])
    torch.testing.assert_close(r0, a0.to(torch.bfloat16).float(), rtol=1e-2, atol=1e-2)


def test_mlp_fwd_count_and_ray_idx(cuda):
    """object-MLP calling convention: compacted rows, device-side count, gathered view dirs."""
    width, in_dim, N, Bn = 128, 63, 32, 64
    g = torch.Generator().manual_seed(2)
    flat = ((torch.rand(ops.mlp_param_count(width, in_dim), generator=g) - 0.5) * 0.2).to(cuda)
    rows = N * Bn
    enc = H.tile(torch.randn(rows, 64, generator=g), 4).to(cuda)
    view_all = torch.randn(200, 32, generator=g).to(torch.bfloat16).to(cuda)
    ridx = torch.randperm(200, generator=g)[:Bn].int().to(cuda)
    wf = ops.pack_weights(width, in_dim, flat)
    full = ops.mlp_fwd(width, rows, N, enc, view_all[ridx.long()].contiguous(), wf)
    cnt = torch.tensor([37], dtype=torch.int32, device=cuda)
    raw = torch.full((rows, 4), -7.0, device=cuda)
    ops.mlp_fwd(width, rows, N, enc, view_all, wf, ray_idx=ridx, count=cnt, raw=raw)
    assert torch.equal(raw[: 37 * N], full[: 37 * N])
    assert (raw[37 * N:] == -7.0).all()


@pytest.mark.parametrize('N,K', [(128, 2), (64, 0), (32, 1)])
def test_composite_fwd(cuda, N, K):
    Bn = 300
    g = torch.Generator().manual_seed(4)
    raw_b = torch.randn(Bn, N, 4, generator=g) * 2
    t_vals = torch.sort(torch.rand(Bn, N + 1, generator=g) * 40, dim=-1).values
    dirs = torch.randn(Bn, 3, generator=g)
    hit = (torch.rand(Bn, max(K, 1), generator=g) < 0.3).int()[:, :K]
    raws, raw_sum = [], torch.zeros(Bn, N, 4)
    slot = torch.full((Bn, max(K, 1)), -1, dtype=torch.int32)
    for k in range(K):
        rows = torch.nonzero(hit[:, k]).flatten()
        r = torch.randn(Bn, N, 4, generator=g)
        raws.append(r.reshape(-1, 4).to(cuda))
        slot[rows, k] = torch.arange(rows.numel(), dtype=torch.int32)
        raw_sum[rows] += r[: rows.numel()]
    tot = raw_b + raw_sum
    for mode, (white, rand) in {ops.BKGD_GREY: (False, False), ops.BKGD_WHITE: (True, False),
                                ops.BKGD_RAND: (False, True)}.items():
        out = ops.composite_fwd(raw_b.reshape(-1, 4).to(cuda), raws, slot.to(cuda), t_vals.to(cuda),
                                dirs.to(cuda), -1.0, mode)
        ref = R.volumetric_rendering(torch.sigmoid(tot[..., :3]),
                                     torch.nn.functional.softplus(tot[..., 3:] - 1.0), t_vals, dirs, white, rand)
        names = ['rgb', 'depth', 'acc', 'weights']
        for nm, gv, rv, tol in zip(names, out[:4], ref[:4], [1e-5, 4e-4, 1e-5, 1e-5]):
            torch.testing.assert_close(gv.cpu(), rv, rtol=1e-5, atol=tol, msg=lambda m: nm + ': ' + m)
        torch.testing.assert_close(out[4].cpu(), ref[5], rtol=0, atol=4e-6)
        torch.testing.assert_close(out[5].cpu(), ref[6], rtol=0, atol=4e-6)


@pytest.mark.parametrize('N,randomized', [(128, False), (128, True), (64, False), (32, True)])
def test_resample(cuda, N, randomized):
    Bn = 257
    g = torch.Generator().manual_seed(6)
    w = torch.rand(Bn, N, generator=g) ** 4
    w[0] = 0.0                      # all-zero weights (math_test.py:183-268 includes this case)
    w[1] = 0.0; w[1, 17] = 1.0      # single hot bin
    w[2] = 1e-9
    t_vals = torch.sort(torch.rand(Bn, N + 1, generator=g) * 40, dim=-1).values
    u = torch.rand(Bn, N + 1, generator=g)
    out = ops.resample(t_vals.to(cuda), w.to(cuda), 0.01, u.to(cuda) if randomized else None)
    rays_o = torch.zeros(Bn, 3); rays_d = torch.ones(Bn, 3); rad = torch.ones(Bn, 1)
    ref, _ = R.resample_along_rays(u, rays_o, rays_d, rad, t_vals, w, randomized, True, 0.01)
    assert (out[:, 1:] >= out[:, :-1]).all(), 'resampled t must be sorted'
    assert (out.cpu() >= t_vals[:, :1] - 1e-6).all() and (out.cpu() <= t_vals[:, -1:] + 1e-6).all()
    # tolerance: 1e-4 * far; fp32 cumsum order shifts the CDF by a few ulp
    torch.testing.assert_close(out.cpu(), ref, rtol=0, atol=4e-3)
    assert (out.cpu() - ref).abs().median() < 1e-5


@pytest.mark.parametrize('N,K,randomized,blm', [(128, 2, True, 0.0), (64, 0, False, 0.0), (32, 1, True, 2.0), (256, 0, True, 0.0)])
def test_composite_resample_fused_is_bit_identical(cuda, N, K, randomized, blm):
    """durf_composite_resample == durf_composite_fwd + durf_resample (+ durf_loss_prep of both levels), bit for bit;
    durf_loss_bwd's optional rendered outputs == durf_composite_fwd's."""
    Bn = 301                                  # not a multiple of the 4 rays per workgroup
    g = torch.Generator().manual_seed(40 + N)
    raw_b = (torch.randn(Bn, N, 4, generator=g) * 2).reshape(-1, 4).to(cuda)
    t_vals = torch.sort(torch.rand(Bn, N + 1, generator=g) * 40, dim=-1).values.to(cuda)
    dirs = torch.randn(Bn, 3, generator=g).to(cuda)
    hit = (torch.rand(Bn, max(K, 1), generator=g) < 0.3).int()[:, :K]
    raws = []
    slot = torch.full((Bn, max(K, 1)), -1, dtype=torch.int32)
    for k in range(K):
        rows = torch.nonzero(hit[:, k]).flatten()
        raws.append(torch.randn(Bn * N, 4, generator=g).to(cuda))
        slot[rows, k] = torch.arange(rows.numel(), dtype=torch.int32)
    slot = slot.to(cuda)
    u = torch.rand(Bn, N + 1, generator=g).to(cuda) if randomized else None
    lossmult = torch.ones(Bn, device=cuda)
    depth = torch.where(torch.rand(Bn, generator=g) < 0.4, torch.rand(Bn, generator=g) * 30 + 0.5, torch.zeros(Bn)).to(cuda)
    sky = torch.where(torch.rand(Bn, generator=g) < 0.2, torch.full((Bn,), 0.975), torch.zeros(Bn)).to(cuda)
    dyn = (torch.rand(Bn, generator=g) < 0.2).int().to(cuda)
    zo = (torch.rand(Bn, generator=g) * 20).to(cuda) * dyn
    eps = 0.7
    # separate launches
    c = ops.composite_fwd(raw_b, raws, slot, t_vals, dirs, -1.0, ops.BKGD_GREY)
    t1 = ops.resample(t_vals, c[3], 0.01, u)
    n0 = ops.loss_prep(t_vals, lossmult, depth, sky, dyn, zo, eps, blm, 0)
    n1 = ops.loss_prep(t1, lossmult, depth, sky, dyn, zo, eps, blm, 1)
    # fused
    norms = torch.full((2, ops.PREP_ROWS), -7.0, device=cuda)
    f = ops.composite_resample(raw_b, raws, slot, t_vals, dirs, -1.0, ops.BKGD_GREY, 0.01, u,
                               prep=dict(lossmult=lossmult, gt_depth=depth, sky=sky, dyn=dyn, zo=zo, eps=eps,
                                         box_loss_mult=blm, level=0, disable_multiscale=False, norms=norms))
    for a, b_, nm in zip(c, f[:6], ['rgb', 'depth', 'acc', 'weights', 't_mids', 't_dists']):
        assert torch.equal(a, b_), nm
    assert torch.equal(t1, f[6]), 't_vals of the next level'
    assert torch.equal(norms[0], n0) and torch.equal(norms[1], n1), (norms, n0, n1)
    # without the loss prep
    f2 = ops.composite_resample(raw_b, raws, slot, t_vals, dirs, -1.0, ops.BKGD_GREY, 0.01, u)
    assert torch.equal(t1, f2[6]) and torch.equal(c[3], f2[3])
    # rendered outputs out of the loss kernel
    out = (torch.empty(Bn, 3, device=cuda), torch.empty(Bn, device=cuda), torch.empty(Bn, device=cuda),
           torch.empty(Bn, N, device=cuda), torch.empty(Bn, N, device=cuda), torch.empty(Bn, N, device=cuda))
    pixels = torch.rand(Bn, 3, generator=g).to(cuda)
    ops.loss_bwd(raw_b, raws, slot, t_vals, dirs, pixels, lossmult, depth, sky, dyn, zo, n0, eps,
                 [1.0, 1.0, 1e-4, 1e-2, 1.0, 1e-6], blm, 0, 0.5, render_out=out)
    for a, b_, nm in zip(c, out, ['rgb', 'depth', 'acc', 'weights', 't_mids', 't_dists']):
        assert torch.equal(a, b_), 'loss_bwd ' + nm


@pytest.mark.gpu
def test_pack_weights_all_matches_the_per_mlp_packers(cuda):
    """durf_pack_weights_all (one launch per step) writes the same streams as durf_pack_weights / _batch"""
    K = 3
    g = torch.Generator().manual_seed(5)
    nb, no = ops.mlp_param_count(256, 60), ops.mlp_param_count(128, 63)
    pb = (torch.rand(nb, generator=g) - 0.5).to(cuda)
    po = (torch.rand(K * no, generator=g) - 0.5).to(cuda)
    for want_bwd in (False, True):
        (bf, bb), (of, ob) = ops.pack_weights_all(pb, K, po, no, want_bwd=want_bwd)
        ref_b = ops.pack_weights(256, 60, pb, want_bwd=want_bwd)
        ref_o = ops.pack_weights_batch(K, po, no, want_bwd=want_bwd)
        assert torch.equal(bf, ref_b[0] if want_bwd else ref_b) and torch.equal(of, ref_o[0])
        if want_bwd:
            assert torch.equal(bb, ref_b[1]) and torch.equal(ob, ref_o[1])
        else:
            assert bb is None and ob is None
    (bf, _), none = ops.pack_weights_all(pb, 0, None, no)
    assert none is None and torch.equal(bf, ops.pack_weights(256, 60, pb))


@pytest.mark.gpu
@pytest.mark.parametrize('B,K,N,rand', [(300, 3, 32, True), (1024, 1, 128, False), (77, 8, 64, True), (256, 0, 32, True)])
def test_fused_prologue_and_compaction_match_the_separate_launches(cuda, B, K, N, rand):
    """durf_ray_prologue == durf_ray_setup + durf_view_enc + durf_sample_t, durf_compact_all == durf_compact_hits +
    durf_compact_classes: the same device code behind one launch each, so bit for bit"""
    b = synthetic.make_batch(B, K, seed=40 + K, allow_multi_hit=True)
    db = H.device_batch(b, cuda)
    rays = db['rays']
    pose = db['init'][b['ts']].contiguous() if K else torch.zeros(0, 6, device=cuda)
    ext = db['ext'].reshape(-1, 3).contiguous() if K else torch.zeros(0, 3, device=cuda)
    near, far = rays.near.reshape(-1).contiguous(), rays.far.reshape(-1).contiguous()
    t_rand = torch.rand(B, N + 1, device=cuda) if rand else None
    o_s, d_s, hit, zo = ops.ray_setup(rays.origins, rays.directions, pose, ext)
    view = ops.view_enc(rays.viewdirs)
    t = ops.sample_t(near, far, N, t_rand)
    got = ops.ray_prologue(rays.origins, rays.directions, pose, ext, rays.viewdirs, near, far, N, t_rand)
    for a, c, name in zip(got, (o_s, d_s, hit, zo, view, t), ('origins_s', 'dirs_s', 'hit', 'zo', 'view', 't_vals')):
        assert torch.equal(a, c), name
    if K:
        idx, count, slot = ops.compact_hits(hit)
        idx2, count4, slot2, dyn = ops.compact_classes(hit, N)
        (i1, c1, s1), (i2, c2, s2, d2) = ops.compact_all(hit, N)
        assert torch.equal(c1, count) and torch.equal(s1, slot) and torch.equal(c2, count4) and torch.equal(s2, slot2)
        assert torch.equal(d2, dyn)
        for k in range(K):
            n = int(count[k])
            assert torch.equal(i1[k, :n], idx[k, :n])
        for k in range(2):
            n = int(count4[k])
            assert torch.equal(i2[k, :n], idx2[k, :n])
