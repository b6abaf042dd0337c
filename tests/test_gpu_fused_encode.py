"""durf_mlp_fwd_enc: the background forward that encodes its own tiles (csrc/mlp_fwd.hip k_mlp_fwd<256, .., ENC>,
csrc/enc_lane.h) against the two launches it replaces (durf_encode_bkgd + durf_mlp_fwd; ops.FUSED_ENCODE = False).
One encoder body serves both, so EVERYTHING must be bit-identical: the encoding tile the weight-gradient GEMMs read, the
raw outputs, the activation stash, the ReLU masks, every rendered quantity of both levels, and a training step's gradient
(reference: obbpose_model.py:205-210, mip.py:155-179,226-282, mip360.py:47-79)."""
import os

import pytest
import torch

from durf_amd import obbpose_model, ops, synthetic, train_boxpose, utils
from tests import helpers as H

pytestmark = pytest.mark.gpu


def _setup(cuda, B, K, N, seed, extra=''):
    utils.clear_gin()
    utils.parse_gin('MipNerfModel.num_samples = %d\nMipNerfModel.density_noise = 0.0\n'
                    'MipNerfModel.no_pose_opt = True\nMipNerfModel.no_yaw_opt = True\n'
                    'Config.randomized = True\nConfig.rand_bkgd = False\n' % N + extra)
    config = utils.configured(utils.Config)
    b = synthetic.make_batch(B, K, seed=seed, hit_range=(0.2, 0.4))
    db = H.device_batch(b, cuda)
    model, variables = obbpose_model.construct_mipnerf(seed, db, device=cuda)
    g = torch.Generator().manual_seed(seed)
    for name in variables.layout.mlp_names():
        for i in range(12):
            bias = variables['params'][name]['Dense_%d' % i]['bias']
            bias.copy_(((torch.rand(bias.shape, generator=g) - 0.5) * 0.1).to(cuda))
    noise = dict(t_rand=torch.rand(B, N + 1, generator=g).to(cuda), u_rand=torch.rand(B, N + 1, generator=g).to(cuda))
    return config, b, db, model, variables, noise


def _same(a, c, what):
    assert a.shape == c.shape and a.dtype == c.dtype, what
    if a.dtype.is_floating_point:
        a, c = a.float(), c.float()
        ok = (a == c) | (torch.isnan(a) & torch.isnan(c))
    else:
        ok = a == c
    assert bool(ok.all()), '%s: %d of %d entries differ' % (what, int((~ok).sum()), ok.numel())


def _both(fn):
    out = {}
    for on in (True, False):
        ops.FUSED_ENCODE = on
        try:
            out[on] = fn()
        finally:
            ops.FUSED_ENCODE = True
    return out[True], out[False]


# (rays, boxes, samples): whole 256-sample blocks; 128-sample blocks of 4 waves (small launches); K = 0 (no compaction,
# no tail rows); ragged ray counts; the metric's own samples per ray
SHAPES = [(512, 3, 64), (96, 1, 32), (300, 0, 32), (517, 3, 64), (256, 8, 32), (1024, 3, 128)]


@pytest.mark.parametrize('B,K,N', SHAPES)
@pytest.mark.parametrize('train', [False, True])
def test_forward_is_bit_identical_to_the_separate_launches(cuda, B, K, N, train):
    config, b, db, model, variables, noise = _setup(cuda, B, K, N, 31 + K)

    def run():
        return model._forward(variables, 0, db['rays'], db['init'], db['ext'], b['ts'], True, False, False, 10.0,
                              train=train, noise=noise,
                              loss_prep=None)
    (ret_f, ctx_f), (ret_s, ctx_s) = _both(run)
    for lvl in range(model.num_levels):
        for i, nm in enumerate(['rgb', 'depth', 'acc', 'weights', 't_vals', 't_mids', 't_dists']):
            _same(ret_f[lvl][i], ret_s[lvl][i], '%s level %d' % (nm, lvl))
        if train:
            lf, ls = ctx_f['levels'][lvl], ctx_s['levels'][lvl]
            _same(lf['raw_b'], ls['raw_b'], 'raw level %d' % lvl)
            dd = ctx_f.get('dedup')
            # the rows the kernels own: all of them, or the compacted rows + the tail rows of a de-duplicated batch
            nrows = B * N if dd is None else int(dd['count'][0]) * N + int(dd['count'][1])
            nt = (nrows + 31) // 32
            enc_f = lf['enc_b'].view(torch.int16).reshape(-1, 4 * 512)[:nt]
            enc_s = ls['enc_b'].view(torch.int16).reshape(-1, 4 * 512)[:nt]
            if dd is not None and nrows % 32:          # a partial last (tail) tile: lanes beyond the tail hold whatever
                enc_f, enc_s = enc_f[:-1], enc_s[:-1]  # (checked through the outputs above)
            _same(enc_f, enc_s, 'encoding tile level %d' % lvl)
            whole = nrows // 32
            kb = ops.mlp_stash_bytes(obbpose_model.W_BKGD, 32) // 1024        # KB per 32-row tile, all regions
            nt_all = lf['stash_b'].numel() // (kb * 1024)
            for j in range(10):                                               # region j: [k-steps][tiles][1 KB]
                ks = 8 if j == 9 else 16
                if j == 8:
                    continue                                                  # the linear bottleneck is not stashed
                off = 16 * j * nt_all * 1024
                a = lf['stash_b'][off:off + ks * nt_all * 1024].reshape(nt_all, ks * 1024)[:whole]
                c = ls['stash_b'][off:off + ks * nt_all * 1024].reshape(nt_all, ks * 1024)[:whole]
                _same(a, c, 'stash region %d level %d' % (j, lvl))
            ma = lf['mask_b'].reshape(9, -1, 1024)[:, :whole]
            mc = ls['mask_b'].reshape(9, -1, 1024)[:, :whole]
            _same(ma, mc, 'ReLU masks level %d' % lvl)


@pytest.mark.parametrize('B,K,N', [(512, 3, 64), (300, 0, 32)])
def test_a_training_step_is_bit_identical(cuda, B, K, N):
    config, b, db, model, variables, noise = _setup(cuda, B, K, N, 41 + K)
    prev = db['init'][0:1]

    def run():
        g, raw, _ = train_boxpose.loss_and_grad(model, config, 0, variables, db, 3.0, 10.0, prev, noise=noise)
        return g.clone(), torch.stack([t.clone() for t in raw['terms']])
    (g_f, t_f), (g_s, t_s) = _both(run)
    _same(g_f, g_s, 'gradient')
    _same(t_f, t_s, 'per-ray loss terms')


@pytest.mark.parametrize('extra', ['MipNerfModel.ray_shape = "cylinder"\n', 'MipNerfModel.disable_integration = True\n',
                                   'MipNerfModel.contraction = False\n', 'MipNerfModel.dynamics = False\n'])
def test_encoder_knobs(cuda, extra):
    config, b, db, model, variables, noise = _setup(cuda, 256, 2, 32, 51, extra)

    def run():
        return model.apply(variables, 0, db['rays'], db['init'], db['ext'], b['ts'], randomized=True, rand_bkgd=False,
                           white_bkgd=False, alpha=10.0, noise=noise)
    ret_f, ret_s = _both(run)
    for lvl in range(model.num_levels):
        for i, nm in enumerate(['rgb', 'depth', 'acc', 'weights', 't_vals']):
            _same(ret_f[lvl][i], ret_s[lvl][i], '%s level %d (%s)' % (nm, lvl, extra.strip()))


@pytest.mark.parametrize('B,K,N', [(512, 3, 64), (300, 1, 32), (256, 8, 32)])
def test_the_object_forward_encodes_its_own_tiles_like_the_separate_launches(cuda, B, K, N):
    """durf_obj_fwd_batch is ONE launch since round 4, on the M-split kernel (k_mlp_fwd_ms: 4 waves x 64 samples, one output
    tile per wave, the object encoder's body at its head, the view-direction tile written from the view fragments).  Against the
    sample-split kernel and the separate entry points it
    replaces -- durf_encode_obj, durf_mlp_fwd(128), durf_expand_view, object by object -- every valid byte must be equal:
    encoding tile, raw, stash, masks, view tile (reference: mip.py:182-223, obbpose_model.py:167-201)."""
    config, b, db, model, variables, noise = _setup(cuda, B, K, N, 61 + K)
    ret, ctx = model._forward(variables, 0, db['rays'], db['init'], db['ext'], b['ts'], True, False, False, 6.5,
                              train=True, noise=noise, loss_prep=None)
    rays = db['rays']
    radii = rays.radii.reshape(-1).contiguous()
    rows = B * N
    L = ops._lib.lib()
    wf_stride = int(L.durf_wpack_fwd_bytes(obbpose_model.W_OBJ))
    enc_stride, view_stride = int(L.durf_obj_enc_stride(B, N)), int(L.durf_obj_view_stride(B, N))
    st_stride, mk_stride = ops.mlp_stash_bytes(obbpose_model.W_OBJ, rows), ops.mlp_mask_bytes(rows)
    total = 0
    for lvl in range(model.num_levels):
        lv = ctx['levels'][lvl]
        slabs = lv['slabs']
        for k in range(K):
            cnt = int(ctx['count'][k])
            total += cnt
            if cnt == 0:
                continue
            idx_k, count_k = ctx['idx'][k].contiguous(), ctx['count'][k:k + 1]
            enc_t, _ = ops.encode_obj(B, idx_k, count_k, lv['t_vals'], ctx['o_s'], ctx['d_s'], radii, 6.5)
            stash = torch.empty(st_stride, dtype=torch.uint8, device=cuda)
            mask = torch.empty(mk_stride, dtype=torch.uint8, device=cuda)
            wf = ctx['packs']['obj'][0][k * wf_stride:(k + 1) * wf_stride]
            # the reference side on the sample-split kernel k_mlp_fwd<128> (the batched launch above ran the M-split kernel)
            os.environ['DURF_OBJ_MSPLIT'] = '0'
            try:
                raw = ops.mlp_fwd(obbpose_model.W_OBJ, rows, N, enc_t, ctx['view'], wf, ray_idx=idx_k, count=count_k, stash=stash,
                                  relu_mask=mask)
            finally:
                del os.environ['DURF_OBJ_MSPLIT']
            nt = cnt * N // 32                                        # whole 32-row tiles (N is a multiple of 32)
            _same(slabs.enc[k * enc_stride:(k + 1) * enc_stride].view(torch.int16).reshape(-1, 2048)[:nt],
                  enc_t.view(torch.int16).reshape(-1, 2048)[:nt], 'object %d level %d: encoding tile' % (k, lvl))
            _same(slabs.raw[k][:cnt * N], raw[:cnt * N], 'object %d level %d: raw' % (k, lvl))
            kb = st_stride // ((rows + 31) // 32 * 1024)
            assert kb * ((rows + 31) // 32) * 1024 == st_stride
            nt_all = (rows + 31) // 32
            got_s, want_s = slabs.stash[k * st_stride:(k + 1) * st_stride], stash
            for j in range(10):
                if j == 8:
                    continue
                ks = 8                                                 # W = 128: 8 k-steps per region (the view layer too)
                off = 8 * j * nt_all * 1024
                _same(got_s[off:off + ks * nt_all * 1024].reshape(nt_all, ks * 1024)[:nt],
                      want_s[off:off + ks * nt_all * 1024].reshape(nt_all, ks * 1024)[:nt], 'object %d level %d: stash %d' % (k, lvl, j))
            _same(slabs.mask[k * mk_stride:(k + 1) * mk_stride].reshape(9, -1, 1024)[:, :nt],
                  mask.reshape(9, -1, 1024)[:, :nt], 'object %d level %d: masks' % (k, lvl))
            if lvl == 0:
                vt = ops.expand_view(rows, N, ctx['view'], ray_idx=idx_k, count=count_k)
                got_v = ctx['view_tiles_obj'][k * view_stride:(k + 1) * view_stride].view(torch.int16).reshape(-1, 1024)[:nt]
                _same(got_v, vt.view(torch.int16).reshape(-1, 1024)[:nt], 'object %d: view tile' % k)
    assert total > 0, 'the batch must contain box-hit rays'
    # ... and the background's view tile, written by its level-0 forward
    dd = ctx['dedup'] if K else None
    if dd is not None:
        want = ops.expand_view(rows, N, ctx['view'], ray_idx=dd['idx'][0], count=dd['count'][0:1], tail_idx=dd['idx'][1],
                               tail_count=dd['count'][1:2])
        nrows = int(dd['count'][0]) * N + int(dd['count'][1])
    else:
        want, nrows = ops.expand_view(rows, N, ctx['view']), rows
    ntv = (nrows + 31) // 32
    _same(ctx['view_tile'].view(torch.int16).reshape(-1, 1024)[:ntv], want.view(torch.int16).reshape(-1, 1024)[:ntv],
          'background view tile')


# ((1024, 8, 128): ~600 (object, pair) items for the 256 workgroups of the M-split grid -- several items per workgroup, of
# different objects; (640, 5, 64): a ragged last pair per object)
@pytest.mark.parametrize('B,K,N', [(512, 3, 64), (300, 1, 32), (256, 8, 32), (1024, 3, 128), (1024, 8, 128), (640, 5, 64)])
def test_msplit_object_kernels_give_the_sample_split_kernels_results_bit_for_bit(cuda, B, K, N):
    """k_mlp_fwd_ms / k_mlp_bwd_ms (4 waves x 64 samples, one tile per wave; the object launches since round 4) against
    k_mlp_fwd<128> / k_mlp_bwd<128> (DURF_OBJ_MSPLIT=0): same MFMA instruction, operands and k order per output, so a whole
    training step's gradient, per-ray loss terms and every object buffer (raw, stash, masks, dz, dz_out) must be equal."""
    config, b, db, model, variables, noise = _setup(cuda, B, K, N, 71 + K)
    prev = db['init'][0:1]
    out = {}
    for ms in ('1', '0'):
        os.environ['DURF_OBJ_MSPLIT'] = ms
        try:
            g, raw, _ = train_boxpose.loss_and_grad(model, config, 0, variables, db, 3.0, 10.0, prev, noise=noise)
            torch.cuda.synchronize()
            lv = raw['ctx']['levels']
            cnt = [int(x) for x in raw['ctx']['count']]
            out[ms] = (g.clone(), torch.stack([t.clone() for t in raw['terms']]),
                       [(l['slabs'].raw.clone(), l['slabs'].stash.clone(), l['slabs'].mask.clone(), l['slabs'].dz.clone(),
                         l['slabs'].dz_out.clone()) for l in lv], cnt)
        finally:
            del os.environ['DURF_OBJ_MSPLIT']
    (g1, t1, s1, cnt), (g0, t0, s0, _) = out['1'], out['0']
    assert sum(cnt) > 0
    _same(g1, g0, 'gradient')
    _same(t1, t0, 'per-ray loss terms')
    rows = B * N
    nt_all = (rows + 31) // 32
    st_stride, mk_stride = ops.mlp_stash_bytes(obbpose_model.W_OBJ, rows), ops.mlp_mask_bytes(rows)
    dzo_stride = int(ops._lib.lib().durf_obj_dzout_stride(B, N))
    for lvl, (a, c) in enumerate(zip(s1, s0)):
        for k in range(K):
            nt = cnt[k] * N // 32
            if nt == 0:
                continue
            _same(a[0][k][:cnt[k] * N], c[0][k][:cnt[k] * N], 'raw, object %d level %d' % (k, lvl))
            for which, name in ((1, 'stash'), (3, 'dz')):
                ga, gc = a[which][k * st_stride:(k + 1) * st_stride], c[which][k * st_stride:(k + 1) * st_stride]
                for j in range(10):
                    if j == 8:
                        continue
                    off = 8 * j * nt_all * 1024
                    _same(ga[off:off + 8 * nt_all * 1024].reshape(nt_all, 8192)[:nt],
                          gc[off:off + 8 * nt_all * 1024].reshape(nt_all, 8192)[:nt], '%s region %d, object %d level %d' % (name, j, k, lvl))
            _same(a[2][k * mk_stride:(k + 1) * mk_stride].reshape(9, -1, 1024)[:, :nt],
                  c[2][k * mk_stride:(k + 1) * mk_stride].reshape(9, -1, 1024)[:, :nt], 'masks, object %d level %d' % (k, lvl))
            _same(a[4][k * dzo_stride:(k + 1) * dzo_stride].reshape(-1, 1024)[:nt],
                  c[4][k * dzo_stride:(k + 1) * dzo_stride].reshape(-1, 1024)[:nt], 'dz_out, object %d level %d' % (k, lvl))


@pytest.mark.parametrize('B,K,N', [(512, 3, 64), (96, 1, 32), (517, 3, 64), (256, 8, 32), (1024, 3, 128)])
def test_the_forward_writes_the_full_raw_layout_like_expand_raw(cuda, B, K, N):
    """DURF_FWD_RAW_FULL: a de-duplicated forward writes raw straight into the [B*N,4] layout (its compacted rows at their
    rays' rows, the one evaluation of a box-hit ray at all N samples of that ray) -- bit for bit what durf_expand_raw makes of
    the compacted rows (ops.FWD_SCATTER_RAW = False), training and inference variant of the kernel."""
    config, b, db, model, variables, noise = _setup(cuda, B, K, N, 71 + K)
    for train in (True, False):
        out = {}
        for on in (True, False):
            ops.FWD_SCATTER_RAW = on
            try:
                ret, ctx = model._forward(variables, 0, db['rays'], db['init'], db['ext'], b['ts'], True, False, False, 10.0,
                                          train=train, noise=noise, loss_prep=None)
                torch.cuda.synchronize()
            finally:
                ops.FWD_SCATTER_RAW = True
            out[on] = (ret, ctx)
        assert out[True][1].get('dedup') is not None or not train, 'the batch must take the de-duplicated path'
        for lvl in range(model.num_levels):
            for i, nm in enumerate(['rgb', 'depth', 'acc', 'weights', 't_vals']):
                _same(out[True][0][lvl][i], out[False][0][lvl][i], '%s level %d' % (nm, lvl))
            if train:
                _same(out[True][1]['levels'][lvl]['raw_b'], out[False][1]['levels'][lvl]['raw_b'], 'raw level %d' % lvl)
