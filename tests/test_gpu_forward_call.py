"""durf_forward (csrc/forward.hip): MipNerfModel.__call__ in inference as ONE C call -- SURVEY 8b's coarse entry point for
hosts that are not Python.  It issues the same stage kernels in the same order as durf_amd/obbpose_model.py does for
train=False, so its results must be BIT-identical to MipNerfModel.apply; and through that path it inherits the parity
with the oracle (and with the reference's own model outputs: tests/test_golden_ref_model.py)."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
import make_ref_model_golden as G  # noqa: E402
from durf_amd import obbpose_model, ops, synthetic, utils  # noqa: E402
from tests import helpers as H  # noqa: E402

pytestmark = pytest.mark.gpu


def _same(a, b, what):
    assert len(a) == len(b)
    for lvl, (x, y) in enumerate(zip(a, b)):
        for i in range(7):
            # (rays that hit two boxes render NaN on both paths: compare with NaN == NaN)
            assert torch.allclose(x[i], y[i], rtol=0, atol=0, equal_nan=True), '%s: level %d output %d differs' % (what, lvl, i)
        assert torch.equal(x[7][0], y[7][0]) and torch.equal(x[7][1], y[7][1])
        assert torch.equal(x[8].reshape(-1).int(), y[8].reshape(-1).int()), what + ': dyn_mask'
        assert torch.allclose(x[9], y[9], rtol=0, atol=0, equal_nan=True), what + ': zo'


@pytest.mark.parametrize('B,K,N,randomized,knobs', [
    (4096, 3, 128, False, {}),                                   # a render chunk of the metric's shape
    (1000, 1, 64, True, {}),                                     # ragged ray count, stratified sampling
    (777, 0, 32, False, {}),                                     # static model
    (640, 8, 32, True, dict(ray_shape='cylinder', disable_integration=True)),
    (512, 2, 96, False, dict(contraction=False, lindisp=False)),
])
def test_one_call_forward_is_bit_identical_to_apply(cuda, B, K, N, randomized, knobs):
    utils.clear_gin()
    lines = ['MipNerfModel.num_samples = %d' % N, 'MipNerfModel.density_noise = 0.0', 'MipNerfModel.no_pose_opt = True',
             'MipNerfModel.no_yaw_opt = True'] + ['MipNerfModel.%s = %r' % kv for kv in knobs.items()]
    utils.parse_gin('\n'.join(lines).replace("'", '"') + '\n')
    b = synthetic.make_batch(B, K, seed=900 + K, allow_multi_hit=K > 1)
    db = H.device_batch(b, cuda)
    model, variables = obbpose_model.construct_mipnerf(3, db, device=cuda)
    g = torch.Generator(device='cpu').manual_seed(4)
    noise = dict(t_rand=torch.rand(B, N + 1, generator=g).to(cuda), u_rand=torch.rand(B, N + 1, generator=g).to(cuda))
    for white in (False, True):
        kw = dict(randomized=randomized, rand_bkgd=False, white_bkgd=white, alpha=6.5, noise=noise if randomized else None)
        want = model.apply(variables, 0, db['rays'], db['init'], db['ext'], b['ts'], **kw)
        got = model.apply_one_call(variables, 0, db['rays'], db['init'], db['ext'], b['ts'], **kw)
        _same(got, want, 'B=%d K=%d N=%d' % (B, K, N))


@pytest.mark.parametrize('K,mode', [(2, 'injected'), (0, 'key'), (3, 'generator'), (1, 'sampling draws only')])
def test_one_call_forward_with_density_noise_and_no_background_colour(cuda, K, mode):
    """MipNerfModel.density_noise > 0 (obbpose_model.py:236-240; the class default, both shipped gin files set 0) and
    rand_bkgd (mip.py:324: no background colour at all) through durf_forward: the normal draws injected, made by the library
    under the host's key, or taken from a torch.Generator -- each bit-identical to apply() given the same source."""
    B, N = 384, 32
    utils.clear_gin()
    utils.parse_gin('MipNerfModel.num_samples = %d\nMipNerfModel.density_noise = 0.1\nMipNerfModel.no_pose_opt = True\n'
                    'MipNerfModel.no_yaw_opt = True\n' % N)
    b = synthetic.make_batch(B, K, seed=931 + K, allow_multi_hit=K > 1)
    db = H.device_batch(b, cuda)
    model, variables = obbpose_model.construct_mipnerf(3, db, device=cuda)
    g = torch.Generator(device='cpu').manual_seed(4)
    noise = dict(t_rand=torch.rand(B, N + 1, generator=g).to(cuda), u_rand=torch.rand(B, N + 1, generator=g).to(cuda))
    if mode == 'injected':
        noise['density'] = [torch.randn(B, N, 1, generator=g).to(cuda) for _ in range(2)]
    kw = dict(randomized=True, rand_bkgd=True, white_bkgd=False, alpha=6.5)
    if mode in ('injected', 'sampling draws only'):
        want = model.apply(variables, 5, db['rays'], db['init'], db['ext'], b['ts'], noise=noise, **kw)
        got = model.apply_one_call(variables, 5, db['rays'], db['init'], db['ext'], b['ts'], noise=noise, **kw)
    elif mode == 'key':
        want = model.apply(variables, 77, db['rays'], db['init'], db['ext'], b['ts'], **kw)
        got = model.apply_one_call(variables, 77, db['rays'], db['init'], db['ext'], b['ts'], **kw)
    else:
        gens = [torch.Generator(device=cuda).manual_seed(9) for _ in range(2)]
        want = model.apply(variables, gens[0], db['rays'], db['init'], db['ext'], b['ts'], **kw)
        got = model.apply_one_call(variables, gens[1], db['rays'], db['init'], db['ext'], b['ts'], **kw)
    _same(got, want, 'density noise, %s' % mode)
    model.density_noise = 0.0
    quiet = model.apply(variables, 77, db['rays'], db['init'], db['ext'], b['ts'], noise=noise, **kw)
    assert not torch.equal(quiet[1][3], want[1][3]), 'the noise must matter in this test'


def test_one_call_forward_reproduces_the_reference_model_outputs(cuda):
    """... and directly: the fixture made by the reference's own MipNerfModel.__call__ (tests/golden/ref_model_*.npz)"""
    case = 'ref_model_waymo_K3_N128'
    c = G.CASES[case]
    gold = np.load(os.path.join(ROOT, 'tests', 'golden', case + '.npz'))
    b, variables_cpu, noise = G.build(case)
    utils.clear_gin()
    utils.parse_gin('MipNerfModel.num_samples = 128\nMipNerfModel.density_noise = 0.0\nMipNerfModel.no_pose_opt = True\n'
                    'MipNerfModel.no_yaw_opt = True\n')
    model = utils.configured(obbpose_model.MipNerfModel)
    variables = variables_cpu.like(variables_cpu.flat.to(cuda))
    db = H.device_batch(b, cuda)
    nz = {k: v.float().to(cuda) for k, v in noise.items()}
    ret = model.apply_one_call(variables, 0, db['rays'], db['init'], db['ext'], b['ts'], randomized=True, rand_bkgd=False,
                               white_bkgd=False, alpha=c['alpha'], noise=nz)
    for lvl in range(2):
        for i, (nm, tol) in enumerate(zip(G.NAMES, (2e-2, 2e-2 * 40, 2e-2, 2e-2, 2e-2 * 40))):
            np.testing.assert_allclose(ret[lvl][i].double().cpu().numpy(), gold['l%d_%s' % (lvl, nm)], rtol=0, atol=tol)
    np.testing.assert_array_equal(ret[0][8].reshape(-1).cpu().numpy(), gold['dyn_mask'])


def test_workspace_size_and_argument_checks(cuda):
    L = ops._lib.lib()
    assert L.durf_forward_workspace_bytes(4096, 128, 3) > L.durf_forward_workspace_bytes(4096, 128, 0) > 0
    a = ops.ForwardArgs()
    a.B, a.N, a.K, a.num_levels = 16, 48, 0, 2                   # num_samples not a multiple of 32
    ws = torch.empty(1 << 20, dtype=torch.uint8, device=cuda)
    import ctypes
    assert L.durf_forward(None, ctypes.byref(a), ws.data_ptr(), ws.numel()) != 0
    assert b'num_samples' in L.durf_last_error()


def test_an_undersized_workspace_is_refused(cuda, monkeypatch):
    """The one-call entry points carve up to ~10 GB of intermediates out of the caller's buffer: a buffer sized for another
    shape has to be refused with a message naming both sizes, not overrun (SURVEY 8b's durf_workspace_bytes; the reference
    raises on a bad batch shape, train_boxpose.py:332-333) -- and before the first launch: the parameters stay untouched."""
    from durf_amd import train_boxpose
    utils.clear_gin()
    utils.parse_gin('MipNerfModel.num_samples = 32\nMipNerfModel.density_noise = 0.0\nMipNerfModel.no_pose_opt = True\n'
                    'MipNerfModel.no_yaw_opt = True\nConfig.randomized = False\n')
    config = utils.configured(utils.Config)
    b = synthetic.make_batch(64, 2, seed=3)
    db = H.device_batch(b, cuda)
    model, variables = obbpose_model.construct_mipnerf(0, db, device=cuda)
    args = (variables, 0, db['rays'], db['init'], db['ext'], b['ts'])
    kw = dict(randomized=False, rand_bkgd=False, white_bkgd=False, alpha=10.0)
    good = model.apply_one_call(*args, **kw)
    real = ops._workspace
    monkeypatch.setattr(ops, '_workspace', lambda dev, n: real(dev, n)[:n - 256])        # same pointer, 256 bytes short
    need = int(ops._lib.lib().durf_forward_workspace_bytes(64, 32, 2))
    with pytest.raises(ops._lib.DurfError, match=r'durf_forward: workspace of %d bytes.* = %d' % (need - 256, need)):
        model.apply_one_call(*args, **kw)
    state = train_boxpose.create_train_state(variables)
    before = variables.flat.clone()
    for update, who in ((True, 'durf_train_step'), (False, 'durf_loss_backward')):
        with pytest.raises(ops._lib.DurfError, match=who + ': workspace of'):
            train_boxpose.train_step_one_call(model, config, 0, state, db, 5e-4, 3.0, 10.0, db['init'][0:1], update=update)
    torch.cuda.synchronize()
    assert torch.equal(variables.flat, before), 'refused before the first launch'
    monkeypatch.setattr(ops, '_workspace', real)
    _same(model.apply_one_call(*args, **kw), good, 'the right size is accepted')


@pytest.mark.parametrize('K,N,hw,chunk', [(3, 128, (96, 100), 8192), (2, 32, (37, 53), 512), (0, 64, (20, 31), 256)])
def test_one_call_per_image_is_bit_identical_to_render_image_over_chunks(cuda, K, N, hw, chunk):
    """durf_render_image (csrc/forward.hip): render_image (obbpose_model.py:421-479) for one device with the chunk loop in C
    over the ray buffer resident on the device -- the last level's rgb / distance / acc of every pixel must be the bits
    render_image produces over durf_forward chunks (ragged last chunk included), which in turn equals apply() chunk by chunk"""
    from durf_amd import train_boxpose
    utils.clear_gin()
    utils.parse_gin('MipNerfModel.num_samples = %d\nMipNerfModel.no_pose_opt = True\nMipNerfModel.no_yaw_opt = True\n' % N)
    config = utils.configured(utils.Config)
    Hh, Ww = hw
    b = synthetic.make_batch(Hh * Ww, K, seed=21 + K)
    db = H.device_batch(b, cuda)
    model, variables = obbpose_model.construct_mipnerf(1, db, device=cuda)
    rays = utils.namedtuple_map(lambda r: r.reshape(Hh, Ww, -1), db['rays'])
    fn = train_boxpose.make_render_fn(model, config, variables, one_call=True)
    want = obbpose_model.render_image(fn, rays, db['init'], db['ext'], b['ts'], 0, 10.0, chunk=chunk)
    got = model.render_image_one_call(variables, rays, db['init'], db['ext'], b['ts'], config.white_bkgd, 10.0, chunk=chunk)
    torch.cuda.synchronize()
    for name, g, w in zip(('rgb', 'distance', 'acc'), got, want):
        assert g.shape == w.shape
        assert torch.allclose(g, w, rtol=0, atol=0, equal_nan=True), name
    # evaluate() takes that path on one device and reports the same PSNR as the chunked render
    case = dict(rays=rays, pixels=db['pixels'].reshape(Hh, Ww, -1), init=db['init'], ext=db['ext'], ts=b['ts'])
    ev = train_boxpose.evaluate(model, config, variables, case, 10.0, chunk=chunk)
    assert torch.allclose(ev['rgb'], want[0], rtol=0, atol=0, equal_nan=True)
    # a workspace sized for a smaller chunk is refused
    import ctypes
    L = ops._lib.lib()
    small = int(L.durf_render_image_workspace_bytes(chunk // 2, N, K, 2))
    assert small < int(L.durf_render_image_workspace_bytes(chunk, N, K, 2))
