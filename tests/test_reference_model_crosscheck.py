"""The reference's MipNerfModel.__call__ (obbpose_model.py:68-261), executed from its own source, against the oracle.

Build container only.  tests/ref_standin.py registers numpy-backed stand-ins for jax / flax.linen / gin / absl and imports
`internal/obbpose_model.py` (with mip, mip360, math, box_helpers, utils) UNMODIFIED from /root/reference; the same
parameters, rays, boxes and PRNG draws then go through the reference's model and through oracle/durf_ref.py model_apply
in float64, and every entry of every level's 10-tuple is compared.  This is the orchestration the function-by-function
check (test_reference_crosscheck.py) cannot see: ray selection and masks, the per-object loop and the merge, the
background mask, contraction, the level loop and what feeds the resampler, the order of the PRNG draws.
tests/golden/ref_model_*.npz hold outputs of the same runs for the machines that have no /root/reference
(tests/golden/make_ref_model_golden.py, tests/test_golden_ref_model.py)."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
from oracle import durf_ref as R  # noqa: E402
from tests import ref_standin  # noqa: E402
import make_ref_model_golden as G  # noqa: E402

pytestmark = pytest.mark.skipif(not ref_standin.available(), reason='reference tree not present')


@pytest.fixture(scope='module')
def ref():
    mods = ref_standin.load()
    yield mods
    ref_standin.unload()


def _close(got, want, tol, what):
    got, want = np.asarray(got, dtype=np.float64), want.detach().double().numpy()
    assert got.shape == want.shape, (what, got.shape, want.shape)
    fin = np.isfinite(want)
    assert (np.isfinite(got) == fin).all(), what + ': non-finite entries differ'
    if fin.any():
        err = np.abs(got[fin] - want[fin]).max()
        assert err <= tol * max(1.0, np.abs(want[fin]).max()), '%s: %g' % (what, err)


@pytest.mark.parametrize('case', sorted(G.CASES))
def test_model_call_matches_the_oracle(ref, case):
    c = G.CASES[case]
    ob, params, noise = G.inputs(case)
    got = G.reference_outputs(ref, case)
    nz = {k: v for k, v in noise.items()} if c['randomized'] else None
    with torch.no_grad():
        want = R.model_apply(params, ob['rays'], ob['ts'], ob['ext'], c['randomized'], False, c['white_bkgd'], c['alpha'],
                             noise=nz, cfg=c['model'])
    assert len(got) == len(want) == c['model'].get('num_levels', 2)
    names = ('rgb', 'distance', 'acc', 'weights', 't_vals', 't_mids', 't_dists')
    for lvl, (g, w) in enumerate(zip(got, want)):
        # Level 0 agrees to 1e-9 (measured <= 4e-10, all of it the stand-in's central-difference `linearize` in the
        # contraction: 1e-15 with contraction off).  Level 1: the oracle forms the resampler's u = linspace(0, 1 - eps)
        # in float32 as JAX with x64 off does, the stand-in in float64 -- a 6e-8 difference in u that moves a sample by up
        # to 3e-7 and a colour by up to 5e-9 (measured): 1e-6.
        for i, nm in enumerate(names):
            _close(g[i], w[i], 1e-9 if lvl == 0 else 1e-6, '%s level %d %s' % (case, lvl, nm))
        _close(g[7][0], w[7][0], 0.0, case + ' box_pose[0]')
        _close(g[7][1], w[7][1], 0.0, case + ' box_rot[0]')
        _close(np.asarray(g[8]).reshape(-1), w[8].reshape(-1).double(), 0.0, case + ' dyn_mask')
        _close(g[9], w[9], 1e-12, case + ' zo')


@pytest.mark.parametrize('devices,chunk', [(1, 64), (3, 50)])
def test_render_image_matches_the_oracle(ref, devices, chunk):
    """obbpose_model.render_image (:421-479), from its own source: flattening of the [H, W] rays, chunking, padding of a
    chunk to a multiple of the device count with edge rays, sharding, un-sharding, dropping the padding, re-assembly --
    around a render function that does what train_boxpose.py:377-397 does (pmap over the shards of
    all_gather(model.apply(..., randomized=False, rand_bkgd=False)))."""
    case = 'ref_model_K3_N32_rand'
    c = G.CASES[case]
    ob, params, _ = G.inputs(case)
    Himg, Wimg = 8, 12                                     # 96 rays; chunk 50 with 3 devices: 50 -> 51, 46 -> 48
    f = lambda t: t.detach().double().numpy()
    rays_hw = ref.utils.BoxRays(*[f(getattr(ob['rays'], n)).reshape(Himg, Wimg, -1) for n in ref.utils.BoxRays._fields])
    tree = ref_standin.flax_tree(params)
    model = ref.obbpose_model.MipNerfModel(**c['model'])

    def render_fn(rng, batch):                             # pmap(in_axes 0 on the batch) of all_gather(model.apply(...))
        per_device = []
        for d in range(ref_standin.Devices.count):
            rays_d = ref.utils.namedtuple_map(lambda r: r[d], batch['rays'])
            per_device.append(model.apply(tree, 0, rays_d, batch['init'][0], batch['ext'][0], batch['ts'][0], randomized=False,
                                          white_bkgd=False, rand_bkgd=False, alpha=batch['alpha'][0]))
        gathered = []
        for lvl in range(len(per_device[0])):
            entries = []
            for i in range(10):
                if i == 7:                                 # [box_pose, box_rot]: a list per device
                    e = [np.stack([np.asarray(dev[lvl][7][j]) for dev in per_device]) for j in range(2)]
                    entries.append([np.stack([x] * len(per_device)) for x in e])
                else:
                    g = np.stack([np.asarray(dev[lvl][i]) for dev in per_device])          # all_gather: [devices, ...]
                    entries.append(np.stack([g] * len(per_device)))                        # on every device
            gathered.append(entries)
        return gathered

    ref_standin.Devices.count = devices
    try:
        got = ref.obbpose_model.render_image(render_fn, rays_hw, tree['params']['box_centers'], f(ob['ext']),
                                             np.array([int(ob['ts'])]), None, c['alpha'], chunk=chunk)
    finally:
        ref_standin.Devices.count = 1
    rays_t = type(ob['rays'])(*[getattr(ob['rays'], n).reshape(Himg, Wimg, -1) for n in type(ob['rays'])._fields])
    want = R.render_image(params, rays_t, ob['ts'], ob['ext'], c['alpha'], chunk=chunk, cfg=c['model'])
    for g, w, nm in zip(got, want, ('rgb', 'distance', 'acc')):
        _close(g, w, 1e-6, 'render_image ' + nm)
