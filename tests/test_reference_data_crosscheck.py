"""The data-side restatements (oracle/durf_data_ref.py) against the reference's own source, build container only:
  * ray generation: `Waymo._generate_rays_multi` (internal/obbpose_dataset.py:1868-1916) is plain numpy on `self`'s camera
    fields, so the UNBOUND method runs on a namespace holding them (cv2 / natsort, which the module imports for its file
    readers, are dummies) -- origins, directions, viewdirs, radii, near, far of every camera, bit for bit;
  * SSIM: `math.compute_ssim` (internal/math.py:66-140) under the numpy stand-ins (jax.scipy.signal.convolve2d -> scipy,
    a real vmap over the channel axis) -- value and map.
These are the restatements the device ray generator (csrc/data.hip) and durf_ssim are tested against on the GPU."""
import os
import sys
import types

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import durf_data_ref as D  # noqa: E402
from tests import ref_standin  # noqa: E402

pytestmark = pytest.mark.skipif(not ref_standin.available(), reason='reference tree not present')


@pytest.fixture(scope='module')
def ref():
    mods = ref_standin.load(dataset=True)
    yield mods
    ref_standin.unload()


def _rig(seed, n_cam=3):
    rs = np.random.default_rng(seed)
    h = np.array([24, 20, 32][:n_cam])
    w = np.array([36, 28, 40][:n_cam])
    focal = rs.uniform(400, 600, n_cam).astype(np.float32)
    pp = np.stack([w / 2 + rs.uniform(-2, 2, n_cam), h / 2 + rs.uniform(-2, 2, n_cam)], -1).astype(np.float32)
    c2w = np.zeros((n_cam, 3, 4), np.float32)
    for i in range(n_cam):
        q, _ = np.linalg.qr(rs.normal(size=(3, 3)))
        c2w[i, :3, :3] = q
        c2w[i, :3, 3] = rs.uniform(-0.5, 0.5, 3)
    return h, w, focal, pp, c2w


def test_generate_rays_multi(ref):
    h, w, focal, pp, c2w = _rig(0)
    self = types.SimpleNamespace(h=h, w=w, focal=focal, principal_point=pp, camtoworlds=c2w, near=0.0, far=40.0,
                                 timesteps=np.arange(len(h)))
    ref.obbpose_dataset.Waymo._generate_rays_multi(self)
    want = D.generate_rays_multi(h, w, focal, pp, c2w, near=0.0, far=40.0)
    for name in ('origins', 'directions', 'viewdirs', 'radii', 'lossmult', 'near', 'far'):
        got = getattr(self.rays, name)
        assert len(got) == len(want[name]) == len(h)
        for cam, (g, x) in enumerate(zip(got, want[name])):
            g, x = np.asarray(g, dtype=np.float64), np.asarray(x, dtype=np.float64)
            if name == 'radii':     # `v * 2 / np.sqrt(12)`: float32 under the NumPy 1.x the reference was written for (and in
                #                     the restatement), float64 under NumPy 2's scalar promotion here -- one float32 rounding
                np.testing.assert_allclose(g, x, rtol=1e-7, atol=0, err_msg='radii of camera %d' % cam)
            else:
                np.testing.assert_array_equal(g, x, err_msg='%s of camera %d' % (name, cam))


@pytest.mark.parametrize('shape', [(32, 48, 3), (20, 17, 1)])
def test_compute_ssim(ref, shape):
    rs = np.random.default_rng(5)
    a = rs.uniform(0, 1, shape)
    b = np.clip(a + rs.normal(0, 0.1, shape), 0, 1)
    got = ref.math.compute_ssim(a, b, 1.0)
    got_map = ref.math.compute_ssim(a, b, 1.0, return_map=True)
    want = D.compute_ssim(a, b, 1.0)
    want_map = D.compute_ssim(a, b, 1.0, return_map=True)
    np.testing.assert_allclose(np.asarray(got), np.asarray(want), rtol=0, atol=1e-6)       # the restatement works in float32
    np.testing.assert_allclose(np.asarray(got_map), np.asarray(want_map), rtol=0, atol=1e-5)
