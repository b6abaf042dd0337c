"""Self-checks of the data-side restatement (oracle/durf_data_ref.py) -- CPU only."""
import numpy as np

from oracle import durf_data_ref as D


def test_generate_rays_geometry():
    h, w = np.array([6]), np.array([8])
    focal = np.array([10.0], np.float32)
    pp = np.array([[4.0, 3.0]], np.float32)
    c2w = np.zeros((1, 3, 4), np.float32)
    c2w[0, :, :3] = np.eye(3)
    c2w[0, :, 3] = (1, 2, 3)
    r = D.generate_rays_multi(h, w, focal, pp, c2w, 0.5, 40.0)
    d = r['directions'][0]
    assert d.shape == (6, 8, 3)
    np.testing.assert_allclose(d[3, 4], [0, 0, -1], atol=1e-7)          # the principal ray looks down -z
    np.testing.assert_allclose(d[0, 0], [-0.4, 0.3, -1], atol=1e-6)     # x right, y up (:1882-1886)
    np.testing.assert_allclose(np.linalg.norm(r['viewdirs'][0], axis=-1), 1, atol=1e-6)
    np.testing.assert_allclose(r['origins'][0][2, 5], [1, 2, 3])
    # rows are 1/focal apart for an identity camera; the last row repeats (:1896-1902)
    np.testing.assert_allclose(r['radii'][0][..., 0], 0.1 * 2 / np.sqrt(12), rtol=1e-5)
    assert (r['near'][0] == 0.5).all() and (r['far'][0] == 40).all() and (r['lossmult'][0] == 1).all()


def test_ssim_properties():
    rs = np.random.default_rng(0)
    a = rs.uniform(0, 1, (24, 30, 3))
    b = rs.uniform(0, 1, (24, 30, 3))
    assert abs(D.compute_ssim(a, a, 1.0) - 1.0) < 1e-12
    assert abs(D.compute_ssim(a, b, 1.0) - D.compute_ssim(b, a, 1.0)) < 1e-12
    assert D.compute_ssim(a, b, 1.0) < 0.2
    assert D.compute_ssim(a, b, 1.0, return_map=True).shape == (14, 20, 3)
    # constant images: variances vanish, ssim = (2 m0 m1 + c1) / (m0^2 + m1^2 + c1)
    x, y = np.full((12, 12, 1), 0.2), np.full((12, 12, 1), 0.6)
    c1 = 0.01 ** 2
    np.testing.assert_allclose(D.compute_ssim(x, y, 1.0), (2 * 0.12 + c1) / (0.04 + 0.36 + c1), rtol=1e-9)
