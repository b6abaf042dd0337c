"""Device ray generation / batch gather and SSIM against the numpy restatement of the reference
(oracle/durf_data_ref.py; obbpose_dataset.py:1868-1916,1551-1583, internal/math.py:66-137)."""
import numpy as np
import pytest
import torch

from durf_amd import metrics, raygen
from oracle import durf_data_ref as D

pytestmark = pytest.mark.gpu


def _rig(seed, n=3):
    rs = np.random.default_rng(seed)
    h = np.array([24, 20, 24][:n]); w = np.array([36, 30, 36][:n])
    focal = rs.uniform(40, 60, n).astype(np.float32)
    pp = np.stack([w / 2 + rs.uniform(-2, 2, n), h / 2 + rs.uniform(-2, 2, n)], -1).astype(np.float32)
    c2w = np.zeros((n, 3, 4), np.float32)
    for i in range(n):
        q, _ = np.linalg.qr(rs.normal(size=(3, 3)))
        c2w[i, :, :3] = q
        c2w[i, :, 3] = rs.uniform(-1, 1, 3)
    images = [rs.uniform(0, 1, (h[i], w[i], 3)).astype(np.float32) for i in range(n)]
    depth = [rs.uniform(0, 30, (h[i], w[i], 1)).astype(np.float32) for i in range(n)]
    sky = [(rs.uniform(0, 1, (h[i], w[i], 1)) < 0.1).astype(np.float32) for i in range(n)]
    return h, w, focal, pp, c2w, images, depth, sky


def test_generate_batch_matches_reference_rays(cuda):
    h, w, focal, pp, c2w, images, depth, sky = _rig(0)
    rays = D.generate_rays_multi(h, w, focal, pp, c2w, near=0.0, far=40.0)
    n_rays = int((h * w).sum())
    idx = np.random.default_rng(1).integers(0, n_rays, 777)
    want_rays, want_px, want_dp, want_sk = D.timestep_batch(rays, images, depth, sky, [0, 1, 2], idx)
    ts = raygen.TimestepData(c2w, focal, pp, h, w, images, depth, sky, device=cuda)
    got, px, dp, sk = raygen.generate_batch(ts, torch.tensor(idx, dtype=torch.int32, device=cuda), 0.0, 40.0)
    for name in ('origins', 'directions', 'viewdirs', 'radii', 'lossmult', 'near', 'far'):
        # radii = |d(y) - d(y+1)| * 2/sqrt(12): a difference of nearly equal vectors, 1 ulp of d is ~5e-6 of it
        torch.testing.assert_close(getattr(got, name).cpu(), torch.tensor(want_rays[name]),
                                   rtol=3e-5 if name == 'radii' else 2e-6, atol=1e-7,
                                   msg=lambda m: name + ': ' + m)
    assert torch.equal(px.cpu(), torch.tensor(want_px))
    assert torch.equal(dp.cpu(), torch.tensor(want_dp)) and torch.equal(sk.cpu(), torch.tensor(want_sk))
    # full images in order (the render path): last row repeats the previous row's radius (:1899)
    full, _, _, _ = raygen.generate_batch(ts, None, 0.0, 40.0)
    r0 = full.radii.cpu()[:h[0] * w[0]].reshape(h[0], w[0])
    assert torch.equal(r0[-1], r0[-2])
    torch.testing.assert_close(r0, torch.tensor(rays['radii'][0][..., 0]), rtol=3e-5, atol=1e-9)


@pytest.mark.parametrize('shape', [(32, 48, 3), (20, 17, 1)])
def test_ssim_matches_reference(cuda, shape):
    rs = np.random.default_rng(5)
    a = rs.uniform(0, 1, shape).astype(np.float32)
    b = np.clip(a + rs.normal(0, 0.1, shape), 0, 1).astype(np.float32)
    want = D.compute_ssim(a, b, 1.0)
    want_map = D.compute_ssim(a, b, 1.0, return_map=True)
    ta, tb = torch.tensor(a, device=cuda), torch.tensor(b, device=cuda)
    got = metrics.compute_ssim(ta, tb, 1.0)
    got_map = metrics.compute_ssim(ta, tb, 1.0, return_map=True)
    assert abs(float(got) - want) < 2e-5
    np.testing.assert_allclose(got_map.cpu().numpy(), want_map, rtol=0, atol=5e-5)
    assert abs(float(metrics.compute_ssim(ta, ta, 1.0)) - 1.0) < 1e-6


@pytest.mark.gpu
def test_ssim_converted_inputs(cuda):
    """Inputs that need a dtype / layout conversion (RGBA slice, float64, bf16): both converted images must stay
    alive for the call (a freed temporary's block is reused by the next conversion -> SSIM(img1, img1) = 1)."""
    rs = np.random.default_rng(6)
    a4 = rs.uniform(0, 1, (24, 40, 4)).astype(np.float32)
    b4 = np.clip(a4 + rs.normal(0, 0.15, a4.shape), 0, 1).astype(np.float32)
    want = D.compute_ssim(a4[..., :3], b4[..., :3], 1.0)
    assert want < 0.95
    ta, tb = torch.tensor(a4, device=cuda), torch.tensor(b4, device=cuda)
    got = metrics.compute_ssim(ta[..., :3], tb[..., :3], 1.0)            # non-contiguous views
    assert abs(float(got) - want) < 2e-5
    got64 = metrics.compute_ssim(ta[..., :3].double(), tb[..., :3].double(), 1.0)
    assert abs(float(got64) - want) < 2e-5
    gotbf = metrics.compute_ssim(ta[..., :3].bfloat16(), tb[..., :3].bfloat16(), 1.0)
    wantbf = D.compute_ssim(ta[..., :3].bfloat16().float().cpu().numpy(), tb[..., :3].bfloat16().float().cpu().numpy(), 1.0)
    assert abs(float(gotbf) - wantbf) < 2e-5


def test_c2f_levels_on_the_device(cuda):
    """C2F multi-resolution schedule (c2f_obb_dataset.py:306-313,843-891) on top of the device ray generator: the
    factor follows the training iteration, the principal point sits at the image centre, and the rays of every level
    match the reference's ray generation at that resolution."""
    rs = np.random.default_rng(3)
    n, H0, W0, f0 = 2, 64, 96, 200.0
    c2w = np.zeros((n, 3, 4), np.float32)
    for i in range(n):
        q, _ = np.linalg.qr(rs.normal(size=(3, 3)))
        c2w[i, :, :3] = q
        c2w[i, :, 3] = rs.uniform(-1, 1, 3)
    levels = {}
    for fac in raygen.C2F_FACTORS:
        h, w = H0 // fac, W0 // fac
        levels[fac] = dict(h=[h] * n, w=[w] * n, focal=[f0 / fac] * n,
                           images=[rs.uniform(0, 1, (h, w, 3)).astype(np.float32) for _ in range(n)])
    data = raygen.C2FTimestepData(c2w, levels, device=cuda)
    steps = (100, 200, 300)
    for it, fac in ((1, 16), (100, 16), (101, 12), (250, 8), (301, 4)):
        td = data.at(it, steps)
        assert td is data.data[fac]
        h, w = H0 // fac, W0 // fac
        pp = np.tile(np.array([[w * 0.5, h * 0.5]], np.float32), (n, 1))
        want = D.generate_rays_multi(np.array([h] * n), np.array([w] * n), np.array([f0 / fac] * n, np.float32), pp, c2w, 0.0, 40.0)
        got, px, _, _ = raygen.generate_batch(td, None, 0.0, 40.0)
        for name in ('origins', 'directions', 'viewdirs', 'radii'):
            flat = np.concatenate([r.reshape(-1, r.shape[-1]) for r in want[name]], 0)
            torch.testing.assert_close(getattr(got, name).cpu(), torch.tensor(flat), rtol=3e-5 if name == 'radii' else 2e-6,
                                       atol=1e-7, msg=lambda m: '%s at factor %d: %s' % (name, fac, m))
        assert torch.equal(px.cpu(), torch.tensor(np.concatenate([im.reshape(-1, 3) for im in levels[fac]['images']], 0)))


def test_index_uploader_ring(cuda):
    """raygen.IndexUploader: per-step pixel indices through a ring of pinned staging buffers with non_blocking copies --
    every upload arrives intact although the slots are reused (more uploads than slots, growing and shrinking sizes, the GPU
    kept busy in between so that the copies really are pending when the host moves on)."""
    up = raygen.IndexUploader(cuda, slots=3)
    rs = np.random.default_rng(3)
    busy = torch.randn(2048, 2048, device=cuda)
    sent, got = [], []
    for i in range(12):
        n = int(rs.integers(1, 9000))
        idx = rs.integers(0, 1 << 30, n)
        busy = busy @ busy * 1e-3                         # work queued ahead of the copy
        sent.append(idx.astype(np.int32))
        got.append(up(idx))
    torch.cuda.synchronize()
    for a, b in zip(sent, got):
        assert b.dtype == torch.int32 and b.device.type == 'cuda'
        np.testing.assert_array_equal(b.cpu().numpy(), a)
    cpu = raygen.IndexUploader('cpu')
    np.testing.assert_array_equal(cpu(sent[0]).numpy(), sent[0])
