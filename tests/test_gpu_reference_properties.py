"""The assertions the REFERENCE holds for this path (internal/math_test.py) run against the HIP kernels through the
C ABI -- the only reference-owned checks that can touch a kernel (the reference has no golden vectors for rendering):

  * math_test.py:183-268  sorted_piecewise_constant_pdf reproduces a piecewise-constant PDF: histogram angle <= 0.5 deg,
                          Jensen-Shannon divergence <= 1e-5, output sorted  -> durf_sorted_piecewise_constant_pdf
  * math_test.py:270-295  large flat PDF: Kolmogorov-Smirnov <= 0.2 (within bins and across the range)
  * math_test.py:297-325  sparse delta: half of the samples land in the heavy bin (+-0.05)
  * math_test.py:327-346  single hot bin: every sample inside it, randomized and deterministic
  * math_test.py:41-50    safe_sin / safe_cos accurate to 1e-4 on [-1e10, 1e10], never NaN up to 1e60
                          -> the fp32 encode kernel (durf_encode_bkgd, out_f32), whose sine IS safe_sin

The kernel draws num_samples = N + 1 per ray (what mip.resample_along_rays asks for, mip.py:405-411) with N <= 256, so
the reference's 1e6-sample statistics are gathered over many rays that share one PDF; its deterministic-mode histogram
test needs 1e6 evenly spaced samples of ONE call and is covered instead by tests/test_gpu_stages.py::test_resample
(exact inverse-CDF values against the oracle)."""
import numpy as np
import pytest
import scipy.special
import scipy.stats
import torch

from durf_amd import ops

pytestmark = pytest.mark.gpu


def test_piecewise_constant_pdf_reproduces_distribution(cuda):
    g = torch.Generator().manual_seed(20202020)
    num_bins, precision = 16, 1e5
    rays = 60000                                    # x 17 samples per ray ~ 1e6 samples per PDF, as in the reference
    data = []
    for _ in range(4):
        delta = torch.round(precision * torch.exp(torch.rand(num_bins + 1, generator=g, dtype=torch.float64) * 6 - 3))
        delta = delta * (torch.rand(num_bins + 1, generator=g) < 0.9)            # some zero-width bins
        bins = torch.cumsum(delta, 0) / precision + torch.randn((), generator=g, dtype=torch.float64) * num_bins / 2
        w = torch.clamp(torch.rand(num_bins, generator=g, dtype=torch.float64) * 1.5 - 0.5, min=0)   # some zero weights
        data.append((bins, w, w / w.sum()))
    data.append((data[-1][0], torch.zeros(num_bins, dtype=torch.float64),
                 torch.ones(num_bins, dtype=torch.float64) / num_bins))          # all-zero weights -> uniform
    for bins, w, h_gt in data:
        b32 = bins.float()
        u = torch.rand(rays, num_bins + 1, generator=g)
        s = ops.sorted_piecewise_constant_pdf(b32[None].expand(rays, -1).contiguous().to(cuda),
                                              w.float()[None].expand(rays, -1).contiguous().to(cuda), u.to(cuda))
        assert (s[:, 1:] >= s[:, :-1]).all(), 'samples must be sorted'
        s = s.cpu().double().numpy().reshape(-1)
        b = b32.double().numpy()
        h_gt = h_gt.numpy()
        hist = np.histogram(s, b)[0].astype(np.float64) / s.size
        while np.any(b[:-1] == b[1:]):                                           # merge zero-width bins (math_test.py:245-252)
            j = int(np.where(b[:-1] == b[1:])[0][0])
            hist = np.concatenate([hist[:j], [hist[j] + hist[j + 1]], hist[j + 2:]])
            h_gt = np.concatenate([h_gt[:j], [h_gt[j] + h_gt[j + 1]], h_gt[j + 2:]])
            b = np.concatenate([b[:j], b[j + 1:]])
        angle = 180 / np.pi * np.arccos(min(1.0, np.mean(hist * h_gt) / np.sqrt(np.mean(hist ** 2) * np.mean(h_gt ** 2))))
        m = (hist + h_gt) / 2
        js = np.sum(scipy.special.kl_div(hist, m) + scipy.special.kl_div(h_gt, m)) / 2
        assert angle <= 0.5, angle
        assert js <= 1e-5, js


def test_piecewise_constant_pdf_large_flat(cuda):
    N = 256
    g = torch.Generator().manual_seed(0)
    bins = torch.arange(N + 1, dtype=torch.float32)[None].to(cuda)
    s = ops.sorted_piecewise_constant_pdf(bins, torch.ones(1, N, device=cuda), torch.rand(1, N + 1, generator=g).to(cuda))
    s = s[0].cpu().numpy()
    assert (s >= 0).all() and (s <= N).all()
    assert scipy.stats.kstest(np.mod(s, 1), 'uniform', (0, 1)).statistic <= 0.2
    assert scipy.stats.kstest(s, 'uniform', (0, N)).statistic <= 0.2


def test_piecewise_constant_pdf_sparse_delta(cuda):
    N = 256
    g = torch.Generator().manual_seed(0)
    bins = torch.arange(N + 1, dtype=torch.float32)[None].to(cuda)
    w = torch.ones(1, N)
    di = (N - 1) // 2
    w[0, di] = N - 1                                  # as much mass as all the other bins together
    s = ops.sorted_piecewise_constant_pdf(bins, w.to(cuda), torch.rand(1, N + 1, generator=g).to(cuda))[0].cpu().numpy()
    assert (s >= 0).all() and (s <= N).all()
    assert scipy.stats.kstest(np.mod(s, 1), 'uniform', (0, 1)).statistic <= 0.2
    assert abs(np.mean((s >= di) & (s <= di + 1)) - 0.5) <= 0.05


@pytest.mark.parametrize('randomized', [False, True])
def test_piecewise_constant_pdf_single_bin(cuda, randomized):
    g = torch.Generator().manual_seed(0)
    bins = torch.tensor([0, 1, 3, 6, 10], dtype=torch.float32)
    rays = 125                                        # x 5 samples = the reference's 625
    for i in range(4):
        w = torch.zeros(4)
        w[i] = 1.0
        u = torch.rand(rays, 5, generator=g).to(cuda) if randomized else None
        s = ops.sorted_piecewise_constant_pdf(bins[None].expand(rays, -1).contiguous().to(cuda),
                                              w[None].expand(rays, -1).contiguous().to(cuda), u).cpu()
        assert (s >= bins[i]).all() and (s <= bins[i + 1]).all()


def test_safe_trig_in_the_encode_kernel(cuda):
    """math_test.py:41-50 on the kernel's safe_sin.  Feature 0 of a zero-variance sample is safe_sin(x), feature 30 is
    safe_sin(x + pi/2) (a zero direction and zero radius make the frustum Gaussian a point at the ray origin: mean =
    o + 0 * t_mean, cov = t_var * 0 + r_var(radius = 0) = 0).  The reference test feeds
    float64 numbers; the kernel (like an fp32 JAX run) sees their fp32 roundings and wraps by fp32(100 pi), so the truth
    is sin(x32 mod fp32(100 pi)) evaluated in float64 -- accurate to 1e-4 over [-1e10, 1e10], never NaN up to fp32's
    range (the reference's 1e60 overflows fp32: 1e38 is the far end here)."""
    t32 = np.float64(np.float32(100 * np.pi))
    pi2 = np.float32(np.pi / 2)

    def truth(x32):
        xd = x32.astype(np.float64)
        return np.where(np.abs(xd) < t32, np.sin(xd), np.sin(np.mod(xd, t32)))

    for max_exp, check_err in ((10, True), (38, False)):
        x = 10 ** np.linspace(-30, max_exp, 10000)
        x = np.concatenate([-x[::-1], np.array([0]), x]).astype(np.float32)
        x = x[np.isfinite(x)]
        B = x.size
        xs = np.concatenate([x, np.zeros((-B) % 32, np.float32)])
        Bp = xs.size
        origins = torch.zeros(Bp, 3)
        origins[:, 0] = torch.tensor(xs)
        dirs = torch.zeros(Bp, 3)
        t_vals = torch.tensor([[0.0, 1.0]]).expand(Bp, 2).contiguous()      # N = 1 sample per ray
        _, enc = ops.encode_bkgd(t_vals.to(cuda), origins.to(cuda), dirs.to(cuda), torch.zeros(Bp, device=cuda), None,
                                 contraction=False, tile=False, f32=True)
        enc = enc.cpu().numpy()[:B]
        assert not np.isnan(enc[:, [0, 30]]).any()
        if check_err:
            assert np.max(np.abs(enc[:, 0] - truth(x))) < 1e-4
            assert np.max(np.abs(enc[:, 30] - truth((x + pi2).astype(np.float32)))) < 1e-4
