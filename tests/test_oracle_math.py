"""The framework-agnostic assertions of the reference's only test file
(internal/math_test.py:41-115,183-346) re-expressed against the CPU oracle: safe trig range
handling, PSNR round trip, the two learning-rate-schedule properties, and the four statistical
properties of sorted_piecewise_constant_pdf.  These are the only reference-held checks that
pin anything on the hot path."""
import numpy as np
import pytest
import scipy.special
import scipy.stats
import torch

from durf_amd import math as dmath
from oracle import durf_ref as R


def _trig_harness(fn, max_exp):
    x = 10 ** np.linspace(-30, max_exp, 10000)
    x = np.concatenate([-x[::-1], np.array([0]), x])
    y_true = getattr(np, fn)(x)
    y = getattr(R, 'safe_' + fn)(torch.tensor(x, dtype=torch.float64)).numpy()
    return y_true, y


def test_safe_trig_accurate_to_1e10_and_never_nan():
    """math_test.py:41-50."""
    for fn in ('sin', 'cos'):
        y_true, y = _trig_harness(fn, 10)
        assert np.max(np.abs(y - y_true)) < 1e-4
        assert not np.isnan(y).any()
        _, y = _trig_harness(fn, 60)
        assert not np.isnan(y).any()


def test_psnr_round_trip():
    """math_test.py:52-55."""
    assert abs(float(R.psnr_to_mse(R.mse_to_psnr(0.07))) - 0.07) < 1e-7
    assert abs(dmath.psnr_to_mse(dmath.mse_to_psnr(0.07)) - 0.07) < 1e-12


@pytest.mark.parametrize('impl', [R.learning_rate_decay, dmath.learning_rate_decay])
def test_learning_rate_decay(impl):
    """math_test.py:57-80: endpoints, geometric mean at the middle, clamped past the end."""
    rs = np.random.RandomState(0)
    for _ in range(10):
        lr_init = np.exp(rs.normal() - 3)
        lr_final = lr_init * np.exp(rs.normal() - 5)
        max_steps = int(np.ceil(100 + 100 * np.exp(rs.normal())))
        f = lambda s: impl(s, lr_init, lr_final, max_steps)
        np.testing.assert_allclose(f(0), lr_init, rtol=1e-6)
        np.testing.assert_allclose(f(max_steps), lr_final, rtol=1e-6)
        np.testing.assert_allclose(f(max_steps / 2), np.sqrt(lr_init * lr_final), rtol=1e-6)
        np.testing.assert_allclose(f(max_steps + 100), lr_final, rtol=1e-6)


@pytest.mark.parametrize('impl', [R.learning_rate_decay, dmath.learning_rate_decay])
def test_delayed_learning_rate_decay(impl):
    """math_test.py:82-115."""
    rs = np.random.RandomState(0)
    for _ in range(10):
        lr_init = np.exp(rs.normal() - 3)
        lr_final = lr_init * np.exp(rs.normal() - 5)
        max_steps = int(np.ceil(100 + 100 * np.exp(rs.normal())))
        delay = int(rs.uniform(low=0.1, high=0.4) * max_steps)
        mult = np.exp(rs.normal() - 3)
        f = lambda s: impl(s, lr_init, lr_final, max_steps, delay, mult)
        np.testing.assert_allclose(f(0), mult * lr_init, rtol=1e-6)
        np.testing.assert_allclose(f(max_steps), lr_final, rtol=1e-6)
        np.testing.assert_allclose(f(delay), impl(delay, lr_init, lr_final, max_steps), rtol=1e-6)
        np.testing.assert_allclose(f(max_steps / 2), np.sqrt(lr_init * lr_final), rtol=1e-6)
        np.testing.assert_allclose(f(max_steps + 100), lr_final, rtol=1e-6)


def test_freq_alpha_rate():
    """math.py:193-219: constant, linear ramp, saturated."""
    for impl in (R.freq_alpha_rate, dmath.freq_alpha_rate):
        assert impl(5, 0.0, 10.0, 10, 110) == 0.0
        assert abs(impl(60, 0.0, 10.0, 10, 110) - 5.0) < 1e-12
        assert impl(500, 0.0, 10.0, 10, 110) == 10.0
        # configs/*.gin:33-36 (init=final=10, delay 0, max 1): the ramp formula gives 0 at step 0
        # and alpha_final from step 1 on -- training starts at step 1 (train_boxpose.py:406,420)
        assert impl(0, 10.0, 10.0, 0, 1) == 0.0
        assert impl(1, 10.0, 10.0, 0, 1) == 10.0


def test_piecewise_constant_pdf_reproduces_distribution():
    """math_test.py:183-268: 4 random 16-bin PDFs (some zero-width bins, some zero weights) +
    an all-zero weight vector; sorted output; histogram angle <= 0.5 deg, JS divergence <= 1e-5."""
    g = torch.Generator().manual_seed(20202020)
    num_bins, num_samples, precision = 16, 1000000, 1e5
    data = []
    for _ in range(4):
        delta = torch.round(precision * torch.exp(torch.rand(num_bins + 1, generator=g, dtype=torch.float64) * 6 - 3))
        delta = delta * (torch.rand(num_bins + 1, generator=g) < 0.9)
        bins = torch.cumsum(delta, 0) / precision + torch.randn((), generator=g, dtype=torch.float64) * num_bins / 2
        w = torch.clamp(torch.rand(num_bins, generator=g, dtype=torch.float64) * 1.5 - 0.5, min=0)
        data.append((bins, w, w / w.sum()))
    data.append((data[-1][0], torch.zeros(num_bins, dtype=torch.float64),
                 torch.ones(num_bins, dtype=torch.float64) / num_bins))
    bins, weights, gt = [torch.stack(x) for x in zip(*data)]
    for randomized in (True, False):
        u = torch.rand(5, num_samples, generator=g, dtype=torch.float64)
        out = []
        for i in range(5):       # one PDF at a time keeps the [bins, samples] mask small
            out.append(R.sorted_piecewise_constant_pdf(u[i:i + 1], bins[i:i + 1], weights[i:i + 1],
                                                       num_samples, randomized))
        samples = torch.cat(out)
        assert samples.shape[-1] == num_samples
        assert (samples[..., 1:] >= samples[..., :-1]).all()
        for s, b, h_gt in zip(samples.numpy(), bins.numpy(), gt.numpy()):
            hist = np.histogram(s, b)[0].astype(np.float64) / num_samples
            while np.any(b[:-1] == b[1:]):
                j = int(np.where(b[:-1] == b[1:])[0][0])
                hist = np.concatenate([hist[:j], [hist[j] + hist[j + 1]], hist[j + 2:]])
                h_gt = np.concatenate([h_gt[:j], [h_gt[j] + h_gt[j + 1]], h_gt[j + 2:]])
                b = np.concatenate([b[:j], b[j + 1:]])
            angle = 180 / np.pi * np.arccos(min(1.0, np.mean(hist * h_gt) /
                                                np.sqrt(np.mean(hist ** 2) * np.mean(h_gt ** 2))))
            m = (hist + h_gt) / 2
            js = np.sum(scipy.special.kl_div(hist, m) + scipy.special.kl_div(h_gt, m)) / 2
            assert angle <= 0.5
            assert js <= 1e-5


def test_piecewise_constant_pdf_large_flat():
    """math_test.py:270-295."""
    num_samples, num_bins = 100, 100000
    g = torch.Generator().manual_seed(0)
    bins = torch.arange(num_bins, dtype=torch.float32)
    weights = torch.ones(num_bins - 1)
    s = R.sorted_piecewise_constant_pdf(torch.rand(1, num_samples, generator=g), bins[None], weights[None],
                                        num_samples, True)[0].numpy()
    assert (s >= 0).all() and (s <= num_bins - 1).all()
    assert scipy.stats.kstest(np.mod(s, 1), 'uniform', (0, 1)).statistic <= 0.2
    assert scipy.stats.kstest(s, 'uniform', (0, num_bins - 1)).statistic <= 0.2


def test_piecewise_constant_pdf_sparse_delta():
    """math_test.py:297-325: half of the mass in one bin."""
    num_samples, num_bins = 100, 100000
    g = torch.Generator().manual_seed(0)
    bins = torch.arange(num_bins, dtype=torch.float32)
    weights = torch.ones(num_bins - 1)
    di = (num_bins - 1) // 2
    weights[di] = num_bins - 2
    s = R.sorted_piecewise_constant_pdf(torch.rand(1, num_samples, generator=g), bins[None], weights[None],
                                        num_samples, True)[0].numpy()
    assert (s >= 0).all() and (s <= num_bins - 1).all()
    assert scipy.stats.kstest(np.mod(s, 1), 'uniform', (0, 1)).statistic <= 0.2
    assert abs(np.mean((s >= di) & (s <= di + 1)) - 0.5) <= 0.05


def test_piecewise_constant_pdf_single_bin():
    """math_test.py:327-346: one-hot weights -> every sample inside the hot bin, both modes."""
    g = torch.Generator().manual_seed(0)
    bins = torch.tensor([0, 1, 3, 6, 10], dtype=torch.float32)
    for randomized in (False, True):
        for i in range(4):
            w = torch.zeros(4)
            w[i] = 1.0
            s = R.sorted_piecewise_constant_pdf(torch.rand(1, 625, generator=g), bins[None], w[None], 625,
                                                randomized)[0]
            assert (s >= bins[i]).all() and (s <= bins[i + 1]).all()


def test_philox_restatement_reproduces_the_generators_known_answer_vectors():
    """oracle/philox_ref.py (the checker of the in-kernel sampling draws, tests/test_gpu_sampling_noise.py) against the
    published Random123 vectors for philox4x32 with 10 rounds, and the shape / range of the draws it derives from them"""
    import numpy as np
    from oracle import philox_ref as P
    for ctr, key, want in P.KAT:
        got = P.philox4x32_10(np.array([ctr], dtype=np.uint32), np.array([key], dtype=np.uint32))[0]
        assert [int(x) for x in got] == list(want)
    t, u = P.step_draws((7 << 32) | 9, 33, 64)
    assert t.shape == u.shape == (33, 65) and t.dtype == np.float32
    assert 0.0 <= t.min() and t.max() < 1.0 and 0.0 <= u.min() and u.max() < 1.0
    t2, _ = P.step_draws((7 << 32) | 10, 33, 64)
    assert not np.array_equal(t, t2)
    # counter i is the flat sample position: a larger batch extends the stream, it does not reshuffle it
    tb, ub = P.step_draws((7 << 32) | 9, 66, 64)
    assert np.array_equal(tb.reshape(-1)[:t.size], t.reshape(-1)) and np.array_equal(ub.reshape(-1)[:u.size], u.reshape(-1))


def test_density_noise_draws_of_the_philox_restatement_are_standard_normal():
    """oracle/philox_ref.density_draws (the checker of durf_density_noise): moments of N(0, 1), independent across levels and
    of the sampling draws made under the same key (their counters differ in word 1)"""
    import numpy as np
    from oracle import philox_ref as P
    n = 1 << 18
    z0, z1 = P.density_draws(20200823, n, 0).astype(np.float64), P.density_draws(20200823, n, 1).astype(np.float64)
    for z in (z0, z1):
        assert abs(z.mean()) < 4 / n ** 0.5 and abs(z.std() - 1) < 0.01
        assert abs((z ** 3).mean()) < 0.05 and abs((z ** 4).mean() - 3) < 0.1
        assert np.isfinite(z).all() and np.abs(z).max() < 5.78          # sqrt(-2 ln 2^-24) = 5.768
    t, u = P.step_draws(20200823, 1 << 10, 255)
    lim = 4 / n ** 0.5
    assert abs(np.corrcoef(z0, z1)[0, 1]) < lim and abs(np.corrcoef(z0, t.ravel())[0, 1]) < lim
    assert abs(np.corrcoef(z0, u.ravel())[0, 1]) < lim
    assert np.array_equal(P.density_draws(20200823, 1000, 1), P.density_draws(20200823, n, 1)[:1000].astype(np.float32))
