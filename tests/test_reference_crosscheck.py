"""Transcription check of the oracle against the reference's OWN source text (build container only).

The reference is JAX and neither jax nor flax is installable here, so its functions cannot run as shipped and the oracle
(oracle/durf_ref.py) stays "parity unpinned" (DESIGN.md 2).  What CAN be done in the build container, where
/root/reference exists: its pure `jnp` functions -- internal/math.py, mip.py, mip360.py, box_helpers.py, which use nothing
of JAX but array arithmetic, `lax.stop_gradient`, `jax.linearize` and `jax.random.uniform` -- are imported UNMODIFIED
from /root/reference with a numpy-backed stand-in registered as the `jax` module (float64; linearize by central
differences; uniform draws replayed from the arrays the oracle is handed), and every one is compared with its
restatement in the oracle on the same random inputs.  A stand-in library pins nothing about JAX's arithmetic -- it
catches a mistyped formula, a swapped argument, a wrong axis.  Nothing here travels to the GPU box: the test is skipped
where /root/reference is absent, and no reference text is copied into the repo."""
import importlib
import os
import sys
import types

import numpy as np
import pytest
import torch

REF = '/root/reference'
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, 'internal')), reason='reference tree not present')

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import durf_ref as R  # noqa: E402


class _Uniform:
    """jax.random.uniform(key, shape, minval=0, maxval=1): replays queued U[0,1) arrays"""
    queue = []

    @classmethod
    def uniform(cls, key, shape, dtype=None, minval=0.0, maxval=1.0):
        u = cls.queue.pop(0)
        assert list(u.shape) == list(shape), (u.shape, shape)
        return u * (maxval - minval) + minval


class _JArr(np.ndarray):
    """jnp arrays are immutable: `u += x` rebinds u to a NEW (broadcast) array where numpy would write in place"""
    __iadd__ = lambda self, o: self + o
    __isub__ = lambda self, o: self - o
    __imul__ = lambda self, o: self * o
    __itruediv__ = lambda self, o: self / o


def _install_numpy_jax():
    jnp = types.ModuleType('jax.numpy')
    for name in dir(np):
        if not name.startswith('_'):
            setattr(jnp, name, getattr(np, name))
    jnp.ndarray = np.ndarray
    jnp.arange = lambda *a, **k: np.arange(*a, **k).view(_JArr)
    jnp.matmul = lambda a, b, precision=None: np.matmul(a, b)
    jnp.linalg = np.linalg
    jnp.float32 = np.float64                       # the check runs in float64 on both sides
    jnp.array = lambda x, dtype=None: np.array(x, dtype=np.float64 if dtype in (None, np.float64) else dtype)
    lax = types.ModuleType('jax.lax')
    lax.stop_gradient = lambda x: x
    lax.Precision = types.SimpleNamespace(HIGHEST=None)
    random = types.ModuleType('jax.random')
    random.uniform = _Uniform.uniform
    random.normal = lambda key, shape: np.zeros(shape)
    random.randint = lambda key, shape, lo, hi: np.zeros(shape)          # randint(0, 1) == 0 (mip.py:324)
    jax = types.ModuleType('jax')

    def linearize(f, x):
        def jvp(t, h=1e-6):
            return (f(x + h * t) - f(x - h * t)) / (2 * h)
        return f(x), jvp
    jax.linearize = linearize
    jax.vmap = lambda f, in_axes=0, out_axes=0: f
    jax.numpy, jax.lax, jax.random = jnp, lax, random
    jsp = types.ModuleType('jax.scipy')
    jax.scipy = jsp
    sys.modules.update({'jax': jax, 'jax.numpy': jnp, 'jax.lax': lax, 'jax.random': random, 'jax.scipy': jsp})


@pytest.fixture(scope='module')
def ref():
    saved = {k: sys.modules.get(k) for k in ('jax', 'jax.numpy', 'jax.lax', 'jax.random', 'jax.scipy', 'internal')}
    _install_numpy_jax()
    sys.path.insert(0, REF)
    for k in [k for k in sys.modules if k == 'internal' or k.startswith('internal.')]:
        del sys.modules[k]
    mods = types.SimpleNamespace(**{n: importlib.import_module('internal.' + n) for n in ('math', 'mip', 'mip360', 'box_helpers')})
    yield mods
    sys.path.remove(REF)
    for k in [k for k in sys.modules if k == 'internal' or k.startswith('internal.')]:
        del sys.modules[k]
    for k, v in saved.items():
        if v is None:
            sys.modules.pop(k, None)
        else:
            sys.modules[k] = v


def T(x):
    return torch.tensor(np.asarray(x), dtype=torch.float64)


def close(got, want, tol=1e-9, what=''):
    got, want = np.asarray(got, dtype=np.float64), want.detach().numpy() if isinstance(want, torch.Tensor) else np.asarray(want)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    fin = np.isfinite(want)
    assert (np.isfinite(got) == fin).all(), what
    err = np.abs(got[fin] - want[fin]).max() if fin.any() else 0.0
    assert err <= tol * max(1.0, np.abs(want[fin]).max() if fin.any() else 1.0), (what, err)


def test_math(ref):
    g = np.random.default_rng(0)
    x = np.concatenate([g.uniform(-400, 400, 2000), g.uniform(-1e4, 1e4, 200)])
    close(ref.math.safe_sin(x), R.safe_sin(T(x)), what='safe_sin')
    close(ref.math.safe_cos(x), R.safe_cos(T(x)), what='safe_cos')
    v = np.concatenate([g.normal(size=(50, 3)), np.zeros((2, 3)), 1e-8 * g.normal(size=(3, 3))])
    close(ref.math.safe_norm(v), R.safe_norm(T(v)), what='safe_norm')
    close(ref.math.mse_to_psnr(np.array([0.3, 1e-3])), R.mse_to_psnr(T([0.3, 1e-3])), what='mse_to_psnr')
    for step in (0, 1, 77, 2500, 9999, 250000):
        a = ref.math.learning_rate_decay(step, 5e-4, 5e-6, 250000, 2500, 0.01)
        b = R.learning_rate_decay(step, 5e-4, 5e-6, 250000, 2500, 0.01)
        assert abs(float(a) - float(b)) <= 1e-12 * abs(float(b)), ('lr', step, a, b)
        a = ref.math.freq_alpha_rate(step, 0.0, 10.0, 2000, 100000)
        b = R.freq_alpha_rate(step, 0.0, 10.0, 2000, 100000)
        assert abs(float(a) - float(b)) <= 1e-12, ('alpha', step, a, b)


@pytest.mark.parametrize('randomized', [False, True])
def test_sorted_piecewise_constant_pdf(ref, randomized):
    g = np.random.default_rng(1)
    B, N = 13, 24
    bins = np.sort(g.uniform(0, 40, (B, N + 1)), -1)
    w = g.uniform(0, 1, (B, N)) * (g.uniform(0, 1, (B, N)) < 0.4)
    w[0] = 0.0                                            # the zero-weight padding path
    u = g.uniform(0, 1, (B, N + 1))
    _Uniform.queue = [u.copy()]
    got = ref.math.sorted_piecewise_constant_pdf(None, bins.copy(), w.copy(), N + 1, randomized)
    want = R.sorted_piecewise_constant_pdf(T(u), T(bins), T(w), N + 1, randomized)
    # (the oracle forms u = linspace(0, 1 - eps_f32) in float32, as JAX with x64 off does; the stand-in in float64: a
    # 6e-8 difference in u moves a sample by up to ~3e-7 of the range)
    close(got, want, tol=1e-6, what='sorted_piecewise_constant_pdf')


def test_box_helpers(ref):
    g = np.random.default_rng(2)
    B, K = 40, 3
    rot = np.concatenate([g.normal(size=(K - 1, 3)), np.zeros((1, 3))])   # incl. the zero rotation vector
    close(ref.box_helpers.aa2matrix(rot), R.aa2matrix(T(rot)), what='aa2matrix')
    mats = np.broadcast_to(ref.box_helpers.aa2matrix(rot), (B, K, 3, 3))
    p = g.normal(size=(B, K, 3))
    close(ref.box_helpers.rotate_matrix(p, mats), R.rotate_matrix(T(p), T(mats)), what='rotate_matrix')
    o, d = g.uniform(-1, 1, (B, 3)), g.normal(size=(B, 3))
    pose = np.broadcast_to(g.uniform(-4, 4, (K, 3)), (B, K, 3))
    oo, do = ref.box_helpers.world2object_rpy(o, d, pose, mats)
    oo2, do2 = R.world2object_rpy(T(o), T(d), T(pose), T(mats))
    close(oo, oo2, what='world2object_rpy origins')
    close(do, do2, what='world2object_rpy dirs')
    dims = np.broadcast_to(np.array([0.6, 0.5, 1.2]), (B, K, 3))
    do[0, 0, 1] = 0.0                                      # a zero direction component: IEEE infinities in the slab test
    zi, zo, hit = ref.box_helpers.ray_box_intersection(oo * 0.2, do, -dims, dims)
    zi2, zo2, hit2 = R.ray_box_intersection(T(oo * 0.2), T(do), T(-dims), T(dims))
    assert hit.sum() > 0
    np.testing.assert_array_equal(np.asarray(hit), hit2.numpy())
    close(zi, zi2, what='z_in')
    close(zo, zo2, what='z_out')


def _rays(g, B):
    o = g.uniform(-0.5, 0.5, (B, 3))
    d = g.normal(size=(B, 3)) * g.uniform(0.5, 1.5, (B, 1))
    return o, d, g.uniform(5e-4, 2e-3, (B, 1))


@pytest.mark.parametrize('ray_shape', ['cone', 'cylinder'])
def test_cast_rays_and_encodings(ref, ray_shape):
    g = np.random.default_rng(3)
    B, N = 9, 16
    o, d, r = _rays(g, B)
    t = np.sort(g.uniform(0, 40, (B, N + 1)), -1)
    mean, cov = ref.mip.cast_rays(t, o, d, r, ray_shape)
    mean2, cov2 = R.cast_rays(T(t), T(o), T(d), T(r), ray_shape)
    close(mean, mean2, what='cast_rays mean')
    close(cov, cov2, tol=1e-8, what='cast_rays cov')
    close(ref.mip.integrated_pos_enc((mean, cov), 0, 10), R.integrated_pos_enc((mean2, cov2), 0, 10), tol=1e-8, what='integrated_pos_enc')
    for alpha in (0.0, 3.3, 10.0):
        close(ref.mip.weighted_ipe((mean, cov), 0, 10, alpha), R.weighted_ipe((mean2, cov2), 0, 10, alpha), tol=1e-8,
              what='weighted_ipe alpha=%g' % alpha)
    # contraction (threshold 0.1, sign flip on (0.1, 0.5)) and its push-forward of the covariance
    scaled = mean * np.array([0.0, 0.01, 0.05, 0.2, 1.0, 3.0, 0.3, 0.004, 1e-9])[:, None, None]
    close(ref.mip360.contract(scaled), R.contract(T(scaled)), what='contract')
    mc, cc = ref.mip360.new_space((scaled, cov))
    mc2, cc2 = R.new_space((T(scaled), cov2))
    close(mc, mc2, what='new_space mean')
    close(cc, cc2, tol=1e-6, what='new_space cov')        # the stand-in linearises by central differences
    v = g.normal(size=(B, 3))
    close(ref.mip.pos_enc(v, 0, 4, True), R.pos_enc(T(v), 0, 4, True), what='pos_enc')


@pytest.mark.parametrize('randomized', [False, True])
def test_sampling_and_rendering(ref, randomized):
    g = np.random.default_rng(4)
    B, N = 7, 32
    o, d, r = _rays(g, B)
    near, far = np.zeros((B, 1)), np.full((B, 1), 40.0)
    t_rand, u_rand = g.uniform(0, 1, (B, N + 1)), g.uniform(0, 1, (B, N + 1))
    _Uniform.queue = [t_rand.copy()]
    t, (m, c) = ref.mip.sample_along_rays(None, o, d, r, N, near, far, randomized, False, 'cone')
    t2, (m2, c2) = R.sample_along_rays(T(t_rand), T(o), T(d), T(r), N, T(near), T(far), randomized, False, 'cone')
    close(t, t2, tol=1e-6, what='sample_along_rays t_vals')          # the oracle's linspace is formed in float32
    rgb, dens = g.uniform(0, 1, (B, N, 3)), g.uniform(0, 2, (B, N, 1)) * (g.uniform(0, 1, (B, N, 1)) < 0.5)
    out = ref.mip.volumetric_rendering(rgb, dens, np.asarray(t), d, False, False, None)
    out2 = R.volumetric_rendering(T(rgb), T(dens), T(np.asarray(t)), T(d), False, False)
    for i, nm in enumerate(('comp_rgb', 'depth', 'acc', 'weights', 't_vals', 't_mids', 't_dists')):
        close(out[i], out2[i], what='volumetric_rendering ' + nm)
    w = np.asarray(out[3])
    _Uniform.queue = [u_rand.copy()]
    tn, _ = ref.mip.resample_along_rays(None, o, d, r, np.asarray(t), w.copy(), randomized, 'cone', True, 0.01)
    tn2, _ = R.resample_along_rays(T(u_rand), T(o), T(d), T(r), T(np.asarray(t)), T(w), randomized, True, 0.01, 'cone')
    close(tn, tn2, tol=1e-6, what='resample_along_rays')
