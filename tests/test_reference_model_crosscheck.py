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
