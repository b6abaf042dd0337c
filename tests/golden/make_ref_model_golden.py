#!/usr/bin/env python3
"""Generate tests/golden/ref_model_*.npz: outputs of the REFERENCE's own MipNerfModel.__call__ on seeded inputs.

Build container only.  /root/reference/internal/obbpose_model.py (with mip, mip360, math, box_helpers, utils) is imported
unmodified under the numpy-backed stand-ins of tests/ref_standin.py (float64; PRNG draws replayed from the arrays stored
in the fixture) and run on a seeded synthetic batch with seeded parameters; the per-level outputs are committed so that
machines without /root/reference (the GPU box) can check the oracle and the HIP path against them
(tests/test_golden_ref_model.py).  Inputs are regenerated from the seeds (durf_amd.synthetic.make_batch,
obbpose_model.construct_mipnerf on the CPU generator); a parameter checksum guards against RNG drift.
A fixture is data: inputs' seeds, draws and expected outputs -- no reference text.
    python tests/golden/make_ref_model_golden.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from durf_amd import obbpose_model, synthetic, utils  # noqa: E402
from tests import helpers as H  # noqa: E402
from tests import ref_standin  # noqa: E402

# `model`: MipNerfModel fields, the same names in the reference (obbpose_model.py:45-66) and in the oracle (MODEL_DEFAULTS)
CASES = {
    'ref_model_K3_N32_rand': dict(B=96, K=3, seed=203, randomized=True, white_bkgd=False, alpha=4.5,
                                  model=dict(num_samples=32, density_noise=0.0)),
    'ref_model_K1_N32_det_white': dict(B=64, K=1, seed=202, randomized=False, white_bkgd=True, alpha=10.0,
                                       model=dict(num_samples=32, density_noise=0.0)),
    'ref_model_K2_N32_cylinder_pe': dict(B=48, K=2, seed=204, randomized=True, white_bkgd=False, alpha=10.0,
                                         model=dict(num_samples=32, density_noise=0.0, ray_shape='cylinder',
                                                    disable_integration=True)),
    'ref_model_K2_N32_static_flat': dict(B=48, K=2, seed=205, randomized=False, white_bkgd=False, alpha=10.0,
                                         model=dict(num_samples=32, density_noise=0.0, dynamics=False, contraction=False)),
    # the shape the metric is quoted on: Waymo knobs, K = 3, 128 samples/ray x 2 levels, stratified sampling
    # use_viewdirs = False: the MLP without a condition -- 10 Dense layers, the rgb head straight off the trunk
    # (obbpose_model.py:221-232,336-352); a model with dynamics = False, the only kind the reference can run with the knob off
    # (K = 2 with dynamics = False: the reference indexes box_centers[0] even for a static scene, so it cannot run K = 0)
    'ref_model_K2_N32_static_noview': dict(B=64, K=2, seed=207, randomized=True, white_bkgd=False, alpha=10.0,
                                           model=dict(num_samples=32, density_noise=0.0, use_viewdirs=False, dynamics=False)),
    'ref_model_waymo_K3_N128': dict(B=128, K=3, seed=206, randomized=True, white_bkgd=False, alpha=10.0,
                                    model=dict(num_samples=128, density_noise=0.0)),
}
NAMES = ('rgb', 'distance', 'acc', 'weights', 't_vals')


def build(case):
    """-> numpy batch, Variables (CPU), noise dict (float64 torch)"""
    c = CASES[case]
    B, K, N, seed = c['B'], c['K'], c['model']['num_samples'], c['seed']
    b = synthetic.make_batch(B, K, seed=seed)
    cb = {k: (torch.tensor(v) if isinstance(v, np.ndarray) else v) for k, v in b.items() if k != 'rays'}
    utils.clear_gin()
    if not c['model'].get('use_viewdirs', True):
        utils.parse_gin('MipNerfModel.use_viewdirs = False\n')
    model, variables = obbpose_model.construct_mipnerf(seed, cb, device='cpu')
    utils.clear_gin()
    g = torch.Generator().manual_seed(seed)
    for nm in variables.layout.mlp_names():
        for i in range(len(variables.layout.layer_shapes(nm))):
            bias = variables['params'][nm]['Dense_%d' % i]['bias']
            bias.copy_((torch.rand(bias.shape, generator=g) - 0.5) * 0.1)
    noise = dict(t_rand=torch.rand(B, N + 1, generator=g, dtype=torch.float64),
                 u_rand=torch.rand(B, N + 1, generator=g, dtype=torch.float64))
    return b, variables, noise


def inputs(case):
    """-> oracle batch (float64), oracle params (float64), noise"""
    b, variables, noise = build(case)
    return H.oracle_batch(b, torch.float64), H.oracle_params_from_variables(variables, torch.float64), noise


def reference_outputs(ref, case):
    c = CASES[case]
    ob, params, noise = inputs(case)
    uniforms = [noise['t_rand'].numpy(), noise['u_rand'].numpy()] if c['randomized'] else []
    return ref_standin.run_model(ref, c['model'], params, ob['rays'], ob['ext'], ob['ts'], c['randomized'], c['white_bkgd'],
                                 c['alpha'], uniforms)


def main():
    gold = os.path.join(ROOT, 'tests', 'golden')
    ref = ref_standin.load()
    try:
        for case in CASES:
            out = reference_outputs(ref, case)
            _, variables, noise = build(case)
            flat = variables.flat.double()
            rec = dict(param_checksum=np.array([float(flat.sum()), float((flat * flat).sum())]),
                       t_rand=noise['t_rand'].numpy(), u_rand=noise['u_rand'].numpy())
            for lvl, r in enumerate(out):
                for i, nm in enumerate(NAMES):
                    rec['l%d_%s' % (lvl, nm)] = np.asarray(r[i], dtype=np.float64)
            rec['dyn_mask'] = np.asarray(out[0][8]).reshape(-1).astype(np.int64)
            rec['zo'] = np.asarray(out[0][9], dtype=np.float64)
            path = os.path.join(gold, case + '.npz')
            np.savez_compressed(path, **rec)
            print(case, '%.1f KB' % (os.path.getsize(path) / 1024), 'rgb mean', float(rec['l1_rgb'].mean()))
    finally:
        ref_standin.unload()


if __name__ == '__main__':
    main()
