"""The mixed launches of a small training step (round 6; csrc/mlp_fwd.hip k_mlp_fwd<.., MIX>, csrc/mlp_bwd.hip k_mlp_bwd<.., MIX>;
durf_mlp_fwd_enc_obj / durf_mlp_bwd_obj): the K object MLPs' forward and backward (obbpose_model.py:174-201) ride as (object,
tile pair) items in the background MLP's persistent launches instead of in launches of their own.  Same MFMA instructions,
operands and k order per output, so EVERYTHING a step produces -- rendered outputs of every level, every logged scalar, the
gradient's norm, parameters and Adam moments after three steps -- must be bit-identical with the mix on and off, on both host
paths (the Python-issued launches and the one C call), and the launches must really have been the mixed ones.  The separate
launches are the paths the oracle comparisons of the suite run at most sizes; tests/test_gpu_dispatch_matrix.py runs an oracle
comparison with the mixed variants selected."""
import os

import pytest
import torch

from durf_amd import obbpose_model, ops, synthetic, train_boxpose, utils
from tests import helpers as H

pytestmark = pytest.mark.gpu

NAMES = ('loss', 'losses', 'obj_losses', 'd_losses', 'n_losses', 'e_losses', 's_losses', 'distr_losses', 'tv_losses',
         'sampling_stats', 'psnr', 'psnrs', 'obj_psnr', 'grad_norm', 'grad_abs_max', 'grad_norm_clipped')


def _run(cuda, fn, mix, B, K, N, extra, hit_range, multi):
    os.environ['DURF_OBJ_MIX'] = '1' if mix else '0'
    try:
        utils.clear_gin()
        utils.parse_gin('MipNerfModel.num_samples = %d\nMipNerfModel.density_noise = 0.0\nMipNerfModel.no_pose_opt = True\n'
                        'MipNerfModel.no_yaw_opt = True\nConfig.randomized = True\nConfig.rand_bkgd = False\n'
                        'Config.grad_max_norm = 1.0\nConfig.grad_max_val = 0.1\n' % N + extra)
        config = utils.configured(utils.Config)
        b = synthetic.make_batch(B, K, seed=60 + K, allow_multi_hit=multi, noise_boxes=0.1, hit_range=hit_range)
        db = H.device_batch(b, cuda)
        model, variables = obbpose_model.construct_mipnerf(3, db, device=cuda)
        state = train_boxpose.create_train_state(variables)
        prev = db['init'][0:1] + 0.01
        rng, log = 5, []
        ops.dispatch_reset()
        for _ in range(3):
            state, stats, rng, pose = fn(model, config, rng, state, db, 5e-4, 0.7, 6.5, prev)
            log.append(([getattr(stats, n).clone() for n in NAMES], [w.clone() for w in stats.weights + stats.samples],
                        int(stats.multi_hit_rays)))
        torch.cuda.synchronize()
        return state, log, ops.dispatch_seen(), float(b['hit_fraction'])
    finally:
        os.environ.pop('DURF_OBJ_MIX', None)


@pytest.mark.parametrize('B,K,N,extra,hit_range,multi', [
    (512, 3, 128, '', (0.05, 0.15), False),                  # the reference's batch at the metric's sample count (cfg3 @ 512)
    (1024, 8, 128, '', (0.05, 0.15), True),                  # cfg5's per-rank shape: two rounds of background blocks; multi-hit rays
    (256, 3, 32, '', (0.05, 0.15), False),                   # fewer background blocks than CUs
    (640, 2, 64, 'MipNerfModel.ray_shape = "cylinder"\nMipNerfModel.disable_integration = True\n', (0.3, 0.6), False),   # many hits
    (700, 1, 32, 'MipNerfModel.num_levels = 3\nConfig.white_bkgd = True\n', (0.0, 0.01), False),      # (almost) no hits, 3 levels
])
def test_mixed_launches_leave_a_step_bit_identical(cuda, B, K, N, extra, hit_range, multi):
    results = {}
    for fn in (train_boxpose.train_step, train_boxpose.train_step_one_call):
        for mix in (False, True):
            results[(fn.__name__, mix)] = _run(cuda, fn, mix, B, K, N, extra, hit_range, multi)
    ref_state, ref_log, ref_seen, hit = results[('train_step', False)]
    assert not ({'FWD_MIX', 'BWD_MIX'} & ref_seen) and {'FWD128_MSPLIT', 'BWD128_MSPLIT'} <= ref_seen, ref_seen
    for key, (state, log, seen, _) in results.items():
        if key[1]:
            assert {'FWD_MIX', 'BWD_MIX'} <= seen and not ({'FWD128_MSPLIT', 'BWD128_MSPLIT'} & seen), (key, seen)
        assert torch.equal(state.variables.flat, ref_state.variables.flat), '%s: parameters after 3 steps' % (key,)
        assert torch.equal(state.m, ref_state.m) and torch.equal(state.v, ref_state.v), '%s: Adam moments' % (key,)
        for step, ((sc, ws, mh), (rsc, rws, rmh)) in enumerate(zip(log, ref_log)):
            assert mh == rmh
            for n, a, c in zip(NAMES, sc, rsc):
                assert torch.allclose(a, c, rtol=0, atol=0, equal_nan=True), '%s step %d: %s' % (key, step, n)
            for a, c in zip(ws, rws):
                assert torch.allclose(a, c, rtol=0, atol=0, equal_nan=True), '%s step %d: weights / samples' % (key, step)


def test_the_item_counters_are_left_zeroed(cuda):
    """a mixed launch draws its (object, tile pair) items off an atomic ticket counter that it must leave zeroed for the launch
    that gets the same counter 16 384 launches later: after many steps every counter the ring handed out reads 0 again --
    observed through the results: 40 more steps stay bit-identical to the unmixed path (a counter left non-zero would skip items)"""
    outs = {}
    for mix in (False, True):
        os.environ['DURF_OBJ_MIX'] = '1' if mix else '0'
        try:
            utils.clear_gin()
            utils.parse_gin('MipNerfModel.num_samples = 32\nMipNerfModel.density_noise = 0.0\nMipNerfModel.no_pose_opt = True\n'
                            'MipNerfModel.no_yaw_opt = True\nConfig.randomized = True\n')
            config = utils.configured(utils.Config)
            b = synthetic.make_batch(384, 3, seed=9)
            db = H.device_batch(b, cuda)
            model, variables = obbpose_model.construct_mipnerf(3, db, device=cuda)
            state = train_boxpose.create_train_state(variables)
            rng = 1
            for _ in range(40):
                state, stats, rng, _ = train_boxpose.train_step_one_call(model, config, rng, state, db, 5e-4, 0.7, 6.5, db['init'][0:1])
            torch.cuda.synchronize()
            outs[mix] = (state.variables.flat.clone(), float(stats.loss))
        finally:
            os.environ.pop('DURF_OBJ_MIX', None)
    assert torch.equal(outs[False][0], outs[True][0]) and outs[False][1] == outs[True][1]
